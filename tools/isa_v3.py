"""Instruction mix of the env loops of k_raster_v3<OBJ=0> / k_raster_v3dr<OBJ=0> from the compiler's ISA (no GPU needed).

    python tools/isa_v3.py [render.hip]        # compiles to /tmp/t/render.s with the library's flags

Per inner-most loop of the kernel that stores frames (buffer_store): VALU / SALU / VMEM / DS instruction counts of the loop body and,
as every iteration shades PPT = 4 pixels per lane, VALU per pixel."""
import collections, os, re, subprocess, sys
src = sys.argv[1] if len(sys.argv) > 1 else "/root/repo/gym-duckietown_amd/csrc/render.hip"
os.makedirs("/tmp/t", exist_ok=True)
if not os.environ.get("ISA_REUSE"): subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-I/root/repo/include", "-I/root/repo/gym-duckietown_amd/csrc",
                "-S", "--cuda-device-only", "-o", "/tmp/t/render.s", src] + sys.argv[2:], check=True, stderr=subprocess.DEVNULL)
txt = open("/tmp/t/render.s").read().split("\n")
for kern in ("k_raster_v3ILb0E", "k_raster_v3drILb0E", "k_raster_v3ILb1E", "k_raster_v3drILb1E"):
    starts = [i for i, l in enumerate(txt) if re.match(rf"^_ZN\d+_GLOBAL__N_1\d+{kern}\S*:", l)]
    if not starts:
        continue
    s = starts[0]
    e = next(i for i in range(s, len(txt)) if txt[i].startswith(".Lfunc_end"))
    body = txt[s:e]
    meta = [l for l in txt[e:e + 80] if re.search(r"NumVgprs|Occupancy|ScratchSize|NumSgprs", l)][:5]
    print(f"== {kern}: {len([l for l in body if l.strip().startswith('v_')])} VALU lines in the kernel;", " ".join(m.strip("; ").strip() for m in meta))
    # basic blocks: label -> lines
    blocks, cur = collections.OrderedDict(), None
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1); blocks[cur] = []
        elif cur and l.strip() and not l.strip().startswith((";", ".")):
            blocks[cur].append(l.strip())
    # loops = maximal runs of blocks annotated "in Loop: Header=BBx_y" -- approximate: group blocks by back edges
    labels = list(blocks)
    idx = {b: i for i, b in enumerate(labels)}
    loops = []
    for b, ls in blocks.items():
        for l in ls:
            m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.match(r"s_branch (\.LBB\d+_\d+)", l)
            if m and m.group(1) in idx and idx[m.group(1)] <= idx[b]:
                loops.append((idx[m.group(1)], idx[b]))
    for lo, hi in sorted(set(loops)):
        ls = [l for b in labels[lo:hi + 1] for l in blocks[b]]
        ops = collections.Counter(l.split()[0] for l in ls)
        if not any(o.startswith("buffer_store") for o in ops):
            continue
        if any(lo2 >= lo and hi2 <= hi and (lo2, hi2) != (lo, hi) and any(o.startswith("buffer_store") for l in [x for b in labels[lo2:hi2 + 1] for x in blocks[b]] for o in [l.split()[0]]) for lo2, hi2 in loops):
            continue                                   # an outer loop of a storing loop
        v = sum(c for o, c in ops.items() if o.startswith("v_"))
        print(f"  loop {labels[lo]}..{labels[hi]}: {len(ls)} instructions, VALU {v} ({v / 4:.1f} per pixel), SALU {sum(c for o, c in ops.items() if o.startswith('s_'))}, "
              f"VMEM {sum(c for o, c in ops.items() if o.startswith(('global_', 'buffer_')))}, DS {sum(c for o, c in ops.items() if o.startswith('ds_'))}")
        print("     ", ", ".join(f"{o} {c}" for o, c in sorted(ops.items(), key=lambda kv: -kv[1]) if o.startswith("v_"))[:600])
