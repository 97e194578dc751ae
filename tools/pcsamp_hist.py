"""Histograms of rocprofv3 PC samples for one kernel: python tools/pcsamp_hist.py <output dir> <kernel regex>

Reads the *pc_sampling*.csv (+ *kernel_trace*.csv for dispatch -> kernel) that rocprofv3 --pc-sampling-beta-enabled wrote and,
when present, the .json of the same run (it carries the code-object offset of every sample, which the CSV does not).
Prints: totals (issued / not issued), not-issued samples by reason, samples by instruction type, the instructions with most
samples (with their issued share, mean active lanes and leading stall reasons) and -- from the JSON -- the same per code-object
offset, so that the histogram can be laid over the ISA listing (llvm-objdump of the code object)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def col(row, *names):
    for n in names:
        for k in row:
            if k and k.lower() == n.lower():
                return row[k]
    return None


def popcount_hex(s):
    try:
        return bin(int(s, 0) if s.lower().startswith("0x") else int(s)).count("1")
    except Exception:
        return -1


def from_csv(d, kre):
    traces = find(d, "*kernel_trace*.csv")
    samples = [f for f in find(d, "*pc_sampling*.csv") if "stats" not in os.path.basename(f)]
    if not samples:
        print("no pc-sampling csv under", d)
        return False
    disp = {}
    for f in traces:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                disp[col(row, "Dispatch_Id")] = col(row, "Kernel_Name") or ""
    rx = re.compile(kre)
    n_all = n_k = 0
    issued = collections.Counter()
    reason = collections.Counter()
    itype = collections.Counter()
    per_inst = collections.defaultdict(lambda: [0, 0, 0, collections.Counter()])   # samples, issued, lanes, reasons
    header = None
    for f in samples:
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            header = rd.fieldnames
            for row in rd:
                n_all += 1
                kname = disp.get(col(row, "Dispatch_Id"), "")
                if disp and not rx.search(kname):
                    continue
                n_k += 1
                iss = (col(row, "Wave_Issued_Instruction") or "").strip().lower() in ("1", "true", "yes")
                rs = (col(row, "Stall_Reason") or "").strip()
                it = (col(row, "Instruction_Type") or "").strip()
                ins = (col(row, "Instruction") or "").strip()
                cm = (col(row, "Instruction_Comment") or "").strip()
                issued[iss] += 1
                if not iss:
                    reason[rs] += 1
                itype[(it, iss)] += 1
                e = per_inst[(ins, cm)]
                e[0] += 1
                e[1] += iss
                e[2] += max(popcount_hex(col(row, "Exec_Mask") or "0"), 0)
                if not iss:
                    e[3][rs] += 1
    print(f"csv columns: {header}")
    print(f"{n_k} samples of kernels matching /{kre}/ ({n_all} in all, {len(disp)} dispatches traced)")
    if not n_k:
        return False
    print(f"\nissued: {issued[True]} ({100.0 * issued[True] / n_k:.1f} %)   not issued: {issued[False]} ({100.0 * issued[False] / n_k:.1f} %)")
    print("\nnot-issued samples by reason:")
    for r, c in reason.most_common():
        print(f"  {c:8d}  {100.0 * c / max(issued[False], 1):5.1f} %  {r}")
    print("\nsamples by instruction type (issued / not issued):")
    types = sorted({t for t, _ in itype})
    for t in sorted(types, key=lambda t: -(itype[(t, True)] + itype[(t, False)])):
        a, b = itype[(t, True)], itype[(t, False)]
        print(f"  {a + b:8d}  {100.0 * (a + b) / n_k:5.1f} %  issued {a:7d}  not {b:7d}  {t}")
    print("\ninstructions with most samples (text [comment]: samples, share, issued share, mean active lanes, leading reasons):")
    for (ins, cm), e in sorted(per_inst.items(), key=lambda kv: -kv[1][0])[:90]:
        top = ", ".join(f"{r} {c}" for r, c in e[3].most_common(3))
        print(f"  {e[0]:7d} {100.0 * e[0] / n_k:5.2f} %  iss {100.0 * e[1] / e[0]:5.1f} %  lanes {e[2] / e[0]:4.1f}  {ins[:70]:70s} [{cm[-40:]}]  {top}")
    return True


def flatten(o, pre, out):
    if isinstance(o, dict):
        for k, v in o.items():
            flatten(v, pre + "." + k if pre else k, out)
    else:
        out[pre] = o


def find_sample_lists(o, path, acc):
    if isinstance(o, dict):
        for k, v in o.items():
            find_sample_lists(v, path + "/" + k, acc)
    elif isinstance(o, list):
        if o and isinstance(o[0], dict) and "pc_sampl" in path.lower():
            acc.append((path, o))
        else:
            for i, v in enumerate(o[:4]):
                find_sample_lists(v, path + f"[{i}]", acc)


def from_json(d, kre):
    files = find(d, "*.json")
    if not files:
        print("\n(no json output)")
        return
    f = files[0]
    sz = os.path.getsize(f)
    print(f"\njson: {f} ({sz / 1e6:.1f} MB)")
    if sz > 3e9:
        print("too large to load")
        return
    with open(f) as fh:
        J = json.load(fh)
    acc = []
    find_sample_lists(J, "", acc)
    strings = None

    def find_strings(o):
        nonlocal strings
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "strings" and isinstance(v, dict):
                    strings = v
                else:
                    find_strings(v)
        elif isinstance(o, list):
            for v in o[:2]:
                find_strings(v)
    find_strings(J)
    insts = []
    if strings:
        for k, v in strings.items():
            if "instruction" in k.lower() and isinstance(v, list):
                insts = v
    # dispatch -> kernel name from the kernel_dispatch records + kernel symbols, if present
    for path, lst in acc:
        flat0 = {}
        flatten(lst[0], "", flat0)
        print(f"sample list {path}: {len(lst)} records; keys of the first: {sorted(flat0)[:40]}")
        per_pc = collections.defaultdict(lambda: [0, 0, collections.Counter()])
        for rec in lst:
            fl = {}
            flatten(rec, "", fl)
            off = next((v for k, v in fl.items() if k.endswith("code_object_offset")), None)
            cid = next((v for k, v in fl.items() if k.endswith("code_object_id")), None)
            iss = next((v for k, v in fl.items() if k.endswith("wave_issued")), None)
            rs = next((v for k, v in fl.items() if k.endswith("reason_not_issued")), None)
            ii = next((v for k, v in fl.items() if k.endswith("inst_index")), None)
            e = per_pc[(cid, off, ii)]
            e[0] += 1
            e[1] += 1 if iss else 0
            if not iss:
                e[2][rs] += 1
        tot = sum(e[0] for e in per_pc.values())
        print(f"per code-object offset (all kernels of the run; {tot} samples), by offset within the 40 hottest 256-byte windows:")
        win = collections.Counter()
        for (cid, off, ii), e in per_pc.items():
            if off is not None:
                win[(cid, off // 256)] += e[0]
        hot = {w for w, _ in win.most_common(40)}
        for (cid, off, ii), e in sorted(per_pc.items(), key=lambda kv: (str(kv[0][0]), kv[0][1] or 0)):
            if off is None or (cid, off // 256) not in hot:
                continue
            text = insts[ii] if isinstance(ii, int) and 0 <= ii < len(insts) else ""
            top = ", ".join(f"{r} {c}" for r, c in e[2].most_common(3))
            print(f"  co {cid} +0x{off:06x}  {e[0]:7d} {100.0 * e[0] / tot:5.2f} %  iss {100.0 * e[1] / e[0]:5.1f} %  {str(text)[:64]:64s} {top}")


if __name__ == "__main__":
    d, kre = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else ".")
    ok = from_csv(d, kre)
    try:
        from_json(d, kre)
    except Exception as ex:   # the json schema is version dependent: the csv histograms above are the primary result
        print("json pass failed:", repr(ex))
