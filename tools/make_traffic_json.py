"""Turn the rocprofv3 PMC passes of tools/prof_bench.sh into profiles/raster_pmc_latest.json
(consumed by bench.py for roofline.traffic).  FETCH_SIZE is doubled (gfx950 reports half the bytes of a
wide coalesced read, MI355X_MICROARCH.md); WRITE_SIZE is taken as is; both are KB -> bytes."""
import glob, json, os, sqlite3, sys, collections

def counters(db):
    con = sqlite3.connect(db); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    kd, ks, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    disp = {r[0]: (r[1], r[2]) for r in cur.execute(f"select d.event_id, s.kernel_name, d.id from {kd} d join {ks} s on d.kernel_id=s.id")}
    acc = collections.defaultdict(lambda: collections.defaultdict(dict))
    for ev, cname, val in cur.execute(f"select e.event_id, i.name, e.value from {pe} e join {pi} i on e.pmc_id=i.id"):
        if ev in disp:
            k, did = disp[ev]
            acc[k][cname][did] = acc[k][cname].get(did, 0.0) + val      # sum over instances of one dispatch
    return acc

out_dir, envs = sys.argv[1], int(sys.argv[2])
source = sys.argv[3] if len(sys.argv) > 3 else out_dir          # label carried into bench.py's roofline.traffic_source
valu_per_pixel = float(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != "-" else None
clock_ghz = float(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] != "-" else None
config = sys.argv[6] if len(sys.argv) > 6 else "c3"           # c3 -> raster_pmc_latest.json, c4 / c5 -> raster_pmc_<config>.json
out_name = "raster_pmc_latest.json" if config == "c3" else f"raster_pmc_{config}.json"
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
detail = {}
for db in glob.glob(os.path.join(out_dir, "*", "*.db")):
    for k, cs in counters(db).items():
        names = ("k_raster", "k_resolve", "k_cam_setup", "k_pix_setup", "k_env_sort", "k_obj_setup")
        if not any(n in k for n in names):
            continue
        short = next(n for n in names if n in k)
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            if c in cs:
                v = sum(cs[c].values()) / len(cs[c])       # mean per launch
                detail.setdefault(short, {})[c + "_KB_per_launch"] = v
                tot[c] += v
hbm = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
json.dump({"envs": envs, "config": config, "hbm_bytes_per_launch": hbm, "fetch_KB_raw": tot["FETCH_SIZE"], "write_KB": tot["WRITE_SIZE"],
           "per_kernel": detail, "source": source, "valu_per_pixel": valu_per_pixel, "clock_ghz": clock_ghz,
           "note": "render pass = every kernel dtsim_render launches; FETCH_SIZE doubled per MI355X_MICROARCH.md"},
          open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", out_name), "w"), indent=1)
print("hbm bytes per render pass:", hbm, detail)
