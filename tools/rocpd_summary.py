"""Summarise rocprofv3 rocpd sqlite output: per-kernel durations and PMC counter means."""
import sqlite3, sys, glob, collections

def summarise(path):
    con = sqlite3.connect(path); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    def T(prefix):
        return [t for t in tabs if t.startswith(prefix)][0]
    kd, ks = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    rows = cur.execute(f"select d.id, s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
    by = collections.defaultdict(list)
    for _id, name, st, en in rows:
        by[name.split('(')[0][-60:]].append((en - st) / 1e3)
    print(f"== {path}")
    for name, ds in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {name:60s} calls {len(ds):4d}  avg {sum(ds)/len(ds):10.2f} us  min {min(ds):10.2f}  total {sum(ds)/1e3:9.3f} ms")
    pe, pi = T("rocpd_pmc_event"), T("rocpd_info_pmc")
    ev = cur.execute(f"select e.event_id, i.name, e.value from {pe} e join {pi} i on e.pmc_id = i.id").fetchall()
    if ev:
        id2name = {r[0]: r[1].split('(')[0][-40:] for r in rows}
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        m = dict(cur.execute(f"select event_id, id from {kd}").fetchall()) if "event_id" in cols else {}
        for event_id, cname, val in ev:
            acc[id2name.get(m.get(event_id), "?")][cname].append(val)
        for kname, d in acc.items():
            print(f"  [pmc] {kname}")
            for cname, vals in sorted(d.items()):
                print(f"      {cname:28s} mean {sum(vals)/len(vals):16.1f}  (n={len(vals)})")

for p in sys.argv[1:]:
    for f in sorted(glob.glob(p)):
        summarise(f)
