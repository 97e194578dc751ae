"""Timeline of k_raster_v3's wavefronts (DT_WAVE_SPANS build variant): how many run at each instant, what the launch's ramp and tail cost.
   DTSIM_LIB=.../libdtsim_spans.so DTSIM_WAVE_SPANS=/tmp/spans.bin python tools/raster_spans.py [c3|c5] [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))
import numpy as np
from dtsim import BatchedSimulator
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
kw = dict(c3=dict(maps="small_loop", extra={}), c5=dict(maps=["loop_only_duckies", "small_loop_only_duckies"], extra=dict(map_cycle=True)))[cfg]
sim = BatchedSimulator(kw["maps"], N, seed=1000, distortion=True, domain_rand=False, camera_width=640, camera_height=480, **kw["extra"])
rng = np.random.default_rng(1234)
for _ in range(6):
    sim.step(rng.uniform(-1, 1, (N, 2)).astype(np.float32))
for _ in range(3):
    sim.render()
raw = np.fromfile(os.environ["DTSIM_WAVE_SPANS"], dtype=np.uint64)
sp = raw[2 * 2048 * 4 * 8:].reshape(-1, 4, 4).astype(np.int64)          # [workgroup][wavefront]{start, tables, loop end, end}
ok = sp[:, :, 3] > 0
print(f"{cfg} N={N}: {int(ok.all(axis=1).sum())} workgroups stamped of {len(sp)}; wavefronts stamped {int(ok.sum())}; nonzero words in the file {int((raw != 0).sum())} of {raw.size}; resolve part nonzero {int((raw[:131072] != 0).sum())}")
sp = sp[ok.all(axis=1)]
t0 = sp[:, :, 0].min()
xcc = (sp[:, 0, 1] >> 32) & 15; bx = (sp[:, 0, 1] >> 40) & 7             # hardware XCC_ID of the workgroup, blockIdx & 7
sp[:, :, 1] = sp[:, :, 0] + (sp[:, :, 1] & 0xFFFFFFFF)
st, tb, le, en = [(sp[:, :, i] - t0) / 100.0 for i in range(4)]          # us
print("XCC_ID == blockIdx & 7 for %.2f %% of the workgroups; XCC_ID histogram %s" % (100 * (xcc == bx).mean(), np.bincount(xcc, minlength=8).tolist()))
span = en.max()
print("kernel span %.0f us; wavefront life: mean %.1f us (p50 %.1f, p99 %.1f, max %.1f); tables %.1f us mean; after the env loop (exact path) %.1f us mean"
      % (span, (en - st).mean(), np.percentile(en - st, 50), np.percentile(en - st, 99), (en - st).max(), (tb - st).mean(), np.where(le > 0, en - le, 0).mean()))
wg_life = en.max(axis=1) - st.min(axis=1)
print("workgroup life (slot held): mean %.1f us, p50 %.1f, p90 %.1f, p99 %.1f, max %.1f;  wavefront busy / slot held: %.1f %%"
      % (wg_life.mean(), *np.percentile(wg_life, [50, 90, 99, 100]), 100 * (en - st).sum() / (4 * wg_life.sum())))
# resident workgroups over time
ev = np.concatenate([np.stack([st.min(axis=1), np.ones(len(sp))], 1), np.stack([en.max(axis=1), -np.ones(len(sp))], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
res = np.cumsum(ev[:, 1]); tt = ev[:, 0]
peak = res.max()
dt = np.diff(tt, append=tt[-1])
print("resident workgroups: peak %d, time-average %.0f (%.1f %% of peak)" % (peak, (res * dt).sum() / span, 100 * (res * dt).sum() / span / peak))
for frac in (0.9, 0.5):
    below = tt[(res < frac * peak) & (tt > span / 2)]
    if len(below):
        print("   residency falls below %2.0f %% of peak at %.0f us (%.1f %% of the span before the end)" % (100 * frac, below[0], 100 * (span - below[0]) / span))
first_full = tt[res >= 0.95 * peak][0]
print("   residency reaches 95 %% of peak after %.0f us" % first_full)
for x in range(8):
    m = xcc == x
    if m.any():
        print("   XCD %d: %5d workgroups, last one ends at %6.0f us, slot-time held %8.0f us, mean workgroup life %.1f us" % (x, int(m.sum()), en[m].max(), wg_life[m].sum(), wg_life[m].mean()))
# dispatch order: start time against workgroup index (rwg = chunk * tiles + tile) -- which chunks run last
n_tiles = 300
ch = np.arange(len(sp)) // n_tiles
last = np.argsort(en.max(axis=1))[-8:]
print("   last workgroups to end: chunks", ch[last].tolist(), "lives", np.round(wg_life[last], 0).tolist())
