"""Wall-clock spans of the persistent wavefronts of k_resolve / k_resolve_obj (DT_WAVE_SPANS build variant, tools/build_variant.sh spans
"-DDT_WAVE_SPANS=1"): DTSIM_LIB=.../libdtsim_spans.so DTSIM_WAVE_SPANS=/tmp/spans.bin python tools/wave_spans.py [c4|c5] [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-duckietown_amd"))
import numpy as np
from dtsim import BatchedSimulator
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
kw = dict(c5=dict(maps=["loop_only_duckies", "small_loop_only_duckies"], dr=False, extra=dict(map_cycle=True)),
          c4=dict(maps="loop_pedestrians", dr=True, extra={}))[cfg]
sim = BatchedSimulator(kw["maps"], N, seed=1000, distortion=True, domain_rand=kw["dr"], camera_width=640, camera_height=480, **kw["extra"])
rng = np.random.default_rng(1234)
for _ in range(6):
    sim.step(rng.uniform(-1, 1, (N, 2)).astype(np.float32))
for _ in range(3):
    sim.render()
raw = np.fromfile(os.environ["DTSIM_WAVE_SPANS"], dtype=np.uint64)
cn = raw[2 * 2048 * 4 * 8 - 8:2 * 2048 * 4 * 8].astype(np.int64)
if cn[0]:
    print("k_resolve pixels: %d; all four samples on one primitive: %.1f %% one tile, %.1f %% ground quad, %.1f %% sky" % (cn[0], 100.0 * cn[1] / cn[0], 100.0 * cn[2] / cn[0], 100.0 * cn[3] / cn[0]))
sp = raw[:2 * 2048 * 4 * 8].reshape(2, 2048 * 4, 8)
for k, name in enumerate(("k_resolve", "k_resolve_obj")):
    s = sp[k][sp[k][:, 1] > 0].astype(np.int64)
    if not len(s):
        print(name, "did not run"); continue
    t0 = s[:, 0].min(); st = (s[:, 0] - t0) / 100.0; en = (s[:, 1] - t0) / 100.0   # microseconds
    work = s[:, 2] > 0
    print(f"{cfg} N={N} {name}: {len(s)} wavefronts ({int(work.sum())} got items); kernel span {en.max():.0f} us")
    print("   start of wavefronts (us): p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % tuple(np.percentile(st, [50, 90, 99, 100])))
    print("   end of wavefronts with items (us): p1 %.0f  p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % tuple(np.percentile(en[work], [1, 10, 50, 90, 99, 100])))
    busy = (en[work] - st[work]).sum() / (len(s) * en.max())
    print("   wavefront-time in use: %.1f %% of (wavefronts x span); items per wavefront: mean %.1f max %d; longest item: p50 %.0f us  p99 %.0f  max %.0f"
          % (100 * busy, s[work, 2].mean(), s[:, 2].max(), *np.percentile(s[work, 3] / 100.0, [50, 99, 100])))
    w = s[work]
    print("   first item: starts %.1f us after the wavefront (p50; p99 %.1f), lasts p50 %.0f us; all items: mean %.1f us; last item p50 %.0f us; between items (span - sum): p50 %.0f us"
          % (np.percentile(w[:, 5], 50) / 100.0, np.percentile(w[:, 5], 99) / 100.0, np.percentile(w[:, 4], 50) / 100.0, (w[:, 6].sum() / w[:, 2].sum()) / 100.0,
             np.percentile(w[:, 7], 50) / 100.0, np.percentile((w[:, 1] - w[:, 0] - w[:, 6]), 50) / 100.0))
    # how much of the span is the tail: time after which fewer than half / a tenth of the wavefronts still run
    for frac in (0.5, 0.1, 0.01):
        t = np.sort(en[work])[int(len(en[work]) * (1 - frac)) - 1]
        print("   %4.0f %% of the wavefronts still running after %.0f us (%.0f %% of the span)" % (100 * frac, t, 100 * t / en.max()))
