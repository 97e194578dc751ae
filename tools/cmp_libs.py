"""Frames of one batch through two builds of the library (A/B aid):  python tools/cmp_libs.py VARIANT [c3|c4|c5]   (lib/libdtsim.so against lib/libdtsim_VARIANT.so, one process each)."""
import os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 3:                                    # child: render and dump
    sys.path.insert(0, os.path.join(root, "gym-duckietown_amd"))
    from dtsim.batched import BatchedSimulator
    cfg = sys.argv[2]
    kw = dict(c3=dict(maps="small_loop", extra={}), c5=dict(maps=["loop_only_duckies", "small_loop_only_duckies"], extra=dict(map_cycle=True)),
              c4=dict(maps="loop_pedestrians", extra={}, dr=True))[cfg]
    sim = BatchedSimulator(kw["maps"], 256, camera_width=640, camera_height=480, distortion=True, domain_rand=kw.get("dr", False), seed=5, max_steps=100000, **kw["extra"])
    acts = np.random.default_rng(9).uniform(0.2, 0.9, (4, 256, 2)).astype(np.float32)
    sim.step(acts, n_steps=4); sim.render(); np.save(sys.argv[3], sim.frames_host()); sim.close()
    sys.exit(0)
variant, cfg = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "c3")
outs = []
for lib, name in ((None, "/tmp/cmp_a.npy"), (os.path.join(root, "gym-duckietown_amd", "lib", f"libdtsim_{variant}.so"), "/tmp/cmp_b.npy")):
    env = dict(os.environ)
    if lib: env["DTSIM_LIB"] = lib
    subprocess.run([sys.executable, __file__, variant, cfg, name], check=True, env=env, stderr=subprocess.DEVNULL)
    outs.append(np.load(name))
d = (outs[0] != outs[1]).any(axis=-1)
print(cfg, "default vs", variant, ": pixels that differ", int(d.sum()), "of", d.size)
