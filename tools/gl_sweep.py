"""A one-off robustness sweep of the GL parity (build container + one gpurun call; nothing of it is committed but its result):

    python -W ignore tools/gl_sweep.py        # 360 more frames of the unmodified reference on llvmpipe -> tests/golden/ref_gl_xtra_*.npz (4 MB, NOT committed)
    gpurun -- 'python -m pytest tests/test_gpu_gl_golden.py -q -s -m gpu -k "xtra and frames_match"'
    rm tests/golden/ref_gl_xtra_*.npz

Result of round 6: profiles/r06_gl_sweep.txt."""
import sys, os, json
sys.path[:0] = ["/root/repo", "/root/repo/gym-duckietown_amd"]
import numpy as np
from oracle import make_gl_golden as MG
from oracle.gl import refgl
extra = {
 "xtra_small_dr": (MG.case_reset_poses, dict(map_name="small_loop_only_duckies", tree="t256", dr=True, W=160, H=120, seeds=list(range(1000, 1080)))),
 "xtra_loop_dr": (MG.case_reset_poses, dict(map_name="loop_only_duckies", tree="t256", dr=True, W=160, H=120, seeds=list(range(2000, 2080)))),
 "xtra_small": (MG.case_reset_poses, dict(map_name="small_loop_only_duckies", tree="t256", dr=False, W=160, H=120, seeds=list(range(3000, 3080)))),
 "xtra_town_dr": (MG.case_reset_poses, dict(map_name="test_town", tree="t128", dr=True, W=160, H=120, seeds=list(range(4000, 4060)))),
 "xtra_town": (MG.case_reset_poses, dict(map_name="test_town", tree="t128", dr=False, W=160, H=120, seeds=list(range(5000, 5060)))),
}
MG.CASES.update(extra)
for name in extra:
    data = MG.build(name)
    np.savez_compressed(os.path.join(MG.OUT, f"ref_gl_{name}.npz"), **data)
    print(name, data["frame"].shape)
