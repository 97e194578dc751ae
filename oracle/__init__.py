"""CPU oracle for the dtsim hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import anything from this package, and only as the checker.  The product
(gym-duckietown_amd/) never imports it and fails loudly if libdtsim.so (the
HIP library) is missing.

Contents
  sim.py        numpy/float64 restatement of the reference's per-step path
                (kinematics, dynamics, lane pose, SAT collision, proximity,
                reward/done, dynamic duckies, reset RNG order).  Every
                function cites the reference file:line it follows.
  raster.py     software restatement of Simulator._render_img
                (simulator.py:1707-1951); pinned to the reference's own
                frames on real OpenGL (tests/golden/ref_gl_*.npz).
  gl/           gl_headless.c (headless Mesa llvmpipe context, no X),
                glshim.py (the slice of pyglet the reference uses, over it),
                refgl.py (runs the reference's Simulator UNMODIFIED on it --
                build container only), glport.py (the reference's GL call
                stream restated so that bench.py's cpu_baseline can run it
                where /root/reference does not exist), measure_filter.py.
  make_gl_golden.py  produced tests/golden/ref_gl_*.npz with refgl.py.
  distortion.py restatement of distortion.py + the three OpenCV calls.
  refstub.py    loads the reference's *own* Python under stubbed third-party
                modules (only where /root/reference exists) -- used to pin
                sim.py and to generate tests/golden/.
  make_golden.py script that produced tests/golden/*.npz / *.json.

Pinning status:
  PINNED against the reference's own code run in the build container
  (tests/test_oracle_vs_reference.py + tests/golden/): map interpretation,
  curves, get_lane_pos2, _valid_pose, _collision, proximity_penalty2,
  compute_reward/_compute_done_reward, DuckieObj.step, Randomizer order,
  distortion._invert_map/_fill_holes; and, round 6, the RENDER PATH: frames of
  the unmodified reference on Mesa llvmpipe (its CI renderer) against
  raster.py (tests/test_gl_golden.py), the HIP raster (tests/test_gpu_gl_golden.py)
  and gl/glport.py (byte-identical).
  PARITY UNPINNED (third-party arithmetic absent from /root/reference, no
  reference test pins it): duckietown_world dynamics (DB18 model + delay),
  get_transform, OpenCV getOptimalNewCameraMatrix/initUndistortRectifyMap/
  remap, gym's RNG flavour.
"""
