"""`not gpu`: the two readings of `get_transform(desc, W, tile_size)` (simulator.py:936-938) on the 8 x 7 loop maps.

`get_transform` lives in duckietown_world (absent here): its second argument is the grid HEIGHT in tile
units (cartesian y = (H - pos_z) * tile_size, README.md:239 "[0.5, 1.5] ... middle of the first column, middle of the
second row"), but the simulator passes `self.grid_width`.  On a square map both readings coincide; on the 8 x 7 loop
maps -- the maps of BASELINE configs C4 / C5 -- the grid_width reading moves every object by (H - W) = -1 tile in z.

Evidence available inside the reference tree for which placement is the INTENDED one: loop_only_duckies.yaml opens with
"a closed loop ... with a few obstacles directly on the road, which have to be avoided" -- under the README reading all
eight duckies stand on drivable tiles, under the grid_width reading half of them stand on grass.  The product default is
therefore the README reading (`transform_uses_width=False`); the flag reproduces the other one, and bench.py prints the
assumption for C4 / C5 (`config.get_transform`).  The arithmetic itself stays parity-unpinned.
"""
import numpy as np
import pytest

from dtsim import assets, maps


def _object_tiles(name, uses_width):
    md = assets.get_map(name)
    mt = maps.interpret_map(md, name, transform_uses_width=uses_width)
    ts = md["tile_size"]
    H, W = len(md["tiles"]), len(md["tiles"][0])
    out = []
    for o in mt.objects:
        i, j = int(np.floor(o.pos[0] / ts)), int(np.floor(o.pos[2] / ts))
        kind = md["tiles"][j][i].split("/")[0] if 0 <= i < W and 0 <= j < H else None
        out.append(((i, j), kind, o.pos.copy()))
    return out, (H, W), ts


@pytest.mark.parametrize("name", ["loop_only_duckies", "loop_pedestrians"])
def test_both_readings_on_the_8x7_loop_maps(name):
    readme, (H, W), ts = _object_tiles(name, False)
    width, _, _ = _object_tiles(name, True)
    assert (H, W) == (7, 8) and len(readme) == len(width) == 8
    drivable = {"straight", "curve_left", "curve_right", "3way_left", "3way_right", "4way"}
    # README reading: pos = (px * ts, 0, pz * ts) -- every obstacle "directly on the road" (the yaml's own header)
    md = assets.get_map(name)
    descs = md["objects"] if isinstance(md["objects"], list) else list(md["objects"].values())
    for (tile, kind, pos), d in zip(readme, descs):
        assert np.allclose(pos, [d["pos"][0] * ts, 0.0, d["pos"][1] * ts])
        assert kind in drivable, (tile, kind)
    # grid_width reading: the same x, z shifted by (H - W) tiles = one tile up; half of the obstacles leave the road
    for (t0, _, p0), (t1, _, p1) in zip(readme, width):
        assert p1[0] == p0[0] and np.isclose(p1[2] - p0[2], (H - W) * ts)
        assert t1 == (t0[0], t0[1] - 1)
    assert sum(k not in drivable for _, k, _ in width) >= 4


def test_readings_coincide_on_square_maps():
    a, (H, W), _ = _object_tiles("small_loop_only_duckies", False)
    b, _, _ = _object_tiles("small_loop_only_duckies", True)
    assert H == W
    for (ta, _, pa), (tb, _, pb) in zip(a, b):
        assert ta == tb and np.array_equal(pa, pb)
