"""Test-only map fixtures shared by the golden-vector generator and the tests (TEST INFRASTRUCTURE, like the rest of
oracle/: nothing under gym-duckietown_amd/ imports this)."""


def junction_map():
    """Every drivable tile kind in every orientation (simulator.py:1151-1335 _get_curve: straight, curve_left/right,
    3way_left/right, 4way -- 2 / 6 / 12 curves per tile), plus grass / asphalt / floor and an empty cell, and a few
    objects."""
    o = ["S", "E", "N", "W"]
    rows = []
    for kind in ("straight", "curve_left", "curve_right", "3way_left", "3way_right"):
        rows.append([f"{kind}/{d}" for d in o] + ["grass", "asphalt"])
    rows.append(["4way", "4way", "floor", "empty", "grass", "4way"])
    return {"tiles": rows, "tile_size": 0.585,
            "objects": [{"kind": "duckie", "pos": [1.3, 0.4], "rotate": 30, "height": 0.08},
                        {"kind": "cone", "pos": [4.6, 3.5], "rotate": -75, "height": 0.1, "static": True},
                        {"kind": "duckie", "pos": [2.5, 4.5], "rotate": 120, "height": 0.08, "optional": True},
                        {"kind": "barrier", "pos": [3.4, 2.6], "rotate": 45, "scale": 0.17}]}     # `scale` form (simulator.py:978-985)
