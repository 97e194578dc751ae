"""Asset directory trees shared by the GL golden generator (reference side) and the tests (oracle / product side).

TEST INFRASTRUCTURE ONLY.  Two duckietown-world style trees, built on demand into a scratch directory from what the repo
commits (deterministic: procedural data, lossless PNG), so that the reference (through get_resource_path / get_texture_file)
and the product (through dtsim.assets.AssetLibrary(root)) read the SAME files:

  t128   tests/golden/assets as committed (128 x 128 tile images, OBJ / MTL meshes incl. the stand-in duckie) + MapFormat1
         YAML files of the fixture maps (dtsim.assets.MAPS)
  t256   the same meshes and maps with the 256 x 256 procedural tile textures of dtsim.assets.make_texture written as PNG
         (the texels the product's fixtures hold in memory; the size k_raster_v3 is specialised for)
"""
from __future__ import annotations

import os
import shutil

import numpy as np
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ASSETS = os.path.join(ROOT, "tests", "golden", "assets")
KINDS = ("grass", "asphalt", "floor", "straight", "curve_left", "curve_right", "3way_left", "4way")
_built = {}


def tree(name: str) -> str:
    """Path of the tree `name` (built once per process)."""
    if name in _built:
        return _built[name]
    import yaml
    from PIL import Image
    from dtsim import assets
    base = os.path.join(tempfile.gettempdir(), f"dtsim_gl_assets_{os.getuid()}_{os.getpid()}", name)
    if os.path.isdir(base):
        shutil.rmtree(base)
    os.makedirs(base)
    if not _built:
        import atexit
        atexit.register(shutil.rmtree, os.path.dirname(base), ignore_errors=True)
    shutil.copytree(os.path.join(ASSETS, "meshes"), os.path.join(base, "meshes"))
    shutil.copytree(os.path.join(ASSETS, "maps"), os.path.join(base, "maps"))
    for m, md in assets.MAPS.items():
        with open(os.path.join(base, "maps", f"{m}.yaml"), "w") as f:
            yaml.safe_dump(md, f)
    os.makedirs(os.path.join(base, "textures"), exist_ok=True)
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(os.path.join(base, "textures", "black_tile.png"))   # objmesh.py:285: what untextured chunks are given in the segmentation view
    if name == "t128":
        shutil.copytree(os.path.join(ASSETS, "textures"), os.path.join(base, "textures"), dirs_exist_ok=True)
    elif name == "t256":
        for kind in KINDS:
            d = os.path.join(base, "textures", "tiles-processed", "photos", kind)
            os.makedirs(d)
            Image.fromarray(assets.make_texture(kind, 256)[..., :3]).save(os.path.join(d, "texture.png"))
    else:
        raise KeyError(name)
    # Texture.bind(segment=True) (graphics.py:52-57) asks duckietown_world for a texture named after the tile KIND ("grass", "straight", ...):
    # the package's legacy per-kind images.  Stand-ins: the same image under that name (only the reference's resource lookup sees them).
    os.makedirs(os.path.join(base, "textures", "legacy"))
    for kind in KINDS:
        src = os.path.join(base, "textures", "tiles-processed", "photos", kind, "texture.png")
        if os.path.isfile(src):
            shutil.copyfile(src, os.path.join(base, "textures", "legacy", f"{kind}.png"))
    _built[name] = base
    return base


def roots(name: str):
    return [tree(name)]
