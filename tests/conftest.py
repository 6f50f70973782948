import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLD, "data")
CAM = np.array([2.8, 0, 0, 0, 0, 0.0])   # camera position + look-at of the reference's main.cpp:626-628


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLD, name))


def build_scene(api, mesh="teapot_normal.obj", rgbs=(0.5, 0.5, 0.5, 0.0), ior=1.0, off=(0, 0, 0), cornell=True):
    """Same scene through any of the three front ends (oracle, compiled reference, product)."""
    if mesh:
        assert api.load_obj(os.path.join(DATA, mesh), rgbs, ior, off) == 1
    if cornell:
        api.add_cornell()
    api.build_bvh()
    return api


def golden_hashes(name="idmap_hashes.txt"):
    out = {}
    with open(os.path.join(GOLD, name)) as f:
        for line in f:
            tag, h, hits, mesh_hits, tsha = line.split()
            out[tag] = dict(hash=h, hits=int(hits), mesh_hits=int(mesh_hits), t_sha256=tsha)
    return out


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import bindings
    return bindings


@pytest.fixture(scope="session")
def gpu_ctx():
    import raytracer_amd as rt
    return rt.Context(0)
