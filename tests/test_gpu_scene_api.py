"""The C++ scene API (raytracer_amd/scene/*.h, same class names as the reference) end to end:
a main.cpp-style program builds the scene with ObjReader / makeCornellBox / BVH / KDTree, renders
through Renderer on the GPU and writes the BMP with saveBmp; the image must match the CPU oracle's
render of the same scene and photon set."""
import os
import subprocess

import numpy as np
import pytest

from conftest import CAM, DATA, ROOT, build_scene, golden
from oracle.bindings import Oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("device_tree", [False, True])
def test_main_style_scene_program(tmp_path, device_tree):
    exe = os.path.join(ROOT, "examples", "main_scene")
    assert os.path.exists(exe), "examples/main_scene missing: run __graft_entry__.build()"
    out = str(tmp_path / "scene.bmp")
    w, h, nph, seed = 96, 72, 3000, 12345
    r = subprocess.run([exe, os.path.join(DATA, "teapot_normal.obj"), out, str(w), str(h), str(nph), str(seed), "1.8", "1.5", "-0.7"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120,
                       env=dict(os.environ, RT_DEVICE_TREE="1" if device_tree else "0"))
    assert r.returncode == 0, r.stdout
    assert "centre ray hits an object" in r.stdout
    raw = np.fromfile(out, np.uint8)
    assert raw[:2].tobytes() == b"BM" and len(raw) == 54 + w * h * 3
    got = raw[54:].reshape(h, w, 3).astype(np.int32)

    orc = build_scene(Oracle(), "teapot_normal.obj", (1.0, 1.0, 1.0, 1.8), 1.5, (0, -0.7, 0))
    orc.seed(seed)
    orc.emit_photons(nph)
    orc.build_kdtree()
    orc.params.photon_num = nph
    orc.params.head_quirk = 0
    rgb, _ = orc.render(w, h, CAM, threads=8)
    ref = orc.quantize(rgb).reshape(h, w, 3).astype(np.int32)
    within = (np.abs(got - ref).max(axis=-1) <= 1).mean()
    # The device-built tree has widened boxes; the reference's exact boxes reject some hits of refracted
    # rays that start on a box face (a triangle's own vertices span its leaf's box), which an exhaustive
    # search finds.  A few glass pixels therefore differ and the bar for that tree is 99 %.
    assert within >= (0.99 if device_tree else 0.999), within


@pytest.mark.parametrize("devices", ["0,0,0", "0,0"])
def test_scene_program_split_over_several_contexts(tmp_path, devices):
    """RT_DEVICES: the scene classes replicate the scene and the photon map on every listed GPU and the Renderer
    drives one host thread per GPU (a one-GPU box lists device 0 several times): bands of 8 rows when the height
    divides into them (72 rows over 3), contiguous row blocks otherwise (72 rows over 2).  Same BMP, byte for byte."""
    exe = os.path.join(ROOT, "examples", "main_scene")
    assert os.path.exists(exe), "examples/main_scene missing: run __graft_entry__.build()"
    w, h, nph, seed = 96, 72, 3000, 12345
    outs = []
    for tag, env in (("one", {}), ("many", {"RT_DEVICES": devices})):
        out = str(tmp_path / f"scene_{tag}.bmp")
        r = subprocess.run([exe, os.path.join(DATA, "teapot_normal.obj"), out, str(w), str(h), str(nph), str(seed), "1.8", "1.5", "-0.7"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout
        outs.append(np.fromfile(out, np.uint8))
    assert len(outs[0]) == 54 + w * h * 3 and outs[0][54:].max() > 0
    assert np.array_equal(outs[0], outs[1])


def test_scene_program_subsurface_two_trees(tmp_path):
    """main.cpp:724-728 as written there -- `new KDTree(photons)` then `new KDTree(volumePhotons)`, both through the
    one-argument constructor: the second tree must become the volume map, not replace the surface map.  The frame is the
    one the compiled reference's getColorAt produced for the subsurface teapot (RMSE-level parity: the march's step
    lengths come from another generator)."""
    exe = os.path.join(ROOT, "examples", "main_scene")
    assert os.path.exists(exe), "examples/main_scene missing: run __graft_entry__.build()"
    g = golden("render_sss_64.npz")
    out = str(tmp_path / "sss.bmp")
    r = subprocess.run([exe, os.path.join(DATA, "teapot_normal.obj"), out, "64", "64", "20000", "12345", "-1.0", "1.3", "-0.7", "0.9", "0.7", "0.6", "50000"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=dict(os.environ, RT_DEVICE_TREE="0"))
    assert r.returncode == 0, r.stdout
    assert f"photons {len(g['photons'])} + {len(g['volume_photons'])}" in r.stdout, r.stdout
    got = np.fromfile(out, np.uint8)[54:].reshape(64, 64, 3)[..., ::-1].astype(np.int32)
    ref = np.floor(np.clip(g["rgb"], 0, 1) * 255).astype(np.int32)
    within = (np.abs(got - ref).max(-1) <= 1).mean()
    rmse = float(np.sqrt(((got - ref) ** 2).mean()))
    assert within >= 0.99 and rmse <= 1.0, (within, rmse)
