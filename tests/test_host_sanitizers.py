"""Host-side C++ (scene assembly, forked BVH build, photon emission walk, threaded kd build, BMP writer)
under AddressSanitizer+UBSan and ThreadSanitizer.  No GPU code is involved: the driver links only
host_scene.cpp and host_photon.cpp."""
import os
import shutil
import subprocess

import pytest

from conftest import DATA, ROOT

CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")
DRIVER = os.path.join(ROOT, "tests", "host", "host_sanitize.cpp")


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_host_code_is_clean_under_sanitizers(san, tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_san")
    cc = subprocess.run(
        ["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=" + san, "-fno-omit-frame-pointer",
         "-o", exe, DRIVER, os.path.join(CSRC, "host_scene.cpp"), os.path.join(CSRC, "host_photon.cpp"), "-lpthread"],
        capture_output=True, text=True)
    if cc.returncode != 0 and "sanitizer" in cc.stderr.lower() and "cannot find" in cc.stderr.lower():
        pytest.skip("sanitizer runtime not installed")
    assert cc.returncode == 0, cc.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    # 80 teapot instances (79 k triangles) is past the size at which the BVH build forks its top levels
    run = subprocess.run([exe, os.path.join(DATA, "teapot_normal.obj"), os.path.join(DATA, "tex.bmp"),
                          str(tmp_path / "out.bmp"), "80"], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout + run.stderr)[-4000:]
    assert "ERROR" not in run.stderr and "WARNING: ThreadSanitizer" not in run.stderr, run.stderr[-4000:]
    assert os.path.getsize(str(tmp_path / "out.bmp")) > 54
