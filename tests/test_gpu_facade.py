"""Runs the C++ mirror of the reference's tests (tests/cpp/test_facade.cpp) on the GPU."""
import importlib.util
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.gpu
def test_reference_style_cpp_tests_on_gpu(tmp_path, rtc, O):
    spec = importlib.util.spec_from_file_location("_rtc_build", ROOT / "raytracer-challenge_amd" / "build.py")
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    exe = b.build_facade_tests()
    ppm = tmp_path / "criterion.ppm"
    script = ROOT / "raytracer-challenge_amd" / "data" / "orbit_animation.lua"   # render_lua (lua.rs:50-91) through the facade
    r = subprocess.run([str(exe), str(ppm), str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout + r.stderr
    # the PPM written through Canvas::write_to_file_simple equals the oracle's encoding of its own render
    scenes = importlib.import_module(rtc.__name__ + ".scenes")
    w, cam = scenes.criterion(400, 300)
    want = O.format_ppm(O.render(w.array(), len(w), w.light, cam, mode=1, nthreads=8))
    assert ppm.read_bytes() == want
    # ... and so does the file written from the quantised Canvas (Camera::render_async_rgb8 -> rtc_render_rgb8)
    assert (tmp_path / "criterion.ppm.rgb8").read_bytes() == want


def test_facade_compiles_against_the_library():
    """not-gpu: the C++ mirror of the reference API compiles and links against librtc.so."""
    spec = importlib.util.spec_from_file_location("_rtc_build", ROOT / "raytracer-challenge_amd" / "build.py")
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    exe = b.build_facade_tests(force=True)
    assert exe is not None and Path(exe).exists()
