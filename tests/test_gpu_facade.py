"""Runs the C++ mirror of the reference's tests (tests/cpp/test_facade.cpp) on the GPU."""
import importlib.util
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def decode_png(png: bytes) -> np.ndarray:
    """A PNG reader for what rtc_canvas_write_png8 writes (8-bit RGB / RGBA, filter 0): chunk CRCs checked, zlib inflates."""
    import struct
    import zlib
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(png):
        n, = struct.unpack(">I", png[pos:pos + 4])
        typ, data = png[pos + 4:pos + 8], png[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", png[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + data) & 0xffffffff == crc
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", data)
        if typ == b"IDAT":
            idat += data
        pos += 12 + n
        if typ == b"IEND":
            break
    assert pos == len(png) and ihdr[2] == 8 and ihdr[4:] == (0, 0, 0)
    w, h, c = ihdr[0], ihdr[1], {2: 3, 6: 4}[ihdr[3]]
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * c)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, c)


@pytest.mark.gpu
def test_reference_style_cpp_tests_on_gpu(tmp_path, rtc, O):
    spec = importlib.util.spec_from_file_location("_rtc_build", ROOT / "raytracer-challenge_amd" / "build.py")
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    exe = b.build_facade_tests()
    ppm = tmp_path / "criterion.ppm"
    import shutil
    for name in ("orbit_animation.lua", "orbit_lib.lua"):   # render_lua (lua.rs:50-91) through the facade, on a copy (it writes beside the script)
        shutil.copy(ROOT / "raytracer-challenge_amd" / "data" / name, tmp_path / name)
    script = tmp_path / "orbit_animation.lua"
    r = subprocess.run([str(exe), str(ppm), str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout + r.stderr
    png = (tmp_path / "orbit_animation.lua.top.png").read_bytes()
    top = rtc.LuaProgram(path=script).job(12)
    want8 = rtc.color_scale255(O.render(top.world.array(), len(top.world), top.world.light, top.camera, mode=1, nthreads=8)).reshape(200, 320, 3)
    diff = np.abs(decode_png(png).astype(np.int16) - want8.astype(np.int16))
    assert diff.max() <= 1 and np.count_nonzero(diff) <= 4
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
