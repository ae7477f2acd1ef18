"""Regenerate the golden canvases under tests/golden/ with the CPU oracle (oracle/).

The reference is Rust and cannot be built in this environment (no cargo/rustc), so
these fixtures are outputs of the oracle — which is itself pinned by the reference's own
known-answer tests (tests/test_oracle_kats.py). Run: python tests/golden/make_golden.py
"""
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "oracle")]
import oracle as O  # noqa: E402
from _bootstrap import package  # noqa: E402

rtc = package()
scenes = importlib.import_module(rtc.__name__ + ".scenes")
HERE = Path(__file__).resolve().parent


def jamis(w=100, h=50):
    world, cam = rtc.load_yaml(path=str(Path(rtc.__file__).parent / "data" / "reflect_refract.yml"))
    view = rtc.Matrix(np.array(list(cam.view_inv)).reshape(4, 4)).inverse()
    return world, rtc.camera(w, h, cam.fov, view)


def main():
    world, cam = jamis()
    img = O.render(world.array(), len(world), world.light, cam, mode=1, nthreads=8)
    np.save(HERE / "jamis_100x50.npy", img)
    (HERE / "jamis_100x50.ppm").write_bytes(O.format_ppm(img))
    world, cam = scenes.test7(80, 60)
    np.save(HERE / "test7_80x60.npy", O.render(world.array(), len(world), world.light, cam, mode=0, nthreads=8))
    world, cam = scenes.synthetic(100, 96, 54)
    np.save(HERE / "synthetic100_96x54.npy", O.render(world.array(), len(world), world.light, cam, mode=1, nthreads=8))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
