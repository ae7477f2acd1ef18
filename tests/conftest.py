import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle binding (oracle/oracle.py) — test infrastructure."""
    import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def rtc():
    """The product package (ctypes binding of librtc.so)."""
    from _bootstrap import package
    pkg = package()
    if not pkg.LIB_PATH.exists():   # a checkout that was never built: compile librtc.so (hipcc cross-compiles gfx950)
        import importlib.util
        spec = importlib.util.spec_from_file_location("_rtc_build", ROOT / "raytracer-challenge_amd" / "build.py")
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        b.build()
    pkg.lib()
    return pkg


@pytest.fixture(scope="session")
def gpu(rtc):
    """A device context; fails loudly if no gfx950 is usable (no CPU fallback exists)."""
    ctx = rtc.Context(0)
    yield ctx
    ctx.close()
