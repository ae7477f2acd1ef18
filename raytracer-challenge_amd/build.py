"""Build librtc.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU. -ffp-contract=off is part of the
contract: the reference (rustc) never fuses a*b+c, and pixel/hit parity needs the same
roundings (SURVEY.md §7 "hard parts").
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "librtc.so"

SOURCES = ["host_math.cpp", "host_ppm.cpp", "host_yaml.cpp", "host_lua.cpp", "rtc_api.cpp", "rtc_group.cpp", "rtc_kernels.hip"]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", f"-I{ROOT / 'include'}", f"-I{CSRC}"]
COMMON += os.environ.get("RTC_CXXFLAGS", "").split()  # experiments, e.g. -DRTC_WAVES_PER_SIMD=4
# Kernel file only: MachineLICM hoists the VGPR materialisation of every f64 literal (pow's ~25
# polynomial coefficients alone are 50 VGPRs) out of the per-sample loop to the kernel entry, where
# they stay live for the whole kernel: 128 VGPRs + 60 B/lane of scratch with it, 112 VGPRs and no
# scratch without (flat culled kernel); 0.129 ms -> 0.113 ms per north-star frame.
KERNEL_ONLY = ["-mllvm", "-disable-machine-licm"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every source to an object in build/ and link librtc.so next to this file."""
    cc = hipcc()
    objdir = ROOT / "build" / "rtc"
    objdir.mkdir(parents=True, exist_ok=True)
    headers = [ROOT / "include" / "rtc.h", CSRC / "rtc_device.h", CSRC / "rtc_internal.h", CSRC / "rtc_bands.h"]
    objs = []
    for name in SOURCES:
        src = CSRC / name
        obj = objdir / (name + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            extra = KERNEL_ONLY if name.endswith(".hip") else []
            cmd = [cc, "--offload-arch=gfx950", *COMMON, *extra, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
    if force or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs), "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB


def build_facade_tests(force: bool = False) -> Path:
    """Compile tests/cpp/test_facade.cpp (C++ mirror of the reference API) against librtc.so."""
    src = ROOT / "tests" / "cpp" / "test_facade.cpp"
    exe = ROOT / "build" / "test_facade"
    hdr = PKG / "host" / "ch1.hpp"
    if not src.exists() or not hdr.exists():
        return None
    exe.parent.mkdir(parents=True, exist_ok=True)
    if force or _stale(exe, [src, hdr, LIB, ROOT / "include" / "rtc.h"]):
        cmd = ["g++", "-O1", "-std=c++17", "-ffp-contract=off", f"-I{ROOT / 'include'}", f"-I{PKG / 'host'}",
               str(src), "-o", str(exe), f"-L{PKG}", "-lrtc", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,$ORIGIN/../raytracer-challenge_amd"]
        subprocess.run(cmd, check=True)
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
