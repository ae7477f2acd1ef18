"""Register / scratch / LDS use of every k_trace instantiation (hipcc -Rpass-analysis=kernel-resource-usage); CPU only.
usage: python tools/kernel_resources.py [extra -D flags]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       f"-I{ROOT / 'include'}", f"-I{ROOT / 'raytracer-challenge_amd' / 'csrc'}", "-mllvm", "-disable-machine-licm",
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[1:], "-c", str(ROOT / "raytracer-challenge_amd" / "csrc" / "rtc_kernels.hip"),
       "-o", "/tmp/rtc_k.o"]
t = subprocess.run(cmd, capture_output=True, text=True).stderr
rows = []
for blk in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = blk.split("\n")[0].strip()
    if "k_trace" not in name and "undeal" not in name:
        continue
    def g(k):
        m = re.search(k + r": (\d+)", blk)
        return m.group(1) if m else "?"
    m = re.search(r"k_traceILi(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
    tag = "k_trace<%s,refl=%s,refr=%s,probe=%s>" % m.groups() if m else name[:44]
    scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    rows.append(f"{tag:44s} VGPR {g('VGPRs'):>4s} SGPR {g('SGPRs'):>4s} scratch {scratch:>5s} occupancy {occ:>2s} LDS {lds:>6s}")
print("\n".join(sorted(rows)))
