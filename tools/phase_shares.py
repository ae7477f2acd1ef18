"""Diagnostic: where does a wave of k_trace spend its cycles? Needs a -DRTC_STAMPS build
(RTC_CXXFLAGS=-DRTC_STAMPS python raytracer-challenge_amd/build.py --force). Prints the share of
each phase (s_memtime deltas summed over all waves). The stamped build serialises memory at every
stamp, so only the SHARES are meaningful, never its run time."""
import ctypes as C
import importlib
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT)]
from _bootstrap import package  # noqa: E402

rtc = package()
scenes = importlib.import_module(rtc.__name__ + ".scenes")
import torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
refl = len(sys.argv) > 2 and sys.argv[2] == "reflective"
w, cam = scenes.synthetic(n, 1920, 1080, with_plane=(n <= 1000), reflective=refl)
ctx = rtc.Context(0)
dw = ctx.upload(w)
y0, y1 = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 1080)  # one rank's row tile
buf = torch.zeros((y1 - y0, 1920, 3), dtype=torch.float64, device="cuda:0")
torch.cuda.synchronize()
for _ in range(3):
    dw.render_rows(cam, y0, y1, buf.data_ptr())
ctx.reset_stats()
dw.render_rows(cam, y0, y1, buf.data_ptr())
ctx.synchronize()
NW = 30 * ((y1 - y0 + 7) // 8) * 8  # waves launched: 8x8-pixel tiles, 32x8 blocks
out = (C.c_ulonglong * 40)()
rtc.lib().rtc_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint32]
rtc.lib().rtc_debug_counters(ctx._h, out, 40)
# STAMP(i): time since the previous stamp, summed over the wave's passes ([8 + i]); secondary passes only ([32 + i])
names = ["(kernel entry -> first stamp)", "pass top: ray generation / previous pass's lighting + frame push", "bundle", "closest-hit walk",
         "hit record + n1/n2 + shadow ray", "shadow set-up (+ listed shadow walk)", "shadow walk (bundle / lists of two-level worlds)", "final lighting + tile store"]
tot = sum(out[8 + i] for i in range(8))
print(f"objects {len(w)}  kernel_ms(stamped) {ctx.last_kernel_ms():.3f}  waves {NW}")
for i, nm in enumerate(names):
    print(f"  {nm:72s} {out[8 + i] / max(tot, 1) * 100:5.1f} %   {out[8 + i] / NW:9.0f} ticks/wave   of which secondary passes {out[32 + i] / NW:9.0f}")
print(f"  closest-hit walks of secondary passes whose bundle could not be bounded: {out[32] / NW:9.0f} ticks/wave (part of the closest-hit walk row)")
d = [out[16 + i] for i in range(16)]
print(f"per wave: closest passes {d[0] / NW:.2f} (unbounded bundle {d[1] / NW:.2f}), exact tests/closest pass {d[2] / max(d[0], 1):.2f}; "
      f"shadow passes {d[3] / NW:.2f} (unbounded {d[4] / NW:.2f}), exact tests/shadow pass {d[5] / max(d[3], 1):.2f}; "
      f"per-lane prefilter evaluations per wave: closest {d[6] / NW:.1f}, shadow {d[7] / NW:.1f}")
print(f"secondary (reflection / refraction) passes per wave {d[12] / NW:.2f}: exact tests per such closest pass {d[13] / max(d[12], 1):.2f}; "
      f"their shadow passes per wave {d[14] / NW:.2f}, exact tests per such shadow pass {d[15] / max(d[14], 1):.2f}")
print(f"two-level cull: groups expanded per closest pass {d[8] / max(d[0], 1):.2f}, per shadow pass {d[9] / max(d[3], 1):.2f}; "
      f"object-level survivors per closest pass {d[10] / max(d[0], 1):.2f}, per shadow pass {d[11] / max(d[3], 1):.2f}")
