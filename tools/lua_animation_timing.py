"""Time lua.rs's render_lua on the GPU for the orbit script (rtc_lua_program_render: one launch per AddFrame, pipelined,
frames copied to the host) against the same jobs rendered one by one through rtc_render_rgb8. python tools/lua_animation_timing.py"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
from _bootstrap import package  # noqa: E402

rtc = package()
data = Path(rtc.__file__).resolve().parent / "data"
rtc.LuaProgram(text="x = 1")   # (loads the library)
for frames, balls, w, h in ((120, 100, 1920, 1080), (120, 24, 600, 400)):
    text = f"FRAMES = {frames} BALLS = {balls} WIDTH, HEIGHT = {w}, {h}\n" + (data / "orbit_animation.lua").read_text()
    t = time.perf_counter()
    prog = rtc.LuaProgram(text=text, base_dir=data)
    t_script = time.perf_counter() - t
    jobs = prog.jobs
    ctx = rtc.Context(0)
    count = [0]
    prog.render(ctx, on_frame=lambda *a: count.__setitem__(0, count[0] + 1))     # warm
    t = time.perf_counter()
    _, st = prog.render(ctx, on_frame=lambda *a: None, with_stats=True)
    t_pipe = time.perf_counter() - t
    dw = ctx.upload(jobs[0].world)
    out = rtc.host_canvas_rgb8(h, w)
    dw.render_rgb8(jobs[0].camera, out=out)
    t = time.perf_counter()
    for j in jobs[:frames]:
        dw.render_rgb8(j.camera, out=out)
    t_serial = time.perf_counter() - t
    rays = sum(st[k] for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract"))
    print(f"{w}x{h}, {len(jobs[0].world)} shapes, {len(jobs)} jobs: script {t_script * 1e3:.1f} ms; rtc_lua_program_render {t_pipe / len(jobs) * 1e3:.3f} ms per frame "
          f"({rays / t_pipe / 1e9:.1f} Grays/s all rays, frames on the host); one rtc_render_rgb8 per frame {t_serial / frames * 1e3:.3f} ms", flush=True)
    ctx.close()
