"""The reference's `render_lua(script)` (ch1/src/lua.rs:50-91) from the command line: run a Lua scene script through the
library's interpreter, render every Render / AddFrame call on the GPU, write the frames (PNG; PPM for ".ppm" names).
    python tools/render_lua.py SCRIPT.lua [OUT_DIR]
Needs an MI355X (there is no CPU path); prints what the script printed and the files written."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
from _bootstrap import package  # noqa: E402


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    rtc = package()
    prog = rtc.LuaProgram(path=argv[1])
    sys.stdout.write(prog.output)
    ctx = rtc.Context(0)
    t = time.perf_counter()
    paths = prog.render_to_files(ctx, argv[2] if len(argv) > 2 else ".")
    dt = time.perf_counter() - t
    for p in paths:
        print(p)
    print(f"{len(paths)} frames in {dt * 1e3:.1f} ms (rendering + PCIe + file writes)")
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
