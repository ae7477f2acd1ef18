"""Import helper: the package directory is `raytracer-challenge_amd/` (hyphen, as the layout
prescribes), which Python cannot import by name. This registers it as
`raytracer_challenge_amd` in sys.modules."""
from __future__ import annotations

import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
PKG_DIR = ROOT / "raytracer-challenge_amd"
NAME = "raytracer_challenge_amd"


def package():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(NAME, PKG_DIR / "__init__.py", submodule_search_locations=[str(PKG_DIR)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod
