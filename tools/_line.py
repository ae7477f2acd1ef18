"""Print selected (dotted) keys of the last JSON line on stdin: python3 tools/_line.py <label> key [key ...]"""
import json
import sys

d = json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1])
out = [sys.argv[1]]
for k in sys.argv[2:]:
    v = d
    for part in k.split("."):
        v = v.get(part) if isinstance(v, dict) else None
    out.append(f"{k.split('.')[-1]}={v}")
print(" ".join(out))
