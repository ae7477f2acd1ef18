"""Static instruction mix of the k_trace variants from hipcc -S output (tools: see DESIGN.md).
usage: python tools/isa_stats.py /tmp/k.s"""
import re
import sys

s = open(sys.argv[1]).read()
for name, label in (("_Z7k_traceILi3ELb0ELb0ELb0E", "flat cull"), ("_Z7k_traceILi3ELb1ELb0ELb0E", "reflect cull"),
                    ("_Z7k_traceILi3ELb1ELb1ELb0E", "refract cull"), ("_Z7k_traceILi4ELb0ELb0ELb0E", "flat cull2")):
    i = s.find("\n" + name)
    if i < 0:
        continue
    j = s.find("s_endpgm", i)
    lines = [l.strip() for l in s[i:j].split("\n")]
    lines = [l for l in lines if l and not l.startswith((";", ".", "_Z")) and not l.endswith(":")]
    v = [l for l in lines if l.startswith("v_")]
    print(f"{label:13s} instr {len(lines):5d}  valu {len(v):5d}  v_fma(c)_f64 {sum(('v_fma_f64' in l or 'v_fmac_f64' in l) for l in v):4d}  "
          f"v_mul/add_f64 {sum(l.startswith(('v_mul_f64', 'v_add_f64')) for l in v):4d}  salu {sum(l.startswith('s_') for l in lines):5d}  "
          f"scratch {sum(l.startswith('scratch_') for l in lines):4d}")
