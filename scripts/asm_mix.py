#!/usr/bin/env python3
"""Static instruction mix of one kernel's ISA (hipcc -S --cuda-device-only), per barrier-delimited segment.
usage: asm_mix.py file.s"""
import collections
import sys

lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN5migan") and ":" in l][0]
end = [i for i, l in enumerate(lines) if "s_endpgm" in l and i > start][0]
seg, segs = 0, collections.defaultdict(collections.Counter)
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_barrier":
        seg += 1
        continue
    cls = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
           "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_")) else "other")
    segs[seg][cls] += 1
    segs[seg]["_" + op] += 1
tot = collections.Counter()
for k in sorted(segs):
    c = segs[k]
    print("segment", k, {x: c[x] for x in ("valu", "mfma", "salu", "lds", "vmem")})
    top = sorted(((v, o) for o, v in c.items() if o.startswith("_v_") or o.startswith("_ds_")), reverse=True)[:18]
    print("    ", " ".join(f"{o[1:]}:{v}" for v, o in top))
    for x in ("valu", "mfma", "salu", "lds", "vmem"):
        tot[x] += c[x]
print(dict(tot))
