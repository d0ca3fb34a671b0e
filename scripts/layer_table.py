#!/usr/bin/env python3
"""Print per-layer ms of several `bench.py --dump-layers` files side by side: python scripts/layer_table.py <dir> <tag>... [-- layer-substring...]"""
import json, sys
args = sys.argv[1:]
filt = []
if "--" in args:
    i = args.index("--"); filt = args[i + 1:]; args = args[:i]
d, tags = args[0], args[1:]
def load(t):
    for pat in (f"{d}/layers_{t}.json", f"{d}/{t}.json"):
        try:
            return {l["layer"]: (l["ms"], l["kernel"]) for l in json.load(open(pat))}
        except FileNotFoundError:
            pass
    raise SystemExit(f"no layer dump for {t}")
T = {t: load(t) for t in tags}
names = list(T[tags[0]].keys())
print("%-28s" % "layer", " ".join("%-9s" % t[:9] for t in tags))
for n in names:
    if filt and not any(f in n for f in filt):
        continue
    print("%-28s" % n, " ".join("%-9.4f" % T[t][n][0] for t in tags), T[tags[-1]][n][1][:44])
print("%-28s" % "sum", " ".join("%-9.3f" % sum(v[0] for v in T[t].values()) for t in tags))
for t in tags:
    try:
        b = json.loads(open(f"{d}/bench_{t}.json").read().strip().splitlines()[-1])
        print("%-12s %.0f img/s  %.3f ms/step" % (t, b["value"], b["ms_per_step"]))
    except Exception:
        pass
