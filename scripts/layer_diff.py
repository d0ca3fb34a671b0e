#!/usr/bin/env python
"""Compare two per-layer dumps written by `bench.py --dump-layers`: layer_diff.py OLD.json NEW.json"""
import json
import sys

old = {d["layer"]: d for d in json.load(open(sys.argv[1]))}
new = {d["layer"]: d for d in json.load(open(sys.argv[2]))}
for k, d in new.items():
    o = old.get(k, {}).get("ms", 0.0)
    flag = "" if abs(d["ms"] - o) < 0.02 else ("  <-- faster" if d["ms"] < o else "  <-- SLOWER")
    print(f"{k:30s} {o:7.3f} -> {d['ms']:7.3f}{flag}")
print(f"{'sum':30s} {sum(d['ms'] for d in old.values()):7.3f} -> {sum(d['ms'] for d in new.values()):7.3f}")
