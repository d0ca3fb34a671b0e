#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per (kernel, grid size) over one or more counter_collection.csv files: one row per
(kernel, grid), one column per counter, values per wave (counter / SQ_WAVES)."""
import collections
import csv
import sys

agg = collections.OrderedDict()
names = []
for path in sys.argv[1:]:
    try:
        rows = list(csv.DictReader(open(path)))
    except OSError as e:
        print("missing", path, e)
        continue
    for r in rows:
        if "migan" not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"].replace("migan::", "").split("(")[0], int(r["Grid_Size"]))
        agg.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] not in names:
            names.append(r["Counter_Name"])
names = [n for n in names if n != "SQ_WAVES"]
print(f"{'kernel':44s} {'threads':>9s} {'waves':>8s} " + " ".join(f"{n.replace('SQ_', '')[:14]:>14s}" for n in names))
for k, v in agg.items():
    m = lambda c: sum(v[c]) / len(v[c]) if v.get(c) else 0.0
    w = max(1.0, m("SQ_WAVES"))
    print(f"{k[0][:44]:44s} {k[1]:9d} {w:8.0f} " + " ".join(f"{m(n) / w:14.1f}" for n in names))
