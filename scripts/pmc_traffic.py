#!/usr/bin/env python3
"""HBM traffic per kernel launch from the two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE,
collected in separate runs with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).
Units and corrections from that guide: the counters are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced read stream, so it is doubled (our reads are 16 B/lane float4 streams; checked
against the layers whose algorithmic read is known exactly).
    python scripts/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv > profiles/<name>.json
"""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha():
    """same digest as bench.py::kernel_source_sha: the sources the measured library was built from"""
    csrc = os.path.join(ROOT, "mi-gan_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hpp", ".h", ".inc", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def avg_by_kernel(path, counter):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "migan" in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            acc.setdefault(name, []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch = avg_by_kernel(sys.argv[1], "FETCH_SIZE")
write = avg_by_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_meta": {"kernel_source_sha": kernel_source_sha(),
                 "note": "bench.py compares this digest with the sources of the library it runs and reports roofline.traffic_stale"}}
for k in fetch:
    f, n = fetch[k]
    w = write.get(k, (0.0, 0))[0]
    out[k] = {"fetch_bytes_per_launch": 2.0 * f * 1024.0, "write_bytes_per_launch": w * 1024.0,
              "hbm_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0, "dispatches_sampled": n,
              "note": "FETCH_SIZE doubled (gfx950 wide-read correction), counters in KiB"}
json.dump(out, sys.stdout, indent=1)
