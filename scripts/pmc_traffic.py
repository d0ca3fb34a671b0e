#!/usr/bin/env python3
"""HBM traffic per kernel launch from the two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE,
collected in separate runs with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).
Units and corrections from that guide: the counters are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced read stream, so it is doubled (our reads are 16 B/lane float4 streams; checked
against the layers whose algorithmic read is known exactly).
    python scripts/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv [forwards] > profiles/<name>.json
`forwards` = how many forwards each pass dispatched (`bench.py --pmc-pass N` runs N + 2 and prints the count); every kernel symbol's
dispatch count must be a whole multiple of it in both passes, or the script stops instead of mis-scaling the table.
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


def sum_by_kernel(path, counter):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "migan" in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            a = acc.setdefault(name, [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


fetch = sum_by_kernel(sys.argv[1], "FETCH_SIZE")
write = sum_by_kernel(sys.argv[2], "WRITE_SIZE")
# `bench.py --pmc-pass N` dispatches nothing but N + 2 identical forwards, one launch per layer.  The forward count comes from the caller
# (inferring it as the smallest dispatch count silently mis-scales every row when no symbol serves exactly one layer, or
# when a pass drops dispatches); without it the old inference is used and the table says so.
if len(sys.argv) > 3:
    forwards, inferred = int(sys.argv[3]), False
else:
    conv = [n for k, (_, n) in fetch.items() if "sepconv" in k or "cm_conv" in k]
    forwards, inferred = (min(conv) if conv else 1), True
for tbl, what in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
    # (the once-per-weight-epoch kernels -- weight planes, style / mapping set-up -- run once, not per forward: scaled like the others, exempt from the check)
    bad = {k: n for k, (_, n) in tbl.items() if n % forwards and n != 1}
    if bad:
        sys.exit(f"pmc_traffic.py: {what} pass: dispatch counts not a multiple of {forwards} forwards"
                 f"{' (inferred -- pass the count bench.py --pmc-pass printed)' if inferred else ''}: {bad}")
if set(fetch) != set(write):
    sys.exit(f"pmc_traffic.py: the two passes saw different kernel symbols: only FETCH {sorted(set(fetch) - set(write))}, only WRITE {sorted(set(write) - set(fetch))}")
out = {"_meta": {"kernel_source_sha": kernel_source_sha(), "forwards": forwards, "forwards_inferred": inferred,
                 "note": "per forward of `bench.py --pmc-pass` (the launches the roofline table times); bench.py compares the digest with the "
                         "sources of the library it runs and reports roofline.traffic_stale"}}
for k, (f, n) in fetch.items():
    w = write.get(k, (0.0, 0))[0]
    per_fwd = (2.0 * f + w) * 1024.0 / forwards
    out[k] = {"hbm_bytes_per_forward": per_fwd, "launches_per_forward": n // forwards if n % forwards == 0 else n / forwards,
              "fetch_bytes_per_forward": 2.0 * f * 1024.0 / forwards, "write_bytes_per_forward": w * 1024.0 / forwards,
              "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0 / n, "dispatches_sampled": n,
              "note": "FETCH_SIZE doubled (gfx950 wide-read correction), counters in KiB"}
json.dump(out, sys.stdout, indent=1)
