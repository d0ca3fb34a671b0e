"""profiles/r06_sq_counters.md from the CSVs of scripts/gpu_sq_counters.sh: per kernel symbol of one migan-512 forward (batch 32, whole-batch launches)
the dynamic instruction counts and the issue / wait shares the SQ counters give, next to the hipEvent duration of the same launches.

usage: python scripts/sq_counters.py gpurun_out/sq > profiles/r06_sq_counters.md
Units (MI355X_MICROARCH.md): SQ_INSTS_* count wave instructions; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES and SQ_LDS_IDX_ACTIVE count cycles summed over SIMDs resp. CUs.  256 CUs x 4 SIMDs."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{d}/cc_*.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = json.load(open(f"{d}/per_launch.json"))
ms = collections.defaultdict(list)
layers = collections.defaultdict(list)
for r in rows:
    if r["layer"].endswith(".dwfir") and "pipedown" in r["kernel"]:
        continue                 # (the fused down=2 launch is listed under the layer name; its `.dwfir` row is a placeholder)
    ms[r["kernel"]].append(r["ms"])
    layers[r["kernel"]].append(r["layer"])


def mean(k, c):
    v = agg[k].get(c)
    return sum(v) / len(v) if v else float("nan")


print("# SQ counters per kernel of one migan-512 forward (batch 32, fp32; round 6)\n")
print("`scripts/gpu_sq_counters.sh` (five `rocprofv3 --kernel-trace --pmc` passes of `bench.py --pmc-pass 4`) + `scripts/sq_counters.py`; one row per kernel")
print("symbol, averages per launch. `ms` = hipEvent duration of the same whole-batch launch (un-profiled run of the same visit). Derived columns:")
print("`VALU/SIMD/us` = wave64 VALU instructions (incl. MFMA) issued per SIMD and microsecond (1024 SIMDs); `cyc/VALU` = SIMD cycles per VALU")
print("instruction at the 2.0 GHz these kernels sustain (`r03_ubench_valu_issue_rates.txt`: a SIMD issues one per 1.0 - 2.4 cycles with eight waves,")
print("one per 1.9 - 3.1 with three, one per 4.6 - 5.2 from a lone wave); `MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration);")
print("`LDS busy` = SQ_LDS_IDX_ACTIVE / (256 CUs x duration), `conflict` = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; `wait` = SQ_WAIT_INST_ANY /")
print("SQ_WAVE_CYCLES (share of a wave's life at an s_waitcnt), `issue` = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES.\n")
print("| kernel (layers) | launches | ms | VALU instr | MFMA | LDS instr | VALU/SIMD/us | cyc/VALU | MFMA busy | LDS busy | conflict | wait | issue |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
order = sorted((k for k in ms if any(k in a for a in agg)), key=lambda k: -sum(ms[k]))
GHZ = 2.0
for k in order:
    full = next(a for a in agg if k in a)
    n = len(ms[k])
    t = sum(ms[k]) / n                     # ms per launch
    valu, mfma, lds = mean(full, "SQ_INSTS_VALU"), mean(full, "SQ_INSTS_MFMA"), mean(full, "SQ_INSTS_LDS")
    per_simd_us = valu / 1024 / (t * 1e3)
    cyc = (t * 1e-3 * GHZ * 1e9) / (valu / 1024)
    mfma_busy = mean(full, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * t * 1e-3 * GHZ * 1e9)
    lds_busy = mean(full, "SQ_LDS_IDX_ACTIVE") / (256 * t * 1e-3 * GHZ * 1e9)
    confl = mean(full, "SQ_LDS_BANK_CONFLICT") / max(mean(full, "SQ_LDS_IDX_ACTIVE"), 1.0)
    wait = mean(full, "SQ_WAIT_INST_ANY") / mean(full, "SQ_WAVE_CYCLES")
    issue = mean(full, "SQ_ACTIVE_INST_ANY") / mean(full, "SQ_WAVE_CYCLES")
    ls = layers[k]
    lab = ", ".join(x.replace("encoder.", "e.").replace("synthesis.", "s.") for x in ls[:3]) + (" ..." if len(ls) > 3 else "")
    print(f"| `{k.replace('migan::', '')[:64]}` ({lab}) | {n} | {t:.3f} | {valu:.3g} | {mfma:.3g} | {lds:.3g} | {per_simd_us:.0f} | {cyc:.1f} | "
          f"{mfma_busy:.0%} | {lds_busy:.0%} | {confl:.0%} | {wait:.0%} | {issue:.0%} |")
