"""profiles/r06_migan256_storage_modes.md from `bench.py --model migan-256 --dtype f32|bf16|f16 --dump-layers` of one GPU visit:
usage: python scripts/storage_modes_table.py gpurun_out/m256 > profiles/r06_migan256_storage_modes.md"""
import json
import sys

src = sys.argv[1]
d, val = {}, {}
for dt in ("f32", "bf16", "f16"):
    line = [x for x in open(f"{src}/b_{dt}.json") if x.startswith("{")]
    b = json.loads(line[-1])
    val[dt] = (b["value"], b["ms_per_step"])
    d[dt] = {}
    for r in json.load(open(f"{src}/pl_{dt}.json")):
        k = r["layer"].replace(".dwfir", "")
        e = d[dt].setdefault(k, [0.0, []])
        e[0] += r["ms"]
        e[1].append(r["kernel"].replace("migan::", "").split("(")[0][:48])
print("# migan-256, batch 32 (BASELINE configs[1]), per layer: fp32 storage (pipelined / 256-pixel-tile kernels where they exist) against 16-bit")
print("# storage (one-tile kernels)\n")
print("`bench.py --model migan-256 --dtype f32|bf16|f16 --dump-layers` on ONE box (round 6; hipEvent per launch, one stream, whole-batch launches, ms per 32")
print("images; a down=2 layer = its `dwfir` launch + its pointwise launch where it is not fused). Forward, two sub-batch streams: "
      + ", ".join(f"{k} {v[0]:.0f} images/s ({v[1]:.3f} ms per step)" for k, v in val.items()) + ".\n")
print("| layer | fp32 ms | fp32 kernel | bf16 ms | f16 ms | 16-bit kernel | bf16 / fp32 |")
print("|---|---|---|---|---|---|---|")
tot = {k: 0.0 for k in d}
for k in d["f32"]:
    a, b, c = d["f32"][k], d["bf16"].get(k, [0.0, ["-"]]), d["f16"].get(k, [0.0, ["-"]])
    for t in tot:
        tot[t] += d[t].get(k, [0.0])[0]
    if a[0] < 0.03:
        continue
    print(f"| `{k}` | {a[0]:.3f} | `{' + '.join(dict.fromkeys(a[1]))}` | {b[0]:.3f} | {c[0]:.3f} | `{' + '.join(dict.fromkeys(b[1]))}` | {b[0] / a[0]:.2f} |")
print(f"| all launches | {tot['f32']:.3f} | | {tot['bf16']:.3f} | {tot['f16']:.3f} | | {tot['bf16'] / tot['f32']:.2f} |")
