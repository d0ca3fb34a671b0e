import json,glob,sys
tag=sys.argv[1]; pat=sys.argv[2] if len(sys.argv)>2 else "b512"
for f in sorted(glob.glob(f"gpurun_out/{tag}/layers_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
fs=sorted(glob.glob(f"gpurun_out/{tag}/layers_[0-9]*.json"))
rows={}
for f in fs:
    for l in json.load(open(f)):
        rows.setdefault(l["layer"],[]).append(l["ms"])
print("files:", [f.split('/')[-1] for f in fs])
for k,v in rows.items():
    if any(q in k for q in pat.split(",")): print("%-28s"%k+"".join("%8.3f"%x for x in v))
print("%-28s"%"sum"+"".join("%8.3f"%sum(rows[k][i] for k in rows) for i in range(len(fs))))
