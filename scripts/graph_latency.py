#!/usr/bin/env python3
"""Experiment: batch-1 latency of the forward replayed from a captured HIP graph vs launched eagerly (python scripts/graph_latency.py [res])."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("mi-gan_amd")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model = pkg.Generator(R)
model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pkg.synth.make_state_dict(R, seed=0).items()})
model = model.to("cuda").eval()
x = torch.from_numpy(pkg.synth.make_input(1, R, seed=1)).to("cuda")
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[0], ts[len(ts) // 2]
with torch.no_grad():
    y0 = model(x).clone()
    print("eager  min/median ms", timeit(lambda: model(x)))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): model(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            y = model(x)
        g.replay(); torch.cuda.synchronize()
        print("graph  max|diff|", float((y - y0).abs().max()))
        print("graph  min/median ms", timeit(lambda: g.replay()))
    except Exception as e:
        print("capture failed:", repr(e)[:400])
