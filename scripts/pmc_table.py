#!/usr/bin/env python3
"""Per-layer table from a rocprofv3 --pmc counter_collection.csv + bench --dump-layers json."""
import collections, csv, json, sys
pmc, layers = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(pmc)))
agg = collections.OrderedDict()
for r in rows:
    if 'migan' not in r['Kernel_Name']: continue
    k = (r['Kernel_Name'], int(r['Grid_Size']) // 256)
    agg.setdefault(k, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
L = json.load(open(layers))
print(f"{'layer':24s} {'ms':>7s} {'WGs':>6s} {'valu/w':>7s} {'mfma/w':>7s} {'salu/w':>7s} {'lds/w':>6s} {'valu_cyc':>9s} {'mfma_cyc':>9s} {'wave_cyc':>9s} {'wait_cyc':>9s} {'xideal':>6s}")
used = collections.Counter()
for l in L:
    cands = [k for k in agg if l['kernel'] in k[0]]
    # match by launch order: same kernel may serve several layers with distinct grid sizes
    wg32 = None
    for k in cands:
        if used[(k, l['layer'])] == 0 and k[1] == l['workgroups_batch1'] * 32 // (1 if True else 1):
            wg32 = k
    if wg32 is None:
        for k in cands:
            if abs(k[1] - l['workgroups_batch1'] * 32) <= l['workgroups_batch1'] * 32 * 0.9 and k[1] >= l['workgroups_batch1'] * 4: wg32 = wg32 or k
    if wg32 is None: continue
    v = agg[wg32]; m = lambda c: sum(v[c]) / len(v[c]) if v.get(c) else 0.0
    w = max(1.0, m('SQ_WAVES'))
    ideal = max(l['mfma_flops'] * 32 / 157.3e12, l['bytes'] * 32 / 6.3e12) * 1e3
    print(f"{l['layer']:24s} {l['ms']:7.3f} {wg32[1]:6d} {m('SQ_INSTS_VALU')/w:7.0f} {m('SQ_INSTS_MFMA')/w:7.0f} {m('SQ_INSTS_SALU')/w:7.0f} {m('SQ_INSTS_LDS')/w:6.0f} "
          f"{4*m('SQ_ACTIVE_INST_VALU')/w:9.0f} {64*m('SQ_INSTS_MFMA')/w:9.0f} {4*m('SQ_WAVE_CYCLES')/w:9.0f} {4*m('SQ_WAIT_ANY')/w:9.0f} {l['ms']/max(ideal,1e-9):6.1f}")
print("total ms", round(sum(l['ms'] for l in L), 3))
