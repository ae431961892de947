"""usage: trace_gaps.py <kernel_trace.csv>: per-iteration timeline of the steady state (kernel durations and the idle gaps
between consecutive kernels), averaged over the iterations between two k_mf_fused_tv launches"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][:48] for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
idx = [i for i, n in enumerate(names) if "k_mf_fused_tv" in n or "k_mf_fused_curv" in n]
idx = idx[len(idx) // 2:]            # second half: steady state
agg = collections.OrderedDict()
cnt = 0
for a, b in zip(idx[:-1], idx[1:]):
    cnt += 1
    for pos, i in enumerate(range(a, b)):
        key = (pos, names[i])
        d = agg.setdefault(key, [0.0, 0.0])
        d[0] += (en[i] - st[i]) / 1e3
        d[1] += (st[i + 1] - en[i]) / 1e3
print("iterations averaged:", cnt, " period us:", (st[idx[-1]] - st[idx[0]]) / 1e3 / max(cnt, 1))
for (pos, n), (dur, gap) in agg.items():
    print(f"{pos:2d} {n:50s} dur {dur / cnt:9.1f} us   gap after {gap / cnt:8.1f} us")
