import sys, glob, csv, collections, os
out = sys.argv[1]
def short(n):
    return n.split('(')[0][:70]
# kernel stats
for f in sorted(glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)[-1:]:
    print('== kernel stats', f)
    rows = list(csv.DictReader(open(f)))
    print("(whole process: set-up -- synthetic frames, warp, 100 power-iteration steps -- is included; the timed region of "
          "bench.py launches only k_mf_fused_tv, k_mf_reduce_stage1 and k_mf_efused)")
    for i, r in enumerate(rows):
        if i >= 14 and not any(t in r['Name'] for t in ('k_mf_', 'k_pw_', 'k_var_')):
            continue
        print(f"{short(r['Name']):70s} calls {r['Calls']:>6s} avg_ns {float(r['AverageNs']):>14.0f} total% {r['Percentage']}")
# pmc
for sub in ('pmc_sq', 'pmc_sq2', 'pmc_fetch', 'pmc_write'):
    for f in sorted(glob.glob(out + f'/{sub}/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1:]:
        print('== pmc', sub)
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in agg.items():
            if not ('k_mf_fused' in k or 'k_mf_efused' in k or 'k_pw_' in k):
                continue
            # steady-state launches: take the median over dispatches
            s = ' '.join(f"{c}={sorted(v)[len(v)//2]:.4g}" for c, v in d.items())
            print(f"{k:60s} n={len(next(iter(d.values())))} {s}")
