"""usage: python scratch/traffic_json.py <prof dir of scratch/prof.sh> <m> <n> <k> [more prof dirs ...]  > profiles/rNN_traffic.json
HBM bytes per launch of the field kernels from the separate FETCH_SIZE / WRITE_SIZE passes (median over the steady-state
dispatches), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: the counters report KiB;
FETCH_SIZE tallies 128-byte read requests at 64 bytes -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.
Counted at the L2 -> fabric boundary, so Infinity-Cache hits are included. bench.py reads the entries (roofline.traffic)."""
import collections, csv, glob, json, os, sys


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def collect(d, sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(d + f"/{sub}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


entries = []
args = sys.argv[1:]
while args:
    d, m, n, k = args[0], int(args[1]), int(args[2]), int(args[3])
    args = args[4:]
    fe, wr = collect(d, "pmc_fetch"), collect(d, "pmc_write")
    for kern in fe:
        if not any(t in kern for t in ("k_mf_fused", "k_mf_dual", "k_mf_primal", "k_curv_", "k_var_", "k_pw_march", "k_pw_fused")) or kern not in wr:
            continue
        fk, wk = med(fe[kern]["FETCH_SIZE"]), med(wr[kern]["WRITE_SIZE"])
        e = {"kernel": kern, "shape": [m, n, k], "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
             "read_bytes": 2 * fk * 1024, "write_bytes": wk * 1024, "hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024,
             "TCC_HIT": med(wr[kern].get("TCC_HIT_sum", [0])), "TCC_MISS": med(wr[kern].get("TCC_MISS_sum", [0])),
             "dispatches": len(fe[kern]["FETCH_SIZE"]), "source": os.path.basename(d.rstrip("/"))}
        entries.append(e)
# the hash scratch/prof.sh recorded ON THE BOX beside the measurements (bench.source_hash() of the profiled tree): a measurement
# is only valid for the sources it was taken on (bench.committed_traffic compares it with the tree being benched)
hashes = {open(os.path.join(d, "source_hash.txt")).read().strip() for d in sys.argv[1::4]}
assert len(hashes) == 1, f"profile directories of different source trees: {hashes}"
print(json.dumps({"note": __doc__.split("\n", 1)[1].strip(), "source_hash": hashes.pop(), "entries": entries}, indent=1))
