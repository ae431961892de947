"""experiments only: tests/fuzz_cases.py over many random cases; prints one line per case and exits non-zero when a deviation
exceeds the tests' bars (1e-10 on x and y, 1e-9 on the histories). usage: fuzz_parity.py [count] [seed]"""
import os, sys, json
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from fuzz_cases import run_case, run_pair_case

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for t in range(count):
    r = run_pair_case(t, rng) if os.environ.get("FUZZ_PAIR", "0") == "1" else run_case(t, rng)
    bad += not r["ok"]
    print(json.dumps(r), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
