"""time of cp_norm_estimate (power iteration of NNE/matrix_norm.m on the device) at 4096^2 x 16 for grids of its two kernels"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ma_amd import engine as E, synth
m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = 16
reg = sys.argv[2] if len(sys.argv) > 2 else "curvature"
dev = torch.device("cuda", 0)
print("setup", flush=True)
imgs = synth.dynamic_test_image(m, n, k, device=dev).transpose(1, 2).contiguous().cpu().numpy().reshape(-1)
u0 = synth.smooth_displacement(m, n, k, amp=2.0, device=dev).cpu().numpy().reshape(-1)
x0 = np.concatenate([u0, np.zeros(k * m * n)])
for grid in sys.argv[3:]:
    os.environ["CP_TUNE_POW_GRID"] = grid
    eng = E.Engine(E.MODEL_MF_NN_CURV if reg == "curvature" else E.MODEL_MF_NN_TV, m, n, k=k, bc="dirichlet" if reg == "curvature" else "neumann",
                   h=(1.0, 1.0), omega=[0.0, float(m), 0.0, float(n)], mu=20.0 if reg == "curvature" else 0.2)
    eng.set_images(imgs); eng.set_primal(x0); eng.linearize()
    print("linearized", flush=True)
    eng.norm_estimate()
    print("first estimate done", flush=True)
    t0 = time.perf_counter()
    est, it = eng.norm_estimate()
    dt = time.perf_counter() - t0
    print(f"{reg} {m}^2 grid {grid}: est {est:.12f} steps {it} {dt*1e3/it:.2f} ms per step", flush=True)
    eng.close()
