import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ma_amd import engine as E, synth
m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fp32 = int(sys.argv[4]); iters = int(sys.argv[5])
rng = np.random.default_rng(1)
imgs = synth.dynamic_test_image(m, n, k).numpy()
u0 = 0.4 * rng.standard_normal((k, 2, n * m))
kmn = k * m * n
out = {}
for r2 in ("0", "1"):
    os.environ["CP_TUNE_Y2_R2"] = r2
    eng = E.Engine(E.MODEL_MF_NN_CURV, m, n, k=k, bc="dirichlet", h=(1.0, 1.0), omega=[0.0, float(m), 0.0, float(n)], mu=20.0, flags=E.FLAG_FP32 if fp32 else 0)
    eng.set_images(np.concatenate([imgs[i].reshape(-1, order="F") for i in range(k)]))
    x0 = np.concatenate([u0.reshape(-1), np.zeros(kmn)])
    eng.set_primal(x0); eng.linearize(); eng.set_nu(1.5 * k)
    y0 = 0.01 * np.random.default_rng(3).standard_normal(4 * kmn)
    eng.set_dual(y0)
    eng.solve(1.0, 0.1, 0.1, iters, 0.0)
    out[r2] = (eng.primal(), eng.dual(), *eng.history())
    eng.close()
x0_, y0_, p0, d0 = out["0"]; x1_, y1_, p1, d1 = out["1"]
print("x diff", np.abs(x0_ - x1_).max(), "y diff", np.abs(y0_ - y1_).max())
yd = np.abs(y0_ - y1_).reshape(4 * k, n, m)
for f in range(4 * k):
    if yd[f].max() > 1e-6:
        jj, ii = np.nonzero(yd[f] > 1e-6)
        print("field", f, "max", yd[f].max(), "count", len(ii), "rows", sorted(set(ii.tolist()))[:12], "cols", sorted(set(jj.tolist()))[:12])
print(p0[:3, :4]); print(p1[:3, :4])
