"""experiments only: per-level cost of one NN-TV CP iteration (K=10, the sizes of config 3's pyramid)"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ma_amd import engine as E, synth
K = int(os.environ.get("LV_K", "10"))
for m in (32, 64, 128, 256, 512, 1024):
    imgs = synth.dynamic_test_image(m, m, K).numpy()
    eng = E.Engine(E.MODEL_MF_NN_TV, m, m, k=K, bc="neumann", h=(1.0, 1.0), mu=0.2)
    eng.set_images(imgs); eng.linearize(None)
    eng.set_nu(0.9 * float(np.sum(eng.singular_values(0))))
    est, _ = eng.norm_estimate()
    tau = np.sqrt(0.99) / est
    iters = 400
    eng.iterate_begin(1.0, tau, tau, 2 * iters + 8)
    eng.iterate_steps(4); eng.sync()
    for p in ("dual", "primal", "eig", "fin"): eng.kernel_time_ms(p)
    t0 = time.perf_counter(); eng.iterate_steps(iters); eng.sync(); dt = time.perf_counter() - t0
    times = {p: eng.kernel_time_ms(p) for p in ("dual", "eig")}
    eng.iterate_end(); eng.sync()
    print(f"m={m:5d} K={K}: {dt/iters*1e6:7.1f} us/it   " + "  ".join(f"{k} {v[0]/max(v[1],1)*1e3:6.1f} us" for k, v in times.items()), flush=True)
    eng.close()
