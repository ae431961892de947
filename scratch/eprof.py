"""experiments only: phase timestamps of k_mf_efused (library built with -DCP_EPROF, CP_ENGINE_LIB points to it)"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from ma_amd import engine as E, synth
for (m, K) in ((64, 10), (256, 10), (1024, 10), (1024, 16)):
    imgs = synth.dynamic_test_image(m, m, K).numpy()
    eng = E.Engine(E.MODEL_MF_NN_TV, m, m, k=K, bc="neumann", h=(1.0, 1.0), mu=0.2)
    eng.set_images(imgs)
    eng.linearize(None)
    eng.set_nu(0.9 * float(np.sum(eng.singular_values(0))))
    est, _ = eng.norm_estimate()
    tau = np.sqrt(0.99) / est
    t0 = time.perf_counter()
    eng.solve(1.0, tau, tau, 300, 0.0)
    dt = time.perf_counter() - t0
    out = (C.c_ulonglong * 32)()
    E.lib.cp_debug_eprof(out)
    ts = [out[i] for i in range(8)]
    d = [(ts[i + 1] - ts[i]) / 100.0 for i in range(7)]
    print(f"m={m} K={K} solve {dt/300*1e6:.1f} us/it  phases us: reduce {d[0]:.2f} helmert {d[1]:.2f} warm-xform {d[2]:.2f} jacobi {d[3]:.2f} "
          f"back {d[4]:.2f} l1 {d[5]:.2f} W+hist {d[6]:.2f} total {(ts[7]-ts[0])/100.0:.2f}  sweeps {out[20]} {out[21]}", flush=True)
    eng.close()
