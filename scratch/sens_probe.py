import numpy as np, os, sys
sys.path.insert(0, '/root/repo')
from ma_amd import engine as E
from ma_amd.drivers import normalize
g = '/root/repo/tests/golden'
f = np.load(os.path.join(g, 'pairwise_e2e.npz')); png = np.load(os.path.join(g, 'data_png.npz'))
vecF = lambda a: np.asarray(a, dtype=np.float64).reshape(-1, order='F')
for tag, a, b, term in (("tvl1_rect", "rect_1", "rect_2", "SAD"), ("tvl2_rir", "rect_in_rect_1", "rect_in_rect_2", "SSD"), ("tvl1_slide", "sliding_rect_1", "sliding_rect_2", "SAD")):
    R = normalize(png[a].astype(np.float64)); T = normalize(png[b].astype(np.float64))
    m, n = R.shape
    lam, tau, iters, steps = f[f"{tag}_par"]
    sigma = (1 - 1e-4) / (8 * tau)
    eng = E.Engine(E.MODEL_TV_L1_REG if term == "SAD" else E.MODEL_TV_L2_REG, m, n, lam=float(lam))
    eng.set_images(vecF(T), vecF(R))
    en = f[f"{tag}_energies"]; sens = f[f"{tag}_sens"]
    out = []
    for s in range(int(steps)):
        eng.linearize(); r = eng.solve(1.0, float(tau), sigma, int(iters), 1e-3)
        ph, dh = eng.history()
        de = abs(ph[-1, 0] - en[s, 0]) / abs(en[s, 0])
        du = -1
        if f"{tag}_s{s}_u" in f: du = np.linalg.norm(eng.primal() - f[f"{tag}_s{s}_u"]) / np.linalg.norm(f[f"{tag}_s{s}_u"])
        if s == steps - 1: du = np.linalg.norm(eng.primal() - f[f"{tag}_u_final"]) / np.linalg.norm(f[f"{tag}_u_final"])
        out.append((s, de, sens[s, 1], du, sens[s, 0], ph.shape[0] == en[s, 2]))
    eng.close()
    print(tag)
    for o in out: print("  step %2d dE %.1e (sens %.1e)  du %.1e (sens %.1e) rows_ok %s" % o)
