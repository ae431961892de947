"""Secondary workloads of SURVEY.md section 8(d): C2 (TV-L1 2048^2), C5 (256 x 512^2 TV-L2 batch), C4-curvature,
var-TV; prints iterations/s and algorithmic GB/s (elements per pixel from section 8d)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ma_amd import engine as E, synth

def run(name, eng, theta, tau, sigma, iters, bytes_iter):
    eng.iterate_begin(theta, tau, sigma, 2 * iters + 4)
    eng.iterate_steps(4); eng.sync()
    for p in ("dual", "primal", "eig", "fin"): eng.kernel_time_ms(p)
    t0 = time.perf_counter(); eng.iterate_steps(iters); eng.sync(); dt = time.perf_counter() - t0
    times = {p: eng.kernel_time_ms(p) for p in ("dual", "primal", "eig", "fin")}
    eng.iterate_end(); eng.sync()
    print(json.dumps({"workload": name, "it_per_s": iters / dt, "ms_per_it": dt / iters * 1e3,
                      "algorithmic_GBps": bytes_iter * iters / dt / 1e9,
                      "kernel_ms": {k: v[0] / max(v[1], 1) for k, v in times.items()}}), flush=True)

which = sys.argv[1:] or ["c2", "c5", "c4", "var"]
if "c2" in which:
    # BASELINE config 2 as worded: Data/rect_1.png -> rect_2.png (fixture copy tests/golden/data_png.npz) upsampled to
    # 2048^2 with the reference's own resampling facility (evaluate_displacement with s_grid, IRE/evaluate_displacement.m:27-31)
    m = n = 2048
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    png = np.load(os.path.join(gd, "data_png.npz"))
    nrm = lambda a: (a - a.min()) / (a.max() - a.min())
    R = E.evaluate_displacement(nrm(png["rect_1"].astype(np.float64)), (1.0, 1.0), s_grid=(m, n), want_grad=False)[0]
    T = E.evaluate_displacement(nrm(png["rect_2"].astype(np.float64)), (1.0, 1.0), s_grid=(m, n), want_grad=False)[0]
    eng = E.Engine(E.MODEL_TV_L1_REG, m, n, lam=0.7)
    eng.set_images(T.reshape(-1, order="F"), R.reshape(-1, order="F")); eng.set_primal(np.zeros(2*m*n)); eng.linearize()
    tau = 5.0; run("C2 TV-L1 rect_1 -> rect_2 upsampled to 2048^2", eng, 1.0, tau, (1 - 1e-4) / (8 * tau), 100, 23 * 8.0 * m * n); eng.close()
if "c5" in which:
    m = n = 512; B = 256
    fr = synth.dynamic_test_image(m, n, 2).numpy()
    T = np.tile(fr[1].reshape(-1, order="F"), B); R = np.tile(fr[0].reshape(-1, order="F"), B)
    for tag, flags, esz in (("fp64", 0, 8.0), ("fp32", E.FLAG_FP32, 4.0)):
        eng = E.Engine(E.MODEL_TV_L2_REG, m, n, lam=50.0, batch=B, flags=flags)
        eng.set_images(T, R); eng.set_primal(np.zeros(2*m*n*B)); eng.linearize()
        tau = 4.0; run(f"C5 TV-L2 256 x 512^2 {tag}", eng, 1.0, tau, (1 - 1e-4) / (8 * tau), 100, 23 * esz * m * n * B); eng.close()
for tag, model, el, mu in (("c4", E.MODEL_MF_NN_CURV, 31, 20.0), ("var", E.MODEL_VAR_TV, 28, 0.1), ("varc", E.MODEL_VAR_CURV, 22, 5.0)):
    if tag not in which: continue
    m = n = int(os.environ.get("BENCH_SIZE", "4096")); k = int(os.environ.get("BENCH_K", "16"))
    dev = torch.device("cuda", 0)
    imgs = synth.dynamic_test_image(m, n, k, device=dev).transpose(1, 2).contiguous().cpu().numpy().reshape(-1)
    u0 = synth.smooth_displacement(m, n, k, amp=2.0, device=dev).cpu().numpy().reshape(-1)
    torch.cuda.empty_cache()
    fl32 = E.FLAG_FP32 if os.environ.get("BENCH_FP32", "0") == "1" else 0
    eng = E.Engine(model, m, n, k=k, bc="dirichlet" if tag in ("c4", "varc") else "neumann", omega=[0.0, float(m), 0.0, float(n)], mu=mu, flags=fl32)
    eng.set_images(imgs)
    eng.set_primal(np.concatenate([u0, np.zeros(k*m*n)]) if tag == "c4" else u0); eng.linearize()
    if tag == "c4": eng.set_nu(30.0)
    run({"c4": "C4 NN-curvature", "var": "var-TV", "varc": "var-curvature"}[tag] + f" {m}^2 x {k}" + (" fp32 storage" if fl32 else ""), eng, 1.0, 0.1, 0.1, 20, el * (4.0 if fl32 else 8.0) * k * m * n); eng.close()
if "c1" in which:
    # C1: TV-L1 denoising of Data/rect1.png (fixture copy) with the committed salt & pepper mask, PDO/test_TV_L1_denoising.m
    from ma_amd import drivers as D
    gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    f = np.load(os.path.join(gd, "pairwise_e2e.npz"))
    noisy = f["c1_noisy"]
    D.tv_l1_denoising(noisy, lam=1.5, tau=0.02, maxIter=1000, tol=1e-5)                # warm-up (module load, first launches)
    t0 = time.perf_counter(); x, y, ph, dh, sigma = D.tv_l1_denoising(noisy, lam=1.5, tau=0.02, maxIter=1000, tol=1e-5)
    dt = time.perf_counter() - t0
    its = ph.shape[0] - 1
    print(json.dumps({"workload": f"C1 TV-L1 denoise {noisy.shape[0]}x{noisy.shape[1]} (whole call, launch-bound)",
                      "iterations": its, "seconds": dt, "it_per_s": its / dt}), flush=True)
if "c3" in which:
    # C3: mf_nn_tv_registration_no_ref_ml, K = 10 synthetic 1024^2 templates, all levels (whole driver, wall clock)
    from ma_amd import drivers as D
    m = n = 1024; k = 10
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    # parameters of the synthetic demo, NNE/demo_mf_nn_tv_registration_no_ref_ml.m:34-41
    optPara = dict(theta=1.0, maxIter=2000, tol=1e-3, outerIter=[16, 2], mu=0.2, nu_factor=[0.9, 0.9], bc="neumann", doPlots=False)
    for tag, fp32 in (("fp64", False), ("fp32 storage", True)):
        keep = {}
        t0 = time.perf_counter()
        u, L, SV = D.mf_nn_tv_registration_no_ref_ml([imgs[i] for i in range(k)], optPara, keep=keep, download_all=False, fp32=fp32)
        dt = time.perf_counter() - t0
        its = sum(v["iterations"] for v in keep.values())
        pw = sum(v["n_power"] for v in keep.values())
        print(json.dumps({"workload": f"C3 mf_nn_tv_registration_no_ref_ml K=10 1024^2, {D.num_levels(m, n)} levels, {tag}",
                          "seconds": dt, "cp_iterations": its, "power_iterations": pw, "outer_iterations": len(keep),
                          "cp_it_per_s_whole_driver": its / dt}), flush=True)

if "c4drv" in which:
    # C4 as worded: the whole mf_nn_curvature_registration_no_ref_ml driver, K = 16 synthetic 4096^2 templates, all levels, one GPU
    # (parameters of the kidney case of NNE/demo_mf_nn_curvature_registration_no_ref_ml.m:72-79)
    from ma_amd import drivers as D
    m = n = int(os.environ.get("C4_SIZE", "4096")); k = 16
    imgs = synth.dynamic_test_image(m, n, k, device=torch.device("cuda", 0)).cpu().numpy()
    torch.cuda.empty_cache()
    optPara = dict(theta=1.0, maxIter=2000, tol=5e-3, outerIter=[16, 2], mu=2e1, nu_factor=[0.95, 0.95], bc="dirichlet", doPlots=False)
    keep = {}
    t0 = time.perf_counter()
    tm = {}
    D._mf_nn_ml([imgs[i] for i in range(k)], optPara, "curvature", keep=keep, download_all=False, timings=tm)
    dt = time.perf_counter() - t0
    print(json.dumps({"host_side_seconds": {a: round(b, 3) for a, b in tm.items()},
                      "n_power_per_level_and_outer": {f"{lev}.{o}": v["n_power"] for (lev, o), v in keep.items() if lev >= 5}}), flush=True)
    its = sum(v["iterations"] for v in keep.values())
    per_level = {}
    for (lev, o), v in keep.items():
        per_level[lev] = per_level.get(lev, 0) + v["iterations"]
    secs = {}
    for (lev, o), v in keep.items():
        for kk, vv in v["seconds"].items():
            secs.setdefault(kk, {}); secs[kk][lev] = round(secs[kk].get(lev, 0.0) + vv, 3)
    print(json.dumps({"seconds_per_phase_and_level": secs}), flush=True)
    print(json.dumps({"workload": f"C4 mf_nn_curvature_registration_no_ref_ml K=16 {m}^2, {D.num_levels(m, n)} levels, fp64, one GPU",
                      "seconds": dt, "cp_iterations": its, "iterations_per_level": per_level,
                      "power_iterations": sum(v["n_power"] for v in keep.values()), "outer_iterations": len(keep)}), flush=True)
