#!/usr/bin/env python
"""Benchmark of the headline workload: CP iterations/s of the NN-TV registration hot path
(mf_nn_tv_registration_no_ref_ml inner iteration, finest level) at 4096^2 x K=16, fp64.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full Chambolle-Pock iteration (dual pass, K x K eigensolve, primal pass, history row).
Prints ONE JSON line (rank 0). See DESIGN.md section "Measurement" for the byte accounting.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic elements per pixel-image per CP iteration, SURVEY.md section 8(d)
ELEMS_ITER = {"tv": 37, "curvature": 31}
# dominant kernel: NN-TV runs one fused kernel per iteration (k_mf_fused_tv: the whole iteration's 37 elements).
# NN-curvature runs two field passes: k_mf_fused_curv (primal step, y1, y3, both Gram matrices, the SVT: 31 - 8 = 23 of the
# survey's elements) dominates; k_curv_y2 does the y2 update the survey charges 8 elements for (reads x(2) x_old(2) y2(2),
# writes y2(2)). With CP_TUNE_CURV_SPLIT=1 (the split passes of round 1) the dual pass (17) dominates.
CURV_SPLIT = os.environ.get("CP_TUNE_CURV_SPLIT", "0") not in ("", "0")
ELEMS_DOM = {"tv": 37, "curvature": 17 if CURV_SPLIT else 23}


def dominant_kernel(reg, k, bc, fp32, m=4096, nl=4096):
    """name of the dominant kernel as rocprofv3 prints it, derived from what cp_engine.hip launches for this configuration"""
    S = "float" if fp32 else "double"
    if reg == "tv":
        lin = "true" if bc == "linear" else "false"
        ka = 16 if k > 12 else (12 if k > 8 else (8 if k > 4 else 4))
        cw = os.environ.get("CP_TUNE_FUSED_CW", "1") if bc != "linear" else "1"       # lane-tile shape (tuning knob, default 1)
        if cw != "1":
            ka = 16
        waves = os.environ.get("CP_TUNE_FUSED_WAVES", "8")                             # columns per workgroup (tuning knob)
        big = ((m + 63) // 64 if m < 512 else 8) * ((nl + 7) // 8) >= 256                # cp_create: every CU gets a workgroup
        steps = -(-((m + 63) // 64) // min((m + 63) // 64, 8))                             # segments per task
        waves = "8" if (waves != "4" and k > 4 and bc != "linear" and cw == "1" and big and steps >= 8 and not fp32) else "4"
        parts = min((m + 63) // 64, 8)
        ks = waves == "4" and bc != "linear" and cw == "1" and parts * nl <= int(os.environ.get("CP_TUNE_FUSED_KS_TASKS", "256"))
        return f"k_mf_fused_tv<{S}, {lin}, {'true' if k == 16 else 'false'}, {ka}, false, {cw}, {waves}, {'true' if ks else 'false'}>"
    if CURV_SPLIT:
        return f"k_mf_dual<{S}, 0, 1>"
    ka = 16 if k > 12 else (12 if k > 8 else (8 if k > 4 else 4))
    return f"k_mf_fused_curv<{S}, {'true' if bc == 'neumann' else 'false'}, {'true' if k == 16 else 'false'}, {ka}>"


def measured_copy_gbs(dev_index):
    """device-to-device copy rate of this box (read + write bytes / time) by the library's own streaming kernel
    (cp_stream_copy_gbps: nontemporal, grid-stride; 8 bytes per lane = the access shape of the field kernels, and
    16 bytes per lane): the practical HBM ceiling next to the vendor peak (SURVEY.md section 8d asks for both)"""
    from ma_amd import engine as E
    return {"8B_per_lane": E.stream_copy_gbps(dev_index, 1 << 31, 8, 4), "16B_per_lane": E.stream_copy_gbps(dev_index, 1 << 31, 16, 4)}


def committed_traffic(kernel, m, n, k):
    """HBM bytes per launch of `kernel` on this workload from the newest committed rocprofv3 PMC measurement
    (profiles/*_traffic.json, written by scratch/prof.sh + scratch/traffic_json.py from separate --pmc passes of this very
    command; corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes). PMC counters cannot be read from inside
    the run, so this is the last measurement of the same kernel and shape, or None."""
    try:
        for f in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
            doc = json.load(open(os.path.join(ROOT, "profiles", f)))
            for e in doc.get("entries", []):
                if e.get("kernel") == kernel and e.get("shape") == [m, n, k]:
                    return e["hbm_bytes_per_launch"] / 1e9, f
    except Exception:
        pass
    return None, None


class _StdoutToStderr:
    """stdout carries the one JSON line of the contract; RCCL prints its version banner (NCCL_DEBUG=VERSION) with printf
    when a communicator is created, so file descriptor 1 points at stderr while the process group comes up"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def _init_process_group(dist, torch, backend, local_rank):
    # a single process without a launcher (CP_FORCE_SLAB_PATH=1 on one GPU): a one-rank group on the loopback address
    for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"),
                     ("MASTER_PORT", "29533")):
        os.environ.setdefault(key, val)
    with _StdoutToStderr():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()                      # communicators exist (and have printed) before stdout comes back
            torch.cuda.synchronize()
        else:
            dist.init_process_group(backend)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--cols", type=int, default=0, help="image columns (default: --size); a slab-shaped problem for tuning")
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--reg", default="tv", choices=["tv", "curvature"])
    ap.add_argument("--workload", default="nn", choices=["nn", "batch"],
                    help="nn: the headline (row slabs over the ranks); batch: BASELINE config 5, 256 independent TV-L2 "
                         "512^2 registrations sharded over the ranks, no communication")
    ap.add_argument("--problems", type=int, default=256)
    ap.add_argument("--fp32", action="store_true", help="CP_FLAG_FP32 storage mode (single GPU); not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-size", type=int, default=1024)
    ap.add_argument("--cpu-iters", type=int, default=40)
    return ap.parse_args()


def cpu_baseline(size, k, iters, reg):
    """The C + OpenMP restatement of the same iteration (oracle/cp_nntv.c: matrix-free, all host threads; NN-TV and
    NN-curvature) timed on a bounded sample of the workload. Returns (iterations/s at the sample size, seconds, threads,
    description)."""
    from ma_amd import synth
    from oracle import drivers as od, operators as oo
    m = n = size
    imgs = synth.dynamic_test_image(m, n, k).numpy()
    u0 = synth.smooth_displacement(m, n, k, amp=2.0).numpy().reshape(k, 2, n * m)
    u0_ref = np.transpose(u0, (2, 1, 0))
    omega = [0.0, float(m), 0.0, float(n)]
    h = (1.0, 1.0)
    mu, nu, tau, sigma = 0.2, 0.9 * 2.0 * k, 0.3, 0.3
    kmn = k * m * n
    from oracle import c_oracle
    bc, mu = ("neumann", mu) if reg == "tv" else ("dirichlet", 20.0)
    g = np.empty(2 * kmn)
    b = np.empty(kmn)
    T_u = np.empty((m, n, k))
    for i in range(k):
        T_u[:, :, i], gi = oo.evaluate_displacement(imgs[i], h, u0_ref[:, :, i], omega)
        g[2 * i * m * n: 2 * (i + 1) * m * n] = gi.reshape(-1, order="F")
        b[i * m * n: (i + 1) * m * n] = T_u[:, :, i].reshape(-1, order="F") - \
            (gi[:, 0] * u0_ref[:, 0, i] + gi[:, 1] * u0_ref[:, 1, i])
    x0 = np.concatenate([u0.reshape(-1), T_u.reshape(-1, order="F")])
    solve = c_oracle.nntv_chambolle_pock if reg == "tv" else c_oracle.nncurv_chambolle_pock
    y0 = np.zeros((6 if reg == "tv" else 4) * kmn)
    solve(m, n, k, bc, h, mu, nu, 1.0, tau, sigma, 1, None, g, b, x0, y0)   # warm-up
    t0 = time.perf_counter()
    solve(m, n, k, bc, h, mu, nu, 1.0, tau, sigma, iters, None, g, b, x0, y0)
    dt = time.perf_counter() - t0
    return iters / dt, dt, c_oracle.threads(), "C + OpenMP restatement oracle/cp_nntv.c (matrix-free, Gram + Jacobi SVD)"


def main_batch(args):
    """BASELINE config 5: --problems independent TV-L2 (SSD_registration) pairs of --size^2 (default 512 there) sharded
    over the ranks; a step = one CP iteration of every problem. No data-path collective."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    backend = os.environ.get("CP_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    if world > 1:
        _init_process_group(dist, torch, backend, local_rank)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from ma_amd import engine as E, synth
    from ma_amd.batch import shard_range
    m = n = args.size if args.size != 4096 else 512
    P = args.problems
    p0, p1 = shard_range(P, world, rank)
    fr = synth.dynamic_test_image(m, n, 2).numpy()
    # per-problem integer shifts from an LCG (seed 12345), SURVEY.md section 8d
    st, shifts = 12345, []
    for _ in range(P):
        st = (1103515245 * st + 12345) % (1 << 31)
        shifts.append(st % 9 - 4)
    T = np.concatenate([np.roll(fr[1], shifts[p], axis=0).reshape(-1, order="F") for p in range(p0, p1)])
    R = np.tile(fr[0].reshape(-1, order="F"), p1 - p0)
    esz = 4.0 if args.fp32 else 8.0
    eng = E.Engine(E.MODEL_TV_L2_REG, m, n, lam=50.0, batch=p1 - p0, device=local_rank, flags=E.FLAG_FP32 if args.fp32 else 0)
    eng.set_images(T, R)
    eng.set_primal(np.zeros(2 * m * n * (p1 - p0)))
    eng.linearize()
    tau = 4.0
    sigma = (1 - 1e-4) / (8 * tau)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eng.iterate_begin(1.0, tau, sigma, args.warmup + args.steps)
    eng.iterate_steps(args.warmup)
    eng.sync()
    for pss in ("dual", "primal", "fin"):
        eng.kernel_time_ms(pss)
    barrier()
    t0 = time.perf_counter()
    eng.iterate_steps(args.steps)
    eng.sync()
    barrier()
    dt = time.perf_counter() - t0
    eng.iterate_end()
    times = {pss: eng.kernel_time_ms(pss) for pss in ("dual", "primal", "fin")}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        dual_s = times["dual"][0] / max(times["dual"][1], 1) * 1e-3
        dual_bytes = 12 * esz * m * n * (p1 - p0)                 # dual pass: 12 of the 23 elements per pixel
        achieved = dual_bytes / dual_s / 1e9 if dual_s > 0 else 0.0
        out = {"metric": f"problem-iterations/sec, batch of {P} TV-L2 registrations {m}^2",
               "value": P * args.steps / dt, "unit": "problem-iterations/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f64" if not args.fp32 else "f64 arithmetic on f32 storage", "data": "synthetic",
               "config": {"workload": f"{P} x TV_registration + SSD_registration, {m}x{n}, lambda=50, tau=4, one CP iteration "
                                      f"of every problem per step", "parallelism": f"problems sharded x{world}",
                          "achieved_GBps_algorithmic": 23 * esz * m * n * P * args.steps / dt / 1e9,
                          "kernel_ms": {kk: v[0] / v[1] for kk, v in times.items() if v[1] > 0}},
               "roofline": {"bound": "hbm", "kernel": "k_pw_dual", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS, "traffic": None}}
        print(json.dumps(out), flush=True)
    eng.close()
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    args = parse()
    if args.workload == "batch":
        return main_batch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    # test hook for a one-GPU box: CP_BENCH_BACKEND=gloo puts every rank on device 0 and stages messages through the host
    backend = os.environ.get("CP_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    if world > 1 or os.environ.get("CP_FORCE_SLAB_PATH", "0") == "1":
        _init_process_group(dist, torch, backend, local_rank)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    from ma_amd import engine as E, synth
    from ma_amd import slab

    m = args.size
    n = args.cols if args.cols > 0 else args.size
    k = args.k
    reg = args.reg
    dev = torch.device("cuda", local_rank)
    model = E.MODEL_MF_NN_TV if reg == "tv" else E.MODEL_MF_NN_CURV

    # deterministic synthetic inputs (no RNG): frames generated on the GPU, handed over as host buffers
    imgs = synth.dynamic_test_image(m, n, k, device=dev)                      # (k, m, n)
    imgs_host = imgs.transpose(1, 2).contiguous().cpu().numpy().reshape(-1)   # k stacked column-major images
    del imgs
    u0 = synth.smooth_displacement(m, n, k, amp=2.0, device=dev)              # [k][2][n][m]
    torch.cuda.empty_cache()

    part = slab.Partition(n, world, rank)
    # NN-TV: bc neumann, mu 0.2 (headline); NN-curvature: bc dirichlet, mu 20
    bc, mu = ("neumann", 0.2) if reg == "tv" else ("dirichlet", 20.0)
    solver = slab.SlabSolver(model, m, n, k, bc=bc, h=(1.0, 1.0), omega=[0.0, float(m), 0.0, float(n)],
                             mu=mu, device=local_rank, part=part, flags=E.FLAG_FP32 if args.fp32 else 0)
    esz = 4.0 if args.fp32 else 8.0                            # bytes per stored element
    solver.set_images(imgs_host)
    del imgs_host
    # x0 = [u0; L0 = 0]; p0 = 0
    u0_loc = u0[:, :, part.j0:part.j1, :].contiguous().cpu().numpy().reshape(-1)
    del u0
    torch.cuda.empty_cache()
    x0 = np.concatenate([u0_loc, np.zeros(k * m * part.nl)])
    solver.set_primal(x0)
    del x0, u0_loc
    solver.linearize()
    nu = 0.9 * float(np.sum(solver.singular_values(0)))        # NNE/mf_nn_tv...:173-175
    solver.set_nu(nu)
    est, n_power = solver.norm_estimate()
    tau = sigma = math.sqrt(0.99 / est ** 2)                   # :195-196

    copy_gbs = measured_copy_gbs(local_rank) if rank == 0 else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one run of warmup + steps iterations: start-up (x_old = x0, the first dual step) before the warm-up, the timed
    # region is exactly `steps` CP iterations, the state is brought back to canonical form after it
    solver.iterate_begin(1.0, tau, sigma, args.warmup + args.steps)
    solver.iterate_steps(args.warmup)
    solver.sync()
    solver.kernel_times()                                      # drop warm-up timings
    barrier()
    t0 = time.perf_counter()
    solver.iterate_steps(args.steps)
    solver.sync()
    barrier()
    dt = time.perf_counter() - t0
    solver.iterate_end()
    solver.sync()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    times = solver.kernel_times()

    if rank == 0:
        kernel = dominant_kernel(reg, k, bc, args.fp32, m, part.nl)
        traffic, traffic_src = committed_traffic(kernel, m, part.nl, k) if world == 1 else (None, None)
        kmn_total = float(k) * m * n
        its = args.steps / dt
        bytes_iter = ELEMS_ITER[reg] * esz * kmn_total
        dual_ms, dual_n = times["dual"]
        dual_avg_s = dual_ms / max(dual_n, 1) * 1e-3
        # the dominant kernel processes this rank's slab; algorithmic bytes per launch = elements x 8 B x pixel-images
        dual_bytes = ELEMS_DOM[reg] * esz * float(k) * m * part.nl
        achieved = dual_bytes / dual_avg_s / 1e9 if dual_avg_s > 0 else 0.0
        out = {
            "metric": "CP iterations/sec, NN-TV registration 4096^2 x K=16" if (reg == "tv" and m == 4096 and n == 4096 and k == 16 and not args.fp32)
            else f"CP iterations/sec, NN-{reg} registration {m}x{n} x K={k}" + (" (fp32 storage mode)" if args.fp32 else ""),
            "value": its,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if not args.fp32 else "f64 arithmetic on f32 storage",
            "data": "synthetic",
            "config": {"workload": f"mf_nn_{reg}_registration_no_ref_ml inner CP iteration, finest level, "
                                   f"{m}x{n}, K={k}, {'fp32 storage' if args.fp32 else 'fp64'}, bc={bc}, theta=1, mu={mu}, nu from schedule, tau=sigma from "
                                   f"power iteration ({n_power} its)",
                       "parallelism": f"row slabs x{world}" if world > 1 else "single GPU",
                       "achieved_GBps_algorithmic": bytes_iter * its / 1e9,
                       "frac_of_hbm_peak": bytes_iter * its / 1e9 / (HBM_PEAK_GBS * world),
                       "kernel_ms": {kname: (v[0] / v[1]) for kname, v in times.items() if v[1] > 0}},
            # achieved / frac: ALGORITHMIC bytes of SURVEY.md section 8(d) per launch / measured launch time (the contract's
            # figure; it exceeds what the kernel moves because fusing removed passes the survey charges for).
            # hbm_achieved / hbm_frac: what the memory system did -- bytes per launch by the PMC counters (traffic) / the
            # same launch time.
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "GB per launch (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE, KiB units)",
                         "traffic_source": traffic_src,
                         "algorithmic_GB_per_launch": dual_bytes / 1e9,
                         "hbm_achieved": (traffic / dual_avg_s) if (traffic and dual_avg_s > 0) else None,
                         "hbm_frac": (traffic / dual_avg_s / HBM_PEAK_GBS) if (traffic and dual_avg_s > 0) else None,
                         # the Gram step on the matrix cores (v_mfma_f64_16x16x4_f64): two 16 x 16 Gram matrices over the
                         # pixels of the slab = 2 * 2 * 256 flop per pixel inside the dominant kernel (rocprofv3:
                         # SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 in profiles/); 78.6 TFLOP/s = AMD's data-sheet fp64 matrix peak.
                         # The kernel is HBM-bound by design: the matrix cores idle, they are used because the contraction
                         # costs nothing there.
                         "gram_mfma": {"flop_per_launch": 4.0 * 256 * m * part.nl,
                                       "achieved_TFLOPs": (4.0 * 256 * m * part.nl / dual_avg_s / 1e12) if dual_avg_s > 0 else None,
                                       "peak_TFLOPs": 78.6,
                                       "frac": (4.0 * 256 * m * part.nl / dual_avg_s / 1e12 / 78.6) if dual_avg_s > 0 else None},
                         "measured_copy_GBps": copy_gbs,
                         "hbm_frac_of_measured_copy": (traffic / dual_avg_s / max(copy_gbs.values())) if (traffic and dual_avg_s > 0 and copy_gbs) else None},
        }
        if not args.no_cpu_baseline:
            csize, citers = args.cpu_size, args.cpu_iters
            cits, cdt, ccores, cdesc = cpu_baseline(csize, k, citers, reg)
            scale = (csize * csize) / float(m * n)
            out["cpu_baseline"] = {
                "value": cits * scale, "unit": "iterations/s", "cores": ccores, "kind": "port",
                "sample": f"{cdesc}: {citers} CP iterations (history rows included) at "
                          f"{csize}^2 x K={k} in {cdt:.1f} s, scaled by pixel count to {m}^2"}
        print(json.dumps(out), flush=True)
    solver.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
