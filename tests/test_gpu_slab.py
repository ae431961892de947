"""Row-slab multi-GPU path on ONE GPU: 2 and 3 ranks (gloo, device tensors staged through the host) against the
single-handle result. The 8-GPU RCCL run itself belongs to the driver's scaling bench."""
import os
import subprocess
import sys

import pytest
from conftest import run_command

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,k,reg,bc", [(2, 16, "tv", "neumann"), (3, 5, "tv", "neumann"),
                                            (2, 16, "curvature", "dirichlet"), (3, 4, "curvature", "neumann")])
def test_slabs_match_single_gpu(world, k, reg, bc):
    env = dict(os.environ, SLAB_K=str(k), SLAB_REG=reg, SLAB_BC=bc, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29400 + world), os.path.join(ROOT, "tests", "slab_worker.py")]
    r = run_command(cmd, env=env, timeout=200)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0


@pytest.mark.parametrize("world", [2, 3])
def test_slabs_with_the_wave_hand_over_kernel(world):
    """the 8-columns-per-workgroup instantiation of k_mf_fused_tv (LDS hand-over between neighbouring waves) is the default
    only for large problems; forced here (CP_TUNE_FUSED_WAVES=8) so that its slab edges -- the last owned column takes its
    (i, j+1) neighbour from the halo column, column groups that end inside a workgroup -- meet the single-handle result"""
    env = dict(os.environ, SLAB_K="16", SLAB_REG="tv", SLAB_BC="neumann", MASTER_ADDR="127.0.0.1", CP_TUNE_FUSED_WAVES="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29440 + world), os.path.join(ROOT, "tests", "slab_worker.py")]
    r = run_command(cmd, env=env, timeout=200)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0


def test_rccl_backend_single_rank():
    """the multi-GPU driver over the RCCL backend (device-tensor all-gather, torch stream hand-over) on one rank"""
    env = dict(os.environ, CP_FORCE_SLAB_PATH="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29431", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--size", "512", "--k", "16",
           "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    r = run_command(cmd, env=env, timeout=250)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0 and '"metric"' in r.stdout


@pytest.mark.parametrize("world,reg", [(2, "tv"), (3, "curvature")])
def test_bench_row_slabs_multi_rank(world, reg):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one rank per slab), except that the
    ranks share one GPU and talk through gloo: one JSON line from rank 0, strong scaling, all ranks finish."""
    import json
    env = dict(os.environ, CP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(29450 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--size",
           "192", "--k", "6", "--reg", reg, "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    r = run_command(cmd, env=env, timeout=250)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["parallelism"] == f"row slabs x{world}"


def _run_worker(world, port, **env_extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **{k: str(v) for k, v in env_extra.items()})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "slab_worker.py")]
    r = run_command(cmd, env=env, timeout=300)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0


@pytest.mark.parametrize("world,reg,bc,fp32,maxiter,tol,expect_stop,chunk", [
    (2, "tv", "neumann", 0, 150, 1.0, 1, 32), (3, "tv", "neumann", 1, 40, 1e-3, 0, 32), (3, "tv", "neumann", 0, 150, 1.0, 1, 4),
    (2, "curvature", "dirichlet", 0, 70, 1e-3, 0, 32), (3, "curvature", "neumann", 1, 40, 1e-3, 0, 32)])
def test_slab_solve_matches_single_handle(world, reg, bc, fp32, maxiter, tol, expect_stop, chunk):
    """chambolle_pock on row slabs (MultiGpuEngine.solve): history row 0, stop rule, iteration counts and histories equal
    cp_solve on one handle; fp64 and fp32 storage; NN-TV (fused kernel) and NN-curvature (two field passes)"""
    _run_worker(world, 29470 + world + 4 * fp32, SLAB_MODE="solve", SLAB_REG=reg, SLAB_BC=bc, SLAB_FP32=fp32,
                SLAB_MAXITER=maxiter, SLAB_TOL=tol, SLAB_EXPECT_STOP=expect_stop, SLAB_CHUNK=chunk)


@pytest.mark.parametrize("world,reg", [(2, "tv"), (3, "curvature")])
def test_multilevel_driver_on_slabs(world, reg):
    """mf_nn_*_registration_no_ref_ml with the two fine levels of three on row slabs (replicated coarse level ->
    slab level -> slab level: both kinds of prolongation) against the single-GPU driver"""
    _run_worker(world, 29490 + world, SLAB_MODE="driver", SLAB_REG=reg)
