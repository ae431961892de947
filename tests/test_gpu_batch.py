"""Batch mode sharded over ranks (BASELINE config 5) on ONE GPU: 2 and 3 gloo ranks against the single-handle batch."""
import os
import subprocess
import sys

import pytest
from conftest import run_command

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,P,fp32", [(2, 7, "0"), (3, 8, "1"), (3, 2, "0")])
def test_batch_shards_match_single_handle(world, P, fp32):
    env = dict(os.environ, BATCH_P=str(P), BATCH_FP32=fp32, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29440 + world), os.path.join(ROOT, "tests", "batch_worker.py")]
    r = run_command(cmd, env=env, timeout=200)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0 and "batch shards == single handle" in r.stdout


def test_multi_image_batch_shards_match_the_drivers():
    """ma_amd.batch.multi_image_registration_batch: image sequences of different sizes (one with 18 images: the general
    K > 16 path) sharded over 2 gloo ranks, mf_nn and variance drivers, equal to the drivers called one by one (bitwise)"""
    env = dict(os.environ, BATCH_P="2", BATCH_FP32="0", BATCH_MULTI="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29446", os.path.join(ROOT, "tests", "batch_worker.py")]
    r = run_command(cmd, env=env, timeout=280)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0 and "multi-image batch == drivers one by one" in r.stdout


def test_bench_batch_workload_two_ranks():
    """bench.py --workload batch with 2 ranks (gloo on one GPU): one JSON line from rank 0, all problems counted"""
    import json
    env = dict(os.environ, CP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29447", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "batch", "--size", "96",
           "--problems", "9", "--steps", "6", "--warmup", "2"]
    r = run_command(cmd, env=env, timeout=250)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["unit"] == "problem-iterations/s" and out["value"] > 0
    assert "roofline" in out and out["roofline"]["bound"] == "hbm"
