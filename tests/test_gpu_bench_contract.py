"""bench.py prints ONE JSON line with the fields the driver and the judge read (metric/value/unit, roofline,
cpu_baseline); checked on a small problem so the run takes seconds."""
import json
import os
import subprocess
import sys

import pytest
from conftest import run_command

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    r = run_command([sys.executable, os.path.join(ROOT, "bench.py")] + args, timeout=250, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_line_fields():
    out = _run(["--size", "256", "--k", "4", "--steps", "5", "--warmup", "2", "--cpu-size", "64", "--cpu-iters", "3",
                "--ref-size", "32", "--ref-iters", "2"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 5 and out["warmup"] == 2 and out["vs_baseline"] is None
    assert out["unit"] == "iterations/s" and out["value"] > 0 and out["dtype"] == "f64" and out["data"] == "synthetic"
    assert abs(out["value"] * out["ms_per_step"] - 1e3) < 1.0
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    # roofline.frac is physical: achieved = the 21 elements x 8 B per pixel-image k_mf_fused_tv must move / its measured launch
    # time; SURVEY 8(d)'s 37-element convention (which can exceed 1) sits under algorithmic_frac
    kms = out["config"]["kernel_ms"]["dual"]
    assert abs(rf["achieved"] - 21 * 8.0 * 4 * 256 * 256 / (kms * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert abs(rf["algorithmic_achieved"] - 37 * 8.0 * 4 * 256 * 256 / (kms * 1e-3) / 1e9) < 1e-6 * rf["algorithmic_achieved"]
    assert abs(rf["algorithmic_frac"] - rf["algorithmic_achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] <= 1.0
    # the kernel name comes from the library; no PMC measurement is committed for this small shape
    assert rf["kernel"].startswith("k_mf_fused_tv<double, false, false, 4,") and rf["traffic"] is None and rf["hbm_frac"] is None
    # default protocol: at least 5 regions and at least 200 timed iterations in all -> 40 regions of 5 steps here
    assert out["repeats"] == 40 and len(out["samples_ms_per_step"]) == 40 and out["handover_fallbacks"] == 0
    assert out["ms_per_step_min"] <= out["ms_per_step"] <= out["ms_per_step_max"]
    assert rf["measured_copy_GBps"]["16B_per_lane"] > 5500.0      # the one-element-per-thread copy (guide: 6.29 TB/s)
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "iterations/s"
    rs = cb["reference_shaped"]
    assert rs["cores"] == 1 and rs["value"] > 0 and "1 thread" in rs["sample"]
    assert "workload" in out["config"]


def test_bench_line_fp32_and_batch_modes():
    out = _run(["--size", "256", "--k", "4", "--steps", "5", "--warmup", "2", "--fp32", "--no-cpu-baseline"])
    assert "f32" in out["dtype"] and "fp32 storage" in out["metric"]
    out = _run(["--workload", "batch", "--size", "128", "--problems", "5", "--steps", "5", "--warmup", "2"])
    assert out["unit"] == "problem-iterations/s" and out["n_gpus"] == 1 and out["value"] > 0


def test_bench_self_launch_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts its two ranks itself (fresh child processes;
    the parent touches no GPU) and still prints exactly one JSON line. Both ranks share device 0 here
    (CP_BENCH_BACKEND=gloo: messages staged through the host)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(CP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = run_command([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "256", "--k", "4", "--steps", "4",
                     "--warmup", "2", "--repeats", "2", "--no-cpu-baseline"], timeout=250, env=env)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["repeats"] == 2 and out["value"] > 0
    assert out["config"]["parallelism"] == "row slabs x2" and out["roofline"]["traffic"] is None
