import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# ---- a process that starts the tests' child processes -------------------------------------------------------------------------
# Tests that launch ranks (torch.distributed.run, bench.py) used subprocess.run from the pytest process, i.e. a fork of a process that
# has initialised the GPU and carries the HIP runtime's threads; once in ~150 launches that fork died with a segmentation fault inside
# subprocess.Popen (round 3). The launcher below is started when conftest is imported -- before anything touches the GPU -- and runs the
# commands on request, so the GPU process never forks.
import atexit
import json
import subprocess
import types

_LAUNCHER_SRC = r"""
import json, subprocess, sys
for line in sys.stdin:
    req = json.loads(line)
    try:
        r = subprocess.run(req["cmd"], env=req["env"], capture_output=True, text=True, timeout=req["timeout"], cwd=req["cwd"])
        out = {"returncode": r.returncode, "stdout": r.stdout, "stderr": r.stderr, "timed_out": False}
    except subprocess.TimeoutExpired as e:
        txt = lambda b: b.decode(errors="replace") if isinstance(b, bytes) else (b or "")
        out = {"returncode": -9, "stdout": txt(e.stdout), "stderr": txt(e.stderr), "timed_out": True}
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()
"""
_launcher = subprocess.Popen([sys.executable, "-u", "-c", _LAUNCHER_SRC], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)


@atexit.register
def _stop_launcher():
    try:
        _launcher.stdin.close()
        _launcher.wait(timeout=10)
    except Exception:
        _launcher.kill()


def run_command(cmd, env=None, timeout=300, cwd=None):
    """subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout) through the launcher process; returns an object
    with returncode / stdout / stderr; raises subprocess.TimeoutExpired like subprocess.run"""
    req = {"cmd": list(cmd), "env": dict(os.environ if env is None else env), "timeout": timeout, "cwd": cwd}
    _launcher.stdin.write(json.dumps(req) + "\n")
    _launcher.stdin.flush()
    line = _launcher.stdout.readline()
    if not line:
        raise RuntimeError("the test launcher process died")
    out = json.loads(line)
    if out["timed_out"]:
        raise subprocess.TimeoutExpired(cmd, timeout, output=out["stdout"], stderr=out["stderr"])
    return types.SimpleNamespace(returncode=out["returncode"], stdout=out["stdout"], stderr=out["stderr"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """The PyTorch wheel bundles its own HIP runtime; libcpengine.so links the system one. Both coexist in one process
    only if torch initialises its device first (the other order makes torch report "No HIP GPUs are available"), so do
    that once per session, whatever order the tests run in. A no-op without a GPU."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
