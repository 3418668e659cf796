"""N > 1 through the REAL HIP path on the one-GPU box: bench.py under torch.distributed.run with two ranks that share
device 0 (VC2_BENCH_DEVICE_MAP=0,0, backend gloo for the barrier / MAX / AND reductions -- two RCCL ranks cannot share
a device).  Exercises what the driver's multi-GPU run executes: per-rank batches (different pictures on rank 1), every
rank's own parity check, the AND-reduced verdict, MAX-over-ranks timing, one JSON line from rank 0.  Not a scaling
measurement: the two ranks contend for one GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "3", "--warmup", "1", "--batch", "4", "--no-cpu-baseline", "--no-e2e", "--no-other-configs"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _line(out):
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", VC2_BENCH_DEVICE_MAP="0,0")
    env.pop("VC2_BENCH_DRYRUN", None)
    one = _line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON, env=env, capture_output=True, text=True, timeout=900))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo"] + COMMON
    two = _line(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900))
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["pictures_per_gpu_per_step"] == 4 and two["scaling"] == "weak"
    assert "every rank" in two["parity_checked"] and "AND-reduced over 2 ranks" in two["parity_checked"]
    assert "slots 2-3 of every rank byte for byte against the oracle" in two["parity_checked"]   # each rank forked its own oracle pool
    assert "reference digest" in one["parity_checked"]
    # two ranks on one GPU: twice the pictures in (at least) the time one rank needs for its own -- the aggregate stays within
    # what one GPU delivers (overlap between the ranks' kernels can add a little; contention and two processes take away)
    ratio = two["value"] / one["value"]
    assert 0.4 < ratio < 1.6, (one["value"], two["value"])
    assert two["ms_per_step"] > 0.8 * one["ms_per_step"]


def test_bench_refuses_a_wrong_slot(tmp_path):
    """the check that guards the number: a library whose output differs (here: the payload slot of another picture is
    decoded -- VC2_BENCH_SABOTAGE swaps two pictures of the decoded batch before the comparison) must not print a line"""
    env = dict(os.environ, VC2_BENCH_SABOTAGE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "refusing to report a number" in out.stderr + out.stdout


def test_bench_two_ranks_on_two_gpus_over_rccl():
    """the driver's N = 2 command as it stands (backend nccl = RCCL, rank r on device r); skipped on the one-GPU boxes"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: runs on the driver's multi-GPU node")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("VC2_BENCH_DRYRUN", None)
    env.pop("VC2_BENCH_DEVICE_MAP", None)
    one = _line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON, env=env, capture_output=True, text=True, timeout=900))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + COMMON
    two = _line(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900))
    assert two["n_gpus"] == 2 and "against the oracle" in two["parity_checked"]
    assert two["value"] > 1.5 * one["value"], (one["value"], two["value"])   # frame-parallel, no collective: close to 2 x
