"""bench.py's launcher contract under torch.distributed.run with 2 ranks on CPU (gloo, --dry): every --config shards its channels
8 / 4 / 1 per rank, issues its collective and prints ONE JSON line from rank 0 with the driver's schema.  The CPU baseline legs (1 thread,
all cores, line-faithful) are exercised on a tiny sample."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("config,per_gpu,nred", [("filtwelch", 1, 2049), ("stft", 8, 0), ("resample", 4, 1024)])
def test_bench_dry_two_ranks(config, per_gpu, nred):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", config, "--dry"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in out
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["channels_per_gpu"] == per_gpu and out["config"]["channels_total"] == 2 * per_gpu and out["config"]["allreduce_floats"] == nred
    assert out["config"]["collective_check"] == ("ok" if nred else "n/a (no collective on this path)")


@pytest.mark.parametrize("config,nch_total,per_gpu", [("stft", 64, 8), ("resample", 32, 4), ("filtwelch", 8, 1)])
def test_bench_dry_eight_ranks_own_the_baseline_channel_blocks(config, nch_total, per_gpu):
    """The shape the driver's 8-GPU run has (BASELINE configs 4 and 5: 64 channels -> 8 per GPU, 32 -> 4 per GPU), with eight gloo ranks on the CPU: rank r
    owns the contiguous block [r per_gpu, (r + 1) per_gpu) -- a pointer offset into the channel-major signal, no repacking -- and the one
    collective of the path agrees with an all_gather + Float64 sum on every rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--config", config, "--dry"], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["channels_total"] == nch_total and out["config"]["channels_per_gpu"] == per_gpu
    assert out["config"]["channel_blocks"] == [[r_ * per_gpu, (r_ + 1) * per_gpu] for r_ in range(8)]
    assert out["config"]["collective_check"] == ("n/a (no collective on this path)" if config == "stft" else "ok")


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2 --dry` with NO launcher around it (the form the driver's single-command BENCH uses) starts two ranks itself and
    reports n_gpus 2; the step's collective is cross-checked against an all_gather + Float64 sum (`collective_check`)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["channels_total"] == 2 and out["config"]["collective_check"] == "ok"


def test_bench_refuses_fewer_devices_than_asked():
    """Without --dry, --gpus 8 on a box with fewer than 8 devices exits non-zero and prints NO JSON line (never `n_gpus: 1` for --gpus 8)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("this box really has 8 devices")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "--gpus 8" in r.stderr


def test_bench_refuses_a_contradicting_launcher():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "contradicts" in r.stderr


def test_cpu_baseline_variants_small_sample():
    sys.path.insert(0, ROOT)
    import bench
    cb = bench.cpu_baseline(18)
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert cb["multi"]["cores"] == (os.cpu_count() or 1) and cb["multi"]["value"] > 0
    assert cb["faithful"]["value"] > 0 and "line-faithful" in cb["faithful"]["sample"]
