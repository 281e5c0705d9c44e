"""bench.py end to end on the GPU box: the JSON contract at N = 1, and the N > 1 code path
(two ranks sharing GPU 0 over gloo -- the driver's real multi-GPU run uses RCCL, covered by
the nccl test below when the box has two devices)."""

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
SMALL = ["--scale", "small", "--queries", "16384", "--rotate", "8", "--min-region-ms", "2", "--cpu-budget", "1.5", "--no-secondary",
         "--max-leg-s", "1.5", "--sustained-s", "1"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(stdout: str) -> dict:
    lines = [l for l in stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def _run(cmd, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=e, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return _last_json(r.stdout)


def test_bench_contract_single_gpu():
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", *SMALL])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "timing"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["dtype"] == "u8" and d["scaling"] == "weak"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    # (pipelined_frac counts SURVEY 8d's ALGORITHMIC bytes -- a byte per reference base, a window per haplotype --; the kernel reads half a
    # byte per base once per query, so on this test's small, cache-warm dataset the fraction can pass 1: hot_small reads 1.05)
    assert 0 < r["pipelined_frac"] < 1.25 and r["frac_of_copy_ceiling"] > r["frac"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and "1" in c["threads_sweep"] and c["cores"] >= 1
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    # the line checks what it timed: two batches of the last timed launch (one at an in-group position >= 11 when the rotation allows)
    v = d["verified"]
    assert v["batches"] >= 1 and v["mismatches"] == 0 and v["rows"] == 4096 * v["batches"]
    assert v["timed_launch_equal"] and not v["relaunched"]        # (the TIMED launch equalled the oracle, not a relaunch of its arguments)


def test_bench_secondary_legs():
    """The default run's short extra legs: ragged cfg3 rows (the reference's default output) and the cfg4 step, under
    `secondary` next to the unchanged headline (here on a small dataset and a small cfg4 grid)."""
    args = [a for a in SMALL if a != "--no-secondary"]
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--sustained-s", "0", *args],
             env={"GVL_CFG4_R": "2", "GVL_CFG4_S": "64"})
    sec = d["secondary"]
    rg, c4 = sec["ragged"], sec["cfg4"]
    assert "error" not in rg and "error" not in c4, sec
    assert rg["ms_per_step"] > 0 and 0 < rg["step_frac"] < 1 and rg["epochs_timed"] >= 2 and rg["batches_per_epoch"] == 8
    assert c4["ms_per_step"] > 0 and 0 < c4["step_frac"] < 1 and c4["kernel_ms"] > 0 and "131072" in c4["workload"]
    # ... and the same step on a dataset of 512 samples per region (the leg says where each dataset's inputs come from)
    cold = sec["cfg4_cold"]
    assert "error" not in cold, cold
    assert cold["ms_per_step"] > 0 and 0 < cold["step_frac"] < 1 and "512 samples" in cold["dataset"] and "64 samples" in c4["dataset"]
    assert c4["inputs"] and cold["inputs"] and cold["input_bytes"]["intervals"] > 4 * 12 * 100_000
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]          # the headline is cfg3's
    # the reference's other output modes + the reference-only fetch + training mode, each a driver-timed leg
    for name in ("onehot_cl", "annotated", "keep_mask", "reference", "random_shifts"):
        leg = sec[name]
        assert "error" not in leg, (name, leg)
        assert leg["ms_per_step"] > 0 and 0 < leg["step_frac"] < 1.3 and leg["algorithmic_bytes_per_step"] > 4096 * 2048 * 4, (name, leg)
    assert sec["random_shifts"]["deterministic_ms_per_step"] > 0


@pytest.mark.parametrize("extra", [["--gather"], ["--strong", "--gather"]], ids=["weak+gather", "strong+gather"])
def test_bench_two_ranks_sharing_one_gpu_gloo(extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "10", "--warmup", "2",
           "--no-cpu-baseline", *SMALL, *extra]
    d = _run(cmd, env={"GVL_BENCH_DEVICE": "0", "GVL_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["gather_ms"] > 0
    strong = "--strong" in extra
    assert d["scaling"] == ("strong" if strong else "weak")
    per_step = 4096 if strong else 8192
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["config"]["windows_per_step_per_rank"] == (2048 if strong else 4096)
    assert "world_size 2" in d["config"]["parallelism"]


def test_bench_two_gpus_rccl():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices (the driver's SCALE run covers RCCL)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--no-cpu-baseline", "--gather", *SMALL]
    d = _run(cmd)
    assert d["n_gpus"] == 2 and d["gather_ms"] > 0 and d["scaling"] == "weak"


def test_bench_refuses_world_size_mismatch():
    """Started under a launcher (WORLD_SIZE set) with another --gpus: refused."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", *SMALL], capture_output=True, text=True,
                       cwd=REPO, timeout=300, env={**os.environ, "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (how the driver runs --gpus 1): the parent starts torch.distributed.run as a
    child before it has touched the GPU, relays the line and exits with the child's code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"GVL_BENCH_DEVICE": "0", "GVL_BENCH_BACKEND": "gloo"})
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--gather", *SMALL],
                       capture_output=True, text=True, cwd=REPO, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["gather_ms"] > 0 and d["scaling"] == "weak"
    assert abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_stream_ptr_is_torchs_current_stream():
    """device._stream_ptr reads the current stream through torch's raw accessor: the same handle as torch.cuda.current_stream()'s, on
    the default stream and inside a `with torch.cuda.stream(...)`."""
    import torch

    from genvarloader_amd import device as gdev

    assert gdev._stream_ptr().value in (None, 0) or gdev._stream_ptr().value == torch.cuda.current_stream().cuda_stream
    assert (gdev._stream_ptr().value or 0) == torch.cuda.current_stream().cuda_stream
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        assert (gdev._stream_ptr().value or 0) == s.cuda_stream
        assert (gdev._stream_ptr(s).value or 0) == s.cuda_stream
    assert (gdev._stream_ptr().value or 0) == torch.cuda.current_stream().cuda_stream
