#!/usr/bin/env python3
"""bench.py -- haplotype windows/sec + achieved HBM GB/s on MI355X.

A "step" is one pass of the hot path over one synthetic batch: ONE launch of the
fused reconstruct -> reverse-complement -> one-hot kernel through the C-ABI
(``gvl_reconstruct``) with every input already resident in HBM.  The default
workload is BASELINE.json ``configs[2]`` -- 4096 windows x 2048 bp, SNP+indel,
reverse-complement on half the rows, uint8 one-hot output -- which is the
configuration the metric is quoted on ("4096x2048bp SNP+indel one-hot");
``--workload cfg2`` gives the SNP-only ``configs[1]``.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches one rank per GPU (torch.distributed.run); rows are
independent, so each rank processes its own 4096-window batch (weak scaling, no
data-path collective) and ``value`` = windows all ranks processed / max-over-ranks time.
Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 measured copy ceiling


def algorithmic_bytes_per_window(L: int, mean_variants: float, haps: bool, onehot: bool) -> float:
    """L*(r + h + 4*o) + 28*V + 61 (r = 1 reference read)."""
    return L * (1 + (1 if haps else 0) + (4 if onehot else 0)) + 28.0 * mean_variants + 61.0


def cpu_baseline(st, bt, haps: bool, budget_s: float = 12.0) -> dict:
    """The oracle (C restatement of the reference's Rust/rayon path: reconstruct ->
    rc_flat_rows -> one-hot) timed on the host cores over the same batch."""
    from oracle import oracle

    try:  # the shipped .so is x86-64-v3; rebuild for this host's CPU when gcc is here
        oracle.build(march="native")
    except Exception:
        pass
    threads = oracle.default_threads()
    K = bt.n_windows
    L = bt.output_length
    out = np.empty(K * L, np.uint8)
    oh = np.empty((K * L, 4), np.uint8)
    oo = np.arange(K + 1, dtype=np.int64) * L
    go = np.ascontiguousarray(bt.geno_offsets)

    def run():
        oracle.reconstruct_haplotypes_from_sparse(
            out, oo, bt.regions, bt.shifts, bt.geno_offset_idx, go, bt.geno_v_idxs, st.v_starts,
            st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char,
            bt.keep, bt.keep_offsets, None, None, True, to_rc=bt.to_rc, onehot_out=oh,
            n_threads=threads)

    run()  # warm
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or time.perf_counter() < t_end:
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
        if len(times) >= 2000:
            break
    med = float(np.median(times))
    return {
        "value": K / med, "unit": "windows/s", "cores": threads, "kind": "port",
        "sample": f"full batch ({K} windows x {L} bp, reconstruct+RC+one-hot), "
                  f"{len(times)} iterations over {sum(times):.1f} s, median",
        "ms_per_batch": med * 1e3,
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="cfg3", choices=["cfg1", "cfg2", "cfg3", "cfg4"])
    ap.add_argument("--haps", action="store_true", help="also materialise haplotype bytes (h=1)")
    ap.add_argument("--contig", type=int, default=None, help="override reference contig length (bp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--slots", type=int, default=2, help="output ring slots")
    ap.add_argument("--streams", type=int, default=4,
                    help="batches kept in flight (one HIP stream each) in the timed region")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # GVL_BENCH_DEVICE / GVL_BENCH_BACKEND exist only so that the N > 1 code path can be smoke
    # tested on a 1-GPU box (several ranks sharing GPU 0 over gloo); the driver never sets them.
    dev_index = int(os.environ.get("GVL_BENCH_DEVICE", local_rank))
    backend = os.environ.get("GVL_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    from genvarloader_amd import HapsDevice, synth

    # ---- synthetic dataset + this rank's batch (rows shard across ranks) -------------
    cfg_idx = int(args.workload[3:])
    st, bt = synth.make_config(args.workload, seed=20260802 + cfg_idx + 1000 * rank, contig=args.contig)
    K, L = bt.n_windows, bt.output_length
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets,
                     geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char, device=f"cuda:{dev_index}")
    dbt = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=bt.to_rc)
    slots = [dev.alloc_output(dbt, K * L, haps=args.haps, onehot=True) for _ in range(max(1, args.slots))]
    stream = torch.cuda.current_stream()
    streams = [stream] + [torch.cuda.Stream() for _ in range(max(0, args.streams - 1))]
    while len(slots) < len(streams) + 1:      # an output slot per batch in flight (+1 being consumed)
        slots.append(dev.alloc_output(dbt, K * L, haps=args.haps, onehot=True))

    def step(i: int, pipelined: bool = True) -> None:
        # a batch is independent of the previous one: the loader keeps `--streams` batches in
        # flight on separate HIP streams, so the latency-bound head of one batch (parameter
        # and variant gathers, scans) overlaps the store-bound tail of another
        dev.launch(dbt, slots[i % len(slots)][1], streams[i % len(streams)] if pipelined else stream)

    def barrier() -> None:
        if dist is not None:
            dist.barrier()

    def join() -> None:
        for st_ in streams[1:]:
            stream.wait_stream(st_)

    # ---- warmup, then EXACTLY K timed steps ---------------------------------------------
    for i in range(args.warmup):
        step(i)
    join()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    join()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    torch.cuda.synchronize()
    wall = t1 - t0

    # ---- the kernel's own duration: same K launches back to back on ONE stream, HIP events on
    # that stream (this is what `rocprofv3 --kernel-trace --stats` reports per launch) -----------
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    for i in range(min(args.warmup, 10)):
        step(i, pipelined=False)
    torch.cuda.synchronize()
    ev0.record(stream)
    for i in range(args.steps):
        step(i, pipelined=False)
    ev1.record(stream)
    torch.cuda.synchronize()
    kern_ms = ev0.elapsed_time(ev1) / args.steps
    # ---- single-batch latency, host wall clock: launch -> synchronize -----------
    lat = []
    for i in range(50):
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        step(i, pipelined=False)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t_a)
    single_ms = float(np.median(lat)) * 1e3
    if dist is not None:
        tt = torch.tensor([wall, kern_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, kern_ms = float(tt[0]), float(tt[1])

    if rank == 0:
        abytes = algorithmic_bytes_per_window(L, bt.mean_variants, args.haps, True) * K
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tf = REPO / "profiles" / "traffic.json"
        if tf.exists():
            try:
                traffic = json.loads(tf.read_text()).get(f"{args.workload}{'+haps' if args.haps else ''}")
            except Exception:
                traffic = None
        res = {
            "metric": "haplotype windows/sec", "value": world * K * args.steps / wall, "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {K} windows x {L} bp per GPU, "
                            + ("SNP+indel, reverse-complement on half the rows" if cfg_idx >= 3 else "SNP-only")
                            + ", uint8 one-hot (K,L,4)" + (" + haplotype bytes" if args.haps else ""),
                "windows_per_batch": K, "length_bp": L, "ploidy": 2,
                "mean_variants_per_window": round(bt.mean_variants, 3),
                "reference_bp": int(st.ref.size), "parallelism": f"rows sharded over {world} GPU(s)",
                "batches_in_flight": len(streams),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "reconstruct_kernel<OH_LC, haps=%s, annot=false>" % ("true" if args.haps else "false"),
                "kernel_ms": kern_ms, "kernel_ms_how": "HIP events around K back-to-back launches on one stream",
                "algorithmic_bytes_per_launch": abytes,
                "pipelined_GBps": abytes * world / (wall / args.steps) / 1e9 / world,
                "single_batch_wall_ms": single_ms,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(st, bt, args.haps, args.cpu_budget)
        print(json.dumps(res), flush=True)

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
