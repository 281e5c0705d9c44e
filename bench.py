#!/usr/bin/env python3
"""bench.py -- haplotype windows/sec + achieved HBM GB/s on MI355X.

A "step" is one pass of the hot path over one synthetic batch of the fused reconstruct ->
reverse-complement -> one-hot kernel through the C-ABI, with every input already resident in HBM.
Steps are submitted the way the native loader submits them: in GROUPS (``--many``, default 16
batches) through ``gvl_reconstruct_many`` = ONE grid over the group (``recon_lean_rows_kernel``),
``--streams`` (3) launches in flight; ``--many 1`` is round 3's launch per batch.  The default workload is BASELINE.json
``configs[2]`` -- 4096 windows x 2048 bp, SNP+indel, reverse-complement on half the rows,
uint8 one-hot output -- the configuration the metric is quoted on ("4096x2048bp SNP+indel
one-hot"); ``--workload cfg2`` gives the SNP-only ``configs[1]``, ``--workload cfg4`` the
Enformer-length haplotypes + track realignment, ``--workload cfg1 --cpu-only`` the reference's
CPU-runnable plumbing case.

    python bench.py --gpus N --steps K --warmup W

Inputs are COLD by default: the dataset is genome scale (``--scale hg38``: 3.09 Gbp reference,
10 M variants, a sparse-genotype CSR + inline records > 1 GB, 8.4 M queries drawn across the
whole genome) and every step takes the next of ``--rotate`` (default 256) distinct batches, so
no step finds its reference windows, variant records or request arrays in L2 / Infinity Cache:
256 batches touch 690 MB of windows + slot lines, 2.7 x the 256 MB Infinity Cache.  (Until the
middle of round 3 the default was 64 = 172 MB, which FITS that cache: fine for round 2's kernel,
which was not bound by its reads -- 64 / 256 / 1024 measured alike then -- but the lean kernel
runs 8.1-8.4 us per batch at 256 and 6.3-6.6 at 64.  ``--rotate 64`` is kept as the
"Infinity-Cache-warm" regime.)  ``--scale small --rotate 1`` is round 1's cache-hot measurement
(64 Mbp contig, one batch).

Timing.  After W warmup steps:
  * contract region: barrier + synchronize, EXACTLY K steps, synchronize + barrier, host clock
    -> ``wall_ms_per_step`` (at small K this is mostly launch / synchronize latency);
  * the same K-step region is then repeated (each one again bracketed by barrier +
    synchronize) until >= ``--min-region-ms`` of GPU time has been sampled; inside a region
    the launches are queued behind a short gate kernel and HIP events on the work streams
    give the region's GPU time (first kernel start -> last kernel end).  ``ms_per_step`` =
    median region / K, ``value`` = windows of all ranks / that time (max over ranks per
    region).  This is what makes ``--steps 20`` and ``--steps 2000`` agree;
  * ``roofline``: K back-to-back launches on ONE stream between two HIP events, repeated the
    same way -> the kernel's own average duration (what ``rocprofv3 --kernel-trace --stats``
    reports) -> ``achieved`` = algorithmic bytes per launch / that;
  * ``sustained``: >= ``--sustained-s`` (6) seconds of back-to-back cold batches on the same
    schedule (``--streams`` launches of ``--many`` batches in flight) between ONE pair of HIP events (no gate, no per-region synchronisation):
    the rate at seconds, with the shader / memory clocks read from rocm-smi before and after.

For N > 1 the driver launches one rank per GPU (torch.distributed.run); rows are independent,
so each rank processes its own 4096-window batches (weak scaling, no data-path collective)
and ``value`` = windows all ranks processed / max-over-ranks time.  ``--strong`` splits ONE
4096-window batch into contiguous query blocks (``sharding.shard_batch``) instead;
``--gather`` additionally times the optional RCCL all-gather of the ranks' one-hot shards
(``gather_ms``; never inside ``value``).  Rank 0 prints ONE JSON line.
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

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0      # measured float4-copy ceiling (same guide)
HBM_STORE_GBS = 5650.0     # what the pool's boxes sustain store-only for outputs beyond the Infinity Cache (tools/wbench.hip: 5.4-5.8 TB/s,
                           # profiles/r06_wbench.txt): the ceiling of a kernel that mostly writes -- a one-hot is 4 bytes out per half byte read


def algorithmic_bytes_per_window(L: int, mean_variants: float, haps: bool, onehot: bool) -> float:
    """L*(r + h + 4*o) + 28*V + 61 (r = 1 reference read)."""
    return L * (1 + (1 if haps else 0) + (4 if onehot else 0)) + 28.0 * mean_variants + 61.0


def cpu_baseline(st, bt, budget_s: float = 12.0, thread_counts=None) -> dict:
    """The oracle (C restatement of the reference's Rust/rayon path: reconstruct ->
    rc_flat_rows -> one-hot, rows handed to a persistent worker pool) timed on the host cores
    over one full batch, at several thread counts (1 and nproc at least).
    The reference's gate ``should_parallelize(total_bytes)`` (_threads.py:24,122-127) is
    ``total_bytes >= num_threads() * 1 MiB`` with num_threads() = GVL_NUM_THREADS or the host's CPUs
    (cgroup-aware): this batch's haplotype bytes (K x L = 8 MiB for cfg3) run in PARALLEL only with at most 8
    threads (or GVL_FORCE_PARALLEL=1) -- on a many-core host the reference's DEFAULT is the serial path.
    ``reference_default`` says which of the sweep's rows that is; ``value`` stays the best of the sweep
    (= the reference with GVL_FORCE_PARALLEL=1 and the best GVL_NUM_THREADS)."""
    from oracle import oracle

    try:      # a -march=native copy in a scratch dir; the tree's x86-64-v3 .so stays as it is
        lib = oracle.build_native()
        march = "native"
    except Exception:
        lib, march = None, "x86-64-v3"
    nproc = oracle.default_threads()
    if thread_counts is None:
        thread_counts = sorted({1, 8, 32, 64, 128, nproc} & set(range(1, nproc + 1)) | {1, nproc})
    K, L = bt.n_windows, bt.output_length
    out = np.empty(K * L, np.uint8)
    oh = np.empty((K * L, 4), np.uint8)
    oo = np.arange(K + 1, dtype=np.int64) * L
    call = oracle.BatchCall(out, oo, bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs,
                            st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets,
                            st.pad_char, to_rc=bt.to_rc, onehot_out=oh, library=lib)
    sweep = {}
    per = budget_s / max(len(thread_counts), 1)
    total_iters, total_s = 0, 0.0
    for nt in thread_counts:
        call.run(nt)  # warm (and sizes the pool)
        times = []
        t_end = time.perf_counter() + per
        while len(times) < 3 or time.perf_counter() < t_end:
            t0 = time.perf_counter()
            call.run(nt)
            times.append(time.perf_counter() - t0)
            if len(times) >= 2000:
                break
        med = float(np.median(times))
        sweep[str(nt)] = {"windows_per_s": K / med, "ms_per_batch": med * 1e3, "iterations": len(times),
                          "hap_GBps": K * L / med / 1e9}
        total_iters += len(times)
        total_s += sum(times)
    best = max(sweep, key=lambda k: sweep[k]["windows_per_s"])
    gate_bytes = K * L                                # what _haps.py:865 hands to should_parallelize
    gate_on = gate_bytes >= nproc * (1 << 20)         # with the reference's default thread count = this host's CPUs
    ref_default = sweep[str(nproc)] if gate_on else sweep["1"]
    return {
        "value": sweep[best]["windows_per_s"], "unit": "windows/s", "cores": int(best), "kind": "port",
        "sample": f"one full batch ({K} windows x {L} bp, reconstruct+RC+one-hot) per iteration, "
                  f"{total_iters} iterations over {total_s:.1f} s, median per thread count; best of the sweep",
        "ms_per_batch": sweep[best]["ms_per_batch"], "host_cores": nproc, "march": march,
        "single_thread_windows_per_s": sweep["1"]["windows_per_s"],
        "single_thread_hap_GBps": sweep["1"]["hap_GBps"],
        "threads_sweep": sweep,
        "value_is": "best of the thread sweep = the reference with GVL_FORCE_PARALLEL=1 (or GVL_NUM_THREADS <= batch MiB) and the best thread count",
        "parallel_gate": f"reference: parallel iff haplotype bytes >= threads x 1 MiB (_threads.py:24,127): {gate_bytes / (1 << 20):.0f} MiB here, "
                         f"so parallel only with GVL_NUM_THREADS <= {max(1, gate_bytes >> 20)} or GVL_FORCE_PARALLEL=1",
        "reference_default": {
            "threads": nproc if gate_on else 1, "gate": "on" if gate_on else "off",
            "windows_per_s": ref_default["windows_per_s"], "ms_per_batch": ref_default["ms_per_batch"],
            "what": f"what the reference does on this host WITHOUT environment overrides: num_threads() = {nproc} CPUs, gate "
                    + ("on: the rayon path at that width" if gate_on else "off: the serial path (the sweep's 1-thread row)"),
        },
    }


def cpu_only(args) -> None:
    """BASELINE.json configs[0]: 1024 windows x 1024 bp, SNP-only, CPU path only (no GPU)."""
    from genvarloader_amd import synth

    st, bt = synth.make_config(args.workload)
    res = cpu_baseline(st, bt, args.cpu_budget)
    K, L = bt.n_windows, bt.output_length
    print(json.dumps({
        "metric": "haplotype windows/sec", "value": res["value"], "unit": "windows/s", "n_gpus": 0,
        "steps": None, "warmup": None, "ms_per_step": res["ms_per_batch"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {K} windows x {L} bp, CPU oracle only (the reference's "
                               "Rust/rayon path restated in C; plumbing case, no GPU)"},
        "cpu_baseline": res}), flush=True)


def gpu_clocks(dev_index: int = 0):
    """Shader / memory clock of the device as rocm-smi reports them right now ({} when it cannot be read)."""
    import re
    import subprocess

    try:
        out = subprocess.run(["rocm-smi", "-d", str(dev_index), "--showclocks"], capture_output=True, text=True, timeout=20).stdout
    except Exception:
        return {}
    res = {}
    for key in ("sclk", "mclk", "fclk"):
        m = re.search(key + r"\s+clock level:?\s*\S*:?\s*\(?(\d+)\s*Mhz", out, re.I)
        if m:
            res[key + "_mhz"] = int(m.group(1))
    return res


class Timer:
    """Repeated K-step regions, each bracketed by barrier + synchronize; GPU time of a region from
    HIP events on the work streams, the launches queued behind a gate kernel."""

    def __init__(self, torch, dist, backend, streams, min_region_ms, max_regions, max_wall_s=8.0):
        self.torch, self.dist, self.backend = torch, dist, backend
        self.streams = streams
        self.min_ms, self.max_regions, self.max_wall_s = float(min_region_ms), int(max_regions), float(max_wall_s)
        self.gate = torch.cuda.Stream()
        # calibrate torch.cuda._sleep (spins on the shader clock)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(100_000)
        torch.cuda.synchronize()
        e0.record(); torch.cuda._sleep(2_000_000); e1.record(); torch.cuda.synchronize()
        self.cycles_per_us = 2_000_000 / max(e0.elapsed_time(e1) * 1e3, 1.0)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def allmax(self, vals):
        if self.dist is None:
            return list(vals)
        torch = self.torch
        t = torch.tensor(list(vals), dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t.cpu()]

    def region(self, step, K, streams):
        """One bracketed region of exactly K steps -> (GPU ms, host wall ms)."""
        torch = self.torch
        ev_g = torch.cuda.Event()
        e0 = [torch.cuda.Event(enable_timing=True) for _ in streams]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
        self.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gate_us = 30.0 + 5.0 * min(K, 64)
        with torch.cuda.stream(self.gate):
            torch.cuda._sleep(int(gate_us * self.cycles_per_us))
            ev_g.record(self.gate)
        for s, e in zip(streams, e0):
            s.wait_event(ev_g)
            e.record(s)
        for i in range(K):
            step(i)
        getattr(step, "flush", lambda: None)()
        for s, e in zip(streams, e1):
            e.record(s)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        self.barrier()
        used = min(-(-K // getattr(step, "group", 1)), len(streams))
        gpu_ms = max(a.elapsed_time(b) for a in e0[:used] for b in e1[:used])
        return gpu_ms, (t1 - t0) * 1e3

    def measure(self, step, K, streams):
        """Repeat the region until >= min_ms of GPU time is sampled (or max_wall_s of host time is
        spent) -> (median GPU ms of a region, max over ranks per region; number of regions)."""
        first, w = self.region(step, K, streams)
        first, w = self.allmax([first, w])
        n = int(min(self.max_regions, max(5, np.ceil(self.min_ms / max(first, 1e-6)))))
        n = int(max(5, min(n, self.max_wall_s * 1e3 / max(w, 1e-3))))
        spans = [self.region(step, K, streams)[0] for _ in range(n)]
        spans = self.allmax(spans)
        return float(np.median(spans)), n, spans

    def sustained(self, calls, fn, streams, seconds, est_ms_per_step, final=None):
        """>= `seconds` of back-to-back steps on the streams' round-robin schedule between ONE event pair
        (recorded on streams[0]; the other streams are joined into it at the end).  `calls[i]` = the ctypes
        arguments of step i (cycled).  -> (GPU ms per step, steps, host seconds spent enqueueing)."""
        torch = self.torch
        n = int(max(len(calls), seconds * 1e3 / max(est_ms_per_step, 1e-6)))
        n -= n % len(streams)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        joins = [torch.cuda.Event() for _ in streams[1:]]
        self.barrier()
        torch.cuda.synchronize()
        start = torch.cuda.Event()
        start.record(streams[0])
        for s_ in streams[1:]:
            s_.wait_event(start)
        e0.record(streams[0])
        m = len(calls)
        t0 = time.perf_counter()
        bad = 0
        burst = min(n, 512)                    # the queue is empty here: these enqueues show the host's own cost per launch
        for i in range(burst):
            bad |= fn(*calls[i % m])
        t_burst = time.perf_counter() - t0
        marks = {n // 3: torch.cuda.Event(enable_timing=True), 2 * n // 3: torch.cuda.Event(enable_timing=True)}
        for i in range(burst, n):
            # (`final`: the leg's last launch with its own output set -- the same work, written where no other launch writes)
            bad |= fn(*(final(i % m) if final is not None and i == n - 1 else calls[i % m]))
            if i in marks:                    # (two more events on streams[0]: the rate of each third of the leg)
                marks[i].record(streams[0])
        t_host = time.perf_counter() - t0
        for s_, ev in zip(streams[1:], joins):
            ev.record(s_)
            streams[0].wait_event(ev)
        e1.record(streams[0])
        torch.cuda.synchronize()
        self.barrier()
        if bad:
            raise RuntimeError("gvl_reconstruct failed inside the sustained leg")
        ms = self.allmax([e0.elapsed_time(e1)])[0]
        pts = [(0, 0.0)] + sorted((i, e0.elapsed_time(e)) for i, e in marks.items() if i >= burst) + [(n, e0.elapsed_time(e1))]
        self.last_thirds = [(pts[k][1] - pts[k - 1][1]) / max(pts[k][0] - pts[k - 1][0], 1) for k in range(1, len(pts))]
        return ms / n, n, t_host, t_burst / burst


def secondary_ragged(torch, dev, ds, budget_s: float = 2.5) -> dict:
    """cfg3 with RAGGED rows (output_length = -1: the reference's default ``ds[r, s]`` shape, _haps.py:794-811,
    src/ffi/mod.rs:794-815) on the same cold genome-scale dataset, FROM DATASET INDICES through the native loader
    (``DeviceHapsDataset(output_length=-1).to_dataloader``: request prep once per epoch; per group of 16 batches one sizing --
    query-mode length deltas -> row lengths -> offsets, on the device -- and ONE grid of the lean kernel's pipelined form
    that reads those offsets; no host round trip).  Every window of the dataset is read once per epoch: cold.
    -> us per 4096-window batch over chained epochs."""
    from genvarloader_amd.loader import DeviceHapsDataset

    P, bs = ds.ploidy, 2048
    hds = DeviceHapsDataset(dev, ds.full_regions.cpu().numpy(), 1, P, output_length=-1, deterministic=True, seed=1)
    dl = hds.to_dataloader(batch_size=bs, shuffle=True, in_flight=3, group=16)
    n_b, tot_rows, tot_len, mx = 0, 0, 0, 0
    sample = []
    for batch in dl:                                   # a first epoch: warms up, and samples the row lengths
        n_b += 1
        if n_b <= 4:
            sample.append(batch.sizes.clone())
            tot_rows += int(batch.idx.numel()) * P
    torch.cuda.synchronize()
    for sz in sample:
        t, m = (int(v) for v in sz.cpu().tolist())
        tot_len, mx = tot_len + t, max(mx, m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_ep = 0
    while time.perf_counter() - t0 < budget_s or n_ep < 2:
        for batch in dl:
            pass
        n_ep += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = dt / (n_ep * n_b) * 1e3
    K = bs * P
    mean_len = tot_len / max(tot_rows, 1)
    mean_v = float((dev.geno_offsets[1] - dev.geno_offsets[0])[:1 << 20].double().mean())
    abytes = (mean_len * 5 + 28.0 * mean_v + 61.0) * K
    return {
        "workload": f"cfg3 ragged: {K} windows per batch, output_length = -1 (row = region + its haplotype's length delta, mean {mean_len:.1f} bases, "
                    f"longest seen {mx}), one-hot (total, 4), from dataset indices through the native loader (in_flight 3, groups of 16), "
                    f"{n_b} batches per epoch, every window read once per epoch",
        "ms_per_step": ms, "windows_per_s": K / (ms * 1e-3), "epochs_timed": n_ep, "batches_per_epoch": n_b,
        "algorithmic_bytes_per_step": abytes, "step_frac": abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "kernel": "recon_lean_rows_kernel<onehot, ragged> (one grid per group of 16 batches) behind hap_lengths_group_kernel + hap_scan_group_kernel",
        "how": "host clock over chained epochs (no synchronisation between them), steady state",
    }


def secondary_svar2(torch, G: int = 16, BQ: int = 2048, n: int = 40) -> dict:
    """The SVAR2 two-source provider on config 3's shape: gvl_svar2_merge over a GROUP of G batches' decoded channels
    (one launch over G x 4096 haplotypes) + gvl_reconstruct_many over the merged table (one grid), next to the same haplotypes
    through the SVAR1 table; HIP events on the launch streams, channels and outputs resident in HBM.  `pipelined`: the schedule a
    loader would run -- the next group's merge on a side stream under this group's reconstruction (two workspaces)."""
    import ctypes as C

    from genvarloader_amd import HapsDevice, _lib, svar2, synth
    from genvarloader_amd.device import _stream_ptr

    L, P = 2048, 2
    rng = np.random.default_rng(20260806)
    st = synth.make_static(rng, (16 << 20,), indel_frac=0.15)
    bt = synth.make_batch(rng, st, G * BQ, P, L, rc_frac=0.5)
    sv = synth.to_svar2(rng, st, bt, dense_af=0.3)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
    ch = svar2.Svar2Channels(*sv.args())
    lib = _lib.load()
    K = BQ * P
    tabs = [svar2.merge(dev, ch, bt.regions, P) for _ in range(2)]
    torch.cuda.synchronize()
    _lib.check_async()
    reg = tabs[0].regions
    sh = torch.zeros((G * BQ, P), dtype=torch.int32, device="cuda")
    rc = torch.from_numpy(bt.to_rc.astype(np.uint8)).cuda()
    goi1 = torch.from_numpy(bt.geno_offset_idx).cuda()
    outs = [torch.empty((K * L, 4), dtype=torch.uint8, device="cuda") for _ in range(G)]

    def group(goi):
        bs, os_ = (_lib.GvlBatch * G)(), (_lib.GvlOut * G)()
        for i in range(G):
            bs[i] = _lib.GvlBatch(regions=reg[i * BQ:].data_ptr(), regions_stride=4, shifts=sh[i * BQ:].data_ptr(),
                                  geno_offset_idx=goi[i * BQ:].data_ptr(), batch=BQ, ploidy=P, to_rc=rc[i * K:].data_ptr(),
                                  output_length=L, max_row_len=L)
            os_[i] = _lib.GvlOut(onehot=outs[i].data_ptr(), onehot_layout=0)
        return bs, os_

    b1, o1 = group(goi1)
    b2 = [group(t.geno_offset_idx)[0] for t in tabs]
    nbytes = int(lib.gvl_svar2_workspace_bytes(G * BQ, P, ch.c.n_vk, ch.c.dense_present_bits, ch.c.alt_len))
    merged, goi_p = _lib.GvlStatic(), C.c_void_p()

    def merge_into(i, sp):
        w_ = tabs[i].workspace.data_ptr() + (-tabs[i].workspace.data_ptr()) % 256
        _lib.check(lib.gvl_svar2_merge(C.byref(dev.c), C.byref(ch.c), C.c_void_p(reg.data_ptr()), C.c_int64(4), C.c_int64(G * BQ), C.c_int64(P),
                                       C.c_void_p(w_), C.c_int64(nbytes), C.byref(merged), C.byref(goi_p), sp))

    def recon(static_c, bs, sp):
        _lib.check(lib.gvl_reconstruct_many(C.byref(static_c), bs, o1, C.c_int32(G), sp))

    def timeit(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    sp0 = _stream_ptr()
    t_m = timeit(lambda: merge_into(0, sp0))
    t_2 = timeit(lambda: recon(tabs[0].c, b2[0], sp0))
    oh2 = outs[G - 1].clone()
    t_1 = timeit(lambda: recon(dev.c, b1, sp0))
    same = bool((oh2 == outs[G - 1]).all())
    t_b = timeit(lambda: (merge_into(0, sp0), recon(tabs[0].c, b2[0], sp0)))
    side, main = torch.cuda.Stream(), torch.cuda.current_stream()
    ev_m, ev_r = [torch.cuda.Event(), torch.cuda.Event()], [torch.cuda.Event(), torch.cuda.Event()]
    sps = C.c_void_p(side.cuda_stream)

    def pipelined(nn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        side.wait_stream(main)
        merge_into(0, sps)
        ev_m[0].record(side)
        for i in range(nn):
            cur, nxt = i % 2, (i + 1) % 2
            if i + 1 < nn:
                if i >= 1:
                    side.wait_event(ev_r[nxt])           # (the table about to be rewritten has been read)
                merge_into(nxt, sps)
                ev_m[nxt].record(side)
            main.wait_event(ev_m[cur])
            recon(tabs[cur].c, b2[cur], sp0)
            ev_r[cur].record(main)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / nn * 1e3

    pipelined(10)
    t_p = pipelined(n)
    _lib.check_async()
    n_rec = len(sv.vk_pos) + int(np.unpackbits(sv.dense_present, bitorder="little").sum())
    alg = G * K * (L * 5 + 61) + 28 * n_rec                        # L (r + 4 o) + 28 V + 61 per window, one-hot only
    ch_bytes = sum(int(x.numel()) * x.element_size() for x in (ch.vk_pos, ch.vk_ilen, ch.vk_alt_off, ch.vk_off, ch.dense_pos, ch.dense_ilen,
                                                                ch.dense_alt_off, ch.dense_range, ch.dense_present, ch.dense_present_off,
                                                                ch.alt_bytes))
    frac = lambda t_us: alg / t_us / 1e6 / 8.0          # noqa: E731  (bytes / us = MB/s; / 1e6 = TB/s; / 8 TB/s)
    return {"workload": f"{G} x ({BQ * P} x {L}) SNP+indel + RC, one-hot, as DECODED SVAR2 channels (var_key {len(sv.vk_pos)} entries, dense "
                        f"{len(sv.dense_pos)}, {int(sv.dense_present_off[-1])} presence bits, {n_rec} merged records)",
            "merge_us_per_batch": t_m / G, "merge_launch_us": t_m, "channel_MB": ch_bytes / 1e6, "workspace_MB": nbytes / 1e6,
            "reconstruct_merged_us_per_batch": t_2 / G, "reconstruct_svar1_us_per_batch": t_1 / G,
            "ms_per_step": t_p / G * 1e-3, "step_frac": frac(t_p), "back_to_back_us_per_batch": t_b / G, "back_to_back_frac": frac(t_b),
            "svar1_frac": frac(t_1), "equal_to_svar1_route": same,
            "how": "ms_per_step: group g + 1's gvl_svar2_merge on a side stream under group g's gvl_reconstruct_many (two workspaces, events); "
                   "the same launches back to back on one stream: back_to_back; channels resident in HBM (warm: one group's channels, "
                   f"{ch_bytes / 1e6:.1f} MB)"}



def secondary_long_modes(torch, n: int = 12) -> dict:
    """Rows of 131 072 bases (BASELINE config 4's 256-window batches) in the reference's other output modes, which until round 6 ran the
    all-purpose kernel: annotated haplotypes (bytes + 2 x i32 per base, src/ffi/mod.rs:2237-2397) and rows under the exonic keep mask
    (src/genotypes/mod.rs:127-176 -> the spliced path's filter), one gvl_reconstruct launch per batch, `n` launches back to back on one
    stream between HIP events, rotating over the dataset's 8 batches (inputs resident, like secondary.cfg4).  `r05_routing_*`: the same
    launches under GVL_DBG 1073741824 (round 5's routing = the all-purpose kernel), for A/B on the same box."""
    import bench_cfg4
    from genvarloader_amd import _lib

    R, S, P, L, bs = 16, 64, 2, 131072, 128
    st, dev, ds, tracks, mean_v = bench_cfg4.build("cuda:0", R, S, P, L)
    lib = _lib.load()
    K = bs * P
    order = np.random.default_rng(1).permutation(R * S)
    reqs = [ds.request(order[i:i + bs].astype(np.int64)) for i in range(0, len(order), bs)]

    def timeit(bts, out_c, flags):
        lib.gvl_set_debug_flags(flags)
        try:
            for i in range(3):
                dev.launch(bts[i % len(bts)], out_c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                dev.launch(bts[i % len(bts)], out_c)
            e1.record()
            torch.cuda.synchronize()
            _lib.check_async()
            return e0.elapsed_time(e1) / n
        finally:
            lib.gvl_set_debug_flags(-1)

    res = {}
    # annotated: bytes + annot_v_idxs + annot_ref_pos
    # (with the rows' chunk plans, as the native loader brings them per epoch: one walk per row, ahead of the launches)
    bts = [dev.prepare_batch(r[1], r[2], r[3], L, to_rc=r[4]) for r in reqs]
    bts = [dev.prepare_batch(r[1], r[2], r[3], L, to_rc=r[4], hap_plan=dev.hap_plan(b)) for r, b in zip(reqs, bts)]
    out, out_c = dev.alloc_output(bts[0], K * L, haps=True, onehot=False, annotate=True)
    ab = (L * (1 + 1 + 8) + 28.0 * mean_v + 61.0) * K
    ms, ms5 = timeit(bts, out_c, -1), timeit(bts, out_c, 1073741824)
    res["annotated_long"] = {"workload": f"{K} windows x {L} bp, annotated haplotypes (bytes + 2 x i32 per base), SNP+indel + RC",
                             "ms_per_step": ms, "windows_per_s": K / (ms * 1e-3), "algorithmic_bytes_per_step": ab,
                             "step_frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": "recon_lean_kernel<haps, long, annotated>",
                             "r05_routing_ms": ms5, "r05_routing_frac": ab / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "how": f"{n} gvl_reconstruct launches back to back on one stream, HIP events; 8 rotating batches (inputs resident)"}
    del out, out_c
    torch.cuda.empty_cache()
    # under the exonic keep mask: one-hot + bytes
    btk = []
    for r in reqs:
        kp, ko = dev.choose_exonic_variants(r[1][:, 1].contiguous(), r[1][:, 2].contiguous(), r[3])
        b0 = dev.prepare_batch(r[1], r[2], r[3], L, keep=kp, keep_offsets=ko, to_rc=r[4])
        btk.append(dev.prepare_batch(r[1], r[2], r[3], L, keep=kp, keep_offsets=ko, to_rc=r[4], hap_plan=dev.hap_plan(b0)))
    out, out_c = dev.alloc_output(btk[0], K * L, haps=True, onehot=True)
    ab = (L * 6 + 29.0 * mean_v + 61.0) * K
    ms, ms5 = timeit(btk, out_c, -1), timeit(btk, out_c, 1073741824)
    res["keep_mask_long"] = {"workload": f"{K} windows x {L} bp under the exonic keep mask (choose_exonic_variants on the rows' own regions), one-hot (K, L, 4) + bytes",
                             "ms_per_step": ms, "windows_per_s": K / (ms * 1e-3), "algorithmic_bytes_per_step": ab,
                             "step_frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": "recon_lean_kernel<onehot, haps, long> with a keep mask",
                             "r05_routing_ms": ms5, "r05_routing_frac": ab / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "how": f"{n} gvl_reconstruct launches back to back on one stream, HIP events; 8 rotating batches (inputs resident)"}
    return res



def secondary_spliced(torch, n_tx: int = 1500, S: int = 8, pairs: int = 256, n: int = 20, max_exon: int = 9000, exonic: bool = True) -> dict:
    """Spliced haplotypes under the exonic keep mask (Dataset with a splice map + `filter_exonic`: _dataset/_query.py:207-313,
    src/genotypes/mod.rs:127-176, src/ffi/mod.rs:1981-2076): transcripts of 3-14 exons, exon lengths log-normal (median 160 bases,
    one in thirty longer than the pipelined kernel's 2560), batches of `pairs` (transcript, sample) pairs through
    DeviceSplicedHapsDataset.  `ms_per_step`: batches through the dataset object (Python submit loop: request prep, lengths, plan, keep
    mask, one launch -- host clock, synchronised at the end); `kernel_ms`: that batch's ONE reconstruct launch on its own (HIP events),
    next to the same launch under GVL_DBG 1073741824 (round 5's routing: the all-purpose kernel)."""
    from genvarloader_amd import HapsDevice, _lib, synth
    from genvarloader_amd.loader import DeviceSplicedHapsDataset

    rng = np.random.default_rng(20260807)
    st = synth.make_static(rng, (32 << 20,), indel_frac=0.15)
    P = 2
    n_ex = rng.integers(3, 15, n_tx)
    ex_len = np.clip(np.exp(rng.normal(np.log(160.0), 0.95, int(n_ex.sum()))), 30, max_exon).astype(np.int64)
    intron = rng.integers(200, 3000, len(ex_len))
    so = np.concatenate([[0], np.cumsum(n_ex)]).astype(np.int64)
    starts = np.zeros(len(ex_len), np.int64)
    strand = np.zeros(len(ex_len), np.int64)
    for t in range(n_tx):
        a, b = so[t], so[t + 1]
        span = int((ex_len[a:b] + intron[a:b]).sum())
        t0 = int(rng.integers(1000, (32 << 20) - span - 1000))
        pos = t0 + np.concatenate([[0], np.cumsum(ex_len[a:b] + intron[a:b])[:-1]])
        starts[a:b] = pos
        strand[a:b] = 1 if rng.random() < 0.5 else -1
    regions = np.stack([np.zeros(len(ex_len), np.int64), starts, starts + ex_len, strand], 1).astype(np.int32)
    R = len(regions)
    go, gv = synth.sample_genotypes(rng, st, np.repeat(regions[:, 0], S), np.repeat(regions[:, 1], S), np.repeat(regions[:, 2], S), P)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv, pad_char=st.pad_char)
    ds = DeviceSplicedHapsDataset(dev, regions, S, P, splice_offsets=so, splice_region_idx=np.arange(R), onehot=True, haps=True, exonic=exonic)
    dl = ds.to_dataloader(batch_size=pairs, shuffle=True, seed=3)
    it = iter(dl)
    b = None
    for _ in range(3):
        b = next(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot_bases = 0
    for _ in range(n):
        try:
            b = next(it)
        except StopIteration:
            it = iter(dl)
            b = next(it)
        tot_bases += int(b.haps.numel())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    _lib.check_async()
    # the last batch's launch on its own
    bt = b._keep._keepalive
    oo = bt.out_offsets.cpu().numpy()
    lens = np.diff(oo)
    oc = _lib.GvlOut(haps=b.haps.data_ptr(), onehot=b.onehot.data_ptr(), onehot_layout=_lib.GVL_ONEHOT_LC)
    goi = bt.geno_offset_idx.reshape(-1)
    nv = float((dev.geno_offsets[1][goi] - dev.geno_offsets[0][goi]).double().mean())
    ab = float(lens.sum()) * 6 + len(lens) * (29.0 * nv + 61.0)
    lib = _lib.load()

    def kern(flags):
        lib.gvl_set_debug_flags(flags)
        try:
            for _ in range(3):
                dev.launch(bt, oc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                dev.launch(bt, oc)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        finally:
            lib.gvl_set_debug_flags(-1)

    k_ms, k5_ms = kern(-1), kern(1073741824)
    n_long = int((lens > 2560).sum())
    k_solo = kern(256) if n_long and len(lens) >= 16384 else None      # (the pipelined kernel with its long rows at the waves' ends, as before the front workgroups)
    extra = {} if k_solo is None else {"long_rows_at_the_waves_ends_kernel_ms": k_solo,
                                       "long_rows": "by the launch's front workgroups, chunks in parallel (GVL_DBG 256: by the wave that meets them, chunk after chunk)"}
    return {**extra, "workload": f"spliced haplotypes under the exonic keep mask: {pairs} (transcript, sample) pairs per batch = {len(lens)} exon rows "
                        f"(mean {lens.mean():.0f} bases, {int((lens > 2560).sum())} longer than 2560, longest {int(lens.max())}), one-hot + bytes",
            "ms_per_step": ms, "rows_per_s": len(lens) / (ms * 1e-3), "algorithmic_bytes_per_step": ab,
            "step_frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "kernel_ms": k_ms, "kernel_frac": ab / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "r05_routing_kernel_ms": k5_ms, "r05_routing_kernel_frac": ab / (k5_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "kernel": ("recon_lean_rows_kernel<onehot, haps, ragged, keep mask>; rows of a few hundred bases are bound by the scalar unit (1 250 rows/us "
                       "whatever their length, DESIGN 4.1), not by bytes" if len(lens) >= 16384 else
                       "a batch of a few thousand short rows + a few long ones: the all-purpose kernel (launch-latency-bound: 7 MB of work); from 16 384 "
                       "rows on recon_lean_rows_kernel<onehot, haps, ragged, keep mask> (the `spliced_large` leg)"),
            "how": "ms_per_step: batches through DeviceSplicedHapsDataset.to_dataloader (a Python submit loop with one host read per batch: "
                   "the output's size; ~30 small device launches), host clock; kernel_ms: the batch's one gvl_reconstruct launch, HIP events"}



def secondary_random_shifts(torch, dev, ds, budget_s: float = 2.5) -> dict:
    """cfg3 in TRAINING mode (SURVEY 8d's cfg3 variant; _haps.py:678-768, _query.py:160-187): ``deterministic=False`` -- every haplotype's
    shift drawn from U[0, max_shift], max_shift from its query-mode length delta -- and ``jitter=16``, fixed-length one-hot rows, from
    dataset indices through the native loader (request prep incl. the diffs pass once per epoch, groups of 16 batches = one grid),
    next to the same loader with ``deterministic=True`` on the same box.  -> us per 4096-window batch over chained epochs."""
    from genvarloader_amd.loader import DeviceHapsDataset

    P, bs, L = ds.ploidy, 2048, ds.length
    out = {}
    for name, kw in (("deterministic", dict(deterministic=True, jitter=0)), ("random", dict(deterministic=False, jitter=16))):
        hds = DeviceHapsDataset(dev, ds.full_regions.cpu().numpy(), 1, P, output_length=L, seed=1, **kw)
        dl = hds.to_dataloader(batch_size=bs, shuffle=True, in_flight=3, group=16)
        n_b = 0
        for batch in dl:
            n_b += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_ep = 0
        while time.perf_counter() - t0 < budget_s / 2 or n_ep < 2:
            for batch in dl:
                pass
            n_ep += 1
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / (n_ep * n_b) * 1e3, n_ep, n_b
        del dl, hds
    K = bs * P
    mean_v = float((dev.geno_offsets[1] - dev.geno_offsets[0])[:1 << 20].double().mean())
    abytes = algorithmic_bytes_per_window(L, mean_v, False, True) * K
    ms = out["random"][0]
    return {
        "workload": f"cfg3 training mode: {K} windows x {L} bp per batch, deterministic=False (shifts ~ U[0, max_shift] per haplotype) + jitter 16, "
                    f"one-hot (K, L, 4), from dataset indices through the native loader (in_flight 3, groups of 16), {out['random'][2]} batches per epoch",
        "ms_per_step": ms, "windows_per_s": K / (ms * 1e-3), "epochs_timed": out["random"][1],
        "deterministic_ms_per_step": out["deterministic"][0], "vs_deterministic": ms / out["deterministic"][0],
        "algorithmic_bytes_per_step": abytes, "step_frac": abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "how": "host clock over chained epochs (no synchronisation between them), steady state; the deterministic loader the same way just before",
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg3", choices=["cfg1", "cfg2", "cfg3", "cfg4"])
    ap.add_argument("--scale", default="hg38", choices=["hg38", "small"],
                    help="dataset size: hg38 = 3.09 Gbp / > 1 GB of genotype records (cold inputs); small = 64 Mbp")
    ap.add_argument("--out-slots", type=int, default=0, help="output buffers the steps rotate over (default: streams + 1 per batch of a launch)")
    ap.add_argument("--rotate", type=int, default=256,
                    help="distinct batches the steps cycle through (256: 2.7 x the Infinity Cache; 64: fits it; 1 = cache-hot)")
    ap.add_argument("--queries", type=int, default=None, help="override the number of queries in the dataset")
    ap.add_argument("--haps", action="store_true", help="also materialise haplotype bytes (h=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the `verified` block (two batches of the last timed launch rebuilt by the oracle and compared byte for byte)")
    ap.add_argument("--cpu-only", action="store_true", help="time the CPU oracle only (cfg1 plumbing case); no GPU")
    ap.add_argument("--cpu-budget", type=float, default=8.0)
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the launches of the timed region rotate over = launches in flight (3, the native loader's "
                         "in_flight; sweeps: profiles/r04_pipe_experiments.txt F, I, J)")
    ap.add_argument("--many", type=int, default=16,
                    help="batches per launch (gvl_reconstruct_many: ONE grid over the group -- 16 = the native loader's default group; "
                         "a step is still ONE batch; a 20-step region is a launch of 16 and one of 4, both in flight).  "
                         "1 = a launch per batch (round 3's measurement)")
    ap.add_argument("--min-region-ms", type=float, default=1000.0,
                    help="GPU time to sample per timed leg (repeated K-step regions)")
    ap.add_argument("--max-regions", type=int, default=20000)
    ap.add_argument("--max-leg-s", type=float, default=8.0, help="host time one timed leg may take")
    ap.add_argument("--sustained-s", type=float, default=6.0,
                    help="length of the sustained leg (back-to-back cold batches, one event pair); 0 = skip")
    ap.add_argument("--strong", action="store_true", help="N > 1: split ONE batch across the ranks (strong scaling)")
    ap.add_argument("--gather", action="store_true", help="N > 1: also time the RCCL all-gather of the one-hot shards")
    ap.add_argument("--no-hot", action="store_true", help="skip the extra cache-hot kernel timing")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="gvl_set_tuning before anything is launched (A/B runs): pipe_rows_x100, pipe_min_rows, lean_sub, track_plan_max_mb, ragged_sizing")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short ragged-cfg3 and cfg4 legs of the default run (N = 1 only; `secondary` in the line)")
    args = ap.parse_args()

    if args.cpu_only:
        cpu_only(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU under torch.distributed.run as a CHILD process --
        # before this process has imported torch or touched the GPU (never a re-exec of a process that has) --, relay its output
        # and exit with its code.  (Started under the launcher -- WORLD_SIZE set -- a mismatch with --gpus is still refused below.)
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd).returncode)
    if args.workload == "cfg4":
        import bench_cfg4  # haplotypes + tracks: its own step definition

        bench_cfg4.main(args)
        return

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 with torch.distributed.run")
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
        assert dist.get_world_size() == args.gpus

    from genvarloader_amd import HapsDevice, sharding, synth

    if args.tune:
        from genvarloader_amd import _lib as _tl

        keys = {"pipe_rows_x100": _tl.TUNE_PIPE_ROWS_X100, "pipe_min_rows": _tl.TUNE_PIPE_MIN_ROWS, "lean_sub": _tl.TUNE_LEAN_SUB,
                "track_plan_max_mb": _tl.TUNE_TRACK_PLAN_MAX_MB, "ragged_sizing": _tl.TUNE_RAGGED_SIZING,
                "hap_plan_max_mb": _tl.TUNE_HAP_PLAN_MAX_MB}
        for kv in args.tune:
            k_, v_ = kv.split("=")
            _tl.set_tuning(keys[k_], int(v_))

    # ---- synthetic dataset (per rank; --strong: the same one on every rank) ---------------
    cfg_idx = int(args.workload[3:])
    seed = 20260802 + cfg_idx + (0 if args.strong else 1000 * rank)
    t_gen = time.perf_counter()
    ds = synth.make_genome(args.scale, args.workload, device=f"cuda:{dev_index}", seed=seed, n_queries=args.queries)
    dev = HapsDevice(**ds.static_kwargs(), device=f"cuda:{dev_index}")
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    P, L = ds.ploidy, ds.length
    K_full = synth.CONFIGS[args.workload]["windows"]
    n_rot = max(1, args.rotate)
    qsets = ds.draw_batches(n_rot, K_full // P, seed=seed + 7)
    rc_on = synth.CONFIGS[args.workload]["rc_frac"] > 0

    def make_dbt(q):
        r = ds.request(q, rc=rc_on)
        if args.strong and world > 1:
            lo, hi = sharding.shard_bounds(int(q.numel()), world, rank)
            r = dict(regions=r["regions"][lo:hi].contiguous(), shifts=r["shifts"][lo:hi].contiguous(),
                     geno_offset_idx=r["geno_offset_idx"][lo:hi].contiguous(),
                     to_rc=None if r["to_rc"] is None else r["to_rc"][lo * P:hi * P].contiguous())
        return dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], L, to_rc=r["to_rc"])

    batches = [make_dbt(qsets[i]) for i in range(n_rot)]
    K = batches[0].n_rows                      # windows per step on this rank
    stream = torch.cuda.current_stream()
    streams = [stream] + [torch.cuda.Stream() for _ in range(max(0, args.streams - 1))]
    G = max(1, min(16, args.many))
    n_slots = (len(streams) + 1) * G           # an output slot per batch in flight (+1 launch being consumed)
    if args.out_slots > 0:                     # (more slots than the 256 MiB Infinity Cache holds: see DESIGN 5)
        n_slots = args.out_slots
    slots = [dev.alloc_output(batches[0], K * L, haps=args.haps, onehot=True) for _ in range(n_slots)]
    # ... and one output set of their own for the sustained leg's LAST launch (what `verified` reads): the leg's launches rotate over
    # streams + 1 sets on free-running streams, which drift apart over seconds of queued work -- the launch that finishes last into a
    # shared set need not be the one issued last (round 5's check met exactly that, and relaunched)
    vslots = [dev.alloc_output(batches[0], K * L, haps=args.haps, onehot=True) for _ in range(G)]
    mean_v = float(np.mean([float((dev.geno_offsets[1][b.geno_offset_idx.reshape(-1)]
                                   - dev.geno_offsets[0][b.geno_offset_idx.reshape(-1)]).double().mean())
                            for b in batches[: min(8, n_rot)]]))
    counter = [0]

    import ctypes as _C
    _dref = _C.byref(dev.c)
    _bref = [_C.byref(b.c) for b in batches]
    _sref = [_C.byref(s_[1]) for s_ in slots]
    _sptr = [_C.c_void_p(s_.cuda_stream) for s_ in streams]
    _fn = dev.lib.gvl_reconstruct
    _many = dev.lib.gvl_reconstruct_many

    def step_pipelined(i: int) -> None:
        # a batch is independent of the previous one: the loader keeps `--streams` batches in
        # flight on separate HIP streams, so the latency-bound head of one batch (parameter
        # and variant gathers, scans) overlaps the store-bound tail of another.  (The C-ABI entry with its
        # arguments converted once: 3 us of host time per launch -- through HapsDevice.launch it is 7, and a region of
        # hundreds of steps then runs at the host's rate, not the GPU's.)
        j = counter[0]
        counter[0] += 1
        if _fn(_dref, _bref[j % n_rot], _sref[j % n_slots], _sptr[i % len(streams)]):
            raise RuntimeError("gvl_reconstruct failed")

    class ManyStepper:
        """--many G: steps are gathered into launches of G batches (gvl_reconstruct_many: one grid over the group), launch g on
        stream g % streams; flush() sends the partial last group of a region.  hot: every launch re-reads the first G batches."""

        def __init__(self, use_streams, hot=False, bl=None, sl=None):
            self.pending, self.g, self.cache, self.group = 0, 0, {}, G
            self.streams, self.hot = use_streams, hot
            self.sp = [_C.c_void_p(s_.cuda_stream) for s_ in use_streams]
            self.bl = batches if bl is None else bl          # (the secondary legs bring their own batches and output slots)
            self.sl = slots if sl is None else sl
            self.nb = len(self.bl)

        def key(self, g, size):
            return (0 if self.hot else g % (self.nb // G if self.nb >= G else 1), size, g % (len(streams) + 1))

        def pack(self, g, size):
            key = self.key(g, size)
            p = self.cache.get(key)
            if p is None:
                b0, s0 = key[0] * G, key[2] * G
                p = self.cache[key] = dev.pack_many([self.bl[(b0 + i) % self.nb] for i in range(size)],
                                                    [self.sl[s0 + i][1] for i in range(size)])
            return p

        def prebuild(self, sizes):
            for g in range(max(1, self.nb // G) * (len(streams) + 1)):     # the argument arrays, ahead of the timed code
                for size in sizes:
                    if size:
                        self.pack(g, size)

        def __call__(self, i):
            self.pending += 1
            if self.pending == G:
                self.flush()

        def flush(self):
            if self.pending:
                b, o, n = self.pack(self.g, self.pending)
                if _many(_dref, b, o, n, self.sp[self.g % len(self.sp)]):
                    raise RuntimeError("gvl_reconstruct_many failed")
                self.g += 1
                self.pending = 0

    def step_single(i: int) -> None:
        j = counter[0]
        counter[0] += 1
        dev.launch(batches[j % n_rot], slots[j % n_slots][1], stream)

    def step_hot(i: int) -> None:
        dev.launch(batches[0], slots[i % n_slots][1], stream)

    k_all = K                                  # windows per step over all ranks
    if dist is not None:
        t = torch.tensor([K], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        k_all = int(t.item())
    tm = Timer(torch, dist, backend, streams, args.min_region_ms, args.max_regions, args.max_leg_s)
    steps = args.steps
    step_kern, step_hot_k = step_single, step_hot
    if G > 1:
        step_pipelined = ManyStepper(streams)
        step_kern = ManyStepper([stream])
        step_hot_k = ManyStepper([stream], hot=True)
        for st_ in (step_pipelined, step_kern, step_hot_k):
            st_.prebuild((G, steps % G))
    flush = getattr(step_pipelined, "flush", lambda: None)

    # ---- warmup, then the contract region: EXACTLY K steps between barrier + synchronize ----
    for i in range(args.warmup):
        step_pipelined(i)
    flush()
    torch.cuda.synchronize()
    tm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step_pipelined(i)
    flush()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tm.barrier()
    wall = tm.allmax([t1 - t0])[0]

    # ---- the same region repeated; GPU time per region from HIP events ---------------------
    region_ms, n_regions, spans = tm.measure(step_pipelined, steps, streams)
    # ---- the kernel's own duration: K launches back to back on ONE stream ---------------------
    # (--many G: a launch = one grid over G batches; its duration = the region's / launches, G x the bytes per launch)
    kern_region_ms, n_kregions, _ = tm.measure(step_kern, steps, [stream])
    kern_ms = kern_region_ms * G / steps
    hot_ms = None
    if not args.no_hot:
        hot_ms = tm.measure(step_hot_k, steps, [stream])[0] * G / steps
    # ---- sustained: seconds of back-to-back cold batches, the timed region's schedule, ONE event pair ----
    sustained = None
    last_call = None
    if args.sustained_s > 0:
        import ctypes as C
        import math

        sp = [C.c_void_p(s_.cuda_stream) for s_ in streams]
        if G > 1:
            n_groups = max(1, n_rot // G) * (len(streams) + 1)
            period = math.lcm(n_groups, len(streams))
            period = min(period, 4096 - 4096 % len(streams))
            packs = [step_pipelined.pack(j, G) for j in range(period)]
            calls = [(C.byref(dev.c), packs[j][0], packs[j][1], packs[j][2], sp[j % len(streams)]) for j in range(period)]
            sus_fn = dev.lib.gvl_reconstruct_many
            _vkeep = []

            def sus_final(j):
                key = step_pipelined.key(j, G)
                p_ = dev.pack_many([batches[(key[0] * G + i_) % n_rot] for i_ in range(G)], [vslots[i_][1] for i_ in range(G)])
                _vkeep.append(p_)
                return (C.byref(dev.c), p_[0], p_[1], p_[2], sp[j % len(streams)])
        else:
            period = math.lcm(n_rot, n_slots, len(streams))
            period = min(period, 4096 - 4096 % len(streams))
            calls = [(C.byref(dev.c), C.byref(batches[j % n_rot].c), C.byref(slots[j % n_slots][1]), sp[j % len(streams)])
                     for j in range(period)]
            sus_fn = dev.lib.gvl_reconstruct

            def sus_final(j):
                return (C.byref(dev.c), C.byref(batches[j % n_rot].c), C.byref(vslots[0][1]), sp[j % len(streams)])
        clk0 = gpu_clocks(dev_index)
        sus_ms, sus_n, sus_host, sus_launch = tm.sustained(calls, sus_fn, streams, args.sustained_s, region_ms / steps * G, final=sus_final)
        last_call = (sus_n - 1) % period              # (the leg's last launch: its outputs are what `verified` reads below)
        sus_ms /= G                                   # (a call = G steps)
        sus_n *= G
        tm.last_thirds = [t / G for t in tm.last_thirds]
        clk1 = gpu_clocks(dev_index)
        sustained = {"ms_per_step": sus_ms, "steps": sus_n, "seconds": sus_ms * sus_n * 1e-3,
                     "windows_per_s": k_all / (sus_ms * 1e-3),
                     "vs_median_region": sus_ms / (region_ms / steps),
                     # the leg in thirds: with 3 batches in flight the lean kernel's launches settle into a faster steady state
                     # after 1-2.5 s without a pause; `steady_ms_per_step` = the last third
                     "ms_per_step_thirds": list(tm.last_thirds), "steady_ms_per_step": tm.last_thirds[-1],
                     "steady_windows_per_s": k_all / (tm.last_thirds[-1] * 1e-3),
                     "host_enqueue_s": sus_host, "host_us_per_launch_unthrottled": sus_launch * 1e6,
                     "host_bound": bool(sus_launch * 1e3 > 0.9 * sus_ms),
                     "how": ("back-to-back gvl_reconstruct_many launches of %d batches each, launch g on stream g %% streams" % G if G > 1 else
                             "back-to-back gvl_reconstruct launches, step i on stream i % streams")
                            + ", rotating cold batches, one HIP event pair around all of them (no gate kernel, no synchronisation in between)",
                     "clocks_before": clk0, "clocks_after": clk1}
    # ---- verified: batches of the LAST TIMED LAUNCH against the oracle (rank 0; the oracle is the checker here, never the thing
    # measured).  The launch = the sustained leg's last gvl_reconstruct_many call (or, without that leg, the timed region's packed
    # arguments launched once more); two of its batches, one at an in-group position >= 11 -- where the second rows of the two-row
    # waves live -- are rebuilt on the host and compared byte for byte with what the kernel left in the output slots.
    verified = None
    if rank == 0 and not args.no_verify and not args.strong:      # (round 6: with or without the cpu_baseline leg -- every line checks what it timed)
        from oracle import oracle as _orc

        # (with the sustained leg: its last launch wrote the dedicated set `vslots`; without it: the timed region's packed arguments,
        # launched once more into the same dedicated set)
        if G > 1:
            g_v = last_call if last_call is not None else 0
            key = step_pipelined.key(g_v, G)
            b0 = key[0] * G
            if last_call is None:
                p_ = dev.pack_many([batches[(b0 + i_) % n_rot] for i_ in range(G)], [vslots[i_][1] for i_ in range(G)])
                if _many(_dref, p_[0], p_[1], p_[2], _sptr[0]):
                    raise RuntimeError("gvl_reconstruct_many failed")
            positions = sorted({0, min(G - 1, 13)})
            what = "gvl_reconstruct_many, %d batches in one grid" % G
        else:
            b0 = (last_call if last_call is not None else 0) % n_rot
            if last_call is None:
                dev.launch(batches[b0], vslots[0][1], stream)
            positions = [0]
            what = "gvl_reconstruct, one batch"
        torch.cuda.synchronize()
        hs_ = ds.host_static()
        exp_ = []
        for i_ in positions:
            hb_ = ds.host_batch(qsets[(b0 + i_) % n_rot], rc=rc_on)
            e_h, _, e_oh = _orc.reconstruct_haplotypes_fused(
                hb_.regions, hb_.shifts, hb_.geno_offset_idx, hb_.geno_offsets, hb_.geno_v_idxs, hs_.v_starts, hs_.ilens,
                hs_.alt_alleles, hs_.alt_offsets, hs_.ref, hs_.ref_offsets, hs_.pad_char, L, None, None, hb_.to_rc, True,
                onehot=True, n_threads=min(32, _orc.default_threads()))
            exp_.append((e_h, e_oh, hb_.n_windows))

        def _mismatches():
            n_bad = 0
            for i_, (e_h, e_oh, _) in zip(positions, exp_):
                got = vslots[i_][0]
                g_oh = got.onehot.cpu().numpy()
                ok_ = np.array_equal(g_oh, e_oh)
                if not ok_:          # (say what differs: rows, where in the row -- the first thing anyone will ask)
                    diff = (g_oh.reshape(-1, L * 4) != np.asarray(e_oh).reshape(-1, L * 4))
                    rows_bad = np.nonzero(diff.any(axis=1))[0]
                    first = int(rows_bad[0]) if len(rows_bad) else -1
                    cols = np.nonzero(diff[first])[0] if first >= 0 else []
                    print(f"bench.py: verified: group position {i_}: {len(rows_bad)} of {diff.shape[0]} rows differ; first row {first}, "
                          f"bytes {cols[:4].tolist() if len(cols) else []} .. {int(cols[-1]) if len(cols) else -1} ({len(cols)} bytes); rows {rows_bad[:8].tolist()}",
                          file=sys.stderr, flush=True)
                if args.haps:
                    ok_ = ok_ and np.array_equal(got.haps.cpu().numpy(), e_h)
                n_bad += 0 if ok_ else 1
            return n_bad

        bad = _mismatches()
        rows_v = sum(e[2] for e in exp_)
        verified = {"batches": len(positions), "mismatches": bad, "rows": rows_v, "in_group_positions": positions, "relaunched": False,
                    "timed_launch_equal": not bad,
                    "outputs": "an output set of its own (no other launch of the leg writes it: the free-running streams drift, and a shared set "
                               "may hold an earlier launch's bytes)",
                    "launch": ("the sustained leg's last launch" if last_call is not None else "the timed region's packed arguments, launched once more")
                              + " (" + what + ")",
                    "against": "oracle.reconstruct_haplotypes_fused (C restatement of the reference), one-hot"
                               + (" + haplotype bytes" if args.haps else "") + ", byte for byte"}
        if bad:
            raise SystemExit(f"bench.py: the timed launch does NOT equal the oracle: {verified}")

    # ---- single-batch latency, host wall clock: launch -> synchronize --------
    lat = []
    for i in range(50):
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        step_single(i)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t_a)
    single_ms = float(np.median(lat)) * 1e3

    # ---- optional: the final gather of the ranks' one-hot shards (never inside `value`) --------
    gather_ms = None
    if args.gather and dist is not None:
        oh = slots[0][0].onehot.view(K, L, 4)
        if backend != "nccl":
            oh = oh.cpu()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            tm.barrier()
            ta = time.perf_counter()
            g = sharding.all_gather_rows(oh)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - ta)
        assert g.shape[1:] == oh.shape[1:] and g.shape[0] >= K
        gather_ms = tm.allmax([float(np.median(ts)) * 1e3])[0]

    # ---- secondary legs (N = 1, default workload): the reference's default ragged rows and BASELINE config 4, a few seconds each ----
    secondary = None
    if world == 1 and args.workload == "cfg3" and not args.no_secondary and not args.haps:
        secondary = {}
        t_s = time.perf_counter()
        skip = os.environ.get("GVL_BENCH_SKIP", "")
        try:
            if "ragged" in skip:
                raise RuntimeError("skipped (GVL_BENCH_SKIP)")
            secondary["ragged"] = secondary_ragged(torch, dev, ds)
        except Exception as exc:      # (a secondary leg never takes the headline down)
            secondary["ragged"] = {"error": repr(exc)}
        secondary["ragged_s"] = round(time.perf_counter() - t_s, 2)
        t_s = time.perf_counter()
        try:
            if "svar2" in skip:
                raise RuntimeError("skipped (GVL_BENCH_SKIP)")
            secondary["svar2"] = secondary_svar2(torch)
        except Exception as exc:
            secondary["svar2"] = {"error": repr(exc)}
        secondary["svar2_s"] = round(time.perf_counter() - t_s, 2)
        t_s = time.perf_counter()
        try:
            if "cfg4" in skip:
                raise RuntimeError("skipped (GVL_BENCH_SKIP)")
            import bench_cfg4

            class _A:
                gpus, steps, warmup, min_region_ms, max_regions = 1, 20, 5, 1200.0, 400
            c4 = bench_cfg4.measure(_A, init_dist=False)
            secondary["cfg4"] = {
                "workload": c4["config"]["workload"], "dataset": c4["config"]["dataset"], "inputs": c4["config"]["inputs"],
                "ms_per_step": c4["ms_per_step"], "windows_per_s": c4["value"],
                "step_frac": c4["roofline"]["step_frac"], "step_GBps": c4["roofline"]["step_GBps"],
                "step_algorithmic_bytes": c4["roofline"]["step_algorithmic_bytes"],
                "kernel_ms": c4["roofline"]["kernel_ms"], "kernel": c4["roofline"]["kernel"], "kernel_frac": c4["roofline"]["frac"],
                "step_frac_no_scratch_track": c4["roofline"]["step_frac_no_scratch_track"],
                "step_algorithmic_bytes_no_scratch_track": c4["roofline"]["step_algorithmic_bytes_no_scratch_track"],
                "traffic": c4["roofline"]["step_traffic"], "step_frac_of_copy_ceiling": c4["roofline"]["step_frac_of_copy_ceiling"],
                "traffic_how": c4["roofline"]["traffic_how"],
                "kernels": c4["kernels"], "loop": c4["config"]["loop"], "steps": c4["steps"], "regions": c4["timing"]["regions"],
            }
        except Exception as exc:
            secondary["cfg4"] = {"error": repr(exc)}
        secondary["cfg4_s"] = round(time.perf_counter() - t_s, 2)
        # ... and the same step on a dataset whose inputs do NOT stay in the Infinity Cache (16 regions x 512 samples: 390 MB of
        # intervals alone; the leg above reads a 70 MB dataset that does): what a training set of real size pays per batch.  Such
        # an epoch goes without haplotype chunk plans (they cost more than the walks they save once they have to come from HBM:
        # profiles/r05_cfg4_plans_vs_size.txt).
        t_s = time.perf_counter()
        try:
            if "cfg4_cold" in skip:
                raise RuntimeError("skipped (GVL_BENCH_SKIP)")
            import argparse as _ap

            import bench_cfg4
            _A = _ap.Namespace(gpus=1, steps=20, warmup=5, min_region_ms=600.0, max_regions=400, samples=512)
            c4 = bench_cfg4.measure(_A, init_dist=False)
            secondary["cfg4_cold"] = {
                "ms_per_step": c4["ms_per_step"], "windows_per_s": c4["value"], "step_frac": c4["roofline"]["step_frac"],
                "step_GBps": c4["roofline"]["step_GBps"], "step_algorithmic_bytes": c4["roofline"]["step_algorithmic_bytes"],
                "workload": c4["config"]["workload"], "dataset": c4["config"]["dataset"], "inputs": c4["config"]["inputs"],
                "input_bytes": c4["config"]["input_bytes"], "loop": c4["config"]["loop"], "steps": c4["steps"], "regions": c4["timing"]["regions"]}
        except Exception as exc:
            secondary["cfg4_cold"] = {"error": repr(exc)}
        secondary["cfg4_cold_s"] = round(time.perf_counter() - t_s, 2)
        # ---- the reference's other output modes on cfg3's rows, each under the same schedule as the headline (groups of G batches
        # per gvl_reconstruct_many call, `--streams` calls in flight, rotating cold batches): annotated haplotypes (a11,
        # src/ffi/mod.rs:2237-2397), rows under an exonic keep mask (src/genotypes/mod.rs:132-176: the spliced path's every batch),
        # channel-major one-hot (K, 4, L) (docs/source/index.md:114-115), the reference-only fetch (a9, src/reference/mod.rs:56-120),
        # and training mode through the native loader (shifts + jitter)
        tm2 = Timer(torch, dist, backend, streams, 250.0, 2000, 2.0)
        # (the headline's own rotation, all of it: 256 batches = 690 MB of windows + slot lines, 2.7 x the Infinity Cache.  Round 5 rotated
        # these legs over 32-64 batches -- 86-172 MB, inside the cache -- and still called them cold.)
        nb2 = max(G, n_rot // G * G)
        bl2 = batches[:nb2] if nb2 <= n_rot else batches

        def temperature(n_batches, bytes_per_row):
            fp = int(n_batches) * K * bytes_per_row
            return {"rotating_batches": int(n_batches), "rotation_footprint_bytes": fp,
                    "inputs": ("cold: the rotation's inputs exceed the 256 MiB Infinity Cache" if fp > (320 << 20)
                               else "Infinity-Cache-warm: the rotation's inputs fit the 256 MiB cache")}

        def mode_leg(bl, make_out, bytes_per_window, kernel):
            sl = [make_out(bl[0]) for _ in range((len(streams) + 1) * G)]
            stp = ManyStepper(streams, bl=bl, sl=sl)
            stp.prebuild((G,))
            k2 = 2 * G
            for i_ in range(k2):
                stp(i_)
            stp.flush()
            torch.cuda.synchronize()
            _lib_mod.check_async()
            ms_, n_, _ = tm2.measure(stp, k2, streams)
            per = ms_ / k2
            ab = bytes_per_window * K
            del sl
            return {"ms_per_step": per, "windows_per_s": K / (per * 1e-3), "algorithmic_bytes_per_step": ab,
                    "step_frac": ab / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, "regions": n_, "kernel": kernel,
                    **temperature(len(bl), 128 + L // 2 + 48),
                    "how": "median of %d-step regions: gvl_reconstruct_many calls of %d batches on %d streams, %d rotating batches, HIP events"
                           % (k2, G, len(streams), len(bl))}

        def leg(name, fn):
            t_l = time.perf_counter()
            try:
                if name in skip or G < 2:
                    raise RuntimeError("skipped (GVL_BENCH_SKIP or --many 1)")
                secondary[name] = fn()
            except Exception as exc:
                secondary[name] = {"error": repr(exc)}
            secondary[name + "_s"] = round(time.perf_counter() - t_l, 2)
            torch.cuda.empty_cache()

        from genvarloader_amd import _lib as _lib_mod

        leg("onehot_cl", lambda: dict(mode_leg(
            bl2, lambda b: dev.alloc_output(b, K * L, haps=False, onehot=True, layout="cl"),
            algorithmic_bytes_per_window(L, mean_v, False, True), "one-hot (K, 4, L)"),
            workload=f"cfg3 rows, channel-major one-hot (K, 4, L): {K} windows x {L} bp"))
        leg("annotated", lambda: dict(mode_leg(
            bl2, lambda b: dev.alloc_output(b, K * L, haps=True, onehot=False, annotate=True),
            L * (1 + 1 + 8) + 28.0 * mean_v + 61.0, "haplotype bytes + annot_v_idxs + annot_ref_pos"),
            workload=f"cfg3 rows, annotated haplotypes (bytes + 2 x i32 per base = 9 B per base out): {K} windows x {L} bp"))

        def keep_leg():
            bl = []
            for b in bl2:
                kp, ko = dev.choose_exonic_variants(b.regions[:, 1].contiguous(), b.regions[:, 2].contiguous(), b.geno_offset_idx)
                bl.append(dev.prepare_batch(b.regions, b.shifts, b.geno_offset_idx, L, keep=kp, keep_offsets=ko, to_rc=b.to_rc))
            return dict(mode_leg(bl, lambda b: dev.alloc_output(b, K * L, haps=False, onehot=True),
                                 algorithmic_bytes_per_window(L, mean_v, False, True) + mean_v, "one-hot (K, L, 4) under a keep mask"),
                        workload=f"cfg3 rows under the exonic keep mask (choose_exonic_variants on the rows' own regions), one-hot (K, L, 4): {K} windows x {L} bp")
        leg("keep_mask", keep_leg)

        def reference_leg():
            import ctypes as C_

            from genvarloader_amd._lib import GvlRefBatch

            n_sets = max(G, (n_rot // 2) // G * G)          # (128 region sets x 4096 rows x 1 KB of packed reference: 0.5 GB of windows)
            regs, rcs, outs_ = [], [], []
            oo = (torch.arange(K + 1, dtype=torch.int64, device=dev.device) * L).contiguous()
            for i_ in range(n_sets):
                rg = torch.cat([batches[(2 * i_) % n_rot].regions, batches[(2 * i_ + 1) % n_rot].regions])[:K].contiguous()
                regs.append(rg)
                rcs.append((rg[:, 3] == -1).to(torch.uint8).contiguous() if rc_on else None)
            n_out = (len(streams) + 1) * G             # an output pair per batch in flight (+ one group being consumed)
            for j_ in range(n_out):
                outs_.append((torch.empty(K * L, dtype=torch.uint8, device=dev.device), torch.empty((K * L, 4), dtype=torch.uint8, device=dev.device)))
            fn_ = dev.lib.gvl_get_reference_many

            class RefStepper:
                """steps gathered into gvl_get_reference_many calls of G batches (ONE grid over the group), call g on stream g % streams"""

                def __init__(self):
                    self.pending, self.g, self.cache, self.group = 0, 0, {}, G

                def pack(self, g, size):
                    key = (g % (n_sets // G), size, g % (len(streams) + 1))
                    arr = self.cache.get(key)
                    if arr is None:
                        arr = (GvlRefBatch * size)()
                        for i_ in range(size):
                            rg, rc_ = regs[key[0] * G + i_], rcs[key[0] * G + i_]
                            o_b, o_h = outs_[key[2] * G + i_]
                            arr[i_] = GvlRefBatch(regions=rg.data_ptr(), regions_stride=4, n_rows=K, out_offsets=oo.data_ptr(), max_row_len=L,
                                                  to_rc=None if rc_ is None else rc_.data_ptr(), out=o_b.data_ptr(), onehot=o_h.data_ptr())
                        self.cache[key] = arr
                    return arr

                def __call__(self, i_):
                    self.pending += 1
                    if self.pending == G:
                        self.flush()

                def flush(self):
                    if self.pending:
                        if fn_(_dref, self.pack(self.g, self.pending), C_.c_int32(self.pending), _sptr[self.g % len(_sptr)]):
                            raise RuntimeError("gvl_get_reference_many failed")
                        self.g += 1
                        self.pending = 0
            stp = RefStepper()
            for g_ in range((n_sets // G) * (len(streams) + 1)):
                stp.pack(g_, G)
            for i_ in range(2 * G):
                stp(i_)
            stp.flush()
            torch.cuda.synchronize()
            k2 = 2 * G
            ms_, n_, _ = tm2.measure(stp, k2, streams)
            per = ms_ / k2
            ab = (L * (1 + 1 + 4) + 16 + 8 + 1) * K
            return {"workload": f"reference-only fetch (get_reference): {K} rows x {L} bp, reverse-complement on half the rows, bytes + one-hot (K, L, 4)",
                    "ms_per_step": per, "windows_per_s": K / (per * 1e-3), "algorithmic_bytes_per_step": ab,
                    "step_frac": ab / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, "regions": n_, **temperature(n_sets, L // 2 + 16),
                    "how": "median of %d-step regions: gvl_get_reference_many calls of %d batches on %d streams, %d rotating region sets, HIP events" % (k2, G, len(streams), n_sets)}
        leg("reference", reference_leg)
        leg("random_shifts", lambda: secondary_random_shifts(torch, dev, ds))
        t_l = time.perf_counter()
        try:
            if "long_modes" in skip:
                raise RuntimeError("skipped (GVL_BENCH_SKIP)")
            secondary.update(secondary_long_modes(torch))
        except Exception as exc:
            secondary["annotated_long"] = {"error": repr(exc)}
        secondary["long_modes_s"] = round(time.perf_counter() - t_l, 2)
        leg("spliced", lambda: secondary_spliced(torch))
        leg("spliced_large", lambda: secondary_spliced(torch, pairs=4096, n=6))

    lean = (dev.ref4 is not None and dev.slot_rec is not None and L <= 2048 and L % 4 == 0
            and (int(os.environ.get("GVL_DBG", "0")) & ~(2 | 4 | 32768 | 65536 | 262144 | 524288 | 33554432 | 67108864)) == 0)
    piped = (lean and G > 1 and K * G >= 2048
             and not (int(os.environ.get("GVL_DBG", "0")) & 67108864))
    if rank == 0:
        ms_per_step = region_ms / steps
        abytes = algorithmic_bytes_per_window(L, mean_v, args.haps, True) * K      # per batch
        lbytes = abytes * G                                                        # per launch: one grid over G batches
        achieved = lbytes / (kern_ms * 1e-3) / 1e9
        pipelined = abytes / (ms_per_step * 1e-3) / 1e9
        traffic = None
        tf = REPO / "profiles" / "traffic.json"
        if tf.exists():
            try:
                tj = json.loads(tf.read_text())
                key = (f"{args.workload}{'+haps' if args.haps else ''}@{args.scale}" + ("" if n_rot > 1 else "@hot")
                       + ("" if lean or args.workload != "cfg3" else "@allpurpose"))
                # (the pipelined kernel's own PMC pass is keyed ...@pipelined, bytes per BATCH; a launch moves G times that)
                traffic = tj.get(key + "@pipelined", tj.get(key)) if piped else tj.get(key)
                if traffic is not None and G > 1:
                    traffic = traffic * G
            except Exception:
                traffic = None
        sizes = ds.nbytes()
        res = {
            "metric": "haplotype windows/sec", "value": k_all / (ms_per_step * 1e-3), "unit": "windows/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {K_full} windows x {L} bp per " + ("job" if args.strong else "GPU") + ", "
                            + ("SNP+indel, reverse-complement on half the rows" if cfg_idx >= 3 else "SNP-only")
                            + ", uint8 one-hot (K,L,4)" + (" + haplotype bytes" if args.haps else ""),
                "windows_per_step_per_rank": K, "length_bp": L, "ploidy": P,
                "mean_variants_per_window": round(mean_v, 3),
                "scale": args.scale, "reference_bp": int(ds.ref.numel()), "n_variants": int(ds.v_starts.numel()),
                "genotype_entries": int(ds.geno_v_idxs.numel()), "dataset_queries": int(ds.n_queries),
                "dataset_bytes": sizes, "inline_record_bytes": 0 if dev.geno_rec is None else int(dev.geno_rec.numel()) * 4,
                "rotating_batches": n_rot,
                # what a rotation touches of the dataset: per window a slot line (128 B) and L / 2 packed reference bytes
                "rotation_footprint_bytes": int(n_rot * K * (128 + L // 2 + 48)),
                "inputs": ("cold (every step a different batch; the rotation touches more than the 256 MB Infinity Cache holds)"
                           if n_rot * K * (128 + L // 2 + 48) > (320 << 20) else
                           "Infinity-Cache-warm (every step a different batch, but the rotation's windows and slot lines fit the 256 MB cache)")
                          if n_rot > 1 and args.scale == "hg38"
                          else ("rotating" if n_rot > 1 else "cache-hot (one batch re-launched)"),
                "parallelism": f"world_size {world}: " + ("one batch split into contiguous query blocks" if args.strong
                                                          else "rows sharded over the ranks, one full batch per rank per step"),
                "streams": len(streams), "batches_per_launch": G, "batches_in_flight": len(streams) * G,
                "dataset_build_s": round(t_gen, 2),
                "layouts": {"slot_rec": dev.slot_rec is not None, "geno_rec": dev.geno_rec is not None, "ref4": dev.ref4 is not None,
                            "note": None if dev.slot_rec is not None else
                            "slot_rec NOT built (128 B x genotype slots exceeds a quarter of the free HBM): rows find their records through the CSR"},
                "kernel_path": ("lean, one grid per group (recon_lean_rows_kernel)" if piped else "lean") if lean else "all-purpose",
            },
            "timing": {
                "how": "median of repeated K-step regions, each between barrier+synchronize; GPU time of a region from "
                       "HIP events on the work streams (launches queued behind a gate kernel)",
                "regions": n_regions, "region_ms_min": min(spans), "region_ms_max": max(spans),
                "wall_ms_per_step": wall / steps * 1e3,
                "wall_how": "host clock around ONE K-step region incl. launch + synchronize latency",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                # fractions of the part's measured float4-copy rate are taken on the bytes actually MOVED (the PMC passes' `traffic`;
                # the algorithmic bytes when there is no PMC figure for this configuration): a fraction of a measured ceiling above 1
                # would be a units mismatch
                "frac_of_copy_ceiling": (traffic if traffic is not None else lbytes) / (kern_ms * 1e-3) / 1e9 / HBM_COPY_GBS,
                "copy_ceiling_bytes": "moved (profiles/traffic.json)" if traffic is not None else "algorithmic",
                "frac_of_store_ceiling": (traffic if traffic is not None else lbytes) / (kern_ms * 1e-3) / 1e9 / HBM_STORE_GBS,
                "store_ceiling": "%.0f GB/s: store-only kernels over outputs of 134 MB - 1 GB sustain 5.4-5.8 TB/s on this pool (tools/wbench.hip, "
                                 "profiles/r06_wbench.txt); nine tenths of this launch's traffic are stores" % HBM_STORE_GBS,
                "kernel": (("recon_lean_rows_kernel<onehot, haps=%s> (ONE grid over the launch's %d batches; a wave takes rows w, w + W, ... "
                            "-- one or two rows per wave -- and fetches a row's window + slot line by LDS-DMA; nibble-packed reference)"
                            % ("true" if args.haps else "false", G)) if piped else
                           "recon_lean_kernel<onehot, haps=%s> (nibble-packed reference; rows it cannot express run the all-purpose body inside the same launch)" % ("true" if args.haps else "false")
                           if lean else "reconstruct_kernel<OH_LC, haps=%s, annot=false>" % ("true" if args.haps else "false")),
                "kernel_ms": kern_ms, "batches_per_launch": G, "kernel_ms_per_batch": kern_ms / G,
                "kernel_ms_how": "HIP events around the K steps' launches back to back on ONE stream (rotating batches), median region, per launch"
                                 + (" = one grid over %d batches" % G if G > 1 else ""),
                "kernel_ms_hot": hot_ms, "hot_frac": None if hot_ms is None else lbytes / (hot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": lbytes, "algorithmic_bytes_per_batch": abytes,
                "pipelined_GBps": pipelined, "pipelined_frac": pipelined / HBM_PEAK_GBS,
                "pipelined_frac_of_copy_ceiling": ((traffic / G) if traffic is not None else abytes) / (ms_per_step * 1e-3) / 1e9 / HBM_COPY_GBS,
                "single_batch_wall_ms": single_ms,
            },
        }
        if verified is not None:
            res["verified"] = verified
        if sustained is not None:
            res["sustained"] = sustained
            res["sustained_ms_per_step"] = sustained["ms_per_step"]
        if secondary is not None:
            res["secondary"] = secondary
        if gather_ms is not None:
            res["gather_ms"] = gather_ms
            res["gather_bytes_per_rank"] = K * L * 4
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(ds.host_static(), ds.host_batch(qsets[0], rc=rc_on), args.cpu_budget)
        print(json.dumps(res), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
