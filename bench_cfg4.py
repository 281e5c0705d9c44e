"""bench.py --workload cfg4: BASELINE.json configs[3] -- 256 windows x 131072 bp (Enformer length), SNP+indel,
one-hot + haplotype bytes + one BigWig-like track realigned to every haplotype (Repeat5p fill).

A step = one batch FROM DATASET INDICES through ``DeviceHapsTracksDataset``: request prep, the
fused reconstruct / one-hot kernel, and ``gvl_tracks_batch`` (scratch-track sizing, painting,
realignment, reversal) -- every input resident in HBM, batches rotate over the dataset.
``value`` = windows / s of that whole step; ``roofline`` = the dominant kernel of the step
(``recon_lean_kernel<onehot, haps, long>``; ``GVL_DBG=1048576``: ``reconstruct_kernel<OH_LC, haps, !annot>``) timed alone with HIP events over back-to-back launches,
``kernels`` lists the other kernels of the step the same way.  Algorithmic bytes per window
(SURVEY 8d): L (1 + 1 + 4) + 28 V + 61 for the haplotype half, + 4 L_track read + 4 L written per
track for the realignment, + 4 L_track written + 12 B per interval for the painting."""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0


def _traffic(key):
    """Memory-side bytes per launch from the PMC passes (profiles/traffic.json), or None."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")) as f:
            return json.load(f).get(key)
    except Exception:
        return None


def build(device, R=16, S=64, P=2, L=131072, seed=20260802 + 4, contig=256 << 20):
    import torch

    from genvarloader_amd import HapsDevice, synth
    from genvarloader_amd.loader import DeviceHapsTracksDataset

    rng = np.random.default_rng(seed)
    st = synth.make_static(rng, (contig,), indel_frac=float(os.environ.get("GVL_CFG4_INDEL_FRAC", 0.15)))     # (0: what the chunks with indels cost)
    full_regions, go, gv = synth.make_grid(rng, st, R, S, P, L)
    starts, ends, vals, offs = [], [], [], [0]
    for r in range(R):
        q0, q1 = int(full_regions[r, 1]), int(full_regions[r, 2])
        n = (q1 - q0 + 200) // 33
        w = rng.geometric(1 / 25, size=(S, n))
        g = rng.geometric(1 / 8, size=(S, n))
        for s_ in range(S):
            s0 = np.cumsum(w[s_] + g[s_]) - w[s_] + q0 - 100
            m = s0 < q1 + 50
            starts.append(s0[m]); ends.append((s0 + w[s_])[m]); vals.append((rng.random(int(m.sum())) * 8).astype(np.float32))
            offs.append(offs[-1] + int(m.sum()))
    tracks = {"cov": (np.concatenate(starts).astype(np.int32), np.concatenate(ends).astype(np.int32), np.concatenate(vals),
                      np.asarray(offs, np.int64))}
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv, pad_char=st.pad_char, device=device)
    ds = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=tracks, output_length=L, onehot=True, haps=True)
    mean_v = float((go[1] - go[0]).mean())
    return st, dev, ds, tracks, mean_v


def main(args) -> None:
    res = measure(args)
    if res is not None:
        print(json.dumps(res), flush=True)


def measure(args, init_dist=True):
    """The cfg4 step + its kernels -> the bench line as a dict (rank 0; None on the other ranks).  bench.py's default run calls
    this with a short budget for its ``secondary.cfg4`` keys (init_dist=False: single process, no process group)."""
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = int(os.environ.get("GVL_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dist = None
    backend = os.environ.get("GVL_BENCH_BACKEND", "nccl")
    if world > 1 and init_dist:
        import torch.distributed as dist

        dist.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))
    R, S, P, L = int(os.environ.get("GVL_CFG4_R", 16)), int(getattr(args, "samples", 0) or os.environ.get("GVL_CFG4_S", 64)), 2, 131072
    st, dev, ds, tracks, mean_v = build(f"cuda:{local}", R, S, P, L, seed=20260802 + 4 + 1000 * rank)
    bs = int(os.environ.get("GVL_CFG4_BS", 128))                # queries per batch = 256 windows
    order = np.random.default_rng(1).permutation(R * S)
    batches = [torch.from_numpy(order[i:i + bs].astype(np.int64)).cuda() for i in range(0, len(order), bs)]
    nb = len(batches)
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(1)]
    keep = [None] * (len(streams) + 1)

    loop = os.environ.get("GVL_CFG4_LOOP", "native")
    if loop == "native":
        # the native ring: epochs chained, shuffled on the device; request prep once per epoch, then per batch
        # reconstruct + gvl_tracks_batch into a ring slot, 3 batches ahead on the loader's streams
        dl = ds.to_dataloader(batch_size=bs, shuffle=True, seed=1, in_flight=int(os.environ.get("GVL_CFG4_INFLIGHT", 3)),
                              group=int(os.environ.get("GVL_CFG4_GROUP", 1)))

        def forever():
            while True:
                yield from dl
        it = forever()

        def step(i):
            keep[0] = next(it)
    else:
        def step(i):
            s = streams[i % len(streams)]
            with torch.cuda.stream(s):
                keep[i % len(keep)] = ds[batches[i % nb]]

    def barrier():
        if dist is not None:
            dist.barrier()

    steps, warm = args.steps, args.warmup
    for i in range(warm):
        step(i)
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    spans = []
    t_total0 = time.perf_counter()
    while True:
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        spans.append(time.perf_counter() - t0)
        barrier()
        n_reg = len(spans)
        stop = (sum(spans) * 1e3 >= args.min_region_ms and n_reg >= 3) or n_reg >= args.max_regions
        if dist is not None:
            t = torch.tensor([1.0 if stop else 0.0], device="cuda" if backend == "nccl" else "cpu")
            dist.broadcast(t, 0)
            stop = bool(t.item())
        if stop:
            break
    if dist is not None:
        t = torch.tensor(spans, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        spans = [float(x) for x in t.cpu()]
    wall = float(np.median(spans))

    # ---- the kernels of the step, each alone on one stream (HIP events around back-to-back launches) ----
    from genvarloader_amd import _lib, device as gdev

    lib = _lib.load()
    cur = torch.cuda.current_stream()
    idx0, reg, sh, goi, rc = ds.request(batches[0])
    dbt = dev.prepare_batch(reg, sh, goi, L, to_rc=rc)
    K = 2 * bs
    slot = dev.alloc_output(dbt, K * L, haps=True, onehot=True)

    def timeit(fn, n=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(n):
            fn()
        e1.record(cur); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n


    def bound(fn, *args):
        # the ctypes arguments are built ONCE: built per call (a dozen pointer conversions) the call costs the host ~ 40 us and
        # a 20 us kernel's back-to-back launches measure the host (round 3 and the first round-4 passes reported the painter at
        # 38-54 us = 0.18-0.24 of peak that way; rocprofv3 has its kernel at 21 us)
        return lambda: _lib.check(fn(*args))

    # the haplotype kernel three ways: with the rows' chunk plans made ahead (gvl_hap_plan: what the native loader does once per
    # epoch -- the step's way), making them itself in front of every launch, and without plans (round 4's kernel: every chunk-wave walks its row)
    plan = dev.hap_plan(dbt)
    dbt_p = dbt if plan is None else dev.prepare_batch(reg, sh, goi, L, to_rc=rc, hap_plan=plan)
    t_recon = timeit(bound(lib.gvl_reconstruct, C.byref(dev.c), C.byref(dbt_p.c), C.byref(slot[1]), gdev._stream_ptr()))
    t_recon_self = timeit(bound(lib.gvl_reconstruct, C.byref(dev.c), C.byref(dbt.c), C.byref(slot[1]), gdev._stream_ptr()))
    t_plan = None if plan is None else timeit(bound(lib.gvl_hap_plan, C.byref(dev.c), C.byref(dbt.c), gdev._ptr(plan), gdev._stream_ptr()))
    lib.gvl_set_debug_flags(int(os.environ.get("GVL_DBG", "0") or 0) | 536870912)
    t_recon_noplan = timeit(bound(lib.gvl_reconstruct, C.byref(dev.c), C.byref(dbt.c), C.byref(slot[1]), gdev._stream_ptr()))
    lib.gvl_set_debug_flags(-1)
    a, e, v, io, pm = ds._itv[0]
    qs, qe = reg[:, 1].contiguous(), reg[:, 2].contiguous()
    diffs = dev.get_diffs_sparse(goi, q_starts=qs, q_ends=qe)
    tlen = (qe - qs).to(torch.int64) - diffs.min(dim=1).values.clamp(max=0).to(torch.int64)
    toff = torch.zeros(bs + 1, dtype=torch.int64, device="cuda"); torch.cumsum(tlen, 0, out=toff[1:])
    scratch = torch.empty(int(toff[-1]), dtype=torch.float32, device="cuda")
    n_itv_batch = int((io[idx0 + 1] - io[idx0]).sum())
    t_paint = timeit(bound(lib.gvl_intervals_to_tracks,
        gdev._ptr(idx0), gdev._ptr(qs), C.c_int64(1), C.c_int64(bs), gdev._ptr(a), gdev._ptr(e), gdev._ptr(v), gdev._ptr(io),
        C.c_int64(int(a.numel())), gdev._ptr(pm), gdev._ptr(scratch), gdev._ptr(toff), C.c_int64(int(tlen.max())), gdev._stream_ptr()))
    # the same painting through gvl_paint_tracks with the bucket index (what the drop-in layer's intervals_to_tracks calls since round 4)
    ts_paint, _keep_ts = gdev.make_track_set(a, e, v, io, pm, ds._bkt[0], "cuda")
    ts_paint.tile_complete = 1 if ds._tile_complete[0] else 0
    t_paint_idx = timeit(bound(lib.gvl_paint_tracks,
        C.byref(ts_paint), gdev._ptr(idx0), gdev._ptr(qs), C.c_int64(1), C.c_int64(bs), gdev._ptr(scratch), gdev._ptr(toff),
        C.c_int64(int(tlen.max())), gdev._stream_ptr()))
    ooff = torch.arange(K + 1, dtype=torch.int64, device="cuda") * L
    tbt = dev.prepare_batch(reg, sh, goi, -1, None, None, rc, ooff, max_row_len=L)
    tout = torch.empty(K * L, dtype=torch.float32, device="cuda")
    par = (C.c_double * 1)(0.0)
    t_realign = timeit(bound(lib.gvl_realign_tracks, C.byref(dev.c), C.byref(tbt.c), gdev._ptr(scratch), gdev._ptr(toff), par,
                             C.c_int64(0), C.c_uint64(0), gdev._ptr(tout), gdev._stream_ptr()))
    dbg = int(os.environ.get("GVL_DBG", "0") or 0)
    lean_long = dev.ref4 is not None and dev.geno_rec is not None and not (dbg & (16384 | 1048576 | 16))
    kernel_name = ("recon_lean_kernel<onehot, haps, long> (one wave per 2048-base chunk)" if lean_long
                   else "reconstruct_kernel<OH_LC, haps=true, annot=false>")
    # the track half as the step runs it: gvl_tracks_batch = scratch sizing (two small kernels; once per EPOCH in the native
    # loop) + per track either the realignment straight from the intervals (tile_complete sets) or paint + realign
    fused = all(ds._tile_complete) and not (dbg & 4194304)
    n_scr = int(lib.gvl_tracks_scratch_bytes(C.c_int64(bs), C.c_int64(P), C.c_int64(ds._stride)))
    arena = torch.empty(((4 * K * L + 255) & ~255) + n_scr, dtype=torch.uint8, device="cuda")
    from genvarloader_amd._lib import GvlBatch
    gbt = GvlBatch(regions=reg.data_ptr(), regions_stride=4, shifts=sh.data_ptr(), geno_offset_idx=goi.data_ptr(), batch=bs, ploidy=P,
                   keep=None, keep_offsets=None, to_rc=None if rc is None else rc.data_ptr(), output_length=L, out_offsets=None, max_row_len=L)
    t_tracks = timeit(bound(lib.gvl_tracks_batch,
        C.byref(dev.c), C.byref(gbt), C.c_void_p(idx0.data_ptr()), ds._track_sets, C.c_int32(1), par, C.c_int64(0), C.c_uint64(0),
        C.c_void_p(arena.data_ptr()), C.c_int64(K * L), C.c_void_p(arena.data_ptr() + ((4 * K * L + 255) & ~255)), C.c_int64(ds._stride),
        gdev._stream_ptr()))
    # ... and what a caller that keeps THREE such calls in flight gets (own outputs and scratch per stream; HIP events around all of them):
    # alone, a call is its dependent chain in front of its stores -- nothing else runs meanwhile; in flight the chains hide behind the
    # other calls' stores
    def in_flight(make, n=30, n_streams=3):
        ss = [torch.cuda.Stream() for _ in range(n_streams)]
        fns = [make(s_) for s_ in ss]
        for f_ in fns:
            f_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for s_ in ss:
            s_.wait_stream(cur)
        for i_ in range(n):
            fns[i_ % n_streams]()
        for s_ in ss:
            cur.wait_stream(s_)
        e1.record(cur); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def make_paint(s_):
        sc_ = torch.empty_like(scratch)
        keep_alive.append(sc_)
        return bound(lib.gvl_paint_tracks, C.byref(ts_paint), gdev._ptr(idx0), gdev._ptr(qs), C.c_int64(1), C.c_int64(bs), gdev._ptr(sc_),
                     gdev._ptr(toff), C.c_int64(int(tlen.max())), gdev._stream_ptr(s_))

    def make_tracks(s_):
        ar_ = torch.empty_like(arena)
        keep_alive.append(ar_)
        return bound(lib.gvl_tracks_batch, C.byref(dev.c), C.byref(gbt), C.c_void_p(idx0.data_ptr()), ds._track_sets, C.c_int32(1), par,
                     C.c_int64(0), C.c_uint64(0), C.c_void_p(ar_.data_ptr()), C.c_int64(K * L),
                     C.c_void_p(ar_.data_ptr() + ((4 * K * L + 255) & ~255)), C.c_int64(ds._stride), gdev._stream_ptr(s_))

    keep_alive = []
    t_paint_fl = in_flight(make_paint)
    t_tracks_fl = in_flight(make_tracks)
    keep_alive.clear()
    if rank == 0:
        hap_bytes = (L * 6 + 28.0 * mean_v + 61.0) * K
        gv_n = dev.geno_v_idxs.numel() if getattr(dev, "geno_v_idxs", None) is not None else 0
        hp_all = int(lib.gvl_hap_plan_bytes(C.c_int64(R * S * P), C.c_int64(L)))
        hap_plan_epoch = hp_all if 0 < hp_all <= (64 << 20) and not (dbg & 536870912) else 0        # (the loader's cap: GVL_TUNE_HAP_PLAN_MAX_MB)
        realign_bytes = 4.0 * float(toff[-1]) + 4.0 * K * L
        paint_bytes = 4.0 * float(toff[-1]) + 12.0 * n_itv_batch
        ms_step = wall / steps * 1e3
        ach = hap_bytes / (t_recon * 1e-3) / 1e9
        res = {
            "metric": "haplotype windows/sec", "value": world * K * steps / wall, "unit": "windows/s", "n_gpus": world,
            "steps": steps, "warmup": warm, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 (haplotypes, one-hot) + f32 (tracks)", "data": "synthetic",
            "config": {"workload": f"cfg4: {K} windows x {L} bp per GPU from dataset indices, SNP+indel, one-hot + haplotype bytes "
                                   "+ 1 track painted from intervals and realigned (Repeat5p)",
                       "windows_per_step_per_rank": K, "length_bp": L, "ploidy": P, "mean_variants_per_window": round(mean_v, 1),
                       "dataset": f"{R} regions x {S} samples, 256 Mbp contig, {int(a.numel())} intervals",
                       # (what a batch reads besides the reference: its queries' interval lists, genotype records and plans.  A small
                       # dataset's stay in the 256 MB Infinity Cache from one epoch to the next; secondary.cfg4_cold's do not)
                       "input_bytes": {"intervals": 12 * int(a.numel()), "epoch_hap_plans": hap_plan_epoch, "genotype_records": 16 * int(gv_n)},
                       "inputs": ("resident in the 256 MB Infinity Cache (every query comes round again within a few batches)"
                                  if 12 * int(a.numel()) + 2 * hap_plan_epoch + 16 * int(gv_n) < (200 << 20) else "from HBM (the dataset does not fit the Infinity Cache)"),
                       "batches_in_flight": len(streams) if loop != "native" else dl.in_flight * dl.group,
                       "loop": "native ring (gvl_loader_*)" if loop == "native" else "python submit loop", "parallelism": f"world_size {world}: one batch per rank per step"},
            "timing": {"how": "median of K-step regions between barrier + synchronize, host clock", "regions": len(spans)},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": _traffic("cfg4" if lean_long else "cfg4@allpurpose"), "kernel": kernel_name, "kernel_ms": t_recon,
                         "kernel_ms_how": "HIP events around 30 back-to-back launches on one stream",
                         "algorithmic_bytes_per_launch": hap_bytes,
                         # SURVEY 8(d) for config 4: the haplotype half + per track 4 L_track read + 4 L written (the
                         # scratch track the painter writes when it runs is this design's own traffic, not counted)
                         "step_algorithmic_bytes": hap_bytes + realign_bytes,
                         "step_GBps": (hap_bytes + realign_bytes) / (ms_step * 1e-3) / 1e9,
                         "step_frac": (hap_bytes + realign_bytes) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         # ... SURVEY's formula charges the realignment a 4 L_track READ of a scratch track; the fused step never
                         # writes or reads one (values come straight from the intervals): the same step priced on what it has to
                         # touch -- the haplotype half + 4 L written per track + 12 B per interval of the batch's lists
                         "step_algorithmic_bytes_no_scratch_track": hap_bytes + 4.0 * K * L + 12.0 * n_itv_batch,
                         "step_frac_no_scratch_track": (hap_bytes + 4.0 * K * L + 12.0 * n_itv_batch) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         # bytes MOVED at the memory side (PMC, this round's build): both kernels of the step
                         "step_traffic": (None if not fused or _traffic("cfg4@r06:recon") is None or _traffic("cfg4@r06:paint") is None
                                          else _traffic("cfg4@r06:recon") + _traffic("cfg4@r06:paint")),
                         "step_frac_of_copy_ceiling": (None if not fused or _traffic("cfg4@r06:recon") is None or _traffic("cfg4@r06:paint") is None
                                                       else (_traffic("cfg4@r06:recon") + _traffic("cfg4@r06:paint")) / (ms_step * 1e-3) / 1e9 / 6290.0),
                         "traffic_how": "profiles/traffic.json cfg4@r06:* = rocprofv3 --pmc WRITE_SIZE + 2 x FETCH_SIZE per launch "
                                        "(profiles/r06_pmc_cfg4.txt); copy ceiling 6.29 TB/s = the part's measured float4 copy rate"},
            "kernels": {
                "recon_lean_kernel<long>, chunk plans made ahead (the step's way)": {"ms": t_recon, "algorithmic_bytes": hap_bytes, "frac": hap_bytes / (t_recon * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "recon_lean_kernel<long> behind its own hap_plan_kernel (a stand-alone gvl_reconstruct)": {"ms": t_recon_self, "algorithmic_bytes": hap_bytes, "frac": hap_bytes / (t_recon_self * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "recon_lean_kernel<long> without chunk plans (GVL_DBG 536870912: round 4's kernel)": {"ms": t_recon_noplan, "algorithmic_bytes": hap_bytes, "frac": hap_bytes / (t_recon_noplan * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "hap_plan_kernel (a batch's rows)": {"ms": t_plan},
                "realign_tracks_kernel": {"ms": t_realign, "algorithmic_bytes": realign_bytes,
                                          "frac": realign_bytes / (t_realign * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "intervals_to_tracks (tiled + per-value)": {"ms": t_paint, "algorithmic_bytes": paint_bytes,
                                                            "frac": paint_bytes / (t_paint * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "gvl_paint_tracks (bucket index: tiled + bitmap)": {"ms": t_paint_idx, "algorithmic_bytes": paint_bytes,
                                                                    "frac": paint_bytes / (t_paint_idx * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                                    "ms_three_in_flight": t_paint_fl,
                                                                    "frac_three_in_flight": paint_bytes / (t_paint_fl * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "gvl_tracks_batch (scratch sizing + " + ("realignment straight from the intervals" if fused else "paint + realign") + ")": {
                    "ms": t_tracks, "algorithmic_bytes": realign_bytes + (0.0 if fused else paint_bytes),
                    "frac": (realign_bytes + (0.0 if fused else paint_bytes)) / (t_tracks * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "ms_three_in_flight": t_tracks_fl,
                    "frac_three_in_flight": (realign_bytes + (0.0 if fused else paint_bytes)) / (t_tracks_fl * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "tracks_path": "realign_tracks_kernel<PAINT> (no scratch track)" if fused else "intervals_to_tracks_tiled_kernel + realign_tracks_kernel",
                "sum_of_kernels_ms": t_recon + (t_tracks if fused else t_realign + t_paint)},
        }
    else:
        res = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return res
