"""BASELINE config 5 on one GPU: a 1M-window epoch (200 regions x 2504 samples x 2 haplotypes,
2048 bp, fused one-hot) through DeviceHapsDataset.to_dataloader -- index math, request prep
and reconstruction all on the device, 3 batches in flight.  Prints windows/s for the epoch.
(The 8-GPU form shards the batch list across ranks; no collective.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth
from genvarloader_amd.loader import DeviceHapsDataset

R, S, P, L = 200, int(os.environ.get("S", 2504)), 2, 2048
bs = int(os.environ.get("BATCH", 2048))            # (region, sample) pairs per batch = 4096 windows
if os.environ.get("SCALE"):
    # SCALE=hg38: the bench's genome-scale dataset (3.09 Gbp, 8.4 M regions x 1 sample x 2 haplotypes, 6.8 GB): an
    # epoch of 16.8 M windows whose batches are COLD (every window read once per epoch)
    g = synth.make_genome(os.environ["SCALE"], "cfg3", device="cuda")
    dev = HapsDevice(**g.static_kwargs())
    full_regions, S = g.full_regions.cpu().numpy(), 1
    R = int(full_regions.shape[0])
    print(f"genome-scale dataset: {R} regions x 1 sample x {P} = {R*P} windows per epoch")
    os.environ.setdefault("REPS", "2")
else:
    rng = np.random.default_rng(20260802 + 5)
    st = synth.make_static(rng, (64 << 20,), indel_frac=0.15)
    t0 = time.time()
    full_regions, go, gv = synth.make_grid(rng, st, R, S, P, L)
    print(f"grid: {R}x{S}x{P} = {R*S*P} windows, CSR nnz {len(gv)} ({time.time()-t0:.1f} s to generate)")
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv, pad_char=st.pad_char)
for det in (True, False):
    ragged = bool(os.environ.get("RAGGED"))          # RAGGED=1: output_length = -1, the reference's default row shape (deterministic only)
    if ragged and not det:
        continue
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=-1 if ragged else L, jitter=0 if det else 16, deterministic=det, seed=1)
    combos = ((1, 1, None, False), (3, 1, None, False), (2, 4, None, False), (3, 4, None, False), (4, 4, None, False),
              (3, 4, None, True), (3, 8, None, False), (3, 16, None, False), (3, 16, None, True), (2, 16, None, False),
              (3, None, None, False), (3, 16, torch.Generator().manual_seed(0), False))      # (group None = the loader's default)
    if os.environ.get("QUICK"):
        combos = ((4, 4, None, False),)
    if os.environ.get("COMBOS"):          # e.g. COMBOS=3x4,3x8t,2x8  (in_flight x group, t = producer thread)
        combos = tuple((int(c.rstrip("t").split("x")[0]), int(c.rstrip("t").split("x")[1]), None, c.endswith("t"))
                       for c in os.environ["COMBOS"].split(","))
    for in_flight, group, gen, thr in combos:
        dl = ds.to_dataloader(batch_size=bs, shuffle=True, generator=gen, in_flight=in_flight, threaded=thr, group=group)
        if os.environ.get("NO_PREFETCH"):          # A/B: every epoch prepared at its start (as before gvl_loader_prefetch_epoch)
            dl.prefetch_epochs = False
        for rep in range(int(os.environ.get("REPS", 4))):                        # first pass warms up
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0; t_first = None
            for batch in dl:
                if t_first is None:
                    t_first = time.perf_counter()
                n += int(batch.idx.numel()) * P
            t_issue = time.perf_counter()
            torch.cuda.synchronize(); t1 = time.perf_counter()
        dt = t1 - t0
        # epochs chained (no synchronisation between them: what a training loop does)
        n_chain = int(os.environ.get("CHAIN", 6))
        torch.cuda.synchronize(); c0 = time.perf_counter()
        for rep in range(n_chain):
            for batch in dl:
                pass
        torch.cuda.synchronize(); dc = (time.perf_counter() - c0) / n_chain
        print(f"deterministic={det} in_flight={in_flight} group={group} {'cpu-shuffle' if gen is not None else 'dev-shuffle'}{' threaded' if thr else ''}: {n} windows in {dt*1e3:.2f} ms -> {n/dt/1e6:.1f} M windows/s; "
              f"first batch after {1e3*(t_first-t0):.2f} ms, loop {1e6*(t_issue-t_first)/max(1,len(dl)-1):.1f} us/batch host, "
              f"steady {1e6*(t1-t_first)/max(1,len(dl)-1):.1f} us per {bs*P}-window batch; {n_chain} epochs chained: {dc*1e3:.2f} ms per epoch = {n/dc/1e6:.1f} M windows/s", flush=True)
