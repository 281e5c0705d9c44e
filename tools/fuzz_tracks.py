"""Randomised parity sweep for the track path (paint -> realign -> reverse) vs the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import synth
import genvarloader_amd.ffi as ffi
from oracle import oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
oracle.build()
bad = 0
t0 = time.time()
for ci in range(n_cases):
    rng = np.random.default_rng(seed0 * 7919 + ci)
    contig = int(rng.integers(5_000, 80_000))
    st = synth.make_static(rng, (contig,), density=float(rng.choice([1 / 200, 1 / 40, 1 / 10, 1 / 3])),
                           indel_frac=float(rng.choice([0.1, 0.4, 0.9])), max_indel=int(rng.choice([3, 30, 100])))
    P = int(rng.choice([1, 2, 3]))
    L = int(rng.choice([1, 5, 100, 255, 256, 700, 2048, 2500, 4100]))
    L = max(1, min(L, contig - 300))
    q = int(rng.integers(1, 16))
    bt = synth.make_batch(rng, st, q, P, L, slack=int(rng.choice([0, 30])), rc_frac=0.5, lookback=int(rng.choice([0, 60])))
    if rng.random() < 0.6:
        bt.shifts = rng.integers(0, int(rng.choice([2, 30, 500])) + 1, bt.shifts.shape).astype(np.int32)
    keep = ko = None
    if rng.random() < 0.3:
        idx = bt.geno_offset_idx.ravel()
        n_per = bt.geno_offsets[1, idx] - bt.geno_offsets[0, idx]
        ko = np.concatenate([[0], np.cumsum(n_per)]).astype(np.int64)
        keep = rng.random(int(ko[-1])) < 0.7
    B = bt.regions.shape[0]
    # tracks must cover every index the walk can read: region length + room for deletions
    tlen = (bt.regions[:, 2] - bt.regions[:, 1]).astype(np.int64) + int(rng.choice([0, 0, 500]))
    diffs = oracle.get_diffs_sparse(bt.geno_offset_idx, bt.geno_v_idxs, bt.geno_offsets, st.ilens, None, None,
                                    bt.regions[:, 1], bt.regions[:, 2], st.v_starts)
    tlen = np.maximum(tlen, (bt.regions[:, 2] - bt.regions[:, 1]) - np.minimum(diffs.min(axis=1), 0) + 3)
    track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
    starts, ends, vals, offs = [], [], [], [0]
    for b in range(B):
        s0 = int(bt.regions[b, 1]); pos = s0 - int(rng.integers(0, 30)); e0 = s0 + int(tlen[b])
        while pos < e0 + 10:
            w = int(rng.geometric(1 / 20)); gap = int(rng.integers(0, 4))
            if rng.random() < 0.1: gap = -min(3, w)               # overlapping intervals: last writer wins
            starts.append(pos + gap); ends.append(pos + gap + w); vals.append(float(rng.normal()))
            pos = max(pos + gap, pos) + w if gap >= 0 else pos + 1
        order = np.argsort(np.array(starts[offs[-1]:]), kind="stable")
        seg = slice(offs[-1], len(starts))
        for arr in (starts, ends, vals):
            a = np.array(arr[seg])[order]; arr[seg] = a.tolist()
        offs.append(len(starts))
    strategy = int(rng.integers(0, 5))
    param = {0: 0.0, 1: 0.0, 2: float(rng.normal()), 3: float(rng.integers(0, 8)), 4: float(rng.integers(0, 5))}[strategy]
    out_offsets = np.arange(B * P + 1, dtype=np.int64) * L
    args = (out_offsets, bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_v_idxs, bt.geno_offsets, st.v_starts, st.ilens,
            np.arange(B, dtype=np.int64), np.array(starts, np.int32), np.array(ends, np.int32), np.array(vals, np.float32),
            np.array(offs, np.int64), track_offsets, np.array([param]), strategy, int(rng.integers(0, 2**62)), keep, ko, bt.to_rc)
    exp = np.full(B * P * L, 3.0, np.float32); oracle.intervals_and_realign_track_fused(exp, *args)
    got = np.full(B * P * L, 5.0, np.float32); ffi.intervals_and_realign_track_fused(got, *args)
    if not np.array_equal(got.view(np.uint32), exp.view(np.uint32)):
        bad += 1
        print(f"MISMATCH case {ci}: strategy={strategy} param={param} P={P} L={L} q={q} V/row={bt.mean_variants:.1f} "
              f"shiftmax={bt.shifts.max()} keep={keep is not None} n_bad={(got.view(np.uint32) != exp.view(np.uint32)).sum()}")
print(f"{n_cases} track cases, {bad} mismatches, {time.time()-t0:.1f} s")
sys.exit(1 if bad else 0)
