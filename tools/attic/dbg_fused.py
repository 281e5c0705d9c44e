"""tools/dbg_fused.py <seed0> <case> [flags...]: one case of tools/fuzz_fused_tracks.py down GVL_DBG paths.
Randomised parity sweep of the realignment straight from intervals (gvl_tracks_batch with gvl_track_set.tile_complete:
realign_tracks_kernel<PAINT>, SrcPainted) against the oracle's paint + realign: random interval lists -- gaps, touching,
OVERLAPPING (the claim is then wrong: the window is rejected, values come from the list itself), dense lists (more than
256 candidates per window), long intervals, lists that end inside the window -- rows of several chunks, all five
insertion fills, jitter, shifts.  python tools/fuzz_fused_tracks.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth
from genvarloader_amd.loader import DeviceHapsTracksDataset
from oracle import oracle

oracle.build()
def run_case(seed0, ci, flags=0, verbose=False):
    from genvarloader_amd import _lib
    _lib.load().gvl_set_debug_flags(flags)
    bad = 0
    rng = np.random.default_rng(seed0 * 104729 + ci)
    R, S, P = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.choice([1, 2, 2, 3]))
    L = int(rng.choice([40, 256, 700, 2048, 2500, 4100, 6144, 9000]))
    contig = int(rng.integers(L + 2_000, L + 60_000))
    st = synth.make_static(rng, (contig,), density=float(rng.choice([1 / 300, 1 / 60, 1 / 12])),
                           indel_frac=float(rng.choice([0.15, 0.5, 0.9])), max_indel=int(rng.choice([3, 30, 300])))
    full_regions, go, gv = synth.make_grid(rng, st, R, S, P, L)
    # the track walk indexes its track with every variant's position relative to the query (tracks/mod.rs:264, :347:
    # bounds-checked in the reference, i.e. a panic): a dataset only lists variants INSIDE its regions, so drop what
    # the synthetic grid lists behind a region's end (the haplotype walk does not mind them; the oracle's C would read
    # the neighbouring query's track)
    keep_v = np.ones(len(gv), bool)
    ends_per_slot = np.repeat(full_regions[:, 2].astype(np.int64) - 40, S * P)
    for o in range(go.shape[1]):
        keep_v[go[0, o]:go[1, o]] = st.v_starts[gv[go[0, o]:go[1, o]]] < ends_per_slot[o]
    cnt = np.array([int(keep_v[go[0, o]:go[1, o]].sum()) for o in range(go.shape[1])], np.int64)
    gv = np.ascontiguousarray(gv[keep_v])
    offs_ = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    go = np.ascontiguousarray(np.stack([offs_[:-1], offs_[1:]]))
    mode = int(rng.integers(0, 5))          # 0 BigWig-like, 1 touching, 2 some overlaps, 3 dense (1-2 bp), 4 long + sparse
    starts, ends, vals, offs = [], [], [], [0]
    for r in range(R):
        q0, q1 = int(full_regions[r, 1]), int(full_regions[r, 2])
        for s_ in range(S):
            pos = q0 - int(rng.integers(0, 200))
            stop = q1 + int(rng.choice([-L // 3, 50, 400]))                 # (lists that end inside the query too)
            while pos < stop:
                if mode == 3: w, gap = int(rng.integers(1, 3)), int(rng.integers(0, 2))
                elif mode == 4: w, gap = int(rng.integers(200, 5000)), int(rng.integers(0, 3000))
                else: w, gap = int(rng.geometric(1 / 20)), (0 if mode == 1 else int(rng.integers(0, 8)))
                a = pos + gap
                if mode == 2 and rng.random() < 0.05 and len(starts) > offs[-1]:
                    a = max(starts[-1] + 1, a - int(rng.integers(1, 20)))          # reaches back into the interval before
                starts.append(a); ends.append(a + w); vals.append(float(rng.normal())); pos = max(pos, a + w)
            offs.append(len(starts))
    tracks = {"t": (np.array(starts, np.int32), np.array(ends, np.int32), np.array(vals, np.float32), np.array(offs, np.int64))}
    strategy = int(rng.integers(0, 5))
    param = {0: 0.0, 1: 0.0, 2: float(rng.normal()), 3: float(rng.integers(0, 8)), 4: float(rng.integers(0, 5))}[strategy]
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv, pad_char=st.pad_char)
    ds = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=tracks, strategy_id=strategy, param=param, base_seed=11,
                                 output_length=L, onehot=False, haps=True, jitter=int(rng.choice([0, 0, 16])), deterministic=bool(rng.random() < 0.5),
                                 seed=int(rng.integers(0, 1000)))
    honest = bool(ds._tile_complete[0])
    ds._track_sets[0].tile_complete = 1          # always claimed: right or wrong, the values must be exact
    idx = torch.from_numpy(rng.permutation(R * S).astype(np.int64)[: int(rng.integers(1, R * S + 1))]).cuda()
    batch = ds[idx]
    torch.cuda.synchronize()
    regions, shifts, goi = batch.regions.cpu().numpy(), batch.shifts.cpu().numpy(), batch.geno_offset_idx.cpu().numpy()
    to_rc = None if batch.to_rc is None else batch.to_rc.cpu().numpy().astype(bool)
    diffs = oracle.get_diffs_sparse(goi, gv, go, st.ilens, None, None, regions[:, 1], regions[:, 2], st.v_starts)
    tlen = (regions[:, 2] - regions[:, 1]).astype(np.int64) - np.minimum(diffs.min(axis=1), 0)
    track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
    n = int(idx.numel())
    out_offsets = np.arange(n * P + 1, dtype=np.int64) * L
    a, e, v, io = tracks["t"]
    exp = np.zeros(n * P * L, np.float32)
    oracle.intervals_and_realign_track_fused(exp, out_offsets, regions, shifts, goi, gv, go, st.v_starts, st.ilens,
                                             idx.cpu().numpy().astype(np.int64), a, e, v, io, track_offsets, np.array([param]), strategy, 11,
                                             None, None, to_rc)
    got = batch.tracks[:, 0].contiguous().cpu().numpy().ravel()
    if not np.array_equal(got.view(np.uint32), exp.view(np.uint32)):
        bad += 1
        w = np.nonzero(got.view(np.uint32) != exp.view(np.uint32))[0]
        print(f"MISMATCH case {ci} (seed {seed0}): mode {mode} honest {honest} L={L} P={P} n={n} strategy {strategy} first bad {w[:5]} of {len(w)} "
              f"(row {w[0] // L}, pos {w[0] % L}) got {got[w[0]]} exp {exp[w[0]]}", flush=True)
        if verbose:
            rows = sorted(set((w // L).tolist()))
            print('   bad rows', rows[:8], 'positions', [(int(x // L), int(x % L)) for x in w[:12]], 'to_rc', None if to_rc is None else [bool(to_rc[r]) for r in rows[:8]])
            r = int(w[0] // L); q = r // P
            print('   region', regions[q], 'shift', shifts.ravel()[r], 'tlen', int(tlen[q]))
            o = goi.ravel()[r]
            vs = gv[go[0, o]:go[1, o]]
            rel = st.v_starts[vs] - regions[q, 1]
            print('   row variants (rel pos, ilen) near the end:', [(int(a_), int(b_)) for a_, b_ in zip(rel, st.ilens[vs]) if a_ > int(tlen[q]) - 400][:40])
            print('   sum ilen', int(st.ilens[vs][(rel >= 0) & (rel < int(tlen[q]))].sum()))
            trk = np.zeros(int(track_offsets[-1]), np.float32)
            oracle.intervals_to_tracks(idx.cpu().numpy().astype(np.int64), regions[:, 1].copy(), a, e, v, io, trk, track_offsets)
            tq = trk[track_offsets[q]:track_offsets[q + 1]]
            print('   track tail', [(i_, float(tq[i_])) for i_ in range(len(tq) - 12, len(tq))])
            print('   exp tail', exp.reshape(-1, L)[r, -20:], ' got tail', got.reshape(-1, L)[r, -20:])
    return bad

if __name__ == '__main__':
    seed0, ci = int(sys.argv[1]), int(sys.argv[2])
    for fl in [int(x) for x in sys.argv[3:]] or [0]:
        print('GVL_DBG', fl, 'mismatch' if run_case(seed0, ci, fl, True) else 'ok')
