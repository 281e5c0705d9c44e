"""Randomised parity sweep of the LEAN kernel (fixed-length rows of at most 2048 bases, one-hot and / or haplotype bytes:
what gvl_reconstruct sends to recon_lean_kernel) against the oracle.  Dense rows, long indels, shifts that
meet indels, windows over contig edges, overflowing slots: every row the lean path hands to its solo general path
is checked the same way.  python tools/fuzz_lean.py [n_cases] [seed]
FUZZ_LONG=1: rows of several 2048-base chunks (the kernel's LONG form: one wave per chunk, the row's walk replayed from
the CSR records) -- chunk borders inside alleles, behind deletions, rows with hundreds of variants.
FUZZ_RAGGED=1: output_length = -1 (rows at out_offsets, any length): the pipelined kernel's ragged form, with FUZZ_LONG the chunked one's.
FUZZ_SUB=n: consecutive chunks of a long row per wave (gvl_set_tuning(GVL_TUNE_LEAN_SUB)).
FUZZ_MANY=1: what bench.py and the native loader launch -- 4-16 batches of 1 500-6 000 queries each through gvl_reconstruct_many on
DEFAULT flags = ONE multi-workgroup grid of recon_lean_rows_kernel, every batch of every launch compared with the oracle; the rows per
wave drawn per case (the built-in policy, 1.5, exactly 2, 3, 8: a wave's second and later rows are the DMA prefetch a row ahead, the
counted vmcnt and the (batch, row) arithmetic of rows w + W, w + 2 W ...).  With FUZZ_RAGGED=1: the ragged form, offsets from
gvl_hap_offsets per batch.
FUZZ_MIXED=1 (round 6): ragged batches of MOSTLY short rows with a few of 2 600 ... 20 000 bases (a spliced batch's exons), the route forced
at any batch size (GVL_TUNE_MIXED_MIN_ROWS = 1): the pipelined kernel's ragged form whose FRONT workgroups run the long rows chunk by
chunk in parallel -- with keep masks, annotations, ploidy 1-3, rows that straddle contig edges; GVL_DBG 256: long rows at the waves' ends."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth
from oracle import oracle


LONG = bool(int(os.environ.get("FUZZ_LONG", "0")))
MANY = bool(int(os.environ.get("FUZZ_MANY", "0")))
MIXED = bool(int(os.environ.get("FUZZ_MIXED", "0")))
RAGGED = bool(int(os.environ.get("FUZZ_RAGGED", "0"))) or MIXED     # output_length = -1: rows at out_offsets (with FUZZ_LONG: the chunked kernel's ragged form)


def one_case(rng):
    n_contigs = int(rng.integers(1, 4))
    contigs = tuple(int(x) for x in rng.integers(3_000, 120_000, n_contigs))
    if LONG:
        contigs = tuple(int(x) for x in rng.integers(12_000, 200_000, n_contigs))
    indel_frac = float(rng.choice([0.0, 0.1, 0.3, 0.6, 0.9]))
    density = float(rng.choice([1 / 400, 1 / 100, 1 / 30, 1 / 8, 1 / 3]))
    max_indel = int(rng.choice([3, 30, 200, 3000]))
    af = (float(rng.choice([0.3, 0.6, 2.0, 6.0])), float(rng.choice([0.5, 0.9, 2.5])))
    st = synth.make_static(rng, contigs, density=density, indel_frac=indel_frac, af_beta=af, max_indel=max_indel,
                           n_frac=float(rng.choice([0.0, 0.01, 0.2])))
    if rng.random() < 0.2 and st.alt_alleles.size:      # ALT bytes that are not ACGT (N, IUPAC, lower case)
        m = rng.random(st.alt_alleles.size) < 0.1
        st.alt_alleles[m] = rng.choice(np.frombuffer(b"NRYacgtn*", np.uint8), int(m.sum()))
    if rng.random() < 0.2:
        m = rng.random(st.ref.size) < 0.05
        st.ref[m] = rng.choice(np.frombuffer(b"RYKMacgtn", np.uint8), int(m.sum()))
    ploidy = int(rng.choice([1, 2, 2, 3]))
    L = int(rng.choice([4, 8, 64, 252, 256, 260, 500, 512, 1000, 1024, 1500, 2044, 2048]))
    if LONG:
        L = int(rng.choice([2052, 2056, 4096, 4100, 5000, 6144, 8192, 8196, 10_000, 16_384, 40_000]))
    L = min(L, ((min(contigs) - 200) // 4) * 4) if min(contigs) > 400 else min(L, 64)
    L = max(L, 4)
    q = int(rng.integers(1, 40)) if not LONG else int(rng.integers(1, 6))
    bt = synth.make_batch(rng, st, q, ploidy, L, slack=int(rng.choice([0, 8, 40])), rc_frac=float(rng.choice([0.0, 0.5, 1.0])),
                          random_shifts=False, lookback=int(rng.choice([0, 40, 300, 3200])),
                          edge_frac=float(rng.choice([0.0, 0.0, 0.3, 1.0])), permute_csr=bool(rng.random() < 0.5),
                          contig_start_frac=float(rng.choice([0.0, 0.0, 0.5])))
    if rng.random() < 0.6:
        hi = int(rng.choice([1, 5, 40, 400, 3000]))
        bt.shifts = rng.integers(0, hi + 1, bt.shifts.shape).astype(np.int32)
    if rng.random() < 0.3:
        bt.regions = np.ascontiguousarray(bt.regions[:, :3])            # stride 3 like the goldens
    if RAGGED:
        bt.regions = bt.regions.copy()
        bt.regions[:, 2] += rng.integers(0, 7, len(bt.regions)).astype(bt.regions.dtype)      # (lengths of every residue mod 4)
        bt.output_length = -1
    if not LONG and rng.random() < 0.3:            # a keep mask (rows of one chunk: the pipelined kernel reads it with the slot line)
        idx = bt.geno_offset_idx.ravel()
        n_per = bt.geno_offsets[1, idx] - bt.geno_offsets[0, idx]
        bt.keep_offsets = np.concatenate([[0], np.cumsum(n_per)]).astype(np.int64)
        bt.keep = rng.random(int(bt.keep_offsets[-1])) < float(rng.choice([0.2, 0.7, 0.95]))
    return st, bt


def mixed_case(rng):
    contigs = tuple(int(x) for x in rng.integers(60_000, 400_000, int(rng.integers(1, 3))))
    st = synth.make_static(rng, contigs, density=float(rng.choice([1 / 400, 1 / 100, 1 / 30])), indel_frac=float(rng.choice([0.0, 0.2, 0.6])),
                           af_beta=(float(rng.choice([0.3, 0.6, 2.0])), float(rng.choice([0.9, 2.5]))), max_indel=int(rng.choice([3, 30, 200])),
                           n_frac=float(rng.choice([0.0, 0.01])))
    ploidy = int(rng.choice([1, 2, 2, 3]))
    q = int(rng.integers(40, 700))
    L = int(rng.choice([60, 150, 250, 400, 900]))
    bt = synth.make_batch(rng, st, q, ploidy, L, slack=int(rng.choice([0, 8, 40])), rc_frac=float(rng.choice([0.0, 0.5, 1.0])),
                          random_shifts=False, lookback=int(rng.choice([40, 300])), edge_frac=float(rng.choice([0.0, 0.0, 0.05])),
                          permute_csr=bool(rng.random() < 0.5), output_length=-1)
    bt.regions = bt.regions.copy()
    bt.regions[:, 2] += rng.integers(0, 7, len(bt.regions)).astype(bt.regions.dtype)
    n_long = int(rng.integers(1, 9))
    lq = rng.choice(q, min(n_long, q), replace=False)
    ends = bt.regions[lq, 1].astype(np.int64) + rng.integers(2600, 20_000, len(lq))
    lim = np.array([st.ref_offsets[c + 1] - st.ref_offsets[c] for c in bt.regions[lq, 0]], np.int64) + int(rng.choice([0, 0, 300]))   # (some over the contig's end)
    bt.regions[lq, 2] = np.minimum(ends, np.maximum(lim, bt.regions[lq, 1] + 2600)).astype(bt.regions.dtype)
    if rng.random() < 0.5:
        idx = bt.geno_offset_idx.ravel()
        n_per = bt.geno_offsets[1, idx] - bt.geno_offsets[0, idx]
        bt.keep_offsets = np.concatenate([[0], np.cumsum(n_per)]).astype(np.int64)
        bt.keep = rng.random(int(bt.keep_offsets[-1])) < float(rng.choice([0.2, 0.7, 0.95]))
    return st, bt


def check(st, bt, want=(True, False), layout="lc", annotate=False):
    onehot, haps = want
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
    assert dev.ref4 is not None and dev.slot_rec is not None
    out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length, bt.keep, bt.keep_offsets, bt.to_rc, haps=haps,
                          onehot=onehot, layout=layout, annotate=annotate)
    args = (bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs, st.v_starts, st.ilens, st.alt_alleles,
            st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, bt.output_length, bt.keep, bt.keep_offsets, bt.to_rc, False)
    eh, eo, eoh = oracle.reconstruct_haplotypes_fused(*args, onehot=True)
    ok = np.array_equal(out.out_offsets.cpu().numpy(), eo)
    if onehot and layout == "cl":
        ok = ok and np.array_equal(out.onehot.cpu().numpy(), eoh.reshape(-1, bt.output_length, 4).transpose(0, 2, 1))
    elif onehot:
        ok = ok and np.array_equal(out.onehot.cpu().numpy(), eoh)
    if haps:
        ok = ok and np.array_equal(out.haps.cpu().numpy(), eh)
    if annotate:
        _, av, ap, _ = oracle.reconstruct_annotated_haplotypes_fused(*args)
        ok = ok and np.array_equal(out.annot_v_idxs.cpu().numpy(), av) and np.array_equal(out.annot_ref_pos.cpu().numpy(), ap)
    return ok


def many_case(rng):
    """One launch of n_b batches (the last one may be shorter) cut out of one big draw -> (static, full batch, cuts, L, P, x100)."""
    contigs = tuple(int(x) for x in rng.integers(40_000, 400_000, int(rng.integers(1, 4))))
    st = synth.make_static(rng, contigs, density=float(rng.choice([1 / 400, 1 / 100, 1 / 30])), indel_frac=float(rng.choice([0.0, 0.15, 0.5])),
                           af_beta=(float(rng.choice([0.3, 0.6, 2.0])), float(rng.choice([0.9, 2.5]))), max_indel=int(rng.choice([3, 30, 200])),
                           n_frac=float(rng.choice([0.0, 0.01])))
    P = 2
    L = int(rng.choice([64, 256, 500, 508, 512, 1024, 1036, 2040, 2048]))
    n_b = int(rng.integers(4, 17))
    per = int(rng.integers(1500, 6001))
    cap = max(1, (40 << 20) // (L * P * n_b))            # (bounds the host side of the comparison: <= 40 MB of haplotype bytes a launch)
    per = min(per, cap)
    last = per if rng.random() < 0.5 else int(rng.integers(1, per + 1))
    nq = per * (n_b - 1) + last
    full = synth.make_batch(rng, st, nq, P, L, slack=int(rng.choice([0, 8, 40])), rc_frac=float(rng.choice([0.0, 0.5])),
                            random_shifts=bool(rng.random() < 0.5), lookback=int(rng.choice([40, 300])),
                            edge_frac=float(rng.choice([0.0, 0.0, 0.02])), permute_csr=bool(rng.random() < 0.5))
    if RAGGED:
        full.regions = full.regions.copy()
        full.regions[:, 2] += rng.integers(0, 7, len(full.regions)).astype(full.regions.dtype)
    cuts = [(i * per, min((i + 1) * per, nq)) for i in range(n_b)]
    x100 = int(rng.choice([0, 0, 150, 200, 300, 800]))
    if not RAGGED and rng.random() < 0.4:          # channel-major one-hot (rows, 4, L): lengths that end in a partial 16-base group too
        full.meta["layout"] = "cl"
    return st, full, cuts, L, P, x100


def check_many(st, full, cuts, L, P, x100, want):
    from genvarloader_amd import _lib

    onehot, haps = want
    layout = full.meta.get("layout", "lc") if onehot else "lc"
    # (bytes only: annotated -- the pipelined kernel's annotated form in one grid; bytes + row-major one-hot: annotated when x100 is 150 / 800)
    annotate = haps and ((not onehot and x100 != 300) or (onehot and x100 in (150, 800)))          # (either one-hot layout)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=full.geno_offsets, geno_v_idxs=full.geno_v_idxs, pad_char=st.pad_char)
    assert dev.ref4 is not None and dev.slot_rec is not None
    _lib.set_tuning(_lib.TUNE_PIPE_ROWS_X100, x100)
    try:
        bts, outs, keep = [], [], []
        for a, b in cuts:
            rc = None if full.to_rc is None else full.to_rc[a * P:b * P]
            if RAGGED:
                d0 = dev.prepare_batch(full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], -1, to_rc=rc)
                oo, tm, _ = dev.hap_offsets(d0)
                total, mx = (int(v) for v in tm.cpu().tolist())
                dbt = dev.prepare_batch(d0.regions, d0.shifts, d0.geno_offset_idx, -1, to_rc=d0.to_rc, out_offsets=oo, max_row_len=mx)
            else:
                dbt = dev.prepare_batch(full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], L, to_rc=rc)
                total = (b - a) * P * L
            o, oc = dev.alloc_output(dbt, total, haps=haps, onehot=onehot, layout=layout, annotate=annotate)
            bts.append(dbt); outs.append(oc); keep.append(o)
        dev.launch_many(dev.pack_many(bts, outs))
        torch.cuda.synchronize()
        _lib.check_async()
    finally:
        _lib.set_tuning(_lib.TUNE_PIPE_ROWS_X100, 0)
    ok = True
    for i, (a, b) in enumerate(cuts):
        rc = None if full.to_rc is None else full.to_rc[a * P:b * P]
        eh, eo, eoh = oracle.reconstruct_haplotypes_fused(
            full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], full.geno_offsets, full.geno_v_idxs, st.v_starts, st.ilens,
            st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, -1 if RAGGED else L, None, None, rc, True, onehot=True,
            n_threads=8)
        ok = ok and np.array_equal(keep[i].out_offsets.cpu().numpy(), eo)
        if onehot and layout == "cl":
            ok = ok and np.array_equal(keep[i].onehot.cpu().numpy(), eoh.reshape(-1, L, 4).transpose(0, 2, 1))
        elif onehot:
            ok = ok and np.array_equal(keep[i].onehot.cpu().numpy(), eoh)
        if haps:
            ok = ok and np.array_equal(keep[i].haps.cpu().numpy(), eh)
        if annotate:
            _, av, ap, _ = oracle.reconstruct_annotated_haplotypes_fused(
                full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], full.geno_offsets, full.geno_v_idxs, st.v_starts, st.ilens,
                st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, -1 if RAGGED else L, None, None, rc, True, n_threads=8)
            ok = ok and np.array_equal(keep[i].annot_v_idxs.cpu().numpy(), av) and np.array_equal(keep[i].annot_ref_pos.cpu().numpy(), ap)
    return ok


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    oracle.build()
    if int(os.environ.get("FUZZ_SUB", "0")) > 0:
        from genvarloader_amd import _lib

        _lib.set_tuning(_lib.TUNE_LEAN_SUB, int(os.environ["FUZZ_SUB"]))
    if MIXED:
        from genvarloader_amd import _lib

        _lib.set_tuning(_lib.TUNE_MIXED_MIN_ROWS, 1)
    if MANY:
        bad, t0, rows = 0, time.time(), 0
        for ci in range(n_cases):
            rng = np.random.default_rng(seed0 * 100003 + ci)
            st, full, cuts, L, P, x100 = many_case(rng)
            want = ((True, False), (True, True), (False, True))[ci % 3]
            rows += cuts[-1][1] * P
            if not check_many(st, full, cuts, L, P, x100, want):
                bad += 1
                print(f"MISMATCH many-case {ci} (seed {seed0}) onehot, haps = {want}: L={L} batches={len(cuts)} per={cuts[0][1]} last={cuts[-1][1] - cuts[-1][0]} "
                      f"x100={x100} V/row={full.mean_variants:.1f}", flush=True)
        print(f"{n_cases} lean many-batch launches ({rows} rows), {bad} mismatches, {time.time()-t0:.1f} s")
        sys.exit(1 if bad else 0)
    bad = 0
    t0 = time.time()
    for ci in range(n_cases):
        rng = np.random.default_rng(seed0 * 100003 + ci)
        st, bt = mixed_case(rng) if MIXED else one_case(rng)
        want = ((True, False), (True, True), (False, True))[ci % 3]         # one-hot only / one-hot + bytes / bytes only
        # (fixed-length rows of one chunk: every fourth case channel-major -- the pipelined kernel's form, also on launches of one small batch)
        layout = "cl" if (want[0] and not RAGGED and ci % 4 == 1) else "lc"          # (long rows: the chunked kernel's channel-major form)
        # (bytes only, rows of one chunk: every other such case annotated -- the pipelined kernel's annotated form; bytes + row-major
        # one-hot: every fourth)
        annotate = not LONG and ((want == (False, True) and ci % 2 == 0) or (want == (True, True) and ci % 4 in (1, 3)))     # (ci % 4 == 1: channel-major)
        tc = time.time()
        ok = check(st, bt, want, layout, annotate)
        if time.time() - tc > float(os.environ.get("FUZZ_SLOW_S", "1e9")):
            print(f"slow case {ci}: {time.time() - tc:.1f} s  L={bt.output_length} P={bt.meta['P']} q={bt.meta['B']} V/row={bt.mean_variants:.1f} want={want} layout={layout}", flush=True)
        if not ok:
            bad += 1
            print(f"MISMATCH case {ci} (seed {seed0}) onehot, haps = {want}: L={bt.output_length} P={bt.meta['P']} q={bt.meta['B']} V/row={bt.mean_variants:.1f} "
                  f"shiftmax={bt.shifts.max()}", flush=True)
    print(f"{n_cases} lean cases, {bad} mismatches, {time.time()-t0:.1f} s")
    sys.exit(1 if bad else 0)
