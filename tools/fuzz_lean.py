"""Randomised parity sweep of the LEAN kernel (fixed-length rows of at most 2048 bases, one-hot and / or haplotype bytes:
what gvl_reconstruct sends to recon_lean_kernel) against the oracle.  Dense rows, long indels, shifts that
meet indels, windows over contig edges, overflowing slots: every row the lean path hands to its solo general path
is checked the same way.  python tools/fuzz_lean.py [n_cases] [seed]
FUZZ_LONG=1: rows of several 2048-base chunks (the kernel's LONG form: one wave per chunk, the row's walk replayed from
the CSR records) -- chunk borders inside alleles, behind deletions, rows with hundreds of variants.
FUZZ_RAGGED=1: output_length = -1 (rows at out_offsets, any length): the pipelined kernel's ragged form, with FUZZ_LONG the chunked one's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth
from oracle import oracle


LONG = bool(int(os.environ.get("FUZZ_LONG", "0")))
RAGGED = bool(int(os.environ.get("FUZZ_RAGGED", "0")))     # output_length = -1: rows at out_offsets (with FUZZ_LONG: the chunked kernel's ragged form)


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
    return st, bt


def check(st, bt, want=(True, False)):
    onehot, haps = want
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
    assert dev.ref4 is not None and dev.slot_rec is not None
    out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length, None, None, bt.to_rc, haps=haps, onehot=onehot)
    args = (bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs, st.v_starts, st.ilens, st.alt_alleles,
            st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, bt.output_length, None, None, bt.to_rc, False)
    eh, eo, eoh = oracle.reconstruct_haplotypes_fused(*args, onehot=True)
    ok = np.array_equal(out.out_offsets.cpu().numpy(), eo)
    if onehot:
        ok = ok and np.array_equal(out.onehot.cpu().numpy(), eoh)
    if haps:
        ok = ok and np.array_equal(out.haps.cpu().numpy(), eh)
    return ok


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    oracle.build()
    bad = 0
    t0 = time.time()
    for ci in range(n_cases):
        rng = np.random.default_rng(seed0 * 100003 + ci)
        st, bt = one_case(rng)
        want = ((True, False), (True, True), (False, True))[ci % 3]         # one-hot only / one-hot + bytes / bytes only
        if not check(st, bt, want):
            bad += 1
            print(f"MISMATCH case {ci} (seed {seed0}) onehot, haps = {want}: L={bt.output_length} P={bt.meta['P']} q={bt.meta['B']} V/row={bt.mean_variants:.1f} "
                  f"shiftmax={bt.shifts.max()}", flush=True)
    print(f"{n_cases} lean cases, {bad} mismatches, {time.time()-t0:.1f} s")
    sys.exit(1 if bad else 0)
