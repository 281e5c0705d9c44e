"""Randomised parity sweep: HIP path vs the oracle over many random configurations
(not part of the test suite; run on the GPU box: python tools/fuzz.py [n_cases] [seed])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth
from oracle import oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
oracle.build()
bad = 0
t0 = time.time()
for ci in range(n_cases):
    rng = np.random.default_rng(seed0 * 100003 + ci)
    n_contigs = int(rng.integers(1, 4))
    contigs = tuple(int(x) for x in rng.integers(3_000, 120_000, n_contigs))
    indel_frac = float(rng.choice([0.0, 0.1, 0.3, 0.6, 0.9]))
    density = float(rng.choice([1 / 400, 1 / 100, 1 / 30, 1 / 8, 1 / 3]))
    max_indel = int(rng.choice([3, 30, 200]))
    af = (float(rng.choice([0.3, 0.6, 2.0, 6.0])), float(rng.choice([0.5, 0.9, 2.5])))
    st = synth.make_static(rng, contigs, density=density, indel_frac=indel_frac, af_beta=af, max_indel=max_indel,
                           n_frac=float(rng.choice([0.0, 0.01, 0.2])))
    ploidy = int(rng.choice([1, 2, 2, 3]))
    L = int(rng.choice([1, 3, 17, 64, 255, 256, 500, 1000, 2048, 2049, 3001, 5000]))
    L = min(L, min(contigs) - 200) if min(contigs) > 400 else min(L, 64)
    L = max(L, 1)
    q = int(rng.integers(1, 40))
    ragged = rng.random() < 0.2
    bt = synth.make_batch(rng, st, q, ploidy, L, slack=int(rng.choice([0, 8, 40])), rc_frac=float(rng.choice([0.0, 0.5, 1.0])),
                          random_shifts=False, output_length=-1 if ragged else None, lookback=int(rng.choice([0, 40, 300])),
                          edge_frac=float(rng.choice([0.0, 0.3, 1.0])), permute_csr=bool(rng.random() < 0.5))
    if not ragged and rng.random() < 0.6:
        hi = int(rng.choice([1, 5, 40, 400, 3000]))
        bt.shifts = rng.integers(0, hi + 1, bt.shifts.shape).astype(np.int32)
    if rng.random() < 0.3:
        idx = bt.geno_offset_idx.ravel()
        n_per = bt.geno_offsets[1, idx] - bt.geno_offsets[0, idx]
        bt.keep_offsets = np.concatenate([[0], np.cumsum(n_per)]).astype(np.int64)
        bt.keep = rng.random(int(bt.keep_offsets[-1])) < rng.random()
    if rng.random() < 0.3:
        bt.regions = np.ascontiguousarray(bt.regions[:, :3])            # stride 3 like the goldens
    annotate = bool(rng.random() < 0.4)
    layout = "cl" if (not ragged and rng.random() < 0.25) else "lc"
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
    out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length, bt.keep, bt.keep_offsets, bt.to_rc,
                          haps=True, onehot=True, layout=layout, annotate=annotate)
    args = (bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs, st.v_starts, st.ilens, st.alt_alleles,
            st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, bt.output_length, bt.keep, bt.keep_offsets, bt.to_rc, False)
    exp, eo, eoh = oracle.reconstruct_haplotypes_fused(*args, onehot=True)
    ok = np.array_equal(out.haps.cpu().numpy(), exp) and np.array_equal(out.out_offsets.cpu().numpy(), eo)
    oh = out.onehot.cpu().numpy()
    if layout == "lc":
        ok = ok and np.array_equal(oh, eoh)
    else:
        ok = ok and np.array_equal(oh, eoh.reshape(bt.n_windows, L, 4).transpose(0, 2, 1))
    if annotate:
        _, av, ap, _ = oracle.reconstruct_annotated_haplotypes_fused(*args)
        ok = ok and np.array_equal(out.annot_v_idxs.cpu().numpy(), av) and np.array_equal(out.annot_ref_pos.cpu().numpy(), ap)
    if not ok:
        bad += 1
        print(f"MISMATCH case {ci}: contigs={contigs} indel={indel_frac} dens={density:.3f} P={ploidy} L={L} q={q} "
              f"ragged={ragged} annot={annotate} layout={layout} V/row={bt.mean_variants:.1f} shiftmax={bt.shifts.max()} keep={bt.keep is not None}")
print(f"{n_cases} cases, {bad} mismatches, {time.time()-t0:.1f} s")
sys.exit(1 if bad else 0)
