"""Randomised parity sweep of the SVAR2 two-source provider: HIP (gvl_svar2_merge + the library's kernels over the merged table) vs the
oracle's provider over the same decoded channels -- random contigs, variant densities (windows that cross the packed form's 16 entries
and the general form's 64-entry tiles), ploidy 1-3, lengths 1-9000, ragged / fixed, shifts, filter_exonic, reverse-complement, the
var_key / dense split from "everything dense" to "everything var_key".  python tools/fuzz_svar2.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import synth, svar2, _lib
from oracle import oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
oracle.build()
bad = 0
t0 = time.time()
for ci in range(n_cases):
    rng = np.random.default_rng(seed0 * 100019 + ci)
    n_contigs = int(rng.integers(1, 4))
    contigs = tuple(int(x) for x in rng.integers(12_000, 150_000, n_contigs))
    indel_frac = float(rng.choice([0.0, 0.1, 0.3, 0.6, 0.9]))
    density = float(rng.choice([1 / 400, 1 / 100, 1 / 30, 1 / 8, 1 / 3]))
    af = (float(rng.choice([0.3, 0.6, 2.0, 6.0])), float(rng.choice([0.5, 0.9, 2.5])))
    st = synth.make_static(rng, contigs, density=density, indel_frac=indel_frac, af_beta=af, max_indel=int(rng.choice([3, 30, 200])),
                           n_frac=float(rng.choice([0.0, 0.01, 0.2])))
    ploidy = int(rng.choice([1, 2, 2, 3]))
    L = int(rng.choice([1, 3, 17, 64, 255, 256, 500, 1000, 2048, 2049, 3001, 5000, 9000]))
    L = max(1, min(L, min(contigs) - 400))
    q = int(rng.integers(1, 40))
    ragged = rng.random() < 0.35
    bt = synth.make_batch(rng, st, q, ploidy, L, slack=int(rng.choice([0, 8, 40])), rc_frac=float(rng.choice([0.0, 0.5, 1.0])),
                          output_length=-1 if ragged else None, lookback=int(rng.choice([0, 40, 300])),
                          edge_frac=float(rng.choice([0.0, 0.3])))
    if not ragged and rng.random() < 0.6:
        bt.shifts = rng.integers(0, int(rng.choice([1, 5, 40, 400, 3000])) + 1, bt.shifts.shape).astype(np.int32)
    sv = synth.to_svar2(rng, st, bt, dense_af=float(rng.choice([0.0, 0.2, 0.5, 0.8, 1.1])), extra=float(rng.choice([0.0, 0.5, 2.0])))
    fe = bool(rng.random() < 0.3)
    a = (bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, bt.output_length)
    exp, eoff = oracle.reconstruct_haplotypes_from_svar2(*a, filter_exonic=fe)
    if bt.to_rc is not None:
        oracle.rc_flat_rows_inplace(exp, eoff, bt.to_rc)
    oh, off, got = svar2.reconstruct_haplotypes_from_svar2(*a, filter_exonic=fe, to_rc=bt.to_rc, onehot=True)
    ok = np.array_equal(off, eoff) and np.array_equal(got, exp) and np.array_equal(oh, oracle.onehot(exp))
    d = svar2.hap_diffs_svar2(bt.regions, ploidy, sv.vk_pos, sv.vk_ilen, sv.vk_off, sv.dense_pos, sv.dense_ilen, sv.dense_range,
                              sv.dense_present, sv.dense_present_off, fe)
    de = oracle.hap_diffs_svar2(bt.regions, ploidy, sv.vk_pos, sv.vk_ilen, sv.vk_off, sv.dense_pos, sv.dense_ilen, sv.dense_range,
                                sv.dense_present, sv.dense_present_off, fe)
    ok = ok and np.array_equal(d, de)
    try:
        _lib.check_async()
    except Exception as exc:
        ok = False
        print("async error:", exc)
    if not ok:
        bad += 1
        nvk, nd = np.diff(sv.vk_off), np.diff(sv.dense_range.reshape(-1, 2), axis=1).ravel()
        print(f"MISMATCH case {ci}: contigs={contigs} indel={indel_frac} dens={density:.3f} P={ploidy} L={L} q={q} ragged={ragged} "
              f"exonic={fe} vk/hap max {nvk.max() if len(nvk) else 0} window max {nd.max() if len(nd) else 0}")
print(f"{n_cases} cases, {bad} mismatches, {time.time()-t0:.1f} s")
sys.exit(1 if bad else 0)
