"""tools/dbg_case.py <seed0> <case> [flags]: one fuzz_lean case in detail (mismatching rows, their variants)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from tools import fuzz_lean
from genvarloader_amd import HapsDevice, _lib
from oracle import oracle

seed0, ci = int(sys.argv[1]), int(sys.argv[2])
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
oracle.build()
rng = np.random.default_rng(seed0 * 100003 + ci)
st, bt = fuzz_lean.one_case(rng)
_lib.load().gvl_set_debug_flags(flags)
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length, None, None, bt.to_rc, haps=False, onehot=True)
args = (bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs, st.v_starts, st.ilens, st.alt_alleles,
        st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, bt.output_length, None, None, bt.to_rc, False)
hp, eo, eoh = oracle.reconstruct_haplotypes_fused(*args, onehot=True)
L, P = bt.output_length, bt.meta["P"]
got = out.onehot.cpu().numpy().reshape(-1, L, 4); exp = eoh.reshape(-1, L, 4)
bad = np.nonzero((got != exp).any(axis=(1, 2)))[0]
print("L", L, "P", P, "rows", got.shape[0], "bad rows", bad)
for k in bad[:4]:
    q = k // P
    o = bt.geno_offset_idx.ravel()[k]
    vs = bt.geno_v_idxs[bt.geno_offsets[0, o]:bt.geno_offsets[1, o]]
    print("row", k, "region", bt.regions[q], "shift", bt.shifts.ravel()[k], "rc", None if bt.to_rc is None else bt.to_rc[k], "n_var", len(vs))
    for v in vs[:70]:
        a = st.alt_alleles[st.alt_offsets[v]:st.alt_offsets[v + 1]].tobytes()
        print("   pos", st.v_starts[v], "ilen", st.ilens[v], "alt", a[:12], len(a))
    print("  exp hap", hp.reshape(-1, L)[k].tobytes(), " exp oh", exp[k].argmax(-1) * (exp[k].sum(-1) > 0) - (exp[k].sum(-1) == 0), "\n  got oh", got[k].argmax(-1) * (got[k].sum(-1) > 0) - (got[k].sum(-1) == 0))
