"""tools/dbg_long2.py <seed0> <case> <flags...>: bytes-only output of one FUZZ_LONG case down several GVL_DBG paths."""
import os, sys
os.environ["FUZZ_LONG"] = "1"
sys.path.insert(0, ".")
import numpy as np, torch
from tools import fuzz_lean
from genvarloader_amd import HapsDevice, _lib
from oracle import oracle

seed0, ci = int(sys.argv[1]), int(sys.argv[2])
oracle.build()
rng = np.random.default_rng(seed0 * 100003 + ci)
st, bt = fuzz_lean.one_case(rng)
args = (bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs, st.v_starts, st.ilens, st.alt_alleles,
        st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, bt.output_length, None, None, bt.to_rc, False)
hp, eo, eoh = oracle.reconstruct_haplotypes_fused(*args, onehot=True)
L, P = bt.output_length, bt.meta["P"]
odd = int((~np.isin(st.ref, np.frombuffer(b"ACGTN", np.uint8))).sum())
print("L", L, "P", P, "rows", hp.size // L, "V/row", bt.mean_variants, "odd ref bytes", odd, "pad", st.pad_char)
for flags in [int(x) for x in sys.argv[3:]] or [0]:
    _lib.load().gvl_set_debug_flags(flags)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
    dbt = dev.prepare_batch(torch.from_numpy(bt.regions).cuda(), torch.from_numpy(bt.shifts).cuda(), torch.from_numpy(bt.geno_offset_idx).cuda(), L,
                            to_rc=None if bt.to_rc is None else torch.from_numpy(bt.to_rc).cuda())
    for want in ((False, True), (True, True)):
        out, out_c = dev.alloc_output(dbt, hp.size, haps=True, onehot=want[0])
        out.haps.fill_(0xAA)
        dev.launch(dbt, out_c)
        torch.cuda.synchronize()
        got = out.haps.cpu().numpy().reshape(-1, L)
        exp = hp.reshape(-1, L)
        bad = got != exp
        per_chunk = [int(bad[:, c:c + 2048].sum()) for c in range(0, L, 2048)]
        print("flags", flags, "onehot too" if want[0] else "bytes only", "bad", int(bad.sum()), "sentinel left", int((got == 0xAA).sum()), "per chunk", per_chunk[:12])
        if bad.any():
            r = int(np.nonzero(bad.any(axis=1))[0][0]); p = int(np.nonzero(bad[r])[0][0]); p0 = max(0, p - 8)
            print("   row", r, "first bad", p, "exp", exp[r, p0:p0 + 40].tobytes(), "got", got[r, p0:p0 + 40].tobytes())
