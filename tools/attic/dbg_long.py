"""tools/dbg_long.py <seed0> <case> [flags]: one FUZZ_LONG case of fuzz_lean in detail (which outputs / rows / positions differ)."""
import os, sys
os.environ["FUZZ_LONG"] = "1"
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
args = (bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs, st.v_starts, st.ilens, st.alt_alleles,
        st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, bt.output_length, None, None, bt.to_rc, False)
hp, eo, eoh = oracle.reconstruct_haplotypes_fused(*args, onehot=True)
L, P = bt.output_length, bt.meta["P"]
print("L", L, "P", P, "rows", hp.size // L, "V/row", bt.mean_variants)
for want in ((True, False), (True, True), (False, True)):
    out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, L, None, None, bt.to_rc, haps=want[1], onehot=want[0])
    torch.cuda.synchronize()
    print(want, "offsets ok", np.array_equal(out.out_offsets.cpu().numpy(), eo))
    if want[0]:
        got = out.onehot.cpu().numpy().reshape(-1, L, 4); exp = eoh.reshape(-1, L, 4)
        bad = np.nonzero((got != exp).any(axis=2))
        print("   onehot bad positions", len(bad[0]), list(zip(bad[0][:6], bad[1][:6])))
    if want[1]:
        got = out.haps.cpu().numpy().reshape(-1, L); exp = hp.reshape(-1, L)
        bad = np.nonzero(got != exp)
        print("   haps bad positions", len(bad[0]), [(r, p, bytes([exp[r, p]]), bytes([got[r, p]])) for r, p in zip(bad[0][:8], bad[1][:8])])
        for r in sorted(set(bad[0]))[:3]:
            ps = bad[1][bad[0] == r]
            print("   row", r, "rc", None if bt.to_rc is None else bt.to_rc[r], "first/last bad", ps.min(), ps.max(), "n", len(ps), "chunks", sorted(set(ps // 2048))[:10])

# bytes only once more into a buffer full of 0xAA: what does the kernel write at all?
dbt = dev.prepare_batch(torch.from_numpy(bt.regions).cuda(), torch.from_numpy(bt.shifts).cuda(), torch.from_numpy(bt.geno_offset_idx).cuda(), L,
                        to_rc=None if bt.to_rc is None else torch.from_numpy(bt.to_rc).cuda())
out, out_c = dev.alloc_output(dbt, hp.size, haps=True, onehot=False)
out.haps.fill_(0xAA)
dev.launch(dbt, out_c)
torch.cuda.synchronize()
got = out.haps.cpu().numpy().reshape(-1, L)
print("sentinel left:", int((got == 0xAA).sum()), "of", got.size, " N:", int((got == ord('N')).sum()), " expected N:", int((hp == ord('N')).sum()))
print("row 0 head exp", hp.reshape(-1, L)[0, :48].tobytes(), "\n           got", got[0, :48].tobytes())
