"""PCIe-inclusive rate of the numpy drop-in layer (genvarloader_amd.ffi): host arrays in, host arrays out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import synth
import genvarloader_amd.ffi as ffi

st, bt = synth.make_config("cfg3")
K, L = bt.n_windows, bt.output_length
if "--readonly" in sys.argv:      # what the reference passes: read-only memmaps -> fingerprinted once per object
    for a in (bt.geno_offsets, bt.geno_v_idxs, st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets):
        a.flags.writeable = False
args = (bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs, st.v_starts, st.ilens, st.alt_alleles,
        st.alt_offsets, st.ref, st.ref_offsets, np.uint8(st.pad_char), np.int64(L), None, None, bt.to_rc, True)
for name, fn, nbytes in (("reconstruct_haplotypes_fused (u8 haplotypes to host)", lambda: ffi.reconstruct_haplotypes_fused(*args), K * L),
                         ("reconstruct_haplotypes_fused_onehot (one-hot to host)", lambda: ffi.reconstruct_haplotypes_fused_onehot(*args), 4 * K * L)):
    for _ in range(3): fn()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    print(f"{name}: {med * 1e3:.2f} ms per {K}-window batch = {K / med / 1e6:.2f} M windows/s ({nbytes / med / 1e9:.1f} GB/s of result over PCIe)")

# round 6: the SVAR2 drop-in entry (decoded channels in, haplotype bytes out): merge + reconstruct + D2H per call
from genvarloader_amd import svar2
rng = np.random.default_rng(5)
sv = synth.to_svar2(rng, st, bt, dense_af=0.3)
sargs = (bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, np.uint8(st.pad_char), np.int64(L))
for name, fn, nbytes in (("reconstruct_haplotypes_from_svar2 (decoded channels in, u8 haplotypes to host)", lambda: svar2.reconstruct_haplotypes_from_svar2(*sargs), K * L),):
    for _ in range(3): fn()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    print(f"{name}: {med * 1e3:.2f} ms per {K}-window batch = {K / med / 1e6:.2f} M windows/s ({nbytes / med / 1e9:.1f} GB/s of result over PCIe)")
