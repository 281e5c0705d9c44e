"""Ragged mode (output_length = -1, the reference's default `ds[r, s]` shape): offsets pass + host sync + kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth

st, bt = synth.make_config("cfg3")
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
reg, sh, goi, rc = (torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (bt.regions, bt.shifts, bt.geno_offset_idx, bt.to_rc.view(np.uint8)))
for L in (bt.output_length, -1):
    for _ in range(5): out = dev.reconstruct(reg, sh, goi, L, to_rc=rc, haps=False, onehot=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); out = dev.reconstruct(reg, sh, goi, L, to_rc=rc, haps=False, onehot=True); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"output_length={L}: {np.median(ts) * 1e6:.1f} us per batch end to end (device-resident request, sync after each)")
