"""tools/dbg_lean.py [flags...]: the lean kernel against the all-purpose kernel on one small batch (one-hot only)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from genvarloader_amd import HapsDevice, synth, _lib

lib = _lib.load()
rng = np.random.default_rng(7)
st = synth.make_static(rng, (300_000,), indel_frac=0.15)
bt = synth.make_batch(rng, st, 64, 2, 2048, rc_frac=0.5, random_shifts=True, edge_frac=0.1)
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
def run(flags):
    lib.gvl_set_debug_flags(flags)
    out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length, to_rc=bt.to_rc, haps=False, onehot=True)
    torch.cuda.synchronize()
    return out.onehot.cpu().numpy()
ref = run(16384)
for f in [int(x) for x in sys.argv[1:]] or [0, 32768]:
    print("flags", f, flush=True)
    got = run(f)
    bad = np.nonzero((got != ref).any(axis=1))[0]
    print("  mismatching bases:", bad.size, "rows:", np.unique(bad // bt.output_length)[:20])
