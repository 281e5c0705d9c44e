"""Ragged LONG rows (output_length = -1, rows of ~L bases): the chunked lean kernel's ragged form against the all-purpose kernel
(GVL_DBG = 1048576) on a cfg4-shaped batch.  python tools/ragged_long_bench.py [L] [queries]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from genvarloader_amd import HapsDevice, _lib, synth

L = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(7)
st = synth.make_static(rng, (256 << 20,), indel_frac=0.15)
bt = synth.make_batch(rng, st, Q, 2, L, slack=32, rc_frac=0.5, output_length=-1)
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
lib = _lib.load()
b = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, -1, to_rc=bt.to_rc)
oo, tm, _ = dev.hap_offsets(b)
total, longest = (int(x) for x in tm.cpu().tolist())
print(f"{Q * 2} rows, {total} bases, longest row {longest}, mean variants per row {float((bt.geno_offsets[1] - bt.geno_offsets[0]).mean()):.0f}")
for flag, name in ((0, "default (chunked lean kernel, ragged form)"), (1048576, "GVL_DBG=1048576 (all-purpose kernel)")):
    lib.gvl_set_debug_flags(flag)
    for want in ((True, True), (True, False)):
        out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, -1, to_rc=bt.to_rc, haps=want[1], onehot=want[0])
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(n):
            out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, -1, to_rc=bt.to_rc, haps=want[1], onehot=want[0])
        ev1.record()
        torch.cuda.synchronize()
        by = total * ((4 if want[0] else 0) + (1 if want[1] else 0) + 1)
        ms = ev0.elapsed_time(ev1) / n
        print(f"  {name:48s} onehot={want[0]} haps={want[1]}: {ms * 1e3:8.1f} us per batch (sizing + reconstruct), {by / ms / 1e6:7.1f} GB/s algorithmic")
lib.gvl_set_debug_flags(-1)
