"""cfg4 timing: haplotypes (+one-hot) and one realigned track, device-resident (not part of the product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, device, synth, _lib
import ctypes as C

st, bt = synth.make_config("cfg4", contig=32 << 20)
B, P = bt.geno_offset_idx.shape
L = bt.output_length
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
rng = np.random.default_rng(5)
tlen = (bt.regions[:, 2] - bt.regions[:, 1]).astype(np.int64) + 4096
track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
tracks = np.repeat(rng.random(int(track_offsets[-1]) // 25 + 1).astype(np.float32) * 8, 25)[: int(track_offsets[-1])]
out_offsets = np.arange(B * P + 1, dtype=np.int64) * L
dtracks = torch.from_numpy(tracks).cuda()
d_off = torch.from_numpy(out_offsets).cuda(); d_toff = torch.from_numpy(track_offsets).cuda()
dbt = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=bt.to_rc)
out = torch.empty(B * P * L, dtype=torch.float32, device="cuda")
strategy = int(sys.argv[1]) if len(sys.argv) > 1 else 0

tbt = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, -1, None, None, None, out_offsets)
par = (C.c_double * 1)(0.0)
def run_tracks():
    _lib.check(dev.lib.gvl_realign_tracks(C.byref(dev.c), C.byref(tbt.c), device._ptr(dtracks), device._ptr(d_toff), par,
                                          C.c_int64(strategy), C.c_uint64(0), device._ptr(out), device._stream_ptr()))

slot = dev.alloc_output(dbt, B * P * L, haps=True, onehot=True)
def run_haps():
    dev.launch(dbt, slot[1])

for name, fn, nbytes in (("tracks", run_tracks, B * P * L * 8), ("haps+onehot", run_haps, B * P * L * 6)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name}: {ms * 1000:.1f} us/batch  ~{nbytes / ms / 1e6:.0f} GB/s algorithmic  ({B * P} rows x {L})")
