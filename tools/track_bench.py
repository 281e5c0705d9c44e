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

# painting: intervals of width ~Geometric(1/25) with gaps, value U(0,8), one list per query
nq = B
widths = rng.geometric(1 / 25, size=(nq, int(tlen.max()) // 20)).astype(np.int64)
gaps = rng.geometric(1 / 8, size=widths.shape).astype(np.int64)
its, ite, itv, ito = [], [], [], [0]
for q in range(nq):
    s0 = np.cumsum(widths[q] + gaps[q]) - widths[q] + int(bt.regions[q, 1]) - 100
    e0 = s0 + widths[q]
    m = s0 < int(bt.regions[q, 1]) + int(tlen[q])
    its.append(s0[m]); ite.append(e0[m]); itv.append((rng.random(int(m.sum())) * 8).astype(np.float32)); ito.append(ito[-1] + int(m.sum()))
its = torch.from_numpy(np.concatenate(its).astype(np.int32)).cuda(); ite = torch.from_numpy(np.concatenate(ite).astype(np.int32)).cuda()
itv = torch.from_numpy(np.concatenate(itv)).cuda(); ito = torch.from_numpy(np.asarray(ito, np.int64)).cuda()
oi = torch.arange(nq, dtype=torch.int64, device="cuda"); qs = torch.from_numpy(np.ascontiguousarray(bt.regions[:, 1])).cuda()
paint_out = torch.empty(int(track_offsets[-1]), dtype=torch.float32, device="cuda")
lib = _lib.load()
pmax = device.intervals_prefix_max(ite, ito)
use_pmax = True
def run_paint():
    _lib.check(lib.gvl_intervals_to_tracks(device._ptr(oi), device._ptr(qs), C.c_int64(1), C.c_int64(nq), device._ptr(its), device._ptr(ite),
                                           device._ptr(itv), device._ptr(ito), C.c_int64(int(its.numel())),
                                           device._ptr(pmax) if use_pmax else None, device._ptr(paint_out), device._ptr(d_toff),
                                           C.c_int64(int(tlen.max())), device._stream_ptr()))
for _ in range(3): run_paint()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run_paint()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
nb = int(track_offsets[-1]) * 4 + int(its.numel()) * 12
use_pmax = False
run_paint(); torch.cuda.synchronize()
e0.record()
for _ in range(20): run_paint()
e1.record(); torch.cuda.synchronize()
print(f"painting, prefix maxima built per call: {e0.elapsed_time(e1) / 20 * 1000:.1f} us/batch")
print(f"painting: {ms * 1000:.1f} us/batch  ~{nb / ms / 1e6:.0f} GB/s algorithmic  ({nq} queries, {int(its.numel())} intervals, {int(track_offsets[-1])} values)")
