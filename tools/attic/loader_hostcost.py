"""Where the native loader's 15 us of host time per batch go: the C call vs the Python around it."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth, _lib
from genvarloader_amd.loader import DeviceHapsDataset

R, S, P, L, bs = 200, 2504, 2, 2048, 2048
rng = np.random.default_rng(20260802 + 5)
st = synth.make_static(rng, (64 << 20,), indel_frac=0.15)
full_regions, go, gv = synth.make_grid(rng, st, R, S, P, L)
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv, pad_char=st.pad_char)
ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L)
dl = ds.to_dataloader(batch_size=bs, shuffle=True, in_flight=3)
for _ in dl: pass
torch.cuda.synchronize()
nat = dl._native
lib, handle, out = dev.lib, nat["handle"], nat["out"]
order = torch.randperm(len(ds), device="cuda")
cur = torch.cuda.current_stream()
_lib.check(lib.gvl_loader_start_epoch(handle, C.c_void_p(order.data_ptr()), C.c_int64(len(ds)), C.c_int32(0), C.c_void_p(cur.cuda_stream)))
sp = C.c_void_p(cur.cuda_stream); ref_out = C.byref(out)
t0 = time.perf_counter(); n = 0
while True:
    lib.gvl_loader_next(handle, sp, ref_out)
    if out.slot < 0: break
    n += 1
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"bare C loop: {1e6 * (t1 - t0) / n:.2f} us per gvl_loader_next (host), {1e6 * (t2 - t0) / n:.2f} us per batch incl. drain")
t0 = time.perf_counter(); n = 0
for b in dl: n += 1
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"python iterator: {1e6 * (t1 - t0) / n:.2f} us per batch (host), {1e6 * (t2 - t0) / n:.2f} us per batch incl. drain")
