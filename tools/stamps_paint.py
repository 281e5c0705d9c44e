"""Timeline of realign_paint_kernel (diagnostic build tools/libgvl_hip_diag.so: PAINT_STAMP in csrc/gvl_tracks_kernels.inc) on BASELINE
config 4's batch (256 rows x 131 072 values, a wave per 2048-value chunk): where a chunk-wave's time goes, how many are alive.
python tools/stamps_paint.py"""
import ctypes as C, os, sys
os.environ["GVL_HIP_LIB"] = os.environ.get("GVL_DIAG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgvl_hip_diag.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_cfg4
from genvarloader_amd import _lib, device as gdev
from genvarloader_amd._lib import GvlBatch

R, S, P, L = 16, 64, 2, 131072
st, dev, ds, tracks, mean_v = bench_cfg4.build("cuda:0", R, S, P, L)
bs = 128
order = np.random.default_rng(1).permutation(R * S)
lib = _lib.load()
idx = torch.from_numpy(order[:bs].astype(np.int64)).cuda()
idx0, reg, sh, goi, rc = ds.request(idx)
K = 2 * bs
n_scr = int(lib.gvl_tracks_scratch_bytes(C.c_int64(bs), C.c_int64(P), C.c_int64(ds._stride)))
arena = torch.empty(((4 * K * L + 255) & ~255) + n_scr, dtype=torch.uint8, device="cuda")
gbt = GvlBatch(regions=reg.data_ptr(), regions_stride=4, shifts=sh.data_ptr(), geno_offset_idx=goi.data_ptr(), batch=bs, ploidy=P,
               keep=None, keep_offsets=None, to_rc=None if rc is None else rc.data_ptr(), output_length=L, out_offsets=None, max_row_len=L)
par = (C.c_double * 1)(0.0)

def call():
    _lib.check(lib.gvl_tracks_batch(C.byref(dev.c), C.byref(gbt), C.c_void_p(idx0.data_ptr()), ds._track_sets, C.c_int32(1), par, C.c_int64(0),
                                    C.c_uint64(0), C.c_void_p(arena.data_ptr()), C.c_int64(K * L),
                                    C.c_void_p(arena.data_ptr() + ((4 * K * L + 255) & ~255)), C.c_int64(ds._stride), gdev._stream_ptr()))

for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    call()
e1.record(); torch.cuda.synchronize()
print(f"gvl_tracks_batch, {K} rows x {L}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call without stamps")
n_chunks = L // 2048
stamps = torch.zeros(64 + K * n_chunks * 8, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
lib.gvl_diag_set_stamps(C.c_void_p(stamps.data_ptr()))
lib.gvl_set_debug_flags(4096)
e0.record(); call(); e1.record(); torch.cuda.synchronize()
lib.gvl_set_debug_flags(-1)
lib.gvl_diag_set_stamps(None)
print(f"the stamped call: {e0.elapsed_time(e1) * 1e3:.1f} us")
s = stamps.cpu().numpy()[64:].reshape(K * n_chunks, 8).astype(np.float64) * 0.01
ok = s[:, 5] > 0
print(f"chunk-waves on the fast path: {int(ok.sum())} of {K * n_chunks}")
for (a_, c_), nm in (((0, 3), "start -> window built (head, entries, list bounds, bucket bounds, candidates, LDS image)"), ((3, 5), "-> stores issued (checks, the chunk's 8 KB)")):
    d = s[ok, c_] - s[ok, a_]
    print(f"   {nm:100s} {np.median(d):6.2f}  ({np.percentile(d, 10):.2f} .. {np.percentile(d, 90):.2f})")
life = s[ok, 5] - s[ok, 0]
t0, t1 = s[ok, 0].min(), s[ok, 5].max()
print(f"   a chunk-wave's lifetime: median {np.median(life):.2f} us (p10 {np.percentile(life, 10):.2f}, p90 {np.percentile(life, 90):.2f}); first start -> last end {t1 - t0:.1f} us")
for f in (0.05, 0.1, 0.25, 0.5, 0.75, 0.9):
    t = t0 + f * (t1 - t0)
    print(f"   waves alive at {f:.2f} of the span: {int(((s[ok, 0] <= t) & (s[ok, 5] >= t)).sum())}   storing: {int(((s[ok, 3] <= t) & (s[ok, 5] >= t)).sum())}")
st_ = np.sort(s[ok, 0]) - t0
print("   waves started by us: " + ", ".join(f"{t}: {int((st_ <= t).sum())}" for t in (1, 2, 5, 10, 15, 20, 25, 30)))
