"""cfg4's haplotype kernel alone (256 x 131 072, one-hot + bytes): chunks per wave (gvl_set_tuning(GVL_TUNE_LEAN_SUB)) x {chunk plans made
ahead, no plans}.  python tools/cfg4_sub.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_cfg4
from genvarloader_amd import _lib, device as gdev

R, S, P, L = int(os.environ.get("GVL_CFG4_R", 16)), int(os.environ.get("GVL_CFG4_S", 64)), 2, 131072
st, dev, ds, tracks, mean_v = bench_cfg4.build("cuda:0", R, S, P, L)
lib = _lib.load()
bs = 128
idx = torch.arange(bs, dtype=torch.int64, device="cuda") * 7 % (R * S)
idx0, reg, sh, goi, rc = ds.request(idx)
dbt = dev.prepare_batch(reg, sh, goi, L, to_rc=rc)
plan = dev.hap_plan(dbt)
dbt_p = dev.prepare_batch(reg, sh, goi, L, to_rc=rc, hap_plan=plan)
slot = dev.alloc_output(dbt, 2 * bs * L, haps=True, onehot=True)
cur = torch.cuda.current_stream()
hap_bytes = (L * 6 + 28.0 * mean_v + 61.0) * 2 * bs


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for _ in range(n):
        fn()
    e1.record(cur); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for sub in (1, 2, 3, 4, 8):
    _lib.set_tuning(_lib.TUNE_LEAN_SUB, sub)
    row = []
    for b in (dbt_p, dbt):
        args = (C.byref(dev.c), C.byref(b.c), C.byref(slot[1]), gdev._stream_ptr())
        ts = sorted(timeit(lambda: lib.gvl_reconstruct(*args)) for _ in range(3))
        row.append(ts[1])
    print(f"sub {sub}: plans {row[0]:.2f} us ({hap_bytes / row[0] / 1e3 / 8000:.3f})   no plans {row[1]:.2f} us ({hap_bytes / row[1] / 1e3 / 8000:.3f})", flush=True)
_lib.set_tuning(_lib.TUNE_LEAN_SUB, 0)
