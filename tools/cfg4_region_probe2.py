"""tools/cfg4_region_probe2.py: what the host does at an epoch boundary of the cfg4 leg (8 batches per epoch): the calls of one region
with their host durations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_cfg4
from genvarloader_amd import loader as L_, _lib

R, S, P, L = 16, 64, 2, 131072
st, dev, ds, tracks, mean_v = bench_cfg4.build("cuda:0", R, S, P, L, seed=20260806)
dl = ds.to_dataloader(batch_size=128, shuffle=True, seed=1, in_flight=int(os.environ.get("GVL_CFG4_INFLIGHT", 3)), group=1)
log = []
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        log.append((name, (time.perf_counter() - t0) * 1e6))
        return r
    return w
lib = ds.dev.lib
import ctypes as C
lib.gvl_loader_next.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
for n in ("gvl_loader_start_epoch", "gvl_loader_prefetch_epoch", "gvl_loader_next", "gvl_loader_set_epoch", "gvl_loader_table_bytes"):
    setattr(lib, n, timed(n.replace("gvl_loader_", ""), getattr(lib, n)))
L_.epoch_order = timed("epoch_order", L_.epoch_order)
dl._epoch_table = timed("_epoch_table", dl._epoch_table)
def forever():
    while True:
        yield from dl
it = forever()
for _ in range(20): keep = next(it)
torch.cuda.synchronize()
for reg in range(3):
    torch.cuda.synchronize()
    del log[:]
    t0 = time.perf_counter()
    marks = []
    for i in range(20):
        n0 = len(log)
        ts = time.perf_counter()
        keep = next(it)
        marks.append(((time.perf_counter() - ts) * 1e6, log[n0:]))
    torch.cuda.synchronize()
    print(f"region {reg}: {(time.perf_counter() - t0) * 1e6:.0f} us")
    for d, calls in marks:
        print(f"  step {d:6.1f} us: " + "  ".join(f"{n} {u:.1f}" for n, u in calls))
