"""Is the cfg4 step host-bound?  Host time of the loop that asks for batches (no synchronisation) against the time until the GPU
has finished them; GVL_CFG4_THREADED=1: the loader's producer thread submits."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools import bench_cfg4

st, dev, ds, tracks, mean_v = bench_cfg4.build("cuda:0", 16, 64, 2, 131072)
dl = ds.to_dataloader(batch_size=128, shuffle=True, seed=1, in_flight=int(os.environ.get("GVL_CFG4_INFLIGHT", 3)), group=1,
                      threaded=bool(int(os.environ.get("GVL_CFG4_THREADED", "0"))))


def forever():
    while True:
        yield from dl


it = forever()
for _ in range(30):
    b = next(it)
torch.cuda.synchronize()
for rep in range(3):
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        b = next(it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} batches: host loop {(t1 - t0) / n * 1e6:.1f} us per batch, until the GPU is done {(t2 - t0) / n * 1e6:.1f} us per batch")
