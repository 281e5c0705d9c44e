"""tools/cfg4_region_probe.py: where a K-step region of the cfg4 leg spends its host time -- per-step host clock deltas of a few regions
(the leg's step at K = 20 reads ~ 9 us per step more than at K = 100: ~ 230 us per region)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_cfg4

R, S, P, L = 16, 64, 2, 131072
st, dev, ds, tracks, mean_v = bench_cfg4.build("cuda:0", R, S, P, L, seed=20260806)
dl = ds.to_dataloader(batch_size=128, shuffle=True, seed=1, in_flight=int(os.environ.get("GVL_CFG4_INFLIGHT", 3)), group=1)
def forever():
    while True:
        yield from dl
it = forever()
keep = None
for _ in range(10): keep = next(it)
torch.cuda.synchronize()
for K in (20, 100):
    spans = []
    stamps = None
    for reg in range(60):
        torch.cuda.synchronize()
        ts = [time.perf_counter()]
        for i in range(K):
            keep = next(it)
            ts.append(time.perf_counter())
        torch.cuda.synchronize()
        ts.append(time.perf_counter())
        spans.append(ts[-1] - ts[0])
        if reg == 30: stamps = ts
    d = np.diff(np.array(stamps)) * 1e6
    print(f"K {K}: median region {np.median(spans) * 1e6:.1f} us = {np.median(spans) * 1e6 / K:.2f} us per step; one region's host deltas (us): "
          + " ".join(f"{x:.0f}" for x in d[:24]) + (" ..." if K > 24 else "") + f" | final sync {d[-1]:.0f}")
# the same with an event per step on the consumer's stream: when each batch is READY on the GPU
K = 20
evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
for reg in range(3):
    torch.cuda.synchronize()
    evs[0].record()
    for i in range(K):
        keep = next(it)
        evs[i + 1].record()
    torch.cuda.synchronize()
    print("K 20 GPU-side ready times (us since region start):", " ".join(f"{evs[0].elapsed_time(e) * 1e3:.0f}" for e in evs[1:]))
