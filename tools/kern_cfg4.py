"""tools/kern_cfg4.py [flags...]: the cfg4 haplotype kernel alone (256 rows x 131072 bp, one-hot + bytes) down several GVL_DBG
paths -- HIP events around back-to-back launches on one stream, and with 3 streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import bench_cfg4
from genvarloader_amd import _lib

R, S, P, L = 16, 64, 2, 131072
st, dev, ds, tracks, mean_v = bench_cfg4.build("cuda:0", R, S, P, L)
bs = 128
order = np.random.default_rng(1).permutation(R * S)
lib = _lib.load()
cur = torch.cuda.current_stream()
slots = []
for b in range(3):
    idx = torch.from_numpy(order[b * bs:(b + 1) * bs].astype(np.int64)).cuda()
    idx0, reg, sh, goi, rc = ds.request(idx)
    dbt = dev.prepare_batch(reg, sh, goi, L, to_rc=rc)
    slots.append((dbt, dev.alloc_output(dbt, 2 * bs * L, haps=True, onehot=True), dev.alloc_output(dbt, 2 * bs * L, haps=False, onehot=True)))
streams = [torch.cuda.Stream() for _ in range(3)]
hap_bytes = (L * 6 + 28.0 * mean_v + 61.0) * 2 * bs


def timeit(fn, n=30):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for i in range(n):
        fn(i)
    e1.record(cur); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def inflight(which, n=60, warm=True):
    if warm:
        inflight(which, 6, False)
    torch.cuda.synchronize()
    e0 = [torch.cuda.Event(enable_timing=True) for _ in streams]; e1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    for s, a in zip(streams, e0):
        a.record(s)
    for i in range(n):
        dev.launch(slots[i % 3][0], slots[i % 3][which][1], stream=streams[i % 3])
    for s, a in zip(streams, e1):
        a.record(s)
    torch.cuda.synchronize()
    return max(a.elapsed_time(b) for a in e0 for b in e1) / n * 1e3


for flags in [int(x) for x in sys.argv[1:]] or [0]:
    lib.gvl_set_debug_flags(flags)
    t_both = timeit(lambda i: dev.launch(slots[i % 3][0], slots[i % 3][1][1]))
    t_oh = timeit(lambda i: dev.launch(slots[i % 3][0], slots[i % 3][2][1]))
    print(f"GVL_DBG={flags}: one-hot + bytes {t_both:.2f} us alone ({hap_bytes / (t_both * 1e-6) / 1e9 / 8000:.3f} of 8 TB/s), {inflight(1):.2f} us with 3 streams;"
          f"  one-hot only {t_oh:.2f} us alone, {inflight(2):.2f} with 3 streams", flush=True)
