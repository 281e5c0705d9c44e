import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench_cfg4
from genvarloader_amd import _lib
R, S, P, L, bs = 16, 64, 2, 131072, 128
st, dev, ds, tracks, mean_v = bench_cfg4.build("cuda:0", R, S, P, L)
lib = _lib.load()
K = bs * P
order = np.random.default_rng(1).permutation(R * S)
reqs = [ds.request(order[i:i + bs].astype(np.int64)) for i in range(0, len(order), bs)]
bts0 = [dev.prepare_batch(r[1], r[2], r[3], L, to_rc=r[4]) for r in reqs]
plans = [dev.hap_plan(b) for b in bts0]
bts = [dev.prepare_batch(r[1], r[2], r[3], L, to_rc=r[4], hap_plan=pl) for r, pl in zip(reqs, plans)]
def timeit(out_c, flags=-1, n=12):
    lib.gvl_set_debug_flags(flags)
    for i in range(3): dev.launch(bts[i % len(bts)], out_c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): dev.launch(bts[i % len(bts)], out_c)
    e1.record(); torch.cuda.synchronize(); lib.gvl_set_debug_flags(-1)
    return e0.elapsed_time(e1) / n * 1e3
import os
if os.environ.get("SUB"): _lib.set_tuning(_lib.TUNE_LEAN_SUB, int(os.environ["SUB"]))
for name, kw in (("haps only", dict(haps=True, onehot=False)), ("haps+annot", dict(haps=True, onehot=False, annotate=True)),
                 ("onehot+haps", dict(haps=True, onehot=True)), ("onehot+haps+annot", dict(haps=True, onehot=True, annotate=True))):
    out, out_c = dev.alloc_output(bts[0], K * L, **kw)
    print(f"{name:20s} lean with plans {timeit(out_c):7.1f} us   no-plans {timeit(out_c, 536870912):7.1f}   all-purpose (r05 routing / no-lean-long) {timeit(out_c, 1073741824 | 1048576):7.1f} us")
    del out, out_c
