"""tools/cfg4_touch.py: is the cold-input penalty of config 4's haplotype kernel (35 -> 40 us when the dataset no longer fits the Infinity
Cache) the latency of its plan reads?  Rotating batches of a 256-sample dataset with their chunk plans made ahead; the kernel's duration per
launch (HIP events, one stream) (a) as is, (b) with the NEXT batch's plan read by a side-stream reduction while this batch's kernel runs
(= resident in the Infinity Cache when its kernel starts), (c) the same batch over and over (everything resident)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_cfg4
from genvarloader_amd import _lib, device as gdev

S = int(os.environ.get("GVL_CFG4_S", 256))
R, P, L = 16, 2, 131072
st, dev, ds, tracks, mean_v = bench_cfg4.build("cuda:0", R, S, P, L, seed=20260806)
lib = _lib.load()
bs = 128
order = np.random.default_rng(1).permutation(R * S)
nb = min(len(order) // bs, int(os.environ.get("NB", 32)))
batches = []
for i in range(nb):
    idx = torch.from_numpy(order[i * bs:(i + 1) * bs].astype(np.int64)).cuda()
    _, reg, sh, goi, rc = ds.request(idx)
    dbt = dev.prepare_batch(reg, sh, goi, L, to_rc=rc)
    plan = dev.hap_plan(dbt)
    batches.append((dev.prepare_batch(reg, sh, goi, L, to_rc=rc, hap_plan=plan), plan))
K = 2 * bs
slots = [dev.alloc_output(batches[0][0], K * L, haps=True, onehot=True) for _ in range(2)]
cur = torch.cuda.current_stream()
side = torch.cuda.Stream()
sp = gdev._stream_ptr()
flush_a = torch.empty(768 << 20, dtype=torch.uint8, device="cuda"); flush_b = torch.empty_like(flush_a)
def run(mode, passes=4):
    evs = []
    sink = torch.zeros((), dtype=torch.int64, device="cuda")
    for p in range(passes):
        for b in range(nb):
            bb = 0 if mode == "same" else b
            dbt, plan = batches[bb]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if mode == "touch":
                nxt = batches[(b + 1) % nb][1]
                side.wait_stream(cur)                 # (starts when this batch's kernel is queued)
            if mode == "flush":                       # (768 MB copied: nothing of the inputs is left in the Infinity Cache or the L2s)
                flush_b.copy_(flush_a)
            e0.record(cur)
            _lib.check(lib.gvl_reconstruct(C.byref(dev.c), C.byref(dbt.c), C.byref(slots[b & 1][1]), sp))
            e1.record(cur)
            if mode == "touch":
                with torch.cuda.stream(side):
                    sink += nxt.view(torch.int64).sum()
                cur.wait_stream(side)                 # (the next kernel starts behind the touch)
            if p > 0: evs.append((e0, e1))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
    print(f"S={S} {nb} rotating batches, GVL_DBG={os.environ.get('GVL_DBG', '0')}, {mode:6s}: haplotype kernel {np.median(t):.2f} us median, {t.mean():.2f} mean")
for mode in ("cold", "flush", "same", "flush"):
    run(mode)
