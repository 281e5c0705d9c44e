"""tools/ref_long.py: gvl_get_reference on config-4-shaped rows (256 x 131 072 bases, half of them reverse-complemented; bytes + one-hot):
the chunked lean kernel's route against the all-purpose kernel (GVL_DBG 2^30), HIP events around back-to-back calls."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvarloader_amd import synth, _lib, device as gdev, ffi

rng = np.random.default_rng(5)
K, L = 256, 131072
ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), 256 << 20)
ref_offsets = np.array([0, ref.size], np.int64)
dev = ffi._ref_static(ref, ref_offsets, ord("N"))
lib = _lib.load()
starts = rng.integers(0, ref.size - L, K).astype(np.int32)
reg = np.stack([np.zeros(K, np.int32), starts, starts + L, np.where(rng.random(K) < 0.5, -1, 1).astype(np.int32)], 1)
oo = (np.arange(K + 1, dtype=np.int64) * L)
to_rc = reg[:, 3] == -1
d = "cuda"
regd, ood, rcd = torch.from_numpy(reg).to(d), torch.from_numpy(oo).to(d), torch.from_numpy(to_rc.astype(np.uint8)).to(d)
out = torch.empty(K * L, dtype=torch.uint8, device=d); oh = torch.empty((K * L, 4), dtype=torch.uint8, device=d)
cur = torch.cuda.current_stream()
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for _ in range(n): fn()
    e1.record(cur); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
args = (C.byref(dev.c), gdev._ptr(regd), C.c_int64(4), C.c_int64(K), gdev._ptr(ood), C.c_int64(L), gdev._ptr(rcd))
res = {}
for name, flags in (("chunked lean kernel", 0), ("all-purpose kernel (GVL_DBG 2^30)", 1073741824)):
    lib.gvl_set_debug_flags(flags)
    for what, o, h in (("bytes + one-hot", out, oh), ("one-hot", None, oh), ("bytes", out, None)):
        a = args + (gdev._ptr(o), gdev._ptr(h), gdev._stream_ptr())
        t = timeit(lambda: _lib.check(lib.gvl_get_reference(*a)))
        nbytes = K * L * (1 + (1 if o is not None else 0) + (4 if h is not None else 0))
        print(f"{name:36s} {what:16s}: {t:7.1f} us = {nbytes / t / 1e6:.2f} TB/s algorithmic ({nbytes / t / 1e6 / 8:.2f} of 8 TB/s)")
        res[(flags, what)] = (out.clone() if o is not None else None, oh.clone() if h is not None else None)
lib.gvl_set_debug_flags(-1)
for what in ("bytes + one-hot", "one-hot", "bytes"):
    a, b = res[(0, what)], res[(1073741824, what)]
    for x, y in zip(a, b):
        if x is not None: assert torch.equal(x, y), what
print("both routes: identical outputs")
