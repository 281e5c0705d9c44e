"""How many of the cfg4 track kernel's chunk-waves leave the common path (counters behind gvl_diag_set_stamps): scalar walk,
no window, more than 4 entries, trips on the per-lane path, positions looked up in the interval list itself."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from genvarloader_amd import _lib
from tools import bench_cfg4

lib = _lib.load()
lib.gvl_diag_set_stamps.argtypes = [C.c_void_p]
lib.gvl_diag_set_stamps.restype = None
R, S = int(os.environ.get("GVL_CFG4_R", 16)), int(os.environ.get("GVL_CFG4_S", 64))
st, dev, ds, tracks, mean_v = bench_cfg4.build("cuda:0", R, S, 2, 131072)
stamps = torch.zeros(64, dtype=torch.int64, device="cuda")
dl = ds.to_dataloader(batch_size=128, shuffle=True, seed=1, in_flight=1, group=1)
lib.gvl_diag_set_stamps(C.c_void_p(stamps.data_ptr()))
n = 0
for b in dl:
    n += 1
torch.cuda.synchronize()
lib.gvl_diag_set_stamps(None)
c = stamps.cpu().numpy()
names = {8: "chunk-waves", 9: "scalar walk", 10: "no window", 11: "more than 4 entries", 12: "entries (sum)", 13: "trips on the per-lane path",
         14: "positions looked up in the list", 15: "entries read from the row's plan"}
print(f"{n} batches, V/row {mean_v:.0f}")
for k, v in names.items():
    print(f"  {v:36s} {int(c[k]):10d}  per chunk-wave {c[k] / max(c[8], 1):8.4f}")
