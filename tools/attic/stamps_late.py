"""Does a workgroup that starts LATE in a launch (code already in the instruction caches, other
workgroups streaming) run its head faster than the first ones?  One launch of `mult` x 4096 rows,
stamps per workgroup relative to the workgroup's own start, grouped by start-time quartile.
usage: stamps_late.py [scale] [mult] [dbg]   (diagnostic build, see stamps.py)"""
import ctypes as C, os, sys
os.environ["GVL_HIP_LIB"] = os.environ.get("GVL_DIAG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgvl_hip_diag.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth, _lib
scale = sys.argv[1] if len(sys.argv) > 1 else "small"
mult = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dbg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
wl = "cfg3"
ds = synth.make_genome(scale, wl, device="cuda")
dev = HapsDevice(**ds.static_kwargs())
lib = _lib.load(); lib.gvl_set_debug_flags(dbg)
K, L = synth.CONFIGS[wl]["windows"] * mult, ds.length
q = ds.draw_batches(1, K // 2, seed=3)[0]
r = ds.request(q, rc=True)
bt = dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], L, to_rc=r["to_rc"])
out = dev.alloc_output(bt, K * L, haps=False, onehot=True)
nwg = (K + 7) // 8
stamps = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
for _ in range(5): dev.launch(bt, out[1])
torch.cuda.synchronize()
lib.gvl_diag_set_stamps(C.c_void_p(stamps.data_ptr()))
dev.launch(bt, out[1]); torch.cuda.synchronize()
lib.gvl_diag_set_stamps(None)
s = stamps.cpu().numpy().reshape(nwg, 16).astype(np.float64) * 10.0
t0 = s[:, 0].min()
start = s[:, 0] - t0
names = ["start", "P1 done", "sync", "records+classify", "plan", "descriptors", "passA issued", "passG done", "end(wave0)"]
order = np.argsort(start)
print(f"== {scale} x{mult} ({K} rows, {nwg} workgroups) dbg={dbg}: kernel end {(s[:, 11].max() - t0):.0f} ns")
for lo, hi in ((0, 0.1), (0.1, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)):
    sel = order[int(lo * nwg):int(hi * nwg)]
    rel = s[sel][:, :9] - s[sel][:, :1]
    print(f"  workgroups starting at {np.median(start[sel]):7.0f} ns (n={len(sel)}): " +
          "  ".join(f"{n.split()[0]} {np.median(rel[:, i]):5.0f}" for i, n in enumerate(names) if i) +
          f"  last wave end {np.median(s[sel][:, 11] - s[sel][:, 0]):5.0f}")
