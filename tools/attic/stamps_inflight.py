"""Phase timeline of ONE stamped launch while other launches are in flight on the other streams
(the regime bench.py's `value` is measured in).  usage: stamps_inflight.py [scale] [rotate] [streams] [dbg]"""
import ctypes as C, os, sys
os.environ["GVL_HIP_LIB"] = os.environ.get("GVL_DIAG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgvl_hip_diag.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth, _lib
scale = sys.argv[1] if len(sys.argv) > 1 else "hg38"
rot = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n_streams = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dbg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
wl = "cfg3"
ds = synth.make_genome(scale, wl, device="cuda")
dev = HapsDevice(**ds.static_kwargs())
lib = _lib.load(); lib.gvl_set_debug_flags(dbg)
K, L = synth.CONFIGS[wl]["windows"], ds.length
qs = ds.draw_batches(rot, K // 2, seed=3)
bts = []
for q in qs:
    r = ds.request(q, rc=True)
    bts.append(dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], L, to_rc=r["to_rc"]))
outs = [dev.alloc_output(bts[0], K * L, haps=False, onehot=True) for _ in range(n_streams + 1)]
streams = [torch.cuda.Stream() for _ in range(n_streams)]
nwg = (K + 7) // 8
stamps = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
acc = []
i = 0
for rep in range(7):
    stamps.zero_(); stamps.view(nwg, 16)[:, 12] = 1 << 62; stamps.view(nwg, 16)[:, 14] = 1 << 62
    torch.cuda.synchronize()
    for step in range(120):
        s = streams[step % n_streams]
        if step == 80:
            lib.gvl_diag_set_stamps(C.c_void_p(stamps.data_ptr()))
        dev.launch(bts[i % rot], outs[step % n_streams][1], stream=s); i += 1
        if step == 80:
            lib.gvl_diag_set_stamps(None)
    torch.cuda.synchronize()
    acc.append(stamps.cpu().numpy().reshape(nwg, 16).astype(np.float64) * 10.0)
names = ["start", "P1 done", "sync", "records+classify", "plan", "descriptors", "passA issued", "passG done", "end(wave0)"]
print(f"== {scale} rotate={rot} streams={n_streams} dbg={dbg}: one stamped launch among launches in flight; per workgroup, ns from ITS OWN start (median / p90)")
rel = np.concatenate([s[:, :9] - s[:, :1] for s in acc])
for j, n in enumerate(names):
    if j: print(f"  {n:28s} {np.median(rel[:, j]):8.0f} {np.percentile(rel[:, j], 90):8.0f}")
lw = np.concatenate([s[:, 11] - s[:, 0] for s in acc]); ew = np.concatenate([s[:, 12] - s[:, 0] for s in acc])
print(f"  {'last wave of the workgroup':28s} {np.median(lw):8.0f} {np.percentile(lw, 90):8.0f}")
print(f"  {'first wave of the workgroup':28s} {np.median(ew):8.0f} {np.percentile(ew, 90):8.0f}")
if dbg & 256:
    ra = np.concatenate([s[:, 15] - s[:, 0] for s in acc]); ra = ra[ra > 0]
    print(f"  {'wave 1: reference bytes back':28s} {np.median(ra):8.0f} {np.percentile(ra, 90):8.0f}   (diagnostic wait, GVL_DBG 256)")
span = np.array([s[:, 11].max() - s[:, 0].min() for s in acc]); spread = np.array([s[:, 0].max() - s[:, 0].min() for s in acc])
print(f"  launch span (first start -> last end) median {np.median(span):.0f} ns; workgroup starts spread over {np.median(spread):.0f} ns")
