"""Phase timeline of the planned kernel (diagnostic build: tools/libgvl_hip_diag.so, -DGVL_DIAG).
usage: stamps.py [cfg3|cfg2] [hg38|small] [rotate] [dbg]   -- stamps of ONE launch after the others have run"""
import ctypes as C, os, sys
os.environ["GVL_HIP_LIB"] = os.environ.get("GVL_DIAG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgvl_hip_diag.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth, _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
scale = sys.argv[2] if len(sys.argv) > 2 else "hg38"
rot = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dbg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ds = synth.make_genome(scale, wl, device="cuda")
dev = HapsDevice(**ds.static_kwargs())
lib = _lib.load()
lib.gvl_set_debug_flags(dbg)
K, L = synth.CONFIGS[wl]["windows"], ds.length
qs = ds.draw_batches(rot, K // 2, seed=3)
bts = []
for q in qs:
    r = ds.request(q, rc=synth.CONFIGS[wl]["rc_frac"] > 0)
    bts.append(dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], L, to_rc=r["to_rc"]))
outs = [dev.alloc_output(bts[0], K * L, haps=False, onehot=True) for _ in range(3)]
nwg = (K + 7) // 8
stamps = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
acc = []
for rep in range(5):
    for i in range(rot):
        dev.launch(bts[i], outs[i % 3][1])
    torch.cuda.synchronize()
    stamps.zero_(); stamps.view(nwg, 16)[:, 12] = 1 << 62; stamps.view(nwg, 16)[:, 14] = 1 << 62
    torch.cuda.synchronize()
    lib.gvl_diag_set_stamps(C.c_void_p(stamps.data_ptr()))
    dev.launch(bts[(rep * 7) % rot] if rot > 1 else bts[0], outs[0][1]); torch.cuda.synchronize()
    lib.gvl_diag_set_stamps(None)
    raw = stamps.cpu().numpy().reshape(nwg, 16)
    acc.append(raw.astype(np.float64) * 10.0)   # memrealtime ticks of 100 MHz -> ns
names = ["start", "P1 done", "sync", "records+classify", "plan", "descriptors", "passA issued", "passG done", "end(wave0)"]
print(f"== {wl} {scale} rotate={rot} dbg={dbg}: medians over 5 stamped launches (ns from the launch's first workgroup start)")
rows = {n: [] for n in names}
extra = {k: [] for k in ("plan ready, earliest wave", "plan ready, latest wave", "end, earliest wave", "end, latest wave")}
for s in acc:
    t0 = s[:, 0].min()
    for i, n in enumerate(names):
        rows[n].append((np.median(s[:, i] - t0), (s[:, i] - t0).max()))
    for name, col in (("plan ready, earliest wave", 14), ("plan ready, latest wave", 13), ("end, earliest wave", 12), ("end, latest wave", 11)):
        c = s[:, col] - t0
        extra[name].append((np.median(c), c.max()))
for n in names:
    a = np.array(rows[n]); print(f"  {n:28s} median {np.median(a[:,0]):8.0f}   max {np.median(a[:,1]):8.0f}")
for n, v in extra.items():
    a = np.array(v); print(f"  {n:28s} median {np.median(a[:,0]):8.0f}   max {np.median(a[:,1]):8.0f}")
