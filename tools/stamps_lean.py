"""Phase timeline of recon_lean_kernel per row class (diagnostic build: tools/libgvl_hip_diag.so, -DGVL_DIAG).
usage: stamps_lean.py [cfg3|cfg2] [hg38|small] [rotate] [dbg]   -- stamps of ONE launch (alone on the device) after the others have run"""
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
stamps = torch.zeros(K * 16, dtype=torch.int64, device="cuda")
names = ["start", "request entries", "contig bounds + barrier", "records + window back", "plan", "row in LDS (A)", "patched (B)", "stores done"]
sub = {11: "A: window parked", 12: "A: runs per lane", 13: "A: fence", 14: "A: picked"}
acc = []
for rep in range(5):
    for i in range(rot):
        dev.launch(bts[i], outs[i % 3][1])
    torch.cuda.synchronize()
    stamps.zero_(); torch.cuda.synchronize()
    lib.gvl_diag_set_stamps(C.c_void_p(stamps.data_ptr()))
    dev.launch(bts[(rep * 7) % rot] if rot > 1 else bts[0], outs[0][1]); torch.cuda.synchronize()
    lib.gvl_diag_set_stamps(None)
    acc.append(stamps.cpu().numpy().reshape(K, 16).astype(np.float64))
# shader-clock ticks per ns: calibrated on the launch itself (first start -> last end in real time vs the longest wave)
print(f"== {wl} {scale} rotate={rot} dbg={dbg}: recon_lean_kernel, 5 stamped launches; per-phase DURATIONS inside a wave, shader-clock ticks (median over the rows of a class)")
cls_names = {0: "SNPs only (speculative window)", 1: "indels, window re-aligned in LDS", 2: "indels, runs re-read", 3: "solo (all-purpose body)"}
for c in (0, 1, 2, 3):
    cnt, rows = [], []
    for s in acc:
        m = (s[:, 8] == c) & (s[:, 0] > 0)
        cnt.append(int(m.sum()))
        if m.any():
            rows.append(s[m])
    print(f"-- {cls_names[c]}: {np.median(cnt):.0f} rows")
    if not rows:
        continue
    r = np.concatenate(rows)
    hi = 5 if c == 3 else 8
    for i in range(1, hi):
        print(f"   {names[i]:28s} +{np.median(r[:, i] - r[:, i - 1]):8.0f} ticks")
    if c == 1:
        prev = 4
        for i in (11, 12, 13, 14):
            print(f"      {sub[i]:25s} +{np.median(r[:, i] - r[:, prev]):8.0f} ticks"); prev = i
    if c != 3:
        print(f"   whole wave                   {np.median(r[:, 7] - r[:, 0]):8.0f} ticks")
    if c == 3:
        print("   why (1 window/shape, 2 slot overflow beyond 64 or no inline records, 3 odd record, 4 / 6 trailing pad, 5 no fixed point):",
              dict(zip(*[x.tolist() for x in np.unique(r[:, 9].astype(int), return_counts=True)])))
