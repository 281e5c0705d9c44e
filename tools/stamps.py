"""Phase timeline of the planned kernel (diagnostic build: tools/libgvl_hip_diag.so)."""
import ctypes as C, os, sys
os.environ["GVL_HIP_LIB"] = os.path.join(os.path.dirname(__file__), "libgvl_hip_diag.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth, _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nwin = int(sys.argv[2]) if len(sys.argv) > 2 else None
st, bt = synth.make_config(wl, windows=nwin)
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
dbt = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length, to_rc=bt.to_rc)
out, oc = dev.alloc_output(dbt, bt.n_windows * bt.output_length, haps=False, onehot=True)
nwg = (bt.n_windows + 7) // 8
stamps = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
def reset():
    stamps.zero_(); stamps.view(nwg, 16)[:, 12] = 1 << 62; stamps.view(nwg, 16)[:, 14] = 1 << 62
reset()
lib = _lib.load()
for i in range(20): dev.launch(dbt, oc)
torch.cuda.synchronize()
lib.gvl_diag_set_stamps(C.c_void_p(stamps.data_ptr()))
dev.launch(dbt, oc); torch.cuda.synchronize()
raw = stamps.cpu().numpy().reshape(nwg, 16)
s = raw.astype(np.float64) * 10.0  # memrealtime ticks of 100 MHz -> ns
t0 = s[:, 0].min()
names = ["start", "P1 done", "sync", "P2 recs", "P3 scans", "P3b desc", "passA issued", "passG done", "end(wave0)"]
for i, n in enumerate(names):
    col = s[:, i] - t0
    print(f"{n:16s} min {col.min():8.0f}  median {np.median(col):8.0f}  max {col.max():8.0f} ns")
d = np.diff(s[:, :9], axis=1)
print("per-WG phase durations (median ns):", np.median(d, axis=0).round(0))

nfb = raw[:, 9]
ng = raw[:, 10]
print("rows on the scalar path:", int(nfb.sum()), "of", bt.n_windows)
end = s[:, 8] - s[:, 0]
for g in sorted(set(ng.tolist())):
    m = ng == g
    print(f"wave0 general trips={g}: {m.sum():4d} WGs  passG median {np.median(d[m, 6]):7.0f} ns  total median {np.median(end[m]):7.0f} max {end[m].max():7.0f}")

for name, col in (("plan ready, earliest wave", 14), ("plan ready, latest wave", 13), ("end, earliest wave", 12), ("end, latest wave", 11)):
    c = s[:, col] - t0
    print(f"{name:28s} min {c.min():8.0f}  p10 {np.percentile(c,10):8.0f}  median {np.median(c):8.0f}  p90 {np.percentile(c,90):8.0f}  max {c.max():8.0f} ns")
