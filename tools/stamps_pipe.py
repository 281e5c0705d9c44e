"""Timeline of recon_lean_rows_kernel (diagnostic build tools/libgvl_hip_diag.so, -DGVL_DIAG: PIPE_STAMP in csrc/gvl_lean_pipe.inc): where a
wave's time goes per row, how many waves are alive, for K rows of L bases (fixed length, one-hot).
python tools/stamps_pipe.py [L] [rows] [GVL_TUNE_PIPE_ROWS_X100]"""
import ctypes as C, os, sys
os.environ["GVL_HIP_LIB"] = os.environ.get("GVL_DIAG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgvl_hip_diag.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth, _lib

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
x100 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = _lib.load()
if x100:
    _lib.set_tuning(_lib.TUNE_PIPE_ROWS_X100, x100)
rng = np.random.default_rng(5)
st = synth.make_static(rng, (64 << 20,), indel_frac=0.15)
bt = synth.make_batch(rng, st, K // 2, 2, L, rc_frac=0.5, slack=16)
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
b = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=bt.to_rc)
out, oc = dev.alloc_output(b, K * L, haps=False, onehot=True)
for _ in range(5):
    dev.launch(b, oc)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    dev.launch(b, oc)
e1.record(); torch.cuda.synchronize()
print(f"L {L}, {K} rows, rows per wave x100 = {x100 or 'built in'}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch without stamps")
stamps = torch.zeros(64 + K * 16, dtype=torch.int64, device="cuda")
stamps[63] = 1
torch.cuda.synchronize()
lib.gvl_diag_set_stamps(C.c_void_p(stamps.data_ptr()))
e0.record(); dev.launch(b, oc); e1.record(); torch.cuda.synchronize()
lib.gvl_diag_set_stamps(None)
print(f"the stamped launch: {e0.elapsed_time(e1) * 1e3:.1f} us")
s = stamps.cpu().numpy()[64:].reshape(K, 16).astype(np.float64) * 0.01          # 100 MHz -> us
it = s[:, 10] / 0.01
t_min = s[:, 3][s[:, 3] > 0].min()
first = it == 1
t0 = s[first, 0].min()
names = {(0, 1): "wave start -> tables built (prologue, barrier)", (1, 2): "-> first window landed (entries, then window + slot line)",
         (3, 4): "iteration top -> records decoded (incl. issuing the next row's prefetch)", (4, 5): "plan", (5, 6): "phase A (+ fence)",
         (6, 7): "phase B", (7, 8): "phase C: stores issued", (8, 9): "closing wait (next row's window)"}
for cls, m in (("first row of a wave", first), ("second row", it == 2), ("third row and later", it >= 3)):
    if not m.any():
        continue
    print(f"-- {cls}: {int(m.sum())} rows; medians (p10 .. p90), us")
    for (a, c), nm in names.items():
        ok = m & (s[:, a] > 0) & (s[:, c] > 0)
        if (a, c) in ((0, 1), (1, 2)) and cls != "first row of a wave":
            continue
        if ok.any():
            d = s[ok, c] - s[ok, a]
            print(f"   {nm:75s} {np.median(d):6.2f}  ({np.percentile(d, 10):.2f} .. {np.percentile(d, 90):.2f})   n = {int(ok.sum())}")
# waves: start = stamp 0 of the first row, end = stamp 8 of the wave's last row
n_waves = int(first.sum())
k = np.arange(K)
w_of = k % n_waves if n_waves else k
end = np.zeros(n_waves); start = np.zeros(n_waves)
np.maximum.at(end, w_of, s[:, 8])
start[w_of[first]] = s[first, 0]
life = end - start
print(f"-- {n_waves} waves: lifetime median {np.median(life):.2f} us (p10 {np.percentile(life, 10):.2f}, p90 {np.percentile(life, 90):.2f}); "
      f"first start -> last end {end.max() - start[start > 0].min():.1f} us")
span0, span1 = start[start > 0].min(), end.max()
for f in (0.1, 0.25, 0.5, 0.75, 0.9):
    t = span0 + f * (span1 - span0)
    print(f"   waves alive at {f:.2f} of the span: {int(((start <= t) & (end >= t)).sum())}")
started = np.sort(start[start > 0]) - span0
print("   waves started by us: " + ", ".join(f"{t:.0f}: {int((started <= t).sum())}" for t in (1, 2, 5, 10, 20, 30, 40, 50)))
