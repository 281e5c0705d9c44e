"""Where a ragged cfg3 batch's time goes (cold hg38-scale dataset): the sizing kernels (gvl_hap_offsets) and the reconstruct launch,
alone and with G batches per reconstruct launch (gvl_reconstruct_many: one grid).  python tools/ragged_parts.py [scale] [G]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvarloader_amd import HapsDevice, synth

scale = sys.argv[1] if len(sys.argv) > 1 else "hg38"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ds = synth.make_genome(scale, "cfg3", device="cuda:0", seed=20260805)
dev = HapsDevice(**ds.static_kwargs(), device="cuda:0")
lib = dev.lib
n_rot = 60
qsets = ds.draw_batches(n_rot, 2048, seed=3)
prep, mx = [], 0
for q in qsets:
    r = ds.request(q, rc=True)
    b0 = dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], -1, to_rc=r["to_rc"])
    oo, tm, _ = dev.hap_offsets(b0)
    mx = max(mx, int(tm.cpu()[1]))
    prep.append((b0, oo, tm, r))
runs = [dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], -1, None, None, r["to_rc"], oo, max_row_len=mx) for b0, oo, tm, r in prep]
K = runs[0].n_rows
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(3)]
slots = [dev.alloc_output(runs[0], K * mx, haps=False, onehot=True) for _ in range(4 * G)]
dref = C.byref(dev.c)

def timeit(fn, n, use):
    for i in range(2 * len(use)): fn(i, use[i % len(use)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.Event(); st.record(use[0])
    for s in use[1:]: s.wait_event(st)
    e0.record(use[0])
    for i in range(n): fn(i, use[i % len(use)])
    for s in use[1:]:
        ev = torch.cuda.Event(); ev.record(s); use[0].wait_event(ev)
    e1.record(use[0]); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def size_only(i, s):
    b0, oo, tm, r = prep[i % n_rot]
    lib.gvl_hap_offsets(dref, C.byref(b0.c), None, C.c_void_p(oo.data_ptr()), C.c_void_p(tm.data_ptr()), C.c_void_p(s.cuda_stream))
def recon_only(i, s):
    lib.gvl_reconstruct(dref, C.byref(runs[i % n_rot].c), C.byref(slots[i % len(slots)][1]), C.c_void_p(s.cuda_stream))
packs = [dev.pack_many([runs[(g * G + j) % n_rot] for j in range(G)], [slots[(g % 4) * G + j][1] for j in range(G)]) for g in range(n_rot // G)]
def recon_many(i, s):
    b, o, n = packs[i % len(packs)]
    lib.gvl_reconstruct_many(dref, b, o, n, C.c_void_p(s.cuda_stream))
def group_step(i, s):       # what the loader submits per group: G sizings, one grid
    g = i % len(packs)
    for j in range(G):
        b0, oo, tm, r = prep[(g * G + j) % n_rot]
        lib.gvl_hap_offsets(dref, C.byref(b0.c), None, C.c_void_p(oo.data_ptr()), C.c_void_p(tm.data_ptr()), C.c_void_p(s.cuda_stream))
    b, o, n = packs[g]
    lib.gvl_reconstruct_many(dref, b, o, n, C.c_void_p(s.cuda_stream))

print(f"ragged cfg3 @ {scale}: {K} rows per batch, longest row {mx}, G = {G}")
for name, fn, per in (("sizing (2 kernels)", size_only, 1), ("reconstruct, 1 batch per launch", recon_only, 1),
                      (f"reconstruct, {G} batches per launch", recon_many, G), (f"group: {G} sizings + one grid", group_step, G)):
    # (every batch -- and its offsets buffer -- always on the same stream: i % n_streams with n_rot a multiple of 1, 3 and 4 x G)
    for ns in (1, 3):
        us = timeit(fn, 600 // per, streams[:ns]) / per
        print(f"  {name:40s} {ns} stream(s): {us:7.2f} us per batch")
