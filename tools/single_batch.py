"""tools/single_batch.py: one launch on its own (gvl_reconstruct, one-hot only), back to back on one stream, rotating cold batches of a
genome-scale dataset: 4096 rows x 2048 bases against 8192 rows x 1024 bases (the same bytes on twice the waves: what a row split over two
waves could reach at best) and 2048 x 2048."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvarloader_amd import HapsDevice, synth, _lib

ds = synth.make_genome("hg38", "cfg3", device="cuda:0", seed=20260805)
dev = HapsDevice(**ds.static_kwargs(), device="cuda:0")
lib = dev.lib
_lib.set_tuning(_lib.TUNE_PIPE_MIN_ROWS, 1 << 24)        # (the wave-per-row kernel whatever the launch's size)
cur = torch.cuda.current_stream()
sp = C.c_void_p(cur.cuda_stream)
for nq, L in ((2048, 2048), (4096, 1024), (1024, 2048), (2048, 2048)):
    qsets = ds.draw_batches(48, nq, seed=5)
    reqs = [ds.request(q, rc=True) for q in qsets]
    bts = [dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], L, to_rc=r["to_rc"]) for r in reqs]
    K = bts[0].n_rows
    slots = [dev.alloc_output(bts[0], K * L, haps=False, onehot=True) for _ in range(4)]
    def fn(i):
        _lib.check(lib.gvl_reconstruct(C.byref(dev.c), C.byref(bts[i % 48].c), C.byref(slots[i % 4][1]), sp))
    for i in range(10): fn(i)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for i in range(96): fn(i)
        e1.record(cur); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 96 * 1e3)
    us = sorted(res)[1]
    print(f"{K} rows x {L} bases, one launch at a time: {us:.2f} us per launch = {K * L * 5 / us / 1e6:.2f} TB/s of reference + one-hot")
