"""Does the ragged form's store alignment matter?  cfg3 rows (cold hg38-scale dataset) through recon_lean_rows_kernel<onehot, RAG> with
every row of the SAME length Lr at out_offsets = k * Lr: Lr = 2048 (every row starts on a line: 8 KB of one-hot per row), 2052, 2064,
2080, 2112 (rows start 16 / 64 / 128 / 256 bytes off), next to the fixed-length form (Lr = 2048, no offsets array).  16 batches per
launch, 3 streams, rotating cold batches.  python tools/ragged_align.py
WANT=haps: haplotype BYTES only (the reference's default output, `with_seqs("haplotypes")`), WANT=both: one-hot + bytes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvarloader_amd import HapsDevice, synth

ds = synth.make_genome("hg38", "cfg3", device="cuda:0", seed=20260805)
dev = HapsDevice(**ds.static_kwargs(), device="cuda:0")
lib = dev.lib
G, n_rot = 16, 64
WANT = os.environ.get("WANT", "onehot")
W_H, W_O = WANT in ("haps", "both"), WANT in ("onehot", "both")
qsets = ds.draw_batches(n_rot, 2048, seed=3)
reqs = [ds.request(q, rc=True) for q in qsets]
K = 4096
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(2)]
dref = C.byref(dev.c)


def measure(Lr, ragged):
    if ragged:
        oo = (torch.arange(K + 1, dtype=torch.int64, device="cuda") * Lr).contiguous()
        bts = [dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], -1, None, None, r["to_rc"], oo, max_row_len=Lr) for r in reqs]
    else:
        bts = [dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], Lr, to_rc=r["to_rc"]) for r in reqs]
    slots = [dev.alloc_output(bts[0], K * Lr, haps=W_H, onehot=W_O) for _ in range(4 * G)]
    packs = [dev.pack_many([bts[(g * G + j) % n_rot] for j in range(G)], [slots[(g % 4) * G + j][1] for j in range(G)]) for g in range(n_rot // G * 4)]
    def fn(i):
        b, o, n = packs[i % len(packs)]
        lib.gvl_reconstruct_many(dref, b, o, n, C.c_void_p(streams[i % 3].cuda_stream))
    for i in range(12): fn(i)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.Event(); st.record(streams[0])
        for s in streams[1:]: s.wait_event(st)
        e0.record(streams[0])
        n = 120
        for i in range(n): fn(i)
        for s in streams[1:]:
            ev = torch.cuda.Event(); ev.record(s); streams[0].wait_event(ev)
        e1.record(streams[0]); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / (n * G) * 1e3)
    return sorted(res)[1]


base = None
CASES = ((2048, False), (2048, True), (2052, True), (2064, True), (2080, True), (2112, True), (2048, False))
if os.environ.get("DECIMAL"):      # fixed-length rows whose one-hot is not a multiple of 128 bytes, through the fixed form and through the ragged one
    CASES = ((2000, False), (2000, True), (1000, False), (1000, True), (500, False), (500, True), (2048, False), (1024, False))
for Lr, rag in CASES:
    us = measure(Lr, rag)
    out_b = Lr * 4096 * ((4 if W_O else 0) + (1 if W_H else 0))
    per_kb = us / (out_b / 1e6)
    alg = Lr * 4096 * (1 + (4 if W_O else 0) + (1 if W_H else 0)) + 4096 * (28 * 2.87 + 61)
    print(f"{'ragged' if rag else 'fixed '} rows of {Lr} ({WANT}): {us:6.2f} us per batch   {per_kb * 1e3:6.1f} ns per MB of output   {alg / us / 1e6 / 8:.2f} of 8 TB/s (algorithmic)", flush=True)
