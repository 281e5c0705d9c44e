"""How many rows of a cfg3 launch does the pipelined lean kernel defer to the wave's end (all-purpose body)?
python tools/pipe_deferred.py [scale=hg38] [G=16]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvarloader_amd import HapsDevice, synth

scale = sys.argv[1] if len(sys.argv) > 1 else "hg38"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ds = synth.make_genome(scale, "cfg3", device="cuda:0", seed=20260805)
dev = HapsDevice(**ds.static_kwargs(), device="cuda:0")
q = ds.draw_batches(1, G * 2048, seed=5)[0]
r = ds.request(q, rc=True)
bt = dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], ds.length, to_rc=r["to_rc"])
out, oc = dev.alloc_output(bt, bt.n_rows * ds.length, haps=False, onehot=True)
st = torch.zeros(64, dtype=torch.int64, device="cuda:0")
dev.lib.gvl_diag_set_stamps(C.c_void_p(st.data_ptr()))
dev.launch(bt, oc)
torch.cuda.synchronize()
dev.lib.gvl_diag_set_stamps(None)
rows, waves = int(st[0]), int(st[1])
print(f"{bt.n_rows} rows in one launch: {rows} deferred ({100.0 * rows / bt.n_rows:.3f} %) by {waves} waves")
names = {1: "not a lean row (edge / range / forced)", 2: "bad or overflowing records", 3: "no fixed point", 4: "trailing pad", 5: "runs leave the window / allele starts < 8 apart"}
print("   reasons:", {names[i]: int(st[2 + i]) for i in names if int(st[2 + i])}, "(rows with other bytes -- bytes-writing launches -- are not listed)")
