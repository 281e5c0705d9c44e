"""Pipelined step time vs number of batches in flight (cfg3), several repetitions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
st, bt = synth.make_config(wl)
L = bt.output_length
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
dbt = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=bt.to_rc)
cur = torch.cuda.current_stream()
all_streams = [cur] + [torch.cuda.Stream() for _ in range(7)]
all_slots = [dev.alloc_output(dbt, bt.n_windows * L, haps=False, onehot=True) for _ in range(9)]
for ns in (1, 2, 3, 4, 5, 6, 8):
    res = []
    for rep in range(5):
        streams, slots = all_streams[:ns], all_slots[:ns + 1]
        def loop(n):
            for i in range(n):
                dev.launch(dbt, slots[i % (ns + 1)][1], streams[i % ns])
        loop(30); torch.cuda.synchronize()
        t0 = time.perf_counter(); loop(1000); torch.cuda.synchronize(); t2 = time.perf_counter()
        res.append(1e6 * (t2 - t0) / 1000)
    print(f"{wl} streams={ns}: " + " ".join(f"{r:.2f}" for r in res) + " us/step")
