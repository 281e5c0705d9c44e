"""Host-side floor of the launch loop: same call as bench.py's pipelined region (explicit stream), tiny vs full batch,
and the same steps replayed from a hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth

for windows in (8, 4096):
    st, bt = synth.make_config("cfg3", windows=windows)
    L = bt.output_length
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
    dbt = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=bt.to_rc)
    for ns in (1, 2, 3, 4):
        cur = torch.cuda.current_stream()
        streams = [cur] + [torch.cuda.Stream() for _ in range(ns - 1)]
        slots = [dev.alloc_output(dbt, bt.n_windows * L, haps=False, onehot=True) for _ in range(ns + 1)]
        def loop(n):
            for i in range(n):
                dev.launch(dbt, slots[i % (ns + 1)][1], streams[i % ns])
        loop(50); torch.cuda.synchronize()
        t0 = time.perf_counter(); loop(3000); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"windows={windows} streams={ns}: host issue {1e6 * (t1 - t0) / 3000:.2f} us/step, wall {1e6 * (t2 - t0) / 3000:.2f} us/step")
    # graph: G steps over ns streams captured once
    for ns, G in ((1, 12), (2, 12), (3, 12), (4, 12)):
        side = [torch.cuda.Stream() for _ in range(ns)]
        slots = [dev.alloc_output(dbt, bt.n_windows * L, haps=False, onehot=True) for _ in range(ns + 1)]
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        with torch.cuda.stream(cap):
            g.capture_begin()
            for s in side: s.wait_stream(cap)
            for i in range(G):
                dev.launch(dbt, slots[i % (ns + 1)][1], side[i % ns])
            for s in side: cap.wait_stream(s)
            g.capture_end()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n): g.replay()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"windows={windows} graph streams={ns}: wall {1e6 * (t2 - t0) / (n * G):.2f} us/step")
