"""In-process A/B of GVL_DBG bit sets: alternates the flag sets round-robin on the same box,
reports the median single-stream launch time and the 3-streams-in-flight step time per set."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth, _lib

sets = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,16").split(",")]
wls = (sys.argv[2] if len(sys.argv) > 2 else "cfg3,cfg2").split(",")
lib = _lib.load()
for wl in wls:
    st, bt = synth.make_config(wl)
    L = bt.output_length
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
    dbt = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=bt.to_rc)
    cur = torch.cuda.current_stream()
    streams = [cur] + [torch.cuda.Stream() for _ in range(2)]
    slots = [dev.alloc_output(dbt, bt.n_windows * L, haps=False, onehot=True) for _ in range(4)]
    res = {f: ([], []) for f in sets}
    for rep in range(7):
        for f in sets:
            lib.gvl_set_debug_flags(f)
            for i in range(20): dev.launch(dbt, slots[i % 4][1], cur)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            for i in range(300): dev.launch(dbt, slots[i % 4][1], cur)
            e1.record(cur); torch.cuda.synchronize()
            res[f][0].append(e0.elapsed_time(e1) / 300 * 1e3)
            t0 = time.perf_counter()
            for i in range(600): dev.launch(dbt, slots[i % 4][1], streams[i % 3])
            torch.cuda.synchronize()
            res[f][1].append((time.perf_counter() - t0) / 600 * 1e6)
    for f in sets:
        a, b = np.array(res[f][0]), np.array(res[f][1])
        print(f"{wl} dbg={f:4d}: launch median {np.median(a):6.2f} us (min {a.min():.2f} max {a.max():.2f})   3-in-flight median {np.median(b):5.2f} us (min {b.min():.2f})")
lib.gvl_set_debug_flags(-1)
