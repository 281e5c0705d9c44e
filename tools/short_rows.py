"""Row rate against row length: 65 536 rows per launch of L = 128 ... 2048 bases (SNP + indel, RC on half the rows), fixed-length and
ragged, one-hot / one-hot + bytes -- what a launch costs per row when the rows are short (a spliced batch's exons).
python tools/short_rows.py [rows per launch] [one length only]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ONLY_L = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # (one length, fixed one-hot only: tools/pmc_pipe_stalls.sh)
rng = np.random.default_rng(5)
st = synth.make_static(rng, (64 << 20,), indel_frac=0.15)
n = 12
for L in ((ONLY_L,) if ONLY_L else (128, 256, 512, 1024, 2048)):
    bt = synth.make_batch(rng, st, K // 2, 2, L, rc_frac=0.5, slack=16)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
    nv = float(bt.mean_variants)
    line = [f"L {L:5d}  variants/row {nv:5.2f}"]
    for name, out_len, haps in ((("fixed oh", L, False),) if ONLY_L else
                                (("fixed oh", L, False), ("fixed oh+bytes", L, True), ("ragged oh", -1, False), ("ragged oh+bytes", -1, True))):
        if out_len < 0:
            b0 = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, -1, to_rc=bt.to_rc)
            oo, tm, _ = dev.hap_offsets(b0)
            tot, mx = (int(x) for x in tm.cpu())
            b = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, -1, None, None, bt.to_rc, oo, max_row_len=mx, total_len=tot)
        else:
            b = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=bt.to_rc)
            tot = K * L
        out, oc = dev.alloc_output(b, tot, haps=haps, onehot=True)
        for _ in range(3):
            dev.launch(b, oc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            dev.launch(b, oc)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        by = tot * (5 if haps else 4) + tot / 2 + K * (29 * nv + 61)
        line.append(f"{name} {us:6.1f} us {K / us:5.0f} rows/us {by / us / 1e6:4.2f} TB/s")
    print(" | ".join(line), flush=True)
