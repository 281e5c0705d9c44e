"""Kernel-time experiments on custom synthetic workloads (not part of the product)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth

def run(name, indel_frac, rc_frac, windows=4096, L=2048, contig=64 << 20, steps=300, haps=False, af=(0.6, 0.9), density=1/300):
    rng = np.random.default_rng(1)
    st = synth.make_static(rng, (contig,), indel_frac=indel_frac, af_beta=af, density=density)
    bt = synth.make_batch(rng, st, windows // 2, 2, L, rc_frac=rc_frac)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                     alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
    dbt = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=bt.to_rc)
    slots = [dev.alloc_output(dbt, bt.n_windows * L, haps=haps, onehot=True) for _ in range(2)]
    for i in range(30): dev.launch(dbt, slots[i & 1][1])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): dev.launch(dbt, slots[i & 1][1])
    e1.record(); torch.cuda.synchronize()
    print(f"{name:34s} V/row={bt.mean_variants:5.2f}  {e0.elapsed_time(e1) / steps * 1000:7.2f} us")

if __name__ == "__main__":
    run("snp only, no rc", 0.0, 0.0)
    run("snp only, rc 0.5", 0.0, 0.5)
    run("snp only, rc 1.0", 0.0, 1.0)
    run("indel 0.15, no rc", 0.15, 0.0)
    run("indel 0.15, rc 0.5", 0.15, 0.5)
    run("indel 0.5, no rc", 0.5, 0.0)
    run("no variants", 0.0, 0.0, density=1e-9)
    run("snp only + haps", 0.0, 0.0, haps=True)
