"""GVL_TRACE=2: average host time of every HIP call site in the native loader loop (cfg5 epoch)."""
import os, sys
os.environ["GVL_TRACE"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth
from genvarloader_amd.loader import DeviceHapsDataset
R, S, P, L, bs = 200, 2504, 2, 2048, 2048
rng = np.random.default_rng(20260802 + 5)
st = synth.make_static(rng, (64 << 20,), indel_frac=0.15)
full_regions, go, gv = synth.make_grid(rng, st, R, S, P, L)
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv, pad_char=st.pad_char)
ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L)
dl = ds.to_dataloader(batch_size=bs, shuffle=True, in_flight=3)
for _ in range(4):
    for _b in dl: pass
torch.cuda.synchronize()
del dl
