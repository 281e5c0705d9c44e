"""Stress the threaded loader: many short epochs, random early exits, loaders created and dropped."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth
from genvarloader_amd.loader import DeviceHapsDataset

R, S, P, L = 20, 50, 2, 512
rng = np.random.default_rng(1)
st = synth.make_static(rng, (2 << 20,), indel_frac=0.15)
full_regions, go, gv = synth.make_grid(rng, st, R, S, P, L)
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv, pad_char=st.pad_char)
ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, jitter=3, deterministic=False, seed=2)
ref = None
t0 = time.time()
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for it in range(n_iter):
    dl = ds.to_dataloader(batch_size=int(rng.integers(1, 64)), shuffle=True, seed=it, in_flight=int(rng.integers(1, 5)), threaded=True)
    for ep in range(int(rng.integers(1, 4))):
        stop_at = int(rng.integers(0, 40)) if rng.random() < 0.5 else -1
        tot = 0
        for bi, b in enumerate(dl):
            tot += int(b.idx.numel())
            if bi == stop_at:
                break
        if stop_at < 0:
            assert tot == R * S, tot
    del dl
torch.cuda.synchronize()
print(f"{n_iter} loaders, ok, {time.time() - t0:.1f} s")
