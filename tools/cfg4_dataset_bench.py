"""BASELINE config 4 end to end from dataset indices: 128 (region, sample) pairs per batch = 256 haplotype rows x
131072 bp, one-hot + haplotype bytes + one realigned track (Repeat5p), through DeviceHapsTracksDataset."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvarloader_amd import HapsDevice, synth
from genvarloader_amd.loader import DeviceHapsTracksDataset

R, S, P, L = 16, 64, 2, 131072
rng = np.random.default_rng(20260802 + 4)
st = synth.make_static(rng, (64 << 20,), indel_frac=0.15)
full_regions, go, gv = synth.make_grid(rng, st, R, S, P, L)
starts, ends, vals, offs = [], [], [], [0]
for r in range(R):
    q0, q1 = int(full_regions[r, 1]), int(full_regions[r, 2])
    n = (q1 - q0 + 200) // 33
    w = rng.geometric(1 / 25, size=(S, n)); g = rng.geometric(1 / 8, size=(S, n))
    for s_ in range(S):
        s0 = np.cumsum(w[s_] + g[s_]) - w[s_] + q0 - 100
        m = s0 < q1 + 50
        starts.append(s0[m]); ends.append((s0 + w[s_])[m]); vals.append((rng.random(int(m.sum())) * 8).astype(np.float32))
        offs.append(offs[-1] + int(m.sum()))
tracks = {"cov": (np.concatenate(starts).astype(np.int32), np.concatenate(ends).astype(np.int32), np.concatenate(vals),
                  np.asarray(offs, np.int64))}
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv, pad_char=st.pad_char)
ds = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=tracks, output_length=L, onehot=True, haps=True)
dl = ds.to_dataloader(batch_size=128, shuffle=True, generator=torch.Generator().manual_seed(0), in_flight=2)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for batch in dl:
        n += batch.tracks.shape[0] * P
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"cfg4 dataset: {n} rows x {L} bp (one-hot + bytes + 1 track) in {dt * 1e3:.2f} ms = {dt / len(dl) * 1e6:.0f} us per 256-row batch, "
      f"{n / dt / 1e3:.0f} k windows/s, {n * L * (6 + 8) / dt / 1e9:.0f} GB/s algorithmic")
