"""Where the host time of a spliced batch goes (cProfile over the dataset object's Python submit loop).  python tools/spliced_host_profile.py"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

# (bench.secondary_spliced builds the dataset; re-use its generator by calling it once for the warm-up, then profile the loader alone)
from genvarloader_amd import HapsDevice, synth
from genvarloader_amd.loader import DeviceSplicedHapsDataset

rng = np.random.default_rng(20260807)
st = synth.make_static(rng, (32 << 20,), indel_frac=0.15)
P, S, n_tx = 2, 8, 1500
n_ex = rng.integers(3, 15, n_tx)
ex_len = np.clip(np.exp(rng.normal(np.log(160.0), 0.95, int(n_ex.sum()))), 30, 9000).astype(np.int64)
intron = rng.integers(200, 3000, len(ex_len))
so = np.concatenate([[0], np.cumsum(n_ex)]).astype(np.int64)
starts = np.zeros(len(ex_len), np.int64); strand = np.zeros(len(ex_len), np.int64)
for t in range(n_tx):
    a, b = so[t], so[t + 1]
    span = int((ex_len[a:b] + intron[a:b]).sum())
    t0 = int(rng.integers(1000, (32 << 20) - span - 1000))
    starts[a:b] = t0 + np.concatenate([[0], np.cumsum(ex_len[a:b] + intron[a:b])[:-1]])
    strand[a:b] = 1 if rng.random() < 0.5 else -1
regions = np.stack([np.zeros(len(ex_len), np.int64), starts, starts + ex_len, strand], 1).astype(np.int32)
go, gv = synth.sample_genotypes(rng, st, np.repeat(regions[:, 0], S), np.repeat(regions[:, 1], S), np.repeat(regions[:, 2], S), P)
dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                 alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv, pad_char=st.pad_char)
ds = DeviceSplicedHapsDataset(dev, regions, S, P, splice_offsets=so, splice_region_idx=np.arange(len(regions)), onehot=True, haps=True, exonic=True)
dl = ds.to_dataloader(batch_size=256, shuffle=True, seed=3)
it = iter(dl)
for _ in range(5):
    next(it)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    next(it)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
