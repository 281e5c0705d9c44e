"""Device-side request prep + loader surface."""

import numpy as np
import pytest
import torch

from genvarloader_amd import synth
from genvarloader_amd.loader import build_request


def _grid_dataset(seed, R, S, P, L, contig=200_000, **kw):
    """Regions x samples x ploidy genotype grid laid out like the reference's sparse genotypes."""
    rng = np.random.default_rng(seed)
    st = synth.make_static(rng, (contig,), indel_frac=kw.pop("indel_frac", 0.2), density=1 / 80)
    base = synth.make_batch(rng, st, R, 1, L, rc_frac=0.5, slack=kw.pop("slack", 40))
    full_regions = base.regions                                     # (R, 4)
    # genotype CSR for every (region, sample, ploid): reuse make_batch's sampler per sample
    lists = []
    for r in range(R):
        lo = np.searchsorted(st.v_starts, full_regions[r, 1] - 40)
        hi = np.searchsorted(st.v_starts, full_regions[r, 2])
        cand = np.arange(lo, hi)
        for s in range(S):
            for p in range(P):
                lists.append(cand[rng.random(len(cand)) < st.af[cand]].astype(np.int32))
    lens = np.array([len(x) for x in lists])
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    geno_offsets = np.ascontiguousarray(np.stack([offs[:-1], offs[1:]]))
    geno_v_idxs = np.concatenate(lists) if lists else np.zeros(0, np.int32)
    return st, full_regions, geno_offsets, geno_v_idxs


def test_build_request_matches_reference_index_math():
    R, S, P = 7, 5, 2
    rng = np.random.default_rng(0)
    full = np.stack([np.zeros(R), rng.integers(0, 1000, R), np.zeros(R), rng.choice([-1, 1], R)], 1).astype(np.int32)
    full[:, 2] = full[:, 1] + 100
    idx = rng.permutation(R * S)[:11]
    regions, goi, to_rc, lengths = build_request(torch.from_numpy(idx), torch.from_numpy(full), S, P)
    r_idx, s_idx = np.unravel_index(idx, (R, S))                   # _torch.py:299, _query.py:161
    np.testing.assert_array_equal(regions.numpy(), full[r_idx])
    exp_goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))   # _haps.py:757-768
    np.testing.assert_array_equal(goi.numpy(), exp_goi)
    np.testing.assert_array_equal(to_rc.numpy().astype(bool), np.repeat(full[r_idx, 3] == -1, P))   # _haps.py:838-843
    np.testing.assert_array_equal(lengths.numpy(), 100)
    # jitter keeps the region length and stays within +-j (_query.py:166-171)
    g = torch.Generator().manual_seed(1)
    rj, _, _, _ = build_request(torch.from_numpy(idx), torch.from_numpy(full), S, P, jitter=5, generator=g)
    d = rj[:, 1].numpy() - full[r_idx, 1]
    assert (np.abs(d) <= 5).all() and len(set(d.tolist())) > 1
    np.testing.assert_array_equal(rj[:, 2].numpy() - rj[:, 1].numpy(), 100)


@pytest.mark.gpu
def test_loader_deterministic_matches_oracle(oracle):
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 6, 9, 2, 777
    st, full_regions, go, gv = _grid_dataset(3, R, S, P, L)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, onehot=True, haps=True)
    assert len(ds) == R * S and ds.shape == (R, S)
    # the prep kernel == the torch statement of the reference's index math
    probe = np.random.default_rng(1).permutation(R * S)[:13]
    _, kr, ks, kg, kc = ds.request(probe)
    tr, tg, tc, _ = build_request(torch.from_numpy(probe).cuda(), ds.full_regions, S, P)
    assert torch.equal(kr, tr) and torch.equal(kg, tg) and torch.equal(kc, tc) and int(ks.abs().sum()) == 0
    seen = 0
    for batch in ds.to_dataloader(batch_size=8, shuffle=False, in_flight=3):
        idx = batch.idx.cpu().numpy()
        r_idx, s_idx = np.unravel_index(idx, (R, S))
        regions = full_regions[r_idx]
        goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
        to_rc = np.repeat(regions[:, 3] == -1, P)
        exp, _, exp_oh = oracle.reconstruct_haplotypes_fused(
            regions, np.zeros_like(goi, dtype=np.int32), goi, go, gv, st.v_starts, st.ilens, st.alt_alleles,
            st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, L, None, None, to_rc, False, onehot=True)
        assert batch.onehot.shape == (len(idx), P, L, 4) and batch.haps.shape == (len(idx), P, L)
        np.testing.assert_array_equal(batch.haps.cpu().numpy().ravel(), exp)
        np.testing.assert_array_equal(batch.onehot.cpu().numpy().reshape(-1, 4), exp_oh)
        seen += len(idx)
    assert seen == R * S


@pytest.mark.gpu
def test_loader_random_shifts_and_jitter(oracle):
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 5, 8, 2, 500
    st, full_regions, go, gv = _grid_dataset(4, R, S, P, L + 80, indel_frac=0.5, slack=0)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, jitter=3, deterministic=False, seed=7,
                           onehot=True, haps=True)
    all_shifts = []
    for batch in ds.to_dataloader(batch_size=10, shuffle=True, generator=torch.Generator().manual_seed(0)):
        regions = batch.regions.cpu().numpy()
        r_idx = batch.idx.cpu().numpy() // S
        jit = regions[:, 1] - full_regions[r_idx, 1]
        assert (np.abs(jit) <= 3).all() and (regions[:, 2] - regions[:, 1] == full_regions[r_idx, 2] - full_regions[r_idx, 1]).all()
        shifts = batch.shifts.cpu().numpy()
        goi = batch.geno_offset_idx.cpu().numpy()
        # the request is internally consistent: shifts in [0, max_shift] (_haps.py:728-730)
        diffs = oracle.get_diffs_sparse(goi, gv, go, st.ilens, None, None, regions[:, 1], regions[:, 2], st.v_starts)
        max_shift = np.clip(diffs, 0, None) + np.clip((regions[:, 2] - regions[:, 1]) - L, 0, None)[:, None]
        assert (shifts >= 0).all() and (shifts <= max_shift).all()
        all_shifts.append(shifts.ravel())
        # and the output is the reference's for exactly that request
        to_rc = None if batch.to_rc is None else batch.to_rc.cpu().numpy().astype(bool)
        exp, _, exp_oh = oracle.reconstruct_haplotypes_fused(
            regions, shifts, goi, go, gv, st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref,
            st.ref_offsets, st.pad_char, L, None, None, to_rc, False, onehot=True)
        np.testing.assert_array_equal(batch.haps.cpu().numpy().ravel(), exp)
        np.testing.assert_array_equal(batch.onehot.cpu().numpy().reshape(-1, 4), exp_oh)
    s = np.concatenate(all_shifts)
    assert s.max() > 0 and len(s) == R * S * P


@pytest.mark.gpu
def test_native_loop_ring_reuse_drop_last_and_epochs(oracle):
    """Many more batches than ring slots, a consumer that lags behind on its own stream, a short
    last batch / drop_last, two epochs over the same loader: every batch must be the oracle's
    and must not be overwritten before the consumer's queued work has read it."""
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 7, 11, 2, 300
    st, full_regions, go, gv = _grid_dataset(9, R, S, P, L, indel_frac=0.3)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, onehot=False, haps=True)
    idx_all = np.arange(R * S)
    r_idx, s_idx = np.unravel_index(idx_all, (R, S))
    goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
    exp, _ = oracle.reconstruct_haplotypes_fused(
        full_regions[r_idx], np.zeros_like(goi, dtype=np.int32), goi, go, gv, st.v_starts, st.ilens, st.alt_alleles,
        st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, L, None, None, np.repeat(full_regions[r_idx, 3] == -1, P),
        False)
    exp = exp.reshape(R * S, P, L)
    for drop_last, group in ((False, 2), (True, 1), (False, None)):      # (None: the largest group the ring budget allows)
        dl = ds.to_dataloader(batch_size=3, shuffle=True, generator=torch.Generator().manual_seed(5), drop_last=drop_last,
                              in_flight=2, group=group)
        assert len(dl) == (R * S) // 3 + (0 if drop_last or (R * S) % 3 == 0 else 1)
        for epoch in range(2):
            copies, idxs = [], []
            lag = torch.zeros(1 << 22, device="cuda")
            for batch in dl:
                lag.add_(1.0)                        # the consumer is busy before it reads the batch
                copies.append(batch.haps.clone())    # read on the consumer stream, queued behind `lag`
                idxs.append(batch.idx.clone())
            torch.cuda.synchronize()
            got_idx = torch.cat(idxs).cpu().numpy()
            n_exp = (R * S) // 3 * 3 if drop_last else R * S
            assert len(got_idx) == n_exp and len(set(got_idx.tolist())) == n_exp
            got = torch.cat(copies).cpu().numpy()
            np.testing.assert_array_equal(got, exp[got_idx])


@pytest.mark.gpu
def test_loader_sampler_path(oracle):
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 4, 5, 2, 260
    st, full_regions, go, gv = _grid_dataset(10, R, S, P, L)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, onehot=True, haps=True)
    sampler = [[0, 19, 3], [7], [5, 5, 12, 1]]
    held = list(ds.to_dataloader(sampler=sampler, in_flight=2))      # sampler batches own their memory
    assert [len(b.idx) for b in held] == [3, 1, 4]
    for b, want in zip(held, sampler):
        idx = np.asarray(want)
        r_idx, s_idx = np.unravel_index(idx, (R, S))
        goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
        exp, _, exp_oh = oracle.reconstruct_haplotypes_fused(
            full_regions[r_idx], np.zeros_like(goi, dtype=np.int32), goi, go, gv, st.v_starts, st.ilens,
            st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, L, None, None,
            np.repeat(full_regions[r_idx, 3] == -1, P), False, onehot=True)
        np.testing.assert_array_equal(b.haps.cpu().numpy().ravel(), exp)
        np.testing.assert_array_equal(b.onehot.cpu().numpy().reshape(-1, 4), exp_oh)


@pytest.mark.gpu
def test_loader_rank_shares_are_disjoint_and_cover():
    """Two ranks' loaders (run one after the other on this GPU) see disjoint, equal shares of the
    same per-epoch permutation -- the N-GPU epoch needs no collective."""
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 5, 7, 2, 256
    st, full_regions, go, gv = _grid_dataset(12, R, S, P, L)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L)
    seen = []
    for rank in range(2):
        dl = ds.to_dataloader(batch_size=4, shuffle=True, rank=rank, world_size=2, seed=3)
        dl.set_epoch(7)
        ids = torch.cat([b.idx.clone() for b in dl]).cpu().numpy()
        assert len(ids) == -(-R * S // 2) and len(dl) == -(-len(ids) // 4)
        seen.append(ids)
    both = np.concatenate(seen)
    assert set(both.tolist()) == set(range(R * S)) and len(both) == R * S + (R * S) % 2


@pytest.mark.gpu
def test_cfg5_epoch_full_size_properties_and_sampled_parity(oracle):
    """BASELINE configs[4]: a 1.0 M-window epoch (200 regions x 2504 samples x 2 haplotypes,
    2048 bp, one-hot) through the native loader in batches of 4096 windows.  Size-independent
    properties over the whole epoch (every dataset index exactly once; every one-hot row has at
    most one channel set and the count of all-zero rows equals the count of N / pad bases; a
    checksum of per-batch checksums equal to the same sum taken in index order) and bit-exact
    parity with the oracle on sampled batches."""
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L, bs = 200, 2504, 2, 2048, 2048
    rng = np.random.default_rng(20260802 + 5)
    st = synth.make_static(rng, (64 << 20,), indel_frac=0.15)
    full_regions, go, gv = synth.make_grid(rng, st, R, S, P, L)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, onehot=True, haps=True)
    w = torch.arange(1, 5, dtype=torch.int64, device="cuda")              # channel weights of the checksum

    def epoch(shuffle, sample_at):
        seen = torch.zeros(R * S, dtype=torch.int32, device="cuda")
        total = torch.zeros((), dtype=torch.int64, device="cuda")
        zero_rows = torch.zeros((), dtype=torch.int64, device="cuda")
        non_acgt = torch.zeros((), dtype=torch.int64, device="cuda")
        samples = []
        for bi, batch in enumerate(ds.to_dataloader(batch_size=bs, shuffle=shuffle, seed=3, in_flight=3)):
            seen.index_add_(0, batch.idx, torch.ones_like(batch.idx, dtype=torch.int32))
            oh = batch.onehot
            rs = oh.sum(-1, dtype=torch.int32)
            assert int(rs.max()) <= 1
            zero_rows += (rs == 0).sum()
            hp = batch.haps
            non_acgt += ((hp != 65) & (hp != 67) & (hp != 71) & (hp != 84)).sum()
            # per-batch checksum weighted by the dataset index so that order does not matter
            per_q = (oh.to(torch.int64) * w).sum(dim=(1, 2, 3))
            total += (per_q * (batch.idx + 1)).sum()
            if bi in sample_at:
                samples.append((batch.idx.cpu().numpy(), hp.cpu().numpy().copy(), oh.cpu().numpy().copy()))
        torch.cuda.synchronize()
        assert int(seen.min()) == 1 and int(seen.max()) == 1
        assert int(zero_rows) == int(non_acgt)
        return int(total), samples

    # (groups of 16 batches = one grid: in-group positions 0, 13, 15, 15, 4 -- positions 10-15 hold the SECOND rows of the two-row waves)
    t_shuffled, samples = epoch(True, {0, 13, 111, 239, 244})
    t_ordered, _ = epoch(False, set())
    assert t_shuffled == t_ordered
    assert len(samples) == 5
    for idx, hp, oh in samples:
        r_idx, s_idx = np.unravel_index(idx, (R, S))
        regions = full_regions[r_idx]
        goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
        to_rc = np.repeat(regions[:, 3] == -1, P)
        exp, _, exp_oh = oracle.reconstruct_haplotypes_fused(
            regions, np.zeros_like(goi, dtype=np.int32), goi, go, gv, st.v_starts, st.ilens, st.alt_alleles,
            st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, L, None, None, to_rc, False, onehot=True, n_threads=8)
        np.testing.assert_array_equal(hp.ravel(), exp)
        np.testing.assert_array_equal(oh.reshape(-1, 4), exp_oh)


@pytest.mark.gpu
@pytest.mark.timeout(120)
def test_threaded_loader_matches_inline_loader_and_survives_abandoned_epochs(oracle):
    """threaded=True (a producer thread submits the batches): same batches as the inline loop for
    the same seed, over several epochs, with a lagging consumer, after an epoch abandoned half way,
    and the loader can be dropped while batches are still in flight."""
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 6, 10, 2, 256
    st, full_regions, go, gv = _grid_dataset(31, R, S, P, L, indel_frac=0.3)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, onehot=False, haps=True, jitter=2,
                           deterministic=False, seed=4)

    def run(threaded, epochs):
        ds._counter = ds._loaders = 0          # same derived loader seed for both runs
        dl = ds.to_dataloader(batch_size=4, shuffle=True, seed=9, in_flight=2, threaded=threaded)
        res = []
        lag = torch.zeros(1 << 20, device="cuda")
        for e in epochs:
            dl.set_epoch(e)
            for bi, batch in enumerate(dl):
                lag.add_(1.0)
                res.append((batch.idx.clone(), batch.regions.clone(), batch.shifts.clone(), batch.haps.clone()))
        torch.cuda.synchronize()
        return dl, res

    _, a = run(False, [0, 1, 2])
    dl_t, b = run(True, [0, 1, 2])
    assert len(a) == len(b) == 3 * 15
    for x, y in zip(a, b):
        for t, u in zip(x, y):
            assert torch.equal(t, u)
    # abandon an epoch half way, start another one, then drop the loader with work in flight
    it = iter(dl_t)
    for _ in range(5):
        next(it)
    del it
    dl_t.set_epoch(0)
    first = next(iter(dl_t))
    assert torch.equal(first.idx, a[0][0])
    del dl_t
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("dbg", [0, 1024, 8, 8192, 2097152, 4194304, 1073741824],
                         ids=["default", "painter-without-bucket-index", "scalar-walk", "painter-image-path", "intervals-without-window", "painter-first",
                              "general-track-kernel"])
def test_haps_tracks_dataset_matches_oracle(oracle, dbg):
    """cfg4's dataset shape at a small size: haplotypes + two realigned tracks per batch from dataset
    indices, against the oracle's fused paint + realign for the same request."""
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsTracksDataset

    R, S, P, L = 5, 6, 2, 700
    st, full_regions, go, gv = _grid_dataset(41, R, S, P, L, indel_frac=0.4, slack=30)
    rng = np.random.default_rng(4)
    tracks = {}
    for name in ("cov", "atac"):
        starts, ends, vals, offs = [], [], [], [0]
        for r in range(R):
            for s_ in range(S):
                pos = int(full_regions[r, 1]) - int(rng.integers(0, 60))
                while pos < int(full_regions[r, 2]) + 40:
                    w, gap = int(rng.geometric(1 / 20)), int(rng.integers(0, 6))
                    starts.append(pos + gap); ends.append(pos + gap + w); vals.append(float(rng.random() * 5)); pos += gap + w
                offs.append(len(starts))
        tracks[name] = (np.array(starts, np.int32), np.array(ends, np.int32), np.array(vals, np.float32), np.array(offs, np.int64))
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    from genvarloader_amd import _lib

    _lib.load().gvl_set_debug_flags(dbg)
    loops = (False, True) if dbg == 0 else (False,)       # the native ring; batches submitted from Python
    for strategy, param, pinned, python_loop in [(*c, pl) for c in ((0, 0.0, 11), (4, 3.0, 11), (3, 6.0, None), (3, 6.0, 5))
                                                 for pl in loops]:
        ds = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=tracks, strategy_id=strategy, param=param,
                                     base_seed=pinned, output_length=L, onehot=False, haps=True)
        assert all(ds._tile_complete)        # BigWig-like lists: the painter needs no leftovers launch (and skips it when dbg == 0)
        seen = 0
        for batch in ds.to_dataloader(batch_size=7, shuffle=True, generator=torch.Generator().manual_seed(1),
                                      python_loop=python_loop, threaded=(strategy == 4 and not python_loop)):
            idx = batch.idx.cpu().numpy()
            r_idx, s_idx = np.unravel_index(idx, (R, S))
            regions = full_regions[r_idx]
            goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
            to_rc = np.repeat(regions[:, 3] == -1, P)
            shifts = np.zeros_like(goi, dtype=np.int32)
            exp_h, _ = oracle.reconstruct_haplotypes_fused(
                regions, shifts, goi, go, gv, st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref,
                st.ref_offsets, st.pad_char, L, None, None, to_rc, False)
            np.testing.assert_array_equal(batch.haps.cpu().numpy().ravel(), exp_h)
            diffs = oracle.get_diffs_sparse(goi, gv, go, st.ilens, None, None, regions[:, 1], regions[:, 2], st.v_starts)
            tlen = (regions[:, 2] - regions[:, 1]).astype(np.int64) - np.minimum(diffs.min(axis=1), 0)
            track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
            out_offsets = np.arange(len(idx) * P + 1, dtype=np.int64) * L
            assert batch.tracks.shape == (len(idx), 2, P, L)
            for t, name in enumerate(("cov", "atac")):
                a, e, v, io = tracks[name]
                exp = np.zeros(len(idx) * P * L, np.float32)
                oracle.intervals_and_realign_track_fused(
                    exp, out_offsets, regions, shifts, goi, gv, go, st.v_starts, st.ilens, idx.astype(np.int64), a, e, v, io,
                    track_offsets, np.array([param]), strategy,
                    # the reference's per-batch seed: xor-reduce of the dataset indices (_reconstruct.py:215-218)
                    pinned if pinned is not None else int(np.bitwise_xor.reduce(idx.astype(np.uint64))), None, None, to_rc)
                got = batch.tracks[:, t].contiguous().cpu().numpy().ravel()
                np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32), err_msg=f"{name} strategy {strategy}")
            seen += len(idx)
        assert seen == R * S
    _lib.load().gvl_set_debug_flags(-1)


@pytest.mark.gpu
def test_random_fill_seeds_are_per_batch_and_the_same_in_both_submit_loops():
    """Non-deterministic datasets draw a fresh FlankSample base seed per BATCH (_reconstruct.py:215-222).  The native loop keys it
    by (draw seed, epoch + 1, batch number) on the device (batch_seeds_kernel); the Python submit loop must key it the same way
    -- it once pinned ONE seed for the whole epoch -- so: both loops deliver identical tracks, batch by batch,
    and the insertion fills of two batches of an epoch differ."""
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsTracksDataset

    R, S, P, L = 4, 6, 2, 600
    st, full_regions, go, gv = _grid_dataset(43, R, S, P, L, indel_frac=0.9, slack=30)
    rng = np.random.default_rng(8)
    starts, ends, vals, offs = [], [], [], [0]
    for r in range(R):
        for s_ in range(S):
            pos = int(full_regions[r, 1]) - 20
            while pos < int(full_regions[r, 2]) + 40:
                w = int(rng.geometric(1 / 9))
                starts.append(pos); ends.append(pos + w); vals.append(float(rng.random() * 5)); pos += w
            offs.append(len(starts))
    tracks = {"t": (np.array(starts, np.int32), np.array(ends, np.int32), np.array(vals, np.float32), np.array(offs, np.int64))}
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)

    def epoch(python_loop):
        ds = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=tracks, strategy_id=3, param=8.0, base_seed=None,
                                     deterministic=False, seed=5, output_length=L, onehot=False, haps=True)
        dl = ds.to_dataloader(batch_size=6, shuffle=False, python_loop=python_loop, draw_stream=1)
        dl.set_epoch(2)
        return [(b.idx.clone(), b.tracks.clone()) for b in dl]

    nat, py = epoch(False), epoch(True)
    assert len(nat) == len(py) == 4
    for (i0, t0), (i1, t1) in zip(nat, py):
        assert torch.equal(i0, i1)
        assert torch.equal(t0.view(torch.int32), t1.view(torch.int32))
    # the same dataset indices in another batch slot of the epoch get another seed: deliver the epoch's queries rotated by one batch
    ds = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=tracks, strategy_id=3, param=8.0, base_seed=None,
                                 deterministic=False, seed=5, output_length=L, onehot=False, haps=True)
    order = torch.arange(R * S)
    rot = torch.cat([order[6:], order[:6]])
    dl = ds.to_dataloader(batch_size=6, sampler=[rot[i:i + 6].tolist() for i in range(0, R * S, 6)], draw_stream=1)
    dl.set_epoch(2)
    shifted = [(b.idx.clone(), b.tracks.clone()) for b in dl]
    assert torch.equal(shifted[0][0].cpu(), py[1][0].cpu())               # batch 1's indices, now delivered as batch 0
    assert not torch.equal(shifted[0][1].view(torch.int32), py[1][1].view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("dbg", [0, 1024, 8192, 2097152, 4194304, 1073741824],
                         ids=["bucket-index", "exact-searches", "painter-image-path", "intervals-without-window", "painter-first", "general-track-kernel"])
def test_tracks_batch_long_rows_jitter_and_dense_lists(oracle, dbg):
    """gvl_tracks_batch on rows of many 2048-value chunks whose starts are not bucket aligned (jitter),
    with sparse, ordinary and very dense interval lists (a dense list overflows the painter's tile:
    per-value kernel; two buckets of candidates beyond the tile: exact searches)."""
    from genvarloader_amd import HapsDevice, _lib
    from genvarloader_amd.loader import DeviceHapsTracksDataset

    R, S, P, L = 3, 4, 2, 9000
    st, full_regions, go, gv = _grid_dataset(77, R, S, P, L, contig=400_000, indel_frac=0.4, slack=50)
    rng = np.random.default_rng(9)
    starts, ends, vals, offs = [], [], [], [0]
    for r in range(R):
        for s_ in range(S):
            mean_w = (3, 25, 400)[(r * S + s_) % 3]
            pos = int(full_regions[r, 1]) - int(rng.integers(0, 300))
            while pos < int(full_regions[r, 2]) + 300:
                w, gap = int(rng.geometric(1 / mean_w)), int(rng.integers(0, 4)) if mean_w < 100 else int(rng.integers(0, 3000))
                starts.append(pos + gap); ends.append(pos + gap + w); vals.append(float(rng.random() * 5)); pos += gap + w
            offs.append(len(starts))
    tracks = {"t": (np.array(starts, np.int32), np.array(ends, np.int32), np.array(vals, np.float32), np.array(offs, np.int64))}
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    _lib.load().gvl_set_debug_flags(dbg)
    try:
        ds = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=tracks, strategy_id=0, output_length=L, jitter=37,
                                     onehot=False, haps=True, seed=3)
        for batch in ds.to_dataloader(batch_size=5, shuffle=True, seed=2, in_flight=2, group=2):
            idx = batch.idx.cpu().numpy()
            regions, shifts = batch.regions.cpu().numpy(), batch.shifts.cpu().numpy()
            goi = batch.geno_offset_idx.cpu().numpy()
            to_rc = batch.to_rc.cpu().numpy().astype(bool)
            diffs = oracle.get_diffs_sparse(goi, gv, go, st.ilens, None, None, regions[:, 1], regions[:, 2], st.v_starts)
            tlen = (regions[:, 2] - regions[:, 1]).astype(np.int64) - np.minimum(diffs.min(axis=1), 0)
            track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
            out_offsets = np.arange(len(idx) * P + 1, dtype=np.int64) * L
            a, e, v, io = tracks["t"]
            exp = np.zeros(len(idx) * P * L, np.float32)
            oracle.intervals_and_realign_track_fused(exp, out_offsets, regions, shifts, goi, gv, go, st.v_starts, st.ilens,
                                                     idx.astype(np.int64), a, e, v, io, track_offsets, np.array([0.0]), 0,
                                                     0, None, None, to_rc)
            got = batch.tracks[:, 0].contiguous().cpu().numpy().ravel()
            np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))
    finally:
        _lib.load().gvl_set_debug_flags(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("dbg", [0, 268435456, 1073741824], ids=["row-plans", "every-chunk-walks", "row-plans-general-track-kernel"])
@pytest.mark.parametrize("python_loop", [False, True], ids=["epoch-table-plans", "per-call-plans"])
@pytest.mark.parametrize("strategy,param", [(0, 0.0), (4, 3.0)], ids=["repeat5p", "interpolate"])
def test_tracks_row_plans_rows_of_many_trips(oracle, dbg, python_loop, strategy, param):
    """Rows of 20 chunks with ~200 variants each (4 trips of the realignment's planned walk), random shifts and jitter: a chunk's
    wave reads its entries from the row's plan (track_plan_kernel: once per epoch with the native ring's table, once per
    gvl_tracks_batch call otherwise) instead of walking the row's variants -- against the oracle, and with GVL_DBG = 268435456
    (no plans: every chunk walks) the same."""
    from genvarloader_amd import HapsDevice, _lib
    from genvarloader_amd.loader import DeviceHapsTracksDataset

    R, S, P, L = 2, 3, 2, 40000
    st, full_regions, go, gv = _grid_dataset(311, R, S, P, L, contig=300_000, indel_frac=0.5, slack=120)
    assert (go[1] - go[0]).max() > 130
    rng = np.random.default_rng(12)
    starts, ends, vals, offs = [], [], [], [0]
    for r in range(R):
        for s_ in range(S):
            pos = int(full_regions[r, 1]) - 200
            while pos < int(full_regions[r, 2]) + 400:
                w, gap = int(rng.geometric(1 / 25)), int(rng.geometric(1 / 8))
                starts.append(pos + gap); ends.append(pos + gap + w); vals.append(float(rng.random() * 5)); pos += gap + w
            offs.append(len(starts))
    tracks = {"t": (np.array(starts, np.int32), np.array(ends, np.int32), np.array(vals, np.float32), np.array(offs, np.int64))}
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    _lib.load().gvl_set_debug_flags(dbg)
    try:
        ds = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=tracks, strategy_id=strategy, param=param, output_length=L,
                                     jitter=53, deterministic=False, onehot=False, haps=True, seed=3)
        n = 0
        for batch in ds.to_dataloader(batch_size=4, shuffle=True, seed=2, in_flight=2, group=1, python_loop=python_loop):
            idx = batch.idx.cpu().numpy()
            regions, shifts = batch.regions.cpu().numpy(), batch.shifts.cpu().numpy()
            goi = batch.geno_offset_idx.cpu().numpy()
            to_rc = batch.to_rc.cpu().numpy().astype(bool)
            n += int((shifts > 0).sum())
            diffs = oracle.get_diffs_sparse(goi, gv, go, st.ilens, None, None, regions[:, 1], regions[:, 2], st.v_starts)
            tlen = (regions[:, 2] - regions[:, 1]).astype(np.int64) - np.minimum(diffs.min(axis=1), 0)
            track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
            out_offsets = np.arange(len(idx) * P + 1, dtype=np.int64) * L
            a, e, v, io = tracks["t"]
            exp = np.zeros(len(idx) * P * L, np.float32)
            oracle.intervals_and_realign_track_fused(exp, out_offsets, regions, shifts, goi, gv, go, st.v_starts, st.ilens,
                                                     idx.astype(np.int64), a, e, v, io, track_offsets, np.array([param]), strategy,
                                                     0, None, None, to_rc)
            got = batch.tracks[:, 0].contiguous().cpu().numpy().ravel()
            np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))
        assert n > 0                      # (some rows start inside their shift)
    finally:
        _lib.load().gvl_set_debug_flags(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 2, 3, 4])
def test_tracks_batch_standalone_any_ploidy(oracle, P):
    """A stand-alone gvl_tracks_batch (the Python submit loop's call) on rows of several chunks: its sizing launch takes a wave per
    (query, haplotype) for ploidy 1 / 2 / 4 (the haplotypes of a query meet in LDS) and a wave per query otherwise (3); two tracks,
    batches that do not fill their last workgroup -- against the oracle, with the rows' plans and without (GVL_DBG 268435456)."""
    from genvarloader_amd import HapsDevice, _lib
    from genvarloader_amd.loader import DeviceHapsTracksDataset

    R, S, L = 3, 5, 9000
    st, full_regions, go, gv = _grid_dataset(500 + P, R, S, P, L, contig=300_000, indel_frac=0.6, slack=120)
    rng = np.random.default_rng(P)
    tracks = {}
    for name in ("a", "b"):
        starts, ends, vals, offs = [], [], [], [0]
        for r in range(R):
            for s_ in range(S):
                pos = int(full_regions[r, 1]) - 150
                while pos < int(full_regions[r, 2]) + 300:
                    w, gap = int(rng.geometric(1 / 30)), int(rng.geometric(1 / 6))
                    starts.append(pos + gap); ends.append(pos + gap + w); vals.append(float(rng.random() * 5)); pos += gap + w
                offs.append(len(starts))
        tracks[name] = (np.array(starts, np.int32), np.array(ends, np.int32), np.array(vals, np.float32), np.array(offs, np.int64))
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    seen = {}
    try:
        for dbg in (0, 268435456):
            _lib.load().gvl_set_debug_flags(dbg)
            ds = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=tracks, strategy_id=4, param=2.0, output_length=L,
                                         jitter=21, onehot=False, haps=True, seed=3)
            assert all(ds._tile_complete)
            for bi, batch in enumerate(ds.to_dataloader(batch_size=7, shuffle=True, seed=2, python_loop=True)):
                idx = batch.idx.cpu().numpy()
                regions, shifts = batch.regions.cpu().numpy(), batch.shifts.cpu().numpy()
                goi = batch.geno_offset_idx.cpu().numpy()
                to_rc = batch.to_rc.cpu().numpy().astype(bool)
                diffs = oracle.get_diffs_sparse(goi, gv, go, st.ilens, None, None, regions[:, 1], regions[:, 2], st.v_starts)
                tlen = (regions[:, 2] - regions[:, 1]).astype(np.int64) - np.minimum(diffs.min(axis=1), 0)
                track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
                out_offsets = np.arange(len(idx) * P + 1, dtype=np.int64) * L
                for t, name in enumerate(("a", "b")):
                    a, e, v, io = tracks[name]
                    exp = np.zeros(len(idx) * P * L, np.float32)
                    oracle.intervals_and_realign_track_fused(exp, out_offsets, regions, shifts, goi, gv, go, st.v_starts, st.ilens,
                                                             idx.astype(np.int64), a, e, v, io, track_offsets, np.array([2.0]), 4,
                                                             0, None, None, to_rc)
                    got = batch.tracks[:, t].contiguous().cpu().numpy().ravel()
                    np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32), err_msg=f"dbg {dbg} batch {bi} track {name}")
                    seen.setdefault((bi, name), []).append(got)
    finally:
        _lib.load().gvl_set_debug_flags(-1)
    assert len(seen) == 6 and all(len(v) == 2 and np.array_equal(v[0].view(np.uint32), v[1].view(np.uint32)) for v in seen.values())


@pytest.fixture(params=[0, -1, 134217728, 67108864, 33554432],
                ids=["default-sized-per-epoch", "sizing-per-group", "sizing-per-batch", "no-pipelined-kernel", "pipelined-one-workgroup"])
def ragged_path(request):
    """Ragged rows reach their output through the lean kernel's pipelined form with their offsets sized ONCE PER EPOCH in the loader's
    table (default), behind one sizing per group of batches (round 4: gvl_set_tuning(GVL_TUNE_RAGGED_SIZING, 1)), behind a sizing per
    batch, through the all-purpose kernel, and through the pipelined form on one workgroup (many rows per wave)."""
    from genvarloader_amd import _lib

    lib = _lib.load()
    if request.param == -1:
        _lib.set_tuning(_lib.TUNE_RAGGED_SIZING, 1)
    else:
        lib.gvl_set_debug_flags(int(request.param))
    yield request.param
    lib.gvl_set_debug_flags(-1)
    _lib.set_tuning(_lib.TUNE_RAGGED_SIZING, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("python_loop", [False, True], ids=["native-ring", "python-loop"])
def test_loader_ragged_and_annotated_modes(oracle, python_loop, ragged_path):
    """The reference loader's other haplotype outputs from dataset indices: ragged rows (its default,
    _haps.py:794-811) and annotated haplotypes (ffi/mod.rs:2237-2397), through the native ring (rows packed
    inside a slot of fixed capacity, sizes stay on the device) and through the Python submit loop
    (exactly-sized batches); sharded across two ranks."""
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 5, 7, 2, 400
    st, full_regions, go, gv = _grid_dataset(21, R, S, P, L, indel_frac=0.5)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)

    def request(idx):
        r_idx, s_idx = np.unravel_index(idx, (R, S))
        regions = full_regions[r_idx]
        goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
        return regions, goi, np.repeat(regions[:, 3] == -1, P), np.zeros_like(goi, dtype=np.int32)

    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=-1, onehot=True, haps=True)
    seen, lens = [], []
    for rank in range(2):
        for batch in ds.to_dataloader(batch_size=6, shuffle=True, seed=4, rank=rank, world_size=2, drop_last=False,
                                      python_loop=python_loop, group=2):
            idx = batch.idx.cpu().numpy()
            regions, goi, to_rc, shifts = request(idx)
            exp, exp_off, exp_oh = oracle.reconstruct_haplotypes_fused(
                regions, shifts, goi, go, gv, st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref,
                st.ref_offsets, st.pad_char, -1, None, None, to_rc, False, onehot=True)
            np.testing.assert_array_equal(batch.out_offsets.cpu().numpy(), exp_off)
            tot = len(exp)
            if python_loop:
                assert batch.sizes is None and batch.haps.numel() == tot
            else:       # a view of the slot's capacity; {total, longest row} on the device
                assert batch.sizes.cpu().tolist() == [tot, int(np.diff(exp_off).max())]
                assert batch.haps.numel() == len(idx) * P * ds.max_row_len() >= tot
            np.testing.assert_array_equal(batch.haps[:tot].cpu().numpy(), exp)
            np.testing.assert_array_equal(batch.onehot[:tot].cpu().numpy(), exp_oh.reshape(-1, 4))
            seen.extend(idx.tolist()); lens.extend(np.diff(exp_off).tolist())
    assert sorted(set(seen)) == list(range(R * S)) and len(set(lens)) > 5          # every index, genuinely ragged

    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, onehot=False, annotate=True)
    n = 0
    for batch in ds.to_dataloader(batch_size=9, shuffle=False, python_loop=python_loop):
        idx = batch.idx.cpu().numpy()
        regions, goi, to_rc, shifts = request(idx)
        exp, av, ap, _ = oracle.reconstruct_annotated_haplotypes_fused(
            regions, shifts, goi, go, gv, st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref,
            st.ref_offsets, st.pad_char, L, None, None, to_rc, False)
        assert batch.haps.shape == (len(idx), P, L)
        np.testing.assert_array_equal(batch.haps.cpu().numpy().ravel(), exp)
        np.testing.assert_array_equal(batch.annot_v_idxs.cpu().numpy().ravel(), av)
        np.testing.assert_array_equal(batch.annot_ref_pos.cpu().numpy().ravel(), ap)
        n += len(idx)
    assert n == R * S

    # ragged AND annotated
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=-1, onehot=False, annotate=True)
    n = 0
    for batch in ds.to_dataloader(batch_size=8, shuffle=True, seed=1, python_loop=python_loop, in_flight=2, group=1,
                                  threaded=not python_loop):          # (the library's producer thread submits the batches)
        idx = batch.idx.cpu().numpy()
        regions, goi, to_rc, shifts = request(idx)
        exp, av, ap, exp_off = oracle.reconstruct_annotated_haplotypes_fused(
            regions, shifts, goi, go, gv, st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref,
            st.ref_offsets, st.pad_char, -1, None, None, to_rc, False)
        tot = len(exp)
        np.testing.assert_array_equal(batch.out_offsets.cpu().numpy(), exp_off)
        np.testing.assert_array_equal(batch.haps[:tot].cpu().numpy(), exp)
        np.testing.assert_array_equal(batch.annot_v_idxs[:tot].cpu().numpy(), av)
        np.testing.assert_array_equal(batch.annot_ref_pos[:tot].cpu().numpy(), ap)
        n += len(idx)
    assert n == R * S


@pytest.mark.gpu
@pytest.mark.parametrize("dbg", [0, 1048576], ids=["chunked-lean-kernel-ragged-form", "all-purpose-kernel"])
@pytest.mark.parametrize("python_loop", [False, True], ids=["native-ring", "python-loop"])
def test_loader_ragged_long_rows(oracle, python_loop, dbg):
    """Ragged rows (output_length = -1) of ~6 000 bases -- longer than the pipelined kernel's ragged form takes -- from dataset
    indices: the chunked lean kernel's ragged form (recon_lean_kernel<.., LONG, RAGL>) behind the loader's sizing, against the
    oracle; GVL_DBG = 1048576: the all-purpose kernel, as before round 4."""
    from genvarloader_amd import HapsDevice, _lib
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 4, 5, 2, 6000
    st, full_regions, go, gv = _grid_dataset(23, R, S, P, L, contig=120_000, indel_frac=0.5)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    _lib.load().gvl_set_debug_flags(dbg)
    try:
        ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=-1, onehot=True, haps=True)
        seen, lens = [], []
        for batch in ds.to_dataloader(batch_size=6, shuffle=True, seed=4, python_loop=python_loop, group=2):
            idx = batch.idx.cpu().numpy()
            r_idx, s_idx = np.unravel_index(idx, (R, S))
            regions = full_regions[r_idx]
            goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
            to_rc, shifts = np.repeat(regions[:, 3] == -1, P), np.zeros_like(goi, dtype=np.int32)
            exp, exp_off, exp_oh = oracle.reconstruct_haplotypes_fused(
                regions, shifts, goi, go, gv, st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref,
                st.ref_offsets, st.pad_char, -1, None, None, to_rc, False, onehot=True)
            tot = len(exp)
            np.testing.assert_array_equal(batch.out_offsets.cpu().numpy(), exp_off)
            np.testing.assert_array_equal(batch.haps[:tot].cpu().numpy(), exp)
            np.testing.assert_array_equal(batch.onehot[:tot].cpu().numpy(), exp_oh.reshape(-1, 4))
            seen.extend(idx.tolist()); lens.extend(np.diff(exp_off).tolist())
        assert sorted(seen) == list(range(R * S)) and min(lens) > 2560 and len(set(x % 4 for x in lens)) > 1
    finally:
        _lib.load().gvl_set_debug_flags(-1)


@pytest.mark.gpu
def test_native_ragged_rows_longer_than_the_slot_bound_are_reported(oracle):
    """A ragged slot reserves max_row_len bases per row.  With a bound that is too small the rows are cut
    to it (nothing is written outside the slot) and the cut is reported like a sticky HIP error."""
    from genvarloader_amd import HapsDevice, _lib
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 4, 3, 2, 300
    st, full_regions, go, gv = _grid_dataset(5, R, S, P, L, indel_frac=0.5)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=-1, onehot=False, haps=True)
    true_bound = ds.max_row_len()
    ds._max_row_len = 200                        # every region is 300 long
    lib = _lib.load()
    lib.gvl_async_error(1)
    n_seen = 0
    with pytest.raises(ValueError, match="max_row_len"):       # the loader polls the flag when the epoch ends
        for batch in ds.to_dataloader(batch_size=5, shuffle=False):
            assert int(batch.sizes[1]) == 200 and int(batch.sizes[0]) == 200 * batch.idx.numel() * P
            n_seen += 1
    assert n_seen == -(-R * S // 5)                               # ... after every batch was delivered
    torch.cuda.synchronize()
    assert lib.gvl_async_error(1) == 0                            # (the poll cleared it)
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=-1, onehot=False, haps=True)
    assert ds.max_row_len() == true_bound >= L
    for batch in ds.to_dataloader(batch_size=5, shuffle=False):
        assert int(batch.sizes[1]) <= true_bound
    torch.cuda.synchronize()
    assert lib.gvl_async_error(1) == 0


@pytest.mark.gpu
def test_loader_draws_depend_on_seed_epoch_and_index_only():
    """The jitter / shift draw of dataset index i in epoch e (``_query.py:160-187``, ``_haps.py:720-730``) is the
    same whatever rank (world_size 1, 2, 8), batch size or submit loop (native ring, producer thread, Python
    loop) delivers the index -- what makes an 8-way sharded epoch equal to the 1-GPU epoch -- and differs
    between epochs and between draw streams."""
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 6, 9, 2, 256
    st, full_regions, go, gv = _grid_dataset(21, R, S, P, L + 64, indel_frac=0.6, slack=0)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, jitter=8, deterministic=False, seed=5, onehot=True, haps=False)

    def draws(epoch, world=1, bs=7, stream=1, **kw):
        got = {}
        for rank in range(world):
            dl = ds.to_dataloader(batch_size=bs, shuffle=True, rank=rank, world_size=world, seed=3, draw_stream=stream, **kw)
            dl.set_epoch(epoch)
            for b in dl:
                idx, reg, sh = b.idx.cpu().numpy(), b.regions.cpu().numpy(), b.shifts.cpu().numpy()
                for i, q in enumerate(idx.tolist()):
                    row = (int(reg[i, 1]), int(reg[i, 2]), tuple(int(x) for x in sh[i]))
                    assert got.setdefault(q, row) == row        # (an index padded into two ranks' shares draws alike)
        assert set(got) == set(range(R * S))
        return got

    ref = draws(4)
    assert len({v[0] - int(full_regions[q // S, 1]) for q, v in ref.items()}) > 3        # jitter is on
    assert any(any(v[2]) for v in ref.values())                                            # ... and so are the shifts
    for kw in (dict(world=2), dict(world=8, bs=2), dict(bs=3), dict(bs=16), dict(threaded=True), dict(python_loop=True),
               dict(world=2, bs=5, python_loop=True), dict(group=2, bs=4)):
        assert draws(4, **kw) == ref, kw
    assert draws(4) == ref                       # the same loader configuration again: reproducible
    other_epoch, other_stream = draws(5), draws(4, stream=2)
    assert sum(other_epoch[q] != ref[q] for q in ref) > len(ref) // 2
    assert sum(other_stream[q] != ref[q] for q in ref) > len(ref) // 2
    a, b = ds.to_dataloader(batch_size=7), ds.to_dataloader(batch_size=7)        # default: a stream per loader
    assert a.draw_seed != b.draw_seed


@pytest.mark.gpu
def test_loader_onehot_only_goes_through_the_lean_kernel(oracle):
    """One-hot only (the benchmark's output): the ring's batches run recon_lean_kernel; bit-exact against the
    oracle incl. reverse-complemented rows, and identical to the all-purpose kernel (GVL_DBG 16384)."""
    from genvarloader_amd import HapsDevice, _lib
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 8, 11, 2, 512
    st, full_regions, go, gv = _grid_dataset(22, R, S, P, L, indel_frac=0.4)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    assert dev.ref4 is not None and dev.slot_rec is not None
    ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, onehot=True, haps=False)
    lib = _lib.load()
    outs = {}
    try:
        for flags in (0, 16384):
            lib.gvl_set_debug_flags(flags)
            got = {}
            for b in ds.to_dataloader(batch_size=9, shuffle=True, seed=1):
                oh = b.onehot.cpu().numpy()
                for i, q in enumerate(b.idx.cpu().numpy().tolist()):
                    got[q] = oh[i].copy()
            outs[flags] = got
    finally:
        lib.gvl_set_debug_flags(-1)
    idx = np.arange(R * S)
    r_idx, s_idx = np.unravel_index(idx, (R, S))
    goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
    _, _, exp_oh = oracle.reconstruct_haplotypes_fused(
        full_regions[r_idx], np.zeros_like(goi, dtype=np.int32), goi, go, gv, st.v_starts, st.ilens, st.alt_alleles,
        st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, L, None, None, np.repeat(full_regions[r_idx, 3] == -1, P), False,
        onehot=True)
    exp_oh = exp_oh.reshape(R * S, P, L, 4)
    for q in idx.tolist():
        np.testing.assert_array_equal(outs[0][q], exp_oh[q])
        np.testing.assert_array_equal(outs[16384][q], exp_oh[q])


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["lc", "cl"])
def test_loader_long_fixed_rows_with_and_without_epoch_chunk_plans(oracle, layout):
    """Fixed-length rows of several chunks through the native ring (config 4's haplotype half), one-hot row-major and channel-major:
    with the epoch's chunk plans (gvl_hap_plan over the epoch table) and without them (GVL_DBG 536870912) == the oracle."""
    from genvarloader_amd import HapsDevice, _lib
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 5, 6, 2, 6_148
    st, full_regions, go, gv = _grid_dataset(23, R, S, P, L, indel_frac=0.3)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    assert dev.ref4 is not None and dev.geno_rec is not None
    idx = np.arange(R * S)
    r_idx, s_idx = np.unravel_index(idx, (R, S))
    goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
    exp, _, exp_oh = oracle.reconstruct_haplotypes_fused(
        full_regions[r_idx], np.zeros_like(goi, dtype=np.int32), goi, go, gv, st.v_starts, st.ilens, st.alt_alleles,
        st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, L, None, None, np.repeat(full_regions[r_idx, 3] == -1, P), False,
        onehot=True)
    exp = exp.reshape(R * S, P, L)
    exp_oh = exp_oh.reshape(R * S, P, L, 4)
    if layout == "cl":
        exp_oh = exp_oh.transpose(0, 1, 3, 2)
    lib = _lib.load()
    try:
        for flags, cap_mb in ((0, 0), (536870912, 0)):       # (a cap below the epoch's plans: test_loader_knobs_turned_between_epochs)
            lib.gvl_set_debug_flags(flags)
            _lib.set_tuning(_lib.TUNE_HAP_PLAN_MAX_MB, cap_mb)
            ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, onehot=True, haps=True, layout=layout)
            seen = 0
            for epoch in range(2):
                for b in ds.to_dataloader(batch_size=7, shuffle=True, seed=3):
                    oh, hp = b.onehot.cpu().numpy(), b.haps.cpu().numpy()
                    for i, q in enumerate(b.idx.cpu().numpy().tolist()):
                        np.testing.assert_array_equal(hp[i], exp[q], err_msg=f"flags {flags} cap {cap_mb} query {q}")
                        np.testing.assert_array_equal(oh[i], exp_oh[q], err_msg=f"flags {flags} cap {cap_mb} query {q}")
                        seen += 1
            assert seen == 2 * R * S
    finally:
        lib.gvl_set_debug_flags(-1)
        _lib.set_tuning(_lib.TUNE_HAP_PLAN_MAX_MB, 0)


@pytest.mark.gpu
def test_tracks_tile_complete_claim_is_checked():
    """gvl_track_set.tile_complete lets the tracks be realigned straight from the intervals (and the painter, where it
    still runs, skip its second launch).  The dataset only sets it for interval sets that qualify (no overlaps, distinct
    starts, <= 256 intervals per two adjacent buckets); a WRONG claim is never silently wrong: the realignment falls
    back to exact per-position lookups, the painter reports it through gvl_async_error."""
    from genvarloader_amd import HapsDevice, _lib
    from genvarloader_amd.loader import DeviceHapsTracksDataset

    R, S, P, L = 3, 4, 2, 600
    st, full_regions, go, gv = _grid_dataset(43, R, S, P, L, indel_frac=0.3, slack=30)
    rng = np.random.default_rng(6)

    def lists(overlap):
        starts, ends, vals, offs = [], [], [], [0]
        for r in range(R):
            for s_ in range(S):
                pos = int(full_regions[r, 1]) - 20
                while pos < int(full_regions[r, 2]) + 20:
                    w = int(rng.integers(5, 40))
                    starts.append(pos); ends.append(pos + w + (15 if overlap else 0)); vals.append(float(rng.random()))
                    pos += w
                offs.append(len(starts))
        return {"t": (np.array(starts, np.int32), np.array(ends, np.int32), np.array(vals, np.float32), np.array(offs, np.int64))}

    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    lib = _lib.load()
    lib.gvl_async_error(1)
    good = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=lists(False), output_length=L, onehot=False, haps=True)
    bad = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=lists(True), output_length=L, onehot=False, haps=True)
    assert good._tile_complete == [True] and bad._tile_complete == [False]
    for _ in good.to_dataloader(batch_size=5):
        pass
    honest = [b.tracks.cpu().numpy().copy() for b in bad.to_dataloader(batch_size=5)]      # overlapping intervals, honestly
    torch.cuda.synchronize()                                                                # declared: the leftovers launch
    assert lib.gvl_async_error(1) == 0
    bad._track_sets[0].tile_complete = 1           # a wrong claim
    # ... read by the realignment straight from the intervals: a window whose candidates overlap is not used, every
    # value is looked up in the list itself -- slower, never wrong
    claimed = [b.tracks.cpu().numpy().copy() for b in bad.to_dataloader(batch_size=5)]
    assert lib.gvl_async_error(1) == 0
    assert len(claimed) == len(honest)
    for x, y in zip(claimed, honest):
        np.testing.assert_array_equal(x, y)
    # ... read by the painter (GVL_DBG 4194304: tracks painted into the scratch track first): reported
    lib.gvl_set_debug_flags(4194304)
    try:
        with pytest.raises(ValueError, match="tile_complete"):
            for _ in bad.to_dataloader(batch_size=5):
                torch.cuda.synchronize()
    finally:
        lib.gvl_set_debug_flags(-1)
    lib.gvl_async_error(1)


@pytest.mark.gpu
def test_sample_sharded_genotypes_give_the_replicated_dataset(oracle):
    """SURVEY 8(e), the alternative to replicas: each rank holds the genotype CSR of ITS samples only
    (sharding.shard_genotypes_by_sample) and iterates the (regions x owned samples) grid; mapped back to the full grid its
    batches are bit-identical to the replicated dataset's, every index is served by exactly one rank, and `pad_to` lets
    ranks with fewer samples run the same number of batches."""
    from genvarloader_amd import HapsDevice, sharding
    from genvarloader_amd.loader import DeviceHapsDataset

    R, S, P, L = 4, 7, 2, 512
    st, full_regions, go, gv = _grid_dataset(31, R, S, P, L, indel_frac=0.4)
    kw = dict(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
              alt_offsets=st.alt_offsets, pad_char=st.pad_char)
    full = DeviceHapsDataset(HapsDevice(geno_offsets=go, geno_v_idxs=gv, **kw), full_regions, S, P, output_length=L,
                             onehot=True, haps=False)
    ref = {}
    for b in full.to_dataloader(batch_size=9):
        oh = b.onehot.cpu().numpy()
        for i, q in enumerate(b.idx.cpu().numpy().tolist()):
            ref[q] = oh[i].copy()
    world = 3
    n_max = R * -(-S // world)
    served = np.zeros(R * S, np.int32)
    for rank in range(world):
        lo, lv, (s0, s1) = sharding.shard_genotypes_by_sample(go, gv, R, S, P, world, rank)
        assert lv.size < gv.size
        ds = DeviceHapsDataset(HapsDevice(geno_offsets=lo, geno_v_idxs=lv, **kw), full_regions, s1 - s0, P, output_length=L,
                               onehot=True, haps=False)
        dl = ds.to_dataloader(batch_size=5, shuffle=True, seed=2, pad_to=n_max)
        assert len(dl) == -(-n_max // 5)
        n_seen, mine = 0, set()
        for b in dl:
            gidx = sharding.global_index(b.idx, S, s0, s1).cpu().numpy()
            assert (sharding.owner_of(gidx, S, world) == rank).all()
            oh = b.onehot.cpu().numpy()
            for i, q in enumerate(gidx.tolist()):
                np.testing.assert_array_equal(oh[i], ref[q])
                mine.add(q)
            n_seen += len(gidx)
        assert n_seen == n_max and len(mine) == R * (s1 - s0)
        served[list(mine)] += 1
    assert (served == 1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("python_loop", [False, True], ids=["native-ring", "python-loop"])
def test_tracks_with_their_own_fill_and_region_level_lists(oracle, python_loop):
    """The reference lowers ONE insertion fill per track and indexes region-level (non-SAMPLE) tracks by r_idx
    (_reconstruct.py:204-236): a dataset with a per-sample Repeat5p track, a per-sample Constant-fill track and a
    region-level Interpolate track, each against the oracle with its own strategy / parameter / list index."""
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsTracksDataset

    R, S, P, L = 4, 5, 2, 640
    st, full_regions, go, gv = _grid_dataset(47, R, S, P, L, indel_frac=0.5, slack=20)
    rng = np.random.default_rng(9)

    def lists(n_lists, per_list_region):
        starts, ends, vals, offs = [], [], [], [0]
        for i in range(n_lists):
            r = per_list_region(i)
            pos = int(full_regions[r, 1]) - int(rng.integers(0, 50))
            while pos < int(full_regions[r, 2]) + 30:
                w, gap = int(rng.geometric(1 / 15)), int(rng.integers(0, 5))
                starts.append(pos + gap); ends.append(pos + gap + w); vals.append(float(rng.random() * 3)); pos += gap + w
            offs.append(len(starts))
        return np.array(starts, np.int32), np.array(ends, np.int32), np.array(vals, np.float32), np.array(offs, np.int64)

    cov, atac, cons = lists(R * S, lambda i: i // S), lists(R * S, lambda i: i // S), lists(R, lambda i: i)
    tracks = {"cov": cov,                                                                              # the dataset's fill
              "atac": dict(starts=atac[0], ends=atac[1], values=atac[2], offsets=atac[3], fill=(2, 7.5)),          # Constant 7.5
              "cons": dict(starts=cons[0], ends=cons[1], values=cons[2], offsets=cons[3], fill=(4, 3.0), region_level=True)}
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    ds = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=tracks, strategy_id=0, param=0.0, base_seed=3,
                                 output_length=L, onehot=False, haps=True)
    spec = {"cov": (cov, 0, 0.0, False), "atac": (atac, 2, 7.5, False), "cons": (cons, 4, 3.0, True)}
    seen = 0
    for batch in ds.to_dataloader(batch_size=6, shuffle=True, seed=1, python_loop=python_loop):
        idx = batch.idx.cpu().numpy()
        r_idx, s_idx = np.unravel_index(idx, (R, S))
        regions = full_regions[r_idx]
        goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
        to_rc = np.repeat(regions[:, 3] == -1, P)
        shifts = np.zeros_like(goi, dtype=np.int32)
        diffs = oracle.get_diffs_sparse(goi, gv, go, st.ilens, None, None, regions[:, 1], regions[:, 2], st.v_starts)
        tlen = (regions[:, 2] - regions[:, 1]).astype(np.int64) - np.minimum(diffs.min(axis=1), 0)
        track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
        out_offsets = np.arange(len(idx) * P + 1, dtype=np.int64) * L
        for t, name in enumerate(("cov", "atac", "cons")):
            (a, e, v, io), strategy, param, by_region = spec[name]
            exp = np.zeros(len(idx) * P * L, np.float32)
            oracle.intervals_and_realign_track_fused(
                exp, out_offsets, regions, shifts, goi, gv, go, st.v_starts, st.ilens,
                (r_idx if by_region else idx).astype(np.int64), a, e, v, io, track_offsets, np.array([param]), strategy, 3,
                None, None, to_rc)
            got = batch.tracks[:, t].contiguous().cpu().numpy().ravel()
            np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32), err_msg=name)
        seen += len(idx)
    assert seen == R * S


@pytest.mark.gpu
@pytest.mark.timeout(180)
@pytest.mark.parametrize("tracks", [False, True], ids=["haps", "haps+tracks"])
def test_epochs_prepared_ahead_deliver_what_epochs_prepared_at_their_start_do(tracks):
    """The native loop fills the NEXT epoch's table while the running epoch's batches are in flight
    (``gvl_loader_prefetch_epoch``).  Chained epochs, a ``set_epoch`` jump (the prepared epoch is not the one that
    starts: it is dropped), an abandoned epoch and a change of batch size must deliver exactly the batches of a
    loader that prepares every epoch at its start -- indices, draws, one-hot and tracks."""
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceHapsDataset, DeviceHapsTracksDataset, DeviceLoader

    R, S, P, L = 5, 7, 2, 512
    st, full_regions, go, gv = _grid_dataset(33, R, S, P, L + 64, indel_frac=0.5, slack=0)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    if tracks:
        rng = np.random.default_rng(3)
        starts, ends, vals, offs = [], [], [], [0]
        for r in range(R):
            for s_ in range(S):
                n = int(rng.integers(5, 40))
                s0 = np.sort(rng.integers(int(full_regions[r, 1]) - 50, int(full_regions[r, 2]) + 50, n)).astype(np.int32)
                starts.append(s0); ends.append(s0 + rng.integers(1, 30, n).astype(np.int32)); vals.append(rng.random(n).astype(np.float32))
                offs.append(offs[-1] + n)
        tr = {"t": (np.concatenate(starts), np.concatenate(ends), np.concatenate(vals), np.asarray(offs, np.int64))}
        ds = DeviceHapsTracksDataset(dev, full_regions, S, P, tracks=tr, output_length=L, jitter=4, deterministic=False, seed=9,
                                     onehot=True, haps=False)
    else:
        ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, jitter=4, deterministic=False, seed=9, onehot=True, haps=False)

    def run(prefetch):
        dl = DeviceLoader(ds, batch_size=4, shuffle=True, seed=11, in_flight=3, group=1, draw_stream=7)
        dl.prefetch_epochs = prefetch
        seen = []

        def epoch(stop_after=None):
            for i, b in enumerate(dl):
                item = [b.idx.cpu().numpy().copy(), b.regions.cpu().numpy().copy(), b.shifts.cpu().numpy().copy(),
                        b.onehot.cpu().numpy().copy()]
                if tracks:
                    item.append(b.tracks.cpu().numpy().copy())
                seen.append(item)
                if stop_after is not None and i == stop_after:
                    break

        epoch(); epoch()                      # chained: the second one was prepared ahead
        dl.set_epoch(7); epoch()              # a jump: what was prepared (epoch 2) is dropped
        epoch(stop_after=2)                   # abandoned after three batches ...
        epoch()                               # ... and the one prepared behind it starts all the same
        dl.set_epoch(3); epoch(); epoch()
        return seen

    a, b = run(True), run(False)
    assert len(a) == len(b) and len(a) > 50
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            np.testing.assert_array_equal(u, v)
    # (and epochs differ from each other: the comparison above is not one batch list seven times)
    assert not np.array_equal(a[0][0], a[9][0])


@pytest.mark.gpu
def test_loader_knobs_turned_between_epochs(oracle):
    """The knobs an epoch table's layout depends on are read ONCE per table fill and kept with the table: turning one
    between an epoch's prefetch and its start -- the documented A/B use -- makes the loader fill that table again instead of reading a
    layout that is not there.  Long fixed rows whose chunk plans (1000 rows x 4 chunks x 336 B = 1.34 MB) exceed a 1 MB cap: epochs
    alternate between "with plans" and "the cap drops them" (the plan part of the table is then EMPTY, the branch no test reached
    before), then between GVL_DBG 536870912 on and off; ragged rows alternate between sizing per epoch and per group.  Every epoch
    == the oracle."""
    import ctypes as C

    from genvarloader_amd import HapsDevice, _lib
    from genvarloader_amd.loader import DeviceHapsDataset

    lib = _lib.load()
    R, S, P, L = 10, 50, 2, 6_148
    st, full_regions, go, gv = _grid_dataset(29, R, S, P, L, indel_frac=0.3)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv, pad_char=st.pad_char)
    idx = np.arange(R * S)
    r_idx, s_idx = np.unravel_index(idx, (R, S))
    goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
    to_rc = np.repeat(full_regions[r_idx, 3] == -1, P)
    exp = oracle.reconstruct_haplotypes_fused(
        full_regions[r_idx], np.zeros_like(goi, dtype=np.int32), goi, go, gv, st.v_starts, st.ilens, st.alt_alleles,
        st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, L, None, None, to_rc, False)[0].reshape(R * S, P, L)
    try:
        ds = DeviceHapsDataset(dev, full_regions, S, P, output_length=L, onehot=False, haps=True)
        dl = ds.to_dataloader(batch_size=16, shuffle=True, seed=5)
        assert dl.prefetch_epochs
        plan_parts = []
        for epoch, (cap_mb, flags) in enumerate([(0, 0), (1, 0), (0, 0), (1, 0), (0, 536870912), (0, 0), (0, 536870912)]):
            # (turned AFTER the previous epoch prefetched this epoch's table under the other setting)
            _lib.set_tuning(_lib.TUNE_HAP_PLAN_MAX_MB, cap_mb)
            lib.gvl_set_debug_flags(flags)
            po = (C.c_int64 * _lib.LOADER_TABLE_PARTS)()
            lib.gvl_loader_table_bytes(C.byref(dl._native["cfg"]), C.c_int64(R * S), po) if dl._native else None
            plan_parts.append(int(po[9]) - int(po[8]) if dl._native else None)
            seen = 0
            for b in dl:
                hp = b.haps.cpu().numpy()
                for i, q in enumerate(b.idx.cpu().numpy().tolist()):
                    np.testing.assert_array_equal(hp[i], exp[q], err_msg=f"epoch {epoch} cap {cap_mb} flags {flags} query {q}")
                    seen += 1
            assert seen == R * S
            torch.cuda.synchronize()
            _lib.check_async()
        assert plan_parts[1] == 0 and plan_parts[2] > (1 << 20) and plan_parts[3] == 0       # the cap really drops the plans
        lib.gvl_set_debug_flags(-1)
        _lib.set_tuning(_lib.TUNE_HAP_PLAN_MAX_MB, 0)
        # ragged rows: offsets sized once per epoch (in the table) / once per group of batches
        R2, S2, L2 = 6, 20, 900
        st2, fr2, go2, gv2 = _grid_dataset(31, R2, S2, P, L2, indel_frac=0.3)
        dev2 = HapsDevice(ref=st2.ref, ref_offsets=st2.ref_offsets, v_starts=st2.v_starts, ilens=st2.ilens,
                          alt_alleles=st2.alt_alleles, alt_offsets=st2.alt_offsets, geno_offsets=go2, geno_v_idxs=gv2, pad_char=st2.pad_char)
        idx2 = np.arange(R2 * S2)
        r2, s2 = np.unravel_index(idx2, (R2, S2))
        goi2 = np.ravel_multi_index((r2[:, None], s2[:, None], np.arange(P)), (R2, S2, P))
        e2, o2 = oracle.reconstruct_haplotypes_fused(
            fr2[r2], np.zeros_like(goi2, dtype=np.int32), goi2, go2, gv2, st2.v_starts, st2.ilens, st2.alt_alleles,
            st2.alt_offsets, st2.ref, st2.ref_offsets, st2.pad_char, -1, None, None, np.repeat(fr2[r2, 3] == -1, P), False)[:2]
        ds2 = DeviceHapsDataset(dev2, fr2, S2, P, output_length=-1, onehot=False, haps=True)
        dl2 = ds2.to_dataloader(batch_size=9, shuffle=True, seed=7, group=2)
        for epoch, mode in enumerate([0, 1, 0, 1, 1, 0]):
            _lib.set_tuning(_lib.TUNE_RAGGED_SIZING, mode)
            seen = 0
            for b in dl2:
                hp, oo = b.haps.cpu().numpy(), b.out_offsets.cpu().numpy()
                for i, q in enumerate(b.idx.cpu().numpy().tolist()):
                    for p in range(P):
                        k = i * P + p
                        np.testing.assert_array_equal(hp[oo[k]:oo[k + 1]], e2[o2[q * P + p]:o2[q * P + p + 1]],
                                                      err_msg=f"epoch {epoch} sizing {mode} query {q} hap {p}")
                    seen += 1
            assert seen == R2 * S2
            torch.cuda.synchronize()
            _lib.check_async()
    finally:
        lib.gvl_set_debug_flags(-1)
        _lib.set_tuning(_lib.TUNE_HAP_PLAN_MAX_MB, 0)
        _lib.set_tuning(_lib.TUNE_RAGGED_SIZING, 0)
