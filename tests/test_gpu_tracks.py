"""GPU parity for the track half: reference goldens + synthetic cfg4-like
batches against the oracle, bit-exact on the f32 bit patterns."""

import numpy as np
import pytest

from tests._fixtures import load_ref_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ffi():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("gpu tests need a HIP device (no CPU fallback exists)")
    import genvarloader_amd.ffi as f

    return f


# the realignment walk has two implementations (wave-scan planner / scalar replay) and the
# variant records two sources: every reference vector goes down each of them
TRACK_PATHS = {0: "default", 8: "scalar-walk", 80: "csr-vrec-gather", 8192: "painter-image-path"}


@pytest.fixture(params=sorted(TRACK_PATHS), ids=[TRACK_PATHS[k] for k in sorted(TRACK_PATHS)])
def tpath(request, ffi):
    from genvarloader_amd import _lib

    lib = _lib.load()
    lib.gvl_set_debug_flags(int(request.param))
    yield request.param
    lib.gvl_set_debug_flags(-1)


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def test_golden_shift_and_realign_tracks_sparse(ffi, tpath):
    cases = load_ref_cases("shift_and_realign_tracks_sparse")
    assert len(cases) == 200
    for ci, (inp, exp) in enumerate(cases):
        out = np.full(int(inp[0][-1]), 7.0, np.float32)
        ffi.shift_and_realign_tracks_sparse(out, *inp)
        np.testing.assert_array_equal(bits(out), bits(exp), err_msg=f"case {ci} strategy {int(inp[13])}")


def test_golden_intervals_to_tracks(ffi, tpath):
    cases = load_ref_cases("intervals_to_tracks")
    assert len(cases) == 200
    for ci, (inp, exp) in enumerate(cases):
        out = np.full(int(inp[-1][-1]), 7.0, np.float32)
        ffi.intervals_to_tracks(*inp[:6], out, inp[6])
        np.testing.assert_array_equal(bits(out), bits(exp), err_msg=f"case {ci}")


def _track_batch(seed, q, L, contig, **kw):
    from genvarloader_amd import synth

    rng = np.random.default_rng(seed)
    st = synth.make_static(rng, (contig,), indel_frac=0.4, density=kw.pop("density", 1 / 60))
    bt = synth.make_batch(rng, st, q, 2, L, rc_frac=0.5, random_shifts=kw.pop("shifts", False),
                          output_length=kw.pop("output_length", None), slack=kw.pop("slack", 32))
    # one reference-coordinate track per query: sorted, non-overlapping intervals, width ~ Geom(1/25)
    B = bt.regions.shape[0]
    starts, ends, vals, offs = [], [], [], [0]
    for b in range(B):
        s0, e0 = int(bt.regions[b, 1]), int(bt.regions[b, 2])
        pos = s0 - int(rng.integers(0, 40))
        while pos < e0 + 20:
            w = int(rng.geometric(1 / 25))
            gap = int(rng.integers(0, 3))
            starts.append(pos + gap); ends.append(pos + gap + w); vals.append(float(rng.random() * 8))
            pos += gap + w
        offs.append(len(starts))
    itv = dict(offset_idxs=np.arange(B, dtype=np.int64), itv_starts=np.array(starts, np.int32),
               itv_ends=np.array(ends, np.int32), itv_values=np.array(vals, np.float32),
               itv_offsets=np.array(offs, np.int64))
    return st, bt, itv


@pytest.mark.parametrize("strategy,param", [(0, 0.0), (1, 0.0), (2, 3.5), (3, 4.0), (4, 1.0), (4, 3.0)])
def test_fused_tracks_synthetic(ffi, oracle, strategy, param, tpath):
    st, bt, itv = _track_batch(40 + strategy, 24, 3000, 200_000, shifts=(strategy % 2 == 0))
    B, P = bt.geno_offset_idx.shape
    L = bt.output_length
    # track length per query as the reference sizes it (_reconstruct.py:191): len - min_p(min(diff, 0))
    diffs = oracle.get_diffs_sparse(bt.geno_offset_idx, bt.geno_v_idxs, bt.geno_offsets, st.ilens, None, None,
                                    bt.regions[:, 1], bt.regions[:, 2], st.v_starts)
    tlen = (bt.regions[:, 2] - bt.regions[:, 1]) - np.minimum(diffs.min(axis=1), 0)
    track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
    out_offsets = np.arange(B * P + 1, dtype=np.int64) * L
    args = (out_offsets, bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_v_idxs, bt.geno_offsets, st.v_starts,
            st.ilens, itv["offset_idxs"], itv["itv_starts"], itv["itv_ends"], itv["itv_values"], itv["itv_offsets"],
            track_offsets, np.array([param]), strategy, 12345, None, None, bt.to_rc)
    exp = np.full(B * P * L, 7.0, np.float32)
    oracle.intervals_and_realign_track_fused(exp, *args)
    got = np.full(B * P * L, 9.0, np.float32)
    ffi.intervals_and_realign_track_fused(got, *args)
    np.testing.assert_array_equal(bits(got), bits(exp))


def test_tracks_long_rows_chunked(ffi, oracle):
    st, bt, itv = _track_batch(77, 3, 40_000, 400_000, density=1 / 100)
    B, P = bt.geno_offset_idx.shape
    L = bt.output_length
    tlen = (bt.regions[:, 2] - bt.regions[:, 1]) + 64
    track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
    out_offsets = np.arange(B * P + 1, dtype=np.int64) * L
    args = (out_offsets, bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_v_idxs, bt.geno_offsets, st.v_starts,
            st.ilens, itv["offset_idxs"], itv["itv_starts"], itv["itv_ends"], itv["itv_values"], itv["itv_offsets"],
            track_offsets, np.array([2.0]), 3, 99, None, None, bt.to_rc)
    exp = np.zeros(B * P * L, np.float32)
    oracle.intervals_and_realign_track_fused(exp, *args)
    got = np.zeros(B * P * L, np.float32)
    ffi.intervals_and_realign_track_fused(got, *args)
    np.testing.assert_array_equal(bits(got), bits(exp))


def test_cfg4_full_haps_and_track(ffi, oracle, tpath):
    """BASELINE configs[3]: 256 windows x 131072 bp, SNP+indel, haplotype one-hot + one
    realigned track (Repeat5p), full size."""
    import time

    import torch

    from genvarloader_amd import HapsDevice, device, synth

    st, bt = synth.make_config("cfg4", contig=32 << 20)
    B, P = bt.geno_offset_idx.shape
    L = bt.output_length
    assert B * P == 256 and L == 131072
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets,
                     geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)
    out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=bt.to_rc, haps=True, onehot=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=bt.to_rc, haps=True, onehot=True)
    torch.cuda.synchronize()
    t_h = time.perf_counter() - t0
    exp, exp_off, exp_oh = oracle.reconstruct_haplotypes_fused(
        bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs, st.v_starts, st.ilens,
        st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, L, None, None, bt.to_rc, True,
        onehot=True, n_threads=8)
    np.testing.assert_array_equal(out.haps.cpu().numpy(), exp)
    np.testing.assert_array_equal(out.onehot.cpu().numpy(), exp_oh)
    # one track per query: value changes every ~25 bp
    rng = np.random.default_rng(5)
    tlen = (bt.regions[:, 2] - bt.regions[:, 1]).astype(np.int64) + 4096
    track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
    tracks = np.repeat(rng.random(int(track_offsets[-1]) // 25 + 1).astype(np.float32) * 8, 25)[: int(track_offsets[-1])]
    out_offsets = np.arange(B * P + 1, dtype=np.int64) * L
    exp_t = np.zeros(B * P * L, np.float32)
    oracle.shift_and_realign_tracks_sparse(exp_t, out_offsets, bt.regions, bt.shifts, bt.geno_offset_idx,
                                           bt.geno_v_idxs, bt.geno_offsets, st.v_starts, st.ilens, tracks,
                                           track_offsets, np.array([0.0]), None, None, 0, 0)
    dtracks = torch.from_numpy(tracks).cuda()
    got = device.realign_tracks(dev, bt.regions, bt.shifts, bt.geno_offset_idx, out_offsets, dtracks,
                                track_offsets, np.array([0.0]), 0, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = device.realign_tracks(dev, bt.regions, bt.shifts, bt.geno_offset_idx, out_offsets, dtracks,
                                track_offsets, np.array([0.0]), 0, 0)
    torch.cuda.synchronize()
    t_t = time.perf_counter() - t0
    np.testing.assert_array_equal(bits(got.cpu().numpy()), bits(exp_t))
    print(f"\ncfg4: V/row={bt.mean_variants:.0f}  haps+onehot {t_h*1e3:.2f} ms  track {t_t*1e3:.2f} ms (host-timed, incl. upload)")


def test_cfg4_full_size_batch_through_the_native_ring(oracle):
    """BASELINE configs[3] as bench.py's cfg4 step runs it: ONE full-size batch (256 windows x 131 072 bp) from dataset
    indices through DeviceHapsTracksDataset's native ring -- the lean kernel's chunked form for the haplotypes (one-hot +
    bytes) and realign_tracks_kernel<PAINT> for the track (a BigWig-like, `tile_complete` interval set, realigned straight
    from its intervals) -- against the haplotype oracle and the oracle's fused paint + realign."""
    import torch

    import bench_cfg4

    R, S, P, L = 2, 64, 2, 131072
    st, dev, ds, tracks, _ = bench_cfg4.build("cuda:0", R, S, P, L, seed=99, contig=48 << 20)
    assert all(ds._tile_complete)
    full_regions = ds.full_regions.cpu().numpy() if hasattr(ds.full_regions, "cpu") else np.asarray(ds.full_regions)
    go, gv = dev.geno_offsets.cpu().numpy(), dev.geno_v_idxs.cpu().numpy()
    batches = list(ds.to_dataloader(batch_size=R * S, shuffle=True, seed=3, in_flight=1, group=1))
    assert len(batches) == 1
    batch = batches[0]
    torch.cuda.synchronize()
    idx = batch.idx.cpu().numpy()
    assert sorted(idx.tolist()) == list(range(R * S))
    r_idx, s_idx = np.unravel_index(idx, (R, S))
    regions = np.ascontiguousarray(full_regions[r_idx])
    goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
    to_rc = np.repeat(regions[:, 3] == -1, P)
    shifts = np.zeros_like(goi, dtype=np.int32)
    exp_h, _, exp_oh = oracle.reconstruct_haplotypes_fused(
        regions, shifts, goi, go, gv, st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets,
        st.pad_char, L, None, None, to_rc, True, onehot=True, n_threads=8)
    np.testing.assert_array_equal(batch.haps.cpu().numpy().ravel(), exp_h)
    np.testing.assert_array_equal(batch.onehot.cpu().numpy().reshape(-1, 4), exp_oh)
    diffs = oracle.get_diffs_sparse(goi, gv, go, st.ilens, None, None, regions[:, 1], regions[:, 2], st.v_starts)
    tlen = (regions[:, 2] - regions[:, 1]).astype(np.int64) - np.minimum(diffs.min(axis=1), 0)
    track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
    out_offsets = np.arange(len(idx) * P + 1, dtype=np.int64) * L
    a, e, v, io = tracks["cov"]
    exp = np.zeros(len(idx) * P * L, np.float32)
    oracle.intervals_and_realign_track_fused(exp, out_offsets, regions, shifts, goi, gv, go, st.v_starts, st.ilens, idx.astype(np.int64),
                                             a, e, v, io, track_offsets, np.array([0.0]), 0, 0, None, None, to_rc)
    got = batch.tracks[:, 0].contiguous().cpu().numpy().ravel()
    np.testing.assert_array_equal(bits(got), bits(exp))
    # size-independent properties at full size: at most one hot channel per base, the one-hot is the bytes' one-hot
    assert int(batch.onehot.sum(dim=-1).max()) <= 1
    assert int(batch.onehot.sum()) == sum(int((batch.haps == c).sum()) for c in b"ACGT")


def test_painting_dense_nested_and_gappy_intervals(ffi, oracle, tpath):
    """Painting beyond the tiled kernel's comfort zone: thousands of 1-3 bp intervals per 2048-value
    chunk (more candidates than one LDS tile -> the per-value kernel takes those chunks), long
    intervals with many short ones nested inside (the walk-back), wide gaps (the prefix-max
    shortcut), negative relative starts, rows of different lengths, an empty list, with and
    without precomputed prefix maxima."""
    import torch

    from genvarloader_amd import device

    rng = np.random.default_rng(77)
    lens = [5000, 2048, 1, 7001, 300]
    qstart = rng.integers(1000, 2000, len(lens)).astype(np.int32)
    lists = []
    for qi, (L, q0) in enumerate(zip(lens, qstart)):
        if qi == 4:
            lists.append((np.zeros(0, np.int32),) * 2 + (np.zeros(0, np.float32),))
            continue
        parts = []
        # dense: every position of [q0 - 50, q0 + 2500) starts a 1-3 bp interval (3 per position in a burst)
        dense = np.repeat(np.arange(q0 - 50, q0 + min(L, 2500)), 1 + (qi == 0) * 2)
        parts.append((dense, dense + rng.integers(1, 4, dense.size)))
        # long intervals holding nested short ones, then gaps
        base = q0 + 2600
        while base < q0 + L + 100:
            w = int(rng.integers(200, 900))
            parts.append((np.array([base]), np.array([base + w])))
            inner = np.sort(rng.integers(base, base + w // 2, 40))
            parts.append((inner, inner + rng.integers(1, 6, inner.size)))
            base += w + int(rng.integers(50, 700))
        s = np.concatenate([p[0] for p in parts]); e = np.concatenate([p[1] for p in parts])
        order = np.argsort(s, kind="stable")
        lists.append((s[order].astype(np.int32), e[order].astype(np.int32), rng.random(s.size).astype(np.float32) + 0.5))
    its = np.concatenate([l[0] for l in lists]); ite = np.concatenate([l[1] for l in lists])
    itv = np.concatenate([l[2] for l in lists])
    ito = np.concatenate([[0], np.cumsum([len(l[0]) for l in lists])]).astype(np.int64)
    out_offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    # query order differs from list order and one list is used twice
    oi = np.array([3, 0, 2, 1, 4, 0], np.int64)
    qs = np.array([qstart[3], qstart[0], qstart[2], qstart[1], qstart[4], qstart[0] + 17], np.int32)
    qlens = [lens[3], lens[0], lens[2], lens[1], lens[4], 4000]
    oo = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    exp = np.full(int(oo[-1]), 7.0, np.float32)
    oracle.intervals_to_tracks(oi, qs, its, ite, itv, ito, exp, oo)
    got = device.intervals_to_tracks(oi, qs, its, ite, itv, ito, oo).cpu().numpy()
    np.testing.assert_array_equal(bits(got), bits(exp))
    pm = device.intervals_prefix_max(ite, ito)
    ref_pm = np.concatenate([np.maximum.accumulate(l[1]) if len(l[1]) else l[1] for l in lists])
    np.testing.assert_array_equal(pm.cpu().numpy(), ref_pm)
    got2 = device.intervals_to_tracks(oi, qs, its, ite, itv, ito, oo, itv_pmax_ends=pm).cpu().numpy()
    np.testing.assert_array_equal(bits(got2), bits(exp))
    torch.cuda.synchronize()


def test_tracks_coordinates_beyond_2_30(ffi, oracle):
    """Same batch twice: as generated, and with every genomic coordinate moved up by 2^30 + 12345
    (regions, variant positions, interval starts / ends).  Tracks are relative to the query
    start, so both must give the same output -- and the shifted one must give it through the
    i64 scalar walk (the scan planner only takes |coordinates| < 2^30)."""
    st, bt, itv = _track_batch(91, 12, 1500, 120_000, shifts=True)
    B, P = bt.geno_offset_idx.shape
    L = bt.output_length
    tlen = (bt.regions[:, 2] - bt.regions[:, 1]) + 40
    track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
    out_offsets = np.arange(B * P + 1, dtype=np.int64) * L
    outs = []
    for off in (0, (1 << 30) + 12345):
        regions = bt.regions.copy(); regions[:, 1:3] += off
        args = (out_offsets, regions, bt.shifts, bt.geno_offset_idx, bt.geno_v_idxs, bt.geno_offsets,
                (st.v_starts.astype(np.int64) + off).astype(np.int32), st.ilens, itv["offset_idxs"],
                (itv["itv_starts"].astype(np.int64) + off).astype(np.int32),
                (itv["itv_ends"].astype(np.int64) + off).astype(np.int32), itv["itv_values"], itv["itv_offsets"],
                track_offsets, np.array([2.0]), 4, 7, None, None, bt.to_rc)
        exp = np.zeros(B * P * L, np.float32)
        oracle.intervals_and_realign_track_fused(exp, *args)
        got = np.full(B * P * L, 9.0, np.float32)
        ffi.intervals_and_realign_track_fused(got, *args)
        np.testing.assert_array_equal(bits(got), bits(exp))
        outs.append(got)
    np.testing.assert_array_equal(bits(outs[0]), bits(outs[1]))


@pytest.mark.parametrize("s_id", [0, 1, 2, 3, 4])
def test_reference_numpy_fallback_tracks(ffi, s_id, tpath):
    """The GPU realignment against vectors from the reference's own numpy fallback."""
    from tests.test_oracle_tracks import _pyref_tracks, _run_pyref_tracks

    d = _pyref_tracks()
    for use_keep in (0, 1):
        got = _run_pyref_tracks(ffi.shift_and_realign_tracks_sparse, d, s_id, use_keep)
        np.testing.assert_array_equal(bits(got), bits(d[f"expected_s{s_id}_k{use_keep}"]))
