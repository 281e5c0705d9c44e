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


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def test_golden_shift_and_realign_tracks_sparse(ffi):
    cases = load_ref_cases("shift_and_realign_tracks_sparse")
    assert len(cases) == 200
    for ci, (inp, exp) in enumerate(cases):
        out = np.full(int(inp[0][-1]), 7.0, np.float32)
        ffi.shift_and_realign_tracks_sparse(out, *inp)
        np.testing.assert_array_equal(bits(out), bits(exp), err_msg=f"case {ci} strategy {int(inp[13])}")


def test_golden_intervals_to_tracks(ffi):
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
def test_fused_tracks_synthetic(ffi, oracle, strategy, param):
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
