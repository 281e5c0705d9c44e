"""Pins the track half of the oracle against the reference's frozen
goldens: tests/parity/test_shift_and_realign_tracks_parity.py, test_intervals_to_tracks_parity.py,
test_prng_parity.py (known answers :58-71 + goldens)."""

import numpy as np
import pytest

from tests._fixtures import load_ref_cases


def test_prng_kats_and_goldens(oracle):
    # tests/parity/test_prng_parity.py:58-71
    for x, e in [(1, 1082269761), (2, 2164539522), (42, 45454805674), (0xDEADBEEF, 4018790486776397394),
                 (2**64 - 1, 1065361344)]:
        assert oracle.xorshift64(x) == e
    assert oracle.hash4(1, 2, 3, 4) == 11323120931611735037
    assert oracle.hash4(0, 0, 0, 0) == 0
    assert oracle.hash4(0xDEADBEEF, 0xCAFE, 0xBABE, 1) == 5244362157944750963
    xs = load_ref_cases("prng_xorshift64")
    assert len(xs) == 119
    for inp, exp in xs:
        assert oracle.xorshift64(int(inp[0])) == int(exp)
    hs = load_ref_cases("prng_hash4")
    assert len(hs) == 61
    for inp, exp in hs:
        assert oracle.hash4(*(int(v) for v in inp)) == int(exp)


def test_shift_and_realign_tracks_sparse_golden(oracle):
    cases = load_ref_cases("shift_and_realign_tracks_sparse")
    assert len(cases) == 200
    strategies = {}
    for ci, (inp, exp) in enumerate(cases):
        out = np.zeros(int(inp[0][-1]), np.float32)
        oracle.shift_and_realign_tracks_sparse(out, *inp)
        # bit-level equality (NaN-safe): the reference compares uint32 views
        np.testing.assert_array_equal(out.view(np.uint32), np.asarray(exp, np.float32).view(np.uint32),
                                      err_msg=f"case {ci}")
        strategies[int(inp[13])] = strategies.get(int(inp[13]), 0) + 1
    assert strategies == {0: 65, 1: 19, 2: 25, 3: 40, 4: 51}  # SURVEY 8(c) census


def test_intervals_to_tracks_golden(oracle):
    cases = load_ref_cases("intervals_to_tracks")
    assert len(cases) == 200
    for ci, (inp, exp) in enumerate(cases):
        out = np.full(int(inp[-1][-1]), 7.0, np.float32)  # must be zeroed by the kernel
        oracle.intervals_to_tracks(*inp[:6], out, inp[6])
        np.testing.assert_array_equal(out.view(np.uint32), np.asarray(exp, np.float32).view(np.uint32),
                                      err_msg=f"case {ci}")


def _pyref_tracks():
    from tests._fixtures import GOLDEN

    z = np.load(GOLDEN / "pyref_tracks.npz")
    return {k: z[k] for k in z.files}


def _run_pyref_tracks(fn, d, s_id, use_keep):
    B, P = d["geno_offset_idx"].shape
    L = int(d["output_length"])
    out_offsets = np.arange(B * P + 1, dtype=np.int64) * L
    out = np.full(B * P * L, 7.0, np.float32)
    fn(out, out_offsets, d["regions"], d["shifts"], d["geno_offset_idx"], d["geno_v_idxs"], d["geno_offsets"],
       d["v_starts"], d["ilens"], d["tracks"], d["track_offsets"], np.array([float(d[f"param_s{s_id}"])]),
       d["keep"] if use_keep else None, d["keep_offsets"] if use_keep else None, s_id, int(d["base_seed"]))
    return out


@pytest.mark.parametrize("s_id", [0, 1, 2, 3, 4])
def test_oracle_vs_reference_numpy_fallback_tracks(oracle, s_id):
    """pyref_tracks.npz: the reference's pure-numpy track fallback (_dataset/_tracks.py:621-824)
    on a 20-row x 900 batch with shifts, 11 variants per row, with and without keep masks."""
    d = _pyref_tracks()
    for use_keep in (0, 1):
        got = _run_pyref_tracks(oracle.shift_and_realign_tracks_sparse, d, s_id, use_keep)
        np.testing.assert_array_equal(got.view(np.uint32), d[f"expected_s{s_id}_k{use_keep}"].view(np.uint32))
