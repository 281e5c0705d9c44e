"""Known-answer tests transcribed (values only) from the reference's own unit tests:

* Rust ``#[cfg(test)]`` in src/reconstruct/mod.rs:906-1538, src/reverse.rs:91-178,
  src/genotypes/mod.rs:183-231, src/reference/mod.rs:134-265
* Python tests/unit/dataset/genotypes/test_reconstruct.py:11-254 (with annotation
  expectations), test_get_diffs.py:5-138, test_reconstruct_trailing_fill.py:13-31

The same table is replayed through the HIP path in tests/test_gpu_parity.py.
"""

import numpy as np
import pytest

MAX = 2147483647
A = lambda *x: np.array(x, np.uint8)  # noqa: E731
S = lambda s: np.frombuffer(s.encode(), np.uint8).copy()  # noqa: E731

# (name, v_idxs, v_starts, ilens, shift, alt, alt_off, ref, ref_start, L, pad, keep,
#  exp_out, exp_av, exp_ap)
ROW_KATS = [
    ("no_variants", [], [], [], 0, A(), [0], A(10, 20, 30, 40, 50), 1, 3, 0, None,
     A(20, 30, 40), None, None),
    ("negative_start", [], [], [], 0, A(), [0], A(1, 2, 3, 4, 5), -2, 5, 9, None,
     A(9, 9, 1, 2, 3), [-1] * 5, [-1, -1, 0, 1, 2]),
    ("snp", [0], [2], [0], 0, A(84), [0, 1], A(65, 67, 71, 84, 65), 0, 5, 0, None,
     A(65, 67, 84, 84, 65), [-1, -1, 0, -1, -1], None),
    ("ins2", [0], [2], [2], 0, A(10, 11, 12), [0, 3], A(1, 2, 3, 4, 5), 0, 5, 0, None,
     A(1, 2, 10, 11, 12), None, None),
    ("del2", [0], [2], [-2], 0, A(30), [0, 1], A(1, 2, 3, 4, 5, 6, 7), 0, 5, 0, None,
     A(1, 2, 30, 6, 7), None, None),
    ("del_spans_start", [0], [1], [-3], 0, A(99), [0, 1], A(1, 2, 3, 4, 5, 6, 7), 3, 5, 0, None,
     A(6, 7, 0, 0, 0), None, [5, 6, MAX, MAX, MAX]),
    ("overshoot_contig", [0], [2], [-5], 0, A(50), [0, 1], A(1, 2, 3, 4), 0, 8, 0, None,
     A(1, 2, 50, 0, 0, 0, 0, 0), None, None),
    ("overlapping_alts", [0, 1], [2, 2], [0, 0], 0, A(20, 30), [0, 1, 2], A(1, 2, 3, 4, 5), 0, 5, 0,
     None, A(1, 2, 20, 4, 5), None, None),
    ("shift_via_ref", [0], [3], [1], 2, A(99, 88), [0, 2], A(1, 2, 3, 4, 5, 6), 0, 5, 0, None,
     A(3, 99, 88, 5, 6), None, None),
    ("shift_into_allele", [0], [3], [1], 4, A(99, 88), [0, 2], A(1, 2, 3, 4, 5, 6, 7, 8), 0, 4, 0,
     None, A(88, 5, 6, 7), None, None),
    ("right_pad", [], [], [], 0, A(), [0], A(1, 2, 3), 0, 6, 0, None,
     A(1, 2, 3, 0, 0, 0), [-1] * 6, [0, 1, 2, MAX, MAX, MAX]),
    ("skip_eq_len", [0], [3], [0], 4, A(88), [0, 1], A(1, 2, 3, 4, 5, 6, 7, 8), 0, 4, 0, None,
     A(5, 6, 7, 8), None, None),
    ("not_enough_distance", [0], [3], [0], 10, A(77), [0, 1], np.arange(1, 16, dtype=np.uint8), 0, 3,
     0, None, A(11, 12, 13), None, None),
    ("keep_mask", [0, 1], [1, 3], [0, 0], 0, A(55, 99), [0, 1, 2], A(1, 2, 3, 4, 5), 0, 5, 0,
     [False, True], A(1, 2, 3, 99, 5), [-1, -1, -1, 1, -1], None),
    # python unit tests, ref = "ACGG"
    ("py_snps", [1], [1, 3], [0, 0], 0, S("AT"), [0, 1, 2], S("ACGG"), 1, 3, ord("N"), None,
     S("CGT"), [-1, -1, 1], [1, 2, 3]),
    ("py_indels", [0, 1], [1, 3], [-1, 1], 0, S("GAT"), [0, 1, 3], S("ACGG"), 0, 4, ord("N"), None,
     S("AGAT"), [-1, 0, 1, 1], [0, 1, 3, 3]),
    ("py_spanning_del_pad", [0], [0], [-1], 0, S("G"), [0, 1], S("ACGG"), 1, 3, ord("N"), None,
     S("GGN"), [-1, -1, -1], [2, 3, MAX]),
    ("py_shift_ins", [0, 1], [1, 3], [1, 1], 1, S("TCGA"), [0, 2, 4], S("ACGG"), 0, 4, ord("N"), None,
     S("TCGG"), [0, 0, -1, 1], [1, 1, 2, 3]),
    ("py_del_past_end", [0], [2], [-2], 0, S("G"), [0, 1], S("ACGTA"), 0, 5, ord("N"), None,
     S("ACGNN"), [-1, -1, 0, -1, -1], [0, 1, 2, MAX, MAX]),
    ("py_overlapping", [0, 1], [1, 1], [0, 0], 0, S("TG"), [0, 1, 2], S("ACGG"), 0, 4, ord("N"), None,
     S("ATGG"), [-1, 0, -1, -1], [0, 1, 2, 3]),
]


@pytest.mark.parametrize("kat", ROW_KATS, ids=[k[0] for k in ROW_KATS])
def test_row_kats(oracle, kat):
    (_, v_idxs, v_starts, ilens, shift, alt, alt_off, ref, ref_start, L, pad, keep,
     exp, exp_av, exp_ap) = kat
    out = np.full(L, 0xFF, np.uint8)
    av = np.full(L, -7, np.int32)
    ap = np.full(L, -7, np.int32)
    oracle.reconstruct_haplotype_from_sparse(
        np.array(v_idxs, np.int32), np.array(v_starts, np.int32), np.array(ilens, np.int32),
        shift, alt, np.array(alt_off, np.int64), ref, ref_start, out, pad,
        None if keep is None else np.array(keep, bool), av, ap)
    np.testing.assert_array_equal(out, exp)
    if exp_av is not None:
        np.testing.assert_array_equal(av, exp_av)
    if exp_ap is not None:
        np.testing.assert_array_equal(ap, exp_ap)
    # annot == plain (reconstruct/mod.rs "annot≡plain"): same bytes without buffers
    out2 = np.full(L, 0xFF, np.uint8)
    oracle.reconstruct_haplotype_from_sparse(
        np.array(v_idxs, np.int32), np.array(v_starts, np.int32), np.array(ilens, np.int32),
        shift, alt, np.array(alt_off, np.int64), ref, ref_start, out2, pad,
        None if keep is None else np.array(keep, bool))
    np.testing.assert_array_equal(out2, exp)


def test_batch_kats(oracle):
    # reconstruct/mod.rs:1414 -- two queries, no variants
    ref = S("ACGTACGTACGT")
    out = np.zeros(8, np.uint8)
    oracle.reconstruct_haplotypes_from_sparse(
        out, [0, 4, 8], [[0, 0, 4], [0, 4, 8]], [[0], [0]], [[0], [1]], [[0, 0], [0, 0]],
        np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.uint8),
        [0], ref, [0, 12], 0)
    assert out.tobytes() == b"ACGTACGT"
    # reconstruct/mod.rs:1468 -- SNP at pos 1 -> "T" on query 0 only
    ref = S("ACGTACGT")
    out = np.zeros(8, np.uint8)
    oracle.reconstruct_haplotypes_from_sparse(
        out, [0, 4, 8], [[0, 0, 4], [0, 4, 8]], [[0], [0]], [[0], [1]], [[0, 1], [1, 1]],
        [0], [1], [0], S("T"), [0, 1], ref, [0, 8], 0)
    assert out.tobytes() == b"ATGTACGT"


def test_trailing_fill_sentinel(oracle):
    # tests/unit/dataset/test_reconstruct_trailing_fill.py:13-31: a 0xFF-filled
    # buffer must be fully overwritten when a DEL runs past the contig end.
    out = np.full(6, 0xFF, np.uint8)
    oracle.reconstruct_haplotype_from_sparse(
        [0], [2], [-5], 0, S("G"), [0, 1], S("ACGT"), 0, out, ord("N"))
    assert out.tobytes() == b"ACGNNN"


RC_KATS = [("ACGT", "ACGT"), ("ACN", "NGT"), ("ACGTA", "TACGT"), ("ACG", "CGT"), ("", "")]


def test_rc_kats(oracle):
    for s, e in RC_KATS:
        d = S(s)
        oracle.rc_flat_rows_inplace(d, [0, len(s)], [True])
        assert d.tobytes() == e.encode()
    # masked rows only (reverse.rs:105-114)
    d = S("ACGTAACG")
    oracle.rc_flat_rows_inplace(d, [0, 4, 8], [True, False])
    assert d.tobytes() == b"ACGTAACG"
    # all 256 bytes == bytes.maketrans (reverse.rs:154-162)
    table = bytes.maketrans(b"ACGT", b"TGCA")
    for b in range(256):
        d = np.array([b], np.uint8)
        oracle.rc_flat_rows_inplace(d, [0, 1], [True])
        assert d[0] == table[b]
    # empty row + all-false (reverse.rs:143-149)
    d = S("AC")
    oracle.rc_flat_rows_inplace(d, [0, 0, 2], [True, False])
    assert d.tobytes() == b"AC"
    # plain reversal of 4-byte elements (reverse.rs:125-141)
    f = np.array([1.0, 2.0, 3.0, 9.0], np.float32)
    oracle.reverse_flat_rows_inplace(f, [0, 3, 4], [True, False])
    np.testing.assert_array_equal(f, [3.0, 2.0, 1.0, 9.0])
    i = np.array([10, 11, 12], np.int32)
    oracle.reverse_flat_rows_inplace(i, [0, 3], [True])
    np.testing.assert_array_equal(i, [12, 11, 10])


def test_get_diffs_kats(oracle):
    g = lambda **kw: oracle.get_diffs_sparse(**kw)[0, 0]  # noqa: E731
    # genotypes/mod.rs:184-212
    assert g(geno_offset_idx=[[0]], geno_v_idxs=[0, 1], geno_offsets=[[0], [2]], ilens=[-2, 3]) == 1
    assert g(geno_offset_idx=[[0]], geno_v_idxs=[], geno_offsets=[[0], [0]], ilens=[]) == 0
    # test_get_diffs.py
    assert g(geno_offset_idx=[[0]], geno_v_idxs=[0, 1, 2], geno_offsets=[0, 3], ilens=[1, -2, 3]) == 2
    assert g(geno_offset_idx=[[0]], geno_v_idxs=[0, 1, 2], geno_offsets=[0, 3], ilens=[1, -2, 3],
             keep=[True, False, True], keep_offsets=[0, 3]) == 4
    assert g(geno_offset_idx=[[0]], geno_v_idxs=[0], geno_offsets=[0, 1], ilens=[-3],
             q_starts=[2], q_ends=[10], v_starts=[0]) == -2
    assert g(geno_offset_idx=[[0]], geno_v_idxs=[0, 1], geno_offsets=[0, 2], ilens=[2, 5],
             q_starts=[0], q_ends=[10], v_starts=[0, 20]) == 2


def test_choose_exonic_kat(oracle):
    # genotypes/mod.rs:215-231
    keep, ko = oracle.choose_exonic_variants([10], [20], [[0]], [0, 1, 2], [[0], [3]],
                                             [12, 19, 19], [0, 0, -2])
    assert keep.tolist() == [True, True, False] and ko.tolist() == [0, 3]


def test_padded_slice_kats(oracle):
    # reference/mod.rs:135-265 via get_reference with one contig
    def ps(arr, start, stop, pad, n=None, rc=None):
        n = max(stop - start, 0) if n is None else n
        return oracle.get_reference([[0, start, stop]], [0, n], np.array(arr, np.uint8),
                                    [0, len(arr)], pad, False, rc).tolist()

    assert ps([1, 2, 3, 4, 5], 1, 4, 0) == [2, 3, 4]
    assert ps([1, 2, 3], -2, 2, 9) == [9, 9, 1, 2]
    assert ps([1, 2, 3], 1, 5, 9) == [2, 3, 9, 9]
    assert ps([1, 2], -1, 3, 9) == [9, 1, 2, 9]
    assert ps([1, 2, 3], 2, 2, 9) == []
    assert ps([1, 2, 3], -5, -1, 7, n=3) == [7, 7, 7]
    assert bytes(ps(list(b"ACGTAA"), 0, 3, ord("N"), rc=[True])) == b"CGT"
