"""Spliced haplotypes: the splice plan (reference KATs + vectors from the reference's own
``build_splice_plan``), its torch restatement, and -- on the GPU -- ``DeviceSplicedHapsDataset`` against the
oracle's plan + ploidy-1 reconstruction over the permuted elements."""
import numpy as np
import pytest
import torch

import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from test_loader import _grid_dataset  # noqa: E402

GOLD = Path(__file__).resolve().parent / "golden"

# /root/reference/tests/unit/splice/test_splice_plan.py: (lengths, splice_row_offsets, n_samples, n_rows,
# permutation, permuted_lengths, permuted_out_offsets, group_offsets)
PLAN_KATS = [
    ([3, 4, 5], [0, 2, 3], 1, 2, [0, 1, 2], [3, 4, 5], [0, 3, 7, 12], [0, 7, 12]),
    ([[10, 11], [20, 21], [30, 31]], [0, 2, 3], 1, 2, [0, 2, 1, 3, 4, 5], [10, 20, 11, 21, 30, 31],
     [0, 10, 30, 41, 62, 92, 123], [0, 30, 62, 92, 123]),
    ([[1, 2], [3, 4], [5, 6], [7, 8]], [0, 2, 4], 2, 1, [0, 2, 1, 3, 4, 6, 5, 7], [1, 3, 2, 4, 5, 7, 6, 8],
     None, [0, 4, 10, 22, 36]),
    ([[5, 6], [7, 8]], [0, 1, 2], 1, 2, [0, 1, 2, 3], [5, 6, 7, 8], None, None),
]


def _check_plan(plan, perm, plen, oo, go):
    np.testing.assert_array_equal(plan["permutation"], perm)
    np.testing.assert_array_equal(plan["permuted_lengths"], plen)
    if oo is not None:
        np.testing.assert_array_equal(plan["permuted_out_offsets"], oo)
    if go is not None:
        np.testing.assert_array_equal(plan["group_offsets"], go)


def test_oracle_splice_plan_reference_kats(oracle):
    for lengths, off, ns, nr, perm, plen, oo, go in PLAN_KATS:
        plan = oracle.build_splice_plan(np.array(lengths, np.int32), np.array(off, np.int64), ns, nr)
        _check_plan(plan, perm, plen, oo, go)
    # test_plan_total_bytes_consistent
    rng = np.random.default_rng(0)
    lengths = rng.integers(1, 20, size=(6, 3), dtype=np.int32)
    plan = oracle.build_splice_plan(lengths, np.array([0, 2, 4, 6], np.int64), 1, 3)
    assert int(plan["permuted_out_offsets"][-1]) == int(plan["group_offsets"][-1]) == int(lengths.sum())
    assert plan["out_shape"] == (3, 1, 3, None)


def _fixture_cases():
    z = np.load(GOLD / "pyref_splice_plan.npz")
    for i in range(int(z["n"])):
        yield {k: z[f"{i}/{k}"] for k in ("lengths", "offsets", "n_samples", "n_rows", "permutation", "permuted_lengths",
                                          "permuted_out_offsets", "group_offsets")}


def test_oracle_splice_plan_matches_reference_vectors(oracle):
    n = 0
    for c in _fixture_cases():
        plan = oracle.build_splice_plan(c["lengths"], c["offsets"], int(c["n_samples"]), int(c["n_rows"]))
        _check_plan(plan, c["permutation"], c["permuted_lengths"], c["permuted_out_offsets"], c["group_offsets"])
        n += 1
    assert n >= 6


def test_torch_splice_plan_matches_reference_vectors():
    """splice_plan_device is plain torch: on CPU tensors it must reproduce the reference's plans."""
    from genvarloader_amd.loader import splice_plan_device

    cases = list(_fixture_cases())
    for lengths, off, ns, nr, perm, plen, oo, go in PLAN_KATS:
        cases.append(dict(lengths=np.array(lengths, np.int32), offsets=np.array(off, np.int64), permutation=np.array(perm),
                          permuted_out_offsets=None if oo is None else np.array(oo), group_offsets=None if go is None else np.array(go)))
    for c in cases:
        lengths = c["lengths"] if c["lengths"].ndim == 2 else c["lengths"][:, None]
        p, oo, go = splice_plan_device(torch.as_tensor(lengths), torch.as_tensor(np.diff(c["offsets"])))
        np.testing.assert_array_equal(p.numpy(), c["permutation"])
        if c["permuted_out_offsets"] is not None:
            np.testing.assert_array_equal(oo.numpy(), c["permuted_out_offsets"])
        if c["group_offsets"] is not None:
            np.testing.assert_array_equal(go.numpy(), c["group_offsets"])


@pytest.mark.gpu
@pytest.mark.parametrize("exonic", [False, True], ids=["all-variants", "exonic-keep-mask"])
@pytest.mark.parametrize("index_on", ["host", "device"])
def test_spliced_dataset_matches_oracle(oracle, exonic, index_on):
    """(splice row, sample) pairs -> one spliced haplotype per ploid: against the oracle's plan over the
    oracle's per-element lengths and its ploidy-1 reconstruction at the plan's offsets
    (_query.py:207-313, _haps.py:876-931, _haps.py:1058-1112)."""
    from genvarloader_amd import HapsDevice
    from genvarloader_amd.loader import DeviceSplicedHapsDataset

    R, S, P, L = 9, 4, 2, 260
    st, full_regions, go, gv = _grid_dataset(63, R, S, P, L, indel_frac=0.5)
    rng = np.random.default_rng(8)
    row_len = np.array([3, 1, 0, 4, 2])                       # (an empty row too)
    so = np.concatenate([[0], np.cumsum(row_len)]).astype(np.int64)
    sr = rng.integers(0, R, int(so[-1])).astype(np.int64)
    n_rows = len(row_len)
    dev = HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
                     alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=go, geno_v_idxs=gv,
                     pad_char=st.pad_char)
    ds = DeviceSplicedHapsDataset(dev, full_regions, S, P, splice_offsets=so, splice_region_idx=sr, onehot=True, haps=True,
                                  annotate=True, exonic=exonic)
    assert len(ds) == n_rows * S
    if index_on == "device":
        ds.host_index_max = -1              # (large batches' way: the index arithmetic as torch device ops)
    seen = 0
    for batch in ds.to_dataloader(batch_size=6, shuffle=True, seed=2):
        pairs = batch.pairs.cpu().numpy()
        rows, smp = pairs // S, pairs % S
        pair_len = row_len[rows]
        off = np.concatenate([[0], np.cumsum(pair_len)]).astype(np.int64)
        r_idx = np.concatenate([sr[so[r]:so[r + 1]] for r in rows]) if len(rows) else np.zeros(0, np.int64)
        s_idx = np.repeat(smp, pair_len)
        np.testing.assert_array_equal(batch.idx.cpu().numpy(), r_idx * S + s_idx)
        B = len(r_idx)
        regions = full_regions[r_idx]
        goi = np.ravel_multi_index((r_idx[:, None], s_idx[:, None], np.arange(P)), (R, S, P))
        keep = ko = None
        if exonic:
            keep, ko = oracle.choose_exonic_variants(regions[:, 1], regions[:, 2], goi, gv, go, st.v_starts, st.ilens)
        diffs = oracle.get_diffs_sparse(goi, gv, go, st.ilens, keep, ko, regions[:, 1], regions[:, 2], st.v_starts)
        lengths = ((regions[:, 2] - regions[:, 1])[:, None] + diffs).astype(np.int32)        # _haps.py:568
        # (this path has one pair layout: pairs are the batch's own (row, sample) grid flattened)
        plan = oracle.build_splice_plan(lengths, off, 1, len(pairs))
        perm = plan["permutation"]
        np.testing.assert_array_equal(batch.permutation.cpu().numpy(), perm)
        np.testing.assert_array_equal(batch.out_offsets.cpu().numpy(), plan["permuted_out_offsets"])
        np.testing.assert_array_equal(batch.group_offsets.cpu().numpy(), plan["group_offsets"])
        q = perm // P
        kp = kop = None
        if exonic:           # the keep mask in permuted order (_haps.py:1086-1101)
            klen = np.diff(ko)[perm]
            kop = np.concatenate([[0], np.cumsum(klen)]).astype(np.int64)
            kp = np.concatenate([keep[ko[k]:ko[k + 1]] for k in perm]) if len(perm) else np.zeros(0, bool)
        total = int(plan["permuted_out_offsets"][-1])
        exp = np.zeros(total, np.uint8)
        av, ap = np.zeros(total, np.int32), np.zeros(total, np.int32)
        oh = np.zeros((total, 4), np.uint8)
        to_rc = (regions[:, 3] == -1)[q]
        oracle.reconstruct_haplotypes_from_sparse(
            exp, plan["permuted_out_offsets"], regions[q], np.zeros((B * P, 1), np.int32), goi.reshape(-1)[perm].reshape(-1, 1),
            go, gv, st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, kp, kop, av, ap,
            to_rc=to_rc, onehot_out=oh)
        np.testing.assert_array_equal(batch.haps.cpu().numpy(), exp)
        np.testing.assert_array_equal(batch.onehot.cpu().numpy(), oh)
        np.testing.assert_array_equal(batch.annot_v_idxs.cpu().numpy(), av)
        np.testing.assert_array_equal(batch.annot_ref_pos.cpu().numpy(), ap)
        seen += len(pairs)
    assert seen == n_rows * S
