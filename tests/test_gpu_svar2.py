"""The SVAR2 two-source provider through the HIP path (C-ABI: gvl_svar2_merge + the library's kernels over
the merged table): the reference's Rust known-answer tests (tests/svar2_kats.py), synthetic batches against the oracle,
the SVAR1 == SVAR2 equivalence, every kernel path, the scatter write, and the device-side error reports.  Bit-exact.
Pinned by KATs only (no 200-case golden exists for these entry points); the codec's key bits are the integrator's."""

import numpy as np
import pytest

from tests import svar2_kats as K
from tests.test_gpu_parity import KERNEL_PATHS, gpu, kpath, make_dev  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sv2(gpu):
    import genvarloader_amd.svar2 as svar2

    return svar2


def _recon_args(oracle, k):
    ch = K.kat_channels(oracle, k)
    return (np.asarray(k["regions"], np.int32), np.asarray(k["shifts"], np.int32), ch["vk_pos"], ch["vk_ilen"], ch["vk_alt_off"],
            ch["vk_off"], ch["dense_pos"], ch["dense_ilen"], ch["dense_alt_off"], ch["dense_range"], ch["dense_present"],
            ch["dense_present_off"], ch["alt_bytes"], K.S(k["ref"]), np.asarray([0, len(k["ref"])], np.int64), ord("N"))


@pytest.mark.parametrize("case", K.RECON_KATS, ids=[c[0] for c in K.RECON_KATS])
def test_reconstruct_kats(gpu, sv2, oracle, kpath, case):
    # src/reconstruct/mod.rs:1540-1813, down every kernel path (the scatter write: the all-purpose kernel's row setup)
    _, k, init, exp = case
    out = K.S(init)
    sv2.reconstruct_haplotypes_from_svar2_into(out, k["out_bounds"], *_recon_args(oracle, k))
    assert out.tobytes() == exp


@pytest.mark.parametrize("case", K.RECON_KATS[:2], ids=[c[0] for c in K.RECON_KATS[:2]])
def test_reconstruct_kats_fused_entry(gpu, sv2, oracle, kpath, case):
    # the same vectors through the fused entry (src/ffi/mod.rs:874-997): gap-free offsets, fixed length
    _, k, init, exp = case
    got, off = sv2.reconstruct_haplotypes_from_svar2(*_recon_args(oracle, k), len(exp))
    assert got.tobytes() == exp and off.tolist() == [0, len(exp)]
    oh, off, haps = sv2.reconstruct_haplotypes_from_svar2(*_recon_args(oracle, k), len(exp), onehot=True)
    np.testing.assert_array_equal(oh, oracle.onehot(np.frombuffer(exp, np.uint8)))


def test_hap_diffs_kat(gpu, sv2, oracle):
    # src/svar2/mod.rs:654-700
    k = K.DIFFS_KAT
    ch = K.kat_channels(oracle, k)
    for fe in (False, True):
        d = sv2.hap_diffs_svar2(np.asarray(k["regions"], np.int32), k["ploidy"], ch["vk_pos"], ch["vk_ilen"], ch["vk_off"],
                                ch["dense_pos"], ch["dense_ilen"], ch["dense_range"], ch["dense_present"], ch["dense_present_off"], fe)
        assert d.tolist() == k["expected"]


def test_track_kat(gpu, sv2, oracle):
    # src/tracks/mod.rs:2509-2566 through the fused entry (src/ffi/mod.rs:1835-1966: the row is sized region length + diff)
    k = K.TRACK_KAT
    ch = K.kat_channels(oracle, k)
    args = (np.asarray(k["regions"], np.int32), np.asarray(k["shifts"], np.int32), ch["vk_pos"], ch["vk_ilen"], ch["vk_off"],
            ch["dense_pos"], ch["dense_ilen"], ch["dense_range"], ch["dense_present"], ch["dense_present_off"],
            np.asarray(k["track"], np.float32), np.asarray(k["track_offsets"], np.int64), k["params"], k["strategy_id"], k["base_seed"])
    got, off = sv2.shift_and_realign_tracks_from_svar2(*args)
    exp, eoff = oracle.shift_and_realign_tracks_from_svar2(*args)
    assert off.tolist() == eoff.tolist() == [0, 2]
    assert got.tolist() == exp.tolist() == k["expected"][:2]


def _merged_streams(m):
    """Per haplotype the merged (pos, ilen, alen, first byte) records of a Svar2Merged, read back from its workspace."""
    import ctypes as C

    import torch

    n = int(m.geno_offset_idx.numel())
    base = m.workspace.data_ptr()

    def view(ptr, nbytes, dt):
        off = int(ptr) - base
        return m.workspace[off:off + nbytes].view(dt).cpu().numpy()

    gs = view(m.c.geno_o_starts, 8 * n, torch.int64)
    ge = view(m.c.geno_o_stops, 8 * n, torch.int64)
    cap = int(m.c.n_geno)
    rec = view(m.c.geno_rec, 16 * cap, torch.int32).reshape(-1, 4) if cap else np.zeros((0, 4), np.int32)
    srec = view(m.c.slot_rec, 128 * n, torch.int32).reshape(n, 8, 4)
    assert m.geno_offset_idx.cpu().numpy().reshape(-1).tolist() == list(range(n))
    _ = C
    return gs, ge, rec, srec


def _check_merge(gpu, sv2, oracle, regions, P, sv, ref, ref_offsets, filter_exonic=False):
    from genvarloader_amd import ffi

    dev = ffi._ref_static(ref, ref_offsets, ord("N"))
    ch = sv2.Svar2Channels(*sv.args() if hasattr(sv, "args") else sv, filter_exonic=filter_exonic)
    m = sv2.merge(dev, ch, regions, P)
    gpu.torch.cuda.synchronize()
    from genvarloader_amd import _lib

    _lib.check_async()
    gs, ge, rec, srec = _merged_streams(m)
    a = sv.args() if hasattr(sv, "args") else sv
    vk_pos, vk_ilen, vk_alt_off, vk_off, d_pos, d_ilen, d_alt_off, d_range, d_present, d_poff, alt = [np.asarray(x) for x in a]
    d_range = d_range.reshape(-1, 2)
    for k in range(len(gs)):
        q = k // P
        pos, src = oracle.merge_hap(vk_pos, vk_off[k], vk_off[k + 1], d_pos, d_range[q, 0], d_range[q, 1], d_present, d_poff[k])
        il = np.array([vk_ilen[s] if s >= 0 else d_ilen[-(s + 1)] for s in src], np.int64)
        a0 = np.array([vk_alt_off[s] if s >= 0 else d_alt_off[-(s + 1)] for s in src], np.int64)
        a1 = np.array([vk_alt_off[s + 1] if s >= 0 else d_alt_off[-(s + 1) + 1] for s in src], np.int64)
        if filter_exonic:
            end = pos.astype(np.int64) - np.minimum(il, 0) + 1
            keep = (pos.astype(np.int64) >= regions[q, 1]) & (end <= regions[q, 2])
            pos, il, a0, a1 = pos[keep], il[keep], a0[keep], a1[keep]
        got = rec[gs[k]:ge[k]]
        assert got[:, 0].tolist() == pos.astype(np.int64).tolist(), f"hap {k}: positions"
        assert got[:, 1].tolist() == il.tolist(), f"hap {k}: ilens"
        alen = np.where(a1 - a0 == 0, 1, a1 - a0)
        assert ((got[:, 2].view(np.uint32)) >> 8).tolist() == alen.tolist(), f"hap {k}: allele lengths"
        c_s = int(ref_offsets[regions[q, 0]])
        first = [int(alt[a0[i]]) if a1[i] > a0[i] else int(ref[c_s + pos[i]]) for i in range(len(pos))]
        assert (got[:, 2].view(np.uint32) & 0xFF).tolist() == first, f"hap {k}: first allele bytes / anchors"
        # the slot line: the same records, or OVERFLOW for more than 8
        if len(pos) > 8:
            assert int(srec[k, 0, 2].view(np.uint32)) == 0xFFFFFFFE
        else:
            assert srec[k, :len(pos), :3].tolist() == got[:, :3].tolist()
            assert all(int(v) == 0xFFFFFFFF for v in srec[k, len(pos):, 2].view(np.uint32))
    return m


@pytest.mark.parametrize("seed,filter_exonic", [(0, False), (1, True), (2, False), (3, True)])
def test_merge_fuzz_small_unsorted_with_ties(gpu, sv2, oracle, seed, filter_exonic):
    """Haplotypes of <= 64 entries per channel are ranked by counting: exactly merge_hap's stable sort (src/svar2/mod.rs:70)
    for ANY order inside the channels -- unsorted runs, ties inside and across channels, absent bits, bit offsets that
    straddle bytes, pure deletions, insertions."""
    rng = np.random.default_rng(100 + seed)
    B, P = 37, 2
    ref = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 5000)]
    ref_offsets = np.array([0, 2000, 5000], np.int64)
    contig = rng.integers(0, 2, B)
    start = rng.integers(0, 1500, B)
    regions = np.stack([contig, start, start + rng.integers(1, 400, B)], 1).astype(np.int32)
    vk_pos, vk_cnt, d_pos, d_rng, bits = [], [], [], [], []
    for q in range(B):
        nd = int(rng.integers(0, 65 if q % 5 == 0 else 12))
        d_rng.append((sum(len(x) for x in d_pos), sum(len(x) for x in d_pos) + nd))
        d_pos.append(rng.integers(regions[q, 1] - 20, regions[q, 1] + 60, nd).clip(0, 1999))       # (many ties, unsorted)
        for _ in range(P):
            na = int(rng.integers(0, 65 if q % 7 == 0 else 10))
            vk_pos.append(rng.integers(regions[q, 1] - 20, regions[q, 1] + 60, na).clip(0, 1999))
            vk_cnt.append(na)
            bits.append(rng.random(nd) < 0.6)
    vk_pos = np.concatenate(vk_pos).astype(np.int32)
    d_pos = np.concatenate(d_pos).astype(np.int32)

    def decoded(n):
        il = np.where(rng.random(n) < 0.5, 0, rng.integers(-6, 5, n)).astype(np.int32)
        ln = np.where(il < 0, 0, il + 1)
        off = np.zeros(n + 1, np.int64)
        np.cumsum(ln, out=off[1:])
        return il, off

    vk_il, vk_ao = decoded(len(vk_pos))
    d_il, d_ao = decoded(len(d_pos))
    d_ao = d_ao + vk_ao[-1]
    alt = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(d_ao[-1]))]
    vk_off = np.concatenate([[0], np.cumsum(vk_cnt)]).astype(np.int64)
    lead = 5                                             # (the first haplotype's bits start inside a byte)
    allbits = np.concatenate([np.zeros(lead, bool)] + bits)
    poff = lead + np.concatenate([[0], np.cumsum([len(b) for b in bits])]).astype(np.int64)
    present = np.packbits(allbits, bitorder="little")
    sv = (vk_pos, vk_il, vk_ao, vk_off, d_pos, d_il, d_ao, np.asarray(d_rng, np.int32), present, poff, alt)
    m = _check_merge(gpu, sv2, oracle, regions, P, sv, ref, ref_offsets, filter_exonic)
    # ... and the bytes: HIP over the merged table == the oracle's provider over the same channels
    shifts = np.zeros((B, P), np.int32)
    exp, eoff = oracle.reconstruct_haplotypes_from_svar2(regions, shifts, *sv, ref, ref_offsets, ord("N"), -1, filter_exonic=filter_exonic)
    got, off = sv2.reconstruct_haplotypes_from_svar2(regions, shifts, *sv, ref, ref_offsets, ord("N"), -1, filter_exonic=filter_exonic)
    np.testing.assert_array_equal(off, eoff)
    np.testing.assert_array_equal(got, exp)
    del m


def _synth(seed, n_q, length, indel, dense_af, out_len=None, rc=0.5, contig=400_000, edge=0.05):
    from genvarloader_amd import synth

    rng = np.random.default_rng(seed)
    st = synth.make_static(rng, (contig,), indel_frac=indel)
    bt = synth.make_batch(rng, st, n_q, 2, length, rc_frac=rc, random_shifts=out_len is None, edge_frac=edge, output_length=out_len)
    return st, bt, synth.to_svar2(rng, st, bt, dense_af=dense_af)


@pytest.mark.parametrize("layout", ["lc", "cl"])
def test_synthetic_batch_vs_oracle_every_path(gpu, sv2, oracle, kpath, layout):
    """A cfg3-shaped SVAR2 batch (SNPs + indels, RC, one-hot next to the bytes) down every kernel path: the merged table feeds the
    slot lines, the inline CSR records, the vrec gather and the scalar walk alike."""
    st, bt, sv = _synth(21, 192, 2048, 0.15, 0.3)
    exp, eoff = oracle.reconstruct_haplotypes_from_svar2(bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, 2048)
    oracle.rc_flat_rows_inplace(exp, eoff, bt.to_rc)
    oh, off, got = sv2.reconstruct_haplotypes_from_svar2(bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, 2048,
                                                         to_rc=bt.to_rc, onehot=True, layout=layout)
    np.testing.assert_array_equal(off, eoff)
    np.testing.assert_array_equal(got, exp)
    eoh = oracle.onehot(exp)
    if layout == "cl":
        eoh = eoh.reshape(-1, 2048, 4).transpose(0, 2, 1)
    np.testing.assert_array_equal(oh, eoh)


def test_cfg3_shaped_batch_full_size(gpu, sv2, oracle):
    """4096 x 2048 (BASELINE config 3's shape) as a SVAR2 batch: HIP == oracle, and == the SVAR1 route on the same haplotypes."""
    st, bt, sv = _synth(22, 2048, 2048, 0.15, 0.3, contig=8 << 20)
    exp, eoff = oracle.reconstruct_haplotypes_from_svar2(bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, 2048)
    oracle.rc_flat_rows_inplace(exp, eoff, bt.to_rc)
    oh, off, got = sv2.reconstruct_haplotypes_from_svar2(bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, 2048,
                                                         to_rc=bt.to_rc, onehot=True)
    np.testing.assert_array_equal(got, exp)
    np.testing.assert_array_equal(oh, oracle.onehot(exp))
    dev = make_dev(gpu, st, bt)
    o1 = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, 2048, None, None, bt.to_rc, haps=True, onehot=True)
    np.testing.assert_array_equal(o1.haps.cpu().numpy(), got)
    np.testing.assert_array_equal(o1.onehot.cpu().numpy(), oh)


@pytest.mark.parametrize("seed,dense_af", [(31, 0.0), (32, 0.5), (33, 1.1)])
def test_ragged_and_diffs_equal_svar1(gpu, sv2, oracle, seed, dense_af):
    """Ragged output (output_length -1: sized by hap_diffs_svar2 on the device) and the length deltas: SVAR2 route == SVAR1 route
    == oracle, with everything in the dense channel (dense_af 0), a mix, and everything in var_key (dense_af > 1)."""
    st, bt, sv = _synth(seed, 300, 1024, 0.25, dense_af, out_len=-1, rc=0.0)
    exp, eoff = oracle.reconstruct_haplotypes_from_svar2(bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, -1)
    got, off = sv2.reconstruct_haplotypes_from_svar2(bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, -1)
    np.testing.assert_array_equal(off, eoff)
    np.testing.assert_array_equal(got, exp)
    g1, o1 = gpu.ffi.reconstruct_haplotypes_fused(bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs,
                                                  st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets,
                                                  st.pad_char, -1)
    np.testing.assert_array_equal(o1, off)
    np.testing.assert_array_equal(g1, got)
    d = sv2.hap_diffs_svar2(bt.regions, 2, sv.vk_pos, sv.vk_ilen, sv.vk_off, sv.dense_pos, sv.dense_ilen, sv.dense_range,
                            sv.dense_present, sv.dense_present_off)
    de = oracle.hap_diffs_svar2(bt.regions, 2, sv.vk_pos, sv.vk_ilen, sv.vk_off, sv.dense_pos, sv.dense_ilen, sv.dense_range,
                                sv.dense_present, sv.dense_present_off)
    np.testing.assert_array_equal(d, de)


def test_filter_exonic_equals_keep_mask_route(gpu, sv2, oracle):
    """filter_exonic during the merge == the SVAR1 route under choose_exonic_variants' keep mask (the spliced path's filter)."""
    from genvarloader_amd import synth

    rng = np.random.default_rng(41)
    st = synth.make_static(rng, (300_000,), indel_frac=0.3)
    bt = synth.make_batch(rng, st, 200, 2, 1024, output_length=-1, lookback=60)
    sv = synth.to_svar2(rng, st, bt, dense_af=0.3)
    exp, eoff = oracle.reconstruct_haplotypes_from_svar2(bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, -1,
                                                         filter_exonic=True)
    got, off = sv2.reconstruct_haplotypes_from_svar2(bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, -1,
                                                     filter_exonic=True)
    np.testing.assert_array_equal(off, eoff)
    np.testing.assert_array_equal(got, exp)
    keep, ko = gpu.ffi.choose_exonic_variants(np.ascontiguousarray(bt.regions[:, 1]), np.ascontiguousarray(bt.regions[:, 2]),
                                              bt.geno_offset_idx, bt.geno_v_idxs, bt.geno_offsets, st.v_starts, st.ilens)
    g1, o1 = gpu.ffi.reconstruct_haplotypes_fused(bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs,
                                                  st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets,
                                                  st.pad_char, -1, keep, ko)
    np.testing.assert_array_equal(o1, off)
    np.testing.assert_array_equal(g1, got)


def test_long_windows_multi_round_merge(gpu, sv2, oracle, kpath):
    """Enformer-length rows (BASELINE config 4's shape, 16 x 2 x 131 072): hundreds of entries per channel, so the merge runs many
    rounds per haplotype; the bytes equal the oracle's and the SVAR1 route's down every kernel path."""
    from genvarloader_amd import synth

    rng = np.random.default_rng(51)
    st = synth.make_static(rng, (2 << 20,), indel_frac=0.15)
    bt = synth.make_batch(rng, st, 16, 2, 131072, random_shifts=True)
    sv = synth.to_svar2(rng, st, bt, dense_af=0.75, extra=0.3)
    assert (np.diff(sv.vk_off) > 64).any() and (np.diff(sv.dense_range.reshape(-1, 2), axis=1) > 64).any()
    exp, eoff = oracle.reconstruct_haplotypes_from_svar2(bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, 131072)
    got, off = sv2.reconstruct_haplotypes_from_svar2(bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, 131072)
    np.testing.assert_array_equal(got, exp)
    if kpath == 0:
        _check_merge(gpu, sv2, oracle, bt.regions, 2, sv, st.ref, st.ref_offsets)
        dev = make_dev(gpu, st, bt)
        o1 = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, 131072, haps=True)
        np.testing.assert_array_equal(o1.haps.cpu().numpy(), got)


@pytest.mark.parametrize("strategy", [0, 1, 2, 3, 4])
def test_tracks_over_merged_table(gpu, sv2, oracle, strategy):
    """shift_and_realign_tracks_from_svar2 (src/ffi/mod.rs:1835-1966): every insertion-fill strategy, HIP == oracle (f32 bit patterns)."""
    st, bt, sv = _synth(61 + strategy, 64, 2048, 0.3, 0.3, out_len=-1, rc=0.0, edge=0.0)
    rng = np.random.default_rng(7)
    # track length per query as the reference sizes it (_reconstruct.py:191): len - min_p(min(diff, 0))
    d = oracle.hap_diffs_svar2(bt.regions, 2, sv.vk_pos, sv.vk_ilen, sv.vk_off, sv.dense_pos, sv.dense_ilen, sv.dense_range,
                               sv.dense_present, sv.dense_present_off)
    tlen = (bt.regions[:, 2] - bt.regions[:, 1]).astype(np.int64) - np.minimum(d.min(axis=1), 0)
    toff = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
    tracks = rng.random(int(toff[-1])).astype(np.float32)
    args = (bt.regions, bt.shifts, sv.vk_pos, sv.vk_ilen, sv.vk_off, sv.dense_pos, sv.dense_ilen, sv.dense_range, sv.dense_present,
            sv.dense_present_off, tracks, toff, [0.25], strategy, 12345)
    exp, eoff = oracle.shift_and_realign_tracks_from_svar2(*args)
    got, off = sv2.shift_and_realign_tracks_from_svar2(*args)
    np.testing.assert_array_equal(off, eoff)
    np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_scatter_write_random_bounds(gpu, sv2, oracle):
    """Rows at shuffled, gapped destinations (gvl_batch.out_bounds): every row lands where its pair says, the gaps keep their bytes."""
    st, bt, sv = _synth(71, 150, 512, 0.2, 0.3, out_len=-1, rc=0.0)
    d = oracle.hap_diffs_svar2(bt.regions, 2, sv.vk_pos, sv.vk_ilen, sv.vk_off, sv.dense_pos, sv.dense_ilen, sv.dense_range,
                               sv.dense_present, sv.dense_present_off)
    lens = np.maximum((bt.regions[:, 2] - bt.regions[:, 1]).astype(np.int64)[:, None] + d, 0).reshape(-1)
    rng = np.random.default_rng(5)
    order = rng.permutation(len(lens))
    gaps = rng.integers(0, 7, len(lens))
    starts = np.zeros(len(lens), np.int64)
    cur = 3
    for k in order:
        starts[k] = cur
        cur += lens[k] + gaps[k]
    bounds = np.stack([starts, starts + lens], 1)
    init = rng.integers(0, 255, cur + 5).astype(np.uint8)
    exp, got = init.copy(), init.copy()
    args = (bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char)
    oracle.reconstruct_haplotypes_from_svar2_into(exp, bounds, *args)
    sv2.reconstruct_haplotypes_from_svar2_into(got, bounds, *args)
    np.testing.assert_array_equal(got, exp)
    # overlapping / out-of-range bounds are refused on the host (check_disjoint_bounds_within, src/ffi/mod.rs:101-139)
    bad = bounds.copy()
    bad[order[1], 0] = bad[order[0], 0]
    with pytest.raises(ValueError):
        sv2.reconstruct_haplotypes_from_svar2_into(got, bad, *args)
    bad = bounds.copy()
    bad[0, 1] = len(got) + 1
    with pytest.raises(ValueError):
        sv2.reconstruct_haplotypes_from_svar2_into(got, bad, *args)


def test_device_side_reports(gpu, sv2, oracle):
    """What only the device can see is reported, never silent: a long unsorted run, a negative position, offsets outside
    their arrays (gvl_async_error codes 4 / 5 / 6)."""
    from genvarloader_amd import _lib, ffi

    ref = np.frombuffer(b"ACGT" * 500, np.uint8)
    dev = ffi._ref_static(ref, np.array([0, 2000], np.int64), ord("N"))
    regions = np.array([[0, 0, 1000]], np.int32)

    def run(vk_pos, vk_off=None, dense_range=((0, 0),), poff=(0, 0)):
        n = len(vk_pos)
        ch = sv2.Svar2Channels(np.asarray(vk_pos, np.int32), np.zeros(n, np.int32), np.arange(n + 1, dtype=np.int64),
                               np.asarray(vk_off if vk_off is not None else [0, n], np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32),
                               np.zeros(1, np.int64), np.asarray(dense_range, np.int32), np.zeros(1, np.uint8), np.asarray(poff, np.int64),
                               np.full(max(n, 1), ord("A"), np.uint8))
        m = sv2.merge(dev, ch, regions, 1)
        gpu.torch.cuda.synchronize()
        return m

    _lib.check_async()
    run(list(range(100)))                                    # sorted, two rounds: fine
    _lib.check_async()
    run(list(range(100))[::-1])                              # unsorted and longer than a tile
    with pytest.raises(ValueError, match="not position-sorted"):
        _lib.check_async()
    run(list(range(40))[::-1])                               # unsorted but one tile: ranked exactly, no report
    _lib.check_async()
    run([5, -3, 9])
    with pytest.raises(ValueError, match="negative position"):
        _lib.check_async()
    run([1, 2, 3], vk_off=[0, 7])                            # the slice leaves the channel
    with pytest.raises(ValueError, match="leave their arrays"):
        _lib.check_async()
    run([1, 2, 3], dense_range=((0, 4),), poff=(0, 4))       # the window leaves the dense channel
    with pytest.raises(ValueError, match="leave their arrays"):
        _lib.check_async()
    _lib.check_async()


def test_c_abi_fused_entry_and_refusals(gpu, sv2, oracle):
    """gvl_svar2_reconstruct (merge + reconstruct in one call) and what the SVAR2 path refuses: keep masks, annotations, a workspace
    that is too small."""
    import ctypes as C

    from genvarloader_amd import _lib, ffi
    from genvarloader_amd.device import _dev, _ptr, _stream_ptr

    torch = gpu.torch
    st, bt, sv = _synth(81, 96, 1024, 0.2, 0.3, rc=0.5)
    dev = ffi._ref_static(st.ref, st.ref_offsets, st.pad_char)
    ch = sv2.Svar2Channels(*sv.args())
    lib = _lib.load()
    n = bt.n_windows
    nbytes = int(lib.gvl_svar2_workspace_bytes(96, 2, ch.c.n_vk, ch.c.dense_present_bits, ch.c.alt_len))
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device="cuda")
    wp = ws.data_ptr() + (-ws.data_ptr()) % 256
    reg, sh, rc = _dev(bt.regions, torch.int32, "cuda"), _dev(bt.shifts, torch.int32, "cuda"), _dev(bt.to_rc, torch.uint8, "cuda")
    haps = torch.empty(n * 1024, dtype=torch.uint8, device="cuda")
    oh = torch.empty((n * 1024, 4), dtype=torch.uint8, device="cuda")
    b = _lib.GvlBatch(regions=reg.data_ptr(), regions_stride=4, shifts=sh.data_ptr(), batch=96, ploidy=2, to_rc=rc.data_ptr(),
                      output_length=1024, max_row_len=1024)
    o = _lib.GvlOut(haps=haps.data_ptr(), onehot=oh.data_ptr(), onehot_layout=_lib.GVL_ONEHOT_LC)
    _lib.check(lib.gvl_svar2_reconstruct(C.byref(dev.c), C.byref(ch.c), C.byref(b), C.byref(o), C.c_void_p(wp), C.c_int64(nbytes), _stream_ptr()))
    torch.cuda.synchronize()
    _lib.check_async()
    exp, eoff = oracle.reconstruct_haplotypes_from_svar2(bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, 1024)
    oracle.rc_flat_rows_inplace(exp, eoff, bt.to_rc)
    np.testing.assert_array_equal(haps.cpu().numpy(), exp)
    np.testing.assert_array_equal(oh.cpu().numpy(), oracle.onehot(exp))
    with pytest.raises(ValueError, match="workspace"):
        _lib.check(lib.gvl_svar2_reconstruct(C.byref(dev.c), C.byref(ch.c), C.byref(b), C.byref(o), C.c_void_p(wp), C.c_int64(nbytes - 1), _stream_ptr()))
    o2 = _lib.GvlOut(haps=haps.data_ptr(), annot_v_idxs=oh.data_ptr(), annot_ref_pos=oh.data_ptr())
    with pytest.raises(_lib.GvlError, match="no keep masks or annotations"):
        _lib.check(lib.gvl_svar2_reconstruct(C.byref(dev.c), C.byref(ch.c), C.byref(b), C.byref(o2), C.c_void_p(wp), C.c_int64(nbytes), _stream_ptr()))
    _ = _ptr


def test_tracks_query_seed_map(gpu, sv2, oracle):
    """The FlankSample fill's seed takes the GLOBAL batch row through `query_seed` (src/tracks/mod.rs:754-760, gvl_batch.query_seed):
    one logical batch realigned in two calls with the groups' global rows == the single fused call; and == the oracle."""
    st, bt, sv = _synth(91, 40, 1024, 0.4, 0.3, out_len=-1, rc=0.0, edge=0.0)
    rng = np.random.default_rng(9)
    d = oracle.hap_diffs_svar2(bt.regions, 2, sv.vk_pos, sv.vk_ilen, sv.vk_off, sv.dense_pos, sv.dense_ilen, sv.dense_range,
                               sv.dense_present, sv.dense_present_off)
    reg_len = (bt.regions[:, 2] - bt.regions[:, 1]).astype(np.int64)
    tlen = reg_len - np.minimum(d.min(axis=1), 0)
    toff = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
    tracks = rng.random(int(toff[-1])).astype(np.float32)
    lens = np.maximum(reg_len[:, None] + d, 0).reshape(-1)
    ooff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    common = (sv.vk_pos, sv.vk_ilen, sv.vk_off, sv.dense_pos, sv.dense_ilen, sv.dense_range, sv.dense_present, sv.dense_present_off)
    qmap = rng.permutation(1000)[:40].astype(np.int64)            # an arbitrary local -> global map
    exp = np.zeros(int(ooff[-1]), np.float32)
    oracle.shift_and_realign_tracks_from_svar2_into(exp, ooff, bt.regions, bt.shifts, *common, tracks, toff, [6.0], 3, 777, qmap)
    got = np.zeros_like(exp)
    sv2.shift_and_realign_tracks_from_svar2_into(got, ooff, bt.regions, bt.shifts, *common, tracks, toff, [6.0], 3, 777, qmap)
    np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))
    plain = np.zeros_like(exp)
    sv2.shift_and_realign_tracks_from_svar2_into(plain, ooff, bt.regions, bt.shifts, *common, tracks, toff, [6.0], 3, 777, None)
    assert (plain.view(np.uint32) != got.view(np.uint32)).any()    # (the map matters: the draws differ)


def test_reference_consensus_vectors(gpu, sv2, kpath):
    """tests/golden/pyref_svar2_consensus.npz -- the reference's independent `_consensus` over its own VCF fixture and 176 synthetic
    haplotypes -- through the HIP path, down every kernel path."""
    from tests._fixtures import load_svar2_consensus
    from tests.test_oracle_svar2 import _consensus_args

    for i, c in enumerate(load_svar2_consensus()):
        got, off = sv2.reconstruct_haplotypes_from_svar2(*_consensus_args(c))
        np.testing.assert_array_equal(off, c["expected_offsets"], err_msg=f"case {i}")
        np.testing.assert_array_equal(got, c["expected"], err_msg=f"case {i}")


@pytest.mark.parametrize("strategy", [0, 3, 4])
def test_tracks_on_the_reference_tests_del_only_records(gpu, sv2, oracle, strategy):
    """The records of the reference's own end-to-end track test (tests/test_svar2_realign_tracks.py: POS 4 GTA>G, POS 10 GGG>G = pure DELs
    at 3 and 9; S0 1|0 0|1, S1 1|1 1|0), everything in var_key and everything dense: the HIP SVAR2 track driver == the SVAR1 realign of
    the oracle (the reference's assertion) == the oracle's SVAR2 driver, f32 bit patterns."""
    haps = [[0], [1], [0, 1], [0]]
    v_starts, ilens = np.array([3, 9], np.int32), np.array([-2, -2], np.int32)
    regions = np.array([[0, 0, 40], [0, 0, 40]], np.int32)
    go = np.concatenate([[0], np.cumsum([len(h) for h in haps])]).astype(np.int64)
    gv = np.concatenate(haps).astype(np.int32)
    tracks = np.random.default_rng(3).random(80).astype(np.float32)
    toff = np.array([0, 40, 80], np.int64)
    shifts = np.zeros((2, 2), np.int32)
    d1 = oracle.get_diffs_sparse(np.arange(4).reshape(2, 2), gv, go, ilens, None, None, regions[:, 1].copy(), regions[:, 2].copy(), v_starts)
    ooff = np.concatenate([[0], np.cumsum((40 + d1).reshape(-1))]).astype(np.int64)
    exp = np.zeros(int(ooff[-1]), np.float32)
    oracle.shift_and_realign_tracks_sparse(exp, ooff, regions, shifts, np.arange(4).reshape(2, 2), gv, go, v_starts, ilens, tracks, toff,
                                           [2.0], None, None, strategy, 11)
    for dense in (False, True):
        if dense:
            args = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(5, np.int64), v_starts, ilens, np.array([[0, 2], [0, 2]], np.int64),
                    np.packbits(np.array([1, 0, 0, 1, 1, 1, 1, 0], bool), bitorder="little"), np.array([0, 2, 4, 6, 8], np.int64))
        else:
            args = (v_starts[gv], ilens[gv], go, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((2, 2), np.int64), np.zeros(0, np.uint8),
                    np.zeros(5, np.int64))
        got, off = sv2.shift_and_realign_tracks_from_svar2(regions, shifts, *args, tracks, toff, [2.0], strategy, 11)
        np.testing.assert_array_equal(off, ooff)
        np.testing.assert_array_equal(np.asarray(got).view(np.uint32), exp.view(np.uint32), err_msg=f"dense {dense}")
