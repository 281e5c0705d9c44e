"""GPU parity tests proper: the HIP path, called through the C-ABI
(``genvarloader_amd`` -> ``libgvl_hip.so``), against the CPU oracle on the same inputs,
against the committed golden fixtures, and -- at BASELINE.json's full sizes -- through
size-independent properties.  Bit-exact everywhere (integer / byte work).

Run with ``pytest -m gpu`` on an MI355X box.
"""

import numpy as np
import pytest

from tests._fixtures import load_pyref, load_ref_cases
from tests.test_oracle_kats import RC_KATS, ROW_KATS, S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch

    from genvarloader_amd import _lib

    if not torch.cuda.is_available():
        pytest.fail("gpu tests need a HIP device (no CPU fallback exists)")
    _lib.load()  # fail loudly when the HIP library is missing
    import genvarloader_amd.device as device
    import genvarloader_amd.ffi as ffi

    class G:
        pass

    g = G()
    g.torch, g.ffi, g.device = torch, ffi, device
    return g


# Every row can reach its output several ways (scan-free plan, packed plan, per-wave scans,
# scalar walk; records from the genotype-inline layout or from geno_v_idxs -> vrec).  The
# reference's parity suite runs every schedule against the same goldens
# (tests/parity/test_rayon_equivalence.py:31-62); here every reference vector goes down every
# kernel path: gvl_set_debug_flags removes one way at a time.
KERNEL_PATHS = {0: "default", 8: "scalar-walk", 64: "csr-inline-records", 80: "csr-vrec-gather", 32: "no-scan-free-plan",
                128: "no-speculative-reads", 512: "per-wave-scans", 2048: "wave-per-row-diffs",
                16384: "no-lean-kernel", 32768: "lean-lists-every-row", 65536: "lean-rereads-indel-rows", 1048576: "no-lean-long",
                # the lean kernel's pipelined form (gvl_lean_pipe.inc) on ONE workgroup -- a wave takes every fourth row of the
                # batch, many rows per wave --; with every row / every indel row deferred to the wave's end; and never
                33554432: "lean-pipelined", 33554432 + 32768: "lean-pipelined-defers-every-row",
                33554432 + 65536: "lean-pipelined-defers-indel-rows", 67108864: "no-lean-pipeline",
                # rows of several chunks without chunk plans (hap_plan_kernel): every chunk-wave walks its row itself, as in round 4
                536870912: "no-chunk-plans",
                # round 4's routing: channel-major one-hot, keep masks, annotations and get_reference on the all-purpose kernel
                1073741824: "no-lean-forms-of-other-modes"}


@pytest.fixture(params=sorted(KERNEL_PATHS), ids=[KERNEL_PATHS[k] for k in sorted(KERNEL_PATHS)])
def kpath(request, gpu):
    from genvarloader_amd import _lib

    lib = _lib.load()
    lib.gvl_set_debug_flags(int(request.param))
    yield request.param
    lib.gvl_set_debug_flags(-1)


def make_dev(g, st, bt):
    return g.device.HapsDevice(
        ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
        alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, geno_offsets=bt.geno_offsets,
        geno_v_idxs=bt.geno_v_idxs, pad_char=st.pad_char)


def oracle_fused(oracle, st, bt, onehot=False, annotate=False, n_threads=8):
    args = (bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs, st.v_starts,
            st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char,
            bt.output_length, bt.keep, bt.keep_offsets, bt.to_rc, True)
    if annotate:
        return oracle.reconstruct_annotated_haplotypes_fused(*args, n_threads=n_threads)
    return oracle.reconstruct_haplotypes_fused(*args, onehot=onehot, n_threads=n_threads)


def check_batch(g, oracle, st, bt, layout="lc", annotate=False):
    """HIP vs oracle: hap bytes, offsets, one-hot (and annotations)."""
    dev = make_dev(g, st, bt)
    out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length, bt.keep,
                          bt.keep_offsets, bt.to_rc, haps=True, onehot=True, layout=layout,
                          annotate=annotate)
    g.torch.cuda.synchronize()
    exp, exp_off, exp_oh = oracle_fused(oracle, st, bt, onehot=True)
    np.testing.assert_array_equal(out.out_offsets.cpu().numpy(), exp_off)
    np.testing.assert_array_equal(out.haps.cpu().numpy(), exp)
    oh = out.onehot.cpu().numpy()
    if layout == "lc":
        np.testing.assert_array_equal(oh, exp_oh)
    else:
        K = bt.n_windows
        np.testing.assert_array_equal(oh, exp_oh.reshape(K, bt.output_length, 4).transpose(0, 2, 1))
    if annotate:
        _, av, ap, _ = oracle_fused(oracle, st, bt, annotate=True)
        np.testing.assert_array_equal(out.annot_v_idxs.cpu().numpy(), av)
        np.testing.assert_array_equal(out.annot_ref_pos.cpu().numpy(), ap)
    # one-hot only (no hap buffer) must give the same one-hot
    out2 = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length, bt.keep,
                           bt.keep_offsets, bt.to_rc, haps=False, onehot=True, layout=layout)
    np.testing.assert_array_equal(out2.onehot.cpu().numpy(), oh)
    return out


# ------------------------------------------------------------------ reference goldens
def test_golden_reconstruct_haplotypes_from_sparse(gpu, kpath):
    cases = load_ref_cases("reconstruct_haplotypes_from_sparse")
    assert len(cases) == 200
    for ci, (inp, exp) in enumerate(cases):
        out = np.full(int(inp[0][-1]), 0xFF, np.uint8)
        gpu.ffi.reconstruct_haplotypes_from_sparse(out, *inp)
        np.testing.assert_array_equal(out, exp, err_msg=f"case {ci}")


def test_golden_get_diffs_sparse(gpu, kpath):
    cases = load_ref_cases("get_diffs_sparse")
    assert len(cases) == 200
    for ci, (inp, exp) in enumerate(cases):
        got = gpu.ffi.get_diffs_sparse(*inp)
        np.testing.assert_array_equal(got, exp, err_msg=f"case {ci}")


def test_golden_choose_exonic_variants(gpu, oracle):
    """Keep mask of the spliced path: the reference's 200 goldens + its Rust KAT
    (genotypes/mod.rs:215-231), and a synthetic batch against the oracle."""
    cases = load_ref_cases("choose_exonic_variants")
    assert len(cases) == 200
    for ci, (inp, exp) in enumerate(cases):
        keep, ko = gpu.ffi.choose_exonic_variants(*inp)
        np.testing.assert_array_equal(keep, exp[0], err_msg=f"case {ci}")
        np.testing.assert_array_equal(ko, exp[1], err_msg=f"case {ci}")
    keep, ko = gpu.ffi.choose_exonic_variants(np.array([10], np.int32), np.array([20], np.int32), np.array([[0]], np.int64),
                                              np.array([0, 1, 2], np.int32), np.array([[0], [3]], np.int64),
                                              np.array([12, 19, 19], np.int32), np.array([0, 0, -2], np.int32))
    assert keep.tolist() == [True, True, False] and ko.tolist() == [0, 3]
    st, bt = _synth(3, (50_000,), 200, 900, indel_frac=0.4, density=1 / 30)
    exp_keep, exp_ko = oracle.choose_exonic_variants(bt.regions[:, 1] + 100, bt.regions[:, 2] - 300, bt.geno_offset_idx,
                                                     bt.geno_v_idxs, bt.geno_offsets, st.v_starts, st.ilens)
    keep, ko = gpu.ffi.choose_exonic_variants(bt.regions[:, 1] + 100, bt.regions[:, 2] - 300, bt.geno_offset_idx,
                                              bt.geno_v_idxs, bt.geno_offsets, st.v_starts, st.ilens)
    np.testing.assert_array_equal(keep, exp_keep)
    np.testing.assert_array_equal(ko, exp_ko)
    assert 0 < keep.sum() < keep.size


REF_PATHS = {0: "lean-route", 32768: "lean-route-defers-every-row", 33554432: "lean-route-one-workgroup", 1073741824: "all-purpose-kernel"}


@pytest.fixture(params=sorted(REF_PATHS), ids=[REF_PATHS[k] for k in sorted(REF_PATHS)])
def refpath(request, gpu):
    """get_reference's routes: the pipelined lean kernel without slot lines (rows of at most 2560 bases), the same with every row
    handed to the all-purpose body, on one workgroup, and round 4's all-purpose kernel."""
    from genvarloader_amd import _lib

    lib = _lib.load()
    lib.gvl_set_debug_flags(int(request.param))
    yield request.param
    lib.gvl_set_debug_flags(-1)


def test_golden_get_reference(gpu, refpath):
    cases = load_ref_cases("get_reference")
    assert len(cases) == 200
    for ci, (inp, exp) in enumerate(cases):
        regions, out_offsets, reference, ref_offsets, pad_char, parallel = inp
        got = gpu.ffi.get_reference(regions, out_offsets, reference, ref_offsets, pad_char, bool(parallel))
        np.testing.assert_array_equal(got, exp, err_msg=f"case {ci}")


def test_golden_rc_alleles_pins_rc_rows(gpu):
    cases = load_ref_cases("rc_alleles")
    for ci, (inp, exp) in enumerate(cases):
        data, seq_offsets, var_offsets, mask = inp
        per_allele = np.repeat(np.asarray(mask, bool), np.diff(var_offsets))
        if len(per_allele) == 0:
            continue
        t = gpu.torch.from_numpy(np.ascontiguousarray(data, np.uint8).copy()).cuda()
        gpu.device.rc_flat_rows_inplace(t, seq_offsets, per_allele)
        np.testing.assert_array_equal(t.cpu().numpy(), exp, err_msg=f"case {ci}")
        buf = np.ascontiguousarray(data, np.uint8).copy()          # and through the reference's signature
        gpu.ffi.rc_alleles(buf, seq_offsets, var_offsets, mask)
        np.testing.assert_array_equal(buf, exp, err_msg=f"case {ci} (ffi.rc_alleles)")


# ------------------------------------------------------------------ known-answer rows
@pytest.mark.parametrize("kat", ROW_KATS, ids=[k[0] for k in ROW_KATS])
def test_row_kats(gpu, kat, kpath):
    (_, v_idxs, v_starts, ilens, shift, alt, alt_off, ref, ref_start, L, pad, keep,
     exp, exp_av, exp_ap) = kat
    n = len(v_idxs)
    common = dict(
        regions=np.array([[0, ref_start, ref_start + L]], np.int32), shifts=np.array([[shift]], np.int32),
        geno_offset_idx=np.array([[0]], np.int64), geno_offsets=np.array([[0], [n]], np.int64),
        geno_v_idxs=np.array(v_idxs, np.int32), v_starts=np.array(v_starts, np.int32),
        ilens=np.array(ilens, np.int32), alt_alleles=np.asarray(alt, np.uint8),
        alt_offsets=np.array(alt_off, np.int64), ref_offsets=np.array([0, len(ref)], np.int64),
        pad_char=pad, keep=None if keep is None else np.array(keep, bool),
        keep_offsets=None if keep is None else np.array([0, n], np.int64))
    out, av, ap, oo = gpu.ffi.reconstruct_annotated_haplotypes_fused(
        ref_=np.asarray(ref, np.uint8), output_length=L, **common)
    np.testing.assert_array_equal(out, exp)
    np.testing.assert_array_equal(oo, [0, L])
    if exp_av is not None:
        np.testing.assert_array_equal(av, exp_av)
    if exp_ap is not None:
        np.testing.assert_array_equal(ap, exp_ap)
    out2, _ = gpu.ffi.reconstruct_haplotypes_fused(ref_=np.asarray(ref, np.uint8), output_length=L, **common)
    np.testing.assert_array_equal(out2, exp)


def test_rc_kats(gpu):
    for s, e in RC_KATS:
        if not s:
            continue
        t = gpu.torch.from_numpy(S(s)).cuda()
        gpu.device.rc_flat_rows_inplace(t, [0, len(s)], [True])
        assert t.cpu().numpy().tobytes() == e.encode()
    table = bytes.maketrans(b"ACGT", b"TGCA")
    t = gpu.torch.arange(256, dtype=gpu.torch.uint8).cuda()
    gpu.device.rc_flat_rows_inplace(t, np.arange(257), np.ones(256, bool))
    assert t.cpu().numpy().tobytes() == bytes(table[b] for b in range(256))
    f = gpu.torch.tensor([1.0, 2.0, 3.0, 9.0], dtype=gpu.torch.float32).cuda()
    gpu.device.reverse_flat_rows_inplace(f, [0, 3, 4], [True, False])
    assert f.cpu().tolist() == [3.0, 2.0, 1.0, 9.0]


# ------------------------------------------------------------------ reference numpy fallback vectors
@pytest.mark.parametrize("name", ["cfg2_small", "cfg3_small", "dense_annot", "snp_dups_shifts"])
def test_pyref_fixtures(gpu, name, kpath):
    d = load_pyref(name)
    annotate = d["expected_annot_v_idxs"] is not None
    args = (d["regions"], d["shifts"], d["geno_offset_idx"], d["geno_offsets"], d["geno_v_idxs"],
            d["v_starts"], d["ilens"], d["alt_alleles"], d["alt_offsets"], d["ref"], d["ref_offsets"],
            d["pad_char"], int(d["output_length"]), d["keep"], d["keep_offsets"], d["to_rc"])
    out, oo = gpu.ffi.reconstruct_haplotypes_fused(*args)
    np.testing.assert_array_equal(out, d["expected"])
    if annotate:
        out, av, ap, _ = gpu.ffi.reconstruct_annotated_haplotypes_fused(*args)
        np.testing.assert_array_equal(out, d["expected"])
        np.testing.assert_array_equal(av, d["expected_annot_v_idxs"])
        np.testing.assert_array_equal(ap, d["expected_annot_ref_pos"])
    oh, _ = gpu.ffi.reconstruct_haplotypes_fused_onehot(*args)
    exp_oh = (d["expected"][:, None] == np.frombuffer(b"ACGT", np.uint8)).astype(np.uint8)
    np.testing.assert_array_equal(oh, exp_oh)


# ------------------------------------------------------------------ synthetic, HIP vs oracle
def _synth(seed, contigs, q, L, **kw):
    from genvarloader_amd import synth

    rng = np.random.default_rng(seed)
    skw = {k: kw.pop(k) for k in ("indel_frac", "density", "af_beta", "n_frac", "max_indel") if k in kw}
    st = synth.make_static(rng, contigs, **skw)
    bt = synth.make_batch(rng, st, q, 2, L, **kw)
    return st, bt


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 63, 255, 256, 257, 1023, 2047, 2048, 2049, 4099, 9000])
def test_lengths_snp_indel_rc(gpu, oracle, L):
    st, bt = _synth(100 + L, (50_000, 70_001), 40, L, indel_frac=0.2, density=1 / 40, rc_frac=0.5,
                    random_shifts=True, edge_frac=0.2, permute_csr=True, slack=8)
    check_batch(gpu, oracle, st, bt, annotate=(L % 2 == 1))


@pytest.mark.parametrize("layout", ["lc", "cl"])
def test_cfg2_full(gpu, oracle, layout, kpath):
    from genvarloader_amd import synth

    st, bt = synth.make_config("cfg2", contig=8 << 20)
    assert bt.n_windows == 4096 and bt.output_length == 2048
    check_batch(gpu, oracle, st, bt, layout=layout)


@pytest.mark.parametrize("layout", ["lc", "cl"])
def test_annotated_next_to_a_onehot(gpu, oracle, layout, kpath):
    """Annotated haplotypes (bytes + variant index + reference coordinate per base, src/ffi/mod.rs:2237-2397) with a one-hot in either
    layout next to them -- the pipelined kernel's <OH, HAPS, .., CL, ANN> forms -- on rows with indels, shifts, reverse-complemented rows and
    rows over their contig's edges, lengths that end in a partial 16-base group."""
    for seed, L in ((41, 2048), (42, 1_996), (43, 260)):
        st, bt = _synth(seed, (60_000, 90_001), 120, L, indel_frac=0.3, density=1 / 35, rc_frac=0.5, random_shifts=True, edge_frac=0.15)
        check_batch(gpu, oracle, st, bt, layout=layout, annotate=True)


def test_cfg3_full_with_properties(gpu, oracle, kpath):
    from genvarloader_amd import synth

    st, bt = synth.make_config("cfg3", contig=8 << 20, random_shifts=True)
    assert bt.n_windows == 4096 and bt.output_length == 2048 and bt.to_rc.any()
    out = check_batch(gpu, oracle, st, bt)
    t = gpu.torch
    K, L = bt.n_windows, bt.output_length
    # property: RC folded into the kernel == forward output then rc_flat_rows_inplace
    dev = make_dev(gpu, st, bt)
    fwd = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=None)
    buf = fwd.haps.clone()
    gpu.device.rc_flat_rows_inplace(buf, np.arange(K + 1) * L, bt.to_rc)
    assert t.equal(buf, out.haps)
    # property: RC is an involution
    gpu.device.rc_flat_rows_inplace(buf, np.arange(K + 1) * L, bt.to_rc)
    assert t.equal(buf, fwd.haps)
    # property: fused one-hot == stand-alone one-hot of the hap bytes; <= 1 hot per base
    assert t.equal(gpu.device.onehot(out.haps), out.onehot)
    assert int(out.onehot.sum(dim=1).max()) <= 1
    n_acgt = sum(int((out.haps == c).sum()) for c in b"ACGT")
    assert int(out.onehot.sum()) == n_acgt


@pytest.mark.parametrize("P,per,n_b,last", [(1, 3, 16, 1), (3, 5, 9, 2), (2, 64, 4, 64)], ids=["16x3rows-P1", "9x15rows-P3", "4x128rows-P2"])
def test_many_batches_grid_shapes(gpu, oracle, P, per, n_b, last):
    """gvl_reconstruct_many over the shapes the row / batch arithmetic of the one-grid kernel has to get right: the maximum of 16
    batches, batches of 3 rows, ploidy 3 (no shift for the query index), a last batch of one query; default path and one workgroup."""
    from genvarloader_amd import _lib, synth

    rng = np.random.default_rng(1234 + per)
    st = synth.make_static(rng, (40_000,), density=1 / 50, indel_frac=0.3, af_beta=(0.6, 0.9), max_indel=12)
    L = 300
    nq = per * (n_b - 1) + last
    full = synth.make_batch(rng, st, nq, P, L, rc_frac=0.5, random_shifts=True, edge_frac=0.1, permute_csr=True)
    dev = make_dev(gpu, st, full)
    cuts = [(i * per, min((i + 1) * per, nq)) for i in range(n_b)]
    lib = _lib.load()
    for flags in (0, 33554432, 67108864):
        lib.gvl_set_debug_flags(flags)
        try:
            bts, outs, keep = [], [], []
            for a, b in cuts:
                dbt = dev.prepare_batch(full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], L, to_rc=full.to_rc[a * P:b * P])
                o, oc = dev.alloc_output(dbt, (b - a) * P * L, haps=True, onehot=True)
                bts.append(dbt); outs.append(oc); keep.append(o)
            dev.launch_many(dev.pack_many(bts, outs))
            gpu.torch.cuda.synchronize()
            for i, (a, b) in enumerate(cuts):
                exp, exp_off, exp_oh = oracle.reconstruct_haplotypes_fused(
                    full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], full.geno_offsets, full.geno_v_idxs, st.v_starts,
                    st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, L, None, None, full.to_rc[a * P:b * P], True,
                    onehot=True, n_threads=4)
                np.testing.assert_array_equal(keep[i].haps.cpu().numpy(), exp, err_msg=f"flags {flags} batch {i}")
                np.testing.assert_array_equal(keep[i].onehot.cpu().numpy(), exp_oh, err_msg=f"flags {flags} batch {i}")
        finally:
            lib.gvl_set_debug_flags(-1)


@pytest.mark.parametrize("want", ["onehot", "both", "haps"])
def test_many_batches_in_one_grid(gpu, oracle, want, kpath):
    """gvl_reconstruct_many: batches of one shape share a grid (gvl_lean_pipe.inc: row k of the launch belongs to batch
    k / rows_per_batch, per-batch arrays behind a table) -- five batches of a dataset with a SHORTER last one, one of them
    without a strand mask, their outputs in separate buffers; every batch must equal the oracle's."""
    from genvarloader_amd import synth

    rng = np.random.default_rng(77)
    st = synth.make_static(rng, (60_000, 90_001), density=1 / 60, indel_frac=0.3, af_beta=(0.6, 0.9), max_indel=30)
    L, P = 512, 2
    # ONE genotype CSR for the dataset: draw a big batch, then cut its queries into the launch's batches
    full = synth.make_batch(rng, st, 6 * 9 + 4, P, L, rc_frac=0.5, random_shifts=True, edge_frac=0.15, permute_csr=True)
    dev = make_dev(gpu, st, full)
    cuts = [(0, 9), (9, 18), (18, 27), (27, 36), (36, 45), (45, 54), (54, 58)]           # the last one shorter
    onehot, haps = want in ("onehot", "both"), want in ("both", "haps")
    bts, outs, keep = [], [], []
    for i, (a, b) in enumerate(cuts):
        rc = None if i == 2 else full.to_rc[a * P:b * P]
        dbt = dev.prepare_batch(full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], L, to_rc=rc)
        o, oc = dev.alloc_output(dbt, (b - a) * P * L, haps=haps, onehot=onehot)
        bts.append(dbt); outs.append(oc); keep.append(o)
    dev.launch_many(dev.pack_many(bts, outs))
    gpu.torch.cuda.synchronize()
    for i, (a, b) in enumerate(cuts):
        rc = None if i == 2 else full.to_rc[a * P:b * P]
        exp, exp_off, exp_oh = oracle.reconstruct_haplotypes_fused(
            full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], full.geno_offsets, full.geno_v_idxs, st.v_starts,
            st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, L, None, None, rc, True,
            onehot=True, n_threads=4)
        if haps:
            np.testing.assert_array_equal(keep[i].haps.cpu().numpy(), exp, err_msg=f"batch {i}")
        if onehot:
            np.testing.assert_array_equal(keep[i].onehot.cpu().numpy(), exp_oh, err_msg=f"batch {i}")
        np.testing.assert_array_equal(keep[i].out_offsets.cpu().numpy(), exp_off, err_msg=f"batch {i}")


def test_cfg1_plumbing(gpu, oracle, kpath):
    from genvarloader_amd import synth

    st, bt = synth.make_config("cfg1", contig=4 << 20, ploidy=1)
    assert bt.n_windows == 1024 and bt.output_length == 1024
    check_batch(gpu, oracle, st, bt)


def test_dense_variants_overflow_tables(gpu, oracle):
    # > 64 segments / patches per row: exercises flush + compaction of the lane tables
    st, bt = _synth(5, (30_000,), 32, 3000, indel_frac=0.5, density=1 / 3, af_beta=(5.0, 1.0),
                    rc_frac=0.5, random_shifts=True, edge_frac=0.1)
    assert bt.mean_variants > 300
    check_batch(gpu, oracle, st, bt, annotate=True)
    st, bt = _synth(6, (30_000,), 32, 3000, indel_frac=0.0, density=1 / 2, af_beta=(8.0, 1.0), rc_frac=0.5)
    check_batch(gpu, oracle, st, bt, annotate=True)


def test_long_rows_chunked(gpu, oracle, kpath):
    # Enformer-like rows: several chunks per row, each wave replays the walk to its chunk (fixed-length rows of a
    # multiple of 4 bases: the lean kernel's LONG form; anything else: the all-purpose kernel)
    st, bt = _synth(8, (1 << 20,), 6, 131072, indel_frac=0.15, rc_frac=0.5, random_shifts=True)
    check_batch(gpu, oracle, st, bt)
    st, bt = _synth(9, (1 << 19,), 5, 40_001, indel_frac=0.3, density=1 / 20, rc_frac=0.5, edge_frac=0.4)
    check_batch(gpu, oracle, st, bt, annotate=True)
    # dense rows over contig edges, chunk borders inside long alleles, a last chunk shorter than the others
    st, bt = _synth(10, (1 << 18, 50_000), 7, 40_004, indel_frac=0.5, density=1 / 12, rc_frac=0.5, edge_frac=0.4,
                    random_shifts=True, max_indel=300)
    check_batch(gpu, oracle, st, bt)
    # channel-major one-hot (rows, 4, L) of long rows: the chunked lean kernel's CL form (round 5); a last chunk of 4 / 1028 bases
    check_batch(gpu, oracle, st, bt, layout="cl")
    st, bt = _synth(11, (1 << 19,), 5, 131072, indel_frac=0.15, rc_frac=0.5, random_shifts=True)
    check_batch(gpu, oracle, st, bt, layout="cl")
    st, bt = _synth(12, (1 << 18,), 6, 7172, indel_frac=0.4, density=1 / 30, rc_frac=0.5, edge_frac=0.3)
    check_batch(gpu, oracle, st, bt, layout="cl")


def test_keep_mask(gpu, oracle, kpath):
    """Rows under a keep mask (src/reconstruct/mod.rs:86-90): a random mask and the exonic mask the reference's spliced path makes
    (genotypes/mod.rs:132-176); fixed-length and ragged rows; with annotations (the all-purpose kernel) and without (the pipelined
    lean kernel reads the row's keep bytes with its slot line; GVL_DBG 2^30: the all-purpose kernel as in round 4)."""
    st, bt = _synth(12, (80_000,), 64, 1500, indel_frac=0.25, density=1 / 80, rc_frac=0.3)
    idx = bt.geno_offset_idx.ravel()
    n_per = bt.geno_offsets[1, idx] - bt.geno_offsets[0, idx]
    bt.keep_offsets = np.concatenate([[0], np.cumsum(n_per)]).astype(np.int64)
    bt.keep = np.random.default_rng(1).random(int(bt.keep_offsets[-1])) < 0.6
    assert (n_per > 8).any() and (n_per <= 8).any()          # rows from the slot line and rows that overflow it
    check_batch(gpu, oracle, st, bt, annotate=True)
    check_batch(gpu, oracle, st, bt)
    # exonic keep-mask produced the reference way (genotypes/mod.rs:132-176)
    keep, ko = oracle.choose_exonic_variants(bt.regions[:, 1], bt.regions[:, 2], bt.geno_offset_idx,
                                             bt.geno_v_idxs, bt.geno_offsets, st.v_starts, st.ilens)
    bt.keep, bt.keep_offsets = keep, ko
    check_batch(gpu, oracle, st, bt)
    # ragged rows under the mask (what the spliced path launches: rows at the caller's offsets)
    dev = make_dev(gpu, st, bt)
    out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, -1, bt.keep, bt.keep_offsets, bt.to_rc, haps=True, onehot=True)
    bt.output_length = -1
    exp, exp_off, exp_oh = oracle_fused(oracle, st, bt, onehot=True)
    np.testing.assert_array_equal(out.out_offsets.cpu().numpy(), exp_off)
    np.testing.assert_array_equal(out.haps.cpu().numpy(), exp)
    np.testing.assert_array_equal(out.onehot.cpu().numpy(), exp_oh)


def test_keep_mask_many_batches_one_grid(gpu, oracle):
    """Six batches of 1 500 queries under keep masks (one of them WITHOUT a mask) in one gvl_reconstruct_many call = one grid of the
    pipelined kernel, two rows per wave: every batch against the oracle; keep bytes at every alignment (rows of 0-12 variants)."""
    from genvarloader_amd import _lib

    st, full = _synth(14, (400_000, 200_000), 9000, 512, indel_frac=0.3, density=1 / 60, rc_frac=0.5, random_shifts=True, permute_csr=True)
    P = 2
    idx = full.geno_offset_idx.ravel()
    n_per = full.geno_offsets[1, idx] - full.geno_offsets[0, idx]
    ko_all = np.concatenate([[0], np.cumsum(n_per)]).astype(np.int64)
    keep_all = np.random.default_rng(2).random(int(ko_all[-1])) < 0.7
    dev = make_dev(gpu, st, full)
    cuts = [(i * 1500, (i + 1) * 1500) for i in range(6)]
    bts, outs, keepo = [], [], []
    for i, (a, b) in enumerate(cuts):
        ko = ko_all[a * P:b * P + 1] - ko_all[a * P]
        kp = keep_all[ko_all[a * P]:ko_all[b * P]]
        kw = dict() if i == 3 else dict(keep=kp, keep_offsets=ko)
        dbt = dev.prepare_batch(full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], 512, to_rc=full.to_rc[a * P:b * P], **kw)
        o, oc = dev.alloc_output(dbt, (b - a) * P * 512, haps=True, onehot=True)
        bts.append(dbt); outs.append(oc); keepo.append((o, kw))
    _lib.set_tuning(_lib.TUNE_PIPE_ROWS_X100, 200)
    try:
        dev.launch_many(dev.pack_many(bts, outs))
        gpu.torch.cuda.synchronize()
        _lib.check_async()
    finally:
        _lib.set_tuning(_lib.TUNE_PIPE_ROWS_X100, 0)
    for i, (a, b) in enumerate(cuts):
        o, kw = keepo[i]
        exp, exp_off, exp_oh = oracle.reconstruct_haplotypes_fused(
            full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], full.geno_offsets, full.geno_v_idxs, st.v_starts,
            st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, 512, kw.get("keep"), kw.get("keep_offsets"),
            full.to_rc[a * P:b * P], True, onehot=True, n_threads=8)
        np.testing.assert_array_equal(o.haps.cpu().numpy(), exp, err_msg=f"batch {i}")
        np.testing.assert_array_equal(o.onehot.cpu().numpy(), exp_oh, err_msg=f"batch {i}")


@pytest.mark.parametrize("seed", [21, 22])
def test_ragged_mode(gpu, oracle, seed, kpath):
    st, bt = _synth(seed, (60_000, 40_000), 100, 700, indel_frac=0.4, density=1 / 25, rc_frac=0.5,
                    edge_frac=0.1, output_length=-1)
    dev = make_dev(gpu, st, bt)
    out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, -1, to_rc=bt.to_rc, haps=True, onehot=True)
    exp, exp_off, exp_oh = oracle_fused(oracle, st, bt, onehot=True)
    assert len(set(np.diff(exp_off))) > 5  # genuinely ragged
    np.testing.assert_array_equal(out.out_offsets.cpu().numpy(), exp_off)
    np.testing.assert_array_equal(out.haps.cpu().numpy(), exp)
    np.testing.assert_array_equal(out.onehot.cpu().numpy(), exp_oh)
    # device-side sizing alone: diffs + offsets + {total, max}
    b = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, -1)
    oo, tm, diffs = dev.hap_offsets(b, want_diffs=True)
    exp_d = oracle.get_diffs_sparse(bt.geno_offset_idx, bt.geno_v_idxs, bt.geno_offsets, st.ilens, None, None,
                                    bt.regions[:, 1], bt.regions[:, 2], st.v_starts)
    np.testing.assert_array_equal(diffs.cpu().numpy(), exp_d)
    np.testing.assert_array_equal(oo.cpu().numpy(), exp_off)
    assert tm.cpu().tolist() == [int(exp_off[-1]), int(np.diff(exp_off).max())]


def test_long_rows_with_the_callers_chunk_plans(gpu, oracle):
    """gvl_hap_plan: the chunk plans of long rows made ONCE over a table of request rows (what the native loader does per epoch) and
    handed to launches over slices of that table as pointers into it (gvl_batch.hap_plan) == a launch without plans (the chunk-waves replay
    the row's walk) == the oracle.  Rows of 20 chunks, dense enough that some chunks hold more than HP_ENT = 8 entries (flagged: their waves walk the
    row), insertions of hundreds of bases across chunk borders, shifts, windows over the contigs' edges."""
    from genvarloader_amd import _lib

    lib = _lib.load()
    for seed, L, kw in ((3, 40_960, dict(density=1 / 40, indel_frac=0.3, max_indel=300, edge_frac=0.2)),
                        (4, 40_004, dict(density=1 / 9, indel_frac=0.6, max_indel=30, edge_frac=0.0)),
                        (5, 131_072, dict(density=1 / 300, indel_frac=0.15, edge_frac=0.0))):
        st, bt = _synth(seed, (1 << 19, 300_000), 12, L, rc_frac=0.5, random_shifts=True, **kw)
        dev = make_dev(gpu, st, bt)
        exp, exp_off, exp_oh = oracle_fused(oracle, st, bt, onehot=True)
        full = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=bt.to_rc)
        plan = dev.hap_plan(full)
        per_row = int(lib.gvl_hap_plan_bytes(1, L))
        assert plan is not None and plan.numel() == per_row * bt.n_windows
        hdr = plan.view(gpu.torch.int32).view(bt.n_windows, -1, 4 + 8 * 8 + 2 * 8)[:, :, 1].cpu().numpy()      # (header, 8 entries, the annotated rows' annex)
        planned = (hdr & 0x100) != 0
        assert planned.any() and (seed != 4 or (~planned).any())
        P = 2
        for a, b in ((0, 5), (5, 12)):                      # two launches over slices of the table
            sl = dev.prepare_batch(bt.regions[a:b], bt.shifts[a:b], bt.geno_offset_idx[a:b], L, to_rc=bt.to_rc[a * P:b * P],
                                   hap_plan=plan[a * P * per_row:])
            out, oc = dev.alloc_output(sl, (b - a) * P * L, haps=True, onehot=True)
            dev.launch(sl, oc)
            gpu.torch.cuda.synchronize()
            np.testing.assert_array_equal(out.haps.cpu().numpy(), exp[a * P * L:b * P * L], err_msg=f"seed {seed} rows {a}:{b}")
            np.testing.assert_array_equal(out.onehot.cpu().numpy(), exp_oh[a * P * L:b * P * L], err_msg=f"seed {seed} rows {a}:{b}")
        for flags in (0, 536870912):                        # no caller's plan: the replay, with and without the plan route compiled in
            lib.gvl_set_debug_flags(flags)
            try:
                out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, L, to_rc=bt.to_rc, haps=True, onehot=True)
                np.testing.assert_array_equal(out.haps.cpu().numpy(), exp, err_msg=f"seed {seed} flags {flags}")
                np.testing.assert_array_equal(out.onehot.cpu().numpy(), exp_oh, err_msg=f"seed {seed} flags {flags}")
            finally:
                lib.gvl_set_debug_flags(-1)


@pytest.mark.parametrize("outputs", ["onehot+haps", "onehot", "haps"])
def test_ragged_long_rows(gpu, oracle, outputs, kpath):
    """Ragged rows (output_length = -1, the reference's default output) of 3-6 chunks each, lengths that end in 0-3 bases
    beyond a group of four, half of them reverse-complemented, regions over the contigs' edges: the chunked lean kernel's ragged
    form (recon_lean_kernel<.., LONG, RAGL>) by default; down the suite's paths also every chunk SOLO (32768), the re-reading
    form (65536) and the all-purpose kernel (16384 / the scalar paths)."""
    st, bt = _synth(41, (90_000, 60_000), 24, 9_000, indel_frac=0.4, density=1 / 40, rc_frac=0.5, edge_frac=0.2, output_length=-1,
                    random_shifts=True)
    rng = np.random.default_rng(5)
    bt.regions = bt.regions.copy()
    bt.regions[:, 2] += rng.integers(0, 5000, len(bt.regions)).astype(np.int32)        # (rows of 9 000 ... 14 000 bases)
    dev = make_dev(gpu, st, bt)
    oh, hp = "onehot" in outputs, "haps" in outputs
    out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, -1, to_rc=bt.to_rc, haps=hp, onehot=oh)
    exp, exp_off, exp_oh = oracle_fused(oracle, st, bt, onehot=True)
    lens = np.diff(exp_off)
    assert lens.min() > 2560 and len(set(lens % 4)) == 4 and len(set(lens)) > 10
    np.testing.assert_array_equal(out.out_offsets.cpu().numpy(), exp_off)
    if hp:
        np.testing.assert_array_equal(out.haps.cpu().numpy(), exp)
    if oh:
        np.testing.assert_array_equal(out.onehot.cpu().numpy(), exp_oh)


def test_offsets_scan_many_rows(gpu, oracle):
    st, bt = _synth(31, (100_000,), 3000, 64, indel_frac=0.5, density=1 / 10, output_length=-1, slack=0)
    dev = make_dev(gpu, st, bt)
    b = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, -1)
    oo, tm, _ = dev.hap_offsets(b)
    _, exp_off = oracle_fused(oracle, st, bt)
    np.testing.assert_array_equal(oo.cpu().numpy(), exp_off)


def test_spliced_caller_offsets(gpu, oracle):
    # ploidy-1 flattened rows with caller-supplied offsets (ffi/mod.rs:1981-2076)
    st, bt = _synth(41, (50_000,), 60, 300, indel_frac=0.3, density=1 / 30, rc_frac=0.5)
    K = bt.n_windows
    reg = np.repeat(bt.regions, 2, axis=0)
    lens = np.random.default_rng(2).integers(0, 400, K)
    oo = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    got = gpu.ffi.reconstruct_haplotypes_spliced_fused(
        reg, bt.shifts.reshape(-1, 1), bt.geno_offset_idx.reshape(-1, 1), oo, bt.geno_offsets, bt.geno_v_idxs,
        st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char,
        None, None, bt.to_rc)
    exp = np.zeros(int(oo[-1]), np.uint8)
    oracle.reconstruct_haplotypes_from_sparse(
        exp, oo, reg, bt.shifts.reshape(-1, 1), bt.geno_offset_idx.reshape(-1, 1), bt.geno_offsets,
        bt.geno_v_idxs, st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets,
        st.pad_char, to_rc=bt.to_rc)
    np.testing.assert_array_equal(got, exp)


def test_get_reference_synthetic(gpu, oracle, refpath):
    st, bt = _synth(51, (30_000, 10_000, 5), 300, 900, edge_frac=0.5, rc_frac=0.5)
    lens = (bt.regions[:, 2] - bt.regions[:, 1]).astype(np.int64)
    oo = np.concatenate([[0], np.cumsum(lens)])
    to_rc = bt.regions[:, 3] == -1
    exp = oracle.get_reference(bt.regions, oo, st.ref, st.ref_offsets, st.pad_char, True, to_rc)
    got = gpu.ffi.get_reference(bt.regions, oo, st.ref, st.ref_offsets, st.pad_char, True, to_rc)
    np.testing.assert_array_equal(got, exp)
    dev = gpu.ffi._ref_static(st.ref, st.ref_offsets, st.pad_char)
    out, oh = dev.get_reference(bt.regions, oo, to_rc, onehot=True)
    np.testing.assert_array_equal(oh.cpu().numpy(), oracle.onehot(exp))
    np.testing.assert_array_equal(out.cpu().numpy(), exp)
    # empty regions (start >= stop: no bytes), regions that lie in front of their contig (stop < 0: all pad), IUPAC / lower-case
    # bytes (the lean route's byte table cannot write them: those rows are deferred to the all-purpose body)
    rng = np.random.default_rng(9)
    reg = bt.regions.copy()
    reg[::7, 2] = reg[::7, 1] - rng.integers(0, 5, len(reg[::7]))             # start >= stop
    reg[3::11, 1] = -rng.integers(400, 900, len(reg[3::11]))                   # start < stop < 0
    reg[3::11, 2] = -rng.integers(1, 50, len(reg[3::11]))
    ref2 = st.ref.copy()
    ref2[rng.random(ref2.size) < 0.01] = ord("r")
    lens2 = np.clip(reg[:, 2].astype(np.int64) - reg[:, 1], 0, None)
    oo2 = np.concatenate([[0], np.cumsum(lens2)])
    exp2 = oracle.get_reference(reg, oo2, ref2, st.ref_offsets, st.pad_char, True, to_rc)
    got2 = gpu.ffi.get_reference(reg, oo2, ref2, st.ref_offsets, st.pad_char, True, to_rc)
    np.testing.assert_array_equal(got2, exp2)
    assert (exp2 == ord("r")).any()


@pytest.mark.parametrize("flags", [0, 32768, 1073741824], ids=["chunked-lean", "every-chunk-solo", "all-purpose-kernel"])
def test_get_reference_long_rows(gpu, oracle, flags):
    """get_reference rows longer than the pipelined form's 2560 bases (`with_seqs("reference")` at Enformer length): the chunked
    lean kernel's ragged form with no walk (a wave per 2048-base chunk) == its every-chunk-solo route == round 4's all-purpose
    kernel == the oracle.  Lengths of any residue mod 4, rows across their contig's edges (padded), reverse-complemented rows,
    empty rows, IUPAC bytes, rows in front of their contig."""
    from genvarloader_amd import _lib

    lib = _lib.load()
    st, bt = _synth(52, (300_000, 70_000, 9), 60, 20_000, edge_frac=0.4, rc_frac=0.5)
    rng = np.random.default_rng(10)
    reg = bt.regions.copy()
    reg[:, 2] = reg[:, 1] + rng.integers(2_052, 45_000, len(reg))
    reg[::9, 2] = reg[::9, 1] - rng.integers(0, 5, len(reg[::9]))               # start >= stop: nothing written
    reg[4::13, 1] = -rng.integers(4_000, 9_000, len(reg[4::13]))                 # start < stop < 0: all pad
    reg[4::13, 2] = -rng.integers(1, 50, len(reg[4::13]))
    ref2 = st.ref.copy()
    ref2[rng.random(ref2.size) < 0.0005] = ord("r")
    lens = np.clip(reg[:, 2].astype(np.int64) - reg[:, 1], 0, None)
    oo = np.concatenate([[0], np.cumsum(lens)])
    to_rc = reg[:, 3] == -1
    exp = oracle.get_reference(reg, oo, ref2, st.ref_offsets, st.pad_char, True, to_rc)
    assert (exp == ord("r")).any() and (lens > 40_000).any()
    lib.gvl_set_debug_flags(flags)
    try:
        dev = gpu.ffi._ref_static(ref2, st.ref_offsets, st.pad_char)
        out, oh = dev.get_reference(reg, oo, to_rc, onehot=True)
        np.testing.assert_array_equal(out.cpu().numpy(), exp)
        np.testing.assert_array_equal(oh.cpu().numpy(), oracle.onehot(exp))
        out2 = dev.get_reference(reg, oo, to_rc)
        np.testing.assert_array_equal(out2.cpu().numpy(), exp)
    finally:
        lib.gvl_set_debug_flags(-1)


def test_get_diffs_modes(gpu, oracle):
    st, bt = _synth(61, (40_000,), 200, 500, indel_frac=0.5, density=1 / 15)
    idx = bt.geno_offset_idx.ravel()
    n_per = bt.geno_offsets[1, idx] - bt.geno_offsets[0, idx]
    ko = np.concatenate([[0], np.cumsum(n_per)]).astype(np.int64)
    keep = np.random.default_rng(3).random(int(ko[-1])) < 0.5
    qs, qe = bt.regions[:, 1] + 100, bt.regions[:, 2] - 100
    for kw in (dict(), dict(keep=keep, keep_offsets=ko), dict(q_starts=qs, q_ends=qe, v_starts=st.v_starts),
               dict(keep=keep, keep_offsets=ko, q_starts=qs, q_ends=qe, v_starts=st.v_starts)):
        exp = oracle.get_diffs_sparse(bt.geno_offset_idx, bt.geno_v_idxs, bt.geno_offsets, st.ilens, **kw)
        got = gpu.ffi.get_diffs_sparse(bt.geno_offset_idx, bt.geno_v_idxs, bt.geno_offsets, st.ilens, **kw)
        np.testing.assert_array_equal(got, exp)


def test_onehot_standalone(gpu, oracle):
    rng = np.random.default_rng(0)
    for n in (0, 1, 3, 4, 5, 1023, 100_003):
        x = rng.integers(0, 256, n, dtype=np.uint8)
        x[: min(n, 6)] = np.frombuffer(b"ACGTNa", np.uint8)[: min(n, 6)]
        got = gpu.device.onehot(gpu.torch.from_numpy(x).cuda()).cpu().numpy()
        np.testing.assert_array_equal(got, oracle.onehot(x))


def test_empty_and_errors(gpu):
    from genvarloader_amd import synth

    rng = np.random.default_rng(0)
    st = synth.make_static(rng, (10_000,))
    bt = synth.make_batch(rng, st, 4, 2, 100)
    dev = make_dev(gpu, st, bt)
    out = dev.reconstruct(bt.regions[:0], bt.shifts[:0], bt.geno_offset_idx[:0], 100)
    assert out.haps.numel() == 0 and out.out_offsets.cpu().tolist() == [0]
    with pytest.raises(ValueError):
        dev.reconstruct(bt.regions[:, :2], bt.shifts, bt.geno_offset_idx, 100)
    with pytest.raises(ValueError):
        dev.reconstruct(bt.regions, bt.shifts[:, :1], bt.geno_offset_idx, 100)
    with pytest.raises(ValueError):
        dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, 100, haps=False, onehot=False)
    with pytest.raises(ValueError):  # channel-major one-hot is fixed-length only
        dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, -1, onehot=True, layout="cl")


@pytest.mark.parametrize("seed", [71, 72, 73])
def test_shift_stress_scan_path(gpu, oracle, seed):
    """Large shifts (far beyond the reference's max_shift) on insertion-rich rows with <= 64
    variants, so that the scan path's shift logic (drop / cut-into-allele / consume) is hit in
    every sub-case, with and without leading pad and DELs spanning the window start."""
    st, bt = _synth(seed, (30_000, 20_000), 96, 600, indel_frac=0.6, density=1 / 25, rc_frac=0.5,
                    edge_frac=0.3, slack=60, lookback=80)
    rng = np.random.default_rng(seed)
    bt.shifts = rng.integers(0, 300, bt.shifts.shape).astype(np.int32)
    bt.shifts[rng.random(bt.shifts.shape) < 0.2] = 0
    idx = bt.geno_offset_idx.ravel()
    assert (bt.geno_offsets[1, idx] - bt.geno_offsets[0, idx]).max() <= 64
    check_batch(gpu, oracle, st, bt, annotate=True)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_snp_only_rows_fast_plan(gpu, oracle, seed):
    """Rows whose kept variants are all SNPs take the scan-free plan: duplicate positions (first
    ALT wins -> those rows must fall back), shifts that end before / on / after a SNP or run past
    the contig end, windows hanging over both contig edges, keep masks that remove the only indel
    of a row, 0..8 and more than 8 variants per row, ragged and fixed length, annotations."""
    rng = np.random.default_rng(1000 + seed)
    contig = 6000
    ref = rng.choice(np.frombuffer(b"ACGTN", np.uint8), contig, p=[0.24, 0.24, 0.24, 0.24, 0.04]).astype(np.uint8)
    ref_offsets = np.array([0, contig], np.int64)
    # variant table: SNPs every ~9 bp with runs of duplicates, a few indels
    pos = np.sort(rng.integers(0, contig, 700)).astype(np.int32)
    dup = rng.random(pos.size) < 0.12
    pos[1:][dup[1:]] = pos[:-1][dup[1:]]                      # duplicates of the previous position
    pos = np.sort(pos)
    ilens = np.zeros(pos.size, np.int32)
    is_indel = rng.random(pos.size) < 0.06
    ilens[is_indel] = rng.integers(-4, 5, int(is_indel.sum()))
    alts, offs = [], [0]
    for p_, il in zip(pos, ilens):
        n = max(1, 1 + int(il))
        alts.append(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))
        offs.append(offs[-1] + n)
    alt_alleles = np.concatenate(alts).astype(np.uint8)
    alt_offsets = np.asarray(offs, np.int64)
    B, P, L = 40, 2, 333
    starts = rng.integers(-150, contig - 100, B).astype(np.int32)
    regions = np.stack([np.zeros(B, np.int32), starts, starts + L + 40, np.where(rng.random(B) < 0.5, 1, -1).astype(np.int32)], 1)
    lists, keeps = [], []
    for b in range(B):
        lo, hi = np.searchsorted(pos, starts[b] - 20), np.searchsorted(pos, starts[b] + L + 60)
        cand = np.arange(lo, hi)
        for p_ in range(P):
            m = rng.random(cand.size) < rng.choice([0.05, 0.15, 0.4])
            sel = cand[m].astype(np.int32)
            lists.append(sel)
            kp = np.ones(sel.size, bool)
            indel = ilens[sel] != 0
            if indel.any() and rng.random() < 0.5:
                kp[indel] = False                              # the mask leaves a SNP-only row
            kp[rng.random(sel.size) < 0.1] = False
            keeps.append(kp)
    lens = np.array([len(x) for x in lists])
    go = np.stack([np.concatenate([[0], np.cumsum(lens)[:-1]]), np.cumsum(lens)]).astype(np.int64)
    gv = np.concatenate(lists).astype(np.int32) if lens.sum() else np.zeros(0, np.int32)
    keep = np.concatenate(keeps) if lens.sum() else np.zeros(0, bool)
    keep_offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    goi = np.arange(B * P, dtype=np.int64).reshape(B, P)
    shifts = rng.integers(0, 25, (B, P)).astype(np.int32)
    shifts[rng.random((B, P)) < 0.3] = 0
    shifts[0, 0] = 5000                                        # runs past the contig end
    to_rc = np.repeat(regions[:, 3] == -1, P)
    assert (lens > 8).any() and (lens == 0).any()
    for use_keep in (False, True):
        for out_len in (L, -1):
            kw = dict(keep=keep if use_keep else None, keep_offsets=keep_offsets if use_keep else None)
            exp, exp_off, exp_oh = oracle.reconstruct_haplotypes_fused(
                regions, shifts, goi, go, gv, pos, ilens, alt_alleles, alt_offsets, ref, ref_offsets, ord("N"),
                out_len, kw["keep"], kw["keep_offsets"], to_rc, False, onehot=True)
            dev = gpu.device.HapsDevice(ref=ref, ref_offsets=ref_offsets, v_starts=pos, ilens=ilens, alt_alleles=alt_alleles,
                                 alt_offsets=alt_offsets, geno_offsets=go, geno_v_idxs=gv, pad_char=ord("N"))
            out = dev.reconstruct(regions, shifts, goi, out_len, kw["keep"], kw["keep_offsets"], to_rc, haps=True,
                                  onehot=True, annotate=True)
            np.testing.assert_array_equal(out.out_offsets.cpu().numpy(), exp_off)
            np.testing.assert_array_equal(out.haps.cpu().numpy(), exp)
            np.testing.assert_array_equal(out.onehot.cpu().numpy(), exp_oh)
            _, av, ap, _ = oracle.reconstruct_annotated_haplotypes_fused(
                regions, shifts, goi, go, gv, pos, ilens, alt_alleles, alt_offsets, ref, ref_offsets, ord("N"),
                out_len, kw["keep"], kw["keep_offsets"], to_rc, False)
            np.testing.assert_array_equal(out.annot_v_idxs.cpu().numpy(), av)
            np.testing.assert_array_equal(out.annot_ref_pos.cpu().numpy(), ap)


def test_coordinates_beyond_2_30_fall_back_to_the_scalar_walk(gpu, oracle):
    """The scan planners work in i32 with 2^30 head-room; a contig longer than 2^30 bp (and a shift
    of 2^30) must be detected per row and replayed by the i64 scalar walk -- in the same
    workgroups as ordinary rows (small coordinates on a second contig)."""
    rng = np.random.default_rng(5)
    big = (1 << 30) + 6000
    block = rng.choice(np.frombuffer(b"ACGT", np.uint8), 1 << 20).astype(np.uint8)
    ref = np.concatenate([np.tile(block, big // block.size + 1)[:big], block[:50_000]])
    ref_offsets = np.array([0, big, big + 50_000], np.int64)
    # variants at the far end of contig 0 and at the start of contig 1 (positions are per contig)
    far = np.sort(rng.integers((1 << 30) - 2000, (1 << 30) + 5500, 260)).astype(np.int32)
    near = np.sort(rng.integers(100, 40_000, 260)).astype(np.int32)
    pos = np.concatenate([near, far])             # the table is per dataset; rows pick by index
    order = np.argsort(pos, kind="stable")
    pos = pos[order]
    ilens = rng.choice([0, 0, 0, 1, -2, 3, -5], pos.size).astype(np.int32)
    alts, offs = [], [0]
    for il in ilens:
        n = max(1, 1 + int(il))
        alts.append(rng.choice(np.frombuffer(b"ACGT", np.uint8), n)); offs.append(offs[-1] + n)
    alt_alleles = np.concatenate(alts).astype(np.uint8); alt_offsets = np.asarray(offs, np.int64)
    B, P, L = 24, 2, 700
    contig = (np.arange(B) % 2).astype(np.int32)                      # alternate big / small contig
    starts = np.where(contig == 0, rng.integers((1 << 30) - 1500, (1 << 30) + 5200, B), rng.integers(0, 39_000, B)).astype(np.int32)
    regions = np.stack([contig, starts, starts + L + 30, np.where(rng.random(B) < 0.5, 1, -1).astype(np.int32)], 1)
    lists = []
    for b in range(B):
        lo, hi = np.searchsorted(pos, starts[b] - 10), np.searchsorted(pos, starts[b] + L + 40)
        cand = np.arange(lo, hi)
        for _ in range(P):
            lists.append(cand[rng.random(cand.size) < 0.3].astype(np.int32))
    lens = np.array([len(x) for x in lists])
    go = np.stack([np.concatenate([[0], np.cumsum(lens)[:-1]]), np.cumsum(lens)]).astype(np.int64)
    gv = np.concatenate(lists).astype(np.int32)
    goi = np.arange(B * P, dtype=np.int64).reshape(B, P)
    shifts = rng.integers(0, 12, (B, P)).astype(np.int32)
    shifts[3, 1] = 1 << 30
    to_rc = np.repeat(regions[:, 3] == -1, P)
    dev = gpu.device.HapsDevice(ref=ref, ref_offsets=ref_offsets, v_starts=pos, ilens=ilens, alt_alleles=alt_alleles,
                                alt_offsets=alt_offsets, geno_offsets=go, geno_v_idxs=gv, pad_char=ord("N"))
    for out_len in (L, -1):
        exp, exp_off, exp_oh = oracle.reconstruct_haplotypes_fused(
            regions, shifts, goi, go, gv, pos, ilens, alt_alleles, alt_offsets, ref, ref_offsets, ord("N"), out_len,
            None, None, to_rc, False, onehot=True)
        out = dev.reconstruct(regions, shifts, goi, out_len, None, None, to_rc, haps=True, onehot=True, annotate=True)
        np.testing.assert_array_equal(out.out_offsets.cpu().numpy(), exp_off)
        np.testing.assert_array_equal(out.haps.cpu().numpy(), exp)
        np.testing.assert_array_equal(out.onehot.cpu().numpy(), exp_oh)
        _, av, ap, _ = oracle.reconstruct_annotated_haplotypes_fused(
            regions, shifts, goi, go, gv, pos, ilens, alt_alleles, alt_offsets, ref, ref_offsets, ord("N"), out_len,
            None, None, to_rc, False)
        np.testing.assert_array_equal(out.annot_v_idxs.cpu().numpy(), av)
        np.testing.assert_array_equal(out.annot_ref_pos.cpu().numpy(), ap)
    del dev
    gpu.torch.cuda.empty_cache()


# ------------------------------------------------------------------ genome-scale generator (bench.py's dataset)
@pytest.mark.parametrize("scale_kw", [dict(contigs=(3_000_000, 1_000_000, 500_000), n_queries=20_000),
                                      dict(contigs=(700_000,) * 300, n_queries=30_000)],
                         ids=["3-contigs", "300-contigs-no-lds-table"])
def test_genome_dataset_batches(gpu, oracle, kpath, scale_kw):
    """bench.py's device-generated dataset: batches drawn across the whole (multi-contig) genome,
    genotype slots scattered over a large CSR; HIP vs oracle on the compacted host copy."""
    from genvarloader_amd import HapsDevice, synth

    ds = synth.GenomeDataset(device="cuda", seed=11, **scale_kw)
    dev = HapsDevice(**ds.static_kwargs())
    assert dev.slot_rec is not None
    hs = ds.host_static()
    for q in ds.draw_batches(2, 512, seed=5):
        r = ds.request(q)
        out = dev.reconstruct(r["regions"], r["shifts"], r["geno_offset_idx"], ds.length, to_rc=r["to_rc"],
                              haps=True, onehot=True)
        hb = ds.host_batch(q)
        exp, exp_off, exp_oh = oracle.reconstruct_haplotypes_fused(
            hb.regions, hb.shifts, hb.geno_offset_idx, hb.geno_offsets, hb.geno_v_idxs, hs.v_starts, hs.ilens,
            hs.alt_alleles, hs.alt_offsets, hs.ref, hs.ref_offsets, hs.pad_char, ds.length, None, None, hb.to_rc, True,
            onehot=True, n_threads=8)
        np.testing.assert_array_equal(out.haps.cpu().numpy(), exp)
        np.testing.assert_array_equal(out.onehot.cpu().numpy(), exp_oh)


@pytest.mark.parametrize("x100", [0, 200, 300], ids=["default-1.5-rows-per-wave", "exactly-2", "exactly-3"])
@pytest.mark.parametrize("want", ["onehot", "both", "ragged-onehot", "ragged-both", "cl-onehot", "cl-both"])
def test_bench_launch_every_row_vs_oracle(gpu, oracle, want, x100):
    """The launch bench.py times and the native loader submits, at its full size and on DEFAULT flags: 16 batches of 4096 x 2048
    (BASELINE config 3) of one dataset through pack_many / launch_many = ONE grid of recon_lean_rows_kernel over 65 536 rows on
    10 923 workgroups, the first half of the waves taking rows w and w + W.  EVERY batch against the oracle -- the second rows of the
    two-row waves live in batches 10-15: their window + slot line arrive by LDS-DMA under the first row's stores, behind a counted
    vmcnt, and their (batch, row) comes from advance() with W > rows_per_batch.  Also with exactly two and exactly three rows per wave
    (gvl_set_tuning), one-hot only (what the bench's headline launches) and one-hot + bytes, fixed-length and ragged rows
    (output_length = -1: offsets from gvl_hap_offsets, the RAG form)."""
    from genvarloader_amd import HapsDevice, _lib, synth

    ds = synth.GenomeDataset(device="cuda", contigs=(6_000_000, 3_000_000, 2_000_000), n_queries=40_000, seed=20260805)
    dev = HapsDevice(**ds.static_kwargs())
    assert dev.slot_rec is not None and dev.ref4 is not None
    hs = ds.host_static()
    ragged, haps, cl = want.startswith("ragged"), want.endswith("both"), want.startswith("cl")
    if x100 == 300 and (haps or cl):
        pytest.skip("three rows per wave: the one-hot-only forms cover the schedule")
    L = -1 if ragged else ds.length
    qsets = ds.draw_batches(16, 2048, seed=3)
    bts, outs, keep = [], [], []
    for q in qsets:
        r = ds.request(q)
        if ragged:
            d0 = dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], -1, to_rc=r["to_rc"])
            oo, tm, _ = dev.hap_offsets(d0)
            total, mx = (int(v) for v in tm.cpu().tolist())
            dbt = dev.prepare_batch(d0.regions, d0.shifts, d0.geno_offset_idx, -1, to_rc=d0.to_rc, out_offsets=oo, max_row_len=mx)
        else:
            dbt = dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], L, to_rc=r["to_rc"])
            total = dbt.n_rows * L
        o, oc = dev.alloc_output(dbt, total, haps=haps, onehot=True, layout="cl" if cl else "lc")
        bts.append(dbt); outs.append(oc); keep.append(o)
    assert sum(b.n_rows for b in bts) == 65_536
    _lib.check_async()
    _lib.set_tuning(_lib.TUNE_PIPE_ROWS_X100, x100)
    try:
        dev.launch_many(dev.pack_many(bts, outs))
        gpu.torch.cuda.synchronize()
        _lib.check_async()
    finally:
        _lib.set_tuning(_lib.TUNE_PIPE_ROWS_X100, 0)
    n_rc = 0
    for i, q in enumerate(qsets):
        hb = ds.host_batch(q)
        n_rc += int(hb.to_rc.sum())
        exp, exp_off, exp_oh = oracle.reconstruct_haplotypes_fused(
            hb.regions, hb.shifts, hb.geno_offset_idx, hb.geno_offsets, hb.geno_v_idxs, hs.v_starts, hs.ilens,
            hs.alt_alleles, hs.alt_offsets, hs.ref, hs.ref_offsets, hs.pad_char, L, None, None, hb.to_rc, True,
            onehot=True, n_threads=8)
        np.testing.assert_array_equal(keep[i].out_offsets.cpu().numpy(), exp_off, err_msg=f"batch {i}")
        if cl:       # (rows, 4, L): plane c of row k = channel c of the row-major one-hot
            np.testing.assert_array_equal(keep[i].onehot.cpu().numpy(), exp_oh.reshape(-1, ds.length, 4).transpose(0, 2, 1), err_msg=f"batch {i}")
        else:
            np.testing.assert_array_equal(keep[i].onehot.cpu().numpy(), exp_oh, err_msg=f"batch {i}")
        if haps:
            np.testing.assert_array_equal(keep[i].haps.cpu().numpy(), exp, err_msg=f"batch {i}")
        if ragged:
            assert len(set(np.diff(exp_off).tolist())) > 20
    assert 0.4 < n_rc / 65_536 < 0.6


def test_slot_records_layout(gpu):
    """gvl_pack_slots: 8 records per slot, EMPTY padding, OVERFLOW for slots with more than 8 variants."""
    st, bt = _synth(41, (60_000,), 300, 600, indel_frac=0.3, density=1 / 60)
    dev = make_dev(gpu, st, bt)
    sr = dev.slot_rec.cpu().numpy().view(np.uint32).reshape(-1, 8, 4)
    go = bt.geno_offsets
    n = go[1] - go[0]
    assert (n > 8).any() and (n <= 8).any()
    for o in range(go.shape[1]):
        if n[o] > 8:
            assert sr[o, 0, 2] == 0xFFFFFFFE and (sr[o, 1:, 2] == 0xFFFFFFFF).all()
            continue
        v = bt.geno_v_idxs[go[0, o]:go[1, o]]
        assert (sr[o, :n[o], 0].view(np.int32) == st.v_starts[v]).all()
        assert (sr[o, :n[o], 1].view(np.int32) == st.ilens[v]).all()
        alen = np.diff(st.alt_offsets)[v]
        assert (sr[o, :n[o], 2] == ((alen.astype(np.uint32) << 8) | st.alt_alleles[st.alt_offsets[v]])).all()
        assert (sr[o, :n[o], 3] == st.alt_offsets[v]).all()
        assert (sr[o, n[o]:, 2] == 0xFFFFFFFF).all()


# ------------------------------------------------------------------ a8 at realistic sizes
@pytest.mark.parametrize("dtype", ["float32", "int32"])
def test_reverse_flat_rows_4_ragged(gpu, dtype):
    """reverse_flat_rows_inplace<T> (reverse.rs:25-38): ragged rows of odd / even / zero / one
    length, masked rows untouched, f32 (tracks) and i32 (annotations), against numpy."""
    rng = np.random.default_rng(77)
    lens = np.concatenate([[0, 1, 2, 3, 4, 5, 255, 256, 257, 2047, 2048, 2049, 131072, 131071],
                           rng.integers(0, 5000, 300)]).astype(np.int64)
    rng.shuffle(lens)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    mask = rng.random(len(lens)) < 0.6
    if dtype == "float32":
        data = rng.standard_normal(int(offs[-1])).astype(np.float32)
        data[::97] = np.nan                                     # bit patterns must survive
    else:
        data = rng.integers(-2**31, 2**31 - 1, int(offs[-1])).astype(np.int32)
    exp = data.copy()
    for i, m in enumerate(mask):
        if m:
            exp[offs[i]:offs[i + 1]] = exp[offs[i]:offs[i + 1]][::-1]
    t = gpu.torch.from_numpy(data.copy()).cuda()
    gpu.device.reverse_flat_rows_inplace(t, offs, mask)
    np.testing.assert_array_equal(t.cpu().numpy().view(np.uint32), exp.view(np.uint32))
    gpu.device.reverse_flat_rows_inplace(t, offs, mask)         # involution
    np.testing.assert_array_equal(t.cpu().numpy().view(np.uint32), data.view(np.uint32))


# ------------------------------------------------------------------ host hints and caches
def test_wrong_max_row_len_hint_is_reported(gpu, oracle):
    """A max_row_len hint smaller than a row would leave that row partly unwritten; the reference
    writes every byte (src/ffi/mod.rs:17-35), so the launch reports it (gvl_async_error)."""
    from genvarloader_amd import _lib

    st, bt = _synth(51, (60_000,), 20, 3000, indel_frac=0.3, output_length=-1)
    dev = make_dev(gpu, st, bt)
    _, exp_off, _ = oracle_fused(oracle, st, bt, onehot=True)
    lens = np.diff(exp_off)
    assert lens.max() > 2048
    _lib.check_async()                                              # clean slate
    good = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, -1, out_offsets=exp_off, max_row_len=int(lens.max()))
    out, oc = dev.alloc_output(good, int(exp_off[-1]), haps=True, onehot=False)
    dev.launch(good, oc)
    gpu.torch.cuda.synchronize()
    _lib.check_async()                                              # a correct hint: nothing to report
    bad = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, -1, out_offsets=exp_off, max_row_len=2048)
    dev.launch(bad, oc)
    gpu.torch.cuda.synchronize()
    with pytest.raises(ValueError, match="max_row_len"):
        _lib.check_async()
    _lib.check_async()                                              # cleared by the check


def test_ffi_cache_sees_in_place_edits(gpu, oracle):
    """The numpy drop-in layer caches the per-dataset arrays in HBM keyed by the host buffers; an
    in-place edit of a cached array must be a miss, not a stale result."""
    st, bt = _synth(52, (40_000,), 30, 500, indel_frac=0.2)
    args = lambda: (bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs, st.v_starts, st.ilens,
                    st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, bt.output_length)
    a, _ = gpu.ffi.reconstruct_haplotypes_fused(*args())
    st.ref[:] = np.where(st.ref == ord("A"), ord("C"), st.ref)      # same buffer, new content
    b, _ = gpu.ffi.reconstruct_haplotypes_fused(*args())
    exp, _ = oracle.reconstruct_haplotypes_fused(*args(), None, None, None, False)
    np.testing.assert_array_equal(b, exp)
    assert not np.array_equal(a, b)


def test_rc_bounded_rows(gpu, oracle):
    """rc_bounded_rows_inplace (reverse.rs:75-84): rows as (start, end) pairs with gaps between them,
    empty rows and masked rows; the Rust test's vector (reverse.rs:165-178) and a random case vs the oracle."""
    t = gpu.torch.from_numpy(S("ACGT--AACC").copy()).cuda()           # the Rust test's vector
    gpu.device.rc_bounded_rows_inplace(t, np.array([[6, 10], [0, 4]], np.int64), [True, False])
    assert t.cpu().numpy().tobytes() == b"ACGT--GGTT"
    rng = np.random.default_rng(3)
    n = 400
    lens = rng.integers(0, 700, n)
    gaps = rng.integers(0, 9, n)
    starts = np.cumsum(lens + gaps) - lens
    bounds = np.stack([starts, starts + lens], 1).astype(np.int64)
    data = np.frombuffer(b"ACGTNacgtRY", np.uint8)[rng.integers(0, 11, int(bounds[-1, 1]) + 5)].copy()
    mask = rng.random(n) < 0.5
    exp = data.copy()
    oracle.rc_bounded_rows_inplace(exp, bounds, mask)
    t = gpu.torch.from_numpy(data.copy()).cuda()
    gpu.device.rc_bounded_rows_inplace(t, bounds, mask)
    np.testing.assert_array_equal(t.cpu().numpy(), exp)


def test_static_upload_for_hosts_without_an_allocator(gpu, oracle):
    """gvl_static_upload / gvl_static_free: the per-dataset arrays go to HBM from HOST pointers inside the
    library (what a C / Rust caller without torch uses); a reconstruct through that gvl_static == oracle."""
    import ctypes as C

    from genvarloader_amd import _lib
    from genvarloader_amd._lib import GvlBatch, GvlOut, GvlStatic

    lib = _lib.load()
    st, bt = _synth(61, (90_000, 40_000), 300, 800, indel_frac=0.3, rc_frac=0.5, random_shifts=True)
    go = np.ascontiguousarray(bt.geno_offsets)
    keep = [np.ascontiguousarray(a) for a in (st.ref, st.ref_offsets, st.v_starts, st.ilens, st.alt_offsets, st.alt_alleles,
                                              go[0], go[1], bt.geno_v_idxs)]
    ptr = lambda a: a.ctypes.data
    host = GvlStatic(ref=ptr(keep[0]), ref_len=keep[0].size, ref_offsets=ptr(keep[1]), n_contigs=keep[1].size - 1,
                     v_starts=ptr(keep[2]), ilens=ptr(keep[3]), alt_offsets=ptr(keep[4]), alt_alleles=ptr(keep[5]),
                     n_variants=keep[2].size, alt_len=keep[5].size, vrec=None, geno_o_starts=ptr(keep[6]),
                     geno_o_stops=ptr(keep[7]), n_geno_offsets=keep[6].size, geno_v_idxs=ptr(keep[8]), n_geno=keep[8].size,
                     pad_char=st.pad_char, geno_rec=None, slot_rec=None)
    exp, _, exp_oh = oracle_fused(oracle, st, bt, onehot=True)
    t = gpu.torch
    for with_layouts in (1, 0):
        dst = C.POINTER(GvlStatic)()
        _lib.check(lib.gvl_static_upload(C.byref(host), C.c_int32(with_layouts), C.byref(dst), None))
        assert bool(dst.contents.slot_rec) == bool(with_layouts) and bool(dst.contents.vrec)
        reg, sh = t.from_numpy(bt.regions).cuda(), t.from_numpy(bt.shifts).cuda()
        goi, rc = t.from_numpy(bt.geno_offset_idx).cuda(), t.from_numpy(bt.to_rc.astype(np.uint8)).cuda()
        K, L = bt.n_windows, bt.output_length
        haps = t.empty(K * L, dtype=t.uint8, device="cuda")
        oh = t.empty((K * L, 4), dtype=t.uint8, device="cuda")
        b = GvlBatch(regions=reg.data_ptr(), regions_stride=4, shifts=sh.data_ptr(), geno_offset_idx=goi.data_ptr(),
                     batch=reg.shape[0], ploidy=2, to_rc=rc.data_ptr(), output_length=L, max_row_len=L)
        o = GvlOut(haps=haps.data_ptr(), onehot=oh.data_ptr(), onehot_layout=0)
        _lib.check(lib.gvl_reconstruct(dst, C.byref(b), C.byref(o), C.c_void_p(t.cuda.current_stream().cuda_stream)))
        t.cuda.synchronize()
        np.testing.assert_array_equal(haps.cpu().numpy(), exp)
        np.testing.assert_array_equal(oh.cpu().numpy(), exp_oh)
        _lib.check(lib.gvl_static_free(dst))


def test_get_reference_many_batches_one_grid(gpu, oracle, refpath):
    """gvl_get_reference_many: several batches of regions in ONE launch == a gvl_get_reference per batch == the oracle; batches of
    different sizes (the last one shorter), rows over their contigs' edges, reverse-complemented rows; a group whose batches do not
    share a shape (one with rows beyond 2560 bases) runs batch by batch."""
    st, bt = _synth(53, (40_000, 15_000, 7), 500, 700, edge_frac=0.3, rc_frac=0.5)
    dev = gpu.ffi._ref_static(st.ref, st.ref_offsets, st.pad_char)
    reg = bt.regions
    cuts = [(0, 150), (150, 300), (300, 450), (450, 500)]

    def group(regs):
        bs, exp = [], []
        for r in regs:
            lens = np.clip(r[:, 2].astype(np.int64) - r[:, 1], 0, None)
            oo = np.concatenate([[0], np.cumsum(lens)])
            to_rc = r[:, 3] == -1
            exp.append(oracle.get_reference(r, oo, st.ref, st.ref_offsets, st.pad_char, True, to_rc))
            bs.append((r, oo, to_rc, int(oo[-1]), int(lens.max())))
        return bs, exp

    for regs in ([reg[a:b] for a, b in cuts], [reg[:150], np.concatenate([reg[150:299], reg[299:300] * np.array([1, 1, 0, 1]) + np.array([0, 0, reg[299, 1] + 3_000, 0])]).astype(np.int32)]):
        bs, exp = group(regs)
        for kw in (dict(onehot=True, haps=True), dict(onehot=False, haps=True), dict(onehot=True, haps=False)):
            outs = dev.get_reference_many(bs, **kw)
            gpu.torch.cuda.synchronize()
            for (o, oh), e in zip(outs, exp):
                if kw["haps"]:
                    np.testing.assert_array_equal(o.cpu().numpy(), e)
                if kw["onehot"]:
                    np.testing.assert_array_equal(oh.cpu().numpy(), oracle.onehot(e))


# ------------------------------------------------------------------ the shipped library has no wrong-answer mode
# GVL_DBG bits 1 / 2 / 4 / 262144 / 524288 / 8388608 / 16777216 are TIMING ABLATIONS (no variants / no stores / no loads / the lean
# kernel without re-alignment or allele bytes / the track kernel stopping early): they exist in -DGVL_DIAG builds only
# (tools/build_diag.sh); the shipped library masks them off on the host and compiles their tests out of the kernels.
ABLATION_BITS = (1, 2, 4, 6, 262144, 524288, 786432, 8388608, 16777216, 1 | 33554432, 2 | 16384, 262144 | 67108864)


@pytest.mark.parametrize("flags", ABLATION_BITS)
def test_ablation_bits_do_not_change_results_in_the_shipped_library(gpu, oracle, flags):
    from genvarloader_amd import _lib, synth

    from tests.test_gpu_tracks import _track_batch, bits

    lib = _lib.load()
    lib.gvl_set_debug_flags(int(flags))
    try:
        rng = np.random.default_rng(5)
        st = synth.make_static(rng, (400_000,), indel_frac=0.3)
        for n_q, length in ((96, 2048), (5000, 512), (6, 20480)):        # wave-per-row lean / pipelined / chunked long rows
            bt = synth.make_batch(rng, st, n_q, 2, length, rc_frac=0.5, random_shifts=True)
            check_batch(gpu, oracle, st, bt)
        stt, btt, itv = _track_batch(3, 12, 3000, 200_000, shifts=True)
        B, P = btt.geno_offset_idx.shape
        L = btt.output_length
        diffs = oracle.get_diffs_sparse(btt.geno_offset_idx, btt.geno_v_idxs, btt.geno_offsets, stt.ilens, None, None,
                                        btt.regions[:, 1], btt.regions[:, 2], stt.v_starts)
        tlen = (btt.regions[:, 2] - btt.regions[:, 1]) - np.minimum(diffs.min(axis=1), 0)
        track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
        args = (np.arange(B * P + 1, dtype=np.int64) * L, btt.regions, btt.shifts, btt.geno_offset_idx, btt.geno_v_idxs, btt.geno_offsets,
                stt.v_starts, stt.ilens, itv["offset_idxs"], itv["itv_starts"], itv["itv_ends"], itv["itv_values"], itv["itv_offsets"],
                track_offsets, np.array([2.0]), 3, 12345, None, None, btt.to_rc)
        exp = np.full(B * P * L, 7.0, np.float32)
        oracle.intervals_and_realign_track_fused(exp, *args)
        got = np.full(B * P * L, 9.0, np.float32)
        gpu.ffi.intervals_and_realign_track_fused(got, *args)
        np.testing.assert_array_equal(bits(got), bits(exp))
    finally:
        lib.gvl_set_debug_flags(-1)


def test_ablation_bits_from_the_environment_do_not_change_results():
    """GVL_DBG=262144 (once: wrong bytes for rows with indels) in the environment of a fresh process: smoke() still passes."""
    import os
    import subprocess
    import sys

    from tests.conftest import REPO

    for v in ("262144", "7"):
        r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=REPO, capture_output=True, text=True,
                           env={**os.environ, "GVL_DBG": v}, timeout=600)
        assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ------------------------------------------------------------------ round 6: long rows with keep masks / annotations on the chunked lean kernel
def _with_keep(rng, bt, frac=0.7):
    """A random keep mask over the batch's variants (what choose_exonic_variants hands the spliced path), rows in k order."""
    goi = bt.geno_offset_idx.reshape(-1)
    n = (bt.geno_offsets[1, goi] - bt.geno_offsets[0, goi]).astype(np.int64)
    bt.keep_offsets = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    bt.keep = rng.random(int(bt.keep_offsets[-1])) < frac
    return bt


@pytest.mark.parametrize("layout", ["lc", "cl"])
def test_long_rows_annotated(gpu, oracle, kpath, layout):
    """Annotated haplotypes of several chunks (src/ffi/mod.rs:2237-2397 at any length): recon_lean_kernel<.., LONG, .., ANN> -- variant
    indices and positions from the walk's entries -- with reverse-complemented rows, windows over contig edges (the all-purpose body's
    rows), shifts, next to a one-hot in either layout; GVL_DBG 1073741824 = round 5's routing (the all-purpose kernel) for A/B."""
    from genvarloader_amd import synth

    rng = np.random.default_rng(61)
    st = synth.make_static(rng, (500_000,), indel_frac=0.3)
    for n_q, length, edge in ((5, 20480, 0.0), (3, 6148, 0.3), (2, 131072, 0.0)):
        bt = synth.make_batch(rng, st, n_q, 2, length, rc_frac=0.5, random_shifts=True, edge_frac=edge)
        check_batch(gpu, oracle, st, bt, layout=layout, annotate=True)


@pytest.mark.parametrize("ragged", [False, True], ids=["fixed", "ragged"])
def test_long_rows_under_a_keep_mask(gpu, oracle, kpath, ragged):
    """Rows of several chunks under a keep mask (the spliced path's exonic filter on long exons): the chunked lean kernel's walk skips
    masked variants (src/reconstruct/mod.rs:86-90) -- fixed-length rows (with and without chunk plans made under the SAME mask) and
    ragged rows (recon_lean_kernel<.., LONG, RAGL>); annotated too."""
    from genvarloader_amd import synth

    rng = np.random.default_rng(62)
    st = synth.make_static(rng, (500_000,), indel_frac=0.3)
    for n_q, length in ((6, 20480), (3, 7000)):
        bt = _with_keep(rng, synth.make_batch(rng, st, n_q, 2, length, rc_frac=0.5, random_shifts=not ragged, edge_frac=0.1,
                                              output_length=-1 if ragged else None))
        check_batch(gpu, oracle, st, bt)
        if not ragged and length % 4 == 0:
            check_batch(gpu, oracle, st, bt, annotate=True)
            # ... and with the rows' chunk plans, made under the mask (gvl_hap_plan reads bt.keep)
            dev = make_dev(gpu, st, bt)
            b0 = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length, bt.keep, bt.keep_offsets, bt.to_rc)
            plan = dev.hap_plan(b0)
            assert plan is not None
            b1 = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length, bt.keep, bt.keep_offsets, bt.to_rc, hap_plan=plan)
            out, out_c = dev.alloc_output(b1, b1.n_rows * bt.output_length, haps=True, onehot=True)
            dev.launch(b1, out_c)
            gpu.torch.cuda.synchronize()
            exp, _, exp_oh = oracle_fused(oracle, st, bt, onehot=True)
            np.testing.assert_array_equal(out.haps.cpu().numpy(), exp)
            np.testing.assert_array_equal(out.onehot.cpu().numpy(), exp_oh)


@pytest.mark.parametrize("crews", [True, False], ids=["front-workgroups", "solo-at-the-wave's-end"])
@pytest.mark.parametrize("mode", ["plain", "keep", "annotated"])
def test_ragged_batch_of_short_rows_with_a_few_long_ones(gpu, oracle, mode, crews):
    """A spliced batch's shape: hundreds of rows of a few hundred bases and a handful of several thousand (ragged, at out_offsets).  With
    the caller's total (gvl_batch.total_len_hint: HapsDevice.reconstruct passes what it reads for the allocation) the pipelined kernel
    takes the batch -- the long rows by its FRONT workgroups, their chunks in parallel (round 6; GVL_DBG 256: by the solo path of the
    wave that meets them, chunk after chunk) -- instead of handing every row to the chunked kernel.
    == the oracle; the crews (the solo path) saw exactly the long rows (+ whatever else the lean path defers)."""
    import ctypes as C

    from genvarloader_amd import synth

    rng = np.random.default_rng(71)
    st = synth.make_static(rng, (400_000,), indel_frac=0.3)
    bt = synth.make_batch(rng, st, 8300, 2, 300, rc_frac=0.5, output_length=-1, slack=10)      # (16 600 rows: the route needs >= 16 384)
    long_q = rng.choice(8300, 9, replace=False)
    bt.regions[long_q, 2] = bt.regions[long_q, 1] + rng.integers(2600, 9000, 9).astype(np.int32)
    bt.regions[long_q, 2] = np.minimum(bt.regions[long_q, 2], 399_000)
    if mode == "keep":
        _with_keep(rng, bt, 0.6)
    dev = make_dev(gpu, st, bt)
    stamps = gpu.torch.zeros(64, dtype=gpu.torch.int64, device="cuda")
    dev.lib.gvl_diag_set_stamps(C.c_void_p(stamps.data_ptr()))
    dev.lib.gvl_set_debug_flags(0 if crews else 256)
    try:
        out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, -1, bt.keep, bt.keep_offsets, bt.to_rc, haps=True,
                              onehot=mode != "annotated", annotate=mode == "annotated")
        gpu.torch.cuda.synchronize()
    finally:
        dev.lib.gvl_diag_set_stamps(None)
        dev.lib.gvl_set_debug_flags(-1)
    from genvarloader_amd import _lib

    _lib.check_async()
    if mode == "annotated":
        exp, av, ap, exp_off = oracle_fused(oracle, st, bt, annotate=True)
        np.testing.assert_array_equal(out.annot_v_idxs.cpu().numpy(), av)
        np.testing.assert_array_equal(out.annot_ref_pos.cpu().numpy(), ap)
    else:
        exp, exp_off, exp_oh = oracle_fused(oracle, st, bt, onehot=True)
        np.testing.assert_array_equal(out.onehot.cpu().numpy(), exp_oh)
    np.testing.assert_array_equal(out.out_offsets.cpu().numpy(), exp_off)
    np.testing.assert_array_equal(out.haps.cpu().numpy(), exp)
    n_long = int((np.diff(exp_off) > 2560).sum())
    deferred, by_crews = int(stamps[0]), int(stamps[15])
    # (the pipelined kernel ran: its solo path and its crews count what they take)
    if crews:
        assert n_long >= 12 and by_crews == n_long and 0 < deferred + 1 < 2000, (n_long, deferred, by_crews)
    else:
        assert n_long >= 12 and by_crews == 0 and n_long <= deferred < 2000, (n_long, deferred)


@pytest.mark.parametrize("mode", ["fixed-oh", "fixed-oh-bytes", "ragged", "ragged-keep", "fixed-annotated", "fixed-cl"])
@pytest.mark.parametrize("q,L", [(8200, 190), (20000, 300), (9000, 1000)], ids=["16400x190", "40000x300", "18000x1000"])
def test_launches_of_short_rows_many_rows_per_wave(gpu, oracle, mode, q, L):
    """Launches of >= 16 384 SHORT rows (a spliced batch's exons; tools/short_rows.py measures them), at the built-in 1 - 1.5 rows per
    wave and at 9 rows per wave (GVL_TUNE_PIPE_ROWS_X100 = 900: every wave pipelines over many rows) == the oracle in every output mode."""
    from genvarloader_amd import _lib

    ragged = mode.startswith("ragged")
    st, bt = _synth(900 + L, (3_000_000,), q, L, indel_frac=0.3, density=1 / 90, rc_frac=0.5, slack=12,
                    **({"output_length": -1} if ragged else {}))
    if mode == "ragged-keep":
        _with_keep(np.random.default_rng(1), bt, 0.6)
    dev = make_dev(gpu, st, bt)
    kw = dict(haps=mode != "fixed-oh" and mode != "fixed-cl", onehot=mode != "fixed-annotated", annotate=mode == "fixed-annotated",
              layout="cl" if mode == "fixed-cl" else "lc")
    got = []
    for x100 in (0, 900):
        _lib.set_tuning(_lib.TUNE_PIPE_ROWS_X100, x100)
        try:
            out = dev.reconstruct(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length, bt.keep, bt.keep_offsets, bt.to_rc, **kw)
            gpu.torch.cuda.synchronize()
        finally:
            _lib.set_tuning(_lib.TUNE_PIPE_ROWS_X100, 0)
        _lib.check_async()
        got.append(out)
    if mode == "fixed-annotated":
        exp, av, ap, exp_off = oracle_fused(oracle, st, bt, annotate=True)
        for out in got:
            np.testing.assert_array_equal(out.annot_v_idxs.cpu().numpy().ravel(), av)
            np.testing.assert_array_equal(out.annot_ref_pos.cpu().numpy().ravel(), ap)
            np.testing.assert_array_equal(out.haps.cpu().numpy().ravel(), exp)
        return
    exp, exp_off, exp_oh = oracle_fused(oracle, st, bt, onehot=True)
    if mode == "fixed-cl":
        exp_oh = np.ascontiguousarray(exp_oh.reshape(-1, L, 4).transpose(0, 2, 1))
    for out in got:
        np.testing.assert_array_equal(out.onehot.cpu().numpy().ravel(), exp_oh.ravel())
        if kw["haps"]:
            np.testing.assert_array_equal(out.haps.cpu().numpy().ravel(), exp)
        if ragged:
            np.testing.assert_array_equal(out.out_offsets.cpu().numpy(), exp_off)


def test_many_ragged_batches_with_a_few_long_rows(gpu, oracle):
    """gvl_reconstruct_many over three ragged batches of short rows with a handful of long ones each (the same bound on the longest row for
    all of them, as a loader gives): ONE grid, whose front workgroups find the long rows of every batch -- a workgroup's 256 rows may
    straddle two batches -- and run them chunk by chunk; every batch == the oracle, and the crews took exactly the long rows."""
    import ctypes as C

    from genvarloader_amd import _lib, synth

    rng = np.random.default_rng(72)
    st = synth.make_static(rng, (600_000,), indel_frac=0.3)
    P, per = 2, 2830                                   # 5660 rows per batch: not a multiple of 256
    full = synth.make_batch(rng, st, 3 * per, P, 280, rc_frac=0.5, output_length=-1, slack=10)
    long_q = rng.choice(3 * per, 11, replace=False)
    full.regions[long_q, 2] = np.minimum(full.regions[long_q, 1] + rng.integers(2600, 7000, 11).astype(np.int32), 599_000)
    dev = make_dev(gpu, st, full)
    lib = _lib.load()
    _lib.set_tuning(_lib.TUNE_MIXED_MIN_ROWS, 1000)
    stamps = gpu.torch.zeros(64, dtype=gpu.torch.int64, device="cuda")
    try:
        prep, mx = [], 0
        for i in range(3):
            a, b = i * per, (i + 1) * per
            b0 = dev.prepare_batch(full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], -1, to_rc=full.to_rc[a * P:b * P])
            oo, tm, _ = dev.hap_offsets(b0)
            tot, m = (int(x) for x in tm.cpu())
            mx = max(mx, m)
            prep.append((a, b, oo, tot))
        bts, outs, keep = [], [], []
        for a, b, oo, tot in prep:
            dbt = dev.prepare_batch(full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], -1, None, None, full.to_rc[a * P:b * P],
                                    oo, max_row_len=mx, total_len=tot)
            o, oc = dev.alloc_output(dbt, tot, haps=True, onehot=True)
            bts.append(dbt); outs.append(oc); keep.append(o)
        lib.gvl_diag_set_stamps(C.c_void_p(stamps.data_ptr()))
        dev.launch_many(dev.pack_many(bts, outs))
        gpu.torch.cuda.synchronize()
    finally:
        lib.gvl_diag_set_stamps(None)
        _lib.set_tuning(_lib.TUNE_MIXED_MIN_ROWS, 0)
    _lib.check_async()
    n_long = 0
    for i, (a, b, oo, tot) in enumerate(prep):
        exp, exp_off, exp_oh = oracle.reconstruct_haplotypes_fused(
            full.regions[a:b], full.shifts[a:b], full.geno_offset_idx[a:b], full.geno_offsets, full.geno_v_idxs, st.v_starts,
            st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, -1, None, None, full.to_rc[a * P:b * P], True,
            onehot=True, n_threads=4)
        np.testing.assert_array_equal(oo.cpu().numpy(), exp_off, err_msg=f"batch {i}")
        np.testing.assert_array_equal(keep[i].haps.cpu().numpy(), exp, err_msg=f"batch {i}")
        np.testing.assert_array_equal(keep[i].onehot.cpu().numpy(), exp_oh, err_msg=f"batch {i}")
        n_long += int((np.diff(exp_off) > 2560).sum())
    assert n_long >= 15 and int(stamps[15]) == n_long, (n_long, int(stamps[15]))


def test_many_ragged_batches_each_reported_against_its_own_bound(gpu, oracle):
    """Two ragged batches in one grid whose callers gave DIFFERENT bounds on the longest row (1 chunk / 2 chunks): a row the pipelined
    kernel hands to its solo path is reported as "longer than max_row_len" against ITS batch's bound (LeanBatch.max_row_len), not the
    group's smallest -- with GVL_DBG 32768 every row goes that way: no report, both batches == the oracle; a bound that IS too small is
    reported."""
    from genvarloader_amd import _lib, synth

    rng = np.random.default_rng(73)
    st = synth.make_static(rng, (900_000,), indel_frac=0.3)
    P, per = 2, 40
    lib = _lib.load()
    fulls = [synth.make_batch(rng, st, per, P, L, rc_frac=0.5, output_length=-1, slack=10) for L in (1700, 2350)]
    # ONE genotype table for both batches: the second batch's offsets behind the first's
    g0, g1 = fulls
    off = int(g0.geno_v_idxs.size)
    geno_offsets = np.concatenate([g0.geno_offsets, g1.geno_offsets + off], axis=1)
    geno_v_idxs = np.concatenate([g0.geno_v_idxs, g1.geno_v_idxs])
    g1.geno_offset_idx = g1.geno_offset_idx + g0.geno_offsets.shape[1]
    dev = gpu.device.HapsDevice(ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens, alt_alleles=st.alt_alleles,
                                alt_offsets=st.alt_offsets, geno_offsets=geno_offsets, geno_v_idxs=geno_v_idxs, pad_char=st.pad_char)
    for shrink in (False, True):
        lib.gvl_set_debug_flags(32768)
        try:
            bts, outs, keep, offs = [], [], [], []
            for bt in fulls:
                b0 = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, -1, to_rc=bt.to_rc)
                oo, tm, _ = dev.hap_offsets(b0)
                tot, m = (int(x) for x in tm.cpu())
                bound = 2048 if (shrink and bt is g1) else m                   # (2048: one chunk -- the second batch's rows are longer)
                dbt = dev.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, -1, None, None, bt.to_rc, oo, max_row_len=bound, total_len=tot)
                o, oc = dev.alloc_output(dbt, tot, haps=True, onehot=True)
                bts.append(dbt); outs.append(oc); keep.append(o); offs.append(oo)
            dev.launch_many(dev.pack_many(bts, outs))
            gpu.torch.cuda.synchronize()
        finally:
            lib.gvl_set_debug_flags(-1)
        if shrink:
            with pytest.raises(Exception, match="max_row_len"):
                _lib.check_async()
            continue
        _lib.check_async()
        for i, bt in enumerate(fulls):
            exp, exp_off, exp_oh = oracle.reconstruct_haplotypes_fused(
                bt.regions, bt.shifts, bt.geno_offset_idx, geno_offsets, geno_v_idxs, st.v_starts, st.ilens, st.alt_alleles, st.alt_offsets,
                st.ref, st.ref_offsets, st.pad_char, -1, None, None, bt.to_rc, True, onehot=True, n_threads=4)
            assert (np.diff(exp_off).max() > 2048) == (i == 1)
            np.testing.assert_array_equal(offs[i].cpu().numpy(), exp_off)
            np.testing.assert_array_equal(keep[i].haps.cpu().numpy(), exp)
            np.testing.assert_array_equal(keep[i].onehot.cpu().numpy(), exp_oh)
