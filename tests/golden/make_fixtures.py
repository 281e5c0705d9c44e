"""Regenerates tests/golden/*.npz.  Runs ONLY in the build container (it reads
/root/reference); the committed .npz files are what travels.

Two kinds of fixture, both plain data (inputs + expected outputs, no pickles):

1. ``ref_<kernel>.npz`` -- the reference's own frozen goldens
   (``/root/reference/tests/parity/golden/<kernel>.npz``: object arrays of
   ``(inputs, expected)`` produced by the reference's Rust build, see
   ``tests/parity/_golden.py:47-54``) re-packed as flat numeric arrays.
2. ``pyref_<name>.npz`` -- seeded synthetic batches shaped like the BASELINE
   configs (scaled down), with expected haplotype bytes / annotations produced
   by the reference's pure-numpy single-row fallback
   ``reconstruct_haplotype_from_sparse`` (``_dataset/_genotypes.py:125-248``).
   The function body is AST-extracted from the reference at generation time and
   exec'd here; it is never written into this repo.

Usage:  python tests/golden/make_fixtures.py
"""

from __future__ import annotations

import ast
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(REPO))

from genvarloader_amd import synth  # noqa: E402

REF_GOLDENS = [
    "reconstruct_haplotypes_from_sparse",
    "get_diffs_sparse",
    "get_reference",
    "rc_alleles",
    "choose_exonic_variants",
    "shift_and_realign_tracks_sparse",
    "intervals_to_tracks",
    "prng_xorshift64",
    "prng_hash4",
]


def pack_cases(cases) -> dict:
    """[(inputs tuple, expected)] -> flat dict of arrays.  Key scheme:
    ``n`` case count; ``{i}/n_in``; ``{i}/in{j}`` (absent => None);
    ``{i}/exp`` or ``{i}/exp{j}`` for tuple outputs (``{i}/n_exp``)."""
    d = {"n": np.int64(len(cases))}
    for i, (inputs, exp) in enumerate(cases):
        if not isinstance(inputs, (tuple, list)):
            inputs = (inputs,)
        d[f"{i}/n_in"] = np.int64(len(inputs))
        for j, x in enumerate(inputs):
            if x is None:
                continue
            d[f"{i}/in{j}"] = np.asarray(x)
        if isinstance(exp, (tuple, list)):
            d[f"{i}/n_exp"] = np.int64(len(exp))
            for j, x in enumerate(exp):
                d[f"{i}/exp{j}"] = np.asarray(x)
        else:
            d[f"{i}/exp"] = np.asarray(exp)
    for k, v in d.items():
        assert v.dtype != object, (k, v.dtype)
    return d


def convert_reference_goldens():
    for name in REF_GOLDENS:
        src = REF / "tests/parity/golden" / f"{name}.npz"
        cases = list(np.load(src, allow_pickle=True)["cases"])
        np.savez_compressed(HERE / f"ref_{name}.npz", **pack_cases(cases))
        print(f"ref_{name}.npz: {len(cases)} cases")


def extract_function(path: Path, name: str):
    """exec one top-level FunctionDef of a reference module in a numpy-only namespace."""
    tree = ast.parse(path.read_text())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            ns = {"np": np}
            exec(compile(mod, str(path), "exec"), ns)
            return ns[name]
    raise KeyError(name)


def pyref_batch(fallback, st, bt, annotate=False):
    """Drive the reference's single-row fallback over a batch the way the Rust
    batch driver does (reconstruct/mod.rs:376-422), then RC like ffi/mod.rs:842-853."""
    B, P = bt.geno_offset_idx.shape
    K = B * P
    L = bt.output_length
    assert L >= 0
    out = np.zeros(K * L, np.uint8)
    av = np.zeros(K * L, np.int32) if annotate else None
    ap = np.zeros(K * L, np.int32) if annotate else None
    comp = np.arange(256, dtype=np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    for k in range(K):
        q = k // P
        oi = bt.geno_offset_idx.ravel()[k]
        o_s, o_e = bt.geno_offsets[0, oi], bt.geno_offsets[1, oi]
        c = bt.regions[q, 0]
        contig = st.ref[st.ref_offsets[c] : st.ref_offsets[c + 1]]
        keep = None
        if bt.keep is not None:
            keep = bt.keep[bt.keep_offsets[k] : bt.keep_offsets[k + 1]]
        sl = slice(k * L, (k + 1) * L)
        fallback(
            bt.geno_v_idxs[o_s:o_e], st.v_starts, st.ilens, int(bt.shifts.ravel()[k]),
            st.alt_alleles, st.alt_offsets, contig, int(bt.regions[q, 1]), out[sl],
            st.pad_char, keep, None if av is None else av[sl], None if ap is None else ap[sl],
        )
        if bt.to_rc is not None and bt.to_rc[k]:
            out[sl] = comp[out[sl][::-1]]
            if annotate:
                av[sl] = av[sl][::-1]
                ap[sl] = ap[sl][::-1]
    return out, av, ap


def save_pyref(name, st, bt, out, av=None, ap=None):
    d = dict(
        ref=st.ref, ref_offsets=st.ref_offsets, v_starts=st.v_starts, ilens=st.ilens,
        alt_alleles=st.alt_alleles, alt_offsets=st.alt_offsets, pad_char=np.uint8(st.pad_char),
        regions=bt.regions, shifts=bt.shifts, geno_offset_idx=bt.geno_offset_idx,
        geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs,
        output_length=np.int64(bt.output_length), expected=out,
    )
    if bt.to_rc is not None:
        d["to_rc"] = bt.to_rc
    if bt.keep is not None:
        d["keep"] = bt.keep
        d["keep_offsets"] = bt.keep_offsets
    if av is not None:
        d["expected_annot_v_idxs"] = av
        d["expected_annot_ref_pos"] = ap
    np.savez_compressed(HERE / f"pyref_{name}.npz", **d)
    print(f"pyref_{name}.npz: K={bt.n_windows} L={bt.output_length} "
          f"V/row={bt.mean_variants:.2f} bytes={out.size}")


def generate_pyref():
    fb = extract_function(REF / "python/genvarloader/_dataset/_genotypes.py",
                          "reconstruct_haplotype_from_sparse")
    # cfg2-like: SNP only
    rng = np.random.default_rng(20260802 + 2)
    st = synth.make_static(rng, (200_000,), indel_frac=0.0)
    bt = synth.make_batch(rng, st, 48, 2, 512)
    save_pyref("cfg2_small", st, bt, *pyref_batch(fb, st, bt)[:1])
    # cfg3-like: SNP + indel, RC on half, random shifts, some windows over contig edges
    rng = np.random.default_rng(20260802 + 3)
    st = synth.make_static(rng, (60_000, 90_000), indel_frac=0.15, density=1 / 60)
    bt = synth.make_batch(rng, st, 64, 2, 509, rc_frac=0.5, random_shifts=True,
                          edge_frac=0.15, permute_csr=True)
    save_pyref("cfg3_small", st, bt, *pyref_batch(fb, st, bt)[:1])
    # annotated, dense variants (many segments per row), keep mask
    rng = np.random.default_rng(20260802 + 11)
    st = synth.make_static(rng, (20_000,), indel_frac=0.4, density=1 / 6, af_beta=(2.0, 1.0))
    bt = synth.make_batch(rng, st, 24, 2, 700, rc_frac=0.5, random_shifts=True, edge_frac=0.2)
    idx = bt.geno_offset_idx.ravel()
    n_per = bt.geno_offsets[1, idx] - bt.geno_offsets[0, idx]
    bt.keep_offsets = np.concatenate([[0], np.cumsum(n_per)]).astype(np.int64)
    bt.keep = rng.random(int(bt.keep_offsets[-1])) < 0.8
    out, av, ap = pyref_batch(fb, st, bt, annotate=True)
    save_pyref("dense_annot", st, bt, out, av, ap)
    # SNP-dominated rows with duplicate positions ("first ALT wins" among SNPs), shifts that end
    # before / on / after a SNP or run past the contig end, windows over both contig edges, keep
    # masks that remove a row's only indel -- the cases the GPU build plans without scans
    rng = np.random.default_rng(20260802 + 12)
    st = synth.make_static(rng, (9_000,), indel_frac=0.05, density=1 / 25, af_beta=(2.0, 2.0))
    dup = rng.random(st.v_starts.size) < 0.12
    st.v_starts[1:][dup[1:]] = st.v_starts[:-1][dup[1:]]          # still sorted
    bt = synth.make_batch(rng, st, 40, 2, 333, rc_frac=0.5, edge_frac=0.3, slack=40)
    bt.shifts = rng.integers(0, 25, bt.shifts.shape).astype(np.int32)
    bt.shifts[rng.random(bt.shifts.shape) < 0.3] = 0
    bt.shifts[0, 0] = 20_000
    idx = bt.geno_offset_idx.ravel()
    n_per = bt.geno_offsets[1, idx] - bt.geno_offsets[0, idx]
    bt.keep_offsets = np.concatenate([[0], np.cumsum(n_per)]).astype(np.int64)
    keep = rng.random(int(bt.keep_offsets[-1])) < 0.9
    for k in range(idx.size):                                      # half the rows lose their indels
        if k % 2 == 0:
            vs = bt.geno_v_idxs[bt.geno_offsets[0, idx[k]]:bt.geno_offsets[1, idx[k]]]
            keep[bt.keep_offsets[k]:bt.keep_offsets[k + 1]] &= st.ilens[vs] == 0
    bt.keep = keep
    out, av, ap = pyref_batch(fb, st, bt, annotate=True)
    save_pyref("snp_dups_shifts", st, bt, out, av, ap)


def extract_namespace(path: Path, func_names, const_prefix="_"):
    """exec several top-level FunctionDefs + the module's simple UPPER_CASE constants in one
    numpy-only namespace (the functions call each other)."""
    tree = ast.parse(path.read_text())
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in func_names:
            node.decorator_list = []
            body.append(node)
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id.upper() == node.targets[0].id and isinstance(node.value, ast.Constant):
            body.append(node)
    ns = {"np": np}
    exec(compile(ast.Module(body=body, type_ignores=[]), str(path), "exec"), ns)
    return ns


def generate_pyref_tracks():
    """Track realignment at a realistic shape through the reference's pure-numpy fallback
    (_dataset/_tracks.py:621-824), all five insertion-fill strategies, keep masks, shifts."""
    ns = extract_namespace(REF / "python/genvarloader/_dataset/_tracks.py",
                           {"_xorshift64", "_hash4", "_apply_insertion_fill", "shift_and_realign_track_sparse"})
    fb = ns["shift_and_realign_track_sparse"]
    rng = np.random.default_rng(20260802 + 13)
    st = synth.make_static(rng, (40_000,), indel_frac=0.5, density=1 / 40, max_indel=12)
    bt = synth.make_batch(rng, st, 10, 2, 900, rc_frac=0.0, random_shifts=True, slack=60)
    B, P = bt.geno_offset_idx.shape
    L = bt.output_length
    tlen = (bt.regions[:, 2] - bt.regions[:, 1]).astype(np.int64) + 80
    track_offsets = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
    tracks = np.repeat(rng.random(int(track_offsets[-1]) // 7 + 1).astype(np.float32) * 8, 7)[: int(track_offsets[-1])]
    idx = bt.geno_offset_idx.ravel()
    n_per = bt.geno_offsets[1, idx] - bt.geno_offsets[0, idx]
    keep_offsets = np.concatenate([[0], np.cumsum(n_per)]).astype(np.int64)
    keep = rng.random(int(keep_offsets[-1])) < 0.85
    params = {0: 0.0, 1: 0.0, 2: 2.5, 3: 5.0, 4: 3.0}
    d = dict(v_starts=st.v_starts, ilens=st.ilens, regions=bt.regions, shifts=bt.shifts,
             geno_offset_idx=bt.geno_offset_idx, geno_offsets=bt.geno_offsets, geno_v_idxs=bt.geno_v_idxs,
             tracks=tracks, track_offsets=track_offsets, keep=keep, keep_offsets=keep_offsets,
             output_length=np.int64(L), base_seed=np.uint64(424242))
    for s_id, par in params.items():
        for use_keep in (0, 1):
            out = np.zeros(B * P * L, np.float32)
            for k in range(B * P):
                q, h = divmod(k, P)
                kp = keep[keep_offsets[k]:keep_offsets[k + 1]] if use_keep else None
                fb(int(idx[k]), bt.geno_v_idxs, bt.geno_offsets, st.v_starts, st.ilens, int(bt.shifts.ravel()[k]),
                   tracks[track_offsets[q]:track_offsets[q + 1]], int(bt.regions[q, 1]), out[k * L:(k + 1) * L],
                   np.array([par], np.float64), kp, s_id, 424242, q, h)
            d[f"expected_s{s_id}_k{use_keep}"] = out
        d[f"param_s{s_id}"] = np.float64(par)
    np.savez_compressed(HERE / "pyref_tracks.npz", **d)
    print(f"pyref_tracks.npz: rows={B * P} L={L} V/row={bt.mean_variants:.2f}")


def generate_pyref_splice_plan():
    """Splice plans (permutation + offsets that put a ploidy-1 kernel call into spliced layout) from the
    reference's own ``build_splice_plan`` (_dataset/_splice.py:54-160), AST-extracted and exec'd with a
    numpy-only namespace, on seeded random inputs incl. empty pairs, E = 1, 2, 3 and 1-D lengths."""
    import types

    ns = extract_namespace(REF / "python/genvarloader/_utils.py", {"lengths_to_offsets"})
    tree = ast.parse((REF / "python/genvarloader/_dataset/_splice.py").read_text())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "build_splice_plan"]
    ns["SplicePlan"] = lambda **kw: types.SimpleNamespace(**kw)
    exec(compile(ast.Module(body=fn, type_ignores=[]), "_splice.py", "exec"), ns)
    build = ns["build_splice_plan"]
    rng = np.random.default_rng(20260802 + 31)
    d = {}
    n = 0
    for n_rows, n_samples, E, max_el in ((3, 2, 2, 4), (5, 3, 1, 3), (4, 1, 3, 5), (6, 4, 2, 1), (2, 2, 0, 4), (7, 2, 2, 6)):
        per_row = rng.integers(0 if n_rows > 3 else 1, max_el + 1, n_rows)
        pair_len = np.repeat(per_row, n_samples)
        off = np.concatenate([[0], np.cumsum(pair_len)]).astype(np.int64)
        B = int(off[-1])
        lengths = rng.integers(0, 50, (B, E) if E else (B,)).astype(np.int32)
        plan = build(lengths=lengths, splice_row_offsets=off, n_samples=n_samples, n_rows=n_rows)
        d[f"{n}/lengths"], d[f"{n}/offsets"] = lengths, off
        d[f"{n}/n_samples"], d[f"{n}/n_rows"] = np.int64(n_samples), np.int64(n_rows)
        d[f"{n}/permutation"] = np.asarray(plan.permutation, np.int64)
        d[f"{n}/permuted_lengths"] = np.asarray(plan.permuted_lengths, np.int32)
        d[f"{n}/permuted_out_offsets"] = np.asarray(plan.permuted_out_offsets, np.int64)
        d[f"{n}/group_offsets"] = np.asarray(plan.group_offsets, np.int64)
        n += 1
    d["n"] = np.int64(n)
    np.savez_compressed(HERE / "pyref_splice_plan.npz", **d)
    print(f"pyref_splice_plan.npz: {n} plans")


if __name__ == "__main__":
    convert_reference_goldens()
    generate_pyref()
    generate_pyref_tracks()
    generate_pyref_splice_plan()
