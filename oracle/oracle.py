"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; nothing in ``genvarloader_amd/`` does.

Function names, argument order and meaning follow the reference's PyO3 entry
points (``/root/reference/src/ffi/mod.rs``) so a parity test reads like the
reference's own ``tests/parity`` replays:

* ``reconstruct_haplotypes_from_sparse``  ffi/mod.rs:634-655 (in place)
* ``reconstruct_haplotypes_fused``        ffi/mod.rs:724-743
* ``get_diffs_sparse``                    ffi/mod.rs:145-157
* ``choose_exonic_variants``              genotypes/mod.rs:132-176
* ``get_reference``                       ffi/mod.rs:2402-2411
* ``rc_flat_rows_inplace``                reverse.rs:56-69
* ``onehot``                              definition in gvl_oracle.c (unpinned)
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(march: str | None = None, force: bool = False) -> Path:
    """Compile gvl_oracle*.c into libgvl_oracle.so (gcc).  Building the checker
    is not using it."""
    so = _HERE / "libgvl_oracle.so"
    srcs = sorted(_HERE.glob("gvl_oracle*.c"))
    stale = (not so.exists()) or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs)
    if stale or force or march:
        cmd = ["make", "-C", str(_HERE), "-B"]
        if march:
            cmd.append(f"MARCH={march}")
        subprocess.run(cmd, check=True, capture_output=True)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = Path(os.environ["GVL_ORACLE_LIB"]) if os.environ.get("GVL_ORACLE_LIB") else _HERE / "libgvl_oracle.so"
        if not so.exists():
            build()
        _LIB = C.CDLL(str(so))
        _LIB.gvlo_choose_exonic_variants.restype = C.c_int64
    return _LIB


def build_native(out_dir: str | os.PathLike | None = None) -> C.CDLL:
    """A second copy of the oracle compiled ``-march=native`` for THIS host, written to a
    scratch directory (never over ``libgvl_oracle.so``, which is x86-64-v3 so that it runs on
    every box the tree travels to).  Used by ``bench.py``'s ``cpu_baseline`` only."""
    import tempfile

    d = Path(out_dir) if out_dir else Path(tempfile.mkdtemp(prefix="gvl_oracle_native_"))
    d.mkdir(parents=True, exist_ok=True)
    so = d / "libgvl_oracle_native.so"
    srcs = [str(s) for s in sorted(_HERE.glob("gvl_oracle*.c"))]
    subprocess.run(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-fvisibility=hidden", *srcs,
                    "-o", str(so), "-shared", "-pthread", "-lm"], check=True, capture_output=True)
    l = C.CDLL(str(so))
    l.gvlo_choose_exonic_variants.restype = C.c_int64
    return l


class BatchCall:
    """``reconstruct_haplotypes_from_sparse`` (+ fused RC / one-hot) with every argument
    converted ONCE, so that a timing loop measures the C code and not numpy conversions.
    ``run(n_threads)`` is one pass over the batch."""

    def __init__(self, out, out_offsets, regions, shifts, geno_offset_idx, geno_offsets, geno_v_idxs,
                 v_starts, ilens, alt_alleles, alt_offsets, ref, ref_offsets, pad_char, *, to_rc=None,
                 onehot_out=None, library: C.CDLL | None = None):
        self.lib = library or lib()
        go = _starts_stops(geno_offsets)
        self.keep = dict(
            out=out, oo=_c(out_offsets, np.int64), reg=_c(regions, np.int32), sh=_c(shifts, np.int32),
            goi=_c(geno_offset_idx, np.int64), go0=np.ascontiguousarray(go[0]), go1=np.ascontiguousarray(go[1]),
            gv=_c(geno_v_idxs, np.int32), vs=_c(v_starts, np.int32), il=_c(ilens, np.int32),
            aa=_c(alt_alleles, np.uint8), ao=_c(alt_offsets, np.int64), rf=_c(ref, np.uint8),
            ro=_c(ref_offsets, np.int64), rc=_c(to_rc, np.bool_), oh=onehot_out)
        k = self.keep
        batch, ploidy = k["goi"].shape
        self.args = [_p(k["out"]), _p(k["oo"]), _p(k["reg"]), C.c_int64(k["reg"].shape[1]), C.c_int64(batch),
                     C.c_int64(ploidy), _p(k["sh"]), _p(k["goi"]), _p(k["go0"]), _p(k["go1"]), _p(k["gv"]),
                     _p(k["vs"]), _p(k["il"]), _p(k["aa"]), _p(k["ao"]), _p(k["rf"]), _p(k["ro"]),
                     C.c_uint8(int(pad_char)), None, None, None, None, _p(k["rc"]), _p(k["oh"])]

    def run(self, n_threads: int = 1) -> None:
        self.lib.gvlo_reconstruct_batch(*self.args, C.c_int(int(n_threads)))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def _starts_stops(geno_offsets):
    """_dataset/_genotypes.py:13-21 (_as_starts_stops)."""
    o = np.asarray(geno_offsets)
    if o.ndim == 1:
        return np.ascontiguousarray(np.stack([o[:-1], o[1:]]), dtype=np.int64)
    return np.ascontiguousarray(o, dtype=np.int64)


def default_threads() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        return os.cpu_count() or 1


def reconstruct_haplotypes_from_sparse(
    out, out_offsets, regions, shifts, geno_offset_idx, geno_offsets, geno_v_idxs,
    v_starts, ilens, alt_alleles, alt_offsets, ref, ref_offsets, pad_char,
    keep=None, keep_offsets=None, annot_v_idxs=None, annot_ref_pos=None,
    parallel=False, *, to_rc=None, onehot_out=None, n_threads=None,
):
    """In-place batch reconstruct (ffi/mod.rs:634-700).  ``to_rc`` / ``onehot_out``
    are the fused-entry extras (ffi/mod.rs:842-853 and the a10 definition)."""
    assert out.dtype == np.uint8 and out.flags.c_contiguous
    out_offsets = _c(out_offsets, np.int64)
    regions = _c(regions, np.int32)
    shifts = _c(shifts, np.int32)
    goi = _c(geno_offset_idx, np.int64)
    go = _starts_stops(geno_offsets)
    batch, ploidy = goi.shape
    assert regions.shape[0] == batch and shifts.shape == goi.shape
    gv = _c(geno_v_idxs, np.int32)
    vs, il = _c(v_starts, np.int32), _c(ilens, np.int32)
    aa, ao = _c(alt_alleles, np.uint8), _c(alt_offsets, np.int64)
    rf, ro = _c(ref, np.uint8), _c(ref_offsets, np.int64)
    kp, ko = _c(keep, np.uint8 if keep is None else np.bool_), _c(keep_offsets, np.int64)
    rc = _c(to_rc, np.bool_)
    for a in (annot_v_idxs, annot_ref_pos):
        assert a is None or (a.dtype == np.int32 and a.flags.c_contiguous)
    nt = n_threads if n_threads is not None else (default_threads() if parallel else 1)
    go0, go1 = np.ascontiguousarray(go[0]), np.ascontiguousarray(go[1])
    lib().gvlo_reconstruct_batch(
        _p(out), _p(out_offsets), _p(regions), C.c_int64(regions.shape[1]),
        C.c_int64(batch), C.c_int64(ploidy), _p(shifts), _p(goi), _p(go0), _p(go1),
        _p(gv), _p(vs), _p(il), _p(aa), _p(ao), _p(rf), _p(ro), C.c_uint8(int(pad_char)),
        _p(kp), _p(ko), _p(annot_v_idxs), _p(annot_ref_pos), _p(rc), _p(onehot_out),
        C.c_int(nt),
    )


def get_diffs_sparse(
    geno_offset_idx, geno_v_idxs, geno_offsets, ilens, keep=None, keep_offsets=None,
    q_starts=None, q_ends=None, v_starts=None, parallel=False, *, n_threads=None,
):
    """ffi/mod.rs:145-185 -> genotypes/mod.rs:15-125."""
    goi = _c(geno_offset_idx, np.int64)
    go = _starts_stops(geno_offsets)
    batch, ploidy = goi.shape
    diffs = np.zeros((batch, ploidy), np.int32)
    go0, go1 = np.ascontiguousarray(go[0]), np.ascontiguousarray(go[1])
    gv, il = _c(geno_v_idxs, np.int32), _c(ilens, np.int32)
    kp, ko = _c(keep, np.bool_), _c(keep_offsets, np.int64)
    qs, qe, vs = _c(q_starts, np.int32), _c(q_ends, np.int32), _c(v_starts, np.int32)
    nt = n_threads if n_threads is not None else (default_threads() if parallel else 1)
    lib().gvlo_get_diffs_sparse(
        _p(goi), C.c_int64(batch), C.c_int64(ploidy), _p(gv), _p(go0), _p(go1), _p(il),
        _p(kp), _p(ko), _p(qs), _p(qe), C.c_int64(1), _p(vs), _p(diffs), C.c_int(nt),
    )
    return diffs


def choose_exonic_variants(starts, ends, geno_offset_idx, geno_v_idxs, geno_offsets,
                           v_starts, ilens):
    """genotypes/mod.rs:132-176 -> (keep bool[], keep_offsets i64[n+1])."""
    goi = _c(geno_offset_idx, np.int64)
    go = _starts_stops(geno_offsets)
    batch, ploidy = goi.shape
    go0, go1 = np.ascontiguousarray(go[0]), np.ascontiguousarray(go[1])
    st, en = _c(starts, np.int32), _c(ends, np.int32)
    gv, vs, il = _c(geno_v_idxs, np.int32), _c(v_starts, np.int32), _c(ilens, np.int32)
    ko = np.zeros(batch * ploidy + 1, np.int64)
    args = (_p(st), _p(en), _p(goi), C.c_int64(batch), C.c_int64(ploidy), _p(gv),
            _p(go0), _p(go1), _p(vs), _p(il))
    n = lib().gvlo_choose_exonic_variants(*args, None, _p(ko))
    keep = np.zeros(n, np.bool_)
    lib().gvlo_choose_exonic_variants(*args, _p(keep), _p(ko))
    return keep, ko


def fused_out_offsets(regions, diffs, output_length):
    """ffi/mod.rs:794-811."""
    regions = _c(regions, np.int32)
    diffs = _c(diffs, np.int32)
    batch, ploidy = diffs.shape
    oo = np.zeros(batch * ploidy + 1, np.int64)
    lib().gvlo_fused_out_offsets(_p(regions), C.c_int64(regions.shape[1]), C.c_int64(batch),
                                 C.c_int64(ploidy), _p(diffs), C.c_int64(int(output_length)),
                                 _p(oo))
    return oo


def reconstruct_haplotypes_fused(
    regions, shifts, geno_offset_idx, geno_offsets, geno_v_idxs, v_starts, ilens,
    alt_alleles, alt_offsets, ref_, ref_offsets, pad_char, output_length,
    keep=None, keep_offsets=None, to_rc=None, parallel=False, *, onehot=False,
    n_threads=None,
):
    """ffi/mod.rs:724-860: diffs (query mode) -> offsets -> reconstruct -> RC.
    With ``onehot=True`` also returns the (total, 4) uint8 one-hot."""
    regions = _c(regions, np.int32)
    diffs = get_diffs_sparse(
        geno_offset_idx, geno_v_idxs, geno_offsets, ilens, keep, keep_offsets,
        np.ascontiguousarray(regions[:, 1]), np.ascontiguousarray(regions[:, 2]),
        v_starts, parallel, n_threads=n_threads,
    )
    out_offsets = fused_out_offsets(regions, diffs, output_length)
    total = int(out_offsets[-1])
    out = np.empty(total, np.uint8)
    oh = np.empty((total, 4), np.uint8) if onehot else None
    reconstruct_haplotypes_from_sparse(
        out, out_offsets, regions, shifts, geno_offset_idx, geno_offsets, geno_v_idxs,
        v_starts, ilens, alt_alleles, alt_offsets, ref_, ref_offsets, pad_char,
        keep, keep_offsets, None, None, parallel, to_rc=to_rc, onehot_out=oh,
        n_threads=n_threads,
    )
    if onehot:
        return out, out_offsets, oh
    return out, out_offsets


def reconstruct_annotated_haplotypes_fused(
    regions, shifts, geno_offset_idx, geno_offsets, geno_v_idxs, v_starts, ilens,
    alt_alleles, alt_offsets, ref_, ref_offsets, pad_char, output_length,
    keep=None, keep_offsets=None, to_rc=None, parallel=False, *, n_threads=None,
):
    """ffi/mod.rs:2237-2397: fused + two i32 annotation buffers; RC rows have
    their annotations reversed (reverse.rs:25-38)."""
    regions = _c(regions, np.int32)
    diffs = get_diffs_sparse(
        geno_offset_idx, geno_v_idxs, geno_offsets, ilens, keep, keep_offsets,
        np.ascontiguousarray(regions[:, 1]), np.ascontiguousarray(regions[:, 2]),
        v_starts, parallel, n_threads=n_threads,
    )
    out_offsets = fused_out_offsets(regions, diffs, output_length)
    total = int(out_offsets[-1])
    out = np.empty(total, np.uint8)
    av = np.empty(total, np.int32)
    ap = np.empty(total, np.int32)
    reconstruct_haplotypes_from_sparse(
        out, out_offsets, regions, shifts, geno_offset_idx, geno_offsets, geno_v_idxs,
        v_starts, ilens, alt_alleles, alt_offsets, ref_, ref_offsets, pad_char,
        keep, keep_offsets, av, ap, parallel, to_rc=to_rc, n_threads=n_threads,
    )
    return out, av, ap, out_offsets


def get_reference(regions, out_offsets, reference, ref_offsets, pad_char, parallel=False,
                  to_rc=None, *, n_threads=None):
    """ffi/mod.rs:2402-2429 -> reference/mod.rs:56-120."""
    regions = _c(regions, np.int32)
    oo = _c(out_offsets, np.int64)
    rf, ro = _c(reference, np.uint8), _c(ref_offsets, np.int64)
    out = np.zeros(int(oo[-1]), np.uint8)
    rc = _c(to_rc, np.bool_)
    nt = n_threads if n_threads is not None else (default_threads() if parallel else 1)
    lib().gvlo_get_reference(_p(regions), C.c_int64(regions.shape[1]),
                              C.c_int64(regions.shape[0]), _p(oo), _p(rf), _p(ro),
                              C.c_uint8(int(pad_char)), _p(rc), _p(out), C.c_int(nt))
    return out


def rc_flat_rows_inplace(data, offsets, to_rc):
    """reverse.rs:56-69."""
    assert data.dtype == np.uint8 and data.flags.c_contiguous
    oo, rc = _c(offsets, np.int64), _c(to_rc, np.bool_)
    lib().gvlo_rc_rows(_p(data), _p(oo), _p(rc), C.c_int64(len(rc)))


def rc_bounded_rows_inplace(data, bounds, to_rc):
    """reverse.rs:75-84: rows as (start, end) pairs."""
    assert data.dtype == np.uint8 and data.flags.c_contiguous
    bd, rc = _c(bounds, np.int64), _c(to_rc, np.bool_)
    lib().gvlo_rc_bounded_rows(_p(data), _p(bd), _p(rc), C.c_int64(len(rc)))


def reverse_flat_rows_inplace(data, offsets, to_rc):
    """reverse.rs:25-38 for 4-byte elements (f32 / i32)."""
    assert data.dtype.itemsize == 4 and data.flags.c_contiguous
    oo, rc = _c(offsets, np.int64), _c(to_rc, np.bool_)
    lib().gvlo_reverse_rows_4(_p(data), _p(oo), _p(rc), C.c_int64(len(rc)))


def reconstruct_haplotype_from_sparse(
    v_idxs, v_starts, ilens, shift, alt_alleles, alt_offsets, ref, ref_start, out,
    pad_char, keep=None, annot_v_idxs=None, annot_ref_pos=None,
):
    """Single row, same signature as the reference's numpy fallback
    (_dataset/_genotypes.py:125-139) / Rust reconstruct/mod.rs:280-319."""
    assert out.dtype == np.uint8 and out.flags.c_contiguous
    vi = _c(v_idxs, np.int32)
    vs, il = _c(v_starts, np.int32), _c(ilens, np.int32)
    aa, ao = _c(alt_alleles, np.uint8), _c(alt_offsets, np.int64)
    rf = _c(ref, np.uint8)
    kp = _c(keep, np.bool_)
    lib().gvlo_reconstruct_row(
        C.c_int64(len(vi)), _p(vi), _p(vs), _p(il), C.c_int64(int(shift)), _p(aa), _p(ao),
        _p(rf), C.c_int64(len(rf)), C.c_int64(int(ref_start)), _p(out), C.c_int64(len(out)),
        C.c_uint8(int(pad_char)), _p(kp), _p(annot_v_idxs), _p(annot_ref_pos),
    )


def onehot(x, layout: str = "lc"):
    """a10 definition.  ``x`` uint8 (..., L); "lc" -> (..., L, 4); "cl" -> (..., 4, L)."""
    x = np.ascontiguousarray(x, np.uint8)
    L = x.shape[-1] if x.ndim else 0
    rows = int(np.prod(x.shape[:-1])) if x.ndim > 1 else 1
    if layout == "lc":
        out = np.empty(x.shape + (4,), np.uint8)
        lib().gvlo_onehot(_p(x), C.c_int64(rows), C.c_int64(L), C.c_int(0), _p(out))
    else:
        out = np.empty(x.shape[:-1] + (4, L), np.uint8)
        lib().gvlo_onehot(_p(x), C.c_int64(rows), C.c_int64(L), C.c_int(1), _p(out))
    return out


def onehot_numpy(x):
    """The 3-line numpy statement of the same definition (cross-check)."""
    x = np.asarray(x, np.uint8)
    return (x[..., None] == np.frombuffer(b"ACGT", np.uint8)).astype(np.uint8)


# ----------------------------------------------------------------------------- tracks (a12)
def xorshift64(x: int) -> int:
    """src/tracks/mod.rs:31-36."""
    lib().gvlo_xorshift64.restype = C.c_uint64
    return int(lib().gvlo_xorshift64(C.c_uint64(int(x) & 0xFFFFFFFFFFFFFFFF)))


def hash4(a: int, b: int, c: int, d: int) -> int:
    """src/tracks/mod.rs:48-54."""
    lib().gvlo_hash4.restype = C.c_uint64
    m = 0xFFFFFFFFFFFFFFFF
    return int(lib().gvlo_hash4(C.c_uint64(int(a) & m), C.c_uint64(int(b) & m), C.c_uint64(int(c) & m),
                                C.c_uint64(int(d) & m)))


def shift_and_realign_tracks_sparse(
    out, out_offsets, regions, shifts, geno_offset_idx, geno_v_idxs, geno_offsets, v_starts, ilens,
    tracks, track_offsets, params, keep=None, keep_offsets=None, strategy_id=0, base_seed=0,
    parallel=False,
):
    """In place; argument order of the reference's wrapper (_tracks.py:42-60 ->
    src/tracks/mod.rs:495-667)."""
    assert out.dtype == np.float32 and out.flags.c_contiguous
    oo = _c(out_offsets, np.int64)
    regions = _c(regions, np.int32)
    shifts = _c(shifts, np.int32)
    goi = _c(geno_offset_idx, np.int64)
    go = _starts_stops(geno_offsets)
    go0, go1 = np.ascontiguousarray(go[0]), np.ascontiguousarray(go[1])
    gv, vs, il = _c(geno_v_idxs, np.int32), _c(v_starts, np.int32), _c(ilens, np.int32)
    tr, to = _c(tracks, np.float32), _c(track_offsets, np.int64)
    pr = _c(params, np.float64)
    kp, ko = _c(keep, np.bool_), _c(keep_offsets, np.int64)
    batch, ploidy = goi.shape
    lib().gvlo_realign_tracks_batch(
        _p(out), _p(oo), _p(regions), C.c_int64(regions.shape[1]), C.c_int64(batch), C.c_int64(ploidy),
        _p(shifts), _p(goi), _p(gv), _p(go0), _p(go1), _p(vs), _p(il), _p(tr), _p(to), _p(pr), _p(kp),
        _p(ko), C.c_int64(int(strategy_id)), C.c_uint64(int(base_seed) & 0xFFFFFFFFFFFFFFFF))


def intervals_to_tracks(offset_idxs, starts, itv_starts, itv_ends, itv_values, itv_offsets, out,
                        out_offsets, parallel=False):
    """In place (src/intervals.rs:19-126); `out` is zeroed first like the reference."""
    assert out.dtype == np.float32 and out.flags.c_contiguous
    oi, st = _c(offset_idxs, np.int64), _c(starts, np.int32)
    a, b, v = _c(itv_starts, np.int32), _c(itv_ends, np.int32), _c(itv_values, np.float32)
    io, oo = _c(itv_offsets, np.int64), _c(out_offsets, np.int64)
    lib().gvlo_intervals_to_tracks(_p(oi), _p(st), C.c_int64(len(st)), _p(a), _p(b), _p(v), _p(io),
                                   _p(out), _p(oo))


def intervals_and_realign_track_fused(
    out, out_offsets, regions, shifts, geno_offset_idx, geno_v_idxs, geno_offsets, v_starts, ilens,
    offset_idxs, itv_starts, itv_ends, itv_values, itv_offsets, track_offsets, params, strategy_id,
    base_seed, keep=None, keep_offsets=None, to_rc=None, parallel=False,
):
    """src/ffi/mod.rs:2551-2672: paint -> realign -> optional reversal (no complement)."""
    regions = _c(regions, np.int32)
    to = _c(track_offsets, np.int64)
    scratch = np.zeros(int(to[-1]), np.float32)
    intervals_to_tracks(offset_idxs, np.ascontiguousarray(regions[:, 1]), itv_starts, itv_ends,
                        itv_values, itv_offsets, scratch, to)
    shift_and_realign_tracks_sparse(out, out_offsets, regions, shifts, geno_offset_idx, geno_v_idxs,
                                    geno_offsets, v_starts, ilens, scratch, to, params, keep,
                                    keep_offsets, strategy_id, base_seed)
    if to_rc is not None:
        reverse_flat_rows_inplace(out, out_offsets, to_rc)


def build_splice_plan(lengths, splice_row_offsets, n_samples: int, n_rows: int) -> dict:
    """The reference's ``build_splice_plan`` (``_dataset/_splice.py:54-160``), restated with plain loops.

    ``lengths`` (B,) or (B, E): per-query lengths in (splice_row, sample, element) C-order, E inner
    cells per query (the ploidy); ``splice_row_offsets`` (n_rows * n_samples + 1): elements per
    (row, sample) pair.  The plan re-targets a ploidy-1 kernel call over the B * E flattened rows so
    that the bytes land in (row, sample, inner, element) C-order: ``permutation`` (new position ->
    old k = query * E + e), ``permuted_lengths``, ``permuted_out_offsets`` (per element) and
    ``group_offsets`` (one spliced sequence per (row, sample, inner) cell)."""
    lengths = np.asarray(lengths)
    off = np.asarray(splice_row_offsets, np.int64)
    n_pairs = int(n_rows) * int(n_samples)
    assert off.shape == (n_pairs + 1,)
    flat = lengths.reshape(lengths.shape[0], -1).astype(np.int32)
    B, E = flat.shape
    perm = []
    cells = [0]
    for p in range(n_pairs):                       # _splice.py:82-88: for pair, for e, for element
        s, e_ = int(off[p]), int(off[p + 1])
        for e in range(E):
            for q in range(s, e_):
                perm.append(q * E + e)
            cells.append(len(perm))
    perm = np.asarray(perm, np.intp)
    assert len(perm) == B * E
    permuted_lengths = flat.reshape(-1)[perm].astype(np.int32)
    out_offsets = np.zeros(len(perm) + 1, np.int64)
    np.cumsum(permuted_lengths, out=out_offsets[1:])
    group_offsets = out_offsets[np.asarray(cells, np.int64)]          # :137-151
    inner = tuple(lengths.shape[1:])
    return dict(permutation=perm, permuted_lengths=permuted_lengths, permuted_out_offsets=out_offsets,
                group_offsets=group_offsets, out_shape=(int(n_rows), int(n_samples), *inner, None))
