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


# ----------------------------------------------------------------------------- SVAR2 two-source provider (8 f4)
# The reference's entry points take (position, 32-bit key) channels and decode keys with the un-vendored crate
# `svar2-codec`.  What the reference states itself is what a decoded key IS (`decode_alt`, src/svar2/mod.rs:17-30);
# keys are therefore SYMBOLIC here -- ("inline", b"T"), ("pure_del", -2), ("lookup", row), the three constructors its
# tests use (encode_alt_inline / encode_pure_del / encode_lookup) -- and the channels cross every boundary DECODED:
# entry e has v_diff ilen[e] and the allele alt_bytes[alt_off[e] : alt_off[e + 1]] (empty = a pure deletion).
def decode_alt(key, lut_bytes=b"", lut_off=(0,)):
    """``decode_alt`` (src/svar2/mod.rs:17-30) over a symbolic key -> (v_diff, allele bytes)."""
    kind, val = key
    if kind == "inline":                      # DecodedKey::Inline { alt } -> (alt.len() - 1, alt)
        alt = bytes(val)
        return len(alt) - 1, alt
    if kind == "pure_del":                    # DecodedKey::PureDel { ilen } -> (ilen, empty)
        return int(val), b""
    if kind == "lookup":                      # DecodedKey::Lookup { row } -> LUT row
        s, e = int(lut_off[int(val)]), int(lut_off[int(val) + 1])
        alt = bytes(bytearray(lut_bytes)[s:e])
        return len(alt) - 1, alt
    raise ValueError(f"unknown key kind {kind!r}")


def decode_channels(vk_keys, dense_keys, lut_bytes=b"", lut_off=(0,)):
    """Decode both channels' symbolic keys into the flat form every SVAR2 entry point here takes:
    ``dict(vk_ilen, vk_alt_off, dense_ilen, dense_alt_off, alt_bytes)`` -- one shared allele pool, var_key alleles first."""
    pool = bytearray()
    out = {}
    for name, keys in (("vk", vk_keys), ("dense", dense_keys)):
        ilen = np.zeros(len(keys), np.int32)
        off = np.zeros(len(keys) + 1, np.int64)
        for i, k in enumerate(keys):
            d, alt = decode_alt(k, lut_bytes, lut_off)
            ilen[i] = d
            off[i] = len(pool)
            pool += alt
            off[i + 1] = len(pool)
        if len(keys) == 0:
            off[0] = len(pool)
        out[name + "_ilen"], out[name + "_alt_off"] = ilen, off
    out["alt_bytes"] = np.frombuffer(bytes(pool), np.uint8).copy()
    return out


def merge_hap(vk_pos, vk_lo, vk_hi, dense_pos, ds, de, dense_present, base_bit):
    """``merge_hap`` (src/svar2/mod.rs:45-72) -> (positions u32, sources): source >= 0 is var_key entry `source`,
    source < 0 is dense entry ``-(source + 1)``."""
    vk_pos, dense_pos = _c(vk_pos, np.int32), _c(dense_pos, np.int32)
    bits = _c(dense_present, np.uint8)
    cap = int(vk_hi - vk_lo) + int(de - ds)
    pos = np.zeros(max(cap, 1), np.uint32)
    src = np.zeros(max(cap, 1), np.int64)
    lib().gvlo_svar2_merge_hap.restype = C.c_int64
    n = lib().gvlo_svar2_merge_hap(_p(vk_pos), C.c_int64(int(vk_lo)), C.c_int64(int(vk_hi)), _p(dense_pos), C.c_int64(int(ds)),
                                   C.c_int64(int(de)), _p(bits), C.c_int64(int(base_bit)), _p(pos), _p(src))
    return pos[:n].copy(), src[:n].copy()


def _svar2_common(regions, shifts, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen, dense_range, dense_present,
                  dense_present_off):
    a = dict(reg=_c(regions, np.int32), sh=_c(shifts, np.int32), vp=_c(vk_pos, np.int32), vi=_c(vk_ilen, np.int32),
             vo=_c(vk_off, np.int64), dp=_c(dense_pos, np.int32), di=_c(dense_ilen, np.int32),
             dr=_c(np.asarray(dense_range).reshape(-1, 2), np.int32), pb=_c(dense_present, np.uint8),
             po=_c(dense_present_off, np.int64))
    assert a["reg"].ndim == 2 and a["reg"].shape[1] >= 3
    return a


def hap_diffs_svar2(regions, ploidy, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen, dense_range, dense_present,
                    dense_present_off, filter_exonic=False):
    """``hap_diffs_svar2`` (src/svar2/mod.rs:78-160) over decoded channels -> i32 (n_q, ploidy)."""
    a = _svar2_common(regions, None, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen, dense_range, dense_present,
                      dense_present_off)
    n_q = a["reg"].shape[0]
    diffs = np.zeros((n_q, int(ploidy)), np.int32)
    lib().gvlo_hap_diffs_svar2(_p(a["reg"]), C.c_int64(a["reg"].shape[1]), C.c_int64(n_q), C.c_int64(int(ploidy)), _p(a["vp"]),
                               _p(a["vi"]), _p(a["vo"]), _p(a["dp"]), _p(a["di"]), _p(a["dr"]), _p(a["pb"]), _p(a["po"]),
                               C.c_int32(1 if filter_exonic else 0), _p(diffs))
    return diffs


def reconstruct_haplotypes_from_svar2_into(
    out, out_bounds, regions, shifts, vk_pos, vk_ilen, vk_alt_off, vk_off, dense_pos, dense_ilen, dense_alt_off,
    dense_range, dense_present, dense_present_off, alt_bytes, ref_, ref_offsets, pad_char, parallel=False,
    filter_exonic=False,
):
    """The core ``reconstruct_haplotypes_from_svar2`` (src/reconstruct/mod.rs:619-826), in place: row k lands at
    ``out[out_bounds[k, 0] : out_bounds[k, 1]]`` (scatter write)."""
    assert out.dtype == np.uint8 and out.flags.c_contiguous
    a = _svar2_common(regions, shifts, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen, dense_range, dense_present,
                      dense_present_off)
    ob = _c(np.asarray(out_bounds).reshape(-1, 2), np.int64)
    n_q, ploidy = a["sh"].shape
    assert ob.shape[0] == n_q * ploidy
    vao, dao, ab = _c(vk_alt_off, np.int64), _c(dense_alt_off, np.int64), _c(alt_bytes, np.uint8)
    rf, ro = _c(ref_, np.uint8), _c(ref_offsets, np.int64)
    lib().gvlo_reconstruct_haplotypes_from_svar2(
        _p(out), _p(ob), _p(a["reg"]), C.c_int64(a["reg"].shape[1]), C.c_int64(n_q), C.c_int64(ploidy), _p(a["sh"]),
        _p(a["vp"]), _p(a["vi"]), _p(vao), _p(a["vo"]), _p(a["dp"]), _p(a["di"]), _p(dao), _p(a["dr"]), _p(a["pb"]),
        _p(a["po"]), _p(ab), _p(rf), _p(ro), C.c_uint8(int(pad_char)), C.c_int32(1 if filter_exonic else 0))


def reconstruct_haplotypes_from_svar2(
    regions, shifts, vk_pos, vk_ilen, vk_alt_off, vk_off, dense_pos, dense_ilen, dense_alt_off, dense_range,
    dense_present, dense_present_off, alt_bytes, ref_, ref_offsets, pad_char, output_length, parallel=False,
    filter_exonic=False,
):
    """The fused PyO3 entry (src/ffi/mod.rs:874-997) over decoded channels -> (out u8, out_offsets i64):
    ``output_length`` -1 = ragged (region length + diff), >= 0 = fixed."""
    regions = _c(regions, np.int32)
    shifts = _c(shifts, np.int32)
    n_q, ploidy = shifts.shape
    if output_length >= 0:
        lens = np.full(n_q * ploidy, int(output_length), np.int64)
    else:
        diffs = hap_diffs_svar2(regions, ploidy, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen, dense_range,
                                dense_present, dense_present_off, filter_exonic)
        ref_len = (regions[:, 2].astype(np.int64) - regions[:, 1].astype(np.int64))[:, None]
        lens = np.maximum(ref_len + diffs.astype(np.int64), 0).reshape(-1)
    off = np.zeros(n_q * ploidy + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    out = np.zeros(int(off[-1]), np.uint8)
    bounds = np.stack([off[:-1], off[1:]], axis=1)                  # bounds_from_offsets, reconstruct/mod.rs:830-838
    reconstruct_haplotypes_from_svar2_into(out, bounds, regions, shifts, vk_pos, vk_ilen, vk_alt_off, vk_off, dense_pos,
                                           dense_ilen, dense_alt_off, dense_range, dense_present, dense_present_off,
                                           alt_bytes, ref_, ref_offsets, pad_char, parallel, filter_exonic)
    return out, off


def shift_and_realign_tracks_from_svar2_into(
    out, out_offsets, regions, shifts, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen, dense_range, dense_present,
    dense_present_off, tracks, track_offsets, params, strategy_id=0, base_seed=0, query_seed=None, parallel=False,
):
    """The core ``shift_and_realign_tracks_from_svar2`` (src/tracks/mod.rs:705-860), in place."""
    assert out.dtype == np.float32 and out.flags.c_contiguous
    a = _svar2_common(regions, shifts, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen, dense_range, dense_present,
                      dense_present_off)
    oo = _c(out_offsets, np.int64)
    n_q, ploidy = a["sh"].shape
    tr, to, pr = _c(tracks, np.float32), _c(track_offsets, np.int64), _c(params, np.float64)
    qs = _c(query_seed, np.int64)
    lib().gvlo_realign_tracks_from_svar2(
        _p(out), _p(oo), _p(a["reg"]), C.c_int64(a["reg"].shape[1]), C.c_int64(n_q), C.c_int64(ploidy), _p(a["sh"]),
        _p(a["vp"]), _p(a["vi"]), _p(a["vo"]), _p(a["dp"]), _p(a["di"]), _p(a["dr"]), _p(a["pb"]), _p(a["po"]),
        _p(tr), _p(to), _p(pr), C.c_int64(int(strategy_id)), C.c_uint64(int(base_seed) & 0xFFFFFFFFFFFFFFFF), _p(qs))


def shift_and_realign_tracks_from_svar2(
    regions, shifts, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen, dense_range, dense_present, dense_present_off,
    tracks, track_offsets, params, strategy_id, base_seed, parallel=False,
):
    """The fused PyO3 entry (src/ffi/mod.rs:1835-1966) over decoded channels -> (out f32, out_offsets i64)."""
    regions = _c(regions, np.int32)
    shifts = _c(shifts, np.int32)
    n_q, ploidy = shifts.shape
    diffs = hap_diffs_svar2(regions, ploidy, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen, dense_range, dense_present,
                            dense_present_off, False)
    ref_len = (regions[:, 2].astype(np.int64) - regions[:, 1].astype(np.int64))[:, None]
    lens = np.maximum(ref_len + diffs.astype(np.int64), 0).reshape(-1)
    off = np.zeros(n_q * ploidy + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    out = np.zeros(int(off[-1]), np.float32)
    shift_and_realign_tracks_from_svar2_into(out, off, regions, shifts, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen,
                                             dense_range, dense_present, dense_present_off, tracks, track_offsets, params,
                                             strategy_id, base_seed, None, parallel)
    return out, off


def split_to_flat(n_regions, ploidy, vk, vk_off, dense_snp, dense_snp_range, dense_snp_present, dense_snp_present_off,
                  dense_indel, dense_indel_range, dense_indel_present, dense_indel_present_off):
    """``split_to_flat`` (src/svar2/mod.rs:176-274), plain loops: genoray's per-class split dense channels -> the flat
    single-dense-channel layout.  ``vk`` / ``dense_*``: lists of (position, key); ranges: lists of (start, end).  Per query
    the window is snp entries then indel entries; per haplotype the presence bits are snp bits then indel bits, LSB-first."""
    def get_bit(bits, i):
        return (bits[i // 8] >> (i % 8)) & 1

    out = dict(vk_pos=[p for p, _ in vk], vk_key=[k for _, k in vk], vk_off=[int(o) for o in vk_off],
               dense_pos=[], dense_key=[], dense_range=[])
    for q in range(n_regions):
        base = len(out["dense_pos"])
        for j in range(*dense_snp_range[q]):
            out["dense_pos"].append(dense_snp[j][0]); out["dense_key"].append(dense_snp[j][1])
        for j in range(*dense_indel_range[q]):
            out["dense_pos"].append(dense_indel[j][0]); out["dense_key"].append(dense_indel[j][1])
        out["dense_range"] += [base, len(out["dense_pos"])]
    total_bits = sum(((dense_snp_range[q][1] - dense_snp_range[q][0]) + (dense_indel_range[q][1] - dense_indel_range[q][0])) * ploidy
                     for q in range(n_regions))
    present = [0] * ((total_bits + 7) // 8)
    off = [0]
    acc = 0
    h = 0
    for q in range(n_regions):
        (ss, se), (is_, ie) = dense_snp_range[q], dense_indel_range[q]
        for _ in range(ploidy):
            for k in range(se - ss):
                if get_bit(dense_snp_present, dense_snp_present_off[h] + k):
                    present[acc // 8] |= 1 << (acc % 8)
                acc += 1
            for k in range(ie - is_):
                if get_bit(dense_indel_present, dense_indel_present_off[h] + k):
                    present[acc // 8] |= 1 << (acc % 8)
                acc += 1
            off.append(acc)
            h += 1
    out["dense_present"] = present
    out["dense_present_off"] = off
    return out
