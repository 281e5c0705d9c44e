"""Drop-in mirror of the reference's FFI entry points for the haplotype path.

Same names, argument order, argument meaning and return shapes as the PyO3
functions the reference's callers import by name (``_haps.py:38-43``,
``_genotypes.py:5-9``, ``_reference.py:26``):

===========================================  ===================================
here                                          reference
===========================================  ===================================
``reconstruct_haplotypes_fused``              ``src/ffi/mod.rs:722-860``
``reconstruct_haplotypes_from_sparse``        ``src/ffi/mod.rs:632-700`` (in place)
``reconstruct_haplotypes_spliced_fused``      ``src/ffi/mod.rs:1981-2076``
``reconstruct_annotated_haplotypes_fused``    ``src/ffi/mod.rs:2237-2397``
``get_diffs_sparse``                          ``src/ffi/mod.rs:143-185``
``get_reference``                             ``src/ffi/mod.rs:2401-2429``
``intervals_to_tracks``                       ``src/ffi/mod.rs:188-240``
``shift_and_realign_tracks_sparse``           ``src/tracks/mod.rs:495-667`` (in place)
``intervals_and_realign_track_fused``         ``src/ffi/mod.rs:2551-2672`` (in place)
``reconstruct_haplotypes_fused_onehot``       new: fused one-hot (no counterpart;
                                              replaces the user-side ``sp.DNA.ohe``)
===========================================  ===================================

numpy in -> numpy out (so a caller that swaps its import keeps working); the
per-dataset arrays (reference, variant table, genotype CSR) are uploaded to HBM
once and cached keyed by their host buffers, the way ``_HapsFfiStatic`` caches
contiguous views on the host (``_haps.py:329-348``).  ``parallel`` is accepted and
ignored (``_threads.py:122-127`` gates rayon; the GPU path is always parallel).
Callers that want to stay on the device use :class:`genvarloader_amd.HapsDevice`.

Errors: bad dtype/shape -> ``ValueError`` (the reference panics,
``ffi/mod.rs:56-76``); HIP failures -> ``GvlError``; no CPU fallback.
"""

from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from . import device as _device
from .device import HapsDevice, _starts_stops

_STATIC_CACHE: "OrderedDict[tuple, HapsDevice]" = OrderedDict()
_STATIC_CACHE_MAX = 4


_FP_CACHE: dict = {}       # id(array) -> (weakref, key): fingerprints of READ-ONLY arrays, taken once per object


def _fingerprint(a: np.ndarray) -> int:
    import zlib

    flat = a.reshape(-1)
    if flat.size == 0:
        return 0
    if a.nbytes <= (1 << 20):
        return zlib.adler32(np.ascontiguousarray(flat).view(np.uint8))
    step = max(1, flat.size // 512)
    fp = zlib.adler32(np.ascontiguousarray(flat[::step]).view(np.uint8))
    fp = zlib.adler32(np.ascontiguousarray(flat[:64]).view(np.uint8), fp)
    return zlib.adler32(np.ascontiguousarray(flat[-64:]).view(np.uint8), fp)


def _key(a):
    """Cache key of a per-dataset host array: address, shape, dtype AND a content fingerprint, so
    that an in-place edit of a cached array is a miss (re-upload) and not silently ignored.

    A READ-ONLY array -- what the reference hands over: ``np.memmap(..., mode="r")`` of the dataset's files,
    ``_haps.py:435-460`` -- cannot be edited through that object: it is fingerprinted ONCE (remembered per
    object, by id + weakref), so a call does not touch a genome-scale memmap at all.  A writable array is
    fingerprinted on every call: exactly below 1 MiB, a strided sample of 512 elements (plus both ends)
    above -- the arrays are per-dataset constants by contract (``_HapsFfiStatic``, _haps.py:233-247), this
    only guards against accidents; ``clear_static_cache()`` forces a re-upload."""
    import weakref

    a = np.asanyarray(a)      # (asanyarray: an np.memmap stays THAT object -- np.asarray makes a fresh base-class view per call)
    if not a.flags.writeable and not _writable_ancestor(a):
        hit = _FP_CACHE.get(id(a))
        if hit is not None and hit[0]() is a:
            return hit[1]
        key = (a.__array_interface__["data"][0], a.shape, a.dtype.str, _fingerprint(a))
        try:
            i = id(a)
            _FP_CACHE[i] = (weakref.ref(a, lambda _r, i=i: _FP_CACHE.pop(i, None)), key)
        except TypeError:
            pass
        return key
    return (a.__array_interface__["data"][0], a.shape, a.dtype.str, _fingerprint(a))


def _writable_ancestor(a) -> bool:
    """A read-only VIEW over a writable array can still change under it (through the base): only an array none of whose
    ancestors is writable is fingerprinted once.  (A read-only np.memmap's base is the mmap object itself: fine.)"""
    b = getattr(a, "base", None)
    while b is not None:
        if isinstance(b, np.ndarray):
            if b.flags.writeable:
                return True
            b = b.base
        else:
            return False
    return False


def _req(a, dt, name, ndim=None):
    """The reference asserts exact dtype + C-contiguity for the big arrays
    (``_ffi_array``, _dataset/_utils.py:13-34); small ones are coerced."""
    a = np.ascontiguousarray(a, dtype=dt)
    if ndim is not None and a.ndim != ndim:
        raise ValueError(f"`{name}` must be {ndim}-D")
    return a


def _static(geno_offsets, geno_v_idxs, v_starts, ilens, alt_alleles, alt_offsets, ref_, ref_offsets,
            pad_char, device="cuda") -> HapsDevice:
    # np.asarray is the identity for ndarrays; anything else becomes an array we keep
    # alive in the cache entry, so an address-based key can never alias a dead buffer
    arrs = tuple(np.asanyarray(a) for a in (geno_offsets, geno_v_idxs, v_starts, ilens, alt_alleles,
                                            alt_offsets, ref_, ref_offsets))
    geno_offsets, geno_v_idxs, v_starts, ilens, alt_alleles, alt_offsets, ref_, ref_offsets = arrs
    key = tuple(_key(a) for a in arrs) + (int(pad_char), str(device))
    dev = _STATIC_CACHE.get(key)
    if dev is None:
        dev = HapsDevice(
            ref=_req(ref_, np.uint8, "ref_", 1), ref_offsets=_req(ref_offsets, np.int64, "ref_offsets", 1),
            v_starts=_req(v_starts, np.int32, "v_starts", 1), ilens=_req(ilens, np.int32, "ilens", 1),
            alt_alleles=_req(alt_alleles, np.uint8, "alt_alleles", 1),
            alt_offsets=_req(alt_offsets, np.int64, "alt_offsets", 1),
            geno_offsets=_starts_stops(geno_offsets), geno_v_idxs=_req(geno_v_idxs, np.int32, "geno_v_idxs", 1),
            pad_char=int(pad_char), device=device)
        # keep the host arrays alive so that the address-based key stays valid
        dev._host_refs = arrs
        _STATIC_CACHE[key] = dev
        while len(_STATIC_CACHE) > _STATIC_CACHE_MAX:
            _STATIC_CACHE.popitem(last=False)
    else:
        _STATIC_CACHE.move_to_end(key)
    return dev


def clear_static_cache() -> None:
    for c in (_STATIC_CACHE, _DIFF_CACHE, _REF_CACHE, _TRACK_CACHE, _ITV_CACHE):
        c.clear()


def _np(t):
    """Device tensor -> numpy.  Large results go through torch's caching pinned-host allocator:
    a fresh pageable array costs a page fault per 4 KiB (32 MiB of one-hot: 5.3 ms), a recycled
    pinned block is one DMA at PCIe speed."""
    if t is None:
        return None
    try:
        return _np_copy(t)
    finally:
        _lib.check_async()       # (the copy synchronised: anything a launch reported is visible now)


def _np_copy(t):
    if t.is_cuda and t.numel() * t.element_size() >= (1 << 20):
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        host.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return host.numpy()
    return t.cpu().numpy()


def reconstruct_haplotypes_fused(
    regions, shifts, geno_offset_idx, geno_offsets, geno_v_idxs, v_starts, ilens, alt_alleles,
    alt_offsets, ref_, ref_offsets, pad_char, output_length, keep=None, keep_offsets=None,
    to_rc=None, parallel=False,
):
    """-> (out_data u8[total], out_offsets i64[K+1])   (ffi/mod.rs:743)."""
    dev = _static(geno_offsets, geno_v_idxs, v_starts, ilens, alt_alleles, alt_offsets, ref_,
                  ref_offsets, pad_char)
    out = dev.reconstruct(regions, shifts, geno_offset_idx, int(output_length), keep, keep_offsets, to_rc)
    return _np(out.haps), _np(out.out_offsets)


def reconstruct_haplotypes_fused_onehot(
    regions, shifts, geno_offset_idx, geno_offsets, geno_v_idxs, v_starts, ilens, alt_alleles,
    alt_offsets, ref_, ref_offsets, pad_char, output_length, keep=None, keep_offsets=None,
    to_rc=None, parallel=False, *, layout="lc", return_haps=False,
):
    """Fused one-hot variant: -> (onehot u8 (total, 4) | (K, 4, L), out_offsets[, haps])."""
    dev = _static(geno_offsets, geno_v_idxs, v_starts, ilens, alt_alleles, alt_offsets, ref_,
                  ref_offsets, pad_char)
    out = dev.reconstruct(regions, shifts, geno_offset_idx, int(output_length), keep, keep_offsets, to_rc,
                          haps=return_haps, onehot=True, layout=layout)
    if return_haps:
        return _np(out.onehot), _np(out.out_offsets), _np(out.haps)
    return _np(out.onehot), _np(out.out_offsets)


def reconstruct_haplotypes_from_sparse(
    out, out_offsets, regions, shifts, geno_offset_idx, geno_offsets, geno_v_idxs, v_starts, ilens,
    alt_alleles, alt_offsets, ref, ref_offsets, pad_char, keep=None, keep_offsets=None,
    annot_v_idxs=None, annot_ref_pos=None, parallel=False,
):
    """In place (ffi/mod.rs:634-655): writes `out` (and the annotation buffers)."""
    if not (isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.flags.c_contiguous):
        raise ValueError("`out` must be a C-contiguous uint8 array")
    dev = _static(geno_offsets, geno_v_idxs, v_starts, ilens, alt_alleles, alt_offsets, ref,
                  ref_offsets, pad_char)
    annotate = annot_v_idxs is not None or annot_ref_pos is not None
    oo = _req(out_offsets, np.int64, "out_offsets", 1)
    if len(oo) and int(oo[-1]) != out.size:
        raise ValueError("out_offsets[-1] must equal len(out)")
    res = dev.reconstruct(regions, shifts, geno_offset_idx, -1, keep, keep_offsets, None,
                          out_offsets=oo, annotate=annotate)
    out[...] = _np(res.haps)
    if annot_v_idxs is not None:
        annot_v_idxs[...] = _np(res.annot_v_idxs)
    if annot_ref_pos is not None:
        annot_ref_pos[...] = _np(res.annot_ref_pos)


def reconstruct_haplotypes_spliced_fused(
    permuted_regions, flat_shifts, flat_geno_offset_idx, out_offsets, geno_offsets, geno_v_idxs,
    v_starts, ilens, alt_alleles, alt_offsets, ref_, ref_offsets, pad_char, keep=None,
    keep_offsets=None, to_rc=None, parallel=False,
):
    """Caller-supplied (permuted) out_offsets, ploidy-1 rows (ffi/mod.rs:1983-2002) -> u8[total]."""
    dev = _static(geno_offsets, geno_v_idxs, v_starts, ilens, alt_alleles, alt_offsets, ref_,
                  ref_offsets, pad_char)
    res = dev.reconstruct(permuted_regions, flat_shifts, flat_geno_offset_idx, -1, keep, keep_offsets,
                          to_rc, out_offsets=_req(out_offsets, np.int64, "out_offsets", 1))
    return _np(res.haps)


def reconstruct_annotated_haplotypes_fused(
    regions, shifts, geno_offset_idx, geno_offsets, geno_v_idxs, v_starts, ilens, alt_alleles,
    alt_offsets, ref_, ref_offsets, pad_char, output_length, keep=None, keep_offsets=None,
    to_rc=None, parallel=False,
):
    """-> (out_data, annot_v_idxs, annot_ref_pos, out_offsets)   (ffi/mod.rs:2237-2397)."""
    dev = _static(geno_offsets, geno_v_idxs, v_starts, ilens, alt_alleles, alt_offsets, ref_,
                  ref_offsets, pad_char)
    out = dev.reconstruct(regions, shifts, geno_offset_idx, int(output_length), keep, keep_offsets, to_rc,
                          annotate=True)
    return _np(out.haps), _np(out.annot_v_idxs), _np(out.annot_ref_pos), _np(out.out_offsets)


def get_diffs_sparse(geno_offset_idx, geno_v_idxs, geno_offsets, ilens, keep=None, keep_offsets=None,
                     q_starts=None, q_ends=None, v_starts=None, parallel=False):
    """-> i32 (B, P)   (ffi/mod.rs:145-157).  Query mode iff q_starts, q_ends, v_starts given."""
    ilens = _req(ilens, np.int32, "ilens", 1)
    n_var = len(ilens)
    has_query = q_starts is not None and q_ends is not None and v_starts is not None
    vs = _req(v_starts, np.int32, "v_starts", 1) if has_query else None
    # diffs only read the CSR + ilens/v_starts: a stub reference/allele table is enough
    dev = _diffs_static(geno_offsets, geno_v_idxs, vs, ilens)
    d = dev.get_diffs_sparse(geno_offset_idx, keep, keep_offsets,
                             q_starts if has_query else None, q_ends if has_query else None)
    return _np(d)


_DIFF_CACHE: "OrderedDict[tuple, HapsDevice]" = OrderedDict()


def choose_exonic_variants(starts, ends, geno_offset_idx, geno_v_idxs, geno_offsets, v_starts, ilens):
    """-> (keep bool[total], keep_offsets i64[K+1])   (src/genotypes/mod.rs:127-176; imported next to
    get_diffs_sparse at _dataset/_genotypes.py:5-9)."""
    dev = _diffs_static(geno_offsets, geno_v_idxs, v_starts, ilens)
    keep, ko = dev.choose_exonic_variants(_req(starts, np.int32, "starts", 1), _req(ends, np.int32, "ends", 1),
                                          _req(geno_offset_idx, np.int64, "geno_offset_idx", 2))
    return _np(keep).astype(np.bool_), _np(ko)


def rc_alleles(byte_data, seq_offsets, var_offsets, to_rc_row):
    """In place: reverse-complement the alleles of the mask-selected rows (src/ffi/mod.rs:2805-2820 ->
    variants::rc_alleles_inplace): row r owns alleles var_offsets[r]..var_offsets[r+1], allele a owns
    bytes seq_offsets[a]..seq_offsets[a+1]; the same rc_row as the haplotype path (gvl_rc_rows)."""
    seq_offsets = _req(seq_offsets, np.int64, "seq_offsets", 1)
    var_offsets = _req(var_offsets, np.int64, "var_offsets", 1)
    per_allele = np.repeat(np.asarray(to_rc_row, np.bool_), np.diff(var_offsets))
    if per_allele.size == 0 or byte_data.size == 0:
        return
    t = torch.from_numpy(np.ascontiguousarray(byte_data, np.uint8)).cuda()
    _device.rc_flat_rows_inplace(t, seq_offsets, per_allele)
    byte_data[...] = t.cpu().numpy()


def _diffs_static(geno_offsets, geno_v_idxs, v_starts, ilens) -> HapsDevice:
    """``v_starts`` None (plain / keep mode): positions are not read, and not part of the key."""
    arrs = tuple(np.asarray(a) for a in (geno_offsets, geno_v_idxs, ilens) + (() if v_starts is None else (v_starts,)))
    geno_offsets, geno_v_idxs, ilens = arrs[:3]
    key = tuple(_key(a) for a in arrs) + (v_starts is None,)
    dev = _DIFF_CACHE.get(key)
    if dev is None:
        # a full dataset that was uploaded with the same genotype CSR + variant table already has
        # everything these entry points read: one HBM copy instead of two
        for full in _STATIC_CACHE.values():
            h = full._host_refs          # (geno_offsets, geno_v_idxs, v_starts, ilens, ...)
            if (_key(h[0]), _key(h[1]), _key(h[3])) == key[:3] and (v_starts is None or _key(h[2]) == key[3]):
                _DIFF_CACHE[key] = dev = full          # (alias: later calls hit directly)
                break
    if dev is None:
        n = len(ilens)
        dev = HapsDevice(ref=np.zeros(1, np.uint8), ref_offsets=np.array([0, 1], np.int64),
                         v_starts=np.zeros(n, np.int32) if v_starts is None else arrs[3], ilens=ilens,
                         alt_alleles=np.zeros(1, np.uint8),
                         alt_offsets=np.zeros(n + 1, np.int64), geno_offsets=_starts_stops(geno_offsets),
                         geno_v_idxs=_req(geno_v_idxs, np.int32, "geno_v_idxs", 1), slot_records=False)
        dev._host_refs = arrs
        _DIFF_CACHE[key] = dev
        while len(_DIFF_CACHE) > _STATIC_CACHE_MAX:
            _DIFF_CACHE.popitem(last=False)
    return dev


def get_reference(regions, out_offsets, reference, ref_offsets, pad_char, parallel=False, to_rc=None):
    """-> u8[total]   (ffi/mod.rs:2402-2411)."""
    dev = _ref_static(reference, ref_offsets, pad_char)
    return _np(dev.get_reference(regions, _req(out_offsets, np.int64, "out_offsets", 1), to_rc))


_REF_CACHE: "OrderedDict[tuple, HapsDevice]" = OrderedDict()


def _ref_static(reference, ref_offsets, pad_char) -> HapsDevice:
    reference, ref_offsets = np.asarray(reference), np.asarray(ref_offsets)
    key = (_key(reference), _key(ref_offsets), int(pad_char))
    dev = _REF_CACHE.get(key)
    if dev is None:
        dev = HapsDevice(ref=_req(reference, np.uint8, "reference", 1),
                         ref_offsets=_req(ref_offsets, np.int64, "ref_offsets", 1),
                         v_starts=np.zeros(0, np.int32), ilens=np.zeros(0, np.int32),
                         alt_alleles=np.zeros(0, np.uint8), alt_offsets=np.zeros(1, np.int64),
                         geno_offsets=np.zeros((2, 1), np.int64), geno_v_idxs=np.zeros(0, np.int32),
                         pad_char=int(pad_char))
        dev._host_refs = (reference, ref_offsets)
        _REF_CACHE[key] = dev
        while len(_REF_CACHE) > _STATIC_CACHE_MAX:
            _REF_CACHE.popitem(last=False)
    return dev


# ------------------------------------------------------------------------------- tracks (a12)
_TRACK_CACHE: "OrderedDict[tuple, HapsDevice]" = OrderedDict()


def _track_static(geno_offsets, geno_v_idxs, v_starts, ilens) -> HapsDevice:
    """Realignment reads only the genotype CSR + v_starts / ilens: the same device arrays as the
    length-delta entry points in query mode (one HBM copy, shared with a full dataset when there is one)."""
    return _diffs_static(geno_offsets, geno_v_idxs, v_starts, ilens)


_ITV_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()


def _itv_static(itv_starts, itv_ends, itv_values, itv_offsets, device="cuda"):
    """The interval arrays are per dataset (the reference memmaps them): upload once, with the
    running maxima of the ends (gvl_intervals_prefix_max), keyed by the host buffers."""
    arrs = tuple(np.asarray(a) for a in (itv_starts, itv_ends, itv_values, itv_offsets))
    key = tuple(_key(a) for a in arrs) + (str(device),)
    ent = _ITV_CACHE.get(key)
    if ent is None:
        d = torch.device(device)
        a = torch.from_numpy(_req(arrs[0], np.int32, "itv_starts", 1)).to(d)
        b = torch.from_numpy(_req(arrs[1], np.int32, "itv_ends", 1)).to(d)
        v = torch.from_numpy(_req(arrs[2], np.float32, "itv_values", 1)).to(d)
        io = torch.from_numpy(_req(arrs[3], np.int64, "itv_offsets", 1)).to(d)
        pm = _device.intervals_prefix_max(b, io, device) if b.numel() else None
        # ... and the painter's coarse bucket index + the gvl_track_set that carries both (gvl_paint_tracks: the tiled + bitmap
        # path; the plain gvl_intervals_to_tracks has no place for the index and paints at 0.17 of the HBM peak)
        bk = _device.intervals_bucket_index(a, pm, io, device) if pm is not None else None
        ts, keep = _device.make_track_set(a, b, v, io, pm, bk, device) if pm is not None else (None, None)
        if ts is not None:
            # BigWig-like lists (no overlaps, distinct starts, <= 256 intervals in two adjacent buckets: checked once, here): the
            # tiled kernel finishes every chunk and the painter needs no second ("leftovers") launch
            from .loader import DeviceHapsTracksDataset

            ts.tile_complete = 1 if DeviceHapsTracksDataset._tiles_complete(a, b, io, bk) else 0
        ent = (a, b, v, io, pm, ts, (arrs, keep, bk))           # arrs: keep the host buffers alive (address key)
        _ITV_CACHE[key] = ent
        while len(_ITV_CACHE) > _STATIC_CACHE_MAX:
            _ITV_CACHE.popitem(last=False)
    else:
        _ITV_CACHE.move_to_end(key)
    return ent[:6]


def intervals_to_tracks(offset_idxs, starts, itv_starts, itv_ends, itv_values, itv_offsets, out, out_offsets,
                        parallel=False):
    """In place: paints `out` (src/ffi/mod.rs:188-240)."""
    a, b, v, io, pm, ts = _itv_static(itv_starts, itv_ends, itv_values, itv_offsets)
    res = _device.intervals_to_tracks(offset_idxs, starts, a, b, v, io, _req(out_offsets, np.int64, "out_offsets", 1),
                                      itv_pmax_ends=pm, track_set=ts)
    out[...] = _np(res)
    _lib.check_async()       # (a tile_complete set whose chunk the tiled kernel could not finish is reported, never returned half painted)


def shift_and_realign_tracks_sparse(out, out_offsets, regions, shifts, geno_offset_idx, geno_v_idxs, geno_offsets,
                                    v_starts, ilens, tracks, track_offsets, params, keep=None, keep_offsets=None,
                                    strategy_id=0, base_seed=0, parallel=False):
    """In place: argument order of the reference's wrapper (_tracks.py:42-60)."""
    dev = _track_static(geno_offsets, geno_v_idxs, v_starts, ilens)
    res = _device.realign_tracks(dev, regions, shifts, geno_offset_idx, _req(out_offsets, np.int64, "out_offsets", 1),
                                 tracks, track_offsets, params, strategy_id, base_seed, keep, keep_offsets)
    out[...] = _np(res)


def intervals_and_realign_track_fused(out, out_offsets, regions, shifts, geno_offset_idx, geno_v_idxs, geno_offsets,
                                      v_starts, ilens, offset_idxs, itv_starts, itv_ends, itv_values, itv_offsets,
                                      track_offsets, params, strategy_id, base_seed, keep=None, keep_offsets=None,
                                      to_rc=None, parallel=False):
    """In place: paint -> realign -> reverse negative-strand rows (src/ffi/mod.rs:2551-2672).
    The scratch track never leaves the device."""
    dev = _track_static(geno_offsets, geno_v_idxs, v_starts, ilens)
    regions = _req(regions, np.int32, "regions", 2)
    a, b, v, io, pm, ts = _itv_static(itv_starts, itv_ends, itv_values, itv_offsets)
    scratch = _device.intervals_to_tracks(offset_idxs, np.ascontiguousarray(regions[:, 1]), a, b, v, io,
                                          _req(track_offsets, np.int64, "track_offsets", 1), itv_pmax_ends=pm, track_set=ts)
    res = _device.realign_tracks(dev, regions, shifts, geno_offset_idx, _req(out_offsets, np.int64, "out_offsets", 1),
                                 scratch, track_offsets, params, strategy_id, base_seed, keep, keep_offsets, to_rc)
    out[...] = _np(res)
    _lib.check_async()
