"""ctypes binding of ``libgvl_hip.so`` (the C-ABI declared in ``include/gvl_hip.h``).

This is the stub a GenVarLoader maintainer would add next to the PyO3 module
(``/root/reference/python/genvarloader/_dataset/_haps.py:38-43`` imports the FFI
symbols by name; INTEGRATION.md shows the swap).  There is no CPU fallback: if
the HIP library is missing or fails to load, importing the compute entry points
raises.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_NAME = "libgvl_hip.so"

# every symbol include/gvl_hip.h declares (tests check the library exports them all)
SYMBOLS = (
    "gvl_abi_version",
    "gvl_set_debug_flags",
    "gvl_set_tuning",
    "gvl_last_error",
    "gvl_async_error",
    "gvl_static_upload",
    "gvl_static_free",
    "gvl_pack_variants",
    "gvl_pack_genotypes",
    "gvl_pack_slots",
    "gvl_pack_slot_vidx",
    "gvl_ref4_bytes",
    "gvl_pack_reference",
    "gvl_reconstruct",
    "gvl_reconstruct_many",
    "gvl_hap_plan_bytes",
    "gvl_hap_plan",
    "gvl_get_diffs_sparse",
    "gvl_hap_offsets",
    "gvl_paint_tracks",
    "gvl_get_reference",
    "gvl_get_reference_many",
    "gvl_keep_offsets",
    "gvl_choose_exonic_variants",
    "gvl_rc_rows",
    "gvl_rc_bounded_rows",
    "gvl_reverse_rows_4",
    "gvl_onehot",
    "gvl_intervals_prefix_max",
    "gvl_intervals_bucket_counts",
    "gvl_intervals_bucket_fill",
    "gvl_intervals_to_tracks",
    "gvl_realign_tracks",
    "gvl_tracks_scratch_bytes",
    "gvl_tracks_batch",
    "gvl_prepare_request",
    "gvl_loader_slot_bytes",
    "gvl_loader_table_bytes",
    "gvl_loader_create",
    "gvl_loader_set_epoch",
    "gvl_loader_prefetch_epoch",
    "gvl_loader_start_epoch",
    "gvl_loader_next",
    "gvl_loader_destroy",
    "gvl_svar2_workspace_bytes",
    "gvl_svar2_merge",
    "gvl_svar2_reconstruct",
)

ABI_VERSION = 11         # include/gvl_hip.h: GVL_ABI_VERSION
TUNE_PIPE_ROWS_X100, TUNE_PIPE_MIN_ROWS, TUNE_LEAN_SUB, TUNE_TRACK_PLAN_MAX_MB, TUNE_RAGGED_SIZING, TUNE_HAP_PLAN_MAX_MB = 0, 1, 2, 3, 4, 5     # GVL_TUNE_*
TUNE_MIXED_MIN_ROWS = 6
GVL_ONEHOT_LC = 0
GVL_ONEHOT_CL = 1

_vp = C.c_void_p
_i64 = C.c_int64


class GvlStatic(C.Structure):
    _fields_ = [
        ("ref", _vp), ("ref_len", _i64), ("ref_offsets", _vp), ("n_contigs", _i64),
        ("v_starts", _vp), ("ilens", _vp), ("alt_offsets", _vp), ("alt_alleles", _vp),
        ("n_variants", _i64), ("alt_len", _i64), ("vrec", _vp),
        ("geno_o_starts", _vp), ("geno_o_stops", _vp), ("n_geno_offsets", _i64),
        ("geno_v_idxs", _vp), ("n_geno", _i64), ("pad_char", C.c_uint8), ("geno_rec", _vp),
        ("slot_rec", _vp), ("ref4", _vp), ("slot_vidx", _vp),
    ]


class GvlBatch(C.Structure):
    _fields_ = [
        ("regions", _vp), ("regions_stride", _i64), ("shifts", _vp), ("geno_offset_idx", _vp),
        ("batch", _i64), ("ploidy", _i64), ("keep", _vp), ("keep_offsets", _vp), ("to_rc", _vp),
        ("output_length", _i64), ("out_offsets", _vp), ("max_row_len", _i64), ("hap_plan", _vp),
        ("out_bounds", _vp), ("total_len_hint", _i64), ("query_seed", _vp),
    ]


class GvlOut(C.Structure):
    _fields_ = [
        ("haps", _vp), ("onehot", _vp), ("onehot_layout", C.c_int32),
        ("annot_v_idxs", _vp), ("annot_ref_pos", _vp), ("out_offsets", _vp),
    ]


class GvlRefBatch(C.Structure):
    """``gvl_ref_batch``: one batch of ``gvl_get_reference_many``."""
    _fields_ = [
        ("regions", _vp), ("regions_stride", _i64), ("n_rows", _i64), ("out_offsets", _vp), ("max_row_len", _i64),
        ("to_rc", _vp), ("out", _vp), ("onehot", _vp),
    ]


class GvlSvar2Batch(C.Structure):
    """``gvl_svar2_batch``: one batch of DECODED SVAR2 channels (device pointers)."""
    _fields_ = [
        ("vk_pos", _vp), ("vk_ilen", _vp), ("vk_alt_off", _vp), ("vk_off", _vp), ("n_vk", _i64),
        ("dense_pos", _vp), ("dense_ilen", _vp), ("dense_alt_off", _vp), ("n_dense", _i64),
        ("dense_range", _vp), ("dense_present", _vp), ("dense_present_bits", _i64), ("dense_present_off", _vp),
        ("alt_bytes", _vp), ("alt_len", _i64), ("filter_exonic", C.c_int32),
    ]


class GvlTrackSet(C.Structure):
    _fields_ = [
        ("itv_starts", _vp), ("itv_ends", _vp), ("itv_values", _vp), ("itv_offsets", _vp),
        ("n_intervals", _i64), ("itv_pmax_ends", _vp),
        ("bkt_offsets", _vp), ("bkt_base", _vp), ("bkt_lo", _vp), ("bkt_hi", _vp), ("tile_complete", C.c_int32),
        ("has_fill", C.c_int32), ("fill_strategy", C.c_int32), ("fill_param", C.c_double), ("list_div", _i64),
    ]


class GvlLoaderConfig(C.Structure):
    _fields_ = [
        ("full_regions", _vp), ("n_regions", _i64), ("n_samples", _i64), ("ploidy", _i64),
        ("batch_size", _i64), ("output_length", _i64), ("jitter", _i64),
        ("rc_neg", C.c_int32), ("deterministic", C.c_int32), ("seed", C.c_uint64),
        ("want_haps", C.c_int32), ("want_onehot", C.c_int32), ("onehot_layout", C.c_int32),
        ("in_flight", C.c_int32), ("n_slots", C.c_int32), ("slot_arenas", C.POINTER(_vp)),
        ("threaded", C.c_int32), ("group", C.c_int32),
        ("want_annot", C.c_int32), ("max_row_len", _i64), ("tracks", _vp), ("n_tracks", C.c_int32),
        ("track_seed_mode", C.c_int32), ("strategy_id", _i64), ("track_param", C.c_double), ("track_seed", C.c_uint64),
        ("scratch_stride", _i64),
    ]


LOADER_SLOT_PARTS = 12       # GVL_LOADER_SLOT_PARTS
LOADER_TABLE_PARTS = 10       # GVL_LOADER_TABLE_PARTS


class GvlLoaderBatch(C.Structure):
    _fields_ = [
        ("slot", C.c_int32), ("batch", _i64), ("idx", _vp), ("onehot", _vp), ("haps", _vp),
        ("regions", _vp), ("geno_offset_idx", _vp), ("shifts", _vp), ("to_rc", _vp), ("out_offsets", _vp),
        ("annot_v_idxs", _vp), ("annot_ref_pos", _vp), ("tracks", _vp), ("sizes", _vp), ("track_seed", _vp),
    ]


class GvlError(RuntimeError):
    pass


_LIB = None


def lib_path() -> Path:
    return Path(os.environ.get("GVL_HIP_LIB", _HERE / LIB_NAME))


def _check_fresh(p: Path) -> None:
    """The in-tree library must have been built from the sources next to it (``__graft_entry__.build_hip`` leaves their
    content hash in ``libgvl_hip.so.content``).  Only checked for the in-tree library with its sources present."""
    import hashlib

    if "GVL_HIP_LIB" in os.environ or os.environ.get("GVL_ALLOW_STALE_LIB"):
        return
    stamp = p.with_suffix(".so.content")
    # (the same set, in the same order, as __graft_entry__.hip_sources())
    srcs = sorted((_HERE / "csrc").glob("*.hip")) + sorted((_HERE / "csrc").glob("*.inc")) + [_HERE.parent / "include" / "gvl_hip.h"]
    if len(srcs) < 2 or not all(f.exists() for f in srcs):
        return
    if not stamp.exists():
        # an in-tree library next to its sources but without its stamp (e.g. a snapshot that shipped the .so alone): nothing says
        # what it was built from -- the situation this check exists for
        raise GvlError(f"{p} has no {stamp.name} next to it: cannot tell whether it was built from the sources in csrc/ -- run "
                       "`python -c 'import __graft_entry__ as g; g.build()'` (GVL_ALLOW_STALE_LIB=1 overrides)")
    h = hashlib.sha256()
    for f in srcs:
        h.update(f.read_bytes())
    if stamp.read_text().strip() != h.hexdigest():
        raise GvlError(f"{p} was not built from the sources next to it (csrc/*.hip, csrc/*.inc, include/gvl_hip.h changed "
                       "since): run `python -c 'import __graft_entry__ as g; g.build()'` (GVL_ALLOW_STALE_LIB=1 overrides)")


def load() -> C.CDLL:
    """Load the HIP library or raise -- never falls back to a CPU path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # libgvl_hip.so links libamdhip64; PyTorch-ROCm bundles its own copy of that runtime.
    # Import torch first so that the process has ONE HIP runtime (torch's) and device
    # pointers / streams handed across the C-ABI belong to it.
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - the C-ABI itself does not need torch
        pass
    p = lib_path()
    if not p.exists():
        raise GvlError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  genvarloader_amd has no CPU fallback."
        )
    _check_fresh(p)
    try:
        lib = C.CDLL(str(p))
    except OSError as e:  # pragma: no cover
        raise GvlError(f"failed to load {p}: {e}") from e
    lib.gvl_last_error.restype = C.c_char_p
    lib.gvl_abi_version.restype = C.c_int
    if lib.gvl_abi_version() != ABI_VERSION:
        raise GvlError(f"{p}: ABI version {lib.gvl_abi_version()}, this binding needs {ABI_VERSION} (rebuild the library)")
    for name in SYMBOLS:
        fn = getattr(lib, name, None)
        if fn is None:
            raise GvlError(f"{p} does not export {name}")
        if name not in ("gvl_last_error", "gvl_loader_slot_bytes", "gvl_loader_table_bytes", "gvl_tracks_scratch_bytes",
                        "gvl_ref4_bytes", "gvl_hap_plan_bytes", "gvl_svar2_workspace_bytes"):
            fn.restype = C.c_int
    lib.gvl_loader_slot_bytes.restype = C.c_int64
    lib.gvl_loader_table_bytes.restype = C.c_int64
    lib.gvl_tracks_scratch_bytes.restype = C.c_int64
    lib.gvl_ref4_bytes.restype = C.c_int64
    lib.gvl_hap_plan_bytes.restype = C.c_int64
    lib.gvl_hap_plan_bytes.argtypes = [C.c_int64, C.c_int64]
    lib.gvl_svar2_workspace_bytes.restype = C.c_int64
    lib.gvl_svar2_workspace_bytes.argtypes = [C.c_int64] * 5
    lib.gvl_ref4_bytes.argtypes = [C.c_int64]
    lib.gvl_set_tuning.argtypes = [C.c_int32, C.c_int64]
    _LIB = lib
    return lib


def set_tuning(key: int, value: int) -> None:
    """``gvl_set_tuning``: a launch-policy override (``TUNE_*``; 0 = the built-in policy).  Results never depend on it."""
    check(load().gvl_set_tuning(int(key), int(value)))


def check(rc: int) -> None:
    if rc == 0:
        return
    msg = load().gvl_last_error().decode(errors="replace")
    if rc == 1:
        raise ValueError(msg)
    raise GvlError(msg)


def check_async(clear: bool = True) -> None:
    """After a synchronisation point: raise if a launch reported an error only the device could see
    (a row longer than its batch's ``max_row_len`` hint)."""
    check(load().gvl_async_error(C.c_int(1 if clear else 0)))
