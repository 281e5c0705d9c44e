"""Device-resident side of the haplotype hot path.

``HapsDevice`` keeps what the reference caches per dataset on the host --
``_HapsFfiStatic`` (``_haps.py:233-247``: v_starts / ilens / alt_alleles /
alt_offsets / ref / ref_offsets) plus the sparse-genotype CSR (``_haps.py:435-460``)
-- resident in HBM, together with the packed 16-byte variant records the kernel
reads.  Its methods take the per-batch arrays of ``ReconstructionRequest``
(``_haps.py:58-93``) and launch the HIP kernels through the C-ABI on the current
torch HIP stream; they never synchronise the host except where the reference's
own return type forces it (ragged output needs the total length to allocate).

PyTorch is only the plumbing here (device memory + streams); every pointer that
crosses into ``libgvl_hip.so`` is a raw ``data_ptr()``.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import GvlBatch, GvlOut, GvlStatic


def _dev(x, dtype: torch.dtype, device) -> torch.Tensor | None:
    """numpy / torch -> contiguous device tensor of `dtype` (no copy when already there)."""
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        t = x
    else:
        a = np.ascontiguousarray(x)
        if a.dtype == np.bool_:
            a = a.view(np.uint8)
        if a.flags.writeable:
            t = torch.from_numpy(a)
        else:       # a read-only memmap (what the reference hands over): it is only ever copied to the device
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)
                t = torch.from_numpy(a)
    if t.dtype == torch.bool:
        t = t.view(torch.uint8) if t.is_contiguous() else t.contiguous().view(torch.uint8)
    if t.dtype != dtype:
        t = t.to(dtype)
    t = t.to(device, non_blocking=True)
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t: torch.Tensor | None):
    return None if t is None else C.c_void_p(t.data_ptr())


class _on_device:
    """`with _on_device(d)` without its constructor's `_get_device_index` (availability probe, environment lookups: ~8 us a time,
    a dozen times per batch of the Python submit loops): the same two calls torch's own context manager makes."""
    __slots__ = ("idx", "prev")

    def __init__(self, d):
        i = d if isinstance(d, int) else getattr(d, "index", None)
        self.idx = -1 if i is None else int(i)             # (-1: no-op, as for torch.cuda.device(None))
        self.prev = -1

    def __enter__(self):
        self.prev = torch.cuda._exchange_device(self.idx)
        return self

    def __exit__(self, *exc):
        torch.cuda._maybe_exchange_device(self.prev)
        return False


if not (hasattr(torch.cuda, "_exchange_device") and hasattr(torch.cuda, "_maybe_exchange_device")):
    _on_device = torch.cuda.device          # (a torch without those: its own context manager)     # noqa: F811

_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream_ptr(stream=None):
    """The HIP stream a launch goes to: `stream`, else the current device's current torch stream (read through torch's raw
    accessor where it has one: `torch.cuda.current_stream()` builds a Stream object per call, several microseconds of a
    launch that costs the device ten)."""
    if stream is not None:
        return C.c_void_p(stream.cuda_stream)
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return C.c_void_p(_RAW_STREAM(_CUR_DEVICE()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _starts_stops(geno_offsets):
    """(n+1,) or (2, n) -> (2, n) int64, as ``_as_starts_stops`` (_genotypes.py:13-21)."""
    if isinstance(geno_offsets, torch.Tensor):
        o = geno_offsets
        if o.dim() == 1:
            return torch.stack([o[:-1], o[1:]]).to(torch.int64).contiguous()
        return o.to(torch.int64).contiguous()
    o = np.asarray(geno_offsets)
    if o.ndim == 1:
        return np.ascontiguousarray(np.stack([o[:-1], o[1:]]), dtype=np.int64)
    return np.ascontiguousarray(o, dtype=np.int64)


@dataclass
class DeviceBatch:
    """Per-batch arrays in HBM + the C struct that points at them."""

    regions: torch.Tensor
    shifts: torch.Tensor
    geno_offset_idx: torch.Tensor
    keep: torch.Tensor | None
    keep_offsets: torch.Tensor | None
    to_rc: torch.Tensor | None
    out_offsets: torch.Tensor | None
    output_length: int
    max_row_len: int
    c: GvlBatch

    @property
    def n_rows(self) -> int:
        return int(self.geno_offset_idx.numel())


@dataclass
class ReconOutput:
    haps: torch.Tensor | None
    onehot: torch.Tensor | None
    out_offsets: torch.Tensor | None
    annot_v_idxs: torch.Tensor | None = None
    annot_ref_pos: torch.Tensor | None = None


class HapsDevice:
    def __init__(self, *, ref, ref_offsets, v_starts, ilens, alt_alleles, alt_offsets,
                 geno_offsets, geno_v_idxs, pad_char=ord("N"), device="cuda", inline_genotypes=None,
                 slot_records=None, packed_reference=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.GvlError("genvarloader_amd needs a HIP device (no CPU fallback)")
        self.device = torch.device(device)
        d = self.device
        self.ref = _dev(ref, torch.uint8, d)
        self.ref_offsets = _dev(ref_offsets, torch.int64, d)
        self.v_starts = _dev(v_starts, torch.int32, d)
        self.ilens = _dev(ilens, torch.int32, d)
        self.alt_alleles = _dev(alt_alleles, torch.uint8, d)
        self.alt_offsets = _dev(alt_offsets, torch.int64, d)
        go = _starts_stops(geno_offsets)
        self.geno_offsets = _dev(go, torch.int64, d)
        self.geno_v_idxs = _dev(geno_v_idxs, torch.int32, d)
        self.pad_char = int(pad_char)
        n_var = int(self.v_starts.numel())
        if int(self.alt_offsets.numel()) != n_var + 1 or int(self.ilens.numel()) != n_var:
            raise ValueError("variant table arrays disagree on n_variants")
        # 16-byte packed records (gvl_pack_variants) -- built once per dataset
        self.vrec = torch.empty((max(n_var, 1), 4), dtype=torch.int32, device=d)
        with _on_device(d):
            _lib.check(self.lib.gvl_pack_variants(
                _ptr(self.v_starts), _ptr(self.ilens), _ptr(self.alt_offsets), _ptr(self.alt_alleles),
                C.c_int64(n_var), _ptr(self.vrec), _stream_ptr()))
        n_go = int(self.geno_offsets.shape[1])
        self.c = GvlStatic(
            ref=self.ref.data_ptr(), ref_len=self.ref.numel(),
            ref_offsets=self.ref_offsets.data_ptr(), n_contigs=self.ref_offsets.numel() - 1,
            v_starts=self.v_starts.data_ptr(), ilens=self.ilens.data_ptr(),
            alt_offsets=self.alt_offsets.data_ptr(), alt_alleles=self.alt_alleles.data_ptr(),
            n_variants=n_var, alt_len=self.alt_alleles.numel(), vrec=self.vrec.data_ptr(),
            geno_o_starts=self.geno_offsets[0].data_ptr(), geno_o_stops=self.geno_offsets[1].data_ptr(),
            n_geno_offsets=n_go, geno_v_idxs=self.geno_v_idxs.data_ptr(),
            n_geno=self.geno_v_idxs.numel(), pad_char=self.pad_char, geno_rec=None, slot_rec=None, ref4=None,
        )
        # Derived layout: the variant's fields next to each genotype CSR entry (gvl_grec, 16 B per
        # entry) removes one of the dependent gathers in the kernel head.  Default: build it
        # when it costs less than a quarter of the free HBM.
        self.geno_rec = None
        n_geno = int(self.geno_v_idxs.numel())
        if inline_genotypes is None:
            inline_genotypes = n_geno > 0 and n_var > 0 and 16 * n_geno <= torch.cuda.mem_get_info(d)[0] // 4
        if inline_genotypes and n_geno > 0 and n_var > 0:
            with _on_device(d):
                self.geno_rec = torch.empty((n_geno, 4), dtype=torch.int32, device=d)
                _lib.check(self.lib.gvl_pack_genotypes(C.byref(self.c), _ptr(self.geno_rec), _stream_ptr()))
            self.c.geno_rec = self.geno_rec.data_ptr()
        # Slot-major records (gvl_srec, 128 B per genotype slot): a row reaches its variants with one
        # read after geno_offset_idx instead of geno_o_starts/stops -> records -> alt_offsets.  Same
        # default: build when it costs less than a quarter of the free HBM (and ALT bytes fit u32).
        self.slot_rec = None
        self.slot_vidx = None
        if slot_records is None:
            slot_records = (n_go > 0 and n_var > 0 and int(self.alt_alleles.numel()) < (1 << 32)
                            and 128 * n_go <= torch.cuda.mem_get_info(d)[0] // 4)
        if slot_records and n_go > 0 and n_var > 0:
            with _on_device(d):
                self.slot_rec = torch.empty((n_go * 8, 4), dtype=torch.int32, device=d)
                _lib.check(self.lib.gvl_pack_slots(C.byref(self.c), _ptr(self.slot_rec), _stream_ptr()))
            self.c.slot_rec = self.slot_rec.data_ptr()
            # ... and the records' variant indices (32 B per slot): what annotated haplotypes need next to the slot line
            with _on_device(d):
                self.slot_vidx = torch.empty((n_go * 8,), dtype=torch.int32, device=d)
                _lib.check(self.lib.gvl_pack_slot_vidx(C.byref(self.c), _ptr(self.slot_vidx), _stream_ptr()))
            self.c.slot_vidx = self.slot_vidx.data_ptr()
        # Nibble-packed reference (gvl_pack_reference, ref_len / 2 bytes): what the lean one-hot kernel reads
        # instead of the byte reference -- half the bytes and cache lines behind the cold window reads.
        self.ref4 = None
        n_ref = int(self.ref.numel())
        if packed_reference is None:      # (also for a reference-only static: gvl_get_reference's lean route reads it)
            packed_reference = n_ref > 0 and n_ref // 2 <= torch.cuda.mem_get_info(d)[0] // 4
        if packed_reference and n_ref > 0:
            with _on_device(d):
                self.ref4 = torch.empty(int(self.lib.gvl_ref4_bytes(n_ref)), dtype=torch.uint8, device=d)
                _lib.check(self.lib.gvl_pack_reference(_ptr(self.ref), C.c_int64(n_ref), _ptr(self.ref4), _stream_ptr()))
            self.c.ref4 = self.ref4.data_ptr()

    # ------------------------------------------------------------------ batches
    def prepare_batch(self, regions, shifts, geno_offset_idx, output_length, keep=None,
                      keep_offsets=None, to_rc=None, out_offsets=None, max_row_len=None, hap_plan=None, total_len=None) -> DeviceBatch:
        d = self.device
        reg = _dev(regions, torch.int32, d)
        goi = _dev(geno_offset_idx, torch.int64, d)
        if reg.dim() != 2 or reg.shape[1] < 3:
            raise ValueError("regions must be (batch, >=3) int32")
        if goi.dim() != 2 or goi.shape[0] != reg.shape[0]:
            raise ValueError("geno_offset_idx must be (batch, ploidy) int64")
        sh = _dev(shifts, torch.int32, d)
        if tuple(sh.shape) != tuple(goi.shape):
            raise ValueError("shifts must be (batch, ploidy) int32")
        kp = _dev(keep, torch.uint8, d)
        ko = _dev(keep_offsets, torch.int64, d)
        rc = _dev(to_rc, torch.uint8, d)
        n_rows = int(goi.numel())
        if rc is not None and rc.numel() != n_rows:
            raise ValueError("to_rc must have batch*ploidy entries")
        if ko is not None and ko.numel() != n_rows + 1:
            raise ValueError("keep_offsets must have batch*ploidy + 1 entries")
        oo = _dev(out_offsets, torch.int64, d)
        if oo is not None and oo.numel() != n_rows + 1:
            raise ValueError("out_offsets must have batch*ploidy + 1 entries")
        output_length = int(output_length)
        if oo is not None and max_row_len is None:
            max_row_len = int((oo[1:] - oo[:-1]).max().item()) if n_rows else 0  # host sync
        mrl = int(max_row_len) if max_row_len is not None else max(output_length, 0)
        c = GvlBatch(
            regions=reg.data_ptr(), regions_stride=reg.shape[1], shifts=sh.data_ptr(),
            geno_offset_idx=goi.data_ptr(), batch=reg.shape[0], ploidy=goi.shape[1],
            keep=None if kp is None else kp.data_ptr(),
            keep_offsets=None if ko is None else ko.data_ptr(),
            to_rc=None if rc is None else rc.data_ptr(), output_length=output_length,
            out_offsets=None if oo is None else oo.data_ptr(), max_row_len=mrl,
            hap_plan=None if hap_plan is None else hap_plan.data_ptr(),
            total_len_hint=0 if (total_len is None or oo is None) else int(total_len),
        )
        if hap_plan is not None:
            # the kernel trusts the plan blindly (its header words carry no tag): a plan made over other rows, or for another row length
            # (= another chunk stride), would give silently wrong bytes -- :meth:`hap_plan` leaves what it was made for with the tensor
            made_for = getattr(hap_plan, "_gvl_plan_of", None)
            if made_for is not None and made_for != (n_rows, output_length):
                raise ValueError(f"hap_plan was made for {made_for[0]} rows of {made_for[1]} bases, not {n_rows} rows of {output_length}")
            if oo is not None:
                raise ValueError("hap_plan goes with fixed-length rows")
        bt = DeviceBatch(reg, sh, goi, kp, ko, rc, oo, output_length, mrl, c)
        bt._hap_plan = hap_plan          # (kept alive with the batch)
        return bt

    def hap_plan(self, bt: DeviceBatch) -> torch.Tensor | None:
        """``gvl_hap_plan``: the chunk plans of a batch's long fixed-length rows (one walk per row; pass the result to
        :meth:`prepare_batch` as ``hap_plan`` for the same request arrays).  None when rows of this length are not planned."""
        n = int(self.lib.gvl_hap_plan_bytes(C.c_int64(bt.n_rows), C.c_int64(bt.output_length)))
        if n <= 0:
            return None
        plan = torch.empty(n, dtype=torch.uint8, device=self.device)
        with _on_device(self.device):
            _lib.check(self.lib.gvl_hap_plan(C.byref(self.c), C.byref(bt.c), _ptr(plan), _stream_ptr()))
        plan._gvl_plan_of = (bt.n_rows, bt.output_length)      # (checked by prepare_batch)
        return plan

    def hap_offsets(self, bt: DeviceBatch, want_diffs=False):
        """Fused-entry sizing (ffi/mod.rs:769-811) on the device.
        -> (out_offsets i64[K+1], total_and_max i64[2], diffs | None); no host sync."""
        n = bt.n_rows
        oo = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        tm = torch.empty(2, dtype=torch.int64, device=self.device)
        diffs = torch.empty(tuple(bt.geno_offset_idx.shape), dtype=torch.int32, device=self.device) if want_diffs else None
        with _on_device(self.device):
            _lib.check(self.lib.gvl_hap_offsets(C.byref(self.c), C.byref(bt.c), _ptr(diffs), _ptr(oo),
                                                _ptr(tm), _stream_ptr()))
        return oo, tm, diffs

    def get_diffs_sparse(self, geno_offset_idx, keep=None, keep_offsets=None, q_starts=None, q_ends=None):
        """get_diffs_sparse (ffi/mod.rs:145-157) -> i32 (B, P) device tensor."""
        d = self.device
        goi = _dev(geno_offset_idx, torch.int64, d)
        kp, ko = _dev(keep, torch.uint8, d), _dev(keep_offsets, torch.int64, d)
        qs, qe = _dev(q_starts, torch.int32, d), _dev(q_ends, torch.int32, d)
        c = GvlBatch(geno_offset_idx=goi.data_ptr(), batch=goi.shape[0], ploidy=goi.shape[1],
                     keep=None if kp is None else kp.data_ptr(),
                     keep_offsets=None if ko is None else ko.data_ptr())
        diffs = torch.empty(tuple(goi.shape), dtype=torch.int32, device=d)
        with _on_device(d):
            _lib.check(self.lib.gvl_get_diffs_sparse(C.byref(self.c), C.byref(c), _ptr(qs), _ptr(qe),
                                                     C.c_int64(1), _ptr(diffs), _stream_ptr()))
        return diffs

    def choose_exonic_variants(self, starts, ends, geno_offset_idx, max_per_row=None):
        """choose_exonic_variants (src/genotypes/mod.rs:127-176) -> (keep u8[total], keep_offsets i64[K+1])
        device tensors; one host sync for the exactly-sized mask -- or none when the caller knows a bound on a row's variants
        (``max_per_row``: the mask is then allocated for rows x that; only its first keep_offsets[-1] bytes mean anything)."""
        d = self.device
        goi = _dev(geno_offset_idx, torch.int64, d)
        st_, en_ = _dev(starts, torch.int32, d), _dev(ends, torch.int32, d)
        B, P = int(goi.shape[0]), int(goi.shape[1])
        ko = torch.empty(B * P + 1, dtype=torch.int64, device=d)
        tm = torch.zeros(2, dtype=torch.int64, device=d)
        with _on_device(d):
            _lib.check(self.lib.gvl_keep_offsets(C.byref(self.c), _ptr(goi), C.c_int64(B), C.c_int64(P), _ptr(ko), _ptr(tm),
                                                 _stream_ptr()))
            if max_per_row is not None:
                total = B * P * max(0, int(max_per_row))
            else:
                total = int(tm[0].item()) if B * P else 0
            keep = torch.empty(total, dtype=torch.uint8, device=d)
            if total:
                _lib.check(self.lib.gvl_choose_exonic_variants(C.byref(self.c), _ptr(st_), _ptr(en_), _ptr(goi), C.c_int64(B),
                                                               C.c_int64(P), _ptr(ko), _ptr(keep), _stream_ptr()))
        return keep, ko

    # -------------------------------------------------------------- reconstruct
    def alloc_output(self, bt: DeviceBatch, total: int, *, haps=True, onehot=False, layout="lc",
                     annotate=False, write_offsets=True) -> tuple[ReconOutput, GvlOut]:
        d = self.device
        n = bt.n_rows
        h = torch.empty(total, dtype=torch.uint8, device=d) if haps else None
        if onehot:
            if layout == "lc":
                oh = torch.empty((total, 4), dtype=torch.uint8, device=d)
            else:
                if bt.output_length < 0 or bt.out_offsets is not None:
                    raise ValueError("channel-major one-hot needs fixed-length rows")
                oh = torch.empty((n, 4, bt.output_length), dtype=torch.uint8, device=d)
        else:
            oh = None
        av = torch.empty(total, dtype=torch.int32, device=d) if annotate else None
        ap = torch.empty(total, dtype=torch.int32, device=d) if annotate else None
        oo = None
        if write_offsets and bt.out_offsets is None:
            oo = torch.empty(n + 1, dtype=torch.int64, device=d)
        c = GvlOut(haps=None if h is None else h.data_ptr(),
                   onehot=None if oh is None else oh.data_ptr(),
                   onehot_layout=_lib.GVL_ONEHOT_LC if layout == "lc" else _lib.GVL_ONEHOT_CL,
                   annot_v_idxs=None if av is None else av.data_ptr(),
                   annot_ref_pos=None if ap is None else ap.data_ptr(),
                   out_offsets=None if oo is None else oo.data_ptr())
        return ReconOutput(h, oh, oo if oo is not None else bt.out_offsets, av, ap), c

    def launch(self, bt: DeviceBatch, out_c: GvlOut, stream=None) -> None:
        """One pass of the hot path over one batch: a single kernel launch."""
        _lib.check(self.lib.gvl_reconstruct(C.byref(self.c), C.byref(bt.c), C.byref(out_c), _stream_ptr(stream)))

    def pack_many(self, bts, out_cs):
        """C arrays for :meth:`launch_many` (build once, launch many times)."""
        n = len(bts)
        if n != len(out_cs) or n == 0:
            raise ValueError("need as many outputs as batches")
        return (GvlBatch * n)(*[b.c for b in bts]), (GvlOut * n)(*out_cs), n

    def launch_many(self, packed, stream=None) -> None:
        """``gvl_reconstruct_many``: several batches, one launch (up to ``GVL_MANY_MAX`` per launch)."""
        b, o, n = packed
        _lib.check(self.lib.gvl_reconstruct_many(C.byref(self.c), b, o, C.c_int32(n), _stream_ptr(stream)))

    def reconstruct(self, regions, shifts, geno_offset_idx, output_length, keep=None, keep_offsets=None,
                    to_rc=None, *, out_offsets=None, haps=True, onehot=False, layout="lc",
                    annotate=False) -> ReconOutput:
        """reconstruct_haplotypes_fused (ffi/mod.rs:722-860) on the device.
        Fixed length: one launch, no host sync.  Ragged (output_length < 0): sizes on
        the device, then one host read of {total, max} to allocate (the reference
        returns an exactly-sized buffer, ffi/mod.rs:814-815)."""
        with _on_device(self.device):
            bt = self.prepare_batch(regions, shifts, geno_offset_idx, output_length, keep, keep_offsets,
                                    to_rc, out_offsets)
            n = bt.n_rows
            if bt.out_offsets is not None:
                total = int(bt.out_offsets[-1].item()) if n else 0
                # (the total this call reads anyway tells the dispatch whether a batch with a few long rows is mostly short ones)
                bt = self.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length, bt.keep, bt.keep_offsets,
                                        bt.to_rc, bt.out_offsets, max_row_len=bt.max_row_len, total_len=total)
            elif bt.output_length >= 0:
                total = n * bt.output_length
            else:
                oo, tm, _ = self.hap_offsets(bt)
                total, mx = (int(v) for v in tm.cpu().tolist())  # host sync (allocation size)
                bt = self.prepare_batch(bt.regions, bt.shifts, bt.geno_offset_idx, bt.output_length,
                                        bt.keep, bt.keep_offsets, bt.to_rc, oo, max_row_len=mx, total_len=total)
            out, out_c = self.alloc_output(bt, total, haps=haps, onehot=onehot, layout=layout,
                                           annotate=annotate)
            if n == 0 or total == 0:
                if bt.out_offsets is None:
                    out.out_offsets.zero_()
                return out
            self.launch(bt, out_c)
            out._keepalive = bt  # batch tensors must outlive the async launch
            return out

    # ------------------------------------------------------------ reference / rc
    def get_reference(self, regions, out_offsets, to_rc=None, *, onehot=False, max_row_len=None):
        """get_reference (ffi/mod.rs:2401-2429) -> u8[total] (and (total, 4) one-hot)."""
        d = self.device
        with _on_device(d):
            reg = _dev(regions, torch.int32, d)
            oo = _dev(out_offsets, torch.int64, d)
            rc = _dev(to_rc, torch.uint8, d)
            n = int(reg.shape[0])
            total = int(oo[-1].item()) if n else 0
            if max_row_len is None:
                max_row_len = int((oo[1:] - oo[:-1]).max().item()) if n else 0
            out = torch.empty(total, dtype=torch.uint8, device=d)
            oh = torch.empty((total, 4), dtype=torch.uint8, device=d) if onehot else None
            if total == 0:
                return (out, oh) if onehot else out
            _lib.check(self.lib.gvl_get_reference(C.byref(self.c), _ptr(reg), C.c_int64(reg.shape[1]),
                                                  C.c_int64(n), _ptr(oo), C.c_int64(max_row_len), _ptr(rc),
                                                  _ptr(out), _ptr(oh), _stream_ptr()))
            return (out, oh) if onehot else out


def _get_reference_many(self, batches, *, onehot=False, haps=True):
    """``gvl_get_reference_many``: up to 16 batches of regions in ONE launch.  ``batches``: a list of ``(regions, out_offsets, to_rc |
    None, total, max_row_len)`` with device tensors (``total`` = ``out_offsets[-1]``, ``max_row_len`` = a bound on the rows' lengths,
    both host ints: no synchronisation here) -> a list of ``(bytes | None, one-hot | None)`` device tensors, one pair per batch."""
    if not haps and not onehot:
        raise ValueError("get_reference_many: ask for the bytes, the one-hot or both")
    d = self.device
    n = len(batches)
    arr = (_lib.GvlRefBatch * n)()
    outs, keep = [], []
    with _on_device(d):
        for i, (regions, out_offsets, to_rc, total, max_row_len) in enumerate(batches):
            reg = _dev(regions, torch.int32, d)
            oo = _dev(out_offsets, torch.int64, d)
            rc = _dev(to_rc, torch.uint8, d)
            o = torch.empty(int(total), dtype=torch.uint8, device=d) if haps else None
            oh = torch.empty((int(total), 4), dtype=torch.uint8, device=d) if onehot else None
            arr[i] = _lib.GvlRefBatch(regions=reg.data_ptr(), regions_stride=reg.shape[1], n_rows=reg.shape[0], out_offsets=oo.data_ptr(),
                                      max_row_len=int(max_row_len), to_rc=None if rc is None else rc.data_ptr(),
                                      out=None if o is None else o.data_ptr(), onehot=None if oh is None else oh.data_ptr())
            outs.append((o, oh)); keep.append((reg, oo, rc))
        _lib.check(self.lib.gvl_get_reference_many(C.byref(self.c), arr, C.c_int32(n), _stream_ptr()))
    for (o, oh), k in zip(outs, keep):          # (the request arrays must outlive the asynchronous launch)
        (o if o is not None else oh)._keepalive = k
    return outs


HapsDevice.get_reference_many = _get_reference_many


def rc_flat_rows_inplace(data: torch.Tensor, offsets, to_rc) -> None:
    """rc_flat_rows_inplace (reverse.rs:56-69) on a device u8 tensor."""
    lib = _lib.load()
    d = data.device
    oo, rc = _dev(offsets, torch.int64, d), _dev(to_rc, torch.uint8, d)
    assert data.dtype == torch.uint8 and data.is_contiguous()
    if data.numel() == 0 or rc.numel() == 0:
        return
    with _on_device(d):
        _lib.check(lib.gvl_rc_rows(_ptr(data), _ptr(oo), _ptr(rc), C.c_int64(rc.numel()), _stream_ptr()))


def rc_bounded_rows_inplace(data: torch.Tensor, bounds, to_rc) -> None:
    """rc_bounded_rows_inplace (reverse.rs:75-84): rows as (start, end) pairs, i64 (n, 2)."""
    lib = _lib.load()
    d = data.device
    bd, rc = _dev(bounds, torch.int64, d), _dev(to_rc, torch.uint8, d)
    assert data.dtype == torch.uint8 and data.is_contiguous() and bd.dim() == 2 and bd.shape[1] == 2
    if data.numel() == 0 or rc.numel() == 0:
        return
    with _on_device(d):
        _lib.check(lib.gvl_rc_bounded_rows(_ptr(data), _ptr(bd), _ptr(rc), C.c_int64(rc.numel()), _stream_ptr()))


def reverse_flat_rows_inplace(data: torch.Tensor, offsets, to_rc) -> None:
    """reverse_flat_rows_inplace<T> (reverse.rs:25-38) for 4-byte elements."""
    lib = _lib.load()
    d = data.device
    oo, rc = _dev(offsets, torch.int64, d), _dev(to_rc, torch.uint8, d)
    assert data.element_size() == 4 and data.is_contiguous()
    if data.numel() == 0 or rc.numel() == 0:
        return
    with _on_device(d):
        _lib.check(lib.gvl_reverse_rows_4(_ptr(data), _ptr(oo), _ptr(rc), C.c_int64(rc.numel()), _stream_ptr()))


def onehot(x: torch.Tensor) -> torch.Tensor:
    """Stand-alone one-hot (a10): u8 (...,) -> u8 (..., 4)."""
    lib = _lib.load()
    assert x.dtype == torch.uint8 and x.is_cuda
    x = x.contiguous()
    out = torch.empty(tuple(x.shape) + (4,), dtype=torch.uint8, device=x.device)
    with _on_device(x.device):
        _lib.check(lib.gvl_onehot(_ptr(x), C.c_int64(x.numel()), _ptr(out), _stream_ptr()))
    return out


# ------------------------------------------------------------------------------- tracks (a12)
def intervals_prefix_max(itv_ends, itv_offsets, device="cuda") -> torch.Tensor:
    """Running max of the interval ends inside each list (once per interval set)."""
    lib = _lib.load()
    d = torch.device(device)
    b, io = _dev(itv_ends, torch.int32, d), _dev(itv_offsets, torch.int64, d)
    pm = torch.empty_like(b)
    with _on_device(d):
        _lib.check(lib.gvl_intervals_prefix_max(_ptr(b), _ptr(io), C.c_int64(int(io.numel()) - 1), _ptr(pm), _stream_ptr()))
    return pm


def intervals_bucket_index(itv_starts, itv_pmax_ends, itv_offsets, device="cuda"):
    """Coarse per-list index for the painter (``gvl_intervals_bucket_counts`` / ``_fill``; once per
    interval set, one host read of the bucket total) -> (bkt_offsets, bkt_base, bkt_lo, bkt_hi)."""
    lib = _lib.load()
    d = torch.device(device)
    a, pm, io = _dev(itv_starts, torch.int32, d), _dev(itv_pmax_ends, torch.int32, d), _dev(itv_offsets, torch.int64, d)
    n_lists = int(io.numel()) - 1
    bo = torch.empty(n_lists + 1, dtype=torch.int64, device=d)
    base = torch.empty(max(n_lists, 1), dtype=torch.int32, device=d)
    tot = torch.zeros(2, dtype=torch.int64, device=d)
    with _on_device(d):
        _lib.check(lib.gvl_intervals_bucket_counts(_ptr(a), _ptr(io), C.c_int64(n_lists), _ptr(bo), _ptr(base), _ptr(tot),
                                                   _stream_ptr()))
        n_b = int(tot[0].item())
        lo = torch.empty(max(n_b, 1), dtype=torch.int32, device=d)
        hi = torch.empty(max(n_b, 1), dtype=torch.int32, device=d)
        _lib.check(lib.gvl_intervals_bucket_fill(_ptr(a), _ptr(pm), _ptr(io), C.c_int64(n_lists), _ptr(bo), _ptr(base),
                                                 C.c_int64(n_b), _ptr(lo), _ptr(hi), _stream_ptr()))
    return bo, base, lo, hi


def make_track_set(itv_starts, itv_ends, itv_values, itv_offsets, itv_pmax_ends=None, bucket_index=None, device="cuda"):
    """A ``gvl_track_set`` over device-resident interval arrays (+ their derived arrays) for ``gvl_paint_tracks``;
    -> (struct, tensors to keep alive)."""
    d = torch.device(device)
    a, b = _dev(itv_starts, torch.int32, d), _dev(itv_ends, torch.int32, d)
    v, io = _dev(itv_values, torch.float32, d), _dev(itv_offsets, torch.int64, d)
    pm = _dev(itv_pmax_ends, torch.int32, d)
    bk = bucket_index
    ts = _lib.GvlTrackSet(
        itv_starts=a.data_ptr(), itv_ends=b.data_ptr(), itv_values=v.data_ptr(), itv_offsets=io.data_ptr(), n_intervals=int(a.numel()),
        itv_pmax_ends=None if pm is None else pm.data_ptr(),
        bkt_offsets=None if bk is None else bk[0].data_ptr(), bkt_base=None if bk is None else bk[1].data_ptr(),
        bkt_lo=None if bk is None else bk[2].data_ptr(), bkt_hi=None if bk is None else bk[3].data_ptr(),
        tile_complete=0, has_fill=0, fill_strategy=0, fill_param=0.0, list_div=1)
    return ts, (a, b, v, io, pm, bk)


def intervals_to_tracks(offset_idxs, starts, itv_starts, itv_ends, itv_values, itv_offsets, out_offsets,
                        device="cuda", starts_stride=1, max_row_len=None, itv_pmax_ends=None, track_set=None) -> torch.Tensor:
    """intervals_to_tracks (src/intervals.rs:19-126) -> f32[out_offsets[-1]] device tensor.  ``track_set`` (from
    :func:`make_track_set`, with the prefix maxima and the bucket index): the painter's tiled + bitmap path (``gvl_paint_tracks``)."""
    lib = _lib.load()
    d = torch.device(device)
    oi, st = _dev(offset_idxs, torch.int64, d), _dev(starts, torch.int32, d)
    a, b = _dev(itv_starts, torch.int32, d), _dev(itv_ends, torch.int32, d)
    v, io = _dev(itv_values, torch.float32, d), _dev(itv_offsets, torch.int64, d)
    oo = _dev(out_offsets, torch.int64, d)
    n = int(oi.numel())
    total = int(oo[-1].item()) if n else 0
    out = torch.empty(total, dtype=torch.float32, device=d)
    if total == 0:
        return out
    if max_row_len is None:
        max_row_len = int((oo[1:] - oo[:-1]).max().item())
    if track_set is not None:
        with _on_device(d):
            _lib.check(lib.gvl_paint_tracks(C.byref(track_set), _ptr(oi), _ptr(st), C.c_int64(starts_stride), C.c_int64(n), _ptr(out),
                                            _ptr(oo), C.c_int64(max_row_len), _stream_ptr()))
        return out
    with _on_device(d):
        pm = _dev(itv_pmax_ends, torch.int32, d)
        _lib.check(lib.gvl_intervals_to_tracks(_ptr(oi), _ptr(st), C.c_int64(starts_stride), C.c_int64(n), _ptr(a),
                                               _ptr(b), _ptr(v), _ptr(io), C.c_int64(int(a.numel())), _ptr(pm),
                                               _ptr(out), _ptr(oo), C.c_int64(max_row_len), _stream_ptr()))
    return out


def realign_tracks(dev: "HapsDevice", regions, shifts, geno_offset_idx, out_offsets, tracks, track_offsets,
                   params, strategy_id=0, base_seed=0, keep=None, keep_offsets=None, to_rc=None, query_seed=None) -> torch.Tensor:
    """shift_and_realign_tracks_sparse (src/tracks/mod.rs:495-667) + the reversal of negative-strand
    rows (src/ffi/mod.rs:2657-2668) -> f32[out_offsets[-1]] device tensor."""
    d = dev.device
    with _on_device(d):
        bt = dev.prepare_batch(regions, shifts, geno_offset_idx, -1, keep, keep_offsets, to_rc, out_offsets)
        qs = _dev(query_seed, torch.int64, d)        # (src/tracks/mod.rs:754-760: the FlankSample seed's query component, per local query)
        if qs is not None:
            if qs.numel() != bt.regions.shape[0]:
                raise ValueError("query_seed must have one entry per query")
            bt.c.query_seed = qs.data_ptr()
        tr, to = _dev(tracks, torch.float32, d), _dev(track_offsets, torch.int64, d)
        total = int(bt.out_offsets[-1].item()) if bt.n_rows else 0
        out = torch.empty(total, dtype=torch.float32, device=d)
        if total == 0:
            return out
        p = (C.c_double * 1)(float(np.asarray(params, np.float64).ravel()[0]))
        _lib.check(dev.lib.gvl_realign_tracks(C.byref(dev.c), C.byref(bt.c), _ptr(tr), _ptr(to), p,
                                              C.c_int64(int(strategy_id)),
                                              C.c_uint64(int(base_seed) & 0xFFFFFFFFFFFFFFFF), _ptr(out),
                                              _stream_ptr()))
        out._keepalive = (bt, tr, to, qs)
        return out
