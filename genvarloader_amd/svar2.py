"""SVAR2 two-source variant provider on the device.

The reference's second genotype source (``/root/reference/src/svar2/mod.rs``): per haplotype the merge of its ``var_key``
calls with the present entries of its query's ``dense`` window (``merge_hap``, :45-72), each entry a (position, 32-bit key)
pair decoded by the third-party crate ``svar2-codec``.  That crate is not part of the reference's tree and its bit layout
is stated nowhere in it, so the channels cross this boundary DECODED -- the form the reference itself gives a decoded key
(``decode_alt``, :17-30; ``VariantsSoa``, :292-306): entry ``e`` has ``v_diff = ilen[e]`` and the allele
``alt_bytes[alt_off[e] : alt_off[e + 1]]``, empty for a pure deletion.  An integrator who links the codec decodes once per
entry (``decode_with``); nothing here guesses the key's bits.

``merge`` runs ONE launch per batch (``gvl_svar2_merge``) that writes the batch's merged variants as a sparse table of the
SVAR1 shape; the result duck-types :class:`HapsDevice`, so every device entry point -- length deltas, ragged sizing,
reconstruction with RC / one-hot in either layout, track realignment -- consumes a SVAR2 batch unchanged.  The numpy-in /
numpy-out functions below mirror the reference's PyO3 entry points (``src/ffi/mod.rs:874-997``, ``:1835-1966``) with the
key arguments replaced by their decoded form:

    reference                                   here
    ``vk_key``                                  ``vk_ilen, vk_alt_off``
    ``dense_key``                               ``dense_ilen, dense_alt_off``
    ``lut_bytes, lut_off``                      ``alt_bytes``  (one pool: inline alleles and LUT rows alike)

Parity: pinned by the reference's Rust known-answer tests and by vectors of its independent Python consensus
(``tests/golden/pyref_svar2_consensus.npz``); no 200-case Rust golden exists for these entry points.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from . import device as _device
from ._lib import GvlBatch, GvlStatic, GvlSvar2Batch
from .device import HapsDevice, _dev, _on_device, _ptr, _stream_ptr


class Svar2Channels:
    """One batch of decoded channels in HBM + the C struct that points at them."""

    def __init__(self, vk_pos, vk_ilen, vk_alt_off, vk_off, dense_pos, dense_ilen, dense_alt_off, dense_range,
                 dense_present, dense_present_off, alt_bytes, *, filter_exonic=False, device="cuda"):
        d = torch.device(device)
        self.device = d
        self.vk_pos = _dev(vk_pos, torch.int32, d).reshape(-1)
        self.vk_ilen = _dev(vk_ilen, torch.int32, d).reshape(-1)
        self.vk_alt_off = _dev(vk_alt_off, torch.int64, d).reshape(-1)
        self.vk_off = _dev(vk_off, torch.int64, d).reshape(-1)
        self.dense_pos = _dev(dense_pos, torch.int32, d).reshape(-1)
        self.dense_ilen = _dev(dense_ilen, torch.int32, d).reshape(-1)
        self.dense_alt_off = _dev(dense_alt_off, torch.int64, d).reshape(-1)
        self.dense_range = _dev(np.asarray(dense_range).reshape(-1, 2) if not isinstance(dense_range, torch.Tensor) else dense_range,
                                torch.int32, d).reshape(-1, 2)
        self.dense_present = _dev(dense_present, torch.uint8, d).reshape(-1)
        self.dense_present_off = _dev(dense_present_off, torch.int64, d).reshape(-1)
        self.alt_bytes = _dev(alt_bytes, torch.uint8, d).reshape(-1)
        n_vk, n_dense = int(self.vk_pos.numel()), int(self.dense_pos.numel())
        if int(self.vk_ilen.numel()) != n_vk or int(self.vk_alt_off.numel()) != n_vk + 1:
            raise ValueError("var_key channel: vk_ilen needs n_vk entries, vk_alt_off n_vk + 1")
        if int(self.dense_ilen.numel()) != n_dense or int(self.dense_alt_off.numel()) != n_dense + 1:
            raise ValueError("dense channel: dense_ilen needs n_dense entries, dense_alt_off n_dense + 1")
        if int(self.vk_off.numel()) != int(self.dense_present_off.numel()) or int(self.vk_off.numel()) < 1:
            raise ValueError("vk_off and dense_present_off need batch*ploidy + 1 entries each")
        self.n_work = int(self.vk_off.numel()) - 1
        self.filter_exonic = bool(filter_exonic)
        self.c = GvlSvar2Batch(
            vk_pos=self.vk_pos.data_ptr(), vk_ilen=self.vk_ilen.data_ptr(), vk_alt_off=self.vk_alt_off.data_ptr(),
            vk_off=self.vk_off.data_ptr(), n_vk=n_vk,
            dense_pos=self.dense_pos.data_ptr(), dense_ilen=self.dense_ilen.data_ptr(),
            dense_alt_off=self.dense_alt_off.data_ptr(), n_dense=n_dense,
            dense_range=self.dense_range.data_ptr(), dense_present=self.dense_present.data_ptr(),
            dense_present_bits=8 * int(self.dense_present.numel()), dense_present_off=self.dense_present_off.data_ptr(),
            alt_bytes=self.alt_bytes.data_ptr(), alt_len=int(self.alt_bytes.numel()),
            filter_exonic=1 if filter_exonic else 0)


class Svar2Merged(HapsDevice):
    """A batch's merged table (``gvl_svar2_merge``).  Duck-types :class:`HapsDevice`: ``.c`` is the merged ``gvl_static``,
    ``.geno_offset_idx`` the (batch, ploidy) i64 device tensor 0 .. batch*ploidy-1 that goes with it."""

    def __init__(self, ref_dev: HapsDevice, ch: Svar2Channels, regions, ploidy: int):  # noqa: D107 (no HapsDevice.__init__: nothing is uploaded)
        self.lib = ref_dev.lib
        self.device = ref_dev.device
        self.pad_char = ref_dev.pad_char
        d = self.device
        reg = _dev(regions, torch.int32, d)
        if reg.dim() != 2 or reg.shape[1] < 3:
            raise ValueError("regions must be (batch, >=3) int32")
        batch, ploidy = int(reg.shape[0]), int(ploidy)
        if batch * ploidy != ch.n_work:
            raise ValueError("vk_off / dense_present_off must have batch*ploidy + 1 entries")
        if int(ch.dense_range.shape[0]) != batch:
            raise ValueError("dense_range must be (batch, 2)")
        nbytes = int(self.lib.gvl_svar2_workspace_bytes(batch, ploidy, ch.c.n_vk, ch.c.dense_present_bits, ch.c.alt_len))
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=d)
        base = self.workspace.data_ptr()
        self._ws_off = (-base) % 256
        self.c = GvlStatic()
        goi = C.c_void_p()
        with _on_device(d):
            _lib.check(self.lib.gvl_svar2_merge(C.byref(ref_dev.c), C.byref(ch.c), _ptr(reg), C.c_int64(reg.shape[1]),
                                                C.c_int64(batch), C.c_int64(ploidy), C.c_void_p(base + self._ws_off),
                                                C.c_int64(nbytes), C.byref(self.c), C.byref(goi), _stream_ptr()))
        off = int(goi.value) - base
        self.geno_offset_idx = self.workspace[off:off + 8 * batch * ploidy].view(torch.int64).reshape(batch, ploidy)
        self.regions = reg
        self._keepalive = (ref_dev, ch)


def merge(ref_dev: HapsDevice, ch: Svar2Channels, regions, ploidy: int) -> Svar2Merged:
    """``gvl_svar2_merge``: one launch; the batch's merged table in a workspace the result owns."""
    return Svar2Merged(ref_dev, ch, regions, ploidy)


def decode_with(decode_alt, keys, lut_bytes=b"", lut_off=(0,)):
    """Decode a channel's keys with the INTEGRATOR's ``decode_alt(key, lut_bytes, lut_off) -> (v_diff, allele bytes)``
    (``src/svar2/mod.rs:17-30`` over the real ``svar2_codec::decode_key``) into ``(ilen i32[n], alt_off i64[n + 1],
    alt_bytes u8)``.  A plain loop: it shows the contract; a production integrator decodes in its own language."""
    ilen = np.zeros(len(keys), np.int32)
    off = np.zeros(len(keys) + 1, np.int64)
    pool = bytearray()
    for i, k in enumerate(keys):
        d, alt = decode_alt(k, lut_bytes, lut_off)
        ilen[i] = d
        pool += bytes(alt)
        off[i + 1] = len(pool)
    return ilen, off, np.frombuffer(bytes(pool), np.uint8).copy()


# ------------------------------------------------------------------------------- numpy in -> numpy out
def _ref_static(ref_, ref_offsets, pad_char):
    from . import ffi

    return ffi._ref_static(ref_, ref_offsets, pad_char)


# (the length deltas and the tracks read no reference: ONE stand-in, kept alive, so that its address-keyed cache entry is hit and the
# real datasets' entries are not pushed out of the cache by a fresh one per call)
_NO_REF = (np.zeros(1, np.uint8), np.array([0, 1], np.int64))
_NO_REF[0].flags.writeable = False
_NO_REF[1].flags.writeable = False


def _no_ref_static():
    return _ref_static(_NO_REF[0], _NO_REF[1], ord("N"))


def _channels(vk_pos, vk_ilen, vk_alt_off, vk_off, dense_pos, dense_ilen, dense_alt_off, dense_range, dense_present,
              dense_present_off, alt_bytes, filter_exonic, device):
    pb = np.ascontiguousarray(dense_present, np.uint8).reshape(-1)
    po = np.ascontiguousarray(dense_present_off, np.int64).reshape(-1)
    if len(po) and int(po[-1]) > 8 * len(pb):
        raise ValueError("dense_present is shorter than dense_present_off[-1] bits")
    return Svar2Channels(vk_pos, vk_ilen, vk_alt_off, vk_off, dense_pos, dense_ilen, dense_alt_off, dense_range, pb, po,
                         alt_bytes, filter_exonic=filter_exonic, device=device)


def hap_diffs_svar2(regions, ploidy, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen, dense_range, dense_present,
                    dense_present_off, filter_exonic=False):
    """``hap_diffs_svar2`` (src/svar2/mod.rs:78-160) -> i32 (n_q, ploidy): ``get_diffs_sparse``'s query-clipped branch over
    the merged table."""
    from . import ffi

    regions = np.ascontiguousarray(regions, np.int32)
    n_vk, n_dense = len(np.asarray(vk_pos).reshape(-1)), len(np.asarray(dense_pos).reshape(-1))
    dev = _no_ref_static()
    ch = _channels(vk_pos, vk_ilen, np.zeros(n_vk + 1, np.int64), vk_off, dense_pos, dense_ilen, np.zeros(n_dense + 1, np.int64),
                   dense_range, dense_present, dense_present_off, np.zeros(0, np.uint8), filter_exonic, dev.device)
    reg0 = regions.copy()
    reg0[:, 0] = 0                       # (the length deltas read no reference: any contig will do for the anchors)
    m = merge(dev, ch, reg0, int(ploidy))
    d = m.get_diffs_sparse(m.geno_offset_idx, None, None, np.ascontiguousarray(regions[:, 1]), np.ascontiguousarray(regions[:, 2]))
    return ffi._np(d)


def check_disjoint_bounds_within(out_bounds, out_len: int) -> None:
    """``check_disjoint_bounds_within`` (src/ffi/mod.rs:101-139): every row in range, rows pairwise disjoint --
    ValueError otherwise (sorted by (start, end) like the Rust, so that a zero-length row may share its start)."""
    b = np.asarray(out_bounds, np.int64).reshape(-1, 2)
    bad = np.nonzero((b[:, 0] < 0) | (b[:, 0] > b[:, 1]) | (b[:, 1] > out_len))[0]
    if len(bad):
        k = int(bad[0])
        raise ValueError(f"out_bounds[{k}] = ({int(b[k, 0])}, {int(b[k, 1])}) is invalid for out.len() = {out_len}; every row must "
                         "satisfy 0 <= start <= end <= out.len()")
    order = np.lexsort((b[:, 1], b[:, 0]))
    s, e = b[order, 0], b[order, 1]
    if len(s) > 1:
        run_max = np.maximum.accumulate(e)[:-1]
        over = np.nonzero(s[1:] < run_max)[0]
        if len(over):
            k = int(order[over[0] + 1])
            raise ValueError(f"out_bounds rows must be pairwise disjoint: row {k} = ({int(b[k, 0])}, {int(b[k, 1])}) overlaps an earlier row")


def reconstruct_haplotypes_from_svar2(
    regions, shifts, vk_pos, vk_ilen, vk_alt_off, vk_off, dense_pos, dense_ilen, dense_alt_off, dense_range,
    dense_present, dense_present_off, alt_bytes, ref_, ref_offsets, pad_char, output_length, parallel=False, *,
    filter_exonic=False, to_rc=None, onehot=False, layout="lc",
):
    """The fused entry (src/ffi/mod.rs:874-997) -> ``(out u8[total], out_offsets i64[K + 1])``; ``output_length`` -1 = ragged
    (region length + ``hap_diffs_svar2``), >= 0 = fixed.  ``to_rc`` / ``onehot`` are this library's fused extras (the
    reference's entry has neither): ``onehot=True`` returns ``(onehot, out_offsets, out)``."""
    from . import ffi

    dev = _ref_static(ref_, ref_offsets, pad_char)
    ch = _channels(vk_pos, vk_ilen, vk_alt_off, vk_off, dense_pos, dense_ilen, dense_alt_off, dense_range, dense_present,
                   dense_present_off, alt_bytes, filter_exonic, dev.device)
    shifts = np.ascontiguousarray(shifts, np.int32)
    m = merge(dev, ch, regions, shifts.shape[1])
    res = m.reconstruct(m.regions, shifts, m.geno_offset_idx, int(output_length), None, None, to_rc, onehot=onehot, layout=layout)
    if onehot:
        return ffi._np(res.onehot), ffi._np(res.out_offsets), ffi._np(res.haps)
    return ffi._np(res.haps), ffi._np(res.out_offsets)


def reconstruct_haplotypes_from_svar2_into(
    out, out_bounds, regions, shifts, vk_pos, vk_ilen, vk_alt_off, vk_off, dense_pos, dense_ilen, dense_alt_off,
    dense_range, dense_present, dense_present_off, alt_bytes, ref_, ref_offsets, pad_char, parallel=False,
    filter_exonic=False,
):
    """The core's scatter write (src/reconstruct/mod.rs:619-826 as the ``_into`` entry drives it, src/ffi/mod.rs:1227-1380):
    row k lands at ``out[out_bounds[k, 0] : out_bounds[k, 1]]``, bytes outside the rows stay what they were."""
    from . import ffi

    if not (isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.flags.c_contiguous):
        raise ValueError("`out` must be a C-contiguous uint8 array")
    ob = np.ascontiguousarray(np.asarray(out_bounds, np.int64).reshape(-1, 2))
    check_disjoint_bounds_within(ob, out.size)
    dev = _ref_static(ref_, ref_offsets, pad_char)
    ch = _channels(vk_pos, vk_ilen, vk_alt_off, vk_off, dense_pos, dense_ilen, dense_alt_off, dense_range, dense_present,
                   dense_present_off, alt_bytes, filter_exonic, dev.device)
    shifts = np.ascontiguousarray(shifts, np.int32)
    m = merge(dev, ch, regions, shifts.shape[1])
    d = m.device
    if ob.shape[0] != m.geno_offset_idx.numel():
        raise ValueError("out_bounds must have one row per (query, hap)")
    with _on_device(d):
        buf = torch.from_numpy(out).to(d) if out.size else torch.empty(0, dtype=torch.uint8, device=d)
        obd = _dev(ob, torch.int64, d)
        sh = _dev(shifts, torch.int32, d)
        mrl = int((ob[:, 1] - ob[:, 0]).max()) if len(ob) else 0
        bt = GvlBatch(regions=m.regions.data_ptr(), regions_stride=m.regions.shape[1], shifts=sh.data_ptr(),
                      geno_offset_idx=m.geno_offset_idx.data_ptr(), batch=m.regions.shape[0], ploidy=shifts.shape[1],
                      output_length=-1, max_row_len=mrl, out_bounds=obd.data_ptr())
        oc = _lib.GvlOut(haps=buf.data_ptr() if out.size else None, onehot_layout=_lib.GVL_ONEHOT_LC)
        if out.size and len(ob):
            _lib.check(m.lib.gvl_reconstruct(C.byref(m.c), C.byref(bt), C.byref(oc), _stream_ptr()))
        out[...] = ffi._np(buf)


def shift_and_realign_tracks_from_svar2_into(
    out, out_offsets, regions, shifts, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen, dense_range, dense_present,
    dense_present_off, tracks, track_offsets, params, strategy_id=0, base_seed=0, query_seed=None, parallel=False,
):
    """The core (src/tracks/mod.rs:705-860), in place: rows at the caller's ``out_offsets``; ``query_seed`` (n_q,) maps the local
    query index to the batch row the FlankSample fill is seeded with (a logical batch that arrives in several calls)."""
    from . import ffi

    if not (isinstance(out, np.ndarray) and out.dtype == np.float32 and out.flags.c_contiguous):
        raise ValueError("`out` must be a C-contiguous float32 array")
    regions = np.ascontiguousarray(regions, np.int32)
    shifts = np.ascontiguousarray(shifts, np.int32)
    n_vk, n_dense = len(np.asarray(vk_pos).reshape(-1)), len(np.asarray(dense_pos).reshape(-1))
    dev = _no_ref_static()
    ch = _channels(vk_pos, vk_ilen, np.zeros(n_vk + 1, np.int64), vk_off, dense_pos, dense_ilen, np.zeros(n_dense + 1, np.int64),
                   dense_range, dense_present, dense_present_off, np.zeros(0, np.uint8), False, dev.device)
    reg0 = regions.copy()
    reg0[:, 0] = 0
    m = merge(dev, ch, reg0, shifts.shape[1])
    res = _device.realign_tracks(m, m.regions, shifts, m.geno_offset_idx, np.ascontiguousarray(out_offsets, np.int64), tracks,
                                 track_offsets, params, strategy_id, base_seed, query_seed=query_seed)
    out[...] = ffi._np(res)


def shift_and_realign_tracks_from_svar2(
    regions, shifts, vk_pos, vk_ilen, vk_off, dense_pos, dense_ilen, dense_range, dense_present, dense_present_off,
    tracks, track_offsets, params, strategy_id, base_seed, parallel=False,
):
    """The fused track entry (src/ffi/mod.rs:1835-1966) -> ``(out f32[total], out_offsets i64[K + 1])``: ragged rows of
    region length + ``hap_diffs_svar2``, realigned by the library's track kernels over the merged table."""
    from . import ffi

    regions = np.ascontiguousarray(regions, np.int32)
    shifts = np.ascontiguousarray(shifts, np.int32)
    n_vk, n_dense = len(np.asarray(vk_pos).reshape(-1)), len(np.asarray(dense_pos).reshape(-1))
    dev = _no_ref_static()
    ch = _channels(vk_pos, vk_ilen, np.zeros(n_vk + 1, np.int64), vk_off, dense_pos, dense_ilen, np.zeros(n_dense + 1, np.int64),
                   dense_range, dense_present, dense_present_off, np.zeros(0, np.uint8), False, dev.device)
    reg0 = regions.copy()
    reg0[:, 0] = 0
    m = merge(dev, ch, reg0, shifts.shape[1])
    with _on_device(m.device):
        bt = m.prepare_batch(m.regions, shifts, m.geno_offset_idx, -1)
        oo, tm, _ = m.hap_offsets(bt)
    res = _device.realign_tracks(m, m.regions, shifts, m.geno_offset_idx, oo, tracks, track_offsets, params, strategy_id, base_seed)
    return ffi._np(res), ffi._np(oo)


def split_to_flat(n_regions, ploidy, vk_pos, vk_key, vk_off, snp_pos, snp_key, snp_range, snp_present, snp_present_off,
                  indel_pos, indel_key, indel_range, indel_present, indel_present_off) -> dict:
    """``split_to_flat`` (src/svar2/mod.rs:176-274) with numpy: genoray's read-bound gather result (``BatchResultSplit``:
    var_key + per-class dense channels ``dense_snp`` / ``dense_indel``, each with its own per-haplotype presence bits) ->
    the flat single-dense-channel layout the SVAR2 entry points take.  Per query the window is its snp entries followed by
    its indel entries; per haplotype the presence bits are its snp bits followed by its indel bits, LSB-first in one stream.
    Ranges are (n_regions, 2); presence offsets are BIT offsets (batch*ploidy + 1)."""
    snp_range = np.asarray(snp_range, np.int64).reshape(n_regions, 2)
    indel_range = np.asarray(indel_range, np.int64).reshape(n_regions, 2)
    ws, wi = snp_range[:, 1] - snp_range[:, 0], indel_range[:, 1] - indel_range[:, 0]
    w = ws + wi
    d_off = np.zeros(n_regions + 1, np.int64)
    np.cumsum(w, out=d_off[1:])
    nd = int(d_off[-1])
    # window entry e of query q: snp entry snp_range[q, 0] + e for e < ws[q], else indel entry indel_range[q, 0] + e - ws[q]
    qi = np.repeat(np.arange(n_regions), w)
    e = np.arange(nd) - d_off[qi]
    is_snp = e < ws[qi]
    src = np.where(is_snp, snp_range[qi, 0] + e, indel_range[qi, 0] + e - ws[qi])
    sp, sk = np.asarray(snp_pos, np.int64).reshape(-1), np.asarray(snp_key, np.int64).reshape(-1)
    ip, ik = np.asarray(indel_pos, np.int64).reshape(-1), np.asarray(indel_key, np.int64).reshape(-1)

    def take(a_snp, a_indel):
        out = np.zeros(nd, np.int64)
        if is_snp.any():
            out[is_snp] = a_snp[src[is_snp]]
        if (~is_snp).any():
            out[~is_snp] = a_indel[src[~is_snp]]
        return out

    dense_pos, dense_key = take(sp, ip), take(sk, ik)
    # presence: haplotype h = q * ploidy + p, bit b of its window: snp bit snp_present_off[h] + b, or indel bit ... + b - ws[q]
    H = n_regions * int(ploidy)
    hq = np.repeat(np.arange(n_regions), ploidy)
    hw = w[hq]
    p_off = np.zeros(H + 1, np.int64)
    np.cumsum(hw, out=p_off[1:])
    nb = int(p_off[-1])
    hi = np.repeat(np.arange(H), hw)
    b = np.arange(nb) - p_off[hi]
    b_snp = b < ws[hq][hi]
    spo, ipo = np.asarray(snp_present_off, np.int64).reshape(-1), np.asarray(indel_present_off, np.int64).reshape(-1)
    sbits = np.unpackbits(np.asarray(snp_present, np.uint8).reshape(-1), bitorder="little")
    ibits = np.unpackbits(np.asarray(indel_present, np.uint8).reshape(-1), bitorder="little")
    bits = np.zeros(nb, np.uint8)
    if b_snp.any():
        bits[b_snp] = sbits[spo[hi[b_snp]] + b[b_snp]]
    if (~b_snp).any():
        bits[~b_snp] = ibits[ipo[hi[~b_snp]] + b[~b_snp] - ws[hq][hi[~b_snp]]]
    return dict(vk_pos=np.asarray(vk_pos, np.int32).reshape(-1),
                vk_key=np.asarray(vk_key, np.int64).reshape(-1).astype(np.uint32).view(np.int32),      # (keys are u32 bit patterns)
                vk_off=np.asarray(vk_off, np.int64).reshape(-1), dense_pos=dense_pos.astype(np.int32),
                dense_key=dense_key.astype(np.uint32).view(np.int32),
                dense_range=np.stack([d_off[:-1], d_off[1:]], axis=1).astype(np.int32).reshape(-1),
                dense_present=np.packbits(bits, bitorder="little"), dense_present_off=p_off)
