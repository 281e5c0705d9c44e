"""Device-side request prep + a ``to_dataloader``-shaped iterator.

In the reference a batch goes ``TorchDataset.__getitem__(idx)`` (``_torch.py:290-307``) ->
``np.unravel_index`` -> ``_getitem_unspliced`` (``_dataset/_query.py:153-204``: gather regions,
jitter, ``to_rc = strand == -1``) -> ``Haps._prepare_request`` (``_haps.py:678-754``:
``ravel_multi_index`` to genotype-offset slots, ``get_diffs_sparse``, random shifts,
offsets) -> the fused kernel -> host arrays -> user transform (one-hot) -> H2D copy.

Here the same steps run on the device with torch integer ops + the HIP kernels, the one-hot
is fused, nothing returns to the host, and ``DeviceLoader`` keeps several batches in flight on
separate HIP streams (the role of the reference's ``buffered`` / ``double_buffered`` modes,
``_torch.py:94-211``, without shared memory or a producer process).

Randomness: jitter and shifts are drawn inside ``gvl_prepare_request`` from a counter-based
hash keyed by (seed, batch counter, row); the *distribution* is the reference's
(``rng.integers(-j, j + 1)``, ``rng.integers(0, max_shift + 1)``), the stream of numbers is
not (numpy's PCG64 is not reproducible on a GPU).  ``deterministic=True`` with ``jitter=0``
involves no randomness and is bit-exact.
"""

from __future__ import annotations

from collections import deque
from dataclasses import dataclass

import os
import numpy as np
import torch

from .device import _on_device
from .sharding import epoch_order


def build_request(idx: torch.Tensor, full_regions: torch.Tensor, n_samples: int, ploidy: int,
                  jitter: int = 0, rc_neg: bool = True, generator: torch.Generator | None = None):
    """idx (b,) dataset indices over the (regions, samples) grid -> per-batch request tensors
    on ``idx.device``: regions (b, 4) i32 (jittered), geno_offset_idx (b, P) i64, to_rc (b*P,)
    u8 | None, lengths (b,) i64.  Pure torch: also runs on CPU tensors (tests)."""
    idx = idx.to(torch.int64)
    r_idx = torch.div(idx, n_samples, rounding_mode="floor")      # np.unravel_index(idx, (R, S))
    s_idx = idx - r_idx * n_samples
    regions = full_regions.index_select(0, r_idx).clone()          # _query.py:164-165 (copy)
    lengths = (regions[:, 2] - regions[:, 1]).to(torch.int64)
    if jitter > 0:                                                 # _query.py:166-171
        off = torch.randint(-jitter, jitter + 1, (idx.numel(),), device=idx.device, generator=generator,
                            dtype=torch.int64).to(torch.int32)
        regions[:, 1] += off
        regions[:, 2] = regions[:, 1] + lengths.to(torch.int32)
    to_rc = None
    if rc_neg:                                                     # _query.py:173-175 + _haps.py:838-843
        to_rc = (regions[:, 3] == -1).repeat_interleave(ploidy).to(torch.uint8)
    # _haps.py:757-768: ravel_multi_index((r, s, p), (R, S, P))
    p = torch.arange(ploidy, device=idx.device, dtype=torch.int64)
    goi = ((r_idx * n_samples + s_idx) * ploidy)[:, None] + p[None, :]
    return regions, goi.contiguous(), to_rc, lengths


@dataclass
class Batch:
    onehot: torch.Tensor | None        # (b, P, L, 4) u8  (or (b, P, 4, L) for layout "cl")
    haps: torch.Tensor | None          # (b, P, L) u8
    idx: torch.Tensor                  # (b,) dataset indices
    regions: torch.Tensor
    shifts: torch.Tensor
    geno_offset_idx: torch.Tensor
    to_rc: torch.Tensor | None
    out_offsets: torch.Tensor | None = None      # ragged mode: (b * P + 1,) i64; haps / onehot are then flat
    annot_v_idxs: torch.Tensor | None = None     # annotate=True: i32, same shape as haps
    annot_ref_pos: torch.Tensor | None = None
    sizes: torch.Tensor | None = None            # ragged batches of the native loop: device i64[2] = {total bases,
    #                                              longest row}; haps / onehot are then views of the slot's whole
    #                                              capacity, rows packed at out_offsets (no host read per batch)


class DeviceHapsDataset:
    """(regions x samples) grid over a :class:`HapsDevice`.

    ``output_length >= 1``: fixed-length rows (crop / pad).  ``output_length = -1``: RAGGED rows like the
    reference's default output (``_haps.py:794-811``: row length = region length + the haplotype's length
    delta): batches carry flat ``haps`` / ``onehot`` plus ``out_offsets``.  ``annotate=True`` adds the
    variant-index / reference-position annotations (``reconstruct_annotated_haplotypes_fused``).  All of
    them ride the native batch loop (ragged rows in ring slots of ``max_row_len()`` bases per row, sized on
    the device); a custom sampler or ``python_loop=True`` submits from Python instead (ragged: one host read
    of the batch's total size per batch, which the exactly-sized allocation forces).

    ``dev``'s genotype offsets must be laid out like the reference's sparse genotypes: slot
    ``ravel_multi_index((region, sample, ploid), (R, S, P))`` (``_haps.py:757-768``)."""

    def __init__(self, dev, regions, n_samples: int, ploidy: int, *, output_length: int, jitter: int = 0,
                 rc_neg: bool = True, deterministic: bool = True, seed: int | None = None, onehot: bool = True,
                 haps: bool = False, layout: str = "lc", annotate: bool = False):
        self.dev = dev
        d = dev.device
        reg = torch.as_tensor(np.ascontiguousarray(regions, np.int32)).to(d)
        if reg.dim() != 2 or reg.shape[1] < 4:
            raise ValueError("regions must be (n_regions, 4) int32 [contig, start, end, strand]")
        self.full_regions = reg
        self.n_regions, self.n_samples, self.ploidy = int(reg.shape[0]), int(n_samples), int(ploidy)
        if int(dev.geno_offsets.shape[1]) < self.n_regions * self.n_samples * self.ploidy:
            raise ValueError("genotype offsets do not cover regions x samples x ploidy")
        self.output_length = int(output_length)
        if self.output_length < 1 and self.output_length != -1:
            raise ValueError("output_length must be >= 1 (fixed-length output) or -1 (ragged)")
        self.ragged, self.annotate = self.output_length == -1, bool(annotate)
        self.jitter, self.rc_neg, self.deterministic = int(jitter), bool(rc_neg), bool(deterministic)
        if self.ragged and not self.deterministic:
            raise ValueError("random shifts crop a fixed-length window: ragged output is deterministic")
        self.onehot, self.haps, self.layout = bool(onehot), bool(haps) or self.annotate, layout
        if self.ragged and layout != "lc":
            raise ValueError("channel-major one-hot needs fixed-length rows")
        self.seed = (0 if seed is None else int(seed)) & 0xFFFFFFFFFFFFFFFF
        self._counter = 0          # random draws are keyed by (seed, counter, dataset index): one tick per request
        self._loaders = 0          # ... and every loader gets its own derived seed, so a new loader does not replay

    def max_row_len(self) -> int:
        """A true bound on a ragged row's length (what a ring slot of the native loop reserves per row):
        the longest region plus the largest sum of insertion lengths over the genotype slots."""
        m = getattr(self, "_max_row_len", None)
        if m is None:
            reg, dev = self.full_regions, self.dev
            longest = int((reg[:, 2] - reg[:, 1]).max().item()) if self.n_regions else 0
            grow = 0
            if int(dev.geno_v_idxs.numel()):
                gain = dev.ilens.index_select(0, dev.geno_v_idxs.to(torch.int64)).clamp_(min=0).to(torch.int64)
                c = torch.cat([gain.new_zeros(1), gain.cumsum(0)])
                go = dev.geno_offsets
                grow = int((c[go[1].clamp(0, gain.numel())] - c[go[0].clamp(0, gain.numel())]).max().item())
            m = self._max_row_len = max(longest + max(grow, 0), 1)
        return m

    def _loader_seed(self, draw_stream: int | None = None) -> int:
        """The seed of a loader's random draws: splitmix64 of the dataset seed and the loader's draw stream --
        by default the number of loaders created from this dataset so far (a new loader does not replay the
        previous one's draws; the k-th loader of every rank's process gets the same stream), or the caller's."""
        from .sharding import splitmix64

        if draw_stream is None:
            self._loaders += 1
            draw_stream = self._loaders
        return splitmix64((self.seed + int(draw_stream)) & 0xFFFFFFFFFFFFFFFF)

    def _draw_key(self) -> tuple[int, int]:
        """(seed, counter) of the next request's random draws: a draw is hash(seed, counter, dataset index).
        Stand-alone requests tick the dataset's own counter; a loader pins (its seed, epoch + 1) for the epoch
        (``_draw_override``), so that an index draws the same whatever batch, batch size or rank delivers it."""
        ov = getattr(self, "_draw_override", None)
        if ov is not None:
            return ov[0], ov[1]
        self._counter += 1
        return self.seed, self._counter

    @property
    def shape(self):
        return (self.n_regions, self.n_samples)

    def __len__(self):
        return self.n_regions * self.n_samples

    def request(self, idx):
        """Device-side ``_prepare_request`` (one launch of ``gvl_prepare_request``): everything
        the kernel needs, no host round trip.  :func:`build_request` states the same index
        math with torch ops (tests compare the two)."""
        import ctypes as C

        from . import _lib
        from .device import _ptr, _stream_ptr

        d = self.dev.device
        idx = torch.as_tensor(np.asarray(idx) if not isinstance(idx, torch.Tensor) else idx)
        idx = idx.to(device=d, dtype=torch.int64).reshape(-1).contiguous()
        b, P = int(idx.numel()), self.ploidy
        regions = torch.empty((b, 4), dtype=torch.int32, device=d)
        goi = torch.empty((b, P), dtype=torch.int64, device=d)
        to_rc = torch.empty(b * P, dtype=torch.uint8, device=d)
        shifts = torch.empty((b, P), dtype=torch.int32, device=d)
        d_seed, d_counter = self._draw_key()
        with _on_device(d):
            _lib.check(self.dev.lib.gvl_prepare_request(
                C.byref(self.dev.c), _ptr(idx), C.c_int64(b), _ptr(self.full_regions), C.c_int64(self.n_regions),
                C.c_int64(self.n_samples), C.c_int64(P), C.c_int64(self.jitter), C.c_int32(int(self.rc_neg)),
                C.c_int32(int(self.deterministic)), C.c_int64(max(self.output_length, 0)), C.c_uint64(d_seed),
                C.c_uint64(d_counter), _ptr(regions), _ptr(goi), _ptr(to_rc), _ptr(shifts), _stream_ptr()))
        return idx, regions, shifts, goi, (to_rc if self.rc_neg else None)

    def __getitem__(self, idx) -> Batch:
        """Two launches (request prep, reconstruct) on the current stream, one allocation: every
        per-batch array is carved out of a single arena so that the host does one allocator
        call per batch."""
        import ctypes as C

        from . import _lib
        from ._lib import GvlBatch, GvlOut

        d = self.dev.device
        if self.ragged or self.annotate:
            # the general path: request prep, then HapsDevice.reconstruct (ragged: sizes on the device,
            # one host read of {total, longest row} to allocate)
            idx_d, regions, shifts, goi, to_rc = self.request(idx)
            out = self.dev.reconstruct(regions, shifts, goi, self.output_length, to_rc=to_rc, haps=self.haps,
                                       onehot=self.onehot, layout=self.layout, annotate=self.annotate)
            b, P = int(idx_d.numel()), self.ploidy
            hp, oh, av, ap = out.haps, out.onehot, out.annot_v_idxs, out.annot_ref_pos
            if not self.ragged:
                L = self.output_length
                hp = None if hp is None else hp.view(b, P, L)
                av = None if av is None else av.view(b, P, L)
                ap = None if ap is None else ap.view(b, P, L)
                if oh is not None:
                    oh = oh.view(b, P, L, 4) if self.layout == "lc" else oh.view(b, P, 4, L)
            batch = Batch(oh, hp, idx_d, regions, shifts, goi, to_rc, out.out_offsets, av, ap)
            batch._arena = out.haps if out.haps is not None else out.onehot      # (record_stream target)
            batch._keep = out
            return batch
        idx = torch.as_tensor(np.asarray(idx) if not isinstance(idx, torch.Tensor) else idx)
        idx = idx.to(device=d, dtype=torch.int64).reshape(-1).contiguous()
        b, P, L = int(idx.numel()), self.ploidy, self.output_length
        K = b * P

        def up(x):
            return (x + 255) & ~255

        sizes = [4 * K * L if self.onehot else 0, K * L if self.haps else 0, 16 * b, 8 * K, 4 * K, K, 8 * (K + 1)]
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + up(sz))
        arena = torch.empty(offs[-1], dtype=torch.uint8, device=d)
        base = arena.data_ptr()
        p_oh, p_hp, p_reg, p_goi, p_sh, p_rc, p_oo = (base + o for o in offs[:-1])
        d_seed, d_counter = self._draw_key()
        lib = self.dev.lib
        stream = C.c_void_p(torch.cuda.current_stream(d).cuda_stream)
        with _on_device(d):
            _lib.check(lib.gvl_prepare_request(
                C.byref(self.dev.c), C.c_void_p(idx.data_ptr()), C.c_int64(b), C.c_void_p(self.full_regions.data_ptr()),
                C.c_int64(self.n_regions), C.c_int64(self.n_samples), C.c_int64(P), C.c_int64(self.jitter),
                C.c_int32(int(self.rc_neg)), C.c_int32(int(self.deterministic)), C.c_int64(L), C.c_uint64(d_seed),
                C.c_uint64(d_counter), C.c_void_p(p_reg), C.c_void_p(p_goi), C.c_void_p(p_rc), C.c_void_p(p_sh),
                stream))
            bt = GvlBatch(regions=p_reg, regions_stride=4, shifts=p_sh, geno_offset_idx=p_goi, batch=b, ploidy=P,
                          keep=None, keep_offsets=None, to_rc=p_rc if self.rc_neg else None, output_length=L,
                          out_offsets=None, max_row_len=L)
            oc = GvlOut(haps=p_hp if self.haps else None, onehot=p_oh if self.onehot else None,
                        onehot_layout=_lib.GVL_ONEHOT_LC if self.layout == "lc" else _lib.GVL_ONEHOT_CL,
                        annot_v_idxs=None, annot_ref_pos=None, out_offsets=p_oo)
            if b:
                _lib.check(lib.gvl_reconstruct(C.byref(self.dev.c), C.byref(bt), C.byref(oc), stream))

        def view(i, dtype, shape):
            n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            return arena[offs[i]:offs[i] + n].view(dtype).view(shape)

        oh = hp = None
        if self.onehot:
            oh = view(0, torch.uint8, (b, P, L, 4) if self.layout == "lc" else (b, P, 4, L))
        if self.haps:
            hp = view(1, torch.uint8, (b, P, L))
        batch = Batch(oh, hp, idx, view(2, torch.int32, (b, 4)), view(4, torch.int32, (b, P)),
                      view(3, torch.int64, (b, P)), view(5, torch.uint8, (K,)) if self.rc_neg else None)
        batch._arena = arena
        return batch

    def to_dataloader(self, batch_size: int = 1, shuffle: bool = False, sampler=None, drop_last: bool = False,
                      generator: torch.Generator | None = None, in_flight: int = 3, rank: int = 0,
                      world_size: int = 1, seed: int = 0, threaded: bool = False, group: int | None = None,
                      python_loop: bool = False, draw_stream: int | None = None, pad_to: int | None = None) -> "DeviceLoader":
        """``Dataset.to_dataloader`` (``_impl.py:1963-2072``) for device consumers.  With
        ``world_size > 1`` the epoch is sharded across ranks like ``DistributedSampler``
        (:func:`genvarloader_amd.sharding.epoch_order`): same permutation on every rank, disjoint
        strided shares, no collective.  ``group`` batches share one host call and one event pair (every
        event between two kernels of a stream costs the GPU a few microseconds of idle queue) and
        ``in_flight`` such groups are submitted ahead; ``group=None`` takes the largest of 16, 8, 4, 2, 1
        whose ring of ``(in_flight + 1) * group`` output slots stays under 4 GiB.  ``threaded=True``: a producer thread inside the library
        submits them (launches overlap the consumer's own host work per batch).

        Random draws (jitter, shifts; ``_query.py:160-187``, ``_haps.py:720-730``): the draw of dataset index *i*
        in epoch *e* (``set_epoch``; epochs count from 0 otherwise) is a function of (dataset seed, ``draw_stream``,
        *e*, *i*) only -- the same whatever rank, batch, batch size or submit loop delivers the index, different
        between epochs and between draw streams.  ``draw_stream=None``: the number of loaders created from this
        dataset so far (the k-th loader of every rank's process draws alike; a new loader does not replay).

        ``pad_to``: wrap the epoch order around to that many indices (``DistributedSampler``'s padding) -- for ranks whose
        datasets differ in size (sample-sharded genotypes, :func:`genvarloader_amd.sharding.shard_genotypes_by_sample`) and
        must run the same number of batches."""
        return DeviceLoader(self, batch_size, shuffle, sampler, drop_last, generator, in_flight, rank, world_size,
                            seed, threaded, group, python_loop=python_loop, draw_stream=draw_stream, pad_to=pad_to)


@dataclass
class TrackBatch(Batch):
    tracks: torch.Tensor | None = None     # (b, n_tracks, P, L) f32, realigned to each haplotype


class _EpochArrays:
    """The epoch's request arrays (the native loop's epoch table) and order, as whole-epoch tensors."""
    __slots__ = ("order", "regions", "goi", "shifts", "to_rc", "seeds", "bs", "P")


class RingBatch:
    """A batch of the native loop (ONE object per ring slot, handed out again when the slot comes round): the
    outputs are views of the slot (valid until the next iteration), the
    request arrays (``idx``, ``regions``, ``shifts``, ``geno_offset_idx``, ``to_rc``) are rows of the epoch table,
    sliced when they are asked for -- an epoch of 245 batches does not pay for 1 200 tensor views up front.
    Same attributes as :class:`Batch` / :class:`TrackBatch`."""
    __slots__ = ("onehot", "haps", "out_offsets", "annot_v_idxs", "annot_ref_pos", "sizes", "tracks", "_ev", "_i", "_b")

    def __init__(self, onehot, haps, out_offsets, annot_v_idxs, annot_ref_pos, sizes, tracks, ev, i, b):
        self.onehot, self.haps, self.out_offsets = onehot, haps, out_offsets
        self.annot_v_idxs, self.annot_ref_pos, self.sizes, self.tracks = annot_v_idxs, annot_ref_pos, sizes, tracks
        self._ev, self._i, self._b = ev, i, b

    def _rows(self, t, per=1):
        lo = self._i * self._ev.bs * per
        return t[lo:lo + self._b * per]

    @property
    def idx(self):
        return self._rows(self._ev.order)

    @property
    def regions(self):
        return self._rows(self._ev.regions)

    @property
    def shifts(self):
        return self._rows(self._ev.shifts)

    @property
    def geno_offset_idx(self):
        return self._rows(self._ev.goi)

    @property
    def to_rc(self):
        return None if self._ev.to_rc is None else self._rows(self._ev.to_rc, self._ev.P)

    @property
    def base_seed(self):
        """The FlankSample base seed of this batch's tracks: a device scalar (per-batch mode) or the pinned int."""
        s = self._ev.seeds
        return s[self._i] if isinstance(s, torch.Tensor) else s


class DeviceHapsTracksDataset(DeviceHapsDataset):
    """Haplotypes + realigned tracks (BASELINE config 4's shape), everything on the device.

    ``tracks``: ``{name: (itv_starts i32, itv_ends i32, itv_values f32, itv_offsets i64)}`` with one
    interval list per (region, sample), list index ``region * n_samples + sample`` -- the
    reference's per-track interval store (``RaggedIntervals``, ``_dataset/_tracks.py``).  A track may
    instead be a dict ``{starts, ends, values, offsets, fill=(strategy_id, param) | None, region_level=False}``:
    ``fill`` is the track's OWN insertion fill (the reference lowers one per track, ``_reconstruct.py:204-208``;
    ``None`` = the dataset's ``strategy_id`` / ``param``), ``region_level=True`` a track with one interval list per
    REGION, shared by its samples (``TrackType`` other than ``SAMPLE``: the reference indexes it by ``r_idx``,
    ``_reconstruct.py:231-236``).  Per
    batch and track: paint the query's intervals into a scratch track of the reference's length
    ``len - min_p(min(diff, 0))`` (``_reconstruct.py:191``; computed on the device), then realign it to every
    haplotype with the insertion-fill strategy (``intervals_and_realign_track_fused``,
    ``src/ffi/mod.rs:2551-2672``); negative-strand rows are reversed.  No host sync per batch.
    Batches own their memory (random access / custom samplers / ``DeviceLoader``'s Python loop)."""

    def __init__(self, dev, regions, n_samples, ploidy, *, tracks: dict, strategy_id: int = 0, param: float = 0.0,
                 base_seed: int | None = None, **kw):
        super().__init__(dev, regions, n_samples, ploidy, **kw)
        if self.ragged or self.annotate:
            raise ValueError("DeviceHapsTracksDataset: fixed-length rows, no annotations")
        from . import device as _device

        d = dev.device
        # base_seed of the seed-dependent fills (FlankSample): None = per batch like the reference
        # (_reconstruct.py:215-222: xor-reduce of the batch's dataset indices when deterministic, a
        # fresh draw otherwise); an int pins it.
        self.strategy_id, self.param = int(strategy_id), float(param)
        self.base_seed = None if base_seed is None else int(base_seed) & 0xFFFFFFFFFFFFFFFF
        self.track_names = list(tracks)
        self._itv, self._bkt = [], []
        n_lists = self.n_regions * self.n_samples
        self._track_opts = []
        for name in self.track_names:
            spec = tracks[name]
            fill, region_level = None, False
            if isinstance(spec, dict):      # a track with its own insertion fill and / or region-level lists
                fill, region_level = spec.get("fill"), bool(spec.get("region_level", False))
                a, b, v, io = spec["starts"], spec["ends"], spec["values"], spec["offsets"]
            else:
                a, b, v, io = spec
            n_lists = self.n_regions if region_level else self.n_regions * self.n_samples
            self._track_opts.append((None if fill is None else (int(fill[0]), float(fill[1])), region_level))
            a = torch.as_tensor(np.ascontiguousarray(a, np.int32)).to(d)
            b = torch.as_tensor(np.ascontiguousarray(b, np.int32)).to(d)
            v = torch.as_tensor(np.ascontiguousarray(v, np.float32)).to(d)
            io = torch.as_tensor(np.ascontiguousarray(io, np.int64)).to(d)
            if int(io.numel()) != n_lists + 1:
                raise ValueError(f"track {name!r}: itv_offsets must have " + ("regions" if region_level else "regions x samples") + " + 1 entries")
            pm = _device.intervals_prefix_max(b, io, d) if int(b.numel()) else None
            self._itv.append((a, b, v, io, pm))
            self._bkt.append(_device.intervals_bucket_index(a, pm, io, d) if pm is not None else None)
        # Does the tiled painter finish every chunk of this interval set on its own?  (BigWig-like tracks do: no overlaps,
        # distinct starts, never more than 256 intervals in two adjacent 2048-position buckets.)  Checked once, here; the
        # painter then skips its second launch (gvl_track_set.tile_complete) and reports a chunk that proves the claim wrong.
        self._tile_complete = [self._tiles_complete(a, b, io, bk) for (a, b, v, io, pm), bk in zip(self._itv, self._bkt)]
        from ._lib import GvlTrackSet

        self._track_sets = (GvlTrackSet * max(len(self._itv), 1))(*[
            GvlTrackSet(itv_starts=a.data_ptr(), itv_ends=b.data_ptr(), itv_values=v.data_ptr(), itv_offsets=io.data_ptr(),
                        n_intervals=int(a.numel()), itv_pmax_ends=None if pm is None else pm.data_ptr(),
                        bkt_offsets=None if bk is None else bk[0].data_ptr(), bkt_base=None if bk is None else bk[1].data_ptr(),
                        bkt_lo=None if bk is None else bk[2].data_ptr(), bkt_hi=None if bk is None else bk[3].data_ptr(),
                        tile_complete=int(tc), has_fill=int(fill is not None), fill_strategy=0 if fill is None else fill[0],
                        fill_param=0.0 if fill is None else fill[1], list_div=self.n_samples if region_level else 1)
            for (a, b, v, io, pm), bk, tc, (fill, region_level) in zip(self._itv, self._bkt, self._tile_complete, self._track_opts)])
        reg = self.full_regions
        max_len = int((reg[:, 2] - reg[:, 1]).max().item()) if self.n_regions else 0
        # scratch track per query: len - min(diff, 0) <= 2 * len (a window cannot lose more than itself)
        self._stride = 2 * max(max_len + 2 * self.jitter, 1)

    @staticmethod
    def _tiles_complete(a, b, io, bk) -> bool:
        """No two intervals of a list overlap or share a start, and no two adjacent buckets of the painter's index hold more
        than 256 intervals (one host read, once per interval set)."""
        n = int(a.numel())
        if bk is None or n == 0:
            return False
        bo, _, lo, hi = bk
        if n > 1:
            nxt_is_list_start = torch.zeros(n, dtype=torch.bool, device=a.device)
            starts = io[1:-1]
            nxt_is_list_start[starts[starts < n]] = True          # interval i begins a list
            adj = ~nxt_is_list_start[1:]                           # pairs (i, i + 1) inside one list
            bad = adj & ((a[1:] <= a[:-1]) | (a[1:] < b[:-1]))
            if bool(bad.any().item()):
                return False
        nb = int(lo.numel()) if int(bo[-1].item()) > 0 else 0
        if nb == 0:
            return True
        g = torch.arange(nb, device=a.device)
        last_of_list = torch.zeros(nb, dtype=torch.bool, device=a.device)
        ends = bo[1:] - 1
        last_of_list[ends[(ends >= 0) & (ends < nb)]] = True
        nxt = torch.where(last_of_list, g, (g + 1).clamp(max=nb - 1))
        cnt = hi[:nb].to(torch.int64)[nxt] - lo[:nb].to(torch.int64)
        return bool((cnt.max() <= 256).item())

    def __getitem__(self, idx) -> TrackBatch:
        import ctypes as C

        from . import _lib
        from ._lib import GvlBatch

        if self.base_seed is not None:
            seed = self.base_seed
        elif self.deterministic:
            if isinstance(idx, torch.Tensor) and idx.is_cuda:      # (one host read; host indices avoid it)
                seed = int(np.bitwise_xor.reduce(idx.detach().cpu().numpy().astype(np.uint64).reshape(-1)))
            else:
                seed = int(np.bitwise_xor.reduce(np.asarray(idx).astype(np.uint64).reshape(-1))) if np.size(idx) else 0
        else:
            from .sharding import splitmix64

            ov = getattr(self, "_draw_override", None)
            # a fresh base seed per BATCH (_reconstruct.py:215-222).  Under a loader: keyed by (draw seed, epoch + 1, batch number of
            # the epoch) exactly as the native loop's batch_seeds_kernel keys it, so both submit loops fill alike
            if ov is None:
                seed = splitmix64(self.seed ^ splitmix64(self._counter + 1))
            else:
                j = int(ov[2]) if len(ov) > 2 else 0
                seed = splitmix64(ov[0] ^ splitmix64(((int(ov[1]) << 32) + j) & 0xFFFFFFFFFFFFFFFF))
        base = super().__getitem__(idx)
        dev, d = self.dev, self.dev.device
        b, P, L = int(base.idx.numel()), self.ploidy, self.output_length
        if b == 0 or not self._itv:
            return TrackBatch(base.onehot, base.haps, base.idx, base.regions, base.shifts, base.geno_offset_idx,
                              base.to_rc, tracks=None)
        K, T = b * P, len(self._itv)
        # ONE native call for the track half (gvl_tracks_batch): scratch-track lengths, then paint +
        # realign per track; output and scratch share one arena, no torch op per batch
        lib = dev.lib
        n_scr = int(lib.gvl_tracks_scratch_bytes(C.c_int64(b), C.c_int64(P), C.c_int64(self._stride)))
        out_bytes = (4 * T * K * L + 255) & ~255
        arena = torch.empty(out_bytes + n_scr, dtype=torch.uint8, device=d)
        bt = GvlBatch(regions=base.regions.data_ptr(), regions_stride=4, shifts=base.shifts.data_ptr(),
                      geno_offset_idx=base.geno_offset_idx.data_ptr(), batch=b, ploidy=P, keep=None, keep_offsets=None,
                      to_rc=None if base.to_rc is None else base.to_rc.data_ptr(), output_length=L, out_offsets=None,
                      max_row_len=L)
        par = (C.c_double * 1)(self.param)
        with _on_device(d):
            _lib.check(lib.gvl_tracks_batch(
                C.byref(dev.c), C.byref(bt), C.c_void_p(base.idx.data_ptr()), self._track_sets, C.c_int32(T), par,
                C.c_int64(self.strategy_id), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_void_p(arena.data_ptr()),
                C.c_int64(K * L), C.c_void_p(arena.data_ptr() + out_bytes), C.c_int64(self._stride),
                C.c_void_p(torch.cuda.current_stream(d).cuda_stream)))
        tracks = arena[:4 * T * K * L].view(torch.float32).view(T, b, P, L).permute(1, 0, 2, 3)
        out = TrackBatch(base.onehot, base.haps, base.idx, base.regions, base.shifts, base.geno_offset_idx, base.to_rc,
                         tracks=tracks)
        out._arena = base._arena
        out.base_seed = seed
        out._keep = (arena,)
        return out

    def to_dataloader(self, batch_size: int = 1, shuffle: bool = False, sampler=None, drop_last: bool = False,
                      generator=None, in_flight: int = 3, rank: int = 0, world_size: int = 1, seed: int = 0,
                      threaded: bool = False, group: int = 1, python_loop: bool = False,
                      draw_stream: int | None = None, pad_to: int | None = None) -> "DeviceLoader":
        """The native ring carries the tracks too (``gvl_tracks_batch`` per batch into the slot, per-batch
        FlankSample seeds computed on the device); ``python_loop=True`` (or a custom sampler) submits every
        batch from Python instead and lets it own its memory.  ``draw_stream`` / ``pad_to`` as in
        :meth:`DeviceHapsDataset.to_dataloader`."""
        return DeviceLoader(self, batch_size, shuffle, sampler, drop_last, generator, in_flight, rank, world_size,
                            seed, threaded, group, python_loop=python_loop, draw_stream=draw_stream, pad_to=pad_to)


def splice_plan_device(lengths: torch.Tensor, pair_len: torch.Tensor):
    """``build_splice_plan`` (``_dataset/_splice.py:54-160``) with torch ops on ``lengths.device``.

    ``lengths`` (B, E) per-query lengths in (splice_row, sample, element) order, ``pair_len``
    (n_pairs,) elements per (row, sample) pair (``sum == B``).  Returns ``permutation`` (B * E: new
    position -> old k = query * E + e, i.e. (pair, e, element) order), ``permuted_out_offsets``
    (B * E + 1) and ``group_offsets`` (n_pairs * E + 1: one spliced sequence per (pair, e) cell).
    No host synchronisation."""
    d = lengths.device
    B, E = int(lengths.shape[0]), int(lengths.shape[1])
    pair_len = pair_len.to(device=d, dtype=torch.int64)
    n_pairs = int(pair_len.numel())
    start = torch.cumsum(pair_len, 0) - pair_len                                  # first element of each pair
    pair_of_q = torch.repeat_interleave(torch.arange(n_pairs, device=d), pair_len, output_size=B)
    i_local = torch.arange(B, device=d) - start[pair_of_q]
    # element q of pair p, inner cell e, goes to E * start[p] + e * len[p] + i  (:82-88)
    dest = (E * start[pair_of_q] + i_local)[:, None] + torch.arange(E, device=d)[None, :] * pair_len[pair_of_q][:, None]
    perm = torch.empty(B * E, dtype=torch.int64, device=d)
    perm[dest.reshape(-1)] = torch.arange(B * E, device=d)
    plen = lengths.reshape(-1).to(torch.int64)[perm]
    out_offsets = torch.zeros(B * E + 1, dtype=torch.int64, device=d)
    torch.cumsum(plen, 0, out=out_offsets[1:])
    cells = torch.zeros(n_pairs * E + 1, dtype=torch.int64, device=d)
    torch.cumsum(pair_len.repeat_interleave(E), 0, out=cells[1:])                 # :137-151
    return perm, out_offsets, out_offsets[cells]


@dataclass
class SplicedBatch:
    haps: torch.Tensor | None           # flat u8: cell (pair, ploid) = [group_offsets[c], group_offsets[c + 1])
    onehot: torch.Tensor | None         # flat (total, 4) u8
    pairs: torch.Tensor                 # (n_pairs,) indices over the (splice rows x samples) grid
    idx: torch.Tensor                   # (B,) dataset indices of the elements, (pair, element) order
    group_offsets: torch.Tensor         # (n_pairs * P + 1,) i64: one spliced haplotype per (pair, ploid)
    out_offsets: torch.Tensor           # (B * P + 1,) i64: the elements inside them, (pair, ploid, element) order
    permutation: torch.Tensor           # (B * P,) new row -> old k = element * P + ploid
    annot_v_idxs: torch.Tensor | None = None
    annot_ref_pos: torch.Tensor | None = None


class DeviceSplicedHapsDataset(DeviceHapsDataset):
    """Spliced haplotypes (``Dataset`` with a splice map, ``_dataset/_query.py:207-313``): an item is a
    (splice row, sample) pair, its value the concatenation of the row's elements (exons) per haplotype.

    ``splice_offsets`` (n_rows + 1) / ``splice_region_idx``: the splice map (``SpliceMap.splice_map``:
    row -> ordered region indices).  Per batch, on the device: element dataset indices, request prep,
    per-element haplotype lengths (``haplotype_lengths_for_plan``, ``_haps.py:536-569``; with
    ``exonic=True`` under the ``choose_exonic_variants`` keep mask), the splice plan
    (:func:`splice_plan_device`), then ONE ploidy-1 launch over the permuted elements that writes every
    element where it belongs in its spliced haplotype (``reconstruct_haplotypes_spliced_fused``,
    ``ffi/mod.rs:1981-2076``); negative-strand elements are reverse-complemented in place when
    ``rc_neg``.  Like the reference: ragged only, deterministic, no jitter (``_query.py:225-226``).
    One host read per batch (the output's total size, for the exactly-sized allocation); the keep mask is sized by a bound."""

    def __init__(self, dev, regions, n_samples, ploidy, *, splice_offsets, splice_region_idx, rc_neg: bool = True,
                 onehot: bool = False, haps: bool = True, annotate: bool = False, exonic: bool = False):
        super().__init__(dev, regions, n_samples, ploidy, output_length=-1, jitter=0, rc_neg=rc_neg, deterministic=True,
                         onehot=onehot, haps=haps, annotate=annotate)
        so = np.ascontiguousarray(splice_offsets, np.int64)
        sr = np.ascontiguousarray(splice_region_idx, np.int64)
        if so.ndim != 1 or so.size < 1 or so[0] != 0 or np.any(np.diff(so) < 0) or so[-1] != sr.size:
            raise ValueError("splice_offsets must be a non-decreasing (n_rows + 1,) array ending at len(splice_region_idx)")
        if sr.size and (sr.min() < 0 or sr.max() >= self.n_regions):
            raise ValueError("splice_region_idx out of range")
        self.n_rows = int(so.size - 1)
        self._so_host, self._len_host, self._sr_host = so, np.diff(so), sr
        d = dev.device
        self._so, self._sr = torch.as_tensor(so).to(d), torch.as_tensor(sr).to(d)
        self.exonic = bool(exonic)
        self.host_index_max = 8192          # batches of up to this many elements: index arithmetic on the host (see __getitem__)

    @property
    def shape(self):
        return (self.n_rows, self.n_samples)

    def __len__(self):
        return self.n_rows * self.n_samples

    def _max_slot_variants(self) -> int:
        """The most variants any genotype slot of the dataset holds (one device reduction + host read, once per dataset)."""
        if getattr(self, "_max_nv", None) is None:
            go = self.dev.geno_offsets
            self._max_nv = int((go[1] - go[0]).max().item()) if go[0].numel() else 0
        return self._max_nv

    def __getitem__(self, pairs) -> SplicedBatch:
        d, P, S = self.dev.device, self.ploidy, self.n_samples
        if isinstance(pairs, torch.Tensor):
            pairs_h = pairs.detach().cpu().numpy()          # (host indices avoid this read)
        else:
            pairs_h = np.asarray(pairs)
        pairs_h = pairs_h.astype(np.int64).reshape(-1)
        if pairs_h.size and (pairs_h.min() < 0 or pairs_h.max() >= len(self)):
            raise IndexError("pair index out of range")
        # The batch's index arithmetic -- element dataset indices, the splice plan's permutation (``build_splice_plan``,
        # ``_dataset/_splice.py:82-88``: element i of pair p, ploid e goes to P * start[p] + e * len[p] + i; it depends on the pairs' element
        # counts only, not on any length), the cells' bounds.  Batches of up to 8192 elements: numpy on the HOST (tens of microseconds) and
        # ONE upload -- as torch device ops it is twenty launches of a few hundred threads each, a quarter of a 256-pair batch's 0.7 ms
        # (round 6); larger batches: the device ops (a 4 096-pair batch: 1.05 against 1.33 ms).
        row_h, smp_h = pairs_h // S, pairs_h % S
        n_pairs = int(pairs_h.size)
        pair_len_h = self._len_host[row_h].astype(np.int64)
        B = int(pair_len_h.sum())                           # elements of the batch: known on the host, no sync
        if B <= self.host_index_max:
            start_h = np.cumsum(pair_len_h) - pair_len_h
            pair_of_q = np.repeat(np.arange(n_pairs, dtype=np.int64), pair_len_h)
            i_local = np.arange(B, dtype=np.int64) - start_h[pair_of_q]
            r_idx_h = self._sr_host[self._so_host[row_h[pair_of_q]] + i_local] if B else np.zeros(0, np.int64)
            ds_idx_h = r_idx_h * S + smp_h[pair_of_q]
            dest = (P * start_h[pair_of_q] + i_local)[:, None] + np.arange(P, dtype=np.int64)[None, :] * pair_len_h[pair_of_q][:, None]
            perm_h = np.empty(B * P, np.int64)
            perm_h[dest.reshape(-1)] = np.arange(B * P, dtype=np.int64)
            cells_h = np.zeros(n_pairs * P + 1, np.int64)
            np.cumsum(np.repeat(pair_len_h, P), out=cells_h[1:])
            packed = torch.from_numpy(np.concatenate([pairs_h, ds_idx_h, perm_h, perm_h // P, cells_h])).to(d)
            o1, o2, o3, o4 = n_pairs, n_pairs + B, n_pairs + B + B * P, n_pairs + B + 2 * B * P
            pairs_d, ds_idx, perm, q_of, cells_idx = packed[:o1], packed[o1:o2], packed[o2:o3], packed[o3:o4], packed[o4:]
        else:
            pairs_d = torch.as_tensor(pairs_h).to(d)
            row, smp = pairs_d // S, pairs_d % S
            pair_len = (self._so[row + 1] - self._so[row])
            start = torch.cumsum(pair_len, 0) - pair_len
            pair_of_q = torch.repeat_interleave(torch.arange(n_pairs, device=d), pair_len, output_size=B)
            i_local = torch.arange(B, device=d) - start[pair_of_q]
            ds_idx = self._sr[self._so[row[pair_of_q]] + i_local] * S + smp[pair_of_q]
            perm, _, _ = splice_plan_device(torch.zeros((B, P), dtype=torch.int32, device=d), pair_len)
            q_of = torch.div(perm, P, rounding_mode="floor")
            cells_idx = torch.zeros(n_pairs * P + 1, dtype=torch.int64, device=d)
            torch.cumsum(pair_len.repeat_interleave(P), 0, out=cells_idx[1:])
        _, regions, shifts, goi, to_rc = self.request(ds_idx)
        # per-element lengths -> plan.  The lengths come out of the ragged sizing of the permuted launch
        # itself (region length + length delta, under the keep mask when exonic), so the plan only needs
        # the permutation first: its offsets are the launch's own out_offsets.
        regions_p = regions.index_select(0, q_of).contiguous()
        goi_p = goi.reshape(-1).index_select(0, perm).view(-1, 1).contiguous()
        shifts_p = torch.zeros((B * P, 1), dtype=torch.int32, device=d)
        to_rc_p = None if to_rc is None else to_rc.index_select(0, perm).contiguous()
        keep = keep_offsets = None
        if self.exonic and B:
            # (no host read for the mask's size: a row has at most the dataset's largest genotype slot's variants)
            mx = self._max_slot_variants()             # (... unless an outlier slot makes that bound absurd: then the exact size, one read)
            keep, keep_offsets = self.dev.choose_exonic_variants(regions_p[:, 1].contiguous(), regions_p[:, 2].contiguous(), goi_p,
                                                                 max_per_row=mx if B * P * mx <= (64 << 20) else None)
        out = self.dev.reconstruct(regions_p, shifts_p, goi_p, -1, keep, keep_offsets, to_rc=to_rc_p, haps=self.haps,
                                   onehot=self.onehot, annotate=self.annotate)
        batch = SplicedBatch(out.haps, out.onehot, pairs_d, ds_idx, out.out_offsets[cells_idx], out.out_offsets, perm,
                             out.annot_v_idxs, out.annot_ref_pos)
        batch._arena = out.haps if out.haps is not None else out.onehot
        batch._keep = out
        return batch

    def to_dataloader(self, batch_size: int = 1, shuffle: bool = False, sampler=None, drop_last: bool = False,
                      generator=None, in_flight: int = 2, rank: int = 0, world_size: int = 1, seed: int = 0) -> "DeviceLoader":
        """Spliced batches are submitted from Python (sizes differ per batch; each owns its memory); the
        epoch order over the (rows x samples) pairs is drawn and sharded like the native loop's."""
        return DeviceLoader(self, batch_size, shuffle, sampler, drop_last, generator, in_flight, rank, world_size,
                            seed, False, 1, python_loop=True)


class DeviceLoader:
    """Iterates batches of a :class:`DeviceHapsDataset`, ``in_flight`` batches ahead, each on
    its own HIP stream; the consumer's current stream waits on the batch's event.

    Without a custom ``sampler`` the batch loop is native (``gvl_loader_*`` in
    ``libgvl_hip.so``): the epoch order goes to the device once, each ``next()`` is one C call
    that releases the previous batch, tops the pipeline up (request prep + reconstruct per
    batch on the loader's streams) and makes the current stream wait for the next batch.  The
    yielded tensors are views of a ring slot: **valid until the next iteration** (the
    contract of the reference's ``double_buffered`` mode, ``_double_buffered_loader.py``).
    With a ``sampler`` (arbitrary index lists) every batch is submitted from Python and owns
    its memory."""

    def __init__(self, ds: DeviceHapsDataset, batch_size=1, shuffle=False, sampler=None, drop_last=False,
                 generator=None, in_flight=2, rank=0, world_size=1, seed=0, threaded=False, group=None,
                 python_loop=False, draw_stream=None, pad_to=None):
        self.pad_to = None if pad_to is None else int(pad_to)
        self.draw_seed = ds._loader_seed(draw_stream)      # (seed of this loader's jitter / shift draws)
        self.threaded = bool(threaded)
        self.group = None if group is None else max(1, min(16, int(group)))
        self.python_loop = bool(python_loop)
        self.ds, self.batch_size, self.shuffle, self.drop_last = ds, int(batch_size), shuffle, drop_last
        self.sampler, self.generator = sampler, generator
        self.rank, self.world_size, self.seed, self.epoch = int(rank), int(world_size), int(seed), 0
        if not (0 <= self.rank < self.world_size):
            raise ValueError("rank out of range")
        if sampler is not None and self.world_size > 1:
            raise ValueError("a custom sampler does its own sharding: pass world_size=1")
        self.in_flight = max(1, min(16, int(in_flight)))
        # native loop: an epoch whose order depends on (seed, epoch number) only is prepared while the epoch before it runs
        # (gvl_loader_prefetch_epoch), so that consecutive epochs leave no gap on the GPU; False: every epoch is prepared at its start
        self.prefetch_epochs = True
        self._native = None
        self.streams = None

    # ---- native loop -------------------------------------------------------------------
    def _native_setup(self):
        import ctypes as C

        from . import _lib
        from ._lib import GvlLoaderBatch, GvlLoaderConfig

        ds, d = self.ds, self.ds.dev.device
        lib = ds.dev.lib
        if self.group is None:
            self.group = 1
            for g in (16, 8, 4, 2):              # the largest group whose ring stays under 4 GiB (and 64 slots)
                if (self.in_flight + 1) * g <= 64 and (self.in_flight + 1) * g * self._slot_bytes(g) <= (4 << 30):
                    self.group = g
                    break
        # slot sets: one per group in flight + the one the consumer holds; GVL_LOADER_EXTRA_SETS adds spare ones (a set's release
        # -- consumer stream -> submit stream, two event hops -- then is off the critical path of the group that reuses it)
        extra = max(1, int(os.environ.get("GVL_LOADER_EXTRA_SETS", "1")))
        while extra > 1 and (self.in_flight + extra) * self.group > 64:
            extra -= 1
        n_slots = (self.in_flight + extra) * self.group
        cfg = GvlLoaderConfig(
            full_regions=ds.full_regions.data_ptr(), n_regions=ds.n_regions, n_samples=ds.n_samples, ploidy=ds.ploidy,
            batch_size=self.batch_size, output_length=ds.output_length, jitter=ds.jitter, rc_neg=int(ds.rc_neg),
            deterministic=int(ds.deterministic), seed=self.draw_seed, want_haps=int(ds.haps), want_onehot=int(ds.onehot),
            onehot_layout=_lib.GVL_ONEHOT_LC if ds.layout == "lc" else _lib.GVL_ONEHOT_CL, in_flight=self.in_flight,
            n_slots=n_slots, slot_arenas=None, threaded=int(self.threaded), group=self.group,
            want_annot=int(ds.annotate), max_row_len=ds.max_row_len() if ds.ragged else 0)
        track_sets = getattr(ds, "_track_sets", None)
        if track_sets is not None and len(getattr(ds, "_itv", ())):
            cfg.tracks = C.cast(track_sets, C.c_void_p)
            cfg.n_tracks = len(ds._itv)
            cfg.track_seed_mode = 1 if ds.base_seed is None else 0
            cfg.strategy_id, cfg.track_param = ds.strategy_id, ds.param
            cfg.track_seed = ds.base_seed or 0
            cfg.scratch_stride = ds._stride
        parts = (C.c_int64 * _lib.LOADER_SLOT_PARTS)()
        nbytes = int(lib.gvl_loader_slot_bytes(C.byref(cfg), parts))
        if nbytes <= 0:
            raise ValueError("bad loader configuration")
        arenas = [torch.empty(nbytes, dtype=torch.uint8, device=d) for _ in range(n_slots)]
        ptrs = (C.c_void_p * n_slots)(*[a.data_ptr() for a in arenas])
        cfg.slot_arenas = C.cast(ptrs, C.POINTER(C.c_void_p))
        handle = C.c_void_p()
        with _on_device(d):
            _lib.check(lib.gvl_loader_create(C.byref(ds.dev.c), C.byref(cfg), C.byref(handle)))
        self._native = dict(handle=handle, arenas=arenas, parts=[int(x) for x in parts], cfg=cfg, ptrs=ptrs,
                            out=GvlLoaderBatch(), views={})

    def _slot_bytes(self, group: int) -> int:
        """Bytes of one ring slot for this dataset (gvl_loader_slot_bytes on a probe configuration)."""
        import ctypes as C

        from . import _lib
        from ._lib import GvlLoaderConfig

        ds = self.ds
        cfg = GvlLoaderConfig(n_regions=ds.n_regions, n_samples=ds.n_samples, ploidy=ds.ploidy, batch_size=self.batch_size,
                              output_length=ds.output_length, want_haps=int(ds.haps), want_onehot=int(ds.onehot),
                              onehot_layout=_lib.GVL_ONEHOT_LC if ds.layout == "lc" else _lib.GVL_ONEHOT_CL,
                              want_annot=int(ds.annotate), max_row_len=ds.max_row_len() if ds.ragged else 0, group=group)
        if len(getattr(ds, "_itv", ())):
            cfg.n_tracks, cfg.scratch_stride = len(ds._itv), ds._stride
        parts = (C.c_int64 * _lib.LOADER_SLOT_PARTS)()
        return max(1, int(ds.dev.lib.gvl_loader_slot_bytes(C.byref(cfg), parts)))

    def _slot_views(self, slot: int, b: int) -> Batch:
        nat, ds = self._native, self.ds
        key = (slot, b)
        v = nat["views"].get(key)
        if v is not None:
            return v
        arena, parts = nat["arenas"][slot], nat["parts"]
        P, L = ds.ploidy, ds.output_length
        K = b * P

        def view(i, dtype, shape):
            n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            return arena[parts[i]:parts[i] + n].view(dtype).view(shape)

        oo = av = ap = tr = sz = None
        if ds.ragged:
            # rows packed at out_offsets inside the slot's capacity (K * max_row_len bases)
            cap = K * int(nat["cfg"].max_row_len)
            oh = view(0, torch.uint8, (cap, 4)) if ds.onehot else None
            hp = view(1, torch.uint8, (cap,)) if ds.haps else None
            oo, sz = view(6, torch.int64, (K + 1,)), view(11, torch.int64, (2,))
            if ds.annotate:
                av, ap = view(7, torch.int32, (cap,)), view(8, torch.int32, (cap,))
        else:
            oh = view(0, torch.uint8, (b, P, L, 4) if ds.layout == "lc" else (b, P, 4, L)) if ds.onehot else None
            hp = view(1, torch.uint8, (b, P, L)) if ds.haps else None
            if ds.annotate:
                av, ap = view(7, torch.int32, (b, P, L)), view(8, torch.int32, (b, P, L))
            T = int(nat["cfg"].n_tracks)
            if T:       # track t of a short last batch still starts at t * batch_size * P * L
                full = view(9, torch.float32, (T, self.batch_size * P, L))
                tr = full[:, :K].view(T, b, P, L).permute(1, 0, 2, 3)
        v = (oh, hp, oo, av, ap, tr, sz)   # the request arrays of a batch are rows of the epoch table (see _epoch_table)
        nat["views"][key] = v
        return v

    def _epoch_table(self, n: int, which: int = 0):
        """The epoch's request arrays (filled by gvl_loader_start_epoch / gvl_loader_prefetch_epoch): one of two
        grow-only device buffers (epochs alternate, so that the next epoch's table can be filled while this
        epoch's batches still read theirs), typed views of its parts."""
        import ctypes as C

        nat, ds, d = self._native, self.ds, self.ds.dev.device
        lib = ds.dev.lib
        from . import _lib

        po = (C.c_int64 * _lib.LOADER_TABLE_PARTS)()
        nbytes = int(lib.gvl_loader_table_bytes(C.byref(nat["cfg"]), C.c_int64(n), po))
        tabs = nat.setdefault("tables", [None, None])
        tab = tabs[which]
        if tab is None or tab.numel() < nbytes:
            # (a table that is replaced by a larger one stays alive until the NEXT replacement: batches of the epoch that used it may
            # still be in flight on the loader's own streams, which torch's allocator knows nothing about)
            nat["table_replaced"] = tab
            tab = tabs[which] = torch.empty(nbytes + nbytes // 4, dtype=torch.uint8, device=d)
        # the typed views are the same every epoch (same buffer, same layout): made once -- a dozen tensor views are 35 us of host
        # time, twice per epoch boundary, and an epoch of BASELINE config 4's bench dataset is 8 batches
        ck = (which, n, nbytes, tuple(po), tab.data_ptr())
        hit = nat.setdefault("table_views", {}).get(which)
        if hit is not None and hit[0] == ck:
            proto = hit[1]
        else:
            P, bs = ds.ploidy, self.batch_size

            def part(i, dtype, shape):
                nb = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
                return tab[int(po[i]):int(po[i]) + nb].view(dtype).view(shape)

            reg, goi = part(0, torch.int32, (n, 4)), part(1, torch.int64, (n, P))
            sh, rc = part(2, torch.int32, (n, P)), part(3, torch.uint8, (n * P,))
            seeds = None
            if int(nat["cfg"].n_tracks) and int(nat["cfg"].track_seed_mode) == 1:
                seeds = part(4, torch.int64, (-(-n // bs),))      # (u64 bit patterns; one device scalar per batch)
            proto = _EpochArrays()
            proto.regions, proto.goi, proto.shifts, proto.to_rc, proto.seeds = reg, goi, sh, (rc if ds.rc_neg else None), seeds
            proto.bs, proto.P = bs, P
            proto.order = None
            nat["table_views"][which] = (ck, proto)
        ev = _EpochArrays()            # (a fresh one per epoch: the caller gives it the epoch's order)
        ev.regions, ev.goi, ev.shifts, ev.to_rc, ev.seeds = proto.regions, proto.goi, proto.shifts, proto.to_rc, proto.seeds
        ev.bs, ev.P = proto.bs, proto.P
        return tab, ev

    def _iter_native(self):
        import ctypes as C

        from . import _lib

        if self._native is None:
            self._native_setup()
        nat, ds, d = self._native, self.ds, self.ds.dev.device
        lib, handle, out = ds.dev.lib, nat["handle"], nat["out"]
        with _on_device(d):
            cur = torch.cuda.current_stream(d)
            g = self.generator

            def make_order(epoch):
                if g is None and self.shuffle and "order_gen" not in nat:
                    nat["order_gen"] = torch.Generator(device=d)        # (re-seeded per epoch: making one is 20 us)
                return self._padded(epoch_order(len(ds), shuffle=self.shuffle, seed=self.seed, epoch=epoch, rank=self.rank,
                                                world=self.world_size, drop_last=self.drop_last and self.world_size > 1,
                                                device=d, generator=g, scratch_generator=nat.get("order_gen")))

            # an epoch whose order is a pure function of (seed, epoch number) is prepared one epoch ahead
            # (gvl_loader_prefetch_epoch): what this epoch needs may already be there
            pure = g is None or not self.shuffle or self.world_size > 1
            ekey = (self.epoch, self.seed, self.shuffle, self.rank, self.world_size, self.drop_last, self.pad_to, len(ds),
                   self.batch_size)
            pf = nat.pop("prefetch", None)
            if pf is not None and pure and pf[0] == ekey:
                _, order, which = pf
            elif self.world_size == 1 and self.shuffle and g is not None and g.device.type != "cuda":
                # host generator: shuffle into a persistent pinned buffer and copy asynchronously
                pin = nat.get("pinned")
                if pin is None or pin.numel() != len(ds):
                    pin = nat["pinned"] = torch.empty(len(ds), dtype=torch.int64).pin_memory()
                torch.randperm(len(ds), generator=g, out=pin)
                order = torch.empty(len(ds), dtype=torch.int64, device=d)
                order.copy_(pin, non_blocking=True)
                order = self._padded(order)
                which = 1 - nat.get("which", 1)
            else:                                  # shuffle on the device: no H2D of the order
                order = make_order(self.epoch)
                which = 1 - nat.get("which", 1)
            _lib.check(lib.gvl_loader_set_epoch(handle, C.c_uint64(self.epoch & 0xFFFFFFFFFFFFFFFF)))
            self.epoch += 1
            n = int(order.numel())
            tab, ev = self._epoch_table(n, which)
            ev.order = order
            if ev.seeds is None:
                ev.seeds = getattr(ds, "base_seed", None)
            _lib.check(lib.gvl_loader_start_epoch(handle, C.c_void_p(order.data_ptr()), C.c_int64(n),
                                                  C.c_int32(int(self.drop_last)), C.c_void_p(tab.data_ptr()),
                                                  C.c_void_p(cur.cuda_stream)))
            nat["order_prev"] = nat.get("order")       # (the epoch before may still have batches in flight that read theirs)
            nat["order"] = order                       # keep the epoch order alive
            nat["which"] = which
            next_epoch = self.epoch

            def prefetch_next():
                # the NEXT epoch's order and table, queued early in this epoch (right behind its first submits, so
                # that those are not held up by the host work here): the epoch boundary then costs nothing -- the
                # next epoch's first batches queue right behind this epoch's last
                nkey = (next_epoch,) + ekey[1:]
                # on the consumer's LIVE current stream (read now, not the one captured when the epoch started: the consumer may
                # have switched streams since): the order's randperm and the table fill are then ordered on ONE stream, and that
                # stream is the one gvl_loader_next sees in the same iteration
                live = torch.cuda.current_stream(d)
                # ... but not ON that stream: the consumer's stream carries the ring's release chain (wait for batch j, release the
                # slot of batch j - 1, ...), and a dozen small kernels in the middle of it hold every release behind them up (BASELINE
                # config 4, 8 batches per epoch: the ring stood still for ~100 us per epoch).  A side stream, ordered behind the point
                # the consumer's stream has reached -- behind this epoch's first batch, hence behind every batch of the epoch that
                # last used the other table -- fills it; the next epoch's batches wait for the fill through the loader's own event.
                side = nat.get("side_stream")
                if side is None:
                    side = nat["side_stream"] = torch.cuda.Stream(device=d)
                mark = nat.get("side_mark")
                if mark is None:
                    mark = nat["side_mark"] = torch.cuda.Event()
                if os.environ.get("GVL_PREFETCH_ON_CONSUMER_STREAM"):
                    side = live
                else:
                    mark.record(live)
                    side.wait_event(mark)
                with torch.cuda.stream(side):
                    norder = make_order(next_epoch)
                    nn = int(norder.numel())
                    ntab, _ = self._epoch_table(nn, 1 - which)
                    _lib.check(lib.gvl_loader_prefetch_epoch(handle, C.c_uint64(next_epoch & 0xFFFFFFFFFFFFFFFF),
                                                             C.c_void_p(norder.data_ptr()), C.c_int64(nn), C.c_int32(int(self.drop_last)),
                                                             C.c_void_p(ntab.data_ptr()), C.c_void_p(side.cuda_stream)))
                # allocated under the side stream, read by the consumer's stream (and by the loader's submit streams, which the
                # consumer's stream is ordered behind): tell the caching allocator, or a block freed later could be handed out again
                # on the side stream while those reads are still queued
                if side is not live:
                    norder.record_stream(live)
                    ntab.record_stream(live)
                nat["prefetch"] = (nkey, norder, 1 - which)
            nxt, ref_out, bs = lib.gvl_loader_next, C.byref(out), self.batch_size
            nxt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]          # plain ints in, no wrapper objects per call
            # the consumer's CURRENT stream, read every iteration (it may change), through the raw getter
            raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
            dev_i = d.index if d.index is not None else torch.cuda.current_device()
            if raw_stream is None:
                raw_stream = lambda i_: torch.cuda.current_stream(i_).cuda_stream
            views, h = self._slot_views, handle.value
            ring = nat.setdefault("ring_batches", {})
            i = 0
            # the next epoch is prepared behind this epoch's SECOND batch: the step that starts an epoch already pays for the epoch's
            # set-up and the ring's first `in_flight` submits, and while it runs the GPU has only the last epoch's tail to work on --
            # one step later the ring is full (measured on config 4's 8-batch epochs: one 250 us host step per epoch against 160 us
            # of queued work; two steps of 110 + 100 us leave no gap)
            pf_at = 1 if (n // bs if self.drop_last else -(-n // bs)) > 1 else 0
            while True:
                rc = nxt(h, raw_stream(dev_i), ref_out)
                if rc:
                    _lib.check(rc)
                if out.slot < 0:
                    # errors only a kernel can see (a ragged row longer than the slot bound): the flag is sticky and
                    # process-global, so polling it here reports anything the batches consumed so far have raised
                    _lib.check_async()
                    return
                if i == pf_at and pure and self.prefetch_epochs:
                    prefetch_next()
                key = (out.slot, out.batch)
                batch = ring.get(key)
                if batch is None:            # one object per (slot, size): its outputs never change, its rows do
                    oh, hp, oo, av, ap, tr, sz = views(*key)
                    batch = ring[key] = RingBatch(oh, hp, oo, av, ap, sz, tr, ev, i, key[1])
                batch._ev, batch._i = ev, i
                i += 1
                yield batch

    def __del__(self):
        nat = getattr(self, "_native", None)
        if nat is not None:
            try:
                self.ds.dev.lib.gvl_loader_destroy(nat["handle"])
            except Exception:
                pass
            self._native = None

    # ---- Python loop (custom samplers) ---------------------------------------------------
    def _index_batches(self):
        if self.sampler is not None:
            for b in self.sampler:                  # a BatchSampler-like iterable of index lists
                yield np.asarray(b, dtype=np.int64).reshape(-1)
            return
        # no sampler: this rank's share of a fresh permutation, every epoch (seed + epoch)
        order = epoch_order(len(self.ds), shuffle=self.shuffle, seed=self.seed, epoch=self.epoch, rank=self.rank,
                            world=self.world_size, drop_last=self.drop_last and self.world_size > 1,
                            device="cpu", generator=self.generator)
        order = self._padded(order).numpy()                                             # host indices: no sync per batch
        self.epoch += 1
        bs, n = self.batch_size, int(order.size)
        for s in range(0, n, bs):
            if self.drop_last and s + bs > n:
                break
            yield order[s:s + bs]

    def _padded(self, order):
        """``pad_to``: the order wrapped around to that length (never shortened)."""
        n = int(order.numel())
        if self.pad_to is None or n == 0 or n >= self.pad_to:
            return order
        reps = -(-self.pad_to // n)
        return torch.cat([order] * reps)[: self.pad_to].contiguous()

    def __len__(self):
        if self.sampler is not None:
            try:
                return len(self.sampler)
            except TypeError:
                pass
        n = len(self.ds)
        if self.world_size > 1:                     # DistributedSampler: equal shares
            n = n // self.world_size if self.drop_last else -(-n // self.world_size)
        if self.pad_to is not None and 0 < n < self.pad_to:
            n = self.pad_to
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def set_epoch(self, epoch: int) -> None:
        """Like ``DistributedSampler.set_epoch``: the permutation is seeded by ``seed + epoch``."""
        self.epoch = int(epoch)

    def __iter__(self):
        if self.sampler is None and not self.python_loop:
            yield from self._iter_native()
            return
        if self.streams is None:
            self.streams = [torch.cuda.Stream(device=self.ds.dev.device) for _ in range(self.in_flight)]
        pending: deque = deque()
        draw = (self.draw_seed, (self.epoch + 1) & 0xFFFFFFFFFFFFFFFF)      # what the native loop keys this epoch's draws by
        it = iter(self._index_batches())
        k = 0

        def submit():
            nonlocal k
            try:
                idx = next(it)
            except StopIteration:
                return False
            st = self.streams[k % self.in_flight]
            k += 1
            st.wait_stream(torch.cuda.current_stream(self.ds.dev.device))
            with torch.cuda.stream(st):
                self.ds._draw_override = (draw[0], draw[1], k - 1)      # (k - 1: this batch's number in the epoch)
                try:
                    batch = self.ds[idx]
                finally:
                    self.ds._draw_override = None
                ev = torch.cuda.Event()
                ev.record(st)
            pending.append((batch, ev))
            return True

        for _ in range(self.in_flight):
            if not submit():
                break
        while pending:
            batch, ev = pending.popleft()
            cur = torch.cuda.current_stream(self.ds.dev.device)
            cur.wait_event(ev)
            batch._arena.record_stream(cur)
            submit()
            yield batch
