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

Randomness: jitter and shifts use a ``torch.Generator`` on the device; the *distribution* is
the reference's (``rng.integers(-j, j + 1)``, ``rng.integers(0, max_shift + 1)``), the stream
of numbers is not (numpy's PCG64 is not reproducible on a GPU).  ``deterministic=True``
involves no randomness and is bit-exact.
"""

from __future__ import annotations

from collections import deque
from dataclasses import dataclass

import numpy as np
import torch


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


class DeviceHapsDataset:
    """(regions x samples) grid over a :class:`HapsDevice`, fixed output length.

    ``dev``'s genotype offsets must be laid out like the reference's sparse genotypes: slot
    ``ravel_multi_index((region, sample, ploid), (R, S, P))`` (``_haps.py:757-768``)."""

    def __init__(self, dev, regions, n_samples: int, ploidy: int, *, output_length: int, jitter: int = 0,
                 rc_neg: bool = True, deterministic: bool = True, seed: int | None = None, onehot: bool = True,
                 haps: bool = False, layout: str = "lc"):
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
        if self.output_length < 1:
            raise ValueError("output_length must be >= 1 (fixed-length output)")
        self.jitter, self.rc_neg, self.deterministic = int(jitter), bool(rc_neg), bool(deterministic)
        self.onehot, self.haps, self.layout = bool(onehot), bool(haps), layout
        self.generator = torch.Generator(device=d)
        self.generator.manual_seed(0 if seed is None else int(seed))

    @property
    def shape(self):
        return (self.n_regions, self.n_samples)

    def __len__(self):
        return self.n_regions * self.n_samples

    def request(self, idx):
        """Device-side ``_prepare_request``: everything the kernel needs, no host round trip."""
        d = self.dev.device
        idx = torch.as_tensor(np.asarray(idx) if not isinstance(idx, torch.Tensor) else idx).to(d).reshape(-1)
        regions, goi, to_rc, lengths = build_request(idx, self.full_regions, self.n_samples, self.ploidy,
                                                     self.jitter, self.rc_neg, self.generator)
        if self.deterministic:                                         # _haps.py:720-722
            shifts = torch.zeros(goi.shape, dtype=torch.int32, device=d)
        else:                                                          # _haps.py:723-730
            diffs = self.dev.get_diffs_sparse(goi, q_starts=regions[:, 1].contiguous(),
                                              q_ends=regions[:, 2].contiguous())
            max_shift = diffs.clamp(min=0).to(torch.int64) + (lengths - self.output_length).clamp(min=0)[:, None]
            u = torch.rand(goi.shape, device=d, generator=self.generator, dtype=torch.float64)
            shifts = torch.minimum((u * (max_shift + 1).to(torch.float64)).floor().to(torch.int64), max_shift)
            shifts = shifts.to(torch.int32)
        return idx, regions, shifts.contiguous(), goi, to_rc

    def __getitem__(self, idx) -> Batch:
        idx, regions, shifts, goi, to_rc = self.request(idx)
        out = self.dev.reconstruct(regions, shifts, goi, self.output_length, to_rc=to_rc, haps=self.haps,
                                   onehot=self.onehot, layout=self.layout)
        b, P, L = goi.shape[0], self.ploidy, self.output_length
        oh = None
        if out.onehot is not None:
            oh = out.onehot.view(b, P, L, 4) if self.layout == "lc" else out.onehot.view(b, P, 4, L)
        hp = out.haps.view(b, P, L) if out.haps is not None else None
        return Batch(oh, hp, idx, regions, shifts, goi, to_rc)

    def to_dataloader(self, batch_size: int = 1, shuffle: bool = False, sampler=None, drop_last: bool = False,
                      generator: torch.Generator | None = None, in_flight: int = 3) -> "DeviceLoader":
        """``Dataset.to_dataloader`` (``_impl.py:1963-2072``) for device consumers."""
        return DeviceLoader(self, batch_size, shuffle, sampler, drop_last, generator, in_flight)


class DeviceLoader:
    """Iterates batches of a :class:`DeviceHapsDataset`, ``in_flight`` batches ahead, each on
    its own HIP stream; the consumer's current stream waits on the batch's event."""

    def __init__(self, ds: DeviceHapsDataset, batch_size=1, shuffle=False, sampler=None, drop_last=False,
                 generator=None, in_flight=3):
        self.ds, self.batch_size, self.shuffle, self.drop_last = ds, int(batch_size), shuffle, drop_last
        self.sampler, self.generator = sampler, generator
        self.in_flight = max(1, int(in_flight))
        self.streams = [torch.cuda.Stream(device=ds.dev.device) for _ in range(self.in_flight)]

    def _index_batches(self):
        if self.sampler is not None:
            for b in self.sampler:                      # a BatchSampler-like iterable of index lists
                yield np.asarray(b, dtype=np.int64).reshape(-1)
            return
        n = len(self.ds)
        order = torch.randperm(n, generator=self.generator).numpy() if self.shuffle else np.arange(n)
        for s in range(0, n, self.batch_size):
            b = order[s:s + self.batch_size]
            if len(b) < self.batch_size and self.drop_last:
                return
            yield b

    def __len__(self):
        n = len(self.ds)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        pending: deque = deque()
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
                batch = self.ds[idx]
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
            for t in (batch.onehot, batch.haps):
                if t is not None:
                    t.record_stream(cur)
            submit()
            yield batch
