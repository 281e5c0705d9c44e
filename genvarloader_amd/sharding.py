"""Row sharding of a batch across the GPUs of one node.

The path shards embarrassingly: rows ``k = (query, hap)`` write disjoint output slices and
only read shared inputs -- the property that makes the reference's rayon ``par_iter`` safe
(``/root/reference/src/reconstruct/mod.rs:374-375,428-452``).  Each rank therefore takes a
contiguous block of QUERIES (so the ``P`` haplotypes of a query, which share ``to_rc`` and the
reference window, stay together and ``(b, P, L)`` reshapes hold), runs the kernel on its
block, and no collective is on the data path.  The per-dataset arrays (reference, variant
table, genotype CSR) are replicated on every GPU.

When the genotype CSR is too large to replicate ("shard or replicate genotype CSR by sample range depending
on size vs 288 GB HBM; with sample-sharded CSR route each query to the GPU owning its sample"), ``shard_genotypes_by_sample``
cuts it by SAMPLE range instead: rank ``r`` keeps the slots of samples ``[s0, s1)`` for every region, compacted, and its
dataset is the ``(regions x owned samples)`` grid -- ``owner_of`` / ``global_index`` / ``local_index`` do the routing.  The
reference, variant table and regions stay replicated (they do not grow with the cohort).

``all_gather_rows`` is the optional final gather for a single consumer (RCCL over xGMI when
the process group backend is ``nccl``; ``gloo`` in the CPU tests): fixed-length rows gather
as equal-size blocks (padded to the largest shard), ragged rows gather lengths first.
"""

from __future__ import annotations

import numpy as np


def splitmix64(x: int) -> int:
    """One step of splitmix64 (seed derivation: per loader, per non-deterministic batch)."""
    z = (int(x) + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def shard_bounds(n_queries: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block of queries for `rank`: sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_queries), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_genotypes_by_sample(geno_offsets, geno_v_idxs, n_regions: int, n_samples: int, ploidy: int, world: int, rank: int):
    """This rank's share of a sparse-genotype CSR sharded by SAMPLE range.

    ``geno_offsets``: ``(2, R*S*P)`` starts / stops (or ``(R*S*P + 1,)``), slot =
    ``ravel_multi_index((region, sample, ploid), (R, S, P))`` (``_haps.py:757-768``); ``geno_v_idxs`` int32.  numpy arrays or
    torch tensors (any device; the result lives where the input does).
    -> ``(local_offsets (2, R*S_loc*P) int64, local_v_idxs int32, (s0, s1))``: the slots of samples ``[s0, s1)`` in
    ``(R, S_loc, P)`` order over a compacted copy of their entries -- what a ``HapsDevice`` + ``DeviceHapsDataset(n_samples =
    s1 - s0)`` of that rank take.  Memory per rank: 1 / world of the entries (+ the same share of the derived layouts)."""
    import torch

    as_np = not isinstance(geno_offsets, torch.Tensor)
    go = torch.as_tensor(np.asarray(geno_offsets) if as_np else geno_offsets).to(torch.int64)
    gv = torch.as_tensor(np.asarray(geno_v_idxs) if not isinstance(geno_v_idxs, torch.Tensor) else geno_v_idxs)
    gv = gv.to(device=go.device, dtype=torch.int32)
    R, S, P = int(n_regions), int(n_samples), int(ploidy)
    if go.dim() == 1:
        go = torch.stack([go[:-1], go[1:]])
    if tuple(go.shape) != (2, R * S * P):
        raise ValueError("geno_offsets must cover regions x samples x ploidy slots")
    s0, s1 = shard_bounds(S, world, rank)
    sl = s1 - s0
    starts = go[0].view(R, S, P)[:, s0:s1, :].reshape(-1)
    stops = go[1].view(R, S, P)[:, s0:s1, :].reshape(-1)
    n = (stops - starts).clamp_min(0)
    off = torch.zeros(n.numel() + 1, dtype=torch.int64, device=go.device)
    torch.cumsum(n, 0, out=off[1:])
    total = int(off[-1].item()) if n.numel() else 0
    row_of = torch.repeat_interleave(torch.arange(n.numel(), device=go.device), n)
    src = starts[row_of] + (torch.arange(total, device=go.device) - off[row_of])
    lv = gv[src] if total else gv[:0]
    lo = torch.stack([off[:-1], off[1:]]).contiguous()
    assert lo.shape[1] == R * sl * P
    if as_np:
        return lo.cpu().numpy(), lv.cpu().numpy(), (s0, s1)
    return lo, lv.contiguous(), (s0, s1)


def global_index(local_idx, n_samples: int, s0: int, s1: int):
    """Dataset index over the (regions x owned samples) grid of a sample shard -> index over the full (regions x samples) grid."""
    sl = int(s1) - int(s0)
    return (local_idx // sl) * int(n_samples) + int(s0) + (local_idx % sl)


def local_index(global_idx, n_samples: int, s0: int, s1: int):
    """Inverse of :func:`global_index` (for indices whose sample this shard owns)."""
    sl = int(s1) - int(s0)
    return (global_idx // int(n_samples)) * sl + (global_idx % int(n_samples)) - int(s0)


def owner_of(global_idx, n_samples: int, world: int):
    """The rank that owns a dataset index's sample under :func:`shard_genotypes_by_sample` (works on ints, numpy arrays, tensors)."""
    base, rem = divmod(int(n_samples), int(world))
    s = global_idx % int(n_samples)
    cut = rem * (base + 1)                     # the first `rem` ranks own base + 1 samples
    if base == 0:
        return s
    return (s < cut) * (s // (base + 1)) + (s >= cut) * (rem + (s - cut) // base)


def shard_batch(rank: int, world: int, regions, shifts, geno_offset_idx, to_rc=None, keep=None,
                keep_offsets=None, out_offsets=None) -> dict:
    """Slice the per-batch arrays (``ReconstructionRequest``, _haps.py:58-93) to this rank's
    queries.  ``keep`` / ``keep_offsets`` / ``out_offsets`` are re-based to the shard."""
    regions = np.asarray(regions)
    goi = np.asarray(geno_offset_idx)
    B, P = goi.shape
    lo, hi = shard_bounds(B, world, rank)
    k0, k1 = lo * P, hi * P
    out = dict(regions=regions[lo:hi], shifts=np.asarray(shifts)[lo:hi], geno_offset_idx=goi[lo:hi],
               to_rc=None if to_rc is None else np.asarray(to_rc)[k0:k1], keep=None, keep_offsets=None,
               out_offsets=None, query_range=(lo, hi), row_range=(k0, k1))
    if keep is not None and keep_offsets is not None:
        ko = np.asarray(keep_offsets, np.int64)
        out["keep"] = np.asarray(keep)[ko[k0]:ko[k1]]
        out["keep_offsets"] = ko[k0:k1 + 1] - ko[k0]
    if out_offsets is not None:
        oo = np.asarray(out_offsets, np.int64)
        out["out_offsets"] = oo[k0:k1 + 1] - oo[k0]
    return out


def shard_svar2_batch(rank: int, world: int, regions, shifts, vk_pos, vk_ilen, vk_alt_off, vk_off, dense_pos, dense_ilen, dense_alt_off,
                      dense_range, dense_present, dense_present_off, alt_bytes) -> dict:
    """This rank's contiguous block of QUERIES of a SVAR2 two-source batch (decoded channels, ``genvarloader_amd.svar2``): the
    haplotypes' var_key slices, the queries' dense windows and the haplotypes' presence bits are cut out and re-based, so that the
    shard is a batch of its own -- same bytes for its rows as the full batch gives them (rows are independent:
    ``src/reconstruct/mod.rs:654-757``).  The allele pool is shared by reference (offsets into it stay what they are)."""
    regions, shifts = np.asarray(regions), np.asarray(shifts)
    B, P = shifts.shape
    lo, hi = shard_bounds(B, world, rank)
    k0, k1 = lo * P, hi * P
    vo = np.asarray(vk_off, np.int64)
    dr = np.asarray(dense_range, np.int64).reshape(-1, 2)
    po = np.asarray(dense_present_off, np.int64)
    v0, v1 = int(vo[k0]), int(vo[k1])
    d0 = int(dr[lo:hi, 0].min()) if hi > lo else 0
    d1 = int(dr[lo:hi, 1].max()) if hi > lo else 0
    d1 = max(d1, d0)
    b0, b1 = int(po[k0]), int(po[k1])
    bits = np.unpackbits(np.asarray(dense_present, np.uint8).reshape(-1), bitorder="little")[b0:b1]
    return dict(
        regions=regions[lo:hi], shifts=shifts[lo:hi],
        vk_pos=np.asarray(vk_pos)[v0:v1], vk_ilen=np.asarray(vk_ilen)[v0:v1], vk_alt_off=np.asarray(vk_alt_off, np.int64)[v0:v1 + 1],
        vk_off=vo[k0:k1 + 1] - v0,
        dense_pos=np.asarray(dense_pos)[d0:d1], dense_ilen=np.asarray(dense_ilen)[d0:d1],
        dense_alt_off=np.asarray(dense_alt_off, np.int64)[d0:d1 + 1],
        dense_range=(dr[lo:hi] - d0).astype(np.int32), dense_present=np.packbits(bits, bitorder="little"),
        dense_present_off=po[k0:k1 + 1] - b0, alt_bytes=np.asarray(alt_bytes, np.uint8),
        query_range=(lo, hi), row_range=(k0, k1))


def all_gather_rows(local, row_lengths=None, group=None):
    """Gather every rank's rows on every rank, in rank order.

    ``local``: torch tensor whose first dimension is this rank's rows (fixed-length rows,
    e.g. ``(rows, L)`` haplotypes or ``(rows, L, 4)`` one-hot), or a flat 1-D tensor of ragged
    rows with ``row_lengths`` (1-D int64 tensor, one entry per local row).
    Returns the concatenated tensor (and the concatenated lengths in the ragged case)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = local.device
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts) if counts else 0
    if counts and min(counts) == mx:
        # equal shards (fixed-length rows of an evenly split batch: the usual case): ONE collective straight into the result --
        # no padded copy per rank, no list of per-rank buffers to concatenate (all_gather_into_tensor; on ROCm the nccl backend's
        # ncclAllGather over xGMI).  gloo has no such collective for every dtype / device: fall back to the list form below.
        out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
        try:
            dist.all_gather_into_tensor(out, local.contiguous(), group=group)
            if row_lengths is None:
                return out
            return out, all_gather_rows(row_lengths.to(dev), None, group)
        except (RuntimeError, NotImplementedError, AttributeError):
            pass
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    data = torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    if row_lengths is None:
        return data
    lens = all_gather_rows(row_lengths.to(dev), None, group)
    return data, lens


def epoch_order(n: int, *, shuffle: bool, seed: int = 0, epoch: int = 0, rank: int = 0, world: int = 1,
                drop_last: bool = False, device="cpu", generator=None, scratch_generator=None):
    """This rank's dataset indices for one epoch, as an int64 tensor on ``device``.

    The epoch is sharded across ranks the way ``torch.utils.data.DistributedSampler`` does it
    (the sampler the reference's ``to_dataloader`` docs pair with DDP, ``_impl.py:1963-2072``):
    every rank draws the SAME permutation (seeded by ``seed + epoch``), the list is padded by
    wrapping (or truncated with ``drop_last``) to a multiple of ``world``, and rank ``r`` takes
    elements ``r, r + world, ...`` -- disjoint, equal-sized, no collective.  With ``world == 1``
    and a caller ``generator`` the permutation comes from that generator instead.
    ``scratch_generator``: a generator on ``device`` to re-seed and draw from instead of making a new one per call (same draws)."""
    import torch

    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    if not shuffle:
        order = torch.arange(n, device=device)
    elif generator is not None and world == 1:
        order = torch.randperm(n, generator=generator, device=generator.device).to(device)
    else:
        g = scratch_generator if scratch_generator is not None else torch.Generator(device=device)
        g.manual_seed(int(seed) + int(epoch))
        order = torch.randperm(n, generator=g, device=device)
    if world == 1:
        return order
    if drop_last:
        total = (n // world) * world
        order = order[:total]
    else:
        total = -(-n // world) * world
        if total > n and n > 0:
            reps = -(-(total - n) // n)
            order = torch.cat([order] + [order] * reps)[:total]
    return order[rank:total:world].contiguous()
