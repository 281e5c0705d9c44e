"""N > 1 path on CPU: world_size-2 ``gloo`` processes shard a batch by queries, each rank
reconstructs its shard (the oracle stands in for the kernel here -- this test is about the
partitioning + gather, the kernel itself is covered by the gpu tests), and the gathered
result must equal the unsharded one.  Also the host-side synthetic generator."""

import os
import socket

import numpy as np
import pytest

from genvarloader_amd import sharding, synth


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 2, 2)


def test_shard_batch_rebases_offsets():
    rng = np.random.default_rng(0)
    st = synth.make_static(rng, (50_000,), indel_frac=0.3, density=1 / 40)
    bt = synth.make_batch(rng, st, 10, 2, 300, rc_frac=0.5)
    idx = bt.geno_offset_idx.ravel()
    n_per = bt.geno_offsets[1, idx] - bt.geno_offsets[0, idx]
    ko = np.concatenate([[0], np.cumsum(n_per)]).astype(np.int64)
    keep = rng.random(int(ko[-1])) < 0.5
    oo = np.arange(21, dtype=np.int64) * 300
    parts = [sharding.shard_batch(r, 3, bt.regions, bt.shifts, bt.geno_offset_idx, bt.to_rc, keep, ko, oo)
             for r in range(3)]
    assert sum(len(p["regions"]) for p in parts) == 10
    np.testing.assert_array_equal(np.concatenate([p["keep"] for p in parts]), keep)
    for p in parts:
        assert p["keep_offsets"][0] == 0 and p["out_offsets"][0] == 0
        assert len(p["keep_offsets"]) == p["geno_offset_idx"].size + 1
        assert len(p["to_rc"]) == p["geno_offset_idx"].size


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, seed, ragged, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle

        rng = np.random.default_rng(seed)
        st = synth.make_static(rng, (80_000,), indel_frac=0.3, density=1 / 50)
        bt = synth.make_batch(rng, st, 37, 2, 400, rc_frac=0.5, output_length=-1 if ragged else None)
        sh = sharding.shard_batch(rank, world, bt.regions, bt.shifts, bt.geno_offset_idx, bt.to_rc)
        out, oo = oracle.reconstruct_haplotypes_fused(
            sh["regions"], sh["shifts"], sh["geno_offset_idx"], bt.geno_offsets, bt.geno_v_idxs, st.v_starts,
            st.ilens, st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, bt.output_length,
            None, None, sh["to_rc"], False)
        if ragged:
            data, lens = sharding.all_gather_rows(torch.from_numpy(out), torch.from_numpy(np.diff(oo)))
            got = (data.numpy(), lens.numpy())
        else:
            rows = torch.from_numpy(out).reshape(-1, bt.output_length)
            got = (sharding.all_gather_rows(rows).numpy().ravel(), None)
        full, full_oo = oracle.reconstruct_haplotypes_fused(
            bt.regions, bt.shifts, bt.geno_offset_idx, bt.geno_offsets, bt.geno_v_idxs, st.v_starts, st.ilens,
            st.alt_alleles, st.alt_offsets, st.ref, st.ref_offsets, st.pad_char, bt.output_length, None, None,
            bt.to_rc, False)
        ok = np.array_equal(got[0], full) and (got[1] is None or np.array_equal(got[1], np.diff(full_oo)))
        q.put((rank, bool(ok), int(got[0].size)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ragged", [False, True])
def test_two_rank_gloo_shard_and_gather(ragged):
    import torch.multiprocessing as mp

    from oracle import oracle

    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 11, ragged, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res


def test_synth_configs_shapes():
    st, bt = synth.make_config("cfg3", contig=1 << 20, windows=256)
    assert bt.regions.shape == (128, 4) and bt.regions.dtype == np.int32
    assert bt.shifts.shape == (128, 2) and bt.geno_offset_idx.dtype == np.int64
    assert bt.geno_offsets.shape[0] == 2 and bt.to_rc.shape == (256,)
    assert (np.diff(st.v_starts) > 0).all() and len(st.alt_offsets) == len(st.v_starts) + 1
    assert ((st.ilens != 0).mean() > 0.05) and bt.output_length == 2048
    st2, bt2 = synth.make_config("cfg3", contig=1 << 20, windows=256)
    np.testing.assert_array_equal(bt.geno_v_idxs, bt2.geno_v_idxs)  # seeded


def test_epoch_order_matches_distributed_sampler_semantics():
    """Every rank draws the same permutation and takes a strided share; shares are equal-sized,
    cover the dataset (padding by wrap-around) or truncate with drop_last."""
    import torch

    from genvarloader_amd.sharding import epoch_order

    n = 103
    for world in (1, 2, 3, 8):
        for shuffle in (False, True):
            for drop_last in (False, True):
                shares = [epoch_order(n, shuffle=shuffle, seed=11, epoch=4, rank=r, world=world, drop_last=drop_last)
                          for r in range(world)]
                sizes = {int(s.numel()) for s in shares}
                assert len(sizes) == 1
                per = n // world if (drop_last and world > 1) else -(-n // world)
                assert sizes == {per}
                allv = torch.stack(shares, 1).reshape(-1)          # interleave back: rank r holds r::world
                g = torch.Generator().manual_seed(11 + 4)
                full = torch.randperm(n, generator=g) if shuffle else torch.arange(n)
                if world == 1:
                    assert torch.equal(allv, full)
                elif drop_last:
                    assert torch.equal(allv, full[: per * world])
                else:
                    assert torch.equal(allv[:n], full) and torch.equal(allv[n:], full[: per * world - n])
    # another epoch -> another permutation; a caller generator is honoured at world 1
    a = epoch_order(n, shuffle=True, seed=1, epoch=0)
    b = epoch_order(n, shuffle=True, seed=1, epoch=1)
    assert not torch.equal(a, b) and torch.equal(torch.sort(a).values, torch.arange(n))
    g1, g2 = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    assert torch.equal(epoch_order(n, shuffle=True, generator=g1), torch.randperm(n, generator=g2))
    with pytest.raises(ValueError):
        epoch_order(n, shuffle=False, rank=2, world=2)


def test_shard_genotypes_by_sample_partitions_the_csr():
    """the genotype CSR cut by sample range -- every (region, sample, ploid) slot lands on exactly one rank
    with its variant list intact, and the index helpers route between the full grid and a rank's own grid."""
    rng = np.random.default_rng(3)
    R, S, P = 5, 11, 2
    n = rng.integers(0, 7, R * S * P)
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    gv = rng.integers(0, 1000, int(off[-1])).astype(np.int32)
    go = np.stack([off[:-1], off[1:]])
    for world in (1, 2, 3, 8, 16):
        seen = np.zeros(R * S, np.int32)
        total_entries = 0
        for rank in range(world):
            lo, lv, (s0, s1) = sharding.shard_genotypes_by_sample(go, gv, R, S, P, world, rank)
            sl = s1 - s0
            assert lo.shape == (2, R * sl * P) and lo.dtype == np.int64 and lv.dtype == np.int32
            total_entries += lv.size
            assert (lo[0, 1:] == lo[1, :-1]).all() and (lo.size == 0 or lo[0, 0] == 0)       # compacted, back to back
            loc = np.arange(R * sl)
            glob = sharding.global_index(loc, S, s0, s1)
            np.testing.assert_array_equal(sharding.local_index(glob, S, s0, s1), loc)
            np.testing.assert_array_equal(sharding.owner_of(glob, S, world), rank)
            seen[glob] += 1
            for q_loc, q in zip(loc, glob):
                for p in range(P):
                    a = lv[lo[0, q_loc * P + p]:lo[1, q_loc * P + p]]
                    b = gv[go[0, q * P + p]:go[1, q * P + p]]
                    np.testing.assert_array_equal(a, b)
        assert (seen == 1).all() and total_entries == gv.size
    import torch

    lo_t, lv_t, rng_t = sharding.shard_genotypes_by_sample(torch.from_numpy(go), torch.from_numpy(gv), R, S, P, 3, 1)
    lo_n, lv_n, rng_n = sharding.shard_genotypes_by_sample(go, gv, R, S, P, 3, 1)
    assert rng_t == rng_n and torch.equal(lo_t, torch.from_numpy(lo_n)) and torch.equal(lv_t, torch.from_numpy(lv_n))
    np.testing.assert_array_equal(sharding.shard_genotypes_by_sample(off, gv, R, S, P, 3, 1)[0], lo_n)    # (n + 1,) offsets too


def test_shard_svar2_batch_rows_equal_the_full_batch(oracle):
    """A SVAR2 two-source batch cut into per-rank query blocks (shard_svar2_batch re-bases var_key slices, dense windows and presence
    bits): every rank's rows == the same rows of the full batch (oracle), for 1 .. 5 ranks, ragged output."""
    from genvarloader_amd import synth

    rng = np.random.default_rng(17)
    st = synth.make_static(rng, (150_000,), indel_frac=0.3)
    bt = synth.make_batch(rng, st, 23, 2, 700, output_length=-1)
    sv = synth.to_svar2(rng, st, bt, dense_af=0.3)
    full, foff = oracle.reconstruct_haplotypes_from_svar2(bt.regions, bt.shifts, *sv.args(), st.ref, st.ref_offsets, st.pad_char, -1)
    for world in (1, 2, 3, 5):
        got = []
        for rank in range(world):
            sh = sharding.shard_svar2_batch(rank, world, bt.regions, bt.shifts, *sv.args())
            out, off = oracle.reconstruct_haplotypes_from_svar2(
                sh["regions"], sh["shifts"], sh["vk_pos"], sh["vk_ilen"], sh["vk_alt_off"], sh["vk_off"], sh["dense_pos"], sh["dense_ilen"],
                sh["dense_alt_off"], sh["dense_range"], sh["dense_present"], sh["dense_present_off"], sh["alt_bytes"], st.ref,
                st.ref_offsets, st.pad_char, -1)
            k0, k1 = sh["row_range"]
            np.testing.assert_array_equal(np.diff(off), np.diff(foff[k0:k1 + 1]))
            got.append(out)
        np.testing.assert_array_equal(np.concatenate(got), full)
