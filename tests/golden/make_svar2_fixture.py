"""Regenerates tests/golden/pyref_svar2_consensus.npz.  Runs ONLY in the build container (it reads /root/reference); the committed
.npz is what travels.

The reference validates its SVAR2 two-source reconstruction end to end against an INDEPENDENT pure-Python consensus
(``_consensus``, /root/reference/tests/test_svar2_reconstruct.py:66-93: position-sorted (pos, ilen, allele) decode records applied
to ref[q_start:q_end]; a pure DEL has an empty allele and keeps its anchor base).  That function is AST-extracted here at generation
time and exec'd -- never written into this repo -- and run over

1. the reference test's own fixture (:21-31: a 40 bp contig, SNP@2 A>G, INS@6 C>CAT, DEL@11 GTA>G, two samples x two ploids), and
2. 320 seeded synthetic haplotypes (genvarloader_amd.synth: SNPs + indels, windows inside their contig, no shifts -- what
   `_consensus` models),

each stored as DECODED two-source channels (several var_key / dense splits of the same haplotypes) + the consensus bytes.  The
tests replay them through the oracle's provider and through the HIP path: both must give the consensus' bytes at the consensus'
lengths (= region length + hap_diffs_svar2).

Usage:  python tests/golden/make_svar2_fixture.py
"""

from __future__ import annotations

import ast
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF_TEST = Path("/root/reference/tests/test_svar2_reconstruct.py")
sys.path.insert(0, str(REPO))

from genvarloader_amd import synth  # noqa: E402


def load_consensus():
    tree = ast.parse(REF_TEST.read_text())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_consensus")
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), str(REF_TEST), "exec"), ns)
    return ns["_consensus"]


def channels_from_haps(haps, regions, P, split, rng):
    """haps[k] = list of (pos, ilen, allele bytes) of haplotype k (position-sorted); split(q, rec) -> True when a record of query q
    goes to the dense channel.  -> the eleven channel arrays."""
    B = len(regions)
    vk_pos, vk_il, vk_al, vk_off = [], [], [], [0]
    d_pos, d_il, d_al, d_rng, bits, poff = [], [], [], [], [], [0]
    for q in range(B):
        rows = [haps[q * P + p] for p in range(P)]
        uniq = sorted({r for row in rows for r in row if split(q, r)}, key=lambda r: (r[0], r[1], r[2]))
        d_rng.append((len(d_pos), len(d_pos) + len(uniq)))
        for r in uniq:
            d_pos.append(r[0]); d_il.append(r[1]); d_al.append(r[2])
        for row in rows:
            for r in row:
                if not split(q, r):
                    vk_pos.append(r[0]); vk_il.append(r[1]); vk_al.append(r[2])
            vk_off.append(len(vk_pos))
            bits += [r in row for r in uniq]
            poff.append(len(bits))
    pool = bytearray()

    def offs(als):
        o = [len(pool)]
        for a in als:
            pool.extend(a)
            o.append(len(pool))
        return np.asarray(o, np.int64)

    vk_ao = offs(vk_al)
    d_ao = offs(d_al)
    return dict(vk_pos=np.asarray(vk_pos, np.int32), vk_ilen=np.asarray(vk_il, np.int32), vk_alt_off=vk_ao,
                vk_off=np.asarray(vk_off, np.int64), dense_pos=np.asarray(d_pos, np.int32), dense_ilen=np.asarray(d_il, np.int32),
                dense_alt_off=d_ao, dense_range=np.asarray(d_rng, np.int32).reshape(B, 2),
                dense_present=np.packbits(np.asarray(bits, bool), bitorder="little"), dense_present_off=np.asarray(poff, np.int64),
                alt_bytes=np.frombuffer(bytes(pool), np.uint8).copy())


def main():
    consensus = load_consensus()
    out = {}
    n_case = 0

    def add(ref, ref_offsets, regions, P, haps, split, rng):
        nonlocal n_case
        ch = channels_from_haps(haps, regions, P, split, rng)
        exp, off = bytearray(), [0]
        for k, recs in enumerate(haps):
            q = k // P
            c, s, e = (int(x) for x in regions[q][:3])
            contig = bytes(ref[ref_offsets[c]:ref_offsets[c + 1]])
            pos = np.asarray([r[0] for r in recs], np.int64)
            il = np.asarray([r[1] for r in recs], np.int64)
            exp += consensus(contig, pos, il, [r[2] for r in recs], s, e)
            off.append(len(exp))
        pre = f"{n_case}/"
        out[pre + "ref"] = np.asarray(ref, np.uint8)
        out[pre + "ref_offsets"] = np.asarray(ref_offsets, np.int64)
        out[pre + "regions"] = np.asarray(regions, np.int32)
        out[pre + "ploidy"] = np.int64(P)
        for k2, v in ch.items():
            out[pre + k2] = v
        out[pre + "expected"] = np.frombuffer(bytes(exp), np.uint8).copy()
        out[pre + "expected_offsets"] = np.asarray(off, np.int64)
        n_case += 1

    # 1. the reference test's own fixture (test_svar2_reconstruct.py:21-31), queries = (region, sample) as SparseVar2Source lays them out
    ref = np.frombuffer(b"ACAGTACATGGGTACTAGCTAGGCTAACCGGTTAACCGGT", np.uint8)
    snp, ins, dele = (2, 0, b"G"), (6, 2, b"CAT"), (11, -2, b"")
    haps = [[snp, dele], [ins, dele], [ins], [ins, dele]]            # S0 1|0 0|1 1|1 ; S1 0|0 1|1 0|1
    regions = [[0, 0, 40], [0, 0, 40]]
    rng = np.random.default_rng(0)
    for split in (lambda q, r: False, lambda q, r: True, lambda q, r: r[1] != 0, lambda q, r: r[1] == 0):
        add(ref, [0, 40], regions, 2, haps, split, rng)
    # ... and sub-windows of it (a DEL spanning the window's start, an INS at its last base)
    for s, e in ((12, 40), (13, 30), (3, 7), (0, 12)):
        add(ref, [0, 40], [[0, s, e], [0, s, e]], 2, haps, lambda q, r: r[1] < 0, rng)

    # 2. seeded synthetic haplotypes
    rng = np.random.default_rng(20260808)
    for seed, (indel, dens, length, n_q) in enumerate(((0.0, 1 / 60, 300, 20), (0.3, 1 / 40, 500, 20), (0.6, 1 / 15, 200, 20),
                                                       (0.2, 1 / 300, 2100, 10), (0.4, 1 / 25, 5000, 10))):
        st = synth.make_static(rng, (40_000, 25_000), density=dens, indel_frac=indel, max_indel=30)
        bt = synth.make_batch(rng, st, n_q, 2, length, output_length=-1, slack=10, lookback=60)
        go = bt.geno_offsets
        haps = []
        for o in bt.geno_offset_idx.reshape(-1):
            recs = []
            for v in bt.geno_v_idxs[go[0, o]:go[1, o]]:
                il = int(st.ilens[v])
                al = b"" if il < 0 else bytes(st.alt_alleles[st.alt_offsets[v]:st.alt_offsets[v + 1]])
                recs.append((int(st.v_starts[v]), il, al))
            haps.append(recs)
        af = {int(v): float(st.af[v]) for v in range(len(st.af))}
        pos_af = {}
        for v in range(len(st.v_starts)):
            pos_af[(int(st.v_starts[v]), int(st.ilens[v]))] = float(st.af[v])
        thr = [0.0, 0.3, 0.7, 1.1][seed % 4]
        add(st.ref, st.ref_offsets, bt.regions[:, :3], 2, haps, lambda q, r, thr=thr: pos_af.get((r[0], r[1]), 0.0) >= thr, rng)
    out["n"] = np.int64(n_case)
    for k, v in out.items():
        assert np.asarray(v).dtype != object, k
    np.savez_compressed(HERE / "pyref_svar2_consensus.npz", **out)
    print(f"wrote pyref_svar2_consensus.npz: {n_case} cases, "
          f"{sum(int(out[f'{i}/expected_offsets'].size) - 1 for i in range(n_case))} haplotypes")


if __name__ == "__main__":
    main()
