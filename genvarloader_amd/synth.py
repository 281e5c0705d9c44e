"""Seeded synthetic inputs for the haplotype hot path (host side, numpy only).

Shapes and dtypes are exactly what the reference's FFI consumes
(``/root/reference/src/ffi/mod.rs:724-743``; layout conventions in
``python/genvarloader/_dataset/_haps.py:844-866``): ``regions`` i32 ``(B, 4)``
``[contig, start, end, strand]``, ``shifts`` i32 ``(B, P)``, ``geno_offset_idx``
i64 ``(B, P)``, ``geno_offsets`` i64 ``(2, n)`` starts/stops, ``geno_v_idxs`` i32,
variant table ``v_starts``/``ilens`` i32 + ``alt_alleles`` u8 / ``alt_offsets`` i64,
``ref`` u8 / ``ref_offsets`` i64.  The role is the one ``_dummy.py:23-213`` plays
in the reference: an in-memory dataset with no files behind it.

The recipe follows SURVEY.md section 8(d): uniform ACGT reference with 1 % N,
Poisson variant density 1/300 bp, SNP-only or 85 % SNP / 15 % atomised indel
(|ilen| ~ Geometric(0.35) clipped to [1, 30]; an insertion's ALT is the anchor
base followed by ilen random bases, a deletion's ALT is the anchor base), each
haplotype carrying each overlapping variant with a per-variant allele frequency
AF ~ Beta(a, b).  With the default Beta(0.6, 0.9) a 2048 bp window sees about 3
variants per haplotype.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

ACGT = np.frombuffer(b"ACGT", np.uint8)


@dataclass
class SynthStatic:
    """Per-dataset arrays (uploaded once; mirror of ``_HapsFfiStatic``,
    ``_haps.py:233-247`` plus ``Reference``, ``_reference.py:31-50``)."""

    ref: np.ndarray
    ref_offsets: np.ndarray
    v_starts: np.ndarray
    ilens: np.ndarray
    alt_alleles: np.ndarray
    alt_offsets: np.ndarray
    v_contig: np.ndarray  # i32 contig of each variant (generator bookkeeping only)
    af: np.ndarray
    pad_char: int = ord("N")


@dataclass
class SynthBatch:
    """Per-batch arrays (``ReconstructionRequest``, ``_haps.py:58-93``) plus the
    sparse-genotype CSR the batch indexes."""

    regions: np.ndarray
    shifts: np.ndarray
    geno_offset_idx: np.ndarray
    geno_offsets: np.ndarray
    geno_v_idxs: np.ndarray
    to_rc: np.ndarray | None
    output_length: int
    keep: np.ndarray | None = None
    keep_offsets: np.ndarray | None = None
    meta: dict = field(default_factory=dict)

    @property
    def n_windows(self) -> int:
        return int(self.geno_offset_idx.size)

    @property
    def mean_variants(self) -> float:
        go = self.geno_offsets
        idx = self.geno_offset_idx.ravel()
        return float((go[1, idx] - go[0, idx]).mean()) if idx.size else 0.0


def make_static(
    rng: np.random.Generator,
    contig_lens=(1 << 20,),
    density: float = 1.0 / 300.0,
    indel_frac: float = 0.0,
    n_frac: float = 0.01,
    af_beta=(0.6, 0.9),
    max_indel: int = 30,
) -> SynthStatic:
    contig_lens = [int(x) for x in contig_lens]
    ref_offsets = np.zeros(len(contig_lens) + 1, np.int64)
    ref_offsets[1:] = np.cumsum(contig_lens)
    total = int(ref_offsets[-1])
    ref = ACGT[rng.integers(0, 4, total, dtype=np.uint8)]
    if n_frac > 0:
        ref[rng.random(total) < n_frac] = ord("N")

    v_starts_l, ilens_l, alts_l, alt_len_l, contig_l = [], [], [], [], []
    for c, clen in enumerate(contig_lens):
        n = int(rng.poisson(clen * density))
        if n == 0:
            continue
        pos = np.unique(rng.integers(0, clen, n, dtype=np.int64))
        n = len(pos)
        is_indel = rng.random(n) < indel_frac
        mag = np.clip(rng.geometric(0.35, n), 1, max_indel).astype(np.int64)
        sign = np.where(rng.random(n) < 0.5, -1, 1)
        il = np.where(is_indel, mag * sign, 0).astype(np.int64)
        # a deletion may not run past the contig end (pos - ilen + 1 <= clen)
        room = clen - 1 - pos
        il = np.where(il < 0, -np.minimum(-il, room), il)
        alt_len = 1 + np.maximum(il, 0)
        off = np.zeros(n + 1, np.int64)
        off[1:] = np.cumsum(alt_len)
        alt = ACGT[rng.integers(0, 4, int(off[-1]), dtype=np.uint8)]
        base = ref[ref_offsets[c] + pos]
        first = off[:-1]
        snp = il == 0
        # SNP: a base different from the reference base
        code = np.searchsorted(ACGT, base)  # N (78) -> 4? searchsorted on sorted ACGT
        code = np.where(code > 3, 0, code)
        alt_snp = ACGT[(code + rng.integers(1, 4, n)) % 4]
        alt[first] = np.where(snp, alt_snp, base)  # indels keep the anchor base
        v_starts_l.append(pos.astype(np.int32))
        ilens_l.append(il.astype(np.int32))
        alts_l.append(alt)
        alt_len_l.append(alt_len)
        contig_l.append(np.full(n, c, np.int32))

    if v_starts_l:
        v_starts = np.concatenate(v_starts_l)
        ilens = np.concatenate(ilens_l)
        alt_alleles = np.concatenate(alts_l)
        alt_len = np.concatenate(alt_len_l)
        v_contig = np.concatenate(contig_l)
    else:
        v_starts = np.zeros(0, np.int32)
        ilens = np.zeros(0, np.int32)
        alt_alleles = np.zeros(0, np.uint8)
        alt_len = np.zeros(0, np.int64)
        v_contig = np.zeros(0, np.int32)
    alt_offsets = np.zeros(len(v_starts) + 1, np.int64)
    alt_offsets[1:] = np.cumsum(alt_len)
    af = rng.beta(af_beta[0], af_beta[1], len(v_starts))
    return SynthStatic(ref, ref_offsets, v_starts, ilens, alt_alleles, alt_offsets, v_contig, af)


def sample_genotypes(rng, st: SynthStatic, contig, start, end, ploidy: int, lookback: int = 40):
    """Sparse genotypes for len(start) queries x ploidy haplotypes: each haplotype carries each
    variant with pos in [start - lookback, end) on its contig with that variant's AF.
    -> (geno_offsets (2, n) i64 starts/stops, geno_v_idxs i32), rows in (query, hap) order."""
    contig = np.asarray(contig, np.int64)
    start = np.asarray(start, np.int64)
    end = np.asarray(end, np.int64)
    B, P = len(start), int(ploidy)
    n_contigs = len(st.ref_offsets) - 1
    c_first = np.searchsorted(st.v_contig, np.arange(n_contigs), "left")
    c_last = np.searchsorted(st.v_contig, np.arange(n_contigs), "right")
    lo = np.zeros(B, np.int64)
    hi_ = np.zeros(B, np.int64)
    for c in range(n_contigs):
        m = contig == c
        if not m.any():
            continue
        vs = st.v_starts[c_first[c] : c_last[c]]
        lo[m] = c_first[c] + np.searchsorted(vs, start[m] - lookback, "left")
        hi_[m] = c_first[c] + np.searchsorted(vs, end[m], "left")
    n_cand = np.repeat(hi_ - lo, P)  # per row k
    row_lo = np.repeat(lo, P)
    K = B * P
    cand_off = np.zeros(K + 1, np.int64)
    cand_off[1:] = np.cumsum(n_cand)
    tot = int(cand_off[-1])
    row_of = np.repeat(np.arange(K), n_cand)
    v_of = row_lo[row_of] + (np.arange(tot) - cand_off[row_of])
    carried = rng.random(tot) < st.af[v_of]
    geno_v_idxs = v_of[carried].astype(np.int32)
    counts = np.bincount(row_of[carried], minlength=K).astype(np.int64)
    offs = np.zeros(K + 1, np.int64)
    offs[1:] = np.cumsum(counts)
    return np.ascontiguousarray(np.stack([offs[:-1], offs[1:]])), geno_v_idxs


def make_grid(rng, st: SynthStatic, n_regions: int, n_samples: int, ploidy: int = 2, length: int = 2048,
              slack: int = 32, rc_frac: float = 0.5):
    """A (regions x samples x ploidy) dataset laid out like the reference's sparse genotypes
    (slot = ravel_multi_index((r, s, p), (R, S, P)), _haps.py:757-768): BASELINE config 5 is
    200 regions x 2504 samples x 2.  -> (full_regions (R, 4) i32, geno_offsets, geno_v_idxs)."""
    R, S = int(n_regions), int(n_samples)
    n_contigs = len(st.ref_offsets) - 1
    clens = np.diff(st.ref_offsets)
    contig = rng.integers(0, n_contigs, R).astype(np.int64)
    span = length + 2 * slack
    start = (rng.random(R) * np.maximum(clens[contig] - span, 1)).astype(np.int64)
    end = start + span
    strand = np.where(rng.random(R) < rc_frac, -1, 1)
    full_regions = np.stack([contig, start, end, strand], axis=1).astype(np.int32)
    go, gv = sample_genotypes(rng, st, np.repeat(contig, S), np.repeat(start, S), np.repeat(end, S), ploidy)
    return full_regions, go, gv


def make_batch(
    rng: np.random.Generator,
    st: SynthStatic,
    n_queries: int,
    ploidy: int = 2,
    length: int = 2048,
    slack: int = 32,
    rc_frac: float = 0.0,
    random_shifts: bool = False,
    output_length: int | None = None,
    lookback: int = 40,
    edge_frac: float = 0.0,
    permute_csr: bool = False,
    contig_start_frac: float = 0.0,
) -> SynthBatch:
    """``n_queries`` regions of ``length + 2*slack`` bp, ``ploidy`` haplotypes each.

    ``edge_frac`` of the regions are pushed over a contig edge (negative start /
    end past the contig) to exercise the padding branches.  ``output_length``
    defaults to ``length`` (fixed-length crop); pass ``-1`` for ragged.  ``contig_start_frac`` of the
    regions begin within 8 bases of their contig's first base (an insertion there is longer than the
    reference in front of it)."""
    B, P = int(n_queries), int(ploidy)
    n_contigs = len(st.ref_offsets) - 1
    clens = np.diff(st.ref_offsets)
    contig = rng.integers(0, n_contigs, B).astype(np.int64)
    span = length + 2 * slack
    hi = np.maximum(clens[contig] - span, 1)
    start = (rng.random(B) * hi).astype(np.int64)
    if edge_frac > 0:
        edge = rng.random(B) < edge_frac
        left = rng.random(B) < 0.5
        off = rng.integers(1, max(2, span // 2), B)
        start = np.where(edge & left, -off, start)
        start = np.where(edge & ~left, clens[contig] - span + off, start)
    if contig_start_frac > 0:       # windows that begin within a few bases of their contig's first base
        near = rng.random(B) < contig_start_frac
        start = np.where(near, rng.integers(0, 8, B), start)
    end = start + span
    strand = np.where(rng.random(B) < rc_frac, -1, 1)
    regions = np.stack([contig, start, end, strand], axis=1).astype(np.int32)

    geno_offsets, geno_v_idxs = sample_genotypes(rng, st, contig, start, end, P, lookback)
    offs = np.concatenate([geno_offsets[0], geno_offsets[1, -1:]]) if geno_offsets.shape[1] else np.zeros(1, np.int64)
    K = B * P
    geno_offset_idx = np.arange(K, dtype=np.int64).reshape(B, P)
    if permute_csr:
        # shuffle which CSR slot each row uses (geno_offset_idx is then non-trivial)
        perm = rng.permutation(K)
        inv = np.empty(K, np.int64)
        inv[perm] = np.arange(K)
        geno_offsets = np.ascontiguousarray(geno_offsets[:, perm])
        geno_offset_idx = inv.reshape(B, P)

    out_len = length if output_length is None else int(output_length)
    if random_shifts and out_len >= 0:
        # _haps.py:728-730: max_shift = clip(diff, 0) + clip(region_len - output_length, 0)
        il = st.ilens[geno_v_idxs].astype(np.int64)
        csum = np.concatenate([[0], np.cumsum(np.maximum(il, 0))])
        approx = csum[offs[1:]] - csum[offs[:-1]]
        max_shift = approx.reshape(B, P) + np.clip((end - start) - out_len, 0, None)[:, None]
        shifts = rng.integers(0, max_shift + 1).astype(np.int32)
    else:
        shifts = np.zeros((B, P), np.int32)
    to_rc = np.repeat(strand == -1, P) if rc_frac > 0 else None
    return SynthBatch(
        regions=regions, shifts=shifts, geno_offset_idx=geno_offset_idx,
        geno_offsets=geno_offsets, geno_v_idxs=geno_v_idxs, to_rc=to_rc,
        output_length=out_len,
        meta=dict(B=B, P=P, length=length, slack=slack, rc_frac=rc_frac),
    )


# --- BASELINE.json configs ---------------------------------
CONFIGS = {
    # name: (n_windows, length, indel_frac, rc_frac, contig_len)
    "cfg1": dict(windows=1024, length=1024, indel_frac=0.0, rc_frac=0.0, contig=64 << 20),
    "cfg2": dict(windows=4096, length=2048, indel_frac=0.0, rc_frac=0.0, contig=64 << 20),
    "cfg3": dict(windows=4096, length=2048, indel_frac=0.15, rc_frac=0.5, contig=64 << 20),
    "cfg4": dict(windows=256, length=131072, indel_frac=0.15, rc_frac=0.0, contig=256 << 20),
}


def make_config(name: str, seed: int | None = None, ploidy: int = 2, contig: int | None = None,
                windows: int | None = None, random_shifts: bool = False):
    """Build (static, batch) for a BASELINE.json config.  Seeds follow SURVEY 8(d):
    ``default_rng(20260802 + cfg_index)``."""
    cfg = CONFIGS[name]
    idx = int(name[3:])
    rng = np.random.default_rng(20260802 + idx if seed is None else seed)
    st = make_static(rng, (contig or cfg["contig"],), indel_frac=cfg["indel_frac"])
    k = windows or cfg["windows"]
    bt = make_batch(rng, st, k // ploidy, ploidy, cfg["length"], rc_frac=cfg["rc_frac"],
                    random_shifts=random_shifts)
    bt.meta["config"] = name
    return st, bt


# --- genome-scale dataset, generated on the device ------------------------------------
# bench.py's default: a reference, variant table and sparse-genotype CSR too large for any
# cache of the part (256 MiB Infinity Cache), with batches drawn across the whole genome, so
# that "achieved HBM GB/s" is measured on cold inputs.  Same recipe as the numpy generators
# above, written with torch ops so that 3.1 Gbp take a second on the GPU; it
# also runs on CPU tensors at small sizes (tests).

# GRCh38 primary assembly, chr1..22, X, Y (bp)
HG38_CONTIGS = (248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
                138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
                83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415)

SCALES = {
    # name: (contig lengths, queries in the dataset)
    "small": ((64 << 20,), 1 << 16),      # SURVEY 8(d)'s 64 Mbp contig: everything cache resident
    "hg38": (HG38_CONTIGS, 1 << 23),      # 3.09 Gbp; 8.4 M queries x 2 haplotypes -> CSR + records > 1 GB
}


class GenomeDataset:
    """Reference + variant table + a (queries x ploidy) sparse-genotype CSR as torch tensors on
    ``device``.  Query ``q`` owns genotype slots ``q * P .. q * P + P - 1`` (the reference's
    ``ravel_multi_index`` layout with one sample per region, ``_haps.py:757-768``)."""

    def __init__(self, device="cuda", contigs=(64 << 20,), n_queries=1 << 16, ploidy=2, length=2048, slack=32,
                 indel_frac=0.15, rc_frac=0.5, density=1.0 / 300.0, n_frac=0.01, af_beta=(0.6, 0.9),
                 max_indel=30, lookback=40, seed=0):
        import torch

        d = torch.device(device)
        self.device = d
        g = torch.Generator(device=d)
        g.manual_seed(int(seed))
        nrng = np.random.default_rng(int(seed))
        contigs = [int(c) for c in contigs]
        self.ploidy, self.length, self.slack, self.pad_char = int(ploidy), int(length), int(slack), ord("N")
        ro = np.zeros(len(contigs) + 1, np.int64)
        ro[1:] = np.cumsum(contigs)
        total = int(ro[-1])
        self.ref_offsets = torch.from_numpy(ro).to(d)
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=d)

        # ---- reference: uniform ACGT, n_frac N; in slabs so that the temporaries stay small
        ref = torch.empty(total, dtype=torch.uint8, device=d)
        slab = 1 << 27
        for s in range(0, total, slab):
            n = min(slab, total - s)
            codes = torch.randint(0, 4, (n,), dtype=torch.int32, device=d, generator=g)
            seg = lut.index_select(0, codes)
            if n_frac > 0:
                seg[torch.rand(n, device=d, generator=g) < n_frac] = ord("N")
            ref[s:s + n] = seg
        self.ref = ref

        # ---- variant table, contig by contig (positions sorted inside a contig)
        vs_l, il_l, alt_l, alen_l, vc_l = [], [], [], [], []
        for c, clen in enumerate(contigs):
            n = int(nrng.poisson(clen * density))
            if n == 0:
                continue
            pos = torch.unique(torch.randint(0, clen, (n,), dtype=torch.int64, device=d, generator=g))
            n = int(pos.numel())
            u = torch.rand(n, device=d, generator=g).clamp_min(1e-12)
            mag = (torch.floor(torch.log(u) / np.log(1.0 - 0.35)) + 1).clamp(1, max_indel).to(torch.int64)
            sign = torch.where(torch.rand(n, device=d, generator=g) < 0.5, -1, 1)
            is_indel = torch.rand(n, device=d, generator=g) < indel_frac
            il = torch.where(is_indel, mag * sign, torch.zeros_like(mag))
            room = clen - 1 - pos                           # a deletion may not run past the contig end
            il = torch.where(il < 0, -torch.minimum(-il, room), il)
            alen = 1 + il.clamp_min(0)
            off = torch.zeros(n + 1, dtype=torch.int64, device=d)
            torch.cumsum(alen, 0, out=off[1:])
            alt = lut.index_select(0, torch.randint(0, 4, (int(off[-1]),), dtype=torch.int32, device=d, generator=g))
            base = ref[int(ro[c]) + pos]
            code = ((base == ord("C")).to(torch.int64) + 2 * (base == ord("G")).to(torch.int64)
                    + 3 * (base == ord("T")).to(torch.int64))          # N -> 0 like the numpy generator
            alt_snp = lut[(code + torch.randint(1, 4, (n,), dtype=torch.int64, device=d, generator=g)) % 4]
            alt[off[:-1]] = torch.where(il == 0, alt_snp, base)       # SNP: another base; indel: the anchor
            vs_l.append(pos.to(torch.int32)); il_l.append(il.to(torch.int32)); alt_l.append(alt)
            alen_l.append(alen); vc_l.append(torch.full((n,), c, dtype=torch.int32, device=d))
        cat = (lambda xs, dt: torch.cat(xs) if xs else torch.zeros(0, dtype=dt, device=d))
        self.v_starts, self.ilens = cat(vs_l, torch.int32), cat(il_l, torch.int32)
        self.alt_alleles = cat(alt_l, torch.uint8)
        v_contig = cat(vc_l, torch.int32)
        alen = cat(alen_l, torch.int64)
        nv = int(self.v_starts.numel())
        self.alt_offsets = torch.zeros(nv + 1, dtype=torch.int64, device=d)
        torch.cumsum(alen, 0, out=self.alt_offsets[1:])
        af = torch.from_numpy(nrng.beta(af_beta[0], af_beta[1], nv).astype(np.float32)).to(d)
        gpos = self.ref_offsets[v_contig.to(torch.int64)] + self.v_starts.to(torch.int64)   # sorted

        # ---- queries: regions drawn across the whole genome (contig ~ length)
        M, P = int(n_queries), self.ploidy
        span = self.length + 2 * self.slack
        w = torch.tensor(contigs, dtype=torch.float64, device=d)
        qc = torch.multinomial(w / w.sum(), M, replacement=True, generator=g)
        clen = torch.tensor(contigs, dtype=torch.int64, device=d)[qc]
        start = (torch.rand(M, device=d, generator=g, dtype=torch.float64) * (clen - span).clamp_min(1)).to(torch.int64)
        end = start + span
        strand = torch.where(torch.rand(M, device=d, generator=g) < rc_frac, -1, 1)
        self.full_regions = torch.stack([qc, start, end, strand], dim=1).to(torch.int32).contiguous()
        self.n_queries = M

        # ---- sparse genotypes: every haplotype carries each variant with pos in
        # [start - lookback, end) on its contig with that variant's allele frequency
        c_s, c_e = self.ref_offsets[qc], self.ref_offsets[qc + 1]
        lo = torch.searchsorted(gpos, torch.maximum(c_s + start - lookback, c_s))
        hi = torch.searchsorted(gpos, torch.minimum(c_s + end, c_e))
        K = M * P
        n_cand = (hi - lo).clamp_min(0).repeat_interleave(P)
        row_lo = lo.repeat_interleave(P)
        cand_off = torch.zeros(K + 1, dtype=torch.int64, device=d)
        torch.cumsum(n_cand, 0, out=cand_off[1:])
        tot = int(cand_off[-1])
        row_of = torch.repeat_interleave(torch.arange(K, device=d), n_cand)
        v_of = row_lo[row_of] + (torch.arange(tot, device=d) - cand_off[row_of])
        carried = torch.rand(tot, device=d, generator=g) < af[v_of]
        self.geno_v_idxs = v_of[carried].to(torch.int32)
        counts = torch.bincount(row_of[carried], minlength=K)
        offs = torch.zeros(K + 1, dtype=torch.int64, device=d)
        torch.cumsum(counts, 0, out=offs[1:])
        self.geno_offsets = torch.stack([offs[:-1], offs[1:]]).contiguous()
        self.mean_variants = float(self.geno_v_idxs.numel()) / max(K, 1)
        self._host_static = None

    # -- what HapsDevice takes
    def static_kwargs(self) -> dict:
        return dict(ref=self.ref, ref_offsets=self.ref_offsets, v_starts=self.v_starts, ilens=self.ilens,
                    alt_alleles=self.alt_alleles, alt_offsets=self.alt_offsets, geno_offsets=self.geno_offsets,
                    geno_v_idxs=self.geno_v_idxs, pad_char=self.pad_char)

    def nbytes(self) -> dict:
        t = lambda x: int(x.numel()) * x.element_size()
        return dict(reference=t(self.ref), variant_table=t(self.v_starts) + t(self.ilens) + t(self.alt_alleles)
                    + t(self.alt_offsets), genotype_csr=t(self.geno_offsets) + t(self.geno_v_idxs),
                    regions=t(self.full_regions))

    def draw_batches(self, n_batches: int, queries_per_batch: int, seed: int = 1):
        """``n_batches`` disjoint random sets of query ids (int64 (n_batches, b) on the device)."""
        import torch

        g = torch.Generator(device=self.device)
        g.manual_seed(int(seed))
        need = n_batches * queries_per_batch
        if need <= self.n_queries:
            q = torch.randperm(self.n_queries, device=self.device, generator=g)[:need]
        else:
            q = torch.randint(0, self.n_queries, (need,), device=self.device, generator=g)
        return q.view(n_batches, queries_per_batch)

    def request(self, q, rc: bool = True) -> dict:
        """Per-batch arrays (``ReconstructionRequest``) for query ids ``q``: shifts = 0."""
        import torch

        P = self.ploidy
        q = q.to(torch.int64)
        reg = self.full_regions.index_select(0, q)
        goi = (q * P)[:, None] + torch.arange(P, device=self.device, dtype=torch.int64)[None, :]
        to_rc = (reg[:, 3] == -1).repeat_interleave(P).to(torch.uint8) if rc else None
        return dict(regions=reg, shifts=torch.zeros((q.numel(), P), dtype=torch.int32, device=self.device),
                    geno_offset_idx=goi.contiguous(), to_rc=to_rc)

    def host_static(self) -> SynthStatic:
        if self._host_static is None:
            h = lambda x: x.cpu().numpy()
            self._host_static = SynthStatic(h(self.ref), h(self.ref_offsets), h(self.v_starts), h(self.ilens),
                                            h(self.alt_alleles), h(self.alt_offsets), np.zeros(0, np.int32),
                                            np.zeros(0, np.float32))
        return self._host_static

    def host_batch(self, q, rc: bool = True) -> SynthBatch:
        """The same batch for the CPU oracle: the rows' CSR slices compacted (geno_offset_idx = arange)."""
        import torch

        P = self.ploidy
        r = self.request(q, rc)
        goi = r["geno_offset_idx"].reshape(-1)
        o_s, o_e = self.geno_offsets[0][goi], self.geno_offsets[1][goi]
        n = o_e - o_s
        K = int(goi.numel())
        offs = torch.zeros(K + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(n, 0, out=offs[1:])
        row_of = torch.repeat_interleave(torch.arange(K, device=self.device), n)
        src = o_s[row_of] + (torch.arange(int(offs[-1]), device=self.device) - offs[row_of])
        gv = self.geno_v_idxs[src]
        go = torch.stack([offs[:-1], offs[1:]])
        h = lambda x: None if x is None else x.cpu().numpy()
        return SynthBatch(regions=h(r["regions"]), shifts=h(r["shifts"]),
                          geno_offset_idx=np.arange(K, dtype=np.int64).reshape(-1, P), geno_offsets=h(go),
                          geno_v_idxs=h(gv), to_rc=None if r["to_rc"] is None else h(r["to_rc"]).astype(bool),
                          output_length=self.length, meta=dict(B=K // P, P=P, length=self.length))


def make_genome(scale: str = "small", workload: str = "cfg3", device="cuda", seed: int | None = None,
                n_queries: int | None = None, contigs=None) -> GenomeDataset:
    """BASELINE.json cfg1/cfg2/cfg3 over a ``SCALES`` genome."""
    cfg = CONFIGS[workload]
    cg, nq = SCALES[scale]
    idx = int(workload[3:])
    return GenomeDataset(device=device, contigs=contigs or cg, n_queries=n_queries or nq, length=cfg["length"],
                         indel_frac=cfg["indel_frac"], rc_frac=cfg["rc_frac"],
                         seed=20260802 + idx if seed is None else seed)


# --- SVAR2 two-source form of a batch ------------------------------------------------
@dataclass
class SynthSvar2:
    """One batch as DECODED SVAR2 channels (``reconstruct_haplotypes_from_svar2``, src/ffi/mod.rs:874-893, with the key
    arguments in the form ``decode_alt`` gives them, src/svar2/mod.rs:17-30)."""

    vk_pos: np.ndarray
    vk_ilen: np.ndarray
    vk_alt_off: np.ndarray
    vk_off: np.ndarray
    dense_pos: np.ndarray
    dense_ilen: np.ndarray
    dense_alt_off: np.ndarray
    dense_range: np.ndarray
    dense_present: np.ndarray
    dense_present_off: np.ndarray
    alt_bytes: np.ndarray

    def args(self):
        """Positional arguments between ``shifts`` and ``ref_`` of the SVAR2 entry points."""
        return (self.vk_pos, self.vk_ilen, self.vk_alt_off, self.vk_off, self.dense_pos, self.dense_ilen, self.dense_alt_off,
                self.dense_range, self.dense_present, self.dense_present_off, self.alt_bytes)


def _alleles_of(st: SynthStatic, v: np.ndarray, base: int):
    """Decoded alleles of variants ``v``: ALT bytes, or EMPTY for a deletion (the generator's deletions carry the anchor base
    only, which is what the SVAR2 provider substitutes for a pure deletion's empty allele).  -> (offsets n + 1 from `base`, bytes)."""
    v = np.asarray(v, np.int64)
    a0 = st.alt_offsets[v]
    ln = np.where(st.ilens[v] < 0, 0, st.alt_offsets[v + 1] - a0).astype(np.int64)
    off = np.zeros(len(v) + 1, np.int64)
    np.cumsum(ln, out=off[1:])
    tot = int(off[-1])
    src = np.repeat(a0 - off[:-1], ln) + np.arange(tot)
    return off + base, st.alt_alleles[src] if tot else np.zeros(0, np.uint8)


def to_svar2(rng: np.random.Generator, st: SynthStatic, bt: SynthBatch, dense_af: float = 0.35, extra: float = 0.5) -> SynthSvar2:
    """The same haplotypes as ``(st, bt)`` in two-channel form: per query the variants with AF >= ``dense_af`` carried by any
    of its haplotypes -- plus ``extra`` x as many neighbours NO haplotype carries (absent bits) -- are its ``dense`` window
    (presence bits per haplotype); everything else a haplotype carries is a ``var_key`` call.  Windows and calls are in
    table order = position order inside a contig."""
    B, P = bt.geno_offset_idx.shape
    go, gv = bt.geno_offsets, bt.geno_v_idxs
    vk_l, vk_cnt, d_l, d_rng, bits_l, bits_off = [], [], [], [], [], [0]
    n_dense = 0
    nv = len(st.v_starts)
    for q in range(B):
        rows = [gv[go[0, o]:go[1, o]].astype(np.int64) for o in bt.geno_offset_idx[q]]
        union = np.unique(np.concatenate(rows)) if rows else np.zeros(0, np.int64)
        dense = union[st.af[union] >= dense_af] if len(union) else union
        if len(union) and extra > 0:
            lo, hi = int(union[0]), int(union[-1])
            cand = np.setdiff1d(np.arange(max(lo - 2, 0), min(hi + 3, nv)), union)
            cand = cand[st.v_contig[cand] == st.v_contig[union[0]]]
            k = min(len(cand), int(np.ceil(extra * max(len(dense), 1))))
            if k:
                dense = np.union1d(dense, rng.choice(cand, k, replace=False))
        d_l.append(dense)
        d_rng.append((n_dense, n_dense + len(dense)))
        n_dense += len(dense)
        for r in rows:
            in_d = np.isin(r, dense)
            vk_l.append(r[~in_d])
            vk_cnt.append(int((~in_d).sum()))
            bits_l.append(np.isin(dense, r))
            bits_off.append(bits_off[-1] + len(dense))
    vk = np.concatenate(vk_l) if vk_l else np.zeros(0, np.int64)
    dn = np.concatenate(d_l) if d_l else np.zeros(0, np.int64)
    bits = np.concatenate(bits_l) if bits_l else np.zeros(0, np.bool_)
    vk_off = np.zeros(B * P + 1, np.int64)
    np.cumsum(vk_cnt, out=vk_off[1:])
    vk_alt_off, vk_bytes = _alleles_of(st, vk, 0)
    d_alt_off, d_bytes = _alleles_of(st, dn, len(vk_bytes))
    return SynthSvar2(
        vk_pos=st.v_starts[vk].astype(np.int32), vk_ilen=st.ilens[vk].astype(np.int32), vk_alt_off=vk_alt_off, vk_off=vk_off,
        dense_pos=st.v_starts[dn].astype(np.int32), dense_ilen=st.ilens[dn].astype(np.int32), dense_alt_off=d_alt_off,
        dense_range=np.asarray(d_rng, np.int32).reshape(B, 2), dense_present=np.packbits(bits, bitorder="little"),
        dense_present_off=np.asarray(bits_off, np.int64), alt_bytes=np.concatenate([vk_bytes, d_bytes]))
