// gvl_svar2.hip -- the SVAR2 two-source variant provider: what the reference does per haplotype on the host with a
// Vec, a stable sort and a decode closure (merge_hap, src/svar2/mod.rs:45-72; the provider closures of
// src/reconstruct/mod.rs:662-735 and src/tracks/mod.rs:744-800) is here ONE launch per batch that writes the batch's merged
// variants as a sparse table of the SVAR1 shape -- per haplotype a contiguous run of 16-byte records, its slot line, and the
// variant table behind them -- so that every reconstruction / diff / track kernel of the library consumes a SVAR2 batch
// unchanged.  A haplotype's run lives at [vk_off[k] + dense_present_off[k], ...): the two prefix sums the caller already
// has bound every haplotype's record count, so no counting pass, no scan and no host round trip are needed.
//
// Two forms, picked per haplotype (svar2_merge_kernel): PACKED -- four haplotypes per wave, one per DPP row of 16 lanes, for
// haplotypes with at most 16 entries in both channels together (a training batch's 2 kb windows) -- and GENERAL -- one haplotype on all
// 64 lanes: tiles of 64 var_key entries and 64 window entries are merged round by round, a round emits every loaded entry
// that no entry still unloaded can precede (var_key first on ties), ranks the emitted entries by counting (readlane
// broadcast) and writes the records.  Ranks by counting are exact stable-sort semantics whatever the order inside a tile: a
// haplotype with <= 64 entries per channel needs no sortedness at all.
#include "gvl_internal.inc"

namespace {
#include "gvl_dev.inc"

struct Svar2Args {
    const int *vk_pos, *vk_ilen; const i64 *vk_alt_off, *vk_off; i64 n_vk;
    const int *dense_pos, *dense_ilen; const i64 *dense_alt_off; i64 n_dense;
    const int *dense_range; const u8 *dense_present; i64 present_bits; const i64 *dense_present_off;
    const u8 *alt_in; i64 alt_len;
    int filter_exonic;
    const int *regions; i64 regions_stride; i64 n_rows; int ploidy;
    const u8 *ref; i64 ref_len; const i64 *ref_offsets; int n_contigs; u32 pad;
    // the table (workspace)
    int *v_starts, *ilens; i64 *alt_offsets; gvl_vrec *vrec; int *geno_v_idxs; gvl_grec *grec;
    i64 *go_starts, *go_stops, *goi; gvl_srec *srec; u8 *alt_out;
    i64 cap;
    int *async_err;
    unsigned merge_blocks;      // workgroups [0, merge_blocks) merge; the ones behind them copy the allele pool into the table
};

struct Svar2Row {
    i64 k, go_start, c_s, c_len, rs, re;
};

// one merged record: entry (pos, il) with the allele alt_in[a0 .. a1) -> record m of the table, rank r inside its haplotype
__device__ __forceinline__ bool svar2_emit(const Svar2Args &A, const Svar2Row &R, i64 m, int r, int pos, int il, i64 a0, i64 a1) {
    i64 alen = a1 - a0;
    bool bad = false;
    if (alen < 0 || a0 < 0 || a1 > A.alt_len) { alen = 0; bad = true; }
    u32 inl = 0;
    i64 a_start;
    if (alen == 0) {
        // a pure deletion decodes to an empty allele; the walk needs the anchor base ref[pos] (src/reconstruct/mod.rs:712-733)
        const u32 b = (pos >= 0 && (i64)pos < R.c_len && R.c_s + pos < A.ref_len) ? (u32)A.ref[R.c_s + pos] : A.pad;
        a_start = A.alt_len + m;
        A.alt_out[a_start] = (u8)b;
        inl = b;
        alen = 1;
    } else {
        a_start = a0;
        const int n4 = alen < 4 ? (int)alen : 4;
        for (int i = 0; i < n4; ++i) inl |= (u32)A.alt_in[a0 + i] << (8 * i);
    }
    // (the SoA half of the table -- what the CSR routes without inline records and the scalar walk read -- costs 2.6 of the launch's
    // 23 us: measured with these four stores compiled out)
    A.v_starts[m] = pos;
    A.ilens[m] = il;
    A.alt_offsets[m] = a_start;
    A.geno_v_idxs[m] = (int)m;
    i32x4 v;
    v.x = pos; v.y = il; v.z = (int)(alen > 2147483647ll ? 2147483647ll : alen); v.w = (int)inl;
    *reinterpret_cast<i32x4 *>(A.vrec + m) = v;
    const u32 al24 = alen > 0xFFFFFFll ? 0xFFFFFFu : (u32)alen;                      // 0xFFFFFF = "ask vrec" (pack_genotypes_kernel)
    i32x4 g;
    g.x = pos; g.y = il; g.z = (int)((al24 << 8) | (inl & 0xFFu)); g.w = (int)m;
    *reinterpret_cast<i32x4 *>(A.grec + m) = g;
    if (r < GVL_SLOT_RECS) {
        i32x4 s;
        s.x = pos; s.y = il; s.z = g.z; s.w = (int)(u32)(u64)a_start;
        *reinterpret_cast<i32x4 *>(A.srec + R.k * GVL_SLOT_RECS + r) = s;
    }
    if (bad && A.async_err) *A.async_err = 6;
    return alen >= 0xFFFFFFll;          // an allele too long for a slot record: the slot goes through the CSR
}

// ---- the general form: ONE haplotype on all 64 lanes, any number of entries, round by round -------------------------------
__device__ __forceinline__ void svar2_merge_one(const Svar2Args &A, const i64 k, const int lane) {
    const i64 q = k / A.ploidy;
    const int *reg = A.regions + q * A.regions_stride;
    Svar2Row R;
    R.k = k;
    int c = rfl(reg[0]);
    R.rs = rfl(reg[1]); R.re = rfl(reg[2]);
    c = (c >= 0 && c < A.n_contigs) ? c : 0;                       // (out of contract otherwise: clamp, like the kernels)
    // (every lane reads the contig's bounds itself -- the same two words -- and nothing waits for them before a record is emitted: the
    // channel tiles below are requested in the same round trip)
    R.c_s = A.ref_offsets[c];
    R.c_len = A.ref_offsets[c + 1] - R.c_s;
    i64 vk_lo = rfl64(A.vk_off[k]), vk_hi = rfl64(A.vk_off[k + 1]);
    i64 ds = rfl(A.dense_range[2 * q]), de = rfl(A.dense_range[2 * q + 1]);
    const i64 bb = rfl64(A.dense_present_off[k]), bb1 = rfl64(A.dense_present_off[k + 1]);
    const bool ok = vk_lo >= 0 && vk_hi >= vk_lo && vk_hi <= A.n_vk && ds >= 0 && de >= ds && de <= A.n_dense && bb >= 0 &&
                    bb + (de - ds) <= A.present_bits && bb1 - bb >= de - ds && vk_lo + bb + (vk_hi - vk_lo) + (de - ds) <= A.cap;
    if (!ok) {
        if (lane == 0 && A.async_err) *A.async_err = 6;
        vk_lo = vk_hi = 0; ds = de = 0;
    }
    R.go_start = ok ? vk_lo + bb : 0;
    const bool multi = (vk_hi - vk_lo > WAVE) || (de - ds > WAVE);
    i64 ia = vk_lo, jb = ds;
    int outn = 0;
    bool anybig = false;
    u32 prevA = 0, prevB = 0;
    int err = 0;
    while (ia < vk_hi || jb < de) {
        const int nA = (int)(vk_hi - ia < WAVE ? vk_hi - ia : WAVE), nB = (int)(de - jb < WAVE ? de - jb : WAVE);
        const bool hasA = lane < nA, hasB = lane < nB;
        int posA = 0, ilA = 0, posB = 0, ilB = 0;
        u32 bit = 0;
        i64 a0A = 0, a1A = 0, a0B = 0, a1B = 0;        // (the alleles' bounds: requested with the tile, not behind the ranks)
        if (hasA) {
            posA = A.vk_pos[ia + lane]; ilA = A.vk_ilen[ia + lane];
            a0A = A.vk_alt_off[ia + lane]; a1A = A.vk_alt_off[ia + lane + 1];
        }
        if (hasB) {
            posB = A.dense_pos[jb + lane]; ilB = A.dense_ilen[jb + lane];
            a0B = A.dense_alt_off[jb + lane]; a1B = A.dense_alt_off[jb + lane + 1];
            const i64 b = bb + (jb - ds) + lane;
            bit = ((u32)A.dense_present[b >> 3] >> (u32)(b & 7)) & 1u;                  // LSB first (src/svar2/mod.rs:35-39)
        }
        if ((hasA && posA < 0) || (hasB && bit && posB < 0)) err = 5;
        bool eligA = hasA, eligB = hasB && bit;
        if (A.filter_exonic) {              // src/reconstruct/mod.rs:699-706: the entry lies entirely inside [start, end)
            const i64 eA = (i64)(u32)posA - (ilA < 0 ? (i64)ilA : 0) + 1, eB = (i64)(u32)posB - (ilB < 0 ? (i64)ilB : 0) + 1;
            eligA = eligA && (i64)(u32)posA >= R.rs && eA <= R.re;
            eligB = eligB && (i64)(u32)posB >= R.rs && eB <= R.re;
        }
        // what this round may emit: no unloaded entry of the OTHER channel can precede it (var_key first on ties)
        const bool moreA = ia + nA < vk_hi, moreB = jb + nB < de;
        const u64 LA = (moreA && nA > 0) ? (u64)(u32)rdl(posA, nA - 1) : (1ull << 32);
        const u64 LB = (moreB && nB > 0) ? (u64)(u32)rdl(posB, nB - 1) : (1ull << 32);
        const bool emitA = hasA && (u64)(u32)posA <= LB, emitB = hasB && (u64)(u32)posB < LA;
        if (multi) {
            // the round logic needs position-sorted channels (genoray's invariant, src/svar2/mod.rs:373-378): report anything else
            const u32 pA = (u32)dpp_mov<0x138, 0xf>(0, posA), pB = (u32)dpp_mov<0x138, 0xf>(0, posB);
            const u32 qA = lane == 0 ? prevA : pA, qB = lane == 0 ? prevB : pB;
            if ((hasA && (u32)posA < qA) || (hasB && (u32)posB < qB)) err = 4;
        }
        const u64 EA = __builtin_amdgcn_ballot_w64(emitA && eligA), EB = __builtin_amdgcn_ballot_w64(emitB && eligB);
        // rank by counting: stable sort by position, var_key ahead of dense on ties, tile order inside a channel
        int rA = 0, rB = 0;
        const int T = nA > nB ? nA : nB;
        for (int t = 0; t < T; ++t) {
            if ((EA >> t) & 1ull) {
                const u32 pa = (u32)rdl(posA, t);
                rA += (pa < (u32)posA || (pa == (u32)posA && t < lane)) ? 1 : 0;
                rB += (pa <= (u32)posB) ? 1 : 0;
            }
            if ((EB >> t) & 1ull) {
                const u32 pb = (u32)rdl(posB, t);
                rA += (pb < (u32)posA) ? 1 : 0;
                rB += (pb < (u32)posB || (pb == (u32)posB && t < lane)) ? 1 : 0;
            }
        }
        bool big = false;
        if (emitA && eligA) big = svar2_emit(A, R, R.go_start + outn + rA, outn + rA, posA, ilA, a0A, a1A) || big;
        if (emitB && eligB) big = svar2_emit(A, R, R.go_start + outn + rB, outn + rB, posB, ilB, a0B, a1B) || big;
        anybig = anybig || __builtin_amdgcn_ballot_w64(big) != 0;
        outn += __builtin_popcountll(EA) + __builtin_popcountll(EB);
        int doneA = __builtin_popcountll(__builtin_amdgcn_ballot_w64(emitA)), doneB = __builtin_popcountll(__builtin_amdgcn_ballot_w64(emitB));
        if (doneA + doneB == 0) { doneA = nA; doneB = nB; err = 4; }       // (only unsorted input stalls a round: consume, report)
        if (doneA > 0) prevA = (u32)rdl(posA, doneA - 1);
        if (doneB > 0) prevB = (u32)rdl(posB, doneB - 1);
        ia += doneA; jb += doneB;
    }
    if (__builtin_amdgcn_ballot_w64(err != 0) != 0 && A.async_err) {
        const int e = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(err == 5) ? 5 : 4);
        if (lane == 0) *A.async_err = e;
    }
    if (lane == 0) {
        A.go_starts[k] = R.go_start;
        A.go_stops[k] = R.go_start + outn;
        A.goi[k] = k;
    }
    // the slot line: entries the records did not fill are EMPTY; a haplotype with more than GVL_SLOT_RECS records (or an allele
    // the 24-bit length field cannot hold) is marked OVERFLOW in entry 0 -- behind the records' own stores
    if (lane < GVL_SLOT_RECS && lane >= outn) {
        i32x4 s;
        s.x = 0; s.y = 0; s.z = (int)GVL_SREC_EMPTY; s.w = 0;
        *reinterpret_cast<i32x4 *>(A.srec + k * GVL_SLOT_RECS + lane) = s;
    }
    if (outn > GVL_SLOT_RECS || anybig) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            i32x4 s;
            s.x = 0; s.y = 0; s.z = (int)GVL_SREC_OVERFLOW; s.w = 0;
            *reinterpret_cast<i32x4 *>(A.srec + k * GVL_SLOT_RECS) = s;
        }
    }
}

// ---- the packed form: FOUR haplotypes per wave, one per DPP row of 16 lanes -- lanes 0..7 of a row hold the haplotype's var_key
// entries, lanes 8..15 its window's entries.  What a training batch looks like (a 2 kb window sees a handful of variants): a wave
// per haplotype spent its 64 lanes and ~600 instructions on five records.  A lane holds at most one entry (the row's first lanes
// the var_key entries, the next ones the window's: at most 16 together); an entry's rank is counted over the 15 other lanes of
// its row (row_ror: a v_mov_dpp each), order = (position, lane): var_key lanes sit below the dense lanes, so ties come out
// var_key first and in channel order -- merge_hap's stable sort, whatever the order inside the channels.  A haplotype that does
// not fit a row (more than 16 entries, offsets out of range) takes the general form behind the wave's packed rows.
template <int N>
__device__ __forceinline__ int row_ror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xf, 0xf, false); }

template <int N>
__device__ __forceinline__ void svar2_rank_step(int &rank, const u32 pos, const int sub, const int key) {
    // key of a lane: (eligible << 31 is not needed: ineligible lanes carry position 0xFFFFFFFF and the flag in `key`'s low bit)
    const int o = row_ror<N>(key);                  // the lane N places below (cyclically) inside the row
    const u32 opos = (u32)row_ror<N>((int)pos);
    const int osub = (sub - N) & 15;
    rank += ((o & 1) && (opos < pos || (opos == pos && osub < sub))) ? 1 : 0;
}

__global__ __launch_bounds__(256) void svar2_merge_kernel(const Svar2Args A) {
    if (blockIdx.x >= A.merge_blocks) {
        // the table's allele pool = the caller's (allele starts stay what they are; the anchor bytes of pure deletions go behind it):
        // copied by the launch's last workgroups, 16 bytes a lane (a separate hipMemcpyAsync was 5 of the launch's 25 us)
        const i64 t = (i64)(blockIdx.x - A.merge_blocks) * blockDim.x + threadIdx.x;
        const i64 b0 = t * 16;
        if (b0 >= A.alt_len) return;
        if (b0 + 16 <= A.alt_len && (((uintptr_t)A.alt_in) & 15) == 0) {
            *reinterpret_cast<u32x4 *>(A.alt_out + b0) = *reinterpret_cast<const u32x4 *>(A.alt_in + b0);
        } else {
            for (i64 i = b0; i < b0 + 16 && i < A.alt_len; ++i) A.alt_out[i] = A.alt_in[i];
        }
        return;
    }
    const int lane = threadIdx.x & (WAVE - 1);
    const i64 w = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const i64 k0 = w * 4;
    if (k0 >= A.n_rows) return;
    const int g = lane >> 4, sub = lane & 15;
    const i64 k = k0 + g;
    const bool valid = k < A.n_rows;
    const i64 kk = valid ? k : k0;
    const i64 q = kk / A.ploidy;
    const int *reg = A.regions + q * A.regions_stride;
    int c = reg[0];
    const i64 rs = reg[1], re = reg[2];
    c = (c >= 0 && c < A.n_contigs) ? c : 0;
    const i64 vk_lo = A.vk_off[kk], vk_hi = A.vk_off[kk + 1];
    const i64 ds = A.dense_range[2 * q], de = A.dense_range[2 * q + 1];
    const i64 bb = A.dense_present_off[kk], bb1 = A.dense_present_off[kk + 1];
    const bool ok = vk_lo >= 0 && vk_hi >= vk_lo && vk_hi <= A.n_vk && ds >= 0 && de >= ds && de <= A.n_dense && bb >= 0 &&
                    bb + (de - ds) <= A.present_bits && bb1 - bb >= de - ds && vk_lo + bb + (vk_hi - vk_lo) + (de - ds) <= A.cap;
    // (a row holds a haplotype whose two channels have at most 16 entries together: var_key entries first, then the window's)
    const int nA = (int)(vk_hi - vk_lo);
    const bool fits = valid && ok && vk_hi - vk_lo <= 16 && de - ds <= 16 && (vk_hi - vk_lo) + (de - ds) <= 16;
    const u64 nofit = __builtin_amdgcn_ballot_w64(valid && !fits);
    Svar2Row R;
    R.k = kk; R.rs = rs; R.re = re;
    R.c_s = A.ref_offsets[c];
    R.c_len = A.ref_offsets[c + 1] - R.c_s;
    R.go_start = vk_lo + bb;
    // ---- this lane's entry
    const bool isA = sub < nA;
    const int e = isA ? sub : sub - nA;
    const bool has = fits && (isA || ds + e < de);
    int pos = -1, il = 0;
    i64 a0 = 0, a1 = 0;
    u32 bit = 1;
    if (has) {
        if (isA) {
            const i64 i = vk_lo + e;
            pos = A.vk_pos[i]; il = A.vk_ilen[i]; a0 = A.vk_alt_off[i]; a1 = A.vk_alt_off[i + 1];
        } else {
            const i64 j = ds + e;
            pos = A.dense_pos[j]; il = A.dense_ilen[j]; a0 = A.dense_alt_off[j]; a1 = A.dense_alt_off[j + 1];
            const i64 b = bb + e;
            bit = ((u32)A.dense_present[b >> 3] >> (u32)(b & 7)) & 1u;                  // LSB first (src/svar2/mod.rs:35-39)
        }
    }
    bool elig = has && bit;
    if (elig && pos < 0 && A.async_err) *A.async_err = 5;
    if (A.filter_exonic) {
        const i64 end = (i64)(u32)pos - (il < 0 ? (i64)il : 0) + 1;
        elig = elig && (i64)(u32)pos >= R.rs && end <= R.re;
    }
    const int key = elig ? 1 : 0;
    const u32 upos = (u32)pos;
    int rank = 0;
    svar2_rank_step<1>(rank, upos, sub, key);  svar2_rank_step<2>(rank, upos, sub, key);  svar2_rank_step<3>(rank, upos, sub, key);
    svar2_rank_step<4>(rank, upos, sub, key);  svar2_rank_step<5>(rank, upos, sub, key);  svar2_rank_step<6>(rank, upos, sub, key);
    svar2_rank_step<7>(rank, upos, sub, key);  svar2_rank_step<8>(rank, upos, sub, key);  svar2_rank_step<9>(rank, upos, sub, key);
    svar2_rank_step<10>(rank, upos, sub, key); svar2_rank_step<11>(rank, upos, sub, key); svar2_rank_step<12>(rank, upos, sub, key);
    svar2_rank_step<13>(rank, upos, sub, key); svar2_rank_step<14>(rank, upos, sub, key); svar2_rank_step<15>(rank, upos, sub, key);
    const u64 E = __builtin_amdgcn_ballot_w64(elig);
    const int total = __builtin_popcount((u32)(E >> (16 * g)) & 0xFFFFu);
    bool big = false;
    if (elig) big = svar2_emit(A, R, R.go_start + rank, rank, pos, il, a0, a1);
    const bool over = total > GVL_SLOT_RECS || ((__builtin_amdgcn_ballot_w64(big) >> (16 * g)) & 0xFFFFull) != 0;
    if (fits && sub == 0) {
        A.go_starts[k] = R.go_start;
        A.go_stops[k] = R.go_start + total;
        A.goi[k] = k;
    }
    if (fits && sub < GVL_SLOT_RECS && sub >= total) {
        i32x4 sr;
        sr.x = 0; sr.y = 0; sr.z = (int)GVL_SREC_EMPTY; sr.w = 0;
        *reinterpret_cast<i32x4 *>(A.srec + k * GVL_SLOT_RECS + sub) = sr;
    }
    if (__builtin_amdgcn_ballot_w64(over) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (fits && over && sub == 0) {
            i32x4 sr;
            sr.x = 0; sr.y = 0; sr.z = (int)GVL_SREC_OVERFLOW; sr.w = 0;
            *reinterpret_cast<i32x4 *>(A.srec + k * GVL_SLOT_RECS) = sr;
        }
    }
    // the haplotypes of this wave that did not fit a row (more than 16 entries, offsets out of range): the general form, one by one
    if (nofit != 0) {
        for (int j = 0; j < 4; ++j)
            if ((nofit >> (16 * j)) & 1ull) svar2_merge_one(A, k0 + j, lane);
    }
}

struct Svar2Layout {
    i64 v_starts, ilens, alt_offsets, vrec, geno_v_idxs, grec, go_starts, go_stops, goi, srec, alt, total;
};
i64 a256(i64 x) { return (x + 255) & ~255ll; }
Svar2Layout svar2_layout(i64 n_rows, i64 cap, i64 alt_len) {
    Svar2Layout L;
    i64 o = 0;
    const i64 c1 = cap > 0 ? cap : 1, r1 = n_rows > 0 ? n_rows : 1;
    L.v_starts = o; o += a256(c1 * 4);
    L.ilens = o; o += a256(c1 * 4);
    L.alt_offsets = o; o += a256((c1 + 1) * 8);
    L.vrec = o; o += a256(c1 * 16);
    L.geno_v_idxs = o; o += a256(c1 * 4);
    L.grec = o; o += a256(c1 * 16);
    L.go_starts = o; o += a256(r1 * 8);
    L.go_stops = o; o += a256(r1 * 8);
    L.goi = o; o += a256(r1 * 8);
    L.srec = o; o += a256(r1 * GVL_SLOT_RECS * 16);
    L.alt = o; o += a256(alt_len + c1 + 16);
    L.total = o;
    return L;
}

}  // namespace

extern "C" {

int64_t gvl_svar2_workspace_bytes(int64_t batch, int64_t ploidy, int64_t n_vk, int64_t dense_present_bits, int64_t alt_len) {
    if (batch < 0 || ploidy <= 0 || n_vk < 0 || dense_present_bits < 0 || alt_len < 0) return 0;
    return svar2_layout(batch * ploidy, n_vk + dense_present_bits, alt_len).total;
}

int gvl_svar2_merge(const gvl_static *st, const gvl_svar2_batch *sv, const int32_t *regions, int64_t regions_stride,
                    int64_t batch, int64_t ploidy, void *workspace, int64_t workspace_bytes, gvl_static *merged,
                    const int64_t **geno_offset_idx, void *stream) {
    if (!st || !sv || !merged || !geno_offset_idx) return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: NULL argument");
    if (batch < 0 || ploidy <= 0 || sv->n_vk < 0 || sv->n_dense < 0 || sv->dense_present_bits < 0 || sv->alt_len < 0)
        return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: negative size");
    const i64 n_rows = batch * ploidy;
    const i64 cap = sv->n_vk + sv->dense_present_bits;
    if (n_rows > 0x7FFFFFF0ll || cap > 0x7FFFFFF0ll) return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: batch too large (rows, entries < 2^31)");
    if (sv->alt_len + cap >= (1ll << 32)) return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_svar2_merge: allele pool of 4 GiB or more");
    const Svar2Layout L = svar2_layout(n_rows, cap, sv->alt_len);
    if (!workspace || workspace_bytes < L.total || ((uintptr_t)workspace & 255))
        return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: workspace missing, misaligned or smaller than gvl_svar2_workspace_bytes()");
    if ((st->ref_len > 0 && !st->ref) || !st->ref_offsets) return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: NULL reference arrays");
    if (n_rows > 0 && (!regions || regions_stride < 3 || !sv->vk_off || !sv->dense_range || !sv->dense_present_off))
        return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: NULL/invalid batch array");
    if (sv->n_vk > 0 && (!sv->vk_pos || !sv->vk_ilen || !sv->vk_alt_off)) return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: NULL var_key channel");
    if (sv->n_dense > 0 && (!sv->dense_pos || !sv->dense_ilen || !sv->dense_alt_off)) return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: NULL dense channel");
    if (sv->dense_present_bits > 0 && !sv->dense_present) return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: NULL presence bits");
    if (sv->alt_len > 0 && !sv->alt_bytes) return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: NULL allele pool");
    u8 *w = (u8 *)workspace;
    hipStream_t s = (hipStream_t)stream;
    Svar2Args A;
    memset(&A, 0, sizeof(A));
    A.vk_pos = sv->vk_pos; A.vk_ilen = sv->vk_ilen; A.vk_alt_off = (const i64 *)sv->vk_alt_off; A.vk_off = (const i64 *)sv->vk_off; A.n_vk = sv->n_vk;
    A.dense_pos = sv->dense_pos; A.dense_ilen = sv->dense_ilen; A.dense_alt_off = (const i64 *)sv->dense_alt_off; A.n_dense = sv->n_dense;
    A.dense_range = sv->dense_range; A.dense_present = sv->dense_present; A.present_bits = sv->dense_present_bits;
    A.dense_present_off = (const i64 *)sv->dense_present_off;
    A.alt_in = sv->alt_bytes; A.alt_len = sv->alt_len; A.filter_exonic = sv->filter_exonic;
    A.regions = regions; A.regions_stride = regions_stride; A.n_rows = n_rows; A.ploidy = (int)ploidy;
    A.ref = st->ref; A.ref_len = st->ref_len; A.ref_offsets = (const i64 *)st->ref_offsets;
    A.n_contigs = (int)(st->n_contigs < 0 ? 0 : (st->n_contigs > 0x7FFFFFFFll ? 0x7FFFFFFF : st->n_contigs));
    A.pad = st->pad_char;
    A.v_starts = (int *)(w + L.v_starts); A.ilens = (int *)(w + L.ilens); A.alt_offsets = (i64 *)(w + L.alt_offsets);
    A.vrec = (gvl_vrec *)(w + L.vrec); A.geno_v_idxs = (int *)(w + L.geno_v_idxs); A.grec = (gvl_grec *)(w + L.grec);
    A.go_starts = (i64 *)(w + L.go_starts); A.go_stops = (i64 *)(w + L.go_stops); A.goi = (i64 *)(w + L.goi);
    A.srec = (gvl_srec *)(w + L.srec); A.alt_out = w + L.alt;
    A.cap = cap;
    A.async_err = async_err_word();
    {
        const i64 mb = (((n_rows + 3) / 4) * WAVE + 255) / 256;        // (a wave per four haplotypes)
        const i64 cb = (sv->alt_len + 4095) / 4096;                      // ... and the pool's copy behind them, 4 KB a workgroup
        if (mb + cb > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: batch too large for one launch");
        A.merge_blocks = (unsigned)mb;
        const i64 grid = mb + cb;
        if (grid > 0) svar2_merge_kernel<<<dim3((unsigned)grid), dim3(256), 0, s>>>(A);
        const int rc = check_launch("gvl_svar2_merge");
        if (rc) return rc;
    }
    gvl_static m = *st;
    m.v_starts = A.v_starts; m.ilens = A.ilens; m.alt_offsets = (const int64_t *)A.alt_offsets; m.alt_alleles = A.alt_out;
    m.n_variants = cap > 0 ? cap : 0; m.alt_len = sv->alt_len + cap;
    m.vrec = A.vrec;
    m.geno_o_starts = (const int64_t *)A.go_starts; m.geno_o_stops = (const int64_t *)A.go_stops; m.n_geno_offsets = n_rows;
    m.geno_v_idxs = A.geno_v_idxs; m.n_geno = cap;
    m.geno_rec = A.grec; m.slot_rec = A.srec; m.slot_vidx = nullptr;
    *merged = m;
    *geno_offset_idx = (const int64_t *)A.goi;
    return GVL_OK;
}

int gvl_svar2_reconstruct(const gvl_static *st, const gvl_svar2_batch *sv, const gvl_batch *bt, const gvl_out *out,
                          void *workspace, int64_t workspace_bytes, void *stream) {
    if (!st || !sv || !bt || !out) return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_reconstruct: NULL struct");
    if (bt->keep || bt->keep_offsets || out->annot_v_idxs || out->annot_ref_pos)
        return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_svar2_reconstruct: the SVAR2 path has no keep masks or annotations (src/reconstruct/mod.rs:746-749)");
    gvl_static merged;
    const int64_t *goi = nullptr;
    int rc = gvl_svar2_merge(st, sv, bt->regions, bt->regions_stride, bt->batch, bt->ploidy, workspace, workspace_bytes, &merged, &goi, stream);
    if (rc) return rc;
    gvl_batch b = *bt;
    b.geno_offset_idx = goi;
    b.hap_plan = nullptr;
    return gvl_reconstruct(&merged, &b, out, stream);
}

}  // extern "C"
