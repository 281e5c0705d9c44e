// gvl_hip.hip -- MI355X (gfx950 / CDNA4) kernels + C-ABI for the GenVarLoader
// haplotype hot path: apply SNPs+indels to a reference window -> optional
// reverse-complement -> uint8 one-hot.  Written for wave64 / gfx950 only.
//
// What it replaces in the reference (file:line under /root/reference):
//   src/reconstruct/mod.rs:39-256   reconstruct_haplotype_core   (walk + copies)
//   src/reconstruct/mod.rs:280-583  SVAR1 provider + batch driver
//   src/genotypes/mod.rs:15-125     get_diffs_sparse
//   src/reverse.rs:25-69            rc_row / rc_flat_rows / reverse_flat_rows
//   src/reference/mod.rs:9-120      padded_slice / get_reference
//   src/ffi/mod.rs:722-860          reconstruct_haplotypes_fused orchestration
//   docs/source/index.md:109-119    user-side seqpro one-hot
//
// Design (DESIGN.md has the long form).  The reference walks a row's variants
// sequentially and memcpy's reference/allele runs.  Here one WAVE owns one
// (row, chunk) of output:
//   1. lanes gather the row's variant records in parallel (one 16-B packed
//      record + one 8-B allele offset per variant, 64 variants per trip);
//   2. the wave replays the reference's sequential walk on the scalar unit
//      (v_readlane -> SGPR state), but instead of copying bytes it emits
//      SEGMENTS (out_start, kind, source delta) into a 64-entry lane-resident
//      table; pure SNPs do not split a reference run, they become PATCHES;
//   3. all 64 lanes then stream the output: 4 bases per lane per trip, one
//      unaligned dword load of reference bytes (256 B per wave-load), SNP
//      patches applied in registers, reverse-complement folded into the store
//      index + LUT, one-hot through a 256-entry LDS LUT, one 16-B store per lane
//      (1 KiB contiguous per wave-store).
// There is no second pass over HBM for RC or one-hot and no intermediate
// haplotype buffer unless the caller asks for the bytes too.  Integer
// gather/scatter: HBM-bound, no MFMA.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "gvl_hip.h"

namespace {

typedef long long i64;
typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned char u8;

constexpr int WAVE = 64;
constexpr int WG_WAVES = 4;
constexpr int WG_THREADS = WAVE * WG_WAVES;
constexpr int GROUP = 4;                   // bases per lane per trip
constexpr int TRIP = WAVE * GROUP;         // 256 bases per wave trip
constexpr int SEG_CAP = 64;                // lane-resident segment table
constexpr int SEG_FLUSH = 59;              // flush before a step could overflow
constexpr int PATCH_FLUSH = 62;

enum : u32 { K_REF = 0, K_ALLELE = 1, K_PAD_LEAD = 2, K_PAD_TRAIL = 3 };
constexpr i64 DELTA_BIAS = 1ll << 40;      // src - out_start + BIAS fits 42 bits

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(4))) u32x4_a4 { u32 x, y, z, w; };
struct __attribute__((aligned(4))) i32x4_a4 { int x, y, z, w; };

__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ i64 rfl64(i64 x) {
    u32 lo = (u32)rfl((int)(u32)(u64)x);
    u32 hi = (u32)rfl((int)(u32)((u64)x >> 32));
    return (i64)(((u64)hi << 32) | lo);
}
__device__ __forceinline__ int rdl(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ i64 rdl64(int lo, int hi, int l) {
    return (i64)(((u64)(u32)rdl(hi, l) << 32) | (u32)rdl(lo, l));
}
__device__ __forceinline__ int bperm(int idx, int x) {
    return __builtin_amdgcn_ds_bpermute(idx << 2, x);
}
__device__ __forceinline__ i64 imin(i64 a, i64 b) { return a < b ? a : b; }
__device__ __forceinline__ i64 imax(i64 a, i64 b) { return a > b ? a : b; }

__device__ __forceinline__ u32 load_u32_unaligned(const u8 *p) {
    u32 v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// reverse.rs:45-53: b ^= (isAT & 0x15) ^ (isCG & 0x04)
__device__ __host__ __forceinline__ u32 comp_byte(u32 b) {
    u32 at = (b == 'A' || b == 'T') ? 0x15u : 0u;
    u32 cg = (b == 'C' || b == 'G') ? 0x04u : 0u;
    return b ^ at ^ cg;
}
// a10: out[..., j, a] = (byte == "ACGT"[a]) as one little-endian dword
__device__ __host__ __forceinline__ u32 onehot_dword(u32 b) {
    return b == 'A' ? 0x00000001u : b == 'C' ? 0x00000100u : b == 'G' ? 0x00010000u
         : b == 'T' ? 0x01000000u : 0u;
}

// LDS tables shared by the wave's lanes: [0] one-hot, [1] one-hot of the
// complement, [2] complement byte.
struct Luts { u32 oh[256]; u32 oh_rc[256]; u32 comp[256]; };

__device__ __forceinline__ void init_luts(Luts &l) {
    for (int b = threadIdx.x; b < 256; b += blockDim.x) {
        u32 c = comp_byte((u32)b);
        l.oh[b] = onehot_dword((u32)b);
        l.oh_rc[b] = onehot_dword(c);
        l.comp[b] = c;
    }
    __syncthreads();
}

struct ReconArgs {
    // static
    const u8 *ref; i64 ref_len; const i64 *ref_offsets;
    const gvl_vrec *vrec; const i64 *alt_offsets; const u8 *alt_alleles; i64 alt_len;
    i64 n_variants;
    const i64 *go_starts; const i64 *go_stops; const int *geno_v_idxs;
    // batch
    const int *regions; i64 regions_stride; const int *shifts; const i64 *geno_offset_idx;
    const u8 *keep; const i64 *keep_offsets; const u8 *to_rc; const i64 *out_offsets;
    i64 fixed_len;      // >= 0 or -1
    i64 n_queries; int ploidy; int chunks; int chunk_len;
    int ref_only;       // get_reference mode: no variants, shift 0, row len from out_offsets
    u32 pad;
    // out
    u8 *haps; u8 *onehot; int onehot_cl; int *av; int *ap; i64 *out_offsets_w;
};

// Per-wave mirror of the segment table for the (rare, divergent) slow path.
struct SegMirror { int out[SEG_CAP]; u32 lo[SEG_CAP]; u32 hi[SEG_CAP]; int a[SEG_CAP]; int b[SEG_CAP]; };

template <bool ANNOT>
__global__ __launch_bounds__(WG_THREADS) void reconstruct_kernel(const ReconArgs A) {
    __shared__ Luts luts;
    __shared__ SegMirror mirror[WG_WAVES];
    init_luts(luts);

    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = rfl((int)(threadIdx.x >> 6));
    SegMirror &M = mirror[wave];

    // unit -> (query, chunk, hap): the haps of a query and neighbouring chunks of a
    // row sit in one workgroup so that they share the reference window in L1/L2.
    const i64 unit = (i64)blockIdx.x * WG_WAVES + wave;
    const i64 n_units = A.n_queries * A.chunks * A.ploidy;
    if (unit >= n_units) return;
    const int hap = (int)(unit % A.ploidy);
    const i64 qc = unit / A.ploidy;
    const int chunk = (int)(qc % A.chunks);
    const i64 query = qc / A.chunks;
    const i64 k = query * A.ploidy + hap;

    // ---- row parameters (level-1 loads; wave-uniform) -------------------------
    const int *reg = A.regions + query * A.regions_stride;
    const i64 c_idx = rfl(reg[0]);
    const i64 ref_start = rfl(reg[1]);
    const int reg_end = rfl(reg[2]);
    i64 shift = 0, o_idx = 0;
    if (!A.ref_only) {
        shift = rfl(A.shifts[k]);
        o_idx = rfl64(A.geno_offset_idx[k]);
    }
    const bool rc = A.to_rc ? (rfl((int)A.to_rc[k]) != 0) : false;
    i64 row_base, L;
    if (A.out_offsets) {
        row_base = rfl64(A.out_offsets[k]);
        L = rfl64(A.out_offsets[k + 1]) - row_base;
    } else {
        row_base = k * A.fixed_len;
        L = A.fixed_len;
    }
    if (A.out_offsets_w && chunk == 0 && lane == 0) {
        A.out_offsets_w[k] = row_base;
        if (k == A.n_queries * A.ploidy - 1) A.out_offsets_w[k + 1] = row_base + L;
    }
    const i64 lo_clip = (i64)chunk * A.chunk_len;
    const i64 hi_clip = imin(lo_clip + A.chunk_len, L);
    if (lo_clip >= L) return;

    // ---- level-2 loads ---------------------------------------------------------
    const i64 c_s = rfl64(A.ref_offsets[c_idx]);
    const i64 R = rfl64(A.ref_offsets[c_idx + 1]) - c_s;
    i64 o_s = 0, n_var = 0, keep_off = 0;
    if (!A.ref_only) {
        o_s = rfl64(A.go_starts[o_idx]);
        n_var = imax(rfl64(A.go_stops[o_idx]) - o_s, 0);
        if (A.keep && A.keep_offsets) keep_off = rfl64(A.keep_offsets[k]);
    }
    const bool has_keep = A.keep && A.keep_offsets;

    // ---- segment / patch tables (lane s holds entry s) ------------------------
    int s_out = 0; u32 s_lo = 0, s_hi = 0; int s_a = 0, s_b = 0;
    int p_out = 0, p_val = 0, p_id = 0;
    int nseg = 0, npatch = 0;
    u32 last_kind = 0xFFu; i64 last_delta = 0;

    auto push = [&](u32 kind, i64 o_start, i64 len, i64 src, int id, int vpos) {
        i64 s = imax(o_start, lo_clip), e = imin(o_start + len, hi_clip);
        if (e <= s) return;
        i64 delta = src - o_start;
        if (kind == K_REF && last_kind == K_REF && delta == last_delta) return;  // extends the open run
        u64 enc = (u64)(delta + DELTA_BIAS) | ((u64)kind << 62);
        if (lane == nseg) {
            s_out = (int)s; s_lo = (u32)enc; s_hi = (u32)(enc >> 32);
            M.out[lane] = (int)s; M.lo[lane] = (u32)enc; M.hi[lane] = (u32)(enc >> 32);
            if (ANNOT) { s_a = id; s_b = vpos; M.a[lane] = id; M.b[lane] = vpos; }
        }
        last_kind = kind; last_delta = delta; ++nseg;
    };
    auto push_patch = [&](i64 o_pos, int byte, int id) {
        if (o_pos < lo_clip || o_pos >= hi_clip) return;
        if (lane == npatch) { p_out = (int)o_pos; p_val = byte; if (ANNOT) p_id = id; }
        ++npatch;
    };

    // ---- walk state: reconstruct/mod.rs:61-83 -----------------------------------
    i64 ref_idx = ref_start, out_idx = 0, shifted = 0;
    bool ref_zero_fill = false;
    if (A.ref_only && ref_start >= (i64)reg_end) ref_zero_fill = true;  // reference/mod.rs:16-18
    if (ref_idx < 0) {
        i64 raw = -ref_idx;
        shifted = imin(shift, raw);
        i64 n = raw - shifted;
        push(K_PAD_LEAD, 0, n, 0, -1, -1);
        out_idx = n;
        ref_idx = 0;
    }

    // variant record registers for the current trip of 64 variants
    int r_pos = 0, r_ilen = 0, r_alen = 0, r_inl = 0, r_vi = 0, r_a0lo = 0, r_a0hi = 0, r_keep = 1;
    i64 vi = 0;          // next variant of the row
    i64 vb = -1;         // base of the loaded trip (-1: none)
    bool walk_done = false;
    i64 emit_pos = lo_clip;

    const u32 padb = A.pad & 0xFFu;
    u8 *hap_row = A.haps ? A.haps + row_base : nullptr;
    u8 *oh_row = A.onehot ? A.onehot + 4 * row_base : nullptr;
    int *av_row = (ANNOT && A.av) ? A.av + row_base : nullptr;
    int *ap_row = (ANNOT && A.ap) ? A.ap + row_base : nullptr;

    for (;;) {
        // =================== fill: replay the reference walk =====================
        while (!walk_done && nseg <= SEG_FLUSH && npatch <= PATCH_FLUSH) {
            bool stop = (vi >= n_var) || (out_idx >= hi_clip);
            if (!stop) {
                if (vb < 0 || vi - vb >= WAVE) {
                    // gather the next 64 variant records (levels 3 and 4)
                    vb = vi;
                    i64 j = vb + lane;
                    bool valid = j < n_var;
                    int v = valid ? A.geno_v_idxs[o_s + j] : 0;
                    v = v < 0 ? 0 : ((i64)v >= A.n_variants ? (int)(A.n_variants - 1) : v);
                    r_vi = v;
                    if (valid) {
                        const i32x4 rec = *reinterpret_cast<const i32x4 *>(A.vrec + v);
                        i64 a0 = A.alt_offsets[v];
                        r_pos = rec.x; r_ilen = rec.y; r_alen = rec.z; r_inl = rec.w;
                        r_a0lo = (int)(u32)(u64)a0; r_a0hi = (int)(u32)((u64)a0 >> 32);
                        r_keep = has_keep ? (int)A.keep[keep_off + j] : 1;
                    }
                }
                const int i = (int)(vi - vb);
                ++vi;
                // --- one step of reconstruct/mod.rs:85-198 ---
                if (has_keep && rdl(r_keep, i) == 0) continue;           // :86-90
                const i64 pos = rdl(r_pos, i);
                const i64 d = rdl(r_ilen, i);
                const i64 alen = rdl(r_alen, i);
                const i64 v_end = pos - imin(0, d) + 1;                   // :96
                if (pos < ref_start && d < 0 && v_end >= ref_start) {     // :99-102
                    ref_idx = v_end;
                    continue;
                }
                if (pos < ref_idx) continue;                              // :108-110
                i64 skip = 0;
                if (shifted < shift) {                                    // :115-146
                    const i64 dist = pos - ref_idx;
                    if (shifted + dist + alen < shift) continue;
                    if (shifted + dist >= shift) {
                        ref_idx += shift - shifted;
                        shifted = shift;
                    } else {
                        skip = shift - shifted - dist;
                        shifted = shift;
                        if (skip == alen) { ref_idx = v_end; continue; }
                        ref_idx = pos;
                    }
                }
                const i64 al = alen - skip;
                const i64 n = pos - ref_idx;
                if (out_idx + n >= L) {                                   // :154-158 (">=")
                    stop = true;
                } else {
                    push(K_REF, out_idx, n, c_s + ref_idx, -1, -1);
                    out_idx += n;
                    const i64 w = imin(al, L - out_idx);                  // :178
                    const int id = rdl(r_vi, i);
                    if (d == 0 && alen == 1 && w == 1) {
                        // pure SNP: the reference run continues, one byte is patched
                        push(K_REF, out_idx, 1, c_s + pos, -1, -1);
                        push_patch(out_idx, rdl(r_inl, i) & 0xFF, id);
                    } else {
                        push(K_ALLELE, out_idx, w, rdl64(r_a0lo, r_a0hi, i) + skip, id, (int)pos);
                    }
                    out_idx += w;
                    ref_idx = v_end;                                      // :193
                    if (out_idx >= L) stop = true;                        // :195-197
                }
            }
            if (stop) {
                // residual shift + tail: reconstruct/mod.rs:200-255
                if (shifted < shift) ref_idx = imin(ref_idx + (shift - shifted), R);
                const i64 u = L - out_idx;
                if (u > 0) {
                    const i64 w = imin(u, R - ref_idx);
                    i64 end = out_idx;
                    if (w > 0) { push(K_REF, out_idx, w, c_s + ref_idx, -1, -1); end += w; }
                    if (end < L) push(K_PAD_TRAIL, end, L - end, 0, -1, -1);
                }
                out_idx = imax(out_idx, L);
                walk_done = true;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // =================== emit [emit_pos, limit) ===============================
        const int cov = (int)imin(imax(out_idx, lo_clip), hi_clip);
        const int limit = walk_done ? (int)hi_clip : (cov & ~3);
        int sc = 0, pc = 0;
        for (int p0 = (int)emit_pos; p0 < limit; p0 += TRIP) {
            const int p = p0 + GROUP * lane;
            const bool act = p < limit;
            // segment holding p: scalar cursor + the few starts inside this trip
            while (sc + 1 < nseg && rdl(s_out, sc + 1) <= p0) ++sc;
            int idx = sc;
            for (int s = sc + 1; s < nseg; ++s) {
                const int st = rdl(s_out, s);
                if (st >= p0 + TRIP) break;
                idx += (p >= st) ? 1 : 0;
            }
            const u32 lo = (u32)bperm(idx, (int)s_lo);
            const u32 hi = (u32)bperm(idx, (int)s_hi);
            int nxt = bperm(idx + 1 < SEG_CAP ? idx + 1 : SEG_CAP - 1, s_out);
            if (idx + 1 >= nseg) nxt = cov;
            const u32 kind = hi >> 30;
            const i64 delta = (i64)((((u64)(hi & 0x3FFFFFFFu)) << 32) | lo) - DELTA_BIAS;
            const i64 src = delta + p;
            const bool full = act && (p + GROUP <= limit);
            const bool fast = full && kind == K_REF && (p + GROUP <= nxt) && src >= 0 &&
                              src + GROUP <= A.ref_len && !ref_zero_fill;
            u32 w = 0;
            int av4[GROUP], ap4[GROUP];
            if (fast) {
                w = load_u32_unaligned(A.ref + src);
                if (ANNOT) {
#pragma unroll
                    for (int i = 0; i < GROUP; ++i) { av4[i] = -1; ap4[i] = (int)(src - c_s) + i; }
                }
            } else if (act) {
                // slow path: group straddles a segment boundary, sits in an allele or a
                // pad run, or is the partial group at the end of the row
                int li = idx;
#pragma unroll
                for (int i = 0; i < GROUP; ++i) {
                    const int pp = p + i;
                    u32 b = 0; int a_v = -1, a_p = -1;
                    if (pp < limit) {
                        while (li + 1 < nseg && M.out[li + 1] <= pp) ++li;
                        const u32 l2 = M.lo[li], h2 = M.hi[li];
                        const u32 k2 = h2 >> 30;
                        const i64 s2 = (i64)((((u64)(h2 & 0x3FFFFFFFu)) << 32) | l2) - DELTA_BIAS + pp;
                        if (ref_zero_fill) {
                            b = 0;
                        } else if (k2 == K_REF) {
                            b = (s2 >= 0 && s2 < A.ref_len) ? (u32)A.ref[s2] : padb;
                            a_p = (int)(s2 - c_s);
                        } else if (k2 == K_ALLELE) {
                            b = (s2 >= 0 && s2 < A.alt_len) ? (u32)A.alt_alleles[s2] : padb;
                            if (ANNOT) { a_v = M.a[li]; a_p = M.b[li]; }
                        } else {
                            b = padb;
                            a_p = (k2 == K_PAD_LEAD) ? -1 : 2147483647;
                        }
                    }
                    w |= b << (8 * i);
                    if (ANNOT) { av4[i] = a_v; ap4[i] = a_p; }
                }
            }
            // SNP patches that land in this trip (sorted; scalar cursor)
            while (pc < npatch) {
                const int pp = rdl(p_out, pc);
                if (pp >= p0 + TRIP) break;
                const u32 pv = (u32)rdl(p_val, pc);
                const u32 dd = (u32)(pp - p);
                if (dd < (u32)GROUP) {
                    const u32 sh = dd * 8;
                    w = (w & ~(0xFFu << sh)) | (pv << sh);
                }
                if (ANNOT) {
                    const int pid = rdl(p_id, pc);
#pragma unroll
                    for (int i = 0; i < GROUP; ++i) if (dd == (u32)i) av4[i] = pid;
                }
                ++pc;
            }
            if (!act) continue;

            // ---- stores: RC folded into the index + LUT ---------------------------
            if (full) {
                const int jo = rc ? (int)(L - GROUP - p) : p;
                u32 ww = rc ? __builtin_bswap32(w) : w;
                const u32 b0 = ww & 0xFF, b1 = (ww >> 8) & 0xFF, b2 = (ww >> 16) & 0xFF, b3 = ww >> 24;
                if (oh_row) {
                    const u32 *t = rc ? luts.oh_rc : luts.oh;
                    if (!A.onehot_cl) {
                        u32x4_a4 o = {t[b0], t[b1], t[b2], t[b3]};
                        *reinterpret_cast<u32x4_a4 *>(oh_row + 4 * (i64)jo) = o;
                    } else {
                        // channel-major (rows, 4, L): plane a holds byte a of each one-hot dword
                        const u32 d0 = t[b0], d1 = t[b1], d2 = t[b2], d3 = t[b3];
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const u32 sh = 8 * a;
                            const u32 v = ((d0 >> sh) & 0xFF) | (((d1 >> sh) & 0xFF) << 8) |
                                          (((d2 >> sh) & 0xFF) << 16) | (((d3 >> sh) & 0xFF) << 24);
                            u8 *dst = oh_row + (i64)a * L + jo;
                            __builtin_memcpy(dst, &v, 4);
                        }
                    }
                }
                if (hap_row) {
                    u32 hv = ww;
                    if (rc) hv = luts.comp[b0] | (luts.comp[b1] << 8) | (luts.comp[b2] << 16) | (luts.comp[b3] << 24);
                    __builtin_memcpy(hap_row + jo, &hv, 4);
                }
                if (ANNOT) {
                    if (av_row) {
                        i32x4_a4 o = rc ? i32x4_a4{av4[3], av4[2], av4[1], av4[0]} : i32x4_a4{av4[0], av4[1], av4[2], av4[3]};
                        *reinterpret_cast<i32x4_a4 *>(av_row + jo) = o;
                    }
                    if (ap_row) {
                        i32x4_a4 o = rc ? i32x4_a4{ap4[3], ap4[2], ap4[1], ap4[0]} : i32x4_a4{ap4[0], ap4[1], ap4[2], ap4[3]};
                        *reinterpret_cast<i32x4_a4 *>(ap_row + jo) = o;
                    }
                }
            } else {
                // partial group at the row end: per-base stores
#pragma unroll
                for (int i = 0; i < GROUP; ++i) {
                    const int pp = p + i;
                    if (pp >= limit) break;
                    const u32 b = (w >> (8 * i)) & 0xFF;
                    const i64 jo = rc ? (L - 1 - pp) : (i64)pp;
                    if (oh_row) {
                        const u32 d = rc ? luts.oh_rc[b] : luts.oh[b];
                        if (!A.onehot_cl) {
                            __builtin_memcpy(oh_row + 4 * jo, &d, 4);
                        } else {
#pragma unroll
                            for (int a = 0; a < 4; ++a) oh_row[(i64)a * L + jo] = (u8)((d >> (8 * a)) & 0xFF);
                        }
                    }
                    if (hap_row) hap_row[jo] = (u8)(rc ? luts.comp[b] : b);
                    if (ANNOT) {
                        if (av_row) av_row[jo] = av4[i];
                        if (ap_row) ap_row[jo] = ap4[i];
                    }
                }
            }
        }
        emit_pos = limit;
        if (walk_done || emit_pos >= hi_clip) break;

        // =================== compact: drop what has been emitted ==================
        {
            int cnt = 0;
            for (int s = 0; s < nseg; ++s) cnt += (rdl(s_out, s) <= (int)emit_pos) ? 1 : 0;
            const int s0 = cnt > 0 ? cnt - 1 : 0;
            if (s0 > 0) {
                const int srcl = lane + s0 < SEG_CAP ? lane + s0 : SEG_CAP - 1;
                s_out = bperm(srcl, s_out);
                s_lo = (u32)bperm(srcl, (int)s_lo);
                s_hi = (u32)bperm(srcl, (int)s_hi);
                if (ANNOT) { s_a = bperm(srcl, s_a); s_b = bperm(srcl, s_b); }
                nseg -= s0;
            }
            M.out[lane] = s_out; M.lo[lane] = s_lo; M.hi[lane] = s_hi;
            if (ANNOT) { M.a[lane] = s_a; M.b[lane] = s_b; }
            int pcnt = 0;
            for (int s = 0; s < npatch; ++s) pcnt += (rdl(p_out, s) < (int)emit_pos) ? 1 : 0;
            if (pcnt > 0) {
                const int srcl = lane + pcnt < WAVE ? lane + pcnt : WAVE - 1;
                p_out = bperm(srcl, p_out);
                p_val = bperm(srcl, p_val);
                if (ANNOT) p_id = bperm(srcl, p_id);
                npatch -= pcnt;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// get_diffs_sparse (genotypes/mod.rs:15-125): one lane per (query, hap) row.
// ---------------------------------------------------------------------------
struct DiffArgs {
    const i64 *geno_offset_idx; i64 n_rows; int ploidy;
    const int *geno_v_idxs; const i64 *go_starts; const i64 *go_stops;
    const int *ilens; const int *v_starts; i64 n_variants;
    const u8 *keep; const i64 *keep_offsets;
    const int *q_starts; const int *q_ends; i64 q_stride;
    int *diffs;
    // fused sizing (ffi/mod.rs:794-811)
    i64 output_length; i64 *lengths;
};

__device__ __forceinline__ i64 row_diff(const DiffArgs &A, i64 k) {
    const i64 query = k / A.ploidy;
    const i64 o_idx = A.geno_offset_idx[k];
    const i64 o_s = A.go_starts[o_idx], o_e = A.go_stops[o_idx];
    const bool has_query = A.q_starts && A.q_ends && A.v_starts;   // mod.rs:35
    const bool has_keep = A.keep && A.keep_offsets;                // mod.rs:36
    i64 acc = 0;
    if (o_e - o_s <= 0) return 0;
    const i64 ks = has_keep ? A.keep_offsets[k] : 0;
    if (has_query) {                                               // mod.rs:48-85
        const i64 q_start = A.q_starts[query * A.q_stride];
        const i64 q_end = A.q_ends[query * A.q_stride];
        i64 ref_idx = q_start;
        for (i64 v = o_s; v < o_e; ++v) {
            if (has_keep && !A.keep[ks + (v - o_s)]) continue;
            const i64 vi = A.geno_v_idxs[v];
            const i64 vs = A.v_starts[vi];
            i64 il = A.ilens[vi];
            const i64 v_end = vs - imin(il, 0) + 1;
            if (v_end <= q_start) continue;
            if (vs >= q_end) break;
            if (vs >= q_start && vs < ref_idx) continue;
            ref_idx = imax(ref_idx, v_end);
            if (il < 0) il += imax(q_start - vs - 1, 0);
            il += imax(v_end - q_end, 0);
            acc += il;
        }
    } else {                                                       // mod.rs:86-103
        for (i64 v = o_s; v < o_e; ++v) {
            if (has_keep && !A.keep[ks + (v - o_s)]) continue;
            acc += A.ilens[A.geno_v_idxs[v]];
        }
    }
    return acc;
}

__global__ __launch_bounds__(256) void diffs_kernel(const DiffArgs A, const int *regions,
                                                     i64 regions_stride) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A.n_rows) return;
    const int d = (int)row_diff(A, k);      // `as i32` truncation
    if (A.diffs) A.diffs[k] = d;
    if (A.lengths) {
        i64 len;
        if (A.output_length >= 0) {
            len = A.output_length;
        } else {
            const int *reg = regions + (k / A.ploidy) * regions_stride;
            len = imax((i64)(reg[2] - reg[1]) + d, 0);
        }
        A.lengths[k + 1] = len;
        if (k == 0) A.lengths[0] = 0;
    }
}

// In-place inclusive scan of lengths[1..n] (lengths[0] = 0) by ONE workgroup;
// also reports {total, max}.  n is the batch's row count (thousands), so a single
// 1024-thread workgroup streaming the array is enough and keeps it one launch.
__global__ __launch_bounds__(1024) void offsets_scan_kernel(i64 *offs, i64 n, i64 *total_and_max) {
    __shared__ i64 wsum[16];
    __shared__ i64 wmax[16];
    __shared__ i64 carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    i64 mx = 0;
    for (i64 base = 0; base < n; base += 1024) {
        const i64 i = base + tid;
        i64 x = i < n ? offs[i + 1] : 0;
        mx = imax(mx, x);
        i64 s = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const i64 y = __shfl_up(s, o, 64);
            if (lane >= o) s += y;
        }
        if (lane == 63) wsum[wv] = s;
        __syncthreads();
        i64 pre = carry_s;
        for (int w = 0; w < wv; ++w) pre += wsum[w];
        if (i < n) offs[i + 1] = s + pre;
        __syncthreads();
        if (tid == 1023) carry_s = s + pre;
        __syncthreads();
    }
    if (total_and_max) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = imax(mx, __shfl_down(mx, o, 64));
        if (lane == 0) wmax[wv] = mx;
        __syncthreads();
        if (tid == 0) {
            i64 m = 0;
            for (int w = 0; w < 16; ++w) m = imax(m, wmax[w]);
            total_and_max[0] = carry_s;
            total_and_max[1] = m;
        }
    }
}

// ---------------------------------------------------------------------------
// Packed variant records.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_variants_kernel(const int *v_starts, const int *ilens,
                                                             const i64 *alt_offsets, const u8 *alt,
                                                             i64 n, gvl_vrec *out) {
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const i64 a0 = alt_offsets[v], a1 = alt_offsets[v + 1];
    const i64 len = a1 - a0;
    u32 inl = 0;
    for (int i = 0; i < 4 && i < len; ++i) inl |= (u32)alt[a0 + i] << (8 * i);
    gvl_vrec r;
    r.pos = v_starts[v];
    r.ilen = ilens[v];
    r.alen = (int)(len < 0 ? 0 : (len > 2147483647ll ? 2147483647ll : len));
    r.inl = inl;
    out[v] = r;
}

// ---------------------------------------------------------------------------
// In-place reverse(-complement) of masked rows: one workgroup per row.
// reverse.rs:25-69.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rc_rows_kernel(u8 *data, const i64 *offsets, const u8 *to_rc,
                                                       i64 n_rows) {
    const i64 r = blockIdx.x;
    if (r >= n_rows || !to_rc[r]) return;
    u8 *row = data + offsets[r];
    const i64 n = offsets[r + 1] - offsets[r];
    for (i64 i = threadIdx.x; i < (n + 1) / 2; i += blockDim.x) {
        const i64 j = n - 1 - i;
        const u32 a = row[i], b = row[j];
        row[i] = (u8)comp_byte(b);
        row[j] = (u8)comp_byte(a);
    }
}

__global__ __launch_bounds__(256) void reverse_rows4_kernel(u32 *data, const i64 *offsets,
                                                             const u8 *to_rc, i64 n_rows) {
    const i64 r = blockIdx.x;
    if (r >= n_rows || !to_rc[r]) return;
    u32 *row = data + offsets[r];
    const i64 n = offsets[r + 1] - offsets[r];
    for (i64 i = threadIdx.x; i < n / 2; i += blockDim.x) {
        const i64 j = n - 1 - i;
        const u32 a = row[i], b = row[j];
        row[i] = b;
        row[j] = a;
    }
}

// ---------------------------------------------------------------------------
// Stand-alone one-hot: 4 bases per lane, 16-B store per lane.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void onehot_kernel(const u8 *in, i64 n, u8 *out) {
    __shared__ Luts luts;
    init_luts(luts);
    const i64 n4 = n / 4;
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += (i64)gridDim.x * blockDim.x) {
        const u32 w = load_u32_unaligned(in + 4 * g);
        u32x4_a4 o = {luts.oh[w & 0xFF], luts.oh[(w >> 8) & 0xFF], luts.oh[(w >> 16) & 0xFF], luts.oh[w >> 24]};
        *reinterpret_cast<u32x4_a4 *>(out + 16 * g) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const i64 j = n4 * 4 + threadIdx.x;
        const u32 d = luts.oh[in[j]];
        __builtin_memcpy(out + 4 * j, &d, 4);
    }
}

// ---------------------------------------------------------------------------
// host side of the C-ABI
// ---------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *what) {
    snprintf(g_err, sizeof(g_err), fmt, what);
    return code;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return GVL_ERR_HIP;
    }
    return GVL_OK;
}

int pick_chunk(i64 max_len, int *chunks, int *chunk_len) {
    // one wave owns `chunk_len` bases of a row; rows up to 4096 bp are one chunk
    i64 cl = 2048;
    if (max_len <= 4096) cl = ((max_len + TRIP - 1) / TRIP) * TRIP;
    if (cl < TRIP) cl = TRIP;
    i64 c = (max_len + cl - 1) / cl;
    if (c < 1) c = 1;
    if (c > 0x7FFFFFFF) return 1;
    *chunks = (int)c;
    *chunk_len = (int)cl;
    return 0;
}

}  // namespace

extern "C" {

int gvl_abi_version(void) { return GVL_ABI_VERSION; }
const char *gvl_last_error(void) { return g_err; }

int gvl_pack_variants(const int32_t *v_starts, const int32_t *ilens, const int64_t *alt_offsets,
                      const uint8_t *alt_alleles, int64_t n_variants, gvl_vrec *vrec_out,
                      void *stream) {
    if (n_variants < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_pack_variants: n_variants < 0");
    if (n_variants == 0) return GVL_OK;
    if (!v_starts || !ilens || !alt_offsets || !vrec_out)
        return fail(GVL_ERR_INVALID, "%s", "gvl_pack_variants: NULL array");
    const unsigned grid = (unsigned)((n_variants + 255) / 256);
    hipLaunchKernelGGL(pack_variants_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       v_starts, ilens, (const i64 *)alt_offsets, alt_alleles, (i64)n_variants, vrec_out);
    return check_launch("gvl_pack_variants");
}

int gvl_reconstruct(const gvl_static *st, const gvl_batch *bt, const gvl_out *out, void *stream) {
    if (!st || !bt || !out) return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: NULL struct");
    if (bt->batch < 0 || bt->ploidy <= 0) return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: bad batch/ploidy");
    if (!out->haps && !out->onehot) return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: no output buffer");
    if (bt->batch == 0) return GVL_OK;
    if ((st->ref_len > 0 && !st->ref) || !st->ref_offsets || !st->geno_o_starts || !st->geno_o_stops)
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: NULL static array");
    if (st->n_geno > 0 && (!st->vrec || !st->alt_offsets || !st->geno_v_idxs))
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: NULL variant table (vrec from gvl_pack_variants is required)");
    if (!bt->regions || !bt->shifts || !bt->geno_offset_idx || bt->regions_stride < 3)
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: NULL/invalid batch array");
    if (bt->output_length < 0 && !bt->out_offsets)
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: ragged mode needs out_offsets (gvl_hap_offsets)");
    if (bt->output_length > 0x7FFFFF00ll || bt->max_row_len > 0x7FFFFF00ll)
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: row length must be < 2^31 - 256");
    if (out->onehot && out->onehot_layout == GVL_ONEHOT_CL && (bt->output_length < 0 || bt->out_offsets))
        return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_reconstruct: channel-major one-hot needs fixed-length rows");
    if (out->onehot_layout != GVL_ONEHOT_LC && out->onehot_layout != GVL_ONEHOT_CL)
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: bad onehot_layout");

    ReconArgs A;
    memset(&A, 0, sizeof(A));
    A.ref = st->ref; A.ref_len = st->ref_len; A.ref_offsets = (const i64 *)st->ref_offsets;
    A.vrec = st->vrec; A.alt_offsets = (const i64 *)st->alt_offsets; A.alt_alleles = st->alt_alleles;
    A.alt_len = st->alt_len; A.n_variants = st->n_variants;
    A.go_starts = (const i64 *)st->geno_o_starts; A.go_stops = (const i64 *)st->geno_o_stops;
    A.geno_v_idxs = st->geno_v_idxs;
    A.regions = bt->regions; A.regions_stride = bt->regions_stride; A.shifts = bt->shifts;
    A.geno_offset_idx = (const i64 *)bt->geno_offset_idx;
    A.keep = bt->keep; A.keep_offsets = (const i64 *)bt->keep_offsets; A.to_rc = bt->to_rc;
    A.out_offsets = (const i64 *)bt->out_offsets;
    A.fixed_len = bt->out_offsets ? -1 : bt->output_length;
    A.n_queries = bt->batch; A.ploidy = (int)bt->ploidy;
    // longest row: fixed mode -> output_length; caller-supplied offsets -> the
    // caller's max_row_len hint (must bound every row, or longer rows are left
    // partly unwritten)
    i64 ml = bt->out_offsets ? bt->max_row_len : bt->output_length;
    if (bt->out_offsets && bt->output_length > ml) ml = bt->output_length;
    if (ml < 0) ml = 0;
    if (pick_chunk(ml, &A.chunks, &A.chunk_len)) return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: too many chunks");
    A.ref_only = 0;
    A.pad = st->pad_char;
    A.haps = out->haps; A.onehot = out->onehot; A.onehot_cl = out->onehot_layout == GVL_ONEHOT_CL;
    A.av = out->annot_v_idxs; A.ap = out->annot_ref_pos; A.out_offsets_w = (i64 *)out->out_offsets;

    const i64 units = bt->batch * bt->ploidy * (i64)A.chunks;
    const i64 grid = (units + WG_WAVES - 1) / WG_WAVES;
    if (grid <= 0) return GVL_OK;
    if (grid > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: batch too large for one launch");
    const bool annot = out->annot_v_idxs || out->annot_ref_pos;
    if (annot)
        hipLaunchKernelGGL(reconstruct_kernel<true>, dim3((unsigned)grid), dim3(WG_THREADS), 0, (hipStream_t)stream, A);
    else
        hipLaunchKernelGGL(reconstruct_kernel<false>, dim3((unsigned)grid), dim3(WG_THREADS), 0, (hipStream_t)stream, A);
    return check_launch("gvl_reconstruct");
}

int gvl_get_reference(const gvl_static *st, const int32_t *regions, int64_t regions_stride,
                      int64_t n_rows, const int64_t *out_offsets, int64_t max_row_len,
                      const uint8_t *to_rc, uint8_t *out, uint8_t *onehot, void *stream) {
    if (!st || n_rows < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference: bad arguments");
    if (n_rows == 0) return GVL_OK;
    if ((st->ref_len > 0 && !st->ref) || !st->ref_offsets || !regions || !out_offsets || regions_stride < 3 || (!out && !onehot))
        return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference: NULL/invalid array");
    if (max_row_len < 0 || max_row_len > 0x7FFFFF00ll)
        return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference: bad max_row_len");
    ReconArgs A;
    memset(&A, 0, sizeof(A));
    A.ref = st->ref; A.ref_len = st->ref_len; A.ref_offsets = (const i64 *)st->ref_offsets;
    A.regions = regions; A.regions_stride = regions_stride;
    A.to_rc = to_rc; A.out_offsets = (const i64 *)out_offsets; A.fixed_len = -1;
    A.n_queries = n_rows; A.ploidy = 1;
    if (pick_chunk(max_row_len, &A.chunks, &A.chunk_len)) return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference: too many chunks");
    A.ref_only = 1;
    A.pad = st->pad_char;
    A.haps = out; A.onehot = onehot;
    const i64 units = n_rows * (i64)A.chunks;
    const i64 grid = (units + WG_WAVES - 1) / WG_WAVES;
    if (grid > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference: batch too large");
    hipLaunchKernelGGL(reconstruct_kernel<false>, dim3((unsigned)grid), dim3(WG_THREADS), 0, (hipStream_t)stream, A);
    return check_launch("gvl_get_reference");
}

static int fill_diff_args(DiffArgs &D, const gvl_static *st, const gvl_batch *bt, const char *who) {
    if (!st || !bt) return fail(GVL_ERR_INVALID, "%s: NULL struct", who);
    if (bt->batch < 0 || bt->ploidy <= 0) return fail(GVL_ERR_INVALID, "%s: bad batch/ploidy", who);
    memset(&D, 0, sizeof(D));
    D.geno_offset_idx = (const i64 *)bt->geno_offset_idx; D.n_rows = bt->batch * bt->ploidy;
    D.ploidy = (int)bt->ploidy;
    D.geno_v_idxs = st->geno_v_idxs; D.go_starts = (const i64 *)st->geno_o_starts;
    D.go_stops = (const i64 *)st->geno_o_stops; D.ilens = st->ilens; D.v_starts = st->v_starts;
    D.n_variants = st->n_variants; D.keep = bt->keep; D.keep_offsets = (const i64 *)bt->keep_offsets;
    if (bt->batch > 0 && (!D.geno_offset_idx || !D.go_starts || !D.go_stops))
        return fail(GVL_ERR_INVALID, "%s: NULL array", who);
    return GVL_OK;
}

int gvl_get_diffs_sparse(const gvl_static *st, const gvl_batch *bt, const int32_t *q_starts,
                         const int32_t *q_ends, int64_t q_stride, int32_t *diffs, void *stream) {
    DiffArgs D;
    int rc = fill_diff_args(D, st, bt, "gvl_get_diffs_sparse");
    if (rc) return rc;
    if (D.n_rows == 0) return GVL_OK;
    if (!diffs) return fail(GVL_ERR_INVALID, "%s", "gvl_get_diffs_sparse: NULL diffs");
    D.q_starts = q_starts; D.q_ends = q_ends; D.q_stride = q_stride > 0 ? q_stride : 1;
    D.diffs = diffs; D.output_length = 0; D.lengths = nullptr;
    const unsigned grid = (unsigned)((D.n_rows + 255) / 256);
    hipLaunchKernelGGL(diffs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, D, (const int *)nullptr, (i64)0);
    return check_launch("gvl_get_diffs_sparse");
}

int gvl_hap_offsets(const gvl_static *st, const gvl_batch *bt, int32_t *diffs, int64_t *out_offsets,
                    int64_t *total_and_max, void *stream) {
    DiffArgs D;
    int rc = fill_diff_args(D, st, bt, "gvl_hap_offsets");
    if (rc) return rc;
    if (!out_offsets) return fail(GVL_ERR_INVALID, "%s", "gvl_hap_offsets: NULL out_offsets");
    if (D.n_rows == 0) {
        hipError_t e = hipMemsetAsync(out_offsets, 0, sizeof(int64_t), (hipStream_t)stream);
        if (e == hipSuccess && total_and_max)
            e = hipMemsetAsync(total_and_max, 0, 2 * sizeof(int64_t), (hipStream_t)stream);
        if (e != hipSuccess) return fail(GVL_ERR_HIP, "gvl_hap_offsets: %s", hipGetErrorString(e));
        return GVL_OK;
    }
    if (!bt->regions || bt->regions_stride < 3) return fail(GVL_ERR_INVALID, "%s", "gvl_hap_offsets: NULL regions");
    D.q_starts = bt->regions + 1; D.q_ends = bt->regions + 2; D.q_stride = bt->regions_stride;
    D.diffs = diffs; D.output_length = bt->output_length; D.lengths = (i64 *)out_offsets;
    const unsigned grid = (unsigned)((D.n_rows + 255) / 256);
    hipLaunchKernelGGL(diffs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, D, bt->regions, (i64)bt->regions_stride);
    rc = check_launch("gvl_hap_offsets(diffs)");
    if (rc) return rc;
    hipLaunchKernelGGL(offsets_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (i64 *)out_offsets, D.n_rows, (i64 *)total_and_max);
    return check_launch("gvl_hap_offsets(scan)");
}

int gvl_rc_rows(uint8_t *data, const int64_t *offsets, const uint8_t *to_rc, int64_t n_rows, void *stream) {
    if (n_rows < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_rc_rows: n_rows < 0");
    if (n_rows == 0) return GVL_OK;
    if (!data || !offsets || !to_rc) return fail(GVL_ERR_INVALID, "%s", "gvl_rc_rows: NULL array");
    if (n_rows > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_rc_rows: too many rows");
    hipLaunchKernelGGL(rc_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, data, (const i64 *)offsets, to_rc, (i64)n_rows);
    return check_launch("gvl_rc_rows");
}

int gvl_reverse_rows_4(void *data, const int64_t *offsets, const uint8_t *to_rc, int64_t n_rows, void *stream) {
    if (n_rows < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_reverse_rows_4: n_rows < 0");
    if (n_rows == 0) return GVL_OK;
    if (!data || !offsets || !to_rc) return fail(GVL_ERR_INVALID, "%s", "gvl_reverse_rows_4: NULL array");
    if (n_rows > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_reverse_rows_4: too many rows");
    hipLaunchKernelGGL(reverse_rows4_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, (u32 *)data, (const i64 *)offsets, to_rc, (i64)n_rows);
    return check_launch("gvl_reverse_rows_4");
}

int gvl_onehot(const uint8_t *in, int64_t n, uint8_t *out, void *stream) {
    if (n < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_onehot: n < 0");
    if (n == 0) return GVL_OK;
    if (!in || !out) return fail(GVL_ERR_INVALID, "%s", "gvl_onehot: NULL array");
    i64 groups = (n / 4 + 255) / 256;
    if (groups < 1) groups = 1;
    if (groups > 8192) groups = 8192;
    hipLaunchKernelGGL(onehot_kernel, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream, in, (i64)n, out);
    return check_launch("gvl_onehot");
}

}  // extern "C"
