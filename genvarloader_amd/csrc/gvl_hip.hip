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
// sequentially and memcpy's reference/allele runs.  Here a workgroup of 8 waves owns
// 8 (row, chunk)s of output, one per wave:
//   1. PLAN: the row's walk is restated as a SEGMENT table (out_start, kind, source
//      delta; <= 64 entries in LDS) plus a PATCH list (pure SNPs do not split a
//      reference run).  Three planners produce it, picked per row:
//        fast    every kept variant is a SNP: no scan, planned by the row's own wave;
//        packed  the first "slow" wave plans all slow rows of the workgroup at once,
//                lane = row x variant, DPP scans in groups of 8 lanes (<= 8 variants);
//        scans   wave-wide DPP scans, 64 variants per trip (any number of variants);
//      and a scalar replay of the reference's loop (recon_wave_scalar) takes what the
//      i32 scans cannot (coordinates >= 2^30, > 64 table entries per chunk).
//   2. STREAM: all 64 lanes stream the output: 4 bases per lane per trip, one
//      unaligned dword load of reference bytes (256 B per wave-load), SNP patches
//      applied in registers, reverse-complement folded into the store index + LUT,
//      one-hot through a 256-entry LDS LUT, one nontemporal 16-B store per lane
//      (1 KiB contiguous per wave-store).
// There is no second pass over HBM for RC or one-hot and no intermediate
// haplotype buffer unless the caller asks for the bytes too.  Integer
// gather/scatter: HBM-bound, no MFMA.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <new>
#include <type_traits>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "gvl_hip.h"

namespace {

typedef long long i64;
typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned char u8;

constexpr int WAVE = 64;
constexpr int WG_WAVES = 8;               // one wave per row, 8 rows per workgroup
constexpr int WG_THREADS = WAVE * WG_WAVES;
constexpr int GROUP = 4;                   // bases per lane per trip
constexpr int TRIP = WAVE * GROUP;         // 256 bases per wave trip
constexpr int SEG_CAP = 64;                // lane-resident segment table
constexpr int SEG_FLUSH = 59;              // flush before a step could overflow
constexpr int PATCH_FLUSH = 62;
constexpr int CHUNK_TRIPS = 8;            // trips per chunk on the planned path (chunk_len <= 2048)

enum : u32 { K_REF = 0, K_ALLELE = 1, K_PAD_LEAD = 2, K_PAD_TRAIL = 3 };
constexpr i64 DELTA_BIAS = 1ll << 40;      // src - out_start + BIAS fits 42 bits

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(4))) u32x4_a4 { u32 x, y, z, w; };
// Output is written once and never re-read by the kernel: nontemporal stores keep it from
// displacing the reference / variant lines in L2 and from piling up as dirty lines that the
// end-of-kernel release has to flush (cfg3: 14.8 -> 12.7 us per launch, 8.1 -> 6.9 us pipelined).
typedef u32 v4u_t __attribute__((ext_vector_type(4)));
typedef v4u_t __attribute__((aligned(4))) v4u_a4;
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef v4i_t __attribute__((aligned(4))) v4i_a4;
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef v4f_t __attribute__((aligned(4))) v4f_a4;
typedef u32 __attribute__((aligned(1))) u32_a1;
__device__ __forceinline__ void store_oh16(u8 *dst, const u32x4_a4 &o) {
    v4u_t v = {o.x, o.y, o.z, o.w};
    __builtin_nontemporal_store(v, reinterpret_cast<v4u_a4 *>(dst));
}
__device__ __forceinline__ void store_i32x4(int *dst, int a, int b, int c, int d) {
    v4i_t v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<v4i_a4 *>(dst));
}
__device__ __forceinline__ void store_f32x4(float *dst, float a, float b, float c, float d) {
    v4f_t v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<v4f_a4 *>(dst));
}
// (write-back, not nontemporal: the painter's scratch track is read back by the realignment right away)
__device__ __forceinline__ void store_f32x4_wb(float *dst, float a, float b, float c, float d) {
    v4f_t v = {a, b, c, d};
    *reinterpret_cast<v4f_a4 *>(dst) = v;
}
__device__ __forceinline__ void store_u32_unaligned(u8 *dst, u32 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<u32_a1 *>(dst));
}
struct __attribute__((aligned(4))) i32x4_a4 { int x, y, z, w; };

// workgroup barrier that orders LDS traffic only: global loads a wave has in flight stay in flight
// (__syncthreads' fence waits for vmcnt(0) on gfx9)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
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
    u32 v;          // (nontemporal loads were measured too: 10-25 % slower, the lines are shared)
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

// LDS tables shared by the workgroup: one-hot dword of a byte, one-hot of its
// complement (= the byte-reversed dword: A<->T, C<->G swaps channels 0<->3, 1<->2),
// and the complement byte.
struct Luts { u32 oh[256]; u32 oh_rc[256]; u32 comp[256]; };

__device__ __forceinline__ void init_luts(Luts &l) {   // 256-thread workgroups
    const int b = threadIdx.x & 255;
    const u32 d = onehot_dword((u32)b);
    l.oh[b] = d;
    l.oh_rc[b] = __builtin_bswap32(d);
    l.comp[b] = comp_byte((u32)b);
    __syncthreads();
}

struct ReconArgs {
    // static
    const u8 *ref; i64 ref_len; const i64 *ref_offsets;
    const gvl_vrec *vrec; const i64 *alt_offsets; const u8 *alt_alleles; i64 alt_len;
    i64 n_variants;
    const i64 *go_starts; const i64 *go_stops; const int *geno_v_idxs; const gvl_grec *grec;
    const gvl_srec *srec;   // slot-major records (nullable): 8 per genotype slot, no CSR hop
    int n_contigs; i64 n_geno_offsets;
    // batch
    const int *regions; i64 regions_stride; const int *shifts; const i64 *geno_offset_idx;
    const u8 *keep; const i64 *keep_offsets; const u8 *to_rc; const i64 *out_offsets;
    i64 fixed_len;      // >= 0 or -1
    i64 n_rows; int ploidy; int ploidy_shift; int chunk_len;
    int dbg;
    int ref_only;       // get_reference mode: no variants, shift 0, row len from out_offsets
    u32 pad;
    // out
    u8 *haps; u8 *onehot; int *av; int *ap; i64 *out_offsets_w;
    u64 *stamps;        // diagnostic builds (-DGVL_DIAG): per-workgroup phase time stamps
    int *async_err;     // host-mapped word: set when the launch finds a row longer than the max_row_len hint
    const u8 *ref4;     // nibble-packed reference (gvl_static.ref4): read by recon_lean_kernel only
};

// Per-wave mirror of the segment table + staging, used only by "general" trips.
struct SegMirror { int out[SEG_CAP]; u32 lo[SEG_CAP]; u32 hi[SEG_CAP]; int a[SEG_CAP]; int b[SEG_CAP]; };
template <bool ANNOT>
struct Stage {
    u32 w[WAVE];
    int av[GROUP][ANNOT ? WAVE : 1];
    int ap[GROUP][ANNOT ? WAVE : 1];
};

__device__ __forceinline__ i64 seg_delta(u32 lo, u32 hi) {
    return (i64)((((u64)(hi & 0x3FFFFFFFu)) << 32) | lo) - DELTA_BIAS;
}

enum { OH_NONE = 0, OH_LC = 1, OH_CL = 2 };

// ---------------------------------------------------------------------------------
// Scalar path: ONE wave replays the reference's walk on the scalar unit and streams its
// (row, chunk).  Handles everything (any number of variants / segments, via flush +
// compaction of the 64-entry lane tables).  It is the fallback of the planned path below
// for rows the cooperative planner does not take (more than CAP_V variants, table
// overflow) -- it costs ~10x more issue slots per row, so it is not the default.
// ---------------------------------------------------------------------------------
template <int OH, bool HAPS, bool ANNOT>
__device__ __forceinline__ void recon_wave_scalar(const ReconArgs &A, const Luts &luts, SegMirror &M,
                                                   Stage<ANNOT> &G, const i64 k, const int chunk,
                                                   const int lane) {
    const i64 query = A.ploidy_shift >= 0 ? (k >> A.ploidy_shift) : (i64)((u32)k / (u32)A.ploidy);

    // ---- row parameters (level-1 loads; wave-uniform) -------------------------
    const int *reg = A.regions + query * A.regions_stride;
    const i64 c_idx = rfl(reg[0]);
    const i64 ref_start = rfl(reg[1]);
    i64 shift = 0, o_idx = 0;
    bool ref_zero_fill = false;
    if (!A.ref_only) {
        shift = rfl(A.shifts[k]);
        o_idx = rfl64(A.geno_offset_idx[k]);
    } else {
        ref_zero_fill = ref_start >= (i64)rfl(reg[2]);   // reference/mod.rs:16-18
    }
    const bool rc = A.to_rc ? (rfl((int)A.to_rc[k]) != 0) : false;
    i64 row_base; int L;
    if (A.out_offsets) {
        row_base = rfl64(A.out_offsets[k]);
        L = (int)(rfl64(A.out_offsets[k + 1]) - row_base);
    } else {
        row_base = k * A.fixed_len;
        L = (int)A.fixed_len;
    }
    if (A.out_offsets_w && chunk == 0 && lane == 0) {
        A.out_offsets_w[k] = row_base;
        if (k == A.n_rows - 1) A.out_offsets_w[k + 1] = row_base + L;
    }
    const int lo_clip = chunk * A.chunk_len;
    if (lo_clip >= L) return;
    const int hi_clip = (L - lo_clip > A.chunk_len) ? lo_clip + A.chunk_len : L;

    // ---- level-2 loads ---------------------------------------------------------
    const i64 c_s = rfl64(A.ref_offsets[c_idx]);
    const i64 R = rfl64(A.ref_offsets[c_idx + 1]) - c_s;
    const bool has_keep = A.keep && A.keep_offsets;
    i64 o_s = 0, keep_off = 0;
    int n_var = 0;
    if (!A.ref_only) {
        o_s = rfl64(A.go_starts[o_idx]);
        const i64 nv = rfl64(A.go_stops[o_idx]) - o_s;
        n_var = nv < 0 ? 0 : (nv > 0x7FFFFFFFll ? 0x7FFFFFFF : (int)nv);
        if (A.dbg & 1) n_var = 0;
        if (has_keep) keep_off = rfl64(A.keep_offsets[k]);
    }

    // ---- segment / patch tables (lane s holds entry s) ------------------------
    int s_out = 0; u32 s_lo = 0, s_hi = 0; int s_a = 0, s_b = 0;
    int p_out = 0, p_val = 0, p_id = 0;
    int nseg = 0, npatch = 0;
    u32 last_kind = 0xFFu; i64 last_delta = 0;

    // Append a segment [o_start, o_start+len) of `kind` whose byte at output position p
    // is source[delta + p].  Output coordinates are ints (< 2^31); a reference run that
    // continues the previous one (same delta) is merged, which is what makes SNPs free.
    auto push = [&](u32 kind, int o_start, int len, i64 src, int id, int vpos) {
        if (len <= 0 || o_start + len <= lo_clip || o_start >= hi_clip) return;
        const i64 delta = src - o_start;
        if (kind == K_REF && last_kind == K_REF && delta == last_delta) return;
        const u64 enc = (u64)(delta + DELTA_BIAS) | ((u64)kind << 62);
        if (lane == nseg) {
            s_out = o_start; s_lo = (u32)enc; s_hi = (u32)(enc >> 32);
            if (ANNOT) { s_a = id; s_b = vpos; }
        }
        last_kind = kind; last_delta = delta; ++nseg;
    };

    // ---- walk state: reconstruct/mod.rs:61-83 -----------------------------------
    i64 ref_idx = ref_start, shifted = 0;
    int out_idx = 0;
    if (ref_idx < 0) {
        const i64 raw = -ref_idx;
        shifted = imin(shift, raw);
        // a pad longer than the row is clamped: the row is then all pad either way
        const int n = (int)imin(raw - shifted, (i64)L);
        push(K_PAD_LEAD, 0, n, 0, -1, -1);
        out_idx = n;
        ref_idx = 0;
    }

    // variant record registers for the current trip of 64 variants
    int r_pos = 0, r_ilen = 0, r_alen = 0, r_inl = 0, r_vi = 0, r_a0lo = 0, r_a0hi = 0, r_keep = 1;
    int vi = 0;          // next variant of the row
    int vb = -WAVE;      // base of the loaded trip
    bool walk_done = false;
    int emit_pos = lo_clip;

    const u32 padb = A.pad & 0xFFu;
    // RC is folded into the store: forward position p lands at L-4-p (group) with the
    // group's 4 bytes reversed (one v_perm with a uniform selector) and complemented
    // (second LUT).  `lane_off` is the lane's share of the store offset, in bases.
    const u32 rc_sel = rc ? 0x00010203u : 0x03020100u;
    const u32 *oh_t = rc ? luts.oh_rc : luts.oh;
    const int lane_pos = rc ? -GROUP * lane : GROUP * lane;
    u8 *hap_row = HAPS ? A.haps + row_base : nullptr;
    u8 *oh_row = OH != OH_NONE ? A.onehot + 4 * row_base : nullptr;
    int *av_row = (ANNOT && A.av) ? A.av + row_base : nullptr;
    int *ap_row = (ANNOT && A.ap) ? A.ap + row_base : nullptr;

    for (;;) {
        // =================== fill: replay the reference walk =====================
        while (!walk_done && nseg <= SEG_FLUSH && npatch <= PATCH_FLUSH) {
            bool stop = (vi >= n_var) || (out_idx >= hi_clip);
            if (!stop) {
                if (vi - vb >= WAVE) {
                    // gather the next 64 variant records (levels 3 and 4)
                    vb = vi;
                    const int j = vb + lane;
                    const bool valid = j < n_var;
                    int v = valid ? A.geno_v_idxs[o_s + j] : 0;
                    v = v < 0 ? 0 : ((i64)v >= A.n_variants ? (int)(A.n_variants - 1) : v);
                    r_vi = v;
                    if (valid) {
                        const i32x4 rec = *reinterpret_cast<const i32x4 *>(A.vrec + v);
                        const i64 a0 = A.alt_offsets[v];
                        r_pos = rec.x; r_ilen = rec.y; r_alen = rec.z; r_inl = rec.w;
                        r_a0lo = (int)(u32)(u64)a0; r_a0hi = (int)(u32)((u64)a0 >> 32);
                        r_keep = has_keep ? (int)A.keep[keep_off + j] : 1;
                    }
                }
                const int i = vi - vb;
                ++vi;
                // --- one step of reconstruct/mod.rs:85-198 ---
                if (has_keep && rdl(r_keep, i) == 0) continue;           // :86-90
                const i64 pos = rdl(r_pos, i);
                const int d = rdl(r_ilen, i);
                const int alen = rdl(r_alen, i);
                const i64 v_end = pos - (d < 0 ? (i64)d : 0) + 1;         // :96
                if (pos < ref_idx) {
                    // :99-102 DEL spanning the window start (only possible while pos < ref_start
                    // <= ref_idx), else :108-110 first ALT wins
                    if (pos < ref_start && d < 0 && v_end >= ref_start) ref_idx = v_end;
                    continue;
                }
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
                const i64 n64 = pos - ref_idx;
                if (n64 >= (i64)(L - out_idx)) {                          // :154-158 (">=")
                    stop = true;
                } else {
                    const int n = (int)n64;
                    const int id = rdl(r_vi, i);
                    if (d == 0 && alen == 1) {
                        // pure SNP (skip == 0 here): the reference run continues through the
                        // variant's own base and one output byte is patched
                        push(K_REF, out_idx, n + 1, c_s + ref_idx, -1, -1);
                        out_idx += n;
                        if (out_idx >= lo_clip && out_idx < hi_clip) {
                            if (lane == npatch) { p_out = out_idx; p_val = rdl(r_inl, i) & 0xFF; if (ANNOT) p_id = id; }
                            ++npatch;
                        }
                        out_idx += 1;
                    } else {
                        push(K_REF, out_idx, n, c_s + ref_idx, -1, -1);
                        out_idx += n;
                        const i64 al = (i64)alen - skip;
                        const int w = (int)imin(al, (i64)(L - out_idx));  // :178
                        push(K_ALLELE, out_idx, w, rdl64(r_a0lo, r_a0hi, i) + skip, id, (int)pos);
                        out_idx += w;
                    }
                    ref_idx = v_end;                                      // :193
                    if (out_idx >= L) stop = true;                        // :195-197
                }
            }
            if (stop) {
                // residual shift + tail: reconstruct/mod.rs:200-255
                if (shifted < shift) ref_idx = imin(ref_idx + (shift - shifted), R);
                const int u = L - out_idx;
                if (u > 0) {
                    const i64 avail = R - ref_idx;
                    const int w = (int)imin((i64)u, avail);
                    int end = out_idx;
                    if (w > 0) { push(K_REF, out_idx, w, c_s + ref_idx, -1, -1); end += w; }
                    if (end < L) push(K_PAD_TRAIL, end, L - end, 0, -1, -1);
                }
                out_idx = out_idx > L ? out_idx : L;
                walk_done = true;
            }
        }

        // =================== emit [emit_pos, limit) ===============================
        // One trip = 256 bases, 4 per lane.  A trip that lies inside ONE reference run
        // (the common case) is "uniform": every lane loads its 4 bytes from a scalar
        // base.  Other trips take the general path: per-lane segment lookup through the
        // LDS mirror; lanes on a boundary / in an allele / pad / row end assemble their
        // 4 bytes one by one.  Then: SNP patches, reverse-complement, one-hot LUT, store.
        const int cov = out_idx < lo_clip ? lo_clip : (out_idx > hi_clip ? hi_clip : out_idx);
        const int limit = walk_done ? hi_clip : (cov & ~3);
        int sc = -1, pc = 0;
        int c_next = emit_pos;      // start of segment sc+1 (or cov)
        bool c_ok = false;          // segment sc is a reference run that stays inside `ref`
        const u8 *c_base = A.ref;   // A.ref + delta of segment sc
        int c_apb = 0;              // annotation: contig coordinate of output position 0
        int np_pos = npatch > 0 ? rdl(p_out, 0) : 0x7FFFFFFF;
        bool mirror_valid = false;
        for (int p0 = emit_pos; p0 < limit; p0 += TRIP) {
            const int t_end = (limit - p0 > TRIP) ? p0 + TRIP : limit;
            while (c_next <= p0) {      // advance to the segment that holds p0
                ++sc;
                const u32 hi = (u32)rdl((int)s_hi, sc);
                const i64 dl = seg_delta((u32)rdl((int)s_lo, sc), hi);
                const int c_start = rdl(s_out, sc);
                c_next = sc + 1 < nseg ? rdl(s_out, sc + 1) : cov;
                const int e = c_next < limit ? c_next : limit;
                c_ok = (hi >> 30) == K_REF && dl + (c_start > emit_pos ? c_start : emit_pos) >= 0 &&
                       dl + e <= A.ref_len && !ref_zero_fill;
                c_base = A.ref + dl;
                if (ANNOT) c_apb = (int)(dl - c_s);
            }
            const int p = p0 + GROUP * lane;
            const bool act = p < limit;
            const bool full = p + GROUP <= limit;
            u32 wv = 0;
            int av4[GROUP], ap4[GROUP];
            if (c_ok && c_next >= t_end && ((t_end - p0) & 3) == 0) {
                // ---- uniform trip -------------------------------------------------
                if (act && !(A.dbg & 4)) wv = load_u32_unaligned(c_base + p0 + (u32)(GROUP * lane));
                if (ANNOT) {
#pragma unroll
                    for (int i = 0; i < GROUP; ++i) { av4[i] = -1; ap4[i] = c_apb + p + i; }
                }
            } else {
                // ---- general trip -------------------------------------------------
                if (!mirror_valid) {
                    M.out[lane] = s_out; M.lo[lane] = s_lo; M.hi[lane] = s_hi;
                    if (ANNOT) { M.a[lane] = s_a; M.b[lane] = s_b; }
                    mirror_valid = true;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                if (act) {
                    // largest li with M.out[li] <= p (binary search; nseg <= 64)
                    int li = 0;
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1) {
                        const int t = li + step;
                        if (t < nseg && M.out[t] <= p) li = t;
                    }
                    const u32 l0 = M.lo[li], h0 = M.hi[li];
                    const int nx = li + 1 < nseg ? M.out[li + 1] : cov;
                    const i64 src = seg_delta(l0, h0) + p;
                    if ((h0 >> 30) == K_REF && p + GROUP <= nx && full && src >= 0 &&
                        src + GROUP <= A.ref_len && !ref_zero_fill) {
                        wv = load_u32_unaligned(A.ref + src);
                        if (ANNOT) {
#pragma unroll
                            for (int i = 0; i < GROUP; ++i) { av4[i] = -1; ap4[i] = (int)(src - c_s) + i; }
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < GROUP; ++i) {
                            const int pp = p + i;
                            u32 b = 0; int a_v = -1, a_p = -1;
                            if (pp < limit) {
                                while (li + 1 < nseg && M.out[li + 1] <= pp) ++li;
                                const u32 l2 = M.lo[li], h2 = M.hi[li];
                                const u32 k2 = h2 >> 30;
                                const i64 s2 = seg_delta(l2, h2) + pp;
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
                            wv |= b << (8 * i);
                            if (ANNOT) { av4[i] = a_v; ap4[i] = a_p; }
                        }
                    }
                }
            }
            // ---- SNP patches that land in this trip (sorted; scalar cursor) ----------
            while (np_pos < t_end) {
                const u32 dd = (u32)(np_pos - p);
                if (dd < (u32)GROUP) {
                    const u32 sh = dd * 8;
                    wv = (wv & ~(0xFFu << sh)) | ((u32)rdl(p_val, pc) << sh);
                }
                if (ANNOT) {
                    const int pid = rdl(p_id, pc);
#pragma unroll
                    for (int i = 0; i < GROUP; ++i) if (dd == (u32)i) av4[i] = pid;
                }
                ++pc;
                np_pos = pc < npatch ? rdl(p_out, pc) : 0x7FFFFFFF;
            }
            // ---- stores ------------------------------------------------------------------
            if (full && !(A.dbg & 2)) {
                // forward: jo = p; RC: jo = L - 4 - p.  In both cases jo = jo0 + lane_pos.
                const int jo = (rc ? L - GROUP - p0 : p0) + lane_pos;
                const u32 ww = __builtin_amdgcn_perm(0u, wv, rc_sel);
                const u32 b0_ = ww & 0xFF, b1_ = (ww >> 8) & 0xFF, b2_ = (ww >> 16) & 0xFF, b3_ = ww >> 24;
                if (OH == OH_LC) {
                    u32x4_a4 o = {oh_t[b0_], oh_t[b1_], oh_t[b2_], oh_t[b3_]};
                    store_oh16(oh_row + 4 * (i64)jo, o);
                } else if (OH == OH_CL) {
                    // channel-major (rows, 4, L): plane a holds byte a of each one-hot dword
                    const u32 d0 = oh_t[b0_], d1 = oh_t[b1_], d2 = oh_t[b2_], d3 = oh_t[b3_];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const u32 sh = 8 * a;
                        const u32 v = ((d0 >> sh) & 0xFF) | (((d1 >> sh) & 0xFF) << 8) |
                                      (((d2 >> sh) & 0xFF) << 16) | (((d3 >> sh) & 0xFF) << 24);
                        store_u32_unaligned(oh_row + (i64)a * L + jo, v);
                    }
                }
                if (HAPS) {
                    u32 hv = ww;
                    if (rc) hv = luts.comp[b0_] | (luts.comp[b1_] << 8) | (luts.comp[b2_] << 16) | (luts.comp[b3_] << 24);
                    store_u32_unaligned(hap_row + jo, hv);
                }
                if (ANNOT) {
                    if (av_row) {
                        i32x4_a4 o = rc ? i32x4_a4{av4[3], av4[2], av4[1], av4[0]} : i32x4_a4{av4[0], av4[1], av4[2], av4[3]};
                        store_i32x4(av_row + jo, o.x, o.y, o.z, o.w);
                    }
                    if (ap_row) {
                        i32x4_a4 o = rc ? i32x4_a4{ap4[3], ap4[2], ap4[1], ap4[0]} : i32x4_a4{ap4[0], ap4[1], ap4[2], ap4[3]};
                        store_i32x4(ap_row + jo, o.x, o.y, o.z, o.w);
                    }
                }
            }
            if (limit & 3) {
                // the partial group at the row end (L % 4 != 0): the owning lane parks its
                // patched bytes; lanes 0..(L&3)-1 then store one base each
                const int p_last = limit & ~3;
                if (p_last >= p0 && p_last < p0 + TRIP) {
                    if (act && !full) {
                        G.w[0] = wv;
                        if (ANNOT) {
#pragma unroll
                            for (int i = 0; i < GROUP; ++i) { G.av[i][0] = av4[i]; G.ap[i][0] = ap4[i]; }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (lane < (limit & 3)) {
                        const int pp = p_last + lane;
                        const u32 b = (G.w[0] >> (8 * lane)) & 0xFF;
                        const i64 jo = rc ? (i64)(L - 1 - pp) : (i64)pp;
                        if (OH != OH_NONE) {
                            const u32 d = oh_t[b];
                            if (OH == OH_LC) {
                                __builtin_memcpy(oh_row + 4 * jo, &d, 4);
                            } else {
#pragma unroll
                                for (int a = 0; a < 4; ++a) oh_row[(i64)a * L + jo] = (u8)((d >> (8 * a)) & 0xFF);
                            }
                        }
                        if (HAPS) hap_row[jo] = (u8)(rc ? luts.comp[b] : b);
                        if (ANNOT) {
                            if (av_row) av_row[jo] = G.av[lane][0];
                            if (ap_row) ap_row[jo] = G.ap[lane][0];
                        }
                    }
                }
            }
        }
        emit_pos = limit;
        if (walk_done || emit_pos >= hi_clip) break;

        // =================== compact: drop what has been emitted ==================
        {
            int cnt = 0;
            for (int s = 0; s < nseg; ++s) cnt += (rdl(s_out, s) <= emit_pos) ? 1 : 0;
            const int s0 = cnt > 0 ? cnt - 1 : 0;
            if (s0 > 0) {
                const int srcl = lane + s0 < SEG_CAP ? lane + s0 : SEG_CAP - 1;
                s_out = bperm(srcl, s_out);
                s_lo = (u32)bperm(srcl, (int)s_lo);
                s_hi = (u32)bperm(srcl, (int)s_hi);
                if (ANNOT) { s_a = bperm(srcl, s_a); s_b = bperm(srcl, s_b); }
                nseg -= s0;
            }
            int pcnt = 0;
            for (int s = 0; s < npatch; ++s) pcnt += (rdl(p_out, s) < emit_pos) ? 1 : 0;
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

// ---------------------------------------------------------------------------------
// wave64 scans on the DPP network (row_shr 1/2/4/8, row_bcast15, row_bcast31; the
// gfx9-family sequence LLVM's atomic optimizer emits) -- no LDS round trips.
// ---------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int identity, int v) {
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}
struct OpMaxU { static constexpr int identity = 0;  __device__ static int f(int a, int b) { return (int)(((u32)a > (u32)b) ? (u32)a : (u32)b); } };
struct OpMaxI { static constexpr int identity = -1; __device__ static int f(int a, int b) { return a > b ? a : b; } };
struct OpAdd  { static constexpr int identity = 0;  __device__ static int f(int a, int b) { return a + b; } };
// saturating add of values in [0, 2^31): a + b never wraps in u32, clamp to 2^31 - 1
struct OpSat  { static constexpr int identity = 0;  __device__ static int f(int a, int b) { u32 t = (u32)a + (u32)b; return (int)(t > 0x7FFFFFFFu ? 0x7FFFFFFFu : t); } };

template <typename Op>
__device__ __forceinline__ int wave_scan_inclusive(int v) {
    constexpr int id = Op::identity;
    v = Op::f(v, dpp_mov<0x111, 0xf>(id, v));   // row_shr:1
    v = Op::f(v, dpp_mov<0x112, 0xf>(id, v));   // row_shr:2
    v = Op::f(v, dpp_mov<0x114, 0xf>(id, v));   // row_shr:4
    v = Op::f(v, dpp_mov<0x118, 0xf>(id, v));   // row_shr:8
    v = Op::f(v, dpp_mov<0x142, 0xa>(id, v));   // row_bcast:15 -> rows 1, 3
    v = Op::f(v, dpp_mov<0x143, 0xc>(id, v));   // row_bcast:31 -> rows 2, 3
    return v;
}
template <typename Op>
__device__ __forceinline__ int wave_scan_exclusive(int v) {
    return dpp_mov<0x138, 0xf>(Op::identity, wave_scan_inclusive<Op>(v));   // wave_shr:1
}
template <typename Op>
__device__ __forceinline__ int wave_scan_exclusive(int v, int &inclusive) {
    inclusive = wave_scan_inclusive<Op>(v);
    return dpp_mov<0x138, 0xf>(Op::identity, inclusive);
}

// ---------------------------------------------------------------------------------
// Planned path.  A workgroup = 8 waves = 8 rows of one chunk index; wave w owns row w.
//   P1  lanes 0..7 of wave 0 load the 8 rows' parameters (one lane per row), one barrier
//   P2  the row's variant records, lane j = variant j; rows with <= 8 variants classify
//       themselves: fast (SNPs only: the scan-free plan in reconstruct_kernel) or slow
//       (planned together by the first slow wave: packed_plan); the others run P3 per wave
//   P3  the reference's sequential walk, restated as wave-wide scans:
//         * shift: the lead pad absorbs it first; while it is open ref_idx does not move, so
//           the variants in front of the one that completes it are dropped and that one is
//           kept / cut at the front / consumed (a ballot + ctz)
//         * "first ALT wins": variant i is applied iff pos_i >= max(ref_idx0, v_end of every
//           applied variant before it) -- an exclusive prefix-max, iterated to its (unique)
//           fixed point when deletions knock out later variants
//         * output offsets of the applied variants: exclusive prefix-sum of
//           (reference run + allele length) -- the indel shift
//         * the loop's ">= L" break: applied = prefix of lanes whose allele starts before L
//         * segment table: one REF run + one ALLELE entry per applied indel (SNPs do not
//           split a run, they become patches), scattered to LDS at prefix-count slots
//   P3b lanes 0..7, one per trip, turn the table into trip descriptors: "uniform" trips
//       (inside ONE reference run) carry the source offset; plus the trip's patch slice
//   P4  the wave streams its row from the descriptors, all reference loads issued first
// Rows the scans do not take (> 64 variants, table overflow, coordinates >= 2^30, negative
// shift, no fixed point in 4 rounds) run recon_wave_scalar instead.
// ---------------------------------------------------------------------------------
struct RowIn {
    i64 c_s, R, ref_start, shift, o_s, keep_off, row_base;
    int k;                       // the row (list mode: rows of a workgroup are not consecutive)
    int n_var, L, rc, flags;     // flags: 1 = no work (row out of range / chunk past the row), 2 = scalar path, 4 = zero fill, 8 = packable (<= 8 variants),
                                 // 16 = records come from the slot-major table (o_s = the slot, n_var = 8 until the line is read)
};
template <bool ANNOT>
struct RowPlan {
    int s_out[SEG_CAP]; u32 s_lo[SEG_CAP], s_hi[SEG_CAP];
    int s_a[ANNOT ? SEG_CAP : 1], s_b[ANNOT ? SEG_CAP : 1];
    int p_out[WAVE], p_val[WAVE], p_id[ANNOT ? WAVE : 1];
};

// Trip descriptors of one row (what P3b produces), 8 trips per chunk; and the row's table sizes.
struct TripDesc {
    int cls[CHUNK_TRIPS], b1[CHUNK_TRIPS], b2[CHUNK_TRIPS], pc0[CHUNK_TRIPS], pcn[CHUNK_TRIPS], idx[CHUNK_TRIPS];
    u32 ldlo[CHUNK_TRIPS], ldhi[CHUNK_TRIPS];
    u32 lo0[CHUNK_TRIPS], hi0[CHUNK_TRIPS], lo1[CHUNK_TRIPS], hi1[CHUNK_TRIPS], lo2[CHUNK_TRIPS], hi2[CHUNK_TRIPS];
};
struct RowMeta { int nseg, npatch, bad, slow, ready, pad0_, pad1_, pad2_; };

// scans inside groups of 8 lanes (DPP row_shr 1/2/4, masked at the group boundary)
template <typename Op>
__device__ __forceinline__ int seg8_scan_inclusive(int v, int j) {
    constexpr int id = Op::identity;
    int t;
    t = dpp_mov<0x111, 0xf>(id, v); v = j >= 1 ? Op::f(v, t) : v;
    t = dpp_mov<0x112, 0xf>(id, v); v = j >= 2 ? Op::f(v, t) : v;
    t = dpp_mov<0x114, 0xf>(id, v); v = j >= 4 ? Op::f(v, t) : v;
    return v;
}
template <typename Op>
__device__ __forceinline__ int seg8_scan_exclusive(int v, int j) {
    const int inc = seg8_scan_inclusive<Op>(v, j);
    const int t = dpp_mov<0x111, 0xf>(Op::identity, inc);
    return j >= 1 ? t : Op::identity;
}

// ---------------------------------------------------------------------------------
// Packed plan: ONE wave plans all 8 rows of the workgroup at once, lane = (row r = lane / 8,
// variant j = lane % 8), for rows with at most 8 variants (99.7 % of cfg2 / cfg3 rows).  The
// same restatement of the walk as the per-wave P3 below, with the scans confined to groups of 8
// lanes; then P3b with lane = (row, trip).  It exists because the plan is a dependent
// instruction chain: eight waves each running it for one row take eight times the issue slots
// (and a row with an indel, whose plan is the longest, ends the launch), one wave running it
// for eight rows takes the same chain once.
// ---------------------------------------------------------------------------------
template <bool ANNOT>
__device__ __forceinline__ void packed_plan(const ReconArgs &A, const RowIn *rin, RowPlan<ANNOT> *plan, TripDesc *desc,
                                            RowMeta *meta, const i32x4 *lrec, const int lane, const int lo_clip, const bool has_keep) {
    const int r = lane >> 3, j = lane & 7, seg_base = lane & ~7;
    const RowIn &ri = rin[r];
    RowPlan<ANNOT> &pl = plan[r];
    const int rflags = ri.flags;
    const bool elig = (rflags & 9) == 8 && meta[r].slow != 0;
    const int L = ri.L;
    const i64 c_s = ri.c_s, R = ri.R;
    const int hi_clip = (L - lo_clip > A.chunk_len) ? lo_clip + A.chunk_len : L;
    const int n_var = ri.n_var;
    const i64 o_s = ri.o_s;
    const i64 rs64 = ri.ref_start;
    bool ok = rs64 > -(1 << 30) && rs64 < (1 << 30);
    const int ref_start = ok ? (int)rs64 : 0;
    const int shift_i = (int)ri.shift;                 // [0, 2^30): checked in P1
    auto seg_byte = [&](u64 m) -> u32 { return (u32)(m >> seg_base) & 0xFFu; };

    // ---- P2: records ----------------------------------------------------------------
    int pos = 0, d = 0, alen = 0, inl = 0, vi = 0; i64 a0 = 0;
    const bool ell = (rflags & 16) != 0;               // slot-major records: 8 entries at srec[slot * 8]
    bool valid = elig && j < n_var;
    if (valid) {
        if (ell) {
            const i32x4 rec = lrec[lane];                  // parked in LDS by P1: lane = row x 8 + entry
            const u32 e = (u32)rec.z;
            valid = e != GVL_SREC_EMPTY;
            pos = rec.x; d = rec.y; alen = (int)(e >> 8); inl = (int)(e & 0xFF); vi = 0;
            a0 = (i64)(u32)rec.w;                          // no alt_offsets hop
        } else if (A.grec) {
            const i32x4 rec = *reinterpret_cast<const i32x4 *>(A.grec + (o_s + j));
            pos = rec.x; d = rec.y; alen = (int)((u32)rec.z >> 8); inl = rec.z & 0xFF; vi = rec.w;
            if (!(d == 0 && alen == 1)) a0 = A.alt_offsets[vi];
        } else {
            int v = A.geno_v_idxs[o_s + j];
            v = v < 0 ? 0 : ((i64)v >= A.n_variants ? (int)(A.n_variants - 1) : v);
            const i32x4 rec = *reinterpret_cast<const i32x4 *>(A.vrec + v);
            a0 = A.alt_offsets[v];
            pos = rec.x; d = rec.y; alen = rec.z; inl = rec.w; vi = v;
        }
        if (valid && has_keep) valid = A.keep[ri.keep_off + j] != 0;
    }
    const bool weird = valid && (pos < 0 || pos >= (1 << 30) || d <= -(1 << 30) || d >= (1 << 30) || alen < 0 ||
                                 alen >= ((ell || A.grec) ? 0xFFFFFF : (1 << 30)));
    ok = ok && seg_byte(__builtin_amdgcn_ballot_w64(weird)) == 0;
    if (!ok) valid = false;

    // ---- P3 ---------------------------------------------------------------------------
    const int raw = ref_start < 0 ? -ref_start : 0;
    const int shifted0 = shift_i < raw ? shift_i : raw;
    const int n_lead = (raw - shifted0 < L) ? raw - shifted0 : L;
    int rem = shift_i - shifted0;
    int ref_idx0 = ref_start < 0 ? 0 : ref_start;
    const int lead_kept = (n_lead > 0 && n_lead > lo_clip && 0 < hi_clip) ? 1 : 0;
    const int E = pos - (d < 0 ? d : 0) + 1;
    const bool is_snp = d == 0 && alen == 1;
    {   // DEL spanning the window start: the last one in order sets ref_idx
        const u32 b = seg_byte(__builtin_amdgcn_ballot_w64(valid && pos < ref_start && d < 0 && E >= ref_start));
        const int src = seg_base + (b ? 31 - __builtin_clz(b) : 0);
        const int e_src = bperm(src, E);
        if (b) ref_idx0 = e_src;
    }
    bool cand = valid && pos >= ref_start;
    int a_skip = 0;     // a0 (a second dependent load for indel lanes) is first touched at the very end
    {   // shift consumption
        const int base = ref_idx0;
        const u32 b = seg_byte(__builtin_amdgcn_ballot_w64(rem > 0 && cand && pos >= base && (pos - base) + alen >= rem));
        const int f = b ? __builtin_ctz(b) : 0;
        const int fl = seg_base + f;
        const int pos_f = bperm(fl, pos), alen_f = bperm(fl, alen), E_f = bperm(fl, E);
        if (rem > 0) {
            if (b == 0) {
                cand = false;
            } else {
                const int dist = pos_f - base;
                if (dist >= rem) {
                    ref_idx0 = base + rem;
                    cand = cand && j >= f;
                } else {
                    const int skip = rem - dist;
                    if (skip == alen_f) {
                        ref_idx0 = E_f;
                        cand = cand && j > f;
                    } else {
                        ref_idx0 = pos_f;
                        cand = cand && j >= f;
                        if (j == f) { alen -= skip; a_skip = skip; }
                    }
                }
                rem = 0;
            }
        }
    }
    const int pm_carry = ref_idx0;
    // first ALT wins: fixed point, all rows at once
    bool inB = cand;
    int PM = 0;
    {
        u64 mB = __builtin_amdgcn_ballot_w64(inB);
        bool stable = false;
#pragma unroll 1
        for (int it = 0; it < 4 && !stable; ++it) {
            PM = seg8_scan_exclusive<OpMaxU>(inB ? E : 0, j);
            PM = PM > pm_carry ? PM : pm_carry;
            inB = cand && pos >= PM;
            const u64 m2 = __builtin_amdgcn_ballot_w64(inB);
            stable = m2 == mB;
            if (it == 3) ok = ok && seg_byte(m2) == seg_byte(mB);
            mB = m2;
        }
    }
    const int n_i = inB ? pos - PM : 0;
    const int S_i = inB ? OpSat::f(n_i, alen) : 0;
    const int X = seg8_scan_exclusive<OpSat>(S_i, j);
    const int allele_out = OpSat::f(OpSat::f(n_lead, X), n_i);
    const bool applied = inB && allele_out < L;
    const int w_i = applied ? ((alen < L - allele_out) ? alen : L - allele_out) : 0;
    const bool nonsnp = applied && !is_snp;
    const bool snp = applied && is_snp;
    const u32 b_app = seg_byte(__builtin_amdgcn_ballot_w64(applied));
    const bool any_applied = b_app != 0;
    int ref_idx_end = ref_idx0, out_idx_end = n_lead;
    {
        const int last = seg_base + (b_app ? 31 - __builtin_clz(b_app) : 0);
        const int e_l = bperm(last, E), o_l = bperm(last, allele_out + w_i);
        if (b_app) { ref_idx_end = e_l; out_idx_end = o_l; }
    }
    const int prevNS = seg8_scan_exclusive<OpMaxI>(nonsnp ? j : -1, j);
    const int pidx = seg_base + (prevNS < 0 ? 0 : prevNS);
    const int p_end = bperm(pidx, allele_out + alen);
    const int p_E = bperm(pidx, E);
    const int run_start = prevNS < 0 ? n_lead : p_end;
    const i64 run_src = c_s + (prevNS < 0 ? ref_idx0 : p_E);
    const bool e_ref = nonsnp && allele_out > run_start && allele_out > lo_clip && run_start < hi_clip;
    const bool e_all = nonsnp && w_i > 0 && allele_out + w_i > lo_clip && allele_out < hi_clip;
    const int slot0 = lead_kept + seg8_scan_exclusive<OpAdd>((e_ref ? 1 : 0) + (e_all ? 1 : 0), j);
    const u32 b_ns = seg_byte(__builtin_amdgcn_ballot_w64(nonsnp));
    const u32 b_snp = seg_byte(__builtin_amdgcn_ballot_w64(snp && allele_out >= lo_clip && allele_out < hi_clip));
    const int n_ent = lead_kept + __builtin_popcount(seg_byte(__builtin_amdgcn_ballot_w64(e_ref))) +
                      __builtin_popcount(seg_byte(__builtin_amdgcn_ballot_w64(e_all)));
    const int npatch = __builtin_popcount(b_snp);
    auto put = [&](int q, u32 kind, int o_start, i64 delta, int id, int vpos) {
        const u64 e = (u64)(delta + DELTA_BIAS) | ((u64)kind << 62);
        pl.s_out[q] = o_start; pl.s_lo[q] = (u32)e; pl.s_hi[q] = (u32)(e >> 32);
        if (ANNOT) { pl.s_a[q] = id; pl.s_b[q] = vpos; }
    };
    if (ok && elig) {
        int q = slot0;
        if (e_ref) { put(q, K_REF, run_start, run_src - run_start, -1, -1); ++q; }
        if (e_all) put(q, K_ALLELE, allele_out, a0 + a_skip - allele_out, vi, pos);
        if ((b_snp >> j) & 1u) {
            const int ps = __builtin_popcount(b_snp & ((1u << j) - 1u));
            pl.p_out[ps] = allele_out; pl.p_val[ps] = inl & 0xFF;
            if (ANNOT) pl.p_id[ps] = vi;
        }
    }
    // last applied indel of the row -> start of the tail run
    int ns_end = 0, ns_E = 0;
    {
        const int ln = seg_base + (b_ns ? 31 - __builtin_clz(b_ns) : 0);
        ns_end = bperm(ln, allele_out + alen);
        ns_E = bperm(ln, E);
    }
    const bool have_ns = b_ns != 0;
    if (rem > 0) ref_idx_end = (int)imin((i64)ref_idx0 + rem, R);
    const int t_start = have_ns ? ns_end : n_lead;
    const i64 t_src = c_s + (have_ns ? ns_E : ((rem > 0 && !any_applied) ? ref_idx_end : ref_idx0));
    int t_end = out_idx_end;
    {
        const int u = L - out_idx_end;
        if (u > 0) {
            const int w = (int)imin((i64)u, R - ref_idx_end);
            if (w > 0) t_end = out_idx_end + w;
        }
    }
    const bool tail_pad = t_end < L && L > lo_clip && t_end < hi_clip;
    const bool tail_ref = t_end > t_start && t_end > lo_clip && t_start < hi_clip;
    const int nseg = n_ent + (tail_ref ? 1 : 0) + (tail_pad ? 1 : 0);
    if (ok && elig) {
        if (j == 0) {
            if (lead_kept) put(0, K_PAD_LEAD, 0, 0, -1, -1);
            int q = n_ent;
            if (tail_ref) { put(q, K_REF, t_start, t_src - t_start, -1, -1); ++q; }
            if (tail_pad) put(q, K_PAD_TRAIL, t_end, 0, -1, -1);
            meta[r].nseg = nseg; meta[r].npatch = npatch;
        }
        if (j >= nseg) pl.s_out[j] = 0x7FFFFFFF;       // sentinels for the 8-wide reads of P3b
        if (j >= npatch) pl.p_out[j] = 0x7FFFFFFF;
    }
    if (j == 0 && elig) meta[r].bad = ok ? 0 : 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- P3b: lane = (row r, trip u = j) -------------------------------------------------
    if (ok && elig) {
        const int u = j;
        const int limit = hi_clip;
        const int p0 = lo_clip + u * TRIP;
        TripDesc &D = desc[r];
        int cls = 3, b1 = limit, b2 = limit, b3 = limit, idx = 0, pc0 = 0, pcn = 0;
        u32 lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0, ldlo = 0, ldhi = 0;
        if (p0 < limit) {
            const int t_end2 = (limit - p0 > TRIP) ? p0 + TRIP : limit;
            {
                int so[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) so[t] = pl.s_out[t];
#pragma unroll
                for (int t = 1; t < 8; ++t) idx += (so[t] <= p0) ? 1 : 0;
                for (int s2 = 8; s2 < nseg; ++s2) idx += (pl.s_out[s2] <= p0) ? 1 : 0;
            }
            auto at = [&](int i) { return i < SEG_CAP ? i : SEG_CAP - 1; };
            b1 = idx + 1 < nseg ? pl.s_out[at(idx + 1)] : limit;
            b2 = idx + 2 < nseg ? pl.s_out[at(idx + 2)] : limit;
            b3 = idx + 3 < nseg ? pl.s_out[at(idx + 3)] : limit;
            lo0 = pl.s_lo[at(idx)]; hi0 = pl.s_hi[at(idx)];
            lo1 = pl.s_lo[at(idx + 1)]; hi1 = pl.s_hi[at(idx + 1)];
            lo2 = pl.s_lo[at(idx + 2)]; hi2 = pl.s_hi[at(idx + 2)];
            cls = b1 >= t_end2 ? 0 : (b2 >= t_end2 ? 1 : (b3 >= t_end2 ? 2 : 3));
            auto in_bounds = [&](u32 lo, u32 hi, int s, int e) {
                const u32 kind = hi >> 30;
                if (kind != K_REF && kind != K_ALLELE) return true;
                const i64 dl = seg_delta(lo, hi);
                const i64 len = kind == K_REF ? A.ref_len : A.alt_len;
                const int s3 = (s - 3 > p0 ? s - 3 : p0), e3 = (e + 3 < t_end2 ? e + 3 : t_end2);
                return dl + s3 >= 0 && dl + e3 <= len;
            };
            bool okb = in_bounds(lo0, hi0, p0, b1 < t_end2 ? b1 : t_end2);
            if (cls >= 1 && cls < 3) okb = okb && in_bounds(lo1, hi1, b1, b2 < t_end2 ? b2 : t_end2);
            if (cls == 2) okb = okb && in_bounds(lo2, hi2, b2, t_end2);
            if (!okb || ((t_end2 - p0) & 3) != 0 || (rflags & 4)) cls = 3;
            if (cls == 0) {
                const u32 kind = hi0 >> 30;
                if (kind == K_REF || kind == K_ALLELE) {
                    const u64 ad = (u64)(kind == K_REF ? A.ref : A.alt_alleles) + (u64)(seg_delta(lo0, hi0) + p0);
                    ldlo = (u32)ad; ldhi = (u32)(ad >> 32);
                } else {
                    cls = 3;
                }
            }
            {
                int po[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) po[t] = pl.p_out[t];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    pc0 += (po[t] < p0) ? 1 : 0;
                    pcn += (po[t] < t_end2) ? 1 : 0;
                }
            }
        }
        D.cls[u] = cls; D.b1[u] = b1; D.b2[u] = b2; D.pc0[u] = pc0; D.pcn[u] = pcn; D.idx[u] = idx;
        D.ldlo[u] = ldlo; D.ldhi[u] = ldhi;
        D.lo0[u] = lo0; D.hi0[u] = hi0; D.lo1[u] = lo1; D.hi1[u] = hi1; D.lo2[u] = lo2; D.hi2[u] = hi2;
    }
    // release the rows' own waves (they poll meta[r].ready)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (elig && j == 0) __hip_atomic_store(&meta[r].ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#ifdef GVL_DIAG
#define GVL_STAMP(i) do { if (A.stamps && tid == 0) A.stamps[((u64)blockIdx.x * gridDim.y + blockIdx.y) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GVL_STAMP(i) do { } while (0)
#endif

template <bool ANNOT>
struct ReconShared {
    Luts luts;
    SegMirror mirror[WG_WAVES];
    Stage<ANNOT> stage[WG_WAVES];
    RowIn rin[WG_WAVES];
    RowPlan<ANNOT> plan[WG_WAVES];
    TripDesc desc[WG_WAVES];
    RowMeta meta[WG_WAVES];
    i32x4 lrec[WG_WAVES * GVL_SLOT_RECS];   // slot-major records of the 8 rows (read once, by wave 0)
};

// What one wave needs in LDS when it runs a row on its own (SOLO): the scalar walk's mirror only ever
// replaces a plan that failed, so the two share their memory.
template <bool ANNOT>
struct SoloLds {
    union { RowPlan<ANNOT> pl; SegMirror M; } u;
    Stage<ANNOT> G;
    RowIn ri;
};

// One workgroup's 8 (row, chunk)s: rows 8 wg .. 8 wg + 7 of the batch (S = the workgroup's tables).
// SOLO: ONE wave runs ONE row on its own -- the caller (recon_lean_kernel, for a row it cannot express)
// has put the row's parameters into solo->ri; no P1, no barrier, no packed plan: the row takes the
// per-wave scans (or the scalar walk) and streams from the byte reference like any row here.
template <int OH, bool HAPS, bool ANNOT, bool SOLO>
__device__ __forceinline__ void recon_body(const ReconArgs &A, ReconShared<ANNOT> *Sp, SoloLds<ANNOT> *solo, Luts &luts, const i64 wg, const int chunk) {
    SegMirror *const mirror = SOLO ? nullptr : Sp->mirror;
    Stage<ANNOT> *const stage = SOLO ? nullptr : Sp->stage;
    RowIn *const rin = SOLO ? nullptr : Sp->rin;
    RowPlan<ANNOT> *const plan = SOLO ? nullptr : Sp->plan;
    TripDesc *const desc = SOLO ? nullptr : Sp->desc;
    RowMeta *const meta = SOLO ? nullptr : Sp->meta;
    i32x4 *const lrec = SOLO ? nullptr : Sp->lrec;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = rfl(tid >> 6);
    const int lo_clip = chunk * A.chunk_len;
    const bool has_keep = A.keep && A.keep_offsets;
    const bool planned_ok = A.chunk_len <= CHUNK_TRIPS * TRIP && !(A.dbg & 8);

    GVL_STAMP(0);
    if (!SOLO && tid < 256) {  // LUTs
        const u32 d = onehot_dword((u32)tid);
        luts.oh[tid] = d;
        luts.oh_rc[tid] = __builtin_bswap32(d);
        luts.comp[tid] = comp_byte((u32)tid);
    }
    // ---- P1 (wave 0): row parameters, one lane per row; then -- second memory level, side by side --
    // the contig bounds and the rows' slot-major variant records (lane = row x 8 + entry: the eight
    // 128-byte lines of the workgroup in ONE wave-load), all parked in LDS.  After the barrier no wave
    // needs a global read before its plan, so the reference bytes it requests next (below) stay in
    // flight behind nothing.
    const bool use_srec = !SOLO && A.srec != nullptr && !A.ref_only && planned_ok && !(A.dbg & 512) && !ANNOT;
    // wave 0's carry from P1a to P1b (the slot-line read stays in flight across the first barrier)
    i32x4 p1_rec = {0, 0, (int)GVL_SREC_EMPTY, 0};
    i64 p1_oidx = 0;
    int p1_fl = 1;
    bool p1_want = false;
    if (!SOLO && tid < WAVE) {
        // Every load of a level is issued before anything waits (no branch in between: an absent
        // array is replaced by a pointer that is always readable, and a lane without a row reads
        // row 0), so P1 costs two memory round trips, not one per array.
        const i64 k_raw = wg * WG_WAVES + tid;
        const bool row_lane = tid < WG_WAVES && k_raw < A.n_rows;
        const i64 k = row_lane ? k_raw : 0;
        const i64 query = A.ploidy_shift >= 0 ? (k >> A.ploidy_shift) : (i64)((u32)k / (u32)A.ploidy);
        const int *reg = A.regions + query * A.regions_stride;
        const int *const dmy4 = A.regions;                                       // >= 12 readable bytes
        const i64 *const dmy8 = reinterpret_cast<const i64 *>(A.ref_offsets);    // >= 8 readable bytes
        // ---- level 1
        const int l_c = reg[0], l_start = reg[1], l_end = reg[2];
        const int l_shift = *(A.ref_only ? dmy4 : A.shifts + k);
        const i64 l_oidx = *(A.ref_only ? dmy8 : A.geno_offset_idx + k);
        const u8 l_rc = *(A.to_rc ? A.to_rc + k : reinterpret_cast<const u8 *>(dmy4));
        const i64 l_oo0 = *(A.out_offsets ? A.out_offsets + k : dmy8);
        const i64 l_oo1 = *(A.out_offsets ? A.out_offsets + k + 1 : dmy8);
        const i64 l_ko = *(has_keep && !A.ref_only ? A.keep_offsets + k : dmy8);
        RowIn ri;
        ri.c_s = ri.R = ri.o_s = 0;
        ri.n_var = 0;
        ri.k = (int)k;
        ri.ref_start = l_start;
        ri.shift = A.ref_only ? 0 : (i64)l_shift;
        ri.keep_off = (has_keep && !A.ref_only) ? l_ko : 0;
        ri.rc = A.to_rc ? (int)l_rc : 0;
        ri.row_base = A.out_offsets ? l_oo0 : k * A.fixed_len;
        ri.L = A.out_offsets ? (int)(l_oo1 - l_oo0) : (int)A.fixed_len;
        const i64 o_idx = A.ref_only ? 0 : l_oidx;
        int fl = 0;
        if (A.ref_only && (i64)l_start >= (i64)l_end) fl |= 4;                   // reference/mod.rs:16-18
        const bool shift_ok = !(ri.shift < 0 || ri.shift >= (1 << 30));
        if (!shift_ok || !planned_ok) fl |= 2;
        if (use_srec && shift_ok && !(A.dbg & 1)) fl |= 16;
        // a row longer than the caller's max_row_len hint would be left partly unwritten: report it
        if (row_lane && chunk == 0 && A.async_err && (i64)ri.L > (i64)gridDim.y * (i64)A.chunk_len) *A.async_err = 1;
        if (!row_lane || lo_clip >= ri.L) fl = 1;
        if (row_lane && A.out_offsets_w && chunk == 0) {
            A.out_offsets_w[k] = ri.row_base;
            if (k == A.n_rows - 1) A.out_offsets_w[k + 1] = ri.row_base + ri.L;
        }
        // ---- level 2: contig bounds (L2 resident), CSR bounds (rows that do not use the slot-major
        // records), and LAST the rows' slot-major records, lane (r, j) = entry j of row r's slot: the
        // first two are waited for here, the slot lines stay in flight across the first barrier (P1b)
        const i64 c_idx = (l_c >= 0 && l_c < A.n_contigs) ? (i64)l_c : 0;       // (out of contract otherwise: clamp)
        const bool csr = !A.ref_only && !(fl & 17);
        const i64 o_safe = (csr && o_idx >= 0 && o_idx < A.n_geno_offsets) ? o_idx : 0;
        const i64 l_cs = A.ref_offsets[c_idx], l_ce = A.ref_offsets[c_idx + 1];
        const i64 l_gs = *(csr ? A.go_starts + o_safe : dmy8);
        const i64 l_ge = *(csr ? A.go_stops + o_safe : dmy8);
        if (use_srec) {
            const int r = lane >> 3;
            const int fl_r = bperm(r, fl);
            const u32 o_lo = (u32)bperm(r, (int)(u32)(u64)o_idx), o_hi = (u32)bperm(r, (int)(u32)((u64)o_idx >> 32));
            const i64 o_r = (i64)(((u64)o_hi << 32) | o_lo);
            p1_want = (fl_r & 17) == 16 && o_r >= 0 && o_r < A.n_geno_offsets;
            p1_rec = *reinterpret_cast<const i32x4 *>(A.srec + ((p1_want ? o_r : 0) * GVL_SLOT_RECS + (lane & 7)));
        }
        if (fl != 1) {
            ri.c_s = l_cs;
            ri.R = l_ce - l_cs;
            if (fl & 16) {
                // the variant count is known once the line is read (a slot with more than 8 falls
                // back to the CSR: P1b)
                ri.o_s = o_idx;
                ri.n_var = GVL_SLOT_RECS;
            } else if (csr) {
                ri.o_s = l_gs;
                const i64 nv = l_ge - l_gs;
                ri.n_var = nv < 0 ? 0 : (nv > 0x7FFFFFFFll ? 0x7FFFFFFF : (int)nv);
                if (A.dbg & 1) ri.n_var = 0;
            }
            if (!(fl & 2) && ri.n_var <= 8 && !(A.dbg & 512)) fl |= 8;           // packable: planned with the other rows
        }
        ri.flags = fl;
        p1_fl = fl; p1_oidx = o_idx;
        if (tid < WG_WAVES) {
            rin[tid] = ri;
            RowMeta m0; m0.nseg = m0.npatch = m0.bad = m0.slow = m0.ready = m0.pad0_ = m0.pad1_ = m0.pad2_ = 0;
            meta[tid] = m0;
        }
    }
    GVL_STAMP(1);
    if (!SOLO) lds_barrier();      // (not __syncthreads: its fence would wait for wave 0's slot-line read)
    // Every wave "uses" the slot-line registers here: a no-op for waves 1..7 (they have nothing in flight),
    // wave 0 needs the line next anyway.  Without it the compiler's wait-count model carries "a load into
    // these registers may be pending" past the reference reads below and, as soon as a register is reused,
    // waits for ALL of them (vmcnt(0) right after the second barrier).
    asm volatile("" :: "v"(p1_rec.x), "v"(p1_rec.y), "v"(p1_rec.z), "v"(p1_rec.w));

    GVL_STAMP(2);
    const RowIn &ri = SOLO ? solo->ri : rin[wave];
    int flags = rfl(ri.flags);
    const i64 k = SOLO ? (i64)rfl(ri.k) : wg * WG_WAVES + wave;
    RowPlan<ANNOT> &pl = SOLO ? solo->u.pl : plan[wave];
    Stage<ANNOT> &G = SOLO ? solo->G : stage[wave];
    const int L = rfl(ri.L);
    const i64 c_s = rfl64(ri.c_s);
    const int hi_clip = (L - lo_clip > A.chunk_len) ? lo_clip + A.chunk_len : L;
    int nseg = 0, npatch = 0;

    // ---- who plans this row ---------------------------------------------------------------
    //  * rows with at most 8 variants ("packable", flags & 8) read their records here, lane j =
    //    variant j.  If every kept variant is a SNP inside the contig and no two share a position
    //    the row is FAST: no scan is needed, its own wave plans it below and starts streaming
    //    while the others are still planning.
    //  * the other packable rows are SLOW: the first slow wave plans all of them at once
    //    (packed_plan), the other slow waves wait for their flag.
    //  * rows with more than 8 variants run the per-wave scans (P2 + P3 further down).
    bool packable = !SOLO && (flags & 11) == 8;
    int row_n_var = rfl(ri.n_var);        // (a slot-major row that overflows its line re-reads these from the CSR)
    i64 row_o_s = rfl64(ri.o_s);

    // ---- the "one reference run" reading of the row: n_lead pad bytes, then the contig from the
    // shifted origin r0, then pad.  It depends on the row parameters only, it is exact for a row
    // without indels and for the part of any row in front of its first indel -- so its reference
    // bytes are requested NOW, next to the variant records, instead of after the plan: one dependent
    // memory level less for SNP-only rows, and a prefetch into L2 for the others.
    int g_n_lead = 0, g_t_end = 0, g_r0 = 0; i64 g_delta = 0; bool g_ok = false;
    {
        const i64 rs = rfl64(ri.ref_start);
        if (!(flags & 3) && rs > -(1 << 30) && rs < (1 << 30)) {
            const int ref_start = (int)rs;
            const i64 R = rfl64(ri.R);
            const int shift_i = (int)rfl64(ri.shift);
            const int raw = ref_start < 0 ? -ref_start : 0;
            const int shifted0 = shift_i < raw ? shift_i : raw;
            g_n_lead = (raw - shifted0 < L) ? raw - shifted0 : L;
            g_r0 = (int)imin((i64)(ref_start < 0 ? 0 : ref_start) + (shift_i - shifted0), R);
            g_t_end = g_n_lead;
            const int u = L - g_n_lead;
            if (u > 0) {
                const int w = (int)imin((i64)u, R - g_r0);
                if (w > 0) g_t_end = g_n_lead + w;
            }
            g_delta = c_s + g_r0 - g_n_lead;
            g_ok = true;
        }
    }
    const int lane4 = GROUP * lane;
    // slot-major records of this row: parked in LDS by wave 0 (no global read, no vmcnt wait).  Then the
    // speculative reference reads: on the slot-major path nothing between here and pass A waits for a
    // global load, so they stay in flight while the row is classified and planned.
    bool ell = (flags & 16) != 0 && packable;        // (tentative until P1b has seen the slot line)
    const bool sp_on = g_ok && !(flags & 4) && !(A.dbg & (4 | 128)) && (ell || row_n_var == 0);
    // lane u decides for trip u (full trips only), one ballot; the loads then differ by an immediate
    // offset only.  (The scalar unit is shared by the waves of a CU: per-trip scalar arithmetic in the
    // head of every wave is what this avoids.)
    const u8 *const sp_base = A.ref + (g_delta + lo_clip) + (u32)lane4;
    u32 spmask = 0;
    if (sp_on) {
        const int p0 = lo_clip + lane * TRIP;
        const bool inside = lane < CHUNK_TRIPS && p0 >= g_n_lead && p0 + TRIP <= g_t_end && p0 + TRIP <= hi_clip &&
                            g_delta + p0 >= 0 && g_delta + p0 + TRIP <= A.ref_len;
        spmask = (u32)__builtin_amdgcn_ballot_w64(inside);
    }
    u32 wq[CHUNK_TRIPS];
    auto issue_spec = [&]() {
#pragma unroll
        for (int u = 0; u < CHUNK_TRIPS; ++u) {
            wq[u] = 0;
            if ((spmask >> u) & 1u) wq[u] = load_u32_unaligned(sp_base + u * TRIP);
        }
    };
    // waves 1..7 request their reference bytes NOW, under wave 0's slot-line read; wave 0 first parks the
    // slot lines (its own reads would otherwise sit in front of them: loads return in order)
    const bool spec_late = use_srec && wave == 0;
    if (!spec_late) issue_spec();
#ifdef GVL_DIAG
    // diagnostic (GVL_DBG 256): when do wave 1's speculative reference bytes arrive?  (stamp 15)
    if (A.stamps && (A.dbg & 256) && wave == 1) {
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) A.stamps[((u64)blockIdx.x * gridDim.y + blockIdx.y) * 16 + 15] = __builtin_amdgcn_s_memrealtime();
    }
#endif

    // Who plans a row: see below (SNP-only rows plan themselves, the first wave that holds a row with
    // an indel plans all such rows of the workgroup).  Measured and dropped in round 2: wave 0 planning
    // EVERY packable row lane-parallel right after P1 while the other waves only keep their reference
    // reads in flight -- a third of the issue slots, but the packed plan is a 3.6 us chain of dependent
    // instructions when one wave runs it alone, and every row then waits for it (cfg3 14.5 vs 12.8 us,
    // cfg2 14.0 vs 12.0 us per launch on the same box).
    // ---- P1b (wave 0): the slot lines have arrived by now (they were requested before the reference
    // reads above, and loads return in order): park them in LDS; a slot with more than 8 variants sends
    // its row through the CSR (third level; rare).  Second LDS-only barrier.
    if (use_srec && tid < WAVE) {
        i32x4 rec = p1_rec;
        if (!p1_want) { rec.x = 0; rec.y = 0; rec.z = (int)GVL_SREC_EMPTY; rec.w = 0; }
        lrec[lane] = rec;
        const u64 ovf = __builtin_amdgcn_ballot_w64(p1_want && (lane & 7) == 0 && (u32)rec.z == GVL_SREC_OVERFLOW);
        if (ovf && tid < WG_WAVES && ((ovf >> (8 * tid)) & 1ull)) {
            const i64 gs = A.go_starts[p1_oidx];
            const i64 nv = A.go_stops[p1_oidx] - gs;
            rin[tid].o_s = gs;
            rin[tid].n_var = nv < 0 ? 0 : (nv > 0x7FFFFFFFll ? 0x7FFFFFFF : (int)nv);
            rin[tid].flags = p1_fl & ~(8 | 16);          // more than 8 variants: not packable, per-wave scans
        }
    }
    if (use_srec) {
        lds_barrier();
        flags = rfl(ri.flags);
        packable = !SOLO && (flags & 11) == 8;
        row_n_var = rfl(ri.n_var);
        row_o_s = rfl64(ri.o_s);
        ell = (flags & 16) != 0 && packable;
    }
    if (spec_late) issue_spec();
    i32x4 srec_v = {0, 0, 0, 0};
    if (ell) srec_v = lrec[wave * GVL_SLOT_RECS + (lane & (GVL_SLOT_RECS - 1))];
    int f_pos = 0, f_inl = 0, f_vi = 0;
    bool f_valid = false, is_fast = false;
    if (packable) {
        int d = 0, alen = 0;
        bool rec_valid;
        if (ell) {
            const u32 e = (u32)srec_v.z;
            rec_valid = lane < GVL_SLOT_RECS && e != GVL_SREC_EMPTY;
            f_pos = srec_v.x; d = srec_v.y; alen = (int)(e >> 8); f_inl = (int)(e & 0xFF); f_vi = 0;
        } else {
            rec_valid = lane < row_n_var;
            if (rec_valid) {
                if (A.grec) {
                    const i32x4 rec = *reinterpret_cast<const i32x4 *>(A.grec + (row_o_s + lane));
                    f_pos = rec.x; d = rec.y; alen = (int)((u32)rec.z >> 8); f_inl = rec.z & 0xFF; f_vi = rec.w;
                } else {
                    int v = A.geno_v_idxs[row_o_s + lane];
                    v = v < 0 ? 0 : ((i64)v >= A.n_variants ? (int)(A.n_variants - 1) : v);
                    const i32x4 rec = *reinterpret_cast<const i32x4 *>(A.vrec + v);
                    f_pos = rec.x; d = rec.y; alen = rec.z; f_inl = rec.w & 0xFF; f_vi = v;
                }
            }
        }
        f_valid = rec_valid;
        if (f_valid && has_keep) f_valid = A.keep[rfl64(ri.keep_off) + lane] != 0;
        const i64 rs = rfl64(ri.ref_start);
        const int pos_prev = dpp_mov<0x138, 0xf>(-1, f_pos);               // wave_shr:1
        const u64 m_slow = __builtin_amdgcn_ballot_w64(
            (f_valid && (d != 0 || alen != 1 || f_pos < 0 || (i64)f_pos + 1 > rfl64(ri.R))) ||
            (lane > 0 && rec_valid && f_pos == pos_prev));
        is_fast = packable && m_slow == 0 && rs > -(1 << 30) && rs < (1 << 30) && !(A.dbg & 32);
        // tell the workgroup: the slow flag, then one tick of the "decided" counter (meta[0].pad0_)
        if (lane == 0) {
            if (packable && !is_fast) meta[wave].slow = 1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_fetch_add(&meta[0].pad0_, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    GVL_STAMP(3);
    // Fast rows go on at once.  A slow row waits until every packable row has decided (no
    // workgroup barrier: the fast waves must not wait for anybody), then the first slow wave
    // plans all slow rows and the others poll their flag.
    const bool slow_row = packable && !is_fast;
    if (slow_row) {
        const int n_packable = __builtin_popcountll(
            __builtin_amdgcn_ballot_w64(lane < WG_WAVES && (rin[lane < WG_WAVES ? lane : 0].flags & 11) == 8));
        while (__hip_atomic_load(&meta[0].pad0_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < n_packable)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const u64 m_slow_rows = __builtin_amdgcn_ballot_w64(lane < WG_WAVES && meta[lane < WG_WAVES ? lane : 0].slow != 0);
        if (wave == __builtin_ctzll(m_slow_rows)) {
            __builtin_amdgcn_s_setprio(3);      // the other slow rows wait for this wave
            packed_plan<ANNOT>(A, rin, plan, desc, meta, lrec, lane, lo_clip, has_keep);
        } else {
            while (__hip_atomic_load(&meta[wave].ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
                __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    const bool packed = slow_row;
    const bool fast_row = packable && is_fast;
    int fp_n_lead = 0, fp_t_end = 0, fp_lead_kept = 0; bool fp_ref = false, fp_pad = false; i64 fp_delta = 0;

    if (fast_row) {
        // ---- fast plan: ONE reference run from the shifted origin r0 + SNP patches.  With SNPs only,
        // whichever way the shift completes (:115-146) or runs out (:200-205), ref_idx lands on
        // ref_idx0 + shift; SNP i sits at n_lead + (pos_i - r0) iff pos_i >= r0 and that is
        // inside the row (:154-158).
        const int n_lead = g_n_lead, r0 = g_r0, t_end = g_t_end;      // (is_fast implies g_ok)
        const int ao = n_lead + (f_pos - r0);
        const bool app = f_valid && f_pos >= r0 && (f_pos - r0) < (L - n_lead);
        const u64 m_p = __builtin_amdgcn_ballot_w64(app && ao >= lo_clip && ao < hi_clip);
        if ((m_p >> lane) & 1ull) {
            const int ps = __builtin_popcountll(m_p & ((1ull << lane) - 1ull));
            pl.p_out[ps] = ao; pl.p_val[ps] = f_inl;
            if (ANNOT) pl.p_id[ps] = f_vi;
        }
        npatch = __builtin_popcountll(m_p);
        const int lead_kept = (n_lead > 0 && n_lead > lo_clip && 0 < hi_clip) ? 1 : 0;
        const bool tail_pad = t_end < L && L > lo_clip && t_end < hi_clip;
        const bool tail_ref = t_end > n_lead && t_end > lo_clip && n_lead < hi_clip;
        nseg = lead_kept + (tail_ref ? 1 : 0) + (tail_pad ? 1 : 0);
        fp_n_lead = n_lead; fp_t_end = t_end; fp_lead_kept = lead_kept; fp_ref = tail_ref; fp_pad = tail_pad;
        fp_delta = c_s + r0 - n_lead;
        if (lane == 0) {
            auto put1 = [&](int q, u32 kind, int o_start, i64 delta) {
                const u64 e = (u64)(delta + DELTA_BIAS) | ((u64)kind << 62);
                pl.s_out[q] = o_start; pl.s_lo[q] = (u32)e; pl.s_hi[q] = (u32)(e >> 32);
                if (ANNOT) { pl.s_a[q] = -1; pl.s_b[q] = -1; }
            };
            int q = 0;
            if (lead_kept) { put1(q, K_PAD_LEAD, 0, 0); ++q; }
            if (tail_ref) { put1(q, K_REF, n_lead, c_s + r0 - n_lead); ++q; }
            if (tail_pad) put1(q, K_PAD_TRAIL, t_end, 0);
        }
        if (lane < 8) {
            if (lane >= nseg) pl.s_out[lane] = 0x7FFFFFFF;
            if (lane >= npatch) pl.p_out[lane] = 0x7FFFFFFF;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    if (!(flags & 3) && !packable) {
        // ---- P2 + P3, 64 variants per trip with carries between trips ------------------------
        const int n_var = row_n_var;
        const i64 o_s = row_o_s;
        const int ref_start = (int)rfl64(ri.ref_start);
        const i64 R = rfl64(ri.R);
        const i64 keep_off = has_keep ? rfl64(ri.keep_off) : 0;
        bool ok = ref_start > -(1 << 30) && ref_start < (1 << 30);
        // leading pad absorbs the shift first (:68-83)
        const int shift_i = (int)rfl64(ri.shift);
        const int raw = ref_start < 0 ? -ref_start : 0;
        const int shifted0 = shift_i < raw ? shift_i : raw;
        const int n_lead = (raw - shifted0 < L) ? raw - shifted0 : L;
        int rem = shift_i - shifted0;                  // shift still open
        int ref_idx0 = ref_start < 0 ? 0 : ref_start;  // ref_idx before the first applied variant
        int pm_carry = ref_idx0;                       // max(ref_idx0, v_end of every applied variant so far)
        int x_carry = 0;                               // output bases produced by the applied variants so far
        bool ended = false;                            // the ">= L" break has happened
        bool have_ns = false; int ns_end = 0, ns_E = 0;   // last applied indel: allele end (out), v_end (ref)
        int ref_idx_end = ref_idx0, out_idx_end = n_lead; // walk state after the last applied variant
        bool any_applied = false;
        const int lead_kept = (n_lead > 0 && n_lead > lo_clip && 0 < hi_clip) ? 1 : 0;
        int n_ent = lead_kept;
        bool past_chunk = false;                       // every later variant lands at or after hi_clip
        auto enc = [](u32 kind, i64 delta, u32 &lo, u32 &hi) {
            const u64 e = (u64)(delta + DELTA_BIAS) | ((u64)kind << 62);
            lo = (u32)e; hi = (u32)(e >> 32);
        };
        auto put = [&](int q, u32 kind, int o_start, i64 delta, int id, int vpos) {
            if (q < SEG_CAP) {
                u32 lo, hi; enc(kind, delta, lo, hi);
                pl.s_out[q] = o_start; pl.s_lo[q] = lo; pl.s_hi[q] = hi;
                if (ANNOT) { pl.s_a[q] = id; pl.s_b[q] = vpos; }
            }
        };

        for (int tb = 0; tb < n_var && ok && !ended && !past_chunk; tb += WAVE) {
            int pos = 0, d = 0, alen = 0, inl = 0, vi = 0; i64 a0 = 0;
            bool valid = tb + lane < n_var;
            if (valid) {
                if (A.grec) {
                    // the variant's fields sit next to the CSR entry: one contiguous read, and
                    // only lanes that will need allele bytes from memory go on to alt_offsets
                    const i32x4 rec = *reinterpret_cast<const i32x4 *>(A.grec + (o_s + tb + lane));
                    pos = rec.x; d = rec.y; alen = (int)((u32)rec.z >> 8); inl = rec.z & 0xFF; vi = rec.w;
                    if (!(d == 0 && alen == 1)) a0 = A.alt_offsets[vi];
                } else {
                    int v = A.geno_v_idxs[o_s + tb + lane];
                    v = v < 0 ? 0 : ((i64)v >= A.n_variants ? (int)(A.n_variants - 1) : v);
                    const i32x4 rec = *reinterpret_cast<const i32x4 *>(A.vrec + v);
                    a0 = A.alt_offsets[v];
                    pos = rec.x; d = rec.y; alen = rec.z; inl = rec.w; vi = v;
                }
                if (has_keep) valid = A.keep[keep_off + tb + lane] != 0;
            }
            // coordinates beyond 2^30 (or nonsense) go to the scalar path: everything below is i32
            // (grec clips alen to 2^24 - 1: such an allele is "weird" here and exact there)
            const bool weird = valid && (pos < 0 || pos >= (1 << 30) || d <= -(1 << 30) || d >= (1 << 30) ||
                                         alen < 0 || alen >= (A.grec ? 0xFFFFFF : (1 << 30)));
            ok = ok && __builtin_amdgcn_ballot_w64(weird) == 0;
            const int E = pos - (d < 0 ? d : 0) + 1;                        // v_ref_end, :96
            const bool is_snp = d == 0 && alen == 1;
            // DEL spanning the window start (:99-102): the last one in order sets ref_idx.  Such
            // variants precede every candidate (sorted by position), so nothing is applied yet.
            // a row with an indel has the longest plan and the slowest trips; it decides when the
            // launch ends, so its wave wins the issue arbitration against waves already streaming
            // (cfg3: -0.4..-0.7 us per launch, nothing lost with several batches in flight)
            if (__builtin_amdgcn_ballot_w64(valid && !is_snp)) __builtin_amdgcn_s_setprio(3);
            const u64 m_span = __builtin_amdgcn_ballot_w64(valid && pos < ref_start && d < 0 && E >= ref_start);
            if (m_span) { ref_idx0 = rdl(E, 63 - __builtin_clzll(m_span)); pm_carry = ref_idx0; ref_idx_end = ref_idx0; }
            bool cand = valid && pos >= ref_start;
            // shift consumption (:115-146).  While the shift is open ref_idx stays put, so the
            // variants in front of the one that completes it are simply dropped; that one (lane f)
            // either starts after the shifted origin, or loses the first `skip` bytes of its
            // allele, or is consumed entirely.
            if (rem > 0) {
                const int base = ref_idx0;
                const u64 m_t = __builtin_amdgcn_ballot_w64(cand && pos >= base && (pos - base) + alen >= rem);
                if (m_t == 0) {
                    cand = false;                      // still open after this trip
                } else {
                    const int f = __builtin_ctzll(m_t);
                    const int dist = rdl(pos, f) - base;
                    if (dist >= rem) {
                        ref_idx0 = base + rem;
                        cand = cand && lane >= f;
                    } else {
                        const int skip = rem - dist;
                        if (skip == rdl(alen, f)) {
                            ref_idx0 = rdl(E, f);
                            cand = cand && lane > f;
                        } else {
                            ref_idx0 = rdl(pos, f);
                            cand = cand && lane >= f;
                            if (lane == f) { alen -= skip; a0 += skip; }
                        }
                    }
                    rem = 0;
                    pm_carry = ref_idx0; ref_idx_end = ref_idx0;
                }
            }
            // first ALT wins (:108-110): fixed point of B = {i : pos_i >= max(carry, max E over B before i)}
            bool inB = cand;
            int PM = 0, pm_incl = 0;
            {
                u64 mB = __builtin_amdgcn_ballot_w64(inB);
                bool stable = false;
#pragma unroll 1
                for (int it = 0; it < 4 && !stable; ++it) {
                    PM = wave_scan_exclusive<OpMaxU>(inB ? E : 0, pm_incl);
                    PM = PM > pm_carry ? PM : pm_carry;
                    inB = cand && pos >= PM;
                    const u64 m2 = __builtin_amdgcn_ballot_w64(inB);
                    stable = m2 == mB;
                    mB = m2;
                }
                ok = ok && stable;
            }
            // output offsets: exclusive prefix sum of (reference run + allele)      (the indel shift)
            const int n_i = inB ? pos - PM : 0;
            const int S_i = inB ? OpSat::f(n_i, alen) : 0;
            int x_incl;
            const int X = OpSat::f(x_carry, wave_scan_exclusive<OpSat>(S_i, x_incl));
            const int allele_out = OpSat::f(OpSat::f(n_lead, X), n_i);
            const bool applied = inB && allele_out < L;                       // :154-158 break
            const int w_i = applied ? ((alen < L - allele_out) ? alen : L - allele_out) : 0;   // :178
            const bool nonsnp = applied && !is_snp;
            const bool snp = applied && is_snp;
            const u64 m_inB = __builtin_amdgcn_ballot_w64(inB);
            const u64 m_app = __builtin_amdgcn_ballot_w64(applied);
            if (m_inB != m_app) ended = true;
            if (m_app) {
                const int last = 63 - __builtin_clzll(m_app);
                ref_idx_end = rdl(E, last);
                out_idx_end = rdl(allele_out, last) + rdl(w_i, last);
                any_applied = true;
                if (rdl(allele_out, last) >= hi_clip) past_chunk = true;
            }
            // the reference run in front of each applied indel starts after the previous applied indel
            const int prevNS = wave_scan_exclusive<OpMaxI>(nonsnp ? lane : -1);
            const int pidx = prevNS < 0 ? 0 : prevNS;
            const int p_end = bperm(pidx, allele_out + alen);                 // not truncated: it has a successor
            const int p_E = bperm(pidx, E);
            const int run_start = prevNS < 0 ? (have_ns ? ns_end : n_lead) : p_end;
            const i64 run_src = c_s + (prevNS < 0 ? (have_ns ? ns_E : ref_idx0) : p_E);
            const bool e_ref = nonsnp && allele_out > run_start && allele_out > lo_clip && run_start < hi_clip;
            const bool e_all = nonsnp && w_i > 0 && allele_out + w_i > lo_clip && allele_out < hi_clip;
            const int slot0 = n_ent + wave_scan_exclusive<OpAdd>((e_ref ? 1 : 0) + (e_all ? 1 : 0));
            const u64 m_ns = __builtin_amdgcn_ballot_w64(nonsnp);
            const u64 m_snp = __builtin_amdgcn_ballot_w64(snp && allele_out >= lo_clip && allele_out < hi_clip);
            const int add_ent = __builtin_popcountll(__builtin_amdgcn_ballot_w64(e_ref)) +
                                __builtin_popcountll(__builtin_amdgcn_ballot_w64(e_all));
            const int add_pat = __builtin_popcountll(m_snp);
            if (n_ent + add_ent + 2 > SEG_CAP || npatch + add_pat > WAVE) ok = false;
            if (ok) {
                int q = slot0;
                if (e_ref) { put(q, K_REF, run_start, run_src - run_start, -1, -1); ++q; }
                if (e_all) put(q, K_ALLELE, allele_out, a0 - allele_out, vi, pos);
                if ((m_snp >> lane) & 1ull) {
                    const int ps = npatch + __builtin_popcountll(m_snp & ((1ull << lane) - 1ull));
                    pl.p_out[ps] = allele_out; pl.p_val[ps] = inl & 0xFF;
                    if (ANNOT) pl.p_id[ps] = vi;
                }
            }
            // carries
            if (tb + WAVE < n_var) {
                // pm_incl is the scan of the last fixed-point round, whose input mask equals
                // the final one (that is what "stable" means)
                const int mx = rdl(pm_incl, 63);
                pm_carry = mx > pm_carry ? mx : pm_carry;
                x_carry = OpSat::f(x_carry, rdl(x_incl, 63));
            }
            if (m_ns) {
                const int last = 63 - __builtin_clzll(m_ns);
                ns_end = rdl(allele_out, last) + rdl(alen, last);
                ns_E = rdl(E, last);
                have_ns = true;
            }
            n_ent += add_ent;
            npatch += add_pat;
        }
        if (ok) {
            // the run after the last applied indel (through any SNPs), then -- if the walk ran to
            // its end -- the rest of the contig and the right pad (:200-255)
            if (rem > 0) ref_idx_end = (int)imin((i64)ref_idx0 + rem, R);     // shift never completed
            const int t_start = have_ns ? ns_end : n_lead;
            const i64 t_src = c_s + (have_ns ? ns_E : ((rem > 0 && !any_applied) ? ref_idx_end : ref_idx0));
            int t_end = out_idx_end;
            bool tail_pad = false;
            if (past_chunk && !ended) {
                t_end = hi_clip;                        // open run: covers the rest of this chunk
            } else {
                const int u = L - out_idx_end;
                if (u > 0) {
                    const i64 avail = R - ref_idx_end;
                    const int w = (int)imin((i64)u, avail);
                    if (w > 0) t_end = out_idx_end + w;
                }
                tail_pad = t_end < L && L > lo_clip && t_end < hi_clip;
            }
            const bool tail_ref = t_end > t_start && t_end > lo_clip && t_start < hi_clip;
            if (lane == 0) {
                if (lead_kept) put(0, K_PAD_LEAD, 0, 0, -1, -1);
                int q = n_ent;
                if (tail_ref) { put(q, K_REF, t_start, t_src - t_start, -1, -1); ++q; }
                if (tail_pad) put(q, K_PAD_TRAIL, t_end, 0, -1, -1);
            }
            nseg = n_ent + (tail_ref ? 1 : 0) + (tail_pad ? 1 : 0);
            // sentinels: P3b counts starts <= p0 over the first 8 slots without bounds checks
            if (lane < 8) {
                if (lane >= nseg) pl.s_out[lane] = 0x7FFFFFFF;
                if (lane >= npatch) pl.p_out[lane] = 0x7FFFFFFF;
            }
        } else {
            flags |= 2;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (flags & 1) return;
    if (!SOLO && packed) {
        if (rfl(meta[wave].bad)) flags |= 2;
        nseg = rfl(meta[wave].nseg); npatch = rfl(meta[wave].npatch);
    }
    GVL_STAMP(4);
#ifdef GVL_DIAG
    if (A.stamps && lane == 0 && (flags & 2)) atomicAdd((unsigned long long *)&A.stamps[((u64)blockIdx.x * gridDim.y + blockIdx.y) * 16 + 9], 1ull);
#endif
    if (flags & 2) {
        recon_wave_scalar<OH, HAPS, ANNOT>(A, luts, SOLO ? solo->u.M : mirror[wave], G, k, chunk, lane);
        return;
    }

    // ---- P3b: trip descriptors, lane u = trip u of this row -------------------------------
    // A trip (256 bases) is described by the <= 3 table entries that cover it (class 0..2 =
    // 1..3 entries; the typical indel trip is REF | ALLELE | REF); anything else (4+ entries,
    // a source that would be read past an array end, the partial last group, zero fill) is
    // class 3 = general.  Entry j spans [b_j, b_{j+1}) with b_0 <= p0.
    const bool ref_zero_fill = (flags & 4) != 0;
    const int limit = hi_clip;
    int d_cls = 3, d_b1 = 0, d_b2 = 0, d_pc0 = 0, d_pcn = 0, d_idx = 0;
    u32 d_ldlo = 0, d_ldhi = 0;     // class 0: address of the trip's first source byte
    u32 d_lo0 = 0, d_hi0 = 0, d_lo1 = 0, d_hi1 = 0, d_lo2 = 0, d_hi2 = 0;
    if (!SOLO && packed) {
        if (lane < CHUNK_TRIPS) {
            const TripDesc &D = desc[wave];
            d_cls = D.cls[lane]; d_b1 = D.b1[lane]; d_b2 = D.b2[lane]; d_pc0 = D.pc0[lane]; d_pcn = D.pcn[lane];
            d_idx = D.idx[lane]; d_ldlo = D.ldlo[lane]; d_ldhi = D.ldhi[lane];
            d_lo0 = D.lo0[lane]; d_hi0 = D.hi0[lane]; d_lo1 = D.lo1[lane]; d_hi1 = D.hi1[lane];
            d_lo2 = D.lo2[lane]; d_hi2 = D.hi2[lane];
        }
    } else if (fast_row) {
        // one reference run (+ pads at a contig edge): a trip inside the run is class 0 with a known
        // address, any other trip is general
        if (lane < CHUNK_TRIPS) {
            const int p0 = lo_clip + lane * TRIP;
            if (p0 < limit) {
                const int t_end = (limit - p0 > TRIP) ? p0 + TRIP : limit;
                d_idx = fp_lead_kept + ((fp_ref && fp_n_lead <= p0) ? 1 : 0) + ((fp_pad && fp_t_end <= p0) ? 1 : 0) - 1;
                if (d_idx < 0) d_idx = 0;
                d_b1 = limit; d_b2 = limit;
                const bool inside = fp_ref && p0 >= fp_n_lead && t_end <= fp_t_end && fp_delta + p0 >= 0 &&
                                    fp_delta + t_end <= A.ref_len && ((t_end - p0) & 3) == 0 && !ref_zero_fill;
                if (inside) {
                    d_cls = 0;
                    const u64 ad = (u64)A.ref + (u64)(fp_delta + p0);
                    d_ldlo = (u32)ad; d_ldhi = (u32)(ad >> 32);
                    if (ANNOT) {
                        const u64 e = (u64)(fp_delta + DELTA_BIAS) | ((u64)K_REF << 62);
                        d_lo0 = (u32)e; d_hi0 = (u32)(e >> 32);
                    }
                }
                if (npatch > 0) {
                    int po[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) po[j] = pl.p_out[j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        d_pc0 += (po[j] < p0) ? 1 : 0;
                        d_pcn += (po[j] < t_end) ? 1 : 0;
                    }
                }
            }
        }
    } else if (lane < CHUNK_TRIPS) {
        const int p0 = lo_clip + lane * TRIP;
        if (p0 < limit) {
            const int t_end = (limit - p0 > TRIP) ? p0 + TRIP : limit;
            // entry holding p0 = (#entries starting at or before p0) - 1.  The first 8 starts are
            // read in one go (independent LDS reads); longer tables continue with a loop.
            int idx = 0;
            {
                int so[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) so[j] = pl.s_out[j];
#pragma unroll
                for (int j = 1; j < 8; ++j) idx += (so[j] <= p0) ? 1 : 0;
                for (int s2 = 8; s2 < nseg; ++s2) idx += (pl.s_out[s2] <= p0) ? 1 : 0;
            }
            auto at = [&](int i) { return i < SEG_CAP ? i : SEG_CAP - 1; };
            const int b1 = idx + 1 < nseg ? pl.s_out[at(idx + 1)] : limit;
            const int b2 = idx + 2 < nseg ? pl.s_out[at(idx + 2)] : limit;
            const int b3 = idx + 3 < nseg ? pl.s_out[at(idx + 3)] : limit;
            d_lo0 = pl.s_lo[at(idx)]; d_hi0 = pl.s_hi[at(idx)];
            d_lo1 = pl.s_lo[at(idx + 1)]; d_hi1 = pl.s_hi[at(idx + 1)];
            d_lo2 = pl.s_lo[at(idx + 2)]; d_hi2 = pl.s_hi[at(idx + 2)];
            int cls = b1 >= t_end ? 0 : (b2 >= t_end ? 1 : (b3 >= t_end ? 2 : 3));
            // every group that overlaps entry j loads the dword at delta_j + p, i.e. up to 3
            // bytes before/after the entry's own span: those must stay inside the array
            auto in_bounds = [&](u32 lo, u32 hi, int s, int e) {
                const u32 kind = hi >> 30;
                if (kind != K_REF && kind != K_ALLELE) return true;
                const i64 dl = seg_delta(lo, hi);
                const i64 len = kind == K_REF ? A.ref_len : A.alt_len;
                const int s3 = (s - 3 > p0 ? s - 3 : p0), e3 = (e + 3 < t_end ? e + 3 : t_end);
                return dl + s3 >= 0 && dl + e3 <= len;
            };
            bool okb = in_bounds(d_lo0, d_hi0, p0, b1 < t_end ? b1 : t_end);
            if (cls >= 1 && cls < 3) okb = okb && in_bounds(d_lo1, d_hi1, b1, b2 < t_end ? b2 : t_end);
            if (cls == 2) okb = okb && in_bounds(d_lo2, d_hi2, b2, t_end);
            if (!okb || ((t_end - p0) & 3) != 0 || ref_zero_fill) cls = 3;
            if (cls == 0) {
                const u32 kind = d_hi0 >> 30;
                if (kind == K_REF || kind == K_ALLELE) {
                    const u64 ad = (u64)(kind == K_REF ? A.ref : A.alt_alleles) + (u64)(seg_delta(d_lo0, d_hi0) + p0);
                    d_ldlo = (u32)ad; d_ldhi = (u32)(ad >> 32);
                } else {
                    cls = 3;    // a trip of pure padding: rare, the general path writes it
                }
            }
            {
                int po[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) po[j] = pl.p_out[j];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    d_pc0 += (po[j] < p0) ? 1 : 0;
                    d_pcn += (po[j] < t_end) ? 1 : 0;
                }
                for (int q = 8; q < npatch; ++q) {
                    const int pp = pl.p_out[q];
                    d_pc0 += pp < p0 ? 1 : 0;
                    d_pcn += pp < t_end ? 1 : 0;
                }
            }
            d_cls = cls; d_b1 = b1; d_b2 = b2; d_idx = idx;
        }
    }
    GVL_STAMP(5);
#ifdef GVL_DIAG
    if (A.stamps && lane == 0) {   // every wave: latest / earliest "plan ready"
        unsigned long long *b = (unsigned long long *)&A.stamps[((u64)blockIdx.x * gridDim.y + blockIdx.y) * 16];
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        atomicMax(b + 13, t);
        atomicMin(b + 14, t);
    }
#endif

    // ---- P4: stream ----------------------------------------------------------------------
    const bool rc = rfl(ri.rc) != 0;
    const i64 row_base = rfl64(ri.row_base);
    const u32 padb = A.pad & 0xFFu;
    const u32 rc_sel = rc ? 0x00010203u : 0x03020100u;
    const u32 *oh_t = rc ? luts.oh_rc : luts.oh;
    const int lane_pos = rc ? -GROUP * lane : GROUP * lane;
    u8 *hap_row = HAPS ? A.haps + row_base : nullptr;
    u8 *oh_row = OH != OH_NONE ? A.onehot + 4 * row_base : nullptr;
    int *av_row = (ANNOT && A.av) ? A.av + row_base : nullptr;
    int *ap_row = (ANNOT && A.ap) ? A.ap + row_base : nullptr;

    // finish one trip: SNP patches, reverse-complement, one-hot LUT, stores
    auto finish = [&](const int p0, const int pc0, const int pcn, u32 wv, int (&av4)[GROUP], int (&ap4)[GROUP]) -> u32 {
        const int p = p0 + GROUP * lane;
        const bool full = p + GROUP <= limit;
        for (int q = pc0; q < pcn; ++q) {
            const u32 dd = (u32)(pl.p_out[q] - p);
            if (dd < (u32)GROUP) {
                const u32 sh = dd * 8;
                wv = (wv & ~(0xFFu << sh)) | ((u32)pl.p_val[q] << sh);
            }
            if (ANNOT) {
                const int pid = pl.p_id[q];
#pragma unroll
                for (int i = 0; i < GROUP; ++i) if (dd == (u32)i) av4[i] = pid;
            }
        }
        if (full && !(A.dbg & 2)) {
            const int jo = (rc ? L - GROUP - p0 : p0) + lane_pos;
            const u32 ww = __builtin_amdgcn_perm(0u, wv, rc_sel);
            const u32 b0_ = ww & 0xFF, b1_ = (ww >> 8) & 0xFF, b2_ = (ww >> 16) & 0xFF, b3_ = ww >> 24;
            if (OH == OH_LC) {
                u32x4_a4 o = {oh_t[b0_], oh_t[b1_], oh_t[b2_], oh_t[b3_]};
                store_oh16(oh_row + 4 * (i64)jo, o);
            } else if (OH == OH_CL) {
                const u32 d0 = oh_t[b0_], d1 = oh_t[b1_], d2 = oh_t[b2_], d3 = oh_t[b3_];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const u32 sh = 8 * a;
                    const u32 v = ((d0 >> sh) & 0xFF) | (((d1 >> sh) & 0xFF) << 8) |
                                  (((d2 >> sh) & 0xFF) << 16) | (((d3 >> sh) & 0xFF) << 24);
                    store_u32_unaligned(oh_row + (i64)a * L + jo, v);
                }
            }
            if (HAPS) {
                u32 hv = ww;
                if (rc) hv = luts.comp[b0_] | (luts.comp[b1_] << 8) | (luts.comp[b2_] << 16) | (luts.comp[b3_] << 24);
                store_u32_unaligned(hap_row + jo, hv);
            }
            if (ANNOT) {
                if (av_row) {
                    i32x4_a4 o = rc ? i32x4_a4{av4[3], av4[2], av4[1], av4[0]} : i32x4_a4{av4[0], av4[1], av4[2], av4[3]};
                    store_i32x4(av_row + jo, o.x, o.y, o.z, o.w);
                }
                if (ap_row) {
                    i32x4_a4 o = rc ? i32x4_a4{ap4[3], ap4[2], ap4[1], ap4[0]} : i32x4_a4{ap4[0], ap4[1], ap4[2], ap4[3]};
                    store_i32x4(ap_row + jo, o.x, o.y, o.z, o.w);
                }
            }
        }
        return wv;
    };

    // the partial group at the row end (L % 4 != 0) -- only class-3 trips can hold it
    auto finish_partial = [&](const int p0, const u32 wv, int (&av4)[GROUP], int (&ap4)[GROUP]) {
        const int p = p0 + GROUP * lane;
        const bool act = p < limit;
        const bool full = p + GROUP <= limit;
        if (limit & 3) {
            // partial group at the row end (L % 4 != 0): per-base stores by lanes 0..(L&3)-1
            const int p_last = limit & ~3;
            if (p_last >= p0 && p_last < p0 + TRIP) {
                if (act && !full) {
                    G.w[0] = wv;
                    if (ANNOT) {
#pragma unroll
                        for (int i = 0; i < GROUP; ++i) { G.av[i][0] = av4[i]; G.ap[i][0] = ap4[i]; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < (limit & 3)) {
                    const int pp = p_last + lane;
                    const u32 b = (G.w[0] >> (8 * lane)) & 0xFF;
                    const i64 jo = rc ? (i64)(L - 1 - pp) : (i64)pp;
                    if (OH != OH_NONE) {
                        const u32 dd = oh_t[b];
                        if (OH == OH_LC) {
                            __builtin_memcpy(oh_row + 4 * jo, &dd, 4);
                        } else {
#pragma unroll
                            for (int a = 0; a < 4; ++a) oh_row[(i64)a * L + jo] = (u8)((dd >> (8 * a)) & 0xFF);
                        }
                    }
                    if (HAPS) hap_row[jo] = (u8)(rc ? luts.comp[b] : b);
                    if (ANNOT) {
                        if (av_row) av_row[jo] = G.av[lane][0];
                        if (ap_row) ap_row[jo] = G.ap[lane][0];
                    }
                }
            }
        }
    };

    // Trip classes: 0 = one entry (uniform), 1/2 = two/three entries ("multi": the typical
    // indel trip), 3 = general.  Loads are issued in the order multi (up to 2 trips), then
    // all class-0 trips, so that everything is in flight before the first wait.
    auto entry_load = [&](const u32 lo, const u32 hi, const int p, const bool pred) -> u32 {
        const u32 kind = hi >> 30;                       // uniform
        if (kind == K_REF || kind == K_ALLELE) {
            const u8 *arr = (kind == K_REF ? A.ref : A.alt_alleles) + seg_delta(lo, hi);
            u32 v = 0;
            if (pred && !(A.dbg & 4)) v = load_u32_unaligned(arr + p);
            return v;
        }
        return padb * 0x01010101u;
    };
    auto bytes_below = [](int k) -> u32 { return k <= 0 ? 0u : (k >= 4 ? 0xFFFFFFFFu : ((1u << (8 * k)) - 1u)); };
    auto multi_issue = [&](const int u, u32 (&w3)[3]) {
        const int p = lo_clip + u * TRIP + GROUP * lane;
        const int cls = rdl(d_cls, u);
        const int b1 = rdl(d_b1, u), b2 = rdl(d_b2, u);
        w3[0] = entry_load((u32)rdl((int)d_lo0, u), (u32)rdl((int)d_hi0, u), p, p < limit && p < b1);
        w3[1] = entry_load((u32)rdl((int)d_lo1, u), (u32)rdl((int)d_hi1, u), p, p < limit && p + GROUP > b1 && p < b2);
        w3[2] = 0;
        if (cls >= 2) w3[2] = entry_load((u32)rdl((int)d_lo2, u), (u32)rdl((int)d_hi2, u), p, p < limit && p + GROUP > b2);
    };
    auto multi_finish = [&](const int u, const u32 (&w3)[3]) {
        const int p0 = lo_clip + u * TRIP;
        const int p = p0 + GROUP * lane;
        const int cls = rdl(d_cls, u);
        const int b1 = rdl(d_b1, u), b2 = rdl(d_b2, u);
        const u32 m1 = bytes_below(b1 - p);           // bytes of entry 0
        const u32 m2 = bytes_below(b2 - p);           // bytes of entries 0 and 1
        const u32 wv = (w3[0] & m1) | (w3[1] & m2 & ~m1) | (w3[2] & ~m2);
        int av4[GROUP], ap4[GROUP];
        if (ANNOT) {
            const u32 lo_[3] = {(u32)rdl((int)d_lo0, u), (u32)rdl((int)d_lo1, u), (u32)rdl((int)d_lo2, u)};
            const u32 hi_[3] = {(u32)rdl((int)d_hi0, u), (u32)rdl((int)d_hi1, u), (u32)rdl((int)d_hi2, u)};
            const int i0 = rdl(d_idx, u);
#pragma unroll
            for (int i = 0; i < GROUP; ++i) {
                const int pp = p + i;
                const int j = pp >= b1 ? ((cls >= 2 && pp >= b2) ? 2 : 1) : 0;
                const u32 lo = j == 0 ? lo_[0] : (j == 1 ? lo_[1] : lo_[2]);
                const u32 hi = j == 0 ? hi_[0] : (j == 1 ? hi_[1] : hi_[2]);
                const u32 kind = hi >> 30;
                const int se = i0 + j < SEG_CAP ? i0 + j : SEG_CAP - 1;
                if (kind == K_REF) { av4[i] = -1; ap4[i] = (int)(seg_delta(lo, hi) + pp - c_s); }
                else if (kind == K_ALLELE) { av4[i] = pl.s_a[se]; ap4[i] = pl.s_b[se]; }
                else { av4[i] = -1; ap4[i] = kind == K_PAD_LEAD ? -1 : 2147483647; }
            }
        }
        finish(p0, rdl(d_pc0, u), rdl(d_pcn, u), wv, av4, ap4);
    };
    u32 gmask, mmask, umask;
    {
        const u64 bg = __builtin_amdgcn_ballot_w64(lane < CHUNK_TRIPS && lo_clip + lane * TRIP < limit && d_cls == 3);
        const u64 bm = __builtin_amdgcn_ballot_w64(lane < CHUNK_TRIPS && lo_clip + lane * TRIP < limit && (d_cls == 1 || d_cls == 2));
        const u64 bu = __builtin_amdgcn_ballot_w64(lane < CHUNK_TRIPS && lo_clip + lane * TRIP < limit && d_cls == 0);
        gmask = (u32)bg; mmask = (u32)bm; umask = (u32)bu;
    }
    // the first two multi trips: loads now, finish after the class-0 loads are out
    int mu_a = -1, mu_b = -1;
    u32 wa[3] = {0, 0, 0}, wb[3] = {0, 0, 0};
    if (mmask) { mu_a = __builtin_ctz(mmask); mmask &= mmask - 1; multi_issue(mu_a, wa); }
    if (mmask) { mu_b = __builtin_ctz(mmask); mmask &= mmask - 1; multi_issue(mu_b, wb); }
    // pass A: class-0 trips, 4 bytes per lane from one scalar base (computed in P3b)
    {
        // trips whose speculative read is the one the plan asks for (lane u holds trip u's descriptor)
        const u64 want = ((u64)d_ldhi << 32) | d_ldlo;
        const u64 spec = (u64)(A.ref + (g_delta + lo_clip)) + (u64)(u32)(lane * TRIP);
        const u32 have = spmask & (u32)__builtin_amdgcn_ballot_w64(lane < CHUNK_TRIPS && d_cls == 0 && want == spec);
        const u32 todo = umask & ~have;
#pragma unroll
        for (int u = 0; u < CHUNK_TRIPS; ++u) {
            if ((todo >> u) & 1u) {
                const u8 *src = reinterpret_cast<const u8 *>(((u64)(u32)rdl((int)d_ldhi, u) << 32) | (u32)rdl((int)d_ldlo, u));
                wq[u] = 0;
                if (lane4 < limit - (lo_clip + u * TRIP) && !(A.dbg & 4)) wq[u] = load_u32_unaligned(src + (u32)lane4);
            }
        }
    }
    if (mu_a >= 0) multi_finish(mu_a, wa);
    if (mu_b >= 0) multi_finish(mu_b, wb);
    while (mmask) {   // a third, fourth ... multi trip in one chunk: one at a time
        const int u = __builtin_ctz(mmask);
        mmask &= mmask - 1;
        u32 w3[3];
        multi_issue(u, w3);
        multi_finish(u, w3);
    }
    GVL_STAMP(6);
#ifdef GVL_DIAG
    if (A.stamps && tid == 0) A.stamps[((u64)blockIdx.x * gridDim.y + blockIdx.y) * 16 + 10] = __builtin_popcount(gmask);
#endif
    // pass G: general trips (a segment boundary / allele / pad / row end inside the trip).
    // A group of 4 bases overlaps at most 4 segments; each contributes one masked dword
    // load from ITS source, so the group costs one memory latency, not one per byte.
    while (gmask) {
        const int u = __builtin_ctz(gmask);
        gmask &= gmask - 1;
        const int p0 = lo_clip + u * TRIP;
        const int p = p0 + GROUP * lane;
        u32 wv = 0;
        int av4[GROUP], ap4[GROUP];
        if (ANNOT) {
#pragma unroll
            for (int i = 0; i < GROUP; ++i) { av4[i] = -1; ap4[i] = -1; }
        }
        if (p < limit) {
            // segment holding p: the trip's first segment + the starts before p (independent
            // LDS reads; a fifth boundary inside one trip falls through to the while loop)
            const int s0 = __builtin_amdgcn_readlane(d_idx, u);
            int seg = s0;
            {
                int c = 0;
#pragma unroll
                for (int j = 1; j <= 4; ++j) {
                    const int sj = s0 + j < SEG_CAP ? s0 + j : SEG_CAP - 1;
                    c += (s0 + j < nseg && pl.s_out[sj] <= p) ? 1 : 0;
                }
                seg += c;
                if (c == 4) while (seg + 1 < nseg && pl.s_out[seg + 1] <= p) ++seg;
            }
            const int g_end = (limit - p > GROUP) ? p + GROUP : limit;
            // table entries seg .. seg+3 (a group of 4 bases overlaps at most 4 segments)
            u32 e_lo[GROUP], e_hi[GROUP]; int e_nx[GROUP];
#pragma unroll
            for (int t = 0; t < GROUP; ++t) {
                const int st = seg + t < SEG_CAP ? seg + t : SEG_CAP - 1;
                const int sn = seg + t + 1 < SEG_CAP ? seg + t + 1 : SEG_CAP - 1;
                e_lo[t] = pl.s_lo[st]; e_hi[t] = pl.s_hi[st];
                e_nx[t] = seg + t + 1 < nseg ? pl.s_out[sn] : limit;
            }
            int cur = p;
            u32 words[GROUP], masks[GROUP];
#pragma unroll
            for (int t = 0; t < GROUP; ++t) {
                words[t] = 0; masks[t] = 0;
                if (cur < g_end) {
                    const u32 lo = e_lo[t], hi = e_hi[t];
                    const u32 kind = hi >> 30;
                    const int s_end = e_nx[t];
                    const int e = s_end < g_end ? s_end : g_end;
                    const u32 m_hi = (e - p) >= 4 ? 0xFFFFFFFFu : ((1u << (8 * (e - p))) - 1u);
                    const u32 m_lo = (1u << (8 * (cur - p))) - 1u;
                    masks[t] = m_hi & ~m_lo;
                    const i64 src = seg_delta(lo, hi) + p;
                    if (ref_zero_fill) {
                        words[t] = 0;
                    } else if (kind == K_REF || kind == K_ALLELE) {
                        const u8 *arr = kind == K_REF ? A.ref : A.alt_alleles;
                        const i64 alen_ = kind == K_REF ? A.ref_len : A.alt_len;
                        if (src >= 0 && src + GROUP <= alen_) {
                            words[t] = load_u32_unaligned(arr + src);
                        } else {   // array edge: byte by byte, out of range -> pad
                            u32 wb = 0;
#pragma unroll
                            for (int i = 0; i < GROUP; ++i) {
                                const i64 s2 = src + i;
                                const u32 bb = (p + i >= cur && p + i < e && s2 >= 0 && s2 < alen_) ? (u32)arr[s2] : padb;
                                wb |= bb << (8 * i);
                            }
                            words[t] = wb;
                        }
                    } else {
                        words[t] = padb * 0x01010101u;
                    }
                    if (ANNOT) {
                        const int st = seg + t < SEG_CAP ? seg + t : SEG_CAP - 1;
#pragma unroll
                        for (int i = 0; i < GROUP; ++i) {
                            if (p + i >= cur && p + i < e) {
                                if (kind == K_REF) { av4[i] = -1; ap4[i] = (int)(src - c_s) + i; }
                                else if (kind == K_ALLELE) { av4[i] = pl.s_a[st]; ap4[i] = pl.s_b[st]; }
                                else { av4[i] = -1; ap4[i] = kind == K_PAD_LEAD ? -1 : 2147483647; }
                            }
                        }
                    }
                    cur = e;
                }
            }
#pragma unroll
            for (int t = 0; t < GROUP; ++t) wv |= words[t] & masks[t];
        }
        wv = finish(p0, rdl(d_pc0, u), rdl(d_pcn, u), wv, av4, ap4);
        finish_partial(p0, wv, av4, ap4);
    }
    GVL_STAMP(7);
    __builtin_amdgcn_s_setprio(0);
    // pass B: finish the class-0 trips.  Store addresses are a scalar base per trip plus a
    // per-lane offset that never changes: forward rows put lane l at +16*l, reverse-complemented
    // rows mirror the index (lane l at +16*(63-l) from the trip's lowest address).
    {
        const int jo_base = rc ? (L - GROUP - lo_clip - GROUP * (WAVE - 1)) : lo_clip;
        const int jo_step = rc ? -TRIP : TRIP;
        const u32 lane_rev = rc ? (u32)(WAVE - 1 - lane) : (u32)lane;
#pragma unroll
        for (int u = 0; u < CHUNK_TRIPS; ++u) {
            if ((umask >> u) & 1u) {
                const int p0 = lo_clip + u * TRIP;
                const int pc0 = rdl(d_pc0, u), pcn = rdl(d_pcn, u);
                u32 wv = wq[u];
                int av4[GROUP], ap4[GROUP];
                if (ANNOT) {
                    const u32 lo = (u32)rdl((int)d_lo0, u), hi = (u32)rdl((int)d_hi0, u);
                    const u32 kind = hi >> 30;
                    const int se = rdl(d_idx, u);
#pragma unroll
                    for (int i = 0; i < GROUP; ++i) {
                        if (kind == K_REF) { av4[i] = -1; ap4[i] = (int)(seg_delta(lo, hi) + p0 + lane4 + i - c_s); }
                        else { av4[i] = pl.s_a[se]; ap4[i] = pl.s_b[se]; }
                    }
                }
                for (int q = pc0; q < pcn; ++q) {
                    const u32 dd = (u32)(pl.p_out[q] - p0 - lane4);
                    if (dd < (u32)GROUP) {
                        const u32 sh = dd * 8;
                        wv = (wv & ~(0xFFu << sh)) | ((u32)pl.p_val[q] << sh);
                    }
                    if (ANNOT) {
                        const int pid = pl.p_id[q];
#pragma unroll
                        for (int i = 0; i < GROUP; ++i) if (dd == (u32)i) av4[i] = pid;
                    }
                }
                if (lane4 < limit - p0 && !(A.dbg & 2)) {
                    const i64 jo0 = (i64)(jo_base + u * jo_step);       // scalar
                    const u32 ww = __builtin_amdgcn_perm(0u, wv, rc_sel);
                    const u32 b0_ = ww & 0xFF, b1_ = (ww >> 8) & 0xFF, b2_ = (ww >> 16) & 0xFF, b3_ = ww >> 24;
                    if (OH == OH_LC) {
                        u32x4_a4 o = {oh_t[b0_], oh_t[b1_], oh_t[b2_], oh_t[b3_]};
                        store_oh16(oh_row + 4 * jo0 + 16u * lane_rev, o);
                    } else if (OH == OH_CL) {
                        const u32 d0 = oh_t[b0_], d1 = oh_t[b1_], d2 = oh_t[b2_], d3 = oh_t[b3_];
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const u32 sh = 8 * a;
                            const u32 v = ((d0 >> sh) & 0xFF) | (((d1 >> sh) & 0xFF) << 8) |
                                          (((d2 >> sh) & 0xFF) << 16) | (((d3 >> sh) & 0xFF) << 24);
                            store_u32_unaligned(oh_row + (i64)a * L + jo0 + 4u * lane_rev, v);
                        }
                    }
                    if (HAPS) {
                        u32 hv = ww;
                        if (rc) hv = luts.comp[b0_] | (luts.comp[b1_] << 8) | (luts.comp[b2_] << 16) | (luts.comp[b3_] << 24);
                        store_u32_unaligned(hap_row + jo0 + 4u * lane_rev, hv);
                    }
                    if (ANNOT) {
                        if (av_row) {
                            i32x4_a4 o = rc ? i32x4_a4{av4[3], av4[2], av4[1], av4[0]} : i32x4_a4{av4[0], av4[1], av4[2], av4[3]};
                            store_i32x4(av_row + jo0 + 4u * lane_rev, o.x, o.y, o.z, o.w);
                        }
                        if (ap_row) {
                            i32x4_a4 o = rc ? i32x4_a4{ap4[3], ap4[2], ap4[1], ap4[0]} : i32x4_a4{ap4[0], ap4[1], ap4[2], ap4[3]};
                            store_i32x4(ap_row + jo0 + 4u * lane_rev, o.x, o.y, o.z, o.w);
                        }
                    }
                }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    GVL_STAMP(8);
#ifdef GVL_DIAG
    if (A.stamps && lane == 0) {   // every wave: latest / earliest end of the workgroup
        unsigned long long *b = (unsigned long long *)&A.stamps[((u64)blockIdx.x * gridDim.y + blockIdx.y) * 16];
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        atomicMax(b + 11, t);
        atomicMin(b + 12, t);
    }
#endif
}

#undef GVL_STAMP

template <int OH, bool HAPS, bool ANNOT>
__global__ __launch_bounds__(WG_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void reconstruct_kernel(const ReconArgs A) {
    __shared__ ReconShared<ANNOT> S;
    recon_body<OH, HAPS, ANNOT, false>(A, &S, nullptr, S.luts, (i64)blockIdx.x, (int)blockIdx.y);
}

typedef void (*recon_fn)(const ReconArgs);
static recon_fn recon_table(int oh, bool haps, bool annot) {
#define GVL_K(o, h, a) reconstruct_kernel<o, h, a>
    if (annot) {
        if (oh == OH_NONE) return GVL_K(OH_NONE, true, true);
        if (oh == OH_LC) return haps ? GVL_K(OH_LC, true, true) : GVL_K(OH_LC, false, true);
        return haps ? GVL_K(OH_CL, true, true) : GVL_K(OH_CL, false, true);
    }
    if (oh == OH_NONE) return GVL_K(OH_NONE, true, false);
    if (oh == OH_LC) return haps ? GVL_K(OH_LC, true, false) : GVL_K(OH_LC, false, false);
    return haps ? GVL_K(OH_CL, true, false) : GVL_K(OH_CL, false, false);
#undef GVL_K
}

#include "gvl_lean.inc"
#include "gvl_lean_pipe.inc"

// ---------------------------------------------------------------------------
// get_diffs_sparse (genotypes/mod.rs:15-125): one lane per (query, hap) row.
// ---------------------------------------------------------------------------
struct DiffArgs {
    const i64 *geno_offset_idx; i64 n_rows; int ploidy;
    const int *geno_v_idxs; const i64 *go_starts; const i64 *go_stops;
    const int *ilens; const int *v_starts; i64 n_variants;
    const gvl_grec *grec;       // nullable: (pos, ilen) next to the CSR entry, one read instead of three
    const u8 *keep; const i64 *keep_offsets;
    const int *q_starts; const int *q_ends; i64 q_stride;
    int *diffs;
    // fused sizing (ffi/mod.rs:794-811)
    i64 output_length; i64 *lengths;
    // the native loop's ragged batches live in slots of a fixed capacity: a row longer than len_cap (> 0) is
    // cut to it and reported through *async_err (never silently)
    i64 len_cap; int *async_err;
};

// length delta of one haplotype (genotype slot o_idx; keep slice at ks; optional query window)
__device__ __forceinline__ i64 row_diff_core(const DiffArgs &A, const i64 o_idx, const bool has_keep, const i64 ks,
                                             const bool has_query, const i64 q_start, const i64 q_end) {
    const i64 o_s = A.go_starts[o_idx], o_e = A.go_stops[o_idx];
    i64 acc = 0;
    if (o_e - o_s <= 0) return 0;
    if (has_query) {                                               // mod.rs:48-85
        i64 ref_idx = q_start;
        for (i64 v = o_s; v < o_e; ++v) {
            if (has_keep && !A.keep[ks + (v - o_s)]) continue;
            i64 vs, il;
            if (A.grec) {
                const int2 r = *reinterpret_cast<const int2 *>(A.grec + v);
                vs = r.x; il = r.y;
            } else {
                const i64 vi = A.geno_v_idxs[v];
                vs = A.v_starts[vi];
                il = A.ilens[vi];
            }
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
            acc += A.grec ? (i64)A.grec[v].ilen : (i64)A.ilens[A.geno_v_idxs[v]];
        }
    }
    return acc;
}

// The same, one WAVE per haplotype, 64 variants per trip (one coalesced read of the records).  The
// walk is sequential only through "skip a variant that starts inside what an applied variant already
// covers" (mod.rs:70): variant i is applied iff NOT (q_start <= pos_i < max(q_start, v_end of every
// applied variant before it)) -- an exclusive prefix-max over the applied set, iterated to its fixed
// point (unique: membership of i depends on the applied set in front of i only; each round settles at
// least one more lane).  The break at the first pos >= q_end is a filter because positions are sorted.
__device__ __forceinline__ i64 wave_excl_prefix_max64(i64 x, const int lane) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const i64 y = __shfl_up(x, o, WAVE);
        if (lane >= o) x = imax(x, y);
    }
    const i64 p = __shfl_up(x, 1, WAVE);
    return lane == 0 ? (i64)(-0x7FFFFFFFFFFFFFFFll - 1) : p;
}
__device__ __forceinline__ i64 row_diff_wave(const DiffArgs &A, const i64 o_idx, const bool has_keep, const i64 ks,
                                              const bool has_query, const i64 q_start, const i64 q_end, const int lane) {
    const i64 o_s = rfl64(A.go_starts[o_idx]), o_e = rfl64(A.go_stops[o_idx]);
    i64 acc = 0, carry = q_start;
    for (i64 b = o_s; b < o_e; b += WAVE) {
        const i64 v = b + lane;
        i64 vs = 0, il = 0;
        bool valid = v < o_e;
        if (valid) {
            if (A.grec) {
                const int2 r = *reinterpret_cast<const int2 *>(A.grec + v);
                vs = r.x; il = r.y;
            } else {
                const i64 vi = A.geno_v_idxs[v];
                il = A.ilens[vi];
                if (has_query) vs = A.v_starts[vi];
            }
            if (has_keep) valid = A.keep[ks + (v - o_s)] != 0;
        }
        i64 x = 0;
        if (!has_query) {                                          // mod.rs:86-103: a (masked) sum
            x = valid ? il : 0;
        } else {                                                   // mod.rs:48-85
            const i64 v_end = vs - imin(il, 0) + 1;
            const bool cand = valid && v_end > q_start && vs < q_end;
            bool inB = cand;
            u64 mB = __builtin_amdgcn_ballot_w64(inB);
            for (int it = 0; it < WAVE + 1; ++it) {
                i64 pm = wave_excl_prefix_max64(inB ? v_end : (i64)(-0x7FFFFFFFFFFFFFFFll - 1), lane);
                pm = imax(pm, carry);
                inB = cand && !(vs >= q_start && vs < pm);
                const u64 m2 = __builtin_amdgcn_ballot_w64(inB);
                if (m2 == mB) break;
                mB = m2;
            }
            if (inB) {
                i64 d = il;
                if (d < 0) d += imax(q_start - vs - 1, 0);
                d += imax(v_end - q_end, 0);
                x = d;
            }
            i64 mx = inB ? v_end : carry;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = imax(mx, __shfl_xor(mx, o, WAVE));
            carry = imax(carry, mx);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, WAVE);
        acc += x;
        if (has_query && __builtin_amdgcn_ballot_w64(v < o_e && vs >= q_end) != 0) break;   // sorted: nothing further counts
    }
    return acc;
}

__device__ __forceinline__ i64 row_diff(const DiffArgs &A, i64 k) {
    const i64 query = k / A.ploidy;
    const bool has_query = A.q_starts && A.q_ends && A.v_starts;   // mod.rs:35
    const bool has_keep = A.keep && A.keep_offsets;                // mod.rs:36
    const i64 ks = has_keep ? A.keep_offsets[k] : 0;
    const i64 q_start = has_query ? (i64)A.q_starts[query * A.q_stride] : 0;
    const i64 q_end = has_query ? (i64)A.q_ends[query * A.q_stride] : 0;
    return row_diff_core(A, A.geno_offset_idx[k], has_keep, ks, has_query, q_start, q_end);
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
        if (A.len_cap > 0 && len > A.len_cap) { len = A.len_cap; if (A.async_err) *A.async_err = 1; }
        A.lengths[k + 1] = len;
        if (k == 0) A.lengths[0] = 0;
    }
}

__global__ __launch_bounds__(256) void diffs_wave_kernel(const DiffArgs A, const int *regions, i64 regions_stride) {
    const int lane = threadIdx.x & (WAVE - 1);
    const i64 k = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (k >= A.n_rows) return;
    const i64 query = k / A.ploidy;
    const bool has_query = A.q_starts && A.q_ends && A.v_starts;
    const bool has_keep = A.keep && A.keep_offsets;
    const i64 ks = has_keep ? rfl64(A.keep_offsets[k]) : 0;
    const i64 q_start = has_query ? (i64)rfl(A.q_starts[query * A.q_stride]) : 0;
    const i64 q_end = has_query ? (i64)rfl(A.q_ends[query * A.q_stride]) : 0;
    const int d = (int)row_diff_wave(A, rfl64(A.geno_offset_idx[k]), has_keep, ks, has_query, q_start, q_end, lane);
    if (lane != 0) return;
    if (A.diffs) A.diffs[k] = d;
    if (A.lengths) {
        i64 len;
        if (A.output_length >= 0) {
            len = A.output_length;
        } else {
            const int *reg = regions + query * regions_stride;
            len = imax((i64)(reg[2] - reg[1]) + d, 0);
        }
        if (A.len_cap > 0 && len > A.len_cap) { len = A.len_cap; if (A.async_err) *A.async_err = 1; }
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

// Scratch-track lengths of a haps + tracks batch (_reconstruct.py:191): per query,
// len - min over haplotypes of min(diff, 0) with diff = the haplotype's length delta inside the
// window (query mode of get_diffs_sparse); written at lengths[q + 1] for the scan.  Also the
// (batch * ploidy + 1) fixed-length output offsets k * L that the realign kernel reads.
__global__ __launch_bounds__(256) void track_lengths_kernel(const DiffArgs A, const int *regions, i64 regions_stride, i64 batch,
                                                             i64 out_len, i64 *lengths, i64 *out_offsets) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 K = batch * A.ploidy;
    for (i64 k = t; k <= K; k += (i64)gridDim.x * blockDim.x) out_offsets[k] = k * out_len;
    const int lane = threadIdx.x & (WAVE - 1);
    const i64 q = t >> 6;                                    // one wave per query
    if (q >= batch) return;
    const int *reg = regions + q * regions_stride;
    const i64 qs = rfl(reg[1]), qe = rfl(reg[2]);
    i64 mn = 0;
    for (int p = 0; p < A.ploidy; ++p) {
        const i64 d = (i64)(int)row_diff_wave(A, rfl64(A.geno_offset_idx[q * A.ploidy + p]), false, 0, true, qs, qe, lane);
        mn = d < mn ? d : mn;
    }
    if (lane == 0) {
        lengths[q + 1] = (qe - qs) - mn;
        if (q == 0) lengths[0] = 0;
    }
}

// The same for every query of an EPOCH (gvl_loader_start_epoch: the jittered regions and the length deltas of
// every query are known there), laid out per batch: batch j's (bs + 1) scan slots at lengths[j * (bs + 1) ..],
// slot 0 = 0; track_scan_batches_kernel then turns each batch's slots into its scratch-track offsets.  A batch
// of the epoch then costs no sizing launch at all (they were 19 us of cfg4's 143 us step).
__global__ __launch_bounds__(256) void track_lengths_epoch_kernel(const DiffArgs A, const int *regions, i64 regions_stride, i64 n,
                                                                   i64 bs, i64 out_len, i64 *lengths, i64 *out_offsets) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 K = bs * A.ploidy;
    for (i64 k = t; k <= K; k += (i64)gridDim.x * blockDim.x) out_offsets[k] = k * out_len;
    const int lane = threadIdx.x & (WAVE - 1);
    const i64 q = t >> 6;                                    // one wave per query
    if (q >= n) return;
    const int *reg = regions + q * regions_stride;
    const i64 qs = rfl(reg[1]), qe = rfl(reg[2]);
    i64 mn = 0;
    for (int p = 0; p < A.ploidy; ++p) {
        const i64 d = (i64)(int)row_diff_wave(A, rfl64(A.geno_offset_idx[q * A.ploidy + p]), false, 0, true, qs, qe, lane);
        mn = d < mn ? d : mn;
    }
    if (lane == 0) {
        const i64 j = q / bs, i = q - j * bs;
        lengths[j * (bs + 1) + i + 1] = (qe - qs) - mn;
        if (i == 0) lengths[j * (bs + 1)] = 0;
    }
}
// one workgroup per batch: inclusive scan of its (bs + 1) slots in place (slot 0 is 0, so slot i = offset of query i)
__global__ __launch_bounds__(256) void track_scan_batches_kernel(i64 *lengths, i64 n, i64 bs) {
    __shared__ i64 wsum[4];
    __shared__ i64 carry_s;
    const i64 j = blockIdx.x;
    const i64 cnt = (n - j * bs < bs ? n - j * bs : bs) + 1;
    i64 *a = lengths + j * (bs + 1);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (i64 base = 0; base < cnt; base += 256) {
        const i64 i = base + tid;
        i64 sc = i < cnt ? a[i] : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const i64 y = __shfl_up(sc, o, 64);
            if (lane >= o) sc += y;
        }
        if (lane == 63) wsum[wv] = sc;
        __syncthreads();
        i64 pre = carry_s;
        for (int w = 0; w < wv; ++w) pre += wsum[w];
        if (i < cnt) a[i] = sc + pre;
        __syncthreads();
        if (tid == 255) carry_s = sc + pre;
        __syncthreads();
    }
}

// The ragged sizing of a whole GROUP of batches in two launches (the native loader's ragged groups; per batch it was two
// latency-bound launches each, 16 us per batch alone -- twice the reconstruct grid's time).  The group's request arrays are
// consecutive rows of the epoch table, its offsets live in the batches' own slots: a pointer per batch in the kernarg segment.
struct HapGroupOut { i64 *offs[GVL_MANY_MAX]; i64 *sizes[GVL_MANY_MAX]; };
__global__ __launch_bounds__(256) void hap_lengths_group_kernel(const DiffArgs A, const int *regions, i64 regions_stride, i64 rows_per_batch,
                                                                 const HapGroupOut O) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A.n_rows) return;
    const int d = (int)row_diff(A, k);      // `as i32` truncation
    const int *reg = regions + (k / A.ploidy) * regions_stride;
    i64 len = imax((i64)(reg[2] - reg[1]) + d, 0);                  // src/ffi/mod.rs:801-807
    if (A.len_cap > 0 && len > A.len_cap) { len = A.len_cap; if (A.async_err) *A.async_err = 1; }
    const i64 b = k / rows_per_batch, i = k - b * rows_per_batch;
    i64 *const offs = O.offs[b];
    offs[i + 1] = len;
    if (i == 0) offs[0] = 0;
}
// one workgroup per batch: the row lengths at offs[1 ..] become offsets in place; sizes = {total, longest row}
__global__ __launch_bounds__(256) void hap_scan_group_kernel(const HapGroupOut O, i64 rows_per_batch, i64 n_rows_total) {
    __shared__ i64 wsum[4];
    __shared__ i64 wmax[4];
    __shared__ i64 carry_s;
    const i64 j = blockIdx.x;
    const i64 n = (n_rows_total - j * rows_per_batch < rows_per_batch) ? n_rows_total - j * rows_per_batch : rows_per_batch;
    i64 *const offs = O.offs[j];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    i64 mx = 0;
    for (i64 base = 0; base < n; base += 256) {
        const i64 i = base + tid;
        const i64 x = i < n ? offs[i + 1] : 0;
        mx = imax(mx, x);
        i64 sc = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const i64 y = __shfl_up(sc, o, 64);
            if (lane >= o) sc += y;
        }
        if (lane == 63) wsum[wv] = sc;
        __syncthreads();
        i64 pre = carry_s;
        for (int w = 0; w < wv; ++w) pre += wsum[w];
        if (i < n) offs[i + 1] = sc + pre;
        __syncthreads();
        if (tid == 255) carry_s = sc + pre;
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = imax(mx, __shfl_down(mx, o, 64));
    if (lane == 0) wmax[wv] = mx;
    __syncthreads();
    if (tid == 0 && O.sizes[j]) {
        O.sizes[j][0] = carry_s;
        O.sizes[j][1] = imax(imax(wmax[0], wmax[1]), imax(wmax[2], wmax[3]));
    }
}

// ---------------------------------------------------------------------------
// choose_exonic_variants (src/genotypes/mod.rs:127-176): keep[v] = the variant lies entirely
// inside its query's [start, end).  Offsets first (counts -> the scan above), then the mask.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void keep_counts_kernel(const i64 *geno_offset_idx, const i64 *go_starts,
                                                          const i64 *go_stops, i64 n_rows, i64 *keep_offsets) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) keep_offsets[0] = 0;
    if (k >= n_rows) return;
    const i64 o = geno_offset_idx[k];
    const i64 n = go_stops[o] - go_starts[o];
    keep_offsets[k + 1] = n > 0 ? n : 0;
}

__global__ __launch_bounds__(256) void exonic_keep_kernel(const int *starts, const int *ends, const i64 *geno_offset_idx,
                                                          i64 n_rows, int ploidy, const int *geno_v_idxs,
                                                          const i64 *go_starts, const i64 *go_stops, const int *v_starts,
                                                          const int *ilens, i64 n_variants, const i64 *keep_offsets, u8 *keep) {
    const int lane = threadIdx.x & (WAVE - 1);
    const i64 k = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;      // one wave per row
    if (k >= n_rows) return;
    const i64 q = k / ploidy;
    const i64 ref_start = starts[q], ref_end = ends[q];
    const i64 o = geno_offset_idx[k];
    const i64 o_s = go_starts[o], o_e = go_stops[o];
    const i64 ks = keep_offsets[k];
    for (i64 v = o_s + lane; v < o_e; v += WAVE) {
        i64 vi = geno_v_idxs[v];
        vi = vi < 0 ? 0 : (vi >= n_variants ? n_variants - 1 : vi);
        const i64 pos = v_starts[vi];
        const i64 il = ilens[vi];
        const i64 end = pos - (il < 0 ? il : 0) + 1;
        keep[ks + (v - o_s)] = (pos >= ref_start && end <= ref_end) ? 1 : 0;
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

__global__ __launch_bounds__(256) void pack_genotypes_kernel(const int *geno_v_idxs, i64 n_geno, const gvl_vrec *vrec,
                                                              i64 n_variants, gvl_grec *out) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_geno) return;
    int v = geno_v_idxs[i];
    v = v < 0 ? 0 : ((i64)v >= n_variants ? (int)(n_variants - 1) : v);
    const i32x4 r = *reinterpret_cast<const i32x4 *>(vrec + v);
    const u32 alen = r.z < 0 ? 0xFFFFFFu : ((u32)r.z > 0xFFFFFFu ? 0xFFFFFFu : (u32)r.z);   // 0xFFFFFF = "ask vrec"
    gvl_grec g;
    g.pos = r.x; g.ilen = r.y; g.alen_inl = (alen << 8) | ((u32)r.w & 0xFFu); g.v_idx = v;
    out[i] = g;
}

// slot-major records: 8 lanes per genotype slot, one 128-byte line each
__global__ __launch_bounds__(256) void pack_slots_kernel(const i64 *go_starts, const i64 *go_stops, i64 n_slots,
                                                          const int *geno_v_idxs, const gvl_vrec *vrec,
                                                          const i64 *alt_offsets, i64 n_variants, gvl_srec *out) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 o = t >> 3;
    const int j = (int)(t & 7);
    if (o >= n_slots) return;
    const i64 o_s = go_starts[o];
    const i64 n = go_stops[o] - o_s;
    gvl_srec r;
    r.pos = 0; r.ilen = 0; r.alen_inl = GVL_SREC_EMPTY; r.a0 = 0;
    bool over = n > GVL_SLOT_RECS;
    int v = 0;
    i32x4 vr = {0, 0, 0, 0};
    if (!over && j < n) {
        v = geno_v_idxs[o_s + j];
        v = v < 0 ? 0 : ((i64)v >= n_variants ? (int)(n_variants - 1) : v);
        vr = *reinterpret_cast<const i32x4 *>(vrec + v);
    }
    // an allele too long for the 24-bit field sends the whole slot through the CSR
    const bool big = !over && j < n && (vr.z < 0 || vr.z >= 0xFFFFFF);
    over = over || (__builtin_amdgcn_ballot_w64(big) >> ((threadIdx.x & 63) & ~7) & 0xFFull) != 0;
    if (over) {
        if (j == 0) r.alen_inl = GVL_SREC_OVERFLOW;
    } else if (j < n) {
        r.pos = vr.x; r.ilen = vr.y; r.alen_inl = ((u32)vr.z << 8) | ((u32)vr.w & 0xFFu);
        r.a0 = (u32)(u64)alt_offsets[v];
    }
    out[t] = r;
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

// reverse.rs:75-84: rows given as (start, end) pairs instead of consecutive offsets
__global__ __launch_bounds__(256) void rc_bounded_rows_kernel(u8 *data, const i64 *bounds, const u8 *to_rc, i64 n_rows) {
    const i64 r = blockIdx.x;
    if (r >= n_rows || !to_rc[r]) return;
    const i64 b0 = bounds[2 * r], b1 = bounds[2 * r + 1];
    if (b1 <= b0) return;
    u8 *row = data + b0;
    const i64 n = b1 - b0;
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
        store_oh16(out + 16 * g, o);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const i64 j = n4 * 4 + threadIdx.x;
        const u32 d = luts.oh[in[j]];
        __builtin_memcpy(out + 4 * j, &d, 4);
    }
}


// ---------------------------------------------------------------------------------
// Tracks (BASELINE config 4): interval painting and realignment of
// reference-coordinate f32 tracks to a haplotype.  One wave per (row, chunk): the walk is
// planned with wave scans (lane j = variant j) into a 64-entry LDS table of the chunk's
// entries, then the wave streams 4 values per lane per trip; values inside a plain track run
// are one 16-B load + one 16-B store.  Rows the planner cannot express (coordinates beyond
// 2^30, > 64 entries per chunk) replay the reference's walk on the scalar unit with the same
// flush/compaction scheme as recon_wave_scalar.
// ---------------------------------------------------------------------------------
struct TrackArgs {
    const i64 *go_starts; const i64 *go_stops; const int *geno_v_idxs; const int *v_starts;
    const int *ilens; i64 n_variants;
    const gvl_grec *grec;       // genotype-inline records (gvl_static.geno_rec; NULL: geno_v_idxs -> v_starts / ilens)
    const int *regions; i64 regions_stride; const int *shifts; const i64 *geno_offset_idx;
    const u8 *keep; const i64 *keep_offsets; const u8 *to_rc; const i64 *out_offsets;
    i64 n_rows; int ploidy; int ploidy_shift; int chunk_len;
    const float *tracks; const i64 *track_offsets;
    double param; i64 strategy; u64 base_seed;
    const u64 *seed_ptr;        // non-NULL: base_seed is read from the device (the native loop's per-batch seeds)
    float *out;
    int dbg;
    const int2 *plan_hdr; const i32x4 *plan_ent;   // non-NULL: the rows' entry tables + per (row, chunk) which of them (track_plan_kernel)
    u64 *stamps;                        // diagnostics (gvl_diag_set_stamps): words 8.. count the chunk-waves that leave the common path
};
// A row's plan (track_plan_kernel): the entries of the WHOLE row -- {first value, kind, delta / position, fill length}, in
// output order, at most PLAN_MAXE -- and per chunk {first entry that reaches into it, how many do}; count < 0: not planned,
// the chunk's wave walks the row's variants itself.
constexpr int PLAN_MAXE = 128;
static inline i64 track_plan_bytes(i64 n_rows, i64 chunks) {      // headers, 256-byte aligned, then the entry tables
    return ((n_rows * chunks * (i64)sizeof(int2) + 255) & ~255ll) + n_rows * (i64)PLAN_MAXE * (i64)sizeof(i32x4);
}
enum : int { T_TRACK = 0, T_REPEAT = 1, T_FILL = 2, T_ZERO = 3 };
struct TrackMirror { int out[SEG_CAP]; int kind[SEG_CAP]; int plo[SEG_CAP]; int phi[SEG_CAP]; int vlen[SEG_CAP]; };

__device__ __forceinline__ u64 xorshift64_dev(u64 x) {   // src/tracks/mod.rs:31-36
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return x;
}
__device__ __forceinline__ u64 hash4_dev(u64 a, u64 b, u64 c, u64 d) {   // :48-54
    u64 h = a;
    h = xorshift64_dev(h ^ b);
    h = xorshift64_dev(h ^ c);
    h = xorshift64_dev(h ^ d);
    return h;
}

// ---- where the realignment reads a query's reference-coordinate track from -------------------------------
// SrcGlobal: a track in memory (gvl_realign_tracks; the scratch track the painter wrote).
// SrcPainted: the query's INTERVALS -- src/intervals.rs:19-126 evaluated at the positions the realignment asks for,
// so that the scratch track is neither written nor read (BASELINE config 4: 67 MB each way per batch, and one
// launch).  Position x of the track = reference position qs + x; its value = that of the last interval (in order)
// that covers it, 0.0 if none.  The wave keeps a WINDOW of the track as the tiled painter keeps a chunk: the
// candidate intervals' values / ends and a bitmap of their starts with its popcount prefix -- for a position in
// the window two LDS reads give the only candidate that can cover it (non-overlapping candidates; a window whose
// candidates overlap, or more than PAINT_TILE of them, is not used).  Any other position is looked up in the
// list itself (binary search + walk back under the running maximum of ends, as the per-value painter does):
// exactness never depends on what the window holds.
constexpr int PAINT_TILE = 256;
constexpr int PAINT_CHUNK = 2048;
constexpr int PAINT_WIN = 4096;                 // window positions (a chunk's 2048 + what its deletions skip + slack)
struct PaintIndex { const i64 *offsets; const int *base; const int *lo; const int *hi; };
struct PaintSrcArgs {                           // the interval set of one track (gvl_track_set) for the kernel
    const i64 *offset_idxs; i64 list_div;
    const int *itv_starts; const int *itv_ends; const float *itv_values; const i64 *itv_offsets; const int *pmax;
    PaintIndex X;
};
struct PaintWin { u32 bm[PAINT_WIN / 32 + 2]; u32 pre[PAINT_WIN / 32]; float cv[PAINT_TILE + 2]; int ce[PAINT_TILE + 2]; };   // (+ zero words behind the bitmap, slots behind the candidates)

struct SrcGlobal {
    const float *track; i64 tlen;
    __device__ __forceinline__ float at(const i64 x) const { return (x >= 0 && x < tlen) ? track[x] : 0.0f; }
    // (0 <= x, x + 4 <= tlen)
    __device__ __forceinline__ void at4(const i64 x, float (&v)[4]) const {
        const float *src = track + x;
        v[0] = src[0]; v[1] = src[1]; v[2] = src[2]; v[3] = src[3];
    }
    __device__ __forceinline__ void at4i(const int x, float (&v)[4]) const { at4((i64)x, v); }
    __device__ __forceinline__ void at8i(const int x, float (&v)[8]) const {
        const float *src = track + x;
#pragma unroll
        for (int g = 0; g < 8; ++g) v[g] = src[g];
    }
};
// One position looked up in the query's interval list itself (intervals.rs:19-126 for one position): binary search for the
// last start at or in front of it, then back under the running maximum of ends.  A real call, and the list's arrays
// are re-read from the kernel's arguments (`ps_kernarg` = where PaintSrcArgs sits in the kernarg segment): this is
// the rare path, and inlined it kept eight more pointers alive on the scalar side through the whole emit loop (the
// kernel spills scalars into vector lanes as it is: every such value costs a v_readlane where it is used).
typedef const PaintSrcArgs __attribute__((address_space(4))) *PaintSrcArgsK;
__device__ __noinline__ float paint_list_value(const u64 ps_kernarg, const i64 idx, const i64 qs, const i64 j) {
#if defined(__HIP_DEVICE_COMPILE__)
    const PaintSrcArgsK ps = (PaintSrcArgsK)(u64)rfl64((i64)ps_kernarg);
    const int *itv_starts = ps->itv_starts, *itv_ends = ps->itv_ends, *pmax = ps->pmax;
    const float *itv_values = ps->itv_values;
    const i64 li = rfl64(idx);
    const i64 s0 = ps->itv_offsets[li], e0 = ps->itv_offsets[li + 1];
    i64 lo = s0, hi = e0;                                      // first interval with start - qs > j
    while (lo < hi) {
        const i64 mid = (lo + hi) >> 1;
        if ((i64)itv_starts[mid] - qs <= j) lo = mid + 1; else hi = mid;
    }
    if (lo > s0 && (i64)pmax[lo - 1] - qs > j)
        for (i64 c = lo - 1; c >= s0; --c)
            if ((i64)itv_ends[c] - qs > j) return itv_values[c];
#endif
    return 0.0f;
}

struct SrcPainted {
    const PaintWin *W; i64 x_lo; int wlen; int base; bool win_ok;
    i64 tlen, qs, idx;
    u64 ps_kernarg;
    u64 *stamps;
    __device__ __forceinline__ float in_win(const int r) const {
        const u32 wd = W->bm[r >> 5];
        const int ig = base + (int)W->pre[r >> 5] + __builtin_popcount(wd & (0xFFFFFFFFu >> (31 - (r & 31))));
        return (ig >= 0 && W->ce[ig] > r) ? W->cv[ig] : 0.0f;
    }
    __device__ __forceinline__ float in_list(const i64 j) const {
        if (stamps) atomicAdd((unsigned long long *)&stamps[14], 1ull);             // positions looked up in the list itself
        return paint_list_value(ps_kernarg, idx, qs, j);
    }
    __device__ __forceinline__ float at(const i64 x) const {
        if (x < 0 || x >= tlen) return 0.0f;
        const i64 r = x - x_lo;
        if (win_ok && r >= 0 && r < wlen) return in_win((int)r);
        return in_list(x);
    }
    // four consecutive positions (0 <= x, x + 4 <= tlen).  Inside the window: the word that holds x's bit and the one
    // behind it as ONE 64-bit word (no special case for a group that straddles two words); a group with at most
    // one interval start behind its first position (all but 1-2 bp intervals) is two candidates and a switch point,
    // read without a branch.
    __device__ __forceinline__ void at4(const i64 x, float (&v)[4]) const {
        const i64 r64 = x - x_lo;
        if (win_ok && r64 >= 0 && r64 + 4 <= wlen) { win4((int)r64, v); return; }
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = at(x + g);
    }
    // (positions below 2^31 - 8: the caller has checked the row's table and track length)
    __device__ __forceinline__ void at4i(const int x, float (&v)[4]) const {
        const int r = x - (int)x_lo;
        if (win_ok && r >= 0 && r + 4 <= wlen) { win4(r, v); return; }
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = at((i64)x + g);
    }
    // eight consecutive positions: up to two interval starts behind the first position = three candidates and two
    // switch points, still without a branch (intervals of 1-3 bases in a row take the per-position form)
    __device__ __forceinline__ void at8i(const int x, float (&v)[8]) const {
        const int r = x - (int)x_lo;
        if (win_ok && r >= 0 && r + 8 <= wlen) {
            const int bp = r & 31, wi = r >> 5;
            const u64 w = ((u64)W->bm[wi + 1] << 32) | W->bm[wi];
            const int pre = base + (int)W->pre[wi];
            const int i0 = pre + __builtin_popcountll(w & (~0ull >> (63 - bp)));
            const u32 m7 = (u32)(w >> (bp + 1)) & 0x7Fu;      // starts at positions x + 1 .. x + 7
            if (__builtin_popcount(m7) <= 2) {
                const u32 m7b = m7 & (m7 - 1u);
                const int t1 = m7 ? __builtin_ctz(m7) + 1 : 8, t2 = m7b ? __builtin_ctz(m7b) + 1 : 8;
                const int ia = i0 < 0 ? PAINT_TILE : i0;         // (slot PAINT_TILE: "no candidate", end 0x80000000)
                const int e_a = W->ce[ia], e_b = W->ce[i0 + 1], e_c = W->ce[i0 + 2];
                const float c_a = W->cv[ia], c_b = W->cv[i0 + 1], c_c = W->cv[i0 + 2];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int e = g < t1 ? e_a : (g < t2 ? e_b : e_c);
                    const float c = g < t1 ? c_a : (g < t2 ? c_b : c_c);
                    v[g] = e > r + g ? c : 0.0f;
                }
            } else {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int ig = pre + __builtin_popcountll(w & (~0ull >> (63 - bp - g)));
                    v[g] = (ig >= 0 && W->ce[ig] > r + g) ? W->cv[ig] : 0.0f;
                }
            }
            return;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) v[g] = at((i64)x + g);
    }
    __device__ __forceinline__ void win4(const int r, float (&v)[4]) const {
        {
            const int bp = r & 31, wi = r >> 5;
            const u64 w = ((u64)W->bm[wi + 1] << 32) | W->bm[wi];
            const int i0 = base + (int)W->pre[wi] + __builtin_popcountll(w & (~0ull >> (63 - bp)));
            const u32 m3 = (u32)(w >> (bp + 1)) & 7u;         // starts at positions x + 1 .. x + 3
            if (__builtin_popcount(m3) <= 1) {
                const int t = m3 ? __builtin_ctz(m3) + 1 : 4;    // positions [t, 4) belong to candidate i0 + 1
                const int ia = i0 < 0 ? PAINT_TILE : i0;         // (slot PAINT_TILE: "no candidate", end 0x80000000)
                const int e_a = W->ce[ia], e_b = W->ce[i0 + 1];
                const float c_a = W->cv[ia], c_b = W->cv[i0 + 1];
#pragma unroll
                for (int g = 0; g < 4; ++g) v[g] = g < t ? (e_a > r + g ? c_a : 0.0f) : (e_b > r + g ? c_b : 0.0f);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ig = base + (int)W->pre[wi] + __builtin_popcountll(w & (~0ull >> (63 - bp - g)));
                    v[g] = (ig >= 0 && W->ce[ig] > r + g) ? W->cv[ig] : 0.0f;
                }
            }
        }
    }
};

__device__ __forceinline__ bool S_win_ok(const SrcPainted &S) { return S.win_ok; }
__device__ __forceinline__ bool S_win_ok(const SrcGlobal &) { return true; }
// One value of an insertion-fill region: src/tracks/mod.rs:87-190, evaluated per position.
// `i` = offset inside the region, `pp` = output index in the row.
template <class Src>
__device__ float fill_value(const TrackArgs &A, const Src &S, i64 vrp, i64 v_len, i64 i,
                            i64 pp, u64 query, u64 hap) {
#pragma clang fp contract(off)
    const i64 tlen = S.tlen;
    auto tr = [&](i64 x) -> float { return S.at(x); };
    if (A.strategy == GVL_FILL_REPEAT_5P) return tr(vrp);
    if (A.strategy == GVL_FILL_REPEAT_5P_NORM) return tr(vrp) / (float)v_len;
    if (A.strategy == GVL_FILL_CONSTANT) return (float)A.param;
    if (A.strategy == GVL_FILL_FLANK_SAMPLE) {
        const i64 width = (i64)A.param;
        const i64 lo = imax(vrp - width, 0), hi = imin(vrp + width, tlen - 1);
        const u64 pool = (u64)(hi - lo + 1);
        const u64 seed = hash4_dev(A.seed_ptr ? *A.seed_ptr : A.base_seed, query, hap, (u64)pp);
        return tr(lo + (i64)(pool ? seed % pool : 0));
    }
    if (A.strategy == GVL_FILL_INTERPOLATE) {
        const i64 order = (i64)A.param;
        const i64 k = (order + 1 + 1) / 2;
        const i64 n = 2 * k;
        auto xs = [&](i64 j) -> double { return j < k ? -(double)j : (double)v_len + (double)(j - k); };
        auto ys = [&](i64 j) -> double {
            return j < k ? (double)tr(imax(vrp - j, 0)) : (double)tr(imin(vrp + 1 + (j - k), tlen - 1));
        };
        const double x = (double)i;
        double acc = 0.0;
        for (i64 a = 0; a < n; ++a) {
            double term = ys(a);
            const double xa = xs(a);
            for (i64 b = 0; b < n; ++b) {
                if (b == a) continue;
                const double xb = xs(b);
                term = term * ((x - xb) / (xa - xb));
            }
            acc = acc + term;
        }
        return (float)acc;
    }
    return 0.0f;
}

// PAINT: the track comes from the query's intervals (SrcPainted; A.tracks is not read), else from memory.
template <bool PAINT>
__global__ __launch_bounds__(256) void realign_tracks_kernel(const TrackArgs A, const PaintSrcArgs PS_) {
    __shared__ TrackMirror mirror[4];
    __shared__ PaintWin wins[PAINT ? 4 : 1];
    // a double trip's 512 values on their way from "eight consecutive ones per lane" (how they are looked up) to "four per lane,
    // one contiguous KB per store instruction" (how they have to be stored, see the tight loop)
    __shared__ __attribute__((aligned(16))) float xpose[4][2 * TRIP];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = rfl((int)(threadIdx.x >> 6));
    TrackMirror &M = mirror[wave];
    const i64 k = (i64)blockIdx.x * 4 + wave;
    if (k >= A.n_rows) return;
    const int chunk = blockIdx.y;
    const i64 query = A.ploidy_shift >= 0 ? (k >> A.ploidy_shift) : (i64)((u32)k / (u32)A.ploidy);
    const i64 hap = k - query * A.ploidy;
    // the row's parameters are wave-uniform: scalar loads (through the constant address space, which is what makes the
    // compiler pick s_load for a global array; as recon_lean_kernel reads its request entries)
    typedef const int __attribute__((address_space(4))) *KInt;
    typedef const i64 __attribute__((address_space(4))) *KI64;
    const i64 *oi_ptr = nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (PAINT) {       // (the pointer itself is a kernel argument: fetched with the others, not behind round 1)
        oi_ptr = ((PaintSrcArgsK)((u64)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(TrackArgs)))->offset_idxs;
        asm volatile("" :: "s"(oi_ptr), "s"(A.out_offsets), "s"(A.track_offsets), "s"(A.regions), "s"(A.shifts), "s"(A.geno_offset_idx));
    }
#endif
    // Round 1: everything the row's number alone addresses, requested as a whole before any of it is used (written value
    // by value, each load became a memory round trip of its own behind the branch in front of its first use).
    const i64 row_base = ((KI64)(u64)A.out_offsets)[k];
    const i64 row_end = ((KI64)(u64)A.out_offsets)[k + 1];
    const i64 t_s = ((KI64)(u64)A.track_offsets)[query];
    const i64 t_e = ((KI64)(u64)A.track_offsets)[query + 1];
    const i64 q_start = ((KInt)(u64)A.regions)[query * A.regions_stride + 1];
    const i64 shift = ((KInt)(u64)A.shifts)[k];
    const i64 o_idx = ((KI64)(u64)A.geno_offset_idx)[k];
    const bool has_keep = A.keep && A.keep_offsets;
    const i64 keep_off = has_keep ? ((KI64)(u64)A.keep_offsets)[k] : 0;
    // (to_rc is a byte array: the aligned word that holds the row's byte)
    const u64 rc_addr = (u64)A.to_rc + (u64)k;
    const int rc_word = A.to_rc ? ((KInt)(rc_addr & ~3ull))[0] : 0;
    i64 idx_raw = 0;                        // (the interval list's number)
    // (the chunk's walk state, if the caller prepared them: eight scalars of the same round)
    int ph_first = 0, ph_count = -1;
    if (A.plan_hdr) {
        const KInt w = (KInt)(u64)(A.plan_hdr + (k * (i64)gridDim.y + (i64)blockIdx.y));
        ph_first = w[0]; ph_count = w[1];
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (PAINT) idx_raw = ((KI64)(u64)oi_ptr)[query];
    asm volatile("" :: "s"(row_base), "s"(row_end), "s"(t_s), "s"(t_e), "s"((int)q_start), "s"((int)shift), "s"(o_idx), "s"(keep_off),
                 "s"(rc_word), "s"(idx_raw));
#endif
    // Round 2: the genotype list's bounds
    const i64 o_s = ((KI64)(u64)A.go_starts)[o_idx];
    const i64 nv64 = ((KI64)(u64)A.go_stops)[o_idx] - o_s;
    const int L = (int)(row_end - row_base);
    const int lo_clip = chunk * A.chunk_len;
    if (lo_clip >= L) return;
    const int hi_clip = (L - lo_clip > A.chunk_len) ? lo_clip + A.chunk_len : L;
    const i64 tlen = t_e - t_s;
    const float *track = PAINT ? nullptr : A.tracks + t_s;
    const int n_var = nv64 < 0 ? 0 : (nv64 > 0x7FFFFFFFll ? 0x7FFFFFFF : (int)nv64);
    const bool rc = ((rc_word >> (8 * (int)(rc_addr & 3ull))) & 0xFF) != 0;
    float *out_row = A.out + row_base;

    int s_out = 0, s_kind = 0, s_plo = 0, s_phi = 0, s_vlen = 0;
    int nseg = 0;
    int last_kind = -1; i64 last_p = 0;
    auto push = [&](int kind, int o_start, int len, i64 pval, int vlen) {
        if (len <= 0 || o_start + len <= lo_clip || o_start >= hi_clip) return;
        if (kind == T_TRACK && last_kind == T_TRACK && pval == last_p) return;   // run continues
        if (lane == nseg) {
            s_out = o_start; s_kind = kind; s_plo = (int)(u32)(u64)pval; s_phi = (int)(u32)((u64)pval >> 32);
            s_vlen = vlen;
            M.out[lane] = o_start; M.kind[lane] = kind; M.plo[lane] = s_plo; M.phi[lane] = s_phi; M.vlen[lane] = vlen;
        }
        last_kind = kind; last_p = pval; ++nseg;
    };

    i64 track_idx = 0, shifted = 0;
    int out_idx = 0;
    int r_pos = 0, r_ilen = 0, r_keep = 1;
    int vi = 0, vb = -WAVE;
    bool walk_done = false;
    int emit_pos = lo_clip;
    if (n_var == 0) {   // src/tracks/mod.rs:240-246: out[:] = track[:length]
        push(T_TRACK, 0, L, 0, 0);
        out_idx = L;
        walk_done = true;
    }
    // The row was planned as a whole (track_plan_kernel: by the native loop with its epoch table, or once per gvl_tracks_batch
    // call): the chunk's entries are `ph_count` consecutive ones of the row's table -- one read, no walk.
    if (!walk_done && ph_count > 0 && ph_count <= SEG_CAP && !(A.dbg & (8 | 268435456))) {
        if (lane < ph_count) {
            const i32x4 e = A.plan_ent[k * (i64)PLAN_MAXE + ph_first + lane];
            M.out[lane] = e.x; M.kind[lane] = (e.y == T_FILL && A.strategy == GVL_FILL_REPEAT_5P) ? (int)T_REPEAT : e.y;
            M.plo[lane] = e.z; M.phi[lane] = e.z >> 31; M.vlen[lane] = e.w;
        }
        nseg = ph_count;
        out_idx = L;
        walk_done = true;
        if (A.stamps && lane == 0) atomicAdd((unsigned long long *)&A.stamps[15], 1ull);       // chunk-waves that read their plan
    }

    // Planned walk: the same restatement as reconstruct_kernel's P3 (lane j = variant j, wave
    // scans for "first ALT wins" and the output offsets), with the track rules: SNPs only take
    // part in the shift (:312-314), every applied indel ends a track run, no lead pad.  It
    // writes the chunk's entries straight into the LDS table; anything it cannot express in
    // i32 / 64 entries leaves the table empty for the scalar walk below.
    if (!walk_done && !(A.dbg & 8)) {
        bool ok = q_start > -(1ll << 30) && q_start < (1ll << 30) && shift >= 0 && shift < (1ll << 30);
        int rem = (int)(ok ? shift : 0);
        int tidx0 = 0, pm_carry = 0, x_carry = 0;
        bool ended = false, past_chunk = false;
        int tidx_end = 0, out_end = 0;
        int n_ent = 0;
        const int tb0 = 0;
        const int qs = (int)(ok ? q_start : 0);
        auto put = [&](int q, int kind, int o_start, i64 pval, int vlen) {
            if (q < SEG_CAP) {
                M.out[q] = o_start; M.kind[q] = kind; M.plo[q] = (int)(u32)(u64)pval;
                M.phi[q] = (int)(u32)((u64)pval >> 32); M.vlen[q] = vlen;
            }
        };
        // (position and length delta sit next to the CSR entry, gvl_grec: one read, not three; a trip's records are
        // requested a trip ahead)
        int nxt_pos = 0, nxt_d = 0;
        if (A.grec && tb0 + lane < n_var) {
            const i32x4 rec = *reinterpret_cast<const i32x4 *>(A.grec + (o_s + tb0 + lane));
            nxt_pos = rec.x; nxt_d = rec.y;
        }
        for (int tb = tb0; tb < n_var && ok && !ended && !past_chunk; tb += WAVE) {
            int pos = 0, d = 0;
            bool valid = tb + lane < n_var;
            const int rec_pos = nxt_pos, rec_d = nxt_d;
            if (A.grec && tb + WAVE + lane < n_var) {
                const i32x4 rec = *reinterpret_cast<const i32x4 *>(A.grec + (o_s + tb + WAVE + lane));
                nxt_pos = rec.x; nxt_d = rec.y;
            }
            if (valid) {
                if (A.grec) {
                    pos = rec_pos; d = rec_d;
                } else {
                    int v = A.geno_v_idxs[o_s + tb + lane];
                    v = v < 0 ? 0 : ((i64)v >= A.n_variants ? (int)(A.n_variants - 1) : v);
                    pos = A.v_starts[v]; d = A.ilens[v];
                }
                if (has_keep) valid = A.keep[keep_off + tb + lane] != 0;
            }
            const bool weird = valid && (pos <= -(1 << 30) || pos >= (1 << 30) || d <= -(1 << 30) || d >= (1 << 30));
            ok = ok && __builtin_amdgcn_ballot_w64(weird) == 0;
            const int vrp = pos - qs;                                   // :264
            const int E = vrp - (d < 0 ? d : 0) + 1;                    // :267
            int v_len = (d > 0 ? d : 0) + 1;                            // :282
            const u64 m_span = __builtin_amdgcn_ballot_w64(valid && d < 0 && vrp < 0 && E >= 0);   // :271-274
            if (m_span) { tidx0 = rdl(E, 63 - __builtin_clzll(m_span)); pm_carry = tidx0; tidx_end = tidx0; }
            bool cand = valid && vrp >= 0;
            if (rem > 0) {                                              // :285-308
                const int base = tidx0;
                const u64 m_t = __builtin_amdgcn_ballot_w64(cand && vrp >= base && (vrp - base) + v_len >= rem);
                if (m_t == 0) {
                    cand = false;
                } else {
                    const int f = __builtin_ctzll(m_t);
                    const int dist = rdl(vrp, f) - base;
                    if (dist >= rem) {
                        tidx0 = base + rem;
                        cand = cand && lane >= f;
                    } else {
                        const int skip = rem - dist;
                        if (skip == rdl(v_len, f)) {
                            tidx0 = rdl(E, f);
                            cand = cand && lane > f;
                        } else {
                            tidx0 = rdl(vrp, f);
                            cand = cand && lane >= f;
                            if (lane == f) v_len -= skip;
                        }
                    }
                    rem = 0;
                    pm_carry = tidx0; tidx_end = tidx0;
                }
            }
            cand = cand && d != 0;                                      // :312-314
            bool inB = cand;
            int PM = 0, pm_incl = 0;
            {
                u64 mB = __builtin_amdgcn_ballot_w64(inB);
                bool stable = false;
#pragma unroll 1
                for (int it = 0; it < 4 && !stable; ++it) {
                    PM = wave_scan_exclusive<OpMaxU>(inB ? E : 0, pm_incl);
                    PM = PM > pm_carry ? PM : pm_carry;
                    inB = cand && vrp >= PM;                            // :277-279
                    const u64 m2 = __builtin_amdgcn_ballot_w64(inB);
                    stable = m2 == mB;
                    mB = m2;
                }
                ok = ok && stable;
            }
            const int n_i = inB ? vrp - PM : 0;                         // :317
            const int S_i = inB ? OpSat::f(n_i, v_len) : 0;
            int x_incl;
            const int X = OpSat::f(x_carry, wave_scan_exclusive<OpSat>(S_i, x_incl));   // out_idx before this variant
            const int fill_out = OpSat::f(X, n_i);
            const bool applied = inB && fill_out < L;                   // :319-321 break
            const int w_i = applied ? ((v_len < L - fill_out) ? v_len : L - fill_out) : 0;   // :329
            const u64 m_inB = __builtin_amdgcn_ballot_w64(inB);
            const u64 m_app = __builtin_amdgcn_ballot_w64(applied);
            if (m_inB != m_app) ended = true;
            if (m_app) {
                const int last = 63 - __builtin_clzll(m_app);
                tidx_end = rdl(E, last);
                out_end = rdl(fill_out, last) + rdl(w_i, last);
                if (rdl(fill_out, last) >= hi_clip) past_chunk = true;
            }
            const bool e_trk = applied && n_i > 0 && fill_out > lo_clip && X < hi_clip;
            const bool e_fil = applied && w_i > 0 && fill_out + w_i > lo_clip && fill_out < hi_clip;
            const int slot0 = n_ent + wave_scan_exclusive<OpAdd>((e_trk ? 1 : 0) + (e_fil ? 1 : 0));
            const int add_ent = __builtin_popcountll(__builtin_amdgcn_ballot_w64(e_trk)) +
                                __builtin_popcountll(__builtin_amdgcn_ballot_w64(e_fil));
            if (n_ent + add_ent + 2 > SEG_CAP) ok = false;
            if (ok) {
                int q = slot0;
                if (e_trk) { put(q, T_TRACK, X, (i64)PM - X, 0); ++q; }
                if (e_fil) put(q, (d > 0 && A.strategy != GVL_FILL_REPEAT_5P) ? T_FILL : T_REPEAT, fill_out, (i64)vrp, v_len);
            }
            if (tb + WAVE < n_var) {
                const int mx = rdl(pm_incl, 63);
                pm_carry = mx > pm_carry ? mx : pm_carry;
                x_carry = OpSat::f(x_carry, rdl(x_incl, 63));
            }
            n_ent += add_ent;
        }
        if (ok) {
            if (!(past_chunk && !ended)) {                              // :365-392 tail
                i64 t_idx = tidx_end;
                if (rem > 0) t_idx = imin((i64)tidx0 + rem, tlen);
                const int u = L - out_end;
                if (u > 0 && lane == 0) {
                    const i64 avail = tlen - t_idx;
                    const int w = (int)imin((i64)u, avail);
                    int end = out_end;
                    int q = n_ent;
                    if (w > 0) {
                        end += w;
                        if (end > lo_clip && out_end < hi_clip) { put(q, T_TRACK, out_end, t_idx - out_end, 0); ++q; }
                    }
                    if (end < L && L > lo_clip && end < hi_clip) { put(q, T_ZERO, end, 0, 0); ++q; }
                    n_ent = q;
                }
                n_ent = rfl(n_ent);
            }
            nseg = n_ent;
            out_idx = L;
            walk_done = true;
        }
    }

    typename std::conditional<PAINT, SrcPainted, SrcGlobal>::type S;
    bool have_win = false;
    if constexpr (PAINT) {
#if defined(__HIP_DEVICE_COMPILE__)
        const PaintSrcArgsK PSk = (PaintSrcArgsK)((u64)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(TrackArgs));
        const i64 dv = PSk->list_div;
        const i64 idx = dv == 1 ? idx_raw : rfl64(idx_raw / dv);
#else
        const i64 idx = 0;
#endif
        S.W = &wins[wave]; S.x_lo = 0; S.wlen = 0; S.base = -1; S.win_ok = false;
        S.tlen = tlen; S.qs = q_start; S.idx = idx;
        S.ps_kernarg = (u64)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(TrackArgs);
        S.stamps = A.stamps;
    } else {
        S.track = track; S.tlen = tlen;
    }
    // the window of a painted source: from where the chunk's first value comes from on (see SrcPainted)
    auto build_window = [&](const i64 x_first) {
        PaintWin &Wn = wins[PAINT ? wave : 0];
        i64 x_lo = x_first - 64;
        x_lo = x_lo < 0 ? 0 : x_lo;
        const i64 wl64 = tlen - x_lo;
        const int wlen = wl64 > PAINT_WIN ? PAINT_WIN : (wl64 < 0 ? 0 : (int)wl64);
        // (the interval set's pointers are read from the kernel's arguments HERE, not held on the scalar side since the
        // kernel's start: the walk in between needs every scalar register it can get)
#if defined(__HIP_DEVICE_COMPILE__)
        const PaintSrcArgsK PS = (PaintSrcArgsK)((u64)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(TrackArgs));
#else
        const PaintSrcArgs *const PS = &PS_;
#endif
        i64 idx = 0;
        if constexpr (PAINT) idx = S.idx;
        // the list's bounds and its bucket index's: four lanes of ONE load (+ the index's first position), then both bucket
        // bounds in one -- written value by value the compiler made each of them a memory round trip of its own, and the
        // kernel's time is the length of a wave's chain of dependent round trips as much as its instruction count
        // (measured: 59 us at 7 waves per SIMD, 67 at 4, 79 at 3)
        int h_lo = 0, h_hi = 0, h_b = 0;
        {
            const i64 *const io = PS->itv_offsets, *const xo = PS->X.offsets;
            const int *const xb = PS->X.base;
            if (lane < 4 && (lane < 2 || xo)) {
                const i64 h = lane < 2 ? io[idx + lane] : xo[idx + (lane - 2)];
                h_lo = (int)(u32)(u64)h; h_hi = (int)(u32)((u64)h >> 32);
            }
            if (lane == 0 && xb) h_b = xb[idx];
        }
        const i64 s0 = rdl64(h_lo, h_hi, 0), e0 = rdl64(h_lo, h_hi, 1);
        if (wlen <= 0 || !PS->X.offsets || e0 <= s0) return;
        const i64 b0 = rdl64(h_lo, h_hi, 2);
        const i64 nb = rdl64(h_lo, h_hi, 3) - b0;
        if (nb <= 0) return;
        const i64 bbase = rdl(h_b, 0);
        i64 ba = (q_start + x_lo - bbase) >> 11, bb = (q_start + x_lo + wlen - 1 - bbase) >> 11;
        ba = ba < 0 ? 0 : (ba > nb - 1 ? nb - 1 : ba);
        bb = bb < 0 ? 0 : (bb > nb - 1 ? nb - 1 : bb);
        int lohi = 0;
        if (lane < 2) lohi = lane == 0 ? PS->X.lo[b0 + ba] : PS->X.hi[b0 + bb];
        i64 lo_c = s0 + rdl(lohi, 0);
        const i64 hi_c = s0 + rdl(lohi, 1);
        if (lo_c > hi_c) lo_c = hi_c;
        if (hi_c - lo_c > PAINT_TILE) return;
        const int n_c = (int)(hi_c - lo_c);
        Wn.bm[lane] = 0u; Wn.bm[WAVE + lane] = 0u;
        if (lane < 2) Wn.bm[2 * WAVE + lane] = 0u;
        if (lane == 0) { Wn.ce[PAINT_TILE] = (int)0x80000000; Wn.cv[PAINT_TILE] = 0.0f; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bool bad = false;
        int n_before = 0;
        int carry_e = (int)0x80000000, carry_s = (int)0x80000000;          // end / start of the candidate in front
        // (all candidate records are requested before the first one is used: one memory round trip; rounds the
        // window has no candidates for are skipped as a whole)
        int c_s[PAINT_TILE / WAVE], c_e[PAINT_TILE / WAVE]; float c_v[PAINT_TILE / WAVE];
#pragma unroll
        for (int r_ = 0; r_ < PAINT_TILE / WAVE; ++r_) {
            const int i = r_ * WAVE + lane;
            c_s[r_] = 0; c_e[r_] = 0; c_v[r_] = 0.0f;
            if (r_ * WAVE < n_c) {
                if (i < n_c) { c_s[r_] = PS->itv_starts[lo_c + i]; c_e[r_] = PS->itv_ends[lo_c + i]; c_v[r_] = PS->itv_values[lo_c + i]; }
            }
        }
        const i64 qx64 = q_start + x_lo;             // the window's first position on the reference
        const bool qx_small = qx64 > -(1ll << 30) && qx64 < (1ll << 30);
        const int qx = (int)qx64;
#pragma unroll
        for (int r_ = 0; r_ < PAINT_TILE / WAVE; ++r_) {
            const int b = r_ * WAVE;
            if (b >= n_c) break;
            const int i = b + lane;
            int sr = 0x7FFFFFFF, er = 0x7FFFFFFF;
            // starts / ends relative to the window: 32-bit arithmetic when nothing can overflow it (always, for real
            // coordinates), else 64-bit and clamped
            const bool small = qx_small && __builtin_amdgcn_ballot_w64(i < n_c && (c_s[r_] <= -(1 << 30) || c_s[r_] >= (1 << 30) ||
                                                                                  c_e[r_] <= -(1 << 30) || c_e[r_] >= (1 << 30))) == 0;
            if (i < n_c) {
                if (small) {
                    sr = c_s[r_] - qx; er = c_e[r_] - qx;
                } else {
                    i64 s64 = (i64)c_s[r_] - qx64, e64 = (i64)c_e[r_] - qx64;
                    s64 = s64 < -(1ll << 30) ? -(1ll << 30) : (s64 > (1ll << 30) ? (1ll << 30) : s64);
                    e64 = e64 < -(1ll << 30) ? -(1ll << 30) : (e64 > (1ll << 30) ? (1ll << 30) : e64);
                    sr = (int)s64; er = (int)e64;
                }
                Wn.ce[i] = er;
                Wn.cv[i] = c_v[r_];
            }
            const int pe = dpp_mov<0x138, 0xf>(carry_e, er), ps = dpp_mov<0x138, 0xf>(carry_s, sr);    // wave_shr:1 (lane 0: the round before)
            if (i < n_c && (sr < pe || sr == ps)) bad = true;
            if (i < n_c && sr >= 0 && sr < wlen) atomicOr(&Wn.bm[sr >> 5], 1u << (sr & 31));
            n_before += __builtin_popcountll(__builtin_amdgcn_ballot_w64(i < n_c && sr < 0));
            const int last = (n_c - b > WAVE ? WAVE : n_c - b) - 1;
            carry_e = rdl(er, last); carry_s = rdl(sr, last);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (__builtin_amdgcn_ballot_w64(bad) != 0) return;
        {   // exclusive popcount prefix over the 128 words: lane owns words 2 lane, 2 lane + 1
            const int c0 = __builtin_popcount(Wn.bm[2 * lane]), c1 = __builtin_popcount(Wn.bm[2 * lane + 1]);
            const int incl = wave_scan_inclusive<OpAdd>(c0 + c1);
            Wn.pre[2 * lane] = (u32)(incl - c0 - c1);
            Wn.pre[2 * lane + 1] = (u32)(incl - c1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (PAINT) { S.x_lo = x_lo; S.wlen = wlen; S.base = n_before - 1; S.win_ok = !(A.dbg & 2097152); }
    };

    if (A.stamps && lane == 0) {
        atomicAdd((unsigned long long *)&A.stamps[8], 1ull);                       // chunk-waves
        if (!walk_done) atomicAdd((unsigned long long *)&A.stamps[9], 1ull);       // ... that replay the walk on the scalar unit
    }
    for (;;) {
        while (!walk_done && nseg <= SEG_FLUSH) {
            bool stop = (vi >= n_var) || (out_idx >= hi_clip);
            if (!stop) {
                if (vi - vb >= WAVE) {
                    vb = vi;
                    const int j = vb + lane;
                    if (j < n_var) {
                        if (A.grec) {
                            const i32x4 rec = *reinterpret_cast<const i32x4 *>(A.grec + (o_s + j));
                            r_pos = rec.x; r_ilen = rec.y;
                        } else {
                            int v = A.geno_v_idxs[o_s + j];
                            v = v < 0 ? 0 : ((i64)v >= A.n_variants ? (int)(A.n_variants - 1) : v);
                            r_pos = A.v_starts[v]; r_ilen = A.ilens[v];
                        }
                        r_keep = has_keep ? (int)A.keep[keep_off + j] : 1;
                    }
                }
                const int i = vi - vb;
                ++vi;
                if (has_keep && rdl(r_keep, i) == 0) continue;
                const i64 vrp = (i64)rdl(r_pos, i) - q_start;           // mod.rs:264
                const i64 d = rdl(r_ilen, i);
                const i64 vre = vrp - (d < 0 ? d : 0) + 1;               // :267
                if (d < 0 && vrp < 0 && vre >= 0) { track_idx = vre; continue; }   // :271-274
                if (vrp < track_idx) continue;                          // :277-279
                i64 v_len = (d > 0 ? d : 0) + 1;                         // :282
                if (shifted < shift) {                                   // :285-308
                    const i64 dist = vrp - track_idx;
                    if (shifted + dist + v_len < shift) continue;
                    if (shifted + dist >= shift) {
                        track_idx += shift - shifted;
                        shifted = shift;
                    } else {
                        const i64 a0 = shift - shifted - dist;
                        shifted = shift;
                        if (a0 == v_len) { track_idx = vre; continue; }
                        track_idx = vrp;
                        v_len -= a0;
                    }
                }
                if (d == 0) continue;                                    // :312-314 SNPs do not move tracks
                const i64 n64 = vrp - track_idx;
                if (n64 >= (i64)(L - out_idx)) {                          // :319-321
                    stop = true;
                } else {
                    const int n = (int)n64;
                    push(T_TRACK, out_idx, n, track_idx - out_idx, 0);
                    out_idx += n;
                    const int w = (int)imin(v_len, (i64)(L - out_idx));  // :329
                    const int vl = (int)imin(v_len, 0x7FFFFFFFll);
                    if (d > 0 && A.strategy != GVL_FILL_REPEAT_5P) push(T_FILL, out_idx, w, vrp, vl);   // :333-346
                    else push(T_REPEAT, out_idx, w, vrp, vl);            // :347-354
                    out_idx += w;
                    track_idx = vre;
                    if (out_idx >= L) stop = true;
                }
            }
            if (stop) {
                if (shifted < shift) track_idx = imin(track_idx + (shift - shifted), tlen);   // :365-369
                const int u = L - out_idx;
                if (u > 0) {
                    const int w = (int)imin((i64)u, tlen - track_idx);
                    int end = out_idx;
                    if (w > 0) { push(T_TRACK, out_idx, w, track_idx - out_idx, 0); end += w; }
                    if (end < L) push(T_ZERO, end, L - end, 0, 0);
                }
                out_idx = out_idx > L ? out_idx : L;
                walk_done = true;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const int cov = out_idx < lo_clip ? lo_clip : (out_idx > hi_clip ? hi_clip : out_idx);
        const int limit = walk_done ? hi_clip : (cov & ~3);
        if (A.dbg & 8388608) return;              // (timing ablation: head + walk only)
        if (PAINT && !have_win && nseg > 0) {
            // the first entry that reaches into the chunk says where its first value comes from
            // (entry 0: the table only holds entries that reach into what is still to be emitted)
            have_win = true;
            const i64 pv = (i64)(((u64)(u32)rfl(M.phi[0]) << 32) | (u32)rfl(M.plo[0]));
            build_window(rfl(M.kind[0]) == T_TRACK ? pv + emit_pos : pv);
        }
        // The round's entries, lane t = entry t (they are few: a chunk with one indel is run | fill | run); what a trip needs of
        // them is wave-uniform -- the entry its first position lies in (a ballot over the starts), that entry's kind, delta and
        // end (readlanes) -- so a trip inside ONE track run, which is all of them in a chunk without an indel and most in any
        // other, needs no per-lane lookup however long the table is.  (Until round 4 only tables of <= 4 entries were
        // dispatched this way, with their starts held in scalar registers: a chunk with two indels took the per-lane path on all
        // of its trips -- 6.5 % of BASELINE config 4's chunk-waves, two thirds of that path's trips.)
        const bool ev = lane < nseg;
        // (kind + 256: the entry's delta does not fit 32 bits -- its high word and a fill's length stay in LDS, read where needed)
        const int e_out = ev ? M.out[lane] : 0x7FFFFFFF, e_plo = ev ? M.plo[lane] : 0;
        const int e_kind = ev ? (M.kind[lane] | (M.phi[lane] == (e_plo >> 31) ? 0 : 256)) : T_ZERO;
        const bool tl32 = tlen < 0x7FFFFF00ll;
        if (A.dbg & 16777216) return;             // (timing ablation: ... + the window)
        if (A.stamps && lane == 0) {
            if (PAINT && !S_win_ok(S)) atomicAdd((unsigned long long *)&A.stamps[10], 1ull);     // ... without a window
            if (nseg > 4) atomicAdd((unsigned long long *)&A.stamps[11], 1ull);                   // ... with more than 4 entries
            atomicAdd((unsigned long long *)&A.stamps[12], (unsigned long long)nseg);
        }
        int p0 = emit_pos;
        while (p0 < limit) {
            // the entry p0 lies in: the last one that starts at or in front of it
            int li = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ev && e_out <= p0)) - 1;
            li = li < 0 ? 0 : li;
            {
                const int kd = rdl(e_kind, li), plo = rdl(e_plo, li);
                int nx = li + 1 < nseg ? rdl(e_out, li + 1) : cov;
                nx = nx < cov ? nx : cov;
                const int run_end = nx < limit ? nx : limit;
                if (nseg > 0 && kd == T_TRACK && tl32 && run_end - p0 >= TRIP) {
                    const i64 xs0 = (i64)plo + p0;
                    const int n2 = (run_end - p0) / (2 * TRIP);
                    // the run's double trips in a tight loop -- position and output pointer advance by a constant, nothing is
                    // selected per trip, eight values per lane (half the lookups per value)
                    if (n2 > 0 && xs0 >= 0 && xs0 + (i64)n2 * (2 * TRIP) <= tlen) {
                        // Stored as they are looked up -- two 16-byte stores per lane, 32 bytes from lane to lane -- every store
                        // instruction would write 64 half-lines: measured, that pattern costs the kernel 10 of its 49 us, and
                        // with one contiguous KB per instruction the stores cost nothing (same time as no stores at all).  So
                        // the eight values go through LDS: written as looked up, read back four per lane.
                        int x = (int)xs0 + 2 * GROUP * lane;
                        float *const xp = xpose[wave];
                        float *o = rc ? out_row + (L - GROUP - (p0 + GROUP * lane)) : out_row + (p0 + GROUP * lane);
                        const int ostep = rc ? -2 * TRIP : 2 * TRIP;
#pragma unroll 1
                        for (int i = 0; i < n2; ++i) {
                            float v8[2 * GROUP];
                            S.at8i(x, v8);
                            {
                                const v4f_t a = {v8[0], v8[1], v8[2], v8[3]}, b = {v8[4], v8[5], v8[6], v8[7]};
                                *reinterpret_cast<v4f_t *>(xp + 2 * GROUP * lane) = a;
                                *reinterpret_cast<v4f_t *>(xp + 2 * GROUP * lane + GROUP) = b;
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            const v4f_t q1 = *reinterpret_cast<const v4f_t *>(xp + GROUP * lane);
                            const v4f_t q2 = *reinterpret_cast<const v4f_t *>(xp + TRIP + GROUP * lane);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            if (!rc) {
                                store_f32x4(o, q1[0], q1[1], q1[2], q1[3]);
                                store_f32x4(o + TRIP, q2[0], q2[1], q2[2], q2[3]);
                            } else {
                                store_f32x4(o, q1[3], q1[2], q1[1], q1[0]);
                                store_f32x4(o - TRIP, q2[3], q2[2], q2[1], q2[0]);
                            }
                            x += 2 * TRIP; o += ostep;
                        }
                        p0 += n2 * (2 * TRIP);
                        continue;
                    }
                    if (xs0 >= 0 && xs0 + TRIP <= tlen) {       // one trip of the run
                        float v4[GROUP];
                        const int p = p0 + GROUP * lane;
                        S.at4i((int)xs0 + GROUP * lane, v4);
                        if (!rc) store_f32x4(out_row + p, v4[0], v4[1], v4[2], v4[3]);
                        else store_f32x4(out_row + (L - GROUP - p), v4[3], v4[2], v4[1], v4[0]);
                        p0 += TRIP;
                        continue;
                    }
                }
            }
            // A trip that crosses entries (or reaches the row's end, or a track's): entry by entry, each wave-uniform -- its
            // kind, delta and range are scalars, a lane takes from it the positions of its group that lie inside.
            if (A.stamps && lane == 0) atomicAdd((unsigned long long *)&A.stamps[13], 1ull);    // trips on this path
            const int p = p0 + GROUP * lane;
            float v4[GROUP];
#pragma unroll
            for (int g = 0; g < GROUP; ++g) v4[g] = 0.0f;
            const int trip_end = (limit - p0 > TRIP) ? p0 + TRIP : limit;
            if (nseg > 0) {
#pragma unroll 1
                for (;;) {
                    const int st_e = rdl(e_out, li), kd = rdl(e_kind, li) & 255;
                    const i64 pv = (i64)(((u64)(u32)rfl(M.phi[li]) << 32) | (u32)rdl(e_plo, li));
                    int nx = li + 1 < nseg ? rdl(e_out, li + 1) : cov;
                    nx = nx < trip_end ? nx : trip_end;
                    const bool mine = p + GROUP > st_e && p < nx;              // some position of the lane's group is this entry's
                    float t4[GROUP];
#pragma unroll
                    for (int g = 0; g < GROUP; ++g) t4[g] = 0.0f;
                    if (kd == T_TRACK) {
                        if (mine) {
                            const i64 x = pv + p;
                            if (x >= 0 && x + GROUP <= tlen) {
                                S.at4(x, t4);
                            } else {
#pragma unroll
                                for (int g = 0; g < GROUP; ++g) t4[g] = S.at(x + g);
                            }
                        }
                    } else if (kd == T_REPEAT) {
                        const float v = S.at(pv);                               // (one position for the whole entry)
#pragma unroll
                        for (int g = 0; g < GROUP; ++g) t4[g] = v;
                    } else if (kd == T_FILL) {
                        const i64 vl = (i64)rfl(M.vlen[li]);
#pragma unroll
                        for (int g = 0; g < GROUP; ++g) {
                            const int pp = p + g;
                            if (pp >= st_e && pp < nx) t4[g] = fill_value(A, S, pv, vl, (i64)(pp - st_e), (i64)pp, (u64)query, (u64)hap);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < GROUP; ++g)
                        if (p + g >= st_e && p + g < nx) v4[g] = t4[g];
                    if (nx >= trip_end || li + 1 >= nseg) break;
                    ++li;
                }
            }
            if (p + GROUP <= limit) {
                if (!rc) store_f32x4(out_row + p, v4[0], v4[1], v4[2], v4[3]);
                else store_f32x4(out_row + (L - GROUP - p), v4[3], v4[2], v4[1], v4[0]);
            } else {
#pragma unroll
                for (int i = 0; i < GROUP; ++i)
                    if (p + i < limit) out_row[rc ? (L - 1 - (p + i)) : (p + i)] = v4[i];
            }
            p0 += TRIP;
        }
        emit_pos = limit;
        if (walk_done || emit_pos >= hi_clip) break;
        {   // compact: keep the segment that holds emit_pos and everything after it
            int cnt = 0;
            for (int s2 = 0; s2 < nseg; ++s2) cnt += (rdl(s_out, s2) <= emit_pos) ? 1 : 0;
            const int s0 = cnt > 0 ? cnt - 1 : 0;
            if (s0 > 0) {
                const int srcl = lane + s0 < SEG_CAP ? lane + s0 : SEG_CAP - 1;
                s_out = bperm(srcl, s_out); s_kind = bperm(srcl, s_kind); s_plo = bperm(srcl, s_plo);
                s_phi = bperm(srcl, s_phi); s_vlen = bperm(srcl, s_vlen);
                nseg -= s0;
            }
            M.out[lane] = s_out; M.kind[lane] = s_kind; M.plo[lane] = s_plo; M.phi[lane] = s_phi; M.vlen[lane] = s_vlen;
        }
    }
}

// ---- row plans: realign_tracks_kernel's planned walk over a WHOLE row, once: every entry of the row into a table in memory
// (at most PLAN_MAXE; 32-bit deltas), and per chunk of the row which of them reach into it.  The kernel's chunk-waves then read
// their entries instead of walking: BASELINE config 4 has 176 variants per row = 3 trips of 64, of which a chunk-wave ran 2.2 on
// average (wave scans, ballots, the records' loads: a quarter of the kernel's instructions) to find the 1.8 entries it
// needs.  One wave per row; a row the planned walk cannot express, with more entries than the table holds or with a delta
// beyond 32 bits is marked unplanned (count -1): its chunk-waves walk as before.
__global__ __launch_bounds__(256) void track_plan_kernel(const TrackArgs A, int2 *hdr, i32x4 *ent_all, const int chunks, const i64 fixed_len,
                                                          const i64 tl_bs) {
    __shared__ int outs[4][PLAN_MAXE];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = rfl((int)(threadIdx.x >> 6));
    const i64 k = (i64)blockIdx.x * 4 + wave;
    if (k >= A.n_rows) return;
    const i64 query = A.ploidy_shift >= 0 ? (k >> A.ploidy_shift) : (i64)((u32)k / (u32)A.ploidy);
    // (fixed_len >= 0: every row has that length -- the loader's epoch table has no row offsets)
    const i64 row_base = fixed_len >= 0 ? 0 : A.out_offsets[k], row_end = fixed_len >= 0 ? fixed_len : A.out_offsets[k + 1];
    const i64 q_start = A.regions[query * A.regions_stride + 1];
    const i64 shift = A.shifts[k];
    const i64 o_idx = A.geno_offset_idx[k];
    const bool has_keep = A.keep && A.keep_offsets;
    const i64 keep_off = has_keep ? A.keep_offsets[k] : 0;
    const i64 o_s = A.go_starts[o_idx];
    const i64 nv64 = A.go_stops[o_idx] - o_s;
    const i64 L64 = row_end - row_base;
    const int L = (int)L64;
    const int n_var = nv64 < 0 ? 0 : (nv64 > 0x7FFFFFFFll ? 0x7FFFFFFF : (int)nv64);
    // (tl_bs > 0: the loader's epoch table -- every batch of tl_bs queries has its own tl_bs + 1 offsets)
    const i64 tq = tl_bs > 0 ? query + query / tl_bs : query;
    const i64 tlen = A.track_offsets[tq + 1] - A.track_offsets[tq];
    i32x4 *const ent = ent_all + k * (i64)PLAN_MAXE;
    bool ok = q_start > -(1ll << 30) && q_start < (1ll << 30) && shift >= 0 && shift < (1ll << 30) && !(A.dbg & 8) && n_var > 0 &&
              L64 > 0 && L64 < 0x7FFFFF00ll;
    int rem = (int)(ok ? shift : 0);
    int tidx0 = 0, pm_carry = 0, x_carry = 0;
    bool ended = false;
    int tidx_end = 0, out_end = 0;
    int n_ent = 0;
    const int qs = (int)(ok ? q_start : 0);
    auto put = [&](int q, int kind, int o_start, i64 pval, int vlen) {
        if (q < PLAN_MAXE) {
            const i32x4 e = {o_start, kind, (int)pval, vlen};
            ent[q] = e;
            outs[wave][q] = o_start;
        }
    };
    for (int tb = 0; tb < n_var && ok && !ended; tb += WAVE) {
        int pos = 0, d = 0;
        bool valid = tb + lane < n_var;
        if (valid) {
            if (A.grec) {
                const i32x4 rec = *reinterpret_cast<const i32x4 *>(A.grec + (o_s + tb + lane));
                pos = rec.x; d = rec.y;
            } else {
                int v = A.geno_v_idxs[o_s + tb + lane];
                v = v < 0 ? 0 : ((i64)v >= A.n_variants ? (int)(A.n_variants - 1) : v);
                pos = A.v_starts[v]; d = A.ilens[v];
            }
            if (has_keep) valid = A.keep[keep_off + tb + lane] != 0;
        }
        const bool weird = valid && (pos <= -(1 << 30) || pos >= (1 << 30) || d <= -(1 << 30) || d >= (1 << 30));
        ok = ok && __builtin_amdgcn_ballot_w64(weird) == 0;
        const int vrp = pos - qs;
        const int E = vrp - (d < 0 ? d : 0) + 1;
        int v_len = (d > 0 ? d : 0) + 1;
        const u64 m_span = __builtin_amdgcn_ballot_w64(valid && d < 0 && vrp < 0 && E >= 0);
        if (m_span) { tidx0 = rdl(E, 63 - __builtin_clzll(m_span)); pm_carry = tidx0; tidx_end = tidx0; }
        bool cand = valid && vrp >= 0;
        if (rem > 0) {
            const int base = tidx0;
            const u64 m_t = __builtin_amdgcn_ballot_w64(cand && vrp >= base && (vrp - base) + v_len >= rem);
            if (m_t == 0) {
                cand = false;
            } else {
                const int f = __builtin_ctzll(m_t);
                const int dist = rdl(vrp, f) - base;
                if (dist >= rem) {
                    tidx0 = base + rem;
                    cand = cand && lane >= f;
                } else {
                    const int skip = rem - dist;
                    if (skip == rdl(v_len, f)) {
                        tidx0 = rdl(E, f);
                        cand = cand && lane > f;
                    } else {
                        tidx0 = rdl(vrp, f);
                        cand = cand && lane >= f;
                        if (lane == f) v_len -= skip;
                    }
                }
                rem = 0;
                pm_carry = tidx0; tidx_end = tidx0;
            }
        }
        cand = cand && d != 0;
        bool inB = cand;
        int PM = 0, pm_incl = 0;
        {
            u64 mB = __builtin_amdgcn_ballot_w64(inB);
            bool stable = false;
#pragma unroll 1
            for (int it = 0; it < 4 && !stable; ++it) {
                PM = wave_scan_exclusive<OpMaxU>(inB ? E : 0, pm_incl);
                PM = PM > pm_carry ? PM : pm_carry;
                inB = cand && vrp >= PM;
                const u64 m2 = __builtin_amdgcn_ballot_w64(inB);
                stable = m2 == mB;
                mB = m2;
            }
            ok = ok && stable;
        }
        const int n_i = inB ? vrp - PM : 0;
        const int S_i = inB ? OpSat::f(n_i, v_len) : 0;
        int x_incl;
        const int X = OpSat::f(x_carry, wave_scan_exclusive<OpSat>(S_i, x_incl));
        const int fill_out = OpSat::f(X, n_i);
        const bool applied = inB && fill_out < L;
        const int w_i = applied ? ((v_len < L - fill_out) ? v_len : L - fill_out) : 0;
        const u64 m_inB = __builtin_amdgcn_ballot_w64(inB);
        const u64 m_app = __builtin_amdgcn_ballot_w64(applied);
        if (m_inB != m_app) ended = true;
        if (m_app) {
            const int last = 63 - __builtin_clzll(m_app);
            tidx_end = rdl(E, last);
            out_end = rdl(fill_out, last) + rdl(w_i, last);
        }
        // (the whole row is "the chunk": every run in front of an applied variant, every fill)
        const bool e_trk = applied && n_i > 0 && fill_out > 0;
        const bool e_fil = applied && w_i > 0;
        const int slot0 = n_ent + wave_scan_exclusive<OpAdd>((e_trk ? 1 : 0) + (e_fil ? 1 : 0));
        const int add_ent = __builtin_popcountll(__builtin_amdgcn_ballot_w64(e_trk)) +
                            __builtin_popcountll(__builtin_amdgcn_ballot_w64(e_fil));
        if (n_ent + add_ent + 2 > PLAN_MAXE) ok = false;
        if (ok) {
            int q = slot0;
            if (e_trk) { put(q, T_TRACK, X, (i64)PM - X, 0); ++q; }
            // (an insertion is T_FILL here whatever the fill: the plan serves every track of the batch, and a track may have
            // its own strategy -- the reader turns it into T_REPEAT for Repeat5p)
            if (e_fil) put(q, d > 0 ? T_FILL : T_REPEAT, fill_out, (i64)vrp, v_len);
        }
        if (tb + WAVE < n_var) {
            const int mx = rdl(pm_incl, 63);
            pm_carry = mx > pm_carry ? mx : pm_carry;
            x_carry = OpSat::f(x_carry, rdl(x_incl, 63));
        }
        n_ent += add_ent;
    }
    if (ok) {      // src/tracks/mod.rs:365-392: the tail
        i64 t_idx = tidx_end;
        if (rem > 0) t_idx = imin((i64)tidx0 + rem, tlen);
        const int u = L - out_end;
        if (u > 0 && lane == 0) {
            const i64 avail = tlen - t_idx;
            const int w = (int)imin((i64)u, avail);
            int end = out_end;
            int q = n_ent;
            if (w > 0) {
                end += w;
                const i64 delta = t_idx - out_end;
                if (delta < -(1ll << 30) || delta > (1ll << 30)) q = PLAN_MAXE + 1;
                else { put(q, T_TRACK, out_end, delta, 0); ++q; }
            }
            if (end < L && q <= PLAN_MAXE) { put(q, T_ZERO, end, 0, 0); ++q; }
            n_ent = q;
        }
        n_ent = rfl(n_ent);
        if (n_ent > PLAN_MAXE) ok = false;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int2 *const h = hdr + k * (i64)chunks;
    for (int c = lane; c < chunks; c += WAVE) {
        int2 v = {0, -1};
        const i64 lo = (i64)c * A.chunk_len, hi = lo + A.chunk_len;
        if (ok && n_ent > 0 && lo < L64) {
            // entries are in output order, the first one starts at 0: [first, last] = the last that starts at or in front of
            // the chunk's first value ... the last that starts in front of its end
            int first = 0, last = 0;
            for (int i = 0; i < n_ent; ++i) {
                const i64 o = outs[wave][i];
                if (o <= lo) first = i;
                if (o < hi) last = i;
            }
            v.x = first; v.y = last - first + 1;
        }
        h[c] = v;
    }
}

// src/intervals.rs:19-126.  One thread per output value: the value is that of the LAST
// interval (in order) that covers the position -- what sequential painting leaves behind.
// `pmax[c]` = max(ends[list start .. c]): a position no earlier interval reaches is 0.0 without
// walking back over the whole list (gaps between intervals are the common case).
struct PaintTile { u32 idx[2 * WAVE]; float cv[PAINT_TILE]; int ce[PAINT_TILE]; };     // start bitmap + prefix, candidate values / ends
struct PaintImage { u32 idx[PAINT_CHUNK]; float cv[PAINT_TILE]; };                      // the leftovers kernel's image of one chunk
struct PaintTodo { int flag; int n_c; i64 lo_c; };      // per (query, chunk): 0 = done, 1 = per-value kernel, 2 = image (candidates [lo_c, lo_c + n_c))

// "Later intervals overwrite earlier ones" = every position takes the candidate with the HIGHEST index
// that covers it: one wave paints candidate indices into an LDS image of the chunk with ds_max (order-free;
// lane = interval for short ones, the whole wave for a long one) and streams the image out through the
// candidates' values.
__device__ __forceinline__ void paint_image(PaintImage &T, const int lane, const i64 lo_c, const int n_c, const i64 qs,
                                            const i64 j0, const int clen, const int *itv_starts, const int *itv_ends,
                                            const float *itv_values, float *row) {
    {   // clear the image
        const u32x4_a4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < PAINT_CHUNK / (4 * WAVE); ++t) *reinterpret_cast<u32x4_a4 *>(&T.idx[4 * (t * WAVE + lane)]) = z;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int b = 0; b < n_c; b += WAVE) {
        const int i = b + lane;
        int s_rel = 0, w = 0;
        if (i < n_c) {
            i64 s64 = (i64)itv_starts[lo_c + i] - qs - j0, e64 = (i64)itv_ends[lo_c + i] - qs - j0;
            s64 = s64 < 0 ? 0 : s64;
            e64 = e64 > clen ? clen : e64;
            if (e64 > s64) { s_rel = (int)s64; w = (int)(e64 - s64); }
            T.cv[i] = itv_values[lo_c + i];
        }
        const u32 tag = (u32)(i + 1);
        const bool is_long = w > 32;
        // short intervals: lane = interval
        for (int t = 0; __builtin_amdgcn_ballot_w64(!is_long && t < w) != 0; ++t)
            if (!is_long && t < w) atomicMax(&T.idx[s_rel + t], tag);
        // long intervals: the whole wave paints one at a time
        u64 m_long = __builtin_amdgcn_ballot_w64(is_long);
        while (m_long) {
            const int l = __builtin_ctzll(m_long);
            m_long &= m_long - 1;
            const int ls = rdl(s_rel, l), lw = rdl(w, l);
            const u32 lt = (u32)(b + l + 1);
            for (int t = lane; t < lw; t += WAVE) atomicMax(&T.idx[ls + t], lt);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < PAINT_CHUNK / TRIP; ++t) {
        const int p = t * TRIP + GROUP * lane;
        if (p < clen) {
            const u32x4_a4 ix = *reinterpret_cast<const u32x4_a4 *>(&T.idx[p]);
            const float v0 = ix.x ? T.cv[ix.x - 1] : 0.0f, v1 = ix.y ? T.cv[ix.y - 1] : 0.0f;
            const float v2 = ix.z ? T.cv[ix.z - 1] : 0.0f, v3 = ix.w ? T.cv[ix.w - 1] : 0.0f;
            if (p + GROUP <= clen) {
                store_f32x4_wb(row + p, v0, v1, v2, v3);
            } else {
                if (p < clen) row[p] = v0;
                if (p + 1 < clen) row[p + 1] = v1;
                if (p + 2 < clen) row[p + 2] = v2;
            }
        }
    }
}


// values [j_begin, j_end) (step j_step) of query q, one thread per value: binary search + walk-back, cut short by
// the running maximum of ends where nothing covers the position
__device__ __forceinline__ void paint_values(const i64 j_begin, const i64 j_end, const i64 j_step, const i64 s0, const i64 e0,
                                             const i64 qs, const int *itv_starts, const int *itv_ends, const float *itv_values,
                                             const int *pmax, float *row) {
    for (i64 j = j_begin; j < j_end; j += j_step) {
        // c = last interval with start - qs <= j  (intervals are sorted by start)
        i64 lo = s0, hi = e0;   // first interval with start - qs > j
        while (lo < hi) {
            const i64 mid = (lo + hi) >> 1;
            if ((i64)itv_starts[mid] - qs <= j) lo = mid + 1; else hi = mid;
        }
        float v = 0.0f;
        if (lo > s0 && (i64)pmax[lo - 1] - qs > j) {
            for (i64 c = lo - 1; c >= s0; --c) {
                if ((i64)itv_ends[c] - qs > j) { v = itv_values[c]; break; }
            }
        }
        __builtin_nontemporal_store(v, row + j);
    }
}

// Two modes.  chunk_todo == NULL: the whole painting, grid (x, n_queries), a grid-stride loop over each row.
// chunk_todo != NULL (after the tiled kernel): only the chunks it left behind.  Almost every chunk is done by then,
// so a workgroup looks at 256 records at once (one per thread, grid = records / 256: 33 workgroups for cfg4 instead
// of 8 320) and works through the few that are flagged: 1 = per-value painting by the whole workgroup, 2 =
// overlapping candidates, wave 0 paints the chunk into the LDS image.
__global__ __launch_bounds__(256) void intervals_to_tracks_kernel(
    const i64 *offset_idxs, const int *starts, i64 starts_stride, i64 n_queries, const int *itv_starts,
    const int *itv_ends, const float *itv_values, const i64 *itv_offsets, const int *pmax, float *out,
    const i64 *out_offsets, int chunk_len, const PaintTodo *chunk_todo, i64 n_chunks, i64 list_div) {
    __shared__ PaintImage image;
    __shared__ int flags[256];
    if (!chunk_todo) {
        const i64 q = blockIdx.y;
        if (q >= n_queries) return;
        const i64 o0 = out_offsets[q];
        const i64 idx = offset_idxs[q] / list_div;
        paint_values((i64)blockIdx.x * blockDim.x + threadIdx.x, out_offsets[q + 1] - o0, (i64)gridDim.x * blockDim.x, itv_offsets[idx],
                     itv_offsets[idx + 1], starts[q * starts_stride], itv_starts, itv_ends, itv_values, pmax, out + o0);
        return;
    }
    const i64 total = n_chunks * n_queries;
    const i64 rec0 = (i64)blockIdx.x * 256;
    const int mine = rec0 + threadIdx.x < total ? chunk_todo[rec0 + threadIdx.x].flag : 0;
    flags[threadIdx.x] = mine;
    if (!__syncthreads_or(mine != 0)) return;
    for (int r = 0; r < 256; ++r) {
        const int flag = flags[r];              // (uniform)
        if (flag == 0) continue;
        const i64 rec = rec0 + r;
        const i64 q = rec / n_chunks, chunk = rec - q * n_chunks;
        const PaintTodo td = chunk_todo[rec];
        const i64 o0 = out_offsets[q];
        const i64 length = out_offsets[q + 1] - o0;
        const i64 idx = offset_idxs[q] / list_div;
        const i64 qs = starts[q * starts_stride];
        const i64 c0 = chunk * chunk_len;
        const i64 c1 = c0 + chunk_len < length ? c0 + chunk_len : length;
        if (flag == 2) {
            if (threadIdx.x < WAVE)
                paint_image(image, (int)threadIdx.x, td.lo_c, td.n_c, qs, c0, (int)(c1 - c0), itv_starts, itv_ends, itv_values,
                            out + o0 + c0);
        } else {
            paint_values(c0 + threadIdx.x, c1, blockDim.x, itv_offsets[idx], itv_offsets[idx + 1], qs, itv_starts, itv_ends,
                         itv_values, pmax, out + o0);
        }
        __syncthreads();                        // the image is reused by the next flagged chunk
    }
}

// Tiled painter: one wave per (query, 2048-value chunk).  The intervals that can touch the chunk
// are [lo_c, hi_c): hi_c = first start at or after the chunk's end, lo_c = first interval whose
// running max of ends passes the chunk's start (two interleaved 64-ary searches, 2 rounds for
// lists of thousands).  "Later intervals overwrite earlier ones" = every position takes the
// candidate with the HIGHEST index that covers it, so the wave paints candidate indices into an
// LDS image of the chunk with ds_max (order-free; lane = interval for short ones, the whole wave
// for a long one) and then streams the image out through the candidates' values.  Chunks with
// more than PAINT_TILE candidates are left to the per-value kernel above.
// Coarse per-list index (gvl_intervals_bucket_*): bucket b of list i covers positions
// [base[i] + 2048 b, base[i] + 2048 (b + 1)); lo[] = first interval whose running max of ends passes
// the bucket's start, hi[] = first interval that starts at or after the bucket's end (both relative
// to the list's first interval).  One lookup per chunk gives a SUPERSET of the chunk's candidates
// (intervals outside the chunk clip to nothing), instead of two dependent 64-ary searches.

__global__ __launch_bounds__(256) void intervals_to_tracks_tiled_kernel(
    const i64 *offset_idxs, const int *starts, i64 starts_stride, i64 n_queries, const int *itv_starts,
    const int *itv_ends, const float *itv_values, const i64 *itv_offsets, const int *pmax, float *out,
    const i64 *out_offsets, int chunk_len, int n_chunks, PaintTodo *chunk_todo, const PaintIndex X, const int force_image,
    int *complete_err, const i64 list_div) {      // non-NULL: the caller vouched that no chunk needs the leftovers launch (gvl_track_set.tile_complete)
    __shared__ PaintTile tiles[4];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = rfl((int)(threadIdx.x >> 6));
    PaintTile &T = tiles[wave];
    __shared__ int probe_s[WAVE], probe_p[WAVE];
    const i64 q = blockIdx.y;
    const i64 chunk = (i64)blockIdx.x * 4 + wave;
    // (per-query values: scalar loads through the constant address space, see realign_tracks_kernel)
    typedef const int __attribute__((address_space(4))) *KInt;
    typedef const i64 __attribute__((address_space(4))) *KI64;
    // Three rounds, each requested as a whole before any of it is used (the asm statements name the values as scalar
    // operands: written value by value the compiler had put every load behind the use of the one before -- eight memory
    // round trips in a row in front of the first candidate record, and a chunk's wave is exactly that chain long):
    //   1. the query's output range, list number and first position,  2. the list's bounds and its bucket index's,
    //   3. (below) both bucket bounds;  then the candidate records.
    const i64 o0 = ((KI64)(u64)out_offsets)[q];
    const i64 o1 = ((KI64)(u64)out_offsets)[q + 1];
    const i64 idx_raw = ((KI64)(u64)offset_idxs)[q];
    const i64 qs = ((KInt)(u64)starts)[q * starts_stride];
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" :: "s"(o0), "s"(o1), "s"(idx_raw), "s"((int)qs));
#endif
    const i64 length = o1 - o0;
    const i64 idx = list_div == 1 ? idx_raw : idx_raw / list_div;
    const i64 s0 = ((KI64)(u64)itv_offsets)[idx], e0 = ((KI64)(u64)itv_offsets)[idx + 1];
    // (without a bucket index the three reads below land on the list's own bounds and are not used)
    const i64 *const xo = X.offsets ? X.offsets : itv_offsets;
    const i64 xb0 = ((KI64)(u64)xo)[idx], xb1 = ((KI64)(u64)xo)[idx + 1];
    const int xbase = ((KInt)(u64)(X.offsets && X.base ? X.base : (const int *)itv_offsets))[idx];
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" :: "s"(s0), "s"(e0), "s"(xb0), "s"(xb1), "s"(xbase));
#endif
    // the first round of both searches probes the same 64 strided entries of the query's list for
    // every chunk: wave 0 fetches them once for the block's 4 chunks
    if (!X.offsets && wave == 0 && e0 > s0) {
        const i64 st = (e0 - s0 + WAVE - 1) / WAVE;
        i64 pp = s0 + (i64)(lane + 1) * st - 1;
        if (pp > e0 - 1) pp = e0 - 1;
        probe_s[lane] = itv_starts[pp];
        probe_p[lane] = pmax[pp];
    }
    __syncthreads();
    if (chunk >= n_chunks) return;
    const i64 j0 = chunk * chunk_len;
    if (j0 >= length) { if (lane == 0 && chunk_todo) chunk_todo[q * n_chunks + chunk].flag = 0; return; }
    const i64 j1 = (length - j0 > chunk_len) ? j0 + chunk_len : length;
    // first start - qs >= j1 and first pmax - qs > j0: both searches advance together so that
    // their probe loads overlap (2 dependent rounds for lists of thousands instead of 4)
    i64 hi_c = 0, lo_c = 0;
    bool indexed = false;
    if (X.offsets) {
        const i64 b0 = xb0;
        const i64 nb = xb1 - b0;
        if (nb > 0) {
            const i64 base = xbase;
            i64 ba = (qs + j0 - base) >> 11, bb = (qs + j1 - 1 - base) >> 11;
            ba = ba < 0 ? 0 : (ba > nb - 1 ? nb - 1 : ba);
            bb = bb < 0 ? 0 : (bb > nb - 1 ? nb - 1 : bb);
            const int r_lo = ((KInt)(u64)X.lo)[b0 + ba], r_hi = ((KInt)(u64)X.hi)[b0 + bb];
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" :: "s"(r_lo), "s"(r_hi));
#endif
            lo_c = s0 + r_lo;
            hi_c = s0 + r_hi;
            if (lo_c > hi_c) lo_c = hi_c;
            indexed = hi_c - lo_c <= PAINT_TILE;
        }
    }
    if (!indexed) {
        i64 a1 = s0, b1 = e0, a2 = s0, b2 = e0;
        bool d1 = false, d2 = false;
        bool first = X.offsets == nullptr;      // (the shared first-round probes are only fetched without an index)
        while (!(d1 && d2)) {
            i64 p1 = 0, p2 = 0, st1 = 1, st2 = 1;
            int k1 = 0, k2 = 0;
            if (!d1) {
                if (a1 >= b1) d1 = true;
                else { st1 = (b1 - a1 + WAVE - 1) / WAVE; p1 = a1 + (i64)(lane + 1) * st1 - 1; if (p1 > b1 - 1) p1 = b1 - 1; k1 = first ? probe_s[lane] : itv_starts[p1]; }
            }
            if (!d2) {
                if (a2 >= b2) d2 = true;
                else { st2 = (b2 - a2 + WAVE - 1) / WAVE; p2 = a2 + (i64)(lane + 1) * st2 - 1; if (p2 > b2 - 1) p2 = b2 - 1; k2 = first ? probe_p[lane] : pmax[p2]; }
            }
            if (!d1) {
                const u64 m = __builtin_amdgcn_ballot_w64((i64)k1 - qs > j1 - 1);
                if (m == 0) { a1 = b1; d1 = true; }
                else {
                    const int f = __builtin_ctzll(m);
                    i64 pf = a1 + (i64)(f + 1) * st1 - 1; if (pf > b1 - 1) pf = b1 - 1;
                    if (st1 == 1) { b1 = pf; a1 = pf; d1 = true; }
                    else { if (f > 0) a1 += (i64)f * st1; b1 = pf; if (a1 >= b1) d1 = true; }
                }
            }
            if (!d2) {
                const u64 m = __builtin_amdgcn_ballot_w64((i64)k2 - qs > j0);
                if (m == 0) { a2 = b2; d2 = true; }
                else {
                    const int f = __builtin_ctzll(m);
                    i64 pf = a2 + (i64)(f + 1) * st2 - 1; if (pf > b2 - 1) pf = b2 - 1;
                    if (st2 == 1) { b2 = pf; a2 = pf; d2 = true; }
                    else { if (f > 0) a2 += (i64)f * st2; b2 = pf; if (a2 >= b2) d2 = true; }
                }
            }
            first = false;
        }
        hi_c = b1; lo_c = b2 < hi_c ? b2 : hi_c;
    }
    const i64 n_c64 = hi_c - lo_c;
    // (chunk_todo == NULL: a `tile_complete` interval set painted without the leftovers launch -- nobody reads the flags)
    PaintTodo *todo = chunk_todo ? chunk_todo + q * n_chunks + chunk : nullptr;
    if (n_c64 > PAINT_TILE) { if (lane == 0) { if (todo) todo->flag = 1; if (complete_err) *complete_err = 2; } return; }   // the per-value path takes it
    if (lane == 0 && todo) todo->flag = 0;
    const int n_c = (int)n_c64;
    const int clen = (int)(j1 - j0);
    float *row = out + o0 + j0;
    // ---- candidates that do not overlap (what a BigWig-like track is): no painting at all.  A start
    // BITMAP of the chunk (64 words) + its exclusive popcount prefix give, for any position, the number of
    // candidates that start at or before it -- i.e. the index of the only interval that can cover it --
    // with two LDS reads; its end says whether it does.  Overlapping candidates or equal starts take the
    // image path below (later intervals win: ds_max of candidate indices).
    {
        T.idx[lane] = 0u;                                     // bitmap: words 0..63, prefix: words 64..127
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bool bad = force_image != 0;
        int n_before = 0;
        int carry_e = (int)0x80000000, carry_s = (int)0x80000000;          // end / start of the candidate in front
        // all candidate records are requested before the first one is used (<= 4 rounds of 64: one memory
        // round trip instead of one per round)
        int c_s[PAINT_TILE / WAVE], c_e[PAINT_TILE / WAVE]; float c_v[PAINT_TILE / WAVE];
        // (every lane of every round reads SOME candidate -- those behind the last one the last one again -- and does not use
        // it: a load under a predicate, or in a round that is skipped, is merged with a default value, and the merge
        // waits for it: one round trip per round again)
#pragma unroll
        for (int r_ = 0; r_ < PAINT_TILE / WAVE; ++r_) { c_s[r_] = 0; c_e[r_] = 0; c_v[r_] = 0.0f; }
        if (n_c > 0) {
#pragma unroll
            for (int r_ = 0; r_ < PAINT_TILE / WAVE; ++r_) {
                const int i = r_ * WAVE + lane;
                const i64 at = lo_c + (i < n_c ? i : n_c - 1);
                c_s[r_] = itv_starts[at]; c_e[r_] = itv_ends[at]; c_v[r_] = itv_values[at];
            }
        }
#pragma unroll
        for (int r_ = 0; r_ < PAINT_TILE / WAVE; ++r_) {
            const int b = r_ * WAVE;
            if (b >= n_c) break;
            const int i = b + lane;
            int sr = 0x7FFFFFFF, er = 0x7FFFFFFF;
            if (i < n_c) {
                i64 s64 = (i64)c_s[r_] - qs - j0, e64 = (i64)c_e[r_] - qs - j0;
                s64 = s64 < -(1ll << 30) ? -(1ll << 30) : (s64 > (1ll << 30) ? (1ll << 30) : s64);
                e64 = e64 < -(1ll << 30) ? -(1ll << 30) : (e64 > (1ll << 30) ? (1ll << 30) : e64);
                sr = (int)s64; er = (int)e64;
                T.ce[i] = er;
                T.cv[i] = c_v[r_];
            }
            int pe = __shfl_up(er, 1, WAVE), ps = __shfl_up(sr, 1, WAVE);
            if (lane == 0) { pe = carry_e; ps = carry_s; }
            if (i < n_c && (sr < pe || sr == ps)) bad = true;
            if (i < n_c && sr >= 0 && sr < clen) atomicOr(&T.idx[sr >> 5], 1u << (sr & 31));
            n_before += __builtin_popcountll(__builtin_amdgcn_ballot_w64(i < n_c && sr < 0));
            const int last = (n_c - b > WAVE ? WAVE : n_c - b) - 1;
            carry_e = rdl(er, last); carry_s = rdl(sr, last);
        }
        if (__builtin_amdgcn_ballot_w64(bad) == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const u32 wbits = T.idx[lane];
            const int cnt = __builtin_popcount(wbits);
            const int incl = wave_scan_inclusive<OpAdd>(cnt);
            T.idx[WAVE + lane] = (u32)(incl - cnt);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int base = n_before - 1;
#pragma unroll
            for (int t = 0; t < PAINT_CHUNK / TRIP; ++t) {
                const int p = t * TRIP + GROUP * lane;
                if (p < clen) {
                    const u32 wd = T.idx[p >> 5];
                    const int pre = base + (int)T.idx[WAVE + (p >> 5)];
                    const int bp = p & 31;                      // (a multiple of 4: the 4 positions share the word)
                    float v[GROUP];
                    const int i0 = pre + __builtin_popcount(wd & (0xFFFFFFFFu >> (31 - bp)));
                    const int i3 = pre + __builtin_popcount(wd & (0xFFFFFFFFu >> (28 - bp)));
                    if (i0 == i3) {
                        const int e0_ = i0 >= 0 ? T.ce[i0] : 0;
                        const float c0 = i0 >= 0 ? T.cv[i0] : 0.0f;
#pragma unroll
                        for (int g = 0; g < GROUP; ++g) v[g] = e0_ > p + g ? c0 : 0.0f;
                    } else {
#pragma unroll
                        for (int g = 0; g < GROUP; ++g) {
                            const int ig = pre + __builtin_popcount(wd & (0xFFFFFFFFu >> (31 - bp - g)));
                            v[g] = (ig >= 0 && T.ce[ig] > p + g) ? T.cv[ig] : 0.0f;
                        }
                    }
                    if (p + GROUP <= clen) {
                        store_f32x4_wb(row + p, v[0], v[1], v[2], v[3]);
                    } else {
                        row[p] = v[0];
                        if (p + 1 < clen) row[p + 1] = v[1];
                        if (p + 2 < clen) row[p + 2] = v[2];
                    }
                }
            }
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // overlapping candidates or equal starts: the leftovers kernel paints this chunk into an LDS image
    if (lane == 0) { if (todo) { todo->flag = 2; todo->n_c = n_c; todo->lo_c = lo_c; } if (complete_err) *complete_err = 2; }
}

// bucket counts of every list (written at counts[i + 1] for the scan) and the list's base position
__global__ __launch_bounds__(256) void bucket_counts_kernel(const int *itv_starts, const i64 *itv_offsets, i64 n_lists,
                                                             i64 *counts, int *base) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) counts[0] = 0;
    if (i >= n_lists) return;
    const i64 s0 = itv_offsets[i], e0 = itv_offsets[i + 1];
    i64 n = 0;
    int b = 0;
    if (e0 > s0) {
        b = itv_starts[s0];
        n = ((i64)itv_starts[e0 - 1] - (i64)b) / PAINT_CHUNK + 1;
    }
    counts[i + 1] = n;
    base[i] = b;
}

__global__ __launch_bounds__(256) void bucket_fill_kernel(const int *itv_starts, const int *pmax, const i64 *itv_offsets,
                                                           i64 n_lists, const i64 *bkt_offsets, const int *base, int *lo_out,
                                                           int *hi_out) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= bkt_offsets[n_lists]) return;
    i64 a = 0, b = n_lists;                      // list i with bkt_offsets[i] <= g < bkt_offsets[i + 1]
    while (b - a > 1) {
        const i64 m = (a + b) >> 1;
        if (bkt_offsets[m] <= g) a = m; else b = m;
    }
    const i64 i = a;
    const i64 s0 = itv_offsets[i], e0 = itv_offsets[i + 1];
    const i64 x0 = (i64)base[i] + (g - bkt_offsets[i]) * PAINT_CHUNK, x1 = x0 + PAINT_CHUNK;
    i64 l = s0, h = e0;                          // first c with pmax[c] > x0
    while (l < h) { const i64 m = (l + h) >> 1; if ((i64)pmax[m] > x0) h = m; else l = m + 1; }
    lo_out[g] = (int)(l - s0);
    l = s0; h = e0;                              // first c with start >= x1
    while (l < h) { const i64 m = (l + h) >> 1; if ((i64)itv_starts[m] >= x1) h = m; else l = m + 1; }
    hi_out[g] = (int)(l - s0);
}

// pmax[c] = max(ends[s .. c]) within each queried list: one wave per list, 64 entries per trip
__global__ __launch_bounds__(256) void intervals_prefix_max_kernel(const i64 *list_idxs, i64 n_lists,
                                                                    const int *itv_ends, const i64 *itv_offsets,
                                                                    int *pmax) {
    const int lane = threadIdx.x & (WAVE - 1);
    for (i64 w = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_lists; w += ((i64)gridDim.x * blockDim.x) >> 6) {
        const i64 idx = list_idxs ? list_idxs[w] : w;
        const i64 s0 = itv_offsets[idx], e0 = itv_offsets[idx + 1];
        int carry = (int)0x80000000;
        for (i64 b = s0; b < e0; b += WAVE) {
            const i64 c = b + lane;
            int v = c < e0 ? itv_ends[c] : (int)0x80000000;
            // inclusive max scan (signed): bias to unsigned order for OpMaxU
            int u = wave_scan_inclusive<OpMaxU>((int)((u32)v ^ 0x80000000u));
            int m = (int)((u32)u ^ 0x80000000u);
            m = m > carry ? m : carry;
            if (c < e0) pmax[c] = m;
            carry = rdl(m, WAVE - 1);
        }
    }
}


// ---------------------------------------------------------------------------------
// Device-side request prep: one thread per query (see gvl_prepare_request in gvl_hip.h).
// ---------------------------------------------------------------------------------
struct PrepArgs {
    DiffArgs D;               // CSR + ilens / v_starts for the shift bound
    const i64 *idx; i64 batch; const int *full_regions; i64 n_regions; i64 n_samples; int ploidy;
    int jitter; int rc_neg; int deterministic; i64 output_length; u64 seed; u64 counter;
    int *regions; i64 *goi; u8 *to_rc; int *shifts;
};

__global__ __launch_bounds__(256) void prepare_request_kernel(const PrepArgs A) {
    // one thread per (query, haplotype): the haplotypes of a query repeat the query's few loads
    // (same cache lines) instead of walking their variants one after the other in one thread
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A.batch * A.ploidy) return;
    const i64 b = k / A.ploidy;
    const int p = (int)(k - b * A.ploidy);
    i64 id = A.idx[b];
    const i64 n = A.n_regions * A.n_samples;
    id = id < 0 ? 0 : (id >= n ? n - 1 : id);
    const i64 r = id / A.n_samples, s = id - r * A.n_samples;       // np.unravel_index(idx, (R, S))
    const int *src = A.full_regions + r * 4;
    int start = src[1], end = src[2];
    const int len = end - start;
    if (A.jitter > 0) {                                              // _query.py:166-171
        // keyed by the DATASET index (not the row's place in the batch): ranks / batches that hold
        // different samples draw independently, whoever holds sample `id` in this epoch draws the same
        const u64 h = hash4_dev(A.seed, A.counter, (u64)id, 0x6a69747465ull);
        start += (int)(h % (u64)(2 * A.jitter + 1)) - A.jitter;
        end = start + len;
    }
    if (p == 0) {
        int *dst = A.regions + b * 4;
        dst[0] = src[0]; dst[1] = start; dst[2] = end; dst[3] = src[3];
    }
    const u8 rc = (A.rc_neg && src[3] == -1) ? 1 : 0;               // _query.py:173-175
    const i64 goi = (r * A.n_samples + s) * A.ploidy + p;           // _haps.py:757-768
    A.goi[k] = goi;
    A.to_rc[k] = rc;
    int shift = 0;
    if (!A.deterministic) {                                         // _haps.py:723-730
        // length delta of this haplotype inside the (jittered) window: genotypes/mod.rs:48-85
        const i64 diff = (i64)(int)row_diff_core(A.D, goi, false, 0, true, (i64)start, (i64)end);
        const i64 max_shift = (diff > 0 ? diff : 0) + ((i64)len - A.output_length > 0 ? (i64)len - A.output_length : 0);
        const u64 h = hash4_dev(A.seed, A.counter, (u64)(id * A.ploidy + p), 0x7368696674ull);
        shift = (int)(h % (u64)(max_shift + 1));
    }
    A.shifts[k] = shift;
}

// ---------------------------------------------------------------------------
// host side of the C-ABI
// ---------------------------------------------------------------------------
thread_local char g_err[512] = "";
u64 *g_stamps = nullptr;

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

// Errors a kernel finds out about (asynchronous, like a sticky HIP error): one host-mapped word the
// device writes and gvl_async_error() reads.
// (the loader's producer thread and the caller's thread can both make the first call: once, and the pointer
// is published before either of them launches with it)
int *g_async_err = nullptr;
std::once_flag g_async_once;
int *async_err_word() {
    std::call_once(g_async_once, [] {
        void *p = nullptr;
        if (hipHostMalloc(&p, sizeof(int), hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && p) {
            *(volatile int *)p = 0;
            __atomic_store_n(&g_async_err, (int *)p, __ATOMIC_RELEASE);
        } else {
            (void)hipGetLastError();
        }
    });
    return __atomic_load_n(&g_async_err, __ATOMIC_ACQUIRE);
}

int pick_chunk(i64 max_len, int *chunks, int *chunk_len) {
    // one wave owns `chunk_len` bases of a row (blockIdx.y = chunk, so <= 65535 chunks);
    // rows up to 2048 bp are one chunk (= CHUNK_TRIPS trips, what the planned path handles)
    i64 cl = 2048;
    if (max_len <= 2048) cl = ((max_len + TRIP - 1) / TRIP) * TRIP;
    if (cl < TRIP) cl = TRIP;
    i64 c = (max_len + cl - 1) / cl;
    if (c > 65535) {
        cl = (((max_len + 65534) / 65535 + 2047) / 2048) * 2048;
        c = (max_len + cl - 1) / cl;
    }
    if (c < 1) c = 1;
    if (c > 65535 || cl > 0x7FFFFF00ll) return 1;
    *chunks = (int)c;
    *chunk_len = (int)cl;
    return 0;
}

// GVL_DBG (read once; gvl_set_debug_flags overrides it): test/diagnostic switches that remove
// one way a row can reach its output, so that the GPU suite can be run down every path:
//     8  every row through the scalar walk (haplotypes and tracks)
//    32  no scan-free plan for SNP-only rows (they join the packed plan)
//   512  no packable rows at all (every row runs the per-wave scans)
//    16  ignore gvl_static.geno_rec (records come from geno_v_idxs -> vrec)
//    64  ignore gvl_static.slot_rec (rows find their records through the CSR)
//   128  no speculative reference reads in front of the plan
//  1024  painter ignores the per-list bucket index (exact 64-ary searches per chunk)
//  2048  length deltas (get_diffs_sparse, ragged sizing) always one wave per row
//  8192  painter always paints an LDS image (no start-bitmap lookup for non-overlapping candidates)
// 16384  no lean kernel (the all-purpose kernel over every row, as before round 3)
// 32768  the lean kernel hands EVERY row to its solo general path (per-wave scans from the byte reference)
// 131072 the native loop sizes the scratch tracks per batch (not once per epoch)
// 262144 / 524288  timing ablations of the lean kernel (WRONG output for rows with indels): no re-alignment / allele
//        bytes (phases A and B as for a SNP-only row); no scan plan either
// 65536  the lean kernel re-reads the runs of a row with indels from memory (never re-aligns the speculative window in LDS)
// 1048576 rows longer than one chunk never take the lean kernel (LONG): the all-purpose kernel as before
// 2097152 realignment from intervals never uses its window (every value looked up in the interval list itself)
// 4194304 tracks are always painted into the scratch track first (no realignment straight from the intervals)
// 8388608 / 16777216  timing ablations of realign_tracks_kernel (NO output): stop behind the walk / behind the window build
// and 1 / 2 / 4 = timing ablations (no variants / no stores / no loads).
int g_debug_override = -1;
int debug_flags() {
    static const int flags = [] { const char *e = getenv("GVL_DBG"); return e ? atoi(e) : 0; }();
    return g_debug_override >= 0 ? g_debug_override : flags;
}

// Launch-policy overrides (gvl_set_tuning): A/B measurements, and tests that must force a schedule.  0 = the built-in policy.
// (Round 4's environment knobs GVL_PIPE_WAVES / _ROWS / _RPW_X100 / _MIN_ROWS, GVL_LEAN_SUB, GVL_*_EXTRA_LDS and
// GVL_TRACK_PLAN_MAX_MB are gone: their experiments concluded -- LABNOTES.md -- and what is left of them is this table.)
i64 g_tune[GVL_TUNE_COUNT] = {0};
i64 tune(int key) { return __atomic_load_n(&g_tune[key], __ATOMIC_RELAXED); }

int log2_exact(i64 v) {
    for (int s = 0; s < 31; ++s) if ((1ll << s) == v) return s;
    return -1;
}

// GVL_TRACE=1: report any HIP call of the loop that holds the host for more than 1 ms.
// GVL_TRACE=2: also sum the host time per call site; gvl_loader_destroy prints the averages.
int trace_level() { static const int lv = [] { const char *e = getenv("GVL_TRACE"); return e ? atoi(e) : 0; }(); return lv; }
bool trace_on() { return trace_level() != 0; }
double now_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
struct TraceSite { const char *tag; double sum_ms; long n; };
TraceSite g_sites[16];
int g_nsites = 0;
std::mutex g_sites_mu;          // (the producer thread and the caller both report under GVL_TRACE=2)
void trace_add(const char *tag, double dt) {
    std::lock_guard<std::mutex> lk(g_sites_mu);
    for (int i = 0; i < g_nsites; ++i) if (g_sites[i].tag == tag) { g_sites[i].sum_ms += dt; ++g_sites[i].n; return; }
    if (g_nsites < 16) g_sites[g_nsites++] = TraceSite{tag, dt, 1};
}
void trace_report() {
    if (trace_level() < 2) return;
    std::lock_guard<std::mutex> lk(g_sites_mu);
    for (int i = 0; i < g_nsites; ++i)
        fprintf(stderr, "[gvl trace] %-24s %8ld calls  %7.2f us each\n", g_sites[i].tag, g_sites[i].n,
                1e3 * g_sites[i].sum_ms / (double)g_sites[i].n);
    g_nsites = 0;
}
template <typename F> hipError_t traced(const char *tag, F f) {
    if (!trace_on()) return f();
    const double t0 = now_ms();
    const hipError_t e = f();
    const double dt = now_ms() - t0;
    if (dt > 1.0) fprintf(stderr, "[gvl trace] %s held the host for %.1f ms\n", tag, dt);
    if (trace_level() >= 2) trace_add(tag, dt);
    return e;
}
}  // namespace

extern "C" {

int gvl_abi_version(void) { return GVL_ABI_VERSION; }
int gvl_set_debug_flags(int flags) { g_debug_override = flags; return GVL_OK; }
int gvl_set_tuning(int32_t key, int64_t value) {
    if (key < 0 || key >= GVL_TUNE_COUNT) return fail(GVL_ERR_INVALID, "%s", "gvl_set_tuning: unknown key");
    __atomic_store_n(&g_tune[key], value < 0 ? 0 : value, __ATOMIC_RELAXED);
    return GVL_OK;
}
// diagnostics (not in gvl_hip.h): a device buffer the kernels may leave counters / time stamps in.  Phase stamps need a
// -DGVL_DIAG build; lean_solo_rows counts the rows and waves that reach it in words 0 and 1 in every build (tools/pipe_deferred.py).
void gvl_diag_set_stamps(void *buf) { g_stamps = (u64 *)buf; }
const char *gvl_last_error(void) { return g_err; }

int gvl_async_error(int clear) {
    int *w = __atomic_load_n(&g_async_err, __ATOMIC_ACQUIRE);
    if (!w) return GVL_OK;
    const int e = *(volatile int *)w;
    if (clear) *(volatile int *)w = 0;
    if (e == 1) return fail(GVL_ERR_INVALID, "%s", "a launch found a row longer than its batch's max_row_len hint: that row was left partly unwritten");
    if (e == 2) return fail(GVL_ERR_INVALID, "%s", "an interval set marked tile_complete has a chunk with overlapping intervals, equal starts or more than 256 candidates: that chunk was left unpainted");
    if (e == 3) return fail(GVL_ERR_INVALID, "%s", "a pipelined launch gave a wave more rows than it can take (grid mis-sized): the surplus rows were left unwritten");
    return e ? fail(GVL_ERR_INVALID, "%s", "asynchronous device-side error") : GVL_OK;
}

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

int gvl_pack_genotypes(const gvl_static *st, gvl_grec *grec_out, void *stream) {
    if (!st || st->n_geno < 0 || st->n_variants < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_pack_genotypes: bad arguments");
    if (st->n_geno == 0) return GVL_OK;
    if (!st->geno_v_idxs || !st->vrec || !grec_out || st->n_variants == 0)
        return fail(GVL_ERR_INVALID, "%s", "gvl_pack_genotypes: NULL array (vrec from gvl_pack_variants is required)");
    const i64 grid = (st->n_geno + 255) / 256;
    if (grid > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_pack_genotypes: too many entries");
    pack_genotypes_kernel<<<dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream>>>(st->geno_v_idxs, st->n_geno, st->vrec,
                                                                                       st->n_variants, grec_out);
    return check_launch("gvl_pack_genotypes");
}

int64_t gvl_ref4_bytes(int64_t ref_len) { return ref_len < 0 ? 0 : (ref_len + 1) / 2 + GVL_REF4_PAD; }

int gvl_pack_reference(const uint8_t *ref, int64_t ref_len, uint8_t *ref4_out, void *stream) {
    if (ref_len < 0 || !ref4_out || (ref_len > 0 && !ref)) return fail(GVL_ERR_INVALID, "%s", "gvl_pack_reference: bad arguments");
    const i64 out_len = gvl_ref4_bytes(ref_len);
    const i64 grid = ((out_len + 3) / 4 + 255) / 256;
    if (grid > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_pack_reference: reference too long");
    pack_ref4_kernel<<<dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream>>>(ref, ref_len, ref4_out, out_len);
    return check_launch("gvl_pack_reference");
}

// ---- per-dataset arrays for callers without a device allocator of their own -----------------------
struct StaticOwner { gvl_static st; void *bufs[16]; int n; };

static void *dev_copy(StaticOwner *o, const void *host, size_t bytes, hipStream_t s, bool *ok) {
    void *d = nullptr;
    if (!*ok) return nullptr;
    if (hipMalloc(&d, bytes ? bytes : 16) != hipSuccess) { (void)hipGetLastError(); *ok = false; return nullptr; }
    o->bufs[o->n++] = d;
    if (bytes && host && hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, s) != hipSuccess) { (void)hipGetLastError(); *ok = false; }
    return d;
}

int gvl_static_upload(const gvl_static *host, int32_t with_layouts, gvl_static **out, void *stream) {
    if (!host || !out) return fail(GVL_ERR_INVALID, "%s", "gvl_static_upload: NULL argument");
    if (host->ref_len < 0 || host->n_contigs < 0 || host->n_variants < 0 || host->alt_len < 0 || host->n_geno_offsets < 0 || host->n_geno < 0)
        return fail(GVL_ERR_INVALID, "%s", "gvl_static_upload: negative size");
    if (!host->ref_offsets || !host->geno_o_starts || !host->geno_o_stops || (host->ref_len > 0 && !host->ref) ||
        (host->n_variants > 0 && (!host->v_starts || !host->ilens || !host->alt_offsets)) || (host->n_geno > 0 && !host->geno_v_idxs))
        return fail(GVL_ERR_INVALID, "%s", "gvl_static_upload: NULL host array");
    StaticOwner *o = new (std::nothrow) StaticOwner;
    if (!o) return fail(GVL_ERR_HIP, "%s", "gvl_static_upload: out of host memory");
    memset(o, 0, sizeof(*o));
    hipStream_t s = (hipStream_t)stream;
    bool ok = true;
    gvl_static &d = o->st;
    d = *host;
    const i64 nv = host->n_variants, ng = host->n_geno, no = host->n_geno_offsets;
    d.ref = (const uint8_t *)dev_copy(o, host->ref, (size_t)host->ref_len, s, &ok);
    d.ref_offsets = (const int64_t *)dev_copy(o, host->ref_offsets, (size_t)(host->n_contigs + 1) * 8, s, &ok);
    d.v_starts = (const int32_t *)dev_copy(o, host->v_starts, (size_t)nv * 4, s, &ok);
    d.ilens = (const int32_t *)dev_copy(o, host->ilens, (size_t)nv * 4, s, &ok);
    d.alt_offsets = (const int64_t *)dev_copy(o, host->alt_offsets, (size_t)(nv + 1) * 8, s, &ok);
    d.alt_alleles = (const uint8_t *)dev_copy(o, host->alt_alleles, (size_t)host->alt_len, s, &ok);
    d.geno_o_starts = (const int64_t *)dev_copy(o, host->geno_o_starts, (size_t)no * 8, s, &ok);
    d.geno_o_stops = (const int64_t *)dev_copy(o, host->geno_o_stops, (size_t)no * 8, s, &ok);
    d.geno_v_idxs = (const int32_t *)dev_copy(o, host->geno_v_idxs, (size_t)ng * 4, s, &ok);
    d.vrec = (const gvl_vrec *)dev_copy(o, nullptr, (size_t)(nv > 0 ? nv : 1) * sizeof(gvl_vrec), s, &ok);
    d.geno_rec = nullptr; d.slot_rec = nullptr;
    int rc = ok ? GVL_OK : fail(GVL_ERR_HIP, "%s", "gvl_static_upload: device allocation / copy failed");
    if (!rc && nv > 0) rc = gvl_pack_variants(d.v_starts, d.ilens, d.alt_offsets, d.alt_alleles, nv, (gvl_vrec *)d.vrec, stream);
    if (!rc && with_layouts && ng > 0 && nv > 0) {
        gvl_grec *g = (gvl_grec *)dev_copy(o, nullptr, (size_t)ng * sizeof(gvl_grec), s, &ok);
        if (ok) { rc = gvl_pack_genotypes(&d, g, stream); if (!rc) d.geno_rec = g; }
    }
    if (!rc && ok && with_layouts && no > 0 && nv > 0 && host->alt_len < (1ll << 32)) {
        gvl_srec *sr = (gvl_srec *)dev_copy(o, nullptr, (size_t)no * GVL_SLOT_RECS * sizeof(gvl_srec), s, &ok);
        if (ok) { rc = gvl_pack_slots(&d, sr, stream); if (!rc) d.slot_rec = sr; }
    }
    d.ref4 = nullptr;
    if (!rc && ok && with_layouts && host->ref_len > 0) {
        uint8_t *r4 = (uint8_t *)dev_copy(o, nullptr, (size_t)gvl_ref4_bytes(host->ref_len), s, &ok);
        if (ok) { rc = gvl_pack_reference(d.ref, host->ref_len, r4, stream); if (!rc) d.ref4 = r4; }
    }
    if (!rc && !ok) rc = fail(GVL_ERR_HIP, "%s", "gvl_static_upload: device allocation failed");
    if (rc) { gvl_static_free(&o->st); return rc; }
    *out = &o->st;
    return GVL_OK;
}

int gvl_static_free(gvl_static *st) {
    if (!st) return GVL_OK;
    StaticOwner *o = reinterpret_cast<StaticOwner *>(st);       // `st` is the first member
    (void)hipDeviceSynchronize();
    for (int i = 0; i < o->n; ++i) (void)hipFree(o->bufs[i]);
    delete o;
    return GVL_OK;
}

int gvl_pack_slots(const gvl_static *st, gvl_srec *srec_out, void *stream) {
    if (!st || st->n_geno_offsets < 0 || st->n_variants < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_pack_slots: bad arguments");
    if (st->n_geno_offsets == 0) return GVL_OK;
    if (st->alt_len >= (1ll << 32)) return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_pack_slots: alt_alleles of 4 GiB or more (use the CSR path)");
    if (!st->geno_o_starts || !st->geno_o_stops || !srec_out || (st->n_geno > 0 && (!st->geno_v_idxs || !st->vrec || !st->alt_offsets || st->n_variants == 0)))
        return fail(GVL_ERR_INVALID, "%s", "gvl_pack_slots: NULL array (vrec from gvl_pack_variants is required)");
    const i64 grid = (st->n_geno_offsets * GVL_SLOT_RECS + 255) / 256;
    if (grid > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_pack_slots: too many slots");
    pack_slots_kernel<<<dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream>>>(
        (const i64 *)st->geno_o_starts, (const i64 *)st->geno_o_stops, st->n_geno_offsets, st->geno_v_idxs, st->vrec,
        (const i64 *)st->alt_offsets, st->n_variants, srec_out);
    return check_launch("gvl_pack_slots");
}

// validate one batch and fill its kernel arguments; `variant` = which template instance it needs
static int fill_recon_args(const gvl_static *st, const gvl_batch *bt, const gvl_out *out, ReconArgs &A, int *chunks,
                           int *variant) {
    if (!st || !bt || !out) return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: NULL struct");
    if (bt->batch < 0 || bt->ploidy <= 0) return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: bad batch/ploidy");
    if (!out->haps && !out->onehot) return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: no output buffer");
    if ((st->ref_len > 0 && !st->ref) || !st->ref_offsets || !st->geno_o_starts || !st->geno_o_stops)
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: NULL static array");
    if (st->n_geno > 0 && (!st->vrec || !st->alt_offsets || !st->geno_v_idxs))
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: NULL variant table (vrec from gvl_pack_variants is required)");
    if (bt->batch > 0 && (!bt->regions || !bt->shifts || !bt->geno_offset_idx || bt->regions_stride < 3))
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: NULL/invalid batch array");
    if (bt->output_length < 0 && !bt->out_offsets)
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: ragged mode needs out_offsets (gvl_hap_offsets)");
    if (bt->output_length > 0x7FFFFF00ll || bt->max_row_len > 0x7FFFFF00ll)
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: row length must be < 2^31 - 256");
    if (out->onehot && out->onehot_layout == GVL_ONEHOT_CL && (bt->output_length < 0 || bt->out_offsets))
        return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_reconstruct: channel-major one-hot needs fixed-length rows");
    if (out->onehot_layout != GVL_ONEHOT_LC && out->onehot_layout != GVL_ONEHOT_CL)
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: bad onehot_layout");
    if (bt->batch * bt->ploidy > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: batch too large for one launch");

    memset(&A, 0, sizeof(A));
    A.ref = st->ref; A.ref_len = st->ref_len; A.ref_offsets = (const i64 *)st->ref_offsets;
    A.vrec = st->vrec; A.alt_offsets = (const i64 *)st->alt_offsets; A.alt_alleles = st->alt_alleles;
    A.alt_len = st->alt_len; A.n_variants = st->n_variants;
    A.go_starts = (const i64 *)st->geno_o_starts; A.go_stops = (const i64 *)st->geno_o_stops;
    A.geno_v_idxs = st->geno_v_idxs;
    A.grec = (debug_flags() & 16) ? nullptr : st->geno_rec;
    A.srec = (debug_flags() & (64 | 512 | 8)) ? nullptr : st->slot_rec;
    A.ref4 = st->ref4;
    A.n_geno_offsets = st->n_geno_offsets;
    A.n_contigs = (int)(st->n_contigs < 0 ? 0 : (st->n_contigs > 0x7FFFFFFFll ? 0x7FFFFFFF : st->n_contigs));
    A.regions = bt->regions; A.regions_stride = bt->regions_stride; A.shifts = bt->shifts;
    A.geno_offset_idx = (const i64 *)bt->geno_offset_idx;
    A.keep = bt->keep; A.keep_offsets = (const i64 *)bt->keep_offsets; A.to_rc = bt->to_rc;
    A.out_offsets = (const i64 *)bt->out_offsets;
    A.fixed_len = bt->out_offsets ? -1 : bt->output_length;
    A.n_rows = bt->batch * bt->ploidy; A.ploidy = (int)bt->ploidy; A.ploidy_shift = log2_exact(bt->ploidy);
    // longest row: fixed mode -> output_length; caller-supplied offsets -> the
    // caller's max_row_len hint (must bound every row, or longer rows are left
    // partly unwritten)
    i64 ml = bt->out_offsets ? bt->max_row_len : bt->output_length;
    if (bt->out_offsets && bt->output_length > ml) ml = bt->output_length;
    if (ml < 0) ml = 0;
    if (pick_chunk(ml, chunks, &A.chunk_len)) return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: too many chunks");
    A.ref_only = 0;
    A.dbg = debug_flags();
    A.pad = st->pad_char;
    A.haps = out->haps; A.onehot = out->onehot;
    A.av = out->annot_v_idxs; A.ap = out->annot_ref_pos; A.out_offsets_w = (i64 *)out->out_offsets;
    A.stamps = g_stamps;
    A.async_err = async_err_word();
    const bool annot = out->annot_v_idxs || out->annot_ref_pos;
    const int oh = !out->onehot ? OH_NONE : (out->onehot_layout == GVL_ONEHOT_CL ? OH_CL : OH_LC);
    *variant = oh | ((out->haps != nullptr) ? 4 : 0) | (annot ? 8 : 0);
    return GVL_OK;
}

static int launch_recon(const ReconArgs &A, int chunks, int variant, void *stream) {
    const i64 grid = (A.n_rows + WG_WAVES - 1) / WG_WAVES;
    if (grid <= 0) return GVL_OK;
    recon_fn fn = recon_table(variant & 3, (variant & 4) != 0, (variant & 8) != 0);
    fn<<<dim3((unsigned)grid, (unsigned)chunks), dim3(WG_THREADS), 0, (hipStream_t)stream>>>(A);
    return check_launch("gvl_reconstruct");
}

// Can this batch take the lean kernel?  Row-major one-hot and / or haplotype bytes, fixed-length rows, no keep mask, no
// annotations, the derived layouts present -- and no path-forcing debug flag (those exist to walk the all-purpose
// kernel).  Rows of one chunk: one wave per row, variants from the slot lines.  Longer rows (a multiple of 4 bases, cut
// into 2048-base chunks): one wave per chunk, variants from the CSR's inline records (LONG).
static bool lean_eligible(const gvl_static *st, const gvl_batch *bt, const gvl_out *out, int chunks, int chunk_len) {
    if (!st->ref4 || (!out->onehot && !out->haps)) return false;
    if (out->annot_v_idxs || out->annot_ref_pos || (out->onehot && out->onehot_layout != GVL_ONEHOT_LC)) return false;
    if (bt->out_offsets || bt->keep || bt->keep_offsets) return false;
    if (bt->output_length <= 0 || (bt->output_length & 3)) return false;
    const i64 n_rows = bt->batch * bt->ploidy;
    if (chunks == 1) {
        if (!st->slot_rec || bt->output_length > LEAN_MAX_TRIPS * TRIP) return false;
        if (n_rows <= 0 || n_rows > 0x7FFFFFF0ll) return false;     // (row indices are ints in the kernel)
    } else {
        if (!st->geno_rec || chunk_len != LEAN_MAX_TRIPS * TRIP || (debug_flags() & 1048576)) return false;
        if (n_rows <= 0 || n_rows * chunks > 0x7FFFFFF0ll) return false;                  // (wave indices are ints in the kernel)
    }
    if (st->alt_len >= (1ll << 32) || st->ref_len >= (1ll << 32) - 8192) return false;     // u32 positions in the kernel
    // (2097152 ... 16777216 concern the track kernels only)
    return (debug_flags() & ~(2 | 4 | 32768 | 65536 | 262144 | 524288 | 1048576 | 2097152 | 4194304 | 8388608 | 16777216 | 33554432 | 67108864 | 268435456)) == 0;
}

// ragged rows (out_offsets) longer than the pipelined form's 2560 bases: the chunked lean kernel's ragged form (<.., LONG, RAGL>)
static bool lean_long_rag_eligible(const gvl_static *st, const gvl_batch *bt, const gvl_out *out, int chunks, int chunk_len) {
    if (!st->ref4 || !st->geno_rec || (!out->onehot && !out->haps) || !bt->out_offsets) return false;
    if (out->annot_v_idxs || out->annot_ref_pos || (out->onehot && out->onehot_layout != GVL_ONEHOT_LC)) return false;
    if (bt->keep || bt->keep_offsets) return false;
    if (chunks < 2 || chunk_len != LEAN_MAX_TRIPS * TRIP || (debug_flags() & (1048576 | 16))) return false;
    const i64 n_rows = bt->batch * bt->ploidy;
    if (n_rows <= 0 || n_rows * chunks > 0x7FFFFFF0ll) return false;
    if (st->alt_len >= (1ll << 32) || st->ref_len >= (1ll << 32) - 8192) return false;
    return (debug_flags() & ~(2 | 4 | 32768 | 65536 | 262144 | 524288 | 1048576 | 2097152 | 4194304 | 8388608 | 16777216 | 33554432 | 67108864 | 268435456)) == 0;
}

static int launch_lean(const ReconArgs &RA, int chunks, void *stream) {
    LeanArgs A;
    memset(&A, 0, sizeof(A));
    A.ref4 = RA.ref4; A.ref_offsets = RA.ref_offsets; A.srec = RA.srec;
    A.regions = RA.regions; A.shifts = RA.shifts; A.geno_offset_idx = RA.geno_offset_idx; A.to_rc = RA.to_rc;
    A.onehot = RA.onehot; A.haps = RA.haps; A.out_offsets_w = RA.out_offsets_w; A.alt_alleles = RA.alt_alleles; A.stamps = RA.stamps;
    A.go_starts = RA.go_starts; A.go_stops = RA.go_stops; A.grec = RA.grec; A.alt_offsets = RA.alt_offsets;
    A.n_geno_offsets = RA.n_geno_offsets;
    A.n_rows = (int)RA.n_rows; A.n_contigs = RA.n_contigs; A.regions_stride = (int)RA.regions_stride;
    A.ploidy_shift = RA.ploidy_shift; A.ploidy = RA.ploidy; A.L = (int)RA.fixed_len; A.dbg = RA.dbg;
    A.chunks = chunks;
    // rows of several chunks: a wave takes `sub` consecutive chunks, the second and later ones resume the first one's walk.
    // 2 by default -- BASELINE config 4's 256 rows x 64 chunks are then 8 192 waves, every wave slot of the part once;
    // gvl_set_tuning(GVL_TUNE_LEAN_SUB) overrides (1 = every chunk its own wave and its own walk)
    const i64 sub_t = tune(GVL_TUNE_LEAN_SUB);
    A.sub = chunks > 1 ? (sub_t > 0 ? (int)(sub_t > 64 ? 64 : sub_t) : 2) : 1;
    const i64 per_row = (chunks + A.sub - 1) / A.sub;
    const unsigned grid = (unsigned)(((i64)A.n_rows * per_row + LEAN_WAVES - 1) / LEAN_WAVES);
    const dim3 g(grid), b(LEAN_THREADS);
    hipStream_t s = (hipStream_t)stream;
    const unsigned xl = 0;
    if (chunks > 1 && RA.out_offsets) {         // ragged long rows (lean_long_rag_eligible)
        A.out_offsets = RA.out_offsets;
        A.out_offsets_w = nullptr;
        A.L = 0;
        if (A.onehot && A.haps) recon_lean_kernel<true, true, true, true><<<g, b, 0, s>>>(A, RA);
        else if (A.onehot) recon_lean_kernel<true, false, true, true><<<g, b, 0, s>>>(A, RA);
        else recon_lean_kernel<false, true, true, true><<<g, b, 0, s>>>(A, RA);
    } else if (chunks > 1) {
        if (A.onehot && A.haps) recon_lean_kernel<true, true, true><<<g, b, 0, s>>>(A, RA);
        else if (A.onehot) recon_lean_kernel<true, false, true><<<g, b, 0, s>>>(A, RA);
        else recon_lean_kernel<false, true, true><<<g, b, 0, s>>>(A, RA);
    } else {
        if (A.onehot && A.haps) recon_lean_kernel<true, true, false><<<g, b, xl, s>>>(A, RA);
        else if (A.onehot) recon_lean_kernel<true, false, false><<<g, b, xl, s>>>(A, RA);
        else recon_lean_kernel<false, true, false><<<g, b, 0, s>>>(A, RA);
    }
    return check_launch("gvl_reconstruct (lean)");
}

// ---- the pipelined form (gvl_lean_pipe.inc): rows of one chunk, `n` batches of the same shape in ONE grid ----------
// Launches with fewer than 8192 rows keep recon_lean_kernel (a wave per row: with one row per wave there is nothing to
// pipeline; gvl_set_tuning(GVL_TUNE_PIPE_MIN_ROWS) overrides); GVL_DBG & 33554432: always, with as few workgroups as 32 rows
// per wave allow (the suite's small batches then run many rows per wave on ONE workgroup); GVL_DBG & 67108864: never.
// RAG: rows at out_offsets (ragged output, output_length = -1, or a caller's plan): row-major one-hot and / or bytes, no keep
// mask, no annotations, the slot-major records and the packed reference present, and the caller's bound on the longest row
// within the pipelined kernel's 10 trips.  There is no wave-per-row lean kernel for these: pipelined form or the all-purpose kernel.
static bool lean_rag_eligible(const gvl_static *st, const gvl_batch *bt, const gvl_out *out) {
    if (!st->ref4 || !st->slot_rec || (!out->onehot && !out->haps) || !bt->out_offsets) return false;
    if (out->annot_v_idxs || out->annot_ref_pos || (out->onehot && out->onehot_layout != GVL_ONEHOT_LC)) return false;
    if (bt->keep || bt->keep_offsets) return false;
    const i64 ml = bt->max_row_len > bt->output_length ? bt->max_row_len : bt->output_length;
    if (ml <= 0 || ml > (i64)PipeCfg<true>::MAXT * TRIP) return false;
    const i64 n_rows = bt->batch * bt->ploidy;
    if (n_rows <= 0 || n_rows > 0x7FFFFFF0ll) return false;
    if (st->alt_len >= (1ll << 32) || st->ref_len >= (1ll << 32) - 8192) return false;
    return (debug_flags() & ~(2 | 4 | 32768 | 65536 | 262144 | 524288 | 1048576 | 2097152 | 4194304 | 8388608 | 16777216 | 33554432 | 268435456)) == 0;
}
static bool lean_pipe_wanted(i64 total_rows, int n_batches = 1) {
    if (debug_flags() & 67108864) return false;
    if (debug_flags() & 33554432) return true;
    const i64 min_t = tune(GVL_TUNE_PIPE_MIN_ROWS);
    const i64 min_rows = min_t > 0 ? min_t : 8192;
    // (a GROUP of small batches -- strong scaling: 4096 / 8 = 512 rows per rank and batch -- is one grid from 2048 rows on: ten
    // launches of 512 rows are ten launch latencies in a row)
    const i64 bar = (n_batches >= 2 && min_rows > 2048) ? 2048 : min_rows;
    return total_rows >= bar;
}
// can these (lean-eligible, one-chunk) batches share a grid?  the same shape and outputs; every batch but the last has the
// first one's row count
static bool lean_pipe_compatible(const ReconArgs *RAs, int n) {
    const ReconArgs &F = RAs[0];
    if (((uintptr_t)F.ref4 & 15) || ((uintptr_t)F.srec & 15)) return false;        // (16-byte DMA sources)
    i64 total = 0;
    for (int i = 0; i < n; ++i) {
        const ReconArgs &R = RAs[i];
        if (R.fixed_len != F.fixed_len || R.ploidy != F.ploidy || R.regions_stride != F.regions_stride ||
            (R.out_offsets != nullptr) != (F.out_offsets != nullptr) ||
            (R.onehot != nullptr) != (F.onehot != nullptr) || (R.haps != nullptr) != (F.haps != nullptr) || R.dbg != F.dbg)
            return false;
        if (R.n_rows <= 0 || R.n_rows > F.n_rows || (i + 1 < n && R.n_rows != F.n_rows)) return false;
        total += R.n_rows;
    }
    return total <= 0x7FFFFFF0ll && F.regions_stride <= 0x7FFFFFFFll;
}
static int launch_lean_rows(const ReconArgs *RAs, int n, void *stream, int rag_chunks = 1) {
    const ReconArgs &RA = RAs[0];
    LeanArgs A;
    LeanMany M;
    memset(&A, 0, sizeof(A));
    memset(&M, 0, sizeof(M));
    A.ref4 = RA.ref4; A.ref_offsets = RA.ref_offsets; A.srec = RA.srec;
    A.regions = RA.regions; A.shifts = RA.shifts; A.geno_offset_idx = RA.geno_offset_idx; A.to_rc = RA.to_rc;
    A.onehot = RA.onehot; A.haps = RA.haps; A.out_offsets_w = RA.out_offsets_w; A.alt_alleles = RA.alt_alleles;
    A.go_starts = RA.go_starts; A.go_stops = RA.go_stops; A.grec = RA.grec; A.alt_offsets = RA.alt_offsets;
    A.n_geno_offsets = RA.n_geno_offsets;
    A.n_contigs = RA.n_contigs; A.regions_stride = (int)RA.regions_stride;
    A.ploidy_shift = RA.ploidy_shift; A.ploidy = RA.ploidy; A.L = (int)RA.fixed_len; A.dbg = RA.dbg;
    A.chunks = 1; A.sub = 1;
    A.rows_per_batch = (int)RA.n_rows; A.n_batches = n;
    A.max_row_len = (int)((i64)RA.chunk_len * (RA.out_offsets ? rag_chunks : 1));
    i64 total = 0;
    for (int i = 0; i < n; ++i) {
        LeanBatch &b = M.b[i];
        b.regions = RAs[i].regions; b.shifts = RAs[i].shifts; b.geno_offset_idx = RAs[i].geno_offset_idx; b.to_rc = RAs[i].to_rc;
        b.onehot = RAs[i].onehot; b.haps = RAs[i].haps; b.out_offsets_w = RAs[i].out_offsets_w; b.n_rows = RAs[i].n_rows;
        b.out_offsets = RAs[i].out_offsets;
        total += RAs[i].n_rows;
    }
    A.n_rows = (int)total;
    // Rows per wave (x 100).  Measured (profiles/r04_pipe_experiments.txt G, K): ONE row per wave -- no row-to-row prefetch at
    // all -- is the best schedule up to ~12 batches per launch (short waves: the hardware's workgroup dispatch balances the chip);
    // above that "two rows per wave", which is 1.5 on average: the first half of the waves take two rows (w, w + W), the second
    // half -- dispatched last -- one, so the grid drains in short waves (125 / 175 measure like 150; exactly 2, or 3, are slower).
    // gvl_set_tuning(GVL_TUNE_PIPE_ROWS_X100) overrides (200 = exactly two rows for every wave, 300 = three, ...).
    i64 x100 = tune(GVL_TUNE_PIPE_ROWS_X100);
    if (x100 < 100) x100 = total >= 49152 ? 150 : 100;
    if (x100 > 100 * (i64)PIPE_MAX_ROWS) x100 = 100 * (i64)PIPE_MAX_ROWS;      // (a wave's deferred-rows mask has a bit per row)
    i64 waves = (total * 100 + x100 - 1) / x100;
    if (debug_flags() & 33554432) {
        i64 rpw = (total + LEAN_WAVES - 1) / LEAN_WAVES;
        rpw = rpw > PIPE_MAX_ROWS ? PIPE_MAX_ROWS : (rpw < 1 ? 1 : rpw);
        waves = (total + rpw - 1) / rpw;
    }
    const unsigned grid = (unsigned)((waves + LEAN_WAVES - 1) / LEAN_WAVES);
    const dim3 g(grid), b(LEAN_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (RA.out_offsets) {
        if (A.onehot && A.haps) recon_lean_rows_kernel<true, true, true><<<g, b, 0, s>>>(A, RA, M);
        else if (A.onehot) recon_lean_rows_kernel<true, false, true><<<g, b, 0, s>>>(A, RA, M);
        else recon_lean_rows_kernel<false, true, true><<<g, b, 0, s>>>(A, RA, M);
    } else {
        if (A.onehot && A.haps) recon_lean_rows_kernel<true, true, false><<<g, b, 0, s>>>(A, RA, M);
        else if (A.onehot) recon_lean_rows_kernel<true, false, false><<<g, b, 0, s>>>(A, RA, M);
        else recon_lean_rows_kernel<false, true, false><<<g, b, 0, s>>>(A, RA, M);
    }
    return check_launch("gvl_reconstruct (lean, pipelined)");
}

int gvl_reconstruct(const gvl_static *st, const gvl_batch *bt, const gvl_out *out, void *stream) {
    ReconArgs A;
    int chunks = 1, variant = 0;
    const int rc = fill_recon_args(st, bt, out, A, &chunks, &variant);
    if (rc) return rc;
    if (A.n_rows > 0 && lean_eligible(st, bt, out, chunks, A.chunk_len)) {
        if (chunks == 1 && lean_pipe_wanted(A.n_rows) && lean_pipe_compatible(&A, 1)) return launch_lean_rows(&A, 1, stream);
        return launch_lean(A, chunks, stream);
    }
    if (A.n_rows > 0 && !(debug_flags() & 67108864) && lean_rag_eligible(st, bt, out) && lean_pipe_compatible(&A, 1))
        return launch_lean_rows(&A, 1, stream, chunks);
    if (A.n_rows > 0 && lean_long_rag_eligible(st, bt, out, chunks, A.chunk_len)) return launch_lean(A, chunks, stream);
    return launch_recon(A, chunks, variant, stream);
}

// `n` batches in one host call.  Batches the lean kernel's pipelined form can take together (one-chunk rows, the same
// shape and outputs) are ONE grid: row k of the launch belongs to batch k / rows_per_batch, a wave takes rows w, w + W, ...
// and keeps its next row's reads in flight under the stores of the row in hand (gvl_lean_pipe.inc) -- which is what makes
// one grid better than launches on separate streams.  (Round 2 measured blockIdx.z = batch for the all-purpose kernel: the
// per-batch arguments behind an index cost 1.2-1.8 us per batch THERE, at their use; here they are fetched a row ahead.)
// Everything else: back-to-back launches on `stream`, every batch validated before the first launch.
int gvl_reconstruct_many(const gvl_static *st, const gvl_batch *bts, const gvl_out *outs, int32_t n, void *stream) {
    if (n < 0 || n > GVL_MANY_MAX || (n > 0 && (!bts || !outs))) return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct_many: bad arguments (n <= GVL_MANY_MAX)");
    ReconArgs A[GVL_MANY_MAX];
    int chunks[GVL_MANY_MAX], variant[GVL_MANY_MAX];
    bool lean[GVL_MANY_MAX];
    bool all_one_chunk_lean = n > 0, all_rag = n > 0 && !(debug_flags() & 67108864);
    i64 total = 0;
    for (int i = 0; i < n; ++i) {
        chunks[i] = 1; variant[i] = 0;
        const int rc = fill_recon_args(st, &bts[i], &outs[i], A[i], &chunks[i], &variant[i]);
        if (rc) return rc;
        lean[i] = A[i].n_rows > 0 && (lean_eligible(st, &bts[i], &outs[i], chunks[i], A[i].chunk_len) ||
                                      (!lean_rag_eligible(st, &bts[i], &outs[i]) && lean_long_rag_eligible(st, &bts[i], &outs[i], chunks[i], A[i].chunk_len)));
        all_one_chunk_lean = all_one_chunk_lean && lean[i] && chunks[i] == 1;
        all_rag = all_rag && A[i].n_rows > 0 && lean_rag_eligible(st, &bts[i], &outs[i]);
        total += A[i].n_rows;
    }
    if (all_one_chunk_lean && lean_pipe_wanted(total, n) && lean_pipe_compatible(A, n)) return launch_lean_rows(A, n, stream);
    if (all_rag && lean_pipe_compatible(A, n)) {
        int min_chunks = chunks[0];             // (the launch reports a row longer than the smallest bound any of its batches gave)
        for (int i = 1; i < n; ++i) min_chunks = chunks[i] < min_chunks ? chunks[i] : min_chunks;
        return launch_lean_rows(A, n, stream, min_chunks);
    }
    for (int i = 0; i < n; ++i) {
        const int rc = lean[i] ? launch_lean(A[i], chunks[i], stream) : launch_recon(A[i], chunks[i], variant[i], stream);
        if (rc) return rc;
    }
    return GVL_OK;
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
    A.n_contigs = (int)(st->n_contigs < 0 ? 0 : (st->n_contigs > 0x7FFFFFFFll ? 0x7FFFFFFF : st->n_contigs));
    A.to_rc = to_rc; A.out_offsets = (const i64 *)out_offsets; A.fixed_len = -1;
    A.n_rows = n_rows; A.ploidy = 1; A.ploidy_shift = 0;
    int chunks = 1;
    if (pick_chunk(max_row_len, &chunks, &A.chunk_len)) return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference: too many chunks");
    A.ref_only = 1;
    A.pad = st->pad_char;
    A.haps = out; A.onehot = onehot;
    const i64 grid = (n_rows + WG_WAVES - 1) / WG_WAVES;
    if (n_rows > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference: batch too large");
    recon_fn fn = recon_table(onehot ? OH_LC : OH_NONE, out != nullptr, false);
    fn<<<dim3((unsigned)grid, (unsigned)chunks), dim3(WG_THREADS), 0, (hipStream_t)stream>>>(A);
    return check_launch("gvl_get_reference");
}

// rows with many variants (mean > 16 per genotype slot, or GVL_DBG=2048): one wave per row instead of one lane
static bool diffs_long_rows(const gvl_static *st) {
    if (debug_flags() & 2048) return true;
    return st->n_geno_offsets > 0 && st->n_geno / st->n_geno_offsets > 16;
}

static int fill_diff_args(DiffArgs &D, const gvl_static *st, const gvl_batch *bt, const char *who) {
    if (!st || !bt) return fail(GVL_ERR_INVALID, "%s: NULL struct", who);
    if (bt->batch < 0 || bt->ploidy <= 0) return fail(GVL_ERR_INVALID, "%s: bad batch/ploidy", who);
    memset(&D, 0, sizeof(D));
    D.geno_offset_idx = (const i64 *)bt->geno_offset_idx; D.n_rows = bt->batch * bt->ploidy;
    D.ploidy = (int)bt->ploidy;
    D.geno_v_idxs = st->geno_v_idxs; D.go_starts = (const i64 *)st->geno_o_starts;
    D.grec = (debug_flags() & 16) ? nullptr : st->geno_rec;
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
    if (diffs_long_rows(st)) {
        const i64 grid = (D.n_rows * WAVE + 255) / 256;
        if (grid > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_get_diffs_sparse: batch too large");
        hipLaunchKernelGGL(diffs_wave_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, D, (const int *)nullptr, (i64)0);
    } else {
        const unsigned grid = (unsigned)((D.n_rows + 255) / 256);
        hipLaunchKernelGGL(diffs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, D, (const int *)nullptr, (i64)0);
    }
    return check_launch("gvl_get_diffs_sparse");
}

static int hap_offsets_impl(const gvl_static *st, const gvl_batch *bt, int32_t *diffs, int64_t *out_offsets,
                            int64_t *total_and_max, int64_t len_cap, void *stream);
int gvl_hap_offsets(const gvl_static *st, const gvl_batch *bt, int32_t *diffs, int64_t *out_offsets,
                    int64_t *total_and_max, void *stream) {
    return hap_offsets_impl(st, bt, diffs, out_offsets, total_and_max, 0, stream);
}
static int hap_offsets_impl(const gvl_static *st, const gvl_batch *bt, int32_t *diffs, int64_t *out_offsets,
                            int64_t *total_and_max, int64_t len_cap, void *stream) {
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
    if (len_cap > 0) { D.len_cap = len_cap; D.async_err = async_err_word(); }
    if (diffs_long_rows(st)) {
        const i64 grid = (D.n_rows * WAVE + 255) / 256;
        if (grid > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_hap_offsets: batch too large");
        hipLaunchKernelGGL(diffs_wave_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, D, bt->regions, (i64)bt->regions_stride);
    } else {
        const unsigned grid = (unsigned)((D.n_rows + 255) / 256);
        hipLaunchKernelGGL(diffs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, D, bt->regions, (i64)bt->regions_stride);
    }
    rc = check_launch("gvl_hap_offsets(diffs)");
    if (rc) return rc;
    hipLaunchKernelGGL(offsets_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (i64 *)out_offsets, D.n_rows, (i64 *)total_and_max);
    return check_launch("gvl_hap_offsets(scan)");
}

int gvl_keep_offsets(const gvl_static *st, const int64_t *geno_offset_idx, int64_t batch, int64_t ploidy,
                     int64_t *keep_offsets, int64_t *total_and_max, void *stream) {
    if (!st || batch < 0 || ploidy <= 0) return fail(GVL_ERR_INVALID, "%s", "gvl_keep_offsets: bad arguments");
    if (!keep_offsets) return fail(GVL_ERR_INVALID, "%s", "gvl_keep_offsets: NULL array");
    const i64 n_rows = batch * ploidy;
    if (n_rows > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_keep_offsets: batch too large");
    if (n_rows > 0 && (!geno_offset_idx || !st->geno_o_starts || !st->geno_o_stops))
        return fail(GVL_ERR_INVALID, "%s", "gvl_keep_offsets: NULL array");
    const unsigned grid = (unsigned)((n_rows + 1 + 255) / 256);
    keep_counts_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>((const i64 *)geno_offset_idx, (const i64 *)st->geno_o_starts,
                                                                          (const i64 *)st->geno_o_stops, n_rows, (i64 *)keep_offsets);
    int rc = check_launch("gvl_keep_offsets(counts)");
    if (rc) return rc;
    offsets_scan_kernel<<<dim3(1), dim3(1024), 0, (hipStream_t)stream>>>((i64 *)keep_offsets, n_rows, (i64 *)total_and_max);
    return check_launch("gvl_keep_offsets(scan)");
}

int gvl_choose_exonic_variants(const gvl_static *st, const int32_t *starts, const int32_t *ends,
                               const int64_t *geno_offset_idx, int64_t batch, int64_t ploidy,
                               const int64_t *keep_offsets, uint8_t *keep, void *stream) {
    if (!st || batch < 0 || ploidy <= 0) return fail(GVL_ERR_INVALID, "%s", "gvl_choose_exonic_variants: bad arguments");
    const i64 n_rows = batch * ploidy;
    if (n_rows == 0) return GVL_OK;
    if (n_rows > 0x3FFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_choose_exonic_variants: batch too large");
    if (!starts || !ends || !geno_offset_idx || !keep_offsets || !st->geno_o_starts || !st->geno_o_stops)
        return fail(GVL_ERR_INVALID, "%s", "gvl_choose_exonic_variants: NULL array");
    if (st->n_geno > 0 && (!keep || !st->geno_v_idxs || !st->v_starts || !st->ilens || st->n_variants <= 0))
        return fail(GVL_ERR_INVALID, "%s", "gvl_choose_exonic_variants: NULL variant table / keep");
    if (st->n_geno == 0) return GVL_OK;
    const unsigned grid = (unsigned)((n_rows + 3) / 4);
    exonic_keep_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(
        starts, ends, (const i64 *)geno_offset_idx, n_rows, (int)ploidy, st->geno_v_idxs, (const i64 *)st->geno_o_starts,
        (const i64 *)st->geno_o_stops, st->v_starts, st->ilens, st->n_variants, (const i64 *)keep_offsets, keep);
    return check_launch("gvl_choose_exonic_variants");
}

int gvl_rc_rows(uint8_t *data, const int64_t *offsets, const uint8_t *to_rc, int64_t n_rows, void *stream) {
    if (n_rows < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_rc_rows: n_rows < 0");
    if (n_rows == 0) return GVL_OK;
    if (!data || !offsets || !to_rc) return fail(GVL_ERR_INVALID, "%s", "gvl_rc_rows: NULL array");
    if (n_rows > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_rc_rows: too many rows");
    hipLaunchKernelGGL(rc_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, data, (const i64 *)offsets, to_rc, (i64)n_rows);
    return check_launch("gvl_rc_rows");
}

int gvl_rc_bounded_rows(uint8_t *data, const int64_t *bounds, const uint8_t *to_rc, int64_t n_rows, void *stream) {
    if (n_rows < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_rc_bounded_rows: n_rows < 0");
    if (n_rows == 0) return GVL_OK;
    if (!data || !bounds || !to_rc) return fail(GVL_ERR_INVALID, "%s", "gvl_rc_bounded_rows: NULL array");
    if (n_rows > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_rc_bounded_rows: too many rows");
    hipLaunchKernelGGL(rc_bounded_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, data, (const i64 *)bounds, to_rc, (i64)n_rows);
    return check_launch("gvl_rc_bounded_rows");
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


int gvl_intervals_prefix_max(const int32_t *itv_ends, const int64_t *itv_offsets, int64_t n_lists,
                             int32_t *pmax_out, void *stream) {
    if (n_lists < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_intervals_prefix_max: negative size");
    if (n_lists == 0) return GVL_OK;
    if (!itv_ends || !itv_offsets || !pmax_out) return fail(GVL_ERR_INVALID, "%s", "gvl_intervals_prefix_max: NULL array");
    i64 grid = (n_lists + 3) / 4;
    if (grid > 65535) grid = 65535;
    intervals_prefix_max_kernel<<<dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream>>>(
        nullptr, (i64)n_lists, itv_ends, (const i64 *)itv_offsets, pmax_out);
    return check_launch("gvl_intervals_prefix_max");
}

int gvl_intervals_bucket_counts(const int32_t *itv_starts, const int64_t *itv_offsets, int64_t n_lists,
                                int64_t *bkt_offsets, int32_t *bkt_base, int64_t *total, void *stream) {
    if (n_lists < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_intervals_bucket_counts: negative size");
    if (!itv_offsets || !bkt_offsets || !bkt_base || (n_lists > 0 && !itv_starts && false))
        return fail(GVL_ERR_INVALID, "%s", "gvl_intervals_bucket_counts: NULL array");
    hipStream_t s = (hipStream_t)stream;
    const i64 grid = (n_lists + 1 + 255) / 256;
    bucket_counts_kernel<<<dim3((unsigned)grid), dim3(256), 0, s>>>(itv_starts, (const i64 *)itv_offsets, (i64)n_lists,
                                                                   (i64 *)bkt_offsets, bkt_base);
    int rc = check_launch("gvl_intervals_bucket_counts");
    if (rc) return rc;
    offsets_scan_kernel<<<dim3(1), dim3(1024), 0, s>>>((i64 *)bkt_offsets, (i64)n_lists, (i64 *)total);
    return check_launch("gvl_intervals_bucket_counts(scan)");
}

int gvl_intervals_bucket_fill(const int32_t *itv_starts, const int32_t *itv_pmax_ends, const int64_t *itv_offsets,
                              int64_t n_lists, const int64_t *bkt_offsets, const int32_t *bkt_base, int64_t n_buckets,
                              int32_t *bkt_lo, int32_t *bkt_hi, void *stream) {
    if (n_lists < 0 || n_buckets < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_intervals_bucket_fill: negative size");
    if (n_buckets == 0) return GVL_OK;
    if (!itv_starts || !itv_pmax_ends || !itv_offsets || !bkt_offsets || !bkt_base || !bkt_lo || !bkt_hi)
        return fail(GVL_ERR_INVALID, "%s", "gvl_intervals_bucket_fill: NULL array");
    const i64 grid = (n_buckets + 255) / 256;
    if (grid > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_intervals_bucket_fill: too many buckets");
    bucket_fill_kernel<<<dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream>>>(
        itv_starts, itv_pmax_ends, (const i64 *)itv_offsets, (i64)n_lists, (const i64 *)bkt_offsets, bkt_base, bkt_lo, bkt_hi);
    return check_launch("gvl_intervals_bucket_fill");
}

// paint launches; `todo` (n_queries * n_chunks bytes, nullable) selects the tiled kernel + the per-value
// kernel for the chunks it leaves, NULL the per-value kernel alone
static bool paint_can_tile(const int32_t *pmax, int64_t max_row_len);
static int paint_launch(const int64_t *offset_idxs, const int32_t *starts, int64_t starts_stride, int64_t n_queries,
                        const int32_t *itv_starts, const int32_t *itv_ends, const float *itv_values,
                        const int64_t *itv_offsets, const int32_t *itv_pmax_ends, float *out, const int64_t *out_offsets,
                        int64_t max_row_len, PaintTodo *todo, hipStream_t s, const PaintIndex X = PaintIndex{nullptr, nullptr, nullptr, nullptr},
                        bool tile_complete = false, i64 list_div = 1) {
    if (list_div < 1) list_div = 1;
    const int chunk_len = 2048;
    const i64 n_chunks = (max_row_len + chunk_len - 1) / chunk_len;
    const bool complete_no_flags = !todo && tile_complete && X.offsets && !(debug_flags() & (8192 | 1024)) && paint_can_tile(itv_pmax_ends, max_row_len);
    if (todo || complete_no_flags) {
        // tile_complete: the interval set's owner vouches that the tiled kernel finishes every chunk (no overlaps, no equal
        // starts, at most 256 candidates in any two adjacent index buckets), so the leftovers launch -- 5.8 us that find
        // nothing -- is skipped; a chunk that needs it after all is reported through gvl_async_error, never silently wrong
        const bool complete = tile_complete && X.offsets && !(debug_flags() & (8192 | 1024));
        intervals_to_tracks_tiled_kernel<<<dim3((unsigned)((n_chunks + 3) / 4), (unsigned)n_queries), dim3(256), 0, s>>>(
            (const i64 *)offset_idxs, starts, (i64)starts_stride, (i64)n_queries, itv_starts, itv_ends, itv_values,
            (const i64 *)itv_offsets, itv_pmax_ends, out, (const i64 *)out_offsets, chunk_len, (int)n_chunks, todo, X,
            (debug_flags() & 8192) ? 1 : 0, complete ? async_err_word() : nullptr, list_div);
        if (complete) return check_launch("gvl_intervals_to_tracks");
        intervals_to_tracks_kernel<<<dim3((unsigned)((n_chunks * n_queries + 255) / 256)), dim3(256), 0, s>>>(
            (const i64 *)offset_idxs, starts, (i64)starts_stride, (i64)n_queries, itv_starts, itv_ends, itv_values,
            (const i64 *)itv_offsets, itv_pmax_ends, out, (const i64 *)out_offsets, chunk_len, todo, n_chunks, list_div);
    } else {
        i64 gx = (max_row_len + 255) / 256;
        if (gx > 1024) gx = 1024;
        intervals_to_tracks_kernel<<<dim3((unsigned)gx, (unsigned)n_queries), dim3(256), 0, s>>>(
            (const i64 *)offset_idxs, starts, (i64)starts_stride, (i64)n_queries, itv_starts, itv_ends, itv_values,
            (const i64 *)itv_offsets, itv_pmax_ends, out, (const i64 *)out_offsets, chunk_len, nullptr, (i64)0, list_div);
    }
    return check_launch("gvl_intervals_to_tracks");
}
// The painter's stream-ordered scratch comes from a pool the LIBRARY owns, one per device (created at the first use on that device):
// with the device's default pool every call paid a driver allocation (the default release threshold is 0: 25 of the 45 us of a
// stand-alone painting of BASELINE config 4's batch), and raising THAT pool's threshold would change the allocator for every other
// hipMallocAsync user of the process.  The library's pools keep up to 256 MiB each across synchronisations.
static hipError_t paint_alloc(void **p, size_t bytes, hipStream_t s) {
    static std::mutex mu;
    static hipMemPool_t pools[64] = {nullptr};
    static bool tried[64] = {false};
    int dev = 0;
    hipMemPool_t pool = nullptr;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        std::lock_guard<std::mutex> lk(mu);
        if (!tried[dev]) {
            tried[dev] = true;
            hipMemPoolProps props;
            memset(&props, 0, sizeof(props));
            props.allocType = hipMemAllocationTypePinned;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            hipMemPool_t np = nullptr;
            if (hipMemPoolCreate(&np, &props) == hipSuccess && np) {
                uint64_t thr = 256ull << 20;
                (void)hipMemPoolSetAttribute(np, hipMemPoolAttrReleaseThreshold, &thr);
                pools[dev] = np;
            }
            (void)hipGetLastError();
        }
        pool = pools[dev];
    }
    if (pool) return hipMallocFromPoolAsync(p, bytes, pool, s);
    return hipMallocAsync(p, bytes, s);          // (no pool of our own: the device's default pool, untouched)
}
static bool paint_can_tile(const int32_t *pmax, int64_t max_row_len) {
    return pmax && max_row_len < 0x7FFFFF00ll && (max_row_len + 2047) / 2048 <= 0x7FFFFFFFll / 4;
}

int gvl_intervals_to_tracks(const int64_t *offset_idxs, const int32_t *starts, int64_t starts_stride,
                            int64_t n_queries, const int32_t *itv_starts, const int32_t *itv_ends,
                            const float *itv_values, const int64_t *itv_offsets, int64_t n_intervals,
                            const int32_t *itv_pmax_ends, float *out, const int64_t *out_offsets,
                            int64_t max_row_len, void *stream) {
    if (n_queries < 0 || max_row_len < 0 || n_intervals < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_intervals_to_tracks: negative size");
    if (n_queries == 0 || max_row_len == 0) return GVL_OK;
    if (!offset_idxs || !starts || !itv_offsets || !out || !out_offsets || starts_stride < 1)
        return fail(GVL_ERR_INVALID, "%s", "gvl_intervals_to_tracks: NULL/invalid array");
    if (n_intervals > 0 && (!itv_starts || !itv_ends || !itv_values))
        return fail(GVL_ERR_INVALID, "%s", "gvl_intervals_to_tracks: NULL interval array");
    if (n_queries > 65535) return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_intervals_to_tracks: more than 65535 queries per call");
    hipStream_t s = (hipStream_t)stream;
    int *scratch = nullptr;
    if (!itv_pmax_ends && n_intervals > 0) {
        // no precomputed prefix maxima: build them for the queried lists in stream-ordered scratch
        if (paint_alloc((void **)&scratch, (size_t)n_intervals * sizeof(int), s) != hipSuccess) {
            (void)hipGetLastError();
            return fail(GVL_ERR_HIP, "%s", "gvl_intervals_to_tracks: scratch allocation failed (pass itv_pmax_ends)");
        }
        i64 grid = (n_queries + 3) / 4;
        intervals_prefix_max_kernel<<<dim3((unsigned)grid), dim3(256), 0, s>>>((const i64 *)offset_idxs, (i64)n_queries, itv_ends,
                                                                             (const i64 *)itv_offsets, scratch);
        itv_pmax_ends = scratch;
    }
    // tiled pass (rows shorter than 2^31, every list has its prefix maxima), then the per-value
    // kernel for the chunks it left (more than PAINT_TILE candidate intervals) -- or for everything
    // when the flag scratch cannot be had
    const i64 n_chunks = (max_row_len + 2047) / 2048;
    PaintTodo *todo = nullptr;
    if (paint_can_tile(itv_pmax_ends, max_row_len) &&
        paint_alloc((void **)&todo, (size_t)(n_queries * n_chunks) * sizeof(PaintTodo), s) != hipSuccess) {
        (void)hipGetLastError();
        todo = nullptr;
    }
    const int rc = paint_launch(offset_idxs, starts, starts_stride, n_queries, itv_starts, itv_ends, itv_values, itv_offsets,
                                itv_pmax_ends, out, out_offsets, max_row_len, todo, s);
    if (todo) (void)hipFreeAsync(todo, s);
    if (scratch) (void)hipFreeAsync(scratch, s);
    return rc;
}

// The painter over an interval set that carries its derived arrays (gvl_track_set: prefix maxima + the coarse bucket index): the
// tiled + bitmap path gvl_tracks_batch uses, for callers of the reference's two-call entry points (intervals_to_tracks, then
// shift_and_realign_tracks_sparse) -- gvl_intervals_to_tracks has no place for the index and paints 0.17 of the HBM peak.
int gvl_paint_tracks(const gvl_track_set *ts, const int64_t *offset_idxs, const int32_t *starts, int64_t starts_stride,
                     int64_t n_queries, float *out, const int64_t *out_offsets, int64_t max_row_len, void *stream) {
    if (!ts || n_queries < 0 || max_row_len < 0 || ts->n_intervals < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_paint_tracks: bad arguments");
    if (n_queries == 0 || max_row_len == 0) return GVL_OK;
    if (!offset_idxs || !starts || !ts->itv_offsets || !out || !out_offsets || starts_stride < 1)
        return fail(GVL_ERR_INVALID, "%s", "gvl_paint_tracks: NULL/invalid array");
    if (ts->n_intervals > 0 && (!ts->itv_starts || !ts->itv_ends || !ts->itv_values))
        return fail(GVL_ERR_INVALID, "%s", "gvl_paint_tracks: NULL interval array");
    if (n_queries > 65535) return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_paint_tracks: more than 65535 queries per call");
    if (!ts->itv_pmax_ends)      // (no derived arrays: the plain entry builds what it needs)
        return gvl_intervals_to_tracks(offset_idxs, starts, starts_stride, n_queries, ts->itv_starts, ts->itv_ends, ts->itv_values,
                                       ts->itv_offsets, ts->n_intervals, nullptr, out, out_offsets, max_row_len, stream);
    hipStream_t s = (hipStream_t)stream;
    PaintIndex X{nullptr, nullptr, nullptr, nullptr};
    if (ts->bkt_offsets && ts->bkt_base && ts->bkt_lo && ts->bkt_hi && !(debug_flags() & 1024))
        X = PaintIndex{(const i64 *)ts->bkt_offsets, ts->bkt_base, ts->bkt_lo, ts->bkt_hi};
    const i64 n_chunks = (max_row_len + 2047) / 2048;
    PaintTodo *todo = nullptr;
    // (a tile_complete set with its index needs no flags and no second launch: no scratch allocation either -- the stream-ordered
    // malloc + free pair cost more than the painting itself)
    const bool complete = ts->tile_complete != 0 && X.offsets && !(debug_flags() & (8192 | 1024));
    if (!complete && paint_can_tile(ts->itv_pmax_ends, max_row_len) &&
        paint_alloc((void **)&todo, (size_t)(n_queries * n_chunks) * sizeof(PaintTodo), s) != hipSuccess) {
        (void)hipGetLastError();
        todo = nullptr;
    }
    const int rc = paint_launch(offset_idxs, starts, starts_stride, n_queries, ts->itv_starts, ts->itv_ends, ts->itv_values,
                                ts->itv_offsets, ts->itv_pmax_ends, out, out_offsets, max_row_len, todo, s, X,
                                ts->tile_complete != 0, ts->list_div);
    if (todo) (void)hipFreeAsync(todo, s);
    return rc;
}

static int realign_tracks_impl(const gvl_static *st, const gvl_batch *bt, const float *tracks,
                               const int64_t *track_offsets, const double *params, int64_t strategy_id,
                               uint64_t base_seed, const u64 *seed_ptr, float *out, void *stream, const PaintSrcArgs *ps = nullptr,
                               int2 *plan_hdr = nullptr, i32x4 *plan_ent = nullptr, bool plan_make = false);
int gvl_realign_tracks(const gvl_static *st, const gvl_batch *bt, const float *tracks,
                       const int64_t *track_offsets, const double *params, int64_t strategy_id,
                       uint64_t base_seed, float *out, void *stream) {
    return realign_tracks_impl(st, bt, tracks, track_offsets, params, strategy_id, base_seed, nullptr, out, stream);
}
static int realign_tracks_impl(const gvl_static *st, const gvl_batch *bt, const float *tracks,
                               const int64_t *track_offsets, const double *params, int64_t strategy_id,
                               uint64_t base_seed, const u64 *seed_ptr, float *out, void *stream, const PaintSrcArgs *ps,
                               int2 *plan_hdr, i32x4 *plan_ent, bool plan_make) {
    if (!st || !bt) return fail(GVL_ERR_INVALID, "%s", "gvl_realign_tracks: NULL struct");
    if (bt->batch < 0 || bt->ploidy <= 0) return fail(GVL_ERR_INVALID, "%s", "gvl_realign_tracks: bad batch/ploidy");
    if (bt->batch == 0) return GVL_OK;
    if (!bt->regions || !bt->shifts || !bt->geno_offset_idx || !bt->out_offsets || bt->regions_stride < 3 ||
        !st->geno_o_starts || !st->geno_o_stops || (!tracks && !ps) || !track_offsets || !out || !params)
        return fail(GVL_ERR_INVALID, "%s", "gvl_realign_tracks: NULL/invalid array");
    if (st->n_geno > 0 && (!st->geno_v_idxs || !st->v_starts || !st->ilens))
        return fail(GVL_ERR_INVALID, "%s", "gvl_realign_tracks: NULL variant table");
    if (strategy_id < 0 || strategy_id > GVL_FILL_INTERPOLATE) return fail(GVL_ERR_INVALID, "%s", "gvl_realign_tracks: bad strategy_id");
    if (bt->max_row_len < 0 || bt->max_row_len > 0x7FFFFF00ll) return fail(GVL_ERR_INVALID, "%s", "gvl_realign_tracks: bad max_row_len");
    TrackArgs A;
    memset(&A, 0, sizeof(A));
    A.go_starts = (const i64 *)st->geno_o_starts; A.go_stops = (const i64 *)st->geno_o_stops;
    A.geno_v_idxs = st->geno_v_idxs; A.v_starts = st->v_starts; A.ilens = st->ilens; A.n_variants = st->n_variants;
    A.grec = (debug_flags() & 16) ? nullptr : st->geno_rec;
    A.regions = bt->regions; A.regions_stride = bt->regions_stride; A.shifts = bt->shifts;
    A.geno_offset_idx = (const i64 *)bt->geno_offset_idx; A.keep = bt->keep; A.keep_offsets = (const i64 *)bt->keep_offsets;
    A.to_rc = bt->to_rc; A.out_offsets = (const i64 *)bt->out_offsets;
    A.n_rows = bt->batch * bt->ploidy; A.ploidy = (int)bt->ploidy; A.ploidy_shift = log2_exact(bt->ploidy);
    int chunks = 1;
    if (pick_chunk(bt->max_row_len, &chunks, &A.chunk_len)) return fail(GVL_ERR_INVALID, "%s", "gvl_realign_tracks: too many chunks");
    A.tracks = tracks; A.track_offsets = (const i64 *)track_offsets;
    A.param = params[0]; A.strategy = strategy_id; A.base_seed = base_seed; A.seed_ptr = seed_ptr;
    A.out = out;
    A.dbg = debug_flags();
    A.stamps = g_stamps;
    if (A.n_rows > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_realign_tracks: batch too large");
    const i64 grid = (A.n_rows + 3) / 4;
    // rows of several chunks: the rows' plans, once per batch (the caller's scratch; every track of the batch reads the same
    // ones -- the walk does not depend on the track): headers (int2 per (row, chunk)), then the entry tables
    // (int2 per (row, chunk) + PLAN_MAXE entries per row: the caller has made sure both fit)
    if (plan_hdr && plan_ent && chunks > 1 && !(A.dbg & (8 | 268435456))) {
        if (plan_make) {
            track_plan_kernel<<<dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream>>>(A, plan_hdr, plan_ent, chunks, -1, 0);
            const int rc = check_launch("gvl_realign_tracks(row plans)");
            if (rc) return rc;
        }
        A.plan_hdr = plan_hdr; A.plan_ent = plan_ent;
    }
    // (GVL_TRACK_EXTRA_LDS: bytes of unused LDS per workgroup, to measure the kernel at fewer waves per SIMD)
    const unsigned xl = 0;
    if (ps) realign_tracks_kernel<true><<<dim3((unsigned)grid, (unsigned)chunks), dim3(256), xl, (hipStream_t)stream>>>(A, *ps);
    else realign_tracks_kernel<false><<<dim3((unsigned)grid, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream>>>(A, PaintSrcArgs());
    return check_launch("gvl_realign_tracks");
}


// scratch layout of gvl_tracks_batch: track_offsets i64 (batch + 1) | out_offsets i64 (batch * ploidy + 1) |
// chunk records (16 B x batch * chunks) | scratch tracks f32 (batch * stride) | row plans (track_plan_bytes)
static void tracks_scratch_parts(i64 batch, i64 ploidy, i64 stride, i64 part[6]) {
    const i64 n_chunks = (stride + 2047) / 2048;
    const i64 sz[5] = {8 * (batch + 1), 8 * (batch * ploidy + 1), batch * n_chunks * (i64)sizeof(PaintTodo), 4 * batch * stride,
                       n_chunks > 1 ? track_plan_bytes(batch * ploidy, n_chunks) : 0};
    i64 off = 0;
    for (int i = 0; i < 5; ++i) { part[i] = off; off += (sz[i] + 255) & ~255ll; }
    part[5] = off;
}

int64_t gvl_tracks_scratch_bytes(int64_t batch, int64_t ploidy, int64_t scratch_stride) {
    if (batch < 0 || ploidy <= 0 || scratch_stride < 0) return -1;
    i64 part[6];
    tracks_scratch_parts(batch, ploidy, scratch_stride, part);
    return part[5] > 0 ? part[5] : 256;
}

static int tracks_batch_impl(const gvl_static *st, const gvl_batch *bt, const int64_t *offset_idxs, const gvl_track_set *tracks,
                             int32_t n_tracks, const double *params, int64_t strategy_id, uint64_t base_seed, const u64 *seed_ptr,
                             float *out, int64_t out_track_stride, void *scratch, int64_t scratch_stride, void *stream,
                             const i64 *pre_track_offsets = nullptr, const i64 *pre_out_offsets = nullptr,
                             const int2 *pre_plan_hdr = nullptr, const i32x4 *pre_plan_ent = nullptr);
int gvl_tracks_batch(const gvl_static *st, const gvl_batch *bt, const int64_t *offset_idxs, const gvl_track_set *tracks,
                     int32_t n_tracks, const double *params, int64_t strategy_id, uint64_t base_seed, float *out,
                     int64_t out_track_stride, void *scratch, int64_t scratch_stride, void *stream) {
    return tracks_batch_impl(st, bt, offset_idxs, tracks, n_tracks, params, strategy_id, base_seed, nullptr, out, out_track_stride,
                             scratch, scratch_stride, stream);
}
static int tracks_batch_impl(const gvl_static *st, const gvl_batch *bt, const int64_t *offset_idxs, const gvl_track_set *tracks,
                             int32_t n_tracks, const double *params, int64_t strategy_id, uint64_t base_seed, const u64 *seed_ptr,
                             float *out, int64_t out_track_stride, void *scratch, int64_t scratch_stride, void *stream,
                             const i64 *pre_track_offsets, const i64 *pre_out_offsets, const int2 *pre_plan_hdr, const i32x4 *pre_plan_ent) {
    if (!st || !bt || n_tracks < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_tracks_batch: bad arguments");
    if (bt->batch < 0 || bt->ploidy <= 0 || bt->output_length < 0)
        return fail(GVL_ERR_INVALID, "%s", "gvl_tracks_batch: needs batch >= 0, ploidy > 0 and a fixed output_length");
    if (bt->batch == 0 || n_tracks == 0) return GVL_OK;
    if (bt->batch > 65535) return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_tracks_batch: more than 65535 queries per call");
    if (!bt->regions || !bt->shifts || !bt->geno_offset_idx || bt->regions_stride < 3 || !offset_idxs || !tracks || !params ||
        !out || !scratch || ((uintptr_t)scratch & 255) || scratch_stride <= 0 || scratch_stride > 0x7FFFFF00ll)
        return fail(GVL_ERR_INVALID, "%s", "gvl_tracks_batch: NULL/invalid array (scratch: gvl_tracks_scratch_bytes(), 256-byte aligned)");
    const i64 B = bt->batch, P = bt->ploidy, L = bt->output_length;
    if (out_track_stride < B * P * L) return fail(GVL_ERR_INVALID, "%s", "gvl_tracks_batch: out_track_stride < batch * ploidy * output_length");
    hipStream_t s = (hipStream_t)stream;
    i64 part[6];
    tracks_scratch_parts(B, P, scratch_stride, part);
    u8 *base = (u8 *)scratch;
    // the rows' plans: the caller's (the native loop prepares them with its epoch table) or made here, once per call
    int2 *plan_hdr = const_cast<int2 *>(pre_plan_hdr);
    i32x4 *plan_ent = const_cast<i32x4 *>(pre_plan_ent);
    bool plan_made = plan_hdr != nullptr && plan_ent != nullptr;
    if (!plan_made) {
        plan_hdr = nullptr; plan_ent = nullptr;
        int pc = 1, pcl = 0;
        if (!pick_chunk(L, &pc, &pcl) && pc > 1 && track_plan_bytes(B * P, pc) <= part[5] - part[4]) {
            plan_hdr = (int2 *)(base + part[4]);
            plan_ent = (i32x4 *)(base + part[4] + ((B * P * (i64)pc * (i64)sizeof(int2) + 255) & ~255ll));
        }
    }
    i64 *track_offsets = (i64 *)(base + part[0]);
    i64 *out_offsets = (i64 *)(base + part[1]);
    PaintTodo *todo = (PaintTodo *)(base + part[2]);
    float *scr = (float *)(base + part[3]);
    // 1. scratch-track lengths -> offsets (the reference sizes the scratch track per query, _reconstruct.py:191);
    // the native loop has them for every batch of the epoch already (gvl_loader_start_epoch)
    DiffArgs D;
    int rc = fill_diff_args(D, st, bt, "gvl_tracks_batch");
    if (rc) return rc;
    D.keep = nullptr; D.keep_offsets = nullptr;
    if (pre_track_offsets && pre_out_offsets) {
        track_offsets = const_cast<i64 *>(pre_track_offsets);
        out_offsets = const_cast<i64 *>(pre_out_offsets);
    } else {
        const i64 grid = (B * WAVE + 255) / 256;            // one wave per query (covers the K + 1 offsets too)
        track_lengths_kernel<<<dim3((unsigned)grid), dim3(256), 0, s>>>(D, bt->regions, (i64)bt->regions_stride, B, L, track_offsets, out_offsets);
        rc = check_launch("gvl_tracks_batch(lengths)");
        if (rc) return rc;
        offsets_scan_kernel<<<dim3(1), dim3(1024), 0, s>>>(track_offsets, B, (i64 *)nullptr);
        rc = check_launch("gvl_tracks_batch(scan)");
        if (rc) return rc;
    }
    // 2. per track: paint the query's intervals into its scratch track, realign it to every haplotype
    gvl_batch rb = *bt;
    rb.out_offsets = (const int64_t *)out_offsets;
    rb.max_row_len = L;
    rb.output_length = -1;
    for (int t = 0; t < n_tracks; ++t) {
        const gvl_track_set &T = tracks[t];
        if (!T.itv_offsets || (T.n_intervals > 0 && (!T.itv_starts || !T.itv_ends || !T.itv_values)))
            return fail(GVL_ERR_INVALID, "%s", "gvl_tracks_batch: NULL interval array");
        PaintIndex X{nullptr, nullptr, nullptr, nullptr};
        if (T.bkt_offsets && T.bkt_base && T.bkt_lo && T.bkt_hi && !(debug_flags() & 1024))
            X = PaintIndex{(const i64 *)T.bkt_offsets, T.bkt_base, T.bkt_lo, T.bkt_hi};
        // the track's own insertion fill (_reconstruct.py:204-208 lowers one per track) or the call's
        const double t_par[1] = {T.has_fill ? T.fill_param : params[0]};
        const int64_t t_strategy = T.has_fill ? (int64_t)T.fill_strategy : strategy_id;
        // An interval set whose owner vouches for non-overlapping intervals (tile_complete) and that has its bucket
        // index is realigned straight from the intervals: the scratch track is neither written nor read (SrcPainted;
        // a window the claim does not hold for falls back to exact per-position lookups, it is never wrong).
        const bool fused = T.tile_complete != 0 && X.offsets && T.itv_pmax_ends && !(debug_flags() & 4194304);
        if (fused) {
            PaintSrcArgs ps{(const i64 *)offset_idxs, T.list_div > 1 ? T.list_div : 1, T.itv_starts, T.itv_ends, T.itv_values,
                            (const i64 *)T.itv_offsets, T.itv_pmax_ends, X};
            rc = realign_tracks_impl(st, &rb, nullptr, (const int64_t *)track_offsets, T.has_fill ? t_par : params, t_strategy, base_seed,
                                     seed_ptr, out + (i64)t * out_track_stride, stream, &ps, plan_hdr, plan_ent, !plan_made);
            if (rc) return rc;
            plan_made = true;
            continue;
        }
        rc = paint_launch(offset_idxs, bt->regions + 1, bt->regions_stride, B, T.itv_starts, T.itv_ends, T.itv_values, T.itv_offsets,
                          T.itv_pmax_ends, scr, (const int64_t *)track_offsets, scratch_stride,
                          paint_can_tile(T.itv_pmax_ends, scratch_stride) ? todo : nullptr, s, X, T.tile_complete != 0,
                          T.list_div > 1 ? T.list_div : 1);
        if (rc) return rc;
        rc = realign_tracks_impl(st, &rb, scr, (const int64_t *)track_offsets, T.has_fill ? t_par : params, t_strategy, base_seed, seed_ptr,
                                 out + (i64)t * out_track_stride, stream, nullptr, plan_hdr, plan_ent, !plan_made);
        if (rc) return rc;
        plan_made = true;
    }
    return GVL_OK;
}

int gvl_prepare_request(const gvl_static *st, const int64_t *idx, int64_t batch,
                        const int32_t *full_regions, int64_t n_regions, int64_t n_samples,
                        int64_t ploidy, int64_t jitter, int32_t rc_neg, int32_t deterministic,
                        int64_t output_length, uint64_t seed, uint64_t counter,
                        int32_t *regions_out, int64_t *geno_offset_idx_out, uint8_t *to_rc_out,
                        int32_t *shifts_out, void *stream) {
    if (batch < 0 || n_regions <= 0 || n_samples <= 0 || ploidy <= 0 || ploidy > 64 || jitter < 0 || jitter > (1 << 20))
        return fail(GVL_ERR_INVALID, "%s", "gvl_prepare_request: bad sizes");
    if (batch == 0) return GVL_OK;
    if (!idx || !full_regions || !regions_out || !geno_offset_idx_out || !to_rc_out || !shifts_out)
        return fail(GVL_ERR_INVALID, "%s", "gvl_prepare_request: NULL array");
    PrepArgs P;
    memset(&P, 0, sizeof(P));
    if (!deterministic) {
        if (!st || !st->geno_o_starts || !st->geno_o_stops || (st->n_geno > 0 && (!st->geno_v_idxs || !st->ilens || !st->v_starts)))
            return fail(GVL_ERR_INVALID, "%s", "gvl_prepare_request: random shifts need the genotype CSR + variant table");
        P.D.geno_v_idxs = st->geno_v_idxs; P.D.go_starts = (const i64 *)st->geno_o_starts;
        P.D.grec = (debug_flags() & 16) ? nullptr : st->geno_rec;
        P.D.go_stops = (const i64 *)st->geno_o_stops; P.D.ilens = st->ilens; P.D.v_starts = st->v_starts;
        P.D.n_variants = st->n_variants; P.D.n_rows = batch * ploidy;
    }
    P.idx = (const i64 *)idx; P.batch = batch; P.full_regions = full_regions; P.n_regions = n_regions;
    P.n_samples = n_samples; P.ploidy = (int)ploidy; P.jitter = (int)jitter; P.rc_neg = rc_neg;
    P.deterministic = deterministic; P.output_length = output_length; P.seed = seed; P.counter = counter;
    P.regions = regions_out; P.goi = (i64 *)geno_offset_idx_out; P.to_rc = to_rc_out; P.shifts = shifts_out;
    const i64 grid = (batch * ploidy + 255) / 256;
    if (grid > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_prepare_request: batch too large");
    prepare_request_kernel<<<dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream>>>(P);
    return check_launch("gvl_prepare_request");
}

// ---- native batch loop ---------------------------------------------------------------
// Epoch level: gvl_loader_start_epoch turns the WHOLE epoch order into request arrays with one launch
// of the prep kernel (the "epoch table": regions / geno_offset_idx / shifts / to_rc for every query of
// the epoch, 26 + 13 P bytes per query), so that a batch costs the host one launch, one event record
// and at most two stream waits -- per GROUP of `group` batches (gvl_reconstruct_many).
__device__ __forceinline__ u64 splitmix64_dev(u64 x) {
    u64 z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// per-batch base_seed of the seed-dependent track fills (_reconstruct.py:215-222): deterministic -> xor of the
// batch's dataset indices; else a draw keyed by (seed, epoch, batch).  One wave per batch.
__global__ __launch_bounds__(256) void batch_seeds_kernel(const i64 *order, i64 n, i64 bs, i64 n_batches, int deterministic,
                                                          u64 seed, u64 epoch, u64 *out) {
    const int lane = threadIdx.x & (WAVE - 1);
    const i64 j = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n_batches) return;
    if (!deterministic) {
        if (lane == 0) out[j] = splitmix64_dev(seed ^ splitmix64_dev((epoch << 32) + (u64)j));
        return;
    }
    const i64 lo = j * bs, hi = (lo + bs < n) ? lo + bs : n;
    u64 acc = 0;
    for (i64 i = lo + lane; i < hi; i += WAVE) acc ^= (u64)order[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u32 lo32 = (u32)__shfl_xor((int)(u32)acc, off, WAVE), hi32 = (u32)__shfl_xor((int)(u32)(acc >> 32), off, WAVE);
        acc ^= ((u64)hi32 << 32) | lo32;
    }
    if (lane == 0) out[j] = acc;
}

struct gvl_loader {
    gvl_static st;
    gvl_loader_config cfg;
    void *arenas[64];
    int64_t part[GVL_LOADER_SLOT_PARTS];
    gvl_track_set tracks[16];     // copy of cfg.tracks
    u64 *e_seeds;                 // epoch table: per-batch track seeds
    hipStream_t streams[16];
    hipEvent_t done[64], released[64], epoch_ready;    // per slot SET (group of `G` slots)
    bool set_used[64];
    bool stream_synced[16];
    const int64_t *order; i64 n_order; i64 n_batches, n_groups;
    i64 submitted, consumed;      // submitted: GROUPS handed to the GPU; consumed: BATCHES handed to the caller
    i64 released_groups;          // groups whose release has been recorded on the consumer's stream
    int G, n_sets;
    int2 *e_plan_hdr; i32x4 *e_plan_ent;    // tracks, rows of several chunks: the rows' plans (track_plan_kernel) of the epoch (or NULL)
    int e_chunks;
    i64 *e_track_offsets, *e_out_offsets;   // tracks: every batch's scratch-track offsets ((bs + 1) per batch) and the k * L output offsets
    u64 counter;                  // the running epoch's number + 1 (keys the random draws together with cfg.seed)
    bool epoch_set;               // gvl_loader_set_epoch named the next epoch (else: epochs started so far)
    u64 next_epoch;
    // epoch table (the caller's memory, like the slots): request arrays of every query of the epoch
    int *e_regions; i64 *e_goi; int *e_shifts; u8 *e_to_rc;
    struct LoaderSync *sync;      // non-NULL: a producer thread submits the groups (cfg.threaded)
    // an epoch prepared ahead of its start (gvl_loader_prefetch_epoch): its table is filled, pf_ready recorded behind it
    hipEvent_t pf_ready;
    bool pf_valid;
    const int64_t *pf_order; i64 pf_n; int pf_drop_last; void *pf_table; u64 pf_counter;
    void *cur_table;              // the running epoch's table
};

// Producer thread state.  `submitted` / `consumed` / `n_batches` / `order` are only touched under
// `mu` once the thread exists; the HIP calls themselves run outside the lock.
struct LoaderSync {
    std::mutex mu;
    std::condition_variable cv_producer, cv_consumer;
    std::thread th;
    bool stop = false, active = false, busy = false;
    int err = GVL_OK;
    char msg[512] = "";
    int device = 0;
};

static i64 align256(i64 x) { return (x + 255) & ~255ll; }
// the track plans of an epoch's rows (rows of several chunks only; an epoch whose plans would not fit GVL_TRACK_PLAN_MAX_MB,
// default 512, goes without: its chunk-waves then walk their rows' variants themselves)
static i64 loader_track_plan_bytes(const gvl_loader_config *cfg, i64 n) {
    if (cfg->output_length <= 2048) return 0;
    int chunks = 1, chunk_len = 0;
    if (pick_chunk(cfg->output_length, &chunks, &chunk_len) || chunks <= 1) return 0;
    const i64 cap_t = tune(GVL_TUNE_TRACK_PLAN_MAX_MB);
    const i64 cap = (cap_t > 0 ? cap_t : 512) << 20;
    const i64 b = track_plan_bytes(n * cfg->ploidy, chunks);
    return b <= cap ? b : 0;
}

static bool loader_ragged(const gvl_loader_config *c) { return c->output_length == -1; }
// bases per row a slot reserves: the fixed length, or the ragged bound
static i64 loader_row_cap(const gvl_loader_config *c) { return loader_ragged(c) ? c->max_row_len : c->output_length; }

int64_t gvl_loader_slot_bytes(const gvl_loader_config *cfg, int64_t *part_offsets) {
    if (!cfg || cfg->batch_size <= 0 || cfg->ploidy <= 0 || loader_row_cap(cfg) <= 0 || cfg->n_tracks < 0) return -1;
    const i64 b = cfg->batch_size, K = b * cfg->ploidy, L = loader_row_cap(cfg);
    const bool hp = cfg->want_haps || cfg->want_annot;
    const i64 scr = cfg->n_tracks > 0 ? gvl_tracks_scratch_bytes(b, cfg->ploidy, cfg->scratch_stride) : 0;
    if (scr < 0) return -1;
    const i64 sizes[GVL_LOADER_SLOT_PARTS] = {cfg->want_onehot ? 4 * K * L : 0, hp ? K * L : 0, 0, 0, 0, 0, 8 * (K + 1),
                                              cfg->want_annot ? 4 * K * L : 0, cfg->want_annot ? 4 * K * L : 0,
                                              4 * (i64)cfg->n_tracks * K * L, scr, 16};
    i64 off = 0;
    for (int i = 0; i < GVL_LOADER_SLOT_PARTS; ++i) {
        if (part_offsets) part_offsets[i] = off;
        off += align256(sizes[i]);
    }
    return off;
}

int64_t gvl_loader_table_bytes(const gvl_loader_config *cfg, int64_t n, int64_t *part_offsets) {
    if (!cfg || cfg->ploidy <= 0 || cfg->batch_size <= 0 || n < 0) return -1;
    const i64 P = cfg->ploidy;
    const i64 nb = (n + cfg->batch_size - 1) / cfg->batch_size;
    const bool tr = cfg->n_tracks > 0;
    const i64 sizes[GVL_LOADER_TABLE_PARTS] = {16 * n, 8 * n * P, 4 * n * P, n * P, 8 * nb,
                                               tr ? 8 * (n + nb) : 0, tr ? 8 * (cfg->batch_size * P + 1) : 0,
                                               tr ? loader_track_plan_bytes(cfg, n) : 0};
    i64 off = 0;
    for (int i = 0; i < GVL_LOADER_TABLE_PARTS; ++i) {
        if (part_offsets) part_offsets[i] = off;
        off += align256(sizes[i]);
    }
    return off > 0 ? off : 256;
}

static int loader_submit(gvl_loader *ld, i64 g);
static void loader_producer_main(gvl_loader *ld);

int gvl_loader_create(const gvl_static *st, const gvl_loader_config *cfg, gvl_loader **out) {
    if (!st || !cfg || !out) return fail(GVL_ERR_INVALID, "%s", "gvl_loader_create: NULL argument");
    const int G = cfg->group <= 0 ? 1 : cfg->group;
    if (G > GVL_MANY_MAX) return fail(GVL_ERR_INVALID, "%s", "gvl_loader_create: group > GVL_MANY_MAX");
    if (cfg->in_flight < 1 || cfg->in_flight > 16 || cfg->n_slots > 64 || cfg->n_slots % G != 0 ||
        cfg->n_slots / G < cfg->in_flight + 1)
        return fail(GVL_ERR_INVALID, "%s", "gvl_loader_create: need 1 <= in_flight <= 16, n_slots <= 64 a multiple of group, "
                                           "n_slots / group >= in_flight + 1");
    if (!cfg->full_regions || !cfg->slot_arenas || cfg->n_regions <= 0 || cfg->n_samples <= 0 || cfg->batch_size <= 0 ||
        cfg->ploidy <= 0 || (cfg->output_length <= 0 && cfg->output_length != -1) ||
        (!cfg->want_haps && !cfg->want_onehot && !cfg->want_annot))
        return fail(GVL_ERR_INVALID, "%s", "gvl_loader_create: bad config");
    if (loader_ragged(cfg) && (cfg->max_row_len <= 0 || cfg->max_row_len > 0x7FFFFF00ll || !cfg->deterministic || cfg->n_tracks > 0 ||
                               (cfg->want_onehot && cfg->onehot_layout != GVL_ONEHOT_LC)))
        return fail(GVL_ERR_INVALID, "%s", "gvl_loader_create: ragged rows (output_length -1) need max_row_len > 0, deterministic != 0, "
                                           "row-major one-hot and no tracks");
    if (cfg->n_tracks < 0 || cfg->n_tracks > 16 || (cfg->n_tracks > 0 && (!cfg->tracks || cfg->scratch_stride <= 0 || cfg->batch_size > 65535 ||
                                                                          cfg->strategy_id < 0 || cfg->strategy_id > GVL_FILL_INTERPOLATE)))
        return fail(GVL_ERR_INVALID, "%s", "gvl_loader_create: tracks need 1..16 interval stores, scratch_stride > 0, batch_size <= 65535 "
                                           "and a valid strategy_id");
    gvl_loader *ld = new (std::nothrow) gvl_loader;
    if (!ld) return fail(GVL_ERR_HIP, "%s", "gvl_loader_create: out of host memory");
    memset(ld, 0, sizeof(*ld));
    ld->st = *st; ld->cfg = *cfg;
    ld->G = G; ld->n_sets = cfg->n_slots / G;
    for (int t = 0; t < cfg->n_tracks; ++t) ld->tracks[t] = cfg->tracks[t];
    ld->cfg.tracks = ld->tracks;
    if (cfg->want_annot) ld->cfg.want_haps = 1;
    if (gvl_loader_slot_bytes(cfg, ld->part) <= 0) { delete ld; return fail(GVL_ERR_INVALID, "%s", "gvl_loader_create: bad slot sizes"); }
    for (int i = 0; i < cfg->n_slots; ++i) {
        ld->arenas[i] = cfg->slot_arenas[i];
        if (!ld->arenas[i] || ((uintptr_t)ld->arenas[i] & 255)) {
            delete ld;
            return fail(GVL_ERR_INVALID, "%s", "gvl_loader_create: slot arenas must be non-NULL and 256-byte aligned");
        }
    }
    ld->cfg.slot_arenas = nullptr;
    bool ok = true;
    for (int i = 0; i < cfg->in_flight && ok; ++i) ok = hipStreamCreateWithFlags(&ld->streams[i], hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < ld->n_sets && ok; ++i)
        ok = hipEventCreateWithFlags(&ld->done[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ld->released[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ld->epoch_ready, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ld->pf_ready, hipEventDisableTiming) == hipSuccess;
    if (!ok) { gvl_loader_destroy(ld); return fail(GVL_ERR_HIP, "%s", "gvl_loader_create: stream / event creation failed"); }
    if (cfg->threaded) {
        LoaderSync *sy = new (std::nothrow) LoaderSync;
        if (!sy) { gvl_loader_destroy(ld); return fail(GVL_ERR_HIP, "%s", "gvl_loader_create: out of host memory"); }
        if (hipGetDevice(&sy->device) != hipSuccess) sy->device = 0;
        ld->sync = sy;
        sy->th = std::thread(loader_producer_main, ld);
    }
    *out = ld;
    return GVL_OK;
}

int gvl_loader_destroy(gvl_loader *ld) {
    if (!ld) return GVL_OK;
    if (ld->sync) {
        {
            std::lock_guard<std::mutex> lk(ld->sync->mu);
            ld->sync->stop = true;
        }
        ld->sync->cv_producer.notify_all();
        if (ld->sync->th.joinable()) ld->sync->th.join();
        delete ld->sync;
        ld->sync = nullptr;
    }
    trace_report();
    for (int i = 0; i < 16; ++i) if (ld->streams[i]) { (void)hipStreamSynchronize(ld->streams[i]); (void)hipStreamDestroy(ld->streams[i]); }
    for (int i = 0; i < 64; ++i) {
        if (ld->done[i]) (void)hipEventDestroy(ld->done[i]);
        if (ld->released[i]) (void)hipEventDestroy(ld->released[i]);
    }
    if (ld->epoch_ready) (void)hipEventDestroy(ld->epoch_ready);
    if (ld->pf_ready) (void)hipEventDestroy(ld->pf_ready);
    delete ld;
    return GVL_OK;
}

int gvl_loader_set_epoch(gvl_loader *ld, uint64_t epoch) {
    if (!ld) return fail(GVL_ERR_INVALID, "%s", "gvl_loader_set_epoch: NULL loader");
    ld->next_epoch = epoch;
    ld->epoch_set = true;
    return GVL_OK;
}

// the request arrays of every query of an epoch, the per-batch track seeds and the scratch-track sizing of its batches,
// into `table` on stream `s`
static int loader_fill_table(gvl_loader *ld, const int64_t *order, i64 n, int32_t drop_last, void *table, u64 counter, hipStream_t s) {
    const gvl_loader_config &c = ld->cfg;
    int64_t po[GVL_LOADER_TABLE_PARTS + 1];
    po[GVL_LOADER_TABLE_PARTS] = gvl_loader_table_bytes(&c, n, po);
    u8 *base = (u8 *)table;
    int *t_regions = (int *)(base + po[0]);
    i64 *t_goi = (i64 *)(base + po[1]);
    int *t_shifts = (int *)(base + po[2]);
    u8 *t_to_rc = base + po[3];
    u64 *t_seeds = (u64 *)(base + po[4]);
    i64 *t_track_offsets = (i64 *)(base + po[5]);
    i64 *t_out_offsets = (i64 *)(base + po[6]);
    const i64 bs = c.batch_size;
    const i64 n_batches = drop_last ? n / bs : (n + bs - 1) / bs;
    const i64 n_used = drop_last ? n_batches * bs : n;
    if (n_used <= 0) return GVL_OK;
    const int rc = gvl_prepare_request(&ld->st, order, n_used, c.full_regions, c.n_regions, c.n_samples, c.ploidy, c.jitter,
                                       c.rc_neg, c.deterministic, c.output_length < 0 ? 0 : c.output_length, c.seed, counter,
                                       t_regions, (int64_t *)t_goi, t_to_rc, t_shifts, s);
    if (rc) return rc;
    if (c.n_tracks > 0 && c.track_seed_mode == 1) {
        const i64 grid = (n_batches * WAVE + 255) / 256;
        batch_seeds_kernel<<<dim3((unsigned)grid), dim3(256), 0, s>>>((const i64 *)order, n_used, bs, n_batches, c.deterministic,
                                                                      c.seed, counter, t_seeds);
        const int rc2 = check_launch("gvl_loader_start_epoch(seeds)");
        if (rc2) return rc2;
    }
    if (c.n_tracks > 0) {
        // the scratch-track sizing of every batch of the epoch: lengths (one wave per query), then one scan per batch
        gvl_batch eb;
        memset(&eb, 0, sizeof(eb));
        eb.regions = t_regions; eb.regions_stride = 4; eb.shifts = t_shifts; eb.geno_offset_idx = (const int64_t *)t_goi;
        eb.batch = n_used; eb.ploidy = c.ploidy; eb.output_length = c.output_length;
        DiffArgs D;
        int rc3 = fill_diff_args(D, &ld->st, &eb, "gvl_loader_start_epoch");
        if (rc3) return rc3;
        D.keep = nullptr; D.keep_offsets = nullptr;
        const i64 grid = (n_used * WAVE + 255) / 256;
        if (grid > 0x7FFFFFFFll) return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_loader_start_epoch: too many queries for the track sizing");
        track_lengths_epoch_kernel<<<dim3((unsigned)grid), dim3(256), 0, s>>>(D, t_regions, 4, n_used, bs, c.output_length,
                                                                               t_track_offsets, t_out_offsets);
        rc3 = check_launch("gvl_loader_start_epoch(track lengths)");
        if (rc3) return rc3;
        track_scan_batches_kernel<<<dim3((unsigned)n_batches), dim3(256), 0, s>>>(t_track_offsets, n_used, bs);
        rc3 = check_launch("gvl_loader_start_epoch(track offsets)");
        if (rc3) return rc3;
        // rows of several chunks: the rows' plans (track_plan_kernel), for every row of the epoch
        if (po[8] > po[7] && !(debug_flags() & (8 | 268435456))) {
            TrackArgs TA;
            memset(&TA, 0, sizeof(TA));
            TA.go_starts = (const i64 *)ld->st.geno_o_starts; TA.go_stops = (const i64 *)ld->st.geno_o_stops;
            TA.geno_v_idxs = ld->st.geno_v_idxs; TA.v_starts = ld->st.v_starts; TA.ilens = ld->st.ilens; TA.n_variants = ld->st.n_variants;
            TA.grec = (debug_flags() & 16) ? nullptr : ld->st.geno_rec;
            TA.regions = t_regions; TA.regions_stride = 4; TA.shifts = t_shifts; TA.geno_offset_idx = t_goi;
            TA.n_rows = n_used * c.ploidy; TA.ploidy = (int)c.ploidy; TA.ploidy_shift = log2_exact(c.ploidy);
            int chunks = 1;
            if (pick_chunk(c.output_length, &chunks, &TA.chunk_len)) return fail(GVL_ERR_INVALID, "%s", "gvl_loader_start_epoch: too many chunks");
            TA.dbg = debug_flags();
            const i64 wgrid = (TA.n_rows + 3) / 4;
            if (wgrid > 0x7FFFFFFFll) return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_loader_start_epoch: too many rows for the walk states");
            TA.track_offsets = t_track_offsets;
            int2 *const ph = (int2 *)(base + po[7]);
            i32x4 *const pe = (i32x4 *)(base + po[7] + align256(TA.n_rows * (i64)chunks * (i64)sizeof(int2)));
            track_plan_kernel<<<dim3((unsigned)wgrid), dim3(256), 0, s>>>(TA, ph, pe, chunks, c.output_length, bs);
            rc3 = check_launch("gvl_loader_start_epoch(row plans)");
            if (rc3) return rc3;
        }
    }
    return GVL_OK;
}

int gvl_loader_prefetch_epoch(gvl_loader *ld, uint64_t epoch, const int64_t *order, int64_t n, int32_t drop_last, void *table, void *stream) {
    if (!ld || n < 0 || (n > 0 && (!order || !table)) || ((uintptr_t)table & 255))
        return fail(GVL_ERR_INVALID, "%s", "gvl_loader_prefetch_epoch: bad arguments (table: gvl_loader_table_bytes() bytes, 256-byte aligned)");
    if (n > (1ll << 31)) return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_loader_prefetch_epoch: more than 2^31 queries per epoch (shard the order)");
    if (table == ld->cur_table) return fail(GVL_ERR_INVALID, "%s", "gvl_loader_prefetch_epoch: `table` is the running epoch's table (alternate between two)");
    ld->pf_valid = false;
    const int rc = loader_fill_table(ld, order, n, drop_last, table, epoch + 1, (hipStream_t)stream);
    if (rc) return rc;
    if (hipEventRecord(ld->pf_ready, (hipStream_t)stream) != hipSuccess)
        return fail(GVL_ERR_HIP, "%s", "gvl_loader_prefetch_epoch: hipEventRecord failed");
    ld->pf_order = order; ld->pf_n = n; ld->pf_drop_last = drop_last; ld->pf_table = table; ld->pf_counter = epoch + 1;
    ld->pf_valid = true;
    return GVL_OK;
}

int gvl_loader_start_epoch(gvl_loader *ld, const int64_t *order, int64_t n, int32_t drop_last, void *table, void *stream) {
    if (!ld || n < 0 || (n > 0 && (!order || !table)) || ((uintptr_t)table & 255))
        return fail(GVL_ERR_INVALID, "%s", "gvl_loader_start_epoch: bad arguments (table: gvl_loader_table_bytes() bytes, 256-byte aligned)");
    if (n > (1ll << 31)) return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_loader_start_epoch: more than 2^31 queries per epoch (shard the order)");
    std::unique_lock<std::mutex> lk;
    if (ld->sync) {      // park the producer: no submit may be in progress while the epoch changes
        lk = std::unique_lock<std::mutex>(ld->sync->mu);
        ld->sync->active = false;
        ld->sync->cv_consumer.wait(lk, [&] { return !ld->sync->busy; });
        ld->sync->err = GVL_OK;
    }
    hipStream_t s = (hipStream_t)stream;
    const gvl_loader_config &c = ld->cfg;
    const bool abandoned = ld->submitted * ld->G < ld->n_batches || ld->consumed < ld->n_batches;
    if (abandoned)   // an abandoned epoch's batches still read their table and fill their slots: let them drain
        for (int i = 0; i < c.in_flight; ++i) (void)hipStreamSynchronize(ld->streams[i]);
    // groups that were handed to the consumer but never released (the epoch ended or was abandoned before the
    // next gvl_loader_next): their release is recorded HERE, on `stream` -- the consumer's stream by contract --
    // so that the new epoch's submits wait for whatever the consumer still has queued on those slots
    if (ld->consumed > 0 && ld->n_sets > 0) {
        const i64 g_last = (ld->consumed - 1) / ld->G;
        for (i64 g = ld->released_groups; g <= g_last; ++g)
            if (hipEventRecord(ld->released[g % ld->n_sets], s) != hipSuccess)
                return fail(GVL_ERR_HIP, "%s", "gvl_loader_start_epoch: hipEventRecord failed");
        ld->released_groups = g_last + 1;
    }
    u64 counter = ld->counter + 1;
    if (ld->epoch_set) { counter = ld->next_epoch + 1; ld->epoch_set = false; }
    // was exactly this epoch prepared ahead (gvl_loader_prefetch_epoch)?  Then its table is filled -- or being filled,
    // pf_ready says when -- and nothing of the running epoch is touched: no wait for that epoch's last batches
    const bool prefetched = ld->pf_valid && ld->pf_order == order && ld->pf_n == n && ld->pf_drop_last == drop_last &&
                            ld->pf_table == table && ld->pf_counter == counter && table != ld->cur_table;
    ld->pf_valid = false;
    if (!prefetched) {
        // the previous epoch's last batches were handed to the consumer: the new table contents must not
        // overtake whatever is still queued on them
        for (int i = 0; i < ld->n_sets; ++i)
            if (ld->set_used[i] && hipStreamWaitEvent(s, ld->done[i], 0) != hipSuccess)
                return fail(GVL_ERR_HIP, "%s", "gvl_loader_start_epoch: hipStreamWaitEvent failed");
    }
    {
        int64_t po[GVL_LOADER_TABLE_PARTS];
        gvl_loader_table_bytes(&c, n, po);
        u8 *base = (u8 *)table;
        ld->e_regions = (int *)(base + po[0]);
        ld->e_goi = (i64 *)(base + po[1]);
        ld->e_shifts = (int *)(base + po[2]);
        ld->e_to_rc = base + po[3];
        ld->e_seeds = (u64 *)(base + po[4]);
        ld->e_track_offsets = (i64 *)(base + po[5]);
        ld->e_out_offsets = (i64 *)(base + po[6]);
        const i64 wsb = c.n_tracks > 0 ? loader_track_plan_bytes(&c, n) : 0;
        int cl = 0;
        ld->e_chunks = 1;
        if (wsb > 0) (void)pick_chunk(c.output_length, &ld->e_chunks, &cl);
        const bool planned = wsb > 0 && !(debug_flags() & (8 | 268435456));
        // (the table was filled for n_used = every query of a whole batch, or all n: the plan's two parts are laid out for that many rows)
        const i64 n_used_rows = (drop_last ? (n / c.batch_size) * c.batch_size : n) * c.ploidy;
        ld->e_plan_hdr = planned ? (int2 *)(base + po[7]) : nullptr;
        ld->e_plan_ent = planned ? (i32x4 *)(base + po[7] + align256(n_used_rows * (i64)ld->e_chunks * (i64)sizeof(int2))) : nullptr;
    }
    const i64 bs = c.batch_size;
    ld->order = order; ld->n_order = n;
    ld->n_batches = drop_last ? n / bs : (n + bs - 1) / bs;
    ld->n_groups = (ld->n_batches + ld->G - 1) / ld->G;
    ld->submitted = ld->consumed = ld->released_groups = 0;
    ld->counter = counter;
    ld->cur_table = table;
    if (prefetched) {
        hipEvent_t t = ld->epoch_ready; ld->epoch_ready = ld->pf_ready; ld->pf_ready = t;
    } else {
        const int rc = loader_fill_table(ld, order, n, drop_last, table, counter, s);
        if (rc) return rc;
        if (hipEventRecord(ld->epoch_ready, s) != hipSuccess)
            return fail(GVL_ERR_HIP, "%s", "gvl_loader_start_epoch: hipEventRecord failed");
    }
    for (int i = 0; i < 16; ++i) ld->stream_synced[i] = false;
    if (ld->sync) {
        ld->sync->active = true;
        lk.unlock();
        ld->sync->cv_producer.notify_all();
    }
    return GVL_OK;
}

static int loader_parts(gvl_loader *ld, i64 j, gvl_loader_batch *o) {
    const i64 bs = ld->cfg.batch_size, P = ld->cfg.ploidy;
    const int slot = (int)(j % ld->cfg.n_slots);
    u8 *base = (u8 *)ld->arenas[slot];
    o->slot = slot;
    o->batch = (j + 1) * bs <= ld->n_order ? bs : ld->n_order - j * bs;
    o->idx = ld->order + j * bs;
    o->onehot = ld->cfg.want_onehot ? base + ld->part[0] : nullptr;
    o->haps = ld->cfg.want_haps ? base + ld->part[1] : nullptr;
    o->annot_v_idxs = ld->cfg.want_annot ? (int32_t *)(base + ld->part[7]) : nullptr;
    o->annot_ref_pos = ld->cfg.want_annot ? (int32_t *)(base + ld->part[8]) : nullptr;
    o->tracks = ld->cfg.n_tracks > 0 ? (float *)(base + ld->part[9]) : nullptr;
    o->sizes = loader_ragged(&ld->cfg) ? (int64_t *)(base + ld->part[11]) : nullptr;
    o->track_seed = (ld->cfg.n_tracks > 0 && ld->cfg.track_seed_mode == 1) ? (const uint64_t *)(ld->e_seeds + j) : nullptr;
    // the request arrays of the batch are rows of the epoch table
    o->regions = ld->e_regions + 4 * j * bs;
    o->geno_offset_idx = (int64_t *)(ld->e_goi + j * bs * P);
    o->shifts = ld->e_shifts + j * bs * P;
    o->to_rc = ld->e_to_rc + j * bs * P;
    o->out_offsets = (int64_t *)(base + ld->part[6]);
    return 0;
}

// submit GROUP g: batches [g G, min((g + 1) G, n_batches)) in one launch
static int loader_submit(gvl_loader *ld, i64 g) {
    const int set = (int)(g % ld->n_sets);
    const int si = (int)(g % ld->cfg.in_flight);
    hipStream_t s = ld->streams[si];
    if (!ld->stream_synced[si]) {
        if (traced("wait epoch_ready", [&] { return hipStreamWaitEvent(s, ld->epoch_ready, 0); }) != hipSuccess) return fail(GVL_ERR_HIP, "%s", "gvl_loader: hipStreamWaitEvent failed");
        ld->stream_synced[si] = true;
    }
    if (ld->set_used[set] && traced("wait released", [&] { return hipStreamWaitEvent(s, ld->released[set], 0); }) != hipSuccess)
        return fail(GVL_ERR_HIP, "%s", "gvl_loader: hipStreamWaitEvent failed");
    const gvl_loader_config &c = ld->cfg;
    gvl_batch bts[GVL_MANY_MAX];
    gvl_out ocs[GVL_MANY_MAX];
    HapGroupOut grp;
    memset(&grp, 0, sizeof(grp));
    bool group_sizing = false;
    int m = 0;
    for (i64 j = g * ld->G; j < (g + 1) * ld->G && j < ld->n_batches; ++j, ++m) {
        gvl_loader_batch o;
        loader_parts(ld, j, &o);
        gvl_batch &bt = bts[m];
        memset(&bt, 0, sizeof(bt));
        bt.regions = o.regions; bt.regions_stride = 4; bt.shifts = o.shifts; bt.geno_offset_idx = o.geno_offset_idx;
        bt.batch = o.batch; bt.ploidy = c.ploidy; bt.to_rc = c.rc_neg ? o.to_rc : nullptr;
        bt.output_length = c.output_length; bt.max_row_len = loader_row_cap(&c);
        gvl_out &oc = ocs[m];
        memset(&oc, 0, sizeof(oc));
        oc.haps = o.haps; oc.onehot = o.onehot; oc.onehot_layout = c.onehot_layout; oc.out_offsets = o.out_offsets;
        oc.annot_v_idxs = o.annot_v_idxs; oc.annot_ref_pos = o.annot_ref_pos;
        if (loader_ragged(&c)) {
            // row lengths and offsets on the device (rows cut to the slot's capacity are reported, never silent);
            // the reconstruct launch below then reads them -- no host round trip.  One sizing per GROUP (below) unless the
            // dataset's rows take the wave-per-row length kernel (or GVL_DBG & 134217728: per batch, as before round 4)
            if (diffs_long_rows(&ld->st) || (debug_flags() & 134217728)) {
                const int rc0 = hap_offsets_impl(&ld->st, &bt, nullptr, o.out_offsets, o.sizes, c.max_row_len, s);
                if (rc0) return rc0;
            } else {
                grp.offs[m] = (i64 *)o.out_offsets; grp.sizes[m] = (i64 *)o.sizes;
                group_sizing = true;
            }
            bt.out_offsets = o.out_offsets;
            oc.out_offsets = nullptr;
        }
    }
    if (group_sizing) {
        // the group's request arrays are consecutive rows of the epoch table: ONE length launch over all of its rows, ONE scan
        // launch with a workgroup per batch
        gvl_batch gb = bts[0];
        i64 total_q = 0;
        for (int i = 0; i < m; ++i) total_q += bts[i].batch;
        gb.batch = total_q;
        DiffArgs D;
        int rc0 = fill_diff_args(D, &ld->st, &gb, "gvl_loader(ragged sizing)");
        if (rc0) return rc0;
        D.q_starts = gb.regions + 1; D.q_ends = gb.regions + 2; D.q_stride = gb.regions_stride;
        D.diffs = nullptr; D.output_length = -1; D.lengths = nullptr;
        D.len_cap = c.max_row_len; D.async_err = async_err_word();
        const i64 rpb = c.batch_size * c.ploidy;
        const unsigned grid = (unsigned)((D.n_rows + 255) / 256);
        hap_lengths_group_kernel<<<dim3(grid), dim3(256), 0, s>>>(D, gb.regions, (i64)gb.regions_stride, rpb, grp);
        rc0 = check_launch("gvl_loader(ragged lengths)");
        if (rc0) return rc0;
        hap_scan_group_kernel<<<dim3((unsigned)m), dim3(256), 0, s>>>(grp, rpb, D.n_rows);
        rc0 = check_launch("gvl_loader(ragged offsets)");
        if (rc0) return rc0;
    }
    int rc = GVL_OK;
    (void)traced("launch reconstruct", [&] { rc = gvl_reconstruct_many(&ld->st, bts, ocs, m, s); return hipSuccess; });
    if (rc) return rc;
    if (c.n_tracks > 0) {
        m = 0;
        for (i64 j = g * ld->G; j < (g + 1) * ld->G && j < ld->n_batches; ++j, ++m) {
            gvl_loader_batch o;
            loader_parts(ld, j, &o);
            const i64 K = c.batch_size * c.ploidy;
            const double par[1] = {c.track_param};
            u8 *base = (u8 *)ld->arenas[o.slot];
            (void)traced("launch tracks", [&] {
                rc = tracks_batch_impl(&ld->st, &bts[m], (const int64_t *)o.idx, ld->tracks, c.n_tracks, par, c.strategy_id, c.track_seed,
                                       (const u64 *)o.track_seed, o.tracks, K * c.output_length, base + ld->part[10], c.scratch_stride, s,
                                       (debug_flags() & 131072) ? nullptr : ld->e_track_offsets + j * (c.batch_size + 1),
                                       (debug_flags() & 131072) ? nullptr : ld->e_out_offsets,
                                       ld->e_plan_hdr ? ld->e_plan_hdr + j * c.batch_size * c.ploidy * (i64)ld->e_chunks : nullptr,
                                       ld->e_plan_ent ? ld->e_plan_ent + j * c.batch_size * c.ploidy * (i64)PLAN_MAXE : nullptr);
                return hipSuccess;
            });
            if (rc) return rc;
        }
    }
    if (traced("record done", [&] { return hipEventRecord(ld->done[set], s); }) != hipSuccess) return fail(GVL_ERR_HIP, "%s", "gvl_loader: hipEventRecord failed");
    ld->set_used[set] = true;
    return GVL_OK;
}

// may group `g` be handed to the GPU?  at most in_flight groups beyond the ones fully handed out
static bool loader_may_submit(const gvl_loader *ld) {
    return ld->submitted < ld->n_groups && ld->submitted < ld->consumed / ld->G + ld->cfg.in_flight;
}

static void loader_producer_main(gvl_loader *ld) {
    LoaderSync *sy = ld->sync;
    (void)hipSetDevice(sy->device);
    std::unique_lock<std::mutex> lk(sy->mu);
    for (;;) {
        sy->cv_producer.wait(lk, [&] { return sy->stop || (sy->active && sy->err == GVL_OK && loader_may_submit(ld)); });
        if (sy->stop) break;
        const i64 g = ld->submitted;
        sy->busy = true;
        lk.unlock();
        const int rc = loader_submit(ld, g);           // HIP calls outside the lock
        lk.lock();
        sy->busy = false;
        if (rc) {
            sy->err = rc;
            snprintf(sy->msg, sizeof(sy->msg), "%s", g_err);    // g_err is this thread's
        } else {
            ld->submitted = g + 1;
        }
        sy->cv_consumer.notify_all();
    }
}

// the consumer moves past batch `consumed - 1`: if that was the last batch of its group, the group's
// slots may be refilled once the consumer's queued work has run
static int loader_release_prev(gvl_loader *ld, hipStream_t cs) {
    const i64 c = ld->consumed;
    if (c > 0 && (c % ld->G == 0 || c == ld->n_batches) && ld->released_groups * ld->G < c) {
        const i64 g = (c - 1) / ld->G;
        if (traced("record released", [&] { return hipEventRecord(ld->released[g % ld->n_sets], cs); }) != hipSuccess)
            return fail(GVL_ERR_HIP, "%s", "gvl_loader_next: hipEventRecord failed");
        ld->released_groups = g + 1;
    }
    return GVL_OK;
}

int gvl_loader_next(gvl_loader *ld, void *consumer_stream, gvl_loader_batch *out) {
    if (!ld || !out) return fail(GVL_ERR_INVALID, "%s", "gvl_loader_next: NULL argument");
    hipStream_t cs = (hipStream_t)consumer_stream;
    LoaderSync *sy = ld->sync;
    int rc = loader_release_prev(ld, cs);      // (`consumed` only changes on this thread)
    if (rc) return rc;
    memset(out, 0, sizeof(*out));
    if (ld->consumed >= ld->n_batches) { out->slot = -1; return GVL_OK; }
    const i64 j = ld->consumed, g = j / ld->G;
    if (sy) {
        std::unique_lock<std::mutex> lk(sy->mu);
        sy->cv_producer.notify_one();          // a release may have opened the window
        sy->cv_consumer.wait(lk, [&] { return sy->err != GVL_OK || ld->submitted > g; });
        if (sy->err != GVL_OK && ld->submitted <= g) {
            snprintf(g_err, sizeof(g_err), "%s", sy->msg);
            return sy->err;
        }
    } else {
        while (loader_may_submit(ld)) {
            rc = loader_submit(ld, ld->submitted);
            if (rc) return rc;
            ++ld->submitted;
        }
    }
    if (j % ld->G == 0 &&
        traced("wait done", [&] { return hipStreamWaitEvent(cs, ld->done[g % ld->n_sets], 0); }) != hipSuccess)
        return fail(GVL_ERR_HIP, "%s", "gvl_loader_next: hipStreamWaitEvent failed");
    loader_parts(ld, j, out);
    if (sy) {
        {
            std::lock_guard<std::mutex> lk(sy->mu);
            ld->consumed = j + 1;              // the window moves: the producer may submit one more group
        }
        sy->cv_producer.notify_one();
    } else {
        ld->consumed = j + 1;
    }
    return GVL_OK;
}

}  // extern "C"
