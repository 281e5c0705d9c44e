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

#include "gvl_internal.inc"

namespace {
#include "gvl_dev.inc"
#include "gvl_diffs.inc"
__global__ __launch_bounds__(256) void pack_ref4_kernel(const u8 *ref, i64 ref_len, u8 *out, i64 out_len) {
    // one thread = 8 bases = one dword of nibbles
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 b0 = t * 8;
    if (t * 4 >= out_len) return;
    u32 w = 0;
    if (b0 + 8 <= ref_len) {
        u64 v;
        __builtin_memcpy(&v, ref + b0, 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) w |= ref4_nibble((u32)(v >> (8 * i)) & 0xFFu) << (4 * i);
    } else {
        for (int i = 0; i < 8; ++i)
            if (b0 + i < ref_len) w |= ref4_nibble((u32)ref[b0 + i]) << (4 * i);
    }
    if (t * 4 + 4 <= out_len) {
        __builtin_memcpy(out + t * 4, &w, 4);
    } else {
        for (i64 i = t * 4; i < out_len; ++i) out[i] = (u8)(w >> (8 * (i - t * 4)));
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

}  // namespace

namespace gvli {
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
//   256  a mixed ragged launch's long rows by the wave that meets them, chunk after chunk (no front workgroups: lean_solo_rows' crews)
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
// 262144 / 524288  [-DGVL_DIAG builds only; masked off otherwise] timing ablations of the lean kernel (WRONG output for rows with
//        indels): no re-alignment / allele bytes (phases A and B as for a SNP-only row); no scan plan either
// 65536  the lean kernel re-reads the runs of a row with indels from memory (never re-aligns the speculative window in LDS)
// 1048576 rows longer than one chunk never take the lean kernel (LONG): the all-purpose kernel as before
// 2097152 realignment from intervals never uses its window (every value looked up in the interval list itself)
// 4194304 tracks are always painted into the scratch track first (no realignment straight from the intervals)
// 8388608 / 16777216  [-DGVL_DIAG builds only] timing ablations of realign_tracks_kernel (NO output): stop behind the walk / behind the window build
// 1073741824 round 4's routing of the other output modes: channel-major one-hot, keep masks, annotations and get_reference run the
//        all-purpose kernel (their lean forms are this round's)
// 536870912 rows of several chunks go without chunk plans (hap_plan_kernel): every chunk-wave of the lean kernel walks its row itself
// and 1 / 2 / 4 = timing ablations (no variants / no stores / no loads) [-DGVL_DIAG builds only].
// (written by gvl_set_debug_flags on one thread while the loader's producer thread reads it: a relaxed atomic)
int g_debug_override = -1;
int debug_flags() {
    static const int flags = [] { const char *e = getenv("GVL_DBG"); return e ? atoi(e) : 0; }();
    const int ov = __atomic_load_n(&g_debug_override, __ATOMIC_RELAXED);
    const int v = ov >= 0 ? ov : flags;
#ifdef GVL_DIAG
    return v;
#else
    return v & ~DBG_ABLATIONS;        // (the timing ablations change the output: diagnostic builds only)
#endif
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

// rows with many variants (mean > 16 per genotype slot, or GVL_DBG=2048): one wave per row instead of one lane
static bool diffs_long_rows(const gvl_static *st) {
    if (debug_flags() & 2048) return true;
    return st->n_geno_offsets > 0 && st->n_geno / st->n_geno_offsets > 16;
}

int fill_diff_args(DiffArgs &D, const gvl_static *st, const gvl_batch *bt, const char *who) {
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

// Stream-ordered scratch (the painter's work lists, the chunk plans of a stand-alone long-row launch) comes from a pool the LIBRARY owns, one per device (created at the first use on that device):
// with the device's default pool every call paid a driver allocation (the default release threshold is 0: 25 of the 45 us of a
// stand-alone painting of BASELINE config 4's batch), and raising THAT pool's threshold would change the allocator for every other
// hipMallocAsync user of the process.  The library's pools keep up to 256 MiB each across synchronisations.
hipError_t pool_alloc(void **p, size_t bytes, hipStream_t s) {
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

}  // namespace gvli

extern "C" {


int gvl_abi_version(void) { return GVL_ABI_VERSION; }
int gvl_set_debug_flags(int flags) { __atomic_store_n(&g_debug_override, flags, __ATOMIC_RELAXED); return GVL_OK; }
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
    if (e == 4) return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: a haplotype with more than 64 entries in a channel is not position-sorted: its records are unspecified");
    if (e == 5) return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: a negative position (positions are u32 below 2^31)");
    if (e == 6) return fail(GVL_ERR_INVALID, "%s", "gvl_svar2_merge: channel offsets (vk_off / dense_range / dense_present_off / allele offsets) leave their arrays: that haplotype got no variants");
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
    d.geno_rec = nullptr; d.slot_rec = nullptr; d.slot_vidx = nullptr;
    int rc = ok ? GVL_OK : fail(GVL_ERR_HIP, "%s", "gvl_static_upload: device allocation / copy failed");
    if (!rc && nv > 0) rc = gvl_pack_variants(d.v_starts, d.ilens, d.alt_offsets, d.alt_alleles, nv, (gvl_vrec *)d.vrec, stream);
    if (!rc && with_layouts && ng > 0 && nv > 0) {
        gvl_grec *g = (gvl_grec *)dev_copy(o, nullptr, (size_t)ng * sizeof(gvl_grec), s, &ok);
        if (ok) { rc = gvl_pack_genotypes(&d, g, stream); if (!rc) d.geno_rec = g; }
    }
    if (!rc && ok && with_layouts && no > 0 && nv > 0 && host->alt_len < (1ll << 32)) {
        gvl_srec *sr = (gvl_srec *)dev_copy(o, nullptr, (size_t)no * GVL_SLOT_RECS * sizeof(gvl_srec), s, &ok);
        if (ok) { rc = gvl_pack_slots(&d, sr, stream); if (!rc) d.slot_rec = sr; }
        if (ok && !rc) {
            int32_t *sv = (int32_t *)dev_copy(o, nullptr, (size_t)no * GVL_SLOT_RECS * sizeof(int32_t), s, &ok);
            if (ok) { rc = gvl_pack_slot_vidx(&d, sv, stream); if (!rc) d.slot_vidx = sv; }
        }
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

__global__ __launch_bounds__(256) void pack_slot_vidx_kernel(const i64 *go_starts, const i64 *go_stops, i64 n_slots, const int *geno_v_idxs, int *out) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_slots * GVL_SLOT_RECS) return;
    const i64 o = t / GVL_SLOT_RECS;
    const int j = (int)(t - o * GVL_SLOT_RECS);
    const i64 s0 = go_starts[o], n = go_stops[o] - s0;
    out[t] = (n >= 0 && j < n && n <= GVL_SLOT_RECS) ? geno_v_idxs[s0 + j] : -1;       // (a slot that overflows its line: read from the CSR's inline records)
}
int gvl_pack_slot_vidx(const gvl_static *st, int32_t *out, void *stream) {
    if (!st || st->n_geno_offsets < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_pack_slot_vidx: bad arguments");
    if (st->n_geno_offsets == 0) return GVL_OK;
    if (!st->geno_o_starts || !st->geno_o_stops || !out || (st->n_geno > 0 && !st->geno_v_idxs)) return fail(GVL_ERR_INVALID, "%s", "gvl_pack_slot_vidx: NULL array");
    const i64 grid = (st->n_geno_offsets * GVL_SLOT_RECS + 255) / 256;
    if (grid > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_pack_slot_vidx: too many slots");
    pack_slot_vidx_kernel<<<dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream>>>((const i64 *)st->geno_o_starts, (const i64 *)st->geno_o_stops,
                                                                                      st->n_geno_offsets, st->geno_v_idxs, out);
    return check_launch("gvl_pack_slot_vidx");
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
    if (bt->out_bounds && bt->out_offsets)
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: out_offsets and out_bounds are exclusive");
    if (bt->output_length < 0 && !bt->out_offsets && !bt->out_bounds)
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: ragged mode needs out_offsets (gvl_hap_offsets) or out_bounds");
    if (bt->output_length > 0x7FFFFF00ll || bt->max_row_len > 0x7FFFFF00ll)
        return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: row length must be < 2^31 - 256");
    if (out->onehot && out->onehot_layout == GVL_ONEHOT_CL && (bt->output_length < 0 || bt->out_offsets || bt->out_bounds))
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
    A.hplan = (const u32 *)bt->hap_plan;
    A.oh_cl = (out->onehot && out->onehot_layout == GVL_ONEHOT_CL) ? 1u : 0u;
    A.slot_vidx = st->slot_vidx;
    A.n_geno_offsets = st->n_geno_offsets;
    A.n_contigs = (int)(st->n_contigs < 0 ? 0 : (st->n_contigs > 0x7FFFFFFFll ? 0x7FFFFFFF : st->n_contigs));
    A.regions = bt->regions; A.regions_stride = bt->regions_stride; A.shifts = bt->shifts;
    A.geno_offset_idx = (const i64 *)bt->geno_offset_idx;
    A.keep = bt->keep; A.keep_offsets = (const i64 *)bt->keep_offsets; A.to_rc = bt->to_rc;
    // (rows at (start, end) pairs -- gvl_batch.out_bounds -- are read through the same pointer with a stride of two)
    const bool ragged_rows = bt->out_offsets || bt->out_bounds;
    A.out_offsets = (const i64 *)(bt->out_bounds ? bt->out_bounds : bt->out_offsets);
    A.oo_shift = bt->out_bounds ? 1 : 0;
    A.fixed_len = ragged_rows ? -1 : bt->output_length;
    A.n_rows = bt->batch * bt->ploidy; A.ploidy = (int)bt->ploidy; A.ploidy_shift = log2_exact(bt->ploidy);
    // longest row: fixed mode -> output_length; caller-supplied offsets -> the
    // caller's max_row_len hint (must bound every row, or longer rows are left
    // partly unwritten)
    i64 ml = ragged_rows ? bt->max_row_len : bt->output_length;
    if (ragged_rows && bt->output_length > ml) ml = bt->output_length;
    if (ml < 0) ml = 0;
    if (pick_chunk(ml, chunks, &A.chunk_len)) return fail(GVL_ERR_INVALID, "%s", "gvl_reconstruct: too many chunks");
    A.row_chunks = *chunks;
    A.ref_only = 0;
    A.dbg = debug_flags();
    A.pad = st->pad_char;
    A.haps = out->haps; A.onehot = out->onehot;
    A.av = out->annot_v_idxs; A.ap = out->annot_ref_pos; A.out_offsets_w = bt->out_bounds ? nullptr : (i64 *)out->out_offsets;
    A.stamps = g_stamps;
    A.async_err = async_err_word();
    const bool annot = out->annot_v_idxs || out->annot_ref_pos;
    const int oh = !out->onehot ? OH_NONE : (out->onehot_layout == GVL_ONEHOT_CL ? OH_CL : OH_LC);
    *variant = oh | ((out->haps != nullptr) ? 4 : 0) | (annot ? 8 : 0);
    return GVL_OK;
}

// Can this batch take the lean kernel?  Row-major one-hot and / or haplotype bytes, fixed-length rows, no keep mask, no
// annotations, the derived layouts present -- and no path-forcing debug flag (those exist to walk the all-purpose
// kernel).  Rows of one chunk: one wave per row, variants from the slot lines.  Longer rows (a multiple of 4 bases, cut
// into 2048-base chunks): one wave per chunk, variants from the CSR's inline records (LONG).
static bool lean_eligible(const gvl_static *st, const gvl_batch *bt, const gvl_out *out, int chunks, int chunk_len) {
    if (!st->ref4 || (!out->onehot && !out->haps)) return false;
    // (annotated haplotypes: bytes + BOTH annotation streams (+ a one-hot in either layout), rows of one chunk, the slot records' variant indices present;
    // only the pipelined kernel has the form)
    // ... rows of several chunks (round 6): the chunked kernel's form, variant indices from the CSR's inline records
    if (out->annot_v_idxs || out->annot_ref_pos) {
        if (!out->annot_v_idxs || !out->annot_ref_pos || !out->haps || (debug_flags() & 1073741824)) return false;
        if (chunks == 1 && (!st->slot_vidx || ((uintptr_t)st->slot_vidx & 15) || (debug_flags() & 67108864))) return false;
    }
    // (channel-major one-hot: the pipelined kernel's form for rows of one chunk, the chunked kernel's for long rows)
    if (out->onehot && out->onehot_layout != GVL_ONEHOT_LC && ((chunks == 1 && (debug_flags() & 67108864)) || (debug_flags() & 1073741824))) return false;
    if (bt->out_offsets) return false;
    // (a keep mask: both arrays or neither; rows of one chunk only, and only the pipelined kernel reads one)
    if ((bt->keep != nullptr) != (bt->keep_offsets != nullptr)) return false;
    // (rows of one chunk: the pipelined kernel reads the mask; longer rows: the chunked kernel's walk and the chunk plans do, round 6)
    if (bt->keep && ((chunks == 1 && (debug_flags() & 67108864)) || (debug_flags() & 1073741824))) return false;
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
    return (debug_flags() & ~(2 | 4 | 32768 | 65536 | 262144 | 524288 | 1048576 | 2097152 | 4194304 | 8388608 | 16777216 | 33554432 | 67108864 | 268435456 | 536870912 | 1073741824)) == 0;
}

// ragged rows (out_offsets) longer than the pipelined form's 2560 bases: the chunked lean kernel's ragged form (<.., LONG, RAGL>)
static bool lean_long_rag_eligible(const gvl_static *st, const gvl_batch *bt, const gvl_out *out, int chunks, int chunk_len) {
    if (!st->ref4 || !st->geno_rec || (!out->onehot && !out->haps) || !bt->out_offsets) return false;
    if (out->annot_v_idxs || out->annot_ref_pos || (out->onehot && out->onehot_layout != GVL_ONEHOT_LC)) return false;
    if ((bt->keep != nullptr) != (bt->keep_offsets != nullptr)) return false;
    if (bt->keep && (debug_flags() & 1073741824)) return false;           // (round 5's routing: rows under a keep mask on the all-purpose kernel)
    if (chunks < 2 || chunk_len != LEAN_MAX_TRIPS * TRIP || (debug_flags() & (1048576 | 16))) return false;
    const i64 n_rows = bt->batch * bt->ploidy;
    if (n_rows <= 0 || n_rows * chunks > 0x7FFFFFF0ll) return false;
    if (st->alt_len >= (1ll << 32) || st->ref_len >= (1ll << 32) - 8192) return false;
    return (debug_flags() & ~(2 | 4 | 32768 | 65536 | 262144 | 524288 | 1048576 | 2097152 | 4194304 | 8388608 | 16777216 | 33554432 | 67108864 | 268435456 | 536870912 | 1073741824)) == 0;
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
    if (out->onehot && out->onehot_layout != GVL_ONEHOT_LC) return false;
    if (out->annot_v_idxs || out->annot_ref_pos) {
        if (!out->annot_v_idxs || !out->annot_ref_pos || !out->haps || !st->slot_vidx || ((uintptr_t)st->slot_vidx & 15) ||
            (debug_flags() & 1073741824)) return false;
    }
    if ((bt->keep != nullptr) != (bt->keep_offsets != nullptr)) return false;
    if (bt->keep && (debug_flags() & 1073741824)) return false;
    const i64 ml = bt->max_row_len > bt->output_length ? bt->max_row_len : bt->output_length;
    const i64 n_rows = bt->batch * bt->ploidy;
    if (n_rows <= 0 || n_rows > 0x7FFFFFF0ll) return false;
    if (ml <= 0) return false;
    if (ml > (i64)PIPE_RAG_MAXT * TRIP) {
        // a batch of MOSTLY short rows with a few long ones (a spliced batch: exons of a few hundred bases and the odd 3' UTR of several
        // thousand): the pipelined kernel still takes it -- a row beyond its 2560 bases is marked and written by its wave's solo path,
        // chunk by chunk -- when the caller's total says the mean row is short (gvl_batch.total_len_hint) and no row is longer than 64 Kb.
        // Anything else with long rows is the chunked kernel's (lean_long_rag_eligible).
        // ... and the batch is large: a wave writes its long rows chunk by chunk at its end, which a launch of a few thousand rows has
        // nothing to hide behind (4 420 exon rows with 4 long ones: 27 us against the all-purpose kernel's 15; 17 372 rows: 34 against 40;
        // 69 586 rows: 89 against 152 -- tools/spliced_bench.py)
        const i64 mixed_t = tune(GVL_TUNE_MIXED_MIN_ROWS);
        const i64 mixed_min = mixed_t > 0 ? mixed_t : 16384;
        if (bt->total_len_hint <= 0 || bt->total_len_hint / n_rows > (i64)PIPE_RAG_MAXT * TRIP / 2 || ml > 65536 || n_rows < mixed_min) return false;
    }
    if (st->alt_len >= (1ll << 32) || st->ref_len >= (1ll << 32) - 8192) return false;
    return (debug_flags() & ~(2 | 4 | 256 | 32768 | 65536 | 262144 | 524288 | 1048576 | 2097152 | 4194304 | 8388608 | 16777216 | 33554432 | 268435456 | 536870912 | 1073741824)) == 0;
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
int gvl_reconstruct(const gvl_static *st, const gvl_batch *bt, const gvl_out *out, void *stream) {
    ReconArgs A;
    int chunks = 1, variant = 0;
    const int rc = fill_recon_args(st, bt, out, A, &chunks, &variant);
    if (rc) return rc;
    if (bt->out_bounds) return launch_recon(A, chunks, variant, stream);      // (scatter write: the all-purpose kernel's row setup reads the pairs)
    if (A.n_rows > 0 && lean_eligible(st, bt, out, chunks, A.chunk_len)) {
        // (rows of one chunk: only the pipelined kernel has the channel-major and annotated forms and reads keep masks)
        const bool pipe_only = chunks == 1 && (A.oh_cl || A.keep || A.av);
        if (chunks == 1 && (lean_pipe_wanted(A.n_rows) || pipe_only) && lean_pipe_compatible(&A, 1)) return launch_lean_rows(&A, 1, stream, 1);
        if (pipe_only) return launch_recon(A, chunks, variant, stream);
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
    bool all_one_chunk_lean = n > 0, all_rag = n > 0 && !(debug_flags() & 67108864), any_cl = false;
    i64 total = 0;
    for (int i = 0; i < n; ++i) {
        chunks[i] = 1; variant[i] = 0;
        const int rc = fill_recon_args(st, &bts[i], &outs[i], A[i], &chunks[i], &variant[i]);
        if (rc) return rc;
        lean[i] = A[i].n_rows > 0 && !bts[i].out_bounds && (lean_eligible(st, &bts[i], &outs[i], chunks[i], A[i].chunk_len) ||
                                      (!lean_rag_eligible(st, &bts[i], &outs[i]) && lean_long_rag_eligible(st, &bts[i], &outs[i], chunks[i], A[i].chunk_len)));
        all_one_chunk_lean = all_one_chunk_lean && lean[i] && chunks[i] == 1;
        any_cl = any_cl || (chunks[i] == 1 && (A[i].oh_cl || A[i].keep || A[i].av));
        all_rag = all_rag && A[i].n_rows > 0 && !bts[i].out_bounds && lean_rag_eligible(st, &bts[i], &outs[i]);
        total += A[i].n_rows;
    }
    if (all_one_chunk_lean && (lean_pipe_wanted(total, n) || any_cl) && lean_pipe_compatible(A, n)) return launch_lean_rows(A, n, stream, 1);
    if (all_rag && lean_pipe_compatible(A, n)) {
        int min_chunks = chunks[0];             // (LeanArgs.max_row_len: the smallest bound; a row is reported against its own batch's, ReconArgs.row_chunks)
        for (int i = 1; i < n; ++i) min_chunks = chunks[i] < min_chunks ? chunks[i] : min_chunks;
        return launch_lean_rows(A, n, stream, min_chunks);
    }
    for (int i = 0; i < n; ++i) {
        int rc;
        if (lean[i] && chunks[i] == 1 && (A[i].oh_cl || A[i].keep || A[i].av))       // (channel-major one-hot, a keep mask: the pipelined kernel's forms, or the all-purpose kernel)
            rc = lean_pipe_compatible(&A[i], 1) ? launch_lean_rows(&A[i], 1, stream, 1) : launch_recon(A[i], chunks[i], variant[i], stream);
        else
            rc = lean[i] ? launch_lean(A[i], chunks[i], stream) : launch_recon(A[i], chunks[i], variant[i], stream);
        if (rc) return rc;
    }
    return GVL_OK;
}

int64_t gvl_hap_plan_bytes(int64_t n_rows, int64_t output_length) {
    int chunks = 1, chunk_len = 0;
    if (n_rows <= 0 || output_length <= 2048 || (output_length & 3) || pick_chunk(output_length, &chunks, &chunk_len)) return 0;
    if (chunks < 2 || chunks > HP_MAX_CHUNKS || chunk_len != LEAN_MAX_TRIPS * TRIP) return 0;
    return hap_plan_bytes(n_rows, chunks);
}

int gvl_hap_plan(const gvl_static *st, const gvl_batch *bt, void *plan, void *stream) {
    if (!st || !bt || !plan) return fail(GVL_ERR_INVALID, "%s", "gvl_hap_plan: NULL argument");
    if (bt->out_offsets || bt->out_bounds) return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_hap_plan: fixed-length rows only");
    if ((bt->keep != nullptr) != (bt->keep_offsets != nullptr)) return fail(GVL_ERR_INVALID, "%s", "gvl_hap_plan: keep and keep_offsets go together");
    if (bt->batch < 0 || bt->ploidy <= 0 || bt->batch * bt->ploidy > 0x7FFFFFF0ll) return fail(GVL_ERR_INVALID, "%s", "gvl_hap_plan: bad batch / ploidy");
    if (bt->batch == 0) return GVL_OK;
    if (gvl_hap_plan_bytes(bt->batch * bt->ploidy, bt->output_length) <= 0)
        return fail(GVL_ERR_UNSUPPORTED, "%s", "gvl_hap_plan: rows of 2 .. 512 chunks of 2048 bases (a multiple of 4 bases) only");
    if (!st->ref_offsets || !st->geno_o_starts || !st->geno_o_stops || !bt->regions || !bt->shifts || !bt->geno_offset_idx || bt->regions_stride < 3)
        return fail(GVL_ERR_INVALID, "%s", "gvl_hap_plan: NULL/invalid array");
    ReconArgs A;
    memset(&A, 0, sizeof(A));
    A.ref_offsets = (const i64 *)st->ref_offsets; A.alt_offsets = (const i64 *)st->alt_offsets;
    A.go_starts = (const i64 *)st->geno_o_starts; A.go_stops = (const i64 *)st->geno_o_stops;
    A.grec = (debug_flags() & 16) ? nullptr : st->geno_rec;       // (without the inline records no chunk is planned: every one flagged)
    A.n_geno_offsets = st->n_geno_offsets;
    A.n_contigs = (int)(st->n_contigs < 0 ? 0 : (st->n_contigs > 0x7FFFFFFFll ? 0x7FFFFFFF : st->n_contigs));
    A.regions = bt->regions; A.regions_stride = bt->regions_stride; A.shifts = bt->shifts;
    A.geno_offset_idx = (const i64 *)bt->geno_offset_idx;
    A.keep = bt->keep; A.keep_offsets = (const i64 *)bt->keep_offsets;       // (a plan made under a keep mask is that mask's)
    A.fixed_len = bt->output_length;
    A.n_rows = bt->batch * bt->ploidy; A.ploidy = (int)bt->ploidy; A.ploidy_shift = log2_exact(bt->ploidy);
    A.dbg = debug_flags();
    int chunks = 1;
    if (pick_chunk(bt->output_length, &chunks, &A.chunk_len)) return fail(GVL_ERR_INVALID, "%s", "gvl_hap_plan: too many chunks");
    return launch_hap_plan(A, chunks, (u32 *)plan, stream);
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
    A.row_chunks = chunks;
    A.ref_only = 1;
    A.pad = st->pad_char;
    A.haps = out; A.onehot = onehot;
    if (n_rows > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference: batch too large");
    // Rows of at most 2560 bases with the packed reference present: the pipelined lean kernel's ragged form with no slot lines (a row
    // inside its contig is one window read + the stream; rows that need padding run the all-purpose body at their wave's end).
    // GVL_DBG 2^30 / 67108864: the all-purpose kernel for every row, as until round 4.
    if (st->ref4 && !((uintptr_t)st->ref4 & 15) && max_row_len > 0 && max_row_len <= (i64)PIPE_RAG_MAXT * TRIP && n_rows <= 0x7FFFFFF0ll &&
        st->ref_len < (1ll << 32) - 8192 &&
        (debug_flags() & ~(2 | 4 | 32768 | 65536 | 33554432 | 268435456 | 536870912)) == 0) {
        A.ref4 = st->ref4;
        A.dbg = debug_flags();
        A.async_err = async_err_word();
        return launch_lean_rows(&A, 1, stream, chunks);
    }
    // Longer rows (`with_seqs("reference")` at Enformer length): the chunked lean kernel's ragged form, a wave per 2048-base chunk with
    // no walk at all.  GVL_DBG 2^30 / 1048576: the all-purpose kernel.
    if (st->ref4 && chunks >= 2 && A.chunk_len == LEAN_MAX_TRIPS * TRIP && n_rows * (i64)chunks <= 0x7FFFFFF0ll &&
        st->ref_len < (1ll << 32) - 8192 && (debug_flags() & ~(2 | 4 | 32768 | 65536 | 262144 | 33554432 | 67108864 | 268435456 | 536870912)) == 0) {
        A.ref4 = st->ref4;
        A.dbg = debug_flags();
        A.async_err = async_err_word();
        return launch_lean(A, chunks, stream);
    }
    return launch_recon(A, chunks, (onehot ? OH_LC : OH_NONE) | (out ? 4 : 0), stream);
}

int gvl_get_reference_many(const gvl_static *st, const gvl_ref_batch *bs, int32_t n, void *stream) {
    if (!st || n < 0 || n > GVL_MANY_MAX || (n > 0 && !bs)) return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference_many: bad arguments (n <= GVL_MANY_MAX)");
    // one grid over the group when every batch could take the pipelined route on its own and they share a shape
    ReconArgs A[GVL_MANY_MAX];
    bool grouped = n >= 2 && st->ref4 && !((uintptr_t)st->ref4 & 15) && st->ref_offsets && st->ref_len < (1ll << 32) - 8192 &&
                   (debug_flags() & ~(2 | 4 | 32768 | 65536 | 33554432 | 268435456 | 536870912)) == 0;
    int min_chunks = 0x7FFFFFFF;
    // every batch is validated before anything is launched (as gvl_reconstruct_many does), with gvl_get_reference's own checks
    for (int i = 0; i < n; ++i) {
        const gvl_ref_batch &b = bs[i];
        if (b.n_rows < 0) return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference_many: bad arguments");
        if (b.n_rows == 0) continue;
        if ((st->ref_len > 0 && !st->ref) || !st->ref_offsets || !b.regions || !b.out_offsets || b.regions_stride < 3 || (!b.out && !b.onehot))
            return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference_many: NULL/invalid array");
        if (b.max_row_len < 0 || b.max_row_len > 0x7FFFFF00ll) return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference_many: bad max_row_len");
        if (b.n_rows > 0x7FFFFFFFll) return fail(GVL_ERR_INVALID, "%s", "gvl_get_reference_many: batch too large");
    }
    grouped = grouped && (st->ref_len == 0 || st->ref);
    for (int i = 0; i < n && grouped; ++i) {
        const gvl_ref_batch &b = bs[i];
        if (b.n_rows <= 0 || b.n_rows > 0x7FFFFFF0ll || !b.regions || !b.out_offsets || b.regions_stride < 3 || (!b.out && !b.onehot) ||
            b.max_row_len <= 0 || b.max_row_len > (i64)PIPE_RAG_MAXT * TRIP) { grouped = false; break; }
        ReconArgs &R = A[i];
        memset(&R, 0, sizeof(R));
        R.ref = st->ref; R.ref_len = st->ref_len; R.ref_offsets = (const i64 *)st->ref_offsets; R.ref4 = st->ref4;
        R.regions = b.regions; R.regions_stride = b.regions_stride;
        R.n_contigs = (int)(st->n_contigs < 0 ? 0 : (st->n_contigs > 0x7FFFFFFFll ? 0x7FFFFFFF : st->n_contigs));
        R.to_rc = b.to_rc; R.out_offsets = (const i64 *)b.out_offsets; R.fixed_len = -1;
        R.n_rows = b.n_rows; R.ploidy = 1; R.ploidy_shift = 0;
        int chunks = 1;
        if (pick_chunk(b.max_row_len, &chunks, &R.chunk_len)) { grouped = false; break; }
        R.row_chunks = chunks;
        min_chunks = chunks < min_chunks ? chunks : min_chunks;
        R.ref_only = 1; R.pad = st->pad_char; R.haps = b.out; R.onehot = b.onehot;
        R.dbg = debug_flags(); R.async_err = async_err_word();
    }
    if (grouped && lean_pipe_compatible(A, n)) return launch_lean_rows(A, n, stream, min_chunks);
    for (int i = 0; i < n; ++i) {
        const int rc = gvl_get_reference(st, bs[i].regions, bs[i].regions_stride, bs[i].n_rows, bs[i].out_offsets, bs[i].max_row_len, bs[i].to_rc,
                                         bs[i].out, bs[i].onehot, stream);
        if (rc) return rc;
    }
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

// The process-wide knobs an epoch table's LAYOUT and CONTENTS depend on, read ONCE per table fill (loader_knobs_now) and kept with
// the table: gvl_loader_start_epoch, the fill and every submit of the epoch take them from that one snapshot, and a table that was
// prefetched under other knobs is simply filled again.  (They were re-read at three different times: a gvl_set_tuning /
// gvl_set_debug_flags between a prefetch and the epoch's start made the part offsets disagree.)
struct LoaderKnobs {
    i64 track_cap_mb, hap_cap_mb;   // GVL_TUNE_TRACK_PLAN_MAX_MB / GVL_TUNE_HAP_PLAN_MAX_MB (0: built in)
    int ragged_per_group;           // GVL_TUNE_RAGGED_SIZING == 1
    int dbg;                        // the GVL_DBG bits that decide what a table holds: 8, 2048, 134217728, 268435456, 536870912
    bool operator==(const LoaderKnobs &o) const {
        return track_cap_mb == o.track_cap_mb && hap_cap_mb == o.hap_cap_mb && ragged_per_group == o.ragged_per_group && dbg == o.dbg;
    }
};
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
    i64 *e_rag_offs, *e_rag_sizes;          // ragged rows: every batch's row offsets ((bs * P + 1) per batch) and {total, longest} of the epoch (or NULL)
    u8 *e_hplan; i64 e_hplan_row;           // rows of several chunks: the epoch's chunk plans (gvl_hap_plan over the whole table; or NULL) and the bytes of one row's
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
    LoaderKnobs cur_kn, pf_kn;    // the knobs the running epoch's / the prefetched table was laid out and filled under
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
static LoaderKnobs loader_knobs_now() {
    LoaderKnobs k;
    k.track_cap_mb = tune(GVL_TUNE_TRACK_PLAN_MAX_MB); k.hap_cap_mb = tune(GVL_TUNE_HAP_PLAN_MAX_MB);
    k.ragged_per_group = tune(GVL_TUNE_RAGGED_SIZING) == 1 ? 1 : 0;
    k.dbg = debug_flags() & (8 | 2048 | 134217728 | 268435456 | 536870912);
    return k;
}
static i64 loader_track_plan_bytes(const gvl_loader_config *cfg, i64 n, const LoaderKnobs &kn) {
    if (cfg->output_length <= 2048) return 0;
    int chunks = 1, chunk_len = 0;
    if (pick_chunk(cfg->output_length, &chunks, &chunk_len) || chunks <= 1) return 0;
    const i64 cap_t = kn.track_cap_mb;
    const i64 cap = (cap_t > 0 ? cap_t : 512) << 20;
    const i64 b = track_plan_bytes(n * cfg->ploidy, chunks);
    return b <= cap ? b : 0;
}

// the haplotype kernel's chunk plans of an epoch's rows (fixed-length rows of several chunks the lean kernel takes, either one-hot layout).
// Under a cap of their own, 64 MB (gvl_set_tuning(GVL_TUNE_HAP_PLAN_MAX_MB)): a plan is 336 bytes per 2048-base chunk (272 + the annotated rows' annex), written an epoch
// ahead and read once.  While the epoch's plans sit in the 256 MB Infinity Cache next to its other inputs a chunk-wave starts from
// two reads instead of a walk (config 4's step: 58.5 against 61.7 us); beyond it they cost more than the walks they save (256
// samples x 16 regions = 142 MB of plans: 71 us per step with them, 58 without: profiles/r05_cfg4_plans_vs_size.txt).
static i64 loader_hap_plan_bytes(const gvl_loader_config *cfg, i64 n, const LoaderKnobs &kn) {
    if (cfg->output_length <= 2048) return 0;          // (annotated epochs too, since round 6: the plans' annex)
    const i64 cap_t = kn.hap_cap_mb;
    const i64 cap = (cap_t > 0 ? cap_t : 64) << 20;
    const i64 b = gvl_hap_plan_bytes(n * cfg->ploidy, cfg->output_length);
    return b <= cap ? b : 0;
}

static bool loader_ragged(const gvl_loader_config *c) { return c->output_length == -1; }
// ragged rows are sized once per EPOCH (in the table) instead of once per group of batches -- whenever the group sizing would be
// used; gvl_set_tuning(GVL_TUNE_RAGGED_SIZING, 1): per group, as in round 4 (GVL_DBG & 134217728: per batch, as before that)
static bool loader_long_rows(const gvl_static *st, const LoaderKnobs &kn) {      // (diffs_long_rows under the snapshot's GVL_DBG 2048)
    return (kn.dbg & 2048) || (st->n_geno_offsets > 0 && st->n_geno / st->n_geno_offsets > 16);
}
static bool loader_epoch_sizing(const gvl_static *st, const gvl_loader_config *c, const LoaderKnobs &kn) {
    return loader_ragged(c) && !loader_long_rows(st, kn) && !(kn.dbg & 134217728) && !kn.ragged_per_group;
}
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

static int64_t loader_table_layout(const gvl_loader_config *cfg, int64_t n, int64_t *part_offsets, const LoaderKnobs &kn);
int64_t gvl_loader_table_bytes(const gvl_loader_config *cfg, int64_t n, int64_t *part_offsets) {
    // (under the knobs as they are NOW: size the table right before the gvl_loader_start_epoch / _prefetch_epoch that fills it, from
    // the same thread -- those take their own snapshot, and the two agree unless another thread turned a knob in between)
    return loader_table_layout(cfg, n, part_offsets, loader_knobs_now());
}
static int64_t loader_table_layout(const gvl_loader_config *cfg, int64_t n, int64_t *part_offsets, const LoaderKnobs &kn) {
    if (!cfg || cfg->ploidy <= 0 || cfg->batch_size <= 0 || n < 0) return -1;
    const i64 P = cfg->ploidy;
    const i64 nb = (n + cfg->batch_size - 1) / cfg->batch_size;
    const bool tr = cfg->n_tracks > 0;
    const i64 sizes[GVL_LOADER_TABLE_PARTS] = {16 * n, 8 * n * P, 4 * n * P, n * P, 8 * nb,
                                               tr ? 8 * (n + nb) : 0, tr ? 8 * (cfg->batch_size * P + 1) : 0,
                                               tr ? loader_track_plan_bytes(cfg, n, kn) : 0, loader_hap_plan_bytes(cfg, n, kn),
                                               loader_ragged(cfg) ? 8 * nb * (cfg->batch_size * P + 1 + 2) : 0};
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
static int loader_fill_table(gvl_loader *ld, const int64_t *order, i64 n, int32_t drop_last, void *table, u64 counter, hipStream_t s,
                             const LoaderKnobs &kn) {
    const gvl_loader_config &c = ld->cfg;
    int64_t po[GVL_LOADER_TABLE_PARTS + 1];
    po[GVL_LOADER_TABLE_PARTS] = loader_table_layout(&c, n, po, kn);
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
    if (po[9] > po[8] && !(kn.dbg & 536870912)) {
        // rows of several chunks: every row's chunk plans, once per epoch (the batches' launches get pointers into them)
        gvl_batch gb;
        memset(&gb, 0, sizeof(gb));
        gb.regions = t_regions; gb.regions_stride = 4; gb.shifts = t_shifts; gb.geno_offset_idx = (const int64_t *)t_goi;
        gb.batch = n_used; gb.ploidy = c.ploidy; gb.output_length = c.output_length; gb.max_row_len = c.output_length;
        const int rc_p = gvl_hap_plan(&ld->st, &gb, base + po[8], s);
        if (rc_p) return rc_p;
    }
    if (po[10] > po[9] && loader_epoch_sizing(&ld->st, &c, kn)) {
        // ragged rows: every batch's row lengths -> offsets + {total, longest row}, for the whole epoch (rows cut to the slots'
        // capacity are reported, never silent)
        gvl_batch gb;
        memset(&gb, 0, sizeof(gb));
        gb.regions = t_regions; gb.regions_stride = 4; gb.shifts = t_shifts; gb.geno_offset_idx = (const int64_t *)t_goi;
        gb.batch = n_used; gb.ploidy = c.ploidy;
        DiffArgs D;
        int rc0 = fill_diff_args(D, &ld->st, &gb, "gvl_loader(ragged sizing)");
        if (rc0) return rc0;
        D.q_starts = gb.regions + 1; D.q_ends = gb.regions + 2; D.q_stride = gb.regions_stride;
        D.diffs = nullptr; D.output_length = -1; D.lengths = nullptr;
        D.len_cap = c.max_row_len; D.async_err = async_err_word();
        const i64 rpb = bs * c.ploidy;
        i64 *const offs = (i64 *)(base + po[9]);
        i64 *const sizes = offs + n_batches * (rpb + 1);
        hap_lengths_epoch_kernel<<<dim3((unsigned)((D.n_rows + 255) / 256)), dim3(256), 0, s>>>(D, gb.regions, (i64)gb.regions_stride, rpb, offs);
        rc0 = check_launch("gvl_loader(ragged lengths, epoch)");
        if (rc0) return rc0;
        hap_scan_epoch_kernel<<<dim3((unsigned)n_batches), dim3(256), 0, s>>>(offs, sizes, rpb, D.n_rows);
        rc0 = check_launch("gvl_loader(ragged offsets, epoch)");
        if (rc0) return rc0;
    }
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
        if (po[8] > po[7] && !(kn.dbg & (8 | 268435456))) {
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
            rc3 = launch_track_plan(TA, ph, pe, chunks, c.output_length, bs, s);
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
    const LoaderKnobs kn = loader_knobs_now();
    const int rc = loader_fill_table(ld, order, n, drop_last, table, epoch + 1, (hipStream_t)stream, kn);
    if (rc) return rc;
    if (hipEventRecord(ld->pf_ready, (hipStream_t)stream) != hipSuccess)
        return fail(GVL_ERR_HIP, "%s", "gvl_loader_prefetch_epoch: hipEventRecord failed");
    ld->pf_order = order; ld->pf_n = n; ld->pf_drop_last = drop_last; ld->pf_table = table; ld->pf_counter = epoch + 1;
    ld->pf_kn = kn;
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
    // (... under the knobs as they are now: a table prefetched under others has another layout / other contents and is filled again)
    const LoaderKnobs kn = loader_knobs_now();
    const bool prefetched = ld->pf_valid && ld->pf_order == order && ld->pf_n == n && ld->pf_drop_last == drop_last &&
                            ld->pf_table == table && ld->pf_counter == counter && table != ld->cur_table && ld->pf_kn == kn;
    ld->cur_kn = kn;
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
        loader_table_layout(&c, n, po, kn);
        u8 *base = (u8 *)table;
        ld->e_regions = (int *)(base + po[0]);
        ld->e_goi = (i64 *)(base + po[1]);
        ld->e_shifts = (int *)(base + po[2]);
        ld->e_to_rc = base + po[3];
        ld->e_seeds = (u64 *)(base + po[4]);
        ld->e_track_offsets = (i64 *)(base + po[5]);
        ld->e_out_offsets = (i64 *)(base + po[6]);
        ld->e_hplan = (loader_hap_plan_bytes(&c, n, kn) > 0 && !(kn.dbg & 536870912)) ? base + po[8] : nullptr;
        ld->e_rag_offs = loader_epoch_sizing(&ld->st, &c, kn) ? (i64 *)(base + po[9]) : nullptr;
        {
            const i64 nb_used = drop_last ? n / c.batch_size : (n + c.batch_size - 1) / c.batch_size;
            ld->e_rag_sizes = ld->e_rag_offs ? ld->e_rag_offs + nb_used * (c.batch_size * c.ploidy + 1) : nullptr;
        }
        ld->e_hplan_row = gvl_hap_plan_bytes(1, c.output_length);
        const i64 wsb = c.n_tracks > 0 ? loader_track_plan_bytes(&c, n, kn) : 0;
        int cl = 0;
        ld->e_chunks = 1;
        if (wsb > 0) (void)pick_chunk(c.output_length, &ld->e_chunks, &cl);
        const bool planned = wsb > 0 && !(kn.dbg & (8 | 268435456));
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
        const int rc = loader_fill_table(ld, order, n, drop_last, table, counter, s, kn);
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
        bt.hap_plan = ld->e_hplan ? ld->e_hplan + j * c.batch_size * c.ploidy * ld->e_hplan_row : nullptr;
        gvl_out &oc = ocs[m];
        memset(&oc, 0, sizeof(oc));
        oc.haps = o.haps; oc.onehot = o.onehot; oc.onehot_layout = c.onehot_layout; oc.out_offsets = o.out_offsets;
        oc.annot_v_idxs = o.annot_v_idxs; oc.annot_ref_pos = o.annot_ref_pos;
        if (loader_ragged(&c)) {
            // row lengths and offsets on the device (rows cut to the slot's capacity are reported, never silent);
            // the reconstruct launch below then reads them -- no host round trip.  One sizing per GROUP (below) unless the
            // dataset's rows take the wave-per-row length kernel (or GVL_DBG & 134217728: per batch, as before round 4)
            if (ld->e_rag_offs) {
                // sized with the epoch's table: the launch reads its rows' offsets THERE (nothing in front of it); the consumer's
                // copies go to the slots behind it
                grp.offs[m] = (i64 *)o.out_offsets; grp.sizes[m] = (i64 *)o.sizes;
                bt.out_offsets = (const int64_t *)(ld->e_rag_offs + j * (c.batch_size * c.ploidy + 1));
                oc.out_offsets = nullptr;
                continue;
            }
            if (loader_long_rows(&ld->st, ld->cur_kn) || (ld->cur_kn.dbg & 134217728)) {
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
    if (ld->e_rag_offs && m > 0) {
        const i64 rpb = c.batch_size * c.ploidy, j0 = g * ld->G;
        hap_offsets_copy_kernel<<<dim3(4, (unsigned)m), dim3(256), 0, s>>>(grp, ld->e_rag_offs + j0 * (rpb + 1), ld->e_rag_sizes + 2 * j0, rpb);
        rc = check_launch("gvl_loader(ragged offsets to the slots)");
        if (rc) return rc;
    }
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
