// gvl_tracks.hip -- BASELINE config 4's track half: the interval painter, realign_tracks_kernel (from a painted track or straight from the
// intervals), the rows' plans, and their C-ABI entry points (src/intervals.rs:19-126, src/tracks/mod.rs:224-406, src/ffi/mod.rs:2551-2672).
#include "gvl_internal.inc"

namespace {
#include "gvl_dev.inc"
#include "gvl_diffs.inc"
#include "gvl_tracks_kernels.inc"
}  // namespace

extern "C" {

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
        if (pool_alloc((void **)&scratch, (size_t)n_intervals * sizeof(int), s) != hipSuccess) {
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
        pool_alloc((void **)&todo, (size_t)(n_queries * n_chunks) * sizeof(PaintTodo), s) != hipSuccess) {
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
        pool_alloc((void **)&todo, (size_t)(n_queries * n_chunks) * sizeof(PaintTodo), s) != hipSuccess) {
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
                               int2 *plan_hdr = nullptr, i32x4 *plan_ent = nullptr, bool plan_make = false, int to_shift = 0);
int gvl_realign_tracks(const gvl_static *st, const gvl_batch *bt, const float *tracks,
                       const int64_t *track_offsets, const double *params, int64_t strategy_id,
                       uint64_t base_seed, float *out, void *stream) {
    return realign_tracks_impl(st, bt, tracks, track_offsets, params, strategy_id, base_seed, nullptr, out, stream);
}
static int realign_tracks_impl(const gvl_static *st, const gvl_batch *bt, const float *tracks,
                               const int64_t *track_offsets, const double *params, int64_t strategy_id,
                               uint64_t base_seed, const u64 *seed_ptr, float *out, void *stream, const PaintSrcArgs *ps,
                               int2 *plan_hdr, i32x4 *plan_ent, bool plan_make, int to_shift) {
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
    A.query_seed = (const i64 *)bt->query_seed;
    A.to_shift = to_shift;
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
    // tracks straight from the intervals with the rows' plans at hand: the fast kernel (chunks it does not take call the general body;
    // GVL_DBG & 1073741824: the general kernel for every chunk, as in round 4)
    if (ps && A.plan_hdr && A.plan_ent && A.out_offsets && !(A.dbg & (1073741824 | GVL_ABL(8388608) | GVL_ABL(16777216) | 2097152)))
        realign_paint_kernel<<<dim3((unsigned)grid, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream>>>(A, *ps);
    else if (ps) realign_tracks_kernel<true><<<dim3((unsigned)grid, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream>>>(A, *ps);
    else realign_tracks_kernel<false><<<dim3((unsigned)grid, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream>>>(A, PaintSrcArgs());
    return check_launch("gvl_realign_tracks");
}


// scratch layout of gvl_tracks_batch: track_offsets i64 (batch + 1) | out_offsets i64 (batch * ploidy + 1) |
// chunk records (16 B x batch * chunks) | scratch tracks f32 (batch * stride) | row plans (track_plan_bytes)
static void tracks_scratch_parts(i64 batch, i64 ploidy, i64 stride, i64 part[6]) {
    const i64 n_chunks = (stride + 2047) / 2048;
    const i64 sz[5] = {16 * (batch + 1), 8 * (batch * ploidy + 1), batch * n_chunks * (i64)sizeof(PaintTodo), 4 * batch * stride,
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

int gvl_tracks_batch(const gvl_static *st, const gvl_batch *bt, const int64_t *offset_idxs, const gvl_track_set *tracks,
                     int32_t n_tracks, const double *params, int64_t strategy_id, uint64_t base_seed, float *out,
                     int64_t out_track_stride, void *scratch, int64_t scratch_stride, void *stream) {
    return tracks_batch_impl(st, bt, offset_idxs, tracks, n_tracks, params, strategy_id, base_seed, nullptr, out, out_track_stride,
                             scratch, scratch_stride, stream);
}
}  // extern "C"

int gvli::tracks_batch_impl(const gvl_static *st, const gvl_batch *bt, const int64_t *offset_idxs, const gvl_track_set *tracks,
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
    // (every track realigned straight from its intervals: nobody reads a scratch track, so only the LENGTHS are needed -- stored as
    // (0, length) pairs, no scan launch)
    bool all_fused = !(debug_flags() & 4194304);
    for (int t = 0; t < n_tracks && all_fused; ++t) {
        const gvl_track_set &T = tracks[t];
        all_fused = T.tile_complete != 0 && T.bkt_offsets && T.bkt_base && T.bkt_lo && T.bkt_hi && !(debug_flags() & 1024) && T.itv_pmax_ends;
    }
    int to_shift = 0;
    if (pre_track_offsets && pre_out_offsets) {
        track_offsets = const_cast<i64 *>(pre_track_offsets);
        out_offsets = const_cast<i64 *>(pre_out_offsets);
    } else {
        const bool per_hap = P == 1 || P == 2 || P == 4;      // a wave per (query, haplotype), 4 / P queries per workgroup
        const i64 grid = per_hap ? (B * P + 3) / 4 : (B * WAVE + 255) / 256;
        // (the kernel also writes the K + 1 row offsets, grid-stride: any grid covers them)
        to_shift = all_fused ? 1 : 0;
        track_lengths_kernel<<<dim3((unsigned)grid), dim3(256), 0, s>>>(D, bt->regions, (i64)bt->regions_stride, B, L, track_offsets, out_offsets, to_shift);
        rc = check_launch("gvl_tracks_batch(lengths)");
        if (rc) return rc;
        if (!all_fused) {
            offsets_scan_kernel<<<dim3(1), dim3(1024), 0, s>>>(track_offsets, B, (i64 *)nullptr);
            rc = check_launch("gvl_tracks_batch(scan)");
            if (rc) return rc;
        }
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
                                     seed_ptr, out + (i64)t * out_track_stride, stream, &ps, plan_hdr, plan_ent, !plan_made, to_shift);
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

int gvli::launch_track_plan(const TrackArgs &TA, int2 *hdr, i32x4 *ent, int chunks, i64 fixed_len, i64 rows_per_batch_q, void *stream) {
    const i64 wgrid = (TA.n_rows + 3) / 4;
    track_plan_kernel<<<dim3((unsigned)wgrid), dim3(256), 0, (hipStream_t)stream>>>(TA, hdr, ent, chunks, fixed_len, rows_per_batch_q);
    return check_launch("gvl_loader_start_epoch(row plans)");
}

extern "C" {



}  // extern "C"
