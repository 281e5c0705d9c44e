/*
 * gvl_oracle_tracks.c -- CPU ORACLE, track half (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference's track realignment path (SURVEY.md 8 row a12,
 * BASELINE config 4): interval painting, shift-and-realign to a haplotype, the five
 * insertion-fill strategies and the PRNG behind FlankSample.  Pinned by the reference's
 * frozen goldens (tests/golden/ref_shift_and_realign_tracks_sparse.npz,
 * ref_intervals_to_tracks.npz, ref_prng_xorshift64.npz, ref_prng_hash4.npz) in
 * tests/test_oracle_tracks.py.  Same rules as gvl_oracle.c: nothing under
 * genvarloader_amd/ may use it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GVLO_EXPORT __attribute__((visibility("default")))

static inline int64_t min64(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

enum { REPEAT_5P = 0, REPEAT_5P_NORM = 1, CONSTANT = 2, FLANK_SAMPLE = 3, INTERPOLATE = 4 };

/* src/tracks/mod.rs:31-36 */
GVLO_EXPORT uint64_t gvlo_xorshift64(uint64_t x)
{
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return x;
}

/* src/tracks/mod.rs:48-54 */
GVLO_EXPORT uint64_t gvlo_hash4(uint64_t a, uint64_t b, uint64_t c, uint64_t d)
{
    uint64_t h = a;
    h = gvlo_xorshift64(h ^ b);
    h = gvlo_xorshift64(h ^ c);
    h = gvlo_xorshift64(h ^ d);
    return h;
}

/* src/tracks/mod.rs:87-190 */
static void apply_insertion_fill(float *out, int64_t out_idx, int64_t writable, int64_t v_len,
                                 const float *track, int64_t track_len, int64_t v_rel_pos,
                                 int64_t strategy, const double *params, uint64_t base_seed,
                                 uint64_t query, uint64_t hap)
{
    if (strategy == REPEAT_5P) {
        float val = track[v_rel_pos];
        for (int64_t i = 0; i < writable; i++) out[out_idx + i] = val;
    } else if (strategy == REPEAT_5P_NORM) {
        float val = track[v_rel_pos] / (float)v_len; /* f32/f32, see mod.rs:110-115 */
        for (int64_t i = 0; i < writable; i++) out[out_idx + i] = val;
    } else if (strategy == CONSTANT) {
        float val = (float)params[0];
        for (int64_t i = 0; i < writable; i++) out[out_idx + i] = val;
    } else if (strategy == FLANK_SAMPLE) {
        int64_t width = (int64_t)params[0];
        int64_t pool_lo = max64(v_rel_pos - width, 0);
        int64_t pool_hi = min64(v_rel_pos + width, track_len - 1);
        uint64_t pool = (uint64_t)(pool_hi - pool_lo + 1);
        for (int64_t i = 0; i < writable; i++) {
            uint64_t seed = gvlo_hash4(base_seed, query, hap, (uint64_t)(out_idx + i));
            out[out_idx + i] = track[pool_lo + (int64_t)(seed % pool)];
        }
    } else if (strategy == INTERPOLATE) {
        int64_t order = (int64_t)params[0];
        int64_t k = (order + 1 + 1) / 2;
        int64_t n = 2 * k;
        double *xs = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
        double *ys = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
        for (int64_t j = 0; j < k; j++) {
            xs[j] = -(double)j;
            ys[j] = (double)track[max64(v_rel_pos - j, 0)];
        }
        for (int64_t j = 0; j < k; j++) {
            xs[k + j] = (double)v_len + (double)j;
            ys[k + j] = (double)track[min64(v_rel_pos + 1 + j, track_len - 1)];
        }
        for (int64_t i = 0; i < writable; i++) {
            double x = (double)i, acc = 0.0;
            for (int64_t a = 0; a < n; a++) {
                double term = ys[a];
                for (int64_t b = 0; b < n; b++) {
                    if (b == a) continue;
                    term *= (x - xs[b]) / (xs[a] - xs[b]);
                }
                acc += term;
            }
            out[out_idx + i] = (float)acc;
        }
        free(xs);
        free(ys);
    }
}

/* src/tracks/mod.rs:224-406 (core) driven by the SVAR1 provider :429-471 */
GVLO_EXPORT void gvlo_realign_track_row(
    int64_t n_variants, const int32_t *v_idxs, const int32_t *v_starts, const int32_t *ilens,
    int64_t shift, const float *track, int64_t track_len, int64_t query_start, float *out,
    int64_t length, const double *params, const uint8_t *keep, int64_t strategy,
    uint64_t base_seed, uint64_t query, uint64_t hap)
{
    if (n_variants == 0) { /* mod.rs:240-246 */
        for (int64_t i = 0; i < length; i++) out[i] = track[i];
        return;
    }
    int64_t track_idx = 0, out_idx = 0, shifted = 0;
    for (int64_t v = 0; v < n_variants; v++) {
        if (keep && !keep[v]) continue;
        int64_t variant = v_idxs[v];
        int64_t v_rel_pos = (int64_t)v_starts[variant] - query_start;
        int64_t v_diff = ilens[variant];
        int64_t v_rel_end = v_rel_pos - min64(v_diff, 0) + 1;
        if (v_diff < 0 && v_rel_pos < 0 && v_rel_end >= 0) { /* :271-274 */
            track_idx = v_rel_end;
            continue;
        }
        if (v_rel_pos < track_idx) continue; /* :277-279 */
        int64_t v_len = max64(v_diff, 0) + 1;
        if (shifted < shift) { /* :285-308 */
            int64_t dist = v_rel_pos - track_idx;
            if (shifted + dist + v_len < shift) {
                continue;
            } else if (shifted + dist >= shift) {
                track_idx += shift - shifted;
                shifted = shift;
            } else {
                int64_t a0 = shift - shifted - dist;
                shifted = shift;
                if (a0 == v_len) {
                    track_idx = v_rel_end;
                    continue;
                }
                track_idx = v_rel_pos;
                v_len -= a0;
            }
        }
        if (v_diff == 0) continue; /* SNPs do not move tracks: :312-314 */
        int64_t n = v_rel_pos - track_idx;
        if (out_idx + n >= length) break;
        for (int64_t i = 0; i < n; i++) out[out_idx + i] = track[track_idx + i];
        out_idx += n;
        int64_t writable = min64(v_len, length - out_idx);
        if (v_diff > 0 && strategy != REPEAT_5P) {
            apply_insertion_fill(out, out_idx, writable, v_len, track, track_len, v_rel_pos, strategy,
                                 params, base_seed, query, hap);
        } else {
            float val = track[v_rel_pos];
            for (int64_t i = 0; i < writable; i++) out[out_idx + i] = val;
        }
        out_idx += writable;
        track_idx = v_rel_end;
        if (out_idx >= length) break;
    }
    if (shifted < shift) {
        track_idx += shift - shifted;
        track_idx = min64(track_idx, track_len);
    }
    int64_t unfilled = length - out_idx;
    if (unfilled > 0) {
        int64_t w = min64(unfilled, track_len - track_idx);
        int64_t end = out_idx;
        if (w > 0) {
            for (int64_t i = 0; i < w; i++) out[out_idx + i] = track[track_idx + i];
            end = out_idx + w;
        }
        for (int64_t i = end; i < length; i++) out[i] = 0.0f;
    }
}

/* src/tracks/mod.rs:495-667 (batch driver; rows independent) */
GVLO_EXPORT void gvlo_realign_tracks_batch(
    float *out, const int64_t *out_offsets, const int32_t *regions, int64_t regions_stride,
    int64_t batch, int64_t ploidy, const int32_t *shifts, const int64_t *geno_offset_idx,
    const int32_t *geno_v_idxs, const int64_t *go_starts, const int64_t *go_stops,
    const int32_t *v_starts, const int32_t *ilens, const float *tracks,
    const int64_t *track_offsets, const double *params, const uint8_t *keep,
    const int64_t *keep_offsets, int64_t strategy, uint64_t base_seed)
{
    for (int64_t q = 0; q < batch; q++) {
        int64_t t_s = track_offsets[q], t_e = track_offsets[q + 1];
        int64_t q_start = regions[q * regions_stride + 1];
        for (int64_t h = 0; h < ploidy; h++) {
            int64_t k = q * ploidy + h;
            int64_t o_idx = geno_offset_idx[k];
            int64_t o_s = go_starts[o_idx], o_e = go_stops[o_idx];
            const uint8_t *kp = (keep && keep_offsets) ? keep + keep_offsets[k] : NULL;
            gvlo_realign_track_row(o_e - o_s, geno_v_idxs + o_s, v_starts, ilens, shifts[k],
                                   tracks + t_s, t_e - t_s, q_start, out + out_offsets[k],
                                   out_offsets[k + 1] - out_offsets[k], params, kp, strategy,
                                   base_seed, (uint64_t)q, (uint64_t)h);
        }
    }
}

/* src/intervals.rs:19-126 */
GVLO_EXPORT void gvlo_intervals_to_tracks(
    const int64_t *offset_idxs, const int32_t *starts, int64_t n_queries, const int32_t *itv_starts,
    const int32_t *itv_ends, const float *itv_values, const int64_t *itv_offsets, float *out,
    const int64_t *out_offsets)
{
    memset(out, 0, sizeof(float) * (size_t)out_offsets[n_queries]);
    for (int64_t q = 0; q < n_queries; q++) {
        int64_t idx = offset_idxs[q];
        int64_t s0 = itv_offsets[idx], e0 = itv_offsets[idx + 1];
        if (s0 == e0) continue;
        float *o = out + out_offsets[q];
        int64_t length = out_offsets[q + 1] - out_offsets[q];
        int64_t qs = starts[q];
        for (int64_t i = s0; i < e0; i++) {
            int64_t start = (int64_t)itv_starts[i] - qs, end = (int64_t)itv_ends[i] - qs;
            if (start >= length) break;
            int64_t s = max64(start, 0), e = min64(end, length);
            for (int64_t j = s; j < e; j++) o[j] = itv_values[i];
        }
    }
}
