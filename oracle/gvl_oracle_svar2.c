/*
 * gvl_oracle_svar2.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE): the SVAR2 two-source
 * variant provider, restated from /root/reference/src/svar2/mod.rs,
 * src/reconstruct/mod.rs:584-826 and src/tracks/mod.rs:668-860.
 *
 * The reference keeps a haplotype's variants in two channels -- `var_key` (per-haplotype calls, CSR
 * `vk_off`) and `dense` (per-query window `dense_range` + LSB-first presence bits) -- as (position,
 * 32-bit key) pairs and decodes a key with the third-party crate `svar2-codec` (git dependency of
 * d-laub/genoray @ 66ba734b, NOT vendored under /root/reference; its bit layout is published nowhere
 * in the reference).  What the reference itself states is what a decoded key IS
 * (`decode_alt`, src/svar2/mod.rs:17-30):
 *     Inline  { alt }  -> (v_diff = alt.len() - 1, allele = alt)
 *     PureDel { ilen } -> (v_diff = ilen,          allele = EMPTY)
 *     Lookup  { row }  -> (v_diff = len - 1,       allele = lut_bytes[lut_off[row] .. lut_off[row + 1]])
 * This file therefore takes the channels DECODED: entry e of a channel has v_diff `ilen[e]` and the allele
 * `alt_bytes[alt_off[e] .. alt_off[e + 1])` (an empty allele = a pure deletion).  oracle.py's `decode_alt`
 * turns symbolic keys (the reference's test inputs: encode_alt_inline / encode_pure_del / encode_lookup) into
 * that form.  Nothing here guesses the codec's bits.
 *
 * Parity status: pinned by the reference's Rust known-answer tests (src/svar2/mod.rs:598-700,
 * src/reconstruct/mod.rs:1540-1813, src/tracks/mod.rs:2480-2567; no 200-case golden exists for these entry
 * points), transcribed in tests/svar2_kats.py; by 192 haplotypes of the reference's INDEPENDENT Python consensus
 * (tests/test_svar2_reconstruct.py:66-93, run at fixture-generation time: tests/golden/pyref_svar2_consensus.npz)
 * -- and, through the shared cores, by everything that pins
 * gvlo_reconstruct_row / gvlo_realign_track_row (the SVAR2 drivers call the same `reconstruct_haplotype_core` /
 * `shift_and_realign_track_core` the SVAR1 drivers call).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GVLO_EXPORT __attribute__((visibility("default")))

/* the pinned cores (gvl_oracle.c, gvl_oracle_tracks.c) */
void gvlo_reconstruct_row(int64_t n_variants, const int32_t *v_idxs, const int32_t *v_starts,
                          const int32_t *ilens, int64_t shift, const uint8_t *alt_alleles,
                          const int64_t *alt_offsets, const uint8_t *ref, int64_t ref_len,
                          int64_t ref_start, uint8_t *out, int64_t length, uint8_t pad_char,
                          const uint8_t *keep, int32_t *av, int32_t *ap);
void gvlo_realign_track_row(int64_t n_variants, const int32_t *v_idxs, const int32_t *v_starts,
                            const int32_t *ilens, int64_t shift, const float *track, int64_t track_len,
                            int64_t query_start, float *out, int64_t length, const double *params,
                            const uint8_t *keep, int64_t strategy, uint64_t base_seed, uint64_t query,
                            uint64_t hap);

static inline int64_t min64(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

/* src/svar2/mod.rs:35-39: LSB-first presence bit */
static inline int present_bit(const uint8_t *dense_present, int64_t base_bit, int64_t j)
{
    int64_t bit = base_bit + j;
    return (dense_present[bit / 8] >> (bit % 8)) & 1;
}

/* One merged entry: position (u32 like the Rust's `as u32`), and which channel entry it is
 * (src >= 0: var_key entry src; src < 0: dense entry -(src + 1)). */
typedef struct { uint32_t pos; int64_t src; } merged_t;

/* src/svar2/mod.rs:45-72 `merge_hap`: var_key entries [vk_lo, vk_hi), then the present entries of the dense
 * window [ds, de), then a STABLE sort by position (var_key stays ahead of dense on ties).  Returns the number
 * of merged entries written to `out` (capacity (vk_hi - vk_lo) + (de - ds)). */
GVLO_EXPORT int64_t gvlo_svar2_merge_hap(const int32_t *vk_pos, int64_t vk_lo, int64_t vk_hi,
                                         const int32_t *dense_pos, int64_t ds, int64_t de,
                                         const uint8_t *dense_present, int64_t base_bit,
                                         uint32_t *out_pos, int64_t *out_src)
{
    int64_t n = 0;
    for (int64_t i = vk_lo; i < vk_hi; i++) { out_pos[n] = (uint32_t)vk_pos[i]; out_src[n] = i; n++; }
    for (int64_t j = ds, k = 0; j < de; j++, k++)
        if (present_bit(dense_present, base_bit, k)) { out_pos[n] = (uint32_t)dense_pos[j]; out_src[n] = -(j + 1); n++; }
    /* stable insertion sort by position (sort_by_key is stable, mod.rs:70) */
    for (int64_t i = 1; i < n; i++) {
        uint32_t p = out_pos[i];
        int64_t s = out_src[i];
        int64_t j = i - 1;
        while (j >= 0 && out_pos[j] > p) { out_pos[j + 1] = out_pos[j]; out_src[j + 1] = out_src[j]; j--; }
        out_pos[j + 1] = p;
        out_src[j + 1] = s;
    }
    return n;
}

typedef struct {
    const int32_t *vk_pos, *vk_ilen; const int64_t *vk_alt_off, *vk_off;
    const int32_t *dense_pos, *dense_ilen; const int64_t *dense_alt_off;
    const int32_t *dense_range; const uint8_t *dense_present; const int64_t *dense_present_off;
    const uint8_t *alt_bytes;
} channels_t;

/* decode_alt of merged entry `src` as data: v_diff and (when asked) the allele's [start, end) in alt_bytes */
static inline int64_t entry_ilen(const channels_t *c, int64_t src)
{
    return src >= 0 ? (int64_t)c->vk_ilen[src] : (int64_t)c->dense_ilen[-(src + 1)];
}
static inline void entry_allele(const channels_t *c, int64_t src, int64_t *a0, int64_t *a1)
{
    if (src >= 0) { *a0 = c->vk_alt_off[src]; *a1 = c->vk_alt_off[src + 1]; }
    else { *a0 = c->dense_alt_off[-(src + 1)]; *a1 = c->dense_alt_off[-(src + 1) + 1]; }
}

static int64_t merge_for(const channels_t *c, int64_t k, int64_t query, uint32_t **pos, int64_t **src, int64_t *cap)
{
    int64_t vk_lo = c->vk_off[k], vk_hi = c->vk_off[k + 1];
    int64_t ds = c->dense_range[2 * query], de = c->dense_range[2 * query + 1];
    int64_t need = (vk_hi - vk_lo) + (de - ds);
    if (need > *cap) {
        *cap = need + 64;
        *pos = (uint32_t *)realloc(*pos, sizeof(uint32_t) * (size_t)*cap);
        *src = (int64_t *)realloc(*src, sizeof(int64_t) * (size_t)*cap);
    }
    return gvlo_svar2_merge_hap(c->vk_pos, vk_lo, vk_hi, c->dense_pos, ds, de, c->dense_present,
                                c->dense_present_off[k], *pos, *src);
}

/* src/svar2/mod.rs:78-160 `hap_diffs_svar2`: get_diffs_sparse's query-clipped branch over the merged list,
 * with the optional exonic filter.  diffs: i32 (n_q, ploidy). */
GVLO_EXPORT void gvlo_hap_diffs_svar2(
    const int32_t *regions, int64_t regions_stride, int64_t n_q, int64_t ploidy,
    const int32_t *vk_pos, const int32_t *vk_ilen, const int64_t *vk_off,
    const int32_t *dense_pos, const int32_t *dense_ilen, const int32_t *dense_range,
    const uint8_t *dense_present, const int64_t *dense_present_off, int32_t filter_exonic, int32_t *diffs)
{
    channels_t c = {vk_pos, vk_ilen, NULL, vk_off, dense_pos, dense_ilen, NULL, dense_range, dense_present,
                    dense_present_off, NULL};
    uint32_t *mpos = NULL; int64_t *msrc = NULL; int64_t cap = 0;
    for (int64_t k = 0; k < n_q * ploidy; k++) {
        int64_t query = k / ploidy;
        int64_t n = merge_for(&c, k, query, &mpos, &msrc, &cap);
        diffs[k] = 0;
        if (n == 0) continue;                                         /* mod.rs:121-123 */
        int64_t q_start = regions[query * regions_stride + 1], q_end = regions[query * regions_stride + 2];
        int64_t ref_idx = q_start, acc = 0;
        for (int64_t m = 0; m < n; m++) {                             /* mod.rs:128-152 */
            int64_t v_start = (int64_t)mpos[m];
            int64_t v_ilen = entry_ilen(&c, msrc[m]);
            int64_t v_end = v_start - min64(v_ilen, 0) + 1;
            if (filter_exonic && (v_start < q_start || v_end > q_end)) continue;
            if (v_end <= q_start) continue;
            if (v_start >= q_end) break;
            if (v_start >= q_start && v_start < ref_idx) continue;
            ref_idx = max64(ref_idx, v_end);
            if (v_ilen < 0) v_ilen += max64(q_start - v_start - 1, 0);
            v_ilen += max64(v_end - q_end, 0);
            acc += v_ilen;
        }
        diffs[k] = (int32_t)acc;
    }
    free(mpos); free(msrc);
}

/* Build the per-haplotype table the shared cores read (what the Rust does with its `provide` closure,
 * src/reconstruct/mod.rs:709-735): merged entry v -> (pos, v_diff, allele); an EMPTY allele (pure DEL) is replaced
 * by the anchor base ref[pos] (mod.rs:712-733).  Returns the number of entries kept (the exonic filter of
 * mod.rs:699-706 drops entries first). */
typedef struct { int32_t *v_idxs, *v_starts, *ilens; int64_t *alt_offsets; uint8_t *alleles; int64_t cap_v, cap_a; } hap_table;

static int64_t build_table(const channels_t *c, int64_t n, const uint32_t *mpos, const int64_t *msrc, int filter_exonic,
                           int64_t ref_start, int64_t ref_end, const uint8_t *contig_ref, int64_t contig_len,
                           uint8_t pad_char, int want_alleles, hap_table *t)
{
    if (n + 1 > t->cap_v) {
        t->cap_v = n + 65;
        t->v_idxs = (int32_t *)realloc(t->v_idxs, sizeof(int32_t) * (size_t)t->cap_v);
        t->v_starts = (int32_t *)realloc(t->v_starts, sizeof(int32_t) * (size_t)t->cap_v);
        t->ilens = (int32_t *)realloc(t->ilens, sizeof(int32_t) * (size_t)t->cap_v);
        t->alt_offsets = (int64_t *)realloc(t->alt_offsets, sizeof(int64_t) * (size_t)t->cap_v);
    }
    int64_t nv = 0, na = 0;
    t->alt_offsets[0] = 0;
    for (int64_t m = 0; m < n; m++) {
        int64_t v_start = (int64_t)mpos[m];
        int64_t v_ilen = entry_ilen(c, msrc[m]);
        if (filter_exonic) {                                          /* retain: mod.rs:699-706 */
            int64_t v_end = v_start - min64(v_ilen, 0) + 1;
            if (!(v_start >= ref_start && v_end <= ref_end)) continue;
        }
        t->v_idxs[nv] = (int32_t)nv;
        t->v_starts[nv] = (int32_t)mpos[m];
        t->ilens[nv] = (int32_t)v_ilen;
        if (want_alleles) {
            int64_t a0, a1;
            entry_allele(c, msrc[m], &a0, &a1);
            int64_t len = a1 - a0;
            int64_t need = na + (len > 0 ? len : 1);
            if (need > t->cap_a) {
                t->cap_a = 2 * need + 64;
                t->alleles = (uint8_t *)realloc(t->alleles, (size_t)t->cap_a);
            }
            if (len > 0) {
                memcpy(t->alleles + na, c->alt_bytes + a0, (size_t)len);
                na += len;
            } else {
                /* pure DEL: the anchor base (mod.rs:712-733; the Rust slice panics past the contig end) */
                t->alleles[na++] = (v_start >= 0 && v_start < contig_len) ? contig_ref[v_start] : pad_char;
            }
        }
        nv++;
        t->alt_offsets[nv] = na;
    }
    return nv;
}

/* src/reconstruct/mod.rs:619-826 `reconstruct_haplotypes_from_svar2` (serial form: rows are independent).
 * out_bounds i64 (n_work, 2): [start, end) of every row in `out` (scatter write). */
GVLO_EXPORT void gvlo_reconstruct_haplotypes_from_svar2(
    uint8_t *out, const int64_t *out_bounds, const int32_t *regions, int64_t regions_stride, int64_t n_q,
    int64_t ploidy, const int32_t *shifts,
    const int32_t *vk_pos, const int32_t *vk_ilen, const int64_t *vk_alt_off, const int64_t *vk_off,
    const int32_t *dense_pos, const int32_t *dense_ilen, const int64_t *dense_alt_off,
    const int32_t *dense_range, const uint8_t *dense_present, const int64_t *dense_present_off,
    const uint8_t *alt_bytes, const uint8_t *ref, const int64_t *ref_offsets, uint8_t pad_char,
    int32_t filter_exonic)
{
    channels_t c = {vk_pos, vk_ilen, vk_alt_off, vk_off, dense_pos, dense_ilen, dense_alt_off, dense_range,
                    dense_present, dense_present_off, alt_bytes};
    uint32_t *mpos = NULL; int64_t *msrc = NULL; int64_t cap = 0;
    hap_table t = {NULL, NULL, NULL, NULL, NULL, 0, 0};
    for (int64_t k = 0; k < n_q * ploidy; k++) {
        int64_t query = k / ploidy;                                   /* mod.rs:662-663 */
        const int32_t *reg = regions + query * regions_stride;
        int64_t c_s = ref_offsets[reg[0]], c_e = ref_offsets[reg[0] + 1];
        int64_t n = merge_for(&c, k, query, &mpos, &msrc, &cap);
        int64_t nv = build_table(&c, n, mpos, msrc, filter_exonic, reg[1], reg[2], ref + c_s, c_e - c_s, pad_char, 1, &t);
        int64_t o_s = out_bounds[2 * k], o_e = out_bounds[2 * k + 1];
        gvlo_reconstruct_row(nv, t.v_idxs, t.v_starts, t.ilens, (int64_t)shifts[k], t.alleles, t.alt_offsets,
                             ref + c_s, c_e - c_s, (int64_t)reg[1], out + o_s, o_e - o_s, pad_char, NULL, NULL, NULL);
    }
    free(mpos); free(msrc);
    free(t.v_idxs); free(t.v_starts); free(t.ilens); free(t.alt_offsets); free(t.alleles);
}

/* src/tracks/mod.rs:705-860 `shift_and_realign_tracks_from_svar2` (no exonic filter there; `query_seed`
 * nullable: the FlankSample seed's query component, mod.rs:754-760). */
GVLO_EXPORT void gvlo_realign_tracks_from_svar2(
    float *out, const int64_t *out_offsets, const int32_t *regions, int64_t regions_stride, int64_t n_q,
    int64_t ploidy, const int32_t *shifts,
    const int32_t *vk_pos, const int32_t *vk_ilen, const int64_t *vk_off,
    const int32_t *dense_pos, const int32_t *dense_ilen, const int32_t *dense_range,
    const uint8_t *dense_present, const int64_t *dense_present_off,
    const float *tracks, const int64_t *track_offsets, const double *params, int64_t strategy,
    uint64_t base_seed, const int64_t *query_seed)
{
    channels_t c = {vk_pos, vk_ilen, NULL, vk_off, dense_pos, dense_ilen, NULL, dense_range, dense_present,
                    dense_present_off, NULL};
    uint32_t *mpos = NULL; int64_t *msrc = NULL; int64_t cap = 0;
    hap_table t = {NULL, NULL, NULL, NULL, NULL, 0, 0};
    for (int64_t k = 0; k < n_q * ploidy; k++) {
        int64_t query = k / ploidy, hap = k % ploidy;
        const int32_t *reg = regions + query * regions_stride;
        int64_t n = merge_for(&c, k, query, &mpos, &msrc, &cap);
        int64_t nv = build_table(&c, n, mpos, msrc, 0, 0, 0, NULL, 0, 0, 0, &t);
        uint64_t q_seed = query_seed ? (uint64_t)query_seed[query] : (uint64_t)query;
        int64_t t_s = track_offsets[query], t_e = track_offsets[query + 1];
        gvlo_realign_track_row(nv, t.v_idxs, t.v_starts, t.ilens, (int64_t)shifts[k], tracks + t_s, t_e - t_s,
                               (int64_t)reg[1], out + out_offsets[k], out_offsets[k + 1] - out_offsets[k], params,
                               NULL, strategy, base_seed, q_seed, (uint64_t)hap);
    }
    free(mpos); free(msrc);
    free(t.v_idxs); free(t.v_starts); free(t.ilens); free(t.alt_offsets); free(t.alleles);
}
