/*
 * gvl_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's haplotype-reconstruction hot path
 * (GenVarLoader v0.42.0, /root/reference).  It exists so that tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg have something to
 * check the HIP path against and to time on the host cores.  Nothing under
 * genvarloader_amd/ may import, link or call it: the product path is the HIP
 * library and fails loudly when that is missing.
 *
 * Parity status: PINNED for haplotype bytes / diffs / reference fetch / RC /
 * exonic mask / annotations -- tests/test_oracle_golden.py replays the
 * reference's own frozen goldens (tests/parity/golden/ NAME.npz, 200 cases each,
 * generated from the reference's Rust build) and its Rust/Python known-answer
 * tests through this file.  One-hot is "parity unpinned": the reference has no
 * one-hot encoder (users call third-party seqpro.DNA.ohe, seqpro==0.22.0, not
 * vendored); the definition restated here is
 *     out[..., j, a] = (x[..., j] == "ACGT"[a])      uint8
 * (docs/source/index.md:109-119 is the only call site).
 *
 * Every function cites the reference file:line it follows.  All arithmetic is
 * i64 like the Rust.  Where the Rust would panic (out-of-contract input, e.g.
 * a slice past the contig end) this file clamps instead of crashing; goldens
 * stay inside the contract (tests/parity/strategies.py:558-685).
 */
#include <limits.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GVLO_EXPORT __attribute__((visibility("default")))

static inline int64_t min64(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------- */
/* a1/a2: one haplotype row.  src/reconstruct/mod.rs:39-256 (core) and
 * :280-319 (SVAR1 provider); numpy mirror _dataset/_genotypes.py:125-248.   */
/* ------------------------------------------------------------------------- */
GVLO_EXPORT void gvlo_reconstruct_row(
    int64_t n_variants, const int32_t *v_idxs, const int32_t *v_starts,
    const int32_t *ilens, int64_t shift, const uint8_t *alt_alleles,
    const int64_t *alt_offsets, const uint8_t *ref, int64_t ref_len,
    int64_t ref_start, uint8_t *out, int64_t length, uint8_t pad_char,
    const uint8_t *keep, int32_t *av, int32_t *ap)
{
    int64_t ref_idx = ref_start, out_idx = 0, shifted = 0;

    /* leading pad absorbs shift first: mod.rs:68-83 */
    if (ref_idx < 0) {
        int64_t raw = -ref_idx;
        shifted = min64(shift, raw);
        int64_t n = raw - shifted;
        int64_t e = min64(n, length); /* Rust would panic if n > length */
        if (e > 0) memset(out, pad_char, (size_t)e); /* out[..n].fill(pad_char) */
        for (int64_t j = 0; (av || ap) && j < e; j++) {
            if (av) av[j] = -1;
            if (ap) ap[j] = -1;
        }
        out_idx += n;
        ref_idx = 0;
    }

    for (int64_t v = 0; v < n_variants; v++) {
        if (keep && !keep[v]) continue; /* mod.rs:86-90 */
        int64_t variant = v_idxs[v];
        int64_t v_pos = v_starts[variant];
        int64_t v_diff = ilens[variant];
        const uint8_t *allele = alt_alleles + alt_offsets[variant];
        int64_t v_len = alt_offsets[variant + 1] - alt_offsets[variant];
        int64_t v_ref_end = v_pos - min64(0, v_diff) + 1; /* mod.rs:96 */

        /* DEL spanning the window start: mod.rs:99-102 */
        if (v_pos < ref_start && v_diff < 0 && v_ref_end >= ref_start) {
            ref_idx = v_ref_end;
            continue;
        }
        /* first ALT wins (bcftools consensus rule): mod.rs:108-110 */
        if (v_pos < ref_idx) continue;

        /* shift consumption: mod.rs:115-146 */
        int64_t skip = 0;
        if (shifted < shift) {
            int64_t dist = v_pos - ref_idx;
            if (shifted + dist + v_len < shift) {
                continue;
            } else if (shifted + dist >= shift) {
                ref_idx += shift - shifted;
                shifted = shift;
            } else {
                skip = shift - shifted - dist;
                shifted = shift;
                if (skip == v_len) {
                    ref_idx = v_ref_end;
                    continue;
                }
                ref_idx = v_pos;
            }
        }
        allele += skip;
        v_len -= skip;

        /* reference run up to the variant: mod.rs:153-175 */
        int64_t n = v_pos - ref_idx;
        if (out_idx + n >= length) break; /* NB ">=" */
        if (ref_idx >= 0 && ref_idx + n <= ref_len) {
            /* copy_from_slice(&ref_[ref_idx..ref_idx + n]): mod.rs:164 */
            if (n > 0) memcpy(out + out_idx, ref + ref_idx, (size_t)n);
        } else { /* out of contract (the Rust slice would panic): clamp */
            for (int64_t j = 0; j < n; j++) {
                int64_t r = ref_idx + j;
                out[out_idx + j] = (r >= 0 && r < ref_len) ? ref[r] : pad_char;
            }
        }
        for (int64_t j = 0; (av || ap) && j < n; j++) {
            if (av) av[out_idx + j] = -1;
            if (ap) ap[out_idx + j] = (int32_t)(ref_idx + j);
        }
        out_idx += n;

        /* the allele, truncated to the space left: mod.rs:178-190 */
        int64_t w = min64(v_len, length - out_idx);
        if (w > 0) memcpy(out + out_idx, allele, (size_t)w);
        for (int64_t j = 0; (av || ap) && j < w; j++) {
            if (av) av[out_idx + j] = (int32_t)variant;
            if (ap) ap[out_idx + j] = (int32_t)v_pos;
        }
        out_idx += w;
        ref_idx = v_ref_end;
        if (out_idx >= length) break;
    }

    /* residual shift: mod.rs:200-205 */
    if (shifted < shift) {
        ref_idx += shift - shifted;
        ref_idx = min64(ref_idx, ref_len);
    }

    /* tail: reference to contig end, then right pad: mod.rs:209-255 */
    int64_t unfilled = length - out_idx;
    if (unfilled > 0) {
        int64_t w = min64(unfilled, ref_len - ref_idx);
        int64_t end = out_idx;
        if (w > 0) {
            memcpy(out + out_idx, ref + ref_idx, (size_t)w);
            for (int64_t j = 0; (av || ap) && j < w; j++) {
                if (av) av[out_idx + j] = -1;
                if (ap) ap[out_idx + j] = (int32_t)(ref_idx + j);
            }
            end = out_idx + w;
        }
        if (end < length) memset(out + end, pad_char, (size_t)(length - end));
        for (int64_t j = end; (av || ap) && j < length; j++) {
            if (av) av[j] = -1;
            if (ap) ap[j] = INT32_MAX;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* a6/a7/a8: reverse / reverse-complement.  src/reverse.rs:25-69.            */
/* ------------------------------------------------------------------------- */
static inline void rc_row(uint8_t *row, int64_t n)
{
    /* reverse, then b ^= (isAT & 0x15) ^ (isCG & 0x04): reverse.rs:45-53 */
    for (int64_t i = 0, j = n - 1; i < j; i++, j--) {
        uint8_t t = row[i];
        row[i] = row[j];
        row[j] = t;
    }
    for (int64_t i = 0; i < n; i++) {
        uint8_t v = row[i];
        uint8_t at = (uint8_t)(-(uint8_t)((v == 'A') | (v == 'T')));
        uint8_t cg = (uint8_t)(-(uint8_t)((v == 'C') | (v == 'G')));
        row[i] = v ^ (at & 21) ^ (cg & 4);
    }
}

GVLO_EXPORT void gvlo_rc_rows(uint8_t *data, const int64_t *offsets,
                              const uint8_t *to_rc, int64_t n_rows)
{
    for (int64_t i = 0; i < n_rows; i++)
        if (to_rc[i]) rc_row(data + offsets[i], offsets[i + 1] - offsets[i]);
}

/* reverse.rs:75-84 */
GVLO_EXPORT void gvlo_rc_bounded_rows(uint8_t *data, const int64_t *bounds,
                                      const uint8_t *to_rc, int64_t n_rows)
{
    for (int64_t i = 0; i < n_rows; i++)
        if (to_rc[i])
            rc_row(data + bounds[2 * i], bounds[2 * i + 1] - bounds[2 * i]);
}

/* reverse.rs:25-38, 4-byte element instantiation (f32 tracks, i32 annots) */
GVLO_EXPORT void gvlo_reverse_rows_4(uint32_t *data, const int64_t *offsets,
                                     const uint8_t *to_rc, int64_t n_rows)
{
    for (int64_t r = 0; r < n_rows; r++) {
        if (!to_rc[r]) continue;
        uint32_t *row = data + offsets[r];
        int64_t n = offsets[r + 1] - offsets[r];
        for (int64_t i = 0, j = n - 1; i < j; i++, j--) {
            uint32_t t = row[i];
            row[i] = row[j];
            row[j] = t;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* a3: batch driver.  src/reconstruct/mod.rs:348-583.  Rows are independent
 * (disjoint out slices); `n_threads` > 1 hands rows to a pthread pool the
 * way rayon's par_iter hands them to its pool (mod.rs:424-539).             */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint8_t *out;
    const int64_t *out_offsets;
    const int32_t *regions;
    int64_t regions_stride;
    int64_t batch, ploidy;
    const int32_t *shifts;
    const int64_t *geno_offset_idx;
    const int64_t *go_starts, *go_stops;
    const int32_t *geno_v_idxs, *v_starts, *ilens;
    const uint8_t *alt_alleles;
    const int64_t *alt_offsets;
    const uint8_t *ref;
    const int64_t *ref_offsets;
    uint8_t pad_char;
    const uint8_t *keep;
    const int64_t *keep_offsets;
    int32_t *av, *ap;
    const uint8_t *to_rc; /* fused entry only (ffi/mod.rs:842-853) */
    /* one-hot stage (a10), fused per row for the timed baseline */
    uint8_t *onehot;
} batch_args;

static uint32_t g_oh_lut[256];
static pthread_once_t g_oh_once = PTHREAD_ONCE_INIT;
static void oh_lut_init(void)
{
    /* little-endian dword {b=='A', b=='C', b=='G', b=='T'} */
    g_oh_lut['A'] = 0x00000001u;
    g_oh_lut['C'] = 0x00000100u;
    g_oh_lut['G'] = 0x00010000u;
    g_oh_lut['T'] = 0x01000000u;
}

static void onehot_row(const uint8_t *in, int64_t n, uint8_t *out)
{
    pthread_once(&g_oh_once, oh_lut_init);
    for (int64_t j = 0; j < n; j++) {
        uint32_t d = g_oh_lut[in[j]];
        memcpy(out + 4 * j, &d, 4);
    }
}

static void batch_row(const batch_args *a, int64_t k)
{
    int64_t query = k / a->ploidy; /* mod.rs:380-381 */
    int64_t o_idx = a->geno_offset_idx[k];
    int64_t o_s = a->go_starts[o_idx], o_e = a->go_stops[o_idx];
    const uint8_t *keep = NULL;
    if (a->keep && a->keep_offsets) keep = a->keep + a->keep_offsets[k];
    const int32_t *reg = a->regions + query * a->regions_stride;
    int64_t c_idx = reg[0];
    int64_t c_s = a->ref_offsets[c_idx], c_e = a->ref_offsets[c_idx + 1];
    int64_t out_s = a->out_offsets[k], out_e = a->out_offsets[k + 1];
    gvlo_reconstruct_row(o_e - o_s, a->geno_v_idxs + o_s, a->v_starts, a->ilens,
                         (int64_t)a->shifts[k], a->alt_alleles, a->alt_offsets,
                         a->ref + c_s, c_e - c_s, (int64_t)reg[1],
                         a->out + out_s, out_e - out_s, a->pad_char, keep,
                         a->av ? a->av + out_s : NULL,
                         a->ap ? a->ap + out_s : NULL);
    if (a->to_rc && a->to_rc[k]) rc_row(a->out + out_s, out_e - out_s);
    if (a->onehot) onehot_row(a->out + out_s, out_e - out_s, a->onehot + 4 * out_s);
}

/* Persistent worker pool.  rayon's global pool (sized once from RAYON_NUM_THREADS,
 * _threads.py:102-115) keeps its workers parked between par_iter calls; so does this
 * one: workers sleep on a condition variable, a job is (fn, n rows, chunk) with an
 * atomic cursor, the caller works too and then waits for the stragglers.  Rows are
 * handed out in chunks of n / (threads * 8) (at least 8) like rayon's adaptive splitting
 * of the row range (reconstruct/mod.rs:424-539).  Between jobs that arrive back to back
 * nobody touches the mutex: workers spin on the generation, check in through an atomic
 * count, and only the last one signals the poster. */
typedef struct {
    const void *args;
    void (*fn)(const void *, int64_t);
    int64_t n;
    int64_t chunk;
    int64_t next; /* atomic cursor */
} pool_job;

static pthread_mutex_t g_job_mu = PTHREAD_MUTEX_INITIALIZER; /* one job at a time */
static struct {
    pthread_mutex_t mu;
    pthread_cond_t cv_work, cv_done;
    pthread_t *th;
    int n_workers;      /* threads besides the caller */
    uint64_t generation; /* bumped per job */
    int pending;         /* workers that have not finished the current job */
    int stop;
    pool_job *job;
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER,
            NULL, 0, 0, 0, 0, NULL};

static void job_drain(pool_job *job)
{
    for (;;) {
        int64_t s = __atomic_fetch_add(&job->next, job->chunk, __ATOMIC_RELAXED);
        if (s >= job->n) break;
        int64_t e = min64(s + job->chunk, job->n);
        for (int64_t k = s; k < e; k++) job->fn(job->args, k);
    }
}

static void *pool_worker(void *born_at)
{
    /* the job generation when this worker was created (gvlo_pool_resize holds the mutex while it creates
     * workers, so no job can be posted in between): a worker born into a pool that has already run jobs must
     * not mistake the old generation count for a new job -- g_pool.job is NULL then */
    uint64_t seen = (uint64_t)(uintptr_t)born_at;
    for (;;) {
        /* like rayon's workers: look for the next job for a while before going to sleep (a condition-variable wake
         * costs tens of microseconds; batches arrive back to back).  The mutex is only taken to sleep: with a few
         * hundred workers, every one of them locking it twice per job WAS the job. */
        for (int spin = 0; spin < 20000; spin++) {
            if (__atomic_load_n(&g_pool.generation, __ATOMIC_ACQUIRE) != seen || __atomic_load_n(&g_pool.stop, __ATOMIC_ACQUIRE)) break;
            __builtin_ia32_pause();
        }
        if (__atomic_load_n(&g_pool.generation, __ATOMIC_ACQUIRE) == seen && !__atomic_load_n(&g_pool.stop, __ATOMIC_ACQUIRE)) {
            pthread_mutex_lock(&g_pool.mu);
            while (!g_pool.stop && g_pool.generation == seen) pthread_cond_wait(&g_pool.cv_work, &g_pool.mu);
            pthread_mutex_unlock(&g_pool.mu);
        }
        if (__atomic_load_n(&g_pool.stop, __ATOMIC_ACQUIRE)) break;
        /* (every worker is counted in `pending` of every job, so the poster cannot return -- and its job cannot go out
         * of scope, nor the next one be posted -- before this worker has checked in below) */
        seen = __atomic_load_n(&g_pool.generation, __ATOMIC_ACQUIRE);
        pool_job *job = __atomic_load_n(&g_pool.job, __ATOMIC_ACQUIRE);
        job_drain(job);
        if (__atomic_sub_fetch(&g_pool.pending, 1, __ATOMIC_ACQ_REL) == 0) {
            pthread_mutex_lock(&g_pool.mu);          /* (under the mutex: the poster checks `pending` under it before it sleeps) */
            pthread_cond_signal(&g_pool.cv_done);
            pthread_mutex_unlock(&g_pool.mu);
        }
    }
    return NULL;
}

static void pool_shutdown_locked(void)
{
    __atomic_store_n(&g_pool.stop, 1, __ATOMIC_RELEASE);
    pthread_cond_broadcast(&g_pool.cv_work);
    pthread_mutex_unlock(&g_pool.mu);
    for (int t = 0; t < g_pool.n_workers; t++) pthread_join(g_pool.th[t], NULL);
    pthread_mutex_lock(&g_pool.mu);
    free(g_pool.th);
    g_pool.th = NULL;
    g_pool.n_workers = 0;
    __atomic_store_n(&g_pool.stop, 0, __ATOMIC_RELEASE);
}

/* (Re)size the pool to `n_threads` (the caller counts as one). */
GVLO_EXPORT int gvlo_pool_resize(int n_threads)
{
    int want = n_threads > 1 ? n_threads - 1 : 0;
    pthread_mutex_lock(&g_pool.mu);
    if (want != g_pool.n_workers) {
        if (g_pool.n_workers) pool_shutdown_locked();
        if (want) {
            g_pool.th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)want);
            int started = 0;
            for (int t = 0; g_pool.th && t < want; t++)
                if (pthread_create(&g_pool.th[started], NULL, pool_worker, (void *)(uintptr_t)g_pool.generation) == 0) started++;
            g_pool.n_workers = started;
        }
    }
    int n = g_pool.n_workers + 1;
    pthread_mutex_unlock(&g_pool.mu);
    return n;
}

static void run_rows(const void *args, void (*fn)(const void *, int64_t),
                     int64_t n, int n_threads)
{
    if (n_threads <= 1 || n <= 1) {
        for (int64_t k = 0; k < n; k++) fn(args, k);
        return;
    }
    pthread_mutex_lock(&g_job_mu);
    int have = gvlo_pool_resize(n_threads);
    /* (at least 8 rows per grab: with hundreds of threads n / (threads * 8) is one or two rows, and the atomic cursor's cache
     * line becomes the job) */
    pool_job job = {args, fn, n, max64(n >= 64 ? 8 : 1, n / ((int64_t)have * 8)), 0};
    pthread_mutex_lock(&g_pool.mu);
    __atomic_store_n(&g_pool.job, &job, __ATOMIC_RELEASE);
    __atomic_store_n(&g_pool.pending, g_pool.n_workers, __ATOMIC_RELEASE);
    __atomic_store_n(&g_pool.generation, g_pool.generation + 1, __ATOMIC_RELEASE);
    pthread_cond_broadcast(&g_pool.cv_work);
    pthread_mutex_unlock(&g_pool.mu);
    job_drain(&job);
    for (int spin = 0; spin < 20000 && __atomic_load_n(&g_pool.pending, __ATOMIC_ACQUIRE) > 0; spin++) __builtin_ia32_pause();
    pthread_mutex_lock(&g_pool.mu);
    while (__atomic_load_n(&g_pool.pending, __ATOMIC_ACQUIRE) > 0) pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
    __atomic_store_n(&g_pool.job, NULL, __ATOMIC_RELEASE);
    pthread_mutex_unlock(&g_pool.mu);
    pthread_mutex_unlock(&g_job_mu);
}

static void batch_row_thunk(const void *a, int64_t k) { batch_row((const batch_args *)a, k); }

GVLO_EXPORT void gvlo_reconstruct_batch(
    uint8_t *out, const int64_t *out_offsets, const int32_t *regions,
    int64_t regions_stride, int64_t batch, int64_t ploidy, const int32_t *shifts,
    const int64_t *geno_offset_idx, const int64_t *go_starts,
    const int64_t *go_stops, const int32_t *geno_v_idxs, const int32_t *v_starts,
    const int32_t *ilens, const uint8_t *alt_alleles, const int64_t *alt_offsets,
    const uint8_t *ref, const int64_t *ref_offsets, uint8_t pad_char,
    const uint8_t *keep, const int64_t *keep_offsets, int32_t *annot_v_idxs,
    int32_t *annot_ref_pos, const uint8_t *to_rc, uint8_t *onehot, int n_threads)
{
    batch_args a = {out, out_offsets, regions, regions_stride, batch, ploidy,
                    shifts, geno_offset_idx, go_starts, go_stops, geno_v_idxs,
                    v_starts, ilens, alt_alleles, alt_offsets, ref, ref_offsets,
                    pad_char, keep, keep_offsets, annot_v_idxs, annot_ref_pos,
                    to_rc, onehot};
    run_rows(&a, batch_row_thunk, batch * ploidy, n_threads);
    /* annotations are reversed (not complemented) for RC rows:
     * ffi/mod.rs:2237-2397 via reverse.rs:25-38 */
    if (to_rc && annot_v_idxs)
        gvlo_reverse_rows_4((uint32_t *)annot_v_idxs, out_offsets, to_rc, batch * ploidy);
    if (to_rc && annot_ref_pos)
        gvlo_reverse_rows_4((uint32_t *)annot_ref_pos, out_offsets, to_rc, batch * ploidy);
}

/* ------------------------------------------------------------------------- */
/* a4: per-row length delta, 3 modes.  src/genotypes/mod.rs:15-125.          */
/* ------------------------------------------------------------------------- */
typedef struct {
    const int64_t *geno_offset_idx;
    int64_t ploidy;
    const int32_t *geno_v_idxs;
    const int64_t *go_starts, *go_stops;
    const int32_t *ilens;
    const uint8_t *keep;
    const int64_t *keep_offsets;
    const int32_t *q_starts, *q_ends;
    int64_t q_stride;
    const int32_t *v_starts;
    int32_t *diffs;
} diffs_args;

static void diffs_row(const void *p, int64_t k)
{
    const diffs_args *a = (const diffs_args *)p;
    int64_t query = k / a->ploidy;
    int64_t o_idx = a->geno_offset_idx[k];
    int64_t o_s = a->go_starts[o_idx], o_e = a->go_stops[o_idx];
    int has_query = a->q_starts && a->q_ends && a->v_starts; /* mod.rs:35 */
    int has_keep = a->keep && a->keep_offsets;               /* mod.rs:36 */
    int64_t acc = 0;
    if (o_e - o_s == 0) {
        acc = 0; /* mod.rs:46-47 */
    } else if (has_query) { /* mod.rs:48-85 */
        int64_t q_start = a->q_starts[query * a->q_stride];
        int64_t q_end = a->q_ends[query * a->q_stride];
        int64_t ref_idx = q_start;
        for (int64_t v = o_s; v < o_e; v++) {
            if (has_keep && !a->keep[a->keep_offsets[k] + (v - o_s)]) continue;
            int64_t vi = a->geno_v_idxs[v];
            int64_t vs = a->v_starts[vi];
            int64_t il = a->ilens[vi];
            int64_t v_end = vs - min64(il, 0) + 1;
            if (v_end <= q_start) continue;
            if (vs >= q_end) break;
            if (vs >= q_start && vs < ref_idx) continue;
            ref_idx = max64(ref_idx, v_end);
            if (il < 0) il += max64(q_start - vs - 1, 0);
            il += max64(v_end - q_end, 0);
            acc += il;
        }
    } else if (has_keep) { /* mod.rs:86-96 */
        int64_t ks = a->keep_offsets[k];
        for (int64_t v = o_s; v < o_e; v++)
            if (a->keep[ks + (v - o_s)]) acc += a->ilens[a->geno_v_idxs[v]];
    } else { /* mod.rs:97-103 */
        for (int64_t v = o_s; v < o_e; v++) acc += a->ilens[a->geno_v_idxs[v]];
    }
    a->diffs[k] = (int32_t)acc; /* `as i32` truncation */
}

GVLO_EXPORT void gvlo_get_diffs_sparse(
    const int64_t *geno_offset_idx, int64_t batch, int64_t ploidy,
    const int32_t *geno_v_idxs, const int64_t *go_starts, const int64_t *go_stops,
    const int32_t *ilens, const uint8_t *keep, const int64_t *keep_offsets,
    const int32_t *q_starts, const int32_t *q_ends, int64_t q_stride,
    const int32_t *v_starts, int32_t *diffs, int n_threads)
{
    diffs_args a = {geno_offset_idx, ploidy, geno_v_idxs, go_starts, go_stops, ilens,
                    keep, keep_offsets, q_starts, q_ends, q_stride, v_starts, diffs};
    run_rows(&a, diffs_row, batch * ploidy, n_threads);
}

/* src/genotypes/mod.rs:132-176.  keep_offsets has batch*ploidy+1 entries and
 * is written first; `keep` must have keep_offsets[-1] entries (call twice, or
 * size it with gvlo_exonic_total). */
GVLO_EXPORT int64_t gvlo_choose_exonic_variants(
    const int32_t *starts, const int32_t *ends, const int64_t *geno_offset_idx,
    int64_t batch, int64_t ploidy, const int32_t *geno_v_idxs,
    const int64_t *go_starts, const int64_t *go_stops, const int32_t *v_starts,
    const int32_t *ilens, uint8_t *keep /* nullable: size query */,
    int64_t *keep_offsets)
{
    int64_t acc = 0;
    keep_offsets[0] = 0;
    for (int64_t k = 0; k < batch * ploidy; k++) {
        int64_t o_idx = geno_offset_idx[k];
        acc += max64(go_stops[o_idx] - go_starts[o_idx], 0);
        keep_offsets[k + 1] = acc;
    }
    if (!keep) return acc;
    for (int64_t q = 0; q < batch; q++) {
        int64_t ref_start = starts[q], ref_end = ends[q];
        for (int64_t h = 0; h < ploidy; h++) {
            int64_t k = q * ploidy + h;
            int64_t o_idx = geno_offset_idx[k];
            int64_t o_s = go_starts[o_idx], o_e = go_stops[o_idx];
            int64_t ks = keep_offsets[k];
            for (int64_t v = o_s; v < o_e; v++) {
                int64_t vi = geno_v_idxs[v];
                int64_t pos = v_starts[vi];
                int64_t end = pos - min64(ilens[vi], 0) + 1;
                keep[ks + (v - o_s)] = (pos >= ref_start && end <= ref_end);
            }
        }
    }
    return acc;
}

/* ------------------------------------------------------------------------- */
/* a5: fused-entry output sizing.  src/ffi/mod.rs:794-811.                   */
/* ------------------------------------------------------------------------- */
GVLO_EXPORT void gvlo_fused_out_offsets(const int32_t *regions, int64_t regions_stride,
                                        int64_t batch, int64_t ploidy,
                                        const int32_t *diffs, int64_t output_length,
                                        int64_t *out_offsets)
{
    int64_t acc = 0;
    out_offsets[0] = 0;
    for (int64_t k = 0; k < batch * ploidy; k++) {
        int64_t q = k / ploidy;
        int64_t len;
        if (output_length >= 0) {
            len = output_length;
        } else {
            const int32_t *reg = regions + q * regions_stride;
            int64_t ref_len = (int64_t)(int32_t)(reg[2] - reg[1]); /* i32 subtract, then widen */
            len = max64(ref_len + diffs[k], 0);
        }
        acc += len;
        out_offsets[k + 1] = acc;
    }
}

/* ------------------------------------------------------------------------- */
/* a9: padded reference fetch.  src/reference/mod.rs:9-50 (padded_slice) and
 * :56-120 (get_reference).  `out` must be zero-initialised by the caller the
 * way Array1::zeros does (mod.rs:66): start >= stop leaves the row untouched. */
/* ------------------------------------------------------------------------- */
static void padded_slice(const uint8_t *arr, int64_t len, int64_t start, int64_t stop,
                         uint8_t pad, uint8_t *out, int64_t out_len)
{
    if (start >= stop) return;
    if (stop < 0) {
        memset(out, pad, (size_t)out_len);
        return;
    }
    int64_t pad_left = max64(-start, 0);
    int64_t pad_right = max64(stop - len, 0);
    /* out[j] <- arr[start + j] where defined, pad elsewhere; identical to the
     * four-branch form at mod.rs:26-49 whenever out_len == stop - start. */
    for (int64_t j = 0; j < out_len; j++) {
        int64_t r = start + j;
        out[j] = (j < pad_left || j >= out_len - pad_right || r < 0 || r >= len) ? pad : arr[r];
    }
}

typedef struct {
    const int32_t *regions;
    int64_t regions_stride;
    const int64_t *out_offsets;
    const uint8_t *reference;
    const int64_t *ref_offsets;
    uint8_t pad_char;
    uint8_t *out;
} ref_args;

static void ref_row(const void *p, int64_t i)
{
    const ref_args *a = (const ref_args *)p;
    const int32_t *reg = a->regions + i * a->regions_stride;
    int64_t c_s = a->ref_offsets[reg[0]], c_e = a->ref_offsets[reg[0] + 1];
    padded_slice(a->reference + c_s, c_e - c_s, reg[1], reg[2], a->pad_char,
                 a->out + a->out_offsets[i], a->out_offsets[i + 1] - a->out_offsets[i]);
}

GVLO_EXPORT void gvlo_get_reference(const int32_t *regions, int64_t regions_stride,
                                    int64_t n, const int64_t *out_offsets,
                                    const uint8_t *reference, const int64_t *ref_offsets,
                                    uint8_t pad_char, const uint8_t *to_rc,
                                    uint8_t *out, int n_threads)
{
    ref_args a = {regions, regions_stride, out_offsets, reference, ref_offsets, pad_char, out};
    run_rows(&a, ref_row, n, n_threads);
    if (to_rc) gvlo_rc_rows(out, out_offsets, to_rc, n);
}

/* ------------------------------------------------------------------------- */
/* a10: one-hot (definition above; parity unpinned).  layout 0 = (n, 4)
 * base-major ("…, L, 4" as seqpro appends the alphabet axis last); layout 1 =
 * channel-major rows: out[row, a, j] for `n_rows` rows of length `row_len`.   */
/* ------------------------------------------------------------------------- */
GVLO_EXPORT void gvlo_onehot(const uint8_t *in, int64_t n_rows, int64_t row_len,
                             int layout, uint8_t *out)
{
    if (layout == 0) {
        onehot_row(in, n_rows * row_len, out);
        return;
    }
    for (int64_t r = 0; r < n_rows; r++)
        for (int64_t j = 0; j < row_len; j++) {
            uint8_t b = in[r * row_len + j];
            uint8_t *o = out + r * 4 * row_len;
            o[0 * row_len + j] = (b == 'A');
            o[1 * row_len + j] = (b == 'C');
            o[2 * row_len + j] = (b == 'G');
            o[3 * row_len + j] = (b == 'T');
        }
}

GVLO_EXPORT int gvlo_version(void) { return 1; }
