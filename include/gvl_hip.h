/*
 * gvl_hip.h -- C-ABI of the MI355X (gfx950) haplotype hot path.
 *
 * This is the drop-in boundary: the entry points below are what GenVarLoader's
 * FFI for haplotype reconstruction would bind (the reference binds a PyO3
 * module, `src/ffi/mod.rs`; INTEGRATION.md shows the ctypes stub a maintainer
 * would add).  Plain pointers and sizes only -- no torch / numpy types.
 *
 * Conventions
 *  - Every data pointer is a DEVICE pointer (HBM) unless it says "host".
 *  - Array dtypes/layouts are exactly the reference's FFI layouts
 *    (python/genvarloader/_dataset/_haps.py:844-866): regions i32 (B, stride>=3)
 *    [contig, start, end, ...], shifts i32 (B, P), geno_offset_idx i64 (B, P),
 *    geno offsets as two i64 rows (starts, stops), geno_v_idxs i32, variant table
 *    v_starts/ilens i32, alt_alleles u8 + alt_offsets i64 (n_variants + 1),
 *    ref u8 + ref_offsets i64 (n_contigs + 1), keep u8(bool) + keep_offsets i64
 *    (B*P + 1), to_rc u8(bool) (B*P).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *    calls are asynchronous on that stream and never synchronise the host.
 *  - Return value 0 = ok; otherwise an error code, with a message available from
 *    gvl_last_error() (thread local).  HIP launch failures are reported, not
 *    swallowed; there is no CPU fallback anywhere in this library.
 */
#ifndef GVL_HIP_H
#define GVL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVL_ABI_VERSION 11

enum {
    GVL_OK = 0,
    GVL_ERR_INVALID = 1, /* bad argument (NULL pointer, negative size, ...) */
    GVL_ERR_HIP = 2,     /* a HIP runtime call / kernel launch failed */
    GVL_ERR_UNSUPPORTED = 3
};

/* one-hot layouts (a10; the reference delegates one-hot to seqpro.DNA.ohe,
 * docs/source/index.md:109-119 -- alphabet "ACGT", non-alphabet -> all zero) */
enum {
    GVL_ONEHOT_LC = 0, /* (total_bases, 4): base-major, what sp.DNA.ohe returns */
    GVL_ONEHOT_CL = 1  /* (rows, 4, L): channel-major; fixed-length rows only   */
};

/* 16-byte packed variant record, built once per dataset by gvl_pack_variants()
 * from v_starts / ilens / alt_offsets / alt_alleles so that the kernel needs one
 * 16-B gather per variant instead of four dependent ones. */
typedef struct gvl_vrec {
    int32_t pos;     /* v_starts[v]                                        */
    int32_t ilen;    /* ilens[v]                                           */
    int32_t alen;    /* alt_offsets[v+1] - alt_offsets[v]                  */
    uint32_t inl;    /* first 4 ALT bytes, little endian (zero padded)     */
} gvl_vrec;

/* 16-byte record per genotype CSR entry, built once per dataset by
 * gvl_pack_genotypes(): the variant's fields next to the CSR entry itself, so that a row's
 * variants are ONE contiguous 16-B-per-lane read (geno_v_idxs[i] -> vrec[v] is two dependent
 * gathers, ~1 us of a 13 us launch).  Optional: 16 B x n_geno of HBM. */
typedef struct gvl_grec {
    int32_t pos;       /* v_starts[v]                                              */
    int32_t ilen;      /* ilens[v]                                                 */
    uint32_t alen_inl; /* (min(alen, 2^24 - 1) << 8) | first ALT byte              */
    int32_t v_idx;     /* geno_v_idxs[i], clamped to [0, n_variants)               */
} gvl_grec;

/* Slot-major records ("one cache line per haplotype"): GVL_SLOT_RECS 16-byte records per genotype
 * slot, slot o at slot_rec[o * GVL_SLOT_RECS ..], built once per dataset by gvl_pack_slots().  A row
 * then reaches its variants with ONE dependent read after geno_offset_idx (the CSR needs
 * geno_offset_idx -> geno_o_starts/stops -> records -> alt_offsets: three more levels, each an HBM +
 * TLB miss at genome scale), and insertions find their ALT bytes without going through alt_offsets.
 * Unused entries have alen_inl == GVL_SREC_EMPTY; a slot with more than GVL_SLOT_RECS variants has
 * GVL_SREC_OVERFLOW in entry 0 and is read through the CSR instead.  Optional: 128 B x n_geno_offsets
 * of HBM (sized for 288 GB parts); not used by the annotated entry (it needs v_idx). */
#define GVL_SLOT_RECS 8
#define GVL_SREC_EMPTY 0xFFFFFFFFu
#define GVL_SREC_OVERFLOW 0xFFFFFFFEu
typedef struct gvl_srec {
    int32_t pos;       /* v_starts[v]                                              */
    int32_t ilen;      /* ilens[v]                                                 */
    uint32_t alen_inl; /* (alen << 8) | first ALT byte; alen < 2^24 - 1 (a longer allele marks its slot
                          GVL_SREC_OVERFLOW: the row then reads the CSR)           */
    uint32_t a0;       /* alt_offsets[v] (the table is only built when alt_len < 2^32) */
} gvl_srec;

/* Nibble-packed reference ("ref4"): base i of `ref` is nibble i & 1 of byte i >> 1 (low nibble first),
 *   bits 0-1  (byte >> 1) & 3  -- A 0, C 1, T 2, G 3
 *   bit  2    the byte is exactly one of "ACGT"
 *   bit  3    the byte is 'N'
 * so that the one-hot of 4 bases is two LDS lookups on 16 bits read from HALF the bytes (and half the
 * cache lines / DRAM pages) of the byte reference: the window reads are the cold, scattered part of
 * the path's traffic.  Any other byte (IUPAC codes, lower case) packs to 0 = "no channel", which is
 * what the one-hot definition gives it.  Built once per dataset by gvl_pack_reference(); optional:
 * ref_len / 2 + GVL_REF4_PAD bytes of HBM (the padding lets a window that starts anywhere inside the
 * reference read 2048 bases ahead without a bounds test per lane).  Only the one-hot-only fixed-length entry reads it (haplotype BYTES
 * always come from `ref`). */
#define GVL_REF4_PAD 1088

/* Per-dataset, device-resident arrays.  Mirrors `_HapsFfiStatic`
 * (_haps.py:233-247) + `Reference` (_reference.py:31-50) + the sparse genotype
 * CSR (`genotypes/offsets.npy`, `variant_idxs.npy`). */
typedef struct gvl_static {
    const uint8_t *ref;          /* packed contigs                          */
    int64_t ref_len;             /* bytes in `ref`                          */
    const int64_t *ref_offsets;  /* n_contigs + 1                           */
    int64_t n_contigs;
    const int32_t *v_starts;     /* n_variants                              */
    const int32_t *ilens;        /* n_variants                              */
    const int64_t *alt_offsets;  /* n_variants + 1                          */
    const uint8_t *alt_alleles;  /* alt_len bytes                           */
    int64_t n_variants;
    int64_t alt_len;
    const gvl_vrec *vrec;        /* n_variants, from gvl_pack_variants()    */
    const int64_t *geno_o_starts; /* n_geno_offsets: row 0 of the (2, n) array */
    const int64_t *geno_o_stops;  /* n_geno_offsets: row 1                     */
    int64_t n_geno_offsets;
    const int32_t *geno_v_idxs;  /* n_geno                                  */
    int64_t n_geno;
    uint8_t pad_char;
    const gvl_grec *geno_rec;    /* nullable: n_geno, from gvl_pack_genotypes() */
    const gvl_srec *slot_rec;    /* nullable: GVL_SLOT_RECS * n_geno_offsets, from gvl_pack_slots() */
    const uint8_t *ref4;         /* nullable: (ref_len + 1) / 2 + GVL_REF4_PAD bytes, from gvl_pack_reference() */
    const int32_t *slot_vidx;    /* nullable: GVL_SLOT_RECS * n_geno_offsets i32, from gvl_pack_slot_vidx(): the variant index of
                                    every slot record (-1: no record) -- what annotated haplotypes need next to the slot line */
} gvl_static;

/* Per-batch arrays.  Mirrors `ReconstructionRequest` (_haps.py:58-93) as
 * marshalled at the fused call site (_haps.py:844-866). */
typedef struct gvl_batch {
    const int32_t *regions;        /* (batch, regions_stride) */
    int64_t regions_stride;        /* >= 3 (runtime arrays are (B, 4)) */
    const int32_t *shifts;         /* (batch, ploidy) */
    const int64_t *geno_offset_idx; /* (batch, ploidy) */
    int64_t batch;
    int64_t ploidy;
    const uint8_t *keep;           /* nullable */
    const int64_t *keep_offsets;   /* nullable; batch*ploidy + 1 */
    const uint8_t *to_rc;          /* nullable; batch*ploidy */
    int64_t output_length;         /* >= 0 fixed length; -1 ragged */
    const int64_t *out_offsets;    /* batch*ploidy + 1.  Required when
                                      output_length < 0 (from gvl_hap_offsets or the
                                      caller's splice plan, ffi/mod.rs:1983-2002);
                                      NULL in fixed mode means row k starts at
                                      k * output_length. */
    int64_t max_row_len;           /* host hint: upper bound of any row's length
                                      (fixed mode: ignored, output_length is used) */
    const void *hap_plan;          /* nullable: the chunk plans of THESE rows (gvl_hap_plan over exactly these
                                      request arrays; fixed-length rows longer than 2048 bases).  NULL: every
                                      chunk-wave of the launch walks its row itself */
    const int64_t *out_bounds;     /* nullable (ABI 11): i64 (batch*ploidy, 2), row k is written at [start, end) =
                                      out_bounds[k] of the outputs -- rows anywhere, in any order, pairwise disjoint
                                      (the scatter write of reconstruct_haplotypes_from_svar2, src/reconstruct/mod.rs:
                                      619-826, whose spliced callers interleave one contig group's rows with another's).
                                      Replaces out_offsets (which must then be NULL); max_row_len bounds the rows as for
                                      out_offsets.  Disjointness is the caller's contract, as it is the Rust core's
                                      (the PyO3 layer checks it on the host: src/ffi/mod.rs:101-139). */
    int64_t total_len_hint;        /* rows at out_offsets (ABI 11): the SUM of the rows' lengths when the caller knows it (it sized the
                                      output: out_offsets[-1]), else 0.  Tells a batch of mostly short rows with a few long ones -- a
                                      spliced batch's exons: the pipelined kernel takes it, rows longer than its 2560 bases by the
                                      launch's front workgroups, chunks in parallel -- from a batch of long rows (the chunked kernel).  Without it every batch whose
                                      max_row_len exceeds 2560 takes the chunked kernel.  Results never depend on it. */
    const int64_t *query_seed;     /* tracks only (ABI 11), nullable: i64 (batch): the GLOBAL batch row of local query q -- the `query`
                                      component of the FlankSample fill's per-position seed (src/tracks/mod.rs:744-760: callers that
                                      split one logical batch across several calls, as the SVAR2 read-bound path's per-contig-group
                                      loop does, pass the group's global rows so that seed-dependent fills match the single fused
                                      call).  NULL: the local index (exact for one call per batch). */
} gvl_batch;

/* Outputs; any of the data pointers may be NULL (that output is skipped), but
 * at least one of haps / onehot must be given. */
typedef struct gvl_out {
    uint8_t *haps;          /* total bytes (= out_offsets[-1]) */
    uint8_t *onehot;        /* 4 * total bytes */
    int32_t onehot_layout;  /* GVL_ONEHOT_LC / GVL_ONEHOT_CL */
    int32_t *annot_v_idxs;  /* total i32, nullable (a11) */
    int32_t *annot_ref_pos; /* total i32, nullable (a11) */
    int64_t *out_offsets;   /* nullable: fixed mode writes k*output_length here
                               (batch*ploidy + 1), the second return value of
                               reconstruct_haplotypes_fused (ffi/mod.rs:743) */
} gvl_out;

int gvl_abi_version(void);
/* Test / diagnostic switches (the bits of the GVL_DBG environment variable, see
 * debug_flags() in gvl_hip.hip); flags < 0 returns to the environment's value. */
int gvl_set_debug_flags(int flags);
/* Launch-policy overrides, process-wide: for A/B measurements and for tests that must force a schedule the built-in policy
 * would not pick at their sizes.  value <= 0 returns the key to the built-in policy.  Results never depend on them.
 *   GVL_TUNE_PIPE_ROWS_X100      rows per wave of the pipelined kernel, times 100 (100 = one, 150 = two for the first half of
 *                                the waves, 200 = exactly two, 300 = three ...; at most 3200).  Built in: 150 from 49 152 rows.
 *   GVL_TUNE_PIPE_MIN_ROWS       launches with fewer rows keep the wave-per-row kernel (built in: 8192; 2048 for groups).
 *   GVL_TUNE_LEAN_SUB            consecutive chunks of a long row one wave takes (built in: 2).
 *   GVL_TUNE_TRACK_PLAN_MAX_MB   row plans of an epoch larger than this are not kept (built in: 512).
 *   GVL_TUNE_RAGGED_SIZING       1: the native loader sizes ragged rows once per group of batches (round 4's way), not once per epoch.
 *   GVL_TUNE_HAP_PLAN_MAX_MB     haplotype chunk plans of an epoch larger than this are not made (built in: 64 -- they pay while they
 *                                stay in the 256 MB Infinity Cache next to the epoch's other inputs, and cost more than the walks
 *                                they save beyond it: profiles/r05_cfg4_plans_vs_size.txt).
 *   GVL_TUNE_MIXED_MIN_ROWS      ragged batches of mostly short rows with a few beyond 2560 bases (gvl_batch.total_len_hint): with fewer
 *                                rows than this the all-purpose kernel keeps them (built in: see lean_rag_eligible, gvl_hip.hip). */
enum { GVL_TUNE_PIPE_ROWS_X100 = 0, GVL_TUNE_PIPE_MIN_ROWS = 1, GVL_TUNE_LEAN_SUB = 2, GVL_TUNE_TRACK_PLAN_MAX_MB = 3, GVL_TUNE_RAGGED_SIZING = 4,
       GVL_TUNE_HAP_PLAN_MAX_MB = 5, GVL_TUNE_MIXED_MIN_ROWS = 6, GVL_TUNE_COUNT = 7 };
int gvl_set_tuning(int32_t key, int64_t value);
const char *gvl_last_error(void);
/* (The flag behind gvl_async_error is PROCESS-global: it says that some launch of this process met the
 * condition, not which one.)
 * Errors only the device can find are reported asynchronously, like a sticky HIP error: after the
 * stream has been synchronised, gvl_async_error() returns GVL_ERR_INVALID (message in
 * gvl_last_error()) if any launch since the last clear met a row longer than its batch's
 * `max_row_len` hint (such a row is left partly unwritten; the reference writes every byte,
 * src/ffi/mod.rs:17-35, so this must not pass silently).  clear != 0 resets the flag. */
int gvl_async_error(int clear);

/* For callers without a device allocator of their own (a C / Rust host; torch callers keep their
 * tensors and fill a gvl_static themselves): copy a dataset's arrays from HOST pointers into freshly
 * hipMalloc'd memory, build vrec and -- with_layouts != 0 -- the optional geno_rec / slot_rec layouts,
 * and return a gvl_static that points at the device copies ("gvl_static_upload/free").
 * `host` uses the same struct with host pointers (vrec / geno_rec / slot_rec ignored).  The copies are
 * asynchronous on `stream` from pageable or pinned memory.  gvl_static_free releases everything. */
int gvl_static_upload(const gvl_static *host, int32_t with_layouts, gvl_static **out, void *stream);
int gvl_static_free(gvl_static *st);

/* Build the packed variant records (once per dataset).
 * Replaces nothing in the reference; it is the HBM layout this path reads
 * instead of the four gathers at src/reconstruct/mod.rs:296-305. */
int gvl_pack_variants(const int32_t *v_starts, const int32_t *ilens,
                      const int64_t *alt_offsets, const uint8_t *alt_alleles,
                      int64_t n_variants, gvl_vrec *vrec_out, void *stream);

/* Build the per-genotype-entry records (once per dataset; optional, see gvl_grec).
 * Needs st->vrec, st->geno_v_idxs, st->n_geno, st->n_variants. */
int gvl_pack_genotypes(const gvl_static *st, gvl_grec *grec_out, void *stream);

/* Build the nibble-packed reference (once per dataset; optional, see "ref4" above).
 * ref4_out: gvl_ref4_bytes(ref_len) = (ref_len + 1) / 2 + GVL_REF4_PAD bytes. */
int64_t gvl_ref4_bytes(int64_t ref_len);
int gvl_pack_reference(const uint8_t *ref, int64_t ref_len, uint8_t *ref4_out, void *stream);

/* Build the slot records' variant indices (once per dataset; optional: the lean path of annotated haplotypes reads them with the
 * slot line).  out: GVL_SLOT_RECS * n_geno_offsets i32. */
int gvl_pack_slot_vidx(const gvl_static *st, int32_t *out, void *stream);

/* Build the slot-major records (once per dataset; optional, see gvl_srec).
 * Needs st->vrec, alt_offsets, geno_o_starts/stops, geno_v_idxs, n_geno_offsets, n_variants;
 * GVL_ERR_UNSUPPORTED when alt_len >= 2^32. */
int gvl_pack_slots(const gvl_static *st, gvl_srec *srec_out, void *stream);

/* Haplotype reconstruction (+RC, +one-hot, +annotations) for a batch.
 * Replaces: reconstruct_haplotypes_fused (src/ffi/mod.rs:722-860) steps 3-4b,
 *   reconstruct_haplotypes_from_sparse (src/ffi/mod.rs:632-700; in place when
 *   batch->out_offsets is given), reconstruct_haplotypes_spliced_fused
 *   (src/ffi/mod.rs:1981-2076; caller-supplied out_offsets, ploidy 1),
 *   reconstruct_annotated_haplotypes_fused (src/ffi/mod.rs:2237-2397),
 *   rc_flat_rows_inplace on the result (src/reverse.rs:56-69), and the user-side
 *   seqpro one-hot (docs/source/index.md:109-119). */
int gvl_reconstruct(const gvl_static *st, const gvl_batch *bt, const gvl_out *out,
                    void *stream);

/* The same for `n` <= GVL_MANY_MAX batches (arrays of n structs) in ONE host call: every batch is validated before
 * the first launch -- what the native loader issues per group of batches.  Equivalent to n calls of gvl_reconstruct;
 * outputs of different batches must not overlap.
 * Batches of the same shape that the lean path takes (fixed-length rows of at most 2048 bases, or rows at out_offsets
 * of at most 2560 bases; row-major one-hot and / or bytes, no keep mask, no annotations; slot_rec + ref4 present; every
 * batch but the last with the first one's row count) are ONE grid: row k of the launch belongs to batch
 * k / rows_per_batch, a wave takes rows w, w + W, ... and keeps its next row's reads in flight under the stores of the row in
 * hand (recon_lean_rows_kernel) -- the form that makes groups pay: 6.3-7.2 us per 4096 x 2048 batch
 * cold against 7.8-8.3 us for launches of single batches.  Anything else: back-to-back launches on `stream`. */
#define GVL_MANY_MAX 16
int gvl_reconstruct_many(const gvl_static *st, const gvl_batch *bts, const gvl_out *outs,
                         int32_t n, void *stream);

/* Chunk plans of long rows (BASELINE config 4: 131 072 bases = 64 chunks of 2048).  The lean kernel gives every pair of chunks
 * of a row its own wave; with a plan such a wave finds where its window starts and which applied variants touch its chunk in two
 * reads, without one it replays the row's walk (src/reconstruct/mod.rs:85-198) up to its chunk.  gvl_hap_plan walks every row of
 * `bt` ONCE (bt->output_length = the fixed row length; regions / shifts / geno_offset_idx of batch * ploidy rows; no keep mask)
 * into `plan`, gvl_hap_plan_bytes(batch * ploidy, output_length) bytes of device memory; row k's plans are the k-th
 * gvl_hap_plan_bytes(1, output_length) bytes, so a plan made over a whole epoch's request table is handed to the epoch's batches
 * as pointers into it (gvl_batch.hap_plan; the native loader does exactly that).  Results never depend on it: a chunk the plan
 * cannot express is flagged and walks its row as before.  gvl_hap_plan_bytes is 0 for rows that are not planned (one chunk, or
 * more than 512). */
int64_t gvl_hap_plan_bytes(int64_t n_rows, int64_t output_length);
int gvl_hap_plan(const gvl_static *st, const gvl_batch *bt, void *plan, void *stream);

/* Per-row length deltas.  Replaces get_diffs_sparse (src/ffi/mod.rs:143-185 ->
 * src/genotypes/mod.rs:15-125).  Query mode iff q_starts, q_ends and st->v_starts
 * are all non-NULL (q_* are read at q_stride elements per query, so
 * regions+1 / regions+2 with stride 4 works); keep mode iff bt->keep and
 * bt->keep_offsets are non-NULL.  Only bt->geno_offset_idx/batch/ploidy/keep* are
 * read from `bt`.  diffs: i32 (batch, ploidy). */
int gvl_get_diffs_sparse(const gvl_static *st, const gvl_batch *bt,
                         const int32_t *q_starts, const int32_t *q_ends,
                         int64_t q_stride, int32_t *diffs, void *stream);

/* Output sizing of the fused entry: diffs in query mode on regions[:,1:3], then
 * len_k = output_length >= 0 ? output_length : max(0, end - start + diff_k) and
 * its exclusive prefix sum.  Replaces src/ffi/mod.rs:769-811.
 * diffs (nullable) i32 (batch, ploidy); out_offsets i64 (batch*ploidy + 1);
 * total_and_max (nullable) i64[2] <- {out_offsets[-1], max row length}. */
int gvl_hap_offsets(const gvl_static *st, const gvl_batch *bt, int32_t *diffs,
                    int64_t *out_offsets, int64_t *total_and_max, void *stream);

/* Padded reference fetch (+RC, +one-hot).  Replaces get_reference
 * (src/ffi/mod.rs:2401-2429 -> src/reference/mod.rs:56-120).  Only st->ref,
 * ref_len, ref_offsets, n_contigs, pad_char are read.  Row i has length
 * out_offsets[i+1]-out_offsets[i] (the caller passes end - start). */
int gvl_get_reference(const gvl_static *st, const int32_t *regions,
                      int64_t regions_stride, int64_t n_rows,
                      const int64_t *out_offsets, int64_t max_row_len,
                      const uint8_t *to_rc, uint8_t *out, uint8_t *onehot,
                      void *stream);

/* The same for a GROUP of up to GVL_MANY_MAX batches of regions in ONE launch (what gvl_reconstruct_many is to gvl_reconstruct: a
 * launch per 4096-row batch is bound by its own start-up -- 11.9 us per batch on one stream against 7 in a group of 16).  Batches of
 * the same shape (rows, outputs asked for, rows of at most 2560 bases) share a grid; anything else runs batch by batch. */
typedef struct gvl_ref_batch {
    const int32_t *regions; int64_t regions_stride; int64_t n_rows;
    const int64_t *out_offsets; int64_t max_row_len;
    const uint8_t *to_rc;          /* nullable */
    uint8_t *out; uint8_t *onehot; /* either may be NULL, not both */
} gvl_ref_batch;
int gvl_get_reference_many(const gvl_static *st, const gvl_ref_batch *batches, int32_t n, void *stream);

/* Keep mask of the spliced path.  Replaces choose_exonic_variants (src/genotypes/mod.rs:127-176):
 * keep[keep_offsets[k] + j] = variant j of row k lies entirely inside its query's exon
 * [starts[q], ends[q]) (pos >= start && pos - min(ilen, 0) + 1 <= end), q = k / ploidy.
 * gvl_keep_offsets first: keep_offsets i64 (batch*ploidy + 1) = running sum of the rows' variant
 * counts, total_and_max (nullable) = {keep length, largest row}; the caller sizes `keep` from
 * the total, then gvl_choose_exonic_variants fills it. */
int gvl_keep_offsets(const gvl_static *st, const int64_t *geno_offset_idx, int64_t batch, int64_t ploidy,
                     int64_t *keep_offsets, int64_t *total_and_max, void *stream);
int gvl_choose_exonic_variants(const gvl_static *st, const int32_t *starts, const int32_t *ends,
                               const int64_t *geno_offset_idx, int64_t batch, int64_t ploidy,
                               const int64_t *keep_offsets, uint8_t *keep, void *stream);

/* In-place reverse-complement of masked rows.  Replaces rc_flat_rows_inplace
 * (src/reverse.rs:56-69; COMP semantics :9-21,:45-53). */
int gvl_rc_rows(uint8_t *data, const int64_t *offsets, const uint8_t *to_rc,
                int64_t n_rows, void *stream);

/* The same for rows given as (start, end) pairs, bounds i64 (n_rows, 2) -- rows need not be
 * adjacent.  Replaces rc_bounded_rows_inplace (src/reverse.rs:75-84). */
int gvl_rc_bounded_rows(uint8_t *data, const int64_t *bounds, const uint8_t *to_rc,
                        int64_t n_rows, void *stream);

/* In-place reversal (no complement) of masked rows of 4-byte elements (f32
 * tracks / i32 annotations).  Replaces reverse_flat_rows_inplace<T>
 * (src/reverse.rs:25-38). */
int gvl_reverse_rows_4(void *data, const int64_t *offsets, const uint8_t *to_rc,
                       int64_t n_rows, void *stream);

/* Stand-alone one-hot of n bytes -> (n, 4) uint8 (GVL_ONEHOT_LC).  Replaces the
 * user-side sp.DNA.ohe(haps) (docs/source/index.md:114). */
int gvl_onehot(const uint8_t *in, int64_t n, uint8_t *out, void *stream);


/* ---- tracks (BASELINE config 4) ------------------------------------ */

/* insertion-fill strategies (python/genvarloader/_dataset/_insertion_fill.py:9-13) */
enum {
    GVL_FILL_REPEAT_5P = 0,
    GVL_FILL_REPEAT_5P_NORM = 1,
    GVL_FILL_CONSTANT = 2,
    GVL_FILL_FLANK_SAMPLE = 3,
    GVL_FILL_INTERPOLATE = 4
};

/* Running maximum of the interval ends inside each list: pmax[c] = max(itv_ends[list start..c]).
 * Built once per interval set (n_lists = number of lists in itv_offsets); lets the painter
 * answer "no interval reaches this position" without walking back over the list. */
int gvl_intervals_prefix_max(const int32_t *itv_ends, const int64_t *itv_offsets, int64_t n_lists,
                             int32_t *pmax_out, void *stream);

/* Paint sorted (start, end, value) intervals into zeroed per-query f32 rows.
 * Replaces intervals_to_tracks (src/ffi/mod.rs:188-240 -> src/intervals.rs:19-126).
 * offset_idxs i64 (n_queries): index into itv_offsets; starts i32 read at `starts_stride`
 * elements per query (regions + 1 with stride 4 works); out f32 (out_offsets[-1]);
 * out_offsets i64 (n_queries + 1).  max_row_len: host bound of any row's length.
 * n_intervals: length of the itv_* arrays.  itv_pmax_ends: from gvl_intervals_prefix_max, or
 * NULL (then it is built for the queried lists in stream-ordered scratch memory). */
int gvl_intervals_to_tracks(const int64_t *offset_idxs, const int32_t *starts, int64_t starts_stride,
                            int64_t n_queries, const int32_t *itv_starts, const int32_t *itv_ends,
                            const float *itv_values, const int64_t *itv_offsets, int64_t n_intervals,
                            const int32_t *itv_pmax_ends, float *out, const int64_t *out_offsets,
                            int64_t max_row_len, void *stream);

/* Shift and realign per-query reference-coordinate tracks to each haplotype.
 * Replaces shift_and_realign_tracks_sparse (src/tracks/mod.rs:495-667; core :224-406,
 * fills :87-190, PRNG :31-54) and the reversal step of intervals_and_realign_track_fused
 * (src/ffi/mod.rs:2657-2668).  From `st` only the genotype CSR + v_starts/ilens are read;
 * from `bt`: regions, shifts, geno_offset_idx, batch, ploidy, keep*, to_rc (reverse only,
 * no complement), out_offsets (required: batch*ploidy + 1), max_row_len.
 * tracks f32 with track_offsets i64 (batch + 1): one reference track per query.
 * params: HOST pointer to the strategy's f64 parameter slot (1 value). */
int gvl_realign_tracks(const gvl_static *st, const gvl_batch *bt, const float *tracks,
                       const int64_t *track_offsets, const double *params, int64_t strategy_id,
                       uint64_t base_seed, float *out, void *stream);

/* One track's interval store (per dataset; the reference's RaggedIntervals, _dataset/_tracks.py):
 * list i = intervals [itv_offsets[i], itv_offsets[i + 1]), sorted by start. */
typedef struct gvl_track_set {
    const int32_t *itv_starts;
    const int32_t *itv_ends;
    const float *itv_values;
    const int64_t *itv_offsets;
    int64_t n_intervals;
    const int32_t *itv_pmax_ends; /* from gvl_intervals_prefix_max(); NULL = per-value painting only */
    /* optional coarse index (gvl_intervals_bucket_counts / _fill); all four or none */
    const int64_t *bkt_offsets;   /* n_lists + 1 */
    const int32_t *bkt_base;      /* n_lists     */
    const int32_t *bkt_lo;        /* n_buckets   */
    const int32_t *bkt_hi;        /* n_buckets   */
    int32_t tile_complete;        /* != 0: the caller vouches (checked once per interval set, see
                                     DeviceHapsTracksDataset) that inside every list starts strictly increase, no interval
                                     begins before its predecessor ends, and no two adjacent index buckets hold more than 256
                                     intervals.  gvl_tracks_batch / the native loop then realign the track straight from the
                                     intervals: no scratch track is written or read and the painter is not launched (a part of
                                     a list the claim does not hold for is looked up interval by interval: slower, never wrong).
                                     Where the painter still runs (GVL_DBG 4194304, gvl_intervals_to_tracks) it needs no second
                                     ("leftovers") launch; a chunk that would have needed it is reported by gvl_async_error().
                                     Needs the bucket index and itv_pmax_ends. */
    int32_t has_fill;             /* != 0: this track's own insertion fill (the reference lowers one per track,
                                     _reconstruct.py:204-208): fill_strategy / fill_param replace the call's strategy_id /
                                     params for this track.  0: the call's */
    int32_t fill_strategy;        /* GVL_FILL_* */
    double fill_param;
    int64_t list_div;             /* > 1: a REGION-level track (TrackType other than SAMPLE, _reconstruct.py:231-236): the
                                     list of a query is offset_idxs[q] / list_div (= dataset index / n_samples = r_idx)
                                     instead of offset_idxs[q].  0 / 1: per (region, sample) lists */
} gvl_track_set;

/* Coarse per-list index for the painter (once per interval set, next to gvl_intervals_prefix_max):
 * buckets of 2048 positions from each list's first start; per bucket the first interval whose running
 * maximum of ends passes the bucket's start and the first interval starting at or after its end.
 * Step 1 writes bkt_offsets (n_lists + 1, exclusive scan of the lists' bucket counts), bkt_base
 * (n_lists) and total[0] = number of buckets (device i64[2]); the caller sizes bkt_lo / bkt_hi from
 * it; step 2 fills them.  A painter with the index does one lookup per (query, 2048-value chunk)
 * instead of two dependent 64-ary searches over the query's list. */
int gvl_intervals_bucket_counts(const int32_t *itv_starts, const int64_t *itv_offsets, int64_t n_lists,
                                int64_t *bkt_offsets, int32_t *bkt_base, int64_t *total, void *stream);
int gvl_intervals_bucket_fill(const int32_t *itv_starts, const int32_t *itv_pmax_ends, const int64_t *itv_offsets,
                              int64_t n_lists, const int64_t *bkt_offsets, const int32_t *bkt_base,
                              int64_t n_buckets, int32_t *bkt_lo, int32_t *bkt_hi, void *stream);

/* intervals_to_tracks (src/intervals.rs:19-126) over an interval set that carries its derived arrays (ABI 7): with
 * ts->itv_pmax_ends + the bucket index the painter takes the tiled + bitmap path gvl_tracks_batch uses (the plain
 * gvl_intervals_to_tracks has no place for the index).  Same result, bit for bit.  ts->tile_complete as in gvl_track_set;
 * ts->list_div > 1: region-level lists.  starts / offset_idxs / out_offsets / max_row_len as gvl_intervals_to_tracks. */
int gvl_paint_tracks(const gvl_track_set *ts, const int64_t *offset_idxs, const int32_t *starts, int64_t starts_stride,
                     int64_t n_queries, float *out, const int64_t *out_offsets, int64_t max_row_len, void *stream);

/* The track half of a haplotypes + tracks batch in ONE call (what Haps/Tracks reconstruction does per
 * batch and track, _reconstruct.py:183-307 -> intervals_and_realign_track_fused, src/ffi/mod.rs:2551-2672):
 * scratch-track length per query = len - min_p(min(diff, 0)) (diff = get_diffs_sparse in query mode),
 * then per track: paint the query's intervals (list offset_idxs[q]) into its scratch track, realign it
 * to every haplotype (fill strategy / params / base_seed as gvl_realign_tracks), reverse negative-strand
 * rows.  bt: regions, shifts, geno_offset_idx, to_rc, batch, ploidy, output_length (fixed, >= 0).
 * out: f32, track t at out + t * out_track_stride, rows (batch * ploidy, output_length).
 * scratch: gvl_tracks_scratch_bytes(batch, ploidy, scratch_stride) bytes of the caller's device memory,
 * 256-byte aligned; scratch_stride = values reserved per query's scratch track (>= 2 * the longest
 * region is always enough).  No allocation, no host synchronisation. */
int64_t gvl_tracks_scratch_bytes(int64_t batch, int64_t ploidy, int64_t scratch_stride);
int gvl_tracks_batch(const gvl_static *st, const gvl_batch *bt, const int64_t *offset_idxs,
                     const gvl_track_set *tracks, int32_t n_tracks, const double *params,
                     int64_t strategy_id, uint64_t base_seed, float *out, int64_t out_track_stride,
                     void *scratch, int64_t scratch_stride, void *stream);

/* ---- device-side request prep -------------------------------------- */

/* Turn dataset indices over the (regions x samples) grid into the per-batch arrays of a
 * ReconstructionRequest, on the device.  Replaces, for fixed-length output:
 *   np.unravel_index + region gather + jitter + to_rc   (_dataset/_query.py:160-175),
 *   _get_geno_offset_idx / ravel_multi_index             (_dataset/_haps.py:757-768),
 *   get_diffs_sparse + random shifts                      (_dataset/_haps.py:715-730).
 * idx i64 (batch); full_regions i32 (n_regions, 4) [contig, start, end, strand].
 * Outputs: regions i32 (batch, 4), geno_offset_idx i64 (batch, ploidy), to_rc u8
 * (batch*ploidy; all zero when rc_neg == 0), shifts i32 (batch, ploidy).
 * jitter > 0: start += U{-jitter..jitter}; deterministic == 0: shift ~ U{0..max_shift} with
 * max_shift = max(diff, 0) + max(region_len - output_length, 0).  Random draws come from the
 * counter-based hash of src/tracks/mod.rs:31-54 keyed by (seed, counter, row): reproducible,
 * but not numpy's stream.  From `st` only the genotype CSR + v_starts / ilens are read. */
int gvl_prepare_request(const gvl_static *st, const int64_t *idx, int64_t batch,
                        const int32_t *full_regions, int64_t n_regions, int64_t n_samples,
                        int64_t ploidy, int64_t jitter, int32_t rc_neg, int32_t deterministic,
                        int64_t output_length, uint64_t seed, uint64_t counter,
                        int32_t *regions_out, int64_t *geno_offset_idx_out, uint8_t *to_rc_out,
                        int32_t *shifts_out, void *stream);

/* ---- native batch loop ----------------------------------------------- */

/* The producer side of Dataset.to_dataloader (_dataset/_impl.py:1963-2072) for device
 * consumers: what the reference does with a producer process + shared-memory double buffer
 * (_torch.py:94-211, _double_buffered_loader.py) is here a ring of output slots in HBM
 * filled `in_flight` batches ahead on the loader's own HIP streams.  Per batch the loader
 * launches gvl_prepare_request + gvl_reconstruct; nothing returns to the host.
 *
 * Slot memory is the caller's (n_slots device buffers of gvl_loader_slot_bytes() each,
 * 256-byte aligned), so a torch caller hands over torch-allocated arenas and builds views.
 * Contract: the batch returned by gvl_loader_next stays valid until the NEXT call of
 * gvl_loader_next (which records its release on `consumer_stream`); n_slots >= in_flight + 1. */
typedef struct gvl_loader gvl_loader;

typedef struct gvl_loader_config {
    const int32_t *full_regions; /* device i32 (n_regions, 4) [contig, start, end, strand] */
    int64_t n_regions, n_samples, ploidy;
    int64_t batch_size;          /* queries per batch (rows = batch_size * ploidy) */
    int64_t output_length;       /* fixed L >= 1 */
    int64_t jitter;              /* see gvl_prepare_request */
    int32_t rc_neg, deterministic;
    uint64_t seed;
    int32_t want_haps, want_onehot, onehot_layout;
    int32_t in_flight;           /* batches submitted ahead, one HIP stream each: 1..16 */
    int32_t n_slots;             /* ring slots: in_flight + 1 .. 64 */
    void *const *slot_arenas;    /* HOST array of n_slots device pointers */
    int32_t threaded;            /* != 0: a producer thread of the library submits the batches, so that the
                                    launches overlap the caller's own per-batch host work */
    int32_t group;               /* batches per launch (gvl_reconstruct_many): 0 / 1 .. GVL_MANY_MAX;
                                    in_flight then counts GROUPS, n_slots must be a multiple of group
                                    and >= (in_flight + 1) * group */
    /* ---- the reference loader's other outputs (_torch.py:94-211, _reconstruct.py:132-307) ---- */
    int32_t want_annot;          /* annotated haplotypes: annot_v_idxs / annot_ref_pos (i32 per base) next to haps
                                    (reconstruct_annotated_haplotypes_fused); forces want_haps */
    int64_t max_row_len;         /* RAGGED rows: output_length == -1 (deterministic only, row-major one-hot).  Row k
                                    has length region length + its haplotype's length delta (_haps.py:794-811),
                                    rows are packed back to back at out_offsets.  A slot holds
                                    batch_size * ploidy * max_row_len bases; a row longer than max_row_len is cut
                                    to it and reported by gvl_async_error() -- choose a true bound (longest
                                    region + the largest sum of insertion lengths of any genotype slot) */
    const gvl_track_set *tracks; /* HOST array of n_tracks interval stores (copied); list index = dataset index.
                                    Fixed-length rows only.  Per batch: gvl_tracks_batch into the slot */
    int32_t n_tracks;
    int32_t track_seed_mode;     /* 0: track_seed for every batch; 1: per batch like the reference
                                    (_reconstruct.py:215-222): deterministic -> xor of the batch's dataset
                                    indices, else a draw keyed by (seed, epoch, batch) */
    int64_t strategy_id;         /* insertion fill, see gvl_realign_tracks */
    double track_param;
    uint64_t track_seed;
    int64_t scratch_stride;      /* values reserved per query's scratch track (gvl_tracks_batch) */
} gvl_loader_config;

typedef struct gvl_loader_batch {
    int32_t slot;                /* ring slot, -1 = epoch finished */
    int64_t batch;               /* queries in this batch (the last one may be short) */
    const int64_t *idx;          /* device: this batch's dataset indices (into the epoch order) */
    uint8_t *onehot;             /* device pointers into the slot (NULL when not requested) */
    uint8_t *haps;
    int32_t *regions;            /* (batch, 4)              \                                      */
    int64_t *geno_offset_idx;    /* (batch, ploidy)          | rows of the loader's epoch table:      */
    int32_t *shifts;             /* (batch, ploidy)          | valid until the epoch ends             */
    uint8_t *to_rc;              /* (batch * ploidy)        /                                        */
    int64_t *out_offsets;        /* (batch * ploidy + 1), in the slot */
    int32_t *annot_v_idxs;       /* want_annot: i32 per base, laid out like haps */
    int32_t *annot_ref_pos;
    float *tracks;               /* n_tracks > 0: f32 (n_tracks, batch_size * ploidy, output_length) -- track t of
                                    a short last batch still starts at t * batch_size * ploidy * output_length */
    int64_t *sizes;              /* ragged: device i64[2] = {total bases of the batch, longest row} */
    const uint64_t *track_seed;  /* device: the base_seed this batch's FlankSample fills used (mode 1) */
} gvl_loader_batch;

/* Bytes of one slot and the offsets of its parts (GVL_LOADER_SLOT_PARTS values: onehot, haps, regions,
 * geno_offset_idx, shifts, to_rc, out_offsets, annot_v_idxs, annot_ref_pos, tracks, track scratch,
 * sizes), for a full batch.  (Parts 2-5 are unused since the request arrays live in the epoch table.) */
#define GVL_LOADER_SLOT_PARTS 12
#define GVL_LOADER_TABLE_PARTS 10
int64_t gvl_loader_slot_bytes(const gvl_loader_config *cfg, int64_t *part_offsets);
/* `st` is copied; the device arrays it points to must outlive the loader. */
int gvl_loader_create(const gvl_static *st, const gvl_loader_config *cfg, gvl_loader **out);
/* Begin an epoch over `order` (device i64[n] dataset indices, already shuffled by the caller;
 * must stay alive until the epoch ends).  `stream` is the stream `order` was produced on AND the stream the
 * consumer reads its batches on (the one passed to gvl_loader_next): batches of the previous epoch that are
 * still held by the consumer are released on it, and the new epoch's table is ordered behind it.  The
 * request arrays of the WHOLE epoch are prepared here, with one launch of the prep kernel
 * (gvl_prepare_request over all n indices) into `table`; random draws are keyed by
 * (cfg.seed, epoch number, dataset index) -- see gvl_loader_set_epoch. */
int gvl_loader_start_epoch(gvl_loader *ld, const int64_t *order, int64_t n, int32_t drop_last,
                           void *table, void *stream);
/* Prepare epoch number `epoch` AHEAD of its start: the same table fill as gvl_loader_start_epoch (same arguments,
 * same draws), into a table that is NOT the running epoch's, on `stream` -- typically right after the running epoch
 * was started, when nothing is queued in front of it.  A later gvl_loader_start_epoch with exactly these arguments
 * for exactly this epoch number (the next in sequence, or the one named by gvl_loader_set_epoch) then finds its table
 * filled: it launches nothing and -- because nothing of the running epoch is overwritten -- does not wait for that
 * epoch's last batches, so the first batches of the new epoch are queued behind them without a gap (without it an
 * epoch boundary costs the table fill + a pipeline refill: ~90 us for BASELINE config 4, 0.25-0.5 ms of a 1.6 ms
 * config-5 epoch).  Anything else (another order, table, length, epoch) is a normal start; a prepared epoch that is
 * never started costs its kernels only.  `order` and `table` must stay alive until that epoch has ended.
 * Contract: call it from the CONSUMER's thread (the one that calls gvl_loader_next / gvl_loader_start_epoch; with
 * cfg.threaded the library's producer thread may be submitting meanwhile: the fill touches only the OTHER table and
 * state that only the consumer's thread reads; error strings are per thread).  `stream` is the stream `order` was produced
 * on; it must be ordered BEHIND the last batch of the epoch that used `table` before (the prepared start waits for nothing:
 * the epoch's first batches wait for this fill through an event recorded on `stream`, but nothing else orders the fill
 * behind the batches that still read the table's previous contents): the consumer's stream once a batch of the running
 * epoch has been waited for there, or -- better, because the consumer's stream also carries the ring's slot releases and the
 * fill's kernels would hold them up -- a side stream that waits for an event recorded at that point of the consumer's
 * stream (what genvarloader_amd/loader.py does). */
int gvl_loader_prefetch_epoch(gvl_loader *ld, uint64_t epoch, const int64_t *order, int64_t n, int32_t drop_last,
                              void *table, void *stream);
/* Name the epoch the next gvl_loader_start_epoch begins (like DistributedSampler.set_epoch).  The jitter /
 * shift draw of dataset index i is a function of (cfg.seed, epoch, i) only -- not of the rank, the batch the
 * index lands in or the batch size -- so every rank passes the same number and an N-GPU epoch draws what the
 * 1-GPU epoch draws; a resumed run names the epoch it resumes.  Without this call epochs count from 0. */
int gvl_loader_set_epoch(gvl_loader *ld, uint64_t epoch);
/* Bytes of the epoch table for n queries and the offsets of its GVL_LOADER_TABLE_PARTS parts (regions i32
 * (n, 4), geno_offset_idx i64 (n, ploidy), shifts i32 (n, ploidy), to_rc u8 (n * ploidy), per-batch track
 * seeds u64 (ceil(n / batch_size)), and -- with tracks -- every batch's scratch-track offsets i64 (batch_size + 1
 * per batch), the k * output_length row offsets i64 (batch_size * ploidy + 1) the realignment reads and, for rows of several
 * 2048-value chunks, the rows' realignment plans (the row's entries, 2 KB per row, + 8 B per (row, chunk); left out when an
 * epoch's would exceed gvl_set_tuning(GVL_TUNE_TRACK_PLAN_MAX_MB), default 512); and, last, for fixed-length rows of several
 * 2048-base chunks the haplotype kernel's chunk plans of every row (gvl_hap_plan_bytes; same cap: an epoch without them plans per
 * launch); and for ragged rows (output_length = -1) every batch's row offsets i64 (batch_size * ploidy + 1 per batch) and its
 * {total, longest row} (2 i64 per batch): the rows' lengths are a function of the table's regions and the genotypes, so they are
 * sized once per epoch, not once per group of batches).  The table is the
 * caller's device memory (256-byte aligned) and must stay alive until the epoch ends; batch j's
 * request arrays are rows [j * batch_size, ...) of its parts. */
int64_t gvl_loader_table_bytes(const gvl_loader_config *cfg, int64_t n, int64_t *part_offsets);
/* Release the previously returned batch, top the pipeline up, and make `consumer_stream` wait
 * for the next batch.  Never blocks the host.  out->slot == -1 when the epoch is over. */
int gvl_loader_next(gvl_loader *ld, void *consumer_stream, gvl_loader_batch *out);
int gvl_loader_destroy(gvl_loader *ld);

/* ---- SVAR2 two-source variant provider (ABI 11) ------------------------------------------------------------
 *
 * The reference's second genotype source (src/svar2/mod.rs): a haplotype's variants are the merge of its `var_key` calls
 * (CSR vk_off, one slice per haplotype) with the PRESENT entries of its query's `dense` window (dense_range per query,
 * LSB-first presence bits per haplotype), position-sorted and stable -- var_key first on ties (merge_hap,
 * src/svar2/mod.rs:45-72).  In the reference every entry is (position, 32-bit key) and a key is decoded by the third-party
 * crate svar2-codec (a git dependency, not part of the reference's tree; its bit layout is not stated anywhere in it).
 * This boundary takes the channels DECODED -- the integrator, who links the codec, runs decode_key once per entry --
 * in the form the reference itself gives a decoded key (decode_alt, src/svar2/mod.rs:17-30; VariantsSoa, :292-306):
 *     entry e:  v_diff = ilen[e],  allele = alt_bytes[alt_off[e] .. alt_off[e + 1])
 * with an EMPTY allele for a pure deletion (the kernels then use the anchor base ref[pos], as the reference's provider
 * does: src/reconstruct/mod.rs:712-733).  Both channels' alleles live in one pool `alt_bytes`; each channel has its own
 * n + 1 offsets into it.
 *
 * gvl_svar2_merge turns one batch of channels into a per-batch sparse table in the caller's workspace -- per haplotype a
 * contiguous run of 16-byte records (gvl_grec) + its slot line (gvl_srec), the variant table behind them, and
 * geno_offset_idx = 0 .. batch*ploidy-1 -- and fills `merged`, a HOST gvl_static that points at it (and at st's
 * reference arrays).  Every entry point of this header that reads a gvl_static then works on the batch unchanged:
 *     hap_diffs_svar2 (src/svar2/mod.rs:78-160)                   = gvl_get_diffs_sparse(merged, ...) in query mode
 *     reconstruct_haplotypes_from_svar2 (src/ffi/mod.rs:874-997)  = gvl_hap_offsets + gvl_reconstruct(merged, ...)
 *     shift_and_realign_tracks_from_svar2 (src/ffi/mod.rs:1835-)  = gvl_hap_offsets + gvl_realign_tracks(merged, ...)
 * filter_exonic != 0 drops the entries that do not lie entirely inside their query's [start, end)
 * (src/reconstruct/mod.rs:699-706, src/svar2/mod.rs:131-133) during the merge.
 *
 * Contract (the reference's own: genoray emits position-sorted runs, src/svar2/mod.rs:373-378): inside a haplotype's
 * var_key slice and inside a dense window positions do not decrease; positions are < 2^31.  A haplotype with at most 64
 * var_key entries and a window of at most 64 entries is merged by rank counting and needs neither; a longer one that is
 * not sorted, a negative position, or offsets that leave their arrays are reported by gvl_async_error() (that
 * haplotype's records are then unspecified, never out of bounds).
 * The table is valid until the workspace is reused; `merged` holds no ownership.  (The merged table's alt_offsets are
 * allele STARTS only -- lengths live in the records -- so the gvl_pack_* builders must not be run on it.) */
typedef struct gvl_svar2_batch {
    const int32_t *vk_pos;            /* n_vk */
    const int32_t *vk_ilen;           /* n_vk: decoded v_diff */
    const int64_t *vk_alt_off;        /* n_vk + 1: allele of entry i = alt_bytes[vk_alt_off[i] .. vk_alt_off[i + 1]) */
    const int64_t *vk_off;            /* batch*ploidy + 1: haplotype k owns var_key entries [vk_off[k], vk_off[k + 1]) */
    int64_t n_vk;
    const int32_t *dense_pos;         /* n_dense */
    const int32_t *dense_ilen;        /* n_dense */
    const int64_t *dense_alt_off;     /* n_dense + 1 */
    int64_t n_dense;
    const int32_t *dense_range;       /* (batch, 2): query q's window = dense entries [ds, de) */
    const uint8_t *dense_present;     /* presence bits, LSB first inside a byte (src/svar2/mod.rs:35-39) */
    int64_t dense_present_bits;       /* bits in dense_present (>= dense_present_off[batch*ploidy]) */
    const int64_t *dense_present_off; /* batch*ploidy + 1 BIT offsets: bit dense_present_off[k] + j = haplotype k carries
                                         entry ds + j of its query's window */
    const uint8_t *alt_bytes;         /* the allele pool */
    int64_t alt_len;
    int32_t filter_exonic;
} gvl_svar2_batch;

int64_t gvl_svar2_workspace_bytes(int64_t batch, int64_t ploidy, int64_t n_vk, int64_t dense_present_bits, int64_t alt_len);
/* regions i32 (batch, regions_stride >= 3) [contig, start, end]: the contig of a pure deletion's anchor base and the
 * exonic filter's bounds.  workspace: gvl_svar2_workspace_bytes(...) bytes of device memory, 256-byte aligned.
 * merged (HOST, out): the batch's table; geno_offset_idx (HOST pointer to a device pointer, out): i64 (batch, ploidy). */
int gvl_svar2_merge(const gvl_static *st, const gvl_svar2_batch *sv, const int32_t *regions, int64_t regions_stride,
                    int64_t batch, int64_t ploidy, void *workspace, int64_t workspace_bytes, gvl_static *merged,
                    const int64_t **geno_offset_idx, void *stream);
/* gvl_svar2_merge + gvl_reconstruct in one call: bt as for gvl_reconstruct (geno_offset_idx is ignored; fixed-length rows,
 * rows at out_offsets or at out_bounds; keep masks and annotations are not part of the SVAR2 path:
 * src/reconstruct/mod.rs:746-749). */
int gvl_svar2_reconstruct(const gvl_static *st, const gvl_svar2_batch *sv, const gvl_batch *bt, const gvl_out *out,
                          void *workspace, int64_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GVL_HIP_H */
