// gvl_lean.hip -- recon_lean_kernel (gvl_lean.inc: a wave per row; rows of several chunks -- BASELINE config 4 -- by chunks) and its launcher.
#include "gvl_internal.inc"

namespace {
#include "gvl_dev.inc"
#include "gvl_recon_body.inc"
#include "gvl_lean.inc"
}  // namespace

namespace gvli {
static void fill_lean_args(LeanArgs &A, const ReconArgs &RA, int chunks) {
    memset(&A, 0, sizeof(A));
    A.ref4 = RA.ref4; A.ref_offsets = RA.ref_offsets; A.srec = RA.srec;
    A.regions = RA.regions; A.shifts = RA.shifts; A.geno_offset_idx = RA.geno_offset_idx; A.to_rc = RA.to_rc;
    A.onehot = RA.onehot; A.haps = RA.haps; A.out_offsets_w = RA.out_offsets_w; A.alt_alleles = RA.alt_alleles; A.stamps = RA.stamps;
    A.go_starts = RA.go_starts; A.go_stops = RA.go_stops; A.grec = RA.grec; A.alt_offsets = RA.alt_offsets;
    A.n_geno_offsets = RA.n_geno_offsets;
    A.n_rows = (int)RA.n_rows; A.n_contigs = RA.n_contigs; A.regions_stride = (int)RA.regions_stride;
    A.ploidy_shift = RA.ploidy_shift; A.ploidy = RA.ploidy; A.L = (int)RA.fixed_len; A.dbg = RA.dbg;
    A.chunks = chunks;
    A.ref_only = RA.ref_only;
    A.keep = RA.keep; A.keep_offsets = RA.keep_offsets;       // (read by the LONG forms and hap_plan_kernel only)
    A.av = RA.av; A.ap = RA.ap;
}

// The chunk plans of a batch's rows (hap_plan_kernel; rows of 2 .. HP_MAX_CHUNKS chunks, fixed length): `plan` holds
// hap_plan_bytes(rows, chunks) bytes.  gvl_hap_plan's launch -- the native loader makes every row's plan of an epoch with it -- and
// launch_lean's own when the batch comes without plans.
int launch_hap_plan(const ReconArgs &RA, int chunks, u32 *plan, void *stream) {
    LeanArgs A;
    fill_lean_args(A, RA, chunks);
    if (A.n_rows <= 0) return GVL_OK;
    hap_plan_kernel<<<dim3((unsigned)((A.n_rows + LEAN_WAVES - 1) / LEAN_WAVES)), dim3(LEAN_THREADS), 0, (hipStream_t)stream>>>(A, plan);
    return check_launch("gvl_hap_plan");
}

int launch_lean(const ReconArgs &RA, int chunks, void *stream) {
    LeanArgs A;
    fill_lean_args(A, RA, chunks);
    // the rows' chunk plans, when the caller brings them (gvl_batch.hap_plan: made once per epoch by the native loader; GVL_DBG &
    // 536870912: ignored).  A stand-alone launch does NOT make them for itself: measured (profiles/r05_cfg4_plans.txt), the
    // planner in front of every launch costs more than the walks it saves (7 us + a stream-ordered allocation against 2.7 us)
    const bool ann = chunks > 1 && RA.av && RA.ap;
    if (chunks > 1 && !RA.out_offsets && chunks <= HP_MAX_CHUNKS && !(debug_flags() & 536870912)) A.hplan = RA.hplan;
    // rows of several chunks: a wave takes `sub` consecutive chunks.  Without plans 2 -- the second chunk resumes the first one's
    // walk; BASELINE config 4's 256 rows x 64 chunks are then 8 192 waves, every wave slot of the part once --, with plans 1: there
    // is no walk to share, and 16 384 short waves start their reads under each other's stores (34.9 against 37.6 us,
    // profiles/r05_cfg4_plans.txt).  gvl_set_tuning(GVL_TUNE_LEAN_SUB) overrides.
    const i64 sub_t = tune(GVL_TUNE_LEAN_SUB);
    // (annotated rows: 1 with or without plans -- their waves spend two thirds of their time storing, and twice as many of them overlap
    // that with each other's walks: 66 against 75 us for 256 x 131 072 annotated rows without plans, tools/ann_parts.py)
    A.sub = chunks > 1 ? (sub_t > 0 ? (int)(sub_t > 64 ? 64 : sub_t) : ((A.hplan || ann) ? 1 : 2)) : 1;
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
    } else if (ann) {      // annotated haplotypes of several chunks (round 6): bytes + both streams (+ a one-hot in either layout)
        if (A.onehot && RA.oh_cl) recon_lean_kernel<true, true, true, false, true, true><<<g, b, 0, s>>>(A, RA);
        else if (A.onehot) recon_lean_kernel<true, true, true, false, false, true><<<g, b, 0, s>>>(A, RA);
        else recon_lean_kernel<false, true, true, false, false, true><<<g, b, 0, s>>>(A, RA);
    } else if (chunks > 1 && RA.oh_cl) {      // channel-major one-hot (rows, 4, L)
        if (A.haps) recon_lean_kernel<true, true, true, false, true><<<g, b, 0, s>>>(A, RA);
        else recon_lean_kernel<true, false, true, false, true><<<g, b, 0, s>>>(A, RA);
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

}  // namespace gvli
