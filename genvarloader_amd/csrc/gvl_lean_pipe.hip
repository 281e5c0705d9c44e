// gvl_lean_pipe.hip -- recon_lean_rows_kernel (gvl_lean_pipe.inc: one grid over a group of batches, a wave pipelines over rows) and its launcher.
#include "gvl_internal.inc"

namespace {
#include "gvl_dev.inc"
#include "gvl_recon_body.inc"
#include "gvl_lean.inc"
#include "gvl_lean_pipe.inc"
}  // namespace

namespace gvli {
// can these (lean-eligible, one-chunk) batches share a grid?  the same shape and outputs; every batch but the last has the
// first one's row count
bool lean_pipe_compatible(const ReconArgs *RAs, int n) {
    const ReconArgs &F = RAs[0];
    if (((uintptr_t)F.ref4 & 15) || ((uintptr_t)F.srec & 15)) return false;        // (16-byte DMA sources)
    i64 total = 0;
    for (int i = 0; i < n; ++i) {
        const ReconArgs &R = RAs[i];
        if (R.fixed_len != F.fixed_len || R.ploidy != F.ploidy || R.regions_stride != F.regions_stride ||
            (R.out_offsets != nullptr) != (F.out_offsets != nullptr) ||
            (R.onehot != nullptr) != (F.onehot != nullptr) || (R.haps != nullptr) != (F.haps != nullptr) || R.dbg != F.dbg ||
            R.oh_cl != F.oh_cl || (R.av != nullptr) != (F.av != nullptr) || (R.ap != nullptr) != (F.ap != nullptr))
            return false;
        if (R.n_rows <= 0 || R.n_rows > F.n_rows || (i + 1 < n && R.n_rows != F.n_rows)) return false;
        total += R.n_rows;
    }
    return total <= 0x7FFFFFF0ll && F.regions_stride <= 0x7FFFFFFFll;
}
int launch_lean_rows(const ReconArgs *RAs, int n, void *stream, int rag_chunks) {
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
    A.ref_only = RA.ref_only;
    A.slot_vidx = RA.slot_vidx;
    A.rows_per_batch = (int)RA.n_rows; A.n_batches = n;
    A.max_row_len = (int)((i64)RA.chunk_len * (RA.out_offsets ? rag_chunks : 1));
    i64 total = 0, longest = 0;
    for (int i = 0; i < n; ++i) {
        LeanBatch &b = M.b[i];
        b.regions = RAs[i].regions; b.shifts = RAs[i].shifts; b.geno_offset_idx = RAs[i].geno_offset_idx; b.to_rc = RAs[i].to_rc;
        b.onehot = RAs[i].onehot; b.haps = RAs[i].haps; b.out_offsets_w = RAs[i].out_offsets_w; b.n_rows = RAs[i].n_rows;
        b.out_offsets = RAs[i].out_offsets;
        b.keep = RAs[i].keep; b.keep_offsets = RAs[i].keep_offsets;
        b.av = RAs[i].av; b.ap = RAs[i].ap;
        // (a batch's own bound: its chunks where the caller's call said so -- gvl_reconstruct(_many) --, the launch's otherwise)
        const int own_chunks = RAs[i].row_chunks > 0 ? RAs[i].row_chunks : rag_chunks;
        b.max_row_len = (i64)RAs[i].chunk_len * (RAs[i].out_offsets ? own_chunks : 1);
        longest = b.max_row_len > longest ? b.max_row_len : longest;
        total += RAs[i].n_rows;
    }
    A.n_rows = (int)total;
    // Rows per wave (x 100).  Measured (profiles/r04_pipe_experiments.txt G, K): ONE row per wave -- no row-to-row prefetch at
    // all -- is the best schedule up to ~12 batches per launch (short waves: the hardware's workgroup dispatch balances the chip);
    // above that "two rows per wave", which is 1.5 on average: the first half of the waves take two rows (w, w + W), the second
    // half -- dispatched last -- one, so the grid drains in short waves (125 / 175 measure like 150; exactly 2, or 3, are slower).
    // gvl_set_tuning(GVL_TUNE_PIPE_ROWS_X100) overrides (200 = exactly two rows for every wave, 300 = three, ...).
    //
    // SHORT rows (round 6; tools/short_rows.py, tools/stamps_pipe.py, profiles/r06_short_rows*.txt, r06_stamps_pipe*.txt): a launch of
    // 65 536 rows takes 52-56 us whether its rows have 128 or 1024 bases -- a SIMD gets through ~1.4 rows per us whatever the schedule
    // (8 waves of 1.5 rows, 7 of 9, 4 of 16: a row is ~900 instructions of every kind, and a wave alone needs ~3 us for them) -- so more
    // rows per wave buy nothing there either: one generation of waves with total / waves rows each measured 61 us against 53.
    i64 x100 = tune(GVL_TUNE_PIPE_ROWS_X100);
    if (x100 < 100) x100 = total >= 49152 ? 150 : 100;
    if (x100 > 100 * (i64)PIPE_MAX_ROWS) x100 = 100 * (i64)PIPE_MAX_ROWS;      // (a wave's deferred-rows mask has a bit per row)
    i64 waves = (total * 100 + x100 - 1) / x100;
    if (debug_flags() & 33554432) {
        i64 rpw = (total + LEAN_WAVES - 1) / LEAN_WAVES;
        rpw = rpw > PIPE_MAX_ROWS ? PIPE_MAX_ROWS : (rpw < 1 ? 1 : rpw);
        waves = (total + rpw - 1) / rpw;
    }
    // ragged rows beyond the streaming loop's trips (the caller's bound says there may be some): one front workgroup per 256 rows finds
    // them and runs their chunks in parallel (lean_solo_rows' crews); GVL_DBG & 256: by the wave that meets them, as before
    // (several batches: the LARGEST bound any of them gave decides)
    if (RA.out_offsets && longest > PIPE_RAG_MAXT * TRIP && !(debug_flags() & 256)) A.front = (int)((total + LEAN_THREADS - 1) / LEAN_THREADS);
    const unsigned grid = (unsigned)((waves + LEAN_WAVES - 1) / LEAN_WAVES) + (unsigned)A.front;
    const dim3 g(grid), b(LEAN_THREADS);
    hipStream_t s = (hipStream_t)stream;
    // (launches without a keep mask that are not gvl_get_reference's: the forms with both compiled out; GVL_DBG & 1073741824 keeps the general ones)
    bool km = A.ref_only != 0 || (A.dbg & 1073741824);
    for (int i = 0; i < n; ++i) km = km || RAs[i].keep != nullptr || RAs[i].keep_offsets != nullptr;
    if (RA.av && RA.ap) {          // annotated haplotypes: bytes + the two annotation streams
        if (RA.out_offsets && A.onehot) recon_lean_rows_kernel<true, true, true, false, true><<<g, b, 0, s>>>(A, RA, M);
        else if (RA.out_offsets) recon_lean_rows_kernel<false, true, true, false, true><<<g, b, 0, s>>>(A, RA, M);
        else if (A.onehot && RA.oh_cl) recon_lean_rows_kernel<true, true, false, true, true><<<g, b, 0, s>>>(A, RA, M);
        else if (A.onehot) recon_lean_rows_kernel<true, true, false, false, true><<<g, b, 0, s>>>(A, RA, M);
        else recon_lean_rows_kernel<false, true, false, false, true><<<g, b, 0, s>>>(A, RA, M);
    } else if (RA.out_offsets && km) {
        if (A.onehot && A.haps) recon_lean_rows_kernel<true, true, true><<<g, b, 0, s>>>(A, RA, M);
        else if (A.onehot) recon_lean_rows_kernel<true, false, true><<<g, b, 0, s>>>(A, RA, M);
        else recon_lean_rows_kernel<false, true, true><<<g, b, 0, s>>>(A, RA, M);
    } else if (RA.out_offsets) {
        if (A.onehot && A.haps) recon_lean_rows_kernel<true, true, true, false, false, false><<<g, b, 0, s>>>(A, RA, M);
        else if (A.onehot) recon_lean_rows_kernel<true, false, true, false, false, false><<<g, b, 0, s>>>(A, RA, M);
        else recon_lean_rows_kernel<false, true, true, false, false, false><<<g, b, 0, s>>>(A, RA, M);
    } else if (RA.oh_cl) {       // channel-major one-hot (rows, 4, L)
        if (A.haps) recon_lean_rows_kernel<true, true, false, true><<<g, b, 0, s>>>(A, RA, M);
        else recon_lean_rows_kernel<true, false, false, true><<<g, b, 0, s>>>(A, RA, M);
    } else if (km) {
        if (A.onehot && A.haps) recon_lean_rows_kernel<true, true, false><<<g, b, 0, s>>>(A, RA, M);
        else if (A.onehot) recon_lean_rows_kernel<true, false, false><<<g, b, 0, s>>>(A, RA, M);
        else recon_lean_rows_kernel<false, true, false><<<g, b, 0, s>>>(A, RA, M);
    } else {
        if (A.onehot && A.haps) recon_lean_rows_kernel<true, true, false, false, false, false><<<g, b, 0, s>>>(A, RA, M);
        else if (A.onehot) recon_lean_rows_kernel<true, false, false, false, false, false><<<g, b, 0, s>>>(A, RA, M);
        else recon_lean_rows_kernel<false, true, false, false, false, false><<<g, b, 0, s>>>(A, RA, M);
    }
    return check_launch("gvl_reconstruct (lean, pipelined)");
}

}  // namespace gvli
