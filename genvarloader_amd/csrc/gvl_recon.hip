// gvl_recon.hip -- the all-purpose kernel: reconstruct_kernel<OH, HAPS, ANNOT> (every row the reference's walk can produce; annotated,
// keep-mask and get_reference launches) and its launcher.
#include "gvl_internal.inc"

namespace {
#include "gvl_dev.inc"
#include "gvl_recon_body.inc"
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

}  // namespace

namespace gvli {
int launch_recon(const ReconArgs &A, int chunks, int variant, void *stream) {
    const i64 grid = (A.n_rows + WG_WAVES - 1) / WG_WAVES;
    if (grid <= 0) return GVL_OK;
    recon_fn fn = recon_table(variant & 3, (variant & 4) != 0, (variant & 8) != 0);
    fn<<<dim3((unsigned)grid, (unsigned)chunks), dim3(WG_THREADS), 0, (hipStream_t)stream>>>(A);
    return check_launch("gvl_reconstruct");
}

}  // namespace gvli
