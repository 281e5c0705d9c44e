// tools/wbench.hip: what the part sustains for WRITE-dominated kernels whose output does not fit the 256 MB Infinity Cache (config 4's
// tracks: 134 MB per batch; annotated long rows: 302 MB) -- store-only and copy kernels over outputs of 32 MB ... 1 GB, launches back to
// back over a rotation of buffers (>= 2 GB touched per timed region), 16 bytes per lane, one contiguous KB per store instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/wbench.hip -o tools/wbench.bin && tools/wbench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32;
typedef long long i64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

// a wave writes 8 KB chunks (8 store instructions of one KB), chunks handed out round robin over the grid's waves
template <bool NT>
__global__ __launch_bounds__(256) void k_store(u32x4 *out, i64 n_chunks) {
    const int lane = threadIdx.x & 63;
    const i64 w = (i64)blockIdx.x * 4 + (threadIdx.x >> 6), W = (i64)gridDim.x * 4;
    const u32x4 v = {(u32)w, 1u, 2u, 3u};
    for (i64 c = w; c < n_chunks; c += W) {
        u32x4 *p = out + c * 512 + lane;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (NT) __builtin_nontemporal_store(v, p + 64 * t); else p[64 * t] = v;
        }
    }
}
__global__ __launch_bounds__(256) void k_copy(const u32x4 *in, u32x4 *out, i64 n_chunks) {
    const int lane = threadIdx.x & 63;
    const i64 w = (i64)blockIdx.x * 4 + (threadIdx.x >> 6), W = (i64)gridDim.x * 4;
    for (i64 c = w; c < n_chunks; c += W) {
        const u32x4 *q = in + c * 512 + lane;
        u32x4 *p = out + c * 512 + lane;
        u32x4 r[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) r[t] = q[64 * t];
#pragma unroll
        for (int t = 0; t < 8; ++t) p[64 * t] = r[t];
    }
}

int main() {
    const i64 pool = 3ll << 30;                 // 3 GB of output buffers (+ 1 GB of input for the copies)
    u32x4 *out, *in;
    CK(hipMalloc(&out, pool)); CK(hipMalloc(&in, 1ll << 30));
    CK(hipMemset(in, 1, 1ll << 30)); CK(hipMemset(out, 0, pool));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-10s %-22s %10s %10s   (launches back to back over a rotation of buffers; GB/s of bytes written, copies: written + read)\n", "output", "kernel", "us/launch", "GB/s");
    for (i64 mb : {32ll, 64ll, 134ll, 256ll, 302ll, 512ll, 1024ll}) {
        const i64 bytes = mb << 20, n_chunks = bytes / 8192;
        const int n_buf = (int)(pool / bytes);
        const int n_in = (int)((1ll << 30) / bytes) > 0 ? (int)((1ll << 30) / bytes) : 1;
        const int reps = (int)((6ll << 30) / bytes) < 8 ? 8 : (int)((6ll << 30) / bytes);
        for (int kind = 0; kind < 4; ++kind) {
            for (int grid : {2048, 4096}) {
                if (kind != 0 && grid != 2048) continue;
                auto launch = [&](int i) {
                    u32x4 *o = out + (i64)(i % n_buf) * (bytes / 16);
                    if (kind == 0) k_store<false><<<grid, 256>>>(o, n_chunks);
                    else if (kind == 1) k_store<true><<<grid, 256>>>(o, n_chunks);
                    else if (kind == 2) k_copy<<<grid, 256>>>(in + (i64)(i % n_in) * (bytes / 16), o, n_chunks);
                    else k_store<false><<<(unsigned)(n_chunks / 4 < 65535 * 16 ? n_chunks / 4 : 65535 * 16), 256>>>(o, n_chunks);      // a wave per chunk
                };
                if (kind == 2 && bytes > (1ll << 30)) continue;
                for (int i = 0; i < 3; ++i) launch(i);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < reps; ++i) launch(i);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / reps;
                const char *nm = kind == 0 ? (grid == 2048 ? "store, 8192 waves" : "store, 16384 waves") : kind == 1 ? "store nt, 8192 waves" : kind == 2 ? "copy, 8192 waves" : "store, a wave per 8 KB";
                printf("%6lld MB  %-22s %10.1f %10.0f\n", mb, nm, us, (kind == 2 ? 2.0 : 1.0) * bytes / us / 1e3);
            }
        }
    }
    return 0;
}
