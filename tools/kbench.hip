// kbench.hip -- micro-benchmarks used to find the floor of the streaming part of the
// hot path on MI355X (not product code; results quoted in DESIGN.md).
//   hipcc -O3 --offload-arch=gfx950 tools/kbench.hip -o build/kbench && build/kbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32;
typedef unsigned char u8;
typedef long long i64;
struct __attribute__((aligned(4))) u32x4_a4 { u32 x, y, z, w; };
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ u32 oh(u32 b) {
    return b == 'A' ? 1u : b == 'C' ? 0x100u : b == 'G' ? 0x10000u : b == 'T' ? 0x1000000u : 0u;
}

// V0: store only. one wave = one row of L bases, 4 bases/lane/trip.
template <int NT>
__global__ __launch_bounds__(256) void k_store(u8 *out, int L, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 row = (i64)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    u8 *o = out + row * 4 * (i64)L;
    for (int p0 = 0; p0 < L; p0 += 256) {
        const int p = p0 + 4 * lane;
        u32x4 v = {1u, 0x100u, 0x10000u, (u32)p};
        if (NT) __builtin_nontemporal_store(v, (u32x4 *)(o + 4 * (i64)p));
        else *(u32x4 *)(o + 4 * (i64)p) = v;
    }
}

// V1: load 4 ref bytes (unaligned) + one-hot via LDS LUT + store.
template <int U, int NT>
__global__ __launch_bounds__(256) void k_stream(const u8 *ref, const i64 *starts, u8 *out, int L, int rows) {
    __shared__ u32 lut[256];
    lut[threadIdx.x] = oh(threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 row = (i64)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const i64 s = starts[row];
    u8 *o = out + row * 4 * (i64)L;
    for (int b0 = 0; b0 < L; b0 += 256 * U) {
        u32 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = b0 + u * 256 + 4 * lane;
            u32 v;
            __builtin_memcpy(&v, ref + s + p, 4);
            w[u] = v;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = b0 + u * 256 + 4 * lane;
            u32x4 v = {lut[w[u] & 0xFF], lut[(w[u] >> 8) & 0xFF], lut[(w[u] >> 16) & 0xFF], lut[w[u] >> 24]};
            if (NT) __builtin_nontemporal_store(v, (u32x4 *)(o + 4 * (i64)p));
            else *(u32x4 *)(o + 4 * (i64)p) = v;
        }
    }
}

// V2: as V1 but with a 3-level dependent chain in front (row -> idx -> start).
template <int U>
__global__ __launch_bounds__(256) void k_chain(const u8 *ref, const i64 *starts, const int *a1, const int *a2,
                                                u8 *out, int L, int rows) {
    __shared__ u32 lut[256];
    lut[threadIdx.x] = oh(threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 row = (i64)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int i1 = __builtin_amdgcn_readfirstlane(a1[row]);
    const int i2 = __builtin_amdgcn_readfirstlane(a2[i1]);
    const i64 s = starts[i2];
    u8 *o = out + row * 4 * (i64)L;
    for (int b0 = 0; b0 < L; b0 += 256 * U) {
        u32 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = b0 + u * 256 + 4 * lane;
            u32 v;
            __builtin_memcpy(&v, ref + s + p, 4);
            w[u] = v;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = b0 + u * 256 + 4 * lane;
            u32x4 v = {lut[w[u] & 0xFF], lut[(w[u] >> 8) & 0xFF], lut[(w[u] >> 16) & 0xFF], lut[w[u] >> 24]};
            *(u32x4 *)(o + 4 * (i64)p) = v;
        }
    }
}

// V3: flat grid-stride stream (no row structure): the plain copy-like ceiling.
__global__ __launch_bounds__(256) void k_flat(const u8 *ref, u8 *out, i64 n4) {
    __shared__ u32 lut[256];
    lut[threadIdx.x] = oh(threadIdx.x);
    __syncthreads();
    for (i64 g = (i64)blockIdx.x * 256 + threadIdx.x; g < n4; g += (i64)gridDim.x * 256) {
        u32 w;
        __builtin_memcpy(&w, ref + 4 * g, 4);
        u32x4 v = {lut[w & 0xFF], lut[(w >> 8) & 0xFF], lut[(w >> 16) & 0xFF], lut[w >> 24]};
        *(u32x4 *)(out + 16 * g) = v;
    }
}

template <typename F>
float timeit(F f, int iters = 300) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 30; ++i) f(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f(i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1000.f;  // us
}

int main() {
    const int rows = 4096, L = 2048;
    const i64 ref_len = 64ll << 20;
    u8 *ref, *out[2];
    i64 *starts; int *a1, *a2;
    CK(hipMalloc(&ref, ref_len + 4096));
    CK(hipMalloc(&out[0], (size_t)rows * L * 4)); CK(hipMalloc(&out[1], (size_t)rows * L * 4));
    CK(hipMalloc(&starts, rows * 8)); CK(hipMalloc(&a1, rows * 4)); CK(hipMalloc(&a2, rows * 4));
    std::vector<u8> h(ref_len);
    srand(1);
    for (i64 i = 0; i < ref_len; ++i) h[i] = "ACGT"[rand() & 3];
    CK(hipMemcpy(ref, h.data(), ref_len, hipMemcpyHostToDevice));
    std::vector<i64> hs(rows); std::vector<int> h1(rows), h2(rows);
    for (int r = 0; r < rows; ++r) {
        hs[r] = ((i64)rand() * 7919 + rand()) % (ref_len - L - 64);
        if (r & 1) hs[r] = hs[r - 1];  // two haps share a window
        h1[r] = (r * 2654435761u) % rows; h2[r] = (r * 40503u + 17) % rows;
    }
    CK(hipMemcpy(starts, hs.data(), rows * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(a1, h1.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(a2, h2.data(), rows * 4, hipMemcpyHostToDevice));
    const double mb = rows * (double)L * 5 / 1e6;
    auto rep = [&](const char *name, float us) { printf("%-28s %8.2f us   %7.1f GB/s (5 B/base)\n", name, us, mb / us * 1e3 / 1e3 * 1e3 / 1e3 * 1e3 / 1e3 * 1e0); };
    (void)rep;
    auto pr = [&](const char *name, float us) { printf("%-28s %8.2f us   %7.1f GB/s\n", name, us, mb * 1e6 / (us * 1e-6) / 1e9); };
    pr("store-only", timeit([&](int i) { k_store<0><<<rows / 4, 256>>>(out[i & 1], L, rows); }));
    pr("store-only nt", timeit([&](int i) { k_store<1><<<rows / 4, 256>>>(out[i & 1], L, rows); }));
    pr("stream U=1", timeit([&](int i) { k_stream<1, 0><<<rows / 4, 256>>>(ref, starts, out[i & 1], L, rows); }));
    pr("stream U=2", timeit([&](int i) { k_stream<2, 0><<<rows / 4, 256>>>(ref, starts, out[i & 1], L, rows); }));
    pr("stream U=4", timeit([&](int i) { k_stream<4, 0><<<rows / 4, 256>>>(ref, starts, out[i & 1], L, rows); }));
    pr("stream U=8", timeit([&](int i) { k_stream<8, 0><<<rows / 4, 256>>>(ref, starts, out[i & 1], L, rows); }));
    pr("stream U=8 nt", timeit([&](int i) { k_stream<8, 1><<<rows / 4, 256>>>(ref, starts, out[i & 1], L, rows); }));
    pr("chain3 + stream U=8", timeit([&](int i) { k_chain<8><<<rows / 4, 256>>>(ref, starts, a1, a2, out[i & 1], L, rows); }));
    pr("flat grid 2048 blocks", timeit([&](int i) { k_flat<<<2048, 256>>>(ref, out[i & 1], (i64)rows * L / 4); }));
    pr("flat grid 8192 blocks", timeit([&](int i) { k_flat<<<8192, 256>>>(ref, out[i & 1], (i64)rows * L / 4); }));
    pr("empty-ish (1 row)", timeit([&](int i) { k_store<0><<<1, 256>>>(out[i & 1], L, 1); }));
    return 0;
}
