// tools/coldbench.hip: what do scattered COLD 2 KB window reads cost next to a streaming one-hot store, as a
// function of the footprint the windows are drawn from?  (No reconstruction logic: the memory system alone.)
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/coldbench.hip -o tools/coldbench.bin
// Per footprint F: 64 rotating tables of 4096 window starts (pairs share a window), launches back to back on
// one stream; kernels: read + one-hot store (nt), read only (one dword per lane written), store only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
typedef unsigned char u8; typedef unsigned int u32; typedef long long i64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ u32 oh(u32 b) { return b == 'A' ? 1u : b == 'C' ? 0x100u : b == 'G' ? 0x10000u : b == 'T' ? 0x1000000u : 0u; }

template <int MODE>   // 0 read + store, 1 read only, 2 store only
__global__ __launch_bounds__(256) void k_win(const u8 *ref, const i64 *starts, u8 *out, u32 *sink, int L, int rows) {
    __shared__ u32 lut[256];
    lut[threadIdx.x] = oh(threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 row = (i64)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const i64 s = MODE == 2 ? 0 : starts[row];
    u8 *o = out + row * 4 * (i64)L;
    u32 w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = u * 256 + 4 * lane;
        u32 v = 0x41434754u;
        if (MODE != 2) __builtin_memcpy(&v, ref + s + p, 4);
        w[u] = v;
    }
    u32 acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = u * 256 + 4 * lane;
        u32x4 v = {lut[w[u] & 0xFF], lut[(w[u] >> 8) & 0xFF], lut[(w[u] >> 16) & 0xFF], lut[w[u] >> 24]};
        if (MODE != 1) __builtin_nontemporal_store(v, (u32x4 *)(o + 4 * (i64)p));
        else acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (MODE == 1 && acc == 0x12345678u) sink[row] = acc;
}

// MODE 3: the reconstruct kernel's memory shape without its arithmetic.  Wave 0 of a 512-thread workgroup
// reads 8 rows' slot indices (coalesced, cold), then their 128-byte slot lines (random, cold) and parks them
// in LDS; the 8 waves read their windows meanwhile (independent of the slot line), wait for the barrier, and
// stream the one-hot out.
// k_shape with the request entries spread over `NARR` more cold arrays (regions / shifts / to_rc / ... of the C-ABI):
// wave 0 reads 8 rows' entries from each before it can go for the slot lines
template <int NARR>
__global__ __launch_bounds__(512) void k_shape_req(const u8 *ref, const i64 *starts, const i64 *slot_idx, const u32x4 *slots,
                                                   const int *req, i64 req_stride, u8 *out, int L, int rows) {
    __shared__ u32 lut[256];
    __shared__ u32x4 lrec[64];
    __shared__ int dep[8];
    if (threadIdx.x < 256) lut[threadIdx.x] = oh(threadIdx.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 row = (i64)blockIdx.x * 8 + wave;
    if (wave == 0) {
        const i64 r8 = (i64)blockIdx.x * 8 + (lane >> 3);
        int acc = 0;
#pragma unroll
        for (int a = 0; a < NARR; ++a) acc += req[(i64)a * req_stride + (r8 < rows ? r8 : 0)];      // level 1: NARR small reads
        const i64 si = slot_idx[r8 < rows ? r8 : 0] + (acc & 0);                                      // (level 2 depends on them)
        lrec[lane] = slots[si * 8 + (lane & 7)];
        if (lane < 8) dep[lane] = acc & 0;
    }
    __syncthreads();
    const i64 s = starts[row < rows ? row : 0] + dep[wave];
    u32 w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { u32 v; __builtin_memcpy(&v, ref + s + u * 256 + 4 * lane, 4); w[u] = v; }
    if (row >= rows) return;
    const u32 salt = lrec[wave * 8 + (lane & 7)].x & 0x20202020u;
    u8 *o = out + row * 4 * (i64)L;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = u * 256 + 4 * lane;
        const u32 x = w[u] | salt;
        u32x4 v = {lut[x & 0xFF], lut[(x >> 8) & 0xFF], lut[(x >> 16) & 0xFF], lut[x >> 24]};
        __builtin_nontemporal_store(v, (u32x4 *)(o + 4 * (i64)p));
    }
}

__global__ __launch_bounds__(512) void k_shape(const u8 *ref, const i64 *starts, const i64 *slot_idx, const u32x4 *slots,
                                               u8 *out, int L, int rows, int hold_ticks = 0, int hold_before_loads = 0) {
    __shared__ u32 lut[256];
    __shared__ u32x4 lrec[64];
    if (threadIdx.x < 256) lut[threadIdx.x] = oh(threadIdx.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 row = (i64)blockIdx.x * 8 + wave;
    if (wave == 0) {
        const i64 r8 = (i64)blockIdx.x * 8 + (lane >> 3);
        const i64 si = slot_idx[r8 < rows ? r8 : 0];
        lrec[lane] = slots[si * 8 + (lane & 7)];
    }
    const i64 s = starts[row < rows ? row : 0];
    // hold: the wave keeps its slot for `hold_ticks` x 10 ns doing nothing (the reconstruct kernel's plan takes
    // microseconds between its reads and its stores), either before or after its window reads are out
    if (hold_before_loads) { const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < hold_before_loads) __builtin_amdgcn_s_sleep(8); }
    u32 w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { u32 v; __builtin_memcpy(&v, ref + s + u * 256 + 4 * lane, 4); w[u] = v; }
    __syncthreads();
    if (hold_ticks) { const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < hold_ticks) __builtin_amdgcn_s_sleep(8); }
    if (row >= rows) return;
    const u32 salt = lrec[wave * 8 + (lane & 7)].x & 0x20202020u;      // (the records take part in the bytes)
    u8 *o = out + row * 4 * (i64)L;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = u * 256 + 4 * lane;
        const u32 x = w[u] | salt;
        u32x4 v = {lut[x & 0xFF], lut[(x >> 8) & 0xFF], lut[(x >> 16) & 0xFF], lut[x >> 24]};
        __builtin_nontemporal_store(v, (u32x4 *)(o + 4 * (i64)p));
    }
}

int main(int argc, char **argv) {
    const int rows = 4096, L = 2048, T = 64;
    const i64 max_fp = (argc > 1 ? atoll(argv[1]) : 6144ll) << 20;
    u8 *ref, *out[4]; i64 *starts; u32 *sink;
    CK(hipMalloc(&ref, max_fp + 4096));
    CK(hipMemset(ref, 'A', max_fp + 4096));
    for (int i = 0; i < 4; ++i) CK(hipMalloc(&out[i], (size_t)rows * L * 4));
    CK(hipMalloc(&starts, (size_t)T * rows * 8)); CK(hipMalloc(&sink, rows * 4));
    std::mt19937_64 rng(7);
    hipStream_t st[4];
    for (int i = 0; i < 4; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    const i64 n_slots = 16ll << 20;                                 // 2 GB of 128-byte slot lines
    u32x4 *slots; i64 *slot_idx;
    CK(hipMalloc(&slots, (size_t)n_slots * 128)); CK(hipMemset(slots, 0, (size_t)n_slots * 128));
    CK(hipMalloc(&slot_idx, (size_t)T * rows * 8));
    {
        std::vector<i64> hi((size_t)T * rows);
        for (auto &x : hi) x = (i64)(rng() % (unsigned long long)n_slots);
        CK(hipMemcpy(slot_idx, hi.data(), hi.size() * 8, hipMemcpyHostToDevice));
    }
    printf("%-10s %-7s %12s %12s %12s %14s %14s %14s\n", "footprint", "align", "read+store", "read only", "store only", "r+s 4 streams", "shape 1 str", "shape 4 str");
    for (i64 fp : {64ll << 20, 256ll << 20, 512ll << 20, 1024ll << 20, 2048ll << 20, 3072ll << 20, 6144ll << 20}) {
        if (fp > max_fp) break;
        for (int aligned = 0; aligned < 2; ++aligned) {
            std::vector<i64> hs((size_t)T * rows);
            for (size_t r = 0; r < hs.size(); ++r) {
                i64 s = (i64)(rng() % (unsigned long long)(fp - L - 64));
                if (aligned) s &= ~(i64)2047;                     // window = one aligned 2 KB block
                hs[r] = (r & 1) ? hs[r - 1] : s;                 // two haplotypes share a window
            }
            CK(hipMemcpy(starts, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
            auto timeit = [&](auto launch, int n_streams) {
                hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
                for (int i = 0; i < 64; ++i) launch(i, st[i % n_streams]);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(a, st[0]));
                const int iters = 640;
                for (int i = 0; i < iters; ++i) launch(i, st[i % n_streams]);
                for (int i = 1; i < n_streams; ++i) { hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, st[i])); CK(hipStreamWaitEvent(st[0], ev, 0)); }
                CK(hipEventRecord(b, st[0])); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                return ms / iters * 1000.f;
            };
            const float t0 = timeit([&](int i, hipStream_t s) { k_win<0><<<rows / 4, 256, 0, s>>>(ref, starts + (size_t)(i % T) * rows, out[i & 3], sink, L, rows); }, 1);
            const float t1 = timeit([&](int i, hipStream_t s) { k_win<1><<<rows / 4, 256, 0, s>>>(ref, starts + (size_t)(i % T) * rows, out[i & 3], sink, L, rows); }, 1);
            const float t2 = timeit([&](int i, hipStream_t s) { k_win<2><<<rows / 4, 256, 0, s>>>(ref, starts + (size_t)(i % T) * rows, out[i & 3], sink, L, rows); }, 1);
            const float t3 = timeit([&](int i, hipStream_t s) { k_win<0><<<rows / 4, 256, 0, s>>>(ref, starts + (size_t)(i % T) * rows, out[i & 3], sink, L, rows); }, 4);
            const float t4 = timeit([&](int i, hipStream_t s) { k_shape<<<rows / 8, 512, 0, s>>>(ref, starts + (size_t)(i % T) * rows, slot_idx + (size_t)(i % T) * rows, slots, out[i & 3], L, rows); }, 1);
            const float t5 = timeit([&](int i, hipStream_t s) { k_shape<<<rows / 8, 512, 0, s>>>(ref, starts + (size_t)(i % T) * rows, slot_idx + (size_t)(i % T) * rows, slots, out[i & 3], L, rows); }, 4);
            printf("%6lld MB  %-7s %9.2f us %9.2f us %9.2f us %11.2f us %11.2f us %11.2f us\n", fp >> 20, aligned ? "2 KB" : "byte", t0, t1, t2, t3, t4, t5);
        }
    }
    // the request entries as 0 / 2 / 4 separate cold arrays in front of the slot lines; window reads AFTER the barrier
    {
        int *req; const i64 req_stride = (i64)T * rows;
        CK(hipMalloc(&req, (size_t)4 * req_stride * 4)); CK(hipMemset(req, 0, (size_t)4 * req_stride * 4));
        std::vector<i64> hs((size_t)T * rows);
        const i64 fp = max_fp < (3072ll << 20) ? max_fp : (3072ll << 20);
        for (size_t r = 0; r < hs.size(); ++r) { i64 s = (i64)(rng() % (unsigned long long)(fp - L - 64)); hs[r] = (r & 1) ? hs[r - 1] : s; }
        CK(hipMemcpy(starts, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
        printf("kernel shape with the window reads behind P1 (as in the reconstruct kernel) and N more cold request arrays: 1 / 4 streams\n");
        auto run = [&](auto launch) {
            float t[2];
            for (int k = 0; k < 2; ++k) {
                const int ns = k ? 4 : 1;
                hipEvent_t a, b2; CK(hipEventCreate(&a)); CK(hipEventCreate(&b2));
                for (int i = 0; i < 64; ++i) launch(i, st[i % ns]);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(a, st[0]));
                for (int i = 0; i < 640; ++i) launch(i, st[i % ns]);
                for (int i = 1; i < ns; ++i) { hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, st[i])); CK(hipStreamWaitEvent(st[0], ev, 0)); }
                CK(hipEventRecord(b2, st[0])); CK(hipEventSynchronize(b2));
                float ms; CK(hipEventElapsedTime(&ms, a, b2)); t[k] = ms / 640 * 1000.f;
            }
            printf("%7.2f us  %7.2f us\n", t[0], t[1]);
        };
        printf("  N = 0: "); run([&](int i, hipStream_t s_) { k_shape_req<0><<<rows / 8, 512, 0, s_>>>(ref, starts + (size_t)(i % T) * rows, slot_idx + (size_t)(i % T) * rows, slots, req + (size_t)(i % T) * rows, req_stride, out[i & 3], L, rows); });
        printf("  N = 2: "); run([&](int i, hipStream_t s_) { k_shape_req<2><<<rows / 8, 512, 0, s_>>>(ref, starts + (size_t)(i % T) * rows, slot_idx + (size_t)(i % T) * rows, slots, req + (size_t)(i % T) * rows, req_stride, out[i & 3], L, rows); });
        printf("  N = 4: "); run([&](int i, hipStream_t s_) { k_shape_req<4><<<rows / 8, 512, 0, s_>>>(ref, starts + (size_t)(i % T) * rows, slot_idx + (size_t)(i % T) * rows, slots, req + (size_t)(i % T) * rows, req_stride, out[i & 3], L, rows); });
    }
    // waves that live longer (3 GB footprint, byte-aligned windows, 4 streams)
    {
        std::vector<i64> hs((size_t)T * rows);
        const i64 fp = max_fp < (3072ll << 20) ? max_fp : (3072ll << 20);
        for (size_t r = 0; r < hs.size(); ++r) { i64 s = (i64)(rng() % (unsigned long long)(fp - L - 64)); hs[r] = (r & 1) ? hs[r - 1] : s; }
        CK(hipMemcpy(starts, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
        printf("kernel shape, %lld MB footprint, waves hold their slot between reads and stores (us): per launch, 1 / 4 streams\n", fp >> 20);
        for (int before = 0; before < 2; ++before)
            for (int hold_us : {0, 2, 4, 6, 8, 12}) {
                float t[2];
                for (int k = 0; k < 2; ++k) {
                    const int ns = k ? 4 : 1;
                    hipEvent_t a, b2; CK(hipEventCreate(&a)); CK(hipEventCreate(&b2));
                    auto launch = [&](int i) { k_shape<<<rows / 8, 512, 0, st[i % ns]>>>(ref, starts + (size_t)(i % T) * rows, slot_idx + (size_t)(i % T) * rows, slots, out[i & 3], L, rows, before ? 0 : hold_us * 100, before ? hold_us * 100 : 0); };
                    for (int i = 0; i < 64; ++i) launch(i);
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(a, st[0]));
                    for (int i = 0; i < 640; ++i) launch(i);
                    for (int i = 1; i < ns; ++i) { hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, st[i])); CK(hipStreamWaitEvent(st[0], ev, 0)); }
                    CK(hipEventRecord(b2, st[0])); CK(hipEventSynchronize(b2));
                    float ms; CK(hipEventElapsedTime(&ms, a, b2)); t[k] = ms / 640 * 1000.f;
                }
                printf("  hold %2d us %s the window reads: %7.2f us  %7.2f us\n", hold_us, before ? "BEFORE" : "after ", t[0], t[1]);
            }
    }
    return 0;
}
