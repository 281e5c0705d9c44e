// tools/satomic_test.hip: does gfx950 execute scalar memory atomics with return (s_atomic_add ... glc, tracked by lgkmcnt)?
// Every wave claims 3 tickets; the host checks that 3 * waves distinct tickets 0 .. 3 * waves - 1 were handed out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned *ctr, unsigned *out) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int i = 0; i < 3; ++i) {
        unsigned v = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
        if ((threadIdx.x & 63) == 0) out[wave * 3 + i] = v;
    }
}
int main() {
    const int waves = 8192 * 4;
    unsigned *ctr, *out;
    hipMalloc(&ctr, 4); hipMalloc(&out, waves * 3 * 4);
    hipMemset(ctr, 0, 4);
    k<<<waves / 4, 256>>>(ctr, out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<unsigned> h(waves * 3);
    unsigned c;
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = c == (unsigned)waves * 3;
    for (size_t i = 0; i < h.size(); ++i) ok = ok && h[i] == i;
    printf("counter %u (expected %d), tickets %s\n", c, waves * 3, ok ? "distinct and complete" : "WRONG");
    return ok ? 0 : 2;
}
