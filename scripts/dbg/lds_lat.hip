// Diagnostic: ds_read_b32 latency (dependent chain) and throughput (8 independent reads in flight, 4 waves) at LDS
// byte offsets below and above 64 KiB of a 160 KiB allocation, for the two B-operand access patterns of stream_fwd.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int base_f, int pattern, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 40960; i += 256) lds[i] = 0.f;      // zeros: the chain stays at its own address
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // pattern 0: 32 consecutive floats of row h (32x32 B operand); pattern 1: 16 consecutive floats of row q (16x16)
    const int off = pattern == 0 ? (lane >> 5) * 144 + (lane & 31) : (lane >> 4) * 144 + (lane & 15);
    const float* p = lds + base_f + off;
    long long t0 = __builtin_amdgcn_s_memtime();
    float v = 0.f;
    for (int i = 0; i < iters; ++i) {
        v = p[(int)v];                                    // dependent: latency
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    for (int i = 0; i < iters; ++i) {
        const float* q = p + (i & 3) * 576;
        s0 += q[0]; s1 += q[16]; s2 += q[576 * 4]; s3 += q[576 * 4 + 16]; s4 += q[1152 * 4]; s5 += q[1152 * 4 + 16];
        s6 += q[1728 * 4]; s7 += q[1728 * 4 + 16];
    }
    long long t2 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = v + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    const int iters = 4000;
    for (int pattern = 0; pattern < 2; ++pattern)
        for (int base : {0, 9216, 18432, 27648}) {
            k<<<256, 256, 163840>>>(out, cyc, base, pattern, iters);
            hipDeviceSynchronize();
            long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
            printf("pattern %d base %6d B: latency %6.1f cycles/read   8-deep stream %6.1f cycles/read (per wave, 4 waves)\n", pattern,
                   base * 4, (double)c[0] / iters, (double)c[1] / (iters * 8.0));
        }
    return 0;
}
