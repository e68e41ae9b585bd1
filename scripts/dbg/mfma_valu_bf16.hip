// Diagnostic: cost of K vector-ALU instructions between two 32x32x2 MFMAs, 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int K, int G>      // K VALU ops after every G-th MFMA
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int iters, float a0) {
    float a = a0 + threadIdx.x, b = 1.f;
    bf16x8 av, bv;
    for (int e = 0; e < 8; ++e) { av[e] = (__bf16)(a + e); bv[e] = (__bf16)1.f; }
    f32x16 A0 = {0};
    float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, A0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (u % G == 0) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    switch (j & 3) {
                        case 0: asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(b)); break;
                        case 1: asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(b)); break;
                        case 2: asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(b)); break;
                        default: asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x3) : "v"(b)); break;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = x0 + x1 + x2 + x3;
    for (int r = 0; r < 16; ++r) s += A0[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int K, int G>
void run(float* out, long long* cyc) {
    const int iters = 1000;
    for (int threads : {256, 512}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<K, G><<<256, threads>>>(out, cyc, 10, 1.f);
        hipEventRecord(e0);
        k<K, G><<<256, threads>>>(out, cyc, iters, 1.f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%2d VALU ops after every %2d MFMA(s) = %5.2f per MFMA, %d waves/SIMD: wave 0 %7.2f ticks per own MFMA; kernel %7.2f ns per MFMA of the SIMD\n", K, G,
               (double)K / G, threads / 256, (double)c / (iters * 16.0), ms * 1e6 / (iters * 16.0 * (threads / 256)));
    }
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    run<0, 1>(out, cyc); run<1, 1>(out, cyc); run<2, 1>(out, cyc); run<4, 1>(out, cyc); run<8, 1>(out, cyc); run<16, 1>(out, cyc);
    run<4, 4>(out, cyc); run<16, 4>(out, cyc); run<32, 8>(out, cyc); run<64, 16>(out, cyc);
    return 0;
}
