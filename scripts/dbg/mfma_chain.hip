// Diagnostic: cycles per fp32 MFMA for dependent chains vs interleaved accumulators, one wave per SIMD (as in
// stream_fwd_kernel).  hipcc --offload-arch=gfx950 -O3 scripts/dbg/mfma_chain.hip -o scripts/dbg/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters, float a0, float b0) {
    float a = a0 + threadIdx.x, b = b0;
    f32x16 A0 = {0}, A1 = {0}, A2 = {0}, A3 = {0};
    f32x4 C0 = {0, 0, 0, 0}, C1 = C0, C2 = C0, C3 = C0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) { A0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A0, 0, 0, 0); }
            if (MODE == 1) { if (u & 1) A1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A1, 0, 0, 0);
                             else A0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A0, 0, 0, 0); }
            if (MODE == 2) { C0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, C0, 0, 0, 0); }
            if (MODE == 3) { if (u & 1) C1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, C1, 0, 0, 0);
                             else C0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, C0, 0, 0, 0); }
            if (MODE == 4) { switch (u & 3) {
                case 0: C0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, C0, 0, 0, 0); break;
                case 1: C1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, C1, 0, 0, 0); break;
                case 2: C2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, C2, 0, 0, 0); break;
                default: C3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, C3, 0, 0, 0); } }
            if (MODE == 5) { switch (u & 3) {
                case 0: A0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A0, 0, 0, 0); break;
                case 1: A1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A1, 0, 0, 0); break;
                case 2: A2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A2, 0, 0, 0); break;
                default: A3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, A3, 0, 0, 0); } }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += A0[r] + A1[r] + A2[r] + A3[r];
    for (int r = 0; r < 4; ++r) s += C0[r] + C1[r] + C2[r] + C3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 256>>>(out, cyc, 10, 1.f, 0.f);
    hipEventRecord(e0);
    k<MODE><<<256, 256>>>(out, cyc, iters, 1.f, 0.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %7.2f memtime ticks/MFMA   %7.2f ns/MFMA\n", name, (double)c / (iters * 16.0), ms * 1e6 / (iters * 16.0));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
    run<0>("32x32x2 one dependent chain", out, cyc);
    run<1>("32x32x2 two interleaved accumulators", out, cyc);
    run<5>("32x32x2 four interleaved accumulators", out, cyc);
    run<2>("16x16x4 one dependent chain", out, cyc);
    run<3>("16x16x4 two interleaved accumulators", out, cyc);
    run<4>("16x16x4 four interleaved accumulators", out, cyc);
    return 0;
}
