// Diagnostic: what an instruction between two MFMAs costs (one wave per SIMD, as in stream_fwd_kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE: 0 none, 1 v_add_u32 (literal), 2 ds_read_b32 imm offset (pipelined 8 deep), 3 v_add + ds_read (address from the add)
template <int MODE, bool BIG>
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int iters, float a0, int zero) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) lds[i] = 1.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float a = a0 + threadIdx.x;
    f32x16 A0 = {0};
    f32x4 C0 = {0, 0, 0, 0}, C1 = C0;
    const float* p = lds + (lane >> 4) * 144 + (lane & 15) + zero;
    int va = lane * 4 + zero;
    float b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = 1.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float bb = b[u & 7];
            if (MODE == 1) { asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(va)); }
            if (MODE == 2) { b[u & 7] = p[u * 576 + (i & 1) * 16]; }
            if (MODE == 3) {
                int ad;
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(ad) : "s"(0x10000 + u * 2304), "v"(va));
                float r;
                asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(ad) : "memory");
                b[u & 7] = r;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (BIG) A0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, A0, 0, 0, 0);
            else if (u & 1) C1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, C1, 0, 0, 0);
            else C0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, C0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = va;
    for (int r = 0; r < 16; ++r) s += A0[r];
    for (int r = 0; r < 4; ++r) s += C0[r] + C1[r];
    for (int u = 0; u < 8; ++u) s += b[u];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, bool BIG>
void run(const char* name, float* out, long long* cyc, int threads = 256) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k<MODE, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    k<MODE, BIG><<<256, threads, 163840>>>(out, cyc, 10, 1.f, 0);
    k<MODE, BIG><<<256, threads, 163840>>>(out, cyc, iters, 1.f, 0);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-8s %d waves/SIMD  %-46s %7.2f cycles per MFMA of the SIMD\n", BIG ? "32x32x2" : "16x16x4", threads / 256, name,
           (double)c / (iters * 16.0 * (threads / 256)));
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    run<0, true>("MFMAs only", out, cyc);
    run<1, true>("+ v_add_u32 (literal) per MFMA", out, cyc);
    run<2, true>("+ ds_read_b32 imm offset per MFMA", out, cyc);
    run<3, true>("+ v_add_u32 -> ds_read_b32 per MFMA", out, cyc);
    run<0, false>("MFMAs only", out, cyc);
    run<1, false>("+ v_add_u32 (literal) per MFMA", out, cyc);
    run<2, false>("+ ds_read_b32 imm offset per MFMA", out, cyc);
    run<3, false>("+ v_add_u32 -> ds_read_b32 per MFMA", out, cyc);
    run<0, true>("MFMAs only", out, cyc, 512);
    run<1, true>("+ v_add_u32 (literal) per MFMA", out, cyc, 512);
    run<2, true>("+ ds_read_b32 imm offset per MFMA", out, cyc, 512);
    run<3, true>("+ v_add_u32 -> ds_read_b32 per MFMA", out, cyc, 512);
    return 0;
}
