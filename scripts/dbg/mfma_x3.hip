// Diagnostic: is a three-term bf16 split of fp32 operands on the bf16 MFMA a faster fp32-grade GEMM
// inner loop than the fp32 MFMA on this chip?
//
//   x = b0 + b1 + b2 (8 + 8 + 8 significand bits, each term rounded to nearest), products b0b0, b0b1, b1b0, b0b2, b1b1, b2b0
//   on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 6 x 33 cycles per (32 x 32 x 16) against 8 x 64 for
//   v_mfma_f32_32x32x2_f32.
//
// What is measured is the MFMA phase of a pointwise-conv tile exactly as conv_gemm_kernel runs it: a K-chunk of the
// activated operand B sits in LDS as fp32 [32 rows][256 frames] (landed by LDS-DMA, prologue applied in place); 8 waves
// (two per SIMD) multiply it with a 128-row weight block.  Variants:
//   F32    fp32 fragments of A in LDS, wave = (row slice, 4 column tiles): 64 MFMAs(32x32x2) per wave and chunk
//   X3W    the split done by the CONSUMING wave on its B fragment (wave = all 4 row slices x one column tile): per k-step
//          8 ds_read_b32 + 36 vector instructions + 12 ds_read_b128 (A planes) + 24 MFMAs(32x32x16)
//   X3P    B already split into three bf16 planes in fragment order (as a staging pass would leave it): per k-step
//          3 + 12 ds_read_b128 + 24 MFMAs -- the upper bound
// plus the numerical error of each against a float64 product of the same fp32 operands.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int M = 128, KC = 32, FT = 256;      // rows of the weight block, K rows per chunk, frames per tile

// three-way split of two floats into bf16 terms rounded to nearest (the form gemm_x3.hip uses), packed: element 0 in the
// low half.  -DX3_TRUNC: the truncating form (and / sub: exact, but biased towards zero in the dropped cross terms)
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk(float a, float b) {
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
#ifdef X3_TRUNC
    const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    hi = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(b0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(b1 & 0xFFFF0000u);
    const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    mid = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(c0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(c1 & 0xFFFF0000u);
    lo = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
#else
    hi = pk(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xFFFF0000u);
    mid = pk(r0, r1);
    lo = pk(r0 - __uint_as_float(mid << 16), r1 - __uint_as_float(mid & 0xFFFF0000u));
#endif
}

// LDS images (floats): A32 [4 rs][16 kp][64 lanes] (k-pair fragments of v_mfma_f32_32x32x2_f32), B32 [32][256],
// A3 [3 planes][4 rt][2 ks][64 lanes] x 16 B, B3 [3 planes][2 ks][8 col tiles][64 lanes] x 16 B
template <int V>
__global__ __launch_bounds__(512, 1) void kern(const float* __restrict__ Ag, const float* __restrict__ Bg, float* __restrict__ Cg,
                                               long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    float* B32 = (float*)sm;                                   // 32 KB
    float* A32 = B32 + KC * FT;                                // 16 KB
    u32x4* A3 = (u32x4*)(A32 + M * KC);                        // 3 * 4 * 2 * 64 * 16 B = 24 KB
    u32x4* B3 = A3 + 3 * 4 * 2 * 64;                           // 3 * 2 * 8 * 64 * 16 B = 48 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    // ---- stage: B32 as is; A32 in k-pair fragment order; the split images
    for (int i = tid; i < KC * FT; i += 512) B32[i] = Bg[i];
    for (int i = tid; i < 4 * 16 * 64; i += 512) {
        const int ln = i & 63, kp = (i >> 6) & 15, rs = i >> 10;
        A32[i] = Ag[(32 * rs + (ln & 31)) * KC + 2 * kp + (ln >> 5)];
    }
    for (int i = tid; i < 4 * 2 * 64; i += 512) {              // A planes: lane (row = ln & 31, k = 16 ks + 8 (ln >> 5) + j)
        const int ln = i & 63, ks = (i >> 6) & 1, rt = i >> 7;
        unsigned hi[4], mid[4], lo[4];
        for (int j = 0; j < 4; ++j) {
            const float* p = Ag + (32 * rt + (ln & 31)) * KC + 16 * ks + 8 * (ln >> 5) + 2 * j;
            split2(p[0], p[1], hi[j], mid[j], lo[j]);
        }
        A3[(0 * 4 + rt) * 128 + ks * 64 + ln] = u32x4{hi[0], hi[1], hi[2], hi[3]};
        A3[(1 * 4 + rt) * 128 + ks * 64 + ln] = u32x4{mid[0], mid[1], mid[2], mid[3]};
        A3[(2 * 4 + rt) * 128 + ks * 64 + ln] = u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
    for (int i = tid; i < 2 * 8 * 64; i += 512) {              // B planes: lane (col = 32 ct + (ln & 31), k = 16 ks + 8 (ln >> 5) + j)
        const int ln = i & 63, ct = (i >> 6) & 7, ks = i >> 9;
        unsigned hi[4], mid[4], lo[4];
        for (int j = 0; j < 4; ++j) {
            const int k0 = 16 * ks + 8 * (ln >> 5) + 2 * j, col = 32 * ct + (ln & 31);
            split2(Bg[k0 * FT + col], Bg[(k0 + 1) * FT + col], hi[j], mid[j], lo[j]);
        }
        B3[((0 * 2 + ks) * 8 + ct) * 64 + ln] = u32x4{hi[0], hi[1], hi[2], hi[3]};
        B3[((1 * 2 + ks) * 8 + ct) * 64 + ln] = u32x4{mid[0], mid[1], mid[2], mid[3]};
        B3[((2 * 2 + ks) * 8 + ct) * 64 + ln] = u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) {
            // wave = (row slice rs, column group of 4 tiles): conv_gemm_kernel<4, 32, false, 0, 8>
            const int rs = wave & 3, half = wave >> 2;
            const float* Ab = A32 + rs * 1024;
            const float* Bb = B32 + 128 * half + 4 * c;
            f32x4 af[4], bf[16];
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) af[kg] = f32x4{Ab[(4 * kg + 0) * 64 + lane], Ab[(4 * kg + 1) * 64 + lane],
                                                          Ab[(4 * kg + 2) * 64 + lane], Ab[(4 * kg + 3) * 64 + lane]};
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) bf[kk] = *(const f32x4*)(Bb + (2 * kk + h) * FT);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk >> 2][kk & 3], bf[kk][t], acc[t], 0, 0, 0);
        } else {
            // wave = all four row slices x column tile `wave`
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 bp[3];
                if constexpr (V == 1) {
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = B32[(16 * ks + 8 * h + j) * FT + 32 * wave + c];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned u0, u1, u2;
                        split2(x[2 * j], x[2 * j + 1], u0, u1, u2);
                        bp[0][j] = u0; bp[1][j] = u1; bp[2][j] = u2;
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < 3; ++p) bp[p] = B3[((p * 2 + ks) * 8 + wave) * 64 + lane];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u32x4 a0 = A3[(0 * 4 + t) * 128 + ks * 64 + lane], a1 = A3[(1 * 4 + t) * 128 + ks * 64 + lane],
                                a2 = A3[(2 * 4 + t) * 128 + ks * 64 + lane];
#define MF(a_, b_) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc[t], 0, 0, 0)
                    // small terms first: they are not rounded away against the large partial sum
#ifdef X3_BIGFIRST
                    MF(a0, bp[0]); MF(a0, bp[1]); MF(a1, bp[0]); MF(a1, bp[1]); MF(a0, bp[2]); MF(a2, bp[0]);
#else
                    MF(a2, bp[0]); MF(a0, bp[2]); MF(a1, bp[1]); MF(a1, bp[0]); MF(a0, bp[1]); MF(a0, bp[0]);
#endif
#undef MF
                }
            }
        }
        asm volatile("" ::: "memory");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    // ---- result of ONE chunk pass (iters = 1 for the accuracy run): C [128][256]
    if (iters == 1 && blockIdx.x == 0) {
        for (int t = 0; t < 4; ++t)
            for (int r = 0; r < 16; ++r) {
                int row, col;
                if (V == 0) { row = 32 * (wave & 3) + (r & 3) + 8 * (r >> 2) + 4 * h; col = 128 * (wave >> 2) + 4 * c + t; }
                else { row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; col = 32 * wave + c; }
                Cg[row * FT + col] = acc[t][r];
            }
    } else if (blockIdx.x == 0 && tid == 0) {
        float s = 0.f;
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        Cg[M * FT] = s;
    }
}

template <int V>
void run(const char* name, const float* dA, const float* dB, float* dC, long long* cyc, const std::vector<double>& ref) {
    const size_t lds = 32768 + 16384 + 24576 + 49152;
    hipFuncSetAttribute((const void*)kern<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kern<V><<<1, 512, lds>>>(dA, dB, dC, cyc, 1);
    std::vector<float> C(M * FT);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0, rmax = 0, e2 = 0, r2 = 0, esum = 0, rabs = 0;
    for (size_t i = 0; i < C.size(); ++i) {
        const double d = C[i] - ref[i];
        emax = std::fmax(emax, std::fabs(d)); rmax = std::fmax(rmax, std::fabs(ref[i]));
        e2 += d * d; r2 += ref[i] * ref[i];
        esum += d * (ref[i] >= 0 ? 1.0 : -1.0);        // signed towards / away from zero
        rabs += std::fabs(ref[i]);
    }
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<V><<<256, 512, lds>>>(dA, dB, dC, cyc, 10);
    hipEventRecord(e0);
    kern<V><<<256, 512, lds>>>(dA, dB, dC, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cy; hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost);
    const double flop = 2.0 * M * KC * FT * iters * 256;
    printf("%-4s  %8.1f memtime ticks per chunk (wave 0)   %7.3f us per chunk per CU   %7.1f TFLOP/s-equivalent   "
           "error vs float64: max %.2e of max|C|, relative L2 %.2e, BIAS (mean signed error away from zero / mean |C|) %+.2e\n", name, (double)cy / iters, ms * 1e3 / iters,
           flop / (ms * 1e-3) / 1e12, emax / rmax, std::sqrt(e2 / r2), esum / rabs);
}

int main() {
    std::vector<float> A(M * KC), B(KC * FT);
    srand(7);
    auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0) * (1.f + 0.001f * (rand() & 1023)); };
    const bool pos = getenv("X3_POS") != nullptr;                   // all-positive operands: a rounding bias cannot cancel
    for (auto& v : A) v = pos ? std::fabs(rnd()) : rnd();
    for (auto& v : B) v = pos ? std::fabs(rnd()) : (rnd() > 0.f ? std::fabs(rnd()) : 0.f);   // post-ReLU-like operand
    std::vector<double> ref(M * FT, 0.0);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < KC; ++k)
            for (int n = 0; n < FT; ++n) ref[m * FT + n] += (double)A[m * KC + k] * (double)B[k * FT + n];
    float *dA, *dB, *dC; long long* cyc;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (M * FT + 64) * 4); hipMalloc(&cyc, 64);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    printf("MFMA phase of one K-chunk (32 rows x 256 frames) against a 128-row weight block, 8 waves, one workgroup per CU, 256 CUs\n");
    run<0>("F32", dA, dB, dC, cyc, ref);
    run<1>("X3W", dA, dB, dC, cyc, ref);
    run<2>("X3P", dA, dB, dC, cyc, ref);
    return 0;
}
