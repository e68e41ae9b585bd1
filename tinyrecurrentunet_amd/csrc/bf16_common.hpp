// Shared device helpers of the bf16 family (bf16.hip, bf16_convt.hip): bf16 pack / unpack, the 16-value butterfly reduction,
// the [octet][frame][8] LDS image and its transposed fragment reads.
#pragma once
#include "common.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
// round to nearest even, both halves in ONE v_cvt_pk_bf16_f32 (two scalar conversions compile to two of them plus an
// SDWA or: 12 vector instructions per octet instead of 4 in kernels whose prologue is bound by exactly those)
typedef float bf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf_pack(float lo, float hi) {
    const bf_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf_bf16x2));
}
// Prologue math on the register pairs of an octet (v_pk_fma_f32: two fp32 FMAs per instruction) and ReLU on the PACKED
// result (v_pk_max_i16: as signed 16-bit integers every negative bf16, -0 included, is below 0 and every positive one keeps
// its order -- the same bits as max(x, 0) before rounding): 16 vector instructions per octet instead of 28, in kernels
// whose B-operand prologue, not the matrix pipe, sets the pace.
typedef short bf_s16x2 __attribute__((ext_vector_type(2)));
// relu?(c0 x + c1) of the 8 values of octet x, packed
template <bool RELU>
__device__ __forceinline__ u32x4 bf_affine8(const u32x4 x, const float* c0, const float* c1) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf_f32x2 p = {bf_lo(x[i]), bf_hi(x[i])};
        const bf_f32x2 a = {c0[2 * i], c0[2 * i + 1]}, b = {c1[2 * i], c1[2 * i + 1]};
        p = __builtin_elementwise_fma(a, p, b);
        unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf_bf16x2));
        if (RELU) u = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(bf_s16x2, u), bf_s16x2{0, 0}));
        o[i] = u;
    }
    return o;
}
// ca x + (cb z + cc) (BatchNorm backward of an octet), packed
__device__ __forceinline__ u32x4 bf_bnbwd8(const u32x4 x, const u32x4 z, const float* ca, const float* cb, const float* cc) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf_f32x2 px = {bf_lo(x[i]), bf_hi(x[i])}, pz = {bf_lo(z[i]), bf_hi(z[i])};
        const bf_f32x2 a = {ca[2 * i], ca[2 * i + 1]}, b = {cb[2 * i], cb[2 * i + 1]}, c = {cc[2 * i], cc[2 * i + 1]};
        const bf_f32x2 p = __builtin_elementwise_fma(a, px, __builtin_elementwise_fma(b, pz, c));
        o[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf_bf16x2));
    }
    return o;
}
__device__ __forceinline__ void bf_unpack8(const u32x4 v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf_lo(v[i]); f[2 * i + 1] = bf_hi(v[i]); }
}
__device__ __forceinline__ u32x4 bf_pack8(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = bf_pack(f[2 * i], f[2 * i + 1]);
    return v;
}

// sum each of 16 per-lane values over the 32 lanes of a half-wave; every lane c returns the total of value
// butterfly16_index(c) (16 cross-lane exchanges instead of 80, and ONE live register instead of 16); lanes c and c ^ 16
// hold the same total: butterfly16_writer(c) picks one.  The 15 halving exchanges stay inside a row of 16 lanes and are
// DPP moves (partner = row mirror / half-row mirror / quad reverse / quad swap: it differs in the stage's bit and agrees in
// the bits already consumed, which is all a butterfly sum needs); the last one crosses rows with v_permlane16_swap.  As
// __shfl_xor they were 16 ds_bpermute round trips in five dependent stages, ~500 cycles of LDS latency per call.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float butterfly16(const float (&x)[16], int c) {
    float y8[8], y4[4], y2[2];
    const bool b8 = c & 8, b4 = c & 4, b2 = c & 2, b1 = c & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) y8[i] = (b8 ? x[i + 8] : x[i]) + dpp_mov<0x140>(b8 ? x[i] : x[i + 8]);          // row_mirror
#pragma unroll
    for (int i = 0; i < 4; ++i) y4[i] = (b4 ? y8[i + 4] : y8[i]) + dpp_mov<0x141>(b4 ? y8[i] : y8[i + 4]);      // row_half_mirror
#pragma unroll
    for (int i = 0; i < 2; ++i) y2[i] = (b2 ? y4[i + 2] : y4[i]) + dpp_mov<0x1B>(b2 ? y4[i] : y4[i + 2]);       // quad_perm 3,2,1,0
    const float y1 = (b1 ? y2[1] : y2[0]) + dpp_mov<0xB1>(b1 ? y2[0] : y2[1]);                                  // quad_perm 1,0,3,2
    // rows (c & 16): vdst's odd rows <-> vsrc's even rows, so the two results are (even row, even row) and (odd row, odd row)
    const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, y1), __builtin_bit_cast(unsigned, y1), false, false);
    return __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
}
__device__ __forceinline__ int butterfly16_index(int c) { return c & 15; }
__device__ __forceinline__ bool butterfly16_writer(int c) { return !(c & 16); }

constexpr int BW_F = 64;                         // frames per step
constexpr int BW_OS = BW_F * 16 + 64;            // LDS bytes between octets

__device__ __forceinline__ u32x2 lds_tr16(const unsigned char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(u32x2, v);
}
// fragment (rows = 32 channels starting at octet oct0, k = 16 frames starting at f0) of an [octet][frame][8] image
__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* img, int oct0, int f0, int lane) {
    const int G = lane >> 4, i = lane & 15;
    const unsigned char* p = img + (oct0 + 2 * (G & 1) + ((i & 3) >> 1)) * BW_OS + (f0 + 8 * (G >> 1) + (i >> 2)) * 16 + 8 * (i & 1);
    const u32x2 lo = lds_tr16(p), hi = lds_tr16(p + 64);
    const u32x4 f = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, f);
}

}  // namespace
