// Three-term bf16 split of fp32 MFMA operands (gemm_x3.hip's header explains the arithmetic): x = hi + mid + lo, every term
// rounded to nearest, the residues exact in fp32; a * b ~ a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0), the dropped
// terms are below 2^-24 |a b|.  Shared by the fused backward kernels (convt_bwd_x3.hip, pw_bwd.hip), where a lane splits
// the 8 consecutive K-values of its own fragment.
#pragma once
#include "bf16_common.hpp"

namespace {

__device__ __forceinline__ void ctx_split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = bf_pack(x0, x1);
    const float r0 = x0 - bf_lo(hi), r1 = x1 - bf_hi(hi);
    mid = bf_pack(r0, r1);
    lo = bf_pack(r0 - bf_lo(mid), r1 - bf_hi(mid));
}
__device__ __forceinline__ void ctx_split8(const f32x4 v0, const f32x4 v1, u32x4& q0, u32x4& q1, u32x4& q2) {
    unsigned a, b, c;
    ctx_split2(v0[0], v0[1], a, b, c); q0[0] = a; q1[0] = b; q2[0] = c;
    ctx_split2(v0[2], v0[3], a, b, c); q0[1] = a; q1[1] = b; q2[1] = c;
    ctx_split2(v1[0], v1[1], a, b, c); q0[2] = a; q1[2] = b; q2[2] = c;
    ctx_split2(v1[2], v1[3], a, b, c); q0[3] = a; q1[3] = b; q2[3] = c;
}
}  // namespace

#define CTX_MF(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc_, 0, 0, 0)
// six products of a three-term pair, the small cross terms first
#define CTX_MF6(acc_, a0_, a1_, a2_, b0_, b1_, b2_) do { CTX_MF(acc_, a2_, b0_); CTX_MF(acc_, a0_, b2_); CTX_MF(acc_, a1_, b1_); CTX_MF(acc_, a1_, b0_); CTX_MF(acc_, a0_, b1_); CTX_MF(acc_, a0_, b0_); } while (0)
