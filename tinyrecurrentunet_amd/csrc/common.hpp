// Shared device/host helpers for libtrunet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "trunet_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define TRUNET_NUM_CU 256

// cache policy of the LDS-DMA operand streams (aux operand of global_load_lds): 0 default, 2 = nt.  A/B build switch.
#ifndef TRUNET_DMA_AUX
#define TRUNET_DMA_AUX 0
#endif

static inline int trunet_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRUNET_OK : TRUNET_ELAUNCH;
}

// wgrad_small.hip: vector-ALU weight gradient of the thin layers; TRUNET_ENOTSUP when the shape is not thin
int trunet_launch_wgrad_small(const trunet_wgrad_args* h, hipStream_t st);
// wgrad_small.hip: fused backward of a pointwise layer with <= 8 output rows (same contract as trunet_pw_bwd)
int trunet_launch_pw_bwd_small(const trunet_pwbwd_args* H, hipStream_t st);

// sum over the 32 lanes that share (lane >> 5)
__device__ __forceinline__ float half_wave_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32);
    return half_wave_sum(v);
}

// block-wide sum for 256-thread blocks; result valid in thread 0 (and broadcast through smem[0])
__device__ __forceinline__ double block_sum_f64(double v, double* smem /* >= 256 */) {
    int t = threadIdx.x;
    smem[t] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) smem[t] += smem[t + s];
        __syncthreads();
    }
    double r = smem[0];
    __syncthreads();
    return r;
}
