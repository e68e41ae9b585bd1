// Input pipeline on the GPU: the reference augments the noise in CPU DataLoader workers
// (dataset.py:79-126: gain -> low-pass biquad -> high-pass biquad through torchaudio) and mixes it with the clean crop
// (dataset.py:380).  Here the whole batch is augmented and mixed by ONE launch on the signals already in HBM:
//   noise' = clamp(hp(clamp(lp(gain * noise)), -1, 1), -1, 1);   noisy = clean + noise'
// (torchaudio.functional.lowpass_biquad / highpass_biquad end in lfilter(clamp=True)).
//
// A biquad is a linear recurrence y[n] = b0 x[n] + b1 x[n-1] + b2 x[n-2] - a1 y[n-1] - a2 y[n-2]: one workgroup per
// signal, 256 threads each own a contiguous chunk and (1) run it from a zero output state, (2) the chunk-to-chunk
// carry s' = A^len s + f is resolved by a 256-step scan, (3) every thread reruns its chunk from its true state.
// Latency-bound at these sizes (64 signals x 96k samples = 24 MB): three passes over L1/L2-resident rows.
#include "common.hpp"

namespace {

struct Biquad { float b0, b1, b2, a1, a2; };

// one stage: out[n] = post(clamp(biquad(g * in[n]))) with post = (+ add[n]) when add != NULL; in may alias out
__device__ void biquad_stage(const float* in, float* out, const float* __restrict__ add, int L, float g, Biquad q,
                             float (*sm)[6]) {
    const int t = threadIdx.x;
    const int CH = (L + 255) / 256;
    const int n0 = min(t * CH, L), n1 = min(n0 + CH, L);
    // inputs just before the chunk (owned by the previous thread, which rewrites them in pass 3 when in == out)
    const float xm1 = n0 >= 1 ? g * in[n0 - 1] : 0.f;
    const float xm2 = n0 >= 2 ? g * in[n0 - 2] : 0.f;
    // pass 1: zero-state response and the homogeneous transition A^len (columns from unit states)
    float y1 = 0.f, y2 = 0.f, x1 = xm1, x2 = xm2;
    float u1 = 1.f, u2 = 0.f, v1 = 0.f, v2 = 1.f;
    for (int n = n0; n < n1; ++n) {
        const float x = g * in[n];
        const float y = fmaf(q.b0, x, fmaf(q.b1, x1, fmaf(q.b2, x2, -fmaf(q.a1, y1, q.a2 * y2))));
        x2 = x1; x1 = x; y2 = y1; y1 = y;
        const float u = -fmaf(q.a1, u1, q.a2 * u2);
        u2 = u1; u1 = u;
        const float v = -fmaf(q.a1, v1, q.a2 * v2);
        v2 = v1; v1 = v;
    }
    // (y[n-1], y[n-2]) after the chunk = M (s1, s2) + f with M = [[u1, v1], [u2, v2]]
    sm[t][0] = u1; sm[t][1] = v1; sm[t][2] = u2; sm[t][3] = v2; sm[t][4] = y1; sm[t][5] = y2;
    __syncthreads();
    if (t == 0) {
        float s1 = 0.f, s2 = 0.f;
        for (int c = 0; c < 256; ++c) {
            const float m00 = sm[c][0], m01 = sm[c][1], m10 = sm[c][2], m11 = sm[c][3], f1 = sm[c][4], f2 = sm[c][5];
            sm[c][4] = s1; sm[c][5] = s2;                 // state entering chunk c
            const float r1 = fmaf(m00, s1, fmaf(m01, s2, f1));
            const float r2 = fmaf(m10, s1, fmaf(m11, s2, f2));
            s1 = r1; s2 = r2;
        }
    }
    __syncthreads();
    // pass 3: rerun from the true state, clamp, mix, store
    y1 = sm[t][4]; y2 = sm[t][5]; x1 = xm1; x2 = xm2;
    for (int n = n0; n < n1; ++n) {
        const float x = g * in[n];
        const float y = fmaf(q.b0, x, fmaf(q.b1, x1, fmaf(q.b2, x2, -fmaf(q.a1, y1, q.a2 * y2))));
        x2 = x1; x1 = x; y2 = y1; y1 = y;
        float o = fminf(fmaxf(y, -1.f), 1.f);
        if (add) o += add[n];
        out[n] = o;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void augment_mix_kernel(const float* __restrict__ noise, const float* __restrict__ clean,
                                                          const float* __restrict__ params, float* noisy,
                                                          float* __restrict__ noise_out, int L) {
    __shared__ float sm[256][6];
    const int b = blockIdx.x;
    const float* p = params + (size_t)b * 11;
    const Biquad lp = {p[1], p[2], p[3], p[4], p[5]};
    const Biquad hp = {p[6], p[7], p[8], p[9], p[10]};
    const float* x = noise + (size_t)b * L;
    float* y = noisy + (size_t)b * L;
    float* aug = noise_out ? noise_out + (size_t)b * L : y;
    biquad_stage(x, aug, nullptr, L, p[0], lp, sm);                   // gain + low-pass (+clamp)
    if (noise_out) {
        biquad_stage(aug, aug, nullptr, L, 1.f, hp, sm);              // high-pass (+clamp): the augmented noise itself
        for (int n = threadIdx.x; n < L; n += 256) y[n] = (clean ? clean[(size_t)b * L + n] : 0.f) + aug[n];
    } else {
        biquad_stage(aug, y, clean ? clean + (size_t)b * L : nullptr, L, 1.f, hp, sm);
    }
}

}  // namespace

extern "C" int trunet_augment_mix(const float* noise, const float* clean, const float* params, float* noisy,
                                  float* noise_out, int B, int L, void* stream) {
    if (!noise || !params || !noisy || B <= 0 || L <= 0) return TRUNET_EINVAL;
    if (noisy == noise || noise_out == noise || (noise_out && noise_out == noisy)) return TRUNET_EINVAL;
    hipLaunchKernelGGL(augment_mix_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, noise, clean, params, noisy,
                       noise_out, L);
    return trunet_launch_status();
}
