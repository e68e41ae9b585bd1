// Fused backward of a ConvTranspose1d(64 -> 64, k = K, stride S, padding S/2) + BatchNorm layer (autograd of
// /root/reference/network.py:67,86 with the BatchNorm of :72,91 behind it and the BatchNorm+ReLU of :65-66,84-85 in front):
//   z[co][p][n] = b[co] + sum_{ci,k} W[ci][co][k] a[ci][q][n],  p = q S - pad + k,   a = max(sc zs + sh, 0)
// ONE pass over (dy, z, zs) produces
//   dz        = ca dy + cb z + cc                                   (BatchNorm backward of z: in LDS only)
//   dW, db    = sum_{q,n} a[ci][q] dz[co][q S - pad + k] ,  sum_{p,n} dz          (per-workgroup partial images)
//   g         = [sc zs + sh > 0] sum_{co,k} W[ci][co][k] dz[co][q S - pad + k]   -> gradient at the source's BatchNorm output
//   stats     = sum g, sum g (zs - mean)                            (BatchNorm backward of the source)
// The separate launches it replaces (conv_gemm data gradient over K tap segments + conv_wgrad over K tap segments) each
// read dy and z (the data gradient reads every dz row K/S times as a tap of different positions) and the source twice.
//
// Decomposition: a tile is (source position q, 32 frames).  A workgroup owns whole frame chunks and walks q = 0 .. Lin-1,
// so the dz rows p = q S - pad .. q S - pad + K - 1 form a sliding window: every dz row is loaded and BatchNorm-transformed
// ONCE per chunk into a ring of K + 2 S row sets and serves all K taps (positions) that need it; every step brings S new dz
// rows (dy into the ring, z into a staging slot, combined in place by the thread that requested the piece) and one source
// row set.  LDS-DMA (global_load_lds) two steps ahead, counted vmcnt, one barrier per step, 16-byte pieces XOR-swizzled
// through the source address as in conv_wgrad_kernel / pw_bwd_kernel.
//
// All 8 waves carry the same MFMA load (16 K per step, two waves per SIMD):
//   waves 0-3  loaders + weight gradient: wave (ci tile, co tile) keeps its K accumulators (one per tap) for the whole
//              kernel; the frame axis is the MFMA K axis; BatchNorm+ReLU of the source is applied to the A fragment on the fly
//   waves 4-7  data gradient: wave (ci tile, co half) keeps W^T fragments of its 32 co rows for all taps in registers;
//              the two co halves of a ci tile are combined through LDS, and the co-half-0 wave runs the epilogue (ReLU mask,
//              store, statistics) from registers after the barrier, in the shadow of the next step.
#include "common.hpp"

namespace {

constexpr int CT_F = 32;            // frames per tile
constexpr int CT_C = 64;            // channels (both sides)
constexpr int CT_SET = CT_C * CT_F; // floats per row set (8 KB)
constexpr int CT_GRID = TRUNET_NUM_CU;

typedef __attribute__((address_space(3))) void* ct_lds_ptr_t;

__device__ __forceinline__ void ct_wait_vmcnt(int n) {
    switch (n) {
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        W_(0) W_(1) W_(2) W_(3) W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15)
        W_(16) W_(17) W_(18) W_(19) W_(20) W_(21) W_(22) W_(23) W_(24) W_(25) W_(26) W_(27) W_(28) W_(29) W_(30) W_(31)
#undef W_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// swizzled float offset of 16-byte piece `pc` (0..7) of row `r` inside a row set
__device__ __forceinline__ int ct_off(int r, int pc) { return r * CT_F + 4 * (pc ^ ((r >> 1) & 7)); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ct_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void ct_bstore(__amdgpu_buffer_rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, 0);
}
__device__ __forceinline__ int ct_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int K, int S>
__global__ __launch_bounds__(512, 2) void convt_bwd_kernel(const trunet_convt_bwd_args a) {
    constexpr int PAD = S / 2;
    constexpr int RDZ = K + 2 * S;      // dz ring: window + the rows of the next two steps
    constexpr int ZS = 2 * S;           // z staging slots
    constexpr int RS = 4;               // source ring: q (+ epilogue of q - 1), q + 1, q + 2
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* DZ = smem;                               // [RDZ][64][32]
    float* ZST = DZ + RDZ * CT_SET;                 // [ZS][64][32]
    float* SRC = ZST + ZS * CT_SET;                 // [RS][64][32]
    float* PART = SRC + RS * CT_SET;                // [2 parity][2 ci tiles][16][64]
    f32x4* CA = (f32x4*)(PART + 2 * 2 * 16 * 64);   // [64] (ca, cb, cc, 0) of dz
    f32x4* CB = CA + CT_C;                          // [64] (sc, sh, mean, 0) of the source

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int NP = a.NP, Lin = a.Lin, Lout = a.Lout;

    for (int r = tid; r < CT_C; r += 512) {
        CA[r] = f32x4{a.ca[r], a.cb[r], a.cc[r], 0.f};
        CB[r] = f32x4{a.s_scale[r], a.s_shift[r], a.s_mean[r], 0.f};
    }
    const int nfc = NP / CT_F;
    const int c_begin = (int)(((long long)blockIdx.x * nfc) / gridDim.x);
    const int c_end = (int)(((long long)(blockIdx.x + 1) * nfc) / gridDim.x);
    __syncthreads();

    if (wave < 4) {
        // =============================================================== loaders + weight gradient
        const int cit = wave & 1, cot = wave >> 1;
        f32x16 acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        float bsum[2] = {0.f, 0.f};
        const int pc = lane & 7;
        // this lane's rows in a row set: groups g = wave, wave + 4 -> row 8 g + (lane >> 3); logical piece folded in
        int row_[2], lc_[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            row_[i] = 8 * (wave + 4 * i) + (lane >> 3);
            lc_[i] = pc ^ ((row_[i] >> 1) & 7);
        }
        const float sc_a = a.s_scale[cit * 32 + c], sh_a = a.s_shift[cit * 32 + c];      // BN+ReLU of this lane's A row

        // DMA of dz rows [pa, pb) (dy -> ring slot p % RDZ, z -> staging slot p % ZS) and, with sq >= 0, of source row set
        // sq; returns the number of wave-instructions issued (rows outside [0, Lout) / [0, Lin) are skipped)
        auto issue = [&](int pa, int pb, int sq, int n0) __attribute__((always_inline)) -> int {
            int n = 0;
            for (int p = max(pa, 0); p < min(pb, Lout); ++p) {
                float* dst = DZ + (p % RDZ) * CT_SET;
                float* zst = ZST + (p % ZS) * CT_SET;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const size_t off = ((size_t)row_[i] * Lout + p) * NP + n0 + 4 * lc_[i];
                    __builtin_amdgcn_global_load_lds(a.dy + off, (ct_lds_ptr_t)(dst + (wave + 4 * i) * 256), 16, 0, TRUNET_DMA_AUX);
                    __builtin_amdgcn_global_load_lds(a.z + off, (ct_lds_ptr_t)(zst + (wave + 4 * i) * 256), 16, 0, TRUNET_DMA_AUX);
                }
                n += 4;
            }
            if (sq >= 0 && sq < Lin) {
                float* sdst = SRC + (sq % RS) * CT_SET;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const size_t off = ((size_t)row_[i] * Lin + sq) * NP + n0 + 4 * lc_[i];
                    __builtin_amdgcn_global_load_lds(a.src + off, (ct_lds_ptr_t)(sdst + (wave + 4 * i) * 256), 16, 0, TRUNET_DMA_AUX);
                }
                n += 2;
            }
            return n;
        };
        // BatchNorm backward of dz rows [pa, pb) in place (this thread's own DMA pieces) + bias-gradient partial sums
        auto prologue = [&](int pa, int pb, int n0) __attribute__((always_inline)) {
            for (int p = max(pa, 0); p < min(pb, Lout); ++p) {
                float* dst = DZ + (p % RDZ) * CT_SET;
                const float* zst = ZST + (p % ZS) * CT_SET;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int o = row_[i] * CT_F + 4 * pc;
                    f32x4 v = *(f32x4*)(dst + o);
                    const f32x4 zz = *(const f32x4*)(zst + o);
                    const f32x4 k = CA[row_[i]];
                    float s = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = fmaf(k[0], v[e], fmaf(k[1], zz[e], k[2]));
                        if (n0 + 4 * lc_[i] + e >= a.N) v[e] = 0.f;
                        s += v[e];
                    }
                    bsum[i] += s;
                    *(f32x4*)(dst + o) = v;
                }
            }
        };
        // dz rows first needed at step q >= 1: the S highest rows of its window
        auto new_lo = [&](int q) { return q * S - PAD + K - S; };

        for (int ch = c_begin; ch < c_end; ++ch) {
            const int n0 = ch * CT_F;
            // ---- run start: the window of step 0 and the rows of step 1, ZS rows at a time (the z staging area holds ZS
            // row sets), sources 0 and 1; then the rows of step 2 go in flight
            {
                const int pa = -PAD, pb = new_lo(1) + S;            // rows of steps 0 and 1
                for (int p = pa; p < pb; p += ZS) {
                    issue(p, min(p + ZS, pb), p == pa ? 0 : (p == pa + ZS ? 1 : -1), n0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    prologue(p, min(p + ZS, pb), n0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if (pb - pa <= ZS) { issue(0, 0, 1, n0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            }
            int n_last = issue(new_lo(2), new_lo(2) + S, 2, n0);        // rows of step 2 (only when Lin > 2)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            for (int q = 0; q < Lin; ++q) {
                const float* Ssrc = SRC + (q % RS) * CT_SET;
                const int ra = cit * 32 + c, rb = cot * 32 + c;
#pragma unroll
                for (int qq = 0; qq < CT_F / 8; ++qq) {
                    f32x4 av = *(const f32x4*)(Ssrc + ct_off(ra, 2 * qq + h));
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] = fmaxf(fmaf(av[e], sc_a, sh_a), 0.f);
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int p = q * S - PAD + k;
                        if (p >= 0 && p < Lout) {
                            const f32x4 bv = *(const f32x4*)(DZ + (p % RDZ) * CT_SET + ct_off(rb, 2 * qq + h));
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[k], 0, 0, 0);
                        }
                    }
                    if (qq == 1 && q >= 1) {
                        // outstanding: rows of step q + 1 (older) and of step q + 2 (the newest n_last): transform the
                        // rows of step q + 1 in the shadow of this step's MFMAs
                        ct_wait_vmcnt(n_last);
                        prologue(new_lo(q + 1), new_lo(q + 1) + S, n0);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();            // every LDS read of step q is done; step q + 1 is staged
                asm volatile("" ::: "memory");
                // free now: the ring slots of the rows that left the window, S staging slots, source slot (q + 3) % RS
                n_last = (q + 3 < Lin) ? issue(new_lo(q + 3), new_lo(q + 3) + S, q + 3, n0) : 0;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // the rings are reused by the next chunk
            asm volatile("" ::: "memory");
        }
        // ---- this workgroup's partial image of dW (native ConvTranspose1d layout (Ci, Co, K)) and db
        float* img = a.w_partials + (size_t)blockIdx.x * a.w_numel;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = cit * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int co = cot * 32 + c;
                img[((size_t)ci * CT_C + co) * K + k] = acc[k][r];
            }
        if (a.b_partials) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v = bsum[i];
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 1);
                if ((lane & 7) == 0) a.b_partials[(size_t)blockIdx.x * a.b_stride + a.b_off + row_[i]] = v;
            }
        }
    } else {
        // =============================================================== data gradient
        const int jj = wave - 4;
        const int cit = jj & 1, chf = jj >> 1;           // ci tile, co half
        // W^T fragments: A[i = ci][k = co] per tap: af[k][kk] = W[ci = 32 cit + (lane & 31)][co = 32 chf + 2 kk + h][k]
        float af[K][16];
        {
            const float* wp = a.W + ((size_t)(cit * 32 + c) * CT_C + chf * 32 + h) * K;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int k = 0; k < K; ++k) af[k][kk] = wp[(size_t)(2 * kk) * K + k];
        }
        float st1[16], st2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { st1[r] = 0.f; st2[r] = 0.f; }
        const size_t dstride = (size_t)Lin * NP;
        const int rowb = (int)(dstride * sizeof(float));
        const int voff = (int)((4 * h * dstride + c) * sizeof(float));

        f32x16 dacc;
        auto epilogue = [&](int q, int n0) __attribute__((always_inline)) {
            // val = own half + the other co half (through LDS), ReLU mask from the raw source row set, store, statistics
            const float* P = PART + ((q & 1) * 2 + cit) * 16 * 64 + lane;
            const float* Ssrc = SRC + (q % RS) * CT_SET;
            const __amdgpu_buffer_rsrc_t ro = ct_rsrc(a.dsrc + (size_t)(32 * cit) * dstride + (size_t)q * NP);
            const bool fin = n0 + c < a.N;
            const int nb = n0 * (int)sizeof(float);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = (r & 3) + 8 * (r >> 2);
                const int row = cit * 32 + ml + 4 * h;
                const float zs = Ssrc[row * CT_F + 4 * ((c >> 2) ^ ((row >> 1) & 7)) + (c & 3)];
                const f32x4 k = CB[row];
                float val = dacc[r] + P[r * 64];
                val = (fmaf(k[0], zs, k[1]) > 0.f) ? val : 0.f;
                ct_bstore(ro, voff, ml * rowb + nb, val);
                const float x = fin ? val : 0.f;
                st1[r] += x;
                st2[r] = fmaf(x, zs - k[2], st2[r]);
            }
        };

        for (int ch = c_begin; ch < c_end; ++ch) {
            const int n0 = ct_uniform(ch * CT_F);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                // step 0 staged
            asm volatile("" ::: "memory");
            for (int q = 0; q < Lin; ++q) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int p = q * S - PAD + k;
                    if (p >= 0 && p < Lout) {
                        const float* Sb = DZ + (p % RDZ) * CT_SET + (chf * 32 + h) * CT_F + (c & 3);
                        const int cpc = c >> 2;
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) {
                            const float b = Sb[kk * (2 * CT_F) + 4 * (cpc ^ (kk & 7))];
                            dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k][kk], b, dacc, 0, 0, 0);
                        }
                    }
                }
                if (chf == 1) {
                    float* P = PART + ((q & 1) * 2 + cit) * 16 * 64 + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) P[r * 64] = dacc[r];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();            // every LDS read of step q is done, partial sums are visible
                asm volatile("" ::: "memory");
                if (chf == 0) epilogue(q, n0);           // registers + LDS slots that stay valid during step q + 1
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // the rings are reused by the next chunk
            asm volatile("" ::: "memory");
        }
        if (chf == 0 && a.partials) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float s1 = half_wave_sum(st1[r]);
                const float s2 = half_wave_sum(st2[r]);
                const int row = cit * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (c == 0) {
                    float* pp = a.partials + ((size_t)blockIdx.x * CT_C + row) * 2;
                    pp[0] = s1;
                    pp[1] = s2;
                }
            }
        }
    }
}

template <int K, int S>
int ct_launch(const trunet_convt_bwd_args* h, hipStream_t st) {
    constexpr int RDZ = K + 2 * S, ZS = 2 * S, RS = 4;
    const size_t lds = ((size_t)(RDZ + ZS + RS) * CT_SET + 2 * 2 * 16 * 64) * sizeof(float) + 2 * CT_C * sizeof(f32x4);
    auto kern = convt_bwd_kernel<K, S>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return TRUNET_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(CT_GRID), dim3(512), lds, st, *h);
    return trunet_launch_status();
}

}  // namespace

extern "C" int trunet_convt_bwd_nparts(void) { return CT_GRID; }

// convt_bwd_x3.hip: the same kernel with both GEMMs on the bf16 MFMA (three-term operand split, fp32-grade result);
// taken when the split path is on (trunet_gemm_x3_enable / TRUNET_GEMM_X3, gemm_x3.hip)
int trunet_launch_convt_bwd_x3(const trunet_convt_bwd_args* h, hipStream_t st);
extern "C" int trunet_gemm_x3_enable(int on);

extern "C" int trunet_convt_bwd(const trunet_convt_bwd_args* h, void* stream) {
    if (!h || !h->dy || !h->z || !h->ca || !h->cb || !h->cc || !h->src || !h->s_scale || !h->s_shift || !h->s_mean ||
        !h->W || !h->dsrc || !h->partials || !h->w_partials)
        return TRUNET_EINVAL;
    if (h->NP <= 0 || (h->NP % TRUNET_TILE_FRAMES) != 0 || h->N <= 0 || h->N > h->NP || h->Lin <= 0 || h->w_numel <= 0)
        return TRUNET_EINVAL;
    if (h->Ci != CT_C || h->Co != CT_C) return TRUNET_ENOTSUP;
    if (h->pad != h->S / 2 || h->Lout != (h->Lin - 1) * h->S - 2 * h->pad + h->K) return TRUNET_EINVAL;
    // 32-bit byte offsets of the buffer stores: 36 channel rows of the gradient tensor below 2 GiB
    if ((size_t)h->Lin * h->NP * sizeof(float) * 36 >= ((size_t)1 << 31)) return TRUNET_ENOTSUP;
    hipStream_t st = (hipStream_t)stream;
    if (trunet_gemm_x3_enable(-1) & (TRUNET_X3_BWD | 8)) return trunet_launch_convt_bwd_x3(h, st);
    if (h->K == 3 && h->S == 1) return ct_launch<3, 1>(h, st);
    if (h->K == 3 && h->S == 2) return ct_launch<3, 2>(h, st);
    if (h->K == 5 && h->S == 2) return ct_launch<5, 2>(h, st);
    return TRUNET_ENOTSUP;
}
