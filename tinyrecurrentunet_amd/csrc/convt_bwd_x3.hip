// The fused ConvTranspose1d(64 -> 64) + BatchNorm backward of convt_bwd.hip with BOTH GEMMs on the bf16 MFMA through the
// three-term operand split of gemm_x3.hip (round 4).  Same decomposition, rings, DMA schedule, prologue and epilogue as
// convt_bwd_kernel<K, S> -- read that file's header first; what differs is how a staged fp32 fragment reaches the matrix pipe:
//   weight gradient (waves 0-3): per 16 frames the source row's 8 values of a lane (two swizzled 16-byte pieces) get
//       BatchNorm+ReLU and are split ONCE, every valid tap's dz row fragment is split, 6 v_mfma_f32_32x32x16_bf16 per tap;
//   data gradient (waves 4-7): W^T of the wave's (ci tile, co half) for all taps sits in registers as three bf16 fragment
//       planes (split once per kernel), per tap and 16 co rows the lane's 8 dz values (8 ds_read_b32) are split, 6 MFMAs.
// Here a staged fragment feeds ONE 32-row tile per wave, so the split is not amortised (about 7 vector instructions per
// MFMA: the waves are vector-bound) -- and still well ahead of the fp32 MFMA: 6 x 33 matrix cycles + ~45 x 4 vector cycles
// per (32 x 32 x 16) against 8 x 64.  convt_bwd_kernel was the most matrix-bound kernel of the step (matrix pipe 62-65 %
// busy, HBM at 0.25 of its peak).
#include "common.hpp"

#include "x3_common.hpp"

namespace {

constexpr int CT_F = 32;            // frames per tile
constexpr int CT_C = 64;            // channels (both sides)
constexpr int CT_SET = CT_C * CT_F; // floats per row set (8 KB)
constexpr int CT_GRID = TRUNET_NUM_CU;
constexpr int CT_LO = CT_C * 16;    // floats per row set of the lo-plane ring (8 bytes per 4 frames)

// (round 4, as in pw_bwd.hip) dz is split ONCE, by the prologue pass that computes it: the dz ring keeps [hi | mid] of a
// piece's 4 frames in the piece's own 16 bytes, a second ring of 4 KB slots the lo plane.  The weight-gradient B fragments
// (K = frames) are then 8-byte reads, the data-gradient B fragments (K = dz rows) transposing reads (ds_read_b64_tr_b16),
// neither with vector work; only the source rows are still split by the wave that consumes them.  Needs (K + 2 S) x 4 KB
// more LDS: k3 s1 fits (126 KB); k3 s2 (166 KB) and k5 s2 (190 KB) do not and keep the consumer-side split.
template <int K, int S>
constexpr bool ct_dzp() { return K == 3 && S == 1; }

typedef short ct_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 ct_tr16(const float* p) {
    const ct_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ct_s16x4*)p);
    return __builtin_bit_cast(u32x2, v);
}
// float offset of the 8 lo bytes of logical piece `pc` of row `r` inside a lo-ring slot (same swizzle as the row set)
__device__ __forceinline__ int ct_lo_off(int r, int pc) { return r * 16 + 2 * (pc ^ ((r >> 1) & 7)); }

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
__global__ __launch_bounds__(512, 1) void convt_bwd_x3_kernel(const trunet_convt_bwd_args a) {
    constexpr int PAD = S / 2;
    constexpr int RDZ = K + 2 * S;      // dz ring: window + the rows of the next two steps
    constexpr int ZS = 2 * S;           // z staging slots
    constexpr int RS = 4;               // source ring: q (+ epilogue of q - 1), q + 1, q + 2
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* DZ = smem;                               // [RDZ][64][32]
    float* ZST = DZ + RDZ * CT_SET;                 // [ZS][64][32]
    float* SRC = ZST + ZS * CT_SET;                 // [RS][64][32]
    float* PART = SRC + RS * CT_SET;                // [2 parity][2 ci tiles][16][64]
    constexpr bool DZP = ct_dzp<K, S>();
    float* LO = PART + 2 * 2 * 16 * 64;             // (DZP) [RDZ][64][16]: lo plane of the dz ring
    f32x4* CA = (f32x4*)(LO + (DZP ? RDZ * CT_LO : 0));   // [64] (ca, cb, cc, 0) of dz
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
                    if constexpr (DZP) {
                        unsigned h0, m0, l0, h1, m1, l1;
                        ctx_split2(v[0], v[1], h0, m0, l0);
                        ctx_split2(v[2], v[3], h1, m1, l1);
                        *(u32x4*)(dst + o) = u32x4{h0, h1, m0, m1};
                        *(u32x2*)(LO + (p % RDZ) * CT_LO + row_[i] * 16 + 2 * pc) = u32x2{l0, l1};
                    } else {
                        *(f32x4*)(dst + o) = v;
                    }
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
                for (int qq = 0; qq < CT_F / 16; ++qq) {
                    // this lane's 8 frames of the K-step: 16 qq + 8 h .. + 7 = pieces 4 qq + 2 h, 4 qq + 2 h + 1 of its row
                    f32x4 av0 = *(const f32x4*)(Ssrc + ct_off(ra, 4 * qq + 2 * h));
                    f32x4 av1 = *(const f32x4*)(Ssrc + ct_off(ra, 4 * qq + 2 * h + 1));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        av0[e] = fmaxf(fmaf(av0[e], sc_a, sh_a), 0.f);
                        av1[e] = fmaxf(fmaf(av1[e], sc_a, sh_a), 0.f);
                    }
                    u32x4 a0, a1, a2;
                    ctx_split8(av0, av1, a0, a1, a2);
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int p = q * S - PAD + k;
                        if (p >= 0 && p < Lout) {
                            const float* Dp = DZ + (p % RDZ) * CT_SET;
                            u32x4 b0, b1, b2;
                            if constexpr (DZP) {
                                const float* Lp = LO + (p % RDZ) * CT_LO;
                                const u32x4 q0 = *(const u32x4*)(Dp + ct_off(rb, 4 * qq + 2 * h));
                                const u32x4 q1 = *(const u32x4*)(Dp + ct_off(rb, 4 * qq + 2 * h + 1));
                                const u32x2 l0 = *(const u32x2*)(Lp + ct_lo_off(rb, 4 * qq + 2 * h));
                                const u32x2 l1 = *(const u32x2*)(Lp + ct_lo_off(rb, 4 * qq + 2 * h + 1));
                                b0 = u32x4{q0[0], q0[1], q1[0], q1[1]};
                                b1 = u32x4{q0[2], q0[3], q1[2], q1[3]};
                                b2 = u32x4{l0[0], l0[1], l1[0], l1[1]};
                            } else {
                                const f32x4 bv0 = *(const f32x4*)(Dp + ct_off(rb, 4 * qq + 2 * h));
                                const f32x4 bv1 = *(const f32x4*)(Dp + ct_off(rb, 4 * qq + 2 * h + 1));
                                ctx_split8(bv0, bv1, b0, b1, b2);
                            }
                            CTX_MF6(acc[k], a0, a1, a2, b0, b1, b2);
                        }
                    }
                    if (qq == 0 && q >= 1) {
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
        const int trr = 8 * (lane >> 5) + ((lane & 15) >> 2), trp = 4 * ((lane >> 4) & 1) + (lane & 3);     // (DZP) transposing reads
        // W^T fragments: A[i = ci][k = co] per tap: af[k][kk] = W[ci = 32 cit + (lane & 31)][co = 32 chf + 2 kk + h][k]
        // W^T fragment planes: lane (row ci = 32 cit + c, k = co = 32 chf + 16 ks + 8 h + j, j < 8) per tap and K-step
        u32x4 A0[K][2], A1[K][2], A2[K][2];
        {
            const float* wp = a.W + ((size_t)(cit * 32 + c) * CT_C + chf * 32 + 8 * h) * K;
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    f32x4 w0, w1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        w0[j] = wp[(size_t)(16 * ks + j) * K + k];
                        w1[j] = wp[(size_t)(16 * ks + 4 + j) * K + k];
                    }
                    ctx_split8(w0, w1, A0[k][ks], A1[k][ks], A2[k][ks]);
                }
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
                        // B fragment of K-step ks: this lane's frame c, dz rows 32 chf + 16 ks + 8 h + j (j < 8); row r keeps
                        // its 16-byte pieces XOR-swizzled by (r >> 1) & 7 = (4 h + (j >> 1)) & 7 (32 chf + 16 ks is a multiple of 16)
                        if constexpr (DZP) {
                            // lane (G = lane >> 4, i = lane & 15) fetches the 4 frames of logical piece 4 (G & 1) + (i & 3) of row
                            // 32 chf + 16 ks + 8 h + 4 t + (i >> 2) and receives frame c of the rows 8 h + 4 t .. + 3 (t = 0, 1)
                            const float* Dq = DZ + (p % RDZ) * CT_SET;
                            const float* Lq = LO + (p % RDZ) * CT_LO;
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks) {
                                const int r0 = chf * 32 + 16 * ks + trr;
                                const u32x2 h0 = ct_tr16(Dq + ct_off(r0, trp)), h1 = ct_tr16(Dq + ct_off(r0 + 4, trp));
                                const u32x2 m0 = ct_tr16(Dq + ct_off(r0, trp) + 2), m1 = ct_tr16(Dq + ct_off(r0 + 4, trp) + 2);
                                const u32x2 l0 = ct_tr16(Lq + ct_lo_off(r0, trp)), l1 = ct_tr16(Lq + ct_lo_off(r0 + 4, trp));
                                const u32x4 b0 = {h0[0], h0[1], h1[0], h1[1]}, b1 = {m0[0], m0[1], m1[0], m1[1]},
                                            b2 = {l0[0], l0[1], l1[0], l1[1]};
                                CTX_MF6(dacc, A0[k][ks], A1[k][ks], A2[k][ks], b0, b1, b2);
                            }
                        } else {
                        const float* Sb = DZ + (p % RDZ) * CT_SET + (chf * 32 + 8 * h) * CT_F + (c & 3);
                        const int cpc = c >> 2;
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            f32x4 x0, x1;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                x0[j] = Sb[(16 * ks + j) * CT_F + 4 * (cpc ^ ((4 * h + (j >> 1)) & 7))];
                                x1[j] = Sb[(16 * ks + 4 + j) * CT_F + 4 * (cpc ^ ((4 * h + 2 + (j >> 1)) & 7))];
                            }
                            u32x4 b0, b1, b2;
                            ctx_split8(x0, x1, b0, b1, b2);
                            CTX_MF6(dacc, A0[k][ks], A1[k][ks], A2[k][ks], b0, b1, b2);
                        }
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
int ctx_launch(const trunet_convt_bwd_args* h, hipStream_t st) {
    constexpr int RDZ = K + 2 * S, ZS = 2 * S, RS = 4;
    const size_t lds = ((size_t)(RDZ + ZS + RS) * CT_SET + 2 * 2 * 16 * 64 + (ct_dzp<K, S>() ? RDZ * CT_LO : 0)) * sizeof(float) +
                       2 * CT_C * sizeof(f32x4);
    static_assert(!ct_dzp<K, S>() || ((size_t)(RDZ + ZS + RS) * CT_SET + 2 * 2 * 16 * 64 + RDZ * CT_LO) * sizeof(float) +
                                         2 * CT_C * sizeof(f32x4) <= 160 * 1024, "LDS");
    auto kern = convt_bwd_x3_kernel<K, S>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return TRUNET_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(CT_GRID), dim3(512), lds, st, *h);
    return trunet_launch_status();
}

}  // namespace

// called by trunet_convt_bwd (convt_bwd.hip) after its argument checks when the bf16-split path is on
int trunet_launch_convt_bwd_x3(const trunet_convt_bwd_args* h, hipStream_t st) {
    if (h->K == 3 && h->S == 1) return ctx_launch<3, 1>(h, st);
    if (h->K == 3 && h->S == 2) return ctx_launch<3, 2>(h, st);
    if (h->K == 5 && h->S == 2) return ctx_launch<5, 2>(h, st);
    return TRUNET_ENOTSUP;
}
