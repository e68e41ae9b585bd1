// bf16 family: fused backward of ConvTranspose1d(64 -> 64, k, s) + BatchNorm (autograd of network.py:67,86: decoder.0-4).
//
// The separate launches (trunet_bf16_wgrad over the tap segments + trunet_bf16_gemm over the taps of (dy, z)) read dy and z
// twice from HBM and K/S more times through L2, and run the BatchNorm-backward prologue on every one of those reads.  Here
// a step is (SOURCE position q, 64 frames), as in the fp32 convt_bwd_kernel: the dz rows p = q S - pad + k of its K taps are a
// sliding window -- a ring of R >= K + S LDS images [octet][frame][8], each row loaded, transformed (dz = ca dy + cb z + cc,
// frames >= N zeroed) and stored ONCE per 64-frame chunk -- and the source row q sits next to it (activated and raw).  From
// that one LDS state the step produces
//   * the weight gradient of every tap:  dW[ci][co][k] += dz[co][p_k] . act[ci][q]   (frames = MFMA K axis: both operands
//     through ds_read_b64_tr_b16; waves 4-7 own one (co tile, ci tile) each for all K taps: K accumulators),
//   * the COMPLETE data gradient of source row q:  dsrc[ci][q] = sum_k sum_co W[ci][co][k] dz[co][p_k]   (dz rows as B operand,
//     W^T fragments of all taps in LDS; waves 0-3 own one (ci tile, frame half) each), with ReLU mask, BatchNorm-backward
//     statistics and the 8-byte stores of trunet_bf16_gemm's epilogue -- no accumulation across steps,
//   * the bias gradient (every dz row counted when it enters the ring).
// 20 (k = 5) or 12 (k = 3) MFMAs per wave and step on either role.  A workgroup owns whole 64-frame chunks.
#include "bf16_common.hpp"

namespace {

constexpr int BCT_THREADS = 512;

template <int K, int S>
__global__ __launch_bounds__(BCT_THREADS, 1) void bconvt_bwd_kernel(const trunet_bconvt_args a) {
    constexpr int PAD = S / 2;
    constexpr int R = (K + S <= 4) ? 4 : 8;            // ring slots (power of two >= K + S)
    constexpr int MAXNEW = K - PAD;                    // dz rows entering the ring at the first step of a chunk
    constexpr int NKS = 4 * K;                         // k-steps of the W^T image per row tile: (tap, 16 dz channels)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
    unsigned char* ring = smem_;                                   // [R][8 octets] dz images
    unsigned char* actb = ring + R * 8 * BW_OS;                    // [2][8 octets] activated source rows (by q parity)
    unsigned char* rawb = actb + 2 * 8 * BW_OS;                    // [2][8 octets] raw source rows
    float* Cd = (float*)(rawb + 2 * 8 * BW_OS);                    // [8][3][8] ca, cb, cc
    float* Cs = Cd + 8 * 24;                                       // [8][2][8] scale, shift of the source
    float* Cm = Cs + 8 * 16;                                       // [8][8] mean of the source
    u32x4* WT = (u32x4*)(Cm + 64);                                 // [2 row tiles][NKS][64] W^T fragments
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;

    for (int i = tid; i < 64; i += BCT_THREADS) {
        const int oct = i >> 3, e = i & 7;
        Cd[oct * 24 + e] = a.ca[i]; Cd[oct * 24 + 8 + e] = a.cb[i]; Cd[oct * 24 + 16 + e] = a.cc[i];
        Cs[oct * 16 + e] = a.s_scale[i]; Cs[oct * 16 + 8 + e] = a.s_shift[i];
        Cm[i] = a.s_mean[i];
    }
    for (int i = tid; i < 2 * NKS * 64; i += BCT_THREADS) WT[i] = ((const u32x4*)a.wfragT)[i];
    {
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (int i = tid; i < (R + 4) * 8 * BW_OS / 16; i += BCT_THREADS) ((u32x4*)smem_)[i] = z4;
    }
    __syncthreads();

    const int nchunks = a.NP / BW_F;
    const int c_begin = (int)(((long long)blockIdx.x * nchunks) / gridDim.x);
    const int c_end = (int)(((long long)(blockIdx.x + 1) * nchunks) / gridDim.x);

    // a step: (chunk, q) and the dz rows [lo, hi] that enter the ring with it
    struct Info { int chunk, q, lo, hi; };
    auto hi_of = [&](int q) { return min(q * S - PAD + K - 1, a.Lout - 1); };
    auto first = [&](int chunk) { Info f; f.chunk = chunk; f.q = 0; f.lo = 0; f.hi = hi_of(0); return f; };
    auto next = [&](const Info& f) {
        Info g;
        if (f.q + 1 < a.Lin) { g.chunk = f.chunk; g.q = f.q + 1; g.lo = f.hi + 1; g.hi = hi_of(g.q); }
        else g = first(f.chunk + 1);
        return g;
    };

    struct Stage { u32x4 dy[MAXNEW], z[MAXNEW], s; };
    Stage st;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    f32x16 wacc[K];                                    // waves 4-7: (co tile, ci tile) of every tap
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[k][r] = 0.f;
    float sacc0 = 0.f, sacc1 = 0.f;                    // waves 0-3: statistics of their (ci tile, frame half)

    auto issue = [&](const Info& f) {                  // wave w loads octet w of every new dz row and of the source row
        const size_t n = (size_t)f.chunk * BW_F + lane;
#pragma unroll
        for (int j = 0; j < MAXNEW; ++j) {
            const int p = f.lo + j;
            if (p <= f.hi) {
                const size_t idx = ((size_t)wave * a.Lout + p) * a.NP + n;
                st.dy[j] = ((const u32x4*)a.dy)[idx];
                st.z[j] = ((const u32x4*)a.z)[idx];
            }
        }
        st.s = ((const u32x4*)a.src)[((size_t)wave * a.Lin + f.q) * a.NP + n];
    };
    auto store = [&](const Info& f) {
        const bool fin = f.chunk * BW_F + lane < a.N;
        const float* cd = Cd + wave * 24;
#pragma unroll
        for (int j = 0; j < MAXNEW; ++j) {
            const int p = f.lo + j;
            if (p <= f.hi) {
                float v[8], w[8];
                bf_unpack8(st.dy[j], v);
                bf_unpack8(st.z[j], w);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = fin ? fmaf(cd[e], v[e], fmaf(cd[8 + e], w[e], cd[16 + e])) : 0.f;
                    bsum[e] += v[e];
                }
                *(u32x4*)(ring + ((p & (R - 1)) * 8 + wave) * BW_OS + lane * 16) = bf_pack8(v);
            }
        }
        {
            const int par = f.q & 1;
            *(u32x4*)(rawb + (par * 8 + wave) * BW_OS + lane * 16) = st.s;
            const float* cs = Cs + wave * 16;
            *(u32x4*)(actb + (par * 8 + wave) * BW_OS + lane * 16) = bf_affine8<true>(st.s, cs, cs + 8);
        }
    };
    auto mma = [&](const Info& f) {
        const int par = f.q & 1;
        const int p0 = f.q * S - PAD;
        if (wave >= 4) {
            // ---- weight gradient: (co tile rt, ci tile ct) for all taps; the source fragment is shared by the taps
            const int rt = (wave - 4) >> 1, ct = (wave - 4) & 1;
#pragma unroll
            for (int kk = 0; kk < BW_F / 16; ++kk) {
                const bf16x8 bfr = lds_frag(actb + par * 8 * BW_OS, 4 * ct, 16 * kk, lane);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int p = p0 + k;
                    if (p >= 0 && p < a.Lout) {        // wave-uniform
                        const bf16x8 af = lds_frag(ring + (p & (R - 1)) * 8 * BW_OS, 4 * rt, 16 * kk, lane);
                        wacc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, wacc[k], 0, 0, 0);
                    }
                }
            }
        } else {
            // ---- data gradient of source row q: (ci tile rt, frame half cb), complete in this step
            const int rt = wave >> 1, cb = wave & 1;
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int p = p0 + k;
                if (p >= 0 && p < a.Lout) {
                    const unsigned char* bcol = ring + ((p & (R - 1)) * 8 + h) * BW_OS + (32 * cb + c) * 16;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const bf16x8 af = __builtin_bit_cast(bf16x8, WT[(rt * NKS + 4 * k + ks) * 64 + lane]);
                        const bf16x8 bfr = __builtin_bit_cast(bf16x8, *(const u32x4*)(bcol + 2 * ks * BW_OS));
                        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, d, 0, 0, 0);
                    }
                }
            }
            const int nn = f.chunk * BW_F + 32 * cb + c;
            const bool fin = nn < a.N;
            float st1[16], st2[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int og = rt * 4 + g4;
                const size_t eidx = (((size_t)og * a.Lin + f.q) * a.NP + nn) * 2 + h;
                const u32x2 zz = *(const u32x2*)(rawb + (par * 8 + og) * BW_OS + (32 * cb + c) * 16 + 8 * h);
                const float zv[4] = {bf_lo(zz[0]), bf_hi(zz[0]), bf_lo(zz[1]), bf_hi(zz[1])};
                const f32x4 e0v = *(const f32x4*)(Cs + og * 16 + 4 * h), e1v = *(const f32x4*)(Cs + og * 16 + 8 + 4 * h);
                const f32x4 muv = *(const f32x4*)(Cm + og * 8 + 4 * h);
                float val[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = (fmaf(e0v[e], zv[e], e1v[e]) > 0.f) ? d[4 * g4 + e] : 0.f;
                u32x2 o;
                o[0] = bf_pack(val[0], val[1]);
                o[1] = bf_pack(val[2], val[3]);
                ((u32x2*)a.dsrc)[eidx] = o;
                const float rv[4] = {bf_lo(o[0]), bf_hi(o[0]), bf_lo(o[1]), bf_hi(o[1])};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = fin ? rv[e] : 0.f;
                    st1[4 * g4 + e] = x;
                    st2[4 * g4 + e] = x * (zv[e] - muv[e]);
                }
            }
            sacc0 += butterfly16(st1, c);
            sacc1 += butterfly16(st2, c);
        }
    };

    if (c_begin < c_end) {
        Info f = first(c_begin);
        issue(f);
        while (f.chunk < c_end) {
            store(f);
            const Info cur = f;
            f = next(f);
            if (f.chunk < c_end) issue(f);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            mma(cur);
            if (cur.q == a.Lin - 1) {       // the next chunk restarts the ring at row 0: nobody may still read this chunk's rows
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }

    // ---- weight-gradient partial image, bias partial row, statistics rows
    if (wave >= 4) {
        const int rt = (wave - 4) >> 1, ct = (wave - 4) & 1;
        float* img = a.w_partials + (size_t)blockIdx.x * a.w_numel;
        const int ci = ct * 32 + c;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                img[((size_t)ci * 64 + co) * K + k] = wacc[k][r];          // W[ci][co][k]
            }
    } else {
        const int rt = wave >> 1, cb = wave & 1;
        const int r = butterfly16_index(c);
        const int ch = rt * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        float* pp = a.partials + ((size_t)(blockIdx.x * 2 + cb) * 64 + ch) * 2;
        if (butterfly16_writer(c)) { pp[0] = sacc0; pp[1] = sacc1; }
    }
    if (a.b_partials) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = wave_sum(bsum[e]);
            if (lane == 0) a.b_partials[(size_t)blockIdx.x * a.b_stride + a.b_off + wave * 8 + e] = v;
        }
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int trunet_bf16_convt_bwd_nparts(void) { return 2 * trunet_conv_wgrad_nparts(); }

extern "C" int trunet_bf16_convt_bwd(const trunet_bconvt_args* h, void* stream) {
    if (!h || !h->dy || !h->z || !h->ca || !h->cb || !h->cc || !h->src || !h->s_scale || !h->s_shift || !h->s_mean ||
        !h->wfragT || !h->dsrc || !h->partials || !h->w_partials)
        return TRUNET_EINVAL;
    if (h->NP <= 0 || (h->NP % BW_F) != 0 || h->N <= 0 || h->N > h->NP || h->Lin <= 0 || h->w_numel <= 0) return TRUNET_EINVAL;
    if (h->pad != h->S / 2 || h->Lout != (h->Lin - 1) * h->S - 2 * h->pad + h->K) return TRUNET_EINVAL;
    if (h->Ci != 64 || h->Co != 64) return TRUNET_ENOTSUP;
    const int K = h->K, S = h->S;
    if (!((K == 3 && S == 1) || (K == 5 && S == 2) || (K == 3 && S == 2))) return TRUNET_ENOTSUP;
    const int R = (K + S <= 4) ? 4 : 8;
    const size_t lds = (size_t)(R + 4) * 8 * BW_OS + (size_t)(8 * 24 + 8 * 16 + 64) * sizeof(float) + (size_t)2 * 4 * K * 64 * 16;
    if (!h->prezero &&
        hipMemsetAsync(h->partials, 0, (size_t)trunet_bf16_convt_bwd_nparts() * 64 * 2 * sizeof(float), ST) != hipSuccess)
        return TRUNET_ELAUNCH;
#define BCT_LAUNCH(KK, SS)                                                                                                          \
    do {                                                                                                                            \
        if (hipFuncSetAttribute((const void*)bconvt_bwd_kernel<KK, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=   \
            hipSuccess)                                                                                                             \
            return TRUNET_ELAUNCH;                                                                                                  \
        hipLaunchKernelGGL((bconvt_bwd_kernel<KK, SS>), dim3(trunet_conv_wgrad_nparts()), dim3(BCT_THREADS), lds, ST, *h);          \
    } while (0)
    if (K == 3 && S == 1) BCT_LAUNCH(3, 1);
    else if (K == 5) BCT_LAUNCH(5, 2);
    else BCT_LAUNCH(3, 2);
#undef BCT_LAUNCH
    return trunet_launch_status();
}
