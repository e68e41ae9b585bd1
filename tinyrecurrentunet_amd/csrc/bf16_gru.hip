// The bidirectional GRU recurrence of gru.hip (nn.GRU in GRUBlock, /root/reference/network.py:48,55) with bf16 OCTET tensors
// on both sides (round 3): gi, the recurrence output, the saved gates, and in the backward dhout, dgi, dghn are
// uint16[C/8][L][NP][8] like every other activation of the bf16 family.  The arithmetic is gru.hip's: W_hh as fp32 MFMA A
// fragments in registers, h_{t-1} / dgh exchanged through LDS in fp32 (the recurrence itself is never rounded), gate math on
// the MFMA C layout; only what crosses HBM is bf16.  Until now the bf16 step ran the fp32 kernels between conversion
// launches (gi written as fp32 by the projection GEMM, hout / dgi / dghn converted to octets, dhout converted back):
// 4.75 GB of 87.6 GB per step.
//
// C layout <-> octets: lane (h, c), accumulator register r <-> unit 32 jt + 8 (r >> 2) + 4 h + (r & 3): the four registers
// of a group q = r >> 2 are one HALF octet (8 bytes) of octet 4 jt + q, so a lane moves 8 bytes per (group, frame) and the
// two halves h of 32 consecutive frames make 512 contiguous bytes per wave instruction.  Column block e of a lane is frame
// c + 32 e (gru.hip: 2 c + e); inside LDS the column index stays NE c + e.
#include <cstdlib>
#include "bf16_common.hpp"

namespace {

constexpr int H = 64;

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

__device__ __forceinline__ void unpack4(const u32x2 v, float (&f)[4]) {
    f[0] = bf_lo(v[0]); f[1] = bf_hi(v[0]); f[2] = bf_lo(v[1]); f[3] = bf_hi(v[1]);
}
__device__ __forceinline__ u32x2 pack4(const float (&f)[4]) {
    u32x2 v = {bf_pack(f[0], f[1]), bf_pack(f[2], f[3])};
    return v;
}
#define BG_LD(p) __builtin_nontemporal_load((const u32x2*)(p))
#define BG_ST(p, v) __builtin_nontemporal_store((v), (u32x2*)(p))

// half-octet index (in units of 8 bytes) of (octet, position, frame, half)
__device__ __forceinline__ size_t hoct(int oct, int L, int pos, int NP, int n, int h) {
    return (((size_t)oct * L + pos) * NP + n) * 2 + h;
}

template <int NE>
__global__ __launch_bounds__(256, NE == 1 ? 2 : 1) void bgru_fwd_kernel(const u32x2* __restrict__ gi, const float* __restrict__ whh0,
                                                                         const float* __restrict__ bhh0, const float* __restrict__ whh1,
                                                                         const float* __restrict__ bhh1, u32x2* __restrict__ hout,
                                                                         u32x2* __restrict__ gates, int L, int NP) {
    constexpr int GFW = 32 * NE;
    __shared__ __attribute__((aligned(16))) float hs[2][2][H][GFW];  // [dir][buf][unit][column]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = wave >> 1, jt = wave & 1;
    const int h0 = lane >> 5, c0 = lane & 31;
    const float* whh = d ? whh1 : whh0;
    const float* bhh = d ? bhh1 : bhh0;
    const int n0 = blockIdx.x * GFW;

    float A[3][32];          // A[g][kk] = W_hh[(g*64 + 32*jt + c)][2*kk + h]
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) A[g][kk] = whh[(size_t)(g * H + 32 * jt + c0) * H + 2 * kk + h0];

    const size_t gs = (size_t)(H / 8) * L * NP * 2;      // one gate plane [H/8][L][NP] in half octets
    for (int t = 0; t < L; ++t) {
        const int pos = d ? (L - 1 - t) : t;
        const int buf = t & 1;
        int c = c0, h = h0;
        asm volatile("" : "+v"(c), "+v"(h));      // (gru.hip: keeps hipcc from hoisting every row address out of the loop)
        f32x16 acc[3][NE];
        float gin[16][NE];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    float v[4];
                    unpack4(BG_LD(gi + hoct(24 * d + 8 * g + 4 * jt + q, L, pos, NP, n0 + c + 32 * e, h)), v);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * q + i;
                        const float b = bhh[g * H + 32 * jt + 8 * q + 4 * h + i];
                        if (g < 2) acc[g][e][r] = v[i] + b;
                        else { gin[r][e] = v[i]; acc[2][e][r] = b; }
                    }
                }
        if (t > 0) {
            const float* hb = &hs[d][buf ^ 1][0][0];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                float b[NE];
#pragma unroll
                for (int e = 0; e < NE; ++e) b[e] = hb[(2 * kk + h) * GFW + NE * c + e];
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int e = 0; e < NE; ++e) acc[g][e] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[g][kk], b[e], acc[g][e], 0, 0, 0);
            }
        }
        float* hw = &hs[d][buf][0][0];
        const float* hp = &hs[d][buf ^ 1][0][0];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                float rr[4], zz[4], nn[4], gh[4], hn[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * q + i;
                    const int u = 32 * jt + 8 * q + 4 * h + i;
                    const float hprev = t > 0 ? hp[u * GFW + NE * c + e] : 0.f;
                    rr[i] = sigmoidf_(acc[0][e][r]);
                    zz[i] = sigmoidf_(acc[1][e][r]);
                    gh[i] = acc[2][e][r];
                    nn[i] = tanhf_(fmaf(rr[i], gh[i], gin[r][e]));
                    hn[i] = fmaf(zz[i], hprev - nn[i], nn[i]);
                    hw[u * GFW + NE * c + e] = hn[i];           // the recurrence carries fp32
                }
                const int n = n0 + c + 32 * e;
                BG_ST(hout + hoct(8 * d + 4 * jt + q, L, pos, NP, n, h), pack4(hn));
                if (gates) {
                    u32x2* gb = gates + (size_t)d * 4 * gs + hoct(4 * jt + q, L, pos, NP, n, h);
                    BG_ST(gb, pack4(rr));
                    BG_ST(gb + gs, pack4(zz));
                    BG_ST(gb + 2 * gs, pack4(nn));
                    BG_ST(gb + 3 * gs, pack4(gh));
                }
            }
        __syncthreads();
    }
}

// backward through time: gru.hip's gru_bwd_kernel on octets
template <int NE>
__global__ __launch_bounds__(256, NE == 1 ? 2 : 1) void bgru_bwd_kernel(const u32x2* __restrict__ dhout, const u32x2* __restrict__ hout,
                                                                         const u32x2* __restrict__ gates, const float* __restrict__ whh0,
                                                                         const float* __restrict__ whh1, u32x2* __restrict__ dgi,
                                                                         u32x2* __restrict__ dghn, int L, int NP) {
    constexpr int GFW = 32 * NE;
    __shared__ __attribute__((aligned(16))) float ds[2][3 * H][GFW];  // [dir][gate row][column]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = wave >> 1, jt = wave & 1;
    const int h = lane >> 5, c = lane & 31;
    const float* whh = d ? whh1 : whh0;
    const int n0 = blockIdx.x * GFW;

    float A[96];             // A[kk] = W_hh[row = 2*kk + h][unit = 32*jt + c], kk over the 192 gate rows
#pragma unroll
    for (int kk = 0; kk < 96; ++kk) A[kk] = whh[(size_t)(2 * kk + h) * H + 32 * jt + c];

    float carry[16][NE];
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int e = 0; e < NE; ++e) carry[r][e] = 0.f;
    const size_t gs = (size_t)(H / 8) * L * NP * 2;

    for (int t = L - 1; t >= 0; --t) {
        const int pos = d ? (L - 1 - t) : t;
        const int ppos = d ? pos + 1 : pos - 1;   // position of h_{t-1}
        float dhz[16][NE];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int n = n0 + c + 32 * e;
                const size_t o = hoct(4 * jt + q, L, pos, NP, n, h);
                float dho[4], rr[4], zz[4], nn[4], gh[4], hp[4];
                unpack4(BG_LD(dhout + (size_t)d * gs + o), dho);
                const u32x2* gb = gates + (size_t)d * 4 * gs + o;
                unpack4(BG_LD(gb), rr);
                unpack4(BG_LD(gb + gs), zz);
                unpack4(BG_LD(gb + 2 * gs), nn);
                unpack4(BG_LD(gb + 3 * gs), gh);
                const u32x2 hv = BG_LD(hout + (size_t)d * gs + hoct(4 * jt + q, L, t > 0 ? ppos : pos, NP, n, h));
                unpack4(hv, hp);
                float drp[4], dzp[4], dnp[4], dgn[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * q + i;
                    const int u = 32 * jt + 8 * q + 4 * h + i;
                    const float hpv = t > 0 ? hp[i] : 0.f;
                    const float dh = dho[i] + carry[r][e];
                    const float dn = dh * (1.f - zz[i]);
                    dnp[i] = dn * (1.f - nn[i] * nn[i]);
                    dzp[i] = dh * (hpv - nn[i]) * zz[i] * (1.f - zz[i]);
                    drp[i] = dnp[i] * gh[i] * rr[i] * (1.f - rr[i]);
                    dgn[i] = dnp[i] * rr[i];
                    dhz[r][e] = dh * zz[i];
                    ds[d][u][NE * c + e] = drp[i];                 // the carry's MFMA runs on the unrounded values
                    ds[d][H + u][NE * c + e] = dzp[i];
                    ds[d][2 * H + u][NE * c + e] = dgn[i];
                }
                u32x2* go = dgi + hoct(24 * d + 4 * jt + q, L, pos, NP, n, h);
                BG_ST(go, pack4(drp));
                BG_ST(go + gs, pack4(dzp));
                BG_ST(go + 2 * gs, pack4(dnp));
                BG_ST(dghn + (size_t)d * gs + o, pack4(dgn));
            }
        __syncthreads();
        f32x16 acc[NE];
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int e = 0; e < NE; ++e) acc[e][r] = dhz[r][e];
        if (t > 0) {
#pragma unroll
            for (int kk = 0; kk < 96; ++kk) {
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[kk], ds[d][2 * kk + h][NE * c + e], acc[e], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int e = 0; e < NE; ++e) carry[r][e] = acc[e][r];
        __syncthreads();
    }
}

// TRUNET_GRU_NE = 1 / 2 forces the column blocks per workgroup of both kernels (A/B measurements); default: forward 2 (one
// workgroup per CU), backward 1 (two per CU), as measured for the fp32 kernels
int bgru_ne(int dflt) {
    static const int v = [] { const char* e = getenv("TRUNET_GRU_NE"); return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0; }();
    return v ? v : dflt;
}

}  // namespace

extern "C" int trunet_bf16_gru_fwd(const void* gi, const float* w_hh, const float* b_hh, const float* w_hh_rev,
                                   const float* b_hh_rev, void* hout, void* gates, int Hh, int L, int NP, void* stream) {
    if (!gi || !w_hh || !b_hh || !w_hh_rev || !b_hh_rev || !hout || (NP % 128) || L <= 0) return TRUNET_EINVAL;
    if (Hh != H) return TRUNET_ENOTSUP;
    if (bgru_ne(2) == 2)
        hipLaunchKernelGGL(bgru_fwd_kernel<2>, dim3(NP / 64), dim3(256), 0, (hipStream_t)stream, (const u32x2*)gi, w_hh, b_hh,
                           w_hh_rev, b_hh_rev, (u32x2*)hout, (u32x2*)gates, L, NP);
    else
        hipLaunchKernelGGL(bgru_fwd_kernel<1>, dim3(NP / 32), dim3(256), 0, (hipStream_t)stream, (const u32x2*)gi, w_hh, b_hh,
                           w_hh_rev, b_hh_rev, (u32x2*)hout, (u32x2*)gates, L, NP);
    return trunet_launch_status();
}

extern "C" int trunet_bf16_gru_bwd(const void* dhout, const void* hout, const void* gates, const float* w_hh,
                                   const float* w_hh_rev, void* dgi, void* dghn, int Hh, int L, int NP, void* stream) {
    if (!dhout || !hout || !gates || !w_hh || !w_hh_rev || !dgi || !dghn || (NP % 128) || L <= 0) return TRUNET_EINVAL;
    if (Hh != H) return TRUNET_ENOTSUP;
    if (bgru_ne(1) == 2)
        hipLaunchKernelGGL(bgru_bwd_kernel<2>, dim3(NP / 64), dim3(256), 0, (hipStream_t)stream, (const u32x2*)dhout,
                           (const u32x2*)hout, (const u32x2*)gates, w_hh, w_hh_rev, (u32x2*)dgi, (u32x2*)dghn, L, NP);
    else
        hipLaunchKernelGGL(bgru_bwd_kernel<1>, dim3(NP / 32), dim3(256), 0, (hipStream_t)stream, (const u32x2*)dhout,
                           (const u32x2*)hout, (const u32x2*)gates, w_hh, w_hh_rev, (u32x2*)dgi, (u32x2*)dghn, L, NP);
    return trunet_launch_status();
}
