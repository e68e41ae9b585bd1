// Bidirectional GRU recurrence over the L = 16 frequency positions (nn.GRU in GRUBlock,
// /root/reference/network.py:48,55), frames-last layout, hidden size H = 64.
//
// A workgroup owns 64 frames; wave w = (direction d = w>>1, unit half jt = w&1).  The recurrent
// matrix slice of the wave (3 gates x 32 units x 64) lives in registers as MFMA A-fragments for the
// whole kernel; h_{t-1} (64 units x 64 frames per direction) is exchanged through LDS; gate
// pre-activations from the input projection (gi, computed by the implicit-GEMM kernel) seed the
// accumulators, so the gate math runs directly on the MFMA C layout (rows = units, cols = frames).
// torch gate order (r, z, n):  r = s(gi_r + W_hr h + b_hr), z = s(gi_z + W_hz h + b_hz),
// n = tanh(gi_n + r * (W_hn h + b_hn)), h' = (1 - z) n + z h.
#include <cstdlib>
#include "common.hpp"

// gi, the gates and the recurrence outputs are touched once per launch: non-temporal policy (same-box A/B: -0.15 ms per step)

namespace {

constexpr int H = 64;

// Gate nonlinearities on the hardware exp2 / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each) instead of libm's expf / tanhf /
// IEEE division: sigmoid to ~3e-7 relative, tanh to ~2e-7 ABSOLUTE (1 - 2/(1+e^{2x}) cancels for small x, which is
// harmless where n enters h' = (1-z) n + z h additively).  The gate math was the largest part of a recurrence step.
#ifndef GRU_LIBM
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
#else
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }
#endif

// NE = column blocks (32 frames each) per workgroup.  NE = 2: one workgroup per CU (96 A-fragment + 96 accumulator registers
// + 32 of gi_n per wave).  NE = 1 (round 3): half the accumulators, so TWO workgroups share a CU (two waves per SIMD, <= 256
// registers each): one's load latency, gate math (transcendentals) and stores run under the other's MFMAs -- with one wave
// per SIMD every phase of a step was exposed (measured 17 us per step against 5 us of MFMA time).
template <int NE> struct GruV;
template <> struct GruV<2> {
    typedef f32x2 T;
    static __device__ __forceinline__ T ld(const float* p) { return __builtin_nontemporal_load((const f32x2*)p); }
    static __device__ __forceinline__ void st(float* p, T v) { __builtin_nontemporal_store(v, (f32x2*)p); }
    static __device__ __forceinline__ float get(const T& v, int e) { return v[e]; }
    static __device__ __forceinline__ void set(T& v, int e, float x) { v[e] = x; }
};
template <> struct GruV<1> {
    typedef float T;
    static __device__ __forceinline__ T ld(const float* p) { return __builtin_nontemporal_load(p); }
    static __device__ __forceinline__ void st(float* p, T v) { __builtin_nontemporal_store(v, p); }
    static __device__ __forceinline__ float get(const T& v, int) { return v; }
    static __device__ __forceinline__ void set(T& v, int, float x) { v = x; }
};

template <int NE>
__global__ __launch_bounds__(256, NE == 1 ? 2 : 1) void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ whh0,
                                                         const float* __restrict__ bhh0, const float* __restrict__ whh1,
                                                         const float* __restrict__ bhh1, float* __restrict__ hout,
                                                         float* __restrict__ gates, int L, int NP) {
    typedef GruV<NE> V;
    typedef typename V::T VT;
    constexpr int GFW = 32 * NE;        // frames per workgroup
    __shared__ __attribute__((aligned(16))) float hs[2][2][H][GFW];  // [dir][buf][unit][frame]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = wave >> 1, jt = wave & 1;
    const int h0 = lane >> 5, c0 = lane & 31;
    const float* whh = d ? whh1 : whh0;
    const float* bhh = d ? bhh1 : bhh0;
    const int n0 = blockIdx.x * GFW;

    // A fragments: A[g][kk] = W_hh[(g*64 + 32*jt + c)][2*kk + h]
    float A[3][32];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) A[g][kk] = whh[(size_t)(g * H + 32 * jt + c0) * H + 2 * kk + h0];

    // Register budget: 96 A-fragment + 48 NE accumulator registers are fixed.  (1) h_{t-1} of this lane's (unit, frame)
    // pairs is re-read from the LDS exchange buffer in the gate phase instead of living in registers across the MFMA
    // loop; (2) the per-lane part of every global address is re-derived from an opaque copy of the lane index inside the
    // step loop: hipcc otherwise hoists ~80 loop-invariant 64-bit row addresses (gi, hout, gates) out of the loop and
    // spills (round 1: 76 VGPRs, 304 B of scratch per lane).
    for (int t = 0; t < L; ++t) {
        const int pos = d ? (L - 1 - t) : t;
        const int buf = t & 1;
        int c = c0, h = h0;
        asm volatile("" : "+v"(c), "+v"(h));
        f32x16 acc[3][NE];
        float gin[16][NE];
        // seed accumulators with gi (+ b_hh); keep gi_n aside, gh_n accumulates on b_hn alone
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const VT v = V::ld(gi + ((size_t)(d * 3 * H + g * H + u) * L + pos) * NP + n0 + NE * c);
                const float b = bhh[g * H + u];
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    if (g < 2) acc[g][e][r] = V::get(v, e) + b;
                    else { gin[r][e] = V::get(v, e); acc[2][e][r] = b; }
                }
            }
        }
        if (t > 0) {
            const float* hb = &hs[d][buf ^ 1][0][0];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const VT b = *(const VT*)(hb + (2 * kk + h) * GFW + NE * c);
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int e = 0; e < NE; ++e)
                        acc[g][e] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[g][kk], V::get(b, e), acc[g][e], 0, 0, 0);
            }
        }
        float* hw = &hs[d][buf][0][0];
        const float* hp = &hs[d][buf ^ 1][0][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * h;
            VT rr, zz, nn, gh, hn, hprev;
#pragma unroll
            for (int e = 0; e < NE; ++e) V::set(hprev, e, 0.f);
            if (t > 0) hprev = *(const VT*)(hp + u * GFW + NE * c);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float r_ = sigmoidf_(acc[0][e][r]);
                const float z_ = sigmoidf_(acc[1][e][r]);
                const float g_ = acc[2][e][r];
                const float n_ = tanhf_(fmaf(r_, g_, gin[r][e]));
                V::set(rr, e, r_); V::set(zz, e, z_); V::set(gh, e, g_); V::set(nn, e, n_);
                V::set(hn, e, fmaf(z_, V::get(hprev, e) - n_, n_));
            }
            *(VT*)(hw + u * GFW + NE * c) = hn;
            const size_t o = ((size_t)(d * H + u) * L + pos) * NP + n0 + NE * c;
            V::st(hout + o, hn);
            if (gates) {
                const size_t gs = (size_t)H * L * NP;   // one [H][L][NP] plane
                float* gb = gates + (size_t)d * 4 * gs + ((size_t)u * L + pos) * NP + n0 + NE * c;
                V::st(gb, rr);
                V::st(gb + gs, zz);
                V::st(gb + 2 * gs, nn);
                V::st(gb + 3 * gs, gh);
            }
        }
        __syncthreads();
    }
}

// Backward through time.  Per step (reverse order of the forward):
//   dh = dhout[pos] + carry;  dn = dh (1-z);  dnp = dn (1-n^2);  dzp = dh (hprev - n) z (1-z);
//   drp = dnp * ghn * r (1-r);  dgi = (drp, dzp, dnp);  dgh = (drp, dzp, dnp*r);
//   carry = dh z + W_hh^T dgh.
template <int NE>
__global__ __launch_bounds__(256, NE == 1 ? 2 : 1) void gru_bwd_kernel(const float* __restrict__ dhout, const float* __restrict__ hout,
                                                         const float* __restrict__ gates, const float* __restrict__ whh0,
                                                         const float* __restrict__ whh1, float* __restrict__ dgi,
                                                         float* __restrict__ dghn, int L, int NP, int N) {
    typedef GruV<NE> V;
    typedef typename V::T VT;
    constexpr int GFW = 32 * NE;
    __shared__ __attribute__((aligned(16))) float ds[2][3 * H][GFW];  // [dir][gate row][frame]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = wave >> 1, jt = wave & 1;
    const int h = lane >> 5, c = lane & 31;
    const float* whh = d ? whh1 : whh0;
    const int n0 = blockIdx.x * GFW;
    (void)N;

    // A fragments of W_hh^T: A[kk] = W_hh[row = 2*kk + h][unit = 32*jt + c], kk over the 192 gate rows
    float A[96];
#pragma unroll
    for (int kk = 0; kk < 96; ++kk) A[kk] = whh[(size_t)(2 * kk + h) * H + 32 * jt + c];

    float carry[16][NE];
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int e = 0; e < NE; ++e) carry[r][e] = 0.f;
    const size_t gs = (size_t)H * L * NP;

    for (int t = L - 1; t >= 0; --t) {
        const int pos = d ? (L - 1 - t) : t;
        const int ppos = d ? pos + 1 : pos - 1;   // position of h_{t-1}
        float dhz[16][NE];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * h;
            const size_t o = ((size_t)u * L + pos) * NP + n0 + NE * c;
            const VT dho = V::ld(dhout + (size_t)d * gs + o);
            const float* gb = gates + (size_t)d * 4 * gs + o;
            const VT rr = V::ld(gb);
            const VT zz = V::ld(gb + gs);
            const VT nn = V::ld(gb + 2 * gs);
            const VT gh = V::ld(gb + 3 * gs);
            VT hp;
#pragma unroll
            for (int e = 0; e < NE; ++e) V::set(hp, e, 0.f);
            if (t > 0) hp = V::ld(hout + (size_t)d * gs + ((size_t)u * L + ppos) * NP + n0 + NE * c);
            VT drp, dzp, dnp, dgn;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float z_ = V::get(zz, e), n_ = V::get(nn, e), r_ = V::get(rr, e);
                const float dh = V::get(dho, e) + carry[r][e];
                const float dn = dh * (1.f - z_);
                const float dnp_ = dn * (1.f - n_ * n_);
                V::set(dnp, e, dnp_);
                V::set(dzp, e, dh * (V::get(hp, e) - n_) * z_ * (1.f - z_));
                V::set(drp, e, dnp_ * V::get(gh, e) * r_ * (1.f - r_));
                V::set(dgn, e, dnp_ * r_);
                dhz[r][e] = dh * z_;
            }
            *(VT*)(&ds[d][u][NE * c]) = drp;
            *(VT*)(&ds[d][H + u][NE * c]) = dzp;
            *(VT*)(&ds[d][2 * H + u][NE * c]) = dgn;
            float* go = dgi + ((size_t)(d * 3 * H + u) * L + pos) * NP + n0 + NE * c;
            V::st(go, drp);
            V::st(go + (size_t)H * L * NP, dzp);
            V::st(go + (size_t)2 * H * L * NP, dnp);
            V::st(dghn + (size_t)d * gs + o, dgn);
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
                const VT b = *(const VT*)(&ds[d][2 * kk + h][NE * c]);
#pragma unroll
                for (int e = 0; e < NE; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[kk], V::get(b, e), acc[e], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int e = 0; e < NE; ++e) carry[r][e] = acc[e][r];
        __syncthreads();
    }
}

}  // namespace

// Measured (rocprofv3, 32,064 frames): forward 555 us with NE = 2 / 570 us with NE = 1 -- it moves 2.1 GB at 3.8 TB/s, the
// second workgroup has nothing to hide; backward 744 / 652 us.  Defaults: forward 2, backward 1; TRUNET_GRU_NE = 1 / 2 forces both.
static int gru_ne(int dflt) {
    static const int v = [] { const char* e = getenv("TRUNET_GRU_NE"); return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0; }();
    return v ? v : dflt;
}

extern "C" int trunet_gru_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* w_hh_rev,
                              const float* b_hh_rev, float* hout, float* gates, int Hh, int L, int NP, void* stream) {
    if (!gi || !w_hh || !b_hh || !w_hh_rev || !b_hh_rev || !hout || (NP % 128) || L <= 0) return TRUNET_EINVAL;
    if (Hh != H) return TRUNET_ENOTSUP;
    if (gru_ne(2) == 2)
        hipLaunchKernelGGL(gru_fwd_kernel<2>, dim3(NP / 64), dim3(256), 0, (hipStream_t)stream, gi, w_hh, b_hh, w_hh_rev,
                           b_hh_rev, hout, gates, L, NP);
    else
        hipLaunchKernelGGL(gru_fwd_kernel<1>, dim3(NP / 32), dim3(256), 0, (hipStream_t)stream, gi, w_hh, b_hh, w_hh_rev,
                           b_hh_rev, hout, gates, L, NP);
    return trunet_launch_status();
}

extern "C" int trunet_gru_bwd(const float* dhout, const float* hout, const float* gates, const float* w_hh,
                              const float* w_hh_rev, float* dgi, float* dghn, int Hh, int L, int NP, int N, void* stream) {
    if (!dhout || !hout || !gates || !w_hh || !w_hh_rev || !dgi || !dghn || (NP % 128) || L <= 0) return TRUNET_EINVAL;
    if (Hh != H) return TRUNET_ENOTSUP;
    if (gru_ne(1) == 2)
        hipLaunchKernelGGL(gru_bwd_kernel<2>, dim3(NP / 64), dim3(256), 0, (hipStream_t)stream, dhout, hout, gates, w_hh,
                           w_hh_rev, dgi, dghn, L, NP, N);
    else
        hipLaunchKernelGGL(gru_bwd_kernel<1>, dim3(NP / 32), dim3(256), 0, (hipStream_t)stream, dhout, hout, gates, w_hh,
                           w_hh_rev, dgi, dghn, L, NP, N);
    return trunet_launch_status();
}

// =====================================================================================
// TGRU (network.py:150 used over time): persistent recurrence over T time steps on sequence-major
// tensors [C][T][SP].  H = 128 hidden units, unidirectional.  A workgroup owns 32 sequences (columns) for all T steps:
// W_hh (384 x 128 floats = 192 KiB) lives in the registers of its 8 waves as MFMA A fragments -- wave (ut, kh) holds
// the three gates of unit tile ut (32 units) for K half kh (64 of the 128 h rows): 96 VGPRs -- h_{t-1} goes through
// LDS (B operand), the K halves are combined through LDS once per step, and the kh = 0 waves run the gate math on the
// MFMA C layout exactly like gru_fwd_kernel.  Replaces a host loop of T (GEMM, cell) launch pairs.
// =====================================================================================
namespace {

constexpr int TH = 128;    // hidden units
constexpr int TS = 32;     // sequences per workgroup

// Global rows as buffer resource (SGPR descriptor of a uniform base) + uniform SGPR row offset + one per-lane VGPR offset:
// sixteen rows x eight tensors of per-lane 64-bit addresses would not fit the register budget next to W_hh, and these
// intrinsics are tracked by the compiler (waits, hazards), unlike hand-written asm loads.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tg_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float tg_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void tg_store(__amdgpu_buffer_rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, 0);
}

__global__ __launch_bounds__(512, 2) void tgru_rec_fwd_kernel(const float* __restrict__ gi_all,
                                                              const float* __restrict__ whh,
                                                              const float* __restrict__ bhn, float* __restrict__ hs,
                                                              float* __restrict__ gates, int T, int SP) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][TH][TS];          // h_{t-1} / h_t, [unit][sequence]
    __shared__ __attribute__((aligned(16))) float part[4][3][16][64];       // K-half-1 partial sums per unit tile
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ut = wave & 3, kh = wave >> 2;
    const int hh = lane >> 5, c = lane & 31;
    const int s0 = blockIdx.x * TS;

    // A fragments: A[g][kk] = W_hh[g*128 + 32*ut + c][64*kh + 2*kk + hh]
    float A[3][32];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) A[g][kk] = whh[(size_t)(g * TH + 32 * ut + c) * TH + 64 * kh + 2 * kk + hh];
    // gi_all already carries b_ih + (b_hr, b_hz, 0): only b_hn must stay inside r * (W_hn h + b_hn)
    float bias_n[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = 32 * ut + (r & 3) + 8 * (r >> 2) + 4 * hh;
        bias_n[r] = (kh == 0) ? bhn[u] : 0.f;
    }
    for (int i = tid; i < TH * TS; i += 512) (&hbuf[0][0][0])[i] = 0.f;      // h_{-1} = 0
    float hprev[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) hprev[r] = 0.f;
    __syncthreads();

    const size_t grow = (size_t)T * SP;              // row stride of gi_all / gates planes
    const size_t hrow = (size_t)(T + 1) * SP;        // row stride of hs
    const int voff_g = (int)((4 * hh * grow + c) * sizeof(float));     // per-lane part of a gi / gates address
    const int voff_h = (int)((4 * hh * hrow + c) * sizeof(float));     // per-lane part of an hs address
    const int growb = (int)(grow * sizeof(float)), hrowb = (int)(hrow * sizeof(float));
    const int planeb = (int)((size_t)TH * grow * sizeof(float));       // gate g of gi_all starts g planes further
    const int gplaneb = planeb;                                        // gates planes [4][H][T][SP] have the same size
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        float gin[16];
        f32x16 acc[3];
        // kh = 0 waves seed the accumulators with gi (+ folded biases); gi_n stays aside, gh_n accumulates on b_hn alone
        if (kh == 0) {
            const __amdgpu_buffer_rsrc_t rg = tg_rsrc(gi_all + (size_t)(32 * ut) * grow + (size_t)t * SP + s0);   // uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = ((r & 3) + 8 * (r >> 2)) * growb;
                acc[0][r] = tg_load(rg, voff_g, ro);
                acc[1][r] = tg_load(rg, voff_g, ro + planeb);
                gin[r] = tg_load(rg, voff_g, ro + 2 * planeb);
                acc[2][r] = bias_n[r];
            }
        } else {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
        }
        if (t > 0) {
            const float* hb = &hbuf[cur][64 * kh][0];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const float b = hb[(2 * kk + hh) * TS + c];
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[g][kk], b, acc[g], 0, 0, 0);
            }
        }
        if (kh == 1) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) part[ut][g][r][lane] = acc[g][r];
        }
        __syncthreads();
        if (kh == 0) {
            float* hw = &hbuf[cur ^ 1][0][0];
            const __amdgpu_buffer_rsrc_t rh = tg_rsrc(hs + (size_t)(32 * ut) * hrow + (size_t)(t + 1) * SP + s0);
            const __amdgpu_buffer_rsrc_t rt = tg_rsrc(gates ? gates + (size_t)(32 * ut) * grow + (size_t)t * SP + s0 : hs);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = (r & 3) + 8 * (r >> 2);
                const int u = 32 * ut + ml + 4 * hh;
                const float ar = acc[0][r] + part[ut][0][r][lane];
                const float az = acc[1][r] + part[ut][1][r][lane];
                const float gh = acc[2][r] + part[ut][2][r][lane];
                const float rr = sigmoidf_(ar);
                const float zz = sigmoidf_(az);
                const float nn = tanhf_(fmaf(rr, gh, gin[r]));
                const float hn = fmaf(zz, hprev[r] - nn, nn);
                hprev[r] = hn;
                hw[u * TS + c] = hn;
                tg_store(rh, voff_h, ml * hrowb, hn);
                if (gates) {
                    const int ro = ml * growb;
                    tg_store(rt, voff_g, ro, rr);
                    tg_store(rt, voff_g, ro + gplaneb, zz);
                    tg_store(rt, voff_g, ro + 2 * gplaneb, nn);
                    tg_store(rt, voff_g, ro + 3 * gplaneb, gh);
                }
            }
        }
        __syncthreads();
    }
}

// Backward through time of the same recurrence, one persistent launch.  Per step t (T-1 .. 0), for the 32 sequences of the
// workgroup:  dh = dhs[:, t+1] + carry;  (drp, dzp, dnp) as in tgru_cell_bwd_kernel;  dgi_all / dgh_all rows of step t;
// carry' = dh z + W_hh^T dgh.  Wave (ut, kh) keeps W_hh^T[unit tile ut][K half kh of the 384 gate rows] as A fragments
// (96 VGPRs); dgh goes through LDS as the B operand; the kh = 0 waves own `carry` on the MFMA C layout and run the
// elementwise phase there.  The last step leaves dL/dh_{-1} unused.
__global__ __launch_bounds__(512, 2) void tgru_rec_bwd_kernel(const float* __restrict__ dhs, const float* __restrict__ hs,
                                                              const float* __restrict__ gates,
                                                              const float* __restrict__ whh, float* __restrict__ dgi_all,
                                                              float* __restrict__ dgh_all, int T, int SP, int S) {
    __shared__ __attribute__((aligned(16))) float dgb[3 * TH][TS];          // dgh of the current step, [gate row][sequence]
    __shared__ __attribute__((aligned(16))) float part[4][16][64];          // K-half-1 partial sums per unit tile
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ut = wave & 3, kh = wave >> 2;
    const int hh = lane >> 5, c = lane & 31;
    const int s0 = blockIdx.x * TS;

    // A fragments of W_hh^T: A[kk] = W_hh[192*kh + 2*kk + hh][32*ut + c]
    float A[96];
#pragma unroll
    for (int kk = 0; kk < 96; ++kk) A[kk] = whh[(size_t)(192 * kh + 2 * kk + hh) * TH + 32 * ut + c];
    float carry[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) carry[r] = 0.f;
    const size_t grow = (size_t)T * SP;
    const size_t hrow = (size_t)(T + 1) * SP;
    const int voff_g = (int)((4 * hh * grow + c) * sizeof(float));
    const int voff_h = (int)((4 * hh * hrow + c) * sizeof(float));
    const int growb = (int)(grow * sizeof(float)), hrowb = (int)(hrow * sizeof(float));
    const int planeb = (int)((size_t)TH * grow * sizeof(float));       // planes of gates / gate blocks of dgi, dgh
    const bool live = s0 + c < S;                    // padded sequences carry no gradient

    for (int t = T - 1; t >= 0; --t) {
        float dzd[16];                               // dh z: the direct path into h_{t-1}
        if (kh == 0) {
            const __amdgpu_buffer_rsrc_t rdh = tg_rsrc(dhs + (size_t)(32 * ut) * hrow + (size_t)(t + 1) * SP + s0);   // uniform bases
            const __amdgpu_buffer_rsrc_t rhp = tg_rsrc(hs + (size_t)(32 * ut) * hrow + (size_t)t * SP + s0);
            const __amdgpu_buffer_rsrc_t rgt = tg_rsrc(gates + (size_t)(32 * ut) * grow + (size_t)t * SP + s0);
            const __amdgpu_buffer_rsrc_t rgi = tg_rsrc(dgi_all + (size_t)(32 * ut) * grow + (size_t)t * SP + s0);
            const __amdgpu_buffer_rsrc_t rgh = tg_rsrc(dgh_all + (size_t)(32 * ut) * grow + (size_t)t * SP + s0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {            // four rows at a time: 24 loads in flight
                float v[4][6];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * q + i;
                    const int mg = ((r & 3) + 8 * (r >> 2)) * growb;
                    const int mh = ((r & 3) + 8 * (r >> 2)) * hrowb;
                    v[i][0] = tg_load(rdh, voff_h, mh);
                    v[i][1] = tg_load(rhp, voff_h, mh);
                    v[i][2] = tg_load(rgt, voff_g, mg);
                    v[i][3] = tg_load(rgt, voff_g, mg + planeb);
                    v[i][4] = tg_load(rgt, voff_g, mg + 2 * planeb);
                    v[i][5] = tg_load(rgt, voff_g, mg + 3 * planeb);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * q + i;
                    const int ml = (r & 3) + 8 * (r >> 2);
                    const int u = 32 * ut + ml + 4 * hh;
                    const float d = live ? v[i][0] + carry[r] : 0.f;
                    const float hp = v[i][1], rr = v[i][2], zz = v[i][3], nn = v[i][4], ghn = v[i][5];
                    const float dn = d * (1.f - zz);
                    const float dnp = dn * (1.f - nn * nn);
                    const float dzp = d * (hp - nn) * zz * (1.f - zz);
                    const float drp = dnp * ghn * rr * (1.f - rr);
                    const float dnr = dnp * rr;
                    dzd[r] = d * zz;
                    dgb[u][c] = drp;
                    dgb[TH + u][c] = dzp;
                    dgb[2 * TH + u][c] = dnr;
                    const int mg = ml * growb;
                    tg_store(rgi, voff_g, mg, drp);
                    tg_store(rgi, voff_g, mg + planeb, dzp);
                    tg_store(rgi, voff_g, mg + 2 * planeb, dnp);
                    tg_store(rgh, voff_g, mg, drp);
                    tg_store(rgh, voff_g, mg + planeb, dzp);
                    tg_store(rgh, voff_g, mg + 2 * planeb, dnr);
                }
            }
        }
        __syncthreads();                              // dgh of step t is in LDS
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (t > 0) {                                  // the gradient of h_{-1} is not needed
            const float* db = &dgb[192 * kh][0];
#pragma unroll
            for (int kk = 0; kk < 96; ++kk) {
                const float b = db[(2 * kk + hh) * TS + c];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[kk], b, acc, 0, 0, 0);
            }
            if (kh == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) part[ut][r][lane] = acc[r];
            }
        }
        __syncthreads();                              // partial sums visible; every read of dgb is done
        if (kh == 0 && t > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) carry[r] = dzd[r] + acc[r] + part[ut][r][lane];
        }
    }
}

}  // namespace

extern "C" int trunet_tgru_rec_bwd(const float* dhs, const float* hs, const float* gates, const float* w_hh, float* dgi_all,
                                   float* dgh_all, int H, int T, int SP, int S, void* stream) {
    if (!dhs || !hs || !gates || !w_hh || !dgi_all || !dgh_all || T <= 0 || SP <= 0 || (SP % TS) || S > SP) return TRUNET_EINVAL;
    if (H != TH || (size_t)4 * TH * T * SP * sizeof(float) >= ((size_t)1 << 31)) return TRUNET_ENOTSUP;   // 32-bit row offsets
    hipLaunchKernelGGL(tgru_rec_bwd_kernel, dim3(SP / TS), dim3(512), 0, (hipStream_t)stream, dhs, hs, gates, w_hh, dgi_all,
                       dgh_all, T, SP, S);
    return trunet_launch_status();
}

extern "C" int trunet_tgru_rec_fwd(const float* gi_all, const float* w_hh, const float* b_hn, float* hs, float* gates,
                                   int H, int T, int SP, void* stream) {
    if (!gi_all || !w_hh || !b_hn || !hs || T <= 0 || SP <= 0 || (SP % TS)) return TRUNET_EINVAL;
    if (H != TH || (size_t)4 * TH * T * SP * sizeof(float) >= ((size_t)1 << 31)) return TRUNET_ENOTSUP;   // 32-bit row offsets
    hipLaunchKernelGGL(tgru_rec_fwd_kernel, dim3(SP / TS), dim3(512), 0, (hipStream_t)stream, gi_all, w_hh, b_hn, hs, gates,
                       T, SP);
    return trunet_launch_status();
}
