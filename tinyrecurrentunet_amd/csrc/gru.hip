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
#include "common.hpp"

namespace {

constexpr int H = 64;
constexpr int GF = 64;  // frames per workgroup

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256, 1) void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ whh0,
                                                         const float* __restrict__ bhh0, const float* __restrict__ whh1,
                                                         const float* __restrict__ bhh1, float* __restrict__ hout,
                                                         float* __restrict__ gates, int L, int NP) {
    __shared__ __attribute__((aligned(16))) float hs[2][2][H][GF];  // [dir][buf][unit][frame]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = wave >> 1, jt = wave & 1;
    const int h = lane >> 5, c = lane & 31;
    const float* whh = d ? whh1 : whh0;
    const float* bhh = d ? bhh1 : bhh0;
    const int n0 = blockIdx.x * GF;

    // A fragments: A[g][kk] = W_hh[(g*64 + 32*jt + c)][2*kk + h]
    float A[3][32];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) A[g][kk] = whh[(size_t)(g * H + 32 * jt + c) * H + 2 * kk + h];

    float hprev[16][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { hprev[r][0] = 0.f; hprev[r][1] = 0.f; }

    for (int t = 0; t < L; ++t) {
        const int pos = d ? (L - 1 - t) : t;
        const int buf = t & 1;
        f32x16 acc[3][2];
        float gin[16][2];
        // seed accumulators with gi (+ b_hh); keep gi_n aside, gh_n accumulates on b_hn alone
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const f32x2 v = *(const f32x2*)(gi + ((size_t)(d * 3 * H + g * H + u) * L + pos) * NP + n0 + 2 * c);
                const float b = bhh[g * H + u];
                if (g < 2) { acc[g][0][r] = v[0] + b; acc[g][1][r] = v[1] + b; }
                else { gin[r][0] = v[0]; gin[r][1] = v[1]; acc[2][0][r] = b; acc[2][1][r] = b; }
            }
        }
        if (t > 0) {
            const float* hb = &hs[d][buf ^ 1][0][0];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const f32x2 b = *(const f32x2*)(hb + (2 * kk + h) * GF + 2 * c);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[g][kk], b[0], acc[g][0], 0, 0, 0);
                    acc[g][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[g][kk], b[1], acc[g][1], 0, 0, 0);
                }
            }
        }
        float* hw = &hs[d][buf][0][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * h;
            f32x2 rr, zz, nn, gh, hn;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                rr[e] = sigmoidf_(acc[0][e][r]);
                zz[e] = sigmoidf_(acc[1][e][r]);
                gh[e] = acc[2][e][r];
                nn[e] = tanhf(fmaf(rr[e], gh[e], gin[r][e]));
                hn[e] = fmaf(zz[e], hprev[r][e] - nn[e], nn[e]);
                hprev[r][e] = hn[e];
            }
            *(f32x2*)(hw + u * GF + 2 * c) = hn;
            const size_t o = ((size_t)(d * H + u) * L + pos) * NP + n0 + 2 * c;
            *(f32x2*)(hout + o) = hn;
            if (gates) {
                const size_t gs = (size_t)H * L * NP;   // one [H][L][NP] plane
                float* gb = gates + (size_t)d * 4 * gs + ((size_t)u * L + pos) * NP + n0 + 2 * c;
                *(f32x2*)(gb) = rr;
                *(f32x2*)(gb + gs) = zz;
                *(f32x2*)(gb + 2 * gs) = nn;
                *(f32x2*)(gb + 3 * gs) = gh;
            }
        }
        __syncthreads();
    }
}

// Backward through time.  Per step (reverse order of the forward):
//   dh = dhout[pos] + carry;  dn = dh (1-z);  dnp = dn (1-n^2);  dzp = dh (hprev - n) z (1-z);
//   drp = dnp * ghn * r (1-r);  dgi = (drp, dzp, dnp);  dgh = (drp, dzp, dnp*r);
//   carry = dh z + W_hh^T dgh.
__global__ __launch_bounds__(256, 1) void gru_bwd_kernel(const float* __restrict__ dhout, const float* __restrict__ hout,
                                                         const float* __restrict__ gates, const float* __restrict__ whh0,
                                                         const float* __restrict__ whh1, float* __restrict__ dgi,
                                                         float* __restrict__ dghn, int L, int NP, int N) {
    __shared__ __attribute__((aligned(16))) float ds[2][3 * H][GF];  // [dir][gate row][frame]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = wave >> 1, jt = wave & 1;
    const int h = lane >> 5, c = lane & 31;
    const float* whh = d ? whh1 : whh0;
    const int n0 = blockIdx.x * GF;
    (void)N;

    // A fragments of W_hh^T: A[kk] = W_hh[row = 2*kk + h][unit = 32*jt + c], kk over the 192 gate rows
    float A[96];
#pragma unroll
    for (int kk = 0; kk < 96; ++kk) A[kk] = whh[(size_t)(2 * kk + h) * H + 32 * jt + c];

    float carry[16][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { carry[r][0] = 0.f; carry[r][1] = 0.f; }
    const size_t gs = (size_t)H * L * NP;

    for (int t = L - 1; t >= 0; --t) {
        const int pos = d ? (L - 1 - t) : t;
        const int ppos = d ? pos + 1 : pos - 1;   // position of h_{t-1}
        float dhz[16][2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * h;
            const size_t o = ((size_t)u * L + pos) * NP + n0 + 2 * c;
            const f32x2 dho = *(const f32x2*)(dhout + (size_t)d * gs + o);
            const float* gb = gates + (size_t)d * 4 * gs + o;
            const f32x2 rr = *(const f32x2*)(gb);
            const f32x2 zz = *(const f32x2*)(gb + gs);
            const f32x2 nn = *(const f32x2*)(gb + 2 * gs);
            const f32x2 gh = *(const f32x2*)(gb + 3 * gs);
            f32x2 hp = {0.f, 0.f};
            if (t > 0) hp = *(const f32x2*)(hout + (size_t)d * gs + ((size_t)u * L + ppos) * NP + n0 + 2 * c);
            f32x2 drp, dzp, dnp, dgn;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float dh = dho[e] + carry[r][e];
                const float dn = dh * (1.f - zz[e]);
                dnp[e] = dn * (1.f - nn[e] * nn[e]);
                dzp[e] = dh * (hp[e] - nn[e]) * zz[e] * (1.f - zz[e]);
                drp[e] = dnp[e] * gh[e] * rr[e] * (1.f - rr[e]);
                dgn[e] = dnp[e] * rr[e];
                dhz[r][e] = dh * zz[e];
            }
            *(f32x2*)(&ds[d][u][2 * c]) = drp;
            *(f32x2*)(&ds[d][H + u][2 * c]) = dzp;
            *(f32x2*)(&ds[d][2 * H + u][2 * c]) = dgn;
            float* go = dgi + ((size_t)(d * 3 * H + u) * L + pos) * NP + n0 + 2 * c;
            *(f32x2*)(go) = drp;
            *(f32x2*)(go + (size_t)H * L * NP) = dzp;
            *(f32x2*)(go + (size_t)2 * H * L * NP) = dnp;
            *(f32x2*)(dghn + (size_t)d * gs + o) = dgn;
        }
        __syncthreads();
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = dhz[r][0]; acc[1][r] = dhz[r][1]; }
        if (t > 0) {
#pragma unroll
            for (int kk = 0; kk < 96; ++kk) {
                const f32x2 b = *(const f32x2*)(&ds[d][2 * kk + h][2 * c]);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[kk], b[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[kk], b[1], acc[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { carry[r][0] = acc[0][r]; carry[r][1] = acc[1][r]; }
        __syncthreads();
    }
}

}  // namespace

extern "C" int trunet_gru_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* w_hh_rev,
                              const float* b_hh_rev, float* hout, float* gates, int Hh, int L, int NP, void* stream) {
    if (!gi || !w_hh || !b_hh || !w_hh_rev || !b_hh_rev || !hout || (NP % 128) || L <= 0) return TRUNET_EINVAL;
    if (Hh != H) return TRUNET_ENOTSUP;
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(NP / GF), dim3(256), 0, (hipStream_t)stream, gi, w_hh, b_hh, w_hh_rev,
                       b_hh_rev, hout, gates, L, NP);
    return trunet_launch_status();
}

extern "C" int trunet_gru_bwd(const float* dhout, const float* hout, const float* gates, const float* w_hh,
                              const float* w_hh_rev, float* dgi, float* dghn, int Hh, int L, int NP, int N, void* stream) {
    if (!dhout || !hout || !gates || !w_hh || !w_hh_rev || !dgi || !dghn || (NP % 128) || L <= 0) return TRUNET_EINVAL;
    if (Hh != H) return TRUNET_ENOTSUP;
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(NP / GF), dim3(256), 0, (hipStream_t)stream, dhout, hout, gates, w_hh,
                       w_hh_rev, dgi, dghn, L, NP, N);
    return trunet_launch_status();
}
