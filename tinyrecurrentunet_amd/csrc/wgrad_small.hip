// Weight gradient of the thin layers (<= 8 output rows, or <= 4 source channels per tap): decoder.5's
// ConvTranspose1d(8->8, k5) and Conv1d(128->8), and encoder.0's Conv1d(4->64, k5).  On the MFMA kernel these are
// 3/4 (or 7/8) padding and pay a DMA-ring tile per 32 frames; here they are pure streams on the vector ALU.
//
//   dW[m][c][seg] = sum_{p, n<N} dz[m][p][n] * pro_seg(src_seg[c][q_seg(p)][n])
//
// grid = (roles, 256 partial images); a role = (MB rows of dz) x (one segment, or all taps of one small tensor) x
// (CB of its channels): every thread keeps its accumulators in registers, walks (position, 256-frame chunk) items with 16-byte loads (frames are contiguous), and
// the block reduces its accumulators once at the end into its part of the partial image (trunet_reduce_partials
// sums the images exactly as for conv_wgrad_kernel).
#include <cstdlib>
#include "common.hpp"

namespace {

constexpr int WS_GRID = TRUNET_NUM_CU;

// NS = segments per role: 1, or TRUNET_MAX_SEG = every segment (the taps of one small tensor: dz is read once)
template <int MB, int CB, int NS>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const trunet_wgrad_args a, const int ngm, const int ngc_max) {
    __shared__ float red[4][NS * MB * CB + MB];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // role -> (m group, channel group, first segment)
    int role = blockIdx.x;          // roles vary fastest: blocks that run together re-read the same rows from L2
    const int mg = role % ngm; role /= ngm;
    const int cg = role % ngc_max;
    const int sfirst = (NS == 1) ? role / ngc_max : 0;
    const int ns = (NS == 1) ? 1 : a.nseg;
    const int m0 = mg * MB, c0 = cg * CB;
    if (c0 >= a.seg[sfirst].nchan) return;              // this segment has fewer channel groups (uniform)
    const bool two = a.a_mode == TRUNET_PRO_BNBWD;
    const bool bias_role = (sfirst == 0 && cg == 0);

    float ka[MB], kb[MB], kc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int m = m0 + i;
        const bool ok = m < a.M;
        const int ch = min(m, a.M - 1) + a.a_m_off;
        ka[i] = ok ? (two ? a.ac0[ch] : 1.f) : 0.f;
        kb[i] = (ok && two) ? a.ac1[ch] : 0.f;
        kc[i] = (ok && two) ? a.ac2[ch] : 0.f;
    }
    // prologue coefficients: per channel, taken from the first segment of the role (taps share their tensor)
    float s0[CB], s1[CB], slo[CB];
    {
        const trunet_seg& sg = a.seg[sfirst];
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int ci = min(c0 + j, sg.nchan - 1);
            const bool on = sg.mode == TRUNET_PRO_BNRELU;
            s0[j] = on ? sg.c0[ci] : 1.f;
            s1[j] = on ? sg.c1[ci] : 0.f;
            slo[j] = on ? 0.f : -3.0e38f;
        }
    }
    float acc[NS][MB][CB], bsum[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int j = 0; j < CB; ++j) acc[k][i][j] = 0.f;
    }

    const int nch = a.NP / 256;
    const int items = a.P * nch;
    const size_t dstr = (size_t)a.a_L * a.NP;           // dz channel stride
    for (int it = blockIdx.y * 4 + wave; it < items; it += gridDim.y * 4) {
        const int pi = it / nch;
        const int p = a.p_begin + pi;
        const int n = (it - pi * nch) * 256 + 4 * lane;
        bool valid[NS];
        int qs[NS];
        bool any = bias_role;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const trunet_seg& sg = a.seg[min(sfirst + k, a.nseg - 1)];
            const int qn = p * sg.pos_mul + sg.pos_off;
            qs[k] = qn / sg.pos_div;
            valid[k] = (k < ns) && (qn >= 0) && (qn - qs[k] * sg.pos_div == 0) && (qs[k] < sg.L);
            any = any || valid[k];
        }
        if (!any) continue;                             // uniform
        const float* pdz = a.a0 + ((size_t)(min(m0, a.M - 1) + a.a_m_off) * a.a_L + p + a.a_pos_off) * a.NP + n;
        const float* pz1 = two ? a.a1 + (pdz - a.a0) : pdz;
        f32x4 dz[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const size_t o = (size_t)min(i, a.M - 1 - min(m0, a.M - 1)) * dstr;
#ifdef TRUNET_THIN_NT
            dz[i] = __builtin_nontemporal_load((const f32x4*)(pdz + o));
#else
            dz[i] = *(const f32x4*)(pdz + o);
#endif
            if (two) {
                const f32x4 zz = *(const f32x4*)(pz1 + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) dz[i][e] = fmaf(ka[i], dz[i][e], fmaf(kb[i], zz[e], kc[i]));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) dz[i][e] *= ka[i];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e >= a.N) dz[i][e] = 0.f;
        }
        if (bias_role) {
#pragma unroll
            for (int i = 0; i < MB; ++i) bsum[i] += (dz[i][0] + dz[i][1]) + (dz[i][2] + dz[i][3]);
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (valid[k]) {
                const trunet_seg& sg = a.seg[min(sfirst + k, a.nseg - 1)];
                const size_t sstr = (size_t)sg.L * a.NP;        // source channel stride
                const float* psrc = sg.src0 + ((size_t)min(c0, sg.nchan - 1) * sg.L + qs[k]) * a.NP + n;
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    f32x4 v = *(const f32x4*)(psrc + (size_t)min(j, sg.nchan - 1 - min(c0, sg.nchan - 1)) * sstr);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], s0[j], s1[j]), slo[j]);
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[k][i][j] = fmaf(dz[i][e], v[e], acc[k][i][j]);
                }
            }
        }
    }
    // ---- block reduction (4 waves x 64 lanes) and this block's entries of the partial image
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                const float v = wave_sum(acc[k][i][j]);
                if (lane == 0) red[wave][(k * MB + i) * CB + j] = v;
            }
        const float b = wave_sum(bsum[i]);
        if (lane == 0) red[wave][NS * MB * CB + i] = b;
    }
    __syncthreads();
    float* img = a.w_partials + (size_t)blockIdx.y * a.w_numel;
    for (int idx = tid; idx < NS * MB * CB + MB; idx += 256) {
        const float v = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
        if (idx < NS * MB * CB) {
            const int k = idx / (MB * CB), r = idx - k * (MB * CB);
            const int i = r / CB, j = r - i * CB;
            const int m = m0 + i, ci = c0 + j;
            if (k < ns) {
                const trunet_seg& sg = a.seg[sfirst + k];
                if (m < a.M && ci < sg.nchan)
                    img[(size_t)(m + a.w_m_off) * a.ldw_m + (size_t)ci * a.ldw_c + sg.woff] = v;
            }
        } else if (bias_role && a.b_partials) {
            const int m = m0 + (idx - NS * MB * CB);
            if (m < a.M) a.b_partials[(size_t)blockIdx.y * a.b_stride + a.b_off + m] = v;
        }
    }
}

// ---- encoder.0's Conv1d(C_in <= 4 -> 64, k = NS taps, stride 2) (network.py:13), round 3:
//   dW[m][c][k] = sum_{p, n < N} dz[m][p][n] x[c][p mul + off_k][n],   db[m] = sum dz[m][p][n]
// The launch is a pure stream of dz (64 x 128 x N floats = 1.05 GB at configs[1]); wgrad_small_kernel<4, 4, 5> gave every
// 4-row group its own blocks and one (position, 256 frames) item per wave: 24 loads per item consumed as soon as they were
// issued, 16 copies of x through the L2, an integer division per item -- 0.67 ms = 1.6 TB/s.
// Here a block is 4 waves, ONE per SIMD (up to 512 registers each): wave w owns dz rows [32 rh + 8 w, + 8) -- rh = row half
// of the block -- and all C_in x NS columns (160 accumulators); a block walks a contiguous range of (frame chunk, position)
// items, positions fastest, all four waves on the same item (the item's x rows come from the L1 three times out of four),
// and the loads of item t + 1 (8 dz rows, C_in x NS x rows: 28 x 16 bytes per lane) are in flight in a second register set
// while item t's 640 FMAs per lane run.  Every dz row is read exactly once.
template <int NS>
__global__ __launch_bounds__(256, 1) void wgrad_first_kernel(const trunet_wgrad_args a) {
    __shared__ float red[4][8 * 4 * NS + 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rh = blockIdx.x & 1, img_i = blockIdx.x >> 1, nimg = gridDim.x >> 1;
    const int m0 = 32 * rh + 8 * wave;
    const trunet_seg& s0 = a.seg[0];
    const int C = s0.nchan;                               // <= 4 (host-checked), same for every tap
    float acc[8][4][NS], bsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < NS; ++k) acc[i][c][k] = 0.f;
    }
    int offk[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) offk[k] = a.seg[k].pos_off;
    const int mul = s0.pos_mul, SL = s0.L;
    const int nch = a.NP / 256;
    const int items = a.P * nch;
    const int i0 = (int)(((long long)img_i * items) / nimg);
    const int i1 = (int)(((long long)(img_i + 1) * items) / nimg);
    const size_t dstr = (size_t)a.a_L * a.NP, sstr = (size_t)SL * a.NP;
    const bool rows_ok = m0 < a.M;                        // M is a multiple of 8 (host-checked)
    // software pipeline: the 8 dz rows of item t + 1 are requested (second register set) before item t's FMAs start, and
    // the x rows of channel c for item t + 1 replace those of item t as soon as channel c's FMAs are done
    f32x4 xv[4][NS];
    auto load_dz = [&](f32x4 (&dz)[8], int chunk, int pi) __attribute__((always_inline)) {
        const float* pdz = a.a0 + ((size_t)(m0 + a.a_m_off) * a.a_L + a.p_begin + pi + a.a_pos_off) * a.NP + chunk * 256 + 4 * lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) dz[i] = *(const f32x4*)(pdz + (size_t)i * dstr);
    };
    // (every load below is UNCONDITIONAL: a load inside a uniform branch makes hipcc's s_waitcnt pass give up counting at
    // the merge point and drain the whole queue -- vmcnt(0) in front of every channel -- which serialises the pipeline;
    // out-of-range taps read a clamped row and are zeroed afterwards, the requests past the last item repeat it)
    auto load_x = [&](int c, int chunk, int pi) __attribute__((always_inline)) {
        const int p = a.p_begin + pi;
        const float* px = s0.src0 + (size_t)min(c, C - 1) * sstr + chunk * 256 + 4 * lane;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int q = p * mul + offk[k];
            xv[c][k] = *(const f32x4*)(px + (size_t)min(max(q, 0), SL - 1) * a.NP);
        }
    };
    auto fix_x = [&](int c, int pi) __attribute__((always_inline)) {       // conv padding / absent channels read as zero
        const int p = a.p_begin + pi;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int q = p * mul + offk[k];
            if (!(c < C && q >= 0 && q < SL)) xv[c][k] = f32x4{0.f, 0.f, 0.f, 0.f};    // uniform, rare
        }
    };
    auto consume = [&](f32x4 (&dz)[8], int chunk, int pi, int nchunk, int npi) __attribute__((always_inline)) {
        if (chunk * 256 + 256 > a.N) {                    // uniform: only the last chunk holds frames >= N
            const int n = chunk * 256 + 4 * lane;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e >= a.N) dz[i][e] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) bsum[i] += (dz[i][0] + dz[i][1]) + (dz[i][2] + dz[i][3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            fix_x(c, pi);
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][c][k] = fmaf(dz[i][e], xv[c][k][e], acc[i][c][k]);
            __builtin_amdgcn_sched_barrier(0);
            load_x(c, nchunk, npi);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (rows_ok && i0 < i1) {
        // dz rows of items t + 1 and t + 2 in flight while item t computes (three register sets): with one item ahead a CU
        // had 32 KB of the HBM stream in flight and the kernel sat at 2.9 TB/s
        f32x4 dzA[8], dzB[8], dzC[8];
        int chunk = i0 / a.P, pi = i0 - chunk * a.P;
        auto next = [&](int& ch, int& pp, bool more) __attribute__((always_inline)) {
            if (more && ++pp == a.P) { pp = 0; ++ch; }     // past the last item: stay on it (a redundant, harmless request)
        };
        int c1 = chunk, p1 = pi;
        next(c1, p1, i0 + 1 < i1);
        load_dz(dzA, chunk, pi);
        load_dz(dzB, c1, p1);
#pragma unroll
        for (int c = 0; c < 4; ++c) load_x(c, chunk, pi);
        for (int it = i0; it < i1; it += 3) {
            // items it (chunk, pi) in dzA, it + 1 (c1, p1) in dzB; it + 2 .. it + 4 are requested below
            int c2 = c1, p2 = p1;
            next(c2, p2, it + 2 < i1);
            int c3 = c2, p3 = p2;
            next(c3, p3, it + 3 < i1);
            int c4 = c3, p4 = p3;
            next(c4, p4, it + 4 < i1);
            load_dz(dzC, c2, p2);
            consume(dzA, chunk, pi, c1, p1);
            load_dz(dzA, c3, p3);
            if (it + 1 < i1) consume(dzB, c1, p1, c2, p2);
            load_dz(dzB, c4, p4);
            if (it + 2 < i1) consume(dzC, c2, p2, c3, p3);
            chunk = c3; pi = p3; c1 = c4; p1 = p4;
        }
    }
    // ---- this block's rows of partial image img_i: wave w writes its 8 rows
    float* img = a.w_partials + (size_t)img_i * a.w_numel;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const float v = wave_sum(acc[i][c][k]);
                if (lane == 0) red[wave][(i * 4 + c) * NS + k] = v;
            }
        const float b = wave_sum(bsum[i]);
        if (lane == 0) red[wave][8 * 4 * NS + i] = b;
    }
    __syncthreads();
    if (rows_ok) {
        for (int idx = lane; idx < 8 * 4 * NS + 8; idx += 64) {
            const float v = red[wave][idx];
            if (idx < 8 * 4 * NS) {
                const int i = idx / (4 * NS), r = idx - i * (4 * NS), c = r / NS, k = r - c * NS;
                if (c < C)
                    img[(size_t)(m0 + i + a.w_m_off) * a.ldw_m + (size_t)c * a.ldw_c + a.seg[k].woff] = v;
            } else if (a.b_partials) {
                a.b_partials[(size_t)img_i * a.b_stride + a.b_off + m0 + (idx - 8 * 4 * NS)] = v;
            }
        }
    }
}

// ---- decoder.5's ConvTranspose1d(8 -> 8, k = NS taps, stride 2) (network.py:109), round 3:
//   dW[ci][co][k] = sum_{p, n < N} dz[co][p][n] a[ci][(p + pad - k) / 2][n]   (taps with p + pad - k even),  db[co] = sum dz
// a = max(c0 z + c1, 0) of the 8-channel pointwise output.  The launch reads dz (8 x 257 rows) and z (8 x 128 rows) once:
// 12 KB per frame, 0.39 GB at configs[1]; wgrad_small_kernel<8, 8, 1> gave every tap its own blocks (dz through the L2 five
// times, 16 loads per item consumed at once): 0.355 ms = 1.1 TB/s.  Same recipe as wgrad_first_kernel: a block of 4 waves,
// one per SIMD, walks a contiguous range of (frame chunk, position) items, all waves on the same item; wave w owns source
// channels {2 w, 2 w + 1} x all 8 dz rows x all taps (80 accumulators); the next item's dz and source rows are requested
// into a second register set before the current item's FMAs; loads are unconditional (clamped rows), invalid taps skip
// their FMAs (uniform).
template <int NS>
__global__ __launch_bounds__(256, 1) void wgrad_last_kernel(const trunet_wgrad_args a) {
    __shared__ float red[4][8 * 2 * NS + 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = 2 * wave;
    const trunet_seg& s0 = a.seg[0];
    const bool on = s0.mode == TRUNET_PRO_BNRELU;
    float k0[2], k1[2], klo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = min(c0 + j, s0.nchan - 1);
        k0[j] = on ? s0.c0[ci] : 1.f;
        k1[j] = on ? s0.c1[ci] : 0.f;
        klo[j] = on ? 0.f : -3.0e38f;
    }
    float acc[8][2][NS], bsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < NS; ++k) acc[i][j][k] = 0.f;
    }
    int offk[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) offk[k] = a.seg[k].pos_off;
    const int SL = s0.L;
    const int nch = a.NP / 256;
    const int items = a.P * nch;
    const int i0 = (int)(((long long)blockIdx.x * items) / gridDim.x);
    const int i1 = (int)(((long long)(blockIdx.x + 1) * items) / gridDim.x);
    const size_t dstr = (size_t)a.a_L * a.NP, sstr = (size_t)SL * a.NP;
    struct Item { f32x4 dz[8]; f32x4 sv[2][NS]; };
    auto request = [&](Item& r, int chunk, int pi) __attribute__((always_inline)) {
        const int p = a.p_begin + pi;
        const int n = chunk * 256 + 4 * lane;
        const float* pdz = a.a0 + ((size_t)a.a_m_off * a.a_L + p + a.a_pos_off) * a.NP + n;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.dz[i] = *(const f32x4*)(pdz + (size_t)min(i, a.M - 1) * dstr);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int q = (p + offk[k]) >> 1;
                r.sv[j][k] = *(const f32x4*)(s0.src0 + (size_t)min(c0 + j, s0.nchan - 1) * sstr +
                                              (size_t)min(max(q, 0), SL - 1) * a.NP + n);
            }
    };
    auto consume = [&](Item& r, int chunk, int pi) __attribute__((always_inline)) {
        const int p = a.p_begin + pi;
        if (chunk * 256 + 256 > a.N) {                    // uniform: only the last chunk holds frames >= N
            const int n = chunk * 256 + 4 * lane;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e >= a.N) r.dz[i][e] = 0.f;
        }
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) bsum[i] += (r.dz[i][0] + r.dz[i][1]) + (r.dz[i][2] + r.dz[i][3]);
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int qn = p + offk[k];
            const int q = qn >> 1;
            if (qn >= 0 && !(qn & 1) && q < SL) {          // uniform: the taps of this position's parity
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 v = r.sv[j][k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], k0[j], k1[j]), klo[j]);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][k] = fmaf(r.dz[i][e], v[e], acc[i][j][k]);
                }
            }
        }
    };
    if (i0 < i1) {
        Item A_, B_;
        int chunk = i0 / a.P, pi = i0 - chunk * a.P;
        auto next = [&](int& ch, int& pp, bool more) __attribute__((always_inline)) {
            if (more && ++pp == a.P) { pp = 0; ++ch; }     // past the last item: stay on it (a redundant, harmless request)
        };
        request(A_, chunk, pi);
        for (int it = i0; it < i1; it += 2) {
            int c1 = chunk, p1 = pi;
            next(c1, p1, it + 1 < i1);
            int c2 = c1, p2 = p1;
            next(c2, p2, it + 2 < i1);
            request(B_, c1, p1);
            consume(A_, chunk, pi);
            request(A_, c2, p2);
            if (it + 1 < i1) consume(B_, c1, p1);
            chunk = c2; pi = p2;
        }
    }
    float* img = a.w_partials + (size_t)blockIdx.x * a.w_numel;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const float v = wave_sum(acc[i][j][k]);
                if (lane == 0) red[wave][(i * 2 + j) * NS + k] = v;
            }
        const float b = wave_sum(bsum[i]);
        if (lane == 0) red[wave][8 * 2 * NS + i] = b;
    }
    __syncthreads();
    for (int idx = lane; idx < 8 * 2 * NS + 8; idx += 64) {
        const float v = red[wave][idx];
        if (idx < 8 * 2 * NS) {
            const int i = idx / (2 * NS), r = idx - i * (2 * NS), j = r / NS, k = r - j * NS;
            if (i < a.M && c0 + j < s0.nchan)
                img[(size_t)(i + a.w_m_off) * a.ldw_m + (size_t)(c0 + j) * a.ldw_c + a.seg[k].woff] = v;
        } else if (wave == 0 && a.b_partials) {
            const int i = idx - 8 * 2 * NS;
            if (i < a.M) a.b_partials[(size_t)blockIdx.x * a.b_stride + a.b_off + i] = v;
        }
    }
}

}  // namespace

// called by trunet_conv_wgrad (gemm_conv.hip) for the thin shapes; returns TRUNET_ENOTSUP when the shape is not thin
int trunet_launch_wgrad_small(const trunet_wgrad_args* h, hipStream_t st) {
    int maxc = 0;
    bool same = true;       // every segment is a tap of the same tensor with the same prologue
    for (int s = 0; s < h->nseg; ++s) {
        maxc = h->seg[s].nchan > maxc ? h->seg[s].nchan : maxc;
        same = same && h->seg[s].src0 == h->seg[0].src0 && h->seg[s].nchan == h->seg[0].nchan &&
               h->seg[s].mode == h->seg[0].mode && h->seg[s].c0 == h->seg[0].c0 && h->seg[s].c1 == h->seg[0].c1;
    }
    static const bool first_ok = !(getenv("TRUNET_WGRAD_FIRST") && getenv("TRUNET_WGRAD_FIRST")[0] == '0');
    if (first_ok && maxc <= 4 && same && h->nseg == 5 && h->M <= 64 && (h->M % 8) == 0 && h->a_mode == TRUNET_PRO_NONE &&
        h->seg[0].mode == TRUNET_PRO_NONE && h->seg[0].pos_div == 1 && (h->NP % 256) == 0) {
        bool taps = true;                           // the five segments are the taps of ONE strided conv
        for (int s = 1; s < h->nseg; ++s)
            taps = taps && h->seg[s].pos_mul == h->seg[0].pos_mul && h->seg[s].pos_div == 1 && h->seg[s].L == h->seg[0].L;
        if (taps) {
            hipLaunchKernelGGL((wgrad_first_kernel<5>), dim3(2 * WS_GRID), dim3(256), 0, st, *h);
            return trunet_launch_status();
        }
    }
    if (maxc <= 4 && same) {                    // encoder.0 Conv1d(4 -> 64, k5): 4 rows of dz x all taps per role
        const int ngm = (h->M + 3) / 4;
        hipLaunchKernelGGL((wgrad_small_kernel<4, 4, TRUNET_MAX_SEG>), dim3(ngm, WS_GRID), dim3(256), 0, st, *h, ngm, 1);
        return trunet_launch_status();
    }
    static const bool last_ok = !(getenv("TRUNET_WGRAD_LAST") && getenv("TRUNET_WGRAD_LAST")[0] == '0');
    if (last_ok && h->M <= 8 && maxc <= 8 && same && h->nseg == 5 && h->a_mode == TRUNET_PRO_NONE && (h->NP % 256) == 0) {
        bool taps = true;                           // the five segments are the taps of ONE stride-2 transposed conv
        for (int s = 0; s < h->nseg; ++s)
            taps = taps && h->seg[s].pos_mul == 1 && h->seg[s].pos_div == 2 && h->seg[s].L == h->seg[0].L &&
                   h->seg[s].mode != TRUNET_PRO_BNBWD;
        if (taps) {
            hipLaunchKernelGGL((wgrad_last_kernel<5>), dim3(WS_GRID), dim3(256), 0, st, *h);
            return trunet_launch_status();
        }
    }
    if (h->M <= 8 && maxc <= 8) {               // decoder.5 ConvTranspose1d(8 -> 8, k5): one tap per role
        hipLaunchKernelGGL((wgrad_small_kernel<8, 8, 1>), dim3(h->nseg, WS_GRID), dim3(256), 0, st, *h, 1, 1);
        return trunet_launch_status();
    }
    return TRUNET_ENOTSUP;
}

// =====================================================================================
// Fused backward of a THIN pointwise layer (<= 8 output rows: decoder.5's Conv1d(128 -> 8) + BatchNorm): the same
// contract as pw_bwd_kernel (trunet_pw_bwd), on the vector ALU.  A role = 8 channels of one source segment; a block
// holds 16 roles x 16 lanes and walks (position, 64-frame chunk) items, 4 frames per lane: the 16 roles read the same
// dy / z rows (one fetch, served 16 times from L1), every role forms dz = ca dy + cb z + cc, accumulates its dW[8][8]
// in registers and writes its 8 rows of the data gradient g = W^T dz (+ previous content) (* ReLU mask) with their
// BatchNorm-backward sums.  One pass over (dy, z, sources) instead of three launches that each re-read dy, z.
// grid = (ceil(roles / 16), 256 partial images).
// =====================================================================================
namespace {

// RPB roles per block, LPR = 256 / RPB lanes per role (4 frames each).  16 x 16 is the original shape (one block covers
// all 128 channels: 256 blocks for 256 partial images = 4 waves per CU, measured 2 ms slower than three separate launches);
// 4 x 64 gives 4 blocks per partial image -- they write disjoint columns of it -- and 1 KiB row pieces per role.
template <int RPB>
__global__ __launch_bounds__(256) void pw_bwd_small_kernel(const trunet_pwbwd_args A, const int ngc_max, const int nroles) {
    constexpr int LPR = 256 / RPB;
    const trunet_wgrad_args& a = A.w;
    __shared__ float Ws[RPB][64];            // per role: W[m][c0 + j]
    const int tid = threadIdx.x;
    const int l16 = tid % LPR;               // lane inside the role
    const int rl = tid / LPR;                // role slot inside the block
    const int role = blockIdx.x * RPB + rl;
    const bool has_role = role < nroles;
    const int cg = has_role ? role % ngc_max : 0, s = has_role ? role / ngc_max : 0;
    const trunet_seg& sg = a.seg[s];
    const trunet_dgrad_out& dg = A.dg[s];
    const int c0 = cg * 8;
    const bool act_role = has_role && c0 < sg.nchan;
    const bool bias_role = (role == 0);
    const int fl = dg.flags;
    for (int idx = l16; idx < 64; idx += LPR) {
        const int m = idx >> 3, j = idx & 7;
        Ws[rl][idx] = (act_role && m < a.M) ? A.W[(size_t)(m + a.w_m_off) * a.ldw_m + (size_t)(c0 + j) * a.ldw_c + sg.woff] : 0.f;
    }
    __syncthreads();
    float accW[8][8], bsum[8], s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bsum[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) accW[i][j] = 0.f;
    }
    const bool on = sg.mode == TRUNET_PRO_BNRELU;
    const int nch = a.NP / (4 * LPR);
    const int items = a.P * nch;
    const size_t dstr = (size_t)a.a_L * a.NP;
    const size_t sstr = (size_t)sg.L * a.NP;
    for (int it = blockIdx.y; it < items; it += gridDim.y) {
        const int pi = it / nch;
        const int p = a.p_begin + pi;
        const int n = (it - pi * nch) * (4 * LPR) + 4 * l16;
        const int q = p + sg.pos_off;
        const bool valid = act_role && q >= 0 && q < sg.L;
        if (!valid && !bias_role) continue;
        f32x4 dz[8];
        {
            const float* pdy = a.a0 + ((size_t)a.a_m_off * a.a_L + p + a.a_pos_off) * a.NP + n;
            const float* pz = a.a1 + (pdy - a.a0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int mm = min(i, a.M - 1);
                const f32x4 dv = *(const f32x4*)(pdy + (size_t)mm * dstr);
                const f32x4 zv = *(const f32x4*)(pz + (size_t)mm * dstr);
                const float ka = (i < a.M) ? a.ac0[mm + a.a_m_off] : 0.f;
                const float kb = (i < a.M) ? a.ac1[mm + a.a_m_off] : 0.f;
                const float kc = (i < a.M) ? a.ac2[mm + a.a_m_off] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) dz[i][e] = (n + e < a.N) ? fmaf(ka, dv[e], fmaf(kb, zv[e], kc)) : 0.f;
            }
        }
        if (bias_role) {
#pragma unroll
            for (int i = 0; i < 8; ++i) bsum[i] += (dz[i][0] + dz[i][1]) + (dz[i][2] + dz[i][3]);
        }
        if (!valid) continue;
        const float* psrc = sg.src0 + ((size_t)c0 * sg.L + q) * a.NP + n;
        float* pout = dg.out + ((size_t)c0 * sg.L + q) * a.NP + n;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 zs = *(const f32x4*)(psrc + (size_t)j * sstr);
            const float sc = on ? sg.c0[c0 + j] : 1.f, sh = on ? sg.c1[c0 + j] : 0.f;
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (fl & TRUNET_DG_ACCUM) g = *(const f32x4*)(pout + (size_t)j * sstr);
            f32x4 act;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pre = fmaf(zs[e], sc, sh);
                act[e] = on ? fmaxf(pre, 0.f) : pre;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float w = Ws[rl][i * 8 + j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    accW[i][j] = fmaf(dz[i][e], act[e], accW[i][j]);
                    g[e] = fmaf(w, dz[i][e], g[e]);
                }
            }
            if (fl & TRUNET_DG_MASK) {
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = (fmaf(zs[e], sc, sh) > 0.f) ? g[e] : 0.f;
            }
            *(f32x4*)(pout + (size_t)j * sstr) = g;
            if (fl & TRUNET_DG_STATS) {
                const float mu = dg.e2[c0 + j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = (n + e < a.N) ? g[e] : 0.f;
                    s1[j] += x;
                    s2[j] = fmaf(x, zs[e] - mu, s2[j]);
                }
            }
        }
    }
    // ---- reduce over the 16 lanes of the role, then this role's entries of the partial image
    auto sum16 = [](float v) {                      // over the LPR lanes of the role
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPR);
        return v;
    };
    float* img = a.w_partials + (size_t)blockIdx.y * a.w_numel;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = sum16(accW[i][j]);
            if (l16 == 0 && act_role && i < a.M && c0 + j < sg.nchan)
                img[(size_t)(i + a.w_m_off) * a.ldw_m + (size_t)(c0 + j) * a.ldw_c + sg.woff] = v;
        }
        const float b = sum16(bsum[i]);
        if (l16 == 0 && bias_role && a.b_partials && i < a.M)
            a.b_partials[(size_t)blockIdx.y * a.b_stride + a.b_off + i] = b;
        const float t1 = sum16(s1[i]);
        const float t2 = sum16(s2[i]);
        if (l16 == 0 && act_role && (fl & TRUNET_DG_STATS) && c0 + i < sg.nchan) {
            float* pp = dg.partials + ((size_t)blockIdx.y * sg.nchan + c0 + i) * 2;
            pp[0] = t1;
            pp[1] = t2;
        }
    }
}

}  // namespace

// called by trunet_pw_bwd (pw_bwd.hip) when the layer has <= 8 output rows; arguments already validated there
int trunet_launch_pw_bwd_small(const trunet_pwbwd_args* H, hipStream_t st) {
    const trunet_wgrad_args* h = &H->w;
    int maxc = 0;
    for (int s = 0; s < h->nseg; ++s) {
        if (h->seg[s].nchan % 8) return TRUNET_ENOTSUP;
        maxc = h->seg[s].nchan > maxc ? h->seg[s].nchan : maxc;
    }
    if (h->NP % 64) return TRUNET_EINVAL;
    const int ngc = maxc / 8;
    const int nroles = h->nseg * ngc;
    if (h->NP % 256 == 0)
        hipLaunchKernelGGL(pw_bwd_small_kernel<4>, dim3((nroles + 3) / 4, WS_GRID), dim3(256), 0, st, *H, ngc, nroles);
    else
        hipLaunchKernelGGL(pw_bwd_small_kernel<16>, dim3((nroles + 15) / 16, WS_GRID), dim3(256), 0, st, *H, ngc, nroles);
    return trunet_launch_status();
}
