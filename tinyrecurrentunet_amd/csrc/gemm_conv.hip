// Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32), frames-last layout.
//
//   out[m][p][n] = epi( sum_seg sum_c A_seg(m,c) * pro_seg(src_seg[c][q_seg(p)][n]) )
//
// One launch = one layer (Conv1d k=1 with one or two concatenated sources, ConvTranspose1d as a
// gather over taps, or the data gradient of either).  Frames n are the GEMM's column axis, so a
// conv tap / stride / pad / crop is a whole-row offset and every global access is a 512-byte
// contiguous row segment.
//
// Workgroup = 4 waves.  The weight block A (MB = 32*RS rows x all K) is loaded ONCE per workgroup
// into LDS in MFMA-fragment order; workgroups are persistent over (position p, 128-frame tile)
// pairs.  Per tile the K rows are staged 32 at a time: global -> registers (prefetch of the next
// chunk overlaps the MFMAs of the current one) -> prologue (BN+ReLU or BN-backward) -> LDS.
// Wave w owns row slice (w % RS) and column group (w / RS): 32 rows x 32*RS frames = RS accumulators.
// Epilogue: bias / ReLU-mask / accumulate, vector stores, per-channel statistics kept in registers
// across all tiles and written once per workgroup (deterministic two-stage reduction).
#include "common.hpp"

namespace {

constexpr int NT = TRUNET_TILE_FRAMES;  // frames per tile
constexpr int KC = 32;                  // K rows per staged chunk

template <int CT>
struct BVec;
template <>
struct BVec<4> { typedef f32x4 type; };
template <>
struct BVec<2> { typedef f32x2 type; };
template <>
struct BVec<1> { typedef float type; };

template <int CT>
__device__ __forceinline__ float vget(const typename BVec<CT>::type& v, int i) {
    if constexpr (CT == 1) return v; else return v[i];
}
template <int CT>
__device__ __forceinline__ void vset(typename BVec<CT>::type& v, int i, float x) {
    if constexpr (CT == 1) v = x; else v[i] = x;
}

struct SegPos { bool valid; int q; };

__device__ __forceinline__ SegPos seg_pos(const trunet_seg& sg, int p) {
    int qn = p * sg.pos_mul + sg.pos_off;
    SegPos r;
    r.q = qn / sg.pos_div;
    r.valid = (qn >= 0) && (qn - r.q * sg.pos_div == 0) && (r.q < sg.L);
    return r;
}

template <int RS>
__global__ __launch_bounds__(256, 1) void conv_gemm_kernel(const trunet_gemm_args a) {
    constexpr int CT = RS;            // column tiles (of 32 frames) per wave
    constexpr int CG = 4 / RS;        // column groups
    constexpr int MB = 32 * RS;       // rows per M block
    typedef typename BVec<CT>::type bvec;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rs = wave % RS;
    const int cg = wave / RS;
    const int h = lane >> 5;
    const int c = lane & 31;
    const int mblk = blockIdx.y;

    // chunk prefix per segment (uniform)
    int nck_total = 0;
    for (int s = 0; s < a.nseg; ++s) nck_total += (a.seg[s].nchan + KC - 1) / KC;
    float* A_lds = smem;
    float* B_lds = smem + (size_t)nck_total * RS * 1024;

    const int ntn = a.NP / NT;
    const int total_tiles = a.P * ntn;

    // ---- load the weight block into LDS in fragment order: [chunk][rs][kg][lane][4]
    if ((int)blockIdx.x < total_tiles) {
        const int totalA = nck_total * RS * 1024;
        for (int idx = tid; idx < totalA; idx += 256) {
            int j = idx & 3;
            int ln = (idx >> 2) & 63;
            int kg = (idx >> 8) & 3;
            int rest = idx >> 10;
            int rr = rest % RS;
            int ch = rest / RS;
            int s = 0, cc = ch;
            while (cc >= (a.seg[s].nchan + KC - 1) / KC) { cc -= (a.seg[s].nchan + KC - 1) / KC; ++s; }
            int kk = 4 * kg + j;
            int ci = cc * KC + 2 * kk + (ln >> 5);
            int m = mblk * MB + 32 * rr + (ln & 31);
            float v = 0.f;
            if (ci < a.seg[s].nchan && m < a.M)
                v = a.W[(size_t)(m + a.w_m_off) * a.ldw_m + (size_t)ci * a.ldw_c + a.seg[s].woff];
            A_lds[idx] = v;
        }
    }
    __syncthreads();

    float st1[16], st2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { st1[r] = 0.f; st2[r] = 0.f; }

    const int srow = tid >> 5;       // staging: row within chunk = srow + 8*i
    const int sf4 = tid & 31;        // staging: float4 index within the 128-frame row

    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int nt = tile / a.P;
        const int p = a.p_begin + (tile - nt * a.P);
        const int n0 = nt * NT;

        f32x16 acc[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        // chunk iterator state (uniform): segment s, chunk cc inside it, A chunk index ach
        int s = 0, cc = 0, ach = 0;
        auto skip_invalid = [&]() {
            while (s < a.nseg && !seg_pos(a.seg[s], p).valid) {
                ach += (a.seg[s].nchan + KC - 1) / KC;
                ++s;
            }
        };
        skip_invalid();

        f32x4 v0[4], v1[4];
        float k0[4], k1[4], k2[4];
        int cur_mode = 0;
        auto issue_loads = [&](int ss, int cci) {
            const trunet_seg& sg = a.seg[ss];
            const int q = seg_pos(sg, p).q;
            cur_mode = sg.mode;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int ci = cci * KC + srow + 8 * i;
                bool ok = ci < sg.nchan;
                size_t off = ((size_t)ci * sg.L + q) * a.NP + n0 + 4 * sf4;
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                v0[i] = ok ? *(const f32x4*)(sg.src0 + off) : z;
                if (sg.mode == TRUNET_PRO_BNBWD) v1[i] = ok ? *(const f32x4*)(sg.src1 + off) : z;
                if (sg.mode != TRUNET_PRO_NONE) {
                    k0[i] = ok ? sg.c0[ci] : 0.f;
                    k1[i] = ok ? sg.c1[ci] : 0.f;
                    if (sg.mode == TRUNET_PRO_BNBWD) k2[i] = ok ? sg.c2[ci] : 0.f;
                }
            }
        };
        auto write_lds = [&](int buf) {
            float* Bb = B_lds + buf * (KC * NT);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = v0[i];
                if (cur_mode == TRUNET_PRO_BNRELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], k0[i], k1[i]), 0.f);
                } else if (cur_mode == TRUNET_PRO_BNBWD) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(k0[i], v[e], fmaf(k1[i], v1[i][e], k2[i]));
                }
                *(f32x4*)(Bb + (srow + 8 * i) * NT + 4 * sf4) = v;
            }
        };

        int buf = 0;
        if (s < a.nseg) {
            issue_loads(s, cc);
            write_lds(0);
        }
        __syncthreads();

        while (s < a.nseg) {
            // next chunk (uniform)
            int s2 = s, cc2 = cc + 1, ach2 = ach;
            if (cc2 >= (a.seg[s].nchan + KC - 1) / KC) {
                ach2 += (a.seg[s].nchan + KC - 1) / KC;
                s2 = s + 1;
                cc2 = 0;
                while (s2 < a.nseg && !seg_pos(a.seg[s2], p).valid) {
                    ach2 += (a.seg[s2].nchan + KC - 1) / KC;
                    ++s2;
                }
            }
            const bool more = s2 < a.nseg;
            if (more) issue_loads(s2, cc2);

            // ---- MFMAs on the current chunk
            const float* Ab = A_lds + (size_t)((ach + cc) * RS + rs) * 1024;
            const float* Bb = B_lds + buf * (KC * NT);
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                f32x4 a4 = *(const f32x4*)(Ab + (kg * 64 + lane) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 2 * (4 * kg + j) + h;
                    bvec b = *(const bvec*)(Bb + row * NT + (32 * RS) * cg + CT * c);
#pragma unroll
                    for (int t = 0; t < CT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], vget<CT>(b, t), acc[t], 0, 0, 0);
                }
            }
            if (more) write_lds(buf ^ 1);
            __syncthreads();
            buf ^= 1;
            s = s2; cc = cc2; ach = ach2;
        }

        // ---- epilogue
        const int nb = n0 + (32 * RS) * cg + CT * c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = mblk * MB + 32 * rs + ml;
            if (m < a.M) {
                const int mg = m + a.m_out_off;
                const size_t off = ((size_t)mg * a.out_L + p + a.out_pos_off) * a.NP + nb;
                bvec val;
#pragma unroll
                for (int t = 0; t < CT; ++t) vset<CT>(val, t, acc[t][r]);
                if (a.epi & TRUNET_EPI_BIAS) {
                    float bv = a.bias[mg];
#pragma unroll
                    for (int t = 0; t < CT; ++t) vset<CT>(val, t, vget<CT>(val, t) + bv);
                }
                if (a.epi & TRUNET_EPI_ACCUM) {
                    bvec old = *(const bvec*)(a.out + off);
#pragma unroll
                    for (int t = 0; t < CT; ++t) vset<CT>(val, t, vget<CT>(val, t) + vget<CT>(old, t));
                }
                bvec zv;
                float e2 = 0.f;
                if (a.epi & TRUNET_EPI_MASK) {
                    zv = *(const bvec*)(a.zmask + off);
                    const float e0 = a.e0[mg], e1 = a.e1[mg];
                    e2 = a.e2 ? a.e2[mg] : 0.f;
#pragma unroll
                    for (int t = 0; t < CT; ++t)
                        vset<CT>(val, t, (fmaf(e0, vget<CT>(zv, t), e1) > 0.f) ? vget<CT>(val, t) : 0.f);
                }
                if (a.epi & TRUNET_EPI_RELU) {
#pragma unroll
                    for (int t = 0; t < CT; ++t) vset<CT>(val, t, fmaxf(vget<CT>(val, t), 0.f));
                }
                *(bvec*)(a.out + off) = val;
                if (a.epi & TRUNET_EPI_STATS) {
#pragma unroll
                    for (int t = 0; t < CT; ++t) {
                        float x = (nb + t < a.N) ? vget<CT>(val, t) : 0.f;
                        st1[r] += x;
                        if (a.epi & TRUNET_EPI_MASK) st2[r] = fmaf(x, vget<CT>(zv, t) - e2, st2[r]);
                        else st2[r] = fmaf(x, x, st2[r]);
                    }
                }
            }
        }
    }

    if (a.epi & TRUNET_EPI_STATS) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s1 = half_wave_sum(st1[r]);
            float s2 = half_wave_sum(st2[r]);
            const int ml = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = mblk * MB + 32 * rs + ml;
            if (c == 0 && m < a.M) {
                const int mg = m + a.m_out_off;
                float* pp = a.partials + ((size_t)(blockIdx.x * CG + cg) * a.M_stat + mg) * 2;
                pp[0] = s1;
                pp[1] = s2;
            }
        }
    }
}

int pick_rs(int M, int nck_total) {
    int rs = M > 64 ? 4 : (M > 32 ? 2 : 1);
    while (rs > 1 && (size_t)nck_total * rs * 4096 > 112 * 1024) rs >>= 1;
    return rs;
}

}  // namespace

extern "C" int trunet_conv_gemm_nparts(int M) {
    // upper bound independent of the K extent: 256 workgroups x up to 4 column groups
    (void)M;
    return TRUNET_NUM_CU * 4;
}

extern "C" int trunet_conv_gemm(const trunet_gemm_args* h, void* stream) {
    if (!h || !h->out || !h->W || h->nseg < 1 || h->nseg > TRUNET_MAX_SEG) return TRUNET_EINVAL;
    if (h->NP <= 0 || (h->NP % NT) != 0 || h->N > h->NP || h->P <= 0 || h->M <= 0) return TRUNET_EINVAL;
    if ((h->epi & TRUNET_EPI_STATS) && !h->partials) return TRUNET_EINVAL;
    if ((h->epi & TRUNET_EPI_MASK) && (!h->zmask || !h->e0 || !h->e1)) return TRUNET_EINVAL;
    if ((h->epi & TRUNET_EPI_BIAS) && !h->bias) return TRUNET_EINVAL;
    int nck = 0;
    for (int s = 0; s < h->nseg; ++s) {
        const trunet_seg& sg = h->seg[s];
        if (!sg.src0 || sg.nchan <= 0 || sg.pos_div <= 0) return TRUNET_EINVAL;
        if (sg.mode == TRUNET_PRO_BNBWD && (!sg.src1 || !sg.c0 || !sg.c1 || !sg.c2)) return TRUNET_EINVAL;
        if (sg.mode == TRUNET_PRO_BNRELU && (!sg.c0 || !sg.c1)) return TRUNET_EINVAL;
        nck += (sg.nchan + KC - 1) / KC;
    }
    const int rs = pick_rs(h->M, nck);
    const size_t lds = (size_t)nck * rs * 4096 + 2 * KC * NT * sizeof(float);
    if (lds > 160 * 1024) return TRUNET_ENOTSUP;
    const int mb = 32 * rs;
    dim3 grid(TRUNET_NUM_CU, (h->M + mb - 1) / mb);
    hipStream_t st = (hipStream_t)stream;
    if (h->epi & TRUNET_EPI_STATS) {
        // statistics rows are indexed by (blockIdx.x*CG + cg); rows of unused parts must read as zero
        size_t bytes = (size_t)trunet_conv_gemm_nparts(h->M) * h->M_stat * 2 * sizeof(float);
        if (h->m_out_off == 0 && h->M == h->M_stat) {
            if (hipMemsetAsync(h->partials, 0, bytes, st) != hipSuccess) return TRUNET_ELAUNCH;
        }
    }
    switch (rs) {
        case 4: {
            hipFuncSetAttribute((const void*)conv_gemm_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(conv_gemm_kernel<4>, grid, dim3(256), lds, st, *h);
            break;
        }
        case 2: {
            hipFuncSetAttribute((const void*)conv_gemm_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(conv_gemm_kernel<2>, grid, dim3(256), lds, st, *h);
            break;
        }
        default: {
            hipFuncSetAttribute((const void*)conv_gemm_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(conv_gemm_kernel<1>, grid, dim3(256), lds, st, *h);
            break;
        }
    }
    return trunet_launch_status();
}

// =====================================================================================
// Weight gradient:  dW(m, c, seg) = sum_{p, n<N} dz[m][p][n] * act_seg[c][q_seg(p)][n]
//
// Both operands are streamed; the reduction axis (frames) is the MFMA K axis, so operand rows
// sit in LDS as [row][64 frames + 4 pad] (pad => conflict-free ds_read_b128 across 16 rows).
// Persistent workgroups over (p, 64-frame chunk); each wave owns up to 5 output tiles of 32x32
// (dz row tile x act row tile) in accumulators for the whole kernel and finally writes its part of
// a per-workgroup partial image of W (native weight addressing), summed by trunet_reduce_partials.
// Two workgroups per CU hide the staging latency (no register prefetch).
// =====================================================================================
namespace {

constexpr int FC = 64;          // frames per chunk
constexpr int LROW = FC + 4;    // LDS row stride (floats)
constexpr int WG_ROWS = 256;    // LDS rows
constexpr int MAXT = 5;         // accumulator tiles per wave
constexpr int WGRAD_GRID = 2 * TRUNET_NUM_CU;

__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const trunet_wgrad_args a) {
    __shared__ __attribute__((aligned(16))) float lds[WG_ROWS * LROW];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int c = lane & 31;
    const int MA = (a.M + 31) & ~31;        // padded dz rows
    const int nrt = MA / 32;

    // global tile enumeration: g -> (seg, ctile, rt), rt fastest
    int ntile_seg[TRUNET_MAX_SEG];
    int G = 0;
#pragma unroll
    for (int s = 0; s < TRUNET_MAX_SEG; ++s) {
        ntile_seg[s] = (s < a.nseg) ? ((a.seg[s].nchan + 31) / 32) * nrt : 0;
        G += ntile_seg[s];
    }
    // per-slot static description (uniform per wave)
    int t_seg[MAXT], t_ct[MAXT], t_rt[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        int g = wave + 4 * i;
        t_seg[i] = -1; t_ct[i] = 0; t_rt[i] = 0;
        if (g < G) {
            int s = 0;
#pragma unroll
            for (int ss = 0; ss < TRUNET_MAX_SEG; ++ss) {
                if (t_seg[i] < 0) {
                    if (g < ntile_seg[ss]) { t_seg[i] = ss; }
                    else g -= ntile_seg[ss];
                }
            }
            (void)s;
            t_ct[i] = g / nrt;
            t_rt[i] = g - t_ct[i] * nrt;
        }
    }

    f32x16 acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    float bsum[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) bsum[i] = 0.f;

    const int srow = tid >> 4;     // 16 rows per pass
    const int sf4 = tid & 15;      // float4 within the 64-frame row
    const int nfc = a.NP / FC;
    const int total_tiles = a.P * nfc;

    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int fcx = tile / a.P;
        const int p = a.p_begin + (tile - fcx * a.P);
        const int n0 = fcx * FC;
        const int nf = n0 + 4 * sf4;

        // ---- stage dz rows
#pragma unroll
        for (int ps = 0; ps < 12; ++ps) {
            const int row = srow + 16 * ps;
            if (row < MA) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < a.M) {
                    const int ch = row + a.a_m_off;
                    const size_t off = ((size_t)ch * a.a_L + p + a.a_pos_off) * a.NP + nf;
                    v = *(const f32x4*)(a.a0 + off);
                    if (a.a_mode == TRUNET_PRO_BNBWD) {
                        f32x4 z = *(const f32x4*)(a.a1 + off);
                        const float k0 = a.ac0[ch], k1 = a.ac1[ch], k2 = a.ac2[ch];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaf(k0, v[e], fmaf(k1, z[e], k2));
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (nf + e >= a.N) v[e] = 0.f;
                        bsum[ps] += v[e];
                    }
                }
                *(f32x4*)(lds + row * LROW + 4 * sf4) = v;
            }
        }
        // ---- stage activation rows of the valid segments, compactly after the dz rows
        int seg_base[TRUNET_MAX_SEG];
        int rows_used = MA;
#pragma unroll
        for (int s = 0; s < TRUNET_MAX_SEG; ++s) {
            seg_base[s] = -1;
            if (s < a.nseg) {
                const trunet_seg& sg = a.seg[s];
                const SegPos sp = seg_pos(sg, p);
                if (sp.valid) {
                    seg_base[s] = rows_used;
                    const int npad = (sg.nchan + 31) & ~31;
                    for (int r0 = 0; r0 < npad; r0 += 16) {
                        const int ci = r0 + srow;
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (ci < sg.nchan) {
                            const size_t off = ((size_t)ci * sg.L + sp.q) * a.NP + nf;
                            v = *(const f32x4*)(sg.src0 + off);
                            if (sg.mode == TRUNET_PRO_BNRELU) {
                                const float k0 = sg.c0[ci], k1 = sg.c1[ci];
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], k0, k1), 0.f);
                            } else if (sg.mode == TRUNET_PRO_BNBWD) {
                                f32x4 z = *(const f32x4*)(sg.src1 + off);
                                const float k0 = sg.c0[ci], k1 = sg.c1[ci], k2 = sg.c2[ci];
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = fmaf(k0, v[e], fmaf(k1, z[e], k2));
                            }
                        }
                        *(f32x4*)(lds + (rows_used + ci) * LROW + 4 * sf4) = v;
                    }
                    rows_used += npad;
                }
            }
        }
        __syncthreads();

        // ---- MFMAs
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            if (t_seg[i] >= 0) {
                int sb = -1;
#pragma unroll
                for (int s = 0; s < TRUNET_MAX_SEG; ++s)
                    if (t_seg[i] == s) sb = seg_base[s];
                if (sb >= 0) {
                    const float* Ar = lds + (t_rt[i] * 32 + c) * LROW + 4 * h;
                    const float* Br = lds + (sb + t_ct[i] * 32 + c) * LROW + 4 * h;
#pragma unroll
                    for (int q = 0; q < FC / 8; ++q) {
                        f32x4 av = *(const f32x4*)(Ar + 8 * q);
                        f32x4 bv = *(const f32x4*)(Br + 8 * q);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[i], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- write this workgroup's partial image
    float* img = a.w_partials + (size_t)blockIdx.x * a.w_numel;
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        if (t_seg[i] >= 0) {
            const trunet_seg& sg = a.seg[t_seg[i]];
            const int ci = t_ct[i] * 32 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = t_rt[i] * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < a.M && ci < sg.nchan)
                    img[(size_t)(m + a.w_m_off) * a.ldw_m + (size_t)ci * a.ldw_c + sg.woff] = acc[i][r];
            }
        }
    }
    if (a.b_partials) {
#pragma unroll
        for (int ps = 0; ps < 12; ++ps) {
            float v = bsum[ps];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            const int row = srow + 16 * ps;
            if (sf4 == 0 && row < a.M)
                a.b_partials[(size_t)blockIdx.x * a.b_stride + a.b_off + row] = v;
        }
    }
}

__global__ void reduce_partials_kernel(float* out, const float* partials, int nparts, int numel, int accumulate) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numel) return;
    double s = 0.0;
    for (int g = 0; g < nparts; ++g) s += (double)partials[(size_t)g * numel + i];
    out[i] = (accumulate ? out[i] : 0.f) + (float)s;
}

}  // namespace

extern "C" int trunet_conv_wgrad_nparts(void) { return WGRAD_GRID; }

extern "C" int trunet_conv_wgrad(const trunet_wgrad_args* h, void* stream) {
    if (!h || !h->a0 || !h->w_partials || h->nseg < 1 || h->nseg > TRUNET_MAX_SEG) return TRUNET_EINVAL;
    if (h->NP <= 0 || (h->NP % NT) != 0 || h->N > h->NP || h->P <= 0 || h->M <= 0 || h->M > 192) return TRUNET_EINVAL;
    if (h->a_mode == TRUNET_PRO_BNBWD && (!h->a1 || !h->ac0 || !h->ac1 || !h->ac2)) return TRUNET_EINVAL;
    const int MA = (h->M + 31) & ~31;
    int tiles = 0, rows = MA, maxrows = 0;
    for (int s = 0; s < h->nseg; ++s) {
        const trunet_seg& sg = h->seg[s];
        if (!sg.src0 || sg.nchan <= 0 || sg.pos_div <= 0) return TRUNET_EINVAL;
        tiles += ((sg.nchan + 31) / 32) * (MA / 32);
        maxrows += (sg.nchan + 31) & ~31;
    }
    // rows staged at once: all segments may be valid unless they are parity-exclusive (pos_div > 1);
    // the host passes transposed-conv taps with pos_div == stride, of which at most ceil(nseg/stride) are valid
    if (h->nseg > 1 && h->seg[0].pos_div > 1) {
        int per = (h->seg[0].nchan + 31) & ~31;
        int d = h->seg[0].pos_div;
        maxrows = per * ((h->nseg + d - 1) / d);
    }
    rows += maxrows;
    if (tiles > 4 * MAXT || rows > WG_ROWS) return TRUNET_ENOTSUP;
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(WGRAD_GRID), dim3(256), 0, (hipStream_t)stream, *h);
    return trunet_launch_status();
}

extern "C" int trunet_reduce_partials(float* out, const float* partials, int nparts, int numel, int accumulate,
                                      void* stream) {
    if (!out || !partials || nparts <= 0 || numel <= 0) return TRUNET_EINVAL;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((numel + 255) / 256), dim3(256), 0, (hipStream_t)stream, out,
                       partials, nparts, numel, accumulate);
    return trunet_launch_status();
}
