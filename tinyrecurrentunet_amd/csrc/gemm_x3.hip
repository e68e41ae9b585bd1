// fp32 implicit-GEMM convolution on the bf16 MFMA with a three-way (24-bit) operand split (round 4).
//
// The fp32 MFMA of this chip runs at the packed-fp32 vector rate (157 TF); the bf16 MFMA at 16x that.  Every fp32 number is
// the sum of three bf16 numbers, x = x0 + x1 + x2 (8 + 8 + 8 significand bits, each term rounded to nearest, the residues
// exact in fp32), and a product of two bf16 numbers is exact in fp32, so
//     a b = a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0) + O(2^-24 |a b|)
// is six v_mfma_f32_32x32x16_bf16 with fp32 accumulation per 16 K-values -- 6 x 33 cycles against 8 x 64 for
// v_mfma_f32_32x32x2_f32 -- and carries the same 24 significant bits per product as an fp32 FMA chain (the three dropped
// terms are below one fp32 ulp of the product).  The forward launches of the body (pointwise convs over one or two
// concatenated sources, transposed convs as tap gathers, the GRU input projection) move from the fp32-MFMA roofline
// (0.47-0.53 of it, scripts/dbg/mfma_x3.hip for the inner loop) towards their HBM stream.
//
// Same launch contract, segment descriptors, LDS-DMA ring and in-place prologue as conv_gemm_kernel (gemm_conv.hip); what
// differs:
//   * the weight block sits in LDS as THREE bf16 fragment planes (split once per persistent workgroup);
//   * a wave owns ALL row tiles of the block (NRT = 2 or 4) and ONE 32-frame column tile of a 256-frame tile, so the
//     split of its B fragment (8 ds_read_b32 + ~36 vector instructions per 16 K-values) is amortised over 6 NRT MFMAs
//     (1.5 vector instructions per MFMA at 128 rows: free next to the bf16 MFMA, profiles/round2_mfma_microbench.txt);
//   * chunks are 16 K-rows (one MFMA K-step): 96 KB of weight planes (128 x 128) + three 16 KB ring slots fit the LDS;
//   * statistics leave the epilogue through the 16-value DPP butterfly (one live register per row tile).
// Scope: launches without a tensor-operand epilogue (no ReLU mask / accumulate: the forward pass), one-tensor prologue
// (BN+ReLU or none), M = 64 or a multiple of 128, frame count padded to 256.  Everything else stays on conv_gemm_kernel.
#include <cstdlib>
#include "common.hpp"

#include "bf16_common.hpp"
#include "gemm_common.hpp"

namespace {

constexpr int X3_KC = 16;          // K rows per chunk = one MFMA K-step
constexpr int X3_FT = 256;         // frames per tile (8 waves x 32)

// three-way split of two floats into bf16 terms, packed (element 0 in the low half): x = hi + mid + lo with every term
// rounded to NEAREST (v_cvt_pk_bf16_f32), so the residues are signed and zero-mean: hi carries 8 significand bits, x - hi
// is exact in fp32, mid its leading 8 bits, lo the rest (exact up to one unit in the 25th bit).  Same instruction count
// as a truncating split (and / sub), without its bias towards zero in the dropped cross terms.
__device__ __forceinline__ void x3_split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = bf_pack(x0, x1);
    const float r0 = x0 - bf_lo(hi), r1 = x1 - bf_hi(hi);
    mid = bf_pack(r0, r1);
    lo = bf_pack(r0 - bf_lo(mid), r1 - bf_hi(mid));
}

// counted wait for the LDS-DMA ring (2 instructions per wave and chunk); the large counts are used while the stores of a
// tile's epilogue (16 NRT per lane, younger than the DMA being waited for) may still be in flight
__device__ __forceinline__ void x3_wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
        case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;       // n >= 63 (the counter's range)
    }
}

template <int NRT>
__global__ __launch_bounds__(512, 1) void conv_gemm_x3_kernel(const trunet_gemm_args a, const int NB) {
    constexpr int KC = X3_KC, FT = X3_FT;
    constexpr int MB = 32 * NRT;
    constexpr int CHF = KC * FT;                 // floats per ring slot
    constexpr int LPW = 2;                       // DMA instructions per wave and chunk (16 rows, 8 waves)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int c = lane & 31;
    const int mblk = blockIdx.y;

    int nck_total = 0, nchan_total = 0;
    for (int s = 0; s < a.nseg; ++s) {
        nck_total += (a.seg[s].nchan + KC - 1) / KC;
        nchan_total += a.seg[s].nchan;
    }
    // LDS carve: weight planes [chunk][row tile][plane][64 lanes] x 16 B | ring | prologue coefficients | bias
    u32x4* A3 = (u32x4*)smem;
    float* R_lds = (float*)(A3 + (size_t)nck_total * NRT * 3 * 64);
    f32x4* C_lds = (f32x4*)(R_lds + (size_t)NB * CHF);
    float* E_lds = (float*)(C_lds + nchan_total);

    const int ntn = a.NP / FT;
    const int total_tiles = a.P * ntn;
    ChunkIt cur;
    cur.placed = false;
    cur.tile = (int)(((long long)blockIdx.x * total_tiles) / gridDim.x);
    cur.tile_end = (int)(((long long)(blockIdx.x + 1) * total_tiles) / gridDim.x);
    const bool has_work = cur.tile < cur.tile_end;

    float sacc[NRT][2];
#pragma unroll
    for (int t = 0; t < NRT; ++t) { sacc[t][0] = 0.f; sacc[t][1] = 0.f; }

    if (has_work) {
        // weight planes: lane ln of (chunk, row tile) holds W[row 32 rt + (ln & 31)][k = 16 cc + 8 (ln >> 5) + j], j < 8
        for (int idx = tid; idx < nck_total * NRT * 64; idx += 512) {
            const int ln = idx & 63;
            const int rest = idx >> 6;
            const int rt = rest % NRT;
            const int ch = rest / NRT;
            int s = 0, cc = ch;
            while (cc >= (a.seg[s].nchan + KC - 1) / KC) { cc -= (a.seg[s].nchan + KC - 1) / KC; ++s; }
            const int m = mblk * MB + 32 * rt + (ln & 31);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ci = cc * KC + 8 * (ln >> 5) + j;
                const bool ok = ci < a.seg[s].nchan && m < a.M;
                const float w = a.W[(size_t)(min(m, a.M - 1) + a.w_m_off) * a.ldw_m +
                                    (size_t)min(ci, a.seg[s].nchan - 1) * a.ldw_c + a.seg[s].woff];
                v[j] = ok ? w : 0.f;
            }
            u32x4 p0, p1, p2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned u0, u1, u2;
                x3_split2(v[2 * j], v[2 * j + 1], u0, u1, u2);
                p0[j] = u0; p1[j] = u1; p2[j] = u2;
            }
            u32x4* dst = A3 + ((size_t)(ch * NRT + rt) * 3) * 64 + ln;
            dst[0] = p0; dst[64] = p1; dst[128] = p2;
        }
        int base = 0;
        for (int s = 0; s < a.nseg; ++s) {
            const trunet_seg& sg = a.seg[s];
            for (int ci = tid; ci < sg.nchan; ci += 512) {
                const bool on = sg.mode == TRUNET_PRO_BNRELU;
                f32x4 k;
                k[0] = on ? sg.c0[ci] : 1.f; k[1] = on ? sg.c1[ci] : 0.f; k[2] = on ? 0.f : -3.0e38f; k[3] = 0.f;
                C_lds[base + ci] = k;
            }
            base += sg.nchan;
        }
        for (int r = tid; r < MB; r += 512) {
            const int m = mblk * MB + r;
            E_lds[r] = (m < a.M && (a.epi & TRUNET_EPI_BIAS)) ? a.bias[m + a.m_out_off] : 0.f;
        }
    }
    __syncthreads();

    if (has_work) {
        // LDS-DMA of one chunk: 16 rows x 256 frames as 1-KiB wave-instructions, instruction g = row g of the chunk; wave w8
        // issues rows 2 w8, 2 w8 + 1 and later transforms exactly the bytes it requested (no barrier of its own)
        auto issue_dma = [&](const ChunkIt& it, int slot) {
            const trunet_seg& sg = a.seg[it.s];
            const int q = seg_pos(sg, it.p).q;
            float* dst = R_lds + (size_t)slot * CHF;
            const size_t boff = (size_t)q * a.NP + it.n0;
            const unsigned rstride = (unsigned)sg.L * (unsigned)a.NP;
            const float* b0 = sg.src0 + boff;
#pragma unroll
            for (int i = 0; i < LPW; ++i) {
                const int g = LPW * wave8 + i;
                const int cch = min(it.cc * KC + g, sg.nchan - 1);        // rows past the segment: finite filler (A = 0)
                const unsigned off = (unsigned)cch * rstride + (unsigned)(4 * lane);
                __builtin_amdgcn_global_load_lds(b0 + off, (lds_ptr_t)(dst + g * FT), 16, 0, TRUNET_DMA_AUX);
            }
        };
        auto transform = [&](const ChunkIt& it, int slot) {
            float* dst = R_lds + (size_t)slot * CHF;
            const int nrow = a.seg[it.s].nchan - it.cc * KC;
            f32x4 v[LPW], k[LPW];
#pragma unroll
            for (int i = 0; i < LPW; ++i) {
                const int g = LPW * wave8 + i;
                v[i] = *(const f32x4*)(dst + g * FT + 4 * lane);
                k[i] = C_lds[it.cbase + it.cc * KC + min(g, nrow - 1)];
            }
#pragma unroll
            for (int i = 0; i < LPW; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(fmaf(v[i][e], k[i][0], k[i][1]), k[i][2]);
#pragma unroll
            for (int i = 0; i < LPW; ++i) *(f32x4*)(dst + (LPW * wave8 + i) * FT + 4 * lane) = v[i];
        };

        if (wave8 >= 4) __builtin_amdgcn_s_setprio(1);
        // ---- software pipeline over the flattened chunk stream (chunk j lives in ring slot j % NB):
        //   iteration i:  request B(i+1) from LDS -> 6 NRT MFMAs of chunk i on the fragments split one iteration ago ->
        //                 in-place prologue of chunk i+2 (its DMA was issued NB-1 iterations ago) -> split B(i+1) ->
        //                 barrier -> DMA of chunk i+1+NB into the slot B(i+1) was read from.
        // The matrix pipe starts right after the barrier (nothing of chunk i is read from the ring any more) and the
        // vector work of the next chunk runs behind the issued MFMAs; the first version did read -> split -> MFMA inside
        // one barrier interval and left the pipe idle for a third of it (100 TF-equivalent at 128 x 128).
        auto read_b = [&](int sl, float (&x)[8]) {
            const float* Bb = R_lds + (size_t)sl * CHF + 32 * wave8 + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = Bb[(8 * h + j) * FT];
        };
        auto split_b = [&](const float (&x)[8], u32x4& q0, u32x4& q1, u32x4& q2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned u0, u1, u2;
                x3_split2(x[2 * j], x[2 * j + 1], u0, u1, u2);
                q0[j] = u0; q1[j] = u1; q2[j] = u2;
            }
        };
        it_enter_tile<KC, FT>(a, cur);
        ChunkIt ld = cur, ldlast = cur;
        for (int d = 0; d < NB; ++d) {
            if (ld.valid) ldlast = ld;
            issue_dma(ldlast, d);                       // past the end: harmless re-load of the last chunk
            if (ld.valid) it_next<KC, FT>(a, ld);
        }
        ChunkIt tf = cur;
        int tslot = 0;
        x3_wait_vmcnt((NB - 2) * LPW);                  // chunks 0 and 1 have landed
        transform(tf, 0);
        it_next<KC, FT>(a, tf);
        tslot = 1;
        if (tf.valid) {
            transform(tf, 1);
            it_next<KC, FT>(a, tf);
            tslot = (2 == NB) ? 0 : 2;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        u32x4 bp0, bp1, bp2;
        {
            float x[8];
            read_b(0, x);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // every wave holds B(0): slot 0 takes chunk NB
            asm volatile("" ::: "memory");
            if (ld.valid) ldlast = ld;
            issue_dma(ldlast, 0);
            if (ld.valid) it_next<KC, FT>(a, ld);
            split_b(x, bp0, bp1, bp2);
        }

        int slot = 0;
        // Chunks since the last epilogue.  Its 16 NRT stores per lane are YOUNGER than the NB - 1 chunks of DMA requested
        // before it, so for the next NB - 1 prologue passes the DMA being waited for is complete as soon as at most
        // (NB - 2) LPW + 16 NRT operations are outstanding (in-order retirement, as everywhere in these kernels): the wait
        // does not drain the tile's stores.
        int since_epi = NB;
        const int relaxed = min(63, (NB - 2) * LPW + 16 * NRT);
        bool first = true;
        int tp = 0, nn = 0, tn0 = 0;
        f32x16 acc[NRT];
        // output rows as buffer resource (uniform base of the tile's row block) + one per-lane offset (4 h rows down, this
        // lane's frame) + a scalar row offset: 64 per-row 64-bit addresses do not fit next to 64 accumulators
        const size_t rowstride = (size_t)a.out_L * a.NP;
        const int rowb = (int)(rowstride * sizeof(float));
        const int voff = (int)((4 * h * rowstride + 32 * wave8 + c) * sizeof(float));
        while (cur.valid) {
            if (first) {
#pragma unroll
                for (int t = 0; t < NRT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
                tp = cur.p;
                tn0 = cur.n0;
                nn = cur.n0 + 32 * wave8 + c;            // this lane's frame
                first = false;
            }
            ChunkIt nxt = cur;
            const bool last = it_next<KC, FT>(a, nxt);
            const int nslot = (slot + 1 == NB) ? 0 : slot + 1;
            float x[8];
            read_b(nslot, x);                            // B of the NEXT chunk (published by the previous barrier)
            {
                const u32x4* Ab = A3 + ((size_t)(cur.ach + cur.cc) * NRT * 3) * 64 + lane;
#pragma unroll
                for (int t = 0; t < NRT; ++t) {
                    const u32x4 a0 = Ab[(t * 3 + 0) * 64], a1 = Ab[(t * 3 + 1) * 64], a2 = Ab[(t * 3 + 2) * 64];
#define X3_MF(a_, b_) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc[t], 0, 0, 0)
                    // the small cross terms first, the leading term last
                    X3_MF(a2, bp0); X3_MF(a0, bp2); X3_MF(a1, bp1); X3_MF(a1, bp0); X3_MF(a0, bp1); X3_MF(a0, bp0);
#undef X3_MF
                }
            }
            if (tf.valid) {                              // prologue pass on chunk i + 2
                x3_wait_vmcnt(since_epi < NB - 1 ? relaxed : (NB - 2) * LPW);
                ++since_epi;
                transform(tf, tslot);
                it_next<KC, FT>(a, tf);
                tslot = (tslot + 1 == NB) ? 0 : tslot + 1;
            }
            u32x4 bn0, bn1, bn2;
            split_b(x, bn0, bn1, bn2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ld.valid) ldlast = ld;
            issue_dma(ldlast, nslot);                    // every wave holds B(i+1): its slot takes chunk i + 1 + NB
            if (ld.valid) it_next<KC, FT>(a, ld);
            bp0 = bn0; bp1 = bn1; bp2 = bn2;
            slot = nslot;
            cur = nxt;
            if (!last) continue;
            // ---- epilogue of the tile: bias (+ ReLU), 128-byte row pieces per half-wave, statistics through the butterfly
            since_epi = 0;
            first = true;
#pragma unroll
            for (int t = 0; t < NRT; ++t) {
                float s1[16], s2[16];
                const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                    a.out + ((size_t)(mblk * MB + 32 * t + a.m_out_off) * a.out_L + tp + a.out_pos_off) * a.NP + tn0, 0,
                    0x7fffffff, 0x00020000);
                const bool fin = nn < a.N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = (r & 3) + 8 * (r >> 2);          // (+ 4 h: in voff); M is a multiple of 32 here
                    float val = acc[t][r] + E_lds[32 * t + ml + 4 * h];
                    if (a.epi & TRUNET_EPI_RELU) val = fmaxf(val, 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val), ro, voff, ml * rowb, 0);
                    const float xs = fin ? val : 0.f;
                    s1[r] = xs;
                    s2[r] = xs * xs;
                }
                if (a.epi & TRUNET_EPI_STATS) {
                    sacc[t][0] += butterfly16(s1, c);
                    sacc[t][1] += butterfly16(s2, c);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the ring before the workgroup exits
    }

    if (a.epi & TRUNET_EPI_STATS) {
        const int r = butterfly16_index(c);
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
            const int m = mblk * MB + 32 * t + 8 * (r >> 2) + 4 * h + (r & 3);
            if (butterfly16_writer(c) && m < a.M) {
                float* pp = a.partials + ((size_t)(blockIdx.x * 8 + wave8) * a.M_stat + m + a.m_out_off) * 2;
                pp[0] = sacc[t][0];
                pp[1] = sacc[t][1];
            }
        }
    }
}

}  // namespace

// 0 = not eligible (the caller launches conv_gemm_kernel), else NRT.  *nb / *lds: ring slots and LDS bytes.
static int g_x3_on = -1;       // -1: not decided yet.  Bit mask: TRUNET_X3_GEMM (1) | TRUNET_X3_BWD (2); default TRUNET_X3_BWD

extern "C" int trunet_gemm_x3_enable(int on) {
    if (g_x3_on < 0) {
        const char* e = getenv("TRUNET_GEMM_X3");
        g_x3_on = (e && e[0] >= '0' && e[0] <= '3' && !e[1]) ? e[0] - '0' : TRUNET_X3_BWD;
    }
    const int prev = g_x3_on;
    if (on >= 0) g_x3_on = on & (TRUNET_X3_GEMM | TRUNET_X3_BWD | 4 | 8);     // 4 / 8 (diagnostics): only pw_bwd / only convt_bwd
    return prev;
}

int trunet_gemm_x3_plan(const trunet_gemm_args* h, int* nb, size_t* lds) {
    if (!(trunet_gemm_x3_enable(-1) & TRUNET_X3_GEMM)) return 0;
    if (h->epi & (TRUNET_EPI_MASK | TRUNET_EPI_ACCUM)) return 0;
    if ((h->NP % X3_FT) != 0) return 0;
    int nrt;
    if (h->M == 64) nrt = 2;
    else if (h->M >= 128 && (h->M % 128) == 0) nrt = 4;
    else return 0;
    int nck = 0, nchan = 0;
    for (int s = 0; s < h->nseg; ++s) {
        if (h->seg[s].mode == TRUNET_PRO_BNBWD) return 0;
        nck += (h->seg[s].nchan + X3_KC - 1) / X3_KC;
        nchan += h->seg[s].nchan;
    }
    const size_t fixed = (size_t)nck * nrt * 3 * 1024 + (size_t)nchan * 16 + 32 * nrt * sizeof(float);
    const size_t slot = (size_t)X3_KC * X3_FT * sizeof(float);
    if (fixed + 2 * slot > 160 * 1024) return 0;
    int n = (int)((160 * 1024 - fixed) / slot);
    if (n > 4) n = 4;
    *nb = n;
    *lds = fixed + n * slot;
    return nrt;
}

int trunet_launch_gemm_x3(const trunet_gemm_args* h, int nrt, int nb, size_t lds, hipStream_t st) {
    dim3 grid(TRUNET_NUM_CU, (h->M + 32 * nrt - 1) / (32 * nrt));
    if (nrt == 2) {
        if (hipFuncSetAttribute((const void*)conv_gemm_x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return TRUNET_ELAUNCH;
        hipLaunchKernelGGL(conv_gemm_x3_kernel<2>, grid, dim3(512), lds, st, *h, nb);
    } else {
        if (hipFuncSetAttribute((const void*)conv_gemm_x3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return TRUNET_ELAUNCH;
        hipLaunchKernelGGL(conv_gemm_x3_kernel<4>, grid, dim3(512), lds, st, *h, nb);
    }
    return trunet_launch_status();
}
