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
#include <cstdlib>
#include "common.hpp"

#include "gemm_common.hpp"

namespace {


// RS: 32-row slices of the M block (wave tiling, see header).  KC: K rows per chunk.  TWO: the operand is
// c0*src0 + c1*src1 + c2 (BatchNorm backward) instead of max(c0*src0 + c1, lo).  EPL: epilogue tensor
// loads (0 none, 1 zmask, 2 zmask + previous output).
//
// Pipeline per K-chunk (ring of NB LDS slots, chunk j in slot j % NB):
//   LDS-DMA (global_load_lds, issued NB chunks ahead, no VGPRs)  ->  in-place prologue pass by the thread that
//   issued the DMA (one chunk ahead of the MFMAs)  ->  fragment reads + MFMAs.  One raw s_barrier per chunk;
//   DMA completion is tracked with counted s_waitcnt vmcnt, so NB-1 chunks (16 KiB each) stay in flight.
// NW = 4: 128-frame tiles, one wave per SIMD.  NW = 8: 256-frame tiles, the second group of four waves takes the
// upper 128 frames of every tile -- same per-wave bookkeeping, twice the MFMAs between two barriers, and two waves
// per SIMD that fill each other's LDS/issue gaps inside the MFMA phase.
template <int RS, int KC, bool TWO, int EPL, int NW>
__global__ __launch_bounds__(64 * NW, (EPL == 0 || NW == 8) ? 2 : 1) void conv_gemm_kernel(const trunet_gemm_args a, const int NB) {
    constexpr int FT = 32 * NW;        // frames per tile
    constexpr int CT = RS;            // column tiles (of 32 frames) per wave
    constexpr int CG = 4 / RS;        // column groups
    constexpr int MB = 32 * RS;       // rows per M block
    constexpr int KP = KC / 2;        // k-pairs per chunk
    constexpr int NSRC = TWO ? 2 : 1;
    constexpr int CHF = KC * FT * NSRC;            // floats per ring slot
    constexpr int LPW = (KC / 8) * NSRC;           // DMA instructions per wave per chunk
    typedef typename BVec<CT>::type bvec;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave8 >> 2;      // which 128 frames of the tile (NW = 8)
    const int wave = wave8 & 3;
    const int rs = wave % RS;
    const int cg = wave / RS;
    const int h = lane >> 5;
    const int c = lane & 31;
    const int mblk = blockIdx.y;

    int nck_total = 0, nchan_total = 0;
    for (int s = 0; s < a.nseg; ++s) {
        nck_total += (a.seg[s].nchan + KC - 1) / KC;
        nchan_total += a.seg[s].nchan;
    }
    // LDS carve (one array): A fragments | ring | prologue coefficients (float4 per channel) | epilogue rows
    float* A_lds = smem;
    float* R_lds = A_lds + (size_t)nck_total * RS * (KP * 64);
    f32x4* C_lds = (f32x4*)(R_lds + (size_t)NB * CHF);
    float* E_lds = (float*)(C_lds + nchan_total);

    const int ntn = a.NP / FT;
    const int total_tiles = a.P * ntn;
    ChunkIt cur;
    cur.placed = false;
    cur.tile = (int)(((long long)blockIdx.x * total_tiles) / gridDim.x);
    cur.tile_end = (int)(((long long)(blockIdx.x + 1) * total_tiles) / gridDim.x);
    const bool has_work = cur.tile < cur.tile_end;

    float st1[16], st2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { st1[r] = 0.f; st2[r] = 0.f; }

    if (has_work) {
        // weight block in fragment order: [chunk][rs][kg][lane][4], k-pair = 4*kg + j
        const int totalA = nck_total * RS * (KP * 64);
        for (int idx = tid; idx < totalA; idx += 64 * NW) {
            const int j = idx & 3;
            const int ln = (idx >> 2) & 63;
            const int rest = idx >> 8;
            const int kg = rest % (KP / 4);
            const int rest2 = rest / (KP / 4);
            const int rr = rest2 % RS;
            const int ch = rest2 / RS;
            int s = 0, cc = ch;
            while (cc >= (a.seg[s].nchan + KC - 1) / KC) { cc -= (a.seg[s].nchan + KC - 1) / KC; ++s; }
            const int kk = 4 * kg + j;
            const int ci = cc * KC + 2 * kk + (ln >> 5);
            const int m = mblk * MB + 32 * rr + (ln & 31);
            // branch-free (clamped address, then select): the compiler otherwise waits for every load before the next
            const bool ok = ci < a.seg[s].nchan && m < a.M;
            const float v = a.W[(size_t)(min(m, a.M - 1) + a.w_m_off) * a.ldw_m +
                                (size_t)min(ci, a.seg[s].nchan - 1) * a.ldw_c + a.seg[s].woff];
            A_lds[idx] = ok ? v : 0.f;
        }
        int base = 0;
        for (int s = 0; s < a.nseg; ++s) {
            const trunet_seg& sg = a.seg[s];
            for (int ci = tid; ci < sg.nchan; ci += 64 * NW) {
                f32x4 k;
                if (TWO) {
                    const bool on = sg.mode == TRUNET_PRO_BNBWD;
                    k[0] = on ? sg.c0[ci] : 1.f; k[1] = on ? sg.c1[ci] : 0.f; k[2] = on ? sg.c2[ci] : 0.f; k[3] = 0.f;
                } else {
                    const bool on = sg.mode == TRUNET_PRO_BNRELU;
                    k[0] = on ? sg.c0[ci] : 1.f; k[1] = on ? sg.c1[ci] : 0.f; k[2] = on ? 0.f : -3.0e38f; k[3] = 0.f;
                }
                C_lds[base + ci] = k;
            }
            base += sg.nchan;
        }
        for (int r = tid; r < MB; r += 64 * NW) {
            const int m = mblk * MB + r;
            const int mg = m + a.m_out_off;
            const bool ok = m < a.M;
            E_lds[r] = (ok && (a.epi & TRUNET_EPI_BIAS)) ? a.bias[mg] : 0.f;
            E_lds[MB + r] = (ok && EPL > 0) ? a.e0[mg] : 0.f;
            E_lds[2 * MB + r] = (ok && EPL > 0) ? a.e1[mg] : 0.f;
            E_lds[3 * MB + r] = (ok && EPL > 0 && a.e2) ? a.e2[mg] : 0.f;
        }
    }
    __syncthreads();

    if (has_work) {
        // ---- LDS-DMA of one chunk: KC rows x FT frames (per tensor) as 1-KiB wave-instructions; instruction g covers
        // slot elements [256 g, 256 g + 256).  Wave w8 issues g = (LPW/NSRC) w8 + i/NSRC for tensor i % NSRC.  The SAME
        // thread later transforms exactly the bytes it requested, so the prologue pass needs no barrier of its own.
        auto issue_dma = [&](const ChunkIt& it, int slot) {
            const trunet_seg& sg = a.seg[it.s];
            const int q = seg_pos(sg, it.p).q;
            float* dst = R_lds + (size_t)slot * CHF;
            // uniform base (position, first frame) + 32-bit per-lane element offset (channel row, frame): a tensor has
            // fewer than 2^32 elements (host-checked), so no 64-bit multiplies per load
            const size_t boff = (size_t)q * a.NP + it.n0;
            const unsigned rstride = (unsigned)sg.L * (unsigned)a.NP;
            const float* b0 = sg.src0 + boff;
            const float* b1 = (TWO ? (sg.src1 ? sg.src1 : sg.src0) : sg.src0) + boff;
#pragma unroll
            for (int i = 0; i < LPW; ++i) {
                const int g = (LPW / NSRC) * wave8 + i / NSRC;
                const int e = g * 256 + 4 * lane;
                const int row = e / FT, col = e % FT;
                const int cch = min(it.cc * KC + row, sg.nchan - 1);      // rows past the segment: finite filler (A = 0)
                const unsigned off = (unsigned)cch * rstride + (unsigned)col;
                const float* gp = ((TWO && (i % NSRC)) ? b1 : b0) + off;
                __builtin_amdgcn_global_load_lds(gp, (lds_ptr_t)(dst + (i % NSRC) * (KC * FT) + g * 256), 16, 0, TRUNET_DMA_AUX);
            }
        };
        // in-place prologue on this thread's own pieces of the chunk in `slot` (all reads first, then the math,
        // then all writes: the compiler must not serialise read-modify-write pairs through LDS aliasing)
        auto transform = [&](const ChunkIt& it, int slot) {
            float* dst = R_lds + (size_t)slot * CHF;
            const int nrow = a.seg[it.s].nchan - it.cc * KC;
            f32x4 v[LPW / NSRC], z[LPW / NSRC], k[LPW / NSRC];
#pragma unroll
            for (int i = 0; i < LPW / NSRC; ++i) {
                const int e = ((LPW / NSRC) * wave8 + i) * 256 + 4 * lane;
                const float* pz = dst + e;
                v[i] = *(const f32x4*)pz;
                if (TWO) z[i] = *(const f32x4*)(pz + KC * FT);
                k[i] = C_lds[it.cbase + it.cc * KC + min(e / FT, nrow - 1)];
            }
#pragma unroll
            for (int i = 0; i < LPW / NSRC; ++i) {
                if (TWO) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][e] = fmaf(k[i][0], v[i][e], fmaf(k[i][1], z[i][e], k[i][2]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(fmaf(v[i][e], k[i][0], k[i][1]), k[i][2]);
                }
            }
#pragma unroll
            for (int i = 0; i < LPW / NSRC; ++i)
                *(f32x4*)(dst + ((LPW / NSRC) * wave8 + i) * 256 + 4 * lane) = v[i];
        };

        // Two waves per SIMD: the second-dispatched half loses every issue arbitration at equal priority; one static
        // s_setprio for it (no per-phase flips) is worth +19 % on the 64-row launches (pw 192 -> 64: 66 -> 79 TF) and
        // is neutral at 128 rows.
        if (NW == 8 && wave8 >= 4) __builtin_amdgcn_s_setprio(1);
        // ---- pipeline prologue
        it_enter_tile<KC, FT>(a, cur);
        ChunkIt ld = cur, ldlast = cur;
        for (int d = 0; d < NB; ++d) {
            if (ld.valid) ldlast = ld;
            issue_dma(ldlast, d);                       // past the end: harmless re-load of the last chunk
            if (ld.valid) it_next<KC, FT>(a, ld);
        }
        ChunkIt tf = cur;                               // chunk whose prologue pass comes next
        int tslot = 0;
        wait_vmcnt((NB - 1) * LPW);
        transform(tf, tslot);
        it_next<KC, FT>(a, tf);
        tslot = 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a raw s_barrier does not wait for the LDS writes above
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        int slot = 0;
        while (cur.valid) {                              // ---- tiles
            f32x16 acc[CT];
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            const int tp = cur.p;
            const int nb = cur.n0 + 128 * half + (32 * RS) * cg + CT * c;
            bvec zv[16], ov[16];
            bool last = false;
            while (!last) {                              // ---- chunks of the tile
                ChunkIt nxt = cur;
                last = it_next<KC, FT>(a, nxt);
                if (EPL > 0 && last) {                   // epilogue operands: issued before the MFMAs
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = 32 * rs + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int m = min(mblk * MB + ml, a.M - 1);
                        const size_t off = ((size_t)(m + a.m_out_off) * a.out_L + tp + a.out_pos_off) * a.NP + nb;
                        zv[r] = *(const bvec*)(a.zmask + off);
                        if (EPL > 1) ov[r] = *(const bvec*)(a.out + off);
                    }
                }
                // prologue pass on the next chunk (its DMA was issued NB-1 chunks ago)
                if (tf.valid) {
                    if (EPL > 0 && last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else wait_vmcnt((NB - 2) * LPW);
                    transform(tf, tslot);
                    it_next<KC, FT>(a, tf);
                    tslot = (tslot + 1 == NB) ? 0 : tslot + 1;
                }
                // fragments of chunk `cur`, then the MFMAs
                {
                    const float* Ab = A_lds + (size_t)((cur.ach + cur.cc) * RS + rs) * (KP * 64);
                    const float* Bb = R_lds + (size_t)slot * CHF + 128 * half + (32 * RS) * cg + CT * c;
                    f32x4 af[KP / 4];
                    bvec bf[KP];
#pragma unroll
                    for (int kg = 0; kg < KP / 4; ++kg) af[kg] = *(const f32x4*)(Ab + (kg * 64 + lane) * 4);
#pragma unroll
                    for (int kk = 0; kk < KP; ++kk) bf[kk] = *(const bvec*)(Bb + (2 * kk + h) * FT);
#pragma unroll
                    for (int kk = 0; kk < KP; ++kk)
#pragma unroll
                        for (int t = 0; t < CT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk >> 2][kk & 3], vget<CT>(bf[kk], t),
                                                                          acc[t], 0, 0, 0);
                    // keep the LDS reads LA k-pairs ahead of the MFMAs that consume them (hipcc otherwise sinks
                    // every ds_read directly in front of its MFMAs and the matrix pipe idles on LDS latency)
                    constexpr int LA = 1 + 4 / CT;
                    __builtin_amdgcn_sched_group_barrier(0x100, KP / 4 + LA, 0);
#pragma unroll
                    for (int kk = 0; kk < KP - LA; ++kk) {
                        __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, LA * CT, 0);
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                // slot `slot` is free for everyone now: refill it with chunk (cur + NB)
                if (ld.valid) ldlast = ld;
                issue_dma(ldlast, slot);
                if (ld.valid) it_next<KC, FT>(a, ld);
                slot = (slot + 1 == NB) ? 0 : slot + 1;
                cur = nxt;
            }
            // ---- epilogue of the tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = 32 * rs + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int m = mblk * MB + ml;
                const int mg = m + a.m_out_off;
                const size_t off = ((size_t)mg * a.out_L + tp + a.out_pos_off) * a.NP + nb;
                bvec val;
                const float bv = E_lds[ml];
#pragma unroll
                for (int t = 0; t < CT; ++t) vset<CT>(val, t, acc[t][r] + bv);
                if (EPL > 1) {
#pragma unroll
                    for (int t = 0; t < CT; ++t) vset<CT>(val, t, vget<CT>(val, t) + vget<CT>(ov[r], t));
                }
                float e2 = 0.f;
                if (EPL > 0) {
                    const float e0 = E_lds[MB + ml], e1 = E_lds[2 * MB + ml];
                    e2 = E_lds[3 * MB + ml];
#pragma unroll
                    for (int t = 0; t < CT; ++t)
                        vset<CT>(val, t, (fmaf(e0, vget<CT>(zv[r], t), e1) > 0.f) ? vget<CT>(val, t) : 0.f);
                }
                if (a.epi & TRUNET_EPI_RELU) {
#pragma unroll
                    for (int t = 0; t < CT; ++t) vset<CT>(val, t, fmaxf(vget<CT>(val, t), 0.f));
                }
#ifdef TRUNET_EPI_NT
                if (m < a.M) __builtin_nontemporal_store(val, (bvec*)(a.out + off));
#else
                if (m < a.M) *(bvec*)(a.out + off) = val;
#endif
                if (a.epi & TRUNET_EPI_STATS) {
#pragma unroll
                    for (int t = 0; t < CT; ++t) {
                        const float x = (nb + t < a.N && m < a.M) ? vget<CT>(val, t) : 0.f;
                        st1[r] += x;
                        if (EPL > 0) st2[r] = fmaf(x, vget<CT>(zv[r], t) - e2, st2[r]);
                        else st2[r] = fmaf(x, x, st2[r]);
                    }
                }
            }
        }
        // drain the ring before the workgroup exits
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
                float* pp = a.partials + ((size_t)((blockIdx.x * (NW / 4) + half) * CG + cg) * a.M_stat + mg) * 2;
                pp[0] = s1;
                pp[1] = s2;
            }
        }
    }
}


// ---- M <= 8 outputs: the same implicit GEMM on the vector ALU (the matrix tiles would be 3/4 padding and the
// layer is a pure HBM stream: decoder.5 pw 128->8, ConvTranspose1d 8->8 and its data gradient).
// Thread = (position p, 4 frames); weights [channel][8] and prologue coefficients sit in LDS (broadcast reads).
template <int EPL>
__global__ __launch_bounds__(256) void conv_smallm_kernel(const trunet_gemm_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double red[256];
    int nchan_total = 0;
    for (int s = 0; s < a.nseg; ++s) nchan_total += a.seg[s].nchan;
    f32x4* Wl = (f32x4*)smem;                 // [nchan_total][2]
    f32x4* Cl = Wl + 2 * nchan_total;         // [nchan_total]
    const int tid = threadIdx.x;
    {
        int base = 0;
        for (int s = 0; s < a.nseg; ++s) {
            const trunet_seg& sg = a.seg[s];
            for (int ci = tid; ci < sg.nchan; ci += 256) {
                float w[8];
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    w[m] = (m < a.M) ? a.W[(size_t)(m + a.w_m_off) * a.ldw_m + (size_t)ci * a.ldw_c + sg.woff] : 0.f;
                f32x4 w0 = {w[0], w[1], w[2], w[3]}, w1 = {w[4], w[5], w[6], w[7]};
                Wl[2 * (base + ci)] = w0;
                Wl[2 * (base + ci) + 1] = w1;
                const bool on = sg.mode == TRUNET_PRO_BNRELU;
                f32x4 k = {on ? sg.c0[ci] : 1.f, on ? sg.c1[ci] : 0.f, on ? 0.f : -3.0e38f, 0.f};
                Cl[base + ci] = k;
            }
            base += sg.nchan;
        }
    }
    __syncthreads();
    // a wave = one item = (position, 256 frames): 1 KiB contiguous per row and load instruction (with 128-frame items a
    // wave read two 512-byte pieces of two positions)
    const int f4 = tid & 63, py = tid >> 6;
    const int ntn = (a.NP + 255) / 256;
    const int items = a.P * ntn;
    float s1[8], s2[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) { s1[m] = 0.f; s2[m] = 0.f; }
    for (int it = blockIdx.x * 4 + py; it < items; it += gridDim.x * 4) {
        const int nt = it / a.P;
        const int p = a.p_begin + (it - nt * a.P);
        const int n = nt * 256 + 4 * f4;
        if (n >= a.NP) continue;              // NP is a multiple of 128: the last item may be half empty
        f32x4 acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        int cb = 0;
        for (int s = 0; s < a.nseg; ++s) {
            const trunet_seg& sg = a.seg[s];
            const SegPos sp = seg_pos(sg, p);
            if (sp.valid) {
                const float* src = sg.src0 + (size_t)sp.q * a.NP + n;
                const size_t cstr = (size_t)sg.L * a.NP;
                // eight channel rows requested before the first is used (with four, the 5 x 8-channel taps of the transposed conv
                // waited twice per tap: 200 -> 150 us; the 128-channel pointwise layer is unchanged at 2.9 TB/s)
                for (int c0 = 0; c0 < sg.nchan; c0 += 8) {
                    f32x4 v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int ci = min(c0 + j, sg.nchan - 1);
#ifdef TRUNET_THIN_NT
                        v[j] = __builtin_nontemporal_load((const f32x4*)(src + ci * cstr));
#else
                        v[j] = *(const f32x4*)(src + ci * cstr);
#endif
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int ci = c0 + j;
                        if (ci < sg.nchan) {
                            const f32x4 k = Cl[cb + ci];
                            const f32x4 w0 = Wl[2 * (cb + ci)], w1 = Wl[2 * (cb + ci) + 1];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[j][e] = fmaxf(fmaf(v[j][e], k[0], k[1]), k[2]);
#pragma unroll
                            for (int m = 0; m < 4; ++m)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    acc[m][e] = fmaf(w0[m], v[j][e], acc[m][e]);
                                    acc[4 + m][e] = fmaf(w1[m], v[j][e], acc[4 + m][e]);
                                }
                        }
                    }
                }
            }
            cb += sg.nchan;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (m < a.M) {
                const int mg = m + a.m_out_off;
                const size_t off = ((size_t)mg * a.out_L + p + a.out_pos_off) * a.NP + n;
                f32x4 val = acc[m];
                if (a.epi & TRUNET_EPI_BIAS) {
                    const float bv = a.bias[mg];
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] += bv;
                }
                f32x4 zv = {0.f, 0.f, 0.f, 0.f};
                float e2 = 0.f;
                if (EPL > 1) {
                    const f32x4 old = *(const f32x4*)(a.out + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] += old[e];
                }
                if (EPL > 0) {
                    zv = *(const f32x4*)(a.zmask + off);
                    const float e0 = a.e0[mg], e1 = a.e1[mg];
                    e2 = a.e2 ? a.e2[mg] : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = (fmaf(e0, zv[e], e1) > 0.f) ? val[e] : 0.f;
                }
                if (a.epi & TRUNET_EPI_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
                }
                *(f32x4*)(a.out + off) = val;
                if (a.epi & TRUNET_EPI_STATS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = (n + e < a.N) ? val[e] : 0.f;
                        s1[m] += x;
                        s2[m] = (EPL > 0) ? fmaf(x, zv[e] - e2, s2[m]) : fmaf(x, x, s2[m]);
                    }
                }
            }
        }
    }
    if (a.epi & TRUNET_EPI_STATS) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const double r1 = block_sum_f64((double)s1[m], red);
            const double r2 = block_sum_f64((double)s2[m], red);
            if (tid == 0 && m < a.M) {
                float* pp = a.partials + ((size_t)blockIdx.x * a.M_stat + m + a.m_out_off) * 2;
                pp[0] = (float)r1;
                pp[1] = (float)r2;
            }
        }
    }
}

int launch_smallm(const trunet_gemm_args* h, hipStream_t st) {
    int nchan_total = 0;
    for (int s = 0; s < h->nseg; ++s) nchan_total += h->seg[s].nchan;
    const size_t lds = (size_t)nchan_total * 3 * sizeof(f32x4);
    const int epl = (h->epi & TRUNET_EPI_MASK) ? ((h->epi & TRUNET_EPI_ACCUM) ? 2 : 1) : 0;
    const int items = h->P * ((h->NP + 255) / 256);
    int grid = (items + 3) / 4;
    if (grid > 1024) grid = 1024;             // = trunet_conv_gemm_nparts: one statistics row per block
    if (epl == 0) hipLaunchKernelGGL(conv_smallm_kernel<0>, dim3(grid), dim3(256), lds, st, *h);
    else if (epl == 1) hipLaunchKernelGGL(conv_smallm_kernel<1>, dim3(grid), dim3(256), lds, st, *h);
    else hipLaunchKernelGGL(conv_smallm_kernel<2>, dim3(grid), dim3(256), lds, st, *h);
    return trunet_launch_status();
}

template <int RS, int KC, bool TWO, int EPL, int NW>
int launch_gemm_nw(const trunet_gemm_args* h, int NB, size_t lds, hipStream_t st) {
    const int mb = 32 * RS;
    dim3 grid((NW == 4 && EPL == 0 && lds <= 80 * 1024) ? 2 * TRUNET_NUM_CU : TRUNET_NUM_CU, (h->M + mb - 1) / mb);
    auto kern = conv_gemm_kernel<RS, KC, TWO, EPL, NW>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return TRUNET_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, *h, NB);
    return trunet_launch_status();
}

template <int RS, int KC, bool TWO, int EPL>
int launch_gemm(const trunet_gemm_args* h, int NB, size_t lds, hipStream_t st, int nw = 4) {
    if constexpr (EPL == 0 || (EPL == 1 && RS <= 2)) {
        if (nw == 8) return launch_gemm_nw<RS, KC, TWO, EPL, 8>(h, NB, lds, st);
    }
    return launch_gemm_nw<RS, KC, TWO, EPL, 4>(h, NB, lds, st);
}

template <int RS>
int launch_gemm_rs(const trunet_gemm_args* h, int kc, bool two, int epl, int NB, size_t lds, hipStream_t st, int nw) {
    if constexpr (RS == 2) {
        // K <= 8 rows of dz (the data gradient of decoder.5's 128 -> 8 pointwise layer): 8-row chunks -- a 32-row chunk
        // moved and multiplied 24 rows of padding per tile
        if (two && kc == 8) {
            if (epl == 0) return launch_gemm<RS, 8, true, 0>(h, NB, lds, st, nw);
            if (epl == 1) return launch_gemm<RS, 8, true, 1>(h, NB, lds, st, nw);
            return launch_gemm<RS, 8, true, 2>(h, NB, lds, st, nw);
        }
    }
    if (two && kc == 32) {
        if (epl == 0) return launch_gemm<RS, 32, true, 0>(h, NB, lds, st, nw);
        if (epl == 1) return launch_gemm<RS, 32, true, 1>(h, NB, lds, st, nw);
        return launch_gemm<RS, 32, true, 2>(h, NB, lds, st, nw);
    }
    if (two) {
        if (epl == 0) return launch_gemm<RS, 16, true, 0>(h, NB, lds, st, nw);
        if (epl == 1) return launch_gemm<RS, 16, true, 1>(h, NB, lds, st, nw);
        return launch_gemm<RS, 16, true, 2>(h, NB, lds, st, nw);
    }
    if (epl == 0) return launch_gemm<RS, 32, false, 0>(h, NB, lds, st, nw);
    if (epl == 1) return launch_gemm<RS, 32, false, 1>(h, NB, lds, st, nw);
    return launch_gemm<RS, 32, false, 2>(h, NB, lds, st, nw);
}

// launch geometry shared by trunet_conv_gemm and (for reporting) the host: row slices, chunk rows, ring slots
struct GemmPlan { int rs, kc, nb, nw; size_t lds; bool two; int epl; };

int plan_gemm(const trunet_gemm_args* h, GemmPlan* pl) {
    bool two = false;
    int nchan_total = 0, kpad32 = 0;
    for (int s = 0; s < h->nseg; ++s) {
        two = two || h->seg[s].mode == TRUNET_PRO_BNBWD;
        nchan_total += h->seg[s].nchan;
        kpad32 += (h->seg[s].nchan + 31) / 32 * 32;
    }
    int rs0 = h->M > 64 ? 4 : (h->M > 32 ? 2 : 1);
    const size_t budget = 160 * 1024;
    static const bool kc8_ok = !(getenv("TRUNET_GEMM_KC8") && getenv("TRUNET_GEMM_KC8")[0] == '0');
    const bool thin_k = kc8_ok && two && rs0 == 2 && h->nseg == 1 && nchan_total <= 8;
    for (int rs = rs0; rs >= 1; rs >>= 1) {
        for (int kc = thin_k ? 8 : 32; kc >= (thin_k ? 8 : (two ? 16 : 32)); kc >>= 1) {
            int kpad = 0;
            for (int s = 0; s < h->nseg; ++s) kpad += (h->seg[s].nchan + kc - 1) / kc * kc;
            const size_t slot = (size_t)kc * NT * sizeof(float) * (two ? 2 : 1);
            const size_t fixed = (size_t)kpad * rs * 128 + (size_t)nchan_total * 16 + 4 * 32 * rs * sizeof(float);
            if (fixed + 3 * slot > budget) continue;
            int nb = (int)((budget - fixed) / slot);
            if (nb > 6) nb = 6;
            if (!(h->epi & TRUNET_EPI_MASK) && fixed + 2 * slot <= budget / 2) nb = (int)((budget / 2 - fixed) / slot);
            pl->rs = rs; pl->kc = kc; pl->nb = nb; pl->lds = fixed + nb * slot; pl->two = two; pl->nw = 4;
            // wide variant (8 waves, 256-frame tiles) for launches without tensor-operand epilogue when it fits
            static const bool wide_ok = !(getenv("TRUNET_GEMM_WIDE") && getenv("TRUNET_GEMM_WIDE")[0] == '0');
            const int epl_w = (h->epi & TRUNET_EPI_MASK) ? ((h->epi & TRUNET_EPI_ACCUM) ? 2 : 1) : 0;
            const bool wide_kind = epl_w == 0 || (epl_w == 1 && rs <= 2);     // register budget of two waves per SIMD
            if (wide_ok && wide_kind && (h->NP % 256) == 0 && fixed + 2 * 2 * slot <= budget &&
                (pl->lds > 80 * 1024 || epl_w > 0)) {
                int nbw = (int)((budget - fixed) / (2 * slot));
                if (nbw > 4) nbw = 4;
                pl->nw = 8; pl->nb = nbw; pl->lds = fixed + (size_t)nbw * 2 * slot;
            } else if (wide_ok && wide_kind && two && kc == 32 && (h->NP % 256) == 0) {
                // two-tensor prologue: 16-row chunks halve the slot, which can make the wide variant fit
                // (transposed-conv data gradient 64 x (3 x 64): 46 -> 54 TF)
                int kpad16 = 0;
                for (int s = 0; s < h->nseg; ++s) kpad16 += (h->seg[s].nchan + 15) / 16 * 16;
                const size_t slot16 = (size_t)16 * NT * sizeof(float) * 2;
                const size_t fixed16 = (size_t)kpad16 * rs * 128 + (size_t)nchan_total * 16 + 4 * 32 * rs * sizeof(float);
                if (fixed16 + 2 * 2 * slot16 <= budget) {
                    int nbw = (int)((budget - fixed16) / (2 * slot16));
                    if (nbw > 4) nbw = 4;
                    pl->kc = 16; pl->nw = 8; pl->nb = nbw; pl->lds = fixed16 + (size_t)nbw * 2 * slot16;
                }
            }
            pl->epl = (h->epi & TRUNET_EPI_MASK) ? ((h->epi & TRUNET_EPI_ACCUM) ? 2 : 1) : 0;
            return TRUNET_OK;
        }
    }
    (void)kpad32;
    return TRUNET_ENOTSUP;
}

}  // namespace

// gemm_x3.hip: the three-term bf16-split kernel for the launches without a tensor-operand epilogue
int trunet_gemm_x3_plan(const trunet_gemm_args* h, int* nb, size_t* lds);
int trunet_launch_gemm_x3(const trunet_gemm_args* h, int nrt, int nb, size_t lds, hipStream_t st);

extern "C" int trunet_conv_gemm_plan(const trunet_gemm_args* h, int* rs, int* kc, int* nb, int* two, int* epl, int* nw) {
    if (!h || !rs || !kc || !nb || !two || !epl || !nw) return TRUNET_EINVAL;
    {
        int xnb = 0;
        size_t xlds = 0;
        const int nrt = (h->M > 8) ? trunet_gemm_x3_plan(h, &xnb, &xlds) : 0;
        if (nrt) {          // conv_gemm_x3_kernel<NRT>: reported as rs = NRT, kc = 16, nw = -8 (bf16x3 MFMA)
            *rs = nrt; *kc = 16; *nb = xnb; *two = 0; *epl = 0; *nw = -8;
            return TRUNET_OK;
        }
    }
    GemmPlan pl;
    if (plan_gemm(h, &pl) != TRUNET_OK) return TRUNET_ENOTSUP;
    *rs = pl.rs; *kc = pl.kc; *nb = pl.nb; *two = pl.two ? 1 : 0; *epl = pl.epl; *nw = pl.nw;
    return TRUNET_OK;
}

extern "C" int trunet_conv_gemm_nparts(int M) {
    (void)M;
    return TRUNET_NUM_CU * 8;   // up to 512 workgroups x up to 4 column groups
}

extern "C" int trunet_conv_gemm(const trunet_gemm_args* h, void* stream) {
    if (!h || !h->out || !h->W || h->nseg < 1 || h->nseg > TRUNET_MAX_SEG) return TRUNET_EINVAL;
    if (h->NP <= 0 || (h->NP % NT) != 0 || h->N > h->NP || h->P <= 0 || h->M <= 0) return TRUNET_EINVAL;
    if ((h->epi & TRUNET_EPI_STATS) && !h->partials) return TRUNET_EINVAL;
    if ((h->epi & TRUNET_EPI_MASK) && (!h->zmask || !h->e0 || !h->e1)) return TRUNET_EINVAL;
    if ((h->epi & TRUNET_EPI_ACCUM) && !(h->epi & TRUNET_EPI_MASK)) return TRUNET_ENOTSUP;
    if ((h->epi & TRUNET_EPI_BIAS) && !h->bias) return TRUNET_EINVAL;
    // extents: every output row p + out_pos_off must exist, statistics rows must cover the launch's channels
    if (h->out_L <= 0 || h->p_begin < 0 || h->m_out_off < 0 || h->p_begin + h->out_pos_off < 0 ||
        h->p_begin + h->P + h->out_pos_off > h->out_L)
        return TRUNET_EINVAL;
    if ((h->epi & TRUNET_EPI_STATS) && h->M_stat < h->M + h->m_out_off) return TRUNET_EINVAL;
    bool any_two = false, any_relu = false;
    for (int s = 0; s < h->nseg; ++s) {
        const trunet_seg& sg = h->seg[s];
        if (!sg.src0 || sg.nchan <= 0 || sg.pos_div <= 0 || sg.L <= 0) return TRUNET_EINVAL;
        if (sg.pos_div > 2) return TRUNET_ENOTSUP;          // seg_pos works with shifts: strides 1 and 2
        if (sg.mode == TRUNET_PRO_BNBWD && (!sg.src1 || !sg.c0 || !sg.c1 || !sg.c2)) return TRUNET_EINVAL;
        if (sg.mode == TRUNET_PRO_BNRELU && (!sg.c0 || !sg.c1)) return TRUNET_EINVAL;
        any_two = any_two || sg.mode == TRUNET_PRO_BNBWD;
        any_relu = any_relu || sg.mode == TRUNET_PRO_BNRELU;
    }
    if (any_two && any_relu) return TRUNET_ENOTSUP;   // one prologue family per launch
    GemmPlan pl;
    if (plan_gemm(h, &pl) != TRUNET_OK) return TRUNET_ENOTSUP;
    hipStream_t st = (hipStream_t)stream;
    if ((h->epi & TRUNET_EPI_STATS) && !(h->epi & TRUNET_EPI_PREZERO)) {
        // statistics rows are indexed by (blockIdx.x*CG + cg); rows of unused parts must read as zero
        size_t bytes = (size_t)trunet_conv_gemm_nparts(h->M) * h->M_stat * 2 * sizeof(float);
        if (hipMemsetAsync(h->partials, 0, bytes, st) != hipSuccess) return TRUNET_ELAUNCH;
    }
    if (h->M <= 8 && !any_two) return launch_smallm(h, st);
    {
        int xnb = 0;
        size_t xlds = 0;
        const int nrt = trunet_gemm_x3_plan(h, &xnb, &xlds);
        if (nrt) return trunet_launch_gemm_x3(h, nrt, xnb, xlds, st);
    }
    switch (pl.rs) {
        case 4: return launch_gemm_rs<4>(h, pl.kc, pl.two, pl.epl, pl.nb, pl.lds, st, pl.nw);
        case 2: return launch_gemm_rs<2>(h, pl.kc, pl.two, pl.epl, pl.nb, pl.lds, st, pl.nw);
        default: return launch_gemm_rs<1>(h, pl.kc, pl.two, pl.epl, pl.nb, pl.lds, st, pl.nw);
    }
}

// =====================================================================================
// Weight gradient:  dW(m, c, seg) = sum_{p, n<N} dz[m][p][n] * act_seg[c][q_seg(p)][n]
//
// Both operands are streamed; the reduction axis (frames) is the MFMA K axis.  A tile is one position p x 32
// frames of every operand row: [dz rows (dy, and z in BN-backward mode)] [rows of up to nvmax valid segments].
// Tiles arrive by LDS-DMA into a ring of NB slots (rows of 128 B, 16-byte pieces XOR-swizzled through the
// per-lane SOURCE address so that the row-per-lane fragment reads are bank-conflict free); the thread that
// requested a piece applies the prologue to it in place (dz = ca*dy + cb*z + cc, BN+ReLU of the activations,
// zeroing of frames >= N) and accumulates the bias gradient on the way.  Each wave owns up to 5 output tiles
// of 32x32 in accumulators for the whole kernel and finally writes its part of a per-workgroup partial image
// of W (native weight addressing), summed by trunet_reduce_partials.
// =====================================================================================
namespace {

constexpr int WFC = 32;         // frames per tile
constexpr int MAXT = 3;         // accumulator tiles per wave (8 waves)
constexpr int WGRAD_GRID = TRUNET_NUM_CU;

__device__ __forceinline__ void wait_vmcnt_any(int n) {
    switch (n) {
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        W_(0) W_(1) W_(2) W_(3) W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15)
        W_(16) W_(17) W_(18) W_(19) W_(20) W_(21) W_(22) W_(23) W_(24) W_(25) W_(26) W_(27) W_(28) W_(29) W_(30) W_(31)
        W_(32) W_(33) W_(34) W_(35) W_(36) W_(37) W_(38) W_(39) W_(40) W_(41) W_(42) W_(43) W_(44) W_(45) W_(46) W_(47)
        W_(48) W_(49) W_(50) W_(51) W_(52) W_(53) W_(54) W_(55) W_(56) W_(57) W_(58) W_(59) W_(60)
#undef W_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// swizzled float offset of 16-byte piece `pc` (0..7) of row `r` inside a slot
__device__ __forceinline__ int wg_off(int r, int pc) { return r * WFC + 4 * (pc ^ ((r >> 1) & 7)); }

template <bool TWO>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(const trunet_wgrad_args a, const int NB, const int nvmax,
                                                            const int rows) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int c = lane & 31;
    const int MA = (a.M + 31) & ~31;          // padded dz rows
    const int nrt = MA / 32;
    const int DZR = TWO ? 2 * MA : MA;        // rows of the dz block (dy [+ z])
    // 8 waves (two per SIMD).  DMA instructions (8 rows x 128 B each) are dealt round-robin: wave w issues the dz
    // row groups g = w + 8 i < Gd and the staged-segment row groups g = w + 8 i < Gs.
    const int Gd = MA / 8;
    const int Gs = (rows - DZR) / 8;
    const int PPW = (Gd - wave + 7) / 8;      // this wave's dz row groups
    const int SPW = (Gs - wave + 7) / 8;      // this wave's staged-segment row groups
    const int LPW = PPW * (TWO ? 2 : 1) + SPW;   // this wave's DMA instructions per tile
    const int SLOT = rows * WFC;              // floats per slot

    float* R_lds = smem;
    f32x4* CA = (f32x4*)(R_lds + (size_t)NB * SLOT);      // [MA] dz coefficients
    int ntot = 0;
    for (int s = 0; s < a.nseg; ++s) ntot += a.seg[s].nchan;
    f32x4* CB = CA + MA;                                     // [sum nchan] activation coefficients

    for (int r = tid; r < MA; r += 512) {
        const int ch = min(r, a.M - 1) + a.a_m_off;
        f32x4 k = {1.f, 0.f, 0.f, 0.f};
        if (TWO) { k[0] = a.ac0[ch]; k[1] = a.ac1[ch]; k[2] = a.ac2[ch]; }
        if (r >= a.M) { k[0] = 0.f; k[1] = 0.f; k[2] = 0.f; }
        CA[r] = k;
    }
    {
        int base = 0;
        for (int s = 0; s < a.nseg; ++s) {
            const trunet_seg& sg = a.seg[s];
            for (int ci = tid; ci < sg.nchan; ci += 512) {
                const bool on = sg.mode == TRUNET_PRO_BNRELU;
                f32x4 k = {on ? sg.c0[ci] : 1.f, on ? sg.c1[ci] : 0.f, on ? 0.f : -3.0e38f, 0.f};
                CB[base + ci] = k;
            }
            base += sg.nchan;
        }
    }

    // global tile enumeration of the OUTPUT: g -> (seg, ctile, rt), rt fastest; wave w owns g = w, w+8, ...
    int ntile_seg[TRUNET_MAX_SEG];
    int G = 0;
#pragma unroll
    for (int s = 0; s < TRUNET_MAX_SEG; ++s) {
        ntile_seg[s] = (s < a.nseg) ? ((a.seg[s].nchan + 31) / 32) * nrt : 0;
        G += ntile_seg[s];
    }
    int t_seg[MAXT], t_ct[MAXT], t_rt[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        int g = wave + 8 * i;
        t_seg[i] = -1; t_ct[i] = 0; t_rt[i] = 0;
        if (g < G) {
#pragma unroll
            for (int ss = 0; ss < TRUNET_MAX_SEG; ++ss) {
                if (t_seg[i] < 0) {
                    if (g < ntile_seg[ss]) t_seg[i] = ss;
                    else g -= ntile_seg[ss];
                }
            }
            t_ct[i] = g / nrt;
            t_rt[i] = g - t_ct[i] * nrt;
        }
    }
    f32x16 acc[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float bsum[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) bsum[i] = 0.f;

    const int nfc = a.NP / WFC;
    const int total_tiles = a.P * nfc;
    const int t_begin = (int)(((long long)blockIdx.x * total_tiles) / gridDim.x);
    const int t_end = (int)(((long long)(blockIdx.x + 1) * total_tiles) / gridDim.x);
    __syncthreads();

    if (t_begin < t_end) {
        const int pc = lane & 7;
        // The workgroup's tile range is cut at position changes: everything that depends on p only (staged
        // segments, per-lane source rows, coefficients) is set up once per run of tiles with the same p.
        int t0 = t_begin;
        while (t0 < t_end) {
            const int pi = t0 / nfc;
            const int p = a.p_begin + pi;
            const int t1 = min(t_end, (pi + 1) * nfc);       // tiles [t0, t1) share p; frame chunk = t - pi*nfc
            // staged segment list: the valid segments in order, the last one repeated up to nvmax entries
            int sl[TRUNET_MAX_SEG], sq[TRUNET_MAX_SEG], sbase[TRUNET_MAX_SEG];
            int nvalid = 0;
            {
                int lasts = 0, lastq = 0;
#pragma unroll
                for (int s = 0; s < TRUNET_MAX_SEG; ++s) { sl[s] = 0; sq[s] = 0; sbase[s] = -1; }
#pragma unroll
                for (int s = 0; s < TRUNET_MAX_SEG; ++s) {
                    if (s < a.nseg) {
                        const SegPos sp = seg_pos(a.seg[s], p);
                        if (sp.valid && nvalid < nvmax) {
#pragma unroll
                            for (int j = 0; j < TRUNET_MAX_SEG; ++j) if (j == nvalid) { sl[j] = s; sq[j] = sp.q; }
                            lasts = s; lastq = sp.q; ++nvalid;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < TRUNET_MAX_SEG; ++j) if (j >= nvalid && j < nvmax) { sl[j] = lasts; sq[j] = lastq; }
                int rb = DZR;
#pragma unroll
                for (int j = 0; j < TRUNET_MAX_SEG; ++j) {
                    if (j < nvalid) {
#pragma unroll
                        for (int s = 0; s < TRUNET_MAX_SEG; ++s) if (sl[j] == s) sbase[s] = rb;
                    }
                    if (j < nvmax) rb += (a.seg[sl[j]].nchan + 31) & ~31;
                }
            }
            // per-lane source pointers (frame 0 of this lane's row, logical piece folded in) and coefficient rows
            const float* pd[3];      // dy rows
            const float* pzr[3];     // z rows (TWO)
            const float* ps[5];      // staged-segment rows
            int cidx[5];             // coefficient index of the staged-segment rows
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                pd[i] = a.a0; pzr[i] = a.a0;
                if (i < PPW) {
                    const int r = 8 * (wave + 8 * i) + (lane >> 3);
                    const int lc = pc ^ ((r >> 1) & 7);
                    const int m = min(r, a.M - 1) + a.a_m_off;
                    const size_t off = ((size_t)m * a.a_L + p + a.a_pos_off) * a.NP + 4 * lc;
                    pd[i] = a.a0 + off;
                    if (TWO) pzr[i] = a.a1 + off;
                }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                ps[i] = a.a0; cidx[i] = 0;
                if (i < SPW) {
                    const int g = wave + 8 * i;
                    int rb = 0, sidx = sl[0], q = sq[0], ch0 = 0;
#pragma unroll
                    for (int j = 0; j < TRUNET_MAX_SEG; ++j) {
                        if (j < nvmax) {
                            const int srp = (a.seg[sl[j]].nchan + 31) & ~31;
                            if (8 * g >= rb && 8 * g < rb + srp) { sidx = sl[j]; q = sq[j]; ch0 = 8 * g - rb; }
                            rb += srp;
                        }
                    }
                    const trunet_seg& sg = a.seg[sidx];
                    const int r = DZR + 8 * g + (lane >> 3);
                    const int lc = pc ^ ((r >> 1) & 7);
                    const int ci = min(ch0 + (lane >> 3), sg.nchan - 1);
                    ps[i] = sg.src0 + ((size_t)ci * sg.L + q) * a.NP + 4 * lc;
                    int cb = 0;
                    for (int s = 0; s < sidx; ++s) cb += a.seg[s].nchan;
                    cidx[i] = cb + ci;
                }
            }

            auto issue_dma = [&](int t, int slot) {
                const int n0 = (t - pi * nfc) * WFC;
                float* dst = R_lds + (size_t)slot * SLOT;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (i < PPW) {
                        const int g = wave + 8 * i;
                        __builtin_amdgcn_global_load_lds(pd[i] + n0, (lds_ptr_t)(dst + g * 256), 16, 0, TRUNET_DMA_AUX);
                        if (TWO) __builtin_amdgcn_global_load_lds(pzr[i] + n0, (lds_ptr_t)(dst + (MA / 8 + g) * 256), 16, 0, TRUNET_DMA_AUX);
                    }
                }
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    if (i < SPW) {
                        const int g = wave + 8 * i;
                        __builtin_amdgcn_global_load_lds(ps[i] + n0, (lds_ptr_t)(dst + (DZR / 8 + g) * 256), 16, 0, TRUNET_DMA_AUX);
                    }
                }
            };
            // prologue pass on this thread's own pieces of tile t
            auto transform = [&](int t, int slot) {
                const int n0 = (t - pi * nfc) * WFC;
                float* dst = R_lds + (size_t)slot * SLOT;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (i < PPW) {
                        const int r = 8 * (wave + 8 * i) + (lane >> 3);
                        const int lc = pc ^ ((r >> 1) & 7);
                        float* pz = dst + r * WFC + 4 * pc;
                        f32x4 v = *(f32x4*)pz;
                        const f32x4 k = CA[r];
                        if (TWO) {
                            const f32x4 z = *(const f32x4*)(pz + MA * WFC);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaf(k[0], v[e], fmaf(k[1], z[e], k[2]));
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= k[0];
                        }
                        float sacc = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (n0 + 4 * lc + e >= a.N) v[e] = 0.f;
                            sacc += v[e];
                        }
                        bsum[i] += sacc;
                        *(f32x4*)pz = v;
                    }
                }
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    if (i < SPW) {
                        const f32x4 k = CB[cidx[i]];
                        float* pz = dst + (DZR + 8 * (wave + 8 * i) + (lane >> 3)) * WFC + 4 * pc;
                        f32x4 v = *(f32x4*)pz;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], k[0], k[1]), k[2]);
                        *(f32x4*)pz = v;
                    }
                }
            };

            // ---- pipeline over tiles [t0, t1): NB tiles in flight, first one transformed
            for (int d = 0; d < NB; ++d) issue_dma(min(t0 + d, t1 - 1), d);
            wait_vmcnt_any((NB - 1) * LPW);
            transform(t0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a raw s_barrier does not wait for the LDS writes above
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int slot = 0;
            for (int t = t0; t < t1; ++t) {
                if (t + 1 < t1) {
                    wait_vmcnt_any((NB - 2) * LPW);
                    transform(t + 1, (slot + 1 == NB) ? 0 : slot + 1);
                }
                const float* S = R_lds + (size_t)slot * SLOT;
#pragma unroll
                for (int i = 0; i < MAXT; ++i) {
                    if (t_seg[i] >= 0) {
                        int sb = -1;
#pragma unroll
                        for (int s = 0; s < TRUNET_MAX_SEG; ++s) if (t_seg[i] == s) sb = sbase[s];
                        if (sb >= 0) {
                            const int ra = t_rt[i] * 32 + c;
                            const int rb = sb + t_ct[i] * 32 + c;
#pragma unroll
                            for (int q = 0; q < WFC / 8; ++q) {
                                const f32x4 av = *(const f32x4*)(S + wg_off(ra, 2 * q + h));
                                const f32x4 bv = *(const f32x4*)(S + wg_off(rb, 2 * q + h));
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[i], 0, 0, 0);
                            }
                            // fragment reads one group ahead of the MFMAs that consume them
                            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                        }
                    }
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue_dma(min(t + NB, t1 - 1), slot);        // refill the slot just released
                slot = (slot + 1 == NB) ? 0 : slot + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // ring is reused by the next run of tiles
            asm volatile("" ::: "memory");
            t0 = t1;
        }
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
        for (int i = 0; i < 3; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            const int row = 8 * (wave + 8 * i) + (lane >> 3);
            if ((lane & 7) == 0 && i < PPW && row < a.M)
                a.b_partials[(size_t)blockIdx.x * a.b_stride + a.b_off + row] = v;
        }
    }
}

// XW columns per block, 256 / XW row groups: wide (64 x 4) for the big flat gradient image, narrow (16 x 16) for the small
// per-layer tables (depthwise weight / bias partials: a few hundred columns x hundreds of rows, where 4 row groups meant ~130
// dependent loads per thread, ~50 us per launch)
template <int XW>
__global__ __launch_bounds__(256) void reduce_partials_kernel(float* out, const float* __restrict__ partials, int nparts,
                                                              int numel, int accumulate) {
    constexpr int YG = 256 / XW;
    __shared__ double red[YG][XW];
    const int x = threadIdx.x % XW, y = threadIdx.x / XW;
    const int i = blockIdx.x * XW + x;
    double s = 0.0;
    if (i < numel)
        for (int g = y; g < nparts; g += YG) s += (double)partials[(size_t)g * numel + i];
    red[y][x] = s;
    __syncthreads();
    if (y == 0 && i < numel) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < YG; ++k) t += red[k][x];
        out[i] = (accumulate ? out[i] : 0.f) + (float)t;
    }
}

}  // namespace

extern "C" int trunet_conv_wgrad_nparts(void) { return WGRAD_GRID; }

extern "C" int trunet_conv_wgrad(const trunet_wgrad_args* h, void* stream) {
    if (!h || !h->a0 || !h->w_partials || h->nseg < 1 || h->nseg > TRUNET_MAX_SEG) return TRUNET_EINVAL;
    if (h->NP <= 0 || (h->NP % NT) != 0 || h->N > h->NP || h->P <= 0 || h->M <= 0 || h->M > 192) return TRUNET_EINVAL;
    if (h->a_mode == TRUNET_PRO_BNBWD && (!h->a1 || !h->ac0 || !h->ac1 || !h->ac2)) return TRUNET_EINVAL;
    // extents: every dz row p + a_pos_off must exist; the partial images must hold the weight
    if (h->a_L <= 0 || h->p_begin < 0 || h->a_m_off < 0 || h->p_begin + h->a_pos_off < 0 ||
        h->p_begin + h->P + h->a_pos_off > h->a_L || h->w_numel <= 0)
        return TRUNET_EINVAL;
    const bool two = h->a_mode == TRUNET_PRO_BNBWD;
    const int MA = (h->M + 31) & ~31;
    int tiles = 0, ntot = 0, allrows = 0, maxsrp = 0;
    for (int s = 0; s < h->nseg; ++s) {
        const trunet_seg& sg = h->seg[s];
        if (!sg.src0 || sg.nchan <= 0 || sg.pos_div <= 0 || sg.L <= 0) return TRUNET_EINVAL;
        if (sg.pos_div > 2) return TRUNET_ENOTSUP;          // seg_pos works with shifts: strides 1 and 2
        if (sg.mode == TRUNET_PRO_BNBWD) return TRUNET_ENOTSUP;
        if (sg.mode == TRUNET_PRO_BNRELU && (!sg.c0 || !sg.c1)) return TRUNET_EINVAL;
        const int srp = (sg.nchan + 31) & ~31;
        tiles += (srp / 32) * (MA / 32);
        allrows += srp;
        ntot += sg.nchan;
        if (srp > maxsrp) maxsrp = srp;
    }
    {   // thin layers (<= 8 rows of dz, or <= 4 channels per segment): stream them on the vector ALU
        int maxc = 0;
        for (int s = 0; s < h->nseg; ++s) maxc = h->seg[s].nchan > maxc ? h->seg[s].nchan : maxc;
        static const bool small_ok = !(getenv("TRUNET_WGRAD_SMALL") && getenv("TRUNET_WGRAD_SMALL")[0] == '0');
        if (small_ok && (maxc <= 4 || (h->M <= 8 && maxc <= 8))) {
            const int rc = trunet_launch_wgrad_small(h, (hipStream_t)stream);
            if (rc != TRUNET_ENOTSUP) return rc;
        }
    }
    // segments with pos_div > 1 are the taps of a strided transposed conv: at most ceil(nseg/stride) are valid at
    // one position (and they have equal channel counts); otherwise every segment may be valid
    int nvmax = h->nseg, segrows = allrows;
    if (h->nseg > 1 && h->seg[0].pos_div > 1) {
        nvmax = (h->nseg + h->seg[0].pos_div - 1) / h->seg[0].pos_div;
        segrows = nvmax * maxsrp;
    }
    const int rows = MA * (two ? 2 : 1) + segrows;
    if (tiles > 8 * MAXT || rows > 512 || MA > 6 * 32 || rows - MA * (two ? 2 : 1) > 320) return TRUNET_ENOTSUP;
    const size_t slot = (size_t)rows * WFC * sizeof(float);
    const size_t fixed = (size_t)(MA + ntot) * sizeof(f32x4);
    int NB = (int)((160 * 1024 - fixed) / slot);
    if (NB > 4) NB = 4;
    if (NB < 2) return TRUNET_ENOTSUP;
    if ((NB - 1) * ((MA / 8 + 7) / 8 * (two ? 2 : 1) + ((rows - MA * (two ? 2 : 1)) / 8 + 7) / 8) > 60) return TRUNET_ENOTSUP;
    const size_t lds = fixed + NB * slot;
    hipStream_t st = (hipStream_t)stream;
    if (two) {
        if (hipFuncSetAttribute((const void*)conv_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TRUNET_ELAUNCH;
        hipLaunchKernelGGL(conv_wgrad_kernel<true>, dim3(WGRAD_GRID), dim3(512), lds, st, *h, NB, nvmax, rows);
    } else {
        if (hipFuncSetAttribute((const void*)conv_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return TRUNET_ELAUNCH;
        hipLaunchKernelGGL(conv_wgrad_kernel<false>, dim3(WGRAD_GRID), dim3(512), lds, st, *h, NB, nvmax, rows);
    }
    return trunet_launch_status();
}

extern "C" int trunet_reduce_partials(float* out, const float* partials, int nparts, int numel, int accumulate,
                                      void* stream) {
    if (!out || !partials || nparts <= 0 || numel <= 0) return TRUNET_EINVAL;
    if (numel <= 4096 && nparts >= 64)
        hipLaunchKernelGGL(reduce_partials_kernel<16>, dim3((numel + 15) / 16), dim3(256), 0, (hipStream_t)stream, out,
                           partials, nparts, numel, accumulate);
    else
        hipLaunchKernelGGL(reduce_partials_kernel<64>, dim3((numel + 63) / 64), dim3(256), 0, (hipStream_t)stream, out,
                           partials, nparts, numel, accumulate);
    return trunet_launch_status();
}
