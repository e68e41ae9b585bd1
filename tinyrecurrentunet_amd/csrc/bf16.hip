// bf16 storage / bf16 MFMA family (BASELINE.json configs[2], a build extension: SURVEY 8d).
//
// At 16x the fp32 MFMA rate every layer of the body is a plain HBM stream, so these kernels are built for memory
// behaviour, not for matrix-pipe scheduling: activations / gradients in the OCTET layout bf16 [C/8][L][NP][8] (the 8
// channels of an octet are the 8 consecutive k values one lane feeds to v_mfma_f32_32x32x16_bf16, i.e. ONE 16-byte global
// load per lane and k-step, 1 KiB contiguous per wave, no LDS staging of activations in the forward / data-gradient GEMM);
// weights as MFMA A fragments in LDS (packed per launch from the fp32 master weights); fp32 accumulators, statistics,
// coefficients, weight gradients.  Taps, strides, F.pad, crops and the concat of network.py:95-98 stay whole-row offsets /
// extra segments.
#include <type_traits>
#include "bf16_common.hpp"

namespace {

// A/B switches for the cache policy of the operand streams (default: temporal)
#ifdef TRUNET_BW_NT
#define BW_LD(p) __builtin_nontemporal_load(p)
#else
#define BW_LD(p) (*(p))
#endif
#ifdef TRUNET_BGL_NT
#define BG_LD(p) __builtin_nontemporal_load(p)
#else
#define BG_LD(p) (*(p))
#endif
constexpr int BG_GRID = 2 * TRUNET_NUM_CU;      // workgroups of 4 waves, two per CU
constexpr int BG_MAXKS = 24;                    // k-steps of 16 channels over all segments (5 x 64 channels = 20)

struct BSegPos { bool valid; int q; };
__device__ __forceinline__ BSegPos bseg_pos(const trunet_bseg& sg, int p) {
    const int qn = p * sg.pos_mul + sg.pos_off;
    BSegPos r;
    // strides are 1 or 2 in this network: no integer division (~25 scalar instructions each) on the per-tile path
    if (sg.pos_div == 1) { r.q = qn; r.valid = (qn >= 0) && (qn < sg.L); }
    else if (sg.pos_div == 2) { r.q = qn >> 1; r.valid = (qn >= 0) && !(qn & 1) && (r.q < sg.L); }
    else {
        r.q = qn / sg.pos_div;
        r.valid = (qn >= 0) && (qn - r.q * sg.pos_div == 0) && (r.q < sg.L);
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// Implicit GEMM (forward of Conv1d k = 1 / ConvTranspose1d / the strided first conv, and their data gradients).
// A wave owns NRT <= 2 row tiles (64 output channels) of a tile (position p, 64 frames = two MFMA column blocks sharing
// every A fragment); for M > 64 two waves split the rows.  The kernel is bound by the bytes a CU keeps in flight (32 KB per
// CU measured 2.0 TB/s), so a wave requests KG k-steps x 2 column blocks (x 2 tensors for a BatchNorm-backward pair) = 16
// KiB before it touches the first: KG = 8 single-tensor, 4 two-tensor (TWO).  Workgroups are renumbered so that each XCD
// (blockIdx % 8) walks a contiguous range of tiles: the taps of a transposed conv re-read a source row at neighbouring
// positions, which then hit that XCD's L2.
constexpr int BG_NCB = 2;                       // 32-frame column blocks per tile

// PRO / EPI: compile-time prologue mode and epilogue flags of the hot launches (the instruction stream of the generic
// form -- runtime mode / flag tests around every k-step and every 4-row output group -- kept the waves ISSUING 47 % of
// their time at 2 waves per SIMD, i.e. the kernel was issue-bound at 2-3 TB/s); -1 = decided at run time (thin layers).
// FULL: every segment has a multiple of 16 channels and M is a multiple of 32 (no channel guards).
template <int NRT, int PRO, int EPI, bool FULL>
__global__ __launch_bounds__(256, 2) void bgemm_kernel(const trunet_bgemm_args a) {
    constexpr bool TWO = PRO < 0 || PRO == TRUNET_PRO_BNBWD;
    constexpr int KG = (TWO || NRT == 4) ? 4 : 8;     // NRT = 4: 128 accumulator registers, half the staging
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
    u32x4* Al = (u32x4*)smem_;                                          // [nrt_all][nks_total][64] A fragments
    const int nrt_all = (a.M + 31) >> 5;
    float* Cf = (float*)(Al + (size_t)nrt_all * a.nks_total * 64);      // [nks_total][2 octets][3][8] coefficients
    float* Ep = Cf + a.nks_total * 2 * 3 * 8;                           // [4][128] epilogue: bias, e0, e1, e2 (LDS, not 128
                                                                        // hoisted registers)
    // (wave through readfirstlane: without it the tile index, its division by P, the segment positions and every row base
    // address were per-LANE integer arithmetic -- 22 quarter-rate v_mul_lo_u32 per tile -- and `continue` on an invalid
    // segment an exec-mask branch)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const int epi = EPI >= 0 ? EPI : a.epi;
    auto has = [&](int f) { return EPI >= 0 ? (EPI & f) != 0 : (epi & f) != 0; };

    for (int i = tid; i < nrt_all * a.nks_total * 64; i += 256) Al[i] = ((const u32x4*)a.wfrag)[i];
    // coefficient image per (k-step, octet half, {c0,c1,c2}, channel): zero coefficients cover padded channels
    for (int i = tid; i < a.nks_total * 2 * 3 * 8; i += 256) Cf[i] = 0.f;
    for (int i = tid; i < 128; i += 256) {
        const bool ok = i < a.M;
        Ep[i] = (ok && has(TRUNET_EPI_BIAS)) ? a.bias[i + (has(TRUNET_EPI_F32OUT) ? a.m_out_off : 0)] : 0.f;
        Ep[128 + i] = (ok && has(TRUNET_EPI_MASK)) ? a.e0[i] : 0.f;
        Ep[256 + i] = (ok && has(TRUNET_EPI_MASK)) ? a.e1[i] : 0.f;
        Ep[384 + i] = (ok && has(TRUNET_EPI_MASK) && a.e2) ? a.e2[i] : 0.f;
    }
    __syncthreads();
    for (int s = 0; s < a.nseg; ++s) {
        const trunet_bseg& sg = a.seg[s];
        const int noct = (sg.nchan + 7) >> 3;
        for (int i = tid; i < noct * 8; i += 256) {
            const int oct = i >> 3, j = i & 7, ks = sg.kstep0 + (oct >> 1), hh = oct & 1;
            float* cf = Cf + ((ks * 2 + hh) * 3) * 8 + j;
            const bool real = i < sg.nchan;
            // a TRUNET_PRO_NONE segment inside a BN+ReLU launch (a post-ReLU source: host contract) gets the identity
            if (sg.mode == TRUNET_PRO_NONE) { cf[0] = real ? 1.f : 0.f; cf[8] = 0.f; cf[16] = 0.f; }
            else if (sg.mode == TRUNET_PRO_BNRELU) { cf[0] = real ? sg.c0[i] : 0.f; cf[8] = real ? sg.c1[i] : 0.f; cf[16] = 0.f; }
            else { cf[0] = real ? sg.c0[i] : 0.f; cf[8] = real ? sg.c1[i] : 0.f; cf[16] = real ? sg.c2[i] : 0.f; }
        }
    }
    __syncthreads();

    const int split = NRT < 4 && nrt_all > 2;           // two waves share a tile (rows split); NRT = 4: one wave, all rows
    const int rt0 = split ? 2 * (wave & 1) : 0;
    const int nrt = FULL ? NRT : min(NRT, nrt_all - rt0);
    const int tslot = split ? (wave >> 1) : wave, tslots = split ? 2 : 4;
    const int nfc = a.NP / (32 * BG_NCB);
    const int total = a.P * nfc;
    const int vb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);      // XCD-contiguous numbering
    const int moct = (a.M + 7) >> 3;

    // statistics: per tile the 16 values of a row tile are summed over the 64 frames of the tile (two column blocks in
    // registers, then a butterfly over the 32 lanes that leaves value r(c) in lane c: 1 live register instead of 16)
    float sacc[NRT][2];
#pragma unroll
    for (int t = 0; t < NRT; ++t) { sacc[t][0] = 0.f; sacc[t][1] = 0.f; }

    for (int tile = vb * tslots + tslot; tile < total; tile += gridDim.x * tslots) {
        const int chunk = tile / a.P;
        const int p = a.p_begin + (tile - chunk * a.P);
        const int n0 = chunk * 32 * BG_NCB;
        f32x16 acc[NRT][BG_NCB];
#pragma unroll
        for (int t = 0; t < NRT; ++t)
#pragma unroll
            for (int cb = 0; cb < BG_NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][cb][r] = 0.f;

        // epilogue operands (ReLU-mask source, gradient to accumulate onto) requested BEFORE the k loop: at their point of
        // use they were eight dependent HBM round trips per tile (one per 4-row group), the whole time of the thin
        // data-gradient launches (K = 8: decoder.5)
        u32x2 zpre[NRT][4][BG_NCB], opre[NRT][4][BG_NCB];
        if (has(TRUNET_EPI_MASK) || has(TRUNET_EPI_ACCUM)) {
#pragma unroll
            for (int t = 0; t < NRT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int cb = 0; cb < BG_NCB; ++cb) {
                        const int og = min((rt0 + t) * 4 + g, moct - 1);                 // rows past M: a valid octet, unused
                        const size_t eidx = (((size_t)og * a.out_L + p + a.out_pos_off) * a.NP + n0 + 32 * cb + c) * 2 + h;
                        if (has(TRUNET_EPI_MASK)) zpre[t][g][cb] = ((const u32x2*)a.zmask)[eidx];
                        if (has(TRUNET_EPI_ACCUM)) opre[t][g][cb] = ((const u32x2*)a.out)[eidx];
                    }
        }
        for (int s = 0; s < a.nseg; ++s) {
            const trunet_bseg& sg = a.seg[s];
            const BSegPos sp = bseg_pos(sg, p);
            if (!sp.valid) continue;
            const int noct = (sg.nchan + 7) >> 3;
            const int nks = (noct + 1) >> 1;
            const size_t ostride = (size_t)sg.L * a.NP;                              // u32x4 elements between octets
            const u32x4* b0 = (const u32x4*)sg.src0 + (size_t)sp.q * a.NP + n0 + c + (size_t)h * ostride;
            const u32x4* b1 = (const u32x4*)sg.src1 + (size_t)sp.q * a.NP + n0 + c + (size_t)h * ostride;
            const int mode = PRO >= 0 ? PRO : sg.mode;
            for (int ks0 = 0; ks0 < nks; ks0 += KG) {
                u32x4 r0[KG][BG_NCB], r1[TWO ? KG : 1][BG_NCB];
#pragma unroll
                for (int j = 0; j < KG; ++j) {
                    const bool ok = FULL ? (ks0 + j < nks) : (2 * (ks0 + j) + h < noct);
                    const u32x4 z4 = {0u, 0u, 0u, 0u};
                    const size_t o = (size_t)(2 * (ks0 + j)) * ostride;
#pragma unroll
                    for (int cb = 0; cb < BG_NCB; ++cb) {
                        if (FULL) {
                            if (ok) {       // wave-uniform
                                r0[j][cb] = BG_LD(b0 + o + 32 * cb);
                                if constexpr (TWO) if (mode == TRUNET_PRO_BNBWD) r1[j][cb] = BG_LD(b1 + o + 32 * cb);
                            }
                        } else {
                            r0[j][cb] = ok ? b0[o + 32 * cb] : z4;
                            if constexpr (TWO) r1[j][cb] = (ok && mode == TRUNET_PRO_BNBWD) ? b1[o + 32 * cb] : z4;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < KG; ++j) {
                    asm volatile("" ::: "memory");          // keep the coefficient / fragment LDS reads of a k-step with it
                    if (ks0 + j < nks) {
                        const int ks = sg.kstep0 + ks0 + j;
                        const float* cf = Cf + ((ks * 2 + h) * 3) * 8;
                        bf16x8 bfrag[BG_NCB];
#pragma unroll
                        for (int cb = 0; cb < BG_NCB; ++cb) {
                            u32x4 fr = r0[j][cb];                                       // NONE: raw operand, no unpack / repack
                            if constexpr (TWO) {
                                if (mode == TRUNET_PRO_BNBWD) fr = bf_bnbwd8(fr, r1[j][cb], cf, cf + 8, cf + 16);
                            }
                            if (mode == TRUNET_PRO_BNRELU) fr = bf_affine8<true>(fr, cf, cf + 8);
                            bfrag[cb] = __builtin_bit_cast(bf16x8, fr);
                        }
#pragma unroll
                        for (int t = 0; t < NRT; ++t) {
                            if (t < nrt) {
                                const bf16x8 afrag = __builtin_bit_cast(bf16x8, Al[((size_t)(rt0 + t) * a.nks_total + ks) * 64 + lane]);
#pragma unroll
                                for (int cb = 0; cb < BG_NCB; ++cb)
                                    acc[t][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag[cb], acc[t][cb], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        // ---- epilogue: bias, accumulate, ReLU mask / ReLU, bf16 store (4 channels = 8 bytes per lane and octet), statistics
        // of the ROUNDED values
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
            if (t >= nrt) continue;
            float st1[16], st2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { st1[r] = 0.f; st2[r] = 0.f; }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m0 = (rt0 + t) * 32 + 8 * g + 4 * h;          // first of this lane's 4 rows
                if (!FULL && m0 >= a.M) continue;
                asm volatile("" ::: "memory");
                f32x4 bv = {0.f, 0.f, 0.f, 0.f}, e0v = bv, e1v = bv, muv = bv;
                if (has(TRUNET_EPI_BIAS)) bv = *(const f32x4*)(Ep + m0);
                if (has(TRUNET_EPI_MASK)) {
                    e0v = *(const f32x4*)(Ep + 128 + m0);
                    e1v = *(const f32x4*)(Ep + 256 + m0);
                    if (has(TRUNET_EPI_STATS)) muv = *(const f32x4*)(Ep + 384 + m0);
                }
#pragma unroll
                for (int cb = 0; cb < BG_NCB; ++cb) {
                    const int nn = n0 + 32 * cb + c;
                    const size_t eidx = (((size_t)((rt0 + t) * 4 + g) * a.out_L + p + a.out_pos_off) * a.NP + nn) * 2 + h;
                    float val[4], zv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i) val[i] = acc[t][cb][4 * g + i] + bv[i];
                    if (has(TRUNET_EPI_ACCUM)) {
                        const u32x2 o = opre[t][g][cb];
                        val[0] += bf_lo(o[0]); val[1] += bf_hi(o[0]); val[2] += bf_lo(o[1]); val[3] += bf_hi(o[1]);
                    }
                    if (has(TRUNET_EPI_MASK)) {
                        const u32x2 zz = zpre[t][g][cb];
                        zv[0] = bf_lo(zz[0]); zv[1] = bf_hi(zz[0]); zv[2] = bf_lo(zz[1]); zv[3] = bf_hi(zz[1]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) val[i] = (fmaf(e0v[i], zv[i], e1v[i]) > 0.f) ? val[i] : 0.f;
                    }
                    if (has(TRUNET_EPI_RELU)) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) val[i] = fmaxf(val[i], 0.f);
                    }
                    if (!FULL) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (m0 + i >= a.M) val[i] = 0.f;   // padded channels of the last octet
                    }
                    if (has(TRUNET_EPI_F32OUT)) {      // fp32 frames-last rows (GRU input projection): 128 B per row and half-wave
                        float* o32 = (float*)a.out + ((size_t)(a.m_out_off + m0) * a.out_L + p + a.out_pos_off) * a.NP + nn;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (FULL || m0 + i < a.M) o32[(size_t)i * a.out_L * a.NP] = val[i];
                        continue;
                    }
                    u32x2 o;
                    o[0] = bf_pack(val[0], val[1]);
                    o[1] = bf_pack(val[2], val[3]);
#ifdef TRUNET_BG_NT
                    __builtin_nontemporal_store(o, (u32x2*)a.out + eidx);
#else
                    ((u32x2*)a.out)[eidx] = o;
#endif
                    if (has(TRUNET_EPI_STATS)) {
                        const float rv[4] = {bf_lo(o[0]), bf_hi(o[0]), bf_lo(o[1]), bf_hi(o[1])};
                        const bool fin = nn < a.N;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float x = fin ? rv[i] : 0.f;
                            st1[4 * g + i] += x;
                            if (has(TRUNET_EPI_MASK)) st2[4 * g + i] = fmaf(x, zv[i] - muv[i], st2[4 * g + i]);
                            else st2[4 * g + i] = fmaf(x, x, st2[4 * g + i]);
                        }
                    }
                }
            }
            if (has(TRUNET_EPI_STATS)) {
                sacc[t][0] += butterfly16(st1, c);
                sacc[t][1] += butterfly16(st2, c);
            }
        }
    }
    if (has(TRUNET_EPI_STATS)) {
        float* pp = a.partials + ((size_t)(blockIdx.x * 4 + wave) * a.M_stat) * 2;
        const int r = butterfly16_index(c);
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
            if (t >= nrt) continue;
            const int m = (rt0 + t) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            if (butterfly16_writer(c) && m < a.M) { pp[2 * m] = sacc[t][0]; pp[2 * m + 1] = sacc[t][1]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight packing: fp32 master weight -> bf16 A fragments
struct PackDesc { int nchan[TRUNET_MAX_SEG], woff[TRUNET_MAX_SEG], ks0[TRUNET_MAX_SEG]; int nseg, nks_total; };

__global__ void pack_weight_kernel(const float* __restrict__ W, u32x4* __restrict__ out, int M, int ldw_m, int ldw_c,
                                   int w_m_off, const PackDesc d) {
    const int i = blockIdx.x * 256 + threadIdx.x;           // (rt, ks, lane)
    const int nrt = (M + 31) >> 5;
    if (i >= nrt * d.nks_total * 64) return;
    const int lane = i & 63, ks = (i >> 6) % d.nks_total, rt = (i >> 6) / d.nks_total;
    const int m = rt * 32 + (lane & 31);
    int s = 0;
    for (int t = 1; t < d.nseg; ++t) if (ks >= d.ks0[t]) s = t;
    const int cbase = (ks - d.ks0[s]) * 16 + 8 * (lane >> 5);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = cbase + j;
        f[j] = (m < M && ch < d.nchan[s]) ? W[(size_t)(m + w_m_off) * ldw_m + (size_t)ch * ldw_c + d.woff[s]] : 0.f;
    }
    out[i] = bf_pack8(f);
}

// every packed image of a step in ONE launch (blockIdx.y = descriptor): the ~40 per-GEMM pack launches of a training step were
// 0.28 ms of 7-microsecond kernels
__global__ void pack_weights_batch_kernel(const trunet_bpack_desc* __restrict__ descs) {
    const trunet_bpack_desc& d = descs[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;           // (rt, ks, lane)
    const int nrt = (d.M + 31) >> 5;
    if (i >= nrt * d.nks_total * 64) return;
    const int lane = i & 63, ks = (i >> 6) % d.nks_total, rt = (i >> 6) / d.nks_total;
    const int m = rt * 32 + (lane & 31);
    int s = 0;
    for (int t = 1; t < d.nseg; ++t) if (ks >= d.ks0[t]) s = t;
    const int cbase = (ks - d.ks0[s]) * 16 + 8 * (lane >> 5);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = cbase + j;
        f[j] = (m < d.M && ch < d.nchan[s]) ? d.W[(size_t)(m + d.w_m_off) * d.ldw_m + (size_t)ch * d.ldw_c + d.woff[s]] : 0.f;
    }
    ((u32x4*)d.out)[i] = bf_pack8(f);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient.  The MFMA K axis is the frame axis, so both operands need 8 consecutive FRAMES of one channel per lane
// while the octet layout delivers 8 CHANNELS of one frame.  A step is (position p, 64 frames): wave w loads the octets
// w, w + 8, ... of dy (, z) and of the valid source segments -- 1 KiB contiguous per wave-load, every operand row read from
// HBM exactly once per step -- applies the prologue (BatchNorm backward on dz, BN + ReLU on the sources, frames >= N
// zeroed) and writes the result to LDS as [octet][frame][8 channels] (octet stride 64 frames * 16 B + 64 B, which makes
// the transposed reads conflict-free).  ds_read_b64_tr_b16 then hands every lane 4 frames of one channel: two reads are
// an MFMA fragment.  The loads of steps t + 1 and t + 2 are in flight in registers while step t's MFMAs run (the kernel
// is bound by how many bytes a CU keeps in flight: one step deep measured 1.5 TB/s); LDS is double-buffered, one raw
// barrier per step.
constexpr int BW_THREADS = 512;
constexpr int BW_MAXD = 2;                       // dz octets per wave (M <= 128)
constexpr int BW_MAXS = 5;                       // source octets per wave (<= 40 over all segments)
constexpr int BW_MAXT = 3;                       // output tiles per wave

// TWO: dz = BatchNorm backward of (dy, z); SMODE: prologue of the sources (TRUNET_PRO_NONE / TRUNET_PRO_BNRELU; a plain
// segment in a BN+ReLU launch must be a post-ReLU source and gets identity coefficients).  Coefficients live in LDS
// ([octet][set][8] floats, broadcast reads): hoisted into scalar registers they spilled (318 SGPRs) and cost a v_readlane
// per use.
//
// DG (trunet_bf16_pw_bwd): the data gradient of a pointwise layer in the same pass -- dsrc_s[c][p + off_s] = sum_m W[m][c] dz[m][p]
// per source s, with the dz image as the MFMA B operand (an [octet][frame][8] row IS a B fragment: one ds_read_b128) and
// W^T as A fragments in LDS; ReLU mask / statistics from a RAW copy of the source octets kept next to the activated one
// (bitwise the arithmetic of the separate trunet_bf16_gemm launch), accumulate / store straight to HBM.  (dy, z) are read
// once instead of twice and the BatchNorm-backward prologue runs once.
template <bool TWO, int SMODE, bool DG>
__global__ __launch_bounds__(BW_THREADS, 1) void bwgrad_kernel(const trunet_bwgrad_args a, int soct_total,
                                                               const trunet_bdgrad_args dg) {
    constexpr int BW_MAXS = DG ? 3 : 5;             // source octets per wave (pointwise layers: <= 24 octets)
    constexpr int BW_MAXT = DG ? 2 : 3;             // weight-gradient tiles per wave
    constexpr int BW_MAXG = 2;                      // data-gradient tiles per wave: (32 source channels, 32 frames)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int moct = (a.M + 7) >> 3;
    const int nrt = (a.M + 31) >> 5;
    constexpr bool two = TWO;
    // one buffer: dz octets, the (activated) source octets, and with DG the raw source octets
    const int img_bytes = (moct + (DG ? 2 : 1) * soct_total) * BW_OS;
    float* Cd = (float*)(smem_ + 2 * img_bytes);                 // [moct][3][8]  ca, cb, cc of dz (zero beyond M)
    float* Cs = Cd + moct * 24;                                  // [source octet][2][8] scale, shift
    float* Cm = Cs + soct_total * 16;                            // DG: [source octet][8] mean
    u32x4* WT = (u32x4*)(Cm + (DG ? soct_total * 8 : 0));        // DG: W^T A fragments [source row tile][k-step][64]
    const int nks_dz = (moct + 1) >> 1;
    if constexpr (DG) {
        for (int i = tid; i < dg.nrt_total * nks_dz * 64; i += BW_THREADS) WT[i] = ((const u32x4*)dg.wfragT)[i];
        int base = 0;
        for (int s = 0; s < a.nseg; ++s) {
            const int no = (a.seg[s].nchan + 7) >> 3;
            for (int i = tid; i < no * 8; i += BW_THREADS)
                Cm[base * 8 + i] = (dg.mean[s] && i < a.seg[s].nchan) ? dg.mean[s][i] : 0.f;
            base += no;
        }
    }
    for (int i = tid; i < moct * 8; i += BW_THREADS) {
        const int oct = i >> 3, e = i & 7;
        const bool ok = i < a.M;
        Cd[oct * 24 + e] = two ? (ok ? a.ac0[i] : 0.f) : (ok ? 1.f : 0.f);
        Cd[oct * 24 + 8 + e] = (two && ok) ? a.ac1[i] : 0.f;
        Cd[oct * 24 + 16 + e] = (two && ok) ? a.ac2[i] : 0.f;
    }
    {
        int base = 0;
        for (int s = 0; s < a.nseg; ++s) {
            const trunet_bseg& sg = a.seg[s];
            const int no = (sg.nchan + 7) >> 3;
            for (int i = tid; i < no * 8; i += BW_THREADS) {
                const int oct = i >> 3, e = i & 7;
                const bool ok = i < sg.nchan;
                const bool bn = sg.mode == TRUNET_PRO_BNRELU;
                Cs[(base + oct) * 16 + e] = ok ? (bn ? sg.c0[i] : 1.f) : 0.f;
                Cs[(base + oct) * 16 + 8 + e] = (ok && bn) ? sg.c1[i] : 0.f;
            }
            base += no;
        }
    }
    // ---- this wave's operand octets
    int soct_seg[BW_MAXS], soct_idx[BW_MAXS], soct_g[BW_MAXS];   // segment, octet within the segment, global source octet
#pragma unroll
    for (int j = 0; j < BW_MAXS; ++j) {
        const int g = wave + 8 * j;
        soct_seg[j] = -1; soct_idx[j] = 0; soct_g[j] = g;
        int base = 0;
        for (int s = 0; s < a.nseg; ++s) {
            const int no = (a.seg[s].nchan + 7) >> 3;
            if (g >= base && g < base + no) { soct_seg[j] = s; soct_idx[j] = g - base; }
            base += no;
        }
    }
    // ---- this wave's output tiles: g = rt * nct_total + (segment, c tile), taken round-robin
    int t_rt[BW_MAXT], t_seg[BW_MAXT], t_ct[BW_MAXT], t_oct0[BW_MAXT];
    {
        int nct_total = 0;
        for (int s = 0; s < a.nseg; ++s) nct_total += (a.seg[s].nchan + 31) >> 5;
#pragma unroll
        for (int i = 0; i < BW_MAXT; ++i) {
            const int g = wave + 8 * i;
            t_rt[i] = -1; t_seg[i] = 0; t_ct[i] = 0; t_oct0[i] = 0;
            if (g < nrt * nct_total) {
                t_rt[i] = g / nct_total;
                int cg = g - t_rt[i] * nct_total, obase = 0;
                for (int s = 0; s < a.nseg; ++s) {
                    const int nct = (a.seg[s].nchan + 31) >> 5;
                    if (cg >= 0 && cg < nct) { t_seg[i] = s; t_ct[i] = cg; t_oct0[i] = obase + 4 * cg; }
                    cg -= nct;
                    obase += (a.seg[s].nchan + 7) >> 3;
                }
            }
        }
    }
    // ---- DG: this wave's data-gradient tiles g = (7 - wave) + 8 i -> (source row tile g / 2, frame half g % 2): the weight-
    // gradient tiles go round-robin from wave 0, so with 12 + 12 tiles (decoder layers) every wave gets three, not 4 / 2
    int d_seg[BW_MAXG], d_rtl[BW_MAXG], d_rtg[BW_MAXG], d_og[BW_MAXG];      // segment, row tile in it / overall, first octet
    float sacc[BW_MAXG][2];
#pragma unroll
    for (int i = 0; i < BW_MAXG; ++i) {
        d_seg[i] = -1; d_rtl[i] = 0; d_rtg[i] = 0; d_og[i] = 0; sacc[i][0] = 0.f; sacc[i][1] = 0.f;
        if constexpr (DG) {
            const int g = (7 - wave) + 8 * i, rtg = g >> 1;     // reversed: the waves with fewer weight-gradient tiles take more
            int rbase = 0, obase = 0;
            for (int s = 0; s < a.nseg; ++s) {
                const int nr = a.seg[s].nchan >> 5;
                if (rtg >= rbase && rtg < rbase + nr) { d_seg[i] = s; d_rtl[i] = rtg - rbase; d_rtg[i] = rtg; d_og[i] = obase + 4 * (rtg - rbase); }
                rbase += nr;
                obase += a.seg[s].nchan >> 3;
            }
        }
    }
    f32x16 acc[BW_MAXT];
#pragma unroll
    for (int i = 0; i < BW_MAXT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float bsum[BW_MAXD][8];
#pragma unroll
    for (int j = 0; j < BW_MAXD; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[j][e] = 0.f;

    const int nf = a.NP / BW_F;
    const int total = a.P * nf;
    const int s_begin = (int)(((long long)blockIdx.x * total) / gridDim.x);
    const int s_end = (int)(((long long)(blockIdx.x + 1) * total) / gridDim.x);

    struct Stage { u32x4 dy[BW_MAXD], z[BW_MAXD], s[BW_MAXS]; };
    Stage sa, sb;                                 // two steps of operand loads in flight
    // a step is (chunk, p); segment positions come from per-piece / per-tile constants hoisted out of the loop (strides are
    // 1 or 2: shifts, no integer division -- the three phases of a step used to redo ~16 divisions per step, ~1000 scalar
    // instructions, the bulk of the kernel's issue slots)
    struct Info { int chunk, p; };
    struct PosC { int mul, off, sh, L; };
    auto posc = [&](int sidx) {
        PosC c;
        const trunet_bseg& sg = a.seg[max(sidx, 0)];
        c.mul = sg.pos_mul; c.off = sg.pos_off; c.sh = sg.pos_div >> 1; c.L = sg.L;       // pos_div in {1, 2} (host-checked)
        return c;
    };
    auto pos = [&](const PosC& c, int p, bool& valid, int& q) {
        const int qn = p * c.mul + c.off;
        q = qn >> c.sh;
        valid = (qn >= 0) && ((qn & c.sh) == 0) && (q < c.L);
    };
    PosC pc_s[BW_MAXS], pc_t[BW_MAXT], pc_d[BW_MAXG];
#pragma unroll
    for (int j = 0; j < BW_MAXS; ++j) pc_s[j] = posc(soct_seg[j]);
#pragma unroll
    for (int i = 0; i < BW_MAXT; ++i) pc_t[i] = posc(t_seg[i]);
#pragma unroll
    for (int i = 0; i < BW_MAXG; ++i) pc_d[i] = posc(d_seg[i]);
    // Everything a step needs from the argument block is read ONCE here: inside the step loop a.seg[<runtime index>] /
    // dg.out[<runtime index>] are scalar loads from the kernarg segment, each followed by s_waitcnt lgkmcnt(0) -- which also
    // waits for every LDS access in flight -- about ten times per step (flags, output pointer and row length per quarter tile).
    const u32x4* s_ptr[BW_MAXS];                  // row 0 of this wave's source octet j
    bool s_plain[BW_MAXS];                        // whole octets of a source without prologue: copied as they are
#pragma unroll
    for (int j = 0; j < BW_MAXS; ++j) {
        const trunet_bseg& sg = a.seg[max(soct_seg[j], 0)];
        s_ptr[j] = (const u32x4*)sg.src0 + (size_t)soct_idx[j] * sg.L * a.NP;
        s_plain[j] = SMODE == TRUNET_PRO_NONE && (sg.nchan & 7) == 0;
    }
    int d_flags[BW_MAXG], d_gstride[BW_MAXG];     // TRUNET_DG_* of the tile's segment; u32x2 elements between its octets
    u32x2* d_out[BW_MAXG];                        // (first octet of the tile, position 0, frame 0, half 0) of its output
#pragma unroll
    for (int i = 0; i < BW_MAXG; ++i) {
        d_flags[i] = 0; d_gstride[i] = 0; d_out[i] = nullptr;
        if constexpr (DG) {
            if (d_seg[i] >= 0) {
                const trunet_bseg& sg = a.seg[d_seg[i]];
                d_flags[i] = dg.flags[d_seg[i]];
                d_gstride[i] = sg.L * a.NP * 2;
                d_out[i] = (u32x2*)dg.out[d_seg[i]] + (size_t)(d_rtl[i] * 4) * sg.L * a.NP * 2;
            }
        }
    }
    auto issue = [&](const Info& f, Stage& r) {
        const int p = f.p;
        const size_t n = (size_t)f.chunk * BW_F + lane;
#pragma unroll
        for (int j = 0; j < BW_MAXD; ++j) {
            const int oct = wave + 8 * j;
            if (oct < moct) {
                const size_t idx = ((size_t)oct * a.a_L + p + a.a_pos_off) * a.NP + n;
                r.dy[j] = BW_LD((const u32x4*)a.a0 + idx);
                if (two) r.z[j] = BW_LD((const u32x4*)a.a1 + idx);
            }
        }
#pragma unroll
        for (int j = 0; j < BW_MAXS; ++j) {
            if (soct_seg[j] >= 0) {
                bool valid; int q;
                pos(pc_s[j], f.p, valid, q);
                if (valid) r.s[j] = BW_LD(s_ptr[j] + (size_t)q * a.NP + n);
            }
        }
    };
    // DG: request the gradient already stored where this step's data-gradient tiles accumulate (skip connections).  Issued one
    // step ahead of its use: read at the point of use it put one full HBM latency into every step of the five encoder launches
    auto issue_oin = [&](const Info& f, u32x2 (&o)[DG ? BW_MAXG : 1][4]) {
        if constexpr (DG) {
            const int h = lane >> 5, c = lane & 31;
#pragma unroll
            for (int i = 0; i < BW_MAXG; ++i) {
                if (d_seg[i] < 0 || !(d_flags[i] & 8)) continue;
#if defined(BW_ABL) && (BW_ABL & 64)     // diagnostic: the accumulate operands are not loaded
                if (a.N >= 0) continue;
#endif
                bool valid; int q;
                pos(pc_d[i], f.p, valid, q);
                if (!valid) continue;
                const int cb = ((7 - wave) + 8 * i) & 1;
                const int nn = f.chunk * BW_F + 32 * cb + c;
                const u32x2* op = d_out[i] + ((size_t)q * a.NP + nn) * 2 + h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) o[i][g4] = op[(size_t)g4 * d_gstride[i]];
            }
        }
    };
    // FULL: all 64 frames of the step are real frames (every chunk but the last one or two): no per-lane frame selects
    auto store = [&](auto full_tag, const Info& f, unsigned char* buf, const Stage& r) {
        constexpr bool FULL = decltype(full_tag)::value;
        const bool fin = FULL || f.chunk * BW_F + lane < a.N;
#pragma unroll
        for (int j = 0; j < BW_MAXD; ++j) {
            const int oct = wave + 8 * j;
            if (oct < moct) {
#if defined(BW_ABL) && (BW_ABL & 1)      // diagnostic (wrong results): no BatchNorm-backward prologue
                *(u32x4*)(buf + oct * BW_OS + lane * 16) = r.dy[j] ^ r.z[j];
                continue;
#endif
                float v[8], w[8];
                bf_unpack8(r.dy[j], v);
                const float* cd = Cd + oct * 24;
                if (two) {
                    bf_unpack8(r.z[j], w);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(cd[e], v[e], fmaf(cd[8 + e], w[e], cd[16 + e]));
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= cd[e];          // 1, or 0 for the padded channels of the last octet
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (!FULL) v[e] = fin ? v[e] : 0.f;
                    bsum[j][e] += v[e];
                }
                *(u32x4*)(buf + oct * BW_OS + lane * 16) = bf_pack8(v);
            }
        }
#pragma unroll
        for (int j = 0; j < BW_MAXS; ++j) {
            if (soct_seg[j] >= 0) {
                bool valid; int q;
                pos(pc_s[j], f.p, valid, q);
                if (valid) {
                    if constexpr (DG) *(u32x4*)(buf + (moct + soct_total + soct_g[j]) * BW_OS + lane * 16) = r.s[j];
#if defined(BW_ABL) && (BW_ABL & 2)      // diagnostic (wrong results): no source prologue
                    if (true) {
#else
                    if (s_plain[j]) {
#endif
                        *(u32x4*)(buf + (moct + soct_g[j]) * BW_OS + lane * 16) = r.s[j];       // raw operand
                    } else {
                        const float* cs = Cs + soct_g[j] * 16;
                        *(u32x4*)(buf + (moct + soct_g[j]) * BW_OS + lane * 16) =
                            bf_affine8<SMODE != TRUNET_PRO_NONE>(r.s[j], cs, cs + 8);
                    }
                }
            }
        }
    };
    auto mma = [&](const Info& f, const unsigned char* buf, const u32x2 (&oin)[DG ? BW_MAXG : 1][4]) {
#pragma unroll
        for (int i = 0; i < BW_MAXT; ++i) {
            if (t_rt[i] < 0) continue;
            bool valid; int q;
            pos(pc_t[i], f.p, valid, q);
            if (!valid) continue;
#if defined(BW_ABL) && (BW_ABL & 4)      // diagnostic: no weight-gradient MFMAs
            continue;
#endif
#pragma unroll
            for (int kk = 0; kk < BW_F / 16; ++kk) {
                const bf16x8 af = lds_frag(buf, 4 * t_rt[i], 16 * kk, lane);
                const bf16x8 bfr = lds_frag(buf, moct + t_oct0[i], 16 * kk, lane);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[i], 0, 0, 0);
            }
        }
        if constexpr (DG) {
            const int h = lane >> 5, c = lane & 31;
#pragma unroll
            for (int i = 0; i < BW_MAXG; ++i) {
                if (d_seg[i] < 0) continue;
                bool valid; int q;
                pos(pc_d[i], f.p, valid, q);
                if (!valid) continue;
                const int cb = ((7 - wave) + 8 * i) & 1;
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
                const unsigned char* bcol = buf + h * BW_OS + (32 * cb + c) * 16;
                for (int ks = 0; ks < nks_dz; ++ks) {
                    const bf16x8 af = __builtin_bit_cast(bf16x8, WT[(d_rtg[i] * nks_dz + ks) * 64 + lane]);
                    const bf16x8 bfr = __builtin_bit_cast(bf16x8, *(const u32x4*)(bcol + 2 * ks * BW_OS));
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, d, 0, 0, 0);
                }
                const int flags = d_flags[i];
                const int nn = f.chunk * BW_F + 32 * cb + c;
                u32x2* outp = d_out[i] + ((size_t)q * a.NP + nn) * 2 + h;
                const bool fin = nn < a.N;
                float st1[16], st2[16];
#if defined(BW_ABL) && (BW_ABL & 16)     // diagnostic: no data-gradient epilogue (one guarded store keeps the MFMAs alive)
                if (a.N < 0) ((float*)dg.out[0])[lane] = d[0] + d[5] + d[10] + d[15];
                continue;
#endif
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int og = d_og[i] + g4;
                    const size_t eidx = (size_t)g4 * d_gstride[i];
                    float val[4], zv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = d[4 * g4 + e];
                    if (flags & 8) {        // accumulate onto the gradient already stored there (skip connection)
                        const u32x2 o = oin[i][g4];
                        val[0] += bf_lo(o[0]); val[1] += bf_hi(o[0]); val[2] += bf_lo(o[1]); val[3] += bf_hi(o[1]);
                    }
                    f32x4 muv = {0.f, 0.f, 0.f, 0.f};
                    if (flags & 2) {        // ReLU mask from the raw source
                        const u32x2 zz = *(const u32x2*)(buf + (moct + soct_total + og) * BW_OS + (32 * cb + c) * 16 + 8 * h);
                        zv[0] = bf_lo(zz[0]); zv[1] = bf_hi(zz[0]); zv[2] = bf_lo(zz[1]); zv[3] = bf_hi(zz[1]);
                        const f32x4 e0v = *(const f32x4*)(Cs + og * 16 + 4 * h), e1v = *(const f32x4*)(Cs + og * 16 + 8 + 4 * h);
                        muv = *(const f32x4*)(Cm + og * 8 + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) val[e] = (fmaf(e0v[e], zv[e], e1v[e]) > 0.f) ? val[e] : 0.f;
                    }
                    u32x2 o;
                    o[0] = bf_pack(val[0], val[1]);
                    o[1] = bf_pack(val[2], val[3]);
#if defined(BW_ABL) && (BW_ABL & 32)     // diagnostic: the data gradient is computed but not stored
                    if (a.N < 0)
#endif
                    outp[eidx] = o;
                    const float rv[4] = {bf_lo(o[0]), bf_hi(o[0]), bf_lo(o[1]), bf_hi(o[1])};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = fin ? rv[e] : 0.f;
                        st1[4 * g4 + e] = x;
                        st2[4 * g4 + e] = x * (zv[e] - muv[e]);
                    }
                }
#if defined(BW_ABL) && (BW_ABL & 8)      // diagnostic: no statistics exchange
                if (a.N < 0)
#endif
                if (flags & 4) {
                    sacc[i][0] += butterfly16(st1, c);
                    sacc[i][1] += butterfly16(st2, c);
                }
            }
        }
    };

    // octets a 32-row / 32-channel tile reads beyond the real ones (M or nchan not a multiple of 32) only feed output rows /
    // columns that are discarded; both buffers are cleared once so that those reads are at least deterministic
    {
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (int i = tid; i < 2 * img_bytes / 16; i += BW_THREADS) ((u32x4*)smem_)[i] = z4;
    }
    __syncthreads();

    // BW_DEPTH steps of operand loads in flight in registers (the waves are parked on memory latency 59 % of the time at
    // depth 2); LDS stays double-buffered: buffer (step & 1)
#ifndef TRUNET_BW_DEPTH
#define TRUNET_BW_DEPTH 2
#endif
    constexpr int DEPTH = TRUNET_BW_DEPTH;
    static_assert(DEPTH == 2 || DEPTH == 3, "stages are named");
    Stage sc;
    Info nx;                                      // the next step to request: advanced incrementally (p fastest)
    nx.chunk = s_begin / a.P;
    nx.p = a.p_begin + (s_begin - nx.chunk * a.P);
    auto advance = [&](Info& f) { if (++f.p == a.p_begin + a.P) { f.p = a.p_begin; ++f.chunk; } };
    Info fa = nx, fb = nx, fc = nx;
    if (s_begin < s_end) { issue(fa, sa); advance(nx); }
    if (s_begin + 1 < s_end) { fb = nx; issue(fb, sb); advance(nx); }
    if (DEPTH == 3 && s_begin + 2 < s_end) { fc = nx; issue(fc, sc); advance(nx); }
    typedef u32x2 OIn[DG ? BW_MAXG : 1][4];
    OIn o0, o1, o2;                                // accumulate operands of the data-gradient tiles: current / next step
    if (DG && s_begin < s_end) issue_oin(fa, o0);
    // fnext: the step after this one (the next stage's Info: a stage is re-issued only inside its own step)
    auto step = [&](int st, Info& f, Stage& r, const Info& fnext, const OIn& ocur, OIn& onext) {
        unsigned char* buf = smem_ + ((st - s_begin) & 1) * img_bytes;
        const bool full = (f.chunk + 1) * BW_F <= a.N;        // uniform
        if (full) store(std::true_type{}, f, buf, r);
        else store(std::false_type{}, f, buf, r);
        const Info cur = f;
        if (st + DEPTH < s_end) { f = nx; issue(f, r); advance(nx); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (DG && st + 1 < s_end) issue_oin(fnext, onext);
        mma(cur, buf, ocur);
    };
    for (int st = s_begin; st < s_end; st += DEPTH) {
        if (DEPTH == 2) {
            step(st, fa, sa, fb, o0, o1);
            if (st + 1 < s_end) step(st + 1, fb, sb, fa, o1, o0);
        } else {
            step(st, fa, sa, fb, o0, o1);
            if (st + 1 < s_end) step(st + 1, fb, sb, fc, o1, o2);
            if (st + 2 < s_end) step(st + 2, fc, sc, fa, o2, o0);
        }
    }

    // ---- partial image of dW (rows m = dz channel, columns c = source channel) and db
    float* img = a.w_partials + (size_t)blockIdx.x * a.w_numel;
    const int cidx = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < BW_MAXT; ++i) {
        if (t_rt[i] < 0) continue;
        const trunet_bseg& sg = a.seg[t_seg[i]];
        const int ci = t_ct[i] * 32 + cidx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = t_rt[i] * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (m < a.M && ci < sg.nchan)
                img[(size_t)(m + a.w_m_off) * a.ldw_m + (size_t)ci * a.ldw_c + sg.woff] = acc[i][r];
        }
    }
    if constexpr (DG) {
        const int h = lane >> 5, c = lane & 31;
        const int r = butterfly16_index(c);
#pragma unroll
        for (int i = 0; i < BW_MAXG; ++i) {
            if (d_seg[i] < 0 || !(dg.flags[d_seg[i]] & 4)) continue;
            const int cb = ((7 - wave) + 8 * i) & 1;
            const int nch = a.seg[d_seg[i]].nchan;
            const int ch = d_rtl[i] * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            float* pp = dg.partials[d_seg[i]] + ((size_t)(blockIdx.x * 2 + cb) * nch + ch) * 2;
            if (butterfly16_writer(c)) { pp[0] = sacc[i][0]; pp[1] = sacc[i][1]; }
        }
    }
    if (a.b_partials) {
#pragma unroll
        for (int j = 0; j < BW_MAXD; ++j) {
            const int oct = wave + 8 * j;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = wave_sum(bsum[j][e]);
                const int m = oct * 8 + e;
                if (lane == 0 && oct < moct && m < a.M) a.b_partials[(size_t)blockIdx.x * a.b_stride + a.b_off + m] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// depthwise conv, octet layout.  A thread owns (octet = blockIdx.y, frame, chunk of positions = blockIdx.z) and walks the
// positions with a sliding window in registers: every input row is loaded once.
// every row is touched exactly once: non-temporal accesses (same-box A/B: -0.5 ms per step; -DTRUNET_DW_TEMPORAL builds the
// default-policy form)
#ifndef TRUNET_DW_TEMPORAL
#define BDW_LD(p) __builtin_nontemporal_load((const u32x4*)(p))
#define BDW_ST(p, v) __builtin_nontemporal_store((v), (u32x4*)(p))
#else
#define BDW_LD(p) (*(const u32x4*)(p))
#define BDW_ST(p, v) (*(u32x4*)(p) = (v))
#endif

template <int NV>
__device__ __forceinline__ void block_reduce_store(float (&v)[NV], float* smem /* [4][NV] */, float* dst, int stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float s = wave_sum(v[i]);
        if (lane == 0) smem[wave * NV + i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NV; i += 256)
        dst[(size_t)i * stride] = (smem[i] + smem[NV + i]) + (smem[2 * NV + i] + smem[3 * NV + i]);
    __syncthreads();
}

__host__ __device__ inline int bdw_chunks(int rows) { return rows >= 32 ? 4 : (rows >= 16 ? 2 : 1); }

template <int K, int S>
__global__ __launch_bounds__(256) void bdw_fwd_kernel(const u32x4* __restrict__ zin, const float* __restrict__ s_in,
                                                      const float* __restrict__ t_in, const float* __restrict__ w,
                                                      const float* __restrict__ b, u32x4* __restrict__ zout,
                                                      float* __restrict__ partials, int C, int Lin, int Lout, int NP, int N) {
    __shared__ float red[4 * 16];
    const int oct = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int nch = gridDim.z;
    const int lo0 = (int)(((long long)blockIdx.z * Lout) / nch), lo1 = (int)(((long long)(blockIdx.z + 1) * Lout) / nch);
    float sc[8], sh[8], bb[8], wk[8][K];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = oct * 8 + j;
        sc[j] = s_in[ch]; sh[j] = t_in[ch]; bb[j] = b[ch];
#pragma unroll
        for (int k = 0; k < K; ++k) wk[j][k] = w[ch * K + k];
    }
    float st[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) st[j] = 0.f;
    const u32x4* src = zin + (size_t)oct * Lin * NP + n;
    u32x4* dst = zout + (size_t)oct * Lout * NP + n;
    auto load_act = [&](int li, float (&act)[8]) {
        if (li >= 0 && li < Lin) {
            float v[8];
            bf_unpack8(BDW_LD(src + (size_t)li * NP), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) act[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), 0.f);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) act[j] = 0.f;
        }
    };
    float win[K][8];                    // win[k] = activated input at li = lo*S + k - K/2
#pragma unroll
    for (int k = 0; k < K - S; ++k) load_act(lo0 * S + k - K / 2, win[k + S]);       // pre-shifted: the loop shifts first
    for (int lo = lo0; lo < lo1; ++lo) {
#pragma unroll
        for (int k = 0; k < K - S; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) win[k][j] = win[k + S][j];
#pragma unroll
        for (int k = K - S; k < K; ++k) load_act(lo * S + k - K / 2, win[k]);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = bb[j];
#pragma unroll
            for (int k = 0; k < K; ++k) acc[j] = fmaf(wk[j][k], win[k][j], acc[j]);
        }
        const u32x4 o = bf_pack8(acc);
        BDW_ST(dst + (size_t)lo * NP, o);
        if (n < N) {
            float r[8];
            bf_unpack8(o, r);
#pragma unroll
            for (int j = 0; j < 8; ++j) { st[2 * j] += r[j]; st[2 * j + 1] = fmaf(r[j], r[j], st[2 * j + 1]); }
        }
    }
    const int part = blockIdx.z * gridDim.x + blockIdx.x;
    block_reduce_store<16>(st, red, partials + ((size_t)part * C + oct * 8) * 2, 1);
}

// backward: groups of S input positions li = m*S + e; the taps of a group touch the dz rows m + LOMIN .. m + LOMAX, which
// are kept (BatchNorm-backward transformed) in a 3-row register window and advance by one row per group.
template <int K, int S>
__global__ __launch_bounds__(256) void bdw_bwd_kernel(const u32x4* __restrict__ dy, const u32x4* __restrict__ z,
                                                      const float* __restrict__ ca, const float* __restrict__ cb,
                                                      const float* __restrict__ cc, const u32x4* __restrict__ zin,
                                                      const float* __restrict__ s_in, const float* __restrict__ t_in,
                                                      const float* __restrict__ mean_in, const float* __restrict__ w,
                                                      u32x4* __restrict__ dy_in, float* __restrict__ partials_in,
                                                      float* __restrict__ w_partials, float* __restrict__ b_partials, int C,
                                                      int Lin, int Lout, int NP, int N) {
    constexpr int LOMIN = -1, LOMAX = 1, W = 3;
    static_assert((K == 3 && S == 1) || (K == 5 && S == 2) || (K == 3 && S == 2), "window derived for these shapes");
    __shared__ float red[4 * 8 * (K + 1)];
    const int oct = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int G = (Lin + S - 1) / S;
    const int nch = gridDim.z;
    const int m0 = (int)(((long long)blockIdx.z * G) / nch), m1 = (int)(((long long)(blockIdx.z + 1) * G) / nch);
    float a0[8], a1[8], a2[8], sc[8], sh[8], mu[8], wk[8][K];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = oct * 8 + j;
        a0[j] = ca[ch]; a1[j] = cb[ch]; a2[j] = cc[ch];
        sc[j] = s_in[ch]; sh[j] = t_in[ch]; mu[j] = mean_in[ch];
#pragma unroll
        for (int k = 0; k < K; ++k) wk[j][k] = w[ch * K + k];
    }
    float st[16], dwb[8 * (K + 1)];       // dwb[j*(K+1) + k] = dw[j][k], [.. + K] = db[j]
#pragma unroll
    for (int j = 0; j < 16; ++j) st[j] = 0.f;
#pragma unroll
    for (int j = 0; j < 8 * (K + 1); ++j) dwb[j] = 0.f;
    const u32x4* pdy = dy + (size_t)oct * Lout * NP + n;
    const u32x4* pz = z + (size_t)oct * Lout * NP + n;
    const u32x4* pin = zin + (size_t)oct * Lin * NP + n;
    u32x4* pout = dy_in + (size_t)oct * Lin * NP + n;
    const bool fin = n < N;
    auto load_dz = [&](int lo, float (&d)[8]) {
        if (lo >= 0 && lo < Lout && fin) {
            float dv[8], zv[8];
            bf_unpack8(BDW_LD(pdy + (size_t)lo * NP), dv);
            bf_unpack8(BDW_LD(pz + (size_t)lo * NP), zv);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = fmaf(a0[j], dv[j], fmaf(a1[j], zv[j], a2[j]));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = 0.f;
        }
    };
    float dzw[W][8];                    // dzw[i] = dz row m + LOMIN + i
#pragma unroll
    for (int i = 1; i < W; ++i) load_dz(m0 + LOMIN + i - 1, dzw[i]);                 // pre-shifted
    for (int m = m0; m < m1; ++m) {
#pragma unroll
        for (int i = 0; i < W - 1; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) dzw[i][j] = dzw[i + 1][j];
        load_dz(m + LOMAX, dzw[W - 1]);
#pragma unroll
        for (int e = 0; e < S; ++e) {
            const int li = m * S + e;
            if (li >= Lin) continue;
            float zi[8], act[8], g[8];
            bf_unpack8(BDW_LD(pin + (size_t)li * NP), zi);
#pragma unroll
            for (int j = 0; j < 8; ++j) { act[j] = fmaxf(fmaf(zi[j], sc[j], sh[j]), 0.f); g[j] = 0.f; }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                constexpr int half = K / 2;
                const int num = e + half - k;                        // relative to m*S; compile-time after unrolling
                if (((num % S) + S) % S != 0) continue;
                const int lo_rel = (num >= 0) ? num / S : -((-num) / S);
                const int wi = lo_rel - LOMIN;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float dzv = dzw[wi][j];
                    g[j] = fmaf(wk[j][k], dzv, g[j]);
                    dwb[j * (K + 1) + k] = fmaf(dzv, act[j], dwb[j * (K + 1) + k]);
                    if (k == half) dwb[j * (K + 1) + K] += dzv;      // e == 0 here: every dz row counted once
                }
            }
            float ov[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = (act[j] > 0.f) ? g[j] : 0.f;
            const u32x4 o = bf_pack8(ov);
            BDW_ST(pout + (size_t)li * NP, o);
            float r[8];
            bf_unpack8(o, r);
#pragma unroll
            for (int j = 0; j < 8; ++j) { st[2 * j] += r[j]; st[2 * j + 1] = fmaf(r[j], zi[j] - mu[j], st[2 * j + 1]); }
        }
    }
    const int part = blockIdx.z * gridDim.x + blockIdx.x;
    block_reduce_store<16>(st, red, partials_in + ((size_t)part * C + oct * 8) * 2, 1);
    // dw / db: [part][C][K] and [part][C]
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 8 * (K + 1); ++i) {
            const float s = wave_sum(dwb[i]);
            if (lane == 0) red[wave * 8 * (K + 1) + i] = s;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 8 * (K + 1); i += 256) {
            const int NVV = 8 * (K + 1);
            const float s = (red[i] + red[NVV + i]) + (red[2 * NVV + i] + red[3 * NVV + i]);
            const int j = i / (K + 1), k = i - j * (K + 1);
            const int ch = oct * 8 + j;
            if (k < K) w_partials[((size_t)part * C + ch) * K + k] = s;
            else b_partials[(size_t)part * C + ch] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// layout changes: fp32 frames-last [C][R] (R = L*NP) <-> bf16 octets [C/8][R][8]
__global__ void from_fl_kernel(const float* __restrict__ x, u32x4* __restrict__ y, int C, size_t R) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // (oct, r)
    const int noct = (C + 7) >> 3;
    if (i >= (size_t)noct * R) return;
    const size_t r = i % R;
    const int oct = (int)(i / R);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = oct * 8 + j;
        f[j] = ch < C ? x[(size_t)ch * R + r] : 0.f;
    }
    y[i] = bf_pack8(f);
}
__global__ void to_fl_kernel(const u32x4* __restrict__ x, float* __restrict__ y, int C, size_t R) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int noct = (C + 7) >> 3;
    if (i >= (size_t)noct * R) return;
    const size_t r = i % R;
    const int oct = (int)(i / R);
    float f[8];
    bf_unpack8(x[i], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = oct * 8 + j;
        if (ch < C) y[(size_t)ch * R + r] = f[j];
    }
}

// (N, C, L) fp32 of the module API <-> ONE octet [L][NP][8] (C <= 8: the network input, its output and the output's cotangent):
// one pass through a 32 x 32 (frame, position) LDS tile per channel instead of a frames-last fp32 intermediate
// (N, C, L) fp32 <-> octets [L][NP][8] bf16 (network input / output / cotangent).  The two layouts are transposes of each other
// (positions contiguous in one, frames in the other): a tile of 64 positions x 16 frames goes through LDS as packed octets,
// so the fp32 side moves 256-byte row pieces (a wave = 64 consecutive positions of one (frame, channel) row) and the octet
// side 256-byte pieces (16 frames x 16 bytes).  (Round 3: the first version moved 128-byte pieces through an fp32 tile of
// 32 x 32 x 8 and ran at 1.1 TB/s.)
__global__ __launch_bounds__(256) void from_ncl_kernel(const float* __restrict__ x, u32x4* __restrict__ y, int N, int C, int L,
                                                       int NP) {
    __shared__ u32x4 t[64][17];
    const int n0 = blockIdx.x * 16, l0 = blockIdx.y * 64;
    const int lx = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int nn = w + 4 * k, n = n0 + nn, l = l0 + lx;
        float f[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) f[cc] = (cc < C && n < N && l < L) ? x[((size_t)n * C + cc) * L + l] : 0.f;
        t[lx][nn] = bf_pack8(f);
    }
    __syncthreads();
    const int nx = threadIdx.x & 15, ly = threadIdx.x >> 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int l = l0 + ly + 16 * k, n = n0 + nx;
        if (l < L && n < NP) y[(size_t)l * NP + n] = t[ly + 16 * k][nx];
    }
}
__global__ __launch_bounds__(256) void to_ncl_kernel(const u32x4* __restrict__ y, float* __restrict__ x, int N, int C, int L,
                                                     int NP) {
    __shared__ u32x4 t[64][17];
    const int n0 = blockIdx.x * 16, l0 = blockIdx.y * 64;
    const int nx = threadIdx.x & 15, ly = threadIdx.x >> 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int l = l0 + ly + 16 * k, n = n0 + nx;
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        t[ly + 16 * k][nx] = (l < L && n < NP) ? y[(size_t)l * NP + n] : z4;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int nn = w + 4 * k, n = n0 + nn, l = l0 + lx;
        float f[8];
        bf_unpack8(t[lx][nn], f);
        if (n < N && l < L) {
#pragma unroll
            for (int cc = 0; cc < 8; ++cc)
                if (cc < C) x[((size_t)n * C + cc) * L + l] = f[cc];
        }
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int trunet_bf16_from_ncl(const float* x, void* y, int N, int C, int L, int NP, void* stream) {
    if (!x || !y || N <= 0 || NP < N || L <= 0) return TRUNET_EINVAL;
    if (C <= 0 || C > 8) return TRUNET_ENOTSUP;
    hipLaunchKernelGGL(from_ncl_kernel, dim3((NP + 15) / 16, (L + 63) / 64), dim3(256), 0, ST, x, (u32x4*)y, N, C, L, NP);
    return trunet_launch_status();
}
extern "C" int trunet_bf16_to_ncl(const void* y, float* x, int N, int C, int L, int NP, void* stream) {
    if (!x || !y || N <= 0 || NP < N || L <= 0) return TRUNET_EINVAL;
    if (C <= 0 || C > 8) return TRUNET_ENOTSUP;
    hipLaunchKernelGGL(to_ncl_kernel, dim3((NP + 15) / 16, (L + 63) / 64), dim3(256), 0, ST, (const u32x4*)y, x, N, C, L, NP);
    return trunet_launch_status();
}

extern "C" int trunet_bf16_gemm_nparts(void) { return BG_GRID * 4; }

static bool bseg_ok(const trunet_bseg& sg) {
    if (!sg.src0 || sg.nchan <= 0 || sg.pos_div <= 0 || sg.L <= 0) return false;
    if (sg.mode == TRUNET_PRO_BNBWD && (!sg.src1 || !sg.c0 || !sg.c1 || !sg.c2)) return false;
    if (sg.mode == TRUNET_PRO_BNRELU && (!sg.c0 || !sg.c1)) return false;
    if (sg.mode != TRUNET_PRO_NONE && sg.mode != TRUNET_PRO_BNRELU && sg.mode != TRUNET_PRO_BNBWD) return false;
    return true;
}

extern "C" int trunet_bf16_gemm(const trunet_bgemm_args* h_in, void* stream) {
    if (!h_in) return TRUNET_EINVAL;
    trunet_bgemm_args hcopy = *h_in;                 // the template dispatch below compares `epi` exactly
    const bool prezero = (hcopy.epi & TRUNET_EPI_PREZERO) != 0;
    hcopy.epi &= ~TRUNET_EPI_PREZERO;
    const trunet_bgemm_args* h = &hcopy;
    if (!h->out || !h->wfrag || h->nseg < 1 || h->nseg > TRUNET_MAX_SEG) return TRUNET_EINVAL;
    if (h->NP <= 0 || (h->NP % 64) != 0 || h->N <= 0 || h->N > h->NP || h->P <= 0 || h->M <= 0) return TRUNET_EINVAL;
    if (h->M > 128 || h->nks_total <= 0 || h->nks_total > BG_MAXKS) return TRUNET_ENOTSUP;
    if (h->out_L <= 0 || h->p_begin < 0 || h->p_begin + h->out_pos_off < 0 || h->p_begin + h->P + h->out_pos_off > h->out_L)
        return TRUNET_EINVAL;
    if ((h->epi & TRUNET_EPI_STATS) && (!h->partials || h->M_stat < h->M)) return TRUNET_EINVAL;
    if ((h->epi & TRUNET_EPI_MASK) && (!h->zmask || !h->e0 || !h->e1)) return TRUNET_EINVAL;
    if ((h->epi & TRUNET_EPI_BIAS) && !h->bias) return TRUNET_EINVAL;
    if ((h->epi & TRUNET_EPI_F32OUT) && ((h->epi & ~(TRUNET_EPI_F32OUT | TRUNET_EPI_BIAS)) || h->m_out_off < 0)) return TRUNET_EINVAL;
    for (int s = 0; s < h->nseg; ++s) {
        const trunet_bseg& sg = h->seg[s];
        if (!bseg_ok(sg) || sg.kstep0 < 0) return TRUNET_EINVAL;
        if (sg.kstep0 + (((sg.nchan + 7) / 8 + 1) / 2) > h->nks_total) return TRUNET_EINVAL;
    }
    if ((h->epi & TRUNET_EPI_STATS) && !prezero) {
        if (hipMemsetAsync(h->partials, 0, (size_t)trunet_bf16_gemm_nparts() * h->M_stat * 2 * sizeof(float), ST) != hipSuccess)
            return TRUNET_ELAUNCH;
    }
    const int nrt_all = (h->M + 31) / 32;
    const size_t lds = (size_t)nrt_all * h->nks_total * 64 * 16 + (size_t)h->nks_total * 2 * 3 * 8 * sizeof(float) + 4 * 128 * sizeof(float);
    // prologue class of the launch: BatchNorm-backward pairs, BN+ReLU (plain segments in it must be post-ReLU sources:
    // they get identity coefficients), or plain
    int pro = TRUNET_PRO_NONE;
    bool full = (h->M % 32) == 0, mixed = false;
    for (int s = 0; s < h->nseg; ++s) {
        const int m = h->seg[s].mode;
        if (m == TRUNET_PRO_BNBWD) pro = TRUNET_PRO_BNBWD;
        else if (m == TRUNET_PRO_BNRELU && pro != TRUNET_PRO_BNBWD) pro = TRUNET_PRO_BNRELU;
        full = full && (h->seg[s].nchan % 16) == 0;
    }
    for (int s = 0; s < h->nseg; ++s) mixed = mixed || (pro == TRUNET_PRO_BNBWD && h->seg[s].mode != TRUNET_PRO_BNBWD);
    if (mixed) return TRUNET_ENOTSUP;
#define BG_LAUNCH(NRT_, PRO_, EPI_, FULL_)                                                                                          \
    do {                                                                                                                            \
        if (hipFuncSetAttribute((const void*)bgemm_kernel<NRT_, PRO_, EPI_, FULL_>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                (int)lds) != hipSuccess)                                                                            \
            return TRUNET_ELAUNCH;                                                                                                  \
        hipLaunchKernelGGL((bgemm_kernel<NRT_, PRO_, EPI_, FULL_>), dim3(BG_GRID), dim3(256), lds, ST, *h);                         \
        return trunet_launch_status();                                                                                              \
    } while (0)
    constexpr int B = TRUNET_EPI_BIAS, S = TRUNET_EPI_STATS, A = TRUNET_EPI_ACCUM, K = TRUNET_EPI_MASK;
    if (full && nrt_all >= 2) {             // the hot launches of the training step
        // 128 rows forward: ONE wave per tile with all four row tiles (the split form runs the prologue twice): -0.15 ms
        if (pro == TRUNET_PRO_BNRELU && h->epi == (B | S) && nrt_all == 4) BG_LAUNCH(4, TRUNET_PRO_BNRELU, B | S, true);
        if (pro == TRUNET_PRO_NONE && h->epi == (B | S) && nrt_all == 4) BG_LAUNCH(4, TRUNET_PRO_NONE, B | S, true);
        if (pro == TRUNET_PRO_BNRELU && h->epi == (B | S)) BG_LAUNCH(2, TRUNET_PRO_BNRELU, B | S, true);
        if (pro == TRUNET_PRO_NONE && h->epi == (B | S)) BG_LAUNCH(2, TRUNET_PRO_NONE, B | S, true);
        if (pro == TRUNET_PRO_BNRELU && h->epi == (B | TRUNET_EPI_F32OUT) && nrt_all == 4)
            BG_LAUNCH(4, TRUNET_PRO_BNRELU, B | TRUNET_EPI_F32OUT, true);
        if (pro == TRUNET_PRO_NONE && h->epi == (K | S)) BG_LAUNCH(2, TRUNET_PRO_NONE, K | S, true);
        if (pro == TRUNET_PRO_BNBWD && h->epi == (K | S)) BG_LAUNCH(2, TRUNET_PRO_BNBWD, K | S, true);
        if (pro == TRUNET_PRO_BNBWD && h->epi == (K | S | A)) BG_LAUNCH(2, TRUNET_PRO_BNBWD, K | S | A, true);
        if (pro == TRUNET_PRO_BNBWD && h->epi == (K | A)) BG_LAUNCH(2, TRUNET_PRO_BNBWD, K | A, true);
        if (pro == TRUNET_PRO_BNBWD && h->epi == 0) BG_LAUNCH(2, TRUNET_PRO_BNBWD, 0, true);
    }
    // the thin layers of the training step (first conv 4 -> 64, decoder.5: 128 -> 8 and 8 -> 8): channel guards stay, the
    // prologue / epilogue are compile-time like above
    constexpr int R = TRUNET_EPI_RELU;
    if (!full) {
        if (nrt_all == 1) {
            if (pro == TRUNET_PRO_BNRELU && h->epi == (B | S)) BG_LAUNCH(1, TRUNET_PRO_BNRELU, B | S, false);
            if (pro == TRUNET_PRO_BNRELU && h->epi == B) BG_LAUNCH(1, TRUNET_PRO_BNRELU, B, false);
            if (pro == TRUNET_PRO_NONE && h->epi == (K | S)) BG_LAUNCH(1, TRUNET_PRO_NONE, K | S, false);
        } else {
            if (pro == TRUNET_PRO_NONE && h->epi == (B | R)) BG_LAUNCH(2, TRUNET_PRO_NONE, B | R, false);
            if (pro == TRUNET_PRO_BNBWD && h->epi == (K | S)) BG_LAUNCH(2, TRUNET_PRO_BNBWD, K | S, false);
            if (pro == TRUNET_PRO_BNBWD && h->epi == 0) BG_LAUNCH(2, TRUNET_PRO_BNBWD, 0, false);
        }
    }
    // everything else (eval-mode forwards, other shapes): flags and modes tested at run time
    if (nrt_all == 1) BG_LAUNCH(1, -1, -1, false);
    BG_LAUNCH(2, -1, -1, false);
#undef BG_LAUNCH
}

extern "C" int trunet_bf16_pack_weight(const float* W, void* wfrag, int M, int ldw_m, int ldw_c, int w_m_off, int nseg,
                                       const int32_t* seg_nchan, const int32_t* seg_woff, void* stream) {
    if (!W || !wfrag || M <= 0 || M > 128 || nseg < 1 || nseg > TRUNET_MAX_SEG || !seg_nchan || !seg_woff) return TRUNET_EINVAL;
    PackDesc d;
    int ks = 0;
    for (int s = 0; s < TRUNET_MAX_SEG; ++s) {
        d.nchan[s] = s < nseg ? seg_nchan[s] : 0;
        d.woff[s] = s < nseg ? seg_woff[s] : 0;
        d.ks0[s] = ks;
        if (s < nseg) {
            if (seg_nchan[s] <= 0) return TRUNET_EINVAL;
            ks += ((seg_nchan[s] + 7) / 8 + 1) / 2;
        }
    }
    if (ks > BG_MAXKS) return TRUNET_ENOTSUP;
    d.nseg = nseg;
    d.nks_total = ks;
    const int total = ((M + 31) / 32) * ks * 64;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, ST, W, (u32x4*)wfrag, M, ldw_m, ldw_c,
                       w_m_off, d);
    const int rc = trunet_launch_status();
    return rc == TRUNET_OK ? ks : rc;
}

extern "C" int trunet_bf16_pack_weights_batch(const trunet_bpack_desc* d_descs, int n, int max_elems, void* stream) {
    if (!d_descs || n <= 0 || max_elems <= 0) return TRUNET_EINVAL;
    hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((max_elems + 255) / 256, n), dim3(256), 0, ST, d_descs);
    return trunet_launch_status();
}

extern "C" int trunet_bf16_wgrad(const trunet_bwgrad_args* h, void* stream) {
    if (!h || !h->a0 || !h->w_partials || h->nseg < 1 || h->nseg > TRUNET_MAX_SEG) return TRUNET_EINVAL;
    if (h->NP <= 0 || (h->NP % BW_F) != 0 || h->N <= 0 || h->N > h->NP || h->P <= 0 || h->M <= 0 || h->w_numel <= 0)
        return TRUNET_EINVAL;
    if (h->a_mode == TRUNET_PRO_BNBWD) { if (!h->a1 || !h->ac0 || !h->ac1 || !h->ac2) return TRUNET_EINVAL; }
    else if (h->a_mode != TRUNET_PRO_NONE) return TRUNET_EINVAL;
    if (h->a_L <= 0 || h->p_begin < 0 || h->p_begin + h->a_pos_off < 0 || h->p_begin + h->P + h->a_pos_off > h->a_L)
        return TRUNET_EINVAL;
    if (h->M > 128) return TRUNET_ENOTSUP;
    int soct = 0, nct = 0;
    for (int s = 0; s < h->nseg; ++s) {
        const trunet_bseg& sg = h->seg[s];
        if (!bseg_ok(sg) || sg.mode == TRUNET_PRO_BNBWD) return TRUNET_EINVAL;
        if (sg.pos_div != 1 && sg.pos_div != 2) return TRUNET_ENOTSUP;          // the kernel's positions are shifts
        // a 32-channel tile reads 4 octets: round every segment up to whole tiles in the LDS image
        soct += ((sg.nchan + 31) / 32) * 4;
        nct += (sg.nchan + 31) / 32;
    }
    if (soct > 8 * BW_MAXS || ((h->M + 31) / 32) * nct > 8 * BW_MAXT) return TRUNET_ENOTSUP;
    const int moct = ((h->M + 31) / 32) * 4;
    const size_t lds = 2 * (size_t)(moct + soct) * BW_OS + (size_t)(moct * 24 + soct * 16) * sizeof(float);
    bool bn = false;
    for (int s = 0; s < h->nseg; ++s) bn = bn || h->seg[s].mode == TRUNET_PRO_BNRELU;
    const bool two = h->a_mode == TRUNET_PRO_BNBWD;
    trunet_bdgrad_args nodg = {};
#define BW_LAUNCH(TWO_, SM_)                                                                                                       \
    do {                                                                                                                           \
        if (hipFuncSetAttribute((const void*)bwgrad_kernel<TWO_, SM_, false>, hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                (int)lds) != hipSuccess)                                                                           \
            return TRUNET_ELAUNCH;                                                                                                 \
        hipLaunchKernelGGL((bwgrad_kernel<TWO_, SM_, false>), dim3(trunet_conv_wgrad_nparts()), dim3(BW_THREADS), lds, ST, *h,     \
                           soct, nodg);                                                                                            \
    } while (0)
    if (two) { if (bn) BW_LAUNCH(true, TRUNET_PRO_BNRELU); else BW_LAUNCH(true, TRUNET_PRO_NONE); }
    else { if (bn) BW_LAUNCH(false, TRUNET_PRO_BNRELU); else BW_LAUNCH(false, TRUNET_PRO_NONE); }
#undef BW_LAUNCH
    return trunet_launch_status();
}

extern "C" int trunet_bf16_pw_bwd_nparts(void) { return 2 * trunet_conv_wgrad_nparts(); }

extern "C" int trunet_bf16_pw_bwd(const trunet_bpwbwd_args* H, void* stream) {
    if (!H) return TRUNET_EINVAL;
    const trunet_bwgrad_args* h = &H->w;
    const trunet_bdgrad_args* d = &H->dg;
    if (!h->a0 || !h->w_partials || h->nseg < 1 || h->nseg > TRUNET_MAX_SEG || !d->wfragT) return TRUNET_EINVAL;
    if (h->NP <= 0 || (h->NP % BW_F) != 0 || h->N <= 0 || h->N > h->NP || h->P <= 0 || h->M <= 0 || h->w_numel <= 0)
        return TRUNET_EINVAL;
    if (h->a_mode != TRUNET_PRO_BNBWD || !h->a1 || !h->ac0 || !h->ac1 || !h->ac2) return TRUNET_ENOTSUP;
    if (h->a_L <= 0 || h->p_begin != 0 || h->a_pos_off != 0 || h->P > h->a_L) return TRUNET_EINVAL;
    if (h->M > 128 || (h->M % 16) != 0) return TRUNET_ENOTSUP;
    int soct = 0, nct = 0, nrt_total = 0;
    bool bn = false;
    for (int s = 0; s < h->nseg; ++s) {
        const trunet_bseg& sg = h->seg[s];
        if (!bseg_ok(sg) || sg.mode == TRUNET_PRO_BNBWD || !d->out[s]) return TRUNET_EINVAL;
        if (sg.pos_mul != 1 || sg.pos_div != 1 || (sg.nchan % 32) != 0) return TRUNET_ENOTSUP;       // pointwise layers
        if ((d->flags[s] & 4) && (!d->partials[s] || !d->mean[s] || !(d->flags[s] & 2))) return TRUNET_EINVAL;
        bn = bn || sg.mode == TRUNET_PRO_BNRELU;
        soct += sg.nchan / 8;
        nct += sg.nchan / 32;
        nrt_total += sg.nchan / 32;
    }
    if (nrt_total != d->nrt_total) return TRUNET_EINVAL;
    if (soct > 8 * 3 || ((h->M + 31) / 32) * nct > 8 * 2 || 2 * nrt_total > 8 * 2) return TRUNET_ENOTSUP;
    const int moct = ((h->M + 31) / 32) * 4, nks = (h->M / 8 + 1) / 2;
    const size_t lds = 2 * (size_t)(moct + 2 * soct) * BW_OS + (size_t)(moct * 24 + soct * 24) * sizeof(float) +
                       (size_t)nrt_total * nks * 64 * 16;
    if (lds > 160 * 1024) return TRUNET_ENOTSUP;
    for (int s = 0; s < h->nseg; ++s)
        if ((d->flags[s] & 4) && !(d->flags[s] & TRUNET_DG_PREZERO))
            if (hipMemsetAsync(d->partials[s], 0, (size_t)trunet_bf16_pw_bwd_nparts() * h->seg[s].nchan * 2 * sizeof(float), ST) !=
                hipSuccess)
                return TRUNET_ELAUNCH;
#define BWD_LAUNCH(SM_)                                                                                                            \
    do {                                                                                                                           \
        if (hipFuncSetAttribute((const void*)bwgrad_kernel<true, SM_, true>, hipFuncAttributeMaxDynamicSharedMemorySize,          \
                                (int)lds) != hipSuccess)                                                                           \
            return TRUNET_ELAUNCH;                                                                                                 \
        hipLaunchKernelGGL((bwgrad_kernel<true, SM_, true>), dim3(trunet_conv_wgrad_nparts()), dim3(BW_THREADS), lds, ST, *h,      \
                           soct, *d);                                                                                              \
    } while (0)
    if (bn) BWD_LAUNCH(TRUNET_PRO_BNRELU); else BWD_LAUNCH(TRUNET_PRO_NONE);
#undef BWD_LAUNCH
    return trunet_launch_status();
}

extern "C" int trunet_bf16_dw_nparts(int NP, int rows) { return (NP / 256) * bdw_chunks(rows); }

extern "C" int trunet_bf16_dwconv_fwd(const void* zin, const float* s_in, const float* t_in, const float* w, const float* b,
                                      void* zout, float* partials, int C, int K, int S, int Lin, int Lout, int NP, int N,
                                      void* stream) {
    if (!zin || !s_in || !t_in || !w || !b || !zout || !partials) return TRUNET_EINVAL;
    if (C <= 0 || (C % 8) || NP <= 0 || (NP % 256) || N <= 0 || N > NP || Lin <= 0 || Lout != (Lin + 2 * (K / 2) - K) / S + 1)
        return TRUNET_EINVAL;
    const dim3 grid(NP / 256, C / 8, bdw_chunks(Lout));
#define BDW_FWD(KK, SS) hipLaunchKernelGGL((bdw_fwd_kernel<KK, SS>), grid, dim3(256), 0, ST, (const u32x4*)zin, s_in, t_in, w, b, \
                                           (u32x4*)zout, partials, C, Lin, Lout, NP, N)
    if (K == 3 && S == 1) BDW_FWD(3, 1);
    else if (K == 5 && S == 2) BDW_FWD(5, 2);
    else if (K == 3 && S == 2) BDW_FWD(3, 2);
    else return TRUNET_ENOTSUP;
#undef BDW_FWD
    return trunet_launch_status();
}

extern "C" int trunet_bf16_dwconv_bwd(const void* dy, const void* z, const float* ca, const float* cb, const float* cc,
                                      const void* zin, const float* s_in, const float* t_in, const float* mean_in,
                                      const float* w, void* dy_in, float* partials_in, float* w_partials, float* b_partials,
                                      int C, int K, int S, int Lin, int Lout, int NP, int N, void* stream) {
    if (!dy || !z || !ca || !cb || !cc || !zin || !s_in || !t_in || !mean_in || !w || !dy_in || !partials_in || !w_partials ||
        !b_partials)
        return TRUNET_EINVAL;
    if (C <= 0 || (C % 8) || NP <= 0 || (NP % 256) || N <= 0 || N > NP || Lin <= 0 || Lout != (Lin + 2 * (K / 2) - K) / S + 1)
        return TRUNET_EINVAL;
    const dim3 grid(NP / 256, C / 8, bdw_chunks(Lin));
#define BDW_BWD(KK, SS) hipLaunchKernelGGL((bdw_bwd_kernel<KK, SS>), grid, dim3(256), 0, ST, (const u32x4*)dy, (const u32x4*)z, ca, \
                                           cb, cc, (const u32x4*)zin, s_in, t_in, mean_in, w, (u32x4*)dy_in, partials_in,           \
                                           w_partials, b_partials, C, Lin, Lout, NP, N)
    if (K == 3 && S == 1) BDW_BWD(3, 1);
    else if (K == 5 && S == 2) BDW_BWD(5, 2);
    else if (K == 3 && S == 2) BDW_BWD(3, 2);
    else return TRUNET_ENOTSUP;
#undef BDW_BWD
    return trunet_launch_status();
}

extern "C" int trunet_bf16_from_frames_last(const float* x, void* y, int C, int L, int NP, void* stream) {
    if (!x || !y || C <= 0 || L <= 0 || NP <= 0) return TRUNET_EINVAL;
    const size_t R = (size_t)L * NP, total = (size_t)((C + 7) / 8) * R;
    hipLaunchKernelGGL(from_fl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST, x, (u32x4*)y, C, R);
    return trunet_launch_status();
}
extern "C" int trunet_bf16_to_frames_last(const void* x, float* y, int C, int L, int NP, void* stream) {
    if (!x || !y || C <= 0 || L <= 0 || NP <= 0) return TRUNET_EINVAL;
    const size_t R = (size_t)L * NP, total = (size_t)((C + 7) / 8) * R;
    hipLaunchKernelGGL(to_fl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ST, (const u32x4*)x, y, C, R);
    return trunet_launch_status();
}
