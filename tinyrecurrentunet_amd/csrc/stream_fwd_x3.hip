// stream_fwd.hip with the pointwise convolutions of the encoder (encoder.1 .. encoder.5: 128-row layers on 32-row tiles) on the
// bf16 MFMA through the exact three-term split of the fp32 operands (x3_common.hpp; round 4): the folded weights arrive as three
// bf16 fragment planes (export.x3_image), the consuming wave splits the 8 K-values of its B fragment between the six MFMAs of
// the previous K-step (order pinned: one wave per SIMD has to overlap its own vector and matrix work).  Everything else is
// stream_fwd.hip verbatim -- read that file's header; this one exists next to it so that the fp32-MFMA kernel stays the
// reference the split one is A/B-tested against.
//
// Eval-mode TRU-Net forward as ONE launch (BatchNorm-folded, single-launch persistent forward for the
// streaming protocol of rt.py:20-27,76-84; network.py:122-171 with repairs R1-R4).
//
// In eval mode BatchNorm is a per-channel affine map (running statistics), so it folds into the conv in front of it
// and every frame becomes independent of every other: no grid-wide reduction, no inter-workgroup traffic at all.
// A workgroup (4 waves, one per SIMD, up to 512 registers each) therefore takes ONE frame through all 24 layers with
// every activation of that frame in its own LDS (<= 144 KiB live), then the next frame (grid-stride).  Nothing but
// the frame's features in, its 8 x 257 output and the five skip tensors (176 KB per workgroup, L2 / Infinity-Cache
// resident, written and read back by the same workgroup) ever leaves the CU.
//
// Matrix layers (pointwise convs, transposed convs as per-tap GEMMs over the parity classes of the output positions,
// the GRU input projection) run on v_mfma_f32_32x32x2_f32 with the WEIGHTS AS REGISTER-RESIDENT A FRAGMENTS -- the host
// exporter (export.py) folds BatchNorm and stores every 32-row weight tile in fragment order, so a wave's load is
// KP/4 + 4 fully coalesced 16-byte loads -- and the activations as the B operand straight from LDS (one ds_read_b32
// per MFMA, conflict-free: 32 consecutive positions of one channel row).  A layer's fragments are requested while the
// previous layer computes (two register sets).  Depthwise convs, the first (C_in -> 64) conv, the 16-step GRU
// recurrence (matrix-vector per frame) and the 8-channel tail run on the vector ALU.
//
// LDS rows are [4 zero guard floats][L positions][>= 4 zero floats]: taps, F.pad (network.py:96-97) and conv padding
// read zeros instead of branching; crops are a column offset.
#include "common.hpp"
#include "x3_common.hpp"

namespace {

constexpr int SF_T = 256;
constexpr int SF_R0 = 0, SF_R1A = 18432, SF_R1B = 18432 + 9216, SF_R2 = 36864, SF_ARENA = 40960;   // floats (160 KiB)
// small fixed buffers behind the three activation regions: features of the frame, first-conv weights (resident for the
// whole kernel), depthwise / last-layer weights of the current block, GRU state
constexpr int SF_XB = SF_R2, SF_W0 = SF_R2 + 1088, SF_DWB = SF_R2 + 2432, SF_GRU = SF_R2 + 3200;
constexpr int SF_SKIP = 8192 + 16384 + 8192 + 8192 + 4096;       // enc0..enc4 per workgroup (floats)

__host__ __device__ constexpr int sf_ls(int L) { return (L + 8 + 15) / 16 * 16; }

#ifndef GRU_LIBM
__device__ __forceinline__ float sf_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float sf_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
#else
__device__ __forceinline__ float sf_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float sf_tanh(float x) { return tanhf(x); }
#endif

__device__ __forceinline__ float sf_dpp_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// The per-lane addresses of a layer's epilogue depend only on the thread index, so the compiler would compute them for
// all layers once, before the frame loop, and keep (spill) hundreds of them: every layer re-derives them from an opaque
// copy of the thread index instead.
__device__ __forceinline__ int sf_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// one row tile's fragments: NQ quads per lane, [quad][lane][4] in the blob (A fragments, then 16 bias values in C layout)
template <int NQ>
__device__ __forceinline__ void sf_load(float* af, const float* tile, int lane) {
    const f32x4* p = (const f32x4*)tile + lane;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
#if defined(SF_ABL) && (SF_ABL & 8)      // diagnostic: no fragment loads at all (timing only)
        const f32x4 t = {1.f + i, 2.f, 3.f, (float)lane};
        (void)p;
#else
        const f32x4 t = p[i * 64];
#endif
        af[4 * i + 0] = t[0]; af[4 * i + 1] = t[1]; af[4 * i + 2] = t[2]; af[4 * i + 3] = t[3];
    }
}

// The B-operand reads of an MFMA stream must be bare `ds_read_b32 v, base offset:imm`: ONE vector-ALU instruction between
// two MFMAs of a wave that is alone on its SIMD costs 12 cycles of matrix pipe, an address add feeding the read 24
// (scripts/dbg/mfma_mix.hip: 64 -> 76 -> 88 cycles per 32x32x2 MFMA, 33 -> 45 -> 56 per 16x16x4; the read alone: +2).
// hipcc folds a constant into the 16-bit offset field only if the whole constant fits, and the buffer offsets inside the
// 160 KiB arena do not: the lane's base is made opaque (in the LDS address space) and the stream's own constants fit.
typedef __attribute__((address_space(3))) const float* sf_lptr;
__device__ __forceinline__ sf_lptr sf_lds_base(const float* p) {
    sf_lptr q = (sf_lptr)p;
    asm volatile("" : "+v"(q));
    return q;
}

// acc += A(af[0..KP)) * B, B[k = 2 kk + h][j = c] = S[kk * rs2] (S is the lane's base: row h, column of this lane).
// One wave per SIMD must hide the LDS latency of its own B operand: the read for k-pair kk + LA is issued right before
// the MFMA of k-pair kk, and the order is pinned (hipcc otherwise sinks every ds_read directly in front of the MFMA that
// consumes it and the matrix pipe idles ~75 cycles per pair of MFMAs: measured 63 % -> see DESIGN.md).
// (NQN > 0 with pf set requests the NQN quads of the NEXT layer's fragments in between the MFMAs instead of as one block
// in front of the layer, which holds each wave for ~1.2k cycles at 64 B/clk/CU of L1 fill.  Measured: no gain -- the wave
// stalls on the full memory queue wherever the loads sit, and the extra live ranges spilled; the call sites use NQN = 0.
// Re-measured after the 16-row-tile rework (no spills any more): the request phase shrinks from 1.17k to 0.16k cycles and
// the MFMA phase grows by 0.93k -- a vector-memory instruction between two MFMAs costs ~45 cycles of matrix pipe.)
template <int KP, int NQN = 0>
__device__ __forceinline__ void sf_mm(f32x16& acc, const float* af, const float* S, int rs2, bool pf = false,
                                      float* fpn = nullptr, const f32x4* pn = nullptr) {
    constexpr int LA = 6;
    float b[KP];
    const sf_lptr S0 = sf_lds_base(S), S1 = sf_lds_base(S + (KP > 32 ? 32 : 0) * rs2);      // 32 k-pairs x 1152 B per base
#define SF_B(kk) ((kk) < 32 ? S0[(kk) * rs2] : S1[((kk) - 32) * rs2])
#if defined(SF_ABL) && (SF_ABL & 1)      // diagnostic: no LDS reads for the B operand (wrong results, timing only)
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) b[kk] = af[kk] + (float)rs2;
#define SF_ABL_NOB 1
#else
#define SF_ABL_NOB 0
#endif
#pragma unroll
    for (int kk = 0; kk < LA && kk < KP && !SF_ABL_NOB; ++kk) b[kk] = SF_B(kk);
#if defined(SF_ABL) && (SF_ABL & 2)      // diagnostic: every unrolled MFMA block runs twice (second pass from the I-cache)
    for (int rep_ = 0; rep_ < 2; ++rep_) {
        asm volatile("" ::: "memory");
#endif
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
        if (kk + LA < KP && !SF_ABL_NOB) b[kk + LA] = SF_B(kk + LA);
        if constexpr (NQN > 0) {
            if (pf) {
#pragma unroll
                for (int i = kk * NQN / KP; i < (kk + 1) * NQN / KP; ++i) {
                    const f32x4 t = pn[i * 64];
                    fpn[4 * i + 0] = t[0]; fpn[4 * i + 1] = t[1]; fpn[4 * i + 2] = t[2]; fpn[4 * i + 3] = t[3];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], b[kk], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#if defined(SF_ABL) && (SF_ABL & 2)
    }
#endif
#undef SF_B
}

// zero the guard columns [-4, 0) and [L, L + 4) of a [rows][ls] buffer
__device__ __forceinline__ void sf_guards(float* lds, int buf, int rows, int ls, int L) {
    for (int i = sf_tid(); i < rows * 8; i += SF_T) {
        const int r = i >> 3, g = i & 7;
        lds[buf + r * ls + (g < 4 ? g : L + g)] = 0.f;
    }
}

// Pointwise conv (+ folded BatchNorm) over one or two sources: dst[m][p] = act(bias[m] + sum_k W[m][k] src[k][p + coff]).
// NRT row tiles of 32: 4 -> one per wave; 2 -> wave (row tile, column-tile parity); 1 -> waves split the column tiles.
template <int KP1, int KP2, int NRT, int NQN = 0>
__device__ __forceinline__ void sf_pw(const float* af, float* lds, int src1, int ls1, int coff1, int src2, int ls2, int dst,
                                      int lsd, int P, int M, int row0, bool relu, float* fpn = nullptr,
                                      const float* tile_next = nullptr) {
    const int tid_ = sf_tid();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), h = lane >> 5, c = lane & 31;
    const int rt = NRT == 4 ? wave : (NRT == 2 ? (wave & 1) : 0);
    const int ct0 = NRT == 4 ? 0 : (NRT == 2 ? (wave >> 1) : wave);
    const int cts = NRT == 4 ? 1 : (NRT == 2 ? 2 : 4);
    const int nct = (P + 31) >> 5;
    auto store = [&](const f32x16& acc, int ct) __attribute__((always_inline)) {
        const int col = ct * 32 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[r] + af[KP1 + KP2 + r];
            if (relu) v = fmaxf(v, 0.f);
            if (col < P && row < M) lds[dst + (row0 + row) * lsd + 4 + col] = v;
        }
    };
    int ct = ct0;
    bool pf = NQN > 0;                       // the next layer's fragments ride along with this wave's first tile
    const f32x4* pn = (const f32x4*)tile_next + lane;
    for (; ct < nct; ct += cts) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        sf_mm<KP1, NQN>(acc, af, lds + src1 + h * ls1 + 4 + ct * 32 + c + coff1, 2 * ls1, pf, fpn, pn);
        pf = false;
        if constexpr (KP2 > 0) sf_mm<KP2>(acc, af + KP1, lds + src2 + h * ls2 + 4 + ct * 32 + c, 2 * ls2);
        store(acc, ct);
    }
    if constexpr (NQN > 0) {
        if (pf) sf_load<NQN>(fpn, tile_next, lane);      // a wave without a tile in this layer still needs its fragments
    }
}

// ---- 32-row tiles on the bf16 MFMA: acc += A * B over KS K-steps of 16.  A: three bf16 planes per K-step in af (12 words per
// K-step: plane 0, 1, 2 x 4 words), B[k][col] = S[k * ls]: S is the lane's base (row 8 h, this lane's column), K-step ks uses
// k = 16 ks + 8 h + j (j < 8): 8 ds_read_b32, issued one K-step ahead, split between the MFMAs of the K-step before.
#define SF_SB() __builtin_amdgcn_sched_barrier(0)
template <int KS>
__device__ __forceinline__ void sf_mm_x3(f32x16& acc, const float* af, const float* S, int ls) {
    const sf_lptr S0 = sf_lds_base(S), S1 = sf_lds_base(S + (KS > 4 ? 64 : 0) * ls);      // 64 rows x 576 B per base
#define SF_B(k) ((k) < 64 ? S0[(k) * ls] : S1[((k) - 64) * ls])
    auto plane = [&](int ks, int pl) __attribute__((always_inline)) {
        const float* p = af + (ks * 3 + pl) * 4;
        return u32x4{__builtin_bit_cast(unsigned, p[0]), __builtin_bit_cast(unsigned, p[1]), __builtin_bit_cast(unsigned, p[2]),
                     __builtin_bit_cast(unsigned, p[3])};
    };
    float x[8];
    u32x4 b0, b1, b2;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = SF_B(j);
    {
        unsigned u0, u1, u2;
#pragma unroll
        for (int w = 0; w < 4; ++w) { ctx_split2(x[2 * w], x[2 * w + 1], u0, u1, u2); b0[w] = u0; b1[w] = u1; b2[w] = u2; }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const u32x4 a0 = plane(ks, 0), a1 = plane(ks, 1), a2 = plane(ks, 2);
        const bool nx = ks + 1 < KS;
        u32x4 n0 = b0, n1 = b1, n2 = b2;
        unsigned u0, u1, u2;
        if (nx) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = SF_B(16 * (ks + 1) + j);
        }
        SF_SB(); CTX_MF(acc, a2, b0); SF_SB();
        SF_SB(); CTX_MF(acc, a0, b2); SF_SB();
        if (nx) { ctx_split2(x[0], x[1], u0, u1, u2); n0[0] = u0; n1[0] = u1; n2[0] = u2; }
        SF_SB(); CTX_MF(acc, a1, b1); SF_SB();
        if (nx) { ctx_split2(x[2], x[3], u0, u1, u2); n0[1] = u0; n1[1] = u1; n2[1] = u2; }
        SF_SB(); CTX_MF(acc, a1, b0); SF_SB();
        if (nx) { ctx_split2(x[4], x[5], u0, u1, u2); n0[2] = u0; n1[2] = u1; n2[2] = u2; }
        SF_SB(); CTX_MF(acc, a0, b1); SF_SB();
        if (nx) { ctx_split2(x[6], x[7], u0, u1, u2); n0[3] = u0; n1[3] = u1; n2[3] = u2; }
        SF_SB(); CTX_MF(acc, a0, b0); SF_SB();
        b0 = n0; b1 = n1; b2 = n2;
    }
#undef SF_B
}

// Pointwise conv (+ folded BatchNorm + ReLU), 128 output rows: one 32-row tile per wave, every wave covers all P columns.
// af: 3 KS quads of planes, then the 16 bias values of the lane's accumulator rows.
template <int KS>
__device__ __forceinline__ void sf_pw_x3(const float* af, float* lds, int src, int ls, int dst, int lsd, int P) {
    const int tid_ = sf_tid();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), h = lane >> 5, c = lane & 31;
    const int nct = (P + 31) >> 5;
    for (int ct = 0; ct < nct; ++ct) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        sf_mm_x3<KS>(acc, af, lds + src + 8 * h * ls + 4 + ct * 32 + c, ls);
        const int col = ct * 32 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v = fmaxf(acc[r] + af[12 * KS + r], 0.f);
            if (col < P) lds[dst + row * lsd + 4 + col] = v;
        }
    }
}

// ---- 16-row tiles (v_mfma_f32_16x16x4_f32) for the 64-channel layers (FGRU.conv, every decoder layer): with 32-row tiles
// two of the four waves hold the SAME row tile's fragments (every weight crosses the L1 twice) and layers with 16 or 32
// positions leave waves or half of every tile idle.  Four 16-row tiles give each wave its own quarter of the weights
// (half the fragment registers and half the fragment traffic), every wave works on every column, and a 16-position
// layer is one exact tile.  Lane l holds A[row l & 15][k = 4 kq + (l >> 4)] and B[k = 4 kq + (l >> 4)][column l & 15];
// accumulator register r of lane l is C[row 4 (l >> 4) + r][column l & 15].
// acc[ct] += A(af[0..KQ)) * B for NCT column tiles of 16: S is the lane's base (row l >> 4, column l & 15), rs4 = 4 rows.
template <int KQ, int NCT>
__device__ __forceinline__ void sf_mm16(f32x4 (&acc)[NCT], const float* af, const float* S, int rs4) {
    constexpr int LA = 8, NT = KQ * NCT;
    float b[NT];
    const sf_lptr S0 = sf_lds_base(S), S1 = sf_lds_base(S + (KQ > 24 ? 24 : 0) * rs4);      // 24 k-quads x 2304 B per base
#define SF_B(x) ((x) / NCT < 24 ? S0[((x) / NCT) * rs4 + ((x) % NCT) * 16] : S1[((x) / NCT - 24) * rs4 + ((x) % NCT) * 16])
#pragma unroll
    for (int x = 0; x < LA && x < NT; ++x) b[x] = SF_B(x);
#pragma unroll
    for (int x = 0; x < NT; ++x) {
        if (x + LA < NT) b[x + LA] = SF_B(x + LA);
        __builtin_amdgcn_sched_barrier(0);
        acc[x % NCT] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[x / NCT], b[x], acc[x % NCT], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef SF_B
}

// Pointwise conv (+ folded BatchNorm) over one or two sources, M <= 64 rows as 16-row tiles.  SPLIT = false: wave = row
// tile, every wave covers all P columns; SPLIT = true (M <= 16): one row tile, the waves split the column groups.
// NCT column tiles of 16 per group (1 for the 16-position layers, else 2).  FULL: P is a multiple of the group width and
// M of 16 (no bounds checks in the epilogue).  rowbase >= 0: first output row of this wave's tile (default 16 x wave).
template <int KQ1, int KQ2, int NCT, bool SPLIT, bool FULL, bool RELU = true>
__device__ __forceinline__ void sf_pw16(const float* af, float* lds, int src1, int ls1, int coff1, int src2, int ls2, int dst,
                                        int lsd, int P, int M, long long* stamps = nullptr, int rowbase = -1) {
    const int tid_ = sf_tid();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), q = lane >> 4, j = lane & 15;
    const int row = (rowbase >= 0 ? rowbase : (SPLIT ? 0 : 16 * wave)) + 4 * q;
    int sidx = 0;
#ifdef SF_STAMPS
#define SF_ISTAMP() do { if (stamps && tid_ == 0) stamps[sidx++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define SF_ISTAMP() do { (void)sidx; (void)stamps; } while (0)
#endif
    SF_ISTAMP();
    const int ng = (P + 16 * NCT - 1) / (16 * NCT);
    for (int g = SPLIT ? wave : 0; g < ng; g += SPLIT ? 4 : 1) {
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c0 = g * 16 * NCT + j;
        SF_ISTAMP();
        sf_mm16<KQ1, NCT>(acc, af, lds + src1 + q * ls1 + 4 + c0 + coff1, 4 * ls1);
        SF_ISTAMP();
        if constexpr (KQ2 > 0) sf_mm16<KQ2, NCT>(acc, af + KQ1, lds + src2 + q * ls2 + 4 + c0, 4 * ls2);
        SF_ISTAMP();
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int col = c0 + 16 * ct;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[ct][r] + af[KQ1 + KQ2 + r];
                if (RELU) v = fmaxf(v, 0.f);
                if (FULL || (col < P && row + r < M)) lds[dst + (row + r) * lsd + 4 + col] = v;
            }
        }
        SF_ISTAMP();
    }
}

// ConvTranspose1d(64 -> 64, k = TAPS, stride S_, padding S_/2) + folded BatchNorm + ReLU on 16-row tiles: wave = row tile.
// Output position p = S_ j + e: class e uses the taps with (e + pad - tap) % S_ == 0 at source column
// j + (e + pad - tap) / S_ -- a dense GEMM per tap.  Only the positions [p0, p0 + Ln) that the next layer reads are
// produced (the crops of network.py:96-97 drop the rest: with them every class is a whole number of column groups).
template <int TAPS, int S_, int NCT, bool FULL>
__device__ __forceinline__ void sf_convT16(const float* af, float* lds, int src, int lsi, int dst, int lsd, int Lout,
                                           int p0, int Ln) {
    const int tid_ = sf_tid();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), q = lane >> 4, j = lane & 15;
    constexpr int PAD = S_ / 2;
    const int pend = min(Lout, p0 + Ln);
    const int row = 16 * wave + 4 * q;
#pragma unroll
    for (int e = 0; e < S_; ++e) {
        const int j0 = p0 > e ? (p0 - e + S_ - 1) / S_ : 0;
        const int nj = (pend - e + S_ - 1) / S_ - j0;
        const int ng = (nj + 16 * NCT - 1) / (16 * NCT);
        for (int g = 0; g < ng; ++g) {
            f32x4 acc[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int c0 = j0 + g * 16 * NCT + j;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                constexpr int BIG = 8 * S_;
                if ((e + PAD - tap + BIG) % S_ == 0) {
                    const int d = (e + PAD - tap + BIG) / S_ - 8;
                    sf_mm16<16, NCT>(acc, af + tap * 16, lds + src + q * lsi + 4 + c0 + d, 4 * lsi);
                }
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int p = S_ * (c0 + 16 * ct) + e;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = fmaxf(acc[ct][r] + af[TAPS * 16 + r], 0.f);
                    if (FULL || p < pend) lds[dst + (row + r) * lsd + 4 + p] = v;
                }
            }
        }
    }
}

// ---- 16-row tiles on the bf16 MFMA (v_mfma_f32_16x16x32_bf16): acc[ct] += A * B over KS K-steps of 32 and NCT column tiles of
// 16.  A: three bf16 planes per K-step in af (12 words per K-step); lane l holds A[row l & 15][k = 32 ks + 8 (l >> 4) + j] and
// B[k = 32 ks + 8 (l >> 4) + j][column l & 15] (j < 8); S is the lane's base (row 8 (l >> 4), column l & 15), element (ks, j,
// ct) at S[(32 ks + j) * ls + 16 ct].  A unit = (K-step, column tile): 8 ds_read_b32 issued one unit ahead, the split of the
// NEXT unit's values between the six MFMAs of this one (the unit is vector-bound: 36 split instructions for 6 x 16 cycles).
template <int KS, int NCT>
__device__ __forceinline__ void sf_mm16_x3(f32x4 (&acc)[NCT], const float* af, const float* S, int ls) {
    constexpr int NU = KS * NCT;
    const sf_lptr S0 = sf_lds_base(S), S1 = sf_lds_base(S + (KS > 2 ? 64 : 0) * ls);      // 64 rows x 576 B per base
#define SF_B(u, j) ((32 * ((u) / NCT) + (j)) < 64 ? S0[(32 * ((u) / NCT) + (j)) * ls + 16 * ((u) % NCT)] \
                                                   : S1[(32 * ((u) / NCT) + (j) - 64) * ls + 16 * ((u) % NCT)])
#define SF_MF16(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc_, 0, 0, 0)
    auto plane = [&](int ks, int pl) __attribute__((always_inline)) {
        const float* p = af + (ks * 3 + pl) * 4;
        return u32x4{__builtin_bit_cast(unsigned, p[0]), __builtin_bit_cast(unsigned, p[1]), __builtin_bit_cast(unsigned, p[2]),
                     __builtin_bit_cast(unsigned, p[3])};
    };
    float x[8];
    u32x4 b0, b1, b2;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = SF_B(0, j);
    {
        unsigned u0, u1, u2;
#pragma unroll
        for (int w = 0; w < 4; ++w) { ctx_split2(x[2 * w], x[2 * w + 1], u0, u1, u2); b0[w] = u0; b1[w] = u1; b2[w] = u2; }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int ks = u / NCT, ct = u % NCT;
        const u32x4 a0 = plane(ks, 0), a1 = plane(ks, 1), a2 = plane(ks, 2);
        const bool nx = u + 1 < NU;
        u32x4 n0 = b0, n1 = b1, n2 = b2;
        unsigned u0, u1, u2;
        if (nx) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = SF_B(u + 1, j);
        }
        SF_SB(); SF_MF16(acc[ct], a2, b0); SF_SB();
        SF_SB(); SF_MF16(acc[ct], a0, b2); SF_SB();
        if (nx) { ctx_split2(x[0], x[1], u0, u1, u2); n0[0] = u0; n1[0] = u1; n2[0] = u2; }
        SF_SB(); SF_MF16(acc[ct], a1, b1); SF_SB();
        if (nx) { ctx_split2(x[2], x[3], u0, u1, u2); n0[1] = u0; n1[1] = u1; n2[1] = u2; }
        SF_SB(); SF_MF16(acc[ct], a1, b0); SF_SB();
        if (nx) { ctx_split2(x[4], x[5], u0, u1, u2); n0[2] = u0; n1[2] = u1; n2[2] = u2; }
        SF_SB(); SF_MF16(acc[ct], a0, b1); SF_SB();
        if (nx) { ctx_split2(x[6], x[7], u0, u1, u2); n0[3] = u0; n1[3] = u1; n2[3] = u2; }
        SF_SB(); SF_MF16(acc[ct], a0, b0); SF_SB();
        b0 = n0; b1 = n1; b2 = n2;
    }
#undef SF_B
}

// sf_pw16 on the split path: KS1 / KS2 K-steps of 32 over the two sources (af: 12 (KS1 + KS2) words of planes, then 4 bias values)
template <int KS1, int KS2, int NCT, bool SPLIT, bool FULL, bool RELU = true>
__device__ __forceinline__ void sf_pw16_x3(const float* af, float* lds, int src1, int ls1, int coff1, int src2, int ls2, int dst,
                                           int lsd, int P, int M, int rowbase = -1) {
    const int tid_ = sf_tid();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), q = lane >> 4, j = lane & 15;
    const int row = (rowbase >= 0 ? rowbase : (SPLIT ? 0 : 16 * wave)) + 4 * q;
    const int ng = (P + 16 * NCT - 1) / (16 * NCT);
    for (int g = SPLIT ? wave : 0; g < ng; g += SPLIT ? 4 : 1) {
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c0 = g * 16 * NCT + j;
        sf_mm16_x3<KS1, NCT>(acc, af, lds + src1 + 8 * q * ls1 + 4 + c0 + coff1, ls1);
        if constexpr (KS2 > 0) sf_mm16_x3<KS2, NCT>(acc, af + 12 * KS1, lds + src2 + 8 * q * ls2 + 4 + c0, ls2);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int col = c0 + 16 * ct;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[ct][r] + af[12 * (KS1 + KS2) + r];
                if (RELU) v = fmaxf(v, 0.f);
                if (FULL || (col < P && row + r < M)) lds[dst + (row + r) * lsd + 4 + col] = v;
            }
        }
    }
}

// sf_convT16 on the split path: per tap K = 64 = two K-steps (af: 24 words per tap, then 4 bias values)
template <int TAPS, int S_, int NCT, bool FULL>
__device__ __forceinline__ void sf_convT16_x3(const float* af, float* lds, int src, int lsi, int dst, int lsd, int Lout,
                                              int p0, int Ln) {
    const int tid_ = sf_tid();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), q = lane >> 4, j = lane & 15;
    constexpr int PAD = S_ / 2;
    const int pend = min(Lout, p0 + Ln);
    const int row = 16 * wave + 4 * q;
#pragma unroll
    for (int e = 0; e < S_; ++e) {
        const int j0 = p0 > e ? (p0 - e + S_ - 1) / S_ : 0;
        const int nj = (pend - e + S_ - 1) / S_ - j0;
        const int ng = (nj + 16 * NCT - 1) / (16 * NCT);
        for (int g = 0; g < ng; ++g) {
            f32x4 acc[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int c0 = j0 + g * 16 * NCT + j;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                constexpr int BIG = 8 * S_;
                if ((e + PAD - tap + BIG) % S_ == 0) {
                    const int d = (e + PAD - tap + BIG) / S_ - 8;
                    sf_mm16_x3<2, NCT>(acc, af + tap * 24, lds + src + 8 * q * lsi + 4 + c0 + d, lsi);
                }
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int p = S_ * (c0 + 16 * ct) + e;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = fmaxf(acc[ct][r] + af[TAPS * 24 + r], 0.f);
                    if (FULL || p < pend) lds[dst + (row + r) * lsd + 4 + p] = v;
                }
            }
        }
    }
}

// depthwise conv (k = K, stride S, padding K/2) + folded BatchNorm + ReLU; weights staged in LDS at `wl` ([C][K] then [C]).
// A lane produces 4 consecutive outputs of one channel row from the 16-byte quads that cover its input window
// (conflict-free: consecutive lanes read consecutive quads of a row).  Lout is a multiple of 4.
template <int K, int S>
__device__ __forceinline__ void sf_dw(float* lds, int src, int lsi, int dst, int lsd, int wl, int C, int Lout) {
    constexpr int NQ = (3 * S + K - 1 + K / 2 + 3) / 4 + 1;      // quads from 4 (j S - 1) on: covers [4 j S - K/2, 4 j S + 3 S + K/2]
    const int Q = Lout >> 2;
    for (int o = sf_tid(); o < C * Q; o += SF_T) {
        const int ch = o / Q, j = o - ch * Q;
        float w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = lds[wl + ch * K + k];
        const float b = lds[wl + C * K + ch];
        float in[4 * NQ];
        const float* ip = lds + src + ch * lsi + 4 + 4 * j * S - 4;        // 16-byte aligned
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4 t = *(const f32x4*)(ip + 4 * q);
            in[4 * q] = t[0]; in[4 * q + 1] = t[1]; in[4 * q + 2] = t[2]; in[4 * q + 3] = t[3];
        }
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = b;
#pragma unroll
            for (int k = 0; k < K; ++k) v = fmaf(w[k], in[4 + e * S + k - K / 2], v);     // column 4 j S + e S + k - K/2
            r[e] = fmaxf(v, 0.f);
        }
        *(f32x4*)(lds + dst + ch * lsd + 4 + 4 * j) = r;
    }
}

__device__ __forceinline__ void sf_stage(float* lds, int at, const float* g, int n) {
    for (int i = sf_tid(); i < n; i += SF_T) lds[at + i] = g[i];
}

// [C][L] dense (global scratch) <-> LDS buffer rows, 16 bytes per access; L = 4 << lq (rows start 16-byte aligned)
__device__ __forceinline__ void sf_save(const float* lds, int buf, int ls, float* g, int C, int lq) {
    for (int i = sf_tid(); i < (C << lq); i += SF_T) {
        const int ch = i >> lq, q = i - (ch << lq);
        ((f32x4*)g)[i] = *(const f32x4*)(lds + buf + ch * ls + 4 + 4 * q);
    }
}
// the same in two halves: up to 16 quads per thread are requested (registers) before a compute phase and written to LDS
// after it, so the L2 / Infinity-Cache latency of the skip tensor hides behind that phase
__device__ __forceinline__ void sf_restore_request(f32x4 (&rr)[16], const float* g, int C, int lq) {
    const int t = sf_tid();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = t + SF_T * j;
        if (i < (C << lq)) rr[j] = ((const f32x4*)g)[i];
    }
}
__device__ __forceinline__ void sf_restore_commit(const f32x4 (&rr)[16], float* lds, int buf, int ls, int C, int lq) {
    const int t = sf_tid();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = t + SF_T * j;
        if (i < (C << lq)) {
            const int ch = i >> lq, q = i - (ch << lq);
            *(f32x4*)(lds + buf + ch * ls + 4 + 4 * q) = rr[j];
        }
    }
    sf_guards(lds, buf, C, ls, 4 << lq);
}
__device__ __forceinline__ void sf_restore(float* lds, int buf, int ls, const float* g, int C, int lq) {
    for (int i = sf_tid(); i < (C << lq); i += SF_T) {
        const int ch = i >> lq, q = i - (ch << lq);
        *(f32x4*)(lds + buf + ch * ls + 4 + 4 * q) = ((const f32x4*)g)[i];
    }
    sf_guards(lds, buf, C, ls, 4 << lq);
}

struct SfArgs {
    const float* x; float* y; const float* blob; float* scratch;
    const float* h_in; float* h_out;            // TG: hidden state of the time-recurrent block, (streams, 128, 16) each
    int N, Cin;
    int o_first, o_pw[5], o_dw[5], o_gi, o_whh, o_fg, o_dpw[6], o_ct[5], o_last;
    int o_tg_rz, o_tg_in, o_tg_hn, o_tg_conv;
};

#define SF_SYNC() __syncthreads()
// diagnostic build only (scripts/stamps_stream.py, -DSF_STAMPS): cycle stamps of workgroup 0's first frame, written
// behind the skip regions of the scratch buffer (memory nothing else in the kernel reads)
#ifdef SF_STAMPS
#define SF_STAMP(i) do { if (blockIdx.x == 0 && tid == 0 && n == 0) \
    ((long long*)(A.scratch + (size_t)gridDim.x * SF_SKIP))[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define SF_STAMP(i) do { } while (0)
#endif

// Every matrix layer: (1) the fragments requested one layer ago move from the staging set to the compute set,
// (2) the NEXT layer's fragments are requested into the staging set, (3) the layer computes.  One compute set for all
// layers keeps a single copy of every unrolled MFMA body: the kernel must stay inside the instruction cache (a first
// version with two alternating sets and straight-line layers was 111 KB of code and ran its MFMAs at ~90 instead of 64
// cycles each, waiting for instruction fetch).
#define SF_TAKE(NQ) do { _Pragma("unroll") for (int i_ = 0; i_ < 4 * (NQ); ++i_) fs[i_] = fp[i_]; } while (0)
#define SF_REQUEST(NQ, OFF, TILE) do { sf_load<NQ>(fp, blob + (OFF) + (size_t)(TILE) * ((NQ) * 256), lane); \
                                       __builtin_amdgcn_sched_barrier(0); } while (0)

constexpr int LSA = sf_ls(128);          // one row stride (144 floats) for every activation buffer: immediate LDS offsets
constexpr int LSG = sf_ls(16);           // ... except the GRU projection [384][32]

// Which matrix layers multiply on the bf16 MFMA (the others run the fp32-MFMA code of stream_fwd.hip); export.x3_image builds the
// image for the mask the library reports.  Measured per mask (1x MI355X, 1024 frames per launch, scripts/dbg/stream_x3_ab.py;
// all-fp32-MFMA kernel 0.4397 ms): 1 -> 0.4086, 3 -> 0.4214, 7 -> 0.4273, 15 -> 0.4384, 19 -> 0.4305, 31 -> 0.4493 ms.  With one
// wave per SIMD the unit (6 MFMAs + the split of the next fragment: ~46 vector / LDS instructions) is bound by instruction issue
// at ~310 cycles: 1.7x over eight 66-cycle v_mfma_f32_32x32x2_f32, but slower than eight 35-cycle v_mfma_f32_16x16x4_f32 on the
// 16-row layers -- so only the 32-row encoder layers take the split path (and the kernel stays inside the 64 KB instruction
// cache: 55.5 KB; mask 31 is 71.5 KB).
#ifndef SFX_MASK
#define SFX_MASK 1
#endif
constexpr bool XE = (SFX_MASK & 1) != 0;     // encoder.1 .. encoder.5 pointwise (32-row tiles)
constexpr bool XD = (SFX_MASK & 2) != 0;     // decoder.1 .. decoder.4 pointwise (192 -> 64)
constexpr bool XC3 = (SFX_MASK & 4) != 0;    // decoder.2 / decoder.4 transposed conv (k3 s1)
constexpr bool XC5 = (SFX_MASK & 8) != 0;    // decoder.1 / decoder.3 transposed conv (k5 s2)
constexpr bool XO = (SFX_MASK & 16) != 0;    // GRU projection, FGRU.conv, decoder.0, decoder.5 pointwise, the TGRU block
constexpr int NQ_E1 = XE ? 16 : 12, NQ_E = XE ? 28 : 20;             // quads per 32-row tile: encoder.1, encoder.2..5
constexpr int NQ_T16 = XO ? 13 : 9;                                   // 16-row tile, K = 128
constexpr int NQ_D0 = XO ? 7 : 5, NQ_C0 = XO ? 19 : 13;               // decoder.0 pointwise, transposed conv (k3 s2)
constexpr int NQ_D = XD ? 19 : 13;                                    // decoder.1..4 pointwise (K = 192)
constexpr int NQ_C3 = XC3 ? 19 : 13, NQ_C5 = XC5 ? 31 : 21;           // transposed convs k3 s1 / k5 s2
constexpr int NQ_CMAX = NQ_C5 > NQ_C3 ? NQ_C5 : NQ_C3;
constexpr int NQ_RZ = XO ? 19 : 13, NQ_IN = XO ? 7 : 5;               // TGRU: r/z tile (K = 64 + 128), n rows of W_ih (K = 64)
constexpr int RQ_E = NQ_E > 2 * NQ_T16 ? NQ_E : 2 * NQ_T16;           // encoder loop request (the last one fetches a projection pass)
constexpr int NQ_MAX = RQ_E > NQ_CMAX ? RQ_E : NQ_CMAX;

// TG: with the time-recurrent block (network.py:150; GRUBlock :45-58) between FGRU.conv and decoder.0: one GRU time step per
// (stream, frequency position), hidden state (128 x 16 per stream) read from / written to HBM.  Its own instance, so that
// the stateless kernel's code footprint (instruction cache) stays what it was.
template <bool TG>
__global__ __launch_bounds__(SF_T, 1) void stream_fwd_x3_kernel(const SfArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* skip = A.scratch + (size_t)blockIdx.x * SF_SKIP;
    float* sk0 = skip, *sk1 = sk0 + 8192, *sk2 = sk1 + 16384, *sk3 = sk2 + 8192, *sk4 = sk3 + 8192;
    float fs[4 * NQ_MAX], fp[4 * NQ_MAX];                      // compute set / staging set (A fragments + bias values of a wave's row tile(s))
    const int Cin = A.Cin;

    // first-conv weights: resident for the whole kernel
    sf_stage(lds, SF_W0, A.blob + A.o_first, 64 * Cin * 5 + 64);
    // features of a frame as 5 registers per thread (zero guard / pad columns included), requested one frame ahead
    float xr[5];
    auto request_x = [&](int nn) __attribute__((always_inline)) {
        constexpr int LSX = sf_ls(257);
        const float* xg = A.x + (size_t)nn * Cin * 257;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int i = tid + SF_T * j;
            const int ch = i / LSX, col = i - ch * LSX - 4;
            xr[j] = (nn < A.N && ch < Cin && col >= 0 && col < 257) ? xg[ch * 257 + col] : 0.f;
        }
    };
    request_x(blockIdx.x);
    for (int n = blockIdx.x; n < A.N; n += gridDim.x) {
        // The weights do not change from frame to frame, so the compiler would hoist EVERY layer's fragment loads out
        // of this loop (thousands of registers, all spilled): make the base pointer opaque once per frame.
        // (An opaque OFFSET, not an opaque pointer: the compiler must still see a global-memory address, or it emits
        // flat loads, which also count on lgkmcnt and would make every LDS wait drain the weight prefetch.)
        int opaque0 = 0;
        asm volatile("" : "+s"(opaque0));
        const float* blob = A.blob + opaque0;
        // ---------------- features -> LDS, first conv (C_in -> 64, k5 s2 p1) + ReLU            network.py:9-21
        SF_STAMP(0);
        SF_REQUEST(NQ_E1, A.o_pw[0], wave);                                 // encoder.1 pw (K = 64)
        {
            constexpr int LSX = sf_ls(257);
            // the frame's features were requested at the end of the previous frame (xr): zero-padded rows into LDS
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int i = tid + SF_T * j;
                if (i < Cin * LSX) lds[SF_XB + i] = xr[j];
            }
            SF_SYNC();
            {
                // thread = (position lo, half of the output channels): its C_in x 5 input window stays in registers,
                // the weights of one output channel are a wave-uniform (broadcast) LDS read
                const int t_ = sf_tid();
                const int lo = t_ & 127, cg = __builtin_amdgcn_readfirstlane(t_ >> 7);
                float xin[4][5];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int k = 0; k < 5; ++k)
                        xin[ci][k] = ci < Cin ? lds[SF_XB + ci * LSX + 4 + 2 * lo - 1 + k] : 0.f;
                if (Cin == 4) {
                    // 20 weights of one output channel = 5 aligned quads (wave-uniform, broadcast reads)
                    for (int co = 32 * cg; co < 32 * cg + 32; co += 2) {
                        float v0 = lds[SF_W0 + 1280 + co], v1 = lds[SF_W0 + 1280 + co + 1];
#pragma unroll
                        for (int q5 = 0; q5 < 5; ++q5) {
                            const f32x4 w0 = *(const f32x4*)(lds + SF_W0 + co * 20 + 4 * q5);
                            const f32x4 w1 = *(const f32x4*)(lds + SF_W0 + (co + 1) * 20 + 4 * q5);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int idx = 4 * q5 + e;                   // = ci * 5 + k
                                v0 = fmaf(w0[e], xin[idx / 5][idx % 5], v0);
                                v1 = fmaf(w1[e], xin[idx / 5][idx % 5], v1);
                            }
                        }
                        lds[SF_R0 + co * LSA + 4 + lo] = fmaxf(v0, 0.f);
                        lds[SF_R0 + (co + 1) * LSA + 4 + lo] = fmaxf(v1, 0.f);
                    }
                } else
                for (int co = 32 * cg; co < 32 * cg + 32; co += 4) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = lds[SF_W0 + 64 * Cin * 5 + co + j];
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) {
                        if (ci < Cin) {
#pragma unroll
                            for (int k = 0; k < 5; ++k)
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    v[j] = fmaf(lds[SF_W0 + (co + j) * Cin * 5 + ci * 5 + k], xin[ci][k], v[j]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) lds[SF_R0 + (co + j) * LSA + 4 + lo] = fmaxf(v[j], 0.f);
                }
            }
            sf_guards(lds, SF_R0, 64, LSA, 128);
            SF_SYNC();
            sf_save(lds, SF_R0, LSA, sk0, 64, 5);
        }
        // ---------------- encoder.1 .. encoder.5 (pointwise + BN + ReLU, depthwise + BN + ReLU)   network.py:24-43
        SF_STAMP(1);
        {   // encoder.1: 64 -> 128, L 128, dw k3 s1
            float dwr[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) dwr[j] = blob[A.o_dw[0] + tid + SF_T * j];
            SF_TAKE(NQ_E1);
            SF_REQUEST(NQ_E, A.o_pw[1], wave);
            if constexpr (XE) sf_pw_x3<4>(fs, lds, SF_R0, LSA, SF_R1A, LSA, 128);
            else sf_pw<32, 0, 4>(fs, lds, SF_R0, LSA, 0, 0, 0, SF_R1A, LSA, 128, 128, 0, true);
            sf_guards(lds, SF_R1A, 128, LSA, 128);
#pragma unroll
            for (int j = 0; j < 2; ++j) lds[SF_DWB + tid + SF_T * j] = dwr[j];
            SF_SYNC();
            sf_dw<3, 1>(lds, SF_R1A, LSA, SF_R0, LSA, SF_DWB, 128, 128);
            sf_guards(lds, SF_R0, 128, LSA, 128);
            SF_SYNC();
            sf_save(lds, SF_R0, LSA, sk1, 128, 5);
        }
        SF_STAMP(2);
        // encoder.2 .. encoder.5: 128 -> 128 pointwise (ONE unrolled 128 x 128 MFMA body), then depthwise k5 s2 / k3 s1 /
        // k5 s2 / k3 s2                                                                       network.py:24-43
        for (int it = 0; it < 4; ++it) {
            const int L = it == 0 ? 128 : (it <= 2 ? 64 : 32);      // positions of this layer's input
            // (every iteration issues the SAME sequence of global loads, selected by offsets and not by branches: with
            // divergent paths hipcc's s_waitcnt pass merges pessimistically at the loop header and makes the MFMAs
            // below wait for the fragments that were only just requested for the NEXT layer)
            float dwr[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) dwr[j] = blob[A.o_dw[1 + it] + tid + SF_T * j];
            if (it == 1) SF_STAMP(27);
            SF_TAKE(RQ_E);
#ifdef SF_STAMPS
            if (it == 1) { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); SF_STAMP(28); }
#endif
            {
                // next: encoder.(3 + it) (20 quads per 32-row tile), or the first projection pass (two 16-row tiles = 18
                // quads; the request reads 20: inside the blob)
                const int off_ = it < 3 ? A.o_pw[2 + it] + wave * (NQ_E * 256) : A.o_gi + 2 * wave * (NQ_T16 * 256);
                SF_REQUEST(RQ_E, off_, 0);
            }
            if (it == 1) SF_STAMP(29);
            if constexpr (XE) sf_pw_x3<8>(fs, lds, SF_R0, LSA, SF_R1A, LSA, L);
            else sf_pw<64, 0, 4>(fs, lds, SF_R0, LSA, 0, 0, 0, SF_R1A, LSA, L, 128, 0, true);
            if (it == 1) SF_STAMP(30);
            sf_guards(lds, SF_R1A, 128, LSA, L);
#pragma unroll
            for (int j = 0; j < 3; ++j) lds[SF_DWB + tid + SF_T * j] = dwr[j];
            SF_SYNC();
            if (it == 1) sf_dw<3, 1>(lds, SF_R1A, LSA, SF_R0, LSA, SF_DWB, 128, 64);
            else if (it == 3) sf_dw<3, 2>(lds, SF_R1A, LSA, SF_R0, LSA, SF_DWB, 128, 16);
            else sf_dw<5, 2>(lds, SF_R1A, LSA, SF_R0, LSA, SF_DWB, 128, L >> 1);
            const int Lo = it == 1 ? 64 : (L >> 1);
            sf_guards(lds, SF_R0, 128, LSA, Lo);
            SF_SYNC();
            if (it == 1) SF_STAMP(31);
            if (it < 3) sf_save(lds, SF_R0, LSA, it == 0 ? sk2 : (it == 1 ? sk3 : sk4), 128, it == 2 ? 3 : 4);
            SF_STAMP(3 + it);
        }
        // GRU input projection (384 x 128, both directions; network.py:45-58,149) over the 16 positions: 24 row tiles of
        // 16 (one exact 16 x 16 tile each; 32-row tiles were half empty), three passes of two tiles per wave
        for (int ps = 0; ps < 3; ++ps) {
            SF_TAKE(2 * NQ_T16);
            {
                // next pass, or FGRU.conv (128 -> 64: four 16-row tiles; the request reads two)
                const int off_ = ps < 2 ? A.o_gi + (8 * (ps + 1) + 2 * wave) * (NQ_T16 * 256) : A.o_fg + wave * (NQ_T16 * 256);
                SF_REQUEST(2 * NQ_T16, off_, 0);
            }
#pragma unroll
            for (int rtl = 0; rtl < 2; ++rtl) {
                if constexpr (XO)
                    sf_pw16_x3<4, 0, 1, false, true, false>(fs + 4 * NQ_T16 * rtl, lds, SF_R0, LSA, 0, 0, 0, SF_R1A, LSG, 16, 384,
                                                            16 * (8 * ps + 2 * wave + rtl));
                else
                    sf_pw16<32, 0, 1, false, true, false>(fs + 4 * NQ_T16 * rtl, lds, SF_R0, LSA, 0, 0, 0, SF_R1A, LSG, 16, 384, nullptr,
                                                          16 * (8 * ps + 2 * wave + rtl));
            }
            SF_STAMP(10 + ps);
        }
        SF_SYNC();
#if defined(SF_ABL) && (SF_ABL & 4)      // diagnostic: encoder + projection only (does a smaller code footprint stay in the I-cache?)
        SF_STAMP(7); SF_STAMP(23); SF_STAMP(26); SF_STAMP(8); SF_STAMP(9); SF_STAMP(13); SF_STAMP(14);
        request_x(n + gridDim.x);
        continue;
#endif
        {
            SF_STAMP(7);
            // recurrence: direction d = tid >> 7 (two waves each); hidden unit j = (tid & 127) >> 1 is owned by the lane
            // PAIR (2 j, 2 j + 1): lane half kh = tid & 1 holds the K-half [32 kh, 32 kh + 32) of the three W_hh rows of
            // the unit (r, z, n: 96 registers for all 16 steps), the halves are combined with one cross-lane add (DPP),
            // both lanes evaluate the gates and the even one writes h: ONE barrier per step, no exchange of W_hh h.
            const int t_ = sf_tid();
            const int d = t_ >> 7, j = (t_ & 127) >> 1, kh = t_ & 1;
            // exporter layout: [direction][24 quads][128 threads][4]: thread (2 j + kh), quad 8 g + i = W_hh[g*64 + j][32 kh +
            // 4 i .. + 3]; then b_hh [2][192] -- every load is 16 bytes per lane, consecutive lanes consecutive
            const f32x4* whh = (const f32x4*)(blob + A.o_whh) + (size_t)d * 24 * 128 + (t_ & 127);
            const float* bhh = blob + A.o_whh + 2 * 24 * 128 * 4 + d * 192;
            float wr[32], wz[32], wn[32];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 tr = whh[i * 128], tz = whh[(8 + i) * 128], tn = whh[(16 + i) * 128];
#pragma unroll
                for (int e = 0; e < 4; ++e) { wr[4 * i + e] = tr[e]; wz[4 * i + e] = tz[e]; wn[4 * i + e] = tn[e]; }
            }
            // opaque copies: the compiler must keep them in registers instead of re-loading them from memory every step
#pragma unroll
            for (int k = 0; k < 32; ++k) asm volatile("" : "+v"(wr[k]), "+v"(wz[k]), "+v"(wn[k]));
            const float br = bhh[j], bz = bhh[64 + j], bn = bhh[128 + j];
            float* hs = lds + SF_GRU + d * 128;              // h of this direction, double-buffered: [2][64]
            if ((t_ & 127) < 64) hs[t_ & 127] = 0.f;
            float hme = 0.f;                                 // h_{t-1}[j]
            SF_SYNC();
            SF_STAMP(23);
            for (int st = 0; st < 16; ++st) {
                const int pos = d ? 15 - st : st;
                const float* hc = hs + (st & 1) * 64 + 32 * kh;
                const float* gi = lds + SF_R1A + (d * 192 + j) * LSG + 4 + pos;
                const float gir = gi[0], giz = gi[64 * LSG], gin = gi[128 * LSG];
                f32x4 hv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) hv[i] = *(const f32x4*)(hc + 4 * i);
                float r0 = 0.f, r1 = 0.f, z0 = 0.f, z1 = 0.f, n0_ = 0.f, n1 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    r0 = fmaf(wr[4 * i], hv[i][0], r0); r1 = fmaf(wr[4 * i + 1], hv[i][1], r1);
                    r0 = fmaf(wr[4 * i + 2], hv[i][2], r0); r1 = fmaf(wr[4 * i + 3], hv[i][3], r1);
                    z0 = fmaf(wz[4 * i], hv[i][0], z0); z1 = fmaf(wz[4 * i + 1], hv[i][1], z1);
                    z0 = fmaf(wz[4 * i + 2], hv[i][2], z0); z1 = fmaf(wz[4 * i + 3], hv[i][3], z1);
                    n0_ = fmaf(wn[4 * i], hv[i][0], n0_); n1 = fmaf(wn[4 * i + 1], hv[i][1], n1);
                    n0_ = fmaf(wn[4 * i + 2], hv[i][2], n0_); n1 = fmaf(wn[4 * i + 3], hv[i][3], n1);
                }
                float gr = r0 + r1, gz = z0 + z1, gn = n0_ + n1;
                // the other K-half: lane ^ 1 through DPP (quad_perm [1,0,3,2]); __shfl_xor compiles to ds_bpermute, an LDS
                // round trip on the critical path of every step
                gr += sf_dpp_xor1(gr); gz += sf_dpp_xor1(gz); gn += sf_dpp_xor1(gn);
                const float r = sf_sigmoid(gir + gr + br);
                const float z = sf_sigmoid(giz + gz + bz);
                const float nn = sf_tanh(fmaf(r, gn + bn, gin));
                hme = (1.f - z) * nn + z * hme;
                if (kh == 0) {
                    hs[((st + 1) & 1) * 64 + j] = hme;
                    lds[SF_R0 + (d * 64 + j) * LSA + 4 + pos] = hme;
                }
                SF_SYNC();
                if (st == 0) SF_STAMP(24);
                if (st == 8) SF_STAMP(25);
            }
            SF_STAMP(26);
            sf_guards(lds, SF_R0, 128, LSA, 16);
            SF_SYNC();
            SF_STAMP(8);
            // FGRU.conv (128 -> 64) + BN + ReLU
            SF_TAKE(NQ_T16);
            if constexpr (TG) SF_REQUEST(NQ_RZ, A.o_tg_rz, wave);             // TGRU r/z rows, first pass
            else SF_REQUEST(NQ_D0, A.o_dpw[0], wave);                         // decoder.0 pw: 64 -> 64
            if constexpr (XO) sf_pw16_x3<4, 0, 1, false, true>(fs, lds, SF_R0, LSA, 0, 0, 0, SF_R1A, LSA, 16, 64);
            else sf_pw16<32, 0, 1, false, true>(fs, lds, SF_R0, LSA, 0, 0, 0, SF_R1A, LSA, 16, 64);
            sf_guards(lds, SF_R1A, 64, LSA, 16);
            if constexpr (TG) {
                // ---------------- TGRU (network.py:150): x = R1A (64 x 16); h_{t-1} of this stream -> R1B as [128][LSG]
                sf_restore(lds, SF_R1B, LSG, A.h_in + (size_t)n * 2048, 128, 2);
                SF_SYNC();
                // r and z rows: W_ih x + W_hh h + (b_ih + b_hh) as ONE K = 64 + 128 GEMM, sixteen 16-row tiles in four
                // passes of one tile per wave -> R0 rows [0, 256) (stride LSG); n rows: W_in x + b_in -> rows [256, 384),
                // W_hn h + b_hn -> rows [384, 512) (kept apart: r multiplies the latter only), two passes of 1 + 1 tiles
                for (int ps = 0; ps < 4; ++ps) {
                    SF_TAKE(NQ_RZ);
                    {
                        // next r/z pass, or the first n pass: W_in tile + W_hn tile, requested as NQ_IN + NQ_T16 quads (for an r/z
                        // tile the same two loads fetch NQ_RZ + 1 contiguous quads)
                        const int off_ = ps < 3 ? A.o_tg_rz + (4 * (ps + 1) + wave) * (NQ_RZ * 256) : A.o_tg_in + wave * (NQ_IN * 256);
                        SF_REQUEST(NQ_IN, off_, 0);
                        const int off2_ = ps < 3 ? off_ + NQ_IN * 256 : A.o_tg_hn + wave * (NQ_T16 * 256);
                        sf_load<NQ_T16>(fp + 4 * NQ_IN, blob + off2_, lane);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (XO)
                        sf_pw16_x3<2, 4, 1, false, true, false>(fs, lds, SF_R1A, LSA, 0, SF_R1B, LSG, SF_R0, LSG, 16, 512,
                                                                16 * (4 * ps + wave));
                    else
                        sf_pw16<16, 32, 1, false, true, false>(fs, lds, SF_R1A, LSA, 0, SF_R1B, LSG, SF_R0, LSG, 16, 512, nullptr,
                                                               16 * (4 * ps + wave));
                }
                for (int ps = 0; ps < 2; ++ps) {
                    SF_TAKE(NQ_IN + NQ_T16);
                    {
                        // next n pass, or TGRU.conv (one tile, requested as NQ_IN quads + the ones behind them: contiguous)
                        const int off_ = ps < 1 ? A.o_tg_in + (4 + wave) * (NQ_IN * 256) : A.o_tg_conv + wave * (NQ_T16 * 256);
                        SF_REQUEST(NQ_IN, off_, 0);
                        const int off2_ = ps < 1 ? A.o_tg_hn + (4 + wave) * (NQ_T16 * 256) : A.o_tg_conv + wave * (NQ_T16 * 256) + NQ_IN * 256;
                        sf_load<NQ_T16>(fp + 4 * NQ_IN, blob + off2_, lane);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (XO) {
                        sf_pw16_x3<2, 0, 1, false, true, false>(fs, lds, SF_R1A, LSA, 0, 0, 0, SF_R0, LSG, 16, 512,
                                                                256 + 16 * (4 * ps + wave));
                        sf_pw16_x3<4, 0, 1, false, true, false>(fs + 4 * NQ_IN, lds, SF_R1B, LSG, 0, 0, 0, SF_R0, LSG, 16, 512,
                                                                384 + 16 * (4 * ps + wave));
                    } else {
                        sf_pw16<16, 0, 1, false, true, false>(fs, lds, SF_R1A, LSA, 0, 0, 0, SF_R0, LSG, 16, 512, nullptr,
                                                              256 + 16 * (4 * ps + wave));
                        sf_pw16<32, 0, 1, false, true, false>(fs + 4 * NQ_IN, lds, SF_R1B, LSG, 0, 0, 0, SF_R0, LSG, 16, 512, nullptr,
                                                              384 + 16 * (4 * ps + wave));
                    }
                }
                SF_SYNC();
                {
                    // gates (torch.nn.GRU order r, z, n): thread = (hidden unit u, 8 of the 16 positions)
                    const int t_ = sf_tid();
                    const int u = t_ >> 1, p0 = 8 * (t_ & 1);
                    const float* gr = lds + SF_R0 + u * LSG + 4 + p0;
                    float* hp = lds + SF_R1B + u * LSG + 4 + p0;
                    float* hg = A.h_out + (size_t)n * 2048 + u * 16 + p0;
#pragma unroll
                    for (int q4 = 0; q4 < 2; ++q4) {
                        const f32x4 ar = *(const f32x4*)(gr + 4 * q4), az = *(const f32x4*)(gr + 128 * LSG + 4 * q4);
                        const f32x4 an = *(const f32x4*)(gr + 256 * LSG + 4 * q4), ah = *(const f32x4*)(gr + 384 * LSG + 4 * q4);
                        const f32x4 hv = *(const f32x4*)(hp + 4 * q4);
                        f32x4 ho;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float r = sf_sigmoid(ar[e]);
                            const float z = sf_sigmoid(az[e]);
                            const float nn = sf_tanh(fmaf(r, ah[e], an[e]));
                            ho[e] = (1.f - z) * nn + z * hv[e];
                        }
                        *(f32x4*)(hp + 4 * q4) = ho;
                        *(f32x4*)(hg + 4 * q4) = ho;
                    }
                }
                SF_SYNC();
                // TGRU.conv (128 -> 64) + BN + ReLU on h_t -> R1A (the decoder's input, as without the block)
                SF_TAKE(NQ_T16);
                SF_REQUEST(NQ_D0, A.o_dpw[0], wave);                          // decoder.0 pw: 64 -> 64
                if constexpr (XO) sf_pw16_x3<4, 0, 1, false, true>(fs, lds, SF_R1B, LSG, 0, 0, 0, SF_R1A, LSA, 16, 64);
                else sf_pw16<32, 0, 1, false, true>(fs, lds, SF_R1B, LSG, 0, 0, 0, SF_R1A, LSA, 16, 64);
                sf_guards(lds, SF_R1A, 64, LSA, 16);
            }
            SF_SYNC();
            // ---------------- decoder.0 (FirstTrCNN): pw 64 -> 64, ConvT k3 s2 -> L 31            network.py:60-76
            SF_TAKE(NQ_D0);
            SF_REQUEST(NQ_C0, A.o_ct[0], wave);
            if constexpr (XO) sf_pw16_x3<2, 0, 1, false, true>(fs, lds, SF_R1A, LSA, 0, 0, 0, SF_R1B, LSA, 16, 64);
            else sf_pw16<16, 0, 1, false, true>(fs, lds, SF_R1A, LSA, 0, 0, 0, SF_R1B, LSA, 16, 64);
            sf_guards(lds, SF_R1B, 64, LSA, 16);
            SF_SYNC();
            sf_restore(lds, SF_R0, LSA, sk4, 128, 3);
            SF_TAKE(NQ_C0);
            SF_REQUEST(NQ_D, A.o_dpw[1], wave);                               // decoder.1 pw: 192 -> 64
            if constexpr (XO) sf_convT16_x3<3, 2, 1, false>(fs, lds, SF_R1B, LSA, SF_R1A, LSA, 31, 0, 31);
            else sf_convT16<3, 2, 1, false>(fs, lds, SF_R1B, LSA, SF_R1A, LSA, 31, 0, 31);
            sf_guards(lds, SF_R1A, 64, LSA, 31);
            SF_SYNC();
        }
        // ---------------- decoder.1 .. decoder.4 (TrCNN): [x1 padded / cropped | skip] -> pw 192 -> 64 -> ConvT
        //                                                                                       network.py:79-100
        // one loop body for the four blocks: i = 1: x1 L 31 (pad right 1), skip enc4 L 32, ConvT k5 s2 -> 65;
        // i = 2: x1 65 (crop left 1), skip enc3 L 64, k3 s1 -> 66;  i = 3: x1 66 (crop 1 each side), skip enc2 L 64,
        // k5 s2 -> 129;  i = 4: x1 129 (crop left 1), skip enc1 L 128, k3 s1 -> 130.  The next block's skip tensor is
        // restored into R0 while this block's ConvT runs.
        SF_STAMP(9);
        for (int i = 1; i <= 4; ++i) {
            const int P = i == 1 ? 32 : (i == 4 ? 128 : 64);
            const int Lo = i == 1 ? 65 : (i == 2 ? 66 : (i == 3 ? 129 : 130));
            SF_TAKE(NQ_D);
            // same load sequence in every iteration (see the encoder loop): the larger of the two transposed convs' fragment
            // counts is requested for both (the tile stride in the blob is that of the layer's own count)
            sf_load<NQ_CMAX>(fp, blob + A.o_ct[i] + (size_t)wave * (((i & 1) ? NQ_C5 : NQ_C3) * 256), lane);
            __builtin_amdgcn_sched_barrier(0);
            // skip tensor of the next block: enc3 (L 64), enc2 (L 64), enc1 (L 128), enc0 (64 x 128): requested a whole
            // block ahead (all workgroups restore at about the same time and the 45 MB of skip tensors live in the
            // Infinity Cache, not in the L2: the burst needs the time), written to R0 after the ConvT (R0's current
            // content, this block's skip, is last read by the pointwise conv below).  Issued AFTER the ConvT's fragments:
            // vmcnt retires in order, and those are needed first
            f32x4 rr[16];
            const float* skn = i == 1 ? sk3 : (i == 2 ? sk2 : (i == 3 ? sk1 : sk0));
            const int skC = i == 4 ? 64 : 128, sklq = i <= 2 ? 4 : 5;
            sf_restore_request(rr, skn, skC, sklq);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (XD) sf_pw16_x3<2, 4, 2, false, true>(fs, lds, SF_R1A, LSA, i == 1 ? 0 : 1, SF_R0, LSA, SF_R1B, LSA, P, 64);
            else sf_pw16<16, 32, 2, false, true>(fs, lds, SF_R1A, LSA, i == 1 ? 0 : 1, SF_R0, LSA, SF_R1B, LSA, P, 64);
            sf_guards(lds, SF_R1B, 64, LSA, P);
            SF_SYNC();
            SF_STAMP(13 + 2 * i);                             // 15, 17, 19, 21: pointwise conv of block i done
            SF_TAKE(NQ_CMAX);
            // next pw: decoder.(i+1) 192 -> 64, or decoder.5 128 -> 8 (one padded tile of NQ_T16 <= NQ_D quads)
            sf_load<NQ_D>(fp, blob + A.o_dpw[i + 1] + (size_t)(i < 4 ? wave : 0) * (NQ_D * 256), lane);
            __builtin_amdgcn_sched_barrier(0);
            // the consumer (next block / decoder.5) reads positions [1, 1 + Pn) of this output
            const int Pn = i <= 2 ? 64 : 128;
            if (i & 1) {
                if constexpr (XC5) sf_convT16_x3<5, 2, 2, true>(fs, lds, SF_R1B, LSA, SF_R1A, LSA, Lo, 1, Pn);
                else sf_convT16<5, 2, 2, true>(fs, lds, SF_R1B, LSA, SF_R1A, LSA, Lo, 1, Pn);
            } else {
                if constexpr (XC3) sf_convT16_x3<3, 1, 2, true>(fs, lds, SF_R1B, LSA, SF_R1A, LSA, Lo, 1, Pn);
                else sf_convT16<3, 1, 2, true>(fs, lds, SF_R1B, LSA, SF_R1A, LSA, Lo, 1, Pn);
            }
            sf_restore_commit(rr, lds, SF_R0, LSA, skC, sklq);
            sf_guards(lds, SF_R1A, 64, LSA, Lo);
            SF_SYNC();
            SF_STAMP(14 + 2 * i);                             // 16, 18, 20, 22: block i done
        }
        SF_STAMP(13);
        {   // ---------------- decoder.5 (LastTrCNN): pw 128 -> 8 (+BN+ReLU), ConvT 8 -> 8 k5 s2 -> 257, linear
            //                                                                                   network.py:102-120
            SF_TAKE(NQ_T16);
            if constexpr (XO) sf_pw16_x3<2, 2, 2, true, false>(fs, lds, SF_R1A, LSA, 1, SF_R0, LSA, SF_R1B, LSA, 128, 8);
            else sf_pw16<16, 16, 2, true, false>(fs, lds, SF_R1A, LSA, 1, SF_R0, LSA, SF_R1B, LSA, 128, 8);
            sf_guards(lds, SF_R1B, 8, LSA, 128);
            // ConvT weights [ci][co][k] + bias, staged as [co / 4][ci][k][co % 4]: a thread's four output channels of one
            // (ci, tap) are one wave-uniform 16-byte read
            for (int i = tid; i < 8 * 8 * 5 + 8; i += SF_T) {
                int d = i;
                if (i < 320) {
                    const int ci = i / 40, r = i - ci * 40, co = r / 5, k = r - co * 5;
                    d = (((co >> 2) * 8 + ci) * 5 + k) * 4 + (co & 3);
                }
                lds[SF_DWB + d] = blob[A.o_last + i];
            }
            request_x(n + gridDim.x);                                       // next frame's features, in flight over the tail
            SF_SYNC();
            float* yg = A.y + (size_t)n * 8 * 257;
            {
                // thread = (source position j, four output channels): outputs p = 2 j (taps 1, 3 at q = j, j - 1) and
                // p = 2 j + 1 (taps 0, 2, 4 at q = j + 1, j, j - 1); the guards cover q = -1 and q = 128.  24 input reads and
                // 40 weight reads for 160 FMAs (the per-output form read two LDS words per FMA)
                const int t_ = sf_tid();
                const int j = t_ & 127, cg = __builtin_amdgcn_readfirstlane(t_ >> 7);
                const float* in = lds + SF_R1B + 4 + j;
                const f32x4* W4 = (const f32x4*)(lds + SF_DWB) + cg * 40;
                f32x4 a0 = *(const f32x4*)(lds + SF_DWB + 320 + 4 * cg), a1 = a0;
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) {
                    const float xm = in[ci * LSA - 1], x0 = in[ci * LSA], xp = in[ci * LSA + 1];
                    const f32x4 w0 = W4[ci * 5], w1 = W4[ci * 5 + 1], w2 = W4[ci * 5 + 2], w3 = W4[ci * 5 + 3],
                                w4 = W4[ci * 5 + 4];
                    a0 += w1 * x0 + w3 * xm;
                    a1 += w0 * xp + w2 * x0 + w4 * xm;
                }
                float* yb = yg + (size_t)(4 * cg) * 257 + 2 * j;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) { yb[c4 * 257] = a0[c4]; yb[c4 * 257 + 1] = a1[c4]; }
                if (t_ < 8) {               // p = 256: taps 1 (q = 128: zero guard) and 3 (q = 127)
                    float v = lds[SF_DWB + 320 + t_];
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci)
                        v = fmaf(lds[SF_DWB + (((t_ >> 2) * 8 + ci) * 5 + 3) * 4 + (t_ & 3)], lds[SF_R1B + ci * LSA + 4 + 127], v);
                    yg[t_ * 257 + 256] = v;
                }
            }
            SF_SYNC();
        }
        SF_STAMP(14);
    }
}

}  // namespace

extern "C" int trunet_stream_fwd_grid(int N);

// which layer groups this build runs on the split path (bit mask: stream_fwd_x3.hip, SFX_MASK): the image must match it
extern "C" int trunet_stream_fwd_x3_mask(void) { return SFX_MASK; }

// Every section of the exported image must lie inside the blob, fragment over-reads (the kernel requests fixed-size blocks
// of up to 21 quads per wave) included: a truncated or foreign artefact is refused here instead of faulting on the GPU.
extern "C" int trunet_stream_fwd_x3_check(const int32_t* h_offsets, int n_offsets, int64_t blob_numel, int Cin) {
    if (!h_offsets || n_offsets != 30 || (Cin != 3 && Cin != 4) || blob_numel <= 0) return TRUNET_EINVAL;
    const int64_t T32 = 256;           // floats per quad of a tile
    int64_t size[30];
    int i = 0;
    size[i++] = 64 * Cin * 5 + 64;                                     // first conv
    size[i++] = 4 * NQ_E1 * T32;                                       // encoder.1 pw (K = 64, 32-row tiles)
    for (int k = 0; k < 4; ++k) size[i++] = 4 * NQ_E * T32;            // encoder.2..5 pw (K = 128)
    { const int ks[5] = {3, 5, 3, 5, 3}; for (int k = 0; k < 5; ++k) size[i++] = 128 * ks[k] + 128; }
    size[i++] = 24 * NQ_T16 * T32;                                     // FGRU input projection (384 x 128, 16-row tiles)
    size[i++] = 2 * 24 * 128 * 4 + 384;                                // W_hh, b_hh
    size[i++] = 4 * NQ_T16 * T32;                                      // FGRU.conv
    size[i++] = 4 * NQ_D0 * T32;                                       // decoder.0 pw
    for (int k = 0; k < 4; ++k) size[i++] = 4 * NQ_D * T32;            // decoder.1..4 pw (K = 192)
    size[i++] = 1 * NQ_T16 * T32;                                      // decoder.5 pw (8 rows)
    size[i++] = 4 * NQ_C0 * T32;                                       // transposed convs: decoder.0 (k3 s2), then k5 / k3 / k5 / k3
    size[i++] = 4 * NQ_C5 * T32; size[i++] = 4 * NQ_C3 * T32; size[i++] = 4 * NQ_C5 * T32; size[i++] = 4 * NQ_C3 * T32;
    size[i++] = 8 * 8 * 5 + 8;                                         // last ConvT
    size[i++] = 16 * NQ_RZ * T32; size[i++] = 8 * NQ_IN * T32; size[i++] = 8 * NQ_T16 * T32; size[i++] = 4 * NQ_T16 * T32;   // TGRU
    const bool tg = h_offsets[26] > 0;
    for (int k = 0; k < 30; ++k) {
        if (k >= 26 && !tg) { if (h_offsets[k] != 0) return TRUNET_EINVAL; continue; }
        const int64_t o = h_offsets[k];
        if (o < 0 || (o & 3)) return TRUNET_EINVAL;
        if (k >= 26 && o == 0) return TRUNET_EINVAL;
        if (o + size[k] + NQ_MAX * T32 > blob_numel) return TRUNET_EINVAL;
    }
    return TRUNET_OK;
}

extern "C" int trunet_stream_fwd_x3(const float* x, float* y, const float* blob, const int32_t* h_offsets, int n_offsets,
                                 int64_t blob_numel, float* scratch, const float* h_in, float* h_out, int N, int Cin,
                                 void* stream) {
    if (!x || !y || !blob || !h_offsets || !scratch || N <= 0) return TRUNET_EINVAL;
    if (Cin != 3 && Cin != 4) return TRUNET_ENOTSUP;
    if ((h_in == nullptr) != (h_out == nullptr)) return TRUNET_EINVAL;
    {
        const int rc = trunet_stream_fwd_x3_check(h_offsets, n_offsets, blob_numel, Cin);
        if (rc != TRUNET_OK) return rc;
    }
    const bool tg = h_in != nullptr;
    if (tg && h_offsets[26] <= 0) return TRUNET_EINVAL;          // state given, block not exported
    SfArgs a;
    a.x = x; a.y = y; a.blob = blob; a.scratch = scratch; a.h_in = h_in; a.h_out = h_out; a.N = N; a.Cin = Cin;
    int i = 0;
    a.o_first = h_offsets[i++];
    for (int k = 0; k < 5; ++k) a.o_pw[k] = h_offsets[i++];
    for (int k = 0; k < 5; ++k) a.o_dw[k] = h_offsets[i++];
    a.o_gi = h_offsets[i++]; a.o_whh = h_offsets[i++]; a.o_fg = h_offsets[i++];
    for (int k = 0; k < 6; ++k) a.o_dpw[k] = h_offsets[i++];
    for (int k = 0; k < 5; ++k) a.o_ct[k] = h_offsets[i++];
    a.o_last = h_offsets[i++];
    a.o_tg_rz = h_offsets[i++]; a.o_tg_in = h_offsets[i++]; a.o_tg_hn = h_offsets[i++]; a.o_tg_conv = h_offsets[i++];
    const int grid = trunet_stream_fwd_grid(N);
    const size_t ldsb = (size_t)SF_ARENA * sizeof(float);
    if (tg) {
        if (hipFuncSetAttribute((const void*)stream_fwd_x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess)
            return TRUNET_ELAUNCH;
        hipLaunchKernelGGL(stream_fwd_x3_kernel<true>, dim3(grid), dim3(SF_T), ldsb, (hipStream_t)stream, a);
    } else {
        if (hipFuncSetAttribute((const void*)stream_fwd_x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess)
            return TRUNET_ELAUNCH;
        hipLaunchKernelGGL(stream_fwd_x3_kernel<false>, dim3(grid), dim3(SF_T), ldsb, (hipStream_t)stream, a);
    }
    return trunet_launch_status();
}
