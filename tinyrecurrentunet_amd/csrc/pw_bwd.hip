// Fused backward of a pointwise (k = 1) convolution layer:  y[m][p][n] = sum_s sum_c W[m][woff_s + c] a_s[c][q_s(p)][n] + b[m],
// a_s = max(c0 z_s + c1, lo) (BatchNorm+ReLU of the source's raw tensor z_s, or the tensor itself).
//
// One pass over (dy, z_y, z_s) produces everything the layer's backward needs:
//   dz         = ca dy + cb z_y + cc                          (BatchNorm backward of y, in LDS, never written)
//   dW, db     = sum_{p,n} dz a_s^T , sum dz                   (per-workgroup partial images, as conv_wgrad)
//   g_s        = W_s^T dz  [+ previous content] [masked by a_s > 0]   -> gradient w.r.t. the source's raw tensor
//   statistics = sum g_s, sum g_s (z_s - mean_s)               (BatchNorm backward of the source)
// The separate conv_gemm (data gradient) + conv_wgrad launches read dy and z_y twice and z_s twice; this kernel
// reads each once (7 tensor passes -> 4), which is what bounds the layer at 128 channels (measured 3.7 TB/s in
// both of the separate kernels).  "Once" includes the ReLU mask and the z_s of the statistics: the data-gradient waves
// read the staged activation a_s back from the slot in the MFMA C layout (mask = a_s > 0; where a_s > 0,
// z_s - mean = a_s / c0 - (c1 / c0 + mean)) -- until round 3 they loaded z_s from global memory a second time, two to
// three tiles after the DMA, and most of those loads missed the L2 (PMC: 3.07 KB per (position, frame) against 2.56
// algorithmic at 128 channels).  A channel with c0 == 0 (BatchNorm weight exactly zero) takes a slow path.
//
// Tile = one position p x 32 frames of every operand row, delivered by LDS-DMA into a ring of NB slots exactly as
// in conv_wgrad_kernel (rows of 128 B, 16-byte pieces XOR-swizzled through the per-lane SOURCE address; the
// requesting thread applies the prologue in place; all 8 waves share DMA issue and prologue).  The MFMA work is
// split by ROLE, one wave of each role per SIMD, so that the two register-hungry accumulator sets never meet in
// one wave:
//   * waves 0-3 (weight gradient): <= 4 output tiles (32x32) each in accumulators for the whole kernel; the frame
//     axis is the MFMA K axis; a wave's tiles share the dz row tile, so its A fragments are read once.
//   * waves 4-7 (data gradient): the rows of dz are the MFMA K axis.  A wave owns one 32-channel row tile of the
//     sources for the whole kernel (plus, with six row tiles, a second one on alternate tiles), keeps the matching
//     W^T fragments in registers, and runs the epilogue (accumulate / ReLU mask / statistics / store) itself.
// The two roles execute different loops with the same barrier sequence.
//
// X3 (round 4; the default, trunet_gemm_x3_enable & TRUNET_X3_BWD): both GEMMs on v_mfma_f32_32x32x16_bf16 through the
// three-term operand split of x3_common.hpp.  Nothing else changes -- ring, DMA schedule, prologue pass, epilogue --: a lane
// reads the 8 consecutive K-values of its fragment from the fp32 slot (weight gradient: two swizzled 16-byte pieces of a
// row, K = frames; data gradient: 8 ds_read_b32 down the dz rows, K = rows), splits them (~44 vector instructions) and
// issues 6 MFMAs; the data-gradient waves keep W^T as three pre-split fragment planes.  Per (32 x 32 x 16): 6 x 33 matrix
// cycles next to ~45 x 4 vector cycles instead of 8 x 64 matrix cycles -- fp32 MFMA runs at the vector rate on this chip
// and was what bounded these kernels (matrix pipe 67 % busy at 0.55 of the HBM rate).
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "x3_common.hpp"

#ifndef PWB_ABL      // diagnostic builds only (scripts/ubench_pwbwd.py): 1 no epilogue loads, 2 no epilogue stores,
#define PWB_ABL 0    // 4 no prologue pass, 8 no DMA refill, 16 no weight-gradient MFMAs, 32 no data-gradient MFMAs,
#endif               // 64 (X3) operand split replaced by three register copies (wrong results, timing only)

namespace {

#if PWB_ABL & 64
__device__ __forceinline__ void pwb_split8(const f32x4 v0, const f32x4 v1, u32x4& q0, u32x4& q1, u32x4& q2) {
    q0 = __builtin_bit_cast(u32x4, v0); q1 = __builtin_bit_cast(u32x4, v1); q2 = q0 ^ q1;
}
#else
#define pwb_split8 ctx_split8
#endif

// (X3) dz is split ONCE, by the prologue pass that computes it, and stays in LDS as three bf16 planes in the bytes its two
// operands (dy, z_y) occupied: physical piece pc of row r holds [hi of its 4 frames | mid of its 4 frames] (2 x 8 B), the same
// piece of the z row [lo | unused] -- every thread writes inside the 32 bytes it has just read, so the pass stays in place
// and free of cross-thread hazards.  The weight-gradient A fragment (K = frames) is then two 8-byte reads per plane, the
// data-gradient B fragment (K = dz rows) two transposing reads per plane (ds_read_b64_tr_b16: 16 lanes fetch 4 rows x 16
// frames and each receives ITS frame of the 4 rows), neither with any vector work: 40 of the 72 fragment splits per tile are
// gone (the source rows have no spare bytes and are still split by the consuming wave).  0: every fragment split by its consumer.
#ifndef PWB_DZ_PLANES
#define PWB_DZ_PLANES 2          // 1: dz only; 2: the source rows of the 128-row layers too (SRCP below)
#endif

typedef short pwb_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 pwb_tr16(const float* p) {
    const pwb_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) pwb_s16x4*)p);
    return __builtin_bit_cast(u32x2, v);
}

// (SRCP) the fp32 activation at float offset F of a slot (a source row's element: row * 32 + 4 * physical piece + frame & 3)
// from its planes: hi / mid are halfwords (frame & 3) of the piece's first / second 8 bytes, lo of the second half of the same
// piece MA rows further up (the z row); zrow_h = that distance in halfwords (negative).  Exact: hi + mid + lo in fp32.
__device__ __forceinline__ float pwb_act3(const float* S, int F, int c3, int zrow_h) {
    const unsigned short* hp = (const unsigned short*)S + 2 * F - c3;
    const float hi = __builtin_bit_cast(float, (unsigned)hp[0] << 16);
    const float mid = __builtin_bit_cast(float, (unsigned)hp[4] << 16);
    const float lo = __builtin_bit_cast(float, (unsigned)hp[zrow_h + 4] << 16);
    return hi + mid + lo;
}

constexpr int WFC = 32;          // frames per tile
constexpr int PMAXT = 4;         // weight-gradient accumulator tiles per wave (waves 0-3)
constexpr int PSW = 6;           // source DMA row groups per loader wave (sum of channels <= 192)
constexpr int PWB_GRID = TRUNET_NUM_CU;
constexpr int PWB_SHARE = 2;     // partial statistics rows per workgroup

// data-gradient schedule of waves 4..7 (index j = wave - 4): row tiles over the concatenated source channels
struct PwbSched {
    int8_t rt[4], period[4], phase[4], share[4];         // primary row tile: tiles t with (t & (period-1)) == phase
    int8_t rt2[4], period2[4], phase2[4], share2[4];     // secondary row tile (-1: none)
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void pwb_wait_vmcnt(int n) {
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
__device__ __forceinline__ int pwb_off(int r, int pc) { return r * WFC + 4 * (pc ^ ((r >> 1) & 7)); }
// The data-gradient epilogue reads its 32x32 tile of a source back from the slot in the MFMA C layout: lane (h, c), register
// r <-> row rb + 4h + ml(r), ml = (r & 3) + 8 (r >> 2), frame c; rb is a multiple of 16, so the row's swizzle term is
// ((4h + ml) >> 1) & 7 = 2h ^ kr with the compile-time kr = ((r & 3) >> 1) + 4 ((r >> 2) & 1): four per-lane offsets
// e0 ^ 4 kr, e0 = pwb_e0(h, c), next to immediate row offsets.
__device__ __forceinline__ int pwb_e0(int h, int c) { return 4 * ((c >> 2) ^ (2 * h)) + (c & 3); }
__device__ __forceinline__ int pwb_erow(int r) { return ((r & 3) + 8 * (r >> 2)) * WFC; }
__device__ __forceinline__ int pwb_ekr(int r) { return 4 * (((r & 3) >> 1) + 4 * ((r >> 2) & 1)); }

struct PSegPos { bool valid; int q; };
__device__ __forceinline__ PSegPos pwb_seg_pos(const trunet_seg& sg, int p) {
    const int q = p + sg.pos_off;
    PSegPos r;
    r.q = q;
    r.valid = (q >= 0) && (q < sg.L);
    return r;
}

// Global rows as buffer resource (SGPR descriptor of a uniform base) + uniform SGPR row offset + one per-lane VGPR
// offset: 64-bit per-row addresses for 16 rows x 3 tensors would not fit the register budget next to the W^T fragments,
// and -- unlike hand-written asm loads -- the compiler tracks these, so a value is never copied or spilled before it has
// arrived (an earlier inline-asm version produced a wrong 32x32 tile about once in a hundred launches under register
// pressure).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pwb_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
#ifndef TRUNET_PWB_DMA_AUX
#define TRUNET_PWB_DMA_AUX TRUNET_DMA_AUX
#endif
// cache policy of the epilogue rows (each is touched once per launch): TRUNET_PWB_AUX = 2 is nt.  A/B build switch.
#ifndef TRUNET_PWB_AUX
#define TRUNET_PWB_AUX 0
#endif
__device__ __forceinline__ float pwb_bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, TRUNET_PWB_AUX));
}
__device__ __forceinline__ void pwb_bstore(__amdgpu_buffer_rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, TRUNET_PWB_AUX);
}

// wave-uniform values the compiler cannot prove uniform (they pass through per-wave role tables)
__device__ __forceinline__ int pwb_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ size_t pwb_uniform(size_t v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((size_t)hi << 32) | lo;
}
template <typename T>
__device__ __forceinline__ T* pwb_uniform(T* p) { return (T*)pwb_uniform((size_t)p); }

// one data-gradient row tile of a wave: what is fixed for the kernel ...
struct DUnit {
    int seg, ct, cb, flags, mask, phase, share;   // segment, 32-channel tile in it, coefficient base, TRUNET_DG_*
    bool zero;                                    // statistics asked and some channel's BatchNorm weight is exactly zero
};
// ... and what changes with the position p
struct DRun {
    bool valid;
    const float* zb;      // (channel 32*ct, position q, frame 0) of the source's raw tensor
    float* ob;            // same element of the data-gradient output
    size_t dstride;       // elements between consecutive channels
    int voff;             // per-lane byte offset: 4h channels down, frame c
    int lrow;             // this lane's first row (channel 32*ct + 4h) of the source as staged in an LDS slot
};

// AK: k-pairs of the data-gradient A fragments = padded dz rows / 2 (16 for M <= 32, 32 for M <= 64, 64 for M <= 128)
// SEC: data-gradient waves carry a second row tile (six row tiles over four waves); it never has statistics
// KSPLIT (layers with TWO source row tiles, e.g. encoder.1's 64 -> 128): the four data-gradient waves work on EVERY tile,
// wave j on row tile j & 1 and the K half j >> 1 of the dz rows; the two halves of a row tile meet through LDS (each wave
// hands over 8 of its 16 accumulator registers and finishes -- accumulate / mask / statistics / store -- the 8 rows it
// keeps).  Without it two waves alternated whole tiles: the active wave carried AK MFMAs next to its SIMD partner's
// weight-gradient MFMAs while the other pair of SIMDs idled, 96 instead of 64 MFMA times per tile (measured 60 TF against
// 85 TF for the four-row-tile layers).
template <int AK, bool SEC, bool KSPLIT = false, bool X3 = false>
__global__ __launch_bounds__(512, 2) void pw_bwd_kernel(const trunet_pwbwd_args A, const PwbSched sch, const int NB,
                                                        const int rows) {
    const trunet_wgrad_args& a = A.w;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int c = lane & 31;
    constexpr bool DZP = X3 && PWB_DZ_PLANES && !(PWB_ABL & 64);      // dz as bf16 planes (see PWB_DZ_PLANES)
    // ... and the SOURCE rows as well where their lo plane fits into the unused second half of the z pieces (one z row per
    // source row: layers with 128 dz rows and <= 128 source channels -- the encoder): [hi | mid] in the row's own pieces,
    // lo in bytes 8..15 of the same piece of z row (source row index).  The thread that transforms source row s also owns
    // dz row s and has consumed that z piece one part earlier.  Every MFMA fragment of these layers is then read, not split;
    // the epilogue rebuilds the activation it masks with from the three planes (exact).
    constexpr bool SRCP = DZP && AK == 64 && !SEC && PWB_DZ_PLANES >= 2;
    constexpr int MA = 2 * AK;                // padded dz rows
    constexpr int nrt = MA / 32;
    constexpr int DZR = 2 * MA;               // rows of the dz block (dy, z)
    constexpr int PDW = MA / 32;              // dz row groups (8 rows) per loader wave: g = wave + 4 i
    const int SLOT = rows * WFC;              // floats per slot

    float* R_lds = smem;
    f32x4* CA = (f32x4*)(R_lds + (size_t)NB * SLOT);      // [MA] dz coefficients
    f32x4* CB = CA + MA;                                     // [sum nchan] (c0, c1, lo, mean) of the sources
    int ntot_ = 0;
    for (int s = 0; s < a.nseg; ++s) ntot_ += a.seg[s].nchan;
    f32x2* CZ = (f32x2*)(CB + ntot_);                        // [sum nchan] (1/c0, c1/c0 + mean): z - mean = a/c0 - (c1/c0 + mean)
    float* XB = (float*)(CZ + ntot_);                        // KSPLIT: [tile parity][row tile][sender half][8][64] floats

    for (int r = tid; r < MA; r += 512) {
        const int ch = min(r, a.M - 1) + a.a_m_off;
        f32x4 k = {a.ac0[ch], a.ac1[ch], a.ac2[ch], 0.f};
        if (r >= a.M) { k[0] = 0.f; k[1] = 0.f; k[2] = 0.f; }
        CA[r] = k;
    }
    {
        int base = 0;
        for (int s = 0; s < a.nseg; ++s) {
            const trunet_seg& sg = a.seg[s];
            const trunet_dgrad_out& dg = A.dg[s];
            for (int ci = tid; ci < sg.nchan; ci += 512) {
                const bool on = sg.mode == TRUNET_PRO_BNRELU;
                f32x4 k = {on ? sg.c0[ci] : 1.f, on ? sg.c1[ci] : 0.f, on ? 0.f : -3.0e38f,
                           ((dg.flags & TRUNET_DG_STATS) && dg.e2) ? dg.e2[ci] : 0.f};
                CB[base + ci] = k;
                // a BatchNorm weight of exactly zero: a = c1 does not carry z; the epilogue then reads z for that channel.
                // Precision of z - mean = a / c0 - (c1 / c0 + mean): the rounding of a (eps |a|) divided by c0, i.e.
                // eps (|z| + |c1 / c0|) -- |beta / gamma| times the rounding z itself carries; channels whose offset is
                // more than 2^16 times their scale take the slow path as well
                const float ic0 = (fabsf(k[0]) >= 1e-30f && fabsf(k[1]) <= 65536.f * fabsf(k[0])) ? 1.f / k[0] : 0.f;
                f32x2 kz = {ic0, fmaf(k[1], ic0, k[3])};
                CZ[base + ci] = kz;
            }
            base += sg.nchan;
        }
    }
    const int nfc = a.NP / WFC;
    const int total_tiles = a.P * nfc;
    const int t_begin = (int)(((long long)blockIdx.x * total_tiles) / gridDim.x);
    const int t_end = (int)(((long long)(blockIdx.x + 1) * total_tiles) / gridDim.x);
    __syncthreads();

    // Both roles walk the same runs of tiles (equal position p) with the same barrier sequence:
    //   [run prologue] B  { [tile t: everything that reads LDS slot t]  B  [register / global work] } ...  B
    if (wave < 4) {
        // =================== loader + weight-gradient role
        const int Gs = (rows - DZR) / 8;
        const int SPW = (Gs - wave + 3) / 4;      // source row groups of this wave (<= PSW)
        const int LPW = 2 * PDW + SPW;            // DMA instructions per tile
        // output tiles g = wave + 4 i -> (row tile g % nrt = wave % nrt, k tile g / nrt)
        const int my_rt = wave % nrt;
        int t_seg[PMAXT], t_ct[PMAXT];
#pragma unroll
        for (int i = 0; i < PMAXT; ++i) {
            int kt = (wave + 4 * i) / nrt;
            t_seg[i] = -1; t_ct[i] = 0;
#pragma unroll
            for (int s = 0; s < TRUNET_MAX_SEG; ++s) {
                if (s < a.nseg && t_seg[i] < 0 && kt >= 0) {
                    const int nt = (a.seg[s].nchan + 31) / 32;
                    if (kt < nt) { t_seg[i] = s; t_ct[i] = kt; }
                    else kt -= nt;
                }
            }
        }
        f32x16 acc[PMAXT];
#pragma unroll
        for (int i = 0; i < PMAXT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float bsum[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bsum[i] = 0.f;

        const int pc = lane & 7;
        int t0 = t_begin;
        while (t0 < t_end) {
            const int pi = t0 / nfc;
            const int p = a.p_begin + pi;
            const int t1 = min(t_end, (pi + 1) * nfc);       // tiles [t0, t1) share p; frame chunk = t - pi*nfc
            // staged segment list: the valid segments in order, the last one repeated up to nseg entries
            int sl[TRUNET_MAX_SEG], sq[TRUNET_MAX_SEG], sbase[TRUNET_MAX_SEG];
            int nvalid = 0;
            {
                int lasts = 0, lastq = 0;
#pragma unroll
                for (int s = 0; s < TRUNET_MAX_SEG; ++s) { sl[s] = 0; sq[s] = 0; sbase[s] = -1; }
#pragma unroll
                for (int s = 0; s < TRUNET_MAX_SEG; ++s) {
                    if (s < a.nseg) {
                        const PSegPos sp = pwb_seg_pos(a.seg[s], p);
                        if (sp.valid) {
#pragma unroll
                            for (int j = 0; j < TRUNET_MAX_SEG; ++j) if (j == nvalid) { sl[j] = s; sq[j] = sp.q; }
                            lasts = s; lastq = sp.q; ++nvalid;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < TRUNET_MAX_SEG; ++j) if (j >= nvalid && j < a.nseg) { sl[j] = lasts; sq[j] = lastq; }
                int rb = DZR;
#pragma unroll
                for (int j = 0; j < TRUNET_MAX_SEG; ++j) {
                    if (j < nvalid) {
#pragma unroll
                        for (int s = 0; s < TRUNET_MAX_SEG; ++s) if (sl[j] == s) sbase[s] = rb;
                    }
                    if (j < a.nseg) rb += (a.seg[sl[j]].nchan + 31) & ~31;
                }
            }
            int rb_run[PMAXT];       // first LDS row of tile i's source rows in this run, -1: segment not valid at p
#pragma unroll
            for (int i = 0; i < PMAXT; ++i) {
                int sb = -1;
#pragma unroll
                for (int s = 0; s < TRUNET_MAX_SEG; ++s) if (t_seg[i] == s) sb = sbase[s];
                rb_run[i] = sb >= 0 ? sb + t_ct[i] * 32 : -1;
            }
            // per-lane source pointers (frame 0 of this lane's row, logical piece folded in) and coefficient rows
            const float* pd[4];       // dy rows  (PDW <= 4 used; a dependent bound here makes hipcc drop the host stub)
            const float* pzr[4];      // z rows
            const float* ps[PSW];     // source rows
            int cidx[PSW];
#pragma unroll
            for (int i = 0; i < PDW; ++i) {
                const int r = 8 * (wave + 4 * i) + (lane >> 3);
                const int lc = pc ^ ((r >> 1) & 7);
                const int m = min(r, a.M - 1) + a.a_m_off;
                const size_t off = ((size_t)m * a.a_L + p + a.a_pos_off) * a.NP + 4 * lc;
                pd[i] = a.a0 + off;
                pzr[i] = a.a1 + off;
            }
#pragma unroll
            for (int i = 0; i < PSW; ++i) {
                ps[i] = a.a0; cidx[i] = 0;
                if (i < SPW) {
                    const int g = wave + 4 * i;
                    int rb = 0, sidx = sl[0], q = sq[0], ch0 = 0;
#pragma unroll
                    for (int j = 0; j < TRUNET_MAX_SEG; ++j) {
                        if (j < a.nseg) {
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

            auto issue_dma = [&](int t, int slot) __attribute__((always_inline)) {
                const int n0 = (t - pi * nfc) * WFC;
                float* dst = R_lds + (size_t)slot * SLOT;
#pragma unroll
                for (int i = 0; i < PDW; ++i) {
                    const int g = wave + 4 * i;
                    __builtin_amdgcn_global_load_lds(pd[i] + n0, (lds_ptr_t)(dst + g * 256), 16, 0, TRUNET_PWB_DMA_AUX);
                    __builtin_amdgcn_global_load_lds(pzr[i] + n0, (lds_ptr_t)(dst + (MA / 8 + g) * 256), 16, 0, TRUNET_PWB_DMA_AUX);
                }
#pragma unroll
                for (int i = 0; i < PSW; ++i) {
                    if (i < SPW) {
                        const int g = wave + 4 * i;
                        __builtin_amdgcn_global_load_lds(ps[i] + n0, (lds_ptr_t)(dst + (DZR / 8 + g) * 256), 16, 0, TRUNET_PWB_DMA_AUX);
                    }
                }
            };
            // in-place prologue of one 8-row group of tile t (this thread's own DMA pieces): part < PDW: dz rows
            // (BatchNorm backward, zero frames >= N, bias sums); part >= PDW: source rows (BatchNorm + ReLU)
            auto tpart = [&](int part, int t, int slot) __attribute__((always_inline)) {
                const int n0 = (t - pi * nfc) * WFC;
                float* dst = R_lds + (size_t)slot * SLOT;
                if (part < PDW) {
#pragma unroll
                    for (int i = 0; i < PDW; ++i) {
                        if (i == part) {
                            const int r = 8 * (wave + 4 * i) + (lane >> 3);
                            const int lc = pc ^ ((r >> 1) & 7);
                            float* pz = dst + r * WFC + 4 * pc;
                            f32x4 v = *(f32x4*)pz;
                            const f32x4 k = CA[r];
                            const f32x4 z = *(const f32x4*)(pz + MA * WFC);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaf(k[0], v[e], fmaf(k[1], z[e], k[2]));
                            float sacc = 0.f;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (n0 + 4 * lc + e >= a.N) v[e] = 0.f;
                                sacc += v[e];
                            }
                            bsum[i] += sacc;
                            if constexpr (DZP) {
                                unsigned h0, m0, l0, h1, m1, l1;
                                ctx_split2(v[0], v[1], h0, m0, l0);
                                ctx_split2(v[2], v[3], h1, m1, l1);
                                *(u32x4*)pz = u32x4{h0, h1, m0, m1};                    // [hi x 4 frames | mid x 4 frames]
                                *(u32x2*)(pz + MA * WFC) = u32x2{l0, l1};               // lo, in the z row's piece
                            } else {
                                *(f32x4*)pz = v;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < PSW; ++i) {
                        if (i == part - PDW && i < SPW) {
                            const f32x4 k = CB[cidx[i]];
                            float* pz = dst + (DZR + 8 * (wave + 4 * i) + (lane >> 3)) * WFC + 4 * pc;
                            f32x4 v = *(f32x4*)pz;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], k[0], k[1]), k[2]);
                            if constexpr (SRCP) {
                                unsigned h0, m0, l0, h1, m1, l1;
                                ctx_split2(v[0], v[1], h0, m0, l0);
                                ctx_split2(v[2], v[3], h1, m1, l1);
                                *(u32x4*)pz = u32x4{h0, h1, m0, m1};
                                *(u32x2*)(pz - MA * WFC + 2) = u32x2{l0, l1};           // z row (source row index), bytes 8..15
                            } else {
                                *(f32x4*)pz = v;
                            }
                        }
                    }
                }
            };

            // ---- run prologue: NB tiles in flight, first one transformed
            for (int d = 0; d < NB; ++d) issue_dma(min(t0 + d, t1 - 1), d);
            pwb_wait_vmcnt((NB - 1) * LPW);
            if (!(PWB_ABL & 4)) {
#pragma unroll
                for (int part = 0; part < PDW + PSW; ++part) tpart(part, t0, 0);
            }
            // the prologue pass' LDS writes must have completed before another wave reads them: a raw s_barrier
            // does not wait for them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int slot = 0;
            for (int t = t0; t < t1; ++t) {
                const float* S = R_lds + (size_t)slot * SLOT;
                const int nslot = (slot + 1 == NB) ? 0 : slot + 1;
                const bool more = t + 1 < t1;
                const int ra = my_rt * 32 + c;
                // four frame groups of MFMAs; the prologue pass of tile t+1 runs in their shadow, a few row groups
                // after each
                if constexpr (X3) {
#pragma unroll
                    for (int qq = 0; qq < WFC / 16; ++qq) {
                        // this lane's 8 frames of the K-step: 16 qq + 8 h .. + 7 = pieces 4 qq + 2 h, 4 qq + 2 h + 1 of its row
                        u32x4 a0, a1, a2;
                        if constexpr (DZP) {
                            // planes of the dz row: [hi | mid] in the row's own pieces, lo in the z row's
                            const float* p0 = S + pwb_off(ra, 4 * qq + 2 * h);
                            const float* p1 = S + pwb_off(ra, 4 * qq + 2 * h + 1);
                            const u32x4 q0 = *(const u32x4*)p0, q1 = *(const u32x4*)p1;
                            const u32x2 l0 = *(const u32x2*)(p0 + MA * WFC), l1 = *(const u32x2*)(p1 + MA * WFC);
                            a0 = u32x4{q0[0], q0[1], q1[0], q1[1]};
                            a1 = u32x4{q0[2], q0[3], q1[2], q1[3]};
                            a2 = u32x4{l0[0], l0[1], l1[0], l1[1]};
                        } else {
                            pwb_split8(*(const f32x4*)(S + pwb_off(ra, 4 * qq + 2 * h)),
                                       *(const f32x4*)(S + pwb_off(ra, 4 * qq + 2 * h + 1)), a0, a1, a2);
                        }
#pragma unroll
                        for (int i = 0; i < PMAXT; ++i) {
                            if (rb_run[i] >= 0 && !(PWB_ABL & 16)) {
                                u32x4 b0, b1, b2;
                                if constexpr (SRCP) {
                                    const float* p0 = S + pwb_off(rb_run[i] + c, 4 * qq + 2 * h);
                                    const float* p1 = S + pwb_off(rb_run[i] + c, 4 * qq + 2 * h + 1);
                                    const u32x4 q0 = *(const u32x4*)p0, q1 = *(const u32x4*)p1;
                                    const u32x2 l0 = *(const u32x2*)(p0 - MA * WFC + 2), l1 = *(const u32x2*)(p1 - MA * WFC + 2);
                                    b0 = u32x4{q0[0], q0[1], q1[0], q1[1]};
                                    b1 = u32x4{q0[2], q0[3], q1[2], q1[3]};
                                    b2 = u32x4{l0[0], l0[1], l1[0], l1[1]};
                                } else {
                                    pwb_split8(*(const f32x4*)(S + pwb_off(rb_run[i] + c, 4 * qq + 2 * h)),
                                               *(const f32x4*)(S + pwb_off(rb_run[i] + c, 4 * qq + 2 * h + 1)), b0, b1, b2);
                                }
                                CTX_MF6(acc[i], a0, a1, a2, b0, b1, b2);
                            }
                        }
                        if (more && !(PWB_ABL & 4)) {
                            if (qq == 0) pwb_wait_vmcnt((NB - 2) * LPW);      // tile t+1 has landed
#pragma unroll
                            for (int q = 2 * qq; q < 2 * qq + 2; ++q) {
                                tpart(q, t + 1, nslot);
                                tpart(q + 4, t + 1, nslot);
                                if (PDW + PSW > 8) tpart(q + 8, t + 1, nslot);
                            }
                        }
                    }
                } else {
#pragma unroll
                for (int q = 0; q < WFC / 8; ++q) {
                    const f32x4 av = *(const f32x4*)(S + pwb_off(ra, 2 * q + h));
#pragma unroll
                    for (int i = 0; i < PMAXT; ++i) {
                        if (rb_run[i] >= 0 && !(PWB_ABL & 16)) {
                            const f32x4 bv = *(const f32x4*)(S + pwb_off(rb_run[i] + c, 2 * q + h));
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[i], 0, 0, 0);
                        }
                    }
                    if (more && !(PWB_ABL & 4)) {
                        if (q == 0) pwb_wait_vmcnt((NB - 2) * LPW);      // tile t+1 has landed
                        tpart(q, t + 1, nslot);
                        tpart(q + 4, t + 1, nslot);
                        if (PDW + PSW > 8) tpart(q + 8, t + 1, nslot);
                    }
                }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // tile t+1's prologue writes have landed
                __builtin_amdgcn_s_barrier();                // every LDS read of tile t is done
                asm volatile("" ::: "memory");
                if (!(PWB_ABL & 8)) issue_dma(min(t + NB, t1 - 1), slot);        // refill the slot just released
                slot = nslot;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // ring is reused by the next run of tiles
            asm volatile("" ::: "memory");
            t0 = t1;
        }
        // this workgroup's partial image of dW, db
        float* img = a.w_partials + (size_t)blockIdx.x * a.w_numel;
#pragma unroll
        for (int i = 0; i < PMAXT; ++i) {
            if (t_seg[i] >= 0) {
                const trunet_seg& sg = a.seg[t_seg[i]];
                const int ci = t_ct[i] * 32 + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = my_rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < a.M && ci < sg.nchan)
                        img[(size_t)(m + a.w_m_off) * a.ldw_m + (size_t)ci * a.ldw_c + sg.woff] = acc[i][r];
                }
            }
        }
        if (a.b_partials) {
#pragma unroll
            for (int i = 0; i < PDW; ++i) {
                float v = bsum[i];
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 1);
                const int row = 8 * (wave + 4 * i) + (lane >> 3);
                if ((lane & 7) == 0 && row < a.M)
                    a.b_partials[(size_t)blockIdx.x * a.b_stride + a.b_off + row] = v;
            }
        }
    } else {
        // =================== data-gradient role (no DMA, no prologue pass: LDS reads, MFMAs, epilogue)
        // (s_setprio 1 for these waves was measured: 4 % slower at 128 channels, neutral at 64)
        const int j = wave - 4;
        const int e0 = pwb_e0(h, c);
        // (DZP) this lane's two transposing reads of a K-step of 16 dz rows: lane (G = lane >> 4, i = lane & 15) fetches the
        // 4 frames of logical piece 4 (G & 1) + (i & 3) of row 8 (G >> 1) + 4 t + (i >> 2) (t = 0, 1) and receives frame
        // 16 (G & 1) + i = c of the rows 8 h + 4 t .. + 3: float offsets inside the slot, K-step 0
        const int trr = 8 * (lane >> 5) + ((lane & 15) >> 2), trp = 4 * ((lane >> 4) & 1) + (lane & 3);
        const int trb0 = pwb_off(trr, trp), trb1 = pwb_off(trr + 4, trp);
        if constexpr (KSPLIT) {
            constexpr int AH = AK / 2;                 // k-pairs per wave
            const int rt = j & 1, kh = j >> 1;         // row tile, K half (uniform)
            // the row tile's segment / 32-channel tile / coefficient base
            int useg = -1, uct = 0, ucb = 0;
            {
                int g = rt, cb = 0;
#pragma unroll
                for (int s = 0; s < TRUNET_MAX_SEG; ++s) {
                    if (s < a.nseg && g >= 0 && useg < 0) {
                        const int nt = (a.seg[s].nchan + 31) / 32;
                        if (g < nt) { useg = s; uct = g; ucb = cb; }
                        else { g -= nt; cb += a.seg[s].nchan; }
                    }
                }
            }
            useg = pwb_uniform(useg); uct = pwb_uniform(uct); ucb = pwb_uniform(ucb);
            const int uflags = pwb_uniform(A.dg[useg].flags);
            const trunet_seg& usg = a.seg[useg];
            // some channel of this row tile has a BatchNorm weight of exactly zero (statistics: slow path below)
            const bool uzero = (uflags & TRUNET_DG_STATS) && __builtin_amdgcn_ballot_w64(CZ[ucb + 32 * uct + c][0] == 0.f) != 0;
            float af[X3 ? 1 : AH];
            u32x4 ap[X3 ? AH / 8 : 1][3];      // X3: W^T fragment planes, lane (row ch, k = dz row 2 kh AH + 16 ks + 8 h + j)
            {
                const int ch = 32 * uct + c;
                const bool chok = ch < usg.nchan;
                const float* wp = A.W + (size_t)a.w_m_off * a.ldw_m + (size_t)min(ch, usg.nchan - 1) * a.ldw_c + usg.woff;
                if constexpr (X3) {
#pragma unroll
                    for (int ks = 0; ks < AH / 8; ++ks) {
                        f32x4 w0, w1;
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            const int m = 2 * kh * AH + 16 * ks + 8 * h + jj;
                            const float v = wp[(size_t)min(m, a.M - 1) * a.ldw_m];
                            const float x = (chok && m < a.M) ? v : 0.f;
                            if (jj < 4) w0[jj] = x; else w1[jj - 4] = x;
                        }
                        pwb_split8(w0, w1, ap[ks][0], ap[ks][1], ap[ks][2]);
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < AH; ++kk) {
                        const int m = 2 * (kh * AH + kk) + h;
                        const float v = wp[(size_t)min(m, a.M - 1) * a.ldw_m];
                        af[kk] = (chok && m < a.M) ? v : 0.f;
                    }
                }
            }
            float sa1[8], sa2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { sa1[i] = 0.f; sa2[i] = 0.f; }
            const int rowsel = 16 * kh;               // this wave finishes rows (i & 3) + 8 (i >> 2) + 16 kh + 4 h of the tile
            const f32x2* CZs = CZ + ucb + 32 * uct + 4 * h + rowsel;
            int t0 = t_begin;
            while (t0 < t_end) {
                const int pi = t0 / nfc;
                const int p = a.p_begin + pi;
                const int t1 = min(t_end, (pi + 1) * nfc);
                const PSegPos sp = pwb_seg_pos(usg, p);
                const bool pvalid = sp.valid && (uflags & TRUNET_DG_STORE);
                int lrow = DZR + 32 * uct + rowsel + 4 * h;        // this lane's first source row in a slot
#pragma unroll
                for (int s = 0; s < TRUNET_MAX_SEG; ++s)
                    if (s < useg && pwb_seg_pos(a.seg[s], p).valid) lrow += (a.seg[s].nchan + 31) & ~31;
                const size_t dstride = pwb_uniform((size_t)usg.L * a.NP);
                const size_t o0 = ((size_t)(32 * uct + rowsel) * usg.L + (sp.valid ? sp.q : 0)) * a.NP;
                const __amdgpu_buffer_rsrc_t rz = pwb_rsrc(pwb_uniform(A.dg[useg].zmask ? A.dg[useg].zmask + o0 : A.dg[useg].out + o0));
                const __amdgpu_buffer_rsrc_t ro = pwb_rsrc(pwb_uniform(A.dg[useg].out + o0));
                const int rowb = (int)(dstride * sizeof(float));
                const int voff = (int)((4 * h * dstride + c) * sizeof(float));
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();                     // tile t0 staged and transformed
                asm volatile("" ::: "memory");
                int slot = 0;
                for (int t = t0; t < t1; ++t) {
                    const float* S = R_lds + (size_t)slot * SLOT;
                    const int n0 = pwb_uniform((t - pi * nfc) * WFC);
                    const int nb = n0 * (int)sizeof(float);
                    float* xs = XB + (size_t)(((t & 1) * 2 + rt) * 2) * 512;       // [sender half][8][64]
                    float zv[8], ov[8];
                    f32x16 dacc;
                    if (pvalid) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { zv[i] = 0.f; ov[i] = 0.f; }
                        if (uflags & TRUNET_DG_MASK) {        // the source's activation a = max(c0 z + c1, lo) as staged in the slot
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                if constexpr (SRCP) zv[i] = pwb_act3(S, lrow * WFC + pwb_erow(i) + (e0 ^ pwb_ekr(i)), c & 3, -MA * WFC * 2);
                                else zv[i] = S[lrow * WFC + pwb_erow(i) + (e0 ^ pwb_ekr(i))];
                            }
                        }
                        if ((uflags & TRUNET_DG_ACCUM) && !(PWB_ABL & 1)) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) ov[i] = pwb_bload(ro, voff, ((i & 3) + 8 * (i >> 2)) * rowb + nb);
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
                        const int cpc = c >> 2;
                        if constexpr (X3) {
                            // B fragment of K-step ks: frame c, dz rows 2 kh AH + 16 ks + 8 h + j (j < 8); a row keeps its 16-byte
                            // pieces XOR-swizzled by (row >> 1) & 7 = (4 h + (j >> 1)) & 7 (the K-step's first row is a multiple of 16)
                            if constexpr (DZP) {
#pragma unroll
                                for (int ks = 0; ks < ((PWB_ABL & 32) ? 1 : AH / 8); ++ks) {
                                    const float* q0 = S + trb0 + (2 * kh * AH + 16 * ks) * WFC;
                                    const float* q1 = S + trb1 + (2 * kh * AH + 16 * ks) * WFC;
                                    const u32x2 h0 = pwb_tr16(q0), h1 = pwb_tr16(q1), m0 = pwb_tr16(q0 + 2), m1 = pwb_tr16(q1 + 2);
                                    const u32x2 l0 = pwb_tr16(q0 + MA * WFC), l1 = pwb_tr16(q1 + MA * WFC);
                                    const u32x4 b0 = {h0[0], h0[1], h1[0], h1[1]}, b1 = {m0[0], m0[1], m1[0], m1[1]},
                                                b2 = {l0[0], l0[1], l1[0], l1[1]};
                                    CTX_MF6(dacc, ap[ks][0], ap[ks][1], ap[ks][2], b0, b1, b2);
                                }
                            } else {
                            const float* Sb = S + (size_t)(2 * kh * AH + 8 * h) * WFC + (c & 3);
#pragma unroll
                            for (int ks = 0; ks < ((PWB_ABL & 32) ? 1 : AH / 8); ++ks) {
                                f32x4 x0, x1;
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) {
                                    x0[jj] = Sb[(16 * ks + jj) * WFC + 4 * (cpc ^ ((4 * h + (jj >> 1)) & 7))];
                                    x1[jj] = Sb[(16 * ks + 4 + jj) * WFC + 4 * (cpc ^ ((4 * h + 2 + (jj >> 1)) & 7))];
                                }
                                u32x4 b0, b1, b2;
                                pwb_split8(x0, x1, b0, b1, b2);
                                CTX_MF6(dacc, ap[ks][0], ap[ks][1], ap[ks][2], b0, b1, b2);
                            }
                            }
                        } else {
                            const float* Sb = S + (size_t)kh * AH * (2 * WFC) + h * WFC + (c & 3);
#pragma unroll
                            for (int kk = 0; kk < ((PWB_ABL & 32) ? 1 : AH); ++kk) {
                                const float b = Sb[kk * (2 * WFC) + 4 * (cpc ^ (kk & 7))];
                                dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], b, dacc, 0, 0, 0);
                            }
                        }
                        // hand over the 8 registers of the rows the partner wave finishes
#pragma unroll
                        for (int i = 0; i < 8; ++i) xs[(kh * 8 + i) * 64 + lane] = kh ? dacc[i] : dacc[8 + i];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                // every LDS read of tile t is done; the halves are in LDS
                    asm volatile("" ::: "memory");
                    if (pvalid) {
                        const bool fin = n0 + c < a.N;
                        auto rows8 = [&](auto ZGc) __attribute__((always_inline)) {
                            constexpr bool ZG = decltype(ZGc)::value;      // as in the general path below
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int ml = (i & 3) + 8 * (i >> 2);
                                float val = (kh ? dacc[8 + i] : dacc[i]) + xs[((1 - kh) * 8 + i) * 64 + lane];
                                if (uflags & TRUNET_DG_ACCUM) val += ov[i];
                                if (uflags & TRUNET_DG_MASK) val = (zv[i] > 0.f) ? val : 0.f;
                                if (!(PWB_ABL & 2)) pwb_bstore(ro, voff, ml * rowb + nb, val);
                                if (uflags & TRUNET_DG_STATS) {
                                    const float x = fin ? val : 0.f;
                                    const f32x2 kz = CZs[ml];
                                    sa1[i] += x;
                                    float zc = fmaf(zv[i], kz[0], -kz[1]);      // z - mean, z from a = c0 z + c1 (where a > 0)
                                    if (ZG) {
                                        const float z = pwb_bload(rz, voff, ml * rowb + nb);
                                        if (kz[0] == 0.f) zc = z - kz[1];
                                    }
                                    sa2[i] = fmaf(x, zc, sa2[i]);
                                }
                            }
                        };
                        if (uzero) rows8(std::true_type());
                        else rows8(std::false_type());
                    }
                    slot = (slot + 1 == NB) ? 0 : slot + 1;
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();                     // ring is reused by the next run of tiles
                asm volatile("" ::: "memory");
                t0 = t1;
            }
            if (uflags & TRUNET_DG_STATS) {
                const trunet_dgrad_out& dg = A.dg[useg];
                const int nch = usg.nchan;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float s1 = half_wave_sum(sa1[i]);
                    const float s2 = half_wave_sum(sa2[i]);
                    const int ch = 32 * uct + rowsel + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (c == 0 && ch < nch) {
                        float* pp = dg.partials + ((size_t)(blockIdx.x * PWB_SHARE) * nch + ch) * 2;
                        pp[0] = s1;
                        pp[1] = s2;
                    }
                }
            }
            return;
        }
        auto make_unit = [&](int rt, int period, int phase, int share) __attribute__((always_inline)) {
            DUnit u;
            u.seg = -1; u.ct = 0; u.cb = 0; u.flags = 0; u.mask = period - 1; u.phase = phase; u.share = share;
            int g = rt, cb = 0;
#pragma unroll
            for (int s = 0; s < TRUNET_MAX_SEG; ++s) {
                if (s < a.nseg && g >= 0 && u.seg < 0) {
                    const int nt = (a.seg[s].nchan + 31) / 32;
                    if (g < nt) { u.seg = s; u.ct = g; u.cb = cb; }
                    else { g -= nt; cb += a.seg[s].nchan; }
                }
            }
            if (u.seg >= 0) u.flags = A.dg[u.seg].flags;
            u.seg = pwb_uniform(u.seg); u.ct = pwb_uniform(u.ct); u.cb = pwb_uniform(u.cb); u.flags = pwb_uniform(u.flags);
            u.zero = (u.flags & TRUNET_DG_STATS) && __builtin_amdgcn_ballot_w64(CZ[u.cb + 32 * u.ct + c][0] == 0.f) != 0;
            return u;
        };
        // W^T fragments of a row tile: fp32 k-pairs, or (X3) three bf16 planes per K-step of 16 dz rows:
        // lane (row ch = 32 ct + c, k = dz row 16 ks + 8 h + j, j < 8)
        struct Frag { float af[X3 ? 1 : AK]; u32x4 ap[X3 ? AK / 8 : 1][3]; };
        auto load_af = [&](const DUnit& u, Frag& fr) __attribute__((always_inline)) {
            const trunet_seg& sg = a.seg[max(u.seg, 0)];
            const int ch = 32 * u.ct + c;
            const bool chok = u.seg >= 0 && ch < sg.nchan;
            const float* wp = A.W + (size_t)a.w_m_off * a.ldw_m + (size_t)min(ch, sg.nchan - 1) * a.ldw_c + sg.woff;
            if constexpr (X3) {
#pragma unroll
                for (int ks = 0; ks < AK / 8; ++ks) {
                    f32x4 w0, w1;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int m = 16 * ks + 8 * h + jj;
                        const float v = wp[(size_t)min(m, a.M - 1) * a.ldw_m];
                        const float x = (chok && m < a.M) ? v : 0.f;
                        if (jj < 4) w0[jj] = x; else w1[jj - 4] = x;
                    }
                    pwb_split8(w0, w1, fr.ap[ks][0], fr.ap[ks][1], fr.ap[ks][2]);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < AK; ++kk) {
                    const int m = 2 * kk + h;
                    const float v = wp[(size_t)min(m, a.M - 1) * a.ldw_m];     // all loads first, then the selects
                    fr.af[kk] = (chok && m < a.M) ? v : 0.f;
                }
            }
        };
        auto begin_unit = [&](const DUnit& u, int p) __attribute__((always_inline)) {
            DRun r;
            r.valid = false; r.zb = nullptr; r.ob = nullptr; r.dstride = 0; r.voff = 0; r.lrow = 0;
            if (u.seg >= 0 && (u.flags & TRUNET_DG_STORE)) {
                const trunet_seg& sg = a.seg[u.seg];
                const PSegPos sp = pwb_seg_pos(sg, p);
                r.valid = sp.valid;
                int lr = DZR + 32 * u.ct + 4 * h;           // the slot holds the segments valid at p, in order
#pragma unroll
                for (int s = 0; s < TRUNET_MAX_SEG; ++s)
                    if (s < u.seg && pwb_seg_pos(a.seg[s], p).valid) lr += (a.seg[s].nchan + 31) & ~31;
                r.lrow = lr;
                r.dstride = pwb_uniform((size_t)sg.L * a.NP);
                const size_t o = ((size_t)(32 * u.ct) * sg.L + (sp.valid ? sp.q : 0)) * a.NP;
                r.zb = pwb_uniform(A.dg[u.seg].zmask + o);
                r.ob = pwb_uniform(A.dg[u.seg].out + o);
                r.voff = (int)((4 * h * r.dstride + c) * sizeof(float));
            }
            return r;
        };
        // State of the row tile whose epilogue is pending: accumulator and epilogue operands (registers only).
        f32x16 dacc;
        float zv[16], ov[16];
        // head: epilogue operand loads (buffer resource + uniform row offset + per-lane offset), then the MFMAs over the
        // dz rows
        auto dgrad_head = [&](const DUnit& u, const DRun& dr, const Frag& fr, const float* S, int n0)
                              __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t ro = pwb_rsrc(pwb_uniform(dr.ob));
            const int rowb = (int)(pwb_uniform(dr.dstride) * sizeof(float));      // bytes between channels
            const int nb = n0 * (int)sizeof(float);
#pragma unroll
            for (int r = 0; r < 16; ++r) { zv[r] = 0.f; ov[r] = 0.f; }
            if (u.flags & TRUNET_DG_MASK) {       // the source's activation a = max(c0 z + c1, lo) as staged in the slot
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (SRCP) zv[r] = pwb_act3(S, dr.lrow * WFC + pwb_erow(r) + (e0 ^ pwb_ekr(r)), c & 3, -MA * WFC * 2);
                    else zv[r] = S[dr.lrow * WFC + pwb_erow(r) + (e0 ^ pwb_ekr(r))];
                }
            }
            if ((u.flags & TRUNET_DG_ACCUM) && !(PWB_ABL & 1)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ov[r] = pwb_bload(ro, dr.voff, ((r & 3) + 8 * (r >> 2)) * rowb + nb);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
            const int cpc = c >> 2;
            if constexpr (X3) {
                // B fragment of K-step ks: frame c, dz rows 16 ks + 8 h + j (j < 8); swizzle term of a row: (4 h + (j >> 1)) & 7
                if constexpr (DZP) {
#pragma unroll
                    for (int ks = 0; ks < ((PWB_ABL & 32) ? 1 : AK / 8); ++ks) {
                        const float* q0 = S + trb0 + 16 * ks * WFC;
                        const float* q1 = S + trb1 + 16 * ks * WFC;
                        const u32x2 h0 = pwb_tr16(q0), h1 = pwb_tr16(q1), m0 = pwb_tr16(q0 + 2), m1 = pwb_tr16(q1 + 2);
                        const u32x2 l0 = pwb_tr16(q0 + MA * WFC), l1 = pwb_tr16(q1 + MA * WFC);
                        const u32x4 b0 = {h0[0], h0[1], h1[0], h1[1]}, b1 = {m0[0], m0[1], m1[0], m1[1]},
                                    b2 = {l0[0], l0[1], l1[0], l1[1]};
                        CTX_MF6(dacc, fr.ap[ks][0], fr.ap[ks][1], fr.ap[ks][2], b0, b1, b2);
                    }
                } else {
                const float* Sb = S + 8 * h * WFC + (c & 3);
#pragma unroll
                for (int ks = 0; ks < ((PWB_ABL & 32) ? 1 : AK / 8); ++ks) {
                    f32x4 x0, x1;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        x0[jj] = Sb[(16 * ks + jj) * WFC + 4 * (cpc ^ ((4 * h + (jj >> 1)) & 7))];
                        x1[jj] = Sb[(16 * ks + 4 + jj) * WFC + 4 * (cpc ^ ((4 * h + 2 + (jj >> 1)) & 7))];
                    }
                    u32x4 b0, b1, b2;
                    pwb_split8(x0, x1, b0, b1, b2);
                    CTX_MF6(dacc, fr.ap[ks][0], fr.ap[ks][1], fr.ap[ks][2], b0, b1, b2);
                }
                }
            } else {
                const float* Sb = S + h * WFC + (c & 3);
#pragma unroll
                for (int kk = 0; kk < ((PWB_ABL & 32) ? 1 : AK); ++kk) {
                    const float b = Sb[kk * (2 * WFC) + 4 * (cpc ^ (kk & 7))];
                    dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(fr.af[kk], b, dacc, 0, 0, 0);
                }
            }
        };
        // tail: accumulate / ReLU mask / statistics / store; FL = the segment's TRUNET_DG_* flags
        auto dgrad_tail = [&](auto FLc, const DUnit& u, const DRun& dr, float (&st1)[16], float (&st2)[16], int n0)
                              __attribute__((always_inline)) {
            constexpr int FL = decltype(FLc)::value;
            const __amdgpu_buffer_rsrc_t ro = pwb_rsrc(pwb_uniform(dr.ob));
            const int rowb = (int)(pwb_uniform(dr.dstride) * sizeof(float));
            const int nb = n0 * (int)sizeof(float);
            const f32x2* CZs = CZ + u.cb + 32 * u.ct + 4 * h;
            const bool fin = n0 + c < a.N;
            // ZG: z of the statistics read from global memory at its point of use (slow; only when a BatchNorm weight is 0)
            auto rows16 = [&](auto ZGc) __attribute__((always_inline)) {
                constexpr bool ZG = decltype(ZGc)::value;
                const __amdgpu_buffer_rsrc_t rz = pwb_rsrc(pwb_uniform(dr.zb));
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = (r & 3) + 8 * (r >> 2);
                    if ((r & 3) == 0) asm volatile("" ::: "memory");   // coefficient reads: four rows at a time
                    float val = dacc[r];
                    if (FL & TRUNET_DG_ACCUM) val += ov[r];
                    if (FL & TRUNET_DG_MASK) val = (zv[r] > 0.f) ? val : 0.f;
                    if (!(PWB_ABL & 2)) pwb_bstore(ro, dr.voff, ml * rowb + nb, val);
                    if (FL & TRUNET_DG_STATS) {
                        const float x = fin ? val : 0.f;
                        const f32x2 kz = CZs[ml];
                        st1[r] += x;
                        float zc = fmaf(zv[r], kz[0], -kz[1]);            // z - mean, z from a = c0 z + c1 (where a > 0)
                        if (ZG) {
                            const float z = pwb_bload(rz, dr.voff, ml * rowb + nb);
                            if (kz[0] == 0.f) zc = z - kz[1];             // kz[1] = mean for such a channel
                        }
                        st2[r] = fmaf(x, zc, st2[r]);
                    }
                }
            };
            if ((FL & TRUNET_DG_STATS) && u.zero) rows16(std::true_type());
            else rows16(std::false_type());
        };
        float sa1[16], sa2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa1[r] = 0.f; sa2[r] = 0.f; }
        auto tail_primary = [&](const DUnit& u, const DRun& dr, int n0) __attribute__((always_inline)) {
            constexpr int S_ = TRUNET_DG_STORE, M_ = TRUNET_DG_MASK, T_ = TRUNET_DG_STATS, C_ = TRUNET_DG_ACCUM;
            if (u.flags == S_) dgrad_tail(std::integral_constant<int, S_>(), u, dr, sa1, sa2, n0);
            else if (u.flags == (S_ | M_ | T_)) dgrad_tail(std::integral_constant<int, S_ | M_ | T_>(), u, dr, sa1, sa2, n0);
            else if (u.flags == (S_ | M_ | T_ | C_)) dgrad_tail(std::integral_constant<int, S_ | M_ | T_ | C_>(), u, dr, sa1, sa2, n0);
            else dgrad_tail(std::integral_constant<int, S_ | M_ | C_>(), u, dr, sa1, sa2, n0);
        };
        auto tail_secondary = [&](const DUnit& u, const DRun& dr, int n0) __attribute__((always_inline)) {
            constexpr int S_ = TRUNET_DG_STORE, M_ = TRUNET_DG_MASK, C_ = TRUNET_DG_ACCUM;      // never statistics
            if (u.flags == S_) dgrad_tail(std::integral_constant<int, S_>(), u, dr, sa1, sa2, n0);
            else dgrad_tail(std::integral_constant<int, S_ | M_ | C_>(), u, dr, sa1, sa2, n0);
        };
        auto active = [&](const DUnit& u, const DRun& dr, int t) __attribute__((always_inline)) {
            return dr.valid && ((t & u.mask) == u.phase);
        };

        const DUnit u1 = make_unit(sch.rt[j], sch.period[j], sch.phase[j], sch.share[j]);
        const DUnit u2 = make_unit(SEC ? sch.rt2[j] : -1, sch.period2[j], sch.phase2[j], sch.share2[j]);
        Frag af1, af2;                                    // (af2 is dead code without SEC)
        load_af(u1, af1);
        if constexpr (SEC) load_af(u2, af2);

        int t0 = t_begin;
        while (t0 < t_end) {
            const int pi = t0 / nfc;
            const int p = a.p_begin + pi;
            const int t1 = min(t_end, (pi + 1) * nfc);
            const DRun r1 = begin_unit(u1, p);
            DRun r2 = r1;
            if constexpr (SEC) r2 = begin_unit(u2, p);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // tile t0 staged and transformed
            asm volatile("" ::: "memory");
            int slot = 0;
            for (int t = t0; t < t1; ++t) {
                const float* S = R_lds + (size_t)slot * SLOT;
                const int n0 = pwb_uniform((t - pi * nfc) * WFC);
                int pending = 0;          // which unit's epilogue waits for the barrier (0 none, 1, 2)
                if (active(u1, r1, t)) { dgrad_head(u1, r1, af1, S, n0); pending = 1; }
                if constexpr (SEC) {
                    if (active(u2, r2, t)) {
                        if (pending) tail_primary(u1, r1, n0);
                        dgrad_head(u2, r2, af2, S, n0);
                        pending = 2;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                // every LDS read of tile t is done
                asm volatile("" ::: "memory");
                // register-only epilogue: overlaps the other waves' next tile
                if (pending == 1) tail_primary(u1, r1, n0);
                if constexpr (SEC) {
                    if (pending == 2) tail_secondary(u2, r2, n0);
                }
                slot = (slot + 1 == NB) ? 0 : slot + 1;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // ring is reused by the next run of tiles
            asm volatile("" ::: "memory");
            t0 = t1;
        }
        // BatchNorm-backward statistics of the primary row tile's source rows
        if (u1.seg >= 0 && (u1.flags & TRUNET_DG_STATS)) {
            const trunet_dgrad_out& dg = A.dg[u1.seg];
            const int nch = a.seg[u1.seg].nchan;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float s1 = half_wave_sum(sa1[r]);
                const float s2 = half_wave_sum(sa2[r]);
                const int ch = 32 * u1.ct + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (c == 0 && ch < nch) {
                    float* pp = dg.partials + ((size_t)(blockIdx.x * PWB_SHARE + u1.share) * nch + ch) * 2;
                    pp[0] = s1;
                    pp[1] = s2;
                }
            }
        }
    }
}

}  // namespace

extern "C" int trunet_pw_bwd_nparts(void) { return PWB_GRID * PWB_SHARE; }

extern "C" int trunet_pw_bwd(const trunet_pwbwd_args* H, void* stream) {
    if (!H || !H->W) return TRUNET_EINVAL;
    const trunet_wgrad_args* h = &H->w;
    if (!h->a0 || !h->a1 || !h->w_partials || h->nseg < 1 || h->nseg > TRUNET_MAX_SEG) return TRUNET_EINVAL;
    if (h->NP <= 0 || (h->NP % TRUNET_TILE_FRAMES) != 0 || h->N > h->NP || h->P <= 0 || h->M <= 0) return TRUNET_EINVAL;
    if (h->a_mode != TRUNET_PRO_BNBWD || !h->ac0 || !h->ac1 || !h->ac2) return TRUNET_ENOTSUP;
    hipStream_t st = (hipStream_t)stream;
    static const bool thin_valu = [] { const char* e = getenv("TRUNET_PWB_THIN_VALU"); return e && e[0] == '1'; }();
    if (h->M <= 8 && thin_valu) {
        // thin layer (decoder.5's Conv1d(128 -> 8)) on the vector-ALU kernel, same contract: slower than three separate
        // launches (A/B, round 2); the default since round 3 is the MFMA kernel below on a 32-row padded dz block
        for (int s = 0; s < h->nseg; ++s) {
            const trunet_seg& sg = h->seg[s];
            const trunet_dgrad_out& dg = H->dg[s];
            if (!sg.src0 || sg.nchan <= 0 || sg.pos_mul != 1 || sg.pos_div != 1 || sg.mode == TRUNET_PRO_BNBWD) return TRUNET_ENOTSUP;
            if (sg.mode == TRUNET_PRO_BNRELU && (!sg.c0 || !sg.c1)) return TRUNET_EINVAL;
            if (!(dg.flags & TRUNET_DG_STORE) || !dg.out) return TRUNET_ENOTSUP;
            if ((dg.flags & TRUNET_DG_STATS) && (!(dg.flags & TRUNET_DG_MASK) || !dg.partials || !dg.e2)) return TRUNET_EINVAL;
            if ((dg.flags & TRUNET_DG_STATS) && !(dg.flags & TRUNET_DG_PREZERO)) {
                const size_t bytes = (size_t)trunet_pw_bwd_nparts() * sg.nchan * 2 * sizeof(float);
                if (hipMemsetAsync(dg.partials, 0, bytes, st) != hipSuccess) return TRUNET_ELAUNCH;
            }
        }
        return trunet_launch_pw_bwd_small(H, st);
    }
    // M <= 32 (decoder.5: 8 rows): one padded 32-row dz tile.  7/8 of its MFMAs multiply zeros, but the layer moves 34 KB per
    // tile for 32 MFMA times per SIMD -- it is bound by its bytes either way -- and one pass over (dy, z, sources) replaces
    // conv_wgrad + two conv_gemm launches that read the 128 source rows twice
    if (h->M > 128 || (h->M > 32 && (h->M % 32) != 0)) return TRUNET_ENOTSUP;
    const int MA = h->M <= 32 ? 32 : (h->M <= 64 ? 64 : 128);
    int ktiles = 0, ntot = 0;
    for (int s = 0; s < h->nseg; ++s) {
        const trunet_seg& sg = h->seg[s];
        const trunet_dgrad_out& dg = H->dg[s];
        if (!sg.src0 || sg.nchan <= 0 || (sg.nchan % 32) != 0) return TRUNET_ENOTSUP;
        if (sg.pos_mul != 1 || sg.pos_div != 1) return TRUNET_ENOTSUP;
        if (sg.mode == TRUNET_PRO_BNBWD) return TRUNET_ENOTSUP;
        if (sg.mode == TRUNET_PRO_BNRELU && (!sg.c0 || !sg.c1)) return TRUNET_EINVAL;
        if (!(dg.flags & TRUNET_DG_STORE) || !dg.out) return TRUNET_ENOTSUP;
        if ((dg.flags & TRUNET_DG_MASK) && !dg.zmask) return TRUNET_EINVAL;
        // the ReLU mask (and z of the statistics) is taken from the source rows the kernel has staged in LDS anyway
        if ((dg.flags & TRUNET_DG_MASK) && dg.zmask != sg.src0) return TRUNET_ENOTSUP;
        if ((dg.flags & TRUNET_DG_STATS) && (!(dg.flags & TRUNET_DG_MASK) || !dg.partials)) return TRUNET_EINVAL;
        if ((dg.flags & TRUNET_DG_ACCUM) && !(dg.flags & TRUNET_DG_MASK)) return TRUNET_ENOTSUP;
        // the epilogue addresses rows as 32-bit byte offsets from a per-(row tile, position) base
        if ((size_t)sg.L * h->NP * sizeof(float) * 36 >= ((size_t)1 << 31)) return TRUNET_ENOTSUP;
        ktiles += sg.nchan / 32;
        ntot += sg.nchan;
    }
    if (ktiles * (MA / 32) > 4 * PMAXT || ntot > 4 * 8 * PSW) return TRUNET_ENOTSUP;
    if (MA == 32 && ktiles != 4) return TRUNET_ENOTSUP;
    PwbSched sch;
    for (int j = 0; j < 4; ++j) {
        sch.rt[j] = -1; sch.period[j] = 1; sch.phase[j] = 0; sch.share[j] = 0;
        sch.rt2[j] = -1; sch.period2[j] = 1; sch.phase2[j] = 0; sch.share2[j] = 0;
    }
    bool sec = false;
    static const bool ksplit_ok = !(getenv("TRUNET_PWB_KSPLIT") && getenv("TRUNET_PWB_KSPLIT")[0] == '0');
    const bool ksplit = ktiles == 2 && ksplit_ok;      // two row tiles: every wave on every tile, K halves (KSPLIT)
    if (ktiles == 2) {          // (fallback) two waves per row tile, alternating tiles
        for (int j = 0; j < 4; ++j) { sch.rt[j] = j & 1; sch.period[j] = 2; sch.phase[j] = j >> 1; sch.share[j] = j >> 1; }
    } else if (ktiles == 4) {   // one row tile per wave, every tile
        for (int j = 0; j < 4; ++j) sch.rt[j] = j;
    } else if (ktiles == 6) {   // row tiles 0-3 as above; 4, 5 by waves (0,1) on even and (2,3) on odd tiles
        if (MA != 64) return TRUNET_ENOTSUP;
        // row tiles 4, 5 are secondary units: their segment must not ask for statistics
        {
            int kt = 0;
            for (int s = 0; s < h->nseg; ++s) {
                const int nt = h->seg[s].nchan / 32;
                if (kt + nt > 4 && (H->dg[s].flags & TRUNET_DG_STATS)) return TRUNET_ENOTSUP;
                kt += nt;
            }
        }
        sec = true;
        for (int j = 0; j < 4; ++j) {
            sch.rt[j] = j;
            sch.rt2[j] = 4 + (j & 1); sch.period2[j] = 2; sch.phase2[j] = j >> 1; sch.share2[j] = j >> 1;
        }
    } else {
        return TRUNET_ENOTSUP;
    }
    const int rows = 2 * MA + ntot;
    const size_t slot = (size_t)rows * WFC * sizeof(float);
    const size_t fixed = (size_t)(MA + ntot) * sizeof(f32x4) + (size_t)ntot * sizeof(f32x2) + (ksplit ? 2 * 2 * 2 * 8 * 64 * sizeof(float) : 0);
    int NB = (int)((160 * 1024 - fixed) / slot);
    if (NB > 4) NB = 4;
    if (NB < 2) return TRUNET_ENOTSUP;
    const size_t lds = fixed + NB * slot;
    for (int s = 0; s < h->nseg; ++s) {
        const trunet_dgrad_out& dg = H->dg[s];
        if ((dg.flags & TRUNET_DG_STATS) && !(dg.flags & TRUNET_DG_PREZERO)) {
            const size_t bytes = (size_t)PWB_GRID * PWB_SHARE * h->seg[s].nchan * 2 * sizeof(float);
            if (hipMemsetAsync(dg.partials, 0, bytes, st) != hipSuccess) return TRUNET_ELAUNCH;
        }
    }
    trunet_pwbwd_args HK = *H;          // the kernel compares the per-segment flags exactly: host-only bits stay on the host
    for (int s = 0; s < TRUNET_MAX_SEG; ++s) HK.dg[s].flags &= ~TRUNET_DG_PREZERO;
#define PWB_LAUNCH_(AK_, SEC_, KS_, X3_)                                                                                \
    do {                                                                                                               \
        auto kern = pw_bwd_kernel<AK_, SEC_, KS_, X3_>;                                                                   \
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return TRUNET_ELAUNCH;                                                                                     \
        hipLaunchKernelGGL(kern, dim3(PWB_GRID), dim3(512), lds, st, HK, sch, NB, rows);                               \
    } while (0)
#define PWB_LAUNCH(AK_, SEC_, KS_) do { if (x3) PWB_LAUNCH_(AK_, SEC_, KS_, true); else PWB_LAUNCH_(AK_, SEC_, KS_, false); } while (0)
    // both GEMMs on the bf16 MFMA through the three-term operand split (fp32-grade; the default: trunet_hip.h)
    const bool x3 = (trunet_gemm_x3_enable(-1) & (TRUNET_X3_BWD | 4)) != 0;
    if (MA == 32) PWB_LAUNCH(16, false, false);
    else if (MA == 64 && sec) PWB_LAUNCH(32, true, false);
    else if (MA == 64 && ksplit) PWB_LAUNCH(32, false, true);
    else if (MA == 64) PWB_LAUNCH(32, false, false);
    else if (ksplit) PWB_LAUNCH(64, false, true);
    else PWB_LAUNCH(64, false, false);
#undef PWB_LAUNCH_
#undef PWB_LAUNCH
    return trunet_launch_status();
}
