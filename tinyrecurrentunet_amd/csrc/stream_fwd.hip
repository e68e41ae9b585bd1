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

namespace {

constexpr int SF_T = 256;
constexpr int SF_R0 = 0, SF_R1A = 18432, SF_R1B = 18432 + 9216, SF_R2 = 36864, SF_ARENA = 40448;   // floats
constexpr int SF_SKIP = 8192 + 16384 + 8192 + 8192 + 4096;       // enc0..enc4 per workgroup (floats)

__host__ __device__ constexpr int sf_ls(int L) { return (L + 8 + 15) / 16 * 16; }

#ifndef GRU_LIBM
__device__ __forceinline__ float sf_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float sf_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
#else
__device__ __forceinline__ float sf_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float sf_tanh(float x) { return tanhf(x); }
#endif

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
        const f32x4 t = p[i * 64];
        af[4 * i + 0] = t[0]; af[4 * i + 1] = t[1]; af[4 * i + 2] = t[2]; af[4 * i + 3] = t[3];
    }
}

// acc += A(af[0..KP)) * B, B[k = 2 kk + h][j = c] = S[kk * rs2] (S is the lane's base: row h, column of this lane).
// Software-pipelined by hand: the B values of the next 8 k-pairs are requested before the MFMAs of the current 8.
template <int KP>
__device__ __forceinline__ void sf_mm(f32x16& acc, const float* af, const float* S, int rs2) {
    static_assert(KP % 8 == 0, "k-pairs in blocks of 8");
    float b0[8], b1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b0[j] = S[j * rs2];
#pragma unroll
    for (int k0 = 0; k0 < KP; k0 += 16) {
        if (k0 + 8 < KP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) b1[j] = S[(k0 + 8 + j) * rs2];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k0 + j], b0[j], acc, 0, 0, 0);
        if (k0 + 8 < KP) {
            if (k0 + 16 < KP) {
#pragma unroll
                for (int j = 0; j < 8; ++j) b0[j] = S[(k0 + 16 + j) * rs2];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k0 + 8 + j], b1[j], acc, 0, 0, 0);
        }
    }
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
template <int KP1, int KP2, int NRT>
__device__ __forceinline__ void sf_pw(const float* af, float* lds, int src1, int ls1, int coff1, int src2, int ls2, int dst,
                                      int lsd, int P, int M, int row0, bool relu) {
    const int tid_ = sf_tid();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), h = lane >> 5, c = lane & 31;
    const int rt = NRT == 4 ? wave : (NRT == 2 ? (wave & 1) : 0);
    const int ct0 = NRT == 4 ? 0 : (NRT == 2 ? (wave >> 1) : wave);
    const int cts = NRT == 4 ? 1 : (NRT == 2 ? 2 : 4);
    const int nct = (P + 31) >> 5;
    for (int ct = ct0; ct < nct; ct += cts) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        sf_mm<KP1>(acc, af, lds + src1 + h * ls1 + 4 + ct * 32 + c + coff1, 2 * ls1);
        if constexpr (KP2 > 0) sf_mm<KP2>(acc, af + KP1, lds + src2 + h * ls2 + 4 + ct * 32 + c, 2 * ls2);
        const int col = ct * 32 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[r] + af[KP1 + KP2 + r];
            if (relu) v = fmaxf(v, 0.f);
            if (col < P && row < M) lds[dst + (row0 + row) * lsd + 4 + col] = v;
        }
    }
}

// ConvTranspose1d(64 -> 64, k = TAPS, stride S_, padding S_/2) + folded BatchNorm + ReLU.  Output position p = S_ j + e:
// class e uses the taps with (e + pad - tap) % S_ == 0 at source column j + (e + pad - tap) / S_ -- a dense GEMM per tap.
template <int TAPS, int S_>
__device__ __forceinline__ void sf_convT(const float* af, float* lds, int src, int lsi, int dst, int lsd, int Lout) {
    const int tid_ = sf_tid();
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), h = lane >> 5, c = lane & 31;
    const int rt = wave & 1;
    constexpr int PAD = S_ / 2;
    int ucount = 0;
#pragma unroll
    for (int e = 0; e < S_; ++e) {
        const int nj = (Lout - e + S_ - 1) / S_;
        const int tiles = (nj + 31) >> 5;
        for (int jt = 0; jt < tiles; ++jt, ++ucount) {
            if ((ucount & 1) != (wave >> 1)) continue;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                constexpr int BIG = 8 * S_;
                if ((e + PAD - tap + BIG) % S_ == 0) {
                    const int d = (e + PAD - tap + BIG) / S_ - 8;
                    sf_mm<32>(acc, af + tap * 32, lds + src + h * lsi + 4 + jt * 32 + c + d, 2 * lsi);
                }
            }
            const int p = S_ * (jt * 32 + c) + e;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = fmaxf(acc[r] + af[TAPS * 32 + r], 0.f);
                if (p < Lout) lds[dst + row * lsd + 4 + p] = v;
            }
        }
    }
}

// depthwise conv (k, stride s, padding k/2) + folded BatchNorm + ReLU; weights staged in LDS at `wl` ([C][k] then [C])
__device__ __forceinline__ void sf_dw(float* lds, int src, int lsi, int dst, int lsd, int wl, int C, int K, int S, int Lout) {
    for (int o = sf_tid(); o < C * Lout; o += SF_T) {
        const int ch = o / Lout, lo = o - ch * Lout;
        float v = lds[wl + C * K + ch];
        const float* in = lds + src + ch * lsi + 4 + lo * S - K / 2;
        for (int k = 0; k < K; ++k) v = fmaf(lds[wl + ch * K + k], in[k], v);
        lds[dst + ch * lsd + 4 + lo] = fmaxf(v, 0.f);
    }
}

__device__ __forceinline__ void sf_stage(float* lds, int at, const float* g, int n) {
    for (int i = sf_tid(); i < n; i += SF_T) lds[at + i] = g[i];
}

// [C][L] dense (global scratch) <-> LDS buffer rows
__device__ __forceinline__ void sf_save(const float* lds, int buf, int ls, float* g, int C, int L) {
    for (int i = sf_tid(); i < C * L; i += SF_T) {
        const int ch = i / L, p = i - ch * L;
        g[i] = lds[buf + ch * ls + 4 + p];
    }
}
__device__ __forceinline__ void sf_restore(float* lds, int buf, int ls, const float* g, int C, int L) {
    for (int i = sf_tid(); i < C * L; i += SF_T) {
        const int ch = i / L, p = i - ch * L;
        lds[buf + ch * ls + 4 + p] = g[i];
    }
    sf_guards(lds, buf, C, ls, L);
}

struct SfArgs {
    const float* x; float* y; const float* blob; float* scratch;
    int N, Cin;
    int o_first, o_pw[5], o_dw[5], o_gi, o_whh, o_fg, o_dpw[6], o_ct[5], o_last;
};

#define SF_SYNC() __syncthreads()
// request a layer's fragments now; the barrier keeps the compiler from sinking the loads to their first use
#define SF_PREFETCH(NQ, SET, OFF) do { sf_load<NQ>(SET, blob + (OFF) + (size_t)tile_of * ((NQ) * 256), lane); \
                                       __builtin_amdgcn_sched_barrier(0); } while (0)

__global__ __launch_bounds__(SF_T, 1) void stream_fwd_kernel(const SfArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* skip = A.scratch + (size_t)blockIdx.x * SF_SKIP;
    float* sk0 = skip, *sk1 = sk0 + 8192, *sk2 = sk1 + 16384, *sk3 = sk2 + 8192, *sk4 = sk3 + 8192;
    float fa[176], fb[176];                      // two fragment sets (A fragments + 16 bias values of one row tile)
    int tile_of;                                  // this wave's row tile in the layer being prefetched
    const int Cin = A.Cin;

    for (int n = blockIdx.x; n < A.N; n += gridDim.x) {
        // The weights do not change from frame to frame, so the compiler would hoist EVERY layer's fragment loads out
        // of this loop (thousands of registers, all spilled): make the base pointer opaque once per frame.
        const float* blob = A.blob;
        asm volatile("" : "+s"(blob));
        // ---------------- features -> LDS, first conv (C_in -> 64, k5 s2 p1) + ReLU            network.py:9-21
        tile_of = wave;
        SF_PREFETCH(12, fa, A.o_pw[0]);                                     // encoder.1 pw: K = 64 -> 32 k-pairs + bias
        {
            constexpr int LSX = sf_ls(257);
            const float* xg = A.x + (size_t)n * Cin * 257;
            for (int i = tid; i < Cin * LSX; i += SF_T) {
                const int ch = i / LSX, col = i - ch * LSX - 4;
                lds[SF_R2 + i] = (col >= 0 && col < 257) ? xg[ch * 257 + col] : 0.f;
            }
            sf_stage(lds, SF_R2 + 4 * LSX, blob + A.o_first, 64 * Cin * 5 + 64);
            SF_SYNC();
            constexpr int LS = sf_ls(128);
            const int wb = SF_R2 + 4 * LSX;
            for (int o = tid; o < 64 * 128; o += SF_T) {
                const int co = o >> 7, lo = o & 127;
                float v = lds[wb + 64 * Cin * 5 + co];
                for (int ci = 0; ci < Cin; ++ci) {
                    const float* in = lds + SF_R2 + ci * LSX + 4 + 2 * lo - 1;
                    const float* w = lds + wb + (co * Cin + ci) * 5;
#pragma unroll
                    for (int k = 0; k < 5; ++k) v = fmaf(w[k], in[k], v);
                }
                lds[SF_R0 + co * LS + 4 + lo] = fmaxf(v, 0.f);
            }
            sf_guards(lds, SF_R0, 64, LS, 128);
            SF_SYNC();
            sf_save(lds, SF_R0, LS, sk0, 64, 128);
        }
        // ---------------- encoder.1 .. encoder.5 (pointwise + BN + ReLU, depthwise + BN + ReLU)   network.py:24-43
        // i: (K of pw, L in, dw kernel, dw stride, L out)
        {   // encoder.1: 64 -> 128, L 128, dw k3 s1
            constexpr int LS = sf_ls(128);
            sf_stage(lds, SF_R2, blob + A.o_dw[0], 128 * 4);
            sf_pw<32, 0, 4>(fa, lds, SF_R0, LS, 0, 0, 0, SF_R1A, LS, 128, 128, 0, true);
            sf_guards(lds, SF_R1A, 128, LS, 128);
            SF_PREFETCH(20, fb, A.o_pw[1]);
            SF_SYNC();
            sf_dw(lds, SF_R1A, LS, SF_R0, LS, SF_R2, 128, 3, 1, 128);
            sf_guards(lds, SF_R0, 128, LS, 128);
            SF_SYNC();
            sf_save(lds, SF_R0, LS, sk1, 128, 128);
        }
        {   // encoder.2: 128 -> 128, L 128 -> 64, dw k5 s2
            constexpr int LS = sf_ls(128), LSO = sf_ls(64);
            sf_stage(lds, SF_R2, blob + A.o_dw[1], 128 * 6);
            sf_pw<64, 0, 4>(fb, lds, SF_R0, LS, 0, 0, 0, SF_R1A, LS, 128, 128, 0, true);
            sf_guards(lds, SF_R1A, 128, LS, 128);
            SF_PREFETCH(20, fa, A.o_pw[2]);
            SF_SYNC();
            sf_dw(lds, SF_R1A, LS, SF_R0, LSO, SF_R2, 128, 5, 2, 64);
            sf_guards(lds, SF_R0, 128, LSO, 64);
            SF_SYNC();
            sf_save(lds, SF_R0, LSO, sk2, 128, 64);
        }
        {   // encoder.3: L 64, dw k3 s1
            constexpr int LS = sf_ls(64);
            sf_stage(lds, SF_R2, blob + A.o_dw[2], 128 * 4);
            sf_pw<64, 0, 4>(fa, lds, SF_R0, LS, 0, 0, 0, SF_R1A, LS, 64, 128, 0, true);
            sf_guards(lds, SF_R1A, 128, LS, 64);
            SF_PREFETCH(20, fb, A.o_pw[3]);
            SF_SYNC();
            sf_dw(lds, SF_R1A, LS, SF_R0, LS, SF_R2, 128, 3, 1, 64);
            sf_guards(lds, SF_R0, 128, LS, 64);
            SF_SYNC();
            sf_save(lds, SF_R0, LS, sk3, 128, 64);
        }
        {   // encoder.4: L 64 -> 32, dw k5 s2
            constexpr int LS = sf_ls(64), LSO = sf_ls(32);
            sf_stage(lds, SF_R2, blob + A.o_dw[3], 128 * 6);
            sf_pw<64, 0, 4>(fb, lds, SF_R0, LS, 0, 0, 0, SF_R1A, LS, 64, 128, 0, true);
            sf_guards(lds, SF_R1A, 128, LS, 64);
            SF_PREFETCH(20, fa, A.o_pw[4]);
            SF_SYNC();
            sf_dw(lds, SF_R1A, LS, SF_R0, LSO, SF_R2, 128, 5, 2, 32);
            sf_guards(lds, SF_R0, 128, LSO, 32);
            SF_SYNC();
            sf_save(lds, SF_R0, LSO, sk4, 128, 32);
        }
        {   // encoder.5: L 32 -> 16, dw k3 s2
            constexpr int LS = sf_ls(32), LSO = sf_ls(16);
            sf_stage(lds, SF_R2, blob + A.o_dw[4], 128 * 4);
            sf_pw<64, 0, 4>(fa, lds, SF_R0, LS, 0, 0, 0, SF_R1A, LS, 32, 128, 0, true);
            sf_guards(lds, SF_R1A, 128, LS, 32);
            SF_PREFETCH(20, fb, A.o_gi);                                    // GRU projection rows 0..127
            SF_SYNC();
            sf_dw(lds, SF_R1A, LS, SF_R0, LSO, SF_R2, 128, 3, 2, 16);
            sf_guards(lds, SF_R0, 128, LSO, 16);
            SF_SYNC();
        }
        // ---------------- FGRU: input projection (384 x 128), bidirectional recurrence over 16 positions, pw 128 -> 64
        //                                                                                       network.py:45-58,149
        {
            constexpr int LS = sf_ls(16);      // 32
            tile_of = 4 + wave;
            SF_PREFETCH(20, fa, A.o_gi);
            tile_of = wave;
            sf_pw<64, 0, 4>(fb, lds, SF_R0, LS, 0, 0, 0, SF_R1A, LS, 16, 128, 0, false);
            tile_of = 8 + wave;
            SF_PREFETCH(20, fb, A.o_gi);
            sf_pw<64, 0, 4>(fa, lds, SF_R0, LS, 0, 0, 0, SF_R1A, LS, 16, 128, 128, false);
            sf_pw<64, 0, 4>(fb, lds, SF_R0, LS, 0, 0, 0, SF_R1A, LS, 16, 128, 256, false);
            tile_of = wave & 1;
            SF_PREFETCH(20, fa, A.o_fg);                                    // FGRU.conv: 128 -> 64
            SF_SYNC();
            // recurrence: direction d = tid >> 7, thread u owns rows u and (u < 64) 128 + u of W_hh[d] (r | z | n)
            const int d = tid >> 7, u = tid & 127;
            const float* whh = blob + A.o_whh + (size_t)d * 192 * 64;
            const float* bhh = blob + A.o_whh + 2 * 192 * 64 + d * 192;
            float wA[64], wB[64];
            const int rowB = 128 + (u & 63);
#pragma unroll
            for (int k = 0; k < 64; k += 4) {
                const f32x4 ta = *(const f32x4*)(whh + u * 64 + k);
                const f32x4 tb = *(const f32x4*)(whh + rowB * 64 + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) { wA[k + e] = ta[e]; wB[k + e] = tb[e]; }
            }
            const float bA = bhh[u], bB = bhh[rowB];
            float* hs = lds + SF_R2 + d * 64;               // h of this direction
            float* ghs = lds + SF_R2 + 128 + d * 192;       // W_hh h + b_hh
            if (u < 64) hs[u] = 0.f;
            SF_SYNC();
            for (int st = 0; st < 16; ++st) {
                const int pos = d ? 15 - st : st;
                float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
                for (int k = 0; k < 64; k += 4) {
                    const f32x4 hv = *(const f32x4*)(hs + k);
                    a0 = fmaf(wA[k], hv[0], a0); a1 = fmaf(wA[k + 1], hv[1], a1);
                    a0 = fmaf(wA[k + 2], hv[2], a0); a1 = fmaf(wA[k + 3], hv[3], a1);
                    b0 = fmaf(wB[k], hv[0], b0); b1 = fmaf(wB[k + 1], hv[1], b1);
                    b0 = fmaf(wB[k + 2], hv[2], b0); b1 = fmaf(wB[k + 3], hv[3], b1);
                }
                ghs[u] = a0 + a1 + bA;
                if (u < 64) ghs[rowB] = b0 + b1 + bB;
                SF_SYNC();
                if (u < 64) {
                    const float* gi = lds + SF_R1A + (d * 192 + u) * LS + 4 + pos;
                    const float r = sf_sigmoid(gi[0] + ghs[u]);
                    const float z = sf_sigmoid(gi[64 * LS] + ghs[64 + u]);
                    const float nn = sf_tanh(fmaf(r, ghs[128 + u], gi[128 * LS]));
                    const float hn = (1.f - z) * nn + z * hs[u];
                    lds[SF_R0 + (d * 64 + u) * LS + 4 + pos] = hn;
                    hs[u] = hn;
                }
                SF_SYNC();
            }
            sf_guards(lds, SF_R0, 128, LS, 16);
            SF_SYNC();
            sf_pw<64, 0, 2>(fa, lds, SF_R0, LS, 0, 0, 0, SF_R1A, LS, 16, 64, 0, true);
            sf_guards(lds, SF_R1A, 64, LS, 16);
            SF_PREFETCH(12, fb, A.o_dpw[0]);                                // decoder.0 pw: 64 -> 64
            SF_SYNC();
            // ---------------- decoder.0 (FirstTrCNN): pw 64 -> 64, ConvT k3 s2 -> L 31            network.py:60-76
            sf_pw<32, 0, 2>(fb, lds, SF_R1A, LS, 0, 0, 0, SF_R1B, LS, 16, 64, 0, true);
            sf_guards(lds, SF_R1B, 64, LS, 16);
            SF_PREFETCH(28, fa, A.o_ct[0]);
            SF_SYNC();
            sf_restore(lds, SF_R0, sf_ls(32), sk4, 128, 32);
            SF_PREFETCH(28, fb, A.o_dpw[1]);                                // decoder.1 pw: 192 -> 64
            sf_convT<3, 2>(fa, lds, SF_R1B, LS, SF_R1A, sf_ls(31), 31);
            sf_guards(lds, SF_R1A, 64, sf_ls(31), 31);
            SF_SYNC();
        }
        // ---------------- decoder.1 .. decoder.4 (TrCNN): [x1 padded / cropped | skip] -> pw 192 -> 64 -> ConvT
        //                                                                                       network.py:79-100
        {   // decoder.1: x1 L 31 (pad right 1), skip enc4 L 32, ConvT k5 s2 -> 65
            constexpr int LSX = sf_ls(31), LS = sf_ls(32), LSO = sf_ls(65);
            sf_pw<32, 64, 2>(fb, lds, SF_R1A, LSX, 0, SF_R0, LS, SF_R1B, LS, 32, 64, 0, true);
            sf_guards(lds, SF_R1B, 64, LS, 32);
            SF_PREFETCH(44, fa, A.o_ct[1]);
            SF_SYNC();
            sf_restore(lds, SF_R0, sf_ls(64), sk3, 128, 64);
            SF_PREFETCH(28, fb, A.o_dpw[2]);
            sf_convT<5, 2>(fa, lds, SF_R1B, LS, SF_R1A, LSO, 65);
            sf_guards(lds, SF_R1A, 64, LSO, 65);
            SF_SYNC();
        }
        {   // decoder.2: x1 L 65 (crop left 1), skip enc3 L 64, ConvT k3 s1 -> 66
            constexpr int LSX = sf_ls(65), LS = sf_ls(64), LSO = sf_ls(66);
            sf_pw<32, 64, 2>(fb, lds, SF_R1A, LSX, 1, SF_R0, LS, SF_R1B, LS, 64, 64, 0, true);
            sf_guards(lds, SF_R1B, 64, LS, 64);
            SF_PREFETCH(28, fa, A.o_ct[2]);
            SF_SYNC();
            sf_restore(lds, SF_R0, sf_ls(64), sk2, 128, 64);
            SF_PREFETCH(28, fb, A.o_dpw[3]);
            sf_convT<3, 1>(fa, lds, SF_R1B, LS, SF_R1A, LSO, 66);
            sf_guards(lds, SF_R1A, 64, LSO, 66);
            SF_SYNC();
        }
        {   // decoder.3: x1 L 66 (crop 1 each side), skip enc2 L 64, ConvT k5 s2 -> 129
            constexpr int LSX = sf_ls(66), LS = sf_ls(64), LSO = sf_ls(129);
            sf_pw<32, 64, 2>(fb, lds, SF_R1A, LSX, 1, SF_R0, LS, SF_R1B, LS, 64, 64, 0, true);
            sf_guards(lds, SF_R1B, 64, LS, 64);
            SF_PREFETCH(44, fa, A.o_ct[3]);
            SF_SYNC();
            sf_restore(lds, SF_R0, sf_ls(128), sk1, 128, 128);
            SF_PREFETCH(28, fb, A.o_dpw[4]);
            sf_convT<5, 2>(fa, lds, SF_R1B, LS, SF_R1A, LSO, 129);
            sf_guards(lds, SF_R1A, 64, LSO, 129);
            SF_SYNC();
        }
        {   // decoder.4: x1 L 129 (crop left 1), skip enc1 L 128, ConvT k3 s1 -> 130
            constexpr int LSX = sf_ls(129), LS = sf_ls(128), LSO = sf_ls(130);
            sf_pw<32, 64, 2>(fb, lds, SF_R1A, LSX, 1, SF_R0, LS, SF_R1B, LS, 128, 64, 0, true);
            sf_guards(lds, SF_R1B, 64, LS, 128);
            SF_PREFETCH(28, fa, A.o_ct[4]);
            SF_SYNC();
            sf_restore(lds, SF_R0, sf_ls(128), sk0, 64, 128);
            tile_of = 0;
            SF_PREFETCH(20, fb, A.o_dpw[5]);                                // decoder.5 pw: 128 -> 8 (one padded row tile)
            tile_of = wave & 1;
            sf_convT<3, 1>(fa, lds, SF_R1B, LS, SF_R1A, LSO, 130);
            sf_guards(lds, SF_R1A, 64, LSO, 130);
            SF_SYNC();
        }
        {   // ---------------- decoder.5 (LastTrCNN): pw 128 -> 8 (+BN+ReLU), ConvT 8 -> 8 k5 s2 -> 257, linear
            //                                                                                   network.py:102-120
            constexpr int LSX = sf_ls(130), LS = sf_ls(128);
            sf_pw<32, 32, 1>(fb, lds, SF_R1A, LSX, 1, SF_R0, LS, SF_R1B, LS, 128, 8, 0, true);
            sf_guards(lds, SF_R1B, 8, LS, 128);
            sf_stage(lds, SF_R2, blob + A.o_last, 8 * 8 * 5 + 8);        // [ci][co][k], bias
            SF_SYNC();
            float* yg = A.y + (size_t)n * 8 * 257;
            for (int o = tid; o < 8 * 257; o += SF_T) {
                const int co = o / 257, p = o - co * 257;
                float v = lds[SF_R2 + 320 + co];
                // taps with (p + 1 - k) even: k = (p + 1) & 1, +2, +4; source position (p + 1 - k) / 2
                for (int k = (p + 1) & 1; k < 5; k += 2) {
                    const int q = (p + 1 - k) >> 1;              // guards cover q = -1 and q = 128
                    const float* in = lds + SF_R1B + 4 + q;
#pragma unroll
                    for (int ci = 0; ci < 8; ++ci) v = fmaf(lds[SF_R2 + (ci * 8 + co) * 5 + k], in[ci * LS], v);
                }
                yg[o] = v;
            }
            SF_SYNC();
        }
    }
}

}  // namespace

extern "C" size_t trunet_stream_fwd_scratch_floats(int grid) { return (size_t)grid * SF_SKIP; }
extern "C" int trunet_stream_fwd_grid(int N) { return N < 2 * TRUNET_NUM_CU ? (N < TRUNET_NUM_CU ? N : TRUNET_NUM_CU) : TRUNET_NUM_CU; }

extern "C" int trunet_stream_fwd(const float* x, float* y, const float* blob, const int32_t* h_offsets, int n_offsets,
                                 float* scratch, int N, int Cin, void* stream) {
    if (!x || !y || !blob || !h_offsets || !scratch || N <= 0) return TRUNET_EINVAL;
    if (Cin != 3 && Cin != 4) return TRUNET_ENOTSUP;
    if (n_offsets != 26) return TRUNET_EINVAL;
    SfArgs a;
    a.x = x; a.y = y; a.blob = blob; a.scratch = scratch; a.N = N; a.Cin = Cin;
    int i = 0;
    a.o_first = h_offsets[i++];
    for (int k = 0; k < 5; ++k) a.o_pw[k] = h_offsets[i++];
    for (int k = 0; k < 5; ++k) a.o_dw[k] = h_offsets[i++];
    a.o_gi = h_offsets[i++]; a.o_whh = h_offsets[i++]; a.o_fg = h_offsets[i++];
    for (int k = 0; k < 6; ++k) a.o_dpw[k] = h_offsets[i++];
    for (int k = 0; k < 5; ++k) a.o_ct[k] = h_offsets[i++];
    a.o_last = h_offsets[i++];
    for (int k = 0; k < 26; ++k) if (h_offsets[k] < 0 || (h_offsets[k] & 3)) return TRUNET_EINVAL;
    const int grid = trunet_stream_fwd_grid(N);
    const size_t ldsb = (size_t)SF_ARENA * sizeof(float);
    if (hipFuncSetAttribute((const void*)stream_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess)
        return TRUNET_ELAUNCH;
    hipLaunchKernelGGL(stream_fwd_kernel, dim3(grid), dim3(SF_T), ldsb, (hipStream_t)stream, a);
    return trunet_launch_status();
}
