// Device helpers of the implicit-GEMM kernels (gemm_conv.hip): segment addressing, the flattened
// stream of K-chunks a persistent workgroup walks, counted vmcnt waits, LDS address-space pointer.
#pragma once
#include "common.hpp"

namespace {


constexpr int NT = TRUNET_TILE_FRAMES;  // frames per tile

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
    // the strides of this network are 1 and 2: shifts instead of an integer division (~25 scalar instructions) on the
    // per-chunk path of every GEMM kernel
    // (the entry points answer TRUNET_ENOTSUP for any other pos_div)
    const int sh = sg.pos_div >> 1;
    r.q = qn >> sh;
    r.valid = (qn >= 0) && ((qn & sh) == 0) && (r.q < sg.L);
    return r;
}

// position in the flattened stream of K-chunks this workgroup walks: tiles in a contiguous range,
// inside a tile the valid segments in order, inside a segment its KC-row chunks
struct ChunkIt {
    int tile, tile_end, p, n0, s, cc, ach, cbase;
    bool valid;
    bool placed;        // p / n0 already hold the tile's position (kept incrementally: no division per tile)
};

template <int KC, int FT = NT>
__device__ __forceinline__ void it_enter_tile(const trunet_gemm_args& a, ChunkIt& it) {
    it.valid = it.tile < it.tile_end;
    if (!it.valid) return;
    if (!it.placed) {           // first tile of the workgroup: one division; afterwards it_next steps p / n0
        const int nt = it.tile / a.P;
        it.p = a.p_begin + (it.tile - nt * a.P);
        it.n0 = nt * FT;
        it.placed = true;
    }
    it.s = 0; it.cc = 0; it.ach = 0; it.cbase = 0;
#ifdef GEMM_SIMPLE_IT      // diagnostic: one always-valid segment (upper bound of what cheaper iterators can give)
    return;
#endif
    while (it.s < a.nseg - 1 && !seg_pos(a.seg[it.s], it.p).valid) {   // host contract: >= 1 valid segment
        it.ach += (a.seg[it.s].nchan + KC - 1) / KC;
        it.cbase += a.seg[it.s].nchan;
        ++it.s;
    }
}

// true when the chunk after `it` belongs to another tile (or the stream ends)
template <int KC, int FT = NT>
__device__ __forceinline__ bool it_next(const trunet_gemm_args& a, ChunkIt& it) {
#ifdef GEMM_SIMPLE_IT
    if (++it.cc < (a.seg[0].nchan + KC - 1) / KC) return false;
    it.cc = 0;
    ++it.tile;
    if (++it.p == a.p_begin + a.P) { it.p = a.p_begin; it.n0 += FT; }
    it.valid = it.tile < it.tile_end;
    return true;
#endif
    const int nck = (a.seg[it.s].nchan + KC - 1) / KC;
    if (++it.cc < nck) return false;
    it.ach += nck;
    it.cbase += a.seg[it.s].nchan;
    it.cc = 0;
    ++it.s;
    while (it.s < a.nseg && !seg_pos(a.seg[it.s], it.p).valid) {
        it.ach += (a.seg[it.s].nchan + KC - 1) / KC;
        it.cbase += a.seg[it.s].nchan;
        ++it.s;
    }
    if (it.s < a.nseg) return false;
    ++it.tile;
    if (++it.p == a.p_begin + a.P) { it.p = a.p_begin; it.n0 += FT; }      // tiles of a workgroup are consecutive, p fastest
    it_enter_tile<KC, FT>(a, it);
    return true;
}

__device__ __forceinline__ void wait_vmcnt(int n) {   // n = LDS-DMA instructions allowed to stay in flight
    switch (n >> 2) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

}  // namespace
