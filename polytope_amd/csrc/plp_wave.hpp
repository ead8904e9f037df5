// plp_wave.hpp -- sub-wavefront "group" primitives for gfx950 (wave64).
//
// A group is GS consecutive lanes (GS in {8,16,32,64}); one LP lives in one group, one
// constraint row per lane, so a wavefront carries 64/GS independent LPs in lockstep.
// Reductions inside a 16-lane DPP row use v_mov_dpp (quad_perm / row_half_mirror / row_mirror),
// which the compiler folds into the consuming VALU op; the 16- and 32-lane hops use
// ds_swizzle / ds_bpermute (LDS crossbar, no LDS memory).
#pragma once
#include "plp_common.hpp"

namespace plp {

struct Grp {
    int lane;       // 0..63 within the wavefront
    int gs;         // group size
    int gbase;      // first lane of my group
    int gl;         // my index inside the group (= my row)
    uint64_t gmask; // gs low bits set
    __device__ Grp(int gs_) {
        lane = threadIdx.x & 63;
        gs = gs_;
        gl = lane & (gs - 1);
        gbase = lane - gl;
        gmask = gs == 64 ? ~0ull : ((1ull << gs) - 1ull);
    }
};

#ifndef PLP_USE_DPP
#define PLP_USE_DPP 1
#endif

// dpp_ctrl encodings (LLVM AMDGPU): quad_perm = sel0|sel1<<2|sel2<<4|sel3<<6,
// row_mirror = 0x140, row_half_mirror = 0x141
#define PLP_DPP_XOR1 0xB1  // quad_perm [1,0,3,2]
#define PLP_DPP_XOR2 0x4E  // quad_perm [2,3,0,1]
#define PLP_DPP_HMIRROR 0x141
#define PLP_DPP_MIRROR 0x140

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = dpp_i<CTRL>(lo);
    hi = dpp_i<CTRL>(hi);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double min_d(double a, double b) { return b < a ? b : a; }
__device__ __forceinline__ unsigned min_u(unsigned a, unsigned b) { return b < a ? b : a; }

// all-reduce(min) over the lanes of my group; every lane of the group gets the result.
__device__ __forceinline__ double grp_min(double v, int gs) {
#if PLP_USE_DPP
    v = min_d(v, dpp_d<PLP_DPP_XOR1>(v));
    v = min_d(v, dpp_d<PLP_DPP_XOR2>(v));
    v = min_d(v, dpp_d<PLP_DPP_HMIRROR>(v));
    if (gs > 8) v = min_d(v, dpp_d<PLP_DPP_MIRROR>(v));
#else
    v = min_d(v, __shfl_xor(v, 1, 64));
    v = min_d(v, __shfl_xor(v, 2, 64));
    v = min_d(v, __shfl_xor(v, 4, 64));
    if (gs > 8) v = min_d(v, __shfl_xor(v, 8, 64));
#endif
    if (gs > 16) v = min_d(v, __shfl_xor(v, 16, 64));
    if (gs > 32) v = min_d(v, __shfl_xor(v, 32, 64));
    return v;
}

// v_min_u32 with a DPP source operand: dst = min(dpp(v), dst).  hipcc does not fold v_mov_dpp
// into the consumer here, so the instruction is written out; the s_nop covers the "VALU write ->
// DPP read" hazard (2 wait states) that the compiler cannot see inside an asm statement.  dst is
// tied to the input, so a lane whose DPP source is disabled keeps its own value (the identity).
#define PLP_MIN_U32_DPP(v, CTRL) \
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(v))

__device__ __forceinline__ unsigned grp_min(unsigned v, int gs) {
#if PLP_USE_DPP
    PLP_MIN_U32_DPP(v, "quad_perm:[1,0,3,2]");
    PLP_MIN_U32_DPP(v, "quad_perm:[2,3,0,1]");
    if (gs > 4) PLP_MIN_U32_DPP(v, "row_half_mirror");
    if (gs > 8) PLP_MIN_U32_DPP(v, "row_mirror");
#else
    v = min_u(v, (unsigned)__shfl_xor((int)v, 1, 64));
    v = min_u(v, (unsigned)__shfl_xor((int)v, 2, 64));
    if (gs > 4) v = min_u(v, (unsigned)__shfl_xor((int)v, 4, 64));
    if (gs > 8) v = min_u(v, (unsigned)__shfl_xor((int)v, 8, 64));
#endif
    if (gs > 16) v = min_u(v, (unsigned)__shfl_xor((int)v, 16, 64));
    if (gs > 32) v = min_u(v, (unsigned)__shfl_xor((int)v, 32, 64));
    return v;
}

// value of lane `src` (absolute lane index; must lie in the caller's own group so that
// divergent groups never read each other's inactive lanes)
__device__ __forceinline__ double bcast(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int bcast(int v, int src) { return __shfl(v, src, 64); }

// same with the byte address (lane << 2) precomputed: one pivot broadcasts NC+2 values from one lane
__device__ __forceinline__ double bcast_addr(double v, int addr) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// ballot restricted to my group, bit i = lane gbase+i
__device__ __forceinline__ uint64_t grp_ballot(bool p, const Grp& g) {
    return (__ballot(p) >> g.gbase) & g.gmask;
}

}  // namespace plp
