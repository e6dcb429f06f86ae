// Shared device helpers for the gfx950 kernels.  wave = 64 lanes everywhere.
//
// Every translation unit is compiled with -ffp-contract=off: the canonical
// distance sum_d (a_d-b_d)^2 must round each mul and add separately so that
// neighbour indices are bit-exact against oracle/tpgref.c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tpgan_ops.h"

#define TPG_WAVE 64

#define TPG_RETURN_IF_LAUNCH_FAILED()                         \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return TPG_ERR_LAUNCH; \
    } while (0)

typedef unsigned long long tpg_u64;

static inline hipStream_t tpg_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float tpg_sq3(float ax, float ay, float az, float bx, float by,
                                         float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    float s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return s;
}

// ---- DPP cross-lane moves (gfx9 encodings) --------------------------------
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_ror:n = 0x120+n,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143, wave_shr:1 = 0x138.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned tpg_dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ tpg_u64 tpg_dpp_u64(tpg_u64 v) {
    const unsigned lo = tpg_dpp_u32<CTRL, ROW_MASK>((unsigned)v);
    const unsigned hi = tpg_dpp_u32<CTRL, ROW_MASK>((unsigned)(v >> 32));
    return ((tpg_u64)hi << 32) | lo;
}

// lane l receives lane l-1's value (lane 0: unspecified / its own): one DPP move per dword
// (wave_shr:1) instead of a ds_bpermute round trip through the LDS crossbar
__device__ __forceinline__ tpg_u64 tpg_wave_shr1_u64(tpg_u64 v) { return tpg_dpp_u64<0x138>(v); }

__device__ __forceinline__ tpg_u64 tpg_readlane_u64(tpg_u64 v, int lane_uniform) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane_uniform);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane_uniform);
    return ((tpg_u64)hi << 32) | lo;
}

// max over the wave, result broadcast to every lane (wave-uniform value).
__device__ __forceinline__ tpg_u64 tpg_wave_max_u64(tpg_u64 v) {
    tpg_u64 o;
    o = tpg_dpp_u64<0xB1>(v); v = o > v ? o : v;          // xor 1
    o = tpg_dpp_u64<0x4E>(v); v = o > v ? o : v;          // xor 2
    o = tpg_dpp_u64<0x124>(v); v = o > v ? o : v;         // row_ror 4
    o = tpg_dpp_u64<0x128>(v); v = o > v ? o : v;         // row_ror 8 -> row totals
    o = tpg_dpp_u64<0x142, 0xA>(v); v = o > v ? o : v;    // row_bcast 15 into rows 1,3
    o = tpg_dpp_u64<0x143, 0xC>(v); v = o > v ? o : v;    // row_bcast 31 into rows 2,3
    return tpg_readlane_u64(v, 63);
}

__device__ __forceinline__ tpg_u64 tpg_wave_min_u64(tpg_u64 v) {
    return ~tpg_wave_max_u64(~v);
}

// 32-bit wave reductions in the INTEGER domain (no float canonicalisation; the DPP move
// folds into v_max_i32_dpp / v_min_u32_dpp).  Full-row steps use bound_ctrl with old = 0
// (every lane is written), the two row_bcast steps keep the accumulator in unwritten rows.
template <int CTRL>
__device__ __forceinline__ int tpg_dpp_full(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int tpg_dpp_rows(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ int tpg_wave_max_i32(int v) {
    v = max(v, tpg_dpp_full<0xB1>(v));
    v = max(v, tpg_dpp_full<0x4E>(v));
    v = max(v, tpg_dpp_full<0x124>(v));
    v = max(v, tpg_dpp_full<0x128>(v));
    v = max(v, tpg_dpp_rows<0x142, 0xA>(v));
    v = max(v, tpg_dpp_rows<0x143, 0xC>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
// same value; the four row maxima are combined on the scalar unit (4 readlanes + 3 s_max
// instead of two masked row_bcast steps of 3 VALU instructions each)
__device__ __forceinline__ int tpg_wave_max_i32_rows(int v) {
    v = max(v, tpg_dpp_full<0xB1>(v));
    v = max(v, tpg_dpp_full<0x4E>(v));
    v = max(v, tpg_dpp_full<0x124>(v));
    v = max(v, tpg_dpp_full<0x128>(v));
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ unsigned tpg_wave_min_u32(unsigned v) {
    v = min(v, (unsigned)tpg_dpp_full<0xB1>((int)v));
    v = min(v, (unsigned)tpg_dpp_full<0x4E>((int)v));
    v = min(v, (unsigned)tpg_dpp_full<0x124>((int)v));
    v = min(v, (unsigned)tpg_dpp_full<0x128>((int)v));
    v = min(v, (unsigned)tpg_dpp_rows<0x142, 0xA>((int)v));
    v = min(v, (unsigned)tpg_dpp_rows<0x143, 0xC>((int)v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// same, over the first 16 lanes only (one DPP row); result read from lane 0
__device__ __forceinline__ int tpg_row16_max_i32(int v) {
    v = max(v, tpg_dpp_full<0xB1>(v));
    v = max(v, tpg_dpp_full<0x4E>(v));
    v = max(v, tpg_dpp_full<0x124>(v));
    v = max(v, tpg_dpp_full<0x128>(v));
    return __builtin_amdgcn_readlane(v, 0);
}
__device__ __forceinline__ unsigned tpg_row16_min_u32(unsigned v) {
    v = min(v, (unsigned)tpg_dpp_full<0xB1>((int)v));
    v = min(v, (unsigned)tpg_dpp_full<0x4E>((int)v));
    v = min(v, (unsigned)tpg_dpp_full<0x124>((int)v));
    v = min(v, (unsigned)tpg_dpp_full<0x128>((int)v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0);
}

// clamp an index into [0, n) -- invalid indices are undefined behaviour upstream;
// here they must never fault the GPU.
__device__ __forceinline__ int tpg_clamp_idx(int id, int n) {
    const unsigned u = (unsigned)id;
    return (int)(u < (unsigned)n ? u : (unsigned)(n - 1));
}
