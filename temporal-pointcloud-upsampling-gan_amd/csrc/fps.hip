// Furthest point sampling for gfx950.
//
// Replaces pointnet2_utils.furthest_point_sample (reference discriminator.py:114).
//
// FPS is m-1 dependent arg-max rounds per cloud, so the only levers are the
// latency of one round and how many clouds run side by side.  Design:
//   * one workgroup per cloud; every point and its running min-distance live
//     in REGISTERS for the whole kernel (PPT points per thread, compile-time),
//     so a round touches no HBM and no LDS for point data except the 12-byte
//     broadcast read of the last selected point from an LDS copy of the cloud;
//   * the round is branch-free: running distances live as float bits compared
//     as ints, with -1.0f marking points that are never eligible;
//   * arg-max with ties to the smallest index (the canonical rule): a thread
//     owns PPT CONSECUTIVE points, so inside a lane the first strict maximum,
//     inside a wave the lowest lane and among the waves the lowest wave hold
//     the smallest index.  One 32-bit DPP max reduction of the value, then
//     ballot(value == max) -> first set lane -> v_readlane of its index: the
//     index never goes through a reduction of its own.  ONE barrier per
//     round: each wave drops its (value, index) into a double-buffered LDS
//     slot and every wave re-reduces the <=16 slots redundantly, so no second
//     barrier and no broadcast step are needed.  (A round is VALU-issue bound
//     on its one CU: ~80 wave instructions x 16 waves.)
//   * built WITHOUT the SLP vectoriser (build.py): the packed fp32 math it makes of this round
//     gave wrong picks on some hosts under multi-stream load (round 3).
//   * clouds of <=256 points run in a single wave with no barrier at all.
// Callers batch frames x {fake,true} x batch into B so that B workgroups run
// concurrently (the reference issues one launch per frame per cloud batch).
#include <cstdlib>

#include "tpg_common.hpp"

#ifdef TPG_FPS_DEBUG
// diagnostic build (tools/fps_wave_trace.py): every wave of the launch with gridDim.x == dbg_grid records, per round,
// the pick it arrived at and the coordinates it then measures against
__device__ int *tpg_fps_dbg_buf = nullptr;
__device__ int tpg_fps_dbg_grid = 0;      // only the launch with this many clouds ...
__device__ int tpg_fps_dbg_m = 0;         // ... this many picks, and the <256, 4, LDS copy> shape (buffer: grid*m*4*4 ints)
extern "C" int tpg_fps_debug_set(int *buf, int grid, int m) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(tpg_fps_dbg_buf), &buf, sizeof(buf)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(tpg_fps_dbg_grid), &grid, sizeof(grid)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(tpg_fps_dbg_m), &m, sizeof(m)) != hipSuccess) return -1;
    return 0;
}
#endif

namespace {

constexpr int FPS_LDS_POINTS = 12288;  // LDS copy of the cloud: 12 B/point -> 144 KiB max

__device__ __forceinline__ tpg_u64 fps_key(float d2, int k) {
    return ((tpg_u64)(__float_as_uint(d2) + 1u) << 32) | (unsigned)(~k);
}

// FPS OF AN FPS PREFIX (round 3).  Let S = the first M picks of a furthest point sampling of a cloud, in pick order, and
// sample m <= M points of S with the same rule (start at its point 0).  Round j of that second sampling maximises, over S,
// the minimum distance to S[0..j): the very quantity the first sampling maximised over the WHOLE cloud when it chose
// S[j], computed by the same fp32 operations -- so S[j] attains the maximum over S as well, every candidate tied with it
// sits at a later position, and the picks are 0, 1, ..., m-1 as long as that maximum was POSITIVE (with a maximum of 0
// -- fewer distinct eligible points than picks -- ties reach back to already chosen positions).  Every set-abstraction
// level after the first samples exactly such a prefix (discriminator.py:114 on the centres of the level before,
// :131-137), so the producing launch leaves a per-cloud flag "my last maximum was positive" (prefix_out) and the
// consuming launch (prefix_in) writes 0..m-1 and returns when it is set -- same result, no rounds -- and runs the full
// algorithm when it is not.  A launch that took the shortcut hands the flag on.
__device__ __forceinline__ bool tpg_fps_prefix_shortcut(const int32_t *prefix_in, int32_t *prefix_out, const int32_t *start,
                                                        int32_t *out, int m, int N, int tid, int nthreads) {
    if (prefix_in == nullptr || start != nullptr || m > N || prefix_in[blockIdx.x] == 0) return false;   // (block-uniform)
    for (int j = tid; j < m; j += nthreads) out[j] = j;
    if (tid == 0 && prefix_out) prefix_out[blockIdx.x] = 1;
    return true;
}

// Running distances are kept as float BITS compared as signed ints: every distance is >= +0,
// whose bit patterns order like ints, and the single negative value -1.0f marks "not eligible"
// (|x|^2 <= 1e-3, or padding) -- min(d, -1) stays -1 and -1 never beats the initial best = -1,
// which is exactly the reference's `continue`.
template <int BLOCK, int PPT, bool USE_LDS>
__global__ __launch_bounds__(BLOCK) void fps_kernel(const float *__restrict__ xyz, int N, int m,
                                                    int32_t *__restrict__ idx,
                                                    const int32_t *__restrict__ start, int skip_origin,
                                                    const int32_t *__restrict__ prefix_in,
                                                    int32_t *__restrict__ prefix_out) {
    extern __shared__ __attribute__((aligned(16))) float fps_smem[];
    constexpr int NW = BLOCK / 64;
    // layout: [2][16] (value,index) slots (256 B), [2][16] xyz of each wave's winner (768 B: clouds too
    // large for an LDS copy), then a copy of the cloud (USE_LDS): xyz of a point side by side, so the
    // winner's coordinates are one address and three offsets
    int2 *slots = reinterpret_cast<int2 *>(fps_smem);
    float *wxyz = fps_smem + 64;          // [2][16][4]
    float *sp = fps_smem + 64 + 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const float *x = xyz + (size_t)blockIdx.x * N * 3;
    int32_t *out = idx + (size_t)blockIdx.x * m;
    if (tpg_fps_prefix_shortcut(prefix_in, prefix_out, start, out, m, N, (int)threadIdx.x, BLOCK)) return;

    float px[PPT], py[PPT], pz[PPT];
    int tp[PPT];                          // running min distance, as float bits (see above)
#pragma unroll
    for (int t = 0; t < PPT; ++t) {
        const int k = tid * PPT + t;
        const bool in = k < N;
        // clamped, UNCONDITIONAL loads (a load under `in ?` becomes a branch per element: 16 of them made
        // hipcc spill 744 bytes per lane in the variant without the LDS copy -- 6 us per round at 16384 points)
        const float *pk = x + (size_t)(in ? k : N - 1) * 3;
        const float vx = pk[0], vy = pk[1], vz = pk[2];
        px[t] = in ? vx : 0.0f;
        py[t] = in ? vy : 0.0f;
        pz[t] = in ? vz : 0.0f;
        float mag = px[t] * px[t];
        mag = mag + py[t] * py[t];
        mag = mag + pz[t] * pz[t];
        tp[t] = __float_as_int((in && (!skip_origin || mag > 1e-3f)) ? 1e10f : -1.0f);
        if (USE_LDS && in) { sp[3 * k] = px[t]; sp[3 * k + 1] = py[t]; sp[3 * k + 2] = pz[t]; }
    }
    int old = start ? tpg_clamp_idx(start[blockIdx.x], N) : 0;       // pointnet2: always point 0
    if (tid == 0) out[0] = old;
    // the first pick's coordinates (clouds without an LDS copy: from memory, once) and point 0's (the
    // pick when no point is eligible), so that NO global load is left inside the round loop
    float ox = x[(size_t)old * 3], oy = x[(size_t)old * 3 + 1], oz = x[(size_t)old * 3 + 2];
    const float zx = x[0], zy = x[1], zz = x[2];
    if (NW > 1 && tid < 32) slots[tid] = make_int2((int)0x80000000, -1);   // tag 0xffff: no round has it
    if (NW > 1 || USE_LDS) __syncthreads();

    for (int j = 1; j < m; ++j) {
        if (USE_LDS) { ox = sp[3 * old]; oy = sp[3 * old + 1]; oz = sp[3 * old + 2]; }
#ifdef TPG_FPS_DEBUG
        if (BLOCK == 256 && PPT == 4 && USE_LDS && tpg_fps_dbg_buf && (int)gridDim.x == tpg_fps_dbg_grid &&
            m == tpg_fps_dbg_m && lane == 0) {
            int *d = tpg_fps_dbg_buf + (((size_t)blockIdx.x * m + j) * NW + wave) * 4;
            d[0] = old; d[1] = __float_as_int(ox); d[2] = __float_as_int(oy); d[3] = __float_as_int(oz);
        }
#endif

        int best = __float_as_int(-1.0f);
        int besti = 0;
        float bx = 0.0f, by = 0.0f, bz = 0.0f;    // (no LDS copy: the lane's best point travels with its index)
#pragma unroll
        for (int t = 0; t < PPT; ++t) {
            const float d = tpg_sq3(px[t], py[t], pz[t], ox, oy, oz);
            // min on the bits: d >= +0 (NaN bits are a large int: the running value stays, as in
            // `d < tp ? d : tp`), and -1.0f is a negative int that stays the minimum
            const int b2 = min(__float_as_int(d), tp[t]);
            tp[t] = b2;
            const bool up = b2 > best;            // strict: first (smallest) index wins inside a lane
            best = up ? b2 : best;
            besti = up ? tid * PPT + t : besti;
            if constexpr (!USE_LDS) { bx = up ? px[t] : bx; by = up ? py[t] : by; bz = up ? pz[t] : bz; }
        }
        // the value's max, then the index of the FIRST lane that attains it (lower lane <=> lower
        // indices): the mask is never empty, the maximum is somebody's value
        int mx = tpg_wave_max_i32_rows(best);
        const int src = __builtin_ctzll(__ballot(best == mx));
        int bi = __builtin_amdgcn_readlane(besti, src);
        if constexpr (!USE_LDS) {
            // no LDS copy of the cloud (N > 12288): the winning lane hands on its point's coordinates
            // itself -- a round trip to L2 for 12 bytes cost 3 of the 3.5 us of a round at 16384 points
            if (lane == src) {
                float *w = wxyz + ((j & 1) * 16 + wave) * 4;
                w[0] = bx; w[1] = by; w[2] = bz;
            }
        }
        int win = wave;
#ifndef TPG_FPS_POLL_EXCHANGE            // default: one barrier per round, double-buffered slots
        if constexpr (NW > 1) {
            int2 *slot = slots + (j & 1) * 16;
            if (lane == 0) slot[wave] = make_int2(mx, bi);
            __syncthreads();
            const int2 sv = lane < NW ? slot[lane] : make_int2((int)0x80000000, 0);
            mx = tpg_row16_max_i32(sv.x);         // every slot value is >= bits(-1.0f) > INT_MIN
            win = __builtin_ctzll(__ballot(sv.x == mx));
            bi = __builtin_amdgcn_readlane(sv.y, win);
        }
#else
        if constexpr (NW > 1) {
            // Alternative exchange WITHOUT a barrier (-DTPG_FPS_POLL_EXCHANGE; tools/tune_fps.py "default" vs "barrier":
            // 0.1 us per round slower, kept because it needs nothing but LDS ordering): every entry carries its
            // round number, and a wave polls the NW entries until all of them carry THIS round's.  Double buffering
            // stays: a wave writes its round-(j+2) entry only after it has seen every wave's round-(j+1) entry, i.e.
            // after every wave has finished reading round j.  (Written while hunting the wrong picks that turned out
            // to come from SLP-vectorised packed math -- see build.py; both exchanges are correct without it.)
            const unsigned tag = (unsigned)j << 16;                        // j <= 16383 (N <= 16384 here)
            volatile int2 *slot = slots + (j & 1) * 16;
            if (lane == 0) {
                slot[wave].x = mx;
                slot[wave].y = (int)(tag | (unsigned)bi);
            }
            int sx, sy;
            do {
                // (y first: an entry whose tag matches was written x-then-y by one lane, LDS keeps a wave's order)
                sy = lane < NW ? slot[lane].y : (int)tag;
                sx = lane < NW ? slot[lane].x : (int)0x80000000;
            } while (__ballot(((unsigned)sy & 0xffff0000u) != tag) != 0ull);
            mx = tpg_row16_max_i32(sx);           // every slot value is >= bits(-1.0f) > INT_MIN
            win = __builtin_ctzll(__ballot(sx == mx));
            bi = __builtin_amdgcn_readlane(sy, win) & 0xffff;
        }
#endif
        old = mx >= 0 ? bi : 0;                   // mx < 0 <=> no eligible point at all
        if constexpr (!USE_LDS) {
            const float *w = wxyz + ((j & 1) * 16 + win) * 4;         // (single wave: its own LDS writes, in order)
            const float4 wv = *reinterpret_cast<const float4 *>(w);
            ox = mx >= 0 ? wv.x : zx; oy = mx >= 0 ? wv.y : zy; oz = mx >= 0 ? wv.z : zz;
        }
        if (tid == 0) out[j] = old;
        if (j == m - 1 && tid == 0 && prefix_out) prefix_out[blockIdx.x] = mx > 0 ? 1 : 0;   // last (= smallest) maximum
    }
    if (m == 1 && tid == 0 && prefix_out) prefix_out[blockIdx.x] = 1;
}

// ---- pruned rounds (round 3) -------------------------------------------------------------------------------------
// A round of the kernel above measures ALL N points against the new pick although, after a few dozen picks, the pick can
// lower the running distance only of points closer to it than they already are to an earlier pick.  Here the cloud is
// put into Morton order of a 16^3 grid first (a counting sort in LDS, once per launch), so that the 64 points a wave
// holds in register slot t -- a TILE -- sit together in space.  Lane t of a wave keeps tile t's bounding box, the maximum
// of its points' running distances and that maximum's point; a round
//   1. bounds, per tile, the distance from the new pick to the box from below (lanes 0..15, one pass of vector code),
//   2. updates only the tiles whose bound -- less a margin far above fp32 rounding -- is below their current maximum
//      (a tile that is skipped cannot change: every one of its distances to the pick is >= the bound >= its maximum
//      >= each running distance), re-deriving maximum and arg-max of an updated tile by a wave reduction,
//   3. takes the round's winner from the 16 tile records instead of from 64 x 16 running distances.
// Same picks as the dense kernel, bit for bit: a point's distance is the same fp32 expression on the same coordinates,
// a skipped update is the identity, and ties go to the smallest ORIGINAL index (the records carry it; waves and tiles no
// longer hold points in index order).  At 4096 -> 1024 a round touches ~2 of a wave's 16 tiles on average.
constexpr int FPP_GRID_BITS = 4;                      // 16 cells per axis
constexpr int FPP_CELLS = 1 << (3 * FPP_GRID_BITS);   // + 1 bin for ineligible points
constexpr int FPP_LDS_POINTS = 8192;                  // LDS copy of the cloud beside the sort's tables

__device__ __forceinline__ float fpp_wave_min_f32(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fminf(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ float fpp_wave_max_f32(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ unsigned fpp_spread3(unsigned v) {          // 4 bits -> every third bit
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}

template <int BLOCK, bool USE_LDS>
__global__ __launch_bounds__(BLOCK) void fps_pruned_kernel(const float *__restrict__ xyz, int N, int m,
                                                           int32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ start, int skip_origin,
                                                           const int32_t *__restrict__ prefix_in,
                                                           int32_t *__restrict__ prefix_out) {
    extern __shared__ __attribute__((aligned(16))) float fps_smem[];
    constexpr int NW = BLOCK / 64, PPT = 16;
    int2 *slots = reinterpret_cast<int2 *>(fps_smem);               // [2][16] (value, index)
    float *wxyz = fps_smem + 64;                                    // [2][16][4] winner coordinates (no LDS copy)
    float *red = fps_smem + 64 + 128;                               // [16][8] block reductions of the prepass
    int *hist = reinterpret_cast<int *>(fps_smem + 64 + 128 + 128); // [FPP_CELLS + 2] (+ scan scratch [32])
    int *scan = hist + FPP_CELLS + 2;
    int *perm = scan + 32;                                          // [BLOCK * PPT] sorted position -> original index
    float *sp = reinterpret_cast<float *>(perm + BLOCK * PPT);      // [3 N] copy of the cloud (USE_LDS)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *x = xyz + (size_t)blockIdx.x * N * 3;
    int32_t *out = idx + (size_t)blockIdx.x * m;
    if (tpg_fps_prefix_shortcut(prefix_in, prefix_out, start, out, m, N, tid, BLOCK)) return;

    // ---- prepass 1: bounding box of the eligible points, cell keys, counting sort into Morton order
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int k = tid; k < N; k += BLOCK) {
        const float vx = x[(size_t)k * 3], vy = x[(size_t)k * 3 + 1], vz = x[(size_t)k * 3 + 2];
        if (USE_LDS) { sp[3 * k] = vx; sp[3 * k + 1] = vy; sp[3 * k + 2] = vz; }
        float mag = vx * vx;
        mag = mag + vy * vy;
        mag = mag + vz * vz;
        if (!skip_origin || mag > 1e-3f) {
            lo[0] = fminf(lo[0], vx); lo[1] = fminf(lo[1], vy); lo[2] = fminf(lo[2], vz);
            hi[0] = fmaxf(hi[0], vx); hi[1] = fmaxf(hi[1], vy); hi[2] = fmaxf(hi[2], vz);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = fpp_wave_min_f32(lo[a]);
        hi[a] = fpp_wave_max_f32(hi[a]);
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[wave * 8 + a] = lo[a]; red[wave * 8 + 3 + a] = hi[a]; }
    }
    for (int i = tid; i < FPP_CELLS + 2; i += BLOCK) hist[i] = 0;
    __syncthreads();
    float inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red[a], h = red[3 + a];
        for (int w = 1; w < NW; ++w) { l = fminf(l, red[w * 8 + a]); h = fmaxf(h, red[w * 8 + 3 + a]); }
        lo[a] = l;
        const float ext = h - l;
        inv[a] = (ext > 0.0f && ext < 3.0e38f) ? (float)(1 << FPP_GRID_BITS) / (ext * 1.000001f) : 0.0f;
    }
    auto key_of = [&](float vx, float vy, float vz) -> int {
        float mag = vx * vx;
        mag = mag + vy * vy;
        mag = mag + vz * vz;
        if (skip_origin && !(mag > 1e-3f)) return FPP_CELLS;
        // (NaN or out-of-box coordinates fall into cell 0 / the last cell: any placement is correct, only slower)
        const int cmax = (1 << FPP_GRID_BITS) - 1;
        int cx = (int)((vx - lo[0]) * inv[0]), cy = (int)((vy - lo[1]) * inv[1]), cz = (int)((vz - lo[2]) * inv[2]);
        cx = cx < 0 ? 0 : (cx > cmax ? cmax : cx);
        cy = cy < 0 ? 0 : (cy > cmax ? cmax : cy);
        cz = cz < 0 ? 0 : (cz > cmax ? cmax : cz);
        return (int)(fpp_spread3((unsigned)cx) | (fpp_spread3((unsigned)cy) << 1) | (fpp_spread3((unsigned)cz) << 2));
    };
    for (int k = tid; k < N; k += BLOCK)
        atomicAdd(&hist[key_of(x[(size_t)k * 3], x[(size_t)k * 3 + 1], x[(size_t)k * 3 + 2])], 1);
    __syncthreads();
    {   // exclusive scan of hist[0 .. FPP_CELLS] in place
        constexpr int L = FPP_CELLS + 1;
        constexpr int per = (L + BLOCK - 1) / BLOCK;
        const int b0 = tid * per < L ? tid * per : L, b1 = b0 + per < L ? b0 + per : L;
        int local = 0;
        for (int i = b0; i < b1; ++i) local += hist[i];
        int incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) scan[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += scan[w];
        int run = base + incl - local;
        for (int i = b0; i < b1; ++i) {
            const int c = hist[i];
            hist[i] = run;
            run += c;
        }
        __syncthreads();
    }
    for (int k = tid; k < N; k += BLOCK) {
        const int pos = atomicAdd(&hist[key_of(x[(size_t)k * 3], x[(size_t)k * 3 + 1], x[(size_t)k * 3 + 2])], 1);
        perm[pos] = k;              // (order inside a cell = arrival order: irrelevant, ties go by original index)
    }
    __syncthreads();

    // ---- prepass 2: the thread's points (slot t of a wave = 64 consecutive sorted positions = a tile), tile records
    float px[PPT], py[PPT], pz[PPT];
    int tp[PPT], oi[PPT];
    float b0x = 3.0e38f, b0y = 3.0e38f, b0z = 3.0e38f, b1x = -3.0e38f, b1y = -3.0e38f, b1z = -3.0e38f;
    int tmax = __float_as_int(-1.0f), targ = 0x7fffffff;
    float tcx = 0.0f, tcy = 0.0f, tcz = 0.0f;
#pragma unroll
    for (int t = 0; t < PPT; ++t) {
        // tile g = sorted positions [64 g, 64 g + 64) goes to wave g % NW, slot g / NW: tiles that are neighbours in
        // space -- the ones a pick reaches together -- land on different waves (with consecutive tiles in one wave the
        // pick's wave updated ~8 tiles while the others idled at the barrier: 1.29 instead of 0.63 us per round)
        const int pos = ((t * NW + wave) * 64) + lane;
        const bool in = pos < N;
        const int k = in ? perm[pos] : 0;
        const float *pk = x + (size_t)k * 3;
        const float vx = pk[0], vy = pk[1], vz = pk[2];
        px[t] = in ? vx : 0.0f;
        py[t] = in ? vy : 0.0f;
        pz[t] = in ? vz : 0.0f;
        oi[t] = in ? k : 0x7fffffff;
        float mag = px[t] * px[t];
        mag = mag + py[t] * py[t];
        mag = mag + pz[t] * pz[t];
        const bool el = in && (!skip_origin || mag > 1e-3f);
        tp[t] = __float_as_int(el ? 1e10f : -1.0f);
        const float l0 = fpp_wave_min_f32(el ? vx : 3.0e38f), l1 = fpp_wave_min_f32(el ? vy : 3.0e38f),
                    l2 = fpp_wave_min_f32(el ? vz : 3.0e38f);
        const float h0 = fpp_wave_max_f32(el ? vx : -3.0e38f), h1 = fpp_wave_max_f32(el ? vy : -3.0e38f),
                    h2 = fpp_wave_max_f32(el ? vz : -3.0e38f);
        const int any = __ballot(el) != 0ull;
        if (lane == t) {
            b0x = l0; b0y = l1; b0z = l2; b1x = h0; b1y = h1; b1z = h2;
            tmax = __float_as_int(any ? 1e10f : -1.0f);
        }
    }
    int old = start ? tpg_clamp_idx(start[blockIdx.x], N) : 0;
    if (tid == 0) out[0] = old;
    float ox = x[(size_t)old * 3], oy = x[(size_t)old * 3 + 1], oz = x[(size_t)old * 3 + 2];
    const float zx = x[0], zy = x[1], zz = x[2];
    if (NW > 1 && tid < 32) slots[tid] = make_int2((int)0x80000000, 0x7fffffff);
    __syncthreads();

    for (int j = 1; j < m; ++j) {
        if (USE_LDS) { ox = sp[3 * old]; oy = sp[3 * old + 1]; oz = sp[3 * old + 2]; }
        // 1. lower bound of the squared distance from the pick to every tile's box (lane t: tile t)
        const float ex = fmaxf(fmaxf(b0x - ox, ox - b1x), 0.0f), ey = fmaxf(fmaxf(b0y - oy, oy - b1y), 0.0f),
                    ez = fmaxf(fmaxf(b0z - oz, oz - b1z), 0.0f);
        float lb = ex * ex;
        lb = lb + ey * ey;
        lb = lb + ez * ez;
        // (the margin: a computed distance is within a few ulp of the exact one, the exact one >= the exact bound,
        //  the computed bound within a few ulp of that -- 1e-5 relative covers it a hundred times over)
        const bool act = lane < PPT && (j == 1 || !(lb * 0.99999f >= __int_as_float(tmax)));
        const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__ballot(act));
        // 2. the tiles the pick can reach
#pragma unroll
        for (int t = 0; t < PPT; ++t) {
            if ((mask >> t) & 1u) {
                const float d = tpg_sq3(px[t], py[t], pz[t], ox, oy, oz);
                const int b2 = min(__float_as_int(d), tp[t]);
                tp[t] = b2;
                const int mx = tpg_wave_max_i32_rows(b2);
                const tpg_u64 bal = __ballot(b2 == mx);
                int a, src;
                if (__popcll(bal) == 1) {
                    src = __builtin_ctzll(bal);
                    a = __builtin_amdgcn_readlane(oi[t], src);
                } else {                                  // several points share the maximum: the smallest index
                    a = (int)tpg_wave_min_u32(b2 == mx ? (unsigned)oi[t] : 0x7fffffffu);
                    src = __builtin_ctzll(__ballot(b2 == mx && oi[t] == a));
                }
                if constexpr (!USE_LDS) {
                    const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[t]), src)),
                                cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[t]), src)),
                                cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[t]), src));
                    if (lane == t) { tcx = cx; tcy = cy; tcz = cz; }
                }
                if (lane == t) { tmax = mx; targ = a; }
            }
        }
        // 3. the wave's winner among its tile records (lanes 0..15 = one DPP row), then the workgroup's
        int mx = tpg_row16_max_i32(tmax);
        int bi = (int)tpg_row16_min_u32((lane < PPT && tmax == mx) ? (unsigned)targ : 0x7fffffffu);
        if constexpr (!USE_LDS) {
            if (lane < PPT && tmax == mx && targ == bi) {
                float *w = wxyz + ((j & 1) * 16 + wave) * 4;
                w[0] = tcx; w[1] = tcy; w[2] = tcz;
            }
        }
        int win = wave;
        if constexpr (NW > 1) {
            int2 *slot = slots + (j & 1) * 16;
            if (lane == 0) slot[wave] = make_int2(mx, bi);
            __syncthreads();
            const int2 sv = lane < NW ? slot[lane] : make_int2((int)0x80000000, 0x7fffffff);
            mx = tpg_row16_max_i32(sv.x);
            bi = (int)tpg_row16_min_u32((lane < NW && sv.x == mx) ? (unsigned)sv.y : 0x7fffffffu);
            win = __builtin_ctzll(__ballot(lane < NW && sv.x == mx && sv.y == bi));
        }
        old = mx >= 0 ? bi : 0;                   // mx < 0 <=> no eligible point at all
        if constexpr (!USE_LDS) {
            const float *w = wxyz + ((j & 1) * 16 + win) * 4;
            const float4 wv = *reinterpret_cast<const float4 *>(w);
            ox = mx >= 0 ? wv.x : zx; oy = mx >= 0 ? wv.y : zy; oz = mx >= 0 ? wv.z : zz;
        }
        if (tid == 0) out[j] = old;
        if (j == m - 1 && tid == 0 && prefix_out) prefix_out[blockIdx.x] = mx > 0 ? 1 : 0;
    }
    if (m == 1 && tid == 0 && prefix_out) prefix_out[blockIdx.x] = 1;
}

// fallback for clouds too large for registers: running distances in HBM scratch.
__global__ __launch_bounds__(1024) void fps_big_kernel(const float *__restrict__ xyz, int N, int m,
                                                       float *__restrict__ temp,
                                                       int32_t *__restrict__ idx,
                                                       const int32_t *__restrict__ start, int skip_origin,
                                                       const int32_t *__restrict__ prefix_in,
                                                       int32_t *__restrict__ prefix_out) {
    __shared__ tpg_u64 slots[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tpg_fps_prefix_shortcut(prefix_in, prefix_out, start, idx + (size_t)blockIdx.x * m, m, N, tid, 1024)) return;
    const float *x = xyz + (size_t)blockIdx.x * N * 3;
    float *tp = temp + (size_t)blockIdx.x * N;
    int32_t *out = idx + (size_t)blockIdx.x * m;
    for (int k = tid; k < N; k += 1024) tp[k] = 1e10f;
    int old = start ? tpg_clamp_idx(start[blockIdx.x], N) : 0;
    if (tid == 0) out[0] = old;
    __syncthreads();
    for (int j = 1; j < m; ++j) {
        const float ox = x[(size_t)old * 3], oy = x[(size_t)old * 3 + 1], oz = x[(size_t)old * 3 + 2];
        float best = -1.0f;
        int besti = 0;
        for (int k = tid; k < N; k += 1024) {
            const float ax = x[(size_t)k * 3], ay = x[(size_t)k * 3 + 1], az = x[(size_t)k * 3 + 2];
            float mag = ax * ax;
            mag = mag + ay * ay;
            mag = mag + az * az;
            if (skip_origin && mag <= 1e-3f) continue;
            const float d = tpg_sq3(ax, ay, az, ox, oy, oz);
            const float t0 = tp[k];
            const float d2 = d < t0 ? d : t0;
            tp[k] = d2;
            if (d2 > best) { best = d2; besti = k; }
        }
        tpg_u64 key = best >= 0.0f ? fps_key(best, besti) : 0ull;
        key = tpg_wave_max_u64(key);
        if (lane == 0) slots[j & 1][wave] = key;
        __syncthreads();
        tpg_u64 v = lane < 16 ? slots[j & 1][lane] : 0ull;
        tpg_u64 o;
        o = tpg_dpp_u64<0xB1>(v); v = o > v ? o : v;
        o = tpg_dpp_u64<0x4E>(v); v = o > v ? o : v;
        o = tpg_dpp_u64<0x124>(v); v = o > v ? o : v;
        o = tpg_dpp_u64<0x128>(v); v = o > v ? o : v;
        key = tpg_readlane_u64(v, 0);
        old = key ? (int)(~(unsigned)key) : 0;
        if (tid == 0) out[j] = old;
        // (the key's upper word is the distance's bits + 1: > 1 <=> a positive distance)
        if (j == m - 1 && tid == 0 && prefix_out) prefix_out[blockIdx.x] = (key >> 32) > 1ull ? 1 : 0;
    }
    if (m == 1 && tid == 0 && prefix_out) prefix_out[blockIdx.x] = 1;
}

template <int BLOCK, int PPT>
void fps_go(const float *xyz, int B, int N, int m, int32_t *idx, const int32_t *start, int skip_origin,
            const int32_t *pin, int32_t *pout, hipStream_t st) {
#ifdef TPG_FPS_NO_LDS_COPY
    int use_lds = 0;
#else
    int use_lds = N <= FPS_LDS_POINTS;
#endif
    size_t smem = 256 + 512 + (use_lds ? sizeof(float) * 3 * (size_t)N : 0);
    if (smem > 48 * 1024) {
        // opt in to > 48 KiB of dynamic LDS once per instantiation (not a stream operation, so
        // it must not happen inside a captured launch sequence): request the maximum this
        // instantiation can ever need.
        static int granted = -1;
        if (granted < 0) {
            const int want = 256 + 512 + (int)sizeof(float) * 3 * FPS_LDS_POINTS;
            granted = hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_kernel<BLOCK, PPT, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
            if (!granted) (void)hipGetLastError();
        }
        if (!granted) {
            use_lds = 0;  // read the selected point from L2 instead
            smem = 256 + 512;
        }
    }
    if (use_lds)
        hipLaunchKernelGGL((fps_kernel<BLOCK, PPT, true>), dim3(B), dim3(BLOCK), smem, st, xyz, N, m, idx, start,
                           skip_origin, pin, pout);
    else
        hipLaunchKernelGGL((fps_kernel<BLOCK, PPT, false>), dim3(B), dim3(BLOCK), smem, st, xyz, N, m, idx, start,
                           skip_origin, pin, pout);
}

template <int BLOCK>
int fps_pruned_go(const float *xyz, int B, int N, int m, int32_t *idx, const int32_t *start, int skip_origin,
                  const int32_t *pin, int32_t *pout, hipStream_t st) {
    const int use_lds = N <= FPP_LDS_POINTS;
    const size_t fixed = sizeof(float) * (64 + 128 + 128) + sizeof(int) * (FPP_CELLS + 2 + 32 + (size_t)BLOCK * 16);
    const size_t smem = fixed + (use_lds ? sizeof(float) * 3 * (size_t)N : 0);
    // > 48 KiB of dynamic LDS: opt in once per kernel (not a stream operation), with the most that kernel can need
    static int granted[2] = {-1, -1};
    if (granted[use_lds] < 0) {
        const void *fn = use_lds ? reinterpret_cast<const void *>(&fps_pruned_kernel<BLOCK, true>)
                                 : reinterpret_cast<const void *>(&fps_pruned_kernel<BLOCK, false>);
        const int cap = BLOCK * 16 < FPP_LDS_POINTS ? BLOCK * 16 : FPP_LDS_POINTS;
        const int want = (int)(fixed + (use_lds ? sizeof(float) * 3 * (size_t)cap : 0));
        granted[use_lds] = want <= 160 * 1024 &&
                           hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        if (!granted[use_lds]) (void)hipGetLastError();
    }
    if (!granted[use_lds]) return 0;
    if (use_lds)
        hipLaunchKernelGGL((fps_pruned_kernel<BLOCK, true>), dim3(B), dim3(BLOCK), smem, st, xyz, N, m, idx, start,
                           skip_origin, pin, pout);
    else
        hipLaunchKernelGGL((fps_pruned_kernel<BLOCK, false>), dim3(B), dim3(BLOCK), smem, st, xyz, N, m, idx, start,
                           skip_origin, pin, pout);
    return 1;
}

}  // namespace

#ifndef TPG_FPS_PRUNE_MIN_POINTS
#define TPG_FPS_PRUNE_MIN_POINTS 8193      // measured (tools/time_fps.py, us per round dense -> pruned): 4096 points 0.63 -> 1.02,
                                           // 8192 0.69 -> 0.91, 16384 1.60 -> 1.04: a pruned round costs ~1 us whatever the
                                           // size (a latency chain of wave reductions per updated tile, one wave per SIMD), a
                                           // dense one grows with N / CU -- so only the clouds of more than 8192 points take it
#endif
#ifndef TPG_FPS_PRUNE_MIN_PICKS
#define TPG_FPS_PRUNE_MIN_PICKS 128        // the sort costs a few dozen rounds
#endif

static int fps_dispatch(const float *xyz, const int32_t *start, int skip_origin, int B, int N, int m, float *temp,
                        int32_t *idx, const int32_t *pin, int32_t *pout, void *stream) {
    if (B < 0 || N <= 0 || m <= 0) return TPG_ERR_ARG;
    if (B == 0) return TPG_OK;
    if (!xyz || !idx) return TPG_ERR_ARG;
    hipStream_t st = tpg_stream(stream);
    // Workgroup shape (tools/tune_fps.py): one wave per SIMD first (4 waves), then up to 16 points per
    // thread before more waves -- a round is VALU-issue bound on its one CU, and every extra wave
    // repeats the reduction and lengthens the barrier.  (4096 points: 0.57 us/round with 256 threads,
    // 0.59 with 512, 0.63 with 1024.)
    static const bool prune = [] { const char *e = getenv("TPG_FPS_PRUNE"); return !(e && e[0] == '0'); }();   // A/B switch
    if (prune && N >= TPG_FPS_PRUNE_MIN_POINTS && N <= 16384 && m >= TPG_FPS_PRUNE_MIN_PICKS) {
        int done;
        if (N <= 4096) done = fps_pruned_go<256>(xyz, B, N, m, idx, start, skip_origin, pin, pout, st);
        else if (N <= 8192) done = fps_pruned_go<512>(xyz, B, N, m, idx, start, skip_origin, pin, pout, st);
        else done = fps_pruned_go<1024>(xyz, B, N, m, idx, start, skip_origin, pin, pout, st);
        if (done) {
            TPG_RETURN_IF_LAUNCH_FAILED();
            return TPG_OK;
        }
    }
#define TPG_FPS_SHAPE(n, block) fps_go<block, (n) / (block)>(xyz, B, N, m, idx, start, skip_origin, pin, pout, st)
#ifndef TPG_FPS_1024_BLOCK
#define TPG_FPS_1024_BLOCK 256
#endif
#ifndef TPG_FPS_2048_BLOCK
#define TPG_FPS_2048_BLOCK 256
#endif
#ifndef TPG_FPS_4096_BLOCK
#define TPG_FPS_4096_BLOCK 256
#endif
#ifndef TPG_FPS_8192_BLOCK
#define TPG_FPS_8192_BLOCK 512
#endif
    if (N <= 256) TPG_FPS_SHAPE(256, 64);
    else if (N <= 512) TPG_FPS_SHAPE(512, 128);
    else if (N <= 1024) TPG_FPS_SHAPE(1024, TPG_FPS_1024_BLOCK);
    else if (N <= 2048) TPG_FPS_SHAPE(2048, TPG_FPS_2048_BLOCK);
    else if (N <= 4096) TPG_FPS_SHAPE(4096, TPG_FPS_4096_BLOCK);
    else if (N <= 8192) TPG_FPS_SHAPE(8192, TPG_FPS_8192_BLOCK);
    else if (N <= 16384) TPG_FPS_SHAPE(16384, 1024);
    else {
        if (!temp) return TPG_ERR_ARG;
        hipLaunchKernelGGL(fps_big_kernel, dim3(B), dim3(1024), 0, st, xyz, N, m, temp, idx, start, skip_origin, pin, pout);
    }
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_fps_start_f32(const float *xyz, const int32_t *start, int skip_origin, int B, int N, int m,
                                 float *temp, int32_t *idx, void *stream) {
    return fps_dispatch(xyz, start, skip_origin, B, N, m, temp, idx, nullptr, nullptr, stream);
}

extern "C" int tpg_fps_f32(const float *xyz, int B, int N, int m, float *temp, int32_t *idx,
                           void *stream) {
    return fps_dispatch(xyz, nullptr, 1, B, N, m, temp, idx, nullptr, nullptr, stream);     // pointnet2 semantics
}

extern "C" int tpg_fps_prefix_f32(const float *xyz, int B, int N, int m, float *temp, int32_t *idx,
                                  const int32_t *prefix_in, int32_t *prefix_out, void *stream) {
    return fps_dispatch(xyz, nullptr, 1, B, N, m, temp, idx, prefix_in, prefix_out, stream);
}
