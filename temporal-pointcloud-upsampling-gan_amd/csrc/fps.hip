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

}  // namespace

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
