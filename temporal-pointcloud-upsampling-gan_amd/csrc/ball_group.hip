// ball_query, grouping_operation, gather_operation for gfx950.
//
// Replaces pointnet2_ops' query_ball_point / group_points / gather_points
// (reference: QueryAndGroup built at discriminator.py:190; grouping_operation at
// gcn_lib/pointnet/gcn.py:207,261 and discriminator.py:270,273; gather_operation
// at discriminator.py:131-137).
//
// ball_query  -- one wave64 per 4 queries, points read straight from L2
//   (see the kernel); the 64
//   lanes test 64 consecutive points per step, a ballot gives the in-radius
//   mask, popcount-prefix gives each hit its output slot in INDEX order, and the
//   wave stops as soon as nsample hits are found.  Upstream runs one thread per
//   query and one block per cloud (8 blocks on a 256-CU part).
// group fwd   -- pure streaming: 16-B index loads, 16-B output stores along the
//   contiguous (s,k) axis, the gathered feature row (N*4 B) stays in L1/L2 and
//   the index vector is reused across a tile of channels.
// group bwd   -- scatter-add privatised in LDS: a workgroup owns a tile of
//   channels of one cloud, accumulates with ds_add_f32 and writes each
//   gradient row once; no global atomics, no zero-fill pass.
#include "tpg_common.hpp"

namespace {

// ---------------------------------------------------------------- ball query
// One wave per BQ_QPW queries, no LDS and no barrier: the 64 lanes read 64 consecutive points of the
// cloud straight from L2 (a cloud is 48 KB .. 200 KB and every wave walks it from index 0, so after the
// first touch it is all cache hits; the next step's points are loaded while the current ones are
// tested), each point is tested against the wave's BQ_QPW queries, a ballot gives the in-radius mask,
// popcount-prefix the output slot in INDEX order -- and the wave STOPS as soon as its queries have
// their nsample hits.  At the step's radii that is after 150..500 of 4096 points: staging the whole
// cloud in LDS per workgroup (the first form of this kernel: 48 KB x 64 workgroups per cloud, 2.8x
// the algorithmic traffic at the memory side, a barrier per chunk) cost more than the search.
constexpr int BQ_WAVES = 4;
constexpr int BQ_QPW = 4;       // queries per wave (one load of a point serves all of them)

__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, int N, int S, float r2,
    int nsample, int32_t *__restrict__ idx) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const float *x = xyz + (size_t)b * N * 3;
    const int q0 = (blockIdx.x * BQ_WAVES + wave) * BQ_QPW;
    if (q0 >= S) return;

    float qx[BQ_QPW], qy[BQ_QPW], qz[BQ_QPW];
    int cnt[BQ_QPW], first[BQ_QPW];
#pragma unroll
    for (int u = 0; u < BQ_QPW; ++u) {
        const int s = min(q0 + u, S - 1);
        const float *q = new_xyz + ((size_t)b * S + s) * 3;
        qx[u] = q[0]; qy[u] = q[1]; qz[u] = q[2];
        cnt[u] = (q0 + u < S) ? 0 : nsample;  // out-of-range queries are "done"
        first[u] = 0;
    }
    const tpg_u64 below = (1ull << lane) - 1ull;
    // this lane's point of the current step (clamped loads: the tail lanes are masked by p < N)
    int pc = min(lane, N - 1);
    float px = x[(size_t)pc * 3], py = x[(size_t)pc * 3 + 1], pz = x[(size_t)pc * 3 + 2];
    for (int base = 0; base < N; base += 64) {
        const int pn = min(base + 64 + lane, N - 1);           // next step's point travels during this one
        const float nx = x[(size_t)pn * 3], ny = x[(size_t)pn * 3 + 1], nz = x[(size_t)pn * 3 + 2];
        const int p = base + lane;
        bool open = false;
#pragma unroll
        for (int u = 0; u < BQ_QPW; ++u) {
            if (cnt[u] >= nsample) continue;                   // wave-uniform
            const bool in = p < N && tpg_sq3(qx[u], qy[u], qz[u], px, py, pz) < r2;
            const tpg_u64 mask = __ballot(in);
            if (mask) {
                if (cnt[u] == 0) first[u] = base + __builtin_ctzll(mask);
                const int pos = cnt[u] + __popcll(mask & below);
                if (in && pos < nsample) idx[((size_t)b * S + (q0 + u)) * nsample + pos] = p;
                cnt[u] += __popcll(mask);
            }
            open = open || cnt[u] < nsample;
        }
        if (!open) break;                                      // every query of the wave is full
        px = nx; py = ny; pz = nz;
    }
#pragma unroll
    for (int u = 0; u < BQ_QPW; ++u) {
        if (q0 + u >= S) continue;
        int32_t *out = idx + ((size_t)b * S + (q0 + u)) * nsample;
        const int have = min(cnt[u], nsample);
        const int fill = cnt[u] ? first[u] : 0;
        for (int l = have + lane; l < nsample; l += 64) out[l] = fill;
    }
}

// ---------------------------------------------------------------- group fwd
constexpr int GF_CT = 8;  // channels per workgroup (index vector reused GF_CT times)

template <bool VEC>
__global__ __launch_bounds__(256) void group_fwd_kernel(const float *__restrict__ feat,
                                                        const int32_t *__restrict__ idx, int C, int N,
                                                        long long SK, float *__restrict__ out) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * GF_CT;
    const int cn = min(GF_CT, C - c0);
    const long long e0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e0 >= SK) return;
    const int32_t *id = idx + (size_t)b * SK + e0;
    int i0, i1, i2, i3;
    const int rem = (int)min((long long)4, SK - e0);
    if (VEC) {
        const int4 v = *reinterpret_cast<const int4 *>(id);
        i0 = v.x; i1 = v.y; i2 = v.z; i3 = v.w;
    } else {
        i0 = id[0]; i1 = rem > 1 ? id[1] : 0; i2 = rem > 2 ? id[2] : 0; i3 = rem > 3 ? id[3] : 0;
    }
    i0 = tpg_clamp_idx(i0, N); i1 = tpg_clamp_idx(i1, N);
    i2 = tpg_clamp_idx(i2, N); i3 = tpg_clamp_idx(i3, N);
    for (int c = 0; c < cn; ++c) {
        const float *f = feat + ((size_t)b * C + c0 + c) * N;
        float *o = out + ((size_t)b * C + c0 + c) * SK + e0;
        const float4 v = make_float4(f[i0], f[i1], f[i2], f[i3]);
        if (VEC) {
            *reinterpret_cast<float4 *>(o) = v;
        } else {
            o[0] = v.x;
            if (rem > 1) o[1] = v.y;
            if (rem > 2) o[2] = v.z;
            if (rem > 3) o[3] = v.w;
        }
    }
}

// ---------------------------------------------------------------- group bwd
// dynamic LDS: acc[ct][N]
__global__ __launch_bounds__(256) void group_bwd_lds_kernel(const float *__restrict__ gout,
                                                            const int32_t *__restrict__ idx, int C,
                                                            int N, long long SK, int ct,
                                                            float *__restrict__ gfeat) {
    extern __shared__ __attribute__((aligned(16))) float gb_acc[];
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * ct;
    const int cn = min(ct, C - c0);
    const int tid = threadIdx.x;
    for (int e = tid; e < cn * N; e += 256) gb_acc[e] = 0.0f;
    __syncthreads();
    const int32_t *id = idx + (size_t)b * SK;
    for (long long e = tid; e < SK; e += 256) {
        const int n = tpg_clamp_idx(id[e], N);
        for (int c = 0; c < cn; ++c)
            atomicAdd(&gb_acc[c * N + n], gout[((size_t)b * C + c0 + c) * SK + e]);
    }
    __syncthreads();
    float *g = gfeat + ((size_t)b * C + c0) * N;
    for (int e = tid; e < cn * N; e += 256) g[e] = gb_acc[e];
}

// very large N: global atomics into a zeroed buffer
__global__ __launch_bounds__(256) void group_bwd_atomic_kernel(const float *__restrict__ gout,
                                                               const int32_t *__restrict__ idx, int C,
                                                               int N, long long SK,
                                                               float *__restrict__ gfeat) {
    const int b = blockIdx.z, c = blockIdx.y;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= SK) return;
    const int n = tpg_clamp_idx(idx[(size_t)b * SK + e], N);
    atomicAdd(gfeat + ((size_t)b * C + c) * N + n, gout[((size_t)b * C + c) * SK + e]);
}

// ---------------------------------------------------------------- gather
__global__ __launch_bounds__(256) void gather_fwd_kernel(const float *__restrict__ feat,
                                                         const int32_t *__restrict__ idx, int C, int N,
                                                         int S, float *__restrict__ out) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const int n = tpg_clamp_idx(idx[(size_t)b * S + s], N);
    out[((size_t)b * C + c) * S + s] = feat[((size_t)b * C + c) * N + n];
}

__global__ __launch_bounds__(256) void gather_bwd_kernel(const float *__restrict__ gout,
                                                         const int32_t *__restrict__ idx, int C, int N,
                                                         int S, float *__restrict__ gfeat) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const int n = tpg_clamp_idx(idx[(size_t)b * S + s], N);
    atomicAdd(gfeat + ((size_t)b * C + c) * N + n, gout[((size_t)b * C + c) * S + s]);
}

// the same gather on channels-last rows: out[b,s,:] = rows[b,idx[b,s],:] (C small: point coordinates).  The rows path
// of the set-abstraction levels keeps clouds as (B,N,3); through the planes form above a centre gather was transpose
// copy -> gather -> transpose copy, three launches each way, twice per level and pass.
__global__ __launch_bounds__(256) void gather_rows_fwd_kernel(const float *__restrict__ rows,
                                                              const int32_t *__restrict__ idx, int N, int S, int C,
                                                              float *__restrict__ out, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long long bs = t / C;
    const int c = (int)(t - bs * C);
    const long long b = bs / S;
    const int n = tpg_clamp_idx(idx[bs], N);
    out[t] = rows[((size_t)b * N + n) * C + c];
}

__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const float *__restrict__ gout,
                                                              const int32_t *__restrict__ idx, int N, int S, int C,
                                                              float *__restrict__ grows, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long long bs = t / C;
    const int c = (int)(t - bs * C);
    const long long b = bs / S;
    const int n = tpg_clamp_idx(idx[bs], N);
    atomicAdd(grows + ((size_t)b * N + n) * C + c, gout[t]);
}

bool grid_ok(long long x, long long y, long long z) {
    return x > 0 && y > 0 && z > 0 && x <= 0x7fffffffLL && y <= 65535 && z <= 65535;
}

}  // namespace

extern "C" int tpg_ball_query_f32(const float *xyz, const float *new_xyz, int B, int N, int S,
                                  float radius, int nsample, int32_t *idx, void *stream) {
    if (B < 0 || N <= 0 || S < 0 || nsample <= 0) return TPG_ERR_ARG;
    if (B == 0 || S == 0) return TPG_OK;
    if (!xyz || !new_xyz || !idx) return TPG_ERR_ARG;
    const float r2 = radius * radius;
    const int per_block = BQ_WAVES * BQ_QPW;
    const long long gx = (S + per_block - 1) / per_block;
    if (!grid_ok(gx, B, 1)) return TPG_ERR_ARG;
    hipLaunchKernelGGL(ball_query_kernel, dim3((unsigned)gx, (unsigned)B), dim3(BQ_WAVES * 64), 0,
                       tpg_stream(stream), xyz, new_xyz, N, S, r2, nsample, idx);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_group_fwd_f32(const float *feat, const int32_t *idx, int B, int C, int N, int S,
                                 int K, float *out, void *stream) {
    if (B < 0 || C < 0 || N <= 0 || S < 0 || K < 0) return TPG_ERR_ARG;
    const long long SK = (long long)S * K;
    if (B == 0 || C == 0 || SK == 0) return TPG_OK;
    if (!feat || !idx || !out) return TPG_ERR_ARG;
    const long long gx = (SK + 1023) / 1024, gy = (C + GF_CT - 1) / GF_CT;
    if (!grid_ok(gx, gy, B)) return TPG_ERR_ARG;
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)B);
    const bool vec = (SK % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(group_fwd_kernel<true>, grid, dim3(256), 0, tpg_stream(stream), feat, idx, C,
                           N, SK, out);
    else
        hipLaunchKernelGGL(group_fwd_kernel<false>, grid, dim3(256), 0, tpg_stream(stream), feat, idx, C,
                           N, SK, out);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_group_bwd_f32(const float *gout, const int32_t *idx, int B, int C, int N, int S,
                                 int K, float *gfeat, void *stream) {
    if (B < 0 || C < 0 || N <= 0 || S < 0 || K < 0) return TPG_ERR_ARG;
    const long long SK = (long long)S * K;
    if (B == 0 || C == 0) return TPG_OK;
    if (!gfeat) return TPG_ERR_ARG;
    hipStream_t st = tpg_stream(stream);
    if (SK == 0) {
        if (hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)B * C * N, st) != hipSuccess) return TPG_ERR_LAUNCH;
        return TPG_OK;
    }
    if (!gout || !idx) return TPG_ERR_ARG;
    const size_t row = sizeof(float) * (size_t)N;
    if (row <= 48 * 1024) {
        // channel tile: fill LDS up to 48 KiB but keep >= ~512 workgroups when C allows
        int ct = (int)((48 * 1024) / row);
        ct = ct > 8 ? 8 : ct;
        while (ct > 1 && (long long)B * ((C + ct - 1) / ct) < 512) ct >>= 1;
        const long long gx = (C + ct - 1) / ct;
        if (!grid_ok(gx, B, 1)) return TPG_ERR_ARG;
        hipLaunchKernelGGL(group_bwd_lds_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256),
                           row * ct, st, gout, idx, C, N, SK, ct, gfeat);
    } else {
        if (hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)B * C * N, st) != hipSuccess) return TPG_ERR_LAUNCH;
        const long long gx = (SK + 255) / 256;
        if (!grid_ok(gx, C, B)) return TPG_ERR_ARG;
        hipLaunchKernelGGL(group_bwd_atomic_kernel, dim3((unsigned)gx, (unsigned)C, (unsigned)B),
                           dim3(256), 0, st, gout, idx, C, N, SK, gfeat);
    }
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_gather_fwd_f32(const float *feat, const int32_t *idx, int B, int C, int N, int S,
                                  float *out, void *stream) {
    if (B < 0 || C < 0 || N <= 0 || S < 0) return TPG_ERR_ARG;
    if (B == 0 || C == 0 || S == 0) return TPG_OK;
    if (!feat || !idx || !out) return TPG_ERR_ARG;
    const long long gx = (S + 255) / 256;
    if (!grid_ok(gx, C, B)) return TPG_ERR_ARG;
    hipLaunchKernelGGL(gather_fwd_kernel, dim3((unsigned)gx, (unsigned)C, (unsigned)B), dim3(256), 0,
                       tpg_stream(stream), feat, idx, C, N, S, out);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_gather_bwd_f32(const float *gout, const int32_t *idx, int B, int C, int N, int S,
                                  float *gfeat, void *stream) {
    if (B < 0 || C < 0 || N <= 0 || S < 0) return TPG_ERR_ARG;
    if (B == 0 || C == 0) return TPG_OK;
    if (!gfeat) return TPG_ERR_ARG;
    hipStream_t st = tpg_stream(stream);
    if (hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)B * C * N, st) != hipSuccess) return TPG_ERR_LAUNCH;
    if (S == 0) return TPG_OK;
    if (!gout || !idx) return TPG_ERR_ARG;
    const long long gx = (S + 255) / 256;
    if (!grid_ok(gx, C, B)) return TPG_ERR_ARG;
    hipLaunchKernelGGL(gather_bwd_kernel, dim3((unsigned)gx, (unsigned)C, (unsigned)B), dim3(256), 0, st,
                       gout, idx, C, N, S, gfeat);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_gather_rows_fwd_f32(const float *rows, const int32_t *idx, int B, int N, int S, int C, float *out,
                                       void *stream) {
    if (B < 0 || C < 0 || N <= 0 || S < 0) return TPG_ERR_ARG;
    const long long total = (long long)B * S * C;
    if (total == 0) return TPG_OK;
    if (!rows || !idx || !out) return TPG_ERR_ARG;
    const long long gx = (total + 255) / 256;
    if (!grid_ok(gx, 1, 1)) return TPG_ERR_ARG;
    hipLaunchKernelGGL(gather_rows_fwd_kernel, dim3((unsigned)gx), dim3(256), 0, tpg_stream(stream), rows, idx, N, S, C, out,
                       total);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_gather_rows_bwd_f32(const float *gout, const int32_t *idx, int B, int N, int S, int C, float *grows,
                                       void *stream) {
    if (B < 0 || C < 0 || N <= 0 || S < 0) return TPG_ERR_ARG;
    if (B == 0 || C == 0) return TPG_OK;
    if (!grows) return TPG_ERR_ARG;
    hipStream_t st = tpg_stream(stream);
    if (hipMemsetAsync(grows, 0, sizeof(float) * (size_t)B * N * C, st) != hipSuccess) return TPG_ERR_LAUNCH;
    const long long total = (long long)B * S * C;
    if (total == 0) return TPG_OK;
    if (!gout || !idx) return TPG_ERR_ARG;
    const long long gx = (total + 255) / 256;
    if (!grid_ok(gx, 1, 1)) return TPG_ERR_ARG;
    hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3((unsigned)gx), dim3(256), 0, st, gout, idx, N, S, C, grows, total);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}
