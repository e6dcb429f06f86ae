// Fixed-radius nearest neighbours on a uniform grid: frnn.frnn_grid_points for clouds where the
// exhaustive search of knn.hip stops being free (reference call sites: loss.py:256-265 -- the mask
// loss searches 16384 x 16384 points per cloud at BASELINE cfg5 -- and the 10^4..10^5-point rollout of
// upsampling_network.py:159-174; discriminator.py:27-32 and gcn_lib/interpolation.py:20,33 at sizes
// where either search will do).
//
// Same results as the exhaustive kernel, bit for bit: a candidate's distance is the same canonical
// fp32 sum on the same coordinates, the K survivors are chosen by the same 64-bit key (dist, idx) --
// which makes the order in which candidates are met irrelevant -- and the cells are cut so that no
// pair closer than r can sit more than one cell apart:
//
//   cell edge  h = max(r, extent / 64) * (1 + 1e-4)     (>= r with a margin far above fp32 rounding)
//   cell(p)    = floor((p - lo) / h) per axis, the same expression for stored points and queries
//
// Build (four small launches per call, all clouds at once):
//   bbox   per cloud: lo, 1/h, grid dims (<= 64 per axis)            one workgroup per cloud
//   count  cell of every point, int atomics on the cell counters     (deterministic totals)
//   scan   exclusive prefix of the counters                          one workgroup per cloud
//   fill   points copied into cell order as (x, y, z, original index)
// Query: one WAVE per query (as knn.hip): the 27 neighbour cells are 9 runs of 3 x-adjacent cells,
// each a contiguous range of the cell-sorted array; the lanes take 64 candidates of the concatenated
// ranges per step and the survivors (d < r^2) enter the wave's K-best list by rank merge
// (knn_select.hpp).  At the particle spacing of the fluid clips a query meets ~75 candidates instead
// of 16384.
#include "knn_select.hpp"
#include "tpg_common.hpp"

namespace {

constexpr int FG_MAXDIM = 64;           // cells per axis
constexpr int FG_WAVES = 4;             // queries per workgroup

struct GridParams {                     // per cloud, 8 floats / ints
    float lo[3];
    float inv_h;
    int dim[3];
    int ncell;
};

__device__ __forceinline__ int cell_axis(float p, float lo, float inv_h) {
    return (int)floorf((p - lo) * inv_h);
}

__global__ __launch_bounds__(1024) void fg_bbox_kernel(const float *__restrict__ p2, const int64_t *__restrict__ len2,
                                                       int P2, float r, int knn_k, GridParams *__restrict__ gp) {
    __shared__ float red[6][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n2 = len2 ? min((int)len2[b], P2) : P2;
    const float *x = p2 + (size_t)b * P2 * 3;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = tid; i < n2; i += 1024)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = x[(size_t)i * 3 + d];
            mn[d] = fminf(mn[d], v);
            mx[d] = fmaxf(mx[d], v);
        }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], s));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], s));
        }
        if (lane == 0) { red[d][wave] = mn[d]; red[3 + d][wave] = mx[d]; }
    }
    __syncthreads();
    if (tid == 0) {
        float ext = 0.0f;
        float lo[3];
        for (int d = 0; d < 3; ++d) {
            float a = red[d][0], c = red[3 + d][0];
            for (int w = 1; w < 16; ++w) { a = fminf(a, red[d][w]); c = fmaxf(c, red[3 + d][w]); }
            lo[d] = n2 > 0 ? a : 0.0f;
            ext = fmaxf(ext, n2 > 0 ? c - a : 0.0f);
        }
        float h = fmaxf(r, ext / (float)FG_MAXDIM) * 1.0001f;
        if (knn_k > 0) {
            // kNN form: ~K/2 points per cell (flat directions count as one cell edge of the coarsest grid), so that the
            // ball inscribed in the 27 cells around a query's cell holds ~2 K points and one pass usually settles it
            float vol = 1.0f;
            for (int d = 0; d < 3; ++d) {
                float a = red[d][0], c = red[3 + d][0];
                for (int w = 1; w < 16; ++w) { a = fminf(a, red[d][w]); c = fmaxf(c, red[3 + d][w]); }
                vol *= fmaxf(n2 > 0 ? c - a : 0.0f, ext / (float)FG_MAXDIM);
            }
            const float want = cbrtf(vol * 0.5f * (float)max(knn_k, 8) / (float)max(n2, 1));
            h = fmaxf(want, ext / (float)FG_MAXDIM) * 1.0001f;
            if (!(h > 0.0f)) h = 1.0f;                      // a cloud of identical points: one cell
        }
        GridParams g;
        g.inv_h = 1.0f / h;
        int nc = 1;
        for (int d = 0; d < 3; ++d) {
            g.lo[d] = lo[d];
            float c = red[3 + d][0];
            for (int w = 1; w < 16; ++w) c = fmaxf(c, red[3 + d][w]);
            int n = n2 > 0 ? cell_axis(c, lo[d], g.inv_h) + 1 : 1;
            n = n < 1 ? 1 : (n > FG_MAXDIM ? FG_MAXDIM : n);
            g.dim[d] = n;
            nc *= n;
        }
        g.ncell = nc;
        gp[b] = g;
    }
}

__device__ __forceinline__ int cell_of(const GridParams &g, float px, float py, float pz) {
    int cx = cell_axis(px, g.lo[0], g.inv_h), cy = cell_axis(py, g.lo[1], g.inv_h), cz = cell_axis(pz, g.lo[2], g.inv_h);
    cx = min(max(cx, 0), g.dim[0] - 1);
    cy = min(max(cy, 0), g.dim[1] - 1);
    cz = min(max(cz, 0), g.dim[2] - 1);
    return (cz * g.dim[1] + cy) * g.dim[0] + cx;
}

// counts[b][cell] += 1; cellid[b][i] = cell        grid (ceil(P2/256), B)
__global__ __launch_bounds__(256) void fg_count_kernel(const float *__restrict__ p2, const int64_t *__restrict__ len2,
                                                       int P2, const GridParams *__restrict__ gp, int cstride,
                                                       int *__restrict__ counts, int *__restrict__ cellid) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n2 = len2 ? min((int)len2[b], P2) : P2;
    if (i >= n2) return;
    const GridParams g = gp[b];
    const float *x = p2 + ((size_t)b * P2 + i) * 3;
    const int c = cell_of(g, x[0], x[1], x[2]);
    cellid[(size_t)b * P2 + i] = c;
    atomicAdd(&counts[(size_t)b * cstride + c], 1);
}

// start[b][c] = exclusive prefix of counts (start has ncell + 1 entries); counts are zeroed again to
// serve as the fill cursors.                          grid (B)
__global__ __launch_bounds__(1024) void fg_scan_kernel(const GridParams *__restrict__ gp, int cstride,
                                                       int *__restrict__ counts, int *__restrict__ start) {
    __shared__ int wsum[32];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nc = gp[b].ncell;
    int *cnt = counts + (size_t)b * cstride, *st = start + (size_t)b * (cstride + 1);
    const int per = (nc + 1023) / 1024;
    const int lo = min(tid * per, nc), hi = min(lo + per, nc);
    int local = 0;
    for (int c = lo; c < hi; ++c) local += cnt[c];
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        int w = lane < 16 ? wsum[lane] : 0;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const int o = __shfl_up(w, d);
            if (lane >= d) w += o;
        }
        if (lane < 16) wsum[16 + lane] = w;
    }
    __syncthreads();
    int run = incl - local + (wave ? wsum[16 + wave - 1] : 0);
    for (int c = lo; c < hi; ++c) {
        const int n = cnt[c];
        st[c] = run;
        cnt[c] = 0;
        run += n;
    }
    if (tid == 0) st[nc] = wsum[16 + 15];
}

// sorted[b][start[cell] + k] = (x, y, z, bits of i)      grid (ceil(P2/256), B)
__global__ __launch_bounds__(256) void fg_fill_kernel(const float *__restrict__ p2, const int64_t *__restrict__ len2,
                                                      int P2, int cstride, const int *__restrict__ cellid,
                                                      const int *__restrict__ start, int *__restrict__ cursor,
                                                      float4 *__restrict__ sorted) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n2 = len2 ? min((int)len2[b], P2) : P2;
    if (i >= n2) return;
    const int c = cellid[(size_t)b * P2 + i];
    const int pos = start[(size_t)b * (cstride + 1) + c] + atomicAdd(&cursor[(size_t)b * cstride + c], 1);
    const float *x = p2 + ((size_t)b * P2 + i) * 3;
    sorted[(size_t)b * P2 + pos] = make_float4(x[0], x[1], x[2], __int_as_float(i));
}

__device__ __forceinline__ tpg_u64 fg_pack(float d, int j) {
    return ((tpg_u64)__float_as_uint(d) << 32) | (unsigned)j;
}

// one wave per query: the K nearest stored points with d < r2, ascending (dist, idx); -1 / -1 padding
__global__ __launch_bounds__(FG_WAVES * 64) void fg_query_kernel(
    const float *__restrict__ p1, const int64_t *__restrict__ len1, int P1, int P2,
    const GridParams *__restrict__ gp, int cstride, const int *__restrict__ start,
    const float4 *__restrict__ sorted, int K, float r2, float *__restrict__ dist, int64_t *__restrict__ idx) {
    __shared__ tpg_u64 slots[FG_WAVES * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    tpg_u64 *slot = slots + wave * 64;
    const int b = blockIdx.y;
    const int i = blockIdx.x * FG_WAVES + wave;
    if (i >= P1) return;
    const size_t q = (size_t)b * P1 + i;
    float *od = dist + q * K;
    int64_t *oi = idx + q * K;
    const int n1 = len1 ? (int)len1[b] : P1;
    if (i >= n1) {
        for (int k = lane; k < K; k += 64) { od[k] = -1.0f; oi[k] = -1; }
        return;
    }
    const GridParams g = gp[b];
    const float qx = p1[q * 3], qy = p1[q * 3 + 1], qz = p1[q * 3 + 2];
    const int cx = cell_axis(qx, g.lo[0], g.inv_h), cy = cell_axis(qy, g.lo[1], g.inv_h), cz = cell_axis(qz, g.lo[2], g.inv_h);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
    const int *st = start + (size_t)b * (cstride + 1);
    const float4 *pts = sorted + (size_t)b * P2;
    // the 9 (dz, dy) runs: [begin, end) of the cell-sorted array, and their running total
    int rb[9], re[9], total = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int zz = cz + t / 3 - 1, yy = cy + t % 3 - 1;
        int bgn = 0, end = 0;
        if (x0 <= x1 && zz >= 0 && zz < g.dim[2] && yy >= 0 && yy < g.dim[1]) {
            const int row = (zz * g.dim[1] + yy) * g.dim[0];
            bgn = st[row + x0];
            end = st[row + x1 + 1];
        }
        rb[t] = bgn;
        re[t] = end;
        total += end - bgn;
    }
    const tpg_u64 INF = ~0ull;
    tpg_u64 best = INF, thr = INF;
    for (int base = 0; base < total; base += 64) {
        // candidate number base + lane of the concatenated runs
        int c = base + lane, pos = -1;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int n = re[t] - rb[t];
            if (pos < 0 && c < n) pos = rb[t] + c;
            c -= (pos < 0) ? n : 0;
        }
        tpg_u64 key = INF;
        if (pos >= 0 && base + lane < total) {
            const float4 p = pts[pos];
            const float t0 = qx - p.x, t1 = qy - p.y, t2 = qz - p.z;
            float d = t0 * t0;
            d = d + t1 * t1;
            d = d + t2 * t2;
            if (d < r2) key = fg_pack(d, __float_as_int(p.w));
        }
        if (K == 1) best = key < best ? key : best;
        else tpg_knn_merge(best, thr, key, K, lane, slot);
    }
    if (K == 1) {
        best = tpg_wave_min_u64(best);
        if (lane == 0) {
            if (best == INF) { od[0] = -1.0f; oi[0] = -1; }
            else { od[0] = __uint_as_float((unsigned)(best >> 32)); oi[0] = (long long)(unsigned)best; }
        }
        return;
    }
    if (lane < K) {
        if (best == INF) { od[lane] = -1.0f; oi[lane] = -1; }
        else { od[lane] = __uint_as_float((unsigned)(best >> 32)); oi[lane] = (long long)(unsigned)best; }
    }
}

// Plain kNN (no radius) on the same grid: the block of (2R+1)^3 cells around the query's cell, R = 1 first; the
// search is settled when the K-th distance lies inside the part of space the block certainly covers (distance from
// the query to the nearest block face that is not a face of the grid, less a rounding margin), else R grows and the
// block is walked again.  Same canonical distance and 64-bit key as knn_kernel: bit-identical lists.
__global__ __launch_bounds__(FG_WAVES * 64) void fg_knn_kernel(
    const float *__restrict__ p1, const int64_t *__restrict__ len1, int P1, int P2,
    const GridParams *__restrict__ gp, int cstride, const int *__restrict__ start,
    const float4 *__restrict__ sorted, int K, float *__restrict__ dist, int64_t *__restrict__ idx) {
    __shared__ tpg_u64 slots[FG_WAVES * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    tpg_u64 *slot = slots + wave * 64;
    const int b = blockIdx.y;
    const int i = blockIdx.x * FG_WAVES + wave;
    if (i >= P1) return;
    const size_t q = (size_t)b * P1 + i;
    float *od = dist + q * K;
    int64_t *oi = idx + q * K;
    const int n1 = len1 ? (int)len1[b] : P1;
    const GridParams g = gp[b];
    const int *st = start + (size_t)b * (cstride + 1);
    if (i >= n1 || st[g.ncell] <= 0) {
        for (int k = lane; k < K; k += 64) { od[k] = 0.0f; oi[k] = 0; }
        return;
    }
    const float4 *pts = sorted + (size_t)b * P2;
    const float qx = p1[q * 3], qy = p1[q * 3 + 1], qz = p1[q * 3 + 2];
    const float h = 1.0f / g.inv_h;
    const int cx = min(max(cell_axis(qx, g.lo[0], g.inv_h), 0), g.dim[0] - 1);
    const int cy = min(max(cell_axis(qy, g.lo[1], g.inv_h), 0), g.dim[1] - 1);
    const int cz = min(max(cell_axis(qz, g.lo[2], g.inv_h), 0), g.dim[2] - 1);
    const tpg_u64 INF = ~0ull;
    tpg_u64 best = INF;
    for (int R = 1;; ++R) {
        const int x0 = max(cx - R, 0), x1 = min(cx + R, g.dim[0] - 1);
        const int y0 = max(cy - R, 0), y1 = min(cy + R, g.dim[1] - 1);
        const int z0 = max(cz - R, 0), z1 = min(cz + R, g.dim[2] - 1);
        best = INF;
        tpg_u64 thr = INF;
        for (int zz = z0; zz <= z1; ++zz)
            for (int yy = y0; yy <= y1; ++yy) {
                const int row = (zz * g.dim[1] + yy) * g.dim[0];
                const int bgn = st[row + x0], end = st[row + x1 + 1];
                for (int base = bgn; base < end; base += 64) {
                    const int pos = base + lane;
                    tpg_u64 key = INF;
                    if (pos < end) {
                        const float4 p = pts[pos];
                        const float t0 = qx - p.x, t1 = qy - p.y, t2 = qz - p.z;
                        float d = t0 * t0;
                        d = d + t1 * t1;
                        d = d + t2 * t2;
                        key = fg_pack(d, __float_as_int(p.w));
                    }
                    if (K == 1) best = key < best ? key : best;
                    else tpg_knn_merge(best, thr, key, K, lane, slot);
                }
            }
        const tpg_u64 kth = K == 1 ? tpg_wave_min_u64(best) : tpg_readlane_u64(best, K - 1);
        if (K == 1) best = kth;
        const bool whole = x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.dim[0] - 1 && y1 == g.dim[1] - 1 && z1 == g.dim[2] - 1;
        if (whole) break;
        float bd = 3.0e38f;
        if (x0 > 0) bd = fminf(bd, qx - (g.lo[0] + (float)x0 * h));
        if (x1 < g.dim[0] - 1) bd = fminf(bd, (g.lo[0] + (float)(x1 + 1) * h) - qx);
        if (y0 > 0) bd = fminf(bd, qy - (g.lo[1] + (float)y0 * h));
        if (y1 < g.dim[1] - 1) bd = fminf(bd, (g.lo[1] + (float)(y1 + 1) * h) - qy);
        if (z0 > 0) bd = fminf(bd, qz - (g.lo[2] + (float)z0 * h));
        if (z1 < g.dim[2] - 1) bd = fminf(bd, (g.lo[2] + (float)(z1 + 1) * h) - qz);
        bd -= 1.0e-4f * h;                               // a point within rounding of a cell face may sit in either cell
        if (kth != INF && bd > 0.0f && __uint_as_float((unsigned)(kth >> 32)) < bd * bd * 0.999999f) break;
    }
    if (lane < K) {
        const tpg_u64 mine = best;
        if (mine == INF) { od[lane] = 0.0f; oi[lane] = 0; }
        else { od[lane] = __uint_as_float((unsigned)(mine >> 32)); oi[lane] = (long long)(unsigned)mine; }
    }
}

constexpr int FG_CELLS = FG_MAXDIM * FG_MAXDIM * FG_MAXDIM;

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t tpg_frnn_grid_workspace_bytes(int B, int P2) {
    if (B <= 0 || P2 <= 0) return 0;
    // params | counters / cursors | starts | cell ids | cell-sorted points
    return align256(sizeof(GridParams) * (size_t)B) + align256(sizeof(int) * (size_t)B * FG_CELLS) +
           align256(sizeof(int) * (size_t)B * (FG_CELLS + 1)) + align256(sizeof(int) * (size_t)B * P2) +
           align256(sizeof(float4) * (size_t)B * P2);
}

// grid build shared by the two entries; r > 0: radius form, knn_k > 0: kNN form
static int fg_build(const float *p2, const int64_t *len2, int B, int P2, float r, int knn_k, void *ws, hipStream_t st,
                    GridParams **gp_o, int **start_o, float4 **sorted_o) {
    unsigned char *w = static_cast<unsigned char *>(ws);
    GridParams *gp = reinterpret_cast<GridParams *>(w);
    w += align256(sizeof(GridParams) * (size_t)B);
    int *counts = reinterpret_cast<int *>(w);
    w += align256(sizeof(int) * (size_t)B * FG_CELLS);
    int *start = reinterpret_cast<int *>(w);
    w += align256(sizeof(int) * (size_t)B * (FG_CELLS + 1));
    int *cellid = reinterpret_cast<int *>(w);
    w += align256(sizeof(int) * (size_t)B * P2);
    float4 *sorted = reinterpret_cast<float4 *>(w);
    if (hipMemsetAsync(counts, 0, sizeof(int) * (size_t)B * FG_CELLS, st) != hipSuccess) return TPG_ERR_LAUNCH;
    const dim3 pg((P2 + 255) / 256, B);
    hipLaunchKernelGGL(fg_bbox_kernel, dim3(B), dim3(1024), 0, st, p2, len2, P2, r, knn_k, gp);
    hipLaunchKernelGGL(fg_count_kernel, pg, dim3(256), 0, st, p2, len2, P2, gp, FG_CELLS, counts, cellid);
    hipLaunchKernelGGL(fg_scan_kernel, dim3(B), dim3(1024), 0, st, gp, FG_CELLS, counts, start);
    hipLaunchKernelGGL(fg_fill_kernel, pg, dim3(256), 0, st, p2, len2, P2, FG_CELLS, cellid, start, counts, sorted);
    *gp_o = gp; *start_o = start; *sorted_o = sorted;
    return TPG_OK;
}

extern "C" int tpg_frnn_grid_f32(const float *p1, const float *p2, const int64_t *len1, const int64_t *len2, int B,
                                 int P1, int P2, int K, float r, float *dist, int64_t *idx, void *ws, void *stream) {
    if (B < 0 || P1 < 0 || P2 < 0 || K < 1 || K > 64 || !(r > 0.0f)) return TPG_ERR_ARG;
    if (B == 0 || P1 == 0) return TPG_OK;
    if (!p1 || !dist || !idx) return TPG_ERR_ARG;
    hipStream_t st = tpg_stream(stream);
    if (P2 == 0) return TPG_ERR_UNSUPPORTED;             // (the exhaustive entry pads an empty search)
    if (!p2 || !ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return TPG_ERR_ARG;
    GridParams *gp; int *start; float4 *sorted;
    const int rc = fg_build(p2, len2, B, P2, r, 0, ws, st, &gp, &start, &sorted);
    if (rc) return rc;
    const float r2 = r * r;     // fp32(r) * fp32(r), the value the exhaustive entry is given
    hipLaunchKernelGGL(fg_query_kernel, dim3((P1 + FG_WAVES - 1) / FG_WAVES, B), dim3(FG_WAVES * 64), 0, st, p1, len1, P1,
                       P2, gp, FG_CELLS, start, sorted, K, r2, dist, idx);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_knn_grid_f32(const float *p1, const float *p2, const int64_t *len1, const int64_t *len2, int B,
                                int P1, int P2, int K, float *dist, int64_t *idx, void *ws, void *stream) {
    if (B < 0 || P1 < 0 || P2 < 0 || K < 1 || K > 64) return TPG_ERR_ARG;
    if (B == 0 || P1 == 0) return TPG_OK;
    if (!p1 || !dist || !idx) return TPG_ERR_ARG;
    hipStream_t st = tpg_stream(stream);
    if (P2 == 0) return TPG_ERR_UNSUPPORTED;             // (the exhaustive entry pads an empty search)
    if (!p2 || !ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return TPG_ERR_ARG;
    GridParams *gp; int *start; float4 *sorted;
    const int rc = fg_build(p2, len2, B, P2, 0.0f, K, ws, st, &gp, &start, &sorted);
    if (rc) return rc;
    hipLaunchKernelGGL(fg_knn_kernel, dim3((P1 + FG_WAVES - 1) / FG_WAVES, B), dim3(FG_WAVES * 64), 0, st, p1, len1, P1,
                       P2, gp, FG_CELLS, start, sorted, K, dist, idx);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}
