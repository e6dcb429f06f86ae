// K nearest neighbours / fixed-radius neighbours / Chamfer nn-search for gfx950.
//
// Replaces pytorch3d.ops.knn_points (reference gcn_lib/pointnet/gcn.py:16-21,38,258;
// discriminator.py:15-20,33-38), frnn.frnn_grid_points (discriminator.py:27-32;
// loss.py:256-265) and the nn-search of chamferdist.ChamferDistance
// (loss.py:125-127,176-181).
//
// Design (one wave64 per query, no LDS for the neighbour list):
//   * the 64 lanes of a wave test 64 consecutive candidates per step; each lane
//     reads its own candidate row (D contiguous floats, 16-B vector loads) while
//     the query row is broadcast from LDS;
//   * a candidate is the 64-bit key (dist_bits << 32 | idx): keys are unique
//     and their unsigned order IS the canonical (dist, idx) order;
//   * the running K best keys live one per lane, sorted ascending across the
//     wave (K <= 64).  A candidate that beats the K-th key is inserted with a
//     ballot (rank = popcount of lanes holding a smaller key) and a one-lane
//     shift -- no shared memory, no sorting network, no per-thread heap;
//   * after warm-up almost every 64-candidate step fails the threshold test
//     with one v_cmp + ballot, so the loop runs at the distance-evaluation rate;
//   * K == 1 (Chamfer, masking loss) never leaves the lane: per-lane running
//     minimum, one DPP wave reduction at the end.
// The kernel is HBM/L2-light (each cloud is a few KB..1.5 MB and stays in L2);
// its bound is VALU issue + cross-lane latency, see DESIGN.md.
#include <cstdlib>
#include <type_traits>

#include <hip/hip_bf16.h>

#include "invert_index.hpp"
#include "knn_select.hpp"
#include "tpg_common.hpp"

namespace {

constexpr int KNN_WAVES = 4;  // waves (= queries) per workgroup

__device__ __forceinline__ tpg_u64 knn_pack(float d, int j) {
    return ((tpg_u64)__float_as_uint(d) << 32) | (unsigned)j;
}

// distance between the wave's query (LDS, broadcast reads) and this lane's row.
template <int D_T>
__device__ __forceinline__ float knn_dist(const float *__restrict__ qs,
                                          const float *__restrict__ c, int D) {
    float acc = 0.0f;
    if constexpr (D_T == 3) {
        const float t0 = qs[0] - c[0], t1 = qs[1] - c[1], t2 = qs[2] - c[2];
        acc = t0 * t0;
        acc = acc + t1 * t1;
        acc = acc + t2 * t2;
    } else if constexpr (D_T > 0) {
        static_assert(D_T % 4 == 0, "vector path needs D % 4 == 0");
        const float4 *c4 = reinterpret_cast<const float4 *>(c);
        const float4 *q4 = reinterpret_cast<const float4 *>(qs);
#pragma unroll
        for (int d = 0; d < D_T / 4; ++d) {
            const float4 cv = c4[d];
            const float4 qv = q4[d];
            float t;
            t = qv.x - cv.x; acc = acc + t * t;
            t = qv.y - cv.y; acc = acc + t * t;
            t = qv.z - cv.z; acc = acc + t * t;
            t = qv.w - cv.w; acc = acc + t * t;
        }
    } else {
        for (int d = 0; d < D; ++d) {
            const float t = qs[d] - c[d];
            acc = acc + t * t;
        }
    }
    return acc;
}

constexpr long long KM_REDO_MARK = -2;   // knn_mfma.hpp: first index slot of a query the filter could not settle

template <int D_T, bool RADIUS>
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_kernel(
    const float *__restrict__ p1, const float *__restrict__ p2,
    const int64_t *__restrict__ len1, const int64_t *__restrict__ len2, int B, int P1, int P2,
    int D, int K, float r2, float *__restrict__ dist, int64_t *__restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) float knn_q[];  // [KNN_WAVES][Dpad]
    __shared__ tpg_u64 knn_slots[KNN_WAVES * 64];                   // rank-merge scratch, one slot row per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Dpad = (D + 3) & ~3;
    float *qs = knn_q + wave * Dpad;
    tpg_u64 *slot = knn_slots + wave * 64;

    const int b = blockIdx.y;
    const int i = blockIdx.x * KNN_WAVES + wave;
    const bool valid = i < P1;
    const size_t q = (size_t)b * P1 + (valid ? i : 0);
    const int n1 = len1 ? (int)len1[b] : P1;
    const int n2 = len2 ? min((int)len2[b], P2) : P2;

    // stage the query row (each wave its own slice; the barrier orders LDS)
    if (valid) {
        const float *qrow = p1 + q * D;
        for (int d = lane; d < D; d += 64) qs[d] = qrow[d];
    }
    __syncthreads();
    if (!valid) return;

    const float pad_d = RADIUS ? -1.0f : 0.0f;
    const long long pad_i = RADIUS ? -1 : 0;
    float *od = dist + q * K;
    int64_t *oi = idx + q * K;
    if (i >= n1 || n2 <= 0) {
        for (int k = lane; k < K; k += 64) { od[k] = pad_d; oi[k] = pad_i; }
        return;
    }

    const float *cbase = p2 + (size_t)b * P2 * D;
    const tpg_u64 INF = ~0ull;

    if (K == 1) {
        tpg_u64 best = INF;
        for (int j = lane; j < n2; j += 64) {
            const float d = knn_dist<D_T>(qs, cbase + (size_t)j * D, D);
            if (!RADIUS || d < r2) {
                const tpg_u64 key = knn_pack(d, j);
                best = key < best ? key : best;
            }
        }
        best = tpg_wave_min_u64(best);
        if (lane == 0) {
            if (best == INF) { od[0] = pad_d; oi[0] = pad_i; }
            else { od[0] = __uint_as_float((unsigned)(best >> 32)); oi[0] = (long long)(unsigned)best; }
        }
        return;
    }

    tpg_u64 best = INF;  // lane l: l-th smallest key so far
    tpg_u64 thr = INF;   // wave-uniform: key of rank K-1
    for (int base = 0; base < n2; base += 64) {
        const int j = base + lane;
        tpg_u64 key = INF;
        if (j < n2) {
            const float d = knn_dist<D_T>(qs, cbase + (size_t)j * D, D);
            if (!RADIUS || d < r2) key = knn_pack(d, j);
        }
        tpg_knn_merge(best, thr, key, K, lane, slot);
    }
    if (lane < K) {
        if (best == INF) { od[lane] = pad_d; oi[lane] = pad_i; }
        else {
            od[lane] = __uint_as_float((unsigned)(best >> 32));
            oi[lane] = (long long)(unsigned)best;
        }
    }
}

// ---- feature-space kNN (D = 32 / 64): a wave keeps its 64 candidate rows in REGISTERS and scores
// them against Q queries (broadcast from LDS) before moving on.  In the one-query-per-wave kernel
// above every query re-reads the whole cloud (P2 * D * 4 B: 128 KB at D = 64), i.e. 1.6 GB of
// L1/L2 traffic for the generator's (24, 512, 64) searches (9 TB/s: the L2 rate) -- that, not
// the arithmetic or the list updates, was its 177 us.  Scoring Q queries per candidate tile
// divides the traffic by Q; the Q neighbour lists stay one key per lane each (2Q VGPRs).  Same
// keys, same total order => the same neighbours as the kernel above, bit for bit.  Q is small
// (2-4): each wave's tiles are a chain of load -> score, and more queries per wave means fewer
// waves to overlap those chains.
template <int D_T, int Q, bool RADIUS>
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_tile_kernel(
    const float *__restrict__ p1, const float *__restrict__ p2, const int64_t *__restrict__ len1,
    const int64_t *__restrict__ len2, int P1, int P2, int K, float r2, float *__restrict__ dist,
    int64_t *__restrict__ idx) {
    static_assert(D_T % 4 == 0, "rows are read as float4");
    __shared__ __attribute__((aligned(16))) float qsm[KNN_WAVES * Q * D_T];
    __shared__ tpg_u64 knn_slots[KNN_WAVES * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    tpg_u64 *slot = knn_slots + wave * 64;
    const int b = blockIdx.y;
    const int q0 = (blockIdx.x * KNN_WAVES + wave) * Q;        // this wave's first query
    float *qs = qsm + wave * Q * D_T;
    for (int t = lane; t < Q * D_T; t += 64) {
        const int qq = t / D_T, d = t - qq * D_T;
        const int i = q0 + qq;
        qs[t] = i < P1 ? p1[((size_t)b * P1 + i) * D_T + d] : 0.0f;
    }
    __syncthreads();
    const int n1 = len1 ? (int)len1[b] : P1;
    const int n2 = len2 ? min((int)len2[b], P2) : P2;
    const tpg_u64 INF = ~0ull;
    tpg_u64 best[Q], thr[Q];
#pragma unroll
    for (int qq = 0; qq < Q; ++qq) { best[qq] = INF; thr[qq] = INF; }
    const float *cbase = p2 + (size_t)b * P2 * D_T;
    for (int base = 0; base < n2; base += 64) {
        const int j = base + lane;
        float4 row[D_T / 4];
        if (j < n2) {
            const float4 *c4 = reinterpret_cast<const float4 *>(cbase + (size_t)j * D_T);
#pragma unroll
            for (int d = 0; d < D_T / 4; ++d) row[d] = c4[d];
        } else {
#pragma unroll
            for (int d = 0; d < D_T / 4; ++d) row[d] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int qq = 0; qq < Q; ++qq) {
            // compiler barrier: the query rows are loop-invariant, and without it LICM hoists all
            // Q*D of them out of the tile loop into registers (512 VGPRs + scratch); with it each
            // query is re-read from LDS (broadcast ds_read_b128) right before it is used
            asm volatile("" ::: "memory");
            const float4 *q4 = reinterpret_cast<const float4 *>(qs + qq * D_T);
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < D_T / 4; ++d) {
                const float4 qv = q4[d];
                float t;
                t = qv.x - row[d].x; acc = acc + t * t;
                t = qv.y - row[d].y; acc = acc + t * t;
                t = qv.z - row[d].z; acc = acc + t * t;
                t = qv.w - row[d].w; acc = acc + t * t;
            }
            tpg_u64 key = INF;
            if (j < n2 && (!RADIUS || acc < r2)) key = knn_pack(acc, j);
            tpg_knn_merge(best[qq], thr[qq], key, K, lane, slot);
        }
    }
    const float pad_d = RADIUS ? -1.0f : 0.0f;
    const long long pad_i = RADIUS ? -1 : 0;
#pragma unroll
    for (int qq = 0; qq < Q; ++qq) {
        const int i = q0 + qq;
        if (i >= P1 || lane >= K) continue;
        const size_t o = ((size_t)b * P1 + i) * K + lane;
        if (i >= n1 || n2 <= 0 || best[qq] == INF) { dist[o] = pad_d; idx[o] = pad_i; }
        else { dist[o] = __uint_as_float((unsigned)(best[qq] >> 32)); idx[o] = (long long)(unsigned)best[qq]; }
    }
}

// The queries the matrix-core filter left open (first index slot == KM_REDO_MARK), exhaustively: a workgroup looks
// at 4 consecutive queries and gives EVERY open one all four of its waves -- each wave scans a quarter of the cloud
// into its own K-best list, wave 0 merges the four lists (4 x 64 keys through the same rank merge).  Open queries
// are a handful per cloud and sit in different workgroups: the launch lasts one quarter-scan, not one scan.
template <int D_T>
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_redo_kernel(
    const float *__restrict__ p1, const float *__restrict__ p2, const int64_t *__restrict__ len1,
    const int64_t *__restrict__ len2, int P1, int P2, int K, float *__restrict__ dist, int64_t *__restrict__ idx) {
    __shared__ __attribute__((aligned(16))) float qs[D_T];
    __shared__ tpg_u64 slots[KNN_WAVES * 64];
    __shared__ tpg_u64 lists[KNN_WAVES * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int n1 = len1 ? (int)len1[b] : P1;
    const int n2 = len2 ? min((int)len2[b], P2) : P2;
    const float *cbase = p2 + (size_t)b * P2 * D_T;
    const tpg_u64 INF = ~0ull;
    for (int u = 0; u < KNN_WAVES; ++u) {
        const int i = blockIdx.x * KNN_WAVES + u;
        if (i >= P1) break;                                            // (uniform over the workgroup)
        const size_t q = (size_t)b * P1 + i;
        if (idx[q * K] != KM_REDO_MARK) continue;                      // (uniform: every thread reads the same slot)
        __syncthreads();                                               // the previous query's LDS is free
        if (threadIdx.x < D_T) qs[threadIdx.x] = p1[q * D_T + threadIdx.x];
        __syncthreads();
        tpg_u64 best = INF, thr = INF;
        if (i < n1)
            for (int base = wave * 64; base < n2; base += KNN_WAVES * 64) {
                const int j = base + lane;
                const tpg_u64 key = j < n2 ? knn_pack(knn_dist<D_T>(qs, cbase + (size_t)j * D_T, D_T), j) : INF;
                tpg_knn_merge(best, thr, key, K, lane, slots + wave * 64);
            }
        lists[wave * 64 + lane] = lane < K ? best : INF;
        __syncthreads();
        if (wave == 0) {
            best = INF;
            thr = INF;
            for (int w = 0; w < KNN_WAVES; ++w) tpg_knn_merge(best, thr, lists[w * 64 + lane], K, lane, slots);
            if (lane < K) {
                if (best == INF) { dist[q * K + lane] = 0.0f; idx[q * K + lane] = 0; }
                else { dist[q * K + lane] = __uint_as_float((unsigned)(best >> 32)); idx[q * K + lane] = (long long)(unsigned)best; }
            }
        }
    }
}

#include "knn_mfma.hpp"

#ifndef TPG_KNN_MFMA_MIN_POINTS
#define TPG_KNN_MFMA_MIN_POINTS 2048    // clouds from which the matrix-core filter beats the tiled exhaustive kernel
#endif

// the filter kernel, then (redo) the exhaustive kernel on the queries it marked
template <int D_T, int M>
int knn_mfma_launch_m(const float *p1, const float *p2, const int64_t *len1, const int64_t *len2, int B, int P1, int P2,
                      int K, float *dist, int64_t *idx, bool redo, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_mfma_kernel<D_T, M>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)km_smem_bytes<D_T>()) != hipSuccess)
            return TPG_ERR_LAUNCH;
        attr_set = true;
    }
    const int gx = (P1 + KM_Q - 1) / KM_Q;
    const long long total = (long long)gx * B;
    if (total > (1ll << 30)) return TPG_ERR_ARG;
    const unsigned grid = (unsigned)((total + 7) / 8 * 8);
    hipLaunchKernelGGL((knn_mfma_kernel<D_T, M>), dim3(grid), dim3(KM_WAVES * 64), km_smem_bytes<D_T>(), st, p1, p2, len1, len2,
                       P1, P2, K, gx, (int)total, dist, idx);
    TPG_RETURN_IF_LAUNCH_FAILED();
    if (redo) {
        const dim3 rg((unsigned)((P1 + KNN_WAVES - 1) / KNN_WAVES), (unsigned)B);
        hipLaunchKernelGGL((knn_redo_kernel<D_T>), rg, dim3(KNN_WAVES * 64), 0, st, p1, p2, len1, len2, P1, P2, K, dist, idx);
        TPG_RETURN_IF_LAUNCH_FAILED();
    }
    return TPG_OK;
}

// minima kept per lane: 2 M points are guaranteed below the threshold, ~2.4 M expected; K + 8 <= 2 M keeps the
// verification margin, a small M keeps the candidate list (64 slots) from overflowing and the insert chain short
template <int D_T>
int knn_mfma_launch(const float *p1, const float *p2, const int64_t *len1, const int64_t *len2, int B, int P1, int P2,
                    int K, float *dist, int64_t *idx, bool redo, hipStream_t st) {
    // (from 16384 points the gap between the K-th and the 2M-th neighbour is what has to beat the rounding bound
    // on low-dimensional clouds: the longest list, whatever K)
    if (K <= 12 && P2 < 16384) return knn_mfma_launch_m<D_T, 10>(p1, p2, len1, len2, B, P1, P2, K, dist, idx, redo, st);
    if (K <= 20 && P2 < 16384) return knn_mfma_launch_m<D_T, 14>(p1, p2, len1, len2, B, P1, P2, K, dist, idx, redo, st);
    return knn_mfma_launch_m<D_T, 16>(p1, p2, len1, len2, B, P1, P2, K, dist, idx, redo, st);
}

template <bool RADIUS>
int knn_launch(const float *p1, const float *p2, const int64_t *len1, const int64_t *len2, int B,
               int P1, int P2, int D, int K, float r2, float *dist, int64_t *idx,
               hipStream_t st) {
    if (B == 0 || P1 == 0) return TPG_OK;
    if (B > 65535) return TPG_ERR_ARG;
    const dim3 grid((unsigned)((P1 + KNN_WAVES - 1) / KNN_WAVES), (unsigned)B), block(KNN_WAVES * 64);
    const size_t smem = sizeof(float) * KNN_WAVES * ((D + 3) & ~3);
    if (smem > 64 * 1024) return TPG_ERR_UNSUPPORTED;
    // the vector path also needs 16-B aligned rows: D % 4 == 0 and aligned bases
    const bool al = ((reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15) == 0;
#ifndef TPG_KNN_NO_MFMA
    // large feature-space clouds: Gram filter on the f32 matrix cores + exact re-ranking (knn_mfma.hpp)
    static const bool mfma_on = [] { const char *e = getenv("TPG_KNN_MFMA"); return !(e && e[0] == '0'); }();   // A/B switch
    if (mfma_on && !RADIUS && al && K > 1 && K <= KM_MAX_K && (D == 32 || D == 64) && P2 >= TPG_KNN_MFMA_MIN_POINTS) {
        if (D == 32) return knn_mfma_launch<32>(p1, p2, len1, len2, B, P1, P2, K, dist, idx, true, st);
        return knn_mfma_launch<64>(p1, p2, len1, len2, B, P1, P2, K, dist, idx, true, st);
    }
#endif
    // feature-space searches with several neighbours: the query-tiled kernel
#ifndef TPG_KNN_NO_TILE
    // feature-space searches: the query-tiled kernel.  Measured on (24,512,D) clouds
    // (tools/tune_knn.py): D = 64: Q = 4 is 2.1x faster than one query per wave (177 -> 85 us),
    // D = 32: Q = 2 is 1.3-1.7x faster up to K ~ 20; larger Q serialises too much work per wave.
#ifdef TPG_KNN_TILE_Q            // sweep builds force one Q
    constexpr int Q64 = TPG_KNN_TILE_Q, Q32 = TPG_KNN_TILE_Q;
    const bool tile32 = true;
#else
    constexpr int Q64 = 4, Q32 = 2;
    const bool tile32 = K <= 24;
#endif
    if (al && K > 1 && ((D == 64 && P1 >= KNN_WAVES * Q64) || (D == 32 && tile32 && P1 >= KNN_WAVES * Q32))) {
        if (D == 32) {
            const dim3 tg((unsigned)((P1 + KNN_WAVES * Q32 - 1) / (KNN_WAVES * Q32)), (unsigned)B);
            hipLaunchKernelGGL((knn_tile_kernel<32, Q32, RADIUS>), tg, block, 0, st, p1, p2, len1, len2, P1, P2, K, r2,
                               dist, idx);
        } else {
            const dim3 tg((unsigned)((P1 + KNN_WAVES * Q64 - 1) / (KNN_WAVES * Q64)), (unsigned)B);
            hipLaunchKernelGGL((knn_tile_kernel<64, Q64, RADIUS>), tg, block, 0, st, p1, p2, len1, len2, P1, P2, K, r2,
                               dist, idx);
        }
        TPG_RETURN_IF_LAUNCH_FAILED();
        return TPG_OK;
    }
#endif
#define TPG_KNN_GO(DT) \
    hipLaunchKernelGGL((knn_kernel<DT, RADIUS>), grid, block, smem, st, p1, p2, len1, len2, B, P1, \
                       P2, D, K, r2, dist, idx)
    if (D == 3) TPG_KNN_GO(3);
    else if (D == 32 && al) TPG_KNN_GO(32);
    else if (D == 64 && al) TPG_KNN_GO(64);
    else TPG_KNN_GO(0);
#undef TPG_KNN_GO
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

// Gradients of both Chamfer directions WITHOUT float atomics (round 3): every output point is written once.
//   out[i] = the point's own term 2 g[i] (a_i - b_nn[i])  and  - sum over the entries j of the inverted
//   index of the OTHER direction (nn2[j] == i, ascending j) of 2 g2[j] (b_j - a_i)
// in the order the oracle's two loops produce them (oracle/tpgref.c tpgref_chamfer_bwd_f32: source points take
// their own term first, target points take it last), so the result is bit-identical to it.
template <bool OWN_FIRST>
__global__ __launch_bounds__(256) void chamfer_bwd_gather_kernel(
    const float *__restrict__ a, const float *__restrict__ bcloud, int B, int N, int M,
    const int64_t *__restrict__ nn, const float *__restrict__ g, const int32_t *__restrict__ offs,
    const int32_t *__restrict__ list, const float *__restrict__ g2, float *__restrict__ ga) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)B * N) return;
    const int b = (int)(t / N), i = (int)(t - (long long)b * N);
    long long j = nn[t];
    j = j < 0 ? 0 : (j >= M ? M - 1 : j);
    const float gg = 2.0f * g[t];
    const float *pa = a + (size_t)t * 3;
    const float *pb = bcloud + ((size_t)b * M + j) * 3;
    const float ax = pa[0], ay = pa[1], az = pa[2];
    float acc[3] = {0.0f, 0.0f, 0.0f};
    const float own[3] = {gg * (ax - pb[0]), gg * (ay - pb[1]), gg * (az - pb[2])};
    if (OWN_FIRST) { acc[0] += own[0]; acc[1] += own[1]; acc[2] += own[2]; }
    const int32_t *of = offs + (size_t)b * (N + 1);
    const int32_t *ls = list + (size_t)b * M;
    const int p1 = of[i + 1];
    for (int p = of[i]; p < p1; ++p) {
        const int e = ls[p];                       // a point of the other cloud whose nearest neighbour is i
        const float h = 2.0f * g2[(size_t)b * M + e];
        const float *q = bcloud + ((size_t)b * M + e) * 3;
        acc[0] -= h * (q[0] - ax);
        acc[1] -= h * (q[1] - ay);
        acc[2] -= h * (q[2] - az);
    }
    if (!OWN_FIRST) { acc[0] += own[0]; acc[1] += own[1]; acc[2] += own[2]; }
    float *oa = ga + (size_t)t * 3;
    oa[0] = acc[0]; oa[1] = acc[1]; oa[2] = acc[2];
}

}  // namespace

extern "C" int tpg_knn_f32(const float *p1, const float *p2, const int64_t *len1,
                           const int64_t *len2, int B, int P1, int P2, int D, int K, float r2,
                           float *dist, int64_t *idx, void *stream) {
    if (B < 0 || P1 < 0 || P2 < 0 || D <= 0 || K <= 0) return TPG_ERR_ARG;
    if (K > TPG_MAX_K) return TPG_ERR_UNSUPPORTED;
    if ((long long)B * P1 == 0) return TPG_OK;
    if (!p1 || !dist || !idx || (P2 > 0 && !p2)) return TPG_ERR_ARG;
    if (r2 >= 0.0f)
        return knn_launch<true>(p1, p2, len1, len2, B, P1, P2, D, K, r2, dist, idx, tpg_stream(stream));
    return knn_launch<false>(p1, p2, len1, len2, B, P1, P2, D, K, r2, dist, idx, tpg_stream(stream));
}

extern "C" int tpg_knn_mfma_f32(const float *p1, const float *p2, const int64_t *len1, const int64_t *len2, int B, int P1,
                                int P2, int D, int K, float *dist, int64_t *idx, int redo, void *stream) {
    if (B < 0 || P1 < 0 || P2 < 0 || K <= 1) return TPG_ERR_ARG;
    if ((D != 32 && D != 64) || K > KM_MAX_K) return TPG_ERR_UNSUPPORTED;
    if ((long long)B * P1 == 0) return TPG_OK;
    if (!p1 || !p2 || !dist || !idx || B > 65535) return TPG_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15) return TPG_ERR_ARG;
    if (D == 32) return knn_mfma_launch<32>(p1, p2, len1, len2, B, P1, P2, K, dist, idx, redo != 0, tpg_stream(stream));
    return knn_mfma_launch<64>(p1, p2, len1, len2, B, P1, P2, K, dist, idx, redo != 0, tpg_stream(stream));
}

extern "C" int tpg_chamfer_fwd_f32(const float *src, const float *tgt, int B, int N, int M,
                                   float *d1, int64_t *i1, float *d2, int64_t *i2, void *stream) {
    if (B < 0 || N <= 0 || M <= 0) return TPG_ERR_ARG;
    if (B == 0) return TPG_OK;
    if (!src || !tgt || !d1 || !i1 || !d2 || !i2) return TPG_ERR_ARG;
    int rc = knn_launch<false>(src, tgt, nullptr, nullptr, B, N, M, 3, 1, -1.0f, d1, i1, tpg_stream(stream));
    if (rc) return rc;
    return knn_launch<false>(tgt, src, nullptr, nullptr, B, M, N, 3, 1, -1.0f, d2, i2, tpg_stream(stream));
}

extern "C" size_t tpg_chamfer_bwd_workspace_bytes(int B, int N, int M) {
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    // per direction: offs (B, dest + 1) + list (B, entries) + radix scratch (B, entries)
    return sizeof(int32_t) * ((size_t)B * ((size_t)N + 1 + 2 * (size_t)M) + (size_t)B * ((size_t)M + 1 + 2 * (size_t)N)) + 64;
}

extern "C" int tpg_chamfer_bwd_f32(const float *src, const float *tgt, int B, int N, int M,
                                   const int64_t *i1, const int64_t *i2, const float *g1,
                                   const float *g2, float *gsrc, float *gtgt, void *ws, void *stream) {
    if (B < 0 || N <= 0 || M <= 0) return TPG_ERR_ARG;
    if (B == 0) return TPG_OK;
    if (!src || !tgt || !i1 || !i2 || !g1 || !g2 || !gsrc || !gtgt || !ws) return TPG_ERR_ARG;
    hipStream_t st = tpg_stream(stream);
    // inverted nearest-neighbour indices: source point <- the target points that chose it (i2), and vice versa
    int32_t *offs_s = static_cast<int32_t *>(ws);
    int32_t *list_s = offs_s + (size_t)B * (N + 1);
    int32_t *tmp_s = list_s + (size_t)B * M;
    int32_t *offs_t = tmp_s + (size_t)B * M;
    int32_t *list_t = offs_t + (size_t)B * (M + 1);
    int32_t *tmp_t = list_t + (size_t)B * N;
    int rc = tpg_inv::launch<int64_t>(i2, B, N, M, offs_s, list_s, tmp_s, st);
    if (rc) return rc;
    rc = tpg_inv::launch<int64_t>(i1, B, M, N, offs_t, list_t, tmp_t, st);
    if (rc) return rc;
    const long long t1 = (long long)B * N, t2 = (long long)B * M;
    hipLaunchKernelGGL(chamfer_bwd_gather_kernel<true>, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, st, src,
                       tgt, B, N, M, i1, g1, offs_s, list_s, g2, gsrc);
    hipLaunchKernelGGL(chamfer_bwd_gather_kernel<false>, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, st, tgt,
                       src, B, M, N, i2, g2, offs_t, list_t, g1, gtgt);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}
