// Feature-space kNN (D = 32 / 64) on LARGE clouds: Gram-matrix filter on the f32 matrix cores, exact
// re-ranking of the survivors.  Included by knn.hip (inside its anonymous namespace).
//
// The exhaustive kernels of knn.hip evaluate the canonical distance sum_d (q_d - p_d)^2 for every pair on
// the VALU (three instructions per pair and dimension): 4.6 ms per search on cfg5's (40, 4096, 32) clouds,
// 30 ms on a 65536-point rollout scene (upsampling_network.py:159-174) -- 97 % of a rollout frame.  The
// neighbour LIST is what must be bit-exact against the oracle, not the way candidates are ruled out:
//
//   approximate  d~(q,p) = |y_q|^2 + |y_p|^2 - 2 y_q.y_p ,  y = x - origin (the cloud's first point)
//
// comes out of v_mfma_f32_32x32x2_f32 (an exact k-ordered fmaf chain, MI355X_MICROARCH.md) at one matrix
// instruction per 32 x 32 pairs and 2 dimensions, and |d~ - d_canonical| <= E_q := c_D (|y_q| + max_p|y_p|)^2
// with c_D = (4 D + 32) 2^-24 (twice the sequential-rounding bound of both sums).  One workgroup owns 256
// queries (8 waves x 32; the queries stay in registers as the B operand) and sweeps the cloud TWICE through
// a double-buffered LDS tile of -2 y_p rows:
//   sweep 1  every lane (query c = lane & 31, rows of its half h = lane >> 5) keeps the 16 smallest of the
//            per-sub-tile minima of its 16 accumulator rows; tau_q = max over the two lanes of a query of
//            their 16th value => at least 32 points have d~ <= tau_q (16 distinct sub-tiles per lane);
//   sweep 2  the same products again; every pair with d~ <= tau_q appends its index to the query's LDS
//            list (64 slots; ~36 entries expected);
//   tail     one wave per query: canonical distance of the listed points (the same knn_dist as the
//            exhaustive kernels), rank by counting over the 64-bit (dist, idx) keys, the first K are the
//            answer IF  tau_q - E_q > d_K  (every unlisted point has d~ > tau_q, hence a canonical distance
//            above d_K: it cannot enter or tie).  Otherwise (duplicate-heavy clouds, list overflow, ragged
//            tails) the query's first index slot is set to -2 and knn_kernel<D, false, true> -- the
//            exhaustive kernel, one wave per flagged query -- redoes it.  The output is the exhaustive
//            kernels' output bit for bit in every case; only the time depends on the data.
#pragma once
#include <type_traits>

typedef float km_f32x16 __attribute__((ext_vector_type(16)));

#ifndef KM_WAVES_PER_WG
#define KM_WAVES_PER_WG 4
#endif
constexpr int KM_WAVES = KM_WAVES_PER_WG;   // waves per workgroup (4: two workgroups per CU, out of step with each other)
constexpr int KM_Q = KM_WAVES * 32;     // queries per workgroup
template <int D_T>
constexpr int km_tile_rows() { return D_T == 32 ? 128 : 64; }   // points per staged tile: two workgroups' LDS must fit a CU
constexpr int KM_CAP = 64;              // candidate slots per query
constexpr int KM_M = 16;                // sub-tile minima kept per lane
constexpr int KM_MAX_K = 24;            // K' = 2 KM_M = 32 guaranteed candidates must exceed K with room to spare
constexpr long long KM_REDO = KM_REDO_MARK;   // first index slot of a query the exhaustive kernel must redo

template <int CTRL>
__device__ __forceinline__ float km_dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

template <int D_T>
constexpr size_t km_smem_bytes() {
    constexpr int KM_TP = km_tile_rows<D_T>();
    return sizeof(float) * (2 * KM_TP * (D_T + 4) + 2 * KM_TP + D_T + KM_WAVES * D_T) + sizeof(int) * (KM_Q + KM_Q * KM_CAP + 2) +
           sizeof(tpg_u64) * KM_WAVES * 64;
}

template <int D_T>
__global__ __launch_bounds__(KM_WAVES * 64, 8 / KM_WAVES) void knn_mfma_kernel(
    const float *__restrict__ p1, const float *__restrict__ p2, const int64_t *__restrict__ len1,
    const int64_t *__restrict__ len2, int P1, int P2, int K, int gx, int total, float *__restrict__ dist,
    int64_t *__restrict__ idx) {
    constexpr int KM_TP = km_tile_rows<D_T>();
    constexpr int LD = D_T + 4;                    // row stride of the LDS tile: b128 reads of 16 rows hit 64 banks
    constexpr int V = D_T / 4;                     // float4 per row
    constexpr int RPP = KM_WAVES * 64 / V;         // rows staged per pass of the workgroup
    constexpr int NP = KM_TP / RPP;                // passes per tile
    constexpr int G = D_T / 8;                     // b128 operand reads per row and half
    static_assert(NP >= 1 && KM_TP % RPP == 0 && (V == 8 || V == 16), "tile staging shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char km_smem[];
    float *tile = reinterpret_cast<float *>(km_smem);                 // [2][KM_TP][LD]   -2 (x - origin)
    float *nrm = tile + 2 * KM_TP * LD;                               // [2][KM_TP]       |x - origin|^2, INF on padding rows
    float *org = nrm + 2 * KM_TP;                                     // [D_T]
    float *qrow = org + D_T;                                          // [KM_WAVES][D_T]  tail: the query as given
    tpg_u64 *keys = reinterpret_cast<tpg_u64 *>(qrow + KM_WAVES * D_T);   // [KM_WAVES][64]
    int *cnt = reinterpret_cast<int *>(keys + KM_WAVES * 64);         // [KM_Q]
    int *cbuf = cnt + KM_Q;                                           // [KM_Q][KM_CAP]
    unsigned *pmax = reinterpret_cast<unsigned *>(cbuf + KM_Q * KM_CAP);  // max_p |y_p|^2 (bits of a non-negative float)

    // workgroups of one cloud on one XCD (its L2 holds the cloud): linear id -> (xcd, slot) -> cloud-major order
    const int chunk = gridDim.x >> 3;
    const int w = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (w >= total) return;
    const int b = w / gx;
    const int q0 = (w - b * gx) * KM_Q;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int n1 = len1 ? (int)len1[b] : P1;
    const int n2 = len2 ? min((int)len2[b], P2) : P2;
    float *od = dist + (size_t)b * P1 * K;
    int64_t *oi = idx + (size_t)b * P1 * K;
    if (n2 < KM_CAP) {                             // nothing to filter: the exhaustive kernel handles (and pads) these
        for (int t = tid; t < KM_Q; t += KM_WAVES * 64)
            if (q0 + t < P1) oi[(size_t)(q0 + t) * K] = KM_REDO;
        return;
    }
    const float *cb = p2 + (size_t)b * P2 * D_T;
    for (int t = tid; t < KM_Q; t += KM_WAVES * 64) cnt[t] = 0;
    if (tid < D_T) org[tid] = cb[tid];
    if (tid == 0) *pmax = 0u;
    __syncthreads();

    // ---- this lane's query: dims 8g + 4h + i of query q0 + 32 wave + c, centred -------------------------
    const int qi = q0 + wave * 32 + c;
    float bq[D_T / 2];
    float nq = 0.0f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 o = *reinterpret_cast<const float4 *>(org + 8 * g + 4 * h);
        if (qi < P1) {
            x = *reinterpret_cast<const float4 *>(p1 + ((size_t)b * P1 + qi) * D_T + 8 * g + 4 * h);
            x.x -= o.x; x.y -= o.y; x.z -= o.z; x.w -= o.w;
        }
        bq[4 * g + 0] = x.x; bq[4 * g + 1] = x.y; bq[4 * g + 2] = x.z; bq[4 * g + 3] = x.w;
        nq += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
    nq += __shfl_xor(nq, 32);

    // ---- staging: thread (row sr + p RPP, float4 sv) of a tile -----------------------------------------
    const int sv = tid % V, sr = tid / V;
    const float4 o4 = *reinterpret_cast<const float4 *>(org + 4 * sv);
    float4 pre[NP];
    bool pv[NP];
    float mymax = 0.0f;
    auto prefetch = [&](int t) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int j = t * KM_TP + sr + p * RPP;
            pv[p] = j < n2;
            pre[p] = pv[p] ? *reinterpret_cast<const float4 *>(cb + (size_t)j * D_T + 4 * sv) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pv[p]) { y.x = pre[p].x - o4.x; y.y = pre[p].y - o4.y; y.z = pre[p].z - o4.z; y.w = pre[p].w - o4.w; }
            float s = y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
            // sum over the V = 8 / 16 consecutive lanes of the row: quad xor 1, xor 2, row_half_mirror[, row_mirror]
            s += km_dpp_f32<0xB1>(s);
            s += km_dpp_f32<0x4E>(s);
            s += km_dpp_f32<0x141>(s);
            if constexpr (V == 16) s += km_dpp_f32<0x140>(s);
            const int row = buf * KM_TP + sr + p * RPP;
            *reinterpret_cast<float4 *>(tile + row * LD + 4 * sv) = make_float4(-2.0f * y.x, -2.0f * y.y, -2.0f * y.z, -2.0f * y.w);
            if (sv == 0) nrm[row] = pv[p] ? s : INFINITY;
            if (pv[p]) mymax = fmaxf(mymax, s);
        }
    };

    float lst[KM_M];
#pragma unroll
    for (int i = 0; i < KM_M; ++i) lst[i] = INFINITY;
    const int ql = wave * 32 + c;                  // this lane's query within the workgroup
    const int NT = (n2 + KM_TP - 1) / KM_TP;

    // One sweep over the cloud.  Within a step the 32-row sub-tiles are software-pipelined: the VALU work on
    // sub-tile s-1's accumulators (minima / hit mask) is interleaved, three instructions per gap, between the 16
    // dependent matrix instructions of sub-tile s (each waits 64 cycles for its predecessor): a wave always has
    // a matrix instruction to offer, so the two waves of a SIMD keep the matrix pipe busy instead of running
    // their VALU phases side by side.
    constexpr int NS = KM_TP / 32;
    auto sweep = [&](auto second_tag, int step0, float tauf) {
        constexpr bool SECOND = decltype(second_tag)::value;
        for (int t = 0; t < NT; ++t) {
            const int cur = (step0 + t) & 1;
            const bool more = !SECOND || t + 1 < NT;
            if (more) prefetch(t + 1 == NT ? 0 : t + 1);
            km_f32x16 prev;
            unsigned hits = 0;
#pragma unroll
            for (int sub = 0; sub <= NS; ++sub) {
                km_f32x16 acc;
                if (sub < NS) {
                    const float *arow = tile + (cur * KM_TP + sub * 32 + c) * LD + 4 * h;
                    float4 a4[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) a4[g] = *reinterpret_cast<const float4 *>(arow + 8 * g);
#pragma unroll
                    for (int grp = 0; grp < 4; ++grp) {
                        const float4 n4 = *reinterpret_cast<const float4 *>(nrm + cur * KM_TP + sub * 32 + 8 * grp + 4 * h);
                        acc[4 * grp + 0] = n4.x + nq; acc[4 * grp + 1] = n4.y + nq;
                        acc[4 * grp + 2] = n4.z + nq; acc[4 * grp + 3] = n4.w + nq;
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].x, bq[4 * g + 0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].y, bq[4 * g + 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].z, bq[4 * g + 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[g].w, bq[4 * g + 3], acc, 0, 0, 0);
                    }
                }
                if (sub > 0) {
                    if constexpr (!SECOND) {
                        float v = fminf(fminf(prev[0], prev[1]), fminf(prev[2], prev[3]));
#pragma unroll
                        for (int r = 4; r < 16; r += 4) v = fminf(v, fminf(fminf(prev[r], prev[r + 1]), fminf(prev[r + 2], prev[r + 3])));
#pragma unroll
                        for (int i = 0; i < KM_M; ++i) {
                            const float lo = fminf(v, lst[i]);
                            v = fmaxf(v, lst[i]);
                            lst[i] = lo;
                        }
                    } else {
                        hits = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) hits |= prev[r] <= tauf ? 1u << r : 0u;
                    }
                }
                if (sub < NS && sub > 0) {
#pragma unroll
                    for (int m = 0; m < D_T / 2; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one matrix instruction
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // three VALU instructions of the sub-tile before
                    }
                }
                if constexpr (SECOND) {
                    if (sub > 0 && __ballot(hits != 0)) {
                        if (hits) {
                            int pos = atomicAdd(&cnt[ql], __popc(hits));
                            const int base = t * KM_TP + (sub - 1) * 32 + 4 * h;
                            while (hits) {
                                const int r = __builtin_ctz(hits);
                                hits &= hits - 1;
                                if (pos < KM_CAP) cbuf[ql * KM_CAP + pos] = base + 8 * (r >> 2) + (r & 3);
                                ++pos;
                            }
                        }
                    }
                }
                prev = acc;
            }
            if (more) commit(cur ^ 1);
            __syncthreads();
        }
    };

    prefetch(0);
    commit(0);
    __syncthreads();
    sweep(std::false_type{}, 0, 0.0f);
    const float tau = fmaxf(lst[KM_M - 1], __shfl_xor(lst[KM_M - 1], 32));
    sweep(std::true_type{}, NT, fminf(tau, 3.0e38f));     // finite: padding rows carry d~ = INF and never hit
    if (sv == 0) atomicMax(pmax, __float_as_uint(mymax));
    __syncthreads();

    // ---- tail: exact re-ranking, one wave per query ------------------------------------------------------
    const float sq_pmax = sqrtf(__uint_as_float(*pmax));
    constexpr float CD = (float)(4 * D_T + 32) * 5.9604644775390625e-8f;       // (4 D + 32) 2^-24
    float *qs = qrow + wave * D_T;
    tpg_u64 *ks = keys + wave * 64;
    const tpg_u64 INF = ~0ull;
    for (int qq = 0; qq < 32; ++qq) {
        const int i = q0 + wave * 32 + qq;
        if (i >= P1) break;
        if (i >= n1) {
            if (lane < K) { od[(size_t)i * K + lane] = 0.0f; oi[(size_t)i * K + lane] = 0; }
            continue;
        }
        const float tq = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(tau), qq));
        const float sq = sqrtf(__uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(nq), qq))) + sq_pmax;
        const float eq = CD * sq * sq;
        const int n = cnt[wave * 32 + qq];
        if (lane < D_T) qs[lane] = p1[((size_t)b * P1 + i) * D_T + lane];
        tpg_u64 key = INF;
        if (n <= KM_CAP && lane < n) {
            const int j = cbuf[(wave * 32 + qq) * KM_CAP + lane];
            key = knn_pack(knn_dist<D_T>(qs, cb + (size_t)j * D_T, D_T), j);
        }
        ks[lane] = key;
        int rank = 0;
        for (int s = 0; s < 64; ++s) rank += ks[s] < key ? 1 : 0;
        const tpg_u64 kth = __ballot(rank == K - 1 && key != INF);
        bool ok = n <= KM_CAP && kth != 0;
        float dk = -1.0f;
        if (ok) {
            dk = __uint_as_float((unsigned)(tpg_readlane_u64(key, __builtin_amdgcn_readfirstlane(__builtin_ctzll(kth))) >> 32));
            ok = tq - eq > dk;
        }
#ifdef KM_DEBUG
        if (!ok && lane == 0) { od[(size_t)i * K] = (float)n; od[(size_t)i * K + 1] = tq; od[(size_t)i * K + 2] = eq; od[(size_t)i * K + 3] = dk; }
#endif
        if (ok) {
            if (rank < K && key != INF) {
                od[(size_t)i * K + rank] = __uint_as_float((unsigned)(key >> 32));
                oi[(size_t)i * K + rank] = (long long)(unsigned)key;
            }
        } else if (lane == 0) {
            oi[(size_t)i * K] = KM_REDO;
        }
    }
}
