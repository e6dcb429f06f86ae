// Feature-space kNN (D = 32 / 64) on LARGE clouds: Gram-matrix filter on the bf16 matrix cores, exact
// re-ranking of the survivors.  Included by knn.hip (inside its anonymous namespace).
//
// The exhaustive kernels of knn.hip evaluate the canonical distance sum_d (q_d - p_d)^2 for every pair on
// the VALU (three instructions per pair and dimension): 4.6 ms per search on cfg5's (40, 4096, 32) clouds,
// 30 ms on a 65536-point rollout scene (upsampling_network.py:159-174) -- 97 % of a rollout frame.  The
// neighbour LIST is what must be bit-exact against the oracle, not the way candidates are ruled out:
//
//   approximate  d~(q,p) = |y_q|^2 + |y_p|^2 - 2 y_q.y_p ,  y = x - origin (the mean of the cloud's first tile of points)
//
// with the product on v_mfma_f32_32x32x16_bf16 from SPLIT operands: t = hi + lo + r, hi = bf16(t),
// lo = bf16(t - hi), |r| <= 2^-16 |t|; hi.hi + hi.lo + lo.hi leaves out lo.lo and the two residual terms,
// together <= 6.06 2^-16 |y_q||y_p|.  (A first form on v_mfma_f32_32x32x2_f32 -- exact fp32, five times tighter --
// was 2-3x slower: the f32 matrix instruction runs at the vector rate and does NOT co-execute with the VALU work
// of the selection, SQ_VALU_MFMA_COEXEC_CYCLES = 0; the bf16 one costs 7 x 32 cycles per 32 x 32 pairs beside it.)
// |y_p|^2 enters as a fourth product: three bf16 terms (exact) in the spare words of the LDS row against (1, 1, 1).
//
//   |d~ - d_canonical| <= E_q := 2^-13 |y_q| max_p|y_p| + (8 D + 64) 2^-24 (|y_q| + max_p|y_p|)^2
//
// One workgroup owns 128 queries (4 waves x 32; the split queries stay in registers as the B operand), two
// workgroups per CU, and sweeps the cloud TWICE through a double-buffered LDS tile of split -2 y_p rows:
//   sweep 1  every lane (query c = lane & 31, rows of its half h = lane >> 5) keeps the M smallest of the
//            minima of its 16 (32 from 4096 points) accumulator rows per entry; tau_q = max over the two lanes of a
//            query of their M-th value => at least 2 M points have d~ <= tau_q (M distinct row groups per lane);
//            M = 10 / 14 / 16 by K (K + 8 <= 2 M), 16 from 16384 points;
//   sweep 2  the same products again; every pair with d~ <= tau_q appends its index to the query's LDS
//            list (64 slots; ~2.4 M entries expected);
//   tail     one wave per query: canonical distance of the listed points (the arithmetic of knn_dist), rank
//            by counting over the 64-bit (dist, idx) keys, the first K are the
//            answer IF  tau_q - E_q > d_K  (every unlisted point has d~ > tau_q, hence a canonical distance
//            above d_K: it cannot enter or tie).  Otherwise (list overflow, ragged tails, and clouds whose
//            neighbour spacing is below the rounding bound: thousands of points on a 2-D sheet in feature space)
//            the query's first index slot is set to -2 and knn_redo_kernel (knn.hip: the
//            exhaustive scan, four waves per flagged query) redoes it.  The output is the exhaustive
//            kernels' output bit for bit in every case; only the time depends on the data.
#pragma once

typedef float km_f32x16 __attribute__((ext_vector_type(16)));

#ifndef KM_WAVES_PER_WG
#define KM_WAVES_PER_WG 4
#endif
constexpr int KM_WAVES = KM_WAVES_PER_WG;   // waves per workgroup (4: two workgroups per CU, out of step with each other)
constexpr int KM_Q = KM_WAVES * 32;     // queries per workgroup
template <int D_T>
constexpr int km_tile_rows() { return D_T == 32 ? 128 : 64; }   // points per staged tile: two workgroups' LDS must fit a CU
constexpr int KM_CAP = 64;              // candidate slots per query
constexpr float KM_BIG = 1.0e30f;       // the norm of a padding row / an empty list entry: finite (no 0 x INF in the matrix unit), never a hit
constexpr int KM_MIN_VALID = 1024;      // fewer valid points than this in a cloud: no filter (32 entries per lane are the floor)
constexpr int KM_MAX_K = 24;            // 2 M guaranteed candidates (M = 10 / 14 / 16 minima kept per lane) must exceed K with room to spare
constexpr long long KM_REDO = KM_REDO_MARK;   // first index slot of a query the exhaustive kernel must redo

template <int CTRL>
__device__ __forceinline__ float km_dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

template <int D_T>
constexpr size_t km_smem_bytes() {
    constexpr int KM_TP = km_tile_rows<D_T>();
    return sizeof(float) * (2 * KM_TP * (D_T + 4) + D_T + KM_WAVES * D_T) + sizeof(int) * (KM_Q + KM_Q * KM_CAP + 2) +
           sizeof(tpg_u64) * KM_WAVES * 64;
}

typedef __bf16 km_bf16x8 __attribute__((ext_vector_type(8)));

// v_min / v_max as they are, for values the compiler's own instructions produced: fminf / fmaxf put a canonicalising
// v_max x, x, x in front of their operands (quiet-NaN semantics), a third of the sorted-insert chain.  NOT for the
// accumulators of a matrix instruction: the hazard recogniser does not look inside inline asm, and a v_min issued
// before the matrix unit has written its result reads the old register (measured: thresholds far too high, 12 % of the
// queries overflowed their candidate list).  No NaN reaches these (finite rows, 1e30 on padding).
__device__ __forceinline__ float km_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float km_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi) (x - hi is exact in fp32): |r| <= 2^-16 |x|
__device__ __forceinline__ void km_split2(float a, float b, unsigned &hi, unsigned &lo) {
    const __hip_bfloat16 ha = __float2bfloat16(a), hb = __float2bfloat16(b);
    const unsigned short ua = *reinterpret_cast<const unsigned short *>(&ha), ub = *reinterpret_cast<const unsigned short *>(&hb);
    hi = (unsigned)ua | ((unsigned)ub << 16);
    const float ra = a - __uint_as_float((unsigned)ua << 16), rb = b - __uint_as_float((unsigned)ub << 16);
    const __hip_bfloat16 la = __float2bfloat16(ra), lb = __float2bfloat16(rb);
    lo = (unsigned)*reinterpret_cast<const unsigned short *>(&la) | ((unsigned)*reinterpret_cast<const unsigned short *>(&lb) << 16);
}

template <int D_T, int KM_M>
__global__ __launch_bounds__(KM_WAVES * 64, 8 / KM_WAVES) void knn_mfma_kernel(
    const float *__restrict__ p1, const float *__restrict__ p2, const int64_t *__restrict__ len1,
    const int64_t *__restrict__ len2, int P1, int P2, int K, int gx, int total, float *__restrict__ dist,
    int64_t *__restrict__ idx) {
    constexpr int KM_TP = km_tile_rows<D_T>();
    constexpr int LD = D_T + 4;                    // LDS row in 32-bit words: D/2 of hi pairs, D/2 of lo pairs, 4 with |y|^2 (16 rows of b128 reads: 64 banks)
    constexpr int V = D_T / 4;                     // float4 per row
    constexpr int RPP = KM_WAVES * 64 / V;         // rows staged per pass of the workgroup
    constexpr int NP = KM_TP / RPP;                // passes per tile
    constexpr int KS = D_T / 16;                   // k-steps of the 32x32x16 matrix instruction
    constexpr int NS = KM_TP / 32;                 // 32-row sub-tiles per staged tile
    static_assert(NP >= 1 && KM_TP % RPP == 0 && (V == 8 || V == 16) && NS % 2 == 0, "tile staging shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char km_smem[];
    unsigned *tile = reinterpret_cast<unsigned *>(km_smem);           // [2][KM_TP][LD]   bf16 pairs of -2 (x - origin): hi | lo | norm triplet
    float *org = reinterpret_cast<float *>(tile + 2 * KM_TP * LD);    // [D_T]
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
    if (n2 < KM_MIN_VALID) {                       // too few points for the thresholds: the exhaustive kernel handles (and pads) these
        for (int t = tid; t < KM_Q; t += KM_WAVES * 64)
            if (q0 + t < P1) oi[(size_t)(q0 + t) * K] = KM_REDO;
        return;
    }
    const float *cb = p2 + (size_t)b * P2 * D_T;
    for (int t = tid; t < KM_Q; t += KM_WAVES * 64) cnt[t] = 0;
    // origin = mean of the cloud's first KM_TP points (a sample of the cloud in index order: near its centre, where the
    // first point alone may sit on the rim and double every |y|, i.e. quadruple the bound E_q); summed in a fixed order
    {
        float *raw = reinterpret_cast<float *>(tile);                  // [KM_TP][D_T] floats fit the first tile buffer
        const int rows0 = min(n2, KM_TP);
        for (int t = tid; t < rows0 * (D_T / 4); t += KM_WAVES * 64)
            reinterpret_cast<float4 *>(raw)[t] = reinterpret_cast<const float4 *>(cb)[t];
        __syncthreads();
        if (tid < D_T) {
            float a = 0.0f;
            for (int r = 0; r < rows0; ++r) a += raw[r * D_T + tid];
            org[tid] = a / (float)rows0;
        }
    }
    if (tid == 0) *pmax = 0u;
    __syncthreads();

    // ---- this lane's query (B operand): dims 16 ks + 8 h .. + 8 of query q0 + 32 wave + c, centred, split ----
    const int qi = q0 + wave * 32 + c;
    uint4 bqh[KS], bql[KS];
    float nq = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float y[8];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 o = *reinterpret_cast<const float4 *>(org + 16 * ks + 8 * h + 4 * v);
            if (qi < P1) {
                x = *reinterpret_cast<const float4 *>(p1 + ((size_t)b * P1 + qi) * D_T + 16 * ks + 8 * h + 4 * v);
                x.x -= o.x; x.y -= o.y; x.z -= o.z; x.w -= o.w;
            }
            y[4 * v] = x.x; y[4 * v + 1] = x.y; y[4 * v + 2] = x.z; y[4 * v + 3] = x.w;
            nq += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        }
        km_split2(y[0], y[1], bqh[ks].x, bql[ks].x);
        km_split2(y[2], y[3], bqh[ks].y, bql[ks].y);
        km_split2(y[4], y[5], bqh[ks].z, bql[ks].z);
        km_split2(y[6], y[7], bqh[ks].w, bql[ks].w);
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
            uint2 hi, lo;
            km_split2(-2.0f * y.x, -2.0f * y.y, hi.x, lo.x);
            km_split2(-2.0f * y.z, -2.0f * y.w, hi.y, lo.y);
            *reinterpret_cast<uint2 *>(tile + row * LD + 2 * sv) = hi;
            *reinterpret_cast<uint2 *>(tile + row * LD + D_T / 2 + 2 * sv) = lo;
            if (sv == 0) {
                // |y_p|^2 as three bf16 terms (8 + 8 + 8 significant bits: exact) in the row's four spare words:
                // one more matrix instruction against (1, 1, 1, 0 ...) starts the accumulator at the norm, instead
                // of four LDS reads and sixteen moves per lane and sub-tile
                const float nv = pv[p] ? s : KM_BIG;
                const __hip_bfloat16 n0 = __float2bfloat16(nv);
                const float r0 = nv - __bfloat162float(n0);
                const __hip_bfloat16 n1 = __float2bfloat16(r0);
                const __hip_bfloat16 n2b = __float2bfloat16(r0 - __bfloat162float(n1));
                const unsigned w0 = (unsigned)*reinterpret_cast<const unsigned short *>(&n0) |
                                    ((unsigned)*reinterpret_cast<const unsigned short *>(&n1) << 16);
                const unsigned w1 = (unsigned)*reinterpret_cast<const unsigned short *>(&n2b);
                *reinterpret_cast<uint4 *>(tile + row * LD + D_T) = make_uint4(w0, w1, 0u, 0u);
            }
            if (pv[p]) mymax = fmaxf(mymax, s);
        }
    };

    // B operand of the norm step: ones at k = 0, 1, 2 (lanes of the lower half), zeros elsewhere
    const uint4 ones = h == 0 ? make_uint4(0x3f803f80u, 0x00003f80u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
    const km_f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float lst[KM_M];
#pragma unroll
    for (int i = 0; i < KM_M; ++i) lst[i] = KM_BIG;
    const int ql = wave * 32 + c;                  // this lane's query within the workgroup
    const int NT = (n2 + KM_TP - 1) / KM_TP;
    // rows per entry of a lane's list of minima: 16 (one sub-tile) on small clouds, 32 on large ones -- a lane must see
    // well over KM_M entries for its KM_M-th smallest to be a tight threshold
    const bool pairs = n2 >= 4096;

    // One sweep over the cloud.  d' = |y_p|^2 - 2 y_p.y_q (the query's own norm is added in the tail): the accumulator
    // starts from the row norms and takes 3 KS matrix instructions (hi.hi + hi.lo + lo.hi) per 32 x 32 pairs.
    // The VALU work on sub-tile s-1 (minima / hit mask) sits between the matrix instructions of sub-tile s.
    auto sweep = [&](auto second_tag, int step0, float tauf) {
        constexpr bool SECOND = decltype(second_tag)::value;
        for (int t = 0; t < NT; ++t) {
            const int cur = (step0 + t) & 1;
            const bool more = !SECOND || t + 1 < NT;
            if (more) prefetch(t + 1 == NT ? 0 : t + 1);
            km_f32x16 prev;
            unsigned hits = 0;
            float held = KM_BIG;
#pragma unroll
            for (int sub = 0; sub <= NS; ++sub) {
                km_f32x16 acc;
                if (sub < NS) {
                    const unsigned *arow = tile + (cur * KM_TP + sub * 32 + c) * LD + 4 * h;
                    uint4 ah[KS], al[KS];
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        ah[ks] = *reinterpret_cast<const uint4 *>(arow + 8 * ks);
                        al[ks] = *reinterpret_cast<const uint4 *>(arow + D_T / 2 + 8 * ks);
                    }
                    const uint4 an = *reinterpret_cast<const uint4 *>(tile + (cur * KM_TP + sub * 32 + c) * LD + D_T);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(km_bf16x8, an), __builtin_bit_cast(km_bf16x8, ones), zero16, 0, 0, 0);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const km_bf16x8 Ah = __builtin_bit_cast(km_bf16x8, ah[ks]), Al = __builtin_bit_cast(km_bf16x8, al[ks]);
                        const km_bf16x8 Bh = __builtin_bit_cast(km_bf16x8, bqh[ks]), Bl = __builtin_bit_cast(km_bf16x8, bql[ks]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc, 0, 0, 0);
                    }
                }
                if (sub > 0) {
                    if constexpr (!SECOND) {
                        float v = fminf(fminf(prev[0], prev[1]), fminf(prev[2], prev[3]));
#pragma unroll
                        for (int r = 4; r < 16; r += 4) v = fminf(v, fminf(fminf(prev[r], prev[r + 1]), fminf(prev[r + 2], prev[r + 3])));
                        if (pairs && (sub & 1)) {
                            held = v;                                   // first half of a 32-row entry
                        } else {
                            v = km_min(v, held);
                            held = KM_BIG;
#pragma unroll
                            for (int i = 0; i < KM_M; ++i) {
                                const float lo = km_min(v, lst[i]);
                                v = km_max(v, lst[i]);
                                lst[i] = lo;
                            }
                        }
                    } else {
                        // bit r of `hits` (reversed: slot 15 - r) = sign of prev[r] - nextafter(tau): one subtract and
                        // one funnel shift per value instead of compare, select and or
                        hits = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            hits = __builtin_amdgcn_alignbit(hits, __float_as_uint(prev[r] - tauf), 31);
                    }
                }
                if constexpr (SECOND) {
                    if (sub > 0 && __ballot(hits != 0)) {
                        if (hits) {
                            int pos = atomicAdd(&cnt[ql], __popc(hits));
                            const int base = t * KM_TP + (sub - 1) * 32 + 4 * h;
                            while (hits) {
                                const int r = 15 - __builtin_ctz(hits);
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
    // hits are d' <= tau, tested as d' - tau_up < 0 with tau_up the next float above tau (below the padding rows' 1e30)
    const float tau_c = fminf(tau, 1.0e29f);
    const float tau_up = __uint_as_float(__float_as_uint(tau_c) + (tau_c >= 0.0f ? 1u : 0xffffffffu));
    sweep(std::true_type{}, NT, tau_c == 0.0f ? 1.0e-45f : tau_up);
    if (sv == 0) atomicMax(pmax, __float_as_uint(mymax));
    __syncthreads();

    // ---- tail: exact re-ranking, one wave per query ------------------------------------------------------
    // |d~ - d_canonical| <= E_q = c1 |y_q| max|y_p| + c2 (|y_q| + max|y_p|)^2 :
    //   c1 = 2^-13: the three dropped cross terms of the split products, 6.06 2^-16 |y_q||y_p|, with a third to spare;
    //   c2 = (8 D + 64) 2^-24: fp32 accumulation of the 3 D / 16 matrix steps (taken as 4 ulp each of the running
    //        magnitude), the two norms, the centring and the canonical sum itself, doubled.
    const float sq_pmax = sqrtf(__uint_as_float(*pmax));
    constexpr float C1 = 1.220703125e-4f;                                         // 2^-13
    constexpr float C2 = (float)(8 * D_T + 64) * 5.9604644775390625e-8f;          // (8 D + 64) 2^-24
    float *qs = qrow + wave * D_T;
    tpg_u64 *ks_ = keys + wave * 64;
    const tpg_u64 INF = ~0ull;
    // a query's candidate rows are a dependent chain (list entry -> row address -> 2 x D bytes from L2): the rows of
    // query qq + 1 are fetched into a second register set before query qq is ranked
    constexpr int V4 = D_T / 4;
    auto fetch = [&](int qq, float4 (&row)[V4], int &j, int &n) {
        const int i = q0 + wave * 32 + qq;
        j = -1;
        n = -1;
        if (qq >= 32 || i >= P1 || i >= n1) return;
        n = cnt[wave * 32 + qq];
        if (n <= KM_CAP && lane < n) {
            j = cbuf[(wave * 32 + qq) * KM_CAP + lane];
            const float4 *c4 = reinterpret_cast<const float4 *>(cb + (size_t)j * D_T);
#pragma unroll
            for (int d = 0; d < V4; ++d) row[d] = c4[d];
        }
    };
    auto finish = [&](int qq, const float4 (&row)[V4], int j, int n) {
        const int i = q0 + wave * 32 + qq;
        if (i >= P1) return;
        if (i >= n1) {
            if (lane < K) { od[(size_t)i * K + lane] = 0.0f; oi[(size_t)i * K + lane] = 0; }
            return;
        }
        const float nqq = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(nq), qq));
        const float tq = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(tau), qq)) + nqq;
        const float sqq = sqrtf(nqq);
        const float eq = C1 * sqq * sq_pmax + C2 * (sqq + sq_pmax) * (sqq + sq_pmax);
        if (lane < D_T) qs[lane] = p1[((size_t)b * P1 + i) * D_T + lane];
        tpg_u64 key = INF;
        if (j >= 0) {
            // the canonical sum of knn_dist (knn.hip), the candidate row in registers
            const float4 *q4 = reinterpret_cast<const float4 *>(qs);
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < V4; ++d) {
                const float4 qv = q4[d];
                float t;
                t = qv.x - row[d].x; acc = acc + t * t;
                t = qv.y - row[d].y; acc = acc + t * t;
                t = qv.z - row[d].z; acc = acc + t * t;
                t = qv.w - row[d].w; acc = acc + t * t;
            }
            key = knn_pack(acc, j);
        }
        ks_[lane] = key;
        int rank = 0;
        const int nk = n <= KM_CAP ? n : 0;
        for (int s = 0; s < nk; ++s) rank += ks_[s] < key ? 1 : 0;
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
    };
    float4 rowA[V4], rowB[V4];
    int jA, nA, jB, nB;
    fetch(0, rowA, jA, nA);
    for (int qq = 0; qq < 32; qq += 2) {
        fetch(qq + 1, rowB, jB, nB);
        finish(qq, rowA, jA, nA);
        fetch(qq + 2, rowA, jA, nA);
        finish(qq + 1, rowB, jB, nB);
    }
}
