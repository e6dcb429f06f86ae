// Row-gather "combine" kernels: the MI355X form of group -> first MLP layer.
//
// The reference gathers neighbour features into a (B,C,S,K) tensor and THEN applies the
// first 1x1 convolution to every one of the S*K grouped positions
// (gcn_lib/pointnet/gcn.py:207-210; QueryAndGroup + mlps[0] at discriminator.py:141-145;
// discriminator.py:270-280).  A 1x1 convolution commutes with a gather, so this build
// applies the first layer to the N un-grouped points (K..32x fewer FLOPs) and gathers its
// OUTPUT rows.  With channels-last rows (B,N,C) a gathered neighbour is one contiguous row
// of C*sizeof(T) bytes, so reads and writes are fully coalesced 16-byte vectors:
//
//   mode GATHER : out[b,s,k,:] = U[b,idx[b,s,k],:]
//   mode SUB    : out[b,s,k,:] = U[b,idx[b,s,k],:] - Q[b,s,:]           (set abstraction,
//                 flow embedding: Q carries the centre-dependent terms)
//   mode EDGE   : out[b,s,k,:] = A[b,idx,:] + lrelu(E[b,idx,:] - E[b,s,:])  (EdgeConv: A =
//                 lrelu(Wn f), E = We f; requires S == N)
//
// Backward is atomics-free: a per-cloud inverted index (destination row -> list of (s,k)
// entries in ascending entry order, csrc/invert_index.hpp) turns the scatter-add into a gather-reduce
// where each destination row is summed by one group of lanes and written exactly once.
// Element types: fp32 and bf16 (bf16 halves the bytes of the largest tensors; arithmetic is
// fp32 in registers, one rounding on store).
#include <hip/hip_bf16.h>

#include <cstdlib>

#include "invert_index.hpp"
#include "tpg_common.hpp"

namespace {

enum { MODE_GATHER = 0, MODE_SUB = 1, MODE_EDGE = 2 };

typedef unsigned int rc_u32x4 __attribute__((ext_vector_type(4)));
typedef float rc_f32x4 __attribute__((ext_vector_type(4)));
#ifndef TPG_RC_NT_STORE
#define TPG_RC_NT_STORE 0        // 1: non-temporal stores for the forward's output rows.  Round 3: with two chains per thread
                                 // 46.7 -> 20.8 us ALONE on cache-resident operands (tools/tune_rowcombine.py), but in the
                                 // step 24.0 instead of 19.2 us per launch and +5 % step time (tools/ab_rowcombine.sh): the
                                 // consumer reads these rows next and wants them in L2.  Off.
#endif

// NE consecutive elements of T <-> NE floats (16-byte vector accesses).
template <typename T, int NE> struct RowIO;
template <int NE> struct RowIO<float, NE> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[NE]) {
#pragma unroll
        for (int i = 0; i < NE / 4; ++i) {
            const float4 x = reinterpret_cast<const float4 *>(p)[i];
            v[4 * i] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
        }
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[NE]) {
#pragma unroll
        for (int i = 0; i < NE / 4; ++i)
            reinterpret_cast<float4 *>(p)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    }
    static __device__ __forceinline__ void store_stream(float *p, const float (&v)[NE]) {
#pragma unroll
        for (int i = 0; i < NE / 4; ++i) {
            const rc_f32x4 w = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
            __builtin_nontemporal_store(w, reinterpret_cast<rc_f32x4 *>(p) + i);
        }
    }
};
template <> struct RowIO<__hip_bfloat16, 8> {
    static __device__ __forceinline__ void load(const __hip_bfloat16 *p, float (&v)[8]) {
        const uint4 x = *reinterpret_cast<const uint4 *>(p);
        const unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(__hip_bfloat16 *p, const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __hip_bfloat16 lo = __float2bfloat16(v[2 * i]);       // round-to-nearest-even
            const __hip_bfloat16 hi = __float2bfloat16(v[2 * i + 1]);
            w[i] = (unsigned)(*reinterpret_cast<const unsigned short *>(&lo)) |
                   ((unsigned)(*reinterpret_cast<const unsigned short *>(&hi)) << 16);
        }
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    static __device__ __forceinline__ void store_stream(__hip_bfloat16 *p, const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __hip_bfloat16 lo = __float2bfloat16(v[2 * i]);
            const __hip_bfloat16 hi = __float2bfloat16(v[2 * i + 1]);
            w[i] = (unsigned)(*reinterpret_cast<const unsigned short *>(&lo)) |
                   ((unsigned)(*reinterpret_cast<const unsigned short *>(&hi)) << 16);
        }
        const rc_u32x4 x = {w[0], w[1], w[2], w[3]};
        __builtin_nontemporal_store(x, reinterpret_cast<rc_u32x4 *>(p));
    }
};
// elements per thread: 8 as soon as one side is bf16 (16-byte bf16 vectors), else 4
template <typename TA, typename TB> struct Elems {
    static constexpr int NE = (sizeof(TA) == 2 || sizeof(TB) == 2) ? 8 : 4;
};

// ------------------------------------------------------------------ forward
// Tunables (tools/tune_rowcombine.py)
#ifndef TPG_RC_FWD_U
#define TPG_RC_FWD_U 1          // output chunks per thread and iteration (independent load chains)
#endif
#ifndef TPG_RC_FWD_CAP
#define TPG_RC_FWD_CAP 16384    // most workgroups of a forward launch
#endif
#ifndef TPG_RC_BWD_CAP
#define TPG_RC_BWD_CAP 4096     // most workgroups of a backward launch
#endif
#ifndef TPG_RC_BWD_WAVE_CAP
#define TPG_RC_BWD_WAVE_CAP 16384   // same, wave-per-row form (4 rows per workgroup and pass)
#endif

// One output chunk = idx read -> row read(s) -> store: a dependent chain of two global loads.  A
// thread can carry TPG_RC_FWD_U chains at once (all index reads, then all row reads, then the
// arithmetic and the stores).  Measured (tools/tune_rowcombine.py): the source rows are L2
// resident (a cloud's first-layer output is a few MB), the kernel is bound by its output
// writes (~3 TB/s), and plain occupancy -- one chain per thread, 64 thin workgroups per CU --
// hides the chain better than batching does (U = 4: -7 %, U = 8: -20 %); hence the defaults.
template <typename TI, typename TO, int MODE>
__global__ __launch_bounds__(256) void rowcombine_fwd_kernel(
    const TI *__restrict__ U, const TI *__restrict__ QE, const int32_t *__restrict__ idx, int N, int S,
    int K, int C, float slope, TO *__restrict__ out, unsigned total, int xcd_order, int ld, float slope_u) {
    // ld = row stride of U / QE in elements (C, or 2C when both are column halves of ONE product: the EdgeConv
    // front end, tpg_rowcombine_edge_fwd); slope_u: EDGE only, LeakyReLU applied to the gathered U row (1 = none)
    constexpr int NE = Elems<TI, TO>::NE;
    constexpr int UF = TPG_RC_FWD_U;
    using In = RowIO<TI, NE>;
    using Out = RowIO<TO, NE>;
    const unsigned cpr = (unsigned)C / NE;  // 16-byte chunks per row
    const unsigned stride = gridDim.x * 256u;
    // consecutive workgroup ids land on the 8 XCDs in turn: with the plain order every XCD's L2 fetches every
    // cloud's source table (PMC, round 1: 1.5x the algorithmic bytes).  xcd_order (grid a multiple of 8): XCD x
    // takes the x-th eighth of each pass over the output, i.e. whole clouds.
    const unsigned lb = xcd_order ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    unsigned t = lb * 256u + threadIdx.x;
    auto one = [&](unsigned tt) {
        const unsigned row = tt / cpr;            // flat (b,s,k)
        const unsigned col = (tt - row * cpr) * NE;
        const unsigned bs = row / (unsigned)K;    // flat (b,s)
        const unsigned b = bs / (unsigned)S;
        const int n = tpg_clamp_idx(idx[row], N);
        float u[NE];
        In::load(U + ((size_t)b * N + n) * ld + col, u);
        if (MODE == MODE_SUB) {
            float q[NE];
            In::load(QE + (size_t)bs * ld + col, q);
#pragma unroll
            for (int i = 0; i < NE; ++i) u[i] = u[i] - q[i];
        } else if (MODE == MODE_EDGE) {
            float en[NE], es[NE];
            In::load(QE + ((size_t)b * N + n) * ld + col, en);
            In::load(QE + (size_t)bs * ld + col, es);   // S == N: centre row of E
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const float d = en[i] - es[i];
                const float a = u[i] > 0.0f ? u[i] : u[i] * slope_u;
                u[i] = a + (d > 0.0f ? d : d * slope);
            }
        }
        if (TPG_RC_NT_STORE) Out::store_stream(out + (size_t)row * C + col, u);
        else Out::store(out + (size_t)row * C + col, u);
    };
    for (; (unsigned long long)t + (unsigned long long)(UF - 1) * stride < total; t += UF * stride) {
        unsigned row[UF], col[UF], bs[UF], b[UF];
        int n[UF];
#pragma unroll
        for (int j = 0; j < UF; ++j) {
            const unsigned tt = t + j * stride;
            row[j] = tt / cpr;
            col[j] = (tt - row[j] * cpr) * NE;
            bs[j] = row[j] / (unsigned)K;
            b[j] = bs[j] / (unsigned)S;
            n[j] = idx[row[j]];
        }
        float u[UF][NE], q[UF][NE], en[UF][NE];
#pragma unroll
        for (int j = 0; j < UF; ++j) {
            n[j] = tpg_clamp_idx(n[j], N);
            In::load(U + ((size_t)b[j] * N + n[j]) * ld + col[j], u[j]);
            if (MODE != MODE_GATHER) In::load(QE + (size_t)bs[j] * ld + col[j], q[j]);      // centre row
            if (MODE == MODE_EDGE) In::load(QE + ((size_t)b[j] * N + n[j]) * ld + col[j], en[j]);
        }
#pragma unroll
        for (int j = 0; j < UF; ++j) {
            if (MODE == MODE_SUB) {
#pragma unroll
                for (int i = 0; i < NE; ++i) u[j][i] = u[j][i] - q[j][i];
            } else if (MODE == MODE_EDGE) {
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const float d = en[j][i] - q[j][i];
                    const float a = u[j][i] > 0.0f ? u[j][i] : u[j][i] * slope_u;
                    u[j][i] = a + (d > 0.0f ? d : d * slope);
                }
            }
            if (TPG_RC_NT_STORE) Out::store_stream(out + (size_t)row[j] * C + col[j], u[j]);
            else Out::store(out + (size_t)row[j] * C + col[j], u[j]);
        }
    }
    for (; t < total; t += stride) one(t);
}

// (the inverted index the backward walks: csrc/invert_index.hpp -- lists in ascending entry order)

// ------------------------------------------------------------------ backward
// one 16-byte chunk of one destination row per thread; gU (and gE for EDGE) written once.
template <typename TI, typename TG, int MODE>
__global__ __launch_bounds__(256) void rowcombine_bwd_kernel(
    const TG *__restrict__ gout, const int32_t *__restrict__ idx, const int32_t *__restrict__ offs,
    const int32_t *__restrict__ list, const TI *__restrict__ E, int N, int S, int K, int C, float slope,
    TI *__restrict__ gU, TI *__restrict__ gE, unsigned total, int ld, const TI *__restrict__ Uraw, float slope_u) {
    // ld: row stride of E / gU / gE (and Uraw) in elements; Uraw (EDGE, may be NULL): the forward's U rows before
    // their LeakyReLU(slope_u) -- gU is then the gradient of those raw rows
    constexpr int NE = Elems<TI, TG>::NE;
    using In = RowIO<TI, NE>;
    using Gr = RowIO<TG, NE>;
    const unsigned cpr = (unsigned)C / NE;
    const size_t SK = (size_t)S * K;
    for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
        const unsigned drow = t / cpr;  // flat (b,n)
        const unsigned col = (t - drow * cpr) * NE;
        const unsigned b = drow / (unsigned)N;
        const unsigned n = drow - b * (unsigned)N;
        const int32_t *of = offs + (size_t)b * (N + 1);
        const int32_t *ls = list + (size_t)b * SK;
        const TG *go = gout + (size_t)b * SK * C + col;
        float acc[NE], accE[NE], en[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) { acc[i] = 0.0f; accE[i] = 0.0f; }
        if (MODE == MODE_EDGE) In::load(E + (size_t)drow * ld + col, en);
        const int p1 = of[n + 1];
        int p = of[n];
        // entries four at a time: the 4 list reads, then the 4 (+4) row reads are independent
        // loads in flight together; the sums keep the sequential order of the entries
        for (; p + 4 <= p1; p += 4) {
            int e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) e[u] = ls[p + u];
            float g[4][NE], es[4][NE];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                Gr::load(go + (size_t)e[u] * C, g[u]);
                if (MODE == MODE_EDGE) In::load(E + ((size_t)b * N + (unsigned)e[u] / (unsigned)K) * ld + col, es[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < NE; ++i) acc[i] += g[u][i];
                if (MODE == MODE_EDGE) {
#pragma unroll
                    for (int i = 0; i < NE; ++i) accE[i] += (en[i] - es[u][i] > 0.0f) ? g[u][i] : g[u][i] * slope;
                }
            }
        }
        for (; p < p1; ++p) {
            const int e = ls[p];
            float g[NE];
            Gr::load(go + (size_t)e * C, g);
#pragma unroll
            for (int i = 0; i < NE; ++i) acc[i] += g[i];
            if (MODE == MODE_EDGE) {
                float es[NE];
                In::load(E + ((size_t)b * N + (unsigned)e / (unsigned)K) * ld + col, es);
#pragma unroll
                for (int i = 0; i < NE; ++i) accE[i] += (en[i] - es[i] > 0.0f) ? g[i] : g[i] * slope;
            }
        }
        if (MODE == MODE_EDGE) {
            // this row as a CENTRE: -sum_k g * lrelu'(E[nbr] - E[n])   (S == N)
            int k = 0;
            for (; k + 4 <= K; k += 4) {
                int nb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) nb[u] = tpg_clamp_idx(idx[(size_t)b * SK + (size_t)n * K + k + u], N);
                float g[4][NE], eb[4][NE];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    Gr::load(go + ((size_t)n * K + k + u) * C, g[u]);
                    In::load(E + ((size_t)b * N + nb[u]) * ld + col, eb[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < NE; ++i) accE[i] -= (eb[u][i] - en[i] > 0.0f) ? g[u][i] : g[u][i] * slope;
            }
            for (; k < K; ++k) {
                const size_t e = (size_t)n * K + k;
                const int nb = tpg_clamp_idx(idx[(size_t)b * SK + e], N);
                float g[NE], eb[NE];
                Gr::load(go + e * C, g);
                In::load(E + ((size_t)b * N + nb) * ld + col, eb);
#pragma unroll
                for (int i = 0; i < NE; ++i) accE[i] -= (eb[i] - en[i] > 0.0f) ? g[i] : g[i] * slope;
            }
            In::store(gE + (size_t)drow * ld + col, accE);
            if (Uraw) {
                float a[NE];
                In::load(Uraw + (size_t)drow * ld + col, a);
#pragma unroll
                for (int i = 0; i < NE; ++i) acc[i] = a[i] > 0.0f ? acc[i] : acc[i] * slope_u;
            }
        }
        In::store(gU + (size_t)drow * ld + col, acc);
    }
}

// The same sums with one WAVE per destination row: the C/NE chunk lanes of a row sit side by side
// (a coalesced row read per entry) and the 64/(C/NE) lane groups take every G-th entry of the row's
// list, partial sums folded across the groups at the end.  A ball query keeps the first `nsample`
// points IN INDEX ORDER, so low-index points sit in hundreds of groups: lists of 8-16 entries on
// average have 100-500 in the tail (tools/rowcombine_lists.py), and with one thread per row the
// launch lasted as long as its longest list (142 us for 67 MB); here the longest list is walked
// G entries at a time.  Needs C/NE to divide 64.  (Walking the rows as a three-stage pipeline -- next
// rows' list bounds and first entries read while this row's gradient rows are in flight -- changed
// nothing: the launch is not bound by that chain.)
template <typename TI, typename TG, int MODE>
__global__ __launch_bounds__(256) void rowcombine_bwd_wave_kernel(
    const TG *__restrict__ gout, const int32_t *__restrict__ idx, const int32_t *__restrict__ offs,
    const int32_t *__restrict__ list, const TI *__restrict__ E, int N, int S, int K, int C, float slope,
    TI *__restrict__ gU, TI *__restrict__ gE, unsigned rows, int ld, const TI *__restrict__ Uraw, float slope_u) {
    constexpr int NE = Elems<TI, TG>::NE;
    using In = RowIO<TI, NE>;
    using Gr = RowIO<TG, NE>;
    const unsigned cpr = (unsigned)C / NE;          // 1, 2, 4 .. 64
    const unsigned G = 64u / cpr;                   // entries in flight per step
    const unsigned lane = threadIdx.x & 63u;
    const unsigned col = (lane % cpr) * NE;
    const unsigned grp = lane / cpr;
    const size_t SK = (size_t)S * K;
    const unsigned wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    for (unsigned drow = wave; drow < rows; drow += nwaves) {
        const unsigned b = drow / (unsigned)N;
        const unsigned n = drow - b * (unsigned)N;
        const int32_t *of = offs + (size_t)b * (N + 1);
        const int32_t *ls = list + (size_t)b * SK;
        const TG *go = gout + (size_t)b * SK * C + col;
        float acc[NE], accE[NE], en[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) { acc[i] = 0.0f; accE[i] = 0.0f; }
        if (MODE == MODE_EDGE) In::load(E + (size_t)drow * ld + col, en);
        const int p1 = of[n + 1];
        int p = of[n] + (int)grp;
        // two entries per lane group in flight
        for (; p + (int)G < p1; p += 2 * (int)G) {
            const int e0 = ls[p], e1 = ls[p + G];
            float g0[NE], g1[NE];
            Gr::load(go + (size_t)e0 * C, g0);
            Gr::load(go + (size_t)e1 * C, g1);
            if (MODE == MODE_EDGE) {
                float s0[NE], s1[NE];
                In::load(E + ((size_t)b * N + (unsigned)e0 / (unsigned)K) * ld + col, s0);
                In::load(E + ((size_t)b * N + (unsigned)e1 / (unsigned)K) * ld + col, s1);
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    accE[i] += (en[i] - s0[i] > 0.0f) ? g0[i] : g0[i] * slope;
                    accE[i] += (en[i] - s1[i] > 0.0f) ? g1[i] : g1[i] * slope;
                }
            }
#pragma unroll
            for (int i = 0; i < NE; ++i) acc[i] += g0[i] + g1[i];
        }
        if (p < p1) {
            const int e0 = ls[p];
            float g0[NE];
            Gr::load(go + (size_t)e0 * C, g0);
#pragma unroll
            for (int i = 0; i < NE; ++i) acc[i] += g0[i];
            if (MODE == MODE_EDGE) {
                float s0[NE];
                In::load(E + ((size_t)b * N + (unsigned)e0 / (unsigned)K) * ld + col, s0);
#pragma unroll
                for (int i = 0; i < NE; ++i) accE[i] += (en[i] - s0[i] > 0.0f) ? g0[i] : g0[i] * slope;
            }
        }
        if (MODE == MODE_EDGE) {
            // this row as a CENTRE: -sum_k g * lrelu'(E[nbr] - E[n])   (S == N)
            for (int k = (int)grp; k < K; k += (int)G) {
                const size_t e = (size_t)n * K + k;
                const int nb = tpg_clamp_idx(idx[(size_t)b * SK + e], N);
                float g0[NE], eb[NE];
                Gr::load(go + e * C, g0);
                In::load(E + ((size_t)b * N + nb) * ld + col, eb);
#pragma unroll
                for (int i = 0; i < NE; ++i) accE[i] -= (eb[i] - en[i] > 0.0f) ? g0[i] : g0[i] * slope;
            }
        }
        // fold the lane groups: lanes with the same chunk are cpr apart
        for (unsigned off = cpr; off < 64u; off <<= 1) {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                acc[i] += __shfl_xor(acc[i], (int)off, 64);
                if (MODE == MODE_EDGE) accE[i] += __shfl_xor(accE[i], (int)off, 64);
            }
        }
        if (grp == 0) {
            if (MODE == MODE_EDGE) {
                In::store(gE + (size_t)drow * ld + col, accE);
                if (Uraw) {
                    float a[NE];
                    In::load(Uraw + (size_t)drow * ld + col, a);
#pragma unroll
                    for (int i = 0; i < NE; ++i) acc[i] = a[i] > 0.0f ? acc[i] : acc[i] * slope_u;
                }
            }
            In::store(gU + (size_t)drow * ld + col, acc);
        }
    }
}

// gQ[b,s,:] = -sum_k gout[b,s,k,:]
template <typename TI, typename TG>
__global__ __launch_bounds__(256) void rowsum_neg_kernel(const TG *__restrict__ gout, int K, int C,
                                                         TI *__restrict__ gQ, unsigned total) {
    constexpr int NE = Elems<TI, TG>::NE;
    const unsigned cpr = (unsigned)C / NE;
    for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
        const unsigned bs = t / cpr;
        const unsigned col = (t - bs * cpr) * NE;
        float acc[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) acc[i] = 0.0f;
        int k = 0;
        for (; k + 4 <= K; k += 4) {           // four independent row reads in flight, sequential sums
            float g[4][NE];
#pragma unroll
            for (int u = 0; u < 4; ++u) RowIO<TG, NE>::load(gout + ((size_t)bs * K + k + u) * C + col, g[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < NE; ++i) acc[i] -= g[u][i];
        }
        for (; k < K; ++k) {
            float g[NE];
            RowIO<TG, NE>::load(gout + ((size_t)bs * K + k) * C + col, g);
#pragma unroll
            for (int i = 0; i < NE; ++i) acc[i] -= g[i];
        }
        RowIO<TI, NE>::store(gQ + (size_t)bs * C + col, acc);
    }
}

unsigned grid_for(unsigned total, unsigned per_thread = 1u, unsigned cap = 16384u) {
    const unsigned blocks = (total + 256u * per_thread - 1u) / (256u * per_thread);
    return blocks < 1u ? 1u : (blocks > cap ? cap : blocks);
}

bool aligned16(const void *a, const void *b, const void *c, const void *d) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
             reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}
bool dtype_ok(int d) { return d == TPG_DTYPE_F32 || d == TPG_DTYPE_BF16; }

template <typename TI, typename TO>
int fwd_go(const void *U, const void *QE, const int32_t *idx, int mode, int B, int N, int S, int K,
           int C, float slope, void *out, hipStream_t st, int ld, float slope_u) {
    constexpr int NE = Elems<TI, TO>::NE;
    if (C % NE) return TPG_ERR_UNSUPPORTED;
    const unsigned long long total64 = (unsigned long long)B * S * K * (C / NE);
    if (total64 >= 0x7fffffffULL) return TPG_ERR_ARG;
    const unsigned total = (unsigned)total64;
    static const bool xcd_on = [] { const char *e = getenv("TPG_RC_XCD"); return !(e && e[0] == '0'); }();   // A/B switch
    unsigned nb = grid_for(total, TPG_RC_FWD_U, TPG_RC_FWD_CAP);
    const int xo = xcd_on && nb >= 64u;
    if (xo) nb = (nb + 7u) & ~7u;
    const dim3 g(nb), blk(256);
    const TI *u = static_cast<const TI *>(U), *q = static_cast<const TI *>(QE);
    TO *o = static_cast<TO *>(out);
    if (mode == MODE_GATHER)
        hipLaunchKernelGGL((rowcombine_fwd_kernel<TI, TO, MODE_GATHER>), g, blk, 0, st, u, q, idx, N, S, K, C, slope, o, total, xo, ld, slope_u);
    else if (mode == MODE_SUB)
        hipLaunchKernelGGL((rowcombine_fwd_kernel<TI, TO, MODE_SUB>), g, blk, 0, st, u, q, idx, N, S, K, C, slope, o, total, xo, ld, slope_u);
    else
        hipLaunchKernelGGL((rowcombine_fwd_kernel<TI, TO, MODE_EDGE>), g, blk, 0, st, u, q, idx, N, S, K, C, slope, o, total, xo, ld, slope_u);
    return TPG_OK;
}

template <typename TI, typename TG>
int bwd_go(const void *gout, const int32_t *idx, const int32_t *offs, const int32_t *list, const void *E,
           int mode, int B, int N, int S, int K, int C, float slope, void *gU, void *gQE, hipStream_t st, int ld,
           const void *Uraw, float slope_u) {
    constexpr int NE = Elems<TI, TG>::NE;
    if (C % NE) return TPG_ERR_UNSUPPORTED;
    const unsigned long long total64 = (unsigned long long)B * N * (C / NE);
    const unsigned long long totq64 = (unsigned long long)B * S * (C / NE);
    if (total64 >= 0x7fffffffULL || totq64 >= 0x7fffffffULL) return TPG_ERR_ARG;
    const unsigned total = (unsigned)total64;
    const dim3 g(grid_for(total, 1u, TPG_RC_BWD_CAP)), blk(256);
    const TG *go = static_cast<const TG *>(gout);
    const TI *e = static_cast<const TI *>(E);
    TI *gu = static_cast<TI *>(gU), *gq = static_cast<TI *>(gQE);
    const TI *ur = static_cast<const TI *>(Uraw);
    const unsigned cpr = (unsigned)(C / NE);
#ifndef TPG_RC_BWD_THREAD_PER_ROW
    // a wave per row needs >= 2 lane groups; for the generator's 16-channel EDGE rows (two chunks per row, ~k entries
    // per list) a wave is 32 groups for ~20 entries of 32 bytes: a thread per chunk walks them faster (cfg5's 163840 rows:
    // K = 10 217 -> 139 us, K = 20 277 -> 248 us, inverse index included)
    const bool by_wave = cpr <= 32 && (64u % cpr) == 0 && !(mode == MODE_EDGE && cpr <= 2);
#else
    const bool by_wave = false;
#endif
    const unsigned rows = (unsigned)((unsigned long long)B * N);
    const unsigned wgs = (rows + 3u) / 4u;                       // 4 waves = 4 rows per workgroup and pass
    const dim3 gw(wgs > TPG_RC_BWD_WAVE_CAP ? TPG_RC_BWD_WAVE_CAP : wgs);
    if (mode == MODE_EDGE) {
        if (by_wave)
            hipLaunchKernelGGL((rowcombine_bwd_wave_kernel<TI, TG, MODE_EDGE>), gw, blk, 0, st, go, idx, offs, list, e, N, S, K, C, slope, gu, gq, rows, ld, ur, slope_u);
        else
            hipLaunchKernelGGL((rowcombine_bwd_kernel<TI, TG, MODE_EDGE>), g, blk, 0, st, go, idx, offs, list, e, N, S, K, C, slope, gu, gq, total, ld, ur, slope_u);
    } else {
        if (by_wave)
            hipLaunchKernelGGL((rowcombine_bwd_wave_kernel<TI, TG, MODE_GATHER>), gw, blk, 0, st, go, idx, offs, list, e, N, S, K, C, slope, gu, gq, rows, ld, ur, slope_u);
        else
            hipLaunchKernelGGL((rowcombine_bwd_kernel<TI, TG, MODE_GATHER>), g, blk, 0, st, go, idx, offs, list, e, N, S, K, C, slope, gu, gq, total, ld, ur, slope_u);
        if (mode == MODE_SUB) {
            const unsigned totq = (unsigned)totq64;
            hipLaunchKernelGGL((rowsum_neg_kernel<TI, TG>), dim3(grid_for(totq)), blk, 0, st, go, K, C, gq, totq);
        }
    }
    return TPG_OK;
}

#define TPG_DISPATCH2(din, dout, CALL)                                                   \
    ((din) == TPG_DTYPE_F32 ? ((dout) == TPG_DTYPE_F32 ? CALL(float, float) : CALL(float, __hip_bfloat16)) \
                            : ((dout) == TPG_DTYPE_F32 ? CALL(__hip_bfloat16, float)                    \
                                                       : CALL(__hip_bfloat16, __hip_bfloat16)))

}  // namespace

extern "C" int tpg_rowcombine_fwd(const void *U, const void *QE, const int32_t *idx, int mode, int dtype_in,
                                  int dtype_out, int B, int N, int S, int K, int C, float slope, void *out,
                                  void *stream) {
    if (B < 0 || N <= 0 || S < 0 || K < 0 || C <= 0 || mode < 0 || mode > 2) return TPG_ERR_ARG;
    if ((long long)B * S * K == 0) return TPG_OK;
    if (!U || !idx || !out || (mode != MODE_GATHER && !QE)) return TPG_ERR_ARG;
    if (mode == MODE_EDGE && S != N) return TPG_ERR_ARG;
    if (!dtype_ok(dtype_in) || !dtype_ok(dtype_out) || !aligned16(U, QE, out, nullptr)) return TPG_ERR_UNSUPPORTED;
    hipStream_t st = tpg_stream(stream);
#define TPG_FWD(TI, TO) fwd_go<TI, TO>(U, QE, idx, mode, B, N, S, K, C, slope, out, st, C, 1.0f)
    const int rc = TPG_DISPATCH2(dtype_in, dtype_out, TPG_FWD);
#undef TPG_FWD
    if (rc) return rc;
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_invert_index(const int32_t *idx, int B, int N, int SK, int32_t *offs, int32_t *list,
                                int32_t *tmp, void *stream) {
    if (B < 0 || N <= 0 || SK < 0) return TPG_ERR_ARG;
    if (B == 0) return TPG_OK;
    if (!idx || !offs || !list) return TPG_ERR_ARG;
    const int rc = tpg_inv::launch<int32_t>(idx, B, N, SK, offs, list, tmp, tpg_stream(stream));
    if (rc) return rc;
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_rowcombine_bwd(const void *gout, const int32_t *idx, const int32_t *offs,
                                  const int32_t *list, const void *E, int mode, int dtype_in, int dtype_out,
                                  int B, int N, int S, int K, int C, float slope, void *gU, void *gQE,
                                  void *stream) {
    if (B < 0 || N <= 0 || S < 0 || K < 0 || C <= 0 || mode < 0 || mode > 2) return TPG_ERR_ARG;
    if (B == 0) return TPG_OK;
    if (!gout || !idx || !offs || !list || !gU) return TPG_ERR_ARG;
    if (mode == MODE_EDGE && (S != N || !E || !gQE)) return TPG_ERR_ARG;
    if (mode == MODE_SUB && !gQE) return TPG_ERR_ARG;
    if (!dtype_ok(dtype_in) || !dtype_ok(dtype_out) || !aligned16(gout, gU, gQE, E)) return TPG_ERR_UNSUPPORTED;
    hipStream_t st = tpg_stream(stream);
#define TPG_BWD(TI, TG) bwd_go<TI, TG>(gout, idx, offs, list, E, mode, B, N, S, K, C, slope, gU, gQE, st, C, nullptr, 1.0f)
    const int rc = TPG_DISPATCH2(dtype_in, dtype_out, TPG_BWD);
#undef TPG_BWD
    if (rc) return rc;
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

// EdgeConv front end on ONE product (round 3): Y (B,N,2C) = f [We; Wn]^T, columns [0,C) = E, [C,2C) = the node
// term BEFORE its LeakyReLU.  The same kernels with a row stride of 2C and the activation of the gathered node row
// (forward) / its derivative (backward) folded in: one GEMM, one data-gradient GEMM and one weight gradient per
// EdgeConv instead of two each, no LeakyReLU launches, no gradient sum of the shared input.
extern "C" int tpg_rowcombine_edge_fwd(const void *Y, const int32_t *idx, int dtype_in, int dtype_out, int B, int N,
                                       int K, int C, float slope_a, float slope_e, void *out, void *stream) {
    if (B < 0 || N <= 0 || K < 0 || C <= 0) return TPG_ERR_ARG;
    if ((long long)B * N * K == 0) return TPG_OK;
    if (!Y || !idx || !out) return TPG_ERR_ARG;
    if (!dtype_ok(dtype_in) || !dtype_ok(dtype_out) || !aligned16(Y, out, nullptr, nullptr)) return TPG_ERR_UNSUPPORTED;
    const int esz = dtype_in == TPG_DTYPE_F32 ? 4 : 2;
    if (((size_t)C * esz) & 15) return TPG_ERR_UNSUPPORTED;             // the node half starts 16-byte aligned
    const char *y = static_cast<const char *>(Y);
    const void *A = y + (size_t)C * esz, *E = y;
    hipStream_t st = tpg_stream(stream);
#define TPG_FWD(TI, TO) fwd_go<TI, TO>(A, E, idx, MODE_EDGE, B, N, N, K, C, slope_e, out, st, 2 * C, slope_a)
    const int rc = TPG_DISPATCH2(dtype_in, dtype_out, TPG_FWD);
#undef TPG_FWD
    if (rc) return rc;
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_rowcombine_edge_bwd(const void *gout, const int32_t *idx, const int32_t *offs, const int32_t *list,
                                       const void *Y, int dtype_in, int dtype_out, int B, int N, int K, int C,
                                       float slope_a, float slope_e, void *gY, void *stream) {
    if (B < 0 || N <= 0 || K < 0 || C <= 0) return TPG_ERR_ARG;
    if (B == 0) return TPG_OK;
    if (!gout || !idx || !offs || !list || !Y || !gY) return TPG_ERR_ARG;
    if (!dtype_ok(dtype_in) || !dtype_ok(dtype_out) || !aligned16(gout, gY, Y, nullptr)) return TPG_ERR_UNSUPPORTED;
    const int esz = dtype_in == TPG_DTYPE_F32 ? 4 : 2;
    if (((size_t)C * esz) & 15) return TPG_ERR_UNSUPPORTED;
    const char *y = static_cast<const char *>(Y);
    char *gy = static_cast<char *>(gY);
    const void *A = y + (size_t)C * esz, *E = y;
    void *gA = gy + (size_t)C * esz, *gE = gy;
    hipStream_t st = tpg_stream(stream);
#define TPG_BWD(TI, TG) bwd_go<TI, TG>(gout, idx, offs, list, E, MODE_EDGE, B, N, N, K, C, slope_e, gA, gE, st, 2 * C, A, slope_a)
    const int rc = TPG_DISPATCH2(dtype_in, dtype_out, TPG_BWD);
#undef TPG_BWD
    if (rc) return rc;
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}
