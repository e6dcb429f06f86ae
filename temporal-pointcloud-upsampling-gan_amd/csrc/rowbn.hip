// Fused training-mode BatchNorm + LeakyReLU (+ max over the K neighbours) on channels-last rows.
//
// Every shared-MLP layer of the discriminators is conv -> BatchNorm2d -> (Leaky)ReLU over a
// (B,C,S,ns) tensor, and the last one is followed by a max over ns (reference
// discriminator.py:63-78,145-150,279-282).  On rows (P = B*S*ns, C) that is
//
//   stats   : per-channel sum / sum of squares over the P rows            (1 read of x)
//   apply   : y = lrelu(a*x + b), a = gamma*rstd, b = beta - mean*a       (1 read, 1 write)
//             optionally reduced on the fly to max over each group of K consecutive rows,
//             writing P/K rows and a one-byte arg-max per (group, channel)
//   backward: g = gy * lrelu'(a*x+b);  dbeta = sum g, dgamma = sum g*xhat   (1 read of x,gy)
//             dx = a * (g - dbeta/P - xhat*dgamma/P)                       (1 read, 1 write)
//
// i.e. 2+2 launches and 2+3 streaming passes instead of PyTorch's BatchNorm (stats + transform),
// LeakyReLU, max-reduce and their three backward kernels.
//
// What the kernels are built around (measured on MI355X, tools/tune_rowbn.py):
//   * reductions are two launches: workgroups write per-channel partials, a small finalize
//     kernel (one workgroup per 4 channels) sums them in fixed order in fp64 -> bitwise
//     reproducible, no float atomics.  (Letting the last-arriving workgroup finish the sum saves
//     the ~5 us launch but serialises G x 2C dependent-latency loads on one CU: 35-90 us
//     measured, against 13 us for the pair of launches.)
//   * few fat workgroups beat many thin ones (workgroup dispatch + the per-channel constant
//     prologue dominate thin ones): grids are sized to ~1-2 workgroups per CU and each thread
//     walks many rows;
//   * the row walk is software-pipelined: the 16-byte loads of the next U rows are in flight
//     while the current U rows are computed (bf16 unpack, FMA, select: ~50 VALU ops per 16 B
//     would otherwise alternate with the memory latency instead of hiding it).
// Segments: a launch may carry `nseg` independent BatchNorm calls of the same module on equally
// sized, consecutive blocks of rows (the T frames of a clip, the fake and the real batch): each
// segment has its own statistics (gridDim.y = nseg), running statistics are updated segment after
// segment exactly as the separate calls would, dgamma / dbeta are summed over the segments.
// Rows are 16-byte vectors per lane (fp32 or bf16 storage, fp32 arithmetic); a thread owns one
// column chunk, so per-channel constants stay in registers.  Sums are taken about a per-channel
// pivot (the first row) to keep E[x^2]-E[x]^2 well conditioned.
#include <hip/hip_bf16.h>

#include "tpg_common.hpp"

namespace {

constexpr int BN_THREADS = 256;

// Tunables (tools/tune_rowbn.py builds variants with -D and times them on the GPU)
#ifndef TPG_BN_MAX_BLOCKS
#define TPG_BN_MAX_BLOCKS 256  // most workgroups (= partial rows) of a reduction launch
#endif
#ifndef TPG_BN_STATS_ROWS
#define TPG_BN_STATS_ROWS 8    // rows per thread a reduction launch aims at (before the cap binds)
#endif
#ifndef TPG_BN_YRED_ROWS
#define TPG_BN_YRED_ROWS 4     // same for the small (gy, y) reduction of the max variant
#endif
#ifndef TPG_BN_STATS_U
#define TPG_BN_STATS_U 8       // rows per pipeline stage of the (one-tensor, read-only) statistics walk
#endif
#ifndef TPG_BN_APPLY_CAP
#define TPG_BN_APPLY_CAP 256   // most workgroups of a streaming (apply) launch: one per CU
#endif
#ifndef TPG_BN_APPLY_ROWS
#define TPG_BN_APPLY_ROWS 8    // rows per thread a streaming launch aims at
#endif
#ifndef TPG_BN_GROUP_CAP
#define TPG_BN_GROUP_CAP 512   // most workgroups of a max-variant launch (one thread walks whole groups)
#endif
#ifndef TPG_BN_UNROLL
#define TPG_BN_UNROLL 4        // rows per pipeline stage, one-tensor kernels
#endif
#ifndef TPG_BN_BWD_U
#define TPG_BN_BWD_U 2         // rows per pipeline stage, two-tensor kernels
#endif
constexpr int BN_MAX_BLOCKS = TPG_BN_MAX_BLOCKS;
constexpr int BN_UNROLL = TPG_BN_UNROLL;
constexpr int BN_BWD_U = TPG_BN_BWD_U;
constexpr int WS_HEAD = 16;    // floats reserved at the start of the workspace

__device__ __forceinline__ float lrelu_f(float z, float slope) { return z > 0.0f ? z : z * slope; }
// pre-activation, written ONCE so that forward and backward see the same sign
__device__ __forceinline__ float bn_z(float v, float mu, float a, float beta) { return (v - mu) * a + beta; }

// ---- a row chunk of NE channels held as raw 16-byte registers until it is used --------------
template <typename T, int NE> struct Chunk;
template <int NE> struct Chunk<float, NE> {
    float4 r[NE / 4];
    __device__ __forceinline__ void load(const float *p) {
#pragma unroll
        for (int i = 0; i < NE / 4; ++i) r[i] = reinterpret_cast<const float4 *>(p)[i];
    }
    __device__ __forceinline__ void unpack(float (&v)[NE]) const {
#pragma unroll
        for (int i = 0; i < NE / 4; ++i) {
            v[4 * i] = r[i].x; v[4 * i + 1] = r[i].y; v[4 * i + 2] = r[i].z; v[4 * i + 3] = r[i].w;
        }
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[NE]) {
#pragma unroll
        for (int i = 0; i < NE / 4; ++i)
            reinterpret_cast<float4 *>(p)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    }
    static __device__ __forceinline__ float one(const float *p) { return *p; }
};
template <> struct Chunk<__hip_bfloat16, 8> {
    uint4 r;
    __device__ __forceinline__ void load(const __hip_bfloat16 *p) { r = *reinterpret_cast<const uint4 *>(p); }
    __device__ __forceinline__ void unpack(float (&v)[8]) const {
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
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
            const __hip_bfloat16 lo = __float2bfloat16(v[2 * i]);
            const __hip_bfloat16 hi = __float2bfloat16(v[2 * i + 1]);
            w[i] = (unsigned)(*reinterpret_cast<const unsigned short *>(&lo)) |
                   ((unsigned)(*reinterpret_cast<const unsigned short *>(&hi)) << 16);
        }
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    static __device__ __forceinline__ float one(const __hip_bfloat16 *p) {
        return __uint_as_float((unsigned)(*reinterpret_cast<const unsigned short *>(p)) << 16);
    }
};
template <typename T> struct BnElems { static constexpr int NE = sizeof(T) == 2 ? 8 : 4; };

// NE per-channel constants starting at channel `col` (a multiple of NE): vector loads when the
// array is 16-byte aligned, `dflt` when the array is absent.  All branches are wave-uniform.
template <int NE>
__device__ __forceinline__ void ld_consts(const float *__restrict__ p, int col, float dflt, float (&out)[NE]) {
    if (p == nullptr) {
#pragma unroll
        for (int i = 0; i < NE; ++i) out[i] = dflt;
    } else if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
#pragma unroll
        for (int i = 0; i < NE / 4; ++i) {
            const float4 t = reinterpret_cast<const float4 *>(p + col)[i];
            out[4 * i] = t.x; out[4 * i + 1] = t.y; out[4 * i + 2] = t.z; out[4 * i + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NE; ++i) out[i] = p[col + i];
    }
}

// ---- software-pipelined walk over the rows r0, r0+step, ... < P ------------------------------
// Stage: the raw registers of U rows.  load(stage, u, row) issues the global loads of one row,
// proc(stage, u, row) consumes them.  The loads of stage n+1 are issued before stage n is
// processed, so 2U rows of loads are in flight per thread.
template <int U, typename Stage, typename Load, typename Proc>
__device__ __forceinline__ void pipelined_rows(long long r, long long step, long long P, Load load, Proc proc) {
    Stage cur, nxt;
    bool have = r + (U - 1) * step < P;
    if (have) {
#pragma unroll
        for (int u = 0; u < U; ++u) load(cur, u, r + u * step);
    }
    while (have) {
        const long long rn = r + U * step;
        const bool have_n = rn + (U - 1) * step < P;
        if (have_n) {
#pragma unroll
            for (int u = 0; u < U; ++u) load(nxt, u, rn + u * step);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) proc(cur, u, r + u * step);
        cur = nxt;
        r = rn;
        have = have_n;
    }
    for (; r < P; r += step) {
        load(cur, 0, r);
        proc(cur, 0, r);
    }
}

// ---- block-level column reduction + "last workgroup finishes" ---------------------------------
// Thread layout shared by every kernel here: a thread owns ONE column chunk (chunk = tid % cpr)
// and walks rows rsub, rsub + rpi, ... (rsub = tid / cpr, rpi = 256 / cpr rows per workgroup
// iteration).  Column reduction of NQ quantities per channel: every thread has acc[NQ][NE] for
// its chunk; output o = (q, i, chunk) sums its rpi partners from LDS.
template <int NQ, int NE>
__device__ __forceinline__ void block_column_reduce(float (&acc)[NQ][NE], int cpr, int rpi, int C,
                                                    float *__restrict__ part) {
    __shared__ float red[BN_THREADS * NQ * NE];
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int i = 0; i < NE; ++i) red[(q * NE + i) * BN_THREADS + tid] = acc[q][i];
    __syncthreads();
    const int nout = NQ * NE * cpr;
    for (int o = tid; o < nout; o += BN_THREADS) {
        const int qi = o / cpr, chunk = o - qi * cpr;   // qi = q * NE + i
        const float *src = red + qi * BN_THREADS + chunk;
        float s = 0.0f;
        for (int r = 0; r < rpi; ++r) s += src[r * cpr];
        const int q = qi / NE, i = qi - q * NE;
        part[((size_t)blockIdx.x * NQ + q) * C + chunk * NE + i] = s;
    }
}

// Sums of the per-workgroup partials of two quantities, for every segment of a launch: one WAVE per channel
// (FIN_CH channels per workgroup), its lanes split into groups, one per segment (SP segments per sweep, a power
// of two; L = 64 / SP lanes each), so that the segments' partials travel side by side and the launch waits for
// memory once per sweep instead of once per segment.  fp64, fixed butterflies, no LDS and no barrier.
constexpr int FIN_CH = 4;
constexpr int FIN_THREADS = 64 * FIN_CH;
constexpr int FIN_R = 8;                       // requests in flight per lane and quantity
struct FinSplit { int SP, L; };
__device__ __forceinline__ FinSplit fin_split(int nseg) {
    int sp = 1;
    while (sp < nseg && sp < 64) sp <<= 1;
    return {sp, 64 / sp};
}
__device__ __forceinline__ double fin_group_sum(double v, int L) {
    for (int m = L >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double fin_wave_sum(double v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double fin_readlane(double v, int l) {          // l wave-uniform
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
// the two sums of segment `seg` (clamped when the group has none: live == false) in every lane of its group
__device__ __forceinline__ void finalize_sums(const float *__restrict__ ws, int seg, bool live, int sub, int L, int G,
                                              int C, int c, double &s0, double &s1) {
    const float *part = ws + WS_HEAD + (size_t)seg * BN_MAX_BLOCKS * 2 * C + c;
    double a0 = 0.0, a1 = 0.0;
    for (int g0 = 0; g0 < G; g0 += L * FIN_R) {
        float v0[FIN_R], v1[FIN_R];
#pragma unroll
        for (int i = 0; i < FIN_R; ++i) {      // unconditional, clamped: all requests of a block in flight
            const int g = g0 + sub + i * L;
            const bool ok = live && g < G;
            const float *pg = part + (size_t)(ok ? g : 0) * 2 * C;
            const float x0 = pg[0], x1 = pg[C];
            v0[i] = ok ? x0 : 0.0f;
            v1[i] = ok ? x1 : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < FIN_R; ++i) {
            a0 += (double)v0[i];
            a1 += (double)v1[i];
        }
    }
    s0 = fin_group_sum(a0, L);
    s1 = fin_group_sum(a1, L);
}

// ------------------------------------------------------------------ forward statistics
// partials -> mean, rstd (+ running statistics and the batch counter, as nn.BatchNorm does)
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void rowbn_stats_kernel(const T *__restrict__ x, long long P, int C,
                                                                 float *__restrict__ ws) {
    constexpr int NE = BnElems<T>::NE;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;  // chunks per row, rows per iteration
    const int tid = threadIdx.x;
    const int chunk = tid % cpr, rsub = tid / cpr;
    x += (size_t)blockIdx.y * P * C;                                         // this segment's rows
    float *part = ws + WS_HEAD + (size_t)blockIdx.y * BN_MAX_BLOCKS * 2 * C;
    float acc[2][NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) { acc[0][i] = 0.0f; acc[1][i] = 0.0f; }
    if (rsub < rpi) {
        float piv[NE];
        {
            Chunk<T, NE> p0;
            p0.load(x + chunk * NE);  // pivot = first row
            p0.unpack(piv);
        }
        struct Stage { Chunk<T, NE> v[TPG_BN_STATS_U]; };
        const T *xc = x + chunk * NE;
        pipelined_rows<TPG_BN_STATS_U, Stage>(
            (long long)blockIdx.x * rpi + rsub, (long long)gridDim.x * rpi, P,
            [&](Stage &s, int u, long long r) { s.v[u].load(xc + r * C); },
            [&](const Stage &s, int u, long long) {
                float v[NE];
                s.v[u].unpack(v);
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const float d = v[i] - piv[i];
                    acc[0][i] += d;
                    acc[1][i] += d * d;
                }
            });
    }
    block_column_reduce<2, NE>(acc, cpr, rpi, C, part);
}

// partials -> mean, rstd (+ running statistics and the batch counter, as nn.BatchNorm does);
// grid = ceil(C/FIN_CH)
template <typename T>
__global__ __launch_bounds__(FIN_THREADS) void rowbn_stats_finalize_kernel(
    const T *__restrict__ x, const float *__restrict__ ws, int G, long long P, int C, float eps,
    float momentum, float *__restrict__ running_mean, float *__restrict__ running_var,
    long long *__restrict__ num_batches_tracked, const float *__restrict__ mean_shift, float *__restrict__ mean,
    float *__restrict__ rstd, int nseg, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ ci_out) {
    // ci_out (nseg,4,C), optional: sc | sh | mu | rs of this BatchNorm as tpg_mlp_consts folds them (gamma / beta NULL = 1 / 0)
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += nseg;
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * FIN_CH + (threadIdx.x >> 6);
    if (c >= C) return;                        // whole wave
    const FinSplit sp = fin_split(nseg);
    const int grp = lane / sp.L, sub = lane - grp * sp.L;
    float rmean = running_mean ? running_mean[c] : 0.0f, rvar = running_mean ? running_var[c] : 0.0f;
    const double shift = mean_shift ? (double)mean_shift[c] : 0.0;   // see tpgan_ops.h
    for (int seg0 = 0; seg0 < nseg; seg0 += sp.SP) {
        const int seg = seg0 + grp;
        const bool live = seg < nseg;
        const int segc = live ? seg : nseg - 1;
        double s, ss;
        finalize_sums(ws, segc, live, sub, sp.L, G, C, c, s, ss);
        const double piv = Chunk<T, BnElems<T>::NE>::one(x + (size_t)segc * P * C + c);
        const double m = s / (double)P;
        double var = ss / (double)P - m * m;  // biased, about the pivot
        var = var < 0.0 ? 0.0 : var;
        if (live && sub == 0) {
            const float mu = (float)(piv + m), rs = (float)(1.0 / sqrt(var + (double)eps));
            mean[(size_t)seg * C + c] = mu;
            rstd[(size_t)seg * C + c] = rs;
            if (ci_out) {
                const float a = (gamma ? gamma[c] : 1.0f) * rs;
                float *o = ci_out + (size_t)seg * 4 * C + c;
                o[0] = a; o[C] = (beta ? beta[c] : 0.0f) - mu * a; o[2 * C] = mu; o[3 * C] = rs;
            }
        }
        if (running_mean) {                    // in call order: the running statistics chain
            const int last = nseg - seg0 < sp.SP ? nseg - seg0 : sp.SP;
            const double pm = piv + m;
            for (int k = 0; k < last; ++k) {
                const double mk = fin_readlane(pm, k * sp.L), vk = fin_readlane(var, k * sp.L);
                const double unbiased = P > 1 ? vk * (double)P / (double)(P - 1) : vk;
                rmean = (float)((1.0 - momentum) * rmean + momentum * (mk + shift));
                rvar = (float)((1.0 - momentum) * rvar + momentum * unbiased);
            }
        }
    }
    if (running_mean && lane == 0) { running_mean[c] = rmean; running_var[c] = rvar; }
}

// ------------------------------------------------------------------ forward apply (+max)
template <typename TI, typename TO>
__global__ __launch_bounds__(BN_THREADS) void rowbn_apply_kernel(
    const TI *__restrict__ x, long long P, int C, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    float slope, TO *__restrict__ y) {
    constexpr int NE = (sizeof(TI) == 2 || sizeof(TO) == 2) ? 8 : 4;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;
    const int chunk = threadIdx.x % cpr, rsub = threadIdx.x / cpr, col = chunk * NE;
    if (rsub >= rpi) return;
    x += (size_t)blockIdx.y * P * C;
    y += (size_t)blockIdx.y * P * C;
    if (mean) { mean += (size_t)blockIdx.y * C; rstd += (size_t)blockIdx.y * C; }
    float a[NE], b[NE], mu[NE];
    ld_consts<NE>(mean, col, 0.0f, mu);
    ld_consts<NE>(rstd, col, 1.0f, a);
    ld_consts<NE>(beta, col, 0.0f, b);
    {
        float gm[NE];
        ld_consts<NE>(gamma, col, 1.0f, gm);
#pragma unroll
        for (int i = 0; i < NE; ++i) a[i] = gm[i] * a[i];
    }
    struct Stage { Chunk<TI, NE> v[BN_UNROLL]; };
    const TI *xc = x + col;
    TO *yc = y + col;
    pipelined_rows<BN_UNROLL, Stage>(
        (long long)blockIdx.x * rpi + rsub, (long long)gridDim.x * rpi, P,
        [&](Stage &s, int u, long long r) { s.v[u].load(xc + r * C); },
        [&](const Stage &s, int u, long long r) {
            float v[NE];
            s.v[u].unpack(v);
#pragma unroll
            for (int i = 0; i < NE; ++i) v[i] = lrelu_f(bn_z(v[i], mu[i], a[i], b[i]), slope);
            Chunk<TO, NE>::store(yc + r * C, v);
        });
}

// groups of K consecutive rows -> one row (max) + arg-max byte per channel (first maximum).
// A thread walks whole groups; the walk over (group, k) is one pipeline (U rows per stage,
// U | K), so the loads of the next group's first rows are in flight while a group is closed.
// Few long groups (a GroupAll level: 8 clouds x 256 rows) leave that walk a handful of threads
// with 64 dependent stages each: then a group is cut into L runs of K/L rows, a thread walks runs,
// writes the run's fp32 maximum and arg-max byte to `part` / `parg`, and rowbn_max_combine_kernel
// takes the first maximum over the L runs (L == 1: straight to y / arg).
template <typename TI, typename TO, int U>
__global__ __launch_bounds__(BN_THREADS) void rowbn_apply_max_kernel(
    const TI *__restrict__ x, long long Gp, int K, int C, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    float slope, TO *__restrict__ y, uint8_t *__restrict__ arg, int L, float *__restrict__ part,
    uint8_t *__restrict__ parg) {
    constexpr int NE = (sizeof(TI) == 2 || sizeof(TO) == 2) ? 8 : 4;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;
    const int chunk = threadIdx.x % cpr, rsub = threadIdx.x / cpr, col = chunk * NE;
    if (rsub >= rpi) return;
    x += (size_t)blockIdx.y * Gp * K * C;
    y += (size_t)blockIdx.y * Gp * C;
    arg += (size_t)blockIdx.y * Gp * C;
    if (L > 1) {
        part += (size_t)blockIdx.y * Gp * L * C;
        parg += (size_t)blockIdx.y * Gp * L * C;
        Gp *= L;                       // from here on a "group" is a run of K / L rows
        K /= L;
    }
    if (mean) { mean += (size_t)blockIdx.y * C; rstd += (size_t)blockIdx.y * C; }
    float a[NE], b[NE], mu[NE];
    ld_consts<NE>(mean, col, 0.0f, mu);
    ld_consts<NE>(rstd, col, 1.0f, a);
    ld_consts<NE>(beta, col, 0.0f, b);
    {
        float gm[NE];
        ld_consts<NE>(gamma, col, 1.0f, gm);
#pragma unroll
        for (int i = 0; i < NE; ++i) a[i] = gm[i] * a[i];
    }
    const long long gstep = (long long)gridDim.x * rpi;
    long long grp = (long long)blockIdx.x * rpi + rsub;
    int k = 0;
    bool have = grp < Gp;
    Chunk<TI, NE> cur[U], nxt[U];
    const TI *xc = x + col;
    if (have) {
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u].load(xc + (grp * K + u) * C);
    }
    float best[NE];
    int bk[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) { best[i] = -INFINITY; bk[i] = 0; }
    while (have) {
        int kn = k + U;
        long long grpn = grp;
        if (kn >= K) { kn = 0; grpn = grp + gstep; }
        const bool have_n = grpn < Gp;
        if (have_n) {
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u].load(xc + (grpn * K + kn + u) * C);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float v[NE];
            cur[u].unpack(v);
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const float z = lrelu_f(bn_z(v[i], mu[i], a[i], b[i]), slope);
                if (z > best[i]) { best[i] = z; bk[i] = k + u; }
            }
        }
        if (kn == 0) {   // group complete
            uint8_t *ab = arg;
            if (L > 1) {                                  // a run: fp32 maximum, k counted in the group
                Chunk<float, NE>::store(part + grp * C + col, best);
                const int kb = (int)(grp % L) * K;
#pragma unroll
                for (int i = 0; i < NE; ++i) bk[i] += kb;
                ab = parg;
            } else {
                Chunk<TO, NE>::store(y + grp * C + col, best);
            }
            // NE arg-max bytes (8-byte aligned for NE = 8, 4-byte for NE = 4)
            if constexpr (NE == 8) {
                const unsigned lo = bk[0] | (bk[1] << 8) | (bk[2] << 16) | ((unsigned)bk[3] << 24);
                const unsigned hi = bk[4] | (bk[5] << 8) | (bk[6] << 16) | ((unsigned)bk[7] << 24);
                *reinterpret_cast<uint2 *>(ab + grp * C + col) = make_uint2(lo, hi);
            } else {
                *reinterpret_cast<unsigned *>(ab + grp * C + col) =
                    bk[0] | (bk[1] << 8) | (bk[2] << 16) | ((unsigned)bk[3] << 24);
            }
#pragma unroll
            for (int i = 0; i < NE; ++i) { best[i] = -INFINITY; bk[i] = 0; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        k = kn;
        grp = grpn;
        have = have_n;
    }
}

// first maximum over the L runs of every group (see rowbn_apply_max_kernel); one thread per
// (group, chunk of NE channels), grid.y = segment
template <typename TO, int NE>
__global__ __launch_bounds__(BN_THREADS) void rowbn_max_combine_kernel(
    const float *__restrict__ part, const uint8_t *__restrict__ parg, long long Gp, int L, int C,
    TO *__restrict__ y, uint8_t *__restrict__ arg) {
    const int cpr = C / NE;
    const long long t = (long long)blockIdx.x * BN_THREADS + threadIdx.x;
    if (t >= Gp * cpr) return;
    const long long grp = t / cpr;
    const int col = (int)(t % cpr) * NE;
    part += (size_t)blockIdx.y * Gp * L * C + (size_t)grp * L * C + col;
    parg += (size_t)blockIdx.y * Gp * L * C + (size_t)grp * L * C + col;
    y += (size_t)blockIdx.y * Gp * C + grp * C + col;
    arg += (size_t)blockIdx.y * Gp * C + grp * C + col;
    float best[NE];
    int run[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) { best[i] = -INFINITY; run[i] = 0; }
#pragma unroll 8
    for (int l = 0; l < L; ++l) {
        Chunk<float, NE> c;
        c.load(part + (size_t)l * C);
        float v[NE];
        c.unpack(v);
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (v[i] > best[i]) { best[i] = v[i]; run[i] = l; }        // strict: the first run wins a tie
    }
    Chunk<TO, NE>::store(y, best);
#pragma unroll
    for (int i = 0; i < NE; ++i) arg[i] = parg[(size_t)run[i] * C + i];
}

// ------------------------------------------------------------------ backward reductions
// partials (sum g, sum g*xhat) -> dgamma, dbeta (fp32) and the two per-channel constants of dx
// (c12 = {sum g / P, sum g*xhat / P}, zero for eval-mode statistics); grid = ceil(C/FIN_CH)
__global__ __launch_bounds__(FIN_THREADS) void rowbn_bwd_finalize_kernel(const float *__restrict__ ws, int G,
                                                                        long long P, int C, int training,
                                                                        float *__restrict__ dgamma,
                                                                        float *__restrict__ dbeta,
                                                                        float *__restrict__ c12, int nseg,
                                                                        const float *__restrict__ mean,
                                                                        const float *__restrict__ rstd,
                                                                        const float *__restrict__ gamma,
                                                                        float *__restrict__ cb) {
    // cb (nseg,4,C), optional: a | f*mu | e | f of this BatchNorm's backward as tpg_mlp_consts folds them
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * FIN_CH + (threadIdx.x >> 6);
    if (c >= C) return;                        // whole wave
    const FinSplit sp = fin_split(nseg);
    const int grp = lane / sp.L, sub = lane - grp * sp.L;
    double ts = 0.0, tsx = 0.0;
    for (int seg0 = 0; seg0 < nseg; seg0 += sp.SP) {
        const int seg = seg0 + grp;
        const bool live = seg < nseg;
        double s, sx;
        finalize_sums(ws, live ? seg : nseg - 1, live, sub, sp.L, G, C, c, s, sx);
        const bool owner = live && sub == 0;
        ts += fin_wave_sum(owner ? s : 0.0);   // fixed order over the segments of the sweep
        tsx += fin_wave_sum(owner ? sx : 0.0);
        if (owner) {
            float *cs = c12 + (size_t)seg * 2 * C;
            const float c1 = training ? (float)(s / (double)P) : 0.0f;       // eval-mode BN: no batch terms
            const float c2 = training ? (float)(sx / (double)P) : 0.0f;
            cs[c] = c1;
            cs[C + c] = c2;
            if (cb) {
                const float mu = mean ? mean[(size_t)seg * C + c] : 0.0f, rs = rstd ? rstd[(size_t)seg * C + c] : 1.0f;
                const float a = (gamma ? gamma[c] : 1.0f) * rs, f = a * rs * c2;
                float *o = cb + (size_t)seg * 4 * C + c;
                o[0] = a; o[C] = f * mu; o[2 * C] = -a * c1; o[3 * C] = f;
            }
        }
    }
    if (lane != 0) return;
    if (dbeta) dbeta[c] = (float)ts;
    if (dgamma) dgamma[c] = (float)tsx;
}

template <typename TI, typename TG>
__global__ __launch_bounds__(BN_THREADS) void rowbn_bwd_reduce_kernel(
    const TG *__restrict__ gy, const TI *__restrict__ x, long long P, int C, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    float slope, float *__restrict__ ws) {
    constexpr int NE = (sizeof(TI) == 2 || sizeof(TG) == 2) ? 8 : 4;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;
    const int tid = threadIdx.x;
    const int chunk = tid % cpr, rsub = tid / cpr, col = chunk * NE;
    gy += (size_t)blockIdx.y * P * C;
    x += (size_t)blockIdx.y * P * C;
    if (mean) { mean += (size_t)blockIdx.y * C; rstd += (size_t)blockIdx.y * C; }
    ws += (size_t)blockIdx.y * BN_MAX_BLOCKS * 2 * C;
    float acc[2][NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) { acc[0][i] = 0.0f; acc[1][i] = 0.0f; }
    if (rsub < rpi) {
        float a[NE], b[NE], mu[NE], rs[NE];
        ld_consts<NE>(mean, col, 0.0f, mu);
        ld_consts<NE>(rstd, col, 1.0f, rs);
        ld_consts<NE>(beta, col, 0.0f, b);
        ld_consts<NE>(gamma, col, 1.0f, a);
#pragma unroll
        for (int i = 0; i < NE; ++i) a[i] = a[i] * rs[i];
        struct Stage { Chunk<TI, NE> v[BN_BWD_U]; Chunk<TG, NE> g[BN_BWD_U]; };
        const TI *xc = x + col;
        const TG *gc = gy + col;
        pipelined_rows<BN_BWD_U, Stage>(
            (long long)blockIdx.x * rpi + rsub, (long long)gridDim.x * rpi, P,
            [&](Stage &s, int u, long long r) { s.v[u].load(xc + r * C); s.g[u].load(gc + r * C); },
            [&](const Stage &s, int u, long long) {
                float v[NE], g[NE];
                s.v[u].unpack(v);
                s.g[u].unpack(g);
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const float gg = bn_z(v[i], mu[i], a[i], b[i]) > 0.0f ? g[i] : g[i] * slope;
                    acc[0][i] += gg;
                    acc[1][i] += gg * ((v[i] - mu[i]) * rs[i]);
                }
            });
    }
    block_column_reduce<2, NE>(acc, cpr, rpi, C, ws + WS_HEAD);
}

// max variant: gy is (P/K, C); only the arg-max row of each group carries gradient.
// With the forward's output y (stored like gy) the arg-max row's pre-activation is
// z = y > 0 ? y : y / slope and xhat = (z - beta) / gamma: two small streaming reads instead of
// one scattered 2/4-byte load of x per (group, channel).  Per-channel guard: the inversion is
// used only where it is well conditioned (|beta| <= 4|gamma|); y == nullptr: always gather.
template <typename TI, typename TG>
__global__ __launch_bounds__(BN_THREADS) void rowbn_bwd_reduce_max_kernel(
    const TG *__restrict__ gy, const TG *__restrict__ y, const TI *__restrict__ x,
    const uint8_t *__restrict__ arg, long long Gp, int K, int C, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    float slope, float *__restrict__ ws, TG *__restrict__ ag) {
    // ag (nseg*Gp, C), optional (needs y): gamma*rstd * lrelu'(y) * gy per (group, channel) -- the MODE_MAX operand of
    // the fused tail's gradient kernels (tpg_mlp_max_prep's output, same bits), a by-product of the rows read here
    constexpr int NE = (sizeof(TI) == 2 || sizeof(TG) == 2) ? 8 : 4;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;
    const int tid = threadIdx.x;
    const int chunk = tid % cpr, rsub = tid / cpr, col = chunk * NE;
    gy += (size_t)blockIdx.y * Gp * C;
    if (ag) ag += (size_t)blockIdx.y * Gp * C;
    if (y) y += (size_t)blockIdx.y * Gp * C;
    x += (size_t)blockIdx.y * Gp * K * C;
    arg += (size_t)blockIdx.y * Gp * C;
    if (mean) { mean += (size_t)blockIdx.y * C; rstd += (size_t)blockIdx.y * C; }
    ws += (size_t)blockIdx.y * BN_MAX_BLOCKS * 2 * C;
    float acc[2][NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) { acc[0][i] = 0.0f; acc[1][i] = 0.0f; }
    if (rsub < rpi) {
        float a[NE], b[NE], mu[NE], rs[NE], ig[NE];
        bool inv[NE];
        const float islope = slope != 0.0f ? 1.0f / slope : 0.0f;
        ld_consts<NE>(mean, col, 0.0f, mu);
        ld_consts<NE>(rstd, col, 1.0f, rs);
        ld_consts<NE>(beta, col, 0.0f, b);
        ld_consts<NE>(gamma, col, 1.0f, a);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            inv[i] = y != nullptr && fabsf(b[i]) <= 4.0f * fabsf(a[i]);
            ig[i] = inv[i] ? 1.0f / a[i] : 0.0f;
            a[i] = a[i] * rs[i];
        }
        for (long long r = (long long)blockIdx.x * rpi + rsub; r < Gp; r += (long long)gridDim.x * rpi) {
            float g[NE], yy[NE];
            Chunk<TG, NE> cg, cy;
            cg.load(gy + r * C + col);
            if (y != nullptr) cy.load(y + r * C + col);
            else cy = cg;
            cg.unpack(g);
            cy.unpack(yy);
            if (ag != nullptr) {
                float av[NE];
#pragma unroll
                for (int i = 0; i < NE; ++i) av[i] = a[i] * (yy[i] > 0.0f ? g[i] : g[i] * slope);
                Chunk<TG, NE>::store(ag + r * C + col, av);
            }
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                if (inv[i]) {
                    const bool pos = yy[i] > 0.0f;
                    const float gg = pos ? g[i] : g[i] * slope;
                    const float z = pos ? yy[i] : yy[i] * islope;
                    acc[0][i] += gg;
                    acc[1][i] += gg * ((z - b[i]) * ig[i]);
                } else {
                    const int c = col + i;
                    const int k = arg[r * C + c];
                    const float v = Chunk<TI, NE>::one(x + (r * K + k) * C + c);
                    const float gg = bn_z(v, mu[i], a[i], b[i]) > 0.0f ? g[i] : g[i] * slope;
                    acc[0][i] += gg;
                    acc[1][i] += gg * ((v - mu[i]) * rs[i]);
                }
            }
        }
    }
    block_column_reduce<2, NE>(acc, cpr, rpi, C, ws + WS_HEAD);
}

// dx = a * (g - c1 - xhat*c2), K == 0
template <typename TI, typename TG>
__global__ __launch_bounds__(BN_THREADS) void rowbn_bwd_apply_kernel(
    const TG *__restrict__ gy, const TI *__restrict__ x, long long P, int C, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    float slope, const float *__restrict__ c12, TI *__restrict__ dx) {
    constexpr int NE = (sizeof(TI) == 2 || sizeof(TG) == 2) ? 8 : 4;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;
    const int chunk = threadIdx.x % cpr, rsub = threadIdx.x / cpr, col = chunk * NE;
    if (rsub >= rpi) return;
    gy += (size_t)blockIdx.y * P * C;
    x += (size_t)blockIdx.y * P * C;
    dx += (size_t)blockIdx.y * P * C;
    if (mean) { mean += (size_t)blockIdx.y * C; rstd += (size_t)blockIdx.y * C; }
    if (c12) c12 += (size_t)blockIdx.y * 2 * C;
    float a[NE], b[NE], mu[NE], rs[NE], c1[NE], c2[NE];
    ld_consts<NE>(mean, col, 0.0f, mu);
    ld_consts<NE>(rstd, col, 1.0f, rs);
    ld_consts<NE>(beta, col, 0.0f, b);
    ld_consts<NE>(gamma, col, 1.0f, a);
    ld_consts<NE>(c12, col, 0.0f, c1);
    ld_consts<NE>(c12 ? c12 + C : nullptr, col, 0.0f, c2);
#pragma unroll
    for (int i = 0; i < NE; ++i) a[i] = a[i] * rs[i];
    struct Stage { Chunk<TI, NE> v[BN_BWD_U]; Chunk<TG, NE> g[BN_BWD_U]; };
    const TI *xc = x + col;
    const TG *gc = gy + col;
    TI *dc = dx + col;
    pipelined_rows<BN_BWD_U, Stage>(
        (long long)blockIdx.x * rpi + rsub, (long long)gridDim.x * rpi, P,
        [&](Stage &s, int u, long long r) { s.v[u].load(xc + r * C); s.g[u].load(gc + r * C); },
        [&](const Stage &s, int u, long long r) {
            float v[NE], g[NE];
            s.v[u].unpack(v);
            s.g[u].unpack(g);
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const float gg = bn_z(v[i], mu[i], a[i], b[i]) > 0.0f ? g[i] : g[i] * slope;
                v[i] = a[i] * (gg - c1[i] - (v[i] - mu[i]) * rs[i] * c2[i]);
            }
            Chunk<TI, NE>::store(dc + r * C, v);
        });
}

// K > 0: g lives only on each group's arg-max row; a thread walks whole groups so that gy and
// the arg-max bytes are read once per group; (group, k) walked as one pipeline like the forward.
// L > 1: a group is walked as L runs of K/L rows by L threads (few long groups, see the forward).
template <typename TI, typename TG, int U>
__global__ __launch_bounds__(BN_THREADS) void rowbn_bwd_apply_max_kernel(
    const TG *__restrict__ gy, const TI *__restrict__ x, const uint8_t *__restrict__ arg, long long Gp, int K,
    int C, const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
    const float *__restrict__ beta, float slope, const float *__restrict__ c12, TI *__restrict__ dx, int L) {
    constexpr int NE = (sizeof(TI) == 2 || sizeof(TG) == 2) ? 8 : 4;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;
    const int chunk = threadIdx.x % cpr, rsub = threadIdx.x / cpr, col = chunk * NE;
    if (rsub >= rpi) return;
    gy += (size_t)blockIdx.y * Gp * C;
    arg += (size_t)blockIdx.y * Gp * C;
    x += (size_t)blockIdx.y * Gp * K * C;
    dx += (size_t)blockIdx.y * Gp * K * C;
    if (mean) { mean += (size_t)blockIdx.y * C; rstd += (size_t)blockIdx.y * C; }
    if (c12) c12 += (size_t)blockIdx.y * 2 * C;
    float a[NE], b[NE], mu[NE], rs[NE], c1[NE], c2[NE];
    ld_consts<NE>(mean, col, 0.0f, mu);
    ld_consts<NE>(rstd, col, 1.0f, rs);
    ld_consts<NE>(beta, col, 0.0f, b);
    ld_consts<NE>(gamma, col, 1.0f, a);
    ld_consts<NE>(c12, col, 0.0f, c1);
    ld_consts<NE>(c12 ? c12 + C : nullptr, col, 0.0f, c2);
#pragma unroll
    for (int i = 0; i < NE; ++i) a[i] = a[i] * rs[i];
    Gp *= L;                           // runs of K / L rows; run r belongs to group r / L
    K /= L;
    const long long gstep = (long long)gridDim.x * rpi;
    long long grp = (long long)blockIdx.x * rpi + rsub;
    int k = 0;
    bool have = grp < Gp;
    Chunk<TI, NE> cur[U], nxt[U];
    Chunk<TG, NE> cgy, ngy;
    unsigned ak_lo = 0, ak_hi = 0, nk_lo = 0, nk_hi = 0;     // NE arg-max bytes
    const TI *xc = x + col;
    auto load_group = [&](long long r_, Chunk<TG, NE> &cg, unsigned &lo, unsigned &hi) {
        const long long g_ = r_ / L;
        cg.load(gy + g_ * C + col);
        if constexpr (NE == 8) {
            const uint2 t = *reinterpret_cast<const uint2 *>(arg + g_ * C + col);
            lo = t.x; hi = t.y;
        } else {
            lo = *reinterpret_cast<const unsigned *>(arg + g_ * C + col);
            hi = 0;
        }
    };
    if (have) {
        load_group(grp, cgy, ak_lo, ak_hi);
        ngy = cgy;
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u].load(xc + (grp * K + u) * C);
    }
    while (have) {
        int kn = k + U;
        long long grpn = grp;
        if (kn >= K) { kn = 0; grpn = grp + gstep; }
        const bool have_n = grpn < Gp;
        if (have_n) {
            if (kn == 0) load_group(grpn, ngy, nk_lo, nk_hi);
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u].load(xc + (grpn * K + kn + u) * C);
        }
        float g[NE];
        cgy.unpack(g);
        const int kb = (int)(grp % L) * K;                 // this run's first k inside its group
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float v[NE];
            cur[u].unpack(v);
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int ak = (int)(((i < 4 ? ak_lo : ak_hi) >> (8 * (i & 3))) & 0xffu);
                float gg = 0.0f;
                if (ak == kb + k + u) gg = bn_z(v[i], mu[i], a[i], b[i]) > 0.0f ? g[i] : g[i] * slope;
                v[i] = a[i] * (gg - c1[i] - (v[i] - mu[i]) * rs[i] * c2[i]);
            }
            Chunk<TI, NE>::store(dx + (grp * K + k + u) * C + col, v);
        }
        if (kn == 0) { cgy = ngy; ak_lo = nk_lo; ak_hi = nk_hi; }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        k = kn;
        grp = grpn;
        have = have_n;
    }
}

// workgroups for a row walk: >= `per_thread` rows per thread, at most `cap` workgroups
int row_blocks(long long rows, int rpi, int per_thread, int cap) {
    long long b = (rows + (long long)rpi * per_thread - 1) / ((long long)rpi * per_thread);
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
int stats_blocks(long long rows, int rpi) { return row_blocks(rows, rpi, TPG_BN_STATS_ROWS, BN_MAX_BLOCKS); }
bool bn_shape_ok(int dtype_a, int dtype_b, int C) {
    const int ne = (dtype_a == TPG_DTYPE_BF16 || dtype_b == TPG_DTYPE_BF16) ? 8 : 4;
    return C > 0 && C % ne == 0 && C / 4 <= BN_THREADS;   // <= 1024 channels (column-sum layout)
}
bool bn_dtype_ok(int d) { return d == TPG_DTYPE_F32 || d == TPG_DTYPE_BF16; }
// workgroups per segment of a reduction when nseg segments share the launch.  (Alone, 2-3
// segments run 17-22 % faster with the full count each -- tools/tune_rowbn.py -- but the step
// gets slower, 66.4 vs 67.3 steps/s on one box: its three branches share the chip, and a launch
// that fills every CU twice over stalls the other two.)
#ifndef TPG_BN_SEG_DIV
#define TPG_BN_SEG_DIV 4      // most segments a launch's workgroup budget is divided by
#endif
int seg_blocks(int blocks, int nseg) {
    const int d = nseg > TPG_BN_SEG_DIV ? TPG_BN_SEG_DIV : nseg;
    const int b = (blocks + d - 1) / d;
    return b < 1 ? 1 : b;
}
// most workgroups PER SEGMENT of a max-variant launch: TPG_BN_GROUP_CAP / min(nseg, TPG_BN_SEG_DIV), twice that for one
// to three segments (tools/tune_rowbn.py, forward apply + max: one segment 17.9 -> 15.9 us, three 55.4 -> 47.5, six
// 85.6 -> 89.1 with the doubled budget, hence not there)
int group_cap(int nseg, int cap_div) { return (nseg <= 3 ? 2 * TPG_BN_GROUP_CAP : TPG_BN_GROUP_CAP) / cap_div; }
int group_unroll(int K) { return K % 4 == 0 ? 4 : (K % 2 == 0 ? 2 : 1); }
// Runs per group for the max-variant walks: 1 unless the launch would have fewer than ~64 K
// threads at work; then the largest power of two that keeps runs of >= 4 rows (a multiple of
// the pipeline depth) and, for the forward, the runs' partial results inside the workspace
// (`room` runs per segment; 0 = no limit).
int group_runs(long long Gp, int K, int cpr, int nseg, long long room) {
    int L = 1;
    while (K % (2 * L) == 0 && (K / (2 * L)) % 4 == 0 && Gp * L * cpr * nseg < 65536 &&
           (room == 0 || Gp * 2 * L <= room))
        L *= 2;
    return L;
}

}  // namespace

extern "C" size_t tpg_rowbn_workspace_bytes(int C, int nseg) {
    if (nseg < 1) nseg = 1;
    return sizeof(float) * (WS_HEAD + (size_t)nseg * ((size_t)BN_MAX_BLOCKS * 2 * C + 2 * (size_t)C));
}

namespace {
// the two launches of the batch statistics (P = rows per segment); ci_out: see rowbn_stats_finalize_kernel
void launch_stats(const void *x, int dtype_in, long long P, int C, float eps, float momentum, float *running_mean,
                  float *running_var, long long *num_batches_tracked, const float *mean_shift, float *mean, float *rstd,
                  float *wsf, int nseg, const float *gamma, const float *beta, float *ci_out, hipStream_t st) {
    // statistics use the INPUT type's vector width
    const int ne = dtype_in == TPG_DTYPE_BF16 ? 8 : 4;
    const int rpi = BN_THREADS / (C / ne);
    const int G = seg_blocks(stats_blocks(P, rpi), nseg);
    const dim3 fg((C + FIN_CH - 1) / FIN_CH);
    if (dtype_in == TPG_DTYPE_BF16) {
        const __hip_bfloat16 *xx = static_cast<const __hip_bfloat16 *>(x);
        hipLaunchKernelGGL(rowbn_stats_kernel<__hip_bfloat16>, dim3(G, nseg), dim3(BN_THREADS), 0, st, xx, P, C, wsf);
        hipLaunchKernelGGL(rowbn_stats_finalize_kernel<__hip_bfloat16>, fg, dim3(FIN_THREADS), 0, st, xx, wsf, G, P,
                           C, eps, momentum, running_mean, running_var, num_batches_tracked, mean_shift, mean, rstd, nseg,
                           gamma, beta, ci_out);
    } else {
        const float *xx = static_cast<const float *>(x);
        hipLaunchKernelGGL(rowbn_stats_kernel<float>, dim3(G, nseg), dim3(BN_THREADS), 0, st, xx, P, C, wsf);
        hipLaunchKernelGGL(rowbn_stats_finalize_kernel<float>, fg, dim3(FIN_THREADS), 0, st, xx, wsf, G, P, C, eps,
                           momentum, running_mean, running_var, num_batches_tracked, mean_shift, mean, rstd, nseg, gamma,
                           beta, ci_out);
    }
}
}  // namespace

// Training-mode batch statistics of x alone (phase STATS of tpg_rowbn_fwd) and, from the same finalize launch, the
// folded constants ci (nseg,4,C) = sc | sh | mu | rs that tpg_mlp_consts would make of them.
extern "C" int tpg_rowbn_stats_consts(const void *x, int dtype_in, long long P, int C, float eps, float momentum,
                                      float *running_mean, float *running_var, long long *num_batches_tracked,
                                      const float *mean_shift, const float *gamma, const float *beta, float *mean,
                                      float *rstd, float *ci, void *ws, int nseg, void *stream) {
    if (P <= 0 || C <= 0 || nseg < 1 || nseg > 65535 || P % nseg) return TPG_ERR_ARG;
    if (!x || !ws || !mean || !rstd || !ci) return TPG_ERR_ARG;
    if (!bn_dtype_ok(dtype_in) || !bn_shape_ok(dtype_in, dtype_in, C)) return TPG_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ws)) & 15) return TPG_ERR_UNSUPPORTED;
    launch_stats(x, dtype_in, P / nseg, C, eps, momentum, running_mean, running_var, num_batches_tracked, mean_shift, mean,
                 rstd, static_cast<float *>(ws), nseg, gamma, beta, ci, tpg_stream(stream));
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_rowbn_fwd(const void *x, int dtype_in, long long P, int K, int C, float eps, float momentum,
                             int training, float *running_mean, float *running_var,
                             long long *num_batches_tracked, const float *mean_shift, const float *gamma,
                             const float *beta, float slope,
                             float *mean, float *rstd, void *y, int dtype_out, uint8_t *argmax, void *ws, int nseg,
                             int phase, void *stream) {
    if (P <= 0 || C <= 0 || K < 0 || K > 256 || nseg < 1 || nseg > 65535 || P % nseg) return TPG_ERR_ARG;
    P /= nseg;                                    // rows per segment from here on
    if (K > 0 && P % K) return TPG_ERR_ARG;
    if (!x || !y || !ws || (K > 0 && !argmax)) return TPG_ERR_ARG;
    if (training ? (!mean || !rstd) : ((mean == nullptr) != (rstd == nullptr))) return TPG_ERR_ARG;
    if (!bn_dtype_ok(dtype_in) || !bn_dtype_ok(dtype_out) || !bn_shape_ok(dtype_in, dtype_out, C))
        return TPG_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(ws)) & 15)
        return TPG_ERR_UNSUPPORTED;
    hipStream_t st = tpg_stream(stream);
    float *wsf = static_cast<float *>(ws);
    if (training && phase != TPG_BN_PHASE_APPLY)
        launch_stats(x, dtype_in, P, C, eps, momentum, running_mean, running_var, num_batches_tracked, mean_shift, mean,
                     rstd, wsf, nseg, nullptr, nullptr, nullptr, st);
    // eval mode: the caller has filled mean / rstd from the running statistics
    if (phase == TPG_BN_PHASE_STATS) {
        TPG_RETURN_IF_LAUNCH_FAILED();
        return TPG_OK;
    }
    const int ne = (dtype_in == TPG_DTYPE_BF16 || dtype_out == TPG_DTYPE_BF16) ? 8 : 4;
    const int rpi_a = BN_THREADS / (C / ne);
    const long long rows_out = K > 0 ? P / K : P;
    // several segments share the chip: fewer workgroups per segment
    const int cap_div = nseg > TPG_BN_SEG_DIV ? TPG_BN_SEG_DIV : nseg;
    // runs' partial maxima live where the statistics' partial sums were (already consumed):
    // (float + byte) per channel and run, BN_MAX_BLOCKS * 2 floats per channel and segment
    const int L = K > 0 ? group_runs(rows_out, K, C / ne, nseg, BN_MAX_BLOCKS * 8 / 5) : 1;
    const dim3 g(K > 0 ? row_blocks(rows_out * L, rpi_a, 1, group_cap(nseg, cap_div))
                       : row_blocks(P, rpi_a, TPG_BN_APPLY_ROWS, TPG_BN_APPLY_CAP / cap_div), nseg);
    const dim3 blk(BN_THREADS);
    const int gu = group_unroll(K / L);
    float *part = static_cast<float *>(ws) + WS_HEAD;
    uint8_t *parg = reinterpret_cast<uint8_t *>(part + (size_t)nseg * rows_out * L * C);
#define TPG_BN_APPLY_MAX(TI, TO, U)                                                                       \
    do {                                                                                                  \
        hipLaunchKernelGGL((rowbn_apply_max_kernel<TI, TO, U>), g, blk, 0, st, static_cast<const TI *>(x), \
                           rows_out, K, C, mean, rstd, gamma, beta, slope, static_cast<TO *>(y), argmax, L, part, \
                           parg);                                                                         \
        if (L > 1) {                                                                                      \
            constexpr int NE_ = (sizeof(TI) == 2 || sizeof(TO) == 2) ? 8 : 4;                             \
            const long long thr = rows_out * (C / NE_);                                                   \
            hipLaunchKernelGGL((rowbn_max_combine_kernel<TO, NE_>), dim3((unsigned)((thr + BN_THREADS - 1) / BN_THREADS), nseg), \
                               blk, 0, st, part, parg, rows_out, L, C, static_cast<TO *>(y), argmax);     \
        }                                                                                                 \
    } while (0)
#define TPG_BN_APPLY(TI, TO)                                                                              \
    do {                                                                                                  \
        if (K > 0) {                                                                                      \
            if (gu == 4) TPG_BN_APPLY_MAX(TI, TO, 4);                                                     \
            else if (gu == 2) TPG_BN_APPLY_MAX(TI, TO, 2);                                                \
            else TPG_BN_APPLY_MAX(TI, TO, 1);                                                             \
        } else                                                                                            \
            hipLaunchKernelGGL((rowbn_apply_kernel<TI, TO>), g, blk, 0, st, static_cast<const TI *>(x), P, C, mean, \
                               rstd, gamma, beta, slope, static_cast<TO *>(y));                           \
    } while (0)
    if (dtype_in == TPG_DTYPE_F32 && dtype_out == TPG_DTYPE_F32) TPG_BN_APPLY(float, float);
    else if (dtype_in == TPG_DTYPE_F32) TPG_BN_APPLY(float, __hip_bfloat16);
    else if (dtype_out == TPG_DTYPE_F32) TPG_BN_APPLY(__hip_bfloat16, float);
    else TPG_BN_APPLY(__hip_bfloat16, __hip_bfloat16);
#undef TPG_BN_APPLY
#undef TPG_BN_APPLY_MAX
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

namespace {
// tpg_rowbn_bwd; c12_out: where the finalize launch writes c12 (NULL: inside ws, where the apply phase reads it),
// cb_out: see rowbn_bwd_finalize_kernel
int rowbn_bwd_impl(const void *gy, int dtype_g, const void *x, int dtype_in, const uint8_t *argmax,
                   const void *y, int dtype_y, long long P, int K, int C, int training,
                   const float *mean, const float *rstd, const float *gamma, const float *beta,
                   float slope, float *dgamma, float *dbeta, void *dx, void *ws, int nseg, int phase,
                   float *c12_out, float *cb_out, void *ag_out, void *stream) {
    if (P <= 0 || C <= 0 || K < 0 || K > 256 || nseg < 1 || nseg > 65535 || P % nseg) return TPG_ERR_ARG;
    P /= nseg;                                    // rows per segment from here on
    if (K > 0 && P % K) return TPG_ERR_ARG;
    if (!gy || !x || !dx || !ws || (K > 0 && !argmax)) return TPG_ERR_ARG;
    if ((mean == nullptr) != (rstd == nullptr) || (training && !mean)) return TPG_ERR_ARG;
    if (!bn_dtype_ok(dtype_in) || !bn_dtype_ok(dtype_g) || !bn_shape_ok(dtype_in, dtype_g, C))
        return TPG_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(dx) |
         reinterpret_cast<uintptr_t>(ws)) & 15)
        return TPG_ERR_UNSUPPORTED;
    // the (gy, y) form of the max-variant sums needs y stored like gy, 16-byte aligned
    if (!(K > 0 && y && dtype_y == dtype_g && !(reinterpret_cast<uintptr_t>(y) & 15))) y = nullptr;
    hipStream_t st = tpg_stream(stream);
    float *wsf = static_cast<float *>(ws);
    float *c12 = c12_out ? c12_out : wsf + WS_HEAD + (size_t)nseg * BN_MAX_BLOCKS * 2 * C;      // (nseg, 2, C)
    const int ne = (dtype_in == TPG_DTYPE_BF16 || dtype_g == TPG_DTYPE_BF16) ? 8 : 4;
    const int rpi = BN_THREADS / (C / ne);
    const long long rows_g = K > 0 ? P / K : P;
    const int G = seg_blocks(K > 0 ? row_blocks(rows_g, rpi, TPG_BN_YRED_ROWS, BN_MAX_BLOCKS) : stats_blocks(rows_g, rpi), nseg);
    const int cap_div = nseg > TPG_BN_SEG_DIV ? TPG_BN_SEG_DIV : nseg;
    const int L = K > 0 ? group_runs(rows_g, K, C / ne, nseg, 0) : 1;
    const int GA = K > 0 ? row_blocks(rows_g * L, rpi, 1, group_cap(nseg, cap_div))
                         : row_blocks(P, rpi, TPG_BN_APPLY_ROWS, TPG_BN_APPLY_CAP / cap_div);
    // no batch statistics and no affine gradients wanted (pure activation [+max]): dx = a * g,
    // nothing to reduce
    const bool wanted = training || dgamma || dbeta;
    const bool need_reduce = wanted && phase != TPG_BN_PHASE_APPLY;
    const bool do_apply = phase != TPG_BN_PHASE_STATS;
    const float *c12_arg = wanted ? c12 : nullptr;     // absent constants read as zero
    const int gu = group_unroll(K / L);
#define TPG_BN_BWD_APPLY_MAX(TI, TG, U)                                                                     \
    hipLaunchKernelGGL((rowbn_bwd_apply_max_kernel<TI, TG, U>), dim3(GA, nseg), dim3(BN_THREADS), 0, st, gg, xx, argmax, \
                       rows_g, K, C, mean, rstd, gamma, beta, slope, c12_arg, static_cast<TI *>(dx), L)
#define TPG_BN_BWD(TI, TG)                                                                                  \
    do {                                                                                                    \
        const TI *xx = static_cast<const TI *>(x);                                                          \
        const TG *gg = static_cast<const TG *>(gy);                                                         \
        if (!need_reduce) {                                                                                 \
        } else if (K > 0)                                                                                   \
            hipLaunchKernelGGL((rowbn_bwd_reduce_max_kernel<TI, TG>), dim3(G, nseg), dim3(BN_THREADS), 0, st, gg, \
                               static_cast<const TG *>(y), xx, argmax, rows_g, K, C, mean, rstd, gamma, beta, \
                               slope, wsf, y ? static_cast<TG *>(ag_out) : nullptr);                        \
        else                                                                                                \
            hipLaunchKernelGGL((rowbn_bwd_reduce_kernel<TI, TG>), dim3(G, nseg), dim3(BN_THREADS), 0, st, gg, xx, P, C, \
                               mean, rstd, gamma, beta, slope, wsf);                                        \
        if (need_reduce)                                                                                    \
            hipLaunchKernelGGL(rowbn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_THREADS), 0, st, \
                               wsf, G, P, C, training, dgamma, dbeta, c12, nseg, mean, rstd, gamma, cb_out); \
        if (!do_apply) {                                                                                    \
        } else if (K > 0) {                                                                                 \
            if (gu == 4) TPG_BN_BWD_APPLY_MAX(TI, TG, 4);                                                   \
            else if (gu == 2) TPG_BN_BWD_APPLY_MAX(TI, TG, 2);                                              \
            else TPG_BN_BWD_APPLY_MAX(TI, TG, 1);                                                           \
        } else                                                                                              \
            hipLaunchKernelGGL((rowbn_bwd_apply_kernel<TI, TG>), dim3(GA, nseg), dim3(BN_THREADS), 0, st, gg, xx, P, C, \
                               mean, rstd, gamma, beta, slope, c12_arg, static_cast<TI *>(dx));            \
    } while (0)
    if (dtype_in == TPG_DTYPE_F32 && dtype_g == TPG_DTYPE_F32) TPG_BN_BWD(float, float);
    else if (dtype_in == TPG_DTYPE_F32) TPG_BN_BWD(float, __hip_bfloat16);
    else if (dtype_g == TPG_DTYPE_F32) TPG_BN_BWD(__hip_bfloat16, float);
    else TPG_BN_BWD(__hip_bfloat16, __hip_bfloat16);
#undef TPG_BN_BWD
#undef TPG_BN_BWD_APPLY_MAX
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}
}  // namespace

extern "C" int tpg_rowbn_bwd(const void *gy, int dtype_g, const void *x, int dtype_in, const uint8_t *argmax,
                             const void *y, int dtype_y, long long P, int K, int C, int training,
                             const float *mean, const float *rstd, const float *gamma, const float *beta,
                             float slope, float *dgamma, float *dbeta, void *dx, void *ws, int nseg, int phase,
                             void *stream) {
    return rowbn_bwd_impl(gy, dtype_g, x, dtype_in, argmax, y, dtype_y, P, K, C, training, mean, rstd, gamma, beta, slope,
                          dgamma, dbeta, dx, ws, nseg, phase, nullptr, nullptr, nullptr, stream);
}

// The reduction half of tpg_rowbn_bwd alone, with its per-channel results handed to the caller:
// c12 (nseg,2,C) = (sum gg / P | sum gg*xhat / P) per segment, dgamma / dbeta (C, may be NULL).  The
// fused MLP tail (csrc/mlp_fused.hip) folds these into the constants of its data- and weight-gradient
// prologues instead of materialising dx.
extern "C" int tpg_rowbn_bwd_sums(const void *gy, int dtype_g, const void *x, int dtype_in, const uint8_t *argmax,
                                  const void *y, int dtype_y, long long P, int K, int C, int training,
                                  const float *mean, const float *rstd, const float *gamma, const float *beta,
                                  float slope, float *dgamma, float *dbeta, float *c12, void *ws, int nseg,
                                  void *stream) {
    if (!c12 || nseg < 1) return TPG_ERR_ARG;
    // (dx is not written in the statistics phase; any non-null pointer passes the argument check)
    // the finalize launch writes c12 where the caller wants it (it used to land in ws and be copied out)
    return rowbn_bwd_impl(gy, dtype_g, x, dtype_in, argmax, y, dtype_y, P, K, C, training, mean, rstd, gamma, beta, slope,
                          dgamma, dbeta, const_cast<void *>(x), ws, nseg, TPG_BN_PHASE_STATS, c12, nullptr, nullptr, stream);
}

// The same, and from the same finalize launch cb (nseg,4,C) = a | f*mu | e | f: what tpg_mlp_consts(..., c12, cb)
// would fold from mean, rstd, gamma and c12 for the fused tail's gradient kernels.  ag, optional, K > 0 with y of
// gy's type: (P/K, C) of gy's type = a * lrelu'(y) * gy, tpg_mlp_max_prep's output from the rows the reduction reads
// anyway.  Returns TPG_ERR_UNSUPPORTED if ag is asked for and y cannot be used (then call tpg_mlp_max_prep).
extern "C" int tpg_rowbn_bwd_sums_consts(const void *gy, int dtype_g, const void *x, int dtype_in, const uint8_t *argmax,
                                         const void *y, int dtype_y, long long P, int K, int C, int training,
                                         const float *mean, const float *rstd, const float *gamma, const float *beta,
                                         float slope, float *dgamma, float *dbeta, float *c12, float *cb, void *ag,
                                         void *ws, int nseg, void *stream) {
    if (!c12 || !cb || nseg < 1) return TPG_ERR_ARG;
    if (ag && !(K > 0 && y && dtype_y == dtype_g && training && !((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(ag)) & 15)))
        return TPG_ERR_UNSUPPORTED;
    return rowbn_bwd_impl(gy, dtype_g, x, dtype_in, argmax, y, dtype_y, P, K, C, training, mean, rstd, gamma, beta, slope,
                          dgamma, dbeta, const_cast<void *>(x), ws, nseg, TPG_BN_PHASE_STATS, c12, cb, ag, stream);
}
