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
// i.e. 2+3 streaming passes at HBM rate instead of PyTorch's BatchNorm (stats + transform),
// LeakyReLU, max-reduce and their three backward kernels.  Rows are 16-byte vectors per lane
// (fp32 or bf16 storage, fp32 arithmetic); column sums use per-thread register accumulators
// over a grid-stride of rows, one LDS tree per workgroup, and a deterministic two-stage
// reduction (per-workgroup partials -> one finalize workgroup in fp64), so statistics are
// bitwise reproducible run to run and there are no float atomics.  Sums are taken about a
// per-channel pivot (the first row) to keep E[x^2]-E[x]^2 well conditioned.
#include <hip/hip_bf16.h>

#include "tpg_common.hpp"

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_BLOCKS = 512;   // partial rows per reduction (2 workgroups per CU)

// ---- 16-byte row chunks <-> NE floats (same helpers as rowgather.hip) ----------------------
template <typename T, int NE> struct BnIO;
template <int NE> struct BnIO<float, NE> {
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
    static __device__ __forceinline__ float one(const float *p) { return *p; }
};
template <> struct BnIO<__hip_bfloat16, 8> {
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

__device__ __forceinline__ float lrelu_f(float z, float slope) { return z > 0.0f ? z : z * slope; }
// NULL statistics = identity (mean 0, rstd 1)
__device__ __forceinline__ float ld_mean(const float *mean, int c) { return mean ? mean[c] : 0.0f; }
__device__ __forceinline__ float ld_rstd(const float *rstd, int c) { return rstd ? rstd[c] : 1.0f; }
// pre-activation, written ONCE so that forward and backward see the same sign
__device__ __forceinline__ float bn_z(float v, float mu, float a, float beta) { return (v - mu) * a + beta; }

// Thread layout shared by every kernel here: a thread owns ONE 16-byte column chunk
// (chunk = tid % cpr) and walks rows rsub, rsub + rpi, ... (rsub = tid / cpr, rpi = 256 / cpr rows
// per workgroup iteration), so per-channel constants are loaded once and stay in registers.
//
// Column reduction of NQ quantities per channel: every thread has acc[NQ][NE] for its chunk;
// all 256 threads take part: output o = (q, i, chunk) sums its rpi partners from LDS.
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

constexpr int BN_UNROLL = 4;  // independent 16-byte loads in flight per thread

// ------------------------------------------------------------------ forward statistics
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void rowbn_stats_kernel(const T *__restrict__ x, long long P, int C,
                                                                 float *__restrict__ part) {
    constexpr int NE = BnElems<T>::NE;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;  // chunks per row, rows per iteration
    const int tid = threadIdx.x;
    const int chunk = tid % cpr, rsub = tid / cpr;
    float acc[2][NE];
    float piv[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) { acc[0][i] = 0.0f; acc[1][i] = 0.0f; }
    if (rsub < rpi) {
        BnIO<T, NE>::load(x + chunk * NE, piv);  // pivot = first row
        const long long step = (long long)gridDim.x * rpi;
        long long r = (long long)blockIdx.x * rpi + rsub;
        for (; r + (BN_UNROLL - 1) * step < P; r += BN_UNROLL * step) {
            float v[BN_UNROLL][NE];
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) BnIO<T, NE>::load(x + (r + u * step) * C + chunk * NE, v[u]);
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u)
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const float d = v[u][i] - piv[i];
                    acc[0][i] += d;
                    acc[1][i] += d * d;
                }
        }
        for (; r < P; r += step) {
            float v[NE];
            BnIO<T, NE>::load(x + r * C + chunk * NE, v);
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const float d = v[i] - piv[i];
                acc[0][i] += d;
                acc[1][i] += d * d;
            }
        }
    }
    block_column_reduce<2, NE>(acc, cpr, rpi, C, part);
}

// Sum the per-workgroup partials of two quantities: a workgroup owns FIN_CH channels, each
// summed by 256/FIN_CH lanes over the partial rows (a handful of independent loads per lane,
// not a 64-deep dependent chain), fp64, one LDS step.  True in the lane that owns channel c.
constexpr int FIN_CH = 4;
constexpr int FIN_LANES = BN_THREADS / FIN_CH;
__device__ __forceinline__ bool finalize_sums(const float *__restrict__ part, int G, int C, int &c, double &s0,
                                              double &s1) {
    __shared__ double red[2][FIN_LANES][FIN_CH];
    const int cl = threadIdx.x % FIN_CH, gl = threadIdx.x / FIN_CH;
    c = blockIdx.x * FIN_CH + cl;
    double a0 = 0.0, a1 = 0.0;
    if (c < C)
        for (int g = gl; g < G; g += FIN_LANES) {
            a0 += part[((size_t)g * 2 + 0) * C + c];
            a1 += part[((size_t)g * 2 + 1) * C + c];
        }
    red[0][gl][cl] = a0;
    red[1][gl][cl] = a1;
    __syncthreads();
    // tree over the FIN_LANES partial sums of each channel
    for (int sft = FIN_LANES / 2; sft > 0; sft >>= 1) {
        if (gl < sft) {
            red[0][gl][cl] += red[0][gl + sft][cl];
            red[1][gl][cl] += red[1][gl + sft][cl];
        }
        __syncthreads();
    }
    if (gl != 0 || c >= C) return false;
    s0 = red[0][0][cl];
    s1 = red[1][0][cl];
    return true;
}

// partials -> mean, rstd (+ running-stat update as nn.BatchNorm does); grid = ceil(C/FIN_CH)
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void rowbn_stats_finalize_kernel(
    const T *__restrict__ x, const float *__restrict__ part, int G, long long P, int C, float eps,
    float momentum, float *__restrict__ running_mean, float *__restrict__ running_var,
    long long *__restrict__ num_batches_tracked, float *__restrict__ mean, float *__restrict__ rstd) {
    int c;
    double s, ss;
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    if (!finalize_sums(part, G, C, c, s, ss)) return;
    const double piv = BnIO<T, BnElems<T>::NE>::one(x + c);
    const double m = s / (double)P;
    double var = ss / (double)P - m * m;  // biased, about the pivot
    var = var < 0.0 ? 0.0 : var;
    mean[c] = (float)(piv + m);
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = P > 1 ? var * (double)P / (double)(P - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * (piv + m));
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
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
    float a[NE], b[NE], mu[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        mu[i] = ld_mean(mean, col + i);
        a[i] = (gamma ? gamma[col + i] : 1.0f) * ld_rstd(rstd, col + i);
        b[i] = beta ? beta[col + i] : 0.0f;
    }
    const long long step = (long long)gridDim.x * rpi;
    long long r = (long long)blockIdx.x * rpi + rsub;
    for (; r + (BN_UNROLL - 1) * step < P; r += BN_UNROLL * step) {
        float v[BN_UNROLL][NE];
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) BnIO<TI, NE>::load(x + (r + u * step) * C + col, v[u]);
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) {
#pragma unroll
            for (int i = 0; i < NE; ++i) v[u][i] = lrelu_f(bn_z(v[u][i], mu[i], a[i], b[i]), slope);
            BnIO<TO, NE>::store(y + (r + u * step) * C + col, v[u]);
        }
    }
    for (; r < P; r += step) {
        float v[NE];
        BnIO<TI, NE>::load(x + r * C + col, v);
#pragma unroll
        for (int i = 0; i < NE; ++i) v[i] = lrelu_f(bn_z(v[i], mu[i], a[i], b[i]), slope);
        BnIO<TO, NE>::store(y + r * C + col, v);
    }
}

// groups of K consecutive rows -> one row (max) + arg-max byte per channel (first maximum)
template <typename TI, typename TO>
__global__ __launch_bounds__(BN_THREADS) void rowbn_apply_max_kernel(
    const TI *__restrict__ x, long long Gp, int K, int C, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    float slope, TO *__restrict__ y, uint8_t *__restrict__ arg) {
    constexpr int NE = (sizeof(TI) == 2 || sizeof(TO) == 2) ? 8 : 4;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;
    const int chunk = threadIdx.x % cpr, rsub = threadIdx.x / cpr, col = chunk * NE;
    if (rsub >= rpi) return;
    float a[NE], b[NE], mu[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        mu[i] = ld_mean(mean, col + i);
        a[i] = (gamma ? gamma[col + i] : 1.0f) * ld_rstd(rstd, col + i);
        b[i] = beta ? beta[col + i] : 0.0f;
    }
    for (long long grp = (long long)blockIdx.x * rpi + rsub; grp < Gp; grp += (long long)gridDim.x * rpi) {
        float best[NE];
        int bk[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) { best[i] = -INFINITY; bk[i] = 0; }
        const TI *xg = x + grp * K * C + col;
        int k = 0;
        for (; k + BN_UNROLL <= K; k += BN_UNROLL) {
            float v[BN_UNROLL][NE];
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) BnIO<TI, NE>::load(xg + (size_t)(k + u) * C, v[u]);
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u)
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const float z = lrelu_f(bn_z(v[u][i], mu[i], a[i], b[i]), slope);
                    if (z > best[i]) { best[i] = z; bk[i] = k + u; }
                }
        }
        for (; k < K; ++k) {
            float v[NE];
            BnIO<TI, NE>::load(xg + (size_t)k * C, v);
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const float z = lrelu_f(bn_z(v[i], mu[i], a[i], b[i]), slope);
                if (z > best[i]) { best[i] = z; bk[i] = k; }
            }
        }
        BnIO<TO, NE>::store(y + grp * C + col, best);
        // NE arg-max bytes (8-byte aligned for NE = 8, 4-byte for NE = 4)
        if constexpr (NE == 8) {
            const unsigned lo = bk[0] | (bk[1] << 8) | (bk[2] << 16) | ((unsigned)bk[3] << 24);
            const unsigned hi = bk[4] | (bk[5] << 8) | (bk[6] << 16) | ((unsigned)bk[7] << 24);
            *reinterpret_cast<uint2 *>(arg + grp * C + col) = make_uint2(lo, hi);
        } else {
            *reinterpret_cast<unsigned *>(arg + grp * C + col) =
                bk[0] | (bk[1] << 8) | (bk[2] << 16) | ((unsigned)bk[3] << 24);
        }
    }
}

// ------------------------------------------------------------------ backward reductions
// part[(block*2+0)*C + c] = sum g, part[(block*2+1)*C + c] = sum g*xhat
template <typename TI, typename TG>
__global__ __launch_bounds__(BN_THREADS) void rowbn_bwd_reduce_kernel(
    const TG *__restrict__ gy, const TI *__restrict__ x, long long P, int C, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    float slope, float *__restrict__ part) {
    constexpr int NE = (sizeof(TI) == 2 || sizeof(TG) == 2) ? 8 : 4;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;
    const int tid = threadIdx.x;
    const int chunk = tid % cpr, rsub = tid / cpr;
    float acc[2][NE], a[NE], b[NE], mu[NE], rs[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        acc[0][i] = 0.0f; acc[1][i] = 0.0f;
        const int c = chunk * NE + i;
        mu[i] = ld_mean(mean, c); rs[i] = ld_rstd(rstd, c);
        a[i] = (gamma ? gamma[c] : 1.0f) * rs[i];
        b[i] = beta ? beta[c] : 0.0f;
    }
    if (rsub < rpi) {
        const long long step = (long long)gridDim.x * rpi;
        long long r = (long long)blockIdx.x * rpi + rsub;
        constexpr int U = 2;
        for (; r + (U - 1) * step < P; r += U * step) {
            float v[U][NE], g[U][NE];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                BnIO<TI, NE>::load(x + (r + u * step) * C + chunk * NE, v[u]);
                BnIO<TG, NE>::load(gy + (r + u * step) * C + chunk * NE, g[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const float gg = bn_z(v[u][i], mu[i], a[i], b[i]) > 0.0f ? g[u][i] : g[u][i] * slope;
                    acc[0][i] += gg;
                    acc[1][i] += gg * ((v[u][i] - mu[i]) * rs[i]);
                }
        }
        for (; r < P; r += step) {
            float v[NE], g[NE];
            BnIO<TI, NE>::load(x + r * C + chunk * NE, v);
            BnIO<TG, NE>::load(gy + r * C + chunk * NE, g);
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const float gg = bn_z(v[i], mu[i], a[i], b[i]) > 0.0f ? g[i] : g[i] * slope;
                acc[0][i] += gg;
                acc[1][i] += gg * ((v[i] - mu[i]) * rs[i]);
            }
        }
    }
    block_column_reduce<2, NE>(acc, cpr, rpi, C, part);
}

// max variant: gy is (P/K, C); only the arg-max row of each group carries gradient.
// With the forward's output y (stored like gy) the arg-max row's pre-activation is
// z = y > 0 ? y : y / slope and xhat = (z - beta) / gamma: two small streaming reads instead of
// one scattered 2/4-byte load of x per (group, channel).  Per-channel guard: the inversion is
// used only where it is well conditioned.
template <typename TI, typename TG>
__global__ __launch_bounds__(BN_THREADS) void rowbn_bwd_reduce_max_y_kernel(
    const TG *__restrict__ gy, const TG *__restrict__ y, const TI *__restrict__ x,
    const uint8_t *__restrict__ arg, long long Gp, int K, int C, const float *__restrict__ mean,
    const float *__restrict__ rstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    float slope, float *__restrict__ part) {
    constexpr int NE = (sizeof(TI) == 2 || sizeof(TG) == 2) ? 8 : 4;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;
    const int tid = threadIdx.x;
    const int chunk = tid % cpr, rsub = tid / cpr;
    float acc[2][NE], a[NE], b[NE], mu[NE], rs[NE], ig[NE];
    bool inv[NE];
    const float islope = slope != 0.0f ? 1.0f / slope : 0.0f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        acc[0][i] = 0.0f; acc[1][i] = 0.0f;
        const int c = chunk * NE + i;
        mu[i] = ld_mean(mean, c); rs[i] = ld_rstd(rstd, c);
        const float gm = gamma ? gamma[c] : 1.0f;
        a[i] = gm * rs[i];
        b[i] = beta ? beta[c] : 0.0f;
        inv[i] = fabsf(b[i]) <= 4.0f * fabsf(gm);
        ig[i] = inv[i] ? 1.0f / gm : 0.0f;
    }
    if (rsub < rpi) {
        for (long long r = (long long)blockIdx.x * rpi + rsub; r < Gp; r += (long long)gridDim.x * rpi) {
            float g[NE], yy[NE];
            BnIO<TG, NE>::load(gy + r * C + chunk * NE, g);
            BnIO<TG, NE>::load(y + r * C + chunk * NE, yy);
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int c = chunk * NE + i;
                if (inv[i]) {
                    const bool pos = yy[i] > 0.0f;
                    const float gg = pos ? g[i] : g[i] * slope;
                    const float z = pos ? yy[i] : yy[i] * islope;
                    acc[0][i] += gg;
                    acc[1][i] += gg * ((z - b[i]) * ig[i]);
                } else {
                    const int k = arg[r * C + c];
                    const float v = BnIO<TI, NE>::one(x + (r * K + k) * C + c);
                    const float gg = bn_z(v, mu[i], a[i], b[i]) > 0.0f ? g[i] : g[i] * slope;
                    acc[0][i] += gg;
                    acc[1][i] += gg * ((v - mu[i]) * rs[i]);
                }
            }
        }
    }
    block_column_reduce<2, NE>(acc, cpr, rpi, C, part);
}

template <typename TI, typename TG>
__global__ __launch_bounds__(BN_THREADS) void rowbn_bwd_reduce_max_kernel(
    const TG *__restrict__ gy, const TI *__restrict__ x, const uint8_t *__restrict__ arg, long long Gp,
    int K, int C, const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const float *__restrict__ beta, float slope, float *__restrict__ part) {
    constexpr int NE = (sizeof(TI) == 2 || sizeof(TG) == 2) ? 8 : 4;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;
    const int tid = threadIdx.x;
    const int chunk = tid % cpr, rsub = tid / cpr;
    float acc[2][NE], a[NE], b[NE], mu[NE], rs[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        acc[0][i] = 0.0f; acc[1][i] = 0.0f;
        const int c = chunk * NE + i;
        mu[i] = ld_mean(mean, c); rs[i] = ld_rstd(rstd, c);
        a[i] = (gamma ? gamma[c] : 1.0f) * rs[i];
        b[i] = beta ? beta[c] : 0.0f;
    }
    if (rsub < rpi) {
        for (long long r = (long long)blockIdx.x * rpi + rsub; r < Gp; r += (long long)gridDim.x * rpi) {
            float g[NE];
            BnIO<TG, NE>::load(gy + r * C + chunk * NE, g);
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int c = chunk * NE + i;
                const int k = arg[r * C + c];
                const float v = BnIO<TI, NE>::one(x + (r * K + k) * C + c);
                const float gg = bn_z(v, mu[i], a[i], b[i]) > 0.0f ? g[i] : g[i] * slope;
                acc[0][i] += gg;
                acc[1][i] += gg * ((v - mu[i]) * rs[i]);
            }
        }
    }
    block_column_reduce<2, NE>(acc, cpr, rpi, C, part);
}

// partials -> dgamma, dbeta (fp32) and the two per-channel constants of dx; grid = ceil(C/FIN_CH)
__global__ __launch_bounds__(BN_THREADS) void rowbn_bwd_finalize_kernel(const float *__restrict__ part, int G,
                                                                        long long P, int C, int training,
                                                                        float *__restrict__ dgamma,
                                                                        float *__restrict__ dbeta,
                                                                        float *__restrict__ c12) {
    int c;
    double s, sx;
    if (!finalize_sums(part, G, C, c, s, sx)) return;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)sx;
    c12[c] = training ? (float)(s / (double)P) : 0.0f;       // eval-mode BN: no batch terms
    c12[C + c] = training ? (float)(sx / (double)P) : 0.0f;
}

// dx = a * (g - c1 - xhat*c2); K > 0: g lives only on each group's arg-max row
template <typename TI, typename TG>
__global__ __launch_bounds__(BN_THREADS) void rowbn_bwd_apply_kernel(
    const TG *__restrict__ gy, const TI *__restrict__ x, const uint8_t *__restrict__ arg, long long P, int K,
    int C, const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
    const float *__restrict__ beta, float slope, const float *__restrict__ c12, TI *__restrict__ dx) {
    constexpr int NE = (sizeof(TI) == 2 || sizeof(TG) == 2) ? 8 : 4;
    const int cpr = C / NE, rpi = BN_THREADS / cpr;
    const int chunk = threadIdx.x % cpr, rsub = threadIdx.x / cpr, col = chunk * NE;
    if (rsub >= rpi) return;
    float a[NE], b[NE], mu[NE], rs[NE], c1[NE], c2[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        mu[i] = ld_mean(mean, col + i);
        rs[i] = ld_rstd(rstd, col + i);
        a[i] = (gamma ? gamma[col + i] : 1.0f) * rs[i];
        b[i] = beta ? beta[col + i] : 0.0f;
        c1[i] = c12[col + i];
        c2[i] = c12[C + col + i];
    }
    const long long step = (long long)gridDim.x * rpi;
    if (K == 0) {
        constexpr int U = 2;
        long long r = (long long)blockIdx.x * rpi + rsub;
        for (; r + (U - 1) * step < P; r += U * step) {
            float v[U][NE], g[U][NE];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                BnIO<TI, NE>::load(x + (r + u * step) * C + col, v[u]);
                BnIO<TG, NE>::load(gy + (r + u * step) * C + col, g[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const float gg = bn_z(v[u][i], mu[i], a[i], b[i]) > 0.0f ? g[u][i] : g[u][i] * slope;
                    v[u][i] = a[i] * (gg - c1[i] - (v[u][i] - mu[i]) * rs[i] * c2[i]);
                }
                BnIO<TI, NE>::store(dx + (r + u * step) * C + col, v[u]);
            }
        }
        for (; r < P; r += step) {
            float v[NE], g[NE];
            BnIO<TI, NE>::load(x + r * C + col, v);
            BnIO<TG, NE>::load(gy + r * C + col, g);
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const float gg = bn_z(v[i], mu[i], a[i], b[i]) > 0.0f ? g[i] : g[i] * slope;
                v[i] = a[i] * (gg - c1[i] - (v[i] - mu[i]) * rs[i] * c2[i]);
            }
            BnIO<TI, NE>::store(dx + r * C + col, v);
        }
        return;
    }
    // K > 0: walk whole groups so that gy / arg-max are read once per group
    const long long Gp = P / K;
    for (long long grp = (long long)blockIdx.x * rpi + rsub; grp < Gp; grp += step) {
        float g[NE];
        int ak[NE];
        BnIO<TG, NE>::load(gy + grp * C + col, g);
#pragma unroll
        for (int i = 0; i < NE; ++i) ak[i] = arg[grp * C + col + i];
        int k = 0;
        for (; k + BN_UNROLL <= K; k += BN_UNROLL) {
            float v[BN_UNROLL][NE];
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) BnIO<TI, NE>::load(x + (grp * K + k + u) * C + col, v[u]);
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) {
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    float gg = 0.0f;
                    if (ak[i] == k + u) gg = bn_z(v[u][i], mu[i], a[i], b[i]) > 0.0f ? g[i] : g[i] * slope;
                    v[u][i] = a[i] * (gg - c1[i] - (v[u][i] - mu[i]) * rs[i] * c2[i]);
                }
                BnIO<TI, NE>::store(dx + (grp * K + k + u) * C + col, v[u]);
            }
        }
        for (; k < K; ++k) {
            float v[NE];
            BnIO<TI, NE>::load(x + (grp * K + k) * C + col, v);
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                float gg = 0.0f;
                if (ak[i] == k) gg = bn_z(v[i], mu[i], a[i], b[i]) > 0.0f ? g[i] : g[i] * slope;
                v[i] = a[i] * (gg - c1[i] - (v[i] - mu[i]) * rs[i] * c2[i]);
            }
            BnIO<TI, NE>::store(dx + (grp * K + k) * C + col, v);
        }
    }
}

// workgroups for a row walk: >= `per_thread` rows per thread, at most `cap` workgroups
int row_blocks(long long rows, int rpi, int per_thread, int cap) {
    long long b = (rows + (long long)rpi * per_thread - 1) / ((long long)rpi * per_thread);
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
int stats_blocks(long long rows, int rpi) { return row_blocks(rows, rpi, 2 * BN_UNROLL, BN_MAX_BLOCKS); }
bool bn_shape_ok(int dtype_a, int dtype_b, int C) {
    const int ne = (dtype_a == TPG_DTYPE_BF16 || dtype_b == TPG_DTYPE_BF16) ? 8 : 4;
    return C > 0 && C % ne == 0 && C / 4 <= BN_THREADS;   // <= 1024 channels (column-sum layout)
}
bool bn_dtype_ok(int d) { return d == TPG_DTYPE_F32 || d == TPG_DTYPE_BF16; }

}  // namespace

extern "C" size_t tpg_rowbn_workspace_bytes(int C) {
    return sizeof(float) * ((size_t)BN_MAX_BLOCKS * 2 * C + 2 * (size_t)C);
}

extern "C" int tpg_rowbn_fwd(const void *x, int dtype_in, long long P, int K, int C, float eps, float momentum,
                             int training, float *running_mean, float *running_var,
                             long long *num_batches_tracked, const float *gamma, const float *beta, float slope,
                             float *mean, float *rstd, void *y, int dtype_out, uint8_t *argmax, void *ws, int phase,
                             void *stream) {
    if (P <= 0 || C <= 0 || K < 0 || K > 256 || (K > 0 && P % K)) return TPG_ERR_ARG;
    if (!x || !y || !ws || (K > 0 && !argmax)) return TPG_ERR_ARG;
    if (training ? (!mean || !rstd) : ((mean == nullptr) != (rstd == nullptr))) return TPG_ERR_ARG;
    if (!bn_dtype_ok(dtype_in) || !bn_dtype_ok(dtype_out) || !bn_shape_ok(dtype_in, dtype_out, C))
        return TPG_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return TPG_ERR_UNSUPPORTED;
    hipStream_t st = tpg_stream(stream);
    float *part = static_cast<float *>(ws);
    if (training && phase != TPG_BN_PHASE_APPLY) {
        // statistics use the INPUT type's vector width
        const int ne = dtype_in == TPG_DTYPE_BF16 ? 8 : 4;
        const int rpi = BN_THREADS / (C / ne);
        const int G = stats_blocks(P, rpi);
        if (dtype_in == TPG_DTYPE_BF16) {
            const __hip_bfloat16 *xx = static_cast<const __hip_bfloat16 *>(x);
            hipLaunchKernelGGL(rowbn_stats_kernel<__hip_bfloat16>, dim3(G), dim3(BN_THREADS), 0, st, xx, P, C, part);
            hipLaunchKernelGGL(rowbn_stats_finalize_kernel<__hip_bfloat16>, dim3((C + FIN_CH - 1) / FIN_CH), dim3(BN_THREADS), 0, st, xx,
                               part, G, P, C, eps, momentum, running_mean, running_var, num_batches_tracked, mean, rstd);
        } else {
            const float *xx = static_cast<const float *>(x);
            hipLaunchKernelGGL(rowbn_stats_kernel<float>, dim3(G), dim3(BN_THREADS), 0, st, xx, P, C, part);
            hipLaunchKernelGGL(rowbn_stats_finalize_kernel<float>, dim3((C + FIN_CH - 1) / FIN_CH), dim3(BN_THREADS), 0, st, xx, part, G,
                               P, C, eps, momentum, running_mean, running_var, num_batches_tracked, mean, rstd);
        }
    }  // eval mode: the caller has filled mean / rstd from the running statistics
    if (phase == TPG_BN_PHASE_STATS) {
        TPG_RETURN_IF_LAUNCH_FAILED();
        return TPG_OK;
    }
    const int ne = (dtype_in == TPG_DTYPE_BF16 || dtype_out == TPG_DTYPE_BF16) ? 8 : 4;
    const int rpi_a = BN_THREADS / (C / ne);
    const long long rows_out = K > 0 ? P / K : P;
    const dim3 g(K > 0 ? row_blocks(rows_out, rpi_a, 1, 8192) : row_blocks(P, rpi_a, BN_UNROLL, 4096));
    const dim3 blk(BN_THREADS);
#define TPG_BN_APPLY(TI, TO)                                                                              \
    do {                                                                                                  \
        if (K > 0)                                                                                        \
            hipLaunchKernelGGL((rowbn_apply_max_kernel<TI, TO>), g, blk, 0, st, static_cast<const TI *>(x), \
                               rows_out, K, C, mean, rstd, gamma, beta, slope, static_cast<TO *>(y), argmax); \
        else                                                                                              \
            hipLaunchKernelGGL((rowbn_apply_kernel<TI, TO>), g, blk, 0, st, static_cast<const TI *>(x), P, C, mean, \
                               rstd, gamma, beta, slope, static_cast<TO *>(y));                           \
    } while (0)
    if (dtype_in == TPG_DTYPE_F32 && dtype_out == TPG_DTYPE_F32) TPG_BN_APPLY(float, float);
    else if (dtype_in == TPG_DTYPE_F32) TPG_BN_APPLY(float, __hip_bfloat16);
    else if (dtype_out == TPG_DTYPE_F32) TPG_BN_APPLY(__hip_bfloat16, float);
    else TPG_BN_APPLY(__hip_bfloat16, __hip_bfloat16);
#undef TPG_BN_APPLY
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_rowbn_bwd(const void *gy, int dtype_g, const void *x, int dtype_in, const uint8_t *argmax,
                             const void *y, int dtype_y, long long P, int K, int C, int training,
                             const float *mean, const float *rstd, const float *gamma, const float *beta,
                             float slope, float *dgamma, float *dbeta, void *dx, void *ws, int phase,
                             void *stream) {
    if (P <= 0 || C <= 0 || K < 0 || K > 256 || (K > 0 && P % K)) return TPG_ERR_ARG;
    if (!gy || !x || !dx || !ws || (K > 0 && !argmax)) return TPG_ERR_ARG;
    if ((mean == nullptr) != (rstd == nullptr) || (training && !mean)) return TPG_ERR_ARG;
    // the (gy, y) form of the max-variant sums needs y stored like gy, 16-byte aligned
    const bool from_y = K > 0 && y && dtype_y == dtype_g && !(reinterpret_cast<uintptr_t>(y) & 15);
    if (!bn_dtype_ok(dtype_in) || !bn_dtype_ok(dtype_g) || !bn_shape_ok(dtype_in, dtype_g, C))
        return TPG_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(dx)) & 15)
        return TPG_ERR_UNSUPPORTED;
    hipStream_t st = tpg_stream(stream);
    float *part = static_cast<float *>(ws);
    float *c12 = part + (size_t)BN_MAX_BLOCKS * 2 * C;
    const int ne = (dtype_in == TPG_DTYPE_BF16 || dtype_g == TPG_DTYPE_BF16) ? 8 : 4;
    const int rpi = BN_THREADS / (C / ne);
    const long long rows_g = K > 0 ? P / K : P;
    const int G = stats_blocks(rows_g, rpi);
    const int GA = K > 0 ? row_blocks(rows_g, rpi, 1, 8192) : row_blocks(P, rpi, 4, 4096);
    // no batch statistics and no affine gradients wanted (pure activation [+max]): dx = a * g,
    // nothing to reduce
    const bool need_reduce = (training || dgamma || dbeta) && phase != TPG_BN_PHASE_APPLY;
    const bool do_apply = phase != TPG_BN_PHASE_STATS;
    if (!(training || dgamma || dbeta) && phase != TPG_BN_PHASE_APPLY && hipMemsetAsync(c12, 0, sizeof(float) * 2 * (size_t)C, st) != hipSuccess)
        return TPG_ERR_LAUNCH;
#define TPG_BN_BWD(TI, TG)                                                                                  \
    do {                                                                                                    \
        const TI *xx = static_cast<const TI *>(x);                                                          \
        const TG *gg = static_cast<const TG *>(gy);                                                         \
        if (!need_reduce) {                                                                                 \
        } else if (from_y)                                                                                  \
            hipLaunchKernelGGL((rowbn_bwd_reduce_max_y_kernel<TI, TG>), dim3(G), dim3(BN_THREADS), 0, st, gg, \
                               static_cast<const TG *>(y), xx, argmax, rows_g, K, C, mean, rstd, gamma, beta, \
                               slope, part);                                                                \
        else if (K > 0)                                                                                     \
            hipLaunchKernelGGL((rowbn_bwd_reduce_max_kernel<TI, TG>), dim3(G), dim3(BN_THREADS), 0, st, gg, xx, \
                               argmax, rows_g, K, C, mean, rstd, gamma, beta, slope, part);                 \
        else                                                                                                \
            hipLaunchKernelGGL((rowbn_bwd_reduce_kernel<TI, TG>), dim3(G), dim3(BN_THREADS), 0, st, gg, xx, P, C, \
                               mean, rstd, gamma, beta, slope, part);                                       \
        if (need_reduce)                                                                                    \
            hipLaunchKernelGGL(rowbn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(BN_THREADS), 0, st, part, G, P, C, \
                               training, dgamma, dbeta, c12);                                               \
        if (do_apply)                                                                                       \
            hipLaunchKernelGGL((rowbn_bwd_apply_kernel<TI, TG>), dim3(GA), dim3(BN_THREADS), 0, st, gg, xx, argmax, \
                               P, K, C, mean, rstd, gamma, beta, slope, c12, static_cast<TI *>(dx));      \
    } while (0)
    if (dtype_in == TPG_DTYPE_F32 && dtype_g == TPG_DTYPE_F32) TPG_BN_BWD(float, float);
    else if (dtype_in == TPG_DTYPE_F32) TPG_BN_BWD(float, __hip_bfloat16);
    else if (dtype_g == TPG_DTYPE_F32) TPG_BN_BWD(__hip_bfloat16, float);
    else TPG_BN_BWD(__hip_bfloat16, __hip_bfloat16);
#undef TPG_BN_BWD
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}
