// Fused spectral normalisation of a 1x1-conv / linear weight.
//
// Every conv and linear of the discriminators carries torch.nn.utils.spectral_norm (reference
// discriminator.py:66-68,246-247,351-359 ...).  PyTorch evaluates its forward pre-hook as a
// chain of ~12 tiny kernels (mv, norm, clamp, div, mv, norm, clamp, div, mv, dot, div + copies)
// and as many again in backward; with ~40 hook calls per discriminator pass that was half of
// all kernel launches of a train step and made the step host-bound.  Here one workgroup does
//
//   v <- normalize(W^T u);  u <- normalize(W v);  sigma = u . (W v);  W_sn = W / sigma
//
// in ONE launch (the weight is <= 0.5 MB and stays in L2), and the backward
//   dW = (G - <G, W_sn> u v^T) / sigma        (u, v constants, as in PyTorch)
// in one more.  Same arithmetic as torch.nn.utils.spectral_norm with n_power_iterations = 1,
// eps = 1e-12; sums run in a different order (1e-7 relative).
#include "tpg_common.hpp"

namespace {

constexpr int SN_THREADS = 1024;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// sum over the workgroup, result broadcast to every thread (scratch: 16 floats)
__device__ __forceinline__ float block_sum(float v, float *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = lane < (SN_THREADS / 64) ? scratch[lane] : 0.0f;
    return wave_sum(t);
}

__global__ __launch_bounds__(SN_THREADS) void spectral_norm_fwd_kernel(
    const float *__restrict__ W, float *__restrict__ u, float *__restrict__ v, int R, int Cn, int iterate,
    float eps, float *__restrict__ Wsn, float *__restrict__ sigma_out) {
    extern __shared__ __attribute__((aligned(16))) float sn_smem[];  // [R] u / s, [Cn] v, [16] scratch
    float *su = sn_smem, *sv = sn_smem + R, *scratch = sn_smem + R + Cn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < R; i += SN_THREADS) su[i] = u[i];
    for (int j = tid; j < Cn; j += SN_THREADS) sv[j] = v[j];
    __syncthreads();
    if (iterate) {
        // t = W^T u   (thread per column, rows walked in order: coalesced across the wave)
        float nrm = 0.0f;
        for (int j = tid; j < Cn; j += SN_THREADS) {
            float t = 0.0f;
            for (int i = 0; i < R; ++i) t += W[(size_t)i * Cn + j] * su[i];
            sv[j] = t;
            nrm += t * t;
        }
        nrm = block_sum(nrm, scratch);
        const float inv = 1.0f / fmaxf(sqrtf(nrm), eps);
        for (int j = tid; j < Cn; j += SN_THREADS) sv[j] *= inv;
        __syncthreads();
    }
    // s = W v   (wave per row)
    for (int i = wave; i < R; i += SN_THREADS / 64) {
        float s = 0.0f;
        for (int j = lane; j < Cn; j += 64) s += W[(size_t)i * Cn + j] * sv[j];
        s = wave_sum(s);
        if (lane == 0) su[i] = s;   // su now holds s = W v (the old u was consumed above)
    }
    __syncthreads();
    float sigma;
    if (iterate) {
        float nrm = 0.0f;
        for (int i = tid; i < R; i += SN_THREADS) nrm += su[i] * su[i];
        nrm = block_sum(nrm, scratch);
        const float inv = 1.0f / fmaxf(sqrtf(nrm), eps);
        sigma = nrm * inv;                         // u . s with u = s * inv
        for (int i = tid; i < R; i += SN_THREADS) u[i] = su[i] * inv;
        for (int j = tid; j < Cn; j += SN_THREADS) v[j] = sv[j];
    } else {
        float d = 0.0f;
        for (int i = tid; i < R; i += SN_THREADS) d += u[i] * su[i];
        sigma = block_sum(d, scratch);
    }
    const size_t n = (size_t)R * Cn;
    for (size_t e = tid; e < n; e += SN_THREADS) Wsn[e] = W[e] / sigma;
    if (tid == 0) *sigma_out = sigma;
}

__global__ __launch_bounds__(SN_THREADS) void spectral_norm_bwd_kernel(
    const float *__restrict__ G, const float *__restrict__ Wsn, const float *__restrict__ u,
    const float *__restrict__ v, const float *__restrict__ sigma, int R, int Cn, float *__restrict__ dW) {
    __shared__ float scratch[16];
    const int tid = threadIdx.x;
    const size_t n = (size_t)R * Cn;
    float d = 0.0f;
    for (size_t e = tid; e < n; e += SN_THREADS) d += G[e] * Wsn[e];
    d = block_sum(d, scratch);
    const float sg = *sigma;
    for (size_t e = tid; e < n; e += SN_THREADS) {
        const int i = (int)(e / Cn), j = (int)(e - (size_t)i * Cn);
        dW[e] = (G[e] - d * u[i] * v[j]) / sg;
    }
}

// ---- batched form: one workgroup per weight, `uses` chained power iterations each ----------
// A discriminator forward calls every spectrally-normalised module once per frame (or per
// flow-embedding pair), and each call advances (u, v) by one power iteration and uses its own
// W / sigma.  desc[m] = {W, u, v, R, Cn, uses, out_off}: for use t the kernel writes, at
// out + out_off + t * stride(R, Cn):  W/sigma_t (R*Cn) | u_t (R) | v_t (Cn) | sigma_t (1).
constexpr int SN_CB = 9;        // 64-column blocks a lane covers in the one-pass W^T u: weights of up to 576 columns
struct SnDesc {
    const float *W;
    float *u;
    float *v;
    long long R, Cn, uses, out_off;
};
__host__ __device__ inline long long sn_stride(long long R, long long Cn) { return (R * Cn + R + Cn + 1 + 3) & ~3LL; }

__global__ __launch_bounds__(SN_THREADS) void spectral_norm_multi_fwd_kernel(const SnDesc *__restrict__ desc,
                                                                             float *__restrict__ out, int iterate,
                                                                             float eps) {
    extern __shared__ __attribute__((aligned(16))) float sn_smem[];
    const SnDesc d = desc[blockIdx.x];
    const int R = (int)d.R, Cn = (int)d.Cn;
    const float *__restrict__ W = d.W;
    float *su = sn_smem, *sv = sn_smem + R, *scratch = sn_smem + R + Cn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < R; i += SN_THREADS) su[i] = d.u[i];
    for (int j = tid; j < Cn; j += SN_THREADS) sv[j] = d.v[j];
    __syncthreads();
    const size_t n = (size_t)R * Cn;
    for (int t = 0; t < (int)d.uses; ++t) {
        float *o = out + d.out_off + (size_t)t * sn_stride(R, Cn);
        float *ou = o + n, *ov = ou + R;
        if (iterate) {
            // t = W^T u: wave w takes rows w, w+16, ...; lanes take columns; the 16 waves' partial
            // column sums meet in an LDS slab and are added in WAVE ORDER by the first wave (a float
            // atomic per (wave, column) summed in arrival order: u, v, sigma -- hence every weight of
            // the step -- differed in the last bit from run to run, which the step then amplifies;
            // tests/test_graph_gpu.py compares a replay with its own body bit for bit)
            // ... in ONE pass over the matrix: a lane keeps the partial sums of its columns lane, lane + 64, ... in
            // registers (SN_CB of them), the slab holds all Cn columns of all 16 waves, and the wave-ordered sums run on
            // every thread, not on wave 0 alone (a pass per 64-column block with two barriers each cost 9 x that for
            // the 515-column weights: 30 us per use, 474 us for the 16 chained uses of cfg4's temporal discriminator)
            float *slab = scratch + 16;                     // [16 waves][min(Cn, 64 SN_CB) columns]
            const int SW = Cn < 64 * SN_CB ? Cn : 64 * SN_CB;
            for (int jb = 0; jb < Cn; jb += 64 * SN_CB) {    // (one trip for every weight of the discriminators)
                float acc[SN_CB];
#pragma unroll
                for (int c = 0; c < SN_CB; ++c) acc[c] = 0.0f;
                for (int i = wave; i < R; i += SN_THREADS / 64) {
                    const float ui = su[i];
                    const float *wr = W + (size_t)i * Cn + jb;
#pragma unroll
                    for (int c = 0; c < SN_CB; ++c) {
                        const int j = lane + 64 * c;
                        if (jb + j < Cn) acc[c] += wr[j] * ui;
                    }
                }
#pragma unroll
                for (int c = 0; c < SN_CB; ++c) {
                    const int j = lane + 64 * c;
                    if (jb + j < Cn) slab[wave * SW + j] = acc[c];
                }
                __syncthreads();
                for (int j = tid; j < SW && jb + j < Cn; j += SN_THREADS) {
                    float t = 0.0f;
#pragma unroll
                    for (int w = 0; w < SN_THREADS / 64; ++w) t += slab[w * SW + j];
                    sv[jb + j] = t;
                }
                __syncthreads();
            }
            float nrm = 0.0f;
            for (int j = tid; j < Cn; j += SN_THREADS) nrm += sv[j] * sv[j];
            nrm = block_sum(nrm, scratch);
            const float inv = 1.0f / fmaxf(sqrtf(nrm), eps);
            for (int j = tid; j < Cn; j += SN_THREADS) sv[j] *= inv;
            __syncthreads();
        }
        for (int i = wave; i < R; i += SN_THREADS / 64) {  // s = W v, wave per row
            float sacc = 0.0f;
            for (int j = lane; j < Cn; j += 64) sacc += W[(size_t)i * Cn + j] * sv[j];
            sacc = wave_sum(sacc);
            if (lane == 0) {
                ou[i] = sacc;               // raw s = W v (eval mode reads it for sigma below)
                if (iterate) su[i] = sacc;  // training: u <- s, normalised below
            }
        }
        __syncthreads();
        float sigma;
        if (iterate) {
            float nrm = 0.0f;
            for (int i = tid; i < R; i += SN_THREADS) nrm += su[i] * su[i];
            nrm = block_sum(nrm, scratch);
            const float inv = 1.0f / fmaxf(sqrtf(nrm), eps);
            sigma = nrm * inv;
            for (int i = tid; i < R; i += SN_THREADS) su[i] *= inv;   // u <- s / |s|
            __syncthreads();
        } else {
            float dsum = 0.0f;
            for (int i = tid; i < R; i += SN_THREADS) dsum += su[i] * ou[i];
            sigma = block_sum(dsum, scratch);
        }
        for (int i = tid; i < R; i += SN_THREADS) ou[i] = su[i];
        for (int j = tid; j < Cn; j += SN_THREADS) ov[j] = sv[j];
        for (size_t e = tid; e < n; e += SN_THREADS) o[e] = W[e] / sigma;
        if (tid == 0) ov[Cn] = sigma;
        __syncthreads();
    }
    if (iterate) {
        for (int i = tid; i < R; i += SN_THREADS) d.u[i] = su[i];
        for (int j = tid; j < Cn; j += SN_THREADS) d.v[j] = sv[j];
    }
}

// ---- split form (round 3): a weight's rows over several workgroups ---------------------------
// The one-workgroup form above pulls every use's three passes over a 0.5 MB weight through ONE CU: 165 us per launch
// at cfg2 (280 us for the temporal update's six chained uses, at the head of the chain that ends the step), 513 us at
// cfg4.  Here a workgroup of four waves owns 32 rows of a weight IN REGISTERS (8 rows x <= 576 columns per wave: the
// weight is read from memory once for all uses) and the ceil(R / 32) workgroups of a weight exchange, once per use,
// their partial column sums and squared norms:
//   exchange 0:   c = W^T u0                       (partials over each workgroup's rows)
//   use t:        v_t = normalize(c [* 1/|s_(t-1)|]);  s = W v_t (own rows);  publish W^T s and |s|^2 (own rows)
//   exchange t+1: |s|^2 -> sigma_t = |s|, u_t = s / |s|;  W^T u_t = (W^T s) / |s| -> next use
// (W^T of the UNnormalised s travels with |s|^2, so a use costs ONE exchange, not two.)  The exchange follows the
// guide's small-payload recipe: every float is an 8-byte {epoch, value} granule stored and polled with relaxed
// agent-scope atomics -- the data is its own flag, no fences, nothing depends on placement or dispatch order; epochs
// count exchanges within the launch, the slots are double-buffered by exchange parity (a workgroup publishes exchange
// x + 2 only after consuming every partial of x + 1, which exists only once every workgroup has consumed x) and are
// zeroed by a memset node in front of every launch.  Sums over workgroups and waves run in index order: bitwise
// reproducible, identical in every workgroup of a weight.  Spins are bounded (a timeout word is set and the launch
// drains).  Training mode only (the power iteration); eval keeps the one-workgroup kernel.
constexpr int SNS_THREADS = 256;
constexpr int SNS_WAVES = SNS_THREADS / 64;
constexpr int SNS_RW = 8;                       // rows per wave
constexpr int SNS_ROWS = SNS_WAVES * SNS_RW;    // rows per workgroup
constexpr int SNS_MAXCN = 64 * SN_CB;
constexpr int SNS_MAXPARTS = 16;                // rows <= 512
constexpr int SNS_XB = 8;                       // granule loads a thread keeps in flight
struct SnSplitDesc {
    const float *W;
    float *u;
    float *v;
    long long R, Cn, uses, out_off, parts, x_off;      // x_off: granules into the exchange buffer
};

__device__ __forceinline__ float sns_block_sum(float v, float *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < SNS_WAVES; ++w) t += scratch[w];
    return t;
}

__global__ __launch_bounds__(SNS_THREADS) void spectral_norm_split_kernel(const SnSplitDesc *__restrict__ desc,
                                                                          const int2 *__restrict__ part_map,
                                                                          float *__restrict__ out,
                                                                          unsigned long long *xws,
                                                                          unsigned *tmo, float eps) {
    __shared__ float sv[SNS_MAXCN], sc[SNS_MAXCN + 1], slab[SNS_WAVES * SNS_MAXCN], su[SNS_ROWS], ss[SNS_ROWS],
        scratch[16], part[SNS_MAXPARTS * (SNS_MAXCN + 1)];
    const int2 pm = part_map[blockIdx.x];
    const SnSplitDesc d = desc[pm.x];
    const int g = pm.y, G = (int)d.parts, R = (int)d.R, Cn = (int)d.Cn, uses = (int)d.uses;
    if (G > SNS_MAXPARTS || Cn > SNS_MAXCN) {              // (uniform over a weight's workgroups: nobody waits)
        if (threadIdx.x == 0) atomicOr(tmo, 2u);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = g * SNS_ROWS;
    const int nrows = R - r0 < SNS_ROWS ? R - r0 : SNS_ROWS;
    const size_t n = (size_t)R * Cn;
    const int XS = Cn + 1;                              // granules per workgroup and exchange
    unsigned long long *X = xws + d.x_off;

    float w[SNS_RW][SN_CB];
#pragma unroll
    for (int r = 0; r < SNS_RW; ++r) {
        const int lr = wave * SNS_RW + r;
        const float *wr = d.W + (size_t)(r0 + (lr < nrows ? lr : 0)) * Cn;
#pragma unroll
        for (int c = 0; c < SN_CB; ++c) {
            const int j = lane + 64 * c;
            w[r][c] = (lr < nrows && j < Cn) ? wr[j] : 0.0f;
        }
    }
    if (tid < SNS_ROWS) su[tid] = tid < nrows ? d.u[r0 + tid] : 0.0f;     // (v0 is not an input of the iteration)
    __syncthreads();

    // partial column sums over this workgroup's rows with coefficients coef[row] (LDS) + a scalar -> exchange x
    auto publish = [&](int x, const float *coef, float scalar, bool columns) {
        unsigned long long *slot = X + ((size_t)(x & 1) * G + g) * XS;
        const unsigned long long tag = (unsigned long long)(x + 1) << 32;
        if (columns) {
            float acc[SN_CB];
#pragma unroll
            for (int c = 0; c < SN_CB; ++c) acc[c] = 0.0f;
#pragma unroll
            for (int r = 0; r < SNS_RW; ++r) {
                const float cf = coef[wave * SNS_RW + r];
#pragma unroll
                for (int c = 0; c < SN_CB; ++c) acc[c] += w[r][c] * cf;
            }
#pragma unroll
            for (int c = 0; c < SN_CB; ++c) {
                const int j = lane + 64 * c;
                if (j < Cn) slab[wave * SNS_MAXCN + j] = acc[c];
            }
            __syncthreads();
            for (int j = tid; j < Cn; j += SNS_THREADS) {
                float t = 0.0f;
#pragma unroll
                for (int ww = 0; ww < SNS_WAVES; ++ww) t += slab[ww * SNS_MAXCN + j];
                __hip_atomic_store(slot + j, tag | (unsigned long long)__float_as_uint(t), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid == 0)
            __hip_atomic_store(slot + Cn, tag | (unsigned long long)__float_as_uint(scalar), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    };
    // sc[0..Cn) (columns) and sc[Cn] (scalar) = the partials of exchange x summed over the workgroups in index order.
    // The G * (Cn + 1) granules of an exchange are one contiguous array: the threads sweep it with SNS_XB independent
    // loads in flight each (a dependent poll per granule cost ~28 us per exchange: 16-24 remote round trips in a row),
    // re-poll what has not arrived, park the values in LDS and add them up per column afterwards.
    auto poll = [&](const unsigned long long *p, unsigned long long q, unsigned want) {
        for (unsigned spins = 0; (unsigned)(q >> 32) != want; ++spins) {
            if (spins > (1u << 22)) {                               // bounded: flag it and drain
                atomicOr(tmo, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            q = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return __uint_as_float((unsigned)q);
    };
    auto consume = [&](int x, bool columns) {
        const unsigned long long *base = X + (size_t)(x & 1) * G * XS;
        const unsigned want = (unsigned)(x + 1);
        if (columns) {
            const int total = G * XS;
            for (int e0 = 0; e0 < total; e0 += SNS_THREADS * SNS_XB) {
                unsigned long long q[SNS_XB];
#pragma unroll
                for (int k = 0; k < SNS_XB; ++k) q[k] = 0ull;                  // tag 0: not arrived
                // every pass re-issues the loads of ALL granules still missing, together (one round trip per pass)
                for (unsigned spins = 0;; ++spins) {
                    bool missing = false;
#pragma unroll
                    for (int k = 0; k < SNS_XB; ++k) {
                        const int e = e0 + k * SNS_THREADS + tid;
                        if (e < total && (unsigned)(q[k] >> 32) != want)
                            q[k] = __hip_atomic_load(base + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int k = 0; k < SNS_XB; ++k) {
                        const int e = e0 + k * SNS_THREADS + tid;
                        missing |= e < total && (unsigned)(q[k] >> 32) != want;
                    }
                    if (!missing) break;
                    if (spins > (1u << 20)) {                                  // bounded: flag it and drain
                        atomicOr(tmo, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int k = 0; k < SNS_XB; ++k) {
                    const int e = e0 + k * SNS_THREADS + tid;
                    if (e < total) part[e] = __uint_as_float((unsigned)q[k]);
                }
            }
            __syncthreads();
            for (int j = tid; j <= Cn; j += SNS_THREADS) {
                float t = 0.0f;
                for (int gg = 0; gg < G; ++gg) t += part[gg * XS + j];
                sc[j] = t;
            }
        } else {
            if (tid < G) {
                const unsigned long long *p = base + (size_t)tid * XS + Cn;
                part[tid] = poll(p, __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), want);
            }
            __syncthreads();
            if (tid == 0) {
                float t = 0.0f;
                for (int gg = 0; gg < G; ++gg) t += part[gg];
                sc[Cn] = t;
            }
        }
        __syncthreads();
    };

    // W / sigma of a finished use, from the registers.  Called AFTER the next exchange's partials are published: the 64 KB
    // of stores then drain while the workgroup would wait for the other workgroups' granules anyway (in front of the
    // publish they cost 6 of a use's 19 us: tools/tune_sn.py with -DTPG_SNS_NO_OUT).
    auto write_scaled = [&](int use, float sg) {
        float *o = out + d.out_off + (size_t)use * sn_stride(R, Cn);
#pragma unroll
        for (int r = 0; r < SNS_RW; ++r) {
            const int lr = wave * SNS_RW + r;
            if (lr < nrows) {
                float *orow = o + (size_t)(r0 + lr) * Cn;
#pragma unroll
                for (int c = 0; c < SN_CB; ++c) {
                    const int j = lane + 64 * c;
                    if (j < Cn) orow[j] = w[r][c] / sg;
                }
            }
        }
    };
    publish(0, su, 0.0f, true);
    float sigma = 0.0f;
    for (int t = 0; t <= uses; ++t) {
        consume(t, t < uses);
        float inv_s = 1.0f;
        if (t > 0) {
            // close use t - 1: sigma, u, and its outputs
            const float nrm = sc[Cn];
            inv_s = 1.0f / fmaxf(sqrtf(nrm), eps);
            sigma = nrm * inv_s;
            float *o = out + d.out_off + (size_t)(t - 1) * sn_stride(R, Cn);
            float *ou = o + n, *ov = ou + R;
            if (tid < SNS_ROWS) su[tid] = ss[tid] * inv_s;
            __syncthreads();
            if (tid < nrows) ou[r0 + tid] = su[tid];
            if (g == 0) {
                for (int j = tid; j < Cn; j += SNS_THREADS) ov[j] = sv[j];
                if (tid == 0) ov[Cn] = sigma;
            }
        }
        if (t == uses) {
            write_scaled(t - 1, sigma);
            break;
        }
        // v_t = normalize(W^T u_(t-1)),  W^T u_(t-1) = (sum of the partials) [* 1/|s|]
        float nv = 0.0f;
        for (int j = tid; j < Cn; j += SNS_THREADS) {
            const float x = t > 0 ? sc[j] * inv_s : sc[j];
            sc[j] = x;
            nv += x * x;
        }
        nv = sns_block_sum(nv, scratch);
        const float inv_v = 1.0f / fmaxf(sqrtf(nv), eps);
        for (int j = tid; j < Cn; j += SNS_THREADS) sv[j] = sc[j] * inv_v;
        __syncthreads();
        // s = W v_t on this workgroup's rows
        float vloc[SN_CB];
#pragma unroll
        for (int c = 0; c < SN_CB; ++c) {
            const int j = lane + 64 * c;
            vloc[c] = j < Cn ? sv[j] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < SNS_RW; ++r) {
            float a = 0.0f;
#pragma unroll
            for (int c = 0; c < SN_CB; ++c) a += w[r][c] * vloc[c];
            a = wave_sum(a);
            if (lane == 0) ss[wave * SNS_RW + r] = a;
        }
        __syncthreads();
        float ns = 0.0f;
        if (tid == 0) {
            for (int i = 0; i < nrows; ++i) ns += ss[i] * ss[i];
        }
        publish(t + 1, ss, ns, t + 1 < uses);
        if (t > 0) write_scaled(t - 1, sigma);
    }
    // the buffers after the last use
    if (tid < nrows) d.u[r0 + tid] = su[tid];
    if (g == 0)
        for (int j = tid; j < Cn; j += SNS_THREADS) d.v[j] = sv[j];
}

// backward: gradients of all uses are packed in one flat buffer (use t of weight m at
// g + g_off + t*R*Cn), dW[m] (at dw + dw_off) = sum over uses; descriptors hold only sizes
// and offsets, so they are built once and stay valid for every later call (and for replay
// from a captured graph).
struct SnBwdDesc {
    long long R, Cn, uses, out_off, g_off, dw_off;
};
// Two launches, each spread over SNB_CH row chunks per weight (one workgroup per weight, as the
// forward has to be, pulled the whole 6 x 3 passes over a 0.5 MB weight through ONE CU: 313 us
// at the end of the discriminator update's backward chain):
//   dots : partial <G_t, Wsn_t> of every (weight, use, row chunk)  -> dots[(m*maxu + t)*SNB_CH + c]
//   apply: dW[e] = sum_t (G_t[e] - dot_t u_t[i] v_t[j]) / sigma_t, dot_t = the SNB_CH partials summed in
//          fixed order; each element is accumulated in a register over the uses and written once.
constexpr int SNB_CH = 8;

__global__ __launch_bounds__(SN_THREADS) void spectral_norm_multi_bwd_dots_kernel(
    const SnBwdDesc *__restrict__ desc, const float *__restrict__ g, const float *__restrict__ out, int maxu,
    float *__restrict__ dots) {
    __shared__ float scratch[16];
    const int m = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const SnBwdDesc d = desc[m];
    const int R = (int)d.R, Cn = (int)d.Cn;
    const size_t n = (size_t)R * Cn;
    const int r0 = (int)((long long)R * c / SNB_CH), r1 = (int)((long long)R * (c + 1) / SNB_CH);
    const size_t e0 = (size_t)r0 * Cn, e1 = (size_t)r1 * Cn;
    for (int t = 0; t < (int)d.uses; ++t) {
        const float *G = g + d.g_off + (size_t)t * n;
        const float *o = out + d.out_off + (size_t)t * sn_stride(R, Cn);
        float dot = 0.0f;
        for (size_t e = e0 + tid; e < e1; e += SN_THREADS) dot += G[e] * o[e];
        dot = block_sum(dot, scratch);
        if (tid == 0) dots[((size_t)m * maxu + t) * SNB_CH + c] = dot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SN_THREADS) void spectral_norm_multi_bwd_apply_kernel(
    const SnBwdDesc *__restrict__ desc, const float *__restrict__ g, const float *__restrict__ out, int maxu,
    const float *__restrict__ dots, float *__restrict__ dw) {
    __shared__ float sdot[64], ssig[64];
    const int m = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const SnBwdDesc d = desc[m];
    const int R = (int)d.R, Cn = (int)d.Cn, U = (int)d.uses;
    const size_t n = (size_t)R * Cn;
    for (int t = tid; t < U; t += SN_THREADS) {
        const float *pd = dots + ((size_t)m * maxu + t) * SNB_CH;
        float s = 0.0f;
        for (int cc = 0; cc < SNB_CH; ++cc) s += pd[cc];
        sdot[t] = s;
        ssig[t] = out[d.out_off + (size_t)t * sn_stride(R, Cn) + n + R + Cn];
    }
    __syncthreads();
    const int r0 = (int)((long long)R * c / SNB_CH), r1 = (int)((long long)R * (c + 1) / SNB_CH);
    float *dW = dw + d.dw_off;
    for (size_t e = (size_t)r0 * Cn + tid; e < (size_t)r1 * Cn; e += SN_THREADS) {
        const int i = (int)(e / Cn), j = (int)(e - (size_t)i * Cn);
        float acc = 0.0f;
        for (int t = 0; t < U; ++t) {
            const float *o = out + d.out_off + (size_t)t * sn_stride(R, Cn);
            acc += (g[d.g_off + (size_t)t * n + e] - sdot[t] * o[n + i] * o[n + R + j]) / ssig[t];
        }
        dW[e] = acc;
    }
}

}  // namespace

extern "C" long long tpg_spectral_norm_multi_stride(int R, int Cn) { return sn_stride(R, Cn); }

extern "C" int tpg_spectral_norm_multi_fwd(const void *desc, int M, int max_rc, float *out, int iterate, float eps,
                                           void *stream) {
    if (M < 0 || !desc || !out) return TPG_ERR_ARG;
    if (M == 0) return TPG_OK;
    // u | v | scratch | column slab (16 waves x Cn <= max_rc columns)
    const size_t smem = sizeof(float) * ((size_t)max_rc + 64 + 16 * (size_t)(max_rc < 64 * SN_CB ? max_rc : 64 * SN_CB));
    if (smem > 64 * 1024) return TPG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(spectral_norm_multi_fwd_kernel, dim3(M), dim3(SN_THREADS), smem, tpg_stream(stream),
                       static_cast<const SnDesc *>(desc), out, iterate, eps);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_spectral_norm_multi_fwd_split(const void *desc, const void *part_map, int total_parts, int max_cn,
                                                 float *out, void *xws, long long xws_words, float eps, void *stream) {
    if (total_parts < 0 || xws_words < 2 || !desc || !part_map || !out || !xws) return TPG_ERR_ARG;
    if (total_parts == 0) return TPG_OK;
    if (max_cn > SNS_MAXCN) return TPG_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(xws) & 15) return TPG_ERR_UNSUPPORTED;
    hipStream_t st = tpg_stream(stream);
    // every polled word back to epoch 0 (a memset node in a captured step); xws_words 8-byte words, the last one holds
    // the timeout flag
    if (hipMemsetAsync(xws, 0, sizeof(unsigned long long) * (size_t)xws_words, st) != hipSuccess) return TPG_ERR_LAUNCH;
    unsigned long long *x = static_cast<unsigned long long *>(xws);
    hipLaunchKernelGGL(spectral_norm_split_kernel, dim3(total_parts), dim3(SNS_THREADS), 0, st,
                       static_cast<const SnSplitDesc *>(desc), static_cast<const int2 *>(part_map), out, x,
                       reinterpret_cast<unsigned *>(x + xws_words - 1), eps);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_spectral_norm_split_rows(void) { return SNS_ROWS; }
extern "C" int tpg_spectral_norm_split_max_cn(void) { return SNS_MAXCN; }
extern "C" int tpg_spectral_norm_split_max_rows(void) { return SNS_ROWS * SNS_MAXPARTS; }

extern "C" long long tpg_spectral_norm_multi_bwd_scratch(int M, int max_uses) {
    return (long long)M * max_uses * SNB_CH;          // floats
}

extern "C" int tpg_spectral_norm_multi_bwd(const void *desc, int M, int max_uses, const float *g, const float *out,
                                           float *dw, float *scratch, void *stream) {
    if (M < 0 || max_uses < 1 || max_uses > 64 || !desc || !g || !out || !dw || !scratch) return TPG_ERR_ARG;
    if (M == 0) return TPG_OK;
    const SnBwdDesc *dd = static_cast<const SnBwdDesc *>(desc);
    hipLaunchKernelGGL(spectral_norm_multi_bwd_dots_kernel, dim3(SNB_CH, M), dim3(SN_THREADS), 0, tpg_stream(stream),
                       dd, g, out, max_uses, scratch);
    hipLaunchKernelGGL(spectral_norm_multi_bwd_apply_kernel, dim3(SNB_CH, M), dim3(SN_THREADS), 0,
                       tpg_stream(stream), dd, g, out, max_uses, scratch, dw);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_spectral_norm_fwd(const float *W, float *u, float *v, int R, int Cn, int iterate, float eps,
                                     float *Wsn, float *sigma, void *stream) {
    if (R <= 0 || Cn <= 0 || !W || !u || !v || !Wsn || !sigma) return TPG_ERR_ARG;
    const size_t smem = sizeof(float) * ((size_t)R + Cn + 32);
    if (smem > 48 * 1024) return TPG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(spectral_norm_fwd_kernel, dim3(1), dim3(SN_THREADS), smem, tpg_stream(stream), W, u, v, R,
                       Cn, iterate, eps, Wsn, sigma);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_spectral_norm_bwd(const float *G, const float *Wsn, const float *u, const float *v,
                                     const float *sigma, int R, int Cn, float *dW, void *stream) {
    if (R <= 0 || Cn <= 0 || !G || !Wsn || !u || !v || !sigma || !dW) return TPG_ERR_ARG;
    hipLaunchKernelGGL(spectral_norm_bwd_kernel, dim3(1), dim3(SN_THREADS), 0, tpg_stream(stream), G, Wsn, u, v,
                       sigma, R, Cn, dW);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}
