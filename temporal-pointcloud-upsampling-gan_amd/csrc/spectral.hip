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

}  // namespace

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
