// Fused radius search + bicubic-kernel weighted average: the reference's `cubic_interpolation`
// (gcn_lib/interpolation.py:107-123 over get_local_neighbor_graph :16-75), which the `--use_vel`
// step calls once per frame AND per sample from a Python loop (train_step_final.py:51-66) and
// which builds a DGL graph per call (FRNN K=32 -> unique -> FRNN again -> optional kNN-4
// padding edges -> two scatter-sums).  Per query q that graph reduces to
//
//   hits  = the <= 32 nearest field points with d^2 < cutoff^2, ascending (d^2, idx)
//   w_j   = bicubic(|p_j - q| / cutoff) * 8 / (pi cutoff^3)
//   out   = sum_j m_j w_j f_j / (sum_j m_j w_j + 1e-6)
//
// with multiplicity m_j = 1, or -- the reference's kNN padding, active for a whole cloud as
// soon as ONE of its queries has no hit, then applied to every query with fewer than 32 hits --
// m_j = 2 for the query's 4 nearest hits (the padding adds edges to the 4 nearest candidates
// without removing the existing ones; candidates beyond the cutoff get weight 0).  Whether the
// padding is active is a per-cloud fact, so one launch writes BOTH averages plus the hit count
// and the caller selects (ops.cubic_interpolation).
//
// One wave per query, same search as knn.hip (D = 3, K = 32, radius); the 32 best keys end up
// one per lane, so weights, multiplicities and the F+1 sums are a handful of wave reductions.
#include "knn_select.hpp"
#include "tpg_common.hpp"

namespace {

constexpr int CI_WAVES = 4;
constexpr int CI_K = 32;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// gcn_lib/interpolation.py:94-104, evaluated in fp32 in the reference's operation order
__device__ __forceinline__ float bicubic_w(float r, float cutoff, float coeff) {
    const float q = r / cutoff;
    float ker = 0.0f;
    if (q >= 0.0f && q <= 0.5f) ker = 6.0f * (q * q * q - q * q) + 1.0f;
    else if (q > 0.5f && q <= 1.0f) { const float t = 1.0f - q; ker = 2.0f * (t * t * t); }
    return ker * coeff;
}

__global__ __launch_bounds__(CI_WAVES * 64) void cubic_interp_kernel(
    const float *__restrict__ query, const float *__restrict__ pos, const float *__restrict__ field, int Nq,
    int Np, int F, float cutoff, float r2, float coeff, float *__restrict__ out_plain,
    float *__restrict__ out_pad, int32_t *__restrict__ hits) {
    __shared__ tpg_u64 ci_slots[CI_WAVES * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    tpg_u64 *slot = ci_slots + wave * 64;
    const int b = blockIdx.y;
    const int i = blockIdx.x * CI_WAVES + wave;
    if (i >= Nq) return;
    const size_t q = (size_t)b * Nq + i;
    const float qx = query[q * 3], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
    const float *cb = pos + (size_t)b * Np * 3;
    const tpg_u64 INF = ~0ull;
    tpg_u64 best = INF, thr = INF;
    for (int base = 0; base < Np; base += 64) {
        const int j = base + lane;
        tpg_u64 key = INF;
        if (j < Np) {
            const float d = tpg_sq3(qx, qy, qz, cb[(size_t)j * 3], cb[(size_t)j * 3 + 1], cb[(size_t)j * 3 + 2]);
            if (d < r2) key = ((tpg_u64)__float_as_uint(d) << 32) | (unsigned)j;
        }
        tpg_knn_merge(best, thr, key, CI_K, lane, slot);
    }
    const bool has = lane < CI_K && best != INF;
    const int nh = __popcll(__ballot(has));
    float w = 0.0f;
    int j = 0;
    if (has) {
        float d2 = __uint_as_float((unsigned)(best >> 32));
        j = (int)(unsigned)best;
        if (d2 < 1e-8f) d2 = 0.0f;                      // l2dist, interpolation.py:13
        w = bicubic_w(sqrtf(d2), cutoff, coeff);
    }
    const float wp = lane < 4 ? 2.0f * w : w;            // padding edges duplicate the 4 nearest hits
    const float den = wave_sum(w), denp = wave_sum(wp);
    const float *fb = field + (size_t)b * Np * F;
    for (int f = 0; f < F; ++f) {
        const float v = has ? fb[(size_t)j * F + f] : 0.0f;
        const float num = wave_sum(w * v), nump = wave_sum(wp * v);
        if (lane == 0) {
            out_plain[q * F + f] = num / (den + 1e-6f);
            out_pad[q * F + f] = nump / (denp + 1e-6f);
        }
    }
    if (lane == 0) hits[q] = nh;
}

}  // namespace

extern "C" int tpg_cubic_interp_f32(const float *query, const float *pos, const float *field, int B, int Nq,
                                    int Np, int F, float cutoff, float *out_plain, float *out_pad,
                                    int32_t *hits, void *stream) {
    if (B < 0 || Nq < 0 || Np < 0 || F <= 0 || !(cutoff > 0.0f)) return TPG_ERR_ARG;
    if ((long long)B * Nq == 0) return TPG_OK;
    if (B > 65535) return TPG_ERR_ARG;
    if (!query || !out_plain || !out_pad || !hits || (Np > 0 && (!pos || !field))) return TPG_ERR_ARG;
    const float r2 = cutoff * cutoff;                                   // fp32(r)*fp32(r), as the radius search
    const float coeff = (float)(8.0 / (3.14159265358979323846 * (double)cutoff * (double)cutoff * (double)cutoff));
    const dim3 grid((unsigned)((Nq + CI_WAVES - 1) / CI_WAVES), (unsigned)B), block(CI_WAVES * 64);
    hipLaunchKernelGGL(cubic_interp_kernel, grid, block, 0, tpg_stream(stream), query, pos, field, Nq, Np, F, cutoff,
                       r2, coeff, out_plain, out_pad, hits);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}
