// The EdgeConv MLP tail at the SMALL channel counts of the generator's Inception-DenseGCN blocks
// (gcn_lib/pointnet/gcn.py:207-211 inside IDGCNLayer, gcn.py:229-231: in_feats/4 = 32 -> half = 16):
//
//     out[n] = max_j  lrelu_s2( W2 . lrelu_s1( W1 . h[n, j] ) ) ,   h (P*K, 16) edge features, W1 16x16, W2 32x16
//
// in ONE forward and ONE backward launch.  The op is a stream over h -- 32 bytes per edge (bf16), 3.3 M edges per
// call at cfg5 -- and the library path around it (GEMM, LeakyReLU, GEMM, activation + max; backward: two
// data-gradient GEMMs, two split-K weight-gradient GEMMs with their partial sums, two LeakyReLU backwards, casts)
// moved every intermediate through HBM: ~10x the bytes and ~16 launches.
//
// Arithmetic: fp32 throughout, on v_mfma_f32_16x16x4_f32 (an exact k-ordered fmaf chain).  A first form with the
// 768 weights as scalar operands of v_fmac (s_load_dwordx16 + wait before every 16 FMAs, 48 times per edge) was
// SLOWER than the library path it replaced; as matrix operands the weights are 12 (forward) / 16 (backward)
// registers per lane, loaded once.
//
// forward  a wave owns 16 points and walks their K edges: per edge slot 4 + 8 matrix instructions, a running
//          first-maximum per (point, channel) in registers, one row of outputs + arg-max bytes per point at the
//          end.  Nothing else is stored: the backward recomputes the hidden layer.
// backward the same walk: z1 again (4), the routed output gradient through W2^T (8) and W1^T (4) to dh, and
//          the two weight gradients as edge-summed outer products (12, operands transposed through 5 KB of the
//          wave's LDS), accumulated in registers across the launch (persistent waves, grid-stride tiles); one
//          slab of 768 floats per wave at the end, summed in wave order by the reduce kernel (no atomics).
#include <hip/hip_bf16.h>

#include "tpg_common.hpp"

namespace {

constexpr int SM_H = 16, SM_C1 = 16, SM_C2 = 32;
constexpr int SM_NW = SM_C1 * SM_H + SM_C2 * SM_C1;   // 768 weights

// Lane roles of the 16x16x4 f32 matrix instruction, D = A . B (cdna_hip_programming.md section 3):
//   A[i = lane & 15][k = lane >> 4],  B[k = lane >> 4][j = lane & 15],  D[row = 4 (lane >> 4) + reg][col = lane & 15].
// Here j / col is always a POINT of the wave's tile of 16 points (at one neighbour slot), so a lane
// (e = lane & 15, q = lane >> 4) loads channels 4q .. 4q+3 of its point's edge row, supplies them as the B operand of
// four k-steps (k-step s, k = q  <->  channel 4q + s: any one-to-one map of k works when A uses the same one), and gets
// back output channels 4q .. 4q+3 of the same point -- the layout the NEXT product wants as its B operand.  The weights
// sit in registers as A operands (lane: row e of W, columns 4q + s).  No transposes, no LDS in the forward.
typedef float sm_f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ void sm_load4(const T *__restrict__ p, float (&v)[4]) {
    if constexpr (sizeof(T) == 2) {
        const uint2 a = *reinterpret_cast<const uint2 *>(p);
        v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
        v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    } else {
        const float4 a = *reinterpret_cast<const float4 *>(p);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    }
}
template <typename T>
__device__ __forceinline__ void sm_store4(T *__restrict__ p, const float (&v)[4]) {
    if constexpr (sizeof(T) == 2) {
        unsigned short b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const __hip_bfloat16 t = __float2bfloat16(v[i]); b[i] = *reinterpret_cast<const unsigned short *>(&t); }
        *reinterpret_cast<uint2 *>(p) = make_uint2((unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16));
    } else {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

#define SM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------- forward
constexpr int SF_WAVES = 4;
template <typename T>
__global__ __launch_bounds__(SF_WAVES * 64) void small_tail_fwd_kernel(
    const T *__restrict__ h, const float *__restrict__ W1, const float *__restrict__ W2, float s1, float s2,
    long long P, int K, T *__restrict__ out, uint8_t *__restrict__ arg) {
    const int lane = threadIdx.x & 63, e = lane & 15, q = lane >> 4;
    const long long tile = (long long)blockIdx.x * SF_WAVES + (threadIdx.x >> 6);
    const long long n = tile * 16 + e;
    if (tile * 16 >= P) return;                                   // whole wave
    const bool live = n < P;
    float w1a[4], w2a[2][4];
    sm_load4(W1 + e * SM_H + 4 * q, w1a);
    sm_load4(W2 + e * SM_C1 + 4 * q, w2a[0]);
    sm_load4(W2 + (16 + e) * SM_C1 + 4 * q, w2a[1]);
    float best[2][4];
    int barg[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { best[t][r] = -INFINITY; barg[t][r] = 0; }
    const T *hp = h + ((size_t)(live ? n : P - 1) * K) * SM_H + 4 * q;
    float hq[4];
    sm_load4(hp, hq);
    for (int j = 0; j < K; ++j) {
        float hn[4];
        sm_load4(hp + (size_t)(j + 1 < K ? j + 1 : j) * SM_H, hn);      // the next edge travels during this one
        sm_f32x4 z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) z1 = SM_MFMA(w1a[s], hq[s], z1);
        float a1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a1[r] = z1[r] > 0.0f ? z1[r] : z1[r] * s1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            sm_f32x4 z2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) z2 = SM_MFMA(w2a[t][s], a1[s], z2);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (z2[r] > best[t][r]) { best[t][r] = z2[r]; barg[t][r] = j; }        // j ascends: the first maximum
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) hq[r] = hn[r];
    }
    if (live) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = best[t][r] > 0.0f ? best[t][r] : best[t][r] * s2;
            sm_store4(out + (size_t)n * SM_C2 + 16 * t + 4 * q, o);
            *reinterpret_cast<unsigned *>(arg + (size_t)n * SM_C2 + 16 * t + 4 * q) =
                (unsigned)barg[t][0] | ((unsigned)barg[t][1] << 8) | ((unsigned)barg[t][2] << 16) | ((unsigned)barg[t][3] << 24);
        }
    }
}

// ------------------------------------------------------------------------------------------ backward
// The weight gradients sum over EDGES, i.e. edges must sit on the k axis (lane >> 4) of a product whose other two
// axes are channels: the four per-edge vectors of a tile (gz2, a1, gz1, h; 16 edges x 16 channels each, gz2 twice)
// are written to the wave's 5 KB of LDS in the layout the products above left them in and read back transposed.
constexpr int SB_WAVES = 4;
constexpr int SB_WAVE_FLOATS = 5 * 256;

template <typename T>
__global__ __launch_bounds__(SB_WAVES * 64) void small_tail_bwd_kernel(
    const T *__restrict__ h, const T *__restrict__ out, const T *__restrict__ gout, const uint8_t *__restrict__ arg,
    const float *__restrict__ W1, const float *__restrict__ W2, float s1, float s2, long long P, int K,
    T *__restrict__ gh, float *__restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float sb_smem[SB_WAVES * SB_WAVE_FLOATS];
    const int lane = threadIdx.x & 63, e = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *xs = sb_smem + wave * SB_WAVE_FLOATS;                  // [5][16 edges][16 channels]: gz2 lo, gz2 hi, a1, gz1, h
    const long long tiles = (P + 15) / 16;
    const long long gw = (long long)blockIdx.x * SB_WAVES + wave, nw = (long long)gridDim.x * SB_WAVES;
    float w1a[4], w1t[4], w2t[2][4];
    sm_load4(W1 + e * SM_H + 4 * q, w1a);                         // A of z1 = W1 h:      W1[e][4q + s]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        w1t[s] = W1[(4 * q + s) * SM_H + e];                      // A of gh = W1^T gz1:  W1[4q + s][e]
        w2t[0][s] = W2[(4 * q + s) * SM_C1 + e];                  // A of ga1 = W2^T gz2: W2[16 t + 4q + s][e]
        w2t[1][s] = W2[(16 + 4 * q + s) * SM_C1 + e];
    }
    sm_f32x4 dw1 = {0.f, 0.f, 0.f, 0.f}, dw2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (long long tile = gw; tile < tiles; tile += nw) {
        const long long n = tile * 16 + e;
        const bool live = n < P;
        const long long nn = live ? n : P - 1;
        // per point: the routed output gradient of channels 16 t + 4q + r and their arg-max slots
        float gp[2][4];
        int ab[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float go[4], oo[4];
            sm_load4(gout + (size_t)nn * SM_C2 + 16 * t + 4 * q, go);
            sm_load4(out + (size_t)nn * SM_C2 + 16 * t + 4 * q, oo);
            const unsigned b = *reinterpret_cast<const unsigned *>(arg + (size_t)nn * SM_C2 + 16 * t + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gp[t][r] = live ? (oo[r] > 0.0f ? go[r] : go[r] * s2) : 0.0f;
                ab[t][r] = (int)((b >> (8 * r)) & 0xffu);
            }
        }
        const T *hp = h + ((size_t)nn * K) * SM_H + 4 * q;
        T *gp_out = gh + ((size_t)nn * K) * SM_H + 4 * q;
        float hq[4];
        sm_load4(hp, hq);
        for (int j = 0; j < K; ++j) {
            float hn[4];
            sm_load4(hp + (size_t)(j + 1 < K ? j + 1 : j) * SM_H, hn);
            sm_f32x4 z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) z1 = SM_MFMA(w1a[s], hq[s], z1);
            float a1[4], dz1[4], gz2[2][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a1[r] = z1[r] > 0.0f ? z1[r] : z1[r] * s1;
                dz1[r] = z1[r] > 0.0f ? 1.0f : s1;
            }
            sm_f32x4 ga1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gz2[t][r] = ab[t][r] == j ? gp[t][r] : 0.0f;
                    ga1 = SM_MFMA(w2t[t][r], gz2[t][r], ga1);
                }
            float gz1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) gz1[r] = ga1[r] * dz1[r];
            sm_f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) g = SM_MFMA(w1t[s], gz1[s], g);
            if (live) {
                const float gv[4] = {g[0], g[1], g[2], g[3]};
                sm_store4(gp_out + (size_t)j * SM_H, gv);
            }
            // transposed read-back for the two outer-product sums (LDS operations of one wave complete in order)
            const float hz[4] = {live ? hq[0] : 0.0f, live ? hq[1] : 0.0f, live ? hq[2] : 0.0f, live ? hq[3] : 0.0f};
            *reinterpret_cast<float4 *>(xs + 0 * 256 + e * 16 + 4 * q) = make_float4(gz2[0][0], gz2[0][1], gz2[0][2], gz2[0][3]);
            *reinterpret_cast<float4 *>(xs + 1 * 256 + e * 16 + 4 * q) = make_float4(gz2[1][0], gz2[1][1], gz2[1][2], gz2[1][3]);
            *reinterpret_cast<float4 *>(xs + 2 * 256 + e * 16 + 4 * q) = make_float4(a1[0], a1[1], a1[2], a1[3]);
            *reinterpret_cast<float4 *>(xs + 3 * 256 + e * 16 + 4 * q) = make_float4(gz1[0], gz1[1], gz1[2], gz1[3]);
            *reinterpret_cast<float4 *>(xs + 4 * 256 + e * 16 + 4 * q) = make_float4(hz[0], hz[1], hz[2], hz[3]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = (4 * s + q) * 16 + e;             // edge 4s + q of the tile, channel e
                const float g2lo = xs[0 * 256 + row], g2hi = xs[1 * 256 + row];
                const float av = xs[2 * 256 + row], g1v = xs[3 * 256 + row], hv = xs[4 * 256 + row];
                dw2[0] = SM_MFMA(g2lo, av, dw2[0]);               // dW2[c][i] += gz2[edge][c] a1[edge][i]
                dw2[1] = SM_MFMA(g2hi, av, dw2[1]);
                dw1 = SM_MFMA(g1v, hv, dw1);                      // dW1[c][i] += gz1[edge][c] h[edge][i]
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) hq[r] = hn[r];
        }
    }
    // this wave's slab: D[row = 4q + r][col = e]
    float *slab = slabs + (size_t)gw * SM_NW;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        slab[(4 * q + r) * SM_H + e] = dw1[r];
        slab[SM_C1 * SM_H + (4 * q + r) * SM_C1 + e] = dw2[0][r];
        slab[SM_C1 * SM_H + (16 + 4 * q + r) * SM_C1 + e] = dw2[1][r];
    }
}

// slabs -> dW1 | dW2: 64 outputs x 16 slab groups per workgroup; a group adds its slabs in slab order, the 16 group
// sums are added in group order (the order is fixed by the launch shape alone: same bits every run)
__global__ __launch_bounds__(1024) void small_tail_reduce_kernel(const float *__restrict__ slabs, int nslab,
                                                                 float *__restrict__ dW1, float *__restrict__ dW2) {
    __shared__ float part[16][64];
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    float s = 0.0f;
#pragma unroll 4
    for (int w = g; w < nslab; w += 16) s += slabs[(size_t)w * SM_NW + i];
    part[g][col] = s;
    __syncthreads();
    if (g == 0) {
        float t = part[0][col];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += part[k][col];
        if (i < SM_C1 * SM_H) dW1[i] = t;
        else dW2[i - SM_C1 * SM_H] = t;
    }
}

int small_bwd_blocks(long long P) {
    const long long tiles = (P + 15) / 16;
    long long blocks = (tiles + SB_WAVES - 1) / SB_WAVES;
    const long long cap = 1024;                              // 4096 persistent waves (four per SIMD: 86 registers, 20 KB of LDS
                                                             // per workgroup; one per SIMD left every load latency bare): 12 MB of slabs
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" size_t tpg_small_tail_workspace_bytes(long long P, int K) {
    (void)K;
    return sizeof(float) * (size_t)small_bwd_blocks(P) * SB_WAVES * SM_NW;
}

extern "C" int tpg_small_tail_fwd(const void *h, int is_bf16, const float *W1, const float *W2, float slope1, float slope2,
                                  long long P, int K, int H, int C1, int C2, void *out, unsigned char *arg, void *stream) {
    if (P < 0 || K <= 0 || K > 255) return TPG_ERR_ARG;
    if (H != SM_H || C1 != SM_C1 || C2 != SM_C2) return TPG_ERR_UNSUPPORTED;
    if (P == 0) return TPG_OK;
    if (!h || !W1 || !W2 || !out || !arg) return TPG_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(arg)) & 15) return TPG_ERR_ARG;
    const long long tiles = (P + 15) / 16;
    const dim3 grid((unsigned)((tiles + SF_WAVES - 1) / SF_WAVES));
    if (is_bf16)
        hipLaunchKernelGGL(small_tail_fwd_kernel<__hip_bfloat16>, grid, dim3(SF_WAVES * 64), 0, tpg_stream(stream),
                           static_cast<const __hip_bfloat16 *>(h), W1, W2, slope1, slope2, P, K, static_cast<__hip_bfloat16 *>(out), arg);
    else
        hipLaunchKernelGGL(small_tail_fwd_kernel<float>, grid, dim3(SF_WAVES * 64), 0, tpg_stream(stream), static_cast<const float *>(h),
                           W1, W2, slope1, slope2, P, K, static_cast<float *>(out), arg);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_small_tail_bwd(const void *h, const void *out, const void *gout, const unsigned char *arg, int is_bf16,
                                  const float *W1, const float *W2, float slope1, float slope2, long long P, int K, int H,
                                  int C1, int C2, void *gh, float *dW1, float *dW2, void *ws, void *stream) {
    if (P < 0 || K <= 0 || K > 255) return TPG_ERR_ARG;
    if (H != SM_H || C1 != SM_C1 || C2 != SM_C2) return TPG_ERR_UNSUPPORTED;
    if (!dW1 || !dW2) return TPG_ERR_ARG;
    hipStream_t st = tpg_stream(stream);
    if (P == 0) {
        if (hipMemsetAsync(dW1, 0, sizeof(float) * SM_C1 * SM_H, st) != hipSuccess) return TPG_ERR_LAUNCH;
        if (hipMemsetAsync(dW2, 0, sizeof(float) * SM_C2 * SM_C1, st) != hipSuccess) return TPG_ERR_LAUNCH;
        return TPG_OK;
    }
    if (!h || !out || !gout || !arg || !W1 || !W2 || !gh || !ws) return TPG_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(gout) |
         reinterpret_cast<uintptr_t>(arg) | reinterpret_cast<uintptr_t>(gh) | reinterpret_cast<uintptr_t>(ws)) & 15)
        return TPG_ERR_ARG;
    const int blocks = small_bwd_blocks(P);
    const size_t smem = 0;
    float *slabs = static_cast<float *>(ws);
    if (is_bf16)
        hipLaunchKernelGGL(small_tail_bwd_kernel<__hip_bfloat16>, dim3(blocks), dim3(SB_WAVES * 64), smem, st,
                           static_cast<const __hip_bfloat16 *>(h), static_cast<const __hip_bfloat16 *>(out),
                           static_cast<const __hip_bfloat16 *>(gout), arg, W1, W2, slope1, slope2, P, K,
                           static_cast<__hip_bfloat16 *>(gh), slabs);
    else
        hipLaunchKernelGGL(small_tail_bwd_kernel<float>, dim3(blocks), dim3(SB_WAVES * 64), smem, st, static_cast<const float *>(h),
                           static_cast<const float *>(out), static_cast<const float *>(gout), arg, W1, W2, slope1, slope2, P, K,
                           static_cast<float *>(gh), slabs);
    TPG_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(small_tail_reduce_kernel, dim3(SM_NW / 64), dim3(1024), 0, st, slabs, blocks * SB_WAVES, dW1, dW2);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}
