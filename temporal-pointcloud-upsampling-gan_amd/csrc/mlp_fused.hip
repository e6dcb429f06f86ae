// Fused shared-MLP tail layers on channels-last bf16 rows: the grouped-feature x MLP-weight
// contraction of set abstraction / flow embedding on hand-written MFMA tiles.
//
// The reference runs every layer of a shared MLP as  conv1x1 -> BatchNorm2d -> (Leaky)ReLU  on a
// (B,C,S,ns) tensor and ends with a max over ns (discriminator.py:63-78,140-148,276-282).  On rows
// (P = B*S*ns, C) the layer l -> l+1 of such a tail is
//
//     x_{l+1} = W_{l+1} . lrelu(BN_l(x_l))              BN in training mode (batch statistics)
//
// One launch of mlp_fwd_kernel does, per tile of rows:
//   prologue  the 16-byte A fragments are loaded straight from x_l (a lane's 8 consecutive
//             channels of one row ARE the A operand of v_mfma_f32_16x16x32_bf16), BatchNorm
//             scale / shift and the LeakyReLU are applied in registers, packed back to bf16;
//   MFMA      against W_{l+1} (bf16, staged once per workgroup in LDS, fragment order);
//   epilogue  x_{l+1} is rounded to bf16 and stored, and the per-channel sums of BN_{l+1}
//             (about a per-wave pivot, combined Chan-style: exact enough for E[x^2]-E[x]^2 at any
//             mean / sigma ratio) are accumulated from the ROUNDED values -- the statistics of the
//             tensor the next stage reads.
// A small finalize launch turns the per-workgroup partials into mean / rstd (+ running statistics,
// batch counter) exactly like rowbn_stats_finalize_kernel.  What this replaces per layer:
// rowbn_apply (read + write of x_l), the hipBLASLt GEMM (read x_l, write x_{l+1}) and rowbn_stats
// (read x_{l+1}): 2 C_l + 1 C_{l+1} of the 3 C_l + 2 C_{l+1} tensor widths moved.
//
// Output channel order trick: an MFMA column index is free to mean any channel.  Tile t, column j
// is channel j*T + t (T = Cout/16 tiles), so a lane's T accumulators of one row are T CONSECUTIVE
// channels: the row store is one 16 / 32-byte vector per lane and 16 lanes write a whole row.
//
// Arithmetic intensity: 2*Cin*Cout flops per 2*(Cin+Cout) bytes = 43 flop/B at 64 -> 128,
// 128 flop/B at 256 -> 256, against a ridge of ~310 flop/B (2.5 PFLOP/s / 8 TB/s): HBM-bound by
// design; the MFMA pipe runs at 10-40 % while the rows stream.
#include <hip/hip_bf16.h>

#include "tpg_common.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A / B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // one 16x16 accumulator tile (4 VGPRs)

constexpr int ML_THREADS = 256;
constexpr int ML_WAVES = ML_THREADS / 64;
constexpr int ML_MAX_BLOCKS = 512;    // most workgroups (= partial rows) per segment
constexpr int ML_WPAD = 8;            // bf16 elements of padding per LDS weight row (16 B)

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    // round-to-nearest-even, NaN stays NaN: the compiler emits v_cvt_pk_bf16_f32
    const __hip_bfloat16 a = __float2bfloat16(lo), b = __float2bfloat16(hi);
    return (unsigned)(*reinterpret_cast<const unsigned short *>(&a)) |
           ((unsigned)(*reinterpret_cast<const unsigned short *>(&b)) << 16);
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// LDS row of weight row n (output channel n): tile-major so that the 16 lanes of a fragment read
// read 16 CONSECUTIVE padded rows (conflict-free ds_read_b128)
template <int COUT> __device__ __forceinline__ int w_lds_row(int n) {
    constexpr int T = COUT / 16;
    return (n % T) * 16 + n / T;
}

// ------------------------------------------------------------------------------------ forward
// grid (G, nseg); a workgroup walks tiles blockIdx.x, blockIdx.x + G, ... of its segment.
// x (nseg*P, CIN) bf16; ss = BatchNorm scale | shift of the INPUT, 2*CIN floats at ss + seg*ss_stride (NULL: none);
// W (nseg or 1, COUT, CIN) f32; y (nseg*P, COUT) bf16; part (nseg, G, 3, COUT) f32 = mean | M2 | n.
template <int CIN, int COUT, int STRIPS>
__global__ __launch_bounds__(ML_THREADS, 2) void mlp_fwd_kernel(
    const __hip_bfloat16 *__restrict__ x, long long P, const float *__restrict__ ss, int ss_stride, float slope,
    const float *__restrict__ W, int w_per_seg, __hip_bfloat16 *__restrict__ y, float *__restrict__ part) {
    constexpr int T = COUT / 16;            // column tiles = consecutive channels per lane
    constexpr int KS = CIN / 32;            // k-steps
    constexpr int BM = ML_WAVES * STRIPS * 16;
    constexpr int WROW = CIN + ML_WPAD;     // bf16 elements per LDS weight row
    extern __shared__ __attribute__((aligned(16))) unsigned char ml_smem[];
    unsigned short *wl = reinterpret_cast<unsigned short *>(ml_smem);                  // [COUT][WROW]
    float *cst = reinterpret_cast<float *>(ml_smem + (size_t)COUT * WROW * 2);         // [2][CIN]
    // [ML_WAVES][3][COUT], used after the tile loop only: it lies ON the weight (12 KB of its own put the 128 -> 256
    // layer at 81 KB, 2 KB too many for two workgroups per CU -- the launch ran at one wave per SIMD in 1.5 rounds)
    float *red = reinterpret_cast<float *>(ml_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int seg = blockIdx.y;
    x += (size_t)seg * P * CIN;
    y += (size_t)seg * P * COUT;
    // ---- stage the weight (fp32 -> bf16) and the input BatchNorm constants
    {
        const float *Ws = W + (w_per_seg ? (size_t)seg * COUT * CIN : 0);
        for (int e = tid; e < COUT * CIN / 4; e += ML_THREADS) {
            const int n = e / (CIN / 4), k = (e - n * (CIN / 4)) * 4;
            const float4 w4 = *reinterpret_cast<const float4 *>(Ws + (size_t)n * CIN + k);
            uint2 p;
            p.x = pack_bf16x2(w4.x, w4.y);
            p.y = pack_bf16x2(w4.z, w4.w);
            *reinterpret_cast<uint2 *>(wl + (size_t)w_lds_row<COUT>(n) * WROW + k) = p;
        }
        for (int c = tid; c < 2 * CIN; c += ML_THREADS)
            cst[c] = ss ? ss[(size_t)seg * ss_stride + c] : (c < CIN ? 1.0f : 0.0f);
    }
    __syncthreads();
    // statistics of the lane's T output channels li*T + t over the rows it sees
    float piv[T], s1[T], s2[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { piv[t] = 0.0f; s1[t] = 0.0f; s2[t] = 0.0f; }
    bool have_piv = false;
    float nrows = 0.0f;                       // valid rows this WAVE has accumulated
    const long long ntiles = (P + BM - 1) / BM;
    const unsigned short *wfrag = wl + (size_t)li * WROW + 8 * lq;    // + t*16*WROW + 32*s
    const float *csc = cst + 8 * lq, *csh = cst + CIN + 8 * lq;       // + 32*s: the lane's 8 channels of k-step s
    uint4 araw[STRIPS][KS];
    auto load_tile = [&](long long tile) {
#pragma unroll
        for (int st = 0; st < STRIPS; ++st) {
            long long row = tile * BM + (wave * STRIPS + st) * 16 + li;
            row = row < P ? row : P - 1;      // tail rows: clamped loads, masked below
            const __hip_bfloat16 *px = x + (size_t)row * CIN + 8 * lq;
#pragma unroll
            for (int s = 0; s < KS; ++s) araw[st][s] = *reinterpret_cast<const uint4 *>(px + 32 * s);
        }
    };
    long long tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const long long row_base = tile * BM + (long long)wave * STRIPS * 16;
        f32x4 acc[STRIPS][T];
#pragma unroll
        for (int st = 0; st < STRIPS; ++st)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[st][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // k-step by k-step: BatchNorm + LeakyReLU on this step's A fragments (registers), then the
        // step's T column tiles -- one A fragment per strip and a few B fragments live at a time
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float sc[8], sh[8];
            {
                const float4 a0 = *reinterpret_cast<const float4 *>(csc + 32 * s), a1 = *reinterpret_cast<const float4 *>(csc + 32 * s + 4);
                const float4 b0 = *reinterpret_cast<const float4 *>(csh + 32 * s), b1 = *reinterpret_cast<const float4 *>(csh + 32 * s + 4);
                sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
                sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
            }
            bf16x8 afrag[STRIPS];
#pragma unroll
            for (int st = 0; st < STRIPS; ++st) {
                const unsigned w[4] = {araw[st][s].x, araw[st][s].y, araw[st][s].z, araw[st][s].w};
                union { unsigned u[4]; bf16x8 v; } cv;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float a = __builtin_fmaf(bf16_lo(w[i]), sc[2 * i], sh[2 * i]);
                    float b = __builtin_fmaf(bf16_hi(w[i]), sc[2 * i + 1], sh[2 * i + 1]);
                    a = fmaxf(a, a * slope);            // LeakyReLU for 0 <= slope <= 1
                    b = fmaxf(b, b * slope);
                    cv.u[i] = pack_bf16x2(a, b);
                }
                afrag[st] = cv.v;
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(wfrag + (size_t)t * 16 * WROW + 32 * s);
#pragma unroll
                for (int st = 0; st < STRIPS; ++st)
                    acc[st][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[st], b, acc[st][t], 0, 0, 0);
                // hipcc would hoist every B-fragment read of the tile to its top (T*KS*4 registers,
                // hundreds of spills): keep at most four fragments' reads ahead of their MFMAs
                if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the next tile's rows travel while this tile's epilogue runs (requesting them k-step by k-step inside the
        // loop, as the data-gradient kernel does, measured 0-7 % SLOWER here: two waves per SIMD already cover it)
        if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x);
        // ---- epilogue: round, store rows, accumulate the statistics of the rounded values
        if (!have_piv) {
            // pivot = this wave's first row (strip 0, row 0: held by the lanes with lq == 0 in
            // register 0), rounded like the stored value
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float v = bf16_lo(pack_bf16x2(acc[0][t][0], 0.0f));
                piv[t] = __shfl(v, li, 64);
            }
            have_piv = true;
        }
#pragma unroll
        for (int st = 0; st < STRIPS; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long row = row_base + st * 16 + 4 * lq + r;
                const bool valid = row < P;
                unsigned o[T / 2];
#pragma unroll
                for (int t = 0; t < T; t += 2) {
                    const unsigned w = pack_bf16x2(acc[st][t][r], acc[st][t + 1][r]);
                    o[t / 2] = w;
                    const float d0 = valid ? bf16_lo(w) - piv[t] : 0.0f;
                    const float d1 = valid ? bf16_hi(w) - piv[t + 1] : 0.0f;
                    s1[t] += d0;
                    s1[t + 1] += d1;
                    s2[t] = __builtin_fmaf(d0, d0, s2[t]);
                    s2[t + 1] = __builtin_fmaf(d1, d1, s2[t + 1]);
                }
                if (valid) {
                    __hip_bfloat16 *py = y + (size_t)row * COUT + li * T;
#pragma unroll
                    for (int v4 = 0; v4 < T / 8; ++v4)
                        reinterpret_cast<uint4 *>(py)[v4] = make_uint4(o[4 * v4], o[4 * v4 + 1], o[4 * v4 + 2], o[4 * v4 + 3]);
                    if constexpr (T == 4) *reinterpret_cast<uint2 *>(py) = make_uint2(o[0], o[1]);
                }
            }
        {
            const long long left = P - row_base;
            nrows += (float)(left <= 0 ? 0 : (left < STRIPS * 16 ? left : STRIPS * 16));
        }
    }
    // ---- per-wave sums -> per-workgroup (mean, M2, n), Chan's combination
#pragma unroll
    for (int t = 0; t < T; ++t) {
        s1[t] += __shfl_xor(s1[t], 16, 64);
        s1[t] += __shfl_xor(s1[t], 32, 64);
        s2[t] += __shfl_xor(s2[t], 16, 64);
        s2[t] += __shfl_xor(s2[t], 32, 64);
    }
    __syncthreads();                          // every wave is done with the weight: `red` takes its place
    if (lq == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int c = li * T + t;
            const float n = nrows;
            const float m = n > 0.0f ? s1[t] / n : 0.0f;              // mean about the pivot
            red[(wave * 3 + 0) * COUT + c] = piv[t] + m;
            red[(wave * 3 + 1) * COUT + c] = n > 0.0f ? s2[t] - s1[t] * m : 0.0f;
            red[(wave * 3 + 2) * COUT + c] = n;
        }
    }
    __syncthreads();
    float *out = part + ((size_t)seg * gridDim.x + blockIdx.x) * 3 * COUT;
    for (int c = tid; c < COUT; c += ML_THREADS) {
        float n = 0.0f, mean = 0.0f, m2 = 0.0f;
#pragma unroll
        for (int w = 0; w < ML_WAVES; ++w) {
            const float nw = red[(w * 3 + 2) * COUT + c];
            if (nw > 0.0f) {
                const float mw = red[(w * 3 + 0) * COUT + c], qw = red[(w * 3 + 1) * COUT + c];
                const float tot = n + nw, delta = mw - mean;
                mean += delta * (nw / tot);
                m2 += qw + delta * delta * (n * nw / tot);
                n = tot;
            }
        }
        out[c] = mean;
        out[COUT + c] = m2;
        out[2 * COUT + c] = n;
    }
}

// per-workgroup (mean, M2, n) -> mean, rstd of every segment (+ running statistics and the batch
// counter, as nn.BatchNorm's forward does); grid = ceil(C / 4), 64 lanes per channel over the
// partials, fp64, fixed order -> bitwise reproducible.  ss_next (nseg, 2, C), optional: scale | shift
// of THIS BatchNorm (gamma * rstd | beta - mean * gamma * rstd) for the consumer's prologue.
constexpr int MF_CH = 4;
__device__ __forceinline__ double mf_wave_sum(double v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);      // fixed butterfly: every lane ends with the same bits
    return v;
}

// The segments of a launch are independent except for the running statistics, and a segment has few partials
// (segments x partials <= the resident workgroups of the producing launch), so a wave's lanes are split into GROUPS,
// one per segment: SP = segments per sweep (a power of two), L = 64 / SP lanes per segment.
struct MfSplit { int SP, L; };
__device__ __forceinline__ MfSplit mf_split(int nseg) {
    int sp = 1;
    while (sp < nseg && sp < 64) sp <<= 1;
    return {sp, 64 / sp};
}
__device__ __forceinline__ double mf_group_sum(double v, int L) {
    for (int m = L >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);      // fixed butterfly inside the group
    return v;
}
__device__ __forceinline__ double mf_readlane(double v, int l) {          // l wave-uniform
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
constexpr int MF_R = 8;      // partials a lane keeps in registers between the two passes

// One WAVE per channel (MF_CH = 4 channels per workgroup), no LDS and no barrier.  All segments of a sweep are
// reduced side by side (a lane group each): the launch waits for memory twice per sweep, not twice per segment
// (segment after segment it was 2 x 16 dependent round trips at cfg4 = 20 us per launch, 42 launches per step).
// Only the running statistics walk the segments in call order, from registers.
__global__ __launch_bounds__(ML_THREADS) void mlp_stats_finalize_kernel(
    const float *__restrict__ part, int G, int C, int nseg, float eps, float momentum,
    float *__restrict__ running_mean, float *__restrict__ running_var, long long *__restrict__ num_batches_tracked,
    const float *__restrict__ mean_shift, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ mean, float *__restrict__ rstd, float *__restrict__ ci_out) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * MF_CH + (threadIdx.x >> 6);
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += nseg;
    if (c >= C) return;                              // whole wave
    const MfSplit sp = mf_split(nseg);
    const int grp = lane / sp.L, sub = lane - grp * sp.L;
    const bool fits = G <= sp.L * MF_R;              // the partials of a segment fit its lanes' registers
    float rmean = running_mean ? running_mean[c] : 0.0f, rvar = running_mean ? running_var[c] : 0.0f;
    const double shift = mean_shift ? (double)mean_shift[c] : 0.0;
    for (int seg0 = 0; seg0 < nseg; seg0 += sp.SP) {
        const int seg = seg0 + grp;
        const bool live = seg < nseg;
        const float *ps = part + (size_t)(live ? seg : nseg - 1) * G * 3 * C + c;
        float pn[MF_R], pm[MF_R], pq[MF_R];
        auto fetch = [&](int g0) {                   // unconditional, clamped: all requests of a block in flight
#pragma unroll
            for (int i = 0; i < MF_R; ++i) {
                const int g = g0 + sub + i * sp.L;
                const bool ok = live && g < G;
                const float *pg = ps + (size_t)(ok ? g : 0) * 3 * C;
                const float m_ = pg[0], q_ = pg[C], n_ = pg[2 * C];
                pm[i] = m_;
                pq[i] = q_;
                pn[i] = ok ? n_ : 0.0f;              // a partial with n = 0 adds nothing to either pass
            }
        };
        // pass 1: N and the weighted mean
        double n = 0.0, sm = 0.0;
        for (int g0 = 0; g0 < G; g0 += sp.L * MF_R) {
            fetch(g0);
#pragma unroll
            for (int i = 0; i < MF_R; ++i) {
                n += (double)pn[i];
                sm += (double)pn[i] * (double)pm[i];
            }
        }
        const double N = mf_group_sum(n, sp.L);
        const double m = N > 0.0 ? mf_group_sum(sm, sp.L) / N : 0.0;
        // pass 2: M2 = sum M2_g + n_g (mean_g - mean)^2
        double q = 0.0;
        for (int g0 = 0; g0 < G; g0 += sp.L * MF_R) {
            if (!fits) fetch(g0);
#pragma unroll
            for (int i = 0; i < MF_R; ++i) {
                const double d = (double)pm[i] - m;
                q += pn[i] > 0.0f ? (double)pq[i] + (double)pn[i] * d * d : 0.0;
            }
        }
        double var = N > 0.0 ? mf_group_sum(q, sp.L) / N : 0.0;          // biased
        var = var < 0.0 ? 0.0 : var;
        const float mu = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
        if (live && sub == 0) {
            if (mean) mean[(size_t)seg * C + c] = mu;
            if (rstd) rstd[(size_t)seg * C + c] = rs;
            if (ci_out) {
                const float a = (gamma ? gamma[c] : 1.0f) * rs;
                float *o = ci_out + (size_t)seg * 4 * C + c;
                o[0] = a; o[C] = (beta ? beta[c] : 0.0f) - mu * a; o[2 * C] = mu; o[3 * C] = rs;
            }
        }
        if (running_mean) {                          // in call order: the running statistics chain
            const int last = nseg - seg0 < sp.SP ? nseg - seg0 : sp.SP;
            for (int k = 0; k < last; ++k) {
                const double Nk = mf_readlane(N, k * sp.L), mk = mf_readlane(m, k * sp.L), vk = mf_readlane(var, k * sp.L);
                const double unbiased = Nk > 1.0 ? vk * Nk / (Nk - 1.0) : vk;
                rmean = (float)((1.0 - momentum) * rmean + momentum * (mk + shift));
                rvar = (float)((1.0 - momentum) * rvar + momentum * unbiased);
            }
        }
    }
    if (running_mean && lane == 0) { running_mean[c] = rmean; running_var[c] = rvar; }
}

// ------------------------------------------------------------------------------------ backward
// Layer x_in (CIN) -> x_out (COUT):  x_out = W . a_in,  a_in = lrelu(sc*x_in + sh)  [BN_in folded].
// Downstream of x_out sits BN_out (+ LeakyReLU, + max over K on the last layer).  With BN_out's
// backward sums c1 = sum(gg)/P, c2 = sum(gg*xhat)/P known, the gradient of x_out is elementwise:
//
//     dx_out = a*gg - f*(x_out - mu) + e      a = gamma*rstd,  f = a*rstd*c2,  e = -a*c1
//
// gg = the gradient arriving at BN_out's output, already multiplied by lrelu'(z):
//   MODE_DENSE  g_out (P,COUT) bf16, produced by the NEXT layer's dgrad epilogue;
//   MODE_MAX    the last layer: a*gg lives on each group's arg-max row only; g_out (P/K,COUT) bf16 holds
//               it per (group, channel) (mlp_max_prep_kernel: a * lrelu'(y) * gout from the forward's
//               output y), arg the row it belongs to.
// dx_out is never stored.  The MFMA operand the kernels build is  d = dx_out - e = a*gg - f*(x - mu)
// (one fma per element -- fma(-f, x, f*mu) -- plus a compare / select for the arg-max row); d is
// the CENTRED gradient, so its bf16 rounding is relative to its own size (with the uncentred
// e - f*x a channel with |mu| >> sigma lost the bits that matter).  The constant e enters as a rank-one
// term: e^T W per input channel in the data gradient's epilogue, e (x) sum_rows(a_in) in the weight
// gradient.
enum { MODE_DENSE = 0, MODE_MAX = 1 };

__device__ __forceinline__ void ld8(const float *p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// 8 consecutive channels of d = dx_out - e of one row -> 4 packed bf16 pairs (fm = f*mu).
//   DENSE: d = a*g - f*x + fm          MAX: d = (arg == k ? ag : 0) - f*x + fm     (gr = g resp. ag = a*gg)
template <int MODE>
__device__ __forceinline__ void d_out8(const uint4 xr, const uint4 gr, const uint2 ar, int k, const float (&a)[8],
                                       const float (&f)[8], const float (&fm)[8], unsigned (&o)[4]) {
    const unsigned xw[4] = {xr.x, xr.y, xr.z, xr.w}, gw[4] = {gr.x, gr.y, gr.z, gr.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = 2 * i + h;
            const float x = h ? bf16_hi(xw[i]) : bf16_lo(xw[i]);
            const float g = h ? bf16_hi(gw[i]) : bf16_lo(gw[i]);
            const float c = __builtin_fmaf(-f[j], x, fm[j]);          // -f * (x - mu)
            if (MODE == MODE_MAX) {
                const int ak = (int)__builtin_amdgcn_ubfe(j < 4 ? ar.x : ar.y, 8 * (j & 3), 8);
                v[h] = c + (ak == k ? g : 0.0f);
            } else {
                v[h] = __builtin_fmaf(a[j], g, c);
            }
        }
        o[i] = pack_bf16x2(v[0], v[1]);
    }
}

// MODE_MAX preparation: ag (rows,C) bf16 = a * (y > 0 ? g : slope*g) per (group, channel); grid.y = segment.
// (lrelu keeps the sign, so the forward's OUTPUT y of the arg-max row tells which branch it took;
// slope == 0: y == 0 there and the gradient is 0 either way.)
__global__ void mlp_max_prep_kernel(const __hip_bfloat16 *__restrict__ g, const __hip_bfloat16 *__restrict__ y,
                                    const float *__restrict__ cb, float slope, long long rows, int C,
                                    __hip_bfloat16 *__restrict__ ag) {
    const int seg = blockIdx.y;
    const long long n8 = rows * C / 8;
    const float *a = cb + (size_t)seg * 4 * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const size_t off = (size_t)seg * rows * C + (size_t)i * 8;
        const int c = (int)((i * 8) % C);
        const uint4 gv = *reinterpret_cast<const uint4 *>(g + off), yv = *reinterpret_cast<const uint4 *>(y + off);
        const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
        float av[8];
        ld8(a + c, av);
        unsigned o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float g0 = bf16_lo(gw[q]), g1 = bf16_hi(gw[q]);
            const float v0 = av[2 * q] * (bf16_lo(yw[q]) > 0.0f ? g0 : g0 * slope);
            const float v1 = av[2 * q + 1] * (bf16_hi(yw[q]) > 0.0f ? g1 : g1 * slope);
            o[q] = pack_bf16x2(v0, v1);
        }
        *reinterpret_cast<uint4 *>(ag + off) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// data gradient: g_in = ((dx_out) . W) * lrelu'(z_in), rounded to bf16 and stored, plus the partial
// sums of BN_in's backward (sum g_in | sum g_in * (x_in - mu); the finalize multiplies by rstd).
// grid (G, nseg).  cbo (nseg,4,COUT) = a|f*mu|e|f of BN_out; cbi (nseg,4,CIN) = sc|sh|mu|rs of BN_in.
// part (nseg, G, 2, CIN).
template <int CIN, int COUT, int MODE, int STRIPS, int OCC>
__global__ __launch_bounds__(ML_THREADS, OCC) void mlp_dgrad_kernel(
    const __hip_bfloat16 *__restrict__ x_out, const __hip_bfloat16 *__restrict__ g_out,
    const uint8_t *__restrict__ arg, int K, const float *__restrict__ cbo,
    const __hip_bfloat16 *__restrict__ x_in, const float *__restrict__ cbi, float slope_in,
    const float *__restrict__ W, int w_per_seg, long long P, __hip_bfloat16 *__restrict__ g_in,
    float *__restrict__ part) {
    constexpr int TI = CIN / 16;            // column tiles of the product = consecutive input channels per lane
    constexpr int KS = COUT / 32;           // k-steps (over the OUTPUT channels)
    constexpr int BM = ML_WAVES * STRIPS * 16;
    constexpr int WROW = COUT + ML_WPAD;
    extern __shared__ __attribute__((aligned(16))) unsigned char ml_smem[];
    unsigned short *wl = reinterpret_cast<unsigned short *>(ml_smem);                  // [CIN][WROW]: W^T
    float *cst = reinterpret_cast<float *>(ml_smem + (size_t)CIN * WROW * 2);          // a | f | e | f*mu, [COUT] each
    float *bias = cst + 4 * COUT;                                                      // [CIN]: e^T W
    float *red = bias + CIN;                                                           // [ML_WAVES][2][CIN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int seg = blockIdx.y;
    x_out += (size_t)seg * P * COUT;
    x_in += (size_t)seg * P * CIN;
    g_in += (size_t)seg * P * CIN;
    if (MODE == MODE_DENSE) g_out += (size_t)seg * P * COUT;
    else { g_out += (size_t)seg * (P / K) * COUT; arg += (size_t)seg * (P / K) * COUT; }
    {
        // W (COUT, CIN) fp32 -> LDS rows = input channel (tile-major), columns = output channel
        const float *Ws = W + (w_per_seg ? (size_t)seg * COUT * CIN : 0);
        for (int e = tid; e < COUT * CIN / 4; e += ML_THREADS) {
            const int n = e / (CIN / 4), k = (e - n * (CIN / 4)) * 4;       // n = output channel, k = input channel
            const float4 w4 = *reinterpret_cast<const float4 *>(Ws + (size_t)n * CIN + k);
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const __hip_bfloat16 b = __float2bfloat16(wv[i]);
                wl[(size_t)w_lds_row<CIN>(k + i) * WROW + n] = *reinterpret_cast<const unsigned short *>(&b);
            }
        }
        const float *cb = cbo + (size_t)seg * 4 * COUT;
        for (int c = tid; c < COUT; c += ML_THREADS) {
            cst[c] = cb[c];                       // a
            cst[COUT + c] = cb[3 * COUT + c];     // f
            cst[2 * COUT + c] = cb[2 * COUT + c]; // e
            cst[3 * COUT + c] = cb[COUT + c];     // f * mu
        }
    }
    __syncthreads();
    // rank-one part of the product: bias[ci] = sum_co e[co] * W[co][ci] (the bf16 weight the MFMA uses)
    for (int ci = tid; ci < CIN; ci += ML_THREADS) {
        const unsigned short *wr = wl + (size_t)w_lds_row<CIN>(ci) * WROW;
        float acc = 0.0f;
        for (int co = 0; co < COUT; ++co) acc = __builtin_fmaf(cst[2 * COUT + co], __uint_as_float((unsigned)wr[co] << 16), acc);
        bias[ci] = acc;
    }
    __syncthreads();
    // BN_in constants of the lane's TI input channels li*TI + t
    float sc[TI], sh[TI], mu[TI], bi[TI], sg[TI], sgx[TI];
#pragma unroll
    for (int t = 0; t < TI; ++t) {
        const int c = li * TI + t;
        sc[t] = cbi[((size_t)seg * 4 + 0) * CIN + c];
        sh[t] = cbi[((size_t)seg * 4 + 1) * CIN + c];
        mu[t] = cbi[((size_t)seg * 4 + 2) * CIN + c];
        bi[t] = bias[c];
        sg[t] = 0.0f;
        sgx[t] = 0.0f;
    }
    const long long ntiles = (P + BM - 1) / BM;
    const unsigned short *wfrag = wl + (size_t)li * WROW + 8 * lq;    // + t*16*WROW + 32*s
    uint4 xraw[STRIPS][KS], graw[STRIPS][KS];
    uint2 araw[STRIPS][KS];
    unsigned xin[STRIPS][4][TI / 2];          // x_in at the accumulator positions (rows 4*lq + r, channels li*TI ..)
    // One wave per SIMD (the wide layers): nothing else covers this wave's waits, and hipcc waits vmcnt(0) at the top
    // of the tile loop -- for EVERYTHING outstanding.  Whatever is requested or stored at the end of a tile is then
    // awaited in full at the top of the next: x_in of the next tile goes into a second register set at the START of
    // the k-loop, and a tile's rows are stored during the NEXT tile's k-loop.
    constexpr bool DEFER = OCC == 1 && CIN == 128;   // (measured: 128 -> 256 +5 %; with 256 input channels the two extra sets cost more than they hide: -13 %)
    unsigned xnext[DEFER ? STRIPS : 1][4][TI / 2];
    unsigned held[DEFER ? STRIPS : 1][4][TI / 2];   // the previous tile's bf16 rows of g_in, not yet stored
    long long held_base = -1;                  // row_base of the held tile (-1: nothing held)
    // x_in at the accumulator positions of a tile (consumed by the epilogue)
    auto load_xin = [&](long long tile, unsigned (&xin)[STRIPS][4][TI / 2]) {
#pragma unroll
        for (int st = 0; st < STRIPS; ++st) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                long long rr = tile * BM + (wave * STRIPS + st) * 16 + 4 * lq + r;
                rr = rr < P ? rr : P - 1;
                const __hip_bfloat16 *pi = x_in + (size_t)rr * CIN + li * TI;
                if constexpr (TI == 4) {
                    const uint2 v = *reinterpret_cast<const uint2 *>(pi);
                    xin[st][r][0] = v.x; xin[st][r][1] = v.y;
                } else {
#pragma unroll
                    for (int v4 = 0; v4 < TI / 8; ++v4) {
                        const uint4 v = reinterpret_cast<const uint4 *>(pi)[v4];
                        xin[st][r][4 * v4] = v.x; xin[st][r][4 * v4 + 1] = v.y; xin[st][r][4 * v4 + 2] = v.z; xin[st][r][4 * v4 + 3] = v.w;
                    }
                }
            }
        }
    };
    // the k-step operands of a tile: the lane's 8 output channels 32 s + 8 lq of row li of every strip
    const __hip_bfloat16 *px[STRIPS], *pg[STRIPS];
    const uint8_t *pa[STRIPS];
    auto point_at = [&](long long tile) {
#pragma unroll
        for (int st = 0; st < STRIPS; ++st) {
            long long row = tile * BM + (wave * STRIPS + st) * 16 + li;
            row = row < P ? row : P - 1;
            px[st] = x_out + (size_t)row * COUT + 8 * lq;
            const long long grow = MODE == MODE_MAX ? (long long)((unsigned)row / (unsigned)K) : row;
            pg[st] = g_out + (size_t)grow * COUT + 8 * lq;
            pa[st] = MODE == MODE_MAX ? arg + (size_t)grow * COUT + 8 * lq : nullptr;
        }
    };
    auto load_step = [&](int st, int s) {
        xraw[st][s] = *reinterpret_cast<const uint4 *>(px[st] + 32 * s);
        graw[st][s] = *reinterpret_cast<const uint4 *>(pg[st] + 32 * s);
        if (MODE == MODE_MAX) araw[st][s] = *reinterpret_cast<const uint2 *>(pa[st] + 32 * s);
        else araw[st][s] = make_uint2(0u, 0u);
    };
    long long tile = blockIdx.x;
    if (tile < ntiles) {
        point_at(tile);
#pragma unroll
        for (int st = 0; st < STRIPS; ++st)
#pragma unroll
            for (int s = 0; s < KS; ++s) load_step(st, s);
        load_xin(tile, xin);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        const long long row_base = tile * BM + (long long)wave * STRIPS * 16;
        // The NEXT tile's k-step operands are requested as soon as this tile's k-step has been turned into its A
        // fragment (the registers are free from then on): they travel during the rest of the k-loop and the epilogue.
        // With the whole request after the epilogue, a wave -- the only one on its SIMD for the wide layers -- sat
        // out the full memory latency once per tile: 65 % of its cycles (SQ_WAIT_ANY), 3.3 of 8 TB/s.
        // (the last tile re-requests itself: UNCONDITIONAL loads keep the loop body straight-line, so that hipcc can
        // count what is outstanding -- behind an `if (has_next)` per request it fell back to vmcnt(0) at the loop top)
        const bool has_next = tile + gridDim.x < ntiles;
        const long long tnext = has_next ? tile + gridDim.x : tile;
        point_at(tnext);
        if constexpr (DEFER) load_xin(tnext, xnext);
        auto store_held = [&]() {
#pragma unroll
            for (int st = 0; st < STRIPS; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long row = held_base + st * 16 + 4 * lq + r;
                    if (row < P) {
                        __hip_bfloat16 *pgo = g_in + (size_t)row * CIN + li * TI;
#pragma unroll
                        for (int v4 = 0; v4 < TI / 8; ++v4)
                            reinterpret_cast<uint4 *>(pgo)[v4] = make_uint4(held[DEFER ? st : 0][r][4 * v4], held[DEFER ? st : 0][r][4 * v4 + 1],
                                                                            held[DEFER ? st : 0][r][4 * v4 + 2], held[DEFER ? st : 0][r][4 * v4 + 3]);
                    }
                }
        };
        f32x4 acc[STRIPS][TI];
#pragma unroll
        for (int st = 0; st < STRIPS; ++st)
#pragma unroll
            for (int t = 0; t < TI; ++t) acc[st][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float ca[8], cf[8], cfm[8];
            if (MODE == MODE_DENSE) ld8(cst + 32 * s + 8 * lq, ca);
            ld8(cst + COUT + 32 * s + 8 * lq, cf);
            ld8(cst + 3 * COUT + 32 * s + 8 * lq, cfm);
            bf16x8 afrag[STRIPS];
#pragma unroll
            for (int st = 0; st < STRIPS; ++st) {
                const long long row = row_base + st * 16 + li;
                union { unsigned u[4]; bf16x8 v; } cv;
                d_out8<MODE>(xraw[st][s], graw[st][s], araw[st][s], MODE == MODE_MAX ? (int)((unsigned)row % (unsigned)K) : 0, ca,
                             cf, cfm, cv.u);
                if (row >= P) { cv.u[0] = 0u; cv.u[1] = 0u; cv.u[2] = 0u; cv.u[3] = 0u; }
                afrag[st] = cv.v;
                load_step(st, s);
            }
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(wfrag + (size_t)t * 16 * WROW + 32 * s);
#pragma unroll
                for (int st = 0; st < STRIPS; ++st)
                    acc[st][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[st], b, acc[st][t], 0, 0, 0);
                if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DEFER) {
                if (s == 0 && held_base >= 0) store_held();
            }
        }
        // ---- epilogue: + e^T W, the activation's derivative from x_in, BN_in's sums, the bf16 rows of g_in
#pragma unroll
        for (int st = 0; st < STRIPS; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long row = row_base + st * 16 + 4 * lq + r;
                const bool valid = row < P;
                const float vm = valid ? 1.0f : 0.0f;        // (rows past P: their d was zeroed, the bias is not)
                unsigned o[TI / 2];
#pragma unroll
                for (int t = 0; t < TI; t += 2) {
                    const float x0 = bf16_lo(xin[st][r][t / 2]), x1 = bf16_hi(xin[st][r][t / 2]);
                    const float z0 = __builtin_fmaf(x0, sc[t], sh[t]), z1 = __builtin_fmaf(x1, sc[t + 1], sh[t + 1]);
                    const float g0 = (acc[st][t][r] + bi[t]) * (z0 > 0.0f ? vm : vm * slope_in);
                    const float g1 = (acc[st][t + 1][r] + bi[t + 1]) * (z1 > 0.0f ? vm : vm * slope_in);
                    o[t / 2] = pack_bf16x2(g0, g1);
                    sg[t] += g0;
                    sg[t + 1] += g1;
                    sgx[t] = __builtin_fmaf(g0, x0 - mu[t], sgx[t]);
                    sgx[t + 1] = __builtin_fmaf(g1, x1 - mu[t + 1], sgx[t + 1]);
                }
                if constexpr (DEFER) {
#pragma unroll
                    for (int w = 0; w < TI / 2; ++w) held[st][r][w] = o[w];
                } else if (valid) {
                    __hip_bfloat16 *pgo = g_in + (size_t)row * CIN + li * TI;
#pragma unroll
                    for (int v4 = 0; v4 < TI / 8; ++v4)
                        reinterpret_cast<uint4 *>(pgo)[v4] = make_uint4(o[4 * v4], o[4 * v4 + 1], o[4 * v4 + 2], o[4 * v4 + 3]);
                    if constexpr (TI == 4) *reinterpret_cast<uint2 *>(pgo) = make_uint2(o[0], o[1]);
                }
            }
        if constexpr (DEFER) {
            held_base = row_base;
#pragma unroll
            for (int st = 0; st < STRIPS; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int w = 0; w < TI / 2; ++w) xin[st][r][w] = xnext[st][r][w];
        } else {
            load_xin(tnext, xin);
        }
    }
    if constexpr (DEFER) {
        if (held_base >= 0) {
            auto store_last = [&]() {
#pragma unroll
                for (int st = 0; st < STRIPS; ++st)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const long long row = held_base + st * 16 + 4 * lq + r;
                        if (row < P) {
                            __hip_bfloat16 *pgo = g_in + (size_t)row * CIN + li * TI;
#pragma unroll
                            for (int v4 = 0; v4 < TI / 8; ++v4)
                                reinterpret_cast<uint4 *>(pgo)[v4] = make_uint4(held[st][r][4 * v4], held[st][r][4 * v4 + 1], held[st][r][4 * v4 + 2], held[st][r][4 * v4 + 3]);
                        }
                    }
            };
            store_last();
        }
    }
    // ---- per-wave sums -> per-workgroup partials (fixed order)
#pragma unroll
    for (int t = 0; t < TI; ++t) {
        sg[t] += __shfl_xor(sg[t], 16, 64);
        sg[t] += __shfl_xor(sg[t], 32, 64);
        sgx[t] += __shfl_xor(sgx[t], 16, 64);
        sgx[t] += __shfl_xor(sgx[t], 32, 64);
    }
    if (lq == 0) {
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            red[(wave * 2 + 0) * CIN + li * TI + t] = sg[t];
            red[(wave * 2 + 1) * CIN + li * TI + t] = sgx[t];
        }
    }
    __syncthreads();
    float *out = part + ((size_t)seg * gridDim.x + blockIdx.x) * 2 * CIN;
    for (int c = tid; c < 2 * CIN; c += ML_THREADS) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < ML_WAVES; ++w) v += red[w * 2 * CIN + c];
        out[c] = v;
    }
}

// partial sums of the data-gradient kernel (sum g | sum g (x - mu)) -> c12 (nseg,2,C) = (sum g / P |
// rstd * sum g (x - mu) / P) per segment, dgamma = sum over segments of sum g xhat, dbeta = of sum g
// (both optional); rstd of segment s at rstd + s*rstd_stride; cb (nseg,4,C), optional: see below; grid ceil(C/4)
__global__ __launch_bounds__(ML_THREADS) void mlp_bwd_finalize_kernel(const float *__restrict__ part, int G, long long P,
                                                                      int C, int nseg, const float *__restrict__ rstd,
                                                                      int rstd_stride, float *__restrict__ c12,
                                                                      float *__restrict__ dgamma,
                                                                      float *__restrict__ dbeta,
                                                                      const float *__restrict__ ci,
                                                                      float *__restrict__ cb) {
    // one wave per channel, a lane group per segment (see mlp_stats_finalize_kernel)
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * MF_CH + (threadIdx.x >> 6);
    if (c >= C) return;
    const MfSplit sp = mf_split(nseg);
    const int grp = lane / sp.L, sub = lane - grp * sp.L;
    double ts = 0.0, tsx = 0.0;
    for (int seg0 = 0; seg0 < nseg; seg0 += sp.SP) {
        const int seg = seg0 + grp;
        const bool live = seg < nseg;
        const int segc = live ? seg : nseg - 1;
        const float *ps = part + (size_t)segc * G * 2 * C + c;
        const float rs = rstd[(size_t)segc * rstd_stride + c];
        double a0 = 0.0, a1 = 0.0;
        for (int g0 = 0; g0 < G; g0 += sp.L * MF_R) {
            float v0[MF_R], v1[MF_R];
#pragma unroll
            for (int i = 0; i < MF_R; ++i) {         // unconditional, clamped: all requests of a block in flight
                const int g = g0 + sub + i * sp.L;
                const bool ok = live && g < G;
                const float *pg = ps + (size_t)(ok ? g : 0) * 2 * C;
                const float x0 = pg[0], x1 = pg[C];
                v0[i] = ok ? x0 : 0.0f;
                v1[i] = ok ? x1 : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < MF_R; ++i) {
                a0 += (double)v0[i];
                a1 += (double)v1[i];
            }
        }
        const double s = mf_group_sum(a0, sp.L), sx = mf_group_sum(a1, sp.L) * (double)rs;
        const bool owner = live && sub == 0;
        ts += mf_wave_sum(owner ? s : 0.0);          // fixed order over the segments of the sweep
        tsx += mf_wave_sum(owner ? sx : 0.0);
        if (owner) {
            const float c1 = (float)(s / (double)P), c2 = (float)(sx / (double)P);
            if (c12) {
                c12[((size_t)seg * 2 + 0) * C + c] = c1;
                c12[((size_t)seg * 2 + 1) * C + c] = c2;
            }
            if (cb) {
                // the folded constants of THIS BatchNorm's backward, for the kernels one layer down:
                // a | f*mu | e | f  (ci = sc | sh | mu | rs of the same BatchNorm)
                const float *ip = ci + (size_t)seg * 4 * C + c;
                const float a = ip[0], mu = ip[2 * C], f = a * ip[3 * C] * c2;
                float *o = cb + (size_t)seg * 4 * C + c;
                o[0] = a; o[C] = f * mu; o[2 * C] = -a * c1; o[3 * C] = f;
            }
        }
    }
    if (lane == 0) {
        if (dbeta) dbeta[c] = (float)ts;
        if (dgamma) dgamma[c] = (float)tsx;
    }
}

// (mean, rstd, gamma, beta[, c12]) -> the folded constants the kernels read:
//   ci (nseg,4,C) = sc | sh | mu | rs            (input side: activation recompute, xhat)
//   cb (nseg,4,C) = a | f*mu | e | f             (output side: dx = a*gg - f*(x - mu) + e), needs c12
__global__ void mlp_consts_kernel(const float *__restrict__ mean, const float *__restrict__ rstd,
                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                  const float *__restrict__ c12, int C, int nseg, float *__restrict__ ci,
                                  float *__restrict__ cb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseg * C) return;
    const int seg = i / C, c = i - seg * C;
    const float mu = mean[i], rs = rstd[i];
    const float a = (gamma ? gamma[c] : 1.0f) * rs;
    const float bz = (beta ? beta[c] : 0.0f) - mu * a;
    if (ci) {
        float *o = ci + (size_t)seg * 4 * C + c;
        o[0] = a; o[C] = bz; o[2 * C] = mu; o[3 * C] = rs;
    }
    if (cb) {
        const float c1 = c12[((size_t)seg * 2 + 0) * C + c], c2 = c12[((size_t)seg * 2 + 1) * C + c];
        float *o = cb + (size_t)seg * 4 * C + c;
        const float f = a * rs * c2;
        o[0] = a; o[C] = f * mu; o[2 * C] = -a * c1; o[3 * C] = f;
    }
}

// weight gradient: dW[co][ci] = sum_rows dx_out[row][co] * a_in[row][ci]
//                             = sum_rows d[row][co] * a_in[row][ci]  +  e[co] * sum_rows a_in[row][ci]
// with d = dx_out - e and a_in = lrelu(sc*x_in + sh) rebuilt from the saved bf16 rows, staged row-major
// in LDS tile by tile and read transposed (ds_read_b64_tr_b16) as MFMA operands with K = rows.
// A thread stages ONE fixed 8-channel chunk of d and one of a_in (its per-channel constants live in
// registers); the next tile's rows are loaded into registers while the current tile is multiplied.
// grid (G, nseg): a workgroup accumulates the whole (COUT x CIN) product of its row tiles in
// registers (wave w owns output-channel tiles w*MT .. w*MT+MT-1) and writes one fp32 slab;
// mlp_wgrad_reduce_kernel sums the G slabs in fixed order.
constexpr int WG_ROWS = 64;                 // rows per staged tile = two MFMA k-steps
constexpr int WG_PAD = 8;                   // bf16 elements of row padding in the staged tiles

__device__ __forceinline__ bf16x8 tr_frag(const unsigned short *tile, int row_stride, int r0, int c0, int lane) {
    // 8 consecutive ROWS r0 .. r0+7 of column c0 + (lane & 15) as one MFMA fragment: two 4-row x
    // 16-column transposing reads.  Lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3.
    typedef __attribute__((ext_vector_type(4))) short s4;
    const int i = lane & 15, q = i >> 2, p = i & 3;
    const unsigned short *a0 = tile + (size_t)(r0 + q) * row_stride + c0 + 4 * p;
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(a0));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(a0 + 4 * row_stride));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

template <int CIN, int COUT, int MODE, int OCC>
__global__ __launch_bounds__(ML_THREADS, OCC) void mlp_wgrad_kernel(
    const __hip_bfloat16 *__restrict__ x_out, const __hip_bfloat16 *__restrict__ g_out,
    const uint8_t *__restrict__ arg, int K, const float *__restrict__ cbo,
    const __hip_bfloat16 *__restrict__ x_in, const float *__restrict__ cbi, float slope_in, long long P,
    float *__restrict__ slab) {
    constexpr int MT = COUT / 16 / ML_WAVES;        // output-channel tiles per wave
    constexpr int NT = CIN / 16;
    constexpr int DROW = COUT + WG_PAD, AROW = CIN + WG_PAD;
    constexpr int DCH = COUT / 8, ACH = CIN / 8;    // 16-byte chunks per row
    constexpr int DN = WG_ROWS * DCH / ML_THREADS;  // chunks of d a thread stages per tile (same channels, rows DR apart)
    constexpr int AN = WG_ROWS * ACH / ML_THREADS;
    constexpr int DR = ML_THREADS / DCH, AR = ML_THREADS / ACH;
    static_assert(MT >= 1 && DN >= 1 && AN >= 1, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char ml_smem[];
    unsigned short *dxt = reinterpret_cast<unsigned short *>(ml_smem);                 // [WG_ROWS][DROW]
    unsigned short *at = dxt + (size_t)WG_ROWS * DROW;                                  // [WG_ROWS][AROW]
    float *ssum = reinterpret_cast<float *>(at + (size_t)WG_ROWS * AROW);               // [AR][CIN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, li = lane & 15;
    const int seg = blockIdx.y;
    x_out += (size_t)seg * P * COUT;
    x_in += (size_t)seg * P * CIN;
    if (MODE == MODE_DENSE) g_out += (size_t)seg * P * COUT;
    else { g_out += (size_t)seg * (P / K) * COUT; arg += (size_t)seg * (P / K) * COUT; }
    // this thread's chunks and their constants
    const int dch = tid % DCH, drow = tid / DCH, ach = tid % ACH, arow = tid / ACH;
    float ca[8], cf[8], cfm[8], sc[8], sh[8], asum[8];
    ld8(cbo + ((size_t)seg * 4 + 0) * COUT + 8 * dch, ca);
    ld8(cbo + ((size_t)seg * 4 + 3) * COUT + 8 * dch, cf);
    ld8(cbo + ((size_t)seg * 4 + 1) * COUT + 8 * dch, cfm);
    ld8(cbi + ((size_t)seg * 4 + 0) * CIN + 8 * ach, sc);
    ld8(cbi + ((size_t)seg * 4 + 1) * CIN + 8 * ach, sh);
#pragma unroll
    for (int i = 0; i < 8; ++i) asum[i] = 0.0f;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const long long ntiles = (P + WG_ROWS - 1) / WG_ROWS;
    uint4 xr[DN], gr[DN], xi[AN];
    uint2 ar[DN];
    auto load_raw = [&](long long tile) {
        const long long r0 = tile * WG_ROWS;
#pragma unroll
        for (int i = 0; i < DN; ++i) {
            long long rr = r0 + drow + i * DR;
            rr = rr < P ? rr : P - 1;
            const long long grow = MODE == MODE_MAX ? (long long)((unsigned)rr / (unsigned)K) : rr;
            xr[i] = *reinterpret_cast<const uint4 *>(x_out + (size_t)rr * COUT + 8 * dch);
            gr[i] = *reinterpret_cast<const uint4 *>(g_out + (size_t)grow * COUT + 8 * dch);
            ar[i] = MODE == MODE_MAX ? *reinterpret_cast<const uint2 *>(arg + (size_t)grow * COUT + 8 * dch) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < AN; ++i) {
            long long rr = r0 + arow + i * AR;
            rr = rr < P ? rr : P - 1;
            xi[i] = *reinterpret_cast<const uint4 *>(x_in + (size_t)rr * CIN + 8 * ach);
        }
    };
    long long tile = blockIdx.x;
    if (tile < ntiles) load_raw(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const long long r0 = tile * WG_ROWS;
        // ---- registers -> the two LDS tiles
#pragma unroll
        for (int i = 0; i < DN; ++i) {
            const int r = drow + i * DR;
            const long long row = r0 + r;
            unsigned o[4];
            d_out8<MODE>(xr[i], gr[i], ar[i], MODE == MODE_MAX ? (int)((unsigned)(row < P ? row : P - 1) % (unsigned)K) : 0, ca, cf, cfm, o);
            if (row >= P) { o[0] = 0u; o[1] = 0u; o[2] = 0u; o[3] = 0u; }
            *reinterpret_cast<uint4 *>(dxt + (size_t)r * DROW + 8 * dch) = make_uint4(o[0], o[1], o[2], o[3]);
        }
#pragma unroll
        for (int i = 0; i < AN; ++i) {
            const int r = arow + i * AR;
            const bool valid = r0 + r < P;
            const unsigned xw[4] = {xi[i].x, xi[i].y, xi[i].z, xi[i].w};
            unsigned o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float a = __builtin_fmaf(bf16_lo(xw[q]), sc[2 * q], sh[2 * q]);
                float b = __builtin_fmaf(bf16_hi(xw[q]), sc[2 * q + 1], sh[2 * q + 1]);
                a = fmaxf(a, a * slope_in);
                b = fmaxf(b, b * slope_in);
                o[q] = valid ? pack_bf16x2(a, b) : 0u;
                asum[2 * q] += bf16_lo(o[q]);          // of the rounded operand the MFMA sees
                asum[2 * q + 1] += bf16_hi(o[q]);
            }
            *reinterpret_cast<uint4 *>(at + (size_t)r * AROW + 8 * ach) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) load_raw(tile + gridDim.x);     // travels during the MFMA phase
        // ---- dW tiles: A' = d^T (m = output channel, k = row), B' = a_in (k = row, n = input channel)
#pragma unroll
        for (int ks = 0; ks < WG_ROWS / 32; ++ks) {
            bf16x8 af[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = tr_frag(dxt, DROW, 32 * ks + 8 * lq, 16 * (wave * MT + m), lane);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const bf16x8 bfr = tr_frag(at, AROW, 32 * ks + 8 * lq, 16 * n, lane);
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr, acc[m][n], 0, 0, 0);
                if ((n & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                     // the tiles are restaged by the next iteration
    }
    // ---- s[ci] = sum_rows a_in: the AR threads of a channel chunk, added in fixed order
#pragma unroll
    for (int i = 0; i < 8; ++i) ssum[(size_t)arow * CIN + 8 * ach + i] = asum[i];
    __syncthreads();
    float sv[NT];                            // s of the lane's columns 16*n + li
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float t = 0.0f;
        for (int r = 0; r < AR; ++r) t += ssum[(size_t)r * CIN + 16 * n + li];
        sv[n] = t;
    }
    // ---- slab (COUT, CIN) fp32: row = output channel 16*(wave*MT+m) + 4*lq + r, column = 16*n + li
    float *out = slab + ((size_t)seg * gridDim.x + blockIdx.x) * COUT * CIN;
    const float *ce = cbo + ((size_t)seg * 4 + 2) * COUT;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 16 * (wave * MT + m) + 4 * lq + r;
            const float e = ce[co];
#pragma unroll
            for (int n = 0; n < NT; ++n) out[(size_t)co * CIN + 16 * n + li] = __builtin_fmaf(e, sv[n], acc[m][n][r]);
        }
}

// dW (nseg, COUT*CIN) = sum over the G slabs: a block sums 64 float4 columns, its four 64-thread
// groups take every fourth slab each and meet in LDS (fixed order: bitwise reproducible)
__global__ __launch_bounds__(256) void mlp_wgrad_reduce_kernel(const float *__restrict__ slab, int G, int n,
                                                                float *__restrict__ dW) {
    __shared__ float4 part[4][64];
    const int seg = blockIdx.y, col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = (blockIdx.x * 64 + col) * 4;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (i < n) {
        const float *ps = slab + (size_t)seg * G * n + i;
#pragma unroll 4
        for (int g = grp; g < G; g += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(ps + (size_t)g * n);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    part[grp][col] = s;
    __syncthreads();
    if (grp == 0 && i < n) {
        float4 t = part[0][col];
#pragma unroll
        for (int q = 1; q < 4; ++q) { t.x += part[q][col].x; t.y += part[q][col].y; t.z += part[q][col].z; t.w += part[q][col].w; }
        *reinterpret_cast<float4 *>(dW + (size_t)seg * n + i) = t;
    }
}

// dx = a * (g - c1 - xhat * c2): BatchNorm backward of the tail's FIRST BatchNorm, whose input came
// from the row gather; g (P,C) bf16 already carries lrelu'.  ci = sc|sh|mu|rs, c12 = c1|c2.
__global__ __launch_bounds__(ML_THREADS) void mlp_bn_bwd_apply_kernel(
    const __hip_bfloat16 *__restrict__ g, const __hip_bfloat16 *__restrict__ x, long long P, int C,
    const float *__restrict__ ci, const float *__restrict__ c12, __hip_bfloat16 *__restrict__ dx) {
    const int cpr = C / 8, rpi = ML_THREADS / cpr;
    const int chunk = threadIdx.x % cpr, rsub = threadIdx.x / cpr, col = chunk * 8;
    if (rsub >= rpi) return;
    const int seg = blockIdx.y;
    g += (size_t)seg * P * C;
    x += (size_t)seg * P * C;
    dx += (size_t)seg * P * C;
    float a[8], mu[8], rs[8], c1[8], c2[8];
    ld8(ci + ((size_t)seg * 4 + 0) * C + col, a);
    ld8(ci + ((size_t)seg * 4 + 2) * C + col, mu);
    ld8(ci + ((size_t)seg * 4 + 3) * C + col, rs);
    ld8(c12 + ((size_t)seg * 2 + 0) * C + col, c1);
    ld8(c12 + ((size_t)seg * 2 + 1) * C + col, c2);
    float e[8], f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {           // dx = a*g + e - f*x
        f[i] = a[i] * rs[i] * c2[i];
        e[i] = a[i] * (mu[i] * rs[i] * c2[i] - c1[i]);
    }
    const long long step = (long long)gridDim.x * rpi;
    long long r = (long long)blockIdx.x * rpi + rsub;
    for (; r + 3 * step < P; r += 4 * step) {
        uint4 gv[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            gv[u] = *reinterpret_cast<const uint4 *>(g + (size_t)(r + u * step) * C + col);
            xv[u] = *reinterpret_cast<const uint4 *>(x + (size_t)(r + u * step) * C + col);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned gw[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w}, xw[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            unsigned o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v0 = __builtin_fmaf(a[2 * i], bf16_lo(gw[i]), __builtin_fmaf(-f[2 * i], bf16_lo(xw[i]), e[2 * i]));
                const float v1 = __builtin_fmaf(a[2 * i + 1], bf16_hi(gw[i]), __builtin_fmaf(-f[2 * i + 1], bf16_hi(xw[i]), e[2 * i + 1]));
                o[i] = pack_bf16x2(v0, v1);
            }
            *reinterpret_cast<uint4 *>(dx + (size_t)(r + u * step) * C + col) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    for (; r < P; r += step) {
        const uint4 gv = *reinterpret_cast<const uint4 *>(g + (size_t)r * C + col);
        const uint4 xv = *reinterpret_cast<const uint4 *>(x + (size_t)r * C + col);
        const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w};
        unsigned o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v0 = __builtin_fmaf(a[2 * i], bf16_lo(gw[i]), __builtin_fmaf(-f[2 * i], bf16_lo(xw[i]), e[2 * i]));
            const float v1 = __builtin_fmaf(a[2 * i + 1], bf16_hi(gw[i]), __builtin_fmaf(-f[2 * i + 1], bf16_hi(xw[i]), e[2 * i + 1]));
            o[i] = pack_bf16x2(v0, v1);
        }
        *reinterpret_cast<uint4 *>(dx + (size_t)r * C + col) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// The same with the row sums the row gather's backward needs next (ROW_SUB: gQ[s] = -sum_k dx[s,k], tpg_rowcombine_bwd):
// a thread owns one 8-channel chunk of one GROUP of K consecutive rows, so the sum is its own -- of the ROUNDED bf16
// values, k ascending, bit for bit what rowsum_neg_kernel makes of the stored rows, without reading them again.
// qneg (nseg * P/K, C) f32.
__global__ __launch_bounds__(ML_THREADS) void mlp_bn_bwd_apply_rowsum_kernel(
    const __hip_bfloat16 *__restrict__ g, const __hip_bfloat16 *__restrict__ x, long long P, int K, int C,
    const float *__restrict__ ci, const float *__restrict__ c12, __hip_bfloat16 *__restrict__ dx,
    float *__restrict__ qneg) {
    const int cpr = C / 8, gpi = ML_THREADS / cpr;
    const int chunk = threadIdx.x % cpr, gsub = threadIdx.x / cpr, col = chunk * 8;
    if (gsub >= gpi) return;
    const int seg = blockIdx.y;
    const long long groups = P / K;
    g += (size_t)seg * P * C;
    x += (size_t)seg * P * C;
    dx += (size_t)seg * P * C;
    qneg += (size_t)seg * groups * C;
    float a[8], mu[8], rs[8], c1[8], c2[8];
    ld8(ci + ((size_t)seg * 4 + 0) * C + col, a);
    ld8(ci + ((size_t)seg * 4 + 2) * C + col, mu);
    ld8(ci + ((size_t)seg * 4 + 3) * C + col, rs);
    ld8(c12 + ((size_t)seg * 2 + 0) * C + col, c1);
    ld8(c12 + ((size_t)seg * 2 + 1) * C + col, c2);
    float e[8], f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {           // dx = a*g + e - f*x
        f[i] = a[i] * rs[i] * c2[i];
        e[i] = a[i] * (mu[i] * rs[i] * c2[i] - c1[i]);
    }
    for (long long s = (long long)blockIdx.x * gpi + gsub; s < groups; s += (long long)gridDim.x * gpi) {
        const size_t base = (size_t)s * K * C + col;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
        auto one = [&](const uint4 gv, const uint4 xv, size_t off) {
            const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w};
            unsigned o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v0 = __builtin_fmaf(a[2 * i], bf16_lo(gw[i]), __builtin_fmaf(-f[2 * i], bf16_lo(xw[i]), e[2 * i]));
                const float v1 = __builtin_fmaf(a[2 * i + 1], bf16_hi(gw[i]), __builtin_fmaf(-f[2 * i + 1], bf16_hi(xw[i]), e[2 * i + 1]));
                o[i] = pack_bf16x2(v0, v1);
                acc[2 * i] -= bf16_lo(o[i]);
                acc[2 * i + 1] -= bf16_hi(o[i]);
            }
            *reinterpret_cast<uint4 *>(dx + off) = make_uint4(o[0], o[1], o[2], o[3]);
        };
        int k = 0;
        for (; k + 4 <= K; k += 4) {
            uint4 gv[4], xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                gv[u] = *reinterpret_cast<const uint4 *>(g + base + (size_t)(k + u) * C);
                xv[u] = *reinterpret_cast<const uint4 *>(x + base + (size_t)(k + u) * C);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) one(gv[u], xv[u], base + (size_t)(k + u) * C);
        }
        for (; k < K; ++k)
            one(*reinterpret_cast<const uint4 *>(g + base + (size_t)k * C), *reinterpret_cast<const uint4 *>(x + base + (size_t)k * C),
                base + (size_t)k * C);
        float *q = qneg + (size_t)s * C + col;
        *reinterpret_cast<float4 *>(q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4 *>(q + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

template <int CIN, int COUT> constexpr int fwd_strips() {
    // two 16-row strips per wave while the accumulators (STRIPS * COUT/16 * 4 registers) and the
    // double-buffered A fragments (2 * STRIPS * CIN/32 * 4) leave two workgroups per CU room
    return (COUT <= 128 && CIN <= 128) ? 2 : 1;
}
template <int CIN, int COUT> constexpr size_t fwd_smem() {
    constexpr size_t w = (size_t)COUT * (CIN + ML_WPAD) * 2, red = sizeof(float) * ML_WAVES * 3 * COUT;
    return (w > red ? w : red) + sizeof(float) * 2 * CIN;      // (the reduction scratch aliases the weight)
}

int fwd_blocks(long long P, int bm, int nseg, int slots = ML_MAX_BLOCKS) {
    const long long tiles = (P + bm - 1) / bm;
    // segments share the chip: ONE round of workgroups in all -- `slots` = what the chip holds at once, 512 at two
    // workgroups per CU, 256 where LDS or registers allow one (six segments at 128 each were 768: one and a half
    // rounds, the second half empty; and every workgroup stages the whole weight before its first tile: 256 KB of
    // fp32 for a 256 x 256 layer, as much as the four tiles it then processed)
    long long cap = slots / nseg;
    if (cap < 16) cap = 16;
    return (int)(tiles < cap ? tiles : cap);
}

template <int CIN, int COUT>
int fwd_launch(const void *x, long long P, int nseg, const float *ss, int ss_stride, float slope, const float *W,
               int w_per_seg, void *y, float *part, int *G_out, hipStream_t st) {
    constexpr int STRIPS = fwd_strips<CIN, COUT>();
    constexpr int BM = ML_WAVES * STRIPS * 16;
    constexpr size_t smem = fwd_smem<CIN, COUT>();
    auto kern = mlp_fwd_kernel<CIN, COUT, STRIPS>;
    if (smem > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem) != hipSuccess)
                return TPG_ERR_UNSUPPORTED;
            raised = true;
        }
    }
    const int G = fwd_blocks(P, BM, nseg, 2 * smem > 160 * 1024 ? 256 : 512);
    *G_out = G;
    hipLaunchKernelGGL(kern, dim3(G, nseg), dim3(ML_THREADS), smem, st, static_cast<const __hip_bfloat16 *>(x), P, ss,
                       ss_stride, slope, W, w_per_seg, static_cast<__hip_bfloat16 *>(y), part);
    return TPG_OK;
}

// two strips per wave only where the accumulators, the three raw operand streams and the 6 per-channel
// constants of the lane's input channels fit 256 registers; workgroups whose weight fills more than
// half the LDS run alone on their CU and may use the whole register file
template <int CIN, int COUT> constexpr int bwd_strips() { return (CIN == 64 && COUT <= 128) ? 2 : 1; }
template <int CIN, int COUT> constexpr size_t dgrad_smem() {
    return (size_t)CIN * (COUT + ML_WPAD) * 2 + sizeof(float) * (4 * COUT + CIN + ML_WAVES * 2 * CIN);
}
template <int CIN, int COUT> constexpr size_t wgrad_smem() {
    return (size_t)WG_ROWS * ((COUT + WG_PAD) + (CIN + WG_PAD)) * 2 + sizeof(float) * (ML_THREADS / (CIN / 8)) * CIN;
}

template <typename Kern> bool raise_lds(Kern kern, size_t smem, bool *raised) {
    if (smem <= 64 * 1024 || *raised) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
        hipSuccess)
        return false;
    *raised = true;
    return true;
}

template <int CIN, int COUT, int MODE>
int dgrad_launch(const void *x_out, const void *g_out, const uint8_t *arg, int K, const float *cbo,
                 const void *x_in, const float *cbi, float slope_in, const float *W, int w_per_seg, long long P, int nseg,
                 void *g_in, float *part, int *G_out, hipStream_t st) {
    constexpr int STRIPS = bwd_strips<CIN, COUT>();
    constexpr int BM = ML_WAVES * STRIPS * 16;
    constexpr size_t smem = dgrad_smem<CIN, COUT>();
    constexpr int OCC = (smem > 60 * 1024 || CIN == 256) ? 1 : 2;
    auto kern = mlp_dgrad_kernel<CIN, COUT, MODE, STRIPS, OCC>;
    static bool raised = false;
    if (!raise_lds(kern, smem, &raised)) return TPG_ERR_UNSUPPORTED;
    const int G = fwd_blocks(P, BM, nseg, OCC == 1 ? 256 : 512);
    *G_out = G;
    hipLaunchKernelGGL(kern, dim3(G, nseg), dim3(ML_THREADS), smem, st, static_cast<const __hip_bfloat16 *>(x_out),
                       static_cast<const __hip_bfloat16 *>(g_out), arg, K, cbo,
                       static_cast<const __hip_bfloat16 *>(x_in), cbi, slope_in, W, w_per_seg, P,
                       static_cast<__hip_bfloat16 *>(g_in), part);
    return TPG_OK;
}

// workgroups (= fp32 slabs) per segment of the weight-gradient launch: enough to fill the chip,
// few enough that writing and re-reading the slabs stays a fraction of the streamed rows
int wgrad_blocks(long long P, int nseg, int Cin, int Cout) {
    const long long tiles = (P + WG_ROWS - 1) / WG_ROWS;
    const int d = nseg > 4 ? 4 : nseg;
    long long g = P * (Cin + Cout) / (8LL * Cin * Cout);         // slab traffic <= 1/2 of the rows'
    const long long lo = 256 / d, hi = 768 / d;                  // ... but never fewer workgroups than CUs
    g = g < lo ? lo : (g > hi ? hi : g);
    // whole rounds: the launch's g * nseg workgroups against what the chip holds at once (one workgroup per CU for
    // the wide layers, two otherwise) -- 576 workgroups on 256 slots were three rounds for two and a quarter of work
    const long long slots = ((long long)Cout * Cin / ML_THREADS >= 128) ? 256 : 512;
    long long k = g * nseg / slots;                              // (down: fewer slabs to write and re-read)
    if (k < 1) k = 1;
    g = k * slots / nseg;
    if (g < 1) g = 1;
    return (int)(g < tiles ? g : tiles);
}

template <int CIN, int COUT, int MODE>
int wgrad_launch(const void *x_out, const void *g_out, const uint8_t *arg, int K, const float *cbo,
                 const void *x_in, const float *cbi, float slope_in, long long P, int nseg, float *slab, int G,
                 hipStream_t st) {
    constexpr size_t smem = wgrad_smem<CIN, COUT>();
    // accumulators: COUT*CIN/256 registers per lane; from 128 of them on a workgroup has its SIMDs alone
    constexpr int OCC = (COUT * CIN / ML_THREADS >= 128) ? 1 : 2;
    auto kern = mlp_wgrad_kernel<CIN, COUT, MODE, OCC>;
    static bool raised = false;
    if (!raise_lds(kern, smem, &raised)) return TPG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(G, nseg), dim3(ML_THREADS), smem, st, static_cast<const __hip_bfloat16 *>(x_out),
                       static_cast<const __hip_bfloat16 *>(g_out), arg, K, cbo,
                       static_cast<const __hip_bfloat16 *>(x_in), cbi, slope_in, P, slab);
    return TPG_OK;
}

bool ml_shape_ok(int Cin, int Cout) {
    return (Cin == 64 || Cin == 128 || Cin == 256) && (Cout == 64 || Cout == 128 || Cout == 256) &&
           !(Cin == 64 && Cout == 256) && !(Cin == 256 && Cout == 64);
}

}  // namespace

#define TPG_ML_SHAPES(X) X(64, 64) X(64, 128) X(128, 64) X(128, 128) X(128, 256) X(256, 128) X(256, 256)

extern "C" size_t tpg_mlp_workspace_bytes(int C, int nseg) {
    if (nseg < 1) nseg = 1;
    // per-workgroup partials (mean | M2 | n) of the forward, (sum g | sum g xhat) of the backward,
    // C = the larger channel count of the layer
    return sizeof(float) * ((size_t)nseg * ML_MAX_BLOCKS * 3 * C + 64);
}

extern "C" int tpg_mlp_fwd(const void *x, long long P, int Cin, int Cout, int nseg, const float *ss_in, int ss_stride,
                           float slope_in, const float *W, int w_per_seg, void *y, float eps, float momentum,
                           float *running_mean, float *running_var, long long *num_batches_tracked,
                           const float *mean_shift, const float *gamma_out, const float *beta_out, float *mean_out,
                           float *rstd_out, float *ci_out, void *ws, void *stream) {
    if (P <= 0 || nseg < 1 || nseg > 65535 || P % nseg) return TPG_ERR_ARG;
    if (!x || !W || !y || !ws || (ss_in && ss_stride < 2 * Cin)) return TPG_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(W) |
         reinterpret_cast<uintptr_t>(ws)) & 15)
        return TPG_ERR_UNSUPPORTED;
    P /= nseg;
    hipStream_t st = tpg_stream(stream);
    float *part = static_cast<float *>(ws);
    int G = 0, rc = TPG_ERR_UNSUPPORTED;
#define TPG_ML_FWD(CI, CO)                                                                                 \
    if (Cin == CI && Cout == CO)                                                                           \
        rc = fwd_launch<CI, CO>(x, P, nseg, ss_in, ss_stride, slope_in, W, w_per_seg, y, part, &G, st);
    TPG_ML_SHAPES(TPG_ML_FWD)
#undef TPG_ML_FWD
    if (rc) return rc;
    // no statistics wanted (a tail without BatchNorm: the generator's EdgeConv MLPs): no finalize launch
    if (mean_out || rstd_out || ci_out || running_mean)
        hipLaunchKernelGGL(mlp_stats_finalize_kernel, dim3((Cout + MF_CH - 1) / MF_CH), dim3(ML_THREADS), 0, st, part, G,
                           Cout, nseg, eps, momentum, running_mean, running_var, num_batches_tracked, mean_shift,
                           gamma_out, beta_out, mean_out, rstd_out, ci_out);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_mlp_consts(const float *mean, const float *rstd, const float *gamma, const float *beta,
                              const float *c12, int C, int nseg, float *ci, float *cb, void *stream) {
    if (C <= 0 || nseg < 1 || !mean || !rstd || (!ci && !cb) || (cb && !c12)) return TPG_ERR_ARG;
    const int n = nseg * C;
    hipLaunchKernelGGL(mlp_consts_kernel, dim3((n + 255) / 256), dim3(256), 0, tpg_stream(stream), mean, rstd, gamma,
                       beta, c12, C, nseg, ci, cb);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_mlp_max_prep(const void *gout, const void *y, const float *cb_out, float slope_out, long long rows,
                                int C, int nseg, void *ag, void *stream) {
    if (rows <= 0 || nseg < 1 || nseg > 65535 || rows % nseg || C <= 0 || C % 8 || !gout || !y || !cb_out || !ag)
        return TPG_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(gout) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(ag)) & 15)
        return TPG_ERR_UNSUPPORTED;
    rows /= nseg;
    const long long n8 = rows * C / 8;
    long long b = (n8 + 255) / 256;
    b = b > 1024 ? 1024 : b;
    hipLaunchKernelGGL(mlp_max_prep_kernel, dim3((unsigned)b, nseg), dim3(256), 0, tpg_stream(stream),
                       static_cast<const __hip_bfloat16 *>(gout), static_cast<const __hip_bfloat16 *>(y), cb_out,
                       slope_out, rows, C, static_cast<__hip_bfloat16 *>(ag));
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_mlp_dgrad(const void *x_out, const void *g_out, const uint8_t *arg, int K, const float *cb_out,
                             const void *x_in, const float *ci_in, float slope_in, const float *W,
                             int w_per_seg, long long P, int Cin, int Cout, int nseg, int mode, void *g_in,
                             float *c12_in, float *dgamma_in, float *dbeta_in, float *cb_in, void *ws, void *stream) {
    if (P <= 0 || nseg < 1 || nseg > 65535 || P % nseg || (mode != MODE_DENSE && mode != MODE_MAX)) return TPG_ERR_ARG;
    if (!x_out || !g_out || !cb_out || !x_in || !ci_in || !W || !g_in || !ws) return TPG_ERR_ARG;
    P /= nseg;
    if (mode == MODE_MAX && (!arg || K <= 0 || K > 256 || P % K)) return TPG_ERR_ARG;
    if (P >= 0x7fffffffLL || !ml_shape_ok(Cin, Cout)) return TPG_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x_out) | reinterpret_cast<uintptr_t>(g_out) | reinterpret_cast<uintptr_t>(x_in) |
         reinterpret_cast<uintptr_t>(g_in) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(ws) |
         reinterpret_cast<uintptr_t>(arg)) & 7)
        return TPG_ERR_UNSUPPORTED;
    hipStream_t st = tpg_stream(stream);
    float *part = static_cast<float *>(ws);
    int G = 0, rc = TPG_ERR_UNSUPPORTED;
#define TPG_ML_DG(CI, CO)                                                                                            \
    if (Cin == CI && Cout == CO)                                                                                     \
        rc = mode == MODE_MAX ? dgrad_launch<CI, CO, MODE_MAX>(x_out, g_out, arg, K, cb_out, x_in, ci_in,   \
                                                               slope_in, W, w_per_seg, P, nseg, g_in, part, &G, st)  \
                              : dgrad_launch<CI, CO, MODE_DENSE>(x_out, g_out, arg, K, cb_out, x_in, ci_in, \
                                                                 slope_in, W, w_per_seg, P, nseg, g_in, part, &G, st);
    TPG_ML_SHAPES(TPG_ML_DG)
#undef TPG_ML_DG
    if (rc) return rc;
    // (ci_in = sc | sh | mu | rs per segment: the finalize reads rs with stride 4*Cin); nothing wanted
    // (an input without BatchNorm: the generator's EdgeConv MLPs): no finalize launch
    if (c12_in || dgamma_in || dbeta_in || cb_in)
        hipLaunchKernelGGL(mlp_bwd_finalize_kernel, dim3((Cin + MF_CH - 1) / MF_CH), dim3(ML_THREADS), 0, st, part, G, P,
                           Cin, nseg, ci_in + 3 * Cin, 4 * Cin, c12_in, dgamma_in, dbeta_in, ci_in, cb_in);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" size_t tpg_mlp_wgrad_workspace_bytes(long long P, int Cin, int Cout, int nseg) {
    if (nseg < 1 || P <= 0) return 0;
    const int G = wgrad_blocks(P / nseg, nseg, Cin, Cout);
    return sizeof(float) * (size_t)G * nseg * Cin * Cout;
}

extern "C" int tpg_mlp_wgrad(const void *x_out, const void *g_out, const uint8_t *arg, int K, const float *cb_out,
                             const void *x_in, const float *ci_in, float slope_in, long long P, int Cin,
                             int Cout, int nseg, int mode, float *dW, void *ws, void *stream) {
    if (P <= 0 || nseg < 1 || nseg > 65535 || P % nseg || (mode != MODE_DENSE && mode != MODE_MAX)) return TPG_ERR_ARG;
    if (!x_out || !g_out || !cb_out || !x_in || !ci_in || !dW || !ws) return TPG_ERR_ARG;
    P /= nseg;
    if (mode == MODE_MAX && (!arg || K <= 0 || K > 256 || P % K)) return TPG_ERR_ARG;
    if (P >= 0x7fffffffLL || !ml_shape_ok(Cin, Cout)) return TPG_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x_out) | reinterpret_cast<uintptr_t>(g_out) | reinterpret_cast<uintptr_t>(x_in) |
         reinterpret_cast<uintptr_t>(dW) | reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(arg)) & 7)
        return TPG_ERR_UNSUPPORTED;
    hipStream_t st = tpg_stream(stream);
    float *slab = static_cast<float *>(ws);
    const int G = wgrad_blocks(P, nseg, Cin, Cout);
    int rc = TPG_ERR_UNSUPPORTED;
#define TPG_ML_WG(CI, CO)                                                                                           \
    if (Cin == CI && Cout == CO)                                                                                    \
        rc = mode == MODE_MAX ? wgrad_launch<CI, CO, MODE_MAX>(x_out, g_out, arg, K, cb_out, x_in, ci_in,  \
                                                               slope_in, P, nseg, slab, G, st)                      \
                              : wgrad_launch<CI, CO, MODE_DENSE>(x_out, g_out, arg, K, cb_out, x_in, ci_in, \
                                                                 slope_in, P, nseg, slab, G, st);
    TPG_ML_SHAPES(TPG_ML_WG)
#undef TPG_ML_WG
    if (rc) return rc;
    const int n = Cin * Cout;
    hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3((n / 4 + 63) / 64, nseg), dim3(256), 0, st, slab, G, n, dW);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_mlp_bn_bwd_apply(const void *g, const void *x, const float *ci, const float *c12, long long P, int C,
                                    int nseg, void *dx, void *stream) {
    if (P <= 0 || nseg < 1 || nseg > 65535 || P % nseg || !g || !x || !ci || !c12 || !dx) return TPG_ERR_ARG;
    if (C <= 0 || C % 8 || C > 2048) return TPG_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15)
        return TPG_ERR_UNSUPPORTED;
    P /= nseg;
    const int rpi = ML_THREADS / (C / 8);
    long long b = (P + (long long)rpi * 8 - 1) / ((long long)rpi * 8);
    const long long cap = 512 / (nseg > 4 ? 4 : nseg);
    b = b < 1 ? 1 : (b > cap ? cap : b);
    hipLaunchKernelGGL(mlp_bn_bwd_apply_kernel, dim3((unsigned)b, nseg), dim3(ML_THREADS), 0, tpg_stream(stream),
                       static_cast<const __hip_bfloat16 *>(g), static_cast<const __hip_bfloat16 *>(x), P, C, ci, c12,
                       static_cast<__hip_bfloat16 *>(dx));
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_mlp_bn_bwd_apply_rowsum(const void *g, const void *x, const float *ci, const float *c12, long long P,
                                           int K, int C, int nseg, void *dx, float *qneg, void *stream) {
    if (P <= 0 || K <= 0 || nseg < 1 || nseg > 65535 || P % nseg || (P / nseg) % K || !g || !x || !ci || !c12 || !dx || !qneg)
        return TPG_ERR_ARG;
    if (C <= 0 || C % 8 || C > 2048) return TPG_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) |
         reinterpret_cast<uintptr_t>(qneg)) & 15)
        return TPG_ERR_UNSUPPORTED;
    P /= nseg;
    const int gpi = ML_THREADS / (C / 8);
    long long b = (P / K + gpi - 1) / gpi;
    // one round of resident workgroups over all segments (six segments at 256 each were a round and a half: 153 us
    // where the plain kernel took 118)
    static const long long slots = [] { const char *e = getenv("TPG_BNROWSUM_SLOTS"); return e ? atoll(e) : 1024ll; }();
    const long long cap = slots / nseg > 0 ? slots / nseg : 1;
    b = b < 1 ? 1 : (b > cap ? cap : b);
    hipLaunchKernelGGL(mlp_bn_bwd_apply_rowsum_kernel, dim3((unsigned)b, nseg), dim3(ML_THREADS), 0, tpg_stream(stream),
                       static_cast<const __hip_bfloat16 *>(g), static_cast<const __hip_bfloat16 *>(x), P, K, C, ci, c12,
                       static_cast<__hip_bfloat16 *>(dx), qneg);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}
