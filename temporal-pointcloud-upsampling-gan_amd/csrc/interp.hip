// three_nn / three_interpolate for gfx950 plus library identification.
//
// pointnet2_utils.three_nn / three_interpolate have NO call site in the
// reference; they are exported so that
// `pointnet2_ops.pointnet2_utils` is import-complete.  Simple thread-per-output
// kernels: the known cloud is read through the scalar/L1 path (every lane of a
// wave walks the same point), which is all these sizes need.
#include "tpg_common.hpp"

namespace {

__global__ __launch_bounds__(256) void three_nn_kernel(const float *__restrict__ unknown,
                                                       const float *__restrict__ known, int n, int m,
                                                       float *__restrict__ dist2,
                                                       int32_t *__restrict__ idx) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *q = unknown + ((size_t)b * n + i) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float *kn = known + (size_t)b * m * 3;
    float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    for (int k = 0; k < m; ++k) {
        const float d = tpg_sq3(qx, qy, qz, kn[k * 3], kn[k * 3 + 1], kn[k * 3 + 2]);
        if (d < b0) { b2 = b1; i2 = i1; b1 = b0; i1 = i0; b0 = d; i0 = k; }
        else if (d < b1) { b2 = b1; i2 = i1; b1 = d; i1 = k; }
        else if (d < b2) { b2 = d; i2 = k; }
    }
    float *od = dist2 + ((size_t)b * n + i) * 3;
    int32_t *oi = idx + ((size_t)b * n + i) * 3;
    od[0] = b0; od[1] = b1; od[2] = b2;
    oi[0] = i0; oi[1] = i1; oi[2] = i2;
}

__global__ __launch_bounds__(256) void three_interp_fwd_kernel(const float *__restrict__ feat,
                                                               const int32_t *__restrict__ idx,
                                                               const float *__restrict__ w, int C,
                                                               int m, int n, float *__restrict__ out) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t *id = idx + ((size_t)b * n + i) * 3;
    const float *ww = w + ((size_t)b * n + i) * 3;
    const float *f = feat + ((size_t)b * C + c) * m;
    float acc = ww[0] * f[tpg_clamp_idx(id[0], m)];
    acc = acc + ww[1] * f[tpg_clamp_idx(id[1], m)];
    acc = acc + ww[2] * f[tpg_clamp_idx(id[2], m)];
    out[((size_t)b * C + c) * n + i] = acc;
}

__global__ __launch_bounds__(256) void three_interp_bwd_kernel(const float *__restrict__ gout,
                                                               const int32_t *__restrict__ idx,
                                                               const float *__restrict__ w, int C,
                                                               int m, int n,
                                                               float *__restrict__ gfeat) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t *id = idx + ((size_t)b * n + i) * 3;
    const float *ww = w + ((size_t)b * n + i) * 3;
    const float go = gout[((size_t)b * C + c) * n + i];
    float *g = gfeat + ((size_t)b * C + c) * m;
#pragma unroll
    for (int l = 0; l < 3; ++l) atomicAdd(g + tpg_clamp_idx(id[l], m), go * ww[l]);
}

}  // namespace

extern "C" const char *tpg_version(void) { return "tpgan_ops 0.1.0 (round 1)"; }
extern "C" const char *tpg_target_arch(void) { return "gfx950"; }

extern "C" int tpg_three_nn_f32(const float *unknown, const float *known, int B, int n, int m,
                                float *dist2, int32_t *idx, void *stream) {
    if (B < 0 || n < 0 || m <= 0) return TPG_ERR_ARG;
    if (B == 0 || n == 0) return TPG_OK;
    if (!unknown || !known || !dist2 || !idx || B > 65535) return TPG_ERR_ARG;
    hipLaunchKernelGGL(three_nn_kernel, dim3((n + 255) / 256, B), dim3(256), 0, tpg_stream(stream),
                       unknown, known, n, m, dist2, idx);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_three_interp_fwd_f32(const float *feat, const int32_t *idx, const float *w, int B,
                                        int C, int m, int n, float *out, void *stream) {
    if (B < 0 || C < 0 || m <= 0 || n < 0) return TPG_ERR_ARG;
    if (B == 0 || C == 0 || n == 0) return TPG_OK;
    if (!feat || !idx || !w || !out || B > 65535 || C > 65535) return TPG_ERR_ARG;
    hipLaunchKernelGGL(three_interp_fwd_kernel, dim3((n + 255) / 256, C, B), dim3(256), 0,
                       tpg_stream(stream), feat, idx, w, C, m, n, out);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_three_interp_bwd_f32(const float *gout, const int32_t *idx, const float *w, int B,
                                        int C, int m, int n, float *gfeat, void *stream) {
    if (B < 0 || C < 0 || m <= 0 || n < 0) return TPG_ERR_ARG;
    if (B == 0 || C == 0) return TPG_OK;
    if (!gfeat || B > 65535 || C > 65535) return TPG_ERR_ARG;
    hipStream_t st = tpg_stream(stream);
    if (hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)B * C * m, st) != hipSuccess) return TPG_ERR_LAUNCH;
    if (n == 0) return TPG_OK;
    if (!gout || !idx || !w) return TPG_ERR_ARG;
    hipLaunchKernelGGL(three_interp_bwd_kernel, dim3((n + 255) / 256, C, B), dim3(256), 0, st, gout, idx,
                       w, C, m, n, gfeat);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}
