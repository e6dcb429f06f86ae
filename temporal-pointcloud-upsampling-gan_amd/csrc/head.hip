// BatchNorm1d + LeakyReLU + dropout mask of a discriminator's classification head, in one launch each way.
//
// The heads (discriminator.py:503-516,598-612: sn-Linear -> BatchNorm1d -> LeakyReLU [-> Dropout] twice, then
// sn-Linear(., 1)) work on B = batch-size rows of 64..256 features.  PyTorch runs the normalisation as collect
// statistics / transform / update running statistics / count, then the activation and the dropout product: ~8 launches
// of ~3 us forward and ~6 backward per hidden layer, four head evaluations per training step, most of them on the
// chain that ends the step (the temporal discriminator's update).  Here a thread owns a feature column and walks the
// B rows: batch statistics (two passes over the column: mean, then centred squares), the affine map, the activation,
// the mask product and the running-statistics update in one launch; the backward recomputes the sign from the saved
// input and reduces d_gamma / d_beta in the same thread.  Training mode only (eval keeps PyTorch's modules).
#include "tpg_common.hpp"

namespace {

__global__ __launch_bounds__(256) void head_bn_act_fwd_kernel(const float *__restrict__ h, int B, int C,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float *running_mean,
                                                              float *running_var, long long *nbt, float momentum, float eps,
                                                              float slope, const float *__restrict__ mask,
                                                              float *__restrict__ y, float *__restrict__ mean_out,
                                                              float *__restrict__ rstd_out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    float s = 0.0f;
    for (int b = 0; b < B; ++b) s += h[(size_t)b * C + c];
    const float mean = s / (float)B;
    float q = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float d = h[(size_t)b * C + c] - mean;
        q += d * d;
    }
    const float var = q / (float)B;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    for (int b = 0; b < B; ++b) {
        const size_t e = (size_t)b * C + c;
        float z = (h[e] - mean) * rstd * g + be;
        z = z > 0.0f ? z : z * slope;
        y[e] = mask ? z * mask[e] : z;
    }
    mean_out[c] = mean;
    rstd_out[c] = rstd;
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (q / (float)(B - 1));
}

__global__ __launch_bounds__(256) void head_bn_act_bwd_kernel(const float *__restrict__ gy, const float *__restrict__ h,
                                                              const float *__restrict__ mean,
                                                              const float *__restrict__ rstd,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float slope,
                                                              const float *__restrict__ mask, int B, int C,
                                                              float *__restrict__ dh, float *__restrict__ dgamma,
                                                              float *__restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float mu = mean[c], rs = rstd[c];
    const float g = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    float sb = 0.0f, sg = 0.0f;
    for (int b = 0; b < B; ++b) {
        const size_t e = (size_t)b * C + c;
        const float xh = (h[e] - mu) * rs;
        float gz = mask ? gy[e] * mask[e] : gy[e];
        gz = (xh * g + be) > 0.0f ? gz : gz * slope;
        sb += gz;
        sg += gz * xh;
    }
    if (dbeta) dbeta[c] = sb;
    if (dgamma) dgamma[c] = sg;
    const float mb = sb / (float)B, mg = sg / (float)B;
    for (int b = 0; b < B; ++b) {
        const size_t e = (size_t)b * C + c;
        const float xh = (h[e] - mu) * rs;
        float gz = mask ? gy[e] * mask[e] : gy[e];
        gz = (xh * g + be) > 0.0f ? gz : gz * slope;
        dh[e] = g * rs * (gz - mb - xh * mg);
    }
}

}  // namespace

extern "C" int tpg_head_bn_act_fwd(const float *h, int B, int C, const float *gamma, const float *beta,
                                   float *running_mean, float *running_var, long long *num_batches_tracked, float momentum,
                                   float eps, float slope, const float *mask, float *y, float *mean, float *rstd,
                                   void *stream) {
    if (B < 0 || C < 0) return TPG_ERR_ARG;
    if (B == 1) return TPG_ERR_ARG;                       // (as nn.BatchNorm1d: more than one value per channel)
    if (B == 0 || C == 0) return TPG_OK;
    if (!h || !y || !mean || !rstd) return TPG_ERR_ARG;
    hipLaunchKernelGGL(head_bn_act_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, tpg_stream(stream), h, B, C, gamma, beta,
                       running_mean, running_var, num_batches_tracked, momentum, eps, slope, mask, y, mean, rstd);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}

extern "C" int tpg_head_bn_act_bwd(const float *gy, const float *h, const float *mean, const float *rstd,
                                   const float *gamma, const float *beta, float slope, const float *mask, int B, int C,
                                   float *dh, float *dgamma, float *dbeta, void *stream) {
    if (B < 0 || C < 0) return TPG_ERR_ARG;
    if (B == 0 || C == 0) return TPG_OK;
    if (!gy || !h || !mean || !rstd || !dh) return TPG_ERR_ARG;
    hipLaunchKernelGGL(head_bn_act_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, tpg_stream(stream), gy, h, mean, rstd,
                       gamma, beta, slope, mask, B, C, dh, dgamma, dbeta);
    TPG_RETURN_IF_LAUNCH_FAILED();
    return TPG_OK;
}
