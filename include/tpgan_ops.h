/*
 * tpgan_ops.h -- C-ABI of libtpgan_hip.so: the MI355X (gfx950) neighbourhood
 * ops under TPU-GAN's generator, discriminators and losses.
 *
 * The reference (zijieli-Jlee/Temporal-Pointcloud-Upsampling-GAN) is pure
 * Python and never binds a C symbol itself: it calls four third-party CUDA
 * extensions through their Python APIs.  Each entry point below is what those
 * Python APIs bind underneath, restated as a plain C function; the comment on
 * each one cites the reference call site(s) it serves.  The ctypes binding the
 * reference side would add is shown in INTEGRATION.md and implemented in
 * temporal-pointcloud-upsampling-gan_amd/_lib.py.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM), row-major contiguous, fp32 /
 *     int32 / int64 as named; the caller owns every buffer, the library never
 *     allocates, never synchronises and holds no state;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - returns 0 on success, <0 on error (tpg_status); never throws;
 *   - canonical arithmetic: sqdist(a,b) = sum_d (a_d-b_d)^2 accumulated in d
 *     order in fp32 without FMA; neighbour order ascending (dist, idx).
 */
#ifndef TPGAN_OPS_H
#define TPGAN_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    TPG_OK = 0,
    TPG_ERR_ARG = -1,         /* bad shape / null pointer / size out of range   */
    TPG_ERR_LAUNCH = -2,      /* hipGetLastError() != hipSuccess after launch   */
    TPG_ERR_UNSUPPORTED = -3  /* e.g. K > TPG_MAX_K                             */
} tpg_status;

#define TPG_MAX_K 64 /* neighbours per query held one-per-lane in a wave64 */

/* library / device identification; safe to call without a GPU */
const char *tpg_version(void);
const char *tpg_target_arch(void); /* "gfx950" */

/* K nearest neighbours in D-dim space, optional radius cut.
 * Replaces pytorch3d.ops.knn_points -- gcn_lib/pointnet/gcn.py:16-21,38,258;
 * discriminator.py:15-20,33-38 -- and frnn.frnn_grid_points --
 * discriminator.py:27-32; loss.py:105,142,229,256-265;
 * gcn_lib/interpolation.py:20,33.
 * p1 (B,P1,D), p2 (B,P2,D); len1/len2 (B) int64 or NULL (= full).
 * r2 < 0: plain kNN, missing slots dist 0 / idx 0.
 * r2 >= 0: only d < r2 (strict); missing slots dist -1 / idx -1.
 * dist (B,P1,K) f32, idx (B,P1,K) i64, sorted ascending (dist, idx). */
int tpg_knn_f32(const float *p1, const float *p2, const int64_t *len1,
                const int64_t *len2, int B, int P1, int P2, int D, int K,
                float r2, float *dist, int64_t *idx, void *stream);

/* Plain 3-D kNN (r2 < 0 form of tpg_knn_f32, D = 3, K <= 64) on the uniform grid of tpg_frnn_grid_f32 with cells of
 * ~K/2 points: the (2R+1)^3 cells around the query's cell, R grown until the K-th distance lies inside the
 * covered region.  For the first EdgeConv's search on large clouds (gcn_lib/pointnet/gcn.py:38 at cfg5 /
 * rollout sizes) and the Chamfer nearest neighbours (loss.py:125-127 at 16384 points).  Bit-identical to
 * tpg_knn_f32(r2 < 0) incl. the (0, 0) padding; same workspace as tpg_frnn_grid_f32. */
int tpg_knn_grid_f32(const float *p1, const float *p2, const int64_t *len1, const int64_t *len2, int B,
                     int P1, int P2, int K, float *dist, int64_t *idx, void *ws, void *stream);

/* Plain kNN (no radius) in D = 32 / 64 feature space, 2 <= K <= 24, 16-byte aligned rows: the path
 * tpg_knn_f32 takes by itself for clouds of >= 2048 points (gcn_lib/pointnet/gcn.py:38,258 at cfg5's
 * 4096-point low-resolution clouds; upsampling_network.py:159-174 rollouts).  A Gram-matrix filter on
 * the bf16 matrix cores (split operands, v_mfma_f32_32x32x16_bf16) rules candidates out with a rounding bound,
 * the survivors are re-ranked with the canonical distance, and queries the bound cannot settle are redone by
 * the exhaustive kernel: the output equals tpg_knn_f32's bit for bit.  redo = 0 (diagnostic) skips that
 * last launch and leaves idx[b][i][0] = -2 on the unsettled queries. */
int tpg_knn_mfma_f32(const float *p1, const float *p2, const int64_t *len1,
                     const int64_t *len2, int B, int P1, int P2, int D, int K,
                     float *dist, int64_t *idx, int redo, void *stream);

/* The same radius search (r > 0, D = 3, K <= 64) on a UNIFORM GRID, for clouds where the exhaustive
 * kernel stops being free (loss.py:256-265 at 16384 points per cloud; the 10^4..10^5-point rollout,
 * upsampling_network.py:159-174): cells of edge max(r, extent/64), points counting-sorted by cell,
 * one wave per query over its 27 neighbour cells.  Results are bit-identical to tpg_knn_f32 with
 * r2 = fp32(r)*fp32(r): same distance arithmetic, same (dist, idx) selection key, -1 / -1 padding.
 * ws: tpg_frnn_grid_workspace_bytes(B, P2) bytes, 256-byte aligned. */
size_t tpg_frnn_grid_workspace_bytes(int B, int P2);
int tpg_frnn_grid_f32(const float *p1, const float *p2, const int64_t *len1, const int64_t *len2, int B, int P1,
                      int P2, int K, float r, float *dist, int64_t *idx, void *ws, void *stream);

/* Chamfer nearest-neighbour search in both directions (D = 3).
 * Replaces chamferdist.ChamferDistance.forward -- loss.py:125-127,176-181.
 * src (B,N,3), tgt (B,M,3) -> d1,i1 (B,N), d2,i2 (B,M). */
int tpg_chamfer_fwd_f32(const float *src, const float *tgt, int B, int N, int M,
                        float *d1, int64_t *i1, float *d2, int64_t *i2,
                        void *stream);
/* Backward of chamferdist's forward (loss.py:125-127,176-181): g1 (B,N), g2 (B,M) = gradients of d1, d2;
 * gsrc (B,N,3) / gtgt (B,M,3) are overwritten.  No float atomics: each output point sums its own term and the
 * terms of the points that chose it as nearest neighbour (inverted i1 / i2, ascending point order) -- bitwise
 * reproducible and bit-identical to the oracle's two loops.  ws: tpg_chamfer_bwd_workspace_bytes(B, N, M) bytes,
 * 4-byte aligned. */
size_t tpg_chamfer_bwd_workspace_bytes(int B, int N, int M);
int tpg_chamfer_bwd_f32(const float *src, const float *tgt, int B, int N, int M,
                        const int64_t *i1, const int64_t *i2, const float *g1,
                        const float *g2, float *gsrc, float *gtgt, void *ws, void *stream);

/* Furthest point sampling.  Replaces pointnet2_utils.furthest_point_sample --
 * discriminator.py:114.  xyz (B,N,3) -> idx (B,m) int32; temp (B,N) scratch. */
int tpg_fps_f32(const float *xyz, int B, int N, int m, float *temp, int32_t *idx,
                void *stream);

/* The same sampling with the "FPS of an FPS prefix" shortcut (round 3; csrc/fps.hip): every set-abstraction level after
 * the first samples the centres of the level before, in pick order (discriminator.py:114 on :131-137's gather), and
 * the furthest point sampling of such a prefix is 0, 1, ..., m-1 as long as the producing sampling's last maximum was
 * positive.  prefix_out (B) int32 (may be NULL): set to that condition per cloud; prefix_in (B) int32 (may be NULL):
 * the producing launch's flags -- clouds whose flag is set get 0..m-1 without a single round, the others the full
 * algorithm.  Results are identical to tpg_fps_f32 in every case. */
int tpg_fps_prefix_f32(const float *xyz, int B, int N, int m, float *temp, int32_t *idx,
                       const int32_t *prefix_in, int32_t *prefix_out, void *stream);

/* Dataset-side farthest point sampling: sampling.py:50-106 `farthest_point_sampling(pts, k,
 * initial_idx)` (used by train_utils.py:126, train_fluid/tempo_dataset.py:78,
 * train_action/msr_dataset.py:94,130 on the host, numba): start (B) int32 = first pick per cloud
 * (NULL = 0), skip_origin = 0 (every point is eligible; 1 = pointnet2's |x|^2 > 1e-3 rule).
 * Squared distances in fp32, arg-max ties to the smallest index (numpy argmax). */
int tpg_fps_start_f32(const float *xyz, const int32_t *start, int skip_origin, int B, int N, int m,
                      float *temp, int32_t *idx, void *stream);

/* gather_operation fwd/bwd -- discriminator.py:131-137.
 * feat (B,C,N), idx (B,S) -> out (B,C,S); bwd overwrites gfeat (B,C,N). */
int tpg_gather_fwd_f32(const float *feat, const int32_t *idx, int B, int C, int N,
                       int S, float *out, void *stream);
int tpg_gather_bwd_f32(const float *gout, const int32_t *idx, int B, int C, int N,
                       int S, float *gfeat, void *stream);

/* gather_operation on channels-last rows (the layout this build keeps clouds in): out[b,s,:] = rows[b,idx[b,s],:];
 * rows (B,N,C), idx (B,S) int32, out (B,S,C).  Same semantics as tpg_gather_fwd_f32 / _bwd_f32 on the transposed
 * tensors (discriminator.py:131-137), without the two transpose copies around it. */
int tpg_gather_rows_fwd_f32(const float *rows, const int32_t *idx, int B, int N, int S, int C, float *out, void *stream);
int tpg_gather_rows_bwd_f32(const float *gout, const int32_t *idx, int B, int N, int S, int C, float *grows,
                            void *stream);

/* ball_query -- inside pointnet2_utils.QueryAndGroup, discriminator.py:190.
 * xyz (B,N,3), new_xyz (B,S,3) -> idx (B,S,nsample) int32. */
int tpg_ball_query_f32(const float *xyz, const float *new_xyz, int B, int N, int S,
                       float radius, int nsample, int32_t *idx, void *stream);

/* grouping_operation fwd/bwd -- gcn_lib/pointnet/gcn.py:207,261;
 * discriminator.py:270,273; 2x per QueryAndGroup.
 * feat (B,C,N), idx (B,S,K) -> out (B,C,S,K); bwd overwrites gfeat (B,C,N). */
int tpg_group_fwd_f32(const float *feat, const int32_t *idx, int B, int C, int N,
                      int S, int K, float *out, void *stream);
int tpg_group_bwd_f32(const float *gout, const int32_t *idx, int B, int C, int N,
                      int S, int K, float *gfeat, void *stream);

/* three_nn / three_interpolate -- pointnet2_utils API completeness (no call
 * site in the reference).  dist2 holds SQUARED distances. */
int tpg_three_nn_f32(const float *unknown, const float *known, int B, int n, int m,
                     float *dist2, int32_t *idx, void *stream);
int tpg_three_interp_fwd_f32(const float *feat, const int32_t *idx, const float *w,
                             int B, int C, int m, int n, float *out, void *stream);
int tpg_three_interp_bwd_f32(const float *gout, const int32_t *idx, const float *w,
                             int B, int C, int m, int n, float *gfeat,
                             void *stream);

/* ---- channels-last "row combine": gather the OUTPUT rows of the first MLP layer ----------
 * A 1x1 convolution commutes with a gather, so the first layer of every grouped MLP on the
 * path (EdgeConv: gcn_lib/pointnet/gcn.py:207-210; set abstraction: QueryAndGroup + mlps[0],
 * discriminator.py:141-145; FlowEmbedding: discriminator.py:270-280) is applied to the N
 * un-grouped points and its rows are gathered:
 *   mode 0 GATHER: out[b,s,k,:] = U[b,idx[b,s,k],:]
 *   mode 1 SUB   : out[b,s,k,:] = U[b,idx[b,s,k],:] - QE[b,s,:]
 *   mode 2 EDGE  : out[b,s,k,:] = U[b,idx,:] + lrelu(QE[b,idx,:] - QE[b,s,:], slope)  (S == N)
 * U (B,N,C), QE (B,S,C) of dtype_in; idx (B,S,K) int32; out (B,S,K,C) of dtype_out.  Rows
 * are channels-last; C % 4 == 0 when both types are f32, else C % 8 == 0; 16-B aligned bases.
 * Arithmetic is fp32 in registers with one rounding on store (f32 in -> bf16 out keeps the
 * difference U - QE exact before the single bf16 rounding). */
typedef enum { TPG_DTYPE_F32 = 0, TPG_DTYPE_BF16 = 1 } tpg_dtype;

int tpg_rowcombine_fwd(const void *U, const void *QE, const int32_t *idx, int mode, int dtype_in,
                       int dtype_out, int B, int N, int S, int K, int C, float slope, void *out,
                       void *stream);

/* per-cloud inverted index of idx (B,SK) with values in [0,N): offs (B,N+1), list (B,SK)
 * = flat (s,k) entry ids grouped by destination row, ASCENDING inside a group (a stable radix sort:
 * the gather-reduce backward then sums in a fixed order -- bitwise reproducible gradients).
 * tmp: B*SK ints of scratch (may be NULL when N <= 256). */
int tpg_invert_index(const int32_t *idx, int B, int N, int SK, int32_t *offs, int32_t *list,
                     int32_t *tmp, void *stream);

/* backward of tpg_rowcombine_fwd; atomics-free gather-reduce over the inverted index.
 * gout (B,S,K,C) of dtype_out; gU (B,N,C) and gQE (grad of QE: (B,S,C) for SUB and EDGE,
 * ignored for GATHER) of dtype_in.  E = the forward's QE (EDGE only, else NULL). */
int tpg_rowcombine_bwd(const void *gout, const int32_t *idx, const int32_t *offs, const int32_t *list,
                       const void *E, int mode, int dtype_in, int dtype_out, int B, int N, int S, int K,
                       int C, float slope, void *gU, void *gQE, void *stream);

/* EdgeConv front end on ONE product (gcn_lib/pointnet/gcn.py:176-180,207-210: node_affine(group(f)) +
 * edge_affine(group(f) - f_i), both 1x1 conv + LeakyReLU on the same grouped input).  With the two convolutions applied
 * BEFORE the gather, their weights stack into one GEMM: Y (B,N,2C) = f [We; Wn]^T -- columns [0,C) = E = We f, columns
 * [C,2C) = the node term before its activation -- and
 *   out[b,n,k,:] = lrelu(Y[b,idx[b,n,k],C:2C], slope_a) + lrelu(Y[b,idx[b,n,k],0:C] - Y[b,n,0:C], slope_e)
 * i.e. tpg_rowcombine_fwd(mode EDGE) reading both operands from one buffer with the node row's activation folded in.
 * Y of dtype_in, idx (B,N,K) int32, out (B,N,K,C) of dtype_out; C as in tpg_rowcombine_fwd and C*sizeof(dtype_in) % 16 == 0.
 * Backward: gY (B,N,2C) of dtype_in = [gE | gA * lrelu'(Y[:,C:2C])], atomics-free over the inverted index of idx
 * (tpg_invert_index), bitwise reproducible. */
int tpg_rowcombine_edge_fwd(const void *Y, const int32_t *idx, int dtype_in, int dtype_out, int B, int N, int K, int C,
                            float slope_a, float slope_e, void *out, void *stream);
int tpg_rowcombine_edge_bwd(const void *gout, const int32_t *idx, const int32_t *offs, const int32_t *list,
                            const void *Y, int dtype_in, int dtype_out, int B, int N, int K, int C, float slope_a,
                            float slope_e, void *gY, void *stream);

/* ---- BatchNorm1d + LeakyReLU + dropout mask of a classification head (discriminator.py:503-516,598-612) ----------
 * h (B,C) fp32 rows of a head's hidden layer, TRAINING mode: batch statistics over the B rows (biased variance, eps),
 *   z = (h - mean) * rstd * gamma + beta;  y = (z > 0 ? z : slope * z) * mask
 * mask (B,C, may be NULL) = the dropout's scaled keep mask (0 or 1 / (1 - p)), drawn by the caller; running statistics
 * (may be NULL) are updated with `momentum` and the unbiased variance, *num_batches_tracked (may be NULL) is incremented;
 * mean / rstd (C) are outputs, kept for the backward.  B == 1 is an argument error, as in nn.BatchNorm1d.
 * Backward: dh (B,C), dgamma / dbeta (C, may be NULL); the activation's sign is recomputed from h. */
int tpg_head_bn_act_fwd(const float *h, int B, int C, const float *gamma, const float *beta, float *running_mean,
                        float *running_var, long long *num_batches_tracked, float momentum, float eps, float slope,
                        const float *mask, float *y, float *mean, float *rstd, void *stream);
int tpg_head_bn_act_bwd(const float *gy, const float *h, const float *mean, const float *rstd, const float *gamma,
                        const float *beta, float slope, const float *mask, int B, int C, float *dh, float *dgamma,
                        float *dbeta, void *stream);

/* ---- fused BatchNorm + LeakyReLU (+ max over K neighbours) on channels-last rows ----------
 * The [conv -> BatchNorm2d -> (Leaky)ReLU]* -> max-over-nsample tail of every shared MLP
 * (discriminator.py:63-78,145-150,279-282) on rows x (P,C), P = B*S*ns:
 *   z = (x - mean) * gamma * rstd + beta;  y = z > 0 ? z : slope*z   (slope 0 = ReLU, 1 = none)
 *   K == 0: y (P,C).   K > 0: y (P/K,C) = max over each group of K consecutive rows, plus the
 *   arg-max row of every (group, channel) as one byte (first maximum).
 * training != 0: mean / rstd are computed from x (biased variance, eps) and written; running
 * statistics (may be NULL) are updated with `momentum` and the unbiased variance, and
 * *num_batches_tracked (may be NULL) is incremented, as nn.BatchNorm's forward does.
 * training == 0: mean / rstd are inputs (the caller derives them from the running statistics);
 * both NULL = identity statistics (mean 0, rstd 1: a pure activation [+ max]).
 * ws: tpg_rowbn_workspace_bytes(C, nseg) bytes of scratch, 16-byte aligned, not shared by launches
 * that may run concurrently.  gamma / beta may be NULL (1 / 0).
 * mean_shift (C, may be NULL): a per-channel constant the caller has LEFT OUT of x -- the bias of
 * the preceding 1x1 conv.  Training-mode BatchNorm(x + b) == BatchNorm(x), so the bias add (a full
 * pass, or a GEMM epilogue the batched GEMM does not have) is skipped; only the running mean sees
 * b: it is updated with mean(x) + mean_shift.  (Eval mode: the caller passes mean - b.)
 * nseg >= 1 SEGMENTS: the P rows are nseg equal consecutive blocks, each an independent call of
 * the same module (the T frames of a clip, the fake and the real batch -- discriminator.py runs
 * them one after the other): statistics per segment (mean / rstd are (nseg, C)), running
 * statistics and num_batches_tracked updated segment after segment in that order, dgamma / dbeta
 * summed over the segments.  K > 0 groups never straddle segments (K | P/nseg).
 * phase: TPG_BN_PHASE_ALL, or the reduction part / the streaming part alone (two calls with the
 * same arguments and workspace = one ALL call; lets a profiler time each kernel by itself). */
#define TPG_BN_PHASE_ALL 0
#define TPG_BN_PHASE_STATS 1
#define TPG_BN_PHASE_APPLY 2
size_t tpg_rowbn_workspace_bytes(int C, int nseg);
int tpg_rowbn_fwd(const void *x, int dtype_in, long long P, int K, int C, float eps, float momentum,
                  int training, float *running_mean, float *running_var, long long *num_batches_tracked,
                  const float *mean_shift, const float *gamma, const float *beta, float slope, float *mean,
                  float *rstd, void *y,
                  int dtype_out, uint8_t *argmax, void *ws, int nseg, int phase, void *stream);
/* gy: (P,C) for K == 0, (P/K,C) for K > 0, of dtype_g; dx (P,C) of dtype_in; dgamma / dbeta (C) f32
 * (may be NULL).  y / dtype_y (K > 0 only, may be NULL): the forward's output; with it the
 * per-channel sums of the max variant are taken from (gy, y) alone -- the pre-activation of the
 * arg-max row is recovered from y -- instead of gathering one element of x per (group, channel);
 * channels where that inversion is ill-conditioned (|beta| > 4|gamma|, or slope == 0 and
 * y == 0 carrying no gradient anyway) use the gather. */
int tpg_rowbn_bwd(const void *gy, int dtype_g, const void *x, int dtype_in, const uint8_t *argmax,
                  const void *y, int dtype_y, long long P, int K, int C, int training, const float *mean,
                  const float *rstd, const float *gamma, const float *beta, float slope, float *dgamma,
                  float *dbeta, void *dx, void *ws, int nseg, int phase, void *stream);

/* the reduction half of tpg_rowbn_bwd alone: c12 (nseg,2,C) = (sum gg / P | sum gg*xhat / P) per segment
 * (gg = gy * lrelu'), dgamma / dbeta as above; dx is not computed. */
int tpg_rowbn_bwd_sums(const void *gy, int dtype_g, const void *x, int dtype_in, const uint8_t *argmax,
                       const void *y, int dtype_y, long long P, int K, int C, int training, const float *mean,
                       const float *rstd, const float *gamma, const float *beta, float slope, float *dgamma,
                       float *dbeta, float *c12, void *ws, int nseg, void *stream);

/* The statistics / sums launches above with the folded per-channel constants of the fused tail (tpg_mlp_consts)
 * written by the SAME finalize launch -- a tail's first and last BatchNorm need no tpg_mlp_consts launch of their own:
 *   tpg_rowbn_stats_consts   : phase STATS of tpg_rowbn_fwd (training mode) + ci (nseg,4,C) = sc | sh | mu | rs
 *   tpg_rowbn_bwd_sums_consts: tpg_rowbn_bwd_sums + cb (nseg,4,C) = a | f*mu | e | f, and (ag non-NULL; K > 0, y given
 *                              in gy's type, else TPG_ERR_UNSUPPORTED) ag (P/K, C) = tpg_mlp_max_prep's output, written
 *                              by the reduction from the rows it reads anyway */
int tpg_rowbn_stats_consts(const void *x, int dtype_in, long long P, int C, float eps, float momentum,
                           float *running_mean, float *running_var, long long *num_batches_tracked,
                           const float *mean_shift, const float *gamma, const float *beta, float *mean, float *rstd,
                           float *ci, void *ws, int nseg, void *stream);
int tpg_rowbn_bwd_sums_consts(const void *gy, int dtype_g, const void *x, int dtype_in, const uint8_t *argmax,
                              const void *y, int dtype_y, long long P, int K, int C, int training, const float *mean,
                              const float *rstd, const float *gamma, const float *beta, float slope, float *dgamma,
                              float *dbeta, float *c12, float *cb, void *ag, void *ws, int nseg, void *stream);

/* ---- fused shared-MLP tail layer on MFMA tiles (csrc/mlp_fused.hip) --------------------------
 * The grouped-feature x MLP-weight contraction of set abstraction / flow embedding
 * (discriminator.py:63-78,140-148,276-282: conv1x1 -> BatchNorm2d -> LeakyReLU per layer) for one
 * layer of the tail, on channels-last bf16 rows:
 *     y = W . lrelu(scale * x + shift)        x (P,Cin) bf16, W (Cout,Cin) f32, y (P,Cout) bf16
 * scale | shift = the INPUT BatchNorm folded per channel (2*Cin floats at ss_in + seg*ss_stride -- the head of
 * a ci block, below -- NULL = identity), applied
 * together with the LeakyReLU in the MFMA A-operand prologue; bf16 MFMA (v_mfma_f32_16x16x32_bf16),
 * fp32 accumulation, one rounding of y; the epilogue accumulates the batch statistics of y (of the
 * ROUNDED values) and a finalize launch writes mean_out / rstd_out (nseg,Cout), updates
 * running_mean / running_var / num_batches_tracked like nn.BatchNorm (all three may be NULL;
 * mean_shift as in tpg_rowbn_fwd) and, if ci_out != NULL, the folded constants of the OUTPUT
 * BatchNorm, ci (nseg,4,Cout) = sc | sh | mu | rs (gamma_out, beta_out; NULL = 1 / 0): its head is the next
 * layer's ss_in (ss_stride = 4*Cout), the backward kernels read all four.  mean_out = rstd_out = ci_out =
 * running_mean = NULL: no statistics wanted (a tail without BatchNorm), no finalize launch.
 * nseg segments = nseg calls of the layer on equal consecutive row blocks, each with its own
 * statistics and (w_per_seg != 0) its own weight W[seg] (successive spectral-norm iterates).
 * Supported (Cin,Cout): (64,64) (64,128) (128,64) (128,128) (128,256) (256,128) (256,256);
 * ws: tpg_mlp_workspace_bytes(max(Cin,Cout), nseg) bytes, 16-byte aligned. */
size_t tpg_mlp_workspace_bytes(int C, int nseg);
int tpg_mlp_fwd(const void *x, long long P, int Cin, int Cout, int nseg, const float *ss_in, int ss_stride,
                float slope_in, const float *W, int w_per_seg, void *y, float eps, float momentum,
                float *running_mean, float *running_var, long long *num_batches_tracked, const float *mean_shift,
                const float *gamma_out, const float *beta_out, float *mean_out, float *rstd_out, float *ci_out,
                void *ws, void *stream);

/* Backward of such a layer, x_in (P,Cin) -> x_out (P,Cout) = W . lrelu(BN_in(x_in)), followed by
 * BN_out (+ LeakyReLU; + max over groups of K rows on a tail's last layer).  With BN_out's backward
 * sums known (c12), the gradient of x_out is elementwise in saved tensors,
 *     dx_out = a*gg - f*(x_out - mu) + e      (a, f*mu, e, f per channel: tpg_mlp_consts, cb (nseg,4,Cout))
 * and is never stored: it is rebuilt in the MFMA operand prologue of both gradient kernels.
 *   mode 0 DENSE: gg = g_out (P,Cout) bf16, the activated gradient written by the next layer's dgrad
 *   mode 1 MAX  : g_out (P/K,Cout) bf16 = a * lrelu'(y) * (gradient of the max output), one value per
 *                 (group, channel), made by tpg_mlp_max_prep from that gradient and the forward's output y;
 *                 arg = the arg-max bytes: the value belongs to row arg of its group, the others get 0
 * The MFMA operand the kernels build is the centred d = dx_out - e (one fma per element); e enters as a rank-one
 * term (e^T W per input channel in tpg_mlp_dgrad's epilogue, e (x) sum_rows a_in in tpg_mlp_wgrad).
 * tpg_mlp_dgrad: g_in (P,Cin) bf16 = (dx_out . W) * lrelu'(z_in) and BN_in's backward sums:
 *   c12_in (nseg,2,Cin), dgamma_in / dbeta_in (Cin, summed over segments) and cb_in (nseg,4,Cin) = the cb of
 *   BN_in for the NEXT tpg_mlp_dgrad / tpg_mlp_wgrad one layer down -- each may be NULL, all NULL = BN_in is
 *   the identity (no finalize launch).
 *   ci_in (nseg,4,Cin) = sc | sh | mu | rs of BN_in (tpg_mlp_consts).  ws: tpg_mlp_workspace_bytes.
 * tpg_mlp_wgrad: dW (nseg,Cout,Cin) f32 = dx_out^T . lrelu(BN_in(x_in)), both operands rebuilt from the
 *   saved rows, staged in LDS and read transposed (ds_read_b64_tr_b16); per-workgroup fp32 slabs summed
 *   in fixed order (bitwise reproducible).  ws: tpg_mlp_wgrad_workspace_bytes(P, Cin, Cout, nseg).
 * tpg_mlp_bn_bwd_apply: dx = a*(g - c1 - xhat*c2) for the tail's first BatchNorm (g already activated).
 * tpg_mlp_bn_bwd_apply_rowsum: the same, plus qneg (nseg * P/K, C) f32 = -sum over each group of K consecutive rows of
 *   the stored (bf16-rounded) dx, k ascending: what tpg_rowcombine_bwd(mode SUB) computes as gQE from those rows --
 *   pass it there as gQE_ready and the rows are not read a second time. */
int tpg_mlp_consts(const float *mean, const float *rstd, const float *gamma, const float *beta, const float *c12,
                   int C, int nseg, float *ci, float *cb, void *stream);
int tpg_mlp_max_prep(const void *gout, const void *y, const float *cb_out, float slope_out, long long rows, int C,
                     int nseg, void *ag, void *stream);
int tpg_mlp_dgrad(const void *x_out, const void *g_out, const uint8_t *arg, int K, const float *cb_out,
                  const void *x_in, const float *ci_in, float slope_in, const float *W,
                  int w_per_seg, long long P, int Cin, int Cout, int nseg, int mode, void *g_in, float *c12_in,
                  float *dgamma_in, float *dbeta_in, float *cb_in, void *ws, void *stream);
size_t tpg_mlp_wgrad_workspace_bytes(long long P, int Cin, int Cout, int nseg);
int tpg_mlp_wgrad(const void *x_out, const void *g_out, const uint8_t *arg, int K, const float *cb_out,
                  const void *x_in, const float *ci_in, float slope_in, long long P, int Cin,
                  int Cout, int nseg, int mode, float *dW, void *ws, void *stream);
int tpg_mlp_bn_bwd_apply(const void *g, const void *x, const float *ci, const float *c12, long long P, int C,
                         int nseg, void *dx, void *stream);
int tpg_mlp_bn_bwd_apply_rowsum(const void *g, const void *x, const float *ci, const float *c12, long long P, int K,
                                int C, int nseg, void *dx, float *qneg, void *stream);

/* ---- fused spectral normalisation of a (R x Cn) conv / linear weight ------------------------
 * torch.nn.utils.spectral_norm's forward pre-hook (n_power_iterations = 1) on every conv and
 * linear of the discriminators (discriminator.py:66-68,246-247,351-359,...) in ONE launch:
 *   iterate != 0 (training):  v <- normalize(W^T u); u <- normalize(W v)   (u, v updated in place)
 *   sigma = u . (W v);  Wsn = W / sigma.      normalize(x) = x / max(|x|_2, eps)
 * backward, u and v constants:  dW = (G - <G, Wsn> u v^T) / sigma. */
int tpg_spectral_norm_fwd(const float *W, float *u, float *v, int R, int Cn, int iterate, float eps,
                          float *Wsn, float *sigma, void *stream);
int tpg_spectral_norm_bwd(const float *G, const float *Wsn, const float *u, const float *v,
                          const float *sigma, int R, int Cn, float *dW, void *stream);

/* Batched form: every spectrally-normalised weight of one discriminator forward in ONE launch.
 * desc: device array of M records {const float* W; float* u; float* v; int64 R, Cn, uses,
 * out_off} (7 x 8 bytes).  Weight m is used `uses` times in the forward (once per frame / per
 * flow-embedding pair); each use advances (u, v) by one power iteration (iterate != 0) and gets
 * its own W / sigma.  For use t the kernel writes at out + out_off + t * stride(R, Cn) floats:
 *   W/sigma_t (R*Cn) | u_t (R) | v_t (Cn) | sigma_t (1),  stride = tpg_spectral_norm_multi_stride.
 * max_rc = max over m of (R + Cn).  u, v are left at their final values.
 * Backward: desc of M records {int64 R, Cn, uses, out_off, g_off, dw_off} (6 x 8 bytes); the
 * gradient of use t of weight m is at g + g_off + t*R*Cn, and
 * dw + dw_off <- sum_t (G_t - <G_t, Wsn_t> u_t v_t^T) / sigma_t.  Both descriptor arrays hold
 * only sizes, offsets and the persistent parameter pointers, so they can be built once. */
long long tpg_spectral_norm_multi_stride(int R, int Cn);
int tpg_spectral_norm_multi_fwd(const void *desc, int M, int max_rc, float *out, int iterate, float eps,
                                void *stream);
/* The same forward in TRAINING mode with every weight's rows split over ceil(R / tpg_spectral_norm_split_rows())
 * workgroups that keep their rows in registers for all uses and exchange partial column sums once per use (round 3:
 * the one-workgroup form is the longest launch at the head of the temporal discriminator's update).  Same output layout.
 * desc: per weight 9 x int64 {W, u, v, R, Cn, uses, out_off, parts, x_off}; x_off = offset, in 8-byte words, of the
 * weight's 2 * parts * (Cn + 1) exchange words inside xws.  part_map: total_parts x int32[2] = (weight, part).
 * xws: xws_words 8-byte words, 16-byte aligned (the exchange words of all weights + one trailing word that is non-zero
 * after the launch if a bounded spin gave up); zeroed by the call itself.  Cn <= tpg_spectral_norm_split_max_cn(). */
int tpg_spectral_norm_multi_fwd_split(const void *desc, const void *part_map, int total_parts, int max_cn, float *out,
                                      void *xws, long long xws_words, float eps, void *stream);
int tpg_spectral_norm_split_rows(void);
int tpg_spectral_norm_split_max_cn(void);
int tpg_spectral_norm_split_max_rows(void);      /* R <= this (at most 16 workgroups per weight) */
/* backward: max_uses = max over m of uses (<= 64); scratch: tpg_spectral_norm_multi_bwd_scratch(M, max_uses)
 * floats (the partial inner products <G_t, Wsn_t> of the row chunks, summed in fixed order). */
long long tpg_spectral_norm_multi_bwd_scratch(int M, int max_uses);
int tpg_spectral_norm_multi_bwd(const void *desc, int M, int max_uses, const float *g, const float *out,
                                float *dw, float *scratch, void *stream);

/* ---- fused radius search + bicubic weighted average (the `--use_vel` advection features) ----
 * gcn_lib/interpolation.py:107-123 `cubic_interpolation(query_pos, field, pos, cutoff)` over
 * get_local_neighbor_graph (:16-75; FRNN K=32, unique, FRNN, kNN-4 padding, DGL scatter-sums),
 * called per frame and per sample by train_step_final.py:51-66.  query (B,Nq,3), pos (B,Np,3),
 * field (B,Np,F).  Per query: the <= 32 nearest field points with d^2 < cutoff^2 (ascending
 * (d^2, idx)), w = bicubic(d / cutoff) * 8 / (pi cutoff^3);
 *   out_plain = sum w f / (sum w + 1e-6)
 *   out_pad   = the same with the 4 nearest hits counted twice (the reference's padding edges)
 *   hits      = number of in-range neighbours found (<= 32)
 * The caller picks out_pad for the queries with hits < 32 of every cloud in which SOME query has
 * hits == 0 (that is when the reference adds its padding edges), out_plain otherwise. */
int tpg_cubic_interp_f32(const float *query, const float *pos, const float *field, int B, int Nq, int Np,
                         int F, float cutoff, float *out_plain, float *out_pad, int32_t *hits,
                         void *stream);

/* The EdgeConv MLP tail at the generator's small channel counts (gcn_lib/pointnet/gcn.py:207-211 inside
 * IDGCNLayer, gcn.py:229-231): out[n] = max_j lrelu_s2(W2 . lrelu_s1(W1 . h[n*K + j])), h (P*K, H) rows of edge
 * features, W1 (C1, H), W2 (C2, C1) fp32, no bias; (H, C1, C2) = (16, 16, 32) only (TPG_ERR_UNSUPPORTED
 * otherwise), 1 <= K <= 255, 0 <= slopes <= 1.  h / out / gout / gh are bf16 (is_bf16 = 1) or fp32; arithmetic
 * is fp32 on the vector ALUs (weights as scalar operands).  Forward: out (P, C2) and the arg-max byte of each
 * (point, channel) (first maximum).  Backward (recomputes the hidden layer from h): gh (P*K, H), dW1, dW2
 * (fp32, per-wave slabs in `ws` summed in a fixed order). */
size_t tpg_small_tail_workspace_bytes(long long P, int K);
int tpg_small_tail_fwd(const void *h, int is_bf16, const float *W1, const float *W2, float slope1, float slope2,
                       long long P, int K, int H, int C1, int C2, void *out, unsigned char *arg, void *stream);
int tpg_small_tail_bwd(const void *h, const void *out, const void *gout, const unsigned char *arg, int is_bf16,
                       const float *W1, const float *W2, float slope1, float slope2, long long P, int K, int H,
                       int C1, int C2, void *gh, float *dW1, float *dW2, void *ws, void *stream);

/* ---- row-wise linear layers of any small channel count (csrc/rowlinear.hip, round 3) -------------------------
 * The 1x1 convolutions that are NOT inside a fused tail: the generator's node / edge affines, bottlenecks,
 * decoders and skip layers (gcn_lib/pointnet/gcn.py:176-180,207-211,253-277; upsampling_network.py:44-104), the
 * first layer of every shared MLP of the discriminators applied to the un-grouped points
 * (discriminator.py:63-78,140-148,276-282: Cin = 3, 6, 131, 259, 515) and the heads' linears
 * (discriminator.py:503-516,598-612) -- in the reference: cuDNN / cuBLAS calls plus separate bias / activation kernels.
 *     y[p, o] = lrelu_slope( sum_c x[p, c] * W[seg(p)][o, c] + bias[o] )
 * x (P, Cin) and y (P, Cout) channels-last rows, f32 or bf16 (tpg_dtype); W (nseg, Cout, Cin) f32: nseg equal
 * consecutive row blocks with their own weights (P % nseg == 0, and (P / nseg) % 128 == 0 when nseg > 1;
 * Cin, Cout <= 1000: tpg_rowlinear_supported);
 * bias (Cout) f32 or NULL; slope in [0, 1], 1 = no activation.  fp32 products and accumulation on
 * v_mfma_f32_16x16x4_f32.  x, y, W 16-byte aligned.
 * Backward (y = the forward's output, needed when slope != 1: the activation's derivative is taken from its sign):
 *   tpg_rowlinear_dgrad: dx (P, Cin) = (gy * lrelu'(y)) . W
 *   tpg_rowlinear_wgrad: dW (nseg, Cout, Cin) f32 = (gy * lrelu'(y))^T . x per segment; db (Cout) f32 or NULL = its
 *                        column sums over all segments; row slabs summed in slab order (bitwise reproducible);
 *                        ws: tpg_rowlinear_wgrad_workspace_bytes(P, nseg, Cin, Cout, db != NULL) bytes. */
int tpg_rowlinear_supported(int Cin, int Cout, int has_bias);
size_t tpg_rowlinear_wgrad_workspace_bytes(long long P, int nseg, int Cin, int Cout, int has_bias);
int tpg_rowlinear_fwd(const void *x, int dtype_in, const float *W, const float *bias, long long P, int nseg, int Cin,
                      int Cout, float slope, void *y, int dtype_out, void *stream);
int tpg_rowlinear_dgrad(const void *gy, const void *y, int dtype_g, const float *W, long long P, int nseg, int Cin,
                        int Cout, float slope, void *dx, int dtype_x, void *stream);
int tpg_rowlinear_wgrad(const void *x, int dtype_x, const void *gy, const void *y, int dtype_g, long long P, int nseg,
                        int Cin, int Cout, float slope, float *dW, float *db, void *ws, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TPGAN_OPS_H */
