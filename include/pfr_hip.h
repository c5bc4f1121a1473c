/* pfr_hip.h — C ABI of libpfr_hip.so: the MI355X (gfx950) kernels behind the feature-extractor hot path of
 * MarQuisCheshire/Pets-Face-Recognition (train step of the CNN backbone into the ArcFace head, and the
 * embedding cosine match / candR@K).
 *
 * The reference is 100 % Python on PyTorch and has NO FFI boundary of its own; every entry
 * point below therefore names the PyTorch call on the reference's path that it replaces (file:line relative
 * to the reference tree).  The reference-side binding a maintainer would add is the ctypes stub shown in
 * INTEGRATION.md (our own host code in pets-face-recognition_amd/_hip/lib.py is exactly that stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (the library never allocates, frees or retains);
 *   - tensors are NHWC ("channels last"), dense, channel count a multiple of 16 bytes / sizeof(element);
 *   - dtype: PFR_F32 = 0 (exact-f32 MFMA, the parity path), PFR_BF16 = 1 (bf16 in, f32 accumulate);
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); all calls are asynchronous;
 *   - workspaces are supplied by the caller; their sizes come from the *_splits / *_blocks / *_mtile queries;
 *   - return value: 0 = ok, <0 = error (PFR_ERR_*), message via pfr_last_error() (thread-local);
 *   - no global mutable state; thread-safe per stream.
 */
#ifndef PFR_HIP_H
#define PFR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFR_F32 0
#define PFR_BF16 1
#define PFR_OK 0
#define PFR_ERR_ARG (-1)
#define PFR_ERR_HIP (-2)
#define PFR_ERR_UNSUPPORTED (-3)

typedef void* pfr_stream_t; /* hipStream_t */

const char* pfr_last_error(void);
int pfr_version(void);
int pfr_device_arch(char* buf, int buflen);
/* Run-time tuning knobs — ONE table (csrc/pfr_api.hip); the library itself reads NO environment variable.  For A/B sweeps inside one
 * process and for pinning a kernel choice.  Keys (default):
 *   "igemm_p" (1) persistent GEMM kernel 0 never / 1 heuristic / 2 whenever eligible;  "igemm_ptile" (-1) its tile 0:128x128 1:64x128
 *   2:128x64 3:64x64 (rows x couts);  "igemm_pkch" (8) its k-step in 16-byte chunks;  "igemm_ppf" (0) its prefetch variants;
 *   "igemm_tile" (-1) / "igemm_kch" (0) / "igemm_big" (1): tile, k-step and 8-wave tiles of the one-tile-per-workgroup kernel;
 *   "sconv" (1) streaming 1x1 kernel 0 / 1 heuristic / 2 whenever eligible;  "sconv3" (1) halo-staged 3x3 64->64 kernel;
 *   "bnb" (0) BatchNorm-backward sums in the data-gradient epilogue: 1 tile kernels, 2 streaming kernels (the engines set 2);
 *   "swgrad" (1), "wgrad_big" (0), "wgrad_tile" (-1), "wgrad_splits" (0), "wgrad9" (1: the 56x56 class, 2: every geometry), "wgrad9_slots" (256 workgroups per launch): weight gradients;
 *   "bnb_tile3" (0) with bnb = 2: the 3x3 / stride-1 data gradients on the 256-row tile kernel leave the BatchNorm-backward sums too;
 *   "slin" (1) streaming Linear kernel for K = 96 j (Swin): 0 off, 1 for M >= 65536 rows, 2 whenever eligible;
 *   "attn_mfma" (1) window attention on MFMA;  "match_order" (1) L2-blocked tile order of the persistent filter GEMM of the gallery match.
 *   "ln_rb" (4) LayerNorm forward, rows in flight per lane group for C <= 512: 4, 6 or 8.
 * Results do not depend on the knobs (same accumulation order per kernel family; alternatives are pinned bit-for-bit or to the oracle by the
 * tests); statistics-partial granularity follows pfr_conv2d_mtile.  pfr_get_tuning reads a knob back. */
int pfr_set_tuning(const char* key, int value);
int pfr_get_tuning(const char* key, int* value);
/* bumped by every pfr_set_tuning call that CHANGES a knob: launch plans that baked a kernel choice in (statistics-partial granularity,
 * partial-row counts) are rebuilt by their owners when the epoch they were built under is over */
int pfr_tuning_epoch(void);

/* C-side executor of a pre-built launch plan (csrc/pfr_plan.hip): the host keeps a step's fixed list of C-ABI calls (fixed device
 * pointers) in a plan and replays it with ONE call instead of one interpreter round trip per launch.  An entry is appended with
 * the index of the entry point's thunk (pfr_plan_thunk_index("pfr_conv2d_fwd"), -1: not plannable) and its arguments as flat
 * 64-bit slots in declaration order WITHOUT the trailing stream (pointers / integers as such, floats as IEEE-754 bits).
 * kind: 0 launch on main | 1 launch on side | 2 fork (record event ev on main, side waits) | 3 record ev on side | 4 main waits
 * for ev | 5 as 4 but only when hook_stops == 1 (2: the hook synchronises with the side stream itself) | 6 hook stop (ev = user tag).  pfr_plan_run(begin, end <0 = all) returns -1 at the
 * end, the index of a kind-6 entry when hook_stops and it reached one (resume at index + 1), <= -2 on error. */
int pfr_plan_thunk_index(const char* name);
void* pfr_plan_create(int n_events);
int pfr_plan_destroy(void* plan);
int pfr_plan_append(void* plan, int kind, int thunk, int ev, const unsigned long long* args, int nargs);
int pfr_plan_size(void* plan);
int pfr_plan_run(void* plan, int begin, int end, pfr_stream_t main_stream, pfr_stream_t side_stream, int hook_stops);

/* ---- convolution / linear (implicit GEMM on MFMA) -------------------------------------------------------
 * pfr_conv2d_fwd replaces nn.Conv2d.forward / nn.Linear.forward / F.linear of the backbone and head
 * (torchvision resnet50 built at configs/dog_fe/fe_dogs_config.py:102-103; F.linear at
 * losses/large_margin.py:71) and — run over dy with flipped/transposed weights and idil_log2 = log2(stride) —
 * the autograd input gradient of the same ops.
 *   x [N][H][W][C], w [Cout][R][S][C], y [N*OH*OW][ldy] (ldy<=0 → Cout); ih = oh*stride - pad + r.
 *   bias [Cout] fp32 or NULL; residual [M][ldy] (y's dtype) or NULL: y = result + bias + residual;
 *   accumulate: y += result; out_relu: y = max(y,0)
 *   pro_scale/pro_shift [C] fp32 or NULL: operand is relu?(scale[c]*x + shift[c]) (fused BN-apply of the producer)
 *   stats_part or NULL: fp32 [ceil(M/mtile)][2][Cout] per-channel (mean, M2 = Σ(y-mean)²) of row groups of the stored y,
 *   mtile = pfr_conv2d_mtile(<the launch's geometry>, pro_scale != 0)  (input of pfr_bn_finalize; deterministic, no atomics):
 *   the m-tile height of the tile kernel that takes this geometry, half of it for the persistent kernel (one partial per wave row),
 *   or the rows of one workgroup's share for the streaming kernels (pfr_sconv.hip: a row range; pfr_sconv3.hip: 4x8-pixel patches
 *   in patch order) — pfr_bn_finalize only needs every group's row count, which is min(mtile, M - t*mtile) in all cases. */
int pfr_conv2d_mtile(int N, int H, int W, int C, int Cout, int R, int S, int stride, int pad, int OH, int OW, int dtype,
                     int out_dtype, int fused_prologue);
/* the same query for pfr_gemm_act_colstats (a launch with an activation epilogue always takes the tile kernel) */
int pfr_gemm_act_mtile(long M, int K, int N, int dtype);
int pfr_conv2d_fwd(const void* x, const void* w, void* y, int dtype, int out_dtype, int N, int H, int W, int C,
                   int Cout, int R, int S, int stride, int pad, int idil_log2, int OH, int OW, int ldy,
                   const float* bias, const void* residual, int accumulate, int out_relu, const float* pro_scale,
                   const float* pro_shift, int pro_relu, float* stats_part, pfr_stream_t stream);
/* data gradient of a block's first conv joined with the residual-branch gradient (autograd of torchvision
 * Bottleneck/BasicBlock `out += identity; out = relu(out)`): dx = dgrad(dy) + (mask ? res : 0).  dy [N][H][W][C], wt the
 * pfr_weight_dgrad_layout weights, dx / res [N][OH][OW][Cout]; res = gradient w.r.t. the block OUTPUT, res_mask = the ReLU
 * bit mask pfr_bn_act_mask wrote in forward ([N*OH*OW][Cout / (8 bf16 | 4 f32)] bytes); pad / idil_log2 as for the plain
 * data gradient through pfr_conv2d_fwd */
int pfr_conv2d_dgrad_join(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int C, int Cout, int R,
                          int S, int pad, int idil_log2, int OH, int OW, const void* res, const unsigned char* res_mask,
                          pfr_stream_t stream);

/* Data gradient that ALSO produces the BatchNorm-backward partial sums of the BN layer(s) whose OUTPUT gradient dx is (replaces
 * pfr_bn_bwd_reduce's pass over the gradient and the BN input; reference: autograd of nn.BatchNorm2d + ReLU after the conv,
 * torchvision resnet Bottleneck.forward).  part[t][0][c] = sum g*mask, part[t][1][c] = sum g*mask*xhat over the rows of m-tile t,
 * g = the value stored to dx, xhat = (x - mean)*invstd; mask = bn_mask bits ([M][Cout/KPACK] bytes of pfr_bn_act_mask) when given,
 * else scale*x + shift > 0.  bn_coef = [4][Cout] (mean, invstd, scale, shift rows, as pfr_bn_finalize writes them); bn2_* = an
 * optional second BN consuming the same gradient through the same bit mask (projection shortcut).  res/res_mask = the residual
 * join of pfr_conv2d_dgrad_join (both or neither), accumulate adds into dx.  pfr_conv2d_dgrad_bn_parts -> number of partial
 * rows per BN for the geometry (feed pfr_bn_bwd_finalize with it), or 0 when the fused form does not apply (run
 * pfr_bn_bwd_reduce instead).  With pfr_set_tuning("bnb", 2) (PFR_FUSE_BNB=2) the fused form is provided by the streaming kernels
 * only (1x1 / stride-1 geometries pfr_sconv.hip takes; parts = its row ranges): one BN, or two with the join; no accumulate, but `res`
 * without `res_mask` is a plain add and may alias dx (a gradient already accumulated there). */
int pfr_conv2d_dgrad_bn_parts(int dtype, int N, int H, int W, int C, int Cout, int R, int S, int idil_log2, int OH, int OW);
int pfr_conv2d_dgrad_bn(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int C, int Cout, int R, int S,
                        int pad, int idil_log2, int OH, int OW, const void* res, const unsigned char* res_mask, int accumulate,
                        const void* bn_x, const float* bn_coef, const unsigned char* bn_mask, float* bn_part, const void* bn2_x,
                        const float* bn2_coef, float* bn2_part, pfr_stream_t stream);

/* Main-branch data gradient of a block with a 1x1 / stride-2 projection shortcut: dx = dgrad(dy) + up2(res_compact) and the
 * BatchNorm-backward sums of the BN whose output gradient dx is.  res_compact [N][OH/2][OW/2][Cout] = the shortcut's gradient,
 * computed densely on its own grid (a plain pfr_conv2d_fwd over its dy with the pfr_weight_dgrad_layout weights), added at the
 * pixels with even (oh, ow).  Streaming form only (pfr_set_tuning("bnb", 2), pfr_conv2d_dgrad_bn_parts > 0, even OH / OW). */
int pfr_conv2d_dgrad_bn_sub(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int C, int Cout, int OH, int OW,
                            const void* res_compact, const void* bn_x, const float* bn_coef, const unsigned char* bn_mask,
                            float* bn_part, pfr_stream_t stream);

/* Recompute form of a bottleneck's last convolution (1x1, stride 1; torchvision Bottleneck.forward `out = relu(bn3(conv3(z)) + identity)`)
 * on the bf16 streaming kernel — the convolution output never reaches HBM: pfr_conv1x1_stats leaves only the BatchNorm partials (tile
 * height pfr_conv1x1_tail_mtile = pfr_conv2d_mtile of the geometry; 0: not taken, use pfr_conv2d_fwd + pfr_bn_act_mask), and after
 * pfr_bn_finalize pfr_conv1x1_bn_tail recomputes it and stores y = relu(a1*conv + b1 + (a2 ? a2*res + b2 : res)) with the ReLU bit mask
 * of pfr_bn_act_mask; bit-identical to the stored form (both round the convolution to bf16 first). */
int pfr_conv1x1_tail_mtile(int dtype, int N, int H, int W, int C, int Cout);
int pfr_conv1x1_stats(const void* x, const void* w, int dtype, int N, int H, int W, int C, int Cout, float* stats_part,
                      pfr_stream_t stream);
int pfr_conv1x1_bn_tail(const void* x, const void* w, void* y, unsigned char* mask, int dtype, int N, int H, int W, int C, int Cout,
                        const float* a1, const float* b1, const void* res, const float* a2, const float* b2, pfr_stream_t stream);

/* The two launches above with `flags` (streaming form with a bit mask only): 1 = dx is stored THROUGH bn_mask (dx = g*mask: every consumer
 * of a block-output gradient — the BN backward of the block's last BN, its projection BN, the residual join — reads it through that mask,
 * so they may then read it plainly); 2 = the first BN's input is not read: only sum g*mask is produced (row 1 of bn_part = 0; bn_x and
 * bn_coef may be NULL) — sum g*mask*xhat then comes out of pfr_bn3_bwd_coef. */
int pfr_conv2d_dgrad_bn_ex(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int C, int Cout, int R, int S,
                           int pad, int idil_log2, int OH, int OW, const void* res, const unsigned char* res_mask, int accumulate,
                           const void* bn_x, const float* bn_coef, const unsigned char* bn_mask, float* bn_part, const void* bn2_x,
                           const float* bn2_coef, float* bn2_part, int flags, pfr_stream_t stream);
int pfr_conv2d_dgrad_bn_sub_ex(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int C, int Cout, int OH, int OW,
                               const void* res_compact, const void* bn_x, const float* bn_coef, const unsigned char* bn_mask,
                               float* bn_part, int flags, pfr_stream_t stream);

/* Backward of a bottleneck's last 1x1 convolution + BatchNorm (autograd of `relu(bn3(conv3(z)) + shortcut)`, torchvision
 * Bottleneck.forward) WITHOUT the BatchNorm's input x = conv3(z) and without its gradient dx (csrc/pfr_bnfree.hip): BN backward is
 * linear, dx = A*G + B*(x - mean) + C0 per channel, and x - mean = (Z - zbar) W^T, so with G1 = G^T Z (pfr_conv2d_wgrad of the masked
 * gradient G [M][C] against Z [M][K]), G2 = Z^T Z, zsum = column sums of Z:
 *   pfr_bn3_bwd_coef:    dbeta = sum of part[t][0][:] (the partials pfr_conv2d_dgrad_bn[_ex] left), dgamma_c = invstd_c * sum_k W[c][k]
 *                        (G1[c][k] - dbeta_c zbar_k); coef [3][C] = A = gamma invstd, B = -gamma invstd^2 dgamma / M, C0 = -gamma invstd dbeta / M
 *   pfr_bn3_bwd_weights: dW = A*G1 + B*(W (G2 - M zbar zbar^T)) + C0 (x) (M zbar)   (fp32, += if accumulate);
 *                        wa_t [K][C] bf16 = A_c W[c][k] (data-gradient weight layout), S [K][K] bf16 = W^T diag(B) W, bias [K] = C0^T W - zbar S
 * and the data gradient is dZ = G wa_t^T + Z S + bias: pfr_conv2d_fwd(Z, S, bias) then pfr_conv2d_dgrad_bn(G, wa_t, res = that, no masks),
 * or — S == NULL: wa_t is then ONE concatenated tensor wcat [K][C + K], row k = [A*W column k | S row k] — a single
 * pfr_conv1x1_dgrad2_bn(G, Z, wcat, bias) launch over both row sources (bias added in fp32 inside the accumulators; BatchNorm-backward
 * sums of the BN dZ feeds as pfr_conv2d_dgrad_bn with a recomputed ReLU mask; pfr_conv1x1_dgrad2_bn_parts = partial rows, 0 = not taken).
 * W [C][K] (wdtype PFR_BF16 / PFR_F32) = the weights the FORWARD convolution multiplied with (the bf16 shadow on the bf16 path): every term
 * that reconstructs x = Z W^T must use the x that was actually normalised; count = M rows. */
int pfr_bn3_bwd_coef(const float* part, int nparts, const float* G1, const float* zsum, const void* W, int wdtype, const float* gamma,
                     const float* invstd, int C, int K, float count, float* dgamma, float* dbeta, float* coef, int accumulate,
                     pfr_stream_t stream);
int pfr_bn3_bwd_weights(const float* coef, const float* G1, const float* G2, const float* zsum, const void* W, int wdtype, int C, int K, float count,
                        float* dW, void* wa_t, void* S, float* bias, int accumulate, pfr_stream_t stream);
int pfr_conv1x1_dgrad2_bn_parts(int dtype, int N, int H, int W, int C1, int C2, int Cout);
int pfr_conv1x1_dgrad2_bn(const void* g, const void* z, const void* wcat, const float* bias, void* dx, int dtype, int N, int H, int W,
                          int C1, int C2, int Cout, const void* bn_x, const float* bn_coef, float* bn_part, pfr_stream_t stream);

/* G2 = X^T X [Q][Q] and the column sums of X [Q] in ONE streaming pass over X [M][Q] (bf16, Q = 64 | 128): the two forward-only inputs
 * of pfr_bn3_bwd_coef / pfr_bn3_bwd_weights.  out = Q*Q floats then Q floats; workspace = pfr_gram_ws_floats(M, Q) floats (0: geometry
 * not supported — pfr_conv2d_wgrad(x, x) + pfr_colsum give the same). */
/* batch statistics of x = Z W^T from pfr_gram_colsum's output for Z (no pass over x): mean_c = W[c] . zbar, var_c = W[c] (G2/M - zbar zbar^T)
 * W[c]^T; part = ONE (mean, M2) partial row [2][C] for pfr_bn_finalize(nparts = 1, rows_per_part = M).  W: the bf16 weights [C][K].
 * The subtraction cancels when a channel is nearly constant (|mean| >> std): the kernel bounds its own rounding error and, where that exceeds
 * 1 % of var + eps, stores 1 to *cancel_flag (may be NULL; host-visible memory lets the caller fall back to a statistics pass over x). */
int pfr_bn_stats_from_gram(const float* gram, const void* W, int dtype, int C, int K, float count, float* part, float eps, int* cancel_flag,
                           pfr_stream_t stream);
/* the same followed by pfr_bn_finalize(nparts = 1, rows_per_part = count) in one launch (results agree to the last bit or two; replaces torch.nn.BatchNorm2d's
 * training-mode statistics + running-stat update for a bottleneck's bn3: torchvision resnet.py Bottleneck.forward, third-party to /root/reference,
 * backbone built at configs/dog_fe/fe_dogs_config.py:102-103) */
int pfr_bn_finalize_from_gram(const float* gram, const void* W, int dtype, int C, int K, float count, const float* gamma, const float* beta,
                              float eps, float momentum, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                              float* shift, int* cancel_flag, pfr_stream_t stream);
long pfr_gram_ws_floats(long M, int Q);
int pfr_gram_colsum(const void* x, int dtype, long M, int Q, float* out, float* workspace, pfr_stream_t stream);

/* pfr_conv2d_wgrad replaces the autograd weight gradient of nn.Conv2d / nn.Linear / F.linear:
 *   dw[co][r][s][c] (fp32) = scale * sum_m dy[m][co] * act(x)[...]  (+ dw if accumulate)
 * workspace: fp32 [pfr_conv2d_wgrad_splits(M,Cout,R*S*C)][Cout][R*S*C] (may be NULL when splits == 1). */
int pfr_conv2d_wgrad_splits(int M, int Cout, int KK);
int pfr_conv2d_wgrad(const void* x, const void* dy, float* dw, float* workspace, int dtype, int N, int H, int W, int C,
                     int Cout, int R, int S, int stride, int pad, int OH, int OW, int lddy, const float* pro_scale,
                     const float* pro_shift, int pro_relu, float scale, int accumulate, pfr_stream_t stream);

/* ---- layout / dtype helpers -------------------------------------------------------------------------- */
/* batch['x'] fp32 NCHW (data_loading/dataset.py:125) → NHWC compute dtype with Cp >= C zero-padded channels */
int pfr_nchw_to_nhwc(const float* x, void* y, int dtype, int N, int C, int H, int W, int Cp, pfr_stream_t stream);
int pfr_cast(const void* x, int src_dtype, void* y, int dst_dtype, size_t n, pfr_stream_t stream);
/* w [O][R][S][I] → wt [I][R][S][O], taps flipped: the weights pfr_conv2d_fwd needs to compute the data gradient */
int pfr_weight_dgrad_layout(const void* w, void* wt, int dtype, int O, int R, int S, int I, pfr_stream_t stream);
/* the same for n_layers convs in one launch; descs: DEVICE array of {const void* w; void* wt; int O, R, S, I;} (24 bytes each) */
int pfr_weight_dgrad_layout_batch(const void* descs, int n_layers, int dtype, pfr_stream_t stream);
int pfr_add(const void* a, const void* b, void* y, int dtype, size_t n, pfr_stream_t stream);
long pfr_colsum_ws_floats(long rows, int C); /* 0 for small row counts */
int pfr_colsum(const void* x, int dtype, long rows, int C, float* out, int accumulate, float* workspace, pfr_stream_t stream);
/* the same with the final merge deferred: pfr_colsum_partial leaves row-block partials in `workspace` (pfr_colsum_ws_floats floats,
 * one workspace per pending tensor); pfr_colsum_parts = how many partial rows that will be (0: too few rows, use pfr_colsum);
 * pfr_colsum_final_batch merges n partial sets in one launch.  descs: DEVICE array of {const float* part; float* out; int n; int C;
 * int accumulate; int mt; int rows; int pad} (40 bytes each); mt > 0: `part` holds the [n][2][C] m-tile statistics a GEMM epilogue
 * left (pfr_gemm_act_colstats, pfr_conv2d_fwd stats_part) for tiles of height mt over `rows` rows, and out = sum_t rows_t * mean_t */
int pfr_colsum_parts(int dtype, long rows, int C);
int pfr_colsum_partial(const void* x, int dtype, long rows, int C, float* workspace, pfr_stream_t stream);
int pfr_colsum_final_batch(const void* descs, int n, int max_C, pfr_stream_t stream);
int pfr_copy2d_f32(const float* src, int ld_src, float* dst, int ld_dst, long rows, int cols, float scale, int accumulate,
                   pfr_stream_t stream);

/* ---- BatchNorm2d (train: batch statistics, eps, momentum, unbiased running var — torchvision resnet) ---- */
int pfr_colreduce_blocks(int C, int dtype, long rows); /* partial rows written by pfr_bn_stats / pfr_bn_bwd_reduce */
int pfr_bn_stats(const void* x, int dtype, long rows, int C, float* part, pfr_stream_t stream);
long pfr_bn_stats_rows_per_part(int C, int dtype, long rows);
/* part: [nparts][2][C] = (mean_t, M2_t) of rows [t*rows_per_part, min(count, (t+1)*rows_per_part)) — written by
 * pfr_conv2d_fwd (rows_per_part = pfr_conv2d_mtile) or pfr_bn_stats (rows_per_part = pfr_bn_stats_rows_per_part). */
long pfr_bn_finalize_ws_floats(int nparts, int C); /* scratch floats for a parallel two-level merge (0: not needed) */
int pfr_bn_finalize(const float* part, int nparts, long rows_per_part, int C, float count, const float* gamma,
                    const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                    float* invstd, float* scale, float* shift, float* workspace, pfr_stream_t stream);
/* Inference embedder (Controller.validation_step / test_step, reference engine/controller.py:31-46, and the per-photo loop
 * of generate_tsv.py:233-251): eval-mode BatchNorm folded into the producing convolution, all layers in ONE launch.
 * descs: device array of ndesc records { const void* src; const float* gamma, *beta, *running_mean, *running_var;
 * void* wout; float* bout; long cout, k; float eps; int src_is_f32; } (80 bytes, natural alignment):
 * wout[co][k] = src[co][k] * gamma/sqrt(var+eps) in `dtype`, bout[co] = beta - mean*gamma/sqrt(var+eps) */
int pfr_fold_bn(const void* descs, int ndesc, int dtype, pfr_stream_t stream);
/* pfr_fold_bn that runs only when a parameter changed since the folded weights were made: a 64-bit position-weighted checksum of the
 * fp32 master buffer [n_master] and of every record's running statistics is taken on the device and compared with the previous
 * one; the master -> compute-dtype cast into `shadow` (may be NULL) and the fold exit at once when they agree.  state: 4 x 64-bit
 * device words owned by the caller, {0, ~0, 0, 0} before the first call; state[2] counts folds done, state[3] calls. */
int pfr_fold_bn_cached(const void* descs, int ndesc, int dtype, const float* master, size_t n_master, void* shadow,
                       unsigned long long* state, pfr_stream_t stream);
int pfr_bn_eval_coeff(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, float* scale, float* shift, pfr_stream_t stream);
/* y = relu?( a1*x1 + b1 (+ a2*x2 + b2 | + x2) ): BN apply, ReLU and the residual add of a bottleneck in one pass */
int pfr_bn_act(const void* x1, const float* a1, const float* b1, const void* x2, const float* a2, const float* b2, void* y,
               int dtype, long rows, int C, int relu, pfr_stream_t stream);
/* same, and (mask != NULL) also writes the sign of the pre-ReLU value as a bit mask: one byte per (row, 16-byte channel
 * chunk) = [rows][C / (8 bf16 | 4 f32)], bit e = (value of channel chunk*KP + e) > 0; the backward pass of a block's last
 * BN (relu(bn(x) + shortcut), torchvision Bottleneck.forward) reads this instead of the activation tensor */
int pfr_bn_act_mask(const void* x1, const float* a1, const float* b1, const void* x2, const float* a2, const float* b2,
                    void* y, unsigned char* mask, int dtype, long rows, int C, int relu, pfr_stream_t stream);
/* backward: g = dout * mask (mask_mode 0 none, 1: out > 0, 2: scale*x+shift > 0, 3: `out` is the bit mask of pfr_bn_act_mask);
 * reduce → partials of (Σg, Σg·x̂); finalize → dgamma, dbeta, coef[3][C]; apply → dx = coef0*g + coef1*x + coef2, gres = g */
int pfr_bn_bwd_reduce(const void* dout, const void* out, const void* x, const float* mean, const float* invstd,
                      const float* scale, const float* shift, int mask_mode, int dtype, long rows, int C, float* part,
                      pfr_stream_t stream);
int pfr_bn_bwd_finalize(const float* part, int nparts, int C, float count, const float* gamma, const float* mean,
                        const float* invstd, float* dgamma, float* dbeta, float* coef, int accumulate, pfr_stream_t stream);
int pfr_bn_bwd_apply(const void* dout, const void* out, const void* x, const float* coef, const float* scale,
                     const float* shift, int mask_mode, void* dx, void* gres, int dtype, long rows, int C,
                     pfr_stream_t stream);

/* ---- pooling (nn.MaxPool2d(3,2,1) fused with the stem's BN+ReLU; nn.AdaptiveAvgPool2d(1)) --------------- */
int pfr_bn_relu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y, uint8_t* idx, int dtype, int N,
                            int H, int W, int C, int relu, pfr_stream_t stream);
int pfr_maxpool_bwd(const void* dy, const uint8_t* idx, void* dz, int dtype, int N, int H, int W, int C, pfr_stream_t stream);
int pfr_avgpool_fwd(const void* x, void* y, int dtype, int N, int HW, int C, pfr_stream_t stream);
int pfr_avgpool_bwd(const void* dy, void* dx, int dtype, int N, int HW, int C, pfr_stream_t stream);

/* ---- ArcFace / CosFace head + (focal) cross-entropy (losses/large_margin.py:30-40,69-84; losses/losses.py:22-28) */
/* Space-to-depth form of the ResNet stem `conv1 = Conv2d(3, 64, 7, stride 2, padding 3)` (torchvision resnet, built at
 * configs/dog_fe/fe_dogs_config.py:102): the same sums as a 4x4 stride-1 pad-2 convolution over the half-resolution image
 * xs [N][H/2][W/2][Cp], channel (p*2+q)*C + c = x[c][2i+p][2j+q] (H, W even; Cp >= 4*C, zero padded), with weights
 * ws [Cout][4][4][Cp], ws[a][b][(p*2+q)*C + c] = w[2a+p-1][2b+q-1][c] (zero outside the 7x7 kernel).  Run it through
 * pfr_conv2d_fwd / pfr_conv2d_wgrad with R = S = 4, stride 1, pad 2, OH = H/2, OW = W/2.
 *   pfr_s2d_input : x fp32 NCHW -> xs (`dtype`)          pfr_s2d_weight: w fp32 [Cout][7][7][C] -> ws (`dtype`)
 *   pfr_s2d_wgrad : dws fp32 [Cout][4][4][Cp] -> dw fp32 [Cout][7][7][C] (+= if accumulate) */
int pfr_s2d_input(const float* x, void* y, int dtype, int N, int C, int H, int W, int Cp, pfr_stream_t stream);
int pfr_s2d_weight(const float* w, void* ws, int dtype, int Cout, int C, int Cp, pfr_stream_t stream);
int pfr_s2d_wgrad(const float* dws, float* dw, int Cout, int C, int Cp, int accumulate, pfr_stream_t stream);
/* y [cols][rows] = transpose of x [rows][cols] (the head's data gradient runs as a split-K GEMM over the class dimension:
 * autograd of F.linear in losses/large_margin.py:71, see losses/_head_hip.py) */
int pfr_transpose2d(const void* x, void* y, int dtype, int rows, int cols, pfr_stream_t stream);
/* match preparation (F.normalize of query / gallery embeddings, engine/controller.py:77-90 via similarity_f): one pass
 * over fp32 rows writes the L2-normalised row in bf16 (GEMM operand) and / or fp32 (exact re-scoring operand) */
int pfr_l2norm_dual(const float* x, void* xn_bf16, float* xn_f32, float* inv_norm, int rows, int D, float eps,
                    pfr_stream_t stream);
int pfr_l2norm_fwd(const void* x, int in_dtype, void* xn, void* xnT, int out_dtype, float* inv_norm, int rows, int D,
                   int ldt, float eps, pfr_stream_t stream);
int pfr_l2norm_bwd(const void* x, int in_dtype, const float* inv_norm, const float* dxn, void* dx, int out_dtype, int rows,
                   int D, int accumulate, pfr_stream_t stream);
/* mode 0 ArcFace hard margin, 1 ArcFace easy margin, 2 CosFace, 3 plain scaled cosine.
 * logits [B][C] fp32 or NULL, loss_rows [B] fp32 or NULL, dcos [B][ldc] (dcos_dtype) or NULL = grad_scale * d loss_row / d cos */
int pfr_margin_ce(const float* cosv, const int64_t* label, int B, int C, int ldc, int mode, float s, float m, float gamma,
                  float grad_scale, const float* grad_scale_dev, float* logits, float* loss_rows, void* dcos, int dcos_dtype,
                  pfr_stream_t stream);
/* standalone margin backward: dcos = s * dlogits (* dphi/dcos on the target column) */
int pfr_margin_bwd(const float* cosv, const int64_t* label, int B, int C, int ldc, int mode, float s, float m,
                   const float* dlogits, void* dcos, int dcos_dtype, pfr_stream_t stream);
int pfr_mean(const float* x, float* out, int n, pfr_stream_t stream);

/* ---- optimiser steps over flat fp32 master buffers (configs/dog_fe/fe_dogs_config.py:123-133; body_dog_fe.py:121-131) */
int pfr_sgd_step(float* p, const float* g, float* mom, void* shadow, int shadow_dtype, size_t n, float lr, float momentum,
                 float weight_decay, float grad_scale, int first_step, pfr_stream_t stream);
int pfr_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, size_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, pfr_stream_t stream);

/* ---- Swin-T feature extractor (models/swin.py:8-241): LayerNorm (29,215), exact GELU (39-43), fused shifted-window
 * attention (101-135).  Linear layers and the Unfold+Linear patch merging (a stride-f conv) use pfr_conv2d_*. */
int pfr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int dtype,
                      long rows, int C, float eps, pfr_stream_t stream);
int pfr_layernorm_bwd_blocks(long rows); /* part is fp32 [2][blocks][C]: dgamma partial rows, then dbeta partial rows (sum each half with pfr_colsum) */
int pfr_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                      const void* dres, void* dx, float* part, int dtype, long rows, int C, pfr_stream_t stream);
/* the same pass that also leaves the per-workgroup column sums of dx (dxsum_part [pfr_layernorm_bwd_blocks(rows)][C], may be NULL):
 * torch autograd computes the bias gradient of the nn.Linear / patch-merging layer in front of the LayerNorm as a separate
 * reduction over that gradient (models/swin.py:157-190 blocks, 84-99 PatchMerging); pfr_layernorm_bwd_dxsum_ok(dtype, C) = 1 when
 * the channel count takes the kernel that can do it */
int pfr_layernorm_bwd_dxsum(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                            const void* dres, void* dx, float* part, float* dxsum_part, int dtype, long rows, int C,
                            pfr_stream_t stream);
int pfr_layernorm_bwd_dxsum_ok(int dtype, int C);
int pfr_gelu_fwd(const void* x, void* y, int dtype, size_t n, pfr_stream_t stream);
int pfr_gelu_bwd(const void* x, const void* dy, void* dx, int dtype, size_t n, pfr_stream_t stream);
/* bias(+mask) table of one attention block: tab fp32 [4][64][64] (-inf outside w*w x w*w) followed by a second copy of the same
 * values in the MFMA kernels' access order (pfr_window_bias_table_floats = both; always allocate that many floats and pass the
 * buffer whole to pfr_window_attn_fwd / _bwd); variant 2*(last window row)+(last window column); built from the (2w-1)x(2w-1) relative-position table `pos` (models/swin.py:65-70,93-95,117-118) and the shifted-window
 * masks (create_mask, models/swin.py:49-62,86-90,122-124); rebuild whenever pos changed */
long pfr_window_bias_table_floats(int window);
int pfr_window_bias_table(const float* pos, float* tab, int window, int shift, pfr_stream_t stream);
/* the tables of n attention blocks in one launch; descs: DEVICE array of {const float* pos; float* tab; int window; int shift;}
 * (24 bytes each; window * window <= 64 is the caller's to check) */
int pfr_window_bias_table_batch(const void* descs, int n, pfr_stream_t stream);
/* qkv [B][H][W][3*heads*head_dim] (q|k|v, each (head, d)); pos: the TABLE built by pfr_window_bias_table;
 * shift = cyclic displacement (0 or w/2); out [B][H][W][heads*head_dim] */
int pfr_window_attn_fwd(const void* qkv, const float* pos, void* out, int dtype, int B, int H, int W, int heads, int head_dim,
                        int window, int shift, float scale, pfr_stream_t stream);
/* dpos_part: fp32 [B*(H/w)*(W/w)*heads][(2w-1)^2] per-workgroup partials of the table gradient (sum with pfr_colsum) */
int pfr_window_attn_bwd(const void* qkv, const float* pos, const void* dout, void* dqkv, float* dpos_part, int dtype, int B,
                        int H, int W, int heads, int head_dim, int window, int shift, float scale, pfr_stream_t stream);
int pfr_nhwc_to_nchw_f32(const float* x, float* y, int N, int C, int HW, int Cp, int accumulate, pfr_stream_t stream);

/* ---- embedding match: running top-K over gallery chunks (engine/controller.py:77-90,143-160; generate_tsv.py:91-125;
 * similarity_f = (cos+1)/2 at configs/dog_fe/fe_dogs_config.py:89-93).  Scores of a chunk come from pfr_conv2d_fwd
 * (plain GEMM of L2-normalised rows, fp32 out).  Order: score descending, ties → lower gallery index. */
long pfr_topk_state_bytes(int rows, int K);
int pfr_topk_reset(void* state, int rows, int K, pfr_stream_t stream);
/* scores [rows][ld] fp32, n valid columns, gallery index of column j = col0 + j; chunks must come in increasing col0.
 * self_idx [rows] int32 or NULL: gallery index to skip per query (all-vs-all evaluation excludes the query itself) */
int pfr_topk_update(const float* scores, int rows, int ld, int n, int col0, int K, void* state, const int* self_idx,
                    pfr_stream_t stream);
/* fused form for every chunk after the first (all running lists full): the match GEMM q[Q][D] x g[n][D]^T with the top-K
 * filter in its epilogue — a score is appended to query r's candidate list cand[r][0..cap) (u64 key<<32 | ~index) iff it
 * beats r's current K-th best; the fp32 score matrix is never written.  pfr_topk_merge then folds the candidates into
 * the running lists.  exclude_self: skip gallery index == query row.  Overflow of a candidate list sets bit 1 of
 * pfr_topk_flags: the caller must redo the match with pfr_topk_update. */
int pfr_match_scores_filter(const void* q, const void* g, int dtype, int Q, int n, int D, int col0, int K, void* state,
                            void* cand, int cap, int exclude_self, pfr_stream_t stream);
int pfr_topk_merge(const void* cand, int cap, int rows, int K, void* state, pfr_stream_t stream);
int pfr_topk_flags(const void* state, int rows, int K, int* out_host, pfr_stream_t stream);
int pfr_topk_finish(const void* state, int rows, int K, float* out_scores, int* out_idx, pfr_stream_t stream);
/* exact fp32 re-scoring of KC candidates per query, keeps the best K.  q: L2-normalised fp32 rows; g: L2-normalised fp32 rows (g_scale
 * NULL), or the RAW gallery rows with g_scale[i] = 1 / max(|g_i|, eps) (pfr_l2norm_dual's inv_norm output): the score is <q, g_i> * g_scale[i]
 * and no normalised fp32 copy of the gallery has to exist (2 GB less written per 1 M x 512 match) */
int pfr_topk_rescore(const float* q, const float* g, const float* g_scale, int rows, int D, const int* cand, int KC, int K,
                     float* out_scores, int* out_idx, pfr_stream_t stream);
/* the same, and the certificate of the two-precision match (the reference scores every pair in fp32 — utils/calc_scores.py's torch.mm — so a
 * reduced-precision candidate selection has to show that it lost nothing): cand_scores [rows][KC] = the selection scores pfr_topk_finish
 * returned with cand; cert [rows][2] fp32: cert[r][0] = max |fp32 score - selection score| over r's candidates, cert[r][1] = (K-th best
 * fp32 score) - (selection score of r's last candidate), +inf when the list holds every eligible gallery row.  A gallery row outside the list
 * can belong to r's exact top-K only if its selection error exceeds cert[r][1]; the host compares that gap with the errors measured on all
 * candidates and re-matches the rows that fail (match.cosine_topk). */
int pfr_topk_rescore_cert(const float* q, const float* g, const float* g_scale, int rows, int D, const int* cand, const float* cand_scores,
                          int KC, int K, float* out_scores, int* out_idx, float* cert, pfr_stream_t stream);
/* out[p] = (cos(emb[idx_a[p]], emb[idx_b[p]]) + 1) / 2, norms clamped at eps (F.cosine_similarity) */
int pfr_pair_similarity(const float* emb, int D, const long* idx_a, const long* idx_b, int P, float eps, float* out,
                        pfr_stream_t stream);

/* sort / scan part of the verification metrics (engine/controller.py:112-183; torchmetrics ROC / AUROC / AveragePrecision /
 * StatScores over the pair scores): scores fp32 [P], labels int32 [P] (0 impostor, != 0 genuine) -> sorted_scores [P] descending
 * (ties: lower pair index first), cum_tp [P] = genuine pairs among the first i+1, run_end [P] = 1 where a run of equal scores ends
 * (the distinct-threshold operating points).  workspace: pfr_pair_curve_ws_bytes(P). */
long pfr_pair_curve_ws_bytes(int P);
int pfr_pair_curve(const float* scores, const int* labels, int P, void* workspace, float* sorted_scores, int* cum_tp,
                   unsigned char* run_end, pfr_stream_t stream);

/* mean-strategy card matching (generate_tsv.py:71-78,91-125): centroid of the L2-normalised photo embeddings of each card;
 * seg [ncards+1] int64 row offsets; cent32 fp32 [ncards][D] (always written), cent (cent_dtype) optional copy */
int pfr_card_centroids(const float* emb, const long* seg, int ncards, int D, float eps, float* cent32, void* cent,
                       int cent_dtype, pfr_stream_t stream);

/* head/body fusion of the inference ranking (generate_tsv.py:91-110), in place over `head_scores`.  head_scores,
 * body_scores: fp32 [rows][ld] centroid dot products of a query-card block against gallery cards col0..col0+n.
 * q_flags [rows], g_flags [all gallery cards] (indexed col0 + j): bit 0 = card has head vectors, bit 1 = has body
 * vectors, bits 2..7 = species `type` (1-based).  thresholds: HOST array of n_types floats (generate_tsv.py:107:
 * [0.9069641, 0.985643]).  Pairs the reference skips (type mismatch, both scores 0) become -inf. */
int pfr_card_fuse_scores(float* head_scores, const float* body_scores, int rows, int ld, int n, int col0,
                         const unsigned char* q_flags, const unsigned char* g_flags, const float* thresholds, int n_types,
                         pfr_stream_t stream);

/* ---- train-time augmentation on the device (configs/dog_fe/fe_dogs_config.py:17-26: RandomAdjustSharpness(0, 0.1),
 * RandomAutocontrast(0.3), RandomCrop((220, 220)), Resize((224, 224)), RandomRotation(5), ToTensor), bit-exact with the
 * Pillow arithmetic torchvision's PIL-image transforms run (oracle/augment_ref.py).
 * pfr_augment_params (HOST arrays in, HOST array out; no device work): flags int32 [N][4] = (apply sharpness, apply
 * autocontrast, crop top, crop left), angles float32 [N] degrees → records int32 [N][12] (the flags + Pillow's 16.16
 * fixed-point inverse rotation about the centre of the out_w x out_h image).  Upload `records` and pass it below.
 * pfr_augment_train: x uint8 [N][H][W][3] → y float32 [N][3][out_h][out_w] in [0, 1]; ws: pfr_augment_ws_bytes. */
int pfr_augment_params(const int* flags, const float* angles, int N, int out_w, int out_h, int* records);
long pfr_augment_ws_bytes(int N, int H, int W);
int pfr_augment_train(const unsigned char* x, int N, int H, int W, int crop_h, int crop_w, int out_h, int out_w,
                      const int* records, float* y, void* ws, pfr_stream_t stream);

/* Linear layer with a fused activation epilogue — the Swin MLP `FeedForward` (reference models/swin.py:40-52: Linear →
 * GELU → Linear) and its autograd.  x [M][K], w [N][K] (nn.Linear layout), y / y2 [M][N], all of `dtype`.
 *   act 2: y2 = x·wT + bias (pre-activation, kept for backward), y = gelu(y2)            (forward of the first Linear)
 *   act 3: y = (x·wT) * gelu'(y2)                         (data gradient of the second Linear joined with GELU backward) */
int pfr_gemm_act(const void* x, const void* w, void* y, int dtype, long M, int K, int N, const float* bias, int act,
                 void* y2, pfr_stream_t stream);
/* the same + per-m-tile column statistics of the stored y in stats_part [ceil(M / mtile)][2][N] (mtile = pfr_conv2d_mtile(M, N, K, K,
 * dtype, dtype, 0)): a bias gradient (column sum of y) without another pass over y (pfr_colsum_final_batch, mt > 0) */
int pfr_gemm_act_colstats(const void* x, const void* w, void* y, int dtype, long M, int K, int N, const float* bias, int act,
                          void* y2, float* stats_part, pfr_stream_t stream);
/* act 3 (y = (x wT) * gelu'(y2), no bias) + plain column SUMS of the stored y: sums_part [parts][N] fp32 with parts =
 * pfr_gemm_act_colsum_parts(M, K, N, dtype); every partial row holds the column sums over a disjoint set of rows (their total = the bias
 * gradient of the Linear layer in front: pfr_colsum_final_batch with mt = 0).  Provided by the streaming Linear kernel (csrc/pfr_slin.hip)
 * for its geometries only: parts == 0 -> pfr_gemm_act_colstats.  The answer depends on the "slin" knobs (pfr_tuning_epoch). */
int pfr_gemm_act_colsum_parts(long M, int K, int N, int dtype);
int pfr_gemm_act_colsums(const void* x, const void* w, void* y, int dtype, long M, int K, int N, void* y2, float* sums_part,
                         pfr_stream_t stream);

/* ---- gradient all-reduce over RCCL / xGMI (csrc/pfr_comm.hip) ---------------------------------------------
 * For hosts that bind this library directly; replaces DistributedDataParallel's bucket all-reduce (utils/__init__.py:114-119).
 * RCCL is resolved with dlopen at first use (no load-time dependency).  pfr_comm_unique_id: rank 0 fills a 128-byte id, the
 * host distributes it; pfr_comm_init: one communicator per process / GPU (current HIP device), NULL on error;
 * pfr_comm_allreduce: in place, `count` elements of dtype, mean over the ranks when average != 0, asynchronous and ORDERED on `stream`: the
 * collective itself runs on a high-priority stream the communicator owns (so that it does not share a hardware queue with the caller's compute
 * streams), tied to `stream` by an event on either side — work enqueued on `stream` before the call is complete before the collective reads
 * `buf`, work enqueued after it sees the reduced values. */
int pfr_comm_unique_id(void* id128);
void* pfr_comm_init(int rank, int world, const void* id128);
int pfr_comm_allreduce(void* comm, void* buf, size_t count, int dtype, int average, pfr_stream_t stream);
int pfr_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* PFR_HIP_H */
