// pfr_bnfree.hip — the backward pass of a bottleneck's last convolution + BatchNorm WITHOUT the BatchNorm's input tensor.
//
// Reference: autograd of `out = relu(bn3(conv3(z)) + shortcut)` in torchvision's Bottleneck.forward (third-party to
// /root/reference; built at configs/dog_fe/fe_dogs_config.py:102-103) = NativeBatchNormBackward + ConvolutionBackward.
//
// conv3 is a 1x1 convolution, x = Z·Wᵀ (Z [M][K] = relu(bn2(c2)), W [C][K]), and BatchNorm backward is LINEAR in (G, x):
//     dx = A∘G + B∘(x − μ) + C0,   A = γ r,  B = −γ r² dγ / M,  C0 = −γ r dβ / M,   dβ = Σ_rows G,  dγ = Σ_rows G∘x̂
// (G = the gradient that reaches bn3's output through the block's ReLU mask, r = 1/sqrt(var + eps)).  Substituting x − μ = (Z − z̄)·Wᵀ:
//     dγ_c   = r_c · Σ_k W[c,k] (G1[c,k] − dβ_c z̄_k)                                   G1 = Gᵀ·Z   (the weight-gradient GEMM of G itself)
//     dZ     = G·(A∘W) + Z·S + bias,     S = Wᵀ diag(B) W [K][K],   bias = C0ᵀW − z̄·S     (ONE GEMM over [G | Z], or two)
//     dW     = A∘G1 + B∘(W·(G2 − M z̄ z̄ᵀ)) + C0 ⊗ (M z̄)                                 G2 = ZᵀZ
// so neither x = conv3's output nor dx is ever read or written in the backward pass: per block the three passes over the widest
// tensor of the network (pfr_bn_bwd_apply: read G, read x, write dx) and the two reads of dx by the data- and weight-gradient GEMMs
// become two reads of G.  Checked against fp64 (tools/bnfree_check.py): dZ as accurate as the materialised form, dW and dγ
// 50x more accurate (no bf16 rounding of x and dx in between).  The two kernels here are the small per-channel / per-weight parts.
#include "pfr_common.h"

// stage 1, one wave per channel c: dβ_c (sum of the producer's partial rows, fixed order), dγ_c, the coefficient rows A, B, C0
template <typename TW>
__global__ __launch_bounds__(256) void bn3_coef_kernel(const float* __restrict__ part, int nparts, const float* __restrict__ G1,
                                                       const float* __restrict__ zsum, const TW* __restrict__ W,
                                                       const float* __restrict__ gamma, const float* __restrict__ invstd, int C, int K,
                                                       float count, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                       float* __restrict__ coef, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  float db = 0.f;
  for (int t = lane; t < nparts; t += 64) db += part[(size_t)t * 2 * C + c];   // (lane-strided, then a fixed butterfly: deterministic)
  db = wave_sum(db);
  const float inv_m = 1.f / count;
  float t1 = 0.f;
  for (int k = lane; k < K; k += 64) t1 = fmaf((float)W[(size_t)c * K + k], G1[(size_t)c * K + k] - db * (zsum[k] * inv_m), t1);
  t1 = wave_sum(t1);
  if (lane == 0) {
    const float r = invstd[c], g = gamma[c];
    const float dg = r * t1;
    coef[c] = g * r;
    coef[C + c] = -g * r * r * dg * inv_m;
    coef[2 * C + c] = -g * r * db * inv_m;
    dgamma[c] = accumulate ? dgamma[c] + dg : dg;
    dbeta[c] = accumulate ? dbeta[c] + db : db;
  }
}

// stage 2.  Blocks [0, K): column k of S and bias_k.  Blocks [K, K + C/4): one wave per channel c: row c of dW and column c of
// the data-gradient weights wa_t[k][c] = A_c W[c][k] (the [Cin][1][1][Cout] layout pfr_conv2d_fwd takes for a data gradient).
template <typename TW>
__global__ __launch_bounds__(256) void bn3_weights_kernel(const float* __restrict__ coef, const float* __restrict__ G1,
                                                          const float* __restrict__ G2, const float* __restrict__ zsum,
                                                          const TW* __restrict__ W, int C, int K, float count,
                                                          float* __restrict__ dW, bf16_t* __restrict__ wa_t, bf16_t* __restrict__ S,
                                                          float* __restrict__ bias, int accumulate, int ldw, int lds_) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const float inv_m = 1.f / count;
  if ((int)blockIdx.x < K) {
    const int k = blockIdx.x;
    // S[j][k] = sum_c W[c][j] B_c W[c][k]: thread -> (j, channel slice); K is 64 / 128 / 256, so 256 / K slices walk the channels
    const int j = tid & (K - 1), part = tid / K, np = 256 / K;
    float s = 0.f;
#pragma unroll 8
    for (int c = part; c < C; c += np) s = fmaf((float)W[(size_t)c * K + j] * coef[C + c], (float)W[(size_t)c * K + k], s);
    red[tid] = s;
    __syncthreads();
    float b = 0.f;
    if (tid < K) {
      for (int q = 1; q < np; ++q) s += red[q * K + tid];     // fixed order
      const bf16_t sb = (bf16_t)s;
      S[(size_t)tid * lds_ + k] = sb;    // symmetric: row / column orientation is the same matrix
      // bias_k = sum_c C0_c W[c][k] - sum_j zbar_j S_bf16[j][k]   (the ROUNDED S: the two terms then cancel as (Z - zbar)·S does)
      b = -(zsum[tid] * inv_m) * (float)sb;
    }
    __syncthreads();
#pragma unroll 4
    for (int c = tid; c < C; c += 256) b = fmaf(coef[2 * C + c], (float)W[(size_t)c * K + k], b);
    red[tid] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) bias[k] = red[0];
    return;
  }
  const int lane = tid & 63;
  const int c = ((int)blockIdx.x - K) * 4 + (tid >> 6);
  if (c >= C) return;
  const float A = coef[c], B = coef[C + c], C0 = coef[2 * C + c];
  for (int k = lane; k < K; k += 64) {
    const float zk = zsum[k] * inv_m;
    float t = 0.f;   // sum_j W[c][j] (G2[j][k] - M zbar_j zbar_k)
#pragma unroll 8
    for (int j = 0; j < K; ++j) t = fmaf((float)W[(size_t)c * K + j], G2[(size_t)j * K + k] - zsum[j] * zk, t);
    const float w = (float)W[(size_t)c * K + k];
    const float v = fmaf(A, G1[(size_t)c * K + k], fmaf(B, t, C0 * zsum[k]));
    dW[(size_t)c * K + k] = accumulate ? dW[(size_t)c * K + k] + v : v;
    wa_t[(size_t)k * ldw + c] = (bf16_t)(A * w);
  }
}

// Batch statistics of x = Z·Wᵀ WITHOUT computing x: mean_c = W[c,:]·z̄ and var_c = W[c,:]·(G2/M − z̄z̄ᵀ)·W[c,:]ᵀ from the Gram matrix and the
// column sums of Z (pfr_gram_colsum) — a bottleneck's bn3 gets its statistics from conv3's INPUT, so the forward pass needs no
// statistics pass over conv3 at all.  W = the bf16 weights the convolution itself uses.  Output: one (mean, M2 = M·var) partial row,
// the form pfr_bn_finalize merges (nparts = 1, rows_per_part = M).  fp32; checked against fp64: invstd to ~1e-6 relative.
struct BnFinArgs {   // the tail of pfr_bn_finalize, done by the same launch (pfr_bn_finalize_from_gram)
  const float* gamma; const float* beta; float eps, momentum;
  float* running_mean; float* running_var; float* mean; float* invstd; float* scale; float* shift;
};
// Cancellation guard: var_c = W_c (G2/M - zbar zbar^T) W_c^T subtracts numbers of the size of E[x^2]; the rounding error of
// that sum is ~ GRAM_EPS * qa with qa = sum_jk |w_j| (|G2_jk| / M + |zbar_j zbar_k|) |w_k| (fp32 unit roundoff x the accumulated rounding of
// the Gram slabs).  When that is more than 1 % of var + eps (|mean| >> std: a nearly constant channel), the channel is REPORTED through
// *cancel_flag (host-visible memory; the engine then goes back to the statistics pass over conv3's output) instead of trusted silently.
#define PFR_GRAM_EPS 1.0e-6f
template <bool FIN>
__global__ __launch_bounds__(256) void bn_stats_gram_kernel(const float* __restrict__ gram, const bf16_t* __restrict__ W, int C, int K,
                                                            float count, float* __restrict__ part, BnFinArgs f, float eps_guard,
                                                            int* __restrict__ cancel_flag) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const float* G2 = gram;
  const float* zsum = gram + (size_t)K * K;
  const float inv_m = 1.f / count;
  float mu = 0.f, q = 0.f, qa = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float wk = (float)W[(size_t)c * K + k], zk = zsum[k] * inv_m;
    float t = 0.f, ta = 0.f;   // sum_j W[c][j] Cov[j][k], and the same sum over absolute values (size of what cancels)
#pragma unroll 8
    for (int j = 0; j < K; ++j) {
      const float wj = (float)W[(size_t)c * K + j], g = G2[(size_t)j * K + k] * inv_m, zz = (zsum[j] * inv_m) * zk;
      t = fmaf(wj, g - zz, t);
      ta = fmaf(fabsf(wj), fabsf(g) + fabsf(zz), ta);
    }
    mu = fmaf(wk, zk, mu);
    q = fmaf(t, wk, q);
    qa = fmaf(ta, fabsf(wk), qa);
  }
  mu = wave_sum(mu);
  q = wave_sum(q);
  qa = wave_sum(qa);
  if (lane == 0) {
    if (cancel_flag && PFR_GRAM_EPS * qa > 0.01f * (fmaxf(q, 0.f) + eps_guard)) *cancel_flag = 1;
    const float m2 = fmaxf(q, 0.f) * count;
    if constexpr (!FIN) {
      part[c] = mu;
      part[C + c] = m2;
    } else {   // the tail of bn_finalize_kernel for this one partial row
      const float g = f.gamma ? f.gamma[c] : 1.f, bb = f.beta ? f.beta[c] : 0.f;
      const float var = fmaxf(m2 / count, 0.f);
      const float invstd = rsqrtf(var + f.eps);
      f.mean[c] = mu;
      f.invstd[c] = invstd;
      f.scale[c] = g * invstd;
      f.shift[c] = bb - mu * g * invstd;
      if (f.running_mean) {
        const float unb = count > 1.f ? var * count / (count - 1.f) : var;
        f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mu;
        f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * unb;
      }
    }
  }
}
extern "C" int pfr_bn_stats_from_gram(const float* gram, const void* W, int dtype, int C, int K, float count, float* part, float eps,
                                      int* cancel_flag, hipStream_t st) {
  PFR_CHECK_ARG(gram && W && part, "pfr_bn_stats_from_gram: null pointer");
  PFR_CHECK_ARG(dtype == PFR_BF16 && C > 0 && K > 0 && count > 0.f, "pfr_bn_stats_from_gram: bf16 weights only");
  hipLaunchKernelGGL(bn_stats_gram_kernel<false>, dim3((C + 3) / 4), dim3(256), 0, st, gram, (const bf16_t*)W, C, K, count, part, BnFinArgs{}, eps, cancel_flag);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
// pfr_bn_stats_from_gram + pfr_bn_finalize(nparts = 1) in ONE launch (the forward pass of a BN-input-free block is a chain of small
// dependent launches: Gram -> slab sum -> statistics -> finalize -> tail; this removes a link): the same arithmetic (pfr_bn_finalize's
// merge of the single row rounds the mean once more: results agree to the last bit or two)
extern "C" int pfr_bn_finalize_from_gram(const float* gram, const void* W, int dtype, int C, int K, float count, const float* gamma,
                                         const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                                         float* invstd, float* scale, float* shift, int* cancel_flag, hipStream_t st) {
  PFR_CHECK_ARG(gram && W && mean && invstd && scale && shift, "pfr_bn_finalize_from_gram: null pointer");
  PFR_CHECK_ARG(dtype == PFR_BF16 && C > 0 && K > 0 && count > 0.f, "pfr_bn_finalize_from_gram: bf16 weights only");
  PFR_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "pfr_bn_finalize_from_gram: running_mean and running_var go together");
  BnFinArgs f{gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift};
  hipLaunchKernelGGL(bn_stats_gram_kernel<true>, dim3((C + 3) / 4), dim3(256), 0, st, gram, (const bf16_t*)W, C, K, count, nullptr, f, eps, cancel_flag);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

extern "C" int pfr_bn3_bwd_coef(const float* part, int nparts, const float* G1, const float* zsum, const void* W, int wdtype, const float* gamma,
                                const float* invstd, int C, int K, float count, float* dgamma, float* dbeta, float* coef,
                                int accumulate, hipStream_t st) {
  PFR_CHECK_ARG(part && G1 && zsum && W && gamma && invstd && dgamma && dbeta && coef, "pfr_bn3_bwd_coef: null pointer");
  PFR_CHECK_ARG(nparts > 0 && C > 0 && K > 0 && count > 0.f, "pfr_bn3_bwd_coef: bad geometry");
  PFR_CHECK_ARG(wdtype == PFR_BF16 || wdtype == PFR_F32, "pfr_bn3_bwd_coef: W is bf16 or f32 (the weights the forward convolution used)");
  if (wdtype == PFR_BF16)
    hipLaunchKernelGGL(bn3_coef_kernel<bf16_t>, dim3((C + 3) / 4), dim3(256), 0, st, part, nparts, G1, zsum, (const bf16_t*)W, gamma, invstd, C, K,
                       count, dgamma, dbeta, coef, accumulate);
  else
    hipLaunchKernelGGL(bn3_coef_kernel<float>, dim3((C + 3) / 4), dim3(256), 0, st, part, nparts, G1, zsum, (const float*)W, gamma, invstd, C, K, count, dgamma,
                     dbeta, coef, accumulate);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

extern "C" int pfr_bn3_bwd_weights(const float* coef, const float* G1, const float* G2, const float* zsum, const void* W, int wdtype, int C, int K,
                                   float count, float* dW, void* wa_t, void* S, float* bias, int accumulate, hipStream_t st) {
  PFR_CHECK_ARG(wdtype == PFR_BF16 || wdtype == PFR_F32, "pfr_bn3_bwd_weights: W is bf16 or f32 (the weights the forward convolution used)");
  PFR_CHECK_ARG(coef && G1 && G2 && zsum && W && dW && wa_t && bias, "pfr_bn3_bwd_weights: null pointer");
  // S == NULL: ONE concatenated weight tensor wa_t = wcat [K][C + K], row k = [A∘W column k | S row k] (pfr_conv1x1_dgrad2_bn)
  const int ldw = S ? C : C + K, lds_ = S ? K : C + K;
  if (!S) S = (bf16_t*)wa_t + C;
  PFR_CHECK_ARG(C > 0 && (K == 64 || K == 128 || K == 256) && count > 0.f, "pfr_bn3_bwd_weights: K must be 64, 128 or 256");
  if (wdtype == PFR_BF16)
    hipLaunchKernelGGL(bn3_weights_kernel<bf16_t>, dim3(K + (C + 3) / 4), dim3(256), 0, st, coef, G1, G2, zsum, (const bf16_t*)W, C, K, count, dW,
                       (bf16_t*)wa_t, (bf16_t*)S, bias, accumulate, ldw, lds_);
  else
    hipLaunchKernelGGL(bn3_weights_kernel<float>, dim3(K + (C + 3) / 4), dim3(256), 0, st, coef, G1, G2, zsum, (const float*)W, C, K, count, dW, (bf16_t*)wa_t,
                     (bf16_t*)S, bias, accumulate, ldw, lds_);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
