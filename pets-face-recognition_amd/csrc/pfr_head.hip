// pfr_head.hip — the ArcFace / CosFace cosine-margin head with softmax cross-entropy (focal) loss.
//
// Reference semantics (file:line in /root/reference):
//   losses/large_margin.py:69-84  ArcMarginProduct.forward  — cos = normalize(x)·normalize(W)ᵀ ; sine = sqrt(1−cos²) ;
//        phi = cos·cos m − sine·sin m ; hard: where(cos > cos(π−m), phi, cos − sin(π−m)·m) ; easy: where(cos > 0, phi, cos) ;
//        out = s·(onehot·phi + (1−onehot)·cos)
//   losses/large_margin.py:30-40  AddMarginProduct.forward  — phi = cos − m
//   losses/losses.py:22-28        FocalLoss.forward         — logp = CE(·,'none') ; p = exp(−logp) ; mean((1−p)^γ·logp)
//   F.normalize: x / max(‖x‖₂, 1e-12)
//
// Kernels here: row L2-normalisation (forward: writes the compute-dtype x̂ and, for the weight, also x̂ᵀ; backward:
// dx = (dx̂ − x̂·(x̂·dx̂))/‖x‖) with wave-shuffle reductions, and ONE fused row kernel for
// margin → scale → log-softmax → loss → ∂loss/∂cos (the three B×C temporaries of the reference never exist).
// The two cosine GEMMs and their gradients run on the MFMA implicit-GEMM kernels (pfr_igemm.hip / pfr_wgrad.hip).
#include "pfr_common.h"

// one wave per row; D arbitrary
template <typename TI, typename TOo>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const TI* __restrict__ x, TOo* __restrict__ xn, TOo* __restrict__ xnT,
                                                         float* __restrict__ inv_norm, int rows, int D, int ldt, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TI* xr = x + (size_t)row * D;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float v = to_f32(xr[d]);
    ss = fmaf(v, v, ss);
  }
  ss = wave_sum(ss);
  const float inv = 1.f / fmaxf(sqrtf(ss), eps);
  if (lane == 0) inv_norm[row] = inv;
  for (int d = lane; d < D; d += 64) {
    const TOo o = from_f32<TOo>(to_f32(xr[d]) * inv);
    xn[(size_t)row * D + d] = o;
    if (xnT) xnT[(size_t)d * ldt + row] = o;
  }
}

extern "C" int pfr_l2norm_dual(const float* x, void* xn_bf16, float* xn_f32, float* inv_norm, int rows, int D, float eps, hipStream_t st);
extern "C" int pfr_l2norm_fwd(const void* x, int in_dtype, void* xn, void* xnT, int out_dtype, float* inv_norm, int rows,
                              int D, int ldt, float eps, hipStream_t st) {
  PFR_CHECK_ARG(x && xn && inv_norm, "pfr_l2norm_fwd: null pointer");
  // fp32 rows without the transposed copy: the register-resident float4 kernel of the match (one read of x, all loads in flight)
  if (in_dtype == PFR_F32 && !xnT && D % 4 == 0 && D <= 2048 && rows > 0 && (out_dtype == PFR_F32 || out_dtype == PFR_BF16))
    return pfr_l2norm_dual((const float*)x, out_dtype == PFR_BF16 ? xn : nullptr, out_dtype == PFR_F32 ? (float*)xn : nullptr, inv_norm, rows, D, eps, st);
  const dim3 grid((rows + 3) / 4), block(256);
  if (ldt <= 0) ldt = rows;
#define L2N(TI, TOo) hipLaunchKernelGGL((l2norm_fwd_kernel<TI, TOo>), grid, block, 0, st, (const TI*)x, (TOo*)xn, (TOo*)xnT, inv_norm, rows, D, ldt, eps)
  if (in_dtype == PFR_F32 && out_dtype == PFR_F32) L2N(float, float);
  else if (in_dtype == PFR_F32 && out_dtype == PFR_BF16) L2N(float, bf16_t);
  else if (in_dtype == PFR_BF16 && out_dtype == PFR_BF16) L2N(bf16_t, bf16_t);
  else if (in_dtype == PFR_BF16 && out_dtype == PFR_F32) L2N(bf16_t, float);
  else { pfr_set_error("pfr_l2norm_fwd: bad dtypes"); return PFR_ERR_UNSUPPORTED; }
#undef L2N
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// gallery / query preparation of the match: ONE pass over fp32 rows (float4 loads kept in registers) writes the normalised
// row in bf16 (GEMM operand) AND in fp32 (exact re-scoring operand).  D % 4 == 0, D <= 2048.
// NK = 16-byte-x4 chunks per lane (2 for D <= 512, 8 up to 2048): the loads are unconditional (clamped index, zeroed by a select) so
// that they are all in flight together — a load in a branch is followed by a full wait, i.e. one memory round trip per chunk
template <int NK>
__global__ __launch_bounds__(256) void l2norm_dual_kernel(const float* __restrict__ x, bf16_t* __restrict__ xb, float* __restrict__ xf,
                                                          float* __restrict__ inv_norm, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
  const int n4 = D >> 2;
  f32x4 v[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) v[k] = xr[min(lane + 64 * k, n4 - 1)];
  __builtin_amdgcn_sched_barrier(0);
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    if (lane + 64 * k >= n4) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) ss = fmaf(v[k][e], v[k][e], ss);
  }
  ss = wave_sum(ss);
  const float inv = 1.f / fmaxf(sqrtf(ss), eps);
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int i = lane + 64 * k;
    if (i < n4) {
      f32x4 o = v[k] * inv;
      if (xf) reinterpret_cast<f32x4*>(xf + (size_t)row * D)[i] = o;
      if (xb) {
        bf16x4 b;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = (bf16_t)o[e];
        reinterpret_cast<bf16x4*>(xb + (size_t)row * D)[i] = b;
      }
    }
  }
}
extern "C" int pfr_l2norm_dual(const float* x, void* xn_bf16, float* xn_f32, float* inv_norm, int rows, int D, float eps,
                               hipStream_t st) {
  PFR_CHECK_ARG(x && (xn_bf16 || xn_f32) && rows > 0, "pfr_l2norm_dual: null pointer");
  PFR_CHECK_ARG(D % 4 == 0 && D <= 2048, "pfr_l2norm_dual: D must be a multiple of 4 and <= 2048");
  if (D <= 512) hipLaunchKernelGGL(l2norm_dual_kernel<2>, dim3((rows + 3) / 4), dim3(256), 0, st, x, (bf16_t*)xn_bf16, xn_f32, inv_norm, rows, D, eps);
  else hipLaunchKernelGGL(l2norm_dual_kernel<8>, dim3((rows + 3) / 4), dim3(256), 0, st, x, (bf16_t*)xn_bf16, xn_f32, inv_norm, rows, D, eps);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// dx = inv_norm · (dxn − xn · (xn·dxn)) ; xn is the normalised row recomputed in fp32 from x and inv_norm
template <typename TI, typename TOo>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const TI* __restrict__ x, const float* __restrict__ inv_norm,
                                                         const float* __restrict__ dxn, TOo* __restrict__ dx, int rows, int D,
                                                         int accumulate) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float inv = inv_norm[row];
  const TI* xr = x + (size_t)row * D;
  const float* gr = dxn + (size_t)row * D;
  float dot = 0.f;
  for (int d = lane; d < D; d += 64) dot = fmaf(to_f32(xr[d]) * inv, gr[d], dot);
  dot = wave_sum(dot);
  for (int d = lane; d < D; d += 64) {
    float v = inv * (gr[d] - to_f32(xr[d]) * inv * dot);
    if (accumulate) v += to_f32(dx[(size_t)row * D + d]);
    dx[(size_t)row * D + d] = from_f32<TOo>(v);
  }
}

extern "C" int pfr_l2norm_bwd(const void* x, int in_dtype, const float* inv_norm, const float* dxn, void* dx, int out_dtype,
                              int rows, int D, int accumulate, hipStream_t st) {
  PFR_CHECK_ARG(x && inv_norm && dxn && dx, "pfr_l2norm_bwd: null pointer");
  const dim3 grid((rows + 3) / 4), block(256);
#define L2B(TI, TOo) hipLaunchKernelGGL((l2norm_bwd_kernel<TI, TOo>), grid, block, 0, st, (const TI*)x, inv_norm, dxn, (TOo*)dx, rows, D, accumulate)
  if (in_dtype == PFR_F32 && out_dtype == PFR_F32) L2B(float, float);
  else if (in_dtype == PFR_BF16 && out_dtype == PFR_BF16) L2B(bf16_t, bf16_t);
  else if (in_dtype == PFR_BF16 && out_dtype == PFR_F32) L2B(bf16_t, float);
  else if (in_dtype == PFR_F32 && out_dtype == PFR_BF16) L2B(float, bf16_t);
  else { pfr_set_error("pfr_l2norm_bwd: bad dtypes"); return PFR_ERR_UNSUPPORTED; }
#undef L2B
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ------------------------------------------------------------------------------------------------
struct MarginParams {
  float s, cos_m, sin_m, th, mm, m;
  int mode;  // 0 = ArcFace hard margin, 1 = ArcFace easy margin, 2 = CosFace (cos − m), 3 = no margin
  float gamma;
};

__device__ __forceinline__ float block_reduce_max(float v, float* sh) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, sh[i]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  __syncthreads();
  return r;
}

// one block per sample row: 256 threads, or 1024 for long rows (three dependent passes over the row — at 10 000 classes and
// one wave per SIMD each pass is ~40 exposed memory round trips: 30 us with 256 threads)
template <typename TG>
__global__ __launch_bounds__(1024) void margin_ce_kernel(const float* __restrict__ cosv, const int64_t* __restrict__ label,
                                                        MarginParams mp, float* __restrict__ logits, float* __restrict__ loss_rows,
                                                        TG* __restrict__ dcos, int C, int ldc, float gscale,
                                                        const float* __restrict__ gscale_dev) {
  __shared__ float sh[16];
  if (gscale_dev) gscale *= gscale_dev[0];
  const int row = blockIdx.x;
  const int nt = blockDim.x;
  const float* cr = cosv + (size_t)row * ldc;
  const int t = (int)label[row];
  // target logit and d(phi)/d(cos)
  const float ct = cr[t];
  float phi, dphi;
  if (mp.mode == 2) {
    phi = ct - mp.m;
    dphi = 1.f;
  } else if (mp.mode == 3) {
    phi = ct;
    dphi = 1.f;
  } else {
    const float sine = sqrtf(fmaxf(1.f - ct * ct, 0.f));  // reference is NaN for |cos| > 1 (rounding); clamped here
    const float ph = ct * mp.cos_m - sine * mp.sin_m;
    const float dph = mp.cos_m + (sine > 0.f ? mp.sin_m * ct / sine : 0.f);
    const bool take = mp.mode == 1 ? (ct > 0.f) : (ct > mp.th);
    phi = take ? ph : (mp.mode == 1 ? ct : ct - mp.mm);
    dphi = take ? dph : 1.f;
  }
  const float lt = mp.s * phi;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < C; j += nt) {
    const float l = (j == t) ? lt : mp.s * cr[j];
    mx = fmaxf(mx, l);
  }
  mx = block_reduce_max(mx, sh);
  float se = 0.f;
  for (int j = threadIdx.x; j < C; j += nt) {
    const float l = (j == t) ? lt : mp.s * cr[j];
    se += expf(l - mx);
  }
  se = block_reduce_sum(se, sh);
  const float lse = mx + logf(se);
  const float logp = lse - lt;  // cross-entropy of this row
  const float pt = expf(-logp);
  float f = 1.f, lossv = logp;
  if (mp.gamma != 0.f) {
    const float om = fmaxf(1.f - pt, 0.f);
    lossv = powf(om, mp.gamma) * logp;
    f = powf(om, mp.gamma) + mp.gamma * logp * pt * powf(om, mp.gamma - 1.f);
  }
  if (threadIdx.x == 0 && loss_rows) loss_rows[row] = lossv;
  const float gs = gscale * f;
  for (int j = threadIdx.x; j < C; j += nt) {
    const float l = (j == t) ? lt : mp.s * cr[j];
    if (logits) logits[(size_t)row * C + j] = l;
    if (dcos) {
      const float p = expf(l - lse);
      float d = (j == t) ? (p - 1.f) * dphi : p;
      dcos[(size_t)row * ldc + j] = from_f32<TG>(d * mp.s * gs);
    }
  }
}

extern "C" int pfr_margin_ce(const float* cosv, const int64_t* label, int B, int C, int ldc, int mode, float s, float m,
                             float gamma, float grad_scale, const float* grad_scale_dev, float* logits, float* loss_rows, void* dcos,
                             int dcos_dtype, hipStream_t st) {
  PFR_CHECK_ARG(cosv && label && B > 0 && C > 0, "pfr_margin_ce: bad args");
  PFR_CHECK_ARG(mode >= 0 && mode <= 3, "pfr_margin_ce: bad margin mode %d", mode);
  MarginParams mp;
  mp.s = s; mp.m = m; mp.mode = mode; mp.gamma = gamma;
  mp.cos_m = (float)cos((double)m);
  mp.sin_m = (float)sin((double)m);
  mp.th = (float)cos(M_PI - (double)m);
  mp.mm = (float)(sin(M_PI - (double)m) * (double)m);
  if (ldc <= 0) ldc = C;
  const int nt = C >= 4096 ? 1024 : 256;
  if (dcos_dtype == PFR_BF16)
    hipLaunchKernelGGL(margin_ce_kernel<bf16_t>, dim3(B), dim3(nt), 0, st, cosv, label, mp, logits, loss_rows, (bf16_t*)dcos, C, ldc, grad_scale, grad_scale_dev);
  else
    hipLaunchKernelGGL(margin_ce_kernel<float>, dim3(B), dim3(nt), 0, st, cosv, label, mp, logits, loss_rows, (float*)dcos, C, ldc, grad_scale, grad_scale_dev);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// mean of a small fp32 vector (the per-row losses) → scalar
__global__ void mean_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
  __shared__ float sh[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += x[i];
  a = block_reduce_sum(a, sh);
  if (threadIdx.x == 0) out[0] = a / n;
}
extern "C" int pfr_mean(const float* x, float* out, int n, hipStream_t st) {
  PFR_CHECK_ARG(x && out && n > 0, "pfr_mean: bad args");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, x, out, n);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// standalone backward of the margin: dcos = s·dlogits (·dphi/dcos on the target column)  — used when
// ArcMarginProduct / AddMarginProduct run as separate modules (losses/large_margin.py:30-40,69-84)
template <typename TG>
__global__ __launch_bounds__(256) void margin_bwd_kernel(const float* __restrict__ cosv, const int64_t* __restrict__ label,
                                                         MarginParams mp, const float* __restrict__ dlogits,
                                                         TG* __restrict__ dcos, int C, int ldc) {
  const int row = blockIdx.x;
  const int t = (int)label[row];
  const float ct = cosv[(size_t)row * ldc + t];
  float dphi = 1.f;
  if (mp.mode == 0 || mp.mode == 1) {
    const float sine = sqrtf(fmaxf(1.f - ct * ct, 0.f));
    const float dph = mp.cos_m + (sine > 0.f ? mp.sin_m * ct / sine : 0.f);
    const bool take = mp.mode == 1 ? (ct > 0.f) : (ct > mp.th);
    dphi = take ? dph : 1.f;
  }
  for (int j = threadIdx.x; j < C; j += 256) {
    float d = dlogits[(size_t)row * C + j] * mp.s;
    if (j == t) d *= dphi;
    dcos[(size_t)row * ldc + j] = from_f32<TG>(d);
  }
}
extern "C" int pfr_margin_bwd(const float* cosv, const int64_t* label, int B, int C, int ldc, int mode, float s, float m,
                              const float* dlogits, void* dcos, int dcos_dtype, hipStream_t st) {
  PFR_CHECK_ARG(cosv && label && dlogits && dcos, "pfr_margin_bwd: null pointer");
  MarginParams mp;
  mp.s = s; mp.m = m; mp.mode = mode; mp.gamma = 0.f;
  mp.cos_m = (float)cos((double)m);
  mp.sin_m = (float)sin((double)m);
  mp.th = (float)cos(M_PI - (double)m);
  mp.mm = (float)(sin(M_PI - (double)m) * (double)m);
  if (ldc <= 0) ldc = C;
  if (dcos_dtype == PFR_BF16)
    hipLaunchKernelGGL(margin_bwd_kernel<bf16_t>, dim3(B), dim3(256), 0, st, cosv, label, mp, dlogits, (bf16_t*)dcos, C, ldc);
  else
    hipLaunchKernelGGL(margin_bwd_kernel<float>, dim3(B), dim3(256), 0, st, cosv, label, mp, dlogits, (float*)dcos, C, ldc);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
