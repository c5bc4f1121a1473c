// pfr_swin.hip — the non-GEMM kernels of the Swin-T feature extractor (BASELINE config 4): LayerNorm, exact GELU,
// fused shifted-window attention (forward and backward), and the NHWC→NCHW fp32 un-permute of the patch-merging weight
// gradient.  Linear layers (to_qkv, to_out, MLP, patch merging as a stride-f conv, head) run on pfr_igemm / pfr_wgrad.
//
// Reference semantics (/root/reference/models/swin.py):
//   LayerNorm (29, 215)                     row mean / biased variance over C, eps 1e-5, affine
//   FeedForward (39-43)                     nn.GELU() = 0.5·x·(1 + erf(x/√2))
//   WindowAttention.forward (101-135)       optional cyclic shift by −w/2 (torch.roll), windows of w×w tokens,
//       dots = q·kᵀ·head_dim^-½ + pos_embedding[rel_idx]  (one (2w−1)² table shared by all heads, 94-95,118)
//       shifted: −inf mask added to the LAST ROW of windows (upper/lower halves) and to the LAST COLUMN of windows
//       (left/right halves) (122-124);  softmax over keys;  out = attn·v;  windows merged, shift rolled back.
// The shift and the window partition are pure addressing here: token (wy,wx) of window (gy,gx) lives at image position
// ((gy·w + wy + d) mod H, (gx·w + wx + d) mod W) with d = w/2 for shifted blocks — nothing is rolled or copied.
#include "pfr_common.h"

// ------------------------------------------------------------------------------------------------ LayerNorm
// one wave per row
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                            int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += to_f32(xr[c]);
  s = wave_sum(s);
  const float mu = s / C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = to_f32(xr[c]) - mu;
    v = fmaf(d, d, v);
  }
  v = wave_sum(v);
  const float rs = rsqrtf(v / C + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  for (int c = lane; c < C; c += 64)
    y[row * C + c] = from_f32<T>(fmaf((to_f32(xr[c]) - mu) * rs, gamma[c], beta[c]));
}

extern "C" int pfr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 int dtype, long rows, int C, float eps, hipStream_t st) {
  PFR_CHECK_ARG(x && gamma && beta && y, "pfr_layernorm_fwd: null pointer");
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == PFR_BF16)
    hipLaunchKernelGGL(layernorm_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, C, eps);
  else
    hipLaunchKernelGGL(layernorm_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, gamma, beta, (float*)y, mean, rstd, rows, C, eps);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// dx = rstd·(g − mean(g) − x̂·mean(g·x̂)),  g = dy·γ ;  per-block partials of dγ = Σ dy·x̂, dβ = Σ dy  → part [2][nblk][C]
// `dres` (optional) is added to dx: the residual branch gradient that joins at the LayerNorm input.
// NKC = channels per lane = ceil(C/64) (compile-time so that the per-lane partial sums live in registers)
template <typename T, int LN_MAXK>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const T* __restrict__ dres,
                                                            T* __restrict__ dx, float* __restrict__ part, long rows, int C,
                                                            int rows_per_block) {
  extern __shared__ float sh[];  // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  // lane owns channels lane, lane+64, ...: dγ / dβ partials stay in registers over all rows of this wave
  float ag[LN_MAXK], ab[LN_MAXK], gm[LN_MAXK];
  constexpr int nkc = LN_MAXK;
#pragma unroll
  for (int k = 0; k < LN_MAXK; ++k) {
    ag[k] = 0.f; ab[k] = 0.f;
    gm[k] = (k < nkc && lane + 64 * k < C) ? gamma[lane + 64 * k] : 0.f;
  }
  for (long row = r0 + wave; row < r1; row += 4) {
    const T* xr = x + row * C;
    const T* gr = dy + row * C;
    const float mu = mean[row], rs = rstd[row];
    float xh[LN_MAXK], dv[LN_MAXK];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXK; ++k) {
      if (k < nkc) {
        const int c = lane + 64 * k;
        const bool ok = c < C;
        xh[k] = ok ? (to_f32(xr[c]) - mu) * rs : 0.f;
        dv[k] = ok ? to_f32(gr[c]) : 0.f;
        const float g = dv[k] * gm[k];
        a += g;
        b = fmaf(g, xh[k], b);
      }
    }
    a = wave_sum(a) / C;
    b = wave_sum(b) / C;
#pragma unroll
    for (int k = 0; k < LN_MAXK; ++k) {
      if (k < nkc) {
        const int c = lane + 64 * k;
        if (c < C) {
          float v = rs * (dv[k] * gm[k] - a - xh[k] * b);
          if (dres) v += to_f32(dres[row * C + c]);
          dx[row * C + c] = from_f32<T>(v);
          ag[k] = fmaf(dv[k], xh[k], ag[k]);
          ab[k] += dv[k];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < LN_MAXK; ++k) {
    if (k < nkc && lane + 64 * k < C) {
      sh[(wave * 2 + 0) * C + lane + 64 * k] = ag[k];
      sh[(wave * 2 + 1) * C + lane + 64 * k] = ab[k];
    }
  }
  __syncthreads();
  // layout [2][nblk][C]: all dgamma partial rows, then all dbeta partial rows (each half is a plain [nblk][C] matrix)
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    const int q = c / C, cc = c % C;
    const float v = sh[(0 * 2 + q) * C + cc] + sh[(1 * 2 + q) * C + cc] + sh[(2 * 2 + q) * C + cc] + sh[(3 * 2 + q) * C + cc];
    part[((size_t)q * gridDim.x + blockIdx.x) * C + cc] = v;
  }
}

extern "C" int pfr_layernorm_bwd_blocks(long rows) {
  long nb = (rows + 15) / 16;   // >= 4 rows per wave: the row loop is latency-bound, so favour many resident waves
  if (nb > 16384) nb = 16384;
  return (int)(nb < 1 ? 1 : nb);
}

template <typename T>
static int ln_bwd_launch(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                         const void* dres, void* dx, float* part, long rows, int C, int nb, int rpb, hipStream_t st) {
  const size_t shb = (size_t)8 * C * sizeof(float);
  const int nkc = (C + 63) / 64;
#define PFR_LN_CASE(K)                                                                                                  \
  if (nkc <= K) {                                                                                                       \
    hipLaunchKernelGGL((layernorm_bwd_kernel<T, K>), dim3(nb), dim3(256), shb, st, (const T*)dy, (const T*)x, mean, rstd, gamma, \
                       (const T*)dres, (T*)dx, part, rows, C, rpb);                                                    \
    return PFR_OK;                                                                                                      \
  }
  PFR_LN_CASE(2) PFR_LN_CASE(3) PFR_LN_CASE(4) PFR_LN_CASE(6) PFR_LN_CASE(8) PFR_LN_CASE(12) PFR_LN_CASE(16) PFR_LN_CASE(24)
  PFR_LN_CASE(32)
#undef PFR_LN_CASE
  pfr_set_error("pfr_layernorm_bwd: C > 2048");
  return PFR_ERR_UNSUPPORTED;
}

extern "C" int pfr_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                 const void* dres, void* dx, float* part, int dtype, long rows, int C, hipStream_t st) {
  PFR_CHECK_ARG(dy && x && mean && rstd && gamma && dx && part, "pfr_layernorm_bwd: null pointer");
  const int nb = pfr_layernorm_bwd_blocks(rows);
  const int rpb = (int)((rows + nb - 1) / nb);
  int rc = dtype == PFR_BF16 ? ln_bwd_launch<bf16_t>(dy, x, mean, rstd, gamma, dres, dx, part, rows, C, nb, rpb, st)
                             : ln_bwd_launch<float>(dy, x, mean, rstd, gamma, dres, dx, part, rows, C, nb, rpb, st);
  if (rc != PFR_OK) return rc;
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ------------------------------------------------------------------------------------------------ GELU (exact, erf)
template <typename T>
__global__ void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, size_t nchunks) {
  constexpr int KP = DT<T>::KPACK;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nchunks; i += stride) {
    float f[KP];
    Chunk<T>::unpack(ld16(x + i * KP), f);
#pragma unroll
    for (int e = 0; e < KP; ++e) f[e] = 0.5f * f[e] * (1.f + erff(f[e] * 0.70710678118654752f));
    st16(y + i * KP, Chunk<T>::pack(f));
  }
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, size_t nchunks) {
  constexpr int KP = DT<T>::KPACK;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nchunks; i += stride) {
    float f[KP], g[KP];
    Chunk<T>::unpack(ld16(x + i * KP), f);
    Chunk<T>::unpack(ld16(dy + i * KP), g);
#pragma unroll
    for (int e = 0; e < KP; ++e) {
      const float v = f[e];
      const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
      g[e] *= cdf + v * pdf;
    }
    st16(dx + i * KP, Chunk<T>::pack(g));
  }
}
extern "C" int pfr_gelu_fwd(const void* x, void* y, int dtype, size_t n, hipStream_t st) {
  PFR_CHECK_ARG(x && y, "pfr_gelu_fwd: null pointer");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  PFR_CHECK_ARG(n % kp == 0, "pfr_gelu_fwd: n %% %d != 0", kp);
  const size_t nch = n / kp;
  unsigned blocks = (unsigned)((nch + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (dtype == PFR_BF16) hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, nch);
  else hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y, nch);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
extern "C" int pfr_gelu_bwd(const void* x, const void* dy, void* dx, int dtype, size_t n, hipStream_t st) {
  PFR_CHECK_ARG(x && dy && dx, "pfr_gelu_bwd: null pointer");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  PFR_CHECK_ARG(n % kp == 0, "pfr_gelu_bwd: n %% %d != 0", kp);
  const size_t nch = n / kp;
  unsigned blocks = (unsigned)((nch + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (dtype == PFR_BF16) hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, nch);
  else hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, nch);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ------------------------------------------------------------------------------------------------ window attention
// One 256-thread workgroup per (image, window, head).  q/k/v/probabilities live in LDS as fp32 (w² ≤ 64 tokens,
// head_dim ≤ 32).  qkv is [B][H][W][3·heads·hd] with channel order (q | k | v) x (head, d), as produced by to_qkv.
#define WA_MAXT 64
#define WA_MAXD 32

struct WinAttn {
  int B, H, W, heads, hd, w, shift;  // shift = displacement d (0 for regular blocks)
  float scale;
};

__device__ __forceinline__ float wa_mask_bias(const WinAttn& a, int gy, int gx, int nwh, int nww, int i, int j,
                                              const float* __restrict__ pos) {
  const int w = a.w;
  const int yi = i / w, xi = i % w, yj = j / w, xj = j % w;
  float b = pos[(yj - yi + w - 1) * (2 * w - 1) + (xj - xi + w - 1)];
  if (a.shift) {
    // last row of windows: tokens from the upper (w-d rows) and lower (d rows) part must not see each other
    if (gy == nwh - 1 && ((yi >= w - a.shift) != (yj >= w - a.shift))) b = -INFINITY;
    // last column of windows: left / right parts
    if (gx == nww - 1 && ((xi >= w - a.shift) != (xj >= w - a.shift))) b = -INFINITY;
  }
  return b;
}

__device__ __forceinline__ size_t wa_token_off(const WinAttn& a, int b, int gy, int gx, int t) {
  int y = gy * a.w + t / a.w + a.shift, x = gx * a.w + t % a.w + a.shift;
  if (y >= a.H) y -= a.H;
  if (x >= a.W) x -= a.W;
  return ((size_t)b * a.H + y) * a.W + x;
}

template <typename T>
__global__ __launch_bounds__(256) void window_attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ pos,
                                                              T* __restrict__ out, WinAttn a) {
  __shared__ float q[WA_MAXT][WA_MAXD + 1], k[WA_MAXT][WA_MAXD + 1], v[WA_MAXT][WA_MAXD + 1];
  __shared__ float s[WA_MAXT][WA_MAXT + 1];
  const int nwh = a.H / a.w, nww = a.W / a.w, nt = a.w * a.w, C = a.heads * a.hd;
  int bid = blockIdx.x;
  const int h = bid % a.heads; bid /= a.heads;
  const int gx = bid % nww; bid /= nww;
  const int gy = bid % nwh;
  const int b = bid / nwh;
  for (int e = threadIdx.x; e < nt * a.hd; e += 256) {
    const int t = e / a.hd, d = e % a.hd;
    const T* base = qkv + wa_token_off(a, b, gy, gx, t) * (3 * C) + h * a.hd + d;
    q[t][d] = to_f32(base[0]);
    k[t][d] = to_f32(base[C]);
    v[t][d] = to_f32(base[2 * C]);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nt * nt; e += 256) {
    const int i = e / nt, j = e % nt;
    float acc = 0.f;
    for (int d = 0; d < a.hd; ++d) acc = fmaf(q[i][d], k[j][d], acc);
    s[i][j] = acc * a.scale + wa_mask_bias(a, gy, gx, nwh, nww, i, j, pos);
  }
  __syncthreads();
  // softmax: one wave handles rows wave, wave+4, ...
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < nt; i += 4) {
    const float x = lane < nt ? s[i][lane] : -INFINITY;
    const float m = wave_max(x);
    const float ex = lane < nt ? __expf(x - m) : 0.f;
    const float sum = wave_sum(ex);
    if (lane < nt) s[i][lane] = ex / sum;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nt * a.hd; e += 256) {
    const int i = e / a.hd, d = e % a.hd;
    float acc = 0.f;
    for (int j = 0; j < nt; ++j) acc = fmaf(s[i][j], v[j][d], acc);
    out[wa_token_off(a, b, gy, gx, i) * C + h * a.hd + d] = from_f32<T>(acc);
  }
}

// backward: recomputes the probabilities; writes dqkv and this workgroup's partial of the position-table gradient
// (dpos_part [nblocks][(2w−1)²], summed afterwards by pfr_colsum → deterministic).
template <typename T>
__global__ __launch_bounds__(256) void window_attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ pos,
                                                              const T* __restrict__ dout, T* __restrict__ dqkv,
                                                              float* __restrict__ dpos_part, WinAttn a) {
  __shared__ float q[WA_MAXT][WA_MAXD + 1], k[WA_MAXT][WA_MAXD + 1], v[WA_MAXT][WA_MAXD + 1], go[WA_MAXT][WA_MAXD + 1];
  __shared__ float s[WA_MAXT][WA_MAXT + 1], ds[WA_MAXT][WA_MAXT + 1];
  __shared__ float dtab[256];
  const int nwh = a.H / a.w, nww = a.W / a.w, nt = a.w * a.w, C = a.heads * a.hd;
  const int ntab = (2 * a.w - 1) * (2 * a.w - 1);
  int bid = blockIdx.x;
  const int h = bid % a.heads; bid /= a.heads;
  const int gx = bid % nww; bid /= nww;
  const int gy = bid % nwh;
  const int b = bid / nwh;
  for (int e = threadIdx.x; e < ntab; e += 256) dtab[e] = 0.f;
  for (int e = threadIdx.x; e < nt * a.hd; e += 256) {
    const int t = e / a.hd, d = e % a.hd;
    const size_t tok = wa_token_off(a, b, gy, gx, t);
    const T* base = qkv + tok * (3 * C) + h * a.hd + d;
    q[t][d] = to_f32(base[0]);
    k[t][d] = to_f32(base[C]);
    v[t][d] = to_f32(base[2 * C]);
    go[t][d] = to_f32(dout[tok * C + h * a.hd + d]);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nt * nt; e += 256) {
    const int i = e / nt, j = e % nt;
    float acc = 0.f, dp = 0.f;
    for (int d = 0; d < a.hd; ++d) {
      acc = fmaf(q[i][d], k[j][d], acc);
      dp = fmaf(go[i][d], v[j][d], dp);   // dP = dO·Vᵀ
    }
    s[i][j] = acc * a.scale + wa_mask_bias(a, gy, gx, nwh, nww, i, j, pos);
    ds[i][j] = dp;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < nt; i += 4) {
    const float x = lane < nt ? s[i][lane] : -INFINITY;
    const float m = wave_max(x);
    const float ex = lane < nt ? __expf(x - m) : 0.f;
    const float sum = wave_sum(ex);
    const float p = ex / sum;
    const float dp = lane < nt ? ds[i][lane] : 0.f;
    const float dot = wave_sum(p * dp);
    if (lane < nt) {
      s[i][lane] = p;
      ds[i][lane] = p * (dp - dot);       // dS = P ∘ (dP − rowsum(dP ∘ P))
    }
  }
  __syncthreads();
  // position-table gradient of this (image, window, head)
  for (int e = threadIdx.x; e < nt * nt; e += 256) {
    const int i = e / nt, j = e % nt;
    const int w = a.w;
    const int idx = ((j / w) - (i / w) + w - 1) * (2 * w - 1) + ((j % w) - (i % w) + w - 1);
    atomicAdd(&dtab[idx], ds[i][j]);
  }
  // dV = Pᵀ·dO ; dQ = dS·K·scale ; dK = dSᵀ·Q·scale
  for (int e = threadIdx.x; e < nt * a.hd; e += 256) {
    const int t = e / a.hd, d = e % a.hd;
    float dv = 0.f, dq = 0.f, dk = 0.f;
    for (int j = 0; j < nt; ++j) {
      dv = fmaf(s[j][t], go[j][d], dv);
      dq = fmaf(ds[t][j], k[j][d], dq);
      dk = fmaf(ds[j][t], q[j][d], dk);
    }
    T* base = dqkv + wa_token_off(a, b, gy, gx, t) * (3 * C) + h * a.hd + d;
    base[0] = from_f32<T>(dq * a.scale);
    base[C] = from_f32<T>(dk * a.scale);
    base[2 * C] = from_f32<T>(dv);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < ntab; e += 256) dpos_part[(size_t)blockIdx.x * ntab + e] = dtab[e];
}

static int wa_check(int B, int H, int W, int heads, int hd, int w, int shift) {
  PFR_CHECK_ARG(w * w <= WA_MAXT && hd <= WA_MAXD && (2 * w - 1) * (2 * w - 1) <= 256, "window attention: window %d / head_dim %d too large", w, hd);
  PFR_CHECK_ARG(H % w == 0 && W % w == 0 && shift >= 0 && shift < w && B > 0 && heads > 0, "window attention: bad geometry");
  return PFR_OK;
}

extern "C" int pfr_window_attn_fwd(const void* qkv, const float* pos, void* out, int dtype, int B, int H, int W, int heads,
                                   int head_dim, int window, int shift, float scale, hipStream_t st) {
  PFR_CHECK_ARG(qkv && pos && out, "pfr_window_attn_fwd: null pointer");
  if (int rc = wa_check(B, H, W, heads, head_dim, window, shift)) return rc;
  WinAttn a{B, H, W, heads, head_dim, window, shift, scale};
  const dim3 grid((unsigned)(B * (H / window) * (W / window) * heads));
  if (dtype == PFR_BF16) hipLaunchKernelGGL(window_attn_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)qkv, pos, (bf16_t*)out, a);
  else hipLaunchKernelGGL(window_attn_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)qkv, pos, (float*)out, a);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// dpos_part: fp32 [B·(H/w)·(W/w)·heads][(2w−1)²]
extern "C" int pfr_window_attn_bwd(const void* qkv, const float* pos, const void* dout, void* dqkv, float* dpos_part, int dtype,
                                   int B, int H, int W, int heads, int head_dim, int window, int shift, float scale,
                                   hipStream_t st) {
  PFR_CHECK_ARG(qkv && pos && dout && dqkv && dpos_part, "pfr_window_attn_bwd: null pointer");
  if (int rc = wa_check(B, H, W, heads, head_dim, window, shift)) return rc;
  WinAttn a{B, H, W, heads, head_dim, window, shift, scale};
  const dim3 grid((unsigned)(B * (H / window) * (W / window) * heads));
  if (dtype == PFR_BF16) hipLaunchKernelGGL(window_attn_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)qkv, pos, (const bf16_t*)dout, (bf16_t*)dqkv, dpos_part, a);
  else hipLaunchKernelGGL(window_attn_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)qkv, pos, (const float*)dout, (float*)dqkv, dpos_part, a);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ------------------------------------------------------------------------------------------------ weight un-permute
// fp32 [N][HW][Cp] (NHWC, padded channels) → [N][C][HW] (the layout of nn.Unfold-ordered Linear weights): the inverse of
// pfr_nchw_to_nhwc, used for the patch-merging weight gradient.
__global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int HW, int Cp,
                                        int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * C * HW) return;
  const int pix = (int)(i % HW);
  const int c = (int)((i / HW) % C);
  const size_t n = i / ((size_t)HW * C);
  const float v = x[(n * HW + pix) * Cp + c];
  y[i] = accumulate ? y[i] + v : v;
}
extern "C" int pfr_nhwc_to_nchw_f32(const float* x, float* y, int N, int C, int HW, int Cp, int accumulate, hipStream_t st) {
  PFR_CHECK_ARG(x && y && Cp >= C, "pfr_nhwc_to_nchw_f32: bad args");
  const size_t n = (size_t)N * C * HW;
  hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, N, C, HW, Cp, accumulate);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
