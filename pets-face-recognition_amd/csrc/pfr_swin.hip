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

// bias(+mask) table of one attention block: tab[variant][i][j], variant = 2*(last window row) + (last window column),
// rows padded to ntp = ceil4(w²) with −inf outside the w² x w² block (one tiny launch per block and step instead of integer
// divisions per score element in every workgroup).
__global__ void window_bias_table_kernel(const float* __restrict__ pos, float* __restrict__ tab, int w, int shift) {
  const int nt = w * w, ntp = (nt + 3) & ~3;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 4 * ntp * ntp) return;
  const int var = e / (ntp * ntp), i = (e / ntp) % ntp, j = e % ntp;
  float b = -INFINITY;
  if (i < nt && j < nt) {
    const int yi = i / w, xi = i % w, yj = j / w, xj = j % w;
    b = pos[(yj - yi + w - 1) * (2 * w - 1) + (xj - xi + w - 1)];
    if (shift) {
      if ((var & 2) && ((yi >= w - shift) != (yj >= w - shift))) b = -INFINITY;
      if ((var & 1) && ((xi >= w - shift) != (xj >= w - shift))) b = -INFINITY;
    }
  }
  tab[e] = b;
}

__device__ __forceinline__ size_t wa_token_off(const WinAttn& a, int b, int gy, int gx, int t) {
  int y = gy * a.w + t / a.w + a.shift, x = gx * a.w + t % a.w + a.shift;
  if (y >= a.H) y -= a.H;
  if (x >= a.W) x -= a.W;
  return ((size_t)b * a.H + y) * a.W + x;
}

// LDS strides (floats): multiples of 4 so that 4 consecutive d / j values are one ds_read_b128
#define WA_SD 36   // q/k/v/go rows: WA_MAXD + 4
#define WA_SS 68   // score rows: WA_MAXT + 4

// register-blocked helpers: each thread owns a 4x4 block of an output matrix and streams the reduction dimension with
// 128-bit LDS reads (8 reads per 64 FMAs instead of 2 reads per FMA).
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <typename T>
__global__ __launch_bounds__(256) void window_attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ tab,
                                                              T* __restrict__ out, WinAttn a) {
  __shared__ __attribute__((aligned(16))) float q[WA_MAXT * WA_SD], k[WA_MAXT * WA_SD], v[WA_MAXT * WA_SD];
  __shared__ __attribute__((aligned(16))) float s[WA_MAXT * WA_SS];
  const int nwh = a.H / a.w, nww = a.W / a.w, nt = a.w * a.w, C = a.heads * a.hd;
  int bid = blockIdx.x;
  const int h = bid % a.heads; bid /= a.heads;
  const int gx = bid % nww; bid /= nww;
  const int gy = bid % nwh;
  const int b = bid / nwh;
  const int ntp = (nt + 3) & ~3, hdp = (a.hd + 3) & ~3;
  constexpr int KPL = DT<T>::KPACK;                      // elements per 16-byte chunk
  const int cpt = (a.hd + KPL - 1) / KPL;                // chunks per token and operand (hd is a multiple of KPL)
  const float* btab = tab + (size_t)(((gy == nwh - 1) ? 2 : 0) + ((gx == nww - 1) ? 1 : 0)) * ntp * ntp;
  for (int e = threadIdx.x; e < ntp * cpt; e += 256) {    // 16-byte loads; rows nt..ntp-1 are zero padding
    const int t = e / cpt, d = (e % cpt) * KPL;
    float fq[KPL], fk[KPL], fv[KPL];
#pragma unroll
    for (int u = 0; u < KPL; ++u) { fq[u] = 0.f; fk[u] = 0.f; fv[u] = 0.f; }
    if (t < nt) {
      const T* base = qkv + wa_token_off(a, b, gy, gx, t) * (3 * C) + h * a.hd + d;
      Chunk<T>::unpack(ld16(base), fq);
      Chunk<T>::unpack(ld16(base + C), fk);
      Chunk<T>::unpack(ld16(base + 2 * C), fv);
    }
#pragma unroll
    for (int u = 0; u < KPL; ++u) { q[t * WA_SD + d + u] = fq[u]; k[t * WA_SD + d + u] = fk[u]; v[t * WA_SD + d + u] = fv[u]; }
  }
  __syncthreads();
  // S = q·kᵀ·scale + bias(+mask): 4x4 blocks
  const int nb = ntp / 4;
  for (int blk = threadIdx.x; blk < nb * nb; blk += 256) {
    const int i0 = (blk / nb) * 4, j0 = (blk % nb) * 4;
    float acc[4][4] = {};
    for (int d = 0; d < hdp; d += 4) {
      f32x4 qa[4], kb[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) { qa[x] = ld4(&q[(i0 + x) * WA_SD + d]); kb[x] = ld4(&k[(j0 + x) * WA_SD + d]); }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          acc[x][y] += qa[x][0] * kb[y][0] + qa[x][1] * kb[y][1] + qa[x][2] * kb[y][2] + qa[x][3] * kb[y][3];
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        s[(i0 + x) * WA_SS + j0 + y] = acc[x][y] * a.scale + btab[(i0 + x) * ntp + j0 + y];   // −inf outside w² x w²
      }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < ntp; i += 4) {
    const float x = (lane < nt && i < nt) ? s[i * WA_SS + lane] : -INFINITY;
    const float m = wave_max(x);
    const float ex = (lane < nt && i < nt) ? __expf(x - m) : 0.f;
    const float sum = wave_sum(ex);
    if (lane < ntp) s[i * WA_SS + lane] = (i < nt && lane < nt) ? ex / sum : 0.f;
  }
  __syncthreads();
  // O = P·V: 4 (tokens) x 4 (d) blocks
  const int ndb = hdp / 4;
  for (int blk = threadIdx.x; blk < nb * ndb; blk += 256) {
    const int i0 = (blk / ndb) * 4, d0 = (blk % ndb) * 4;
    f32x4 acc[4] = {};
    for (int j = 0; j < ntp; j += 4) {
      f32x4 pr[4], vv[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) { pr[x] = ld4(&s[(i0 + x) * WA_SS + j]); vv[x] = ld4(&v[(j + x) * WA_SD + d0]); }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x] += pr[x][y] * vv[y];
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int i = i0 + x;
      if (i < nt) {
        T* o = out + wa_token_off(a, b, gy, gx, i) * C + h * a.hd + d0;
#pragma unroll
        for (int y = 0; y < 4; ++y)
          if (d0 + y < a.hd) o[y] = from_f32<T>(acc[x][y]);
      }
    }
  }
}

// backward: recomputes the probabilities; writes dqkv and this workgroup's partial of the position-table gradient
// (dpos_part [nblocks][(2w−1)²], summed afterwards by pfr_colsum → deterministic).
template <typename T>
__global__ __launch_bounds__(256) void window_attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ tab,
                                                              const T* __restrict__ dout, T* __restrict__ dqkv,
                                                              float* __restrict__ dpos_part, WinAttn a) {
  __shared__ __attribute__((aligned(16))) float q[WA_MAXT * WA_SD], k[WA_MAXT * WA_SD], v[WA_MAXT * WA_SD], go[WA_MAXT * WA_SD];
  __shared__ __attribute__((aligned(16))) float s[WA_MAXT * WA_SS], ds[WA_MAXT * WA_SS];  // P, dS
  __shared__ float dtab[256];
  const int nwh = a.H / a.w, nww = a.W / a.w, nt = a.w * a.w, C = a.heads * a.hd;
  const int ntab = (2 * a.w - 1) * (2 * a.w - 1);
  int bid = blockIdx.x;
  const int h = bid % a.heads; bid /= a.heads;
  const int gx = bid % nww; bid /= nww;
  const int gy = bid % nwh;
  const int b = bid / nwh;
  const int ntp = (nt + 3) & ~3, hdp = (a.hd + 3) & ~3;
  for (int e = threadIdx.x; e < ntab; e += 256) dtab[e] = 0.f;
  constexpr int KPL = DT<T>::KPACK;
  const int cpt = (a.hd + KPL - 1) / KPL;
  const float* btab = tab + (size_t)(((gy == nwh - 1) ? 2 : 0) + ((gx == nww - 1) ? 1 : 0)) * ntp * ntp;
  for (int e = threadIdx.x; e < ntp * cpt; e += 256) {
    const int t = e / cpt, d = (e % cpt) * KPL;
    float fq[KPL], fk[KPL], fv[KPL], fg[KPL];
#pragma unroll
    for (int u = 0; u < KPL; ++u) { fq[u] = 0.f; fk[u] = 0.f; fv[u] = 0.f; fg[u] = 0.f; }
    if (t < nt) {
      const size_t tok = wa_token_off(a, b, gy, gx, t);
      const T* base = qkv + tok * (3 * C) + h * a.hd + d;
      Chunk<T>::unpack(ld16(base), fq);
      Chunk<T>::unpack(ld16(base + C), fk);
      Chunk<T>::unpack(ld16(base + 2 * C), fv);
      Chunk<T>::unpack(ld16(dout + tok * C + h * a.hd + d), fg);
    }
#pragma unroll
    for (int u = 0; u < KPL; ++u) {
      q[t * WA_SD + d + u] = fq[u]; k[t * WA_SD + d + u] = fk[u]; v[t * WA_SD + d + u] = fv[u]; go[t * WA_SD + d + u] = fg[u];
    }
  }
  __syncthreads();
  const int nb = ntp / 4;
  for (int blk = threadIdx.x; blk < nb * nb; blk += 256) {
    const int i0 = (blk / nb) * 4, j0 = (blk % nb) * 4;
    float acc[4][4] = {}, dpa[4][4] = {};
    for (int d = 0; d < hdp; d += 4) {
      f32x4 qa[4], kb[4], ga[4], vb[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        qa[x] = ld4(&q[(i0 + x) * WA_SD + d]); kb[x] = ld4(&k[(j0 + x) * WA_SD + d]);
        ga[x] = ld4(&go[(i0 + x) * WA_SD + d]); vb[x] = ld4(&v[(j0 + x) * WA_SD + d]);
      }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          acc[x][y] += qa[x][0] * kb[y][0] + qa[x][1] * kb[y][1] + qa[x][2] * kb[y][2] + qa[x][3] * kb[y][3];
          dpa[x][y] += ga[x][0] * vb[y][0] + ga[x][1] * vb[y][1] + ga[x][2] * vb[y][2] + ga[x][3] * vb[y][3];   // dP = dO·Vᵀ
        }
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const int i = i0 + x, j = j0 + y;
        const bool ok = i < nt && j < nt;
        s[i * WA_SS + j] = acc[x][y] * a.scale + btab[i * ntp + j];
        ds[i * WA_SS + j] = ok ? dpa[x][y] : 0.f;
      }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < ntp; i += 4) {
    const bool okr = i < nt;
    const float x = (lane < nt && okr) ? s[i * WA_SS + lane] : -INFINITY;
    const float m = wave_max(x);
    const float ex = (lane < nt && okr) ? __expf(x - m) : 0.f;
    const float sum = wave_sum(ex);
    const float p = okr ? ex / sum : 0.f;
    const float dp = (lane < nt && okr) ? ds[i * WA_SS + lane] : 0.f;
    const float dot = wave_sum(p * dp);
    if (lane < ntp) {
      const float dsv = p * (dp - dot);     // dS = P ∘ (dP − rowsum(dP ∘ P))
      s[i * WA_SS + lane] = p;
      ds[i * WA_SS + lane] = dsv;
    }
  }
  __syncthreads();
  // position-table gradient of this (image, window, head)
  for (int e = threadIdx.x; e < nt * nt; e += 256) {
    const int i = e / nt, j = e % nt;
    const int w = a.w;
    const int idx = ((j / w) - (i / w) + w - 1) * (2 * w - 1) + ((j % w) - (i % w) + w - 1);
    atomicAdd(&dtab[idx], ds[i * WA_SS + j]);
  }
  // dV = Pᵀ·dO ; dQ = dS·K·scale ; dK = dSᵀ·Q·scale : 4 (tokens) x 4 (d) blocks, reduction over the other token index
  const int ndb = hdp / 4;
  for (int blk = threadIdx.x; blk < nb * ndb; blk += 256) {
    const int t0 = (blk / ndb) * 4, d0 = (blk % ndb) * 4;
    f32x4 dv[4] = {}, dq[4] = {}, dk[4] = {};
    for (int j = 0; j < ntp; j += 4) {
      f32x4 gj[4], kj[4], qj[4];
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        gj[y] = ld4(&go[(j + y) * WA_SD + d0]); kj[y] = ld4(&k[(j + y) * WA_SD + d0]); qj[y] = ld4(&q[(j + y) * WA_SD + d0]);
      }
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const f32x4 pT = ld4(&s[(j + y) * WA_SS + t0]);     // P[j+y][t0..t0+3]
        const f32x4 dT = ld4(&ds[(j + y) * WA_SS + t0]);    // dS[j+y][t0..t0+3]
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          dv[x] += pT[x] * gj[y];
          dk[x] += dT[x] * qj[y];
        }
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const f32x4 drow = ld4(&ds[(t0 + x) * WA_SS + j]);  // dS[t0+x][j..j+3]
#pragma unroll
        for (int y = 0; y < 4; ++y) dq[x] += drow[y] * kj[y];
      }
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int t = t0 + x;
      if (t < nt) {
        T* base = dqkv + wa_token_off(a, b, gy, gx, t) * (3 * C) + h * a.hd + d0;
#pragma unroll
        for (int y = 0; y < 4; ++y)
          if (d0 + y < a.hd) {
            base[y] = from_f32<T>(dq[x][y] * a.scale);
            base[C + y] = from_f32<T>(dk[x][y] * a.scale);
            base[2 * C + y] = from_f32<T>(dv[x][y]);
          }
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < ntab; e += 256) dpos_part[(size_t)blockIdx.x * ntab + e] = dtab[e];
}

static int wa_check(int B, int H, int W, int heads, int hd, int w, int shift) {
  PFR_CHECK_ARG(w * w <= WA_MAXT && hd <= WA_MAXD && (2 * w - 1) * (2 * w - 1) <= 256, "window attention: window %d / head_dim %d too large", w, hd);
  PFR_CHECK_ARG(H % w == 0 && W % w == 0 && shift >= 0 && shift < w && B > 0 && heads > 0, "window attention: bad geometry");
  return PFR_OK;
}

extern "C" long pfr_window_bias_table_floats(int window) {
  const int ntp = (window * window + 3) & ~3;
  return 4L * ntp * ntp;
}
// tab: fp32 [4][ceil4(w²)][ceil4(w²)], recomputed whenever pos changes (once per block and step)
extern "C" int pfr_window_bias_table(const float* pos, float* tab, int window, int shift, hipStream_t st) {
  PFR_CHECK_ARG(pos && tab && window * window <= WA_MAXT, "pfr_window_bias_table: bad args");
  const long n = pfr_window_bias_table_floats(window);
  hipLaunchKernelGGL(window_bias_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pos, tab, window, shift);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

extern "C" int pfr_window_attn_fwd(const void* qkv, const float* pos, void* out, int dtype, int B, int H, int W, int heads,
                                   int head_dim, int window, int shift, float scale, hipStream_t st) {
  PFR_CHECK_ARG(qkv && pos && out, "pfr_window_attn_fwd: null pointer");
  PFR_CHECK_ARG(head_dim % (dtype == PFR_BF16 ? 8 : 4) == 0, "pfr_window_attn_fwd: head_dim must be a multiple of the 16-byte chunk");
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
  PFR_CHECK_ARG(head_dim % (dtype == PFR_BF16 ? 8 : 4) == 0, "pfr_window_attn_bwd: head_dim must be a multiple of the 16-byte chunk");
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
