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
#include "pfr_mma.h"
#include <stdlib.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));

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

// Chunked LayerNorm (C a multiple of the 16-byte chunk, ≤ 256 chunks): a row is owned by G = 16 / 32 / 64 lanes (64/G rows per
// wave), each lane keeps CPL chunks of ITS channels in registers — one 16-byte load per chunk instead of three 2-byte
// passes — row statistics are xor-shuffles inside the G-lane group, γ/β stay in registers over a grid-stride row loop.
struct LnGeom { int G, cpl, cpr; };
static bool ln_geom(int C, int kp, LnGeom* g) {
  if (C % kp) return false;
  g->cpr = C / kp;
  g->G = g->cpr <= 16 ? 16 : (g->cpr <= 32 ? 32 : 64);
  g->cpl = (g->cpr + g->G - 1) / g->G;
  return g->cpl <= 4;
}
static int ln_grid(long rows, int G, int rb) {   // one batch of rb x (256 / G) rows per workgroup
  const long per = (long)rb * 4 * (64 / G);
  const long nb = (rows + per - 1) / per;
  return (int)(nb < 1 ? 1 : nb);
}

// Sum over the G lanes that own a row, every lane receiving the total.  The 16-lane part is four DPP adds (quad_perm [1,0,3,2] and
// [2,3,0,1], row_half_mirror, row_mirror: the same pairings, hence the same bits, as the xor-1/2/4/8 butterfly); with G a run-time
// value it was a loop of ds_bpermute + lgkmcnt(0) round trips, 8-12 per row and the longest stretch between a row's loads and its
// stores (round 5: 53 -> ? us for the 401 408 x 96 forward).
template <int CTRL>
__device__ __forceinline__ float ln_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int G>
__device__ __forceinline__ float ln_group_sum(float s) {
  s += ln_dpp<0xB1>(s);
  s += ln_dpp<0x4E>(s);
  s += ln_dpp<0x141>(s);
  s += ln_dpp<0x140>(s);
  if constexpr (G >= 32) s += __shfl_xor(s, 16, 64);
  if constexpr (G >= 64) s += __shfl_xor(s, 32, 64);
  return s;
}

// Non-temporal hint on the row loads of the LayerNorm kernels (A/B builds: -DPFR_LN_NT=1 forward, -DPFR_LNB_NT=2 backward = the `nt`
// cache-policy bit).  Measured (round 5, profiles/r05_ln.txt): COLD the hint is worth 1.4x (forward 44.5 -> 31.9 us at 401 408 x 96, 4.9 TB/s,
// faster than a device copy), but inside the Swin-T step, where every operand was written by the kernel before, it costs 0.5-0.8 % of
// the step (the lines are dropped from L2 / MALL before their next reader) — so both default to plain loads.
#ifndef PFR_LNB_NT
#define PFR_LNB_NT 0
#endif
#ifndef PFR_LN_NT
#define PFR_LN_NT 0
#endif
static constexpr bool pfr_ln_nt = PFR_LN_NT != 0;
template <typename T, int CPL, int G, int RB>
__global__ __launch_bounds__(256, (CPL == 1 ? (RB == 8 ? 6 : 8) : (CPL == 2 ? 5 : 2))) void layernorm_fwd2_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, T* __restrict__ y,
                                                             float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                             int C, float eps, int cpr) {
  constexpr int KP = DT<T>::KPACK;
  // One batch of RB x (256 / G) consecutive rows per workgroup, RB rows in flight per lane group, no row loop.  Round 5 (53.3 → ? µs at
  // 401 408 x 96, 18.3 → ? at 6 272 x 768):
  //  * the former grid-stride loop of 2048 resident workgroups ran ceil(3.06) = 4 latency-bound iterations on the stage-1 tensor, the
  //    last one 6 % full; with one batch per workgroup the hardware scheduler balances the tail;
  //  * BUFFER addressing: a 32-bit byte offset per lane + a wave-uniform offset per row of the batch instead of RB 64-bit addresses
  //    each for x, y, mean and rstd (the kernel spilled at eight waves per SIMD), and the descriptor's range check instead of clamps
  //    and branches: rows past the end and the lanes past a row's last chunk read zeros and their stores are dropped — straight-line code;
  //  * γ / β go to LDS AFTER the x loads are issued (as 2 x CPL x KP registers per lane they cost two of the eight waves per SIMD).
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (G - 1), grp = lane / G;
  constexpr int rpw = 64 / G;
  const uint32_t rowb = (uint32_t)C * (uint32_t)sizeof(T);
  const uint32_t total = (uint32_t)rows * rowb;   // (host: < 2^31)
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)total, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)total, 0x00020000);
  const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(mean, 0, mean ? (int)(rows * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(rstd, 0, rstd ? (int)(rows * 4) : 0, 0x00020000);
  bool okc[CPL];
  uint32_t cb[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = sub + G * k;
    okc[k] = c < cpr;
    cb[k] = okc[k] ? (uint32_t)c * 16u : 0x7ffffff0u;   // past the descriptor's range: loads give 0, stores are dropped
  }
  const float invC = 1.f / (float)C;
  constexpr uint32_t stride = 4u * rpw;                  // rows between two rows of a lane group's batch: the workgroup's rows are contiguous
  const uint32_t strideb = stride * rowb;
  const uint32_t row = blockIdx.x * (RB * stride) + wave * rpw + grp;
  const uint32_t voff = row * rowb;
  {
    u32x4 raw[RB][CPL];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int k = 0; k < CPL; ++k)
        raw[b][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(voff + cb[k]), (int)(b * strideb), pfr_ln_nt ? 2 : 0));
    __shared__ __attribute__((aligned(16))) float sgam[2048], sbet[2048];
    for (int c = threadIdx.x; c < C; c += 256) {
      sgam[c] = gamma[c];
      sbet[c] = beta[c];
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);   // every load of the batch is issued before the first is consumed (hipcc hoists row 0's unpack, and its wait, above the other loads)
    // statistics of every row in flight first, all stores afterwards
    float mus[RB], rss[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      float f[CPL][KP];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        Chunk<T>::unpack(raw[b][k], f[k]);
#pragma unroll
        for (int e = 0; e < KP; ++e) s += f[k][e];
      }
      s = ln_group_sum<G>(s);
      const float mu = s * invC;
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < CPL; ++k)
#pragma unroll
        for (int e = 0; e < KP; ++e) {
          const float d = okc[k] ? f[k][e] - mu : 0.f;
          v = fmaf(d, d, v);
        }
      v = ln_group_sum<G>(v);
      mus[b] = mu;
      rss[b] = rsqrtf(v * invC + eps);
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const float mu = mus[b], rs = rss[b];
#pragma unroll
      for (int k = 0; k < CPL; ++k)   // (opaque copy: without it the unpacked floats of all RB rows stay live from the statistics phase — spills)
        asm volatile("" : "+v"(raw[b][k]));
      const uint32_t moff = sub == 0 ? row * 4u : 0x7ffffff0u;   // one lane per row stores; the others' are dropped by the range check
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, mu), mr, (int)moff, (int)(b * stride * 4u), 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, rs), rr, (int)moff, (int)(b * stride * 4u), 0);
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        float f[KP];
        Chunk<T>::unpack(raw[b][k], f);
        const float* gp = sgam + (okc[k] ? sub + G * k : 0) * KP;   // (lanes past the row: γ[0..], a defined value — 0 x uninitialised LDS could be NaN; their store is dropped)
        const float* bp = sbet + (okc[k] ? sub + G * k : 0) * KP;
#pragma unroll
        for (int e = 0; e < KP; ++e) f[e] = fmaf((f[e] - mu) * rs, gp[e], bp[e]);
        buffer_store_b128_sync(Chunk<T>::pack(f), yr, voff + cb[k], b * strideb);
      }
    }
  }
}

template <typename T>
static bool ln_fwd2_launch(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, long rows,
                           int C, float eps, hipStream_t st) {
  LnGeom g;
  if (!ln_geom(C, DT<T>::KPACK, &g)) return false;
  if ((double)rows * C * sizeof(T) >= 2147483648.0 * 0.75) return false;   // 32-bit buffer offsets (the batch's last row may lie past the end)
  const int rbk = pfr_knob(KNOB_LN_RB);
  const int rb = g.cpl > 1 ? 2 : (rbk == 6 || rbk == 8 ? rbk : 4);
  const dim3 grid(ln_grid(rows, g.G, rb));
#define PFR_LNF(K, GG, R)                                                                                               \
  if (g.cpl == K && g.G == GG && rb == R) {                                                                             \
    hipLaunchKernelGGL((layernorm_fwd2_kernel<T, K, GG, R>), grid, dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, C, eps, g.cpr); \
    return true;                                                                                                        \
  }
  PFR_LNF(1, 16, 4) PFR_LNF(1, 32, 4) PFR_LNF(1, 64, 4) PFR_LNF(1, 16, 6) PFR_LNF(1, 32, 6) PFR_LNF(1, 64, 6) PFR_LNF(1, 16, 8) PFR_LNF(1, 32, 8)
  PFR_LNF(1, 64, 8) PFR_LNF(2, 64, 2) PFR_LNF(3, 64, 2) PFR_LNF(4, 64, 2)
#undef PFR_LNF
  return false;
}

extern "C" int pfr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 int dtype, long rows, int C, float eps, hipStream_t st) {
  PFR_CHECK_ARG(x && gamma && beta && y, "pfr_layernorm_fwd: null pointer");
  if (dtype == PFR_BF16 ? ln_fwd2_launch<bf16_t>(x, gamma, beta, y, mean, rstd, rows, C, eps, st)
                        : ln_fwd2_launch<float>(x, gamma, beta, y, mean, rstd, rows, C, eps, st)) {
    PFR_CHECK_LAUNCH();
    return PFR_OK;
  }
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

template <typename T, int CPL, int G>
__global__ __launch_bounds__(256) void layernorm_bwd2_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const T* __restrict__ dres,
                                                             T* __restrict__ dx, float* __restrict__ part, float* __restrict__ dxsum, long rows,
                                                             int C, int cpr) {
  constexpr int KP = DT<T>::KPACK;
  extern __shared__ float sh[];  // [4 waves][3][C]   (third row set: column sums of the stored dx, optional)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (G - 1), grp = lane / G, rpw = 64 / G;
  float gm[CPL][KP], ag[CPL][KP], ab[CPL][KP], ad[CPL][KP];
  bool okc[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = sub + G * k;
    okc[k] = c < cpr;
#pragma unroll
    for (int e = 0; e < KP; ++e) {
      gm[k][e] = okc[k] ? gamma[c * KP + e] : 0.f;
      ag[k][e] = 0.f;
      ab[k][e] = 0.f;
      ad[k][e] = 0.f;
    }
  }
  const float invC = 1.f / (float)C;
  const long stride = (long)gridDim.x * 4 * rpw;
  for (long row = ((long)blockIdx.x * 4 + wave) * rpw + grp; row < rows; row += stride) {
    const float mu = mean[row], rs = rstd[row];
    float xh[CPL][KP], dv[CPL][KP];
    float a = 0.f, b = 0.f;
    u32x4 rx[CPL], rd[CPL], rr[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {     // all operands of the row requested together (clamped chunk for lanes beyond the row)
      const size_t off = (size_t)row * C + (okc[k] ? sub + G * k : 0) * KP;
      rx[k] = ld16_nt(x + off);   // last uses of the saved activation and of dy
      rd[k] = ld16_nt(dy + off);
      if (dres) rr[k] = ld16_nt(dres + off);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      if (okc[k]) {
        Chunk<T>::unpack(rx[k], xh[k]);
        Chunk<T>::unpack(rd[k], dv[k]);
      }
#pragma unroll
      for (int e = 0; e < KP; ++e) {
        xh[k][e] = okc[k] ? (xh[k][e] - mu) * rs : 0.f;
        if (!okc[k]) dv[k][e] = 0.f;
        const float g = dv[k][e] * gm[k][e];
        a += g;
        b = fmaf(g, xh[k][e], b);
      }
    }
    a = ln_group_sum<G>(a);
    b = ln_group_sum<G>(b);
    a *= invC;
    b *= invC;
#pragma unroll
    for (int k = 0; k < CPL; ++k)
      if (okc[k]) {
        const size_t off = (size_t)row * C + (sub + G * k) * KP;
        float r[KP], o[KP];
        if (dres) Chunk<T>::unpack(rr[k], r);
#pragma unroll
        for (int e = 0; e < KP; ++e) {
          float v = rs * (dv[k][e] * gm[k][e] - a - xh[k][e] * b);
          if (dres) v += r[e];
          o[e] = v;
          ag[k][e] = fmaf(dv[k][e], xh[k][e], ag[k][e]);
          ab[k][e] += dv[k][e];
        }
        const u32x4 po = Chunk<T>::pack(o);
        st16(dx + off, po);
        if (dxsum) {   // Σ rows of dx AS STORED: the bias gradient of the layer that produced this LayerNorm's input
          Chunk<T>::unpack(po, o);
#pragma unroll
          for (int e = 0; e < KP; ++e) ad[k][e] += o[e];
        }
      }
  }
  // lane groups of the wave own the same channels: fold them (xor offsets ≥ G), then the 4 waves through LDS
#pragma unroll
  for (int k = 0; k < CPL; ++k)
#pragma unroll
    for (int e = 0; e < KP; ++e)
      for (int o = G; o < 64; o <<= 1) {
        ag[k][e] += __shfl_xor(ag[k][e], o, 64);
        ab[k][e] += __shfl_xor(ab[k][e], o, 64);
        if (dxsum) ad[k][e] += __shfl_xor(ad[k][e], o, 64);
      }
  if (grp == 0) {
#pragma unroll
    for (int k = 0; k < CPL; ++k)
      if (okc[k]) {
#pragma unroll
        for (int e = 0; e < KP; ++e) {
          sh[(wave * 3 + 0) * C + (sub + G * k) * KP + e] = ag[k][e];
          sh[(wave * 3 + 1) * C + (sub + G * k) * KP + e] = ab[k][e];
          sh[(wave * 3 + 2) * C + (sub + G * k) * KP + e] = ad[k][e];
        }
      }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < (dxsum ? 3 : 2) * C; c += 256) {
    const int q = c / C, cc = c % C;
    const float v = sh[(0 * 3 + q) * C + cc] + sh[(1 * 3 + q) * C + cc] + sh[(2 * 3 + q) * C + cc] + sh[(3 * 3 + q) * C + cc];
    if (q < 2) part[((size_t)q * gridDim.x + blockIdx.x) * C + cc] = v;
    else dxsum[(size_t)blockIdx.x * C + cc] = v;
  }
}

// The same pass with BUFFER addressing and RB rows in flight per lane group (tensors below 1.5 GB; round 5).  The row loop of
// layernorm_bwd2_kernel is latency-bound — 16 resident waves per CU x one row x three 16-byte loads per lane ≈ 49 KB in flight per
// CU per ~2.7 µs iteration = the 4.6 TB/s it measures at 401 408 x 96 — so this one keeps two rows per lane group in flight at the same
// register budget: 32-bit lane offsets + wave-uniform row offsets instead of 64-bit addresses per operand, γ in LDS, the range check
// of the descriptor instead of row / chunk predicates (rows past the end and lanes past the row read zeros — they add 0 to every sum —
// and their stores are dropped), a missing residual = an empty descriptor.  Every row's result is computed BEFORE the first store of
// the iteration is issued (the stores are inline asm: a compiler-counted wait for a younger load would also wait for them).
// Per lane the rows are visited in the same order as before: dγ / dβ / Σdx partials are bit-identical to layernorm_bwd2_kernel's.
template <typename T, int CPL, int G>
__global__ __launch_bounds__(256, (CPL == 1 ? 4 : (CPL == 2 ? 2 : 1))) void layernorm_bwd3_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const T* __restrict__ dres,
                                                             T* __restrict__ dx, float* __restrict__ part, float* __restrict__ dxsum, long rows,
                                                             int C, int cpr) {
  constexpr int KP = DT<T>::KPACK;
  constexpr int RB = CPL == 1 ? 2 : 1;
  extern __shared__ float sh[];  // [4 waves][3][C] partial sums at the end, then [C] γ
  float* const sgam = sh + 12 * C;
  for (int c = threadIdx.x; c < C; c += 256) sgam[c] = gamma[c];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (G - 1), grp = lane / G;
  constexpr int rpw = 64 / G;
  const uint32_t rowb = (uint32_t)C * (uint32_t)sizeof(T);
  const uint32_t total = (uint32_t)rows * rowb;
  const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dy), 0, (int)total, 0x00020000);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)total, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dres ? dres : dy), 0, dres ? (int)total : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t dxr = __builtin_amdgcn_make_buffer_rsrc(dx, 0, (int)total, 0x00020000);
  const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mean), 0, (int)(rows * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rstd), 0, (int)(rows * 4), 0x00020000);
  float ag[CPL][KP], ab[CPL][KP], ad[CPL][KP];
  bool okc[CPL];
  uint32_t cb[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = sub + G * k;
    okc[k] = c < cpr;
    cb[k] = okc[k] ? (uint32_t)c * 16u : 0x7ffffff0u;
#pragma unroll
    for (int e = 0; e < KP; ++e) {
      ag[k][e] = 0.f;
      ab[k][e] = 0.f;
      ad[k][e] = 0.f;
    }
  }
  const float invC = 1.f / (float)C;
  const uint32_t stride = gridDim.x * 4u * rpw;
  const uint32_t strideb = __builtin_amdgcn_readfirstlane(stride * rowb);
  const uint32_t nit = ((uint32_t)rows + RB * stride - 1) / (RB * stride);
  uint32_t row = (blockIdx.x * 4u + wave) * rpw + grp;
  uint32_t voff = row * rowb;
  for (uint32_t it = 0; it < nit; ++it, row += RB * stride, voff += RB * stride * rowb) {
    u32x4 rx[RB][CPL], rd[RB][CPL], rs_[RB][CPL], po[RB][CPL];
    float mus[RB], rss[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      mus[b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mr, (int)(row * 4u), (int)(b * stride * 4u), 0));
      rss[b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sr, (int)(row * 4u), (int)(b * stride * 4u), 0));
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        rx[b][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(voff + cb[k]), (int)(b * strideb), PFR_LNB_NT));
        rd[b][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(dyr, (int)(voff + cb[k]), (int)(b * strideb), PFR_LNB_NT));
        rs_[b][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(voff + cb[k]), (int)(b * strideb), PFR_LNB_NT));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const float mu = mus[b], rs = rss[b];
      float xh[CPL][KP], dv[CPL][KP];
      float a = 0.f, bsum = 0.f;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        Chunk<T>::unpack(rx[b][k], xh[k]);
        Chunk<T>::unpack(rd[b][k], dv[k]);
        const float* gp = sgam + (okc[k] ? sub + G * k : 0) * KP;
#pragma unroll
        for (int e = 0; e < KP; ++e) {
          xh[k][e] = okc[k] ? (xh[k][e] - mu) * rs : 0.f;
          const float g = dv[k][e] * gp[e];
          a += g;
          bsum = fmaf(g, xh[k][e], bsum);
        }
      }
      a = ln_group_sum<G>(a);
      bsum = ln_group_sum<G>(bsum);
      a *= invC;
      bsum *= invC;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        float r[KP], o[KP];
        Chunk<T>::unpack(rs_[b][k], r);
        const float* gp = sgam + (okc[k] ? sub + G * k : 0) * KP;
#pragma unroll
        for (int e = 0; e < KP; ++e) {
          float v = rs * (dv[k][e] * gp[e] - a - xh[k][e] * bsum);
          if (dres) v += r[e];
          o[e] = okc[k] ? v : 0.f;
          ag[k][e] = fmaf(dv[k][e], xh[k][e], ag[k][e]);
          ab[k][e] += dv[k][e];
        }
        po[b][k] = Chunk<T>::pack(o);
        if (dxsum) {   // Σ rows of dx AS STORED: the bias gradient of the layer that produced this LayerNorm's input
          Chunk<T>::unpack(po[b][k], o);
#pragma unroll
          for (int e = 0; e < KP; ++e) ad[k][e] += o[e];
        }
      }
    }
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int k = 0; k < CPL; ++k) buffer_store_b128_sync(po[b][k], dxr, voff + cb[k], b * strideb);
  }
  // lane groups of the wave own the same channels: fold them (xor offsets ≥ G), then the 4 waves through LDS
#pragma unroll
  for (int k = 0; k < CPL; ++k)
#pragma unroll
    for (int e = 0; e < KP; ++e)
      for (int o = G; o < 64; o <<= 1) {
        ag[k][e] += __shfl_xor(ag[k][e], o, 64);
        ab[k][e] += __shfl_xor(ab[k][e], o, 64);
        if (dxsum) ad[k][e] += __shfl_xor(ad[k][e], o, 64);
      }
  if (grp == 0) {
#pragma unroll
    for (int k = 0; k < CPL; ++k)
      if (okc[k]) {
#pragma unroll
        for (int e = 0; e < KP; ++e) {
          sh[(wave * 3 + 0) * C + (sub + G * k) * KP + e] = ag[k][e];
          sh[(wave * 3 + 1) * C + (sub + G * k) * KP + e] = ab[k][e];
          sh[(wave * 3 + 2) * C + (sub + G * k) * KP + e] = ad[k][e];
        }
      }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < (dxsum ? 3 : 2) * C; c += 256) {
    const int q = c / C, cc = c % C;
    const float v = sh[(0 * 3 + q) * C + cc] + sh[(1 * 3 + q) * C + cc] + sh[(2 * 3 + q) * C + cc] + sh[(3 * 3 + q) * C + cc];
    if (q < 2) part[((size_t)q * gridDim.x + blockIdx.x) * C + cc] = v;
    else dxsum[(size_t)blockIdx.x * C + cc] = v;
  }
}

extern "C" int pfr_layernorm_bwd_blocks(long rows) {
  long nb = (rows + 15) / 16;   // >= 4 rows per wave: the row loop is latency-bound, so favour many resident waves
  if (nb > 1024) nb = 1024;     // (measured: 1024 workgroups 0.73 ms per Swin-T step, 2048 0.79, 512 0.87)  grid-stride kernels: the partial rows [2][nb][C] are summed by pfr_colsum afterwards
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

// 1 when the chunked kernel (which can also emit the column sums of dx) takes this geometry
extern "C" int pfr_layernorm_bwd_dxsum_ok(int dtype, int C) {
  LnGeom g;
  return ln_geom(C, dtype == PFR_BF16 ? 8 : 4, &g) ? 1 : 0;
}

// dxsum_part (optional, [pfr_layernorm_bwd_blocks(rows)][C]): per-workgroup column sums of dx as stored — the bias gradient of the
// Linear (or patch-merging conv) whose output this LayerNorm normalises comes out of this pass instead of a second read of dx
extern "C" int pfr_layernorm_bwd_dxsum(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                       const void* dres, void* dx, float* part, float* dxsum_part, int dtype, long rows, int C,
                                       hipStream_t st) {
  PFR_CHECK_ARG(dy && x && mean && rstd && gamma && dx && part, "pfr_layernorm_bwd: null pointer");
  const int nb = pfr_layernorm_bwd_blocks(rows);
  const int rpb = (int)((rows + nb - 1) / nb);
  {
    LnGeom g;
    const int kp = dtype == PFR_BF16 ? 8 : 4;
    if (ln_geom(C, kp, &g)) {
      const size_t shb = (size_t)13 * C * sizeof(float);
      const bool small = (double)rows * C * (dtype == PFR_BF16 ? 2 : 4) < 1.5e9;   // 32-bit buffer offsets
#define PFR_LNB(TT, K, GG)                                                                                              \
  if (g.cpl == K && g.G == GG) {                                                                                        \
    if (small) hipLaunchKernelGGL((layernorm_bwd3_kernel<TT, K, GG>), dim3(nb), dim3(256), shb, st, (const TT*)dy, (const TT*)x, mean, rstd, gamma, (const TT*)dres, (TT*)dx, part, dxsum_part, rows, C, g.cpr); \
    else hipLaunchKernelGGL((layernorm_bwd2_kernel<TT, K, GG>), dim3(nb), dim3(256), shb, st, (const TT*)dy, (const TT*)x, mean, rstd, gamma, (const TT*)dres, (TT*)dx, part, dxsum_part, rows, C, g.cpr); \
  }
#define PFR_LNB_ALL(TT) PFR_LNB(TT, 1, 16) PFR_LNB(TT, 1, 32) PFR_LNB(TT, 1, 64) PFR_LNB(TT, 2, 64) PFR_LNB(TT, 3, 64) PFR_LNB(TT, 4, 64)
      if (dtype == PFR_BF16) { PFR_LNB_ALL(bf16_t) }
      else { PFR_LNB_ALL(float) }
#undef PFR_LNB_ALL
#undef PFR_LNB
      PFR_CHECK_LAUNCH();
      return PFR_OK;
    }
  }
  PFR_CHECK_ARG(!dxsum_part, "pfr_layernorm_bwd_dxsum: this channel count takes the generic kernel (no dx column sums): see pfr_layernorm_bwd_dxsum_ok");
  int rc = dtype == PFR_BF16 ? ln_bwd_launch<bf16_t>(dy, x, mean, rstd, gamma, dres, dx, part, rows, C, nb, rpb, st)
                             : ln_bwd_launch<float>(dy, x, mean, rstd, gamma, dres, dx, part, rows, C, nb, rpb, st);
  if (rc != PFR_OK) return rc;
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
extern "C" int pfr_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                 const void* dres, void* dx, float* part, int dtype, long rows, int C, hipStream_t st) {
  return pfr_layernorm_bwd_dxsum(dy, x, mean, rstd, gamma, dres, dx, part, nullptr, dtype, rows, C, st);
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
    for (int e = 0; e < KP; ++e) { float cdf, pdf; gelu_cdf_pdf(f[e], cdf, pdf); f[e] *= cdf; }
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
      float cdf, pdf;
      gelu_cdf_pdf(v, cdf, pdf);
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
  // windows per column / row and multiply-shift reciprocals of the divisors of the unit / token decode (a workgroup is one wave and one
  // (window, head): its ~0.5 µs of index arithmetic — five integer divisions by run-time values, ~150 instructions — ran before the
  // first load could be issued)
  int nwh, nww;
  FastDiv dheads, dnww, dnwh, dw;
};
static WinAttn wa_make(int B, int H, int W, int heads, int hd, int w, int shift, float scale) {
  WinAttn a;
  a.B = B; a.H = H; a.W = W; a.heads = heads; a.hd = hd; a.w = w; a.shift = shift; a.scale = scale;
  a.nwh = H / w; a.nww = W / w;
  a.dheads = make_fastdiv((uint32_t)heads); a.dnww = make_fastdiv((uint32_t)a.nww); a.dnwh = make_fastdiv((uint32_t)a.nwh);
  a.dw = make_fastdiv((uint32_t)w);
  return a;
}

// bias(+mask) table of one attention block: tab[variant][i][j], variant = 2*(last window row) + (last window column),
// rows padded to WA_MAXT = 64 with −inf outside the w² x w² block (one tiny launch per block and step instead of integer
// divisions per score element in every workgroup).
// second copy in the MFMA kernels' ACCESS order: a wave reads, for one query half `it`, key tile jt and row group g, the four
// biases of keys 32jt + 8g + 4·half … +3 for query 32it + (lane&31) — in [var][it][jt][g][half][query][4] order those 64 16-byte
// pieces are 1 KB contiguous (one coalesced request stream instead of 32 rows 256 bytes apart: the bias reads were the largest
// consumer of the texture-address unit in both attention kernels)
__device__ __forceinline__ size_t wa_perm_index(int var, int i, int j) {
  const int it = i >> 5, il = i & 31, jt = j >> 5, jl = j & 31;
  return ((((((size_t)var * 2 + it) * 2 + jt) * 4 + (jl >> 3)) * 2 + ((jl >> 2) & 1)) * 32 + il) * 4 + (jl & 3);
}
__global__ void window_bias_table_kernel(const float* __restrict__ pos, float* __restrict__ tab, int w, int shift) {
  const int nt = w * w;
  constexpr int ntp = WA_MAXT;
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
  tab[4 * ntp * ntp + wa_perm_index(var, i, j)] = b;
}

__device__ __forceinline__ size_t wa_token_off(const WinAttn& a, int b, int gy, int gx, int t) {
  const int tr = (int)fdiv((uint32_t)t, a.dw);
  int y = gy * a.w + tr + a.shift, x = gx * a.w + (t - tr * a.w) + a.shift;
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
  const float* btab = tab + (size_t)(((gy == nwh - 1) ? 2 : 0) + ((gx == nww - 1) ? 1 : 0)) * WA_MAXT * WA_MAXT;
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
        s[(i0 + x) * WA_SS + j0 + y] = acc[x][y] * a.scale + btab[(i0 + x) * WA_MAXT + j0 + y];   // −inf outside w² x w²
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
  const float* btab = tab + (size_t)(((gy == nwh - 1) ? 2 : 0) + ((gx == nww - 1) ? 1 : 0)) * WA_MAXT * WA_MAXT;
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
        s[i * WA_SS + j] = acc[x][y] * a.scale + btab[i * WA_MAXT + j];
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

// ================================================================================================
// MFMA window attention (bf16, head_dim 32, w² ≤ 64): ONE WAVE per (image, window, head), no inter-wave traffic.
//
// All products run on v_mfma_f32_32x32x16_bf16 over 64x64 (padded) token tiles.  Scores are produced TRANSPOSED
// (rows j = keys, columns i = queries): the 32x32 accumulator then holds, per lane, one query column and 16 keys in
// registers, so the softmax over keys is an in-lane reduction plus ONE lane^32 exchange, and the probabilities are
// ALREADY in the B-operand layout of the next product (Oᵀ = Vᵀ·Pᵀ) — no shuffles, no LDS round trip.  The register order
// of the accumulator rows is (r&3) + 8(r>>2) + 4(lane>>5); the A operand (Vᵀ, Kᵀ, Qᵀ, dOᵀ) is read from a natural
// [token][d] LDS tile with ds_read_b64_tr_b16 at exactly those rows, so the reduction index is permuted identically on
// both sides.
// [64 tokens][32 d] bf16 tiles: UNPADDED 64-byte rows whose four 16-byte chunks are XOR-swizzled with (row >> 2) & 3 (round 5).  Both read
// patterns are then bank-conflict free: the row fragments (ds_read_b128, 16-lane groups of rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}:
// rows with equal row % 4 share a 16-dword bank window and differ in (row >> 2) & 3, i.e. in the chunk position) and the transposed
// fragments (ds_read_b64_tr_b16, four consecutive rows x 16 dwords per 32-lane group: same swizzle value, windows 0-15 / 16-31 / 32-47 /
// 48-63).  The 80-byte padded rows before were conflict free for the row fragments only: PMC SQ_LDS_BANK_CONFLICT was 36 % (forward) /
// 31 % (backward) of SQ_LDS_IDX_ACTIVE, and the LDS array the busiest unit of both kernels.
#define WA_RS 64
__device__ __forceinline__ int wa_swz(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }

__device__ __forceinline__ bf16x8 wa_rowfrag(const char* tile, int tok_tile, int ks, int lane) {
  // A/B operand with the reduction over d: row (token) = lane&31 (+32·tok_tile), d = 16·ks + 8·(lane>>5) … +7
  const int row = (lane & 31) + 32 * tok_tile;
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(tile + row * WA_RS + wa_swz(row, 2 * ks + (lane >> 5)) * 16));
}
__device__ __forceinline__ bf16x8 wa_trfrag(const char* tile, int tok_tile, int t, int lane) {
  // A operand [M = d][K = token] from the [token][d] tile: lane (d = lane&31, half = lane>>5), reduction slots e = 0..7 ↔
  // token 32·tok_tile + 16·t + 8·(e>>2) + 4·half + (e&3)   (the accumulator-row order, see above)
  const int g = lane >> 4, s4 = lane & 15;
  // 8 bytes at (row, d = 16·(g&1) + 4·(s4&3) …): chunk 2·(g&1) + ((s4&3) >> 1), second half of the chunk for odd s4&3
  const int row = 32 * tok_tile + 16 * t + (g >> 1) * 4 + (s4 >> 2), ch = 2 * (g & 1) + ((s4 & 3) >> 1), off = (s4 & 1) * 8;
  const char* a = tile + row * WA_RS + wa_swz(row, ch) * 16 + off;
  const char* b = tile + (row + 8) * WA_RS + wa_swz(row + 8, ch) * 16 + off;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b));
  u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  u32x4 u = {l2[0], l2[1], h2[0], h2[1]};
  return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ bf16x8 wa_accfrag(const f32x16& v, int t) {   // accumulator rows 8t … 8t+7 → bf16 B operand
  bf16x8 f;
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (bf16_t)v[8 * t + e];
  return f;
}
__device__ __forceinline__ int wa_accrow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Workgroups are dealt round-robin to the 8 XCDs (each with an L2 of its own).  The heads of one window read interleaved 64-byte
// slices of the same [token][3C] rows: with unit = blockIdx every XCD fetched whole 128-byte lines for half their bytes (and wrote
// half lines).  Each XCD therefore takes a CONTIGUOUS range of units, walked in order.
__device__ __forceinline__ int wa_xcd_unit(int blk, int n) {
  const int per = n >> 3, rem = n & 7, x = blk & 7;
  return x * per + min(x, rem) + (blk >> 3);
}
struct WaUnit { int b, gy, gx, h, var, unit; };
__device__ __forceinline__ WaUnit wa_decode(const WinAttn& a, int unit) {
  const int nwh = a.nwh, nww = a.nww;
  WaUnit u;
  u.unit = unit;
  int q = (int)fdiv((uint32_t)unit, a.dheads);
  u.h = unit - q * a.heads; unit = q;
  q = (int)fdiv((uint32_t)unit, a.dnww);
  u.gx = unit - q * nww; unit = q;
  q = (int)fdiv((uint32_t)unit, a.dnwh);
  u.gy = unit - q * nwh;
  u.b = q;
  u.var = ((u.gy == nwh - 1) ? 2 : 0) + ((u.gx == nww - 1) ? 1 : 0);
  return u;
}
// loads one operand's [64 tokens][32 d] slice into the LDS tile (zeros past nt).  Lane = (token row lane>>2 (+16·p), 16-byte chunk
// lane&3): the four lanes of a quad read one contiguous 64-byte head row, i.e. ONE request for the texture-address unit — with the
// (row lane&31, chunk lane>>5) assignment of round 1 every lane was a request of its own and address processing, not bytes, bounded
// the kernel.  toff: the window's token → row-of-the-[tokens][·] tensor table in LDS (−1 = padding).
__device__ __forceinline__ void wa_load_tile(char* tile, const bf16_t* base, const int* toff, size_t rowstride, int lane) {
  const int ch = lane & 3;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = (lane >> 2) + 16 * p;
    // padding rows carry ~(a valid row) in the table: the load is UNCONDITIONAL (a branch around it makes the compiler wait for
    // every load before the next one — 12-16 serialised round trips per wave, which was 40 % of these kernels) and zeroed by a select
    const int tk = toff[row];
    u32x4 v = ld16(base + (size_t)(tk < 0 ? ~tk : tk) * rowstride + ch * 8);
    if (tk < 0) v = u32x4{0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(tile + row * WA_RS + wa_swz(row, ch) * 16) = v;
  }
}

// Output rows through LDS: an accumulator holds 4 consecutive d of one token per lane (8-byte pieces, four store instructions per
// 64-byte head row, 32 separate rows each); staged as [token][d] (stride WA_TS) the wave stores 16 bytes per lane with the four
// lanes of a quad on one row.  `tile`: which 32-token half; rows with a negative table entry (padding) are not stored.
#define WA_TS 72   // LDS row stride in bytes of a [32 tokens][32 x bf16] staging / quarter tile (64 B + 8 B pad)
__device__ __forceinline__ void wa_store_tile(char* stage, const f32x16& acc, float mul, bf16_t* dst, const int* toff, int tile,
                                              size_t rowstride, int lane) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bf16x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(acc[4 * g + e] * mul);
    *reinterpret_cast<bf16x4*>(stage + (lane & 31) * WA_TS + (8 * g + 4 * (lane >> 5)) * 2) = v;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = (lane >> 2) + 16 * p;
    const int tk = toff[32 * tile + row];
    const char* src = stage + row * WA_TS + (lane & 3) * 16;
    const u32x2 lo = *reinterpret_cast<const u32x2*>(src), hi = *reinterpret_cast<const u32x2*>(src + 8);
    if (tk >= 0) st16(dst + (size_t)tk * rowstride + (lane & 3) * 8, u32x4{lo[0], lo[1], hi[0], hi[1]});
  }
  __builtin_amdgcn_wave_barrier();
}

// (round 5: a PERSISTENT form — one wave walking its XCD's units with the next unit's q / k / v rows and bias rows in flight during the
//  current unit's products, 216 instead of 156 VGPRs — was bit-identical and slower: 111 vs 101 us at 128 x 56² x 3 heads cold, 65 vs 60 at
//  28², no difference inside the Swin-T step.  The kernel is bound by instruction issue at 2-3 waves per SIMD, not by the exposed loads.)
__global__ __launch_bounds__(64) void window_attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ tab,
                                                                  bf16_t* __restrict__ out, WinAttn a) {
  __shared__ __attribute__((aligned(16))) char lq[64 * WA_RS], lk[64 * WA_RS], lv[64 * WA_RS], lo[32 * WA_TS];
  __shared__ int toff[64];
  const int lane = threadIdx.x;
  const WaUnit u = wa_decode(a, wa_xcd_unit(blockIdx.x, gridDim.x));
  const int nt = a.w * a.w, C = a.heads * a.hd;
  const float* ptab = tab + (size_t)(4 + u.var) * WA_MAXT * WA_MAXT + (lane >> 5) * 128 + (lane & 31) * 4;   // access-order copy
  toff[lane] = lane < nt ? (int)wa_token_off(a, u.b, u.gy, u.gx, lane) : ~(int)wa_token_off(a, u.b, u.gy, u.gx, 0);   // one wave: LDS operations complete in order
  __builtin_amdgcn_wave_barrier();
  const bf16_t* qb = qkv + u.h * a.hd;
  wa_load_tile(lq, qb, toff, 3 * (size_t)C, lane);
  wa_load_tile(lk, qb + C, toff, 3 * (size_t)C, lane);
  // the bias rows of the first query half are requested with the tiles (those of the second half while the first is computed):
  // behind the barrier they were a second, fully exposed round trip per half (33 of 128 µs at 128 x 56² x 3 heads); they come from
  // the access-order copy of the table (wa_perm_index): 1 KB contiguous per instruction
  f32x4 bbv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bbv[q] = *reinterpret_cast<const f32x4*>(ptab + q * 256);
  wa_load_tile(lv, qb + 2 * C, toff, 3 * (size_t)C, lane);
  __syncthreads();
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {
    // Sᵀ[j][i] = K·Qᵀ for this query half
    f32x16 sacc[2];
    f32x16 oacc;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[jt][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        sacc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa_rowfrag(lk, jt, ks, lane), wa_rowfrag(lq, it, ks, lane), sacc[jt], 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bb = bbv[4 * jt + g];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaf(sacc[jt][4 * g + e], a.scale, bb[e]);
          sacc[jt][4 * g + e] = v;
          mx = fmaxf(mx, v);
        }
      }
    if (it == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) bbv[q] = *reinterpret_cast<const f32x4*>(ptab + (8 + q) * 256);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (mx == -INFINITY) mx = 0.f;   // padded query row: every score is −inf
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __expf(sacc[jt][e] - mx);
        sacc[jt][e] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
    // Oᵀ[d][i] = Σ_j Vᵀ[d][j]·Pᵀ[j][i]   (normalised afterwards: one multiply per output instead of per probability)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa_trfrag(lv, jt, t, lane), wa_accfrag(sacc[jt], t), oacc, 0, 0, 0);
    wa_store_tile(lo, oacc, inv, out + u.h * a.hd, toff, it, (size_t)C, lane);
  }
}

// backward, ONE orientation (rows j = keys, columns i = queries, as in the forward kernel): per query half it produces Pᵀ and dSᵀ
// in accumulator registers → dQᵀ = Kᵀ·dSᵀ directly; for dVᵀ = dOᵀ·P and dKᵀ = Qᵀ·dS the reduction runs over the queries, which sit
// across lanes, so the two 32x32 bf16 quarters go through a 2.3 KB LDS tile each ([query][key], written as 8-byte rows, read back with
// the transposing ds_read_b64_tr_b16 into B-operand layout).  Nothing is recomputed (round 1 rebuilt S, P and dP in the second
// orientation: +16 MFMAs, +64 exponentials and 64 scattered bias loads per window-head).  V is only ever a row-fragment operand and
// is held in registers straight from global memory (no LDS tile); the bins are folded into per-lane float sums after each
// query half: 20 KB of LDS per wave = 8 window-heads per CU (round 1: 25.6 KB, 6).
__device__ __forceinline__ bf16x8 wa_trfrag_q(const char* tile, int t, int lane) {
  // B operand [K = query][N = key] from the quarter tile; reduction slots e ↔ query 16·t + 8·(e>>2) + 4·half + (e&3) as in wa_trfrag
  const int g = lane >> 4, s4 = lane & 15;
  const char* a = tile + (16 * t + (g >> 1) * 4 + (s4 >> 2)) * WA_TS + ((g & 1) * 16 + (s4 & 3) * 4) * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a + 8 * WA_TS));
  u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  u32x4 u = {l2[0], l2[1], h2[0], h2[1]};
  return __builtin_bit_cast(bf16x8, u);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void window_attn_bwd_mfma_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ tab,
                                                                  const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv,
                                                                  float* __restrict__ dpos_part, WinAttn a) {
  __shared__ __attribute__((aligned(16))) char lq[64 * WA_RS], lk[64 * WA_RS], lg[64 * WA_RS];
  __shared__ __attribute__((aligned(16))) char tp[32 * WA_TS];   // P quarter, then (same bytes, LDS operations are in order) the dS quarter
  // Position-table gradient bins are accumulated as 64-bit INTEGERS: on gfx950 a wave64 `ds_add_f32` costs 190 (idle) … 950
  // (loaded CU) cycles, `ds_add_u64` 14 … 46 (tools/probe/lds_atomic_probe.hip) — the float scatter was 70 % of this kernel's
  // LDS-array time.  Block floating point: each query half scales its dS values by the power of two that puts the largest
  // magnitude at 2^30 (values more than 2^-30 below the largest are truncated — finer than fp32 accumulation), so the
  // sums are order-independent AND exactly homogeneous (doubling dO doubles every bin bit for bit, tests/test_fullsize_gpu.py).
  __shared__ unsigned long long dtab[256];   // 20 KB of LDS per wave in total: 8 window-heads per CU
  __shared__ __attribute__((aligned(16))) int cj[64];
  __shared__ int toff[64];
  const int lane = threadIdx.x;
  const WaUnit u = wa_decode(a, wa_xcd_unit(blockIdx.x, gridDim.x));
  const int nt = a.w * a.w, C = a.heads * a.hd, w = a.w;
  const int ntab = (2 * w - 1) * (2 * w - 1);
  const float* ptab = tab + (size_t)(4 + u.var) * WA_MAXT * WA_MAXT + (lane >> 5) * 128 + (lane & 31) * 4;   // access-order copy
  for (int e = lane; e < 256; e += 64) dtab[e] = 0ull;
  float dacc[4] = {0.f, 0.f, 0.f, 0.f};   // this lane's bins (e = lane + 64·k) over both query halves
  cj[lane] = lane < nt ? (lane / w) * (2 * w - 1) + lane % w : -(1 << 20);
  toff[lane] = lane < nt ? (int)wa_token_off(a, u.b, u.gy, u.gx, lane) : ~(int)wa_token_off(a, u.b, u.gy, u.gx, 0);   // one wave: LDS operations complete in order
  __builtin_amdgcn_wave_barrier();
  size_t tokoff[2];
  bool ok[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int tk = toff[(lane & 31) + 32 * tt];
    ok[tt] = tk >= 0;
    tokoff[tt] = (size_t)(ok[tt] ? tk : ~tk);       // padding rows point at a valid row (loads are unconditional, stores are not)
  }
  const bf16_t* qb = qkv + u.h * a.hd;
  // V row fragments (key j = lane&31 + 32·jt, d = 16·ks + 8·(lane>>5) …): the same 16-byte chunks wa_load_tile would stage
  bf16x8 vf[2][2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 v = ld16(qb + 2 * C + tokoff[jt] * (3 * (size_t)C) + 16 * ks + 8 * (lane >> 5));
      if (!ok[jt]) v = u32x4{0u, 0u, 0u, 0u};
      vf[jt][ks] = __builtin_bit_cast(bf16x8, v);
    }
  wa_load_tile(lq, qb, toff, 3 * (size_t)C, lane);
  wa_load_tile(lk, qb + C, toff, 3 * (size_t)C, lane);
  wa_load_tile(lg, dout + u.h * a.hd, toff, (size_t)C, lane);
  __syncthreads();

  f32x16 dv[2], dk[2];   // dVᵀ / dKᵀ [d][key], accumulated over both query halves
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dv[jt][e] = 0.f; dk[jt][e] = 0.f; }
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {
    const int i = (lane & 31) + 32 * it;
    f32x16 sa[2], dp[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { sa[jt][e] = 0.f; dp[jt][e] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 qf = wa_rowfrag(lq, it, ks, lane), gf = wa_rowfrag(lg, it, ks, lane);
        sa[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa_rowfrag(lk, jt, ks, lane), qf, sa[jt], 0, 0, 0);   // Sᵀ = K·Qᵀ
        dp[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[jt][ks], gf, dp[jt], 0, 0, 0);                      // dPᵀ = V·dOᵀ
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(ptab + (8 * it + 4 * jt + g) * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaf(sa[jt][4 * g + e], a.scale, bb[e]);
          sa[jt][4 * g + e] = v;
          mx = fmaxf(mx, v);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (mx == -INFINITY) mx = 0.f;
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __expf(sa[jt][e] - mx);
        sa[jt][e] = pv;
        sum += pv;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    float dl = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        sa[jt][e] *= inv;
        dl = fmaf(sa[jt][e], dp[jt][e], dl);
      }
    dl += __shfl_xor(dl, 32, 64);
    // dSᵀ = Pᵀ∘(dPᵀ − δ_i) (into dp; sa keeps Pᵀ); position-table gradient; dQᵀ = Kᵀ·dSᵀ
    // table index of (i, j) = base(i) + cj[j]; padded rows / columns give a negative index and are skipped
    const int yi = i / w, xi = i - yi * w;
    const int base_i = i < nt ? (w - 1 - yi) * (2 * w - 1) + (w - 1 - xi) : -(1 << 20);
    float amax = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float dsv = sa[jt][e] * (dp[jt][e] - dl);
        dp[jt][e] = dsv;
        amax = fmaxf(amax, fabsf(dsv));
      }
#pragma unroll
    for (int o = 32; o; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    // scale = 2^(30 - exponent(amax)) (1 when everything is zero / not finite); its reciprocal is exact as well.  |x·scale| < 2^31:
    // ONE float→int conversion (toward zero) per value, sign-extended to the 64-bit bin
    const int ex = (int)((__float_as_uint(amax) >> 23) & 255u);
    const int kexp = (ex == 0 || ex == 255) ? 0 : min(30 - (ex - 127), 120);
    const float sc = __uint_as_float((uint32_t)(kexp + 127) << 23);
    const float inv_sc = __uint_as_float((uint32_t)(127 - kexp) << 23);
    unsigned long long* dt = dtab;
    // padded rows / columns have dS == 0 exactly: they add 0 to a spare bin behind the table instead of branching around the atomic
    const int dump = ntab + (lane & 15);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int4 c4 = *reinterpret_cast<const int4*>(cj + 32 * jt + 8 * g + 4 * (lane >> 5));
        const int cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = base_i + cc[e];
          const int xi32 = (int)(dp[jt][4 * g + e] * sc);                    // exact product (power of two), truncated toward zero
          atomicAdd(&dt[min((unsigned)idx, (unsigned)dump)], (unsigned long long)(long long)xi32);   // negative index → huge unsigned → spare bin
        }
      }
    // fold this half's bins into the lane's float sums and clear them for the next half (its scale differs)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dacc[q] = fmaf((float)(long long)dtab[lane + 64 * q], inv_sc, dacc[q]);
      dtab[lane + 64 * q] = 0ull;
    }
    f32x16 dq;
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[e] = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa_trfrag(lk, jt, t, lane), wa_accfrag(dp[jt], t), dq, 0, 0, 0);
    wa_store_tile(tp, dq, a.scale, dqkv + u.h * a.hd, toff, it, 3 * (size_t)C, lane);
    // dVᵀ[d][j] += Σ_i dOᵀ[d][i]·P[i][j],  dKᵀ[d][j] += Σ_i Qᵀ[d][i]·dS[i][j]   (i over this query half), one key quarter at a time
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 pv;
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] = (bf16_t)sa[jt][4 * g + e];
        *reinterpret_cast<bf16x4*>(tp + (lane & 31) * WA_TS + (8 * g + 4 * (lane >> 5)) * 2) = pv;     // [query lane&31][key 8g + 4·half …+3]
      }
      bf16x8 pb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) pb[t] = wa_trfrag_q(tp, t, lane);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 sv;
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[e] = (bf16_t)dp[jt][4 * g + e];
        *reinterpret_cast<bf16x4*>(tp + (lane & 31) * WA_TS + (8 * g + 4 * (lane >> 5)) * 2) = sv;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        dv[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa_trfrag(lg, it, t, lane), pb[t], dv[jt], 0, 0, 0);
        dk[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa_trfrag(lq, it, t, lane), wa_trfrag_q(tp, t, lane), dk[jt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    wa_store_tile(tp, dk[jt], a.scale, dqkv + C + u.h * a.hd, toff, jt, 3 * (size_t)C, lane);
    wa_store_tile(tp, dv[jt], 1.f, dqkv + 2 * C + u.h * a.hd, toff, jt, 3 * (size_t)C, lane);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (lane + 64 * q < ntab) dpos_part[(size_t)u.unit * ntab + lane + 64 * q] = dacc[q];
}

static bool wa_use_mfma(int dtype, int hd, int w) {
  const bool off = pfr_knob(KNOB_ATTN_MFMA) == 0;
  return !off && dtype == PFR_BF16 && hd == 32 && w * w <= 64;
}

static int wa_check(int B, int H, int W, int heads, int hd, int w, int shift) {
  PFR_CHECK_ARG(w * w <= WA_MAXT && hd <= WA_MAXD && (2 * w - 1) * (2 * w - 1) < 256, "window attention: window %d / head_dim %d too large", w, hd);
  PFR_CHECK_ARG(H % w == 0 && W % w == 0 && shift >= 0 && shift < w && B > 0 && heads > 0, "window attention: bad geometry");
  return PFR_OK;
}

extern "C" long pfr_window_bias_table_floats(int window) {
  (void)window;
  return 8L * WA_MAXT * WA_MAXT;   // natural [4][64][64] + the MFMA kernels' access-order copy
}
// the tables of n attention blocks in ONE launch (12 launches of ~6 µs per Swin-T step otherwise); descs: DEVICE array of BiasTabDesc
struct BiasTabDesc {
  const float* pos;
  float* tab;
  int w, shift;
};
__global__ void window_bias_table_batch_kernel(const BiasTabDesc* __restrict__ descs) {
  const BiasTabDesc d = descs[blockIdx.y];
  const int w = d.w, shift = d.shift, nt = w * w;
  constexpr int ntp = WA_MAXT;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 4 * ntp * ntp) return;
  const int var = e / (ntp * ntp), i = (e / ntp) % ntp, j = e % ntp;
  float b = -INFINITY;
  if (i < nt && j < nt) {
    const int yi = i / w, xi = i % w, yj = j / w, xj = j % w;
    b = d.pos[(yj - yi + w - 1) * (2 * w - 1) + (xj - xi + w - 1)];
    if (shift) {
      if ((var & 2) && ((yi >= w - shift) != (yj >= w - shift))) b = -INFINITY;
      if ((var & 1) && ((xi >= w - shift) != (xj >= w - shift))) b = -INFINITY;
    }
  }
  d.tab[e] = b;
  d.tab[4 * ntp * ntp + wa_perm_index(var, i, j)] = b;
}
extern "C" int pfr_window_bias_table_batch(const void* descs, int n, hipStream_t st) {
  PFR_CHECK_ARG(descs && n > 0 && n <= 65535, "pfr_window_bias_table_batch: bad args");
  const long m = 4L * WA_MAXT * WA_MAXT;
  hipLaunchKernelGGL(window_bias_table_batch_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)n), dim3(256), 0, st, (const BiasTabDesc*)descs);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// tab: fp32 [4][64][64] (+ the permuted copy), recomputed whenever pos changes (once per block and step)
extern "C" int pfr_window_bias_table(const float* pos, float* tab, int window, int shift, hipStream_t st) {
  PFR_CHECK_ARG(pos && tab && window * window <= WA_MAXT, "pfr_window_bias_table: bad args");
  const long n = 4L * WA_MAXT * WA_MAXT;
  hipLaunchKernelGGL(window_bias_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pos, tab, window, shift);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

extern "C" int pfr_window_attn_fwd(const void* qkv, const float* pos, void* out, int dtype, int B, int H, int W, int heads,
                                   int head_dim, int window, int shift, float scale, hipStream_t st) {
  PFR_CHECK_ARG(qkv && pos && out, "pfr_window_attn_fwd: null pointer");
  PFR_CHECK_ARG(head_dim % (dtype == PFR_BF16 ? 8 : 4) == 0, "pfr_window_attn_fwd: head_dim must be a multiple of the 16-byte chunk");
  if (int rc = wa_check(B, H, W, heads, head_dim, window, shift)) return rc;
  const WinAttn a = wa_make(B, H, W, heads, head_dim, window, shift, scale);
  const dim3 grid((unsigned)(B * (H / window) * (W / window) * heads));
  if (wa_use_mfma(dtype, head_dim, window)) hipLaunchKernelGGL(window_attn_fwd_mfma_kernel, grid, dim3(64), 0, st, (const bf16_t*)qkv, pos, (bf16_t*)out, a);
  else if (dtype == PFR_BF16) hipLaunchKernelGGL(window_attn_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)qkv, pos, (bf16_t*)out, a);
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
  const WinAttn a = wa_make(B, H, W, heads, head_dim, window, shift, scale);
  const dim3 grid((unsigned)(B * (H / window) * (W / window) * heads));
  if (wa_use_mfma(dtype, head_dim, window)) hipLaunchKernelGGL(window_attn_bwd_mfma_kernel, grid, dim3(64), 0, st, (const bf16_t*)qkv, pos, (const bf16_t*)dout, (bf16_t*)dqkv, dpos_part, a);
  else if (dtype == PFR_BF16) hipLaunchKernelGGL(window_attn_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)qkv, pos, (const bf16_t*)dout, (bf16_t*)dqkv, dpos_part, a);
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
