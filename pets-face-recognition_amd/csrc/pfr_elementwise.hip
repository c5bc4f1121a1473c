// pfr_elementwise.hip — the HBM-bound kernels of the ResNet feature extractor: layout conversion, train/eval
// BatchNorm (statistics finalise, apply+ReLU+residual, backward reduce / apply), max/avg pooling, optimiser steps.
//
// Reference call sites replaced (all via torchvision resnet50, configs/dog_fe/fe_dogs_config.py:102-103):
// nn.BatchNorm2d (eps 1e-5, momentum 0.1, unbiased running var), nn.ReLU, residual add, nn.MaxPool2d(3,2,1),
// nn.AdaptiveAvgPool2d(1); torch.optim.SGD(momentum 0.9, per-group lr / weight decay, fe_dogs_config.py:123-133)
// and torch.optim.AdamW (configs/dog_fe/body_dog_fe.py:121-131).
//
// Everything is NHWC with 16-byte channel chunks per lane (8 bf16 / 4 f32): a wave instruction moves 1 KiB of
// contiguous HBM.  Per-channel reductions keep the channel chunk in the lane and reduce over rows: per-thread
// partials → LDS across the row-lanes of a block → one deterministic partial row per block (no atomics).
#include "pfr_common.h"
#include <hip/hip_ext.h>

// ------------------------------------------------------------------------------------------------
// layout / dtype conversion
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int HW, int Cp) {
  // 16-byte-store path: one thread per (n, pixel, chunk of KP output channels); its KP plane loads are unconditional (clamped
  // channel, zeroed afterwards) so they are all in flight at once.  Reads are coalesced per channel plane.
  constexpr int KP = DT<T>::KPACK;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (Cp % KP == 0) {
    const int cpr = Cp / KP;
    if (i >= (size_t)N * HW * cpr) return;
    const int chunk = (int)(i % cpr);
    const size_t np = i / cpr, n = np / HW, pix = np % HW;
    const int c0 = chunk * KP;
    float v[KP];
    if (C == 3) {   // RGB frames: exactly three plane loads
      const float* src = x + n * 3 * HW + pix;
      const float r = src[0], g = src[HW], b = src[2 * (size_t)HW];
#pragma unroll
      for (int e = 0; e < KP; ++e) v[e] = 0.f;
      v[0] = r; v[1] = g; v[2] = b;
      st16(y + np * Cp + c0, Chunk<T>::pack(v));
      return;
    }
#pragma unroll
    for (int e = 0; e < KP; ++e) v[e] = x[(n * C + min(c0 + e, C - 1)) * HW + pix];
#pragma unroll
    for (int e = 0; e < KP; ++e)
      if (c0 + e >= C) v[e] = 0.f;
    st16(y + np * Cp + c0, Chunk<T>::pack(v));
    return;
  }
  if (i >= (size_t)N * HW) return;
  const size_t n = i / HW, pix = i % HW;
  T* dst = y + i * Cp;
  for (int c = 0; c < Cp; ++c) {
    float v = c < C ? x[(n * C + c) * HW + pix] : 0.f;
    dst[c] = from_f32<T>(v);
  }
}

extern "C" int pfr_nchw_to_nhwc(const float* x, void* y, int dtype, int N, int C, int H, int W, int Cp, hipStream_t st) {
  PFR_CHECK_ARG(x && y && Cp >= C && C > 0, "pfr_nchw_to_nhwc: bad args");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  const size_t n = (size_t)N * H * W * (Cp % kp == 0 ? Cp / kp : 1);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (dtype == PFR_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, x, (bf16_t*)y, N, C, H * W, Cp);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(blocks), dim3(256), 0, st, x, (float*)y, N, C, H * W, Cp);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

template <typename TI, typename TOo>
__global__ void cast_kernel(const TI* __restrict__ x, TOo* __restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t done = 0;
  if constexpr (sizeof(TI) == 4 && sizeof(TOo) == 2) {   // fp32 master weights -> bf16 shadow: 8 values per thread, 2 x 16-byte loads, one 16-byte store
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
      const size_t n8 = n / 8;
      for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
        float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        st16(y + 8 * i, Chunk<TOo>::pack(f));
      }
      done = 8 * n8;
    }
  }
  for (size_t i = done + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = from_f32<TOo>(to_f32(x[i]));
}

extern "C" int pfr_cast(const void* x, int src_dtype, void* y, int dst_dtype, size_t n, hipStream_t st) {
  PFR_CHECK_ARG(x && y, "pfr_cast: null pointer");
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) return PFR_OK;
  if (src_dtype == PFR_F32 && dst_dtype == PFR_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, st, (const float*)x, (bf16_t*)y, n);
  else if (src_dtype == PFR_BF16 && dst_dtype == PFR_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (float*)y, n);
  else if (src_dtype == PFR_F32 && dst_dtype == PFR_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y, n);
  else {
    pfr_set_error("pfr_cast: unsupported dtype pair %d -> %d", src_dtype, dst_dtype);
    return PFR_ERR_UNSUPPORTED;
  }
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// weights [O][R][S][I] -> data-gradient weights [I][R][S][O] with both taps flipped
template <typename T>
__global__ void weight_dgrad_kernel(const T* __restrict__ w, T* __restrict__ wt, int O, int R, int S, int I) {
  const size_t n = (size_t)O * R * S * I;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    // i indexes the OUTPUT [ci][r][s][o] so stores are coalesced
    size_t rest = i;
    const int o = rest % O; rest /= O;
    const int s = rest % S; rest /= S;
    const int r = rest % R; rest /= R;
    const int ci = (int)rest;
    wt[i] = w[(((size_t)o * R + (R - 1 - r)) * S + (S - 1 - s)) * I + ci];
  }
}

extern "C" int pfr_weight_dgrad_layout(const void* w, void* wt, int dtype, int O, int R, int S, int I, hipStream_t st) {
  PFR_CHECK_ARG(w && wt, "pfr_weight_dgrad_layout: null pointer");
  const size_t n = (size_t)O * R * S * I;
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (dtype == PFR_BF16)
    hipLaunchKernelGGL(weight_dgrad_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)w, (bf16_t*)wt, O, R, S, I);
  else
    hipLaunchKernelGGL(weight_dgrad_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)w, (float*)wt, O, R, S, I);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// every conv of a network in ONE launch: blockIdx.y = layer (descriptor table in HBM), grid-stride over the layer's elements
struct WtDesc {
  const void* w;
  void* wt;
  int O, R, S, I;
};
// one tap of one layer is an O x I matrix transpose (input row stride RS*I, output row stride RS*O): 64x64 tiles through LDS so
// that both the loads (along ci) and the stores (along o) are contiguous
template <typename T>
__global__ __launch_bounds__(256) void weight_dgrad_batch_kernel(const WtDesc* __restrict__ descs) {
  const WtDesc d = descs[blockIdx.y];
  const T* __restrict__ w = reinterpret_cast<const T*>(d.w);
  T* __restrict__ wt = reinterpret_cast<T*>(d.wt);
  const uint32_t O = d.O, I = d.I, RS = d.R * d.S;
  const uint32_t to = (O + 63) / 64, ti = (I + 63) / 64, ntiles = RS * to * ti;
  __shared__ T tile[64][64 + 4 / sizeof(T)];
  const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint32_t tap = t % RS, rest = t / RS;
    const uint32_t o0 = (rest % to) * 64, c0 = (rest / to) * 64;
    __syncthreads();
    if (c0 + tx < I)
#pragma unroll
      for (uint32_t k = 0; k < 16; ++k)
        if (const uint32_t r = ty + 4 * k; o0 + r < O) tile[r][tx] = w[((size_t)(o0 + r) * RS + tap) * I + c0 + tx];
    __syncthreads();
    if (o0 + tx < O)
#pragma unroll
      for (uint32_t k = 0; k < 16; ++k)
        if (const uint32_t r = ty + 4 * k; c0 + r < I) wt[((size_t)(c0 + r) * RS + (RS - 1 - tap)) * O + o0 + tx] = tile[tx][r];    // both taps flipped = the tap index reversed
  }
}

extern "C" int pfr_weight_dgrad_layout_batch(const void* descs, int n_layers, int dtype, hipStream_t st) {
  PFR_CHECK_ARG(descs && n_layers > 0 && n_layers <= 65535, "pfr_weight_dgrad_layout_batch: bad args");
  const dim3 grid(128, n_layers);
  if (dtype == PFR_BF16)
    hipLaunchKernelGGL(weight_dgrad_batch_kernel<bf16_t>, grid, dim3(256), 0, st, (const WtDesc*)descs);
  else
    hipLaunchKernelGGL(weight_dgrad_batch_kernel<float>, grid, dim3(256), 0, st, (const WtDesc*)descs);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ------------------------------------------------------------------------------------------------
// column-reduction scaffold: thread = (chunk column, row lane)
struct ColGeom {
  int cw;      // chunk columns handled per block (power of two <= 256)
  int rl;      // row lanes per block = 256 / cw
  int cpr;     // chunks per row (C / KP)
  int gy;      // blocks along columns
  int gx;      // blocks along rows
};
static ColGeom col_geom(int C, int kp, size_t rows, size_t target = 512) {
  ColGeom g;

  g.cpr = C / kp;
  int cw = 1;
  while (cw < g.cpr && cw < 256) cw <<= 1;
  g.cw = cw;
  g.rl = 256 / cw;
  g.gy = (g.cpr + cw - 1) / cw;
  size_t want = target / g.gy;
  size_t maxb = (rows + g.rl * 4 - 1) / (g.rl * 4);  // at least 4 rows per lane
  if (want > maxb) want = maxb;
  if (want < 1) want = 1;
  g.gx = (int)want;
  return g;
}
extern "C" int pfr_colreduce_blocks(int C, int dtype, long rows) {
  return col_geom(C, dtype == PFR_BF16 ? 8 : 4, (size_t)rows).gx;
}

template <int NQ, int KP>
__device__ __forceinline__ void col_block_reduce(float (&v)[NQ][KP], float* lds, int cw, int rl, int col, int rlane,
                                                 int cglob, int cpr, float* out_row, int C) {
  // lds: [rl][cw][NQ*KP]
  float* mine = lds + ((size_t)rlane * cw + col) * (NQ * KP);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int e = 0; e < KP; ++e) mine[q * KP + e] = v[q][e];
  __syncthreads();
  if (rlane == 0 && cglob < cpr) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < KP; ++e) {
        float a = 0.f;
        for (int r = 0; r < rl; ++r) a += lds[((size_t)r * cw + col) * (NQ * KP) + q * KP + e];
        out_row[(size_t)q * C + cglob * KP + e] = a;
      }
  }
}

// ---- standalone batch statistics of an NHWC tensor → partials [gx][2][C] = (mean_b, M2_b) of block b's row range
//      [b*rpb, (b+1)*rpb); sums are taken around the block's first row (shift) to avoid E[x²]−E[x]² cancellation.
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, float* __restrict__ part, size_t rows,
                                                       int C, int cw, int rl, int cpr, size_t rpb) {
  constexpr int KP = DT<T>::KPACK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int col = threadIdx.x % cw, rlane = threadIdx.x / cw;
  const int cglob = blockIdx.y * cw + col;
  const size_t rbeg = (size_t)blockIdx.x * rpb;
  const size_t rend = rbeg + rpb < rows ? rbeg + rpb : rows;
  float v[2][KP], k[KP];
#pragma unroll
  for (int e = 0; e < KP; ++e) { v[0][e] = 0.f; v[1][e] = 0.f; k[e] = 0.f; }
  if (cglob < cpr && rbeg < rows) {
    Chunk<T>::unpack(ld16(x + rbeg * C + cglob * KP), k);
    for (size_t r = rbeg + rlane; r < rend; r += rl) {
      float f[KP];
      Chunk<T>::unpack(ld16(x + r * C + cglob * KP), f);
#pragma unroll
      for (int e = 0; e < KP; ++e) { const float d = f[e] - k[e]; v[0][e] += d; v[1][e] = fmaf(d, d, v[1][e]); }
    }
  }
  // reduce over row lanes, then convert to (mean, M2)
  float* mine = lds + ((size_t)rlane * cw + col) * (2 * KP);
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < KP; ++e) mine[q * KP + e] = v[q][e];
  __syncthreads();
  if (rlane == 0 && cglob < cpr) {
    const float nt = (float)(rend > rbeg ? rend - rbeg : 1);
    float* out_row = part + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < KP; ++e) {
      float a = 0.f, b = 0.f;
      for (int r = 0; r < rl; ++r) {
        a += lds[((size_t)r * cw + col) * (2 * KP) + e];
        b += lds[((size_t)r * cw + col) * (2 * KP) + KP + e];
      }
      out_row[cglob * KP + e] = k[e] + a / nt;
      out_row[(size_t)C + cglob * KP + e] = b - a * a / nt;
    }
  }
}

// rows handled by one partial of pfr_bn_stats
extern "C" long pfr_bn_stats_rows_per_part(int C, int dtype, long rows) {
  ColGeom g = col_geom(C, dtype == PFR_BF16 ? 8 : 4, (size_t)rows);
  return (long)(((size_t)rows + g.gx - 1) / g.gx);
}

extern "C" int pfr_bn_stats(const void* x, int dtype, long rows, int C, float* part, hipStream_t st) {
  PFR_CHECK_ARG(x && part, "pfr_bn_stats: null pointer");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  PFR_CHECK_ARG(C % kp == 0, "pfr_bn_stats: C %% %d != 0", kp);
  ColGeom g = col_geom(C, kp, (size_t)rows);
  const size_t rpb = ((size_t)rows + g.gx - 1) / g.gx;
  const size_t sh = (size_t)256 * 2 * kp * sizeof(float);
  if (dtype == PFR_BF16)
    hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, dim3(g.gx, g.gy), dim3(256), sh, st, (const bf16_t*)x, part, (size_t)rows, C, g.cw, g.rl, g.cpr, rpb);
  else
    hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(g.gx, g.gy), dim3(256), sh, st, (const float*)x, part, (size_t)rows, C, g.cw, g.rl, g.cpr, rpb);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ---- finalise: partials [nparts][2][C] of (mean_t, M2_t), part t covering rows [t*rpp, min(count,(t+1)*rpp)) →
//      mean, invstd, scale = γ·invstd, shift = β − mean·scale, running stats.
// Two-pass Chan merge about the group / grand mean.  Level 1 (only when there are many parts) merges groups of parts
// in parallel (grid = C/16 x groups) into a small [groups][2][C] (+ counts) buffer, level 2 finishes.
__device__ __forceinline__ float part_rows(long total, long rpp, int i) {
  const long left = total - (long)i * rpp;
  return (float)(left < rpp ? left : rpp);
}

// merges parts [pbeg, pend) for 16 channels; returns (via lane pl==0) n, mean, M2.  counts: optional per-part row counts.
// These launches sit on the dependency chain of every BatchNorm, so their latency is what matters: up to 256 parts (always, as
// pfr_bn_finalize groups larger sets first) every thread issues ALL its loads up front from clamped addresses and keeps the values in
// registers for the second (M2) pass — one memory round trip instead of one per part and pass.  Summation order is unchanged.
#define PFR_MERGE_K 16
__device__ __forceinline__ void merge_parts(const float* __restrict__ part, const float* __restrict__ counts, int pbeg,
                                            int pend, long rpp, long total, int C, int c, int cl, int pl,
                                            float (*l1)[17], float (*l2)[17], float& n_out, float& mean_out, float& m2_out) {
  const bool fast = pend - pbeg <= 16 * PFR_MERGE_K;
  float mu[PFR_MERGE_K], q2[PFR_MERGE_K], nk[PFR_MERGE_K];
  float a = 0.f, n = 0.f;
  if (fast) {
    const int cc = c < C ? c : C - 1;
    const float* cptr = counts ? counts : part;   // (any readable floats when there are no counts: multiplied by 0)
    const float csel = counts ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < PFR_MERGE_K; ++k) {
      const int i = pbeg + pl + 16 * k;
      const int ic = i < pend ? i : pbeg;
      mu[k] = part[((size_t)ic * 2 + 0) * C + cc];
      q2[k] = part[((size_t)ic * 2 + 1) * C + cc];
      // row count of the part: from `counts` or computed.  Blended arithmetically (csel is 0 or 1) so that the load is always used:
      // a select lets the compiler move the load back into a branch, and a load in a branch is followed by a full vmcnt(0) wait
      nk[k] = fmaf(cptr[ic], csel, (1.f - csel) * part_rows(total, rpp, ic));
    }
    __builtin_amdgcn_sched_barrier(0);   // (the compiler otherwise sinks each load to its first use: one round trip per part again)
#pragma unroll
    for (int k = 0; k < PFR_MERGE_K; ++k)
      if (!(c < C && pbeg + pl + 16 * k < pend)) { nk[k] = 0.f; q2[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < PFR_MERGE_K; ++k)
      if (pbeg + pl + 16 * k < pend) {     // (same operations, in the same order, as the loop below)
        n += nk[k];
        a = fmaf(nk[k], mu[k], a);
      }
  } else if (c < C) {
    for (int i = pbeg + pl; i < pend; i += 16) {
      const float nt = counts ? counts[i] : part_rows(total, rpp, i);
      n += nt;
      a = fmaf(nt, part[((size_t)i * 2 + 0) * C + c], a);
    }
  }
  l1[pl][cl] = a;
  l2[pl][cl] = n;
  __syncthreads();
  float sa = 0.f, sn = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { sa += l1[i][cl]; sn += l2[i][cl]; }
  const float mean = sn > 0.f ? sa / sn : 0.f;
  __syncthreads();
  float m2 = 0.f;
  if (fast) {
#pragma unroll
    for (int k = 0; k < PFR_MERGE_K; ++k)
      if (c < C && pbeg + pl + 16 * k < pend) {
        const float d = mu[k] - mean;
        m2 += q2[k] + nk[k] * d * d;
      }
  } else if (c < C) {
    for (int i = pbeg + pl; i < pend; i += 16) {
      const float nt = counts ? counts[i] : part_rows(total, rpp, i);
      const float d = part[((size_t)i * 2 + 0) * C + c] - mean;
      m2 += part[((size_t)i * 2 + 1) * C + c] + nt * d * d;
    }
  }
  l1[pl][cl] = m2;
  __syncthreads();
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s2 += l1[i][cl];
  __syncthreads();
  n_out = sn;
  mean_out = mean;
  m2_out = s2;
}

__global__ __launch_bounds__(256) void bn_merge_groups_kernel(const float* __restrict__ part, int nparts, long rpp, long total,
                                                              int C, int per_group, float* __restrict__ gpart,
                                                              float* __restrict__ gcount) {
  __shared__ float l1[16][17], l2[16][17];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int g = blockIdx.y;
  const int pbeg = g * per_group, pend = min(nparts, pbeg + per_group);
  float n, mean, m2;
  merge_parts(part, nullptr, pbeg, pend, rpp, total, C, c, cl, pl, l1, l2, n, mean, m2);
  if (pl == 0 && c < C) {
    gpart[((size_t)g * 2 + 0) * C + c] = mean;
    gpart[((size_t)g * 2 + 1) * C + c] = m2;
    if (c == 0) gcount[g] = n;
  }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ counts,
                                                          int nparts, long rpp, int C, float count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ mean_out,
                                                          float* __restrict__ invstd_out, float* __restrict__ scale,
                                                          float* __restrict__ shift) {
  __shared__ float l1[16][17], l2[16][17];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float n, mean, m2;
  // the per-channel operands of the tail are requested before the merge (their latency overlaps the merge's own loads)
  const int cc = c < C ? c : C - 1;
  const float g = gamma ? gamma[cc] : 1.f, bb = beta ? beta[cc] : 0.f;
  const float rm = running_mean ? running_mean[cc] : 0.f, rv = running_mean ? running_var[cc] : 0.f;
  merge_parts(part, counts, 0, nparts, rpp, (long)count, C, c, cl, pl, l1, l2, n, mean, m2);
  if (pl == 0 && c < C) {
    float var = fmaxf(m2 / count, 0.f);  // biased batch variance
    const float invstd = rsqrtf(var + eps);
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    scale[c] = g * invstd;
    shift[c] = bb - mean * g * invstd;
    if (running_mean) {
      const float unb = count > 1.f ? var * count / (count - 1.f) : var;
      running_mean[c] = (1.f - momentum) * rm + momentum * mean;
      running_var[c] = (1.f - momentum) * rv + momentum * unb;
    }
  }
}

#define PFR_BN_GROUPS 64
// scratch floats pfr_bn_finalize needs when nparts is large (0 → none needed)
extern "C" long pfr_bn_finalize_ws_floats(int nparts, int C) {
  return nparts > 4 * PFR_BN_GROUPS ? (long)PFR_BN_GROUPS * (2 * C + 1) : 0;
}

extern "C" int pfr_bn_finalize(const float* part, int nparts, long rows_per_part, int C, float count, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                               float* mean, float* invstd, float* scale, float* shift, float* workspace, hipStream_t st) {
  PFR_CHECK_ARG(part && mean && invstd && scale && shift, "pfr_bn_finalize: null pointer");
  PFR_CHECK_ARG(rows_per_part > 0 && (long)nparts * rows_per_part >= (long)count, "pfr_bn_finalize: parts do not cover count");
  const float* counts = nullptr;
  if (workspace && pfr_bn_finalize_ws_floats(nparts, C) > 0) {
    const int per_group = (nparts + PFR_BN_GROUPS - 1) / PFR_BN_GROUPS;
    const int groups = (nparts + per_group - 1) / per_group;
    float* gpart = workspace;
    float* gcount = workspace + (size_t)PFR_BN_GROUPS * 2 * C;
    hipLaunchKernelGGL(bn_merge_groups_kernel, dim3((C + 15) / 16, groups), dim3(256), 0, st, part, nparts, rows_per_part,
                       (long)count, C, per_group, gpart, gcount);
    PFR_CHECK_LAUNCH();
    part = gpart;
    counts = gcount;
    nparts = groups;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, part, counts, nparts, rows_per_part, C, count,
                     gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// eval-mode BN folded into the producing convolution (inference embedder): w'[co][k] = w[co][k]·γ/√(rv+eps),
// b'[co] = β − rm·γ/√(rv+eps).  ONE launch for all convolutions of a network: blockIdx.y = layer, descriptors in HBM.
struct FoldDesc {
  const void* src;     // weights [Cout][K]: fp32 master (src_f32 = 1) or compute dtype
  const float* gamma;  // may be null (→ 1)
  const float* beta;   // may be null (→ 0)
  const float* rm;
  const float* rv;
  void* wout;          // [Cout][K] compute dtype
  float* bout;         // [Cout]
  long cout, k;
  float eps;
  int src_f32;
};
template <typename T>
__global__ __launch_bounds__(256) void fold_bn_kernel(const FoldDesc* __restrict__ descs) {
  const FoldDesc d = descs[blockIdx.y];
  const long n = d.cout * d.k;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long co = i / d.k;
    const float sc = (d.gamma ? d.gamma[co] : 1.f) * rsqrtf(d.rv[co] + d.eps);
    const float w = d.src_f32 ? reinterpret_cast<const float*>(d.src)[i] : to_f32(reinterpret_cast<const T*>(d.src)[i]);
    reinterpret_cast<T*>(d.wout)[i] = from_f32<T>(w * sc);
    if (i - co * d.k == 0) d.bout[co] = (d.beta ? d.beta[co] : 0.f) - d.rm[co] * sc;
  }
}
extern "C" int pfr_fold_bn(const void* descs, int ndesc, int dtype, hipStream_t st) {
  PFR_CHECK_ARG(descs && ndesc > 0, "pfr_fold_bn: bad args");
  const dim3 grid(64, (unsigned)ndesc);
  if (dtype == PFR_BF16) hipLaunchKernelGGL(fold_bn_kernel<bf16_t>, grid, dim3(256), 0, st, (const FoldDesc*)descs);
  else hipLaunchKernelGGL(fold_bn_kernel<float>, grid, dim3(256), 0, st, (const FoldDesc*)descs);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ---- fold only when something changed.  The fold (and the master -> shadow cast) of an inference forward costs ~150 us — 15 % of a
// single-image forward — although parameters rarely change between two inference calls.  Whether they did is decided ON THE DEVICE
// (no host bookkeeping can see every writer: optimizer kernels, `.data` edits, checkpoint loads, running-statistics updates): a
// position-weighted 64-bit checksum of the fp32 master buffer and of every layer's running statistics (integer adds: order-free,
// exact) is compared with the one the current folded weights were made from; cast and fold exit at once when they agree.
// state: 4 x u64 device words {checksum being accumulated, checksum of the folded weights, folds done, calls}.
#define PFR_HASH_BLOCKS 512
__global__ __launch_bounds__(256) void param_hash_kernel(const uint32_t* __restrict__ master, size_t n_master,
                                                         const FoldDesc* __restrict__ descs, unsigned long long* __restrict__ state) {
  unsigned long long h = 0;
  if (blockIdx.x < PFR_HASH_BLOCKS) {
    const size_t n4 = n_master / 4, stride = (size_t)PFR_HASH_BLOCKS * 256;
    auto mix = [&](const u32x4 v, size_t i) {
      const uint32_t k = (uint32_t)(8 * i + 1);   // odd 32-bit multipliers 2*(4i+e)+1: one v_mad_u64_u32 per word
      h += (unsigned long long)v[0] * k;
      h += (unsigned long long)v[1] * (k + 2u);
      h += (unsigned long long)v[2] * (k + 4u);
      h += (unsigned long long)v[3] * (k + 6u);
    };
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {   // four 16-byte loads in flight per lane
      const u32x4* q = reinterpret_cast<const u32x4*>(master) + i;
      const u32x4 v0 = __builtin_nontemporal_load(q), v1 = __builtin_nontemporal_load(q + stride);
      const u32x4 v2 = __builtin_nontemporal_load(q + 2 * stride), v3 = __builtin_nontemporal_load(q + 3 * stride);
      mix(v0, i); mix(v1, i + stride); mix(v2, i + 2 * stride); mix(v3, i + 3 * stride);
    }
    for (; i < n4; i += stride) mix(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(master) + i), i);
    if (blockIdx.x == 0)
      for (size_t i = n4 * 4 + threadIdx.x; i < n_master; i += 256) h += (unsigned long long)master[i] * (uint32_t)(2 * i + 1);
  } else {
    const FoldDesc d = descs[blockIdx.x - PFR_HASH_BLOCKS];
    const unsigned long long base = ((unsigned long long)(blockIdx.x - PFR_HASH_BLOCKS + 1) << 40) | 1ull;
    for (long c = threadIdx.x; c < d.cout; c += 256) {
      h += (unsigned long long)__float_as_uint(d.rm[c]) * (base + 4ull * c);
      h += (unsigned long long)__float_as_uint(d.rv[c]) * (base + 4ull * c + 2);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
  // ONE atomic per workgroup (thousands of same-address atomics serialise: 4 per workgroup cost 40 us of the kernel's 62)
  __shared__ unsigned long long part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&state[0], part[0] + part[1] + part[2] + part[3]);
}
__global__ void param_hash_commit_kernel(unsigned long long* state) {
  if (state[0] != state[1]) { state[1] = state[0]; state[2] += 1; }
  state[0] = 0;
  state[3] += 1;
}
template <typename T>
__global__ __launch_bounds__(256) void fold_bn_if_kernel(const FoldDesc* __restrict__ descs, const unsigned long long* __restrict__ state) {
  if (state[0] == state[1]) return;
  const FoldDesc d = descs[blockIdx.y];
  // one row (cout) per wave pass: the per-row scale is computed once, the row is walked in 16-byte pieces where alignment allows
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long co = (long)blockIdx.x * 4 + wave; co < d.cout; co += (long)gridDim.x * 4) {
    const float sc = (d.gamma ? d.gamma[co] : 1.f) * rsqrtf(d.rv[co] + d.eps);
    if (lane == 0) d.bout[co] = (d.beta ? d.beta[co] : 0.f) - d.rm[co] * sc;
    T* out = reinterpret_cast<T*>(d.wout) + co * d.k;
    if (d.src_f32) {
      const float* src = reinterpret_cast<const float*>(d.src) + co * d.k;
      for (long i = lane; i < d.k; i += 64) out[i] = from_f32<T>(src[i] * sc);
    } else {
      const T* src = reinterpret_cast<const T*>(d.src) + co * d.k;
      for (long i = lane; i < d.k; i += 64) out[i] = from_f32<T>(to_f32(src[i]) * sc);
    }
  }
}
template <typename TI, typename TOo>
__global__ void cast_if_kernel(const TI* __restrict__ x, TOo* __restrict__ y, size_t n, const unsigned long long* __restrict__ state) {
  if (state[0] == state[1]) return;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = from_f32<TOo>(to_f32(x[i]));
}
extern "C" int pfr_fold_bn_cached(const void* descs, int ndesc, int dtype, const float* master, size_t n_master, void* shadow,
                                  unsigned long long* state, hipStream_t st) {
  PFR_CHECK_ARG(descs && ndesc > 0 && master && state, "pfr_fold_bn_cached: bad args");
  PFR_CHECK_ARG(dtype == PFR_F32 || dtype == PFR_BF16, "pfr_fold_bn_cached: bad dtype %d", dtype);
  hipLaunchKernelGGL(param_hash_kernel, dim3(PFR_HASH_BLOCKS + (unsigned)ndesc), dim3(256), 0, st, (const uint32_t*)master, n_master,
                     (const FoldDesc*)descs, state);
  PFR_CHECK_LAUNCH();
  if (shadow && dtype == PFR_BF16) {
    unsigned blocks = (unsigned)((n_master + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((cast_if_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, st, master, (bf16_t*)shadow, n_master, state);
    PFR_CHECK_LAUNCH();
  }
  const dim3 grid(64, (unsigned)ndesc);
  if (dtype == PFR_BF16) hipLaunchKernelGGL(fold_bn_if_kernel<bf16_t>, grid, dim3(256), 0, st, (const FoldDesc*)descs, state);
  else hipLaunchKernelGGL(fold_bn_if_kernel<float>, grid, dim3(256), 0, st, (const FoldDesc*)descs, state);
  PFR_CHECK_LAUNCH();
  hipLaunchKernelGGL(param_hash_commit_kernel, dim3(1), dim3(1), 0, st, state);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// eval-mode BN: scale/shift from running statistics
__global__ void bn_eval_coeff_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                     float eps, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = rsqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * is;
  shift[c] = b - rm[c] * g * is;
}
extern "C" int pfr_bn_eval_coeff(int C, const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, float* scale, float* shift, hipStream_t st) {
  PFR_CHECK_ARG(running_mean && running_var && scale && shift, "pfr_bn_eval_coeff: null pointer");
  hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3((C + 255) / 256), dim3(256), 0, st, C, gamma, beta, running_mean, running_var, eps, scale, shift);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ---- apply: y = act( a1[c]*x1 + b1[c]  (+ a2[c]*x2 + b2[c]  |  + x2) )
// thread = (channel chunk, row lane): the per-channel coefficients are loaded once into registers, rows are streamed.
// HAS2 (second operand present) is a template parameter and the load batch ends with a scheduling barrier: see bn_bwd_reduce_kernel
template <typename T, bool HAS2>
__global__ __launch_bounds__(256) void bn_act_kernel(const T* __restrict__ x1, const float* __restrict__ a1,
                                                     const float* __restrict__ b1, const T* __restrict__ x2,
                                                     const float* __restrict__ a2, const float* __restrict__ b2,
                                                     T* __restrict__ y, unsigned char* __restrict__ mask, size_t rows, int C,
                                                     int cw, int rl, int cpr, int relu) {
  constexpr int KP = DT<T>::KPACK;
  const int col = threadIdx.x % cw, rlane = threadIdx.x / cw;
  const int cglob = blockIdx.y * cw + col;
  if (cglob >= cpr) return;
  float A1[KP], B1[KP], A2[KP], B2[KP];
#pragma unroll
  for (int e = 0; e < KP; ++e) {
    A1[e] = a1[cglob * KP + e];
    B1[e] = b1[cglob * KP + e];
    A2[e] = a2 ? a2[cglob * KP + e] : 1.f;
    B2[e] = a2 ? b2[cglob * KP + e] : 0.f;
  }
  auto body = [&](u32x4 v1, u32x4 v2, size_t row) {
    const size_t off = row * C + cglob * KP;
    float f[KP], g[KP];
    Chunk<T>::unpack(v1, f);
    if constexpr (HAS2) Chunk<T>::unpack(v2, g);
    unsigned bits = 0;
#pragma unroll
    for (int e = 0; e < KP; ++e) {
      float z = fmaf(f[e], A1[e], B1[e]);
      if constexpr (HAS2) z += fmaf(g[e], A2[e], B2[e]);
      bits |= (z > 0.f ? 1u : 0u) << e;
      f[e] = relu ? fmaxf(z, 0.f) : z;
    }
    st16(y + off, Chunk<T>::pack(f));
    if (mask) mask[row * cpr + cglob] = (unsigned char)bits;   // ReLU mask for the backward pass: 1 bit instead of 16
  };
  // four rows per thread in flight: at ~2 µs HBM latency the load queue, not the bandwidth, limits a 1-deep loop
  const size_t step = (size_t)gridDim.x * rl;
  size_t r = (size_t)blockIdx.x * rl + rlane;
  for (; r + 3 * step < rows; r += 4 * step) {
    u32x4 v1[4], v2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t off = (r + u * step) * C + cglob * KP;
      v1[u] = ld16_nt(x1 + off);   // the conv output is not needed again before the backward pass
      if constexpr (HAS2) v2[u] = ld16_nt(x2 + off);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) body(v1[u], v2[u], r + u * step);
  }
  for (; r < rows; r += step) {
    const size_t off = r * C + cglob * KP;
    u32x4 v2 = {};
    if constexpr (HAS2) v2 = ld16(x2 + off);
    body(ld16(x1 + off), v2, r);
  }
}

extern "C" int pfr_bn_act_mask(const void* x1, const float* a1, const float* b1, const void* x2, const float* a2,
                               const float* b2, void* y, unsigned char* mask, int dtype, long rows, int C, int relu,
                               hipStream_t st);
extern "C" int pfr_bn_act(const void* x1, const float* a1, const float* b1, const void* x2, const float* a2,
                          const float* b2, void* y, int dtype, long rows, int C, int relu, hipStream_t st) {
  return pfr_bn_act_mask(x1, a1, b1, x2, a2, b2, y, nullptr, dtype, rows, C, relu, st);
}
extern "C" int pfr_bn_act_mask(const void* x1, const float* a1, const float* b1, const void* x2, const float* a2,
                               const float* b2, void* y, unsigned char* mask, int dtype, long rows, int C, int relu,
                               hipStream_t st) {
  PFR_CHECK_ARG(x1 && a1 && b1 && y, "pfr_bn_act: null pointer");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  PFR_CHECK_ARG(C % kp == 0, "pfr_bn_act: C %% %d != 0", kp);
  // workgroups per launch (measured with the load batches intact): three-stream passes (two reads + one write) are fastest with ONE
  // workgroup per CU (256), the read-write pass with two; 384 or >= 768 lose 1-6 %
  ColGeom g = col_geom(C, kp, (size_t)rows, x2 ? 256 : 512);
#define PFR_BNACT(TT, H2) hipLaunchKernelGGL((bn_act_kernel<TT, H2>), dim3(g.gx, g.gy), dim3(256), 0, st, (const TT*)x1, a1, b1, (const TT*)x2, a2, b2, (TT*)y, mask, (size_t)rows, C, g.cw, g.rl, g.cpr, relu)
  if (dtype == PFR_BF16) { if (x2) PFR_BNACT(bf16_t, true); else PFR_BNACT(bf16_t, false); }
  else { if (x2) PFR_BNACT(float, true); else PFR_BNACT(float, false); }
#undef PFR_BNACT
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ---- backward reduce: g = dout * mask ; partials of Σ g and Σ g·x̂  (x̂ = (x − mean)·invstd)
// mask_mode 0: none; 1: out > 0 (materialised post-activation tensor); 2: scale·x + shift > 0 (recomputed)
// MASK is a template parameter: with a run-time mode the optional loads (`if (mode == 3) bits = …`) sit in branches and hipcc drains
// the memory queue (vmcnt(0)) behind the first row of every four-row batch — two round trips per batch instead of one
// (round 4: the FIN variant — reduce + finalize in one launch through a last-workgroup hand-over — measured slower than the separate
// 6.7 us finalize launch it removed, profiles/r03_finalize_fusion.txt, and was retired)
template <typename T, int MASK>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                            const T* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ part, size_t rows, int C,
                                                            int cw, int rl, int cpr) {
  constexpr int mask_mode = MASK;
  constexpr int KP = DT<T>::KPACK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int col = threadIdx.x % cw, rlane = threadIdx.x / cw;
  const int cglob = blockIdx.y * cw + col;
  float v[2][KP];
#pragma unroll
  for (int e = 0; e < KP; ++e) { v[0][e] = 0.f; v[1][e] = 0.f; }
  if (cglob < cpr) {
    float mu[KP], is[KP], sc[KP], sh[KP];
#pragma unroll
    for (int e = 0; e < KP; ++e) {
      mu[e] = mean[cglob * KP + e];
      is[e] = invstd[cglob * KP + e];
      sc[e] = mask_mode == 2 ? scale[cglob * KP + e] : 0.f;
      sh[e] = mask_mode == 2 ? shift[cglob * KP + e] : 0.f;
    }
    const unsigned char* mk = reinterpret_cast<const unsigned char*>(out);   // mask_mode 3: one byte per (row, chunk)
    auto body = [&](u32x4 vg, u32x4 vx, u32x4 vo, unsigned bits) {
      float g[KP], xv[KP], o[KP];
      Chunk<T>::unpack(vg, g);
      Chunk<T>::unpack(vx, xv);
      if (mask_mode == 1) Chunk<T>::unpack(vo, o);
#pragma unroll
      for (int e = 0; e < KP; ++e) {
        float gg = g[e];
        if (mask_mode == 1) gg = o[e] > 0.f ? gg : 0.f;
        if (mask_mode == 3) gg = (bits >> e) & 1u ? gg : 0.f;
        if (mask_mode == 2) gg = fmaf(xv[e], sc[e], sh[e]) > 0.f ? gg : 0.f;
        v[0][e] += gg;
        v[1][e] = fmaf(gg, (xv[e] - mu[e]) * is[e], v[1][e]);
      }
    };
    const size_t step = (size_t)gridDim.x * rl;
    size_t r = (size_t)blockIdx.x * rl + rlane;
#ifndef PFR_BNR_ROWS
#define PFR_BNR_ROWS 4
#endif
    constexpr int UB = PFR_BNR_ROWS;
    for (; r + (UB - 1) * step < rows; r += UB * step) {   // UB rows per thread in flight
      u32x4 vg[UB], vx[UB], vo[UB];
      unsigned bits[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const size_t off = (r + u * step) * C + cglob * KP;
        bits[u] = 0;
        vg[u] = ld16_nt(dout + off);
        vx[u] = ld16_nt(x + off);
        if (mask_mode == 1) vo[u] = ld16(out + off);
        if (mask_mode == 3) bits[u] = mk[(r + u * step) * cpr + cglob];
      }
      __builtin_amdgcn_sched_barrier(0);   // all loads of the batch are issued before any of them is consumed
#pragma unroll
      for (int u = 0; u < UB; ++u) body(vg[u], vx[u], vo[u], bits[u]);
    }
    for (; r < rows; r += step) {
      const size_t off = r * C + cglob * KP;
      u32x4 vo = {};
      if (mask_mode == 1) vo = ld16(out + off);
      body(ld16(dout + off), ld16(x + off), vo, mask_mode == 3 ? mk[r * cpr + cglob] : 0u);
    }
  }
  col_block_reduce<2, KP>(v, lds, cw, rl, col, rlane, cglob, cpr, part + (size_t)blockIdx.x * 2 * C, C);
}

extern "C" int pfr_bn_bwd_reduce(const void* dout, const void* out, const void* x, const float* mean,
                                 const float* invstd, const float* scale, const float* shift, int mask_mode, int dtype,
                                 long rows, int C, float* part, hipStream_t st) {
  PFR_CHECK_ARG(dout && x && mean && invstd && part, "pfr_bn_bwd_reduce: null pointer");
  PFR_CHECK_ARG((mask_mode != 1 && mask_mode != 3) || out, "pfr_bn_bwd_reduce: mask_mode 1 / 3 needs out / the bit mask");
  PFR_CHECK_ARG(mask_mode != 2 || (scale && shift), "pfr_bn_bwd_reduce: mask_mode 2 needs scale/shift");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  PFR_CHECK_ARG(C % kp == 0, "pfr_bn_bwd_reduce: C %% %d != 0", kp);
  ColGeom g = col_geom(C, kp, (size_t)rows);
  const size_t shb = (size_t)256 * 2 * kp * sizeof(float);
#define PFR_BNR(TT, M) hipLaunchKernelGGL((bn_bwd_reduce_kernel<TT, M>), dim3(g.gx, g.gy), dim3(256), shb, st, (const TT*)dout, (const TT*)out, (const TT*)x, mean, invstd, scale, shift, part, (size_t)rows, C, g.cw, g.rl, g.cpr)
#define PFR_BNR4(TT) do { switch (mask_mode) { case 0: PFR_BNR(TT, 0); break; case 1: PFR_BNR(TT, 1); break; case 2: PFR_BNR(TT, 2); break; default: PFR_BNR(TT, 3); } } while (0)
  PFR_CHECK_ARG(mask_mode >= 0 && mask_mode <= 3, "pfr_bn_bwd_reduce: bad mask_mode");
  if (dtype == PFR_BF16) PFR_BNR4(bf16_t); else PFR_BNR4(float);
#undef PFR_BNR4
#undef PFR_BNR
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}


// ---- backward finalise: partials → dgamma, dbeta and the per-channel coefficients of
//      dx = cg·g + cx·x + c0     (cg = γ·invstd, cx = −γ·invstd²·Σgx̂/M, c0 = −γ·invstd·Σg/M − cx·mean)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nparts, int C, float count,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ coef,
                                                              int accumulate) {
  // block: 4 channels x 64 part-lanes (the partial rows are summed by many short strided loops in parallel)
  __shared__ float l1[64][5], l2[64][5];
  const int cl = threadIdx.x & 3, pl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  float a = 0.f, b = 0.f;
  // (latency-critical like bn_finalize_kernel: batches of 8 parts per thread are requested together from clamped addresses, and the
  // per-channel operands of the tail before the reduction)
  const int cc = c < C ? c : C - 1;
  const float g = gamma ? gamma[cc] : 1.f, is = invstd[cc], mu = mean[cc];
  const float dg0 = (dgamma && accumulate) ? dgamma[cc] : 0.f, db0 = (dbeta && accumulate) ? dbeta[cc] : 0.f;
  for (int i0 = pl; i0 < nparts; i0 += 64 * 8) {
    float va[8], vb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + 64 * k, ic = i < nparts ? i : pl;
      va[k] = part[((size_t)ic * 2 + 0) * C + cc];
      vb[k] = part[((size_t)ic * 2 + 1) * C + cc];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (c < C && i0 + 64 * k < nparts) {
        a += va[k];
        b += vb[k];
      }
  }
  l1[pl][cl] = a;
  l2[pl][cl] = b;
  __syncthreads();
  if (pl == 0 && c < C) {
    float sg = 0.f, sgx = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) { sg += l1[i][cl]; sgx += l2[i][cl]; }
    if (dgamma) dgamma[c] = accumulate ? dg0 + sgx : sgx;
    if (dbeta) dbeta[c] = accumulate ? db0 + sg : sg;
    const float cg = g * is;
    const float cx = -g * is * is * sgx / count;
    const float c0 = -g * is * sg / count - cx * mu;
    coef[c] = cg;
    coef[C + c] = cx;
    coef[2 * C + c] = c0;
  }
}

extern "C" int pfr_bn_bwd_finalize(const float* part, int nparts, int C, float count, const float* gamma,
                                   const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef,
                                   int accumulate, hipStream_t st) {
  PFR_CHECK_ARG(part && mean && invstd && coef, "pfr_bn_bwd_finalize: null pointer");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, part, nparts, C, count, gamma, mean, invstd, dgamma, dbeta, coef, accumulate);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ---- backward apply: g = dout·mask ; dx = cg·g + cx·x + c0 ; optional gres = g  (gradient of the residual input)
template <typename T, int MASK>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                           const T* __restrict__ x, const float* __restrict__ coef,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           T* __restrict__ dx, T* __restrict__ gres,
                                                           size_t rows, int C, int cw, int rl, int cpr) {
  constexpr int mask_mode = MASK;   // (compile time: see bn_bwd_reduce_kernel)
  constexpr int KP = DT<T>::KPACK;
  const int col = threadIdx.x % cw, rlane = threadIdx.x / cw;
  const int cglob = blockIdx.y * cw + col;
  if (cglob >= cpr) return;
  float cg[KP], cx[KP], c0[KP], sc[KP], sh[KP];
#pragma unroll
  for (int e = 0; e < KP; ++e) {
    const int c = cglob * KP + e;
    cg[e] = coef[c];
    cx[e] = coef[C + c];
    c0[e] = coef[2 * C + c];
    sc[e] = mask_mode == 2 ? scale[c] : 0.f;
    sh[e] = mask_mode == 2 ? shift[c] : 0.f;
  }
  const unsigned char* mk = reinterpret_cast<const unsigned char*>(out);
  auto body = [&](u32x4 vg, u32x4 vx, u32x4 vo, unsigned bits, size_t off) {
    float g[KP], xv[KP], o[KP];
    Chunk<T>::unpack(vg, g);
    Chunk<T>::unpack(vx, xv);
    if (mask_mode == 1) Chunk<T>::unpack(vo, o);
#pragma unroll
    for (int e = 0; e < KP; ++e) {
      float gg = g[e];
      if (mask_mode == 1) gg = o[e] > 0.f ? gg : 0.f;
      if (mask_mode == 3) gg = (bits >> e) & 1u ? gg : 0.f;
      if (mask_mode == 2) gg = fmaf(xv[e], sc[e], sh[e]) > 0.f ? gg : 0.f;
      g[e] = gg;
      xv[e] = fmaf(cg[e], gg, fmaf(cx[e], xv[e], c0[e]));
    }
    st16(dx + off, Chunk<T>::pack(xv));
    if (gres) st16(gres + off, Chunk<T>::pack(g));
  };
  const size_t step = (size_t)gridDim.x * rl;
  size_t r = (size_t)blockIdx.x * rl + rlane;
  for (; r + 3 * step < rows; r += 4 * step) {   // four rows per thread in flight (dx may alias dout: loads precede stores)
    u32x4 vg[4], vx[4], vo[4];
    unsigned bits[4] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t off = (r + u * step) * C + cglob * KP;
      vg[u] = ld16_nt(dout + off);
      vx[u] = ld16_nt(x + off);
      if (mask_mode == 1) vo[u] = ld16(out + off);
      if (mask_mode == 3) bits[u] = mk[(r + u * step) * cpr + cglob];
    }
    __builtin_amdgcn_sched_barrier(0);   // all loads of the batch are issued before any of them is consumed
#pragma unroll
    for (int u = 0; u < 4; ++u) body(vg[u], vx[u], vo[u], bits[u], (r + u * step) * C + cglob * KP);
  }
  for (; r < rows; r += step) {
    const size_t off = r * C + cglob * KP;
    u32x4 vo = {};
    if (mask_mode == 1) vo = ld16(out + off);
    body(ld16(dout + off), ld16(x + off), vo, mask_mode == 3 ? mk[r * cpr + cglob] : 0u, off);
  }
}

extern "C" int pfr_bn_bwd_apply(const void* dout, const void* out, const void* x, const float* coef, const float* scale,
                                const float* shift, int mask_mode, void* dx, void* gres, int dtype, long rows, int C,
                                hipStream_t st) {
  PFR_CHECK_ARG(dout && x && coef && dx, "pfr_bn_bwd_apply: null pointer");
  PFR_CHECK_ARG(mask_mode >= 0 && mask_mode <= 3, "pfr_bn_bwd_apply: bad mask_mode");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  PFR_CHECK_ARG(C % kp == 0, "pfr_bn_bwd_apply: C %% %d != 0", kp);
  ColGeom g = col_geom(C, kp, (size_t)rows, 256);   // one workgroup per CU (see pfr_bn_act_mask)
  hipEvent_t stop = pfr_tls_stop_event;
  pfr_tls_stop_event = nullptr;
#define PFR_BNA(TT, M)                                                                                                                        \
  do {                                                                                                                                        \
    if (stop) hipExtLaunchKernelGGL((bn_bwd_apply_kernel<TT, M>), dim3(g.gx, g.gy), dim3(256), 0, st, nullptr, stop, 0, (const TT*)dout, (const TT*)out, (const TT*)x, coef, scale, shift, (TT*)dx, (TT*)gres, (size_t)rows, C, g.cw, g.rl, g.cpr); \
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, M>), dim3(g.gx, g.gy), dim3(256), 0, st, (const TT*)dout, (const TT*)out, (const TT*)x, coef, scale, shift, (TT*)dx, (TT*)gres, (size_t)rows, C, g.cw, g.rl, g.cpr); \
  } while (0)
#define PFR_BNA4(TT) do { switch (mask_mode) { case 0: PFR_BNA(TT, 0); break; case 1: PFR_BNA(TT, 1); break; case 2: PFR_BNA(TT, 2); break; default: PFR_BNA(TT, 3); } } while (0)
  if (dtype == PFR_BF16) PFR_BNA4(bf16_t); else PFR_BNA4(float);
#undef PFR_BNA4
#undef PFR_BNA
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// y[c][r] = x[r][c]  (row-major [rows][cols] → [cols][rows]); 64x64 tiles through LDS, 2- or 4-byte elements
template <typename T>
__global__ __launch_bounds__(256) void transpose2d_kernel(const T* __restrict__ x, T* __restrict__ y, int rows, int cols) {
  __shared__ T tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = x[(size_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 64; i += 4)
    if (c0 + i < cols && r0 + tx < rows) y[(size_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}
extern "C" int pfr_transpose2d(const void* x, void* y, int dtype, int rows, int cols, hipStream_t st) {
  PFR_CHECK_ARG(x && y && rows > 0 && cols > 0, "pfr_transpose2d: bad args");
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (dtype == PFR_BF16) hipLaunchKernelGGL(transpose2d_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, rows, cols);
  else hipLaunchKernelGGL(transpose2d_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, rows, cols);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ---- space-to-depth form of the ResNet stem (conv 7x7, stride 2, pad 3, C = 3):
//        out[oh][ow] = Σ_{r,s} x[2oh−3+r][2ow−3+s]·w[r][s]  =  Σ_{a,b<4} Σ_{p,q<2} xs[oh−2+a][ow−2+b][(p,q,c)]·ws[a][b][(p,q,c)]
//      with xs[i][j][(p*2+q)*C + c] = x[c][2i+p][2j+q] and ws[a][b][(p*2+q)*C+c] = w[2a+p−1][2b+q−1][c] (0 outside 0..6):
//      a 4x4 stride-1 pad-2 conv over a half-resolution 16-channel image — 256 instead of 392 k-columns per output pixel
//      and 32-byte instead of 16-byte gather pieces (measured: forward 423 → 305 µs, weight gradient 622 → 320 µs).
template <typename T>
__global__ __launch_bounds__(256) void s2d_input_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int H, int W,
                                                        int Cp) {
  const int H2 = H >> 1, W2 = W >> 1;
  const size_t npix = (size_t)N * H2 * W2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
    const int j2 = (int)(i % W2);
    const int i2 = (int)((i / W2) % H2);
    const size_t n = i / ((size_t)W2 * H2);
    constexpr int KP = DT<T>::KPACK;
    T* o = y + i * Cp;
    for (int c0 = 0; c0 < Cp; c0 += KP) {   // 16-byte stores
      float v[KP];
#pragma unroll
      for (int e = 0; e < KP; ++e) {
        const int ch = c0 + e;
        v[e] = 0.f;
        if (ch < 4 * C) {
          const int pq = ch / C, c = ch - pq * C;
          v[e] = x[((n * C + c) * H + 2 * i2 + (pq >> 1)) * W + 2 * j2 + (pq & 1)];
        }
      }
      st16(o + c0, Chunk<T>::pack(v));
    }
  }
}
// The stem's own form (3 input channels, 16-channel pixels, W % 4 == 0): a thread makes TWO horizontally adjacent space-to-depth pixels from
// six 16-byte loads (4 consecutive columns of the 2 rows of each of the 3 planes) and writes their 64 contiguous bytes; the generic kernel above
// issues 12 scalar loads per pixel (round 6: 78 -> ~50 us for 256 x 3 x 224 x 224, bit-identical).
__global__ __launch_bounds__(256) void s2d_input3_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int N, int H, int W) {
  const int H2 = H >> 1, W4 = W >> 2;
  const size_t npair = (size_t)N * H2 * W4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npair; i += (size_t)gridDim.x * 256) {
    const int j4 = (int)(i % W4);
    const int i2 = (int)((i / W4) % H2);
    const size_t n = i / ((size_t)W4 * H2);
    f32x4 v[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 2; ++r)
        v[c][r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + ((n * 3 + c) * H + 2 * i2 + r) * W + 4 * j4));
    bf16_t* o = y + ((n * H2 + i2) * (size_t)(W >> 1) + 2 * j4) * 16;
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      float a[8], b[8];
      // channel ch = pq * 3 + c, pq = (row parity << 1) | column parity
#pragma unroll
      for (int ch = 0; ch < 16; ++ch) {
        float val = 0.f;
        if (ch < 12) {
          const int pq = ch / 3, c = ch - pq * 3;
          val = v[c][pq >> 1][2 * px + (pq & 1)];
        }
        if (ch < 8) a[ch] = val; else b[ch - 8] = val;
      }
      st16(o + px * 16, Chunk<bf16_t>::pack(a));
      st16(o + px * 16 + 8, Chunk<bf16_t>::pack(b));
    }
  }
}
template <typename T>
__global__ void s2d_weight_kernel(const float* __restrict__ w, T* __restrict__ ws, int Co, int C, int Cp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over Co*4*4*Cp
  if (i >= Co * 16 * Cp) return;
  const int ch = i % Cp, b = (i / Cp) % 4, a = (i / (4 * Cp)) % 4, co = i / (16 * Cp);
  float v = 0.f;
  if (ch < 4 * C) {
    const int pq = ch / C, c = ch - pq * C;
    const int r = 2 * a + (pq >> 1) - 1, s = 2 * b + (pq & 1) - 1;
    if (r >= 0 && r < 7 && s >= 0 && s < 7) v = w[((co * 7 + r) * 7 + s) * C + c];
  }
  ws[i] = from_f32<T>(v);
}
__global__ void s2d_wgrad_kernel(const float* __restrict__ dws, float* __restrict__ dw, int Co, int C, int Cp, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over Co*7*7*C
  if (i >= Co * 49 * C) return;
  const int c = i % C, s = (i / C) % 7, r = (i / (7 * C)) % 7, co = i / (49 * C);
  const int a = (r + 1) >> 1, p = (r + 1) & 1, b = (s + 1) >> 1, q = (s + 1) & 1;
  const float v = dws[((co * 4 + a) * 4 + b) * Cp + (p * 2 + q) * C + c];
  dw[i] = accumulate ? dw[i] + v : v;
}
extern "C" int pfr_s2d_input(const float* x, void* y, int dtype, int N, int C, int H, int W, int Cp, hipStream_t st) {
  PFR_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0 && Cp >= 4 * C && Cp % (dtype == PFR_BF16 ? 8 : 4) == 0, "pfr_s2d_input: bad geometry");
  const size_t npix = (size_t)N * (H / 2) * (W / 2);
  unsigned blocks = (unsigned)((npix + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  if (dtype == PFR_BF16 && C == 3 && Cp == 16 && W % 4 == 0 && H % 2 == 0) {
    const size_t npair = npix / 2;
    unsigned b2 = (unsigned)((npair + 255) / 256);
    if (b2 > 16384) b2 = 16384;
    hipLaunchKernelGGL(s2d_input3_kernel, dim3(b2), dim3(256), 0, st, x, (bf16_t*)y, N, H, W);
  } else if (dtype == PFR_BF16) hipLaunchKernelGGL(s2d_input_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, x, (bf16_t*)y, N, C, H, W, Cp);
  else hipLaunchKernelGGL(s2d_input_kernel<float>, dim3(blocks), dim3(256), 0, st, x, (float*)y, N, C, H, W, Cp);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
extern "C" int pfr_s2d_weight(const float* w, void* ws, int dtype, int Cout, int C, int Cp, hipStream_t st) {
  PFR_CHECK_ARG(w && ws && Cout > 0 && C > 0 && Cp >= 4 * C, "pfr_s2d_weight: bad args");
  const int n = Cout * 16 * Cp;
  if (dtype == PFR_BF16) hipLaunchKernelGGL(s2d_weight_kernel<bf16_t>, dim3((n + 255) / 256), dim3(256), 0, st, w, (bf16_t*)ws, Cout, C, Cp);
  else hipLaunchKernelGGL(s2d_weight_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, st, w, (float*)ws, Cout, C, Cp);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
extern "C" int pfr_s2d_wgrad(const float* dws, float* dw, int Cout, int C, int Cp, int accumulate, hipStream_t st) {
  PFR_CHECK_ARG(dws && dw && Cout > 0 && C > 0 && Cp >= 4 * C, "pfr_s2d_wgrad: bad args");
  const int n = Cout * 49 * C;
  hipLaunchKernelGGL(s2d_wgrad_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dws, dw, Cout, C, Cp, accumulate);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// elementwise y = a + b (gradient joins of the residual graph)
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, size_t nchunks) {
  constexpr int KP = DT<T>::KPACK;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nchunks; i += stride) {
    float f[KP], g[KP];
    Chunk<T>::unpack(ld16(a + i * KP), f);
    Chunk<T>::unpack(ld16(b + i * KP), g);
#pragma unroll
    for (int e = 0; e < KP; ++e) f[e] += g[e];
    st16(y + i * KP, Chunk<T>::pack(f));
  }
}
extern "C" int pfr_add(const void* a, const void* b, void* y, int dtype, size_t n, hipStream_t st) {
  PFR_CHECK_ARG(a && b && y, "pfr_add: null pointer");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  PFR_CHECK_ARG(n % kp == 0, "pfr_add: n %% %d != 0", kp);
  const size_t nch = n / kp;
  unsigned blocks = (unsigned)((nch + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (dtype == PFR_BF16)
    hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, nch);
  else
    hipLaunchKernelGGL(add_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)y, nch);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ------------------------------------------------------------------------------------------------
// fused BN-apply + ReLU + MaxPool(3,2,1) forward (stem) with argmax index, and its backward
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, T* __restrict__ y,
                                                                  uint8_t* __restrict__ idx, int N, int H, int W, int C,
                                                                  int OH, int OW, int relu) {
  constexpr int KP = DT<T>::KPACK;
  const int cpr = C / KP;
  const size_t nch = (size_t)N * OH * OW * cpr;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nch; i += stride) {
    const int cc = (int)(i % cpr);
    size_t pix = i / cpr;
    const int ow = (int)(pix % OW); pix /= OW;
    const int oh = (int)(pix % OH);
    const int n = (int)(pix / OH);
    float best[KP], sc[KP], sh[KP];
    int bi[KP];
#pragma unroll
    for (int e = 0; e < KP; ++e) {
      best[e] = -INFINITY; bi[e] = 0;
      sc[e] = scale ? scale[cc * KP + e] : 1.f;
      sh[e] = shift ? shift[cc * KP + e] : 0.f;
    }
    // the nine (clamped) loads are issued before any is consumed; out-of-range taps are skipped by a flag
    u32x4 tv[9];
    bool tok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ih = oh * 2 - 1 + t / 3, iw = ow * 2 - 1 + t % 3;
      tok[t] = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      const int ihc = min(max(ih, 0), H - 1), iwc = min(max(iw, 0), W - 1);
      tv[t] = ld16(x + (((size_t)n * H + ihc) * W + iwc) * C + cc * KP);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float f[KP];
      Chunk<T>::unpack(tv[t], f);
#pragma unroll
      for (int e = 0; e < KP; ++e) {
        float z = fmaf(f[e], sc[e], sh[e]);
        if (relu) z = fmaxf(z, 0.f);
        z = to_f32(from_f32<T>(z));
        if (tok[t] && z > best[e]) { best[e] = z; bi[e] = t; }
      }
    }
    st16(y + i * KP, Chunk<T>::pack(best));
    if (idx) {
      if constexpr (KP == 8) {
        uint64_t pk = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) pk |= (uint64_t)bi[e] << (8 * e);
        *reinterpret_cast<uint64_t*>(idx + i * 8) = pk;
      } else {
        uint32_t pk = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk |= (uint32_t)bi[e] << (8 * e);
        *reinterpret_cast<uint32_t*>(idx + i * 4) = pk;
      }
    }
  }
}

extern "C" int pfr_bn_relu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y, uint8_t* idx,
                                       int dtype, int N, int H, int W, int C, int relu, hipStream_t st) {
  PFR_CHECK_ARG(x && y, "pfr_bn_relu_maxpool_fwd: null pointer");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  PFR_CHECK_ARG(C % kp == 0, "pfr_bn_relu_maxpool_fwd: C %% %d != 0", kp);
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const size_t nch = (size_t)N * OH * OW * (C / kp);
  unsigned blocks = (unsigned)((nch + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  if (dtype == PFR_BF16)
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, scale, shift, (bf16_t*)y, idx, N, H, W, C, OH, OW, relu);
  else
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, scale, shift, (float*)y, idx, N, H, W, C, OH, OW, relu);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// backward: dz[n,h,w,c] = Σ over the ≤4 windows covering (h,w) whose argmax is this tap of dy  (gather, no atomics)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          T* __restrict__ dz, int N, int H, int W, int C, int OH, int OW) {
  constexpr int KP = DT<T>::KPACK;
  const int cpr = C / KP;
  const size_t nch = (size_t)N * H * W * cpr;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nch; i += stride) {
    const int cc = (int)(i % cpr);
    size_t pix = i / cpr;
    const int w = (int)(pix % W); pix /= W;
    const int h = (int)(pix % H);
    const int n = (int)(pix / H);
    float acc[KP];
#pragma unroll
    for (int e = 0; e < KP; ++e) acc[e] = 0.f;
    // An input pixel (h, w) belongs to at most 2 x 2 windows: oh ∈ {(h+1)/2 (tap r = h+1-2oh ∈ {0,1}...)}.  Enumerate them
    // branch-free: window rows oh0 = (h+1)>>1 (tap r0 = h+1-2*oh0 ∈ {0,1}) and oh0-1 (tap r0+2, only if r0 == 0); same for
    // columns.  All four (clamped) loads are issued before any of them is used.
    const int oh0 = (h + 1) >> 1, r0 = h + 1 - 2 * oh0, ow0 = (w + 1) >> 1, s0 = w + 1 - 2 * ow0;
    u32x4 gv[4];
    uint64_t pk[4];
    int tap[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int dh = q >> 1, dw = q & 1;
      const int oh = oh0 - dh, ow = ow0 - dw;
      const int r = r0 + 2 * dh, sx = s0 + 2 * dw;
      ok[q] = oh >= 0 && oh < OH && ow >= 0 && ow < OW && r < 3 && sx < 3;
      tap[q] = r * 3 + sx;
      const int ohc = min(max(oh, 0), OH - 1), owc = min(max(ow, 0), OW - 1);
      const size_t o = (((size_t)n * OH + ohc) * OW + owc) * cpr + cc;
      gv[q] = ld16(dy + o * KP);
      if constexpr (KP == 8) pk[q] = *reinterpret_cast<const uint64_t*>(idx + o * 8);
      else pk[q] = *reinterpret_cast<const uint32_t*>(idx + o * 4);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float g[KP];
      Chunk<T>::unpack(gv[q], g);
#pragma unroll
      for (int e = 0; e < KP; ++e)
        acc[e] += (ok[q] && (int)((pk[q] >> (8 * e)) & 0xff) == tap[q]) ? g[e] : 0.f;   // (a select: the branchy form costs an exec-mask save / restore per element)
    }
    st16(dz + i * KP, Chunk<T>::pack(acc));
  }
}

// The same gather for even H, W in 2 x 2 INPUT-pixel blocks (round 5): the four pixels of a block lie in the windows (a, b), (a, b+1), (a+1, b),
// (a+1, b+1) only, so one thread loads those four (dy chunk, argmax bytes) pairs once — 8 loads per 4 outputs instead of 32, of which
// 14 went to clamped, unused windows — and adds them per pixel in the order of maxpool_bwd_kernel (window rows high to low, columns high to
// low): bit-identical.  Pixel (2a, 2b) takes tap 4 of W00; (2a, 2b+1) tap 3 of W01, tap 5 of W00; (2a+1, 2b) tap 1 of W10, tap 7 of W00;
// (2a+1, 2b+1) tap 0 of W11, 2 of W10, 6 of W01, 8 of W00.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd2x2_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                             T* __restrict__ dz, int N, int H, int W, int C, int OH, int OW) {
  constexpr int KP = DT<T>::KPACK;
  const int cpr = C / KP, H2 = H >> 1, W2 = W >> 1;
  const size_t nblk = (size_t)N * H2 * W2 * cpr;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nblk; i += stride) {
    const int cc = (int)(i % cpr);
    size_t t = i / cpr;
    const int b = (int)(t % W2); t /= W2;
    const int a = (int)(t % H2);
    const int n = (int)(t / H2);
    u32x4 gv[4];
    uint64_t pk[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {      // q = 2 * (window row a + (q >> 1)) + (window column b + (q & 1)); clamped loads, all issued before use
      const int oh = a + (q >> 1), ow = b + (q & 1);
      ok[q] = oh < OH && ow < OW;
      const size_t o = (((size_t)n * OH + min(oh, OH - 1)) * OW + min(ow, OW - 1)) * cpr + cc;
      gv[q] = ld16(dy + o * KP);
      if constexpr (KP == 8) pk[q] = *reinterpret_cast<const uint64_t*>(idx + o * 8);
      else pk[q] = *reinterpret_cast<const uint32_t*>(idx + o * 4);
    }
    float g[4][KP];
#pragma unroll
    for (int q = 0; q < 4; ++q) Chunk<T>::unpack(gv[q], g[q]);
    // per output pixel: (window, tap) pairs in the accumulation order of maxpool_bwd_kernel
    constexpr int NWIN[4] = {1, 2, 2, 4};
    constexpr int WIN[4][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, 0, 0, 0}, {3, 2, 1, 0}};
    constexpr int TAP[4][4] = {{4, 0, 0, 0}, {3, 5, 0, 0}, {1, 7, 0, 0}, {0, 2, 6, 8}};
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      float acc[KP];
#pragma unroll
      for (int e = 0; e < KP; ++e) acc[e] = 0.f;
#pragma unroll
      for (int k = 0; k < NWIN[px]; ++k) {
        const int q = WIN[px][k], tap = TAP[px][k];
#pragma unroll
        for (int e = 0; e < KP; ++e) acc[e] += (ok[q] && (int)((pk[q] >> (8 * e)) & 0xff) == tap) ? g[q][e] : 0.f;
      }
      const size_t o = (((size_t)n * H + 2 * a + (px >> 1)) * W + 2 * b + (px & 1)) * cpr + cc;
      st16(dz + o * KP, Chunk<T>::pack(acc));
    }
  }
}

extern "C" int pfr_maxpool_bwd(const void* dy, const uint8_t* idx, void* dz, int dtype, int N, int H, int W, int C,
                               hipStream_t st) {
  PFR_CHECK_ARG(dy && idx && dz, "pfr_maxpool_bwd: null pointer");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const size_t nch = (size_t)N * H * W * (C / kp);
  unsigned blocks = (unsigned)((nch + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  if (H % 2 == 0 && W % 2 == 0) {
    unsigned b2 = (unsigned)((nch / 4 + 255) / 256);
    if (b2 > 16384) b2 = 16384;
    if (dtype == PFR_BF16)
      hipLaunchKernelGGL(maxpool_bwd2x2_kernel<bf16_t>, dim3(b2), dim3(256), 0, st, (const bf16_t*)dy, idx, (bf16_t*)dz, N, H, W, C, OH, OW);
    else
      hipLaunchKernelGGL(maxpool_bwd2x2_kernel<float>, dim3(b2), dim3(256), 0, st, (const float*)dy, idx, (float*)dz, N, H, W, C, OH, OW);
    PFR_CHECK_LAUNCH();
    return PFR_OK;
  }
  if (dtype == PFR_BF16)
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)dy, idx, (bf16_t*)dz, N, H, W, C, OH, OW);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dy, idx, (float*)dz, N, H, W, C, OH, OW);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// (round 4: the variant that gathered the max-pool gradient inside the BatchNorm-backward passes — bit-identical, measured neutral —
// was retired; profiles/HISTORY.md)

// ------------------------------------------------------------------------------------------------
// global average pool [N][HW][C] -> [N][C] and its backward
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int HW, int C) {
  constexpr int KP = DT<T>::KPACK;
  const int cpr = C / KP;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * cpr) return;
  const int n = (int)(i / cpr), cc = (int)(i % cpr);
  float a[KP];
#pragma unroll
  for (int e = 0; e < KP; ++e) a[e] = 0.f;
  for (int p = 0; p < HW; ++p) {
    float f[KP];
    Chunk<T>::unpack(ld16(x + ((size_t)n * HW + p) * C + cc * KP), f);
#pragma unroll
    for (int e = 0; e < KP; ++e) a[e] += f[e];
  }
  const float inv = 1.f / HW;
#pragma unroll
  for (int e = 0; e < KP; ++e) a[e] *= inv;
  st16(y + i * KP, Chunk<T>::pack(a));
}
template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int HW, int C) {
  constexpr int KP = DT<T>::KPACK;
  const int cpr = C / KP;
  const size_t nch = (size_t)N * HW * cpr;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float inv = 1.f / HW;
  for (; i < nch; i += stride) {
    const int cc = (int)(i % cpr);
    const size_t n = i / ((size_t)HW * cpr);
    float f[KP];
    Chunk<T>::unpack(ld16(dy + (n * cpr + cc) * KP), f);
#pragma unroll
    for (int e = 0; e < KP; ++e) f[e] *= inv;
    st16(dx + i * KP, Chunk<T>::pack(f));
  }
}
extern "C" int pfr_avgpool_fwd(const void* x, void* y, int dtype, int N, int HW, int C, hipStream_t st) {
  PFR_CHECK_ARG(x && y, "pfr_avgpool_fwd: null pointer");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  const size_t n = (size_t)N * (C / kp);
  const unsigned blocks = (unsigned)((n + 63) / 64);
  if (dtype == PFR_BF16)
    hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(blocks), dim3(64), 0, st, (const bf16_t*)x, (bf16_t*)y, N, HW, C);
  else
    hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(blocks), dim3(64), 0, st, (const float*)x, (float*)y, N, HW, C);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
extern "C" int pfr_avgpool_bwd(const void* dy, void* dx, int dtype, int N, int HW, int C, hipStream_t st) {
  PFR_CHECK_ARG(dy && dx, "pfr_avgpool_bwd: null pointer");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  const size_t nch = (size_t)N * HW * (C / kp);
  unsigned blocks = (unsigned)((nch + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (dtype == PFR_BF16)
    hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, N, HW, C);
  else
    hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dy, (float*)dx, N, HW, C);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// column sums of a [rows][C] matrix in compute dtype → fp32 (bias gradients)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int rows, int C,
                                                     int accumulate) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  __shared__ float l[4][64];
  float a = 0.f;
  if (c < C) {   // four independent chains: one chain of rows / 4 dependent loads is a memory round trip each (19 us at 256 rows)
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int r = rl;
    for (; r + 12 < rows; r += 16) {
      a += to_f32(x[(size_t)r * C + c]);
      a1 += to_f32(x[(size_t)(r + 4) * C + c]);
      a2 += to_f32(x[(size_t)(r + 8) * C + c]);
      a3 += to_f32(x[(size_t)(r + 12) * C + c]);
    }
    for (; r < rows; r += 4) a += to_f32(x[(size_t)r * C + c]);
    a = (a + a1) + (a2 + a3);
  }
  l[rl][threadIdx.x & 63] = a;
  __syncthreads();
  if (rl == 0 && c < C) {
    a = l[0][threadIdx.x] + l[1][threadIdx.x] + l[2][threadIdx.x] + l[3][threadIdx.x];
    out[c] = accumulate ? out[c] + a : a;
  }
}
// final merge of fp32 partials [n][C]: block = 16 columns x 16 row lanes, four independent accumulators per thread (the
// single-pass kernel above walks n/4 dependent loads per thread, which is latency-bound for n in the hundreds)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int C,
                                                           int accumulate) {
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  __shared__ float l[16][17];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < C) {
    int r = rl;
    for (; r + 48 < n; r += 64) {
      a0 += part[(size_t)r * C + c];
      a1 += part[(size_t)(r + 16) * C + c];
      a2 += part[(size_t)(r + 32) * C + c];
      a3 += part[(size_t)(r + 48) * C + c];
    }
    for (; r < n; r += 16) a0 += part[(size_t)r * C + c];
  }
  l[rl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && c < C) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += l[r][cl];
    out[c] = accumulate ? out[c] + a : a;
  }
}
// partial column sums of row block blockIdx.y: grid (ceil(C/64), nblk) → part [nblk][C]
template <typename T>
__global__ __launch_bounds__(256) void colsum_part_kernel(const T* __restrict__ x, float* __restrict__ part, long rows, int C,
                                                          long rows_per_block) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  __shared__ float l[4][64];
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    long r = r0 + rl;
    for (; r + 28 < r1; r += 32) {   // eight rows per thread requested together
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = to_f32(x[(size_t)(r + 4 * k) * C + c]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; k += 2) { a0 += v[k]; a1 += v[k + 1]; }
    }
    for (; r + 4 < r1; r += 8) {
      a0 += to_f32(x[(size_t)r * C + c]);
      a1 += to_f32(x[(size_t)(r + 4) * C + c]);
    }
    for (; r < r1; r += 4) a0 += to_f32(x[(size_t)r * C + c]);
  }
  l[rl][threadIdx.x & 63] = a0 + a1;
  __syncthreads();
  if (rl == 0 && c < C) part[(size_t)blockIdx.y * C + c] = l[0][threadIdx.x] + l[1][threadIdx.x] + l[2][threadIdx.x] + l[3][threadIdx.x];
}

// 16-byte-chunk variant (C % KPACK == 0): thread = (channel chunk, row lane), grid-stride rows → part [gridDim.x][C]
template <typename T>
__global__ __launch_bounds__(256) void colsum_chunk_kernel(const T* __restrict__ x, float* __restrict__ part, size_t rows, int C,
                                                           int cw, int rl, int cpr) {
  constexpr int KP = DT<T>::KPACK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int col = threadIdx.x % cw, rlane = threadIdx.x / cw;
  const int cglob = blockIdx.y * cw + col;
  float v[2][KP];
#pragma unroll
  for (int e = 0; e < KP; ++e) { v[0][e] = 0.f; v[1][e] = 0.f; }
  if (cglob < cpr) {
    const size_t step = (size_t)gridDim.x * rl;
    size_t r = (size_t)blockIdx.x * rl + rlane;
    for (; r + 3 * step < rows; r += 4 * step) {   // four independent loads in flight
      u32x4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = ld16(x + (r + u * step) * C + cglob * KP);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float a[KP];
        Chunk<T>::unpack(q[u], a);
#pragma unroll
        for (int e = 0; e < KP; ++e) v[u & 1][e] += a[e];
      }
    }
    for (; r < rows; r += step) {
      float a[KP];
      Chunk<T>::unpack(ld16(x + r * C + cglob * KP), a);
#pragma unroll
      for (int e = 0; e < KP; ++e) v[0][e] += a[e];
    }
  }
  float w[1][KP];
#pragma unroll
  for (int e = 0; e < KP; ++e) w[0][e] = v[0][e] + v[1][e];
  col_block_reduce<1, KP>(w, lds, cw, rl, col, rlane, cglob, cpr, part + (size_t)blockIdx.x * C, C);
}

static long colsum_blocks(long rows, int C) {
  if (rows <= 256) return 0;  // single-kernel path
  long want = 2048 / ((C + 63) / 64);
  if (want < 8) want = 8;
  const long maxb = (rows + 255) / 256;
  return want < maxb ? want : maxb;
}
extern "C" long pfr_colsum_ws_floats(long rows, int C) {
  const long a = colsum_blocks(rows, C);
  return (a > 512 ? a : (a ? 512 : 0)) * C;   // chunked path: at most 512 row blocks
}

extern "C" int pfr_colsum(const void* x, int dtype, long rows, int C, float* out, int accumulate, float* workspace,
                          hipStream_t st) {
  PFR_CHECK_ARG(x && out, "pfr_colsum: null pointer");
  const long nblk = (workspace || rows > 2048) ? colsum_blocks(rows, C) : 0;   // ≤ 2048 rows may run without workspace
  if (nblk > 0) {
    PFR_CHECK_ARG(workspace, "pfr_colsum: workspace of pfr_colsum_ws_floats(rows, C) floats required for %ld rows", rows);
    const int kp = dtype == PFR_BF16 ? 8 : 4;
    if (C % kp == 0) {
      ColGeom g = col_geom(C, kp, (size_t)rows, 512);
      const size_t shb = (size_t)256 * kp * sizeof(float);
      if (dtype == PFR_BF16)
        hipLaunchKernelGGL(colsum_chunk_kernel<bf16_t>, dim3(g.gx, g.gy), dim3(256), shb, st, (const bf16_t*)x, workspace, (size_t)rows, C, g.cw, g.rl, g.cpr);
      else
        hipLaunchKernelGGL(colsum_chunk_kernel<float>, dim3(g.gx, g.gy), dim3(256), shb, st, (const float*)x, workspace, (size_t)rows, C, g.cw, g.rl, g.cpr);
      PFR_CHECK_LAUNCH();
      hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, st, (const float*)workspace, out, g.gx, C, accumulate);
      PFR_CHECK_LAUNCH();
      return PFR_OK;
    }
    const long rpb = (rows + nblk - 1) / nblk;
    const dim3 grid((C + 63) / 64, (unsigned)nblk);
    if (dtype == PFR_BF16)
      hipLaunchKernelGGL(colsum_part_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, workspace, rows, C, rpb);
    else
      hipLaunchKernelGGL(colsum_part_kernel<float>, grid, dim3(256), 0, st, (const float*)x, workspace, rows, C, rpb);
    PFR_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, st, (const float*)workspace, out, (int)nblk, C, accumulate);
    PFR_CHECK_LAUNCH();
    return PFR_OK;
  }
  if (dtype == PFR_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3((C + 63) / 64), dim3(256), 0, st, (const bf16_t*)x, out, (int)rows, C, accumulate);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, dim3((C + 63) / 64), dim3(256), 0, st, (const float*)x, out, (int)rows, C, accumulate);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ---- column sums with a DEFERRED final merge: pfr_colsum_partial leaves the row-block partials in the caller's (per-tensor)
// workspace and returns their count; pfr_colsum_final_batch merges any number of such partial sets in ONE launch (Swin: ~100 bias /
// LayerNorm / position-table gradients per step were one tiny final launch each).
// number of partial rows pfr_colsum_partial writes for this geometry (0: too few rows, use pfr_colsum)
extern "C" int pfr_colsum_parts(int dtype, long rows, int C) {
  const long nblk = colsum_blocks(rows, C);
  if (nblk <= 0) return 0;
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  return C % kp == 0 ? col_geom(C, kp, (size_t)rows, 512).gx : (int)nblk;
}
extern "C" int pfr_colsum_partial(const void* x, int dtype, long rows, int C, float* workspace, hipStream_t st) {
  PFR_CHECK_ARG(x && workspace && colsum_blocks(rows, C) > 0, "pfr_colsum_partial: null pointer or too few rows (%ld) for the partial path", rows);
  const long nblk = colsum_blocks(rows, C);
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  if (C % kp == 0) {
    ColGeom g = col_geom(C, kp, (size_t)rows, 512);
    const size_t shb = (size_t)256 * kp * sizeof(float);
    if (dtype == PFR_BF16)
      hipLaunchKernelGGL(colsum_chunk_kernel<bf16_t>, dim3(g.gx, g.gy), dim3(256), shb, st, (const bf16_t*)x, workspace, (size_t)rows, C, g.cw, g.rl, g.cpr);
    else
      hipLaunchKernelGGL(colsum_chunk_kernel<float>, dim3(g.gx, g.gy), dim3(256), shb, st, (const float*)x, workspace, (size_t)rows, C, g.cw, g.rl, g.cpr);
    PFR_CHECK_LAUNCH();
    return PFR_OK;
  }
  const long rpb = (rows + nblk - 1) / nblk;
  const dim3 grid((C + 63) / 64, (unsigned)nblk);
  if (dtype == PFR_BF16)
    hipLaunchKernelGGL(colsum_part_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, workspace, rows, C, rpb);
  else
    hipLaunchKernelGGL(colsum_part_kernel<float>, grid, dim3(256), 0, st, (const float*)x, workspace, rows, C, rpb);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

struct ColsumDesc {
  const float* part;
  float* out;
  int n, C, accumulate;
  int mt, rows, pad;   // mt > 0: `part` = [n][2][C] tile statistics (mean, M2) of m-tiles of height mt over `rows` rows: sum = Σ rows_t·mean_t
};
__global__ __launch_bounds__(256) void colsum_final_batch_kernel(const ColsumDesc* __restrict__ descs) {
  const ColsumDesc d = descs[blockIdx.y];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  if (blockIdx.x * 16 >= d.C) return;   // (uniform per workgroup)
  __shared__ float l[16][17];
  // eight partial rows per thread are requested together (clamped addresses): the merges of a bucket boundary sit at the end of the
  // side stream's work, and with one load per loop iteration a 3 136-tile statistic took 196 serial memory round trips (87 µs)
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  if (c < d.C) {
    const size_t rs = d.mt > 0 ? (size_t)2 * d.C : (size_t)d.C;
    for (int r0 = rl; r0 < d.n; r0 += 16 * 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = d.part[(size_t)min(r0 + 16 * k, d.n - 1) * rs + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = r0 + 16 * k;
        if (r < d.n) acc[k] = d.mt > 0 ? fmaf(v[k], (float)min(d.mt, d.rows - r * d.mt), acc[k]) : acc[k] + v[k];
      }
    }
  }
  const float a0 = acc[0] + acc[1], a1 = acc[2] + acc[3], a2 = acc[4] + acc[5], a3 = acc[6] + acc[7];
  l[rl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && c < d.C) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += l[r][cl];
    d.out[c] = d.accumulate ? d.out[c] + a : a;
  }
}
extern "C" int pfr_colsum_final_batch(const void* descs, int n, int max_C, hipStream_t st) {
  PFR_CHECK_ARG(descs && n > 0 && n <= 65535 && max_C > 0, "pfr_colsum_final_batch: bad args");
  hipLaunchKernelGGL(colsum_final_batch_kernel, dim3((max_C + 15) / 16, n), dim3(256), 0, st, (const ColsumDesc*)descs);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ------------------------------------------------------------------------------------------------
// optimiser steps over flat fp32 master buffers (+ compute-dtype shadow copy of the parameters)
// four parameters per thread (16-byte loads of p / g / momentum issued together, one 8- or 16-byte shadow store): the scalar version
// made three dependent 4-byte round trips per parameter
template <typename TS>
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom, TS* __restrict__ shadow,
                           size_t n, float lr, float momentum, float wd, float gscale, int first) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(mom) |
                     reinterpret_cast<uintptr_t>(shadow)) & 15) == 0;
  const size_t n4 = vec ? n / 4 : 0;
  const bool use_mom = momentum != 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 w = reinterpret_cast<const f32x4*>(p)[i];
    const f32x4 gr = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 m = {0.f, 0.f, 0.f, 0.f};
    if (use_mom && !first) m = reinterpret_cast<const f32x4*>(mom)[i];
    f32x4 b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = fmaf(wd, w[e], gr[e] * gscale);
      b[e] = use_mom ? (first ? d : fmaf(momentum, m[e], d)) : d;
      w[e] = fmaf(-lr, b[e], w[e]);
    }
    if (use_mom) reinterpret_cast<f32x4*>(mom)[i] = b;
    reinterpret_cast<f32x4*>(p)[i] = w;
    if (shadow) {
      if constexpr (sizeof(TS) == 2) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (bf16_t)w[e];
        *reinterpret_cast<bf16x4*>(shadow + 4 * i) = v;
      } else {
        reinterpret_cast<f32x4*>(shadow)[i] = w;
      }
    }
  }
  for (size_t i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float w = p[i];
    float d = fmaf(wd, w, g[i] * gscale);
    float b = d;
    if (momentum != 0.f) {
      b = first ? d : fmaf(momentum, mom[i], d);
      mom[i] = b;
    }
    w = fmaf(-lr, b, w);
    p[i] = w;
    if (shadow) shadow[i] = from_f32<TS>(w);
  }
}
extern "C" int pfr_sgd_step(float* p, const float* g, float* mom, void* shadow, int shadow_dtype, size_t n, float lr,
                            float momentum, float weight_decay, float grad_scale, int first_step, hipStream_t st) {
  PFR_CHECK_ARG(p && g && (momentum == 0.f || mom), "pfr_sgd_step: null pointer");
  if (n == 0) return PFR_OK;
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (shadow && shadow_dtype == PFR_BF16)
    hipLaunchKernelGGL(sgd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, p, g, mom, (bf16_t*)shadow, n, lr, momentum, weight_decay, grad_scale, first_step);
  else
    hipLaunchKernelGGL(sgd_kernel<float>, dim3(blocks), dim3(256), 0, st, p, g, mom, (float*)shadow, n, lr, momentum, weight_decay, grad_scale, first_step);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

template <typename TS>
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             TS* __restrict__ shadow, size_t n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2, float gscale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float w = p[i];
    const float gr = g[i] * gscale;
    w *= 1.f - lr * wd;
    const float mm = b1 * m[i] + (1.f - b1) * gr;
    const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mm;
    v[i] = vv;
    const float denom = sqrtf(vv) / sqrtf(bc2) + eps;
    w -= (lr / bc1) * mm / denom;
    p[i] = w;
    if (shadow) shadow[i] = from_f32<TS>(w);
  }
}
extern "C" int pfr_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, size_t n,
                              float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                              float grad_scale, hipStream_t st) {
  PFR_CHECK_ARG(p && g && m && v && step >= 1, "pfr_adamw_step: bad args");
  if (n == 0) return PFR_OK;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (shadow && shadow_dtype == PFR_BF16)
    hipLaunchKernelGGL(adamw_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, p, g, m, v, (bf16_t*)shadow, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
  else
    hipLaunchKernelGGL(adamw_kernel<float>, dim3(blocks), dim3(256), 0, st, p, g, m, v, (float*)shadow, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// strided 2-D fp32 copy (extracting the real channels of the channel-padded stem weight gradient)
__global__ void copy2d_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, size_t rows, int cols,
                              float scale, int accumulate) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = rows * cols, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const size_t r = i / cols;
    const int c = (int)(i % cols);
    const float v = src[r * lds_ + c] * scale;
    dst[r * ldd + c] = accumulate ? dst[r * ldd + c] + v : v;
  }
}
extern "C" int pfr_copy2d_f32(const float* src, int ld_src, float* dst, int ld_dst, long rows, int cols, float scale,
                              int accumulate, hipStream_t st) {
  PFR_CHECK_ARG(src && dst && rows >= 0 && cols >= 0, "pfr_copy2d_f32: bad args");
  const size_t n = (size_t)rows * cols;
  if (n == 0) return PFR_OK;
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(copy2d_kernel, dim3(blocks), dim3(256), 0, st, src, ld_src, dst, ld_dst, (size_t)rows, cols, scale, accumulate);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
