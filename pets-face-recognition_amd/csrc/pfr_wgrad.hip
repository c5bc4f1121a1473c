// pfr_wgrad.hip — weight gradient of an NHWC convolution / linear layer on MFMA.
//
// Replaces the autograd weight-gradient of every `nn.Conv2d` / `nn.Linear` / `F.linear` on the reference's hot
// path (backward of torchvision resnet50, configs/dog_fe/fe_dogs_config.py:102-103, and of
// losses/large_margin.py:71):
//
//   dw[co][r][s][c] = sum_m dy[m][co] * act(x)[n, oh*stride - pad + r, ow*stride - pad + s, c]
//
// GEMM view: rows = co (MFMA A), cols = kk = (r,s,c) (MFMA B), reduction = m.  BOTH operands are m-major in HBM
// (channels contiguous), so tiles are staged into LDS exactly as they lie in memory ([m][channel], coalesced
// 16-byte chunks, no register shuffling) and the per-lane "8 consecutive k" MFMA operand is produced by the
// gfx950 LDS transpose read `ds_read_b64_tr_b16` (bf16) or plain `ds_read_b32` (f32: one k per lane).
// The reduction over m is split across workgroups; each split writes an fp32 partial slab
// [split][Cout][KK] (deterministic), summed by pfr_wgrad_reduce.
#include "pfr_mma.h"
#include <stdlib.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradParams {
  const void* x;
  const void* dy;
  float* dw;
  int N, H, W, C, R, S, OH, OW, stride, pad;
  int Cout, lddy, M, KK;
  int splits, mchunk;
  int v2;      // use the LDS-DMA kernel (no fused prologue)
  int simple;  // 1x1, stride 1, pad 0: the gathered row of x is row m itself (no (n,oh,ow) decomposition)
  const float* pro_scale;
  const float* pro_shift;
  int pro_relu;
  FastDiv div_ohow, div_ow;
  int tilesP, tilesQ;
  int adv_q1, adv_q2, adv_r2;   // wgrad3: 32 rows = adv_q1 images + adv_q2 output rows + adv_r2 output columns
};

#ifndef PFR_WGRAD_MUL
#define PFR_WGRAD_MUL 2
#endif
template <typename T> struct WG {
  static constexpr int BMR = PFR_WGRAD_MUL * 64 / (int)sizeof(T);  // reduction rows per k-step (one barrier each)
};

template <typename T, int BP, int BQ, bool PRO>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
  constexpr int KP = DT<T>::KPACK;
  constexpr int BMR = WG<T>::BMR;
  constexpr int TP = BP / 64, TQ = BQ / 64;
  constexpr int RSP = BP * (int)sizeof(T) + 64;  // row strides: ≡ 16 dwords (mod 64) → tr-read rows on disjoint banks
  constexpr int RSQ = BQ * (int)sizeof(T) + 64;
  constexpr int TILEP = BMR * RSP, TILEQ = BMR * RSQ;
  constexpr int STAGE = TILEP + TILEQ;
  constexpr int CPRP = BP / KP, CPRQ = BQ / KP;   // 16-byte chunks per tile row
  constexpr int NCHP = BMR * CPRP / 256, NCHQ = BMR * CPRQ / 256;  // chunks per thread
  static_assert(NCHP >= 1 && NCHQ >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave >> 1, wq = wave & 1;

  const uint32_t t = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tilesP * p.tilesQ;
  const int split = t / ntile, tile = t % ntile;
  const int tq = tile % p.tilesQ, tpp = tile / p.tilesQ;
  const int co0 = tpp * BP, kk0 = tq * BQ;
  const int mbeg = split * p.mchunk;
  const int mend = min(p.M, mbeg + p.mchunk);

  // P (dy) chunk geometry of this thread
  const int pch = tid % CPRP, prow0 = tid / CPRP;  // rows prow0 + i*(256/CPRP)
  const int pco = co0 + pch * KP;
  // Q (x) chunk geometry
  const int qch = tid % CPRQ, qrow0 = tid / CPRQ;
  const int kk = kk0 + qch * KP;
  const bool kkok = kk < p.KK;
  const int tap = kkok ? kk / p.C : 0;
  const int ci = kk - tap * p.C;
  const int tr = tap / p.S, ts = tap - tr * p.S;

  const char* xb = reinterpret_cast<const char*>(p.x);
  const char* dyb = reinterpret_cast<const char*>(p.dy);

  float psc[PRO ? KP : 1], psh[PRO ? KP : 1];
  if constexpr (PRO) {
    if (kkok) {
#pragma unroll
      for (int e = 0; e < KP; ++e) { psc[e] = p.pro_scale[ci + e]; psh[e] = p.pro_shift[ci + e]; }
    }
  }

  u32x4 preg[NCHP], qreg[NCHQ];
  auto gload = [&](int mb) {
#pragma unroll
    for (int i = 0; i < NCHP; ++i) {
      const int m = mb + prow0 + i * (256 / CPRP);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (m < mend && pco < p.Cout) v = ld16(dyb + ((size_t)m * p.lddy + pco) * sizeof(T));
      preg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NCHQ; ++i) {
      const int m = mb + qrow0 + i * (256 / CPRQ);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (m < mend && kkok && p.simple) {
        v = ld16(xb + ((size_t)m * p.C + ci) * sizeof(T));
        if constexpr (PRO) {
          float f[KP];
          Chunk<T>::unpack(v, f);
#pragma unroll
          for (int e = 0; e < KP; ++e) {
            float z = fmaf(f[e], psc[e], psh[e]);
            f[e] = p.pro_relu ? fmaxf(z, 0.f) : z;
          }
          v = Chunk<T>::pack(f);
        }
      } else if (m < mend && kkok) {
        const uint32_t n_img = fdiv((uint32_t)m, p.div_ohow);
        const uint32_t rem = m - n_img * (uint32_t)(p.OH * p.OW);
        const uint32_t oh = fdiv(rem, p.div_ow);
        const uint32_t ow = rem - oh * p.OW;
        const int ih = (int)oh * p.stride - p.pad + tr, iw = (int)ow * p.stride - p.pad + ts;
        if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
          v = ld16(xb + ((size_t)((n_img * p.H + ih) * p.W + iw) * p.C + ci) * sizeof(T));
          if constexpr (PRO) {
            float f[KP];
            Chunk<T>::unpack(v, f);
#pragma unroll
            for (int e = 0; e < KP; ++e) {
              float z = fmaf(f[e], psc[e], psh[e]);
              f[e] = p.pro_relu ? fmaxf(z, 0.f) : z;
            }
            v = Chunk<T>::pack(f);
          }
        }
      }
      qreg[i] = v;
    }
  };
  auto lstore = [&](int buf) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NCHP; ++i) st16(base + (prow0 + i * (256 / CPRP)) * RSP + pch * 16, preg[i]);
#pragma unroll
    for (int i = 0; i < NCHQ; ++i) st16(base + TILEP + (qrow0 + i * (256 / CPRQ)) * RSQ + qch * 16, qreg[i]);
  };

  f32x16 acc[TP][TQ];
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (mend - mbeg + BMR - 1) / BMR;
  if (nk > 0) {
    gload(mbeg);
    lstore(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(mbeg + (kt + 1) * BMR);
    const char* bp = smem + buf * STAGE + (wp * (BP / 2)) * (int)sizeof(T);
    const char* bq = smem + buf * STAGE + TILEP + (wq * (BQ / 2)) * (int)sizeof(T);
    if constexpr (sizeof(T) == 2) {
      // lane l: g = l>>4 ; block column (channel) = (g&1)*16 + (l&15) = l&31 ; k-half = g>>1 = l>>5
      const int g = lane >> 4, s = lane & 15;
      const int roff = (g >> 1) * 8 + (s >> 2);          // row within the 16-row k-group (plus q*4)
      const int coff = ((g & 1) * 16 + (s & 3) * 4) * 2;  // byte offset of this lane's 8-byte piece
#pragma unroll
      for (int kg = 0; kg < BMR / 16; ++kg) {
        bf16x8 fp[TP], fq[TQ];
#pragma unroll
        for (int i = 0; i < TP; ++i) {
          const char* a = bp + (kg * 16 + roff) * RSP + i * 64 + coff;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a + 4 * RSP));
          u32x4 u;
          u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          u[0] = l2[0]; u[1] = l2[1]; u[2] = h2[0]; u[3] = h2[1];
          fp[i] = __builtin_bit_cast(bf16x8, u);
        }
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
          const char* a = bq + (kg * 16 + roff) * RSQ + j * 64 + coff;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a + 4 * RSQ));
          u32x4 u;
          u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          u[0] = l2[0]; u[1] = l2[1]; u[2] = h2[0]; u[3] = h2[1];
          fq[j] = __builtin_bit_cast(bf16x8, u);
        }
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
          for (int j = 0; j < TQ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[i], fq[j], acc[i][j], 0, 0, 0);
      }
    } else {
      const int row = lane & 31, kh = lane >> 5;
#pragma unroll
      for (int e = 0; e < BMR / 2; ++e) {
        float fp[TP], fq[TQ];
#pragma unroll
        for (int i = 0; i < TP; ++i) fp[i] = *reinterpret_cast<const float*>(bp + (2 * e + kh) * RSP + (i * 32 + row) * 4);
#pragma unroll
        for (int j = 0; j < TQ; ++j) fq[j] = *reinterpret_cast<const float*>(bq + (2 * e + kh) * RSQ + (j * 32 + row) * 4);
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
          for (int j = 0; j < TQ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fp[i], fq[j], acc[i][j], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  float* out = p.dw + (size_t)split * p.Cout * p.KK;
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      const int col = kk0 + wq * (BQ / 2) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wp * (BP / 2) + i * 32 + acc_row(r, lane);
        if (co < p.Cout && col < p.KK) out[(size_t)co * p.KK + col] = acc[i][j][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// v2 (no fused prologue): both operand tiles are staged by LDS-DMA (buffer_load … lds) exactly as they lie in HBM
// ([m][channel]); rows are unpadded (wave-linear DMA destination) and the 32-byte granules of a row are XOR-swizzled with
// row bits so that the two 16-lane groups of a `ds_read_b64_tr_b16` (4 rows x 2 granules = 8 pieces of 32 B) cover all
// 64 banks exactly once.  64 reduction rows (bf16) per barrier = 16 MFMAs per wave; out-of-range rows / padding taps /
// channel tails are zero-filled by the buffer descriptor's bounds check.
template <int GPR> __device__ __forceinline__ int gran_swz(int row) {   // GPR = 32-byte granules per LDS row
  return GPR >= 8 ? ((row & 3) << 1) : (((row >> 1) & 1) << 1);
}

template <typename T, int BP, int BQ>
__global__ __launch_bounds__(256) void wgrad2_kernel(WgradParams p) {
  constexpr int KP = DT<T>::KPACK;
  constexpr int BMR = 128 / (int)sizeof(T);              // reduction rows per k-step: 64 bf16 / 32 f32
  constexpr int TP = BP / 64, TQ = BQ / 64;
  constexpr int RSP = BP * (int)sizeof(T), RSQ = BQ * (int)sizeof(T);   // unpadded row bytes
  constexpr int GPRP = RSP / 32, GPRQ = RSQ / 32;
  constexpr int TILEP = BMR * RSP, TILEQ = BMR * RSQ, STAGE = TILEP + TILEQ;
  constexpr int RPIP = 1024 / RSP, RPIQ = 1024 / RSQ;    // rows covered by one wave-wide DMA instruction
  constexpr int NDP = BMR / (4 * RPIP), NDQ = BMR / (4 * RPIQ);   // DMA instructions per thread per k-step
  constexpr bool SWZ = sizeof(T) == 2;                    // f32 fragments are read with ds_read_b32: no swizzle needed
  static_assert(NDP >= 1 && NDQ >= 1, "tile too wide");
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 1, wq = wave & 1;

  const uint32_t t = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tilesP * p.tilesQ;
  const int split = t / ntile, tile = t % ntile;
  const int tq = tile % p.tilesQ, tpp = tile / p.tilesQ;
  const int co0 = tpp * BP, kk0 = tq * BQ;
  const int mbeg = split * p.mchunk;
  const int mend = min(p.M, mbeg + p.mchunk);

  // DMA geometry: lane l of a wave instruction lands at tile row (l / CPR), physical 16-byte chunk (l % CPR)
  constexpr int CPRP = RSP / 16, CPRQ = RSQ / 16;
  const int prsub = lane / CPRP, qrsub = lane / CPRQ;
  int pchunk = lane % CPRP, qchunk = lane % CPRQ;
  if constexpr (SWZ) {   // logical chunk fetched into this physical slot (row & 7 depends on the lane only: passes add multiples of 8... see RPI)
    pchunk = ((((pchunk >> 1) ^ gran_swz<GPRP>((wave * RPIP + prsub))) << 1) | (pchunk & 1));
    qchunk = ((((qchunk >> 1) ^ gran_swz<GPRQ>((wave * RPIQ + qrsub))) << 1) | (qchunk & 1));
  }
  const int pco = co0 + pchunk * KP;
  const int kk = kk0 + qchunk * KP;
  const bool kkok = kk < p.KK;
  const int tap = kkok ? kk / p.C : 0;
  const int ci = kk - tap * p.C;
  const int tr = tap / p.S, ts = tap - tr * p.S;

  const uint32_t OOB = 0xFFFFFF00u;
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.C * sizeof(T)), 0x00020000);
  __amdgpu_buffer_rsrc_t drsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, (int)((size_t)p.M * p.lddy * sizeof(T)), 0x00020000);

  auto gload = [&](int buf, int mb) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < NDP; ++j) {
      const int m = mb + (j * 4 + wave) * RPIP + prsub;
      const uint32_t off = (m < mend && pco < p.Cout) ? (uint32_t)(((size_t)m * p.lddy + pco) * sizeof(T)) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(drsrc, (__attribute__((address_space(3))) void*)(base + (j * 4 + wave) * RPIP * RSP),
                                               16, (int)off, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NDQ; ++j) {
      const int m = mb + (j * 4 + wave) * RPIQ + qrsub;
      uint32_t off = OOB;
      if (m < mend && kkok) {
        if (p.simple) {
          off = (uint32_t)(((size_t)m * p.C + ci) * sizeof(T));
        } else {
          const uint32_t n_img = fdiv((uint32_t)m, p.div_ohow);
          const uint32_t rem = m - n_img * (uint32_t)(p.OH * p.OW);
          const uint32_t oh = fdiv(rem, p.div_ow);
          const uint32_t ow = rem - oh * p.OW;
          const int ih = (int)oh * p.stride - p.pad + tr, iw = (int)ow * p.stride - p.pad + ts;
          if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
            off = (uint32_t)(((size_t)((n_img * p.H + ih) * p.W + iw) * p.C + ci) * sizeof(T));
        }
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(base + TILEP + (j * 4 + wave) * RPIQ * RSQ),
                                               16, (int)off, 0, 0, 0);
    }
  };

  f32x16 acc[TP][TQ];
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (mend - mbeg + BMR - 1) / BMR;
  if (nk > 0) gload(0, mbeg);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const char* bp = smem + buf * STAGE;
    const char* bq = smem + buf * STAGE + TILEP;
    bool issued = false;
    auto mid = [&]() {
      if (!issued && kt + 1 < nk) gload(buf ^ 1, mbeg + (kt + 1) * BMR);
      issued = true;
    };
    if constexpr (sizeof(T) == 2) {
      // lane l: g = l>>4 ; fragment column (channel) = (g&1)*16 + (l&15) = l&31 ; k-half = g>>1 = l>>5
      const int g = lane >> 4, s4 = lane & 15;
      // fragments of k-group kg+1 are fetched (ds_read_b64_tr_b16) while the MFMAs of k-group kg run: register double buffer
      bf16x8 fp[2][TP], fq[2][TQ];
      auto rd = [&](int kg, int bsel) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int r = kg * 16 + (g >> 1) * 8 + q * 4 + (s4 >> 2);
#pragma unroll
          for (int i = 0; i < TP; ++i) {
            const int gran = (wp * (BP / 2) + i * 32) / 16 + (g & 1);
            const char* a = bp + r * RSP + ((gran ^ gran_swz<GPRP>(r)) << 5) + (s4 & 3) * 8;
            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a));
            u32x2 v2 = __builtin_bit_cast(u32x2, v);
            u32x4 u = __builtin_bit_cast(u32x4, fp[bsel][i]);
            u[2 * q] = v2[0]; u[2 * q + 1] = v2[1];
            fp[bsel][i] = __builtin_bit_cast(bf16x8, u);
          }
#pragma unroll
          for (int j = 0; j < TQ; ++j) {
            const int gran = (wq * (BQ / 2) + j * 32) / 16 + (g & 1);
            const char* a = bq + r * RSQ + ((gran ^ gran_swz<GPRQ>(r)) << 5) + (s4 & 3) * 8;
            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a));
            u32x2 v2 = __builtin_bit_cast(u32x2, v);
            u32x4 u = __builtin_bit_cast(u32x4, fq[bsel][j]);
            u[2 * q] = v2[0]; u[2 * q + 1] = v2[1];
            fq[bsel][j] = __builtin_bit_cast(bf16x8, u);
          }
        }
      };
      rd(0, 0);
#pragma unroll
      for (int kg = 0; kg < BMR / 16; ++kg) {
        const int bsel = kg & 1;
        if (kg + 1 < BMR / 16) rd(kg + 1, bsel ^ 1);
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
          for (int j = 0; j < TQ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[bsel][i], fq[bsel][j], acc[i][j], 0, 0, 0);
        if (kg == 0) mid();
      }
    } else {
      const int row = lane & 31, kh = lane >> 5;
#pragma unroll
      for (int e = 0; e < BMR / 2; ++e) {
        float fp[TP], fq[TQ];
#pragma unroll
        for (int i = 0; i < TP; ++i) fp[i] = *reinterpret_cast<const float*>(bp + (2 * e + kh) * RSP + (wp * (BP / 2) + i * 32 + row) * 4);
#pragma unroll
        for (int j = 0; j < TQ; ++j) fq[j] = *reinterpret_cast<const float*>(bq + (2 * e + kh) * RSQ + (wq * (BQ / 2) + j * 32 + row) * 4);
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
          for (int j = 0; j < TQ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fp[i], fq[j], acc[i][j], 0, 0, 0);
        if (e == 3) mid();
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  float* out = p.dw + (size_t)split * p.Cout * p.KK;
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      const int col = kk0 + wq * (BQ / 2) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wp * (BP / 2) + i * 32 + acc_row(r, lane);
        if (co < p.Cout && col < p.KK) out[(size_t)co * p.KK + col] = acc[i][j][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// v3 (bf16, no fused prologue): the v2 data path with the per-stage VECTOR-ALU work removed.  PMC counters of v2 (3x3, 256 ch,
// 14x14, bs 256): 316 VALU instructions per wave per 64-row stage against 16 MFMAs — address arithmetic of the transpose reads
// (recomputed per read), register moves assembling the fragments, and a divide-based (n, oh, ow) decode per gathered row; at 4
// cycles per wave64 VALU instruction that is 2.5x the MFMA time of the stage, so the matrix pipe idles behind the vector ALU.
//   * transpose reads: one base VGPR per fragment (lane part of the swizzled address, computed once) + the stage / k-group /
//     half position as the instruction's immediate offset (the stage loop is unrolled over the 4 ring slots);
//   * dy (and x of 1x1 stride-1 layers): constant per-lane offsets; the stage advances the buffer descriptor's base and shrinks
//     num_records on the scalar ALU, so row tails are zero-filled by the bounds check with no vector compare;
//   * gathered x (3x3 / strided): (n, oh, ow) is decoded once and then ADVANCED by 32 rows per stage with carries;
//   * ring of 4 stages of 32 rows, counted vmcnt; reads are inline asm (a compiler-visible LDS read drains the DMAs in flight).
#ifndef PFR_WGRAD_NST
#define PFR_WGRAD_NST 4   // LDS ring depth of wgrad3_kernel (3: 48 KB per 128x128 workgroup, three workgroups per CU)
#endif
__device__ __forceinline__ u32x2 lds_read_tr16(uint32_t addr, int imm) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(imm) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void wg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NW waves in a WGP x WGQ grid over the (cout, kk) tile: 4 waves 2x2 (64 KB of LDS, two workgroups per CU) or 8 waves 4x2 on a
// 256-wide tile (128 KB, one workgroup per CU: half the operand bytes staged per MFMA, and the 64x128 wave tile needs 12 transpose
// reads per 8 MFMAs instead of 8 per 4 — at full MFMA rate the 2x2 form saturates the 128 B/clk LDS port)
template <int BP, int BQ, int NW = 4, int WGQ = 2>
__global__ __launch_bounds__(NW * 64) void wgrad3_kernel(WgradParams p) {
  using T = bf16_t;
  constexpr int KP = 8, BMR = 32, NST = PFR_WGRAD_NST;
  constexpr int WGP = NW / WGQ;
  constexpr int TP = BP / (WGP * 32), TQ = BQ / (WGQ * 32);
  constexpr int RSP = BP * 2, RSQ = BQ * 2;
  constexpr int GPRP = RSP / 32, GPRQ = RSQ / 32;
  constexpr int TILEP = BMR * RSP, TILEQ = BMR * RSQ, STAGE = TILEP + TILEQ;
  constexpr int RPIP = 1024 / RSP, RPIQ = 1024 / RSQ;
  constexpr int NDP = BMR / (NW * RPIP), NDQ = BMR / (NW * RPIQ);
  constexpr int IPS = NDP + NDQ, D = NST - 1;
  static_assert(NDP >= 1 && NDQ >= 1 && NST * STAGE <= 160 * 1024 && TP >= 1 && TQ >= 1, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WGQ, wq = wave % WGQ;
  const uint32_t t = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tilesP * p.tilesQ;
  const int split = t / ntile, tile = t % ntile;
  const int tq = tile % p.tilesQ, tpp = tile / p.tilesQ;
  const int co0 = tpp * BP, kk0 = tq * BQ;
  const int mbeg = split * p.mchunk;
  const int mend = min(p.M, mbeg + p.mchunk);
  const int nrows = mend - mbeg;
  const int nk = (nrows + BMR - 1) / BMR;

  // ---- DMA lane geometry (as v2): lane l of a wave instruction lands at tile row l / CPR, physical 16-byte chunk l % CPR
  constexpr int CPRP = RSP / 16, CPRQ = RSQ / 16;
  const int prsub = lane / CPRP, qrsub = lane / CPRQ;
  int pchunk = lane % CPRP, qchunk = lane % CPRQ;
  pchunk = ((((pchunk >> 1) ^ gran_swz<GPRP>(wave * RPIP + prsub)) << 1) | (pchunk & 1));
  qchunk = ((((qchunk >> 1) ^ gran_swz<GPRQ>(wave * RPIQ + qrsub)) << 1) | (qchunk & 1));
  const int pco = co0 + pchunk * KP;
  const int kk = kk0 + qchunk * KP;
  const bool kkok = kk < p.KK;
  const int tap = kkok ? kk / p.C : 0;
  const int ci = kk - tap * p.C;
  const int tr = tap / p.S, ts = tap - tr * p.S;
  const uint32_t OOB = 0xFFFFFF00u;

  uint32_t voffP[NDP], voffQ[NDQ];
#pragma unroll
  for (int j = 0; j < NDP; ++j)
    voffP[j] = pco < p.Cout ? (uint32_t)((((j * NW + wave) * RPIP + prsub) * p.lddy + pco) * 2) : OOB;
  // gathered-x walker state per DMA pass (general form only)
  int w_ow[NDQ], w_oh[NDQ], w_rem[NDQ];
  uint32_t w_nb[NDQ];
  const uint32_t hwc2 = (uint32_t)p.H * p.W * p.C * 2, wc2 = (uint32_t)p.W * p.C * 2, c2 = (uint32_t)p.C * 2;
#pragma unroll
  for (int j = 0; j < NDQ; ++j) {
    const int rowj = (j * NW + wave) * RPIQ + qrsub;
    if (p.simple) {
      voffQ[j] = kkok ? (uint32_t)((rowj * p.C + ci) * 2) : OOB;
    } else {
      const uint32_t m = (uint32_t)(mbeg + rowj);
      const uint32_t n_img = fdiv(m, p.div_ohow);
      const uint32_t rem = m - n_img * (uint32_t)(p.OH * p.OW);
      const uint32_t oh = fdiv(rem, p.div_ow);
      w_ow[j] = (int)(rem - oh * p.OW);
      w_oh[j] = (int)oh;
      w_nb[j] = n_img * hwc2 + (uint32_t)ci * 2;
      w_rem[j] = kkok ? nrows - rowj : 0;
    }
  }
  const int dih = tr - p.pad, diw = ts - p.pad;

  const char* dyb = reinterpret_cast<const char*>(p.dy);
  const char* xb = reinterpret_cast<const char*>(p.x);
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.C * sizeof(T)), 0x00020000);

  auto issue = [&](int slot, int s) __attribute__((always_inline)) {
    char* base = smem + slot * STAGE;
    const int mb = mbeg + s * BMR;
    const int left = mend - mb;   // > 0
    __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dyb + (size_t)mb * p.lddy * 2), 0,
                                                                  left * p.lddy * 2, 0x00020000);
#pragma unroll
    for (int j = 0; j < NDP; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dr, (__attribute__((address_space(3))) void*)(base + (j * NW + wave) * RPIP * RSP), 16,
                                               (int)voffP[j], 0, 0, 0);
    if (p.simple) {
      __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb + (size_t)mb * p.C * 2), 0,
                                                                    left * p.C * 2, 0x00020000);
#pragma unroll
      for (int j = 0; j < NDQ; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(qr, (__attribute__((address_space(3))) void*)(base + TILEP + (j * NW + wave) * RPIQ * RSQ),
                                                 16, (int)voffQ[j], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < NDQ; ++j) {
        const int ih = w_oh[j] * p.stride + dih, iw = w_ow[j] * p.stride + diw;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && w_rem[j] > 0;
        const uint32_t off = ok ? w_nb[j] + (uint32_t)ih * wc2 + (uint32_t)iw * c2 : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(base + TILEP + (j * NW + wave) * RPIQ * RSQ),
                                                 16, (int)off, 0, 0, 0);
        // advance this lane's row by BMR output pixels: columns, then rows, then images (each wraps at most once)
        int ow = w_ow[j] + p.adv_r2;
        const int c1 = ow >= p.OW ? 1 : 0;
        ow -= c1 ? p.OW : 0;
        int oh = w_oh[j] + p.adv_q2 + c1;
        const int cc = oh >= p.OH ? 1 : 0;
        oh -= cc ? p.OH : 0;
        w_ow[j] = ow;
        w_oh[j] = oh;
        w_nb[j] += (uint32_t)(p.adv_q1 + cc) * hwc2;
        w_rem[j] -= BMR;
      }
    }
  };

  f32x16 acc[TP][TQ];
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- transpose-read lane bases: row (g>>1)*8 + (s4>>2) of a k-group, swizzled 32-byte granule, 8-byte piece s4&3
  const int g = lane >> 4, s4 = lane & 15;
  const int r0 = (g >> 1) * 8 + (s4 >> 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t laneP[TP], laneQ[TQ];
#pragma unroll
  for (int i = 0; i < TP; ++i) {
    const int gran = (wp * (BP / WGP) + i * 32) / 16 + (g & 1);
    laneP[i] = lds0 + r0 * RSP + ((gran ^ gran_swz<GPRP>(r0)) << 5) + (s4 & 3) * 8;
  }
#pragma unroll
  for (int j = 0; j < TQ; ++j) {
    const int gran = (wq * (BQ / WGQ) + j * 32) / 16 + (g & 1);
    laneQ[j] = lds0 + TILEP + r0 * RSQ + ((gran ^ gran_swz<GPRQ>(r0)) << 5) + (s4 & 3) * 8;
  }

#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nk) issue(s, s);

  constexpr int NR = 2 * (TP + TQ);   // transpose reads per k-group
  auto stage = [&](auto slot_c, int kt) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_c)::value;
    const int younger = min(D - 1, nk - 1 - kt);
    if (younger >= 2) wg_wait_vm<2 * IPS>();
    else if (younger == 1) wg_wait_vm<IPS>();
    else wg_wait_vm<0>();
    __builtin_amdgcn_s_barrier();   // every wave's share of stage kt is in LDS; everyone is done reading slot (SLOT + D) % NST
    u32x2 hp[2][TP][2], hq[2][TQ][2];
    auto rd = [&](auto kg_c, int b) __attribute__((always_inline)) {
      constexpr int KG = decltype(kg_c)::value;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int i = 0; i < TP; ++i) {
          constexpr int off = SLOT * STAGE + (KG * 16 + 4) * RSP;   // largest offset of the pair; the ds immediate is 16 bits
          constexpr int hi = off >= 65536 ? ((SLOT * STAGE) & ~32767) : 0;
          hp[b][i][q] = lds_read_tr16(laneP[i] + hi, SLOT * STAGE - hi + (KG * 16 + q * 4) * RSP);
        }
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
          constexpr int off = SLOT * STAGE + (KG * 16 + 4) * RSQ + TILEP;
          constexpr int hi = off >= 65536 ? ((SLOT * STAGE) & ~32767) : 0;
          hq[b][j][q] = lds_read_tr16(laneQ[j] + hi, SLOT * STAGE - hi + (KG * 16 + q * 4) * RSQ);
        }
      }
    };
    auto mma = [&](int b) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(hp[b][i][q]));
#pragma unroll
      for (int j = 0; j < TQ; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(hq[b][j][q]));
#pragma unroll
      for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
          const u32x4 ua = {hp[b][i][0][0], hp[b][i][0][1], hp[b][i][1][0], hp[b][i][1][1]};
          const u32x4 ub = {hq[b][j][0][0], hq[b][j][0][1], hq[b][j][1][0], hq[b][j][1][1]};
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[i][j],
                                                               0, 0, 0);
        }
    };
    rd(std::integral_constant<int, 0>{}, 0);
    rd(std::integral_constant<int, 1>{}, 1);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR) : "memory");
    mma(0);
    if (kt + D < nk) issue((SLOT + D) % NST, kt + D);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mma(1);
  };
  for (int kt0 = 0; kt0 < nk; kt0 += NST) {
    stage(std::integral_constant<int, 0>{}, kt0);
    if (kt0 + 1 < nk) stage(std::integral_constant<int, 1>{}, kt0 + 1);
    if (kt0 + 2 < nk) stage(std::integral_constant<int, 2>{}, kt0 + 2);
    if constexpr (NST > 3) { if (kt0 + 3 < nk) stage(std::integral_constant<int, 3>{}, kt0 + 3); }
  }

  float* out = p.dw + (size_t)split * p.Cout * p.KK;
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      const int col = kk0 + wq * (BQ / WGQ) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wp * (BP / WGP) + i * 32 + acc_row(r, lane);
        if (co < p.Cout && col < p.KK) out[(size_t)co * p.KK + col] = acc[i][j][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Streaming weight gradient of the HBM-bound 1x1 / stride-1 layers (bf16; round 3).  The tile kernels above hand every 32-row stage
// from the DMA to the MFMAs through a workgroup barrier and reach 0.45-0.55 of the HBM bound on the 56x56 layers (two operands
// streamed once, a [Cout][C] output of 16-64 k values).  Here, as in pfr_sconv.hip, every wave owns a PRIVATE ring of LDS-DMA
// slots and there is NO barrier in the loop: 256 persistent workgroups of 8 waves; the workgroup's output tile [PW][QW] is split
// over WP x WQ waves (each GP x GQ column blocks of 64 channels = 2GP x 2GQ MFMA tiles), and WM = 8 / (WP*WQ) wave groups take
// different 16-row reduction steps; the chip walks the tensors as ONE moving window (step b of the tensor goes to workgroup
// (b / WM) % nsplit, wave group b % WM).  A step's operand block is [16 rows][64 channels] = 2 DMA instructions (8 rows x 128 B,
// whole lines); 32-byte granules are XOR-swizzled by row on the SOURCE side so that the transposing `ds_read_b64_tr_b16` reads are
// conflict free.  Operand blocks shared by waves of a workgroup are fetched once per wave (from L2 the second time): the price of
// the missing barrier is 1.2-2x the LDS-DMA volume, which only shapes with a narrow operand can afford (host heuristic).
// Wave groups are summed through LDS at the end (fixed order), one fp32 slab per workgroup, then wgrad_reduce_kernel.
struct SwgradParams {
  const void* x;     // [M][Q]
  const void* dy;    // [M][lddy], Cout = P columns used
  float* slabs;      // [nsplit][P][Q]
  int M, P, Q, lddy;
  int nsplit, tilesP, tilesQ;
  int WP, WQ, WM;    // wave grid (WP * WQ * WM == 8)
  int nit;           // reduction steps per wave
};
template <int GP, int GQ, int NS>
__global__ __launch_bounds__(512, 1) void swgrad_kernel(SwgradParams p) {
  constexpr int NB = GP + GQ;              // operand blocks per step
  constexpr int SLOT = NB * 2048, IPS = NB * 2;
  constexpr int TP = 2 * GP, TQ = 2 * GQ;
  static_assert(8 * NS * SLOT <= 160 * 1024 && TP * TQ <= 8, "tile geometry");
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / (p.WP * p.WQ), wr = wave % (p.WP * p.WQ), wq = wr / p.WP, wp = wr % p.WP;
  // workgroup -> (split, tile): the tiles of one split run on one XCD (they share an operand; block b is observed on XCD b % 8)
  const int b = blockIdx.x, xcd = b & 7, j8 = b >> 3, ntile = p.tilesP * p.tilesQ;
  const int tile = j8 % ntile, split = (j8 / ntile) * 8 + xcd;
  if (split >= p.nsplit) return;
  const int tq = tile % p.tilesQ, tpp = tile / p.tilesQ;
  const int co0 = (tpp * p.WP + wp) * GP * 64, c0 = (tq * p.WQ + wq) * GQ * 64;

  // ---- DMA lane geometry: instruction h of a block covers rows 8h .. 8h+7 x 128 B; lane -> (row, physical 16-byte chunk)
  const int drow = lane >> 3, pc = lane & 7;
  uint32_t voffA[2], voffB[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * h + drow;
    const int lc = ((((pc >> 1) ^ (((row >> 1) & 1) << 1)) << 1) | (pc & 1));   // logical chunk landing in physical chunk pc
    voffA[h] = (uint32_t)((row * p.lddy + co0 + lc * 8) * 2);
    voffB[h] = (uint32_t)((row * p.Q + c0 + lc * 8) * 2);
  }
  const char* dyb = reinterpret_cast<const char*>(p.dy);
  const char* xb = reinterpret_cast<const char*>(p.x);
  char* const ring = smem + wave * (NS * SLOT);
  const long nb = ((long)p.M + 15) >> 4;
  auto issue = [&](int slot, long it) __attribute__((always_inline)) {
    const long blk = (it * p.nsplit + split) * p.WM + wm;
    const long m0 = blk << 4;
    const long left = blk < nb ? (long)p.M - m0 : 0;                 // rows past M (and steps past the end) read zeros
    const long rows = left > 16 ? 16 : left;
    const long mb = blk < nb ? m0 : 0;
    __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dyb + mb * p.lddy * 2), 0, (int)(rows * p.lddy * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb + mb * p.Q * 2), 0, (int)(rows * p.Q * 2), 0x00020000);
    char* base = ring + slot * SLOT;
#pragma unroll
    for (int i = 0; i < GP; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, (__attribute__((address_space(3))) void*)(base + i * 2048 + h * 1024), 16,
                                                 (int)voffA[h], i * 128, 0, 0);
#pragma unroll
    for (int j = 0; j < GQ; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(br, (__attribute__((address_space(3))) void*)(base + (GP + j) * 2048 + h * 1024), 16,
                                                 (int)voffB[h], j * 128, 0, 0);
  };

  f32x16 acc[TP][TQ];
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- transpose-read lane bases (one per 32-channel half of a block): row (g>>1)*8 + (s4>>2), swizzled 32-byte granule, piece s4&3
  const int g = lane >> 4, s4 = lane & 15;
  const int r0 = (g >> 1) * 8 + (s4 >> 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)(wave * (NS * SLOT));
  uint32_t laneT[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int gran = 2 * t + (g & 1);
    laneT[t] = lds0 + r0 * 128 + ((gran ^ (((r0 >> 1) & 1) << 1)) << 5) + (s4 & 3) * 8;
  }

#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s, s);

  auto step = [&](auto slot_c, long it) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value;
    issue((S + NS - 1) % NS, it + NS - 1);
    wg_wait_vm<(NS - 1) * IPS>();
    u32x2 hp[TP][2], hq[TQ][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int i = 0; i < TP; ++i) hp[i][q] = lds_read_tr16(laneT[i & 1], S * SLOT + (i >> 1) * 2048 + q * 512);
#pragma unroll
      for (int j = 0; j < TQ; ++j) hq[j][q] = lds_read_tr16(laneT[j & 1], S * SLOT + (GP + (j >> 1)) * 2048 + q * 512);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
      for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(hp[i][q]));
#pragma unroll
    for (int j = 0; j < TQ; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(hq[j][q]));
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
      for (int j = 0; j < TQ; ++j) {
        const u32x4 ua = {hp[i][0][0], hp[i][0][1], hp[i][1][0], hp[i][1][1]};
        const u32x4 ub = {hq[j][0][0], hq[j][0][1], hq[j][1][0], hq[j][1][1]};
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[i][j], 0, 0, 0);
      }
  };
  for (long it = 0; it < p.nit; it += NS) {
    step(std::integral_constant<int, 0>{}, it);
    if (it + 1 < p.nit) step(std::integral_constant<int, 1>{}, it + 1);
    if constexpr (NS > 2) { if (it + 2 < p.nit) step(std::integral_constant<int, 2>{}, it + 2); }
    if constexpr (NS > 3) { if (it + 3 < p.nit) step(std::integral_constant<int, 3>{}, it + 3); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // (MFMA -> VALU read-after-write distance across the loop's back edge is software-managed: see pfr_sconv.hip)
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j) asm volatile("s_nop 15" : "+v"(acc[i][j]));
  __syncthreads();

  // ---- sum the WM wave groups (tree over LDS, fixed order), then one slab per workgroup
  float* red = reinterpret_cast<float*>(smem);
  const int nwt = p.WP * p.WQ;                 // waves per group
  for (int stride = p.WM >> 1; stride >= 1; stride >>= 1) {
    if (wm >= stride && wm < 2 * stride) {
      float* dst = red + (size_t)((wm - stride) * nwt + wr) * (TP * TQ * 16 * 64);
#pragma unroll
      for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) dst[((i * TQ + j) * 16 + e) * 64 + lane] = acc[i][j][e];
    }
    __syncthreads();
    if (wm < stride) {
      const float* src = red + (size_t)(wm * nwt + wr) * (TP * TQ * 16 * 64);
#pragma unroll
      for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += src[((i * TQ + j) * 16 + e) * 64 + lane];
    }
    __syncthreads();
  }
  if (wm == 0) {
    float* out = p.slabs + (size_t)split * p.P * p.Q;
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
      for (int j = 0; j < TQ; ++j) {
        const int col = c0 + j * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = co0 + i * 32 + acc_row(e, lane);
          out[(size_t)co * p.Q + col] = acc[i][j][e];
        }
      }
  }
}

// pfr_set_tuning("swgrad"): 0 never, 1 where measured faster (default), 2 wherever the geometry allows
static int swgrad_mode() { return pfr_knob(KNOB_SWGRAD); }
struct SwgradPlan { int gp, gq, wp, wq, wm, tilesP, tilesQ, nsplit; };
// geometry-only decision (pfr_conv2d_wgrad_splits sees only M, Cout, KK: a 3x3 layer with the same (Cout, KK) merely gets room
// for this many slabs; the launcher takes the streaming kernel only for 1x1 / stride-1 bf16 launches without a fused prologue)
// any_mode: the decision under mode 2 whatever the current mode is (workspace sizing: a plan built under one mode must have room for the
// slabs of a launch under another — pfr_set_tuning can change the mode between the two)
static bool swgrad_plan(int M, int P, int Q, SwgradPlan* sp, bool any_mode = false) {
  const int mode = any_mode ? 2 : swgrad_mode();
  if (mode == 0 || P % 64 || Q % 64 || P < 64 || Q < 64 || P > 1024 || Q > 1024) return false;
  int gp, gq;
  if (P >= 128 && P >= Q) { gp = 2; gq = 1; } else if (Q >= 128) { gp = 1; gq = 2; } else { gp = 1; gq = 1; }
  // workgroup tile: up to 256 x 128 / 128 x 256 / 256 x 256 channels in 8 waves
  int pw = P, qw = Q;
  if (pw > 256) pw = 256;
  if (qw > 256) qw = 256;
  if (P % pw || Q % qw || pw % (gp * 64) || qw % (gq * 64)) return false;
  int wp = pw / (gp * 64), wq = qw / (gq * 64);
  if (wp * wq > 8) { if (qw > 128 && gq == 1) qw = 128; else if (pw > 128 && gp == 1) pw = 128; wp = pw / (gp * 64); wq = qw / (gq * 64); }
  if (wp * wq > 8 || 8 % (wp * wq) || P % pw || Q % qw) return false;
  const int wm = 8 / (wp * wq);
  const int tilesP = P / pw, tilesQ = Q / qw, ntile = tilesP * tilesQ;
  if (ntile > 32 || 256 % ntile) return false;
  int nsplit = 256 / ntile;
  const long slab = (long)P * Q * 4;
  while (nsplit > 8 && nsplit * slab > (48L << 20)) nsplit >>= 1;
  const long nb = ((long)M + 15) / 16;
  if (nb < (long)nsplit * wm * 8) return false;     // too few steps per wave for a pipeline
  if (mode == 1) {
    // measured (tools/wgrad_bench.py): the private copies pay only where one operand is narrow
    const int amp_num = P * wq + Q * wp, amp_den = P + Q;   // LDS-DMA volume over unique bytes, per workgroup tile
    if (amp_num * 10 > amp_den * 13 || M < 100000) return false;
  }
  sp->gp = gp; sp->gq = gq; sp->wp = wp; sp->wq = wq; sp->wm = wm; sp->tilesP = tilesP; sp->tilesQ = tilesQ; sp->nsplit = nsplit;
  return true;
}
template <int GP, int GQ, int NS>
static void swgrad_go(const SwgradParams& sp, hipStream_t st) {
  constexpr int lds = 8 * NS * (GP + GQ) * 2048;
  static std::atomic<unsigned long long> attr{0};
  PFR_MAX_LDS_ONCE(attr, lds, (const void*)swgrad_kernel<GP, GQ, NS>);
  hipLaunchKernelGGL((swgrad_kernel<GP, GQ, NS>), dim3(256), dim3(512), lds, st, sp);
}
static int swgrad_launch(const WgradParams& p, const SwgradPlan& pl, float* slabs, hipStream_t st) {
  SwgradParams sp;
  sp.x = p.x; sp.dy = p.dy; sp.slabs = slabs;
  sp.M = p.M; sp.P = p.Cout; sp.Q = p.KK; sp.lddy = p.lddy;
  sp.nsplit = pl.nsplit; sp.tilesP = pl.tilesP; sp.tilesQ = pl.tilesQ;
  sp.WP = pl.wp; sp.WQ = pl.wq; sp.WM = pl.wm;
  const long nb = ((long)p.M + 15) / 16, per = (long)pl.nsplit * pl.wm;
  sp.nit = (int)((nb + per - 1) / per);
  if (pl.gp == 2) swgrad_go<2, 1, 3>(sp, st);
  else if (pl.gq == 2) swgrad_go<1, 2, 3>(sp, st);
  else swgrad_go<1, 1, 4>(sp, st);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// ------------------------------------------------------------------------------------------------
// Gram matrix + column sums of an activation tensor in ONE streaming pass: G2 = XᵀX [Q][Q] and s = column sums of X [Q]
// (X [M][Q] bf16, Q = 64 or 128) — the two forward-only quantities of the BN-input-free backward (pfr_bnfree.hip).  The structure
// of swgrad_kernel with a single operand: 256 persistent workgroups, every wave a private LDS-DMA ring of [16 rows][64 channels]
// blocks, no barrier in the loop; a wave owns 64 rows of G2 (its column block, WA = Q/64 of them) and ALL Q columns, the WM = 8/WA
// wave groups take different 16-row steps.  The transposing reads of a block serve as A and as B operand; the column sums are
// one more MFMA against a fragment of ones.  One fp32 slab [Q*Q + Q] per workgroup, summed by wgrad_reduce_kernel.
struct GramParams {
  const void* x;
  float* slabs;     // [nsplit][Q*Q + Q]
  int M, Q, nsplit, nit;
};
template <int GQ, int NS>
__global__ __launch_bounds__(512, 1) void gram_kernel(GramParams p) {
  constexpr int SLOT = GQ * 2048, IPS = GQ * 2, TQ = 2 * GQ, WA = GQ, WM = 8 / GQ;
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WA, wa = wave % WA;
  const int split = blockIdx.x;
  if (split >= p.nsplit) return;
  const int drow = lane >> 3, pc = lane & 7;
  uint32_t voff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * h + drow;
    const int lc = ((((pc >> 1) ^ (((row >> 1) & 1) << 1)) << 1) | (pc & 1));
    voff[h] = (uint32_t)((row * p.Q + lc * 8) * 2);
  }
  const char* xb = reinterpret_cast<const char*>(p.x);
  char* const ring = smem + wave * (NS * SLOT);
  const long nb = ((long)p.M + 15) >> 4;
  auto issue = [&](int slot, long it) __attribute__((always_inline)) {
    const long blk = (it * p.nsplit + split) * WM + wm;
    const long m0 = blk << 4;
    const long left = blk < nb ? (long)p.M - m0 : 0;
    const long rows = left > 16 ? 16 : left;
    const long mb = blk < nb ? m0 : 0;
    __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb + mb * p.Q * 2), 0, (int)(rows * p.Q * 2), 0x00020000);
    char* base = ring + slot * SLOT;
#pragma unroll
    for (int j = 0; j < GQ; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(br, (__attribute__((address_space(3))) void*)(base + j * 2048 + h * 1024), 16,
                                                 (int)voff[h], j * 128, 0, 0);
  };
  f32x16 acc[2][TQ], accz[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) accz[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < TQ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }
  const int g = lane >> 4, s4 = lane & 15;
  const int r0 = (g >> 1) * 8 + (s4 >> 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)(wave * (NS * SLOT));
  uint32_t laneT[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int gran = 2 * t + (g & 1);
    laneT[t] = lds0 + r0 * 128 + ((gran ^ (((r0 >> 1) & 1) << 1)) << 5) + (s4 & 3) * 8;
  }
  const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s, s);
  auto step = [&](auto slot_c, long it) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value;
    issue((S + NS - 1) % NS, it + NS - 1);
    wg_wait_vm<(NS - 1) * IPS>();
    u32x2 hq[TQ][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < TQ; ++j) hq[j][q] = lds_read_tr16(laneT[j & 1], S * SLOT + (j >> 1) * 2048 + q * 512);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < TQ; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(hq[j][q]));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // this wave's 64 rows of G2: the 32-channel halves 2*wa + i of the block are the A operand (wa is wave-uniform)
      u32x4 ua;
      if (GQ == 1 || wa == 0) ua = (u32x4){hq[i][0][0], hq[i][0][1], hq[i][1][0], hq[i][1][1]};
      else ua = (u32x4){hq[(TQ > 2 ? 2 : 0) + i][0][0], hq[(TQ > 2 ? 2 : 0) + i][0][1], hq[(TQ > 2 ? 2 : 0) + i][1][0], hq[(TQ > 2 ? 2 : 0) + i][1][1]};
#pragma unroll
      for (int j = 0; j < TQ; ++j) {
        const u32x4 ub = {hq[j][0][0], hq[j][0][1], hq[j][1][0], hq[j][1][1]};
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[i][j], 0, 0, 0);
      }
      accz[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ones), accz[i], 0, 0, 0);
    }
  };
  for (long it = 0; it < p.nit; it += NS) {
    step(std::integral_constant<int, 0>{}, it);
    if (it + 1 < p.nit) step(std::integral_constant<int, 1>{}, it + 1);
    if constexpr (NS > 2) { if (it + 2 < p.nit) step(std::integral_constant<int, 2>{}, it + 2); }
    if constexpr (NS > 3) { if (it + 3 < p.nit) step(std::integral_constant<int, 3>{}, it + 3); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    asm volatile("s_nop 15" : "+v"(accz[i]));
#pragma unroll
    for (int j = 0; j < TQ; ++j) asm volatile("s_nop 15" : "+v"(acc[i][j]));
  }
  __syncthreads();
  // ---- sum the WM wave groups (tree over LDS, fixed order)
  float* red = reinterpret_cast<float*>(smem);
  constexpr int NT = 2 * TQ + 2;     // accumulator tiles per wave
  for (int stride = WM >> 1; stride >= 1; stride >>= 1) {
    if (wm >= stride && wm < 2 * stride) {
      float* dst = red + (size_t)((wm - stride) * WA + wa) * (NT * 16 * 64);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) dst[((i * TQ + j) * 16 + e) * 64 + lane] = acc[i][j][e];
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[((2 * TQ + i) * 16 + e) * 64 + lane] = accz[i][e];
      }
    }
    __syncthreads();
    if (wm < stride) {
      const float* src = red + (size_t)(wm * WA + wa) * (NT * 16 * 64);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += src[((i * TQ + j) * 16 + e) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 16; ++e) accz[i][e] += src[((2 * TQ + i) * 16 + e) * 64 + lane];
      }
    }
    __syncthreads();
  }
  if (wm == 0) {
    float* out = p.slabs + (size_t)split * ((size_t)p.Q * p.Q + p.Q);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < TQ; ++j) {
        const int col = j * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) out[(size_t)(wa * 64 + i * 32 + acc_row(e, lane)) * p.Q + col] = acc[i][j][e];
      }
      if ((lane & 31) == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) out[(size_t)p.Q * p.Q + wa * 64 + i * 32 + acc_row(e, lane)] = accz[i][e];
      }
    }
  }
}

// (round 6: an LDS-free form — the four split-lanes as adjacent lanes of one wave, merged by two shuffles, bit-identical — lets the launch run
//  beside the main stream's 160 KB-LDS streaming workgroups instead of waiting 35-85 us for them; the step got 0.09 ms SLOWER (17.35 vs 17.26 ms,
//  four interleaved rounds, profiles/r06_ab.txt): what the side stream gains the main stream's kernels lose.  Not kept.)
// sums the split slabs: block = 64 columns (of 4 floats) x 4 split-lanes; every lane keeps 4 independent loads in flight
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, size_t n4,
                                                           int splits, float scale, int accumulate) {
  __shared__ f32x4 red[4][64];
  const int cx = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + cx;  // index of a 4-float group
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (i < n4) {
    const f32x4* p4 = reinterpret_cast<const f32x4*>(part) + i;
    int s = sl;
    for (; s + 12 < splits; s += 16) {
      a0 += p4[(size_t)s * n4];
      a1 += p4[(size_t)(s + 4) * n4];
      a2 += p4[(size_t)(s + 8) * n4];
      a3 += p4[(size_t)(s + 12) * n4];
    }
    for (; s < splits; s += 4) a0 += p4[(size_t)s * n4];
  }
  red[sl][cx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && i < n4) {
    f32x4 a = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
    a *= scale;
    f32x4* o = reinterpret_cast<f32x4*>(dw) + i;
    if (accumulate) a += *o;
    *o = a;
  }
}

// halo-staged 3x3 / stride-1 weight gradient (pfr_wgrad9.hip)
int wgrad9_launch(const void* x, const void* dy, float* slabs, int N, int H, int W, int C, int Cout, int R, int S, int stride, int pad,
                  int lddy, hipStream_t st);
int wgrad9_max_splits(int Cout, int KK);   // (mode-independent: the slabs the kernel WOULD write when enabled)

static int wgrad_v3() { return 1; }   // (the round-2 kernel without the VALU-free stage loop lost its A/B: profiles/HISTORY.md)

template <typename T, int BP, int BQ>
static int launch_wgrad(WgradParams& p, hipStream_t st) {
  p.tilesP = (p.Cout + BP - 1) / BP;
  p.tilesQ = (p.KK + BQ - 1) / BQ;
  const dim3 grid((unsigned)(p.tilesP * p.tilesQ * p.splits));
  if (p.pro_scale)
    hipLaunchKernelGGL((wgrad_kernel<T, BP, BQ, true>), grid, dim3(256), 0, st, p);
  else if (p.v2 && sizeof(T) == 2 && wgrad_v3()) {
    constexpr int lds = PFR_WGRAD_NST * 32 * (BP + BQ) * 2;   // ring of NST stages of 32 rows
    static std::atomic<unsigned long long> attr{0};
    PFR_MAX_LDS_ONCE(attr, lds, (const void*)wgrad3_kernel<BP, BQ>);
    hipLaunchKernelGGL((wgrad3_kernel<BP, BQ>), grid, dim3(256), lds, st, p);
  }
  else if (p.v2)
    hipLaunchKernelGGL((wgrad2_kernel<T, BP, BQ>), grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((wgrad_kernel<T, BP, BQ, false>), grid, dim3(256), 0, st, p);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

static void wgrad_tiles(int Cout, int KK, int* bp, int* bq) {
  *bp = Cout >= 128 ? 128 : 64;
  *bq = KK >= 128 ? 128 : 64;
  const int forced = pfr_knob(KNOB_WGRAD_TILE);   // tuning: tools/tile_sweep.py
  if (forced >= 0) { *bp = (forced & 1) ? 64 : 128; *bq = (forced & 2) ? 64 : 128; }
}

// 8-wave 256x256 tiles (wgrad3_kernel<256, 256, 8, 2>; bf16, no fused prologue).  pfr_set_tuning("wgrad_big"):
// 0 never (default), 1 whole-tile geometries, 2 wherever the geometry allows.  MEASURED (tools/wgrad_bench.py,
// profiles/r03_wgrad_tiles.txt): half the operand bytes per MFMA, yet 0.84-0.89x on most ResNet-50 / Swin-T geometries and at
// best 1.06-1.18x on three — one 8-wave workgroup per CU behind ONE barrier per 32-row stage hides less latency than two
// independent 4-wave workgroups; the kernel is bound by that hand-over, not by LDS or L2 bandwidth.
static int wgrad_big_mode() { return pfr_knob(KNOB_WGRAD_BIG); }
static bool wgrad_big_geom(int M, int Cout, int KK) {
  const int mode = wgrad_big_mode();
  if (mode == 0 || Cout < 256 || KK < 256 || Cout % 8 || KK % 8) return false;
  if (mode >= 2) return true;
  return Cout % 256 == 0 && KK % 256 == 0;
}

// number of m-splits the launcher uses (the caller sizes the workspace as splits*Cout*KK floats).
// Two workgroups fit a CU (64 KB of LDS each) and a CU runs two about as fast as one (the kernel waits on DMA latency), so a
// launch of W = tiles*s workgroups takes ceil(W / 512) rounds of M/s rows: s is chosen to minimise
//     t_pass * (ceil(W/512)*512 / W)  +  s * slab * 2 / 4 TB/s          (fp32 partial slabs written once, read once)
// with t_pass = the pass at full occupancy (600 TFLOP/s or 4 TB/s of operand bytes, whichever is slower).  Measured on MI355X
// (profiles/repro/split_sweep.sh): 3x3 256ch 14x14: 10 -> 14 splits = 98 -> 82 us; 3x3 512ch 7x7: 4 -> 3 splits = 115 -> 91 us.
static int wgrad_tile_splits(int M, int Cout, int KK) {
  int bp, bq;
  wgrad_tiles(Cout, KK, &bp, &bq);
  // (the 256x256 form is only taken by bf16 launches without a fused prologue; others run 128-wide tiles over the same number of
  // slabs, which is merely a different trade-off, not an error)
  const bool big = wgrad_big_geom(M, Cout, KK);
  if (big) bp = bq = 256;
  const long slots = big ? 256 : 512;
  const long tiles = (long)((Cout + bp - 1) / bp) * ((KK + bq - 1) / bq);
  const long maxs = (M + 255) / 256;       // at least 256 reduction rows per split
  const long slab = (long)Cout * KK * 4;
  constexpr long slab_mb = 48;
  long cap = (slab_mb << 20) / slab;
  if (cap > maxs) cap = maxs;
  if (cap < 1) cap = 1;
  const double t_flop = 2.0 * M * Cout * KK / 6.0e14, t_byte = 2.0 * M * ((double)Cout + KK) / 4.0e12;
  const double t_pass = t_flop > t_byte ? t_flop : t_byte;
  long best = 1;
  double best_t = 1e30;
  for (long s = 1; s <= cap && s * tiles <= 4096; ++s) {
    const long w = tiles * s, rounds = (w + slots - 1) / slots;
    const double t = t_pass * (double)(rounds * slots) / (double)w + (double)s * slab * 2.0 / 4.0e12;
    if (t < best_t * 0.999) { best_t = t; best = s; }
  }
  const int forced = pfr_knob(KNOB_WGRAD_SPLITS);   // tuning sweeps (a host that raises it re-queries pfr_conv2d_wgrad_splits: epoch)
  if (forced > 0) best = forced < maxs ? forced : maxs;
  return (int)best;
}
// the caller's workspace: room for the slabs of whichever kernel takes the launch — under ANY setting of the tuning knobs, so that a
// workspace sized when a plan was built is large enough for a launch after a later pfr_set_tuning("wgrad9" / "swgrad", ...)
extern "C" int pfr_conv2d_wgrad_splits(int M, int Cout, int KK) {
  int best = wgrad_tile_splits(M, Cout, KK);
  SwgradPlan spl;
  if (swgrad_plan(M, Cout, KK, &spl, true) && spl.nsplit > best) best = spl.nsplit;   // (the streaming 1x1 kernel's)
  if (wgrad9_max_splits(Cout, KK) > best) best = wgrad9_max_splits(Cout, KK);     // (the halo-staged 3x3 kernel's)
  return best;
}

extern "C" int pfr_conv2d_wgrad(const void* x, const void* dy, float* dw, float* workspace, int dtype, int N, int H,
                                int W, int C, int Cout, int R, int S, int stride, int pad, int OH, int OW, int lddy,
                                const float* pro_scale, const float* pro_shift, int pro_relu, float scale,
                                int accumulate, hipStream_t stream) {
  PFR_CHECK_ARG(x && dy && dw, "pfr_conv2d_wgrad: null pointer");
  PFR_CHECK_ARG(dtype == PFR_F32 || dtype == PFR_BF16, "pfr_conv2d_wgrad: bad dtype %d", dtype);
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  PFR_CHECK_ARG(C % kp == 0, "pfr_conv2d_wgrad: C=%d must be a multiple of %d", C, kp);
  PFR_CHECK_ARG(Cout % kp == 0 && (lddy <= 0 || lddy % kp == 0), "pfr_conv2d_wgrad: Cout=%d / lddy must be multiples of %d", Cout, kp);
  WgradParams p;
  p.x = x; p.dy = dy;
  p.N = N; p.H = H; p.W = W; p.C = C; p.R = R; p.S = S; p.OH = OH; p.OW = OW; p.stride = stride; p.pad = pad;
  p.Cout = Cout; p.lddy = lddy > 0 ? lddy : Cout; p.M = N * OH * OW; p.KK = R * S * C;
  p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.pro_relu = pro_relu;
  p.div_ohow = make_fastdiv((uint32_t)(OH * OW));
  p.div_ow = make_fastdiv((uint32_t)OW);
  p.splits = wgrad_tile_splits(p.M, Cout, p.KK);
  const int ws_splits = pfr_conv2d_wgrad_splits(p.M, Cout, p.KK);
  { const int ohow = OH * OW, r1 = 32 % ohow; p.adv_q1 = 32 / ohow; p.adv_q2 = r1 / OW; p.adv_r2 = r1 % OW; }
  p.v2 = pro_scale ? 0 : 1;
  const int bmr = p.v2 ? (dtype == PFR_BF16 ? 64 : 32) : PFR_WGRAD_MUL * (dtype == PFR_BF16 ? 32 : 16);
  p.simple = (R == 1 && S == 1 && stride == 1 && pad == 0) ? 1 : 0;
  int mchunk = (p.M + p.splits - 1) / p.splits;
  mchunk = (mchunk + bmr - 1) / bmr * bmr;
  p.mchunk = mchunk;
  const bool direct = (ws_splits == 1 && scale == 1.0f && !accumulate);   // no slabs, no reduce pass
  PFR_CHECK_ARG(direct || workspace, "pfr_conv2d_wgrad: workspace required (splits=%d)", p.splits);
  p.dw = direct ? dw : workspace;
  int bp, bq, rc;
  wgrad_tiles(Cout, p.KK, &bp, &bq);
  SwgradPlan spl;
  int n9 = 0;
  if (dtype == PFR_BF16 && p.v2 && !p.simple)
    n9 = wgrad9_launch(x, dy, p.dw, N, H, W, C, Cout, R, S, stride, pad, p.lddy, stream);
  if (n9 < 0) { pfr_set_error("pfr_conv2d_wgrad: launch failed"); return PFR_ERR_HIP; }
  if (n9 > 0) {
    p.splits = n9;
    rc = PFR_OK;
  } else if (dtype == PFR_BF16 && p.v2 && p.simple && !direct && p.lddy % 8 == 0 && (long)p.M * p.lddy * 2 < (1L << 31) &&
      (long)p.M * p.KK * 2 < (1L << 31) && swgrad_plan(p.M, Cout, p.KK, &spl)) {
    rc = swgrad_launch(p, spl, workspace, stream);
    if (rc != PFR_OK) return rc;
    p.splits = spl.nsplit;
  } else if (dtype == PFR_BF16 && p.v2 && wgrad_v3() && wgrad_big_geom(p.M, Cout, p.KK)) {
    p.tilesP = (p.Cout + 255) / 256;
    p.tilesQ = (p.KK + 255) / 256;
    constexpr int lds = PFR_WGRAD_NST * 32 * (256 + 256) * 2;
    static std::atomic<unsigned long long> attr{0};
    PFR_MAX_LDS_ONCE(attr, lds, (const void*)wgrad3_kernel<256, 256, 8, 2>);
    hipLaunchKernelGGL((wgrad3_kernel<256, 256, 8, 2>), dim3((unsigned)(p.tilesP * p.tilesQ * p.splits)), dim3(512), lds, stream, p);
    PFR_CHECK_LAUNCH();
    rc = PFR_OK;
  } else {
#define PFR_WG_DISPATCH(T)                                        \
  if (bp == 128 && bq == 128) rc = launch_wgrad<T, 128, 128>(p, stream); \
  else if (bp == 128) rc = launch_wgrad<T, 128, 64>(p, stream);   \
  else if (bq == 128) rc = launch_wgrad<T, 64, 128>(p, stream);   \
  else rc = launch_wgrad<T, 64, 64>(p, stream);
  if (dtype == PFR_BF16) { PFR_WG_DISPATCH(bf16_t) } else { PFR_WG_DISPATCH(float) }
#undef PFR_WG_DISPATCH
  }
  if (rc != PFR_OK) return rc;
  if (!direct) {
    const size_t n4 = (size_t)Cout * p.KK / 4;  // Cout*KK is a multiple of 16 (both are multiples of the 16-byte chunk)
    const unsigned blocks = (unsigned)((n4 + 63) / 64);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, workspace, dw, n4, p.splits, scale, accumulate);
    PFR_CHECK_LAUNCH();
  }
  return PFR_OK;
}

// G2 = XᵀX and the column sums of X in one pass (csrc/pfr_bnfree.hip's forward-only quantities).  out: [Q*Q] then [Q] floats;
// workspace: pfr_gram_ws_floats floats (0: geometry not supported — use pfr_conv2d_wgrad(x, x) + pfr_colsum)
extern "C" long pfr_gram_ws_floats(long M, int Q) {
  if ((Q != 64 && Q != 128) || M < 16 * 256 || M * Q * 2 >= (1L << 31)) return 0;
  return 256L * ((long)Q * Q + Q);
}
extern "C" int pfr_gram_colsum(const void* x, int dtype, long M, int Q, float* out, float* workspace, hipStream_t st) {
  PFR_CHECK_ARG(x && out && workspace, "pfr_gram_colsum: null pointer");
  PFR_CHECK_ARG(dtype == PFR_BF16 && pfr_gram_ws_floats(M, Q) > 0, "pfr_gram_colsum: bf16, Q = 64 | 128, M >= 4096 only");
  GramParams gp;
  gp.x = x; gp.slabs = workspace; gp.M = (int)M; gp.Q = Q; gp.nsplit = 256;
  const int wm = Q == 64 ? 8 : 4;
  const long nb = (M + 15) / 16, per = 256L * wm;
  gp.nit = (int)((nb + per - 1) / per);
  if (Q == 64) {
    static std::atomic<unsigned long long> attr{0};   // (LDS: the rings, then — reused — the wave-group sums: 4 waves x 6 tiles / 4 waves x 10 tiles of 4 KiB)
    PFR_MAX_LDS_ONCE(attr, 160 * 1024, (const void*)gram_kernel<1, 4>);
    hipLaunchKernelGGL((gram_kernel<1, 4>), dim3(256), dim3(512), 160 * 1024, st, gp);
  } else {
    static std::atomic<unsigned long long> attr{0};
    PFR_MAX_LDS_ONCE(attr, 160 * 1024, (const void*)gram_kernel<2, 3>);
    hipLaunchKernelGGL((gram_kernel<2, 3>), dim3(256), dim3(512), 160 * 1024, st, gp);
  }
  PFR_CHECK_LAUNCH();
  const size_t n4 = ((size_t)Q * Q + Q) / 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, workspace, out, n4, 256, 1.0f, 0);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
