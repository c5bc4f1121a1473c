// pfr_wgrad9.hip — weight gradient of the 3x3 / stride-1 / pad-1 convolutions (bf16), halo-staged.
//
// Replaces the autograd weight gradient of torchvision's `conv3x3` in BasicBlock / Bottleneck.conv2 (third-party to
// /root/reference; the backbone is built at configs/dog_fe/fe_dogs_config.py:102-103):
//
//   dw[co][tr][ts][ci] = sum_{n,oh,ow} dy[n,oh,ow,co] * x[n, oh+tr-1, ow+ts-1, ci]
//
// The tile kernel (wgrad3_kernel, pfr_wgrad.hip) treats the nine taps as nine independent slices of the GEMM's kk axis: x is
// gathered from L2 once per tap and dy once per 128 kk-columns, 1.4 GB (64->64 at 56x56) / 0.9 GB (256->256 at 14x14) of
// L2->LDS traffic per launch = 13 TB/s — that kernel sits on the L2, at 0.23-0.30 of its MFMA / HBM bound.
//
// Here a workgroup owns a [64 co] x [9 taps] x [64 ci] block of dw over a run of image rows.  Rows of dy and of x are staged into
// LDS in padded row slots [0 | p_0 .. p_{W-1} | 0 ..] of PW = 8 / 16 / 32 / 64 pixel positions (128 B each: the 64 channels of the
// block), and the row sequence is EXTENDED by one all-zero row after every image (row index e = n (H + 1) + r).  In that layout a
// tap is a pure ADDRESS SHIFT: for the 16 positions of a k-group of dy, the matching x positions are the same positions + (ts - 1)
// in the row slot tr - 1 below / above — an immediate offset on a lane base — and everything outside the image reads zeros that the
// DMA wrote (out-of-range lanes of a buffer load).  Pad positions of dy are zero, so whatever finite x value sits opposite them
// contributes nothing; the MFMA work is (W / PW)(H / (H + 1)) = 0.82-0.86 efficient.
//
//   * stage = 128 dy positions (SR = 2 / 4 / 8 extended rows) + the SR + 2 rows of x around them, ring of 3 stages filled by
//     LDS-DMA; L2->LDS traffic = 0.3 GB at 56x56 (x rows twice: a stage brings its own halo rows), 0.25 GB at 14x14;
//   * 8 waves = (ci half) x (k-group parity) x (tap group: taps 0-4 | 5-8); a wave multiplies BOTH co halves of a k-group of dy
//     with its taps' x fragments: 1.44 LDS fragment reads per MFMA (a 32x32 wave tile over all nine taps needs 2.22);
//   * one barrier per stage, one step before the stage ends (see the main loop); the two waves of a SIMD issue their DMA shares one
//     step apart;
//   * the rows a wave's DMA instructions fetch are tabulated per 64 stages (v_readlane in the loop);
//   * partial sums per (split, block pair) go to fp32 slabs [split][Cout][9*C] summed by wgrad_reduce_kernel (fixed order).
//
// Measured (bs 256, tools/wgrad_bench.py, incl. the reduce pass): 64->64 at 56x56 114 -> 68 us, 128->128 at 28x28 84 -> 75 us,
// 256->256 at 14x14 84 -> 70 us.  What the kernel time is made of (profiles/r04_wgrad9.txt): ~34 us of MFMAs at the sustained
// clock, ~10 us of stage hand-overs / DMA issue, ~4 us prologue, ~12 us epilogue (the parity hand-over through LDS and 37.7 MB of
// slab stores: 256 workgroups x 147 KB, the price of one accumulator set per CU).
#include "pfr_mma.h"
#include <stdlib.h>

struct Wg9Params {
  const void* x;
  const void* dy;
  float* slabs;
  int H, W, C, Cout, lddy;
  int nrows;   // N * (H + 1) extended rows
  int rps;     // extended rows per split (a multiple of the stage's rows)
  int nsplit, ncb, nib;
  FastDiv div_h;   // by H + 1
};

__device__ __forceinline__ u32x2 w9_read_tr16(uint32_t addr, int imm) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(imm) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void w9_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void w9_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int w9_swz(int row) { return ((row >> 1) & 1) << 1; }   // 32-byte granule XOR of a 128-byte row (as gran_swz<4>)

template <int PW>
__global__ __launch_bounds__(512, 1) void wgrad9_kernel(Wg9Params p) {
  constexpr int SR = 128 / PW;          // extended image rows per stage
  constexpr int IPR = PW / 8;           // 1-KiB DMA instructions per row slot
  constexpr int NST = 3;
  constexpr int SLOTB = PW * 128, GST = 16384, XST = (SR + 2) * SLOTB;
  constexpr int XR = NST * GST + 1024, END = XR + NST * XST + 1024;
  constexpr int XI = (SR + 2) * IPR, IPW = (16 + XI + 7) / 8;   // DMA instructions of a stage: 16 of dy, XI of x; per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave roles: ci half hq, k-group parity grp, tap group tg (taps 0-4 / 5-8; waves w and w + 4 share a SIMD: one of each group)
  const int hq = wave & 1, grp = (wave >> 1) & 1, tg = wave >> 2;

  // block -> (split, co block, ci block): the block pairs of one split share their rows of dy and x, keep them on one XCD
  const int npairs = p.ncb * p.nib;
  int split, pair;
  {
    const int t = blockIdx.x;
    if ((p.nsplit & 7) == 0) {
      const int xcd = t & 7, l = t >> 3;
      split = xcd * (p.nsplit >> 3) + l / npairs;
      pair = l % npairs;
    } else {
      split = t / npairs;
      pair = t % npairs;
    }
  }
  const int cb = pair / p.nib, ib = pair % p.nib;
  const int H = p.H, W = p.W, C = p.C, lddy = p.lddy, nrows = p.nrows;   // nrows: EXTENDED rows, N * (H + 1)
  const int ebeg = split * p.rps;
  const int eend = min(ebeg + p.rps, nrows);
  const int nst = p.rps / SR;
  const FastDiv div_h1 = p.div_h;   // by H + 1

  // ---- LDS starts as zeros (guards; positions a row slot's neighbours are read at before their first DMA)
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int off = tid * 16; off < END; off += 512 * 16) *reinterpret_cast<u32x4*>(smem + off) = z;
    __syncthreads();
  }

  // ---- DMA lane geometry: instruction j covers positions [8*(j % IPR), +8) of row slot j / IPR; lane l -> position l / 8, physical
  // 16-byte chunk l % 8 of the 128-byte pixel, which holds the logical chunk given by the read-side swizzle
  const int part = wave % IPR;
  const uint32_t OOB = 0x80000000u;   // (+ a row offset < 2 GiB: no 32-bit wrap whichever way the bounds check counts the SGPR offset)
  uint32_t voffG, voffX;
  {
    const int pos = part * 8 + (lane >> 3), c = lane & 7;
    const int lc = ((((c >> 1) ^ w9_swz(pos)) << 1) | (c & 1));
    const int px = pos - 1;
    const bool ok = px >= 0 && px < W;
    voffG = ok ? (uint32_t)((px * lddy + cb * 64 + lc * 8) * 2) : OOB;
    voffX = ok ? (uint32_t)((px * C + ib * 64 + lc * 8) * 2) : OOB;
  }
  const char* dyb = reinterpret_cast<const char*>(p.dy);
  const char* xb = reinterpret_cast<const char*>(p.x);
  const int growb = W * lddy * 2, xrowb = W * C * 2;

  // extended row e = n * (H + 1) + r: image row r of image n, r == H is the all-zero row between two images.
  // The rows a wave's DMA instructions fetch are tabulated 64 stages at a time: tab[jj] holds, in lane l, the byte offset of the
  // tensor row instruction jj reads in stage tbase + l (~0: no such row — out of the split / the tensor, or a zero row).  In the
  // loop an instruction costs v_readlane + add + compare + select: the row offset is the buffer instruction's SGPR offset and
  // num_records = row offset + one row's bytes (0 for a missing row): gfx950 range-checks lane offset + SGPR offset (measured: with
  // num_records = one row's bytes every row but the first came back as zeros), and this bound is right under either reading; pad
  // lanes sit 2 GiB out (tensors here stay below 2 GiB: no 32-bit wrap).
  uint32_t tab[IPW];
  auto build = [&](int tbase) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < IPW; ++jj) {
      const int id = wave + 8 * jj;
      const int rowoff = jj < 2 ? id / IPR : (id - 16) / IPR - 1;
      const int e = ebeg + (tbase + lane) * SR + rowoff;
      const uint32_t ue = (uint32_t)max(e, 0);
      const uint32_t n = (uint32_t)(((uint64_t)__umulhi(ue, div_h1.mul) + ue) >> div_h1.shr);
      const uint32_t r = ue - n * (uint32_t)(H + 1);
      const bool ok = e >= 0 && e < (jj < 2 ? eend : nrows) && r < (uint32_t)H;
      tab[jj] = ok ? (n * (uint32_t)H + r) * (uint32_t)(jj < 2 ? growb : xrowb) : 0xFFFFFFFFu;
    }
  };
  auto issue = [&](int li, int slot) __attribute__((always_inline)) {   // li: stage - tbase
#pragma unroll
    for (int jj = 0; jj < IPW; ++jj) {
      const int id = wave + 8 * jj;
      if (jj < 2) {          // dy rows e0 .. e0 + SR - 1
        const uint32_t so = (uint32_t)__builtin_amdgcn_readlane((int)tab[jj], li);
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dyb), 0, so != 0xFFFFFFFFu ? (int)so + growb : 0, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + slot * GST + id * 1024), 16,
                                                 (int)voffG, (int)so, 0, 0);
      } else if (8 * (jj - 2) + 7 < XI || id - 16 < XI) {   // x rows e0 - 1 .. e0 + SR (the halo rows of a stage are staged with it)
        const int jx = id - 16;
        const uint32_t so = (uint32_t)__builtin_amdgcn_readlane((int)tab[jj], li);
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), 0, so != 0xFFFFFFFFu ? (int)so + xrowb : 0, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + XR + slot * XST + jx * 1024), 16,
                                                 (int)voffX, (int)so, 0, 0);
      }
    }
  };

  f32x16 acc[5][2];   // [tap of the group][co half]
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][h][e] = 0.f;

  // ---- transpose-read lane bases (as wgrad3_kernel): row (g>>1)*8 + (s4>>2) of a k-group, swizzled 32-byte granule, 8-byte piece.
  // k-group kgi = 2 i + parity of a stage sits at byte kgi * 2048 of the stage's dy region AND of its x region (whose first row slot
  // is the halo row above): tap (tr, ts) of k-group i is the IMMEDIATE i * 4096 + tr * SLOTB on the lane base of column shift ts.
  const int g = lane >> 4, s4 = lane & 15;
  const int r0 = (g >> 1) * 8 + (s4 >> 2);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t laneA[2], laneB[3];
#pragma unroll
  for (int h = 0; h < 2; ++h) laneA[h] = lds0 + grp * 2048 + r0 * 128 + (((h * 2 + (g & 1)) ^ w9_swz(r0)) << 5) + (s4 & 3) * 8;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int rr = r0 + d - 1;   // position shift of tap column ts = d
    laneB[d] = (uint32_t)((int)lds0 + XR + grp * 2048 + rr * 128 + (((hq * 2 + (g & 1)) ^ w9_swz(rr)) << 5) + (s4 & 3) * 8);
  }

  // ---- prologue: stages 0 and 1; stage 0 handed over
  int tbase = 0;
  build(0);
  issue(0, 0);
  if (1 < nst) {
    issue(1, 1);
    if ((IPW - 3) * 8 + 7 < XI || wave + (IPW - 3) * 8 < XI) w9_wait_vm<IPW>();   // stage 1 stays in flight
    else w9_wait_vm<IPW - 1>();
  } else {
    w9_wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();

  // ---- main loop.  A wave's k-group is one step: NT taps, each [wait] [2 MFMAs: both co halves against the tap's x fragment]
  // [read the same tap's x fragment of the NEXT k-group into the registers just consumed]; the dy fragments of the next k-group
  // (two buffers) are read at the head of the step.  Every wait leaves exactly 2 NT + 2 younger reads in flight.  LDS reads per MFMA:
  // 1.44 (a 32x32 wave tile over all nine taps: 2.22 — measured at 0.55 of the matrix pipe, the LDS port and the issue slots of two
  // waves per SIMD both near their limits).
  // One barrier per stage, placed one step BEFORE the stage ends: stage s + 1 is handed over (every wave has waited for its own DMA
  // share, then the barrier) while the last step of stage s is still to be issued, and the first reads of stage s + 1 go out under
  // it.  The DMA issued after that barrier (stage s + 2) fills the slot of stage s - 1, which nobody reads any more.
  auto run = [&](auto tgc) __attribute__((always_inline)) {
    constexpr int TG = decltype(tgc)::value;
    constexpr int NT = TG == 0 ? 5 : 4, T0 = TG * 5;
    u32x2 hA[2][2][2], hB[NT][2];   // [buffer][co half][q], [tap][q]
    uint32_t aS[2] = {laneA[0], laneA[1]}, bS[3] = {laneB[0], laneB[1], laneB[2]};   // stage s
    uint32_t aN[2] = {laneA[0], laneA[1]}, bN[3] = {laneB[0], laneB[1], laneB[2]};   // stage s + 1
    auto rdA = [&](auto ic, const uint32_t (&ab)[2]) __attribute__((always_inline)) {
      constexpr int I = decltype(ic)::value;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        hA[I & 1][h][0] = w9_read_tr16(ab[h], I * 4096);
        hA[I & 1][h][1] = w9_read_tr16(ab[h], I * 4096 + 512);
      }
    };
    auto rdB = [&](auto ic, auto tc, const uint32_t (&bb)[3]) __attribute__((always_inline)) {
      constexpr int I = decltype(ic)::value, T = decltype(tc)::value;
      constexpr int DR = (T0 + T) / 3, DS = (T0 + T) % 3;
      hB[T][0] = w9_read_tr16(bb[DS], I * 4096 + DR * SLOTB);
      hB[T][1] = w9_read_tr16(bb[DS], I * 4096 + DR * SLOTB + 512);
    };
    auto mma = [&](auto ic, auto tc) __attribute__((always_inline)) {
      constexpr int I = decltype(ic)::value, T = decltype(tc)::value;
#pragma unroll
      for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(hB[T][q]));
      const u32x4 ub = {hB[T][0][0], hB[T][0][1], hB[T][1][0], hB[T][1][1]};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(hA[I & 1][h][q]));
        const u32x4 ua = {hA[I & 1][h][0][0], hA[I & 1][h][0][1], hA[I & 1][h][1][0], hA[I & 1][h][1][1]};
        acc[T][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[T][h], 0, 0, 0);
      }
    };
    // step I of a stage: k-group I is consumed, k-group I + 1 (of the next stage when I == 3) is read
    auto step = [&](auto ic) __attribute__((always_inline)) {
      constexpr int I = decltype(ic)::value;
      constexpr int J = (I + 1) & 3;
      if constexpr (I < 3) rdA(std::integral_constant<int, J>{}, aS);
      else {   // J == 0 of the next stage: buffer parity of k-group 4
        hA[0][0][0] = w9_read_tr16(aN[0], 0); hA[0][0][1] = w9_read_tr16(aN[0], 512);
        hA[0][1][0] = w9_read_tr16(aN[1], 0); hA[0][1][1] = w9_read_tr16(aN[1], 512);
      }
      auto tap = [&](auto tc) __attribute__((always_inline)) {
        w9_wait_lgkm<2 * NT + 2>();
        mma(ic, tc);
        if constexpr (I < 3) rdB(std::integral_constant<int, J>{}, tc, bS);
        else rdB(std::integral_constant<int, 0>{}, tc, bN);
      };
      tap(std::integral_constant<int, 0>{});
      tap(std::integral_constant<int, 1>{});
      tap(std::integral_constant<int, 2>{});
      tap(std::integral_constant<int, 3>{});
      if constexpr (NT == 5) tap(std::integral_constant<int, 4>{});
    };

    // the first k-group's fragments
    rdA(std::integral_constant<int, 0>{}, aS);
    rdB(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, bS);
    rdB(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, bS);
    rdB(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, bS);
    rdB(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, bS);
    if constexpr (NT == 5) rdB(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, bS);

    int slot = 0;   // ring slot of stage s
    for (int s = 0; s < nst; ++s) {
      const int slot1 = slot == NST - 1 ? 0 : slot + 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) aN[h] = laneA[h] + (uint32_t)(slot1 * GST);
#pragma unroll
      for (int d = 0; d < 3; ++d) bN[d] = laneB[d] + (uint32_t)(slot1 * XST);
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      // The two waves of a SIMD (tap groups 0 and 1) issue their DMA shares one step apart: an LDS-DMA instruction holds its wave for
      // 60-180 cycles, and with both waves in that state the SIMD's matrix pipe stands still (measured: 9 of 71 us).
      auto issue_next = [&]() __attribute__((always_inline)) {
        if (s + 2 < nst) {
          if (s + 2 - tbase == 64) { tbase = s + 2; build(tbase); }
          issue(s + 2 - tbase, slot == 0 ? NST - 1 : slot - 1);
        }
      };
      if (s + 1 < nst) {
        w9_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if constexpr (TG == 0) issue_next();
      }
      step(std::integral_constant<int, 3>{});   // (the last stage reads a stale slot ahead: never used)
      if constexpr (TG == 1) issue_next();
#pragma unroll
      for (int h = 0; h < 2; ++h) aS[h] = aN[h];
#pragma unroll
      for (int d = 0; d < 3; ++d) bS[d] = bN[d];
      slot = slot1;
    }
    w9_wait_lgkm<0>();
  };
  if (tg == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});

  // ---- the two k-group parities add up through LDS in the accumulators' own lane layout (16-byte pieces: 4 rows of a column), then
  // parity 0 lays the sums out as the block itself, [64 co][9 taps][64 ci], and the slab is stored from there in 16-byte pieces
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
  const int ntap = tg == 0 ? 5 : 4;
  f32x4* priv = reinterpret_cast<f32x4*>(smem) + (size_t)((tg == 0 ? hq * 10 : 20 + hq * 8) * 4) * 64 + lane;   // [acc][row quad][lane]
  if (grp == 1) {
#pragma unroll
    for (int t = 0; t < 5; ++t)
      if (t < ntap)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
            priv[((t * 2 + h) * 4 + r4) * 64] = f32x4{acc[t][h][r4 * 4], acc[t][h][r4 * 4 + 1], acc[t][h][r4 * 4 + 2], acc[t][h][r4 * 4 + 3]};
  }
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int t = 0; t < 5; ++t)
      if (t < ntap)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 o = priv[((t * 2 + h) * 4 + r4) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][h][r4 * 4 + e] += o[e];
          }
  }
  __syncthreads();
  if (grp == 0) {
    const int cil = hq * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < 5; ++t)
      if (t < ntap)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((h * 32 + acc_row(r, lane)) * 9 + tg * 5 + t) * 64 + cil] = acc[t][h][r];
  }
  __syncthreads();
  {
    const int KK = 9 * C;
    float* out = p.slabs + (size_t)split * p.Cout * KK + (size_t)(cb * 64) * KK + ib * 64;
#pragma unroll 2
    for (int k = 0; k < 18; ++k) {
      const int q4 = k * 512 + tid;
      const int co = q4 / 144, rem = q4 - co * 144;
      const f32x4 v = *reinterpret_cast<const f32x4*>(red + q4 * 4);
      *reinterpret_cast<f32x4*>(out + (size_t)co * KK + (rem >> 4) * C + (rem & 15) * 4) = v;
    }
  }
}

// ---- host side -----------------------------------------------------------------------------------
// PFR_WGRAD9 / pfr_set_tuning("wgrad9"): 0 off; 1 (default) the 56x56-class layers only (row slots of 64 positions); 2 every geometry
// the kernel takes.  Per launch the kernel wins everywhere (x1.13-1.7, profiles/r04_wgrad9.txt) and with the weight gradients serialised
// behind the main stream mode 2 is worth +1.0 % of the step; NEXT TO the main stream (the default: side stream) its workgroups — one per
// CU for the whole launch, all of the LDS, 8 x 228 VGPRs — keep the main stream's small dependent launches waiting for a CU, and the
// step LOSES 0.4-0.6 % with mode 2 (leaving 16-64 CUs unused gets that back, no more); at 56x56, where the tile kernel is slowest, the
// two effects cancel (+0.1 %).  Interleaved A/B in profiles/r04_wgrad9.txt.
static int wgrad9_mode() { return pfr_knob(KNOB_WGRAD9); }

// upper bound of the slabs a launch with this (Cout, KK) may write (the caller's workspace: pfr_conv2d_wgrad_splits)
int wgrad9_max_splits(int Cout, int KK) {
  if (KK % 576 || Cout % 64) return 0;     // (independent of wgrad9_mode(): workspace sizing must cover a later change of the mode)
  const int npairs = (Cout / 64) * (KK / 576);
  return npairs <= 256 ? 256 / npairs : 0;
}

// 0: geometry not taken (the tile kernel runs); otherwise the number of slabs written to `slabs`
int wgrad9_launch(const void* x, const void* dy, float* slabs, int N, int H, int W, int C, int Cout, int R, int S, int stride, int pad,
                  int lddy, hipStream_t st) {
  if (!wgrad9_mode() || R != 3 || S != 3 || stride != 1 || pad != 1 || C % 64 || Cout % 64 || lddy < Cout) return 0;
  // row slot: [0 | W pixels | 0 ...]; at W = 7 the slot is [0 | 7 pixels] and the right neighbour of the last pixel is the next slot's pad
  const int pw = W + 1 <= 8 ? 8 : W + 2 <= 16 ? 16 : W + 2 <= 32 ? 32 : W + 2 <= 64 ? 64 : 0;
  if (!pw || W < 4) return 0;
  if (wgrad9_mode() == 1 && pw != 64) return 0;   // (see wgrad9_mode)
  const int npairs = (Cout / 64) * (C / 64);
  if (npairs > 256) return 0;
  if ((long)N * H * W * lddy * 2 >= (1L << 31) || (long)N * H * W * C * 2 >= (1L << 31)) return 0;   // (row offsets + the pad lanes' offset stay below 2^32)
  const int sr = 128 / pw, nrows = N * (H + 1);   // extended rows: one all-zero row after every image
  // ("wgrad9_slots" < 256 leaves CUs to the main stream's kernels while this one runs on the side stream: VERDICT r4 item 2-ii)
  // (clamped to [npairs, 256]: the caller's slab workspace holds wgrad9_max_splits() = 256 / npairs slabs, whatever the knob says)
  int slots = pfr_knob(KNOB_WGRAD9_SLOTS);
  slots = slots < npairs ? npairs : (slots > 256 ? 256 : slots);
  int nsplit = slots / npairs;
  int rps = (nrows + nsplit - 1) / nsplit;
  rps = (rps + sr - 1) / sr * sr;
  nsplit = (nrows + rps - 1) / rps;
  Wg9Params p;
  p.x = x; p.dy = dy; p.slabs = slabs;
  p.H = H; p.W = W; p.C = C; p.Cout = Cout; p.lddy = lddy;
  p.nrows = nrows; p.rps = rps; p.nsplit = nsplit; p.ncb = Cout / 64; p.nib = C / 64;
  p.div_h = make_fastdiv((uint32_t)(H + 1));
  const dim3 grid((unsigned)(npairs * nsplit));
  constexpr int RED = 4 * 9 * 16 * 64 * 4;   // the parity hand-over at the end
#define PFR_W9_GO(PWV)                                                                                                       \
  {                                                                                                                          \
    constexpr int end = 3 * 16384 + 1024 + 3 * (128 / PWV + 2) * PWV * 128 + 1024;                                           \
    constexpr int lds = end > RED ? end : RED;                                                                               \
    static std::atomic<unsigned long long> attr{0};                                                                                                \
    PFR_MAX_LDS_ONCE(attr, lds, (const void*)wgrad9_kernel<PWV>); \
    hipLaunchKernelGGL((wgrad9_kernel<PWV>), grid, dim3(512), lds, st, p);                                                   \
  }
  if (pw == 8) PFR_W9_GO(8) else if (pw == 16) PFR_W9_GO(16) else if (pw == 32) PFR_W9_GO(32) else PFR_W9_GO(64)
#undef PFR_W9_GO
  if (hipGetLastError() != hipSuccess) return -1;
  return nsplit;
}
