// pfr_sconv3.hip — weight-stationary, HALO-staged 3x3 convolution on MFMA for the 64-channel layers (bf16).
//
// Replaces, for 3x3 / stride 1 / pad 1 / 64 -> 64 channels (torchvision Bottleneck.conv2 of layer1 at 56x56, BasicBlock convs of
// layer1; /root/reference/configs/dog_fe/fe_dogs_config.py:102-103), `nn.Conv2d.forward` and — with the tap-flipped,
// channel-transposed weights — its autograd input gradient.
//
// Why: the implicit-GEMM tile kernel gathers the activation rows of a tile once PER TAP (9 x 16 KB for a 128-row tile) and
// re-stages the 72 KB weight matrix for every tile: 220 KB of LDS-DMA per 32 KB of HBM traffic — the kernel is bound by the
// L2 -> LDS fill, at 0.24 of its HBM bound.  Here (the "LDS-staged input patches" of the north star):
//   * the whole weight matrix [64][3*3*64] stays in LDS for the life of a persistent workgroup (72 KB, XOR-swizzled);
//   * a wave computes a 4 x 8 pixel patch of one image x 64 output channels; its input is ONE halo tile of 6 x 10 pixels x 64
//     channels (7.5 KB instead of 9 x 4 KB), fetched by 8 LDS-DMA instructions into the wave's private 2-slot ring while the
//     previous patch is computed; the nine taps read their B fragments from that tile (row = (pr + r) * 10 + pc + s);
//     image borders are zeros out of the buffer descriptor's bounds check;
//   * no workgroup barrier in the loop; epilogue, BatchNorm statistics and the store-data hazard handling as pfr_sconv.hip:
//     a patch row is 8 pixels x 128 B = 1 KiB of contiguous output per store instruction.
// LDS bank layout of the halo tile: pixel (hr, hc) lies in row 10*hr + hc (128 B); its 16-byte channel chunk c is stored at
// chunk c ^ f(hr, hc), f = ((hr & 3) << 1) | ((hc >> 1) & 1).  A 16-lane group of a fragment read covers 4 consecutive hr and 4
// consecutive hc for every tap, so (row parity = hc & 1, f) takes 16 different values: conflict free for all nine taps.
// Accumulation order = the tile kernel's (taps row-major, channels ascending, 16 per MFMA): results are bit-identical.
#include "pfr_igemm.h"
#include <stdlib.h>
#ifndef PFR_S3_PD
#define PFR_S3_PD 3
#endif

struct Sconv3Params {
  const void* x;
  const void* w;        // [64][3][3][64]
  void* y;
  int N, H, W;          // C = Cout = 64
  float* stats_part;    // [nparts][2][64] (mean, M2) or nullptr
  const float* bias;    // INFER: per-cout bias of the BN-folded convolution
  int relu;             // INFER: ReLU after the bias
  int nblk, bpw;        // patches in total, patches per workgroup (contiguous in patch order)
  int tiles_x, tpi;     // W / 8, patches per image
  FastDiv div_tpi, div_tx;
#ifdef PFR_S3_TRACE
  unsigned long long* trace;   // [waves][4]: cycles waiting for the tile, in the 36 steps, total; patches
#endif
};
#ifdef PFR_S3_TRACE
static unsigned long long* g_s3_trace = nullptr;
extern "C" void pfr_debug_sconv3_trace(void* p) { g_s3_trace = (unsigned long long*)p; }
#endif

// INFER: y = relu?(result + bias) — the BN-folded inference plan; added to the bf16-rounded result in the read-back pass
// (one more bf16 rounding of the pre-activation than the tile kernel's fp32 epilogue, as in pfr_sconv.hip)
template <bool STATS, bool INFER = false>
__global__ __launch_bounds__(256, 1) void sconv3_kernel(Sconv3Params p) {
  static_assert(!(STATS && INFER), "the inference variant publishes no statistics");
  constexpr int WB = 64 * 1152;          // weight bytes
  constexpr int GB = 8192;               // ring slot: 64 pixel rows of 128 B (60 used)
  constexpr int GI = 8;                  // DMA instructions per halo tile
  constexpr int SB = 4;                  // store instructions per patch
  constexpr int NS = 2;
  extern __shared__ __attribute__((aligned(128))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float bias8[8];     // INFER: this lane's 8 couts in the read-back layout, fetched before any DMA is in flight
  if constexpr (INFER) {
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = p.bias[(lane & 7) * 8 + e];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const int blk_lo = blockIdx.x * p.bpw;
  const int blk_hi = blk_lo + p.bpw < p.nblk ? blk_lo + p.bpw : p.nblk;
  if (blk_lo >= p.nblk) return;
  const int nb = blk_hi - blk_lo;
  const int my_blocks = nb > wave ? (nb - wave + 3) / 4 : 0;

  __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, WB, 0x00020000);
  const int act_bytes = p.N * p.H * p.W * 128;
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, act_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, act_bytes, 0x00020000);
  const uint32_t OOBB = 0xF0000000u;

  // ---- weights -> LDS (once): row = cout (1152 B = 72 chunks), chunk c stored at c ^ ((row >> 1) & 7)
  for (int t = wave; t < WB / 1024; t += 4) {
    const int L = (t << 10) + (lane << 4);
    const int row = L / 1152, pc = (L - row * 1152) >> 4;
    const int lc = pc ^ ((row >> 1) & 7);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(smem + (t << 10)), 16, row * 1152 + (lc << 4), 0, 0, 0);
  }

  // ---- per-lane constants
  const int frow = lane & 31, fhalf = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t ring0 = lds0 + WB + wave * (NS * GB + 4096);
  char* const ringp = smem + WB + wave * (NS * GB + 4096);
  // A fragments (weights): row frow of tile i, chunk (2*kq + fhalf) ^ ((frow >> 1) & 7): the low three chunk bits take four
  // lane-dependent values (kq & 3), everything else is an immediate
  const int swz_w = (frow >> 1) & 7;
  uint32_t wsw[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wsw[q] = lds0 + (uint32_t)(frow * 1152 + ((((2 * q) | fhalf) ^ swz_w) << 4));
  // B fragments (halo tile): pixel (pr + r, pc + s), chunk (2*kk + fhalf) ^ f
  const int pr = frow >> 3, pc = frow & 7;
  uint32_t xa[3][4], xb[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xa[r][kk] = (uint32_t)((kk ^ ((pr + r) & 3)) << 5);
#pragma unroll
  for (int s = 0; s < 3; ++s) xb[s] = (uint32_t)((pr * 10 + pc) * 128 + ((fhalf ^ (((pc + s) >> 1) & 1)) << 4));
  // epilogue window (as pfr_sconv.hip): 32 pixel rows of 128 B, chunk ^ ((row >> 1) & 7)
  const int sx = (frow >> 1) & 7;
  const uint32_t ew_w = (uint32_t)(frow * 128 + (sx << 4) + fhalf * 8);
  const int e_row = lane >> 3, e_ch = lane & 7;
  const uint32_t ew_r = (uint32_t)(e_row * 128 + ((e_ch ^ (e_row >> 1)) << 4));

  // ---- loader: halo tile of patch `blk` (index within this workgroup's range)
  // DMA instruction t, lane l: LDS row rho = 8t + (l >> 3) = 10*hr + hc, stored chunk j = l & 7 holds channel chunk j ^ f(hr, hc)
  int hrv[GI], hcv[GI];
  uint32_t rel[GI];
#pragma unroll
  for (int t = 0; t < GI; ++t) {
    const int rho = 8 * t + (lane >> 3);
    const int hr = rho / 10, hc = rho - hr * 10;
    hrv[t] = rho < 60 ? hr : 1 << 20;
    hcv[t] = hc;
    const int f = ((hr & 3) << 1) | ((hc >> 1) & 1);
    rel[t] = (uint32_t)((hr * p.W + hc) * 128 + (((lane & 7) ^ f) << 4));
  }
  uint32_t goff[GI];
  int t_h0 = 0, t_w0 = 0, t_base = 0;
  bool t_ok = false;
  auto tile_head = [&](int blk) __attribute__((always_inline)) {
    const int bid = blk_lo + blk;
    const uint32_t n_img = fdiv((uint32_t)bid, p.div_tpi);
    const uint32_t rem = (uint32_t)bid - n_img * (uint32_t)p.tpi;
    const uint32_t ty = fdiv(rem, p.div_tx), tx = rem - ty * (uint32_t)p.tiles_x;
    t_h0 = (int)ty * 4 - 1;
    t_w0 = (int)tx * 8 - 1;
    t_base = (((int)n_img * p.H + t_h0) * p.W + t_w0) * 128;
    t_ok = blk < nb;
  };
  auto tile_part = [&](int t) __attribute__((always_inline)) {
    const bool ok = t_ok && (unsigned)(t_h0 + hrv[t]) < (unsigned)p.H && (unsigned)(t_w0 + hcv[t]) < (unsigned)p.W;
    goff[t] = ok ? (uint32_t)(t_base + (int)rel[t]) : OOBB;
  };
  auto issue_one = [&](int slot, int t) __attribute__((always_inline)) {
#ifdef PFR_S3_NODMA
    if (blockIdx.x < 100000) return;
#endif
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(ringp + slot * GB + t * 1024), 16, (int)goff[t], 0, 0, 0);
  };

  // prologue: the first patch's tile is requested now, the addresses of the second are ready for the first loop pass
  tile_head(wave);
#pragma unroll
  for (int t = 0; t < GI; ++t) { tile_part(t); issue_one(0, t); }
  tile_head(wave + 4);
#pragma unroll
  for (int t = 0; t < GI; ++t) tile_part(t);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GI) : "memory");   // this wave's share of the weights has landed
  __builtin_amdgcn_s_barrier();

  f32x2 s1[4], s2[4], ksh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { s1[e] = (f32x2){0.f, 0.f}; s2[e] = (f32x2){0.f, 0.f}; ksh[e] = (f32x2){0.f, 0.f}; }
  typedef __attribute__((address_space(3))) const u32x4* lds_cptr;
  typedef __attribute__((address_space(3))) u32x2* lds_w8ptr;
  const uint32_t ewin = ring0 + NS * GB;    // this wave's epilogue window (4 KB behind its ring)

  // ---- epilogue of one patch, cut into pieces that are issued in the shadow of the NEXT patch's MFMAs
  uint32_t e_ybase = 0;
  bool first = true;
  auto epi_head = [&](int blk) __attribute__((always_inline)) {
    const uint32_t bidu = (uint32_t)(blk_lo + blk);
    const uint32_t n_img = fdiv(bidu, p.div_tpi);
    const uint32_t rem = bidu - n_img * (uint32_t)p.tpi;
    const uint32_t ty = fdiv(rem, p.div_tx), tx = rem - ty * (uint32_t)p.tiles_x;
    e_ybase = ((n_img * (uint32_t)p.H + ty * 4) * (uint32_t)p.W + tx * 8) * 128;
  };
  auto epi_write = [&](const f32x16 (&a)[2], int c) __attribute__((always_inline)) {   // chunk c = ii*4 + qd of the window rows
    const int ii = c >> 2, qd = c & 3;
    bf16x4 v;
    v[0] = (bf16_t)a[ii][4 * qd];
    v[1] = (bf16_t)a[ii][4 * qd + 1];
    v[2] = (bf16_t)a[ii][4 * qd + 2];
    v[3] = (bf16_t)a[ii][4 * qd + 3];
    *(lds_w8ptr)(uintptr_t)(ewin + (ew_w ^ (uint32_t)(c << 4))) = __builtin_bit_cast(u32x2, v);
  };
  auto epi_shift = [&]() __attribute__((always_inline)) {
    if (STATS && first) {
      const u32x4 v = *(lds_cptr)(uintptr_t)(ewin + (uint32_t)(e_ch << 4));
#pragma unroll
      for (int e = 0; e < 4; ++e) ksh[e] = (f32x2){__uint_as_float(v[e] << 16), __uint_as_float(v[e] & 0xffff0000u)};
    }
    first = false;
  };
  auto epi_row = [&](int ps) __attribute__((always_inline)) {
    const u32x4 v = *(lds_cptr)(uintptr_t)(ewin + ((ew_r ^ (uint32_t)((ps & 1) << 6)) + ps * 1024));
    if constexpr (STATS) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 f = {__uint_as_float(v[e] << 16), __uint_as_float(v[e] & 0xffff0000u)};
        const f32x2 d = f - ksh[e];
        s1[e] += d;
        s2[e] = __builtin_elementwise_fma(d, d, s2[e]);
      }
    }
    // patch row ps: 8 pixels x 128 B contiguous
#ifdef PFR_S3_NOSTORE
    if (v[0] != 0x12345u) return;
#endif
    // store data registers are read late under a deep vector-memory queue and hipcc re-uses them at once: store + EXP_CNT wait
    // in one asm statement (buffer_store_b128_sync, pfr_mma.h)
    if constexpr (INFER) {
      float f[8];
      Chunk<bf16_t>::unpack(v, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        f[e] += bias8[e];
        if (p.relu) f[e] = fmaxf(f[e], 0.f);
      }
      buffer_store_b128_sync(Chunk<bf16_t>::pack(f), yrsrc, (uint32_t)(lane << 4), e_ybase + (uint32_t)(ps * p.W * 128));
    } else {
      buffer_store_b128_sync(v, yrsrc, (uint32_t)(lane << 4), e_ybase + (uint32_t)(ps * p.W * 128));
    }
  };

  // ---- one patch: 36 k16 steps into `cur`; in their shadow the DMA of the next tile (steps 0-7), the addresses of the one after
  //      (8-16) and the epilogue of the previous patch `prv` (17-35)
  // The fragments of step q + PD are requested before the MFMAs of step q are issued (one wave per SIMD: nothing else hides the
  // LDS latency; left to itself hipcc sinks every read to its use: 7600 cycles per patch for 2304 cycles of MFMA).
  constexpr int PD = PFR_S3_PD;
  int slot = 0;
  bool stores_pending = false;
#ifdef PFR_S3_TRACE
  unsigned long long tr_wait = 0, tr_loop = 0;
#endif
  auto patch = [&](f32x16 (&cur)[2], const f32x16 (&prv)[2], int bi, bool has_prev) __attribute__((always_inline)) {
    const int blk = wave + 4 * bi;
    // this patch's tile has landed when only the stores of the last epilogue (issued after it) are outstanding
#ifdef PFR_S3_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
#endif
    if (stores_pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PFR_S3_TRACE
    const unsigned long long tr1 = __builtin_amdgcn_s_memtime();
#endif
    const uint32_t sbase = ring0 + slot * GB;
    uint32_t xs[3];
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) xs[s_] = sbase + xb[s_];
    u32x4 fq[PD + 2], fp0[PD + 2], fp1[PD + 2];
    auto rd = [&](int kq, int b) __attribute__((always_inline)) {
      const int tap = kq >> 2, kk = kq & 3, r = tap / 3, s_ = tap - r * 3;
      fq[b] = *(lds_cptr)(uintptr_t)(xs[s_] + xa[r][kk] + (uint32_t)((r * 10 + s_) * 128));
      const uint32_t wa = wsw[kq & 3] + (uint32_t)(((2 * kq) & ~7) << 4);
      fp0[b] = *(lds_cptr)(uintptr_t)(wa);
      fp1[b] = *(lds_cptr)(uintptr_t)(wa + 32 * 1152);
    };
#pragma unroll
    for (int q = 0; q < PD; ++q) rd(q, q);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kq = 0; kq < 36; ++kq) {
      const int b = kq % (PD + 2);
      // Issue order inside a step (one wave per SIMD issues in order, and an MFMA waits at ISSUE for the matrix pipe: two MFMAs
      // back to back block the wave for 28 of every 64 cycles): MFMA | the three fragment reads of step kq + PD | MFMA | shadow work,
      // so that each MFMA's 32 pipe cycles cover ~7 issue slots of other work.
      const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (kq == 0) cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp0[b]), __builtin_bit_cast(bf16x8, fq[b]), z, 0, 0, 0);
      else cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp0[b]), __builtin_bit_cast(bf16x8, fq[b]), cur[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // (the slot of step kq's fragments is not re-used before step kq + 1 issues its reads: PD + 2 buffers)
      if (kq + PD < 36) rd(kq + PD, (kq + PD) % (PD + 2));
      __builtin_amdgcn_sched_barrier(0);
      if (kq == 0) cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp1[b]), __builtin_bit_cast(bf16x8, fq[b]), z, 0, 0, 0);
      else cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp1[b]), __builtin_bit_cast(bf16x8, fq[b]), cur[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- shadow work of this step
      if (kq < 8) issue_one(slot ^ 1, kq);                 // next patch's tile (the slot's last reader was the previous patch)
      else if (kq == 8) tile_head(blk + 8);                // addresses of the patch after the next
      else if (kq < 17) tile_part(kq - 9);
      else if (has_prev) {
        if (kq == 17) epi_head(blk - 4);
        else if (kq < 26) epi_write(prv, kq - 18);
        else if (kq == 26) epi_shift();
        else if (kq < 31) epi_row(kq - 27);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // MFMA -> VALU read-after-write distance is software-managed; explicit wait states tied to the accumulators (pfr_sconv.hip)
    asm volatile("s_nop 15" : "+v"(cur[0]));
    asm volatile("s_nop 15" : "+v"(cur[1]));
#ifdef PFR_S3_TRACE
    { const unsigned long long tr2 = __builtin_amdgcn_s_memtime(); tr_wait += tr1 - tr0; tr_loop += tr2 - tr1; }
#endif
    stores_pending = has_prev;
    slot ^= 1;
  };

#ifdef PFR_S3_TRACE
  const unsigned long long tr_begin = __builtin_amdgcn_s_memtime();
#endif
  f32x16 accA[2], accB[2];
#pragma unroll 1
  for (int bi = 0; bi < my_blocks; bi += 2) {
    patch(accA, accB, bi, bi > 0);
    if (bi + 1 < my_blocks) patch(accB, accA, bi + 1, true);
  }
  // the last patch's epilogue
  if (my_blocks > 0) {
    epi_head(wave + 4 * (my_blocks - 1));
    if (my_blocks & 1) {
#pragma unroll
      for (int c = 0; c < 8; ++c) epi_write(accA, c);
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) epi_write(accB, c);
    }
    epi_shift();
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) epi_row(ps);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PFR_S3_TRACE
  if (p.trace && lane == 0) {
    unsigned long long* t = p.trace + ((size_t)blockIdx.x * 4 + wave) * 4;
    t[0] = tr_wait; t[1] = tr_loop; t[2] = __builtin_amdgcn_s_memtime() - tr_begin; t[3] = (unsigned long long)my_blocks;
  }
#endif
  if (STATS) {
    __syncthreads();
    const float nval = (float)(my_blocks * 32);
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][3][64]
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float a = s1[e][h], q = s2[e][h];
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          q += __shfl_xor(q, o, 64);
        }
        if (lane < 8) {
          const int c = lane * 8 + e * 2 + h;
          const float k = ksh[e][h];
          red[(wave * 3 + 0) * 64 + c] = nval > 0.f ? k + a / nval : 0.f;
          red[(wave * 3 + 1) * 64 + c] = nval > 0.f ? q - a * a / nval : 0.f;
          red[(wave * 3 + 2) * 64 + c] = nval;
        }
      }
    __syncthreads();
    if (tid < 64) {
      float n = 0.f, a = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float nw = red[(w * 3 + 2) * 64 + tid];
        n += nw;
        a = fmaf(nw, red[(w * 3 + 0) * 64 + tid], a);
      }
      const float mean = n > 0.f ? a / n : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float nw = red[(w * 3 + 2) * 64 + tid];
        const float d = red[(w * 3 + 0) * 64 + tid] - mean;
        m2 += red[(w * 3 + 1) * 64 + tid] + nw * d * d;
      }
      p.stats_part[((size_t)blockIdx.x * 2 + 0) * 64 + tid] = mean;
      p.stats_part[((size_t)blockIdx.x * 2 + 1) * 64 + tid] = m2;
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
// geometry-only eligibility: 3x3, stride 1, pad 1, 64 -> 64 channels, bf16, W % 8 == 0, H % 4 == 0 (PFR_SCONV / "sconv" tuning: 0 off)
// pfr_set_tuning("sconv3"): 0 keeps the tile kernel for the 3x3 layers
bool sconv3_geom(int N, int H, int W, int C, int Cout, int R, int S, int stride, int pad, int idil_log2, int OH, int OW, int dtype,
                 int out_dtype, int* bpw) {
  if (!pfr_knob(KNOB_SCONV3)) return false;
  if (sconv_mode() == 0 || dtype != PFR_BF16 || out_dtype != PFR_BF16) return false;
  if (R != 3 || S != 3 || stride != 1 || pad != 1 || idil_log2 != 0 || C != 64 || Cout != 64) return false;
  if (OH != H || OW != W || (W & 7) || (H & 3) || (long)N * H * W * 128 >= ((long)1 << 31)) return false;
  const int nblk = N * (H / 4) * (W / 8);
  if (sconv_mode() == 1 && nblk < 256 * 8) return false;   // too few patches per workgroup for a pipeline
  if (bpw) *bpw = (nblk + 255) / 256;
  return true;
}

int sconv3_try_launch(IgemmParams& p, int dtype, int out_dtype, hipStream_t st) {
  if (p.ldy != p.Cout || p.accumulate || p.pro_scale || p.act || p.bnb_part[0] || p.residual) return 1;
  const bool infer = p.bias != nullptr;      // inference form: bias (+ ReLU), no statistics; training form: neither
  if (infer ? p.stats_part != nullptr : p.out_relu != 0) return 1;
  int bpw;
  if (!sconv3_geom(p.N, p.H, p.W, p.C, p.Cout, p.R, p.S, p.ostride, p.pad, p.idil_log2, p.OH, p.OW, dtype, out_dtype, &bpw)) return 1;
  if (p.stats_part && p.want_mtile && p.want_mtile != bpw * 32) return 1;
  Sconv3Params sp;
  sp.x = p.x; sp.w = p.w; sp.y = p.y;
  sp.N = p.N; sp.H = p.H; sp.W = p.W;
  sp.stats_part = p.stats_part;
  sp.bias = p.bias; sp.relu = p.out_relu;
  sp.tiles_x = p.W / 8;
  sp.tpi = (p.H / 4) * sp.tiles_x;
  sp.nblk = p.N * sp.tpi;
  sp.bpw = bpw;
  sp.div_tpi = make_fastdiv((uint32_t)sp.tpi);
  sp.div_tx = make_fastdiv((uint32_t)sp.tiles_x);
#ifdef PFR_S3_TRACE
  sp.trace = g_s3_trace;
#endif
  const int lds = 64 * 1152 + 4 * (2 * 8192 + 4096);
  const dim3 grid((unsigned)((sp.nblk + bpw - 1) / bpw)), block(256);
  static std::atomic<unsigned long long> attr_set{0};
  PFR_MAX_LDS_ONCE(attr_set, 160 * 1024, (const void*)sconv3_kernel<true>, (const void*)sconv3_kernel<false>, (const void*)sconv3_kernel<false, true>);
  if (infer) hipLaunchKernelGGL((sconv3_kernel<false, true>), grid, block, lds, st, sp);
  else if (p.stats_part) hipLaunchKernelGGL(sconv3_kernel<true>, grid, block, lds, st, sp);
  else hipLaunchKernelGGL(sconv3_kernel<false>, grid, block, lds, st, sp);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
