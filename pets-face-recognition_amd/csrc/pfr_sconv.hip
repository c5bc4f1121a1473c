// pfr_sconv.hip — weight-stationary STREAMING 1x1 convolution / GEMM on MFMA for the HBM-bound layers (bf16).
//
// Replaces, for the geometries it takes, the same reference calls as pfr_igemm.hip: `nn.Conv2d(k=1)` forward and its autograd
// input gradient inside torchvision's Bottleneck (/root/reference/configs/dog_fe/fe_dogs_config.py:102-103), incl. the residual
// join of the block's first conv (`out += identity; relu` backward).  y[m][n] = sum_k x[m][k] * w[n][k]  (+ res[m][n] where mask).
//
// Why another kernel: at batch 256 the 1x1 convolutions of layer1-3 move 130-980 MB for 20-60 us of MFMA work — their ceiling is
// HBM.  The tile kernels reach 3.3-4.5 TB/s on them: every 128-row tile re-stages its weight panel through LDS (a third to a half
// of the LDS-DMA volume), keeps at most one k-step (12-16 KB) of loads in flight behind a workgroup barrier, and pays a
// prologue/epilogue per tile.  tools/probe/stream_probe.hip shows what the memory system gives a PERSISTENT workgroup whose waves
// each own a private LDS-DMA ring: 6.3-6.5 TB/s read-only, 5.6-5.9 TB/s with an output stream — this kernel is that skeleton
// with the MFMAs and the epilogue hung into it:
//   * grid = npanels x nranges = 256 persistent workgroups of 8 waves (two per SIMD, <= 256 registers each; a first version with
//     4 waves of 512 registers was instruction-issue bound: one wave per SIMD issues one instruction per ~4 cycles);
//   * the weight panel [<= 256 couts][K] is loaded into LDS ONCE per workgroup (<= 64 KiB, XOR-swizzled 16-byte chunks);
//   * a workgroup owns a set of 32-row blocks (block-interleaved over the workgroups, or a contiguous range when the block count
//     does not divide); a wave takes a 64- or 128-cout slice of the panel and walks its blocks.  The rows of a block arrive as K/64
//     "granules" ([32 rows][64 k] = 4 KiB, full 128-byte lines) through the wave's PRIVATE ring of NS slots with counted
//     `s_waitcnt vmcnt(n)` — no workgroup barrier anywhere in the loop, waves drift freely, so one wave's MFMAs overlap another's
//     epilogue and a third's DMA issue;
//   * epilogue per wave: accumulators -> bf16 -> the just-consumed ring slot as 4 KiB staging window -> whole 128-byte row
//     segments -> buffer stores; BatchNorm statistics (shifted sums of the values AS STORED) stay in registers over all blocks
//     of the wave and leave as ONE (mean, M2) partial per workgroup: pfr_bn_finalize sees <= 256 partials (one launch).
// Results are bit-identical to igemm_kernel's (same k order inside and across MFMAs); statistics partials differ only in how the
// rows are grouped.
#include "pfr_igemm.h"
// Non-temporal hint on the epilogue operand rows: the residual-gradient rows of the data-gradient joins (their last use) and the
// BatchNorm-input rows of the launches that leave the BatchNorm-backward sums (next read 100+ MB of traffic later).  Same-box A/B of the
// ResNet-50 step, six interleaved pairs (tools/lib_ab.sh): +0.18 % (+2 … +71 img/s, never negative).  -DPFR_SCONV_NT_OFF: plain loads.
#ifdef PFR_SCONV_NT_OFF
#define PFR_SCONV_RES_NT ""
#define PFR_SCONV_BNX_NT ""
#else
#define PFR_SCONV_RES_NT " nt"
#define PFR_SCONV_BNX_NT " nt"
#endif
#include <stdlib.h>
#include <string.h>

struct SconvParams {
  const void* x;
  const void* w;
  void* y;
  int M, K, N;            // output rows, reduction length (= input channels), output channels (ldy == N)
  int H, W, OH, OW, ostride;   // ostride > 1: row m = (n, oh, ow) reads input pixel (n, oh*ostride, ow*ostride)
  float* stats_part;      // [nranges][2][N] (mean, M2) or nullptr
  const void* res;        // join: residual-branch gradient [M][N] bf16, or nullptr
  const unsigned char* res_mask;   // join: its ReLU bit mask [M][N/8]
  int res_sub;            // join: `res` is COMPACT [n][OH/2][OW/2][N] and adds to the pixels with even (oh, ow) only (the data
                          // gradient of a 1x1 / stride-2 projection shortcut, computed densely on its own grid); others add 0
  const float* bias;      // inference epilogue (EP 2 / 3): per-cout bias of the BN-folded convolution
  int relu;               // inference epilogue: ReLU after bias (+ residual)
  // EP 4 / 5: BatchNorm-backward sums of the BN whose output gradient y is (pfr_conv2d_dgrad_bn): x of that BN [M][N], its
  // coefficient rows [4][N] (mean, invstd, scale, shift), its ReLU bit mask [M][N/8] or nullptr (then scale*x + shift > 0),
  // partial sums [nranges][2][N] = (sum g*mask, sum g*mask*xhat), g = the value stored to y
  const void* bnx;
  const float* bn_coef;
  const unsigned char* bn_mask;
  float* bn_part;
  // EP 6: a second BN consuming the same gradient through the same bit mask (projection shortcut of the previous block)
  const void* bnx2;
  const float* bn_coef2;
  float* bn_part2;
  // EP 10 / 11 (block tail): y = relu(a1*conv + b1 + (EP 10: res | EP 11: a2*res + b2)) on the bf16-rounded convolution value, and the
  // sign of the pre-ReLU value as one bit per element (mask_out [M][N/8]) — pfr_bn_act_mask's arithmetic in this epilogue
  const float* tail_a1; const float* tail_b1; const float* tail_a2; const float* tail_b2;
  unsigned char* mask_out;
  // EP 12 (two-source data gradient): the reduction runs over [x | x2] — x [M][K - K2] followed by x2 [M][K2] — against weight rows of
  // K columns, and the accumulators start from cbias [N] instead of zero: dZ = G·(A∘W) + Z·S + bias of pfr_bnfree.hip in ONE launch
  const void* x2;
  int K2, x2bytes;
  const float* cbias;
  int store_masked;       // BNB with a bit mask: y is stored THROUGH the mask (g*mask; every consumer of a block-output gradient
                          // reads it through that mask anyway, so they may then skip it: pfr_conv2d_dgrad_bn_ex)
  int npanels, nranges, R;    // R: rows per range (multiple of the 32-row block height)
  int npw;                    // couts of a workgroup's weight panel (64, 128 or 256)
  int interleave;             // 1: block-interleaved row assignment (needs M % 32 == 0 and M / 32 divisible by nranges)
  int xbytes;                 // bytes of x (buffer bound: rows past the end read zeros)
  FastDiv div_ohow, div_ow;
};

// TP: 32-cout accumulator tiles per wave (the wave's column slice is TP*32 couts), NS: ring slots per wave, STATS: BatchNorm partials
// EP (epilogue): 0 plain; 1 JOIN: y = result + (mask bit ? res : 0) — the residual join of a block's first data gradient
// (pfr_conv2d_dgrad_join); 2: y = relu?(result + bias); 3: y = relu?(result + bias + res) — the BN-folded inference convolutions
// (Controller.validation_step / generate_tsv embedder).  EP 2 / 3 add to the bf16-rounded result in the read-back pass (the window
// holds bf16), i.e. one more bf16 rounding of the pre-activation than the tile kernel's fp32 epilogue.
// EP 4: plain data gradient + the BatchNorm-backward sums of the BN it feeds (mask recomputed or bit mask); EP 5: the join + those
// sums (bit mask) — pfr_bn_bwd_reduce's pass over (gradient, BN input, mask) becomes one extra row read in this epilogue.
// EP 7 / 8 = EP 5 / 6 WITHOUT the first BN's input: only sum g*mask is produced for it (row 1 of its partials = 0) — the
// BN-input-free backward of pfr_bnfree.hip gets sum g*mask*xhat from the weight-gradient GEMM instead; no x row read, no coefficients.
// EP 9 (with STATS): statistics ONLY — the convolution is computed, rounded to bf16 and enters the BatchNorm partials exactly as in the
// plain variant, but nothing is stored (first pass of the recompute form of a bottleneck's last convolution, pfr_conv1x1_stats).
// EP 10 / 11: the second pass (pfr_conv1x1_bn_tail): the convolution again, then the block tail relu(bn3(.) + shortcut) and its ReLU
// bit mask in the epilogue — the convolution output itself never reaches HBM.
// EP 12 = EP 4 (data gradient + BatchNorm-backward sums) over TWO row sources with an fp32 bias in the accumulators (see SconvParams).
template <int TP, int NS, bool STATS, int EP = 0>
__global__ __launch_bounds__(512, 2) void sconv_kernel(SconvParams p) {
  constexpr bool JOIN = EP == 1 || EP == 5 || EP == 6 || EP == 7 || EP == 8;
  constexpr bool HASRES = EP == 1 || EP == 3 || EP == 5 || EP == 6 || EP == 7 || EP == 8 || EP == 10 || EP == 11;
  constexpr bool BNB = EP == 4 || EP == 5 || EP == 6 || EP == 7 || EP == 8 || EP == 12;
  constexpr bool SRC2 = EP == 12;
  constexpr bool BNB2 = EP == 6 || EP == 8;     // + the projection-shortcut BN of the previous block (same g, same mask, its own x)
  constexpr bool NOX = EP == 7 || EP == 8;      // the first BN's input is not read (bit mask required)
  constexpr bool INFER = EP == 2 || EP == 3;
  constexpr bool NOSTORE = EP == 9;
  constexpr bool TAIL = EP == 10 || EP == 11;
  static_assert(!(STATS && EP != 0 && EP != 9) && !(EP == 9 && !STATS), "only the plain / statistics-only variants publish statistics");
  constexpr int NPV = TP * 32;           // couts per wave
  constexpr int GB = 4096;               // granule bytes: [32 rows][64 k] bf16
  constexpr int GI = 4;                  // DMA instructions per granule
  constexpr int NCG = NPV / 64;          // 64-cout column groups of the epilogue
  constexpr int SB = NPV / 16;           // data store instructions per block (32 rows x NPV couts x 2 B / 1 KiB)
  constexpr int SBT = NOSTORE ? 0 : (TAIL ? 2 * SB : SB);   // vector-memory STORE instructions a block's epilogue really issues (+ mask bytes)
  // residual (+ mask) and BN-input (+ mask) load instructions per block (BNB always issues its mask-byte load: with a recomputed
  // mask it reads one dummy byte, the count per block stays a compile-time constant)
  constexpr int RL = (JOIN ? 2 * SB : (HASRES ? SB : 0)) + (BNB ? (NOX ? SB : 2 * SB) : 0) + (BNB2 ? SB : 0);
  extern __shared__ __attribute__((aligned(128))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.K, RB = K * 2;                 // weight row bytes
  const int KG = K >> 6;                         // granules per block
  const int KG1 = SRC2 ? (K - p.K2) >> 6 : KG;   // ... of which the first KG1 come from x, the rest from x2
  const bool tail8 = SRC2 && p.K2 == 64;         // a 64-wide x2 part: its 8 weight chunks per row swizzle in a group of 8 (as K == 64)
  const int NPW = p.npw;                         // couts of the workgroup's weight panel
  const int wbytes = NPW * RB;
  const int nsub = NPW / NPV;                    // column slices per panel (1, 2 or 4): wave -> (slice, row lane)
  const int sub = wave % nsub, wrow = wave / nsub, nrw = 8 / nsub;

  // (panel, range) of this workgroup: the panels of one range run on the same XCD (block b is observed on XCD b % 8 — speed only)
  const int b = blockIdx.x;
  const int xcd = b & 7, q8 = b >> 3;
  const int pn = q8 % p.npanels, rq = q8 / p.npanels;
  const int r = rq * 8 + xcd;
  const int n0 = pn * NPW + sub * NPV;           // first cout of this wave
  // rows of this workgroup: the contiguous range [r*R, (r+1)*R), or — interleaved mode, every block whole — the 32-row blocks
  // r, r + nranges, r + 2*nranges, ... (chip-wide the workgroups then read ONE moving window of the tensor instead of 256 streams:
  // +5-8 % HBM throughput, tools/probe/stream_probe.hip); block b of the workgroup starts at row blk_row(b)
  const int il = p.interleave;
  const int row_lo = il ? r * 32 : r * p.R;
  if (r >= p.nranges || row_lo >= p.M) return;
  const int row_hi = il ? p.M : (row_lo + p.R < p.M ? row_lo + p.R : p.M);
  const int bstep = il ? p.nranges * 32 : 32;     // row distance of consecutive blocks of this workgroup
  const int nblk = il ? (p.M / 32 - r + p.nranges - 1) / p.nranges : (row_hi - row_lo + 31) >> 5;

  __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.N * RB, 0x00020000);
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.xbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t x2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(SRC2 ? p.x2 : p.x), 0, SRC2 ? p.x2bytes : 16, 0x00020000);
  __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.M * p.N * 2, 0x00020000);
  const uint32_t OOBB = 0xF0000000u;
  __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.y), 0,
                                                                   p.res_sub ? (p.M >> 2) * p.N * 2 : p.M * p.N * 2, 0x00020000);
  // (join without a mask = plain add of `res`, e.g. a gradient already accumulated in y itself: the mask-byte load then reads one
  //  dummy byte so that the per-block load count stays a compile-time constant)
  const bool has_rmask = p.res_mask != nullptr;
  __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(has_rmask ? p.res_mask : (const unsigned char*)p.y), 0,
                                                                   has_rmask ? p.M * (p.N >> 3) : 16, 0x00020000);

  // inference epilogue: this lane's 8 couts of every column group (read-back layout), fetched before any DMA is in flight
  float bias8[NCG][8];
  if constexpr (INFER) {
    const int e_ch0 = lane & 7;
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) bias8[g][e] = p.bias[n0 + g * 64 + e_ch0 * 8 + e];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float ta1[NCG][8], tb1[NCG][8], ta2[NCG][8], tb2[NCG][8];
  __amdgpu_buffer_rsrc_t korsrc = __builtin_amdgcn_make_buffer_rsrc(TAIL ? (void*)p.mask_out : p.y, 0, TAIL ? p.M * (p.N >> 3) : 16, 0x00020000);
  if constexpr (TAIL) {
    const int e_ch0 = lane & 7;
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = n0 + g * 64 + e_ch0 * 8 + e;
        ta1[g][e] = p.tail_a1[c]; tb1[g][e] = p.tail_b1[c];
        ta2[g][e] = EP == 11 ? p.tail_a2[c] : 1.f; tb2[g][e] = EP == 11 ? p.tail_b2[c] : 0.f;
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // BN-backward coefficients of this lane's 8 couts per column group (read-back layout): xhat = ca*x + cb, mask = sc*x + sh > 0
  float bca[NCG][8], bcb[NCG][8], bsc[NCG][8], bsh[NCG][8];
  f32x2 b1[NCG][4], b2[NCG][4];
  float cca[NCG][8], ccb[NCG][8];   // second BN: xhat coefficients and its sum g*mask*xhat
  f32x2 c2[NCG][4];
  __amdgpu_buffer_rsrc_t cxrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(BNB2 ? p.bnx2 : p.y), 0, p.M * p.N * 2, 0x00020000);
  __amdgpu_buffer_rsrc_t bxrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(BNB ? p.bnx : p.y), 0, p.M * p.N * 2, 0x00020000);
  __amdgpu_buffer_rsrc_t bmrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(BNB && p.bn_mask ? p.bn_mask : (const unsigned char*)p.y), 0, BNB && p.bn_mask ? p.M * (p.N >> 3) : 16, 0x00020000);
  const bool bn_bits = BNB && p.bn_mask != nullptr;
  if constexpr (BNB) {
    const int e_ch0 = lane & 7;
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = n0 + g * 64 + e_ch0 * 8 + e;
        if constexpr (NOX) {
          bca[g][e] = 0.f; bcb[g][e] = 0.f; bsc[g][e] = 0.f; bsh[g][e] = 1.f;
        } else {
          const float mu = p.bn_coef[c], is = p.bn_coef[p.N + c];
          bca[g][e] = is;
          bcb[g][e] = -mu * is;
          bsc[g][e] = bn_bits ? 0.f : p.bn_coef[2 * p.N + c];
          bsh[g][e] = bn_bits ? 1.f : p.bn_coef[3 * p.N + c];
        }
      }
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) { b1[g][e] = (f32x2){0.f, 0.f}; b2[g][e] = (f32x2){0.f, 0.f}; c2[g][e] = (f32x2){0.f, 0.f}; }
    if constexpr (BNB2) {
#pragma unroll
      for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = n0 + g * 64 + e_ch0 * 8 + e;
          const float mu = p.bn_coef2[c], is = p.bn_coef2[p.N + c];
          cca[g][e] = is;
          ccb[g][e] = -mu * is;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- weight panel -> LDS (once): linear LDS image, XOR swizzle on the source side
  {
    const int ninst = wbytes >> 10;
    for (int t = wave; t < ninst; t += 8) {
      const int L = (t << 10) + (lane << 4);
      const int row = L / RB, pc = (L - row * RB) >> 4;
      const int lc = (K == 64 || (tail8 && pc >= KG1 * 8)) ? (pc ^ ((row >> 1) & 7)) : (pc ^ (row & 15));
      const uint32_t off = (uint32_t)((pn * NPW + row) * RB + (lc << 4));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(smem + (t << 10)), 16, (int)off, 0, 0, 0);
    }
  }

  // accumulator-initial bias (EP 12): acc[i][4*qd + j] belongs to cout i*32 + 8*qd + 4*(lane >> 5) + j
  f32x16 cb[TP];
  if constexpr (SRC2) {
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
#pragma unroll
        for (int j = 0; j < 4; ++j) cb[i][4 * qd + j] = p.cbias[n0 + i * 32 + 8 * qd + 4 * (lane >> 5) + j];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- per-lane constants (LDS byte addresses are 32-bit)
  const int frow = lane & 31, fhalf = lane >> 5;
  const int sx = (frow >> 1) & 7;                                     // swizzle of a granule / window row (128-byte rows)
  const int sw = (K == 64) ? ((frow >> 1) & 7) : (frow & 15);         // swizzle of a weight row
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t ring0 = lds0 + wbytes + wave * (NS * GB);            // this wave's ring (wave-uniform)
  char* const ringp = smem + wbytes + wave * (NS * GB);
  // fragment read offsets inside a granule for k16 step s: row frow, chunk (2s + fhalf) ^ sx
  uint32_t xo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) xo[s] = (uint32_t)(frow * 128 + (((2 * s + fhalf) ^ sx) << 4));
  const uint32_t wlane = lds0 + (uint32_t)((sub * NPV + frow) * RB);  // weight row of this lane (tile i adds i*32*RB)
  // epilogue window (the ring slot the block's last granule came through): write side (row frow) / read-back side
  const uint32_t ew_w = (uint32_t)(frow * 128 + (sx << 4) + fhalf * 8);          // ^ (c << 4) for chunk c
  const int e_row = lane >> 3, e_ch = lane & 7;
  const uint32_t ew_r = (uint32_t)(e_row * 128 + ((e_ch ^ (e_row >> 1)) << 4)); // pass ps: + ps*1024, ^ ((ps & 1) << 6)
  const uint32_t y_lane = (uint32_t)((e_row * p.N + e_ch * 8) * 2);              // lane part of the output byte offset
  const uint32_t k_lane = (uint32_t)(e_row * (p.N >> 3) + e_ch);                 // lane part of the mask byte offset

  // ---- loader state
  const int g_row = lane >> 3, g_pc = lane & 7;                       // DMA: row within an 8-row instruction, physical chunk
  int l_blk = wrow, l_kc = 0;
  uint32_t rowoff[GI], rowoff2[GI];
  auto set_rows = [&](int blk) {
#pragma unroll
    for (int t = 0; t < GI; ++t) {
      const int m = row_lo + blk * bstep + t * 8 + g_row;
      uint32_t off = OOBB;
      if constexpr (SRC2)
        rowoff2[t] = blk < nblk ? (uint32_t)m * (uint32_t)(p.K2 * 2) + (uint32_t)((g_pc ^ ((t * 4 + (g_row >> 1)) & 7)) << 4) : OOBB;
      if (blk < nblk) {
        uint32_t xr = (uint32_t)m;
        if (p.ostride != 1) {
          const uint32_t n_img = fdiv((uint32_t)m, p.div_ohow);
          const uint32_t rem = (uint32_t)m - n_img * (uint32_t)(p.OH * p.OW);
          const uint32_t oh = fdiv(rem, p.div_ow), ow = rem - oh * p.OW;
          xr = (n_img * p.H + oh * p.ostride) * p.W + ow * p.ostride;
        }
        const int lc = g_pc ^ ((t * 4 + (g_row >> 1)) & 7);
        off = xr * (uint32_t)(SRC2 ? (K - p.K2) * 2 : RB) + (uint32_t)(lc << 4);     // rows past the end lie beyond num_records: zero-filled
      }
      rowoff[t] = off;
    }
  };
  set_rows(l_blk);
  auto issue = [&](int slot) {
    if (SRC2 && l_kc >= KG1) {
#pragma unroll
      for (int t = 0; t < GI; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x2rsrc, (__attribute__((address_space(3))) void*)(ringp + slot * GB + t * 1024), 16,
                                                 (int)(rowoff2[t] + (uint32_t)((l_kc - KG1) << 7)), 0, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < GI; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(ringp + slot * GB + t * 1024), 16,
                                                 (int)(rowoff[t] + (uint32_t)(l_kc << 7)), 0, 0, 0);
    }
    if (++l_kc == KG) {
      l_kc = 0;
      l_blk += nrw;
      set_rows(l_blk);
    }
  };

  // ---- ring prologue: NS-1 granules in flight; the weight panel is complete behind them
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * GI) : "memory");
  __builtin_amdgcn_s_barrier();

  // ---- statistics state (per lane: 8 couts of every column group, fixed over all rows)
  f32x2 s1[NCG][4], s2[NCG][4], ksh[NCG][4];
#pragma unroll
  for (int g = 0; g < NCG; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[g][e] = (f32x2){0.f, 0.f}; s2[g][e] = (f32x2){0.f, 0.f}; ksh[g][e] = (f32x2){0.f, 0.f}; }

  int slot = 0;
  uint32_t hist = 0;   // bit i: iteration (now - 1 - i) ended with an epilogue (its SB stores are younger than older DMAs)
  const int my_blocks = nblk > wrow ? (nblk - wrow + nrw - 1) / nrw : 0;

  auto wait_granule = [&]() {
    // the granule to consume has landed when only the NS-1 younger granules and the stores of the epilogues of the last
    // NS-1 iterations are outstanding (gfx950 retires a wave's loads and stores in issue order)
#ifdef PFR_SCONV_SAFEWAIT
    const int ne = 0;
#else
    const int ne = __builtin_popcount(hist & ((1u << (NS - 1)) - 1));
#endif
    // (join: the RL residual / mask loads of a block are issued at its start, which follows an epilogue: every epilogue in the
    //  window stands for SB stores + RL loads younger than the granule.  The first block's loads follow no epilogue and go
    //  uncounted: a smaller count than the true one only waits longer.)
    constexpr int EV = SBT + RL;
    if (ne == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * GI) : "memory");
    else if (ne == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * GI + EV > 63 ? 63 : (NS - 1) * GI + EV) : "memory");
    else if (ne == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * GI + 2 * EV > 63 ? 63 : (NS - 1) * GI + 2 * EV) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * GI + 3 * EV > 63 ? 63 : (NS - 1) * GI + 3 * EV) : "memory");
    hist <<= 1;
  };
  typedef __attribute__((address_space(3))) const u32x4* lds_cptr;
  typedef __attribute__((address_space(3))) u32x2* lds_w8ptr;

#pragma unroll 1
  for (int bi = 0; bi < my_blocks; ++bi) {
    const int m0 = row_lo + (wrow + nrw * bi) * bstep;
    f32x16 acc[TP];
    // join: the block's residual rows and mask bytes, requested now and consumed by the epilogue.  Inline asm: hipcc would wait
    // vmcnt(0) — draining the DMA ring — for any load it knows of; these are counted by hand (wait_granule, epilogue).
    u32x4 rres[NCG][4];
    uint32_t rmk[NCG][4];
    if constexpr (HASRES) {
      const uint32_t rbase = (uint32_t)((m0 * p.N + n0) * 2), kbase = (uint32_t)(m0 * (p.N >> 3) + (n0 >> 3));
#pragma unroll
      for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          uint32_t rvo = y_lane + (uint32_t)(g * 128), rso = rbase + (uint32_t)(ps * 8 * p.N * 2);
          if (JOIN && p.res_sub) {
            // compact residual: row m = (n, oh, ow) reads compact row (n, oh/2, ow/2) when both are even, else zeros (out of range)
            const uint32_t m = (uint32_t)(m0 + ps * 8 + e_row);
            const uint32_t n_img = fdiv(m, p.div_ohow), rem = m - n_img * (uint32_t)(p.OH * p.OW);
            const uint32_t oh = fdiv(rem, p.div_ow), ow = rem - oh * (uint32_t)p.OW;
            const uint32_t crow = (n_img * (uint32_t)(p.OH >> 1) + (oh >> 1)) * (uint32_t)(p.OW >> 1) + (ow >> 1);
            rvo = (((oh | ow) & 1u) || m >= (uint32_t)p.M) ? OOBB : (crow * (uint32_t)p.N + (uint32_t)(n0 + g * 64 + e_ch * 8)) * 2u;
            rso = 0;
          }
          // (s_nop: the scalar offsets come straight from the SALU; nothing inside an asm statement is padded by the compiler)
          asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" PFR_SCONV_RES_NT
                       : "=v"(rres[g][ps])
                       : "v"(rvo), "s"(rrsrc), "s"(rso)
                       : "memory");
          if constexpr (JOIN)
            asm volatile("s_nop 4\n\tbuffer_load_ubyte %0, %1, %2, %3 offen"
                         : "=v"(rmk[g][ps])
                         : "v"(has_rmask ? k_lane + (uint32_t)(g * 8) : 0u), "s"(mrsrc), "s"(has_rmask ? kbase + (uint32_t)(ps * p.N) : 0u)
                         : "memory");
        }
    }
    u32x4 bxr[NCG][4], cxr[NCG][4];
    uint32_t bmk[NCG][4];
    if constexpr (BNB2) {
      const uint32_t rbase = (uint32_t)((m0 * p.N + n0) * 2);
#pragma unroll
      for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
          asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" PFR_SCONV_BNX_NT
                       : "=v"(cxr[g][ps])
                       : "v"(y_lane + (uint32_t)(g * 128)), "s"(cxrsrc), "s"(rbase + (uint32_t)(ps * 8 * p.N * 2))
                       : "memory");
    }
    if constexpr (BNB) {
      const uint32_t rbase = (uint32_t)((m0 * p.N + n0) * 2), kbase = (uint32_t)(m0 * (p.N >> 3) + (n0 >> 3));
#pragma unroll
      for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          if constexpr (!NOX)
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" PFR_SCONV_BNX_NT
                         : "=v"(bxr[g][ps])
                         : "v"(y_lane + (uint32_t)(g * 128)), "s"(bxrsrc), "s"(rbase + (uint32_t)(ps * 8 * p.N * 2))
                         : "memory");
          // (recomputed mask: the descriptor covers 16 bytes of y, every lane reads byte 0 or gets the out-of-range zero)
          asm volatile("s_nop 4\n\tbuffer_load_ubyte %0, %1, %2, %3 offen"
                       : "=v"(bmk[g][ps])
                       : "v"(bn_bits ? k_lane + (uint32_t)(g * 8) : 0u), "s"(bmrsrc), "s"(bn_bits ? kbase + (uint32_t)(ps * p.N) : 0u)
                       : "memory");
        }
    }
    // ---- granules of the block: the first one starts the accumulators from a constant-zero C operand
#pragma unroll 1
    for (int kc = 0; kc < KG; ++kc) {
      issue((slot + NS - 1) % NS);
      wait_granule();
      const uint32_t gbase = ring0 + slot * GB;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const u32x4 fq = *(lds_cptr)(uintptr_t)(gbase + xo[s]);
        const uint32_t swk = (tail8 && kc >= KG1) ? (uint32_t)sx : (uint32_t)sw;
        const uint32_t wa = wlane + ((((uint32_t)(kc * 8 + 2 * s) | (uint32_t)fhalf) ^ swk) << 4);
        u32x4 fp[TP];
#pragma unroll
        for (int i = 0; i < TP; ++i) fp[i] = *(lds_cptr)(uintptr_t)(wa + i * 32 * RB);
        if (s == 0 && kc == 0) {
#pragma unroll
          for (int i = 0; i < TP; ++i) {
            const f32x16 z0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const f32x16 z = SRC2 ? cb[i] : z0;
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp[i]), __builtin_bit_cast(bf16x8, fq), z, 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int i = 0; i < TP; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp[i]), __builtin_bit_cast(bf16x8, fq), acc[i], 0, 0, 0);
        }
      }
      slot = (slot + 1) % NS;
    }
    // MFMA results are written back pass by pass and the MFMA -> VALU read-after-write distance is SOFTWARE-managed on CDNA.
    // hipcc counts those wait states along the code LAYOUT; it laid the epilogue block in front of the loop it follows and
    // missed the hazard across the back edge (observed: the last pass of the last MFMA — 8 registers x 16 lanes — read stale
    // by the conversions below, a few hundred rows per launch).  16 explicit wait states, tied to the accumulators.
#pragma unroll
    for (int i = 0; i < TP; ++i) asm volatile("s_nop 15" : "+v"(acc[i]));
    // ---- epilogue: the slot of the block's last granule is free until the next DMA is issued into it: it is the staging window
    hist |= 1u;
    const uint32_t wbase = ring0 + ((slot + NS - 1) % NS) * GB;
    const uint32_t ybase = (uint32_t)((m0 * p.N + n0) * 2);
    if constexpr (HASRES || BNB) {
      // the residual / BN-input loads are older than the KG granules issued during this block
      if (KG == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GI) : "memory");
      else if (KG == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GI) : "memory");
      else if (KG == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * GI) : "memory");
      else if (KG == 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * GI) : "memory");
      else if (KG == 10) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(10 * GI) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * GI) : "memory");
      // The destination registers of the inline-asm loads above are DEFINED here as far as the compiler is concerned: their
      // "=v" outputs were valid to it at the asm statement itself, so nothing but this tie stops it from copying / spilling them before the
      // data has landed.  (Empty statements: no instruction is emitted.)
#pragma unroll
      for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          if constexpr (HASRES) asm volatile("" : "+v"(rres[g][ps]));
          if constexpr (JOIN) asm volatile("" : "+v"(rmk[g][ps]));
          if constexpr (BNB && !NOX) asm volatile("" : "+v"(bxr[g][ps]));
          if constexpr (BNB) asm volatile("" : "+v"(bmk[g][ps]));
          if constexpr (BNB2) asm volatile("" : "+v"(cxr[g][ps]));
        }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = 0; g < NCG; ++g) {
      // accumulators (lane: row frow, couts 8*qd + 4*fhalf + 0..3 of tile i) -> window [row][128 B], chunk ^ ((row >> 1) & 7)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const f32x16& a = acc[2 * g + ii];
          bf16x4 v;
          v[0] = (bf16_t)a[4 * qd];
          v[1] = (bf16_t)a[4 * qd + 1];
          v[2] = (bf16_t)a[4 * qd + 2];
          v[3] = (bf16_t)a[4 * qd + 3];
          *(lds_w8ptr)(uintptr_t)(wbase + (ew_w ^ (uint32_t)((ii * 4 + qd) << 4))) = __builtin_bit_cast(u32x2, v);
        }
      }
      if (STATS && bi == 0) {   // shift of the statistics: the wave's first stored row (row 0 of the window: swizzle 0)
        const u32x4 v = *(lds_cptr)(uintptr_t)(wbase + (uint32_t)(e_ch << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) ksh[g][e] = (f32x2){__uint_as_float(v[e] << 16), __uint_as_float(v[e] & 0xffff0000u)};
      }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const u32x4 v = *(lds_cptr)(uintptr_t)(wbase + ((ew_r ^ (uint32_t)((ps & 1) << 6)) + ps * 1024));
        if constexpr (STATS) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2 f = {__uint_as_float(v[e] << 16), __uint_as_float(v[e] & 0xffff0000u)};
            const f32x2 d = f - ksh[g][e];
            s1[g][e] += d;
            s2[g][e] = __builtin_elementwise_fma(d, d, s2[g][e]);
          }
        }
        if constexpr (NOSTORE) continue;
        if constexpr (EP != 0) {
          float f[8], rr8[8];
          Chunk<bf16_t>::unpack(v, f);
          if constexpr (HASRES) Chunk<bf16_t>::unpack(rres[g][ps], rr8);
          if constexpr (TAIL) {
            uint32_t bits = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float z = fmaf(f[e], ta1[g][e], tb1[g][e]);
              z += fmaf(rr8[e], ta2[g][e], tb2[g][e]);
              bits |= (z > 0.f ? 1u : 0u) << e;
              f[e] = fmaxf(z, 0.f);
            }
            buffer_store_byte_sync(bits, korsrc, k_lane + (uint32_t)(g * 8), (uint32_t)(m0 * (p.N >> 3) + (n0 >> 3)) + (uint32_t)(ps * p.N));
          } else if constexpr (JOIN) {
            const uint32_t bits = has_rmask ? rmk[g][ps] : 0xffu;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += ((bits >> e) & 1u) ? rr8[e] : 0.f;
          } else if constexpr (INFER) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              f[e] += bias8[g][e];
              if constexpr (HASRES) f[e] += rr8[e];
              if (p.relu) f[e] = fmaxf(f[e], 0.f);
            }
          }
          u32x4 o = Chunk<bf16_t>::pack(f);
          if constexpr (BNB) {
            // sums over the value AS STORED (what pfr_bn_bwd_reduce would read back) through the BN's own ReLU mask
            float xv[8];
            if constexpr (NOX) {
#pragma unroll
              for (int e = 0; e < 8; ++e) xv[e] = 0.f;
            } else {
              Chunk<bf16_t>::unpack(bxr[g][ps], xv);
            }
            const uint32_t bits = bmk[g][ps];
            // (two-source form: the accumulators start from the bias, so rows past M hold the bias, not 0: they must not enter the sums)
            const bool rowok = !SRC2 || (m0 + ps * 8 + e_row) < p.M;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x2 gs = {rowok ? __uint_as_float(o[e] << 16) : 0.f, rowok ? __uint_as_float(o[e] & 0xffff0000u) : 0.f};
              f32x2 gm;
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int q = 2 * e + h;
                const bool keep = bn_bits ? ((bits >> q) & 1u) != 0 : fmaf(xv[q], bsc[g][q], bsh[g][q]) > 0.f;
                gm[h] = keep ? gs[h] : 0.f;
              }
              if (p.store_masked) o[e] = (__float_as_uint(gm[0]) >> 16) | (__float_as_uint(gm[1]) & 0xffff0000u);
              b1[g][e] += gm;
              if constexpr (!NOX) {
                const f32x2 xh = {fmaf(xv[2 * e], bca[g][2 * e], bcb[g][2 * e]), fmaf(xv[2 * e + 1], bca[g][2 * e + 1], bcb[g][2 * e + 1])};
                b2[g][e] = __builtin_elementwise_fma(gm, xh, b2[g][e]);
              }
              if constexpr (BNB2) {
                const float x0 = __uint_as_float(cxr[g][ps][e] << 16), x1 = __uint_as_float(cxr[g][ps][e] & 0xffff0000u);
                const f32x2 yh = {fmaf(x0, cca[g][2 * e], ccb[g][2 * e]), fmaf(x1, cca[g][2 * e + 1], ccb[g][2 * e + 1])};
                c2[g][e] = __builtin_elementwise_fma(gm, yh, c2[g][e]);
              }
            }
          }
          buffer_store_b128_sync(o, yrsrc, y_lane + (uint32_t)(g * 128), ybase + (uint32_t)(ps * 8 * p.N * 2));
          continue;
        }
        // rows past M lie beyond num_records: the store is dropped
        // (store + wait for the data read-out in one statement: buffer_store_b128_sync, pfr_mma.h)
        buffer_store_b128_sync(v, yrsrc, y_lane + (uint32_t)(g * 128), ybase + (uint32_t)(ps * 8 * p.N * 2));
      }
    }
  }
  // drain: the dummy granules issued past the end must have landed before the LDS is reused or released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (BNB) {
    __syncthreads();   // every wave is past its last panel read and ring use: the LDS is reused below
    // rows past M: the store was dropped, but their (zero + residual-out-of-range-zero) value entered the sums as g = 0: nothing to undo
    float* red = reinterpret_cast<float*>(smem);   // [8 waves][3][NPV]
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float a = b1[g][e][h], q = b2[g][e][h], q2 = BNB2 ? c2[g][e][h] : 0.f;
#pragma unroll
          for (int o = 8; o < 64; o <<= 1) {   // lanes with equal e_ch hold partials of the same couts
            a += __shfl_xor(a, o, 64);
            q += __shfl_xor(q, o, 64);
            if constexpr (BNB2) q2 += __shfl_xor(q2, o, 64);
          }
          if (lane < 8) {
            const int c = g * 64 + lane * 8 + e * 2 + h;
            red[(wave * 3 + 0) * NPV + c] = a;
            red[(wave * 3 + 1) * NPV + c] = q;
            red[(wave * 3 + 2) * NPV + c] = q2;
          }
        }
    __syncthreads();
    for (int c = tid; c < NPW; c += 512) {
      const int sl = c / NPV, cc = c - sl * NPV;
      float a = 0.f, q = 0.f, q2 = 0.f;
      for (int wr = 0; wr < nrw; ++wr) {   // fixed order: deterministic
        const int w = wr * nsub + sl;
        a += red[(w * 3 + 0) * NPV + cc];
        q += red[(w * 3 + 1) * NPV + cc];
        q2 += red[(w * 3 + 2) * NPV + cc];
      }
      p.bn_part[((size_t)r * 2 + 0) * p.N + pn * NPW + c] = a;
      p.bn_part[((size_t)r * 2 + 1) * p.N + pn * NPW + c] = q;
      if constexpr (BNB2) {   // the second BN sees the same g and mask: its first sum is the same number
        p.bn_part2[((size_t)r * 2 + 0) * p.N + pn * NPW + c] = a;
        p.bn_part2[((size_t)r * 2 + 1) * p.N + pn * NPW + c] = q2;
      }
    }
  }
  if (STATS) {
    __syncthreads();   // every wave is past its last panel read and ring use: the LDS is reused below
    // rows this wave processed; those past M contributed y = 0, i.e. d = -ksh: taken out again below
    int nproc = my_blocks * 32, ninv = 0;
    if (my_blocks > 0) {
      const int last_end = row_lo + (wrow + nrw * (my_blocks - 1)) * bstep + 32;
      ninv = last_end > p.M ? last_end - p.M : 0;
    }
    const float nval = (float)(nproc - ninv), finv = (float)ninv;
    float* red = reinterpret_cast<float*>(smem);   // [8 waves][3][NPV]: mean, M2, n
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float a = s1[g][e][h], q = s2[g][e][h];
#pragma unroll
          for (int o = 8; o < 64; o <<= 1) {   // lanes with equal e_ch hold partials of the same couts
            a += __shfl_xor(a, o, 64);
            q += __shfl_xor(q, o, 64);
          }
          const float k = ksh[g][e][h];
          a += finv * k;
          q -= finv * k * k;
          if (lane < 8) {
            const int c = g * 64 + lane * 8 + e * 2 + h;
            red[(wave * 3 + 0) * NPV + c] = nval > 0.f ? k + a / nval : 0.f;
            red[(wave * 3 + 1) * NPV + c] = nval > 0.f ? q - a * a / nval : 0.f;
            red[(wave * 3 + 2) * NPV + c] = nval;
          }
        }
    __syncthreads();
    // merge the row lanes of every column slice (Chan): thread -> (slice, cout)
    for (int c = tid; c < NPW; c += 512) {
      const int sl = c / NPV, cc = c - sl * NPV;
      float n = 0.f, a = 0.f;
      for (int wr = 0; wr < nrw; ++wr) {
        const int w = wr * nsub + sl;
        const float nw = red[(w * 3 + 2) * NPV + cc];
        n += nw;
        a = fmaf(nw, red[(w * 3 + 0) * NPV + cc], a);
      }
      const float mean = n > 0.f ? a / n : 0.f;
      float m2 = 0.f;
      for (int wr = 0; wr < nrw; ++wr) {
        const int w = wr * nsub + sl;
        const float nw = red[(w * 3 + 2) * NPV + cc];
        const float d = red[(w * 3 + 0) * NPV + cc] - mean;
        m2 += red[(w * 3 + 1) * NPV + cc] + nw * d * d;
      }
      p.stats_part[((size_t)r * 2 + 0) * p.N + pn * NPW + c] = mean;
      p.stats_part[((size_t)r * 2 + 1) * p.N + pn * NPW + c] = m2;
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
int sconv_mode() { return pfr_knob(KNOB_SCONV); }   // pfr_set_tuning("sconv"): 0 off, 1 heuristic (default), 2 whenever eligible

// panel width for (N, K): the widest of 256 / 128 / 64 couts that divides N and keeps the panel within 64 KiB
static int sconv_panel(int N, int K, long limit = 65536) {
  for (int np = 256; np >= 64; np >>= 1)
    if (N % np == 0 && (long)np * K * 2 <= limit) return np;
  return 0;
}

struct SconvPlan {
  int np, npanels, nranges, R, ns, tp;
};
// geometry-only eligibility — the ONE test shared by pfr_conv2d_mtile, pfr_conv2d_dgrad_bn_parts and the launch: bf16, K in {64, 128,
// 256, 512}, panel fits, enough rows, output AND input extents (in_rows = N*H*W: 4x the output rows of a stride-2 1x1) below 2 GiB
// (k2 > 0: the two-source form, K = k1 + k2 with k1 in {256, 512} and k2 = k1 / 4: the panel may then take 96 KiB, two ring slots)
bool sconv_plan(int M, int N, int K, long in_rows, int dtype, int out_dtype, SconvPlan* sp, int k2 = 0) {
  const int mode = sconv_mode();
  if (mode == 0 || dtype != PFR_BF16 || out_dtype != PFR_BF16) return false;
  if (k2 ? !((K - k2 == 256 || K - k2 == 512) && (k2 == 64 || k2 == 128)) : (K != 64 && K != 128 && K != 256 && K != 512)) return false;
  if (N % 64 != 0 || (long)M * N * 2 >= ((long)1 << 31) || (long)M * K * 2 >= ((long)1 << 31) || in_rows * K * 2 >= ((long)1 << 31)) return false;
  const int np = sconv_panel(N, K, k2 ? 96 * 1024 : 65536);
  if (!np) return false;
  const int npanels = N / np;
  if (npanels > 32 || (256 % npanels) != 0) return false;
  // re-reading x once per panel (from L2: the panels of a row range run on one XCD) must stay cheaper than what the tile kernels do
  if (mode == 1) {
    // (measured, tools/sconv_bench.py: K = 512 with 4 panels 81 vs 94 us for the tile kernel, with 16 panels 68-72 vs 76-78)
    // (round 6: up to 32 panels and down to 8192 rows — the 7x7 layers of ResNet-50 at bs 256, 12 544 rows: 512 -> 2048 forward and its
    //  data-gradient join, the furthest launch from its bound on the tile kernel (0.17) — train step 17.13 -> 17.03 ms over six interleaved
    //  rounds, profiles/r06_ab.txt #9; either limit alone changes no launch)
    constexpr int maxp = 32, maxk = 512;
    if (npanels > maxp || (!k2 && K > maxk) || (K < 512 && npanels > 8)) return false;
    if (M < 256 * 32) return false;    // too few rows per workgroup for a pipeline
  }
  int nranges = 256 / npanels;
  long R = ((long)M + nranges - 1) / nranges;
  R = (R + 31) / 32 * 32;
  nranges = (int)((M + R - 1) / R);
  const long wbytes = (long)np * K * 2;
  int ns = (int)((160 * 1024 - wbytes) / (8 * 4096));
  if (ns > 4) ns = 4;
  if (ns < 2) return false;
  sp->np = np; sp->npanels = npanels; sp->nranges = nranges; sp->R = (int)R; sp->ns = ns;
  sp->tp = np >= 128 ? 4 : 2;
  return true;
}

template <int TP, bool STATS, int EP = 0>
static int sconv_launch_ns(SconvParams& sp, const SconvPlan& pl, hipStream_t st) {
  const int lds = pl.np * sp.K * 2 + 8 * pl.ns * 4096;
  const dim3 grid(256), block(512);
#define PFR_SCONV_GO(NSV)                                                                                       \
  do {                                                                                                          \
    auto kern = sconv_kernel<TP, NSV, STATS, EP>;                                                                     \
    static std::atomic<unsigned long long> attr_set{0};                                                         \
    PFR_MAX_LDS_ONCE(attr_set, 160 * 1024, (const void*)kern);                                                  \
    hipLaunchKernelGGL(kern, grid, block, lds, st, sp);                                                         \
  } while (0)
  if (pl.ns >= 4) PFR_SCONV_GO(4);
  else if (pl.ns == 3) PFR_SCONV_GO(3);
  else PFR_SCONV_GO(2);
#undef PFR_SCONV_GO
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// takes the launch when it is a plain 1x1 convolution of an eligible geometry; returns 1 when it is not
int sconv_try_launch(IgemmParams& p, int dtype, int out_dtype, hipStream_t st) {
  if (p.R != 1 || p.S != 1 || p.pad != 0 || p.idil_log2 != 0 || p.ldy != p.Cout) return 1;
  if (p.pro_scale || p.act) return 1;
  // y += conv: the join with res = y itself and no mask (each wave reads its rows of y at the block start and writes them in its epilogue)
  const bool acc_inplace = p.accumulate && !p.residual && !p.bias && !p.stats_part && !p.out_relu && !p.bnb_part[0] && !p.res_sub;
  if (p.accumulate && !acc_inplace) return 1;
  // BatchNorm-backward sums in the epilogue (pfr_conv2d_dgrad_bn): one BN, no statistics / bias; with the join its bit mask is required
  const bool bnb = p.bnb_part[0] != nullptr;
  // (a residual WITHOUT its mask is a plain add and goes with either kind of BN mask; a masked residual = the block join = bit mask)
  const bool nox = bnb && (p.bnb_flags & 2);
  if (bnb && (sconv_bnb_mode() != 2 || p.bias || p.stats_part || p.out_relu || (p.residual && p.res_mask && !p.bnb_mask) ||
              (p.bnb_part[1] && !(p.residual && p.bnb_mask)) || (nox && !(p.residual && p.bnb_mask)) ||
              ((p.bnb_flags & 1) && !p.bnb_mask)))
    return 1;
  // inference form: bias (+ plain residual add) (+ ReLU), no statistics; training forms: no bias / ReLU, residual only as the join
  const bool infer = p.bias != nullptr;
  if (infer && (p.stats_part || p.res_mask)) return 1;
  if (!infer && p.out_relu) return 1;
  const bool join = !infer && (p.residual != nullptr || acc_inplace);
  // residual only in its data-gradient join form (bit mask), or — with the BN sums / as the accumulate form — as a plain add
  if (join && ((!p.res_mask && !bnb && !acc_inplace) || p.stats_part)) return 1;
  if (p.res_sub && (!bnb || p.res_mask || p.ostride != 1 || (p.OH & 1) || (p.OW & 1))) return 1;
  if (p.ostride != 1 && (p.H != p.OH * p.ostride || p.W != p.OW * p.ostride)) return 1;
  SconvPlan pl;
  if (!sconv_plan(p.M, p.Cout, p.K, (long)p.N * p.H * p.W, dtype, out_dtype, &pl)) return 1;
  if (p.stats_part && p.want_mtile && p.want_mtile != pl.R) return 1;   // the caller sized stats_part for another kernel's partials
  SconvParams sp;
  sp.x = p.x; sp.w = p.w; sp.y = p.y;
  sp.M = p.M; sp.K = p.K; sp.N = p.Cout;
  sp.H = p.H; sp.W = p.W; sp.OH = p.OH; sp.OW = p.OW; sp.ostride = p.ostride;
  sp.stats_part = p.stats_part;
  sp.res = acc_inplace ? p.y : p.residual; sp.res_mask = p.res_mask;
  sp.res_sub = p.res_sub;
  sp.bias = p.bias; sp.relu = p.out_relu;
  sp.bnx = p.bnb_x[0]; sp.bn_coef = p.bnb_coef[0]; sp.bn_mask = p.bnb_mask; sp.bn_part = p.bnb_part[0];
  sp.bnx2 = p.bnb_x[1]; sp.bn_coef2 = p.bnb_coef[1]; sp.bn_part2 = p.bnb_part[1];
  sp.store_masked = bnb ? (p.bnb_flags & 1) : 0;
  sp.npanels = pl.npanels; sp.nranges = pl.nranges; sp.R = pl.R; sp.npw = pl.np;
  constexpr bool il_on = true;   // block-interleaved row assignment (+5-7 %: profiles/r03_stream_probe.txt)
  sp.interleave = (il_on && p.M % 32 == 0 && (p.M / 32) % pl.nranges == 0 && (long)pl.R * pl.nranges == p.M) ? 1 : 0;
  sp.xbytes = (int)((long)p.N * p.H * p.W * p.K * 2);
  sp.div_ohow = p.div_ohow; sp.div_ow = p.div_ow;
  if (bnb) {   // 64-cout column slices keep the residual / BN-input rows and both sums inside the register budget
    pl.tp = 2;
    if (p.bnb_part[1]) return nox ? sconv_launch_ns<2, false, 8>(sp, pl, st) : sconv_launch_ns<2, false, 6>(sp, pl, st);
    if (nox) return sconv_launch_ns<2, false, 7>(sp, pl, st);
    return p.residual ? sconv_launch_ns<2, false, 5>(sp, pl, st) : sconv_launch_ns<2, false, 4>(sp, pl, st);
  }
  if (infer) {
    if (p.residual) return pl.tp == 2 ? sconv_launch_ns<2, false, 3>(sp, pl, st) : sconv_launch_ns<4, false, 3>(sp, pl, st);
    return pl.tp == 2 ? sconv_launch_ns<2, false, 2>(sp, pl, st) : sconv_launch_ns<4, false, 2>(sp, pl, st);
  }
  if (join) return pl.tp == 2 ? sconv_launch_ns<2, false, 1>(sp, pl, st) : sconv_launch_ns<4, false, 1>(sp, pl, st);
  if (pl.tp == 2) return p.stats_part ? sconv_launch_ns<2, true>(sp, pl, st) : sconv_launch_ns<2, false>(sp, pl, st);
  return p.stats_part ? sconv_launch_ns<4, true>(sp, pl, st) : sconv_launch_ns<4, false>(sp, pl, st);
}

// pfr_conv2d_dgrad_bn through the streaming kernel: pfr_set_tuning("bnb"): 1 = tile kernels (round-2 form),
// 2 = streaming kernels only (geometries they do not take keep the separate pfr_bn_bwd_reduce pass)
int sconv_bnb_mode() { return pfr_knob(KNOB_BNB); }
// partial rows (= row ranges) the streaming kernel leaves for a 1x1 data gradient + BN sums of this geometry, 0 when it does not take it
int sconv_bnb_parts(int M, int N, int K, int dtype) {
  SconvPlan pl;
  if (sconv_bnb_mode() != 2 || !sconv_plan(M, N, K, M, dtype, dtype, &pl)) return 0;
  return pl.nranges;
}

// rows per statistics partial when this kernel takes a (post-op free) 1x1 launch of the geometry, 0 when it does not
int sconv_mtile(int M, int N, int K, long in_rows, int dtype, int out_dtype) {
  SconvPlan pl;
  if (!sconv_plan(M, N, K, in_rows, dtype, out_dtype, &pl)) return 0;
  return pl.R;
}

// ------------------------------------------------------------------------------------------------ recompute form of a block's last conv
// A bottleneck's last convolution (1x1, Cout = 4 x Cin) is HBM-bound on its OUTPUT: writing c3 and reading it back for the block tail
// moves 8x the bytes of its input.  Since the backward pass no longer needs c3 (pfr_bnfree.hip), the forward pass computes it twice
// instead: pfr_conv1x1_stats leaves only the BatchNorm partials, and — after pfr_bn_finalize — pfr_conv1x1_bn_tail recomputes the tile
// and applies relu(a1*c3 + b1 + shortcut) (+ ReLU bit mask) in its epilogue, on the same bf16-rounded values the stored tensor held:
// results are bit-identical to pfr_conv2d_fwd + pfr_bn_act_mask.
static bool tail_geom(int dtype, int N, int H, int W, int C, int Cout, SconvPlan* pl) {
  if (dtype != PFR_BF16 || (long)N * H * W >= ((long)1 << 31)) return false;
  return sconv_plan(N * H * W, Cout, C, (long)N * H * W, dtype, dtype, pl);
}
// rows per statistics partial of pfr_conv1x1_stats (= pfr_conv2d_mtile of the same geometry), 0: geometry not taken by the streaming kernel
extern "C" int pfr_conv1x1_tail_mtile(int dtype, int N, int H, int W, int C, int Cout) {
  SconvPlan pl;
  return tail_geom(dtype, N, H, W, C, Cout, &pl) ? pl.R : 0;
}
static void tail_params(SconvParams& sp, const SconvPlan& pl, const void* x, const void* w, int N, int H, int W, int C, int Cout) {
  memset(&sp, 0, sizeof(sp));
  sp.x = x; sp.w = w;
  sp.M = N * H * W; sp.K = C; sp.N = Cout;
  sp.H = H; sp.W = W; sp.OH = H; sp.OW = W; sp.ostride = 1;
  sp.npanels = pl.npanels; sp.nranges = pl.nranges; sp.R = pl.R; sp.npw = pl.np;
  constexpr bool il_on = true;   // block-interleaved row assignment (+5-7 %: profiles/r03_stream_probe.txt)
  sp.interleave = (il_on && sp.M % 32 == 0 && (sp.M / 32) % pl.nranges == 0 && (long)pl.R * pl.nranges == sp.M) ? 1 : 0;
  sp.xbytes = (int)((long)sp.M * C * 2);
  sp.div_ohow = make_fastdiv((uint32_t)(H * W)); sp.div_ow = make_fastdiv((uint32_t)W);
}
extern "C" int pfr_conv1x1_stats(const void* x, const void* w, int dtype, int N, int H, int W, int C, int Cout, float* stats_part,
                                 hipStream_t st) {
  PFR_CHECK_ARG(x && w && stats_part, "pfr_conv1x1_stats: null pointer");
  SconvPlan pl;
  if (!tail_geom(dtype, N, H, W, C, Cout, &pl)) {
    pfr_set_error("pfr_conv1x1_stats: geometry not taken by the streaming kernel (pfr_conv1x1_tail_mtile == 0)");
    return PFR_ERR_UNSUPPORTED;
  }
  SconvParams sp;
  tail_params(sp, pl, x, w, N, H, W, C, Cout);
  sp.y = stats_part;     // (descriptor base of dummy loads only: nothing is stored)
  sp.stats_part = stats_part;
  return pl.tp == 2 ? sconv_launch_ns<2, true, 9>(sp, pl, st) : sconv_launch_ns<4, true, 9>(sp, pl, st);
}
extern "C" int pfr_conv1x1_bn_tail(const void* x, const void* w, void* y, unsigned char* mask, int dtype, int N, int H, int W, int C,
                                   int Cout, const float* a1, const float* b1, const void* res, const float* a2, const float* b2,
                                   hipStream_t st) {
  PFR_CHECK_ARG(x && w && y && mask && a1 && b1 && res, "pfr_conv1x1_bn_tail: null pointer");
  PFR_CHECK_ARG((a2 == nullptr) == (b2 == nullptr), "pfr_conv1x1_bn_tail: a2 and b2 go together");
  SconvPlan pl;
  if (!tail_geom(dtype, N, H, W, C, Cout, &pl)) {
    pfr_set_error("pfr_conv1x1_bn_tail: geometry not taken by the streaming kernel (pfr_conv1x1_tail_mtile == 0)");
    return PFR_ERR_UNSUPPORTED;
  }
  SconvParams sp;
  tail_params(sp, pl, x, w, N, H, W, C, Cout);
  sp.y = y; sp.res = res; sp.mask_out = mask;
  sp.tail_a1 = a1; sp.tail_b1 = b1; sp.tail_a2 = a2; sp.tail_b2 = b2;
  if (a2) { pl.tp = 2; return sconv_launch_ns<2, false, 11>(sp, pl, st); }   // (64-cout slices: four coefficient sets in registers)
  return pl.tp == 2 ? sconv_launch_ns<2, false, 10>(sp, pl, st) : sconv_launch_ns<4, false, 10>(sp, pl, st);
}

// ------------------------------------------------------------------------------------------------ two-source data gradient (pfr_bnfree.hip)
// dx [M][Cout] = [g | z] · wcatᵀ + bias with g [M][C1] (the masked block-output gradient), z [M][C2] (conv3's input), wcat [Cout][C1 + C2]
// (= [A∘W | S] rows, pfr_bn3_bwd_weights) — conv3's BN-input-free data gradient in ONE launch, bias added in fp32 inside the accumulators —
// plus the BatchNorm-backward sums of the BN whose output gradient dx is (recomputed ReLU mask: coef rows 2, 3), as pfr_conv2d_dgrad_bn.
static bool dgrad2_geom(int dtype, int N, int H, int W, int C1, int C2, int Cout, SconvPlan* pl) {
  if (dtype != PFR_BF16 || sconv_bnb_mode() != 2 || (long)N * H * W >= ((long)1 << 31)) return false;
  const long M = (long)N * H * W;
  if (M * (C1 + C2) * 2 >= ((long)1 << 31)) return false;
  return sconv_plan((int)M, Cout, C1 + C2, M, dtype, dtype, pl, C2);
}
// partial rows of bn_part ([parts][2][Cout]) the launch writes; 0: geometry not taken (run pfr_conv2d_fwd(z, S, bias) + pfr_conv2d_dgrad_bn)
extern "C" int pfr_conv1x1_dgrad2_bn_parts(int dtype, int N, int H, int W, int C1, int C2, int Cout) {
  SconvPlan pl;
  return dgrad2_geom(dtype, N, H, W, C1, C2, Cout, &pl) ? pl.nranges : 0;
}
extern "C" int pfr_conv1x1_dgrad2_bn(const void* g, const void* z, const void* wcat, const float* bias, void* dx, int dtype, int N, int H,
                                     int W, int C1, int C2, int Cout, const void* bn_x, const float* bn_coef, float* bn_part,
                                     hipStream_t st) {
  PFR_CHECK_ARG(g && z && wcat && bias && dx && bn_x && bn_coef && bn_part, "pfr_conv1x1_dgrad2_bn: null pointer");
  SconvPlan pl;
  if (!dgrad2_geom(dtype, N, H, W, C1, C2, Cout, &pl)) {
    pfr_set_error("pfr_conv1x1_dgrad2_bn: geometry not taken by the streaming kernel (pfr_conv1x1_dgrad2_bn_parts == 0)");
    return PFR_ERR_UNSUPPORTED;
  }
  SconvParams sp;
  tail_params(sp, pl, g, wcat, N, H, W, C1 + C2, Cout);
  sp.xbytes = (int)((long)sp.M * C1 * 2);
  sp.x2 = z; sp.K2 = C2; sp.x2bytes = (int)((long)sp.M * C2 * 2);
  sp.cbias = bias;
  sp.y = dx;
  sp.bnx = bn_x; sp.bn_coef = bn_coef; sp.bn_part = bn_part;
  pl.tp = 2;
  return sconv_launch_ns<2, false, 12>(sp, pl, st);
}
