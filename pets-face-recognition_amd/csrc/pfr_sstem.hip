// pfr_sstem.hip — weight-stationary, HALO-staged stem convolution on MFMA (bf16): the 7x7 / stride-2 / pad-3 first convolution of torchvision's
// ResNet in its space-to-depth form, a 4x4 / stride-1 / pad-2 convolution over [N][H][W][16] -> 64 channels (pfr_s2d_input / pfr_s2d_weight).
//
// Replaces, for that geometry, `nn.Conv2d(3, 64, 7, stride 2, padding 3).forward` of the reference's backbone
// (/root/reference/configs/dog_fe/fe_dogs_config.py:102-103 -> torchvision resnet50.conv1) — the same launch pfr_igemm.hip's tile kernel took.
//
// Why (profiles/r06_ab.txt #5): at bs 256 the tile kernel stages 25 088 tiles x 96 KB = 2.4 GB through LDS for 103 MB of input — every pixel once
// per tap (16 x), the 32 KB weight matrix once per tile — and runs at 2.2 TB/s of its 514 MB of HBM traffic (235 us).  Here, as pfr_sconv3.hip:
//   * the weight matrix [64][4*4*16] stays in LDS for the life of a persistent workgroup (32 KB; 16-byte chunks XOR-swizzled by cout & 15);
//   * a wave computes a 4 x 8 pixel patch x 64 output channels from ONE halo tile of 7 x 11 pixels x 32 B (three LDS-DMA instructions into its
//     private 2-slot ring, issued under the previous patch's MFMAs); a tap (r, s) is ONE k-group: its B fragment is the 16 channels of pixel
//     (pr + r, pc + s) of the tile, its A fragments the tap's 32-byte piece of the 64 weight rows: 16 k-steps x 2 MFMAs per patch;
//   * no workgroup barrier in the loop; 4 waves and 72 KB of LDS per workgroup: two workgroups per CU, the second covers the first's stalls;
//   * epilogue, BatchNorm statistics and the store-data hazard handling are pfr_sconv3.hip's: the patch leaves through a 4 KB LDS window as four
//     1 KiB rows of 8 pixels x 128 B.
// LDS layout of the halo tile: pixel (hr, hc) is pixel index q = 12 hr + hc (pitch 12: 24 chunks per row, 22 used); its two 16-byte channel chunks
// are stored at chunk (h ^ ((q >> 3) & 1)).  MFMA column n (= lane & 31) is NOT pixel (n >> 3, n & 7): the sixteen lanes of a ds_read_b128 group are
// {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}, and with pitch 12 the pixel rows 0, 2 of a patch cover the residues 0..15 of q mod 16 exactly once,
// rows 1, 3 likewise — so group one takes patch rows 0 and 2, group two rows 1 and 3 (rho(n) below), for every tap shift: conflict free.
// Accumulation order = the tile kernel's (taps row-major, 16 channels per MFMA): results are bit-identical (tests/test_kernels_gpu.py).
#include "pfr_igemm.h"
#include <stdlib.h>

struct SstemParams {
  const void* x;        // [N][H][W][16] bf16 (space-to-depth image)
  const void* w;        // [64][4][4][16]
  void* y;              // [N][H][W][64]
  int N, H, W;
  float* stats_part;    // [nparts][2][64] (mean, M2) or nullptr
  const float* bias;    // INFER: per-cout bias of the BN-folded convolution
  int relu;             // INFER: ReLU after the bias
  int nblk, bpw;        // patches in total, patches per workgroup (contiguous in patch order)
  int tiles_x, tpi;     // W / 8, patches per image
  FastDiv div_tpi, div_tx;
};

// patch pixel (as 8 * pr + pc) of MFMA column n, see the header
__device__ __forceinline__ int sstem_rho(int n) {
  const bool g2 = (n >= 4 && n < 12) || (n >= 16 && n < 20) || n >= 28;      // second ds_read_b128 lane group
  const int a = g2 ? (n < 12 ? n - 4 : (n < 20 ? n - 8 : n - 16)) : (n < 4 ? n : (n < 16 ? n - 8 : n - 12));   // 0..15 inside the group
  const int pr = (a >> 3) * 2 + (g2 ? 1 : 0);
  return pr * 8 + (a & 7);
}

#ifndef PFR_S3_PD
#define PFR_S3_PD 3
#endif
// INFER: y = relu?(result + bias) — the BN-folded inference plan; added to the bf16-rounded result in the read-back pass
// (one more bf16 rounding of the pre-activation than the tile kernel's fp32 epilogue, as in pfr_sconv.hip)
template <bool STATS, bool INFER = false>
__global__ __launch_bounds__(256, 2) void sstem_kernel(SstemParams p) {
  static_assert(!(STATS && INFER), "the inference variant publishes no statistics");
  constexpr int WB = 64 * 512;           // weight bytes
  constexpr int GB = 3072;               // ring slot: 7 halo rows of 12 pixels x 32 B (2688 B) in three 1 KiB DMA instructions
  constexpr int GI = 3;                  // DMA instructions per halo tile
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
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.N * p.H * p.W * 32, 0x00020000);
  __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, act_bytes, 0x00020000);
  const uint32_t OOBB = 0xF0000000u;

  // ---- weights -> LDS (once): row = cout (512 B = 32 chunks), chunk c stored at c ^ (row & 15)
  for (int t = wave; t < WB / 1024; t += 4) {
    const int L = (t << 10) + (lane << 4);
    const int row = L >> 9, pc = (L & 511) >> 4;
    const int lc = pc ^ (row & 15);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(smem + (t << 10)), 16, row * 512 + (lc << 4), 0, 0, 0);
  }

  // ---- per-lane constants
  const int frow = lane & 31, fhalf = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t ring0 = lds0 + WB + wave * (NS * GB + 4096);
  char* const ringp = smem + WB + wave * (NS * GB + 4096);
  // A fragments (weights): row frow of tile i, chunk (2*t + fhalf) ^ (frow & 15) of tap t: the low four chunk bits take eight lane-dependent
  // values (t & 7), bit 4 (t >> 3) is an immediate
  uint32_t wsw[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) wsw[q] = lds0 + (uint32_t)(frow * 512 + ((((2 * q) | fhalf) ^ (frow & 15)) << 4));
  // B fragments (halo tile): this lane's patch pixel rho -> tile pixel index q0 = 12 pr + pc; tap (r, s) reads pixel q0 + 12 r + s, chunk
  // fhalf ^ bit 3 of that index
  const int rho = sstem_rho(frow);
  const int q0 = (rho >> 3) * 12 + (rho & 7);
  uint32_t xof[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int q = q0 + (t >> 2) * 12 + (t & 3);
    xof[t] = (uint32_t)(q * 32 + ((fhalf ^ ((q >> 3) & 1)) << 4));
  }
  // epilogue window (as pfr_sconv3.hip): 32 pixel rows of 128 B, chunk ^ ((row >> 1) & 7); this lane's accumulator column is pixel rho
  const int sx = (rho >> 1) & 7;
  const uint32_t ew_w = (uint32_t)(rho * 128 + (sx << 4) + fhalf * 8);
  const int e_row = lane >> 3, e_ch = lane & 7;
  const uint32_t ew_r = (uint32_t)(e_row * 128 + ((e_ch ^ (e_row >> 1)) << 4));

  // ---- loader: halo tile of patch `blk` (index within this workgroup's range)
  // DMA instruction t, lane l: chunk ci = 64 t + l of the slot = halo row hr = ci / 24, chunk cw = ci % 24 of that row: pixel hc = cw / 2
  // (hc = 11: the pitch's pad pixel), stored half cw & 1 holds channel half (cw & 1) ^ bit 3 of the pixel index 12 hr + hc
  int hrv[GI], hcv[GI];
  uint32_t rel[GI];
#pragma unroll
  for (int t = 0; t < GI; ++t) {
    const int ci = 64 * t + lane;
    const int hr = ci / 24, cw = ci - hr * 24, hc = cw >> 1;
    hrv[t] = (hr < 7 && hc < 11) ? hr : 1 << 20;
    hcv[t] = hc;
    const int q = hr * 12 + hc;
    rel[t] = (uint32_t)((hr * p.W + hc) * 32 + (((cw & 1) ^ ((q >> 3) & 1)) << 4));
  }
  uint32_t goff[GI];
  int t_h0 = 0, t_w0 = 0, t_base = 0;
  bool t_ok = false;
  auto tile_head = [&](int blk) __attribute__((always_inline)) {
    const int bid = blk_lo + blk;
    const uint32_t n_img = fdiv((uint32_t)bid, p.div_tpi);
    const uint32_t rem = (uint32_t)bid - n_img * (uint32_t)p.tpi;
    const uint32_t ty = fdiv(rem, p.div_tx), tx = rem - ty * (uint32_t)p.tiles_x;
    t_h0 = (int)ty * 4 - 2;
    t_w0 = (int)tx * 8 - 2;
    t_base = (((int)n_img * p.H + t_h0) * p.W + t_w0) * 32;
    t_ok = blk < nb;
  };
  auto tile_part = [&](int t) __attribute__((always_inline)) {
    const bool ok = t_ok && (unsigned)(t_h0 + hrv[t]) < (unsigned)p.H && (unsigned)(t_w0 + hcv[t]) < (unsigned)p.W;
    goff[t] = ok ? (uint32_t)(t_base + (int)rel[t]) : OOBB;
  };
  auto issue_one = [&](int slot, int t) __attribute__((always_inline)) {
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

  // ---- one patch: 16 k16 steps (= taps) into `cur`; in their shadow the DMA of the next tile (steps 0-2), the addresses of the one after
  //      (3-6) and the epilogue of the previous patch `prv` (7-15)
  // The fragments of step q + PD are requested before the MFMAs of step q are issued (as pfr_sconv3.hip).
  constexpr int PD = PFR_S3_PD;
  int slot = 0;
  bool stores_pending = false;
  auto patch = [&](f32x16 (&cur)[2], const f32x16 (&prv)[2], int bi, bool has_prev) __attribute__((always_inline)) {
    const int blk = wave + 4 * bi;
    // this patch's tile has landed when only the stores of the last epilogue (issued after it) are outstanding
    if (stores_pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t sbase = ring0 + slot * GB;
    u32x4 fq[PD + 2], fp0[PD + 2], fp1[PD + 2];
    auto rd = [&](int kq, int b) __attribute__((always_inline)) {
      fq[b] = *(lds_cptr)(uintptr_t)(sbase + xof[kq]);
      const uint32_t wa = wsw[kq & 7] + (uint32_t)((kq >> 3) << 8);
      fp0[b] = *(lds_cptr)(uintptr_t)(wa);
      fp1[b] = *(lds_cptr)(uintptr_t)(wa + 32 * 512);
    };
#pragma unroll
    for (int q = 0; q < PD; ++q) rd(q, q);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kq = 0; kq < 16; ++kq) {
      const int b = kq % (PD + 2);
      const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (kq == 0) cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp0[b]), __builtin_bit_cast(bf16x8, fq[b]), z, 0, 0, 0);
      else cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp0[b]), __builtin_bit_cast(bf16x8, fq[b]), cur[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kq + PD < 16) rd(kq + PD, (kq + PD) % (PD + 2));
      __builtin_amdgcn_sched_barrier(0);
      if (kq == 0) cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp1[b]), __builtin_bit_cast(bf16x8, fq[b]), z, 0, 0, 0);
      else cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp1[b]), __builtin_bit_cast(bf16x8, fq[b]), cur[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- shadow work of this step
      if (kq < 3) issue_one(slot ^ 1, kq);                 // next patch's tile (the slot's last reader was the previous patch)
      else if (kq == 3) tile_head(blk + 8);                // addresses of the patch after the next
      else if (kq < 7) tile_part(kq - 4);
      else if (has_prev) {
        if (kq == 7) { epi_head(blk - 4); epi_write(prv, 0); }
        else if (kq < 11) { epi_write(prv, 2 * (kq - 8) + 1); epi_write(prv, 2 * (kq - 8) + 2); }
        else if (kq == 11) { epi_write(prv, 7); epi_shift(); }
        else epi_row(kq - 12);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // MFMA -> VALU read-after-write distance is software-managed; explicit wait states tied to the accumulators (pfr_sconv.hip)
    asm volatile("s_nop 15" : "+v"(cur[0]));
    asm volatile("s_nop 15" : "+v"(cur[1]));
    stores_pending = has_prev;
    slot ^= 1;
  };

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
// geometry-only eligibility: 4x4, stride 1, pad 2, 16 -> 64 channels, bf16, same-size output, W % 8 == 0, H % 4 == 0  ("sstem" tuning: 0 off)
bool sstem_geom(int N, int H, int W, int C, int Cout, int R, int S, int stride, int pad, int idil_log2, int OH, int OW, int dtype,
                int out_dtype, int* bpw) {
  if (!pfr_knob(KNOB_SSTEM)) return false;
  if (sconv_mode() == 0 || dtype != PFR_BF16 || out_dtype != PFR_BF16) return false;
  if (R != 4 || S != 4 || stride != 1 || pad != 2 || idil_log2 != 0 || C != 16 || Cout != 64) return false;
  if (OH != H || OW != W || (W & 7) || (H & 3) || (long)N * H * W * 128 >= ((long)1 << 31)) return false;
  const int nblk = N * (H / 4) * (W / 8);
  if (nblk < 512 * 4) return false;   // fewer than one patch per wave
  if (bpw) *bpw = (nblk + 511) / 512;
  return true;
}

int sstem_try_launch(IgemmParams& p, int dtype, int out_dtype, hipStream_t st) {
  if (p.ldy != p.Cout || p.accumulate || p.pro_scale || p.act || p.bnb_part[0] || p.residual) return 1;
  const bool infer = p.bias != nullptr;      // inference form: bias (+ ReLU), no statistics; training form: neither
  if (infer ? p.stats_part != nullptr : p.out_relu != 0) return 1;
  int bpw;
  if (!sstem_geom(p.N, p.H, p.W, p.C, p.Cout, p.R, p.S, p.ostride, p.pad, p.idil_log2, p.OH, p.OW, dtype, out_dtype, &bpw)) return 1;
  if (p.stats_part && p.want_mtile && p.want_mtile != bpw * 32) return 1;
  SstemParams sp;
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
  const int lds = 64 * 512 + 4 * (2 * 3072 + 4096);
  const dim3 grid((unsigned)((sp.nblk + bpw - 1) / bpw)), block(256);
  static std::atomic<unsigned long long> attr_set{0};
  PFR_MAX_LDS_ONCE(attr_set, 160 * 1024, (const void*)sstem_kernel<true>, (const void*)sstem_kernel<false>, (const void*)sstem_kernel<false, true>);
  if (infer) hipLaunchKernelGGL((sstem_kernel<false, true>), grid, block, lds, st, sp);
  else if (p.stats_part) hipLaunchKernelGGL(sstem_kernel<true>, grid, block, lds, st, sp);
  else hipLaunchKernelGGL(sstem_kernel<false>, grid, block, lds, st, sp);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
