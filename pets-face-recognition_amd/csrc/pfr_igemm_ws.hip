// pfr_igemm_ws.hip — wave-specialised persistent implicit-GEMM convolution (forward with BatchNorm statistics, plain data
// gradients) for bf16: ONE 512-thread workgroup per CU = 4 MFMA waves + 4 memory waves.
//
// Same arithmetic and operand layouts as pfr_igemm.hip / pfr_igemm_p.hip (bit-identical outputs); it replaces the same
// `nn.Conv2d` forward / input-gradient calls of the reference's backbone (torchvision resnet50 built at /root/reference
// configs/dog_fe/fe_dogs_config.py:102-103).
//
// Why (measured, tools/p_trace.py + tools/probe/dma_waves_probe.hip): on gfx950 a wave can issue one 1 KiB LDS-DMA
// instruction per ~51-57 clocks (~18-20 B/clk); the CU's path into LDS saturates at ~71 B/clk only when FOUR waves (one
// per SIMD) issue back to back.  A wave that issues its own operand DMA therefore spends ~125 clocks of in-order issue time per
// instruction during which its SIMD's matrix pipe starves: a k-step of a 4-wave 128x128 tile takes 1150 cycles for 256 cycles
// of MFMA work.  Here the roles are split:
//   * memory waves 4-7 (one per SIMD): tile decode, gather addressing, the LDS-DMA of every k-step of every tile of this
//     workgroup (a 4-slot ring that runs 3 stages ahead, across tile boundaries, counted `s_waitcnt vmcnt`), AND the output
//     side: they read the finished tile from an LDS staging buffer and store whole 16-byte row segments, accumulating the
//     per-channel BatchNorm statistics on the way (one partial per 64 rows), spread over the first k-steps of the next tile;
//   * MFMA waves 0-3 (one per SIMD, 2x2 over a 256 x 128 tile: 128 accumulator registers each): only {barrier, ds_read_b128,
//     v_mfma_f32_32x32x16_bf16}; the fragments of the next k-group — also the first of the next k-step / next tile — are read
//     while the current MFMAs run (the barrier that publishes a stage is taken one stage early); at the end of a tile the
//     accumulators are dropped into the staging buffer as bf16 and the next tile starts at once.
// A 256 x 128 tile stages 24 KiB per 32-wide k-step for 512 MFMA cycles per wave: 47 B/clk at full matrix rate, inside what
// the four memory waves deliver.  LDS: 4 x 24 KiB ring + 64 KiB staging = 160 KiB.
#include "pfr_igemm.h"
#include <stdlib.h>

template <int N, int B, int LOWBIT>
struct HalvingW {   // cross-lane sum by halving (see pfr_igemm_p.hip): N values left, next lane bit B
  __device__ __forceinline__ static void sum(float* v, int lane) {
    if constexpr (B >= LOWBIT) {
      constexpr int H = N / 2;
      const bool up = (lane >> B) & 1;
#pragma unroll
      for (int k = 0; k < H; ++k) {
        const float send = up ? v[k] : v[k + H];
        const float keep = up ? v[k + H] : v[k];
        v[k] = keep + __shfl_xor(send, 1 << B, 64);
      }
      HalvingW<H, B - 1, LOWBIT>::sum(v, lane);
    }
  }
  __device__ __forceinline__ static void pick(float* v, int lane) {
    if constexpr (B >= LOWBIT) {
      constexpr int H = N / 2;
      const bool up = (lane >> B) & 1;
#pragma unroll
      for (int k = 0; k < H; ++k) v[k] = up ? v[k + H] : v[k];
      HalvingW<H, B - 1, LOWBIT>::pick(v, lane);
    }
  }
};

template <int BQ, int BP>
__global__ __launch_bounds__(512, 2) void igemm_ws_kernel(IgemmParams p, int total_tiles) {
  typedef bf16_t T;
  constexpr int KP = 8, KCH = 4, ROWB = 64, BK = 32, RPI = 16;
  constexpr int NST = 4;
  constexpr int TP = BP / 64, TQ = BQ / 64;          // 32x32 accumulator tiles per MFMA wave (2 x 2 waves)
  constexpr int QCH = BQ / (4 * RPI), PCH = BP / (4 * RPI);   // DMA instructions per memory wave per k-step
  constexpr int NLD = QCH + PCH;
  constexpr int STAGE = (BP + BQ) * ROWB;
  constexpr int ORB = BP * 2;                        // bytes per row of the output staging tile
  constexpr int NCH = ORB / 16;                      // 16-byte chunks per staged row (16 or 8)
  constexpr int RPS = 64 / NCH;                      // rows per wave-wide 16-byte access (4 or 8)
  constexpr int OUTB = BQ * ORB;
  constexpr int NSI = (BQ / 4) / RPS;                // store instructions per memory wave per tile (16 or 8)
  constexpr int SMEM = NST * STAGE + OUTB;
  static_assert(SMEM <= 160 * 1024, "LDS");
  static_assert(NLD * (NST - 1) <= 63, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  char* outb = smem + NST * STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, G = gridDim.x;
  const int mlim = p.pclass ? p.mclass : p.M;
  const int nk_plain = p.K / BK;

  auto tile_of = [&](int r) __attribute__((always_inline)) -> int {
    const int base = r * G;
    const int nr = min(G, total_tiles - base);
    return (b < nr) ? base + (int)xcd_remap((uint32_t)b, (uint32_t)nr) : -1;
  };
  auto decode = [&](int m, int ph, int pw, uint32_t& n_img, uint32_t& oh, uint32_t& ow) __attribute__((always_inline)) {
    if (p.pclass) {
      n_img = fdiv((uint32_t)m, p.div_chw);
      const uint32_t rem = m - n_img * (uint32_t)((p.OH >> 1) * (p.OW >> 1));
      const uint32_t i = fdiv(rem, p.div_cw);
      oh = 2 * i + ph;
      ow = 2 * (rem - i * (p.OW >> 1)) + pw;
    } else {
      n_img = fdiv((uint32_t)m, p.div_ohow);
      const uint32_t rem = m - n_img * (uint32_t)(p.OH * p.OW);
      oh = fdiv(rem, p.div_ow);
      ow = rem - oh * p.OW;
    }
  };
  auto tile_nk = [&](int t) __attribute__((always_inline)) -> int {
    if (!p.pclass) return nk_plain;
    const int cls = (t / p.tilesN) / p.tpc;
    const int tr0 = (p.pad + (cls >> 1)) & 1, ts0 = (p.pad + (cls & 1)) & 1;
    return ((p.R - tr0 + 1) / 2) * ((p.S - ts0 + 1) / 2) * (p.C / BK);
  };
  if (tile_of(0) < 0) return;
#ifdef PFR_IGEMM_TRACE
  unsigned long long c_a = 0, c_b = 0, c_c = 0, c_d = 0, c_t = 0;
  long long n_ks = 0, n_tl = 0;
#define WSTART() do { c_t = __builtin_amdgcn_s_memtime(); } while (0)
#define WSTAMP(acc) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc += n_ - c_t; c_t = n_; } while (0)
#else
#define WSTART() do {} while (0)
#define WSTAMP(acc) do {} while (0)
#endif

  if (wave >= 4) {
    // =========================================================================================== memory waves
    const int lw = wave - 4;
    const int rsub = lane / KCH;
    const int lc = (lane % KCH) ^ row_swizzle<KCH>(rsub);     // (rows of a pass start at multiples of 16: swizzle = f(rsub))
    const int dmask = (1 << p.idil_log2) - 1;
    const uint32_t OOBB = 0xF0000000u;
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.C * sizeof(T)), 0x00020000);
    __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((size_t)p.Cout * p.K * sizeof(T)), 0x00020000);
    __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.M * p.ldy * sizeof(T)), 0x00020000);
    const int tstep = p.pclass ? 2 : 1;
    int ihb[QCH], iwb[QCH], pixb[QCH];
    uint32_t qbase[QCH], wbase[PCH];
    int u_tr = 0, u_ts = 0, cbyte = 0, tapbyte = 0, l_tr0 = 0, l_ts0 = 0, l_nk = 0, lr = 0;
    bool ldone = false;
    auto newtap = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < QCH; ++j) {
        int ih = ihb[j] + u_tr, iw = iwb[j] + u_ts;
        bool ok = (((ih | iw) & dmask) == 0);
        ih >>= p.idil_log2;
        iw >>= p.idil_log2;
        ok = ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W) && (u_tr < p.R);
        const uint32_t off = (uint32_t)(((pixb[j] + ih * p.W + iw) * p.C + lc * KP) * (int)sizeof(T));
        qbase[j] = ok ? off : OOBB;
      }
    };
    auto l_begin = [&](int t) __attribute__((always_inline)) {
      const int tn = t % p.tilesN, tm = t / p.tilesN;
      const int n0 = tn * BP;
      const int cls = p.pclass ? tm / p.tpc : 0;
      const int ph = cls >> 1, pw = cls & 1;
      const int m0 = (p.pclass ? tm - cls * p.tpc : tm) * BQ;
#pragma unroll
      for (int j = 0; j < QCH; ++j) {
        const int m = m0 + (j * 4 + lw) * RPI + rsub;
        if (m < mlim) {
          uint32_t n_img, oh, ow;
          decode(m, ph, pw, n_img, oh, ow);
          ihb[j] = (int)oh * p.ostride - p.pad;
          iwb[j] = (int)ow * p.ostride - p.pad;
          pixb[j] = n_img * p.H * p.W;
        } else {
          ihb[j] = -(1 << 28);
          iwb[j] = -(1 << 28);
          pixb[j] = 0;
        }
      }
#pragma unroll
      for (int j = 0; j < PCH; ++j) {
        const int row = n0 + (j * 4 + lw) * RPI + rsub;
        wbase[j] = row < p.Cout ? (uint32_t)(((size_t)row * p.K + lc * KP) * sizeof(T)) : OOBB;
      }
      l_tr0 = p.pclass ? ((p.pad + ph) & 1) : 0;
      l_ts0 = p.pclass ? ((p.pad + pw) & 1) : 0;
      l_nk = tile_nk(t);
      u_tr = l_tr0;
      u_ts = l_ts0;
      cbyte = 0;
      tapbyte = (u_tr * p.S + u_ts) * p.C * (int)sizeof(T);
      newtap();
    };
    auto l_next = [&]() __attribute__((always_inline)) {
      for (;;) {
        ++lr;
        const int t = tile_of(lr);
        if (t < 0) { ldone = true; return; }
        l_begin(t);
        if (l_nk > 0) return;
      }
    };
    auto gload = [&](int buf) __attribute__((always_inline)) {
      char* base = smem + buf * STAGE;
#ifdef PFR_IGEMM_TRACE
      if (!(p.dbg & 16))
#endif
      {
#pragma unroll
      for (int j = 0; j < PCH; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(base + (j * 4 + lw) * RPI * ROWB), 16,
                                                 (int)(wbase[j] + (uint32_t)(tapbyte + cbyte)), 0, 0, 0);
#pragma unroll
      for (int j = 0; j < QCH; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(base + (BP + (j * 4 + lw) * RPI) * ROWB),
                                                 16, (int)(qbase[j] + (uint32_t)cbyte), 0, 0, 0);
      }
      if (--l_nk == 0) { l_next(); return; }
      cbyte += BK * (int)sizeof(T);
      if (cbyte >= p.C * (int)sizeof(T)) {
        cbyte = 0;
        u_ts += tstep;
        if (u_ts >= p.S) { u_ts = l_ts0; u_tr += tstep; }
        tapbyte = (u_tr * p.S + u_ts) * p.C * (int)sizeof(T);
        newtap();
      }
    };

    // ---- output side state: the tile that is being stored (the one BEFORE the tile the MFMA waves are computing)
    const int ech = lane % NCH, erow = lane / NCH;
    const bool do_stats = p.stats_part != nullptr;
    float s1[KP], s2[KP], ksh[KP];
    int st_t = -1;        // tile in the staging buffer (-1: none)
    int st_done = NSI;    // store instructions of it already issued
    constexpr int CH = NSI / 4;   // row-segment stores per group (their LDS reads are issued together)
    auto store_some = [&](int n) __attribute__((always_inline)) -> int {   // issues up to n more stores of tile st_t; returns how many
      if (st_t < 0) return 0;
      const int tn = st_t % p.tilesN, tm = st_t / p.tilesN;
      const int n0 = tn * BP;
      const int cls = p.pclass ? tm / p.tpc : 0;
      const int ph = cls >> 1, pw = cls & 1;
      const int m0 = (p.pclass ? tm - cls * p.tpc : tm) * BQ;
      const int co = n0 + ech * KP;
      const int rowb = p.ldy * (int)sizeof(T);
      const int r0 = lw * (BQ / 4);     // this wave's 64 rows of the tile
      int cnt = 0;
#pragma unroll 1
      for (; n >= CH && st_done < NSI; n -= CH, st_done += CH, cnt += CH) {
        u32x4 v[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int row = r0 + (st_done + u) * RPS + erow;
          v[u] = *reinterpret_cast<const u32x4*>(outb + row * ORB + ((ech ^ (NCH == 16 ? (row & 15) : ((row >> 1) & 7))) << 4));
        }
        if (do_stats && st_done == 0) {
          // shift = row 0 of this wave's 64 rows (all lanes of a column group read the same chunk: an LDS broadcast)
          Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(outb + r0 * ORB + ((ech ^ (NCH == 16 ? (r0 & 15) : ((r0 >> 1) & 7))) << 4)), ksh);
#pragma unroll
          for (int e = 0; e < KP; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          if (do_stats) {
            float f[KP];
            Chunk<T>::unpack(v[u], f);
#pragma unroll
            for (int e = 0; e < KP; ++e) {
              const float d = f[e] - ksh[e];
              s1[e] += d;
              s2[e] = fmaf(d, d, s2[e]);
            }
          }
          int mm = m0 + r0 + (st_done + u) * RPS + erow;
          if (p.pclass) {
            uint32_t n_img, oh, ow;
            decode(mm, ph, pw, n_img, oh, ow);
            mm = (int)((n_img * p.OH + oh) * p.OW + ow);
          }
#ifdef PFR_IGEMM_TRACE
          if (p.dbg & 2) { asm volatile("" ::"v"(v[u])); continue; }
#endif
          __builtin_amdgcn_raw_buffer_store_b128(v[u], yrsrc, mm * rowb + co * (int)sizeof(T), 0, 0);
        }
      }
      if (st_done == NSI) {
        if (do_stats) {
          constexpr int LOWBIT = NCH == 16 ? 4 : 3;
          float v[2 * KP], kk[KP];
#pragma unroll
          for (int e = 0; e < KP; ++e) { v[e] = s1[e]; v[KP + e] = s2[e]; kk[e] = ksh[e]; }
          HalvingW<2 * KP, 5, LOWBIT>::sum(v, lane);
          constexpr int NLEFT = (2 * KP) >> (6 - LOWBIT);
          const bool up = (lane >> 5) & 1;
          int ebase = 0;
#pragma unroll
          for (int bb = 4; bb >= LOWBIT; --bb) ebase += ((lane >> bb) & 1) * ((2 * KP) >> (6 - bb));
          HalvingW<KP, 4, LOWBIT>::pick(kk, lane);
          const float inv = 1.f / (float)(BQ / 4);
          float* dstp = p.stats_part + ((size_t)(tm * 4 + lw) * 2) * p.Cout + co;
#pragma unroll
          for (int k = 0; k < NLEFT; ++k) {
            const float other = __shfl_xor(v[k], 32, 64);
            const float a = up ? other : v[k], b2 = up ? v[k] : other;
            dstp[(up ? p.Cout : 0) + ebase + k] = up ? (b2 - a * a * inv) : (kk[k] + a * inv);
          }
        }
        st_t = -1;
      }
      return cnt;
    };
    // counted wait: at most n vector-memory operations of this wave may remain outstanding (n is rounded DOWN to a supported
    // immediate: waiting for more than necessary is always safe)
    auto wait_vm = [&](int n) __attribute__((always_inline)) {
#define PFR_WVM(N) else if (n >= N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
      if (n >= NLD + 4 * CH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD + 4 * CH) : "memory");
      PFR_WVM(NLD + 2 * CH);
      PFR_WVM(NLD + CH);
      PFR_WVM(NLD);
      PFR_WVM(4 * CH);
      PFR_WVM(2 * CH);
      PFR_WVM(CH);
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef PFR_WVM
    };

    // ---- schedule
    l_begin(tile_of(0));
    if (l_nk == 0) l_next();
    int issued = 0, slot = 0;
#pragma unroll 1
    for (; issued < NST - 1 && !ldone; ++issued) { gload(slot); slot = (slot + 1 == NST) ? 0 : slot + 1; }
    // stages 0 and 1 landed (stage 2 may stay in flight) -> publish
    if (issued >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int consumed = 0;
#pragma unroll 1
    for (int r = 0;; ++r) {
      const int t = tile_of(r);
      if (t < 0) break;
      const int nk = tile_nk(t);
      // the previous tile is stored during the first k-steps of this one, all of it before this tile's LAST k-step ends
      // (the MFMA waves drop this tile's accumulators into the staging buffer inside that last k-step)
      const int spread = nk - 1 < 4 ? nk - 1 : 4;            // (nk >= 2: igemm_ws_eligible)
      const int per = CH * ((4 + spread - 1) / spread);
#pragma unroll 1
      for (int kt = 0; kt < nk; ++kt) {
        // order: the new stage's DMA first, then this k-step's share of the output stores: the counted wait below then
        // leaves exactly those (the youngest operations) outstanding and still covers stage `consumed + 1`
        bool loaded = false;
        WSTART();
        if (!ldone) { gload(slot); slot = (slot + 1 == NST) ? 0 : slot + 1; ++issued; loaded = true; }
        WSTAMP(c_a);
        const int nst = (st_t >= 0 && kt < nk - 1) ? store_some(kt == nk - 2 ? NSI : per) : 0;
        WSTAMP(c_b);
        ++consumed;
        // (steady state: a stage was issued, no stores this k-step: one compare instead of the cascade)
        if (loaded && nst == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else wait_vm((loaded ? NLD : 0) + nst);
        WSTAMP(c_c);
        __builtin_amdgcn_s_barrier();
        WSTAMP(c_d);
      }
      // the MFMA waves have dropped tile t into the staging buffer (before the barrier above)
      st_t = t;
      st_done = 0;
    }
    store_some(NSI);   // the last tile
#ifdef PFR_IGEMM_TRACE
    if (p.trace && tid == 256) {
      long long* o = p.trace + (size_t)b * 8;
      o[2] = (long long)c_a; o[3] = (long long)c_b; o[4] = (long long)c_c; o[5] = (long long)c_d;
    }
#endif
    return;
  }

  // ============================================================================================= MFMA waves
  const int wp = wave >> 1, wq = wave & 1;
  f32x16 acc[TP][TQ];
  u32x4 fp[2][TP], fq[2][TQ];
  const int frow = lane & 31, fhalf = lane >> 5, fsw = row_swizzle<KCH>(frow);
  auto rdfrag = [&](int slot, int kg, int bsel) __attribute__((always_inline)) {
    const char* sb = smem + slot * STAGE;
    const char* ldsP = sb + (wp * (BP / 2)) * ROWB;
    const char* ldsQ = sb + (BP + wq * (BQ / 2)) * ROWB;
    const int off = (((kg * 2 + fhalf) ^ fsw) << 4);
#pragma unroll
    for (int i = 0; i < TP; ++i) fp[bsel][i] = *reinterpret_cast<const u32x4*>(ldsP + (i * 32 + frow) * ROWB + off);
#pragma unroll
    for (int j = 0; j < TQ; ++j) fq[bsel][j] = *reinterpret_cast<const u32x4*>(ldsQ + (j * 32 + frow) * ROWB + off);
  };
  const uint32_t out_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)outb;
  // total k-steps of this workgroup (the last one has no successor stage to prefetch from)
  int total_ks = 0;
#pragma unroll 1
  for (int r = 0;; ++r) {
    const int t = tile_of(r);
    if (t < 0) break;
    total_ks += tile_nk(t);
  }
  __builtin_amdgcn_s_barrier();     // stages 0 and 1 are in LDS
  int slot_c = 0, ks = 0;
  if (total_ks > 0) rdfrag(0, 0, 0);
#pragma unroll 1
  for (int r = 0;; ++r) {
    const int t = tile_of(r);
    if (t < 0) break;
    const int nk = tile_nk(t);
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
      for (int j = 0; j < TQ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      const int nslot = (slot_c + 1 == NST) ? 0 : slot_c + 1;
      const bool has_next = ks + 1 < total_ks;
      WSTART();
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        const int bsel = kg & 1;
#ifdef PFR_IGEMM_TRACE
        if (!(p.dbg & 64))
#endif
        {
        if (kg == 0) rdfrag(slot_c, 1, 1);
        else if (has_next) rdfrag(nslot, 0, 0);
        }
#ifdef PFR_IGEMM_TRACE
        if (p.dbg & 32) continue;
#endif
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
          for (int j = 0; j < TQ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp[bsel][i]),
                                                                 __builtin_bit_cast(bf16x8, fq[bsel][j]), acc[i][j], 0, 0, 0);
      }
      slot_c = nslot;
      ++ks;
      if (kt == nk - 1) {
        // tile finished: accumulators -> staging tile [BQ rows][BP couts] bf16 (chunks swizzled by row); inline-asm stores: a
        // compiler-visible LDS store would make hipcc drain the LDS-DMA queue first (there is none on these waves, but the
        // waits it inserts are conservative) and they must be complete before the barrier below
        const int row0 = wq * (BQ / 2) + (lane & 31);
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
          const int row = row0 + j * 32;
          const int sw = NCH == 16 ? (row & 15) : ((row >> 1) & 7);
#pragma unroll
          for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const int cb = (wp * (BP / 2) + i * 32 + 8 * qd + 4 * (lane >> 5)) * 2;
              const uint32_t dst = out_lds + row * ORB + ((((cb >> 4) ^ sw)) << 4) + (cb & 15);
              bf16x4 v;
              v[0] = (bf16_t)acc[i][j][4 * qd];
              v[1] = (bf16_t)acc[i][j][4 * qd + 1];
              v[2] = (bf16_t)acc[i][j][4 * qd + 2];
              v[3] = (bf16_t)acc[i][j][4 * qd + 3];
              asm volatile("ds_write_b64 %0, %1" ::"v"(dst), "v"(__builtin_bit_cast(u32x2, v)) : "memory");
            }
        }
      }
#ifdef PFR_IGEMM_TRACE
      asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
      ++n_ks;
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WSTAMP(c_a);
      __builtin_amdgcn_s_barrier();
      WSTAMP(c_b);
    }
#ifdef PFR_IGEMM_TRACE
    ++n_tl;
#endif
  }
#ifdef PFR_IGEMM_TRACE
  if (p.trace && tid == 0) {
    long long* o = p.trace + (size_t)b * 8;
    o[0] = (long long)c_a; o[1] = (long long)c_b; o[6] = n_ks; o[7] = n_tl;
  }
#endif
}

// ------------------------------------------------------------------------------------------------
static int ws_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}
static int g_ws_mode = -1;
int igemm_ws_mode() {
  if (g_ws_mode < 0) g_ws_mode = getenv("PFR_IGEMM_WS") ? atoi(getenv("PFR_IGEMM_WS")) : 0;
  return g_ws_mode;
}
void igemm_ws_set_mode(int v) { g_ws_mode = v; }

// eligible: bf16 in/out, C a multiple of 32, whole 256-row tiles, Cout a multiple of 64, no post-ops / prologue / filter,
// every tile has >= 2 k-steps (K >= 64) or none (the zero classes of a 1x1 stride-2 data gradient are NOT supported here)
bool igemm_ws_eligible(const IgemmParams& p, int dtype, int out_dtype) {
  if (dtype != PFR_BF16 || out_dtype != PFR_BF16) return false;
  if (p.C % 32 != 0 || p.pro_scale || p.act || p.bias || p.accumulate || p.out_relu || p.residual || p.bnb_part[0]) return false;
  const bool pcl = igemm_pclass_ok(p);
  if (pcl && (p.R == 1 || p.C / 32 < 2)) return false;
  const int mrows = pcl ? p.N * (p.OH / 2) * (p.OW / 2) : p.M;
  if (mrows % 256 != 0 || p.Cout % 64 != 0 || p.ldy != p.Cout) return false;
  if (p.K / 32 < 2) return false;
  if ((size_t)p.M * p.ldy * 2 >= ((size_t)1 << 31)) return false;
  return true;
}

template <int BQ, int BP>
static int launch_ws(IgemmParams& p, hipStream_t st) {
  p.pclass = igemm_pclass_ok(p) ? 1 : 0;
  p.mclass = p.N * (p.OH / 2) * (p.OW / 2);
  p.tpc = (p.mclass + BQ - 1) / BQ;
  p.div_chw = make_fastdiv((uint32_t)((p.OH / 2) * (p.OW / 2) > 0 ? (p.OH / 2) * (p.OW / 2) : 1));
  p.div_cw = make_fastdiv((uint32_t)(p.OW / 2 > 0 ? p.OW / 2 : 1));
  p.tilesM = p.pclass ? 4 * p.tpc : (p.M + BQ - 1) / BQ;
  p.tilesN = (p.Cout + BP - 1) / BP;
  const int total = p.tilesM * p.tilesN;
  const int grid = total < ws_num_cus() ? total : ws_num_cus();
  hipLaunchKernelGGL((igemm_ws_kernel<BQ, BP>), dim3((unsigned)grid), dim3(512), 0, st, p, total);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

int igemm_ws_launch(IgemmParams& p, hipStream_t st) {
  if (p.Cout % 128 == 0) return launch_ws<256, 128>(p, st);
  return launch_ws<256, 64>(p, st);
}
