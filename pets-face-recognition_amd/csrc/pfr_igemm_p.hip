// pfr_igemm_p.hip — persistent, cross-tile-pipelined implicit-GEMM convolution on MFMA (forward, data gradient, GEMM).
//
// Same arithmetic, operand layouts and epilogue semantics as pfr_igemm.hip (it replaces the same `nn.Conv2d` / `nn.Linear`
// calls of the reference: torchvision resnet50 at /root/reference/configs/dog_fe/fe_dogs_config.py:102-103, `F.linear` at
// losses/large_margin.py:71, models/swin.py Linear layers) — what changes is WHEN things happen inside a compute unit.
//
// Why: the one-tile-per-workgroup kernel serialises, per tile, {address set-up → first DMA latency → k-loop with a 2-slot
// ring (one exposed L2/HBM round trip per k-step) → transpose → store}.  For the short reductions that dominate ResNet-50
// (K = 64 … 512: a 128x128 tile has 2 … 16 k-steps) that chain, not MFMA or HBM, bounds the tile (14 µs per
// tile of which 0.9 µs are MFMA).  Here
//   * a workgroup is PERSISTENT: 2 per CU (4 waves each, 80 KiB LDS each), each walks tiles b, b+G, b+2G … of the launch;
//   * the LDS-DMA ring (NST = 4 slots of one k-step: (BP + BQ) rows x 64 B) is fed by a loader state that runs AHEAD of the
//     MFMA consumer across tile boundaries: while tile t is in its last k-steps, its epilogue and its stores, the k-steps of
//     tile t+1 are already in flight (counted `s_waitcnt vmcnt(N)`: only the oldest stage is waited for; gfx950 retires
//     loads and stores of a wave in issue order, so outstanding output stores never have to drain either);
//   * the epilogue is per WAVE (its 64x64 accumulator block goes through a private 4 KiB LDS window in 32-row chunks,
//     XOR-swizzled so that the ds_read_b128 side is conflict free and the ds_write_b64 side 2-way — the minimum for 16-byte
//     aligned rows): no workgroup barrier between the last MFMA of a tile and the first of the next;
//   * the second workgroup of the CU fills the matrix pipe while this one runs its epilogue.
// BatchNorm statistics partials are published per HALF tile (the 64 rows of one wave row): pfr_conv2d_mtile reports that
// granularity to the caller.
#include "pfr_igemm.h"
#include <stdlib.h>
#include <string.h>

// Cross-lane sum of NV per-lane values over the lanes that differ in bits [LOWBIT, 6) — by halving: at every step a lane
// keeps one half of the values and hands the other half to its partner, so NV values cost NV - (NV >> steps) exchanges
// (14 for 16 values over 8 lanes) instead of NV per step.  Afterwards v[0 .. (NV >> (6 - LOWBIT))) hold complete sums of the
// original indices  k + sum_b bit_b(lane) * (NV >> (6 - b))  for b = 5 … LOWBIT.
template <int N, int B, int LOWBIT>
struct Halving {   // N values left, next lane bit B (compile-time indices only: no register-array selects)
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
      Halving<H, B - 1, LOWBIT>::sum(v, lane);
    }
  }
  __device__ __forceinline__ static void pick(float* v, int lane) {
    if constexpr (B >= LOWBIT) {
      constexpr int H = N / 2;
      const bool up = (lane >> B) & 1;
#pragma unroll
      for (int k = 0; k < H; ++k) v[k] = up ? v[k + H] : v[k];
      Halving<H, B - 1, LOWBIT>::pick(v, lane);
    }
  }
};
template <int NV, int LOWBIT>
__device__ __forceinline__ void halving_sum(float* v, int lane) { Halving<NV, 5, LOWBIT>::sum(v, lane); }

// LEAN: epilogue specialised for whole tiles of bf16 output without post-ops (the train-step forward convs with their
// BatchNorm statistics, and the plain data gradients): no bounds checks, buffer stores whose row offset is a scalar.
// PF: operand fragments of the NEXT k-group are read from LDS while the MFMAs of the current one run — also across k-steps and
// tile boundaries (the barrier that publishes a stage is taken one stage early, so the first fragments of stage s+1 are
// already readable during stage s): the ~300-cycle ds_read latency at the head of every k-step disappears from the chain.
// WS (wave specialisation): a FIFTH wave per workgroup does nothing but the operand DMA (address arithmetic, tile decode,
// issue, counted waits) while the four MFMA waves only {barrier, ds_read, MFMA, epilogue}.  Why: an LDS-DMA instruction costs
// its issuing wave ~125 cycles of in-order issue time whenever the CU's texture-address path is busy (it sustains 1 KiB per
// ~16-23 clocks), and the MFMA pipe of that wave's SIMD starves meanwhile — with the issue on its own wave the two
// resources (TA ~ 44-65 B/clk/CU, MFMA) run concurrently instead of back to back (tools/p_trace.py: a k-step of a 4-wave
// workgroup takes 1150 cycles with the DMA issue inline, 660 without, for 256 cycles of MFMA work).
template <typename T, typename TO, int BQ, int BP, int KCH_, int NST_, bool LEAN, bool PF = false, bool WS = false>
__global__ __launch_bounds__(WS ? 320 : 256, WS ? 3 : (((BP + BQ) * KCH_ * 16 * NST_ + 4 * 4096 <= 80 * 1024) ? 2 : 1)) void igemm_p_kernel(IgemmParams p, int total_tiles) {
  constexpr int KP = DT<T>::KPACK;
  constexpr int KCH = KCH_;
  constexpr int ROWB = KCH * 16;
  constexpr int BK = KCH * KP;
  constexpr int RPI = 64 / KCH;
  constexpr int NW = 4, WP = 2, WQ = 2;
  constexpr int TP = BP / (WP * 32), TQ = BQ / (WQ * 32);
  constexpr int LW = WS ? 1 : NW;                             // waves that issue the operand DMA
  constexpr int QCH = BQ / (LW * RPI), PCH = BP / (LW * RPI);
  static_assert(TP >= 1 && TQ >= 1 && QCH >= 1 && PCH >= 1, "tile too small");
  constexpr int NLD = QCH + PCH;
  constexpr int NST = NST_;
  static_assert(NST >= 2 && NST <= 4, "ring depth");
  constexpr int STAGE = (BP + BQ) * ROWB;
  constexpr int KPO = 16 / (int)sizeof(TO);
  constexpr int EI = (sizeof(TO) == 2 && TP >= 2) ? 2 : 1;   // 32-cout accumulator tiles per epilogue chunk
  constexpr int NEI = TP / EI;                                // epilogue chunks per 32-row group
  constexpr int ERB = EI * 32 * (int)sizeof(TO);              // bytes per staged output row (64 or 128)
  constexpr int NCH = ERB / 16;                               // 16-byte chunks per staged row
  constexpr int EWAVE = 32 * ERB;                             // staging window of one wave
  constexpr int RPS = 64 / NCH;                               // rows covered by one wave-wide 16-byte access
  constexpr int SMEM = NST * STAGE + NW * EWAVE;
  static_assert(SMEM <= 160 * 1024, "LDS");
  static_assert(!PF || NST == 4, "the early-barrier schedule needs a 4-slot ring");
  static_assert(!(WS && PF), "WS uses the plain schedule");
  static_assert(!WS || NLD * (NST - 1) <= 63, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WQ, wq = wave % WQ;
  const int b = blockIdx.x, G = gridDim.x;

  // round r of this workgroup -> logical tile (consecutive logical tiles of a round sit on one XCD: they share A rows / L2)
  auto tile_of = [&](int r) __attribute__((always_inline)) -> int {
    const int base = r * G;
    const int nr = min(G, total_tiles - base);
    return (b < nr) ? base + (int)xcd_remap((uint32_t)b, (uint32_t)nr) : -1;
  };
  auto decode = [&](int m, int mlim, int ph, int pw, uint32_t& n_img, uint32_t& oh, uint32_t& ow) __attribute__((always_inline)) -> bool {
    if (m >= mlim) return false;
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
    return true;
  };
  const int mlim = p.pclass ? p.mclass : p.M;

  // ------------------------------------------------------------------------------------------ loader (runs ahead)
  const int rsub = lane / KCH;
  const int lw = WS ? 0 : wave;                      // index of this wave among the DMA-issuing waves
  const bool is_loader = WS && wave == NW;
  // logical 16-byte chunk this lane fetches for pass j (swizzle on the SOURCE side; for LW = 4 it does not depend on j)
  auto lcj = [&](int j) __attribute__((always_inline)) -> int { return (lane % KCH) ^ row_swizzle<KCH>((j * LW + lw) * RPI + rsub); };
  const int dmask = (1 << p.idil_log2) - 1;
  const uint32_t OOBB = 0xF0000000u;   // beyond num_records: the LDS-DMA writes zeros
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.C * sizeof(T)), 0x00020000);
  __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((size_t)p.Cout * p.K * sizeof(T)), 0x00020000);
  const int tstep = p.pclass ? 2 : 1;
  const int nk_plain = p.K / BK;

  int ihb[QCH], iwb[QCH], pixb[QCH];
  uint32_t qbase[QCH], wbase[PCH];
  int u_tr = 0, u_ts = 0, cbyte = 0, tapbyte = 0, l_tr0 = 0, l_ts0 = 0;
  int l_nk = 0;        // k-steps the loader still has to issue for its current tile
  int lr = 0;          // round of the loader's current tile
  bool ldone = false;

  auto newtap = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < QCH; ++j) {
      int ih = ihb[j] + u_tr, iw = iwb[j] + u_ts;
      bool ok = (((ih | iw) & dmask) == 0);
      ih >>= p.idil_log2;
      iw >>= p.idil_log2;
      ok = ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W) && (u_tr < p.R);
      const uint32_t off = (uint32_t)(((pixb[j] + ih * p.W + iw) * p.C + lcj(j) * KP) * (int)sizeof(T));
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
      const int m = m0 + (j * LW + lw) * RPI + rsub;
      uint32_t n_img, oh, ow;
      if (decode(m, mlim, ph, pw, n_img, oh, ow)) {
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
      const int row = n0 + (j * LW + lw) * RPI + rsub;
      wbase[j] = row < p.Cout ? (uint32_t)(((size_t)row * p.K + lcj(j) * KP) * sizeof(T)) : OOBB;
    }
    l_tr0 = p.pclass ? ((p.pad + ph) & 1) : 0;
    l_ts0 = p.pclass ? ((p.pad + pw) & 1) : 0;
    if (p.pclass) {
      const int ntr = (p.R - l_tr0 + 1) / 2, nts = (p.S - l_ts0 + 1) / 2;
      l_nk = ntr * nts * (p.C / BK);
    } else {
      l_nk = nk_plain;
    }
    u_tr = l_tr0;
    u_ts = l_ts0;
    cbyte = 0;
    tapbyte = (u_tr * p.S + u_ts) * p.C * (int)sizeof(T);
    newtap();
  };
  // moves the loader to its next tile that has k-steps (parity classes of a 1x1 / stride-2 data gradient other than (0,0) have
  // none: their outputs are zeros) or marks it done
  auto l_next = [&]() __attribute__((always_inline)) {
    for (;;) {
      ++lr;
      const int t = tile_of(lr);
      if (t < 0) { ldone = true; return; }
      l_begin(t);
      if (l_nk > 0) return;
    }
  };
  // issues the LDS-DMA of the loader's next k-step into ring slot `buf`, then advances (possibly into the next tile)
  auto gload = [&](int buf) __attribute__((always_inline)) {
    char* base = smem + buf * STAGE;
#ifdef PFR_IGEMM_TRACE
    if (!(p.dbg & 16))   // experiment: no operand DMA at all
#endif
    {
#pragma unroll
    for (int j = 0; j < PCH; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(base + (j * LW + lw) * RPI * ROWB),
                                               16, (int)(wbase[j] + (uint32_t)(tapbyte + cbyte)), 0, 0, 0);
#pragma unroll
    for (int j = 0; j < QCH; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(base + (BP + (j * LW + lw) * RPI) * ROWB),
                                               16, (int)(qbase[j] + (uint32_t)cbyte), 0, 0, 0);
    }
    if (--l_nk == 0) {
      l_next();
      return;
    }
    cbyte += BK * (int)sizeof(T);
    if (cbyte >= p.C * (int)sizeof(T)) {
      cbyte = 0;
      u_ts += tstep;
      if (u_ts >= p.S) { u_ts = l_ts0; u_tr += tstep; }
      tapbyte = (u_tr * p.S + u_ts) * p.C * (int)sizeof(T);
      newtap();
    }
  };

  if constexpr (WS) {
    if (is_loader) {
      // ---- the loader wave: fill all ring slots, then per stage {wait until it has landed, barrier (= publish it and learn
      //      that the stage before it has been consumed), refill the slot that stage left}
      const int t0 = tile_of(0);
      if (t0 < 0) return;
      l_begin(t0);
      if (l_nk == 0) l_next();
      int issued = 0, slot = 0;
#pragma unroll 1
      for (; issued < NST && !ldone; ++issued) { gload(slot); slot = (slot + 1 == NST) ? 0 : slot + 1; }
#pragma unroll 1
      for (int s = 0; s < issued; ++s) {
        const int younger = issued - (s + 1);
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST >= 4 ? 3 * NLD : 0) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST >= 3 ? 2 * NLD : 0) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s >= 1 && !ldone) { gload(slot); slot = (slot + 1 == NST) ? 0 : slot + 1; ++issued; }
      }
      return;
    }
  }
  if constexpr (!WS) {
    const int t0 = tile_of(0);
    if (t0 < 0) return;
    l_begin(t0);
    if (l_nk == 0) l_next();
  } else {
    if (tile_of(0) < 0) return;
  }
  int inflight = 0, slot_l = 0, slot_c = 0;
#ifdef PFR_IGEMM_TRACE
  // profiling build: shader-clock totals per workgroup: [2] waiting for the oldest stage + barrier, [3] MFMA k-steps (incl. the
  // DMA issue inside), [4] epilogues; [0]/[1] wall clock (100 MHz) at start / end, [5] tiles, [6] k-steps
  unsigned long long c_wait = 0, c_mma = 0, c_epi = 0, c_t;
  long long n_tiles = 0, n_ks = 0;
  const long long w_start = wall_clock64();
#define PSTAMP(acc) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc += n_ - c_t; c_t = n_; } while (0)
#define PSTART() do { c_t = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(acc) do {} while (0)
#define PSTART() do {} while (0)
#endif
  auto issue = [&]() __attribute__((always_inline)) {
    gload(slot_l);
    slot_l = (slot_l + 1 == NST) ? 0 : slot_l + 1;
    ++inflight;
  };
  if constexpr (!WS) {
#pragma unroll 1
    for (int s = 0; s < NST - 1 && !ldone; ++s) issue();
  }
  // PF: fragment registers, double-buffered by k-group parity, live across k-steps and tiles
  constexpr int NKG = KCH / 2;
  u32x4 fp[2][TP], fq[2][TQ];
  const int frow = lane & 31, fhalf = lane >> 5, fsw = row_swizzle<KCH>(frow);
  auto rdfrag = [&](int slot, int kg, int bsel) __attribute__((always_inline)) {
    const char* sb = smem + slot * STAGE;
    const char* ldsP = sb + (wp * (BP / WP)) * ROWB;
    const char* ldsQ = sb + (BP + wq * (BQ / WQ)) * ROWB;
    const int off = (((kg * 2 + fhalf) ^ fsw) << 4);
#pragma unroll
    for (int i = 0; i < TP; ++i) fp[bsel][i] = *reinterpret_cast<const u32x4*>(ldsP + (i * 32 + frow) * ROWB + off);
#pragma unroll
    for (int j = 0; j < TQ; ++j) fq[bsel][j] = *reinterpret_cast<const u32x4*>(ldsQ + (j * 32 + frow) * ROWB + off);
  };
  if constexpr (PF) {
    // stages 0 and 1 must have landed (stage 2 may stay in flight), then the first fragments of stage 0
    if (inflight >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rdfrag(0, 0, 0);
  }

  char* ewin = smem + NST * STAGE + wave * EWAVE;   // this wave's epilogue staging window
  const uint32_t ewin_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ewin;   // its LDS byte address
  const int ech = lane % NCH, erow = lane / NCH;    // epilogue: 16-byte chunk / row (within a pass) of this lane
  char* yb = reinterpret_cast<char*>(p.y);
  __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.M * p.ldy * sizeof(TO)), 0x00020000);

  // ------------------------------------------------------------------------------------------ consumer
#pragma unroll 1
  for (int r = 0;; ++r) {
    const int t = tile_of(r);
    if (t < 0) break;
    const int tn = t % p.tilesN, tm = t / p.tilesN;
    const int n0 = tn * BP;
    const int cls = p.pclass ? tm / p.tpc : 0;
    const int ph = cls >> 1, pw = cls & 1;
    const int m0 = (p.pclass ? tm - cls * p.tpc : tm) * BQ;
    int nk = nk_plain;
    if (p.pclass) {
      const int tr0 = (p.pad + ph) & 1, ts0 = (p.pad + pw) & 1;
      nk = ((p.R - tr0 + 1) / 2) * ((p.S - ts0 + 1) / 2) * (p.C / BK);
    }

    f32x16 acc[TP][TQ];
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
      for (int j = 0; j < TQ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if constexpr (WS) {
#pragma unroll 1
      for (int kt = 0; kt < nk; ++kt) {
        PSTART();
        __builtin_amdgcn_s_barrier();          // the loader wave arrives here once this stage has landed
        PSTAMP(c_wait);
        const char* base = smem + slot_c * STAGE;
        mma_kstep_sw<T, KCH, TP, TQ>(base + (wp * (BP / WP)) * ROWB, base + (BP + wq * (BQ / WQ)) * ROWB, lane, acc);
        slot_c = (slot_c + 1 == NST) ? 0 : slot_c + 1;
#ifdef PFR_IGEMM_TRACE
        asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
        PSTAMP(c_mma);
        ++n_ks;
#endif
      }
    } else if constexpr (PF) {
#pragma unroll 1
      for (int kt = 0; kt < nk; ++kt) {
        PSTART();
        if (!ldone) issue();                 // stage s+3 into the slot stage s-1 left (everyone passed the last barrier)
        const int nslot = (slot_c + 1 == NST) ? 0 : slot_c + 1;
        const bool has_next = inflight >= 2;
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg) {
          const int bsel = kg & 1;
          if (kg + 1 < NKG) rdfrag(slot_c, kg + 1, bsel ^ 1);
          else if (has_next) rdfrag(nslot, 0, bsel ^ 1);   // first fragments of the next stage (next tile's, at a tile end)
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int i = 0; i < TP; ++i)
#pragma unroll
              for (int j = 0; j < TQ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp[bsel][i]),
                                                                     __builtin_bit_cast(bf16x8, fq[bsel][j]), acc[i][j], 0, 0, 0);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int i = 0; i < TP; ++i)
#pragma unroll
                for (int j = 0; j < TQ; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fp[bsel][i][e]), __uint_as_float(fq[bsel][j][e]),
                                                                    acc[i][j], 0, 0, 0);
          }
        }
        --inflight;
        slot_c = nslot;
#ifdef PFR_IGEMM_TRACE
        asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
        PSTAMP(c_mma);
        ++n_ks;
#endif
        // publish stage s+2 for the next iteration: only stage s+3 may stay in flight
        if (inflight >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        PSTAMP(c_wait);
      }
    } else
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      // the oldest stage in flight is the one to consume: wait until only the (inflight - 1) younger ones are outstanding
      const int younger = inflight - 1;
      PSTART();
      if (NST >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
      else if (NST >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      PSTAMP(c_wait);
      const char* base = smem + slot_c * STAGE;
      mma_kstep_sw<T, KCH, TP, TQ>(base + (wp * (BP / WP)) * ROWB, base + (BP + wq * (BQ / WQ)) * ROWB, lane, acc, [&]() {
        if (!ldone) issue();
      });
      --inflight;
      slot_c = (slot_c + 1 == NST) ? 0 : slot_c + 1;
#ifdef PFR_IGEMM_TRACE
      asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
      PSTAMP(c_mma);
      ++n_ks;
#endif
    }
    PSTART();

    // ---------------------------------------------------------------------------------------- per-wave epilogue
    if constexpr (LEAN) {
      static_assert(sizeof(TO) == 2, "lean epilogue: bf16 output");
      constexpr int LOWBIT = NCH == 8 ? 3 : 2;
      const bool do_stats = p.stats_part != nullptr;
      const int rowb = p.ldy * (int)sizeof(TO);
      float s1[NEI][KPO], s2[NEI][KPO], ksh[NEI][KPO];
#pragma unroll
      for (int ii = 0; ii < NEI; ++ii)
#pragma unroll
        for (int e = 0; e < KPO; ++e) { s1[ii][e] = 0.f; s2[ii][e] = 0.f; ksh[ii][e] = 0.f; }
      const int mrow0 = m0 + wq * (BQ / WQ);
      const int co0 = n0 + wp * (BP / WP) + ech * KPO;
      const int voff0 = ((mrow0 + erow) * p.ldy + co0) * (int)sizeof(TO);
#pragma unroll
      for (int j = 0; j < TQ; ++j) {
#pragma unroll
        for (int ii = 0; ii < NEI; ++ii) {
          {
            const int row = lane & 31;
            const int sw = (row >> 1) & (NCH - 1);
#pragma unroll
            for (int ie = 0; ie < EI; ++ie) {
              const int i = ii * EI + ie;
#pragma unroll
              for (int qd = 0; qd < 4; ++qd) {
                const int cb = (ie * 32 + 8 * qd + 4 * (lane >> 5)) * (int)sizeof(TO);
                const uint32_t dst = ewin_lds + row * ERB + ((((cb >> 4) ^ sw)) << 4) + (cb & 15);
                bf16x4 v;
                v[0] = (bf16_t)acc[i][j][4 * qd];
                v[1] = (bf16_t)acc[i][j][4 * qd + 1];
                v[2] = (bf16_t)acc[i][j][4 * qd + 2];
                v[3] = (bf16_t)acc[i][j][4 * qd + 3];
                asm volatile("ds_write_b64 %0, %1" ::"v"(dst), "v"(__builtin_bit_cast(u32x2, v)) : "memory");
              }
            }
          }
          asm volatile("" ::: "memory");
          if (do_stats && j == 0) Chunk<TO>::unpack(*reinterpret_cast<const u32x4*>(ewin + (ech << 4)), ksh[ii]);
#pragma unroll
          for (int ps = 0; ps < 32 / RPS; ++ps) {
            const int row = ps * RPS + erow;
            const u32x4 v = *reinterpret_cast<const u32x4*>(ewin + row * ERB + ((ech ^ ((row >> 1) & (NCH - 1))) << 4));
            if (do_stats) {
              float f[KPO];
              Chunk<TO>::unpack(v, f);
#pragma unroll
              for (int e = 0; e < KPO; ++e) {
                const float d = f[e] - ksh[ii][e];
                s1[ii][e] += d;
                s2[ii][e] = fmaf(d, d, s2[ii][e]);
              }
            }
#ifdef PFR_IGEMM_TRACE
            if (p.dbg & 2) { asm volatile("" ::"v"(v)); continue; }   // experiment: no output stores
#endif
            if (!p.pclass) {
              __builtin_amdgcn_raw_buffer_store_b128(v, yrsrc, voff0, (j * 32 + ps * RPS) * rowb + ii * EI * 32 * (int)sizeof(TO), 0);
            } else {
              uint32_t n_img, oh, ow;
              decode(mrow0 + j * 32 + row, mlim, ph, pw, n_img, oh, ow);
              const int mm = (int)((n_img * p.OH + oh) * p.OW + ow);
              __builtin_amdgcn_raw_buffer_store_b128(v, yrsrc, (mm * p.ldy + co0 + ii * EI * 32) * (int)sizeof(TO), 0, 0);
            }
          }
          asm volatile("" ::: "memory");
        }
      }
      if (do_stats) {
        const float inv = 1.f / (float)(BQ / WQ);
#pragma unroll
        for (int ii = 0; ii < NEI; ++ii) {
          float v[2 * KPO], kk[KPO];
#pragma unroll
          for (int e = 0; e < KPO; ++e) { v[e] = s1[ii][e]; v[KPO + e] = s2[ii][e]; kk[e] = ksh[ii][e]; }
          halving_sum<2 * KPO, LOWBIT>(v, lane);
          // lanes with bit 5 clear now hold s1 of their channels, lanes with bit 5 set s2 of the same channels: swap across
          constexpr int NLEFT = (2 * KPO) >> (6 - LOWBIT);
          const bool up = (lane >> 5) & 1;
          int ebase = 0;
#pragma unroll
          for (int b = 4; b >= LOWBIT; --b) ebase += ((lane >> b) & 1) * ((2 * KPO) >> (6 - b));
          // kk: pick the shift of the channels this lane ends up with (bits 4 … LOWBIT choose among KPO values)
          Halving<KPO, 4, LOWBIT>::pick(kk, lane);
          float* dstp = p.stats_part + ((size_t)(tm * WQ + wq) * 2) * p.Cout + co0 + ii * EI * 32;
#pragma unroll
          for (int k = 0; k < NLEFT; ++k) {
            const float other = __shfl_xor(v[k], 32, 64);
            const float a = up ? other : v[k], b2 = up ? v[k] : other;   // a = Σ d, b2 = Σ d²
            const float outv = up ? (b2 - a * a * inv) : (kk[k] + a * inv);
            dstp[(up ? p.Cout : 0) + ebase + k] = outv;
          }
        }
      }
#ifdef PFR_IGEMM_TRACE
      PSTAMP(c_epi);
      ++n_tiles;
#endif
      continue;
    }
    const bool post = p.bias || p.accumulate || p.out_relu || p.residual || p.act;
    const bool ld16ok = (p.ldy * (int)sizeof(TO)) % 16 == 0;
    float s1[NEI][KPO], s2[NEI][KPO], ksh[NEI][KPO];
#pragma unroll
    for (int ii = 0; ii < NEI; ++ii)
#pragma unroll
      for (int e = 0; e < KPO; ++e) { s1[ii][e] = 0.f; s2[ii][e] = 0.f; ksh[ii][e] = 0.f; }
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
#pragma unroll
      for (int ii = 0; ii < NEI; ++ii) {
        // phase 1: accumulators of EI 32x32 tiles -> staging window [32 rows m][EI*32 couts], 16-byte chunks swizzled by row
        {
          const int row = lane & 31;
          const int sw = (row >> 1) & (NCH - 1);
#pragma unroll
          for (int ie = 0; ie < EI; ++ie) {
            const int i = ii * EI + ie;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const int cb = (ie * 32 + 8 * qd + 4 * (lane >> 5)) * (int)sizeof(TO);   // byte column of 4 consecutive couts
              // The staging stores are issued as inline asm ON PURPOSE: for a compiler-visible LDS store hipcc drains every
              // LDS-DMA in flight first (`s_waitcnt vmcnt(0)`: it cannot prove that the DMA destination — the ring — and this
              // window do not overlap), which would empty the cross-tile prefetch ring at every tile.
              const uint32_t dst = ewin_lds + row * ERB + ((((cb >> 4) ^ sw)) << 4) + (cb & 15);
              if constexpr (sizeof(TO) == 4) {
                f32x4 v = {acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]};
                asm volatile("ds_write_b128 %0, %1" ::"v"(dst), "v"(v) : "memory");
              } else {
                bf16x4 v;
                v[0] = (bf16_t)acc[i][j][4 * qd];
                v[1] = (bf16_t)acc[i][j][4 * qd + 1];
                v[2] = (bf16_t)acc[i][j][4 * qd + 2];
                v[3] = (bf16_t)acc[i][j][4 * qd + 3];
                asm volatile("ds_write_b64 %0, %1" ::"v"(dst), "v"(__builtin_bit_cast(u32x2, v)) : "memory");
              }
            }
          }
        }
        asm volatile("" ::: "memory");   // (LDS is in order per wave: the reads below see the writes above)
        // phase 2: whole 16-byte row segments: post-ops / statistics / store
        const int co = n0 + wp * (BP / WP) + ii * EI * 32 + ech * KPO;
        const bool vec_ok = (co + KPO <= p.Cout) && ld16ok;
        float bia[KPO];
#pragma unroll
        for (int e = 0; e < KPO; ++e) bia[e] = (p.bias && co + e < p.Cout) ? p.bias[co + e] : 0.f;
        if (p.stats_part && j == 0) {
          // statistics are accumulated around a per-channel shift (row 0 of the wave's block): no E[x²]−E[x]² cancellation
          const u32x4 k0 = *reinterpret_cast<const u32x4*>(ewin + (ech << 4));   // row 0: swizzle term 0
          Chunk<TO>::unpack(k0, ksh[ii]);
        }
#pragma unroll
        for (int ps = 0; ps < 32 / RPS; ++ps) {
          const int row = ps * RPS + erow;
          const u32x4 raw = *reinterpret_cast<const u32x4*>(ewin + row * ERB + ((ech ^ ((row >> 1) & (NCH - 1))) << 4));
          int m = m0 + wq * (BQ / WQ) + j * 32 + row;
          if (m >= mlim || co >= p.Cout) continue;
          if (p.pclass) {
            uint32_t n_img, oh, ow;
            decode(m, mlim, ph, pw, n_img, oh, ow);
            m = (int)((n_img * p.OH + oh) * p.OW + ow);
          }
          u32x4 v = raw;
          float f[KPO];
          Chunk<TO>::unpack(v, f);
          char* dst = yb + ((size_t)m * p.ldy + co) * sizeof(TO);
          if (post) {
            if (p.residual) {
              float g[KPO];
              const char* rsrc = reinterpret_cast<const char*>(p.residual) + ((size_t)m * p.ldy + co) * sizeof(TO);
              if (vec_ok) {
                Chunk<TO>::unpack(ld16(rsrc), g);
              } else {
#pragma unroll
                for (int e = 0; e < KPO; ++e) g[e] = (co + e < p.Cout) ? to_f32(reinterpret_cast<const TO*>(rsrc)[e]) : 0.f;
              }
              if (p.res_mask) {
                const unsigned bits = p.res_mask[(size_t)m * (p.ldy / KPO) + co / KPO];
#pragma unroll
                for (int e = 0; e < KPO; ++e) g[e] = (bits >> e) & 1u ? g[e] : 0.f;
              }
#pragma unroll
              for (int e = 0; e < KPO; ++e) f[e] += g[e];
            }
            if (p.accumulate) {
              float g[KPO];
              if (vec_ok) {
                Chunk<TO>::unpack(ld16(dst), g);
              } else {
#pragma unroll
                for (int e = 0; e < KPO; ++e) g[e] = (co + e < p.Cout) ? to_f32(reinterpret_cast<TO*>(dst)[e]) : 0.f;
              }
#pragma unroll
              for (int e = 0; e < KPO; ++e) f[e] += g[e];
            }
#pragma unroll
            for (int e = 0; e < KPO; ++e) {
              f[e] += bia[e];
              if (p.out_relu) f[e] = fmaxf(f[e], 0.f);
            }
            v = Chunk<TO>::pack(f);
            Chunk<TO>::unpack(v, f);  // statistics see the value as stored
            if (p.act == 2) {         // exact (erf) GELU of the STORED pre-activation; both tensors are kept for backward
              st16(reinterpret_cast<char*>(p.y2) + ((size_t)m * p.ldy + co) * sizeof(TO), v);
#pragma unroll
              for (int e = 0; e < KPO; ++e) { float cdf, pdf; gelu_cdf_pdf(f[e], cdf, pdf); f[e] *= cdf; }
              v = Chunk<TO>::pack(f);
            } else if (p.act == 3) {  // GELU backward fused into the data-gradient GEMM: dz = dh ∘ gelu'(z)
              float z[KPO];
              Chunk<TO>::unpack(ld16(reinterpret_cast<const char*>(p.y2) + ((size_t)m * p.ldy + co) * sizeof(TO)), z);
#pragma unroll
              for (int e = 0; e < KPO; ++e) {
                float cdf, pdf;
          gelu_cdf_pdf(z[e], cdf, pdf);
                f[e] *= cdf + z[e] * pdf;
              }
              v = Chunk<TO>::pack(f);
            }
          }
          if (p.stats_part) {
#pragma unroll
            for (int e = 0; e < KPO; ++e) {
              const float d = f[e] - ksh[ii][e];
              s1[ii][e] += d;
              s2[ii][e] = fmaf(d, d, s2[ii][e]);
            }
          }
          if (vec_ok) {
            st16(dst, v);
          } else {
#pragma unroll
            for (int e = 0; e < KPO; ++e)
              if (co + e < p.Cout) reinterpret_cast<TO*>(dst)[e] = from_f32<TO>(f[e]);
          }
        }
        asm volatile("" ::: "memory");
      }
    }
    if (p.stats_part) {
      // lanes with equal (lane % NCH) hold partials of the same channels; one partial row per wave row (half tile)
      const int rows = min(BQ / WQ, mlim - (m0 + wq * (BQ / WQ)));
#pragma unroll
      for (int ii = 0; ii < NEI; ++ii) {
#pragma unroll
        for (int o = NCH; o < 64; o <<= 1)
#pragma unroll
          for (int e = 0; e < KPO; ++e) {
            s1[ii][e] += __shfl_xor(s1[ii][e], o, 64);
            s2[ii][e] += __shfl_xor(s2[ii][e], o, 64);
          }
        const int co = n0 + wp * (BP / WP) + ii * EI * 32 + ech * KPO;
        if (lane < NCH && rows > 0) {
          const float nt = (float)rows;
          float* dst = p.stats_part + ((size_t)(tm * WQ + wq) * 2) * p.Cout;
#pragma unroll
          for (int e = 0; e < KPO; ++e)
            if (co + e < p.Cout) {
              dst[co + e] = ksh[ii][e] + s1[ii][e] / nt;                      // block mean
              dst[p.Cout + co + e] = s2[ii][e] - s1[ii][e] * s1[ii][e] / nt;  // block M2 = Σ (x − mean)²
            }
        }
      }
    }
#ifdef PFR_IGEMM_TRACE
    PSTAMP(c_epi);
    ++n_tiles;
#endif
  }
#ifdef PFR_IGEMM_TRACE
  if (p.trace && tid == 0) {
    long long* o = p.trace + (size_t)b * 8;
    o[0] = w_start; o[1] = wall_clock64(); o[2] = (long long)c_wait; o[3] = (long long)c_mma; o[4] = (long long)c_epi;
    o[5] = n_tiles; o[6] = n_ks;
  }
#endif
}

// ------------------------------------------------------------------------------------------------
int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

int igemm_p_enabled() { return pfr_knob(KNOB_IGEMM_P); }
int igemm_p_forced_tile() { return pfr_knob(KNOB_IGEMM_PTILE); }

template <typename T, typename TO, int BQ, int BP>
static int launch_p(IgemmParams& p, hipStream_t st) {
  p.pclass = igemm_pclass_ok(p) ? 1 : 0;
  p.mclass = p.N * (p.OH / 2) * (p.OW / 2);
  p.tpc = (p.mclass + BQ - 1) / BQ;
  p.div_chw = make_fastdiv((uint32_t)((p.OH / 2) * (p.OW / 2) > 0 ? (p.OH / 2) * (p.OW / 2) : 1));
  p.div_cw = make_fastdiv((uint32_t)(p.OW / 2 > 0 ? p.OW / 2 : 1));
  // a 1x1 / stride-2 data gradient only has taps for class (0,0): when it ACCUMULATES into dx the other three classes
  // (which would add zeros) are not visited at all; otherwise they just store zeros (no k-steps)
  const int ncls = (p.pclass && p.R == 1 && p.S == 1 && p.accumulate && !p.bias && !p.residual && !p.out_relu && !p.act) ? 1 : 4;
  p.tilesM = p.pclass ? ncls * p.tpc : (p.M + BQ - 1) / BQ;
  p.tilesN = (p.Cout + BP - 1) / BP;
  const int total = p.tilesM * p.tilesN;
  const int grid = total < 2 * num_cus() ? total : 2 * num_cus();
  const dim3 g((unsigned)grid), blk(256);
  const bool k8 = pfr_knob(KNOB_IGEMM_PKCH) == 8   /* 128-byte k-steps when C allows */ && p.C % (8 * DT<T>::KPACK) == 0;
  if constexpr (sizeof(TO) == 2) {
    // whole tiles, no post-ops, 32-bit byte offsets: the lean epilogue
    const bool post = p.bias || p.accumulate || p.out_relu || p.residual || p.act;
    const int mrows = p.pclass ? p.mclass : p.M;
    const bool lean = !post && mrows % BQ == 0 && p.Cout % BP == 0 && (p.ldy * (int)sizeof(TO)) % 16 == 0 &&
                      (size_t)p.M * p.ldy * sizeof(TO) < ((size_t)1 << 31);
    if (lean) {
      const int g_p_pf = pfr_knob(KNOB_IGEMM_PPF);
      if (g_p_pf == 3) {
        hipLaunchKernelGGL((igemm_p_kernel<T, TO, BQ, BP, 4, 4, true, false, true>), g, dim3(320), 0, st, p, total);
      } else if (g_p_pf == 2 && p.C % (8 * DT<T>::KPACK) == 0) {
        // 128-byte k-steps, 4-slot ring of 32 KiB stages: one workgroup per CU
        const int grid1 = total < num_cus() ? total : num_cus();
        hipLaunchKernelGGL((igemm_p_kernel<T, TO, BQ, BP, 8, 4, true, true>), dim3((unsigned)grid1), blk, 0, st, p, total);
      } else if (g_p_pf >= 1) hipLaunchKernelGGL((igemm_p_kernel<T, TO, BQ, BP, 4, 4, true, true>), g, blk, 0, st, p, total);
      else if (k8) hipLaunchKernelGGL((igemm_p_kernel<T, TO, BQ, BP, 8, 2, true>), g, blk, 0, st, p, total);
      else hipLaunchKernelGGL((igemm_p_kernel<T, TO, BQ, BP, 4, 4, true>), g, blk, 0, st, p, total);
      PFR_CHECK_LAUNCH();
      return PFR_OK;
    }
  }
  hipLaunchKernelGGL((igemm_p_kernel<T, TO, BQ, BP, 4, 4, false>), g, blk, 0, st, p, total);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

template <typename T, typename TO>
static int launch_p_tile(IgemmParams& p, int bq, int bp, hipStream_t st) {
  if (bq == 128 && bp == 128) return launch_p<T, TO, 128, 128>(p, st);
  if (bq == 64 && bp == 128) return launch_p<T, TO, 64, 128>(p, st);
  if (bq == 128 && bp == 64) return launch_p<T, TO, 128, 64>(p, st);
  return launch_p<T, TO, 64, 64>(p, st);
}

// eligibility: k-step-uniform taps (C a multiple of the k-step), no fused operand prologue, no top-K filter epilogue
int igemm_p_launch(IgemmParams& p, int dtype, int out_dtype, int bq, int bp, hipStream_t st) {
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  if (p.C % (4 * kp) != 0 || p.pro_scale || p.act == 4) return 1;
  if (dtype == PFR_BF16) {
    if (out_dtype == PFR_BF16) return launch_p_tile<bf16_t, bf16_t>(p, bq, bp, st);
    return launch_p_tile<bf16_t, float>(p, bq, bp, st);
  }
  return launch_p_tile<float, float>(p, bq, bp, st);
}
