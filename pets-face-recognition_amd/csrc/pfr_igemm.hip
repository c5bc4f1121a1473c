// pfr_igemm.hip — NHWC implicit-GEMM convolution on MFMA (forward, data-gradient, plain GEMM).
//
// Replaces, on the reference's hot path, every `nn.Conv2d` / `nn.Linear` / `F.linear` forward call and the
// autograd input-gradient of the same ops (torchvision resnet50 built at /root/reference
// configs/dog_fe/fe_dogs_config.py:102-103; `F.linear` at losses/large_margin.py:71).
//
//   y[m][co] = sum_{r,s,c} act(x)[n, oh*ostride - pad + r, ow*ostride - pad + s, c] * w[co][r][s][c]
//   m = (n*OH + oh)*OW + ow,  act = identity or the fused BatchNorm-apply(+ReLU) prologue of the PRODUCER layer.
//
// The data gradient of a stride-u conv is the same kernel run over dy with `idil_log2 = log2(u)` (input dilation:
// only taps whose dilated coordinate is a multiple of u exist) and tap-flipped, channel-transposed weights.
//
// MI355X mapping: 256-thread workgroups (4 waves, 2x2), 128x128 / 128x64 / 64x128 / 64x64 output tiles,
// v_mfma_f32_32x32x16_bf16 (or the exact-f32 v_mfma_f32_32x32x2_f32 for the parity path), operand tiles gathered
// straight from NHWC HBM in 16-byte channel chunks into double-buffered LDS (80-byte padded rows), accumulators
// transposed through LDS in the epilogue so that HBM stores are whole 16-byte row segments, per-channel
// sum / sum-of-squares partials for the FOLLOWING train-mode BatchNorm produced from the same LDS tile
// (deterministic: one partial row per m-tile, no atomics), XCD-aware tile order (n-tiles of one m-tile adjacent).
#include "pfr_igemm.h"
#include <stdlib.h>

#ifdef PFR_IGEMM_TRACE
static long long* g_igemm_trace = nullptr;
static int g_igemm_dbg = 0;
extern "C" void pfr_debug_igemm_trace(void* buf) { g_igemm_trace = (long long*)buf; }
extern "C" void pfr_debug_igemm_flags(int f) { g_igemm_dbg = f; }
#define TSTAMP(i) do { if (p.trace && threadIdx.x == 0) p.trace[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define TSTAMP(i) do {} while (0)
#endif

// tile geometry knobs: chunks of 16 B per LDS row per k-step, and LDS ring depth
#ifndef PFR_IGEMM_KCH
#define PFR_IGEMM_KCH 8
#endif
#ifndef PFR_IGEMM_OCC4
#define PFR_IGEMM_OCC4 4   // resident workgroups per CU requested for the 64-byte-row 4-wave tiles (caps VGPR+AGPR at 128)
#endif
#ifndef PFR_IGEMM_NST
#define PFR_IGEMM_NST 2
#endif

template <typename T, typename TO, int BQ, int BP, bool PRO, bool FAST, int KCH_, int NW, int WP, int NST_, bool FILT = false, bool BNB = false, bool EPRE = false>
__global__ __launch_bounds__(NW * 64, (KCH_ == 4 && NW == 4 && NST_ == 2 && sizeof(T) == 2 && sizeof(TO) == 2 && !PRO) ? (BNB ? 2 : PFR_IGEMM_OCC4) : 1) void igemm_kernel(IgemmParams p) {
  constexpr int KP = DT<T>::KPACK;
  constexpr int KCH = KCH_;                    // 16-byte chunks per LDS row per k-step (4: 64-B rows, 8: 128-B rows)
  constexpr int ROWB = KCH * 16;
  constexpr int BK = KCH * KP;                 // k elements per k-step
  constexpr int RPI = 64 / KCH;                // tile rows covered by one wave-wide DMA instruction
  constexpr int NT = NW * 64;                  // threads: NW waves in a WP x WQ grid over (couts, rows)
  constexpr int WQ = NW / WP;
  constexpr int TP = BP / (WP * 32), TQ = BQ / (WQ * 32);  // 32x32 accumulator tiles per wave
  constexpr int QCH = BQ / (NW * RPI), PCH = BP / (NW * RPI);  // DMA instructions per thread per operand per k-step
  static_assert(TP >= 1 && TQ >= 1 && QCH >= 1 && PCH >= 1, "tile too small for this wave grid");
  constexpr int NLD = QCH + PCH;
  constexpr int NST = NST_;                    // LDS ring depth: loads are issued NST-1 k-steps ahead of their use
  static_assert(NST >= 2 && NST <= 4 && (!PRO || NST <= 3), "unsupported ring depth");
  constexpr int STAGE = (BP + BQ) * ROWB;
  constexpr int KPO = 16 / (int)sizeof(TO);
  constexpr int OROWB = BP * (int)sizeof(TO) + 16;
  constexpr int EPI = BQ * OROWB;
  constexpr int RED = NW * BP * 2 * 4;
  constexpr int PROB = PRO ? 2 * 2048 * 4 : 0;  // fused-prologue coefficients (scale, shift) of up to 2048 channels
  // FILT (top-K filter epilogue, act 4): scores are compared straight from the accumulators — no transpose buffer
  // FILT: + two winner queues of one entry per thread (u64 key|~column and the query row) and their counters: see "deferred append" below
  constexpr int FQCAP = NW * 64, FQOFF = NST * STAGE, FQBYTES = FILT ? 2 * FQCAP * 12 + 16 : 0;
  constexpr int SMEM = FILT ? NST * STAGE + FQBYTES : ((NST * STAGE + PROB > EPI + RED) ? NST * STAGE + PROB : EPI + RED);
  static_assert(SMEM <= 160 * 1024, "tile does not fit the 160 KiB LDS");
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / WQ, wq = wave % WQ;
  TSTAMP(0);

  // FILT launches are persistent (one workgroup per CU walks the tiles: a 256x256x512 match tile is 8 k-steps, and a fresh workgroup
  // per tile left a launch gap after every one of them); all other launches have one tile per workgroup and run the body once.
  const uint32_t ntile = (uint32_t)(p.tilesM * p.tilesN);
  // tile of this workgroup at iteration `it` (FILT: persistent; otherwise one iteration).  -> false: past the end; tm_ < 0: an idle slot
  // of the blocked order (ragged group), skipped
  auto tile_at = [&](uint32_t it, int& tm_, int& tn_) -> bool {
    if (FILT && p.fo_qg > 0) {
      const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
      const int n_lo = (int)((long)xcd * p.tilesN / 8), n_hi = (int)((long)(xcd + 1) * p.tilesN / 8);
      const int ngg = (p.tilesN + 7) / 8 + p.fo_gg - 1;          // gallery groups per XCD, the same count on every XCD (ragged tails idle)
      const int ng = ngg / p.fo_gg, nq = (p.tilesM + p.fo_qg - 1) / p.fo_qg;
      if (it >= (uint32_t)(ng * nq)) return false;
      const int q = it / ng, j = it - q * ng;
      tm_ = q * p.fo_qg + slot / p.fo_gg;
      tn_ = n_lo + j * p.fo_gg + slot % p.fo_gg;
      if (tm_ >= p.tilesM || tn_ >= n_hi) tm_ = -1;
      return true;
    }
    const uint32_t tt_ = blockIdx.x + it * gridDim.x;
#ifdef PFR_IGEMM_TRACE
    if (!FILT && (p.dbg & 16)) {   // experiment: XCDs 0-3 only (launch_tile_k doubled the grid)
      const uint32_t xcd = blockIdx.x & 7, t4 = (blockIdx.x >> 3) * 4 + xcd;
      if (xcd >= 4 || t4 >= ntile) return false;
      tn_ = t4 % p.tilesN;
      tm_ = t4 / p.tilesN;
      return true;
    }
#endif
    if (tt_ >= ntile) return false;
    const uint32_t t_ = FILT ? tt_ : xcd_remap(blockIdx.x, gridDim.x);
    tn_ = t_ % p.tilesN;
    tm_ = t_ / p.tilesN;
    return true;
  };
  bool prefetched = false;   // FILT: k-step 0 of this tile went out under the previous tile's epilogue
  // FILT, deferred append.  A winner needs a slot of its query's candidate list (one returning global atomic) before it can be written:
  // done in the epilogue that found it, that dependent round trip ended every tile (1.9 of 20 us: every tile has a wave with a winner, and the
  // next tile's first barrier waits for it).  Instead the epilogue of tile t only PUSHES its winners into LDS queue t & 1; right after the first
  // barrier of tile t + 1 (every wave is past that epilogue) thread i takes entry i and issues the atomic; the entry is written to the list after
  // the first barrier of tile t + 2, when the slot has long arrived — both under a k-loop.  The queues drain after the last tile.  (The order of
  // a candidate list is irrelevant: pfr_topk_merge sorts by key.)
  unsigned long long* const fq_key = reinterpret_cast<unsigned long long*>(smem + FQOFF);           // [2][FQCAP]
  int* const fq_row = reinterpret_cast<int*>(smem + FQOFF + 2 * FQCAP * 8);                           // [2][FQCAP]
  int* const fq_cnt = reinterpret_cast<int*>(smem + FQOFF + 2 * FQCAP * 12);                          // [2] pushes attempted (may exceed FQCAP)
  int fq_slot = -1;          // slot of this thread's entry of queue (ft & 1) (atomic in flight / arrived), -1: none
  uint32_t ft = 0;           // tiles this workgroup has started
  if constexpr (FILT) {
    if (tid < 2) fq_cnt[tid] = 0;   // (published by the first tile's first barrier, long before its epilogue pushes)
  }
  // after the first barrier of a tile (or at the drain): write the entries whose slot has arrived, take slots for the newer queue
  auto fq_step = [&]() {
    if constexpr (FILT) {
      const int b = (int)(ft & 1);
      unsigned long long* const cand = reinterpret_cast<unsigned long long*>(p.y);
      if (fq_slot >= 0) {
        if (fq_slot < p.cap) cand[(size_t)fq_row[b * FQCAP + tid] * p.cap + fq_slot] = fq_key[b * FQCAP + tid];
        fq_slot = -1;
      }
      const int n = min(fq_cnt[b ^ 1], FQCAP);
      if (tid < n) fq_slot = atomicAdd(&p.ccnt[fq_row[(b ^ 1) * FQCAP + tid]], 1);
    }
  };
  for (uint32_t it = 0;; ++it) {
  int tm, tn;
  if (!tile_at(it, tm, tn)) {
    if constexpr (FILT) {   // drain the winner queues: the last tile's entries have no slot yet, the one before's no write
      __syncthreads();
      fq_step();
      ++ft;
      if (fq_slot >= 0 && fq_slot < p.cap)
        reinterpret_cast<unsigned long long*>(p.y)[(size_t)fq_row[(ft & 1) * FQCAP + tid] * p.cap + fq_slot] = fq_key[(ft & 1) * FQCAP + tid];
    }
    TSTAMP(6);
    break;
  }
  if (tm < 0) continue;
  TSTAMP(5);
  if (!FILT && it > 0) break;
  const int n0 = tn * BP;
  const int cls = p.pclass ? tm / p.tpc : 0;
  const int ph = cls >> 1, pw = cls & 1;
  const int m0 = (p.pclass ? tm - cls * p.tpc : tm) * BQ;   // first row of the tile (within its parity class, if any)
  const int mlim = p.pclass ? p.mclass : p.M;
  // tile row -> (image, oh, ow); returns false past the end
  auto decode = [&](int m, uint32_t& n_img, uint32_t& oh, uint32_t& ow) -> bool {
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

  // staging geometry: wave w, pass j covers tile rows (j*4 + w)*RPI .. +RPI, lane l -> row l/KCH, physical chunk l%KCH;
  // the logical (k) chunk it must fetch is phys ^ swizzle(row), which is the same for every pass j.
  const int rsub = lane / KCH;
  const int lc = (lane % KCH) ^ row_swizzle<KCH>(wave * RPI + rsub);

  // ---- per-row gather state for the activation (Q) operand
  int ihb[QCH], iwb[QCH], pixb[QCH];
#pragma unroll
  for (int j = 0; j < QCH; ++j) {
#ifdef PFR_IGEMM_TRACE
    const int m = ((p.dbg & 1) ? 0 : m0) + (j * NW + wave) * RPI + rsub;
#else
    const int m = m0 + (j * NW + wave) * RPI + rsub;
#endif
    uint32_t n_img, oh, ow;
    if (decode(m, n_img, oh, ow)) {
      ihb[j] = (int)oh * p.ostride - p.pad;
      iwb[j] = (int)ow * p.ostride - p.pad;
      pixb[j] = n_img * p.H * p.W;
    } else {
      ihb[j] = -(1 << 28);
      iwb[j] = -(1 << 28);
      pixb[j] = 0;
    }
  }
  // ---- k-chunk -> (tap r, tap s, channel c) walker of this thread's logical chunk column
  int kel = lc * KP;
  int tap = kel / p.C;
  int c = kel - tap * p.C;
  int tr = tap / p.S;
  int ts = tap - tr * p.S;

  const int dmask = (1 << p.idil_log2) - 1;
  const uint32_t OOB = 0xFFFFFF00u;  // beyond num_records: the buffer load returns (and the LDS-DMA writes) zeros
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.x), 0, (int)((size_t)p.N * p.H * p.W * p.C * sizeof(T)), 0x00020000);
  __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((size_t)p.Cout * p.K * sizeof(T)), 0x00020000);

  float* pcoef = reinterpret_cast<float*>(smem + NST * STAGE);  // [2][C] scale, shift (PRO only)
  if constexpr (PRO) {
    for (int i = tid; i < p.C; i += NT) {
      pcoef[i] = p.pro_scale[i];
      pcoef[p.C + i] = p.pro_shift[i];
    }
  }

  // validity masks / channel offsets of the (up to two) k-steps in flight, consumed by the prologue fix-up
  uint32_t okA = 0, okB = 0;
  int cA = 0, cB = 0;

  // FAST path (C % BK == 0, every conv but the stem): within a k-step the tap (r,s) is uniform over the workgroup, so
  // the per-row gather offsets are recomputed only when the tap changes (every C/BK k-steps, select-based, no divergent
  // branches) and a k-step costs one v_add per DMA.  Invalid rows carry an offset that stays out of range.
  const uint32_t OOBB = 0xF0000000u;
  uint32_t qbase[QCH], wbase[PCH];
  uint32_t okcur = 0;
  int u_tr = 0, u_ts = 0, cbyte = 0;  // uniform (scalar) walker state
  auto newtap = [&]() {
    okcur = 0;
#pragma unroll
    for (int j = 0; j < QCH; ++j) {
      int ih = ihb[j] + u_tr, iw = iwb[j] + u_ts;
      bool ok = (((ih | iw) & dmask) == 0);
      ih >>= p.idil_log2;
      iw >>= p.idil_log2;
      ok = ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W) && (u_tr < p.R);
      const uint32_t off = (uint32_t)(((pixb[j] + ih * p.W + iw) * p.C + lc * KP) * (int)sizeof(T));
      qbase[j] = ok ? off : OOBB;
      okcur |= ok ? (1u << j) : 0u;
    }
  };
  // tap walk: all taps (step 1), or — in parity-class mode — only the taps whose dilated coordinate is even for this class
  const int tstep = p.pclass ? 2 : 1;
  const int tr0 = p.pclass ? ((p.pad + ph) & 1) : 0, ts0 = p.pclass ? ((p.pad + pw) & 1) : 0;
  int nk_all = (p.K + BK - 1) / BK;
  int tapbyte = 0;  // byte offset of the current tap inside a weight row
  if constexpr (FAST) {
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      const int row = n0 + (j * NW + wave) * RPI + rsub;
      wbase[j] = row < p.Cout ? (uint32_t)(((size_t)row * p.K + lc * KP) * sizeof(T)) : OOBB;
    }
    if (p.pclass) {
      const int ntr = (p.R - tr0 + 1) / 2, nts = (p.S - ts0 + 1) / 2;
      nk_all = ntr * nts * (p.C / BK);
    }
    u_tr = tr0;
    u_ts = ts0;
    if (!FILT && p.krot && !p.pclass) {
      // rotated k-loop: this workgroup walks the k-steps kt0, kt0 + 1, ..., nk - 1, 0, ..., kt0 - 1 — concurrently running workgroups then
      // request DIFFERENT weight columns / taps at any instant instead of all the same ones (fp32 summation order differs per tile)
      const int spt = p.C / BK;                                   // k-steps per tap
      const int kt0 = (int)(((uint32_t)(tm * p.tilesN + tn) * (uint32_t)p.krot) % (uint32_t)nk_all);
      const int tap0 = kt0 / spt;
      cbyte = (kt0 - tap0 * spt) * BK * (int)sizeof(T);
      u_tr = tap0 / p.S;
      u_ts = tap0 - u_tr * p.S;
    }
    tapbyte = (u_tr * p.S + u_ts) * p.C * (int)sizeof(T);
    newtap();
  }

  // issues the LDS-DMA of one k-step into ring slot `buf` (weights + gathered activations, zero-filled when invalid)
  // (FAST) DMA instructions [lo, hi) of the NLD a wave issues per k-step: weights first, then activations; the walker does not move
  auto gload_part = [&](int buf, int lo, int hi) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < PCH; ++j)
      if (j >= lo && j < hi)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(base + (j * NW + wave) * RPI * ROWB),
                                                 16, (int)(wbase[j] + (uint32_t)(tapbyte + cbyte)), 0, 0, 0);
#pragma unroll
    for (int j = 0; j < QCH; ++j)
      if (PCH + j >= lo && PCH + j < hi)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(base + (BP + (j * NW + wave) * RPI) * ROWB),
                                                 16, (int)(qbase[j] + (uint32_t)cbyte), 0, 0, 0);
  };
  auto gload = [&](int buf, bool issue = true) {   // issue = false: the k-step's DMA is already in flight (FILT / spread issue), only the walker steps
    char* base = smem + buf * STAGE;
    if constexpr (FAST) {
#ifdef PFR_IGEMM_TRACE
      if (!(p.dbg & 8))   // experiment: no weight staging at all (halves the L2->LDS fill volume of a 128x128 tile)
#endif
      if (issue) {
#pragma unroll
      for (int j = 0; j < PCH; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(base + (j * NW + wave) * RPI * ROWB),
                                                 16, (int)(wbase[j] + (uint32_t)(tapbyte + cbyte)), 0, 0, 0);
      }
#ifdef PFR_IGEMM_TRACE
      if (!(p.dbg & 4) || tapbyte == 0)
#endif
      if (issue) {
#pragma unroll
      for (int j = 0; j < QCH; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(base + (BP + (j * NW + wave) * RPI) * ROWB),
                                                 16, (int)(qbase[j] + (uint32_t)cbyte), 0, 0, 0);
      }
      okA = okB; cA = cB;
      okB = okcur; cB = cbyte / (int)sizeof(T) + lc * KP;
      cbyte += BK * (int)sizeof(T);
      if (cbyte >= p.C * (int)sizeof(T)) {
        cbyte = 0;
        u_ts += tstep;
        if (u_ts >= p.S) { u_ts = ts0; u_tr += tstep; if (!FILT && p.krot && !p.pclass && u_tr >= p.R) u_tr = 0; }
        tapbyte = (u_tr * p.S + u_ts) * p.C * (int)sizeof(T);
        newtap();
      }
      return;
    }
    const bool kok = kel < p.K;
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      const int row = n0 + (j * NW + wave) * RPI + rsub;
      const uint32_t off = (kok && row < p.Cout) ? (uint32_t)(((size_t)row * p.K + kel) * sizeof(T)) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(base + (j * NW + wave) * RPI * ROWB),
                                               16, (int)off, 0, 0, 0);
    }
    uint32_t okm = 0;
#pragma unroll
    for (int j = 0; j < QCH; ++j) {
      int ih = ihb[j] + tr, iw = iwb[j] + ts;
      bool ok = kok && (((ih | iw) & dmask) == 0);
      ih >>= p.idil_log2;
      iw >>= p.idil_log2;
      ok = ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
      const uint32_t off = ok ? (uint32_t)(((size_t)(pixb[j] + ih * p.W + iw) * p.C + c) * sizeof(T)) : OOB;
      if (ok) okm |= 1u << j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(base + (BP + (j * NW + wave) * RPI) * ROWB),
                                               16, (int)off, 0, 0, 0);
    }
    okA = okB; cA = cB;
    okB = okm; cB = c;
    // advance to the next k-step
    kel += BK;
    c += BK;
    while (c >= p.C) {
      c -= p.C;
      if (++ts == p.S) { ts = 0; ++tr; }
    }
  };

  // fused prologue: BN-apply(+ReLU) IN PLACE on this thread's own (already landed) activation chunks of ring slot `buf`;
  // padding / out-of-range chunks stay exactly zero.  (okm, cc) describe that k-step.
  auto fixup = [&](int buf, uint32_t okm, int cc) {
    if constexpr (PRO) {
      char* base = smem + buf * STAGE;
      float sc[KP], sh[KP];
#pragma unroll
      for (int e = 0; e < KP; e += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(pcoef + cc + e);
        const f32x4 b = *reinterpret_cast<const f32x4*>(pcoef + p.C + cc + e);
#pragma unroll
        for (int u = 0; u < 4; ++u) { sc[e + u] = a[u]; sh[e + u] = b[u]; }
      }
#pragma unroll
      for (int j = 0; j < QCH; ++j) {
        if (okm & (1u << j)) {
          char* ptr = base + (BP + (j * NW + wave) * RPI + rsub) * ROWB + (lane % KCH) * 16;
          float f[KP];
          Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(ptr), f);
#pragma unroll
          for (int e = 0; e < KP; ++e) {
            const float z = fmaf(f[e], sc[e], sh[e]);
            f[e] = p.pro_relu ? fmaxf(z, 0.f) : z;
          }
          st16(ptr, Chunk<T>::pack(f));
        }
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

  // FILT: this lane's thresholds (the keys of its query rows' current K-th best) are requested HERE, a whole k-loop before the epilogue compares
  // against them: loaded in the epilogue they were a memory round trip at the end of every tile (1 us of 20).  A threshold that a concurrent
  // pfr_topk_merge raises in the meantime is only a looser, still valid bound.
  uint32_t tks[FILT ? TQ : 1];
  if constexpr (FILT) {
    const uint32_t* thrk = reinterpret_cast<const uint32_t*>(p.y2);
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      const int m = m0 + wq * (BQ / WQ) + j * 32 + (lane & 31);
      tks[j] = m < p.M ? thrk[m] : 0xFFFFFFFFu;
    }
  }

  // ---- NST-slot ring with counted waits: the DMA of k-steps kt+2 … kt+NST-1 stays in flight across the barrier that
  //      publishes kt+1 (raw s_barrier: no implicit vmcnt(0) drain); NST = 2 is the classic double buffer.
  auto wait_pending = [&](int tiles) {  // wait until at most `tiles` k-steps of DMA remain outstanding
    if (NST >= 4 && tiles >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
    else if (NST >= 3 && tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  const int nk = nk_all;
  const int npre = nk < NST - 1 ? nk : NST - 1;
  TSTAMP(1);
  for (int s0 = 0; s0 < npre; ++s0) gload(s0, !(FILT && s0 == 0 && prefetched));   // (FILT: k-step 0 of a later tile went out under the previous tile's epilogue)
  prefetched = false;
  wait_pending(npre - 1);
  if constexpr (PRO) {
    __syncthreads();  // coefficients visible
    if (npre > 1) fixup(0, okA, cA); else fixup(0, okB, cB);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  TSTAMP(2);
  const int fb = (int)(ft & 1);   // FILT: the winner queue this tile's epilogue pushes into
  if constexpr (FILT) {
    fq_step();
    if (tid == 0) fq_cnt[fb] = 0;   // (its old entries were written just above; the pushes come after this tile's k-loop barriers)
    ++ft;
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int slot = kt % NST;
    const char* base = smem + slot * STAGE;
#ifdef PFR_GLOAD_FIRST
    if (kt + NST - 1 < nk) gload((kt + NST - 1) % NST);
    mma_kstep_sw<T, KCH, TP, TQ>(base + (wp * (BP / WP)) * ROWB, base + (BP + wq * (BQ / WQ)) * ROWB, lane, acc);
#elif defined(PFR_IGEMM_SPREAD)
    // experiment (round 6): the next k-step's NLD DMA instructions leave in `np` parts, one behind each of the k-groups kg0 .. kg0 + np - 1,
    // instead of one burst of NLD instructions (each holds its wave for 60-180 cycles).  dma_sched = np | kg0(waves 0-3) << 4 | kg0(waves 4-7) << 8
    if constexpr (FAST && !PRO && !FILT && KCH == 8) {
      const int np = p.dma_sched & 15, kg0 = (NW == 8 && (threadIdx.x >> 8)) ? ((p.dma_sched >> 8) & 15) : ((p.dma_sched >> 4) & 15);
      const bool more = kt + NST - 1 < nk;
      mma_kstep_sw<T, KCH, TP, TQ>(base + (wp * (BP / WP)) * ROWB, base + (BP + wq * (BQ / WQ)) * ROWB, lane, acc, every_kg([&](int kg) {
        const int rel = kg - kg0;
        if (more && rel >= 0 && rel < np) {
          gload_part((kt + NST - 1) % NST, rel * NLD / np, (rel + 1) * NLD / np);
          if (rel == np - 1) gload((kt + NST - 1) % NST, false);
        }
      }));
    } else {
      mma_kstep_sw<T, KCH, TP, TQ>(base + (wp * (BP / WP)) * ROWB, base + (BP + wq * (BQ / WQ)) * ROWB, lane, acc, [&]() {
        if (kt + NST - 1 < nk) gload((kt + NST - 1) % NST);
      }, (NW == 8 && (threadIdx.x >> 8)) ? 1 : 0);
    }
#else
    // the next tile's DMA is issued behind the first k-group's MFMAs (see mma_kstep_sw) — behind the SECOND by waves 4-7 of an
    // 8-wave tile: waves w and w + 4 share a SIMD, an LDS-DMA instruction holds its wave for 60-180 cycles, and with both of them
    // in their DMA burst the SIMD's matrix pipe stands still (3x3 256 ch at 14x14: 68.2 -> 63.9 us; one k-group later again, or
    // two, the DMA lands after the k-step's barrier: 71 / 75 us; profiles/r04_tile_variants.txt)
    mma_kstep_sw<T, KCH, TP, TQ>(base + (wp * (BP / WP)) * ROWB, base + (BP + wq * (BQ / WQ)) * ROWB, lane, acc, [&]() {
      if (kt + NST - 1 < nk) gload((kt + NST - 1) % NST);
    }, (NW == 8 && (threadIdx.x >> 8)) ? ((p.dma_sched >> 8) & 15) : ((p.dma_sched >> 4) & 15));
#endif
    if (kt + 1 < nk) {
      const int last = kt + NST - 1 < nk - 1 ? kt + NST - 1 : nk - 1;  // newest k-step in flight
      const int pending = last - (kt + 1);
      wait_pending(pending);
      if constexpr (PRO) {
        if (pending >= 1) fixup((kt + 1) % NST, okA, cA); else fixup((kt + 1) % NST, okB, cB);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  TSTAMP(3);
  if constexpr (FILT) {
    // ---- top-K filter epilogue (gallery match): a lane owns query row m of each 32x32 tile and 16 gallery columns of it
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(p.y);
    // the first k-step of this workgroup's NEXT tile goes out now, under the compare loops (every wave is past the k-loop's last
    // barrier: ring slot 0 is free)
    if constexpr (FAST) {
      int tmx, tnx;
      if (tile_at(it + 1, tmx, tnx) && tmx >= 0) {
        prefetched = true;
        const int n0n = tnx * BP, m0n = tmx * BQ;
#pragma unroll
        for (int j = 0; j < PCH; ++j) {
          const int row = n0n + (j * NW + wave) * RPI + rsub;
          const uint32_t off = row < p.Cout ? (uint32_t)(((size_t)row * p.K + lc * KP) * sizeof(T)) : OOBB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(smem + (j * NW + wave) * RPI * ROWB), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < QCH; ++j) {
          const int m = m0n + (j * NW + wave) * RPI + rsub;
          const uint32_t off = m < p.M ? (uint32_t)(((size_t)m * p.C + lc * KP) * sizeof(T)) : OOBB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(smem + (BP + (j * NW + wave) * RPI) * ROWB), 16, (int)off, 0, 0, 0);
        }
      }
    }
    // Fast path (round 5; the common case of every fused chunk): the tile's columns all exist, nothing is excluded and every threshold
    // of this wave is the key of a POSITIVE score — for those the float order IS the key order, so the exact test is ONE compare per score
    // (the generic loop below spends ~10 vector instructions per score on key conversion and the column tests before it may branch).
    // Winners are counted first; a (lane, row) then takes all its slots with ONE atomic (the per-winner atomics were ~28 serialised
    // round trips per wave and tile) and writes them in a second pass.  Candidate ORDER inside a list is irrelevant (pfr_topk_merge sorts
    // by key).  Rows past M carry the key 0xFFFFFFFF = NaN as a float: no score compares greater.
    bool fastw = n0 + BP <= p.Cout && !p.self_excl;
#pragma unroll
    for (int j = 0; j < TQ; ++j) fastw = fastw && __all(tks[j] > 0x80000000u);
    if (fastw) {
      // per 32x32 accumulator block: the lane's maximum of its 16 scores first (8 three-input maxima instead of 16 compare + add pairs); only a
      // block in which SOME lane of the wave has a winner (one in five in a late chunk) runs the count / push code at all — with the whole
      // row range of a lane in one pass, 83 % of the waves walked all 128 guarded pushes of it
#pragma unroll
      for (int j = 0; j < TQ; ++j) {
        const int m = m0 + wq * (BQ / WQ) + j * 32 + (lane & 31);
        const float thr = fkey_inv(tks[j]);
#pragma unroll
        for (int i = 0; i < TP; ++i) {
          float mx = acc[i][j][0];
#pragma unroll
          for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[i][j][r]);
          if (!__any(mx > thr)) continue;
          if (mx > thr) {
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) cnt += acc[i][j][r] > thr ? 1 : 0;
#ifdef PFR_FILT_IMMEDIATE
            int slot = atomicAdd(&p.ccnt[m], cnt);
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (acc[i][j][r] > thr) {
                const int col = n0 + wp * (BP / WP) + i * 32 + acc_row(r, lane);
                if (slot < p.cap) cand[(size_t)m * p.cap + slot] = ((unsigned long long)fkey(acc[i][j][r]) << 32) | (uint32_t)(~(uint32_t)(p.col0 + col));
                ++slot;
              }
#else
            int pos = atomicAdd(&fq_cnt[fb], cnt);   // LDS: this lane's places in the tile's winner queue
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (acc[i][j][r] > thr) {
                const int col = n0 + wp * (BP / WP) + i * 32 + acc_row(r, lane);
                const unsigned long long e = ((unsigned long long)fkey(acc[i][j][r]) << 32) | (uint32_t)(~(uint32_t)(p.col0 + col));
                if (pos < FQCAP) {
                  fq_key[fb * FQCAP + pos] = e;
                  fq_row[fb * FQCAP + pos] = m;
                } else {                             // queue full (a candidate-rich early segment): appended at once
                  const int slot = atomicAdd(&p.ccnt[m], 1);
                  if (slot < p.cap) cand[(size_t)m * p.cap + slot] = e;
                }
                ++pos;
              }
#endif
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      const int m = m0 + wq * (BQ / WQ) + j * 32 + (lane & 31);
      if (m >= p.M) continue;
      const uint32_t tk = tks[j];
#pragma unroll
      for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int col = n0 + wp * (BP / WP) + i * 32 + acc_row(r, lane);
          const uint32_t key = fkey(acc[i][j][r]);
          if (key > tk && col < p.Cout && !(p.self_excl && p.col0 + col == m)) {
            const int slot = atomicAdd(&p.ccnt[m], 1);
            if (slot < p.cap) cand[(size_t)m * p.cap + slot] = ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)(p.col0 + col));
          }
        }
    }
    continue;   // (every wave is past the k-loop's last barrier: the ring is free for the next tile's DMA)
  }
  // ---- epilogue phase 1: accumulators -> LDS tile [BQ rows m][BP couts] of TO
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      const int mrow = wq * (BQ / WQ) + j * 32 + (lane & 31);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int co = wp * (BP / WP) + i * 32 + 8 * qd + 4 * (lane >> 5);
        char* dst = smem + mrow * OROWB + co * (int)sizeof(TO);
        if constexpr (sizeof(TO) == 4) {
          f32x4 v = {acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]};
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
          bf16x4 v;
          v[0] = (bf16_t)acc[i][j][4 * qd];
          v[1] = (bf16_t)acc[i][j][4 * qd + 1];
          v[2] = (bf16_t)acc[i][j][4 * qd + 2];
          v[3] = (bf16_t)acc[i][j][4 * qd + 3];
          *reinterpret_cast<bf16x4*>(dst) = v;
        }
      }
    }
  __syncthreads();
  TSTAMP(4);

  // ---- epilogue phase 2: row-major 16-byte chunks: bias / accumulate / relu / BN partial sums / store
  constexpr int CPR = BP * (int)sizeof(TO) / 16;  // chunks per output row
  constexpr int RPP = NT / CPR;                   // rows per pass
  static_assert(CPR <= 64, "statistics reduction assumes <= 64 chunks per row");
  const int oc = tid % CPR, rl = tid / CPR;
  const int co = n0 + oc * KPO;
  float s1[KPO], s2[KPO], bia[KPO], kshift[KPO];
  // statistics are accumulated around a per-channel shift K (the tile's first row) so that the tile variance does not
  // suffer the E[x²]−E[x]² cancellation; each tile publishes (mean_t, M2_t) and pfr_bn_finalize merges them (Chan).
  Chunk<TO>::unpack(*reinterpret_cast<const u32x4*>(smem + oc * 16), kshift);
#pragma unroll
  for (int e = 0; e < KPO; ++e) {
    s1[e] = 0.f;
    s2[e] = 0.f;
    bia[e] = (p.bias && co + e < p.Cout) ? p.bias[co + e] : 0.f;
  }
  const bool vec_ok = (co + KPO <= p.Cout) && ((p.ldy * (int)sizeof(TO)) % 16 == 0);
  char* yb = reinterpret_cast<char*>(p.y);
  // BN-backward partial sums (BNB): x̂ = x·ca + cb with ca = invstd, cb = −mean·invstd
  float bs1[2][KPO], bs2[2][KPO], bca[2][KPO], bcb[2][KPO], bsc[KPO], bsh[KPO];
  if constexpr (BNB) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < KPO; ++e) {
        bs1[q][e] = 0.f; bs2[q][e] = 0.f; bca[q][e] = 0.f; bcb[q][e] = 0.f;
        if (p.bnb_part[q] && co + e < p.Cout) {
          const float mu = p.bnb_coef[q][co + e], is = p.bnb_coef[q][p.Cout + co + e];
          bca[q][e] = is;
          bcb[q][e] = -mu * is;
        }
      }
#pragma unroll
    for (int e = 0; e < KPO; ++e) {
      const bool need = !p.bnb_mask && co + e < p.Cout;
      bsc[e] = need ? p.bnb_coef[0][2 * p.Cout + co + e] : 0.f;
      bsh[e] = need ? p.bnb_coef[0][3 * p.Cout + co + e] : 0.f;
    }
  }
  // EPRE instantiation (launches whose epilogue reads operand rows back: the residual + its ReLU bit mask of the data-gradient
  // join, the accumulate target, the GELU pre-activation): those rows are PREFETCHED four at a time with unconditional loads from
  // clamped addresses — inside the per-row control flow every such load is followed by a full vmcnt(0) wait, one memory round trip
  // per row (8-16 per tile).  A separate instantiation because the 32 extra VGPRs cost the plain launches occupancy.
  constexpr int ITERS = (BQ + RPP - 1) / RPP;
  constexpr int GRP = EPRE ? (ITERS < 4 ? ITERS : 4) : 1;
  const bool all_vec = (p.Cout % KPO == 0) && ((p.ldy * (int)sizeof(TO)) % 16 == 0);
  // second stream: the accumulate target OR the GELU pre-activation (both at once keeps the per-row loads)
  const bool pre = EPRE && all_vec && (p.residual || p.accumulate || p.act == 3) && !(p.accumulate && p.act == 3);
  const int cc = co < p.Cout ? co : 0;
#pragma unroll 4
  for (int base = 0; base < ITERS; base += GRP) {
  int ms[GRP];
  bool okr[GRP];
  u32x4 pr[GRP], pb[GRP];
  unsigned pmk[GRP];
#pragma unroll
  for (int i = 0; i < GRP; ++i) {
    const int rr = rl + (base + i) * RPP;
    int m = m0 + rr;
    okr[i] = rr < BQ && m < mlim && co < p.Cout;
    if (!okr[i]) m = m0;
    if (p.pclass) {
      uint32_t n_img, oh, ow;
      decode(m, n_img, oh, ow);
      m = (int)((n_img * p.OH + oh) * p.OW + ow);
    }
    ms[i] = m;
    if constexpr (EPRE) {
      if (pre) {
        const size_t eo = ((size_t)m * p.ldy + cc) * sizeof(TO);
        if (p.residual) {
          pr[i] = ld16(reinterpret_cast<const char*>(p.residual) + eo);
          if (p.res_mask) pmk[i] = p.res_mask[(size_t)m * (p.ldy / KPO) + cc / KPO];
        }
        if (p.accumulate || p.act == 3) pb[i] = ld16((p.accumulate ? yb : reinterpret_cast<const char*>(p.y2)) + eo);
      }
    }
  }
  if constexpr (EPRE) __builtin_amdgcn_sched_barrier(0);   // the prefetch group is issued before any of it is consumed
#pragma unroll
  for (int i = 0; i < GRP; ++i) {
    if (!okr[i]) continue;
    const int rr = rl + (base + i) * RPP;
    const int m = ms[i];
    u32x4 v = *reinterpret_cast<const u32x4*>(smem + rr * OROWB + oc * 16);
    float f[KPO];
    Chunk<TO>::unpack(v, f);
    char* dst = yb + ((size_t)m * p.ldy + co) * sizeof(TO);
    const bool post = p.bias || p.accumulate || p.out_relu || p.residual || p.act;
    if (post) {
      if (p.residual) {
        float g[KPO];
        const char* rsrc = reinterpret_cast<const char*>(p.residual) + ((size_t)m * p.ldy + co) * sizeof(TO);
        if (pre) {
          Chunk<TO>::unpack(pr[i], g);
        } else if (vec_ok) {
          Chunk<TO>::unpack(ld16(rsrc), g);
        } else {
#pragma unroll
          for (int e = 0; e < KPO; ++e) g[e] = (co + e < p.Cout) ? to_f32(reinterpret_cast<const TO*>(rsrc)[e]) : 0.f;
        }
        if (p.res_mask) {   // residual-branch gradient join: only where the block's ReLU was active
          const unsigned bits = pre ? pmk[i] : (unsigned)p.res_mask[(size_t)m * (p.ldy / KPO) + co / KPO];
#pragma unroll
          for (int e = 0; e < KPO; ++e) g[e] = (bits >> e) & 1u ? g[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < KPO; ++e) f[e] += g[e];
      }
      if (p.accumulate) {
        float g[KPO];
        if (pre) {
          Chunk<TO>::unpack(pb[i], g);
        } else if (vec_ok) {
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
        Chunk<TO>::unpack(pre ? pb[i] : ld16(reinterpret_cast<const char*>(p.y2) + ((size_t)m * p.ldy + co) * sizeof(TO)), z);
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
        const float d = f[e] - kshift[e];
        s1[e] += d;
        s2[e] = fmaf(d, d, s2[e]);
      }
    }
    if constexpr (BNB) {
      const size_t eo = ((size_t)m * p.ldy + co) * sizeof(TO);
      const unsigned bits = p.bnb_mask ? p.bnb_mask[(size_t)m * (p.ldy / KPO) + co / KPO] : 0u;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (!p.bnb_part[q]) continue;
        float xv[KPO];
        Chunk<TO>::unpack(ld16(reinterpret_cast<const char*>(p.bnb_x[q]) + eo), xv);
#pragma unroll
        for (int e = 0; e < KPO; ++e) {
          const bool keep = p.bnb_mask ? ((bits >> e) & 1u) : (fmaf(xv[e], bsc[e], bsh[e]) > 0.f);
          const float gg = keep ? f[e] : 0.f;
          bs1[q][e] += gg;
          bs2[q][e] = fmaf(gg, fmaf(xv[e], bca[q][e], bcb[q][e]), bs2[q][e]);
        }
      }
    }
#ifdef PFR_IGEMM_TRACE
    if (p.dbg & 2) continue;
#endif
    if (vec_ok) {
      st16(dst, v);
    } else {
#pragma unroll
      for (int e = 0; e < KPO; ++e)
        if (co + e < p.Cout) reinterpret_cast<TO*>(dst)[e] = from_f32<TO>(f[e]);
    }
  }
  }
  TSTAMP(5);
  if (p.stats_part) {
    // lanes with equal (lane % CPR) hold partials of the same channels
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < KPO; ++e) {
        s1[e] += __shfl_xor(s1[e], o, 64);
        s2[e] += __shfl_xor(s2[e], o, 64);
      }
    float* red = reinterpret_cast<float*>(smem + EPI);  // [NW waves][2][BP]
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < KPO; ++e) {
        red[(wave * 2 + 0) * BP + oc * KPO + e] = s1[e];
        red[(wave * 2 + 1) * BP + oc * KPO + e] = s2[e];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS only: the output stores stay in flight
    __builtin_amdgcn_s_barrier();
    if (tid < BP && n0 + tid < p.Cout) {
      const int ch = tid;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        a += red[(w * 2 + 0) * BP + ch];
        b += red[(w * 2 + 1) * BP + ch];
      }
      const float nt = (float)min(BQ, mlim - m0);
      const float k = to_f32(*reinterpret_cast<const TO*>(smem + ch * (int)sizeof(TO)));
      p.stats_part[((size_t)tm * 2 + 0) * p.Cout + n0 + ch] = k + a / nt;          // tile mean
      p.stats_part[((size_t)tm * 2 + 1) * p.Cout + n0 + ch] = b - a * a / nt;      // tile M2 = Σ (x − mean_t)²
    }
  }
  if constexpr (BNB) {
    float* red = reinterpret_cast<float*>(smem + EPI);  // [NW waves][2][BP]
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (!p.bnb_part[q]) continue;   // (uniform)
#pragma unroll
      for (int o = CPR; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < KPO; ++e) {
          bs1[q][e] += __shfl_xor(bs1[q][e], o, 64);
          bs2[q][e] += __shfl_xor(bs2[q][e], o, 64);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // the previous set's readers are done with `red`
      if (lane < CPR) {
#pragma unroll
        for (int e = 0; e < KPO; ++e) {
          red[(wave * 2 + 0) * BP + oc * KPO + e] = bs1[q][e];
          red[(wave * 2 + 1) * BP + oc * KPO + e] = bs2[q][e];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid < BP && n0 + tid < p.Cout) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          a += red[(w * 2 + 0) * BP + tid];
          b += red[(w * 2 + 1) * BP + tid];
        }
        p.bnb_part[q][((size_t)tm * 2 + 0) * p.Cout + n0 + tid] = a;
        p.bnb_part[q][((size_t)tm * 2 + 1) * p.Cout + n0 + tid] = b;
      }
    }
  }
  TSTAMP(6);
  if constexpr (!FILT) break;
  }
}

// ------------------------------------------------------------------------------------------------
template <typename T, typename TO, int BQ, int BP, int KCH, int NW, int WP, int NST = 2>
static int launch_tile_k(IgemmParams& p, hipStream_t st) {
  const bool fast = (p.C % (KCH * DT<T>::KPACK)) == 0;
  p.pclass = (fast && p.idil_log2 == 1 && p.ostride == 1 && p.R > 1 && (p.OH % 2) == 0 && (p.OW % 2) == 0 && !p.stats_part &&
              !p.bnb_part[0]) ? 1 : 0;
  p.mclass = p.N * (p.OH / 2) * (p.OW / 2);
  p.tpc = (p.mclass + BQ - 1) / BQ;
  p.div_chw = make_fastdiv((uint32_t)((p.OH / 2) * (p.OW / 2) > 0 ? (p.OH / 2) * (p.OW / 2) : 1));
  p.div_cw = make_fastdiv((uint32_t)(p.OW / 2 > 0 ? p.OW / 2 : 1));
  p.tilesM = p.pclass ? 4 * p.tpc : (p.M + BQ - 1) / BQ;
  p.tilesN = (p.Cout + BP - 1) / BP;
  p.krot = pfr_knob(KNOB_IGEMM_KROT);
  p.dma_sched = pfr_knob(KNOB_IGEMM_DMA);
  dim3 grid((unsigned)(p.tilesM * p.tilesN)), block(NW * 64);
#ifdef PFR_IGEMM_TRACE
  if (p.dbg & 16) grid.x = (grid.x + 3) / 4 * 8;   // experiment: the tiles run on XCDs 0-3 only (blocks of XCDs 4-7 exit at once)
#endif
  if (p.pro_scale) {
    if (fast) hipLaunchKernelGGL((igemm_kernel<T, TO, BQ, BP, true, true, KCH, NW, WP, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((igemm_kernel<T, TO, BQ, BP, true, false, KCH, NW, WP, 2>), grid, block, 0, st, p);
  } else {
    if constexpr (sizeof(T) == sizeof(TO)) {
      if (p.bnb_part[0]) {
        if (!fast || p.pclass) { pfr_set_error("conv2d: BN-backward sums need the k-step-uniform, non-parity-class path"); return PFR_ERR_UNSUPPORTED; }
        hipLaunchKernelGGL((igemm_kernel<T, TO, BQ, BP, false, true, KCH, NW, WP, NST, false, true>), grid, block, 0, st, p);
        PFR_CHECK_LAUNCH();
        return PFR_OK;
      }
    }
    if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) {
      // (measured: −10 % on the data-gradient joins of ResNet-50 = +0.5 % on the step; the GELU-backward GEMMs of Swin-T gain 4 % in
      // isolation but the step does not, and a plain forward residual add is 3 % SLOWER with it — both keep the plain kernel)
      if (fast && (p.res_mask || p.accumulate) && p.act != 3 && p.Cout % 8 == 0 && p.ldy % 8 == 0) {
        hipLaunchKernelGGL((igemm_kernel<T, TO, BQ, BP, false, true, KCH, NW, WP, NST, false, false, true>), grid, block, 0, st, p);
        PFR_CHECK_LAUNCH();
        return PFR_OK;
      }
    }
    if (fast) hipLaunchKernelGGL((igemm_kernel<T, TO, BQ, BP, false, true, KCH, NW, WP, NST>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((igemm_kernel<T, TO, BQ, BP, false, false, KCH, NW, WP, NST>), grid, block, 0, st, p);
  }
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

// k-step width: 128-byte rows (16 MFMAs per barrier) pay off for long reductions; short / bandwidth-bound layers
// (K = 64 … 256) run better with 64-byte rows and more resident workgroups.
template <typename T, typename TO, int BQ, int BP>
static int launch_tile(IgemmParams& p, hipStream_t st) {
  const int kch = pfr_knob(KNOB_IGEMM_KCH);   // tuning: force 64-/128-byte k-steps
  if (kch == 4) return launch_tile_k<T, TO, BQ, BP, 4, 4, 2>(p, st);
  if (kch == 8 && p.C % (8 * DT<T>::KPACK) == 0) return launch_tile_k<T, TO, BQ, BP, 8, 4, 2>(p, st);
  if (p.K >= 512 && p.C % (8 * DT<T>::KPACK) == 0) return launch_tile_k<T, TO, BQ, BP, 8, 4, 2>(p, st);
  return launch_tile_k<T, TO, BQ, BP, 4, 4, 2>(p, st);
}

// Tile choice.  Long reductions with many output channels use 8-wave 256-row tiles (256x256 / 256x128: twice / 1.33x
// the FLOPs per byte staged through LDS of the 128x128 tile); everything else the 4-wave tiles.
// Returns the m-tile height (also what the caller sizes stats_part with) and the variant id.
enum { TILE_128x128, TILE_64x128, TILE_128x64, TILE_64x64, TILE_256x256, TILE_256x128 };
static int pick_tile(int M, int Cout, int K, int dtype, int out_dtype, int* bq) {
  const int forced = pfr_knob(KNOB_IGEMM_TILE);   // tuning: tools/tile_sweep.py
  if (forced >= 0) {
    const bool big_ok = dtype == PFR_BF16 && out_dtype == PFR_BF16 && K >= 64 && (K % 64) == 0;
    if ((forced == TILE_256x256 || forced == TILE_256x128) && !big_ok) { /* fall through to the heuristic */ }
    else {
      *bq = (forced == TILE_256x256 || forced == TILE_256x128) ? 256 : ((forced == TILE_128x128 || forced == TILE_128x64) ? 128 : 64);
      return forced;
    }
  }
  const bool allow_big = pfr_knob(KNOB_IGEMM_BIG) != 0;
  // (8-wave tiles for K < 512 and 64-row tiles for the short-K layers were measured: no gain / slower)
  if (allow_big && dtype == PFR_BF16 && out_dtype == PFR_BF16 && K >= 512 && (K % 64) == 0) {
    const long t256 = (long)((M + 255) / 256);
    // one 8-wave tile per CU: a launch of T tiles runs ceil(T / 256) rounds — 257..365 tiles (a second round less than 43 % full) go to
    // the 4-wave tiles instead (Swin-T stage 4, 6272 x 768 -> 3072: 300 tiles, 79 / 103 us against 71 / 88 us for GELU / GELU backward)
    const long T = t256 * ((Cout + 255) / 256);
    const bool rounds_ok = T * 10 >= ((T + 255) / 256) * 256 * 7;
    if (Cout >= 192 && T >= 160 && rounds_ok) { *bq = 256; return TILE_256x256; }
    // (Cout = 128, a single column of 256x128 tiles, loses 10-17 % to the 128x128 tile: tools/tile_sweep.py)
    // (round 5: a 256x192 tile — no idle columns at Cout = 192 / 384 / 576, the Swin widths — was bit-identical and within 1 % of the
    //  256x256 tile on every Swin-T geometry, 65.6 vs 65.9 us at 25088 x 1536 -> 384: the idle quarter is not what bounds them)
    if (rounds_ok && Cout >= 256 && t256 * ((Cout + 127) / 128) >= 320) { *bq = 256; return TILE_256x128; }
  }
  const int bp = Cout > 64 ? 128 : 64;   // (Cout = 96: the 128-wide tile with a quarter of its columns idle still wins by ~10 %)
  const long tiles128 = (long)((M + 127) / 128) * ((Cout + bp - 1) / bp);
  // 64-row tiles only when 128-row tiles would leave most CUs idle; at ~1.5 workgroups per CU (the 7x7 layers: 392 tiles)
  // the 128-row tile still wins by 20-30 %: it re-fetches the 4.7 MB weight matrix half as often (measured)
  const int b = tiles128 >= 384 ? 128 : 64;
  *bq = b;
  if (bp == 128) return b == 128 ? TILE_128x128 : TILE_64x128;
  return b == 128 ? TILE_128x64 : TILE_64x64;
}

// Persistent kernel (pfr_igemm_p.hip) or one tile per workgroup?  PFR_IGEMM_P: 0 never, 1 heuristic (default), 2 whenever
// eligible.  Returns 1 and the persistent tile in (*bq, *bp) when the persistent kernel takes the launch.
static int pick_persistent(int M, int Cout, int K, int C, int dtype, int out_dtype, int has_pro, int act, int pclass, int* bq, int* bp) {
  const int mode = igemm_p_enabled();
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  if (mode == 0 || C % (4 * kp) != 0 || has_pro || act == 4) return 0;
  const int forced = igemm_p_forced_tile();   // 0:128x128 1:64x128 2:128x64 3:64x64 (PFR_IGEMM_PTILE / pfr_set_tuning)
  int q, pcols = Cout > 64 ? 128 : 64;
  const long tiles128 = (long)((M + 127) / 128) * ((Cout + pcols - 1) / pcols);
  q = tiles128 >= 1024 ? 128 : 64;   // at least ~2 tiles per persistent workgroup, else halve the tile height
  if (forced >= 0) { q = (forced == 0 || forced == 2) ? 128 : 64; pcols = (forced == 0 || forced == 1) ? 128 : 64; }
  *bq = q; *bp = pcols;
  if (mode >= 2) return 1;
  // Heuristic from the per-layer A/B (tools/gemm_ab.py, profiles/r02_gemm_ab.json).  Both kernels are bound by the same
  // resource — the vector-memory path into LDS (~24 B/clk/CU of LDS-DMA + stores) — and tie within +-5 % on
  // most ResNet-50 geometries; the persistent kernel wins clearly (1.3-1.5x) on the parity-class data gradients of the
  // stride-2 3x3 convs, whose four classes have different reduction lengths (one tile per workgroup leaves the short ones idle).
  return pclass ? 1 : 0;
}

template <typename T, typename TO>
static int launch_igemm(IgemmParams& p, int dtype, int out_dtype, hipStream_t st) {
  int bq, bp;
  // A launch that publishes statistics AND has post-ops (pfr_gemm_act_colstats) takes the plain tile kernel: its partial
  // granularity is what pfr_gemm_act_mtile reports (the alternative kernels decide on flags pfr_conv2d_mtile cannot see).
  const bool stats_postop = p.stats_part && (p.bias || p.act || p.accumulate || p.residual || p.out_relu);
  if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) {
    {
      const int rc = slin_try_launch(p, dtype, out_dtype, st);
      if (rc != 1) return rc;
    }
    if (!stats_postop) {
      int rc = sconv3_try_launch(p, dtype, out_dtype, st);
      if (rc != 1) return rc;
      rc = sstem_try_launch(p, dtype, out_dtype, st);
      if (rc != 1) return rc;
      rc = sconv_try_launch(p, dtype, out_dtype, st);
      if (rc != 1) return rc;
    }
  }
  if (p.res_sub) {   // only the streaming join reads a compact residual: never fall through to a kernel that would read it as full size
    pfr_set_error("pfr_conv2d_dgrad_bn_sub: geometry not taken by the streaming kernel");
    return PFR_ERR_UNSUPPORTED;
  }
  // a statistics launch goes only to a kernel whose partial granularity is the one the caller sized stats_part for
  const auto mtile_ok = [&](int mt) { return !p.stats_part || !p.want_mtile || p.want_mtile == mt; };
  const int pcl = igemm_pclass_ok(p) ? 1 : 0;
  if (!stats_postop && !p.bnb_part[0] && pick_persistent(p.M, p.Cout, p.K, p.C, dtype, out_dtype, p.pro_scale != nullptr, p.act, pcl, &bq, &bp) &&
      mtile_ok(bq / 2)) {
    const int rc = igemm_p_launch(p, dtype, out_dtype, bq, bp, st);
    if (rc != 1) return rc;
  }
  const int v = pick_tile(p.M, p.Cout, p.K, dtype, out_dtype, &bq);
  if (!mtile_ok(bq)) {
    pfr_set_error("conv / gemm statistics launch: stats_part was sized for %d-row partials (pfr_conv2d_mtile / pfr_gemm_act_mtile) but this "
                  "launch (ldy, bias / residual / ReLU / accumulate post-ops) can only take the %d-row tile kernel", p.want_mtile, bq);
    return PFR_ERR_ARG;
  }
  if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) {
    // (round 4 re-measured the alternatives on the 14x14 256->256 3x3 layer, 66.8 us / 886 TFLOP/s as is: 64-byte k-steps in a 4- / 3-slot
    //  ring 72.0 / 73.2 us — DMA latency is not what stalls the tile; 4 waves with 128x128 or 64x256 register tiles, compiler-scheduled,
    //  82.9 / 83.6 us — one wave per SIMD needs the hand-scheduled k-loop of pfr_sconv3.hip.  profiles/r04_tile_variants.txt)
    if (v == TILE_256x256) return launch_tile_k<T, TO, 256, 256, 8, 8, 2>(p, st);
    if (v == TILE_256x128) return launch_tile_k<T, TO, 256, 128, 8, 8, 2>(p, st);
  }
  switch (v) {
    case TILE_128x128: return launch_tile<T, TO, 128, 128>(p, st);
    case TILE_64x128: return launch_tile<T, TO, 64, 128>(p, st);
    case TILE_128x64: return launch_tile<T, TO, 128, 64>(p, st);
    default: return launch_tile<T, TO, 64, 64>(p, st);
  }
}

// rows per BatchNorm-statistics partial of a convolution launch with this geometry (the caller sizes stats_part with it):
// the m-tile height, or half of it when the persistent kernel (one partial per wave row) takes the launch
extern "C" int pfr_conv2d_mtile(int N, int H, int W, int C, int Cout, int R, int S, int stride, int pad, int OH, int OW, int dtype,
                                int out_dtype, int fused_prologue) {
  const int M = N * OH * OW, K = R * S * C;
  int bq, bp;
  if (!fused_prologue) {
    int bpw;
    if (sconv3_geom(N, H, W, C, Cout, R, S, stride, pad, 0, OH, OW, dtype, out_dtype, &bpw)) return bpw * 32;
    if (sstem_geom(N, H, W, C, Cout, R, S, stride, pad, 0, OH, OW, dtype, out_dtype, &bpw)) return bpw * 32;
    if (R == 1 && S == 1 && pad == 0 && (stride == 1 || (H == OH * stride && W == OW * stride))) {
      // 1x1: the streaming kernel publishes one partial per workgroup row range
      const int mt = sconv_mtile(M, Cout, K, (long)N * H * W, dtype, out_dtype);
      if (mt) return mt;
    }
  }
  // (statistics and the parity-class mode exclude each other, so the heuristic never picks the persistent kernel for a
  //  launch that publishes statistics; PFR_IGEMM_P=2 does)
  if (pick_persistent(M, Cout, K, C, dtype, out_dtype, fused_prologue, 0, 0, &bq, &bp)) return bq / 2;
  pick_tile(M, Cout, K, dtype, out_dtype, &bq);
  return bq;
}

struct BnbArgs {
  const void* x[2];
  const float* coef[2];
  float* part[2];
  const unsigned char* mask;
  int flags = 0;      // pfr_conv2d_dgrad_bn_ex: 1 = store through the mask, 2 = BN 0 without input / coefficients
  int res_sub = 0;    // the residual is the compact gradient of a stride-2 projection shortcut (streaming join only)
};
static int conv2d_fwd_impl(const void* x, const void* w, void* y, int dtype, int out_dtype, int N, int H, int W,
                           int C, int Cout, int R, int S, int stride, int pad, int idil_log2, int OH, int OW,
                           int ldy, const float* bias, const void* residual, int accumulate, int out_relu,
                           const float* pro_scale, const float* pro_shift, int pro_relu, float* stats_part,
                           const unsigned char* res_mask, hipStream_t stream, const BnbArgs* bnb = nullptr);
extern "C" int pfr_conv2d_fwd(const void* x, const void* w, void* y, int dtype, int out_dtype, int N, int H, int W,
                              int C, int Cout, int R, int S, int stride, int pad, int idil_log2, int OH, int OW,
                              int ldy, const float* bias, const void* residual, int accumulate, int out_relu,
                              const float* pro_scale, const float* pro_shift, int pro_relu, float* stats_part,
                              hipStream_t stream) {
  return conv2d_fwd_impl(x, w, y, dtype, out_dtype, N, H, W, C, Cout, R, S, stride, pad, idil_log2, OH, OW, ldy, bias, residual,
                         accumulate, out_relu, pro_scale, pro_shift, pro_relu, stats_part, nullptr, stream);
}
// data gradient joined with the residual-branch gradient of a block: dx = dgrad(dy) + (mask ? res : 0), where `res` is the
// gradient that arrived at the block output and `res_mask` the ReLU bit mask the forward tail wrote (pfr_bn_act_mask)
extern "C" int pfr_conv2d_dgrad_join(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int C, int Cout,
                                     int R, int S, int pad, int idil_log2, int OH, int OW, const void* res,
                                     const unsigned char* res_mask, hipStream_t stream) {
  PFR_CHECK_ARG(res && res_mask, "pfr_conv2d_dgrad_join: null pointer");
  PFR_CHECK_ARG(Cout % (dtype == PFR_BF16 ? 8 : 4) == 0, "pfr_conv2d_dgrad_join: Cout must be a multiple of the 16-byte chunk");
  return conv2d_fwd_impl(dy, wt, dx, dtype, dtype, N, H, W, C, Cout, R, S, 1, pad, idil_log2, OH, OW, Cout, nullptr, res, 0, 0,
                         nullptr, nullptr, 0, nullptr, res_mask, stream);
}
// Number of [2][Cin] partial rows pfr_conv2d_dgrad_bn writes per BN layer for this geometry (= m-tiles of the launch), or 0
// when the fused form is not available for it (the parity-class data gradients of stride-2 convs on even extents, channel
// counts that are not k-step multiples): the caller then runs pfr_bn_bwd_reduce on the finished gradient instead.
extern "C" int pfr_conv2d_dgrad_bn_parts(int dtype, int N, int H, int W, int C, int Cout, int R, int S, int idil_log2, int OH,
                                         int OW) {
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  if (sconv_bnb_mode() == 2) {   // streaming kernels: 1x1 / stride 1 data gradients they take (one BN; see pfr_sconv.hip) ...
    if (R == 3 && S == 3 && idil_log2 == 0 && OH == H && OW == W && pfr_knob(KNOB_BNB_TILE3) && dtype == PFR_BF16 && C % 64 == 0 &&
        Cout % 64 == 0) {
      // ... and (round 5, "bnb_tile3") the 3x3 / stride-1 data gradients on the 8-wave 256-row TILE kernel: MFMA-bound launches whose
      // epilogue has the slack for one more row stream (the BN input of the tile's own rows), replacing a reduce pass over (g, x).
      // Not the 64 -> 64 layers the halo-staged kernel takes (it has no such epilogue and is 1.7x the tile kernel there).
      int bpw, bq;
      if (sconv3_geom(N, H, W, C, Cout, R, S, 1, 1, 0, OH, OW, dtype, dtype, &bpw)) return 0;
      const int M = N * OH * OW;
      const int v = pick_tile(M, Cout, R * S * C, dtype, dtype, &bq);
      return (v == TILE_256x256 || v == TILE_256x128) ? (M + bq - 1) / bq : 0;
    }
    if (R != 1 || S != 1 || idil_log2 != 0 || OH != H || OW != W) return 0;
    return sconv_bnb_parts(N * OH * OW, Cout, C, dtype);
  }
  if (C % (4 * kp) != 0 || Cout % kp != 0) return 0;
  if (idil_log2 == 1 && (OH % 2) == 0 && (OW % 2) == 0) return 0;
  int bq;
  const int M = N * OH * OW;
  const int v = pick_tile(M, Cout, R * S * C, dtype, dtype, &bq);
  if ((v == TILE_256x256 || v == TILE_256x128) && C % 64 != 0) return 0;
  (void)H; (void)W;
  return (M + bq - 1) / bq;
}

// Data gradient (as pfr_conv2d_fwd over dy with tap-flipped weights; optional residual join `res` through `res_mask`, optional
// accumulation into dx) that ALSO produces the BatchNorm-backward partial sums of the BN layer(s) whose output gradient dx is:
//   part[t][0][c] = Σ_rows-of-tile-t g·mask,  part[t][1][c] = Σ g·mask·x̂,   g = the value stored to dx, x̂ = (x − mean)·invstd,
// mask = bit mask `bn_mask` ([M][Cout/KPACK] bytes, pfr_bn_act_mask) when given, else scale·x + shift > 0 (coef rows 2, 3).
// bn_x / bn_coef ([4][Cout]: mean, invstd, scale, shift) / bn_part describe the first BN, bn2_* an optional second one that
// consumes the same gradient through the same mask (projection shortcut).  Replaces pfr_bn_bwd_reduce's pass over (g, x).
static int dgrad_bn_impl(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int C, int Cout, int R, int S, int pad,
                         int idil_log2, int OH, int OW, const void* res, const unsigned char* res_mask, int accumulate, const void* bn_x,
                         const float* bn_coef, const unsigned char* bn_mask, float* bn_part, const void* bn2_x, const float* bn2_coef,
                         float* bn2_part, int flags, hipStream_t stream) {
  const bool nox = (flags & 2) != 0;
  PFR_CHECK_ARG(bn_part && (nox || (bn_x && bn_coef)), "pfr_conv2d_dgrad_bn: null pointer");
  PFR_CHECK_ARG(!(flags & 3) || (bn_mask && sconv_bnb_mode() == 2), "pfr_conv2d_dgrad_bn_ex: flags need the bit mask and the streaming form");
  PFR_CHECK_ARG(!nox || (res && res_mask), "pfr_conv2d_dgrad_bn_ex: flag 2 (no BN input) goes with the block join (res + res_mask)");
  PFR_CHECK_ARG(!bn2_part || (bn2_x && bn2_coef && bn_mask), "pfr_conv2d_dgrad_bn: the second BN needs x, coef and the shared bit mask");
  PFR_CHECK_ARG(sconv_bnb_mode() == 2 ? (res != nullptr || res_mask == nullptr) : ((res == nullptr) == (res_mask == nullptr)),
                "pfr_conv2d_dgrad_bn: res and res_mask go together (streaming form: res without a mask = plain add, res may be dx)");
  PFR_CHECK_ARG(pfr_conv2d_dgrad_bn_parts(dtype, N, H, W, C, Cout, R, S, idil_log2, OH, OW) > 0,
                "pfr_conv2d_dgrad_bn: geometry not supported by the fused form (see pfr_conv2d_dgrad_bn_parts)");
  PFR_CHECK_ARG(sconv_bnb_mode() != 2 || (!accumulate && (!res || !res_mask || bn_mask) && (!bn2_part || res)),
                "pfr_conv2d_dgrad_bn: the streaming form (bnb mode 2) takes no accumulation, a bit mask with the masked join, a second BN only with the join");
  BnbArgs b;
  b.x[0] = bn_x; b.coef[0] = bn_coef; b.part[0] = bn_part;
  b.x[1] = bn2_x; b.coef[1] = bn2_coef; b.part[1] = bn2_part;
  b.mask = bn_mask;
  b.flags = flags;
  return conv2d_fwd_impl(dy, wt, dx, dtype, dtype, N, H, W, C, Cout, R, S, 1, pad, idil_log2, OH, OW, Cout, nullptr, res, accumulate,
                         0, nullptr, nullptr, 0, nullptr, res_mask, stream, &b);
}
extern "C" int pfr_conv2d_dgrad_bn(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int C, int Cout,
                                   int R, int S, int pad, int idil_log2, int OH, int OW, const void* res,
                                   const unsigned char* res_mask, int accumulate, const void* bn_x, const float* bn_coef,
                                   const unsigned char* bn_mask, float* bn_part, const void* bn2_x, const float* bn2_coef,
                                   float* bn2_part, hipStream_t stream) {
  return dgrad_bn_impl(dy, wt, dx, dtype, N, H, W, C, Cout, R, S, pad, idil_log2, OH, OW, res, res_mask, accumulate, bn_x, bn_coef, bn_mask,
                       bn_part, bn2_x, bn2_coef, bn2_part, 0, stream);
}
// the same with `flags` (streaming form + bit mask only): 1 = dx is stored THROUGH bn_mask (dx = g*mask: every consumer of a block-output
// gradient reads it through that mask, so they may then read it plainly); 2 = the first BN's input is not read: only sum g*mask is
// produced (row 1 of bn_part = 0; bn_x / bn_coef may be NULL) — for the BN-input-free backward of pfr_bn3_bwd_coef
extern "C" int pfr_conv2d_dgrad_bn_ex(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int C, int Cout,
                                      int R, int S, int pad, int idil_log2, int OH, int OW, const void* res,
                                      const unsigned char* res_mask, int accumulate, const void* bn_x, const float* bn_coef,
                                      const unsigned char* bn_mask, float* bn_part, const void* bn2_x, const float* bn2_coef,
                                      float* bn2_part, int flags, hipStream_t stream) {
  return dgrad_bn_impl(dy, wt, dx, dtype, N, H, W, C, Cout, R, S, pad, idil_log2, OH, OW, res, res_mask, accumulate, bn_x, bn_coef, bn_mask,
                       bn_part, bn2_x, bn2_coef, bn2_part, flags, stream);
}

// dx = dgrad(dy) + up2(res_compact) + BN sums: the main-branch data gradient of a block whose projection shortcut is a 1x1 / stride-2
// conv.  The shortcut's gradient is computed DENSELY on its own (OH/2 x OW/2) grid by a plain pfr_conv2d_fwd into res_compact
// [N][OH/2][OW/2][Cout] and added here at the pixels with even (oh, ow) — instead of a scattered accumulate pass over dx — and the
// launch leaves the BatchNorm-backward sums of the BN whose output gradient dx is (bit mask), as pfr_conv2d_dgrad_bn.  Streaming
// kernels only: pfr_conv2d_dgrad_bn_parts (with pfr_set_tuning("bnb", 2)) > 0 and even OH, OW are required.
static int dgrad_bn_sub_impl(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int C, int Cout, int OH, int OW,
                             const void* res_compact, const void* bn_x, const float* bn_coef, const unsigned char* bn_mask,
                             float* bn_part, int flags, hipStream_t stream) {
  const bool nox = (flags & 2) != 0;
  PFR_CHECK_ARG(dy && wt && dx && res_compact && bn_mask && bn_part && (nox || (bn_x && bn_coef)), "pfr_conv2d_dgrad_bn_sub: null pointer");
  PFR_CHECK_ARG(sconv_bnb_mode() == 2 && OH == H && OW == W && !(OH & 1) && !(OW & 1) &&
                    pfr_conv2d_dgrad_bn_parts(dtype, N, H, W, C, Cout, 1, 1, 0, OH, OW) > 0,
                "pfr_conv2d_dgrad_bn_sub: needs the streaming form (bnb mode 2, an eligible 1x1 geometry, even extents)");
  BnbArgs b;
  b.x[0] = bn_x; b.coef[0] = bn_coef; b.part[0] = bn_part;
  b.x[1] = nullptr; b.coef[1] = nullptr; b.part[1] = nullptr;
  b.mask = bn_mask;
  b.res_sub = 1;
  b.flags = flags;
  return conv2d_fwd_impl(dy, wt, dx, dtype, dtype, N, H, W, C, Cout, 1, 1, 1, 0, 0, OH, OW, Cout, nullptr, res_compact, 0, 0, nullptr,
                         nullptr, 0, nullptr, nullptr, stream, &b);
}
extern "C" int pfr_conv2d_dgrad_bn_sub(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int C, int Cout, int OH,
                                       int OW, const void* res_compact, const void* bn_x, const float* bn_coef,
                                       const unsigned char* bn_mask, float* bn_part, hipStream_t stream) {
  return dgrad_bn_sub_impl(dy, wt, dx, dtype, N, H, W, C, Cout, OH, OW, res_compact, bn_x, bn_coef, bn_mask, bn_part, 0, stream);
}
// with the flags of pfr_conv2d_dgrad_bn_ex
extern "C" int pfr_conv2d_dgrad_bn_sub_ex(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int C, int Cout, int OH,
                                          int OW, const void* res_compact, const void* bn_x, const float* bn_coef,
                                          const unsigned char* bn_mask, float* bn_part, int flags, hipStream_t stream) {
  return dgrad_bn_sub_impl(dy, wt, dx, dtype, N, H, W, C, Cout, OH, OW, res_compact, bn_x, bn_coef, bn_mask, bn_part, flags, stream);
}

static int conv2d_fwd_impl(const void* x, const void* w, void* y, int dtype, int out_dtype, int N, int H, int W,
                           int C, int Cout, int R, int S, int stride, int pad, int idil_log2, int OH, int OW,
                           int ldy, const float* bias, const void* residual, int accumulate, int out_relu,
                           const float* pro_scale, const float* pro_shift, int pro_relu, float* stats_part,
                           const unsigned char* res_mask, hipStream_t stream, const BnbArgs* bnb) {
  PFR_CHECK_ARG(x && w && y, "pfr_conv2d_fwd: null pointer");
  PFR_CHECK_ARG(dtype == PFR_F32 || dtype == PFR_BF16, "pfr_conv2d_fwd: bad dtype %d", dtype);
  PFR_CHECK_ARG(out_dtype == dtype || out_dtype == PFR_F32, "pfr_conv2d_fwd: out_dtype must be dtype or f32");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  PFR_CHECK_ARG(C % kp == 0, "pfr_conv2d_fwd: C=%d must be a multiple of %d (16-byte channel chunks)", C, kp);
  PFR_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout > 0 && R > 0 && S > 0 && OH > 0 && OW > 0 && stride > 0,
                "pfr_conv2d_fwd: bad geometry");
  PFR_CHECK_ARG((long)N * OH * OW < (1L << 31) && (long)N * H * W * C < (1L << 31), "pfr_conv2d_fwd: tensor too large");
  PFR_CHECK_ARG(!pro_scale || (C % 4 == 0 && pro_shift && C <= 2048), "pfr_conv2d_fwd: prologue needs scale and shift, C <= 2048");
  IgemmParams p;
  p.x = x; p.w = w; p.y = y;
  p.N = N; p.H = H; p.W = W; p.C = C;
  p.R = R; p.S = S; p.OH = OH; p.OW = OW; p.ostride = stride; p.pad = pad; p.idil_log2 = idil_log2;
  p.Cout = Cout; p.ldy = ldy > 0 ? ldy : Cout;
  p.M = N * OH * OW; p.K = R * S * C;
  p.stats_part = stats_part; p.bias = bias; p.residual = residual; p.accumulate = accumulate; p.out_relu = out_relu;
  p.act = 0; p.y2 = nullptr; p.ccnt = nullptr; p.cap = 0; p.col0 = 0; p.self_excl = 0; p.res_mask = res_mask; p.res_sub = 0;
  p.bnb_mask = nullptr;
  for (int q = 0; q < 2; ++q) { p.bnb_x[q] = nullptr; p.bnb_coef[q] = nullptr; p.bnb_part[q] = nullptr; }
  if (bnb) { p.res_sub = bnb->res_sub; p.bnb_flags = bnb->flags; p.bnb_mask = bnb->mask; for (int q = 0; q < 2; ++q) { p.bnb_x[q] = bnb->x[q]; p.bnb_coef[q] = bnb->coef[q]; p.bnb_part[q] = bnb->part[q]; } }
  p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.pro_relu = pro_relu;
  if (stats_part) p.want_mtile = pfr_conv2d_mtile(N, H, W, C, Cout, R, S, stride, pad, OH, OW, dtype, out_dtype, pro_scale != nullptr);
  p.div_ohow = make_fastdiv((uint32_t)(OH * OW));
  p.div_ow = make_fastdiv((uint32_t)OW);
#ifdef PFR_IGEMM_TRACE
  p.trace = g_igemm_trace;
  p.dbg = g_igemm_dbg;
#endif
  if (dtype == PFR_BF16) {
    if (out_dtype == PFR_BF16) return launch_igemm<bf16_t, bf16_t>(p, dtype, out_dtype, stream);
    return launch_igemm<bf16_t, float>(p, dtype, out_dtype, stream);
  }
  return launch_igemm<float, float>(p, dtype, out_dtype, stream);
}

// Linear layer with a fused activation epilogue (Swin MLP: models/swin.py FeedForward = Linear → GELU → Linear):
//   act 2: y = gelu(x·Wᵀ + b) and y2 = x·Wᵀ + b (the pre-activation, kept for backward)        [forward of fc1]
//   act 3: y = (x·Wᵀ) ∘ gelu'(y2)                                                              [data gradient of fc2]
static int gemm_act_impl(const void* x, const void* w, void* y, int dtype, long M, int K, int N, const float* bias, int act,
                         void* y2, float* stats_part, hipStream_t stream);
// rows per statistics partial of pfr_gemm_act_colstats for this geometry (a launch with post-ops always takes the tile kernel)
extern "C" int pfr_gemm_act_mtile(long M, int K, int N, int dtype) {
  int bq;
  pick_tile((int)M, N, K, dtype, dtype, &bq);
  return bq;
}
extern "C" int pfr_gemm_act(const void* x, const void* w, void* y, int dtype, long M, int K, int N, const float* bias,
                            int act, void* y2, hipStream_t stream) {
  return gemm_act_impl(x, w, y, dtype, M, K, N, bias, act, y2, nullptr, stream);
}
// the same, and the epilogue also leaves the per-m-tile column statistics of the stored output (tile mean, tile M2; tile height
// pfr_conv2d_mtile(M, N, K, K, …)) — the column SUM of y (a bias gradient) is then sum_t rows_t * mean_t without another pass over y
extern "C" int pfr_gemm_act_colstats(const void* x, const void* w, void* y, int dtype, long M, int K, int N, const float* bias,
                                     int act, void* y2, float* stats_part, hipStream_t stream) {
  PFR_CHECK_ARG(stats_part, "pfr_gemm_act_colstats: null stats_part");
  return gemm_act_impl(x, w, y, dtype, M, K, N, bias, act, y2, stats_part, stream);
}
// GELU backward on a data gradient (act 3) + plain column SUMS of the stored output for the bias gradient of the layer in front (replaces
// the fc1.bias gradient of the reference's FeedForward, models/swin.py:39-52, without a pass over the widest gradient tensor): sums_part
// [parts][N] fp32, parts = pfr_gemm_act_colsum_parts(M, K, N, dtype); each partial row = column sums over a disjoint set of rows
// (pfr_colsum_final_batch with mt = 0 adds them up).  Provided by the streaming Linear kernel only: parts == 0 -> use pfr_gemm_act_colstats.
static void gemm_act_params(IgemmParams& p, const void* x, const void* w, void* y, long M, int K, int N, const float* bias, int act, void* y2) {
  p.x = x; p.w = w; p.y = y;
  p.N = (int)M; p.H = 1; p.W = 1; p.C = K;
  p.R = 1; p.S = 1; p.OH = 1; p.OW = 1; p.ostride = 1; p.pad = 0; p.idil_log2 = 0;
  p.Cout = N; p.ldy = N;
  p.M = (int)M; p.K = K;
  p.stats_part = nullptr; p.bias = bias; p.residual = nullptr; p.accumulate = 0; p.out_relu = 0;
  p.pro_scale = nullptr; p.pro_shift = nullptr; p.pro_relu = 0;
  p.act = act; p.y2 = y2; p.ccnt = nullptr; p.cap = 0; p.col0 = 0; p.self_excl = 0; p.res_mask = nullptr; p.res_sub = 0;
  p.bnb_mask = nullptr;
  for (int q = 0; q < 2; ++q) { p.bnb_x[q] = nullptr; p.bnb_coef[q] = nullptr; p.bnb_part[q] = nullptr; }
}
extern "C" int pfr_gemm_act_colsum_parts(long M, int K, int N, int dtype) {
  if (dtype != PFR_BF16 || M <= 0 || M >= (1L << 31) || K % 8 || N % 8) return 0;
  IgemmParams p;
  int dummy = 0;
  gemm_act_params(p, &dummy, &dummy, &dummy, M, K, N, nullptr, 3, &dummy);
  return slin_colsum_parts(p, dtype);
}
extern "C" int pfr_gemm_act_colsums(const void* x, const void* w, void* y, int dtype, long M, int K, int N, void* y2, float* sums_part,
                                    hipStream_t stream) {
  PFR_CHECK_ARG(x && w && y && y2 && sums_part, "pfr_gemm_act_colsums: null pointer");
  PFR_CHECK_ARG(pfr_gemm_act_colsum_parts(M, K, N, dtype) > 0, "pfr_gemm_act_colsums: geometry not taken (pfr_gemm_act_colsum_parts == 0): use pfr_gemm_act_colstats");
  IgemmParams p;
  gemm_act_params(p, x, w, y, M, K, N, nullptr, 3, y2);
  const int rc = slin_colsum_launch(p, dtype, sums_part, stream);
  if (rc == 1) { pfr_set_error("pfr_gemm_act_colsums: the streaming kernel did not take the launch"); return PFR_ERR_UNSUPPORTED; }
  return rc;
}
static int gemm_act_impl(const void* x, const void* w, void* y, int dtype, long M, int K, int N, const float* bias, int act,
                         void* y2, float* stats_part, hipStream_t stream) {
  PFR_CHECK_ARG(x && w && y && y2, "pfr_gemm_act: null pointer");
  PFR_CHECK_ARG(dtype == PFR_F32 || dtype == PFR_BF16, "pfr_gemm_act: bad dtype %d", dtype);
  PFR_CHECK_ARG(act == 2 || act == 3, "pfr_gemm_act: act must be 2 (gelu, keep pre-activation) or 3 (multiply by gelu')");
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  PFR_CHECK_ARG(K % kp == 0 && N % kp == 0, "pfr_gemm_act: K and N must be multiples of %d", kp);
  PFR_CHECK_ARG(M > 0 && M < (1L << 31) && (long)M * K < (1L << 31), "pfr_gemm_act: tensor too large");
  IgemmParams p;
  p.x = x; p.w = w; p.y = y;
  p.N = (int)M; p.H = 1; p.W = 1; p.C = K;
  p.R = 1; p.S = 1; p.OH = 1; p.OW = 1; p.ostride = 1; p.pad = 0; p.idil_log2 = 0;
  p.Cout = N; p.ldy = N;
  p.M = (int)M; p.K = K;
  p.stats_part = stats_part; p.bias = bias; p.residual = nullptr; p.accumulate = 0; p.out_relu = 0;
  p.pro_scale = nullptr; p.pro_shift = nullptr; p.pro_relu = 0;
  p.act = act; p.y2 = y2; p.ccnt = nullptr; p.cap = 0; p.col0 = 0; p.self_excl = 0; p.res_mask = nullptr; p.res_sub = 0;
  p.bnb_mask = nullptr;
  for (int q = 0; q < 2; ++q) { p.bnb_x[q] = nullptr; p.bnb_coef[q] = nullptr; p.bnb_part[q] = nullptr; }
  if (stats_part) p.want_mtile = pfr_gemm_act_mtile(M, K, N, dtype);
  p.div_ohow = make_fastdiv(1u);
  p.div_ow = make_fastdiv(1u);
#ifdef PFR_IGEMM_TRACE
  p.trace = g_igemm_trace;
  p.dbg = g_igemm_dbg;
#endif
  if (dtype == PFR_BF16) return launch_igemm<bf16_t, bf16_t>(p, dtype, dtype, stream);
  return launch_igemm<float, float>(p, dtype, dtype, stream);
}

// Gallery match with the running top-K filter fused into the GEMM epilogue (replaces the score
// chunk + pfr_topk_update pair for every chunk after the first): scores = q·gᵀ (q [Q][D], g [n][D], L2-normalised rows of
// `dtype`), a score enters query r's candidate list iff its key exceeds the key of r's current K-th best.
template <typename T, int BQ, int BP, int KCH, int NW, int WP>
static int launch_filter_k(IgemmParams& p, hipStream_t st) {
  p.pclass = 0; p.mclass = 0; p.tpc = 1;
  p.div_chw = make_fastdiv(1u); p.div_cw = make_fastdiv(1u);
  p.tilesM = (p.M + BQ - 1) / BQ;
  p.tilesN = (p.Cout + BP - 1) / BP;
  const int total = p.tilesM * p.tilesN;
  const bool persist = true;   // (one tile per workgroup lost its A/B: profiles/r04_tile_variants.txt)
  const int slots = num_cus() * (NW == 8 ? 1 : 2);
  const dim3 grid((unsigned)(persist && total > slots ? slots : total)), block(NW * 64);
  p.fo_qg = p.fo_gg = 0;
  if ((int)grid.x == slots && slots % 8 == 0 && pfr_knob(KNOB_MATCH_ORDER)) {
    // L2-blocked order (IgemmParams::fo_qg): super-steps of qg query tiles x gg gallery tiles per XCD
    const int wpx = slots / 8;
    int qg = 1;
    while (qg * 2 <= 8 && qg * 2 <= p.tilesM && wpx % (qg * 2) == 0) qg *= 2;
    p.fo_qg = qg;
    p.fo_gg = wpx / qg;
  }
  hipLaunchKernelGGL((igemm_kernel<T, float, BQ, BP, false, true, KCH, NW, WP, 2, true>), grid, block, 0, st, p);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}

extern "C" int pfr_match_scores_filter(const void* q, const void* g, int dtype, int Q, int n, int D, int col0, int K,
                                       void* state, void* cand, int cap, int exclude_self, hipStream_t stream) {
  PFR_CHECK_ARG(q && g && state && cand, "pfr_match_scores_filter: null pointer");
  PFR_CHECK_ARG(dtype == PFR_F32 || dtype == PFR_BF16, "pfr_match_scores_filter: bad dtype %d", dtype);
  const int kp = dtype == PFR_BF16 ? 8 : 4;
  PFR_CHECK_ARG(D % kp == 0 && Q > 0 && n > 0 && K >= 1 && K <= 512 && cap >= 1, "pfr_match_scores_filter: bad geometry");
  const TopkState t = topk_state(state, Q, K);
  IgemmParams p;
  p.x = q; p.w = g; p.y = cand;
  p.N = Q; p.H = 1; p.W = 1; p.C = D;
  p.R = 1; p.S = 1; p.OH = 1; p.OW = 1; p.ostride = 1; p.pad = 0; p.idil_log2 = 0;
  p.Cout = n; p.ldy = n;
  p.M = Q; p.K = D;
  p.stats_part = nullptr; p.bias = nullptr; p.residual = nullptr; p.accumulate = 0; p.out_relu = 0;
  p.pro_scale = nullptr; p.pro_shift = nullptr; p.pro_relu = 0;
  p.act = 4; p.y2 = t.thrk; p.ccnt = t.ccnt; p.cap = cap; p.col0 = col0; p.self_excl = exclude_self; p.res_mask = nullptr; p.res_sub = 0;
  p.bnb_mask = nullptr;
  for (int q = 0; q < 2; ++q) { p.bnb_x[q] = nullptr; p.bnb_coef[q] = nullptr; p.bnb_part[q] = nullptr; }
  p.div_ohow = make_fastdiv(1u);
  p.div_ow = make_fastdiv(1u);
#ifdef PFR_IGEMM_TRACE
  p.trace = g_igemm_trace;
  p.dbg = g_igemm_dbg;
#endif
  PFR_CHECK_ARG(D % (8 * kp) == 0, "pfr_match_scores_filter: D must be a multiple of %d (128-byte k-steps)", 8 * kp);
  if (dtype == PFR_BF16) {
    if ((long)((Q + 255) / 256) * ((n + 255) / 256) >= 160) return launch_filter_k<bf16_t, 256, 256, 8, 8, 2>(p, stream);
    return launch_filter_k<bf16_t, 128, 128, 8, 4, 2>(p, stream);
  }
  return launch_filter_k<float, 128, 128, 8, 4, 2>(p, stream);
}
