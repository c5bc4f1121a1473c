// pfr_slin.hip — weight-stationary STREAMING Linear layer for the HBM-bound GEMMs of the Swin stages 1-2 (bf16).
//
// Replaces, for the geometries it takes, the same reference calls as pfr_igemm.hip's plain-GEMM form: `nn.Linear` forward / input
// gradient of /root/reference/models/swin.py (to_qkv :91, to_out :99, FeedForward :39-43) incl. the residual add, the fused GELU of
// the MLP (act 2: writes the pre-activation and its GELU) and GELU backward on the data gradient (act 3).
//     y[m][n] = sum_k x[m][k] * w[n][k]  (+ res[m][n]) (+ bias[n])  (then act)          x [M][K], w [N][K], y [M][N], all bf16
//
// Why another kernel: at batch 128 the Linear layers of stage 1 / 2 have M = 401 408 / 100 352 rows and K, N in 96 .. 768 — 150-700 MB
// of activations for 7-30 GFLOP: their ceiling is HBM.  The tile kernel reaches 0.3-0.5 of it there (a 128x128 tile is three k-steps
// behind a prologue and a two-pass epilogue, the weight tile re-staged per row tile), and pfr_sconv.hip's geometry rules (K a power
// of two, panel counts that divide 256) exclude every channel count of the form 96 * 2^k.  Same skeleton as pfr_sconv.hip, simpler
// parts (round 5):
//   * ONE persistent workgroup of 8 waves per CU, grid = npanels x nranges (the panels of a row range on one XCD); the weight panel
//     [NP = 64 | 96 couts][K] is copied to LDS once per workgroup (rows padded by 16 B: the row stride is an odd multiple of 16 B, so the
//     16-lane groups of ds_read_b128 are conflict free); the only workgroup barrier of the loop-carrying part follows that copy (the
//     column-sum form has one more at the very end).  (A 192-cout panel with 4 waves exists for experiments: it lost every A/B.)
//   * every wave owns 32-row blocks (interleaved over all waves of the chip: one moving window of the tensor) and walks K in chunks of
//     96 columns: the chunk [32 rows][96] arrives by 16-byte buffer loads in whole 192-byte row segments into REGISTERS, one chunk — and,
//     across a block boundary, one block — ahead of its use (the next block's first chunk and its residual / pre-activation rows are
//     requested under the current block's last MFMAs; the first block's before the weight copy), is written to the wave's private
//     6.5 KB LDS tile (208-byte rows) and read back as MFMA fragments;
//   * the loads are inline asm with HAND-COUNTED s_waitcnt vmcnt(n): the epilogue's stores are inline asm (buffer_store_b128_sync), and
//     hipcc's own count for a load that is OLDER than such stores comes out as if they did not exist, i.e. it waits for them — the full
//     write latency once per block.  Every epilogue operand is requested in straight-line code (a load inside a runtime branch, or a
//     select on a freshly loaded register, is followed by a full wait: the first version ran at 0.5-1.1x of the tile kernel for that);
//   * epilogue per block: accumulators -> bf16 -> the wave's LDS tile -> whole row segments of 16 bytes per lane (fully contiguous
//     for a single-panel layer) with residual / bias / GELU applied in the read-back pass — the ARITHMETIC of igemm_kernel's epilogue
//     (round to bf16, then add in fp32, round again), so results are bit-identical to the tile kernel's (same k order inside and
//     across the MFMAs); GELU backward can also leave the column sums of its stored output (pfr_gemm_act_colsums: the bias gradient
//     of the Linear in front), one partial row per row range, reduced in a fixed order without atomics.
// Measured: profiles/r05_slin.txt (1.08-1.55x the tile kernel cold, Swin-T step -3.4 %), race screen tools/slin_stress.py.
#include "pfr_igemm.h"
#ifdef PFR_SLIN_NT_ON   // A/B builds (-DPFR_SLIN_NT_ON): the row loads non-temporal — measured inside the Swin-T step: profiles/r05_ln.txt; off
#define PFR_SLIN_NT " nt"
#else
#define PFR_SLIN_NT ""
#endif

struct SlinParams {
  const bf16_t* x;
  const bf16_t* w;
  bf16_t* y;
  int M, K, N;
  const float* bias;      // [N] or nullptr
  const bf16_t* res;      // [M][N] or nullptr
  bf16_t* y2;             // act 2: pre-activation out; act 3: pre-activation in
  int act;                // 0 | 2 | 3 (IgemmParams::act)
  float* colsum;          // GELU backward only: [nranges][N] column sums of the stored y over each range's rows (pfr_gemm_act_colsums), or nullptr
  int npanels, nranges, nblk;
};

// EP (epilogue): what the read-back pass does — a template parameter, because the operands it needs (residual rows, the GELU
// pre-activation) are REQUESTED AT THE START of a block and consumed after its MFMAs, in straight-line code: a load inside a runtime
// branch is followed by a full wait (hipcc), i.e. one exposed memory round trip per epilogue pass.
enum { SLIN_PLAIN = 0, SLIN_BIAS = 1, SLIN_BIAS_RES = 2, SLIN_GELU = 3, SLIN_GELU_BWD = 4 };

// NW waves per workgroup: 8 (two per SIMD, <= 256 registers) with the 64- / 96-cout panels, 4 (<= 512 registers: two sets of row-operand
// registers next to 96 accumulators) with the 192-cout panel; ONE workgroup per CU either way
template <int NT, int EP>
__global__ __launch_bounds__(NT == 6 ? 256 : 512, 1) void slin_kernel(SlinParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NT * 32;                 // couts of the workgroup's weight panel
  constexpr int XSTR = 208;                   // chunk row: 96 bf16 + 16 B pad
  constexpr int XBUF = 32 * XSTR;             // one chunk buffer
  constexpr int OSTR = NP * 2 + 16;           // epilogue window row
  constexpr int CPR = NP / 8;                 // 16-byte pieces per window row
  constexpr int NPASS = 32 * CPR / 64;        // read-back passes per block
  constexpr int NW = NT == 6 ? 4 : 8, NTHR = NW * 64;
  // read-back passes run over the window in NH halves of CPH pieces per row (a 192-cout panel: two halves of 96 couts, one chunk
  // buffer's worth each); pass ps = h * PH + q touches piece i = q * 64 + lane of half h: row i / CPH, cout piece h * CPH + i % CPH
  constexpr int NH = NT == 6 ? 2 : 1, TH = NT / NH, CPH = CPR / NH, OSTRH = TH * 64 + 16, PH = NPASS / NH;
  constexpr int NBQ = (CPH == 8) ? 1 : 3;     // i % CPH repeats every 3 passes (64 % 12 = 4), every pass for 8 pieces per row
  constexpr int NBP = NH * NBQ;               // distinct (pass -> cout piece) patterns of a lane
  constexpr bool HAS_BIAS = EP == SLIN_BIAS || EP == SLIN_BIAS_RES || EP == SLIN_GELU;
  constexpr bool HAS_ROWS = EP == SLIN_BIAS_RES || EP == SLIN_GELU_BWD;   // a second [M][N] operand read in the epilogue
  static_assert(32 * OSTRH <= XBUF, "the epilogue window lives in the wave's chunk buffer");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.K, WSTR = K * 2 + 16, KC = K / 96;
  const int wbytes = (NP * WSTR + 127) & ~127;
  char* const xw = smem + wbytes + wave * XBUF;

  // (panel, range) of this workgroup: the panels of one range run on the same XCD (block b is observed on XCD b % 8 — speed only)
  const int b = blockIdx.x;
  const int xcd = b & 7, q8 = b >> 3;
  const int pn = q8 % p.npanels, r = (q8 / p.npanels) * 8 + xcd;
  const int n0 = pn * NP;


  const int bstride = p.nranges * NW;
  int blk = r * NW + wave;
  const int frow = lane & 31, fhalf = lane >> 5;
  const char* const wl = smem + frow * WSTR + fhalf * 16;      // this lane's weight-fragment base (tile t adds t*32*WSTR)
  const char* const xl = xw + frow * XSTR + fhalf * 16;        // this lane's x-fragment base inside a chunk buffer
  __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.M * p.N * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t y2rsrc = __builtin_amdgcn_make_buffer_rsrc(EP == SLIN_GELU ? p.y2 : p.y, 0, (int)((size_t)p.M * p.N * 2), 0x00020000);
  const bf16_t* const rows2 = EP == SLIN_BIAS_RES ? p.res : (EP == SLIN_GELU_BWD ? p.y2 : p.x);

  // per-lane geometry of the chunk pieces (piece i = j*64 + lane of the chunk's 384 16-byte pieces: row i / 12, piece i % 12 — whole
  // 192-byte row segments) and of the read-back pieces; bias of the (up to three) cout patterns of this lane, loaded once
  // (piece j of a lane: row r0 + 5 j + (c0 + 4 j) / 12, piece (c0 + 4 j) % 12 with (r0, c0) = (lane / 12, lane % 12): 64 = 5 * 12 + 4)
  const int lr0 = lane / 12, lc0 = lane - lr0 * 12;
  auto crow = [&](int j) { return lr0 + 5 * j + (lc0 + 4 * j) / 12; };
  auto cpc = [&](int j) { return (lc0 + 4 * j) % 12; };
  float bia[NBP][8];
  if constexpr (HAS_BIAS) {
#pragma unroll
    for (int q = 0; q < NBP; ++q) {
      const int c16 = (q / NBQ) * CPH + ((q % NBQ) * 64 + lane) % CPH;
      const int co = n0 + c16 * 8 < p.N ? n0 + c16 * 8 : 0;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + co), b1 = *reinterpret_cast<const f32x4*>(p.bias + co + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bia[q][e] = b0[e]; bia[q][4 + e] = b1[e]; }
    }
  }
  // column sums of the stored output (pfr_gemm_act_colsums): per lane, the 8 couts of each of its cout-piece patterns
  float cs[NBP][8];
#pragma unroll
  for (int q = 0; q < NBP; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[q][e] = 0.f;
  // chunk c of block bk -> registers.  UNCONDITIONAL loads from clamped rows and nothing that reads the registers before the LDS write
  // of the next iteration: rows past M compute garbage that is never stored.
  u32x4 R[6];
  // Inline asm + hand-counted waits (vm_wait below): the compiler's own s_waitcnt for a load that is OLDER than the epilogue's stores
  // came out as vmcnt(0..5) — i.e. it also waited for the stores, the full write latency once per block.  gfx950 retires a wave's
  // vector-memory operations in issue order, so "at most n younger operations outstanding" is exactly what vmcnt(n) tests.  Rows past
  // M lie past num_records and read zeros.
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (int)((size_t)p.M * K * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t r2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(rows2), 0,
                                                                    HAS_ROWS ? (int)((size_t)p.M * p.N * 2) : 16, 0x00020000);
  auto gload = [&](int bk, int c) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const uint32_t off = (uint32_t)(((bk * 32 + crow(j)) * K + c * 96 + cpc(j) * 8) * 2);
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" PFR_SLIN_NT : "=v"(R[j]) : "v"(off), "s"(xrsrc) : "memory");
    }
  };
  // wait until at most n vector-memory operations issued after the x chunk in R are outstanding; R (and everything older) has landed
  auto vm_wait_R = [&](int n) {
#define PFR_SLIN_W(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5])::"memory")
    if (n == 0) PFR_SLIN_W(0);
    else if (n <= 4) PFR_SLIN_W(4);
    else if (n <= 6) PFR_SLIN_W(6);
    else if (n <= 8) PFR_SLIN_W(8);
    else if (n <= 12) PFR_SLIN_W(12);
    else PFR_SLIN_W(24);
#undef PFR_SLIN_W
  };
  // the block's second row operand (residual / GELU pre-activation), one block AHEAD like the x rows: two register sets, the block
  // loop is unrolled by two so that each copy of the body names its sets statically
  constexpr int NR = HAS_ROWS ? NPASS : 1;
  auto rload = [&](u32x4 (&RSx)[NR], int bk) {
    if constexpr (HAS_ROWS) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int i = (ps % PH) * 64 + lane;
        const int row = i / CPH, c16 = (ps / PH) * CPH + (i - row * CPH);
        const int m = bk * 32 + row, co = n0 + c16 * 8;
        const uint32_t off = (m < p.M && co < p.N) ? (uint32_t)(((size_t)m * p.N + co) * 2) : 0xF0000000u;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" PFR_SLIN_NT : "=v"(RSx[ps]) : "v"(off), "s"(r2rsrc) : "memory");
      }
    }
  };
  auto mma_chunk = [&](f32x16 (&acc)[NT], int c) {
    const char* const xb = xl;
    const char* const wb = wl + c * 192;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const u32x4 xf = *reinterpret_cast<const u32x4*>(xb + s * 32);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const u32x4 wf = *reinterpret_cast<const u32x4*>(wb + t * 32 * WSTR + s * 32);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xf), acc[t], 0, 0, 0);
      }
    }
  };
  auto put_chunk = [&]() {
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<u32x4*>(xw + crow(j) * XSTR + cpc(j) * 16) = R[j];
  };
  // one block: x chunks through the wave's LDS tile (ONE buffer: the registers are the second one), the NEXT block's first chunk and
  // row operand requested under the last chunk's MFMAs, epilogue from RScur.  -> the next block, or -1
  constexpr int NST = (EP == SLIN_GELU ? 2 : 1) * NPASS;    // vector-memory stores of one epilogue
  auto block = [&](int bk, u32x4 (&RScur)[NR], u32x4 (&RSnext)[NR], bool first) -> int {
    const int m0 = bk * 32;
    const int nxt = bk + bstride;
    const int nxc = nxt < p.nblk ? nxt : bk;        // (the last block re-loads itself: no branch around the prefetch)
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    // chunk 0 (and, older, this block's row operand) was requested under the previous block's last MFMAs; the only younger operations
    // are that block's NST stores, which stay in flight
    if (first) vm_wait_R(0); else vm_wait_R(NST);
    if constexpr (HAS_ROWS) {   // (the row operand landed before R: every use of it is ordered behind the wait by this empty statement)
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) asm volatile("" : "+v"(RScur[ps])::"memory");
    }
    for (int c = 0; c + 1 < KC; ++c) {
      put_chunk();
      gload(bk, c + 1);
      mma_chunk(acc, c);
      vm_wait_R(0);
    }
    put_chunk();
    rload(RSnext, nxc);
    gload(nxc, 0);
    mma_chunk(acc, KC - 1);
    // ---- epilogue: accumulators -> bf16 window in the wave's LDS tile, 96 couts (one chunk-buffer's worth) at a time, then whole
    //      16-byte row pieces
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
      for (int t = 0; t < TH; ++t)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          bf16x4 v;
          const f32x16& a = acc[h * TH + t];
          v[0] = (bf16_t)a[4 * qd]; v[1] = (bf16_t)a[4 * qd + 1]; v[2] = (bf16_t)a[4 * qd + 2]; v[3] = (bf16_t)a[4 * qd + 3];
          *reinterpret_cast<bf16x4*>(xw + frow * OSTRH + (t * 32 + 8 * qd + 4 * fhalf) * 2) = v;
        }
#pragma unroll
      for (int q = 0; q < PH; ++q) {
        // piece i of this half: row i / CPH, 16-byte piece i % CPH of the half's CPH pieces
        const int i = q * 64 + lane;
        const int row = i / CPH, c16 = i - row * CPH;
        const int m = m0 + row, co = n0 + (h * CPH + c16) * 8;
        const bool ok = m < p.M && co < p.N;
        u32x4 v = *reinterpret_cast<const u32x4*>(xw + row * OSTRH + c16 * 16);
        float f[8];
        if constexpr (EP != SLIN_PLAIN) {
          Chunk<bf16_t>::unpack(v, f);
          if constexpr (EP == SLIN_BIAS_RES) {
            float g[8];
            Chunk<bf16_t>::unpack(RScur[h * PH + q], g);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += g[e];
          }
          if constexpr (HAS_BIAS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += bia[h * NBQ + q % NBQ][e];
          }
          v = Chunk<bf16_t>::pack(f);
          Chunk<bf16_t>::unpack(v, f);
        }
        const uint32_t off = ok ? (uint32_t)(((size_t)m * p.N + co) * 2) : 0xF0000000u;     // (past num_records: the store is dropped)
        if constexpr (EP == SLIN_GELU) {          // exact GELU of the STORED pre-activation; both tensors are kept for backward
          buffer_store_b128_sync(v, y2rsrc, off, 0);
#pragma unroll
          for (int e = 0; e < 8; ++e) { float cdf, pdf; gelu_cdf_pdf(f[e], cdf, pdf); f[e] *= cdf; }
          v = Chunk<bf16_t>::pack(f);
        } else if constexpr (EP == SLIN_GELU_BWD) {   // GELU backward on the data gradient: dz = dh * gelu'(z)
          float z[8];
          Chunk<bf16_t>::unpack(RScur[h * PH + q], z);
#pragma unroll
          for (int e = 0; e < 8; ++e) { float cdf, pdf; gelu_cdf_pdf(z[e], cdf, pdf); f[e] *= cdf + z[e] * pdf; }
          v = Chunk<bf16_t>::pack(f);
          if (p.colsum) {          // (sums see the value as stored; rows past M and couts past N add nothing)
            Chunk<bf16_t>::unpack(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[h * NBQ + q % NBQ][e] += ok ? f[e] : 0.f;
          }
        }
        buffer_store_b128_sync(v, yrsrc, off, 0);
      }
    }
    return nxt < p.nblk ? nxt : -1;
  };
  u32x4 RS0[NR], RS1[NR];
  const bool idle = r >= p.nranges || blk >= p.nblk;     // (a workgroup / wave without rows still helps to copy the panel)
  if (idle) blk = 0;
  // the first block's operands are requested BEFORE the weight panel is copied: their round trip runs under the copy and its barrier
  rload(RS0, blk);
  gload(blk, 0);
  // ---- weight panel -> LDS (once); rows past N are zero.  Four unconditional loads in flight per thread.
  {
    const int cpr = K >> 3;                    // 16-byte pieces per weight row
    const int total = NP * cpr;
    for (int i0 = tid; i0 < total; i0 += 4 * NTHR) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * NTHR < total ? i0 + u * NTHR : total - 1;
        const int n = i / cpr, c16 = i - n * cpr;
        const int nn = n0 + n < p.N ? n0 + n : p.N - 1;
        v[u] = *reinterpret_cast<const u32x4*>(p.w + (size_t)nn * K + c16 * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * NTHR;
        if (i < total) {
          const int n = i / cpr, c16 = i - n * cpr;
          *reinterpret_cast<u32x4*>(smem + n * WSTR + c16 * 16) = n0 + n < p.N ? v[u] : (u32x4){0u, 0u, 0u, 0u};
        }
      }
    }
  }
  __syncthreads();
  if (idle) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (EP != SLIN_GELU_BWD || !p.colsum || r >= p.nranges) return;     // (a rowless wave of a working range still joins the column sums)
  } else {
    blk = block(blk, RS0, RS1, true);
    while (blk >= 0) {
      blk = block(blk, RS1, RS0, false);
      if (blk < 0) break;
      blk = block(blk, RS0, RS1, false);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if constexpr (EP == SLIN_GELU_BWD) {
    if (p.colsum) {
      // every lane's sums go to the wave's LDS tile as [pattern][lane][8]; after the barrier thread i < NP adds up column i over all waves,
      // patterns and the lanes that held its cout piece under that pattern: fixed order, no atomics.  (NBP * 2 KB <= the 6.5 KB tile: the
      // 64- / 96-cout panels; the host does not offer the 192-cout panel with column sums.)
      static_assert(NT == 6 || NBP * 2048 <= XBUF, "column-sum scratch must fit the wave's LDS tile");
      if constexpr (NT != 6) {
        float* const scr = reinterpret_cast<float*>(xw);
#pragma unroll
        for (int q = 0; q < NBP; ++q)
#pragma unroll
          for (int e = 0; e < 8; ++e) scr[(q * 64 + lane) * 8 + e] = cs[q][e];
        __syncthreads();
        for (int i = tid; i < NP; i += NTHR) {
          const int piece = i >> 3, e = i & 7;
          float a = 0.f;
          for (int w = 0; w < NW; ++w) {
            const float* const sw = reinterpret_cast<const float*>(smem + wbytes + w * XBUF);
            for (int q = 0; q < NBP; ++q) {
              // lanes l with (q * 64 + l) % CPH == piece
              const int first = ((piece - q * 64) % CPH + CPH) % CPH;
              for (int l = first; l < 64; l += CPH) a += sw[(q * 64 + l) * 8 + e];
            }
          }
          if (n0 + i < p.N) p.colsum[(size_t)r * p.N + n0 + i] = a;
        }
      }
    }
  }
}

// "slin" knob: 0 off, 1 where the geometry is HBM-bound (M >= 65536 rows: Swin stages 1-2 at batch 128), 2 whenever eligible
static int slin_launch(IgemmParams& p, int dtype, int out_dtype, float* colsum, int* parts_only, hipStream_t st);
int slin_try_launch(IgemmParams& p, int dtype, int out_dtype, hipStream_t st) { return slin_launch(p, dtype, out_dtype, nullptr, nullptr, st); }
// pfr_gemm_act_colsums: GELU backward + column sums of the stored output, one partial row per row range (0 partial rows: not this kernel's launch)
int slin_colsum_parts(IgemmParams& p, int dtype) {
  int parts = 0;
  return slin_launch(p, dtype, dtype, nullptr, &parts, nullptr) == PFR_OK ? parts : 0;
}
int slin_colsum_launch(IgemmParams& p, int dtype, float* colsum, hipStream_t st) { return slin_launch(p, dtype, dtype, colsum, nullptr, st); }
// parts_only != nullptr: no launch, *parts_only = the row ranges (= column-sum partial rows) the launch would have
static int slin_launch(IgemmParams& p, int dtype, int out_dtype, float* colsum, int* parts_only, hipStream_t st) {
  const int mode = pfr_knob(KNOB_SLIN);
  if (!mode || dtype != PFR_BF16 || out_dtype != PFR_BF16) return 1;
  if (p.R != 1 || p.S != 1 || p.pad != 0 || p.idil_log2 != 0 || p.ostride != 1 || p.ldy != p.Cout) return 1;
  if (p.pro_scale || p.stats_part || p.accumulate || p.out_relu || p.res_mask || p.res_sub || p.bnb_part[0]) return 1;
  // the forms Swin's Linear layers come in: plain (data gradients), bias, bias + residual, fused GELU (bias), GELU backward (no bias)
  int ep;
  if (p.act == 0) {
    if (p.residual && !p.bias) return 1;
    ep = p.residual ? SLIN_BIAS_RES : (p.bias ? SLIN_BIAS : SLIN_PLAIN);
  } else if (p.act == 2) {
    if (!p.bias || p.residual || !p.y2) return 1;
    ep = SLIN_GELU;
  } else if (p.act == 3) {
    if (p.bias || p.residual || !p.y2) return 1;
    ep = SLIN_GELU_BWD;
  } else {
    return 1;
  }
  const int K = p.K, N = p.Cout, M = p.M;
  if (K % 96 || K > 768 || N % 32 || N < 64) return 1;
  if ((long)M * K * 2 >= (1L << 31) || (long)M * N * 2 >= (1L << 31) || M < 4096) return 1;
  // below 65536 rows only the Swin-T stage-3 shapes it wins cold by 8-20 % (25088 x 384 -> 1536 all forms, -> 384 with residual; tools/gemm_act_tiles.py)
  const bool st3 = M >= 16384 && M < 65536 && K == 384 && (N == 1536 || N == 384);
  if (mode == 1 && M < 65536 && !st3) return 1;
  // panel: the widest of 192 / 96 / 64 couts that divides N and fits the LDS next to the waves' tiles
  constexpr int XBUF = 32 * 208;                   // a wave's chunk buffer
  const int force = pfr_knob(KNOB_SLIN_NP);        // experiments: force the panel width (0: the widest that fits)
  int np = 0;
  // (warm, back-to-back, tools/slin_np.py: the 96-cout panel with 8 waves beats the 192-cout panel with 4 on every shape — 41 -> 28 us at
  //  192 -> 192, 113 -> 101 us at 96 -> 384 — although x is then read once per panel from L2; N = 192 runs best as three 64-cout panels)
  for (int c : {96, 64, 192}) {
    if (force ? c != force : (c == 192 || (c == 96 && N == 192))) continue;
    if (N % c == 0 && ((c * (K * 2 + 16) + 127) & ~127) + (c == 192 ? 4 : 8) * XBUF <= 160 * 1024) { np = c; break; }
  }
  if (!np) return 1;
  // Measured inside the Swin-T step (warm operands, tools/swin_ab.sh + bench.py --detail, profiles/r05_slin.txt): the 8-wave forms
  // (64- / 96-cout panels) win 15-25 % over the tile kernel; the 4-wave 192-cout-panel form loses 10-30 % and is kept for experiments only.
  // (round 5, same-box A/B of the step: + GELU backward at K = 192 (118 vs 148 us cold) and the stage-3 forms: 11708 -> 11735 img/s)
  if (mode == 1 && p.act != 0 && K != 96 && !st3 && !(K == 192 && p.act == 3)) return 1;      // (fused GELU at K = 192: 133 vs 130 us for the tile kernel; at K = 96: 183 vs 222)
  const int nw = np == 192 ? 4 : 8;
  const int lds = ((np * (K * 2 + 16) + 127) & ~127) + nw * XBUF;
  const int npanels = N / np;
  const int slots = num_cus();                     // one workgroup per CU (registers)
  int nranges = (slots / npanels) / 8 * 8;
  if (nranges < 8) return 1;
  SlinParams sp;
  sp.x = (const bf16_t*)p.x; sp.w = (const bf16_t*)p.w; sp.y = (bf16_t*)p.y;
  sp.M = M; sp.K = K; sp.N = N;
  sp.bias = p.bias; sp.res = (const bf16_t*)p.residual; sp.y2 = (bf16_t*)p.y2; sp.act = p.act;
  sp.nblk = (M + 31) / 32;
  if (nranges * nw > sp.nblk) nranges = ((sp.nblk + nw - 1) / nw + 7) / 8 * 8;
  sp.npanels = npanels; sp.nranges = nranges;
  sp.colsum = colsum;
  if (parts_only) { if (np == 192) return 1; *parts_only = nranges; return PFR_OK; }
  if (colsum && (ep != SLIN_GELU_BWD || np == 192)) return 1;
  const dim3 grid((unsigned)(npanels * nranges)), block((unsigned)(nw * 64));
  // The kernel's loads are inline asm behind hand-counted s_waitcnt vmcnt(n): a compiler-inserted vector-memory operation between them
  // (a scratch spill or reload of R / RS / acc under other flags or another hipcc) would shift the count and the MFMAs would read
  // registers that have not landed — silently.  An instantiation that uses ANY scratch is therefore never launched: the tile kernel
  // takes the launch (checked once per instantiation; tests/test_host_logic.py asserts the same from the ISA of the shipped flags).
#define PFR_SLIN_GO(NTV, EPV)                                                              \
  do {                                                                                     \
    static std::atomic<int> scratch_ok{0};   /* 0 unknown, 1 no scratch, 2 scratch */       \
    if (!scratch_ok.load(std::memory_order_relaxed)) {                                     \
      hipFuncAttributes fa;                                                                \
      const bool ok = hipFuncGetAttributes(&fa, (const void*)slin_kernel<NTV, EPV>) == hipSuccess && fa.localSizeBytes == 0; \
      scratch_ok.store(ok ? 1 : 2, std::memory_order_relaxed);                             \
    }                                                                                      \
    if (scratch_ok.load(std::memory_order_relaxed) != 1) return 1;                         \
    static std::atomic<unsigned long long> attr{0};                                        \
    PFR_MAX_LDS_ONCE(attr, 160 * 1024, (const void*)slin_kernel<NTV, EPV>);                \
    hipLaunchKernelGGL((slin_kernel<NTV, EPV>), grid, block, lds, st, sp);                 \
  } while (0)
#define PFR_SLIN_EP(NTV)                                                                   \
  switch (ep) {                                                                            \
    case SLIN_PLAIN: PFR_SLIN_GO(NTV, SLIN_PLAIN); break;                                  \
    case SLIN_BIAS: PFR_SLIN_GO(NTV, SLIN_BIAS); break;                                    \
    case SLIN_BIAS_RES: PFR_SLIN_GO(NTV, SLIN_BIAS_RES); break;                            \
    case SLIN_GELU: PFR_SLIN_GO(NTV, SLIN_GELU); break;                                    \
    default: PFR_SLIN_GO(NTV, SLIN_GELU_BWD); break;                                       \
  }
  if (np == 192) { PFR_SLIN_EP(6) } else if (np == 96) { PFR_SLIN_EP(3) } else { PFR_SLIN_EP(2) }
#undef PFR_SLIN_EP
#undef PFR_SLIN_GO
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
