// pfr_common.h — shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Everything here is written for wave64 / MFMA / 160 KiB LDS; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// dtype enum of the C-ABI (include/pfr_hip.h)
#define PFR_F32 0
#define PFR_BF16 1

// error codes of the C-ABI
#define PFR_OK 0
#define PFR_ERR_ARG (-1)
#define PFR_ERR_HIP (-2)
#define PFR_ERR_UNSUPPORTED (-3)

void pfr_set_error(const char* fmt, ...);

// Run-time tuning knobs: ONE table (pfr_api.hip), set through pfr_set_tuning(name, value) — the library reads no environment
// variables; a host that wants start-up values passes them through that call (the Python host maps PFR_TUNING="name=v,name=v" onto it).
// Every CHANGE bumps pfr_tuning_epoch(): launch plans bake kernel choices in and are rebuilt by the engines when the epoch moved.
enum PfrKnob {
  KNOB_IGEMM_P,       // persistent GEMM kernel: 0 never, 1 heuristic (default), 2 whenever eligible
  KNOB_IGEMM_PTILE,   // its tile: -1 heuristic, 0:128x128 1:64x128 2:128x64 3:64x64
  KNOB_IGEMM_PKCH,    // its k-step: 8 = 128-byte rows when C allows (default), 4 = 64-byte rows
  KNOB_IGEMM_PPF,     // its fragment-prefetch / loader-wave variants (0 default; 1-3: bit-identical alternatives kept for the A/B)
  KNOB_IGEMM_TILE,    // tile kernel: -1 heuristic, 0..5 force a tile (tools/tile_sweep.py)
  KNOB_IGEMM_KCH,     // tile kernel k-step: 0 heuristic, 4 / 8 forced
  KNOB_IGEMM_BIG,     // 8-wave 256-row tiles: 1 allowed (default), 0 never
  KNOB_SCONV,         // streaming 1x1 kernel: 0 off, 1 heuristic (default), 2 whenever eligible
  KNOB_SCONV3,        // halo-staged 3x3 64->64 kernel: 1 on (default)
  KNOB_BNB,           // BatchNorm-backward sums in the data-gradient epilogue: 0 off (library default), 1 tile kernels, 2 streaming kernels (what the engine sets)
  KNOB_SWGRAD,        // streaming 1x1 weight gradient: 0 never, 1 where measured faster (default), 2 wherever the geometry allows
  KNOB_WGRAD_BIG,     // 256x256 8-wave weight-gradient tiles: 0 off (default), 1 heuristic, 2 forced
  KNOB_WGRAD_TILE,    // weight-gradient tile: -1 heuristic (tools/tile_sweep.py)
  KNOB_WGRAD_SPLITS,  // force the split count of the tile weight gradient (0: model)
  KNOB_WGRAD9,        // halo-staged 3x3 weight gradient: 0 off, 1 the 56x56 class (default), 2 every geometry
  KNOB_WGRAD9_SLOTS,  // workgroups of a halo-staged weight-gradient launch (256 = one per CU; fewer leaves CUs to the main stream)
  KNOB_ATTN_MFMA,     // window attention on MFMA for bf16 / head_dim 32: 1 (default), 0 = the register-blocked fp32-style kernels
  KNOB_BNB_TILE3,     // with bnb = 2: the 3x3 / stride-1 data gradients on the 256-row tile kernel leave the BatchNorm-backward sums too
  KNOB_SLIN,          // streaming Linear kernel (pfr_slin.hip; K a multiple of 96): 0 off, 1 M >= 65536 rows (default), 2 whenever eligible
  KNOB_SLIN_NP,       // experiments: force its weight-panel width (0 auto, 64 / 96 / 192)
  KNOB_MATCH_ORDER,   // persistent filter GEMM of the gallery match: 1 = L2-blocked tile order per XCD (default), 0 = linear order
  KNOB_LN_RB,         // LayerNorm forward: rows in flight per lane group for C <= 512 (one 16-byte chunk per lane): 4 (default), 6, 8
  KNOB_SSTEM,         // halo-staged stem convolution (pfr_sstem.hip: 4x4 over 16 channels -> 64): 1 on (default), 0 = the tile kernel
  KNOB_IGEMM_DMA,     // -DPFR_IGEMM_SPREAD experiment builds only: DMA issue schedule of the 128-byte-k-step tile kernels (IgemmParams::dma_sched)
  KNOB_IGEMM_KROT,    // tile kernel: workgroup t starts its k-loop at k-step (t * krot) % nk (0 = every workgroup at k-step 0; summation order changes)
  KNOB_COUNT
};
extern int g_pfr_knob[KNOB_COUNT];
static inline int pfr_knob(int k) { return g_pfr_knob[k]; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: a host process that drives several GPUs must raise the
// limit on each of them (a process-wide `static bool` set it on the first device only, and the > 64 KiB launch then failed
// on the second).  `done` = one bit per device ordinal, set after the attribute call; two threads racing both make the (idempotent) call.
#include <atomic>
static inline bool pfr_lds_attr_needed(std::atomic<unsigned long long>& done, unsigned long long* bit) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  *bit = 1ull << (dev & 63);
  return !(done.load(std::memory_order_acquire) & *bit);
}
#define PFR_MAX_LDS_ONCE(done, bytes, ...)                                                                          \
  do {                                                                                                              \
    unsigned long long bit_;                                                                                        \
    if (pfr_lds_attr_needed(done, &bit_)) {                                                                         \
      const void* fns_[] = {__VA_ARGS__};                                                                           \
      for (const void* f_ : fns_) (void)hipFuncSetAttribute(f_, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); \
      done.fetch_or(bit_, std::memory_order_release);                                                               \
    }                                                                                                               \
  } while (0)

// When non-null, the next launch of an entry point that supports it (pfr_bn_bwd_apply) attaches this event to its kernel as the
// dispatch's own completion signal (hipExtLaunchKernel stopEvent) and clears it: the plan executor uses it for the fork that
// follows, instead of a separate hipEventRecord whose marker packet delays the next kernel of the stream by ~7 us.
extern thread_local hipEvent_t pfr_tls_stop_event;

#define PFR_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      pfr_set_error(__VA_ARGS__);           \
      return PFR_ERR_ARG;                   \
    }                                       \
  } while (0)

#define PFR_CHECK_LAUNCH()                                            \
  do {                                                                \
    hipError_t e_ = hipGetLastError();                                \
    if (e_ != hipSuccess) {                                           \
      pfr_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return PFR_ERR_HIP;                                             \
    }                                                                 \
  } while (0)

// Exact-GELU pieces: Φ(z) = (1 + erf(z/√2))/2 and φ(z) = e^{-z²/2}/√(2π) from ONE exponential (Abramowitz & Stegun 7.1.26,
// |erf error| <= 1.5e-7, i.e. below fp32 resolution of the products it enters).  libm's erff is a branchy ~40-instruction
// polynomial per element; in the fused GELU epilogues of the Swin MLP GEMMs that vector-ALU work was a third of the kernel time.
__device__ __forceinline__ void gelu_cdf_pdf(float z, float& cdf, float& pdf) {
  const float ax = fabsf(z) * 0.70710678118654752f;
  const float ex = __expf(-ax * ax);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float half_erfc = 0.5f * p * t * ex;
  cdf = z >= 0.f ? 1.f - half_erfc : half_erfc;
  pdf = 0.3989422804014327f * ex;
}

// ---- per-dtype traits: a "chunk" is always 16 bytes -------------------------------------------
template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int KPACK = 4;  // elements per 16-byte chunk
  static constexpr int ID = PFR_F32;
};
template <> struct DT<bf16_t> {
  static constexpr int KPACK = 8;
  static constexpr int ID = PFR_BF16;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 16-byte chunk <-> KPACK floats
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  __device__ __forceinline__ static void unpack(const u32x4& c, float* f) {
    f[0] = __uint_as_float(c[0]); f[1] = __uint_as_float(c[1]);
    f[2] = __uint_as_float(c[2]); f[3] = __uint_as_float(c[3]);
  }
  __device__ __forceinline__ static u32x4 pack(const float* f) {
    u32x4 c;
    c[0] = __float_as_uint(f[0]); c[1] = __float_as_uint(f[1]);
    c[2] = __float_as_uint(f[2]); c[3] = __float_as_uint(f[3]);
    return c;
  }
};
template <> struct Chunk<bf16_t> {
  static constexpr int N = 8;
  __device__ __forceinline__ static void unpack(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(c[i] << 16);
      f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ static u32x4 pack(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x2 p;
      p[0] = (bf16_t)f[2 * i];
      p[1] = (bf16_t)f[2 * i + 1];
      c[i] = __builtin_bit_cast(uint32_t, p);
    }
    return c;
  }
};

// order-preserving float -> uint32 key (larger score = larger key) used by the top-K selection
__device__ __forceinline__ uint32_t fkey(float s) {
  const uint32_t u = __float_as_uint(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}
// running top-K state layout (pfr_topk_state_bytes): cur u64 [rows][K] | cur_n int [rows] | flags (64 B) |
// thrk u32 [rows] (key of the current K-th best, 0 while the list is not full) | ccnt int [rows] (candidate counters)
struct TopkState {
  unsigned long long* cur;
  int* cur_n;
  int* flags;
  uint32_t* thrk;
  int* ccnt;
};
__host__ __device__ __forceinline__ TopkState topk_state(void* state, int rows, int K) {
  TopkState t;
  char* b = reinterpret_cast<char*>(state);
  t.cur = reinterpret_cast<unsigned long long*>(b);
  t.cur_n = reinterpret_cast<int*>(b + (size_t)rows * K * 8);
  t.flags = t.cur_n + rows;
  t.thrk = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(t.flags) + 64);
  t.ccnt = reinterpret_cast<int*>(t.thrk + rows);
  return t;
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }
// streaming (last-use) load: non-temporal hint, the line need not stay in L2 / MALL
#ifdef PFR_NO_NT   // A/B builds: plain loads
__device__ __forceinline__ u32x4 ld16_nt(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
#else
__device__ __forceinline__ u32x4 ld16_nt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// unsigned division by a runtime constant (host-precomputed), n < 2^31
struct FastDiv {
  uint32_t d, mul, shr;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  if (d == 1) { f.mul = 0; f.shr = 0; return f; }
  uint32_t l = 0;
  while ((1u << l) < d) ++l;              // l = ceil(log2 d)
  uint64_t m = ((uint64_t(1) << (32 + l)) + d - 1) / d;  // may exceed 32 bits by one bit
  f.mul = (uint32_t)(m - (uint64_t(1) << 32));
  f.shr = l;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  if (f.d == 1) return n;
  uint32_t t = __umulhi(n, f.mul);
  return (uint32_t)(((uint64_t)t + n) >> f.shr);
}

// XCD-aware bijective remap of a 1-D block id: consecutive logical tiles land on the same XCD
// (block b is observed to run on XCD b % 8 — speed only, never correctness).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk) {
  const uint32_t NX = 8;
  uint32_t xcd = bid % NX, q = nblk / NX, r = nblk % NX;
  uint32_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + bid / NX;
}
