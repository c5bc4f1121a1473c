// pfr_igemm.h — parameter block shared by the implicit-GEMM kernels (pfr_igemm.hip: one tile per workgroup;
// pfr_igemm_p.hip: persistent workgroups with a DMA ring that streams across tile boundaries).
#pragma once
#include "pfr_mma.h"

struct IgemmParams {
  const void* x;
  const void* w;
  void* y;
  int N, H, W, C;
  int R, S, OH, OW, ostride, pad, idil_log2;
  int Cout, ldy;
  int M, K;
  float* stats_part;  // [tilesM][2][Cout] (tile mean, tile M2) or nullptr
  int want_mtile = 0;  // rows per statistics partial the caller sized stats_part for (pfr_conv2d_mtile / pfr_gemm_act_mtile): a kernel
                       // with another granularity must not take the launch (it would write a different number of partial rows)
  const float* bias;  // [Cout] or nullptr
  const void* residual;  // [M][ldy] of TO or nullptr: y = result (+bias) + residual   (transformer residual streams)
  int accumulate;
  const float* pro_scale;  // [C] or nullptr
  const float* pro_shift;
  int pro_relu;
  int out_relu;
  int act;     // 0 none; 2: y2 = z (pre-activation), y = gelu(z); 3: y = z ∘ gelu'(y2)   (y2: [M][ldy] of TO; needs 16-B rows)
               // 4: top-K filter (gallery match): nothing is stored; scores above the row's threshold key are appended
               //    to the row's candidate list  (y = cand u64 [M][cap], y2 = thrk u32 [M])
  void* y2;
  int* ccnt;   // act 4: candidate counters [M]
  int res_sub;                     // streaming join only: `residual` is compact [N][OH/2][OW/2][Cout], added at even (oh, ow)
  const unsigned char* res_mask;   // optional bit mask of `residual` ([M][ldy / KPACK] bytes, pfr_bn_act_mask): masked-out elements add 0
  int cap, col0, self_excl;   // act 4: list capacity, gallery index of column 0, skip column == row (all-vs-all evaluation)
  // act 4, persistent launch: L2-blocked tile order.  fo_qg > 0: the workgroups of one XCD (block b runs on XCD b % 8 — speed only) own a
  // contiguous range of gallery (column) tiles and walk it in super-steps of fo_qg query tiles x fo_gg gallery tiles (fo_qg * fo_gg =
  // workgroups per XCD), query group outermost: what an XCD works on at any time is a few MB of operands that stay in its L2, instead
  // of 256 different gallery tiles chip-wide per step (each gallery tile was then fetched from HBM / MALL once per query tile).
  int fo_qg = 0, fo_gg = 0;
  FastDiv div_ohow, div_ow;
  int tilesM, tilesN;
  int dma_sched = 0x101;   // -DPFR_IGEMM_SPREAD builds: parts | first k-group of waves 0-3 << 4 | of waves 4-7 << 8 (KNOB_IGEMM_DMA)
  int krot = 0;   // tile kernel, FAST non-parity-class path: first k-step of workgroup tile t = (t * krot) % nk (KNOB_IGEMM_KROT)
  // parity-class mode (data gradient of a stride-2 conv, FAST path): output rows are processed per (oh%2, ow%2) class so
  // that only the taps that exist for that class are visited (a 3x3/s2 dgrad does 9/4 instead of 9 taps per output).
  int pclass, mclass, tpc;
  FastDiv div_chw, div_cw;
  // BatchNorm-backward partial sums fused into the epilogue of the launch that PRODUCES the gradient g of a BN output
  // (replaces pfr_bn_bwd_reduce's pass over g and x): for up to two BN layers that consume g (a block's last BN and the BN
  // of its projection shortcut), x = that BN's input [M][ldy] of TO, coef = its [4][Cout] (mean, invstd, scale, shift),
  // part = [tilesM][2][Cout]: Σ g·mask and Σ g·mask·x̂ over the tile's rows.  mask: bit mask bnb_mask ([M][ldy / KPACK]) when
  // given, else scale·x + shift > 0 of set 0.
  int bnb_flags = 0;   // bit 0: store the gradient THROUGH bnb_mask (g*mask); bit 1: no input / coefficients for BN 0 (only sum g*mask)
  const void* bnb_x[2];
  const float* bnb_coef[2];
  float* bnb_part[2];
  const unsigned char* bnb_mask;
#ifdef PFR_IGEMM_TRACE
  long long* trace;   // [grid][8] wall-clock stamps of workgroup phases (profiling builds only)
  int dbg;            // 1: gather every tile from rows 0.. (L2-hot operands)   2: skip the output stores
#endif
};

// persistent kernel (pfr_igemm_p.hip): returns PFR_OK when it took the launch, 1 when the geometry is not eligible
int igemm_p_launch(IgemmParams& p, int dtype, int out_dtype, int bq, int bp, hipStream_t st);
int igemm_p_enabled();
int igemm_p_forced_tile();
// weight-stationary streaming kernel for HBM-bound 1x1 convolutions (pfr_sconv.hip): pfr_set_tuning("sconv"): 0 off,
// 1 heuristic (default), 2 whenever eligible.  sconv_try_launch returns 1 when it does not take the launch; sconv_mtile the rows per
// statistics partial of a post-op-free 1x1 launch it WOULD take (0: not its geometry) — both decide on geometry alone.
int sconv_mode();
int sconv_try_launch(IgemmParams& p, int dtype, int out_dtype, hipStream_t st);
int sconv_mtile(int M, int N, int K, long in_rows, int dtype, int out_dtype);
// halo-staged weight-stationary 3x3 kernel for 64 -> 64 channels (pfr_sconv3.hip); same conventions as sconv_*
bool sconv3_geom(int N, int H, int W, int C, int Cout, int R, int S, int stride, int pad, int idil_log2, int OH, int OW, int dtype,
                 int out_dtype, int* bpw);
int sconv3_try_launch(IgemmParams& p, int dtype, int out_dtype, hipStream_t st);
// halo-staged stem convolution, 4x4 / pad 2 over 16 channels -> 64 (pfr_sstem.hip); same conventions as sconv3_*
bool sstem_geom(int N, int H, int W, int C, int Cout, int R, int S, int stride, int pad, int idil_log2, int OH, int OW, int dtype,
                int out_dtype, int* bpw);
int sstem_try_launch(IgemmParams& p, int dtype, int out_dtype, hipStream_t st);
// streaming Linear kernel for K = 96 * j (pfr_slin.hip): returns 1 when it does not take the launch
int slin_try_launch(IgemmParams& p, int dtype, int out_dtype, hipStream_t st);
int slin_colsum_parts(IgemmParams& p, int dtype);
int slin_colsum_launch(IgemmParams& p, int dtype, float* colsum, hipStream_t st);
int num_cus();               // pfr_igemm_p.hip: CUs of the current device (256 when it cannot be asked)
int sconv_bnb_mode();
int sconv_bnb_parts(int M, int N, int K, int dtype);
// parity-class mode of the persistent kernel: data gradient of a stride-2 conv (input dilation 1 << 1) over even output sizes
static inline bool igemm_pclass_ok(const IgemmParams& p) {
  return p.idil_log2 == 1 && p.ostride == 1 && (p.OH % 2) == 0 && (p.OW % 2) == 0 && !p.stats_part;
}
