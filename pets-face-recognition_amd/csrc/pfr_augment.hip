// pfr_augment.hip — the reference's train-time image augmentation on the device, bit-exact with the PIL pipeline it runs
// in its dataloader workers.
//
// Reference semantics replaced (/root/reference/configs/dog_fe/fe_dogs_config.py:17-26):
//   ToPILImage → RandomAdjustSharpness(0, p=0.1) → RandomAutocontrast(p=0.3) → RandomCrop((220, 220)) →
//   Resize((224, 224)) → RandomRotation(5) → ToTensor
// torchvision's PIL-image transforms are calls into Pillow (requirements.txt:3,6); the arithmetic below is Pillow's:
//   sharpness 0    ImageFilter.SMOOTH: 3x3 (1,1,1,1,5,1,1,1,1)/13, 1-pixel frame copied, +0.5 and truncate
//   autocontrast   per band lo/hi of the histogram → lut[i] = clip(int(i*scale + offset)), scale = 255.0/(hi-lo) (doubles)
//   resize         Resample.c: 22-bit fixed-point triangle filter, horizontal pass to 8 bits, then vertical pass
//   rotate         Geometry.c affine_fixed: 16.16 fixed-point inverse map, nearest sample, fill 0
//   ToTensor       byte / 255 in float32, CHW
// All of it is integer / byte work bound by HBM traffic (150 KB per image), so there are two plain passes:
//   aug_pre   one workgroup per image that needs it (≈37 % of a batch): the blurred copy and/or the per-band lo/hi
//   aug_post  one thread per OUTPUT pixel: inverse-rotate → the ≤ ksize² taps of the two resize passes, read through the
//             autocontrast LUT (LDS) from the blurred or original image at the crop offset → float32 NCHW.
// The crop, both resize passes, the rotation and ToTensor never touch HBM in between.  Random decisions are inputs
// (device int32 [N][12] records built by pfr_augment_params on the host).
#include "pfr_common.h"
#include <math.h>
#include <mutex>
#include <vector>

#define AUG_PREC 22     // Resample.c PRECISION_BITS = 32 - 8 - 2
#define AUG_MAXK 9      // taps per output sample supported (ksize = 2*ceil(max(scale,1)) + 1 → down-scaling up to 4x)
#define AUG_SLABS 8     // row slabs per image in aug_pre (one workgroup each); lo/hi partials are merged by aug_post
#define AUG_REC 12      // ints per image record: sharp, contrast, top, left, rot_valid, a0, a1, a2, a3, a4, a5, pad

// ---- host: coefficient tables (Resample.c precompute_coeffs + normalize_coeffs_8bpc, bilinear) ---------------------------
struct CoefTable { int in, out, ksize; int* dev; };   // dev: [out][2 + ksize] = (first tap, taps, k...)
static std::mutex g_coef_mu;
static std::vector<CoefTable> g_coef;

static int build_coef(int in_size, int out_size, std::vector<int>& tab) {
  double scale = (double)((float)in_size - 0.0f) / out_size, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  if (ksize > AUG_MAXK) return -1;
  tab.assign((size_t)out_size * (2 + ksize), 0);
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale, ss = 1.0 / filterscale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      double t = (x + xmin - center + 0.5) * ss;
      if (t < 0.0) t = -t;
      const double w = t < 1.0 ? 1.0 - t : 0.0;
      k[x] = w;
      ww += w;
    }
    int* row = &tab[(size_t)xx * (2 + ksize)];
    row[0] = xmin;
    row[1] = xmax;
    for (int x = 0; x < xmax; ++x) {
      const double v = ww != 0.0 ? k[x] / ww : k[x];
      row[2 + x] = v < 0 ? (int)(-0.5 + v * (1 << AUG_PREC)) : (int)(0.5 + v * (1 << AUG_PREC));
    }
  }
  return ksize;
}

static bool get_coef(int in_size, int out_size, CoefTable* out) {
  std::lock_guard<std::mutex> lk(g_coef_mu);
  for (auto& c : g_coef)
    if (c.in == in_size && c.out == out_size) { *out = c; return true; }
  std::vector<int> tab;
  const int ks = build_coef(in_size, out_size, tab);
  if (ks < 0) return false;
  CoefTable c{in_size, out_size, ks, nullptr};
  if (hipMalloc(&c.dev, tab.size() * sizeof(int)) != hipSuccess) return false;
  if (hipMemcpy(c.dev, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return false;
  g_coef.push_back(c);
  *out = c;
  return true;
}

// ---- host: decision records --------------------------------------------------------------------------------------------
// Image.rotate(angle, NEAREST, expand=False, center=None) → Image.transform(AFFINE) → Geometry.c affine_fixed
static double py_round15(double v) {
  // Python round(v, 15) for |v| <= 1: correctly rounded decimal → nearest double (float.__round__ goes through a
  // 17-significant-digit string as well)
  char buf[64];
  snprintf(buf, sizeof buf, "%.15f", v);
  return strtod(buf, nullptr);
}
static int fix16(double v) {
  v = v * 65536.0 + 0.5;
  return v >= 0.0 ? (int)v : (int)floor(v);
}

extern "C" int pfr_augment_params(const int* flags, const float* angles, int N, int out_w, int out_h, int* records) {
  PFR_CHECK_ARG(flags && angles && records && N > 0, "pfr_augment_params: bad args");
  for (int i = 0; i < N; ++i) {
    int* r = records + (size_t)i * AUG_REC;
    for (int j = 0; j < 4; ++j) r[j] = flags[i * 4 + j];
    for (int j = 4; j < AUG_REC; ++j) r[j] = 0;
    double angle = fmod((double)angles[i], 360.0);
    if (angle < 0.0) angle += 360.0;          // Python's % has the sign of the divisor
    if (angle == 0.0) continue;               // Image.rotate returns a copy
    if (angle == 180.0 || ((angle == 90.0 || angle == 270.0) && out_w == out_h)) {
      pfr_set_error("pfr_augment_params: right-angle rotation (Image.transpose path) is outside RandomRotation(5)");
      return PFR_ERR_UNSUPPORTED;
    }
    const double cx = out_w / 2.0, cy = out_h / 2.0, a = -(angle * (M_PI / 180.0));   // math.radians = x * (pi/180)
    double m[6] = {py_round15(cos(a)), py_round15(sin(a)), 0.0, py_round15(-sin(a)), py_round15(cos(a)), 0.0};
    m[2] = m[0] * -cx + m[1] * -cy + m[2];
    m[5] = m[3] * -cx + m[4] * -cy + m[5];
    m[2] += cx;
    m[5] += cy;
    r[4] = 1;
    r[5] = fix16(m[0]);
    r[6] = fix16(m[1]);
    r[7] = fix16(m[2] + m[0] * 0.5 + m[1] * 0.5);
    r[8] = fix16(m[3]);
    r[9] = fix16(m[4]);
    r[10] = fix16(m[5] + m[3] * 0.5 + m[4] * 0.5);
  }
  return PFR_OK;
}

// ---- device ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_min_i(int v) {
  for (int o = 32; o; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
  for (int o = 32; o; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// blurred copy (sharp) and per-band lo / hi (contrast) of one row slab of one image;
// lohi [N][AUG_SLABS][8] = lo0, hi0, lo1, hi1, lo2, hi2 of the slab
__global__ __launch_bounds__(1024) void aug_pre_kernel(const uint8_t* __restrict__ x, int H, int W, const int* __restrict__ rec,
                                                       uint8_t* __restrict__ blur, int* __restrict__ lohi) {
  const int n = blockIdx.x;
  const int sharp = rec[n * AUG_REC + 0], contrast = rec[n * AUG_REC + 1];
  if (!sharp && !contrast) return;
  const uint8_t* src = x + (size_t)n * H * W * 3;
  uint8_t* dst = blur + (size_t)n * H * W * 3;
  int lo[3] = {255, 255, 255}, hi[3] = {0, 0, 0};
  const int rowb = W * 3;
  const int rows = (H + AUG_SLABS - 1) / AUG_SLABS, r0 = blockIdx.y * rows, r1 = min(H, r0 + rows);
  for (int i = r0 * rowb + threadIdx.x; i < r1 * rowb; i += 1024) {
    const int yy = i / rowb, xb = i - yy * rowb, xx = xb / 3, c = xb - xx * 3;
    int v = src[i];
    if (sharp) {
      if (yy > 0 && yy < H - 1 && xx > 0 && xx < W - 1 && H >= 3 && W >= 3) {
        int s = 4 * v;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -3; dx <= 3; dx += 3) s += src[i + dy * rowb + dx];
        v = (2 * s + 13) / 26;          // = (UINT8)(S/13 + 0.5) of Filter.c (S/13 + 0.5 is never within 0.038 of an integer)
        v = v > 255 ? 255 : v;
      }
      dst[i] = (uint8_t)v;
    }
    lo[c] = min(lo[c], v);
    hi[c] = max(hi[c], v);
  }
  if (!contrast) return;
  __shared__ int red[16][6];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int a = wave_min_i(lo[c]), b = wave_max_i(hi[c]);
    if (lane == 0) { red[wv][2 * c] = a; red[wv][2 * c + 1] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    int v = red[0][threadIdx.x];
    for (int w = 1; w < 16; ++w) v = (threadIdx.x & 1) ? max(v, red[w][threadIdx.x]) : min(v, red[w][threadIdx.x]);
    lohi[(n * AUG_SLABS + blockIdx.y) * 8 + threadIdx.x] = v;
  }
}

// ImageOps.autocontrast's LUT entry in Python-double arithmetic.  The reference rounds the product before the add; the
// library is built with -ffp-contract=fast (which ignores contraction pragmas), so the products are pinned in registers by
// empty asm statements to keep the compiler from forming an fma.
__device__ __forceinline__ uint8_t autocontrast_lut(int ix, int lo, int hi) {
  if (hi <= lo) return (uint8_t)ix;
  const double scale = 255.0 / (double)(hi - lo);
  double offset = (double)(-lo) * scale;
  double prod = (double)ix * scale;
  asm volatile("" : "+v"(offset), "+v"(prod));
  const int v = (int)(prod + offset);
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ __launch_bounds__(256) void aug_post_kernel(const uint8_t* __restrict__ x, const uint8_t* __restrict__ blur, int H, int W,
                                                       int crop_h, int crop_w, int out_h, int out_w, const int* __restrict__ rec,
                                                       const int* __restrict__ lohi, const int* __restrict__ cx, int ksx,
                                                       const int* __restrict__ cy, int ksy, float* __restrict__ y) {
  __shared__ uint8_t lut[3][256];
  const int n = blockIdx.y;
  const int* r = rec + n * AUG_REC;
  const int sharp = r[0], contrast = r[1], top = r[2], left = r[3];
  if (contrast) {
    for (int i = threadIdx.x; i < 768; i += 256) {
      const int c = i >> 8;
      int lo = 255, hi = 0;
#pragma unroll
      for (int sl = 0; sl < AUG_SLABS; ++sl) {
        lo = min(lo, lohi[(n * AUG_SLABS + sl) * 8 + 2 * c]);
        hi = max(hi, lohi[(n * AUG_SLABS + sl) * 8 + 2 * c + 1]);
      }
      lut[c][i & 255] = autocontrast_lut(i & 255, lo, hi);
    }
    __syncthreads();
  }
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= out_h * out_w) return;
  const int oy = p / out_w, ox = p - oy * out_w;
  int sx = ox, sy = oy;
  bool inside = true;
  if (r[4]) {
    sx = (r[7] + r[5] * ox + r[6] * oy) >> 16;
    sy = (r[10] + r[8] * ox + r[9] * oy) >> 16;
    inside = sx >= 0 && sx < out_w && sy >= 0 && sy < out_h;
  }
  int res[3] = {0, 0, 0};
  if (inside) {
    const uint8_t* src = (sharp ? blur : x) + (size_t)n * H * W * 3;
    const uint8_t* img_end = src + (size_t)H * W * 3;
    // horizontal pass first (8-bit intermediate), then vertical (Resample.c ImagingResampleInner); an axis whose size
    // does not change is not resampled at all
    const bool rx = crop_w != out_w, ry = crop_h != out_h;
    const int* kx = cx + (size_t)sx * (2 + ksx);
    const int* ky = cy + (size_t)sy * (2 + ksy);
    const int x0 = rx ? kx[0] : sx, nx = rx ? kx[1] : 1;
    const int y0 = ry ? ky[0] : sy, ny = ry ? ky[1] : 1;
    int acc[3] = {1 << (AUG_PREC - 1), 1 << (AUG_PREC - 1), 1 << (AUG_PREC - 1)};
    for (int ty = 0; ty < ny; ++ty) {
      const uint8_t* row = src + ((size_t)(top + y0 + ty) * W + left + x0) * 3;
      int h[3];
      if (rx) {
        int a[3] = {1 << (AUG_PREC - 1), 1 << (AUG_PREC - 1), 1 << (AUG_PREC - 1)};
        if (nx <= 4 && row + 12 <= img_end) {
          // up to 4 neighbouring pixels = 12 contiguous bytes: three (unaligned) dword loads instead of 12 byte loads
          uint32_t w[3];
          __builtin_memcpy(w, row, 12);
          const int k0 = kx[2], k1 = nx > 1 ? kx[3] : 0, k2 = nx > 2 ? kx[4] : 0, k3 = nx > 3 ? kx[5] : 0;
          const int kk[4] = {k0, k1, k2, k3};
#pragma unroll
          for (int tx = 0; tx < 4; ++tx)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const int b = tx * 3 + c;
              int v = (w[b >> 2] >> ((b & 3) * 8)) & 255;
              if (contrast) v = lut[c][v];
              a[c] += v * kk[tx];
            }
        } else {
          for (int tx = 0; tx < nx; ++tx) {
            const int k = kx[2 + tx];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              int v = row[tx * 3 + c];
              if (contrast) v = lut[c][v];
              a[c] += v * k;
            }
          }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int v = a[c] >> AUG_PREC;
          h[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          int v = row[c];
          h[c] = contrast ? lut[c][v] : v;
        }
      }
      if (ry) {
        const int k = ky[2 + ty];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += h[c] * k;
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) res[c] = h[c];
      }
    }
    if (ry) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int v = acc[c] >> AUG_PREC;
        res[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
      }
    }
  }
  const size_t plane = (size_t)out_h * out_w;
  float* yo = y + (size_t)n * 3 * plane + p;
#pragma unroll
  for (int c = 0; c < 3; ++c) yo[c * plane] = (float)res[c] / 255.0f;   // ToTensor: float32 division
}

extern "C" long pfr_augment_ws_bytes(int N, int H, int W) { return (long)N * H * W * 3 + (long)N * AUG_SLABS * 8 * 4 + 256; }

extern "C" int pfr_augment_train(const unsigned char* x, int N, int H, int W, int crop_h, int crop_w, int out_h, int out_w,
                                 const int* records, float* y, void* ws, hipStream_t st) {
  PFR_CHECK_ARG(x && records && y && ws, "pfr_augment_train: null pointer");
  PFR_CHECK_ARG(N > 0 && N <= 65535 && crop_h > 0 && crop_w > 0 && crop_h <= H && crop_w <= W && out_h > 0 && out_w > 0,
                "pfr_augment_train: bad sizes (N=%d %dx%d crop %dx%d out %dx%d)", N, H, W, crop_h, crop_w, out_h, out_w);
  CoefTable tx, ty;
  if (!get_coef(crop_w, out_w, &tx) || !get_coef(crop_h, out_h, &ty)) {
    pfr_set_error("pfr_augment_train: resize %dx%d -> %dx%d needs more than %d taps or table allocation failed", crop_h, crop_w,
                  out_h, out_w, AUG_MAXK);
    return PFR_ERR_UNSUPPORTED;
  }
  uint8_t* blur = (uint8_t*)ws;
  int* lohi = (int*)((uint8_t*)ws + (((size_t)N * H * W * 3 + 255) & ~(size_t)255));
  hipLaunchKernelGGL(aug_pre_kernel, dim3(N, AUG_SLABS), dim3(1024), 0, st, x, H, W, records, blur, lohi);
  PFR_CHECK_LAUNCH();
  hipLaunchKernelGGL(aug_post_kernel, dim3((out_h * out_w + 255) / 256, N), dim3(256), 0, st, x, blur, H, W, crop_h, crop_w, out_h,
                     out_w, records, lohi, tx.dev, tx.ksize, ty.dev, ty.ksize, y);
  PFR_CHECK_LAUNCH();
  return PFR_OK;
}
