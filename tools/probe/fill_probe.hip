// L2 -> LDS fill-rate probe (gfx950): how many bytes per clock per CU can a workgroup pull from L2-resident memory into LDS
//   mode 0: LDS-DMA (buffer_load_dwordx4 ... lds), 64-byte row pieces (4 lanes per row, 16 rows per wave instruction)
//   mode 1: LDS-DMA, 128-byte row pieces (8 lanes per row, 8 rows per wave instruction)
//   mode 2: global_load_dwordx4 to VGPRs + ds_write_b128, 128-byte row pieces
//   mode 3: global_load_dwordx4 to VGPRs only (no LDS), 128-byte row pieces
//   mode 4: LDS-DMA, fully contiguous 1 KiB per wave instruction
// hipcc --offload-arch=gfx950 -O3 tools/probe/fill_probe.hip -o tools/probe/fill_probe && tools/probe/fill_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void fill_kernel(const char* __restrict__ src, size_t region, int rowstride, int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 16384];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (size_t)blockIdx.x % (region / (256 * (size_t)rowstride)) * (256 * (size_t)rowstride);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 256 * rowstride, 0x00020000);
  constexpr int LPR = (MODE == 0) ? 4 : 8;          // lanes per row piece
  constexpr int RPI = 64 / LPR;                      // rows per wave instruction
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const int kofs = (it * LPR * 16) % rowstride;    // walk along k like a GEMM k-loop
    char* slot = smem + (it & 3) * 16384;
    // one "stage" = 16 KiB = 16 wave instructions of 1 KiB, 4 per wave
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = ((j * 4 + wave) * RPI + lane / LPR) % 256;
      uint32_t off = (uint32_t)(row * rowstride + kofs + (lane % LPR) * 16);
      if (MODE == 4) off = (uint32_t)(((it * 16 + j * 4 + wave) * 1024) % (256 * rowstride - 1024) + lane * 16);
      if (MODE == 0 || MODE == 1 || MODE == 4) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(slot + (j * 4 + wave) * 1024), 16, (int)off, 0, 0, 0);
      } else {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
        if (MODE == 2) *reinterpret_cast<u32x4*>(slot + (j * 4 + wave) * 1024 + lane * 16) = v;
        else { acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3]; }
      }
    }
    if (MODE == 0 || MODE == 1 || MODE == 4) {
      if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // keep up to ~2 stages in flight
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && tid == 0) sink[blockIdx.x] = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) + smem[lane];
}

template <int MODE>
static void run(const char* name, const char* src, size_t region, int rowstride, int wgs, float* sink) {
  const int iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(fill_kernel<MODE>, dim3(wgs), dim3(256), 0, 0, src, region, rowstride, 50, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(fill_kernel<MODE>, dim3(wgs), dim3(256), 0, 0, src, region, rowstride, iters, sink);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)wgs * iters * 16384.0;
  printf("%-44s rowstride %5d  wgs %4d: %7.2f TB/s  = %5.1f B/clk/CU @2.1GHz\n", name, rowstride, wgs, bytes / ms / 1e9,
         bytes / (ms * 1e-3) / 256.0 / 2.1e9);
}

int main() {
  const size_t region = 24u << 20;   // L2/MALL resident
  char* src; float* sink;
  hipMalloc(&src, region + (1 << 20)); hipMemset(src, 1, region + (1 << 20));
  hipMalloc(&sink, 1 << 16);
  for (int wgs : {256, 512, 1024}) {
    for (int rs : {128, 512, 2048}) {
      run<0>("LDS-DMA 64-B row pieces", src, region, rs, wgs, sink);
      run<1>("LDS-DMA 128-B row pieces", src, region, rs, wgs, sink);
      run<2>("global_load + ds_write_b128 (128-B pieces)", src, region, rs, wgs, sink);
      run<3>("global_load only (128-B pieces)", src, region, rs, wgs, sink);
      run<4>("LDS-DMA contiguous 1 KiB", src, region, rs, wgs, sink);
    }
  }
  return 0;
}
