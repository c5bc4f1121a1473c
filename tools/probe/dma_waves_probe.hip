// How many waves must issue LDS-DMA concurrently to saturate a CU's L2->LDS path?  1 workgroup per CU, NW waves each issuing
// 1 KiB DMA instructions back to back (with a counted wait keeping ~8 in flight per wave); 64-byte and 128-byte row pieces.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int LPR>
__global__ void k(const char* __restrict__ src, int rowstride, int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) char smem[8 * 16384];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (size_t)(blockIdx.x % 64) * (256 * (size_t)rowstride);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 256 * rowstride, 0x00020000);
  constexpr int RPI = 64 / LPR;
  for (int it = 0; it < iters; ++it) {
    const int kofs = (it * LPR * 16) % rowstride;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = ((it * 8 + j) * RPI + lane / LPR) % 256;
      const uint32_t off = (uint32_t)(row * rowstride + kofs + (lane % LPR) * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 16384 + ((it & 1) * 8 + j) * 1024), 16, (int)off, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink && tid == 0) sink[blockIdx.x] = smem[lane];
}
int main() {
  char* src; float* sink;
  hipMalloc(&src, 64u << 20); hipMemset(src, 1, 64u << 20);
  hipMalloc(&sink, 1 << 16);
  const int iters = 4000;
  for (int lpr : {4, 8})
    for (int nw = 1; nw <= 8; ++nw)
      for (int wgs : {256, 512}) {
        if (wgs == 512 && nw > 4) continue;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(a);
          if (lpr == 4) hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(nw * 64), 0, 0, src, 512, rep ? iters : 20, sink);
          else hipLaunchKernelGGL(k<8>, dim3(wgs), dim3(nw * 64), 0, 0, src, 512, rep ? iters : 20, sink);
          hipEventRecord(b); hipDeviceSynchronize();
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)wgs * nw * iters * 8192.0;
        printf("%3d-B pieces  %d waves/WG x %d WG/CU: %6.2f TB/s = %5.1f B/clk/CU, %5.1f B/clk/wave (1 KiB per %5.0f clk per wave) @2.1GHz\n",
               lpr * 16, nw, wgs / 256, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.1e9, bytes / (ms * 1e-3) / 256 / 2.1e9 / (nw * wgs / 256),
               1024.0 / (bytes / (ms * 1e-3) / 256 / 2.1e9 / (nw * wgs / 256)));
      }
  return 0;
}
