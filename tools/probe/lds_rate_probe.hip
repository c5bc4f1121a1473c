// Probe: sustained rate of LDS fragment reads with 8 waves per CU (2 per SIMD), no MFMA: ds_read_b64_tr_b16 / ds_read_b64 / ds_read_b128.
// hipcc --offload-arch=gfx950 -O3 lds_rate_probe.hip -o lds_rate_probe && ./lds_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k(unsigned* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid * 16; i < 65536; i += NW * 64 * 16) *(u32x4*)(smem + i) = u32x4{1u, 2u, 3u, 4u};
  __syncthreads();
  const int g = lane >> 4, s4 = lane & 15, r0 = (g >> 1) * 8 + (s4 >> 2);
  unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + r0 * 128 + ((((g & 1)) ^ (((r0 >> 1) & 1) << 1)) << 5) + (s4 & 3) * 8 + (tid >> 6) * 2048;
  if (MODE == 2) addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (lane & 31) * 128 + ((((lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4) + (tid >> 6) * 4096;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    u32x2 v[16]; u32x4 w[8];
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"((j & 7) * 512 + (j >> 3) * 16384));
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"((j & 7) * 512 + (j >> 3) * 16384));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[j]) : "v"(addr), "n"(j * 32768 / 8 % 32768));
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (MODE < 2) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { asm volatile("" : "+v"(v[j])); acc += v[j][0]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { asm volatile("" : "+v"(w[j])); acc += w[j][0]; }
    }
  }
  if (acc == 12345u) out[0] = acc;
}
template <int MODE, int NW> void run(const char* name, int nper) {
  unsigned* d; hipMalloc(&d, 4);
  hipFuncSetAttribute((const void*)k<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 65536);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4000;
  k<MODE, NW><<<256, NW * 64, 131072>>>(d, 100);
  hipEventRecord(a); k<MODE, NW><<<256, NW * 64, 131072>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double reads = (double)iters * nper * NW;   // wave-instructions per CU
  printf("%-22s waves %d: %.3f ms, %.2f ns per wave-instruction per CU (%.1f cycles at 2.4 GHz), %.0f B/ns/CU\n", name, NW, ms, ms * 1e6 / reads,
         ms * 1e6 / reads * 2.4, (MODE == 2 ? 1024.0 : 512.0) * reads / (ms * 1e6));
}
int main() {
  run<0, 8>("ds_read_b64_tr_b16", 16); run<0, 4>("ds_read_b64_tr_b16", 16); run<0, 16>("ds_read_b64_tr_b16", 16);
  run<1, 8>("ds_read_b64", 16); run<2, 8>("ds_read_b128", 8); run<2, 4>("ds_read_b128", 8);
  return 0;
}
