// LDS atomic throughput probe: 64-lane ds_add_f32 vs ds_add_u32 vs plain read-modify-write, distinct addresses with the
// bank pattern of the window-attention position-table scatter.   hipcc --offload-arch=gfx950 -O3 lds_atomic_probe.hip -o p && ./p
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int iters) {
  __shared__ float tabf[256];
  __shared__ unsigned tabu[256];
  __shared__ unsigned long long tabq[256];
  const int lane = threadIdx.x;
  for (int e = lane; e < 256; e += 64) { tabf[e] = 0.f; tabu[e] = 0u; tabq[e] = 0ull; }
  __syncthreads();
  const int i = lane & 31, yi = i / 7, xi = i % 7;
  const int base = (6 - yi) * 13 + (6 - xi) + (lane >> 5) * 4;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int idx = (base + e * 3) & 255;
      if (MODE == 0) atomicAdd(&tabf[idx], 1.0f + e);
      else if (MODE == 1) atomicAdd(&tabu[idx], 1u + e);
      else if (MODE == 3) atomicAdd(&tabq[idx], (unsigned long long)(1u + e));
      else tabf[idx] += 1.0f + e;   // (racy; timing only)
    }
  }
  __syncthreads();
  long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + lane] = tabf[lane] + (float)tabu[lane] + (float)tabq[lane];
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096 * 64 * 4); hipMalloc(&cyc, 4096 * 8);
  long long h[4096];
  for (int mode = 0; mode < 4; ++mode)
    for (int blocks : {1, 256 * 8}) {
      for (int r = 0; r < 2; ++r) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, cyc, 20);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, cyc, 20);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, cyc, 20);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, cyc, 20);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0; for (int b = 0; b < blocks; ++b) s += h[b];
      printf("mode %d (%s) blocks %5d: %.1f cycles (clock64 ticks) per 64-lane op\n", mode, mode == 0 ? "ds_add_f32" : mode == 1 ? "ds_add_u32" : mode == 3 ? "ds_add_u64" : "plain rmw", blocks, s / blocks / (20.0 * 32));
    }
  return 0;
}
