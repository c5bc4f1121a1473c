// HBM -> LDS streaming probe (gfx950): what a PERSISTENT workgroup whose waves each own a private LDS-DMA ring can pull from
// HBM (source far larger than the 256 MiB Infinity Cache), as a function of the bytes a wave keeps in flight, with and without
// an output stream of 16-byte stores next to it.  Models the data movement of a weight-stationary 1x1 convolution:
//   per wave:  for block b in my blocks:  [wait block b landed] [issue DMA of block b+NS-1] [read block b from LDS] [store OUT bytes]
// hipcc --offload-arch=gfx950 -O3 tools/probe/stream_probe.hip -o tools/probe/stream_probe && tools/probe/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// BLK: KiB per block (= DMA instructions per block), NS: ring slots per wave, OUTK: KiB stored per block, NW: waves per workgroup
template <int BLK, int NS, int OUTK, int NW, int RS = 0>
__global__ __launch_bounds__(NW * 64, 1) void stream_kernel(const char* __restrict__ src, char* __restrict__ dst, long nblocks, int contiguous) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ring = smem + wave * (NS * BLK * 1024);
  const long nwaves = (long)gridDim.x * NW;
  const long me = (long)blockIdx.x * NW + wave;
  // block order: interleaved over all waves of the chip (contiguous == 0) or a private contiguous range per wave (1)
  const long per = (nblocks + nwaves - 1) / nwaves;
  auto blk_of = [&](long i) -> long { return contiguous ? me * per + i : me + i * nwaves; };
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7ffffff0, 0x00020000);
  auto issue = [&](long i, int slot) {
    const long b = blk_of(i);
    if (i >= per || b >= nblocks) {   // keep the instruction count uniform: out-of-range lanes read nothing (zeros are written)
#pragma unroll
      for (int j = 0; j < BLK; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + slot * BLK * 1024 + j * 1024), 16, (int)0x7ffffff8, 0, 0, 0);
      return;
    }
    uint32_t off = (uint32_t)(b * (BLK * 1024) % 0x70000000L);
    if (RS) {
      // strided rows (a [rows][RS bytes] matrix read in 128-byte k-chunks): block b = (row group b / (RS/128), k-chunk b % (RS/128));
      // an instruction covers 8 rows x 128 B at row stride RS
      const long rg = b / (RS / 128), kc = b % (RS / 128);
      off = (uint32_t)((rg * (BLK * 8) * RS + kc * 128) % 0x70000000L);
#pragma unroll
      for (int j = 0; j < BLK; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + slot * BLK * 1024 + j * 1024), 16,
                                                 (int)(off + (j * 8 + (lane >> 3)) * RS + (lane & 7) * 16), 0, 0, 0);
      return;
    }
#pragma unroll
    for (int j = 0; j < BLK; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + slot * BLK * 1024 + j * 1024), 16,
                                               (int)(off + j * 1024 + lane * 16), 0, 0, 0);
  };
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s, s);
  int slot = 0;
  for (long i = 0; i < per; ++i) {
    // outstanding, oldest first: DMA(i) DMA(i+1..i+NS-2) [stores of block i-1 were issued before DMA(i+NS-2)]
    issue(i + NS - 1, (slot + NS - 1) % NS);
    // now younger than DMA(i): (NS-1) blocks of DMA + the stores of block i-1
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(((NS - 1) * BLK + OUTK) > 63 ? 63 : ((NS - 1) * BLK + OUTK)) : "memory");
    const char* sb = ring + slot * BLK * 1024;
#pragma unroll
    for (int j = 0; j < BLK; ++j) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(sb + j * 1024 + lane * 16);
      acc[0] ^= v[0]; acc[1] += v[1]; acc[2] ^= v[2]; acc[3] += v[3];
    }
    const long b = blk_of(i);
    if (OUTK > 0 && b < nblocks) {
      char* o = dst + (size_t)b * (OUTK * 1024);
#pragma unroll
      for (int j = 0; j < OUTK; ++j) {
        u32x4 w = acc; w[0] += j;
        *reinterpret_cast<u32x4*>(o + j * 1024 + lane * 16) = w;
      }
    }
    slot = (slot + 1) % NS;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] == 0x12345678 && dst) dst[0] = 1;
}

template <int BLK, int NS, int OUTK, int NW, int RS = 0>
static void run(const char* src, char* dst, size_t src_bytes, int wg_per_cu, int contiguous) {
  const int lds = NW * NS * BLK * 1024;
  if (lds * wg_per_cu > 160 * 1024) return;
  auto k = stream_kernel<BLK, NS, OUTK, NW, RS>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const long nblocks = (long)(src_bytes / (BLK * 1024));
  const int grid = 256 * wg_per_cu;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, src, dst, nblocks / 8, contiguous);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, src, dst, nblocks, contiguous);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double rd = (double)nblocks * BLK * 1024, wr = (double)nblocks * OUTK * 1024;
  printf("rs %4d blk %2d KiB ring %d out %2d KiB waves/wg %d wg/cu %d %s: in-flight/CU %3d KiB  %6.1f us  read %5.2f TB/s  read+write %5.2f TB/s\n", RS, BLK, NS, OUTK, NW,
         wg_per_cu, contiguous ? "contig" : "interl", (NS - 1) * BLK * NW * wg_per_cu, ms * 1e3, rd / ms / 1e9, (rd + wr) / ms / 1e9);
}

int main() {
  const size_t src_bytes = (size_t)1536 << 20, dst_bytes = (size_t)6 << 30;
  char *src, *dst;
  hipMalloc(&src, src_bytes + (1 << 20)); hipMemset(src, 1, src_bytes);
  hipMalloc(&dst, dst_bytes);
  hipMemset(dst, 0, dst_bytes);
  for (int c = 0; c < 2; ++c) {
    run<8, 2, 0, 4>(src, dst, src_bytes, 1, c);
    run<8, 3, 0, 4>(src, dst, src_bytes, 1, c);
    run<8, 4, 0, 4>(src, dst, src_bytes, 1, c);
    run<8, 2, 0, 4>(src, dst, src_bytes, 2, c);
    run<8, 3, 0, 4>(src, dst, src_bytes, 2, c);
    run<16, 2, 0, 4>(src, dst, src_bytes, 1, c);
    run<16, 3, 0, 4>(src, dst, src_bytes, 1, c);
    run<32, 2, 0, 4>(src, dst, src_bytes, 1, c);
    run<8, 3, 0, 8>(src, dst, src_bytes, 1, c);
    run<4, 4, 0, 8>(src, dst, src_bytes, 1, c);
    run<4, 3, 0, 8>(src, dst, src_bytes, 2, c);
  }
  // with an output stream: out/in = 1/4 (256->64), 1 (64->64), 4 (64->256)
  run<8, 3, 2, 4>(src, dst, src_bytes, 1, 0);
  run<8, 4, 2, 4>(src, dst, src_bytes, 1, 0);
  run<32, 2, 8, 4>(src, dst, src_bytes, 1, 0);
  run<8, 3, 8, 4>(src, dst, src_bytes, 1, 0);
  run<8, 4, 8, 4>(src, dst, src_bytes, 1, 0);
  run<8, 3, 32, 4>(src, dst, src_bytes, 1, 0);
  run<8, 4, 32, 4>(src, dst, src_bytes, 1, 0);
  run<4, 4, 16, 8>(src, dst, src_bytes, 1, 0);
  run<8, 3, 8, 4>(src, dst, src_bytes, 1, 1);
  run<8, 3, 32, 4>(src, dst, src_bytes, 1, 1);
  // the streaming kernel's shapes: 8 waves, 4 KiB granules (32 rows x 128 B), row stride 128 (K=64) / 512 (K=256) / 1024 (K=512)
  run<4, 4, 0, 8, 0>(src, dst, src_bytes, 1, 0);
  run<4, 4, 0, 8, 512>(src, dst, src_bytes, 1, 0);
  run<4, 4, 0, 8, 1024>(src, dst, src_bytes, 1, 0);
  run<4, 3, 1, 8, 512>(src, dst, src_bytes, 1, 0);
  run<4, 4, 1, 8, 512>(src, dst, src_bytes, 1, 0);
  run<4, 4, 1, 8, 0>(src, dst, src_bytes, 1, 0);
  run<4, 2, 1, 8, 1024>(src, dst, src_bytes, 1, 0);
  run<4, 4, 16, 8, 0>(src, dst, src_bytes, 1, 0);
  run<4, 4, 4, 8, 0>(src, dst, src_bytes, 1, 0);
  run<4, 4, 1, 8, 512>(src, dst, src_bytes, 1, 1);
  return 0;
}
