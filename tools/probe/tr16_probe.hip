// Probe: documents the lane/element mapping of ds_read_b64_tr_b16 on gfx950 (used by the wgrad kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out, int mode){
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  // mode 0: lane-linear addresses (lane l -> elements [4l,4l+4))
  // mode 1: [4 rows][16 cols] block per 16-lane group with row stride 160 elements (320 B)
  int off;
  if (mode == 0) off = 4 * l;
  else { int g = l >> 4, s = l & 15; off = (s >> 2) * 160 + (g & 1) * 16 + (s & 3) * 4 + (g >> 1) * 8 * 160; }
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + off));
  for (int j=0;j<4;j++) out[l*4+j] = (unsigned short)v[j];
}
int main(){
  unsigned short h[4096], o[256]; for (int i=0;i<4096;i++) h[i]=i;
  unsigned short *di,*dout; hipMalloc(&di,sizeof(h)); hipMalloc(&dout,sizeof(o));
  hipMemcpy(di,h,sizeof(h),hipMemcpyHostToDevice);
  for (int mode=0; mode<2; ++mode){
    hipLaunchKernelGGL(k,1,64,0,0,di,dout,mode); hipMemcpy(o,dout,sizeof(o),hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l=0;l<64;l++){ printf("lane %2d:", l); for(int j=0;j<4;j++) printf(" %5d", o[l*4+j]); printf("\n"); }
  }
  return 0;
}
