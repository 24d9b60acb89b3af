#include <cstdio>
#include <hip/hip_runtime.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out){
  __shared__ short lds[1024];
  for(int i=threadIdx.x;i<1024;i+=64) lds[i]=i;
  __syncthreads();
  int l=threadIdx.x;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + (l>>4)*64 + (l&15)*4));
  for(int j=0;j<4;j++) out[l*4+j]=v[j];
}
int main(){
  short* d; hipMalloc(&d, 256*2); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for(int l=0;l<64;l++){ printf("lane %2d:", l); for(int j=0;j<4;j++) printf(" %4d", h[l*4+j]); printf("\n"); }
  return 0;
}
