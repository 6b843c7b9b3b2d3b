#include <hip/hip_runtime.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int mode){
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for(int i=threadIdx.x;i<4096;i+=64) lds[i]=i;
  __syncthreads();
  int lane=threadIdx.x;
  int off;  // element offset
  if(mode==0) off = lane*4;                       // contiguous 8B per lane
  else off = (lane>>4)*256 + ((lane&15)>>2)*64 + (lane&3)*4;   // 4 rows x 16 cols blocks, row pitch 64 elems, group stride 256
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds+off));
  for(int j=0;j<4;j++) out[lane*4+j]=(unsigned short)r[j];
}
#include <stdio.h>
int main(){
  unsigned short* d; hipMalloc(&d, 64*4*2);
  unsigned short h[256];
  for(int mode=0;mode<2;mode++){
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for(int l=0;l<64;l++){ printf("lane %2d:", l); for(int j=0;j<4;j++) printf(" %4d", h[l*4+j]); printf("\n"); }
  }
  return 0;
}
