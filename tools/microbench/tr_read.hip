#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 4 * 4];
  int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = in[i];
  __syncthreads();
  // lane l supplies address of 4 contiguous elements: lds + l*4
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + l * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short h[1024], o[256];
  for (int i = 0; i < 1024; ++i) h[i] = i;
  unsigned short *d, *e;
  hipMalloc(&d, 2048); hipMalloc(&e, 512);
  hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, 1, 64, 0, 0, d, e);
  hipMemcpy(o, e, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", o[l*4+j]); printf("\n"); }
  return 0;
}
