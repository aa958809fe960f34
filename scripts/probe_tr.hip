// Probe of ds_read_b64_tr_b16 lane/element mapping on gfx950 (diagnostic, not product code).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out, int stride) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int lane = threadIdx.x;
  int g = lane >> 4, t = lane & 15;
  // group g reads a 4x16 block: rows r = t>>2 (row stride `stride` elements), cols (t&3)*4..+4, block col base g*16
  short* addr = lds + (t >> 2) * stride + g * 16 + (t & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {64, 128}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d: expect lane l elem j = lds[j*stride + (l>>4)*16 + (l&15)]\n", stride);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (h[l*4+j] != j*stride + (l>>4)*16 + (l&15)) bad++;
    printf("mismatches: %d\n", bad);
    for (int l = 0; l < 20; ++l) printf("lane %2d: %d %d %d %d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
