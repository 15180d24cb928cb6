// does v_dot2c_f32_bf16 (__builtin_amdgcn_fdot2_f32_bf16) equal the fp32 sum of the two exact products?  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* a, const unsigned* b, float* o0, float* o1) {
  const int i = threadIdx.x + blockIdx.x * blockDim.x;
  const bf16x2 x = __builtin_bit_cast(bf16x2, a[i]), y = __builtin_bit_cast(bf16x2, b[i]);
  o0[i] = __builtin_amdgcn_fdot2_f32_bf16(x, y, 1.5f, false);
  o1[i] = 1.5f + (float)x[0] * (float)y[0] + (float)x[1] * (float)y[1];
}
int main() {
  const int n = 4096;
  unsigned *a, *b, ha[n], hb[n];
  float *o0, *o1, h0[n], h1[n];
  srand(1);
  for (int i = 0; i < n; ++i) {
    float f[4];
    for (int j = 0; j < 4; ++j) f[j] = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    unsigned u[4];
    for (int j = 0; j < 4; ++j) { unsigned v; memcpy(&v, &f[j], 4); u[j] = v >> 16; }
    ha[i] = u[0] | (u[1] << 16);
    hb[i] = u[2] | (u[3] << 16);
  }
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&o0, n * 4); hipMalloc(&o1, n * 4);
  hipMemcpy(a, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb, n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(a, b, o0, o1);
  hipMemcpy(h0, o0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h1, o1, n * 4, hipMemcpyDeviceToHost);
  double me = 0; int bad = 0;
  for (int i = 0; i < n; ++i) { double e = fabs((double)h0[i] - h1[i]); if (e > me) me = e; if (e > 1e-5) ++bad; }
  printf("dot2 vs scalar: max |diff| %.3e, %d of %d differ by > 1e-5; sample %f %f\n", me, bad, n, h0[0], h1[0]);
  return 0;
}
