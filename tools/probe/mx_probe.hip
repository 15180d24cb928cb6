// Lane/byte layout probe for v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (e4m3) operands: tries layout hypotheses against an exact
// integer reference (values in {-2..2}, exactly representable in e4m3) and reports which one matches; then checks the E8M0 scales.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k2(const uint8_t* A, const uint8_t* B, float* C, int hyp) {
  // A's block scale of (row i, hardware block = lane >> 4) is 127 + ((i + kb) % 3); B's scales are 1.  The result equals the
  // reference only if the k's this lane loads are the k's the hardware scales with this lane's byte.
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  uint8_t ab[32], bb[32];
  for (int t = 0; t < 32; ++t) {
    int kk;
    if (hyp == 0) kk = 32 * g + t;
    else if (hyp == 1) kk = (t < 16) ? 16 * g + t : 64 + 16 * g + (t - 16);
    else kk = (t >> 3) * 32 + 8 * g + (t & 7);
    ab[t] = A[r * 128 + kk];
    bb[t] = B[r * 128 + kk];
  }
  v8i a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = ab[4 * i] | (ab[4 * i + 1] << 8) | (ab[4 * i + 2] << 16) | (ab[4 * i + 3] << 24);
    b[i] = bb[4 * i] | (bb[4 * i + 1] << 8) | (bb[4 * i + 2] << 16) | (bb[4 * i + 3] << 24);
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127 + ((r + g) % 3), 0, 127);
  for (int e = 0; e < 4; ++e) C[(4 * g + e) * 16 + r] = c[e];
}

__global__ void k(const uint8_t* A, const uint8_t* B, float* C, int hyp, int sa, int sb, int opa, int opb) {
  // A [16][128] row-major fp8 bytes, B [16 cols][128 k] (i.e. Bt: B[k][j] stored at Bt[j][k]), C [16][16]
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  uint8_t ab[32], bb[32];
  for (int t = 0; t < 32; ++t) {
    int kk;
    if (hyp == 0) kk = 32 * g + t;                                 // 32 consecutive k per lane
    else if (hyp == 1) kk = (t < 16) ? 16 * g + t : 64 + 16 * g + (t - 16);   // two K=64 halves, 16 consecutive each
    else kk = (t >> 3) * 32 + 8 * g + (t & 7);                      // four K=32 quarters, 8 consecutive each
    ab[t] = A[r * 128 + kk];
    bb[t] = B[r * 128 + kk];
  }
  v8i a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = ab[4 * i] | (ab[4 * i + 1] << 8) | (ab[4 * i + 2] << 16) | (ab[4 * i + 3] << 24);
    b[i] = bb[4 * i] | (bb[4 * i + 1] << 8) | (bb[4 * i + 2] << 16) | (bb[4 * i + 3] << 24);
  }
  f32x4 c = {0, 0, 0, 0};
  if (opa == 0 && opb == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  else if (opa == 1 && opb == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, sa, 2, sb);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 3, sa, 3, sb);
  for (int e = 0; e < 4; ++e) C[(4 * g + e) * 16 + r] = c[e];      // C/D: col = lane & 15, row = 4 (lane >> 4) + e
}

static uint8_t enc(int v) {   // small integers in e4m3: 0, ±1 (0x38), ±2 (0x40), ±0.5 (0x30)
  uint8_t m = 0;
  int a = abs(v);
  if (a == 1) m = 0x38; else if (a == 2) m = 0x40; else if (a == 3) m = 0x44; else if (a == 4) m = 0x48;
  return v < 0 ? (m | 0x80) : m;
}

int main() {
  uint8_t hA[16 * 128], hB[16 * 128];
  int iA[16 * 128], iB[16 * 128];
  srand(7);
  for (int i = 0; i < 16 * 128; ++i) { iA[i] = rand() % 9 - 4; iB[i] = rand() % 9 - 4; hA[i] = enc(iA[i]); hB[i] = enc(iB[i]); }
  float ref[256];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int s = 0; for (int kk = 0; kk < 128; ++kk) s += iA[i * 128 + kk] * iB[j * 128 + kk]; ref[i * 16 + j] = (float)s; }
  uint8_t *dA, *dB; float* dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 1024);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  float hC[256];
  for (int hyp = 0; hyp < 3; ++hyp) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, hyp, 0x7f7f7f7f, 0x7f7f7f7f, 0, 0);
    hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += hC[i] != ref[i];
    printf("hypothesis %d: %d / 256 mismatches (C[0][0] %.1f ref %.1f, C[3][5] %.1f ref %.1f)\n", hyp, bad, hC[0], ref[0], hC[3 * 16 + 5], ref[3 * 16 + 5]);
  }
  // which k's does a lane's scale byte cover?  reference under "lane (row, kb) scales k = 32 kb … 32 kb + 31 of its row"
  for (int which = 0; which < 2; ++which) {      // which = 0: the scaled operand is the FIRST builtin operand (rows of C = register index?) — try both roles
    float ref2[256];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      float s = 0; for (int kb = 0; kb < 4; ++kb) { int t = 0; for (int kk = 32 * kb; kk < 32 * kb + 32; ++kk) t += iA[i * 128 + kk] * iB[j * 128 + kk]; s += (float)t * (float)(1 << ((i + kb) % 3)); }
      ref2[which == 0 ? i * 16 + j : j * 16 + i] = s; }
    for (int hyp = 0; hyp < 3; ++hyp) {
      hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, dA, dB, dC, hyp);
      hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
      int bad = 0; for (int i = 0; i < 256; ++i) bad += hC[i] != ref2[i];
      printf("per-block scales, C[%s], data hypothesis %d: %d / 256 mismatches\n", which == 0 ? "i][j" : "j][i", hyp, bad);
    }
  }
  // scales: E8M0 bytes; opsel picks the byte of the 32-bit scale operand
  struct { int sa, sb, opa, opb; const char* what; } t[] = {
    {0x7f7f7f80, 0x7f7f7f7f, 0, 0, "scaleA byte0 = 128 (x2), opsel 0"},
    {0x7f7f807f, 0x7f7f7f7f, 0, 0, "scaleA byte1 = 128, opsel 0 (expect x1)"},
    {0x7f7f807f, 0x7f817f7f, 1, 2, "scaleA byte1 = 128 opselA 1, scaleB byte2 = 129 opselB 2 (expect x8)"},
    {(int)0x7e7f7f7f, (int)0x7d7f7f7f, 3, 3, "byte3: A 126, B 125, opsel 3 (expect x1/8)"}};
  for (auto& c : t) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, 0, c.sa, c.sb, c.opa, c.opb);
    hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
    printf("%s: C[3][5] / ref = %.4f, C[9][2] / ref = %.4f\n", c.what, hC[3 * 16 + 5] / ref[3 * 16 + 5], hC[9 * 16 + 2] / ref[9 * 16 + 2]);
  }
  return 0;
}
