// Checks the operand / result lane maps of v_mfma_i32_32x32x32_i8 on gfx950 with exact integer data
// (the guide gives them for bf16 and says to check other dtypes).  Expected, by analogy with 32x32x16 bf16:
//   A: lane l holds A[row l&31][k = 16*(l>>5) + b], b = 0..15   (16 bytes = 4 VGPRs)
//   B: lane l holds B[k = 16*(l>>5) + b][col l&31]
//   D: lane l, reg g: D[row (g&3) + 8*(g>>2) + 4*(l>>5)][col l&31]
// Build: hipcc -O2 --offload-arch=gfx950 tools/mfma_i8_probe.hip -o tools/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const int8_t* A /*[32][32] row-major*/, const int8_t* B /*[32][32] k-major: B[k][n]*/, int* Dm /*[32][32]*/) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  union { int8_t b[16]; i32x4 v; } fa, fb;
  for (int b = 0; b < 16; b++) { fa.b[b] = A[r * 32 + 16 * h + b]; fb.b[b] = B[(16 * h + b) * 32 + r]; }
  i32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa.v, fb.v, acc, 0, 0, 0);
  for (int g = 0; g < 16; g++) Dm[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = acc[g];
}
int main() {
  int8_t hA[1024], hB[1024]; int hD[1024], ref[1024];
  srand(7);
  for (int i = 0; i < 1024; i++) { hA[i] = (int8_t)(rand() % 256 - 128); hB[i] = (int8_t)(rand() % 256 - 128); }
  hA[5] = -128; hB[7] = -128; hA[40] = 127; hB[99] = 127;
  for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) { int s = 0; for (int kk = 0; kk < 32; kk++) s += (int)hA[m * 32 + kk] * (int)hB[kk * 32 + n]; ref[m * 32 + n] = s; }
  int8_t *dA, *dB; int* dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; i++) bad += hD[i] != ref[i];
  printf("mfma_i32_32x32x32_i8 lane maps: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as expected", bad);
  return bad != 0;
}
