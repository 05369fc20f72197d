// Throughput of v_mad_u64_u32 by operand kind (gfx950): the accumulate-in-64-bit multiply-add every reduction kernel of
// this library is built from.  Variants: multiplier from an SGPR vs a VGPR, one or two distinct VGPR multiplicands,
// and the split form v_mul_lo_u32 + v_mul_hi_u32 + 64-bit add.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mad.hip -o tools/ubench_mad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048, ILP = 16;

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, const uint32_t* __restrict__ sv) {
  uint64_t acc[ILP];
  uint32_t x[4], y[4];
  const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
#pragma unroll
  for (int i = 0; i < ILP; i++) acc[i] = seed * (t + i);
#pragma unroll
  for (int i = 0; i < 4; i++) { x[i] = seed * (t + 17 + i) | 1; y[i] = seed * (t + 91 + i) | 3; }
  const uint32_t s0 = sv[0], s1 = sv[1], s2 = sv[2], s3 = sv[3];
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      if (OP == 0) acc[i] += (uint64_t)x[i & 3] * (i & 1 ? s0 : s1);              // VGPR x SGPR
      else if (OP == 1) acc[i] += (uint64_t)x[i & 3] * y[(i >> 2) & 3];            // VGPR x VGPR
      else if (OP == 2) acc[i] += (uint64_t)x[i & 3] * x[i & 3];                   // same VGPR twice
      else if (OP == 3) {                                                           // asm: force exact operand shape
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x[i & 3]), "v"(y[(i >> 2) & 3]) : "vcc");
      } else if (OP == 4) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x[i & 3]), "s"(i & 1 ? s2 : s3) : "vcc");
      } else if (OP == 5) {                                                         // mul_lo + mul_hi + add64
        const uint32_t lo = x[i & 3] * y[(i >> 2) & 3], hi = __umulhi(x[i & 3], y[(i >> 2) & 3]);
        acc[i] += ((uint64_t)hi << 32) | lo;
      }
    }
    // keep the multiplicands changing so nothing is hoisted
    x[it & 3] += 2; y[(it >> 1) & 3] += 4;
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) r ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32);
  out[t] = r;
}

template <int OP> static int run(const char* name, uint32_t* dout, uint32_t* dsv, int blocks) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, dout, 12345u, dsv); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, dout, 12345u + r, dsv);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double ops = (double)blocks * 256 * ITERS * ILP;
  const double per_wave_cycles = 2.4e9 * (ms * 1e-3) / (ops / 64 / (256.0 * 4));   // cycles per wave64 instruction per SIMD
  printf("%-44s %8.3f ms  %.3e lane-ops/s  %.2f cycles per wave64 op\n", name, ms, ops / (ms * 1e-3), per_wave_cycles);
  return 0;
}
int main() {
  uint32_t *dout, *dsv; const int blocks = 256 * 8;
  CHECK(hipMalloc(&dout, blocks * 256 * 4)); CHECK(hipMalloc(&dsv, 16));
  uint32_t h[4] = {3, 5, 7, 11}; CHECK(hipMemcpy(dsv, h, 16, hipMemcpyHostToDevice));
  if (run<0>("mad_u64: VGPR x SGPR (compiler)", dout, dsv, blocks)) return 1;
  if (run<1>("mad_u64: VGPR x VGPR (compiler)", dout, dsv, blocks)) return 1;
  if (run<2>("mad_u64: VGPR x same VGPR (compiler)", dout, dsv, blocks)) return 1;
  if (run<3>("mad_u64: VGPR x VGPR (asm)", dout, dsv, blocks)) return 1;
  if (run<4>("mad_u64: VGPR x SGPR (asm)", dout, dsv, blocks)) return 1;
  if (run<5>("mul_lo + mul_hi + add64", dout, dsv, blocks)) return 1;
  return 0;
}
