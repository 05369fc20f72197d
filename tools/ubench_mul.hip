// Issue cost of the 32-bit multiply flavours on gfx950 and of two forms of the Shoup butterfly multiply
// (y*w mod p with the companion wp = floor(w 2^32 / p)) used by the NTT kernels.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mul.hip -o tools/ubench_mul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048, ILP = 16;

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, const uint32_t* __restrict__ sv) {
  uint32_t a[ILP];
  const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
#pragma unroll
  for (int i = 0; i < ILP; i++) a[i] = seed * (t + i) | 1;
  const uint32_t w = sv[0], wp = sv[1], p = sv[2], np = 0u - p;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(w));
      else if (OP == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "s"(w));
      else if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "s"(w));
      else if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "s"(w));
      else if (OP == 4) {   // Shoup, as in zq_device.h
        const uint32_t qh = __umulhi(a[i], wp);
        a[i] = a[i] * w - qh * p;
      } else if (OP == 5) { // the same through v_mad_u64_u32 (hi of y*wp; lo of y*w + qh*(-p))
        uint64_t h, r;
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(h) : "v"(a[i]), "s"(wp) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a[i]), "s"(w) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"((uint32_t)(h >> 32)), "s"(np) : "vcc");
        a[i] = (uint32_t)r;
      } else if (OP == 6) { // mixed: mul_hi for the quotient, one mad pair for the remainder
        const uint32_t qh = __umulhi(a[i], wp);
        uint64_t r;
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a[i]), "s"(w) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(qh), "s"(np) : "vcc");
        a[i] = (uint32_t)r;
      } else if (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "s"(w));
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) r ^= a[i];
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
  const double per_wave_cycles = 2.4e9 * (ms * 1e-3) / (ops / 64 / (256.0 * 4));
  printf("%-52s %8.3f ms  %.2f cycles per wave64 op (at 2.4 GHz)\n", name, ms, per_wave_cycles);
  return 0;
}
int main() {
  uint32_t *dout, *dsv; const int blocks = 256 * 8;
  CHECK(hipMalloc(&dout, blocks * 256 * 4)); CHECK(hipMalloc(&dsv, 16));
  uint32_t h[4] = {123456789u, 493827156u, 1073741441u, 0}; CHECK(hipMemcpy(dsv, h, 16, hipMemcpyHostToDevice));
  if (run<7>("v_add_u32", dout, dsv, blocks)) return 1;
  if (run<0>("v_mul_lo_u32", dout, dsv, blocks)) return 1;
  if (run<1>("v_mul_hi_u32", dout, dsv, blocks)) return 1;
  if (run<2>("v_mul_u32_u24", dout, dsv, blocks)) return 1;
  if (run<3>("v_mad_u32_u24", dout, dsv, blocks)) return 1;
  if (run<4>("Shoup multiply: mul_hi + 2 mul_lo + sub", dout, dsv, blocks)) return 1;
  if (run<5>("Shoup multiply: 3 v_mad_u64_u32", dout, dsv, blocks)) return 1;
  if (run<6>("Shoup multiply: mul_hi + 2 v_mad_u64_u32", dout, dsv, blocks)) return 1;
  return 0;
}
