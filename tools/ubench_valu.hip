// Instruction-throughput microbenchmark for the integer/FP64 VALU ops the
// modular arithmetic in labrador_snark_amd/csrc is built from (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
// Prints lane-ops/s for each op with every CU busy; used to choose between
// Shoup / Montgomery / mad_u64 formulations (numbers recorded in DESIGN.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

enum Op { MUL_LO, MUL_HI, MAD_U64, MUL_U24, MAD_U24, ADD_U32, MIN_U32, FMA_F64, FMA_F32,
          SHOUP_LAZY, MONT_STRICT, MAC64_MIN, ADD_U64, NOPS };
static const char* op_names[] = {"v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24",
  "v_mad_u32_u24", "v_add_u32", "v_min_u32", "v_fma_f64", "v_fma_f32", "shoup_lazy(3mul+sub)",
  "mont_strict(4mul+3)", "mac64+condsub(mad64+sub+min)", "add_u64"};

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, uint32_t p, uint32_t pinv) {
  uint32_t x[ILP];
  uint64_t a64[ILP];
  double d[ILP];
  float f[ILP];
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
#pragma unroll
  for (int i = 0; i < ILP; i++) {
    x[i] = seed * (t + 1) + i * 7919u;
    a64[i] = x[i];
    d[i] = (double)x[i];
    f[i] = (float)x[i];
  }
  uint32_t w = seed | 1u, wp = seed * 2654435761u;
  double dw = 1.0000001, dc = 0.5;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      if (OP == MUL_LO) x[i] = x[i] * w;
      else if (OP == MUL_HI) x[i] = __umulhi(x[i], w) + 1u;  // +1 folded? keep separate measure below
      else if (OP == MAD_U64) a64[i] = (uint64_t)(uint32_t)a64[i] * w + a64[i];
      else if (OP == MUL_U24) x[i] = __umul24(x[i], w);
      else if (OP == MAD_U24) x[i] = __umul24(x[i], w) + wp;
      else if (OP == ADD_U32) x[i] = x[i] + w;
      else if (OP == MIN_U32) x[i] = min(x[i], x[i] - p);
      else if (OP == FMA_F64) d[i] = __builtin_fma(d[i], dw, dc);
      else if (OP == FMA_F32) f[i] = __builtin_fmaf(f[i], 1.0000001f, 0.5f);
      else if (OP == SHOUP_LAZY) {
        uint32_t qh = __umulhi(x[i], wp);
        x[i] = x[i] * w - qh * p;
      } else if (OP == MONT_STRICT) {
        uint64_t T = (uint64_t)x[i] * w;
        uint32_t m = (uint32_t)T * pinv;
        uint32_t hi = (uint32_t)(T >> 32), mh = __umulhi(m, p);
        uint32_t r = hi - mh;
        x[i] = (hi < mh) ? r + p : r;
      } else if (OP == MAC64_MIN) {
        a64[i] = (uint64_t)x[i] * w + a64[i];
        uint32_t hi = (uint32_t)(a64[i] >> 32);
        hi = min(hi, hi - p);
        a64[i] = ((uint64_t)hi << 32) | (uint32_t)a64[i];
        x[i] += 3u;
      } else if (OP == ADD_U64) a64[i] = a64[i] + (((uint64_t)w << 32) | wp);
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) acc ^= x[i] ^ (uint32_t)a64[i] ^ (uint32_t)(a64[i] >> 32) ^ (uint32_t)d[i] ^ (uint32_t)f[i];
  out[t] = acc;
}

template <int OP>
static int run(uint32_t* dout, int blocks) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, dout, 12345u, 1073741441u, 0x3fffe80fu);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, dout, 12345u + rep, 1073741441u, 0x3fffe80fu);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double ops = (double)blocks * 256 * ITERS * ILP;
  printf("%-32s %8.3f ms  %10.3e lane-ops/s  (%.2f slots vs 7.86e13 full-rate)\n", op_names[OP], best,
         ops / (best * 1e-3), 7.86e13 / (ops / (best * 1e-3)));
  return 0;
}

int main() {
  int blocks = 256 * 8;
  uint32_t* dout;
  CHECK(hipMalloc(&dout, (size_t)blocks * 256 * 4));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d clock=%d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  if (run<MUL_LO>(dout, blocks)) return 1;
  if (run<MUL_HI>(dout, blocks)) return 1;
  if (run<MAD_U64>(dout, blocks)) return 1;
  if (run<MUL_U24>(dout, blocks)) return 1;
  if (run<MAD_U24>(dout, blocks)) return 1;
  if (run<ADD_U32>(dout, blocks)) return 1;
  if (run<MIN_U32>(dout, blocks)) return 1;
  if (run<FMA_F64>(dout, blocks)) return 1;
  if (run<FMA_F32>(dout, blocks)) return 1;
  if (run<SHOUP_LAZY>(dout, blocks)) return 1;
  if (run<MONT_STRICT>(dout, blocks)) return 1;
  if (run<MAC64_MIN>(dout, blocks)) return 1;
  if (run<ADD_U64>(dout, blocks)) return 1;
  CHECK(hipFree(dout));
  return 0;
}
