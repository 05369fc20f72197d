// Matrix-core ceiling of MI355X for the instruction the int8 GEMMs of this library are made of (v_mfma_i32_32x32x32_i8),
// measured the way VERDICT round 2 asked for: back-to-back independent MFMAs, operands RANDOM and register resident,
// one and two waves per SIMD, s_memtime (shader clock) / s_memrealtime (100 MHz constant clock) stamps around the loop of
// every wave — plus the same loop with an HBM stream feeding its B operands at the byte-per-MFMA ratio of the product
// kernel (k_gemm_i8_w3<6>: 20.55 GB per 5.66e7 wave-MFMAs = 363 B per MFMA: one 1-KiB wave load per ~3 MFMAs).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_i8.hip -o tools/ubench_mfma_i8
// Run:   tools/ubench_mfma_i8 > profiles/rNN_mfma_ceiling.json
// POP/s = int8 multiply-adds x 2 per second; 32768 multiply-adds per MFMA.  Datasheet dense int8 peak: 5 POP/s at 2.4 GHz
// (1024 SIMDs x 2048 operations per cycle).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e = (x);                                                                       \
    if (e != hipSuccess) {                                                                    \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);   \
      return 1;                                                                               \
    }                                                                                         \
  } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// hipcc's wait-count pass cannot follow loads that are consumed several loop trips after they were issued (it falls back to
// vmcnt(0..1) inside the loop — the very thing the product kernel's assembly K loop exists to avoid), so the stream variant issues
// its loads and counts its waits by hand: 16 loads in flight, the oldest one is complete at vmcnt(15)
#define GLOAD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(ptr) : "memory")
#define WAIT_OLDEST_OF_16(reg) asm volatile("s_waitcnt vmcnt(15)" : "+v"(reg) : : "memory")

constexpr int NACC = 3;   // independent accumulator tiles per wave (an MFMA result is not read again for 3 instructions)
constexpr int NOPS = 4;   // distinct A and B register operands cycled through

// MODE 0: operands stay in registers.  MODE 1: every third MFMA's B operand arrives from HBM (each wave streams its own
// region with 16-byte non-temporal loads, three loads in flight).  ZERO: all-zero operands (the "idle data" clock).
__global__ void k_fill(uint32_t* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = mix((uint32_t)i * 2654435761u + (uint32_t)(i >> 32));
}

// MODE = streamed B operands per 12-MFMA trip (0: none; 2: 171 B per MFMA, the product kernel's ratio — 20.55 GB per 1.13e8 MFMAs;
// 4: twice that)
// HOT: the streamed operands come from a 1-MiB region every wave shares (L2 hits, like the A tiles of the Ajtai product that 32
// jobs share) and every group's second MFMA takes its B operand from LDS (one ds_read_b128 per 3 MFMAs, the product kernels' ratio)
// MIX (with HOT): every other streamed operand comes from the wave's own HBM region, the others from the shared hot region — the
// traffic of a step-3 launch that multiplies 128 sets per pass over B: half of a wave's A operands are HBM misses (85 B per MFMA),
// the other half were just fetched by the wave that shares its row tile (L2 hits), and a B fragment from LDS feeds 3 MFMAs
template <int MODE, bool ZERO, bool HOT = false, bool MIX = false>
__global__ void __launch_bounds__(256) k_mfma(uint64_t* __restrict__ stamps, int* __restrict__ sink, const v4i* __restrict__ stream,
                                              size_t stream_words_per_wave, int iters, uint32_t seed) {
  __shared__ v4i lds[HOT ? 256 * 4 : 1];
  if (HOT) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) lds[i * 256 + threadIdx.x][j] = (int)mix(seed * 7u + threadIdx.x * 4099u + i * 16u + j);
    __syncthreads();
  }
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  v4i a[NOPS], b[NOPS];
#pragma unroll
  for (int i = 0; i < NOPS; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      a[i][j] = ZERO ? 0 : (int)mix(seed + (wave * 64 + lane) * 131u + i * 17u + j);
      b[i][j] = ZERO ? 0 : (int)mix(seed * 3u + (wave * 64 + lane) * 257u + i * 29u + j);
    }
  v16i acc[NACC];
#pragma unroll
  for (int n = 0; n < NACC; n++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc[n][j] = 0;
  const size_t cold_words = stream_words_per_wave;
  const v4i* cold = stream + (size_t)wave * cold_words + lane;   // MIX: this wave's own region
  size_t cpos = 0;
  if (HOT) stream_words_per_wave = 4096;   // 64 KiB per wave slot, 16 slots: 1 MiB shared by all waves
  const v4i* src = stream + (size_t)(HOT ? wave % 16 : wave) * stream_words_per_wave + lane;
  // MODE 1: 16 one-KiB wave loads in flight (the product kernel keeps 72 KB per CU = 18 KB per wave in flight): a load is used 48
  // MFMAs (~1 us) after it was issued
  v4i ring[16];
  size_t pos = 0;
  if (MODE != 0) {
#pragma unroll
    for (int i = 0; i < 16; i++) { GLOAD(ring[i], src + pos); pos += 64; }
  }
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  constexpr int LPT = MODE ? MODE : 4;     // streamed operands per 12-MFMA trip
  constexpr int TRIPS = 16 / LPT;          // trips unrolled so that one pass over the ring of 16 is one outer iteration
  for (int it = 0; it < iters; it += TRIPS) {
    // 12 MFMAs per trip: 4 groups of 3; a group's first MFMA takes a streamed B operand when it is that group's turn
#pragma unroll
    for (int tt = 0; tt < TRIPS; tt++) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        v4i bs = b[g % NOPS];
        if (MODE != 0 && g % (4 / LPT) == 0) {
          const int slot = tt * LPT + g / (4 / LPT);
          WAIT_OLDEST_OF_16(ring[slot]);
          bs = ring[slot];
          if (MIX && slot % 2 == 0) {
            GLOAD(ring[slot], cold + cpos);
            cpos += 64;
            if (cpos + 64 > cold_words) cpos = 4096 * 16 / 16;   // stay clear of the hot region at the start of the buffer
          } else {
            GLOAD(ring[slot], src + pos);
            pos += 64;
            if (pos + 64 > stream_words_per_wave) pos = 0;
          }
        }
        acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[g % NOPS], bs, acc[0], 0, 0, 0);
        v4i bl = b[(g + 1) % NOPS];
        if (HOT) bl = lds[((tt + g) % 4) * 256 + threadIdx.x];
        acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[(g + 1) % NOPS], bl, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[(g + 2) % NOPS], b[(g + 3) % NOPS], acc[2], 0, 0, 0);
      }
    }
    if (HOT) asm volatile("" ::: "memory");   // the LDS reads stay inside the loop
  }
  // The compiler does not know that the hand-issued loads still write the ring registers: they must have landed before those
  // registers are dead and handed to other values (a late load into a reused register — e.g. the one holding the address of the
  // stamps — is a wild store).  The operands keep all 16 alive up to this wait.
  if (MODE != 0)
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(ring[0]), "+v"(ring[1]), "+v"(ring[2]), "+v"(ring[3]), "+v"(ring[4]), "+v"(ring[5]), "+v"(ring[6]), "+v"(ring[7]),
                   "+v"(ring[8]), "+v"(ring[9]), "+v"(ring[10]), "+v"(ring[11]), "+v"(ring[12]), "+v"(ring[13]), "+v"(ring[14]), "+v"(ring[15])
                 :
                 : "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int x = 0;
#pragma unroll
  for (int n = 0; n < 3; n++)
#pragma unroll
    for (int j = 0; j < 16; j++) x ^= acc[n][j];
  if (x == 0x7fffffff) sink[wave] = x;   // keeps the accumulators alive
  if (lane == 0) { stamps[2 * wave] = t1 - t0; stamps[2 * wave + 1] = r1 - r0; }
}

struct Result { double ms, pops, shader_ghz, busy; };

template <int MODE, bool ZERO, bool HOT = false, bool MIX = false>
static int run(int blocks, int iters, uint64_t* dstamps, int* dsink, const v4i* dstream, size_t words_per_wave, Result* res) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int waves = blocks * 4;
  hipLaunchKernelGGL((k_mfma<MODE, ZERO, HOT, MIX>), dim3(blocks), dim3(256), 0, 0, dstamps, dsink, dstream, words_per_wave, iters / 8, 1u);   // warm-up
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_mfma<MODE, ZERO, HOT, MIX>), dim3(blocks), dim3(256), 0, 0, dstamps, dsink, dstream, words_per_wave, iters, 7u);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<uint64_t> st(2 * (size_t)waves);
  CHECK(hipMemcpy(st.data(), dstamps, st.size() * 8, hipMemcpyDeviceToHost));
  double ticks = 0, real = 0;
  for (int w = 0; w < waves; w++) { ticks += (double)st[2 * w]; real += (double)st[2 * w + 1]; }
  const double mfmas = 12.0 * iters;
  res->ms = ms;
  res->pops = (double)waves * mfmas * 32768.0 * 2.0 / (ms * 1e-3) / 1e15;
  res->shader_ghz = ticks / real * 0.1;                   // s_memrealtime ticks at 100 MHz
  // cycles a SIMD's matrix pipe is busy per MFMA: 32 for 32x32x32 i8 (8 passes x 4 cycles); waves per SIMD share the pipe
  // (meaningful at one wave per SIMD; with two, blocks that start late run alone for part of their loop)
  res->busy = mfmas * 32.0 * (waves / 1024.0 > 1.0 ? waves / 1024.0 : 1.0) / (ticks / waves);
  return 0;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400000;
  const size_t words_per_wave = (size_t)1 << 16;          // 1 MiB of 16-byte words per wave: 2 GiB for 2048 waves, far beyond L2 + MALL
  uint64_t* dstamps = nullptr;
  int* dsink = nullptr;
  v4i* dstream = nullptr;
  CHECK(hipMalloc(reinterpret_cast<void**>(&dstamps), 2 * 2048 * 8));
  CHECK(hipMalloc(reinterpret_cast<void**>(&dsink), 2048 * 4));
  CHECK(hipMalloc(reinterpret_cast<void**>(&dstream), 2048 * words_per_wave * 16));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(dstream), 2048 * words_per_wave * 4);   // random bytes
  CHECK(hipDeviceSynchronize());
  printf("{\n \"instruction\": \"v_mfma_i32_32x32x32_i8\", \"mfmas_per_wave\": %.0f, \"datasheet_peak_POPs\": 5.0,\n", 12.0 * iters);
  printf(" \"note\": \"POP/s = 2 x int8 multiply-adds per second over the whole launch (HIP events); shader clock = s_memtime / s_memrealtime "
         "ticks around the loop of every wave; pipe_busy = 32 cycles per MFMA x waves per SIMD / loop cycles\",\n \"runs\": [\n");
  struct Case { const char* name; int blocks; int mode; bool zero; } cases[] = {
      {"registers_random_1_wave_per_simd", 256, 0, false}, {"registers_random_2_waves_per_simd", 512, 0, false},
      {"registers_zero_1_wave_per_simd", 256, 0, true},
      {"hbm_stream_171B_per_mfma_1_wave_per_simd", 256, 2, false}, {"hbm_stream_171B_per_mfma_2_waves_per_simd", 512, 2, false},
      {"hbm_stream_341B_per_mfma_1_wave_per_simd", 256, 4, false},
      {"l2_hot_171B_per_mfma_plus_lds_operand_per_3_mfma_1_wave_per_simd", 256, 5, false},
      {"hbm_85B_plus_l2_hot_85B_per_mfma_plus_lds_operand_per_3_mfma_1_wave_per_simd", 256, 6, false}};
  for (size_t i = 0; i < sizeof(cases) / sizeof(cases[0]); i++) {
    Result r{};
    int rc;
    const Case& c = cases[i];
    if (c.mode == 0 && !c.zero) rc = run<0, false>(c.blocks, iters, dstamps, dsink, dstream, words_per_wave, &r);
    else if (c.mode == 0) rc = run<0, true>(c.blocks, iters, dstamps, dsink, dstream, words_per_wave, &r);
    else if (c.mode == 2) rc = run<2, false>(c.blocks, iters, dstamps, dsink, dstream, words_per_wave, &r);
    else if (c.mode == 5) rc = run<2, false, true>(c.blocks, iters, dstamps, dsink, dstream, words_per_wave, &r);
    else if (c.mode == 6) rc = run<2, false, true, true>(c.blocks, iters, dstamps, dsink, dstream, words_per_wave, &r);
    else rc = run<4, false>(c.blocks, iters, dstamps, dsink, dstream, words_per_wave, &r);
    if (rc) return rc;
    const double stream_gbs = c.mode == 5 ? 0.0 : (double)c.blocks * 4 * ((double)(c.mode == 6 ? 1 : c.mode) * iters) * 1024.0 / (r.ms * 1e-3) / 1e9;
    printf("  {\"case\": \"%s\", \"waves\": %d, \"ms\": %.3f, \"POPs\": %.3f, \"frac_of_datasheet\": %.3f, \"in_loop_shader_clock_GHz\": %.3f, "
           "\"matrix_pipe_busy\": %.3f, \"hbm_stream_GBps\": %.0f}%s\n",
           c.name, c.blocks * 4, r.ms, r.pops, r.pops / 5.0, r.shader_ghz, r.busy, stream_gbs, i + 1 < sizeof(cases) / sizeof(cases[0]) ? "," : "");
  }
  printf(" ]\n}\n");
  return 0;
}
