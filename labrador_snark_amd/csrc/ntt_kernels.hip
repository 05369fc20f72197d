// ntt_kernels.hip — batched 64-point negacyclic NTT/INTT and the conversions
// between the boundary's coefficient form (u64 mod q) and the device's hat form
// (u32 residues per RNS limb, NTT domain).  gfx950 only.
//
// Replaces the transform inside `PLAN.negacyclic_polymul` and the u64<->Zq
// marshalling around it (src/algebraic.rs:385-399, 407-420).
#include "kernels.h"

#include <mutex>
#include <vector>

namespace lab {
typedef unsigned __int128 u128;

// Host-side record of which device tables belong to a single-limb strict-32 context (LimbTable.wide): the launchers take
// device pointers and pick the kernel instantiation from it.
static std::mutex g_wide_mu;
static std::vector<const LimbTable*> g_wide_tabs;
void limb_tables_register(const LimbTable* dev_tabs, bool wide) {
  std::lock_guard<std::mutex> g(g_wide_mu);
  for (size_t i = 0; i < g_wide_tabs.size(); i++)
    if (g_wide_tabs[i] == dev_tabs) { g_wide_tabs.erase(g_wide_tabs.begin() + i); break; }
  if (wide) g_wide_tabs.push_back(dev_tabs);
}
bool limb_tables_wide(const LimbTable* dev_tabs) {
  std::lock_guard<std::mutex> g(g_wide_mu);
  for (const LimbTable* t : g_wide_tabs)
    if (t == dev_tabs) return true;
  return false;
}

// One polynomial per thread, one wave per block: the wave's 64 polynomials
// (16 KiB) are fetched with 16-byte coalesced loads, transposed through LDS,
// transformed entirely in registers (192 lazy butterflies, twiddles from
// scalar loads) and written back the same way.
// Algorithmic traffic: 256 B in + 256 B out per polynomial per limb.
template <bool FWD, bool WIDE>
__global__ void __launch_bounds__(64)
k_ntt64(const uint32_t* __restrict__ src, size_t src_ps, uint32_t* __restrict__ dst, size_t dst_ps, size_t count,
        const LimbTable* __restrict__ tabs, int escale, size_t src_js, size_t dst_js) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[TILE_WORDS];
  const int limb = blockIdx.y;
  const LimbTable* tab = tabs + limb;
  const size_t base = (size_t)blockIdx.x * 64;
  const int lane = threadIdx.x;
  uint32_t x[D];
  tile_load_u32(x, src + (size_t)limb * src_ps + (size_t)blockIdx.z * src_js, base, count, lds, lane);
  if (FWD) ntt64_fwd<WIDE>(x, tab); else ntt64_inv<WIDE>(x, tab, escale);
  tile_store_u32(x, dst + (size_t)limb * dst_ps + (size_t)blockIdx.z * dst_js, base, count, lds, lane);
}

// Inverse transform straight from the slot-major staging array a multi-job GEMM leaves behind (k_gemm_i8:
// stage[gemm job][slot j][row][set s], mfma_kernels.hip): thread (row, s) reads its 64 slots with 64 coalesced 4-byte
// loads (a wave = 64 neighbouring (row, s): 256 B per slot), transforms, and the wave writes whole 256-byte
// polynomials into the dense [count][64] plane the CRT kernel reads — polynomial (s % nsub) * set_stride + row of
// source job gemm_job * (ns / nsub) + s / nsub.  Replaces k_slot_transpose + the tile load of k_ntt64 for results
// that are only wanted as coefficients (t = A s).
__global__ void __launch_bounds__(64)
k_intt_from_stage(const uint32_t* __restrict__ stage, unsigned rows, unsigned rows_pad, unsigned ns, unsigned ns_valid, unsigned nsub,
                  uint32_t* __restrict__ dst, size_t dst_js, size_t set_stride, const LimbTable* __restrict__ tab) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[TILE_WORDS];
  const int lane = threadIdx.x;
  const size_t per_slot = (size_t)rows_pad * ns;
  const size_t f0 = (size_t)blockIdx.x * 64;
  const uint32_t* src = stage + (size_t)blockIdx.y * D * per_slot + f0 + lane;
  uint32_t x[D];
#pragma unroll
  for (int j = 0; j < D; j++) x[j] = __builtin_nontemporal_load(src + (size_t)j * per_slot);
  ntt64_inverse(x, tab, 0);
#pragma unroll
  for (int c = 0; c < 16; c++)
    *reinterpret_cast<uint4*>(lds + lane * TILE_PITCH + c * 4) = make_uint4(x[c * 4 + 0], x[c * 4 + 1], x[c * 4 + 2], x[c * 4 + 3]);
  __syncthreads();   // block == one wave
  const unsigned sub = ns / nsub;
#pragma unroll 4
  for (int i = 0; i < 16; i++) {
    const int p = i * 4 + (lane >> 4);
    const size_t f = f0 + p;
    const unsigned row = (unsigned)(f / ns), s = (unsigned)(f % ns), sl = s % nsub;
    if (row < rows && sl < ns_valid) {
      const uint4 o = *reinterpret_cast<const uint4*>(lds + p * TILE_PITCH + (lane & 15) * 4);
      uint32_t* d = dst + ((size_t)blockIdx.y * sub + s / nsub) * dst_js + ((size_t)sl * set_stride + row) * D + (lane & 15) * 4;
      *reinterpret_cast<uint4*>(d) = o;
    }
  }
}

hipError_t launch_intt_from_stage(const uint32_t* stage, size_t rows, size_t rows_pad, size_t ns, size_t ns_valid, size_t nsub,
                                  size_t gemm_jobs, uint32_t* dst, size_t dst_js, size_t set_stride, const LimbTable* tab, hipStream_t s) {
  if (rows == 0 || gemm_jobs == 0) return hipSuccess;
  if ((rows_pad * ns) % 64 || ns % nsub) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_intt_from_stage, dim3((unsigned)(rows_pad * ns / 64), (unsigned)gemm_jobs), dim3(64), 0, s, stage, (unsigned)rows,
                     (unsigned)rows_pad, (unsigned)ns, (unsigned)ns_valid, (unsigned)nsub, dst, dst_js, set_stride, tab);
  return hipGetLastError();
}

hipError_t launch_ntt(bool forward, ConstHatView src, HatView dst, size_t count, int nlimbs, const LimbTable* tabs,
                      int escale, hipStream_t s, size_t njobs, size_t src_js, size_t dst_js) {
  if (count == 0 || njobs == 0) return hipSuccess;
  dim3 grid((unsigned)((count + 63) / 64), (unsigned)nlimbs, (unsigned)njobs);
  const bool wide = limb_tables_wide(tabs);
#define LAB_NTT(F, W) hipLaunchKernelGGL((k_ntt64<F, W>), grid, dim3(64), 0, s, src.base, src.plane_stride, dst.base, dst.plane_stride, \
                                         count, tabs, escale, src_js, dst_js)
  if (forward) { if (wide) LAB_NTT(true, true); else LAB_NTT(true, false); }
  else { if (wide) LAB_NTT(false, true); else LAB_NTT(false, false); }
#undef LAB_NTT
  return hipGetLastError();
}

// coefficient u64 in [0,q) -> residue mod each limb prime (q < 2^40, so x = hi*2^32 + lo with hi < 2^8).
// Division-free: hi*2^32 mod p by a Shoup multiplication with the tabulated 2^32 mod p; lo mod p by conditional
// subtractions when p > 2^29 (lo < 2^32 < 8p), by mod_u64 for the small limb of the digit basis.
template <bool WIDE = false>
__device__ __forceinline__ uint32_t reduce_u40(uint64_t v, const LimbTable& tb) {
  const uint32_t p = tb.p, lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  if constexpr (WIDE) return (uint32_t)mod_u64_near(v, p, tb.inv_p);   // p >= 2^30: quotient < 2^10
  if (p > (1u << 29)) {
    uint32_t b = lo;
    b = min(b, b - 4 * p);   // wraps (and is discarded by min) unless b >= 4p; 4p < 2^32 as p < 2^30
    b = min(b, b - 2 * p);
    b = min(b, b - p);
    if (hi == 0) return b;
    const uint32_t a = mul_shoup_lazy(hi, tb.q_hi_mul, tb.q_hi_mul_sh, p);   // [0, 2p)
    return csub(csub(a + b, 2 * p), p);                                      // a + b < 3p < 2^32
  }
  return (uint32_t)mod_u64_near(v, p, tb.inv_p);
}

__global__ void __launch_bounds__(256)
k_reduce_u64(const JobPtrs src, uint32_t* __restrict__ dst, size_t dst_ps, size_t nwords,
             int nlimbs, const LimbTable* __restrict__ tabs, size_t dst_js) {
  const uint64_t* __restrict__ coeffs = static_cast<const uint64_t*>(src.p[blockIdx.y]);
  dst += (size_t)blockIdx.y * dst_js;
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 2;
  for (; i < nwords; i += stride) {
    // nwords is a multiple of 64, so the pair (i, i+1) is always in range
    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(coeffs + i);
    for (int l = 0; l < nlimbs; l++)
      *reinterpret_cast<uint2*>(dst + (size_t)l * dst_ps + i) =
          tabs[l].wide ? make_uint2(reduce_u40<true>(v.x, tabs[l]), reduce_u40<true>(v.y, tabs[l]))
                       : make_uint2(reduce_u40(v.x, tabs[l]), reduce_u40(v.y, tabs[l]));
  }
}

// ---- to_hat in one pass: coefficients (u64 mod q) -> residues -> forward NTT, for every limb, without the
// intermediate residue planes (k_reduce_u64 + k_ntt64 move 44 bytes per coefficient, this kernel 8 + 4 per limb).
// One wave per block, one polynomial per thread: the wave's 64 polynomials (32 KiB of u64) come in two halves of 32
// coefficients through the LDS tile (lane l, iteration i fetches the 16-byte piece i*64 + l of the half: polynomial
// (i*64 + l) / 16, piece % 16 — 256 B contiguous per polynomial), are read back transposed (row pitch 68 words:
// conflict-free) and stay in registers while each limb is reduced, transformed and stored.
// NORM: the pass also adds the squared Euclidean norm of the job's coefficients (raw residues, like poly_norm,
// src/util.rs:195-202) to norm_sum[job] — saturating: a polynomial with a coefficient above norm_vmax or a partial sum
// above norm_limit counts as norm_limit + 1, so the total exceeds norm_limit exactly when the true norm does and never
// overflows ((blocks + 1) * (norm_limit + 1) < 2^64 is the caller's business).
template <bool NORM, bool SRC16, bool WIDE = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_to_hat_fused(const JobPtrs src, uint32_t* __restrict__ dst, size_t dst_ps, size_t dst_js, size_t count, int nlimbs,
               const LimbTable* __restrict__ tabs, uint64_t in_bound, unsigned long long* __restrict__ norm_sum,
               uint64_t norm_limit, uint32_t norm_vmax, unsigned* __restrict__ above16 = nullptr) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[TILE_WORDS];
  const uint64_t* __restrict__ coeffs = static_cast<const uint64_t*>(src.p[blockIdx.y]);
  dst += (size_t)blockIdx.y * dst_js;
  const size_t base = (size_t)blockIdx.x * 64;
  const int lane = threadIdx.x;
  // q < 2^40: a coefficient is a 32-bit low word and an 8-bit high part (packed four to a register)
  uint32_t clo[D], chi[D / 4];
#pragma unroll
  for (int k = 0; k < D / 4; k++) chi[k] = 0;
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  if constexpr (SRC16) {
    // 16-bit coefficients (decomposition digits): the wave's 64 polynomials are 8 KiB contiguous, 8 pieces of 16 bytes
    // (8 coefficients) per polynomial; LDS row pitch 36 words
    const uint16_t* __restrict__ c16 = static_cast<const uint16_t*>(src.p[blockIdx.y]);
    u32x4_t v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int idx = i * 64 + lane, poly = idx >> 3, piece = idx & 7;
      const u32x4_t z = {0, 0, 0, 0};
      v[i] = base + poly < count ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(c16 + (base + poly) * D) + piece) : z;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int idx = i * 64 + lane, poly = idx >> 3, piece = idx & 7;
      *reinterpret_cast<u32x4_t*>(lds + poly * 36 + piece * 4) = v[i];
    }
    __syncthreads();   // block == one wave
#pragma unroll
    for (int piece = 0; piece < 8; piece++) {
      const u32x4_t t = *reinterpret_cast<const u32x4_t*>(lds + lane * 36 + piece * 4);
      const int k = piece * 8;
      clo[k] = t.x & 0xFFFFu; clo[k + 1] = t.x >> 16; clo[k + 2] = t.y & 0xFFFFu; clo[k + 3] = t.y >> 16;
      clo[k + 4] = t.z & 0xFFFFu; clo[k + 5] = t.z >> 16; clo[k + 6] = t.w & 0xFFFFu; clo[k + 7] = t.w >> 16;
    }
    __syncthreads();
  } else {
#pragma unroll
  for (int half = 0; half < 2; half++) {
    u32x4_t v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int idx = i * 64 + lane, poly = idx >> 4, piece = idx & 15;
      const u32x4_t z = {0, 0, 0, 0};
      v[i] = base + poly < count
                 ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(coeffs + (base + poly) * D + half * 32) + piece)
                 : z;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int idx = i * 64 + lane, poly = idx >> 4, piece = idx & 15;
      *reinterpret_cast<u32x4_t*>(lds + poly * TILE_PITCH + piece * 4) = v[i];
    }
    __syncthreads();   // block == one wave
#pragma unroll
    for (int piece = 0; piece < 16; piece++) {
      const u32x4_t t = *reinterpret_cast<const u32x4_t*>(lds + lane * TILE_PITCH + piece * 4);
      const int k = half * 32 + piece * 2;
      clo[k] = t.x; clo[k + 1] = t.z;
      chi[k / 4] |= ((t.y & 0xFFu) << (8 * (k & 3))) | ((t.w & 0xFFu) << (8 * ((k + 1) & 3)));
    }
    __syncthreads();
  }
  }
  if constexpr (NORM) {
    uint64_t ss = 0;
    bool over = false;
#pragma unroll
    for (int k = 0; k < D; k++) {
      over = over || clo[k] > norm_vmax;
      ss += (uint64_t)clo[k] * clo[k];               // clo <= norm_vmax < 2^24 when it counts: 64 terms < 2^54
    }
#pragma unroll
    for (int k = 0; k < D / 4; k++) over = over || chi[k] != 0;
    if (above16) {   // does any coefficient of this job need more than 16 bits?  (the pass over Pi has a cheaper form when none does)
      bool big = false;
#pragma unroll
      for (int k = 0; k < D; k++) big = big || clo[k] > 0xFFFFu;
#pragma unroll
      for (int k = 0; k < D / 4; k++) big = big || chi[k] != 0;
      if (__builtin_amdgcn_ballot_w64(big) != 0 && lane == 0) atomicOr(&above16[blockIdx.y], 1u);
    }
    if (over || ss > norm_limit) ss = norm_limit + 1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off);   // <= 64 * (2^48 + 1)
    if (lane == 0) atomicAdd(&norm_sum[blockIdx.y], ss > norm_limit ? norm_limit + 1 : ss);
  }
#pragma unroll 1
  for (int l = 0; l < nlimbs; l++) {
    const LimbTable* tab = tabs + l;
    uint32_t x[D];
    if constexpr (WIDE) {                // single-limb strict-32 path: q itself, 2^30 <= q < 2^32
      if (in_bound <= tab->p) {
#pragma unroll
        for (int k = 0; k < D; k++) x[k] = clo[k];
      } else {
        const uint32_t p = tab->p;
        const double ip = tab->inv_p;
#pragma unroll
        for (int k = 0; k < D; k++)
          x[k] = (uint32_t)mod_u64_near(((uint64_t)((chi[k / 4] >> (8 * (k & 3))) & 0xFFu) << 32) | clo[k], p, ip);
      }
    } else
    if (in_bound <= tab->p) {            // the caller vouches for coefficients below in_bound (decomposition digits): nothing to reduce
#pragma unroll
      for (int k = 0; k < D; k++) x[k] = clo[k];
    } else if (in_bound <= 4ull * tab->p) {   // ... or only a little above p (a witness within its norm bound)
      const uint32_t p = tab->p;
#pragma unroll
      for (int k = 0; k < D; k++) x[k] = csub(csub(clo[k], 2 * p), p);
    } else if (tab->p > (1u << 29)) {
#pragma unroll
      for (int k = 0; k < D; k++) x[k] = reduce_u40(((uint64_t)((chi[k / 4] >> (8 * (k & 3))) & 0xFFu) << 32) | clo[k], *tab);
    } else {   // a digit-basis limb below 2^23 (the u16 limb, the matrix-core basis), or q itself as the only prime (v < p)
      const uint32_t p = tab->p, m = (uint32_t)(4294967296.0 * tab->inv_p) - 1, r1 = tab->r1;
#pragma unroll
      for (int k = 0; k < D; k++) x[k] = reduce_small16((chi[k / 4] >> (8 * (k & 3))) & 0xFFu, clo[k], p, m, r1);
    }
    ntt64_fwd<WIDE>(x, tab);
    tile_store_u32(x, dst + (size_t)l * dst_ps, base, count, lds, lane);
  }
}

hipError_t launch_to_hat_fused(const JobPtrs& coeffs, HatView dst, size_t dst_js, size_t count, int nlimbs, const LimbTable* tabs,
                               size_t njobs, hipStream_t s, uint64_t in_bound, unsigned long long* norm_sum, uint64_t norm_limit,
                               bool src_u16, unsigned* above16) {
  if (count == 0 || njobs == 0) return hipSuccess;
  const dim3 g((unsigned)((count + 63) / 64), (unsigned)njobs);
  if (norm_sum) {
    // exactness of the saturating sum: coefficients that count are <= floor(sqrt(norm_limit)) < 2^24
    if (norm_limit >= (1ull << 48) || (u128)(g.x + 1) * (norm_limit + 1) >= ((u128)1 << 64)) return hipErrorInvalidValue;
    uint32_t vmax = (uint32_t)sqrt((double)norm_limit);
    while ((uint64_t)vmax * vmax > norm_limit) vmax--;
    while ((uint64_t)(vmax + 1) * (vmax + 1) <= norm_limit) vmax++;
    if (src_u16) return hipErrorInvalidValue;
    if (limb_tables_wide(tabs))
      hipLaunchKernelGGL((k_to_hat_fused<true, false, true>), g, dim3(64), 0, s, coeffs, dst.base, dst.plane_stride, dst_js, count, nlimbs, tabs,
                         in_bound, norm_sum, norm_limit, vmax, above16);
    else
    hipLaunchKernelGGL((k_to_hat_fused<true, false>), g, dim3(64), 0, s, coeffs, dst.base, dst.plane_stride, dst_js, count, nlimbs, tabs,
                       in_bound, norm_sum, norm_limit, vmax, above16);
  } else if (limb_tables_wide(tabs)) {
    if (src_u16)
      hipLaunchKernelGGL((k_to_hat_fused<false, true, true>), g, dim3(64), 0, s, coeffs, dst.base, dst.plane_stride, dst_js, count, nlimbs, tabs,
                         in_bound, nullptr, (uint64_t)0, (uint32_t)0);
    else
      hipLaunchKernelGGL((k_to_hat_fused<false, false, true>), g, dim3(64), 0, s, coeffs, dst.base, dst.plane_stride, dst_js, count, nlimbs,
                         tabs, in_bound, nullptr, (uint64_t)0, (uint32_t)0);
  } else if (src_u16) {
    hipLaunchKernelGGL((k_to_hat_fused<false, true>), g, dim3(64), 0, s, coeffs, dst.base, dst.plane_stride, dst_js, count, nlimbs, tabs,
                       in_bound, nullptr, (uint64_t)0, (uint32_t)0);
  } else {
    hipLaunchKernelGGL((k_to_hat_fused<false, false>), g, dim3(64), 0, s, coeffs, dst.base, dst.plane_stride, dst_js, count, nlimbs, tabs,
                       in_bound, nullptr, (uint64_t)0, (uint32_t)0);
  }
  return hipGetLastError();
}

hipError_t launch_reduce_u64_jobs(const JobPtrs& coeffs, HatView dst, size_t dst_js, size_t count, int nlimbs,
                                  const LimbTable* tabs, size_t njobs, hipStream_t s) {
  if (count == 0 || njobs == 0) return hipSuccess;
  size_t nwords = count * D;
  unsigned blocks = (unsigned)((nwords / 2 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_reduce_u64, dim3(blocks, (unsigned)njobs), dim3(256), 0, s, coeffs, dst.base, dst.plane_stride, nwords,
                     nlimbs, tabs, dst_js);
  return hipGetLastError();
}

hipError_t launch_reduce_u64(const uint64_t* coeffs, HatView dst, size_t count, int nlimbs, const LimbTable* tabs,
                             hipStream_t s) {
  JobPtrs t;
  t.p[0] = const_cast<uint64_t*>(coeffs);
  return launch_reduce_u64_jobs(t, dst, 0, count, nlimbs, tabs, 1, s);
}

__device__ __forceinline__ uint64_t mulmod_q(uint32_t d, uint64_t c, uint64_t q, double inv_q) {
  // d < 2^30, c < q < 2^40: the product fits 64 bits when q < 2^34, else c is split at bit 20 (quotients < 2^50)
  if (q < (1ull << 34)) return mod_u64_near((uint64_t)d * c, q, inv_q);
  const uint64_t chi = c >> 20, clo = c & 0xFFFFF;
  uint64_t t = mod_u64_near((uint64_t)d * chi, q, inv_q);
  t = mod_u64_near(t << 20, q, inv_q);
  return mod_u64_near(t + (uint64_t)d * clo, q, inv_q);
}

// hat residues (already inverse-transformed, canonical) -> signed CRT -> mod q.
// The integer being reconstructed is the exact negacyclic result; it lies in
// (-M/2, M/2), which lab_ctx guarantees through max_len.  No integer divisions: see mod_u64.
// L (limbs) is a template parameter so that the Garner recurrence is straight-line code with the table in scalar
// registers; a thread converts two neighbouring coefficients (8-byte loads per limb, one 16-byte store).
template <int L>
__device__ __forceinline__ uint64_t crt_one(const uint32_t (&r)[L], const CrtTable* __restrict__ crt, uint64_t q, double inv_q) {
  if (L == 1 && (uint64_t)crt->p[0] == q) return r[0];
  uint32_t d[L];
  // Garner mixed-radix digits
  d[0] = r[0];
#pragma unroll
  for (int a = 1; a < L; a++) {
    const uint32_t p = crt->p[a];
    uint32_t t = r[a];
#pragma unroll
    for (int b = 0; b < a; b++) {
      // a digit of limb b is below p_b: one conditional subtraction when p_b < 2p (primes of one size), else a full reduction
      const uint32_t db = crt->p[b] < 2 * p ? csub(d[b], p) : (uint32_t)mod_u64_near(d[b], p, crt->inv_p[a]);
      const uint32_t diff = t >= db ? t - db : t + p - db;
      t = csub(mul_shoup_lazy(diff, crt->ginv[a][b][0], crt->ginv[a][b][1], p), p);
    }
    d[a] = t;
  }
  // sign: value > (M-1)/2 ?  (lexicographic comparison of the digits, most significant first)
  bool neg = false, decided = false;
#pragma unroll
  for (int a = L - 1; a >= 0; a--) {
    const uint32_t hd = crt->half_digits[a];
    if (!decided && d[a] != hd) { neg = d[a] > hd; decided = true; }
  }
  uint64_t acc;
  if (crt->lazy_sum) {   // sum_a (p_a - 1)(q - 1) < 2^63 (e.g. 23-bit limbs, q below 2^34): the mixed-radix sum first, one reduction
    acc = d[0];
#pragma unroll
    for (int a = 1; a < L; a++) acc += (uint64_t)d[a] * crt->prefix_mod_q[a];
    acc = mod_u64_near(acc, q, inv_q);                     // quotient below sum_a p_a < 2^32
  } else {
    acc = d[0] < q ? d[0] : mod_u64_near(d[0], q, inv_q);
#pragma unroll
    for (int a = 1; a < L; a++) {
      acc += mulmod_q(d[a], crt->prefix_mod_q[a], q, inv_q);
      acc -= acc >= q ? q : 0;
    }
  }
  const uint64_t mm = crt->m_mod_q;
  if (neg) acc = acc >= mm ? acc - mm : acc + q - mm;
  return acc;
}

// crt_one<3> for limbs of one size (CrtTable.fast3), straight-line: the differences of the Garner recurrence are kept positive by
// adding 2p (every digit of an earlier limb is below 2p of a later one, and Shoup's product takes any 32-bit operand), the
// mixed-radix sum is below 2^63 with a quotient by q below 2^32 (one double multiply, off by one at most in either direction).
struct Crt3K {
  uint32_t p1, p2, g10, g10s, g20, g20s, g21, g21s, h0, h1, h2;
  uint64_t P1, P2, q, mm;
  double inv_q;
};
__device__ __forceinline__ Crt3K crt3_load(const CrtTable* __restrict__ crt) {
  Crt3K k;
  k.p1 = crt->p[1]; k.p2 = crt->p[2];
  k.g10 = crt->ginv[1][0][0]; k.g10s = crt->ginv[1][0][1];
  k.g20 = crt->ginv[2][0][0]; k.g20s = crt->ginv[2][0][1];
  k.g21 = crt->ginv[2][1][0]; k.g21s = crt->ginv[2][1][1];
  k.h0 = crt->half_digits[0]; k.h1 = crt->half_digits[1]; k.h2 = crt->half_digits[2];
  k.P1 = crt->prefix_mod_q[1]; k.P2 = crt->prefix_mod_q[2]; k.q = crt->q; k.mm = crt->m_mod_q; k.inv_q = crt->inv_q;
  return k;
}
__device__ __forceinline__ uint64_t crt3_fast(uint32_t r0, uint32_t r1, uint32_t r2, const Crt3K& k) {
  const uint32_t d0 = r0;
  const uint32_t d1 = csub(mul_shoup_lazy(r1 + 2 * k.p1 - d0, k.g10, k.g10s, k.p1), k.p1);
  const uint32_t t = mul_shoup_lazy(r2 + 2 * k.p2 - d0, k.g20, k.g20s, k.p2);              // [0, 2 p2)
  const uint32_t d2 = csub(mul_shoup_lazy(t + 2 * k.p2 - d1, k.g21, k.g21s, k.p2), k.p2);
  const bool neg = d2 > k.h2 || (d2 == k.h2 && (d1 > k.h1 || (d1 == k.h1 && d0 > k.h0)));
  const uint64_t acc = (uint64_t)d0 + (uint64_t)d1 * k.P1 + (uint64_t)d2 * k.P2;
  const uint32_t qh = (uint32_t)((double)acc * k.inv_q);
  int64_t r = (int64_t)(acc - (uint64_t)qh * k.q);
  r += r < 0 ? (int64_t)k.q : 0;
  r -= r >= (int64_t)k.q ? (int64_t)k.q : 0;
  uint64_t v = (uint64_t)r;
  if (neg) v = v >= k.mm ? v - k.mm : v + k.q - k.mm;
  return v;
}
// the digits of four values (decompose_polynomial_vec, src/util.rs:360-442, as in k_decompose_u16x4) to o + k * kstride, k < exp.
// fdiv: every quotient fits 32 bits ((q >> 32) < base).
__device__ __forceinline__ void digits_of_quad(uint64_t (&v)[4], uint16_t* __restrict__ o, size_t kstride, int exp, uint32_t b32, uint32_t bm,
                                               uint64_t half, bool fdiv) {
  if (fdiv) {
    const double inv_b = 1.0 / (double)b32;
    const uint32_t h32 = (uint32_t)half;
    uint32_t w[4], d[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint32_t r0 = divmod_by_double(v[c], b32, inv_b, &w[c]);
      d[c] = r0 > h32 ? b32 - r0 : r0;
    }
    *reinterpret_cast<uint2*>(o) = make_uint2(d[0] | (d[1] << 16), d[2] | (d[3] << 16));
    for (int k = 1; k < exp; k++) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const uint32_t r0 = divmod_by_double(w[c], b32, inv_b, &w[c]);
        d[c] = r0 > h32 ? b32 - r0 : r0;
      }
      *reinterpret_cast<uint2*>(o + (size_t)k * kstride) = make_uint2(d[0] | (d[1] << 16), d[2] | (d[3] << 16));
    }
    return;
  }
  for (int k = 0; k < exp; k++) {
    uint32_t d[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      // v = qq*b + r0.  The reference continues with (v - r) / b, r = r0 or b - r0: that is qq in both cases (r0 > b/2 gives
      // (qq - 1)*b + 2*r0 with b < 2*r0 < 2b), and v - r never wraps mod q here (base <= q): one division per digit
      uint64_t qq;
      const uint32_t r0 = divmod_u40_u16((uint32_t)(v[c] >> 32), (uint32_t)v[c], b32, bm, &qq);
      d[c] = r0 > half ? b32 - r0 : r0;
      v[c] = qq;
    }
    *reinterpret_cast<uint2*>(o + (size_t)k * kstride) = make_uint2(d[0] | (d[1] << 16), d[2] | (d[3] << 16));
  }
}

// Four coefficients per thread (16-byte loads per limb, 32 bytes out).  DIG: the value is also decomposed on the spot
// (decompose_polynomial_vec, src/util.rs:360-442, in the library-internal form of k_decompose_u16x4: 16-bit digits, base below
// 2^16 and not above q) — the digits of t = A s leave with t itself, its coefficients are not read back.
template <int L, bool DIG>
__global__ void __launch_bounds__(256)
k_crt_to_u64(const uint32_t* __restrict__ src, size_t src_ps, const JobPtrs dstp, size_t nquads,
             const CrtTable* __restrict__ crt, size_t src_js, const CrtDigits dg) {
  uint64_t* __restrict__ coeffs = static_cast<uint64_t*>(dstp.p[blockIdx.y]);
  uint16_t* __restrict__ digits = DIG ? static_cast<uint16_t*>(dg.out16.p[blockIdx.y]) : nullptr;
  src += (size_t)blockIdx.y * src_js;
  const uint64_t q = crt->q;
  const double inv_q = crt->inv_q;
  const uint32_t b32 = dg.base, bm = DIG ? (uint32_t)(0x100000000ull / dg.base) : 0;
  const uint64_t half = DIG ? (dg.base / 2) % q : 0;
  const bool fast3 = L == 3 && crt->fast3 != 0, fdiv = DIG && (q >> 32) < dg.base;   // (uniform)
  Crt3K k3{};
  if (L == 3 && fast3) k3 = crt3_load(crt);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nquads; i += stride) {
    uint32_t r[4][L];
#pragma unroll
    for (int l = 0; l < L; l++) {
      typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
      const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(src + (size_t)l * src_ps) + i);
      r[0][l] = v.x; r[1][l] = v.y; r[2][l] = v.z; r[3][l] = v.w;
    }
    uint64_t v[4];
    if constexpr (L == 3) {
      if (fast3) {
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = crt3_fast(r[c][0], r[c][1], r[c][2], k3);
      } else {
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = crt_one<L>(r[c], crt, q, inv_q);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++) v[c] = crt_one<L>(r[c], crt, q, inv_q);
    }
    ulonglong2 o0, o1;
    o0.x = v[0]; o0.y = v[1]; o1.x = v[2]; o1.y = v[3];
    reinterpret_cast<ulonglong2*>(coeffs)[2 * i] = o0;
    reinterpret_cast<ulonglong2*>(coeffs)[2 * i + 1] = o1;
    if constexpr (DIG) {
      const size_t w0 = i * 4;
      const uint32_t e = (uint32_t)(w0 / D), g = (uint32_t)dg.group, eg = e / g;
      uint16_t* const o = digits + ((size_t)eg * dg.exp * g + (e - eg * g)) * D + w0 % D;
      digits_of_quad(v, o, (size_t)g * D, dg.exp, b32, bm, half, fdiv);
    }
  }
}

// k_intt_from_stage + k_crt_to_u64 in one pass, for results a multi-job GEMM left in slot-major staging arrays of EVERY limb
// (stage + l * stage_ps: [gemm job][slot j][row][set s]) that are only wanted as coefficients mod q (t = A s, src/proofgen.rs:41-49):
// the dense inverse-transform output ([L][count][64] words written and read back, as much traffic as the staging arrays
// themselves) never exists.  Block = 4 waves = 64 neighbouring (row, s).  Wave l < L: thread (row, s) reads its 64 slots of limb l
// (coalesced: 256 B per slot and wave), inverse-transforms them in registers and parks the polynomial in LDS ([limb][polynomial]
// [64 + 4]); then thread (polynomial, quad) of all four waves reconstructs four neighbouring coefficients (crt_one) and the block
// stores whole polynomials: 512 B of t, 128 B of every digit.  (All limbs in the registers of one wave — 192 of them — is slower:
// it spills at two waves per SIMD and waits on its loads at one.)  DIG as in k_crt_to_u64.
// Polynomial (s % nsub) * set_stride + row of proof job * (ns / nsub) + s / nsub.
template <int L, bool DIG>
__global__ void __launch_bounds__(256)
k_stage_to_coeffs(const uint32_t* __restrict__ stage, size_t stage_ps, unsigned rows, unsigned rows_pad, unsigned ns, unsigned ns_valid,
                  unsigned nsub, const JobPtrs dstp, size_t set_stride, const LimbTable* __restrict__ tabs,
                  const CrtTable* __restrict__ crt, const CrtDigits dg) {
  static_assert(L >= 2 && L <= 3, "a wave per limb, at most 3 x 17 KB of LDS");
  __shared__ __attribute__((aligned(16))) uint32_t lds[L * TILE_WORDS];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const size_t per_slot = (size_t)rows_pad * ns;
  const size_t f0 = (size_t)blockIdx.x * 64;
  if (w < L) {
    const uint32_t* src = stage + (size_t)w * stage_ps + (size_t)blockIdx.y * D * per_slot + f0 + lane;
    uint32_t x[D];
#pragma unroll
    for (int j = 0; j < D; j++) x[j] = __builtin_nontemporal_load(src + (size_t)j * per_slot);
    ntt64_inverse(x, tabs + w, 0);
#pragma unroll
    for (int c = 0; c < 16; c++)
      *reinterpret_cast<uint4*>(lds + w * TILE_WORDS + lane * TILE_PITCH + c * 4) = make_uint4(x[c * 4 + 0], x[c * 4 + 1], x[c * 4 + 2], x[c * 4 + 3]);
  }
  __syncthreads();
  const uint64_t q = crt->q;
  const double inv_q = crt->inv_q;
  const uint32_t b32 = dg.base, bm = DIG ? (uint32_t)(0x100000000ull / dg.base) : 0;
  const uint64_t half = DIG ? (dg.base / 2) % q : 0;
  const bool fast3 = L == 3 && crt->fast3 != 0, fdiv = DIG && (q >> 32) < dg.base;   // (uniform)
  Crt3K k3{};
  if (L == 3 && fast3) k3 = crt3_load(crt);
  const unsigned sub = ns / nsub;
#pragma unroll 1
  for (int i = 0; i < 4; i++) {
    const int pl = i * 16 + (t >> 4), quad = t & 15;
    const size_t f = f0 + pl;
    const unsigned row = (unsigned)(f / ns), s = (unsigned)(f % ns), sl = s % nsub;
    if (row >= rows || sl >= ns_valid) continue;
    uint32_t r[4][L];
#pragma unroll
    for (int l = 0; l < L; l++) {
      const uint4 v = *reinterpret_cast<const uint4*>(lds + l * TILE_WORDS + pl * TILE_PITCH + quad * 4);
      r[0][l] = v.x; r[1][l] = v.y; r[2][l] = v.z; r[3][l] = v.w;
    }
    uint64_t v[4];
    if constexpr (L == 3) {
      if (fast3) {
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = crt3_fast(r[c][0], r[c][1], r[c][2], k3);
      } else {
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = crt_one<L>(r[c], crt, q, inv_q);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++) v[c] = crt_one<L>(r[c], crt, q, inv_q);
    }
    const unsigned proof = blockIdx.y * sub + s / nsub;
    const size_t e = (size_t)sl * set_stride + row;          // polynomial of the proof's result
    const unsigned w0 = quad * 4;                           // first of the four coefficients
    uint64_t* const o64 = static_cast<uint64_t*>(dstp.p[proof]) + e * D + w0;
    ulonglong2 o0, o1;
    o0.x = v[0]; o0.y = v[1]; o1.x = v[2]; o1.y = v[3];
    reinterpret_cast<ulonglong2*>(o64)[0] = o0;
    reinterpret_cast<ulonglong2*>(o64)[1] = o1;
    if constexpr (DIG) {
      const size_t g = dg.group, eg = e / g;
      uint16_t* const o = static_cast<uint16_t*>(dg.out16.p[proof]) + (eg * dg.exp * g + (e - eg * g)) * D + w0;
      digits_of_quad(v, o, g * D, dg.exp, b32, bm, half, fdiv);
    }
  }
}

// `stage`: nlimbs staging arrays stage_ps words apart, gemm_jobs jobs of ns sets each; coeffs.p[proof] (and dg->out16.p[proof])
// for proof = gemm job * (ns / nsub) + s / nsub.  hipErrorNotSupported when the digits do not have the 16-bit fast form.
hipError_t launch_stage_to_coeffs(const uint32_t* stage, size_t stage_ps, size_t rows, size_t rows_pad, size_t ns, size_t ns_valid,
                                  size_t nsub, size_t gemm_jobs, const JobPtrs& coeffs, size_t set_stride, const LimbTable* tabs,
                                  const CrtTable* crt, int nlimbs, hipStream_t s, const CrtDigits* dg) {
  if (rows == 0 || gemm_jobs == 0) return hipSuccess;
  if ((rows_pad * ns) % 64 || ns % nsub || gemm_jobs * (ns / nsub) > (size_t)MAX_JOBS || gemm_jobs > 65535) return hipErrorInvalidValue;
  if (dg && (dg->base >= 65536 || dg->base < 2 || (uint64_t)dg->base > dg->q || dg->q >= (1ull << 40) || dg->group == 0 || dg->exp <= 0 ||
             dg->group >> 32))
    return hipErrorNotSupported;
  const dim3 g((unsigned)(rows_pad * ns / 64), (unsigned)gemm_jobs);
  CrtDigits none{};
#define LAB_STC_CASE(LL)                                                                                                             \
  case LL:                                                                                                                           \
    if (dg) hipLaunchKernelGGL((k_stage_to_coeffs<LL, true>), g, dim3(256), 0, s, stage, stage_ps, (unsigned)rows, (unsigned)rows_pad, \
                               (unsigned)ns, (unsigned)ns_valid, (unsigned)nsub, coeffs, set_stride, tabs, crt, *dg);                  \
    else hipLaunchKernelGGL((k_stage_to_coeffs<LL, false>), g, dim3(256), 0, s, stage, stage_ps, (unsigned)rows, (unsigned)rows_pad, \
                            (unsigned)ns, (unsigned)ns_valid, (unsigned)nsub, coeffs, set_stride, tabs, crt, none);                    \
    break;
  switch (nlimbs) {
    LAB_STC_CASE(2) LAB_STC_CASE(3)
    default: return hipErrorNotSupported;
  }
#undef LAB_STC_CASE
  return hipGetLastError();
}

// dg (optional): also write the decomposition digits of every value (see CrtDigits in kernels.h); returns
// hipErrorNotSupported when the digits do not have the 16-bit fast form
hipError_t launch_crt_to_u64_jobs(ConstHatView src, size_t src_js, const JobPtrs& coeffs, size_t count, const CrtTable* crt,
                                  int nlimbs, size_t njobs, hipStream_t s, const CrtDigits* dg) {
  if (count == 0 || njobs == 0) return hipSuccess;
  if ((src.plane_stride | src_js) & 3) return hipErrorInvalidValue;
  if (dg && (dg->base >= 65536 || dg->base < 2 || (uint64_t)dg->base > dg->q || dg->q >= (1ull << 40) || dg->group == 0 || dg->exp <= 0 ||
             count >> 32 || dg->group >> 32))
    return hipErrorNotSupported;
  const size_t nquads = count * D / 4;
  unsigned blocks = (unsigned)((nquads + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  const dim3 g(blocks, (unsigned)njobs);
  CrtDigits none{};
#define LAB_CRT_CASE(LL)                                                                                                             \
  case LL:                                                                                                                           \
    if (dg) hipLaunchKernelGGL((k_crt_to_u64<LL, true>), g, dim3(256), 0, s, src.base, src.plane_stride, coeffs, nquads, crt, src_js, *dg); \
    else hipLaunchKernelGGL((k_crt_to_u64<LL, false>), g, dim3(256), 0, s, src.base, src.plane_stride, coeffs, nquads, crt, src_js, none); \
    break;
  switch (nlimbs) {
    LAB_CRT_CASE(1) LAB_CRT_CASE(2) LAB_CRT_CASE(3) LAB_CRT_CASE(4)
    default: return hipErrorInvalidValue;
  }
#undef LAB_CRT_CASE
  return hipGetLastError();
}

hipError_t launch_crt_to_u64(ConstHatView src, uint64_t* coeffs, size_t count, const CrtTable* crt, int nlimbs, hipStream_t s) {
  JobPtrs t;
  t.p[0] = coeffs;
  return launch_crt_to_u64_jobs(src, 0, t, count, crt, nlimbs, 1, s, nullptr);
}

__global__ void __launch_bounds__(256)
k_synth_fill_u64(uint64_t* __restrict__ coeffs, size_t nwords, uint64_t seed_mixed, uint64_t first_word,
                 uint64_t bound) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nwords; i += stride) coeffs[i] = synth_residue(seed_mixed, first_word + i, bound);
}

static uint64_t host_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

hipError_t launch_synth_fill_u64(uint64_t* coeffs, size_t count, uint64_t seed, uint64_t first_elem, uint64_t bound,
                                 hipStream_t s) {
  if (count == 0) return hipSuccess;
  size_t nwords = count * D;
  unsigned blocks = (unsigned)((nwords + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_synth_fill_u64, dim3(blocks), dim3(256), 0, s, coeffs, nwords, host_splitmix64(seed),
                     first_elem * D, bound);
  return hipGetLastError();
}

}  // namespace lab
