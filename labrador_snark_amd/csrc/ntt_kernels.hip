// ntt_kernels.hip — batched 64-point negacyclic NTT/INTT and the conversions
// between the boundary's coefficient form (u64 mod q) and the device's hat form
// (u32 residues per RNS limb, NTT domain).  gfx950 only.
//
// Replaces the transform inside `PLAN.negacyclic_polymul` and the u64<->Zq
// marshalling around it (src/algebraic.rs:385-399, 407-420).
#include "kernels.h"

namespace lab {

// One polynomial per thread, one wave per block: the wave's 64 polynomials
// (16 KiB) are fetched with 16-byte coalesced loads, transposed through LDS,
// transformed entirely in registers (192 lazy butterflies, twiddles from
// scalar loads) and written back the same way.
// Algorithmic traffic: 256 B in + 256 B out per polynomial per limb.
template <bool FWD>
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
  if (FWD) ntt64_forward(x, tab); else ntt64_inverse(x, tab, escale);
  tile_store_u32(x, dst + (size_t)limb * dst_ps + (size_t)blockIdx.z * dst_js, base, count, lds, lane);
}

hipError_t launch_ntt(bool forward, ConstHatView src, HatView dst, size_t count, int nlimbs, const LimbTable* tabs,
                      int escale, hipStream_t s, size_t njobs, size_t src_js, size_t dst_js) {
  if (count == 0 || njobs == 0) return hipSuccess;
  dim3 grid((unsigned)((count + 63) / 64), (unsigned)nlimbs, (unsigned)njobs);
  if (forward)
    hipLaunchKernelGGL(k_ntt64<true>, grid, dim3(64), 0, s, src.base, src.plane_stride, dst.base, dst.plane_stride,
                       count, tabs, escale, src_js, dst_js);
  else
    hipLaunchKernelGGL(k_ntt64<false>, grid, dim3(64), 0, s, src.base, src.plane_stride, dst.base, dst.plane_stride,
                       count, tabs, escale, src_js, dst_js);
  return hipGetLastError();
}

// coefficient u64 in [0,q) -> residue mod each limb prime (q < 2^40, so x = hi*2^32 + lo with hi < 2^8).
// Division-free: hi*2^32 mod p by a Shoup multiplication with the tabulated 2^32 mod p; lo mod p by conditional
// subtractions when p > 2^29 (lo < 2^32 < 8p), by mod_u64 for the small limb of the digit basis.
__device__ __forceinline__ uint32_t reduce_u40(uint64_t v, const LimbTable& tb) {
  const uint32_t p = tb.p, lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  if (p > (1u << 29)) {
    uint32_t b = lo;
    b = min(b, b - 4 * p);   // wraps (and is discarded by min) unless b >= 4p; 4p < 2^32 as p < 2^30
    b = min(b, b - 2 * p);
    b = min(b, b - p);
    if (hi == 0) return b;
    const uint32_t a = mul_shoup_lazy(hi, tb.q_hi_mul, tb.q_hi_mul_sh, p);   // [0, 2p)
    return csub(csub(a + b, 2 * p), p);                                      // a + b < 3p < 2^32
  }
  return (uint32_t)mod_u64(v, p, tb.inv_p);
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
      *reinterpret_cast<uint2*>(dst + (size_t)l * dst_ps + i) = make_uint2(reduce_u40(v.x, tabs[l]), reduce_u40(v.y, tabs[l]));
  }
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
  // d < 2^30, c < q < 2^40: the product fits 64 bits when q < 2^34, else c is split at bit 20
  if (q < (1ull << 34)) return mod_u64((uint64_t)d * c, q, inv_q);
  const uint64_t chi = c >> 20, clo = c & 0xFFFFF;
  uint64_t t = mod_u64((uint64_t)d * chi, q, inv_q);
  t = mod_u64(t << 20, q, inv_q);
  return mod_u64(t + (uint64_t)d * clo, q, inv_q);
}

// hat residues (already inverse-transformed, canonical) -> signed CRT -> mod q.
// The integer being reconstructed is the exact negacyclic result; it lies in
// (-M/2, M/2), which lab_ctx guarantees through max_len.  No integer divisions: see mod_u64.
__global__ void __launch_bounds__(256)
k_crt_to_u64(const uint32_t* __restrict__ src, size_t src_ps, const JobPtrs dstp, size_t nwords,
             const CrtTable* __restrict__ crt, size_t src_js) {
  uint64_t* __restrict__ coeffs = static_cast<uint64_t*>(dstp.p[blockIdx.y]);
  src += (size_t)blockIdx.y * src_js;
  const int L = crt->nlimbs;
  const uint64_t q = crt->q;
  const double inv_q = crt->inv_q;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nwords; i += stride) {
    uint32_t r[MAX_LIMBS], d[MAX_LIMBS];
    for (int l = 0; l < L; l++) r[l] = src[(size_t)l * src_ps + i];
    if (L == 1 && (uint64_t)crt->p[0] == q) { coeffs[i] = r[0]; continue; }
    // Garner mixed-radix digits
    d[0] = r[0];
    for (int a = 1; a < L; a++) {
      const uint32_t p = crt->p[a];
      uint32_t t = r[a];
      for (int b = 0; b < a; b++) {
        // digits of the other limbs are < 2^30: below 2p for the 30-bit primes, anything for a small last prime
        const uint32_t db = p > (1u << 29) ? csub(d[b], p) : (uint32_t)mod_u64(d[b], p, crt->inv_p[a]);
        uint32_t diff = t >= db ? t - db : t + p - db;
        t = csub(mul_shoup_lazy(diff, crt->ginv[a][b][0], crt->ginv[a][b][1], p), p);
      }
      d[a] = t;
    }
    // sign: value > (M-1)/2 ?
    bool neg = false;
    for (int a = L - 1; a >= 0; a--) {
      if (d[a] != crt->half_digits[a]) { neg = d[a] > crt->half_digits[a]; break; }
    }
    uint64_t acc = d[0] < q ? d[0] : mod_u64(d[0], q, inv_q);
    for (int a = 1; a < L; a++) {
      acc += mulmod_q(d[a], crt->prefix_mod_q[a], q, inv_q);
      acc -= acc >= q ? q : 0;
    }
    if (neg) acc = acc >= crt->m_mod_q ? acc - crt->m_mod_q : acc + q - crt->m_mod_q;
    coeffs[i] = acc;
  }
}

hipError_t launch_crt_to_u64_jobs(ConstHatView src, size_t src_js, const JobPtrs& coeffs, size_t count, const CrtTable* crt,
                                  size_t njobs, hipStream_t s) {
  if (count == 0 || njobs == 0) return hipSuccess;
  size_t nwords = count * D;
  unsigned blocks = (unsigned)((nwords + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_crt_to_u64, dim3(blocks, (unsigned)njobs), dim3(256), 0, s, src.base, src.plane_stride, coeffs, nwords, crt,
                     src_js);
  return hipGetLastError();
}

hipError_t launch_crt_to_u64(ConstHatView src, uint64_t* coeffs, size_t count, const CrtTable* crt, hipStream_t s) {
  JobPtrs t;
  t.p[0] = coeffs;
  return launch_crt_to_u64_jobs(src, 0, t, count, crt, 1, s);
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
