// zq_device.h — modular arithmetic and the in-register 64-point negacyclic NTT
// for gfx950 (CDNA4).  Device-only; included by the *.hip kernel files.
//
// Replaces, on the device, the arithmetic the reference performs one i128 `%`
// at a time (src/algebraic.rs:24-38, 95-134) and the transform hidden inside
// `PLAN.negacyclic_polymul` (src/algebraic.rs:396).
//
// Design facts (measured with tools/ubench_valu.hip on MI355X):
//  * every 32-bit integer VALU op — add, min, mul_lo, mul_hi — issues at the
//    same rate (4 cycles / wave64); v_mad_u64_u32 ~1.2x that.  So the cost of a
//    modular multiply is its instruction COUNT, and 32-bit multiplies are not
//    the quarter-rate ops they are on older GCN parts.
//  * limb primes are < 2^30 so that Harvey's lazy butterflies keep values in
//    [0,4p) inside 32 bits and 8 products of canonical residues can be summed
//    in a 64-bit accumulator before its high word needs a reduction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lab {

constexpr int D = 64;          // src/constants.rs:15
constexpr int LOGD = 6;
constexpr int MAX_LIMBS = 4;

// Per-limb constants, resident in device memory (one per RNS prime).
struct LimbTable {
  uint32_t p;            // prime, p == 1 (mod 128), p < 2^30 (p < 2^32 when `wide`)
  uint32_t p2;           // 2p
  uint32_t pinv;         // p^-1 mod 2^32  (signed Montgomery REDC)
  uint32_t r1;           // 2^32 mod p
  uint32_t r2;           // 2^64 mod p
  uint32_t wide;         // 1: 2^30 <= p < 2^32, q itself as the only prime ("single-limb strict-32 path", see Zp<true> below): strict
                         // arithmetic, fwd/inv/scale_lo hold Montgomery forms w * 2^32 mod p in [..][0]
  uint32_t pad[2];
  uint32_t q_hi_mul;     // 2^32 mod p, used to fold 64-bit coefficients: x = hi*2^32+lo
  uint32_t q_hi_mul_sh;  // Shoup companion of q_hi_mul
  double inv_p;          // 1.0 / p for the division-free reductions (mod_u64)
  uint32_t pad2[4];
  // twiddles: index m+i of stage m (m = 1,2,4,...,32), psi^brv(m+i); [k][0]=w, [k][1]=floor(w*2^32/p)
  uint32_t fwd[D][2];
  uint32_t inv[D][2];    // inverse twiddles for Gentleman-Sande, same indexing
  // final INTT scaling n^-1 * R^e (e = 0..3) and the same times inv[1] (last-stage twiddle), Shoup pairs
  uint32_t scale_lo[4][2];
  uint32_t scale_hi[4][2];
};

// x mod m without an integer division: the quotient estimate from one double multiply is off by a few units at most
// (exactly: by less than 2 + x * 2^-52 / m), the remainder is then fixed up in exact 64-bit arithmetic.  m < 2^62.
__device__ __forceinline__ uint64_t mod_u64(uint64_t x, uint64_t m, double inv_m) {
  const uint64_t qh = (uint64_t)((double)x * inv_m);
  int64_t r = (int64_t)(x - qh * m);
  while (r < 0) r += (int64_t)m;
  while (r >= (int64_t)m) r -= (int64_t)m;
  return (uint64_t)r;
}
// floor(x / m) the same way
__device__ __forceinline__ uint64_t div_u64(uint64_t x, uint64_t m, double inv_m) {
  uint64_t qh = (uint64_t)((double)x * inv_m);
  int64_t r = (int64_t)(x - qh * m);
  while (r < 0) { r += (int64_t)m; qh--; }
  while (r >= (int64_t)m) { r -= (int64_t)m; qh++; }
  return qh;
}

// the same when x / m < 2^50 (the quotient estimate is then off by less than 3): branch-free fix-ups
__device__ __forceinline__ uint64_t mod_u64_near(uint64_t x, uint64_t m, double inv_m) {
  const uint64_t qh = (uint64_t)((double)x * inv_m);
  int64_t r = (int64_t)(x - qh * m);
  r += r < 0 ? (int64_t)m : 0;
  r += r < 0 ? (int64_t)m : 0;
  r -= r >= (int64_t)m ? (int64_t)m : 0;
  r -= r >= (int64_t)m ? (int64_t)m : 0;
  return (uint64_t)r;
}
__device__ __forceinline__ uint64_t div_u64_near(uint64_t x, uint64_t m, double inv_m) {
  uint64_t qh = (uint64_t)((double)x * inv_m);
  int64_t r = (int64_t)(x - qh * m);
#pragma unroll
  for (int k = 0; k < 2; k++) { const bool f = r < 0; r += f ? (int64_t)m : 0; qh -= f ? 1 : 0; }
#pragma unroll
  for (int k = 0; k < 2; k++) { const bool f = r >= (int64_t)m; r -= f ? (int64_t)m : 0; qh += f ? 1 : 0; }
  return qh;
}

__device__ __forceinline__ uint32_t csub(uint32_t x, uint32_t m) {  // x in [0,2m) -> [0,m)
  return min(x, x - m);
}

// Shoup multiplication by a constant w with companion wp = floor(w*2^32/p).
// Valid for any y < 2^32; result in [0,2p).
__device__ __forceinline__ uint32_t mul_shoup_lazy(uint32_t y, uint32_t w, uint32_t wp, uint32_t p) {
  uint32_t qh = __umulhi(y, wp);
  return y * w - qh * p;
}

// Signed Montgomery reduction of a 64-bit value T < p*2^32: returns T*2^-32 mod p, canonical.
__device__ __forceinline__ uint32_t redc64(uint64_t T, uint32_t p, uint32_t pinv) {
  uint32_t lo = (uint32_t)T, hi = (uint32_t)(T >> 32);
  uint32_t m = lo * pinv;
  uint32_t t = hi - __umulhi(m, p);  // in (-p, p) as a wrapped u32
  return min(t, t + p);
}

// a*b*2^-32 mod p for canonical a,b.
__device__ __forceinline__ uint32_t mont_mul(uint32_t a, uint32_t b, uint32_t p, uint32_t pinv) {
  return redc64((uint64_t)a * b, p, pinv);
}

// Bring the high word of a 64-bit accumulator back below p.  Precondition:
// acc < p*2^32 + 8*p^2 (at most 8 canonical products were added since the last call), so hi < 3p
// (the two conditional subtractions in fact accept any hi < 4p).
__device__ __forceinline__ uint64_t acc_fold(uint64_t acc, uint32_t p, uint32_t p2) {
  uint32_t hi = (uint32_t)(acc >> 32);
  hi = min(hi, hi - p2);
  hi = min(hi, hi - p);
  return ((uint64_t)hi << 32) | (uint32_t)acc;
}

// Cheaper fold for multiply-add bound loops: keeps the high word below 2p instead of p.  Precondition: the high word
// was < 2p before at most 8 canonical products were added, so it is < 2p + 8p^2/2^32 < 4p < 2^32 now (p < 2^30).
// acc_fold (which accepts any high word < 4p) must be applied before redc64.
__device__ __forceinline__ uint64_t acc_fold_lazy(uint64_t acc, uint32_t p2) {
  uint32_t hi = (uint32_t)(acc >> 32);
  hi = min(hi, hi - p2);
  return ((uint64_t)hi << 32) | (uint32_t)acc;
}

// ---- 64-point negacyclic NTT, one polynomial per thread, fully unrolled.
// Forward: Cooley-Tukey, natural order in, bit-reversed order out, values kept
// lazily in [0,4p) (Harvey), canonical on exit.  Twiddle indices are compile
// time constants and `tab` is wave-uniform, so the twiddles arrive through
// scalar loads.
__device__ __forceinline__ void ntt64_forward(uint32_t (&x)[D], const LimbTable* __restrict__ tab) {
  const uint32_t p = tab->p, p2 = tab->p2;
  int t = D;
#pragma unroll
  for (int m = 1; m < D; m <<= 1) {
    t >>= 1;
#pragma unroll
    for (int i = 0; i < m; i++) {
      const uint32_t w = tab->fwd[m + i][0], wp = tab->fwd[m + i][1];
#pragma unroll
      for (int j = 2 * i * t; j < 2 * i * t + t; j++) {
        uint32_t X = x[j];
        if (m > 1) X = csub(X, p2);            // stage 1 inputs are canonical already
        uint32_t T = mul_shoup_lazy(x[j + t], w, wp, p);
        x[j] = X + T;
        x[j + t] = X - T + p2;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < D; j++) x[j] = csub(csub(x[j], p2), p);
}

// Inverse: Gentleman-Sande, bit-reversed in, natural out, values in [0,2p),
// the n^-1 * R^e scaling folded into the last stage, canonical on exit.
__device__ __forceinline__ void ntt64_inverse(uint32_t (&x)[D], const LimbTable* __restrict__ tab, int e) {
  const uint32_t p = tab->p, p2 = tab->p2;
  int t = 1;
#pragma unroll
  for (int m = D; m > 2; m >>= 1) {
    const int h = m >> 1;
#pragma unroll
    for (int i = 0; i < h; i++) {
      const uint32_t w = tab->inv[h + i][0], wp = tab->inv[h + i][1];
#pragma unroll
      for (int j = 2 * i * t; j < 2 * i * t + t; j++) {
        uint32_t U = x[j], V = x[j + t];
        x[j] = csub(U + V, p2);
        x[j + t] = mul_shoup_lazy(U - V + p2, w, wp, p);
      }
    }
    t <<= 1;
  }
  const uint32_t s0 = tab->scale_lo[e][0], s0p = tab->scale_lo[e][1];
  const uint32_t s1 = tab->scale_hi[e][0], s1p = tab->scale_hi[e][1];
#pragma unroll
  for (int j = 0; j < D / 2; j++) {
    uint32_t U = x[j], V = x[j + D / 2];
    x[j] = csub(mul_shoup_lazy(U + V, s0, s0p, p), p);
    x[j + D / 2] = csub(mul_shoup_lazy(U - V + p2, s1, s1p, p), p);
  }
}

// ---- Arithmetic policy of the transform-domain kernels.  Zp<false>: limb primes below 2^30, lazy 32-bit arithmetic (values
// up to 4p in a word, 64-bit accumulators of products folded every 8 terms).  Zp<true>: the single-limb strict-32 path for an
// NTT-friendly q in [2^30, 2^32) (reference anchor: the commented 32-bit Q of src/constants.rs:195-196 behind the seam
// src/algebraic.rs:379-404): q itself is the only prime, residues are canonical 32-bit words (4 B per coefficient, no CRT),
// sums and differences are corrected with the carry, products are reduced exactly, and a sum of products is kept as two
// 64-bit accumulators (low and high words of the 64-bit products: no fold for 2^32 terms).
struct ZpConsts { uint32_t p, p2, pinv, r2; double inv_p; };
__device__ __forceinline__ ZpConsts zp_consts(const LimbTable* __restrict__ t) { return ZpConsts{t->p, t->p2, t->pinv, t->r2, t->inv_p}; }

template <bool WIDE> struct Zp;
template <> struct Zp<false> {
  typedef uint64_t acc_t;
  static __device__ __forceinline__ acc_t zero() { return 0; }
  static __device__ __forceinline__ void mad(acc_t& a, uint32_t x, uint32_t y) { a += (uint64_t)x * y; }
  static __device__ __forceinline__ acc_t fold(acc_t a, const ZpConsts& c) { return acc_fold(a, c.p, c.p2); }         // after <= 8 mads
  static __device__ __forceinline__ acc_t fold_lazy(acc_t a, const ZpConsts& c) { return acc_fold_lazy(a, c.p2); }
  // the sum as a plain canonical residue (the accumulator folded or holding at most 8 products)
  static __device__ __forceinline__ uint32_t finish(acc_t a, const ZpConsts& c) {
    return mont_mul(redc64(acc_fold(a, c.p, c.p2), c.p, c.pinv), c.r2, c.p, c.pinv);
  }
  static __device__ __forceinline__ uint32_t mul(uint32_t x, uint32_t y, const ZpConsts& c) {
    return mont_mul(mont_mul(x, y, c.p, c.pinv), c.r2, c.p, c.pinv);
  }
  static __device__ __forceinline__ uint32_t add(uint32_t x, uint32_t y, const ZpConsts& c) { return csub(x + y, c.p); }
  static __device__ __forceinline__ uint32_t sub(uint32_t x, uint32_t y, const ZpConsts& c) { return csub(x + c.p - y, c.p); }
};
template <> struct Zp<true> {
  struct acc_t { uint64_t lo, hi; };
  static __device__ __forceinline__ acc_t zero() { return acc_t{0, 0}; }
  static __device__ __forceinline__ void mad(acc_t& a, uint32_t x, uint32_t y) {
    const uint64_t t = (uint64_t)x * y;
    a.lo += (uint32_t)t;
    a.hi += t >> 32;
  }
  static __device__ __forceinline__ acc_t fold(acc_t a, const ZpConsts&) { return a; }
  static __device__ __forceinline__ acc_t fold_lazy(acc_t a, const ZpConsts&) { return a; }
  static __device__ __forceinline__ uint32_t add(uint32_t x, uint32_t y, const ZpConsts& c) {
    const uint32_t s = x + y;
    return (s < x || s >= c.p) ? s - c.p : s;
  }
  static __device__ __forceinline__ uint32_t sub(uint32_t x, uint32_t y, const ZpConsts& c) {
    const uint32_t d = x - y;
    return x < y ? d + c.p : d;
  }
  static __device__ __forceinline__ uint32_t mul(uint32_t x, uint32_t y, const ZpConsts& c) {
    return (uint32_t)mod_u64_near((uint64_t)x * y, c.p, c.inv_p);          // quotient < 2^32
  }
  static __device__ __forceinline__ uint32_t finish(acc_t a, const ZpConsts& c) {   // (hi * 2^32 + lo) mod p, quotients < 2^34
    const uint64_t h = mod_u64_near(a.hi, c.p, c.inv_p);
    const uint64_t t = mod_u64_near(h << 32, c.p, c.inv_p) + mod_u64_near(a.lo, c.p, c.inv_p);
    return (uint32_t)(t >= c.p ? t - c.p : t);
  }
  // x * w mod p for a constant held in Montgomery form wm = w * 2^32 mod p: signed REDC, the sign taken from the compare
  static __device__ __forceinline__ uint32_t mul_const(uint32_t x, uint32_t wm, uint32_t p, uint32_t pinv) {
    const uint64_t ab = (uint64_t)x * wm;
    const uint32_t hi = (uint32_t)(ab >> 32), mh = __umulhi((uint32_t)ab * pinv, p);
    const uint32_t t = hi - mh;
    return hi < mh ? t + p : t;
  }
};

// The 64-point transforms of the strict-32 path: canonical values throughout, twiddles in Montgomery form (LimbTable.wide).
__device__ __forceinline__ void ntt64_forward_wide(uint32_t (&x)[D], const LimbTable* __restrict__ tab) {
  const uint32_t p = tab->p, pinv = tab->pinv;
  const ZpConsts c{p, 0, pinv, 0, 0.0};
  int t = D;
#pragma unroll
  for (int m = 1; m < D; m <<= 1) {
    t >>= 1;
#pragma unroll
    for (int i = 0; i < m; i++) {
      const uint32_t w = tab->fwd[m + i][0];
#pragma unroll
      for (int j = 2 * i * t; j < 2 * i * t + t; j++) {
        const uint32_t X = x[j], T = Zp<true>::mul_const(x[j + t], w, p, pinv);
        x[j] = Zp<true>::add(X, T, c);
        x[j + t] = Zp<true>::sub(X, T, c);
      }
    }
  }
}
// inverse; the escale argument of the lazy path (Montgomery factors left by PW_MUL_KEEP_SCALE) has no meaning here:
// products of the strict path are always plain
__device__ __forceinline__ void ntt64_inverse_wide(uint32_t (&x)[D], const LimbTable* __restrict__ tab) {
  const uint32_t p = tab->p, pinv = tab->pinv;
  const ZpConsts c{p, 0, pinv, 0, 0.0};
  int t = 1;
#pragma unroll
  for (int m = D; m > 1; m >>= 1) {
    const int h = m >> 1;
#pragma unroll
    for (int i = 0; i < h; i++) {
      const uint32_t w = tab->inv[h + i][0];
#pragma unroll
      for (int j = 2 * i * t; j < 2 * i * t + t; j++) {
        const uint32_t U = x[j], V = x[j + t];
        x[j] = Zp<true>::add(U, V, c);
        x[j + t] = Zp<true>::mul_const(Zp<true>::sub(U, V, c), w, p, pinv);
      }
    }
    t <<= 1;
  }
  const uint32_t ninv = tab->scale_lo[0][0];
#pragma unroll
  for (int j = 0; j < D; j++) x[j] = Zp<true>::mul_const(x[j], ninv, p, pinv);
}
template <bool WIDE>
__device__ __forceinline__ void ntt64_fwd(uint32_t (&x)[D], const LimbTable* __restrict__ tab) {
  if constexpr (WIDE) ntt64_forward_wide(x, tab); else ntt64_forward(x, tab);
}
template <bool WIDE>
__device__ __forceinline__ void ntt64_inv(uint32_t (&x)[D], const LimbTable* __restrict__ tab, int e) {
  if constexpr (WIDE) ntt64_inverse_wide(x, tab); else ntt64_inverse(x, tab, e);
}

// ---- wave-tile transposition through LDS: a one-wave block owns 64 consecutive
// polynomials (16 KiB of u32).  Global accesses are 16 B per lane, fully
// coalesced (1 KiB per instruction); each thread ends up with one whole
// polynomial in registers.  Row pitch 68 words keeps ds_read_b128/ds_write_b128
// conflict-free in both directions.
constexpr int TILE_PITCH = D + 4;
constexpr int TILE_WORDS = D * TILE_PITCH;

// Loads polys [base, base+64) of `plane` (u32 [count][64]); rows past `count` read as zero.
__device__ __forceinline__ void tile_load_u32(uint32_t (&x)[D], const uint32_t* __restrict__ plane, size_t base,
                                              size_t count, uint32_t* lds, int lane) {
  const uint4* g = reinterpret_cast<const uint4*>(plane + base * D);
  if (base + 64 <= count) {  // wave-uniform: full tile, 16 independent loads in flight
    uint4 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      // batches are streamed once: non-temporal loads/stores measured +5..7 % on the 2^20-batch transform
      typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
      const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(g + i * 64 + lane));
      v[i] = make_uint4(t.x, t.y, t.z, t.w);
    }
#pragma unroll
    for (int i = 0; i < 16; i++)
      *reinterpret_cast<uint4*>(lds + (i * 4 + (lane >> 4)) * TILE_PITCH + (lane & 15) * 4) = v[i];
  } else {
#pragma unroll 1
    for (int i = 0; i < 16; i++) {
      int poly = i * 4 + (lane >> 4);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (base + poly < count) v = g[i * 64 + lane];
      *reinterpret_cast<uint4*>(lds + poly * TILE_PITCH + (lane & 15) * 4) = v;
    }
  }
  __syncthreads();  // block == one wave (launch_bounds 64): lowers to a wave-level barrier
#pragma unroll
  for (int c = 0; c < 16; c++) {
    uint4 v = *reinterpret_cast<const uint4*>(lds + lane * TILE_PITCH + c * 4);
    x[c * 4 + 0] = v.x; x[c * 4 + 1] = v.y; x[c * 4 + 2] = v.z; x[c * 4 + 3] = v.w;
  }
  __syncthreads();
}

__device__ __forceinline__ void tile_store_u32(const uint32_t (&x)[D], uint32_t* __restrict__ plane, size_t base,
                                               size_t count, uint32_t* lds, int lane) {
#pragma unroll
  for (int c = 0; c < 16; c++)
    *reinterpret_cast<uint4*>(lds + lane * TILE_PITCH + c * 4) =
        make_uint4(x[c * 4 + 0], x[c * 4 + 1], x[c * 4 + 2], x[c * 4 + 3]);
  __syncthreads();
  uint4* g = reinterpret_cast<uint4*>(plane + base * D);
  if (base + 64 <= count) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const uint4 o = *reinterpret_cast<const uint4*>(lds + (i * 4 + (lane >> 4)) * TILE_PITCH + (lane & 15) * 4);
      typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
      u32x4_t t = {o.x, o.y, o.z, o.w};
      __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(g + i * 64 + lane));
    }
  } else {
#pragma unroll 1
    for (int i = 0; i < 16; i++) {
      int poly = i * 4 + (lane >> 4);
      uint4 v = *reinterpret_cast<const uint4*>(lds + poly * TILE_PITCH + (lane & 15) * 4);
      if (base + poly < count) g[i * 64 + lane] = v;
    }
  }
  __syncthreads();
}

// splitmix64-based synthetic residues, identical to oracle orc_synth_residue (test/bench inputs).
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t synth_residue(uint64_t seed_mixed, uint64_t idx, uint64_t bound) {
  uint64_t h = splitmix64(seed_mixed ^ (idx * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull));
  return __umul64hi(h, bound);
}

// v < 2^40 mod a prime p < 2^23 (the limbs of a digit basis that are not main-basis primes): Barrett with m ~ 2^32 / p,
// all in 32 bits.  umulhi(x, m) underestimates floor(x / p) by at most 2, so the remainder is below 3p < 2^25.
__device__ __forceinline__ uint32_t reduce_small16(uint32_t hi8, uint32_t lo, uint32_t p, uint32_t m, uint32_t r1) {
  uint32_t b = lo - __umulhi(lo, m) * p;          // lo mod p, in [0, 3p)
  uint32_t s = hi8 * r1 + b;                       // hi * (2^32 mod p) + ...: < 2^8 * 2^23 + 2^25 < 2^32
  s -= __umulhi(s, m) * p;
  return csub(csub(s, 2 * p), p);
}

// ---- mod-q helpers of the coefficient-domain kernels (q < 2^40)
__device__ __forceinline__ uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) { uint64_t s = a + b; return s >= q ? s - q : s; }
__device__ __forceinline__ uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }


// x = hi*2^32 + lo < 2^40 divided by b < 2^16 in 32-bit arithmetic: two steps of long division in base 2^16, each a
// multiply-high by m = floor(2^32 / b) (underestimates the quotient by at most 2) and two conditional corrections.
// Returns the remainder; *quot = floor(x / b).
__device__ __forceinline__ uint32_t divmod_u40_u16(uint32_t hi, uint32_t lo, uint32_t b, uint32_t m, uint64_t* quot) {
  const uint32_t t1 = (hi << 16) | (lo >> 16);            // < 2^24
  uint32_t q1 = __umulhi(t1, m), r1 = t1 - q1 * b;
  { const bool f = r1 >= b; r1 -= f ? b : 0; q1 += f; }
  { const bool f = r1 >= b; r1 -= f ? b : 0; q1 += f; }
  const uint32_t t2 = (r1 << 16) | (lo & 0xFFFFu);        // < b * 2^16 <= 2^32
  uint32_t q2 = __umulhi(t2, m), r2 = t2 - q2 * b;
  { const bool f = r2 >= b; r2 -= f ? b : 0; q2 += f; }
  { const bool f = r2 >= b; r2 -= f ? b : 0; q2 += f; }
  *quot = ((uint64_t)q1 << 16) + q2;
  return r2;
}


// one step of the decomposition when the quotient fits 32 bits: v = quot * b + r0, by one double multiply (off by one at most)
__device__ __forceinline__ uint32_t divmod_by_double(uint64_t v, uint32_t b, double inv_b, uint32_t* quot) {
  uint32_t qh = (uint32_t)((double)v * inv_b);
  uint32_t r0 = (uint32_t)v - qh * b;                      // the true remainder lies in [-b, 2b): 32-bit wrap-around is harmless
  { const bool f = (int32_t)r0 < 0; r0 += f ? b : 0; qh -= f; }
  { const bool f = r0 >= b; r0 -= f ? b : 0; qh += f; }
  *quot = qh;
  return r0;
}
__device__ __forceinline__ uint32_t divmod_by_double(uint32_t v, uint32_t b, double inv_b, uint32_t* quot) {
  uint32_t qh = (uint32_t)((double)v * inv_b);
  uint32_t r0 = v - qh * b;
  { const bool f = (int32_t)r0 < 0; r0 += f ? b : 0; qh -= f; }
  { const bool f = r0 >= b; r0 -= f ? b : 0; qh += f; }
  *quot = qh;
  return r0;
}

}  // namespace lab
