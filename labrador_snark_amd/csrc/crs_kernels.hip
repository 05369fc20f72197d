// crs_kernels.hip — on-device expansion of the reference's CRS random oracle
// (src/structs.rs:27-190): every coefficient is
//     ChaCha20Rng::from_seed(base_seed + offset).gen_range(0..Q)      (src/structs.rs:167-171)
// with a 256-bit big-endian seed incremented once per coefficient (:155-165).
// One thread per coefficient: 256-bit seed add, ChaCha20 block(s) (RFC 7539
// block function, 64-bit counter / zero stream as rand_chacha 0.3.1 lays it
// out), rand 0.8.5's UniformInt<i128>::sample_single (u128 candidate from four
// consecutive words, widening multiply by Q, accept if lo <= (Q << lz) - 1).
// ~1 100 integer ops per coefficient: VALU-bound (≈3·10^10 coefficients/s).
// gfx950 only.  SURVEY §8(f) row f1.
#include "kernels.h"

namespace lab {

struct CrsSeed {
  uint32_t limb[8];  // base seed as a 256-bit big-endian integer: limb[0] most significant
};

#define LAB_QR(a, b, c, d)                                        \
  a += b; d ^= a; d = __builtin_rotateleft32(d, 16);              \
  c += d; b ^= c; b = __builtin_rotateleft32(b, 12);              \
  a += b; d ^= a; d = __builtin_rotateleft32(d, 8);               \
  c += d; b ^= c; b = __builtin_rotateleft32(b, 7);

// One ChaCha20 block of the generator seeded with base + offset + i (block counter `counter`), and rand's draws from it:
// candidate d = words 4d .. 4d+3.  Returns true (and the value) at the first accepted candidate.
__device__ __forceinline__ bool crs_block_draw(const CrsSeed& base, uint64_t off_lo, uint64_t off_hi, uint64_t i, uint32_t counter, uint64_t q,
                                               uint64_t zone_lo, uint64_t zone_hi, uint64_t* result) {
  // seed = base + offset + i  (256-bit big-endian arithmetic, wraps at 2^256 like increment_seed)
  const uint64_t add_lo = off_lo + i;
  const uint64_t add_hi = off_hi + (add_lo < off_lo ? 1 : 0);
  const uint32_t addw[4] = {(uint32_t)add_lo, (uint32_t)(add_lo >> 32), (uint32_t)add_hi, (uint32_t)(add_hi >> 32)};
  uint32_t key[8];
  uint32_t carry = 0;
#pragma unroll
  for (int w = 7; w >= 0; w--) {
    const uint64_t s = (uint64_t)base.limb[w] + (w >= 4 ? addw[7 - w] : 0u) + carry;
    carry = (uint32_t)(s >> 32);
    // the seed BYTES are the big-endian integer; ChaCha's key words are little-endian words of those bytes
    key[w] = __builtin_bswap32((uint32_t)s);
  }
  uint32_t x0 = 0x61707865u, x1 = 0x3320646eu, x2 = 0x79622d32u, x3 = 0x6b206574u;
  uint32_t x4 = key[0], x5 = key[1], x6 = key[2], x7 = key[3], x8 = key[4], x9 = key[5], x10 = key[6], x11 = key[7];
  uint32_t x12 = counter, x13 = 0, x14 = 0, x15 = 0;
#pragma unroll
  for (int rnd = 0; rnd < 10; rnd++) {
    LAB_QR(x0, x4, x8, x12) LAB_QR(x1, x5, x9, x13) LAB_QR(x2, x6, x10, x14) LAB_QR(x3, x7, x11, x15)
    LAB_QR(x0, x5, x10, x15) LAB_QR(x1, x6, x11, x12) LAB_QR(x2, x7, x8, x13) LAB_QR(x3, x4, x9, x14)
  }
  const uint32_t blk[16] = {x0 + 0x61707865u, x1 + 0x3320646eu, x2 + 0x79622d32u, x3 + 0x6b206574u,
                            x4 + key[0], x5 + key[1], x6 + key[2], x7 + key[3],
                            x8 + key[4], x9 + key[5], x10 + key[6], x11 + key[7],
                            x12 + counter, x13, x14, x15};
  bool done = false;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    if (!done) {
      const uint64_t x = (uint64_t)blk[4 * d] | ((uint64_t)blk[4 * d + 1] << 32);      // first next_u64()
      const uint64_t y = (uint64_t)blk[4 * d + 2] | ((uint64_t)blk[4 * d + 3] << 32);  // second: high half
      // (y*2^64 + x) * q as 256 bits; q < 2^40 (each 64 x 64 product once: its low and high words share the partial products)
      const unsigned __int128 pa = (unsigned __int128)x * q, pb = (unsigned __int128)y * q;
      const uint64_t a_lo = (uint64_t)pa, a_hi = (uint64_t)(pa >> 64), b_lo = (uint64_t)pb, b_hi = (uint64_t)(pb >> 64);
      const uint64_t t = a_hi + b_lo;
      const uint64_t c = t < a_hi ? 1 : 0;
      const uint64_t lo_lo = a_lo, lo_hi = t, hi = b_hi + c;
      const bool ok = (lo_hi < zone_hi) || (lo_hi == zone_hi && lo_lo <= zone_lo);
      if (ok) { *result = hi; done = true; }
    }
  }
  return done;
}

// rand 0.8.5's UniformInt<i128>::sample_single on the four candidates of one block (candidate d = words 4d .. 4d+3)
__device__ __forceinline__ bool crs_draw(const uint32_t (&blk)[16], uint64_t q, uint64_t zone_lo, uint64_t zone_hi, uint64_t* result) {
  bool done = false;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    if (!done) {
      const uint64_t x = (uint64_t)blk[4 * d] | ((uint64_t)blk[4 * d + 1] << 32);
      const uint64_t y = (uint64_t)blk[4 * d + 2] | ((uint64_t)blk[4 * d + 3] << 32);
      const unsigned __int128 pa = (unsigned __int128)x * q, pb = (unsigned __int128)y * q;
      const uint64_t a_lo = (uint64_t)pa, a_hi = (uint64_t)(pa >> 64), b_lo = (uint64_t)pb, b_hi = (uint64_t)(pb >> 64);
      const uint64_t t = a_hi + b_lo;
      const uint64_t c = t < a_hi ? 1 : 0;
      const bool ok = (t < zone_hi) || (t == zone_hi && a_lo <= zone_lo);
      if (ok) { *result = b_hi + c; done = true; }
    }
  }
  return done;
}

// What the first ChaCha20 block (counter 0) of every generator of one launch has in common.  The seed of coefficient i is the
// 256-bit integer U = base + offset + i; as long as adding i does not carry out of the low 64 bits, key words 0..5 are those of
// base + offset: the column-round quarter rounds (x0, x4, x8, x12) and (x1, x5, x9, x13) are then the same for every coefficient
// (computed once on the host), and so are the first three operations of the other two (x14 = x15 = 0).
struct CrsFast {
  uint32_t key[6];     // key words 0..5 (little-endian words of the big-endian seed bytes)
  uint64_t low;        // low 64 bits of base + offset
  uint32_t qr0[4];     // x0, x4, x8, x12 after the first column round
  uint32_t qr1[4];     // x1, x5, x9, x13
  uint32_t a2, d2;     // x2 + x6 and rotl16 of it
  uint32_t a3, d3;     // x3 + x7 and rotl16 of it
};

// the second half of a quarter round whose a += b; d ^= a; d <<<= 16 was done on the host
#define LAB_QR_TAIL(a, b, c, d)                                   \
  c += d; b ^= c; b = __builtin_rotateleft32(b, 12);              \
  a += b; d ^= a; d = __builtin_rotateleft32(d, 8);               \
  c += d; b ^= c; b = __builtin_rotateleft32(b, 7);

// First block of coefficient i on the fast path; false when the coefficient needs the general path (carry out of the low
// 64 bits of the seed, or all four candidates rejected — *carry tells which)
__device__ __forceinline__ bool crs_first_block(const CrsFast& f, uint64_t i, uint64_t q, uint64_t zone_lo, uint64_t zone_hi,
                                                uint64_t* result, bool* carry) {
  const uint64_t lo = f.low + i;
  *carry = lo < f.low;
  const uint32_t k6 = __builtin_bswap32((uint32_t)(lo >> 32)), k7 = __builtin_bswap32((uint32_t)lo);
  uint32_t x0 = f.qr0[0], x4 = f.qr0[1], x8 = f.qr0[2], x12 = f.qr0[3];
  uint32_t x1 = f.qr1[0], x5 = f.qr1[1], x9 = f.qr1[2], x13 = f.qr1[3];
  uint32_t x2 = f.a2, x6 = f.key[2], x10 = k6, x14 = f.d2;
  uint32_t x3 = f.a3, x7 = f.key[3], x11 = k7, x15 = f.d3;
  LAB_QR_TAIL(x2, x6, x10, x14) LAB_QR_TAIL(x3, x7, x11, x15)
  LAB_QR(x0, x5, x10, x15) LAB_QR(x1, x6, x11, x12) LAB_QR(x2, x7, x8, x13) LAB_QR(x3, x4, x9, x14)
#pragma unroll
  for (int rnd = 1; rnd < 10; rnd++) {
    LAB_QR(x0, x4, x8, x12) LAB_QR(x1, x5, x9, x13) LAB_QR(x2, x6, x10, x14) LAB_QR(x3, x7, x11, x15)
    LAB_QR(x0, x5, x10, x15) LAB_QR(x1, x6, x11, x12) LAB_QR(x2, x7, x8, x13) LAB_QR(x3, x4, x9, x14)
  }
  const uint32_t blk[16] = {x0 + 0x61707865u, x1 + 0x3320646eu, x2 + 0x79622d32u, x3 + 0x6b206574u,
                            x4 + f.key[0], x5 + f.key[1], x6 + f.key[2], x7 + f.key[3],
                            x8 + f.key[4], x9 + f.key[5], x10 + k6, x11 + k7,
                            x12, x13, x14, x15};
  return crs_draw(blk, q, zone_lo, zone_hi, result);
}

// Block = 1024 coefficients per pass.  Every thread computes the FIRST block of its generator; one block holds four candidates
// and a candidate is rejected with probability up to 1/2 (the 33-bit q), so about one coefficient in 16 needs a second block —
// i.e. nearly every 64-lane wave would have to run it.  The rejected coefficients of CRS_GROUP passes are therefore queued in
// LDS and worked off together by as many threads as there are entries (further blocks of those generators only): ~6 %
// extra block computations instead of ~100 %, and the queue phase keeps about half of the block's waves busy.  The same queue
// takes the (one in 2^64 / count) coefficients whose seed addition carries out of the low 64 bits: they start over at block 0 on
// the general code.
constexpr int CRS_BLOCK = 1024, CRS_GROUP = 8;
__global__ void __launch_bounds__(CRS_BLOCK)
k_crs_expand(uint64_t* __restrict__ out, size_t nwords, CrsSeed base, CrsFast fast, uint64_t off_lo, uint64_t off_hi, uint64_t q,
             uint64_t zone_lo, uint64_t zone_hi) {
  __shared__ uint32_t queue[CRS_GROUP * CRS_BLOCK];      // carry << 31 | (pass of the group << 10) | thread: room for every coefficient
  __shared__ uint32_t qlen;
  const size_t pass_stride = (size_t)gridDim.x * CRS_BLOCK;
  for (size_t g0 = (size_t)blockIdx.x * CRS_BLOCK; g0 < nwords; g0 += pass_stride * CRS_GROUP) {
    if (threadIdx.x == 0) qlen = 0;
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g < CRS_GROUP; g++) {
      const size_t i = g0 + g * pass_stride + threadIdx.x;
      if (i < nwords) {
        uint64_t r;
        bool carry;
        const bool ok = crs_first_block(fast, (uint64_t)i, q, zone_lo, zone_hi, &r, &carry);
        if (ok && !carry) out[i] = r;
        else queue[atomicAdd(&qlen, 1u)] = (carry ? 0x80000000u : 0u) | ((uint32_t)g << 10) | threadIdx.x;
      }
    }
    __syncthreads();
    const uint32_t m = qlen;
    for (uint32_t e = threadIdx.x; e < m; e += CRS_BLOCK) {
      const uint32_t ent = queue[e];
      const size_t k = g0 + (size_t)((ent >> 10) & 0x1FFFFFu) * pass_stride + (ent & 1023u);
      uint64_t r = 0;
      for (uint32_t counter = (ent >> 31) ? 0u : 1u;; counter++)
        if (crs_block_draw(base, off_lo, off_hi, (uint64_t)k, counter, q, zone_lo, zone_hi, &r)) break;
      out[k] = r;
    }
    __syncthreads();
  }
}

static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
static inline void host_qr(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
  a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12);
  a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7);
}

hipError_t launch_crs_expand(uint64_t* coeffs, size_t count, const uint8_t base_seed[32], uint64_t off_lo,
                             uint64_t off_hi, uint64_t q, hipStream_t s) {
  if (count == 0) return hipSuccess;
  CrsSeed base;
  for (int w = 0; w < 8; w++)
    base.limb[w] = ((uint32_t)base_seed[4 * w] << 24) | ((uint32_t)base_seed[4 * w + 1] << 16) |
                   ((uint32_t)base_seed[4 * w + 2] << 8) | (uint32_t)base_seed[4 * w + 3];
  // zone = (q << leading_zeros_128(q)) - 1
  unsigned __int128 range = q;
  int lz = 64 + __builtin_clzll(q);
  unsigned __int128 zone = (range << lz) - 1;
  size_t nwords = count * D;
  // a block works off its queue of rejected coefficients once per CRS_GROUP passes: give every block a whole group when the
  // launch is large enough to fill the chip that way (256 CUs x 2 blocks), never more than 8192 blocks
  unsigned blocks = (unsigned)((nwords + CRS_BLOCK - 1) / CRS_BLOCK);
  const unsigned grouped = (unsigned)((nwords + (size_t)CRS_BLOCK * CRS_GROUP - 1) / ((size_t)CRS_BLOCK * CRS_GROUP));
  if (grouped >= 512) blocks = grouped;
  else if (blocks > 512) blocks = 512;
  if (blocks > 8192) blocks = 8192;
  // U0 = base + offset (256-bit big-endian limbs) and the part of the first block that depends on its upper 192 bits only
  uint32_t u[8];
  {
    const uint32_t addw[4] = {(uint32_t)off_lo, (uint32_t)(off_lo >> 32), (uint32_t)off_hi, (uint32_t)(off_hi >> 32)};
    uint32_t carry = 0;
    for (int w = 7; w >= 0; w--) {
      const uint64_t t = (uint64_t)base.limb[w] + (w >= 4 ? addw[7 - w] : 0u) + carry;
      carry = (uint32_t)(t >> 32);
      u[w] = (uint32_t)t;
    }
  }
  CrsFast f;
  for (int w = 0; w < 6; w++) f.key[w] = __builtin_bswap32(u[w]);
  f.low = ((uint64_t)u[6] << 32) | u[7];
  f.qr0[0] = 0x61707865u; f.qr0[1] = f.key[0]; f.qr0[2] = f.key[4]; f.qr0[3] = 0;   // x12 = block counter 0
  host_qr(f.qr0[0], f.qr0[1], f.qr0[2], f.qr0[3]);
  f.qr1[0] = 0x3320646eu; f.qr1[1] = f.key[1]; f.qr1[2] = f.key[5]; f.qr1[3] = 0;
  host_qr(f.qr1[0], f.qr1[1], f.qr1[2], f.qr1[3]);
  f.a2 = 0x79622d32u + f.key[2]; f.d2 = rotl32(f.a2, 16);                               // x14 = 0: d ^= a is a
  f.a3 = 0x6b206574u + f.key[3]; f.d3 = rotl32(f.a3, 16);
  hipLaunchKernelGGL(k_crs_expand, dim3(blocks), dim3(CRS_BLOCK), 0, s, coeffs, nwords, base, f, off_lo, off_hi, q,
                     (uint64_t)zone, (uint64_t)(zone >> 64));
  return hipGetLastError();
}

}  // namespace lab
