// mfma_kernels.hip — the batched outer commitment  out_s[rho] = sum_{m,c} M[m][rho][c] * V_s[m][c]  (src/proofgen.rs:101-132
// for many proofs at once) as exact int8 GEMMs on the gfx950 matrix cores.
//
// With 8 or more vector sets the VALU formulation (linalg_kernels.hip) is bound by the issue rate of v_mad_u64_u32
// (1.65e13 multiply-adds/s chip-wide).  Here every residue x in [0, p) is written in signed base-256 digits
// x = sum_i d_i 256^i, d_i in [-128, 127] (PL = 4 digit planes for the 30-bit limb primes, 3 for the primes below 2^23 of
// the digit basis made for this path (lab_ctx_enable_digit_basis_mfma), 2 for a prime < 2^15), and
// per NTT slot j and limb the sum over k = (m, c) becomes PL x PL int8 GEMMs
//     D[i][rho][(i', s)] = sum_k Md_i[rho][k] * Vd_i'[k][s]          (int32, exact: |D| <= K * 2^14 < 2^31 for K < 2^17)
//     out_s[rho] = sum_{i,i'} 256^(i+i') D[i][rho][(i',s)]  mod p      (fused into the GEMM's epilogue)
// on v_mfma_i32_32x32x32_i8 (32768 multiply-adds per 32 cycles per SIMD: 16 — or 9 — byte products per modular product
// still leave several times the VALU rate; what limits the long launches is power).  The matrix is stored once, at CRS build time, in MFMA A-operand tile order:
//     M8[limb][slot j][plane i][row tile rt][k tile kt][lane 64][16 B]
// (lane l of a tile holds row rt*32 + (l & 31), k = kt*32 + 16*(l >> 5) + 0..15: one wave-wide 16-byte load is one
// A operand, 1 KiB contiguous), PL bytes per residue.  The vectors are re-packed per batch into B-operand
// tiles V8[limb][slot j][k tile][col tile][lane][16 B], column n = i' * NS + s.
#include "kernels.h"
#include <cstdlib>

namespace lab {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// signed base-256 digits of x (x < 2^31): digit i in [-128,127] for i < planes-1, the rest in the last plane
__device__ __forceinline__ void signed_digits(uint32_t x, int planes, int8_t (&d)[4]) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (i < planes - 1) {
      const int8_t lo = (int8_t)(x & 0xFF);
      d[i] = lo;
      x = (x - (uint32_t)(int32_t)lo) >> 8;   // exact: x - lo is a multiple of 256 (lo sign-extended)
    } else if (i == planes - 1) {
      d[i] = (int8_t)x;
    } else {
      d[i] = 0;
    }
  }
}

// ---- matrix -> A-operand tiles.  Block = (row rho, k tile kt): the 32 polynomials M[rho][kt*32 .. +31] (8 KiB of
// u32 residues) go through LDS, thread (j, h) then owns slot j and k half h and emits 16 bytes per plane.
template <typename MT>
__global__ void __launch_bounds__(128)
k_pack_matrix_i8(int8_t* __restrict__ dst, const MT* __restrict__ src, unsigned nmat, unsigned rows, unsigned cols,
                 unsigned rt_count, unsigned kt_count, int planes) {
  __shared__ uint32_t tile[32][D + 1];
  const unsigned rho = blockIdx.x, kt = blockIdx.y;
  const unsigned ktotal = nmat * cols;
  for (unsigned e = threadIdx.x; e < 32 * D; e += 128) {
    const unsigned q = e / D, w = e % D, k = kt * 32 + q;
    uint32_t v = 0;
    if (k < ktotal && rho < rows) v = src[(((size_t)(k / cols) * rows + rho) * cols + k % cols) * D + w];
    tile[q][w] = v;
  }
  __syncthreads();
  const unsigned j = threadIdx.x & 63, h = threadIdx.x >> 6;
  union { int8_t b[16]; uint4 v; } outp[4];
#pragma unroll
  for (int b = 0; b < 16; b++) {
    int8_t d[4];
    signed_digits(tile[16 * h + b][j], planes, d);
#pragma unroll
    for (int i = 0; i < 4; i++) outp[i].b[b] = d[i];
  }
  const unsigned rt = rho / 32, lane = (h << 5) | (rho & 31);
  for (int i = 0; i < planes; i++) {
    const size_t t = (((size_t)j * planes + i) * rt_count + rt) * kt_count + kt;
    *reinterpret_cast<uint4*>(dst + t * 1024 + lane * 16) = outp[i].v;
  }
}

// The same for u32 residues with whole 128-byte lines on both sides: block = 8 rows x one k tile (64 KiB of residues through
// LDS, 16-byte loads); a wave's lanes are 8 rows x 8 slots, so the 8 rows of a (slot, plane, k half) are 8 neighbouring
// lanes of the tile = one full line per store group (the one-row blocks above write 16-byte pieces of 64 different tiles per
// instruction: 0.9 TB/s, a quarter of the streamed outer commitment's time).
constexpr int PM_PP = D + 4;                 // words per polynomial in LDS (16-byte aligned)
constexpr int PM_RP = 32 * PM_PP + 8;        // words per row: 8 rows x 8 slots of a wave fall into 64 different banks

__device__ __forceinline__ uint32_t signed_digit_word(uint32_t x, uint32_t offs);
__device__ __forceinline__ uint32_t digit_offsets(int planes);

__global__ void __launch_bounds__(256)
k_pack_matrix_i8x8(int8_t* __restrict__ dst, const uint32_t* __restrict__ src, unsigned nmat, unsigned rows, unsigned cols,
                   unsigned rt_count, unsigned kt_count, int planes) {
  extern __shared__ __attribute__((aligned(16))) uint32_t pm_tile[];   // [8][PM_RP]
  const unsigned row0 = blockIdx.x * 8, kt = blockIdx.y, t = threadIdx.x;
  const unsigned ktotal = nmat * cols;
#pragma unroll 4
  for (int it = 0; it < 16; it++) {
    const unsigned id = it * 256 + t, r = id >> 9, rem = id & 511, qp = rem >> 4, piece = rem & 15;
    const unsigned k = kt * 32 + qp, rho = row0 + r;
    u32x4_t v = {0, 0, 0, 0};
    if (k < ktotal && rho < rows)
      v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(src + (((size_t)(k / cols) * rows + rho) * cols + k % cols) * D) + piece);
    *reinterpret_cast<u32x4_t*>(&pm_tile[r * PM_RP + qp * PM_PP + piece * 4]) = v;
  }
  __syncthreads();
  const uint32_t offs = digit_offsets(planes);
#pragma unroll 1
  for (int it = 0; it < 4; it++) {
    const unsigned id = it * 256 + t, r = id & 7, g = id >> 6, j = (g & 7) * 8 + ((id >> 3) & 7), h = g >> 3;
    uint32_t w[16];
#pragma unroll
    for (int b = 0; b < 16; b++) w[b] = signed_digit_word(pm_tile[r * PM_RP + (16 * h + b) * PM_PP + j], offs);
    const unsigned rho = row0 + r, rt = rho / 32, lane = (h << 5) | (rho & 31);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (i >= planes) break;
      u32x4_t o;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t sel = 0x04000400u + 0x01010101u * i;
        const uint32_t lo = __builtin_amdgcn_perm(w[4 * q + 1], w[4 * q], sel);
        const uint32_t hi = __builtin_amdgcn_perm(w[4 * q + 3], w[4 * q + 2], sel);
        o[q] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
      }
      const size_t tl = (((size_t)j * planes + i) * rt_count + rt) * kt_count + kt;
      *reinterpret_cast<u32x4_t*>(dst + tl * 1024 + lane * 16) = o;
    }
  }
}

hipError_t launch_pack_matrix_i8(int8_t* dst, const void* src, bool src_u16, size_t nmat, size_t rows, size_t cols,
                                 size_t rt_count, size_t kt_count, int planes, hipStream_t s) {
  if (!src_u16) {
    // more than the 64 KB default of dynamic LDS: the attribute belongs to the (function, current device) pair, so it is set on
    // every launch (a process may hold contexts on several devices; the call is a table lookup after the first time)
    const size_t lds = (size_t)8 * PM_RP * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pack_matrix_i8x8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pack_matrix_i8x8, dim3((unsigned)(rt_count * 4), (unsigned)kt_count), dim3(256), lds, s, dst,
                       static_cast<const uint32_t*>(src), (unsigned)nmat, (unsigned)rows, (unsigned)cols, (unsigned)rt_count,
                       (unsigned)kt_count, planes);
    return hipGetLastError();
  }
  dim3 g((unsigned)(rt_count * 32), (unsigned)kt_count);
  if (src_u16)
    hipLaunchKernelGGL(k_pack_matrix_i8<uint16_t>, g, dim3(128), 0, s, dst, static_cast<const uint16_t*>(src), (unsigned)nmat,
                       (unsigned)rows, (unsigned)cols, (unsigned)rt_count, (unsigned)kt_count, planes);
  else
    hipLaunchKernelGGL(k_pack_matrix_i8<uint32_t>, g, dim3(128), 0, s, dst, static_cast<const uint32_t*>(src), (unsigned)nmat,
                       (unsigned)rows, (unsigned)cols, (unsigned)rt_count, (unsigned)kt_count, planes);
  return hipGetLastError();
}

// ---- vectors -> B-operand tiles.  Block = (k tile kt, k half h, group of 8 sets): the 16 polynomials
// V_s[kt*32 + 16h .. +15] of its 8 sets (8 x 4 KiB, 16-byte loads) go through LDS; thread (s, j) then emits 16 bytes
// per digit plane i' into column n = i'*NS + s: the 8 sets of a group are 8 neighbouring lanes, i.e. one 128-byte
// line per (slot, plane).
constexpr int PV_SETS = 8;
constexpr int PV_PITCH = 16 * D + 4;     // words per set in LDS (16-byte aligned rows, 2-way bank conflicts at most)

// all four signed digits of x at once: adding 128 at every digit position below the top one and flipping those bits
// back gives byte i = digit i (x = y - sum 128*256^i, y = x + offs, so digit i = byte_i(y) - 128 = byte_i(y) ^ 0x80)
__device__ __forceinline__ uint32_t signed_digit_word(uint32_t x, uint32_t offs) { return (x + offs) ^ offs; }
__device__ __forceinline__ uint32_t digit_offsets(int planes) { return planes >= 4 ? 0x00808080u : planes == 3 ? 0x8080u : planes == 2 ? 0x80u : 0u; }

__global__ void __launch_bounds__(256)
k_pack_vectors_i8(int8_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t set_stride, size_t k_stride, unsigned ktotal,
                  unsigned kt_count, unsigned nt_count, int planes, unsigned ns, unsigned ns_valid, size_t src_job_stride,
                  size_t dst_job_stride, unsigned nsub, size_t src_sub_stride) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[PV_SETS * PV_PITCH];
  const unsigned groups = ns / PV_SETS, job = blockIdx.z / groups;
  const unsigned kt = blockIdx.x, h = blockIdx.y, s0 = (blockIdx.z % groups) * PV_SETS;
  src += (size_t)job * src_job_stride;
  dst += (size_t)job * dst_job_stride;
  const unsigned k0 = kt * 32 + 16 * h;
  // thread t fetches quad t of every set's 16 x 64-word slab (polynomial k0 + t/16, words 4*(t%16)..): all 8 loads are
  // issued before the first LDS write
  const unsigned wq = threadIdx.x, k = k0 + wq / (D / 4);
  u32x4_t v[PV_SETS];
#pragma unroll
  for (int sset = 0; sset < PV_SETS; sset++) {
    // a job's ns sets may come from ns / nsub source jobs of nsub sets each (two 32-column proofs as one 64-set job);
    // sets past ns_valid (per source job) are padding: zero columns, never read
    const unsigned sg = s0 + sset, sub = sg / nsub, sl = sg - sub * nsub;
    const u32x4_t z = {0, 0, 0, 0};
    v[sset] = (k < ktotal && sl < ns_valid)
                  ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(src + (size_t)sub * src_sub_stride + (size_t)sl * set_stride +
                                                                                (size_t)k * k_stride) + (wq % (D / 4)))
                  : z;
  }
#pragma unroll
  for (int sset = 0; sset < PV_SETS; sset++) *reinterpret_cast<u32x4_t*>(&lds[sset * PV_PITCH + wq * 4]) = v[sset];
  __syncthreads();
  const uint32_t offs = digit_offsets(planes);
#pragma unroll
  for (int it = 0; it < PV_SETS * D / 256; it++) {
    const unsigned e = threadIdx.x + 256 * it, sset = e % PV_SETS, j = e / PV_SETS;
    uint32_t w[16];
#pragma unroll
    for (int b = 0; b < 16; b++) w[b] = signed_digit_word(lds[sset * PV_PITCH + b * D + j], offs);
    // byte i of w[4q .. 4q+3] -> dword q of plane i
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (i >= planes) break;
      u32x4_t o;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t sel = 0x04000400u + 0x01010101u * i;   // perm selectors: bytes (i of src1, i of src0) twice
        const uint32_t lo = __builtin_amdgcn_perm(w[4 * q + 1], w[4 * q], sel);        // [.., .., b_i(w1), b_i(w0)] in the low half
        const uint32_t hi = __builtin_amdgcn_perm(w[4 * q + 3], w[4 * q + 2], sel);
        o[q] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
      }
      const unsigned n = i * ns + s0 + sset, nt = n / 32, lane = (h << 5) | (n % 32);
      const size_t t = ((size_t)j * kt_count + kt) * nt_count + nt;
      *reinterpret_cast<u32x4_t*>(dst + t * 1024 + lane * 16) = o;
    }
  }
}

// ---- decomposition digits -> B-operand tiles in one pass (the vectors of the outer commitment: to_hat of the digits
// and k_pack_vectors_i8 fused, for every limb; the transform-domain copy of the digits never exists).
// Block = 8 waves = 8 sets (proofs) x 64 consecutive polynomials k0..k0+63 of each set's digit array (16-bit words,
// [K][64]).  Thread (set, k) loads its polynomial (128 B), and for every limb transforms it in registers (the caller
// guarantees digits below every limb prime: no reduction), turns the 64 slot values into signed-digit words and hands
// them to the block through LDS, 16 slots at a time; thread (slot, k block of 16, set) then gathers the 16 words of its
// 16 k, picks each plane's byte with v_perm and stores 16 bytes per plane — the 8 sets of the block are 8 neighbouring
// lanes of the B tile, i.e. one 128-byte line per (slot, plane, k block).
template <int NSETS> constexpr int DV_ROW = NSETS * D + 4;  // words per slot row of the exchange buffer: NSETS sets x 64 k (+ padding, 16-byte aligned)

struct PlaneCounts { int n[MAX_LIMBS]; unsigned nt[MAX_LIMBS]; };   // digit planes and column tiles per limb

// MATRIX64: the sets are the columns of [ktotal][r] matrices of 64-bit coefficients below q < 2^40 (phi', W of steps 6-8), nsub
// columns of proof job * (ns / nsub) + s / nsub per tile set (src.p[proof]; element (x, col) at (x*r + col)*64); every limb
// prime below 2^29 (reduce_small16).  blockIdx.z = job, its tiles at + job * job_stride of every limb.
// NSETS = 8 or 4 sets (waves) per block: with 4, two blocks share a CU and one loads while the other transforms.
// FROM_T (with !MATRIX64): the digits are made here from the sets' t = A s (DigitsFromT, kernels.h) instead of being read.
template <bool MATRIX64, int NSETS, bool FROM_T = false>
__device__ __forceinline__ void digits_to_v8_body(const JobPtrs& src, size_t ktotal, int8_t* __restrict__ dst, size_t limb_stride,
                                                  unsigned kt_count, unsigned ns, unsigned ns_valid, int nlimbs,
                                                  const LimbTable* __restrict__ tabs, const PlaneCounts& planes, unsigned r, unsigned nsub,
                                                  size_t job_stride, const DigitsFromT& dt = DigitsFromT{}) {
  static_assert(!FROM_T || !MATRIX64, "digits on the fly: the 16-bit digit form");
  constexpr int ROW = DV_ROW<NSETS>, SB = NSETS == 8 ? 3 : 2;          // log2(NSETS)
  static_assert(NSETS == 8 || NSETS == 4, "8 or 4 sets per block");
  // [slot of the pass][set][k]: writes and reads conflict-free; MATRIX64: first the load tile [set][x][16 coefficients], pitch 36 words
  constexpr int LDSW = MATRIX64 && 64 * NSETS * 36 > 16 * ROW ? 64 * NSETS * 36 : 16 * ROW;
  __shared__ __attribute__((aligned(16))) uint32_t ex[LDSW];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const size_t k0 = (size_t)blockIdx.x * D;
  const unsigned s_mine = blockIdx.y * NSETS + w;
  // MATRIX64: low words and packed high bytes of the 64 coefficients; digits: two 16-bit digits per register (64 fewer registers
  // than one each)
  uint32_t clo[MATRIX64 ? D : D / 2];
  uint32_t chi[MATRIX64 ? D / 4 : 1];
  if constexpr (MATRIX64) {
    dst += (size_t)blockIdx.z * job_stride;
    const unsigned proof = blockIdx.z * (ns / nsub) + s_mine / nsub, col = s_mine % nsub;
    // the block's 64 x NSETS polynomials: for every x the NSETS columns are contiguous (NSETS * 512 B).  Four passes of 16
    // coefficients: piece id = it * threads + t -> polynomial id >> 3 (x = poly / NSETS, column poly % NSETS), 16-byte piece
    // id & 7 — eight lanes fetch one full 128-byte line — parked at row column * 64 + x of the tile, read back by the
    // thread that owns the polynomial (wave = column, lane = x).
    const unsigned col0 = col - w;                                    // first column of the block
    const uint64_t* mat = static_cast<const uint64_t*>(src.p[proof]);
    auto fetch = [&](int q, u32x4_t (&v)[8]) {
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const unsigned id = it * (64 * NSETS) + t, poly = id >> 3, piece = id & 7, x = poly / NSETS, c = poly % NSETS;
        const u32x4_t z = {0, 0, 0, 0};
        v[it] = (col0 + c < ns_valid && k0 + x < ktotal)
                    ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(mat + ((k0 + x) * r + col0 + c) * D + q * 16) + piece)
                    : z;
      }
    };
#pragma unroll
    for (int i = 0; i < D / 4; i++) chi[i] = 0;
    u32x4_t cur[8], nxt[8];
    fetch(0, cur);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (q < 3) fetch(q + 1, nxt);
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const unsigned id = it * (64 * NSETS) + t, poly = id >> 3, piece = id & 7, x = poly / NSETS, c = poly % NSETS;
        *reinterpret_cast<u32x4_t*>(&ex[(c * 64 + x) * 36 + piece * 4]) = cur[it];
      }
      __syncthreads();
#pragma unroll
      for (int piece = 0; piece < 8; piece++) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&ex[(w * 64 + lane) * 36 + piece * 4]);
        const int k = q * 16 + piece * 2;
        clo[k] = v.x; clo[k + 1] = v.z;
        chi[k / 4] |= ((v.y & 0xFFu) << (8 * (k & 3))) | ((v.w & 0xFFu) << (8 * ((k + 1) & 3)));
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 8; it++) cur[it] = nxt[it];
    }
  } else if constexpr (FROM_T) {
    const size_t kpos = k0 + lane;
    const bool have = s_mine < ns_valid && kpos < ktotal;
    const size_t mt = dt.m0 + kpos / dt.kappa, rho = kpos % dt.kappa;     // matrix (i, k) = (mt / t1, mt % t1), column rho
    const unsigned kd = (unsigned)(mt % dt.t1);                          // which digit
    const uint64_t* p = have ? static_cast<const uint64_t*>(src.p[s_mine]) + ((mt / dt.t1) * dt.kappa + rho) * D : nullptr;
    const uint32_t b32 = dt.base, bm = (uint32_t)(0x100000000ull / dt.base), half = dt.base / 2;
    const bool fdiv = (dt.q >> 32) < dt.base;                             // every quotient fits 32 bits (uniform)
    const double inv_b = 1.0 / (double)dt.base;
    auto digit = [&](uint64_t v) -> uint32_t {   // digit kd of v (decompose_polynomial_vec: centred remainder, the quotient goes on)
      uint32_t r0 = 0;
      if (fdiv) {
        uint32_t w32;
        r0 = divmod_by_double(v, b32, inv_b, &w32);
        for (unsigned j = 0; j < kd; j++) r0 = divmod_by_double(w32, b32, inv_b, &w32);
      } else {
        for (unsigned j = 0; j <= kd; j++) {
          uint64_t qq;
          r0 = divmod_u40_u16((uint32_t)(v >> 32), (uint32_t)v, b32, bm, &qq);
          v = qq;
        }
      }
      return r0 > half ? b32 - r0 : r0;
    };
#pragma unroll
    for (int i = 0; i < D / 2; i++) {
      ulonglong2 v = make_ulonglong2(0, 0);
      if (have) v = *reinterpret_cast<const ulonglong2*>(p + 2 * i);
      clo[i] = have ? (digit(v.x) | (digit(v.y) << 16)) : 0u;
    }
  } else {
    const bool have = s_mine < ns_valid && k0 + lane < ktotal;
    const uint16_t* p = have ? static_cast<const uint16_t*>(src.p[s_mine]) + (k0 + lane) * D : nullptr;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      u32x4_t v = {0, 0, 0, 0};
      if (have) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p) + i);
      clo[4 * i] = v.x; clo[4 * i + 1] = v.y; clo[4 * i + 2] = v.z; clo[4 * i + 3] = v.w;
    }
  }
  // gather role: set (fastest: neighbouring lanes of the B tile), k block, slot inside the pass
  const unsigned g_set = t & (NSETS - 1), g_kb = (t >> SB) & 3, g_jl = t >> (SB + 2);
  const unsigned g_s = blockIdx.y * NSETS + g_set;
#pragma unroll 1
  for (int l = 0; l < nlimbs; l++) {
    uint32_t x[D];
    if constexpr (MATRIX64) {
      const LimbTable* tab = tabs + l;
      const uint32_t p = tab->p, m = (uint32_t)(4294967296.0 * tab->inv_p) - 1, r1 = tab->r1;
#pragma unroll
      for (int k = 0; k < D; k++) x[k] = reduce_small16((chi[k / 4] >> (8 * (k & 3))) & 0xFFu, clo[k], p, m, r1);
    } else {
#pragma unroll
      for (int k = 0; k < D / 2; k++) { x[2 * k] = clo[k] & 0xFFFFu; x[2 * k + 1] = clo[k] >> 16; }
    }
    ntt64_forward(x, tabs + l);
    const int pl = planes.n[l];
    const uint32_t offs = digit_offsets(pl);
    const unsigned nt_count = planes.nt[l];
    int8_t* dl = dst + (size_t)l * limb_stride;
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int c = 0; c < 16; c++) ex[c * ROW + w * D + lane] = signed_digit_word(x[16 * q + c], offs);
      __syncthreads();
      uint32_t wd[16];
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&ex[g_jl * ROW + g_set * D + g_kb * 16 + 4 * b]);
        wd[4 * b] = v.x; wd[4 * b + 1] = v.y; wd[4 * b + 2] = v.z; wd[4 * b + 3] = v.w;
      }
      const unsigned j = 16 * q + g_jl;
      const size_t kt = k0 / 32 + g_kb / 2;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (i >= pl) break;
        u32x4_t o;
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
          const uint32_t sel = 0x04000400u + 0x01010101u * i;
          const uint32_t lo = __builtin_amdgcn_perm(wd[4 * qq + 1], wd[4 * qq], sel);
          const uint32_t hi = __builtin_amdgcn_perm(wd[4 * qq + 3], wd[4 * qq + 2], sel);
          o[qq] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        }
        const unsigned n = i * ns + g_s, nt = n / 32, ln = ((g_kb & 1) << 5) | (n % 32);
        if (kt < kt_count) *reinterpret_cast<u32x4_t*>(dl + (((size_t)j * kt_count + kt) * nt_count + nt) * 1024 + ln * 16) = o;
      }
      __syncthreads();
    }
  }
}

// the two instantiations as kernels of their own: the occupancy hint is per kernel.  Digits: 4 sets per block, <= 168 registers,
// three waves per SIMD (the kernel waits on LDS and store latency a third of the time at two; at 128 registers it spills);
// matrices: 256 registers, two.
template <bool MATRIX64, int NSETS> __global__ void k_digits_to_v8(const JobPtrs, size_t, int8_t*, size_t, unsigned, unsigned, unsigned, int,
                                                                   const LimbTable*, const PlaneCounts, unsigned, unsigned, size_t);
template <>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_digits_to_v8<false, 4>(const JobPtrs src, size_t ktotal, int8_t* __restrict__ dst, size_t limb_stride, unsigned kt_count, unsigned ns,
                         unsigned ns_valid, int nlimbs, const LimbTable* __restrict__ tabs, const PlaneCounts planes, unsigned r,
                         unsigned nsub, size_t job_stride) {
  digits_to_v8_body<false, 4>(src, ktotal, dst, limb_stride, kt_count, ns, ns_valid, nlimbs, tabs, planes, r, nsub, job_stride);
}
template <>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_digits_to_v8<true, 4>(const JobPtrs src, size_t ktotal, int8_t* __restrict__ dst, size_t limb_stride, unsigned kt_count, unsigned ns,
                        unsigned ns_valid, int nlimbs, const LimbTable* __restrict__ tabs, const PlaneCounts planes, unsigned r,
                        unsigned nsub, size_t job_stride) {
  digits_to_v8_body<true, 4>(src, ktotal, dst, limb_stride, kt_count, ns, ns_valid, nlimbs, tabs, planes, r, nsub, job_stride);
}

// the digits made on the fly: two waves per SIMD (the 64-bit loads and the divisions want registers; the pass is a thousandth of
// the streamed step it serves)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_digits_to_v8_from_t(const JobPtrs src, size_t ktotal, int8_t* __restrict__ dst, size_t limb_stride, unsigned kt_count, unsigned ns,
                      unsigned ns_valid, int nlimbs, const LimbTable* __restrict__ tabs, const PlaneCounts planes, const DigitsFromT dt) {
  digits_to_v8_body<false, 4, true>(src, ktotal, dst, limb_stride, kt_count, ns, ns_valid, nlimbs, tabs, planes, 0u, 1u, (size_t)0, dt);
}

// one launch for every limb; sets past ns_valid and polynomials past ktotal come out as zero columns / rows.  K tiles past
// ceil(ktotal / 64) * 2 (kt_count padding) are NOT written: the caller clears the buffer once.
hipError_t launch_digits_to_v8(const JobPtrs& src16, size_t ktotal, int8_t* dst, size_t limb_stride, size_t kt_count, int ns,
                               int ns_valid, int nlimbs, const LimbTable* tabs, const int* planes, const size_t* nt, hipStream_t s,
                               const DigitsFromT* dt) {
  if (ns % 8 || ns > 64 || nlimbs > MAX_LIMBS) return hipErrorInvalidValue;
  PlaneCounts pc;
  for (int l = 0; l < MAX_LIMBS; l++) { pc.n[l] = l < nlimbs ? planes[l] : 0; pc.nt[l] = l < nlimbs ? (unsigned)nt[l] : 0; }
  if (dt) {
    if (dt->base < 2 || dt->base >= 65536 || (uint64_t)dt->base > dt->q || dt->q >= (1ull << 40) || dt->t1 == 0 || dt->kappa == 0)
      return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_digits_to_v8_from_t, dim3((unsigned)((ktotal + D - 1) / D), (unsigned)(ns / 4)), dim3(256), 0, s, src16, ktotal, dst,
                       limb_stride, (unsigned)kt_count, (unsigned)ns, (unsigned)ns_valid, nlimbs, tabs, pc, *dt);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((k_digits_to_v8<false, 4>), dim3((unsigned)((ktotal + D - 1) / D), (unsigned)(ns / 4)), dim3(256), 0, s, src16, ktotal, dst,
                     limb_stride, (unsigned)kt_count, (unsigned)ns, (unsigned)ns_valid, nlimbs, tabs, pc, 0u, 1u, (size_t)0);
  return hipGetLastError();
}

// the columns of [n][r] coefficient matrices (one per proof, 64-bit words below q < 2^40) -> reduce -> transform -> B-operand
// tiles of every limb: `ns` (32 or 64) sets per job = ns / 32 proofs x 32 columns (r <= 32 of them valid)
hipError_t launch_coefs_to_v8(const JobPtrs& src64, size_t n, size_t r, int8_t* dst, size_t limb_stride, size_t job_stride, size_t kt_count,
                              int ns, size_t njobs, int nlimbs, const LimbTable* tabs, const int* planes, const size_t* nt, hipStream_t s) {
  if ((ns != 32 && ns != 64) || r > 32 || nlimbs > MAX_LIMBS || njobs * (ns / 32) > MAX_JOBS || njobs > 65535) return hipErrorInvalidValue;
  PlaneCounts pc;
  for (int l = 0; l < MAX_LIMBS; l++) { pc.n[l] = l < nlimbs ? planes[l] : 0; pc.nt[l] = l < nlimbs ? (unsigned)nt[l] : 0; }
  hipLaunchKernelGGL((k_digits_to_v8<true, 4>), dim3((unsigned)((n + D - 1) / D), (unsigned)(ns / 4), (unsigned)njobs), dim3(256), 0, s, src64, n, dst,
                     limb_stride, (unsigned)kt_count, (unsigned)ns, (unsigned)r, nlimbs, tabs, pc, (unsigned)r, 32u, job_stride);
  return hipGetLastError();
}

// ---- rows of a coefficient matrix (64-bit words below q < 2^40, e.g. a chunk of a CRS matrix fresh from k_crs_expand) ->
// reduce -> transform of every limb -> A-operand tiles of a lab_mat8 (3-plane limbs: primes below 2^23), in one pass: the
// streamed outer commitment's slab never exists as transform-domain residues.  Same structure as k_digits_to_v8<true, 4>
// with the roles of rows and columns exchanged: block = 4 waves = 4 matrix rows x 64 consecutive columns; thread = one
// polynomial; a tile lane is (k half, row & 31), so the 4 rows of a block are 4 neighbouring lanes (64-byte runs).
// cols % 64 == 0 and kbase % 32 == 0 (the caller's business): a block's 64 columns are exactly two k tiles.
struct A8Dst { int8_t* d[MAX_LIMBS]; };

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_coefs_to_a8(const uint64_t* __restrict__ coefs, unsigned nrows, unsigned cols, unsigned row_base, unsigned kbase, const A8Dst dst,
              unsigned rt_count, unsigned kt_count, int nlimbs, const LimbTable* __restrict__ tabs) {
  constexpr int NSETS = 4, ROW = DV_ROW<NSETS>, SB = 2;
  constexpr int LDSW = 64 * NSETS * 36 > 16 * ROW ? 64 * NSETS * 36 : 16 * ROW;
  __shared__ __attribute__((aligned(16))) uint32_t ex[LDSW];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const unsigned c0 = blockIdx.x * D, row0 = blockIdx.y * NSETS;
  uint32_t clo[D], chi[D / 4];
  {
    // piece id = it * 256 + t -> polynomial id >> 3 (row poly / 64 of the block, column poly % 64), 16-byte piece id & 7 of the
    // 128 bytes of this pass: eight lanes fetch one full line; parked at row (row * 64 + column) of the tile
    auto fetch = [&](int q, u32x4_t (&v)[8]) {
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const unsigned id = it * 256 + t, poly = id >> 3, piece = id & 7, r = poly >> 6, x = poly & 63;
        const u32x4_t z = {0, 0, 0, 0};
        v[it] = (row0 + r < nrows && c0 + x < cols)
                    ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(coefs + ((size_t)(row0 + r) * cols + c0 + x) * D + q * 16) + piece)
                    : z;
      }
    };
#pragma unroll
    for (int i = 0; i < D / 4; i++) chi[i] = 0;
    u32x4_t cur[8], nxt[8];
    fetch(0, cur);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (q < 3) fetch(q + 1, nxt);
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const unsigned id = it * 256 + t, poly = id >> 3, piece = id & 7;
        *reinterpret_cast<u32x4_t*>(&ex[poly * 36 + piece * 4]) = cur[it];
      }
      __syncthreads();
#pragma unroll
      for (int piece = 0; piece < 8; piece++) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&ex[(w * 64 + lane) * 36 + piece * 4]);
        const int k = q * 16 + piece * 2;
        clo[k] = v.x; clo[k + 1] = v.z;
        chi[k / 4] |= ((v.y & 0xFFu) << (8 * (k & 3))) | ((v.w & 0xFFu) << (8 * ((k + 1) & 3)));
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 8; it++) cur[it] = nxt[it];
    }
  }
  // gather role: row (fastest: neighbouring lanes of the A tile), k block of 16, slot inside the pass
  const unsigned g_set = t & (NSETS - 1), g_kb = (t >> SB) & 3, g_jl = t >> (SB + 2);
  const unsigned g_row = row_base + row0 + g_set;
  const unsigned rt = g_row / 32, ln = ((g_kb & 1) << 5) | (g_row & 31);
  const size_t kt = (size_t)(kbase + c0) / 32 + g_kb / 2;
#pragma unroll 1
  for (int l = 0; l < nlimbs; l++) {
    uint32_t x[D];
    {
      const LimbTable* tab = tabs + l;
      const uint32_t p = tab->p, m = (uint32_t)(4294967296.0 * tab->inv_p) - 1, r1 = tab->r1;
#pragma unroll
      for (int k = 0; k < D; k++) x[k] = reduce_small16((chi[k / 4] >> (8 * (k & 3))) & 0xFFu, clo[k], p, m, r1);
    }
    ntt64_forward(x, tabs + l);
    const uint32_t offs = digit_offsets(3);
    int8_t* dl = dst.d[l];
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int c = 0; c < 16; c++) ex[c * ROW + w * D + lane] = signed_digit_word(x[16 * q + c], offs);
      __syncthreads();
      uint32_t wd[16];
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&ex[g_jl * ROW + g_set * D + g_kb * 16 + 4 * b]);
        wd[4 * b] = v.x; wd[4 * b + 1] = v.y; wd[4 * b + 2] = v.z; wd[4 * b + 3] = v.w;
      }
      const unsigned j = 16 * q + g_jl;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        u32x4_t o;
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
          const uint32_t sel = 0x04000400u + 0x01010101u * i;
          const uint32_t lo = __builtin_amdgcn_perm(wd[4 * qq + 1], wd[4 * qq], sel);
          const uint32_t hi = __builtin_amdgcn_perm(wd[4 * qq + 3], wd[4 * qq + 2], sel);
          o[qq] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        }
        if (kt < kt_count && rt < rt_count)
          *reinterpret_cast<u32x4_t*>(dl + ((((size_t)j * 3 + i) * rt_count + rt) * kt_count + kt) * 1024 + ln * 16) = o;
      }
      __syncthreads();
    }
  }
}

// nrows x cols polynomials (row-major, 64-bit coefficients below q) -> rows row_base.. and k positions kbase.. of the tiles dst[l]
// ([64 slots][3 planes][rt_count][kt_count] KiB per limb).  cols % 64 == 0, kbase % 32 == 0, every limb prime below 2^23.
hipError_t launch_coefs_to_a8(const uint64_t* coefs, size_t nrows, size_t cols, size_t row_base, size_t kbase, int8_t* const* dst,
                              size_t rt_count, size_t kt_count, int nlimbs, const LimbTable* tabs, hipStream_t s) {
  if (nrows == 0 || cols == 0) return hipSuccess;
  if ((cols % 64) || (kbase % 32) || nlimbs > MAX_LIMBS || (nrows + 3) / 4 > 65535) return hipErrorInvalidValue;
  A8Dst d;
  for (int l = 0; l < MAX_LIMBS; l++) d.d[l] = l < nlimbs ? dst[l] : nullptr;
  hipLaunchKernelGGL(k_coefs_to_a8, dim3((unsigned)(cols / 64), (unsigned)((nrows + 3) / 4)), dim3(256), 0, s, coefs, (unsigned)nrows,
                     (unsigned)cols, (unsigned)row_base, (unsigned)kbase, d, (unsigned)rt_count, (unsigned)kt_count, nlimbs, tabs);
  return hipGetLastError();
}

// ---- X = P^T S of one proof (G = S^T S when P is S) from B-operand tiles: the witness columns packed for step 1, and
// the columns of phi' / W packed the same way (3-plane limbs).
// A B-operand tile (lane l: column l & 31, 16 consecutive k of half l >> 5) is byte for byte an A-operand tile with the
// columns as rows, so the tiles of plane i of P and plane i' of S multiply directly:
// D[i][i'][col][col'] = sum_x P_i[x][col] S_i'[x][col'].
// One wave per (proof, NTT slot): 3 (6) tiles per k tile, 9 MFMAs, two k tiles in flight behind the one being multiplied;
// all nine int32 tiles of the wave have the same element in the same register of the same lane, so the recombination
// needs no LDS.  Blocks of one proof share an XCD (the 64 slots of a result line are written by 64 blocks).
// NP = 2: two P operands against the same S tiles in one pass (phi'^T S and W^T S: the S tiles are read once).
template <bool SAME, int NP>
__global__ void __launch_bounds__(64)
k_gram_v8(const int8_t* __restrict__ P8a, const int8_t* __restrict__ P8b, const int8_t* __restrict__ S8, size_t v8_job_stride,
          unsigned kt_count, unsigned kt_real, unsigned nt_count, unsigned sub, unsigned r, uint32_t* __restrict__ outa,
          uint32_t* __restrict__ outb, size_t out_job_stride, uint32_t p, uint32_t pw0, uint32_t pw1, uint32_t pw2, uint32_t pw3,
          uint32_t pw4, uint32_t bias, const LimbTable* __restrict__ tab) {
  static_assert(!SAME || NP == 1, "G = S^T S has one operand");
  const unsigned job = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
  const unsigned g = job / sub, sb = job % sub;
  const size_t off = (size_t)g * v8_job_stride / 16 + (size_t)j * kt_count * nt_count * 64 + lane;
  const u32x4_t* sbase = reinterpret_cast<const u32x4_t*>(S8) + off;
  const u32x4_t* pbase[2] = {reinterpret_cast<const u32x4_t*>(P8a) + off, reinterpret_cast<const u32x4_t*>(NP > 1 ? P8b : P8a) + off};
  i32x16_t acc[NP][3][3];
#pragma unroll
  for (int o = 0; o < NP; o++)
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[o][a][b][e] = 0;
  u32x4_t st[3][3], pt[SAME ? 1 : 3][NP][3];       // [ring slot]([operand])[plane]
  auto load = [&](unsigned kt, int slot) {
#pragma unroll
    for (int pl = 0; pl < 3; pl++) {
      const size_t o = ((size_t)kt * nt_count + pl * sub + sb) * 64;
      st[slot][pl] = sbase[o];
      if constexpr (!SAME) {
#pragma unroll
        for (int op = 0; op < NP; op++) pt[slot][op][pl] = pbase[op][o];
      }
    }
  };
  load(0, 0);
  if (kt_real > 1) load(1, 1);
  for (unsigned kt = 0; kt < kt_real; kt += 3) {
#pragma unroll
    for (int u = 0; u < 3; u++) {
      if (kt + u < kt_real) {
        if (kt + u + 2 < kt_real) load(kt + u + 2, (u + 2) % 3);
#pragma unroll
        for (int op = 0; op < NP; op++)
#pragma unroll
          for (int a = 0; a < 3; a++) {
            const u32x4_t av = SAME ? st[u][a] : pt[SAME ? 0 : u][op][a];
            const i32x4_t af = {(int)av.x, (int)av.y, (int)av.z, (int)av.w};
#pragma unroll
            for (int b = 0; b < 3; b++) {
              const i32x4_t bf = {(int)st[u][b].x, (int)st[u][b].y, (int)st[u][b].z, (int)st[u][b].w};
              acc[op][a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[op][a][b], 0, 0, 0);
            }
          }
      }
    }
  }
  const uint32_t pw[5] = {pw0, pw1, pw2, pw3, pw4};
  const uint32_t p2 = tab->p2, pinv = tab->pinv, r2 = tab->r2;
  const unsigned col = lane & 31;
#pragma unroll
  for (int op = 0; op < NP; op++) {
    uint32_t* o = (op ? outb : outa) + (size_t)job * out_job_stride + j;
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const unsigned row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      uint64_t s64 = 0;                               // 9 products < 2^32 * p: the high word stays below 9p < 2^27
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) s64 += (uint64_t)((uint32_t)acc[op][a][b][e] ^ 0x80000000u) * pw[a + b];
      uint32_t hi = (uint32_t)(s64 >> 32);
      hi = min(hi, hi - 8 * p); hi = min(hi, hi - 4 * p); hi = min(hi, hi - p2); hi = min(hi, hi - p);
      const uint32_t v = mont_mul(redc64(((uint64_t)hi << 32) | (uint32_t)s64, p, pinv), r2, p, pinv);
      if (row < r && col < r) o[((size_t)row * r + col) * D] = csub(v + p - bias, p);   // row: column of P, col: column of S
    }
  }
}

// P8a null (or S8): G = S^T S.  P8b non-null: a second operand, result to out_b.
hipError_t launch_gram_v8(const int8_t* P8a, const int8_t* P8b, const int8_t* S8, size_t v8_job_stride, size_t kt_count, size_t kt_real,
                          size_t nt_count, size_t sub, size_t r, size_t njobs, uint32_t* out_a, uint32_t* out_b, size_t out_job_stride,
                          uint32_t p, const uint32_t* pw, uint32_t bias, const LimbTable* tab, hipStream_t s) {
  if (r > 32 || njobs == 0 || p >= (1u << 23) || (v8_job_stride % 16)) return hipErrorInvalidValue;
  const dim3 g((unsigned)njobs, D);
#define LAB_GRAM_ARGS v8_job_stride, (unsigned)kt_count, (unsigned)kt_real, (unsigned)nt_count, (unsigned)sub, (unsigned)r, out_a, out_b, \
                      out_job_stride, p, pw[0], pw[1], pw[2], pw[3], pw[4], bias, tab
  if (!P8a || P8a == S8) hipLaunchKernelGGL((k_gram_v8<true, 1>), g, dim3(64), 0, s, S8, S8, S8, LAB_GRAM_ARGS);
  else if (!P8b) hipLaunchKernelGGL((k_gram_v8<false, 1>), g, dim3(64), 0, s, P8a, P8a, S8, LAB_GRAM_ARGS);
  else hipLaunchKernelGGL((k_gram_v8<false, 2>), g, dim3(64), 0, s, P8a, P8b, S8, LAB_GRAM_ARGS);
#undef LAB_GRAM_ARGS
  return hipGetLastError();
}

// V_s[k] = src + s*set_stride + k*k_stride (words): k_stride = 64 for vectors stored set by set, r*64 for the columns of
// an [n][r] witness matrix
hipError_t launch_pack_vectors_i8(int ns, int ns_valid, int8_t* dst, const uint32_t* src, size_t set_stride, size_t k_stride,
                                  size_t ktotal, size_t kt_count, size_t nt_count, int planes, size_t njobs, size_t src_job_stride,
                                  size_t dst_job_stride, size_t nsub, size_t src_sub_stride, hipStream_t s) {
  if (ns % PV_SETS || (set_stride % 4) || (k_stride % 4) || (src_job_stride % 4) || njobs * (ns / PV_SETS) > 65535) return hipErrorInvalidValue;
  dim3 g((unsigned)kt_count, 2, (unsigned)(njobs * (ns / PV_SETS)));
  hipLaunchKernelGGL(k_pack_vectors_i8, g, dim3(256), 0, s, dst, src, set_stride, k_stride, (unsigned)ktotal, (unsigned)kt_count,
                     (unsigned)nt_count, planes, (unsigned)ns, (unsigned)ns_valid, src_job_stride, dst_job_stride, (unsigned)nsub,
                     src_sub_stride);
  return hipGetLastError();
}

// ---- the GEMM, recombination fused.  Block = 4 waves = (digit plane i) x (row tile): with 4 planes the block owns
// ONE 32-row tile of one slot j and its four waves hold the four digit planes of the matrix; with 2 planes two row
// tiles.  Each wave multiplies its A plane with all NT column tiles (NT accumulator tiles of 16 registers).  K is
// walked in stages of KS k-tiles.  A operands (KS KiB per wave and stage, streamed once from HBM with non-temporal
// 16-byte loads) sit in a 3-deep register ring: two stages are in flight behind the one being multiplied.  B operands
// (KS*NT KiB per block and stage, shared by the 4 waves) come from L2 through a double-buffered LDS stage.  At the end
// the waves park their int32 tiles in LDS and the block writes out_s[row][j] = sum_{i,i'} 256^(i+i') D_i[row][(i',s)]
// mod p directly: no int32 matrix ever goes to memory.
// Blocks are numbered so that all blocks that walk the same operand streams run on ONE XCD at the same time: XCD
// x = blockIdx.x % 8 takes slots j = x, x+8, ...; within a slot the job index (a "job" = one set of NS vectors with its
// own B operand stream, e.g. one proof's witness columns against the shared matrix A) runs fastest, then the row
// group — so the first block pulls an operand tile from HBM/MALL and its neighbours hit that XCD's L2.
constexpr int G8_KS = 4;


// WAVES = 4 or 8 waves per block: with 8 the block owns two 32-row tiles (of a 4-plane limb) that share every B stage,
// which halves the L2 -> LDS operand traffic per multiply and the B staging registers per thread.
// A limb whose prime is below 2^23 has 3 digit planes: blocks of 3 or 6 waves (one or two row tiles), 3 or 6 column tiles
// for 32 or 64 sets, 9 instead of 16 byte products per modular product.
template <int NT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
k_gemm_i8(const Gemm8Args a) {
  // B staging during the K loop (2 x KS*NT KiB), afterwards the int32 tiles of one wave group at a time (GW x RN*4 KiB)
  constexpr int NTHR = 64 * WAVES;
  constexpr int GW = WAVES % 3 == 0 ? 3 : 4;       // waves that recombine together: the planes of one row tile (4 or 3 planes) or of two (2 planes)
  constexpr int RN = NT > 4 ? NT / 2 : NT;         // column tiles recombined per pass (NT = 8, 6: two passes of 4, 3)
  constexpr int LQ = 2 * G8_KS * NT * 64 > GW * RN * 16 * 64 / 4 ? 2 * G8_KS * NT * 64 : GW * RN * 16 * 64 / 4;
  __shared__ u32x4_t lds[LQ];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;   // LDS byte address
  const unsigned planes = a.planes, rt_count = a.rt_count, kt_count = a.kt_count;
  const unsigned x = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const unsigned per_j = a.rgroups * a.njobs;
  const unsigned j = x + 8 * (seq / per_j), sub = seq % per_j;
  const unsigned job = sub % a.njobs, rg = sub / a.njobs;
  const unsigned tiles_per_block = WAVES / planes; // row tiles per block
  const unsigned i = w % planes, rtl = w / planes, rt = rg * tiles_per_block + rtl;
  const unsigned ji = j * planes + i;
  const u32x4_t* A = reinterpret_cast<const u32x4_t*>(a.M8) + ((size_t)ji * rt_count + min(rt, rt_count - 1)) * kt_count * 64 + lane;
  const u32x4_t* B = reinterpret_cast<const u32x4_t*>(a.V8 + (size_t)job * a.v8_job_stride) + (size_t)j * kt_count * NT * 64;
  constexpr int BQ = G8_KS * NT * 64;              // B quads per stage
  constexpr int BPT = (BQ + NTHR - 1) / NTHR;      // per thread (the last one guarded when 64*WAVES does not divide BQ)
  constexpr bool BEVEN = BQ % NTHR == 0;

  i32x16_t acc[NT];
#pragma unroll
  for (int n = 0; n < NT; n++)
#pragma unroll
    for (int g = 0; g < 16; g++) acc[n][g] = 0;

  const unsigned nstages = kt_count / G8_KS;       // kt_count is padded to a multiple of 3 * G8_KS
  const bool reuse_a = a.njobs > 1;
  u32x4_t a0[G8_KS], a1[G8_KS], a2[G8_KS];
  // B operands on their way to LDS: up to 4 column tiles a ring of 3 stages like A (two stages in flight: sums as
  // short as K = 1024 are 8 stages, there is no steady state to hide a one-stage prefetch behind); with 8 column
  // tiles (long sums only) one stage, for the registers
  constexpr bool BRING = BPT <= 4;
  u32x4_t b0[BPT], b1[BRING ? BPT : 1], b2[BRING ? BPT : 1];
  auto loadA = [&](unsigned st, u32x4_t (&dst)[G8_KS]) {
#pragma unroll
    for (int s = 0; s < G8_KS; s++)   // read once (outer commitment): non-temporal; shared by many jobs (Ajtai): keep it in L2
      dst[s] = reuse_a ? A[((size_t)st * G8_KS + s) * 64] : __builtin_nontemporal_load(A + ((size_t)st * G8_KS + s) * 64);
  };
  auto loadB = [&](unsigned st, u32x4_t (&dst)[BPT]) {
#pragma unroll
    for (int q = 0; q < BPT; q++)
      if (BEVEN || t + q * NTHR < BQ) dst[q] = B[(size_t)st * BQ + t + q * NTHR];
  };
  auto parkB = [&](int buf, const u32x4_t (&src)[BPT]) {
#pragma unroll
    for (int q = 0; q < BPT; q++)
      if (BEVEN || t + q * NTHR < BQ) lds[buf * BQ + t + q * NTHR] = src[q];
  };
  // one stage: multiply stage st (A operands in `cur`, B operands in LDS buffer st & 1) while later stages' operands
  // are fetched: A of stage st+2 into `nxt` (the ring slot stage st-1 used); B of stage st+2 into `bld` with the ring
  // (st+1 without), and at the end of the stage the B operands of stage st+1 (`bpk`) are parked in the other LDS buffer
  auto step = [&](unsigned st, u32x4_t (&cur)[G8_KS], u32x4_t (&nxt)[G8_KS], auto& bld, auto& bpk) {
    const int buf = st & 1;
    if constexpr (BRING) {
      if (st + 2 < nstages) { loadB(st + 2, bld); loadA(st + 2, nxt); }
    } else {
      if (st + 1 < nstages) loadB(st + 1, bld);    // issued before the A loads: waiting for it leaves those in flight
      if (st + 2 < nstages) loadA(st + 2, nxt);
    }
    // B fragments are read from LDS PF multiplies ahead of their use (a 32x32x32 MFMA issues in 32 cycles, an LDS
    // read takes ~100): fragment i of the stage is tile (k-step i / NT, column tile i % NT).  The reads are inline
    // asm with counted waits: left to the scheduler they are sunk next to their use and every other MFMA waits
    // a full LDS latency.
    constexpr int NM = G8_KS * NT, PF = 4;
    const uint32_t fbase = lds_base + (uint32_t)buf * (BQ * 16) + lane * 16;
    u32x4_t fr[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[i]) : "v"(fbase), "n"(i * 1024));
#pragma unroll
    for (int i = 0; i < NM; i++) {
      const int ks = i / NT, n = i % NT;
      const i32x4_t af = {(int)cur[ks].x, (int)cur[ks].y, (int)cur[ks].z, (int)cur[ks].w};
      // fragment i has landed once at most the reads issued after it are outstanding
      if (NM - 1 - i >= PF - 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[i % PF]) : "n"(PF - 1));
      else if (NM - 1 - i == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fr[i % PF]));
      else if (NM - 1 - i == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fr[i % PF]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[i % PF]));
      const u32x4_t bq = fr[i % PF];
      const i32x4_t bf = {(int)bq.x, (int)bq.y, (int)bq.z, (int)bq.w};
      acc[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[n], 0, 0, 0);
      if (i + PF < NM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[i % PF]) : "v"(fbase), "n"((i + PF) * 1024));
    }
    if (st + 1 < nstages) parkB(buf ^ 1, bpk);
    __syncthreads();
  };
  loadB(0, b0);
  loadA(0, a0);
  if constexpr (BRING) {
    if (nstages > 1) { loadB(1, b1); loadA(1, a1); }
    parkB(0, b0);
    __syncthreads();
    for (unsigned st = 0; st < nstages; st += 3) {   // nstages is a multiple of 3 (kt_count is padded to 12 k-tiles)
      step(st, a0, a2, b2, b1);       // fetch B(st+2) -> b2, park B(st+1) = b1
      step(st + 1, a1, a0, b0, b2);
      step(st + 2, a2, a1, b1, b0);
    }
  } else {
    if (nstages > 1) loadA(1, a1);
    parkB(0, b0);
    __syncthreads();
    for (unsigned st = 0; st < nstages; st += 3) {
      step(st, a0, a2, b0, b0);       // fetch B(st+1) -> b0, park it at the end of the same stage
      step(st + 1, a1, a0, b0, b0);
      step(st + 2, a2, a1, b0, b0);
    }
  }
  // ---- recombination.  Wave w parks column tile n (compact index nc), register g, lane l at
  // dl[((w*RN + nc)*16 + g)*64 + l] (register-major: the readers' 32 consecutive columns hit 32 different banks); element (row, col) of a 32x32 tile sits in register
  // g = (row & 3) | ((row >> 3) << 2) of lane ((row >> 2) & 1) * 32 + col.  With 8 column tiles (64 sets x 4 planes:
  // column n = i'*64 + s lies in tile 2i' + s/32) the sets are recombined in two passes of 32, each parking the four
  // tiles of its half.
  int32_t* dl = reinterpret_cast<int32_t*>(lds);
  constexpr int HALVES = NT / RN;
  constexpr int WG = WAVES / GW;                    // the wave groups take turns in the buffer
  const unsigned tiles_per_pass = GW / planes;      // row tiles whose planes sit in one wave group
  bool first_pass = true;
#pragma unroll
  for (int wg = 0; wg < WG; wg++)
#pragma unroll
  for (int half = 0; half < HALVES; half++) {
  if (!first_pass) __syncthreads();               // everybody finished reading the previous pass
  first_pass = false;
  if (w / GW == wg) {
#pragma unroll
    for (int n = 0; n < NT; n++)
      if (n % HALVES == half) {
#pragma unroll
        for (int g = 0; g < 16; g++) dl[(((w % GW) * RN + n / HALVES) * 16 + g) * 64 + lane] = acc[n][g];
      }
  }
  __syncthreads();
  // out = sum_{i,i'} 256^(i+i') D_i[row][(i',s)] mod p.  For the 30-bit limb primes everything stays in 32-bit
  // conditional subtractions and one Montgomery reduction: |D| <= K * 2^14 < 2^31 - 766 (K < 2^17 - 1 is enforced at pack
  // time), so D + 2p lies in (0, 4p) and fits 32 bits; c_t = sum over i+i'=t of at most 4 canonical residues < 4p.
  // A small prime (the 2-plane limb of the digit basis) takes a division-free double-reciprocal path.
  const uint32_t p = a.p;
  const LimbTable* tab = a.tab;
  const uint32_t p2 = tab->p2, pinv = tab->pinv, r2 = tab->r2;
  const bool big = GW != 3 && p > (1u << 29);
  uint32_t pw[7];                                 // 256^e mod p, from the host
#pragma unroll
  for (int e = 0; e < 7; e++) pw[e] = a.pw[e];
  const double inv_p = tab->inv_p;
  const unsigned ns = a.ns, nsh = ns / HALVES;     // sets per pass
  uint32_t* out = a.out + (size_t)job * a.out_job_stride;
  for (unsigned o = t; o < tiles_per_pass * 32 * nsh; o += NTHR) {
    const unsigned s = half * nsh + o % nsh, row = (o / nsh) % 32, tl = o / (nsh * 32);   // tl: row tile inside this pass
    const unsigned grow = (rg * tiles_per_block + wg * tiles_per_pass + tl) * 32 + row;
    if (s % a.nsub >= a.ns_valid || grow >= a.rows) continue;
    const unsigned g = (row & 3) | ((row >> 3) << 2), lh = ((row >> 2) & 1) * 32;
    uint32_t res;
    if (big) {
      // every int32 D enters as the unsigned D + 2^31 (one xor), so the whole sum is unsigned multiply-adds with the
      // weights 256^(i+i') mod p in scalar registers; the surplus 2^31 * sum 256^(i+i') mod p (a.bias, from the host) is
      // taken off at the end.  A product is < 2^32 * p: it adds less than p (+ a carry) to the high word, which a lazy
      // fold keeps below 2p: three products after the start, two after every fold stay below 4p < 2^32.
      uint64_t acc = 0;                             // a 30-bit prime always has 4 digit planes (mfma_planes)
#pragma unroll
      for (unsigned pi = 0; pi < 4; pi++)
#pragma unroll
        for (unsigned pv = 0; pv < 4; pv++) {
          const unsigned n = pv * ns + s, seq = pi * 4 + pv;
          const int32_t d = dl[(((tl * 4 + pi) * RN + (n / 32) / HALVES) * 16 + g) * 64 + lh + (n % 32)];
          acc += (uint64_t)((uint32_t)d ^ 0x80000000u) * pw[pi + pv];
          if (seq >= 2 && seq % 2 == 0) acc = acc_fold_lazy(acc, p2);
        }
      const uint32_t v = mont_mul(redc64(acc_fold(acc, p, p2), p, pinv), r2, p, pinv);
      res = csub(v + p - a.bias, p);
    } else if (GW == 3) {
      // a prime in [2^15, 2^23) with 3 planes: the same unsigned sum, 9 products < 2^32 * p each, no folds on the way
      // (the high word stays below 9p < 2^27); it is brought below p by conditional subtractions of 8p, 4p, 2p, p
      uint64_t acc = 0;
#pragma unroll
      for (unsigned pi = 0; pi < 3; pi++)
#pragma unroll
        for (unsigned pv = 0; pv < 3; pv++) {
          const unsigned n = pv * ns + s;
          const int32_t d = dl[(((tl * 3 + pi) * RN + (n / 32) / HALVES) * 16 + g) * 64 + lh + (n % 32)];
          acc += (uint64_t)((uint32_t)d ^ 0x80000000u) * pw[pi + pv];
        }
      uint32_t hi = (uint32_t)(acc >> 32);
      hi = min(hi, hi - 8 * p); hi = min(hi, hi - 4 * p); hi = min(hi, hi - p2); hi = min(hi, hi - p);
      const uint32_t v = mont_mul(redc64(((uint64_t)hi << 32) | (uint32_t)acc, p, pinv), r2, p, pinv);
      res = csub(v + p - a.bias, p);
    } else {
      auto modp = [&](int64_t v) -> int64_t {      // |v| < 2^40: floor(v / p) from one double multiply is off by at most 1
        const int64_t qq = (int64_t)__builtin_floor((double)v * inv_p);
        int64_t r = v - qq * (int64_t)p;
        r += r < 0 ? (int64_t)p : 0;
        r -= r >= (int64_t)p ? (int64_t)p : 0;
        return r;
      };
      int64_t ct[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (unsigned pi = 0; pi < 4; pi++)
#pragma unroll
        for (unsigned pv = 0; pv < 4; pv++) {
          if (pi < planes && pv < planes) {
            const unsigned n = pv * ns + s;
            ct[pi + pv] += dl[(((tl * planes + pi) * RN + (n / 32) / HALVES) * 16 + g) * 64 + lh + (n % 32)];
          }
        }
      int64_t r = 0;
#pragma unroll
      for (int e = 6; e >= 0; e--) r = modp(r * 256 + modp(ct[e]));
      res = (uint32_t)r;
    }
    // A result word belongs to a 256-byte ring element whose other 63 slots come from 63 other blocks.  With few blocks
    // (the outer commitment) the scattered 4-byte stores are harmless; with many (the Ajtai product of a batch: 6.7e7
    // stores per launch) every one of them makes L2 fetch a line to merge into, so results go to a slot-major
    // staging array with full-line stores and k_slot_transpose moves them into place.
    if (a.stage) a.stage[(((size_t)job * D + j) * (rt_count * 32) + grow) * ns + s] = res;
    else out[(size_t)s * a.out_set_stride + (size_t)grow * D + j] = res;
  }
  }  // (wave group, half)
}

// ---- the GEMM for 3-plane limbs, one row tile per wave (the prover's step 1, 3, 4, 9 launches in the matrix-core digit basis).
// k_gemm_i8 above gives every wave ONE (plane, row tile) A operand and lets it re-read the whole B stage from LDS: one
// LDS fragment per MFMA, 6 waves on 4 SIMDs (two SIMDs carry two waves, two carry one), a block-wide barrier every 4
// k-tiles with the B operands staged through VGPRs by the waves that issue the MFMAs — and, as compiled, a drain of the
// vector-memory queue (s_waitcnt vmcnt(0)) inside every stage: the compiler's wait-count pass does not follow loads that
// are consumed several loop trips after they were issued, so the HBM stream and the matrix instructions take turns.
// Here a wave owns all THREE digit planes of one 32-row tile against all NT column tiles: 3*NT accumulator tiles (288
// registers at 64 sets — one wave per SIMD, 512 registers each), so
//   * every B fragment read from LDS feeds 3 MFMAs (a third of the LDS traffic per multiply),
//   * a block is 4 waves = 4 row tiles = one wave per SIMD: the four SIMDs carry equal MFMA load, and one B stage from
//     L2 serves 128 matrix rows instead of 64,
//   * B stages arrive by LDS-DMA (global_load_lds_dwordx4) into a 4-slot ring, three stages ahead: no staging registers,
//     no ds_write, nothing for the MFMA waves to park,
//   * the barrier at the top of stage st certifies the B operands of stage st+1 (each wave has waited for its own pieces
//     of that stage), so the first fragments of the next stage are read before the barrier and the MFMA stream does
//     not drain across it,
//   * A operands (3 KiB per wave and k-tile, straight from HBM, non-temporal when read once) sit in a 4-slot register
//     ring, three stages (6 k-tiles, 18 KiB per wave, 72 KiB per CU) in flight,
//   * the nine int32 tiles a result needs are in the same register of the same lane of ONE wave: the recombination
//     sum_{i,i'} 256^(i+i') D[i][i'] mod p runs in registers, no LDS parking, no block barrier.
// The K loop is assembly (gemm_w3_asm.inc, generated by tools/gen_gemm_w3_asm.py): every wait is counted by hand, and the
// accumulator tiles beyond the 256 accumulation registers sit in vector registers (the compiler assigns one register class
// to the results of all matrix instructions of a kernel).  Stage = 2 k-tiles; kt_count is a multiple of 12, so the stage
// count is even and the 4-stage loop body may end after its second stage.
#include "gemm_w3_asm.inc"
constexpr int W3_KS = 2;       // k-tiles per stage
constexpr int W3_RING = 4;     // ring slots (A registers and B LDS slots): 3 stages in flight behind the one being multiplied

// DIAG != 0 (lab_ctx_gemm_diag, NT = 6 only): every block also records the shader-clock and constant-clock ticks its K loop
// took; DIAG 2 / 3 run the loop without its loads / without its matrix instructions (the results are then meaningless).
template <int NT, int DIAG = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_gemm_i8_w3(const Gemm8Args a) {
  static_assert(NT == 3 || NT == 6, "32 or 64 sets of 3 planes");
  static_assert(DIAG == 0 || NT == 6, "the diagnostic variants exist for the outer commitment's shape");
  constexpr int HALVES = NT / 3;                    // 32-set halves: column n = i' * ns + s lies in tile i' * HALVES + s / 32
  constexpr int PIECES = W3_KS * NT;                // 1 KiB DMA pieces per stage
  constexpr int PPW = (PIECES + 3) / 4;             // ... per wave (NT = 3: the last wave repeats the third wave's pieces)
  __shared__ u32x4_t bring[W3_RING * PIECES * 64];  // B ring: slot r at r * PIECES KiB
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const unsigned rt_count = a.rt_count, kt_count = a.kt_count;
  const unsigned x = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const unsigned per_j = a.rgroups * a.njobs;
  const unsigned j = x + 8 * (seq / per_j), sub = seq % per_j;
  const unsigned job = sub % a.njobs, rg = sub / a.njobs;
  const unsigned rt = rg * 4 + w;
  const unsigned rtc = min(rt, rt_count - 1);       // a wave past the last row tile multiplies a copy of it and stores nothing
  const int first_piece = min(w * PPW, PIECES - PPW);
  uint64_t a0 = (uint64_t)(uintptr_t)a.M8 + (((size_t)j * 3 + 0) * rt_count + rtc) * kt_count * 1024;
  uint64_t a1 = (uint64_t)(uintptr_t)a.M8 + (((size_t)j * 3 + 1) * rt_count + rtc) * kt_count * 1024;
  uint64_t a2 = (uint64_t)(uintptr_t)a.M8 + (((size_t)j * 3 + 2) * rt_count + rtc) * kt_count * 1024;
  uint64_t bg = (uint64_t)(uintptr_t)a.V8 + (size_t)job * a.v8_job_stride + (size_t)j * kt_count * NT * 1024 + first_piece * 1024 + lane * 16;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)bring;
  const uint32_t voff = lane * 16, vlds = lds0 + lane * 16;
  const uint32_t m0b = __builtin_amdgcn_readfirstlane(lds0 + first_piece * 1024);
  // stages that hold terms, rounded up to an even count (the loop body is two stages; kt_count is a multiple of 4, so
  // the extra stage is zero padding inside the allocation)
  const uint32_t nstages = __builtin_amdgcn_readfirstlane(min(kt_count / W3_KS, 2 * ((a.kt_real + 2 * W3_KS - 1) / (2 * W3_KS))));

  // accumulator tile (plane i, column tile n) = acc[i * NT + n]; the first 16 in accumulation registers
  i32x16_t acc[3 * NT];
#pragma unroll
  for (int k = 0; k < 3 * NT; k++)
#pragma unroll
    for (int g = 0; g < 16; g++) acc[k][g] = 0;
  if constexpr (NT == 6) {
#define W3_ACC_OPERANDS                                                                                                          \
  "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]), "+a"(acc[8]),  \
      "+a"(acc[9]), "+a"(acc[10]), "+a"(acc[11]), "+a"(acc[12]), "+a"(acc[13]), "+a"(acc[14]), "+a"(acc[15]), "+v"(acc[16]),     \
      "+v"(acc[17]), "+v"(bg)
    uint64_t tick0 = 0, real0 = 0;
    if constexpr (DIAG != 0) { tick0 = __builtin_amdgcn_s_memtime(); real0 = __builtin_amdgcn_s_memrealtime(); }
    if constexpr (DIAG == 2)
      asm volatile(W3_LOOP_ASM_NT6_NOLOAD : W3_ACC_OPERANDS : "s"(a0), "s"(a1), "s"(a2), "v"(voff), "v"(vlds), "s"(m0b), "s"(nstages)
                   : W3_LOOP_CLOBBERS_NT6_NTLOAD);
    else if constexpr (DIAG == 3)
      asm volatile(W3_LOOP_ASM_NT6_NOMFMA : W3_ACC_OPERANDS : "s"(a0), "s"(a1), "s"(a2), "v"(voff), "v"(vlds), "s"(m0b), "s"(nstages)
                   : W3_LOOP_CLOBBERS_NT6_NTLOAD);
    else
    if (a.njobs > 1)   // the matrix is shared by many jobs (Ajtai): keep it in L2
      asm volatile(W3_LOOP_ASM_NT6_PLAIN : W3_ACC_OPERANDS : "s"(a0), "s"(a1), "s"(a2), "v"(voff), "v"(vlds), "s"(m0b), "s"(nstages)
                   : W3_LOOP_CLOBBERS_NT6_PLAIN);
    else               // read once (outer commitment): non-temporal
      asm volatile(W3_LOOP_ASM_NT6_NTLOAD : W3_ACC_OPERANDS : "s"(a0), "s"(a1), "s"(a2), "v"(voff), "v"(vlds), "s"(m0b), "s"(nstages)
                   : W3_LOOP_CLOBBERS_NT6_NTLOAD);
    if constexpr (DIAG != 0) {
      const uint64_t tick1 = __builtin_amdgcn_s_memtime(), real1 = __builtin_amdgcn_s_memrealtime();
      if (t == 0 && a.diag_buf && 2 * (size_t)blockIdx.x + 1 < a.diag_words) {
        a.diag_buf[2 * blockIdx.x] = tick1 - tick0;
        a.diag_buf[2 * blockIdx.x + 1] = real1 - real0;
      }
    }
#undef W3_ACC_OPERANDS
  } else {
#define W3_ACC_OPERANDS                                                                                                          \
  "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]), "+a"(acc[8]),  \
      "+v"(bg)
    if (a.njobs > 1)
      asm volatile(W3_LOOP_ASM_NT3_PLAIN : W3_ACC_OPERANDS : "s"(a0), "s"(a1), "s"(a2), "v"(voff), "v"(vlds), "s"(m0b), "s"(nstages)
                   : W3_LOOP_CLOBBERS_NT3_PLAIN);
    else
      asm volatile(W3_LOOP_ASM_NT3_NTLOAD : W3_ACC_OPERANDS : "s"(a0), "s"(a1), "s"(a2), "v"(voff), "v"(vlds), "s"(m0b), "s"(nstages)
                   : W3_LOOP_CLOBBERS_NT3_NTLOAD);
#undef W3_ACC_OPERANDS
  }

  // ---- recombination in registers: out_s[row] = sum_{i,i'} 256^(i+i') D[i][(i', s)] mod p, every D as the unsigned D + 2^31
  if (rt >= rt_count) return;
  const uint32_t p = a.p;
  const LimbTable* tab = a.tab;
  const uint32_t p2 = tab->p2, pinv = tab->pinv, r2 = tab->r2;
  uint32_t pw[5];
#pragma unroll
  for (int e = 0; e < 5; e++) pw[e] = a.pw[e];
  const unsigned ns = a.ns;
  uint32_t* out = a.out + (size_t)job * a.out_job_stride;
#pragma unroll
  for (int h = 0; h < HALVES; h++) {
    const unsigned s = h * 32 + (lane & 31);
    const bool s_ok = s % a.nsub < a.ns_valid;
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const unsigned row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), grow = rt * 32 + row;
      uint64_t s64 = 0;                             // 9 products < 2^32 * p: the high word stays below 9p < 2^27
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int iv = 0; iv < 3; iv++) s64 += (uint64_t)((uint32_t)acc[i * NT + iv * HALVES + h][e] ^ 0x80000000u) * pw[i + iv];
      uint32_t hi = (uint32_t)(s64 >> 32);
      hi = min(hi, hi - 8 * p); hi = min(hi, hi - 4 * p); hi = min(hi, hi - p2); hi = min(hi, hi - p);
      const uint32_t v = mont_mul(redc64(((uint64_t)hi << 32) | (uint32_t)s64, p, pinv), r2, p, pinv);
      const uint32_t res = csub(v + p - a.bias, p);
      if (s_ok && grow < a.rows) {
        if (a.stage) a.stage[(((size_t)job * D + j) * (rt_count * 32) + grow) * ns + s] = res;
        else out[(size_t)s * a.out_set_stride + (size_t)grow * D + j] = res;
      }
    }
  }
}

// recombination of one job's accumulator tiles (see k_gemm_i8_w3p) into the slot-major staging array (the kernel is only
// launched for many jobs, which always go through it).  SMALL: sums of at most 2^15 terms over a 23-bit prime — the products of
// equal weight were added up by the matrix instructions themselves (10 tiles: weight w, half h), the five weights carry the
// Montgomery factor (pwm[e] = 256^e * 2^32 mod p), so ONE reduction of the lifted 64-bit sum gives the residue.
template <int NT, bool SMALL>
__device__ __forceinline__ void w3p_recombine(const i32x16_t (&acc)[SMALL ? 5 * (NT / 3) : 3 * NT], const Gemm8Args& a, unsigned job, unsigned j, unsigned rt, int lane,
                                              uint32_t p, uint32_t p2, uint32_t pinv, uint32_t r2, const uint32_t (&pw)[5], uint64_t lift) {
  constexpr int HALVES = NT / 3;
  const unsigned ns = a.ns, rt_count = a.rt_count;
  const bool whole = rt * 32 + 32 <= a.rows;        // every row of this tile exists (uniform)
  uint32_t* st = a.stage + (((size_t)job * D + j) * (rt_count * 32) + rt * 32) * ns;
#pragma unroll
  for (int h = 0; h < HALVES; h++) {
    const unsigned s = h * 32 + (lane & 31);
    const bool s_ok = s % a.nsub < a.ns_valid;
    uint32_t res[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      if constexpr (SMALL) {
        // tile (w, h) already holds the sum of the products of weight 256^w (the matrix instructions added them up)
        int64_t s64 = (int64_t)lift;
#pragma unroll
        for (int w = 0; w < 5; w++) s64 += (int64_t)acc[w * HALVES + h][e] * (int32_t)pw[w];
        // 0 <= s64 < 2^57: the high word is below 2^25 < 8p
        uint32_t hi = (uint32_t)((uint64_t)s64 >> 32);
        hi = min(hi, hi - 4 * p); hi = min(hi, hi - p2); hi = min(hi, hi - p);
        res[e] = redc64(((uint64_t)hi << 32) | (uint32_t)s64, p, pinv);
      } else {
        uint64_t s64 = 0;                             // 9 products < 2^32 * p: the high word stays below 9p
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int iv = 0; iv < 3; iv++) s64 += (uint64_t)((uint32_t)acc[SMALL ? 0 : i * NT + iv * HALVES + h][e] ^ 0x80000000u) * pw[i + iv];
        uint32_t hi = (uint32_t)(s64 >> 32);
        hi = min(hi, hi - 8 * p); hi = min(hi, hi - 4 * p); hi = min(hi, hi - p2); hi = min(hi, hi - p);
        const uint32_t v = mont_mul(redc64(((uint64_t)hi << 32) | (uint32_t)s64, p, pinv), r2, p, pinv);
        res[e] = csub(v + p - a.bias, p);
      }
    }
    if (!s_ok) continue;
    if (whole) {
#pragma unroll
      for (int e = 0; e < 16; e++) st[(size_t)((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * ns + s] = res[e];
    } else {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const unsigned row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (rt * 32 + row < a.rows) st[(size_t)row * ns + s] = res[e];
      }
    }
  }
}

// ---- the same GEMM for MANY jobs that share the matrix (the Ajtai product t = A S of a batch: 32 two-proof jobs of 64 sets
// against a kappa x n matrix whose K loop is only n / 32 = 32 k-tiles long).  In k_gemm_i8_w3 every (job, 128 rows, slot) is its
// own block: 11.9 us of matrix instructions sit between 3.9 us of block start-up (dispatch, argument loads, 288 accumulator
// moves, the first loads' round trip) and 6 us of recombination, with one wave per SIMD and nothing to overlap them — the
// matrix pipe is 54 % busy (in-kernel stamps; tools/ubench_mfma_i8.hip puts the ceiling of this operand
// traffic at 2.6 POP/s against 1.77).  Here a block is PERSISTENT over `jobs_per_block` jobs of its (slot, row group):
//   * the B stages 0..2 of job b+1 are on their way (LDS-DMA: no register is in flight while compiler code runs) before the
//     results of job b are recombined, the A operands (L2 hits: the matrix is shared) are issued at the top of its loop;
//   * the first k-tile's matrix instructions take 0 as their C operand: the accumulators are never zeroed;
//   * dispatch, argument loads and address set-up are paid once per block;
//   * sums of at most 2^15 terms (|D| <= K * 2^14 < 2^29: three of them add up in 32 bits): the products (plane i, plane i') of
//     equal weight 256^(i+i') accumulate into ONE tile (the loop variant `merged` of the generator: 10 accumulator tiles
//     instead of 18, all in accumulation registers), and five signed multiply-adds recombine them instead of 9 unsigned ones.
// DIAG (lab_ctx_gemm_diag mode 4): every block records shader / constant clock ticks of its whole life and of its K loops alone
// in diag_buf[4 * block .. + 3]
template <int NT, bool SMALL, bool DIAG = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_gemm_i8_w3p(const Gemm8Args a) {
  uint64_t d_t0 = 0, d_r0 = 0, d_loop_t = 0, d_loop_r = 0;
  if constexpr (DIAG) { d_t0 = __builtin_amdgcn_s_memtime(); d_r0 = __builtin_amdgcn_s_memrealtime(); }
  static_assert(NT == 6, "64 sets of 3 planes");
  constexpr int PIECES = W3_KS * NT;
  constexpr int PPW = (PIECES + 3) / 4;
  __shared__ u32x4_t bring[W3_RING * PIECES * 64];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const unsigned rt_count = a.rt_count, kt_count = a.kt_count;
  const unsigned jpb = a.jobs_per_block, nchunks = (a.njobs + jpb - 1) / jpb;
  const unsigned x = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const unsigned per_j = a.rgroups * nchunks;
  const unsigned j = x + 8 * (seq / per_j), sub = seq % per_j;
  const unsigned chunk = sub % nchunks, rg = sub / nchunks;
  const unsigned rt = rg * 4 + w;
  const unsigned rtc = min(rt, rt_count - 1);
  const int first_piece = min(w * PPW, PIECES - PPW);
  const uint64_t a0 = (uint64_t)(uintptr_t)a.M8 + (((size_t)j * 3 + 0) * rt_count + rtc) * kt_count * 1024;
  const uint64_t a1 = (uint64_t)(uintptr_t)a.M8 + (((size_t)j * 3 + 1) * rt_count + rtc) * kt_count * 1024;
  const uint64_t a2 = (uint64_t)(uintptr_t)a.M8 + (((size_t)j * 3 + 2) * rt_count + rtc) * kt_count * 1024;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)bring;
  const uint32_t voff = lane * 16, vlds = lds0 + lane * 16;
  const uint32_t m0b = __builtin_amdgcn_readfirstlane(lds0 + first_piece * 1024);
  const uint32_t nstages = __builtin_amdgcn_readfirstlane(min(kt_count / W3_KS, 2 * ((a.kt_real + 2 * W3_KS - 1) / (2 * W3_KS))));
  const unsigned job0 = chunk * jpb, job1 = min(a.njobs, job0 + jpb);
  auto bg_of = [&](unsigned job) {
    return (uint64_t)(uintptr_t)a.V8 + (size_t)job * a.v8_job_stride + (size_t)j * kt_count * NT * 1024 + first_piece * 1024 + lane * 16;
  };
  const uint32_t p = a.p;
  const LimbTable* tab = a.tab;
  const uint32_t p2 = tab->p2, pinv = tab->pinv, r2 = tab->r2;
  uint32_t pw[5];
#pragma unroll
  for (int e = 0; e < 5; e++) pw[e] = SMALL ? mont_mul(a.pw[e], tab->r2, p, pinv) : a.pw[e];   // SMALL: 256^e * 2^32 mod p
  // SMALL (chosen by the launcher): every tile entry is below K * 2^14 with 3 * K * 2^14 < 2^31, so three of them add up in 32
  // bits and the weighted sum of the nine tiles stays below 3 * 2^54 in absolute value; `lift` (a multiple of p above that)
  // makes it non-negative, and the high word of the lifted sum is below 2^25 < 8p for the 23-bit primes this path is taken for
  const uint64_t lift = (((1ull << 56) / p) + 1) * (uint64_t)p;

  uint64_t bg = bg_of(job0);
  asm volatile(W3P_BISSUE_ASM_NT6 : "+v"(bg) : "s"(m0b), "s"(nstages) : W3P_BISSUE_CLOBBERS);
  for (unsigned job = job0; job < job1; job++) {
    // SMALL: the products of equal weight add up inside the matrix instructions — tile (i + i', h) instead of (i, i', h), 10 instead of 18
    i32x16_t acc[SMALL ? 5 * (NT / 3) : 3 * NT];
    uint64_t lt0 = 0, lr0 = 0;
    if constexpr (DIAG) { lt0 = __builtin_amdgcn_s_memtime(); lr0 = __builtin_amdgcn_s_memrealtime(); }
    if constexpr (SMALL)
      asm volatile(W3PM_LOOP_ASM_NT6
                   : "=&a"(acc[0]), "=&a"(acc[1]), "=&a"(acc[2]), "=&a"(acc[3]), "=&a"(acc[4]), "=&a"(acc[5]), "=&a"(acc[6]), "=&a"(acc[7]),
                     "=&a"(acc[8]), "=&a"(acc[9]), "+v"(bg)
                   : "s"(a0), "s"(a1), "s"(a2), "v"(voff), "v"(vlds), "s"(m0b), "s"(nstages)
                   : W3_LOOP_CLOBBERS_NT6_PLAIN);
    else
    asm volatile(W3P_LOOP_ASM_NT6
                 : "=&a"(acc[0]), "=&a"(acc[1]), "=&a"(acc[2]), "=&a"(acc[3]), "=&a"(acc[4]), "=&a"(acc[5]), "=&a"(acc[6]), "=&a"(acc[7]),
                   "=&a"(acc[8]), "=&a"(acc[9]), "=&a"(acc[SMALL ? 0 : 10]), "=&a"(acc[SMALL ? 0 : 11]), "=&a"(acc[SMALL ? 0 : 12]),
                   "=&a"(acc[SMALL ? 0 : 13]), "=&a"(acc[SMALL ? 0 : 14]), "=&a"(acc[SMALL ? 0 : 15]), "=&v"(acc[SMALL ? 0 : 16]),
                   "=&v"(acc[SMALL ? 0 : 17]), "+v"(bg)
                 : "s"(a0), "s"(a1), "s"(a2), "v"(voff), "v"(vlds), "s"(m0b), "s"(nstages)
                 : W3_LOOP_CLOBBERS_NT6_PLAIN);
    if constexpr (DIAG) { d_loop_t += __builtin_amdgcn_s_memtime() - lt0; d_loop_r += __builtin_amdgcn_s_memrealtime() - lr0; }
    if (job + 1 < job1) {   // the next job's first B stages travel while this job's results are recombined
      bg = bg_of(job + 1);
      asm volatile(W3P_BISSUE_ASM_NT6 : "+v"(bg) : "s"(m0b), "s"(nstages) : W3P_BISSUE_CLOBBERS);
    }
    if (rt >= rt_count) continue;
    w3p_recombine<NT, SMALL>(acc, a, job, j, rt, lane, p, p2, pinv, r2, pw, lift);
  }
  if constexpr (DIAG) {
    if (t == 0 && a.diag_buf && 4 * (size_t)blockIdx.x + 3 < a.diag_words) {
      a.diag_buf[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - d_t0;
      a.diag_buf[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - d_r0;
      a.diag_buf[4 * blockIdx.x + 2] = d_loop_t;
      a.diag_buf[4 * blockIdx.x + 3] = d_loop_r;
    }
  }
}

// stage[job][j][row][s] -> out[job*out_job_stride + s*out_set_stride + row*64 + j]; block = (row, job)
__global__ void __launch_bounds__(256)
k_slot_transpose(uint32_t* __restrict__ out, size_t out_set_stride, size_t out_job_stride, const uint32_t* __restrict__ stage,
                 unsigned rows_pad, unsigned ns, unsigned ns_valid, unsigned nsub) {
  __shared__ uint32_t tile[D][65];
  const unsigned row = blockIdx.x, job = blockIdx.y;
  for (unsigned e = threadIdx.x; e < D * ns; e += 256) {
    const unsigned j = e / ns, s = e % ns;
    tile[j][s] = stage[(((size_t)job * D + j) * rows_pad + row) * ns + s];
  }
  __syncthreads();
  const unsigned sub = ns / nsub;                 // source jobs per GEMM job: their outputs are out_job_stride apart
  for (unsigned e = threadIdx.x; e < D * ns; e += 256) {
    const unsigned s = e / D, j = e % D;
    if (s % nsub < ns_valid)
      out[((size_t)job * sub + s / nsub) * out_job_stride + (size_t)(s % nsub) * out_set_stride + (size_t)row * D + j] = tile[j][s];
  }
}

hipError_t launch_slot_transpose(uint32_t* out_plane, size_t out_set_stride, size_t out_job_stride, const uint32_t* stage,
                                 size_t rows, size_t rows_pad, size_t ns, size_t ns_valid, size_t njobs, size_t nsub, hipStream_t s) {
  if (rows == 0 || njobs == 0 || ns > 64) return ns > 64 ? hipErrorInvalidValue : hipSuccess;
  hipLaunchKernelGGL(k_slot_transpose, dim3((unsigned)rows, (unsigned)njobs), dim3(256), 0, s, out_plane, out_set_stride,
                     out_job_stride, stage, (unsigned)rows_pad, (unsigned)ns, (unsigned)ns_valid, (unsigned)nsub);
  return hipGetLastError();
}

// Blocks of 8 waves (two row tiles per B stage) when the limb has 4 planes, at least 2 row tiles and at least 2 column
// tiles; 6 (3) waves for a 3-plane limb with at least 2 (with 1) row tiles.  The caller derives rgroups from this.
int gemm_i8_waves(int planes, size_t rt_count, size_t nt) {
  if (planes == 3 && (nt == 6 || nt == 3) && rt_count >= 4) return 4;   // k_gemm_i8_w3: 4 waves = 4 row tiles x 3 planes each
  if (planes == 3) return rt_count >= 2 ? 6 : 3;
  return planes == 4 && rt_count >= 2 && nt >= 2 ? 8 : 4;
}

hipError_t launch_gemm_i8(const Gemm8Args& a, size_t nt, int waves, hipStream_t s) {
  dim3 g((unsigned)(D * a.rgroups * a.njobs));     // D = 64 slots = 8 per XCD
#define LAB_G8(NTV)                                                                                   \
  if (waves == 8) hipLaunchKernelGGL((k_gemm_i8<NTV, 8>), g, dim3(512), 0, s, a);                     \
  else hipLaunchKernelGGL((k_gemm_i8<NTV, 4>), g, dim3(256), 0, s, a)
#define LAB_G6(NTV)                                                                                   \
  if (waves == 6) hipLaunchKernelGGL((k_gemm_i8<NTV, 6>), g, dim3(384), 0, s, a);                     \
  else hipLaunchKernelGGL((k_gemm_i8<NTV, 3>), g, dim3(192), 0, s, a)
  if (a.planes == 3 && waves == 4) {                // 3-plane limbs, one row tile (all three planes) per wave
    if (nt == 6 && ((a.diag == 1 && a.njobs == 1) || (a.diag == 4 && a.njobs > 1 && a.njobs < 4))) hipLaunchKernelGGL((k_gemm_i8_w3<6, 1>), g, dim3(256), 0, s, a);
    else if (nt == 6 && a.diag == 2 && a.njobs == 1) hipLaunchKernelGGL((k_gemm_i8_w3<6, 2>), g, dim3(256), 0, s, a);
    else if (nt == 6 && a.diag == 3 && a.njobs == 1) hipLaunchKernelGGL((k_gemm_i8_w3<6, 3>), g, dim3(256), 0, s, a);
    else if (nt == 6 && a.njobs >= 4 && (a.diag == 0 || a.diag == 4)) {
      // many jobs share the matrix: persistent blocks, about four per CU over the launch
      Gemm8Args b = a;
      unsigned jpb = 16;
      while (jpb > 1 && (size_t)D * a.rgroups * ((a.njobs + jpb - 1) / jpb) < 1024) jpb /= 2;
      b.jobs_per_block = jpb;
      dim3 gp((unsigned)(D * a.rgroups * ((a.njobs + jpb - 1) / jpb)));
      const bool small = (uint64_t)a.kt_real * 32 * 3 * 16384 < (1ull << 31) && a.p > (1u << 22);
      if (a.diag == 4 && small) hipLaunchKernelGGL((k_gemm_i8_w3p<6, true, true>), gp, dim3(256), 0, s, b);
      else if (small) hipLaunchKernelGGL((k_gemm_i8_w3p<6, true>), gp, dim3(256), 0, s, b);
      else hipLaunchKernelGGL((k_gemm_i8_w3p<6, false>), gp, dim3(256), 0, s, b);
    }
    else if (nt == 6) hipLaunchKernelGGL((k_gemm_i8_w3<6>), g, dim3(256), 0, s, a);
    else if (nt == 3) hipLaunchKernelGGL((k_gemm_i8_w3<3>), g, dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (a.planes == 3) {                              // 3-plane limbs: blocks of 3 or 6 waves
    if (waves != 3 && waves != 6) return hipErrorInvalidValue;
    switch (nt) {
      case 1: LAB_G6(1); break;
      case 2: LAB_G6(2); break;
      case 3: LAB_G6(3); break;
      case 6: LAB_G6(6); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  if (waves != 4 && waves != 8) return hipErrorInvalidValue;
  switch (nt) {
    case 1:
      if (waves != 4) return hipErrorInvalidValue;
      hipLaunchKernelGGL((k_gemm_i8<1, 4>), g, dim3(256), 0, s, a);
      break;
    case 2: LAB_G8(2); break;
    case 4: LAB_G8(4); break;
    case 8: LAB_G8(8); break;
    default: return hipErrorInvalidValue;
  }
#undef LAB_G6
#undef LAB_G8
  return hipGetLastError();
}

}  // namespace lab
