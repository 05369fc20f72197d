// proof_kernels.hip — coefficient-domain kernels of the prover: digit
// decomposition, exact norm sums, the JL projection, the phi'' aggregation and
// the small mod-q elementwise steps.  All integer; gfx950 only.
//
// These replace src/util.rs:107-223, 360-442, 511-526 and the loops of
// src/proofgen.rs:227-255, 320-358, 429-456 of the reference.
#include "kernels.h"

namespace lab {

// (a*b) mod q for a,b < q < 2^40 without 128-bit division: split b at bit 20.
__device__ __forceinline__ uint64_t mulmod40(uint64_t a, uint64_t b, uint64_t q) {
  uint64_t t = (a * (b >> 20)) % q;          // < 2^60
  t = (t << 20) % q;
  return (t + a * (b & 0xFFFFF)) % q;        // < 2^61
}
__device__ __forceinline__ uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) { uint64_t s = a + b; return s >= q ? s - q : s; }
__device__ __forceinline__ uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }

// ---- decompose_polynomial_vec, bug-compatible (src/util.rs:360-442):
// per coefficient v: while v != 0 { r0 = v % b; r = r0 > b/2 ? (b - r0) % b : r0; push r; v = (v - r)/b }
// digits are non-negative (the centred sign is lost), digits past `exp` dropped.
// Output element (e, digit k) goes to index ((e/group)*exp + k)*group + e%group:
// group = count gives the reference's [exp][count] layout, group = 1 gives
// [count][exp], group = kappa gives [i][k][rho] for the t_i decomposition.
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

// OT = uint16_t: digits (< base <= 2^16) stored as 16-bit words for consumers inside the library (to_hat of the digits)
// FAST: base < 2^16 and base <= q — all quotients by `base` in 32-bit integer arithmetic (divmod_u40_u16)
template <typename OT, bool FAST>
__global__ void __launch_bounds__(256)
k_decompose(const JobPtrs outp, const JobPtrs inp, size_t nwords, uint64_t q, uint64_t base, int exp, size_t group) {
  OT* __restrict__ out = static_cast<OT*>(outp.p[blockIdx.y]);
  const uint64_t* __restrict__ in = static_cast<const uint64_t*>(inp.p[blockIdx.y]);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const uint64_t half = (base / 2) % q;  // `val > b/2` compares with mod_positive(b/2, Q) (algebraic.rs:293-297)
  const double inv_base = 1.0 / (double)base;
  const bool small_q = base > q;         // only then do the reference's `% Q` after `% base` do anything
  const uint32_t b32 = (uint32_t)base, bm = FAST ? (uint32_t)(0x100000000ull / base) : 0;
  for (; i < nwords; i += stride) {
    uint64_t v = in[i];
    const uint32_t e = (uint32_t)(i / D), g = (uint32_t)group, eg = e / g;   // element counts are below 2^32 (launch_decompose_jobs checks)
    OT* const o = out + ((size_t)eg * exp * g + (e - eg * g)) * D + i % D;
    for (int k = 0; k < exp; k++) {
      uint64_t r = 0;
      if (FAST) {
        // same arithmetic (v = 0 gives r = 0 and stays 0): r0 = v % base; r = r0 > half ? base - r0 : r0; v = (v - r mod q) / base
        uint64_t qq;
        const uint32_t r0 = divmod_u40_u16((uint32_t)(v >> 32), (uint32_t)v, b32, bm, &qq);
        r = r0 > half ? b32 - r0 : r0;
        const uint64_t w = submod(v, r, q);
        (void)divmod_u40_u16((uint32_t)(w >> 32), (uint32_t)w, b32, bm, &v);
      } else if (v != 0) {
        // r0 = (v % base) % q ; r = r0 > half ? ((base - r0) % base) % q : r0 ; v = ((v - r mod q) / base) % q
        // v < q < 2^40 and base >= 2: quotients by `base` through mod_u64/div_u64 (no integer division)
        uint64_t r0 = mod_u64_near(v, base, inv_base);
        if (small_q) r0 %= q;
        r = r0;
        if (r0 > half) {
          r = base - r0;                  // 0 < r0 < base here, so (base - r0) % base == base - r0
          if (small_q) r %= q;
        }
        v = div_u64_near(submod(v, r, q), base, inv_base);
        if (small_q) v %= q;
      }
      o[(size_t)k * g * D] = (OT)r;
    }
  }
}

hipError_t launch_decompose_jobs(const JobPtrs& out, const JobPtrs& in, size_t count, uint64_t q, uint64_t base, int exp,
                                 size_t group, size_t njobs, hipStream_t s, bool out_u16) {
  if (count == 0 || exp == 0 || njobs == 0) return hipSuccess;
  if (count >> 32 || group >> 32 || group == 0) return hipErrorInvalidValue;
  size_t nwords = count * D;
  unsigned blocks = (unsigned)((nwords + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  const bool fast = base < 65536 && base <= q && q < (1ull << 40);
  const dim3 gr(blocks, (unsigned)njobs);
  if (out_u16 && base > 65536) return hipErrorInvalidValue;
#define LAB_DEC(OT, F) hipLaunchKernelGGL((k_decompose<OT, F>), gr, dim3(256), 0, s, out, in, nwords, q, base, exp, group)
  if (out_u16) { if (fast) LAB_DEC(uint16_t, true); else LAB_DEC(uint16_t, false); }
  else { if (fast) LAB_DEC(uint64_t, true); else LAB_DEC(uint64_t, false); }
#undef LAB_DEC
  return hipGetLastError();
}

hipError_t launch_decompose(uint64_t* out, const uint64_t* in, size_t count, uint64_t q, uint64_t base, int exp,
                            size_t group, hipStream_t s) {
  JobPtrs o, i;
  o.p[0] = out;
  i.p[0] = const_cast<uint64_t*>(in);
  return launch_decompose_jobs(o, i, count, q, base, exp, group, 1, s, false);
}

// ---- poly_norm (src/util.rs:195-202): exact sum of squares of the RAW residues
// of each polynomial as a 128-bit integer (lo,hi).  One wave per polynomial, one
// coefficient per lane, wave shuffle reduction with carry.
__global__ void __launch_bounds__(256)
k_norm_sq(uint64_t* __restrict__ lo, uint64_t* __restrict__ hi, const uint64_t* __restrict__ in, size_t count) {
  const int lane = threadIdx.x & 63;
  size_t poly = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t stride = (size_t)gridDim.x * 4;
  for (; poly < count; poly += stride) {
    const uint64_t v = in[poly * D + lane];
    uint64_t l = v * v, h = __umul64hi(v, v);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint64_t ol = __shfl_down(l, off), oh = __shfl_down(h, off);
      const uint64_t nl = l + ol;
      h = h + oh + (nl < l ? 1 : 0);
      l = nl;
    }
    if (lane == 0) { lo[poly] = l; hi[poly] = h; }
  }
}

hipError_t launch_norm_sq(uint64_t* lo, uint64_t* hi, const uint64_t* in, size_t count, hipStream_t s) {
  if (count == 0) return hipSuccess;
  unsigned blocks = (unsigned)((count + 3) / 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_norm_sq, dim3(blocks), dim3(256), 0, s, lo, hi, in, count);
  return hipGetLastError();
}

// ---- elementwise mod-q helpers on coefficient arrays
// mode 0: out = s1*a                       (scale_polynomial, src/util.rs:176-180)
// mode 1: out = s1*a + b                   (phi'' = psi*phi' + W, src/proofgen.rs:236-254)
// mode 2: out = a + b
// mode 3: out = sigma_inv(a) per polynomial (src/util.rs:118-137)
__global__ void __launch_bounds__(256)
k_coef_elementwise(const JobPtrs outp, const JobPtrs ap, const JobPtrs bp, size_t nwords, uint64_t q, const JobU64 s1v, int mode) {
  uint64_t* __restrict__ out = static_cast<uint64_t*>(outp.p[blockIdx.y]);
  const uint64_t* __restrict__ a = static_cast<const uint64_t*>(ap.p[blockIdx.y]);
  const uint64_t* __restrict__ b = static_cast<const uint64_t*>(bp.p[blockIdx.y]);
  const uint64_t s1 = s1v.v[blockIdx.y];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nwords; i += stride) {
    uint64_t r;
    if (mode == 0) r = mulmod40(a[i], s1, q);
    else if (mode == 1) r = addmod(mulmod40(a[i], s1, q), b[i], q);
    else if (mode == 2) r = addmod(a[i], b[i], q);
    else {
      const size_t poly = i / D, k = i % D;
      const uint64_t src = a[poly * D + (k == 0 ? 0 : D - k)];
      r = k == 0 ? src : (src == 0 ? 0 : q - src);
    }
    out[i] = r;
  }
}

hipError_t launch_coef_elementwise_jobs(const JobPtrs& out, const JobPtrs& a, const JobPtrs& b, size_t count, uint64_t q,
                                        const JobU64& s1, int mode, size_t njobs, hipStream_t s) {
  if (count == 0 || njobs == 0) return hipSuccess;
  size_t nwords = count * D;
  unsigned blocks = (unsigned)((nwords + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_coef_elementwise, dim3(blocks, (unsigned)njobs), dim3(256), 0, s, out, a, b, nwords, q, s1, mode);
  return hipGetLastError();
}

hipError_t launch_coef_elementwise(uint64_t* out, const uint64_t* a, const uint64_t* b, size_t count, uint64_t q,
                                   uint64_t s1, int mode, hipStream_t s) {
  JobPtrs o, pa, pb;
  JobU64 sv;
  o.p[0] = out; pa.p[0] = const_cast<uint64_t*>(a); pb.p[0] = const_cast<uint64_t*>(b); sv.v[0] = s1;
  return launch_coef_elementwise_jobs(o, pa, pb, count, q, sv, mode, 1, s);
}

// H[i][k] = inv2 * (X[i][k] + X[k][i])   (src/proofgen.rs:334-350), X: [r][r][64]
__global__ void __launch_bounds__(256)
k_symmetrize_half(const JobPtrs Hp, const JobPtrs Xp, unsigned r, uint64_t q, uint64_t inv2) {
  uint64_t* __restrict__ H = static_cast<uint64_t*>(Hp.p[blockIdx.y]);
  const uint64_t* __restrict__ X = static_cast<const uint64_t*>(Xp.p[blockIdx.y]);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)r * r * D;
  if (i >= n) return;
  const unsigned c = i % D, e = i / D, row = e / r, col = e % r;
  const uint64_t s = addmod(X[i], X[((size_t)col * r + row) * D + c], q);
  H[i] = mulmod40(s, inv2, q);
}

hipError_t launch_symmetrize_half_jobs(const JobPtrs& H, const JobPtrs& X, size_t r, uint64_t q, uint64_t inv2, size_t njobs,
                                       hipStream_t s) {
  if (r == 0 || njobs == 0) return hipSuccess;
  size_t n = r * r * D;
  hipLaunchKernelGGL(k_symmetrize_half, dim3((unsigned)((n + 255) / 256), (unsigned)njobs), dim3(256), 0, s, H, X, (unsigned)r, q, inv2);
  return hipGetLastError();
}

hipError_t launch_symmetrize_half(uint64_t* H, const uint64_t* X, size_t r, uint64_t q, uint64_t inv2, hipStream_t s) {
  JobPtrs h, x;
  h.p[0] = H; x.p[0] = const_cast<uint64_t*>(X);
  return launch_symmetrize_half_jobs(h, x, r, q, inv2, 1, s);
}

// upper triangle (i <= j) of an [r][r][64] array in pair order i*r - i(i-1)/2 + (j-i) (src/structs.rs:101-106)
__global__ void __launch_bounds__(256)
k_gather_upper(const JobPtrs outp, const JobPtrs inp, unsigned r) {
  uint64_t* __restrict__ out = static_cast<uint64_t*>(outp.p[blockIdx.y]);
  const uint64_t* __restrict__ in = static_cast<const uint64_t*>(inp.p[blockIdx.y]);
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)r * r * D;
  if (idx >= n) return;
  const unsigned c = idx % D, e = idx / D, i = e / r, j = e % r;
  if (j < i) return;
  const size_t pair = (size_t)i * r - ((size_t)i * (i - 1)) / 2 + (j - i);
  out[pair * D + c] = in[idx];
}

hipError_t launch_gather_upper_jobs(const JobPtrs& out, const JobPtrs& in, size_t r, size_t njobs, hipStream_t s) {
  if (r == 0 || njobs == 0) return hipSuccess;
  size_t n = r * r * D;
  hipLaunchKernelGGL(k_gather_upper, dim3((unsigned)((n + 255) / 256), (unsigned)njobs), dim3(256), 0, s, out, in, (unsigned)r);
  return hipGetLastError();
}

hipError_t launch_gather_upper(uint64_t* out, const uint64_t* in, size_t r, hipStream_t s) {
  JobPtrs o, i;
  o.p[0] = out; i.p[0] = const_cast<uint64_t*>(in);
  return launch_gather_upper_jobs(o, i, r, 1, s);
}

// proj[m] mod q (Zq::lift, src/proofgen.rs:186)
__global__ void k_lift_i64(const JobPtrs outp, const JobPtrs inp, unsigned n, uint64_t q) {
  uint64_t* __restrict__ out = static_cast<uint64_t*>(outp.p[blockIdx.y]);
  const int64_t* __restrict__ in = static_cast<const int64_t*>(inp.p[blockIdx.y]);
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = in[i];
  const uint64_t m = (uint64_t)(v < 0 ? -v : v) % q;
  out[i] = v < 0 ? (m == 0 ? 0 : q - m) : m;
}

// dst[job*words + i] = src[job][i] / the reverse: small per-job arrays to and from one contiguous staging buffer
__global__ void k_gather_jobs(uint64_t* __restrict__ dst, const JobPtrs srcp, unsigned words) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words) dst[(size_t)blockIdx.y * words + i] = static_cast<const uint64_t*>(srcp.p[blockIdx.y])[i];
}
__global__ void k_scatter_jobs(const JobPtrs dstp, const uint64_t* __restrict__ src, unsigned words) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words && dstp.p[blockIdx.y]) static_cast<uint64_t*>(dstp.p[blockIdx.y])[i] = src[(size_t)blockIdx.y * words + i];
}
hipError_t launch_gather_jobs(uint64_t* dst, const JobPtrs& src, size_t words, size_t njobs, hipStream_t s) {
  if (words == 0 || njobs == 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather_jobs, dim3((unsigned)((words + 255) / 256), (unsigned)njobs), dim3(256), 0, s, dst, src, (unsigned)words);
  return hipGetLastError();
}
hipError_t launch_scatter_jobs(const JobPtrs& dst, const uint64_t* src, size_t words, size_t njobs, hipStream_t s) {
  if (words == 0 || njobs == 0) return hipSuccess;
  hipLaunchKernelGGL(k_scatter_jobs, dim3((unsigned)((words + 255) / 256), (unsigned)njobs), dim3(256), 0, s, dst, src, (unsigned)words);
  return hipGetLastError();
}

hipError_t launch_lift_i64_jobs(const JobPtrs& out, const JobPtrs& in, size_t n, uint64_t q, size_t njobs, hipStream_t s) {
  if (n == 0 || njobs == 0) return hipSuccess;
  hipLaunchKernelGGL(k_lift_i64, dim3((unsigned)((n + 255) / 256), (unsigned)njobs), dim3(256), 0, s, out, in, (unsigned)n, q);
  return hipGetLastError();
}

hipError_t launch_lift_i64(uint64_t* out, const int64_t* in, size_t n, uint64_t q, hipStream_t s) {
  JobPtrs o, i;
  o.p[0] = out; i.p[0] = const_cast<int64_t*>(in);
  return launch_lift_i64_jobs(o, i, n, q, 1, s);
}

// ---- JL projection (src/proofgen.rs:429-456, src/util.rs:511-526):
//   p[m] += sum_col Pi_i[m][col] * s_i[col]     over the integers, Pi in {-1,0,1}
// Pi: int8 [r][256][cols]; S: coefficient form [n][r][64] (s_i[col] = S[col/64][i][col%64]).
// HBM-bound on Pi.  A block owns 16 rows m of one i and a 4096-column chunk:
// each thread keeps 16 witness coefficients in registers and reuses them for
// the 16 rows (16 B of Pi per row per thread).  Row sums are reduced with wave
// shuffles, then one 64-bit atomic per row per block (exact, order independent).
constexpr int JL_ROWS = 16;
constexpr int JL_COLS_PER_THREAD = 16;

__global__ void __launch_bounds__(256)
k_jl_project(unsigned long long* __restrict__ proj, const int8_t* __restrict__ Pi, const uint64_t* __restrict__ S,
             unsigned n, unsigned r) {
  __shared__ long long red[4][JL_ROWS];
  const size_t cols = (size_t)n * D;
  const unsigned i = blockIdx.z, m0 = blockIdx.y * JL_ROWS;
  const size_t col0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * JL_COLS_PER_THREAD;
  long long acc[JL_ROWS];
#pragma unroll
  for (int m = 0; m < JL_ROWS; m++) acc[m] = 0;
  if (col0 < cols) {
    // 16 consecutive columns lie inside one polynomial (16 | 64)
    const uint64_t* sp = S + ((col0 / D) * r + i) * D + (col0 % D);
    long long s[JL_COLS_PER_THREAD];
#pragma unroll
    for (int c = 0; c < JL_COLS_PER_THREAD; c += 2) {
      const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(sp + c);
      s[c] = (long long)v.x; s[c + 1] = (long long)v.y;
    }
    const int8_t* pp = Pi + ((size_t)i * 256 + m0) * cols + col0;
#pragma unroll
    for (int m = 0; m < JL_ROWS; m++) {
      const uint4 w = *reinterpret_cast<const uint4*>(pp + (size_t)m * cols);
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
      long long a = 0;
#pragma unroll
      for (int c = 0; c < JL_COLS_PER_THREAD; c++) {
        const int8_t e = (int8_t)(ww[c >> 2] >> ((c & 3) * 8));
        a += e > 0 ? s[c] : (e < 0 ? -s[c] : 0);
      }
      acc[m] = a;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int m = 0; m < JL_ROWS; m++) {
    long long a = acc[m];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if (lane == 0) red[wave][m] = a;
  }
  __syncthreads();
  if (threadIdx.x < JL_ROWS) {
    const long long t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(&proj[m0 + threadIdx.x], (unsigned long long)t);
  }
}

hipError_t launch_jl_project(int64_t* proj /*[256], zeroed by caller*/, const int8_t* Pi, const uint64_t* S, size_t n,
                             size_t r, hipStream_t s) {
  if (n == 0 || r == 0) return hipSuccess;
  const size_t cols = n * D;
  dim3 grid((unsigned)((cols + 4095) / 4096), 256 / JL_ROWS, (unsigned)r);
  hipLaunchKernelGGL(k_jl_project, grid, dim3(256), 0, s, reinterpret_cast<unsigned long long*>(proj), Pi, S,
                     (unsigned)n, (unsigned)r);
  return hipGetLastError();
}

// ---- the Pi part of phi'' (src/proofgen.rs:244-253):
//   W[x][i] = sum_{m<256} omega[m] * sigma_inv( poly(Pi_i[m][64x .. 64x+63] lifted to Zq) )    (mod q)
// With u[col] = sum_m omega[m]*Pi_i[m][col] this is W[x][i][0] = u[64x], W[x][i][64-k] = -u[64x+k].
// Output layout [n][r][64] like phi'.  A thread owns one column; lanes run over
// consecutive columns so every Pi row segment is read coalesced.
__global__ void __launch_bounds__(256)
k_phi_pi_part(uint64_t* __restrict__ W, const int8_t* __restrict__ Pi, const uint64_t* __restrict__ omega, unsigned n,
              unsigned r, uint64_t q) {
  __shared__ uint64_t om[256];
  om[threadIdx.x] = omega[threadIdx.x];
  __syncthreads();
  const size_t cols = (size_t)n * D;
  const unsigned i = blockIdx.y;
  const size_t col = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= cols) return;
  const int8_t* pp = Pi + (size_t)i * 256 * cols + col;
  uint64_t pos = 0, neg = 0;  // 256 terms below 2^40 each
#pragma unroll 8
  for (int m = 0; m < 256; m++) {
    const int8_t e = pp[(size_t)m * cols];
    pos += e > 0 ? om[m] : 0;
    neg += e < 0 ? om[m] : 0;
  }
  const uint64_t u = submod(pos % q, neg % q, q);
  const size_t x = col / D, k = col % D;
  const uint64_t val = k == 0 ? u : (u == 0 ? 0 : q - u);
  W[(x * r + i) * D + (k == 0 ? 0 : D - k)] = val;
}

hipError_t launch_phi_pi_part(uint64_t* W, const int8_t* Pi, const uint64_t* omega, size_t n, size_t r, uint64_t q,
                              hipStream_t s) {
  if (n == 0 || r == 0) return hipSuccess;
  const size_t cols = n * D;
  hipLaunchKernelGGL(k_phi_pi_part, dim3((unsigned)((cols + 255) / 256), (unsigned)r), dim3(256), 0, s, W, Pi, omega,
                     (unsigned)n, (unsigned)r, q);
  return hipGetLastError();
}

}  // namespace lab

namespace lab {

// ---- small helpers for the verifier (src/verification.rs:157-178, 291, 312, 332, 350, 413, 433)
// flag[0] += number of mismatching words (0 = equal)
__global__ void __launch_bounds__(256)
k_compare_u64(unsigned int* __restrict__ flag, const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, size_t nwords) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned int bad = 0;
  for (; i < nwords; i += stride) bad += a[i] != b[i] ? 1u : 0u;
  if (bad) atomicAdd(flag, bad);
}

hipError_t launch_compare_u64(unsigned int* flag, const uint64_t* a, const uint64_t* b, size_t nwords, hipStream_t s) {
  if (nwords == 0) return hipSuccess;
  unsigned blocks = (unsigned)((nwords + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_compare_u64, dim3(blocks), dim3(256), 0, s, flag, a, b, nwords);
  return hipGetLastError();
}

// X[i][j] == X[j][i] for an [r][r][64] array (checks 8 and 9)
__global__ void __launch_bounds__(256)
k_symmetry_u64(unsigned int* __restrict__ flag, const uint64_t* __restrict__ X, unsigned r) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)r * r * D;
  if (idx >= n) return;
  const unsigned c = idx % D, e = idx / D, i = e / r, j = e % r;
  if (X[idx] != X[((size_t)j * r + i) * D + c]) atomicAdd(flag, 1u);
}

hipError_t launch_symmetry_u64(unsigned int* flag, const uint64_t* X, size_t r, hipStream_t s) {
  if (r == 0) return hipSuccess;
  size_t n = r * r * D;
  hipLaunchKernelGGL(k_symmetry_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, flag, X, (unsigned)r);
  return hipGetLastError();
}

// check 18 (src/verification.rs:340-352): res = s1 + sum_i H[i][i] - b  (one polynomial, mod q)
__global__ void k_check18(uint64_t* __restrict__ res, const uint64_t* __restrict__ s1, const uint64_t* __restrict__ H,
                          const uint64_t* __restrict__ b, unsigned r, uint64_t q) {
  const unsigned c = threadIdx.x;
  if (c >= D) return;
  uint64_t acc = s1[c];
  for (unsigned i = 0; i < r; i++) acc = addmod(acc, H[((size_t)i * r + i) * D + c], q);
  res[c] = submod(acc, b[c], q);
}

// out[c] = base[c] + sum_i X[i][i][c] mod q per job (the <phi''_i, s_i> part of b'' read off the diagonal of phi''^T S)
__global__ void k_add_trace_jobs(const JobPtrs outp, const JobPtrs basep, const JobPtrs Xp, unsigned r, uint64_t q) {
  uint64_t* __restrict__ out = static_cast<uint64_t*>(outp.p[blockIdx.x]);
  const uint64_t* __restrict__ base = static_cast<const uint64_t*>(basep.p[blockIdx.x]);
  const uint64_t* __restrict__ X = static_cast<const uint64_t*>(Xp.p[blockIdx.x]);
  const unsigned c = threadIdx.x;
  uint64_t acc = base[c];
  for (unsigned i = 0; i < r; i++) acc = addmod(acc, X[((size_t)i * r + i) * D + c], q);
  out[c] = acc;
}

hipError_t launch_add_trace_jobs(const JobPtrs& out, const JobPtrs& base, const JobPtrs& X, size_t r, uint64_t q, size_t njobs,
                                 hipStream_t s) {
  if (njobs == 0) return hipSuccess;
  hipLaunchKernelGGL(k_add_trace_jobs, dim3((unsigned)njobs), dim3(64), 0, s, out, base, X, (unsigned)r, q);
  return hipGetLastError();
}

hipError_t launch_check18(uint64_t* res, const uint64_t* s1, const uint64_t* H, const uint64_t* b, size_t r, uint64_t q,
                          hipStream_t s) {
  hipLaunchKernelGGL(k_check18, dim3(1), dim3(64), 0, s, res, s1, H, b, (unsigned)r, q);
  return hipGetLastError();
}

}  // namespace lab

namespace lab {

// Wave-wide integer sum through DPP (no LDS round trips): inclusive prefix sums inside each row of 16 lanes
// (row_shr 1, 2, 4, 8), then row 0 -> row 1 and row 2 -> row 3 (row_bcast:15), then lane 31 -> rows 2, 3
// (row_bcast:31).  The total is valid in lane 63 only.
__device__ __forceinline__ int wave_sum_to_lane63(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
  return x;
}

// ---- one pass over Pi for both of its uses (src/proofgen.rs:429-456 and :244-253):
//   proj[m]    += sum_col Pi_i[m][col] * s_i[col]                       (JL projection, integers)
//   W[x][i][.]  = sum_m omega[m] * sigma_inv(poly(Pi_i[m][64x..]))      (Pi part of phi'')
// A block owns witness column i and a 4096-column chunk and walks all 256 rows: a thread keeps 16
// witness coefficients and 16+16 column accumulators in registers, loads 16 B of Pi per row (8 rows issued
// together, the next 8 in flight while these are consumed: a CU needs > 50 KB outstanding to stream at HBM rate),
// reduces the row sum across the wave with DPP adds.  Pi (1 byte per entry) is read exactly once.
// Every product is (entry in {-1,0,1}) x (20-bit half of a value < 2^40), and every sum of them stays below 2^31
// (row part: 16 columns x 64 lanes x 2^20; column part: 256 rows x 2^20), so all accumulation is v_mad_i32_i24.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_pi_fused(const JobPtrs projp, const JobPtrs Wp, const JobPtrs Pip, const JobPtrs Sp, const JobPtrs omegap, unsigned n,
           unsigned r, uint64_t q) {
  unsigned long long* __restrict__ proj = static_cast<unsigned long long*>(projp.p[blockIdx.z]);
  uint64_t* __restrict__ W = static_cast<uint64_t*>(Wp.p[blockIdx.z]);
  const int8_t* __restrict__ Pi = static_cast<const int8_t*>(Pip.p[blockIdx.z]);
  const uint64_t* __restrict__ S = static_cast<const uint64_t*>(Sp.p[blockIdx.z]);
  const uint64_t* __restrict__ omega = static_cast<const uint64_t*>(omegap.p[blockIdx.z]);
  constexpr int GR = 4;                      // rows per load group
  __shared__ int om_lo[256], om_hi[256];
  __shared__ long long rowsum[4][256];
  {
    const uint64_t o = omega[threadIdx.x];
    om_lo[threadIdx.x] = (int)(o & 0xFFFFF);
    om_hi[threadIdx.x] = (int)(o >> 20);
  }
  __syncthreads();
  const size_t cols = (size_t)n * D;
  const unsigned i = blockIdx.y;
  const size_t col0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  const bool active = col0 < cols;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int s_lo[16], s_hi[16], wl[16], wh[16];
#pragma unroll
  for (int c = 0; c < 16; c++) { s_lo[c] = 0; s_hi[c] = 0; wl[c] = 0; wh[c] = 0; }
  if (active) {
    const uint64_t* sp = S + ((col0 / D) * r + i) * D + (col0 % D);
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
      const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(sp + c);
      s_lo[c] = (int)(v.x & 0xFFFFF); s_hi[c] = (int)(v.x >> 20);
      s_lo[c + 1] = (int)(v.y & 0xFFFFF); s_hi[c + 1] = (int)(v.y >> 20);
    }
  }
  // inactive threads (ragged last chunk) read column 0 of their row: no divergence in the loop, their witness
  // coefficients are 0 (row sums unaffected) and their column sums are never stored
  const int8_t* pp = Pi + (size_t)i * 256 * cols + (active ? col0 : 0);
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t cur[GR], nxt[GR];
#pragma unroll
  for (int u = 0; u < GR; u++) cur[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(pp + (size_t)u * cols));
  for (int m0 = 0; m0 < 256; m0 += GR) {
    if (m0 + GR < 256) {
#pragma unroll
      for (int u = 0; u < GR; u++)
        nxt[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(pp + (size_t)(m0 + GR + u) * cols));
    }
#pragma unroll
    for (int u = 0; u < GR; u++) {
      const int m = m0 + u;
      const uint32_t ww[4] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
      const int ol = om_lo[m], oh = om_hi[m];
      int a_lo = 0, a_hi = 0;
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const int e = (int)(int8_t)(ww[c >> 2] >> ((c & 3) * 8));   // inactive threads: s = 0, W not stored
        a_lo = __mul24(e, s_lo[c]) + a_lo;
        a_hi = __mul24(e, s_hi[c]) + a_hi;
        wl[c] = __mul24(e, ol) + wl[c];
        wh[c] = __mul24(e, oh) + wh[c];
      }
      a_lo = wave_sum_to_lane63(a_lo);
      a_hi = wave_sum_to_lane63(a_hi);
      if (lane == 63) rowsum[wave][m] = (long long)a_lo + ((long long)a_hi << 20);
    }
#pragma unroll
    for (int u = 0; u < GR; u++) cur[u] = nxt[u];
  }
  __syncthreads();
  {
    const int m = threadIdx.x;
    const long long t = rowsum[0][m] + rowsum[1][m] + rowsum[2][m] + rowsum[3][m];
    if (t != 0) atomicAdd(&proj[m], (unsigned long long)t);
  }
  if (active) {
    const size_t x = col0 / D, k0 = col0 % D;
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const long long v = (long long)wl[c] + ((long long)wh[c] << 20);   // sum_m omega[m]*Pi[m][col], |v| < 2^49
      const uint64_t mag = (uint64_t)(v < 0 ? -v : v) % q;
      const uint64_t u = v < 0 ? (mag == 0 ? 0 : q - mag) : mag;
      const size_t k = k0 + c;
      W[(x * r + i) * D + (k == 0 ? 0 : D - k)] = k == 0 ? u : (u == 0 ? 0 : q - u);
    }
  }
}

// ---- the same pass over Pi stored at 2 bits per entry (lab_prove_input.Pi2; the reference lifts Pi to 16-byte Zq,
// src/proofgen.rs:445-453, this path reads a quarter of the int8 form's bytes).  Entry (i, m, col) is the 2-bit field
// 2*(col % 4) of byte Pi2[(i*256 + m) * cols/4 + col/4], holding e + 1 in {0, 1, 2} — so a thread's 16 columns are one
// dword per row, and every product is unsigned (v_mad_u32_u24); the surplus sum_col s[col] (rows) resp. sum_m omega[m]
// (columns) that the +1 adds is taken off once per block / per thread.  8 rows are in flight per thread behind the 8 being
// consumed.  SMALL: the caller guarantees witness coefficients below 2^24 (a witness inside its norm bound
// beta < 2^24, checked by the digit-basis step 1 of the same call): one multiply-add per entry for the projection.
template <bool SMALL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_pi_fused2(const JobPtrs projp, const JobPtrs Wp, const JobPtrs Pip, const JobPtrs Sp, const JobPtrs omegap, unsigned n,
            unsigned r, uint64_t q) {
  unsigned long long* __restrict__ proj = static_cast<unsigned long long*>(projp.p[blockIdx.z]);
  uint64_t* __restrict__ W = static_cast<uint64_t*>(Wp.p[blockIdx.z]);
  const uint32_t* __restrict__ Pi = static_cast<const uint32_t*>(Pip.p[blockIdx.z]);
  const uint64_t* __restrict__ S = static_cast<const uint64_t*>(Sp.p[blockIdx.z]);
  const uint64_t* __restrict__ omega = static_cast<const uint64_t*>(omegap.p[blockIdx.z]);
  constexpr int GR = 8;                      // rows per load group
  __shared__ unsigned om_lo[256], om_hi[256];
  __shared__ long long rowsum[4][256];
  __shared__ long long ssum[4], osum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    const uint64_t o = omega[threadIdx.x];
    om_lo[threadIdx.x] = (unsigned)(o & 0xFFFFF);
    om_hi[threadIdx.x] = (unsigned)(o >> 20);
    // sum_m omega[m] (< 2^48): halves summed across the wave (64 x 2^20 and 64 x 2^20: no overflow)
    const int lo = wave_sum_to_lane63((int)(o & 0xFFFFF)), hi = wave_sum_to_lane63((int)(o >> 20));
    if (lane == 63) osum[wave] = (long long)lo + ((long long)hi << 20);
  }
  const size_t cols = (size_t)n * D;
  const unsigned i = blockIdx.y;
  const size_t col0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  const bool active = col0 < cols;
  unsigned s_lo[16], s_hi[SMALL ? 1 : 16], wl[16], wh[16];
#pragma unroll
  for (int c = 0; c < 16; c++) { s_lo[c] = 0; wl[c] = 0; wh[c] = 0; }
  if (!SMALL) {
#pragma unroll
    for (int c = 0; c < 16; c++) s_hi[SMALL ? 0 : c] = 0;
  }
  long long my_s = 0;
  if (active) {
    const uint64_t* sp = S + ((col0 / D) * r + i) * D + (col0 % D);
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
      const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(sp + c);
      if (SMALL) {
        s_lo[c] = (unsigned)v.x & 0xFFFFFF; s_lo[c + 1] = (unsigned)v.y & 0xFFFFFF;
      } else {
        s_lo[c] = (unsigned)(v.x & 0xFFFFF); s_hi[SMALL ? 0 : c] = (unsigned)(v.x >> 20);
        s_lo[c + 1] = (unsigned)(v.y & 0xFFFFF); s_hi[SMALL ? 0 : c + 1] = (unsigned)(v.y >> 20);
      }
      my_s += (long long)(SMALL ? (v.x & 0xFFFFFF) : v.x) + (long long)(SMALL ? (v.y & 0xFFFFFF) : v.y);
    }
  }
  {  // this block's sum of witness coefficients (< 4096 * 2^40): 20-bit pieces summed across the wave
    const int p0 = wave_sum_to_lane63((int)(my_s & 0xFFFFF)), p1 = wave_sum_to_lane63((int)((my_s >> 20) & 0xFFFFF)),
              p2 = wave_sum_to_lane63((int)(my_s >> 40));
    if (lane == 63) ssum[wave] = (long long)p0 + ((long long)p1 << 20) + ((long long)p2 << 40);
  }
  __syncthreads();
  // inactive threads (ragged last chunk) read dword 0 of their row: no divergence in the loop, their witness
  // coefficients are 0 (row sums unaffected) and their column sums are never stored
  // row m of this matrix = a wave-uniform base (scalar registers, advanced per row) + one 32-bit dword index per lane
  const size_t row_dwords = cols / 16;
  const uint32_t* rows = Pi + (size_t)i * 256 * row_dwords;
  const unsigned voff = active ? (unsigned)(col0 / 16) : 0u;
  uint32_t cur[GR], nxt[GR];
#pragma unroll
  for (int u = 0; u < GR; u++) cur[u] = __builtin_nontemporal_load(rows + (size_t)u * row_dwords + voff);
#pragma unroll 1
  for (int m0 = 0; m0 < 256; m0 += GR) {
    if (m0 + GR < 256) {
      const uint32_t* nrows = rows + (size_t)(m0 + GR) * row_dwords;
#pragma unroll
      for (int u = 0; u < GR; u++) nxt[u] = __builtin_nontemporal_load(nrows + (size_t)u * row_dwords + voff);
    }
#pragma unroll
    for (int u = 0; u < GR; u++) {
      const int m = m0 + u;
      const uint32_t w = cur[u];
      const unsigned ol = om_lo[m], oh = om_hi[m];
      unsigned a_lo = 0, a_hi = 0;
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const unsigned g = (w >> (2 * c)) & 3u;          // e + 1
        a_lo = __umul24(g, s_lo[c]) + a_lo;
        if (!SMALL) a_hi = __umul24(g, s_hi[SMALL ? 0 : c]) + a_hi;
        wl[c] = __umul24(g, ol) + wl[c];
        wh[c] = __umul24(g, oh) + wh[c];
      }
      if (SMALL) { a_hi = a_lo >> 20; a_lo &= 0xFFFFF; }   // a < 16 * 2 * 2^24: split so that the wave sums stay below 2^32
      a_lo = (unsigned)wave_sum_to_lane63((int)a_lo);
      a_hi = (unsigned)wave_sum_to_lane63((int)a_hi);
      if (lane == 63) rowsum[wave][m] = (long long)a_lo + ((long long)a_hi << 20);
    }
#pragma unroll
    for (int u = 0; u < GR; u++) cur[u] = nxt[u];
  }
  __syncthreads();
  {
    const int m = threadIdx.x;
    const long long t = rowsum[0][m] + rowsum[1][m] + rowsum[2][m] + rowsum[3][m] - (ssum[0] + ssum[1] + ssum[2] + ssum[3]);
    if (t != 0) atomicAdd(&proj[m], (unsigned long long)t);
  }
  if (active) {
    const size_t x = col0 / D, k0 = col0 % D;
    const long long om_total = osum[0] + osum[1] + osum[2] + osum[3];
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const long long v = (long long)wl[c] + ((long long)wh[c] << 20) - om_total;   // sum_m omega[m]*Pi[m][col], |v| < 2^49
      const uint64_t mag = (uint64_t)(v < 0 ? -v : v) % q;
      const uint64_t u = v < 0 ? (mag == 0 ? 0 : q - mag) : mag;
      const size_t k = k0 + c;
      W[(x * r + i) * D + (k == 0 ? 0 : D - k)] = k == 0 ? u : (u == 0 ? 0 : q - u);
    }
  }
}

// Pi int8 [count] in {-1, 0, 1} -> 2 bits per entry (e + 1), 16 entries per thread
__global__ void k_pi_pack(uint32_t* __restrict__ dst, const int8_t* __restrict__ src, size_t ndwords) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ndwords) return;
  const uint4 v = *reinterpret_cast<const uint4*>(src + k * 16);
  const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
  uint32_t o = 0;
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const int e = (int)(int8_t)(ww[c >> 2] >> ((c & 3) * 8));
    o |= (uint32_t)((e + 1) & 3) << (2 * c);
  }
  dst[k] = o;
}

hipError_t launch_pi_pack(uint8_t* dst, const int8_t* src, size_t count, hipStream_t s) {
  if (count % 16) return hipErrorInvalidValue;
  const size_t nd = count / 16;
  if (nd == 0) return hipSuccess;
  hipLaunchKernelGGL(k_pi_pack, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, reinterpret_cast<uint32_t*>(dst), src, nd);
  return hipGetLastError();
}

// packed = true: `Pi` holds the 2-bit form (see k_pi_fused2); small_witness: witness coefficients are known to be below 2^24
hipError_t launch_pi_fused2_jobs(const JobPtrs& proj /*zeroed*/, const JobPtrs& W, const JobPtrs& Pi2, const JobPtrs& S,
                                 const JobPtrs& omega, size_t n, size_t r, uint64_t q, size_t njobs, bool small_witness, hipStream_t s) {
  if (n == 0 || r == 0 || njobs == 0) return hipSuccess;
  const size_t cols = n * D;
  const dim3 g((unsigned)((cols + 4095) / 4096), (unsigned)r, (unsigned)njobs);
  if (small_witness)
    hipLaunchKernelGGL(k_pi_fused2<true>, g, dim3(256), 0, s, proj, W, Pi2, S, omega, (unsigned)n, (unsigned)r, q);
  else
    hipLaunchKernelGGL(k_pi_fused2<false>, g, dim3(256), 0, s, proj, W, Pi2, S, omega, (unsigned)n, (unsigned)r, q);
  return hipGetLastError();
}

hipError_t launch_pi_fused_jobs(const JobPtrs& proj /*zeroed*/, const JobPtrs& W, const JobPtrs& Pi, const JobPtrs& S,
                                const JobPtrs& omega, size_t n, size_t r, uint64_t q, size_t njobs, hipStream_t s) {
  if (n == 0 || r == 0 || njobs == 0) return hipSuccess;
  const size_t cols = n * D;
  hipLaunchKernelGGL(k_pi_fused, dim3((unsigned)((cols + 4095) / 4096), (unsigned)r, (unsigned)njobs), dim3(256), 0, s, proj, W, Pi, S,
                     omega, (unsigned)n, (unsigned)r, q);
  return hipGetLastError();
}

hipError_t launch_pi_fused(int64_t* proj /*zeroed*/, uint64_t* W, const int8_t* Pi, const uint64_t* S,
                           const uint64_t* omega, size_t n, size_t r, uint64_t q, hipStream_t s) {
  JobPtrs p, w, pi, sp, om;
  p.p[0] = proj; w.p[0] = W; pi.p[0] = const_cast<int8_t*>(Pi); sp.p[0] = const_cast<uint64_t*>(S);
  om.p[0] = const_cast<uint64_t*>(omega);
  return launch_pi_fused_jobs(p, w, pi, sp, om, n, r, q, 1, s);
}

}  // namespace lab
