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
// ---- decompose_polynomial_vec, bug-compatible (src/util.rs:360-442):
// per coefficient v: while v != 0 { r0 = v % b; r = r0 > b/2 ? (b - r0) % b : r0; push r; v = (v - r)/b }
// digits are non-negative (the centred sign is lost), digits past `exp` dropped.
// Output element (e, digit k) goes to index ((e/group)*exp + k)*group + e%group:
// group = count gives the reference's [exp][count] layout, group = 1 gives
// [count][exp], group = kappa gives [i][k][rho] for the t_i decomposition.
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

// The library-internal form (16-bit digits, base < 2^16 <= ... <= q) four coefficients per thread: 32 bytes in, one 8-byte
// store per digit (the one-word-per-thread kernel above is bound by its 2-byte stores and its block count).
__global__ void __launch_bounds__(256)
k_decompose_u16x4(const JobPtrs outp, const JobPtrs inp, size_t nquads, uint64_t q, uint32_t b32, int exp, size_t group) {
  uint16_t* __restrict__ out = static_cast<uint16_t*>(outp.p[blockIdx.y]);
  const ulonglong2* __restrict__ in = static_cast<const ulonglong2*>(inp.p[blockIdx.y]);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const uint64_t half = (b32 / 2) % q;
  const uint32_t bm = (uint32_t)(0x100000000ull / b32);
  for (; i < nquads; i += stride) {
    const ulonglong2 lo = in[2 * i], hi = in[2 * i + 1];
    uint64_t v[4] = {lo.x, lo.y, hi.x, hi.y};
    const size_t w0 = i * 4;
    const uint32_t e = (uint32_t)(w0 / D), g = (uint32_t)group, eg = e / g;
    uint16_t* const o = out + ((size_t)eg * exp * g + (e - eg * g)) * D + w0 % D;
    for (int k = 0; k < exp; k++) {
      uint32_t r[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        // v = qq*b + r0.  The reference continues with (v - r) / b, r = r0 or b - r0: that is qq in both cases (r0 > b/2 gives
        // (qq - 1)*b + 2*r0 with b < 2*r0 < 2b), and v - r never wraps mod q here (base <= q): one division per digit
        uint64_t qq;
        const uint32_t r0 = divmod_u40_u16((uint32_t)(v[c] >> 32), (uint32_t)v[c], b32, bm, &qq);
        r[c] = r0 > half ? b32 - r0 : r0;
        v[c] = qq;
      }
      *reinterpret_cast<uint2*>(o + (size_t)k * g * D) = make_uint2(r[0] | (r[1] << 16), r[2] | (r[3] << 16));
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
  if (out_u16 && fast) {
    const size_t nquads = nwords / 4;
    unsigned b4 = (unsigned)((nquads + 255) / 256);
    if (b4 > 8192) b4 = 8192;
    hipLaunchKernelGGL(k_decompose_u16x4, dim3(b4, (unsigned)njobs), dim3(256), 0, s, out, in, nquads, q, (uint32_t)base, exp, group);
    return hipGetLastError();
  }
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
// src/proofgen.rs:445-453, this path reads a quarter of the int8 form's bytes).
// In the int8 kernel above every entry costs a byte extract and four 24-bit multiply-adds (two for the projection, two
// for W) plus its share of a wave reduction per row: 5 VALU instructions per entry, 34e9 entries per 64-proof batch —
// the pass is bound by VALU issue, not by its stream, and a denser storage alone buys nothing.  Here
//   * W[col] = sum_m omega[m] Pi[m][col] runs on the matrix cores: omega in 7-bit limbs is the A operand (row = limb),
//     a 64-row x 16-column tile of Pi expanded to int8 the B operand (v_mfma_i32_16x16x64_i8: D[limb][col], exact in
//     int32: 256 rows x 127), 4 k-steps per 16 columns, 4 accumulator registers per column tile;
//   * the storage is TILED so that one dword per lane is that B operand: lane (kq = l >> 4, cc = l & 15) of tile
//     (i, s, ct) holds rows 64s + 16kq + 0..15 of column 16ct + cc, row 4j + k in the 2-bit field at bit 8k + 2j, value
//     entry + 1 in {0, 1, 2}: four (shift, and) pairs give four selector dwords and v_perm_b32 with the byte table
//     {0xFF, 0x00, 0x01} turns each into four int8 rows — 11 instructions for 16 entries;
//   * the projection keeps the VALU (one extract and one unsigned 24-bit multiply-add per entry when the witness is
//     inside its norm bound, SMALL; two otherwise), but a lane now owns 16 ROWS of one column, so its 16 row sums
//     accumulate over the wave's CT column tiles in registers and are reduced across the 16 lanes of a DPP row once per
//     64 x 16CT block of entries instead of once per row.
// ~3 (4) instead of 5.3 instructions per entry.  The surplus sum_col s[col] that the +1 of the fields adds to the row
// sums is taken off once per block.
// Word address: ((i*4 + s) * (cols/16) + ct) * 64 + lane.  Block = 4 waves x CT column tiles of witness column i.
constexpr int PI_LIMBS = 6;                        // 7-bit limbs of omega (< 2^42)

// TINY (a SMALL variant): every witness coefficient of the job fits 16 bits (flag from the witness transform's pass).  The
// projection then takes TWO entries per multiply-add: byte kk of the dwords of two neighbouring column tiles side by side
// (one v_perm_b32 per pair and kk), field j of both masked in place (one v_and_b32) and v_dot2_u32_u16 against the two
// coefficients packed as halves — 4 + 16 + 16 = 36 instructions for 32 entries instead of 64.  The row sums carry the factor
// 4^j of the in-place fields (taken off after the lane reduction).  `flags` / `want`: the launcher starts the TINY and the plain
// SMALL kernel back to back; a block leaves at once when its job's flag says the other one serves it.
template <bool SMALL, int CT, bool TINY = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SMALL ? 3 : 2, SMALL ? 3 : 2)))
k_pi_mfma(const JobPtrs projp, const JobPtrs Wp, const JobPtrs Pip, const JobPtrs Sp, const JobPtrs omegap, unsigned n, unsigned r,
          unsigned ib, uint64_t q, const unsigned* __restrict__ flags = nullptr, unsigned want = 0) {
  static_assert(!TINY || (SMALL && CT % 2 == 0), "TINY is a SMALL variant over pairs of column tiles");
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
  if (flags && (flags[blockIdx.z] != 0) != (want != 0)) return;
  unsigned long long* __restrict__ proj = static_cast<unsigned long long*>(projp.p[blockIdx.z]);
  uint64_t* __restrict__ W = static_cast<uint64_t*>(Wp.p[blockIdx.z]);
  const uint32_t* __restrict__ Pi = static_cast<const uint32_t*>(Pip.p[blockIdx.z]);
  const uint64_t* __restrict__ S = static_cast<const uint64_t*>(Sp.p[blockIdx.z]);
  const uint64_t* __restrict__ omega = static_cast<const uint64_t*>(omegap.p[blockIdx.z]);
  __shared__ u32x4_t om_op[4][64];                 // A operand of k-step s: lane (limb = l & 15, kq = l >> 4): limb of omega[64s + 16kq + 0..15]
  // row sums.  SMALL: the 16 partial sums of a k-step (< 2^29 each) are added across groups of 4 lanes (two DPP steps) and
  // parked at [wave][s][lane / 4][16 rows]; they are added up across the 4 groups of a row group and the 4 waves once, at
  // the end (16 KB instead of 64: three blocks per CU — the waves of this kernel sit in s_waitcnt half of their time) (LDS atomics into one shared area, and
  // several witness columns per block to share it, were both slower).  Otherwise: per-wave totals from DPP reductions.
  __shared__ u32x4_t parked[SMALL ? 4 * 4 * 16 * 4 : 1];    // (after two DPP steps: one lane in four parks)
  __shared__ long long rowsum[SMALL ? 1 : 4][SMALL ? 1 : 256];
  __shared__ long long wstage_own[SMALL ? 1 : 4 * CT * 32];   // W halves of a wave's columns on their way to the lanes that finish them
  static_assert(!SMALL || sizeof(long long) * 4 * CT * 32 <= sizeof(u32x4_t) * 4 * 4 * 16 * 4, "W staging fits the parking area");
  __shared__ long long ssum[4];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cc = lane & 15, kq = lane >> 4;
  {
    const int s = t >> 6, l = t & 63, limb = l & 15, g = l >> 4;
    u32x4_t v = {0, 0, 0, 0};
    if (limb < PI_LIMBS) {
#pragma unroll
      for (int b = 0; b < 16; b++) {
        const uint32_t d = (uint32_t)(omega[64 * s + 16 * g + b] >> (7 * limb)) & 127u;
        v[b >> 2] |= d << (8 * (b & 3));
      }
    }
    om_op[s][l] = v;
  }
  __syncthreads();
  const size_t cols = (size_t)n * D, ctiles = cols / 16;
  const size_t ct0 = ((size_t)blockIdx.x * 4 + wave) * CT;             // this wave's first column tile
  long long my_s = 0;
  const unsigned i = blockIdx.y;
  (void)ib;
  {
    // witness coefficients of this lane's column in each of its tiles (tiles past the end: 0, nothing stored)
    unsigned s_lo[CT], s_hi[SMALL ? 1 : CT];
#pragma unroll
    for (int c = 0; c < CT; c++) {
      const size_t col = (ct0 + c) * 16 + cc;
      uint64_t v = 0;
      if (ct0 + c < ctiles) v = S[((col / D) * r + i) * D + (col % D)];
      if (SMALL) { v &= 0xFFFFFF; s_lo[c] = (unsigned)v; }
      else { s_lo[c] = (unsigned)(v & 0xFFFFF); s_hi[SMALL ? 0 : c] = (unsigned)(v >> 20); }
      if (kq == 0) my_s += (long long)v;            // each column once per wave
    }
    unsigned s_pair[TINY ? CT / 2 : 1];              // TINY: the coefficients of column tiles 2c, 2c + 1 as 16-bit halves
    if constexpr (TINY) {
#pragma unroll
      for (int c = 0; c < CT / 2; c++) s_pair[c] = (s_lo[2 * c] & 0xFFFFu) | (s_lo[2 * c + 1] << 16);
    }
    i32x4_t acc[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) acc[c] = i32x4_t{0, 0, 0, 0};
    // tiles of k-step s: consecutive column tiles are consecutive 256-byte blocks; past the end the last tile is read again
    const uint32_t* base = Pi + (size_t)i * 4 * ctiles * 64;
    auto tile_ptr = [&](int s, int c) { return base + ((size_t)s * ctiles + min(ct0 + c, ctiles - 1)) * 64 + lane; };
    // the CT tiles of the next k-step are requested while this one is consumed (after the last k-step: a harmless re-read)
    uint32_t cur[CT], nxt[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) cur[c] = __builtin_nontemporal_load(tile_ptr(0, c));
#pragma unroll 1
    for (int s = 0; s < 4; s++) {
      const int sn = s + (s < 3 ? 1 : 0);
#pragma unroll
      for (int c = 0; c < CT; c++) nxt[c] = __builtin_nontemporal_load(tile_ptr(sn, c));
      const u32x4_t ao = om_op[s][lane];
      const i32x4_t af = {(int)ao.x, (int)ao.y, (int)ao.z, (int)ao.w};
      unsigned p_lo[16], p_hi[SMALL ? 1 : 16];
#pragma unroll
      for (int k = 0; k < 16; k++) p_lo[k] = 0;
      if (!SMALL) {
#pragma unroll
        for (int k = 0; k < 16; k++) p_hi[SMALL ? 0 : k] = 0;
      }
#pragma unroll
      for (int c = 0; c < CT; c++) {
        const uint32_t w = cur[c];
        // 2-bit fields -> int8 rows: selector dword j = fields j of the four bytes, table {0: -1, 1: 0, 2: +1}
        i32x4_t bf;
#pragma unroll
        for (int j = 0; j < 4; j++) bf[j] = (int)__builtin_amdgcn_perm(0u, 0x020100FFu, (w >> (2 * j)) & 0x03030303u);
        acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf, acc[c], 0, 0, 0);
        if constexpr (TINY) {
          if (c & 1) {                                // tiles c - 1, c together
            const uint32_t w0 = cur[c - 1];
            const u16x2_t sp = __builtin_bit_cast(u16x2_t, s_pair[c / 2]);
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
              // byte kk of w0 in byte 0, byte kk of w in byte 2 (selector 0x0C: a zero byte)
              const uint32_t both = __builtin_amdgcn_perm(w, w0, 0x0C000C00u | ((4u + kk) << 16) | (unsigned)kk);
#pragma unroll
              for (int j = 0; j < 4; j++) {           // row 4j + kk: field j of both bytes, in place (factor 4^j)
                const u16x2_t g2 = __builtin_bit_cast(u16x2_t, both & (0x00030003u << (2 * j)));
                p_lo[4 * j + kk] = __builtin_amdgcn_udot2(g2, sp, p_lo[4 * j + kk], false);
              }
            }
          }
        } else {
        const unsigned sl = s_lo[c], sh = SMALL ? 0u : s_hi[SMALL ? 0 : c];
#pragma unroll
        for (int k = 0; k < 16; k++) {             // row 4j + kk of this lane's block sits at bit 8kk + 2j
          const unsigned g = (w >> (8 * (k & 3) + 2 * (k >> 2))) & 3u;
          p_lo[k] = __umul24(g, sl) + p_lo[k];
          if (!SMALL) p_hi[SMALL ? 0 : k] = __umul24(g, sh) + p_hi[SMALL ? 0 : k];
        }
        }
      }
#pragma unroll
      for (int c = 0; c < CT; c++) cur[c] = nxt[c];
      if constexpr (SMALL) {
#pragma unroll
        for (int k = 0; k < 16; k++) {             // sums of 4 neighbouring lanes (< 2^31) in lanes 3, 7, 11, ...
          unsigned x = p_lo[k];
          x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
          x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
          p_lo[k] = TINY ? x >> (2 * (k >> 2)) : x;  // TINY: row 4j + kk carries the factor 4^j (sums < 2 * 2^6 * 2^16 * CT * 4 < 2^31)
        }
        if ((lane & 3) == 3) {
          u32x4_t* dst = parked + (((size_t)wave * 4 + s) * 16 + (lane >> 2)) * 4;
#pragma unroll
          for (int k4 = 0; k4 < 4; k4++) dst[k4] = u32x4_t{p_lo[4 * k4], p_lo[4 * k4 + 1], p_lo[4 * k4 + 2], p_lo[4 * k4 + 3]};
        }
      } else {
        // row sums of this k-step across the 16 lanes of each DPP row (row_shr 1, 2, 4, 8): the totals of rows
        // 64s + 16kq + k are valid in lane 15 of row kq
#pragma unroll
        for (int k = 0; k < 16; k++) {
          unsigned a = p_lo[k], b = p_hi[SMALL ? 0 : k];
#pragma unroll
          for (int st = 0; st < 2; st++) {
            unsigned x = st ? b : a;
            x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
            x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
            x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
            x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
            if (st) b = x; else a = x;
          }
          if (cc == 15) rowsum[wave][64 * s + 16 * kq + k] = (long long)a + ((long long)b << 20);
        }
      }
    }
    // (acc is finished after the row sums of the block, below: its staging area aliases the parking area)
    {  // this wave's sum of witness coefficients (< 256 * 2^40): 20-bit pieces summed across the wave
      const int p0 = wave_sum_to_lane63((int)(my_s & 0xFFFFF)), p1 = wave_sum_to_lane63((int)((my_s >> 20) & 0xFFFFF)),
                p2 = wave_sum_to_lane63((int)((my_s >> 40) & 0xFFFFF));
      if (lane == 63) ssum[wave] = (long long)p0 + ((long long)p1 << 20) + ((long long)p2 << 40);
    }
    __syncthreads();
    {
      const int m = threadIdx.x;
      long long tot = 0;
      if constexpr (SMALL) {   // row m = 64s + 16kq + k: the k-th dword of lane groups 4kq .. 4kq+3 of k-step s, every wave
        const int sm = m >> 6, kqm = (m >> 4) & 3, km = m & 15;
        const uint32_t* pw = reinterpret_cast<const uint32_t*>(parked);
        for (int wv = 0; wv < 4; wv++)
          for (int g4 = 0; g4 < 4; g4++) tot += pw[((((size_t)wv * 4 + sm) * 16 + 4 * kqm + g4) * 4) * 4 + km];
      } else {
        tot = rowsum[0][m] + rowsum[1][m] + rowsum[2][m] + rowsum[3][m];
      }
      tot -= ssum[0] + ssum[1] + ssum[2] + ssum[3];
      if (tot != 0) atomicAdd(&proj[m], (unsigned long long)tot);
    }
    __syncthreads();                                 // the parking area is free again
    // W: D[limb][col] of tile c sits in register limb & 3 of lane (limb >> 2) * 16 + col: limbs 0..3 in lanes 0..15,
    // limbs 4, 5 in lanes 16..31 (the matrix-core operand holds the entries themselves, -1/0/+1: no surplus here).  The
    // two halves of every column go through LDS so that all 64 lanes share the reductions mod q and the stores: lane j
    // finishes columns j, j + 64, ... of the wave's 16 CT.  
    long long* wpart = (SMALL ? reinterpret_cast<long long*>(parked) : wstage_own) + (size_t)wave * CT * 32;
    if (kq < 2) {
#pragma unroll
      for (int c = 0; c < CT; c++) {
        // |acc| <= 256 * 127 < 2^15: three limbs (kq = 0) and two limbs (kq = 1) combine in 32 bits
        const int low = acc[c][0] + (acc[c][1] << 7) + (kq == 0 ? acc[c][2] << 14 : 0);
        const long long part = kq == 0 ? (long long)low + ((long long)acc[c][3] << 21) : (long long)low << 28;
        wpart[(c * 2 + kq) * 16 + cc] = part;
      }
    }
    __syncthreads();
    const double inv_q = 1.0 / (double)q;
#pragma unroll 1
    for (int jj = 0; jj < CT / 4; jj++) {
      const int cw = lane + 64 * jj, c = cw >> 4, c16 = cw & 15;
      if (ct0 + c < ctiles) {
        const long long v = wpart[(c * 2) * 16 + c16] + wpart[(c * 2 + 1) * 16 + c16];     // sum_m omega[m] * Pi[m][col], |v| < 2^49
        const uint64_t mag = mod_u64_near((uint64_t)(v < 0 ? -v : v), q, inv_q);
        const size_t col = (ct0 + c) * 16 + c16, x = col / D, k = col % D;
        // coefficient k of the column lands at D - k with its sign flipped (k = 0 stays): one sign for both
        const bool minus = (v < 0) != (k != 0);
        W[(x * r + i) * D + (k == 0 ? 0 : D - k)] = (minus && mag != 0) ? q - mag : mag;
      }
    }
  }
}

// Pi int8 [mats][256][cols] in {-1, 0, 1} -> the tiled 2-bit form of k_pi_mfma; one thread per output dword
__global__ void k_pi_pack(uint32_t* __restrict__ dst, const int8_t* __restrict__ src, size_t mats, size_t cols) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t ctiles = cols / 16, per_mat = 4 * ctiles * 64;
  if (k >= mats * per_mat) return;
  const size_t i = k / per_mat, w = k % per_mat;
  const size_t s = w / (ctiles * 64), ct = (w / 64) % ctiles, lane = w % 64;
  const size_t row0 = 64 * s + 16 * (lane >> 4), col = 16 * ct + (lane & 15);
  uint32_t o = 0;
#pragma unroll
  for (int t = 0; t < 16; t++) {
    const int e = src[(i * 256 + row0 + t) * cols + col];
    o |= (uint32_t)((e + 1) & 3) << (8 * (t & 3) + 2 * (t >> 2));
  }
  dst[k] = o;
}

hipError_t launch_pi_pack(uint8_t* dst, const int8_t* src, size_t mats, size_t cols, hipStream_t s) {
  if (cols % 16) return hipErrorInvalidValue;
  const size_t nd = mats * 256 * cols / 16;
  if (nd == 0) return hipSuccess;
  hipLaunchKernelGGL(k_pi_pack, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, reinterpret_cast<uint32_t*>(dst), src, mats, cols);
  return hipGetLastError();
}

// small_witness: witness coefficients are known to be below 2^24
hipError_t launch_pi_fused2_jobs(const JobPtrs& proj /*zeroed*/, const JobPtrs& W, const JobPtrs& Pi2, const JobPtrs& S,
                                 const JobPtrs& omega, size_t n, size_t r, uint64_t q, size_t njobs, bool small_witness, hipStream_t s,
                                 const unsigned* above16) {
  if (n == 0 || r == 0 || njobs == 0) return hipSuccess;
  constexpr int CT = 16;
  const size_t cols = n * D, ctiles = cols / 16;
  const size_t xblocks = (ctiles + 4 * CT - 1) / (4 * CT);
  const unsigned ib = 1;     // witness columns per block (more than one would need the row sums accumulated: slower, see the kernel)
  const dim3 g((unsigned)xblocks, (unsigned)((r + ib - 1) / ib), (unsigned)njobs);
  if (small_witness && above16) {   // per job: the 16-bit form or the 24-bit form (the block of the other kernel returns at once)
    hipLaunchKernelGGL((k_pi_mfma<true, CT, true>), g, dim3(256), 0, s, proj, W, Pi2, S, omega, (unsigned)n, (unsigned)r, ib, q, above16, 0u);
    hipLaunchKernelGGL((k_pi_mfma<true, CT>), g, dim3(256), 0, s, proj, W, Pi2, S, omega, (unsigned)n, (unsigned)r, ib, q, above16, 1u);
  } else if (small_witness)
    hipLaunchKernelGGL((k_pi_mfma<true, CT>), g, dim3(256), 0, s, proj, W, Pi2, S, omega, (unsigned)n, (unsigned)r, ib, q);
  else
    hipLaunchKernelGGL((k_pi_mfma<false, CT>), g, dim3(256), 0, s, proj, W, Pi2, S, omega, (unsigned)n, (unsigned)r, ib, q);
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
