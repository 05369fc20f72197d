// wide_kernels.hip — the sum-of-products kernel of the single-limb strict-32 path (zq_device.h, Zp<true>): an NTT-friendly
// q in [2^30, 2^32) as its own prime, 4 bytes per coefficient, no CRT (path (b); reference anchor: the
// commented 32-bit Q of src/constants.rs:195-196 behind the seam src/algebraic.rs:379-404).
//
// Every transform-domain reduction of the path — inner products, the streaming mat-vecs of the outer commitments
// (src/proofgen.rs:101-153, 364-378), the Ajtai product (:41-49), the Gram products (:59-70, 258-277) — is
//     out[o][j] = sum_{k < K} A[abase(o) + k*ak + j] * B[bbase(o) + k*bk + j]   mod q,   j = 0..63,
// for a five-level output index o with per-level strides into out, A and B.  The launchers of linalg_kernels.hip describe
// their operation in this form (WideDot) when the context's tables are `wide` and call launch_wide_dot instead of their
// lazy-arithmetic kernels.  One wave per WD_ROWS output polynomials: a lane owns 4 slots of one polynomial in 4 (16-byte
// loads, a wave instruction covers 4 consecutive terms), products go to two 64-bit accumulators per slot (low and high
// words: exact for 2^32 terms, nothing to fold), one exact reduction per output.
#include "kernels.h"

namespace lab {

constexpr int WD_ROWS = 4;   // consecutive values of the fastest output index a wave computes together

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 wd_load(const uint4* p) {
  if constexpr (NT) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  } else {
    return *p;
  }
}

// A wave owns WD_ROWS outputs that differ in the fastest index only: an operand whose stride along that index is 0 (the vector
// of a mat-vec, the P column of a Gram row) is loaded once for all of them.  STREAM: A is read once (a CRS matrix): its loads
// are non-temporal so that they do not evict the re-used B operand.
template <bool STREAM>
__global__ void __launch_bounds__(256)
k_wide_dot(const WideDot d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 4, jq = lane & 15;
  const unsigned n4g = (d.n[4] + WD_ROWS - 1) / WD_ROWS;
  const size_t total = (size_t)d.n[0] * d.n[1] * d.n[2] * d.n[3] * n4g;
  const size_t o = (size_t)blockIdx.x * 4 + wave;
  if (o >= total) return;                              // wave-uniform; no barrier below
  unsigned idx[5];
  {
    size_t t = o;
    idx[4] = (unsigned)(t % n4g) * WD_ROWS; t /= n4g;
#pragma unroll
    for (int l = 3; l >= 0; l--) { idx[l] = (unsigned)(t % d.n[l]); t /= d.n[l]; }
  }
  size_t oo = 0, ao = 0, bo = 0;
#pragma unroll
  for (int l = 0; l < 5; l++) { oo += idx[l] * d.os[l]; ao += idx[l] * d.as[l]; bo += idx[l] * d.bs[l]; }
  unsigned K = d.K;
  if (d.chunk_level >= 0) {                            // this output is one chunk of a Ktotal-term sum
    unsigned c = 0;
#pragma unroll
    for (int l = 0; l < 5; l++) c = l == d.chunk_level ? idx[l] : c;      // (no dynamic register-array index)
    const unsigned beg = c * d.K;
    K = beg >= d.Ktotal ? 0 : min(d.K, d.Ktotal - beg);
  }
  const ZpConsts c = zp_consts(d.tab);
  const unsigned nrow = min((unsigned)WD_ROWS, d.n[4] - idx[4]);          // rows of this wave that exist
  const bool a_shared = d.as[4] == 0, b_shared = d.bs[4] == 0;
  // rows past the end recompute the last one and are not stored
  const uint4* Ab = reinterpret_cast<const uint4*>(d.A + ao) + jq;
  const uint4* Bb = reinterpret_cast<const uint4*>(d.B + bo) + jq;
  size_t ra[WD_ROWS], rb[WD_ROWS];
#pragma unroll
  for (int r = 0; r < WD_ROWS; r++) {
    const unsigned rr = min((unsigned)r, nrow - 1);
    ra[r] = rr * d.as[4] / 4; rb[r] = rr * d.bs[4] / 4;
  }
  const size_t ak4 = d.ak / 4, bk4 = d.bk / 4;
  Zp<true>::acc_t acc[WD_ROWS][4];
#pragma unroll
  for (int r = 0; r < WD_ROWS; r++)
#pragma unroll
    for (int e = 0; e < 4; e++) acc[r][e] = Zp<true>::zero();
#define WD_FETCH(kk, a, b)                                                                              \
  {                                                                                                     \
    if (a_shared) {                                                                                     \
      a[0] = wd_load<STREAM>(Ab + (size_t)(kk) * ak4);                                                  \
      _Pragma("unroll") for (int r = 1; r < WD_ROWS; r++) a[r] = a[0];                                  \
    } else {                                                                                            \
      _Pragma("unroll") for (int r = 0; r < WD_ROWS; r++) a[r] = wd_load<STREAM>(Ab + ra[r] + (size_t)(kk) * ak4); \
    }                                                                                                   \
    if (b_shared) {                                                                                     \
      b[0] = Bb[(size_t)(kk) * bk4];                                                                    \
      _Pragma("unroll") for (int r = 1; r < WD_ROWS; r++) b[r] = b[0];                                  \
    } else {                                                                                            \
      _Pragma("unroll") for (int r = 0; r < WD_ROWS; r++) b[r] = Bb[rb[r] + (size_t)(kk) * bk4];        \
    }                                                                                                   \
  }
#define WD_TERM(a, b)                                                                                   \
  _Pragma("unroll") for (int r = 0; r < WD_ROWS; r++) {                                                 \
    Zp<true>::mad(acc[r][0], a[r].x, b[r].x); Zp<true>::mad(acc[r][1], a[r].y, b[r].y);                \
    Zp<true>::mad(acc[r][2], a[r].z, b[r].z); Zp<true>::mad(acc[r][3], a[r].w, b[r].w);                \
  }
  unsigned k = sub;
  for (; k + 4 < K; k += 8) {                          // two terms of this lane (x WD_ROWS rows) in flight
    uint4 a0[WD_ROWS], b0[WD_ROWS], a1[WD_ROWS], b1[WD_ROWS];
    WD_FETCH(k, a0, b0)
    WD_FETCH(k + 4, a1, b1)
    WD_TERM(a0, b0)
    WD_TERM(a1, b1)
  }
  for (; k < K; k += 4) {
    uint4 a0[WD_ROWS], b0[WD_ROWS];
    WD_FETCH(k, a0, b0)
    WD_TERM(a0, b0)
  }
#undef WD_FETCH
#undef WD_TERM
#pragma unroll
  for (int r = 0; r < WD_ROWS; r++) {
    uint32_t x[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      x[e] = Zp<true>::finish(acc[r][e], c);
      x[e] = Zp<true>::add(x[e], __shfl_xor(x[e], 16), c);
      x[e] = Zp<true>::add(x[e], __shfl_xor(x[e], 32), c);
    }
    if (sub == 0 && (unsigned)r < nrow) *(reinterpret_cast<uint4*>(d.out + oo + r * d.os[4]) + jq) = make_uint4(x[0], x[1], x[2], x[3]);
  }
}

hipError_t launch_wide_dot(const WideDot& d, hipStream_t s) {
  size_t total = 1;
  for (int l = 0; l < 5; l++) {
    if (d.n[l] == 0) return hipSuccess;
    total *= l == 4 ? (d.n[l] + WD_ROWS - 1) / WD_ROWS : d.n[l];
  }
  if ((d.ak % 4) || (d.bk % 4) || total > ((size_t)1 << 33)) return hipErrorInvalidValue;
  for (int l = 0; l < 5; l++)
    if ((d.os[l] % 4) || (d.as[l] % 4) || (d.bs[l] % 4)) return hipErrorInvalidValue;
  if (d.stream_a) hipLaunchKernelGGL(k_wide_dot<true>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, d);
  else hipLaunchKernelGGL(k_wide_dot<false>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, d);
  return hipGetLastError();
}

}  // namespace lab
