// wide_kernels.hip — the sum-of-products kernel of the single-limb strict-32 path (zq_device.h, Zp<true>): an NTT-friendly
// q in [2^30, 2^32) as its own prime, 4 bytes per coefficient, no CRT (path (b); reference anchor: the
// commented 32-bit Q of src/constants.rs:195-196 behind the seam src/algebraic.rs:379-404).
//
// Every transform-domain reduction of the path — inner products, the streaming mat-vecs of the outer commitments
// (src/proofgen.rs:101-153, 364-378), the Ajtai product (:41-49), the Gram products (:59-70, 258-277) — is
//     out[o][j] = sum_{k < K} A[abase(o) + k*ak + j] * B[bbase(o) + k*bk + j]   mod q,   j = 0..63,
// for a five-level output index o with per-level strides into out, A and B.  The launchers of linalg_kernels.hip describe
// their operation in this form (WideDot) when the context's tables are `wide` and call launch_wide_dot instead of their
// lazy-arithmetic kernels.  One wave per output polynomial: a lane owns 4 slots of one polynomial in 4 (16-byte loads,
// a wave instruction covers 4 consecutive terms), products go to two 64-bit accumulators per slot (low and high words:
// exact for 2^32 terms, nothing to fold), one exact reduction per output.
#include "kernels.h"

namespace lab {

__global__ void __launch_bounds__(256)
k_wide_dot(const WideDot d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 4, jq = lane & 15;
  const size_t total = (size_t)d.n[0] * d.n[1] * d.n[2] * d.n[3] * d.n[4];
  const size_t o = (size_t)blockIdx.x * 4 + wave;
  if (o >= total) return;                              // wave-uniform; no barrier below
  unsigned idx[5];
  {
    size_t t = o;
#pragma unroll
    for (int l = 4; l >= 0; l--) { idx[l] = (unsigned)(t % d.n[l]); t /= d.n[l]; }
  }
  size_t oo = 0, ao = 0, bo = 0;
#pragma unroll
  for (int l = 0; l < 5; l++) { oo += idx[l] * d.os[l]; ao += idx[l] * d.as[l]; bo += idx[l] * d.bs[l]; }
  unsigned K = d.K;
  if (d.chunk_level >= 0) {                            // this output is one chunk of a Ktotal-term sum
    const unsigned c = idx[d.chunk_level], beg = c * d.K;
    K = beg >= d.Ktotal ? 0 : min(d.K, d.Ktotal - beg);
  }
  const ZpConsts c = zp_consts(d.tab);
  const uint4* A = reinterpret_cast<const uint4*>(d.A + ao) + jq;
  const uint4* B = reinterpret_cast<const uint4*>(d.B + bo) + jq;
  const size_t ak4 = d.ak / 4, bk4 = d.bk / 4;
  Zp<true>::acc_t acc[4];
#pragma unroll
  for (int e = 0; e < 4; e++) acc[e] = Zp<true>::zero();
  unsigned k = sub;
  for (; k + 12 < K; k += 16) {                        // four terms of this lane in flight
    uint4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { a[u] = A[(size_t)(k + 4 * u) * ak4]; b[u] = B[(size_t)(k + 4 * u) * bk4]; }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      Zp<true>::mad(acc[0], a[u].x, b[u].x); Zp<true>::mad(acc[1], a[u].y, b[u].y);
      Zp<true>::mad(acc[2], a[u].z, b[u].z); Zp<true>::mad(acc[3], a[u].w, b[u].w);
    }
  }
  for (; k < K; k += 4) {
    const uint4 a = A[(size_t)k * ak4], b = B[(size_t)k * bk4];
    Zp<true>::mad(acc[0], a.x, b.x); Zp<true>::mad(acc[1], a.y, b.y);
    Zp<true>::mad(acc[2], a.z, b.z); Zp<true>::mad(acc[3], a.w, b.w);
  }
  uint32_t x[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    x[e] = Zp<true>::finish(acc[e], c);
    x[e] = Zp<true>::add(x[e], __shfl_xor(x[e], 16), c);
    x[e] = Zp<true>::add(x[e], __shfl_xor(x[e], 32), c);
  }
  if (sub == 0) *(reinterpret_cast<uint4*>(d.out + oo) + jq) = make_uint4(x[0], x[1], x[2], x[3]);
}

hipError_t launch_wide_dot(const WideDot& d, hipStream_t s) {
  size_t total = 1;
  for (int l = 0; l < 5; l++) {
    if (d.n[l] == 0) return hipSuccess;
    total *= d.n[l];
  }
  if ((d.ak % 4) || (d.bk % 4) || total > ((size_t)1 << 33)) return hipErrorInvalidValue;
  for (int l = 0; l < 5; l++)
    if ((d.os[l] % 4) || (d.as[l] % 4) || (d.bs[l] % 4)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_wide_dot, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, d);
  return hipGetLastError();
}

}  // namespace lab
