// linalg_kernels.hip — transform-domain vector/matrix kernels: pointwise ring
// ops and the streaming matrix-vector product that dominates the prover.
// gfx950 only.
//
// Replaces the loop nests built on `polynomial_vec_inner_product`
// (src/util.rs:496-509) in Prover::proof_gen (src/proofgen.rs:41-49, 101-132):
// the reference transforms both operands of every product and accumulates in
// the coefficient domain; here operands are transformed once, products are
// accumulated per NTT slot in 64-bit registers and reduced once per output.
#include "kernels.h"

namespace lab {

// ---------------------------------------------------------------- pointwise
template <int OP>
__global__ void __launch_bounds__(256)
k_pointwise(uint32_t* __restrict__ out, size_t out_ps, const uint32_t* __restrict__ a, size_t a_ps,
            const uint32_t* __restrict__ b, size_t b_ps, size_t nquads, const LimbTable* __restrict__ tabs) {
  const int limb = blockIdx.y;
  const uint32_t p = tabs[limb].p, pinv = tabs[limb].pinv, r2 = tabs[limb].r2;
  const uint4* A = reinterpret_cast<const uint4*>(a + (size_t)limb * a_ps);
  const uint4* B = reinterpret_cast<const uint4*>(b + (size_t)limb * b_ps);
  uint4* O = reinterpret_cast<uint4*>(out + (size_t)limb * out_ps);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nquads; i += stride) {
    const uint4 x = A[i], y = B[i];
    uint4 r;
    if (OP == PW_MUL) {  // (x*y*R^-1) * R^2 * R^-1 = x*y
      r.x = mont_mul(mont_mul(x.x, y.x, p, pinv), r2, p, pinv);
      r.y = mont_mul(mont_mul(x.y, y.y, p, pinv), r2, p, pinv);
      r.z = mont_mul(mont_mul(x.z, y.z, p, pinv), r2, p, pinv);
      r.w = mont_mul(mont_mul(x.w, y.w, p, pinv), r2, p, pinv);
    } else if (OP == PW_MUL_KEEP_SCALE) {
      r.x = mont_mul(x.x, y.x, p, pinv); r.y = mont_mul(x.y, y.y, p, pinv);
      r.z = mont_mul(x.z, y.z, p, pinv); r.w = mont_mul(x.w, y.w, p, pinv);
    } else if (OP == PW_ADD) {
      r.x = csub(x.x + y.x, p); r.y = csub(x.y + y.y, p); r.z = csub(x.z + y.z, p); r.w = csub(x.w + y.w, p);
    } else {
      r.x = csub(x.x + p - y.x, p); r.y = csub(x.y + p - y.y, p);
      r.z = csub(x.z + p - y.z, p); r.w = csub(x.w + p - y.w, p);
    }
    O[i] = r;
  }
}

hipError_t launch_pointwise(PointwiseOp op, HatView out, ConstHatView a, ConstHatView b, size_t count, int nlimbs,
                            const LimbTable* tabs, hipStream_t s) {
  if (count == 0) return hipSuccess;
  size_t nquads = count * D / 4;
  unsigned blocks = (unsigned)((nquads + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  dim3 grid(blocks, (unsigned)nlimbs);
#define LAUNCH_PW(OPV)                                                                                              \
  hipLaunchKernelGGL(k_pointwise<OPV>, grid, dim3(256), 0, s, out.base, out.plane_stride, a.base, a.plane_stride, \
                     b.base, b.plane_stride, nquads, tabs)
  switch (op) {
    case PW_MUL: LAUNCH_PW(PW_MUL); break;
    case PW_ADD: LAUNCH_PW(PW_ADD); break;
    case PW_SUB: LAUNCH_PW(PW_SUB); break;
    case PW_MUL_KEEP_SCALE: LAUNCH_PW(PW_MUL_KEEP_SCALE); break;
  }
#undef LAUNCH_PW
  return hipGetLastError();
}

// ------------------------------------------------------ streaming mat-vec
// partial[m][rho][j] = sum_c M[m][rho][c][j] * V[m][c][j]   (mod p, per limb)
//
// The HBM-bound kernel of the prover (outer commitment, src/proofgen.rs:101-132;
// single right-hand side per matrix).  Layout: a matrix row is `cols`
// polynomials = cols*256 B contiguous.  A block of 4 waves walks MV_ROWS rows
// at once so the vector chunk it loads (from L2) is reused MV_ROWS times; each
// lane loads 16 B (4 NTT slots of one polynomial), a wave instruction covers 4
// consecutive polynomials (1 KiB) and the 4 waves of the block cover 4 KiB of
// every row per step.  Products accumulate in 64-bit registers (v_mad_u64_u32),
// the high word is folded every 8 terms, and one Montgomery reduction per
// output slot finishes the sum.
// Algorithmic traffic per launch: nmat*rows*cols*256 B (matrix) per limb.
constexpr int MV_ROWS = 4;
constexpr int MV_WAVES = 4;

__global__ void __launch_bounds__(64 * MV_WAVES)
k_matvec_partial(uint32_t* __restrict__ partial, size_t partial_ps, const uint32_t* __restrict__ M, size_t m_ps,
                 const uint32_t* __restrict__ V, size_t v_ps, unsigned rows, unsigned cols,
                 const LimbTable* __restrict__ tabs) {
  __shared__ uint32_t red[MV_WAVES][MV_ROWS][D];
  const int limb = blockIdx.z;
  const unsigned mat = blockIdx.y;
  const unsigned row0 = blockIdx.x * MV_ROWS;
  const uint32_t p = tabs[limb].p, p2 = tabs[limb].p2, pinv = tabs[limb].pinv, r2 = tabs[limb].r2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 4;          // polynomial within the 4-poly chunk
  const int jq = lane & 15;           // slots 4*jq .. 4*jq+3

  const uint4* Vp = reinterpret_cast<const uint4*>(V + (size_t)limb * v_ps + (size_t)mat * cols * D);
  const uint4* Mp[MV_ROWS];
#pragma unroll
  for (int r = 0; r < MV_ROWS; r++) {
    unsigned row = min(row0 + r, rows - 1);  // clamp: out-of-range rows recompute the last row and are not stored
    Mp[r] = reinterpret_cast<const uint4*>(M + (size_t)limb * m_ps + ((size_t)mat * rows + row) * cols * D);
  }

  uint64_t acc[MV_ROWS][4];
#pragma unroll
  for (int r = 0; r < MV_ROWS; r++)
#pragma unroll
    for (int k = 0; k < 4; k++) acc[r][k] = 0;

  // chunk index ch covers polynomials 4*ch .. 4*ch+3; this lane's polynomial is 4*ch+sub.
  // Fast path: 8 full chunks per accumulator fold, loads of 4 chunks x (MV_ROWS+1) streams in flight.
  const unsigned full_chunks = cols / 4;
  const unsigned nchunks = (cols + 3) / 4;
  unsigned ch = wave;
  while (ch + 7u * MV_WAVES < full_chunks) {
#pragma unroll
    for (int half = 0; half < 2; half++) {
      uint4 v[4], m[4][MV_ROWS];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const size_t off = (size_t)((ch + u * MV_WAVES) * 4 + sub) * 16 + jq;  // uint4 index of (poly, 4 slots)
        v[u] = Vp[off];
#pragma unroll
        for (int r = 0; r < MV_ROWS; r++) m[u][r] = Mp[r][off];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int r = 0; r < MV_ROWS; r++) {
          acc[r][0] += (uint64_t)m[u][r].x * v[u].x;
          acc[r][1] += (uint64_t)m[u][r].y * v[u].y;
          acc[r][2] += (uint64_t)m[u][r].z * v[u].z;
          acc[r][3] += (uint64_t)m[u][r].w * v[u].w;
        }
      ch += 4 * MV_WAVES;
    }
#pragma unroll
    for (int r = 0; r < MV_ROWS; r++)
#pragma unroll
      for (int k = 0; k < 4; k++) acc[r][k] = acc_fold(acc[r][k], p, p2);
  }
  // tail: at most 8 chunks are left for this wave (the last may be partial)
  for (; ch < nchunks; ch += MV_WAVES) {
    const unsigned c = ch * 4 + sub;
    if (c < cols) {
      const size_t off = (size_t)c * 16 + jq;
      const uint4 v = Vp[off];
#pragma unroll
      for (int r = 0; r < MV_ROWS; r++) {
        const uint4 m = Mp[r][off];
        acc[r][0] += (uint64_t)m.x * v.x;
        acc[r][1] += (uint64_t)m.y * v.y;
        acc[r][2] += (uint64_t)m.z * v.z;
        acc[r][3] += (uint64_t)m.w * v.w;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < MV_ROWS; r++)
#pragma unroll
    for (int k = 0; k < 4; k++) acc[r][k] = acc_fold(acc[r][k], p, p2);

  // per-lane partials -> canonical residues (scale 2^-32), then sum the 4 lane groups of the wave
#pragma unroll
  for (int r = 0; r < MV_ROWS; r++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t x = redc64(acc[r][k], p, pinv);
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);             // < 4p < 2^32
      x = csub(csub(x, p2), p);
      if (sub == 0) red[wave][r][jq * 4 + k] = x;
    }
  }
  __syncthreads();
  // 256 threads: thread t finishes output (row = t>>6, slot = t&63)
  {
    const int r = threadIdx.x >> 6, j = threadIdx.x & 63;
    if (r < MV_ROWS && row0 + r < rows) {
      uint32_t x = red[0][r][j];
#pragma unroll
      for (int w = 1; w < MV_WAVES; w++) x += red[w][r][j];
      x = csub(csub(x, p2), p);
      x = mont_mul(x, r2, p, pinv);       // undo the 2^-32 of redc64: plain residue
      partial[(size_t)limb * partial_ps + ((size_t)mat * rows + row0 + r) * D + j] = x;
    }
  }
}

const char* matvec_kernel_name() { return "k_matvec_partial"; }

hipError_t launch_matvec_partial(HatView partial, ConstHatView M, ConstHatView V, size_t nmat, size_t rows,
                                 size_t cols, int nlimbs, const LimbTable* tabs, hipStream_t s) {
  if (nmat == 0 || rows == 0) return hipSuccess;
  dim3 grid((unsigned)((rows + MV_ROWS - 1) / MV_ROWS), (unsigned)nmat, (unsigned)nlimbs);
  hipLaunchKernelGGL(k_matvec_partial, grid, dim3(64 * MV_WAVES), 0, s, partial.base, partial.plane_stride, M.base,
                     M.plane_stride, V.base, V.plane_stride, (unsigned)rows, (unsigned)cols, tabs);
  return hipGetLastError();
}

// out[rho][j] = sum_m partial[m][rho][j] mod p
__global__ void __launch_bounds__(256)
k_sum_terms(uint32_t* __restrict__ out, size_t out_ps, const uint32_t* __restrict__ partial, size_t partial_ps,
            unsigned nterms, size_t row_words, const LimbTable* __restrict__ tabs) {
  const int limb = blockIdx.y;
  const uint32_t p = tabs[limb].p;
  const uint32_t* P = partial + (size_t)limb * partial_ps;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < row_words; i += stride) {
    uint64_t acc = 0;
    for (unsigned m = 0; m < nterms; m++) acc += P[(size_t)m * row_words + i];
    out[(size_t)limb * out_ps + i] = (uint32_t)(acc % p);
  }
}

hipError_t launch_sum_terms(HatView out, ConstHatView partial, size_t nterms, size_t rows, int nlimbs,
                            const LimbTable* tabs, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  size_t row_words = rows * D;
  unsigned blocks = (unsigned)((row_words + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_sum_terms, dim3(blocks, (unsigned)nlimbs), dim3(256), 0, s, out.base, out.plane_stride,
                     partial.base, partial.plane_stride, (unsigned)nterms, row_words, tabs);
  return hipGetLastError();
}

}  // namespace lab
