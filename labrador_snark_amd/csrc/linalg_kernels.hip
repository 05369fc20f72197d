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
#include <type_traits>
#include <cstdlib>
#include <cstdio>

namespace lab {

// ---------------------------------------------------------------- pointwise
template <int OP, bool WIDE>
__global__ void __launch_bounds__(256)
k_pointwise(uint32_t* __restrict__ out, size_t out_ps, const uint32_t* __restrict__ a, size_t a_ps,
            const uint32_t* __restrict__ b, size_t b_ps, size_t nquads, const LimbTable* __restrict__ tabs, unsigned nlimbs,
            size_t out_js, size_t a_js, size_t b_js) {
  const int limb = blockIdx.y % nlimbs;
  const size_t job = blockIdx.y / nlimbs;
  const uint32_t p = tabs[limb].p, pinv = tabs[limb].pinv, r2 = tabs[limb].r2;
  const uint4* A = reinterpret_cast<const uint4*>(a + (size_t)limb * a_ps + job * a_js);
  const uint4* B = reinterpret_cast<const uint4*>(b + (size_t)limb * b_ps + job * b_js);
  uint4* O = reinterpret_cast<uint4*>(out + (size_t)limb * out_ps + job * out_js);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nquads; i += stride) {
    const uint4 x = A[i], y = B[i];
    uint4 r;
    if constexpr (WIDE) {  // single-limb strict-32 path: plain products (KEEP_SCALE has no scale to keep), carry-corrected sums
      const ZpConsts c = zp_consts(tabs + limb);
      typedef Zp<true> Z;
      if (OP == PW_MUL || OP == PW_MUL_KEEP_SCALE) {
        r.x = Z::mul(x.x, y.x, c); r.y = Z::mul(x.y, y.y, c); r.z = Z::mul(x.z, y.z, c); r.w = Z::mul(x.w, y.w, c);
      } else if (OP == PW_ADD) {
        r.x = Z::add(x.x, y.x, c); r.y = Z::add(x.y, y.y, c); r.z = Z::add(x.z, y.z, c); r.w = Z::add(x.w, y.w, c);
      } else {
        r.x = Z::sub(x.x, y.x, c); r.y = Z::sub(x.y, y.y, c); r.z = Z::sub(x.z, y.z, c); r.w = Z::sub(x.w, y.w, c);
      }
    } else
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
                            const LimbTable* tabs, hipStream_t s, size_t njobs, size_t out_js, size_t a_js, size_t b_js) {
  if (count == 0 || njobs == 0) return hipSuccess;
  size_t nquads = count * D / 4;
  unsigned blocks = (unsigned)((nquads + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  dim3 grid(blocks, (unsigned)(nlimbs * njobs));
  const bool wide = limb_tables_wide(tabs);
#define LAUNCH_PW1(OPV, W)                                                                                              \
  hipLaunchKernelGGL((k_pointwise<OPV, W>), grid, dim3(256), 0, s, out.base, out.plane_stride, a.base, a.plane_stride, \
                     b.base, b.plane_stride, nquads, tabs, (unsigned)nlimbs, out_js, a_js, b_js)
#define LAUNCH_PW(OPV) do { if (wide) LAUNCH_PW1(OPV, true); else LAUNCH_PW1(OPV, false); } while (0)
  switch (op) {
    case PW_MUL: LAUNCH_PW(PW_MUL); break;
    case PW_ADD: LAUNCH_PW(PW_ADD); break;
    case PW_SUB: LAUNCH_PW(PW_SUB); break;
    case PW_MUL_KEEP_SCALE: LAUNCH_PW(PW_MUL_KEEP_SCALE); break;
  }
#undef LAUNCH_PW
#undef LAUNCH_PW1
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
constexpr int MV_WAVES = 8;

// four consecutive NTT slots of a matrix polynomial: 16 B when residues are u32, 8 B when the limb is a
// small prime stored as u16 (the compact third limb of the digit basis)
template <typename MT> struct Quad;
// Streamed-once operands (the CRS matrices) are read with non-temporal loads: they bypass the cache
// allocation that would otherwise evict the re-used vector chunk, and measured +11 % on the mat-vec
// (6.08 -> 6.77 TB/s at nmat=48, 1024 x 1024, 3 limbs).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <> struct Quad<uint32_t> {
  typedef uint4 raw;
  static __device__ __forceinline__ uint4 load(const raw* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  static __device__ __forceinline__ raw load_raw(const raw* p) { return load(p); }
  static __device__ __forceinline__ uint4 unpack(const raw& v) { return v; }
};
template <> struct Quad<uint16_t> {
  typedef uint2 raw;
  static __device__ __forceinline__ uint4 load(const raw* p) {
    const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
    return make_uint4(v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16);
  }
  static __device__ __forceinline__ raw load_raw(const raw* p) {
    const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
    return make_uint2(v.x, v.y);
  }
  static __device__ __forceinline__ uint4 unpack(const raw& v) {
    return make_uint4(v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16);
  }
};

// NB = number of independent right-hand-side sets ("proofs in the batch") that share the streamed matrices:
// the common reference string is read once for all of them.  Vector set b lives at V + b*v_bs, its partials at
// partial + b*partial_bs (word strides inside one limb plane).
template <typename MT, int NB>
__global__ void __launch_bounds__(64 * MV_WAVES)
k_matvec_partial(uint32_t* __restrict__ partial, size_t partial_ps, size_t partial_bs, const MT* __restrict__ M, size_t m_ps,
                 const uint32_t* __restrict__ V, size_t v_ps, size_t v_bs, unsigned rows, unsigned cols,
                 const LimbTable* __restrict__ tabs, unsigned nlimbs, size_t js) {
  typedef typename Quad<MT>::raw mraw;
  constexpr int ROWS = NB >= 4 ? 2 : MV_ROWS;     // keep the accumulator file at <= 64 x u64 per lane
  constexpr int GROUP = NB >= 4 ? 2 : 4;          // chunks whose loads are issued together
  __shared__ uint32_t red[MV_WAVES][ROWS * NB][D];
  const int limb = blockIdx.z % nlimbs;
  {  // independent jobs (own matrix, vector and output, all `js` words apart inside a plane)
    const size_t job = blockIdx.z / nlimbs;
    partial += job * js; M += job * js; V += job * js;
  }
  const unsigned mat = blockIdx.y;
  const unsigned row0 = blockIdx.x * ROWS;
  const uint32_t p = tabs[limb].p, p2 = tabs[limb].p2, pinv = tabs[limb].pinv, r2 = tabs[limb].r2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 4;          // polynomial within the 4-poly chunk
  const int jq = lane & 15;           // slots 4*jq .. 4*jq+3

  const uint4* Vp = reinterpret_cast<const uint4*>(V + (size_t)limb * v_ps + (size_t)mat * cols * D);
  const size_t vb4 = v_bs / 4;
  const mraw* Mp[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    unsigned row = min(row0 + r, rows - 1);  // clamp: out-of-range rows recompute the last row and are not stored
    Mp[r] = reinterpret_cast<const mraw*>(M + (size_t)limb * m_ps + ((size_t)mat * rows + row) * cols * D);
  }

  uint64_t acc[ROWS][NB][4];
#pragma unroll
  for (int r = 0; r < ROWS; r++)
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int k = 0; k < 4; k++) acc[r][b][k] = 0;

  // chunk index ch covers polynomials 4*ch .. 4*ch+3; this lane's polynomial is 4*ch+sub.
  // Fast path: 8 full chunks per accumulator fold, loads of GROUP chunks x (ROWS+NB) streams in flight.
  const unsigned full_chunks = cols / 4;
  const unsigned nchunks = (cols + 3) / 4;
  unsigned ch = wave;
  while (ch + 7u * MV_WAVES < full_chunks) {
#pragma unroll
    for (int part = 0; part < 8 / GROUP; part++) {
      uint4 v[GROUP][NB], m[GROUP][ROWS];
#pragma unroll
      for (int u = 0; u < GROUP; u++) {
        const size_t off = (size_t)((ch + u * MV_WAVES) * 4 + sub) * 16 + jq;  // uint4 index of (poly, 4 slots)
#pragma unroll
        for (int b = 0; b < NB; b++) v[u][b] = Vp[off + b * vb4];
#pragma unroll
        for (int r = 0; r < ROWS; r++) m[u][r] = Quad<MT>::load(Mp[r] + off);
      }
#pragma unroll
      for (int u = 0; u < GROUP; u++)
#pragma unroll
        for (int r = 0; r < ROWS; r++)
#pragma unroll
          for (int b = 0; b < NB; b++) {
            acc[r][b][0] += (uint64_t)m[u][r].x * v[u][b].x;
            acc[r][b][1] += (uint64_t)m[u][r].y * v[u][b].y;
            acc[r][b][2] += (uint64_t)m[u][r].z * v[u][b].z;
            acc[r][b][3] += (uint64_t)m[u][r].w * v[u][b].w;
          }
      ch += GROUP * MV_WAVES;
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[r][b][k] = acc_fold(acc[r][b][k], p, p2);
  }
  // tail: at most 8 chunks are left for this wave (the last may be partial)
  for (; ch < nchunks; ch += MV_WAVES) {
    const unsigned c = ch * 4 + sub;
    if (c < cols) {
      const size_t off = (size_t)c * 16 + jq;
      uint4 v[NB];
#pragma unroll
      for (int b = 0; b < NB; b++) v[b] = Vp[off + b * vb4];
#pragma unroll
      for (int r = 0; r < ROWS; r++) {
        const uint4 m = Quad<MT>::load(Mp[r] + off);
#pragma unroll
        for (int b = 0; b < NB; b++) {
          acc[r][b][0] += (uint64_t)m.x * v[b].x;
          acc[r][b][1] += (uint64_t)m.y * v[b].y;
          acc[r][b][2] += (uint64_t)m.z * v[b].z;
          acc[r][b][3] += (uint64_t)m.w * v[b].w;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; r++)
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int k = 0; k < 4; k++) acc[r][b][k] = acc_fold(acc[r][b][k], p, p2);

  // per-lane partials -> canonical residues (scale 2^-32), then sum the 4 lane groups of the wave
#pragma unroll
  for (int r = 0; r < ROWS; r++)
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t x = redc64(acc[r][b][k], p, pinv);
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);             // < 4p < 2^32
        x = csub(csub(x, p2), p);
        if (sub == 0) red[wave][r * NB + b][jq * 4 + k] = x;
      }
  __syncthreads();
  // finish the ROWS*NB outputs: 64 lanes per output, MV_WAVES outputs at a time
  for (int o = wave; o < ROWS * NB; o += MV_WAVES) {
    const int r = o / NB, b = o % NB, j = lane;
    if (row0 + r < rows) {
      uint32_t x = 0;                     // MV_WAVES canonical residues: reduce after every 4 (4p < 2^32)
#pragma unroll
      for (int w0 = 0; w0 < MV_WAVES; w0 += 4) {
        uint32_t y = 0;
#pragma unroll
        for (int w = w0; w < w0 + 4 && w < MV_WAVES; w++) y += red[w][o][j];
        x = csub(x + csub(csub(y, p2), p), p);
      }
      x = mont_mul(x, r2, p, pinv);       // undo the 2^-32 of redc64: plain residue
      partial[(size_t)limb * partial_ps + (size_t)b * partial_bs + ((size_t)mat * rows + row0 + r) * D + j] = x;
    }
  }
}

// ---- batched mat-vec with the matrix tile staged through LDS ("sets" kernel)
// For NB >= 2 vector sets the register file cannot hold NB x rows accumulators AND enough matrix loads in flight to
// cover HBM latency.  Here a 512-thread block owns ST_ROWS = 16 matrix rows and walks the columns in tiles of
// ST_COLS polynomials: the tile (16 rows x ST_COLS x 256 B) is fetched once with coalesced non-temporal loads,
// parked in LDS (double buffered, next tile prefetched through registers) and read by every thread.
// Thread = (slot quad jq, vector set b, row group rg): it accumulates RG = 16*NB/32 rows x 4 slots for ITS set over
// all columns, so accumulators are RG*4 u64 per lane and no cross-thread reduction exists.  The 4 (or more) lanes
// of a wave that share (row, jq) read the same LDS address (broadcast), the vector chunk comes from L2 once per
// 16 rows.  HBM traffic: the matrix, once per batch.
constexpr int ST_ROWS = 16;
constexpr int ST_COLS = 4;
constexpr int ST_THREADS = 512;
// 16 rows per block, 8 when there are 8 vector sets (keeps rows-per-thread, hence the accumulator file, at 2)
__host__ __device__ constexpr int st_rows(int nb) { return nb >= 8 ? ST_ROWS / 2 : ST_ROWS; }

template <typename MT, int NB>
__global__ void __launch_bounds__(ST_THREADS, 4)
k_matvec_sets(uint32_t* __restrict__ partial, size_t partial_ps, size_t partial_bs, const MT* __restrict__ M, size_t m_ps,
              const uint32_t* __restrict__ V, size_t v_ps, size_t v_bs, unsigned rows, unsigned cols,
              const LimbTable* __restrict__ tabs) {
  typedef typename Quad<MT>::raw mraw;
  constexpr int SR = st_rows(NB);                             // matrix rows per block
  constexpr int RG = SR * NB / 32;                            // rows per thread (2 for NB = 4, 8; 1 for NB = 2)
  constexpr int QPT = SR * ST_COLS * 16 / ST_THREADS;         // matrix quads per thread per tile
  constexpr int VQ = NB * ST_COLS * 16;                       // vector quads per tile (<= ST_THREADS)
  __shared__ mraw tile_m[2][SR * ST_COLS * 16];
  __shared__ uint4 tile_v[2][VQ];
  const int limb = blockIdx.z;
  const unsigned mat = blockIdx.y;
  const unsigned row0 = blockIdx.x * SR;
  const uint32_t p = tabs[limb].p, p2 = tabs[limb].p2, pinv = tabs[limb].pinv, r2 = tabs[limb].r2;
  const int t = threadIdx.x;
  const int jq = t & 15, b = (t >> 4) % NB, rg = (t >> 4) / NB;
  const MT* Mbase = M + (size_t)limb * m_ps + (size_t)mat * rows * cols * D;

  // producer role for the matrix: thread t fetches quads q = t + i*ST_THREADS of the tile; quad q -> row
  // q / (ST_COLS*16), (poly, slot quad) = q % (ST_COLS*16): consecutive threads read consecutive bytes of a row
  const mraw* src[QPT];
  int src_poly[QPT];
#pragma unroll
  for (int i = 0; i < QPT; i++) {
    const int qd = t + i * ST_THREADS;
    const unsigned row = min(row0 + qd / (ST_COLS * 16), rows - 1);
    src[i] = reinterpret_cast<const mraw*>(Mbase + (size_t)row * cols * D) + qd % (ST_COLS * 16);
    src_poly[i] = (qd % (ST_COLS * 16)) / 16;
  }
  // producer role for the vectors: thread t < VQ fetches quad (set t / (ST_COLS*16), poly, slot quad)
  const bool vprod = t < VQ;
  const int v_poly = (t % (ST_COLS * 16)) / 16;
  const uint4* vsrc = reinterpret_cast<const uint4*>(V + (size_t)limb * v_ps + (size_t)(vprod ? t / (ST_COLS * 16) : 0) * v_bs +
                                                     (size_t)mat * cols * D) + t % (ST_COLS * 16);
  uint64_t acc[RG][4];
#pragma unroll
  for (int r = 0; r < RG; r++)
#pragma unroll
    for (int k = 0; k < 4; k++) acc[r][k] = 0;

  const unsigned ntiles = (cols + ST_COLS - 1) / ST_COLS;
  // two tiles are kept in flight in registers (sets A and B) on top of the one being consumed from LDS; the tile
  // loop is unrolled by two so both register sets are statically indexed
  mraw mA[QPT], mB[QPT];
  uint4 vA = make_uint4(0, 0, 0, 0), vB = make_uint4(0, 0, 0, 0);
  auto fetch = [&](unsigned tl, mraw (&mreg)[QPT], uint4& vq) {
    const unsigned c0 = tl * ST_COLS;
#pragma unroll
    for (int i = 0; i < QPT; i++) {
      mraw z = {};
      mreg[i] = (c0 + src_poly[i] < cols) ? Quad<MT>::load_raw(src[i] + (size_t)c0 * 16) : z;
    }
    vq = (vprod && c0 + v_poly < cols) ? vsrc[(size_t)c0 * 16] : make_uint4(0, 0, 0, 0);
  };
  auto park = [&](int buf, const mraw (&mreg)[QPT], const uint4& vq) {
#pragma unroll
    for (int i = 0; i < QPT; i++) tile_m[buf][t + i * ST_THREADS] = mreg[i];
    if (vprod) tile_v[buf][t] = vq;
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int cc = 0; cc < ST_COLS; cc++) {
      const uint4 v = tile_v[buf][(b * ST_COLS + cc) * 16 + jq];
#pragma unroll
      for (int r = 0; r < RG; r++) {
        const uint4 m = Quad<MT>::unpack(tile_m[buf][((rg * RG + r) * ST_COLS + cc) * 16 + jq]);
        acc[r][0] += (uint64_t)m.x * v.x;
        acc[r][1] += (uint64_t)m.y * v.y;
        acc[r][2] += (uint64_t)m.z * v.z;
        acc[r][3] += (uint64_t)m.w * v.w;
      }
    }
  };
  auto fold = [&]() {
#pragma unroll
    for (int r = 0; r < RG; r++)
#pragma unroll
      for (int k = 0; k < 4; k++) acc[r][k] = acc_fold(acc[r][k], p, p2);
  };
  // prologue: tile 0 -> LDS[0], tile 1 -> set A, tile 2 -> set B
  fetch(0, mA, vA);
  park(0, mA, vA);
  if (ntiles > 1) fetch(1, mA, vA);
  if (ntiles > 2) fetch(2, mB, vB);
  __syncthreads();
  for (unsigned tl = 0; tl < ntiles; tl += 2) {
    // even step: consume tile tl from LDS[0]; tile tl+1 (set A) is parked into LDS[1]; set A is refilled with tl+3
    compute(0);
    if (tl + 1 < ntiles) {
      park(1, mA, vA);
      if (tl + 3 < ntiles) fetch(tl + 3, mA, vA);
    }
    __syncthreads();
    if (tl + 1 >= ntiles) { fold(); break; }
    // odd step: consume tile tl+1 from LDS[1]; tile tl+2 (set B) is parked into LDS[0]; set B is refilled with tl+4
    compute(1);
    fold();  // 2 tiles x ST_COLS = 8 terms since the last fold
    if (tl + 2 < ntiles) {
      park(0, mB, vB);
      if (tl + 4 < ntiles) fetch(tl + 4, mB, vB);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < RG; r++) {
    const unsigned row = row0 + rg * RG + r;
    if (row < rows) {
      uint4 o;
      o.x = mont_mul(redc64(acc[r][0], p, pinv), r2, p, pinv);
      o.y = mont_mul(redc64(acc[r][1], p, pinv), r2, p, pinv);
      o.z = mont_mul(redc64(acc[r][2], p, pinv), r2, p, pinv);
      o.w = mont_mul(redc64(acc[r][3], p, pinv), r2, p, pinv);
      *reinterpret_cast<uint4*>(partial + (size_t)limb * partial_ps + (size_t)b * partial_bs +
                                ((size_t)mat * rows + row) * D + jq * 4) = o;
    }
  }
}

// ---- batched mat-vec with LANE = ROW ("bcast" kernel)
// The sets kernel above reads one LDS quad per 4..8 multiply-adds and is bound by the LDS, not by HBM, from 4 sets
// on.  Here the roles are turned round: a 1024-thread block owns BC_ROWS = 64 matrix rows, LANE = ROW and
// WAVE = SLOT QUAD.  The vector operand v[set][col][4 slots of this wave] is then wave-uniform: one broadcast LDS
// read serves 64 rows, and one matrix quad read serves all NB sets.  The matrix tile (64 rows x BC_COLS
// polynomials) is fetched with coalesced non-temporal loads (a row's 512 B contiguous), parked in LDS row-major with
// a pitch of BC_COLS*16+1 quads, and read back transposed (lane = row: the 16 lanes of an LDS lane group hit 16
// different 16-byte slots).  Per column and wave: 1 + NB LDS reads feed 4*NB multiply-adds on 64 lanes.
// (Scalar loads of v straight from global memory were tried first: no LDS at all, but every tile then waits a
// full HBM latency on the scalar path, 3x slower.)
// Accumulators: NB x 4 u64 per lane.
constexpr int BC_ROWS = 64;
constexpr int BC_COLS = 2;
constexpr int BC_THREADS = 1024;

template <typename MT, int NB>
__global__ void __launch_bounds__(BC_THREADS)
k_matvec_bcast(uint32_t* __restrict__ partial, size_t partial_ps, size_t partial_bs, const MT* __restrict__ M, size_t m_ps,
               const uint32_t* __restrict__ V, size_t v_ps, size_t v_bs, unsigned rows, unsigned cols,
               const LimbTable* __restrict__ tabs) {
  typedef typename Quad<MT>::raw mraw;
  constexpr int PITCH = BC_COLS * 16 + 1;
  constexpr int QPT = BC_ROWS * BC_COLS * 16 / BC_THREADS;    // matrix quads per thread per tile (2)
  constexpr int VQ = NB * BC_COLS * 16;                       // vector quads per tile (<= 256)
  __shared__ mraw tile[2][BC_ROWS * PITCH];
  __shared__ uint4 tile_v[2][VQ];
  const int limb = blockIdx.z;
  const unsigned mat = blockIdx.y;
  const unsigned row0 = blockIdx.x * BC_ROWS;
  const uint32_t p = tabs[limb].p, p2 = tabs[limb].p2, pinv = tabs[limb].pinv, r2 = tabs[limb].r2;
  const int t = threadIdx.x, lane = t & 63;
  const int jq = __builtin_amdgcn_readfirstlane(t >> 6);
  const MT* Mbase = M + (size_t)limb * m_ps + (size_t)mat * rows * cols * D;

  const mraw* src[QPT];
  int dst[QPT], src_poly[QPT];
#pragma unroll
  for (int i = 0; i < QPT; i++) {
    const int qd = t + i * BC_THREADS;
    const int rr = qd / (BC_COLS * 16), within = qd % (BC_COLS * 16);
    const unsigned row = min(row0 + rr, rows - 1);
    src[i] = reinterpret_cast<const mraw*>(Mbase + (size_t)row * cols * D) + within;
    src_poly[i] = within / 16;
    dst[i] = rr * PITCH + within;
  }
  // producer role for the vectors: thread t < VQ fetches quad (set t / (BC_COLS*16), poly, slot quad)
  const bool vprod = t < VQ;
  const int v_poly = (t % (BC_COLS * 16)) / 16;
  const uint4* vsrc = reinterpret_cast<const uint4*>(V + (size_t)limb * v_ps + (size_t)(vprod ? t / (BC_COLS * 16) : 0) * v_bs +
                                                     (size_t)mat * cols * D) + t % (BC_COLS * 16);

  uint64_t acc[NB][4];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int k = 0; k < 4; k++) acc[b][k] = 0;

  const unsigned ntiles = (cols + BC_COLS - 1) / BC_COLS;
  mraw mA[QPT], mB[QPT];
  uint4 vA = make_uint4(0, 0, 0, 0), vB = make_uint4(0, 0, 0, 0);
  auto fetch = [&](unsigned tl, mraw (&mreg)[QPT], uint4& vq) {
    const unsigned c0 = tl * BC_COLS;
#pragma unroll
    for (int i = 0; i < QPT; i++) {
      mraw z = {};
      mreg[i] = (c0 + src_poly[i] < cols) ? Quad<MT>::load_raw(src[i] + (size_t)c0 * 16) : z;
    }
    vq = (vprod && c0 + v_poly < cols) ? vsrc[(size_t)c0 * 16] : make_uint4(0, 0, 0, 0);
  };
  auto park = [&](int buf, const mraw (&mreg)[QPT], const uint4& vq) {
#pragma unroll
    for (int i = 0; i < QPT; i++) tile[buf][dst[i]] = mreg[i];
    if (vprod) tile_v[buf][t] = vq;
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int cc = 0; cc < BC_COLS; cc++) {
      const uint4 m = Quad<MT>::unpack(tile[buf][lane * PITCH + cc * 16 + jq]);
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const uint4 v = tile_v[buf][(b * BC_COLS + cc) * 16 + jq];   // wave-uniform address: broadcast
        acc[b][0] += (uint64_t)m.x * v.x;
        acc[b][1] += (uint64_t)m.y * v.y;
        acc[b][2] += (uint64_t)m.z * v.z;
        acc[b][3] += (uint64_t)m.w * v.w;
        if (NB > 4 && (b & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keep at most 4 vector quads live
      }
    }
  };
  auto fold = [&]() {
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int k = 0; k < 4; k++) acc[b][k] = acc_fold_lazy(acc[b][k], p2);
  };
  fetch(0, mA, vA);
  park(0, mA, vA);
  if (ntiles > 1) fetch(1, mA, vA);
  if (ntiles > 2) fetch(2, mB, vB);
  __syncthreads();
  for (unsigned tl = 0; tl < ntiles; tl += 2) {
    compute(0);
    if (tl + 1 < ntiles) {
      park(1, mA, vA);
      if (tl + 3 < ntiles) fetch(tl + 3, mA, vA);
    }
    __syncthreads();
    if (tl + 1 >= ntiles) break;
    compute(1);
    if (tl & 2) fold();  // 4 tiles x BC_COLS = 8 terms since the last fold
    if (tl + 2 < ntiles) {
      park(0, mB, vB);
      if (tl + 4 < ntiles) fetch(tl + 4, mB, vB);
    }
    __syncthreads();
  }
  const unsigned row = row0 + lane;
  if (row < rows) {
#pragma unroll
    for (int b = 0; b < NB; b++) {
      uint4 o;
      o.x = mont_mul(redc64(acc_fold(acc[b][0], p, p2), p, pinv), r2, p, pinv);
      o.y = mont_mul(redc64(acc_fold(acc[b][1], p, p2), p, pinv), r2, p, pinv);
      o.z = mont_mul(redc64(acc_fold(acc[b][2], p, p2), p, pinv), r2, p, pinv);
      o.w = mont_mul(redc64(acc_fold(acc[b][3], p, p2), p, pinv), r2, p, pinv);
      *reinterpret_cast<uint4*>(partial + (size_t)limb * partial_ps + (size_t)b * partial_bs +
                                ((size_t)mat * rows + row) * D + jq * 4) = o;
    }
  }
}

// ---- the bcast kernel with the tiles moved by LDS-DMA (global_load_lds_dwordx4) into a 3-deep LDS ring
// Same lane = row / wave = slot-quad arithmetic as k_matvec_bcast, but no staging registers: two tiles stay in
// flight behind the one being consumed, one raw s_barrier per tile, counted vmcnt waits.  This is what lets 8 sets
// (64 accumulator VGPRs) fit the 128-VGPR budget of a 1024-thread block.  One DMA instruction writes 1 KiB of LDS
// contiguously (wave-uniform base + lane*16), i.e. 2 matrix rows of a u32 tile (4 of a u16 tile); pieces are spaced
// 1040 B apart so that the transposed reads (lane = row) spread over the banks (2-way / 4-way instead of 32-way).
// Requires an even column count (a DMA cannot zero-fill); other shapes take the register-staged kernel.
// What bounds it (measured, tools/ubench_mad.hip and timing-only builds): with 8 sets the kernel is VALU-bound, not
// LDS-bound — dropping 3/4 of the vector LDS reads changes nothing, dropping 3/4 of the multiply-adds does.  A
// v_mad_u64_u32 whose two multiplicands are distinct VGPRs issues in 9.3 cycles per wave64 (6.8 with an SGPR
// multiplier, 4.9 when it re-reads its own accumulator): 1.65e13 multiply-adds/s chip-wide, of which the 8-set
// launches reach 85-90 %.  Step 3 at 8 sets needs 1.0e11 of them per u32 limb pair: that, no longer HBM, is the
// floor of this formulation.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS reads of the DMA ring are written as inline asm: the compiler's wait-count pass treats an LDS read it can
// see as possibly aliasing every LDS-DMA in flight and drains the DMA queue (vmcnt(0)) in front of it, which
// would serialise the ring.  The s_waitcnt sits inside the same statement, so the results are defined when the
// statement ends and nothing can be scheduled between the reads and the wait.
// GS vector quads (sets b0 .. b0+GS-1 of column cc) from vaddr + V0 + i*VS, optionally preceded by the matrix quad.
__device__ __forceinline__ uint4 unpack_quad(const u32x4_t& r) { return make_uint4(r.x, r.y, r.z, r.w); }
__device__ __forceinline__ uint4 unpack_quad(const u32x2_t& r) {
  return make_uint4(r.x & 0xFFFFu, r.x >> 16, r.y & 0xFFFFu, r.y >> 16);
}
template <int MOFF>
__device__ __forceinline__ void lds_issue_m(u32x4_t& m, uint32_t maddr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(m) : "v"(maddr), "n"(MOFF));
}
template <int MOFF>
__device__ __forceinline__ void lds_issue_m(u32x2_t& m, uint32_t maddr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(m) : "v"(maddr), "n"(MOFF));
}
template <int GS, int V0, int VS, typename MR>
__device__ __forceinline__ void lds_read_v(MR& m, u32x4_t (&v)[4], uint32_t vaddr) {
  // "+v"(m): the matrix quad issued just before becomes valid at this wait too
  if constexpr (GS == 4) {
    asm volatile(
        "ds_read_b128 %1, %5 offset:%6\n\t"
        "ds_read_b128 %2, %5 offset:%7\n\t"
        "ds_read_b128 %3, %5 offset:%8\n\t"
        "ds_read_b128 %4, %5 offset:%9\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "+v"(m), "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
        : "v"(vaddr), "n"(V0), "n"(V0 + VS), "n"(V0 + 2 * VS), "n"(V0 + 3 * VS));
  } else {
    asm volatile(
        "ds_read_b128 %1, %3 offset:%4\n\t"
        "ds_read_b128 %2, %3 offset:%5\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "+v"(m), "=&v"(v[0]), "=&v"(v[1])
        : "v"(vaddr), "n"(V0), "n"(V0 + VS));
  }
}

// Shapes served (BcastArgs): the batched mat-vec (ngroups = 1: NB vector sets, grid = row blocks x matrices x limbs)
// and the matrix-matrix product out[set][row] = sum_c M[row][c] * V[c][set] with many sets (the Ajtai commitment
// t_i = A s_i of every witness column of every proof in the batch): the sets are cut into `ngroups` groups of NB,
// and blockIdx.x is decoded so that the groups of one row block run on ONE XCD at the same time — the first of them
// pulls the matrix tile from HBM, the others hit that XCD's L2.
struct BcastArgs {
  uint32_t* partial;        // out + limb*partial_ps + job*partial_js + group*partial_gs + b*partial_bs + (mat*rows+row)*64
  size_t partial_ps, partial_bs, partial_gs, partial_js;
  const void* M;            // element (mat, row, c) at M + limb*m_ps + mat*m_ms + row*m_rs + c*m_cs (strides in residues)
  size_t m_ps, m_ms, m_rs, m_cs;
  const uint32_t* V;        // V + limb*v_ps + job*v_js + group*v_gs + b*v_bs + (mat*cols + c)*v_cs
  size_t v_ps, v_bs, v_cs, v_gs, v_js;
  unsigned rows, cols, rowblocks, ngroups, groups_per_job;
  const LimbTable* tabs;
};

// TC = ring-slot width in columns (2 or 4), DEPTH = ring slots (3: two tiles in flight behind the one being
// consumed; 2: one, with twice the work between barriers when TC = 4).
template <typename MT, int NB, int TC, int DEPTH>
__global__ void __launch_bounds__(BC_THREADS)
k_matvec_bcast_dma(const BcastArgs a) {
  typedef typename Quad<MT>::raw mraw;
  constexpr int RAWB = sizeof(mraw);                 // bytes per slot quad in the matrix: 16 (u32) / 8 (u16)
  constexpr int ROWB = TC * 16 * RAWB;               // tile bytes per matrix row
  constexpr int RPI = 1024 / ROWB;                   // rows per DMA instruction (1 KiB pieces)
  constexpr int LPR = 64 / RPI;                      // lanes per row
  constexpr int LPE = LPR / TC;                      // lanes per ring element: 16 (u32) / 8 (u16)
  constexpr int NI = BC_ROWS / RPI;                  // matrix DMA instructions per tile
  constexpr int MI = NI / 16;                        // ... per wave
  constexpr int PIECE = 1024 + 16;
  constexpr int SPP = 4 / TC;                        // vector sets per DMA instruction (a set's TC columns are TC*256 B)
  constexpr int NV = NB / SPP;                       // vector DMA instructions per tile
  constexpr int G = MI + 1;                          // DMA instructions per wave per tile
  constexpr int FOLD_TILES = 8 / TC;                 // tiles between accumulator folds (8 terms)
  static_assert(RPI >= 1 && MI >= 1 && NV >= 1 && NV <= 16 && (DEPTH == 2 || DEPTH == 3) && (TC == 2 || TC == 4), "tile shape");
  // ring slots as separate LDS objects: the compiler then knows that accesses to one slot do not alias the DMA
  // writes into another
  __shared__ __attribute__((aligned(16))) unsigned char tm0[NI * PIECE], tm1[NI * PIECE], tm2[DEPTH == 3 ? NI * PIECE : 16];
  __shared__ uint4 tv0[NB * TC * 16], tv1[NB * TC * 16], tv2[DEPTH == 3 ? NB * TC * 16 : 1];
  const int limb = blockIdx.z;
  const unsigned mat = blockIdx.y;
  unsigned rb = blockIdx.x, grp = 0;
  if (a.ngroups > 1) {      // XCD x = blockIdx.x % 8 walks (row block x + 8k, group 0 .. ngroups-1), k = 0, 1, ...
    const unsigned x = blockIdx.x & 7, seq = blockIdx.x >> 3;
    rb = x + 8 * (seq / a.ngroups);
    grp = seq % a.ngroups;
    if (rb >= a.rowblocks) return;   // whole block, before any barrier
  }
  const unsigned job = grp / a.groups_per_job, gin = grp % a.groups_per_job;
  const unsigned rows = a.rows, cols = a.cols;
  const size_t v_bs = a.v_bs, v_cs = a.v_cs;
  const unsigned row0 = rb * BC_ROWS;
  const LimbTable* tabs = a.tabs;
  const uint32_t p = tabs[limb].p, p2 = tabs[limb].p2, pinv = tabs[limb].pinv, r2 = tabs[limb].r2;
  const int t = threadIdx.x, lane = t & 63;
  const int jq = __builtin_amdgcn_readfirstlane(t >> 6);   // wave index = slot quad
  const unsigned char* Mbase = reinterpret_cast<const unsigned char*>(static_cast<const MT*>(a.M) + (size_t)limb * a.m_ps +
                                                                      (size_t)mat * a.m_ms);
  const uint32_t m_rsb = (uint32_t)(a.m_rs * sizeof(MT)), m_csb = (uint32_t)(a.m_cs * sizeof(MT));   // strides in bytes
  uint32_t* partial = a.partial + (size_t)limb * a.partial_ps + (size_t)job * a.partial_js + (size_t)gin * a.partial_gs;
  const size_t partial_bs = a.partial_bs;

  // sources as wave-uniform base + 32-bit lane offset (the launcher checks that 64 rows + TC columns span < 2^32 bytes)
  // lane -> (row k*RPI + lane/LPR of the block, column (lane%LPR)/LPE of the tile, 16-byte piece of the element)
  uint32_t moffs[MI];
#pragma unroll
  for (int i = 0; i < MI; i++) {
    const unsigned row = min(row0 + (jq * MI + i) * RPI + lane / LPR, rows - 1) - row0;
    moffs[i] = row * m_rsb + ((lane % LPR) / LPE) * m_csb + (lane % LPE) * 16;
  }
  const unsigned char* mrow0 = Mbase + (size_t)row0 * m_rsb;
  const int kv = jq % NV;                              // waves beyond NV repeat a vector piece (same bytes, same place)
  const uint32_t* vbase = a.V + (size_t)limb * a.v_ps + (size_t)job * a.v_js + (size_t)gin * a.v_gs + (size_t)(kv * SPP) * v_bs +
                          (size_t)mat * cols * v_cs;
  // lane -> (set kv*SPP + lane/(16*TC), column (lane/16)%TC, slot quad lane%16); bytes
  const uint32_t voff = (uint32_t)((lane / (16 * TC)) * v_bs + ((lane / 16) % TC) * v_cs + (lane % 16) * 4) * 4;

  uint64_t acc[NB][4];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int k = 0; k < 4; k++) acc[b][k] = 0;

  const unsigned ntiles = cols / TC;
  // tile index clamped: past the end the last tile is fetched again into a slot nobody reads, so every step issues
  // and waits for the same number of DMA instructions (straight-line counts the compiler can follow)
  auto issue = [&](unsigned tl, unsigned char* tm, uint4* tv) {
    tl = min(tl, ntiles - 1);
    const unsigned char* mb = mrow0 + (size_t)tl * TC * m_csb;
    const unsigned char* vb = reinterpret_cast<const unsigned char*>(vbase + (size_t)tl * TC * v_cs);
#pragma unroll
    for (int i = 0; i < MI; i++)
      __builtin_amdgcn_global_load_lds((gptr_t)(mb + moffs[i]), (lptr_t)(tm + (jq * MI + i) * PIECE), 16, 0, 2);
    __builtin_amdgcn_global_load_lds((gptr_t)(vb + voff), (lptr_t)(tv + kv * 64), 16, 0, 0);
  };
  const uint32_t moff = (lane / RPI) * PIECE + (lane % RPI) * ROWB + jq * RAWB;
  const uint32_t voffl = jq * 16;
  constexpr int GS = NB < 4 ? NB : 4;
  // one tile: consume ring slot (tm, tv) while tile tl+DEPTH-1 is sent to slot (nm, nv), the one tile tl-1 was read from
  auto step = [&](unsigned tl, unsigned char* tm, uint4* tv, unsigned char* nm, uint4* nv) {
    // this wave's pieces of tile tl have landed (with 3 slots tile tl+1's G instructions stay in flight) ...
    // (the builtin, not inline asm: the compiler's own wait-count pass has to see it)
    __builtin_amdgcn_s_waitcnt(0x0F70 | (DEPTH == 3 ? G : 0));   // vmcnt(G) / vmcnt(0), expcnt/lgkmcnt untouched
    // ... and after the barrier so have everybody's; everybody has also finished reading tile tl-1
    __builtin_amdgcn_s_barrier();
    issue(tl + DEPTH - 1, nm, nv);
    const uint32_t maddr = (uint32_t)(uintptr_t)(lptr_t)tm + moff;
    const uint32_t vaddr = (uint32_t)(uintptr_t)(lptr_t)tv + voffl;      // wave-uniform: broadcast reads
    auto column = [&](auto cc_tag) {
      constexpr int cc = decltype(cc_tag)::value;
      typename std::conditional<RAWB == 16, u32x4_t, u32x2_t>::type mr;
      lds_issue_m<cc * 16 * RAWB>(mr, maddr);
      uint4 m;
#pragma unroll
      for (int g = 0; g < NB / GS; g++) {
        u32x4_t v[4];
        if (g == 0) lds_read_v<GS, cc * 256, TC * 256>(mr, v, vaddr);
        else lds_read_v<GS, (GS * TC + cc) * 256, TC * 256>(mr, v, vaddr);
        if (g == 0)
          m = unpack_quad(mr);
#pragma unroll
        for (int i = 0; i < GS; i++) {
          const int b = g * GS + i;
          acc[b][0] += (uint64_t)m.x * v[i].x;
          acc[b][1] += (uint64_t)m.y * v[i].y;
          acc[b][2] += (uint64_t)m.z * v[i].z;
          acc[b][3] += (uint64_t)m.w * v[i].w;
        }
      }
    };
    static_assert(NB / GS <= 2, "column() handles at most two groups of vector sets");
    column(std::integral_constant<int, 0>{});
    column(std::integral_constant<int, 1>{});
    if constexpr (TC == 4) {
      column(std::integral_constant<int, 2>{});
      column(std::integral_constant<int, 3>{});
    }
    if ((tl & (FOLD_TILES - 1)) == FOLD_TILES - 1) {      // 8 terms since the last fold
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[b][k] = acc_fold_lazy(acc[b][k], p2);
    }
  };
  if constexpr (DEPTH == 3) {
    issue(0, tm0, tv0);
    issue(1, tm1, tv1);
    for (unsigned tl = 0; tl < ntiles; tl += 3) {
      step(tl, tm0, tv0, tm2, tv2);
      if (tl + 1 >= ntiles) break;
      step(tl + 1, tm1, tv1, tm0, tv0);
      if (tl + 2 >= ntiles) break;
      step(tl + 2, tm2, tv2, tm1, tv1);
    }
  } else {
    issue(0, tm0, tv0);
    for (unsigned tl = 0; tl < ntiles; tl += 2) {
      step(tl, tm0, tv0, tm1, tv1);
      if (tl + 1 >= ntiles) break;
      step(tl + 1, tm1, tv1, tm0, tv0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // no DMA may still be writing this block's LDS when it ends
  const unsigned row = row0 + lane;
  if (row < rows) {
#pragma unroll
    for (int b = 0; b < NB; b++) {
      uint4 o;
      o.x = mont_mul(redc64(acc_fold(acc[b][0], p, p2), p, pinv), r2, p, pinv);
      o.y = mont_mul(redc64(acc_fold(acc[b][1], p, p2), p, pinv), r2, p, pinv);
      o.z = mont_mul(redc64(acc_fold(acc[b][2], p, p2), p, pinv), r2, p, pinv);
      o.w = mont_mul(redc64(acc_fold(acc[b][3], p, p2), p, pinv), r2, p, pinv);
      *reinterpret_cast<uint4*>(partial + (size_t)b * partial_bs + ((size_t)mat * rows + row) * D + jq * 4) = o;
    }
  }
}

static void set_last_matvec_kernel(const char* n);
// tile shape per matrix type and column count: 4 columns x 2 ring slots when the column count allows it, else 2 x 3
// (measured round 1: the other points of {2, 4} x {2, 3} are slower or do not fit the LDS)
template <typename MT, int NB>
static hipError_t launch_bcast_dma(const BcastArgs& a, dim3 g, hipStream_t s) {
  const int tc = a.cols % 4 == 0 ? 4 : 2, depth = tc == 4 ? 2 : 3;
  if (sizeof(MT) == 4 && a.ngroups == 1 && a.m_cs == (size_t)D) {   // a mat-vec launch (not the Ajtai / C, D shapes)
    char name[64];
    snprintf(name, sizeof name, "k_matvec_bcast_dma<unsigned int, %d, %d, %d>", NB, tc, depth);
    set_last_matvec_kernel(name);
  }
  if (tc == 2 && depth == 3) hipLaunchKernelGGL((k_matvec_bcast_dma<MT, NB, 2, 3>), g, dim3(BC_THREADS), 0, s, a);
  else if (tc == 2) hipLaunchKernelGGL((k_matvec_bcast_dma<MT, NB, 2, 2>), g, dim3(BC_THREADS), 0, s, a);
  else if (depth == 2) hipLaunchKernelGGL((k_matvec_bcast_dma<MT, NB, 4, 2>), g, dim3(BC_THREADS), 0, s, a);
  else {
    if constexpr (sizeof(MT) == 2) hipLaunchKernelGGL((k_matvec_bcast_dma<MT, NB, 4, 3>), g, dim3(BC_THREADS), 0, s, a);
    else return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// out[job][set][row] = sum_c A[row][c] * S[job][c][set]: every job's sets share A (sets % 8 == 0, inner % BC_COLS == 0,
// rows >= 32; strides in words inside a limb plane).  Returns hipErrorNotSupported for shapes this kernel does not take.
hipError_t launch_matmul_bcast(HatView out, size_t out_set_stride, size_t out_job_stride, ConstHatView A, ConstHatView S,
                               size_t s_set_stride, size_t s_col_stride, size_t s_job_stride, size_t njobs, size_t rows,
                               size_t inner, size_t nsets, int nlimbs, const LimbTable* tabs, hipStream_t s) {
  if (njobs == 0 || rows == 0 || nsets == 0) return hipSuccess;
  if (limb_tables_wide(tabs)) {   // single-limb strict-32 path: (job, set, row) outputs of an `inner`-term sum
    if (nlimbs != 1) return hipErrorInvalidValue;
    WideDot d{};
    d.out = out.base; d.A = A.base; d.B = S.base; d.tab = tabs;
    d.n[0] = d.n[1] = 1; d.n[2] = (unsigned)njobs; d.n[3] = (unsigned)nsets; d.n[4] = (unsigned)rows;
    d.os[2] = out_job_stride; d.os[3] = out_set_stride; d.os[4] = D;
    d.as[4] = inner * D; d.ak = D;
    d.bs[2] = s_job_stride; d.bs[3] = s_set_stride; d.bk = s_col_stride;
    d.K = d.Ktotal = (unsigned)inner; d.chunk_level = -1;
    return launch_wide_dot(d, s);
  }
  if (nsets % 8 || inner % BC_COLS || inner == 0 || rows < (size_t)BC_ROWS / 2 || (uint64_t)(64 * inner + 2) * 256 >= (1ull << 32) ||
      (s_set_stride * 2 + s_col_stride) * 4 >= (1ull << 32) || njobs * (nsets / 8) > 65535)
    return hipErrorNotSupported;
  BcastArgs a;
  a.partial = out.base; a.partial_ps = out.plane_stride; a.partial_bs = out_set_stride; a.partial_gs = 8 * out_set_stride;
  a.partial_js = out_job_stride;
  a.M = A.base; a.m_ps = A.plane_stride; a.m_ms = 0; a.m_rs = inner * D; a.m_cs = D;
  a.V = S.base; a.v_ps = S.plane_stride; a.v_bs = s_set_stride; a.v_cs = s_col_stride; a.v_gs = 8 * s_set_stride; a.v_js = s_job_stride;
  a.rows = (unsigned)rows; a.cols = (unsigned)inner; a.rowblocks = (unsigned)((rows + BC_ROWS - 1) / BC_ROWS);
  a.groups_per_job = (unsigned)(nsets / 8); a.ngroups = (unsigned)(njobs * (nsets / 8));
  a.tabs = tabs;
  // ngroups == 1 would be read as the plain row-block grid by the kernel: it is one, with 8 * ceil(rb/8) blocks
  dim3 g(a.ngroups > 1 ? 8 * a.ngroups * ((a.rowblocks + 7) / 8) : a.rowblocks, 1, (unsigned)nlimbs);
  return launch_bcast_dma<uint32_t, 8>(a, g, s);
}

// out_b[rho] = sum_m G_b[m] * C[m][rho] for nb digit vectors sharing C (nb in {2,4,8}), as a lane = rho product on the
// LDS-DMA kernel: "rows" are the rho, "columns" the terms, split into chunks of `chunk` terms that become the
// kernel's matrices (partial sums per chunk in `partial`, to be added up by launch_sum_terms).
// Returns hipErrorNotSupported for shapes this path does not take.
template <typename MT>
static hipError_t launch_axpy_bcast(int nb, uint32_t* partial, size_t partial_ps, size_t partial_bs, const uint32_t* G, size_t g_ps,
                                    size_t g_bs, const MT* C, size_t c_ps, size_t nterms, size_t chunk, size_t len, int nlimbs,
                                    const LimbTable* tabs, hipStream_t s) {
  if (limb_tables_wide(tabs)) {   // single-limb strict-32 path: (set, chunk, rho) outputs, chunks of `chunk` terms
    if (sizeof(MT) != 4 || nlimbs != 1 || chunk == 0) return hipErrorInvalidValue;
    WideDot d{};
    d.out = partial; d.A = G; d.B = reinterpret_cast<const uint32_t*>(C); d.tab = tabs;
    d.n[0] = d.n[1] = 1; d.n[2] = (unsigned)nb; d.n[3] = (unsigned)((nterms + chunk - 1) / chunk); d.n[4] = (unsigned)len;
    d.os[2] = partial_bs; d.os[3] = len * D; d.os[4] = D;
    d.as[2] = g_bs; d.as[3] = chunk * D; d.ak = D;
    d.bs[3] = chunk * len * D; d.bs[4] = D; d.bk = len * D;
    d.K = (unsigned)chunk; d.Ktotal = (unsigned)nterms; d.chunk_level = 3;
    return launch_wide_dot(d, s);
  }
  if (nb < 2 || len < (size_t)BC_ROWS / 2 || chunk % BC_COLS || nterms % chunk || (63 * D + len * D) * sizeof(MT) >= (1ull << 32) ||
      g_bs * 8 >= (1ull << 32))
    return hipErrorNotSupported;
  BcastArgs a;
  a.partial = partial; a.partial_ps = partial_ps; a.partial_bs = partial_bs; a.partial_gs = 0; a.partial_js = 0;
  a.M = C; a.m_ps = c_ps; a.m_ms = chunk * len * D; a.m_rs = D; a.m_cs = len * D;
  a.V = G; a.v_ps = g_ps; a.v_bs = g_bs; a.v_cs = D; a.v_gs = 0; a.v_js = 0;
  a.rows = (unsigned)len; a.cols = (unsigned)chunk; a.rowblocks = (unsigned)((len + BC_ROWS - 1) / BC_ROWS);
  a.ngroups = 1; a.groups_per_job = 1; a.tabs = tabs;
  dim3 g(a.rowblocks, (unsigned)(nterms / chunk), (unsigned)nlimbs);
  switch (nb) {
    case 2: return launch_bcast_dma<MT, 2>(a, g, s);
    case 4: return launch_bcast_dma<MT, 4>(a, g, s);
    case 8: return launch_bcast_dma<MT, 8>(a, g, s);
    default: return hipErrorNotSupported;
  }
}
// chunk length for the path above: nterms / (the smallest chunk count in 8..64 that leaves an even chunk length, else
// the largest below 8); 0 = none
size_t axpy_bcast_chunk(size_t nterms, size_t len) {
  if (len < (size_t)BC_ROWS / 2) return 0;
  for (size_t nch = 8; nch <= 64; nch++)
    if (nterms % nch == 0 && (nterms / nch) % BC_COLS == 0) return nterms / nch;
  for (size_t nch = 7; nch >= 1; nch--)
    if (nterms % nch == 0 && (nterms / nch) % BC_COLS == 0) return nterms / nch;
  return 0;
}
hipError_t launch_axpy_bcast_u32(int nb, HatView partial, size_t partial_bs, ConstHatView G, size_t g_bs, ConstHatView C,
                                 size_t nterms, size_t chunk, size_t len, int nlimbs, const LimbTable* tabs, hipStream_t s) {
  return launch_axpy_bcast<uint32_t>(nb, partial.base, partial.plane_stride, partial_bs, G.base, G.plane_stride, g_bs, C.base,
                                     C.plane_stride, nterms, chunk, len, nlimbs, tabs, s);
}
hipError_t launch_axpy_bcast_u16(int nb, uint32_t* partial_plane, size_t partial_bs, const uint32_t* G_plane, size_t g_bs,
                                 const uint16_t* C, size_t nterms, size_t chunk, size_t len, const LimbTable* tab, hipStream_t s) {
  return launch_axpy_bcast<uint16_t>(nb, partial_plane, 0, partial_bs, G_plane, 0, g_bs, C, 0, nterms, chunk, len, 1, tab, s);
}

static void set_last_matvec_kernel(const char* n);
template <typename MT>
static hipError_t launch_matvec_nb(int nb, uint32_t* partial, size_t partial_ps, size_t partial_bs, const MT* M, size_t m_ps,
                                   const uint32_t* V, size_t v_ps, size_t v_bs, size_t nmat, size_t rows, size_t cols,
                                   int nlimbs, const LimbTable* tabs, hipStream_t s, size_t njobs = 1, size_t js = 0) {
  if (nmat == 0 || rows == 0 || nlimbs == 0 || njobs == 0) return hipSuccess;
  if (njobs > 1 && nb != 1) return hipErrorInvalidValue;   // jobs are served by the single-set kernel only
  if (limb_tables_wide(tabs)) {   // single-limb strict-32 path: (job, set, matrix, row) outputs of a `cols`-term sum
    if (sizeof(MT) != 4 || nlimbs != 1) return hipErrorInvalidValue;
    WideDot d{};
    d.out = partial; d.A = reinterpret_cast<const uint32_t*>(M); d.B = V; d.tab = tabs;
    d.n[0] = 1; d.n[1] = (unsigned)njobs; d.n[2] = (unsigned)nb; d.n[3] = (unsigned)nmat; d.n[4] = (unsigned)rows;
    d.os[1] = js; d.os[2] = partial_bs; d.os[3] = rows * D; d.os[4] = D;
    d.as[1] = js; d.as[3] = rows * cols * D; d.as[4] = cols * D; d.ak = D;
    d.bs[1] = js; d.bs[2] = v_bs; d.bs[3] = cols * D; d.bk = D;
    d.K = d.Ktotal = (unsigned)cols; d.chunk_level = -1;
    d.stream_a = nmat * rows * cols >= ((size_t)1 << 22);   // a matrix of 1 GiB or more is a stream, not a working set
    set_last_matvec_kernel("k_wide_dot");
    return launch_wide_dot(d, s);
  }
  if (nb > 1 && rows >= (size_t)BC_ROWS / 2 && (v_bs % 4) == 0 && cols % BC_COLS == 0 &&
      (uint64_t)(64 * cols + 2) * 256 < (1ull << 32) && v_bs * 8 < (1ull << 32)) {  // ... tiles by LDS-DMA
    dim3 g((unsigned)((rows + BC_ROWS - 1) / BC_ROWS), (unsigned)nmat, (unsigned)nlimbs);
    BcastArgs a;
    a.partial = partial; a.partial_ps = partial_ps; a.partial_bs = partial_bs; a.partial_gs = 0; a.partial_js = 0;
    a.M = M; a.m_ps = m_ps; a.m_ms = rows * cols * D; a.m_rs = cols * D; a.m_cs = D;
    a.V = V; a.v_ps = v_ps; a.v_bs = v_bs; a.v_cs = D; a.v_gs = 0; a.v_js = 0;
    a.rows = (unsigned)rows; a.cols = (unsigned)cols; a.rowblocks = g.x; a.ngroups = 1; a.groups_per_job = 1;
    a.tabs = tabs;
    switch (nb) {
      case 2: return launch_bcast_dma<MT, 2>(a, g, s);
      case 4: return launch_bcast_dma<MT, 4>(a, g, s);
      case 8: return launch_bcast_dma<MT, 8>(a, g, s);
      default: return hipErrorInvalidValue;
    }
  }
  if (nb > 1 && nb < 8 && rows >= (size_t)BC_ROWS / 2 && (v_bs % 4) == 0) {  // tall matrix, several vector sets: "bcast" kernel
    dim3 g((unsigned)((rows + BC_ROWS - 1) / BC_ROWS), (unsigned)nmat, (unsigned)nlimbs);
#define LAB_BC(NBV)                                                                                                  \
  hipLaunchKernelGGL((k_matvec_bcast<MT, NBV>), g, dim3(BC_THREADS), 0, s, partial, partial_ps, partial_bs, M, m_ps, V, \
                     v_ps, v_bs, (unsigned)rows, (unsigned)cols, tabs)
    switch (nb) {
      case 2: LAB_BC(2); break;
      case 4: LAB_BC(4); break;
      default: return hipErrorInvalidValue;
    }
#undef LAB_BC
    return hipGetLastError();
  }
  if (nb > 1 && rows >= (size_t)ST_ROWS) {  // fewer rows: LDS-staged "sets" kernel
    const unsigned sr = (unsigned)st_rows(nb);
    if (sizeof(MT) == 4) set_last_matvec_kernel(nb == 2 ? "k_matvec_sets<unsigned int, 2>" : nb == 4 ? "k_matvec_sets<unsigned int, 4>" : "k_matvec_sets<unsigned int, 8>");
    dim3 g((unsigned)((rows + sr - 1) / sr), (unsigned)nmat, (unsigned)nlimbs);
#define LAB_ST(NBV)                                                                                                 \
  hipLaunchKernelGGL((k_matvec_sets<MT, NBV>), g, dim3(ST_THREADS), 0, s, partial, partial_ps, partial_bs, M, m_ps, V, \
                     v_ps, v_bs, (unsigned)rows, (unsigned)cols, tabs)
    switch (nb) {
      case 2: LAB_ST(2); break;
      case 4: LAB_ST(4); break;
      case 8: LAB_ST(8); break;
      default: return hipErrorInvalidValue;
    }
#undef LAB_ST
    return hipGetLastError();
  }
  if (sizeof(MT) == 4) set_last_matvec_kernel(nb == 1 ? "k_matvec_partial<unsigned int, 1>" : nb == 2 ? "k_matvec_partial<unsigned int, 2>" : nb == 4 ? "k_matvec_partial<unsigned int, 4>" : "k_matvec_partial<unsigned int, 8>");
  const unsigned rpb = nb >= 4 ? 2 : MV_ROWS;
  dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)nmat, (unsigned)(nlimbs * njobs)), block(64 * MV_WAVES);
#define LAB_MV(NBV)                                                                                                  \
  hipLaunchKernelGGL((k_matvec_partial<MT, NBV>), grid, block, 0, s, partial, partial_ps, partial_bs, M, m_ps, V, v_ps, \
                     v_bs, (unsigned)rows, (unsigned)cols, tabs, (unsigned)nlimbs, js)
  switch (nb) {
    case 1: LAB_MV(1); break;
    case 2: LAB_MV(2); break;
    case 4: LAB_MV(4); break;
    case 8: LAB_MV(8); break;
    default: return hipErrorInvalidValue;
  }
#undef LAB_MV
  return hipGetLastError();
}

// name of the kernel the last u32 mat-vec dispatch chose (for lab_ctx_profile_kernel)
static void set_last_matvec_kernel(const char* n);
// name of the kernel the calling thread's most recent mat-vec launch went to (copied into the context by the caller)
static thread_local char g_last_matvec_kernel[64] = "k_matvec_partial<unsigned int, 1>";
const char* matvec_kernel_name() { return g_last_matvec_kernel; }
static void set_last_matvec_kernel(const char* n) { snprintf(g_last_matvec_kernel, sizeof g_last_matvec_kernel, "%s", n); }

hipError_t launch_matvec_partial(HatView partial, ConstHatView M, ConstHatView V, size_t nmat, size_t rows,
                                 size_t cols, int nlimbs, const LimbTable* tabs, hipStream_t s, size_t njobs, size_t js) {
  return launch_matvec_nb<uint32_t>(1, partial.base, partial.plane_stride, 0, M.base, M.plane_stride, V.base,
                                    V.plane_stride, 0, nmat, rows, cols, nlimbs, tabs, s, njobs, js);
}

// one limb whose matrix residues are stored as u16 (`tab` = that limb's table; partial/V already offset to its plane)
hipError_t launch_matvec_partial_u16(uint32_t* partial_plane, const uint16_t* M, const uint32_t* V_plane, size_t nmat,
                                     size_t rows, size_t cols, const LimbTable* tab, hipStream_t s) {
  return launch_matvec_nb<uint16_t>(1, partial_plane, 0, 0, M, 0, V_plane, 0, 0, nmat, rows, cols, 1, tab, s);
}

// nb right-hand-side sets at once (nb in {1,2,4,8}); *_bs = word stride between consecutive sets inside a plane
hipError_t launch_matvec_partial_batched(int nb, HatView partial, size_t partial_bs, ConstHatView M, ConstHatView V,
                                         size_t v_bs, size_t nmat, size_t rows, size_t cols, int nlimbs,
                                         const LimbTable* tabs, hipStream_t s) {
  return launch_matvec_nb<uint32_t>(nb, partial.base, partial.plane_stride, partial_bs, M.base, M.plane_stride, V.base,
                                    V.plane_stride, v_bs, nmat, rows, cols, nlimbs, tabs, s);
}
hipError_t launch_matvec_partial_batched_u16(int nb, uint32_t* partial_plane, size_t partial_bs, const uint16_t* M,
                                             const uint32_t* V_plane, size_t v_bs, size_t nmat, size_t rows, size_t cols,
                                             const LimbTable* tab, hipStream_t s) {
  return launch_matvec_nb<uint16_t>(nb, partial_plane, 0, partial_bs, M, 0, V_plane, 0, v_bs, nmat, rows, cols, 1, tab, s);
}

// out[rho][j] = sum_m partial[m][rho][j] mod p
__global__ void __launch_bounds__(256)
k_sum_terms(uint32_t* __restrict__ out, size_t out_ps, const uint32_t* __restrict__ partial, size_t partial_ps,
            unsigned nterms, size_t row_words, const LimbTable* __restrict__ tabs, unsigned nlimbs, size_t out_js,
            size_t partial_js) {
  const int limb = blockIdx.y % nlimbs;
  const size_t job = blockIdx.y / nlimbs;
  const uint32_t p = tabs[limb].p;
  const uint32_t* P = partial + (size_t)limb * partial_ps + job * partial_js;
  out += job * out_js;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < row_words; i += stride) {
    uint64_t acc = 0;
    unsigned m = 0;
    for (; m + 8 <= nterms; m += 8) {  // 8 independent loads in flight
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = P[(size_t)(m + u) * row_words + i];
#pragma unroll
      for (int u = 0; u < 8; u++) acc += v[u];
    }
    for (; m < nterms; m++) acc += P[(size_t)m * row_words + i];
    out[(size_t)limb * out_ps + i] = (uint32_t)(acc % p);
  }
}

hipError_t launch_sum_terms(HatView out, ConstHatView partial, size_t nterms, size_t rows, int nlimbs,
                            const LimbTable* tabs, hipStream_t s, size_t njobs, size_t out_js, size_t partial_js) {
  if (rows == 0 || njobs == 0) return hipSuccess;
  size_t row_words = rows * D;
  unsigned blocks = (unsigned)((row_words + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_sum_terms, dim3(blocks, (unsigned)(nlimbs * njobs)), dim3(256), 0, s, out.base, out.plane_stride,
                     partial.base, partial.plane_stride, (unsigned)nterms, row_words, tabs, (unsigned)nlimbs, out_js, partial_js);
  return hipGetLastError();
}

}  // namespace lab

namespace lab {

// ------------------------------------------------ multi right-hand-side commitment
// out[i][rho][j] = sum_x A[rho][x][j] * S[x][i][j]        (inner Ajtai commitment,
// src/proofgen.rs:41-49: t_i = A s_i for all i < r at once).
// Per NTT slot this is a (rows x inner)(inner x ncols) product; with ncols = 32
// it is VALU-bound (32 multiply-adds per 4 bytes of A), so the tile is shaped
// for register reuse: a lane owns one slot j, MM_RT rows and MM_CT columns
// (MM_RT*MM_CT 64-bit accumulators); wave w of the block owns columns
// MM_CT*w..; A values are shared by the block's waves through L1, S by all
// blocks through L2.
constexpr int MM_RT = 4;
constexpr int MM_CT = 8;

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3)))
k_matmul_t(uint32_t* __restrict__ out, size_t out_ps, const uint32_t* __restrict__ A, size_t a_ps,
           const uint32_t* __restrict__ S, size_t s_ps, unsigned rows, unsigned inner, unsigned ncols,
           const LimbTable* __restrict__ tabs) {
  const int limb = blockIdx.z;
  const uint32_t p = tabs[limb].p, p2 = tabs[limb].p2, pinv = tabs[limb].pinv, r2 = tabs[limb].r2;
  const int j = threadIdx.x;                               // slot
  const unsigned col0 = (blockIdx.y * blockDim.y + threadIdx.y) * MM_CT;
  const unsigned row0 = blockIdx.x * MM_RT;
  const uint32_t* Ap = A + (size_t)limb * a_ps;
  const uint32_t* Sp = S + (size_t)limb * s_ps;
  if (col0 >= ncols) return;

  size_t arow[MM_RT];
#pragma unroll
  for (int r = 0; r < MM_RT; r++) arow[r] = (size_t)min(row0 + r, rows - 1) * inner * D + j;
  unsigned scol[MM_CT];
#pragma unroll
  for (int c = 0; c < MM_CT; c++) scol[c] = min(col0 + c, ncols - 1) * D + j;

  uint64_t acc[MM_RT][MM_CT];
#pragma unroll
  for (int r = 0; r < MM_RT; r++)
#pragma unroll
    for (int c = 0; c < MM_CT; c++) acc[r][c] = 0;

  unsigned x = 0;
  // main loop: 8 terms per accumulator fold, in four groups of 2 whose 24 loads are all issued before their
  // multiply-adds (a plain per-x loop waits on every load and runs latency-bound)
  for (; x + 8 <= inner; x += 8) {
#pragma unroll
    for (int half = 0; half < 4; half++) {
      uint32_t a[2][MM_RT], s[2][MM_CT];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const size_t xx = x + half * 2 + u;
#pragma unroll
        for (int r = 0; r < MM_RT; r++) a[u][r] = Ap[arow[r] + xx * D];
#pragma unroll
        for (int c = 0; c < MM_CT; c++) s[u][c] = Sp[xx * ncols * D + scol[c]];
      }
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int r = 0; r < MM_RT; r++)
#pragma unroll
          for (int c = 0; c < MM_CT; c++) acc[r][c] += (uint64_t)a[u][r] * s[u][c];
    }
#pragma unroll
    for (int r = 0; r < MM_RT; r++)
#pragma unroll
      for (int c = 0; c < MM_CT; c++) acc[r][c] = acc_fold(acc[r][c], p, p2);
  }
  for (; x < inner; x++) {  // fewer than 8 terms left
    uint32_t a[MM_RT], s[MM_CT];
#pragma unroll
    for (int r = 0; r < MM_RT; r++) a[r] = Ap[arow[r] + (size_t)x * D];
#pragma unroll
    for (int c = 0; c < MM_CT; c++) s[c] = Sp[(size_t)x * ncols * D + scol[c]];
#pragma unroll
    for (int r = 0; r < MM_RT; r++)
#pragma unroll
      for (int c = 0; c < MM_CT; c++) acc[r][c] += (uint64_t)a[r] * s[c];
  }
#pragma unroll
  for (int r = 0; r < MM_RT; r++)
#pragma unroll
    for (int c = 0; c < MM_CT; c++) acc[r][c] = acc_fold(acc[r][c], p, p2);
#pragma unroll
  for (int r = 0; r < MM_RT; r++)
#pragma unroll
    for (int c = 0; c < MM_CT; c++) {
      if (row0 + r < rows && col0 + c < ncols) {
        uint32_t v = mont_mul(redc64(acc[r][c], p, pinv), r2, p, pinv);
        out[(size_t)limb * out_ps + ((size_t)(col0 + c) * rows + row0 + r) * D + j] = v;
      }
    }
}

hipError_t launch_matmul_t(HatView out, ConstHatView A, ConstHatView S, size_t rows, size_t inner, size_t ncols,
                           int nlimbs, const LimbTable* tabs, hipStream_t s) {
  if (rows == 0 || ncols == 0) return hipSuccess;
  if (limb_tables_wide(tabs)) {   // single-limb strict-32 path: out[col][row] = sum_x A[row][x] S[x][col]
    if (nlimbs != 1) return hipErrorInvalidValue;
    WideDot d{};
    d.out = out.base; d.A = A.base; d.B = S.base; d.tab = tabs;
    d.n[0] = d.n[1] = d.n[2] = 1; d.n[3] = (unsigned)ncols; d.n[4] = (unsigned)rows;
    d.os[3] = rows * D; d.os[4] = D;
    d.as[4] = inner * D; d.ak = D;
    d.bs[3] = D; d.bk = ncols * D;
    d.K = d.Ktotal = (unsigned)inner; d.chunk_level = -1;
    return launch_wide_dot(d, s);
  }
  unsigned col_groups = (unsigned)((ncols + MM_CT - 1) / MM_CT);
  unsigned wy = col_groups < 4 ? col_groups : 4;
  dim3 block(64, wy);
  dim3 grid((unsigned)((rows + MM_RT - 1) / MM_RT), (col_groups + wy - 1) / wy, (unsigned)nlimbs);
  hipLaunchKernelGGL(k_matmul_t, grid, block, 0, s, out.base, out.plane_stride, A.base, A.plane_stride, S.base,
                     S.plane_stride, (unsigned)rows, (unsigned)inner, (unsigned)ncols, tabs);
  return hipGetLastError();
}


// ------------------------------------------------ sum of polynomial-times-vector terms
// partial[chunk][rho][j] = sum_{m in chunk} G[m][j] * C[m][rho][j]
// The C and D parts of the outer commitments (src/proofgen.rs:137-148, 364-378:
// rhs += poly_by_poly_vec(g_ij^(k), C_ijk)).  C is streamed once; a wave
// instruction covers 4 consecutive rho (1 KiB contiguous).
constexpr int AX_CHUNK = 64;

// NB = number of independent digit-vector sets (proofs) sharing the streamed C: set b reads G + b*g_bs and writes
// partial + b*partial_bs (word strides inside one limb plane).
template <typename MT, int NB>
__global__ void __launch_bounds__(256)
k_axpy_partial(uint32_t* __restrict__ partial, size_t partial_ps, size_t partial_bs, const uint32_t* __restrict__ G,
               size_t g_ps, size_t g_bs, const MT* __restrict__ C, size_t c_ps, unsigned nterms, unsigned len,
               const LimbTable* __restrict__ tabs) {
  typedef typename Quad<MT>::raw mraw;
  const int limb = blockIdx.z;
  const uint32_t p = tabs[limb].p, p2 = tabs[limb].p2, pinv = tabs[limb].pinv, r2 = tabs[limb].r2;
  const unsigned rho = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int jq = threadIdx.x & 15;
  const unsigned m0 = blockIdx.y * AX_CHUNK, m1 = min(m0 + AX_CHUNK, nterms);
  if (rho >= len) return;
  const uint4* Gp = reinterpret_cast<const uint4*>(G + (size_t)limb * g_ps) + jq;
  const size_t gb4 = g_bs / 4;
  const mraw* Cp = reinterpret_cast<const mraw*>(C + (size_t)limb * c_ps) + (size_t)rho * 16 + jq;
  uint64_t acc[NB][4];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int k = 0; k < 4; k++) acc[b][k] = 0;
  unsigned m = m0;
  while (m < m1) {
    const unsigned me = min(m + 8, m1);
    for (; m < me; m++) {
      const uint4 c = Quad<MT>::load(Cp + (size_t)m * len * 16);
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const uint4 g = Gp[(size_t)m * 16 + b * gb4];
        acc[b][0] += (uint64_t)g.x * c.x; acc[b][1] += (uint64_t)g.y * c.y;
        acc[b][2] += (uint64_t)g.z * c.z; acc[b][3] += (uint64_t)g.w * c.w;
      }
    }
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int k = 0; k < 4; k++) acc[b][k] = acc_fold(acc[b][k], p, p2);
  }
#pragma unroll
  for (int b = 0; b < NB; b++) {
    uint4 o;
    o.x = mont_mul(redc64(acc[b][0], p, pinv), r2, p, pinv);
    o.y = mont_mul(redc64(acc[b][1], p, pinv), r2, p, pinv);
    o.z = mont_mul(redc64(acc[b][2], p, pinv), r2, p, pinv);
    o.w = mont_mul(redc64(acc[b][3], p, pinv), r2, p, pinv);
    reinterpret_cast<uint4*>(partial + (size_t)limb * partial_ps + (size_t)b * partial_bs)[((size_t)blockIdx.y * len + rho) * 16 + jq] = o;
  }
}

size_t axpy_num_chunks(size_t nterms) { return (nterms + AX_CHUNK - 1) / AX_CHUNK; }

template <typename MT>
static hipError_t launch_axpy_nb(int nb, uint32_t* partial, size_t partial_ps, size_t partial_bs, const uint32_t* G, size_t g_ps,
                                 size_t g_bs, const MT* C, size_t c_ps, size_t nterms, size_t len, int nlimbs,
                                 const LimbTable* tabs, hipStream_t s) {
  if (nterms == 0 || len == 0 || nlimbs == 0) return hipSuccess;
  if (limb_tables_wide(tabs)) {   // single-limb strict-32 path: (set, chunk, rho) outputs, chunks of AX_CHUNK terms
    if (sizeof(MT) != 4 || nlimbs != 1) return hipErrorInvalidValue;
    WideDot d{};
    d.out = partial; d.A = G; d.B = reinterpret_cast<const uint32_t*>(C); d.tab = tabs;
    d.n[0] = d.n[1] = 1; d.n[2] = (unsigned)nb; d.n[3] = (unsigned)axpy_num_chunks(nterms); d.n[4] = (unsigned)len;
    d.os[2] = partial_bs; d.os[3] = len * D; d.os[4] = D;
    d.as[2] = g_bs; d.as[3] = (size_t)AX_CHUNK * D; d.ak = D;
    d.bs[3] = (size_t)AX_CHUNK * len * D; d.bs[4] = D; d.bk = len * D;
    d.K = AX_CHUNK; d.Ktotal = (unsigned)nterms; d.chunk_level = 3;
    return launch_wide_dot(d, s);
  }
  dim3 grid((unsigned)((len + 15) / 16), (unsigned)axpy_num_chunks(nterms), (unsigned)nlimbs);
#define LAB_AX(NBV)                                                                                                   \
  hipLaunchKernelGGL((k_axpy_partial<MT, NBV>), grid, dim3(256), 0, s, partial, partial_ps, partial_bs, G, g_ps, g_bs, C, \
                     c_ps, (unsigned)nterms, (unsigned)len, tabs)
  switch (nb) {
    case 1: LAB_AX(1); break;
    case 2: LAB_AX(2); break;
    case 4: LAB_AX(4); break;
    case 8: LAB_AX(8); break;
    default: return hipErrorInvalidValue;
  }
#undef LAB_AX
  return hipGetLastError();
}

hipError_t launch_axpy_partial(HatView partial, ConstHatView G, ConstHatView C, size_t nterms, size_t len, int nlimbs,
                               const LimbTable* tabs, hipStream_t s) {
  return launch_axpy_nb<uint32_t>(1, partial.base, partial.plane_stride, 0, G.base, G.plane_stride, 0, C.base,
                                  C.plane_stride, nterms, len, nlimbs, tabs, s);
}

hipError_t launch_axpy_partial_u16(uint32_t* partial_plane, const uint32_t* G_plane, const uint16_t* C, size_t nterms,
                                   size_t len, const LimbTable* tab, hipStream_t s) {
  return launch_axpy_nb<uint16_t>(1, partial_plane, 0, 0, G_plane, 0, 0, C, 0, nterms, len, 1, tab, s);
}

// nb digit-vector sets sharing C (nb in {1,2,4,8}); *_bs = word stride between consecutive sets inside a plane
hipError_t launch_axpy_partial_batched(int nb, HatView partial, size_t partial_bs, ConstHatView G, size_t g_bs,
                                       ConstHatView C, size_t nterms, size_t len, int nlimbs, const LimbTable* tabs,
                                       hipStream_t s) {
  return launch_axpy_nb<uint32_t>(nb, partial.base, partial.plane_stride, partial_bs, G.base, G.plane_stride, g_bs, C.base,
                                  C.plane_stride, nterms, len, nlimbs, tabs, s);
}
hipError_t launch_axpy_partial_batched_u16(int nb, uint32_t* partial_plane, size_t partial_bs, const uint32_t* G_plane,
                                           size_t g_bs, const uint16_t* C, size_t nterms, size_t len, const LimbTable* tab,
                                           hipStream_t s) {
  return launch_axpy_nb<uint16_t>(nb, partial_plane, 0, partial_bs, G_plane, 0, g_bs, C, 0, nterms, len, 1, tab, s);
}

// u32 residues of one limb plane -> u16 (the limb prime is < 2^16)
__global__ void __launch_bounds__(256)
k_pack_u16(uint16_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t nquads) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nquads; i += stride) {
    const uint4 v = reinterpret_cast<const uint4*>(src)[i];
    reinterpret_cast<uint2*>(dst)[i] = make_uint2(v.x | (v.y << 16), v.z | (v.w << 16));
  }
}
__global__ void __launch_bounds__(256)
k_unpack_u16(uint32_t* __restrict__ dst, const uint16_t* __restrict__ src, size_t nquads) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nquads; i += stride) {
    const uint2 v = reinterpret_cast<const uint2*>(src)[i];
    reinterpret_cast<uint4*>(dst)[i] = make_uint4(v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16);
  }
}

hipError_t launch_pack_u16(uint16_t* dst, const uint32_t* src, size_t count, bool unpack, uint32_t* udst,
                           hipStream_t s) {
  if (count == 0) return hipSuccess;
  size_t nquads = count * D / 4;
  unsigned blocks = (unsigned)((nquads + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (!unpack) hipLaunchKernelGGL(k_pack_u16, dim3(blocks), dim3(256), 0, s, dst, src, nquads);
  else hipLaunchKernelGGL(k_unpack_u16, dim3(blocks), dim3(256), 0, s, udst, (const uint16_t*)dst, nquads);
  return hipGetLastError();
}

// out[k] = poly * v[k]  (poly_by_poly_vec, src/util.rs:289-296); plain products
template <bool WIDE>
__global__ void __launch_bounds__(256)
k_poly_by_vec(uint32_t* __restrict__ out, size_t out_ps, const uint32_t* __restrict__ poly, size_t poly_ps,
              const uint32_t* __restrict__ v, size_t v_ps, size_t nquads, const LimbTable* __restrict__ tabs, unsigned nlimbs,
              size_t js) {
  const int limb = blockIdx.y % nlimbs;
  const size_t job = blockIdx.y / nlimbs;
  const uint32_t p = tabs[limb].p, pinv = tabs[limb].pinv, r2 = tabs[limb].r2;
  const uint4* Pq = reinterpret_cast<const uint4*>(poly + (size_t)limb * poly_ps + job * js);
  const uint4* Vq = reinterpret_cast<const uint4*>(v + (size_t)limb * v_ps + job * js);
  uint4* O = reinterpret_cast<uint4*>(out + (size_t)limb * out_ps + job * js);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nquads; i += stride) {
    const uint4 a = Pq[i & 15], b = Vq[i];
    uint4 r;
    if constexpr (WIDE) {
      const ZpConsts c = zp_consts(tabs + limb);
      r.x = Zp<true>::mul(a.x, b.x, c); r.y = Zp<true>::mul(a.y, b.y, c); r.z = Zp<true>::mul(a.z, b.z, c); r.w = Zp<true>::mul(a.w, b.w, c);
      O[i] = r;
      continue;
    }
    r.x = mont_mul(mont_mul(a.x, b.x, p, pinv), r2, p, pinv);
    r.y = mont_mul(mont_mul(a.y, b.y, p, pinv), r2, p, pinv);
    r.z = mont_mul(mont_mul(a.z, b.z, p, pinv), r2, p, pinv);
    r.w = mont_mul(mont_mul(a.w, b.w, p, pinv), r2, p, pinv);
    O[i] = r;
  }
}

hipError_t launch_poly_by_vec(HatView out, ConstHatView poly, ConstHatView v, size_t len, int nlimbs,
                              const LimbTable* tabs, hipStream_t s, size_t njobs, size_t js) {
  if (len == 0 || njobs == 0) return hipSuccess;
  size_t nquads = len * D / 4;
  unsigned blocks = (unsigned)((nquads + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (limb_tables_wide(tabs))
    hipLaunchKernelGGL(k_poly_by_vec<true>, dim3(blocks, (unsigned)(nlimbs * njobs)), dim3(256), 0, s, out.base, out.plane_stride,
                       poly.base, poly.plane_stride, v.base, v.plane_stride, nquads, tabs, (unsigned)nlimbs, js);
  else
    hipLaunchKernelGGL(k_poly_by_vec<false>, dim3(blocks, (unsigned)(nlimbs * njobs)), dim3(256), 0, s, out.base, out.plane_stride,
                       poly.base, poly.plane_stride, v.base, v.plane_stride, nquads, tabs, (unsigned)nlimbs, js);
  return hipGetLastError();
}

}  // namespace lab

namespace lab {

// ------------------------------------------------ tiled Gram product with split reduction
// partial[chunk][i][k][j] = sum_{x in chunk} P[x][i][j] * S[x][k][j]
// Block = 8 waves = a 16 x 32 tile of outputs (lane = slot j, wave = 8 x 8 accumulators).  The 48 polynomials of a
// step x (16 rows of P, 32 columns of S) are fetched ONCE per block and shared through LDS, GL_XS steps per stage,
// double buffered: 16 LDS reads feed 64 multiply-adds, so the kernel runs at the v_mad_u64_u32 issue rate instead of
// the L2 -> L1 rate that a register-tile-only version (12 global loads per 32 multiply-adds) was bound by.  When
// P == S and one column tile covers all of S the rows are read from the column stage (no second fetch).  The reduction
// over x is cut into chunks for parallelism and summed by k_sum_terms.
constexpr int GR_CHUNK = 64;     // smallest chunk: gram_num_chunks(n) bounds the number of partial results
constexpr int GL_XS = 4, GL_TI = 16, GL_TK = 32, GL_NP = GL_TI + GL_TK;

template <bool SAME>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_gram_lds(uint32_t* __restrict__ partial, size_t partial_ps, const uint32_t* __restrict__ P, size_t p_ps,
           const uint32_t* __restrict__ S, size_t s_ps, unsigned n, unsigned r, unsigned nchunks, unsigned xchunk,
           const LimbTable* __restrict__ tabs, unsigned nlimbs, size_t js) {
  constexpr int NP = SAME ? GL_TK : GL_NP;                 // polynomials staged per step
  constexpr int QPT = GL_XS * NP * (D / 4) / 512;          // 16-byte pieces per thread and stage
  __shared__ __attribute__((aligned(16))) uint32_t lds[2][GL_XS * NP * D];
  const unsigned limb = (blockIdx.z / nchunks) % nlimbs, chunk = blockIdx.z % nchunks;
  const size_t job = blockIdx.z / (nchunks * nlimbs);
  partial += job * js + (size_t)limb * partial_ps;
  P += job * js + (size_t)limb * p_ps;
  S += job * js + (size_t)limb * s_ps;
  const uint32_t p = tabs[limb].p, p2 = tabs[limb].p2, pinv = tabs[limb].pinv, r2 = tabs[limb].r2;
  const int t = threadIdx.x, j = t & 63, w = t >> 6, wi = w >> 2, wk = w & 3;
  const unsigned i0 = blockIdx.x * GL_TI, k0 = blockIdx.y * GL_TK;
  const unsigned xbeg = chunk * xchunk, xend = min(xbeg + xchunk, n);
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

  // this thread's pieces of a stage: piece e = t + 512*q -> step e / (NP*16), polynomial (e / 16) % NP, quad e % 16
  const uint32_t* srcp[QPT];
  unsigned xs_of[QPT];
#pragma unroll
  for (int q = 0; q < QPT; q++) {
    const unsigned e = t + 512 * q, slot = (e / 16) % NP;
    xs_of[q] = e / (NP * 16);
    srcp[q] = slot < GL_TK ? S + (size_t)min(k0 + slot, r - 1) * D + (e % 16) * 4
                           : P + (size_t)min(i0 + slot - GL_TK, r - 1) * D + (e % 16) * 4;
  }
  u32x4_t pre[QPT];
  auto fetch = [&](unsigned x0) {
#pragma unroll
    for (int q = 0; q < QPT; q++) {
      const unsigned x = x0 + xs_of[q];
      const u32x4_t z = {0, 0, 0, 0};
      pre[q] = x < xend ? *reinterpret_cast<const u32x4_t*>(srcp[q] + (size_t)x * r * D) : z;
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int q = 0; q < QPT; q++) *reinterpret_cast<u32x4_t*>(&lds[buf][(t + 512 * q) * 4]) = pre[q];
  };
  // LDS word offsets of this wave's 8 rows and 8 columns inside one step of a stage
  unsigned prow[8], scol[8];
#pragma unroll
  for (int a = 0; a < 8; a++) {
    prow[a] = (SAME ? min(i0 + wi * 8 + a, r - 1) : GL_TK + wi * 8 + a) * D + j;
    scol[a] = (wk * 8 + a) * D + j;
  }
  uint64_t acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; a++)
#pragma unroll
    for (int b = 0; b < 8; b++) acc[a][b] = 0;

  fetch(xbeg);
  park(0);
  __syncthreads();
  int buf = 0;
  unsigned since_fold = 0;
  for (unsigned x = xbeg; x < xend; x += GL_XS, buf ^= 1) {
    const bool more = x + GL_XS < xend;
    if (more) fetch(x + GL_XS);
#pragma unroll
    for (int xs = 0; xs < GL_XS; xs++) {
      uint32_t pv[8], sv[8];
#pragma unroll
      for (int a = 0; a < 8; a++) pv[a] = lds[buf][xs * NP * D + prow[a]];
#pragma unroll
      for (int b = 0; b < 8; b++) sv[b] = lds[buf][xs * NP * D + scol[b]];
#pragma unroll
      for (int a = 0; a < 8; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) acc[a][b] += (uint64_t)pv[a] * sv[b];
    }
    since_fold += GL_XS;
    if (since_fold >= 8) {     // at most 8 products (< 2^60 each) on top of a folded accumulator
      since_fold = 0;
#pragma unroll
      for (int a = 0; a < 8; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) acc[a][b] = acc_fold(acc[a][b], p, p2);
    }
    if (more) park(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 8; a++)
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const unsigned i = i0 + wi * 8 + a, k = k0 + wk * 8 + b;
      if (i < r && k < r)
        partial[(((size_t)chunk * r + i) * r + k) * D + j] = mont_mul(redc64(acc_fold(acc[a][b], p, p2), p, pinv), r2, p, pinv);
    }
}

size_t gram_num_chunks(size_t n) { return (n + GR_CHUNK - 1) / GR_CHUNK; }

// chunks actually used for a launch: enough blocks for three rounds over the CUs, never more than gram_num_chunks(n)
size_t gram_chunks(size_t n, size_t r, int nlimbs, size_t njobs) {
  const size_t tiles = ((r + GL_TI - 1) / GL_TI) * ((r + GL_TK - 1) / GL_TK) * (size_t)nlimbs * njobs;
  size_t want = (768 + tiles - 1) / tiles, most = gram_num_chunks(n);
  if (want > most) want = most;
  if (want < 1) want = 1;
  const size_t xchunk = ((n + want - 1) / want + 7) / 8 * 8;
  return (n + xchunk - 1) / xchunk;
}

hipError_t launch_gram_tiled(HatView partial, ConstHatView P, ConstHatView S, size_t n, size_t r, int nlimbs,
                             const LimbTable* tabs, hipStream_t s, size_t njobs, size_t js) {
  if (r == 0 || n == 0 || njobs == 0) return hipSuccess;
  const size_t nchunks = gram_chunks(n, r, nlimbs, njobs);
  const size_t want = nchunks;
  const unsigned xchunk = (unsigned)(((n + want - 1) / want + 7) / 8 * 8);
  if (limb_tables_wide(tabs)) {   // single-limb strict-32 path: (job, chunk, i, k) outputs, chunks of xchunk terms
    if (nlimbs != 1) return hipErrorInvalidValue;
    WideDot d{};
    d.out = partial.base; d.A = P.base; d.B = S.base; d.tab = tabs;
    d.n[0] = 1; d.n[1] = (unsigned)njobs; d.n[2] = (unsigned)nchunks; d.n[3] = (unsigned)r; d.n[4] = (unsigned)r;
    d.os[1] = js; d.os[2] = r * r * D; d.os[3] = r * D; d.os[4] = D;
    d.as[1] = js; d.as[2] = (size_t)xchunk * r * D; d.as[3] = D; d.ak = r * D;
    d.bs[1] = js; d.bs[2] = (size_t)xchunk * r * D; d.bs[4] = D; d.bk = r * D;
    d.K = xchunk; d.Ktotal = (unsigned)n; d.chunk_level = 2;
    return launch_wide_dot(d, s);
  }
  dim3 grid((unsigned)((r + GL_TI - 1) / GL_TI), (unsigned)((r + GL_TK - 1) / GL_TK), (unsigned)(nlimbs * nchunks * njobs));
  const bool same = P.base == S.base && P.plane_stride == S.plane_stride && r <= (size_t)GL_TK;
  if (same)
    hipLaunchKernelGGL(k_gram_lds<true>, grid, dim3(512), 0, s, partial.base, partial.plane_stride, P.base, P.plane_stride, S.base,
                       S.plane_stride, (unsigned)n, (unsigned)r, (unsigned)nchunks, xchunk, tabs, (unsigned)nlimbs, js);
  else
    hipLaunchKernelGGL(k_gram_lds<false>, grid, dim3(512), 0, s, partial.base, partial.plane_stride, P.base, P.plane_stride, S.base,
                       S.plane_stride, (unsigned)n, (unsigned)r, (unsigned)nchunks, xchunk, tabs, (unsigned)nlimbs, js);
  return hipGetLastError();
}

}  // namespace lab
