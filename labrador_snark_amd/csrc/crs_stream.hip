// crs_stream.hip — the outer commitment with the CRS matrices STREAMED from the seed (SURVEY §8(f) row f1 as written):
//   out_s[rho] = sum_m < M_m[rho, :], V_s[m] >,    M_m[rho][c] = the polynomial whose 64 coefficients are
//   ChaCha20Rng::from_seed(base_seed + off_m + (rho*cols + c)*64 + k).gen_range(0..q),  k = 0..63
// — the rows CRS::fetch_B_ik_row regenerates on every fetch (src/structs.rs:35-45, 74-88, 147-171) — without the
// matrices ever being resident: at the reference's ranks one B_ik alone is 1 TiB.  The work is cut into slabs of R = 128
// matrix rows x MC matrices (MC * cols below the int32 bound of the int8 GEMM): a slab is expanded on the device
// (k_crs_expand), transformed into the basis of the vectors, packed as signed byte planes in MFMA operand order and
// multiplied with every vector set on the matrix cores (k_gemm_i8*); the partial results of the MC-chunks are added
// mod each limb prime — or, when the whole sum of nmat * cols terms exceeds the exactness bound of the basis but one chunk does
// not, every chunk's result is brought back to coefficients mod q (INTT + CRT of ns * 128 polynomials: nothing) and the
// chunks are added mod q: the basis then only has to hold ONE chunk's sum (three 23-bit primes instead of four at kappa = 2^14:
// a quarter less transform, packing and matrix-core work per regenerated coefficient).
// Only the slab (a fixed working set, independent of the matrix size) ever exists in HBM.
// The expansion is ~1100 integer operations per coefficient (x ~2 at the 33-bit modulus: half the candidates are
// rejected) against 27 byte products in the GEMM: this path is bound by the ChaCha20/VALU rate, not by HBM or MFMA.
#include "context.h"

using namespace lab;
typedef unsigned __int128 u128;

namespace lab {
int mfma_planes(uint32_t p);
int mat8_alloc(lab_ctx* c, int basis, size_t nmat_max, size_t rows, size_t cols, lab_mat8** out);
int mat8_repack(lab_ctx* c, lab_mat8* m, const uint32_t* hat_planes, size_t plane_stride, size_t nmat, size_t rows_valid);
bool mat8_can_fill_rows(const lab_mat8* m);
int mat8_fill_begin(lab_ctx* c, lab_mat8* m, size_t nmat, size_t rows_valid);
int mat8_fill_rows(lab_ctx* c, lab_mat8* m, const uint64_t* coefs, size_t m_idx, size_t row0, size_t nrows);

// polynomials per expansion launch: 2^20 x 64 coefficients = 8192 blocks x 8 passes of k_crs_expand — every block then works off
// its rejected coefficients once per EIGHT passes (launches of 2^18 polynomials gave each block two passes per queue phase and
// ran 19 % below the kernel's stand-alone rate)
constexpr size_t CRS_CHUNK_POLYS = (size_t)1 << 20;

struct CrsStream {
  lab_ctx* c = nullptr;
  int basis = 0;
  size_t rows = 0, cols = 0, R = 0, MC = 0, ns = 0;
  lab_mat8* m8 = nullptr;
  uint32_t* slab = nullptr;       // [L][MC*R*cols][64]
  uint64_t* coefs = nullptr;      // [chunk][64]
  // One pass over the regenerated rows serves up to CRS_MAX_GROUPS groups of vector sets (64 proofs each: the prover's batches of
  // more than 64 proofs share the expansion); per group:
  uint32_t* tmp_all = nullptr;          // [L][group][ns*R][64]: the (chunk, slab) results of every group
  uint32_t* tmp2 = nullptr;             // reduce: [L][ns*R][64] scratch of the inverse transform
  uint32_t* stage = nullptr;            // several groups in one GEMM launch: its slot-major staging array
  uint64_t* acc[CRS_MAX_GROUPS] = {};   // reduce: [ns][rows][64] coefficients mod q — the running sum over the K chunks
  uint64_t* cur = nullptr;              // reduce: [ns*R][64] the current (chunk, slab)'s sums mod q (shared: used and consumed in turn)
  size_t ngroups = 1;
  size_t chunk = 0;
  bool fused = false;             // slab tiles straight from the expanded coefficients (mat8_fill_rows)
  bool reduce = false;            // chunk results are added mod q in the coefficient domain (see the header)
  ~CrsStream() {
    if (m8) lab_mat8_destroy(m8);
    if (slab) (void)hipFree(slab);
    if (coefs) (void)hipFree(coefs);
    for (size_t g = 0; g < CRS_MAX_GROUPS; g++)
      if (acc[g]) (void)hipFree(acc[g]);
    if (tmp_all) (void)hipFree(tmp_all);
    if (tmp2) (void)hipFree(tmp2);
    if (stage) (void)hipFree(stage);
    if (cur) (void)hipFree(cur);
  }
};

// rows per slab / matrices per chunk the engine will use
// rows per slab: the GEMM over a slab is one block per NTT slot and four 32-row tiles — 128 rows are 64 blocks, a quarter of the
// chip's 256 CUs, and ran at 2.3 TB/s; 512 rows fill it.  Taken when the slab (its byte planes in operand order) stays below
// 40 GB and free memory holds it next to the engine's other buffers (`other_bytes`) and 1 GB to spare, else 256 / 128.
size_t crs_stream_rows(lab_ctx* c, int basis, size_t rows, size_t mc, size_t cols, size_t other_bytes) {
  if (rows <= 128) return rows;
  const int L = basis_nlimbs(c, basis);
  const size_t kt = ((mc * cols + 31) / 32 + 11) / 12 * 12;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
  const uint32_t* primes = basis ? c->dg_primes : c->primes;
  size_t planes_total = 0;                      // signed byte planes of all limbs (3 per 23-bit prime, 4 per 30-bit one)
  for (int l = 0; l < L; l++) planes_total += (size_t)std::min(4, mfma_planes(primes[l]));
  for (size_t R = 512; R > 128; R /= 2) {
    if (rows < R) continue;
    size_t slab = (size_t)planes_total * D * (R / 32) * kt * 1024;
    // without the one-pass coefficients -> tiles kernel (3-plane limbs, cols % 64 == 0) the slab also exists as residues
    if (planes_total != (size_t)3 * L || cols % 64) slab += (size_t)L * mc * R * cols * D * 4;
    if (slab <= ((size_t)40 << 30) && slab + other_bytes + ((size_t)1 << 30) <= free_b) return R;
  }
  return 128;
}
// matrices per K chunk: the int32 accumulators of the int8 GEMM hold 2^17 - 2 terms; a sum that (together with `extra_terms` the
// caller adds to the result in the transform domain) exceeds the exactness bound of the basis is cut into chunks that respect it,
// and their results are added mod q, not mod the limb primes (*reduce).  false: not even one matrix fits.
bool crs_stream_plan(const lab_ctx* c, int basis, size_t nmat, size_t cols, uint64_t extra_terms, size_t* mc_out, bool* reduce) {
  if (cols == 0 || nmat == 0) return false;
  const uint64_t limit = basis ? c->dg_terms : c->max_len;
  size_t mc = ((1u << 17) - 2) / cols;
  *reduce = (unsigned __int128)nmat * cols + extra_terms > limit;
  if (*reduce) {
    if (extra_terms >= limit) return false;
    if ((limit - extra_terms) / cols < mc) mc = (size_t)((limit - extra_terms) / cols);
  }
  if (mc > nmat) mc = nmat;
  *mc_out = mc;
  return mc > 0;
}
size_t crs_stream_scratch_bytes(lab_ctx* c, int basis, size_t nmat, size_t rows, size_t cols, int ns, uint64_t extra_terms) {
  // the V8 tiles (or prepacked digit tiles) of one chunk: sized like a packed matrix of MC x cols
  size_t mc = 0;
  bool reduce = false;
  if (!crs_stream_plan(c, basis, nmat, cols, extra_terms, &mc, &reduce)) return 0;
  const size_t kt = ((mc * cols + 31) / 32 + 11) / 12 * 12;
  const int L = basis_nlimbs(c, basis);
  return (size_t)L * Carver::padded((size_t)D * kt * 8 * 1024) + 8192;   // 8 column tiles at most per limb
}

int crs_stream_create(lab_ctx* c, int basis, size_t nmat, size_t rows, size_t cols, int ns, CrsStream** out, uint64_t extra_terms,
                      size_t ngroups) {
  if (ngroups == 0 || ngroups > CRS_MAX_GROUPS) return fail(c, LAB_ERR_ARG, "streamed commitment: %zu groups of vector sets (1..%d)", ngroups, CRS_MAX_GROUPS);
  CrsStream* st = new CrsStream();
  st->c = c; st->basis = basis; st->rows = rows; st->cols = cols; st->ns = (size_t)ns; st->ngroups = ngroups;
  if (!crs_stream_plan(c, basis, nmat, cols, extra_terms, &st->MC, &st->reduce)) st->MC = 0;
  if (st->MC == 0) { delete st; return fail(c, LAB_ERR_RANGE, "streamed commitment: %zu columns per matrix exceed the int32 bound of the int8 GEMM or the exactness bound of this basis", cols); }
  {   // what the engine allocates besides the slab: the expansion chunk, per group the chunk results (twice with `reduce`) and sums
    const int Lb = basis_nlimbs(c, basis);
    const size_t other = CRS_CHUNK_POLYS * D * 8 + ngroups * ((size_t)2 * Lb * ns * 512 * D * 4 + (st->reduce ? (size_t)ns * rows * D * 8 : 0)) +
                         (size_t)ns * 512 * D * 8;
    st->R = crs_stream_rows(c, basis, rows, st->MC, cols, other);
  }
  const int L = basis_nlimbs(c, basis);
  const size_t slab_elems = st->MC * st->R * cols;
  st->chunk = slab_elems < (CRS_CHUNK_POLYS) ? slab_elems : (CRS_CHUNK_POLYS);
  int rc = mat8_alloc(c, basis, st->MC, st->R, cols, &st->m8);
  if (rc) { delete st; return rc; }
  st->fused = mat8_can_fill_rows(st->m8) && c->q < (1ull << 40);
  hipError_t e = hipSuccess;
  if (st->fused) {   // expansion chunks of whole rows; no slab of residues
    const size_t rows_per = std::max<size_t>(1, ((CRS_CHUNK_POLYS) / cols));
    st->chunk = std::min(rows_per, st->R) * cols;
  } else {
    e = hipMalloc(reinterpret_cast<void**>(&st->slab), (size_t)L * slab_elems * D * 4);
  }
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&st->coefs), st->chunk * D * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&st->tmp_all), (size_t)L * ngroups * ns * st->R * D * 4);
  // (rows past a ragged slab and sets past ns_valid are never stored by the GEMM but are read by the chunk reduction)
  if (e == hipSuccess) e = hipMemsetAsync(st->tmp_all, 0, (size_t)L * ngroups * ns * st->R * D * 4, c->stream);
  if (e == hipSuccess && st->reduce) e = hipMalloc(reinterpret_cast<void**>(&st->tmp2), (size_t)L * ns * st->R * D * 4);
  if (e == hipSuccess && ngroups > 1) e = hipMalloc(reinterpret_cast<void**>(&st->stage), mfma_stage_bytes(st->m8, ns, ngroups));
  for (size_t g = 0; g < ngroups; g++)
    if (e == hipSuccess && st->reduce) e = hipMalloc(reinterpret_cast<void**>(&st->acc[g]), (size_t)ns * rows * D * 8);
  if (e == hipSuccess && st->reduce) e = hipMalloc(reinterpret_cast<void**>(&st->cur), (size_t)ns * st->R * D * 8);
  if (e != hipSuccess) { delete st; return fail(c, LAB_ERR_HIP, "streamed commitment: hipMalloc of the slab: %s", hipGetErrorString(e)); }
  *out = st;
  return LAB_OK;
}
void crs_stream_destroy(CrsStream* st) { delete st; }

// expand rows [rho0, rho0 + nr) of matrices [m0, m0 + mc) into the slab (transform domain) and pack it
static int fill_slab(CrsStream* st, const uint8_t seed[32], const uint64_t* mat_offsets, size_t m0, size_t mc, size_t rho0, size_t nr) {
  lab_ctx* c = st->c;
  const int L = basis_nlimbs(c, st->basis);
  const size_t cols = st->cols, slab_elems = st->MC * st->R * cols;
  if (st->fused) {
    // coefficients -> tiles in one pass (k_coefs_to_a8): whole rows per expansion chunk, no transform-domain slab
    int rc = mat8_fill_begin(c, st->m8, mc, nr);
    if (rc) return rc;
    const size_t rows_per = st->chunk / cols;
    for (size_t m = 0; m < mc; m++) {
      const u128 base = (((u128)mat_offsets[2 * (m0 + m) + 1] << 64) | mat_offsets[2 * (m0 + m)]) + (u128)rho0 * cols * D;
      for (size_t r0 = 0; r0 < nr; r0 += rows_per) {
        const size_t rows = nr - r0 < rows_per ? nr - r0 : rows_per;
        const u128 off = base + (u128)r0 * cols * D;
        {
          PhaseSample ps(c, PH_EXPAND, (double)rows * cols * D);
          LAB_HIP(c, launch_crs_expand(st->coefs, rows * cols, seed, (uint64_t)off, (uint64_t)(off >> 64), c->q, c->stream));
        }
        PhaseSample ps(c, PH_TILES, (double)rows * cols * D);
        if ((rc = mat8_fill_rows(c, st->m8, st->coefs, m, r0, rows))) return rc;
      }
    }
    return LAB_OK;
  }
  if (nr < st->R)   // ragged last slab: rows past the end read as zero
    LAB_HIP(c, hipMemsetAsync(st->slab, 0, (size_t)L * slab_elems * D * 4, c->stream));
  for (size_t m = 0; m < mc; m++) {
    // rows rho0.. of matrix m0+m are one contiguous range of seeds: off_m + rho0*cols*64, nr*cols polynomials
    const u128 base = (((u128)mat_offsets[2 * (m0 + m) + 1] << 64) | mat_offsets[2 * (m0 + m)]) + (u128)rho0 * cols * D;
    const size_t total = nr * cols;
    for (size_t done = 0; done < total; done += st->chunk) {
      const size_t cnt = total - done < st->chunk ? total - done : st->chunk;
      const u128 off = base + (u128)done * D;
      {
        PhaseSample ps(c, PH_EXPAND, (double)cnt * D);
        LAB_HIP(c, launch_crs_expand(st->coefs, cnt, seed, (uint64_t)off, (uint64_t)(off >> 64), c->q, c->stream));
      }
      // element (m, row, col) of the slab at (m*R + row)*cols + col
      HatView dst{st->slab + ((m * st->R) * cols + done) * D, slab_elems * D};
      PhaseSample ps(c, PH_TILES, (double)cnt * D);
      int rc = to_hat(c, st->coefs, dst, cnt, st->basis);
      if (rc) return rc;
    }
  }
  PhaseSample ps(c, PH_TILES, 0.0);
  return mat8_repack(c, st->m8, st->slab, slab_elems * D, mc, nr);
}

// The engine.  Vector sets either as transform-domain vectors (V: set s at V.base + s*v_set_stride words, element
// (m, c) at + (m*cols + c)*64) or as 16-bit decomposition digits per set (dig.p[s]: [nmat*cols][64] words, the prover's
// step 3).  out: set s at out.base + s*out_set_stride.  `scratch` >= crs_stream_scratch_bytes().
// Loop order: K chunk outside, row slab inside — the vector operand of a chunk (the digits of t transformed and packed as
// B-operand tiles of every limb) is made ONCE per chunk and multiplied with every slab of that chunk's matrices (slab outside
// re-made it for each of the kappa1 / 128 slabs: 5.6 % of a kappa = 2^14 step).
int crs_stream_run(CrsStream* st, const uint8_t seed[32], const uint64_t* mat_offsets, size_t nmat, const CrsGroup* groups, size_t ngroups) {
  lab_ctx* c = st->c;
  const int L = basis_nlimbs(c, st->basis);
  const LimbTable* tabs = basis_tabs(c, st->basis);
  const size_t cols = st->cols, rows = st->rows;
  if (ngroups == 0 || ngroups > st->ngroups) return fail(c, LAB_ERR_ARG, "streamed commitment: %zu groups, the engine was created for %zu", ngroups, st->ngroups);
  for (size_t g = 0; g < ngroups; g++)
    if ((size_t)groups[g].ns > st->ns || groups[g].ns_valid > groups[g].ns) return fail(c, LAB_ERR_ARG, "streamed commitment: group %zu has more sets than the engine was created for", g);
  if (st->reduce)
    for (size_t g = 0; g < ngroups; g++) LAB_HIP(c, hipMemsetAsync(st->acc[g], 0, (size_t)groups[g].ns * rows * D * 8, c->stream));
  // every group's (chunk, slab) results land in its slice of tmp_all ([L][group][ns*R][64])
  const size_t gstride = st->ns * st->R * D, tplane = st->ngroups * gstride;
  auto tmp_view = [&](size_t gi) { return HatView{st->tmp_all + gi * gstride, tplane}; };
  // Several groups (the sub-batches of a call of more than 64 proofs) go through ONE GEMM launch per limb as its jobs when they
  // can: the slab is then read from HBM once for all of them (the other jobs' blocks of a slot and row group hit L2) instead of
  // once per group.  Needs digit tiles (each group's in its own scratch, equally spaced), equal padded set counts, and the
  // coefficient-domain accumulation (nothing is written straight into a group's output).
  bool multi = ngroups >= 2 && st->stage && st->reduce;
  int ns_valid_max = 0;
  for (size_t g = 0; g < ngroups && multi; g++) {
    multi = (groups[g].dig || groups[g].t64) && (size_t)groups[g].ns == st->ns &&
            static_cast<char*>(groups[g].scratch) - static_cast<char*>(groups[0].scratch) ==
                (ptrdiff_t)g * (static_cast<char*>(groups[1].scratch) - static_cast<char*>(groups[0].scratch)) &&
            static_cast<char*>(groups[1].scratch) > static_cast<char*>(groups[0].scratch);
    ns_valid_max = groups[g].ns_valid > ns_valid_max ? groups[g].ns_valid : ns_valid_max;
  }
  for (size_t m0 = 0; m0 < nmat; m0 += st->MC) {
    const size_t mc = nmat - m0 < st->MC ? nmat - m0 : st->MC;
    const bool first = m0 == 0;
    bool tiles_made = false;
    for (size_t rho0 = 0; rho0 < rows; rho0 += st->R) {
      const size_t nr = rows - rho0 < st->R ? rows - rho0 : st->R;
      int rc = fill_slab(st, seed, mat_offsets, m0, mc, rho0, nr);
      if (rc) return rc;
      if (!tiles_made)
        for (size_t gi = 0; gi < ngroups; gi++) {
          const CrsGroup& G = groups[gi];
          if (!G.dig && !G.t64) continue;
          PhaseSample ps(c, PH_DIGITS, (double)G.ns_valid * mc * cols * D);
          if (G.t64) {   // the digits of this chunk's matrices straight from the sets' 64-bit vectors
            DigitsFromT dt = G.dt;
            dt.m0 = m0;
            if (dt.kappa != cols) return fail(c, LAB_ERR_ARG, "streamed commitment: digits on the fly need vectors of the matrices' column count");
            if ((rc = digits_to_v8_impl(c, st->m8, *G.t64, G.ns, G.ns_valid, static_cast<int8_t*>(G.scratch), &dt))) return rc;
            continue;
          }
          JobPtrs d;
          for (int s = 0; s < MAX_JOBS; s++)
            d.p[s] = (s < G.ns_valid && G.dig->p[s]) ? static_cast<void*>(static_cast<uint16_t*>(G.dig->p[s]) + m0 * cols * D) : nullptr;
          if ((rc = digits_to_v8_impl(c, st->m8, d, G.ns, G.ns_valid, static_cast<int8_t*>(G.scratch)))) return rc;
        }
      if (multi) {
        const MfmaJobBuffers jb{(size_t)(static_cast<char*>(groups[1].scratch) - static_cast<char*>(groups[0].scratch)), st->stage};
        PhaseSample ps(c, PH_GEMM, mfma_gemm_bytes(st->m8, (int)st->ns, ns_valid_max) * (double)ngroups);
        if ((rc = matvec_mfma_impl(c, tmp_view(0), st->R * D, st->m8, ConstHatView(), 0, (int)st->ns, ns_valid_max, groups[0].scratch, D, ngroups,
                                   gstride, 0, false, nullptr, static_cast<const int8_t*>(groups[0].scratch), nullptr, nullptr, 0, &jb))) return rc;
      }
      for (size_t gi = 0; gi < ngroups; gi++) {
        const CrsGroup& G = groups[gi];
        const int ns = G.ns, ns_valid = G.ns_valid;
        const bool direct = first && !st->reduce;
        // (a group's slice of tmp_all has the engine's set count st->ns as its set pitch only through R * D: sets are R * D apart)
        HatView dst = direct ? HatView{G.out.base + rho0 * D, G.out.plane_stride} : tmp_view(gi);
        const size_t dst_ss = direct ? G.out_set_stride : st->R * D;
        if (!multi) {
          PhaseSample ps(c, PH_GEMM, mfma_gemm_bytes(st->m8, ns, ns_valid));
          if (G.dig || G.t64 || tiles_made) {
            if ((rc = matvec_mfma_impl(c, dst, dst_ss, st->m8, ConstHatView(), 0, ns, ns_valid, G.scratch, D, 1, 0, 0, false, nullptr,
                                       static_cast<const int8_t*>(G.scratch)))) return rc;
          } else {   // first slab of the chunk: pack the transform-domain vectors, keep the tiles of every limb for the other slabs
            ConstHatView v(G.V->base + m0 * cols * D, G.V->plane_stride);
            if ((rc = matvec_mfma_impl(c, dst, dst_ss, st->m8, v, G.v_set_stride, ns, ns_valid, G.scratch, D, 1, 0, 0, false, nullptr, nullptr, nullptr,
                                       static_cast<int8_t*>(G.scratch)))) return rc;
          }
        }
        if (st->reduce) {
          // this (chunk, slab)'s exact sums -> coefficients mod q, added mod q to the rows of the running sum
          const size_t cnt = (size_t)ns * st->R;
          uint64_t* cur = st->cur;
          if ((rc = from_hat(c, ConstHatView(tmp_view(gi)), cnt, st->tmp2, cur, st->basis))) return rc;
          JobPtrs o, b;
          JobU64 zero;
          for (int s = 0; s < MAX_JOBS; s++) {
            o.p[s] = s < ns_valid ? st->acc[gi] + ((size_t)s * rows + rho0) * D : nullptr;
            b.p[s] = s < ns_valid ? cur + (size_t)s * st->R * D : nullptr;
            zero.v[s] = 0;
          }
          LAB_HIP(c, launch_coef_elementwise_jobs(o, o, b, nr, c->q, zero, 2, (size_t)ns_valid, c->stream));
        } else if (!first)
          LAB_HIP(c, launch_pointwise(PW_ADD, HatView{G.out.base + rho0 * D, G.out.plane_stride}, HatView{G.out.base + rho0 * D, G.out.plane_stride},
                                      tmp_view(gi), nr, L, tabs, c->stream, (size_t)ns_valid, G.out_set_stride, G.out_set_stride, st->R * D));
      }
      tiles_made = true;
    }
  }
  if (st->reduce)   // the sums mod q back into the transform domain, set by set
    for (size_t gi = 0; gi < ngroups; gi++)
      for (int s = 0; s < groups[gi].ns_valid; s++) {
        int rc = to_hat(c, st->acc[gi] + (size_t)s * rows * D,
                        HatView{groups[gi].out.base + (size_t)s * groups[gi].out_set_stride, groups[gi].out.plane_stride}, rows, st->basis);
        if (rc) return rc;
      }
  return LAB_OK;
}

}  // namespace lab

extern "C" int lab_crs_matvec_mfma(lab_ctx* c, lab_vec* out, size_t out_off, const uint8_t base_seed[32], const uint64_t* mat_offsets,
                                   size_t nmat, size_t rows, size_t cols, const lab_vec* V, size_t v_off, int ns) {
  if (!c) return LAB_ERR_ARG;
  if (!base_seed || !mat_offsets || nmat == 0 || rows == 0 || cols == 0) return fail(c, LAB_ERR_ARG, "lab_crs_matvec_mfma: bad argument");
  if (ns != 8 && ns != 16 && ns != 32 && ns != 64) return fail(c, LAB_ERR_ARG, "lab_crs_matvec_mfma: ns must be 8, 16, 32 or 64");
  int rc;
  if ((rc = check_range(c, out, out_off, (size_t)ns * rows, "lab_crs_matvec_mfma(out)"))) return rc;
  if ((rc = check_range(c, V, v_off, (size_t)ns * nmat * cols, "lab_crs_matvec_mfma(V)"))) return rc;
  if (V->basis != out->basis || V->compact || out->compact) return fail(c, LAB_ERR_ARG, "lab_crs_matvec_mfma: V and out must be plain vectors of one basis");
  const uint64_t limit = V->basis ? c->dg_terms : c->max_len;
  if (cols > limit) return fail(c, LAB_ERR_RANGE, "lab_crs_matvec_mfma: one matrix row of %zu terms exceeds the exactness bound %llu of this basis", cols, (unsigned long long)limit);
  LAB_HIP(c, hipSetDevice(c->device));
  if ((rc = ensure_scratch(c, crs_stream_scratch_bytes(c, V->basis, nmat, rows, cols, ns)))) return rc;
  CrsStream* st = nullptr;
  if ((rc = crs_stream_create(c, V->basis, nmat, rows, cols, ns, &st))) return rc;
  ConstHatView v = V->view(v_off);
  CrsGroup grp{out->view(out_off), rows * D, nullptr, &v, nmat * cols * D, ns, ns, c->scratch};
  rc = crs_stream_run(st, base_seed, mat_offsets, nmat, &grp, 1);
  if (rc == LAB_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, LAB_ERR_HIP, "lab_crs_matvec_mfma: stream synchronisation failed");
  crs_stream_destroy(st);
  return rc;
}
