// crs_stream.hip — the outer commitment with the CRS matrices STREAMED from the seed (SURVEY §8(f) row f1 as written):
//   out_s[rho] = sum_m < M_m[rho, :], V_s[m] >,    M_m[rho][c] = the polynomial whose 64 coefficients are
//   ChaCha20Rng::from_seed(base_seed + off_m + (rho*cols + c)*64 + k).gen_range(0..q),  k = 0..63
// — the rows CRS::fetch_B_ik_row regenerates on every fetch (src/structs.rs:35-45, 74-88, 147-171) — without the
// matrices ever being resident: at the reference's ranks one B_ik alone is 1 TiB.  The work is cut into slabs of R = 128
// matrix rows x MC matrices (MC * cols below the int32 bound of the int8 GEMM): a slab is expanded on the device
// (k_crs_expand), transformed into the basis of the vectors, packed as signed byte planes in MFMA operand order and
// multiplied with every vector set on the matrix cores (k_gemm_i8*); the partial results of the MC-chunks are added
// mod each limb prime.  Only the slab (a fixed working set, independent of the matrix size) ever exists in HBM.
// The expansion is ~1100 integer operations per coefficient (x ~2 at the 33-bit modulus: half the candidates are
// rejected) against 27 byte products in the GEMM: this path is bound by the ChaCha20/VALU rate, not by HBM or MFMA.
#include "context.h"

using namespace lab;
typedef unsigned __int128 u128;

namespace lab {
int mat8_alloc(lab_ctx* c, int basis, size_t nmat_max, size_t rows, size_t cols, lab_mat8** out);
int mat8_repack(lab_ctx* c, lab_mat8* m, const uint32_t* hat_planes, size_t plane_stride, size_t nmat, size_t rows_valid);
bool mat8_can_fill_rows(const lab_mat8* m);
int mat8_fill_begin(lab_ctx* c, lab_mat8* m, size_t nmat, size_t rows_valid);
int mat8_fill_rows(lab_ctx* c, lab_mat8* m, const uint64_t* coefs, size_t m_idx, size_t row0, size_t nrows);

struct CrsStream {
  lab_ctx* c = nullptr;
  int basis = 0;
  size_t rows = 0, cols = 0, R = 0, MC = 0, ns = 0;
  lab_mat8* m8 = nullptr;
  uint32_t* slab = nullptr;       // [L][MC*R*cols][64]
  uint64_t* coefs = nullptr;      // [chunk][64]
  uint32_t* tmp = nullptr;        // [L][ns*R][64]
  size_t chunk = 0;
  bool fused = false;             // slab tiles straight from the expanded coefficients (mat8_fill_rows)
  ~CrsStream() {
    if (m8) lab_mat8_destroy(m8);
    if (slab) (void)hipFree(slab);
    if (coefs) (void)hipFree(coefs);
    if (tmp) (void)hipFree(tmp);
  }
};

// rows per slab / matrices per chunk the engine will use
size_t crs_stream_rows(size_t rows) { return rows < 128 ? rows : 128; }
size_t crs_stream_mats(size_t nmat, size_t cols) {
  size_t mc = cols ? ((1u << 17) - 2) / cols : 0;
  return mc > nmat ? nmat : mc;
}
size_t crs_stream_scratch_bytes(lab_ctx* c, int basis, size_t nmat, size_t rows, size_t cols, int ns) {
  // the V8 tiles (or prepacked digit tiles) of one chunk: sized like a packed matrix of MC x cols
  const size_t mc = crs_stream_mats(nmat, cols);
  const size_t kt = ((mc * cols + 31) / 32 + 11) / 12 * 12;
  const int L = basis_nlimbs(c, basis);
  return (size_t)L * Carver::padded((size_t)D * kt * 8 * 1024) + 8192;   // 8 column tiles at most per limb
}

int crs_stream_create(lab_ctx* c, int basis, size_t nmat, size_t rows, size_t cols, int ns, CrsStream** out) {
  CrsStream* st = new CrsStream();
  st->c = c; st->basis = basis; st->rows = rows; st->cols = cols; st->ns = (size_t)ns;
  st->R = crs_stream_rows(rows);
  st->MC = crs_stream_mats(nmat, cols);
  if (st->MC == 0) { delete st; return fail(c, LAB_ERR_RANGE, "streamed commitment: %zu columns per matrix exceed the int32 bound of the int8 GEMM", cols); }
  const int L = basis_nlimbs(c, basis);
  const size_t slab_elems = st->MC * st->R * cols;
  st->chunk = slab_elems < ((size_t)1 << 18) ? slab_elems : ((size_t)1 << 18);
  int rc = mat8_alloc(c, basis, st->MC, st->R, cols, &st->m8);
  if (rc) { delete st; return rc; }
  st->fused = mat8_can_fill_rows(st->m8) && c->q < (1ull << 40);
  hipError_t e = hipSuccess;
  if (st->fused) {   // expansion chunks of whole rows; no slab of residues
    const size_t rows_per = std::max<size_t>(1, (((size_t)1 << 18) / cols));
    st->chunk = std::min(rows_per, st->R) * cols;
  } else {
    e = hipMalloc(reinterpret_cast<void**>(&st->slab), (size_t)L * slab_elems * D * 4);
  }
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&st->coefs), st->chunk * D * 8);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&st->tmp), (size_t)L * ns * st->R * D * 4);
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
        LAB_HIP(c, launch_crs_expand(st->coefs, rows * cols, seed, (uint64_t)off, (uint64_t)(off >> 64), c->q, c->stream));
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
      LAB_HIP(c, launch_crs_expand(st->coefs, cnt, seed, (uint64_t)off, (uint64_t)(off >> 64), c->q, c->stream));
      // element (m, row, col) of the slab at (m*R + row)*cols + col
      HatView dst{st->slab + ((m * st->R) * cols + done) * D, slab_elems * D};
      int rc = to_hat(c, st->coefs, dst, cnt, st->basis);
      if (rc) return rc;
    }
  }
  return mat8_repack(c, st->m8, st->slab, slab_elems * D, mc, nr);
}

// The engine.  Vector sets either as transform-domain vectors (V: set s at V.base + s*v_set_stride words, element
// (m, c) at + (m*cols + c)*64) or as 16-bit decomposition digits per set (dig.p[s]: [nmat*cols][64] words, the prover's
// step 3).  out: set s at out.base + s*out_set_stride.  `scratch` >= crs_stream_scratch_bytes().
int crs_stream_run(CrsStream* st, const uint8_t seed[32], const uint64_t* mat_offsets, size_t nmat, HatView out, size_t out_set_stride,
                   const ConstHatView* V, size_t v_set_stride, const JobPtrs* dig, int ns, int ns_valid, void* scratch) {
  lab_ctx* c = st->c;
  const int L = basis_nlimbs(c, st->basis);
  const LimbTable* tabs = basis_tabs(c, st->basis);
  const size_t cols = st->cols;
  for (size_t rho0 = 0; rho0 < st->rows; rho0 += st->R) {
    const size_t nr = st->rows - rho0 < st->R ? st->rows - rho0 : st->R;
    for (size_t m0 = 0; m0 < nmat; m0 += st->MC) {
      const size_t mc = nmat - m0 < st->MC ? nmat - m0 : st->MC;
      int rc = fill_slab(st, seed, mat_offsets, m0, mc, rho0, nr);
      if (rc) return rc;
      const bool first = m0 == 0;
      HatView dst = first ? HatView{out.base + rho0 * D, out.plane_stride} : HatView{st->tmp, (size_t)ns * st->R * D};
      const size_t dst_ss = first ? out_set_stride : st->R * D;
      if (dig) {
        JobPtrs d;
        for (int s = 0; s < MAX_JOBS; s++)
          d.p[s] = (s < ns_valid && dig->p[s]) ? static_cast<void*>(static_cast<uint16_t*>(dig->p[s]) + m0 * cols * D) : nullptr;
        if ((rc = digits_to_v8_impl(c, st->m8, d, ns, ns_valid, static_cast<int8_t*>(scratch)))) return rc;
        if ((rc = matvec_mfma_impl(c, dst, dst_ss, st->m8, ConstHatView(), 0, ns, ns_valid, scratch, D, 1, 0, 0, false, nullptr,
                                   static_cast<const int8_t*>(scratch)))) return rc;
      } else {
        ConstHatView v(V->base + m0 * cols * D, V->plane_stride);
        if ((rc = matvec_mfma_impl(c, dst, dst_ss, st->m8, v, v_set_stride, ns, ns_valid, scratch, D, 1, 0, 0))) return rc;
      }
      if (!first)
        LAB_HIP(c, launch_pointwise(PW_ADD, HatView{out.base + rho0 * D, out.plane_stride}, HatView{out.base + rho0 * D, out.plane_stride},
                                    HatView{st->tmp, (size_t)ns * st->R * D}, nr, L, tabs, c->stream, (size_t)ns_valid, out_set_stride,
                                    out_set_stride, st->R * D));
    }
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
  if ((u128)nmat * cols > limit) return fail(c, LAB_ERR_RANGE, "lab_crs_matvec_mfma: reduction length %zu exceeds the exactness bound %llu of this basis", nmat * cols, (unsigned long long)limit);
  LAB_HIP(c, hipSetDevice(c->device));
  if ((rc = ensure_scratch(c, crs_stream_scratch_bytes(c, V->basis, nmat, rows, cols, ns)))) return rc;
  CrsStream* st = nullptr;
  if ((rc = crs_stream_create(c, V->basis, nmat, rows, cols, ns, &st))) return rc;
  ConstHatView v = V->view(v_off);
  rc = crs_stream_run(st, base_seed, mat_offsets, nmat, out->view(out_off), rows * D, &v, nmat * cols * D, nullptr, ns, ns, c->scratch);
  if (rc == LAB_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, LAB_ERR_HIP, "lab_crs_matvec_mfma: stream synchronisation failed");
  crs_stream_destroy(st);
  return rc;
}
