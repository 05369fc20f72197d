// mfma_api.hip — host side of the int8-MFMA outer commitment (mfma_kernels.hip): the packed-matrix handle
// (lab_mat8), the stand-alone batched op lab_matvec_sum_mfma and the helper prove.hip calls.
#include "context.h"

using namespace lab;

struct lab_mat8 {
  lab_ctx* ctx = nullptr;
  int basis = 0;
  int nlimbs = 0;
  size_t nmat = 0, rows = 0, cols = 0;
  size_t rt_count = 0, kt_count = 0;      // 32-row tiles, 32-k tiles (kt padded to a multiple of 12)
  int planes[MAX_LIMBS] = {0, 0, 0, 0};   // signed base-256 digit planes per limb
  int8_t* d[MAX_LIMBS] = {nullptr, nullptr, nullptr, nullptr};
  size_t bytes = 0;
};

namespace lab {

int mfma_planes(uint32_t p) {   // digits d_i in [-128,127] (the last one unrestricted within i8) covering [0, p)
  int bits = 0;
  while (bits < 32 && ((uint64_t)1 << bits) < p) bits++;   // p - 1 < 2^bits
  return (bits + 1 + 7) / 8;
}

// column tiles for ns sets whose residues need `planes_v` digit planes
static size_t nt_for(int ns, int planes_v) { return ((size_t)ns * planes_v + 31) / 32; }

size_t mfma_scratch_bytes(const lab_mat8* m, int ns) {
  size_t v8 = 0, dm = 0;
  for (int l = 0; l < m->nlimbs; l++) {
    size_t nt = nt_for(ns, m->planes[l]);
    if (nt == 3) nt = 4;
    v8 = std::max(v8, (size_t)D * m->kt_count * nt * 1024);
    dm = std::max(dm, (size_t)D * m->planes[l] * m->rt_count * 32 * nt * 32 * 4);
  }
  return Carver::padded(v8) + Carver::padded(dm);
}

// out_s[rho] (s < ns_valid, hat elements at out.base + s*out_set_stride) = sum_k M[rho][k] V_s[k], V_s at V.base + s*v_set_stride;
// ns in {8,16,32} is the padded set count the GEMM runs with; `scratch` holds mfma_scratch_bytes(m, ns).
// When the measurement hook is on, the GEMM launches of the 4-plane (u32) limbs are bracketed by events.
int matvec_mfma_impl(lab_ctx* c, HatView out, size_t out_set_stride, const lab_mat8* m, ConstHatView V, size_t v_set_stride, int ns,
                     int ns_valid, void* scratch) {
  const uint32_t* primes = m->basis ? c->dg_primes : c->primes;
  const size_t ktotal = m->nmat * m->cols;
  Carver cv(scratch);
  size_t v8max = 0;
  for (int l = 0; l < m->nlimbs; l++) {
    size_t nt = nt_for(ns, m->planes[l]);
    if (nt == 3) nt = 4;
    v8max = std::max(v8max, (size_t)D * m->kt_count * nt * 1024);
  }
  int8_t* v8 = cv.take<int8_t>(v8max);
  int32_t* dm = reinterpret_cast<int32_t*>(cv.base + cv.off);
  for (int l = 0; l < m->nlimbs; l++) {
    const int pl = m->planes[l];
    size_t nt = nt_for(ns, pl);
    if (nt == 3) nt = 4;                       // the GEMM is instantiated for 1, 2, 4 column tiles
    LAB_HIP(c, launch_pack_vectors_i8(ns, ns_valid, v8, V.base + (size_t)l * V.plane_stride, v_set_stride, ktotal, m->kt_count, nt, pl,
                                      c->stream));
    const bool prof = c->prof_on && pl == 4;
    if (prof && c->prof_used == c->prof_events.size()) {
      hipEvent_t a, b;
      LAB_HIP(c, hipEventCreate(&a));
      LAB_HIP(c, hipEventCreate(&b));
      c->prof_events.emplace_back(a, b);
    }
    if (prof) LAB_HIP(c, hipEventRecord(c->prof_events[c->prof_used].first, c->stream));
    LAB_HIP(c, launch_gemm_i8(dm, m->d[l], v8, pl, m->rt_count, m->kt_count, nt, c->stream));
    if (prof) {
      LAB_HIP(c, hipEventRecord(c->prof_events[c->prof_used++].second, c->stream));
      c->prof_kernel = nt == 4 ? "k_gemm_i8<4>" : nt == 2 ? "k_gemm_i8<2>" : "k_gemm_i8<1>";
      // matrix digit planes once, the B-operand tiles once per 128-row block group (from L2), the int32 results
      c->prof_bytes += (double)D * pl * m->rt_count * m->kt_count * 1024 + (double)D * m->kt_count * nt * 1024 +
                       (double)D * pl * m->rt_count * 32 * nt * 32 * 4;
    }
    LAB_HIP(c, launch_gemm_i8_epilogue(out.base + (size_t)l * out.plane_stride, out_set_stride, dm, m->rows, m->rt_count, ns, ns_valid,
                                       nt * 32, pl, pl, primes[l], c->stream));
  }
  return LAB_OK;
}

}  // namespace lab

extern "C" int lab_mat8_pack(lab_ctx* c, const lab_vec* M, size_t m_off, size_t nmat, size_t rows, size_t cols, lab_mat8** out) {
  if (!c || !out) return LAB_ERR_ARG;
  int rc;
  if (nmat == 0 || rows == 0 || cols == 0) return fail(c, LAB_ERR_ARG, "lab_mat8_pack: empty dimension");
  if ((rc = check_range(c, M, m_off, nmat * rows * cols, "lab_mat8_pack(M)"))) return rc;
  if (nmat * cols >= (1u << 17)) return fail(c, LAB_ERR_RANGE, "lab_mat8_pack: %zu products per sum exceed the int32 exactness bound 131071 of the int8 formulation", nmat * cols);
  const uint64_t limit = M->basis ? c->dg_terms : c->max_len;
  if (nmat * cols > limit) return fail(c, LAB_ERR_RANGE, "lab_mat8_pack: reduction length %zu exceeds the exactness bound %llu of this basis", nmat * cols, (unsigned long long)limit);
  LAB_HIP(c, hipSetDevice(c->device));
  lab_mat8* m = new lab_mat8();
  m->ctx = c; m->basis = M->basis; m->nlimbs = M->nlimbs();
  m->nmat = nmat; m->rows = rows; m->cols = cols;
  m->rt_count = (rows + 31) / 32;
  m->kt_count = ((nmat * cols + 31) / 32 + 11) / 12 * 12;   // the GEMM walks K in stages of 4 k-tiles, 3 stages per loop trip
  const uint32_t* primes = M->basis ? c->dg_primes : c->primes;
  const int nfull = M->planes32();
  for (int l = 0; l < m->nlimbs; l++) {
    m->planes[l] = mfma_planes(primes[l]);
    const size_t bytes = (size_t)D * m->planes[l] * m->rt_count * m->kt_count * 1024;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d[l]), bytes);
    if (e != hipSuccess) {
      for (int k = 0; k < l; k++) (void)hipFree(m->d[k]);
      delete m;
      return fail(c, LAB_ERR_HIP, "lab_mat8_pack: hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    m->bytes += bytes;
    const bool u16 = M->compact && l >= nfull;
    const void* src = u16 ? static_cast<const void*>(M->d16 + m_off * D)
                          : static_cast<const void*>(M->d + (size_t)l * M->count * D + m_off * D);
    e = launch_pack_matrix_i8(m->d[l], src, u16, nmat, rows, cols, m->rt_count, m->kt_count, m->planes[l], c->stream);
    if (e != hipSuccess) {
      for (int k = 0; k <= l; k++) (void)hipFree(m->d[k]);
      delete m;
      return fail(c, LAB_ERR_HIP, "lab_mat8_pack: %s", hipGetErrorString(e));
    }
  }
  *out = m;
  return LAB_OK;
}

extern "C" int lab_mat8_destroy(lab_mat8* m) {
  if (!m) return LAB_OK;
  if (m->ctx) {
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
  }
  for (int l = 0; l < MAX_LIMBS; l++)
    if (m->d[l]) (void)hipFree(m->d[l]);
  delete m;
  return LAB_OK;
}

extern "C" size_t lab_mat8_bytes(const lab_mat8* m) { return m ? m->bytes : 0; }

namespace lab {
bool mat8_matches(const lab_mat8* m, int basis, size_t nmat, size_t rows, size_t cols) {
  return m && m->basis == basis && m->nmat == nmat && m->rows == rows && m->cols == cols;
}
int mat8_basis(const lab_mat8* m) { return m->basis; }
}  // namespace lab

extern "C" int lab_matvec_sum_mfma(lab_ctx* c, lab_vec* out, size_t out_off, const lab_mat8* m, const lab_vec* V, size_t v_off, int ns) {
  if (!c || !m) return LAB_ERR_ARG;
  if (ns != 8 && ns != 16 && ns != 32) return fail(c, LAB_ERR_ARG, "lab_matvec_sum_mfma: ns must be 8, 16 or 32");
  int rc;
  if ((rc = check_range(c, out, out_off, (size_t)ns * m->rows, "lab_matvec_sum_mfma(out)"))) return rc;
  if ((rc = check_range(c, V, v_off, (size_t)ns * m->nmat * m->cols, "lab_matvec_sum_mfma(V)"))) return rc;
  if (V->basis != m->basis || out->basis != m->basis || V->compact || out->compact)
    return fail(c, LAB_ERR_ARG, "lab_matvec_sum_mfma: V and out must be plain vectors in the packed matrix's basis");
  LAB_HIP(c, hipSetDevice(c->device));
  if ((rc = ensure_scratch(c, mfma_scratch_bytes(m, ns)))) return rc;
  return matvec_mfma_impl(c, out->view(out_off), m->rows * D, m, V->view(v_off), m->nmat * m->cols * D, ns, ns, c->scratch);
}
