// mfma_api.hip — host side of the int8-MFMA outer commitment (mfma_kernels.hip): the packed-matrix handle
// (lab_mat8), the stand-alone batched op lab_matvec_sum_mfma and the helper prove.hip calls.
#include <cstdio>
#include <cstdlib>

#include "context.h"

using namespace lab;

struct lab_mat8 {
  lab_ctx* ctx = nullptr;
  int basis = 0;
  int nlimbs = 0;
  size_t nmat = 0, rows = 0, cols = 0;
  size_t rows_alloc = 0;                  // mat8_alloc: rows of the buffer layout (`rows` = those the current contents fill)
  size_t rt_count = 0, kt_count = 0;      // 32-row tiles, 32-k tiles (kt padded to a multiple of 12)
  int planes[MAX_LIMBS] = {0, 0, 0, 0};   // signed base-256 digit planes per limb
  int8_t* d[MAX_LIMBS] = {nullptr, nullptr, nullptr, nullptr};
  size_t bytes = 0;
  bool filled = false;                    // mat8_fill_begin has cleared the tiles once
};

namespace lab {

int mfma_planes(uint32_t p) {   // smallest P such that every residue in [0, p) has P signed base-256 digits in [-128, 127]:
  uint64_t top = 0;             // the largest such number is 127 * (256^P - 1) / 255 (32639, 8355711, 2139127679): a prime in
  for (int planes = 1; planes <= 4; planes++) {   // (8355711, 2^23) needs 4 planes although it has 23 bits
    top = top * 256 + 127;
    if ((uint64_t)p - 1 <= top) return planes;
  }
  return 5;                     // p > 2139127680: not representable in 4 planes (lab_mat8_pack refuses)
}

// column tiles for ns sets whose residues need `planes_v` digit planes
static size_t nt_for(int ns, int planes_v) {
  const size_t nt = ((size_t)ns * planes_v + 31) / 32;
  if (planes_v == 3) return nt > 3 ? 6 : nt;                // 3-plane limbs: 1, 2, 3, 6 column tiles
  return nt == 3 ? 4 : (nt > 4 ? 8 : nt);                   // the GEMM is instantiated for 1, 2, 4, 8 column tiles
}

static size_t v8_job_bytes(const lab_mat8* m, int ns) {
  size_t v8 = 0;
  for (int l = 0; l < m->nlimbs; l++) v8 = std::max(v8, (size_t)D * m->kt_count * nt_for(ns, m->planes[l]) * 1024);
  return Carver::padded(v8);
}

// B-operand tiles of every job, plus (several jobs: many blocks) the slot-major staging array of the results
static size_t stage_bytes(const lab_mat8* m, int ns, size_t njobs) {
  return njobs > 1 ? Carver::padded(njobs * (size_t)D * m->rt_count * 32 * ns * 4) : 0;
}
size_t mfma_stage_bytes(const lab_mat8* m, int ns, size_t njobs) { return Carver::padded(njobs * (size_t)D * m->rt_count * 32 * ns * 4); }
size_t mfma_scratch_bytes(const lab_mat8* m, int ns, size_t njobs) { return v8_job_bytes(m, ns) * njobs + stage_bytes(m, ns, njobs); }
double mfma_gemm_bytes(const lab_mat8* m, int ns, int ns_valid) {
  const double kt_real = (double)((m->nmat * m->cols + 31) / 32);
  double b = 0.0;
  for (int l = 0; l < m->nlimbs; l++)
    b += (double)D * m->planes[l] * ((m->rows + 31) / 32) * kt_real * 1024 + (double)D * kt_real * nt_for(ns, m->planes[l]) * 1024 +
         (double)ns_valid * m->rows * D * 4;
  return b;
}

// recombination weights 256^e mod p and the bias that entering every int32 D as D + 2^31 adds to the sum
static void recomb_consts(uint32_t p, int pl, uint32_t (&pw)[7], uint32_t* bias) {
  pw[0] = 1 % p;
  for (int e = 1; e < 7; e++) pw[e] = (uint32_t)(((uint64_t)pw[e - 1] * 256) % p);
  uint64_t sum = 0;
  for (int i = 0; i < pl; i++) sum += pw[i];
  sum %= p;
  *bias = (uint32_t)((sum * sum % p) * ((1ull << 31) % p) % p);
}

// ---- X_a = P_a^T S per proof on the matrix cores (steps 6-8: phi'^T S and W^T S), a < nops <= 2.
// P_a and S are [n][r] matrices of transform-domain residues in the basis of m (every limb 3 planes), element (x, i) at
// base + job*v_job_stride + (x*r + i)*64; out_a[i*r + k] = sum_x P_a[x][i] S[x][k] at outs[a].base + job*out_job_stride.
// Same tiles, same pairing of two 32-column proofs into one 64-set tile as the Ajtai product of matvec_mfma_impl.
static bool gram_mfma_pairs(size_t njobs, int ns) { return njobs >= 2 && njobs % 2 == 0 && ns == 32; }
// bytes of the witness tiles of every limb (matvec_mfma_impl's `keep_tiles`) and of gram_mfma_impl's scratch (two operands)
size_t gram_mfma_tile_bytes(const lab_mat8* m, size_t njobs) {
  const bool paired = gram_mfma_pairs(njobs, 32);
  return (size_t)m->nlimbs * v8_job_bytes(m, paired ? 64 : 32) * (paired ? njobs / 2 : njobs);
}
size_t gram_mfma_scratch_bytes(const lab_mat8* m, size_t njobs) { return 2 * gram_mfma_tile_bytes(m, njobs); }
bool gram_mfma_ok(const lab_mat8* m, size_t n, size_t r) {
  if (!m || m->nmat * m->cols != n || r <= 16 || r > 32) return false;
  for (int l = 0; l < m->nlimbs; l++)
    if (m->planes[l] != 3) return false;
  return true;
}
// Pcoefs[a].p[job]: the [n][r][64] coefficient matrix (64-bit words below q) of proof `job`; s_tiles: the witness tiles of
// every limb as matvec_mfma_impl(keep_tiles) left them; out_a[i*r + k] = sum_x P_a[x][i] S[x][k] (transform domain, basis of
// m) at outs[a].base + job*out_job_stride.  One pass per operand makes its tiles of all limbs (reduce + NTT + packing fused,
// k_digits_to_v8<true>), one launch per limb multiplies both operands with the witness tiles.
int gram_mfma_impl(lab_ctx* c, const lab_mat8* m, int nops, const HatView* outs, const JobPtrs* Pcoefs, const int8_t* s_tiles,
                   size_t out_job_stride, size_t n, size_t r, size_t njobs, void* scratch) {
  if (!gram_mfma_ok(m, n, r) || nops < 1 || nops > 2) return fail(c, LAB_ERR_ARG, "gram_mfma_impl: shape not taken");
  const uint32_t* primes = m->basis ? c->dg_primes : c->primes;
  const LimbTable* tabs = m->basis ? c->dg_tabs : c->d_tabs;
  const size_t src_jobs = njobs;
  const bool paired = gram_mfma_pairs(njobs, 32);
  const int ns = paired ? 64 : 32;
  if (paired) njobs /= 2;
  const size_t v8_stride = v8_job_bytes(m, ns), limb_stride = v8_stride * njobs;
  size_t nts[MAX_LIMBS];
  for (int l = 0; l < m->nlimbs; l++) nts[l] = nt_for(ns, 3);
  int8_t* p8[2] = {static_cast<int8_t*>(scratch), static_cast<int8_t*>(scratch) + (size_t)m->nlimbs * limb_stride};
  for (int a = 0; a < nops; a++) {
    // k tiles past ceil(n / 64) * 2 are not written (zero padding of kt_count): the multiplication stops after ceil(n / 32)
    if (((n + 63) / 64) * 2 < (n + 31) / 32) return fail(c, LAB_ERR_ARG, "gram_mfma_impl: internal: tile count");
    LAB_HIP(c, launch_coefs_to_v8(Pcoefs[a], n, r, p8[a], limb_stride, v8_stride, m->kt_count, ns, njobs, m->nlimbs, tabs, m->planes, nts,
                                  c->stream));
  }
  for (int l = 0; l < m->nlimbs; l++) {
    uint32_t pw[7], bias;
    recomb_consts(primes[l], 3, pw, &bias);
    LAB_HIP(c, launch_gram_v8(p8[0] + (size_t)l * limb_stride, nops > 1 ? p8[1] + (size_t)l * limb_stride : nullptr,
                              s_tiles + (size_t)l * limb_stride, v8_stride, m->kt_count, (n + 31) / 32, nts[l], paired ? 2 : 1, r, src_jobs,
                              outs[0].base + (size_t)l * outs[0].plane_stride,
                              nops > 1 ? outs[1].base + (size_t)l * outs[1].plane_stride : nullptr, out_job_stride, primes[l], pw, bias,
                              tabs + l, c->stream));
  }
  return LAB_OK;
}

// For every job b < njobs: out_s[rho] (s < ns_valid; hat elements at out.base + b*out_job_stride + s*out_set_stride)
//   = sum_k M[rho][k] V_s[k],  V_s[k] at V.base + b*v_job_stride + s*v_set_stride + k*v_k_stride
// (v_k_stride = 64 words for vectors stored set by set).  ns in {8,16,32,64} is the padded set count the GEMM runs with;
// `scratch` holds mfma_scratch_bytes(m, ns, njobs).  One pack + one GEMM launch per limb; the jobs share the matrix
// stream through L2.  With `timed` (the outer commitment) and the measurement hook on, the GEMM launches of the 4-plane
// limbs are bracketed by events.
int matvec_mfma_impl(lab_ctx* c, HatView out, size_t out_set_stride, const lab_mat8* m, ConstHatView V, size_t v_set_stride, int ns,
                     int ns_valid, void* scratch, size_t v_k_stride, size_t njobs, size_t out_job_stride, size_t v_job_stride,
                     bool timed, MfmaCoefOut* coef, const int8_t* prepacked, MfmaGramOut* gram, int8_t* keep_tiles, size_t keep_limb_stride,
                     const MfmaJobBuffers* jb) {
  const uint32_t* primes = m->basis ? c->dg_primes : c->primes;
  const size_t ktotal = m->nmat * m->cols;
  // Many 32-set jobs sharing the matrix (the Ajtai product of a batch): two source jobs are multiplied as ONE 64-set job,
  // so every matrix tile fetched from L2 feeds 8 column tiles instead of 4 (the launch is bound by that operand traffic).
  const size_t nsub = ns;                                    // sets per source job
  size_t sub_stride_v = 0;
  bool paired = false;
  if (njobs >= 2 && njobs % 2 == 0 && ns == 32) {
    bool same = m->planes[0] >= 3;                           // every limb with 4 (or every limb with 3) planes
    for (int l = 1; l < m->nlimbs; l++) same = same && m->planes[l] == m->planes[0];
    if (same) {
      paired = true;
      sub_stride_v = v_job_stride;
      njobs /= 2; ns = 64; v_job_stride *= 2;
    }
  }
  bool gram_ok = true;
  int8_t* v8 = static_cast<int8_t*>(scratch);
  const size_t v8_stride = v8_job_bytes(m, ns);
  uint32_t* stage = (njobs > 1 || paired) ? reinterpret_cast<uint32_t*>(v8 + v8_stride * njobs) : nullptr;
  if (jb) {   // every job's tiles in a buffer of its own ([limb] each), the staging array elsewhere
    if (!prepacked || paired || njobs < 2) return fail(c, LAB_ERR_ARG, "matvec_mfma_impl: per-job tile buffers need prepacked tiles of several unpaired jobs");
    stage = jb->stage;
  }
  // results wanted as coefficients only, with room for every limb's staging array: keep them all (see MfmaCoefOut)
  const bool keep_stages = stage && !jb && coef && coef->stage_planes && m->nlimbs >= 2 && m->nlimbs <= 3 &&
                           coef->stage_ps >= njobs * D * (m->rt_count * 32) * (size_t)ns;
  for (int l = 0; l < m->nlimbs; l++) {
    const int pl = m->planes[l];
    const size_t nt = nt_for(ns, pl);
    if (keep_stages) stage = coef->stage_planes + (size_t)l * coef->stage_ps;
    // the caller wants the tiles of every limb afterwards (keep_limb_stride: this call packs a slice of the jobs of a larger array)
    if (keep_tiles) v8 = keep_tiles + (size_t)l * (keep_limb_stride ? keep_limb_stride : v8_stride * njobs);
    if (prepacked) v8 = const_cast<int8_t*>(prepacked) + (size_t)l * v8_stride * (jb ? 1 : njobs);   // digits_to_v8_impl made every limb's tiles
    else
      LAB_HIP(c, launch_pack_vectors_i8(ns, ns_valid, v8, V.base + (size_t)l * V.plane_stride, v_set_stride, v_k_stride, ktotal,
                                        m->kt_count, nt, pl, njobs, v_job_stride, v8_stride, nsub, sub_stride_v, c->stream));
    Gemm8Args g;
    g.out = out.base + (size_t)l * out.plane_stride; g.out_set_stride = out_set_stride; g.out_job_stride = out_job_stride;
    g.M8 = m->d[l]; g.V8 = v8; g.v8_job_stride = jb ? jb->job_stride : v8_stride;
    g.planes = (unsigned)pl; g.rt_count = (unsigned)m->rt_count; g.kt_count = (unsigned)m->kt_count;
    g.kt_real = (unsigned)((ktotal + 31) / 32);
    const int waves = gemm_i8_waves(pl, m->rt_count, nt);
    const int tpb = (pl == 3 && waves == 4) ? 4 : waves / pl;       // row tiles per block
    g.rgroups = (unsigned)((m->rt_count + tpb - 1) / tpb); g.njobs = (unsigned)njobs; g.rows = (unsigned)m->rows;
    g.ns = (unsigned)ns; g.ns_valid = (unsigned)ns_valid; g.nsub = (unsigned)nsub; g.p = primes[l];
    g.tab = (m->basis ? c->dg_tabs : c->d_tabs) + l;
    recomb_consts(primes[l], pl, g.pw, &g.bias);
    g.stage = stage;
    if ((timed && c->gemm_diag >= 1 && c->gemm_diag <= 3) || (njobs > 1 && c->gemm_diag == 4)) {
      g.diag = c->gemm_diag; g.diag_buf = c->gemm_diag_buf; g.diag_words = c->gemm_diag_words;
    }
    const bool prof = timed && c->prof_on && pl >= 3;   // `timed`: the outer commitment (the measurement hook's kernel)
    if (prof && c->prof_used == c->prof_events.size()) {
      hipEvent_t a, b;
      LAB_HIP(c, hipEventCreate(&a));
      LAB_HIP(c, hipEventCreate(&b));
      c->prof_events.emplace_back(a, b);
    }
    if (prof) LAB_HIP(c, hipEventRecord(c->prof_events[c->prof_used].first, c->stream));
    LAB_HIP(c, launch_gemm_i8(g, nt, waves, c->stream));
    if (prof) {
      LAB_HIP(c, hipEventRecord(c->prof_events[c->prof_used++].second, c->stream));
      char kname[48];
      const bool persistent = pl == 3 && waves == 4 && nt == 6 && njobs >= 4;      // (launch_gemm_i8's choice)
      const bool small = (uint64_t)g.kt_real * 32 * 3 * 16384 < (1ull << 31) && g.p > (1u << 22);
      if (persistent) snprintf(kname, sizeof kname, "k_gemm_i8_w3p<6, %s, false>", small ? "true" : "false");
      else if (pl == 3 && waves == 4) snprintf(kname, sizeof kname, "k_gemm_i8_w3<%d, %d>", (int)nt, nt == 6 && njobs == 1 ? g.diag : 0);
      else snprintf(kname, sizeof kname, "k_gemm_i8<%d, %d>", (int)nt, waves);
      c->prof_kernel = kname;
      // the matrix digit planes once (the other jobs of a launch find them in L2), every job's B-operand tiles (once per 32-row
      // block, from L2) and residues
      c->prof_bytes += (double)D * pl * m->rt_count * m->kt_count * 1024 +
                       (double)njobs * ((double)D * m->kt_count * nt * 1024 + (double)ns_valid * m->rows * D * 4);
    }
    if (gram && pl == 3 && !prepacked && ns >= 32 && gram->r <= 32 && (size_t)ns_valid == gram->r)   // one tile per plane
      LAB_HIP(c, launch_gram_v8(nullptr, nullptr, v8, v8_stride, m->kt_count, (ktotal + 31) / 32, nt, paired ? 2 : 1, gram->r, njobs * (paired ? 2 : 1),
                                gram->out + (size_t)l * gram->plane_stride, nullptr, gram->job_stride, primes[l], g.pw, g.bias,
                                (m->basis ? c->dg_tabs : c->d_tabs) + l, c->stream));
    else if (gram) gram_ok = false;
    if (keep_stages) continue;
    if (stage && coef)
      LAB_HIP(c, launch_intt_from_stage(stage, m->rows, m->rt_count * 32, ns, ns_valid, nsub, njobs, coef->dense + (size_t)l * coef->plane_stride,
                                        coef->job_stride, coef->set_stride, (m->basis ? c->dg_tabs : c->d_tabs) + l, c->stream));
    else if (stage)
      LAB_HIP(c, launch_slot_transpose(g.out, out_set_stride, out_job_stride, stage, m->rows, m->rt_count * 32, ns, ns_valid, njobs,
                                       nsub, c->stream));
  }
  if (coef) coef->done = stage != nullptr;
  if (keep_stages) { coef->staged = true; coef->st_rows_pad = m->rt_count * 32; coef->st_ns = (size_t)ns; coef->st_nsub = nsub; coef->st_jobs = njobs; }
  if (gram) gram->done = gram_ok;
  return LAB_OK;
}

// B-operand tiles of all limbs of `m` from 16-bit decomposition digits (one job of ns sets: set s at src16.p[s], [K][64]
// words, K = m's sum length): dst holds mfma_prepacked_bytes(m, ns).  Digits must be below every limb prime of m's basis.
size_t mfma_prepacked_bytes(const lab_mat8* m, int ns) { return v8_job_bytes(m, ns) * m->nlimbs; }
int digits_to_v8_impl(lab_ctx* c, const lab_mat8* m, const JobPtrs& src16, int ns, int ns_valid, int8_t* dst, const DigitsFromT* dt) {
  size_t nt[MAX_LIMBS] = {0, 0, 0, 0};
  for (int l = 0; l < m->nlimbs; l++) nt[l] = nt_for(ns, m->planes[l]);
  const size_t ktotal = m->nmat * m->cols;
  if (m->kt_count * 32 > (ktotal + D - 1) / D * D)   // K tiles past the last 64 digits polynomials are padding: zero once
    LAB_HIP(c, hipMemsetAsync(dst, 0, mfma_prepacked_bytes(m, ns), c->stream));
  LAB_HIP(c, launch_digits_to_v8(src16, ktotal, dst, v8_job_bytes(m, ns), m->kt_count, ns, ns_valid, m->nlimbs,
                                 m->basis ? c->dg_tabs : c->d_tabs, m->planes, nt, c->stream, dt));
  return LAB_OK;
}
uint32_t basis_min_prime(const lab_ctx* c, int basis) {
  const uint32_t* primes = basis ? c->dg_primes : c->primes;
  uint32_t mn = primes[0];
  for (int l = 1; l < basis_nlimbs(c, basis); l++) mn = primes[l] < mn ? primes[l] : mn;
  return mn;
}
uint32_t mat8_min_prime(const lab_ctx* c, const lab_mat8* m) {
  const uint32_t* primes = m->basis ? c->dg_primes : c->primes;
  uint32_t mn = primes[0];
  for (int l = 1; l < m->nlimbs; l++) mn = primes[l] < mn ? primes[l] : mn;
  return mn;
}

}  // namespace lab

extern "C" int lab_mat8_pack(lab_ctx* c, const lab_vec* M, size_t m_off, size_t nmat, size_t rows, size_t cols, lab_mat8** out) {
  if (!c || !out) return LAB_ERR_ARG;
  int rc;
  if (nmat == 0 || rows == 0 || cols == 0) return fail(c, LAB_ERR_ARG, "lab_mat8_pack: empty dimension");
  if ((rc = check_range(c, M, m_off, nmat * rows * cols, "lab_mat8_pack(M)"))) return rc;
  if (nmat * cols >= (1u << 17) - 1) return fail(c, LAB_ERR_RANGE, "lab_mat8_pack: %zu products per sum exceed the int32 exactness bound 131070 of the int8 formulation", nmat * cols);
  const uint64_t limit = M->basis ? c->dg_terms : c->max_len;
  if (nmat * cols > limit) return fail(c, LAB_ERR_RANGE, "lab_mat8_pack: reduction length %zu exceeds the exactness bound %llu of this basis", nmat * cols, (unsigned long long)limit);
  LAB_HIP(c, hipSetDevice(c->device));
  lab_mat8* m = new lab_mat8();
  m->ctx = c; m->basis = M->basis; m->nlimbs = M->nlimbs();
  m->nmat = nmat; m->rows = rows; m->cols = cols;
  m->rt_count = (rows + 31) / 32;
  m->kt_count = ((nmat * cols + 31) / 32 + 11) / 12 * 12;   // the GEMM walks K in stages of 4 k-tiles, 3 stages per loop trip
  const uint32_t* primes = M->basis ? c->dg_primes : c->primes;
  // the recombination epilogue of the GEMM is the lazy arithmetic of limb primes below 2^30 (acc_fold_lazy, redc64): the main basis of
  // a strict-32 context (its single prime in [2^30, 2^32), LimbTable.wide) is refused whether or not the prime fits four planes
  if (M->basis == 0 && c->wide) {
    delete m;
    return fail(c, LAB_ERR_RANGE, "lab_mat8_pack: the main basis of a single-limb context with q >= 2^30 has no matrix-core form (use the digit basis)");
  }
  for (int l = 0; l < m->nlimbs; l++)
    if (mfma_planes(primes[l]) > 4 || primes[l] >= (1u << 30)) {
      delete m;
      return fail(c, LAB_ERR_RANGE, "lab_mat8_pack: limb prime %u does not fit four signed byte planes below 2^30", primes[l]);
    }
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

namespace lab {
// an empty packed matrix of up to nmat_max x rows x cols elements whose contents mat8_repack() replaces (the slab of the
// streamed outer commitment, crs_stream.hip)
int mat8_alloc(lab_ctx* c, int basis, size_t nmat_max, size_t rows, size_t cols, lab_mat8** out) {
  if (nmat_max * cols >= (1u << 17) - 1) return fail(c, LAB_ERR_RANGE, "mat8_alloc: %zu products per sum exceed the int32 exactness bound", nmat_max * cols);
  if (basis == 0 && c->wide) return fail(c, LAB_ERR_RANGE, "mat8_alloc: the main basis of a single-limb context with q >= 2^30 has no matrix-core form");
  lab_mat8* m = new lab_mat8();
  m->ctx = c; m->basis = basis; m->nlimbs = basis_nlimbs(c, basis);
  m->nmat = nmat_max; m->rows = rows; m->rows_alloc = rows; m->cols = cols;
  m->rt_count = (rows + 31) / 32;
  m->kt_count = ((nmat_max * cols + 31) / 32 + 11) / 12 * 12;
  const uint32_t* primes = basis ? c->dg_primes : c->primes;
  for (int l = 0; l < m->nlimbs; l++) {
    m->planes[l] = mfma_planes(primes[l]);
    const size_t bytes = (size_t)D * m->planes[l] * m->rt_count * m->kt_count * 1024;
    if (m->planes[l] > 4 || hipMalloc(reinterpret_cast<void**>(&m->d[l]), bytes) != hipSuccess) {
      lab_mat8_destroy(m);
      return fail(c, LAB_ERR_HIP, "mat8_alloc: cannot allocate %zu bytes", bytes);
    }
    m->bytes += bytes;
  }
  *out = m;
  return LAB_OK;
}
// repack from transform-domain planes [limb][nmat][rows][cols][64] (plane stride in words); nmat <= the allocated count
// (source layout [nmat][rows_alloc][cols]); only the first rows_valid rows of every matrix are meaningful: the GEMM stores no others
int mat8_repack(lab_ctx* c, lab_mat8* m, const uint32_t* hat_planes, size_t plane_stride, size_t nmat, size_t rows_valid) {
  if (nmat == 0 || ((nmat * m->cols + 31) / 32) > m->kt_count || rows_valid == 0 || rows_valid > m->rows_alloc)
    return fail(c, LAB_ERR_ARG, "mat8_repack: shape exceeds the allocation");
  m->nmat = nmat;
  m->rows = rows_valid;
  for (int l = 0; l < m->nlimbs; l++)
    LAB_HIP(c, launch_pack_matrix_i8(m->d[l], hat_planes + (size_t)l * plane_stride, false, nmat, m->rows_alloc, m->cols, m->rt_count,
                                     m->kt_count, m->planes[l], c->stream));
  return LAB_OK;
}
// The same contents written straight from coefficient rows (k_coefs_to_a8: reduce + transform + byte planes in one pass):
// mat8_fill_begin declares the shape of the next contents (clearing the tiles when it shrinks, so that padding reads as
// zero), mat8_fill_rows writes rows [row0, row0 + nrows) of matrix m_idx from `coefs` ([nrows][cols][64] 64-bit words).
bool mat8_can_fill_rows(const lab_mat8* m) {
  if (m->cols % 64) return false;
  for (int l = 0; l < m->nlimbs; l++)
    if (m->planes[l] != 3) return false;
  return true;
}
int mat8_fill_begin(lab_ctx* c, lab_mat8* m, size_t nmat, size_t rows_valid) {
  if (nmat == 0 || ((nmat * m->cols + 31) / 32) > m->kt_count || rows_valid == 0 || rows_valid > m->rows_alloc)
    return fail(c, LAB_ERR_ARG, "mat8_fill_begin: shape exceeds the allocation");
  if (nmat < m->nmat || rows_valid < m->rows || !m->filled)
    for (int l = 0; l < m->nlimbs; l++)
      LAB_HIP(c, hipMemsetAsync(m->d[l], 0, (size_t)D * m->planes[l] * m->rt_count * m->kt_count * 1024, c->stream));
  m->nmat = nmat;
  m->rows = rows_valid;
  m->filled = true;
  return LAB_OK;
}
int mat8_fill_rows(lab_ctx* c, lab_mat8* m, const uint64_t* coefs, size_t m_idx, size_t row0, size_t nrows) {
  if (m_idx >= m->nmat || row0 + nrows > m->rows_alloc) return fail(c, LAB_ERR_ARG, "mat8_fill_rows: outside the declared shape");
  const LimbTable* tabs = m->basis ? c->dg_tabs : c->d_tabs;
  LAB_HIP(c, launch_coefs_to_a8(coefs, nrows, m->cols, row0, m_idx * m->cols, m->d, m->rt_count, m->kt_count, m->nlimbs, tabs, c->stream));
  return LAB_OK;
}
}  // namespace lab

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
  if (ns != 8 && ns != 16 && ns != 32 && ns != 64) return fail(c, LAB_ERR_ARG, "lab_matvec_sum_mfma: ns must be 8, 16, 32 or 64");
  int rc;
  if ((rc = check_range(c, out, out_off, (size_t)ns * m->rows, "lab_matvec_sum_mfma(out)"))) return rc;
  if ((rc = check_range(c, V, v_off, (size_t)ns * m->nmat * m->cols, "lab_matvec_sum_mfma(V)"))) return rc;
  if (V->basis != m->basis || out->basis != m->basis || V->compact || out->compact)
    return fail(c, LAB_ERR_ARG, "lab_matvec_sum_mfma: V and out must be plain vectors in the packed matrix's basis");
  LAB_HIP(c, hipSetDevice(c->device));
  if ((rc = ensure_scratch(c, mfma_scratch_bytes(m, ns, 1)))) return rc;
  return matvec_mfma_impl(c, out->view(out_off), m->rows * D, m, V->view(v_off), m->nmat * m->cols * D, ns, ns, c->scratch, D, 1, 0, 0, true);
}
