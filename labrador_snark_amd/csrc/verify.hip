// verify.hip — lab_verify_core: Verifier::verify (src/verification.rs:25-438) as one
// sequence of gfx950 kernel launches, built from the same kernels as the prover
// (checks 15, 19, 20 are the prover's steps 1, 3+4, 9 on transcript data).
// All checks are evaluated; the first failing protocol line in the reference's
// order is reported.  SURVEY §8(f) row f3.
#include <math.h>
#include <string.h>

#include <vector>

#include "context.h"

using namespace lab;
typedef unsigned __int128 u128;

namespace lab {
int axpy_commit_impl(lab_ctx* c, HatView out, ConstHatView G, ConstHatView C, size_t nterms, size_t len, HatView partial);
int inner_product_impl(lab_ctx* c, HatView out, ConstHatView v1, ConstHatView v2, size_t len, HatView partial);
size_t inner_product_splits(size_t len);
}

extern "C" int lab_verify_core(lab_ctx* c, const lab_prove_input* in, const lab_transcript* pf, const uint64_t* b_poly,
                               int64_t bconst, double beta_prime, int* failed_line, double* line14_sum) {
  if (!c) return LAB_ERR_ARG;
  if (!in || !pf || !b_poly || !failed_line) return fail(c, LAB_ERR_ARG, "lab_verify_core: null argument");
  if (in->struct_size != sizeof(lab_prove_input))
    return fail(c, LAB_ERR_ARG, "lab_verify_core: lab_prove_input.struct_size is %zu, this library's layout has %zu bytes (stale binding?)",
                in->struct_size, sizeof(lab_prove_input));
  const size_t n = in->n, r = in->r, kappa = in->kappa, k1 = in->kappa1, k2 = in->kappa2;
  const int t1 = in->t1, t2 = in->t2;
  if (n == 0 || r == 0 || kappa == 0 || k1 == 0) return fail(c, LAB_ERR_ARG, "lab_verify_core: empty dimension");
  if (k1 != k2) return fail(c, LAB_ERR_LEN_MISMATCH, "lab_verify_core: kappa1 != kappa2");
  if (t1 < 1 || t2 < 1 || t1 > 64 || t2 > 64 || in->b1 < 2 || in->b2 < 2 || bconst < 2)
    return fail(c, LAB_ERR_DECOMPOSE, "lab_verify_core: bad decomposition parameters");
  if (!in->phi || !in->a || (!in->Pi && !in->Pi2) || !in->omega || !in->alpha || !in->beta || !in->c)
    return fail(c, LAB_ERR_ARG, "lab_verify_core: null input array");
  if (!pf->u_1 || !pf->b_prime_prime || !pf->u_2 || !pf->z || !pf->t_i_all || !pf->g_mat || !pf->h_mat)
    return fail(c, LAB_ERR_ARG, "lab_verify_core: null transcript field");
  if (r > 4096 || n > (1u << 22) || kappa > (1u << 24) || k1 > (1u << 24)) return fail(c, LAB_ERR_ARG, "lab_verify_core: dimension too large");
  const size_t pairs = r * (r + 1) / 2, nr = n * r, rk = r * kappa, rr = r * r;
  int rc;
  if ((rc = check_range(c, in->A, in->a_off, kappa * n, "lab_verify_core(A)"))) return rc;
  if ((rc = check_range(c, in->B, in->b_off, r * (size_t)t1 * k1 * kappa, "lab_verify_core(B)"))) return rc;
  if ((rc = check_range(c, in->C, in->c_off, pairs * (size_t)t2 * k2, "lab_verify_core(C)"))) return rc;
  if ((rc = check_range(c, in->Dm, in->d_off, pairs * (size_t)t1 * k2, "lab_verify_core(D)"))) return rc;
  const int bdig = in->B->basis;
  if (in->C->basis != bdig || in->Dm->basis != bdig) return fail(c, LAB_ERR_ARG, "lab_verify_core: B, C, D must share one RNS basis");
  if (in->A->basis != 0 || in->A->compact) return fail(c, LAB_ERR_ARG, "lab_verify_core: A must be a main-basis vector");
  if (bdig) {
    const uint64_t bm = (uint64_t)(in->b1 > in->b2 ? in->b1 : in->b2);
    if ((u128)r * t1 * kappa + (u128)pairs * t2 > c->dg_terms || (u128)pairs * t1 > c->dg_terms || bm - 1 > c->dg_vmax)
      return fail(c, LAB_ERR_RANGE, "lab_verify_core: outer commitments exceed the digit basis bounds");
  }
  if (!c->direct) {
    u128 M = 1;
    for (int l = 0; l < c->nlimbs; l++) M *= c->primes[l];
    const u128 half = (M - 1) / 2, per = (u128)D * (u128)(c->q - 1);
    const u128 bmax = (u128)(in->b1 > in->b2 ? in->b1 : in->b2);
    size_t longest = n > nr + rr ? n : nr + rr;
    if (kappa > longest) longest = kappa;
    if ((u128)longest > half / (per * (u128)(c->q - 1)) ||
        (!bdig && (u128)r * t1 * kappa + (u128)pairs * t2 > half / (per * bmax)))
      return fail(c, LAB_ERR_RANGE, "lab_verify_core: a reduction exceeds the CRT range of this context");
  }
  LAB_HIP(c, hipSetDevice(c->device));
  const int L = c->nlimbs;
  hipStream_t s = c->stream;
  const LimbTable* tabs = c->d_tabs;
  const uint64_t q = c->q;

  // ---------------- workspace
  size_t hat_total = 0;
  auto HT = [&](size_t cnt) { size_t o = hat_total; hat_total += cnt; return o; };
  const size_t hphi = HT(nr), hab = HT(2), hc = HT(r), hz = HT(n), ht = HT(rk), hG = HT(rr), hH = HT(rr), happ = HT(rr),
               ha = HT(rr), haa = HT(rr), htmp = HT(rr), hphipp = HT(nr), hphif = HT(nr), hbphi = HT(nr), hbb = HT(2),
               hb = HT(2), h15l = HT(kappa), h15r = HT(kappa), hv = HT(r), hone = HT(4);
  size_t hatd_total = 0;
  auto HDT = [&](size_t cnt) { size_t o = hatd_total; hatd_total += cnt; return o; };
  const size_t htdec = HDT(rk * t1), hGd = HDT(pairs * t2), hHd = HDT(pairs * t1), hlhs = HDT(k1), hrhs = HDT(k1), hu2 = HDT(k2);
  size_t partd_elems = r * (size_t)t1 * k1;
  if (axpy_num_chunks(pairs * t2) * k2 > partd_elems) partd_elems = axpy_num_chunks(pairs * t2) * k2;
  if (axpy_num_chunks(pairs * t1) * k2 > partd_elems) partd_elems = axpy_num_chunks(pairs * t1) * k2;
  const size_t hpartd = HDT(partd_elems);
  const size_t hatd_words = hatd_total * D;
  const int Ld = basis_nlimbs(c, bdig);
  const LimbTable* tabsd = basis_tabs(c, bdig);
  size_t part_elems = 512;
  if (axpy_num_chunks(n) * r > part_elems) part_elems = axpy_num_chunks(n) * r;
  if (axpy_num_chunks(r) * (kappa > r ? kappa : r) > part_elems) part_elems = axpy_num_chunks(r) * (kappa > r ? kappa : r);
  const size_t hpart = HT(part_elems);
  const size_t hat_words = hat_total * D;
  size_t tmp_elems = rk;
  if (nr > tmp_elems) tmp_elems = nr;
  if (k1 > tmp_elems) tmp_elems = k1;
  if (rr > tmp_elems) tmp_elems = rr;
  const size_t npoly_norm = 2 * n + rk * t1 + rr * t2 + rr * t1;
  const size_t coef_elems = rr + nr + nr + rr + nr + 1 + 2 * n + rk * t1 + rr * t2 + rr * t1 + pairs + pairs * t2 + pairs +
                            pairs * t1 + 2 * kappa + r + r + 8 + k1 + k2 + 4;
  size_t bytes = Carver::padded((size_t)L * hat_words * 4) + Carver::padded((size_t)Ld * hatd_words * 4) +
                 Carver::padded((size_t)L * tmp_elems * D * 4) +
                 coef_elems * D * 8 + 2 * Carver::padded(npoly_norm * 8) + 64 * 256 + (1 << 16);
  bytes += Staging::need(in->phi, nr * D * 8) + Staging::need(in->a, rr * D * 8) +
           (in->Pi2 ? Staging::need(in->Pi2, r * 256 * n * (D / 4)) + Carver::padded(nr * D * 8) + 4096 : Staging::need(in->Pi, r * 256 * n * D)) +
           Staging::need(in->omega, 256 * 8) + Staging::need(in->alpha, D * 8) + Staging::need(in->beta, D * 8) +
           Staging::need(in->c, r * D * 8) + Staging::need(pf->u_1, k1 * D * 8) + Staging::need(pf->u_2, k2 * D * 8) +
           Staging::need(pf->z, n * D * 8) + Staging::need(pf->t_i_all, rk * D * 8) + Staging::need(pf->g_mat, rr * D * 8) +
           Staging::need(pf->h_mat, rr * D * 8) + Staging::need(pf->b_prime_prime, D * 8) + Staging::need(b_poly, D * 8);
  if ((rc = ensure_scratch(c, bytes))) return rc;
  Carver cv(c->scratch);
  uint32_t* hat_base = cv.take<uint32_t>((size_t)L * hat_words);
  uint32_t* dense_tmp = cv.take<uint32_t>((size_t)L * tmp_elems * D);
  uint32_t* hatd_base = cv.take<uint32_t>((size_t)Ld * hatd_words);
  auto HV = [&](size_t off) { return HatView{hat_base + off * D, hat_words}; };
  auto HD = [&](size_t off) { return HatView{hatd_base + off * D, hatd_words}; };
  auto CO = [&](size_t elems) { return cv.take<uint64_t>(elems * D); };
  uint64_t *capp = CO(rr), *cW = CO(nr), *cphipp = CO(nr), *caa = CO(rr), *cphif = CO(nr), *cb = CO(1), *czdec = CO(2 * n),
           *ctdec = CO(rk * t1), *cGdec = CO(rr * t2), *cHdec = CO(rr * t1), *cGup = CO(pairs), *cGupd = CO(pairs * t2),
           *cHup = CO(pairs), *cHupd = CO(pairs * t1), *c15l = CO(kappa), *c15r = CO(kappa), *cv16 = CO(r), *cw17 = CO(r),
           *csc = CO(8), *cu1c = CO(k1), *cu2c = CO(k2), *cbb = CO(2), *cab = CO(2);
  uint64_t* nlo = cv.take<uint64_t>(npoly_norm);
  uint64_t* nhi = cv.take<uint64_t>(npoly_norm);
  unsigned int* flags = cv.take<unsigned int>(32);
  Staging st(c, &cv);
  const uint64_t *dphi, *da, *domega, *dalpha, *dbeta, *dc, *du1, *du2, *dz, *dt, *dG, *dH, *dbpp, *dbpoly;
  const int8_t* dPi;
  if ((rc = st.in(in->phi, nr * D, &dphi))) return rc;
  if ((rc = st.in(in->a, rr * D, &da))) return rc;
  const uint8_t* dPi2 = nullptr;
  dPi = nullptr;
  if (in->Pi2) { if ((rc = st.in(in->Pi2, r * 256 * n * (D / 4), &dPi2))) return rc; }
  else if ((rc = st.in(in->Pi, r * 256 * n * D, &dPi))) return rc;
  if ((rc = st.in(in->omega, (size_t)256, &domega))) return rc;
  if ((rc = st.in(in->alpha, (size_t)D, &dalpha))) return rc;
  if ((rc = st.in(in->beta, (size_t)D, &dbeta))) return rc;
  if ((rc = st.in(in->c, r * D, &dc))) return rc;
  if ((rc = st.in((const uint64_t*)pf->u_1, k1 * D, &du1))) return rc;
  if ((rc = st.in((const uint64_t*)pf->u_2, k2 * D, &du2))) return rc;
  if ((rc = st.in((const uint64_t*)pf->z, n * D, &dz))) return rc;
  if ((rc = st.in((const uint64_t*)pf->t_i_all, rk * D, &dt))) return rc;
  if ((rc = st.in((const uint64_t*)pf->g_mat, rr * D, &dG))) return rc;
  if ((rc = st.in((const uint64_t*)pf->h_mat, rr * D, &dH))) return rc;
  if ((rc = st.in((const uint64_t*)pf->b_prime_prime, (size_t)D, &dbpp))) return rc;
  if ((rc = st.in(b_poly, (size_t)D, &dbpoly))) return rc;
  LAB_HIP(c, hipMemsetAsync(flags, 0, 32 * sizeof(unsigned int), s));
  enum { F8 = 0, F9, F15, F16, F17, F18, F19, F20 };

  // transforms of state, tape and transcript
  if ((rc = to_hat(c, dphi, HV(hphi), nr))) return rc;
  LAB_HIP(c, hipMemcpyAsync(cab, dalpha, D * 8, hipMemcpyDeviceToDevice, s));
  LAB_HIP(c, hipMemcpyAsync(cab + D, dbeta, D * 8, hipMemcpyDeviceToDevice, s));
  if ((rc = to_hat(c, cab, HV(hab), 2))) return rc;
  if ((rc = to_hat(c, dc, HV(hc), r))) return rc;
  if ((rc = to_hat(c, dz, HV(hz), n))) return rc;
  if ((rc = to_hat(c, dt, HV(ht), rk))) return rc;
  if ((rc = to_hat(c, dG, HV(hG), rr))) return rc;
  if ((rc = to_hat(c, dH, HV(hH), rr))) return rc;
  if ((rc = to_hat(c, da, HV(ha), rr))) return rc;

  // lines 3-4 (src/verification.rs:38-89): a'' = psi a', phi''_i
  const uint64_t psi = in->psi % q;
  LAB_HIP(c, launch_coef_elementwise(capp, da, nullptr, rr, q, psi, 0, s));
  if (dPi2) {   // 2-bit Pi: the prover's fused pass with an all-zero witness (its projection output is not used here)
    uint64_t* zeroS = cv.take<uint64_t>(nr * D);
    uint64_t* dummy_proj = cv.take<uint64_t>(256);
    LAB_HIP(c, hipMemsetAsync(zeroS, 0, nr * D * 8, s));
    LAB_HIP(c, hipMemsetAsync(dummy_proj, 0, 256 * 8, s));
    JobPtrs pj, wj, pij, sj, oj;
    pj.p[0] = dummy_proj; wj.p[0] = cW; pij.p[0] = const_cast<uint8_t*>(dPi2); sj.p[0] = zeroS; oj.p[0] = const_cast<uint64_t*>(domega);
    LAB_HIP(c, launch_pi_fused2_jobs(pj, wj, pij, sj, oj, n, r, q, 1, false, s));
  } else {
    LAB_HIP(c, launch_phi_pi_part(cW, dPi, domega, n, r, q, s));
  }
  LAB_HIP(c, launch_coef_elementwise(cphipp, dphi, cW, nr, q, psi, 1, s));
  if ((rc = to_hat(c, capp, HV(happ), rr))) return rc;
  if ((rc = to_hat(c, cphipp, HV(hphipp), nr))) return rc;
  // line 5 (:96-111): a_ij = alpha a_ij + beta a''_ij, canonicalised
  LAB_HIP(c, launch_poly_by_vec(HV(haa), HV(hab), HV(ha), rr, L, tabs, s));
  LAB_HIP(c, launch_poly_by_vec(HV(htmp), HV(hab + 1), HV(happ), rr, L, tabs, s));
  LAB_HIP(c, launch_pointwise(PW_ADD, HV(haa), HV(haa), HV(htmp), rr, L, tabs, s));
  if ((rc = from_hat(c, HV(haa), rr, dense_tmp, caa))) return rc;
  if ((rc = to_hat(c, caa, HV(haa), rr))) return rc;
  // line 6 (:118-133): phi_i = alpha phi_i + beta phi''_i, canonicalised
  LAB_HIP(c, launch_poly_by_vec(HV(hphif), HV(hab), HV(hphi), nr, L, tabs, s));
  LAB_HIP(c, launch_poly_by_vec(HV(hbphi), HV(hab + 1), HV(hphipp), nr, L, tabs, s));
  LAB_HIP(c, launch_pointwise(PW_ADD, HV(hphif), HV(hphif), HV(hbphi), nr, L, tabs, s));
  if ((rc = from_hat(c, HV(hphif), nr, dense_tmp, cphif))) return rc;
  if ((rc = to_hat(c, cphif, HV(hphif), nr))) return rc;
  // line 7 (:140-148): b = alpha b + beta b''
  LAB_HIP(c, hipMemcpyAsync(cbb, dbpoly, D * 8, hipMemcpyDeviceToDevice, s));
  LAB_HIP(c, hipMemcpyAsync(cbb + D, dbpp, D * 8, hipMemcpyDeviceToDevice, s));
  if ((rc = to_hat(c, cbb, HV(hbb), 2))) return rc;
  LAB_HIP(c, launch_pointwise(PW_MUL, HV(hb), HV(hab), HV(hbb), 2, L, tabs, s));
  LAB_HIP(c, launch_pointwise(PW_ADD, HV(hb), HV(hb), HV(hb + 1), 1, L, tabs, s));
  if ((rc = from_hat(c, HV(hb), 1, dense_tmp, cb))) return rc;

  // checks 8, 9 (:157-178)
  LAB_HIP(c, launch_symmetry_u64(flags + F8, dG, r, s));
  LAB_HIP(c, launch_symmetry_u64(flags + F9, dH, r, s));

  // lines 10-13 (:185-225) decompositions, line 14 (:231-267) norms
  LAB_HIP(c, launch_decompose(czdec, dz, n, q, (uint64_t)bconst, 2, n, s));                 // [2][n]
  LAB_HIP(c, launch_decompose(ctdec, dt, rk, q, (uint64_t)in->b1, t1, kappa, s));           // [i][k][rho]
  LAB_HIP(c, launch_decompose(cGdec, dG, rr, q, (uint64_t)in->b2, t2, 1, s));               // [i][j][k]
  LAB_HIP(c, launch_decompose(cHdec, dH, rr, q, (uint64_t)in->b1, t1, 1, s));               // [i][j][k]
  size_t o_z = 0, o_t = 2 * n, o_g = o_t + rk * t1, o_h = o_g + rr * t2;
  LAB_HIP(c, launch_norm_sq(nlo + o_z, nhi + o_z, czdec, 2 * n, s));
  LAB_HIP(c, launch_norm_sq(nlo + o_t, nhi + o_t, ctdec, rk * t1, s));
  LAB_HIP(c, launch_norm_sq(nlo + o_g, nhi + o_g, cGdec, rr * t2, s));
  LAB_HIP(c, launch_norm_sq(nlo + o_h, nhi + o_h, cHdec, rr * t1, s));

  // check 15 (:274-296): A z == sum_i c_i t_i
  LAB_HIP(c, launch_matvec_partial(HV(h15l), in->A->view(in->a_off), HV(hz), 1, kappa, n, L, tabs, s));
  if ((rc = axpy_commit_impl(c, HV(h15r), HV(hc), HV(ht), r, kappa, HV(hpart)))) return rc;
  if ((rc = from_hat(c, HV(h15l), kappa, dense_tmp, c15l))) return rc;
  if ((rc = from_hat(c, HV(h15r), kappa, dense_tmp, c15r))) return rc;
  LAB_HIP(c, launch_compare_u64(flags + F15, c15l, c15r, kappa * D, s));

  // check 16 (:303-314): <z,z> == sum_ij g_ij c_i c_j  = sum_j c_j (sum_i c_i g_ij)
  if ((rc = inner_product_impl(c, HV(hone), HV(hz), HV(hz), n, HV(hpart)))) return rc;
  if ((rc = axpy_commit_impl(c, HV(hv), HV(hc), HV(hG), r, r, HV(hpart)))) return rc;
  if ((rc = from_hat(c, HV(hv), r, dense_tmp, cv16))) return rc;
  if ((rc = to_hat(c, cv16, HV(hv), r))) return rc;
  if ((rc = inner_product_impl(c, HV(hone + 1), HV(hv), HV(hc), r, HV(hpart)))) return rc;
  if ((rc = from_hat(c, HV(hone), 2, dense_tmp, csc))) return rc;
  LAB_HIP(c, launch_compare_u64(flags + F16, csc, csc + D, D, s));

  // check 17 (:320-334): sum_i <phi_i,z> c_i == sum_ij h_ij c_i c_j
  if ((rc = axpy_commit_impl(c, HV(hv), HV(hz), HV(hphif), n, r, HV(hpart)))) return rc;   // w_i = sum_x z[x] phi[x][i]
  if ((rc = from_hat(c, HV(hv), r, dense_tmp, cw17))) return rc;
  if ((rc = to_hat(c, cw17, HV(hv), r))) return rc;
  if ((rc = inner_product_impl(c, HV(hone), HV(hv), HV(hc), r, HV(hpart)))) return rc;
  if ((rc = axpy_commit_impl(c, HV(hv), HV(hc), HV(hH), r, r, HV(hpart)))) return rc;
  if ((rc = from_hat(c, HV(hv), r, dense_tmp, cv16))) return rc;
  if ((rc = to_hat(c, cv16, HV(hv), r))) return rc;
  if ((rc = inner_product_impl(c, HV(hone + 1), HV(hv), HV(hc), r, HV(hpart)))) return rc;
  if ((rc = from_hat(c, HV(hone), 2, dense_tmp, csc + 2 * D))) return rc;
  LAB_HIP(c, launch_compare_u64(flags + F17, csc + 2 * D, csc + 3 * D, D, s));

  // check 18 (:340-352): sum_ij a_ij g_ij + sum_i h_ii - b == 0
  if ((rc = inner_product_impl(c, HV(hone + 2), HV(haa), HV(hG), rr, HV(hpart)))) return rc;
  if ((rc = from_hat(c, HV(hone + 2), 1, dense_tmp, csc + 4 * D))) return rc;
  LAB_HIP(c, launch_check18(csc + 5 * D, csc + 4 * D, dH, cb, r, q, s));
  LAB_HIP(c, hipMemsetAsync(csc + 6 * D, 0, D * 8, s));
  LAB_HIP(c, launch_compare_u64(flags + F18, csc + 5 * D, csc + 6 * D, D, s));

  // check 19 (:372-415): recompute u_1 from t and g
  if ((rc = to_hat(c, ctdec, HD(htdec), rk * t1, bdig))) return rc;
  if ((rc = matvec_partial_any(c, HD(hpartd), in->B, in->b_off, HD(htdec), r * (size_t)t1, k1, kappa))) return rc;
  LAB_HIP(c, launch_sum_terms(HD(hlhs), HD(hpartd), r * (size_t)t1, k1, Ld, tabsd, s));
  LAB_HIP(c, launch_gather_upper(cGup, dG, r, s));
  LAB_HIP(c, launch_decompose(cGupd, cGup, pairs, q, (uint64_t)in->b2, t2, 1, s));
  if ((rc = to_hat(c, cGupd, HD(hGd), pairs * t2, bdig))) return rc;
  if ((rc = axpy_commit_any(c, HD(hrhs), HD(hGd), in->C, in->c_off, pairs * t2, k2, HD(hpartd)))) return rc;
  LAB_HIP(c, launch_pointwise(PW_ADD, HD(hlhs), HD(hlhs), HD(hrhs), k1, Ld, tabsd, s));
  if ((rc = from_hat(c, HD(hlhs), k1, dense_tmp, cu1c, bdig))) return rc;
  LAB_HIP(c, launch_compare_u64(flags + F19, cu1c, du1, k1 * D, s));

  // check 20 (:421-435): recompute u_2 from h
  LAB_HIP(c, launch_gather_upper(cHup, dH, r, s));
  LAB_HIP(c, launch_decompose(cHupd, cHup, pairs, q, (uint64_t)in->b1, t1, 1, s));
  if ((rc = to_hat(c, cHupd, HD(hHd), pairs * t1, bdig))) return rc;
  if ((rc = axpy_commit_any(c, HD(hu2), HD(hHd), in->Dm, in->d_off, pairs * t1, k2, HD(hpartd)))) return rc;
  if ((rc = from_hat(c, HD(hu2), k2, dense_tmp, cu2c, bdig))) return rc;
  LAB_HIP(c, launch_compare_u64(flags + F20, cu2c, du2, k2 * D, s));

  // ---------------- results
  unsigned int hflags[32];
  std::vector<uint64_t> hlo(npoly_norm), hhi(npoly_norm);
  LAB_HIP(c, hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
  LAB_HIP(c, hipMemcpyAsync(hlo.data(), nlo, npoly_norm * 8, hipMemcpyDeviceToHost, s));
  LAB_HIP(c, hipMemcpyAsync(hhi.data(), nhi, npoly_norm * 8, hipMemcpyDeviceToHost, s));
  LAB_HIP(c, hipStreamSynchronize(s));
  // line 14 in the reference's f64 order (src/verification.rs:231-255; util.rs:188-202)
  auto nf = [&](size_t idx) { return (double)(((u128)hhi[idx] << 64) | hlo[idx]); };
  double sum = 0.0;
  for (int i = 0; i < 2; i++) {
    double v = 0.0;
    for (size_t x = 0; x < n; x++) v += nf(o_z + (size_t)i * n + x);
    sum += v;
  }
  for (size_t i = 0; i < r; i++)
    for (int k = 0; k < t1; k++) {
      double v = 0.0;
      for (size_t rho = 0; rho < kappa; rho++) v += nf(o_t + (i * t1 + k) * kappa + rho);
      sum += v;
    }
  for (size_t e = 0; e < rr; e++)
    for (int k = 0; k < t2; k++) sum += nf(o_g + e * t2 + k);
  for (size_t e = 0; e < rr; e++)
    for (int k = 0; k < t1; k++) sum += nf(o_h + e * t1 + k);
  if (line14_sum) *line14_sum = sum;
  int line = 0;
  if (hflags[F8]) line = 8;
  else if (hflags[F9]) line = 9;
  else if (sum > beta_prime) line = 14;
  else if (hflags[F15]) line = 15;
  else if (hflags[F16]) line = 16;
  else if (hflags[F17]) line = 17;
  else if (hflags[F18]) line = 18;
  else if (hflags[F19]) line = 19;
  else if (hflags[F20]) line = 20;
  *failed_line = line;
  return LAB_OK;
}
