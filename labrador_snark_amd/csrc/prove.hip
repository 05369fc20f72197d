// prove.hip — lab_prove_core: Prover::proof_gen (src/proofgen.rs:30-427) as one
// deterministic sequence of gfx950 kernel launches on the context's stream.
//
// Structure: witness/state are transformed once; every ring product runs in
// the transform domain; a result that feeds another product goes through the
// coefficient domain (exact CRT, mod q) first, so every transform-domain
// reduction starts from canonical residues and stays inside the CRT range.
// Zq-scalar multiplications (psi, omega, 2^-1) and digit decompositions are
// coefficient-domain kernels.  Step numbers follow SURVEY.md §3.2.
#include <math.h>
#include <string.h>

#include "context.h"

using namespace lab;
typedef unsigned __int128 u128;

namespace lab {
int axpy_commit_impl(lab_ctx* c, HatView out, ConstHatView G, ConstHatView C, size_t nterms, size_t len, HatView partial);
int inner_product_impl(lab_ctx* c, HatView out, ConstHatView v1, ConstHatView v2, size_t len, HatView partial);
}

namespace {

struct HatArena {
  size_t total = 0;
  size_t take(size_t count) { size_t o = total; total += count; return o; }
};

uint64_t inv2_mod(uint64_t q) { return (q + 1) / 2; }  // q odd: 2 * (q+1)/2 = q + 1 == 1 (mod q); equals 2^(Q-2) mod Q (src/proofgen.rs:341-346)

// terms * 64 * (q-1) * vmax must stay below M/2 (generic mode); always true in direct mode
bool exact_ok(const lab_ctx* c, u128 terms, u128 vmax) {
  if (c->direct) return true;
  u128 M = 1;
  for (int l = 0; l < c->nlimbs; l++) M *= c->primes[l];
  u128 half = (M - 1) / 2;
  u128 per = (u128)D * (u128)(c->q - 1);
  // avoid overflow: compare terms <= half / (per * vmax)
  if (vmax == 0) return true;
  u128 denom = per * vmax;
  return terms <= half / denom;
}

}  // namespace

extern "C" int lab_prove_core(lab_ctx* c, const lab_prove_input* in, lab_transcript* out) {
  if (!c) return LAB_ERR_ARG;
  if (!in || !out) return fail(c, LAB_ERR_ARG, "lab_prove_core: null argument");
  const size_t n = in->n, r = in->r, kappa = in->kappa, k1 = in->kappa1, k2 = in->kappa2;
  const int t1 = in->t1, t2 = in->t2;
  if (n == 0 || r == 0 || kappa == 0 || k1 == 0) return fail(c, LAB_ERR_ARG, "lab_prove_core: empty dimension");
  if (k1 != k2) return fail(c, LAB_ERR_LEN_MISMATCH, "lab_prove_core: u_1 = lhs + rhs needs kappa1 == kappa2 (src/proofgen.rs:153)");
  if (t1 < 1 || t2 < 1 || t1 > 64 || t2 > 64) return fail(c, LAB_ERR_DECOMPOSE, "lab_prove_core: t1/t2 out of range (t2 = -1 is what RuntimeConstants gives at Q=8191, n=1024)");
  if (in->b1 < 2 || in->b2 < 2) return fail(c, LAB_ERR_DECOMPOSE, "lab_prove_core: decomposition base < 2 never terminates in the reference");
  if (!in->S || !in->phi || !in->a || !in->Pi || !in->omega || !in->alpha || !in->beta || !in->c)
    return fail(c, LAB_ERR_ARG, "lab_prove_core: null input array");
  if (r > 4096 || n > (1u << 22) || kappa > (1u << 24) || k1 > (1u << 24)) return fail(c, LAB_ERR_ARG, "lab_prove_core: dimension too large");
  const size_t pairs = r * (r + 1) / 2;
  const size_t nr = n * r, rk = r * kappa, rr = r * r;
  int rc;
  if ((rc = check_range(c, in->A, in->a_off, kappa * n, "lab_prove_core(A)"))) return rc;
  if ((rc = check_range(c, in->B, in->b_off, r * (size_t)t1 * k1 * kappa, "lab_prove_core(B)"))) return rc;
  if ((rc = check_range(c, in->C, in->c_off, pairs * (size_t)t2 * k2, "lab_prove_core(C)"))) return rc;
  if ((rc = check_range(c, in->Dm, in->d_off, pairs * (size_t)t1 * k2, "lab_prove_core(D)"))) return rc;
  // B, C, D may live in the context's digit basis (lab_ctx_enable_digit_basis): steps 3, 4, 9 then run in it
  const int bdig = in->B->basis;
  if (in->C->basis != bdig || in->Dm->basis != bdig) return fail(c, LAB_ERR_ARG, "lab_prove_core: B, C, D must share one RNS basis");
  if (in->A->basis != 0 || in->A->compact) return fail(c, LAB_ERR_ARG, "lab_prove_core: A must be a main-basis vector");
  // exactness of every transform-domain reduction (see header comment)
  const u128 qm1 = c->q - 1;
  const u128 bmax = (u128)(in->b1 > in->b2 ? in->b1 : in->b2);
  if (!exact_ok(c, n, qm1) || !exact_ok(c, rr + nr, qm1) || !exact_ok(c, r, qm1))
    return fail(c, LAB_ERR_RANGE, "lab_prove_core: a reduction exceeds the CRT range of this context; create it with a larger max_len");
  if (bdig) {
    if ((u128)r * t1 * kappa + (u128)pairs * t2 > c->dg_terms || (u128)pairs * t1 > c->dg_terms || bmax - 1 > c->dg_vmax)
      return fail(c, LAB_ERR_RANGE, "lab_prove_core: outer commitments exceed the digit basis bounds (max_terms %llu, digit_bound %llu)",
                  (unsigned long long)c->dg_terms, (unsigned long long)(c->dg_vmax + 1));
  } else if (!exact_ok(c, (u128)r * t1 * kappa + (u128)pairs * t2, bmax) || !exact_ok(c, (u128)pairs * t1, (u128)in->b1)) {
    return fail(c, LAB_ERR_RANGE, "lab_prove_core: a reduction exceeds the CRT range of this context; create it with a larger max_len");
  }
  if ((u128)r * n * D * qm1 >= ((u128)1 << 63)) return fail(c, LAB_ERR_RANGE, "lab_prove_core: JL projection does not fit 64 bits");
  LAB_HIP(c, hipSetDevice(c->device));

  // ---------------- workspace plan
  const int L = c->nlimbs;
  HatArena ha;
  const size_t hS = ha.take(nr), hphi = ha.take(nr), hT = ha.take(rk), hG = ha.take(rr), hphipp = ha.take(nr),
               happ = ha.take(rr), hGc = ha.take(rr), hbpp = ha.take(2), hphif = ha.take(nr), hbphi = ha.take(nr),
               hX = ha.take(rr), hc = ha.take(r), hz = ha.take(n), hab = ha.take(2);
  // second arena, in the basis of B/C/D, for the digit-operand reductions (steps 3, 4, 9)
  HatArena hd;
  const size_t hTdec = hd.take(rk * t1), hGdec = hd.take(pairs * t2), hlhs = hd.take(k1), hrhs = hd.take(k1),
               hHdec = hd.take(pairs * t1), hu2 = hd.take(k2);
  const int Ld = basis_nlimbs(c, bdig);
  const LimbTable* tabsd = basis_tabs(c, bdig);
  size_t tmp_elems = rk;
  if (nr > tmp_elems) tmp_elems = nr;
  if (k1 > tmp_elems) tmp_elems = k1;
  if (rr > tmp_elems) tmp_elems = rr;
  size_t part_elems = 512;  // inner_product_impl splits a long inner product into <= 512 partials
  if (gram_num_chunks(n) * rr > part_elems) part_elems = gram_num_chunks(n) * rr;
  const size_t hpart = ha.take(part_elems);
  size_t partd_elems = r * (size_t)t1 * k1;
  const size_t ax_chunks_c = axpy_num_chunks(pairs * t2), ax_chunks_d = axpy_num_chunks(pairs * t1);
  if (ax_chunks_c * k2 > partd_elems) partd_elems = ax_chunks_c * k2;
  if (ax_chunks_d * k2 > partd_elems) partd_elems = ax_chunks_d * k2;
  const size_t hpartd = hd.take(partd_elems);
  const size_t hat_words = ha.total * D, hatd_words = hd.total * D;

  const size_t coef_elems = rk /*cT*/ + rk * t1 /*cTdec*/ + rr /*cG*/ + pairs /*cGup*/ + pairs * t2 /*cGdec*/ +
                            nr /*cW*/ + nr /*cphipp*/ + rr /*capp*/ + nr /*cphif*/ + rr /*cX*/ + rr /*cH*/ +
                            pairs /*cHup*/ + pairs * t1 /*cHdec*/ + k1 /*cu1*/ + k2 /*cu2*/ + n /*cz*/ + 1 /*cbpp*/ +
                            2 /*cab*/;
  size_t bytes = Carver::padded((size_t)L * hat_words * 4) + Carver::padded((size_t)Ld * hatd_words * 4) +
                 Carver::padded((size_t)L * tmp_elems * D * 4) +
                 coef_elems * (D * 8) + 32 * 256 /*alignment padding of the carves*/ + 3 * 4096 /*proj*/;
  bytes += Staging::need(in->S, nr * D * 8) + Staging::need(in->phi, nr * D * 8) + Staging::need(in->a, rr * D * 8) +
           Staging::need(in->Pi, r * 256 * n * D) + Staging::need(in->omega, 256 * 8) +
           Staging::need(in->alpha, D * 8) + Staging::need(in->beta, D * 8) + Staging::need(in->c, r * D * 8);
  if ((rc = ensure_scratch(c, bytes + (1 << 16)))) return rc;
  Carver cv(c->scratch);
  uint32_t* hat_base = cv.take<uint32_t>((size_t)L * hat_words);
  uint32_t* hatd_base = cv.take<uint32_t>((size_t)Ld * hatd_words);
  auto HV = [&](size_t off) { return HatView{hat_base + off * D, hat_words}; };
  auto HD = [&](size_t off) { return HatView{hatd_base + off * D, hatd_words}; };
  auto CO = [&](size_t elems) { return cv.take<uint64_t>(elems * D); };
  // results the caller wants in device memory are produced in place (no staging copy)
  auto OUT = [&](uint64_t* user, size_t elems) { return (user && is_device_pointer(user)) ? user : CO(elems); };
  uint64_t *cT = OUT(out->t_i_all, rk), *cTdec = CO(rk * t1), *cG = OUT(out->g_mat, rr), *cGup = CO(pairs),
           *cGdec = CO(pairs * t2), *cW = CO(nr), *cphipp = CO(nr), *capp = CO(rr), *cphif = CO(nr), *cX = CO(rr),
           *cH = OUT(out->h_mat, rr), *cHup = CO(pairs), *cHdec = CO(pairs * t1), *cu1 = OUT(out->u_1, k1),
           *cu2 = OUT(out->u_2, k2), *cz = OUT(out->z, n), *cbpp = CO(1), *cab = CO(2);
  int64_t* dproj_int = cv.take<int64_t>(256);
  uint64_t* dproj = cv.take<uint64_t>(256);
  Staging st(c, &cv);
  const uint64_t *dS, *dphi, *da, *domega, *dalpha, *dbeta, *dc;
  const int8_t* dPi;
  if ((rc = st.in(in->S, nr * D, &dS))) return rc;
  if ((rc = st.in(in->phi, nr * D, &dphi))) return rc;
  if ((rc = st.in(in->a, rr * D, &da))) return rc;
  if ((rc = st.in(in->Pi, r * 256 * n * D, &dPi))) return rc;
  if ((rc = st.in(in->omega, (size_t)256, &domega))) return rc;
  if ((rc = st.in(in->alpha, (size_t)D, &dalpha))) return rc;
  if ((rc = st.in(in->beta, (size_t)D, &dbeta))) return rc;
  if ((rc = st.in(in->c, r * D, &dc))) return rc;
  // from_hat scratch: a dense [L][count][64] buffer (its plane stride is count*64, not the arena's)
  uint32_t* dense_tmp = cv.take<uint32_t>((size_t)L * tmp_elems * D);
  hipStream_t s = c->stream;
  const LimbTable* tabs = c->d_tabs;
  const uint64_t q = c->q;

  // ---------------- transforms of the inputs
  if ((rc = to_hat(c, dS, HV(hS), nr))) return rc;
  if ((rc = to_hat(c, dphi, HV(hphi), nr))) return rc;
  LAB_HIP(c, hipMemcpyAsync(cab, dalpha, D * 8, hipMemcpyDeviceToDevice, s));
  LAB_HIP(c, hipMemcpyAsync(cab + D, dbeta, D * 8, hipMemcpyDeviceToDevice, s));
  if ((rc = to_hat(c, cab, HV(hab), 2))) return rc;
  if ((rc = to_hat(c, dc, HV(hc), r))) return rc;

  // ---------------- step 1 (src/proofgen.rs:41-49): t_i = A s_i
  LAB_HIP(c, launch_matmul_t(HV(hT), in->A->view(in->a_off), HV(hS), kappa, n, r, L, tabs, s));
  if ((rc = from_hat(c, HV(hT), rk, dense_tmp, cT))) return rc;

  // ---------------- step 2 (:59-70): G = S^T S
  if ((rc = gram_impl(c, HV(hG), HV(hS), HV(hS), n, r, HV(hpart)))) return rc;
  if ((rc = from_hat(c, HV(hG), rr, dense_tmp, cG))) return rc;

  // ---------------- step 3 (:101-132): lhs = sum_{i,k} B_ik t_i^(k)
  LAB_HIP(c, launch_decompose(cTdec, cT, rk, q, (uint64_t)in->b1, t1, kappa, s));  // [i][k][rho]
  if ((rc = to_hat(c, cTdec, HD(hTdec), rk * t1, bdig))) return rc;
  if ((rc = matvec_partial_any(c, HD(hpartd), in->B, in->b_off, HD(hTdec), r * (size_t)t1, k1, kappa))) return rc;
  LAB_HIP(c, launch_sum_terms(HD(hlhs), HD(hpartd), r * (size_t)t1, k1, Ld, tabsd, s));

  // ---------------- step 4 (:137-148): rhs = sum_{i<=j,k} g_ij^(k) C_ijk ; u_1 = lhs + rhs (:153)
  LAB_HIP(c, launch_gather_upper(cGup, cG, r, s));
  LAB_HIP(c, launch_decompose(cGdec, cGup, pairs, q, (uint64_t)in->b2, t2, 1, s));  // [pair][k]
  if ((rc = to_hat(c, cGdec, HD(hGdec), pairs * t2, bdig))) return rc;
  if ((rc = axpy_commit_any(c, HD(hrhs), HD(hGdec), in->C, in->c_off, pairs * t2, k2, HD(hpartd)))) return rc;
  LAB_HIP(c, launch_pointwise(PW_ADD, HD(hlhs), HD(hlhs), HD(hrhs), k1, Ld, tabsd, s));
  if ((rc = from_hat(c, HD(hlhs), k1, dense_tmp, cu1, bdig))) return rc;

  // ---------------- step 5 (:161-186, 429-456): JL projection over the integers
  // (one pass over Pi also produces W, the Pi part of phi'' used in step 6)
  LAB_HIP(c, hipMemsetAsync(dproj_int, 0, 256 * 8, s));
  LAB_HIP(c, launch_pi_fused(dproj_int, cW, dPi, dS, domega, n, r, q, s));
  LAB_HIP(c, launch_lift_i64(dproj, dproj_int, 256, q, s));

  // ---------------- step 6 (:205-289): a'' = psi a', phi'' = psi phi' + sum_m omega_m sigma_inv(pi^(m)), b''
  const uint64_t psi = in->psi % q;
  LAB_HIP(c, launch_coef_elementwise(capp, da, nullptr, rr, q, psi, 0, s));
  LAB_HIP(c, launch_coef_elementwise(cphipp, dphi, cW, nr, q, psi, 1, s));
  if ((rc = to_hat(c, capp, HV(happ), rr))) return rc;
  if ((rc = to_hat(c, cphipp, HV(hphipp), nr))) return rc;
  if ((rc = to_hat(c, cG, HV(hGc), rr))) return rc;
  if ((rc = inner_product_impl(c, HV(hbpp), HV(happ), HV(hGc), rr, HV(hpart)))) return rc;
  if ((rc = inner_product_impl(c, HV(hbpp + 1), HV(hphipp), HV(hS), nr, HV(hpart)))) return rc;
  LAB_HIP(c, launch_pointwise(PW_ADD, HV(hbpp), HV(hbpp), HV(hbpp + 1), 1, L, tabs, s));
  if ((rc = from_hat(c, HV(hbpp), 1, dense_tmp, cbpp))) return rc;

  // ---------------- step 7 (:301-314): phi_i = alpha phi^(F)_i + beta phi''_i  (layout [n][r] like phi')
  LAB_HIP(c, launch_poly_by_vec(HV(hphif), HV(hab), HV(hphi), nr, L, tabs, s));
  LAB_HIP(c, launch_poly_by_vec(HV(hbphi), HV(hab + 1), HV(hphipp), nr, L, tabs, s));
  LAB_HIP(c, launch_pointwise(PW_ADD, HV(hphif), HV(hphif), HV(hbphi), nr, L, tabs, s));
  if (!c->direct) {  // canonicalise before the next product (see header comment)
    if ((rc = from_hat(c, HV(hphif), nr, dense_tmp, cphif))) return rc;
    if ((rc = to_hat(c, cphif, HV(hphif), nr))) return rc;
  }

  // ---------------- step 8 (:320-358): H = 2^-1 (Phi^T S + S^T Phi)
  if ((rc = gram_impl(c, HV(hX), HV(hphif), HV(hS), n, r, HV(hpart)))) return rc;
  if ((rc = from_hat(c, HV(hX), rr, dense_tmp, cX))) return rc;
  LAB_HIP(c, launch_symmetrize_half(cH, cX, r, q, inv2_mod(q), s));

  // ---------------- step 9 (:364-378): u_2 = sum_{i<=j,k} h_ij^(k) D_ijk
  LAB_HIP(c, launch_gather_upper(cHup, cH, r, s));
  LAB_HIP(c, launch_decompose(cHdec, cHup, pairs, q, (uint64_t)in->b1, t1, 1, s));
  if ((rc = to_hat(c, cHdec, HD(hHdec), pairs * t1, bdig))) return rc;
  if ((rc = axpy_commit_any(c, HD(hu2), HD(hHdec), in->Dm, in->d_off, pairs * t1, k2, HD(hpartd)))) return rc;
  if ((rc = from_hat(c, HD(hu2), k2, dense_tmp, cu2, bdig))) return rc;

  // ---------------- step 10 (:380-399): z = sum_i c_i s_i
  LAB_HIP(c, launch_matvec_partial(HV(hz), HV(hS), HV(hc), 1, n, r, L, tabs, s));
  if ((rc = from_hat(c, HV(hz), n, dense_tmp, cz))) return rc;

  // ---------------- copy-out and the two verifier callbacks that can reject
  struct { uint64_t* dst; const uint64_t* src; size_t words; } outs[] = {
      {out->u_1, cu1, k1 * D}, {out->projection, dproj, 256}, {out->b_prime_prime, cbpp, D}, {out->u_2, cu2, k2 * D},
      {out->z, cz, n * D}, {out->t_i_all, cT, rk * D}, {out->g_mat, cG, rr * D}, {out->h_mat, cH, rr * D}};
  for (auto& o : outs)
    if (o.dst && o.dst != o.src) LAB_HIP(c, hipMemcpyAsync(o.dst, o.src, o.words * 8, hipMemcpyDefault, s));
  if (out->projection_int) LAB_HIP(c, hipMemcpyAsync(out->projection_int, dproj_int, 256 * 8, hipMemcpyDefault, s));
  int64_t hproj[256];
  uint64_t hbpp0 = 0, homega[256];
  LAB_HIP(c, hipMemcpyAsync(hproj, dproj_int, sizeof hproj, hipMemcpyDeviceToHost, s));
  LAB_HIP(c, hipMemcpyAsync(&hbpp0, cbpp, 8, hipMemcpyDeviceToHost, s));
  LAB_HIP(c, hipMemcpyAsync(homega, domega, sizeof homega, hipMemcpyDeviceToHost, s));
  LAB_HIP(c, hipStreamSynchronize(s));

  // Verifier::valid_projection (src/verification.rs:568-579): sqrt(f64(sum p^2)) <= sqrt(128) * beta
  __int128 ss = 0;
  for (int m = 0; m < 256; m++) ss += (__int128)hproj[m] * hproj[m];
  const double norm = sqrt((double)ss);
  if (!(norm <= sqrt(128.0) * (double)in->beta_bound))
    return fail(c, LAB_ERR_JL_REJECTED, "JL projection rejected: norm %.6g > %.6g", norm, sqrt(128.0) * (double)in->beta_bound);
  // verify_b_prime_prime (src/verification.rs:532-551): b''[0] == <omega, projection> + psi * b'
  u128 chk = 0;
  for (int m = 0; m < 256; m++) {
    __int128 pm = hproj[m] % (__int128)q;
    if (pm < 0) pm += q;
    chk = (chk + (u128)(homega[m] % q) * (u128)pm) % q;
  }
  chk = (chk + (u128)psi * (u128)(in->b_prime % q)) % q;
  if ((uint64_t)chk != hbpp0)
    return fail(c, LAB_ERR_BPP_CHECK, "verify_b_prime_prime check failed: b_prime_prime_k constant coefficient is: %llu, and check_val is: %llu",
                (unsigned long long)hbpp0, (unsigned long long)(uint64_t)chk);
  return LAB_OK;
}
