// mirror_test.cpp — parity of the C++ host mirror (include/labrador.hpp) against the CPU oracle, written the way
// the reference's tests/proptest.rs reads.  Test infrastructure: links oracle/liblabrador_oracle.so as the checker.
// Build + run: see tests/test_cpp_mirror.py (g++, on the GPU box).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../include/labrador.hpp"
#include "../../oracle/labrador_oracle.h"

using namespace labrador;

#define CHECK(cond)                                                       \
  do {                                                                    \
    if (!(cond)) {                                                        \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      exit(1);                                                            \
    }                                                                     \
  } while (0)

static Rq generate_polynomial(std::mt19937_64& rng, uint64_t q) {  // src/util.rs:27-35
  Rq p;
  for (auto& c : p) c = rng() % q;
  return p;
}

static void ring_tests(uint64_t q) {
  Context ctx(q);
  std::mt19937_64 rng(q);
  // test_NTT_preserves_result (tests/proptest.rs:13-24): device product == classic product
  for (int it = 0; it < 20; it++) {
    Rq a = generate_polynomial(rng, q), b = generate_polynomial(rng, q), want;
    orc_polymul(q, a.data(), b.data(), want.data());
    CHECK(multiply(ctx, a, b) == want);
  }
  // test_linearity_of_poly_vec_inner_product (tests/proptest.rs:36-48), N = 16
  for (int it = 0; it < 5; it++) {
    std::vector<Rq> a(16), b(16);
    for (auto& p : a) p = generate_polynomial(rng, q);
    for (auto& p : b) p = generate_polynomial(rng, q);
    uint64_t c = rng() % q;
    Rq ab = polynomial_vec_inner_product(ctx, a, b), want;
    orc_inner_product(q, a[0].data(), b[0].data(), 16, want.data());
    CHECK(ab == want);
    Rq lhs = polynomial_vec_inner_product(ctx, a, scale_poly_vec(ctx, b, c));
    CHECK(lhs == scale_poly_vec(ctx, std::vector<Rq>{ab}, c)[0]);
  }
  // assert!(v1.len() == v2.len()) -> LengthMismatch
  bool threw = false;
  try {
    polynomial_vec_inner_product(ctx, std::vector<Rq>(3), std::vector<Rq>(4));
  } catch (const LengthMismatch&) {
    threw = true;
  }
  CHECK(threw);
  // decompose_polynomial_vec / poly_by_poly_vec / add_poly_vec / norms / sigma_inv
  std::vector<Rq> v(7);
  for (auto& p : v) p = generate_polynomial(rng, q);
  auto dec = decompose_polynomial_vec(ctx, v, 9, 4);
  std::vector<uint64_t> want(4 * 7 * D);
  CHECK(orc_decompose_vec(q, v[0].data(), 7, 9, 4, want.data()) == 0);
  for (int k = 0; k < 4; k++)
    for (int i = 0; i < 7; i++) CHECK(memcmp(dec[k][i].data(), &want[((size_t)k * 7 + i) * D], D * 8) == 0);
  Rq p = generate_polynomial(rng, q);
  auto pv = poly_by_poly_vec(ctx, p, v);
  auto sum = add_poly_vec(ctx, pv, v);
  auto sg = sigma_inv_vec(ctx, v);
  for (int i = 0; i < 7; i++) {
    Rq w, w2, w3;
    orc_polymul(q, p.data(), v[i].data(), w.data());
    CHECK(pv[i] == w);
    orc_poly_add(q, w.data(), v[i].data(), w2.data());
    CHECK(sum[i] == w2);
    orc_sigma_inv(q, v[i].data(), w3.data());
    CHECK(sg[i] == w3);
  }
  CHECK(vec_poly_norm_squared(ctx, v) == orc_vec_poly_norm_squared(v[0].data(), 7));
  printf("ring tests ok: q=%llu nlimbs=%d\n", (unsigned long long)q, ctx.nlimbs());
}

static void constants_tests() {
  for (uint64_t q : {8191ull, (1ull << 32) + 15}) {
    for (auto nr : {std::pair<size_t, size_t>{2, 2}, {16, 16}}) {
      RuntimeConstants k(q, nr.first, nr.second);
      orc_constants o;
      orc_runtime_constants(q, nr.first, nr.second, &o);
      CHECK(k.BETA_BOUND == o.beta_bound && k.B == o.b && k.T_1 == o.t1 && k.B_1 == o.b1 && k.T_2 == o.t2 && k.B_2 == o.b2);
      CHECK(k.BETA_PRIME == o.beta_prime && k.STD == o.std);
    }
  }
  printf("constants ok\n");
}

static void prover_test(uint64_t q) {
  const size_t n = 2, r = 2, kappa = 5, kappa1 = 4, pairs = 3;
  Context ctx(q);
  RuntimeConstants k(q, n, r, kappa, kappa1, kappa1);
  const int t1 = (int)k.T_1, t2 = (int)k.T_2;
  std::mt19937_64 rng(99);
  const uint64_t wb = std::max<uint64_t>(1, (uint64_t)((double)k.BETA_BOUND / std::sqrt((double)(n * r * D))));
  std::vector<Rq> S(n * r), phi(n * r), a(r * r);
  for (auto& p : S) for (auto& c : p) c = rng() % (wb + 1);
  for (auto& p : phi) p = generate_polynomial(rng, q);
  for (size_t i = 0; i < r; i++) for (size_t j = i; j < r; j++) { a[i * r + j] = generate_polynomial(rng, q); a[j * r + i] = a[i * r + j]; }
  ChallengeTape tape;
  tape.Pi.resize(r * 256 * n * D);
  for (auto& e : tape.Pi) { uint64_t u = rng() % 4; e = u == 0 ? -1 : (u == 1 ? 1 : 0); }
  tape.psi = rng() % q;
  tape.omega.resize(256);
  for (auto& w : tape.omega) w = rng() % q;
  tape.alpha = generate_polynomial(rng, q);
  tape.beta = generate_polynomial(rng, q);
  tape.c.resize(r);
  for (auto& p : tape.c) for (auto& c : p) c = rng() % 3;
  // synthetic CRS: same splitmix64 streams on both sides
  ResidentCRS crs(ctx, k);
  ctx.check(lab_vec_fill_synthetic(ctx.handle(), crs.A.handle(), 0, crs.A.size(), 1000, 0, q));
  ctx.check(lab_vec_fill_synthetic(ctx.handle(), crs.B.handle(), 0, crs.B.size(), 1001, 0, q));
  ctx.check(lab_vec_fill_synthetic(ctx.handle(), crs.C.handle(), 0, crs.C.size(), 1002, 0, q));
  ctx.check(lab_vec_fill_synthetic(ctx.handle(), crs.Dm.handle(), 0, crs.Dm.size(), 1003, 0, q));
  std::vector<uint64_t> A(crs.A.size() * D), B(crs.B.size() * D), C(crs.C.size() * D), Dm(crs.Dm.size() * D);
  orc_synth_fill(1000, 0, A.size(), q, A.data());
  orc_synth_fill(1001, 0, B.size(), q, B.data());
  orc_synth_fill(1002, 0, C.size(), q, C.data());
  orc_synth_fill(1003, 0, Dm.size(), q, Dm.data());
  Rq bpoly;
  orc_state_b(q, n, r, S[0].data(), phi[0].data(), a[0].data(), bpoly.data());
  State st{phi, a, bpoly[0]};
  // oracle
  orc_prove_in in{};
  in.q = q; in.n = n; in.r = r; in.kappa = kappa; in.kappa1 = kappa1; in.kappa2 = kappa1;
  in.b1 = k.B_1; in.b2 = k.B_2; in.beta_bound = k.BETA_BOUND; in.t1 = t1; in.t2 = t2;
  in.S = S[0].data(); in.phi = phi[0].data(); in.a = a[0].data(); in.b_prime = bpoly[0];
  in.A = A.data(); in.B = B.data(); in.C = C.data(); in.Dm = Dm.data();
  in.Pi = tape.Pi.data(); in.psi = tape.psi; in.omega = tape.omega.data(); in.alpha = tape.alpha.data();
  in.beta = tape.beta.data(); in.c = tape.c[0].data();
  std::vector<uint64_t> ot(r * kappa * D), oG(r * r * D), oH(r * r * D), ou1(kappa1 * D), ou2(kappa1 * D), oproj(256), obpp(D), oz(n * D);
  std::vector<int64_t> oproj_int(256);
  orc_prove_out out{};
  out.t = ot.data(); out.G = oG.data(); out.H = oH.data(); out.u1 = ou1.data(); out.u2 = ou2.data();
  out.proj_int = oproj_int.data(); out.proj = oproj.data(); out.bpp = obpp.data(); out.z = oz.data();
  CHECK(orc_prove_core(&in, &out) == 0);
  CHECK(out.bpp_ok);
  Prover prover(S, k, ctx);
  if (!out.jl_valid) {
    bool threw = false;
    try { prover.proof_gen(st, crs, [&] { return tape; }); } catch (const std::runtime_error& e) { threw = std::string(e.what()) == "failed JL..."; }
    CHECK(threw);
    printf("prover test: tape rejected by valid_projection on both sides (q=%llu)\n", (unsigned long long)q);
    return;
  }
  Transcript T = prover.proof_gen(st, crs, [&] { return tape; });
  CHECK(memcmp(T.t_i_all[0].data(), ot.data(), ot.size() * 8) == 0);
  CHECK(memcmp(T.g_mat[0].data(), oG.data(), oG.size() * 8) == 0);
  CHECK(memcmp(T.h_mat[0].data(), oH.data(), oH.size() * 8) == 0);
  CHECK(memcmp(T.u_1[0].data(), ou1.data(), ou1.size() * 8) == 0);
  CHECK(memcmp(T.u_2[0].data(), ou2.data(), ou2.size() * 8) == 0);
  CHECK(memcmp(T.z[0].data(), oz.data(), oz.size() * 8) == 0);
  CHECK(memcmp(T.b_prime_prime.data(), obpp.data(), D * 8) == 0);
  CHECK(memcmp(T.projection_int.data(), oproj_int.data(), 256 * 8) == 0);
  // the batched call, first on the multiply-add kernels, then with the CRS packed for the matrix cores: the same
  // three proofs (one of them with a wrong b') give the same transcripts and per-proof status codes
  {
    State wrong = st;
    wrong.b_prime = (st.b_prime + 1) % q;
    std::vector<const std::vector<Rq>*> ws = {&S, &S, &S};
    std::vector<const State*> sts = {&st, &wrong, &st};
    std::vector<const ChallengeTape*> tps = {&tape, &tape, &tape};
    for (int pass = 0; pass < 2; pass++) {
      if (pass == 1) {
        crs.pack_mfma(ctx, k);
        CHECK(crs.B8.bytes() > 0 && crs.A8.bytes() > 0);
      }
      std::vector<int> status;
      std::vector<Transcript> Tb = Prover::prove_batch(ctx, k, ws, sts, crs, tps, status);
      CHECK(status.size() == 3 && status[0] == LAB_OK && status[1] == LAB_ERR_BPP_CHECK && status[2] == LAB_OK);
      for (int j : {0, 2}) {
        CHECK(memcmp(Tb[j].u_1[0].data(), ou1.data(), ou1.size() * 8) == 0);
        CHECK(memcmp(Tb[j].u_2[0].data(), ou2.data(), ou2.size() * 8) == 0);
        CHECK(memcmp(Tb[j].t_i_all[0].data(), ot.data(), ot.size() * 8) == 0);
        CHECK(memcmp(Tb[j].z[0].data(), oz.data(), oz.size() * 8) == 0);
        CHECK(memcmp(Tb[j].h_mat[0].data(), oH.data(), oH.size() * 8) == 0);
      }
    }
  }
  // the same CRS kept in the matrix-core digit basis (primes below 2^23, three byte planes; step 1 in that basis while the
  // witness keeps its norm bound): identical transcripts
  if (ctx.nlimbs() > 1) {
    ResidentCRS crs23(ctx, k, true);
    ctx.check(lab_vec_fill_synthetic(ctx.handle(), crs23.A.handle(), 0, crs23.A.size(), 1000, 0, q));
    ctx.check(lab_vec_fill_synthetic(ctx.handle(), crs23.B.handle(), 0, crs23.B.size(), 1001, 0, q));
    ctx.check(lab_vec_fill_synthetic(ctx.handle(), crs23.C.handle(), 0, crs23.C.size(), 1002, 0, q));
    ctx.check(lab_vec_fill_synthetic(ctx.handle(), crs23.Dm.handle(), 0, crs23.Dm.size(), 1003, 0, q));
    crs23.pack_mfma(ctx, k);
    CHECK(crs23.A8d.bytes() > 0);
    std::vector<const std::vector<Rq>*> ws = {&S, &S};
    std::vector<const State*> sts = {&st, &st};
    std::vector<const ChallengeTape*> tps = {&tape, &tape};
    std::vector<int> status;
    std::vector<Transcript> Tb = Prover::prove_batch(ctx, k, ws, sts, crs23, tps, status);
    CHECK(status.size() == 2 && status[0] == LAB_OK && status[1] == LAB_OK);
    const int flags = lab_ctx_prove_flags(ctx.handle());   // 1: step 1 in the digit basis; 2: witness above the norm bound, redone
    CHECK(flags == 1 || flags == 2);
    printf("matrix-core basis pass: prove flags %d\n", flags);
    for (int j : {0, 1}) {
      CHECK(memcmp(Tb[j].u_1[0].data(), ou1.data(), ou1.size() * 8) == 0);
      CHECK(memcmp(Tb[j].u_2[0].data(), ou2.data(), ou2.size() * 8) == 0);
      CHECK(memcmp(Tb[j].t_i_all[0].data(), ot.data(), ot.size() * 8) == 0);
      CHECK(memcmp(Tb[j].z[0].data(), oz.data(), oz.size() * 8) == 0);
    }
  }
  // a wrong b' trips verify_b_prime_prime (src/verification.rs:550)
  State bad = st;
  bad.b_prime = (st.b_prime + 1) % q;
  bool threw = false;
  try { prover.prove_with_tape(bad, crs, tape); } catch (const Error& e) { threw = e.code == LAB_ERR_BPP_CHECK; }
  CHECK(threw);
  (void)pairs;
  printf("prover test ok: q=%llu\n", (unsigned long long)q);
}

int main() {
  ring_tests(8191);
  ring_tests((1ull << 32) + 15);
  constants_tests();
  prover_test(8191);
  prover_test((1ull << 32) + 15);
  printf("ALL OK\n");
  return 0;
}
