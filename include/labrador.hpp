// labrador.hpp — header-only C++ host mirror of the reference's ring/vector interface over the C ABI in
// labrador_hip.h.  Names, argument meaning and error behaviour follow the Rust reference
// (RatioAeterna/LaBRADOR-SNARK): `Rq` is a padded 64-coefficient polynomial, `polynomial_vec_inner_product`
// throws where the reference's assert! panics (src/util.rs:497-502), `Prover::proof_gen` throws
// "failed JL..." after more than five rejected projections (src/proofgen.rs:169-181), and so on.
// It contains no arithmetic of its own: every operation is a call into liblabrador_hip.so.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "labrador_hip.h"

namespace labrador {

constexpr size_t D = LAB_D;  // src/constants.rs:15
using Rq = std::array<uint64_t, D>;  // canonical residues, index = degree, zero padded (src/algebraic.rs:385-394)

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
struct LengthMismatch : Error {  // the reference's assert!(v1.len() == v2.len(), ...)
  using Error::Error;
};

class Context {  // stands for the lazy statics Q and PLAN (src/constants.rs:193-198)
 public:
  explicit Context(uint64_t q, int device = 0, uint64_t max_len = 0) {
    int rc = lab_ctx_create(q, device, max_len, &h_);
    if (rc != LAB_OK) throw Error(rc, rc == LAB_ERR_NO_DEVICE ? "no usable GPU (there is no CPU fallback)" : "lab_ctx_create failed");
    lab_ctx_info(h_, &info_);
  }
  ~Context() { lab_ctx_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  lab_ctx* handle() const { return h_; }
  uint64_t q() const { return info_.q; }
  int nlimbs() const { return info_.nlimbs; }
  // second RNS basis for B, C, D (and a copy of A): primes below 2^23, three byte planes per residue on the matrix cores
  void enable_matrix_core_basis(uint64_t max_terms, uint64_t digit_bound) const {
    check(lab_ctx_enable_digit_basis_mfma(h_, max_terms, digit_bound));
  }
  void check(int rc) const {
    if (rc == LAB_OK) return;
    std::string msg = lab_last_error(h_);
    if (rc == LAB_ERR_LEN_MISMATCH) throw LengthMismatch(rc, msg.empty() ? "inner product not defined on vectors of unequal length" : msg);
    throw Error(rc, msg);
  }

 private:
  lab_ctx* h_ = nullptr;
  lab_info info_{};
};

class HatVec {  // device-resident ring elements in the transform domain
 public:
  HatVec(const Context& c, size_t count) : c_(c), count_(count) { c_.check(lab_vec_create(c_.handle(), count, &h_)); }
  // in the context's digit basis (Context::enable_matrix_core_basis)
  HatVec(const Context& c, size_t count, bool digit) : c_(c), count_(count) {
    c_.check(digit ? lab_vec_create_digit(c_.handle(), count, 0, &h_) : lab_vec_create(c_.handle(), count, &h_));
  }
  HatVec(const Context& c, const std::vector<Rq>& v) : HatVec(c, v.size()) {
    c_.check(lab_vec_load_u64(c_.handle(), h_, 0, v.empty() ? nullptr : v[0].data(), v.size()));
  }
  ~HatVec() { lab_vec_destroy(h_); }
  HatVec(const HatVec&) = delete;
  HatVec& operator=(const HatVec&) = delete;
  lab_vec* handle() const { return h_; }
  size_t size() const { return count_; }
  std::vector<Rq> to_host(size_t off = 0, size_t count = SIZE_MAX) const {
    if (count == SIZE_MAX) count = count_ - off;
    std::vector<Rq> out(count);
    c_.check(lab_vec_store_u64(c_.handle(), h_, off, count, count ? out[0].data() : nullptr));
    return out;
  }

 private:
  const Context& c_;
  size_t count_;
  lab_vec* h_ = nullptr;
};

// &lhs * &rhs  — Rq::multiply (src/algebraic.rs:379-404)
inline Rq multiply(const Context& c, const Rq& lhs, const Rq& rhs) {
  Rq out;
  c.check(lab_negacyclic_polymul_u64(c.handle(), out.data(), lhs.data(), rhs.data(), 1));
  return out;
}

// src/util.rs:496-509
inline Rq polynomial_vec_inner_product(const Context& c, const std::vector<Rq>& v1, const std::vector<Rq>& v2) {
  if (v1.size() != v2.size())
    throw LengthMismatch(LAB_ERR_LEN_MISMATCH, "inner product not defined on vectors of unequal length. v1 length: " +
                                                   std::to_string(v1.size()) + ", v2 length: " + std::to_string(v2.size()));
  if (v1.empty()) return Rq{};  // Rq::new(vec![])
  HatVec a(c, v1), b(c, v2), out(c, 1);
  c.check(lab_inner_product(c.handle(), out.handle(), 0, a.handle(), 0, b.handle(), 0, v1.size()));
  return out.to_host()[0];
}

// src/util.rs:289-296
inline std::vector<Rq> poly_by_poly_vec(const Context& c, const Rq& poly, const std::vector<Rq>& v1) {
  if (v1.empty()) return {};
  HatVec p(c, std::vector<Rq>{poly}), v(c, v1), out(c, v1.size());
  c.check(lab_poly_by_poly_vec(c.handle(), out.handle(), 0, p.handle(), 0, v.handle(), 0, v1.size()));
  return out.to_host();
}

// src/util.rs:298-309
inline std::vector<Rq> add_poly_vec(const Context& c, const std::vector<Rq>& v1, const std::vector<Rq>& v2) {
  if (v1.size() != v2.size()) throw LengthMismatch(LAB_ERR_LEN_MISMATCH, "summation not defined on vectors of unequal length");
  if (v1.empty()) return {};
  HatVec a(c, v1), b(c, v2), out(c, v1.size());
  c.check(lab_pointwise(c.handle(), LAB_OP_ADD, out.handle(), 0, a.handle(), 0, b.handle(), 0, v1.size()));
  return out.to_host();
}

// src/util.rs:167-180
inline std::vector<Rq> scale_poly_vec(const Context& c, const std::vector<Rq>& v, uint64_t s) {
  std::vector<Rq> out(v.size());
  if (!v.empty()) c.check(lab_scale(c.handle(), out[0].data(), v[0].data(), v.size(), s));
  return out;
}

// src/util.rs:107-137
inline std::vector<Rq> sigma_inv_vec(const Context& c, const std::vector<Rq>& v) {
  std::vector<Rq> out(v.size());
  if (!v.empty()) c.check(lab_sigma_inv(c.handle(), out[0].data(), v[0].data(), v.size()));
  return out;
}

// src/util.rs:360-373 — result[k][i] = k-th digit polynomial of vec[i]
inline std::vector<std::vector<Rq>> decompose_polynomial_vec(const Context& c, const std::vector<Rq>& vec, int64_t base, int exp) {
  std::vector<Rq> flat((size_t)(exp > 0 ? exp : 0) * vec.size());
  if (!vec.empty()) c.check(lab_decompose(c.handle(), flat.empty() ? nullptr : flat[0].data(), vec[0].data(), vec.size(), base, exp));
  std::vector<std::vector<Rq>> res((size_t)(exp > 0 ? exp : 0));
  for (size_t k = 0; k < res.size(); k++) res[k].assign(flat.begin() + k * vec.size(), flat.begin() + (k + 1) * vec.size());
  return res;
}

// src/util.rs:188-202: exact integer norms come from the device, the f64 fold happens here in index order
inline double vec_poly_norm_squared(const Context& c, const std::vector<Rq>& v) {
  std::vector<uint64_t> lo(v.size()), hi(v.size());
  if (!v.empty()) c.check(lab_norm_sq(c.handle(), v[0].data(), v.size(), lo.data(), hi.data()));
  double s = 0.0;
  for (size_t i = 0; i < v.size(); i++) s += (double)(((unsigned __int128)hi[i] << 64) | lo[i]);
  return s;
}
inline double poly_norm(const Context& c, const Rq& p) { return vec_poly_norm_squared(c, std::vector<Rq>{p}); }

// src/constants.rs:205-264, with explicit commitment ranks (0 = the reference's n*64)
struct RuntimeConstants {
  size_t N, R, KAPPA, KAPPA_1, KAPPA_2;
  int64_t BETA_BOUND, B, T_1, B_1, T_2, B_2;
  double STD, GAMMA, GAMMA_1, GAMMA_2, BETA_PRIME;
  RuntimeConstants(uint64_t q, size_t n, size_t r, size_t kappa = 0, size_t kappa1 = 0, size_t kappa2 = 0) : N(n), R(r) {
    const double TAU = 71.0, Q = (double)q;
    KAPPA = kappa ? kappa : n * D;
    KAPPA_1 = kappa1 ? kappa1 : n * D;
    KAPPA_2 = kappa2 ? kappa2 : n * D;
    BETA_BOUND = (int64_t)std::floor(std::sqrt(30.0 / 128.0) * Q / 125.0);
    STD = (double)BETA_BOUND / std::sqrt((double)(r * n * D));
    B = (int64_t)std::round(std::sqrt(std::sqrt(12. * (double)r * TAU) * STD));
    T_1 = (int64_t)std::round(std::log2(Q) / std::log2((double)B));
    B_1 = (int64_t)std::pow(Q, 1.0 / (double)T_1);
    T_2 = (int64_t)std::round(std::log2(std::sqrt(24. * (double)(n * D)) * (STD * STD)) / std::log2((double)B));
    B_2 = (int64_t)std::round(std::pow(std::sqrt((double)(24 * (n * D))) * (STD * STD), 1.0 / (double)T_2));
    const double b1 = (double)B_1, b2 = (double)B_2, t1 = (double)T_1, t2 = (double)T_2, rr = (double)r;
    GAMMA = (double)(BETA_BOUND * BETA_BOUND) * TAU;
    GAMMA_1 = ((b1 * b1 * t1) / 12.0) * rr * (double)KAPPA * (double)D + ((b2 * b2 * t2) / 12.0) * ((rr * rr + rr) / 2.0) * (double)D;
    GAMMA_2 = ((b1 * b1 * t1) / 12.0) * ((rr * rr + rr) / 2.0) * (double)D;
    BETA_PRIME = (2.0 / ((double)B * (double)B)) * GAMMA + GAMMA_1 + GAMMA_2;
  }
};

// the computed fields of src/structs.rs:192-209, flat row-major
struct Transcript {
  std::vector<Rq> u_1, u_2, z, t_i_all /* [r][kappa] */, g_mat /* [r][r] */, h_mat;
  std::vector<int64_t> projection_int;
  std::vector<uint64_t> projection;
  Rq b_prime_prime{};
};

// the verifier's interactive randomness as the prover consumes it (src/verification.rs:441-513, 553-566)
struct ChallengeTape {
  std::vector<int8_t> Pi;  // [r][256][n*64]
  uint64_t psi = 0;
  std::vector<uint64_t> omega;  // [256]
  Rq alpha{}, beta{};
  std::vector<Rq> c;  // [r]
};

// matrices re-stored for the matrix cores (lab_mat8_pack): signed base-256 digit planes in MFMA operand-tile order
class PackedMatrix {
 public:
  PackedMatrix() = default;
  PackedMatrix(const Context& c, const HatVec& M, size_t nmat, size_t rows, size_t cols) {
    c.check(lab_mat8_pack(c.handle(), M.handle(), 0, nmat, rows, cols, &m_));
  }
  PackedMatrix(PackedMatrix&& o) noexcept : m_(o.m_) { o.m_ = nullptr; }
  PackedMatrix& operator=(PackedMatrix&& o) noexcept {
    if (this != &o) { reset(); m_ = o.m_; o.m_ = nullptr; }
    return *this;
  }
  PackedMatrix(const PackedMatrix&) = delete;
  PackedMatrix& operator=(const PackedMatrix&) = delete;
  ~PackedMatrix() { reset(); }
  const lab_mat8* handle() const { return m_; }
  size_t bytes() const { return lab_mat8_bytes(m_); }

 private:
  void reset() { if (m_) lab_mat8_destroy(m_); m_ = nullptr; }
  lab_mat8* m_ = nullptr;
};

struct ResidentCRS {  // A, B_ik, C_ijk, D_ijk in the transform domain (stand-in for CRS::fetch_*, src/structs.rs:55-144)
  HatVec A, B, C, Dm;
  PackedMatrix A8, B8, C8, D8, A8d;   // optional (pack_mfma): the CRS-streaming steps of a batch then run on the matrix cores
  bool digit = false;                 // B, C, D live in the context's digit basis
  // matrix_core_basis: B, C, D in the digit basis of primes below 2^23 (enabled on the context here if it is not yet)
  ResidentCRS(const Context& c, const RuntimeConstants& k, bool matrix_core_basis = false)
      : A(c, k.KAPPA * k.N), B(c, k.R * (size_t)k.T_1 * k.KAPPA_1 * k.KAPPA, enable(c, k, matrix_core_basis)),
        C(c, k.R * (k.R + 1) / 2 * (size_t)k.T_2 * k.KAPPA_2, matrix_core_basis),
        Dm(c, k.R * (k.R + 1) / 2 * (size_t)k.T_1 * k.KAPPA_2, matrix_core_basis), digit(matrix_core_basis) {}
  void pack_mfma(const Context& c, const RuntimeConstants& k) {
    const size_t pairs = k.R * (k.R + 1) / 2;
    B8 = PackedMatrix(c, B, k.R * (size_t)k.T_1, k.KAPPA_1, k.KAPPA);
    A8 = PackedMatrix(c, A, 1, k.KAPPA, k.N);
    C8 = PackedMatrix(c, C, pairs * (size_t)k.T_2, k.KAPPA_2, 1);   // element (rho, m) = C[m][rho]
    D8 = PackedMatrix(c, Dm, pairs * (size_t)k.T_1, k.KAPPA_2, 1);
    if (digit) {   // A once more in the basis of B, C, D: step 1 of a batch runs there while the witness keeps its norm bound
      HatVec Ad(c, k.KAPPA * k.N, true);
      const std::vector<Rq> coeffs = A.to_host();
      c.check(lab_vec_load_u64(c.handle(), Ad.handle(), 0, coeffs[0].data(), coeffs.size()));
      A8d = PackedMatrix(c, Ad, 1, k.KAPPA, k.N);
      c.check(lab_ctx_synchronize(c.handle()));
    }
  }

 private:
  static bool enable(const Context& c, const RuntimeConstants& k, bool on) {
    lab_info di{};
    if (on && (lab_ctx_digit_info(c.handle(), &di) != LAB_OK || di.nlimbs == 0)) {
      const size_t pairs = k.R * (k.R + 1) / 2;
      c.enable_matrix_core_basis(k.R * (size_t)k.T_1 * k.KAPPA + pairs * (size_t)std::max(k.T_1, k.T_2), (uint64_t)std::max(k.B_1, k.B_2));
    }
    return on;
  }
};

struct State {  // the parts of src/structs.rs:269-286 the prover reads
  std::vector<Rq> phi;  // [n][r]
  std::vector<Rq> a;    // [r][r]
  uint64_t b_prime = 0;
};

class Prover {  // src/proofgen.rs:14-28
 public:
  Prover(const std::vector<Rq>& witness /* [n][r] */, const RuntimeConstants& constants, const Context& ctx)
      : witness_(witness), k_(constants), c_(ctx) {}

  // One attempt with a given tape.  Throws Error(LAB_ERR_JL_REJECTED) / Error(LAB_ERR_BPP_CHECK) like lab_prove_core.
  Transcript prove_with_tape(const State& st, const ResidentCRS& crs, const ChallengeTape& tape) const {
    lab_prove_input in = input(k_, witness_, st, crs, tape);
    Transcript T;
    lab_transcript out = output(k_, T);
    c_.check(lab_prove_core(c_.handle(), &in, &out));
    return T;
  }

  // Many independent proofs (own witness, state, tape) under ONE common reference string: lab_prove_core_batch.  The
  // CRS-streaming steps run once per group of proofs (up to 64 per pass over B after crs.pack_mfma()).  status[j] is
  // LAB_OK / LAB_ERR_JL_REJECTED / LAB_ERR_BPP_CHECK for proof j; transcripts are bit-identical to prove_with_tape.
  static std::vector<Transcript> prove_batch(const Context& c, const RuntimeConstants& k,
                                             const std::vector<const std::vector<Rq>*>& witnesses,
                                             const std::vector<const State*>& states, const ResidentCRS& crs,
                                             const std::vector<const ChallengeTape*>& tapes, std::vector<int>& status) {
    const size_t np = witnesses.size();
    if (states.size() != np || tapes.size() != np) throw LengthMismatch(LAB_ERR_LEN_MISMATCH, "prove_batch: one witness, state and tape per proof");
    std::vector<lab_prove_input> ins(np);
    std::vector<lab_transcript> outs(np);
    std::vector<Transcript> T(np);
    std::vector<const lab_prove_input*> pin(np);
    std::vector<lab_transcript*> pout(np);
    for (size_t j = 0; j < np; j++) {
      ins[j] = input(k, *witnesses[j], *states[j], crs, *tapes[j]);
      outs[j] = output(k, T[j]);
      pin[j] = &ins[j];
      pout[j] = &outs[j];
    }
    status.assign(np, LAB_OK);
    c.check(lab_prove_core_batch(c.handle(), np, pin.data(), pout.data(), status.data()));
    return T;
  }

  // proof_gen(&mut self, st, crs): `sample` is called for a fresh tape each time the projection is rejected;
  // more than five rejections -> "failed JL..." (src/proofgen.rs:169-181)
  template <typename Sampler>
  Transcript proof_gen(const State& st, const ResidentCRS& crs, Sampler&& sample) const {
    for (int rejections = 0;; rejections++) {
      if (rejections > 5) throw std::runtime_error("failed JL...");
      try {
        return prove_with_tape(st, crs, sample());
      } catch (const Error& e) {
        if (e.code != LAB_ERR_JL_REJECTED) throw;
      }
    }
  }

 private:
  static lab_prove_input input(const RuntimeConstants& k, const std::vector<Rq>& witness, const State& st, const ResidentCRS& crs,
                               const ChallengeTape& tape) {
    lab_prove_input in{};
    in.struct_size = sizeof(lab_prove_input);
    in.n = k.N; in.r = k.R; in.kappa = k.KAPPA; in.kappa1 = k.KAPPA_1; in.kappa2 = k.KAPPA_2;
    in.b1 = k.B_1; in.b2 = k.B_2; in.t1 = (int)k.T_1; in.t2 = (int)k.T_2; in.beta_bound = k.BETA_BOUND;
    in.S = witness[0].data(); in.phi = st.phi[0].data(); in.a = st.a[0].data(); in.b_prime = st.b_prime;
    in.A = crs.A.handle(); in.B = crs.B.handle(); in.C = crs.C.handle(); in.Dm = crs.Dm.handle();
    in.A8 = crs.A8.handle(); in.B8 = crs.B8.handle(); in.C8 = crs.C8.handle(); in.D8 = crs.D8.handle();
    in.A8d = crs.A8d.handle();
    in.Pi = tape.Pi.data(); in.psi = tape.psi; in.omega = tape.omega.data();
    in.alpha = tape.alpha.data(); in.beta = tape.beta.data(); in.c = tape.c[0].data();
    return in;
  }
  static lab_transcript output(const RuntimeConstants& k, Transcript& T) {
    T.u_1.resize(k.KAPPA_1); T.u_2.resize(k.KAPPA_2); T.z.resize(k.N); T.t_i_all.resize(k.R * k.KAPPA);
    T.g_mat.resize(k.R * k.R); T.h_mat.resize(k.R * k.R); T.projection_int.resize(256); T.projection.resize(256);
    lab_transcript out{};
    out.u_1 = T.u_1[0].data(); out.projection_int = T.projection_int.data(); out.projection = T.projection.data();
    out.b_prime_prime = T.b_prime_prime.data(); out.u_2 = T.u_2[0].data(); out.z = T.z[0].data();
    out.t_i_all = T.t_i_all[0].data(); out.g_mat = T.g_mat[0].data(); out.h_mat = T.h_mat[0].data();
    return out;
  }
  const std::vector<Rq>& witness_;
  const RuntimeConstants& k_;
  const Context& c_;
};

}  // namespace labrador
