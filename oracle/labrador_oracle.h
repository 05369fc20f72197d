/*
 * labrador_oracle.h — CPU restatement of the LaBRADOR-SNARK prover hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / reported CPU baseline, never as the thing shipped.
 *
 * Every function restates a function of the Rust reference
 * (RatioAeterna/LaBRADOR-SNARK @ 2024-08-07); the file:line it follows is
 * given next to it (paths relative to the reference root).  No reference
 * source is copied: the reference is Rust over i128 `Zq` / `Polynomial<Zq>`
 * values, this is flat C over uint64_t[64] coefficient rows.
 *
 * PARITY PINNING STATUS: the reference holds NO golden vectors or known-answer
 * tests for this path, and it cannot be compiled here (no Rust toolchain, no
 * vendored crates).  The oracle is therefore pinned by
 *   (1) the reference's four property tests (tests/proptest.rs:13-24, 36-48,
 *       51-64, 67-81) re-asserted on it, including the saved regression inputs
 *       tests/proptest.proptest-regressions:9-11,
 *   (2) the reference's CI invariant verify(proof_gen(..)) == true
 *       (src/main.rs:106-107) re-asserted on orc_prove_core/orc_verify_core,
 *   (3) an independent pure-Python big-int twin (oracle/pyref.py) that must
 *       agree with it bit-for-bit (tests/test_oracle.py).
 * Value-level parity with the Rust binary itself is UNPINNED.
 *
 * Conventions: a ring element is uint64_t[64], canonical residues in [0,q),
 * index = degree, zero padded (the reference strips trailing zeros,
 * src/algebraic.rs:303-315; orc_trimmed_len() recovers that length).
 * q < 2^60 so 64-term sums of products fit unsigned __int128.
 */
#ifndef LABRADOR_ORACLE_H
#define LABRADOR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_D 64 /* src/constants.rs:15 */

/* ---- Zq / Rq (src/algebraic.rs) ------------------------------------- */

/* mod_positive on a signed 128-bit value given as (lo,hi); src/util.rs:16-23 */
uint64_t orc_mod_positive_i128(uint64_t lo, int64_t hi, uint64_t q);

/* number of stored coefficients after trailing-zero trim; algebraic.rs:303-315 */
int orc_trimmed_len(const uint64_t *p);

/* classic branch of Rq::multiply: schoolbook then X^64+1 fold;
 * src/algebraic.rs:402-403 + 352-376.  The semantic definition of the ring. */
void orc_polymul(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out);

/* Same result, computed the way the reference spends its time: an i128 `%`
 * after every coefficient multiply and add (algebraic.rs:30-38,95-134) and
 * the term-by-term fold (352-376).  Used only as the timed CPU baseline B0. */
void orc_polymul_faithful(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out);

/* NTT branch of Rq::multiply as the reference executes it: exact integer
 * negacyclic product taken mod 2^64 (what concrete-ntt 0.1.1
 * native64::Plan32::negacyclic_polymul returns), reinterpreted as two's
 * complement, then mod q; src/algebraic.rs:384-399, 417-420.  Equals
 * orc_polymul only while d*(q-1)^2 < 2^63. */
void orc_polymul_ntt_branch(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out);

void orc_poly_add(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out);   /* algebraic.rs:441-505 */
void orc_poly_sub(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out);   /* algebraic.rs:461-515 */
void orc_poly_scale(uint64_t q, const uint64_t *p, uint64_t s, uint64_t *out);        /* util.rs:176-180 */
void orc_sigma_inv(uint64_t q, const uint64_t *a, uint64_t *out);                     /* util.rs:118-137 */

/* ---- vector helpers (src/util.rs) ------------------------------------ */

/* polynomial_vec_inner_product; util.rs:496-509.  v1,v2: [len][64] */
void orc_inner_product(uint64_t q, const uint64_t *v1, const uint64_t *v2, size_t len, uint64_t *out);
/* poly_by_poly_vec; util.rs:289-296 */
void orc_poly_by_poly_vec(uint64_t q, const uint64_t *p, const uint64_t *v, size_t len, uint64_t *out);
/* add_poly_vec; util.rs:298-309 */
void orc_add_poly_vec(uint64_t q, const uint64_t *a, const uint64_t *b, size_t len, uint64_t *out);

/* decompose_polynomial (bug-compatible: digits are non-negative, sign of the
 * centred representative is lost; digits past `exp` are dropped); util.rs:377-442.
 * out: [exp][64].  Returns 0, or -1 if base < 2 (the reference never terminates). */
int orc_decompose(uint64_t q, const uint64_t *p, int64_t base, int exp, uint64_t *out);
/* decompose_polynomial_vec; util.rs:360-373.  v: [len][64]; out: [exp][len][64] */
int orc_decompose_vec(uint64_t q, const uint64_t *v, size_t len, int64_t base, int exp, uint64_t *out);

/* poly_norm: exact sum of squares of the RAW residues (lo,hi of the i128) and
 * its `as f64` cast; util.rs:195-202 */
double orc_poly_norm(const uint64_t *p, uint64_t *lo, uint64_t *hi);
/* vec_poly_norm_squared: left-to-right f64 sum of poly_norm; util.rs:188-192 */
double orc_vec_poly_norm_squared(const uint64_t *v, size_t len);
/* l2_norm over i128 given as int64 (values fit: |p| <= r*n*d*q); util.rs:216-219 */
double orc_l2_norm_i64(const int64_t *v, size_t len);

/* matmul(Pi, s) for one i: integer mat-vec without reduction;
 * util.rs:511-526 via proofgen.rs:429-456.  pi: [256][cols] in {-1,0,1};
 * s: [cols] raw residues; acc[256] += result. */
void orc_jl_accumulate(const int8_t *pi, const uint64_t *s, size_t cols, int64_t *acc);

/* ---- RuntimeConstants::new; src/constants.rs:232-264 ----------------- */
typedef struct {
  uint64_t n, r;
  int64_t beta_bound;
  double std;
  int64_t b, t1, b1, t2, b2;
  double gamma, gamma1, gamma2, beta_prime;
  uint64_t kappa, kappa1, kappa2;
} orc_constants;
void orc_runtime_constants(uint64_t q, uint64_t n, uint64_t r, orc_constants *out);

/* ---- deterministic synthetic inputs (shared with the GPU fill kernel) -- */
uint64_t orc_splitmix64(uint64_t x);
/* idx-th synthetic residue of stream `seed`, uniform in [0,bound) */
uint64_t orc_synth_residue(uint64_t seed, uint64_t idx, uint64_t bound);
void orc_synth_fill(uint64_t seed, uint64_t first_idx, size_t count, uint64_t bound, uint64_t *out);

/* ---- the prover core: Prover::proof_gen as a deterministic function ----
 * src/proofgen.rs:30-427 with the verifier's randomness supplied as a tape
 * and the CRS rows supplied as explicit matrices (the reference regenerates
 * them from a ChaCha20 seed, src/structs.rs:55-171 — out of scope here). */
typedef struct {
  uint64_t q;
  size_t n, r, kappa, kappa1, kappa2;
  int64_t b1, b2, beta_bound;
  int t1, t2;
  /* witness and state: [n][r][64] (ndarray row-major, column i = s_i; proofgen.rs:45) */
  const uint64_t *S, *phi;
  const uint64_t *a;       /* [r][r][64]  st.a_prime_k[0]                */
  uint64_t b_prime;        /* st.b_prime_k[0] = b.eval(0)  structs.rs:373 */
  /* CRS as matrices */
  const uint64_t *A;       /* [kappa][n][64]              structs.rs:55-72   */
  const uint64_t *B;       /* [r*t1][kappa1][kappa][64]   structs.rs:74-88   */
  const uint64_t *C;       /* [pairs*t2][kappa2][64], index pair(i,j)*t2+k; structs.rs:90-114 */
  const uint64_t *Dm;      /* [pairs*t1][kappa2][64], index pair(i,j)*t1+k; structs.rs:116-144 */
  /* challenge tape */
  const int8_t *Pi;        /* [r][256][n*64]  verification.rs:553-566 */
  uint64_t psi;            /* L = 1           verification.rs:491-501 */
  const uint64_t *omega;   /* [256]           verification.rs:503-513 */
  const uint64_t *alpha, *beta; /* [64] each  verification.rs:441-456 */
  const uint64_t *c;       /* [r][64]         verification.rs:460-489 */
} orc_prove_in;

typedef struct {
  uint64_t *t;        /* [r][kappa][64] */
  uint64_t *G, *H;    /* [r][r][64]     */
  uint64_t *u1;       /* [kappa1][64]   */
  uint64_t *u2;       /* [kappa2][64]   */
  int64_t *proj_int;  /* [256] integer projection */
  uint64_t *proj;     /* [256] Zq::lift(projection)  proofgen.rs:186 */
  uint64_t *bpp;      /* [64]  b''                                */
  uint64_t *z;        /* [n][64] */
  uint64_t *phi_final;/* [r][n][64] (internal to the reference; exposed for checks) or NULL */
  int jl_valid;       /* Verifier::valid_projection   verification.rs:568-579 */
  int bpp_ok;         /* verify_b_prime_prime          verification.rs:532-551 */
} orc_prove_out;

/* returns 0 on success, <0 on malformed parameters */
int orc_prove_core(const orc_prove_in *in, orc_prove_out *out);

/* Verifier::verify restated (src/verification.rs:25-438) on the same explicit
 * CRS matrices.  `b_poly` = st.b_k[0] ([64]); `bconst` = constants.B;
 * `beta_prime` = constants.BETA_PRIME.  Returns 0 if every check passes,
 * otherwise the number of the first failing protocol line (8..20). */
int orc_verify_core(const orc_prove_in *in, const orc_prove_out *proof, const uint64_t *b_poly,
                    int64_t bconst, double beta_prime, double *line14_sum);

/* State::gen_f's `b` (src/structs.rs:320-341): b = sum a_ij<s_i,s_j> + sum <phi_i,s_i> */
void orc_state_b(uint64_t q, size_t n, size_t r, const uint64_t *S, const uint64_t *phi,
                 const uint64_t *a, uint64_t *b_out);

/* ---- CRS random oracle (src/structs.rs:27-190) — SURVEY §8(f) row f1 ----------------
 * Every coefficient is `ChaCha20Rng::from_seed(seed).gen_range(0..Q)` on i128 with a 256-bit big-endian seed
 * that is incremented once per coefficient (src/structs.rs:35-45, 155-171).  The ChaCha20 block function is
 * pinned by RFC 7539 / djb test vectors (tests/test_crs.py); the rand_chacha 0.3.1 word order and rand 0.8.5
 * `UniformInt<i128>::sample_single` (Lemire widening multiply with the `(range << lz) - 1` zone) are restated
 * from their published algorithms — those crates are not in the reference tree, so that part is PARITY UNPINNED. */
void orc_chacha20_block(const uint32_t state[16], uint32_t out[16]);
/* one coefficient from one 32-byte seed; *draws (optional) = number of u128 candidates consumed */
uint64_t orc_crs_coeff(const uint8_t seed[32], uint64_t q, int *draws);
/* out[k] = coefficient for seed = base_seed + offset + k  (offset = off_hi*2^64 + off_lo), k < count */
void orc_crs_fill(const uint8_t base_seed[32], uint64_t off_lo, uint64_t off_hi, size_t count, uint64_t q, uint64_t *out);
/* seed offsets, in coefficient units, of the reference's fetch_* (copied literally, including their
 * inconsistent scaling): which = 0: fetch_A_row(row) :62 | 1: fetch_B_ik_row(i,k,row) :78-82 |
 * 2: fetch_C_ijk(i,j,k) :94-108 | 3: fetch_D_ijk(i,j,k) :120-138.  Result as lo/hi of a u128. */
void orc_crs_offset(int which, size_t n, size_t r, size_t kappa, size_t kappa1, size_t kappa2, int t1,
                    size_t i, size_t j, size_t k, size_t row, uint64_t *lo, uint64_t *hi);

/* ---- timed CPU baselines (bench.py cpu_baseline leg) ------------------ */
/* B0: `count` reference-faithful inner-product terms (polymul_faithful + add);
 * returns a checksum so the work cannot be optimised away. */
uint64_t orc_bench_inner_product_faithful(uint64_t q, const uint64_t *v1, const uint64_t *v2, size_t len, int reps);

#ifdef __cplusplus
}
#endif
#endif
