/*
 * labrador_hip.h — C ABI of the MI355X-native LaBRADOR prover hot path.
 *
 * This is the drop-in boundary: plain C, raw pointers and sizes, no torch or
 * C++ types.  Each entry point names the reference interface it replaces
 * (file:line in RatioAeterna/LaBRADOR-SNARK @ 2024-08-07).  INTEGRATION.md
 * shows the Rust `extern "C"` block and the `proof_gen` call sites a reference
 * maintainer would add.
 *
 * Data at the boundary
 *   - A ring element of R_q = Z_q[X]/(X^64+1) is uint64_t[64]: canonical
 *     residues in [0,q), index = degree, zero padded — exactly the slices the
 *     reference hands to concrete-ntt (src/algebraic.rs:385-394).
 *   - Arrays of ring elements are row-major and contiguous:
 *       witness S        [n][r][64]   (ndarray Array2<Rq> (N,R), src/proofgen.rs:461)
 *       t_i_all          [r][kappa][64]                     (src/proofgen.rs:35)
 *       g_mat, h_mat     [r][r][64]                         (src/proofgen.rs:59,320)
 *       Pi_i             int8 [r][256][n*64] in {-1,0,1}    (src/verification.rs:553-566)
 *   - Every `const uint64_t*` / `uint64_t*` / `int8_t*` argument may be a host
 *     pointer or a device pointer; the library detects which and stages host
 *     buffers through the context's stream.
 *   - `lab_vec` is a device-resident array of ring elements kept in the
 *     transform ("hat") domain: u32 residues [nlimbs][count][64], one plane per
 *     RNS prime, each polynomial already through the forward negacyclic NTT.
 *     Matrices that are reused (the Ajtai matrix A, the CRS blocks B, C, D,
 *     the witness) live in this form so products never re-transform them —
 *     the reference re-transforms both operands on every product
 *     (src/algebraic.rs:385-399).
 *
 * Moduli
 *   q is a parameter of the context (a process-wide lazy static in the
 *   reference, src/constants.rs:193-198).  Any odd q in [3, 2^40) works: the
 *   exact integer negacyclic result is recovered by CRT over 1..4 NTT-friendly
 *   30-bit primes and only then reduced mod q — the same thing
 *   concrete-ntt's native64::Plan32 does for the reference, without its mod
 *   2^64 wrap (src/algebraic.rs:417-420).  If q itself is a prime with
 *   q == 1 (mod 128) and q < 2^30 the context uses q as its single limb.
 *
 * Errors
 *   Every function returns LAB_OK (0) or a negative lab_status; nothing
 *   unwinds or aborts across the boundary.  The reference panics instead
 *   (src/util.rs:497-502, src/proofgen.rs:175-177, src/verification.rs:550);
 *   the host wrappers turn the codes back into those panics/exceptions.
 *   lab_last_error() returns a human-readable message for the last failure.
 *
 * Threading: a context owns one HIP stream and scratch memory and is not
 * re-entrant; distinct contexts are independent (the reference uses process
 * globals Q, PLAN, NTT_ENABLED, MOD_SUSPENSION — src/constants.rs:193-201).
 */
#ifndef LABRADOR_HIP_H
#define LABRADOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAB_D 64 /* src/constants.rs:15 */

typedef enum {
  LAB_OK = 0,
  LAB_ERR_ARG = -1,           /* null pointer, bad size, unsupported modulus           */
  LAB_ERR_LEN_MISMATCH = -2,  /* "inner product not defined on vectors of unequal length", src/util.rs:497-502 */
  LAB_ERR_RANGE = -3,         /* reduction length exceeds the context's exactness bound */
  LAB_ERR_HIP = -4,           /* a HIP runtime call failed                             */
  LAB_ERR_NO_DEVICE = -5,     /* no usable GPU: the product has no CPU fallback        */
  LAB_ERR_JL_REJECTED = -6,   /* projection norm too large, src/verification.rs:568-579 / src/proofgen.rs:170-181 */
  LAB_ERR_BPP_CHECK = -7,     /* verify_b_prime_prime failed, src/verification.rs:532-551 */
  LAB_ERR_DECOMPOSE = -8      /* base < 2 (the reference loops forever, src/util.rs:410-417) or exp out of range */
} lab_status;

typedef struct lab_ctx lab_ctx;
typedef struct lab_vec lab_vec;

typedef struct {
  uint64_t q;
  int nlimbs;             /* RNS primes in use (1 when q itself is the NTT prime) */
  uint32_t primes[4];
  uint64_t max_len;       /* longest inner product of canonical operands that is still exact */
  int device;
} lab_info;

/* ---- context ----------------------------------------------------------
 * Replaces the lazy statics Q and PLAN (src/constants.rs:193-198).
 * `max_len` = longest reduction (inner-product length) the caller will ask
 * for; it fixes the number of RNS limbs.  Pass 0 for the default (65536). */
int lab_ctx_create(uint64_t q, int device, uint64_t max_len, lab_ctx **out);
void lab_ctx_destroy(lab_ctx *ctx);
int lab_ctx_info(const lab_ctx *ctx, lab_info *out);
const char *lab_last_error(const lab_ctx *ctx);
/* Use the caller's HIP stream (e.g. torch's current stream) instead of the context's own. */
int lab_ctx_set_stream(lab_ctx *ctx, void *hip_stream);
int lab_ctx_synchronize(lab_ctx *ctx);

/* ---- the reference's operator seam -------------------------------------
 * (*PLAN).negacyclic_polymul(&mut prod, &lhs, &rhs) + transform_slice_u64_to_zq
 * (src/algebraic.rs:396-397; raw form benches/ntt_speed.rs:17-19), batched:
 * prod[b] = lhs[b] * rhs[b] in R_q, already reduced mod q.  [batch][64] each. */
int lab_negacyclic_polymul_u64(lab_ctx *ctx, uint64_t *prod, const uint64_t *lhs, const uint64_t *rhs, size_t batch);

/* Batched forward / inverse 64-point negacyclic NTT in place on device residues
 * u32 [nlimbs][batch][64] (limb l reduced mod primes[l]); the transforms inside
 * Plan32::fwd / ::inv that negacyclic_polymul is built from.  Forward output is
 * in bit-reversed order; inverse consumes that order.  BASELINE config 2. */
int lab_ntt_forward(lab_ctx *ctx, uint32_t *dev_residues, size_t batch);
int lab_ntt_inverse(lab_ctx *ctx, uint32_t *dev_residues, size_t batch);

/* ---- device-resident transform-domain vectors --------------------------- */
int lab_vec_create(lab_ctx *ctx, size_t count, lab_vec **out);
void lab_vec_destroy(lab_vec *v);
size_t lab_vec_count(const lab_vec *v);
/* coefficient form -> hat form (replaces the per-product pad + forward transform, src/algebraic.rs:385-396) */
int lab_vec_load_u64(lab_ctx *ctx, lab_vec *dst, size_t dst_off, const uint64_t *coeffs, size_t count);
/* hat form -> coefficient form, reduced mod q (src/algebraic.rs:397, 417-420 without the wrap) */
int lab_vec_store_u64(lab_ctx *ctx, const lab_vec *src, size_t src_off, size_t count, uint64_t *coeffs);
/* Synthetic CRS / witness data: element k, coefficient j = splitmix64 stream
 * (seed, (first_elem+k)*64+j) scaled into [0,bound); identical to the oracle's
 * orc_synth_fill.  Stands in for CRS::fetch_* (src/structs.rs:55-171), whose
 * ChaCha20 expansion is out of scope this round. */
int lab_vec_fill_synthetic(lab_ctx *ctx, lab_vec *dst, size_t dst_off, size_t count, uint64_t seed,
                           uint64_t first_elem, uint64_t bound);

/* ---- src/util.rs vector helpers on hat vectors ---------------------------- */

/* polynomial_vec_inner_product (src/util.rs:496-509): out[out_off] = sum_i v1[off1+i]*v2[off2+i] */
int lab_inner_product(lab_ctx *ctx, lab_vec *out, size_t out_off, const lab_vec *v1, size_t off1,
                      const lab_vec *v2, size_t off2, size_t len);

/* Sum of `nmat` matrix-vector products, each row an inner product:
 *   out[out_off + rho] = sum_{m<nmat} < M[m_off + (m*rows + rho)*cols ..+cols], V[v_off + m*cols ..+cols] >
 * nmat = 1: the rows of Prover::proof_gen's inner Ajtai commitment for one s_i
 * (src/proofgen.rs:41-49) and verifier check 15 (src/verification.rs:274-279);
 * nmat = r*t1: the outer commitment sum_{i,k} B_ik t_i^(k)
 * (src/proofgen.rs:101-132, the rayon map + sequential add_poly_vec fold). */
int lab_matvec_sum(lab_ctx *ctx, lab_vec *out, size_t out_off, const lab_vec *M, size_t m_off, size_t nmat,
                   size_t rows, size_t cols, const lab_vec *V, size_t v_off);

#ifdef __cplusplus
}
#endif
#endif /* LABRADOR_HIP_H */
