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
 *   q == 1 (mod 128) and q < 2^32 the context uses q as its single limb
 *   (4 bytes per coefficient, no CRT): below 2^30 on the lazy kernels, in
 *   [2^30, 2^32) on the strict-32 kernels (lab_info.nlimbs == 1 either way;
 *   LAB_CTX_RNS_ONLY keeps the generic RNS path).  The main basis of such a
 *   strict-32 context has no matrix-core form (lab_mat8_pack refuses it);
 *   the 23-bit digit basis (lab_ctx_enable_digit_basis_mfma) serves it.
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
typedef struct lab_mat8 lab_mat8;   /* matrices packed for the matrix cores (lab_mat8_pack) */

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
/* flags: LAB_CTX_RNS_ONLY — never use q as its own NTT prime (the generic exact path even for an NTT-friendly q), e.g. to
 * time both paths of SURVEY section 7 at one modulus.  lab_ctx_create == flags 0: a prime q == 1 (mod 128) below 2^32 runs
 * single-limb (4 bytes per coefficient, no CRT; strict 32-bit arithmetic when q >= 2^30), every other q on RNS limbs. */
#define LAB_CTX_RNS_ONLY 1
int lab_ctx_create_ex(uint64_t q, int device, uint64_t max_len, int flags, lab_ctx **out);
void lab_ctx_destroy(lab_ctx *ctx);
int lab_ctx_info(const lab_ctx *ctx, lab_info *out);
const char *lab_last_error(const lab_ctx *ctx);
/* Use the caller's HIP stream (e.g. torch's current stream) instead of the context's own. */
int lab_ctx_set_stream(lab_ctx *ctx, void *hip_stream);
int lab_ctx_synchronize(lab_ctx *ctx);

/* Measurement hook (no reference counterpart): between begin and end every launch of the streaming
 * mat-vec kernel (the prover's dominant, HBM-bound kernel) is bracketed by HIP events on the context's
 * stream.  end() synchronises and returns the summed kernel time, the launch count and the algorithmic
 * bytes those launches moved (matrix + vector + output, all limbs). */
int lab_ctx_profile_begin(lab_ctx *ctx);
int lab_ctx_profile_end(lab_ctx *ctx, double *kernel_ms, uint64_t *launches, double *algorithmic_bytes);
/* The same hook for the STREAMED outer commitment (lab_prove_input.crs_seed / lab_crs_matvec_mfma), whose time is not one
 * kernel's: between begin and end a bounded uniform sample (at most 1024 launches per phase) of the launches of phase
 * 0 = CRS expansion (k_crs_expand; units: coefficients), 1 = coefficients -> matrix-core tiles (units: coefficients),
 * 2 = the int8 GEMMs over a slab (units: algorithmic bytes), 3 = decomposition digits -> tiles (units: digits) is bracketed by
 * HIP events on the context's stream.  Returns the sampled time / launches / units and the totals over every launch of the
 * phase since begin (call before lab_ctx_profile_end or after: the sample is kept until the next begin). */
int lab_ctx_profile_phase(lab_ctx *ctx, int phase, double *sampled_ms, uint64_t *sampled_launches, double *sampled_units,
                          uint64_t *launches, double *units);
/* name (template arguments included) of the kernel the most recent launch timed by the hook was dispatched to */
const char *lab_ctx_profile_kernel(const lab_ctx *ctx);
/* Diagnostic variants of that kernel (tools/gemm_ablation.py; the matrix-core outer commitment at 64 sets only, otherwise
 * ignored).  mode 0: the product kernel.  1: the same loop, and every block writes the shader-clock (s_memtime) and
 * constant-clock (s_memrealtime) ticks its K loop took to ticks[2*block], ticks[2*block + 1] (device memory, `words`
 * uint64 long; NULL: nothing is written).  2: the loop without its global loads and LDS-DMA (multiply side alone),
 * 3: without its matrix instructions and LDS reads (load side alone) — the commitments computed in modes 2 and 3 are
 * MEANINGLESS; these modes exist to time the two sides of the loop separately.  4: like 1, for the launches that multiply
 * one matrix with many jobs (the Ajtai product of a batch) instead.  Stays in force until set again. */
int lab_ctx_gemm_diag(lab_ctx *ctx, int mode, uint64_t *ticks, size_t words);
/* how the most recent lab_prove_core[_batch] call ran: 1 = step 1 in the digit basis (lab_prove_input.A8d), 2 = that was tried,
 * a witness exceeded beta_bound and the batch was redone in the main basis; 0 = main basis from the start.  A call with more than
 * 64 proofs runs in sub-batches of 64 and reports the OR of theirs (3: one was redone, another was not), plus 4 when sub-batches
 * shared ONE pass over B in step 3 (B streamed from the seed, or resident as B8: their digit tiles were the jobs of one launch). */
int lab_ctx_prove_flags(const lab_ctx *ctx);

/* Digit basis (no reference counterpart; a storage/bandwidth optimisation).  The outer commitments multiply CRS
 * matrices by decomposition DIGITS (< b_1, b_2 ~ 2^11, src/util.rs:389-442), so their exact integer results need
 * far fewer CRT bits than products of two canonical elements.  After this call the context also owns a second,
 * smaller RNS basis — a prefix of the main limb primes plus, when it saves space, one NTT prime below 2^16 —
 * that is exact for up to `max_terms` products of a canonical element with a value < digit_bound.
 * lab_vec_create_digit() makes vectors in that basis; with compact = 1 the small limb is stored as u16
 * (e.g. 4+4+2 = 10 bytes per coefficient instead of 12 at q = 2^32+15).  Digit-basis vectors are accepted by
 * lab_vec_load_u64 / _store_u64 / _fill_synthetic, lab_crs_expand, lab_pointwise, lab_matvec_sum and
 * lab_axpy_commit (compact only as their streamed matrix M / C; the caller guarantees V / G hold digits below
 * digit_bound) and by lab_prove_core / lab_verify_core for B, C, D.  Fails with LAB_ERR_RANGE when no smaller
 * basis exists (e.g. q is its own NTT prime). */
int lab_ctx_enable_digit_basis(lab_ctx *ctx, uint64_t max_terms, uint64_t digit_bound);
/* The same, with a basis made for the matrix-core path (lab_mat8_pack): primes below 2^23, whose residues are three
 * signed bytes (9 instead of 16 byte products per modular product, 3 bytes of packed matrix per limb and
 * coefficient).  No u16 limb; works for every q, also a directly NTT-friendly one.  One of the two per context. */
int lab_ctx_enable_digit_basis_mfma(lab_ctx *ctx, uint64_t max_terms, uint64_t digit_bound);
/* nlimbs/primes of the digit basis; max_len = max_terms; device = 1 when the last limb is a u16-storable prime */
int lab_ctx_digit_info(const lab_ctx *ctx, lab_info *out);
int lab_vec_create_digit(lab_ctx *ctx, size_t count, int compact, lab_vec **out);
/* bit 0: digit basis, bit 1: compact */
int lab_vec_basis(const lab_vec *v);

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
 * orc_synth_fill.  A cheap stand-in for CRS::fetch_* (src/structs.rs:55-171) where the reference's own
 * ChaCha20 expansion (lab_crs_expand below, lab_prove_input.crs_seed) is not what is being tested or timed. */
int lab_vec_fill_synthetic(lab_ctx *ctx, lab_vec *dst, size_t dst_off, size_t count, uint64_t seed,
                           uint64_t first_elem, uint64_t bound);

/* CRS::fetch_next_n / random_oracle_gen on the device (src/structs.rs:35-45, 147-171):
 * `count` polynomials whose 64*count coefficients are ChaCha20Rng::from_seed(seed).gen_range(0..q) for the
 * consecutive 256-bit big-endian seeds base_seed + offset, +1 per coefficient (offset = offset_hi*2^64 + offset_lo,
 * in coefficient units — the reference's fetch_* offsets are listed in INTEGRATION.md).  lab_crs_expand stores
 * them in the transform domain, lab_crs_expand_u64 as coefficients (host or device pointer). */
int lab_crs_expand(lab_ctx *ctx, lab_vec *dst, size_t dst_off, size_t count, const uint8_t base_seed[32],
                   uint64_t offset_lo, uint64_t offset_hi);
int lab_crs_expand_u64(lab_ctx *ctx, uint64_t *coeffs, size_t count, const uint8_t base_seed[32],
                       uint64_t offset_lo, uint64_t offset_hi);

/* The outer commitment  out_s[rho] = sum_{m<nmat} < M_m[rho, :], V_s[m] >  (src/proofgen.rs:101-132) with the matrices M_m
 * streamed from the seed instead of resident: M_m[rho][c] is the polynomial fetch_next_n returns for the 64 seeds
 * base_seed + off_m + (rho*cols + c)*64 .. (src/structs.rs:147-171), off_m = mat_offsets[2m] + 2^64 * mat_offsets[2m+1]
 * (coefficient units; for B_ik: CRS::fetch_B_ik_row's offset of row 0, src/structs.rs:78-82).  Slabs of 128 rows are
 * expanded, transformed, packed and multiplied on the matrix cores (ns in {8,16,32,64} vector sets laid out as for
 * lab_matvec_sum_mfma) and discarded: only a fixed working set exists in device memory.  Bit-identical to
 * lab_crs_expand + lab_mat8_pack + lab_matvec_sum_mfma.  SURVEY §8(f) row f1. */
int lab_crs_matvec_mfma(lab_ctx *ctx, lab_vec *out, size_t out_off, const uint8_t base_seed[32], const uint64_t *mat_offsets,
                        size_t nmat, size_t rows, size_t cols, const lab_vec *V, size_t v_off, int ns);

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

/* The same sum for nb (1, 2, 4 or 8) independent vector sets that share the matrices — e.g. the outer commitments
 * of nb independent proofs under one common reference string, which is then streamed from HBM once instead of nb
 * times:  out[out_off + b*rows + rho] = sum_m < M[m][rho,:], V[v_off + (b*nmat + m)*cols ..+cols] >. */
int lab_matvec_sum_batched(lab_ctx *ctx, lab_vec *out, size_t out_off, const lab_vec *M, size_t m_off, size_t nmat,
                           size_t rows, size_t cols, const lab_vec *V, size_t v_off, int nb);

/* The inner Ajtai commitments for all r witness columns at once
 * (src/proofgen.rs:41-49; also verifier check 15, src/verification.rs:274-279):
 *   out[out_off + i*rows + rho] = < A[a_off + rho*inner ..+inner], S[:, i] >,  S = [inner][ncols] row-major
 * i.e. t_i_all[i][rho] with A = [kappa][n], S = witness [n][r]. */
int lab_ajtai_commit(lab_ctx *ctx, lab_vec *out, size_t out_off, const lab_vec *A, size_t a_off, size_t rows,
                     size_t inner, const lab_vec *S, size_t s_off, size_t ncols);

/* Gram-type products over witness-shaped arrays P, S = [n][r] row-major:
 *   out[out_off + i*r + k] = sum_x P[p_off + x*r + i] * S[s_off + x*r + k]
 * g_mat = S^T S (src/proofgen.rs:59-70); the two inner products of each h_ij (src/proofgen.rs:334-335). */
int lab_gram(lab_ctx *ctx, lab_vec *out, size_t out_off, const lab_vec *P, size_t p_off, const lab_vec *S,
             size_t s_off, size_t n, size_t r);

/* ---- the batched outer commitment on the matrix cores (no reference counterpart; exact) ----
 * For 8 or more proofs per batch lab_matvec_sum_batched is bound by the 64-bit multiply-add issue rate, not by HBM.
 * lab_mat8_pack re-stores the matrices M (as for lab_matvec_sum: nmat matrices of rows x cols elements, plain or
 * compact, either basis) once as signed base-256 digit planes in MFMA operand-tile order (same bytes per residue);
 * lab_matvec_sum_mfma then computes, for ns in {8,16,32,64} vector sets V_s (set s at v_off + s*nmat*cols),
 *   out[out_off + s*rows + rho] = sum_{m,c} M[m][rho][c] * V_s[m*cols + c]
 * as exact int8 GEMMs (v_mfma_i32_32x32x32_i8, int32 accumulation: nmat*cols must stay below 2^17) followed by a
 * recombination mod each limb prime.  Bit-identical to lab_matvec_sum. */
int lab_mat8_pack(lab_ctx *ctx, const lab_vec *M, size_t m_off, size_t nmat, size_t rows, size_t cols, lab_mat8 **out);
int lab_mat8_destroy(lab_mat8 *m);
size_t lab_mat8_bytes(const lab_mat8 *m);
int lab_matvec_sum_mfma(lab_ctx *ctx, lab_vec *out, size_t out_off, const lab_mat8 *M8, const lab_vec *V, size_t v_off, int ns);

/* sum_m poly_by_poly_vec(G[m], C[m]) — the C and D parts of the outer commitments
 * (src/proofgen.rs:137-148 `rhs`, 364-378 `u_2`):
 *   out[out_off + rho] = sum_{m<nterms} G[g_off + m] * C[c_off + m*len + rho] */
int lab_axpy_commit(lab_ctx *ctx, lab_vec *out, size_t out_off, const lab_vec *G, size_t g_off, const lab_vec *C,
                    size_t c_off, size_t nterms, size_t len);
/* The same for nb in {1,2,4,8} independent digit vectors sharing C (C is streamed once for all of them — the C / D
 * commitments of nb proofs under one common reference string): set b reads G[g_off + b*nterms + m] and writes
 * out[out_off + b*len + rho]. */
int lab_axpy_commit_batched(lab_ctx *ctx, lab_vec *out, size_t out_off, const lab_vec *G, size_t g_off, const lab_vec *C,
                            size_t c_off, size_t nterms, size_t len, int nb);

/* z = sum_i c_i * s_i (src/proofgen.rs:380-399): out[out_off + x] = sum_{i<r} S[s_off + x*r + i] * c[c_off + i] */
int lab_amortize_z(lab_ctx *ctx, lab_vec *out, size_t out_off, const lab_vec *S, size_t s_off, size_t n, size_t r,
                   const lab_vec *c, size_t c_off);

/* Element-wise ring ops on hat vectors: Rq Mul / Add / Sub (src/algebraic.rs:441-531), add_poly_vec (src/util.rs:298-309) */
typedef enum { LAB_OP_MUL = 0, LAB_OP_ADD = 1, LAB_OP_SUB = 2 } lab_op;
int lab_pointwise(lab_ctx *ctx, lab_op op, lab_vec *out, size_t out_off, const lab_vec *a, size_t a_off,
                  const lab_vec *b, size_t b_off, size_t count);
/* poly_by_poly_vec (src/util.rs:289-296): out[k] = p * v[k] */
int lab_poly_by_poly_vec(lab_ctx *ctx, lab_vec *out, size_t out_off, const lab_vec *p, size_t p_off,
                         const lab_vec *v, size_t v_off, size_t len);

/* ---- coefficient-domain helpers (u64 residues mod q, host or device pointers) ---- */

/* decompose_polynomial_vec, bug-compatible with src/util.rs:360-442 (non-negative digits, sign of the centred
 * representative lost, digits beyond `exp` dropped).  coeffs [count][64] -> out [exp][count][64]. */
int lab_decompose(lab_ctx *ctx, uint64_t *out, const uint64_t *coeffs, size_t count, int64_t base, int exp);
/* poly_norm (src/util.rs:195-202): exact sum of squares of the raw residues of each polynomial as a 128-bit
 * integer norm = hi[k]*2^64 + lo[k]; the caller casts to f64 and folds in index order like
 * vec_poly_norm_squared (src/util.rs:188-192). */
int lab_norm_sq(lab_ctx *ctx, const uint64_t *coeffs, size_t count, uint64_t *lo, uint64_t *hi);
/* scale_poly_vec (src/util.rs:167-180): out[k] = s * v[k] mod q */
int lab_scale(lab_ctx *ctx, uint64_t *out, const uint64_t *coeffs, size_t count, uint64_t s);
/* sigma_inv_vec (src/util.rs:107-137) */
int lab_sigma_inv(lab_ctx *ctx, uint64_t *out, const uint64_t *coeffs, size_t count);
/* Prover::jl_project's integer mat-vecs (src/proofgen.rs:429-456, src/util.rs:511-526):
 *   proj[m] = sum_{i<r} sum_col Pi[i][m][col] * S[col/64][i][col%64]   over the integers (no reduction)
 * Pi: int8 [r][256][n*64], S: witness coefficients [n][r][64], proj: int64[256] (|proj| < r*n*64*q < 2^63). */
int lab_jl_project(lab_ctx *ctx, int64_t *proj, const int8_t *Pi, const uint64_t *S, size_t n, size_t r);
/* sample_jl_projection's matrices (src/verification.rs:553-566) from int8 {-1,0,1}, `mats` row-major matrices of 256 x cols
 * (cols a multiple of 16), to the tiled 2-bit form of lab_prove_input.Pi2 (mats * 256 * cols / 4 bytes).  Host or device
 * pointers. */
int lab_pi_pack(lab_ctx *ctx, uint8_t *packed, const int8_t *Pi, size_t mats, size_t cols);

/* ---- Prover::proof_gen as one deterministic call (src/proofgen.rs:30-427) ----
 * The verifier's interactive randomness (src/verification.rs:441-513, 553-566) is supplied as a challenge tape and
 * the CRS rows (regenerated from a ChaCha20 seed on every fetch in the reference, src/structs.rs:55-171) as
 * resident transform-domain matrices.  kappa, kappa1, kappa2 are explicit (the reference hard-wires all three to
 * n*64, src/constants.rs:237-239).  Pair index of (i<=j): i*r - i(i-1)/2 + (j-i) (src/structs.rs:101-106). */
typedef struct {
  size_t struct_size;      /* = sizeof(lab_prove_input): any other value is refused with LAB_ERR_ARG, so a binding built
                            * against an older (shorter) layout is caught instead of being read past its end */
  size_t n, r, kappa, kappa1, kappa2;
  int64_t b1, b2;          /* RuntimeConstants::B_1, B_2 (src/constants.rs:245,247) */
  int t1, t2;              /* RuntimeConstants::T_1, T_2 (src/constants.rs:244,246) */
  int64_t beta_bound;      /* RuntimeConstants::BETA_BOUND (src/constants.rs:241) */
  const uint64_t *S;       /* witness            [n][r][64]  (src/proofgen.rs:16)        */
  const uint64_t *phi;     /* st.phi_k[0]        [n][r][64]  (src/structs.rs:278)        */
  const uint64_t *a;       /* st.a_prime_k[0]    [r][r][64]  (src/structs.rs:284)        */
  uint64_t b_prime;        /* st.b_prime_k[0]                (src/structs.rs:285,373)    */
  const lab_vec *A;  size_t a_off;   /* [kappa][n]              fetch_A_row    (src/structs.rs:55-72)   */
  const lab_vec *B;  size_t b_off;   /* [r*t1][kappa1][kappa]   fetch_B_ik_row (src/structs.rs:74-88)   */
  const lab_vec *C;  size_t c_off;   /* [pairs*t2][kappa2]      fetch_C_ijk    (src/structs.rs:90-114)  */
  const lab_vec *Dm; size_t d_off;   /* [pairs*t1][kappa2]      fetch_D_ijk    (src/structs.rs:116-144) */
  const int8_t *Pi;        /* [r][256][n*64] in {-1,0,1}  sample_jl_projection (src/verification.rs:553-566) */
  uint64_t psi;            /* generate_psi, L = 1          (src/verification.rs:491-501) */
  const uint64_t *omega;   /* [256] generate_omega         (src/verification.rs:503-513) */
  const uint64_t *alpha;   /* [64]  fetch_alpha, K = 1     (src/verification.rs:441-447) */
  const uint64_t *beta;    /* [64]  fetch_beta             (src/verification.rs:449-456) */
  const uint64_t *c;       /* [r][64] fetch_challenge      (src/verification.rs:460-489) */
  const struct lab_mat8 *B8; /* optional: the B matrices packed for the matrix cores (lab_mat8_pack(B, b_off, r*t1, kappa1,
                              * kappa)); when set, step 3 of lab_prove_core[_batch] runs as int8 GEMMs over up to 64 proofs
                              * per pass and B itself may be NULL (lab_verify_core still needs B) */
  const struct lab_mat8 *A8; /* optional: lab_mat8_pack(A, a_off, 1, kappa, n): step 1 (t_i = A s_i) as int8 GEMMs */
  const struct lab_mat8 *C8; /* optional: lab_mat8_pack(C, c_off, pairs*t2, kappa2, 1): step 4 (sum g^(k) C) as int8 GEMMs */
  const struct lab_mat8 *D8; /* optional: lab_mat8_pack(Dm, d_off, pairs*t1, kappa2, 1): step 9 (sum h^(k) D) likewise; with
                              * C8 and D8 set the prover does not read C and Dm, which may then be NULL (lab_verify_core needs them) */
  const struct lab_mat8 *A8d; /* optional: A packed from a copy in the digit basis of B, C, D (same kappa x n elements).  Step 1 of
                               * a batch then runs in that basis whenever sqrt(64 n) (q-1) beta_bound is inside its CRT range
                               * (|A s| <= |A|_2 |s|_2): with the matrix-core basis 9 instead of 16 byte products per modular
                               * product.  The pass that transforms S also sums its squared norm; a witness above beta_bound
                               * (never produced by generate_witness, src/proofgen.rs:460-518) makes the call redo the batch
                               * in the main basis, so results never depend on this field */
  const uint8_t *Pi2;      /* optional, instead of Pi (which may then be NULL): the projection matrices at 2 bits per entry in the
                            * tiled order lab_pi_pack produces (r * 256 * n*16 bytes): dword ((i*4 + s) * (n*4) + ct) * 64 + lane
                            * holds rows 64s + 16*(lane >> 4) + 0..15 of column 16ct + (lane & 15) of Pi_i, row 4j + k in the
                            * field at bit 8k + 2j, as entry + 1 in {0, 1, 2} — one dword per lane is one matrix-core operand.
                            * A quarter of the bytes of the int8 form (the reference lifts Pi to 16-byte Zq,
                            * src/proofgen.rs:445-453) and W = sum_m omega_m sigma_inv(pi^(m)) on the matrix cores */
  const uint8_t *crs_seed; /* optional, 32 bytes: CRS::base_seed (src/structs.rs:29, 173-189).  With B == NULL and B8 == NULL the
                            * B_ik matrices are STREAMED from this seed (the rows fetch_B_ik_row regenerates on every fetch,
                            * src/structs.rs:74-88): step 3 expands a slab of rows on the device, multiplies it with every proof
                            * of the batch on the matrix cores and discards it (lab_crs_matvec_mfma) — B is never resident, so
                            * the reference's own ranks kappa1 = kappa = n*64 (1 TiB per B_ik) are reachable.  A, C, D stay
                            * resident vectors (expand them with lab_crs_expand).  lab_verify_core does not take this mode */
} lab_prove_input;

/* The computed fields of `Transcript` (src/structs.rs:192-209); pi_i_all, psi, omega, alpha, beta, c are the tape.
 * All buffers are caller-allocated (host or device); any pointer may be NULL to skip that copy-out. */
typedef struct {
  uint64_t *u_1;            /* [kappa1][64] */
  int64_t *projection_int;  /* [256] integers before Zq::lift (src/proofgen.rs:163) */
  uint64_t *projection;     /* [256] (src/proofgen.rs:186) */
  uint64_t *b_prime_prime;  /* [64]  */
  uint64_t *u_2;            /* [kappa2][64] */
  uint64_t *z;              /* [n][64] */
  uint64_t *t_i_all;        /* [r][kappa][64] */
  uint64_t *g_mat;          /* [r][r][64] */
  uint64_t *h_mat;          /* [r][r][64] */
} lab_transcript;

/* Returns LAB_OK, LAB_ERR_JL_REJECTED when Verifier::valid_projection would reject this Pi (the reference then
 * resamples up to 5 times, src/proofgen.rs:169-181 — the caller does that with a fresh tape), or
 * LAB_ERR_BPP_CHECK when verify_b_prime_prime's assertion would fire (src/verification.rs:550). */
int lab_prove_core(lab_ctx *ctx, const lab_prove_input *in, lab_transcript *out);

/* `nproofs` independent proofs under ONE common reference string (same A, B, C, D handles, dimensions and
 * constants; each with its own witness, state and challenge tape).  The per-proof kernels take all proofs of the call as
 * jobs of one launch; the outer commitment sum_{i,k} B_ik t_i^(k) — whose cost is streaming (B8 / B) or regenerating
 * (crs_seed) the B matrices — is issued once per group of proofs: up to 64 per pass over B on the matrix cores (exact int8
 * GEMMs), up to 8 with the multi-set multiply-add kernel when B is only present as residues.  More than 64 proofs run as
 * sub-batches of 64; with B streamed from crs_seed (up to 256 proofs) the sub-batches SHARE one pass over the regenerated rows:
 * each runs steps 1-2, then one streamed step 3 serves all of them, then each finishes — the expansion, ~75 % of such a step,
 * is paid once per call.  With B resident as B8 (up to 1024 proofs) the full sub-batches share the pass over B the same way:
 * their digit tiles are the jobs of ONE launch per limb, B comes from HBM once and from L2 for the other jobs (+8 % proofs/s at
 * 256 proofs per call; lab_ctx_prove_flags() & 4).  Every transcript is bit-identical to what lab_prove_core returns for that
 * proof alone.
 * status[j] (if status != NULL) receives LAB_OK / LAB_ERR_JL_REJECTED / LAB_ERR_BPP_CHECK for proof j and the call
 * returns LAB_OK; with status == NULL the first failing proof's code is returned. */
int lab_prove_core_batch(lab_ctx *ctx, size_t nproofs, const lab_prove_input *const *in, lab_transcript *const *out,
                         int *status);

/* ---- Verifier::verify (src/verification.rs:25-438) on the same kernels — SURVEY §8(f) row f3 ----
 * `in` carries the state, the tape stored in the transcript (pi_i_all as int8, psi, omega, alpha, beta, c) and the
 * resident CRS exactly as for lab_prove_core (S is not read and may be NULL); `proof` the computed transcript fields
 * (u_1, projection, b_prime_prime, u_2, z, t_i_all, g_mat, h_mat; projection_int is not read).
 * b_poly = st.b_k[0] ([64]); bconst = RuntimeConstants::B; beta_prime = RuntimeConstants::BETA_PRIME.
 * On LAB_OK, *failed_line is 0 when every check passes, otherwise the number (8..20) of the first failing
 * protocol line in the reference's order; *line14_sum receives the f64 norm sum of line 14, folded in the
 * reference's order (src/verification.rs:231-255). */
int lab_verify_core(lab_ctx *ctx, const lab_prove_input *in, const lab_transcript *proof, const uint64_t *b_poly,
                    int64_t bconst, double beta_prime, int *failed_line, double *line14_sum);

#ifdef __cplusplus
}
#endif
#endif /* LABRADOR_HIP_H */
