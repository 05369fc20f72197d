// kernels.h — host-callable launchers for the gfx950 kernels (implemented in
// ntt_kernels.hip / linalg_kernels.hip / proof_kernels.hip).  Internal to the
// shared library; the public surface is include/labrador_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zq_device.h"

namespace lab {

// A view of `count` ring elements inside a hat-domain allocation:
// limb l, element k, slot j at base[l * plane_stride + (k) * 64 + j].
struct HatView {
  uint32_t* base;        // already offset to the first element of limb 0
  size_t plane_stride;   // u32 words between consecutive limb planes
};
struct ConstHatView {
  const uint32_t* base;
  size_t plane_stride;
  ConstHatView() : base(nullptr), plane_stride(0) {}
  ConstHatView(const uint32_t* b, size_t s) : base(b), plane_stride(s) {}
  ConstHatView(const HatView& v) : base(v.base), plane_stride(v.plane_stride) {}
};

// Per-job operands of a launch that serves several independent proofs ("jobs") at once: coefficient-domain arrays are
// separate allocations (often the caller's), so their device pointers travel by value in the kernel arguments;
// transform-domain operands live in one arena per basis and are addressed with a job stride (words inside a plane).
constexpr int MAX_JOBS = 64;
struct JobPtrs { void* p[MAX_JOBS]; };
struct JobU64 { uint64_t v[MAX_JOBS]; };

// CRT constants for hat -> coefficient conversion (device resident).
struct CrtTable {
  uint64_t q;
  int nlimbs;
  uint32_t p[MAX_LIMBS];
  // Garner: inv[i][j] = p_j^-1 mod p_i (j < i), with Shoup companions
  uint32_t ginv[MAX_LIMBS][MAX_LIMBS][2];
  // prefix[i] = (p_0*...*p_{i-1}) mod q ; mmodq = (p_0*...*p_{L-1}) mod q
  uint64_t prefix_mod_q[MAX_LIMBS];
  uint64_t m_mod_q;
  // half-modulus test in mixed radix: M/2 digits (floor), for the sign decision
  uint32_t half_digits[MAX_LIMBS];
  // reciprocals for the division-free reductions (mod_u64)
  double inv_q;
  double inv_p[MAX_LIMBS];
  // sum_a (p_a - 1) * (q - 1) + max p < 2^63: the mixed-radix digits times the prefixes can be summed in 64 bits before the one mod q
  uint32_t lazy_sum;
  // three limbs of one size (p_b < 2 p_a, all below 2^30), lazy_sum and q >= 2^24: the straight-line form crt3_fast (ntt_kernels.hip)
  uint32_t fast3;
};

// ---- ntt_kernels.hip
// njobs > 1: job b works on src + b*src_js / dst + b*dst_js (word strides inside a plane)
// single-limb strict-32 contexts (LimbTable.wide): the launchers pick their instantiation from the device table pointer
void limb_tables_register(const LimbTable* dev_tabs, bool wide);
bool limb_tables_wide(const LimbTable* dev_tabs);
// The sum-of-products form every transform-domain reduction takes on the single-limb strict-32 path (wide_kernels.hip):
// out[o][j] = sum_{k<K} A[abase(o) + k*ak + j] * B[bbase(o) + k*bk + j] mod p for the five-level output index o = (i0..i4), i4
// fastest; the bases are sum_l i_l * stride_l (words).  chunk_level >= 0: index n[chunk_level] walks chunks of K terms of a
// Ktotal-term sum (its A/B strides advance by one chunk; the last chunk is short).
struct WideDot {
  uint32_t* out;
  const uint32_t* A;
  const uint32_t* B;
  unsigned n[5];
  size_t os[5], as[5], bs[5];
  unsigned K, Ktotal;
  int chunk_level;
  size_t ak, bk;
  const LimbTable* tab;
  bool stream_a;           // A is read once (a CRS matrix): non-temporal loads
};
hipError_t launch_wide_dot(const WideDot& d, hipStream_t s);
hipError_t launch_ntt(bool forward, ConstHatView src, HatView dst, size_t count, int nlimbs, const LimbTable* tabs,
                      int escale, hipStream_t s, size_t njobs = 1, size_t src_js = 0, size_t dst_js = 0);
// inverse transform of a multi-job GEMM's slot-major staging array into dense [count][64] planes (see k_intt_from_stage)
hipError_t launch_intt_from_stage(const uint32_t* stage, size_t rows, size_t rows_pad, size_t ns, size_t ns_valid, size_t nsub,
                                  size_t gemm_jobs, uint32_t* dst, size_t dst_js, size_t set_stride, const LimbTable* tab, hipStream_t s);
hipError_t launch_reduce_u64(const uint64_t* coeffs, HatView dst, size_t count, int nlimbs, const LimbTable* tabs,
                             hipStream_t s);
hipError_t launch_reduce_u64_jobs(const JobPtrs& coeffs, HatView dst, size_t dst_js, size_t count, int nlimbs,
                                  const LimbTable* tabs, size_t njobs, hipStream_t s);
// reduce + forward NTT in one pass (to_hat), njobs polynomials vectors at once
hipError_t launch_to_hat_fused(const JobPtrs& coeffs, HatView dst, size_t dst_js, size_t count, int nlimbs, const LimbTable* tabs,
                               size_t njobs, hipStream_t s, uint64_t in_bound = ~0ull /* coefficients are below this */,
                               unsigned long long* norm_sum = nullptr /* [njobs], += saturating sum of squares */, uint64_t norm_limit = 0,
                               bool src_u16 = false /* coefficients are 16-bit words (digits from launch_decompose_jobs) */,
                               unsigned* above16 = nullptr /* with norm_sum: [njobs], |= 1 when a coefficient of the job is >= 2^16 */);
hipError_t launch_crt_to_u64(ConstHatView src, uint64_t* coeffs, size_t count, const CrtTable* crt, int nlimbs, hipStream_t s);
// decomposition digits written by the CRT pass itself: out16.p[job] as launch_decompose_jobs(.., out_u16 = true) lays them out
struct CrtDigits { JobPtrs out16; uint32_t base; int exp; size_t group; uint64_t q; };
hipError_t launch_crt_to_u64_jobs(ConstHatView src, size_t src_js, const JobPtrs& coeffs, size_t count, const CrtTable* crt,
                                  int nlimbs, size_t njobs, hipStream_t s, const CrtDigits* dg = nullptr);
// inverse transform of every limb + CRT (+ digits) straight from the staging arrays of a multi-job GEMM (see k_stage_to_coeffs)
hipError_t launch_stage_to_coeffs(const uint32_t* stage, size_t stage_ps, size_t rows, size_t rows_pad, size_t ns, size_t ns_valid,
                                  size_t nsub, size_t gemm_jobs, const JobPtrs& coeffs, size_t set_stride, const LimbTable* tabs,
                                  const CrtTable* crt, int nlimbs, hipStream_t s, const CrtDigits* dg = nullptr);
hipError_t launch_synth_fill_u64(uint64_t* coeffs, size_t count, uint64_t seed, uint64_t first_elem, uint64_t bound,
                                 hipStream_t s);

// ---- linalg_kernels.hip
enum PointwiseOp { PW_MUL = 0, PW_ADD = 1, PW_SUB = 2, PW_MUL_KEEP_SCALE = 3 };
// out[k] = a[k] (op) b[k] for k < count; PW_MUL returns plain (e=0) products, PW_MUL_KEEP_SCALE leaves the 2^-32 factor.
// (njobs > 1: job b works on operand + b*<its job stride>, word strides inside a plane — here and below)
hipError_t launch_pointwise(PointwiseOp op, HatView out, ConstHatView a, ConstHatView b, size_t count, int nlimbs,
                            const LimbTable* tabs, hipStream_t s, size_t njobs = 1, size_t out_js = 0, size_t a_js = 0,
                            size_t b_js = 0);
// partial[(m*rows + rho)] = <M[m][rho][:], V[m][:]>   (hat elements, plain scale)
hipError_t launch_matvec_partial(HatView partial, ConstHatView M, ConstHatView V, size_t nmat, size_t rows,
                                 size_t cols, int nlimbs, const LimbTable* tabs, hipStream_t s, size_t njobs = 1, size_t js = 0);
// out[rho] = sum_{m<nterms} partial[m*rows + rho]
hipError_t launch_sum_terms(HatView out, ConstHatView partial, size_t nterms, size_t rows, int nlimbs,
                            const LimbTable* tabs, hipStream_t s, size_t njobs = 1, size_t out_js = 0, size_t partial_js = 0);
const char* matvec_kernel_name();
// single-limb variants whose matrix residues are stored as u16 (compact last limb of the digit basis)
hipError_t launch_matvec_partial_u16(uint32_t* partial_plane, const uint16_t* M, const uint32_t* V_plane, size_t nmat,
                                     size_t rows, size_t cols, const LimbTable* tab, hipStream_t s);
hipError_t launch_axpy_partial_u16(uint32_t* partial_plane, const uint32_t* G_plane, const uint16_t* C, size_t nterms,
                                   size_t len, const LimbTable* tab, hipStream_t s);
// the same streams for nb independent vector sets sharing the matrices (nb in {1,2,4,8})
hipError_t launch_matvec_partial_batched(int nb, HatView partial, size_t partial_bs, ConstHatView M, ConstHatView V,
                                         size_t v_bs, size_t nmat, size_t rows, size_t cols, int nlimbs,
                                         const LimbTable* tabs, hipStream_t s);
hipError_t launch_matvec_partial_batched_u16(int nb, uint32_t* partial_plane, size_t partial_bs, const uint16_t* M,
                                             const uint32_t* V_plane, size_t v_bs, size_t nmat, size_t rows, size_t cols,
                                             const LimbTable* tab, hipStream_t s);
hipError_t launch_axpy_partial_batched(int nb, HatView partial, size_t partial_bs, ConstHatView G, size_t g_bs,
                                       ConstHatView C, size_t nterms, size_t len, int nlimbs, const LimbTable* tabs,
                                       hipStream_t s);
hipError_t launch_axpy_partial_batched_u16(int nb, uint32_t* partial_plane, size_t partial_bs, const uint32_t* G_plane,
                                           size_t g_bs, const uint16_t* C, size_t nterms, size_t len, const LimbTable* tab,
                                           hipStream_t s);
size_t axpy_bcast_chunk(size_t nterms, size_t len);
hipError_t launch_axpy_bcast_u32(int nb, HatView partial, size_t partial_bs, ConstHatView G, size_t g_bs, ConstHatView C,
                                 size_t nterms, size_t chunk, size_t len, int nlimbs, const LimbTable* tabs, hipStream_t s);
hipError_t launch_axpy_bcast_u16(int nb, uint32_t* partial_plane, size_t partial_bs, const uint32_t* G_plane, size_t g_bs,
                                 const uint16_t* C, size_t nterms, size_t chunk, size_t len, const LimbTable* tab, hipStream_t s);
// pack (unpack = false: dst16 <- src32) or unpack (unpack = true: udst32 <- dst16) one limb plane of `count` elements
hipError_t launch_pack_u16(uint16_t* dst16, const uint32_t* src32, size_t count, bool unpack, uint32_t* udst32,
                           hipStream_t s);
// out[i*rows + rho] = <A[rho,:], S[:,i]>   A: [rows][inner], S: [inner][ncols]
hipError_t launch_matmul_bcast(HatView out, size_t out_set_stride, size_t out_job_stride, ConstHatView A, ConstHatView S,
                               size_t s_set_stride, size_t s_col_stride, size_t s_job_stride, size_t njobs, size_t rows,
                               size_t inner, size_t nsets, int nlimbs, const LimbTable* tabs, hipStream_t s);
hipError_t launch_matmul_t(HatView out, ConstHatView A, ConstHatView S, size_t rows, size_t inner, size_t ncols,
                           int nlimbs, const LimbTable* tabs, hipStream_t s);
// ---- mfma_kernels.hip: exact int8 GEMM formulation of the batched outer commitment
hipError_t launch_pack_matrix_i8(int8_t* dst, const void* src, bool src_u16, size_t nmat, size_t rows, size_t cols,
                                 size_t rt_count, size_t kt_count, int planes, hipStream_t s);
hipError_t launch_pack_vectors_i8(int ns, int ns_valid, int8_t* dst, const uint32_t* src, size_t set_stride, size_t k_stride,
                                  size_t ktotal, size_t kt_count, size_t nt_count, int planes, size_t njobs, size_t src_job_stride,
                                  size_t dst_job_stride, size_t nsub, size_t src_sub_stride, hipStream_t s);
struct Gemm8Args {
  uint32_t* out;             // this limb's plane: out + job*out_job_stride + s*out_set_stride + row*64 + j
  size_t out_set_stride, out_job_stride;
  const int8_t* M8;
  const int8_t* V8;          // + job*v8_job_stride (bytes)
  size_t v8_job_stride;
  unsigned planes, rt_count, kt_count, rgroups /* row groups per slot */, njobs, rows, ns, ns_valid;
  unsigned kt_real;         // k-tiles that hold terms (the rest of kt_count is zero padding): k_gemm_i8_w3 stops after them
  unsigned nsub;            // sets per source job (= ns unless several source jobs are multiplied as one job); ns_valid counts within it
  unsigned jobs_per_block = 1;   // k_gemm_i8_w3p: jobs a persistent block works off (set by launch_gemm_i8)
  unsigned tile_r, tile_j;   // block-order super-tile (set by launch_gemm_i8)
  uint32_t p;
  const LimbTable* tab;     // this limb's table (device): Montgomery constants for the recombination
  uint32_t pw[7];           // 256^e mod p, e = 0..6 (recombination weights)
  uint32_t bias;            // 2^31 * (sum_{i<planes} 256^i)^2 mod p: what entering every D as D + 2^31 adds to the sum
  uint32_t* stage;          // non-null: results go to stage[job][slot][row][set] (full-line stores), see k_slot_transpose
  int diag = 0;             // lab_ctx_gemm_diag: 0 = the product kernel; 1 = + in-kernel clock stamps; 2 / 3 = load / multiply side removed
  uint64_t* diag_buf = nullptr;   // [block][2]: shader-clock ticks (s_memtime) and constant-clock ticks (s_memrealtime) of the K loop
  size_t diag_words = 0;
};
hipError_t launch_slot_transpose(uint32_t* out_plane, size_t out_set_stride, size_t out_job_stride, const uint32_t* stage,
                                 size_t rows, size_t rows_pad, size_t ns, size_t ns_valid, size_t njobs, size_t nsub, hipStream_t s);
int gemm_i8_waves(int planes, size_t rt_count, size_t nt);   // waves per block (4 or 8) the launcher will use
hipError_t launch_gemm_i8(const Gemm8Args& a, size_t nt, int waves, hipStream_t s);
// X = P^T S per proof (G = S^T S when P8a is null or S8; two operands in one pass when P8b is given) from the B-operand tiles of its (at most 32) columns, 3-plane limb (see
// k_gram_v8); `sub` source jobs share one tile set (2 when two proofs were packed as one 64-set job)
hipError_t launch_gram_v8(const int8_t* P8a, const int8_t* P8b, const int8_t* S8, size_t v8_job_stride, size_t kt_count, size_t kt_real,
                          size_t nt_count, size_t sub, size_t r, size_t njobs, uint32_t* out_a, uint32_t* out_b, size_t out_job_stride,
                          uint32_t p, const uint32_t* pw, uint32_t bias, const LimbTable* tab, hipStream_t s);
// 16-bit decomposition digits ([ktotal][64] per set, src16.p[s]) -> transform -> B-operand tiles of every limb
// (dst + l*limb_stride), see k_digits_to_v8; digits must be below every limb prime
// launch_digits_to_v8 with the digits made on the fly (dt): src16.p[s] is then the set's t = A s itself ([r][kappa][64] 64-bit
// coefficients below q), and K position x of the call stands for digit (m0 + x / kappa) % t1 of polynomial
// ((m0 + x / kappa) / t1) * kappa + x % kappa — the [i][k][rho] order of decompose_polynomial_vec's output (src/util.rs:360-442),
// whose 16-bit copy (6 bytes per coefficient of t) then never exists.  base below 2^16 and not above q < 2^40.
struct DigitsFromT { uint32_t base, t1; size_t kappa, m0; uint64_t q; };
hipError_t launch_digits_to_v8(const JobPtrs& src16, size_t ktotal, int8_t* dst, size_t limb_stride, size_t kt_count, int ns,
                               int ns_valid, int nlimbs, const LimbTable* tabs, const int* planes, const size_t* nt, hipStream_t s, const DigitsFromT* dt = nullptr);
// the columns of [n][r] coefficient matrices (64-bit words below q < 2^40, one matrix per proof) -> B-operand tiles of every limb
// (limb primes below 2^23), ns / 32 proofs per job
hipError_t launch_coefs_to_v8(const JobPtrs& src64, size_t n, size_t r, int8_t* dst, size_t limb_stride, size_t job_stride, size_t kt_count,
                              int ns, size_t njobs, int nlimbs, const LimbTable* tabs, const int* planes, const size_t* nt, hipStream_t s);
// rows of a coefficient matrix (64-bit words below q) -> A-operand tiles of every limb (3-plane limbs), see k_coefs_to_a8
hipError_t launch_coefs_to_a8(const uint64_t* coefs, size_t nrows, size_t cols, size_t row_base, size_t kbase, int8_t* const* dst,
                              size_t rt_count, size_t kt_count, int nlimbs, const LimbTable* tabs, hipStream_t s);
// tiled Gram: partial[chunk][i][k], gram_chunks(..) <= gram_num_chunks(n) chunks, to be summed with launch_sum_terms(rows = r*r)
size_t gram_num_chunks(size_t n);
size_t gram_chunks(size_t n, size_t r, int nlimbs, size_t njobs);
hipError_t launch_gram_tiled(HatView partial, ConstHatView P, ConstHatView S, size_t n, size_t r, int nlimbs,
                             const LimbTable* tabs, hipStream_t s, size_t njobs = 1, size_t js = 0);
// partial[chunk*len + rho] = sum_{m in chunk} G[m] * C[m*len + rho]; axpy_num_chunks(nterms) chunks
size_t axpy_num_chunks(size_t nterms);
hipError_t launch_axpy_partial(HatView partial, ConstHatView G, ConstHatView C, size_t nterms, size_t len, int nlimbs,
                               const LimbTable* tabs, hipStream_t s);
hipError_t launch_poly_by_vec(HatView out, ConstHatView poly, ConstHatView v, size_t len, int nlimbs,
                              const LimbTable* tabs, hipStream_t s, size_t njobs = 1, size_t js = 0);

// ---- proof_kernels.hip (coefficient domain, residues mod q as u64)
hipError_t launch_decompose(uint64_t* out, const uint64_t* in, size_t count, uint64_t q, uint64_t base, int exp,
                            size_t group, hipStream_t s);
hipError_t launch_gather_upper(uint64_t* out, const uint64_t* in, size_t r, hipStream_t s);
// the same kernels serving njobs independent proofs per launch (per-job pointers by value, see JobPtrs)
hipError_t launch_decompose_jobs(const JobPtrs& out, const JobPtrs& in, size_t count, uint64_t q, uint64_t base, int exp,
                                 size_t group, size_t njobs, hipStream_t s, bool out_u16 = false /* digits as 16-bit words */);
hipError_t launch_gather_upper_jobs(const JobPtrs& out, const JobPtrs& in, size_t r, size_t njobs, hipStream_t s);
hipError_t launch_gather_jobs(uint64_t* dst, const JobPtrs& src, size_t words, size_t njobs, hipStream_t s);
hipError_t launch_scatter_jobs(const JobPtrs& dst, const uint64_t* src, size_t words, size_t njobs, hipStream_t s);
hipError_t launch_lift_i64_jobs(const JobPtrs& out, const JobPtrs& in, size_t n, uint64_t q, size_t njobs, hipStream_t s);
hipError_t launch_coef_elementwise_jobs(const JobPtrs& out, const JobPtrs& a, const JobPtrs& b, size_t count, uint64_t q,
                                        const JobU64& s1, int mode, size_t njobs, hipStream_t s);
hipError_t launch_symmetrize_half_jobs(const JobPtrs& H, const JobPtrs& X, size_t r, uint64_t q, uint64_t inv2, size_t njobs,
                                       hipStream_t s);
hipError_t launch_pi_fused_jobs(const JobPtrs& proj, const JobPtrs& W, const JobPtrs& Pi, const JobPtrs& S, const JobPtrs& omega,
                                size_t n, size_t r, uint64_t q, size_t njobs, hipStream_t s);
// the same pass over Pi at 2 bits per entry in MFMA-operand tiles (field = entry + 1; see k_pi_mfma); launch_pi_pack converts
// `mats` int8 matrices of 256 x cols
// above16 (with small_witness): [njobs] flags from launch_to_hat_fused — jobs whose witness coefficients all fit 16 bits take the
// projection on v_dot2_u32_u16 (two entries per multiply-add)
hipError_t launch_pi_fused2_jobs(const JobPtrs& proj, const JobPtrs& W, const JobPtrs& Pi2, const JobPtrs& S, const JobPtrs& omega,
                                 size_t n, size_t r, uint64_t q, size_t njobs, bool small_witness, hipStream_t s,
                                 const unsigned* above16 = nullptr);
hipError_t launch_pi_pack(uint8_t* dst, const int8_t* src, size_t mats, size_t cols, hipStream_t s);
hipError_t launch_lift_i64(uint64_t* out, const int64_t* in, size_t n, uint64_t q, hipStream_t s);
hipError_t launch_norm_sq(uint64_t* lo, uint64_t* hi, const uint64_t* in, size_t count, hipStream_t s);
// mode 0: s1*a ; 1: s1*a + b ; 2: a + b ; 3: sigma_inv(a)
hipError_t launch_coef_elementwise(uint64_t* out, const uint64_t* a, const uint64_t* b, size_t count, uint64_t q,
                                   uint64_t s1, int mode, hipStream_t s);
hipError_t launch_symmetrize_half(uint64_t* H, const uint64_t* X, size_t r, uint64_t q, uint64_t inv2, hipStream_t s);
hipError_t launch_jl_project(int64_t* proj, const int8_t* Pi, const uint64_t* S, size_t n, size_t r, hipStream_t s);
hipError_t launch_phi_pi_part(uint64_t* W, const int8_t* Pi, const uint64_t* omega, size_t n, size_t r, uint64_t q,
                              hipStream_t s);
// JL projection and the Pi part of phi'' in one pass over Pi (proj must be zeroed)
hipError_t launch_pi_fused(int64_t* proj, uint64_t* W, const int8_t* Pi, const uint64_t* S, const uint64_t* omega,
                           size_t n, size_t r, uint64_t q, hipStream_t s);
hipError_t launch_compare_u64(unsigned int* flag, const uint64_t* a, const uint64_t* b, size_t nwords, hipStream_t s);
hipError_t launch_symmetry_u64(unsigned int* flag, const uint64_t* X, size_t r, hipStream_t s);
hipError_t launch_add_trace_jobs(const JobPtrs& out, const JobPtrs& base, const JobPtrs& X, size_t r, uint64_t q, size_t njobs,
                                 hipStream_t s);
hipError_t launch_check18(uint64_t* res, const uint64_t* s1, const uint64_t* H, const uint64_t* b, size_t r, uint64_t q,
                          hipStream_t s);

// ---- crs_kernels.hip: CRS::random_oracle_gen on the device (src/structs.rs:35-45, 147-171)
hipError_t launch_crs_expand(uint64_t* coeffs, size_t count, const uint8_t base_seed[32], uint64_t off_lo,
                             uint64_t off_hi, uint64_t q, hipStream_t s);

}  // namespace lab
