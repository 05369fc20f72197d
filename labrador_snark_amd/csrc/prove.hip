// prove.hip — lab_prove_core: Prover::proof_gen (src/proofgen.rs:30-427) as one
// deterministic sequence of gfx950 kernel launches on the context's stream.
//
// Structure: witness/state are transformed once; every ring product runs in
// the transform domain; a result that feeds another product goes through the
// coefficient domain (exact CRT, mod q) first, so every transform-domain
// reduction starts from canonical residues and stays inside the CRT range.
// Zq-scalar multiplications (psi, omega, 2^-1) and digit decompositions are
// coefficient-domain kernels.  Step numbers follow SURVEY.md §3.2.
// phi'' and the final phi (src/proofgen.rs:236-254, 301-314) are never formed: they only occur inside inner products
// with the witness, which are linear — see steps 6-8 below.
#include <math.h>
#include <string.h>

#include <coroutine>
#include <vector>

#include "context.h"

using namespace lab;
typedef unsigned __int128 u128;

namespace lab {
int axpy_commit_impl(lab_ctx* c, HatView out, ConstHatView G, ConstHatView C, size_t nterms, size_t len, HatView partial);
int inner_product_impl(lab_ctx* c, HatView out, ConstHatView v1, ConstHatView v2, size_t len, HatView partial);
size_t inner_product_splits(size_t len);
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

namespace {

struct Job {  // device pointers of one proof of the batch
  uint64_t *cT, *cTdec, *cG, *cGup, *cGdec, *cW, *capp, *cX1, *cX2, *cX, *cH, *cHup, *cHdec, *cu1, *cu2, *cz, *cbpp, *cab;
  int64_t* dproj_int;
  uint64_t* dproj;
  const uint64_t *dS, *dphi, *da, *domega, *dalpha, *dbeta, *dc;
  const int8_t* dPi;
  const uint8_t* dPi2;
};

}  // namespace

// Prover::proof_gen for `np` independent proofs that share one common reference string.  The per-proof kernels take all
// proofs of the call as jobs of one launch; the outer commitment (step 3), whose cost is streaming the CRS matrices B_ik, is
// issued once per group of proofs — up to 64 per pass over B on the matrix cores (B8 or B regenerated from crs_seed), up to 8
// with the multi-set multiply-add kernel — so B is read (or expanded) once per group instead of once per proof.
// A batch of more than MAX_JOBS proofs is cut into sub-batches of MAX_JOBS.  With B streamed from the seed the sub-batches must
// SHARE the pass over the regenerated rows (the expansion is ~80 % of such a step and does not depend on the number of proofs), so
// the body below is a coroutine: every sub-batch runs steps 1-2 (up to the digits of t), suspends, the driver runs ONE streamed
// step 3 for all of them (crs_stream_run with one group of vector sets per sub-batch) and resumes each for steps 4-10.  A batch that
// fits one sub-batch, or has B resident, runs the same body straight through.
struct ProveTask {
  struct promise_type {
    int rc = LAB_OK;
    ProveTask get_return_object() { return ProveTask{std::coroutine_handle<promise_type>::from_promise(*this)}; }
    std::suspend_never initial_suspend() noexcept { return {}; }
    std::suspend_always final_suspend() noexcept { return {}; }
    void return_value(int v) { rc = v; }
    void unhandled_exception() { rc = LAB_ERR_HIP; }
  };
  std::coroutine_handle<promise_type> h;
  ProveTask(std::coroutine_handle<promise_type> hh) : h(hh) {}
  ProveTask(ProveTask&& o) noexcept : h(o.h) { o.h = nullptr; }
  ProveTask(const ProveTask&) = delete;
  ~ProveTask() { if (h) h.destroy(); }
  bool done() const { return h.done(); }
  int rc() const { return h.promise().rc; }
  void resume() { h.resume(); }
};
constexpr int REDO_IN_MAIN_BASIS = 1 << 20;   // internal result of a sub-batch (never leaves this file)
struct ProveShared {           // the streamed step 3 the sub-batches of one call share
  std::vector<CrsGroup> groups;
  int rc = LAB_OK;
  // ... and the dense inverse-transform scratch: the sub-batches enqueue their kernels one after the other on one stream and
  // nothing in it outlives a launch sequence, so ONE area (the largest any of them needs) at dense_off of the allocation serves
  // all of them (6 GB per sub-batch at kappa = 2^14)
  size_t dense_bytes = 0, dense_off = 0;
  // B resident as byte planes (B8): the sub-batches' digit tiles are the jobs of ONE matrix-core launch per limb — B is read from
  // HBM once for all of them (the other jobs' blocks hit L2) — through a staging array at stage_off
  bool resident = false;
  size_t stage_off = 0;
};
#define LAB_HIPC(ctx, expr)                                                                           \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess) co_return lab::fail(ctx, LAB_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// the 32-byte seed of two proofs names the same CRS (by content: callers marshal it per proof)
static bool same_seed(const uint8_t* a, const uint8_t* b) {
  if (a == b) return true;
  return a && b && memcmp(a, b, 32) == 0;
}

static int prove_batch_impl(lab_ctx* c, size_t np, const lab_prove_input* const* ins, lab_transcript* const* outs, int* status,
                            bool allow_digit_ajtai = true);

// own_scratch: the call sizes c->scratch itself and carves from its start; otherwise it carves from scratch_off inside an
// allocation the driver made (region sized by a size_only call: *size_only receives the bytes and nothing is launched)
static ProveTask prove_batch_co(lab_ctx* c, size_t np, const lab_prove_input* const* ins, lab_transcript* const* outs, int* status,
                                bool allow_digit_ajtai, ProveShared* sh, bool own_scratch, size_t scratch_off, size_t* size_only) {
  if (!c) co_return LAB_ERR_ARG;
  if (np == 0) co_return LAB_OK;
  PtrCacheScope ptr_cache;   // every boundary pointer is classified host/device once per call
  if (!ins || !outs) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: null argument");
  for (size_t j = 0; j < np; j++) {
    if (!ins[j] || !outs[j]) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: null argument");
    if (ins[j]->struct_size != sizeof(lab_prove_input))
      co_return fail(c, LAB_ERR_ARG, "lab_prove_core: lab_prove_input.struct_size is %zu, this library's layout has %zu bytes (stale binding?)",
                  ins[j]->struct_size, sizeof(lab_prove_input));
  }
  if (np > (size_t)MAX_JOBS) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: internal: a sub-batch of %zu proofs", np);
  const lab_prove_input* in = ins[0];
  const size_t n = in->n, r = in->r, kappa = in->kappa, k1 = in->kappa1, k2 = in->kappa2;
  const int t1 = in->t1, t2 = in->t2;
  if (n == 0 || r == 0 || kappa == 0 || k1 == 0) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: empty dimension");
  if (k1 != k2) co_return fail(c, LAB_ERR_LEN_MISMATCH, "lab_prove_core: u_1 = lhs + rhs needs kappa1 == kappa2 (src/proofgen.rs:153)");
  if (t1 < 1 || t2 < 1 || t1 > 64 || t2 > 64) co_return fail(c, LAB_ERR_DECOMPOSE, "lab_prove_core: t1/t2 out of range (t2 = -1 is what RuntimeConstants gives at Q=8191, n=1024)");
  if (in->b1 < 2 || in->b2 < 2) co_return fail(c, LAB_ERR_DECOMPOSE, "lab_prove_core: decomposition base < 2 never terminates in the reference");
  if (r > 4096 || n > (1u << 22) || kappa > (1u << 24) || k1 > (1u << 24)) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: dimension too large");
  for (size_t j = 0; j < np; j++) {
    const lab_prove_input* x = ins[j];
    if (!x->S || !x->phi || !x->a || (!x->Pi && !x->Pi2) || !x->omega || !x->alpha || !x->beta || !x->c)
      co_return fail(c, LAB_ERR_ARG, "lab_prove_core: null input array");
    if ((x->Pi2 != nullptr) != (in->Pi2 != nullptr))
      co_return fail(c, LAB_ERR_ARG, "lab_prove_core_batch: either every proof of a batch hands over Pi2 (2 bits per entry) or none");
    if (x->n != n || x->r != r || x->kappa != kappa || x->kappa1 != k1 || x->kappa2 != k2 || x->t1 != t1 || x->t2 != t2 ||
        x->b1 != in->b1 || x->b2 != in->b2 || x->beta_bound != in->beta_bound || x->A != in->A || x->B != in->B || !same_seed(x->crs_seed, in->crs_seed) || x->B8 != in->B8 || x->A8 != in->A8 || x->C8 != in->C8 || x->D8 != in->D8 || x->A8d != in->A8d ||
        x->C != in->C || x->Dm != in->Dm || x->a_off != in->a_off || x->b_off != in->b_off || x->c_off != in->c_off ||
        x->d_off != in->d_off)
      co_return fail(c, LAB_ERR_ARG, "lab_prove_core_batch: all proofs of a batch must share dimensions, constants and the CRS");
  }
  const size_t pairs = r * (r + 1) / 2;
  const size_t nr = n * r, rk = r * kappa, rr = r * r;
  int rc;
  if ((rc = check_range(c, in->A, in->a_off, kappa * n, "lab_prove_core(A)"))) co_return rc;
  // step 3 runs on the matrix cores when the B matrices were packed for them (lab_mat8_pack); B itself may then be absent
  const lab_mat8* b8 = in->B8;
  // ... or streamed from the CRS seed, slab by slab (crs_stream.hip): B is then never resident
  const bool stream_b = !b8 && !in->B && in->crs_seed;
  if (!b8 && !in->B && !stream_b) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: neither B, B8 nor crs_seed given");
  if (in->B && (rc = check_range(c, in->B, in->b_off, r * (size_t)t1 * k1 * kappa, "lab_prove_core(B)"))) co_return rc;
  // (C and D themselves may be absent when their packed forms C8 / D8 are given: a prover-only service drops the residues, 46 GB
  // at kappa = 2^14, like it drops B)
  if (!in->C && !in->C8) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: neither C nor C8 given");
  if (!in->Dm && !in->D8) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: neither D nor D8 given");
  if (in->C && (rc = check_range(c, in->C, in->c_off, pairs * (size_t)t2 * k2, "lab_prove_core(C)"))) co_return rc;
  if (in->Dm && (rc = check_range(c, in->Dm, in->d_off, pairs * (size_t)t1 * k2, "lab_prove_core(D)"))) co_return rc;
  // B, C, D may live in the context's digit basis (lab_ctx_enable_digit_basis): steps 3, 4, 9 then run in it
  const int bdig = in->C ? in->C->basis : mat8_basis(in->C8);
  if ((in->B && in->B->basis != bdig) || (in->Dm ? in->Dm->basis : mat8_basis(in->D8)) != bdig)
    co_return fail(c, LAB_ERR_ARG, "lab_prove_core: B, C, D must share one RNS basis");
  const lab_mat8* a8 = in->A8;
  if (a8 && !mat8_matches(a8, 0, 1, kappa, n))
    co_return fail(c, LAB_ERR_ARG, "lab_prove_core: A8 was not packed from the kappa x n matrix A in the main basis");
  // Step 1 in the basis of B, C, D (lab_prove_input.A8d): |<A row, s_i>| <= |A row|_2 |s_i|_2 <= sqrt(64 n) (q-1) beta_bound must
  // lie inside that basis' CRT range; the witness norm is summed (saturating, exact) by the pass that transforms S and
  // checked at the end of the call — a witness above beta_bound makes the call start over without this shortcut.
  const lab_mat8* a8d = allow_digit_ajtai ? in->A8d : nullptr;
  uint64_t norm_limit = 0;
  if (in->A8d && (!bdig || !mat8_matches(in->A8d, bdig, 1, kappa, n)))
    co_return fail(c, LAB_ERR_ARG, "lab_prove_core: A8d was not packed from the kappa x n matrix A in the basis of B, C, D");
  if (a8d) {
    long double M = 1.0L;
    for (int l = 0; l < c->dg_nlimbs; l++) M *= (long double)c->dg_primes[l];
    // (z = sum_{i<r} c_i s_i of step 10 runs in that basis too under sd_only: its Cauchy-Schwarz bound has r in place of n)
    const long double need = sqrtl((long double)(n > r ? n : r) * D) * (long double)(c->q - 1) * (long double)in->beta_bound * 1.0001L;
    const size_t blocks = (nr + 63) / 64 + 1;
    if (in->beta_bound <= 0 || in->beta_bound >= (int64_t)1 << 24 || need >= M / 2 ||
        (u128)blocks * ((u128)in->beta_bound * in->beta_bound + 1) >= ((u128)1 << 64))
      a8d = nullptr;
    else
      norm_limit = (uint64_t)in->beta_bound * (uint64_t)in->beta_bound;
  }
  // C and D packed the same way: element (rho, m) of the "matrix" is C[m][rho], i.e. pairs*t matrices of kappa2 x 1
  const lab_mat8 *c8 = in->C8, *d8 = in->D8;
  if ((c8 && !mat8_matches(c8, bdig, pairs * (size_t)t2, k2, 1)) || (d8 && !mat8_matches(d8, bdig, pairs * (size_t)t1, k2, 1)))
    co_return fail(c, LAB_ERR_ARG, "lab_prove_core: C8 / D8 were not packed as lab_mat8_pack(C, c_off, pairs*t2, kappa2, 1) / (D, d_off, pairs*t1, kappa2, 1)");
  if (b8 && !mat8_matches(b8, bdig, r * (size_t)t1, k1, kappa))
    co_return fail(c, LAB_ERR_ARG, "lab_prove_core: B8 was not packed from r*t1 matrices of kappa1 x kappa elements in the basis of C, D");
  if (in->A->basis != 0 || in->A->compact) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: A must be a main-basis vector");
  // exactness of every transform-domain reduction (see header comment)
  const u128 qm1 = c->q - 1;
  const u128 bmax = (u128)(in->b1 > in->b2 ? in->b1 : in->b2);
  if (!exact_ok(c, n, qm1) || !exact_ok(c, rr + nr, qm1) || !exact_ok(c, r, qm1))
    co_return fail(c, LAB_ERR_RANGE, "lab_prove_core: a reduction exceeds the CRT range of this context; create it with a larger max_len");
  // (B streamed: the K chunks of step 3 are reduced mod q one by one when the whole sum exceeds the basis — crs_stream.hip — so
  // only one chunk's sum has to be exact)
  u128 outer_terms = (u128)r * t1 * kappa;
  const uint64_t rhs_terms = (uint64_t)pairs * t2;
  if (stream_b) {
    size_t mc = 0;
    bool reduce = false;
    if (!crs_stream_plan(c, bdig, r * (size_t)t1, kappa, rhs_terms, &mc, &reduce))
      co_return fail(c, LAB_ERR_RANGE, "lab_prove_core: one row of B (%zu terms) exceeds the int32 bound of the int8 GEMM or the exactness bound of the CRS basis", kappa);
    if (reduce) outer_terms = (u128)mc * kappa;
  }
  if (bdig) {
    if (outer_terms + (u128)pairs * t2 > c->dg_terms || (u128)pairs * t1 > c->dg_terms || bmax - 1 > c->dg_vmax)
      co_return fail(c, LAB_ERR_RANGE, "lab_prove_core: outer commitments exceed the digit basis bounds (max_terms %llu, digit_bound %llu)",
                  (unsigned long long)c->dg_terms, (unsigned long long)(c->dg_vmax + 1));
  } else if (!exact_ok(c, outer_terms + (u128)pairs * t2, bmax) || !exact_ok(c, (u128)pairs * t1, (u128)in->b1)) {
    co_return fail(c, LAB_ERR_RANGE, "lab_prove_core: a reduction exceeds the CRT range of this context; create it with a larger max_len");
  }
  if ((u128)r * n * D * qm1 >= ((u128)1 << 63)) co_return fail(c, LAB_ERR_RANGE, "lab_prove_core: JL projection does not fit 64 bits");
  LAB_HIPC(c, hipSetDevice(c->device));

  // the digits of t go straight into B-operand tiles of every limb (transform + packing fused) when they are 16-bit words
  // below every prime of B's basis: their transform-domain copy then never exists
  const bool fuse_t = (b8 || stream_b) && in->b1 <= 65536 && in->b2 <= 65536 &&
                      (uint64_t)in->b1 <= (b8 ? mat8_min_prime(c, b8) : basis_min_prime(c, bdig));
  // ---------------- workspace plan: one arena per basis, proof j owns elements [j*per, (j+1)*per) of every plane
  const int L = c->nlimbs;
  const bool sd_only = a8d && gram_mfma_ok(a8d, n, r);   // every product with the witness in the basis of B, C, D (see hPd below)
  HatArena ha;
  // t = A s leaves the matrix-core product of several proofs through the GEMM's staging array straight into coefficients
  // (MfmaCoefOut): its transform-domain copy is then never written — 0.4 + 0.5 GB per proof at kappa = 2^14
  const bool t_coef_only = (a8d || in->A8) && np > 1;
  const size_t hS = ha.take(sd_only ? 1 : nr), hphi = ha.take(sd_only ? 1 : nr), hT = ha.take(a8d || t_coef_only ? 1 : rk), hG = ha.take(rr),
               hphipp = ha.take(sd_only ? 1 : nr),
               happ = ha.take(rr), hGc = ha.take(rr), hbpp = ha.take(2), hX1 = ha.take(rr), hX2 = ha.take(rr),
               hX = ha.take(rr), hc = ha.take(r), hz = ha.take(n), hab = ha.take(2);
  size_t part_elems = 512;  // inner_product_impl splits a long inner product into <= 512 partials
  if (gram_num_chunks(n) * rr > part_elems) part_elems = gram_num_chunks(n) * rr;
  const size_t hpart = ha.take(part_elems);
  // second arena, in the basis of B/C/D, for the digit-operand reductions (steps 3, 4, 9)
  HatArena hd;
  const size_t hTdec = hd.take(fuse_t ? 1 : rk * t1), hGdec = hd.take(pairs * t2), hlhs = hd.take(k1), hrhs = hd.take(k1),
               hHdec = hd.take(pairs * t1), hu2 = hd.take(k2);
  // partial sums of the multiply-add kernels (not needed for the steps that run on the matrix cores)
  size_t partd_elems = (b8 || stream_b) ? 1 : r * (size_t)t1 * k1;
  if (!in->C8 && axpy_partial_elems(pairs * t2, k2) > partd_elems) partd_elems = axpy_partial_elems(pairs * t2, k2);
  if (!in->D8 && axpy_partial_elems(pairs * t1, k2) > partd_elems) partd_elems = axpy_partial_elems(pairs * t1, k2);
  const size_t hpartd = hd.take(partd_elems);
  const size_t hSd = a8d ? hd.take(nr) : 0, hTd = a8d ? hd.take(t_coef_only ? 1 : rk) : 0;   // witness and t = A s in the digit basis
  const size_t hGd = a8d ? hd.take(rr) : 0;                                 // G = S^T S from the same packed operands
  // Under the same norm bound every other product with the witness is exact in that basis too (|<x, s>| <= |x|_2 |s|_2 for any
  // x with entries below q): phi'^T S, W^T S (steps 6-8) on the matrix cores from packed tiles, z = sum_i c_i s_i (step 10)
  // from the digit-basis witness — the main-basis transforms of S, phi' and W are then never made.
  const size_t hX1d = sd_only ? hd.take(rr) : 0, hXWd = sd_only ? hd.take(rr) : 0, hzd = sd_only ? hd.take(n) : 0,
               hcd = sd_only ? hd.take(r) : 0;
  const int Ld = basis_nlimbs(c, bdig);
  const LimbTable* tabsd = basis_tabs(c, bdig);
  const size_t per = ha.total, perd = hd.total;
  const size_t hat_words = np * per * D, hatd_words = np * perd * D;
  size_t tmp_small = nr;
  if (k1 > tmp_small) tmp_small = k1;
  if (rr > tmp_small) tmp_small = rr;
  // The chain t = A S -> inverse transform -> CRT runs `t_chunk` proofs at a time when the dense inverse-transform output of all
  // proofs at once would be large (0.4 GB per proof at kappa = 2^14: with it, fewer proofs fit one streamed pass over B)
  const size_t Lmax = (size_t)(c->nlimbs > basis_nlimbs(c, bdig) ? c->nlimbs : basis_nlimbs(c, bdig));
  size_t t_chunk = np;
  if (t_coef_only && a8d && r <= 32 && in->b1 <= 65536 && in->b2 <= 65536)   // (the slices finish with the CRT pass that writes 16-bit digits)
    while (t_chunk % 4 == 0 && t_chunk > 16 && t_chunk * Lmax * rk * D * 4 > ((size_t)6 << 30)) t_chunk /= 2;
  const size_t tmp_elems = t_coef_only ? tmp_small : (rk > tmp_small ? rk : tmp_small);   // from_hat scratch per job

  // decomposition digits travel between k_decompose / the CRT pass and the transforms as 16-bit words when both bases allow it
  const bool dig16 = in->b1 <= 65536 && in->b2 <= 65536;
  // B streamed: the digit tiles of a K chunk are made from t itself (DigitsFromT, kernels.h) — the 16-bit copy of all digits
  // (6 bytes per coefficient of t: 0.2 GB per proof at kappa = 2^14) is never written, and more proofs fit one pass over B
  const bool t_digits_lazy = stream_b && fuse_t && dig16 && in->b1 >= 2 && in->b1 < 65536 && (uint64_t)in->b1 <= c->q && c->q < (1ull << 40);
  const size_t tdec_elems = t_digits_lazy ? 0 : dig16 ? (rk * t1 + 3) / 4 : rk * t1;   // in units of 64 64-bit words
  const size_t coef_elems = rk /*cT*/ + tdec_elems /*cTdec*/ + rr /*cG*/ + pairs /*cGup*/ + pairs * t2 /*cGdec*/ +
                            nr /*cW*/ + rr /*capp*/ + 2 * rr /*cX1, cX2*/ + rr /*cX*/ + rr /*cH*/ +
                            pairs /*cHup*/ + pairs * t1 /*cHdec*/ + k1 /*cu1*/ + k2 /*cu2*/ + n /*cz*/ + 1 /*cbpp*/ +
                            2 /*cab*/;
  size_t dense_words = np * Lmax * tmp_elems * D;
  if (t_coef_only && t_chunk * Lmax * rk * D > dense_words) dense_words = t_chunk * Lmax * rk * D;
  // t = A S of up to 32 columns per proof leaves the matrix cores in one slot-major staging array per limb; with room for all of
  // them the inverse transform and the CRT (with the digits of t) are ONE pass over them (launch_stage_to_coeffs): the dense
  // array is that room ([limb][gemm job][slot][row, padded to 32][set])
  const int a8_ns = r <= 8 ? 8 : r <= 16 ? 16 : 32;       // witness columns per Ajtai GEMM pass
  const size_t t_stage_ps = t_chunk * (size_t)a8_ns * D * ((kappa + 31) / 32 * 32);
  const int t_limbs = a8d ? basis_nlimbs(c, bdig) : c->nlimbs;
  const bool t_staged = t_coef_only && r <= 32 && t_limbs >= 2 && t_limbs <= 3 && c->q < (1ull << 40) && kappa < ((size_t)1 << 32) &&
                        (!(in->b1 <= 65536 && in->b2 <= 65536) || (in->b1 >= 2 && in->b1 < 65536 && (uint64_t)in->b1 <= c->q));
  if (t_staged && (size_t)t_limbs * t_stage_ps > dense_words) dense_words = (size_t)t_limbs * t_stage_ps;
  if (sh && Carver::padded(dense_words * 4) > sh->dense_bytes) {
    if (!size_only) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: internal: the shared scratch was sized for less");
    sh->dense_bytes = Carver::padded(dense_words * 4);
  }
  size_t bytes = Carver::padded((size_t)L * hat_words * 4) + Carver::padded((size_t)Ld * hatd_words * 4) +
                 (sh ? 0 : Carver::padded(dense_words * 4)) +
                 np * (coef_elems * (D * 8) + 32 * 256 /*alignment padding of the carves*/ + 3 * 4096 /*proj*/);
  for (size_t j = 0; j < np; j++) {   // t produced in place in a device-resident caller buffer needs no copy here (268 MB per proof at kappa = 2^14)
    if (outs[j]->t_i_all && is_device_pointer(outs[j]->t_i_all)) bytes -= rk * (size_t)(D * 8);
  }
  for (size_t j = 0; j < np; j++) {
    const lab_prove_input* x = ins[j];
    bytes += Staging::need(x->S, nr * D * 8) + Staging::need(x->phi, nr * D * 8) + Staging::need(x->a, rr * D * 8) +
             (x->Pi2 ? Staging::need(x->Pi2, r * 256 * n * (D / 4)) : Staging::need(x->Pi, r * 256 * n * D)) + Staging::need(x->omega, 256 * 8) + Staging::need(x->alpha, D * 8) +
             Staging::need(x->beta, D * 8) + Staging::need(x->c, r * D * 8);
  }
  // int8-MFMA step 3: padded set count of the largest group and its scratch (B-operand tiles + int32 results)
  const int mf_ns = np > 32 ? 64 : np > 16 ? 32 : np > 8 ? 16 : 8;
  size_t mf_bytes = b8 ? mfma_scratch_bytes(b8, mf_ns, 1) + 4096 : 0;
  if (stream_b && !fuse_t) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: the streamed outer commitment needs decomposition bases up to 65536 below the limb primes");
  if (stream_b && crs_stream_scratch_bytes(c, bdig, r * (size_t)t1, k1, kappa, mf_ns, rhs_terms) + 4096 > mf_bytes)
    mf_bytes = crs_stream_scratch_bytes(c, bdig, r * (size_t)t1, k1, kappa, mf_ns, rhs_terms) + 4096;
  if (fuse_t && b8 && mfma_prepacked_bytes(b8, mf_ns) + 4096 > mf_bytes) mf_bytes = mfma_prepacked_bytes(b8, mf_ns) + 4096;
  if (a8 && mfma_scratch_bytes(a8, a8_ns, np) + 4096 > mf_bytes) mf_bytes = mfma_scratch_bytes(a8, a8_ns, np) + 4096;
  if (a8d && mfma_scratch_bytes(a8d, a8_ns, t_chunk) + 4096 > mf_bytes) mf_bytes = mfma_scratch_bytes(a8d, a8_ns, t_chunk) + 4096;
  if (sd_only && gram_mfma_scratch_bytes(a8d, np) + 4096 > mf_bytes) mf_bytes = gram_mfma_scratch_bytes(a8d, np) + 4096;
  const size_t stile_bytes = sd_only ? gram_mfma_tile_bytes(a8d, np) + 4096 : 0;   // the witness tiles of every limb, kept from step 1 to step 7
  bytes += stile_bytes;
  if (c8 && mfma_scratch_bytes(c8, mf_ns, 1) + 4096 > mf_bytes) mf_bytes = mfma_scratch_bytes(c8, mf_ns, 1) + 4096;
  if (d8 && mfma_scratch_bytes(d8, mf_ns, 1) + 4096 > mf_bytes) mf_bytes = mfma_scratch_bytes(d8, mf_ns, 1) + 4096;
  bytes += mf_bytes;
  if (size_only) { *size_only = bytes + (1 << 16); co_return LAB_OK; }
  if (own_scratch && (rc = ensure_scratch(c, bytes + (1 << 16)))) co_return rc;
  Carver cv(static_cast<char*>(c->scratch) + (own_scratch ? 0 : scratch_off));
  char* mf_scratch = mf_bytes ? cv.take<char>(mf_bytes) : nullptr;
  int8_t* s_tiles = stile_bytes ? cv.take<int8_t>(stile_bytes) : nullptr;
  uint32_t* hat_base = cv.take<uint32_t>((size_t)L * hat_words);
  uint32_t* hatd_base = cv.take<uint32_t>((size_t)Ld * hatd_words);
  // from_hat scratch: a dense [L][count][64] buffer (its plane stride is count*64, not the arena's)
  uint32_t* dense_tmp = sh ? reinterpret_cast<uint32_t*>(static_cast<char*>(c->scratch) + sh->dense_off) : cv.take<uint32_t>(dense_words);   // job j: + j * max(L, Ld)*tmp_elems*64 words (the t chain: + j * max(L, Ld)*rk*64)
  auto HV = [&](size_t j, size_t off) { return HatView{hat_base + (j * per + off) * D, hat_words}; };
  auto HD = [&](size_t j, size_t off) { return HatView{hatd_base + (j * perd + off) * D, hatd_words}; };
  auto CO = [&](size_t elems) { return cv.take<uint64_t>(elems * D); };
  // results the caller wants in device memory are produced in place (no staging copy)
  auto OUT = [&](uint64_t* user, size_t elems) { return (user && is_device_pointer(user)) ? user : CO(elems); };
  std::vector<Job> jobs(np);
  // [np][256] integer projections | [np][256] omega | [np][64] b'': contiguous, so the host-side checks need ONE copy
  uint64_t* host_stage = cv.take<uint64_t>(np * (256 + 256 + D + 1));   // ... | [np] squared witness norms (digit-basis step 1)
  unsigned* above16 = cv.take<unsigned>(np);   // [np] flags: a witness coefficient of the proof needs more than 16 bits (digit-basis step 1)
  int64_t* proj_all = reinterpret_cast<int64_t*>(host_stage);
  Staging st(c, &cv);  // inputs only (no pending outputs are registered on it)
  for (size_t j = 0; j < np; j++) {
    Job& J = jobs[j];
    const lab_prove_input* x = ins[j];
    lab_transcript* o = outs[j];
    J.cT = OUT(o->t_i_all, rk); J.cTdec = CO(tdec_elems); J.cG = OUT(o->g_mat, rr); J.cGup = CO(pairs); J.cGdec = CO(pairs * t2);
    J.cW = CO(nr); J.capp = CO(rr); J.cX1 = CO(rr); J.cX2 = CO(rr); J.cX = CO(rr); J.cH = OUT(o->h_mat, rr);
    J.cHup = CO(pairs); J.cHdec = CO(pairs * t1); J.cu1 = OUT(o->u_1, k1); J.cu2 = OUT(o->u_2, k2); J.cz = OUT(o->z, n);
    J.cbpp = OUT(o->b_prime_prime, 1); J.cab = CO(2);
    J.dproj_int = proj_all + j * 256;
    J.dproj = OUT(o->projection, 4);
    if ((rc = st.in(x->S, nr * D, &J.dS))) co_return rc;
    if ((rc = st.in(x->phi, nr * D, &J.dphi))) co_return rc;
    if ((rc = st.in(x->a, rr * D, &J.da))) co_return rc;
    J.dPi = nullptr; J.dPi2 = nullptr;
    if (x->Pi2) { if ((rc = st.in(x->Pi2, r * 256 * n * (D / 4), &J.dPi2))) co_return rc; }
    else if ((rc = st.in(x->Pi, r * 256 * n * D, &J.dPi))) co_return rc;
    if ((rc = st.in(x->omega, (size_t)256, &J.domega))) co_return rc;
    if ((rc = st.in(x->alpha, (size_t)D, &J.dalpha))) co_return rc;
    if ((rc = st.in(x->beta, (size_t)D, &J.dbeta))) co_return rc;
    if ((rc = st.in(x->c, r * D, &J.dc))) co_return rc;
  }
  hipStream_t s = c->stream;
  const LimbTable* tabs = c->d_tabs;
  const uint64_t q = c->q;

  // Every per-proof kernel below is launched ONCE for all np proofs (a "job" grid dimension): transform-domain operands
  // are addressed with the arena's job stride, coefficient arrays through per-job pointers passed by value.
  const size_t js = per * D, jsd = perd * D;                       // job strides (words inside a plane) of the two arenas
  const size_t tmp_js = Lmax * tmp_elems * D;   // per-job slice of the from_hat scratch
  const size_t tmp_js_t = Lmax * rk * D;        // ... of the t chain (t_chunk jobs at a time)
  auto JP = [&](auto member) {
    JobPtrs t;
    for (size_t j = 0; j < np; j++) t.p[j] = const_cast<void*>(static_cast<const void*>(jobs[j].*member));
    return t;
  };
  // `darena`: the offset lives in the second arena (operands of the digit reductions, in the basis `bdig` of B/C/D)
  auto to_hat_jobs = [&](const JobPtrs& src, size_t off, size_t count, bool darena, uint64_t in_bound = ~0ull, bool src16 = false) -> int {
    const int basis = darena ? bdig : 0;
    HatView dst = darena ? HD(0, off) : HV(0, off);
    const size_t st = darena ? jsd : js;
    LAB_HIP(c, launch_to_hat_fused(src, dst, st, count, basis_nlimbs(c, basis), basis_tabs(c, basis), np, s, in_bound, nullptr, 0, src16));   // reduce + NTT
    return LAB_OK;
  };
  auto from_hat_jobs = [&](size_t off, size_t count, const JobPtrs& dst, bool darena) -> int {
    const int basis = darena ? bdig : 0;
    HatView src = darena ? HD(0, off) : HV(0, off);
    HatView t{dense_tmp, count * D};
    LAB_HIP(c, launch_ntt(false, src, t, count, basis_nlimbs(c, basis), basis_tabs(c, basis), 0, s, np, darena ? jsd : js, tmp_js));
    LAB_HIP(c, launch_crt_to_u64_jobs(t, tmp_js, dst, count, basis_crt(c, basis), basis_nlimbs(c, basis), np, s));
    return LAB_OK;
  };
  auto gram_jobs = [&](size_t out_off, size_t p_off, size_t s_off) -> int {   // out[i][k] = sum_x P[x][i] S[x][k]
    const size_t chunks = gram_chunks(n, r, L, np);
    if (chunks == 1) {
      LAB_HIP(c, launch_gram_tiled(HV(0, out_off), HV(0, p_off), HV(0, s_off), n, r, L, tabs, s, np, js));
    } else {
      LAB_HIP(c, launch_gram_tiled(HV(0, hpart), HV(0, p_off), HV(0, s_off), n, r, L, tabs, s, np, js));
      LAB_HIP(c, launch_sum_terms(HV(0, out_off), HV(0, hpart), chunks, rr, L, tabs, s, np, js, js));
    }
    return LAB_OK;
  };
  auto inner_product_jobs = [&](size_t out_off, size_t a_off, size_t b_off, size_t len) -> int {
    const size_t k = inner_product_splits(len);
    if (k == 1) {
      LAB_HIP(c, launch_matvec_partial(HV(0, out_off), HV(0, a_off), HV(0, b_off), 1, 1, len, L, tabs, s, np, js));
    } else {
      LAB_HIP(c, launch_matvec_partial(HV(0, hpart), HV(0, a_off), HV(0, b_off), k, 1, len / k, L, tabs, s, np, js));
      LAB_HIP(c, launch_sum_terms(HV(0, out_off), HV(0, hpart), k, 1, L, tabs, s, np, js, js));
    }
    return LAB_OK;
  };

  // ================= phase A: transforms
  if (!sd_only) {
    if ((rc = to_hat_jobs(JP(&Job::dS), hS, nr, false))) co_return rc;
    if ((rc = to_hat_jobs(JP(&Job::dphi), hphi, nr, false))) co_return rc;
  }
  if ((rc = to_hat_jobs(JP(&Job::dalpha), hab, 1, false))) co_return rc;
  if ((rc = to_hat_jobs(JP(&Job::dbeta), hab + 1, 1, false))) co_return rc;
  if ((rc = to_hat_jobs(JP(&Job::dc), hc, r, false))) co_return rc;
  // ================= step 1 (src/proofgen.rs:41-49): t_i = A s_i for every column of every proof; A is shared, so the
  // whole batch is one (kappa x n)(n x np*r) product (falls back to one launch per proof for shapes it does not take)
  uint64_t* norm_sums = host_stage + np * (256 + 256 + D);
  bool t_in_dense = false, g_in_digit = false, t_digits_done = false, t_done = false;
  // t (and its digits) of proofs jb .. jb + cnt - 1 from the staging arrays the product left in the dense array
  auto t_from_stages = [&](const MfmaCoefOut& co, size_t jb, size_t cnt, int valid, int basis) -> int {
    JobPtrs pt, pd;
    for (size_t i = 0; i < (size_t)MAX_JOBS; i++) {
      pt.p[i] = i < cnt ? jobs[jb + i].cT : nullptr;
      pd.p[i] = i < cnt ? jobs[jb + i].cTdec : nullptr;
    }
    CrtDigits dg{pd, (uint32_t)in->b1, t1, kappa, q};
    LAB_HIP(c, launch_stage_to_coeffs(co.stage_planes, co.stage_ps, kappa, co.st_rows_pad, co.st_ns, (size_t)valid, co.st_nsub, co.st_jobs, pt, kappa,
                                      basis_tabs(c, basis), basis_crt(c, basis), basis_nlimbs(c, basis), s, dig16 && !t_digits_lazy ? &dg : nullptr));
    t_done = true;
    if (dig16 && !t_digits_lazy) t_digits_done = true;
    return LAB_OK;
  };
  if (a8d) {  // matrix cores, in the basis of B, C, D (see a8d above): S once more into that basis, its norm summed on the way
    LAB_HIPC(c, hipMemsetAsync(norm_sums, 0, np * 8, s));
    LAB_HIPC(c, hipMemsetAsync(above16, 0, np * 4, s));
    // (coefficients above beta_bound are caught by the norm sum and the batch is redone: the transform may assume the bound)
    LAB_HIPC(c, launch_to_hat_fused(JP(&Job::dS), HD(0, hSd), jsd, nr, Ld, tabsd, np, s, (uint64_t)in->beta_bound + 1,
                                   reinterpret_cast<unsigned long long*>(norm_sums), norm_limit, false, above16));
    // the witness tiles of every limb are kept for steps 6-8 as one array over all proofs of the call: [limb][job]
    const bool tiles_paired = np >= 2 && np % 2 == 0 && a8_ns == 32;
    const size_t tile_jobs = tiles_paired ? np / 2 : np;
    const size_t tile_job_bytes = s_tiles ? gram_mfma_tile_bytes(a8d, np) / ((size_t)Ld * tile_jobs) : 0;
    for (size_t jb = 0; jb < np; jb += t_chunk) {       // t_chunk proofs at a time (all of them unless the dense output would be large)
      for (size_t i0 = 0; i0 < r; i0 += 32) {
        const int valid = (int)(r - i0 < 32 ? r - i0 : 32);
        MfmaCoefOut co{dense_tmp + i0 * kappa * D, rk * D, tmp_js_t, kappa, false};   // t is only wanted as coefficients
        if (t_staged) { co.stage_planes = dense_tmp; co.stage_ps = t_stage_ps; }
        MfmaGramOut go{HD(jb, hGd).base, hatd_words, perd * D, r, false};            // step 2 (G = S^T S) from the same tiles
        int8_t* keep = s_tiles ? s_tiles + (jb / (tiles_paired ? 2 : 1)) * tile_job_bytes : nullptr;
        if ((rc = matvec_mfma_impl(c, HD(jb, hTd + i0 * kappa), kappa * D, a8d, HD(jb, hSd + i0), D, a8_ns, valid, mf_scratch, r * D, t_chunk,
                                   perd * D, perd * D, false, &co, nullptr, r <= 32 ? &go : nullptr, keep, tile_job_bytes * tile_jobs))) co_return rc;
        t_in_dense = co.done;
        g_in_digit = go.done;
        if (co.staged && (rc = t_from_stages(co, jb, t_chunk, valid, bdig))) co_return rc;
      }
      if (t_chunk < np && !t_done) {   // a slice of the proofs: its CRT (with the digits of t) before the next slice reuses the dense array
        if (!t_in_dense || !dig16) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: internal: sliced t chain without the dense path");
        JobPtrs pt, pd;
        for (size_t i = 0; i < (size_t)MAX_JOBS; i++) {
          pt.p[i] = i < t_chunk ? jobs[jb + i].cT : nullptr;
          pd.p[i] = i < t_chunk ? jobs[jb + i].cTdec : nullptr;
        }
        CrtDigits dg{pd, (uint32_t)in->b1, t1, kappa, q};
        LAB_HIPC(c, launch_crt_to_u64_jobs(HatView{dense_tmp, rk * D}, tmp_js_t, pt, rk, basis_crt(c, bdig), Ld, t_chunk, s, t_digits_lazy ? nullptr : &dg));
        t_digits_done = true;
      }
    }
  } else if (a8) {   // matrix cores: per proof, its r witness columns as the vector sets (element x of column i at S[x][i])
    // one launch per limb and 32 columns: the proofs are the kernel's jobs and share the A stream through L2
    for (size_t i0 = 0; i0 < r; i0 += 32) {
      const int valid = (int)(r - i0 < 32 ? r - i0 : 32);
      MfmaCoefOut co{dense_tmp + i0 * kappa * D, rk * D, tmp_js_t, kappa, false};
      if (t_staged) { co.stage_planes = dense_tmp; co.stage_ps = t_stage_ps; }
      if ((rc = matvec_mfma_impl(c, HV(0, hT + i0 * kappa), kappa * D, a8, HV(0, hS + i0), D, a8_ns, valid, mf_scratch, r * D, np,
                                 per * D, per * D, false, &co))) co_return rc;
      t_in_dense = co.done;
      if (co.staged && (rc = t_from_stages(co, 0, np, valid, 0))) co_return rc;
    }
  } else {
    hipError_t e = launch_matmul_bcast(HV(0, hT), kappa * D, per * D, in->A->view(in->a_off), HV(0, hS), D, r * D, per * D, np,
                                       kappa, n, r, L, tabs, s);
    if (e == hipErrorNotSupported) {
      for (size_t j = 0; j < np; j++)
        LAB_HIPC(c, launch_matmul_t(HV(j, hT), in->A->view(in->a_off), HV(j, hS), kappa, n, r, L, tabs, s));
    } else {
      LAB_HIPC(c, e);
    }
  }
  if (t_done || (t_in_dense && t_chunk < np)) {
    // (finished above: from the staging arrays, or slice by slice)
  } else if (t_in_dense) {   // the inverse transform already ran from the GEMM's staging array: only the CRT is left
    const int basis = a8d ? bdig : 0;
    // ... and the decomposition of t (its digits in [i][k][rho] order), in the same pass when they have the 16-bit form
    CrtDigits dg{JP(&Job::cTdec), (uint32_t)in->b1, t1, kappa, q};
    hipError_t e = dig16 && !t_digits_lazy ? launch_crt_to_u64_jobs(HatView{dense_tmp, rk * D}, tmp_js_t, JP(&Job::cT), rk, basis_crt(c, basis),
                                                  basis_nlimbs(c, basis), np, s, &dg)
                         : hipErrorNotSupported;
    if (e == hipSuccess) t_digits_done = true;
    else if (e == hipErrorNotSupported)
      LAB_HIPC(c, launch_crt_to_u64_jobs(HatView{dense_tmp, rk * D}, tmp_js_t, JP(&Job::cT), rk, basis_crt(c, basis), basis_nlimbs(c, basis), np, s));
    else LAB_HIPC(c, e);
  } else if ((rc = a8d ? from_hat_jobs(hTd, rk, JP(&Job::cT), true) : from_hat_jobs(hT, rk, JP(&Job::cT), false))) {
    co_return rc;
  }
  // step 2 (:59-70): G = S^T S
  if (sd_only && !g_in_digit) co_return fail(c, LAB_ERR_ARG, "lab_prove_core: internal: the digit-basis Gram product was not taken");
  if (g_in_digit) {
    if ((rc = from_hat_jobs(hGd, rr, JP(&Job::cG), true))) co_return rc;
  } else {
    if ((rc = gram_jobs(hG, hS, hS))) co_return rc;
    if ((rc = from_hat_jobs(hG, rr, JP(&Job::cG), false))) co_return rc;
  }
  // digits of t in [i][k][rho] order, into the basis of B
  if (!t_digits_done && !t_digits_lazy) LAB_HIPC(c, launch_decompose_jobs(JP(&Job::cTdec), JP(&Job::cT), rk, q, (uint64_t)in->b1, t1, kappa, np, s, dig16));
  if (!fuse_t && (rc = to_hat_jobs(JP(&Job::cTdec), hTdec, rk * t1, true, (uint64_t)in->b1, dig16))) co_return rc;

  // ================= step 3 (:101-132) for the whole batch: lhs_j = sum_{i,k} B_ik t_{j,i}^(k); B streamed once per group
  for (size_t j0 = 0; b8 && j0 < np;) {   // matrix cores: up to 64 proofs per pass over B, result written in place
    const size_t left = np - j0;
    const int ns = left > 32 ? 64 : left > 16 ? 32 : left > 8 ? 16 : 8, valid = (int)(left < (size_t)ns ? left : (size_t)ns);
    if (fuse_t) {
      JobPtrs dig;
      for (int i = 0; i < valid; i++) dig.p[i] = jobs[j0 + i].cTdec;
      if ((rc = digits_to_v8_impl(c, b8, dig, ns, valid, reinterpret_cast<int8_t*>(mf_scratch)))) co_return rc;
      if (sh && sh->resident && !size_only) {   // (one group: a sub-batch has at most 64 proofs) the driver multiplies every sub-batch's tiles at once
        sh->groups.push_back(CrsGroup{HD(0, hlhs), perd * D, nullptr, nullptr, 0, ns, valid, mf_scratch});
        co_await std::suspend_always{};
        if (sh->rc) co_return sh->rc;
        break;
      }
    }
    if ((rc = matvec_mfma_impl(c, HD(j0, hlhs), perd * D, b8, HD(j0, hTdec), perd * D, ns, valid, mf_scratch, D, 1, 0, 0, true, nullptr,
                               fuse_t ? reinterpret_cast<const int8_t*>(mf_scratch) : nullptr))) co_return rc;
    j0 += valid;
  }
  if (stream_b) {   // B streamed from the seed: up to 64 proofs per pass over the regenerated rows
    const size_t nmat = r * (size_t)t1;
    // CRS::fetch_B_ik_row (src/structs.rs:78-82): offset of row 0 of B_ik = kappa*n*d + (i*T_1 + k) * kappa_1*kappa, copied literally
    std::vector<uint64_t> offs(2 * nmat);
    for (size_t m = 0; m < nmat; m++) {
      const u128 off = (u128)kappa * n * D + (u128)m * k1 * kappa;
      offs[2 * m] = (uint64_t)off; offs[2 * m + 1] = (uint64_t)(off >> 64);
    }
    // (np <= MAX_JOBS: one group of vector sets)
    const int ns = np > 32 ? 64 : np > 16 ? 32 : np > 8 ? 16 : 8, valid = (int)np;
    JobPtrs dig;
    for (int i = 0; i < MAX_JOBS; i++) dig.p[i] = i < valid ? (t_digits_lazy ? jobs[i].cT : jobs[i].cTdec) : nullptr;
    CrsGroup grp{HD(0, hlhs), perd * D, t_digits_lazy ? nullptr : &dig, nullptr, 0, ns, valid, mf_scratch};
    if (t_digits_lazy) { grp.t64 = &dig; grp.dt = DigitsFromT{(uint32_t)in->b1, (uint32_t)t1, kappa, 0, q}; }
    if (sh) {   // the driver streams B once for every sub-batch of the call
      sh->groups.push_back(grp);
      co_await std::suspend_always{};
      if (sh->rc) co_return sh->rc;
    } else {
      CrsStream* cs = nullptr;
      if ((rc = crs_stream_create(c, bdig, nmat, k1, kappa, ns, &cs, rhs_terms))) co_return rc;
      rc = crs_stream_run(cs, in->crs_seed, offs.data(), nmat, &grp, 1);
      if (rc == LAB_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(c, LAB_ERR_HIP, "lab_prove_core: streamed outer commitment failed");
      crs_stream_destroy(cs);
      if (rc) co_return rc;
    }
  }
  for (size_t j0 = 0; !b8 && !stream_b && j0 < np;) {
    int g = 1;
    while (g < 8 && j0 + 2 * (size_t)g <= np) g *= 2;
    if (g == 1) {
      if ((rc = matvec_partial_any(c, HD(j0, hpartd), in->B, in->b_off, HD(j0, hTdec), r * (size_t)t1, k1, kappa))) co_return rc;
    } else {
      if ((rc = matvec_partial_batched_any(c, g, HD(j0, hpartd), perd * D, in->B, in->b_off, HD(j0, hTdec), perd * D,
                                           r * (size_t)t1, k1, kappa))) co_return rc;
    }
    for (int b = 0; b < g; b++)
      LAB_HIPC(c, launch_sum_terms(HD(j0 + b, hlhs), HD(j0 + b, hpartd), r * (size_t)t1, k1, Ld, tabsd, s));
    j0 += g;
  }

  // groups of 8/4/2/1 proofs for the kernels that stream shared CRS data
  auto for_groups = [&](auto&& fn) -> int {
    for (size_t j0 = 0; j0 < np;) {
      int g = 1;
      while (g < 8 && j0 + 2 * (size_t)g <= np) g *= 2;
      int e = fn(j0, g);
      if (e) return e;
      j0 += g;
    }
    return LAB_OK;
  };

  // ================= step 4 (:137-148): rhs = sum_{i<=j,k} g_ij^(k) C_ijk ; u_1 = lhs + rhs (:153)
  LAB_HIPC(c, launch_gather_upper_jobs(JP(&Job::cGup), JP(&Job::cG), r, np, s));
  LAB_HIPC(c, launch_decompose_jobs(JP(&Job::cGdec), JP(&Job::cGup), pairs, q, (uint64_t)in->b2, t2, 1, np, s, dig16));  // [pair][k]
  if ((rc = to_hat_jobs(JP(&Job::cGdec), hGdec, pairs * t2, true, (uint64_t)in->b2, dig16))) co_return rc;
  // sum_m digits[m] * Cm[m][rho] for every proof: on the matrix cores when Cm was packed for them (up to 64 proofs per
  // pass), else with the multiply-add kernels in groups of 8
  auto commit_digits = [&](const lab_mat8* m8, const lab_vec* Cm, size_t cm_off, size_t out_off, size_t dig_off, size_t nterms) -> int {
    if (m8) {
      for (size_t j0 = 0; j0 < np;) {
        const size_t left = np - j0;
        const int ns = left > 32 ? 64 : left > 16 ? 32 : left > 8 ? 16 : 8, valid = (int)(left < (size_t)ns ? left : (size_t)ns);
        int e = matvec_mfma_impl(c, HD(j0, out_off), perd * D, m8, HD(j0, dig_off), perd * D, ns, valid, mf_scratch, D, 1, 0, 0);
        if (e) return e;
        j0 += valid;
      }
      return LAB_OK;
    }
    return for_groups([&](size_t j0, int g) {   // Cm streamed once per group
      return axpy_commit_batched_any(c, g, HD(j0, out_off), perd * D, HD(j0, dig_off), perd * D, Cm, cm_off, nterms, k2,
                                     HD(j0, hpartd), perd * D);
    });
  };
  if ((rc = commit_digits(c8, in->C, in->c_off, hrhs, hGdec, pairs * t2))) co_return rc;
  LAB_HIPC(c, launch_pointwise(PW_ADD, HD(0, hlhs), HD(0, hlhs), HD(0, hrhs), k1, Ld, tabsd, s, np, jsd, jsd, jsd));
  if ((rc = from_hat_jobs(hlhs, k1, JP(&Job::cu1), true))) co_return rc;

  // ================= step 5 (:161-186, 429-456): JL projection over the integers
  // (one pass over Pi also produces W, the Pi part of phi'' used in step 6)
  LAB_HIPC(c, hipMemsetAsync(proj_all, 0, np * 256 * 8, s));
  if (in->Pi2)   // 2 bits per entry; with the digit-basis step 1 the witness is inside its norm bound (coefficients < 2^24) or the batch is redone
    LAB_HIPC(c, launch_pi_fused2_jobs(JP(&Job::dproj_int), JP(&Job::cW), JP(&Job::dPi2), JP(&Job::dS), JP(&Job::domega), n, r, q, np,
                                     a8d != nullptr, s, a8d ? above16 : nullptr));
  else
    LAB_HIPC(c, launch_pi_fused_jobs(JP(&Job::dproj_int), JP(&Job::cW), JP(&Job::dPi), JP(&Job::dS), JP(&Job::domega), n, r, q, np, s));
  LAB_HIPC(c, launch_lift_i64_jobs(JP(&Job::dproj), JP(&Job::dproj_int), 256, q, np, s));

  // ================= steps 6-8 (:205-289, :301-314, :320-358).  phi''_i = psi phi'_i + W_i and phi_i = alpha phi'_i + beta phi''_i
  // are only ever used inside inner products with the witness, which are linear in them:
  //   phi'^T S =: X1,  W^T S =: XW           two Gram products on operands already in the transform domain
  //   phi''^T S = psi X1 + XW =: X2          (r x r polynomials, coefficient domain)
  //   sum_i <phi''_i, s_i> = trace(X2)       the second term of b''
  //   Phi^T S = alpha X1 + beta X2           step 8's product
  // so the n*r polynomials phi'' and phi are never formed.  (Gram results are brought back to [0, q) before they are
  // multiplied again: see the header comment on CRT range.)
  JobU64 psi;
  {
    JobPtrs none;
    for (size_t j = 0; j < np; j++) { psi.v[j] = ins[j]->psi % q; none.p[j] = nullptr; }
    LAB_HIPC(c, launch_coef_elementwise_jobs(JP(&Job::capp), JP(&Job::da), none, rr, q, psi, 0, np, s));   // a'' = psi a'
  }
  if ((rc = to_hat_jobs(JP(&Job::capp), happ, rr, false))) co_return rc;
  if (!sd_only && (rc = to_hat_jobs(JP(&Job::cW), hphipp, nr, false))) co_return rc;   // W in the transform domain
  if ((rc = to_hat_jobs(JP(&Job::cG), hGc, rr, false))) co_return rc;
  if ((rc = inner_product_jobs(hbpp, happ, hGc, rr))) co_return rc;               // sum a''_ij g_ij
  if ((rc = from_hat_jobs(hbpp, 1, JP(&Job::cab), false))) co_return rc;
  if (sd_only) {   // both Gram products on the matrix cores, in the basis of the witness tiles
    const HatView xo[2] = {HD(0, hX1d), HD(0, hXWd)};
    const JobPtrs xp[2] = {JP(&Job::dphi), JP(&Job::cW)};
    if ((rc = gram_mfma_impl(c, a8d, 2, xo, xp, s_tiles, jsd, n, r, np, mf_scratch))) co_return rc;
    if ((rc = from_hat_jobs(hX1d, rr, JP(&Job::cX1), true))) co_return rc;
    if ((rc = from_hat_jobs(hXWd, rr, JP(&Job::cX), true))) co_return rc;        // XW
  } else {
    if ((rc = gram_jobs(hX1, hphi, hS))) co_return rc;
    if ((rc = gram_jobs(hX2, hphipp, hS))) co_return rc;
    if ((rc = from_hat_jobs(hX1, rr, JP(&Job::cX1), false))) co_return rc;
    if ((rc = from_hat_jobs(hX2, rr, JP(&Job::cX), false))) co_return rc;           // XW
  }
  LAB_HIPC(c, launch_coef_elementwise_jobs(JP(&Job::cX2), JP(&Job::cX1), JP(&Job::cX), rr, q, psi, 1, np, s));   // X2 = psi X1 + XW
  LAB_HIPC(c, launch_add_trace_jobs(JP(&Job::cbpp), JP(&Job::cab), JP(&Job::cX2), r, q, np, s));                  // b''
  if ((rc = to_hat_jobs(JP(&Job::cX1), hX1, rr, false))) co_return rc;
  if ((rc = to_hat_jobs(JP(&Job::cX2), hX2, rr, false))) co_return rc;
  LAB_HIPC(c, launch_poly_by_vec(HV(0, hX1), HV(0, hab), HV(0, hX1), rr, L, tabs, s, np, js));
  LAB_HIPC(c, launch_poly_by_vec(HV(0, hX2), HV(0, hab + 1), HV(0, hX2), rr, L, tabs, s, np, js));
  LAB_HIPC(c, launch_pointwise(PW_ADD, HV(0, hX), HV(0, hX1), HV(0, hX2), rr, L, tabs, s, np, js, js, js));
  if ((rc = from_hat_jobs(hX, rr, JP(&Job::cX), false))) co_return rc;
  LAB_HIPC(c, launch_symmetrize_half_jobs(JP(&Job::cH), JP(&Job::cX), r, q, inv2_mod(q), np, s));
  // digits of the upper triangle of H for step 9
  LAB_HIPC(c, launch_gather_upper_jobs(JP(&Job::cHup), JP(&Job::cH), r, np, s));
  LAB_HIPC(c, launch_decompose_jobs(JP(&Job::cHdec), JP(&Job::cHup), pairs, q, (uint64_t)in->b1, t1, 1, np, s, dig16));
  if ((rc = to_hat_jobs(JP(&Job::cHdec), hHdec, pairs * t1, true, (uint64_t)in->b1, dig16))) co_return rc;

  // ================= step 9 (:364-378): u_2 = sum_{i<=j,k} h_ij^(k) D_ijk ; D streamed once per group
  if ((rc = commit_digits(d8, in->Dm, in->d_off, hu2, hHdec, pairs * t1))) co_return rc;
  if ((rc = from_hat_jobs(hu2, k2, JP(&Job::cu2), true))) co_return rc;

  // ================= step 10 (:380-399): z = sum_i c_i s_i
  if (sd_only) {
    if ((rc = to_hat_jobs(JP(&Job::dc), hcd, r, true))) co_return rc;
    LAB_HIPC(c, launch_matvec_partial(HD(0, hzd), HD(0, hSd), HD(0, hcd), 1, n, r, Ld, tabsd, s, np, jsd));
    if ((rc = from_hat_jobs(hzd, n, JP(&Job::cz), true))) co_return rc;
  } else {
    LAB_HIPC(c, launch_matvec_partial(HV(0, hz), HV(0, hS), HV(0, hc), 1, n, r, L, tabs, s, np, js));
    if ((rc = from_hat_jobs(hz, n, JP(&Job::cz), false))) co_return rc;
  }

  // ================= copy-out and the two verifier callbacks that can reject
  const size_t hst_words = np * (256 + 256 + D + 1);
  if (c->pinned_words < hst_words) {   // page-locked: the copy is one DMA, not a staged pageable transfer
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr; c->pinned_words = 0;
    LAB_HIPC(c, hipHostMalloc(reinterpret_cast<void**>(&c->pinned), hst_words * 8, hipHostMallocDefault));
    c->pinned_words = hst_words;
  }
  uint64_t* const hst = c->pinned;
  const int64_t* hproj = reinterpret_cast<const int64_t*>(hst);
  const uint64_t* homega = hst + np * 256;
  const uint64_t* hbpp_st = hst + np * 512;
  LAB_HIPC(c, launch_gather_jobs(host_stage + np * 256, JP(&Job::domega), 256, np, s));
  LAB_HIPC(c, launch_gather_jobs(host_stage + np * 512, JP(&Job::cbpp), D, np, s));
  {  // integer projections to device-resident caller buffers (host ones are filled from the staged copy below)
    JobPtrs pi;
    bool any = false;
    for (size_t j = 0; j < np; j++) {
      void* u = outs[j]->projection_int;
      pi.p[j] = (u && is_device_pointer(u)) ? u : nullptr;
      any = any || pi.p[j];
    }
    if (any) LAB_HIPC(c, launch_scatter_jobs(pi, host_stage, 256, np, s));
  }
  for (size_t j = 0; j < np; j++) {
    Job& J = jobs[j];
    lab_transcript* o = outs[j];
    struct { uint64_t* dst; const uint64_t* src; size_t words; } cp[] = {
        {o->u_1, J.cu1, k1 * D}, {o->projection, J.dproj, 256}, {o->b_prime_prime, J.cbpp, D}, {o->u_2, J.cu2, k2 * D},
        {o->z, J.cz, n * D}, {o->t_i_all, J.cT, rk * D}, {o->g_mat, J.cG, rr * D}, {o->h_mat, J.cH, rr * D}};
    for (auto& e : cp)   // results produced in place (device-resident caller buffers) need no copy
      if (e.dst && e.dst != e.src) LAB_HIPC(c, hipMemcpyAsync(e.dst, e.src, e.words * 8, hipMemcpyDefault, s));
  }
  LAB_HIPC(c, hipMemcpyAsync(hst, host_stage, hst_words * 8, hipMemcpyDeviceToHost, s));
  LAB_HIPC(c, hipStreamSynchronize(s));
  if (a8d) {   // a witness above the declared norm bound: step 1 may have left the CRT range; redo the batch in the main basis
    const uint64_t* hnorm = hst + np * (256 + 256 + D);
    bool over = false;
    for (size_t j = 0; j < np; j++) over = over || hnorm[j] > norm_limit;
    if (over) {
      // (a sub-batch that shares the scratch allocation with others leaves the redo to the driver: the main-basis run needs more
      // workspace than its region, and the others are still using theirs)
      if (!own_scratch) co_return REDO_IN_MAIN_BASIS;
      ProveTask redo = prove_batch_co(c, np, ins, outs, status, false, nullptr, true, 0, nullptr);
      const int e = redo.done() ? redo.rc() : fail(c, LAB_ERR_HIP, "lab_prove_core: internal: the main-basis run did not finish");
      c->prove_flags |= 2;
      co_return e;
    }
  }
  c->prove_flags = a8d ? 1 : 0;
  for (size_t j = 0; j < np; j++) {
    void* u = outs[j]->projection_int;
    if (u && !is_device_pointer(u)) memcpy(u, hproj + j * 256, 256 * 8);
  }

  int first_fail = LAB_OK;
  for (size_t j = 0; j < np; j++) {
    const lab_prove_input* x = ins[j];
    int stj = LAB_OK;
    // Verifier::valid_projection (src/verification.rs:568-579): sqrt(f64(sum p^2)) <= sqrt(128) * beta
    __int128 ss = 0;
    for (int m = 0; m < 256; m++) ss += (__int128)hproj[j * 256 + m] * hproj[j * 256 + m];
    const double norm = sqrt((double)ss);
    if (!(norm <= sqrt(128.0) * (double)x->beta_bound)) {
      stj = fail(c, LAB_ERR_JL_REJECTED, "JL projection rejected (proof %zu of the batch): norm %.6g > %.6g", j, norm,
                 sqrt(128.0) * (double)x->beta_bound);
    } else {
      // verify_b_prime_prime (src/verification.rs:532-551): b''[0] == <omega, projection> + psi * b'
      u128 chk = 0;
      for (int m = 0; m < 256; m++) {
        __int128 pm = hproj[j * 256 + m] % (__int128)q;
        if (pm < 0) pm += q;
        chk = (chk + (u128)(homega[j * 256 + m] % q) * (u128)pm) % q;
      }
      chk = (chk + (u128)(x->psi % q) * (u128)(x->b_prime % q)) % q;
      if ((uint64_t)chk != hbpp_st[j * D])
        stj = fail(c, LAB_ERR_BPP_CHECK, "verify_b_prime_prime check failed: b_prime_prime_k constant coefficient is: %llu, and check_val is: %llu",
                   (unsigned long long)hbpp_st[j * D], (unsigned long long)(uint64_t)chk);
    }
    if (status) status[j] = stj;
    if (stj != LAB_OK && first_fail == LAB_OK) first_fail = stj;
  }
  co_return status ? LAB_OK : first_fail;
}

static int prove_batch_impl(lab_ctx* c, size_t np, const lab_prove_input* const* ins, lab_transcript* const* outs, int* status,
                            bool allow_digit_ajtai) {
  if (!c) return LAB_ERR_ARG;
  if (np == 0) return LAB_OK;
  PtrCacheScope ptr_cache;
  if (np <= (size_t)MAX_JOBS) {
    ProveTask t = prove_batch_co(c, np, ins, outs, status, allow_digit_ajtai, nullptr, true, 0, nullptr);
    return t.done() ? t.rc() : fail(c, LAB_ERR_HIP, "lab_prove_core: internal: the call did not finish");
  }
  const size_t nsub = (np + MAX_JOBS - 1) / MAX_JOBS;
  auto sub_np = [&](size_t k) { return k + 1 < nsub ? (size_t)MAX_JOBS : np - k * MAX_JOBS; };
  const lab_prove_input* in = ins ? ins[0] : nullptr;
  // (every sub-batch checks that its own proofs share dimensions, constants and the CRS; across sub-batches:)
  for (size_t k = 1; k < nsub && in; k++) {
    const lab_prove_input* x = ins[k * MAX_JOBS];
    if (!x) return fail(c, LAB_ERR_ARG, "lab_prove_core: null argument");
    if (x->n != in->n || x->r != in->r || x->kappa != in->kappa || x->kappa1 != in->kappa1 || x->kappa2 != in->kappa2 || x->t1 != in->t1 ||
        x->t2 != in->t2 || x->b1 != in->b1 || x->b2 != in->b2 || x->beta_bound != in->beta_bound || x->A != in->A || x->B != in->B ||
        !same_seed(x->crs_seed, in->crs_seed) || x->B8 != in->B8 || x->A8 != in->A8 || x->C8 != in->C8 || x->D8 != in->D8 ||
        x->A8d != in->A8d || x->C != in->C || x->Dm != in->Dm || x->a_off != in->a_off || x->b_off != in->b_off ||
        x->c_off != in->c_off || x->d_off != in->d_off)
      return fail(c, LAB_ERR_ARG, "lab_prove_core_batch: all proofs of a batch must share dimensions, constants and the CRS");
  }
  const bool shared_stream = in && !in->B8 && !in->B && in->crs_seed && (in->C || in->C8) && nsub <= (size_t)CRS_MAX_GROUPS;
  // B resident as byte planes: the sub-batches share one pass over it as the jobs of one launch per limb (up to 16 of them)
  const bool shared_res = in && in->B8 && nsub <= 16;
  const bool shared = shared_stream || shared_res;
  if (!shared) {   // B resident as residues: every sub-batch streams it from HBM itself
    int first = LAB_OK;
    for (size_t k = 0; k < nsub; k++) {
      const size_t j0 = k * MAX_JOBS;
      ProveTask t = prove_batch_co(c, sub_np(k), ins + j0, outs + j0, status ? status + j0 : nullptr, allow_digit_ajtai, nullptr, true, 0, nullptr);
      const int e = t.done() ? t.rc() : LAB_ERR_HIP;
      if (e != LAB_OK && first == LAB_OK) first = e;
      if (e != LAB_OK && e != LAB_ERR_JL_REJECTED && e != LAB_ERR_BPP_CHECK) return e;
    }
    return first;
  }
  // ---- B streamed: one pass over the regenerated rows for all sub-batches
  std::vector<size_t> off(nsub + 1, 0);
  ProveShared sh;
  for (size_t k = 0; k < nsub; k++) {
    const size_t j0 = k * MAX_JOBS;
    size_t need = 0;
    ProveTask t = prove_batch_co(c, sub_np(k), ins + j0, outs + j0, nullptr, allow_digit_ajtai, &sh, false, 0, &need);
    if (!t.done() || t.rc() != LAB_OK) return t.done() ? t.rc() : LAB_ERR_HIP;
    off[k + 1] = off[k] + ((need + 255) & ~(size_t)255);
  }
  sh.dense_off = off[nsub];
  sh.resident = shared_res;
  sh.stage_off = sh.dense_off + sh.dense_bytes;
  const size_t stage_bytes = shared_res ? mfma_stage_bytes(in->B8, MAX_JOBS, nsub) : 0;
  int rc = ensure_scratch(c, sh.stage_off + stage_bytes);
  if (rc) return rc;
  sh.groups.reserve(nsub);
  std::vector<ProveTask> tasks;
  tasks.reserve(nsub);
  std::vector<int> early(nsub, 0);      // (resident B) a sub-batch that went through without waiting for the shared launch
  int first = LAB_OK, flags = 0;
  for (size_t k = 0; k < nsub; k++) {   // steps 1-2 of every sub-batch
    const size_t j0 = k * MAX_JOBS;
    tasks.emplace_back(prove_batch_co(c, sub_np(k), ins + j0, outs + j0, status ? status + j0 : nullptr, allow_digit_ajtai, &sh, false, off[k], nullptr));
    if (!tasks.back().done()) continue;
    const int e = tasks.back().rc();
    if (shared_res && (e == LAB_OK || e == LAB_ERR_JL_REJECTED || e == LAB_ERR_BPP_CHECK || e == REDO_IN_MAIN_BASIS)) {
      early[k] = 1;                      // its digits do not have the tile form: it multiplied by itself (finished, or to be redone)
      continue;
    }
    sh.rc = e != LAB_OK ? e : LAB_ERR_ARG;   // failed before step 3: let the suspended ones leave
    for (size_t i = 0; i + 1 < tasks.size(); i++)
      if (!tasks[i].done()) tasks[i].resume();
    return sh.rc;
  }
  if (shared_res) {
    // full groups that sit equally spaced in the allocation go through one launch per limb as its jobs; the others (a last, shorter
    // sub-batch: the launch stores every job's sets up to ONE valid count) one by one
    const lab_mat8* b8 = in->B8;
    const size_t ng = sh.groups.size();
    size_t nmulti = 0;
    if (ng >= 2) {
      const ptrdiff_t ds = static_cast<char*>(sh.groups[1].scratch) - static_cast<char*>(sh.groups[0].scratch);
      const ptrdiff_t dout = sh.groups[1].out.base - sh.groups[0].out.base;
      nmulti = 1;
      while (nmulti < ng && ds > 0 && dout > 0 && sh.groups[nmulti].ns == sh.groups[0].ns && sh.groups[nmulti].ns_valid == sh.groups[0].ns_valid &&
             sh.groups[nmulti].out.plane_stride == sh.groups[0].out.plane_stride &&
             static_cast<char*>(sh.groups[nmulti].scratch) - static_cast<char*>(sh.groups[0].scratch) == (ptrdiff_t)nmulti * ds &&
             sh.groups[nmulti].out.base - sh.groups[0].out.base == (ptrdiff_t)nmulti * dout)
        nmulti++;
      if (nmulti < 2 || sh.groups[0].ns != MAX_JOBS) nmulti = 0;
      if (nmulti) {
        const CrsGroup& G = sh.groups[0];
        const MfmaJobBuffers jb{(size_t)ds, reinterpret_cast<uint32_t*>(static_cast<char*>(c->scratch) + sh.stage_off)};
        rc = matvec_mfma_impl(c, G.out, G.out_set_stride, b8, ConstHatView(), 0, G.ns, G.ns_valid, G.scratch, D, nmulti, (size_t)dout, 0, true, nullptr,
                              static_cast<const int8_t*>(G.scratch), nullptr, nullptr, 0, &jb);
        flags |= 4;
      }
    }
    for (size_t g = nmulti; g < ng && rc == LAB_OK; g++) {
      const CrsGroup& G = sh.groups[g];
      rc = matvec_mfma_impl(c, G.out, G.out_set_stride, b8, ConstHatView(), 0, G.ns, G.ns_valid, G.scratch, D, 1, 0, 0, true, nullptr,
                            static_cast<const int8_t*>(G.scratch));
    }
    sh.rc = rc;
  } else {
    const size_t r = in->r, kappa = in->kappa, k1 = in->kappa1, nmat = r * (size_t)in->t1, pairs = r * (r + 1) / 2;
    std::vector<uint64_t> offs(2 * nmat);
    for (size_t m = 0; m < nmat; m++) {   // CRS::fetch_B_ik_row (src/structs.rs:78-82), as in the body
      const u128 o = (u128)kappa * in->n * D + (u128)m * k1 * kappa;
      offs[2 * m] = (uint64_t)o; offs[2 * m + 1] = (uint64_t)(o >> 64);
    }
    int ns_max = 8;
    for (const CrsGroup& g : sh.groups) ns_max = g.ns > ns_max ? g.ns : ns_max;
    CrsStream* cs = nullptr;
    rc = crs_stream_create(c, in->C ? in->C->basis : mat8_basis(in->C8), nmat, k1, kappa, ns_max, &cs, (uint64_t)pairs * in->t2, sh.groups.size());
    if (rc == LAB_OK) {
      rc = crs_stream_run(cs, in->crs_seed, offs.data(), nmat, sh.groups.data(), sh.groups.size());
      if (sh.groups.size() >= 2) flags |= 4;
      if (rc == LAB_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, LAB_ERR_HIP, "lab_prove_core: streamed outer commitment failed");
      crs_stream_destroy(cs);
    }
    sh.rc = rc;
  }
  std::vector<size_t> redo;
  for (size_t k = 0; k < nsub; k++) {     // steps 4-10 of every sub-batch
    if (!early[k]) tasks[k].resume();
    const int e = tasks[k].done() ? tasks[k].rc() : LAB_ERR_HIP;
    if (e == REDO_IN_MAIN_BASIS) { redo.push_back(k); continue; }
    flags |= c->prove_flags;
    if (e != LAB_OK && first == LAB_OK) first = e;
  }
  tasks.clear();
  // a sub-batch with a witness above its norm bound: once more on the main-basis kernels, alone (B regenerated for it alone), now
  // that nobody else holds a region of the scratch allocation
  for (size_t k : redo) {
    if (sh.rc) break;
    const size_t j0 = k * MAX_JOBS;
    ProveTask t = prove_batch_co(c, sub_np(k), ins + j0, outs + j0, status ? status + j0 : nullptr, false, nullptr, true, 0, nullptr);
    const int e = t.done() ? t.rc() : LAB_ERR_HIP;
    flags |= 2;
    if (e != LAB_OK && first == LAB_OK) first = e;
  }
  c->prove_flags = flags;
  return sh.rc ? sh.rc : first;
}

extern "C" int lab_prove_core(lab_ctx* c, const lab_prove_input* in, lab_transcript* out) {
  if (!c) return LAB_ERR_ARG;
  if (!in || !out) return fail(c, LAB_ERR_ARG, "lab_prove_core: null argument");
  return prove_batch_impl(c, 1, &in, &out, nullptr);
}

extern "C" int lab_prove_core_batch(lab_ctx* c, size_t nproofs, const lab_prove_input* const* in, lab_transcript* const* out,
                                    int* status) {
  return prove_batch_impl(c, nproofs, in, out, status);
}
