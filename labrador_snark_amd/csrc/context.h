// context.h — host-side state behind the opaque lab_ctx / lab_vec handles.
// Internal to the shared library.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/labrador_hip.h"
#include "kernels.h"

struct lab_mat8;

namespace lab {
// Phases of the streamed outer commitment (crs_stream.hip) the measurement hook samples: HIP events around a bounded sample of
// the launches of each phase (a kappa = 2^14 step has ~10^5 expansion launches), the totals counted for every launch.
enum ProfPhaseId { PH_EXPAND = 0 /* k_crs_expand: units = coefficients */, PH_TILES = 1 /* k_coefs_to_a8 / transform + pack: coefficients */,
                   PH_GEMM = 2 /* k_gemm_i8*: units = algorithmic bytes */, PH_DIGITS = 3 /* k_digits_to_v8: digits */, PH_COUNT = 4 };
struct ProfPhase {
  struct Sample { hipEvent_t a, b; double units; };
  std::vector<Sample> ev;      // [0, used): recorded samples; the rest: created events waiting for reuse
  size_t used = 0;
  uint64_t launches = 0, stride = 1;
  double units = 0.0;          // over ALL launches
};
constexpr size_t PROF_PHASE_CAP = 1024;
}  // namespace lab

struct lab_ctx {
  uint64_t q = 0;
  int device = 0;
  int nlimbs = 0;
  bool direct = false;             // q itself is the single NTT prime
  bool wide = false;               // ... and 2^30 <= q < 2^32: the strict-32 kernels (LimbTable.wide)
  uint32_t primes[lab::MAX_LIMBS] = {0, 0, 0, 0};
  uint64_t max_len = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  lab::LimbTable* d_tabs = nullptr;   // [nlimbs] on device
  lab::CrtTable* d_crt = nullptr;
  lab::LimbTable h_tabs[lab::MAX_LIMBS];
  lab::CrtTable h_crt;
  // growable device scratch, carved per call
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  uint64_t* pinned = nullptr;      // page-locked host words for the small device-to-host copies that end a call (grown on demand)
  size_t pinned_words = 0;
  mutable std::string err;
  // digit basis (lab_ctx_enable_digit_basis): RNS basis for reductions whose vector operand is a decomposition
  // digit (< dg_vmax): dg_nlimbs limbs = a prefix of the main primes plus, when dg_small, one prime < 2^16 whose
  // matrix residues can be stored as u16 ("compact" vectors)
  int dg_nlimbs = 0;
  bool dg_small = false;
  uint32_t dg_primes[lab::MAX_LIMBS] = {0, 0, 0, 0};
  uint64_t dg_terms = 0, dg_vmax = 0;
  lab::LimbTable* dg_tabs = nullptr;
  lab::CrtTable* dg_crt = nullptr;
  lab::LimbTable dg_h_tabs[lab::MAX_LIMBS];
  lab::CrtTable dg_h_crt;
  // measurement hook (lab_ctx_profile_begin/end): HIP events around the mat-vec kernel
  bool prof_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  double prof_bytes = 0.0;
  int gemm_diag = 0;              // lab_ctx_gemm_diag (include/labrador_hip_diag.h): diagnostic variant of the timed GEMM
  uint64_t* gemm_diag_buf = nullptr;
  size_t gemm_diag_words = 0;
  std::string prof_kernel;        // kernel the last profiled mat-vec launch was dispatched to
  lab::ProfPhase prof_phase[lab::PH_COUNT];
  int prove_flags = 0;            // last lab_prove_core[_batch]: 1 = step 1 ran in the digit basis, 2 = tried and redone in the main basis
};

struct lab_vec {
  lab_ctx* ctx = nullptr;
  size_t count = 0;
  int basis = 0;           // 0 = main basis, 1 = digit basis
  bool compact = false;    // digit basis only: the small last limb is stored as u16 in d16
  uint32_t* d = nullptr;   // [planes32][count][64]
  uint16_t* d16 = nullptr; // [count][64] when compact
  int nlimbs() const { return basis ? ctx->dg_nlimbs : ctx->nlimbs; }
  int planes32() const { return nlimbs() - (compact ? 1 : 0); }
  const lab::LimbTable* tabs() const { return basis ? ctx->dg_tabs : ctx->d_tabs; }
  const lab::CrtTable* crt() const { return basis ? ctx->dg_crt : ctx->d_crt; }
  lab::HatView view(size_t off) const { return lab::HatView{d + off * lab::D, count * lab::D}; }
};

namespace lab {

int fail(const lab_ctx* ctx, int code, const char* fmt, ...);
// ensures ctx->scratch holds at least `bytes`; contents are not preserved
int ensure_scratch(lab_ctx* ctx, size_t bytes);
bool is_device_pointer(const void* p);
// remembers is_device_pointer() answers for the lifetime of the outermost scope object on this thread
struct PtrCacheScope {
  bool owner;
  PtrCacheScope();
  ~PtrCacheScope();
};

#define LAB_HIP(ctx, expr)                                                                         \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return lab::fail(ctx, LAB_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// Brackets the launches issued during its lifetime on c->stream as one sample of phase `ph` (when the hook is on and the
// phase's sampling stride selects this launch); `units` are counted either way.
struct PhaseSample {
  lab_ctx* c;
  ProfPhase* p = nullptr;
  PhaseSample(lab_ctx* ctx, int ph, double units) : c(ctx) {
    if (!c->prof_on) return;
    ProfPhase& P = c->prof_phase[ph];
    P.units += units;
    const bool take = (P.launches++ % P.stride) == 0;
    if (!take) return;
    if (P.used == PROF_PHASE_CAP) {   // keep every other sample, double the stride: a uniform sample of a run of any length
      for (size_t i = 1; 2 * i < P.used; i++) std::swap(P.ev[i], P.ev[2 * i]);
      P.used = (P.used + 1) / 2;
      P.stride *= 2;
      if ((P.launches - 1) % P.stride) return;
    }
    if (P.used == P.ev.size()) {
      ProfPhase::Sample sm{nullptr, nullptr, 0.0};
      if (hipEventCreate(&sm.a) != hipSuccess || hipEventCreate(&sm.b) != hipSuccess) return;
      P.ev.push_back(sm);
    }
    P.ev[P.used].units = units;
    if (hipEventRecord(P.ev[P.used].a, c->stream) == hipSuccess) p = &P;
  }
  ~PhaseSample() {
    if (p && hipEventRecord(p->ev[p->used].b, c->stream) == hipSuccess) p->used++;
  }
};

// carve helper: sequential 256-byte aligned sub-allocations out of ctx->scratch
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base(static_cast<char*>(b)) {}
  template <typename T>
  T* take(size_t n) {
    T* r = reinterpret_cast<T*>(base + off);
    off += (n * sizeof(T) + 255) & ~size_t(255);
    return r;
  }
  static size_t padded(size_t bytes) { return (bytes + 255) & ~size_t(255); }
};

}  // namespace lab

namespace lab {

// Host/device staging for boundary pointers.  Usage: sum need() over all
// pointers (+ temporaries), ensure_scratch, then in()/out() carve and copy, and
// finish() copies results back (synchronising only if something went to the host).
struct Staging {
  lab_ctx* c;
  Carver* cv;
  struct Pending { void* host; const void* dev; size_t bytes; };
  Pending pend[16];
  int npend = 0;
  Staging(lab_ctx* ctx, Carver* carver) : c(ctx), cv(carver) {}
  static size_t need(const void* p, size_t bytes) { return (p && !is_device_pointer(p)) ? Carver::padded(bytes) : 0; }
  template <typename T>
  int in(const T* p, size_t n, const T** dev) {
    if (!p || is_device_pointer(p)) { *dev = p; return LAB_OK; }
    T* slot = cv->take<T>(n);
    LAB_HIP(c, hipMemcpyAsync(slot, p, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *dev = slot;
    return LAB_OK;
  }
  template <typename T>
  T* out(T* p, size_t n) {
    if (!p || is_device_pointer(p)) return p;
    T* slot = cv->take<T>(n);
    pend[npend++] = Pending{p, slot, n * sizeof(T)};
    return slot;
  }
  int finish() {
    if (!npend) return LAB_OK;
    for (int i = 0; i < npend; i++)
      LAB_HIP(c, hipMemcpyAsync(pend[i].host, pend[i].dev, pend[i].bytes, hipMemcpyDeviceToHost, c->stream));
    LAB_HIP(c, hipStreamSynchronize(c->stream));
    return LAB_OK;
  }
};

// out[i*r+k] = sum_x P[x*r+i]*S[x*r+k]; `partial` must hold gram_num_chunks(n)*r*r elements (unused when 1 chunk)
int gram_impl(lab_ctx* c, HatView out, ConstHatView P, ConstHatView S, size_t n, size_t r, HatView partial);
int check_range(lab_ctx* c, const lab_vec* v, size_t off, size_t n, const char* what);
// int8-MFMA outer commitment (mfma_api.hip)
size_t mfma_scratch_bytes(const lab_mat8* m, int ns, size_t njobs);
// algorithmic bytes of the GEMM launches of all limbs of `m` against ns sets: digit planes of the k-tiles that hold terms, the
// B-operand tiles once, the results
double mfma_gemm_bytes(const lab_mat8* m, int ns, int ns_valid);
// `coef`: the caller only wants the results as coefficients — when the launch goes through the slot-major staging array
// (several jobs) the inverse transform reads it directly and writes plane l of the dense array `coef->dense`
// ([limb][count][64] words, job stride coef->job_stride, set s -> polynomial s * coef->set_stride + row) instead of `out`;
// coef->done tells whether that happened (else `out` holds the transform-domain results as usual)
struct MfmaCoefOut {
  uint32_t* dense; size_t plane_stride, job_stride, set_stride; bool done;
  // optional: room for the staging array of EVERY limb (stage_ps words each).  When the call goes through staging arrays that fit,
  // it leaves them there (staged = true, their geometry below) and runs no inverse transform: the caller finishes with
  // launch_stage_to_coeffs (transform + CRT in one pass, no dense copy)
  uint32_t* stage_planes = nullptr; size_t stage_ps = 0;
  bool staged = false; size_t st_rows_pad = 0, st_ns = 0, st_nsub = 0, st_jobs = 0;
};
// `gram`: the sets are the (at most 32) witness columns of each job and every limb has 3 planes — also produce
// G[i][k] = sum_x V_i[x] V_k[x] per source job from the packed tiles (transform domain, basis of m): element i*r + k of
// gram->out + job * gram->job_stride, plane l at + l * gram->plane_stride; gram->done tells whether that happened
struct MfmaGramOut { uint32_t* out; size_t plane_stride, job_stride, r; bool done; };
// `jb`: the prepacked tiles of job b live in a buffer of their own at prepacked + b * job_stride (each laid out [limb] as
// digits_to_v8_impl writes it for one job) and the slot-major staging array of the results is `stage`
// (mfma_stage_bytes(m, ns, njobs)) — the groups of a streamed pass, whose tiles sit in their sub-batches' workspaces
struct MfmaJobBuffers { size_t job_stride; uint32_t* stage; };
size_t mfma_stage_bytes(const lab_mat8* m, int ns, size_t njobs);
int matvec_mfma_impl(lab_ctx* c, HatView out, size_t out_set_stride, const lab_mat8* m, ConstHatView V, size_t v_set_stride, int ns,
                     int ns_valid, void* scratch, size_t v_k_stride, size_t njobs, size_t out_job_stride, size_t v_job_stride,
                     bool timed = false, MfmaCoefOut* coef = nullptr, const int8_t* prepacked = nullptr,
                     MfmaGramOut* gram = nullptr, int8_t* keep_tiles = nullptr, size_t keep_limb_stride = 0,
                     const MfmaJobBuffers* jb = nullptr);
// X_a = P_a^T S per proof on the matrix cores (mfma_api.hip): P_a as [n][r] coefficient matrices, the witness as the tiles
// matvec_mfma_impl(keep_tiles) made of it for the Ajtai product
bool gram_mfma_ok(const lab_mat8* m, size_t n, size_t r);
size_t gram_mfma_tile_bytes(const lab_mat8* m, size_t njobs);
size_t gram_mfma_scratch_bytes(const lab_mat8* m, size_t njobs);
int gram_mfma_impl(lab_ctx* c, const lab_mat8* m, int nops, const HatView* outs, const JobPtrs* Pcoefs, const int8_t* s_tiles,
                   size_t out_job_stride, size_t n, size_t r, size_t njobs, void* scratch);
// B-operand tiles of every limb straight from 16-bit decomposition digits (to_hat + packing fused, mfma_kernels.hip);
// hand the buffer to matvec_mfma_impl as `prepacked` (njobs = 1)
size_t mfma_prepacked_bytes(const lab_mat8* m, int ns);
int digits_to_v8_impl(lab_ctx* c, const lab_mat8* m, const JobPtrs& src16, int ns, int ns_valid, int8_t* dst, const DigitsFromT* dt = nullptr);
uint32_t mat8_min_prime(const lab_ctx* c, const lab_mat8* m);
// the outer commitment with the matrices streamed from the CRS seed (crs_stream.hip)
struct CrsStream;
// extra_terms: terms the caller adds to the result in the transform domain (the prover's rhs of u_1) — they count against the
// exactness bound of the basis together with the streamed sum
size_t crs_stream_scratch_bytes(lab_ctx* c, int basis, size_t nmat, size_t rows, size_t cols, int ns, uint64_t extra_terms = 0);
// matrices per K chunk of the streamed sum and whether the chunks are added mod q in the coefficient domain (the whole sum exceeds
// the exactness bound of the basis, one chunk does not); false: not even one matrix row fits
bool crs_stream_plan(const lab_ctx* c, int basis, size_t nmat, size_t cols, uint64_t extra_terms, size_t* mc, bool* reduce);
// ngroups: groups of vector sets one pass over the regenerated rows will serve (each with its own results and tile scratch)
constexpr int CRS_MAX_GROUPS = 4;
int crs_stream_create(lab_ctx* c, int basis, size_t nmat, size_t rows, size_t cols, int ns, CrsStream** out, uint64_t extra_terms = 0,
                      size_t ngroups = 1);
void crs_stream_destroy(CrsStream* st);
// one group of up to 64 vector sets: results (set s at out.base + s*out_set_stride), the sets as 16-bit digits (dig->p[s]:
// [nmat*cols][64] words) or as transform-domain vectors (V, set stride v_set_stride), padded / valid set counts, and the group's
// own crs_stream_scratch_bytes() of tile scratch
struct CrsGroup {
  HatView out; size_t out_set_stride;
  const JobPtrs* dig; const ConstHatView* V; size_t v_set_stride;
  int ns, ns_valid;
  void* scratch;
  // instead of dig: the sets' 64-bit coefficient vectors whose digits are the operand (t64->p[s], see DigitsFromT; dt.m0 is set per chunk)
  const JobPtrs* t64 = nullptr;
  DigitsFromT dt{};
};
int crs_stream_run(CrsStream* st, const uint8_t seed[32], const uint64_t* mat_offsets, size_t nmat, const CrsGroup* groups, size_t ngroups);
uint32_t basis_min_prime(const lab_ctx* c, int basis);
bool mat8_matches(const lab_mat8* m, int basis, size_t nmat, size_t rows, size_t cols);
int mat8_basis(const lab_mat8* m);
// launch_matvec_partial wrapped in the profiling events when the hook is active
int matvec_partial_timed(lab_ctx* c, HatView partial, ConstHatView M, ConstHatView V, size_t nmat, size_t rows, size_t cols,
                         int nlimbs = 0, const LimbTable* tabs = nullptr);
// hat -> coefficient conversion of `count` elements of `src` into device memory `dcoeffs`, using `tmp` (>= count elements of hat scratch)
int from_hat(lab_ctx* c, ConstHatView src, size_t count, uint32_t* tmp, uint64_t* dcoeffs, int basis = 0);
int to_hat(lab_ctx* c, const uint64_t* dcoeffs, HatView dst, size_t count, int basis = 0);
inline int basis_nlimbs(const lab_ctx* c, int basis) { return basis ? c->dg_nlimbs : c->nlimbs; }
inline const LimbTable* basis_tabs(const lab_ctx* c, int basis) { return basis ? c->dg_tabs : c->d_tabs; }
inline const CrtTable* basis_crt(const lab_ctx* c, int basis) { return basis ? c->dg_crt : c->d_crt; }
// sum_m <M[m][rho,:], V[m]> partials with M possibly compact: all limbs of the matrix's basis
int matvec_partial_any(lab_ctx* c, HatView partial, const lab_vec* M, size_t m_off, ConstHatView V, size_t nmat,
                       size_t rows, size_t cols);
int matvec_partial_batched_any(lab_ctx* c, int nb, HatView partial, size_t partial_bs, const lab_vec* M, size_t m_off,
                               ConstHatView V, size_t v_bs, size_t nmat, size_t rows, size_t cols);
size_t axpy_partial_elems(size_t nterms, size_t len);
int axpy_commit_batched_any(lab_ctx* c, int nb, HatView out, size_t out_bs, ConstHatView G, size_t g_bs, const lab_vec* Cm,
                            size_t c_off, size_t nterms, size_t len, HatView partial, size_t partial_bs);
int axpy_commit_any(lab_ctx* c, HatView out, ConstHatView G, const lab_vec* Cm, size_t c_off, size_t nterms, size_t len,
                    HatView partial);

}  // namespace lab
