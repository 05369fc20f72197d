// capi.hip — implementation of the C ABI in include/labrador_hip.h: context
// construction (limb primes, twiddle and CRT tables), device vectors, and the
// per-operation entry points that sequence the gfx950 kernels on the context's
// stream.  There is NO CPU fallback: without a GPU lab_ctx_create fails with
// LAB_ERR_NO_DEVICE.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include <unordered_map>

#include "context.h"

using namespace lab;
typedef unsigned __int128 u128;

// ------------------------------------------------------------- utilities
namespace lab {

int fail(const lab_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  // a failed HIP call (an allocation that did not fit) also parks its code in the thread's "last error"; the launchers return
  // hipGetLastError() after every kernel launch and would report it again for an innocent launch of a later call
  if (code == LAB_ERR_HIP) (void)hipGetLastError();
  return code;
}

int ensure_scratch(lab_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return LAB_OK;
  LAB_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->scratch) LAB_HIP(ctx, hipFree(ctx->scratch));
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  size_t want = bytes + std::min<size_t>(bytes / 8, (size_t)1 << 30) + (1u << 20);   // some slack for the next, slightly larger call
  LAB_HIP(ctx, hipMalloc(&ctx->scratch, want));
  ctx->scratch_bytes = want;
  return LAB_OK;
}

// hipPointerGetAttributes costs about a microsecond; a batch of 64 proofs asks about ~1700 pointers (every input and
// output, several times), which was 2 ms of GPU idle time per call.  Inside a PtrCacheScope (one API call: nothing is
// allocated or freed by the caller meanwhile) answers are remembered per address.
static thread_local std::unordered_map<const void*, bool>* g_ptr_cache = nullptr;
PtrCacheScope::PtrCacheScope() : owner(g_ptr_cache == nullptr) {
  if (owner) g_ptr_cache = new std::unordered_map<const void*, bool>();
}
PtrCacheScope::~PtrCacheScope() {
  if (owner) { delete g_ptr_cache; g_ptr_cache = nullptr; }
}

bool is_device_pointer(const void* p) {
  if (g_ptr_cache) {
    auto it = g_ptr_cache->find(p);
    if (it != g_ptr_cache->end()) return it->second;
  }
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  bool dev = false;
  if (e != hipSuccess) (void)hipGetLastError();  // plain malloc memory: clear the sticky error
  else dev = attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
  if (g_ptr_cache) (*g_ptr_cache)[p] = dev;
  return dev;
}

}  // namespace lab

// ------------------------------------------------------------- number theory (host)
static uint64_t powmod(uint64_t b, uint64_t e, uint64_t m) {
  u128 r = 1, x = b % m;
  while (e) {
    if (e & 1) r = r * x % m;
    x = x * x % m;
    e >>= 1;
  }
  return (uint64_t)r;
}
static uint64_t invmod_prime(uint64_t a, uint64_t p) { return powmod(a % p, p - 2, p); }

static bool is_prime_u64(uint64_t n) {
  if (n < 2) return false;
  static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (uint64_t s : small) {
    if (n % s == 0) return n == s;
  }
  uint64_t d = n - 1;
  int r = 0;
  while ((d & 1) == 0) { d >>= 1; r++; }
  for (uint64_t a : small) {
    uint64_t x = powmod(a, d, n);
    if (x == 1 || x == n - 1) continue;
    bool comp = true;
    for (int i = 1; i < r; i++) {
      x = (uint64_t)((u128)x * x % n);
      if (x == n - 1) { comp = false; break; }
    }
    if (comp) return false;
  }
  return true;
}

static uint32_t brv6(uint32_t k) {
  uint32_t r = 0;
  for (int i = 0; i < LOGD; i++) r |= ((k >> i) & 1u) << (LOGD - 1 - i);
  return r;
}
static uint32_t shoup(uint32_t w, uint32_t p) { return (uint32_t)(((uint64_t)w << 32) / p); }

// NTT-friendly primes below 2^30 (p == 1 mod 128), largest first.
static const uint32_t kLimbPrimes[MAX_LIMBS] = {1073741441u, 1073739649u, 1073738753u, 1073736449u};

// wide: 2^30 <= p < 2^32 (the single-limb strict-32 path, zq_device.h Zp<true>): constants of the transforms in Montgomery form
static bool build_limb_table(uint32_t p, LimbTable* t, bool wide = false) {
  memset(t, 0, sizeof *t);
  t->p = p;
  t->p2 = wide ? 0u : 2 * p;
  t->wide = wide ? 1u : 0u;
  // p^-1 mod 2^32 by Newton iteration
  uint32_t inv = p;
  for (int i = 0; i < 5; i++) inv *= 2u - p * inv;
  t->pinv = inv;
  t->r1 = (uint32_t)(((uint64_t)1 << 32) % p);
  t->r2 = (uint32_t)((u128)t->r1 * t->r1 % p);
  t->q_hi_mul = t->r1;
  t->q_hi_mul_sh = shoup(t->r1, p);
  t->inv_p = 1.0 / (double)p;
  // primitive 128th root of unity: psi^64 == -1
  uint64_t psi = 0;
  for (uint64_t g = 2; g < 2000; g++) {
    uint64_t c = powmod(g, (p - 1) / (2 * D), p);
    if (powmod(c, D, p) == p - 1) { psi = c; break; }
  }
  if (!psi) return false;
  uint64_t psi_inv = invmod_prime(psi, p);
  for (uint32_t k = 0; k < (uint32_t)D; k++) {
    uint32_t w = (uint32_t)powmod(psi, brv6(k), p), wi = (uint32_t)powmod(psi_inv, brv6(k), p);
    if (wide) { w = (uint32_t)(((u128)w << 32) % p); wi = (uint32_t)(((u128)wi << 32) % p); }
    t->fwd[k][0] = w;  t->fwd[k][1] = shoup(w, p);
    t->inv[k][0] = wi; t->inv[k][1] = shoup(wi, p);
  }
  uint64_t ninv = invmod_prime(D, p);
  if (wide) {   // n^-1 * 2^32 mod p; the R^e variants of the lazy path do not exist here
    t->scale_lo[0][0] = (uint32_t)(((u128)ninv << 32) % p);
    return true;
  }
  uint64_t re = 1;
  for (int e = 0; e < 4; e++) {
    uint32_t lo = (uint32_t)((u128)ninv * re % p);
    uint32_t hi = (uint32_t)((u128)lo * t->inv[1][0] % p);
    t->scale_lo[e][0] = lo; t->scale_lo[e][1] = shoup(lo, p);
    t->scale_hi[e][0] = hi; t->scale_hi[e][1] = shoup(hi, p);
    re = (uint64_t)((u128)re * t->r1 % p);
  }
  return true;
}

static void build_crt_table(uint64_t q, int nlimbs, const uint32_t* primes, CrtTable* t) {
  memset(t, 0, sizeof *t);
  t->q = q;
  t->nlimbs = nlimbs;
  u128 M = 1;
  for (int a = 0; a < nlimbs; a++) {
    t->p[a] = primes[a];
    t->prefix_mod_q[a] = (uint64_t)(M % q);
    M *= primes[a];
    for (int b = 0; b < a; b++) {
      uint32_t g = (uint32_t)invmod_prime(primes[b] % primes[a], primes[a]);
      t->ginv[a][b][0] = g;
      t->ginv[a][b][1] = shoup(g, primes[a]);
    }
  }
  t->m_mod_q = (uint64_t)(M % q);
  t->inv_q = 1.0 / (double)q;
  for (int a = 0; a < nlimbs; a++) t->inv_p[a] = 1.0 / (double)primes[a];
  u128 h = (M - 1) / 2;
  for (int a = 0; a < nlimbs; a++) {
    t->half_digits[a] = (uint32_t)(h % primes[a]);
    h /= primes[a];
  }
  u128 worst = primes[0];
  for (int a = 1; a < nlimbs; a++) worst += (u128)(primes[a] - 1) * (q - 1);
  t->lazy_sum = worst < ((u128)1 << 63) ? 1u : 0u;
  bool one_size = nlimbs == 3 && t->lazy_sum && q >= (1ull << 24);
  for (int a = 0; a < nlimbs && one_size; a++) {
    if (primes[a] >= (1u << 30)) one_size = false;
    for (int b = 0; b < a; b++)
      if (primes[b] >= 2 * (uint64_t)primes[a]) one_size = false;
  }
  t->fast3 = one_size ? 1u : 0u;
}

// ------------------------------------------------------------- context
extern "C" int lab_ctx_create(uint64_t q, int device, uint64_t max_len, lab_ctx** out) {
  return lab_ctx_create_ex(q, device, max_len, 0, out);
}

extern "C" int lab_ctx_create_ex(uint64_t q, int device, uint64_t max_len, int flags, lab_ctx** out) {
  if (!out) return LAB_ERR_ARG;
  *out = nullptr;
  if (q < 3 || (q & 1) == 0 || q >= (1ull << 40)) return LAB_ERR_ARG;
  (void)hipGetLastError();   // whatever an earlier user of the thread left behind is not this context's
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return LAB_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) return LAB_ERR_ARG;
  lab_ctx* c = new lab_ctx();
  c->q = q;
  c->device = device;
  if (max_len == 0) max_len = 65536;
  if (flags & ~LAB_CTX_RNS_ONLY) { delete c; return LAB_ERR_ARG; }
  if (!(flags & LAB_CTX_RNS_ONLY) && q < (1ull << 32) && q % (2 * D) == 1 && is_prime_u64(q)) {
    // q is its own NTT prime: one limb, 4 bytes per coefficient, no CRT.  Below 2^30 on the lazy kernels; in [2^30, 2^32)
    // on the strict-32 instantiations (LimbTable.wide)
    c->direct = true;
    c->wide = q >= (1ull << 30);
    c->nlimbs = 1;
    c->primes[0] = (uint32_t)q;
    c->max_len = 1ull << 40;  // arithmetic is mod q itself: any length is exact
  } else {
    u128 M = 1;
    u128 unit = (u128)D * (u128)(q - 1) * (u128)(q - 1);
    for (int L = 1; L <= MAX_LIMBS; L++) {
      M *= kLimbPrimes[L - 1];
      u128 cap = ((M - 1) / 2) / unit;
      if (cap >= max_len) {
        c->nlimbs = L;
        c->max_len = cap > (u128)(1ull << 62) ? (1ull << 62) : (uint64_t)cap;
        break;
      }
    }
    if (!c->nlimbs) { delete c; return LAB_ERR_RANGE; }
    for (int l = 0; l < c->nlimbs; l++) c->primes[l] = kLimbPrimes[l];
  }
  for (int l = 0; l < c->nlimbs; l++)
    if (!build_limb_table(c->primes[l], &c->h_tabs[l], c->wide)) { delete c; return LAB_ERR_ARG; }
  build_crt_table(c->q, c->nlimbs, c->primes, &c->h_crt);

  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) { c->own_stream = true; e = hipMalloc(&c->d_tabs, sizeof(LimbTable) * c->nlimbs); }
  if (e == hipSuccess) e = hipMalloc(&c->d_crt, sizeof(CrtTable));
  if (e == hipSuccess) e = hipMemcpy(c->d_tabs, c->h_tabs, sizeof(LimbTable) * c->nlimbs, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(c->d_crt, &c->h_crt, sizeof(CrtTable), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    lab_ctx_destroy(c);
    return LAB_ERR_HIP;
  }
  limb_tables_register(c->d_tabs, c->wide);
  *out = c;
  return LAB_OK;
}

extern "C" void lab_ctx_destroy(lab_ctx* c) {
  if (!c) return;
  if (c->d_tabs) limb_tables_register(c->d_tabs, false);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto& ev : c->prof_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  for (auto& ph : c->prof_phase)
    for (auto& sm : ph.ev) { (void)hipEventDestroy(sm.a); (void)hipEventDestroy(sm.b); }
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->d_tabs) (void)hipFree(c->d_tabs);
  if (c->d_crt) (void)hipFree(c->d_crt);
  if (c->dg_tabs) (void)hipFree(c->dg_tabs);
  if (c->dg_crt) (void)hipFree(c->dg_crt);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int lab_ctx_info(const lab_ctx* c, lab_info* out) {
  if (!c || !out) return LAB_ERR_ARG;
  out->q = c->q;
  out->nlimbs = c->nlimbs;
  for (int l = 0; l < 4; l++) out->primes[l] = l < c->nlimbs ? c->primes[l] : 0;
  out->max_len = c->max_len;
  out->device = c->device;
  return LAB_OK;
}

// candidates for the compact last limb: NTT-friendly primes below 2^16 (p == 1 mod 128), ascending
static std::vector<uint32_t> small_ntt_primes() {
  std::vector<uint32_t> v;
  for (uint32_t p = 2 * D + 1; p < 65536; p += 2 * D)
    if (is_prime_u64(p)) v.push_back(p);
  return v;
}

extern "C" int lab_ctx_enable_digit_basis(lab_ctx* c, uint64_t max_terms, uint64_t digit_bound) {
  if (!c) return LAB_ERR_ARG;
  if (max_terms == 0 || digit_bound < 2 || digit_bound > c->q) return fail(c, LAB_ERR_ARG, "lab_ctx_enable_digit_basis: bad bounds");
  if (c->direct) return fail(c, LAB_ERR_ARG, "lab_ctx_enable_digit_basis: q is its own NTT prime; the main basis is already minimal");
  if (c->dg_nlimbs) return fail(c, LAB_ERR_ARG, "lab_ctx_enable_digit_basis: already enabled");
  // smallest storage (4 B per full limb + 2 B for a small one) with M/2 >= max_terms*64*(q-1)*(digit_bound-1)
  const u128 need = (u128)max_terms * D * (u128)(c->q - 1) * (u128)(digit_bound - 1);
  const std::vector<uint32_t> small_primes = small_ntt_primes();
  int best_bytes = 1 << 30, best_full = -1;
  uint32_t best_small = 0;
  for (int nfull = 0; nfull <= c->nlimbs; nfull++) {
    u128 M = 1;
    for (int l = 0; l < nfull; l++) M *= c->primes[l];
    if ((M - 1) / 2 >= need) {
      if (nfull > 0 && 4 * nfull < best_bytes) { best_bytes = 4 * nfull; best_full = nfull; best_small = 0; }
      continue;
    }
    if (nfull == MAX_LIMBS) break;
    for (uint32_t sp : small_primes)
      if ((M * sp - 1) / 2 >= need) {
        if (4 * nfull + 2 < best_bytes) { best_bytes = 4 * nfull + 2; best_full = nfull; best_small = sp; }
        break;
      }
  }
  if (best_full < 0 || best_bytes >= 4 * c->nlimbs)
    return fail(c, LAB_ERR_RANGE, "lab_ctx_enable_digit_basis: no digit basis smaller than the main basis for these bounds");
  c->dg_nlimbs = best_full + (best_small ? 1 : 0);
  c->dg_small = best_small != 0;
  for (int l = 0; l < best_full; l++) c->dg_primes[l] = c->primes[l];
  if (best_small) c->dg_primes[best_full] = best_small;
  c->dg_terms = max_terms;
  c->dg_vmax = digit_bound - 1;
  for (int l = 0; l < c->dg_nlimbs; l++)
    if (!build_limb_table(c->dg_primes[l], &c->dg_h_tabs[l])) { c->dg_nlimbs = 0; return fail(c, LAB_ERR_ARG, "digit basis: no root of unity"); }
  build_crt_table(c->q, c->dg_nlimbs, c->dg_primes, &c->dg_h_crt);
  LAB_HIP(c, hipSetDevice(c->device));
  LAB_HIP(c, hipMalloc(&c->dg_tabs, sizeof(LimbTable) * c->dg_nlimbs));
  LAB_HIP(c, hipMalloc(&c->dg_crt, sizeof(CrtTable)));
  LAB_HIP(c, hipMemcpy(c->dg_tabs, c->dg_h_tabs, sizeof(LimbTable) * c->dg_nlimbs, hipMemcpyHostToDevice));
  LAB_HIP(c, hipMemcpy(c->dg_crt, &c->dg_h_crt, sizeof(CrtTable), hipMemcpyHostToDevice));
  return LAB_OK;
}

// The digit basis for the matrix-core path: primes below 2^23 - 0x8080, whose residues are three signed bytes — 9
// instead of 16 byte products per modular product and 3 instead of 4 bytes of matrix per limb and coefficient.
extern "C" int lab_ctx_enable_digit_basis_mfma(lab_ctx* c, uint64_t max_terms, uint64_t digit_bound) {
  if (!c) return LAB_ERR_ARG;
  if (max_terms == 0 || digit_bound < 2 || digit_bound > c->q) return fail(c, LAB_ERR_ARG, "lab_ctx_enable_digit_basis_mfma: bad bounds");
  if (c->dg_nlimbs) return fail(c, LAB_ERR_ARG, "lab_ctx_enable_digit_basis_mfma: a digit basis is already enabled");
  const u128 need = (u128)max_terms * D * (u128)(c->q - 1) * (u128)(digit_bound - 1);
  // NTT-friendly (p == 1 mod 128) primes, descending from the largest residue three signed digits can hold
  const uint32_t top = (1u << 23) - 0x8080;
  int L = 0;
  u128 M = 1;
  for (uint32_t p = top / (2 * D) * (2 * D) + 1; p > (1u << 22) && L < MAX_LIMBS; p -= 2 * D) {
    if (p > top || !is_prime_u64(p)) continue;
    c->dg_primes[L++] = p;
    M *= p;
    if ((M - 1) / 2 >= need) break;
  }
  if ((M - 1) / 2 < need) {
    for (int l = 0; l < MAX_LIMBS; l++) c->dg_primes[l] = 0;
    return fail(c, LAB_ERR_RANGE, "lab_ctx_enable_digit_basis_mfma: %d primes below 2^23 do not cover these bounds", MAX_LIMBS);
  }
  c->dg_nlimbs = L;
  c->dg_small = false;
  c->dg_terms = max_terms;
  c->dg_vmax = digit_bound - 1;
  for (int l = 0; l < c->dg_nlimbs; l++)
    if (!build_limb_table(c->dg_primes[l], &c->dg_h_tabs[l])) { c->dg_nlimbs = 0; return fail(c, LAB_ERR_ARG, "digit basis: no root of unity"); }
  build_crt_table(c->q, c->dg_nlimbs, c->dg_primes, &c->dg_h_crt);
  LAB_HIP(c, hipSetDevice(c->device));
  LAB_HIP(c, hipMalloc(&c->dg_tabs, sizeof(LimbTable) * c->dg_nlimbs));
  LAB_HIP(c, hipMalloc(&c->dg_crt, sizeof(CrtTable)));
  LAB_HIP(c, hipMemcpy(c->dg_tabs, c->dg_h_tabs, sizeof(LimbTable) * c->dg_nlimbs, hipMemcpyHostToDevice));
  LAB_HIP(c, hipMemcpy(c->dg_crt, &c->dg_h_crt, sizeof(CrtTable), hipMemcpyHostToDevice));
  return LAB_OK;
}

extern "C" int lab_ctx_digit_info(const lab_ctx* c, lab_info* out) {
  if (!c || !out) return LAB_ERR_ARG;
  out->q = c->q;
  out->nlimbs = c->dg_nlimbs;
  for (int l = 0; l < 4; l++) out->primes[l] = l < c->dg_nlimbs ? c->dg_primes[l] : 0;
  out->max_len = c->dg_terms;
  out->device = c->dg_small ? 1 : 0;  // reused field: 1 = last limb is a u16-storable prime
  return LAB_OK;
}

extern "C" const char* lab_last_error(const lab_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" int lab_ctx_set_stream(lab_ctx* c, void* hip_stream) {
  if (!c) return LAB_ERR_ARG;
  LAB_HIP(c, hipStreamSynchronize(c->stream));
  if (c->own_stream && c->stream) LAB_HIP(c, hipStreamDestroy(c->stream));
  c->stream = static_cast<hipStream_t>(hip_stream);
  c->own_stream = false;
  return LAB_OK;
}

extern "C" int lab_ctx_synchronize(lab_ctx* c) {
  if (!c) return LAB_ERR_ARG;
  LAB_HIP(c, hipStreamSynchronize(c->stream));
  return LAB_OK;
}

// ------------------------------------------------------------- measurement hook
extern "C" int lab_ctx_profile_begin(lab_ctx* c) {
  if (!c) return LAB_ERR_ARG;
  LAB_HIP(c, hipStreamSynchronize(c->stream));
  c->prof_on = true;
  c->prof_used = 0;
  c->prof_bytes = 0.0;
  for (auto& ph : c->prof_phase) { ph.used = 0; ph.launches = 0; ph.stride = 1; ph.units = 0.0; }
  return LAB_OK;
}

extern "C" int lab_ctx_profile_phase(lab_ctx* c, int phase, double* sampled_ms, uint64_t* sampled_launches, double* sampled_units,
                                     uint64_t* launches, double* units) {
  if (!c) return LAB_ERR_ARG;
  if (phase < 0 || phase >= PH_COUNT) return fail(c, LAB_ERR_ARG, "lab_ctx_profile_phase: phase %d (0..%d)", phase, PH_COUNT - 1);
  LAB_HIP(c, hipStreamSynchronize(c->stream));
  const ProfPhase& P = c->prof_phase[phase];
  double ms = 0.0, u = 0.0;
  for (size_t i = 0; i < P.used; i++) {
    float t = 0.f;
    LAB_HIP(c, hipEventElapsedTime(&t, P.ev[i].a, P.ev[i].b));
    ms += t;
    u += P.ev[i].units;
  }
  if (sampled_ms) *sampled_ms = ms;
  if (sampled_launches) *sampled_launches = P.used;
  if (sampled_units) *sampled_units = u;
  if (launches) *launches = P.launches;
  if (units) *units = P.units;
  return LAB_OK;
}

extern "C" const char* lab_ctx_profile_kernel(const lab_ctx* c) {
  return c ? c->prof_kernel.c_str() : "";
}

extern "C" int lab_ctx_gemm_diag(lab_ctx* c, int mode, uint64_t* ticks, size_t words) {
  if (!c) return LAB_ERR_ARG;
  if (mode < 0 || mode > 4) return fail(c, LAB_ERR_ARG, "lab_ctx_gemm_diag: mode %d (0..4)", mode);
  LAB_HIP(c, hipStreamSynchronize(c->stream));
  c->gemm_diag = mode;
  c->gemm_diag_buf = mode ? ticks : nullptr;
  c->gemm_diag_words = mode && ticks ? words : 0;
  return LAB_OK;
}

extern "C" int lab_ctx_prove_flags(const lab_ctx* c) { return c ? c->prove_flags : 0; }

extern "C" int lab_ctx_profile_end(lab_ctx* c, double* kernel_ms, uint64_t* launches, double* algorithmic_bytes) {
  if (!c) return LAB_ERR_ARG;
  LAB_HIP(c, hipStreamSynchronize(c->stream));
  double ms = 0.0;
  for (size_t i = 0; i < c->prof_used; i++) {
    float t = 0.f;
    LAB_HIP(c, hipEventElapsedTime(&t, c->prof_events[i].first, c->prof_events[i].second));
    ms += t;
  }
  if (kernel_ms) *kernel_ms = ms;
  if (launches) *launches = c->prof_used;
  if (algorithmic_bytes) *algorithmic_bytes = c->prof_bytes;
  c->prof_on = false;
  return LAB_OK;
}

int lab::matvec_partial_timed(lab_ctx* c, HatView partial, ConstHatView M, ConstHatView V, size_t nmat, size_t rows, size_t cols,
                              int nlimbs, const LimbTable* tabs) {
  if (nlimbs <= 0) { nlimbs = c->nlimbs; tabs = c->d_tabs; }
  if (!c->prof_on) {
    LAB_HIP(c, launch_matvec_partial(partial, M, V, nmat, rows, cols, nlimbs, tabs, c->stream));
    return LAB_OK;
  }
  if (c->prof_used == c->prof_events.size()) {
    hipEvent_t a, b;
    LAB_HIP(c, hipEventCreate(&a));
    LAB_HIP(c, hipEventCreate(&b));
    c->prof_events.emplace_back(a, b);
  }
  auto& ev = c->prof_events[c->prof_used++];
  LAB_HIP(c, hipEventRecord(ev.first, c->stream));
  LAB_HIP(c, launch_matvec_partial(partial, M, V, nmat, rows, cols, nlimbs, tabs, c->stream));
  LAB_HIP(c, hipEventRecord(ev.second, c->stream));
  c->prof_kernel = matvec_kernel_name();
  c->prof_bytes += (double)nlimbs * 256.0 * ((double)nmat * rows * cols + (double)nmat * cols + (double)nmat * rows);
  return LAB_OK;
}

// ------------------------------------------------------------- staging helpers
// Returns a device pointer holding `bytes` of `p`: `p` itself if it is device
// memory, otherwise a copy placed in `stage` (device scratch).
static int stage_in(lab_ctx* c, const void* p, void* stage, size_t bytes, const void** dev) {
  if (is_device_pointer(p)) { *dev = p; return LAB_OK; }
  LAB_HIP(c, hipMemcpyAsync(stage, p, bytes, hipMemcpyHostToDevice, c->stream));
  *dev = stage;
  return LAB_OK;
}

// ------------------------------------------------------------- polymul seam
extern "C" int lab_negacyclic_polymul_u64(lab_ctx* c, uint64_t* prod, const uint64_t* lhs, const uint64_t* rhs,
                                          size_t batch) {
  if (!c || (batch && (!prod || !lhs || !rhs))) return fail(c, LAB_ERR_ARG, "lab_negacyclic_polymul_u64: null argument");
  if (batch == 0) return LAB_OK;
  LAB_HIP(c, hipSetDevice(c->device));
  const size_t words = batch * D, L = c->nlimbs;
  const size_t hat_bytes = Carver::padded(L * words * 4), u64_bytes = Carver::padded(words * 8);
  int rc = ensure_scratch(c, 2 * hat_bytes + 3 * u64_bytes);
  if (rc) return rc;
  Carver cv(c->scratch);
  uint32_t* ha = cv.take<uint32_t>(L * words);
  uint32_t* hb = cv.take<uint32_t>(L * words);
  uint64_t* sa = cv.take<uint64_t>(words);
  uint64_t* sb = cv.take<uint64_t>(words);
  uint64_t* so = cv.take<uint64_t>(words);
  const void *da, *db;
  if ((rc = stage_in(c, lhs, sa, words * 8, &da))) return rc;
  if ((rc = stage_in(c, rhs, sb, words * 8, &db))) return rc;
  HatView va{ha, words}, vb{hb, words};
  LAB_HIP(c, launch_reduce_u64((const uint64_t*)da, va, batch, c->nlimbs, c->d_tabs, c->stream));
  LAB_HIP(c, launch_reduce_u64((const uint64_t*)db, vb, batch, c->nlimbs, c->d_tabs, c->stream));
  LAB_HIP(c, launch_ntt(true, va, va, batch, c->nlimbs, c->d_tabs, 0, c->stream));
  LAB_HIP(c, launch_ntt(true, vb, vb, batch, c->nlimbs, c->d_tabs, 0, c->stream));
  LAB_HIP(c, launch_pointwise(PW_MUL_KEEP_SCALE, va, va, vb, batch, c->nlimbs, c->d_tabs, c->stream));
  LAB_HIP(c, launch_ntt(false, va, va, batch, c->nlimbs, c->d_tabs, 1, c->stream));  // scale n^-1 * 2^32 undoes the REDC factor
  const bool out_dev = is_device_pointer(prod);
  uint64_t* dout = out_dev ? prod : so;
  LAB_HIP(c, launch_crt_to_u64(va, dout, batch, c->d_crt, c->nlimbs, c->stream));
  if (!out_dev) {
    LAB_HIP(c, hipMemcpyAsync(prod, so, words * 8, hipMemcpyDeviceToHost, c->stream));
    LAB_HIP(c, hipStreamSynchronize(c->stream));
  }
  return LAB_OK;
}

// ------------------------------------------------------------- raw NTT (config 2)
static int ntt_inplace(lab_ctx* c, uint32_t* dev, size_t batch, bool fwd) {
  if (!c || (batch && !dev)) return fail(c, LAB_ERR_ARG, "lab_ntt: null argument");
  if (batch == 0) return LAB_OK;
  if (!is_device_pointer(dev)) return fail(c, LAB_ERR_ARG, "lab_ntt: residues must be device memory");
  LAB_HIP(c, hipSetDevice(c->device));
  HatView v{dev, batch * D};
  LAB_HIP(c, launch_ntt(fwd, v, v, batch, c->nlimbs, c->d_tabs, 0, c->stream));
  return LAB_OK;
}
extern "C" int lab_ntt_forward(lab_ctx* c, uint32_t* dev, size_t batch) { return ntt_inplace(c, dev, batch, true); }
extern "C" int lab_ntt_inverse(lab_ctx* c, uint32_t* dev, size_t batch) { return ntt_inplace(c, dev, batch, false); }

// ------------------------------------------------------------- vectors
static int vec_create(lab_ctx* c, size_t count, int basis, bool compact, lab_vec** out) {
  if (!c || !out || count == 0) return fail(c, LAB_ERR_ARG, "lab_vec_create: bad argument");
  *out = nullptr;
  if (basis && !c->dg_nlimbs) return fail(c, LAB_ERR_ARG, "lab_vec_create_digit: call lab_ctx_enable_digit_basis first");
  if (compact && !(basis && c->dg_small)) return fail(c, LAB_ERR_ARG, "lab_vec_create_digit: this digit basis has no u16 limb");
  LAB_HIP(c, hipSetDevice(c->device));
  lab_vec* v = new lab_vec();
  v->ctx = c;
  v->count = count;
  v->basis = basis;
  v->compact = compact;
  // one allocation: the u32 planes, then (compact) the u16 plane
  const size_t b32 = (size_t)v->planes32() * count * D * sizeof(uint32_t);
  const size_t b16 = compact ? count * D * sizeof(uint16_t) : 0;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, b32 + b16 + 256);
  if (e != hipSuccess) {
    delete v;
    return fail(c, LAB_ERR_HIP, "hipMalloc(%zu elements, %zu bytes): %s", count, b32 + b16, hipGetErrorString(e));
  }
  v->d = static_cast<uint32_t*>(p);
  v->d16 = compact ? reinterpret_cast<uint16_t*>(static_cast<char*>(p) + b32) : nullptr;
  *out = v;
  return LAB_OK;
}

extern "C" int lab_vec_create(lab_ctx* c, size_t count, lab_vec** out) { return vec_create(c, count, 0, false, out); }
extern "C" int lab_vec_create_digit(lab_ctx* c, size_t count, int compact, lab_vec** out) {
  return vec_create(c, count, 1, compact != 0, out);
}
extern "C" int lab_vec_basis(const lab_vec* v) { return v ? (v->basis | (v->compact ? 2 : 0)) : -1; }

extern "C" void lab_vec_destroy(lab_vec* v) {
  if (!v) return;
  if (v->ctx && v->ctx->stream) (void)hipStreamSynchronize(v->ctx->stream);
  if (v->d) (void)hipFree(v->d);
  delete v;
}

extern "C" size_t lab_vec_count(const lab_vec* v) { return v ? v->count : 0; }

// operands of the transform-domain ops must be plain u32 vectors of one basis (only the streamed matrix of
// lab_matvec_sum / lab_axpy_commit may be compact)
static int check_plain(lab_ctx* c, const lab_vec* v, int basis, const char* what) {
  if (v->compact) return fail(c, LAB_ERR_ARG, "%s: a compact (u16-limb) vector is only valid as the matrix of lab_matvec_sum / lab_axpy_commit", what);
  if (v->basis != basis) return fail(c, LAB_ERR_ARG, "%s: operands live in different RNS bases", what);
  return LAB_OK;
}

int lab::check_range(lab_ctx* c, const lab_vec* v, size_t off, size_t n, const char* what) {
  if (!v) return fail(c, LAB_ERR_ARG, "%s: null vector", what);
  if (v->ctx != c) return fail(c, LAB_ERR_ARG, "%s: vector belongs to another context", what);
  if (off > v->count || n > v->count - off)
    return fail(c, LAB_ERR_LEN_MISMATCH, "%s: range [%zu,%zu) exceeds vector of %zu elements", what, off, off + n, v->count);
  return LAB_OK;
}

int lab::from_hat(lab_ctx* c, ConstHatView src, size_t count, uint32_t* tmp, uint64_t* dcoeffs, int basis) {
  HatView t{tmp, count * D};
  LAB_HIP(c, launch_ntt(false, src, t, count, basis_nlimbs(c, basis), basis_tabs(c, basis), 0, c->stream));
  LAB_HIP(c, launch_crt_to_u64(t, dcoeffs, count, basis_crt(c, basis), basis_nlimbs(c, basis), c->stream));
  return LAB_OK;
}

int lab::to_hat(lab_ctx* c, const uint64_t* dcoeffs, HatView dst, size_t count, int basis) {
  JobPtrs src;
  src.p[0] = const_cast<uint64_t*>(dcoeffs);
  LAB_HIP(c, launch_to_hat_fused(src, dst, 0, count, basis_nlimbs(c, basis), basis_tabs(c, basis), 1, c->stream));   // reduce + NTT
  return LAB_OK;
}

// coefficients (device) -> `dst` elements [dst_off, dst_off+count) in dst's basis; compact vectors go through a
// u32 staging area (`stage32`, dst->nlimbs() * count * 64 words) and get their last plane packed to u16
static int store_hat_from_coeffs(lab_ctx* c, const uint64_t* dcoeffs, lab_vec* dst, size_t dst_off, size_t count,
                                 uint32_t* stage32) {
  if (!dst->compact) return to_hat(c, dcoeffs, dst->view(dst_off), count, dst->basis);
  HatView t{stage32, count * D};
  int rc = to_hat(c, dcoeffs, t, count, dst->basis);
  if (rc) return rc;
  const int nfull = dst->planes32();
  for (int l = 0; l < nfull; l++)
    LAB_HIP(c, hipMemcpyAsync(dst->d + (size_t)l * dst->count * D + dst_off * D, stage32 + (size_t)l * count * D,
                              count * D * 4, hipMemcpyDeviceToDevice, c->stream));
  LAB_HIP(c, launch_pack_u16(dst->d16 + dst_off * D, stage32 + (size_t)nfull * count * D, count, false, nullptr, c->stream));
  return LAB_OK;
}

int lab::matvec_partial_any(lab_ctx* c, HatView partial, const lab_vec* M, size_t m_off, ConstHatView V, size_t nmat,
                            size_t rows, size_t cols) {
  const int nfull = M->planes32();
  int rc = LAB_OK;
  if (nfull > 0 && (rc = matvec_partial_timed(c, partial, M->view(m_off), V, nmat, rows, cols, nfull, M->tabs()))) return rc;
  if (M->compact)
    LAB_HIP(c, launch_matvec_partial_u16(partial.base + (size_t)nfull * partial.plane_stride, M->d16 + m_off * D,
                                         V.base + (size_t)nfull * V.plane_stride, nmat, rows, cols, M->tabs() + nfull,
                                         c->stream));
  return LAB_OK;
}

extern "C" int lab_vec_load_u64(lab_ctx* c, lab_vec* dst, size_t dst_off, const uint64_t* coeffs, size_t count) {
  if (!c) return LAB_ERR_ARG;
  int rc = check_range(c, dst, dst_off, count, "lab_vec_load_u64");
  if (rc) return rc;
  if (count == 0) return LAB_OK;
  if (!coeffs) return fail(c, LAB_ERR_ARG, "lab_vec_load_u64: null coefficients");
  LAB_HIP(c, hipSetDevice(c->device));
  const void* dsrc = coeffs;
  const bool host = !is_device_pointer(coeffs);
  const size_t stage_bytes = dst->compact ? Carver::padded((size_t)dst->nlimbs() * count * D * 4) : 0;
  if ((rc = ensure_scratch(c, stage_bytes + (host ? count * D * 8 : 0)))) return rc;
  Carver cv(c->scratch);
  uint32_t* stage32 = dst->compact ? cv.take<uint32_t>((size_t)dst->nlimbs() * count * D) : nullptr;
  if (host) {
    uint64_t* hs = cv.take<uint64_t>(count * D);
    if ((rc = stage_in(c, coeffs, hs, count * D * 8, &dsrc))) return rc;
  }
  if ((rc = store_hat_from_coeffs(c, (const uint64_t*)dsrc, dst, dst_off, count, stage32))) return rc;
  if (host) LAB_HIP(c, hipStreamSynchronize(c->stream));  // the staged host buffer may be reused by the caller
  return LAB_OK;
}

extern "C" int lab_vec_store_u64(lab_ctx* c, const lab_vec* src, size_t src_off, size_t count, uint64_t* coeffs) {
  if (!c) return LAB_ERR_ARG;
  int rc = check_range(c, src, src_off, count, "lab_vec_store_u64");
  if (rc) return rc;
  if (count == 0) return LAB_OK;
  if (!coeffs) return fail(c, LAB_ERR_ARG, "lab_vec_store_u64: null output");
  LAB_HIP(c, hipSetDevice(c->device));
  const size_t words = count * D, L = src->nlimbs();
  const bool out_dev = is_device_pointer(coeffs);
  if ((rc = ensure_scratch(c, 2 * Carver::padded(L * words * 4) + Carver::padded(words * 8)))) return rc;
  Carver cv(c->scratch);
  uint32_t* tmp = cv.take<uint32_t>(L * words);
  uint32_t* unp = cv.take<uint32_t>(L * words);
  uint64_t* so = cv.take<uint64_t>(words);
  HatView t{tmp, words};
  ConstHatView sv = src->view(src_off);
  if (src->compact) {  // rebuild dense u32 planes first
    const int nfull = src->planes32();
    for (int l = 0; l < nfull; l++)
      LAB_HIP(c, hipMemcpyAsync(unp + (size_t)l * words, src->d + (size_t)l * src->count * D + src_off * D, words * 4,
                                hipMemcpyDeviceToDevice, c->stream));
    LAB_HIP(c, launch_pack_u16(const_cast<uint16_t*>(src->d16) + src_off * D, nullptr, count, true, unp + (size_t)nfull * words, c->stream));
    sv = ConstHatView(unp, words);
  }
  LAB_HIP(c, launch_ntt(false, sv, t, count, (int)L, src->tabs(), 0, c->stream));
  LAB_HIP(c, launch_crt_to_u64(t, out_dev ? coeffs : so, count, src->crt(), (int)L, c->stream));
  if (!out_dev) {
    LAB_HIP(c, hipMemcpyAsync(coeffs, so, words * 8, hipMemcpyDeviceToHost, c->stream));
    LAB_HIP(c, hipStreamSynchronize(c->stream));
  }
  return LAB_OK;
}

extern "C" int lab_vec_fill_synthetic(lab_ctx* c, lab_vec* dst, size_t dst_off, size_t count, uint64_t seed,
                                      uint64_t first_elem, uint64_t bound) {
  if (!c) return LAB_ERR_ARG;
  int rc = check_range(c, dst, dst_off, count, "lab_vec_fill_synthetic");
  if (rc) return rc;
  if (bound == 0 || bound > c->q) return fail(c, LAB_ERR_ARG, "lab_vec_fill_synthetic: bound must be in [1,q]");
  LAB_HIP(c, hipSetDevice(c->device));
  // generate coefficient-domain residues in chunks, then transform in place
  const size_t chunk = 1u << 18;  // 128 MiB of u64 staging at most
  const size_t cn = count < chunk ? count : chunk;
  if ((rc = ensure_scratch(c, Carver::padded(cn * D * 8) + (dst->compact ? (size_t)dst->nlimbs() * cn * D * 4 : 0)))) return rc;
  Carver cv(c->scratch);
  uint64_t* stage = cv.take<uint64_t>(cn * D);
  uint32_t* stage32 = dst->compact ? cv.take<uint32_t>((size_t)dst->nlimbs() * cn * D) : nullptr;
  for (size_t done = 0; done < count; done += chunk) {
    size_t n = count - done < chunk ? count - done : chunk;
    LAB_HIP(c, launch_synth_fill_u64(stage, n, seed, first_elem + done, bound, c->stream));
    if ((rc = store_hat_from_coeffs(c, stage, dst, dst_off + done, n, stage32))) return rc;
  }
  return LAB_OK;
}

// ------------------------------------------------------------- CRS expansion (ChaCha20 + gen_range)
extern "C" int lab_crs_expand_u64(lab_ctx* c, uint64_t* coeffs, size_t count, const uint8_t base_seed[32],
                                  uint64_t offset_lo, uint64_t offset_hi) {
  if (!c) return LAB_ERR_ARG;
  if (count == 0) return LAB_OK;
  if (!coeffs || !base_seed) return fail(c, LAB_ERR_ARG, "lab_crs_expand_u64: null argument");
  LAB_HIP(c, hipSetDevice(c->device));
  int rc = ensure_scratch(c, Staging::need(coeffs, count * D * 8));
  if (rc) return rc;
  Carver cv(c->scratch);
  Staging st(c, &cv);
  uint64_t* d = st.out(coeffs, count * D);
  LAB_HIP(c, launch_crs_expand(d, count, base_seed, offset_lo, offset_hi, c->q, c->stream));
  return st.finish();
}

extern "C" int lab_crs_expand(lab_ctx* c, lab_vec* dst, size_t dst_off, size_t count, const uint8_t base_seed[32],
                              uint64_t offset_lo, uint64_t offset_hi) {
  if (!c) return LAB_ERR_ARG;
  int rc = check_range(c, dst, dst_off, count, "lab_crs_expand");
  if (rc) return rc;
  if (!base_seed) return fail(c, LAB_ERR_ARG, "lab_crs_expand: null seed");
  LAB_HIP(c, hipSetDevice(c->device));
  const size_t chunk = 1u << 18;
  const size_t cn = count < chunk ? count : chunk;
  if ((rc = ensure_scratch(c, Carver::padded(cn * D * 8) + (dst->compact ? (size_t)dst->nlimbs() * cn * D * 4 : 0)))) return rc;
  Carver cv(c->scratch);
  uint64_t* stage = cv.take<uint64_t>(cn * D);
  uint32_t* stage32 = dst->compact ? cv.take<uint32_t>((size_t)dst->nlimbs() * cn * D) : nullptr;
  for (size_t done = 0; done < count; done += chunk) {
    const size_t n = count - done < chunk ? count - done : chunk;
    // offset advances by 64 seeds per polynomial
    unsigned __int128 off = (((unsigned __int128)offset_hi << 64) | offset_lo) + (unsigned __int128)done * D;
    LAB_HIP(c, launch_crs_expand(stage, n, base_seed, (uint64_t)off, (uint64_t)(off >> 64), c->q, c->stream));
    if ((rc = store_hat_from_coeffs(c, stage, dst, dst_off + done, n, stage32))) return rc;
  }
  return LAB_OK;
}

// ------------------------------------------------------------- inner products / mat-vec
extern "C" int lab_matvec_sum(lab_ctx* c, lab_vec* out, size_t out_off, const lab_vec* M, size_t m_off, size_t nmat,
                              size_t rows, size_t cols, const lab_vec* V, size_t v_off) {
  if (!c) return LAB_ERR_ARG;
  int rc;
  if ((rc = check_range(c, out, out_off, rows, "lab_matvec_sum(out)"))) return rc;
  if (nmat && rows && cols > ((size_t)-1) / nmat / rows) return fail(c, LAB_ERR_ARG, "lab_matvec_sum: size overflow");
  if ((rc = check_range(c, M, m_off, nmat * rows * cols, "lab_matvec_sum(M)"))) return rc;
  if ((rc = check_range(c, V, v_off, nmat * cols, "lab_matvec_sum(V)"))) return rc;
  if (nmat == 0 || rows == 0 || cols == 0) return fail(c, LAB_ERR_ARG, "lab_matvec_sum: empty dimension");
  const int basis = M->basis;
  if ((rc = check_plain(c, V, basis, "lab_matvec_sum(V)")) || (rc = check_plain(c, out, basis, "lab_matvec_sum(out)"))) return rc;
  // main basis: exact for canonical operands up to max_len terms; digit basis: exact for up to dg_terms terms
  // provided the entries of V are digits below the bound given to lab_ctx_enable_digit_basis (caller's contract)
  const uint64_t limit = basis ? c->dg_terms : c->max_len;
  if ((u128)nmat * cols > limit)
    return fail(c, LAB_ERR_RANGE, "lab_matvec_sum: reduction length %zu exceeds the exactness bound %llu of this basis",
                nmat * cols, (unsigned long long)limit);
  if (rows > 0x7fffffffu || cols > 0x7fffffffu / 64 || nmat > 65535) return fail(c, LAB_ERR_ARG, "lab_matvec_sum: dimension too large");
  LAB_HIP(c, hipSetDevice(c->device));
  if (nmat == 1) return matvec_partial_any(c, out->view(out_off), M, m_off, V->view(v_off), 1, rows, cols);
  const size_t pw = nmat * rows * D;
  if ((rc = ensure_scratch(c, (size_t)M->nlimbs() * pw * 4))) return rc;
  HatView partial{static_cast<uint32_t*>(c->scratch), pw};
  if ((rc = matvec_partial_any(c, partial, M, m_off, V->view(v_off), nmat, rows, cols))) return rc;
  LAB_HIP(c, launch_sum_terms(out->view(out_off), partial, nmat, rows, M->nlimbs(), M->tabs(), c->stream));
  return LAB_OK;
}

namespace lab {
// partial[b][m][rho] for nb vector sets sharing the (possibly compact) matrices M; V set b at v.base + b*v_bs
int matvec_partial_batched_any(lab_ctx* c, int nb, HatView partial, size_t partial_bs, const lab_vec* M, size_t m_off,
                               ConstHatView V, size_t v_bs, size_t nmat, size_t rows, size_t cols) {
  const int nfull = M->planes32();
  const bool prof = c->prof_on && nfull > 0;
  if (prof && c->prof_used == c->prof_events.size()) {
    hipEvent_t a, b;
    LAB_HIP(c, hipEventCreate(&a));
    LAB_HIP(c, hipEventCreate(&b));
    c->prof_events.emplace_back(a, b);
  }
  if (prof) LAB_HIP(c, hipEventRecord(c->prof_events[c->prof_used].first, c->stream));
  if (nfull > 0)
    LAB_HIP(c, launch_matvec_partial_batched(nb, partial, partial_bs, M->view(m_off), V, v_bs, nmat, rows, cols, nfull,
                                             M->tabs(), c->stream));
  if (prof) {
    LAB_HIP(c, hipEventRecord(c->prof_events[c->prof_used++].second, c->stream));
    c->prof_kernel = matvec_kernel_name();
    c->prof_bytes += (double)nfull * 256.0 * ((double)nmat * rows * cols + (double)nb * nmat * cols + (double)nb * nmat * rows);
  }
  if (M->compact)
    LAB_HIP(c, launch_matvec_partial_batched_u16(nb, partial.base + (size_t)nfull * partial.plane_stride, partial_bs,
                                                 M->d16 + m_off * D, V.base + (size_t)nfull * V.plane_stride, v_bs, nmat,
                                                 rows, cols, M->tabs() + nfull, c->stream));
  return LAB_OK;
}

// number of equal chunks a long inner product is cut into (a single block would otherwise stream it alone)
size_t inner_product_splits(size_t len) {
  size_t k = 1;
  while (k < 512 && len % (2 * k) == 0 && len / (2 * k) >= 64) k *= 2;
  return k;
}
// out[0] = <v1, v2>; `partial` must hold inner_product_splits(len) elements
int inner_product_impl(lab_ctx* c, HatView out, ConstHatView v1, ConstHatView v2, size_t len, HatView partial) {
  const size_t k = inner_product_splits(len);
  if (k == 1) {
    LAB_HIP(c, launch_matvec_partial(out, v1, v2, 1, 1, len, c->nlimbs, c->d_tabs, c->stream));
    return LAB_OK;
  }
  LAB_HIP(c, launch_matvec_partial(partial, v1, v2, k, 1, len / k, c->nlimbs, c->d_tabs, c->stream));
  LAB_HIP(c, launch_sum_terms(out, partial, k, 1, c->nlimbs, c->d_tabs, c->stream));
  return LAB_OK;
}
}  // namespace lab

extern "C" int lab_matvec_sum_batched(lab_ctx* c, lab_vec* out, size_t out_off, const lab_vec* M, size_t m_off, size_t nmat,
                                      size_t rows, size_t cols, const lab_vec* V, size_t v_off, int nb) {
  if (!c) return LAB_ERR_ARG;
  if (nb != 1 && nb != 2 && nb != 4 && nb != 8) return fail(c, LAB_ERR_ARG, "lab_matvec_sum_batched: nb must be 1, 2, 4 or 8");
  int rc;
  if (nmat == 0 || rows == 0 || cols == 0) return fail(c, LAB_ERR_ARG, "lab_matvec_sum_batched: empty dimension");
  if (rows > 0x7fffffffu || cols > 0x7fffffffu / 64 || nmat > 65535) return fail(c, LAB_ERR_ARG, "lab_matvec_sum_batched: dimension too large");
  if ((rc = check_range(c, out, out_off, (size_t)nb * rows, "lab_matvec_sum_batched(out)"))) return rc;
  if ((rc = check_range(c, M, m_off, nmat * rows * cols, "lab_matvec_sum_batched(M)"))) return rc;
  if ((rc = check_range(c, V, v_off, (size_t)nb * nmat * cols, "lab_matvec_sum_batched(V)"))) return rc;
  const int basis = M->basis;
  if ((rc = check_plain(c, V, basis, "lab_matvec_sum_batched(V)")) || (rc = check_plain(c, out, basis, "lab_matvec_sum_batched(out)"))) return rc;
  const uint64_t limit = basis ? c->dg_terms : c->max_len;
  if ((u128)nmat * cols > limit) return fail(c, LAB_ERR_RANGE, "lab_matvec_sum_batched: reduction length %zu exceeds the exactness bound %llu of this basis", nmat * cols, (unsigned long long)limit);
  LAB_HIP(c, hipSetDevice(c->device));
  const size_t pw = (size_t)nb * nmat * rows * D;
  if ((rc = ensure_scratch(c, (size_t)M->nlimbs() * pw * 4))) return rc;
  HatView partial{static_cast<uint32_t*>(c->scratch), pw};
  ConstHatView vv = V->view(v_off);
  if ((rc = matvec_partial_batched_any(c, nb, partial, nmat * rows * D, M, m_off, vv, nmat * cols * D, nmat, rows, cols))) return rc;
  for (int b = 0; b < nb; b++) {
    HatView pb{partial.base + (size_t)b * nmat * rows * D, pw};
    LAB_HIP(c, launch_sum_terms(out->view(out_off + (size_t)b * rows), pb, nmat, rows, M->nlimbs(), M->tabs(), c->stream));
  }
  return LAB_OK;
}

extern "C" int lab_inner_product(lab_ctx* c, lab_vec* out, size_t out_off, const lab_vec* v1, size_t off1,
                                 const lab_vec* v2, size_t off2, size_t len) {
  if (!c) return LAB_ERR_ARG;
  int rc;
  if ((rc = check_range(c, out, out_off, 1, "lab_inner_product(out)"))) return rc;
  if ((rc = check_range(c, v1, off1, len, "lab_inner_product(v1)"))) return rc;
  if ((rc = check_range(c, v2, off2, len, "lab_inner_product(v2)"))) return rc;
  if (len == 0) return fail(c, LAB_ERR_ARG, "lab_inner_product: empty vectors");
  if ((rc = check_plain(c, v1, 0, "lab_inner_product")) || (rc = check_plain(c, v2, 0, "lab_inner_product")) ||
      (rc = check_plain(c, out, 0, "lab_inner_product"))) return rc;
  if (len > c->max_len) return fail(c, LAB_ERR_RANGE, "lab_inner_product: length %zu exceeds the exactness bound %llu of this context", len, (unsigned long long)c->max_len);
  if (len > 0x7fffffffu / 64) return fail(c, LAB_ERR_ARG, "lab_inner_product: length too large");
  LAB_HIP(c, hipSetDevice(c->device));
  const size_t k = inner_product_splits(len);
  if ((rc = ensure_scratch(c, (size_t)c->nlimbs * k * D * 4))) return rc;
  return inner_product_impl(c, out->view(out_off), v1->view(off1), v2->view(off2), len, HatView{static_cast<uint32_t*>(c->scratch), k * D});
}

// ------------------------------------------------------------- remaining hat-domain ops
extern "C" int lab_ajtai_commit(lab_ctx* c, lab_vec* out, size_t out_off, const lab_vec* A, size_t a_off, size_t rows,
                                size_t inner, const lab_vec* S, size_t s_off, size_t ncols) {
  if (!c) return LAB_ERR_ARG;
  int rc;
  if (rows == 0 || inner == 0 || ncols == 0) return fail(c, LAB_ERR_ARG, "lab_ajtai_commit: empty dimension");
  if (rows > 0x7fffffffu || inner > 0x1ffffffu || ncols > 0xffffffu) return fail(c, LAB_ERR_ARG, "lab_ajtai_commit: dimension too large");
  if ((rc = check_range(c, out, out_off, rows * ncols, "lab_ajtai_commit(out)"))) return rc;
  if ((rc = check_range(c, A, a_off, rows * inner, "lab_ajtai_commit(A)"))) return rc;
  if ((rc = check_range(c, S, s_off, inner * ncols, "lab_ajtai_commit(S)"))) return rc;
  if ((rc = check_plain(c, A, 0, "lab_ajtai_commit")) || (rc = check_plain(c, S, 0, "lab_ajtai_commit")) ||
      (rc = check_plain(c, out, 0, "lab_ajtai_commit"))) return rc;
  if (inner > c->max_len) return fail(c, LAB_ERR_RANGE, "lab_ajtai_commit: reduction length %zu exceeds the exactness bound %llu", inner, (unsigned long long)c->max_len);
  LAB_HIP(c, hipSetDevice(c->device));
  hipError_t e = launch_matmul_bcast(out->view(out_off), rows * D, 0, A->view(a_off), S->view(s_off), D, ncols * D, 0, 1, rows,
                                     inner, ncols, c->nlimbs, c->d_tabs, c->stream);
  if (e == hipErrorNotSupported)
    e = launch_matmul_t(out->view(out_off), A->view(a_off), S->view(s_off), rows, inner, ncols, c->nlimbs, c->d_tabs, c->stream);
  LAB_HIP(c, e);
  return LAB_OK;
}

extern "C" int lab_gram(lab_ctx* c, lab_vec* out, size_t out_off, const lab_vec* P, size_t p_off, const lab_vec* S,
                        size_t s_off, size_t n, size_t r) {
  if (!c) return LAB_ERR_ARG;
  int rc;
  if (n == 0 || r == 0 || r > 65535 || n > 0x7fffffffu / 64) return fail(c, LAB_ERR_ARG, "lab_gram: bad dimension");
  if ((rc = check_range(c, out, out_off, r * r, "lab_gram(out)"))) return rc;
  if ((rc = check_range(c, P, p_off, n * r, "lab_gram(P)"))) return rc;
  if ((rc = check_range(c, S, s_off, n * r, "lab_gram(S)"))) return rc;
  if ((rc = check_plain(c, P, 0, "lab_gram")) || (rc = check_plain(c, S, 0, "lab_gram")) || (rc = check_plain(c, out, 0, "lab_gram"))) return rc;
  if (n > c->max_len) return fail(c, LAB_ERR_RANGE, "lab_gram: reduction length %zu exceeds the exactness bound %llu", n, (unsigned long long)c->max_len);
  LAB_HIP(c, hipSetDevice(c->device));
  const size_t chunks = gram_num_chunks(n);
  if ((rc = ensure_scratch(c, (size_t)c->nlimbs * chunks * r * r * D * 4))) return rc;
  return gram_impl(c, out->view(out_off), P->view(p_off), S->view(s_off), n, r,
                   HatView{static_cast<uint32_t*>(c->scratch), chunks * r * r * D});
}

int lab::gram_impl(lab_ctx* c, HatView out, ConstHatView P, ConstHatView S, size_t n, size_t r, HatView partial) {
  const size_t chunks = gram_chunks(n, r, c->nlimbs, 1);
  if (chunks == 1) {
    LAB_HIP(c, launch_gram_tiled(out, P, S, n, r, c->nlimbs, c->d_tabs, c->stream));
    return LAB_OK;
  }
  LAB_HIP(c, launch_gram_tiled(partial, P, S, n, r, c->nlimbs, c->d_tabs, c->stream));
  LAB_HIP(c, launch_sum_terms(out, partial, chunks, r * r, c->nlimbs, c->d_tabs, c->stream));
  return LAB_OK;
}

namespace lab {
// shared by lab_axpy_commit and prove_core; `tmp_partial` must hold nlimbs * chunks * len * 64 u32
int axpy_commit_impl(lab_ctx* c, HatView out, ConstHatView G, ConstHatView C, size_t nterms, size_t len, HatView partial) {
  const size_t chunks = axpy_num_chunks(nterms);
  if (chunks == 1) {
    LAB_HIP(c, launch_axpy_partial(out, G, C, nterms, len, c->nlimbs, c->d_tabs, c->stream));
    return LAB_OK;
  }
  LAB_HIP(c, launch_axpy_partial(partial, G, C, nterms, len, c->nlimbs, c->d_tabs, c->stream));
  LAB_HIP(c, launch_sum_terms(out, partial, chunks, len, c->nlimbs, c->d_tabs, c->stream));
  return LAB_OK;
}
}  // namespace lab

int lab::axpy_commit_any(lab_ctx* c, HatView out, ConstHatView G, const lab_vec* Cm, size_t c_off, size_t nterms, size_t len,
                         HatView partial) {
  const size_t chunks = axpy_num_chunks(nterms);
  HatView target = chunks == 1 ? out : partial;
  const int nfull = Cm->planes32();
  if (nfull > 0) LAB_HIP(c, launch_axpy_partial(target, G, Cm->view(c_off), nterms, len, nfull, Cm->tabs(), c->stream));
  if (Cm->compact)
    LAB_HIP(c, launch_axpy_partial_u16(target.base + (size_t)nfull * target.plane_stride, G.base + (size_t)nfull * G.plane_stride,
                                       Cm->d16 + c_off * D, nterms, len, Cm->tabs() + nfull, c->stream));
  if (chunks > 1) LAB_HIP(c, launch_sum_terms(out, partial, chunks, len, Cm->nlimbs(), Cm->tabs(), c->stream));
  return LAB_OK;
}

// nb sets at once: out_b = out.base + b*out_bs, G_b = G.base + b*g_bs, partial_b = partial.base + b*partial_bs
// `partial` holds max(axpy_num_chunks(nterms), min(64, ...)) * len elements per set: callers size it with
// axpy_partial_elems(nterms, len)
int lab::axpy_commit_batched_any(lab_ctx* c, int nb, HatView out, size_t out_bs, ConstHatView G, size_t g_bs, const lab_vec* Cm,
                                 size_t c_off, size_t nterms, size_t len, HatView partial, size_t partial_bs) {
  size_t chunks = axpy_num_chunks(nterms);
  const int nfull = Cm->planes32();
  // several digit vectors, long commitments: lane = rho product on the LDS-DMA kernel (partials fit the same scratch)
  const size_t bchunk = nb >= 2 ? axpy_bcast_chunk(nterms, len) : 0;
  if (bchunk) {
    hipError_t e = hipSuccess;
    if (nfull > 0)
      e = launch_axpy_bcast_u32(nb, partial, partial_bs, G, g_bs, Cm->view(c_off), nterms, bchunk, len, nfull, Cm->tabs(), c->stream);
    if (e == hipSuccess && Cm->compact)
      e = launch_axpy_bcast_u16(nb, partial.base + (size_t)nfull * partial.plane_stride, partial_bs,
                                G.base + (size_t)nfull * G.plane_stride, g_bs, Cm->d16 + c_off * D, nterms, bchunk, len,
                                Cm->tabs() + nfull, c->stream);
    if (e != hipErrorNotSupported) {
      LAB_HIP(c, e);
      for (int b = 0; b < nb; b++)
        LAB_HIP(c, launch_sum_terms(HatView{out.base + (size_t)b * out_bs, out.plane_stride},
                                    HatView{partial.base + (size_t)b * partial_bs, partial.plane_stride}, nterms / bchunk, len,
                                    Cm->nlimbs(), Cm->tabs(), c->stream));
      return LAB_OK;
    }
  }
  HatView target = chunks == 1 ? out : partial;
  const size_t target_bs = chunks == 1 ? out_bs : partial_bs;
  if (nfull > 0)
    LAB_HIP(c, launch_axpy_partial_batched(nb, target, target_bs, G, g_bs, Cm->view(c_off), nterms, len, nfull, Cm->tabs(), c->stream));
  if (Cm->compact)
    LAB_HIP(c, launch_axpy_partial_batched_u16(nb, target.base + (size_t)nfull * target.plane_stride, target_bs,
                                               G.base + (size_t)nfull * G.plane_stride, g_bs, Cm->d16 + c_off * D, nterms, len,
                                               Cm->tabs() + nfull, c->stream));
  if (chunks > 1)
    for (int b = 0; b < nb; b++)
      LAB_HIP(c, launch_sum_terms(HatView{out.base + (size_t)b * out_bs, out.plane_stride},
                                  HatView{partial.base + (size_t)b * partial_bs, partial.plane_stride}, chunks, len,
                                  Cm->nlimbs(), Cm->tabs(), c->stream));
  return LAB_OK;
}

extern "C" int lab_axpy_commit(lab_ctx* c, lab_vec* out, size_t out_off, const lab_vec* G, size_t g_off,
                               const lab_vec* Cm, size_t c_off, size_t nterms, size_t len) {
  if (!c) return LAB_ERR_ARG;
  int rc;
  if (nterms == 0 || len == 0 || len > 0x7fffffffu / 64 || nterms > 0x3fffffu) return fail(c, LAB_ERR_ARG, "lab_axpy_commit: bad dimension");
  if ((rc = check_range(c, out, out_off, len, "lab_axpy_commit(out)"))) return rc;
  if ((rc = check_range(c, G, g_off, nterms, "lab_axpy_commit(G)"))) return rc;
  if ((rc = check_range(c, Cm, c_off, nterms * len, "lab_axpy_commit(C)"))) return rc;
  const int basis = Cm->basis;
  if ((rc = check_plain(c, G, basis, "lab_axpy_commit(G)")) || (rc = check_plain(c, out, basis, "lab_axpy_commit(out)"))) return rc;
  const uint64_t limit = basis ? c->dg_terms : c->max_len;
  if (nterms > limit) return fail(c, LAB_ERR_RANGE, "lab_axpy_commit: %zu terms exceed the exactness bound %llu of this basis", nterms, (unsigned long long)limit);
  LAB_HIP(c, hipSetDevice(c->device));
  const size_t chunks = axpy_num_chunks(nterms);
  if ((rc = ensure_scratch(c, (size_t)Cm->nlimbs() * chunks * len * D * 4))) return rc;
  return axpy_commit_any(c, out->view(out_off), G->view(g_off), Cm, c_off, nterms, len,
                         HatView{static_cast<uint32_t*>(c->scratch), chunks * len * D});
}

size_t lab::axpy_partial_elems(size_t nterms, size_t len) {
  size_t chunks = axpy_num_chunks(nterms);
  const size_t bchunk = axpy_bcast_chunk(nterms, len);
  if (bchunk && nterms / bchunk > chunks) chunks = nterms / bchunk;
  return chunks * len;
}

extern "C" int lab_axpy_commit_batched(lab_ctx* c, lab_vec* out, size_t out_off, const lab_vec* G, size_t g_off,
                                       const lab_vec* Cm, size_t c_off, size_t nterms, size_t len, int nb) {
  if (!c) return LAB_ERR_ARG;
  int rc;
  if (nb != 1 && nb != 2 && nb != 4 && nb != 8) return fail(c, LAB_ERR_ARG, "lab_axpy_commit_batched: nb must be 1, 2, 4 or 8");
  if (nterms == 0 || len == 0 || len > 0x7fffffffu / 64 || nterms > 0x3fffffu) return fail(c, LAB_ERR_ARG, "lab_axpy_commit_batched: bad dimension");
  if ((rc = check_range(c, out, out_off, (size_t)nb * len, "lab_axpy_commit_batched(out)"))) return rc;
  if ((rc = check_range(c, G, g_off, (size_t)nb * nterms, "lab_axpy_commit_batched(G)"))) return rc;
  if ((rc = check_range(c, Cm, c_off, nterms * len, "lab_axpy_commit_batched(C)"))) return rc;
  const int basis = Cm->basis;
  if ((rc = check_plain(c, G, basis, "lab_axpy_commit_batched(G)")) || (rc = check_plain(c, out, basis, "lab_axpy_commit_batched(out)"))) return rc;
  const uint64_t limit = basis ? c->dg_terms : c->max_len;
  if (nterms > limit) return fail(c, LAB_ERR_RANGE, "lab_axpy_commit_batched: %zu terms exceed the exactness bound %llu of this basis", nterms, (unsigned long long)limit);
  LAB_HIP(c, hipSetDevice(c->device));
  const size_t pe = axpy_partial_elems(nterms, len);
  if ((rc = ensure_scratch(c, (size_t)Cm->nlimbs() * nb * pe * D * 4))) return rc;
  return axpy_commit_batched_any(c, nb, out->view(out_off), len * D, G->view(g_off), nterms * D, Cm, c_off, nterms, len,
                                 HatView{static_cast<uint32_t*>(c->scratch), (size_t)nb * pe * D}, pe * D);
}

extern "C" int lab_amortize_z(lab_ctx* c, lab_vec* out, size_t out_off, const lab_vec* S, size_t s_off, size_t n,
                              size_t r, const lab_vec* cv, size_t c_off) {
  // z[x] = <S[x,:], c>: a mat-vec with rows = n, cols = r (src/proofgen.rs:387-399)
  return lab_matvec_sum(c, out, out_off, S, s_off, 1, n, r, cv, c_off);
}

extern "C" int lab_pointwise(lab_ctx* c, lab_op op, lab_vec* out, size_t out_off, const lab_vec* a, size_t a_off,
                             const lab_vec* b, size_t b_off, size_t count) {
  if (!c) return LAB_ERR_ARG;
  int rc;
  if ((rc = check_range(c, out, out_off, count, "lab_pointwise(out)"))) return rc;
  if ((rc = check_range(c, a, a_off, count, "lab_pointwise(a)"))) return rc;
  if ((rc = check_range(c, b, b_off, count, "lab_pointwise(b)"))) return rc;
  PointwiseOp k;
  switch (op) {
    case LAB_OP_MUL: k = PW_MUL; break;
    case LAB_OP_ADD: k = PW_ADD; break;
    case LAB_OP_SUB: k = PW_SUB; break;
    default: return fail(c, LAB_ERR_ARG, "lab_pointwise: unknown op %d", (int)op);
  }
  if ((rc = check_plain(c, b, a->basis, "lab_pointwise")) || (rc = check_plain(c, out, a->basis, "lab_pointwise")) ||
      (rc = check_plain(c, a, a->basis, "lab_pointwise"))) return rc;
  LAB_HIP(c, hipSetDevice(c->device));
  LAB_HIP(c, launch_pointwise(k, out->view(out_off), a->view(a_off), b->view(b_off), count, a->nlimbs(), a->tabs(), c->stream));
  return LAB_OK;
}

extern "C" int lab_poly_by_poly_vec(lab_ctx* c, lab_vec* out, size_t out_off, const lab_vec* p, size_t p_off,
                                    const lab_vec* v, size_t v_off, size_t len) {
  if (!c) return LAB_ERR_ARG;
  int rc;
  if ((rc = check_range(c, out, out_off, len, "lab_poly_by_poly_vec(out)"))) return rc;
  if ((rc = check_range(c, p, p_off, 1, "lab_poly_by_poly_vec(p)"))) return rc;
  if ((rc = check_range(c, v, v_off, len, "lab_poly_by_poly_vec(v)"))) return rc;
  if ((rc = check_plain(c, p, 0, "lab_poly_by_poly_vec")) || (rc = check_plain(c, v, 0, "lab_poly_by_poly_vec")) ||
      (rc = check_plain(c, out, 0, "lab_poly_by_poly_vec"))) return rc;
  LAB_HIP(c, hipSetDevice(c->device));
  LAB_HIP(c, launch_poly_by_vec(out->view(out_off), p->view(p_off), v->view(v_off), len, c->nlimbs, c->d_tabs, c->stream));
  return LAB_OK;
}

// ------------------------------------------------------------- coefficient-domain ops
extern "C" int lab_decompose(lab_ctx* c, uint64_t* out, const uint64_t* coeffs, size_t count, int64_t base, int exp) {
  if (!c) return LAB_ERR_ARG;
  if (base < 2) return fail(c, LAB_ERR_DECOMPOSE, "lab_decompose: base %lld < 2 never terminates in the reference (src/util.rs:410-417)", (long long)base);
  if (exp < 0 || exp > 64) return fail(c, LAB_ERR_DECOMPOSE, "lab_decompose: exp %d out of range", exp);
  if (count == 0 || exp == 0) return LAB_OK;
  if (!out || !coeffs) return fail(c, LAB_ERR_ARG, "lab_decompose: null argument");
  LAB_HIP(c, hipSetDevice(c->device));
  const size_t nin = count * D, nout = nin * (size_t)exp;
  int rc = ensure_scratch(c, Staging::need(coeffs, nin * 8) + Staging::need(out, nout * 8));
  if (rc) return rc;
  Carver cv(c->scratch);
  Staging st(c, &cv);
  const uint64_t* din;
  if ((rc = st.in(coeffs, nin, &din))) return rc;
  uint64_t* dout = st.out(out, nout);
  LAB_HIP(c, launch_decompose(dout, din, count, c->q, (uint64_t)base, exp, count, c->stream));
  return st.finish();
}

extern "C" int lab_norm_sq(lab_ctx* c, const uint64_t* coeffs, size_t count, uint64_t* lo, uint64_t* hi) {
  if (!c) return LAB_ERR_ARG;
  if (count == 0) return LAB_OK;
  if (!coeffs || !lo || !hi) return fail(c, LAB_ERR_ARG, "lab_norm_sq: null argument");
  LAB_HIP(c, hipSetDevice(c->device));
  int rc = ensure_scratch(c, Staging::need(coeffs, count * D * 8) + Staging::need(lo, count * 8) + Staging::need(hi, count * 8));
  if (rc) return rc;
  Carver cv(c->scratch);
  Staging st(c, &cv);
  const uint64_t* din;
  if ((rc = st.in(coeffs, count * D, &din))) return rc;
  uint64_t* dlo = st.out(lo, count);
  uint64_t* dhi = st.out(hi, count);
  LAB_HIP(c, launch_norm_sq(dlo, dhi, din, count, c->stream));
  return st.finish();
}

static int coef_unary(lab_ctx* c, uint64_t* out, const uint64_t* coeffs, size_t count, uint64_t s, int mode, const char* what) {
  if (!c) return LAB_ERR_ARG;
  if (count == 0) return LAB_OK;
  if (!out || !coeffs) return fail(c, LAB_ERR_ARG, "%s: null argument", what);
  LAB_HIP(c, hipSetDevice(c->device));
  const size_t nw = count * D;
  int rc = ensure_scratch(c, Staging::need(coeffs, nw * 8) + Staging::need(out, nw * 8));
  if (rc) return rc;
  Carver cv(c->scratch);
  Staging st(c, &cv);
  const uint64_t* din;
  if ((rc = st.in(coeffs, nw, &din))) return rc;
  uint64_t* dout = st.out(out, nw);
  if (mode == 3 && dout == din) return fail(c, LAB_ERR_ARG, "%s: in-place operation not supported", what);
  LAB_HIP(c, launch_coef_elementwise(dout, din, nullptr, count, c->q, s % c->q, mode, c->stream));
  return st.finish();
}

extern "C" int lab_scale(lab_ctx* c, uint64_t* out, const uint64_t* coeffs, size_t count, uint64_t s) {
  return coef_unary(c, out, coeffs, count, s, 0, "lab_scale");
}
extern "C" int lab_sigma_inv(lab_ctx* c, uint64_t* out, const uint64_t* coeffs, size_t count) {
  return coef_unary(c, out, coeffs, count, 0, 3, "lab_sigma_inv");
}

extern "C" int lab_pi_pack(lab_ctx* c, uint8_t* packed, const int8_t* Pi, size_t mats, size_t cols) {
  if (!c) return LAB_ERR_ARG;
  if (!packed || !Pi || cols % 16) return fail(c, LAB_ERR_ARG, "lab_pi_pack: null pointer or a column count that is not a multiple of 16");
  const size_t count = mats * 256 * cols;
  if (count == 0) return LAB_OK;
  LAB_HIP(c, hipSetDevice(c->device));
  PtrCacheScope ptr_cache;
  int rc = ensure_scratch(c, Staging::need(Pi, count) + Staging::need(packed, count / 4));
  if (rc) return rc;
  Carver cv(c->scratch);
  Staging st(c, &cv);
  const int8_t* dpi;
  if ((rc = st.in(Pi, count, &dpi))) return rc;
  uint8_t* dout = st.out(packed, count / 4);
  LAB_HIP(c, launch_pi_pack(dout, dpi, mats, cols, c->stream));
  return st.finish();
}

extern "C" int lab_jl_project(lab_ctx* c, int64_t* proj, const int8_t* Pi, const uint64_t* S, size_t n, size_t r) {
  if (!c) return LAB_ERR_ARG;
  if (!proj || !Pi || !S || n == 0 || r == 0) return fail(c, LAB_ERR_ARG, "lab_jl_project: bad argument");
  if (r > 65535 || n > (1u << 24)) return fail(c, LAB_ERR_ARG, "lab_jl_project: dimension too large");
  // |proj| <= r*n*64*(q-1) must fit int64
  if ((u128)r * n * D * (c->q - 1) >= ((u128)1 << 63)) return fail(c, LAB_ERR_RANGE, "lab_jl_project: projection does not fit 64 bits");
  LAB_HIP(c, hipSetDevice(c->device));
  const size_t npi = r * 256 * n * D, ns = n * r * D;
  int rc = ensure_scratch(c, Staging::need(Pi, npi) + Staging::need(S, ns * 8) + Staging::need(proj, 256 * 8));
  if (rc) return rc;
  Carver cv(c->scratch);
  Staging st(c, &cv);
  const int8_t* dpi;
  const uint64_t* ds;
  if ((rc = st.in(Pi, npi, &dpi))) return rc;
  if ((rc = st.in(S, ns, &ds))) return rc;
  int64_t* dproj = st.out(proj, 256);
  LAB_HIP(c, hipMemsetAsync(dproj, 0, 256 * 8, c->stream));
  LAB_HIP(c, launch_jl_project(dproj, dpi, ds, n, r, c->stream));
  return st.finish();
}
