/*
 * fastcpu.c — baseline B1, the "honest-fast" CPU prover, and the CPU side of
 * BASELINE config 1 (benches/ntt_speed.rs:51-81 restated).  TEST / MEASUREMENT INFRASTRUCTURE, like the rest of
 * oracle/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product never does.
 *
 * B0 (labrador_oracle.c, orc_polymul_faithful) mirrors how the reference spends its time: an i128 `%` after every
 * coefficient operation, a schoolbook 64x64 product and a term-by-term fold per ring product.  B1 is what a careful
 * CPU implementation of the same prover does instead, i.e. the same algorithm the GPU path runs:
 *   - residues mod 1..3 NTT-friendly 30-bit primes (q itself when it is one below 2^32: path (b) of SURVEY §7),
 *   - every operand transformed once (64-point negacyclic NTT, Shoup butterflies), products slot-wise,
 *   - inner products accumulated in the transform domain in 64-bit registers (16 terms between reductions),
 *   - one inverse transform + Garner CRT + `mod q` per result,
 *   - the CRS matrices resident in memory in the transform domain (as on the GPU),
 *   - phi'' and phi never formed (their inner products with the witness are linear: X1 = phi'^T S, XW = W^T S),
 *   - OpenMP over the output rows of every step.
 * Its transcript equals orc_prove_core's bit for bit (tests/test_fastcpu.py), which is what makes it a baseline
 * for the same job rather than a different computation.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "labrador_oracle.h"

#define D 64
#define MAXL 3
typedef unsigned __int128 u128;
typedef __int128 i128;

typedef struct {
  uint32_t p;
  uint64_t mu;              /* floor(2^64 / p): Barrett constant for 64-bit values */
  uint32_t w[D], ws[D];     /* psi^bitrev(k) and Shoup companions, forward (Cooley-Tukey, natural in -> bit-reversed out) */
  uint32_t wi[D], wis[D];   /* inverse twiddles (Gentleman-Sande) */
  uint32_t ninv, ninvs;     /* 64^-1 mod p */
} fc_limb;

typedef struct fc_ctx {
  uint64_t q;
  int L;
  int direct;               /* q itself is the single NTT prime (q < 2^32, q == 1 mod 128) */
  fc_limb limb[MAXL];
  uint32_t ginv[MAXL][MAXL];  /* Garner: p_j^-1 mod p_i for j < i */
  int threads;
} fc_ctx;

static uint64_t powmod(uint64_t b, uint64_t e, uint64_t m) {
  u128 r = 1, x = b % m;
  while (e) { if (e & 1) r = r * x % m; x = x * x % m; e >>= 1; }
  return (uint64_t)r;
}
static int is_prime_u64(uint64_t n) {
  if (n < 2) return 0;
  static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (int i = 0; i < 12; i++) { if (n == small[i]) return 1; if (n % small[i] == 0) return 0; }
  uint64_t d = n - 1; int s = 0;
  while ((d & 1) == 0) { d >>= 1; s++; }
  for (int i = 0; i < 12; i++) {
    uint64_t x = powmod(small[i], d, n);
    if (x == 1 || x == n - 1) continue;
    int comp = 1;
    for (int k = 1; k < s && comp; k++) { x = (uint64_t)((u128)x * x % n); if (x == n - 1) comp = 0; }
    if (comp) return 0;
  }
  return 1;
}
static inline uint32_t shoup(uint32_t w, uint32_t p) { return (uint32_t)(((uint64_t)w << 32) / p); }
/* x * w mod p with w' = floor(w 2^32 / p); x < 2^32 (any), result in [0, p) */
static inline uint32_t mul_shoup(uint32_t x, uint32_t w, uint32_t ws, uint32_t p) {
  const uint64_t qh = ((uint64_t)x * ws) >> 32;
  const uint64_t r = (uint64_t)x * w - qh * p;          /* in [0, 2p): 64-bit so that p up to 2^32 works (direct mode) */
  return (uint32_t)(r >= p ? r - p : r);
}
static inline uint32_t barrett64(uint64_t t, const fc_limb *l) {   /* t mod p for any 64-bit t */
  uint64_t qh = (uint64_t)(((u128)t * l->mu) >> 64);
  uint64_t r = t - qh * l->p;
  while (r >= l->p) r -= l->p;
  return (uint32_t)r;
}
static int bitrev6(int k) { int r = 0; for (int b = 0; b < 6; b++) r |= ((k >> b) & 1) << (5 - b); return r; }

static int limb_init(fc_limb *l, uint32_t p) {
  l->p = p;
  l->mu = (uint64_t)((((u128)1) << 64) / p);
  /* a primitive 128th root of unity psi: g^((p-1)/128) for the first g with psi^64 == -1 */
  uint64_t psi = 0;
  for (uint64_t g = 2; g < 1000 && !psi; g++) {
    uint64_t c = powmod(g, (p - 1) / 128, p);
    if (powmod(c, 64, p) == (uint64_t)p - 1) psi = c;
  }
  if (!psi) return -1;
  uint64_t psi_inv = powmod(psi, p - 2, p);
  for (int k = 0; k < D; k++) {
    l->w[k] = (uint32_t)powmod(psi, bitrev6(k), p);
    l->ws[k] = shoup(l->w[k], p);
    l->wi[k] = (uint32_t)powmod(psi_inv, bitrev6(k), p);
    l->wis[k] = shoup(l->wi[k], p);
  }
  l->ninv = (uint32_t)powmod(64, p - 2, p);
  l->ninvs = shoup(l->ninv, p);
  return 0;
}

/* forward negacyclic NTT in place (natural order in, bit-reversed out), values in [0, p) */
static void ntt_fwd(uint32_t *a, const fc_limb *l) {
  const uint32_t p = l->p;
  int t = D, m = 1;
  while (m < D) {
    t >>= 1;
    for (int i = 0; i < m; i++) {
      const uint32_t w = l->w[m + i], ws = l->ws[m + i];
      uint32_t *x = a + 2 * i * t, *y = x + t;
      for (int j = 0; j < t; j++) {
        const uint64_t u = x[j], v = mul_shoup(y[j], w, ws, p);
        uint64_t s = u + v; s = s >= p ? s - p : s;
        uint64_t d = u + p - v; d = d >= p ? d - p : d;
        x[j] = (uint32_t)s; y[j] = (uint32_t)d;
      }
    }
    m <<= 1;
  }
}
/* inverse (bit-reversed in, natural out), scaled by 64^-1 */
static void ntt_inv(uint32_t *a, const fc_limb *l) {
  const uint32_t p = l->p;
  int t = 1, m = D;
  while (m > 1) {
    const int h = m >> 1;
    for (int i = 0; i < h; i++) {
      const uint32_t w = l->wi[h + i], ws = l->wis[h + i];
      uint32_t *x = a + 2 * i * t, *y = x + t;
      for (int j = 0; j < t; j++) {
        const uint64_t u = x[j], v = y[j];
        uint64_t s = u + v; s = s >= p ? s - p : s;
        uint64_t d = u + p - v; d = d >= p ? d - p : d;
        x[j] = (uint32_t)s; y[j] = mul_shoup((uint32_t)d, w, ws, p);
      }
    }
    t <<= 1; m = h;
  }
  for (int j = 0; j < D; j++) a[j] = mul_shoup(a[j], l->ninv, l->ninvs, p);
}

static const uint32_t FC_PRIMES[MAXL] = {1073741441u, 1073739649u, 1073738753u};   /* == 1 mod 128, the GPU path's limb primes */

fc_ctx *fc_create(uint64_t q, uint64_t max_len, int threads) {
  fc_ctx *c = (fc_ctx *)calloc(1, sizeof *c);
  c->q = q;
#ifdef _OPENMP
  c->threads = threads > 0 ? threads : omp_get_max_threads();
#else
  c->threads = 1;
#endif
  if (max_len == 0) max_len = 65536;
  if (q < ((uint64_t)1 << 32) && q % 128 == 1 && is_prime_u64(q)) {
    c->direct = 1; c->L = 1;
    if (limb_init(&c->limb[0], (uint32_t)q)) { free(c); return NULL; }
    return c;
  }
  /* smallest limb count with M/2 >= max_len * 64 * (q-1)^2 */
  long double need = (long double)max_len * 64.0L * (long double)(q - 1) * (long double)(q - 1), M = 1.0L;
  for (c->L = 0; c->L < MAXL;) {
    M *= (long double)FC_PRIMES[c->L];
    if (limb_init(&c->limb[c->L], FC_PRIMES[c->L])) { free(c); return NULL; }
    c->L++;
    if (M / 2 >= need) break;
  }
  if (M / 2 < need) { free(c); return NULL; }
  for (int i = 0; i < c->L; i++)
    for (int j = 0; j < i; j++) c->ginv[i][j] = (uint32_t)powmod(c->limb[j].p, c->limb[i].p - 2, c->limb[i].p);
  return c;
}
void fc_destroy(fc_ctx *c) { free(c); }
int fc_limbs(const fc_ctx *c) { return c->L; }
int fc_threads(const fc_ctx *c) { return c->threads; }

/* hat arrays: uint32 [L][count][64] (plane stride = count*64) */
void fc_to_hat(const fc_ctx *c, const uint64_t *coeffs, size_t count, uint32_t *hat) {
#pragma omp parallel for schedule(static) num_threads(c->threads)
  for (size_t k = 0; k < count; k++)
    for (int l = 0; l < c->L; l++) {
      uint32_t *h = hat + ((size_t)l * count + k) * D;
      for (int j = 0; j < D; j++) h[j] = barrett64(coeffs[k * D + j], &c->limb[l]);
      ntt_fwd(h, &c->limb[l]);
    }
}
/* the shared synthetic stream (orc_synth_fill / lab_vec_fill_synthetic) straight into the transform domain */
void fc_fill_synthetic_hat(const fc_ctx *c, uint64_t seed, uint64_t first_elem, size_t count, uint64_t bound, uint32_t *hat) {
#pragma omp parallel for schedule(static) num_threads(c->threads)
  for (size_t k = 0; k < count; k++) {
    uint64_t v[D];
    orc_synth_fill(seed, (first_elem + k) * D, D, bound, v);
    for (int l = 0; l < c->L; l++) {
      uint32_t *h = hat + ((size_t)l * count + k) * D;
      for (int j = 0; j < D; j++) h[j] = barrett64(v[j], &c->limb[l]);
      ntt_fwd(h, &c->limb[l]);
    }
  }
}
/* uniform residues straight in the transform domain (a uniform matrix is uniform in either domain): what bench.py fills
 * the 6.4e9-coefficient B with for the B1 timing — transforming that many polynomials on the host would take longer
 * than the measurement itself, and the prover's time does not depend on the values */
void fc_fill_random_hat(const fc_ctx *c, uint64_t seed, size_t count, uint32_t *hat) {
  const size_t words = count * D;
  for (int l = 0; l < c->L; l++) {
    uint32_t *h = hat + (size_t)l * words;
    const uint64_t p = c->limb[l].p;
#pragma omp parallel for schedule(static) num_threads(c->threads)
    for (size_t k = 0; k < words; k += 2) {
      const uint64_t x = orc_splitmix64(seed + ((uint64_t)l << 56) + k);
      h[k] = (uint32_t)(((x & 0xFFFFFFFFu) * p) >> 32);
      if (k + 1 < words) h[k + 1] = (uint32_t)(((x >> 32) * p) >> 32);
    }
  }
}
/* exact signed value of the residues (Garner), reduced mod q */
static inline uint64_t crt_mod_q(const fc_ctx *c, const uint32_t r[MAXL]) {
  if (c->direct) return r[0];
  uint32_t d[MAXL];
  for (int i = 0; i < c->L; i++) {
    const fc_limb *li = &c->limb[i];
    uint64_t t = r[i];
    /* t = (r_i - (d_0 + d_1 p_0 + ...)) * (p_0 ... p_{i-1})^-1 mod p_i, mixed-radix digit by digit */
    for (int j = 0; j < i; j++) {
      uint64_t dj = d[j] % li->p;
      t = (t + li->p - dj) % li->p;
      t = (uint64_t)((u128)t * c->ginv[i][j] % li->p);
    }
    d[i] = (uint32_t)t;
  }
  u128 v = 0, M = 1;
  for (int i = c->L - 1; i >= 0; i--) v = v * c->limb[i].p + d[i];
  for (int i = 0; i < c->L; i++) M *= c->limb[i].p;
  if (v > (M - 1) / 2) {            /* negative representative */
    u128 mag = M - v;
    uint64_t m = (uint64_t)(mag % c->q);
    return m ? c->q - m : 0;
  }
  return (uint64_t)(v % c->q);
}
void fc_from_hat(const fc_ctx *c, const uint32_t *hat, size_t count, uint64_t *coeffs) {
#pragma omp parallel for schedule(static) num_threads(c->threads)
  for (size_t k = 0; k < count; k++) {
    uint32_t tmp[MAXL][D];
    for (int l = 0; l < c->L; l++) {
      memcpy(tmp[l], hat + ((size_t)l * count + k) * D, D * 4);
      ntt_inv(tmp[l], &c->limb[l]);
    }
    for (int j = 0; j < D; j++) {
      uint32_t r[MAXL] = {0, 0, 0};
      for (int l = 0; l < c->L; l++) r[l] = tmp[l][j];
      coeffs[k * D + j] = crt_mod_q(c, r);
    }
  }
}

/* out[64] (one limb) = sum_{k<len} a[k*astride ..+64] (.) b[k*bstride ..+64]  slot-wise mod p; 16 products between folds */
__attribute__((target_clones("avx2", "default")))
static void mac_row(const fc_limb *l, const uint32_t *a, size_t astride, const uint32_t *b, size_t bstride, size_t len, uint32_t *out) {
  uint64_t acc[D];
  for (int j = 0; j < D; j++) acc[j] = 0;
  size_t k = 0;
  const size_t burst = l->p <= (1u << 30) ? 16 : 1;      /* products < 2^60 (limb primes) resp. < 2^64 (q itself, direct mode) */
  while (k < len) {
    const size_t end = k + burst < len ? k + burst : len;
    for (; k < end; k++) {
      const uint32_t *x = a + k * astride, *y = b + k * bstride;
      for (int j = 0; j < D; j++) acc[j] += (uint64_t)x[j] * y[j];     /* < 16 * 2^60 */
    }
    for (int j = 0; j < D; j++) acc[j] = barrett64(acc[j], l);
  }
  for (int j = 0; j < D; j++) out[j] = (uint32_t)acc[j];
}
__attribute__((target_clones("avx2", "default")))
static void add_row(const fc_limb *l, uint32_t *acc, const uint32_t *x) {
  for (int j = 0; j < D; j++) { uint64_t s = (uint64_t)acc[j] + x[j]; acc[j] = (uint32_t)(s >= l->p ? s - l->p : s); }
}
__attribute__((target_clones("avx2", "default")))
static void mul_row(const fc_limb *l, const uint32_t *x, const uint32_t *y, uint32_t *out) {
  for (int j = 0; j < D; j++) out[j] = barrett64((uint64_t)x[j] * y[j], l);
}

/* negacyclic product through the transform (config 1's "NTT" arm): one product, nothing reused */
void fc_polymul(const fc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out) {
  uint32_t ha[MAXL][D], hb[MAXL][D];
  for (int l = 0; l < c->L; l++) {
    for (int j = 0; j < D; j++) { ha[l][j] = barrett64(a[j], &c->limb[l]); hb[l][j] = barrett64(b[j], &c->limb[l]); }
    ntt_fwd(ha[l], &c->limb[l]);
    ntt_fwd(hb[l], &c->limb[l]);
    mul_row(&c->limb[l], ha[l], hb[l], ha[l]);
    ntt_inv(ha[l], &c->limb[l]);
  }
  for (int j = 0; j < D; j++) {
    uint32_t r[MAXL] = {0, 0, 0};
    for (int l = 0; l < c->L; l++) r[l] = ha[l][j];
    out[j] = crt_mod_q(c, r);
  }
}

static int cmp_double(const void *x, const void *y) { double a = *(const double *)x, b = *(const double *)y; return a < b ? -1 : a > b; }
static double now_ns(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e9 + ts.tv_nsec; }
/* benches/ntt_speed.rs:51-81 restated: one pair of uniform polynomials, `samples` timed single products per arm, median ns.
 * arm 0: classic (schoolbook with an i128 `%` per operation + fold = orc_polymul_faithful), arm 1: through the NTT. */
double fc_bench_polymul(uint64_t q, int arm, int samples, uint64_t seed) {
  uint64_t a[D], b[D], o[D];
  orc_synth_fill(seed, 0, D, q, a);
  orc_synth_fill(seed, D, D, q, b);
  fc_ctx *c = fc_create(q, 1, 1);
  if (!c) return -1.0;
  double *t = (double *)malloc(sizeof(double) * (size_t)samples);
  volatile uint64_t sink = 0;
  for (int s = -10; s < samples; s++) {
    const double t0 = now_ns();
    if (arm == 0) orc_polymul_faithful(q, a, b, o); else fc_polymul(c, a, b, o);
    const double t1 = now_ns();
    sink += o[0];
    if (s >= 0) t[s] = t1 - t0;
  }
  qsort(t, (size_t)samples, sizeof(double), cmp_double);
  const double med = t[samples / 2];
  free(t);
  fc_destroy(c);
  return med;
}

/* ---- the prover (same steps and outputs as orc_prove_core; CRS matrices in the transform domain) ---- */
typedef struct {
  const orc_prove_in *in;      /* A, B, C, Dm fields are ignored */
  const uint32_t *A, *B, *C, *Dm;   /* hat arrays [L][count][64] with the element order of orc_prove_in */
} fc_prove_in;

static inline size_t pair_index(size_t i, size_t j, size_t r) { return i * r - (i * (i - 1)) / 2 + (j - i); }
static uint64_t inv2_mod(uint64_t q) { return (q + 1) / 2; }
#define HAT(name, count) uint32_t *name = (uint32_t *)malloc((size_t)c->L * (count) * D * 4)
#define PL(ptr, count, l) ((ptr) + (size_t)(l) * (count) * D)

int fc_prove(const fc_ctx *c, const fc_prove_in *fin, orc_prove_out *out) {
  const orc_prove_in *in = fin->in;
  const uint64_t q = c->q;
  const size_t n = in->n, r = in->r, kappa = in->kappa, k1 = in->kappa1, k2 = in->kappa2;
  const int t1 = in->t1, t2 = in->t2, L = c->L, T = c->threads;
  if (in->q != q || k1 != k2 || t1 < 1 || t2 < 1 || in->b1 < 2 || in->b2 < 2) return -1;
  const size_t nr = n * r, rr = r * r, rk = r * kappa, pairs = r * (r + 1) / 2;
  (void)T;
  HAT(hS, nr); HAT(hphi, nr);
  fc_to_hat(c, in->S, nr, hS);
  fc_to_hat(c, in->phi, nr, hphi);

  /* step 1: t[i][rho] = <A[rho,:], s_i> */
  HAT(hT, rk);
#pragma omp parallel for collapse(2) schedule(static) num_threads(T)
  for (size_t rho = 0; rho < kappa; rho++)
    for (size_t i = 0; i < r; i++)
      for (int l = 0; l < L; l++)
        mac_row(&c->limb[l], PL(fin->A, kappa * n, l) + rho * n * D, D, PL(hS, nr, l) + i * D, r * D, n, PL(hT, rk, l) + (i * kappa + rho) * D);
  fc_from_hat(c, hT, rk, out->t);
  free(hT);

  /* Gram-type products out[i][j] = sum_x P[x][i] S[x][j] */
#define GRAM(hout, hP)                                                                                          \
  _Pragma("omp parallel for collapse(2) schedule(static) num_threads(T)")                                       \
  for (size_t i = 0; i < r; i++)                                                                                \
    for (size_t j = 0; j < r; j++)                                                                              \
      for (int l = 0; l < L; l++)                                                                               \
        mac_row(&c->limb[l], PL(hP, nr, l) + i * D, r * D, PL(hS, nr, l) + j * D, r * D, n, PL(hout, rr, l) + (i * r + j) * D);
  /* step 2: G = S^T S */
  HAT(hG, rr);
  GRAM(hG, hS)
  fc_from_hat(c, hG, rr, out->G);

  /* step 3: lhs = sum_{i,k} B_ik t_i^(k) */
  const size_t nmat = r * (size_t)t1;
  uint64_t *tdec = (uint64_t *)malloc(nmat * kappa * D * 8);     /* [i][k][rho] */
  int bad = 0;
#pragma omp parallel for schedule(static) num_threads(T)
  for (size_t i = 0; i < r; i++)
    if (orc_decompose_vec(q, out->t + i * kappa * D, kappa, in->b1, t1, tdec + i * t1 * kappa * D)) bad = 1;
  if (bad) return -8;
  HAT(htdec, nmat * kappa);
  fc_to_hat(c, tdec, nmat * kappa, htdec);
  free(tdec);
  HAT(hu1, k1);
#pragma omp parallel for schedule(dynamic, 1) num_threads(T)
  for (size_t rho = 0; rho < k1; rho++)
    for (int l = 0; l < L; l++) {
      uint32_t *o = PL(hu1, k1, l) + rho * D, part[D];
      memset(o, 0, D * 4);
      for (size_t m = 0; m < nmat; m++) {
        mac_row(&c->limb[l], PL(fin->B, nmat * k1 * kappa, l) + ((m * k1 + rho) * kappa) * D, D, PL(htdec, nmat * kappa, l) + m * kappa * D, D, kappa, part);
        add_row(&c->limb[l], o, part);
      }
    }
  free(htdec);

  /* digits of the upper triangle of an r x r matrix in [pair][k] order, transformed */
#define UPPER_DIGITS(hdst, src, base, exp)                                                         \
  {                                                                                                \
    uint64_t *dec = (uint64_t *)malloc(pairs * (size_t)(exp) * D * 8);                             \
    for (size_t i = 0; i < r; i++)                                                                 \
      for (size_t j = i; j < r; j++)                                                               \
        if (orc_decompose(q, (src) + (i * r + j) * D, (base), (exp), dec + pair_index(i, j, r) * (size_t)(exp) * D)) bad = 1; \
    fc_to_hat(c, dec, pairs * (size_t)(exp), hdst);                                                \
    free(dec);                                                                                     \
  }
  /* out[rho] (+)= sum_m dig[m] * Cm[m][rho] */
#define COMMIT(hout, hdig, hC, nterms, accumulate)                                                 \
  _Pragma("omp parallel for schedule(static) num_threads(T)")                                      \
  for (size_t rho = 0; rho < k2; rho++)                                                            \
    for (int l = 0; l < L; l++) {                                                                  \
      uint32_t part[D];                                                                            \
      mac_row(&c->limb[l], PL(hdig, (nterms), l), D, PL(hC, (nterms) * k2, l) + rho * D, k2 * D, (nterms), part); \
      if (accumulate) add_row(&c->limb[l], PL(hout, k2, l) + rho * D, part);                       \
      else memcpy(PL(hout, k2, l) + rho * D, part, D * 4);                                         \
    }
  /* step 4: rhs = sum g_ij^(k) C_ijk ; u1 = lhs + rhs */
  HAT(hgdec, pairs * (size_t)t2);
  UPPER_DIGITS(hgdec, out->G, in->b2, t2)
  if (bad) return -8;
  COMMIT(hu1, hgdec, fin->C, pairs * (size_t)t2, 1)
  free(hgdec);
  fc_from_hat(c, hu1, k1, out->u1);
  free(hu1);

  /* step 5: JL projection over the integers, and W_i[x] = sum_m omega_m sigma_inv(pi_i^(m)[x]) in the same pass */
  const size_t cols = n * D;
  uint64_t *W = (uint64_t *)malloc(nr * D * 8);
  int64_t projt[256];
  memset(projt, 0, sizeof projt);
  for (size_t i = 0; i < r; i++) {
    const int8_t *pi = in->Pi + i * 256 * cols;
    int64_t *wraw = (int64_t *)calloc(cols, 8);          /* sum_m omega_m Pi[m][col] over the integers, |.| < 2^48 */
#pragma omp parallel num_threads(T)
    {
      int64_t myp[256];
      memset(myp, 0, sizeof myp);
#pragma omp for schedule(static)
      for (size_t cb = 0; cb < cols; cb += 1024) {
        const size_t ce = cb + 1024 < cols ? cb + 1024 : cols;
        for (int m = 0; m < 256; m++) {
          const int8_t *row = pi + (size_t)m * cols;
          const int64_t om = (int64_t)in->omega[m];
          int64_t acc = 0;
          for (size_t col = cb; col < ce; col++) {
            const int64_t e = row[col];
            acc += e * (int64_t)in->S[((col / D) * r + i) * D + (col % D)];
            wraw[col] += e * om;
          }
          myp[m] += acc;
        }
      }
#pragma omp critical
      for (int m = 0; m < 256; m++) projt[m] += myp[m];
    }
    for (size_t x = 0; x < n; x++)
      for (int k = 0; k < D; k++) {
        const int64_t v = wraw[x * D + k];
        const uint64_t mag = (uint64_t)(v < 0 ? -v : v) % q;
        const uint64_t u = v < 0 ? (mag ? q - mag : 0) : mag;
        W[(x * r + i) * D + (k == 0 ? 0 : D - k)] = k == 0 ? u : (u ? q - u : 0);
      }
    free(wraw);
  }
  memcpy(out->proj_int, projt, sizeof projt);
  out->jl_valid = orc_l2_norm_i64(out->proj_int, 256) <= sqrt(128.) * (double)in->beta_bound;
  for (int m = 0; m < 256; m++) { i128 v = out->proj_int[m] % (i128)q; if (v < 0) v += q; out->proj[m] = (uint64_t)v; }

  /* steps 6-8 by linearity: X1 = phi'^T S, XW = W^T S, X2 = psi X1 + XW, b'' = <psi a', G> + trace(X2),
   * Phi^T S = alpha X1 + beta X2, H = 2^-1 (Phi^T S + S^T Phi) */
  HAT(hW, nr);
  fc_to_hat(c, W, nr, hW);
  free(W);
  HAT(hX1, rr); HAT(hXW, rr);
  GRAM(hX1, hphi)
  GRAM(hXW, hW)
  free(hW);
  uint64_t *X1 = (uint64_t *)malloc(rr * D * 8), *X2 = (uint64_t *)malloc(rr * D * 8), *app = (uint64_t *)malloc(rr * D * 8);
  fc_from_hat(c, hX1, rr, X1);
  fc_from_hat(c, hXW, rr, X2);
  const uint64_t psi = in->psi % q;
  for (size_t e = 0; e < rr * D; e++) {
    X2[e] = (uint64_t)(((u128)psi * X1[e] + X2[e]) % q);
    app[e] = (uint64_t)((u128)psi * in->a[e] % q);
  }
  HAT(happ, rr); HAT(hGc, rr);
  fc_to_hat(c, app, rr, happ);
  fc_to_hat(c, out->G, rr, hGc);
  HAT(hbl, 1);
  for (int l = 0; l < L; l++) mac_row(&c->limb[l], PL(happ, rr, l), D, PL(hGc, rr, l), D, rr, PL(hbl, 1, l));
  uint64_t bl[D];
  fc_from_hat(c, hbl, 1, bl);
  for (int k = 0; k < D; k++) {
    u128 s = bl[k];
    for (size_t i = 0; i < r; i++) s += X2[(i * r + i) * D + k];
    out->bpp[k] = (uint64_t)(s % q);
  }
  {
    u128 chk = 0;
    for (int m = 0; m < 256; m++) chk = (chk + (u128)(in->omega[m] % q) * out->proj[m]) % q;
    chk = (chk + (u128)psi * (in->b_prime % q)) % q;
    out->bpp_ok = out->bpp[0] == (uint64_t)chk;
  }
  uint32_t hal[MAXL][D], hbe[MAXL][D];
  {
    HAT(tmp, 1);
    fc_to_hat(c, in->alpha, 1, tmp);
    for (int l = 0; l < L; l++) memcpy(hal[l], PL(tmp, 1, l), D * 4);
    fc_to_hat(c, in->beta, 1, tmp);
    for (int l = 0; l < L; l++) memcpy(hbe[l], PL(tmp, 1, l), D * 4);
    free(tmp);
  }
  fc_to_hat(c, X1, rr, hX1);
  fc_to_hat(c, X2, rr, hXW);
  for (size_t e = 0; e < rr; e++)
    for (int l = 0; l < L; l++) {
      uint32_t p1[D], p2[D];
      mul_row(&c->limb[l], PL(hX1, rr, l) + e * D, hal[l], p1);
      mul_row(&c->limb[l], PL(hXW, rr, l) + e * D, hbe[l], p2);
      add_row(&c->limb[l], p1, p2);
      memcpy(PL(hX1, rr, l) + e * D, p1, D * 4);
    }
  fc_from_hat(c, hX1, rr, X1);           /* Phi^T S */
  const uint64_t half = inv2_mod(q);
  for (size_t i = 0; i < r; i++)
    for (size_t j = 0; j < r; j++)
      for (int k = 0; k < D; k++)
        out->H[(i * r + j) * D + k] = (uint64_t)((u128)half * (((u128)X1[(i * r + j) * D + k] + X1[(j * r + i) * D + k]) % q) % q);
  free(X1); free(X2); free(app); free(hX1); free(hXW); free(happ); free(hGc); free(hbl); free(hG);

  /* step 9: u2 = sum h_ij^(k) D_ijk */
  HAT(hhdec, pairs * (size_t)t1);
  UPPER_DIGITS(hhdec, out->H, in->b1, t1)
  if (bad) return -8;
  HAT(hu2, k2);
  COMMIT(hu2, hhdec, fin->Dm, pairs * (size_t)t1, 0)
  free(hhdec);
  fc_from_hat(c, hu2, k2, out->u2);
  free(hu2);

  /* step 10: z[x] = sum_i c_i s_i[x] */
  HAT(hc, r); HAT(hz, n);
  fc_to_hat(c, in->c, r, hc);
#pragma omp parallel for schedule(static) num_threads(T)
  for (size_t x = 0; x < n; x++)
    for (int l = 0; l < L; l++) mac_row(&c->limb[l], PL(hS, nr, l) + x * r * D, D, PL(hc, r, l), D, r, PL(hz, n, l) + x * D);
  fc_from_hat(c, hz, n, out->z);
  free(hc); free(hz); free(hS); free(hphi);
  return 0;
}
