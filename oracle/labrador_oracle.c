/*
 * labrador_oracle.c — CPU restatement of the LaBRADOR-SNARK prover hot path.
 * TEST INFRASTRUCTURE ONLY (see labrador_oracle.h for the rules and for the
 * parity-pinning status: value-level parity with the Rust binary is UNPINNED,
 * the oracle is pinned by the reference's property tests, its CI invariant and
 * an independent Python twin).
 *
 * Citations are file:line in the reference tree (RatioAeterna/LaBRADOR-SNARK).
 */
#include "labrador_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef __int128 i128;

#define D ORC_D

/* ------------------------------------------------------------------ Zq */

/* src/util.rs:16-23 mod_positive: truncating remainder, then lift into [0,q) */
static inline uint64_t modpos(i128 v, uint64_t q) {
  i128 r = v % (i128)q;
  if (r < 0) r += (i128)q;
  return (uint64_t)r;
}

uint64_t orc_mod_positive_i128(uint64_t lo, int64_t hi, uint64_t q) {
  i128 v = ((i128)hi << 64) | (i128)(u128)lo;
  return modpos(v, q);
}

/* src/algebraic.rs:56-63 Neg: 0 -> 0, v -> Q - v */
static inline uint64_t zq_neg(uint64_t v, uint64_t q) { return v == 0 ? 0 : q - v; }
static inline uint64_t zq_add(uint64_t a, uint64_t b, uint64_t q) { return modpos((i128)a + (i128)b, q); }   /* :95-101 */
static inline uint64_t zq_sub(uint64_t a, uint64_t b, uint64_t q) { return modpos((i128)a - (i128)b, q); }   /* :111-118 */
static inline uint64_t zq_mul(uint64_t a, uint64_t b, uint64_t q) { return modpos((i128)a * (i128)b, q); }   /* :120-126 */

/* ------------------------------------------------------------------ Rq */

int orc_trimmed_len(const uint64_t *p) { /* Polynomial::new strips trailing zeros; algebraic.rs:303-315 */
  int len = D;
  while (len > 0 && p[len - 1] == 0) len--;
  return len;
}

/* algebraic.rs:402-403 (schoolbook over Zq, up to 127 coefficients) then
 * Rq::reduction :352-376: c[deg>=64] folds into c[deg mod 64] with sign (-1)^(deg/64).
 * Exact sums are taken before the single reduction: identical result, since
 * every Zq op in the reference is the same residue class. */
void orc_polymul(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out) {
  u128 lo[D], hi[D];
  memset(lo, 0, sizeof lo);
  memset(hi, 0, sizeof hi);
  for (int i = 0; i < D; i++) {
    if (a[i] == 0) continue;
    for (int j = 0; j < D; j++) {
      u128 pr = (u128)a[i] * b[j];
      int k = i + j;
      if (k < D) lo[k] += pr; else hi[k - D] += pr;
    }
  }
  for (int k = 0; k < D; k++) {
    uint64_t p = (uint64_t)(lo[k] % q), m = (uint64_t)(hi[k] % q);
    out[k] = p >= m ? p - m : p + q - m;
  }
}

/* The same product computed with the reference's per-operation reductions:
 * every coefficient product and every accumulate goes through Zq::new
 * (algebraic.rs:30-38 -> util.rs:16-23, an i128 `%`), then the fold adds the
 * 63 high terms one at a time (algebraic.rs:364-374).  Timed as baseline B0. */
void orc_polymul_faithful(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out) {
  uint64_t c[2 * D - 1];
  memset(c, 0, sizeof c);
  for (int i = 0; i < D; i++)
    for (int j = 0; j < D; j++) {
      uint64_t pr = zq_mul(a[i], b[j], q);
      c[i + j] = zq_add(c[i + j], pr, q);
    }
  uint64_t red[D];
  memcpy(red, c, sizeof red);
  for (int deg = D; deg < 2 * D - 1; deg++) {
    i128 factor = ((deg / D) & 1) ? -1 : 1;                       /* :367 */
    uint64_t term = modpos((i128)c[deg] * factor, q);             /* :371 */
    red[deg % D] = zq_add(red[deg % D], term, q);                 /* :373 */
  }
  memcpy(out, red, sizeof red);
}

/* algebraic.rs:384-399: operands as u64, PLAN.negacyclic_polymul (exact
 * integer negacyclic convolution, result taken mod 2^64 — concrete-ntt 0.1.1
 * native64::Plan32, source not in the reference tree), then :417-420 two's
 * complement reinterpretation and Zq::new. */
void orc_polymul_ntt_branch(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out) {
  uint64_t acc[D];
  memset(acc, 0, sizeof acc);
  for (int i = 0; i < D; i++)
    for (int j = 0; j < D; j++) {
      uint64_t pr = a[i] * b[j]; /* wraps mod 2^64 */
      int k = i + j;
      if (k < D) acc[k] += pr; else acc[k - D] -= pr;
    }
  for (int k = 0; k < D; k++) {
    i128 v = (acc[k] >> 63) ? (i128)acc[k] - ((i128)1 << 64) : (i128)acc[k];
    out[k] = modpos(v, q);
  }
}

void orc_poly_add(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out) {
  for (int k = 0; k < D; k++) out[k] = zq_add(a[k], b[k], q);
}
void orc_poly_sub(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out) {
  for (int k = 0; k < D; k++) out[k] = zq_sub(a[k], b[k], q);
}
void orc_poly_scale(uint64_t q, const uint64_t *p, uint64_t s, uint64_t *out) { /* util.rs:176-180 */
  for (int k = 0; k < D; k++) out[k] = zq_mul(p[k], s, q);
}
void orc_sigma_inv(uint64_t q, const uint64_t *a, uint64_t *out) { /* util.rs:118-137 */
  uint64_t t[D];
  t[0] = a[0];
  for (int deg = 1; deg < D; deg++) t[D - deg] = zq_neg(a[deg], q);
  memcpy(out, t, sizeof t);
}

/* ------------------------------------------------------- vector helpers */

void orc_inner_product(uint64_t q, const uint64_t *v1, const uint64_t *v2, size_t len, uint64_t *out) {
  uint64_t acc[D], pr[D]; /* util.rs:503-508 */
  memset(acc, 0, sizeof acc);
  for (size_t i = 0; i < len; i++) {
    orc_polymul(q, v1 + i * D, v2 + i * D, pr);
    orc_poly_add(q, acc, pr, acc);
  }
  memcpy(out, acc, sizeof acc);
}

void orc_poly_by_poly_vec(uint64_t q, const uint64_t *p, const uint64_t *v, size_t len, uint64_t *out) {
  for (size_t i = 0; i < len; i++) orc_polymul(q, p, v + i * D, out + i * D); /* util.rs:289-296 */
}

void orc_add_poly_vec(uint64_t q, const uint64_t *a, const uint64_t *b, size_t len, uint64_t *out) {
  for (size_t i = 0; i < len; i++) orc_poly_add(q, a + i * D, b + i * D, out + i * D); /* util.rs:298-309 */
}

/* util.rs:377-387 centered_rep: `val > b/2` compares against mod_positive(b/2,Q)
 * (algebraic.rs:293-297); the "centred" value is (b - val) % b lifted into Zq,
 * i.e. NON-NEGATIVE — the sign is lost. */
static inline uint64_t centered_rep(uint64_t val, int64_t b, uint64_t q) {
  if (val > modpos((i128)(b / 2), q)) return modpos(((i128)b - (i128)val) % (i128)b, q);
  return val;
}

int orc_decompose(uint64_t q, const uint64_t *p, int64_t base, int exp, uint64_t *out) {
  if (base < 2 || exp < 0) return -1; /* base 1 never terminates in the reference (util.rs:410-417) */
  memset(out, 0, (size_t)exp * D * sizeof(uint64_t));
  for (int deg = 0; deg < D; deg++) {
    uint64_t v = p[deg];
    int k = 0;
    while (v != 0) {                                                    /* util.rs:410 */
      uint64_t r0 = modpos((i128)v % (i128)base, q);                    /* :413, algebraic.rs:75-81 */
      uint64_t r = centered_rep(r0, base, q);
      if (k < exp) out[(size_t)k * D + deg] = r;                        /* :425-440 keeps the first exp digits */
      k++;
      v = modpos((i128)zq_sub(v, r, q) / (i128)base, q);                /* :415, integer division algebraic.rs:136-142 */
      if (k > 200) return -2;
    }
  }
  return 0;
}

int orc_decompose_vec(uint64_t q, const uint64_t *v, size_t len, int64_t base, int exp, uint64_t *out) {
  uint64_t *tmp = (uint64_t *)malloc((size_t)(exp > 0 ? exp : 1) * D * sizeof(uint64_t));
  for (size_t i = 0; i < len; i++) {
    int rc = orc_decompose(q, v + i * D, base, exp, tmp);
    if (rc) { free(tmp); return rc; }
    for (int k = 0; k < exp; k++) memcpy(out + ((size_t)k * len + i) * D, tmp + (size_t)k * D, D * sizeof(uint64_t)); /* util.rs:366-370 */
  }
  free(tmp);
  return 0;
}

double orc_poly_norm(const uint64_t *p, uint64_t *lo, uint64_t *hi) { /* util.rs:195-202 */
  u128 s = 0;
  for (int k = 0; k < D; k++) s += (u128)p[k] * p[k];
  if (lo) *lo = (uint64_t)s;
  if (hi) *hi = (uint64_t)(s >> 64);
  return (double)(i128)s; /* `norm as f64`, round-to-nearest-even like Rust's cast */
}

double orc_vec_poly_norm_squared(const uint64_t *v, size_t len) { /* util.rs:188-192: sequential f64 sum */
  double s = 0.0;
  for (size_t i = 0; i < len; i++) s += orc_poly_norm(v + i * D, NULL, NULL);
  return s;
}

double orc_l2_norm_i64(const int64_t *v, size_t len) { /* util.rs:216-219 */
  i128 s = 0;
  for (size_t i = 0; i < len; i++) s += (i128)v[i] * v[i];
  return sqrt((double)s);
}

void orc_jl_accumulate(const int8_t *pi, const uint64_t *s, size_t cols, int64_t *acc) {
  for (int m = 0; m < 256; m++) { /* util.rs:511-526 matmul, 259-272 vec_inner_product: plain integers */
    i128 t = 0;
    const int8_t *row = pi + (size_t)m * cols;
    for (size_t c = 0; c < cols; c++) t += (i128)row[c] * (i128)s[c];
    acc[m] += (int64_t)t;
  }
}

/* -------------------------------------------------- RuntimeConstants */

void orc_runtime_constants(uint64_t q, uint64_t N, uint64_t R, orc_constants *o) {
  const double TAU = 71.0; /* src/constants.rs:22 */
  double Q = (double)q;
  o->n = N; o->r = R;
  o->kappa = o->kappa1 = o->kappa2 = N * D;                                          /* :237-239 */
  o->beta_bound = (int64_t)floor(sqrt(30.0 / 128.0) * Q / 125.0);                    /* :241 */
  o->std = (double)o->beta_bound / sqrt((double)(R * N * D));                        /* :242 */
  o->b = (int64_t)round(sqrt(sqrt(12. * (double)R * TAU) * o->std));                 /* :243 */
  o->t1 = (int64_t)round(log2(Q) / log2((double)o->b));                              /* :244 */
  o->b1 = (int64_t)pow(Q, 1.0 / (double)o->t1);                                      /* :245 (truncating cast) */
  o->t2 = (int64_t)round(log2(sqrt(24. * (double)(N * D)) * (o->std * o->std)) / log2((double)o->b)); /* :246 */
  o->b2 = (int64_t)round(pow(sqrt((double)(24 * (N * D))) * (o->std * o->std), 1.0 / (double)o->t2)); /* :247 */
  double b1 = (double)o->b1, b2 = (double)o->b2, t1 = (double)o->t1, t2 = (double)o->t2, r = (double)R;
  o->gamma = (double)(o->beta_bound * o->beta_bound) * TAU;                          /* :250 */
  o->gamma1 = ((b1 * b1 * t1) / 12.0) * r * (double)o->kappa * (double)D +
              ((b2 * b2 * t2) / 12.0) * ((r * r + r) / 2.0) * (double)D;             /* :251 */
  o->gamma2 = ((b1 * b1 * t1) / 12.0) * ((r * r + r) / 2.0) * (double)D;             /* :252 */
  o->beta_prime = (2.0 / ((double)o->b * (double)o->b)) * o->gamma + o->gamma1 + o->gamma2; /* :254 */
}

/* ------------------------------------------------ synthetic inputs */

uint64_t orc_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

uint64_t orc_synth_residue(uint64_t seed, uint64_t idx, uint64_t bound) {
  uint64_t h = orc_splitmix64(orc_splitmix64(seed) ^ (idx * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull));
  return (uint64_t)(((u128)h * bound) >> 64);
}

void orc_synth_fill(uint64_t seed, uint64_t first_idx, size_t count, uint64_t bound, uint64_t *out) {
  for (size_t i = 0; i < count; i++) out[i] = orc_synth_residue(seed, first_idx + i, bound);
}

/* ------------------------------------------------ prover core */

static inline size_t pair_index(size_t i, size_t j, size_t r) { /* structs.rs:101-106: i*R - i(i-1)/2 + (j-i) */
  return i * r - (i * (i - 1)) / 2 + (j - i);
}

/* column i of an [n][r][64] array as a contiguous [n][64] vector (proofgen.rs:45 `witness.column(i).to_vec()`) */
static uint64_t *column(const uint64_t *M, size_t n, size_t r, size_t i) {
  uint64_t *c = (uint64_t *)malloc(n * D * sizeof(uint64_t));
  for (size_t x = 0; x < n; x++) memcpy(c + x * D, M + (x * r + i) * D, D * sizeof(uint64_t));
  return c;
}

/* phi''_i (proofgen.rs:234-255): psi*phi'_i + sum_m omega[m]*sigma_inv(row m of Pi_i as n polys) */
static void phi_prime_prime(const orc_prove_in *in, size_t i, uint64_t *out /* [n][64] */) {
  uint64_t q = in->q;
  size_t n = in->n, r = in->r, cols = n * D;
  uint64_t tmp[D], lifted[D], conj[D];
  for (size_t x = 0; x < n; x++) orc_poly_scale(q, in->phi + (x * r + i) * D, in->psi, out + x * D); /* :236-242, util.rs:148-155 with L=1 */
  const int8_t *pi = in->Pi + i * 256 * cols;
  for (int m = 0; m < 256; m++) {
    for (size_t x = 0; x < n; x++) {
      for (int k = 0; k < D; k++) lifted[k] = modpos((i128)pi[(size_t)m * cols + x * D + k], q); /* proofgen.rs:445-451 Zq::lift; util.rs:472-482 */
      orc_sigma_inv(q, lifted, conj);                                                          /* :249 */
      orc_poly_scale(q, conj, in->omega[m], tmp);                                              /* :250-251 */
      orc_poly_add(q, out + x * D, tmp, out + x * D);                                          /* :252 */
    }
  }
}

static uint64_t inv2(uint64_t q) { /* proofgen.rs:341-346: 2^(Q-2) mod Q */
  u128 result = 1, base = 2 % q;
  uint64_t e = q - 2;
  while (e) {
    if (e & 1) result = result * base % q;
    base = base * base % q;
    e >>= 1;
  }
  return (uint64_t)result;
}

int orc_prove_core(const orc_prove_in *in, orc_prove_out *out) {
  uint64_t q = in->q;
  size_t n = in->n, r = in->r, kappa = in->kappa, k1 = in->kappa1, k2 = in->kappa2;
  int t1 = in->t1, t2 = in->t2;
  if (q < 3 || q >= (1ull << 60) || k1 != k2 || t1 < 1 || t2 < 1 || in->b1 < 2 || in->b2 < 2) return -1;
  uint64_t acc[D], pr[D];

  uint64_t **scol = (uint64_t **)malloc(r * sizeof *scol);
  for (size_t i = 0; i < r; i++) scol[i] = column(in->S, n, r, i);

  /* step 1, proofgen.rs:41-49: t_i[rho] = <A[rho,:], s_i> */
  for (size_t rho = 0; rho < kappa; rho++)
    for (size_t i = 0; i < r; i++)
      orc_inner_product(q, in->A + rho * n * D, scol[i], n, out->t + (i * kappa + rho) * D);

  /* step 2, :59-70: G[i][j] = <s_i, s_j> */
  for (size_t i = 0; i < r; i++)
    for (size_t j = 0; j < r; j++)
      orc_inner_product(q, scol[i], scol[j], n, out->G + (i * r + j) * D);

  /* step 3, :101-132: lhs = sum_{i,k} B_ik * t_i^(k) */
  uint64_t *lhs = (uint64_t *)calloc(k1 * D, sizeof(uint64_t));
  uint64_t *tdec = (uint64_t *)malloc((size_t)t1 * kappa * D * sizeof(uint64_t));
  for (size_t i = 0; i < r; i++) {
    int rc = orc_decompose_vec(q, out->t + i * kappa * D, kappa, in->b1, t1, tdec); /* :105-106 */
    if (rc) return rc;
    for (int k = 0; k < t1; k++) {
      const uint64_t *Bik = in->B + ((i * (size_t)t1 + (size_t)k) * k1) * kappa * D;
      for (size_t rho = 0; rho < k1; rho++) {
        orc_inner_product(q, Bik + rho * kappa * D, tdec + (size_t)k * kappa * D, kappa, pr); /* :115-116 */
        orc_poly_add(q, lhs + rho * D, pr, lhs + rho * D);                                     /* :127-131 */
      }
    }
  }
  free(tdec);

  /* step 4, :137-148: rhs = sum_{i<=j,k} g_ij^(k) * C_ijk */
  uint64_t *rhs = (uint64_t *)calloc(k2 * D, sizeof(uint64_t));
  uint64_t *gdec = (uint64_t *)malloc((size_t)t2 * D * sizeof(uint64_t));
  for (size_t i = 0; i < r; i++)
    for (size_t j = i; j < r; j++) {
      int rc = orc_decompose(q, out->G + (i * r + j) * D, in->b2, t2, gdec); /* :140 */
      if (rc) return rc;
      for (int k = 0; k < t2; k++) {
        const uint64_t *Cijk = in->C + ((pair_index(i, j, r) * (size_t)t2 + (size_t)k) * k2) * D;
        for (size_t rho = 0; rho < k2; rho++) {
          orc_polymul(q, gdec + (size_t)k * D, Cijk + rho * D, pr);           /* :144 */
          orc_poly_add(q, rhs + rho * D, pr, rhs + rho * D);                   /* :145 */
        }
      }
    }
  free(gdec);
  orc_add_poly_vec(q, lhs, rhs, k1, out->u1); /* :153 */
  free(lhs); free(rhs);

  /* step 5, :161-186 + 429-456: JL projection over the integers */
  memset(out->proj_int, 0, 256 * sizeof(int64_t));
  for (size_t i = 0; i < r; i++) orc_jl_accumulate(in->Pi + i * 256 * n * D, scol[i], n * D, out->proj_int);
  out->jl_valid = orc_l2_norm_i64(out->proj_int, 256) <= sqrt(128.) * (double)in->beta_bound; /* verification.rs:568-579 */
  for (int m = 0; m < 256; m++) out->proj[m] = modpos((i128)out->proj_int[m], q);             /* proofgen.rs:186 */

  /* step 6, :205-289 with upper_bound = 1, L = 1 */
  uint64_t *phipp = (uint64_t *)malloc(r * n * D * sizeof(uint64_t)); /* [r][n][64] */
  for (size_t i = 0; i < r; i++) phi_prime_prime(in, i, phipp + i * n * D);
  uint64_t bl[D], br[D], app[D], g[D];
  memset(bl, 0, sizeof bl);
  memset(br, 0, sizeof br);
  for (size_t i = 0; i < r; i++)
    for (size_t j = 0; j < r; j++) {
      orc_poly_scale(q, in->a + (i * r + j) * D, in->psi, app);   /* :228-230 a'' = multiply_poly_ints(a', psi) */
      orc_inner_product(q, scol[i], scol[j], n, g);                /* :261-264 */
      orc_polymul(q, app, g, pr);                                  /* :265 */
      orc_poly_add(q, bl, pr, bl);
    }
  for (size_t i = 0; i < r; i++) {
    orc_inner_product(q, phipp + i * n * D, scol[i], n, pr);       /* :272-275 */
    orc_poly_add(q, br, pr, br);
  }
  orc_poly_add(q, bl, br, out->bpp);                               /* :278 */
  {
    /* verification.rs:532-551: const term == <omega, projection> + psi*b' */
    uint64_t chk = 0;
    for (int m = 0; m < 256; m++) chk = zq_add(chk, zq_mul(in->omega[m], out->proj[m], q), q);
    chk = zq_add(chk, zq_mul(in->psi, in->b_prime, q), q);
    out->bpp_ok = (out->bpp[0] == chk);
  }

  /* step 7, :301-314: phi_i = alpha*phi^(F)_i + beta*phi''_i */
  uint64_t *phif = out->phi_final ? out->phi_final : (uint64_t *)malloc(r * n * D * sizeof(uint64_t));
  for (size_t i = 0; i < r; i++)
    for (size_t x = 0; x < n; x++) {
      orc_polymul(q, in->alpha, in->phi + (x * r + i) * D, acc);
      orc_polymul(q, in->beta, phipp + (i * n + x) * D, pr);
      orc_poly_add(q, acc, pr, phif + (i * n + x) * D);
    }
  free(phipp);

  /* step 8, :320-358: H[i][j] = 2^-1 (<phi_i,s_j> + <phi_j,s_i>).  The
   * reference runs this with MOD_SUSPENSION (plain integers, no fold) and then
   * recompute_mod (algebraic.rs:318-324); the net effect is ring arithmetic. */
  uint64_t half = inv2(q);
  for (size_t i = 0; i < r; i++)
    for (size_t j = 0; j < r; j++) {
      orc_inner_product(q, phif + i * n * D, scol[j], n, acc);
      orc_inner_product(q, phif + j * n * D, scol[i], n, pr);
      orc_poly_add(q, acc, pr, acc);
      orc_poly_scale(q, acc, half, out->H + (i * r + j) * D);
    }
  if (!out->phi_final) free(phif);

  /* step 9, :364-378: u2 = sum_{i<=j,k} h_ij^(k) * D_ijk */
  memset(out->u2, 0, k2 * D * sizeof(uint64_t));
  uint64_t *hdec = (uint64_t *)malloc((size_t)t1 * D * sizeof(uint64_t));
  for (size_t i = 0; i < r; i++)
    for (size_t j = i; j < r; j++) {
      int rc = orc_decompose(q, out->H + (i * r + j) * D, in->b1, t1, hdec); /* :370 */
      if (rc) return rc;
      for (int k = 0; k < t1; k++) {
        const uint64_t *Dijk = in->Dm + ((pair_index(i, j, r) * (size_t)t1 + (size_t)k) * k2) * D;
        for (size_t rho = 0; rho < k2; rho++) {
          orc_polymul(q, hdec + (size_t)k * D, Dijk + rho * D, pr);            /* :371 */
          orc_poly_add(q, out->u2 + rho * D, pr, out->u2 + rho * D);           /* :373-375 */
        }
      }
    }
  free(hdec);

  /* step 10, :380-399: z = sum_i c_i * s_i */
  memset(out->z, 0, n * D * sizeof(uint64_t));
  for (size_t i = 0; i < r; i++)
    for (size_t x = 0; x < n; x++) {
      orc_polymul(q, in->c + i * D, scol[i] + x * D, pr);
      orc_poly_add(q, out->z + x * D, pr, out->z + x * D);
    }

  for (size_t i = 0; i < r; i++) free(scol[i]);
  free(scol);
  return 0;
}

void orc_state_b(uint64_t q, size_t n, size_t r, const uint64_t *S, const uint64_t *phi,
                 const uint64_t *a, uint64_t *b_out) {
  uint64_t acc[D], g[D], pr[D]; /* structs.rs:320-341 */
  memset(acc, 0, sizeof acc);
  uint64_t **scol = (uint64_t **)malloc(r * sizeof *scol);
  for (size_t i = 0; i < r; i++) scol[i] = column(S, n, r, i);
  for (size_t i = 0; i < r; i++)
    for (size_t j = 0; j < r; j++) {
      orc_inner_product(q, scol[i], scol[j], n, g);
      orc_polymul(q, a + (i * r + j) * D, g, pr);
      orc_poly_add(q, acc, pr, acc);
    }
  for (size_t i = 0; i < r; i++) {
    uint64_t *pc = column(phi, n, r, i);
    orc_inner_product(q, pc, scol[i], n, pr);
    orc_poly_add(q, acc, pr, acc);
    free(pc);
  }
  memcpy(b_out, acc, sizeof acc);
  for (size_t i = 0; i < r; i++) free(scol[i]);
  free(scol);
}

/* ------------------------------------------------ verifier core */

static int poly_eq(const uint64_t *a, const uint64_t *b) { return memcmp(a, b, D * sizeof(uint64_t)) == 0; }

int orc_verify_core(const orc_prove_in *in, const orc_prove_out *pf, const uint64_t *b_poly,
                    int64_t bconst, double beta_prime, double *line14_sum) {
  uint64_t q = in->q;
  size_t n = in->n, r = in->r, kappa = in->kappa, k1 = in->kappa1, k2 = in->kappa2;
  int t1 = in->t1, t2 = in->t2;
  uint64_t acc[D], pr[D], pr2[D];
  int fail = 0;

  /* lines 3-4 (verification.rs:38-89): a'' = psi*a', phi''_i as in the prover */
  uint64_t *app = (uint64_t *)malloc(r * r * D * sizeof(uint64_t));
  for (size_t e = 0; e < r * r; e++) orc_poly_scale(q, in->a + e * D, in->psi, app + e * D);
  uint64_t *phipp = (uint64_t *)malloc(r * n * D * sizeof(uint64_t));
  for (size_t i = 0; i < r; i++) phi_prime_prime(in, i, phipp + i * n * D);
  /* line 5 (:96-111): a_ij = alpha*a_ij + beta*a''_ij */
  uint64_t *aa = (uint64_t *)malloc(r * r * D * sizeof(uint64_t));
  for (size_t e = 0; e < r * r; e++) {
    orc_polymul(q, in->alpha, in->a + e * D, acc);
    orc_polymul(q, in->beta, app + e * D, pr);
    orc_poly_add(q, acc, pr, aa + e * D);
  }
  /* line 6 (:118-133) */
  uint64_t *phi = (uint64_t *)malloc(r * n * D * sizeof(uint64_t));
  for (size_t i = 0; i < r; i++)
    for (size_t x = 0; x < n; x++) {
      orc_polymul(q, in->alpha, in->phi + (x * r + i) * D, acc);
      orc_polymul(q, in->beta, phipp + (i * n + x) * D, pr);
      orc_poly_add(q, acc, pr, phi + (i * n + x) * D);
    }
  /* line 7 (:140-148): b = alpha*b + beta*b'' */
  uint64_t bb[D];
  orc_polymul(q, in->alpha, b_poly, acc);
  orc_polymul(q, in->beta, pf->bpp, pr);
  orc_poly_add(q, acc, pr, bb);

  /* checks 8, 9 (:157-178) */
  for (size_t i = 0; i < r && !fail; i++)
    for (size_t j = 0; j < r; j++)
      if (!poly_eq(pf->G + (i * r + j) * D, pf->G + (j * r + i) * D)) { fail = 8; break; }
  for (size_t i = 0; i < r && !fail; i++)
    for (size_t j = 0; j < r; j++)
      if (!poly_eq(pf->H + (i * r + j) * D, pf->H + (j * r + i) * D)) { fail = 9; break; }

  /* lines 10-14 (:185-267): decompositions and the f64 norm sum, in the reference's order */
  uint64_t *tdec_all = NULL, *gdec_all = NULL, *hdec_all = NULL;
  if (!fail) {
    double sum = 0.0;
    uint64_t *zdec = (uint64_t *)malloc(2 * n * D * sizeof(uint64_t));
    if (orc_decompose_vec(q, pf->z, n, bconst, 2, zdec)) fail = 10;
    if (!fail) for (int i = 0; i < 2; i++) sum += orc_vec_poly_norm_squared(zdec + (size_t)i * n * D, n);
    free(zdec);
    tdec_all = (uint64_t *)malloc(r * (size_t)t1 * kappa * D * sizeof(uint64_t));
    for (size_t i = 0; i < r && !fail; i++) {
      if (orc_decompose_vec(q, pf->t + i * kappa * D, kappa, in->b1, t1, tdec_all + i * (size_t)t1 * kappa * D)) fail = 11;
      if (!fail) for (int k = 0; k < t1; k++) sum += orc_vec_poly_norm_squared(tdec_all + (i * (size_t)t1 + k) * kappa * D, kappa);
    }
    gdec_all = (uint64_t *)malloc(r * r * (size_t)t2 * D * sizeof(uint64_t));
    hdec_all = (uint64_t *)malloc(r * r * (size_t)t1 * D * sizeof(uint64_t));
    for (size_t e = 0; e < r * r && !fail; e++) {
      if (orc_decompose(q, pf->G + e * D, in->b2, t2, gdec_all + e * (size_t)t2 * D)) fail = 12;
      if (!fail) for (int k = 0; k < t2; k++) sum += orc_poly_norm(gdec_all + (e * (size_t)t2 + k) * D, NULL, NULL);
    }
    for (size_t e = 0; e < r * r && !fail; e++) {
      if (orc_decompose(q, pf->H + e * D, in->b1, t1, hdec_all + e * (size_t)t1 * D)) fail = 13;
      if (!fail) for (int k = 0; k < t1; k++) sum += orc_poly_norm(hdec_all + (e * (size_t)t1 + k) * D, NULL, NULL);
    }
    if (line14_sum) *line14_sum = sum;
    if (!fail && sum > beta_prime) fail = 14; /* :265 */
  }

  /* check 15 (:274-296): A z == sum c_i t_i */
  for (size_t rho = 0; rho < kappa && !fail; rho++) {
    orc_inner_product(q, in->A + rho * n * D, pf->z, n, acc);
    memset(pr2, 0, sizeof pr2);
    for (size_t i = 0; i < r; i++) {
      orc_polymul(q, in->c + i * D, pf->t + (i * kappa + rho) * D, pr);
      orc_poly_add(q, pr2, pr, pr2);
    }
    if (!poly_eq(acc, pr2)) fail = 15;
  }
  /* check 16 (:303-314): <z,z> == sum g_ij c_i c_j */
  if (!fail) {
    orc_inner_product(q, pf->z, pf->z, n, acc);
    memset(pr2, 0, sizeof pr2);
    for (size_t i = 0; i < r; i++)
      for (size_t j = 0; j < r; j++) {
        orc_polymul(q, pf->G + (i * r + j) * D, in->c + i * D, pr);
        orc_polymul(q, pr, in->c + j * D, pr);
        orc_poly_add(q, pr2, pr, pr2);
      }
    if (!poly_eq(acc, pr2)) fail = 16;
  }
  /* check 17 (:320-334): sum <phi_i,z> c_i == sum h_ij c_i c_j */
  if (!fail) {
    memset(acc, 0, sizeof acc);
    for (size_t i = 0; i < r; i++) {
      orc_inner_product(q, phi + i * n * D, pf->z, n, pr);
      orc_polymul(q, pr, in->c + i * D, pr);
      orc_poly_add(q, acc, pr, acc);
    }
    memset(pr2, 0, sizeof pr2);
    for (size_t i = 0; i < r; i++)
      for (size_t j = 0; j < r; j++) {
        orc_polymul(q, pf->H + (i * r + j) * D, in->c + i * D, pr);
        orc_polymul(q, pr, in->c + j * D, pr);
        orc_poly_add(q, pr2, pr, pr2);
      }
    if (!poly_eq(acc, pr2)) fail = 17;
  }
  /* check 18 (:340-352): sum a_ij g_ij + sum h_ii - b == 0 */
  if (!fail) {
    memset(acc, 0, sizeof acc);
    for (size_t i = 0; i < r; i++) {
      for (size_t j = 0; j < r; j++) {
        orc_polymul(q, aa + (i * r + j) * D, pf->G + (i * r + j) * D, pr);
        orc_poly_add(q, acc, pr, acc);
      }
      orc_poly_add(q, acc, pf->H + (i * r + i) * D, acc);
    }
    orc_poly_sub(q, acc, bb, acc);
    if (orc_trimmed_len(acc) != 0) fail = 18;
  }
  /* check 19 (:372-415): recompute u1 */
  if (!fail) {
    uint64_t *u = (uint64_t *)calloc(k1 * D, sizeof(uint64_t));
    for (size_t i = 0; i < r; i++)
      for (int k = 0; k < t1; k++) {
        const uint64_t *Bik = in->B + ((i * (size_t)t1 + (size_t)k) * k1) * kappa * D;
        for (size_t rho = 0; rho < k1; rho++) {
          orc_inner_product(q, Bik + rho * kappa * D, tdec_all + (i * (size_t)t1 + k) * kappa * D, kappa, pr);
          orc_poly_add(q, u + rho * D, pr, u + rho * D);
        }
      }
    for (size_t i = 0; i < r; i++)
      for (size_t j = i; j < r; j++)
        for (int k = 0; k < t2; k++) {
          const uint64_t *Cijk = in->C + ((pair_index(i, j, r) * (size_t)t2 + (size_t)k) * k2) * D;
          for (size_t rho = 0; rho < k2; rho++) {
            orc_polymul(q, gdec_all + ((i * r + j) * (size_t)t2 + k) * D, Cijk + rho * D, pr);
            orc_poly_add(q, u + rho * D, pr, u + rho * D);
          }
        }
    if (memcmp(u, pf->u1, k1 * D * sizeof(uint64_t)) != 0) fail = 19;
    free(u);
  }
  /* check 20 (:421-435): recompute u2 */
  if (!fail) {
    uint64_t *u = (uint64_t *)calloc(k2 * D, sizeof(uint64_t));
    for (size_t i = 0; i < r; i++)
      for (size_t j = i; j < r; j++)
        for (int k = 0; k < t1; k++) {
          const uint64_t *Dijk = in->Dm + ((pair_index(i, j, r) * (size_t)t1 + (size_t)k) * k2) * D;
          for (size_t rho = 0; rho < k2; rho++) {
            orc_polymul(q, hdec_all + ((i * r + j) * (size_t)t1 + k) * D, Dijk + rho * D, pr);
            orc_poly_add(q, u + rho * D, pr, u + rho * D);
          }
        }
    if (memcmp(u, pf->u2, k2 * D * sizeof(uint64_t)) != 0) fail = 20;
    free(u);
  }
  free(app); free(phipp); free(aa); free(phi);
  free(tdec_all); free(gdec_all); free(hdec_all);
  return fail;
}

/* ------------------------------------------------ CRS random oracle */

#define ROTL32(x, n) (((x) << (n)) | ((x) >> (32 - (n))))
#define QR(a, b, c, d) \
  a += b; d ^= a; d = ROTL32(d, 16); c += d; b ^= c; b = ROTL32(b, 12); \
  a += b; d ^= a; d = ROTL32(d, 8);  c += d; b ^= c; b = ROTL32(b, 7);

void orc_chacha20_block(const uint32_t st[16], uint32_t out[16]) { /* RFC 7539 section 2.3 */
  uint32_t x[16];
  memcpy(x, st, sizeof x);
  for (int i = 0; i < 10; i++) {
    QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
    QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
  }
  for (int i = 0; i < 16; i++) out[i] = x[i] + st[i];
}

static int clz128(u128 x) {
  uint64_t hi = (uint64_t)(x >> 64), lo = (uint64_t)x;
  if (hi) return __builtin_clzll(hi);
  return 64 + (lo ? __builtin_clzll(lo) : 64);
}

/* src/structs.rs:167-171.  ChaCha20Rng::from_seed: key = the 32 seed bytes as little-endian words, 64-bit block
 * counter from 0, 64-bit stream id 0.  gen_range(0..Q) on i128 -> UniformInt::sample_single_inclusive(0, Q-1):
 *   range = Q; zone = (range << range.leading_zeros()) - 1;
 *   loop { v = next_u64() | next_u64() << 64; (hi, lo) = v.wmul(range); if lo <= zone { return hi } } */
uint64_t orc_crs_coeff(const uint8_t seed[32], uint64_t q, int *draws) {
  uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
  for (int i = 0; i < 8; i++)
    st[4 + i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
  st[14] = st[15] = 0;
  const u128 range = q;
  const u128 zone = (range << clz128(range)) - 1;
  uint32_t blk[16];
  int nd = 0;
  for (uint64_t counter = 0;; counter++) {
    st[12] = (uint32_t)counter;
    st[13] = (uint32_t)(counter >> 32);
    orc_chacha20_block(st, blk);
    for (int d = 0; d < 4; d++) { /* 4 u128 candidates per 16-word block */
      uint64_t x = (uint64_t)blk[4 * d] | ((uint64_t)blk[4 * d + 1] << 32);
      uint64_t y = (uint64_t)blk[4 * d + 2] | ((uint64_t)blk[4 * d + 3] << 32);
      nd++;
      /* 256-bit product (y*2^64 + x) * range, range < 2^64 */
      u128 a = (u128)x * (uint64_t)range, b = (u128)y * (uint64_t)range;
      u128 t = (a >> 64) + (uint64_t)b;
      u128 lo = (t << 64) | (uint64_t)a;
      u128 hi = (b >> 64) + (t >> 64);
      if (lo <= zone) {
        if (draws) *draws = nd;
        return (uint64_t)hi;
      }
    }
  }
}

/* 256-bit big-endian seed arithmetic: BigUint::from_bytes_be(base) + increment (src/structs.rs:59-68), then
 * increment_seed per coefficient (:155-165; wraps silently at 2^256 like the byte loop does) */
static void seed_add(uint8_t s[32], u128 inc) {
  unsigned carry = 0;
  for (int i = 31; i >= 0; i--) {
    unsigned add = (i >= 16) ? (unsigned)((inc >> (8 * (31 - i))) & 0xff) : 0;
    unsigned v = s[i] + add + carry;
    s[i] = (uint8_t)v;
    carry = v >> 8;
  }
}

void orc_crs_fill(const uint8_t base_seed[32], uint64_t off_lo, uint64_t off_hi, size_t count, uint64_t q, uint64_t *out) {
  uint8_t s[32];
  memcpy(s, base_seed, 32);
  seed_add(s, ((u128)off_hi << 64) | off_lo);
  for (size_t k = 0; k < count; k++) {
    out[k] = orc_crs_coeff(s, q, NULL);
    seed_add(s, 1);
  }
}

void orc_crs_offset(int which, size_t n, size_t r, size_t kappa, size_t kappa1, size_t kappa2, int t1,
                    size_t i, size_t j, size_t k, size_t row, uint64_t *lo, uint64_t *hi) {
  u128 N = n, R = r, K = kappa, K1 = kappa1, K2 = kappa2, T1 = (u128)t1, Dd = D;
  u128 offset_A = K * N * Dd;                                   /* :78, :94, :120 */
  u128 size_B = K1 * K;                                         /* :81 — in polynomials, not coefficients (sic) */
  u128 sum_pairs = i > 0 ? (u128)i * R - ((u128)i * (i - 1)) / 2 : 0; /* :101-105 */
  u128 inc = 0;
  if (which == 0) inc = (u128)row * (N * Dd);                                                       /* :62 */
  else if (which == 1) inc = offset_A + ((u128)i * T1 + k) * size_B + (u128)row * (K * Dd);          /* :82 */
  else if (which == 2) inc = offset_A + (R * T1) * size_B * Dd + ((u128)k + T1 * (sum_pairs + (j - i))) * (K2 * Dd); /* :106-108 (T_1, sic) */
  else inc = offset_A + (R * T1) * size_B * Dd + (R * (R + 1) / 2) * (K2 * Dd) + ((u128)k + T1 * (sum_pairs + (j - i))) * (K2 * Dd); /* :127-138 */
  *lo = (uint64_t)inc;
  *hi = (uint64_t)(inc >> 64);
}

/* ------------------------------------------------ timed baseline */

uint64_t orc_bench_inner_product_faithful(uint64_t q, const uint64_t *v1, const uint64_t *v2, size_t len, int reps) {
  uint64_t acc[D], pr[D], chk = 0;
  for (int rep = 0; rep < reps; rep++) {
    memset(acc, 0, sizeof acc);
    for (size_t i = 0; i < len; i++) {
      orc_polymul_faithful(q, v1 + i * D, v2 + i * D, pr);
      for (int k = 0; k < D; k++) acc[k] = zq_add(acc[k], pr[k], q);
    }
    for (int k = 0; k < D; k++) chk ^= acc[k] + (uint64_t)rep;
  }
  return chk;
}
