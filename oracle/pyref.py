"""pyref.py — pure-Python big-int twin of the LaBRADOR-SNARK hot-path oracle.

TEST INFRASTRUCTURE ONLY (same rules as labrador_oracle.h): imported by tests/
and by tests/golden/make_golden.py, never by the product.

It is written independently of labrador_oracle.c and mirrors the *shape* of the
Rust reference instead of the flat C arrays: an Rq is a Python list of residues
with trailing zeros stripped (`Polynomial::new`, src/algebraic.rs:303-315), every
Zq operation reduces immediately (src/algebraic.rs:30-38), `decompose` is the
literal while-loop (src/util.rs:389-442).  tests/test_oracle.py requires the two
restatements to agree bit-for-bit.  PARITY PINNING: see labrador_oracle.h —
value-level parity with the Rust binary is unpinned (no KATs exist upstream).

Citations are file:line in the reference tree.
"""
import math

D = 64          # src/constants.rs:15
TAU = 71.0      # src/constants.rs:22
T_OP = 15.0     # src/constants.rs:23


def trunc_rem(a, b):
    """Rust `%` on i128: remainder has the sign of the dividend."""
    r = abs(a) % abs(b)
    return -r if a < 0 else r


def trunc_div(a, b):
    """Rust `/` on i128: truncates toward zero."""
    s = -1 if (a < 0) != (b < 0) else 1
    return s * (abs(a) // abs(b))


def mod_positive(v, q):  # src/util.rs:16-23
    r = trunc_rem(v, q)
    return r + q if r < 0 else r


class Ring:
    """R_q = Z_q[X]/(X^64+1) with the reference's semantics, q a parameter."""

    def __init__(self, q):
        self.q = q

    # ---- Zq (src/algebraic.rs:24-297)
    def zq(self, v):
        return mod_positive(v, self.q)

    def neg(self, a):  # :56-63
        return 0 if a == 0 else self.q - a

    # ---- Rq
    @staticmethod
    def trim(c):  # Polynomial::new strips trailing zeros
        c = list(c)
        while c and c[-1] == 0:
            c.pop()
        return c

    def new(self, coeffs):  # Rq::new -> reduction, src/algebraic.rs:308-315, 352-376
        coeffs = [self.zq(c) for c in coeffs]
        red = self.trim(coeffs[:D])
        for deg in range(D, len(coeffs)):
            factor = (-1) ** (deg // D)
            term = [0] * (deg % D + 1)
            term[deg % D] = self.zq(coeffs[deg] * factor)
            red = self.add(red, self.trim(term))
        return red

    def add(self, a, b):  # :441-505 (Polynomial +, zero-extended, re-trimmed)
        n = max(len(a), len(b))
        return self.trim([self.zq((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) for i in range(n)])

    def sub(self, a, b):  # :461-515
        n = max(len(a), len(b))
        return self.trim([self.zq((a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0)) for i in range(n)])

    def mul(self, a, b):  # classic branch, :402-403
        if not a or not b:
            return []
        prod = [0] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                prod[i + j] = self.zq(prod[i + j] + self.zq(x * y))
        return self.new(prod)

    def mul_ntt_branch(self, a, b):  # :384-399 and :417-420
        pa = list(a) + [0] * (D - len(a))
        pb = list(b) + [0] * (D - len(b))
        out = [0] * D
        for i in range(D):
            for j in range(D):
                k = i + j
                if k < D:
                    out[k] += pa[i] * pb[j]
                else:
                    out[k - D] -= pa[i] * pb[j]
        res = []
        for v in out:
            u = v % (1 << 64)                      # PLAN.negacyclic_polymul: exact product mod 2^64
            s = u - (1 << 64) if u >= (1 << 63) else u
            res.append(self.zq(s))
        return self.trim(res)

    def scale(self, p, s):  # src/util.rs:176-180
        return self.new([self.zq(x * s) for x in p])

    def multiply_poly_ints(self, p, ints):  # src/util.rs:148-155
        res = []
        for c in ints:
            res = self.add(res, self.scale(p, c))
        return res

    def sigma_inv(self, a):  # src/util.rs:118-137
        out = [0] * D
        for deg, c in enumerate(a):
            if deg == 0:
                out[0] = c
            else:
                out[D - deg] = self.neg(c)
        return self.new(out)

    def inner(self, v1, v2):  # src/util.rs:496-509
        assert len(v1) == len(v2), "inner product not defined on vectors of unequal length"
        res = []
        for x, y in zip(v1, v2):
            res = self.add(res, self.mul(x, y))
        return res

    def poly_by_poly_vec(self, p, v):  # src/util.rs:289-296
        return [self.mul(p, x) for x in v]

    def add_poly_vec(self, a, b):  # src/util.rs:298-309
        assert len(a) == len(b)
        return [self.add(x, y) for x, y in zip(a, b)]

    # ---- decomposition (src/util.rs:377-442), literal
    def centered_rep(self, val, b):
        if val > mod_positive(b // 2, self.q):
            return self.zq(trunc_rem(b - val, b))
        return val

    def decompose(self, p, base, exp):
        if base < 2:
            raise ValueError("base < 2 never terminates in the reference")
        digits = []
        for coeff in p:
            ds = []
            v = coeff
            while v != 0:
                rem = self.centered_rep(self.zq(trunc_rem(v, base)), base)
                ds.append(rem)
                v = self.zq(trunc_div(self.zq(v - rem), base))
            digits.append(ds)
        out = []
        for k in range(exp):
            poly = []
            for i, ds in enumerate(digits):
                if len(ds) > k:
                    term = [0] * (i + 1)
                    term[i] = ds[k]
                    poly = self.add(poly, self.new(term))
            out.append(poly)
        return out

    def decompose_vec(self, vec, base, exp):  # src/util.rs:360-373
        res = [[[] for _ in vec] for _ in range(exp)]
        for i, t in enumerate(vec):
            dec = self.decompose(t, base, exp)
            for j in range(exp):
                res[j][i] = dec[j]
        return res

    # ---- norms (src/util.rs:188-219)
    @staticmethod
    def poly_norm_int(p):
        return sum(x * x for x in p)

    @classmethod
    def poly_norm(cls, p):
        return float(cls.poly_norm_int(p))  # `as f64`: round-to-nearest-even, same as Python int -> float

    @classmethod
    def vec_poly_norm_squared(cls, v):
        s = 0.0
        for p in v:
            s += cls.poly_norm(p)
        return s

    # ---- coefficient (un)flattening (src/util.rs:446-482)
    @staticmethod
    def concat(vec):
        out = []
        for p in vec:
            out.extend(list(p) + [0] * (D - len(p)))
        return out

    def unconcat(self, flat):
        assert len(flat) % D == 0
        return [self.new(flat[i:i + D]) for i in range(0, len(flat), D)]

    @staticmethod
    def pad(p):
        return list(p) + [0] * (D - len(p))


def l2_norm(v):  # src/util.rs:216-219
    return math.sqrt(float(sum(x * x for x in v)))


def rust_round(x):  # f64::round: half away from zero
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def runtime_constants(q, n, r):  # src/constants.rs:232-264
    Q = float(q)
    c = {"n": n, "r": r, "kappa": n * D, "kappa1": n * D, "kappa2": n * D}
    c["beta_bound"] = int(math.floor(math.sqrt(30.0 / 128.0) * Q / 125.0))
    c["std"] = float(c["beta_bound"]) / math.sqrt(float(r * n * D))
    c["b"] = int(rust_round(math.sqrt(math.sqrt(12.0 * float(r) * TAU) * c["std"])))
    c["t1"] = int(rust_round(math.log2(Q) / math.log2(float(c["b"])))) if c["b"] > 1 else None
    if c["t1"] is None or c["t1"] == 0:
        return c
    c["b1"] = int(math.pow(Q, 1.0 / float(c["t1"])))
    c["t2"] = int(rust_round(math.log2(math.sqrt(24.0 * float(n * D)) * c["std"] ** 2) / math.log2(float(c["b"]))))
    c["b2"] = int(rust_round(math.pow(math.sqrt(float(24 * n * D)) * c["std"] ** 2, 1.0 / float(c["t2"])))) if c["t2"] != 0 else None
    if c["b2"] is None:
        return c
    b1, b2, t1, t2, rr = float(c["b1"]), float(c["b2"]), float(c["t1"]), float(c["t2"]), float(r)
    c["gamma"] = float(c["beta_bound"] ** 2) * TAU
    c["gamma1"] = ((b1 * b1 * t1) / 12.0) * rr * float(c["kappa"]) * float(D) + ((b2 * b2 * t2) / 12.0) * ((rr * rr + rr) / 2.0) * float(D)
    c["gamma2"] = ((b1 * b1 * t1) / 12.0) * ((rr * rr + rr) / 2.0) * float(D)
    c["beta_prime"] = (2.0 / float(c["b"]) ** 2) * c["gamma"] + c["gamma1"] + c["gamma2"]
    return c


def pair_index(i, j, r):  # src/structs.rs:101-106
    return i * r - (i * (i - 1)) // 2 + (j - i)


def prove_core(R, p):
    """Prover::proof_gen (src/proofgen.rs:30-427) as a deterministic function.

    `p` is a dict: n, r, kappa, kappa1, kappa2, b1, t1, b2, t2, beta_bound,
    S[n][r] / phi[n][r] / a[r][r] (Rq lists), b_prime (int),
    A[kappa][n], B[r*t1][kappa1][kappa], C[pairs*t2][kappa2], Dm[pairs*t1][kappa2],
    Pi[r][256][n*64] (ints in {-1,0,1}), psi (int), omega[256], alpha, beta, c[r].
    """
    n, r, kappa, k1, k2 = p["n"], p["r"], p["kappa"], p["kappa1"], p["kappa2"]
    t1, t2 = p["t1"], p["t2"]
    assert k1 == k2
    scol = [[p["S"][x][i] for x in range(n)] for i in range(r)]
    out = {}
    # 1 (:41-49)
    t = [[R.inner(p["A"][rho], scol[i]) for rho in range(kappa)] for i in range(r)]
    # 2 (:59-70)
    G = [[R.inner(scol[i], scol[j]) for j in range(r)] for i in range(r)]
    # 3 (:101-132)
    lhs = [[] for _ in range(k1)]
    for i in range(r):
        tdec = R.decompose_vec(t[i], p["b1"], t1)
        for k in range(t1):
            Bik = p["B"][i * t1 + k]
            prod = [R.inner(Bik[rho], tdec[k]) for rho in range(k1)]
            lhs = R.add_poly_vec(prod, lhs)
    # 4 (:137-148)
    rhs = [[] for _ in range(k2)]
    for i in range(r):
        for j in range(i, r):
            gdec = R.decompose(G[i][j], p["b2"], t2)
            for k in range(t2):
                prod = R.poly_by_poly_vec(gdec[k], p["C"][pair_index(i, j, r) * t2 + k])
                rhs = R.add_poly_vec(prod, rhs)
    u1 = R.add_poly_vec(lhs, rhs)  # :153
    # 5 (:161-186, 429-456)
    proj_int = [0] * 256
    for i in range(r):
        s = R.concat(scol[i])
        for m in range(256):
            row = p["Pi"][i][m]
            proj_int[m] += sum(row[c] * s[c] for c in range(n * D))
    jl_valid = l2_norm(proj_int) <= math.sqrt(128.0) * float(p["beta_bound"])
    proj = [R.zq(v) for v in proj_int]
    # 6 (:205-289)
    psi, omega = [p["psi"]], p["omega"]
    app = [[R.multiply_poly_ints(p["a"][i][j], psi) for j in range(r)] for i in range(r)]
    phipp = []
    for i in range(r):
        lhs_v = [R.multiply_poly_ints(p["phi"][x][i], psi) for x in range(n)]
        rhs_v = [[] for _ in range(n)]
        for m in range(256):
            polys = R.unconcat([R.zq(v) for v in p["Pi"][i][m]])
            conj = [R.sigma_inv(q_) for q_ in polys]
            rhs_v = R.add_poly_vec(rhs_v, [R.scale(c_, omega[m]) for c_ in conj])
        phipp.append(R.add_poly_vec(lhs_v, rhs_v))
    bl = []
    for i in range(r):
        for j in range(r):
            bl = R.add(bl, R.mul(app[i][j], R.inner(scol[i], scol[j])))
    br = []
    for i in range(r):
        br = R.add(br, R.inner(phipp[i], scol[i]))
    bpp = R.add(bl, br)
    chk = 0
    for m in range(256):
        chk = R.zq(chk + R.zq(omega[m] * proj[m]))
    chk = R.zq(chk + R.zq(p["psi"] * p["b_prime"]))
    bpp_ok = (bpp[0] if bpp else 0) == chk
    # 7 (:301-314)
    phif = [R.add_poly_vec(R.poly_by_poly_vec(p["alpha"], [p["phi"][x][i] for x in range(n)]),
                           R.poly_by_poly_vec(p["beta"], phipp[i])) for i in range(r)]
    # 8 (:320-358)
    half = pow(2, R.q - 2, R.q)
    H = [[R.scale(R.add(R.inner(phif[i], scol[j]), R.inner(phif[j], scol[i])), half) for j in range(r)] for i in range(r)]
    # 9 (:364-378)
    u2 = [[] for _ in range(k2)]
    for i in range(r):
        for j in range(i, r):
            hdec = R.decompose(H[i][j], p["b1"], t1)
            for k in range(t1):
                prod = R.poly_by_poly_vec(hdec[k], p["Dm"][pair_index(i, j, r) * t1 + k])
                u2 = R.add_poly_vec(u2, prod)
    # 10 (:380-399)
    z = [[] for _ in range(n)]
    for i in range(r):
        prod = R.poly_by_poly_vec(p["c"][i], scol[i])
        z = R.add_poly_vec(z, prod)
    out.update(t=t, G=G, H=H, u1=u1, u2=u2, proj_int=proj_int, proj=proj, bpp=bpp, z=z,
               jl_valid=jl_valid, bpp_ok=bpp_ok, phi_final=phif)
    return out


# ---------------------------------------------------------------- CRS random oracle (src/structs.rs:27-190)
# ChaCha20 block function: RFC 7539 section 2.3.  Word order / u128 assembly / gen_range: rand_chacha 0.3.1 and
# rand 0.8.5 restated from their published algorithms (crates not in the reference tree -> parity unpinned).

def _rotl(x, n):
    return ((x << n) & 0xFFFFFFFF) | (x >> (32 - n))


def _qr(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = _rotl(s[b] ^ s[c], 7)


def chacha20_block(state):
    s = list(state)
    for _ in range(10):
        _qr(s, 0, 4, 8, 12); _qr(s, 1, 5, 9, 13); _qr(s, 2, 6, 10, 14); _qr(s, 3, 7, 11, 15)
        _qr(s, 0, 5, 10, 15); _qr(s, 1, 6, 11, 12); _qr(s, 2, 7, 8, 13); _qr(s, 3, 4, 9, 14)
    return [(a + b) & 0xFFFFFFFF for a, b in zip(s, state)]


def crs_coeff(seed_int, q):
    """generate_random_coeff (src/structs.rs:167-171) for the 256-bit big-endian seed value `seed_int`."""
    seed = (seed_int % (1 << 256)).to_bytes(32, "big")
    key = [int.from_bytes(seed[4 * i:4 * i + 4], "little") for i in range(8)]
    rng_range = q
    zone = ((rng_range << (128 - rng_range.bit_length())) - 1) & ((1 << 128) - 1)
    counter = 0
    while True:
        blk = chacha20_block([0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + key +
                             [counter & 0xFFFFFFFF, counter >> 32, 0, 0])
        for d in range(4):
            x = blk[4 * d] | (blk[4 * d + 1] << 32)
            y = blk[4 * d + 2] | (blk[4 * d + 3] << 32)
            v = x | (y << 64)
            prod = v * rng_range
            hi, lo = prod >> 128, prod & ((1 << 128) - 1)
            if lo <= zone:
                return hi
        counter += 1


def crs_fill(base_seed_bytes, offset, count, q):
    base = int.from_bytes(base_seed_bytes, "big")
    return [crs_coeff(base + offset + k, q) for k in range(count)]


def crs_offsets(n, r, kappa, kappa1, kappa2, t1):
    """The reference's seed offsets in coefficient units, copied literally (src/structs.rs:62, 78-82, 94-108, 120-138)."""
    offset_A = kappa * n * D
    size_B = kappa1 * kappa

    def pairs(i):
        return i * r - (i * (i - 1)) // 2 if i > 0 else 0

    return {
        "A": lambda row: row * (n * D),
        "B": lambda i, k, row: offset_A + (i * t1 + k) * size_B + row * (kappa * D),
        "C": lambda i, j, k: offset_A + (r * t1) * size_B * D + (k + t1 * (pairs(i) + (j - i))) * (kappa2 * D),
        "D": lambda i, j, k: offset_A + (r * t1) * size_B * D + (r * (r + 1) // 2) * (kappa2 * D) +
        (k + t1 * (pairs(i) + (j - i))) * (kappa2 * D),
    }
