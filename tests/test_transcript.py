"""Transcript byte layout (src/structs.rs:192-220): decode the serialised stream with an
independent mini-reader and compare with the arrays it was built from.  Format restated from the published
bincode/ndarray formats (crates absent from the reference tree): parity unpinned."""
import struct

import numpy as np

from labrador_snark_amd import transcript as tr

D = 64


class Reader:
    def __init__(self, b):
        self.b, self.o = b, 0

    def u8(self):
        v = self.b[self.o]; self.o += 1; return v

    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.o)[0]; self.o += 8; return v

    def zq(self):
        lo, hi = struct.unpack_from("<QQ", self.b, self.o); self.o += 16
        assert hi == 0
        return lo

    def vec_zq(self):
        return [self.zq() for _ in range(self.u64())]

    def rq(self):
        c = self.vec_zq()
        assert not c or c[-1] != 0            # trimmed
        return c + [0] * (D - len(c))

    def vec_rq(self):
        return [self.rq() for _ in range(self.u64())]


class T:
    pass


def test_bincode_layout_round_trip():
    q, n, r, kappa, kappa1 = 8191, 2, 2, 3, 2
    rng = np.random.default_rng(0)
    t = T()
    t.u_1 = rng.integers(0, q, (kappa1, D), dtype=np.uint64); t.u_1[1] = 0; t.u_1[0, 40:] = 0   # zero poly, short poly
    t.projection = rng.integers(0, q, 256, dtype=np.uint64)
    t.b_prime_prime = rng.integers(0, q, D, dtype=np.uint64)
    t.u_2 = rng.integers(0, q, (kappa1, D), dtype=np.uint64)
    t.z = rng.integers(0, q, (n, D), dtype=np.uint64)
    t.t_i_all = rng.integers(0, q, (r, kappa, D), dtype=np.uint64)
    t.g_mat = rng.integers(0, q, (r, r, D), dtype=np.uint64)
    t.h_mat = rng.integers(0, q, (r, r, D), dtype=np.uint64)
    tape = dict(Pi=rng.integers(-1, 2, (r, 256, n * D)).astype(np.int8), psi=77, omega=rng.integers(0, q, 256, dtype=np.uint64),
                alpha=rng.integers(0, q, D, dtype=np.uint64), beta=rng.integers(0, q, D, dtype=np.uint64),
                c=rng.integers(0, 3, (r, D)).astype(np.uint64))
    data = tr.serialize_transcript(t, q, tape)
    rd = Reader(data)
    assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64), t.u_1)
    assert rd.u64() == r
    for i in range(r):
        assert rd.u8() == 1 and (rd.u64(), rd.u64()) == (256, n * D)
        flat = np.array(rd.vec_zq(), dtype=np.uint64).reshape(256, n * D)
        assert np.array_equal(flat, np.where(tape["Pi"][i] < 0, q - 1, tape["Pi"][i].astype(np.int64)).astype(np.uint64))
    assert rd.vec_zq() == [int(x) for x in t.projection]
    assert rd.u64() == 1 and rd.vec_zq() == [77]
    assert rd.u64() == 1 and rd.vec_zq() == [int(x) for x in tape["omega"]]
    assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64)[0], t.b_prime_prime)
    assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64)[0], tape["alpha"])
    assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64)[0], tape["beta"])
    assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64), t.u_2)
    assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64), tape["c"])
    assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64), t.z)
    assert rd.u64() == r
    for i in range(r):
        assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64), t.t_i_all[i])
    for M in (t.g_mat, t.h_mat):
        assert rd.u8() == 1 and (rd.u64(), rd.u64()) == (r, r)
        assert np.array_equal(np.array(rd.vec_rq(), dtype=np.uint64).reshape(r, r, D), M)
    assert rd.o == len(data)
    # the zero polynomial is an empty Vec (8-byte length only), a 40-coefficient one carries 40 Zq
    assert data[8:16] == struct.pack("<Q", 40)
    assert tr.gzip_size_estimate(data) < len(data)


def test_bincode_layout_hand_built_bytes():
    """A transcript small enough to write its bytes out by hand from the published formats (bincode 1.x: little-endian
    fixed-width integers, u64 length prefixes, struct fields in declaration order; ndarray 0.15 serde: u8 version 1, the
    dimension tuple, then the elements as a sequence; Zq = one i128; Rq = Vec<Zq> of the trimmed coefficients,
    src/algebraic.rs:422-429).  This pins the serialiser to those formats — the reference holds no transcript bytes to
    pin it to the Rust binary (parity unpinned, as the module says)."""
    q = 8191
    zq = lambda v: struct.pack("<QQ", v, 0)
    u64 = lambda v: struct.pack("<Q", v)
    rq = lambda cs: u64(len(cs)) + b"".join(zq(c) for c in cs)
    t = T()
    poly = lambda cs: np.array(list(cs) + [0] * (D - len(cs)), dtype=np.uint64)
    t.u_1 = np.stack([poly([5, 0, 7]), poly([])])                 # kappa1 = 2
    t.projection = np.arange(256, dtype=np.uint64)
    t.b_prime_prime = poly([9])
    t.u_2 = np.stack([poly([1]), poly([0, 2])])
    t.z = np.stack([poly([3, 4])])                                # n = 1
    t.t_i_all = np.stack([np.stack([poly([6])])])                 # r = 1, kappa = 1
    t.g_mat = np.stack([np.stack([poly([8, 8])])])
    t.h_mat = np.stack([np.stack([poly([])])])
    Pi = np.zeros((1, 256, D), dtype=np.int8)
    Pi[0, 0, 0], Pi[0, 255, 63] = -1, 1
    omega = np.full(256, 2, dtype=np.uint64)
    tape = dict(Pi=Pi, psi=77, omega=omega, alpha=poly([1, 1]), beta=poly([0, 0, 3]), c=np.stack([poly([q - 1])]))
    lifted = [0] * (256 * D)
    lifted[0], lifted[-1] = q - 1, 1                              # Zq::lift(-1) = q - 1 (src/proofgen.rs:445-451)
    want = b"".join([
        u64(2) + rq([5, 0, 7]) + rq([]),                                              # u_1: Vec<Rq>
        u64(1) + b"\x01" + u64(256) + u64(D) + u64(256 * D) + b"".join(zq(v) for v in lifted),   # pi_i_all: Vec<Array2<Zq>>
        u64(256) + b"".join(zq(v) for v in range(256)),                                # projection: Vec<Zq>
        u64(1) + u64(1) + zq(77),                                                      # psi: Vec<Vec<Zq>>
        u64(1) + u64(256) + zq(2) * 256,                                               # omega
        u64(1) + rq([9]),                                                              # b_prime_prime: Vec<Rq>
        u64(1) + rq([1, 1]), u64(1) + rq([0, 0, 3]),                                   # alpha, beta
        u64(2) + rq([1]) + rq([0, 2]),                                                 # u_2
        u64(1) + rq([q - 1]),                                                          # c
        u64(1) + rq([3, 4]),                                                           # z
        u64(1) + u64(1) + rq([6]),                                                     # t_i_all: Vec<Vec<Rq>>
        b"\x01" + u64(1) + u64(1) + u64(1) + rq([8, 8]),                               # g_mat: Array2<Rq>
        b"\x01" + u64(1) + u64(1) + u64(1) + rq([]),                                   # h_mat
    ])
    assert tr.serialize_transcript(t, q, tape) == want
    # the 2-bit tape (lab_prove_input.Pi2) serialises to the same bytes
    from labrador_snark_amd import pack_pi_host, unpack_pi_host
    tape2 = {k: v for k, v in tape.items() if k != "Pi"}
    tape2["Pi2"] = pack_pi_host(Pi)
    assert np.array_equal(unpack_pi_host(tape2["Pi2"], D), Pi)
    assert tr.serialize_transcript(t, q, tape2) == want
