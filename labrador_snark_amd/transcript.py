"""Proof bytes: `bincode::serialize(&Transcript)` (src/structs.rs:192-220) from the prover's flat arrays.
SURVEY §8(f) row f2.  Host-side formatting only (numpy); nothing here is on the timed path.

Layout (bincode 1.3.3 defaults: little-endian, fixed-width integers, u64 length prefixes; serde derive order):
    u_1: Vec<Rq> | pi_i_all: Vec<Array2<Zq>> | projection: Vec<Zq> | psi: Vec<Vec<Zq>> | omega: Vec<Vec<Zq>> |
    b_prime_prime: Vec<Rq> | alpha: Vec<Rq> | beta: Vec<Rq> | u_2: Vec<Rq> | c: Vec<Rq> | z: Vec<Rq> |
    t_i_all: Vec<Vec<Rq>> | g_mat: Array2<Rq> | h_mat: Array2<Rq>                       (src/structs.rs:195-208)
  Zq        = struct { value: i128 }  -> 16 bytes LE                                     (src/algebraic.rs:24-27)
  Rq        = serialises as data_vec(): Vec<Zq> of the TRIMMED coefficients              (src/algebraic.rs:422-429)
  Array2<T> = ndarray 0.15 serde: u8 version = 1, dim (u64, u64), then u64 len + row-major elements
bincode, serde and ndarray are not in the reference tree: this layout is restated from their published formats
and is PARITY UNPINNED (no reference test covers Transcript bytes).  `size_in_bytes` in the reference is the
gzip(best) length produced by flate2's miniz_oxide backend; `gzip_size_estimate` uses zlib level 9, which is not
byte-identical to miniz_oxide, so it is an estimate, not a parity claim.
"""
import struct
import zlib

import numpy as np

D = 64


def _u64(n):
    return struct.pack("<Q", n)


def _zq_block(vals):
    """Zq values (uint64 array) -> consecutive 16-byte little-endian i128s."""
    v = np.ascontiguousarray(vals, dtype="<u8").ravel()
    out = np.zeros((v.size, 2), dtype="<u8")
    out[:, 0] = v
    return out.tobytes()


def _vec_zq(vals):
    v = np.asarray(vals, dtype=np.uint64).ravel()
    return _u64(v.size) + _zq_block(v)


def _rq(poly):
    """One Rq: Vec<Zq> of the coefficients with trailing zeros stripped (Polynomial::new)."""
    p = np.asarray(poly, dtype=np.uint64)
    nz = np.nonzero(p)[0]
    ln = int(nz[-1]) + 1 if nz.size else 0
    return _u64(ln) + _zq_block(p[:ln])


def _vec_rq(polys):
    polys = np.asarray(polys, dtype=np.uint64).reshape(-1, D)
    return _u64(polys.shape[0]) + b"".join(_rq(p) for p in polys)


def _array2_header(rows, cols):
    return b"\x01" + _u64(rows) + _u64(cols)


def serialize_transcript(T, q, tape=None):
    """Bytes of bincode::serialize(&Transcript) for transcript `T` (fields as produced by prove_core)."""
    tape = tape if tape is not None else T.tape
    host = lambda a: a.cpu().numpy().view(np.uint64) if hasattr(a, "cpu") else np.asarray(a)
    parts = []
    parts.append(_vec_rq(host(T.u_1)))
    if tape.get("Pi") is not None:
        Pi = tape["Pi"]
        Pi = Pi.cpu().numpy() if hasattr(Pi, "cpu") else np.asarray(Pi)
    else:                                              # the tiled 2-bit form (lab_prove_input.Pi2): expanded here only
        from .prover import unpack_pi_host
        P2 = tape["Pi2"]
        P2 = P2.cpu().numpy() if hasattr(P2, "cpu") else np.asarray(P2)
        Pi = unpack_pi_host(P2, P2.shape[-1] * 4 // 256)
    parts.append(_u64(Pi.shape[0]))
    for i in range(Pi.shape[0]):                       # Zq::lift: -1 -> q-1 (src/proofgen.rs:445-451)
        lifted = np.where(Pi[i] < 0, np.uint64(q - 1), Pi[i].astype(np.uint64)).astype(np.uint64)
        parts.append(_array2_header(Pi.shape[1], Pi.shape[2]) + _u64(lifted.size) + _zq_block(lifted))
    parts.append(_vec_zq(host(T.projection)))
    parts.append(_u64(1) + _vec_zq([int(tape["psi"]) % q]))          # psi: Vec<Vec<Zq>>, upper_bound = 1, L = 1
    parts.append(_u64(1) + _vec_zq(host(tape["omega"])))
    parts.append(_vec_rq(host(T.b_prime_prime)))
    parts.append(_vec_rq(host(tape["alpha"])))
    parts.append(_vec_rq(host(tape["beta"])))
    parts.append(_vec_rq(host(T.u_2)))
    parts.append(_vec_rq(host(tape["c"])))
    parts.append(_vec_rq(host(T.z)))
    t = host(T.t_i_all)
    parts.append(_u64(t.shape[0]) + b"".join(_vec_rq(t[i]) for i in range(t.shape[0])))
    for M in (host(T.g_mat), host(T.h_mat)):
        r0, r1 = M.shape[0], M.shape[1]
        parts.append(_array2_header(r0, r1) + _vec_rq(M.reshape(r0 * r1, D)))
    return b"".join(parts)


def gzip_size_estimate(data):
    """Length of a gzip stream of `data` at zlib level 9 (the reference uses flate2/miniz_oxide `best`:
    same format and level, different encoder, so sizes differ slightly)."""
    co = zlib.compressobj(9, zlib.DEFLATED, 16 + zlib.MAX_WBITS)
    return len(co.compress(data) + co.flush())
