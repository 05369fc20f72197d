import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# moduli exercised everywhere: the reference default (src/constants.rs:195), its commented-out
# "32-bit" alternative (:196, really 2^32+15), and an NTT-friendly prime below 2^30 (single-limb path)
Q_DEFAULT = 8191
Q_REF33 = (1 << 32) + 15
Q_DIRECT = 1073738753
# the largest prime below 2^32 with q == 1 (mod 128): SURVEY §7's "NTT-friendly 32-bit q" (path (b)): one limb, 4 bytes per
# coefficient, on the strict-32 kernels (csrc/zq_device.h Zp<true>, csrc/wide_kernels.hip)
Q_NTT32 = 4294966657
# the smallest NTT-friendly prime above 2^30: single-limb strict-32 too, but it would FIT four signed byte planes (below 2139127680),
# so nothing but the context's `wide` flag keeps it off the lazy matrix-core epilogue
Q_WIDE31 = 1073741953


class RnsOnly(int):
    """A modulus whose context is created with LAB_CTX_RNS_ONLY: the generic exact RNS path although q is NTT-friendly."""
    rns_only = True


Q_NTT32_RNS = RnsOnly(Q_NTT32)
# Q_NTT32 runs single-limb on the strict-32 kernels (SURVEY section 7 path (b)); Q_NTT32_RNS is the same modulus on three RNS
# limbs (path (a)): every test parametrised over ALL_Q compares both with the oracle
ALL_Q = [Q_DEFAULT, Q_REF33, Q_DIRECT, Q_NTT32, Q_NTT32_RNS, Q_WIDE31]


def ctx_kw(q):
    """keyword arguments for lab.Context(q, ...) that honour the RnsOnly marker"""
    return {"rns_only": True} if getattr(q, "rns_only", False) else {}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return _has_gpu()


@pytest.fixture(scope="session")
def contexts():
    """One library context per modulus, shared by the GPU tests of a session."""
    import labrador_snark_amd as lab
    cache = {}

    def get(q, max_len=0):
        rns = bool(getattr(q, "rns_only", False))
        key = (int(q), max_len, rns)
        if key not in cache:
            cache[key] = lab.Context(int(q), 0, max_len, rns_only=rns)
        return cache[key]

    yield get
    for c in cache.values():
        c.close()
