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
# the largest prime below 2^32 with q == 1 (mod 128): SURVEY §7's "NTT-friendly 32-bit q" (path (b)); on the GPU it runs
# through the exact 3-limb RNS path (a single-limb strict-32 policy is not built), in oracle/fastcpu.c as one limb
Q_NTT32 = 4294966657
ALL_Q = [Q_DEFAULT, Q_REF33, Q_DIRECT, Q_NTT32]


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
        key = (q, max_len)
        if key not in cache:
            cache[key] = lab.Context(q, 0, max_len)
        return cache[key]

    yield get
    for c in cache.values():
        c.close()
