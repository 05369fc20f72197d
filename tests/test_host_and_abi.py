"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header declares
(no compute calls without a GPU), the Python binding covers them, and the product path fails loudly without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "labrador_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lab_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("lab_ctx_create", "lab_negacyclic_polymul_u64", "lab_ntt_forward", "lab_ntt_inverse",
                 "lab_inner_product", "lab_matvec_sum", "lab_ajtai_commit", "lab_gram", "lab_axpy_commit",
                 "lab_decompose", "lab_norm_sq", "lab_jl_project", "lab_amortize_z", "lab_prove_core"):
        assert must in names


def test_shared_library_exports_every_declared_symbol():
    import labrador_snark_amd as lab
    assert os.path.exists(lab.SO_PATH), "build the extension first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = lab.load()
    for name in declared_functions():
        assert hasattr(lib, name), "%s declared in include/labrador_hip.h but not exported" % name


def test_python_binding_covers_the_header():
    from labrador_snark_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_no_cpu_fallback(gpu_available):
    import labrador_snark_amd as lab
    if gpu_available:
        pytest.skip("GPU present")
    with pytest.raises(lab.NoDeviceError):
        lab.Context(8191)


def test_product_code_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "labrador_snark_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "coracle" not in text and "pyref" not in text and "labrador_oracle" not in text, f


def test_shard_ranges_partition_the_work():
    from labrador_snark_amd import parallel
    for total in (0, 1, 7, 96, 97):
        for world in (1, 2, 3, 8):
            got = []
            for rank in range(world):
                lo, hi = parallel.shard_range(total, rank, world)
                got.extend(range(lo, hi))
            assert got == list(range(total))
    assert parallel.ik_shard(32, 3, 0, 8) == (0, 12) and parallel.ik_shard(32, 3, 7, 8) == (84, 96)
