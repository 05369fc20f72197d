"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header declares
(no compute calls without a GPU), the Python binding covers them, and the product path fails loudly without a GPU."""
import ctypes
import os
import re
import sys

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


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of every boundary struct as the C compiler lays it out (include/labrador_hip.h) equal the
    ctypes mirrors in _lib.py field by field — a binding that forgets a field (INTEGRATION.md once did) shows up here."""
    import subprocess
    from labrador_snark_amd import _lib
    structs = {"lab_prove_input": _lib.lab_prove_input, "lab_transcript": _lib.lab_transcript, "lab_info": _lib.lab_info}
    # the field names of each struct, parsed from the header itself
    src = open(os.path.join(ROOT, "include", "labrador_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    fields = {}
    for body, name in re.findall(r"typedef struct \{(.*?)\}\s*(lab_[a-z_]+)\s*;", src, flags=re.S):
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[\d+\])?\s*$", part.strip())
                names.append(m.group(1))
        fields[name] = names
    assert set(structs) <= set(fields)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "labrador_hip.h"', 'int main(void) {']
    for sname in structs:
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (sname, sname))
        for f in fields[sname]:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (sname, f, sname, f))
    lines += ['  return 0;', '}']
    cfile = tmp_path / "layout.c"
    cfile.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(cfile), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for sname, cls in structs.items():
        assert [f for f, _ in cls._fields_] == fields[sname], "%s: field order differs from the header" % sname
        assert int(got[sname]) == ctypes.sizeof(cls), sname
        for f, _ in cls._fields_:
            assert int(got["%s.%s" % (sname, f)]) == getattr(cls, f).offset, (sname, f)


def test_integration_doc_carries_the_generated_rust_binding():
    """INTEGRATION.md's Rust binding is the output of tools/gen_rust_binding.py on the current header: every field of the
    boundary structs in header order and every exported entry point."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_binding
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- BEGIN GENERATED RUST BINDING -->\n```rust\n(.*?)\n```\n<!-- END GENERATED RUST BINDING -->", doc, flags=re.S)
    assert m, "generated binding block missing from INTEGRATION.md"
    assert m.group(1) == gen_rust_binding.generate(), "run tools/gen_rust_binding.py and paste its output into INTEGRATION.md"
    from labrador_snark_amd import _lib
    structs, funcs = gen_rust_binding.parse(open(os.path.join(ROOT, "include", "labrador_hip.h")).read())
    assert sorted(name for name, _, _ in funcs) == declared_functions()
    by_name = dict(structs)
    assert [f for f, _ in by_name["lab_prove_input"]] == [f for f, _ in _lib.lab_prove_input._fields_]
    assert [f for f, _ in by_name["lab_transcript"]] == [f for f, _ in _lib.lab_transcript._fields_]


def test_generated_gemm_loop_is_what_the_generator_emits():
    """labrador_snark_amd/csrc/gemm_w3_asm.inc (the assembly K loop of the dominant GEMM) is exactly the output of
    tools/gen_gemm_w3_asm.py: the committed include was not edited by hand and the generator has not drifted from it."""
    import subprocess
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "gen_gemm_w3_asm.py")]).decode()
    assert out == open(os.path.join(ROOT, "labrador_snark_amd", "csrc", "gemm_w3_asm.inc")).read()
