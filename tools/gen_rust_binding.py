#!/usr/bin/env python3
"""Generates the Rust side of the C ABI (`src/hip.rs` of the reference crate) from include/labrador_hip.h: the
#[repr(C)] structs and the `extern "C"` block, field by field and entry point by entry point.  INTEGRATION.md embeds
the output; tests/test_host_and_abi.py regenerates it and requires the document to contain it verbatim, so the printed
binding cannot drift from the header again.

    python tools/gen_rust_binding.py            # prints the Rust text
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCALARS = {"int": "i32", "size_t": "usize", "uint64_t": "u64", "int64_t": "i64", "uint32_t": "u32", "uint16_t": "u16",
           "uint8_t": "u8", "int8_t": "i8", "double": "f64", "void": "c_void", "char": "c_char", "lab_op": "i32",
           "lab_ctx": "LabCtx", "lab_vec": "LabVec", "lab_mat8": "LabMat8", "lab_info": "LabInfo",
           "lab_prove_input": "LabProveInput", "lab_transcript": "LabTranscript", "lab_crs_stream": "LabCrsStream"}
FIELD_NAMES = {"S": "s", "A": "a_mat", "B": "b_mat", "C": "c_mat", "Dm": "d_mat"}
KEYWORDS = {"in": "input", "type": "kind", "ref": "reference", "fn": "func", "mod": "modulus", "loop": "lp", "move": "mv"}


def rust_ident(name, field=False):
    if field and name in FIELD_NAMES:
        return FIELD_NAMES[name]
    n = re.sub(r"(?<=[a-z0-9])(?=[A-Z])", "_", name).lower()
    return KEYWORDS.get(n, n)


def rust_type(ctype):
    """`const uint64_t *`, `lab_vec **`, `const lab_prove_input *const *`, `int` ... -> Rust."""
    t = ctype.replace("struct ", "").strip()
    toks = re.findall(r"const|\*|[A-Za-z_][A-Za-z0-9_]*", t)
    base, base_const, ptrs = None, False, []
    i = 0
    while i < len(toks):
        tok = toks[i]
        if tok == "const" and base is None:
            base_const = True
        elif tok == "const":                 # qualifies the pointer to its left
            ptrs[-1] = "const"
        elif tok == "*":
            ptrs.append("mut")
        else:
            if base is None:
                base = tok
            else:
                raise ValueError("cannot parse type %r" % ctype)
        i += 1
    out = SCALARS[base]
    # the innermost pointer's pointee constness is the base's; every outer level's is the qualifier of the level inside
    quals = [("const" if base_const else "mut")] + ptrs[:-1] if ptrs else []
    for q in quals:
        out = "*%s %s" % (q, out)
    return out


def parse(header_text):
    src = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    structs = []
    for body, name in re.findall(r"typedef struct \{(.*?)\}\s*(lab_[a-z0-9_]+)\s*;", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"((?:const )?(?:struct )?[A-Za-z_][A-Za-z0-9_]*)\s*(.*)$", decl)
            base, rest = m.group(1), m.group(2)
            for part in rest.split(","):
                part = part.strip()
                pm = re.match(r"(\**)\s*([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[(\d+)\])?$", part)
                stars, fname, arr = pm.group(1), pm.group(2), pm.group(3)
                ty = rust_type(base + " " + stars)
                if arr:
                    ty = "[%s; %s]" % (ty, arr)
                fields.append((fname, ty))
        structs.append((name, fields))
    funcs = []
    body = re.sub(r"typedef struct \{.*?\}\s*lab_[a-z0-9_]+\s*;", "", src, flags=re.S)
    body = re.sub(r"typedef enum \{.*?\}\s*lab_[a-z0-9_]+\s*;", "", body, flags=re.S)
    for ret, name, args in re.findall(r"([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(lab_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", body, flags=re.S):
        params = []
        for a in " ".join(args.split()).split(","):
            a = a.strip()
            if a in ("", "void"):
                continue
            am = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[\d*\])?$", a)
            ty, pname, arr = am.group(1).strip(), am.group(2), am.group(3)
            if arr:
                ty += " *"
            params.append((rust_ident(pname), rust_type(ty)))
        ret = " ".join(ret.split())
        funcs.append((name, params, None if ret == "void" else rust_type(ret)))
    return structs, funcs


def wrap(prefix, items, indent, width=118):
    lines, cur = [], prefix
    for k, it in enumerate(items):
        piece = it + ("," if k + 1 < len(items) else "")
        if len(cur) + 1 + len(piece) > width and cur.strip():
            lines.append(cur.rstrip())
            cur = indent + piece
        else:
            cur = cur + ("" if cur.endswith("(") or not cur.strip() else " ") + piece
    lines.append(cur)
    return lines


def generate(header_text=None):
    if header_text is None:
        header_text = open(os.path.join(ROOT, "include", "labrador_hip.h")).read()
    structs, funcs = parse(header_text)
    out = ["use std::os::raw::{c_char, c_void};", "",
           "#[repr(C)] pub struct LabCtx { _p: [u8; 0] }", "#[repr(C)] pub struct LabVec { _p: [u8; 0] }",
           "#[repr(C)] pub struct LabMat8 { _p: [u8; 0] }", ""]
    for name, fields in structs:
        out.append("#[repr(C)]")
        out.append("pub struct %s {" % SCALARS[name])
        out += wrap("    ", ["pub %s: %s" % (rust_ident(f, True), t) for f, t in fields], "    ")
        out[-1] += ","
        out.append("}")
    out.append("")
    out.append('extern "C" {')
    for name, params, ret in funcs:
        sig = wrap("    pub fn %s(" % name, ["%s: %s" % p for p in params], " " * (12 + len(name)))
        sig[-1] += ")" + (" -> %s" % ret if ret else "") + ";"
        out += sig
    out.append("}")
    return "\n".join(out)


if __name__ == "__main__":
    sys.stdout.write(generate() + "\n")
