"""The Rust binding crate (bindings/rdr-sys) is source only here (no Rust toolchain): check it against
the header it mirrors, declaration by declaration."""
import os
import re

from renderer_amd.renderer import ABI_SYMBOLS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "rdr_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|void\*?|const char\*)\s+(rdr_\w+)\s*\(([^;{]*?)\)\s*;", text):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def _rust_functions():
    text = open(os.path.join(ROOT, "bindings", "rdr-sys", "src", "lib.rs")).read()
    out = {}
    for m in re.finditer(r"pub fn (rdr_\w+)\(([^)]*)\)", text):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else args.count(",") + 1
    return out


def test_every_header_function_is_bound_with_the_same_arity():
    header, rust = _header_functions(), _rust_functions()
    assert set(header) == set(ABI_SYMBOLS), set(header) ^ set(ABI_SYMBOLS)
    assert set(rust) == set(header), set(rust) ^ set(header)
    for name, n in header.items():
        assert rust[name] == n, (name, n, rust[name])


def test_struct_fields_follow_the_header_order():
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rdr_abi.h")).read(), flags=re.S)
    rust = open(os.path.join(ROOT, "bindings", "rdr-sys", "src", "lib.rs")).read()
    for name in ("rdr_material", "rdr_scene_desc", "rdr_stats", "rdr_config", "rdr_ubo", "rdr_mesh_draw", "rdr_meshlet"):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S).group(1)
        c_fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
            for part in names.split(","):
                c_fields.append(re.sub(r"\[.*?\]|\*|\s", "", part.strip().split()[-1]))
        r_body = re.search(r"pub struct %s \{(.*?)\n?\}" % name, rust, re.S).group(1)
        r_fields = re.findall(r"pub (\w+):", r_body)
        assert r_fields == c_fields, (name, r_fields, c_fields)
