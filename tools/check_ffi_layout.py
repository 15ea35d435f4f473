#!/usr/bin/env python3
"""Mechanical check of rust/rmf_crowdsim_gpu/src/ffi.rs against include/crowdstep.h.

No Rust toolchain exists in the build image, so the Rust binding cannot be compiled here.  What
CAN be checked without rustc is that the two declarations of the C ABI say the same thing:

  1. the same set of functions, each with the same number of arguments, of the same types, in
     the same order, and the same return type;
  2. the same structs with the same fields, of the same types, in the same order;
  3. the same constants (#define / enum values vs `pub const`);
  4. the same callback signatures (typedef'd function pointers vs `Option<unsafe extern "C" fn>`);
  5. the sizes and field offsets #[repr(C)] gives the Rust structs on x86-64 (computed here by the
     repr(C) rules: every field aligned to its own alignment, size rounded up to the largest) equal
     the C compiler's: a generated translation unit of static_asserts over sizeof / offsetof is
     compiled with g++ against the real header.

Run: python tools/check_ffi_layout.py   (exit code 0 = in agreement; used by
tests/test_rust_shim_layout.py).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "crowdstep.h")
FFI_RS = os.path.join(ROOT, "rust", "rmf_crowdsim_gpu", "src", "ffi.rs")

# canonical type spelling shared by both sides
C_SCALARS = {
    "double": "f64", "float": "f32", "uint8_t": "u8", "uint32_t": "u32", "int32_t": "i32",
    "uint64_t": "u64", "size_t": "usize", "int": "c_int", "char": "c_char", "void": "void",
}
RUST_SCALARS = {"f64", "f32", "u8", "u32", "i32", "u64", "usize", "c_int", "c_char"}
SIZE_ALIGN = {"f64": (8, 8), "f32": (4, 4), "u8": (1, 1), "u32": (4, 4), "i32": (4, 4), "u64": (8, 8),
              "usize": (8, 8), "c_int": (4, 4), "c_char": (1, 1), "ptr": (8, 8)}


def strip_c_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def canon_c_type(t):
    """'const double*' -> '*const f64'; 'cs_engine*' -> '*mut cs_engine'; 'const cs_x**' -> '*mut *const cs_x';
    'const void* const*' -> '*const *const c_void'; 'void* const*' -> '*const *mut c_void' (a `const` qualifies what
    stands to its left, or the base type when it comes first)."""
    t = t.strip()
    parts = [p.split() for p in t.split("*")]  # parts[0] = base type words, parts[k] = qualifiers after the k-th star
    stars = len(parts) - 1
    base = [w for w in parts[0] if w not in ("const", "struct")]
    assert len(base) == 1, t
    name = C_SCALARS.get(base[0], base[0])
    if stars == 0:
        return name
    if name == "void":
        name = "c_void"
    # pointee constness per level: level 1 points at the base, level k at the (k-1)-th pointer
    pointee_const = ["const" in parts[0]] + ["const" in parts[k] for k in range(1, stars)]
    out = name
    for k in range(stars):
        out = ("*const " if pointee_const[k] else "*mut ") + out
    return out


def split_args(arglist):
    arglist = arglist.strip()
    if arglist in ("", "void"):
        return []
    return [a.strip() for a in arglist.split(",")]


def c_arg_type(arg):
    """'const double* xy' -> canonical type (the name, if present, is dropped)."""
    m = re.match(r"^(.*?[\s\*])([A-Za-z_]\w*)$", arg.strip())
    if m and m.group(2) not in C_SCALARS and not m.group(2).startswith("cs_") and m.group(1).strip():
        return canon_c_type(m.group(1))
    if m and m.group(1).strip() and (m.group(2) in ("flags", "out", "n", "cap") or True):
        # `T name` form: the last identifier is the parameter name unless it is the type itself
        head = m.group(1).strip()
        if head and head not in ("const",):
            return canon_c_type(head)
    return canon_c_type(arg)


def parse_header(path):
    text = strip_c_comments(open(path).read())
    consts, structs, fns, callbacks = {}, {}, {}, {}
    for m in re.finditer(r"#define\s+(CS_\w+)\s+([0-9xXa-fA-F]+)[uU]?\s*$", text, flags=re.M):
        consts[m.group(1)] = int(m.group(2), 0)
    for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", text, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = [s.strip() for s in item.split("=")]
                nxt = int(val, 0)
            else:
                name = item
            consts[name] = nxt
            nxt += 1
    for m in re.finditer(r"typedef\s+([\w\s\*]+?)\(\s*\*\s*(cs_\w+)\s*\)\s*\((.*?)\)\s*;", text, flags=re.S):
        callbacks[m.group(2)] = (canon_c_type(m.group(1)), [c_arg_type(a) for a in split_args(m.group(3))])
    for m in re.finditer(r"typedef\s+struct\s+(cs_\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"^(.*?[\s\*])([\w\s,]+)$", decl)
            ctype, names = mm.group(1), [n.strip() for n in mm.group(2).split(",")]
            for n in names:
                fields.append((n, canon_c_type(ctype)))
        structs[m.group(1)] = fields
    body = re.sub(r"typedef\s+struct\s+cs_\w+\s*\{.*?\}\s*cs_\w+\s*;", " ", text, flags=re.S)
    body = re.sub(r"^\s*#.*$", " ", body, flags=re.M)  # preprocessor lines
    body = re.sub(r"extern\s+\"C\"\s*\{", " ", body)
    body = re.sub(r"typedef[^;]*;", " ", body)
    body = re.sub(r"enum\s*\{.*?\}\s*;", " ", body, flags=re.S)
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(cs_\w+)\s*\(([^()]*)\)\s*;", body, flags=re.S):
        ret = canon_c_type(m.group(1))
        fns[m.group(2)] = (ret, [c_arg_type(a) for a in split_args(m.group(3))])
    return consts, structs, fns, callbacks


def canon_rust_type(t):
    return " ".join(t.strip().split())


def parse_rust_fn_sig(sig):
    """'(user: *mut c_void, n: usize) -> usize' -> (ret, [types])"""
    m = re.match(r"^\((.*)\)\s*(?:->\s*(.+))?$", sig.strip(), flags=re.S)
    args = []
    for a in split_args(m.group(1)):
        args.append(canon_rust_type(a.split(":", 1)[1]))
    return (canon_rust_type(m.group(2)) if m.group(2) else "void", args)


def parse_rust(path):
    text = re.sub(r"//[^\n]*", " ", open(path).read())
    consts, structs, fns, callbacks = {}, {}, {}, {}
    for m in re.finditer(r"pub const (CS_\w+):\s*\w+\s*=\s*([0-9xXa-fA-F_]+)\s*;", text):
        consts[m.group(1)] = int(m.group(2).replace("_", ""), 0)
    for m in re.finditer(r"pub type (cs_\w+)\s*=\s*Option<unsafe extern \"C\" fn(\(.*?\)(?:\s*->\s*[\w\s\*]+)?)>\s*;", text, flags=re.S):
        callbacks[m.group(1)] = parse_rust_fn_sig(m.group(2))
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*pub struct (cs_\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(","):
            decl = decl.strip()
            if not decl:
                continue
            name, t = decl.split(":", 1)
            name = name.replace("pub", "").strip()
            if name.startswith("_"):
                continue  # the opaque handle's zero-sized marker
            fields.append((name, canon_rust_type(t)))
        structs[m.group(1)] = fields
    ext = re.search(r"extern \"C\" \{(.*?)\n\}", text, flags=re.S)
    for m in re.finditer(r"pub fn (cs_\w+)(\(.*?\)(?:\s*->\s*[^;]+)?);", ext.group(1), flags=re.S):
        fns[m.group(1)] = parse_rust_fn_sig(m.group(2))
    return consts, structs, fns, callbacks


def rust_layout(fields, structs, callbacks):
    """repr(C) on x86-64: (size, align, [(name, offset)])."""
    off, align, out = 0, 1, []
    for name, t in fields:
        if t.startswith("*") or t in callbacks:
            s, a = SIZE_ALIGN["ptr"]
        elif t in SIZE_ALIGN:
            s, a = SIZE_ALIGN[t]
        elif t in structs:
            s, a, _ = rust_layout(structs[t], structs, callbacks)
        else:
            raise SystemExit(f"unknown Rust field type {t!r}")
        off = (off + a - 1) // a * a
        out.append((name, off))
        off += s
        align = max(align, a)
    return (off + align - 1) // align * align, align, out


def main():
    c_consts, c_structs, c_fns, c_cbs = parse_header(HEADER)
    r_consts, r_structs, r_fns, r_cbs = parse_rust(FFI_RS)
    problems = []

    def compare(kind, c, r):
        for name in sorted(set(c) | set(r)):
            if name not in r:
                problems.append(f"{kind} {name}: in crowdstep.h, missing from ffi.rs")
            elif name not in c:
                problems.append(f"{kind} {name}: in ffi.rs, not in crowdstep.h")
            elif c[name] != r[name]:
                problems.append(f"{kind} {name}: crowdstep.h {c[name]} != ffi.rs {r[name]}")

    r_structs_cmp = {k: v for k, v in r_structs.items() if k not in ("cs_engine", "cs_mesh")}  # (opaque handles)
    compare("const", c_consts, r_consts)
    compare("callback", c_cbs, r_cbs)
    compare("struct", c_structs, r_structs_cmp)
    compare("fn", c_fns, r_fns)

    # sizes / offsets: Rust's repr(C) rule here, the C compiler's through static_assert
    lines = ['#include <cstddef>', '#include "crowdstep.h"']
    for name, fields in sorted(r_structs_cmp.items()):
        if name not in c_structs:
            continue
        size, _, offs = rust_layout(fields, r_structs, r_cbs)
        lines.append(f'static_assert(sizeof({name}) == {size}, "sizeof {name}");')
        for fname, off in offs:
            lines.append(f'static_assert(offsetof({name}, {fname}) == {off}, "offsetof {name}.{fname}");')
    lines.append("int main() { return 0; }")
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "layout_check.cpp")
        with open(src, "w") as f:
            f.write("\n".join(lines) + "\n")
        proc = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src],
                              capture_output=True, text=True)
        if proc.returncode != 0:
            problems.append("layout static_asserts failed:\n" + proc.stderr)
    n_assert = sum(1 for ln in lines if ln.startswith("static_assert"))
    if problems:
        print("\n".join(problems))
        return 1
    print(f"ffi.rs == crowdstep.h: {len(c_fns)} functions, {len(c_structs)} structs, {len(c_cbs)} callbacks, "
          f"{len(c_consts)} constants, {n_assert} size/offset assertions compiled")
    return 0


if __name__ == "__main__":
    sys.exit(main())
