#!/usr/bin/env python3
"""Print the `extern "C"` block of rust/innr-hip's `mod ffi` from include/innr_hip.h (one binding per prototype), so the
shim covers the whole header by construction. tests/test_abi.py::test_rust_shim_binds_the_whole_header checks the committed
lib.rs against the header; this script is how the block is regenerated after the header changes:

    python tools/gen_rust_ffi.py            # prints the block
    python tools/gen_rust_ffi.py --write    # rewrites the block between the markers in rust/innr-hip/src/lib.rs
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "innr_hip.h")
SHIM = os.path.join(ROOT, "rust", "innr-hip", "src", "lib.rs")
BEGIN, END = "        // ---- generated from include/innr_hip.h by tools/gen_rust_ffi.py: begin", "        // ---- generated: end"

SCALAR = {"int": "c_int", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "float": "f32", "innr_status": "c_int",
          "uint8_t": "u8", "long": "c_long"}
OPAQUE = {"innr_ctx": "InnrCtx", "innr_batch": "InnrBatch", "innr_docs": "InnrDocs", "innr_comm": "InnrComm",
          "innr_knn_stats": "InnrKnnStats"}
RUST_KEYWORDS = {"in", "type", "ref", "box", "move", "loop", "match", "mod", "fn", "use", "as", "where"}


def rust_type(ctype: str) -> str:
    ctype = ctype.strip()
    stars = ctype.count("*")
    base = ctype.replace("*", " ").replace("const", " ").split()
    const = ctype.startswith("const")
    b = " ".join(base)
    if stars == 0:
        return SCALAR[b]
    inner = {"void": "c_void", "char": "c_char"}.get(b) or SCALAR.get(b) or OPAQUE[b]
    out = inner
    for i in range(stars):
        out = ("*const " if (const and i == 0) else "*mut ") + out
    return out


def prototypes():
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    hdr = re.sub(r"^\s*#.*$", "", hdr, flags=re.M)  # preprocessor lines
    for m in re.finditer(r"([\w\s\*]+?)\b(innr_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;", hdr):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args not in ("", "void"):
            for i, a in enumerate(args.split(",")):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                ty, nm = (mm.group(1), mm.group(2)) if mm and not a.endswith("*") else (a, f"a{i}")
                if nm in RUST_KEYWORDS:
                    nm += "_"
                params.append((nm.lower(), rust_type(ty)))
        yield name, ret, params


def block() -> str:
    lines = [BEGIN]
    for name, ret, params in prototypes():
        args = ", ".join(f"{n}: {t}" for n, t in params)
        r = "" if ret == "void" else f" -> {rust_type(ret)}"
        lines.append(f"        pub fn {name}({args}){r};")
    lines.append(END)
    return "\n".join(lines)


if __name__ == "__main__":
    b = block()
    if "--write" in sys.argv:
        s = open(SHIM).read()
        i, j = s.index(BEGIN), s.index(END) + len(END)
        open(SHIM, "w").write(s[:i] + b + s[j:])
    else:
        print(b)
