"""CPU tier: the Rust binding (rust/src/*.rs, source only -- no Rust toolchain in this image) is tied mechanically to
include/hades252.h, and checked against the lints of the crate it is meant for.

Reference constraints: `#![deny(missing_docs)]` and `#![no_std]` (src/lib.rs:7-8), trait `Strategy<T>`
(src/strategies.rs:31-163), feature pattern (Cargo.toml:25-26).

* every `fn hades252_*` of every `extern "C"` block: name declared in the header, same arity, and every parameter /
  return type is the Rust FFI equivalent of the C type (table tools/gen_rust_ffi.py::C_TO_RUST);
* rust/src/hip_sys.rs (the complete generated binding) is current;
* every `pub` item is preceded by a `///` doc comment (deny(missing_docs));
* no `std::` path and no bare `Vec` / `vec!` / `String` / `Box` without an `alloc` import (no_std);
* the library is named for the linker exactly once (`#[link]` on one extern block; build.rs gives the search path only);
* hip.rs stays thin (<= 60 lines) and implements every required method of the trait.
"""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_ffi as G  # noqa: E402

RUST_DIR = os.path.join(ROOT, "rust", "src")
RUST_FILES = sorted(f for f in os.listdir(RUST_DIR) if f.endswith(".rs"))


def read(name):
    with open(os.path.join(RUST_DIR, name)) as f:
        return f.read()


def strip_rust_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return "\n".join(re.sub(r"//.*$", "", line) for line in text.splitlines())


def extern_blocks(text):
    """Bodies of the `extern "C" { ... }` blocks (extern blocks hold no nested braces)."""
    return re.findall(r'extern\s+"C"\s*\{(.*?)\}', strip_rust_comments(text), flags=re.S)


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def norm_rust_type(t):
    t = re.sub(r"\s+", " ", t).strip()
    t = t.replace("core::ffi::", "").replace("std::ffi::", "").replace("std::os::raw::", "")
    return t


def rust_decls(text):
    decls = []
    for body in extern_blocks(text):
        for m in re.finditer(r"(?:pub\s+)?fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", body, flags=re.S):
            name, args, ret = m.group(1), m.group(2), m.group(3)
            params = []
            for a in split_args(args):
                pname, ptype = a.split(":", 1)
                params.append((pname.strip(), norm_rust_type(ptype)))
            decls.append((name, params, norm_rust_type(ret) if ret else "()"))
    return decls


HEADER_FUNCS = {name: (ret, plist) for name, ret, plist in G.parse_header()[0]}


def test_header_parser_sees_every_declared_symbol():
    import test_abi
    assert sorted(HEADER_FUNCS) == test_abi.declared_symbols()


@pytest.mark.parametrize("fname", RUST_FILES)
def test_extern_blocks_match_the_header(fname):
    decls = rust_decls(read(fname))
    if fname in ("hip.rs", "hip_extras.rs", "hip_sys.rs"):
        assert decls, "%s: no extern \"C\" declarations found" % fname
    for name, params, ret in decls:
        assert name in HEADER_FUNCS, "%s: %s is not declared in include/hades252.h" % (fname, name)
        c_ret, c_params = HEADER_FUNCS[name]
        assert len(params) == len(c_params), "%s: %s takes %d arguments, the header says %d" % (
            fname, name, len(params), len(c_params))
        for (pname, ptype), (ctype, cname) in zip(params, c_params):
            assert ptype == G.rust_type(ctype), "%s: %s(%s): Rust %s vs C %s (= %s)" % (
                fname, name, cname, ptype, ctype, G.rust_type(ctype))
            assert pname.replace("r#", "") == cname or fname != "hip_sys.rs"
        assert ret == G.rust_type(c_ret), "%s: %s returns %s, the header says %s" % (fname, name, ret, c_ret)


def test_what_perm_binds_is_what_the_header_says():
    """The one call the reference's `perm(&mut [BlsScalar])` turns into (src/strategies.rs:140)."""
    d = {n: (p, r) for n, p, r in rust_decls(read("hip.rs"))}
    assert d["hades252_perm_batch"] == ([("states", "*mut u64"), ("n_perms", "usize")], "i32")
    assert d["hades252_perm_batch_multi"] == ([("states", "*mut u64"), ("n_perms", "usize"), ("n_devices", "i32")], "i32")
    assert d["hades252_strerror"] == ([("code", "i32")], "*const c_char")


def test_generated_sys_file_is_current_and_complete():
    assert read("hip_sys.rs") == G.render(), "rust/src/hip_sys.rs is stale: python tools/gen_rust_ffi.py"
    assert sorted(n for n, _, _ in rust_decls(read("hip_sys.rs"))) == sorted(HEADER_FUNCS)
    consts = dict(re.findall(r"pub const (HADES252_\w+): [iu]32 = (-?\d+);", read("hip_sys.rs")))
    assert consts["HADES252_WIDTH"] == "5" and consts["HADES252_ERR_OUT_OF_CONSTANTS"] == "-6"
    assert {c for c, _, _ in G.parse_header()[1]} == set(consts)


PUB_ITEM = re.compile(r"^\s*pub\s+(?:unsafe\s+)?(?:const\s+fn|fn|struct|enum|trait|const|static|type|mod|use)\b|^\s*pub\s+\w+\s*:")


@pytest.mark.parametrize("fname", RUST_FILES)
def test_every_pub_item_is_documented(fname):
    """`#![deny(missing_docs)]` (src/lib.rs:7) makes an undocumented public item a hard error."""
    lines = read(fname).splitlines()
    for i, line in enumerate(lines):
        if not PUB_ITEM.match(line) or re.match(r"^\s*pub\s*\(", line) or re.match(r"^\s*pub\s+use\b", line):
            continue
        j = i - 1
        while j >= 0 and re.match(r"^\s*#\[", lines[j]):      # attributes sit between the doc comment and the item
            j -= 1
        assert j >= 0 and lines[j].lstrip().startswith("///"), "%s:%d: public item without a doc comment: %s" % (
            fname, i + 1, line.strip())


@pytest.mark.parametrize("fname", RUST_FILES)
def test_no_std_discipline(fname):
    """`#![no_std]` (src/lib.rs:8): no `std::` outside #[cfg(test)], heap types only through `alloc`."""
    text = strip_rust_comments(read(fname))
    body = text.split("#[cfg(test)]")[0]
    assert not re.search(r"\bstd::", body), fname + ": std:: path in a no_std crate"
    for ident, imp in ((r"\bVec\b", "alloc::vec::Vec"), (r"\bvec!", "alloc::vec"), (r"\bString\b", "alloc::string::String"),
                       (r"\bBox\b", "alloc::boxed::Box")):
        if re.search(ident, body):
            assert re.search(r"use\s+" + re.escape(imp) + r"\s*;", body), "%s uses %s without `use %s;`" % (fname, ident, imp)


def test_library_is_named_for_the_linker_once():
    links = sum(len(re.findall(r"#\[link\s*\(\s*name\s*=\s*\"hades252\"", strip_rust_comments(read(f)))) for f in RUST_FILES)
    assert links == 1
    build_rs = strip_rust_comments(open(os.path.join(ROOT, "rust", "build.rs")).read())
    assert "rustc-link-lib" not in build_rs and "rustc-link-search" in build_rs


def test_hip_rs_is_thin_and_implements_the_trait():
    text = read("hip.rs")
    assert len(text.splitlines()) <= 60
    code = strip_rust_comments(text)
    assert re.search(r"impl\s+Strategy<BlsScalar>\s+for\s+HipStrategy", code)
    for required in ("add_round_key", "quintic_s_box", "mul_matrix", "perm"):     # src/strategies.rs:50-65, :140
        assert re.search(r"\bfn\s+%s\b" % required, code), required
    # the length rule of the reference: copy_from_slice panics unless len == WIDTH (src/strategies/scalar.rs:48)
    assert "data.len() % WIDTH == 0" in code


def test_pin_guard_holds_the_borrow():
    """the guard must keep the slice borrowed while its pages are registered."""
    code = strip_rust_comments(read("hip_extras.rs"))
    assert re.search(r"pub struct PinGuard<'a>", code) and "PhantomData<&'a mut [BlsScalar]>" in code
    assert re.search(r"pub fn new\(data: &'a mut \[BlsScalar\]\)", code)


def test_integration_md_shows_the_wiring():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for must in ('#[cfg(feature = "hip")]', "extern crate alloc;", "mod hip;", "pub use hip::HipStrategy;", "hip = []"):
        assert must in text, must
