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
* hip.rs stays thin (<= 80 lines), implements every required method of the trait and never runs a lone permutation on the GPU;
* the ownership rules a compiler would enforce on the guards, as far as text can show them;
* rust/dusk-hades-0.24.1-hip.patch wires exactly the items the modules define, and applies to the reference when it is here.
"""
import shutil
import subprocess
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
    added = "\n".join(l[1:] for l in PATCH.splitlines() if l.startswith("+") and not l.startswith("+++"))
    build_rs = strip_rust_comments(added)
    assert "rustc-link-lib" not in build_rs and "rustc-link-search=native=" in build_rs and "HADES252_LIB_DIR" in build_rs


def test_hip_rs_is_thin_and_implements_the_trait():
    text = read("hip.rs")
    assert len(text.splitlines()) <= 80
    code = strip_rust_comments(text)
    assert re.search(r"impl\s+Strategy<BlsScalar>\s+for\s+HipStrategy", code)
    for required in ("add_round_key", "quintic_s_box", "mul_matrix", "perm"):     # src/strategies.rs:50-65, :140
        assert re.search(r"\bfn\s+%s\b" % required, code), required
    # the length rule of the reference: copy_from_slice panics unless len == WIDTH (src/strategies/scalar.rs:48)
    assert "data.len() % WIDTH == 0" in code


def test_small_calls_stay_on_the_reference_cpu_path():
    """one GPU call costs ~65 us, one CPU permutation ~50 us -- a literal drop-in under a caller that
    permutes one state per call (README.md:60-61) must not be slower than the reference.  The CPU leg is the REFERENCE'S
    `ScalarStrategy` in the caller's crate, not anything of this repository."""
    code = strip_rust_comments(read("hip.rs"))
    assert re.search(r"pub const MIN_GPU_STATES: usize = (\d+);", code).group(1) == "2"
    assert re.search(r"pub min_gpu_states: usize,", code)
    assert "min_gpu_states: MIN_GPU_STATES" in code and "#[derive(Default)]" not in code
    perm = code[code.index("fn perm("):]
    i_switch, i_ffi = perm.index("if n < self.min_gpu_states"), perm.index("hades252_perm_batch")
    assert i_switch < i_ffi
    assert "data.chunks_mut(WIDTH).for_each(|state| ScalarStrategy::new().perm(state))" in perm[i_switch:i_ffi]
    assert "hades_oracle" not in code and "oracle" not in read("hip.rs").lower()
    # the GPU tests force the device where they mean to test it
    extras = strip_rust_comments(read("hip_extras.rs"))
    assert extras.count("min_gpu_states: 0 }") >= 2 and "min_gpu_states: usize::MAX" in extras


def test_guards_own_what_they_lock():
    """What rustc would enforce, as far as text shows it (no compiler in this image; rust/README.md has the recipe)."""
    code = strip_rust_comments(read("hip_extras.rs"))
    for guard in ("PinnedStates", "PinGuard"):
        decl = re.search(r"((?:#\[[^\]]*\]\s*)*)pub struct %s\b" % guard, code)
        assert decl and "Clone" not in decl.group(1) and "Copy" not in decl.group(1), guard + " must not be Clone/Copy"
        assert not re.search(r"impl(?:<'a>)?\s+(?:Clone|Copy)\s+for\s+%s" % guard, code)
        assert not re.search(r"unsafe\s+impl(?:<'a>)?\s+(?:Send|Sync)\s+for\s+%s" % guard, code)
        g = guard + ("<'a>" if guard == "PinGuard" else "")
        for trait in ("Deref", "DerefMut", "Drop"):
            assert re.search(r"impl(?:<'a>)?\s+%s\s+for\s+%s" % (trait, re.escape(g)), code), (guard, trait)
    drops = dict(re.findall(r"impl(?:<'a>)?\s+Drop\s+for\s+(\w+).*?fn drop\(&mut self\)\s*\{(.*?)\n    \}", code, flags=re.S))
    assert "hades252_host_free(self.ptr" in drops["PinnedStates"]
    assert "hades252_host_unregister(self.ptr" in drops["PinGuard"]
    # register and unregister are guarded by the same emptiness condition
    assert "if !data.is_empty()" in code and "if self.len != 0" in drops["PinGuard"]
    # fields private: nobody can forge a guard around a pointer the library never saw
    for guard in ("PinnedStates", "PinGuard<'a>"):
        body = re.search(r"pub struct %s \{(.*?)\}" % re.escape(guard), code, flags=re.S).group(1)
        assert "pub " not in body
    # every FFI call's return code is looked at (Drop may ignore it: nothing sensible to do there)
    for m in re.finditer(r"\bhades252_(\w+)\(", code.split("#[cfg(test)]")[0]):
        name = m.group(1)
        before = code[max(0, m.start() - 90):m.start()]
        if re.search(r"fn\s+$", before) or name in ("host_free", "host_unregister"):
            continue
        assert "check(" in before, "return code of hades252_%s is dropped" % name


PATCH_PATH = os.path.join(ROOT, "rust", "dusk-hades-0.24.1-hip.patch")
with open(PATCH_PATH) as _f:
    PATCH = _f.read()


def test_patch_wires_what_the_modules_define():
    added = [l[1:] for l in PATCH.splitlines() if l.startswith("+") and not l.startswith("+++")]
    removed = [l for l in PATCH.splitlines() if l.startswith("-") and not l.startswith("---")]
    assert not removed, "the patch only adds lines"
    assert sorted(re.findall(r"^\+\+\+ b/(\S+)", PATCH, flags=re.M)) == ["Cargo.toml", "build.rs", "src/lib.rs", "src/strategies.rs"]
    text = "\n".join(added)
    for must in ("hip = []", "extern crate alloc;", "mod hip;", "mod hip_extras;", "pub mod hip_sys;",
                 "pub use hip::{HipStrategy, MIN_GPU_STATES};", "pub use hip_extras::{PinGuard, PinnedStates};"):
        assert must in text, must
    # every re-exported name is a `pub` item of the module it is taken from
    for mod, names in (("hip.rs", ("HipStrategy", "MIN_GPU_STATES")), ("hip_extras.rs", ("PinGuard", "PinnedStates"))):
        code = strip_rust_comments(read(mod))
        for n in names:
            assert re.search(r"pub (?:struct|const) %s\b" % n, code), (mod, n)
    # every added item line is feature-gated and (for mods) documented: deny(missing_docs), src/lib.rs:7
    for i, l in enumerate(added):
        if re.match(r"\s*(?:pub )?(?:mod|use|extern crate) ", l):
            j = i - 1
            while added[j].lstrip().startswith("#[allow"):
                j -= 1
            assert added[j].strip() == '#[cfg(feature = "hip")]', l
            if " mod " in " " + l:
                assert added[j - 1].lstrip().startswith("///"), l
    # the modules resolve their own imports inside src/strategies/: super = strategies, crate = dusk_hades
    assert "use super::{ScalarStrategy, Strategy};" in read("hip.rs") and "use crate::WIDTH;" in read("hip.rs")
    assert "use super::hip::{check, HipStrategy};" in read("hip_extras.rs") and "pub(crate) fn check" in read("hip.rs")


@pytest.mark.skipif(not os.path.isdir("/root/reference/src") or shutil.which("git") is None,
                    reason="needs the reference checkout (absent on the GPU box)")
def test_patch_and_apply_script_work_on_the_reference(tmp_path):
    crate = tmp_path / "dusk-hades"
    shutil.copytree("/root/reference", crate)
    script = os.path.join(ROOT, "rust", "apply.sh")
    for _ in range(2):                                                # idempotent
        subprocess.run(["sh", script, str(crate)], check=True, capture_output=True)
    for f in ("hip.rs", "hip_extras.rs", "hip_sys.rs"):
        assert (crate / "src" / "strategies" / f).read_text() == read(f)
    assert (crate / "tests" / "kat_scalar.rs").exists() and (crate / "assets" / "ark.bin").exists()
    assert 'version = "0.24.1"' in (crate / "Cargo.toml").read_text()
    lib = (crate / "src" / "lib.rs").read_text()
    assert lib.index("extern crate alloc;") > lib.index("#![no_std]") and lib.count('feature = "hip"') == 2
    assert (crate / "src" / "strategies.rs").read_text().count('feature = "hip"') == 5
    assert (crate / "Cargo.toml").read_text().rstrip().endswith("hip = []")
    subprocess.run(["git", "apply", "--check", "-R", PATCH_PATH], cwd=crate, check=True)


def test_pin_guard_holds_the_borrow():
    """the guard must keep the slice borrowed while its pages are registered."""
    code = strip_rust_comments(read("hip_extras.rs"))
    assert re.search(r"pub struct PinGuard<'a>", code) and "PhantomData<&'a mut [BlsScalar]>" in code
    assert re.search(r"pub fn new\(data: &'a mut \[BlsScalar\]\)", code)


def test_integration_md_shows_the_wiring():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for must in ("rust/apply.sh", "dusk-hades-0.24.1-hip.patch", "cargo test --test kat_scalar", "rust/README.md",
                 "MIN_GPU_STATES", "min_gpu_states"):
        assert must in text, must
