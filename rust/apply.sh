#!/bin/sh
# Wire libhades252 into a checkout of dusk-hades 0.24.1 as the cargo feature `hip`.
#   usage: rust/apply.sh /path/to/dusk-hades
# Copies the three strategy modules and the GPU-free known-answer test, then patches Cargo.toml, build.rs, src/lib.rs and
# src/strategies.rs (dusk-hades-0.24.1-hip.patch).  Idempotent: a second run only refreshes the copied files.
set -eu
here=$(cd "$(dirname "$0")" && pwd)
crate=${1:?usage: apply.sh <dusk-hades checkout>}
[ -f "$crate/src/strategies/scalar.rs" ] || { echo "$crate is not a dusk-hades checkout" >&2; exit 2; }
grep -q '^version = "0.24.1"' "$crate/Cargo.toml" || echo "warning: $crate is not dusk-hades 0.24.1; the patch may not apply" >&2
cp "$here/src/hip.rs" "$here/src/hip_extras.rs" "$here/src/hip_sys.rs" "$crate/src/strategies/"
mkdir -p "$crate/tests"
cp "$here/tests/kat_scalar.rs" "$crate/tests/"
cd "$crate"
if grep -q '^hip = \[\]' Cargo.toml; then
    echo "feature hip already wired; copied files refreshed"
elif command -v git >/dev/null 2>&1; then
    git apply "$here/dusk-hades-0.24.1-hip.patch"
else
    patch -p1 < "$here/dusk-hades-0.24.1-hip.patch"
fi
echo "done.  First:  cargo test --test kat_scalar        (pure CPU: pins this library's vectors to the real crate)"
echo "       then:   HADES252_LIB_DIR=<dir of libhades252.so> LD_LIBRARY_PATH=\$HADES252_LIB_DIR cargo test --features hip"
