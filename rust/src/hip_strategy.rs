//! `HipStrategy` -- batched, GPU-backed implementor of `dusk_hades::Strategy<BlsScalar>`.
//!
//! SOURCE ONLY: this image has no Rust toolchain, so this file has never been compiled here.
//! It is the binding a maintainer of dusk-hades would add; everything it
//! calls is exercised from C++ and Python through the same C ABI (`include/hades252.h`).
//!
//! Intended location in the reference tree: `src/strategies/hip.rs`, gated behind a cargo
//! feature `hip` (pattern of the existing `plonk` feature, Cargo.toml:25-26, src/lib.rs:29-30,
//! src/strategies.rs:20-21), because the crate is `#![no_std]` (src/lib.rs:8) and linking a
//! shared library needs `std`.
//!
//! Behaviour: `perm(&mut data)` permutes every WIDTH-sized chunk of `data` in place
//! (`data.len() == WIDTH` is exactly `ScalarStrategy::perm`); a length that is not a multiple
//! of WIDTH panics, like `copy_from_slice` does in the reference (src/strategies/scalar.rs:48);
//! any error from the library panics (release builds abort: Cargo.toml:20), nothing unwinds
//! through the FFI.

use super::{ScalarStrategy, Strategy};
use crate::WIDTH;
use dusk_bls12_381::BlsScalar;

#[link(name = "hades252")]
extern "C" {
    /// include/hades252.h: in place on host memory, `n_perms * 5 * 4` u64 Montgomery limbs.
    fn hades252_perm_batch(states: *mut u64, n_perms: usize) -> i32;
    /// include/hades252.h: shards the batch over the node's GPUs (0 = all), no collective.
    fn hades252_perm_batch_multi(states: *mut u64, n_perms: usize, n_devices: i32) -> i32;
    /// include/hades252.h: page-locked host memory -- pin a long-lived buffer ONCE instead of on every call.
    fn hades252_host_alloc(out: *mut *mut core::ffi::c_void, bytes: usize) -> i32;
    fn hades252_host_free(p: *mut core::ffi::c_void) -> i32;
    fn hades252_host_register(p: *mut core::ffi::c_void, bytes: usize) -> i32;
    fn hades252_host_unregister(p: *mut core::ffi::c_void) -> i32;
    /// include/hades252.h: the callers of `perm` on host memory -- leaves / messages in, 32 bytes per tree / message out.
    fn hades252_merkle_root(leaves: *const u64, n_leaves: usize, arity: i32, tag_mont: *const u64, out_idx: i32,
                            pad: *const u64, root: *mut u64) -> i32;
    fn hades252_merkle_root_multi(leaves: *const u64, n_leaves: usize, arity: i32, tag_mont: *const u64, out_idx: i32,
                                  n_workers: i32, flags: u32, root: *mut u64) -> i32;
    fn hades252_sponge_hash(msgs: *const u64, n_msgs: usize, msg_len: usize, capacity_mont: *const u64, pad_mode: i32,
                            digests: *mut u64) -> i32;
    fn hades252_strerror(code: i32) -> *const core::ffi::c_char;
}

/// A batch of states in page-locked host memory (`hades252_host_alloc`): `perm` on it goes straight to DMA
/// (measured from a native caller: 277 / 290 M permutations/s at 2^22 / 2^24 states = 93 / 98 % of the host
/// link's bidirectional ceiling, against 83-170 M/s when the library has to page-lock an ordinary slice inside
/// every call).  Derefs to `[BlsScalar]`, so it is used like the `Vec<BlsScalar>` it replaces.
pub struct PinnedStates {
    ptr: *mut BlsScalar,
    len: usize,
}

impl PinnedStates {
    /// `n_states * WIDTH` scalars, zero-initialised.
    pub fn new(n_states: usize) -> Self {
        let len = n_states * WIDTH;
        let mut p: *mut core::ffi::c_void = core::ptr::null_mut();
        HipStrategy::check(unsafe { hades252_host_alloc(&mut p, len.max(1) * 32) });
        unsafe { core::ptr::write_bytes(p as *mut u8, 0, len * 32) };
        Self { ptr: p as *mut BlsScalar, len }
    }
}

impl core::ops::Deref for PinnedStates {
    type Target = [BlsScalar];
    fn deref(&self) -> &[BlsScalar] {
        unsafe { core::slice::from_raw_parts(self.ptr, self.len) }
    }
}

impl core::ops::DerefMut for PinnedStates {
    fn deref_mut(&mut self) -> &mut [BlsScalar] {
        unsafe { core::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}

impl Drop for PinnedStates {
    fn drop(&mut self) {
        unsafe { hades252_host_free(self.ptr as *mut core::ffi::c_void) };
    }
}

/// Page-locks an existing slice in place for as long as the guard lives (`hades252_host_register`): for callers
/// that cannot change where their states are allocated.
pub struct PinGuard(*mut core::ffi::c_void);

impl PinGuard {
    pub fn new(data: &mut [BlsScalar]) -> Self {
        let p = data.as_mut_ptr() as *mut core::ffi::c_void;
        HipStrategy::check(unsafe { hades252_host_register(p, data.len() * 32) });
        Self(p)
    }
}

impl Drop for PinGuard {
    fn drop(&mut self) {
        unsafe { hades252_host_unregister(self.0) };
    }
}

/// Batched Hades252 strategy on MI355X.  Stateless like `ScalarStrategy` (src/strategies/scalar.rs:11-13) --
/// the two fields are plain configuration, nothing is carried from one `perm` to the next; the library is
/// re-entrant, so strategies on different threads may run concurrently.
#[derive(Default)]
pub struct HipStrategy {
    /// 0 = current device only; n > 0 = shard host batches over the first n GPUs.
    pub devices: i32,
    /// Batches of fewer permutations than this stay on the reference's own `ScalarStrategy`
    /// (0 = always use the GPU, the default).  Measured on MI355X: one GPU call costs ~65 us for up to
    /// 768 states (50 us of it the kernel), one CPU permutation ~52 us: the GPU wins from two states
    /// on.  (libhades252 itself has no CPU path.)
    pub cpu_below: usize,
}

impl HipStrategy {
    /// Constructs a new `HipStrategy` (mirrors `ScalarStrategy::new`, scalar.rs:17-19).
    pub fn new() -> Self {
        Default::default()
    }

    /// Root of the arity-`arity` tree over `leaves` (`parent = perm([tag, c_0 .., 0 ..])[out_idx]`, the node shape of
    /// dusk-poseidon's tree, README.md:9; `tag` and `out_idx` are that crate's convention and a parameter here).  The
    /// leaves travel to the GPU in chunks behind the hashing of the first level: 2^24 leaves in ~15 ms (13.3 ms
    /// on resident data).  `pad`: one digest per level for ragged trees, or `None` for zeros.
    pub fn merkle_root(leaves: &[BlsScalar], arity: usize, tag: &BlsScalar, out_idx: usize, pad: Option<&[BlsScalar]>) -> BlsScalar {
        let mut root = BlsScalar::zero();
        Self::check(unsafe {
            hades252_merkle_root(leaves.as_ptr() as *const u64, leaves.len(), arity as i32, tag as *const BlsScalar as *const u64,
                                 out_idx as i32, pad.map_or(core::ptr::null(), |p| p.as_ptr() as *const u64),
                                 &mut root as *mut BlsScalar as *mut u64)
        });
        root
    }

    /// The same root for a FULL tree (`arity^k` leaves) with the sub-trees sharded over `self.devices` GPUs (0 = all).
    pub fn merkle_root_sharded(&self, leaves: &[BlsScalar], arity: usize, tag: &BlsScalar, out_idx: usize) -> BlsScalar {
        let mut root = BlsScalar::zero();
        Self::check(unsafe {
            hades252_merkle_root_multi(leaves.as_ptr() as *const u64, leaves.len(), arity as i32,
                                       tag as *const BlsScalar as *const u64, out_idx as i32, self.devices, 0,
                                       &mut root as *mut BlsScalar as *mut u64)
        });
        root
    }

    /// Sponge digests (rate 4) of `msgs.len() / msg_len` fixed-length messages; `pad_one`: append a single 1 first.
    pub fn sponge_hash(msgs: &[BlsScalar], msg_len: usize, capacity: &BlsScalar, pad_one: bool) -> Vec<BlsScalar> {
        assert!(msg_len > 0 && msgs.len() % msg_len == 0, "whole messages only");
        let mut out = vec![BlsScalar::zero(); msgs.len() / msg_len];
        Self::check(unsafe {
            hades252_sponge_hash(msgs.as_ptr() as *const u64, out.len(), msg_len, capacity as *const BlsScalar as *const u64,
                                 pad_one as i32, out.as_mut_ptr() as *mut u64)
        });
        out
    }

    fn check(rc: i32) {
        if rc != 0 {
            let msg = unsafe { core::ffi::CStr::from_ptr(hades252_strerror(rc)) };
            panic!("libhades252: {:?} ({})", msg, rc);
        }
    }
}

// `BlsScalar` is `#[repr(transparent)]`-like over `[u64; 4]` (4 x u64 little-endian Montgomery
// limbs: `internal_repr()`, assets/HOWTO.md:45-47; `from_raw([u64; 4])`, round_constants.rs:41).
// The cast below relies on that layout; the `_bytes` entry point of the library is the
// layout-independent alternative (`to_bytes()` / `from_bytes()`).
const _: () = assert!(core::mem::size_of::<BlsScalar>() == 32);

impl Strategy<BlsScalar> for HipStrategy {
    // The three per-operation methods are never on the batched path; they keep the reference's
    // semantics by delegating to the scalar implementation (src/strategies/scalar.rs:23-49).
    fn add_round_key<'b, I>(&mut self, constants: &mut I, words: &mut [BlsScalar])
    where
        I: Iterator<Item = &'b BlsScalar>,
    {
        ScalarStrategy::new().add_round_key(constants, words)
    }

    fn quintic_s_box(&mut self, value: &mut BlsScalar) {
        ScalarStrategy::new().quintic_s_box(value)
    }

    fn mul_matrix<'b, I>(&mut self, constants: &mut I, values: &mut [BlsScalar])
    where
        I: Iterator<Item = &'b BlsScalar>,
    {
        ScalarStrategy::new().mul_matrix(constants, values)
    }

    /// Overrides the provided `perm` (src/strategies.rs:140-157): all 67 rounds of every state
    /// run in one GPU kernel.
    fn perm(&mut self, data: &mut [BlsScalar]) {
        assert!(
            data.len() % WIDTH == 0,
            "Hades252 state length must be a multiple of WIDTH"
        );
        let n_perms = data.len() / WIDTH;
        if n_perms < self.cpu_below {
            for chunk in data.chunks_mut(WIDTH) {
                ScalarStrategy::new().perm(chunk);
            }
            return;
        }
        let ptr = data.as_mut_ptr() as *mut u64;
        let rc = unsafe {
            if self.devices > 0 {
                hades252_perm_batch_multi(ptr, n_perms, self.devices)
            } else {
                hades252_perm_batch(ptr, n_perms)
            }
        };
        Self::check(rc);
    }
}

#[cfg(test)]
mod tests {
    use super::*;

    // Mirrors hades_det (src/strategies/scalar.rs:62-74) and adds the cross-check the
    // reference uses between its two strategies (src/strategies/gadget.rs:166-175).
    #[test]
    fn hip_matches_scalar() {
        let mut a = [BlsScalar::from(17u64); WIDTH];
        let mut b = a;
        ScalarStrategy::new().perm(&mut a);
        HipStrategy::new().perm(&mut b);
        assert_eq!(a, b);
    }

    #[test]
    fn hip_batch_matches_scalar() {
        let mut batch: Vec<BlsScalar> = (0..5 * 1000u64).map(BlsScalar::from).collect();
        let mut expect = batch.clone();
        for chunk in expect.chunks_mut(WIDTH) {
            ScalarStrategy::new().perm(chunk);
        }
        HipStrategy::new().perm(&mut batch);
        assert_eq!(batch, expect);
    }
}
