//! Optional companions of `HipStrategy` (`src/strategies/hip_extras.rs`, same `hip` feature): page-locked state
//! buffers and the callers of `perm` -- Merkle root and sponge hash -- on host memory.  Nothing here is needed by
//! `Strategy::perm` itself (`hip.rs`).  Never compiled in this image (no Rust toolchain);
//! `tests/test_rust_shim_signatures.py` ties the extern block to `include/hades252.h`.
//!
//! The crate is `#![no_std]` (src/lib.rs:8); `Vec` comes from `alloc`, which the `hip` feature may assume because a
//! shared library is linked anyway.  src/lib.rs gains `#[cfg(feature = "hip")] extern crate alloc;`.

use super::hip::{check, HipStrategy};
use crate::WIDTH;
use alloc::vec;
use alloc::vec::Vec;
use core::ffi::c_void;
use core::marker::PhantomData;
use core::ops::{Deref, DerefMut};
use dusk_bls12_381::BlsScalar;

extern "C" {
    fn hades252_host_alloc(out: *mut *mut c_void, bytes: usize) -> i32;
    fn hades252_host_free(p: *mut c_void) -> i32;
    fn hades252_host_register(p: *mut c_void, bytes: usize) -> i32;
    fn hades252_host_unregister(p: *mut c_void) -> i32;
    fn hades252_merkle_root(leaves: *const u64, n_leaves: usize, arity: i32, tag_mont: *const u64, out_idx: i32,
                            pad: *const u64, root: *mut u64) -> i32;
    fn hades252_merkle_root_multi(leaves: *const u64, n_leaves: usize, arity: i32, tag_mont: *const u64, out_idx: i32,
                                  n_workers: i32, flags: u32, root: *mut u64) -> i32;
    fn hades252_sponge_hash(msgs: *const u64, n_msgs: usize, msg_len: usize, capacity_mont: *const u64, pad_mode: i32,
                            digests: *mut u64) -> i32;
}

/// A batch of states in page-locked host memory (`hades252_host_alloc`): `perm` on it goes straight to DMA (93-98 % of
/// the host link's bidirectional ceiling from 2^22 states on).  Derefs to `[BlsScalar]`, like the `Vec` it replaces.
pub struct PinnedStates {
    ptr: *mut BlsScalar,
    len: usize,
}

impl PinnedStates {
    /// `n_states * WIDTH` scalars, all zero.
    pub fn new(n_states: usize) -> Self {
        let len = n_states * WIDTH;
        let mut p: *mut c_void = core::ptr::null_mut();
        check(unsafe { hades252_host_alloc(&mut p, len.max(1) * 32) });
        assert!(!p.is_null(), "hades252_host_alloc returned no memory");
        // all-zero limbs are BlsScalar::zero() in Montgomery form
        unsafe { core::ptr::write_bytes(p as *mut u8, 0, len * 32) };
        Self { ptr: p as *mut BlsScalar, len }
    }
}

impl Deref for PinnedStates {
    type Target = [BlsScalar];
    fn deref(&self) -> &[BlsScalar] {
        unsafe { core::slice::from_raw_parts(self.ptr, self.len) }
    }
}

impl DerefMut for PinnedStates {
    fn deref_mut(&mut self) -> &mut [BlsScalar] {
        unsafe { core::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}

impl Drop for PinnedStates {
    fn drop(&mut self) {
        unsafe { hades252_host_free(self.ptr as *mut c_void) };
    }
}

/// Page-locks an existing slice in place (`hades252_host_register`) for callers that cannot change where their states
/// are allocated.  The guard HOLDS the mutable borrow: the slice can be neither moved, grown nor freed while its pages
/// are registered, and is used through the guard (it derefs to `[BlsScalar]`).
pub struct PinGuard<'a> {
    ptr: *mut BlsScalar,
    len: usize,
    borrow: PhantomData<&'a mut [BlsScalar]>,
}

impl<'a> PinGuard<'a> {
    /// Registers `data` (no-op for an empty slice); the pages are unlocked when the guard is dropped.
    pub fn new(data: &'a mut [BlsScalar]) -> Self {
        if !data.is_empty() {
            check(unsafe { hades252_host_register(data.as_mut_ptr() as *mut c_void, data.len() * 32) });
        }
        Self { ptr: data.as_mut_ptr(), len: data.len(), borrow: PhantomData }
    }
}

impl<'a> Deref for PinGuard<'a> {
    type Target = [BlsScalar];
    fn deref(&self) -> &[BlsScalar] {
        unsafe { core::slice::from_raw_parts(self.ptr, self.len) }
    }
}

impl<'a> DerefMut for PinGuard<'a> {
    fn deref_mut(&mut self) -> &mut [BlsScalar] {
        unsafe { core::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}

impl<'a> Drop for PinGuard<'a> {
    fn drop(&mut self) {
        if self.len != 0 {
            unsafe { hades252_host_unregister(self.ptr as *mut c_void) };
        }
    }
}

fn limbs(s: &BlsScalar) -> *const u64 {
    s as *const BlsScalar as *const u64
}

impl HipStrategy {
    /// Root of the arity-`arity` tree over `leaves`, `parent = perm([tag, c_0 .., 0 ..])[out_idx]` (the node shape of
    /// dusk-poseidon's tree, README.md:9; `tag` and `out_idx` are that crate's convention, parameters here).
    /// `pad`: one digest per level for ragged trees, `None` = zeros.
    pub fn merkle_root(leaves: &[BlsScalar], arity: usize, tag: &BlsScalar, out_idx: usize, pad: Option<&[BlsScalar]>) -> BlsScalar {
        let mut root = BlsScalar::zero();
        let pad_ptr = pad.map_or(core::ptr::null(), |p| p.as_ptr() as *const u64);
        check(unsafe {
            hades252_merkle_root(leaves.as_ptr() as *const u64, leaves.len(), arity as i32, limbs(tag), out_idx as i32,
                                 pad_ptr, &mut root as *mut BlsScalar as *mut u64)
        });
        root
    }

    /// The same root for a FULL tree (`arity^k` leaves), its sub-trees sharded over `self.devices` GPUs (0 = all).
    pub fn merkle_root_sharded(&self, leaves: &[BlsScalar], arity: usize, tag: &BlsScalar, out_idx: usize) -> BlsScalar {
        let mut root = BlsScalar::zero();
        check(unsafe {
            hades252_merkle_root_multi(leaves.as_ptr() as *const u64, leaves.len(), arity as i32, limbs(tag),
                                       out_idx as i32, self.devices, 0, &mut root as *mut BlsScalar as *mut u64)
        });
        root
    }

    /// Sponge digests (rate 4) of `msgs.len() / msg_len` fixed-length messages; `pad_one` appends a single 1 first.
    pub fn sponge_hash(msgs: &[BlsScalar], msg_len: usize, capacity: &BlsScalar, pad_one: bool) -> Vec<BlsScalar> {
        assert!(msg_len > 0 && msgs.len() % msg_len == 0, "whole messages only");
        let mut out = vec![BlsScalar::zero(); msgs.len() / msg_len];
        check(unsafe {
            hades252_sponge_hash(msgs.as_ptr() as *const u64, out.len(), msg_len, limbs(capacity), pad_one as i32,
                                 out.as_mut_ptr() as *mut u64)
        });
        out
    }
}

#[cfg(test)]
mod tests {
    use super::*;
    use crate::{ScalarStrategy, Strategy};

    // Mirrors hades_det (src/strategies/scalar.rs:62-74) and the cross-check the reference runs between its two
    // strategies (src/strategies/gadget.rs:166-175).
    #[test]
    fn hip_matches_scalar() {
        let mut a = [BlsScalar::from(17u64); WIDTH];
        let (mut b, mut c) = (a, a);
        ScalarStrategy::new().perm(&mut a);
        // one state goes to the CPU by default (`MIN_GPU_STATES`); 0 forces the device
        HipStrategy { devices: 0, min_gpu_states: 0 }.perm(&mut b);
        HipStrategy::new().perm(&mut c);
        assert_eq!(a, b);
        assert_eq!(a, c);
    }

    // the size switch changes where a batch runs, never its result -- on both sides of every threshold
    #[test]
    fn hip_size_switch_is_invisible() {
        for n in [0usize, 1, 2, 3, 23, 24, 25] {
            let mut expect: Vec<BlsScalar> = (0..(WIDTH * n) as u64).map(|v| BlsScalar::from(v * v + 7)).collect();
            let (mut gpu, mut cpu, mut dflt) = (expect.clone(), expect.clone(), expect.clone());
            for chunk in expect.chunks_mut(WIDTH) {
                ScalarStrategy::new().perm(chunk);
            }
            HipStrategy { devices: 0, min_gpu_states: 0 }.perm(&mut gpu);
            HipStrategy { devices: 0, min_gpu_states: usize::MAX }.perm(&mut cpu);
            HipStrategy::new().perm(&mut dflt);
            assert_eq!(gpu, expect);
            assert_eq!(cpu, expect);
            assert_eq!(dflt, expect);
        }
    }

    #[test]
    fn hip_batch_matches_scalar_pinned_and_guarded() {
        let mut expect: Vec<BlsScalar> = (0..5 * 1000u64).map(BlsScalar::from).collect();
        let mut plain = expect.clone();
        let mut pinned = PinnedStates::new(1000);
        pinned.copy_from_slice(&expect);
        for chunk in expect.chunks_mut(WIDTH) {
            ScalarStrategy::new().perm(chunk);
        }
        HipStrategy::new().perm(&mut pinned);
        assert_eq!(&pinned[..], &expect[..]);
        {
            let mut guard = PinGuard::new(&mut plain[..]);
            HipStrategy::new().perm(&mut guard);
        }
        assert_eq!(plain, expect);
    }
}
