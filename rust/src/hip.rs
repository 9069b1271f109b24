//! `HipStrategy`: the batched MI355X implementor of `Strategy<BlsScalar>` over the C ABI of `include/hades252.h`.
//! Goes to `src/strategies/hip.rs` of dusk-hades behind the cargo feature `hip`.  Never
//! compiled in this image (no Rust toolchain); `tests/test_rust_shim_signatures.py` ties the extern block to the header.

use super::{ScalarStrategy, Strategy};
use crate::WIDTH;
use dusk_bls12_381::BlsScalar;

#[link(name = "hades252")]
extern "C" {
    fn hades252_perm_batch(states: *mut u64, n_perms: usize) -> i32;
    fn hades252_perm_batch_multi(states: *mut u64, n_perms: usize, n_devices: i32) -> i32;
    fn hades252_strerror(code: i32) -> *const core::ffi::c_char;
}

// The pointer cast in `perm` relies on `BlsScalar` being 4 x u64 Montgomery limbs (`internal_repr()`,
// assets/HOWTO.md:45-47; `from_raw([u64; 4])`, src/round_constants.rs:41).
const _: () = assert!(core::mem::size_of::<BlsScalar>() == 32 && core::mem::align_of::<BlsScalar>() == 8);

/// Panics on a non-zero return code of libhades252 (the reference panics too, and release builds abort:
/// Cargo.toml:20), so nothing unwinds through the FFI.
pub(crate) fn check(rc: i32) {
    if rc != 0 {
        let msg = unsafe { core::ffi::CStr::from_ptr(hades252_strerror(rc)) };
        panic!("libhades252: {:?} ({})", msg, rc);
    }
}

/// States per call from which the GPU wins against THIS crate's serial CPU path: one `hades252_perm_batch` call costs
/// about 65 us whatever it carries up to 256 states, one `ScalarStrategy::perm` about 50 us on a host core (measured,
/// INTEGRATION.md "when NOT to route"), so a single permutation -- the reference's own call shape, README.md:60-61 --
/// stays on the CPU and two or more go to the device.  A caller with its own pool of T threads can raise it to ~1.3 T.
pub const MIN_GPU_STATES: usize = 2;

/// Batched Hades252 strategy on MI355X; stateless like `ScalarStrategy` (src/strategies/scalar.rs:11-13).
pub struct HipStrategy {
    /// 0 = the current device; n > 0 = shard host batches over the first n GPUs (no collective).
    pub devices: i32,
    /// Calls with fewer states than this run the reference's own `ScalarStrategy::perm` per state (never slower than
    /// not using this strategy); 0 = always the GPU.
    pub min_gpu_states: usize,
}

impl Default for HipStrategy {
    fn default() -> Self {
        Self { devices: 0, min_gpu_states: MIN_GPU_STATES }
    }
}

impl HipStrategy {
    /// Constructs a new `HipStrategy` (mirrors `ScalarStrategy::new`, src/strategies/scalar.rs:17-19).
    pub fn new() -> Self {
        Default::default()
    }
}

impl Strategy<BlsScalar> for HipStrategy {
    fn add_round_key<'b, I: Iterator<Item = &'b BlsScalar>>(&mut self, constants: &mut I, words: &mut [BlsScalar]) {
        ScalarStrategy::new().add_round_key(constants, words)
    }
    fn quintic_s_box(&mut self, value: &mut BlsScalar) {
        ScalarStrategy::new().quintic_s_box(value)
    }
    fn mul_matrix<'b, I: Iterator<Item = &'b BlsScalar>>(&mut self, constants: &mut I, values: &mut [BlsScalar]) {
        ScalarStrategy::new().mul_matrix(constants, values)
    }
    /// Overrides the provided `perm` (src/strategies.rs:140-157): every WIDTH-sized chunk of `data` is permuted in place,
    /// `min_gpu_states` or more by one GPU call, fewer by `ScalarStrategy` (bit-identical either way);
    /// `data.len() == WIDTH` is exactly `ScalarStrategy::perm`.
    fn perm(&mut self, data: &mut [BlsScalar]) {
        assert!(data.len() % WIDTH == 0, "Hades252 state length must be a multiple of WIDTH");
        let (p, n) = (data.as_mut_ptr() as *mut u64, data.len() / WIDTH);
        if n < self.min_gpu_states {
            return data.chunks_mut(WIDTH).for_each(|state| ScalarStrategy::new().perm(state));
        }
        check(unsafe { if self.devices > 0 { hades252_perm_batch_multi(p, n, self.devices) } else { hades252_perm_batch(p, n) } });
    }
}
