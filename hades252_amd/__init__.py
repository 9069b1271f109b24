"""hades252_amd -- MI355X-native batched Hades252 permutation (host-side mirror of dusk-hades)."""
WIDTH = 5                 # reference src/lib.rs:27
TOTAL_FULL_ROUNDS = 8     # reference src/lib.rs:21
PARTIAL_ROUNDS = 59       # reference src/lib.rs:25
