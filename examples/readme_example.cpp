// The reference's README example (README.md:50-65) against the C++ mirror:
//   let mut strategy = ScalarStrategy::new();
//   strategy.perm(&mut state);
// plus hades_det (src/strategies/scalar.rs:62-74).  Needs a GPU at run time.
//   g++ -std=c++17 -I include examples/readme_example.cpp -L hades252_amd/csrc -lhades252 -Wl,-rpath,$PWD/hades252_amd/csrc
#include <cstdio>
#include <cstring>

#include "hades252.hpp"

using namespace dusk_hades;

int main(int argc, char **argv) {
    (void)argc; (void)argv;
    // values 17 and 19 in Montgomery form: v * 2^256 mod p
    const BlsScalar s17 = {{0x00000024ffffffdbull, 0xe5974b91003cb425ull, 0x98a3c6d69b9bc73aull, 0x3ea6d0fafc3ce490ull}};
    const BlsScalar s19 = {{0x00000028ffffffd7ull, 0x96a0bb8500434429ull, 0xcbbc66b675146725ull, 0x6ef033ae55c6ef6full}};
    // perm([17;5])[0] in Montgomery form (tests/golden/kat.json)
    const BlsScalar expect0 = {{0x0a9d33d0bb50f2a6ull, 0x48f157ad50c3e9dbull, 0x66a2f16c53ec6d12ull, 0x253ba342efa3cd20ull}};
    BlsScalar x[WIDTH], y[WIDTH], z[WIDTH];
    for (std::size_t i = 0; i < WIDTH; i++) x[i] = s17, y[i] = s17, z[i] = s19;
    ScalarStrategy strategy = ScalarStrategy::new_();
    strategy.perm(x, WIDTH);
    strategy.perm(y, WIDTH);
    strategy.perm(z, WIDTH);
    bool det = std::memcmp(x, y, sizeof x) == 0, diff = std::memcmp(x, z, sizeof x) != 0;
    bool kat = std::memcmp(&x[0], &expect0, sizeof expect0) == 0;
    std::printf("known answer perm([17;5])[0]: %s\n", kat ? "ok" : "MISMATCH");
    std::printf("hades_det: x==y %d, x!=z %d\n", det, diff);
    std::printf("perm([17;5])[0] limbs: %016llx %016llx %016llx %016llx\n", (unsigned long long)x[0].limbs[0],
                (unsigned long long)x[0].limbs[1], (unsigned long long)x[0].limbs[2], (unsigned long long)x[0].limbs[3]);
    std::printf("rounds() = %zu\n", ScalarStrategy::rounds());
    try {
        BlsScalar bad[4] = {};
        strategy.perm(bad, 4);
        return 2;
    } catch (const HadesPanic &) {
        std::printf("len != k*WIDTH rejected\n");
    }
    // the callers on host memory and the device-memory helpers, checked against each other: a 2-leaf tree of arity 2 is
    // ONE permutation of [tag, leaf0, leaf1, 0, 0] -- its word 1 is the root; the same state permuted in a DeviceBuffer
    const BlsScalar leaves[2] = {s17, s19};
    const BlsScalar root = merkle_root(leaves, 2, 2, /*tag=*/s19);
    BlsScalar st[WIDTH] = {s19, s17, s19, BlsScalar{}, BlsScalar{}}, back[WIDTH];
    DeviceBuffer dev(sizeof st);
    dev.upload(st, sizeof st);
    check(hades252_perm_batch_dev(dev.ptr(), 1, nullptr), "perm_batch_dev");
    dev.download(back, sizeof back);
    strategy.perm(st, WIDTH);
    const bool callers = std::memcmp(&root, &st[1], sizeof root) == 0 && std::memcmp(back, st, sizeof st) == 0;
    BlsScalar dig{};
    sponge_hash(leaves, 1, 2, /*capacity=*/s17, /*pad_one=*/false, &dig);      // one block: perm([cap, m0, m1, 0, 0])[1]
    BlsScalar sp[WIDTH] = {s17, s17, s19, BlsScalar{}, BlsScalar{}};
    strategy.perm(sp, WIDTH);
    const bool sponge = std::memcmp(&dig, &sp[1], sizeof dig) == 0;
    std::printf("merkle_root / DeviceBuffer / sponge_hash against perm: %s\n", callers && sponge ? "ok" : "MISMATCH");
    return det && diff && kat && callers && sponge ? 0 : 1;
}
