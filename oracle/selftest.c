/* Sanitizer self-test of the CPU oracle (test infrastructure): built with
 * -fsanitize=address,undefined by tests/test_oracle_sanitized.py and run as a program. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void hades_oracle_perm(uint64_t *state);
void hades_oracle_perm_trace(uint64_t *state, uint64_t *trace);
void hades_oracle_perm_batch(uint64_t *states, size_t n, int n_threads);
void hades_oracle_gen_b(uint64_t *out, uint64_t first_elem, size_t n_elems, uint64_t seed);
void hades_oracle_gen_a(uint64_t *out, uint64_t first_elem, size_t n_elems);
void hades_oracle_merkle4_level(const uint64_t *children, uint64_t *parents, size_t n_parents,
                                const uint64_t *tag_mont, int out_idx, int n_threads);
int hades_oracle_from_bytes(const uint8_t *bytes, uint64_t *limbs);
void hades_oracle_to_bytes(const uint64_t *limbs, uint8_t *bytes);

int main(void) {
    /* perm([1;5])[0] in Montgomery form */
    const uint64_t one_mont[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL,
                                  0x1824b159acc5056fULL};
    const uint64_t expect0[4] = {0x935feb66a5e6cf3cULL, 0x2409c7dd1a61ab1cULL, 0x832c33cbf2dd481fULL,
                                 0x23338e018f505a2aULL};
    uint64_t st[20], tr[67 * 20];
    for (int w = 0; w < 5; w++) memcpy(st + 4 * w, one_mont, 32);
    hades_oracle_perm_trace(st, tr);
    if (memcmp(st, expect0, 32) != 0 || memcmp(tr + 66 * 20, st, 160) != 0) {
        printf("KAT mismatch\n");
        return 1;
    }
    size_t n = 3000;
    uint64_t *a = malloc(n * 160), *b = malloc(n * 160);
    hades_oracle_gen_b(a, 0, n * 5, 0x4861646573323532ULL);
    memcpy(b, a, n * 160);
    hades_oracle_perm_batch(a, n, 7);
    hades_oracle_perm_batch(b, n, 1);
    if (memcmp(a, b, n * 160) != 0) {
        printf("thread-count dependence\n");
        return 1;
    }
    size_t n_par = n * 5 / 4 / 4;
    uint64_t *par = malloc(n_par * 32);
    hades_oracle_merkle4_level(a, par, n_par, one_mont, 1, 3);
    uint8_t bytes[32];
    uint64_t back[4];
    hades_oracle_to_bytes(par, bytes);
    if (hades_oracle_from_bytes(bytes, back) != 0 || memcmp(back, par, 32) != 0) {
        printf("byte round trip\n");
        return 1;
    }
    hades_oracle_gen_a(b, 5, 10);
    free(a); free(b); free(par);
    printf("oracle selftest ok\n");
    return 0;
}
