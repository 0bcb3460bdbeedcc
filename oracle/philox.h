/* oracle/philox.h — TEST INFRASTRUCTURE (oracle).  Not part of the shipped library.
 *
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
 * SC'11), restated from the published algorithm.  The reference has no counter-based generator
 * (it uses libc rand()); this is the production-mode generator named by BASELINE.json's
 * north_star, and the oracle carries its own copy so that the HIP implementation is checked
 * against an independent CPU statement.  Known-answer vectors (Random123 kat_vectors) are pinned in
 * tests/test_oracle_rng.py.
 *
 * Draw convention shared by oracle and product ("philox mode"):
 *   draw k (0-based, per UE) of UE `ue` in a trial = philox4x32_10(ctr = {ue, k, nUE, variant},
 *   key = {seed_lo, seed_hi})[0] >> 1      — a 31-bit value, same range as glibc rand().
 */
#ifndef ORACLE_PHILOX_H
#define ORACLE_PHILOX_H
#include <stdint.h>

static inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline int philox_draw31(uint64_t seed, uint32_t nUE, uint32_t variant, uint32_t ue, uint32_t k) {
    uint32_t ctr[4] = {ue, k, nUE, variant};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t out[4];
    philox4x32_10(ctr, key, out);
    return (int)(out[0] >> 1);
}

#endif
