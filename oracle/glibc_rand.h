/* oracle/glibc_rand.h — TEST INFRASTRUCTURE (oracle).  Not part of the shipped library.
 *
 * Clone of glibc's TYPE_3 additive-feedback `rand()` / `srand()` (the generator the reference
 * programs consume through libc: RandomAccessSimulatorBeta.c:69 `srand(randomSeed)`, every
 * `rand()` call site listed in SURVEY.md §3.3).  glibc is a third-party dependency of the reference
 * that is present in this image (glibc 2.35); its published algorithm (stdlib/random_r.c,
 * __srandom_r / __random_r, degree 31, separation 3) is restated here so the oracle's draw stream
 * can be positioned / replayed explicitly.  tests/test_oracle_rng.py pins it against libc's own
 * rand() through ctypes (10^6 draws x several seeds).
 *
 *   state ring r[0..30]:  r[0] = seed (0 -> 1);  r[i] = 16807 * r[i-1] mod (2^31-1)  (Schrage)
 *   f = 3, b = 0;  each draw: r[f] += r[b]; out = r[f] >> 1; f,b advance mod 31
 *   the first 310 outputs are discarded by srandom.
 */
#ifndef ORACLE_GLIBC_RAND_H
#define ORACLE_GLIBC_RAND_H
#include <stdint.h>

typedef struct {
    uint32_t r[31];
    int f, b;
    uint64_t ndraws; /* draws handed out since seeding (excludes the 310 warm-up discards) */
} glibc_rand_t;

static inline uint32_t glibc_rand_raw_(glibc_rand_t *g) {
    g->r[g->f] += g->r[g->b];
    uint32_t out = g->r[g->f] >> 1;
    if (++g->f == 31) g->f = 0;
    if (++g->b == 31) g->b = 0;
    return out;
}

static inline void glibc_srand(glibc_rand_t *g, unsigned int seed) {
    int32_t word = (int32_t)(seed == 0 ? 1u : seed);
    g->r[0] = (uint32_t)word;
    for (int i = 1; i < 31; i++) {
        int32_t hi = word / 127773, lo = word % 127773;
        word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        g->r[i] = (uint32_t)word;
    }
    g->f = 3;
    g->b = 0;
    for (int i = 0; i < 310; i++) (void)glibc_rand_raw_(g);
    g->ndraws = 0;
}

/* next rand() value, 0 .. RAND_MAX (2^31-1) */
static inline int glibc_rand(glibc_rand_t *g) {
    g->ndraws++;
    return (int)glibc_rand_raw_(g);
}

#endif
