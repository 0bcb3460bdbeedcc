/* oracle/oracle_internal.h — TEST INFRASTRUCTURE: the RNG handle shared by prach_oracle.c and noma_oracle.c */
#ifndef ORACLE_INTERNAL_H
#define ORACLE_INTERNAL_H
#include "glibc_rand.h"
#include <stdint.h>
struct oracle_rng {
    int mode;
    uint64_t seed;
    glibc_rand_t g;
    uint64_t consumed;
};
#endif
