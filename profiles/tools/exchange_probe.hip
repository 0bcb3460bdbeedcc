// profiles/tools/exchange_probe.hip — floor of the cluster's per-subframe exchange on this GPU: G workgroups (1024 threads, one per CU, all on
// one XCD through the packed launch of prach_lcluster.hip) publish NP + 1 self-validating 8-byte granules each and gather everybody's, ITER
// times, with nothing else to do.  Prints microseconds per round for write-through (sc1) and L2-resident (plain) publishes.
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/exchange_probe.hip -o /tmp/exchange_probe && /tmp/exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef long long __attribute__((address_space(1))) *gptr;
constexpr int G = 32, NP = 54, MBS = 64, ITER = 20000;
__device__ __forceinline__ long long mk(unsigned v, unsigned tag) { return (long long)(((unsigned long long)tag << 32) | v); }
template <bool PLAIN, bool PACK>
__global__ __launch_bounds__(1024) void probe(long long *mb_, unsigned long long *cycles, int *bad) {
    int b = blockIdx.x;
    if (PACK) { if (b & 7) return; b >>= 3; }
    if (b >= G) return;
    gptr mb = (gptr)mb_;
    const int tid = threadIdx.x;
    __shared__ int sum[64];
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned acc = 0;
    for (int it = 0; it < ITER; it++) {
        const unsigned tag = (unsigned)it + 1u;
        gptr par = mb + (size_t)(it & 1) * G * MBS;
        if (tid < NP + 1) { // publish own mailbox: header + NP buckets
            const long long v = mk((unsigned)(b * 1000 + tid), tag);
            if (PLAIN) __hip_atomic_store(par + b * MBS + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(par + b * MBS + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // gather: G * (NP + 1) granules over 1024 threads (two each)
        for (int k = tid; k < G * (NP + 1); k += 1024) {
            const int wg = k / (NP + 1), p = k - wg * (NP + 1);
            long long g = __hip_atomic_load(par + wg * MBS + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while ((unsigned)((unsigned long long)g >> 32) != tag) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { *bad = 1; break; }
                g = __hip_atomic_load(par + wg * MBS + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            acc += (unsigned)g;
        }
        __syncthreads();
    }
    if (tid < 64) sum[tid] = (int)acc;
    if (b == 0 && tid == 0) *cycles = __builtin_readcyclecounter() - t0;
}
template <bool PLAIN, bool PACK> static void run(const char *name) {
    long long *mb; unsigned long long *cyc; int *bad;
    CHECK(hipMalloc(&mb, sizeof(long long) * 2 * G * MBS)); CHECK(hipMemset(mb, 0, sizeof(long long) * 2 * G * MBS));
    CHECK(hipMalloc(&cyc, 8)); CHECK(hipMalloc(&bad, 4)); CHECK(hipMemset(bad, 0, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipMemset(mb, 0, sizeof(long long) * 2 * G * MBS));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<PLAIN, PACK>), dim3(PACK ? G * 8 : G), dim3(1024), 0, 0, mb, cyc, bad);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    int hb; CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    std::printf("%-44s %.3f us per round%s\n", name, 1e3 * ms / ITER, hb ? "  (TIMED OUT)" : "");
    CHECK(hipFree(mb)); CHECK(hipFree(cyc)); CHECK(hipFree(bad));
}
int main() {
    run<false, false>("sc1 publish, clusters across the XCDs");
    run<false, true>("sc1 publish, cluster packed on one XCD");
    run<true, true>("L2-resident publish, packed on one XCD");
    return 0;
}
