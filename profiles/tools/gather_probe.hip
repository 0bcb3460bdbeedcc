// gather_probe.hip — what the memory system of an MI355X sustains for the batch kernel's event pattern: every lane reads one random 32-byte record of a
// table (two 16-byte loads), changes it and writes it back; 64 lanes per wavefront-instruction touch 64 different 128-byte lines.  Prints records / s for
// several table sizes (L2 / Infinity Cache / HBM resident), batches in flight per wavefront and wavefronts per CU.   hipcc --offload-arch=gfx950 -O3 gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int DEPTH, bool WRITE>
__global__ __launch_bounds__(256) void probe(v4i *__restrict__ tab, const unsigned long long nrec, const int iters, unsigned *sink) {
    unsigned long long x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull;
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        unsigned long long idx[DEPTH];
        v4i a[DEPTH], b[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) { x = x * 6364136223846793005ull + 1442695040888963407ull; idx[d] = (x >> 20) % nrec; }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) { a[d] = tab[2 * idx[d]]; b[d] = tab[2 * idx[d] + 1]; }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            a[d].x += 1; b[d].w ^= a[d].y;
            acc += (unsigned)a[d].x;
            if (WRITE) { tab[2 * idx[d]] = a[d]; tab[2 * idx[d] + 1] = b[d]; }
        }
    }
    if (acc == 0xdeadbeef) *sink = acc;
}
template <int DEPTH, bool WRITE> double run(v4i *tab, unsigned long long nrec, int blocks, int iters, unsigned *sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<DEPTH, WRITE>), dim3(blocks), dim3(256), 0, 0, tab, nrec, iters / 8, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<DEPTH, WRITE>), dim3(blocks), dim3(256), 0, 0, tab, nrec, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 256.0 * iters * DEPTH / (ms * 1e-3);
}
int main() {
    const unsigned long long maxb = 4ull << 30;
    v4i *tab; unsigned *sink;
    if (hipMalloc(&tab, maxb) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(tab, 0, maxb);
    const unsigned long long sizes[] = {16ull << 20, 128ull << 20, 1ull << 30, 4ull << 30};
    for (unsigned long long sz : sizes) {
        const unsigned long long nrec = sz / 32;
        for (int wpc : {4, 8, 16}) { // wavefronts per CU (256-thread blocks: 4 wavefronts each)
            const int blocks = 256 * wpc / 4;
            printf("table %5llu MB, %2d waves/CU: read-only depth1 %.3e depth4 %.3e | read+write depth1 %.3e depth2 %.3e depth4 %.3e records/s\n", sz >> 20, wpc,
                   run<1, false>(tab, nrec, blocks, 2000, sink), run<4, false>(tab, nrec, blocks, 500, sink),
                   run<1, true>(tab, nrec, blocks, 2000, sink), run<2, true>(tab, nrec, blocks, 1000, sink), run<4, true>(tab, nrec, blocks, 500, sink));
        }
    }
    return 0;
}
