// tests/tools/gpu_flat_equiv.hip — TEST INFRASTRUCTURE (GPU box): flat_equiv.hip's comparison with both forms compiled FOR THE DEVICE — the branch-free and the
// branched state machine run by every thread of a kernel on the same random cases (generated on the host by the same generator), field by field.
// usage: gpu_flat_equiv [cases = 4000000] [seed = 1]    exit code 0 = no difference.
#define FLAT_EQUIV_NO_MAIN
#include "flat_equiv.hip"
#include <vector>

__global__ void equiv_kernel(const Case *cases, long n, int *first_bad, Out *oa, Out *ob) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const Case c = cases[k];
    const Out a = run_branched(c), b = run_flat(c);
    if (!same(a, b)) { if (atomicCAS(first_bad, -1, (int)k) == -1) { *oa = a; *ob = b; } }
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 4000000;
    rng_s = 0x9E3779B97F4A7C15ull ^ (uint64_t)(argc > 2 ? atol(argv[2]) : 1);
    std::vector<Case> h((size_t)n);
    for (long k = 0; k < n; k++) gen(h[(size_t)k]);
    Case *d; int *bad; Out *oa, *ob;
    if (hipMalloc(&d, sizeof(Case) * (size_t)n) != hipSuccess || hipMalloc(&bad, 4) != hipSuccess || hipMalloc(&oa, sizeof(Out)) != hipSuccess || hipMalloc(&ob, sizeof(Out)) != hipSuccess) { printf("hipMalloc failed\n"); return 2; }
    hipMemcpy(d, h.data(), sizeof(Case) * (size_t)n, hipMemcpyHostToDevice);
    int m1 = -1; hipMemcpy(bad, &m1, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(equiv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, n, bad, oa, ob);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    int fb; hipMemcpy(&fb, bad, 4, hipMemcpyDeviceToHost);
    if (fb >= 0) {
        Out a, b; hipMemcpy(&a, oa, sizeof(Out), hipMemcpyDeviceToHost); hipMemcpy(&b, ob, sizeof(Out), hipMemcpyDeviceToHost);
        const Case &c = h[(size_t)fb];
        printf("case %d DIFFERS on the device: t=%d i=%d sdur=%d granted=%d tcu=%d d1=%d d2=%d | nP=%u backoff=%u aT=%d maxRar=%d maxMsg2=%d withnoma=%d | in: tx=%d tb=%d bo=%d act=%d conn=%d pre=%d rar=%d mrc=%d pend=%d\n",
               fb, c.t, c.i, c.sdur, c.granted, c.tcu, c.d1, c.d2, c.K.fmP.d, c.K.fmB.d, c.K.aT, c.K.maxRar, c.K.maxMsg2, (int)c.K.withnoma, c.u.tx, c.u.tb, c.u.bo, c.u.act, c.u.conn, c.u.pre, c.u.rar, c.u.mrc, c.u.pend);
        show("branched", a); show("flat", b);
        const Out ha = run_branched(c), hb = run_flat(c);
        show("host br.", ha); show("host flat", hb);
        return 1;
    }
    printf("gpu_flat_equiv: %ld cases on the device, 0 differences\n", n);
    return 0;
}
