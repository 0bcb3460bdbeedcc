// prach_stream.hip — the reference's rand() stream, generated on the device.
//
// glibc's TYPE_3 generator is the lagged-Fibonacci recurrence r[i] = r[i-31] + r[i-3] (mod 2^32), rand() =
// r[344+k] >> 1 (stdlib/random_r.c).  It is linear, so the host jumps ahead with cached powers
// of the 31x31 companion matrix (prach_host.c: prach_internal_glibc_seeds) and hands ONE 31-word window per
// chunk of 63 488 outputs to the device; here one wavefront per chunk rolls the recurrence forward 31 values
// at a time: y[j] = w[j] + (j < 3 ? w[28+j] : y[j-3]) is three interleaved running sums, i.e. an inclusive
// scan along stride 3 (four __shfl_up steps) plus the broadcast tail.  A 100k-UE trial's ~45 M draws take
// ~0.1 ms instead of a 180 MB host-generated upload.
#include "prach_device.h"

namespace prach {

__global__ __launch_bounds__(256) void glibc_stream_kernel(const unsigned *__restrict__ seeds, int *__restrict__ out,
                                                           const unsigned long long n) {
    const int lane = threadIdx.x & 63;
    const unsigned long long c = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned long long start = c * (unsigned long long)STREAM_CHUNK;
    if (start >= n) return;
    unsigned w = lane < 31 ? seeds[c * 31 + lane] : 0u;
    const int tailsrc = 28 + lane % 3;
    for (int blk = 0; blk < STREAM_CHUNK / 31; blk++) {
        const unsigned long long k0 = start + (unsigned long long)blk * 31ull;
        if (k0 >= n) break;
        unsigned t = w, u;
        u = __shfl_up(t, 3);  if (lane >= 3) t += u;
        u = __shfl_up(t, 6);  if (lane >= 6) t += u;
        u = __shfl_up(t, 12); if (lane >= 12) t += u;
        u = __shfl_up(t, 24); if (lane >= 24) t += u;
        const unsigned y = t + __shfl(w, tailsrc);
        if (lane < 31 && k0 + lane < n) out[k0 + lane] = (int)(y >> 1);
        w = y;
    }
}

hipError_t launch_glibc_stream(const unsigned *seeds, int *out, unsigned long long n, hipStream_t stream) {
    const unsigned long long nchunks = (n + STREAM_CHUNK - 1) / STREAM_CHUNK;
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(glibc_stream_kernel, dim3((unsigned)((nchunks + 3) / 4)), dim3(256), 0, stream, seeds, out, n);
    return hipGetLastError();
}

} // namespace prach
