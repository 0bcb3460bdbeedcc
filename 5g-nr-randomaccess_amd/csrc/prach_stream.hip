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

// one wavefront: the chunk c of a window of n values
__device__ __forceinline__ void glibc_stream_chunk(const unsigned *__restrict__ seeds, int *__restrict__ out, const unsigned long long n, const unsigned long long c) {
    const int lane = threadIdx.x & 63;
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

__global__ __launch_bounds__(256) void glibc_stream_kernel(const unsigned *__restrict__ seeds, int *__restrict__ out,
                                                           const unsigned long long n) {
    glibc_stream_chunk(seeds, out, n, (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
}

// The windows of MANY trials in one launch (blockIdx.y = job): a chunk is a serial chain of 2048 steps (0.45 ms whatever the window's
// length), so a launch per trial costs 0.45 ms each — 100 trials of a sweep point: 45 ms of a 245 ms call — while all chunks of all
// windows side by side take about as long as one.
__global__ __launch_bounds__(256) void glibc_stream_jobs_kernel(const StreamJob *__restrict__ jobs) {
    const StreamJob J = jobs[blockIdx.y];
    glibc_stream_chunk(J.seeds, J.out, J.n, (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
}

hipError_t launch_glibc_stream(const unsigned *seeds, int *out, unsigned long long n, hipStream_t stream) {
    const unsigned long long nchunks = (n + STREAM_CHUNK - 1) / STREAM_CHUNK;
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(glibc_stream_kernel, dim3((unsigned)((nchunks + 3) / 4)), dim3(256), 0, stream, seeds, out, n);
    return hipGetLastError();
}

hipError_t launch_glibc_stream_jobs(const StreamJob *jobs, int njobs, unsigned long long max_n, hipStream_t stream) {
    const unsigned long long nchunks = (max_n + STREAM_CHUNK - 1) / STREAM_CHUNK;
    if (nchunks == 0 || njobs <= 0) return hipSuccess;
    if (njobs > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(glibc_stream_jobs_kernel, dim3((unsigned)((nchunks + 3) / 4), (unsigned)njobs), dim3(256), 0, stream, jobs);
    return hipGetLastError();
}

} // namespace prach
