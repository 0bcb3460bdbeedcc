// profiles/tools/copy_probe.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths the streaming-regime
// kernel uses (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Reads a 1 GiB buffer once per kernel with (a) 8-byte non-temporal loads per lane (the 8 + 4 byte hot record),
// (b) 16-byte loads per lane, (c) 4-byte loads; writes one word per workgroup.  Known bytes: 1 GiB per kernel.
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/copy_probe.hip -o /tmp/copy_probe
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- /tmp/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define G __attribute__((address_space(1)))
template <class T> __global__ void read_kernel(const T *__restrict__ in, int *out, size_t n) {
    const G T *p = (const G T *)in;
    int acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = __builtin_nontemporal_load(p + i);
        const int *w = reinterpret_cast<const int *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x12345678) out[blockIdx.x] = acc; // (never true for the zero-filled buffer + pattern: keeps the loads alive)
}
__global__ void write_kernel(v2i *out, size_t n) { // 8-byte stores per lane
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { v2i v; v.x = (int)i; v.y = 1; out[i] = v; }
}
int main() {
    const size_t bytes = 1ull << 30;
    void *buf; int *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4096 * 4);
    hipMemset(buf, 1, bytes);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(read_kernel<v2i>, dim3(4096), dim3(256), 0, 0, (const v2i *)buf, out, bytes / 8);
        hipLaunchKernelGGL(read_kernel<v4i>, dim3(4096), dim3(256), 0, 0, (const v4i *)buf, out, bytes / 16);
        hipLaunchKernelGGL(read_kernel<int>, dim3(4096), dim3(256), 0, 0, (const int *)buf, out, bytes / 4);
        hipLaunchKernelGGL(write_kernel, dim3(4096), dim3(256), 0, 0, (v2i *)buf, bytes / 8);
    }
    hipDeviceSynchronize();
    printf("copy_probe: 3 x {read 8 B/lane, read 16 B/lane, read 4 B/lane, write 8 B/lane} over %zu bytes\n", bytes);
    return 0;
}
