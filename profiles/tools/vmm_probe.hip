// vmm_probe.hip — which forms of growing a reserved virtual range the HIP runtime of this image accepts (the engine's arena: prach_engine.hip grow_vmm_arena).
// Touches memory ONLY behind a hipMemSetAccess that returned hipSuccess.   hipcc --offload-arch=gfx950 -O2 vmm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
static bool ck(const char *what, hipError_t e) { printf("%-64s -> %s\n", what, hipGetErrorName(e)); fflush(stdout); (void)hipGetLastError(); return e == hipSuccess; }
#define CK(x) ck(#x, (x))
int main() {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    void *p = nullptr;
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    const size_t R = total_b; // (the engine reserves the device's whole memory size in address space)
    printf("reserve %zu bytes\n", R);
    if (!CK(hipMemAddressReserve(&p, R, 0, nullptr, 0))) return 1;
    char *base = (char *)p;
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemGenericAllocationHandle_t h1, h2, h3;
    const size_t A = (size_t)256 << 20, B = (size_t)1 << 30;
    if (!CK(hipMemCreate(&h1, A, &prop, 0)) || !CK(hipMemMap(base, A, 0, h1, 0))) return 1;
    const bool a1 = CK(hipMemSetAccess(base, A, &acc, 1));
    if (a1) CK(hipMemset(base, 1, A));
    if (!CK(hipMemCreate(&h2, B, &prop, 0)) || !CK(hipMemMap(base + A, B, 0, h2, 0))) return 1;
    bool a2 = CK(hipMemSetAccess(base + A, B, &acc, 1));          // the engine's form: the new piece alone
    if (!a2) a2 = CK(hipMemSetAccess(base, A + B, &acc, 1));     // the whole mapped range from the base
    if (a2) { CK(hipMemset(base + A, 1, B)); CK(hipDeviceSynchronize()); }
    if (!CK(hipMemCreate(&h3, B, &prop, 0)) || !CK(hipMemMap(base + A + B, B, 0, h3, 0))) return 1;
    bool a3 = CK(hipMemSetAccess(base + A + B, B, &acc, 1));
    if (!a3) a3 = CK(hipMemSetAccess(base, A + 2 * B, &acc, 1));
    if (a3) { CK(hipMemset(base + A + B, 1, B)); CK(hipDeviceSynchronize()); }
    return 0;
}
