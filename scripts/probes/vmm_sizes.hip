// Probe: which chunk sizes / offsets hipMemCreate + hipMemMap + hipMemSetAccess take (chunks appended at the running end of one reservation).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
int main() {
    hipSetDevice(0);
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum); hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
    printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
    const size_t MB = 1 << 20;
    for (size_t align : {2 * MB, (size_t) 1 << 30}) {
        void *base = nullptr;
        hipError_t e = hipMemAddressReserve(&base, (size_t) 64 << 30, align, nullptr, 0);
        printf("reserve 64 GB, alignment %zu MB: %s, base %p\n", align / MB, hipGetErrorString(e), base);
        size_t end = 0;
        for (size_t mb : {2, 6, 22, 288, 22, 100, 1026, 2, 4096, 22, 3, 64, 22}) {
            const size_t n = mb * MB;
            hipMemGenericAllocationHandle_t h;
            e = hipMemCreate(&h, n, &prop, 0);
            if (e != hipSuccess) { printf("  create %zu MB: %s\n", mb, hipGetErrorString(e)); continue; }
            e = hipMemMap((char *) base + end, n, 0, h, 0);
            if (e != hipSuccess) { printf("  map %zu MB at %zu MB: %s\n", mb, end / MB, hipGetErrorString(e)); hipMemRelease(h); (void) hipGetLastError(); continue; }
            hipMemAccessDesc d = {}; d.location = prop.location; d.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess((char *) base + end, n, &d, 1);
            printf("  %zu MB at %zu MB: %s\n", mb, end / MB, e == hipSuccess ? "ok" : hipGetErrorString(e));
            if (e != hipSuccess) { hipMemUnmap((char *) base + end, n); hipMemRelease(h); (void) hipGetLastError(); continue; }
            hipMemset((char *) base + end, 1, n);
            e = hipDeviceSynchronize();
            if (e != hipSuccess) printf("    memset: %s\n", hipGetErrorString(e));
            end += n;
        }
    }
    return 0;
}
