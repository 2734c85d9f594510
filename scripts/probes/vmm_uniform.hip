// Probe: chunks of ONE size per reservation (two reservations filled alternately), a chunk in the middle given back and mapped again.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static hipMemAllocationProp prop;
static bool mapAt(char *va, size_t n, hipMemGenericAllocationHandle_t *out) {
    hipMemGenericAllocationHandle_t h;
    hipError_t e = hipMemCreate(&h, n, &prop, 0);
    if (e != hipSuccess) { printf("  create %zu MB: %s\n", n >> 20, hipGetErrorString(e)); return false; }
    e = hipMemMap(va, n, 0, h, 0);
    if (e != hipSuccess) { printf("  map %zu MB: %s\n", n >> 20, hipGetErrorString(e)); (void) hipMemRelease(h); return false; }
    hipMemAccessDesc d = {}; d.location = prop.location; d.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(va, n, &d, 1);
    if (e != hipSuccess) { printf("  set access %zu MB at %p: %s\n", n >> 20, (void *) va, hipGetErrorString(e)); (void) hipMemUnmap(va, n); (void) hipMemRelease(h); return false; }
    *out = h;
    return true;
}
int main() {
    (void) hipSetDevice(0);
    prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    const size_t MB = 1 << 20;
    void *a = nullptr, *b = nullptr;
    printf("reserve: %s %s\n", hipGetErrorString(hipMemAddressReserve(&a, (size_t) 16 << 30, 2 * MB, nullptr, 0)), hipGetErrorString(hipMemAddressReserve(&b, (size_t) 600 << 30, 2 * MB, nullptr, 0)));
    const size_t ca = 32 * MB, cb = 1024 * MB;
    std::vector<hipMemGenericAllocationHandle_t> ha(40), hb(20);
    int okA = 0, okB = 0;
    for (int i = 0; i < 40; i++) {
        okA += mapAt((char *) a + i * ca, ca, &ha[i]);
        if (i % 2 == 0) okB += mapAt((char *) b + (i / 2) * cb, cb, &hb[i / 2]);
    }
    printf("32 MB chunks: %d of 40, 1 GB chunks: %d of 20\n", okA, okB);
    (void) hipMemset(a, 1, 40 * ca); (void) hipMemset(b, 1, 20 * cb);
    printf("memset: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    // give chunk 7 of each back, map a new one there
    printf("unmap: %s %s\n", hipGetErrorString(hipMemUnmap((char *) a + 7 * ca, ca)), hipGetErrorString(hipMemUnmap((char *) b + 7 * cb, cb)));
    (void) hipMemRelease(ha[7]); (void) hipMemRelease(hb[7]);
    printf("again: %d %d\n", (int) mapAt((char *) a + 7 * ca, ca, &ha[7]), (int) mapAt((char *) b + 7 * cb, cb, &hb[7]));
    (void) hipMemset(a, 2, 40 * ca); (void) hipMemset(b, 2, 20 * cb);
    printf("memset: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    // two adjacent chunks in one range of calls?  one hipMemSetAccess over two mappings
    hipMemGenericAllocationHandle_t h1, h2;
    bool c1 = hipMemCreate(&h1, cb, &prop, 0) == hipSuccess && hipMemCreate(&h2, cb, &prop, 0) == hipSuccess;
    char *va = (char *) b + 20 * cb;
    hipError_t e1 = hipMemMap(va, cb, 0, h1, 0), e2 = hipMemMap(va + cb, cb, 0, h2, 0);
    hipMemAccessDesc d = {}; d.location = prop.location; d.flags = hipMemAccessFlagsProtReadWrite;
    printf("two mappings, one set-access call: create %d, map %s %s, set access %s\n", (int) c1, hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(hipMemSetAccess(va, 2 * cb, &d, 1)));
    // a different size in the 1 GB reservation
    hipMemGenericAllocationHandle_t hx;
    printf("512 MB chunk behind the 1 GB ones: %d\n", (int) mapAt((char *) b + 23 * cb, 512 * MB, &hx));
    printf("1 GB chunk behind that: %d\n", (int) mapAt((char *) b + 24 * cb, cb, &hx));
    return 0;
}
