// What does a process that holds ~140 GB of device memory cost when it ends - released chunk by chunk, or just left to the driver - and
// what does the NEXT process pay for getting the same memory right afterwards?  (round 5: the module processes of a workflow follow each
// other within milliseconds.)   exit_probe.bin <GB> <chunk MB> <release|leave> [vmm|malloc] [seconds to stay alive behind the release]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const double t00 = now();
    const size_t gb = argc > 1 ? atol(argv[1]) : 140, chunk = (size_t) (argc > 2 ? atol(argv[2]) : 256) << 20;
    const bool release = argc > 3 && !strcmp(argv[3], "release"), vmm = !(argc > 4 && !strcmp(argv[4], "malloc"));
    hipSetDevice(0); hipFree(0);
    const double t0 = now();
    const size_t total = gb << 30, n = total / chunk;
    std::vector<void *> ptrs; std::vector<hipMemGenericAllocationHandle_t> hs;
    char *base = nullptr;
    if (vmm) {
        if (hipMemAddressReserve((void **) &base, total, 2 << 20, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        hipMemAccessDesc d = {}; d.location = prop.location; d.flags = hipMemAccessFlagsProtReadWrite;
        for (size_t i = 0; i < n; i++) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess || hipMemMap(base + i * chunk, chunk, 0, h, 0) != hipSuccess || hipMemSetAccess(base + i * chunk, chunk, &d, 1) != hipSuccess) { printf("map %zu failed\n", i); return 1; }
            hs.push_back(h);
        }
    } else for (size_t i = 0; i < n; i++) { void *p; if (hipMalloc(&p, chunk) != hipSuccess) { printf("malloc %zu failed\n", i); return 1; } ptrs.push_back(p); }
    const double t1 = now();
    if (vmm) hipMemset(base, 1, total); else for (void *p : ptrs) hipMemset(p, 1, chunk);
    hipDeviceSynchronize();
    const double t2 = now();
    double t3 = t2;
    if (release) {
        if (vmm) { for (size_t i = 0; i < n; i++) { hipMemUnmap(base + i * chunk, chunk); hipMemRelease(hs[i]); } hipMemAddressFree(base, total); }
        else for (void *p : ptrs) hipFree(p);
        t3 = now();
    }
    if (argc > 5) { usleep((useconds_t) (atof(argv[5]) * 1e6)); }      // stay alive for a while behind the release
    printf("%zu GB in %zu chunks of %zu MB (%s): runtime up %.3f s, mapped %.3f s, touched %.3f s, %s %.3f s\n", gb, n, chunk >> 20, vmm ? "vmm" : "hipMalloc", t0 - t00, t1 - t0, t2 - t1, release ? "released" : "left to the driver", t3 - t2);
    fflush(stdout);
    _exit(0);
}
