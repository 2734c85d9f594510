// Probe: what device memory costs on this box - hipMalloc (fresh, and again after hipFree) against the virtual-memory API
// (hipMemAddressReserve + hipMemCreate / hipMemMap / hipMemSetAccess), and whether a kernel runs as fast on mapped memory.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/vmm_probe.hip -o /tmp/vmm_probe && /tmp/vmm_probe [GB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k_copy(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) b[i] = a[i];
}
static void bandwidth(const char *what, void *a, void *b, size_t bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *) a, (uint4 *) b, bytes / 16);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %s: copy of %.1f GB, pass %d: %.2f ms = %.2f TB/s (read + write)\n", what, bytes / 1e9, rep, ms, 2.0 * bytes / ms / 1e9);
    }
}
int main(int argc, char **argv) {
    const size_t GB = 1ull << 30;
    const size_t total = (argc > 1 ? (size_t) atol(argv[1]) : 64) * GB;
    CK(hipSetDevice(0));
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot)); printf("device memory: %.1f GB free of %.1f\n", fr / 1e9, tot / 1e9);
    for (int round = 0; round < 3; round++) {
        void *p[2]; double t0 = now();
        CK(hipMalloc(&p[0], total / 2)); CK(hipMalloc(&p[1], total / 2));
        double t1 = now();
        printf("round %d: hipMalloc of 2 x %.0f GB: %.3f s (%.1f ms/GB)\n", round, total / 2e9, t1 - t0, 1e3 * (t1 - t0) / (total / 1e9));
        bandwidth("hipMalloc", p[0], p[1], total / 2);
        t0 = now(); CK(hipFree(p[0])); CK(hipFree(p[1])); t1 = now();
        printf("round %d: hipFree: %.3f s\n", round, t1 - t0);
    }
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("allocation granularity (recommended): %zu bytes\n", gran);
    for (size_t chunk : {GB, 8 * GB}) {
        for (int round = 0; round < 2; round++) {
            void *base = nullptr;
            double t0 = now();
            CK(hipMemAddressReserve(&base, 4 * total, 2u << 20, nullptr, 0));
            double tRes = now();
            std::vector<hipMemGenericAllocationHandle_t> hs;
            double tc = 0, tm = 0, ta = 0;
            for (size_t off = 0; off < total; off += chunk) {
                hipMemGenericAllocationHandle_t h; double a = now();
                CK(hipMemCreate(&h, chunk, &prop, 0)); double b = now();
                CK(hipMemMap((char *) base + off, chunk, 0, h, 0)); double c = now();
                hipMemAccessDesc d = {}; d.location = prop.location; d.flags = hipMemAccessFlagsProtReadWrite;
                CK(hipMemSetAccess((char *) base + off, chunk, &d, 1)); double e = now();
                tc += b - a; tm += c - b; ta += e - c; hs.push_back(h);
            }
            printf("VMM, chunks of %zu GB, round %d: reserve %.4f s, create %.3f s, map %.3f s, set access %.3f s for %.0f GB (%.1f ms/GB)\n", chunk / GB, round, tRes - t0, tc, tm, ta, total / 1e9,
                   1e3 * (tc + tm + ta) / (total / 1e9));
            bandwidth("VMM", base, (char *) base + total / 2, total / 2);
            t0 = now();
            for (size_t i = 0; i < hs.size(); i++) { CK(hipMemUnmap((char *) base + i * chunk, chunk)); CK(hipMemRelease(hs[i])); }
            CK(hipMemAddressFree(base, 4 * total));
            printf("VMM: unmap + release + free: %.3f s\n", now() - t0);
        }
    }
    return 0;
}
