// Probe: does hipMemsetAsync / hipMemset fill a range beyond 4 GB completely?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fill(uint32_t *p, size_t n, uint32_t v) { for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = v; }
__global__ void k_count(const uint32_t *p, size_t n, uint32_t v, unsigned long long *out, unsigned long long *first) {
    unsigned long long c = 0;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) if (p[i] != v) { c++; atomicMin(first, (unsigned long long) i); }
    if (c) atomicAdd(out, c);
}
int main() {
    (void) hipSetDevice(0);
    const size_t words = (size_t) 2400 << 20;      // 9.4 GB
    uint32_t *p; unsigned long long *d, h[2];
    if (hipMalloc(&p, words * 4 + 4096) != hipSuccess || hipMalloc(&d, 16) != hipSuccess) { printf("no memory\n"); return 1; }
    for (int variant = 0; variant < 4; variant++) {
        for (size_t bytes : {(size_t) 3 << 30, ((size_t) 4 << 30) + 4096, words * 4 + 4, words * 4 - 12}) {
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, p, words + 1, 0xDEADBEEFu);
            h[0] = 0; h[1] = ~0ull; (void) hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
            hipError_t e;
            uint32_t want = 0;
            if (variant == 0) e = hipMemsetAsync(p, 0, bytes, 0);
            else if (variant == 1) e = hipMemset(p, 0, bytes);
            else if (variant == 2) { e = hipMemsetAsync(p, 0xA5, bytes, 0); want = 0xA5A5A5A5u; }
            else { e = hipMemsetD32Async((hipDeviceptr_t) p, 0, bytes / 4, 0); }
            hipLaunchKernelGGL(k_count, dim3(4096), dim3(256), 0, 0, p, bytes / 4, want, d, d + 1);
            (void) hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("variant %d (%s), %zu bytes: %s, %llu words not set, first at word %llu\n", variant, variant == 0 ? "hipMemsetAsync 0" : variant == 1 ? "hipMemset 0" : variant == 2 ? "hipMemsetAsync 0xA5" : "hipMemsetD32Async",
                   bytes, hipGetErrorString(e), h[0], h[0] ? h[1] : 0ull);
        }
    }
    return 0;
}
