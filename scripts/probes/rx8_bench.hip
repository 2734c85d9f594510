// What would sort 1's global passes cost on 8-byte tuples (keys only) instead of 12-byte (u64 key, u32 value) pairs?  (round 5)
// The library's own radix.h over n random 41-bit keys: pairs (today's layout) against keys only, per pass.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I carpedeam_amd/csrc scripts/probes/rx8_bench.hip -o scripts/probes/rx8_bench.bin
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include "radix.h"

void cdm_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
hipError_t cdmMallocRaw(void **p, size_t bytes) { return hipMalloc(p, bytes); }
void cdmFree(void *p) { hipFree(p); }
const char *cdmGetenv(const char *name) { return getenv(name); }

__global__ void fill(uint64_t *k, uint32_t *v, size_t n) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i < n) { uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; k[i] = x & ((1ull << 41) - 1); if (v) v[i] = (uint32_t) i; }
}
int main(int argc, char **argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 4100000000ull;
    uint64_t *k0, *k1; uint32_t *v0, *v1;
    if (hipMalloc(&k0, n * 8) || hipMalloc(&k1, n * 8) || hipMalloc(&v0, n * 4) || hipMalloc(&v1, n * 4)) { printf("out of memory\n"); return 1; }
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 2; rep++) {
        bool inFirst; float ms = 0;
        fill<<<(unsigned) ((n + 255) / 256), 256, 0, s>>>(k0, v0, n);
        if (rx::sortPairs<uint64_t, uint32_t>(s, 256, k0, k1, v0, v1, n, 14, 41, inFirst, &ms)) return 1;
        printf("pairs u64+u32, bits [14,41): %.1f ms for 3 passes = %.2f per pass, %.2f TB/s algorithmic\n", ms, ms / 3, n * 24.0 / (ms / 3) / 1e9); fflush(stdout);
        fill<<<(unsigned) ((n + 255) / 256), 256, 0, s>>>(k0, nullptr, n);
        if (rx::sortPairs<uint64_t, rx::NoValue>(s, 256, k0, k1, nullptr, nullptr, n, 14, 41, inFirst, &ms)) return 1;
        printf("keys u64 only, bits [14,41): %.1f ms for 3 passes = %.2f per pass, %.2f TB/s algorithmic\n", ms, ms / 3, n * 16.0 / (ms / 3) / 1e9); fflush(stdout);
    }
    return 0;
}
