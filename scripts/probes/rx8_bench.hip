// What do sort 1's global passes cost on 8-byte tuples instead of 12-byte (u64 key, u32 value) pairs, and is the slot-key sort
// (radix.h sortSlotKeys: head pass, segmented passes) the same permutation as the plain one?  (round 5)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I carpedeam_amd/csrc scripts/probes/rx8_bench.hip -o scripts/probes/rx8_bench.bin
//   rx8_bench.bin [n] [percent of empty slots]
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include "radix.h"

void cdm_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
hipError_t cdmMallocRaw(void **p, size_t bytes) { return hipMalloc(p, bytes); }
void cdmFree(void *p) { hipFree(p); }
const char *cdmGetenv(const char *name) { return getenv(name); }

__device__ __forceinline__ uint64_t mix(uint64_t i) { uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x; }
// slot keys: canonical-looking k-mers (the smaller of two random 40-bit values: crowds the low digits as real ones do), a strand bit, ~0 for
// `emptyPct` percent of the slots
__global__ void fill(uint64_t *k, uint32_t *v, size_t n, unsigned emptyPct) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t a = mix(i) & ((1ull << 40) - 1), b = mix(i + n) & ((1ull << 40) - 1), c = mix(i + 2 * n);
    const bool empty = (c % 100) < emptyPct;
    k[i] = empty ? ~0ull : ((a < b ? a : b) >> ((c >> 40) % 3 == 0 ? 12 : 0) << ((c >> 40) % 3 == 0 ? 12 : 0)) | ((c >> 8) & 1 ? 1ull << 63 : 0ull);      // (a third with 12 zero low bits: equal keys)
    if (v) v[i] = (uint32_t) i;
}
// plain result (key, value = slot) against slot tuples: same order?
__global__ void compare(const uint64_t *pk, const uint32_t *pv, const uint64_t *st, const unsigned long long *seg, uint64_t live, int shift, int lowBits, unsigned long long *bad) {
    const uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x;
    if (i >= live) return;
    int d = 0;
    for (int s = 256; s > 0; s >>= 1) if (d + s < rx::BINS && seg[d + s] <= i) d += s;
    const uint64_t t = st[i];
    const uint64_t kmer = ((uint64_t) d << shift) | (t >> rx::SLOT_KEY_SHIFT);
    const uint64_t want = pk[i];
    const bool ok = (kmer >> lowBits) == ((want & ((1ull << 40) - 1)) >> lowBits) && ((t >> 32) & 1) == (want >> 63) && (uint32_t) t == pv[i] && (kmer & ((1ull << lowBits) - 1)) == (want & ((1ull << lowBits) - 1));
    if (!ok) atomicAdd(bad, 1ull);
}
int main(int argc, char **argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 4100000000ull;
    const unsigned emptyPct = argc > 2 ? (unsigned) atoi(argv[2]) : 1;
    uint64_t *k0, *k1, *s0, *s1; uint32_t *v0, *v1; unsigned long long *seg, *bad;
    if (hipMalloc(&k0, n * 8) || hipMalloc(&k1, n * 8) || hipMalloc(&v0, n * 4) || hipMalloc(&v1, n * 4) || hipMalloc(&s0, n * 8) || hipMalloc(&s1, n * 8) || hipMalloc(&seg, (rx::BINS + 1) * 8) || hipMalloc(&bad, 8)) { printf("out of memory\n"); return 1; }
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 2; rep++) {
        bool inFirst; float ms = 0, launches = 0;
        fill<<<(unsigned) ((n + 255) / 256), 256, 0, s>>>(k0, v0, n, emptyPct);
        if (rx::sortPairs<uint64_t, uint32_t>(s, 256, k0, k1, v0, v1, n, 14, 41, inFirst, &ms)) return 1;
        printf("pairs u64+u32, bits [14,41): %.1f ms for 3 passes = %.2f per pass, %.2f TB/s algorithmic\n", ms, ms / 3, n * 24.0 / (ms / 3) / 1e9); fflush(stdout);
        const uint64_t *pk = inFirst ? k0 : k1; const uint32_t *pv = inFirst ? v0 : v1;
        fill<<<(unsigned) ((n + 255) / 256), 256, 0, s>>>(s0, nullptr, n, emptyPct);
        unsigned long long live = 0; uint64_t *res = nullptr;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, s);
        if (rx::sortSlotKeys(s, 256, s0, s1, n, 40, 14, nullptr, seg, live, res, &ms, &launches)) return 1;
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float all = 0; hipEventElapsedTime(&all, e0, e1);
        printf("slot keys: %llu of %llu slots live; %.1f ms for %d pass launches (%.2f each), %.1f ms with histograms and layout\n", live, (unsigned long long) n, ms, (int) launches, ms / launches, all); fflush(stdout);
        hipMemsetAsync(bad, 0, 8, s);
        if (live) compare<<<(unsigned) ((live + 255) / 256), 256, 0, s>>>(pk, pv, res, seg, live, 31, 14, bad);
        unsigned long long hb = 0; hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
        printf("slot tuples against the plain sort's pairs: %llu of %llu differ%s\n", hb, live, hb ? "  <-- WRONG" : ""); fflush(stdout);
        if (hipGetLastError() != hipSuccess) { printf("hip error\n"); return 1; }
    }
    return 0;
}
