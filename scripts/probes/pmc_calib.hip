// What do rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of this library?  (round 5)
//
// The guide (MI355X_MICROARCH.md, HBM section) calibrates ONE pattern: a wide coalesced streaming read (16 B per lane) is reported at
// half its bytes.  scripts/pmc_summary.py doubled FETCH_SIZE for every kernel - the gather kernels (k_rescore, k_correct_fast,
// k_xr_score) included, whose "wasted traffic" figures then rest on an uncalibrated counter.  Every kernel below moves a KNOWN number
// of bytes from / to a buffer far beyond the 256 MB Infinity Cache; run it under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o p --output-format csv -- ./pmc_calib.bin
//     rocprofv3 --kernel-trace --pmc WRITE_SIZE ...
// and scripts/probes/pmc_calib.py divides the counters by the byte counts this program prints.
//
//   cal_stream<B>      every lane reads B = 4 / 8 / 16 bytes, lanes consecutive (streaming)
//   cal_gather<S>      a THREAD per random S-byte segment (S = 32 / 64 / 128, aligned to S), read as S / 16 loads of 16 B: k_rescore's and
//                      k_xr_score's pattern (a thread walks one target's words)
//   cal_gather_coop<S> S / 4 consecutive lanes read one random S-byte segment, 4 B each: the staging of k_correct_fast
//   cal_gather4        every lane one random 4-byte word (metadata look-ups by id)
//   cal_write<B>       streaming stores of B bytes per lane;  cal_scatter<S>: runs of S bytes at random places (the radix pass's stores)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) { x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x; }

template <int B> struct Vec;
template <> struct Vec<4> { typedef uint32_t T; };
template <> struct Vec<8> { typedef uint2 T; };
template <> struct Vec<16> { typedef uint4 T; };
__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <int B>
__global__ __launch_bounds__(256) void cal_stream(const char *buf, uint64_t bytes, uint32_t *sink) {
    typedef typename Vec<B>::T T;
    const T *p = reinterpret_cast<const T *>(buf);
    const uint64_t n = bytes / B;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) acc ^= fold(p[i]);
    if (acc == 0x12345u) sink[0] = acc;
}
template <int S>
__global__ __launch_bounds__(256) void cal_gather(const char *buf, uint64_t bytes, uint64_t nSeg, uint32_t *sink) {
    const uint64_t segs = bytes / S;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < nSeg; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint4 *p = reinterpret_cast<const uint4 *>(buf + (mix(i) % segs) * S);
#pragma unroll
        for (int j = 0; j < S / 16; j++) acc ^= fold(p[j]);
    }
    if (acc == 0x12345u) sink[0] = acc;
}
template <int S>
__global__ __launch_bounds__(256) void cal_gather_coop(const char *buf, uint64_t bytes, uint64_t nSeg, uint32_t *sink) {
    constexpr int LPS = S / 4;                  // lanes per segment
    const uint64_t segs = bytes / S;
    uint32_t acc = 0;
    const uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t i = t / LPS; i < nSeg; i += stride / LPS) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(buf + (mix(i) % segs) * S);
        acc ^= p[t % LPS];
    }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void cal_gather4(const char *buf, uint64_t bytes, uint64_t nWords, uint32_t *sink) {
    const uint64_t words = bytes / 4;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < nWords; i += (uint64_t) gridDim.x * blockDim.x)
        acc ^= reinterpret_cast<const uint32_t *>(buf)[mix(i) % words];
    if (acc == 0x12345u) sink[0] = acc;
}
template <int B>
__global__ __launch_bounds__(256) void cal_write(char *buf, uint64_t bytes) {
    typedef typename Vec<B>::T T;
    T *p = reinterpret_cast<T *>(buf);
    const uint64_t n = bytes / B;
    T v; memset(&v, 0x5A, sizeof(v));
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) p[i] = v;
}
// S / 8 consecutive lanes store one random S-byte run, 8 B each (a radix pass's runs of consecutive keys per digit)
template <int S>
__global__ __launch_bounds__(256) void cal_scatter(char *buf, uint64_t bytes, uint64_t nSeg) {
    constexpr int LPS = S / 8;
    const uint64_t segs = bytes / S;
    const uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t i = t / LPS; i < nSeg; i += stride / LPS)
        reinterpret_cast<uint64_t *>(buf + (mix(i) % segs) * S)[t % LPS] = i;
}

int main(int argc, char **argv) {
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 8ull) << 30;        // GiB of buffer
    const uint64_t nSeg = argc > 2 ? strtoull(argv[2], 0, 10) : (64ull << 20);        // segments per gather kernel
    char *buf; uint32_t *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("out of memory\n"); return 1; }
    hipMemset(buf, 1, bytes); hipMemset(sink, 0, 64);
    hipDeviceSynchronize();
    const dim3 g(256 * 32), b(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](const char *name, uint64_t moved, auto launch) {
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("CAL %s bytes %llu ms %.3f GBps %.1f\n", name, (unsigned long long) moved, ms, moved / 1e6 / ms); fflush(stdout);
    };
    timed("cal_stream<16>", bytes, [&] { hipLaunchKernelGGL(cal_stream<16>, g, b, 0, 0, buf, bytes, sink); });
    timed("cal_stream<8>", bytes, [&] { hipLaunchKernelGGL(cal_stream<8>, g, b, 0, 0, buf, bytes, sink); });
    timed("cal_stream<4>", bytes, [&] { hipLaunchKernelGGL(cal_stream<4>, g, b, 0, 0, buf, bytes, sink); });
    timed("cal_gather<32>", nSeg * 32, [&] { hipLaunchKernelGGL(cal_gather<32>, g, b, 0, 0, buf, bytes, nSeg, sink); });
    timed("cal_gather<64>", nSeg * 64, [&] { hipLaunchKernelGGL(cal_gather<64>, g, b, 0, 0, buf, bytes, nSeg, sink); });
    timed("cal_gather<128>", nSeg * 128, [&] { hipLaunchKernelGGL(cal_gather<128>, g, b, 0, 0, buf, bytes, nSeg, sink); });
    timed("cal_gather_coop<32>", nSeg * 32, [&] { hipLaunchKernelGGL(cal_gather_coop<32>, g, b, 0, 0, buf, bytes, nSeg, sink); });
    timed("cal_gather_coop<64>", nSeg * 64, [&] { hipLaunchKernelGGL(cal_gather_coop<64>, g, b, 0, 0, buf, bytes, nSeg, sink); });
    timed("cal_gather_coop<128>", nSeg * 128, [&] { hipLaunchKernelGGL(cal_gather_coop<128>, g, b, 0, 0, buf, bytes, nSeg, sink); });
    timed("cal_gather4", nSeg * 4 * 4, [&] { hipLaunchKernelGGL(cal_gather4, g, b, 0, 0, buf, bytes, nSeg * 4, sink); });
    timed("cal_write<16>", bytes, [&] { hipLaunchKernelGGL(cal_write<16>, g, b, 0, 0, buf, bytes); });
    timed("cal_write<8>", bytes, [&] { hipLaunchKernelGGL(cal_write<8>, g, b, 0, 0, buf, bytes); });
    timed("cal_write<4>", bytes, [&] { hipLaunchKernelGGL(cal_write<4>, g, b, 0, 0, buf, bytes); });
    timed("cal_scatter<64>", nSeg * 64, [&] { hipLaunchKernelGGL(cal_scatter<64>, g, b, 0, 0, buf, bytes, nSeg); });
    timed("cal_scatter<128>", nSeg * 128, [&] { hipLaunchKernelGGL(cal_scatter<128>, g, b, 0, 0, buf, bytes, nSeg); });
    timed("cal_scatter<256>", nSeg * 256, [&] { hipLaunchKernelGGL(cal_scatter<256>, g, b, 0, 0, buf, bytes, nSeg); });
    if (hipDeviceSynchronize() != hipSuccess) { printf("failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    return 0;
}
