#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cstdint>
template <int BITS, int BS, int IPT> using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, BITS, rocprim::block_radix_rank_algorithm::match>>;
__global__ void fill(uint64_t *k, uint32_t *v, size_t n) { size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; if (i < n) { uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; k[i] = x & ((1ull << 41) - 1); v[i] = (uint32_t) i; } }
template <typename C> float run(uint64_t *k0, uint64_t *k1, uint32_t *v0, uint32_t *v1, size_t n, int b0, int b1, const char *name) {
    rocprim::double_buffer<uint64_t> kb(k0, k1); rocprim::double_buffer<uint32_t> vb(v0, v1);
    size_t tb = 0; void *tmp = nullptr;
    if (rocprim::radix_sort_pairs<C>(nullptr, tb, kb, vb, n, b0, b1, 0) != hipSuccess) { printf("%s: size query failed\n", name); return -1; }
    hipMalloc(&tmp, tb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 3; it++) {
        hipEventRecord(e0, 0);
        hipError_t e = rocprim::radix_sort_pairs<C>(tmp, tb, kb, vb, n, b0, b1, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        if (e != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: sort failed\n", name); return -1; }
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%s bits [%d,%d): %.1f ms\n", name, b0, b1, best); fflush(stdout);
    hipFree(tmp); return best;
}
template <typename C> float runk(uint64_t *k0, uint64_t *k1, size_t n, int b0, int b1, const char *name) {
    rocprim::double_buffer<uint64_t> kb(k0, k1);
    size_t tb = 0; void *tmp = nullptr;
    if (rocprim::radix_sort_keys<C>(nullptr, tb, kb, n, b0, b1, 0) != hipSuccess) { printf("%s: size query failed\n", name); return -1; }
    hipMalloc(&tmp, tb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 3; it++) {
        hipEventRecord(e0, 0);
        hipError_t e = rocprim::radix_sort_keys<C>(tmp, tb, kb, n, b0, b1, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        if (e != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: sort failed\n", name); return -1; }
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("keys %s bits [%d,%d): %.1f ms\n", name, b0, b1, best); fflush(stdout);
    hipFree(tmp); return best;
}
int main(int argc, char **argv) {
    size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000000ull;
    uint64_t *k0, *k1; uint32_t *v0, *v1;
    hipMalloc(&k0, n * 8); hipMalloc(&k1, n * 8); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
    fill<<<(unsigned) ((n + 255) / 256), 256>>>(k0, v0, n); hipDeviceSynchronize();
    run<rocprim::default_config>(k0, k1, v0, v1, n, 9, 41, "default 4x8");
    run<Cfg<9, 512, 16>>(k0, k1, v0, v1, n, 14, 41, "cfg 9b 512x16 (3 passes)");
    run<Cfg<9, 512, 20>>(k0, k1, v0, v1, n, 14, 41, "cfg 9b 512x20 (3 passes)");
    run<Cfg<9, 256, 32>>(k0, k1, v0, v1, n, 14, 41, "cfg 9b 256x32 (3 passes)");
    run<Cfg<9, 768, 12>>(k0, k1, v0, v1, n, 14, 41, "cfg 9b 768x12 (3 passes)");
    run<Cfg<9, 640, 14>>(k0, k1, v0, v1, n, 14, 41, "cfg 9b 640x14 (3 passes)");
    runk<rocprim::default_config>(k0, k1, n, 9, 41, "default 4x8");
    runk<Cfg<9, 1024, 8>>(k0, k1, n, 14, 41, "cfg 9b 1024x8 (3 passes)");
    runk<Cfg<8, 512, 24>>(k0, k1, n, 9, 41, "cfg 8b 512x24");
    runk<Cfg<8, 1024, 12>>(k0, k1, n, 9, 41, "cfg 8b 1024x12");
    runk<Cfg<8, 768, 16>>(k0, k1, n, 9, 41, "cfg 8b 768x16");
    runk<Cfg<8, 512, 16>>(k0, k1, n, 9, 41, "cfg 8b 512x16");
    runk<Cfg<8, 1024, 16>>(k0, k1, n, 9, 41, "cfg 8b 1024x16");
    return 0;
}
