// Probe: a launch of more than 2^32 threads - does it run, and how many threads show up?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_mark(unsigned long long *cnt, unsigned long long *maxIdx) {
    const unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
    if ((threadIdx.x & 63) == 0) { atomicAdd(cnt, 64ull); atomicMax(maxIdx, i); }
}
int main() {
    (void) hipSetDevice(0);
    unsigned long long *d, h[2];
    (void) hipMalloc(&d, 16);
    for (unsigned long long blocks : {1000000ull, 16777215ull, 16777216ull, 16777217ull, 18500000ull, 40000000ull}) {
        (void) hipMemset(d, 0, 16);
        hipLaunchKernelGGL(k_mark, dim3((unsigned) blocks), dim3(256), 0, 0, d, d + 1);
        hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
        (void) hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%llu blocks x 256 = %llu threads: launch %s, sync %s, %llu threads ran, highest index %llu\n", blocks, blocks * 256, hipGetErrorString(e1), hipGetErrorString(e2), h[0], h[1]);
    }
    return 0;
}
