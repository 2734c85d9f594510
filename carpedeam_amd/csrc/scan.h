// Device-wide exclusive prefix sum, hand-written for gfx950 (no library): reduce per 4096-item tile, scan the tile sums
// (recursively, one level per factor 4096), then scan inside the tiles with their base.  The input is a functor of the index,
// so lengths packed in records, flags and plain arrays all go through the same three kernels; in-place use (out == the array
// the functor reads) is fine: every tile is read completely before it is written.
#pragma once
#include "common.h"
#include "devutil.h"

namespace cdmscan {

constexpr int SC_NT = 256, SC_ITEMS = 16, SC_TILE = SC_NT * SC_ITEMS;

template <typename T> struct LoadArray { const T *p; __device__ __forceinline__ T operator()(size_t i) const { return p[i]; } };
template <typename T, typename TIn> struct LoadAs { const TIn *p; __device__ __forceinline__ T operator()(size_t i) const { return (T) p[i]; } };

template <typename T, typename Load>
__global__ __launch_bounds__(SC_NT) void k_scan_reduce(Load load, size_t n, T *__restrict__ partial) {
    const size_t base = (size_t) blockIdx.x * SC_TILE;
    T c = 0;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) { const size_t i = base + threadIdx.x + (size_t) SC_NT * j; if (i < n) c += load(i); }
    c = cdm_block_sum<T>(c);
    if (threadIdx.x == 0) partial[blockIdx.x] = c;
}
// exclusive scan of one tile; tileBase = exclusive scan of the tile sums (NULL: a single tile starting at 0)
template <typename T, typename Load>
__global__ __launch_bounds__(SC_NT) void k_scan_apply(Load load, size_t n, const T *__restrict__ tileBase, T *__restrict__ out) {
    __shared__ T sItems[SC_TILE + SC_TILE / 16 + 1];           // one pad per 16 items: thread t then walks 16 t .. 16 t + 15
    const size_t base = (size_t) blockIdx.x * SC_TILE;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) { const int li = threadIdx.x + SC_NT * j; const size_t i = base + li; sItems[li + (li >> 4)] = (i < n) ? load(i) : (T) 0; }
    __syncthreads();
    T v[SC_ITEMS], c = 0;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) { const int li = threadIdx.x * SC_ITEMS + j; v[j] = sItems[li + (li >> 4)]; c += v[j]; }
    T tot;
    T run = cdm_block_excl_sum<T>(c, tot) + (tileBase ? tileBase[blockIdx.x] : (T) 0);
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) { const int li = threadIdx.x * SC_ITEMS + j; sItems[li + (li >> 4)] = run; run += v[j]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) { const int li = threadIdx.x + SC_NT * j; const size_t i = base + li; if (i < n) out[i] = sItems[li + (li >> 4)]; }
}

// Temporaries of a scan (tile sums, one buffer per level).  The scan is asynchronous: keep this object alive until the stream
// has been synchronised (the caching allocator may hand a freed block to another context's stream).
struct ScanTemp {
    std::vector<void *> bufs;
    ScanTemp() = default;
    ScanTemp(const ScanTemp &) = delete;
    ScanTemp &operator=(const ScanTemp &) = delete;
    ~ScanTemp() { for (void *p : bufs) cdmFree(p); }
    void *get(size_t bytes) { void *p = nullptr; if (cdmMallocRaw(&p, bytes) != hipSuccess) return nullptr; bufs.push_back(p); return p; }
};

// out[i] = sum of load(j) for j < i, i in [0, n).  Asynchronous on s.
template <typename T, typename Load>
inline int exclusiveScanFn(hipStream_t s, ScanTemp &tmp, Load load, T *out, size_t n, int depth = 0) {
    if (n == 0) return CDM_OK;
    const size_t tiles = (n + SC_TILE - 1) / SC_TILE;
    if (tiles == 1) { hipLaunchKernelGGL((k_scan_apply<T, Load>), dim3(1), dim3(SC_NT), 0, s, load, n, (const T *) nullptr, out); return CDM_OK; }
    if (depth >= 4) { cdm_set_error("exclusive scan: input too long"); return CDM_ERR_INVALID; }
    T *partial = reinterpret_cast<T *>(tmp.get((tiles + 1) * sizeof(T)));
    if (!partial) { cdm_set_error("exclusive scan: out of device memory"); return CDM_ERR_HIP; }
    hipLaunchKernelGGL((k_scan_reduce<T, Load>), dim3((unsigned) tiles), dim3(SC_NT), 0, s, load, n, partial);
    if (int rc = exclusiveScanFn<T, LoadArray<T>>(s, tmp, LoadArray<T>{partial}, partial, tiles, depth + 1)) return rc;
    hipLaunchKernelGGL((k_scan_apply<T, Load>), dim3((unsigned) tiles), dim3(SC_NT), 0, s, load, n, (const T *) partial, out);
    return CDM_OK;
}
template <typename T>
inline int exclusiveScan(hipStream_t s, ScanTemp &tmp, const T *in, T *out, size_t n) { return exclusiveScanFn<T, LoadArray<T>>(s, tmp, LoadArray<T>{in}, out, n); }

// ---------------------------------------------------------------------------------------------- inclusive max-scan (u64)
// out[i] = max of load(j) for j <= i.  Same three steps; the tile bases are the inclusive max-scan of the tile maxima, shifted by one.
typedef unsigned long long mx_t;
__device__ __forceinline__ mx_t mxMax(mx_t a, mx_t b) { return a > b ? a : b; }
// exclusive prefix max over the block (identity 0); total = max over the block
__device__ __forceinline__ mx_t blockExclMax(mx_t v, mx_t &total) {
    __shared__ mx_t sWave[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    mx_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const mx_t o = cdm_shfl_up_t<mx_t>(incl, d); if (lane >= d) incl = mxMax(incl, o); }
    mx_t excl = cdm_shfl_up_t<mx_t>(incl, 1);
    if (lane == 0) excl = 0;
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    mx_t base = 0, tot = 0;
    for (int w = 0; w < nw; w++) { const mx_t c = sWave[w]; if (w < wave) base = mxMax(base, c); tot = mxMax(tot, c); }
    __syncthreads();
    total = tot;
    return mxMax(base, excl);
}
template <typename Load>
__global__ __launch_bounds__(SC_NT) void k_maxscan_reduce(Load load, size_t n, mx_t *__restrict__ partial) {
    const size_t base = (size_t) blockIdx.x * SC_TILE;
    mx_t c = 0;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) { const size_t i = base + threadIdx.x + (size_t) SC_NT * j; if (i < n) c = mxMax(c, load(i)); }
    mx_t tot;
    (void) blockExclMax(c, tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
// tileIncl = inclusive max-scan of the tile maxima (NULL: a single tile)
template <typename Load>
__global__ __launch_bounds__(SC_NT) void k_maxscan_apply(Load load, size_t n, const mx_t *__restrict__ tileIncl, mx_t *__restrict__ out) {
    __shared__ mx_t sItems[SC_TILE + SC_TILE / 16 + 1];
    const size_t base = (size_t) blockIdx.x * SC_TILE;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) { const int li = threadIdx.x + SC_NT * j; const size_t i = base + li; sItems[li + (li >> 4)] = (i < n) ? load(i) : 0ull; }
    __syncthreads();
    mx_t v[SC_ITEMS], c = 0;
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) { const int li = threadIdx.x * SC_ITEMS + j; v[j] = sItems[li + (li >> 4)]; c = mxMax(c, v[j]); }
    mx_t tot;
    mx_t run = mxMax(blockExclMax(c, tot), (tileIncl && blockIdx.x > 0) ? tileIncl[blockIdx.x - 1] : 0ull);
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) { const int li = threadIdx.x * SC_ITEMS + j; run = mxMax(run, v[j]); sItems[li + (li >> 4)] = run; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SC_ITEMS; j++) { const int li = threadIdx.x + SC_NT * j; const size_t i = base + li; if (i < n) out[i] = sItems[li + (li >> 4)]; }
}
template <typename Load>
inline int inclusiveMaxScanFn(hipStream_t s, ScanTemp &tmp, Load load, mx_t *out, size_t n, int depth = 0) {
    if (n == 0) return CDM_OK;
    const size_t tiles = (n + SC_TILE - 1) / SC_TILE;
    if (tiles == 1) { hipLaunchKernelGGL((k_maxscan_apply<Load>), dim3(1), dim3(SC_NT), 0, s, load, n, (const mx_t *) nullptr, out); return CDM_OK; }
    if (depth >= 4) { cdm_set_error("max scan: input too long"); return CDM_ERR_INVALID; }
    mx_t *partial = reinterpret_cast<mx_t *>(tmp.get((tiles + 1) * sizeof(mx_t)));
    if (!partial) { cdm_set_error("max scan: out of device memory"); return CDM_ERR_HIP; }
    hipLaunchKernelGGL((k_maxscan_reduce<Load>), dim3((unsigned) tiles), dim3(SC_NT), 0, s, load, n, partial);
    if (int rc = inclusiveMaxScanFn<LoadArray<mx_t>>(s, tmp, LoadArray<mx_t>{partial}, partial, tiles, depth + 1)) return rc;
    hipLaunchKernelGGL((k_maxscan_apply<Load>), dim3((unsigned) tiles), dim3(SC_NT), 0, s, load, n, (const mx_t *) partial, out);
    return CDM_OK;
}

}  // namespace cdmscan
