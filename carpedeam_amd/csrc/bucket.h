// Finishing a most-significant-digit sort inside small buckets.
//
// Both sorts of kmermatcher (kmermatcher.cpp:412 by k-mer, :431 by (rep, id, diagonal)) order a few 10^9 tuples.  A
// least-significant-digit radix sort streams the whole array through HBM once per 8 key bits.  Here only the TOP bits go through
// those global passes (rocPRIM onesweep); what is left are runs of equal high bits ("buckets") that are contiguous in memory,
// and they are finished on chip: a wavefront takes a group of consecutive whole buckets (up to 256 elements, or one bucket of
// up to 512), builds one word per element = (bucket ordinal, low key bits, position) and sorts the words with a bitonic
// network held in registers (exchanges between lanes are shuffles).  The position is part of the compared word, so the result
// is the same stable order the reference's std::sort / ips4o comparators produce on the full key.
//
// Work distribution: the array is cut into ranges of BK_T slots; a block owns the buckets that START in its range and walks them
// in chunks of whole buckets (at most BK_C slots in LDS); wave w of the block owns the buckets that start in the w-th quarter
// of the chunk.  A bucket larger than BK_MAXB is appended to a list and finished by the caller: the listed ranges are
// gathered, sorted on the complete key by rocPRIM and scattered back (bucketSortKeys below, the fused k-mer kernel's
// fallback in kmermatch.hip).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace bucket {

constexpr int BK_NT = 256;              // threads per block
constexpr int BK_WAVES = BK_NT / 64;
constexpr int BK_C = 2048;              // chunk capacity (slots in LDS)
constexpr int BK_T = 1024;              // ownership granule
constexpr int BK_WORDS = BK_C / 64;
constexpr int BK_WIN = BK_C / BK_WAVES; // a wave owns the buckets starting in its window of the chunk
constexpr int BK_GROUP = 256;           // largest group of several buckets
constexpr int BK_MAXB = 512;            // largest single bucket finished in registers
constexpr int BK_ORD = 8;               // bits of a bucket ordinal inside a group

// per-chunk bookkeeping in LDS
struct ChunkLds {
    unsigned long long bits[BK_WORDS];  // "starts a bucket" bit per slot, written with wave ballots
    unsigned long long found64;
    uint16_t pre[BK_WORDS + 1];         // bucket starts before word w
    uint16_t sB[BK_C + 2];              // ordered bucket starts, sB[nB] = chunk length
    unsigned int found, nB;
};
// first p >= x that starts a bucket (p == 0 or high bits differ from p - 1), n if there is none.  Block-uniform.  The array is
// sorted on the high bits: one parallel probe of the next BK_NT slots, then a binary search (a bucket may be very large).
template <typename HiOf>
__device__ __forceinline__ uint64_t findBoundary(const HiOf &hiOf, uint64_t n, uint64_t x, ChunkLds &c) {
    if (x == 0) return 0;
    if (x >= n) return n;
    if (threadIdx.x == 0) c.found = BK_NT;
    __syncthreads();
    const uint64_t p = x + threadIdx.x;
    if (p < n && hiOf(p) != hiOf(p - 1)) atomicMin(&c.found, threadIdx.x);
    __syncthreads();
    const unsigned int f = c.found;
    __syncthreads();
    if (f < (unsigned int) BK_NT) return x + f;
    if (x + BK_NT >= n) return n;
    if (threadIdx.x == 0) {
        const auto h = hiOf(x - 1);
        uint64_t lo = x + BK_NT, hi = n;                // first index in [lo, hi) whose high bits differ from h, or n
        while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (hiOf(mid) == h) lo = mid + 1; else hi = mid; }
        c.found64 = lo;
    }
    __syncthreads();
    const uint64_t r = c.found64;
    __syncthreads();
    return r;
}

// ordered list of the bucket starts from the bit words; runs on wave 0, the block syncs around it
__device__ __forceinline__ void listBucketStarts(ChunkLds &c, int len) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned long long w = (lane < BK_WORDS) ? c.bits[lane] : 0ull;
        const int cnt = __popcll(w);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        int o = incl - cnt;
        if (lane <= BK_WORDS) c.pre[lane] = (uint16_t) o;       // lanes >= BK_WORDS hold the total
        while (w) { const int b = __ffsll(w) - 1; w &= w - 1; c.sB[o++] = (uint16_t) (lane * 64 + b); }
        if (lane == 63) { c.nB = (unsigned int) incl; c.sB[incl] = (uint16_t) len; }
    }
}
// ordinal of the bucket that slot i belongs to
__device__ __forceinline__ int ordOf(const ChunkLds &c, int i) {
    return (int) c.pre[i >> 6] + __popcll(c.bits[i >> 6] & ((2ull << (i & 63)) - 1ull)) - 1;
}
// next group of whole buckets [j, j1) for a wave that owns the buckets [.., jEnd): elements [g0, g1).  Wave-uniform.
struct Group { int g0, g1, j0, j1; };
__device__ __forceinline__ bool nextGroup(const ChunkLds &c, int &j, int jEnd, Group &g) {
    if (j >= jEnd) return false;
    g.j0 = j; g.g0 = c.sB[j];
    int jj = j + 1, e = c.sB[jj];
    while (jj < jEnd && jj - j < (1 << BK_ORD)) {
        const int e2 = c.sB[jj + 1];
        if (e2 - g.g0 > BK_GROUP) break;
        e = e2; jj++;
    }
    g.j1 = jj; g.g1 = e; j = jj;
    return true;
}

template <typename W> __device__ __forceinline__ W shflXorW(W a, int m);
template <> __device__ __forceinline__ uint32_t shflXorW<uint32_t>(uint32_t a, int m) { return (uint32_t) __shfl_xor((int) a, m, 64); }
template <> __device__ __forceinline__ uint64_t shflXorW<uint64_t>(uint64_t a, int m) {
    return ((uint64_t) (uint32_t) __shfl_xor((int) (a >> 32), m, 64) << 32) | (uint32_t) __shfl_xor((int) (uint32_t) a, m, 64);
}
template <typename W> __device__ __forceinline__ W shflUpW(W a, int d);
template <> __device__ __forceinline__ uint32_t shflUpW<uint32_t>(uint32_t a, int d) { return (uint32_t) __shfl_up((int) a, d, 64); }
template <> __device__ __forceinline__ uint64_t shflUpW<uint64_t>(uint64_t a, int d) {
    return ((uint64_t) (uint32_t) __shfl_up((int) (a >> 32), d, 64) << 32) | (uint32_t) __shfl_up((int) (uint32_t) a, d, 64);
}

// Bitonic sorting network over 64 R words held in registers, element i = lane * R + r: exchanges at distance < R stay inside a
// lane, the others go through lane shuffles.  Unused slots hold the all-ones word and end up last.
template <int R, typename W>
__device__ __forceinline__ void bitonicRegs(W (&v)[R], int lane) {
#pragma unroll
    for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < R) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int p = r ^ j;
                    if (p > r) {
                        const bool up = (((lane * R + r) & k) == 0);
                        const W a = v[r], b = v[p];
                        const W lo = a < b ? a : b, hi = a < b ? b : a;
                        v[r] = up ? lo : hi; v[p] = up ? hi : lo;
                    }
                }
            } else {
                const bool lower = ((lane & (j / R)) == 0);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const bool up = (((lane * R + r) & k) == 0);
                    const W a = v[r], b = shflXorW<W>(a, j / R);
                    const W lo = a < b ? a : b, hi = a < b ? b : a;
                    v[r] = (up == lower) ? lo : hi;
                }
            }
        }
    }
}
// sorts the gm <= 64 R elements i = 0.. of a group by mk(i) and hands the sorted register array to done(v)
template <int R, typename W, typename MakeComp, typename Done>
__device__ __forceinline__ void sortGroupRegs(int gm, int lane, const MakeComp &mk, const Done &done) {
    W v[R];
#pragma unroll
    for (int r = 0; r < R; r++) { const int i = lane * R + r; v[r] = (i < gm) ? mk(i) : (W) ~(W) 0; }
    bitonicRegs<R, W>(v, lane);
    done(v);
}
template <typename W, typename MakeComp, typename Done>
__device__ __forceinline__ void sortGroup(int gm, int lane, const MakeComp &mk, const Done &done) {
    if (gm <= 64) sortGroupRegs<1, W>(gm, lane, mk, done);
    else if (gm <= 128) sortGroupRegs<2, W>(gm, lane, mk, done);
    else if (gm <= 256) sortGroupRegs<4, W>(gm, lane, mk, done);
    else sortGroupRegs<8, W>(gm, lane, mk, done);
}
// makes the LDS writes of a wave visible to its other lanes (one wave works on a group, no block barrier)
__device__ __forceinline__ void waveLdsSync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct BigList {
    unsigned long long *list;       // (start, end) of every bucket left to the caller
    unsigned int *cnt;
    __device__ __forceinline__ void add(uint64_t s, uint64_t e) const { const unsigned int q = atomicAdd(cnt, 1u); list[2 * (size_t) q] = s; list[2 * (size_t) q + 1] = e; }
};

struct SortArgs {
    const uint64_t *in; uint64_t *out; uint64_t n;
    int shiftHi;                    // bucket id = key >> shiftHi (the array is sorted on it)
    int ign;                        // lowest bits that ride along uncompared
    uint32_t cap, maxBucket;        // chunk / single-bucket capacity in use (tests lower them to reach the big-bucket path)
    BigList big;
};
struct HiOfKey {
    const uint64_t *in; int shift;
    __device__ __forceinline__ uint64_t operator()(uint64_t p) const { return in[p] >> shift; }
};

// keys only, out of place: out = in with every bucket stably sorted on bits [ign, shiftHi)
__global__ __launch_bounds__(BK_NT) void k_bucket_sort(SortArgs a) {
    __shared__ uint64_t sKey[BK_C];
    __shared__ ChunkLds c;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const HiOfKey hiOf{a.in, a.shiftHi};
    const uint64_t r0 = (uint64_t) blockIdx.x * BK_T;
    uint64_t pos = findBoundary(hiOf, a.n, r0, c);
    const uint64_t end = (r0 + BK_T >= a.n) ? a.n : findBoundary(hiOf, a.n, r0 + BK_T, c);
    const uint64_t lowMask = (1ull << a.shiftHi) - 1ull;
    const int lowW = a.shiftHi - a.ign;
    while (pos < end) {
        int len = (int) min((uint64_t) a.cap, end - pos);
        const bool cut = pos + (uint64_t) len < end;
        for (int i = tid; i < BK_C; i += BK_NT) {       // BK_C is a multiple of the block size: whole waves, ballots are complete
            bool first = false;
            if (i < len) {
                const uint64_t k = a.in[pos + i];
                sKey[i] = k;
                first = (i == 0) || ((k >> a.shiftHi) != (a.in[pos + i - 1] >> a.shiftHi));
            }
            const unsigned long long m = __ballot(first);
            if (lane == 0) c.bits[i >> 6] = m;
        }
        __syncthreads();
        listBucketStarts(c, len);
        __syncthreads();
        int nB = (int) c.nB;
        if (cut) {
            if (nB == 1) {      // the bucket at pos does not fit a chunk
                const uint64_t bEnd = findBoundary(hiOf, a.n, pos + 1, c);
                if (tid == 0) a.big.add(pos, bEnd);
                pos = bEnd;
                __syncthreads();
                continue;
            }
            nB--; len = c.sB[nB];     // drop the partial bucket at the end of the chunk (sB[nB] is the new length)
        }
        // wave w: buckets starting in [w BK_WIN, (w + 1) BK_WIN)
        int j = c.pre[wave * (BK_WIN / 64)];
        const int jEnd = min(nB, (int) c.pre[(wave + 1) * (BK_WIN / 64)]);
        Group g;
        while (nextGroup(c, j, jEnd, g)) {
            const int gm = g.g1 - g.g0;
            if (gm > (int) a.maxBucket) { if (lane == 0) a.big.add(pos + g.g0, pos + g.g1); continue; }
            const int idxBits = gm > 256 ? 9 : 8;
            const uint64_t idxMask = (1ull << idxBits) - 1ull;
            const int g0 = g.g0, j0 = g.j0, ign = a.ign;
            sortGroup<uint64_t>(gm, lane,
                [&](int i) {
                    const uint64_t low = (sKey[g0 + i] & lowMask) >> ign;
                    return ((((uint64_t) (ordOf(c, g0 + i) - j0) << lowW) | low) << idxBits) | (uint64_t) i;
                },
                [&](auto &v) {
                    constexpr int R = sizeof(v) / sizeof(v[0]);
#pragma unroll
                    for (int r = 0; r < R; r++) { const int p = lane * R + r; if (p < gm) a.out[pos + g0 + p] = sKey[g0 + (int) (v[r] & idxMask)]; }
                });
        }
        pos += (uint64_t) len;
        __syncthreads();
    }
}

// copies the listed ranges between the array and a dense staging buffer (ranges sorted by start, off = prefix sums of sizes)
template <typename T, bool GATHER>
__global__ __launch_bounds__(256) void k_big_copy(const unsigned long long *__restrict__ ranges /* start, end, off */, unsigned int cnt, T *arr, T *dense) {
    for (unsigned int r = blockIdx.x; r < cnt; r += gridDim.x) {
        const unsigned long long s = ranges[3 * (size_t) r], e = ranges[3 * (size_t) r + 1], o = ranges[3 * (size_t) r + 2];
        for (unsigned long long i = threadIdx.x; i < e - s; i += 256) {
            if (GATHER) dense[o + i] = arr[s + i]; else arr[s + i] = dense[o + i];
        }
    }
}

// host: reads the big-bucket list, returns it sorted by start with prefix offsets (device copy in `ranges`)
inline int loadBigList(hipStream_t s, const unsigned long long *bigList, unsigned int cnt, DevBuf<unsigned long long> &ranges, uint64_t &total,
                       unsigned long long *firstStart = nullptr) {
    std::vector<unsigned long long> raw(2 * (size_t) cnt);
    if (hipMemcpyAsync(raw.data(), bigList, raw.size() * 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return CDM_ERR_HIP;
    std::vector<std::pair<unsigned long long, unsigned long long>> v(cnt);
    for (unsigned int i = 0; i < cnt; i++) v[i] = {raw[2 * (size_t) i], raw[2 * (size_t) i + 1]};
    std::sort(v.begin(), v.end());
    std::vector<unsigned long long> h(3 * (size_t) cnt);
    total = 0;
    for (unsigned int i = 0; i < cnt; i++) { h[3 * (size_t) i] = v[i].first; h[3 * (size_t) i + 1] = v[i].second; h[3 * (size_t) i + 2] = total; total += v[i].second - v[i].first; }
    if (firstStart) *firstStart = cnt ? v[0].first : ~0ull;
    if (!ranges.alloc(h.size())) return CDM_ERR_HIP;
    if (hipMemcpyAsync(ranges.p, h.data(), h.size() * 8, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return CDM_ERR_HIP;
    return CDM_OK;
}

// CDM_BUCKET_CAP=<chunk slots>[,<largest bucket>] lowers the capacities (tests: reach the big-bucket path on small inputs)
inline void capacities(uint32_t &cap, uint32_t &maxBucket) {
    cap = BK_C; maxBucket = BK_MAXB;
    if (const char *e = getenv("CDM_BUCKET_CAP")) {
        const long v = atol(e);
        if (v >= 2 && v <= BK_C) cap = (uint32_t) v;
        if (const char *comma = strchr(e, ',')) { const long m = atol(comma + 1); if (m >= 1 && m <= BK_MAXB) maxBucket = (uint32_t) m; }
    }
}
inline size_t bigListSlots(uint64_t n, uint32_t cap, uint32_t maxBucket) { return 2 * (size_t) (n / std::min(cap, maxBucket + 1) + 2); }

// `in` is stably sorted on key bits [shiftHi, top); afterwards `out` is stably sorted on [ign, top).  Synchronises the stream.
inline int bucketSortKeys(hipStream_t s, const uint64_t *in, uint64_t *out, uint64_t n, int shiftHi, int ign, int top) {
    if (n == 0) return CDM_OK;
    uint32_t cap, maxBucket; capacities(cap, maxBucket);
    DevBuf<unsigned long long> bigList; DevBuf<unsigned int> bigCnt;
    if (!bigList.alloc(bigListSlots(n, cap, maxBucket)) || !bigCnt.alloc(1)) return CDM_ERR_HIP;
    hipMemsetAsync(bigCnt.p, 0, 4, s);
    SortArgs a; a.in = in; a.out = out; a.n = n; a.shiftHi = shiftHi; a.ign = ign; a.cap = cap; a.maxBucket = maxBucket; a.big.list = bigList.p; a.big.cnt = bigCnt.p;
    hipLaunchKernelGGL(k_bucket_sort, dim3((unsigned) ((n + BK_T - 1) / BK_T)), dim3(BK_NT), 0, s, a);
    unsigned int cnt = 0;
    if (hipMemcpyAsync(&cnt, bigCnt.p, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return CDM_ERR_HIP;
    if (cnt == 0) return CDM_OK;
    DevBuf<unsigned long long> ranges; uint64_t total = 0;
    if (int rc = loadBigList(s, bigList.p, cnt, ranges, total)) return rc;
    if (getenv("CDM_BUCKET_STATS")) fprintf(stderr, "bucketSortKeys: n %llu shiftHi %d: %u big buckets, %llu elements\n", (unsigned long long) n, shiftHi, cnt, (unsigned long long) total);
    DevBuf<uint64_t> d0, d1; DevBuf<char> tmp; size_t tb = 0;
    if (!d0.alloc(total) || !d1.alloc(total)) return CDM_ERR_HIP;
    const unsigned int grid = std::min<unsigned int>(cnt, 1u << 20);
    hipLaunchKernelGGL((k_big_copy<uint64_t, true>), dim3(grid), dim3(256), 0, s, (const unsigned long long *) ranges.p, cnt, const_cast<uint64_t *>(in), d0.p);
    rocprim::double_buffer<uint64_t> db(d0.p, d1.p);
    if (rocprim::radix_sort_keys(nullptr, tb, db, (size_t) total, ign, top, s) != hipSuccess || !tmp.alloc(tb + 256)) return CDM_ERR_HIP;
    if (rocprim::radix_sort_keys(tmp.p, tb, db, (size_t) total, ign, top, s) != hipSuccess) return CDM_ERR_HIP;
    hipLaunchKernelGGL((k_big_copy<uint64_t, false>), dim3(grid), dim3(256), 0, s, (const unsigned long long *) ranges.p, cnt, out, db.current());
    if (hipStreamSynchronize(s) != hipSuccess) return CDM_ERR_HIP;
    return CDM_OK;
}

}  // namespace bucket
