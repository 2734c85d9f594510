// Finishing a most-significant-digit sort inside small buckets.
//
// Both sorts of kmermatcher (kmermatcher.cpp:412 by k-mer, :431 by (rep, id, diagonal)) order a few 10^9 tuples.  A
// least-significant-digit radix sort streams the whole array through HBM once per 8 key bits.  Here only the TOP bits (27 for the
// k-mer sort; 32 for the group-key sort in its radix variant) go through those global passes (the onesweep of radix.h); what is left are runs
// of equal high bits ("buckets") that are contiguous in
// memory, and they are finished on chip: a wavefront takes a group of consecutive whole buckets (up to 256 elements, or one
// bucket of up to 512), builds one word per element = (bucket ordinal, low key bits, position) and sorts the words with a
// bitonic network held in registers (exchanges between lanes are DPP row permutations / ds_swizzle).  The position is part of the compared word, so the
// result is the same stable order the reference's std::sort / ips4o comparators produce on the full key.
//
// Work distribution: no block-level synchronisation at all.  The array is cut into ranges of WV_OWN slots; a WAVE owns the
// buckets that START in its range, stages a window of WV_WIN slots (its range plus the longest bucket it can finish) in its
// private part of the LDS, finds the bucket starts with ballots and walks them group by group.  A bucket that does not fit
// (more than BK_MAXB elements) is appended to a list and finished by the caller: the listed ranges are gathered, sorted on the
// complete key by the global radix sort and scattered back (bucketSortKeys below, the fused k-mer kernel's fallback in kmermatch.hip).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "radix.h"
#include "scan.h"

namespace bucket {

constexpr int BK_NT = 256;              // threads per block: 4 independent waves
constexpr int BK_WAVES = BK_NT / 64;
constexpr int WV_OWN = 256;             // a wave owns the buckets that start in its range of this many slots
constexpr int WV_WIN = 768;             // slots a wave stages: the owned range + the longest bucket it finishes itself
constexpr int WV_WORDS = WV_WIN / 64;
constexpr int WV_FIRST = 512;           // staged up front; the rest only if the last owned bucket runs on
constexpr int BK_GROUP = 256;           // largest group of several buckets
constexpr int BK_MAXB = 512;            // largest single bucket finished in registers
constexpr int BK_ORD = 8;               // bits of a bucket ordinal inside a group

// makes the LDS writes of a wave visible to its other lanes (one wave works on its own data, no block barrier)
__device__ __forceinline__ void waveLdsSync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// per-wave bookkeeping in LDS: "starts a bucket" bit per window slot (written with ballots) and the ordinal of every slot's bucket
template <int WIN> struct WaveLdsT {
    unsigned long long bits[WIN / 64];
    uint16_t ord[WIN];
};
typedef WaveLdsT<WV_WIN> WaveLds;
// first set bit in [from, limit), -1 if none.  Wave-uniform.
template <typename WL> __device__ __forceinline__ int firstSetFrom(const WL &w, int from, int limit) {
    for (int word = from >> 6; word * 64 < limit; word++) {
        unsigned long long m = w.bits[word];
        if (word == (from >> 6)) m &= ~0ull << (from & 63);
        if (m) { const int p = word * 64 + __ffsll(m) - 1; return p < limit ? p : -1; }
    }
    return -1;
}
// last set bit in (lo, hi], -1 if none.  Wave-uniform.
template <typename WL> __device__ __forceinline__ int lastSetIn(const WL &w, int lo, int hi) {
    const int wl = (lo + 1) >> 6;
    for (int word = hi >> 6; word >= wl; word--) {
        unsigned long long m = w.bits[word];
        if (word == (hi >> 6)) m &= (2ull << (hi & 63)) - 1ull;
        if (word == wl) m &= ~0ull << ((lo + 1) & 63);
        if (m) return word * 64 + 63 - __clzll(m);
    }
    return -1;
}

struct BigList {
    unsigned long long *list;       // (start, end) of every range left to the caller
    unsigned int *cnt;
    // list == NULL: the caller knows the big ranges already (segmentedSortKeys) and handles them itself
    __device__ __forceinline__ void add(uint64_t s, uint64_t e) const { if (!list) return; const unsigned int q = atomicAdd(cnt, 1u); list[2 * (size_t) q] = s; list[2 * (size_t) q + 1] = e; }
};

// value of lane (lane ^ M): DPP lane permutations inside a row of 16 (no LDS round trip), ds_swizzle across rows of a half,
// a general shuffle only between the two halves of the wave
template <int M> __device__ __forceinline__ uint32_t xorLane32(uint32_t a) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "lane distance");
    if (M == 1) return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) a, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    if (M == 2) return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) a, 0x4E, 0xF, 0xF, false);        // quad_perm [2,3,0,1]
    if (M == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, (int) a, 0x104, 0xF, 0x5, false);                    // row_shl:4 into banks 0, 2
        return (uint32_t) __builtin_amdgcn_update_dpp(t, (int) a, 0x114, 0xF, 0xA, false);                // row_shr:4 into banks 1, 3
    }
    if (M == 8) return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) a, 0x128, 0xF, 0xF, false);       // row_ror:8
    if (M == 16) return (uint32_t) __builtin_amdgcn_ds_swizzle((int) a, 0x401F);                          // xor 0x10 within 32 lanes
    return (uint32_t) __shfl_xor((int) a, 32, 64);
}
template <typename W, int M> struct XorLane;
template <int M> struct XorLane<uint32_t, M> { __device__ static __forceinline__ uint32_t get(uint32_t a) { return xorLane32<M>(a); } };
template <int M> struct XorLane<uint64_t, M> {
    __device__ static __forceinline__ uint64_t get(uint64_t a) { return ((uint64_t) xorLane32<M>((uint32_t) (a >> 32)) << 32) | xorLane32<M>((uint32_t) a); }
};
typedef unsigned __int128 u128;         // sort word of the aggregation when (ordinal, id, diagonal, index) needs more than 64 bits
template <int M> struct XorLane<u128, M> {
    __device__ static __forceinline__ u128 get(u128 a) { return ((u128) XorLane<uint64_t, M>::get((uint64_t) (a >> 64)) << 64) | XorLane<uint64_t, M>::get((uint64_t) a); }
};
// d is a constant once the network loops are unrolled: the switch folds away
template <typename W> __device__ __forceinline__ W xorLane(W a, int d) {
    switch (d) {
        case 1: return XorLane<W, 1>::get(a);
        case 2: return XorLane<W, 2>::get(a);
        case 4: return XorLane<W, 4>::get(a);
        case 8: return XorLane<W, 8>::get(a);
        case 16: return XorLane<W, 16>::get(a);
        default: return XorLane<W, 32>::get(a);
    }
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
                        const bool sw = (b < a) == up;           // one compare, the direction is folded into the lane mask
                        v[r] = sw ? b : a; v[p] = sw ? a : b;
                    }
                }
            } else {
                const bool lower = ((lane & (j / R)) == 0);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const bool up = (((lane * R + r) & k) == 0);
                    const W a = v[r], b = xorLane<W>(a, j / R);
                    v[r] = ((b < a) == (up == lower)) ? b : a;
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
// value of lane - 1 (wave_shr:1), lane 0 keeps its own
__device__ __forceinline__ uint64_t prevLane64(uint64_t a) {
    const uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp((int) (uint32_t) a, (int) (uint32_t) a, 0x138, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp((int) (uint32_t) (a >> 32), (int) (uint32_t) (a >> 32), 0x138, 0xF, 0xF, false);
    return ((uint64_t) hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readLane64(uint64_t a, int l) {
    return ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (a >> 32), l) << 32) | (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) a, l);
}

// The walk of one wave over the buckets it owns.  T fetch(g) loads the tuple at global index g into registers, put(i, t) stores
// it in window slot i and returns its key, keyAt(g) returns the key of the tuple at g; two tuples are in the same bucket when
// their keys agree in the bits of hiMask.  groupFn(g0, gm) finishes the group of whole buckets in the window slots
// [g0, g0 + gm).  w.ord[i] = ordinal of slot i's bucket in the window.
// WIN / FIRST: slots of the window / of its first batch (the defaults are WV_WIN / WV_FIRST).
template <typename T, int WIN = WV_WIN, int FIRST = WV_FIRST, typename Fetch, typename Put, typename KeyAt, typename GroupFn>
__device__ __forceinline__ void waveBuckets(uint64_t r0, uint64_t n, int own, uint32_t maxBucket, uint64_t hiMask, const BigList &big, WaveLdsT<WIN> &w, int lane,
                                            const Fetch &fetch, const Put &put, const KeyAt &keyAt, const GroupFn &groupFn) {
    constexpr int WV_WIN = WIN, WV_FIRST = FIRST, WV_WORDS = WIN / 64;      // (shadow the defaults)
    const int avail = (int) min((uint64_t) WV_WIN, n - r0);
    uint64_t carry = r0 ? keyAt(r0 - 1) : 0ull;     // key in front of the row being flagged (wave-uniform)
    int ordBase = -1;
    // all loads of a batch of rows are issued before the first one is consumed
    auto loadRows = [&](auto rowsTag, int t0, auto fullTag) {
        constexpr int ROWS = decltype(rowsTag)::value;
        constexpr bool FULL = decltype(fullTag)::value;         // every slot of these rows exists
        T tup[ROWS];
#pragma unroll
        for (int t = 0; t < ROWS; t++) { const int i = (t0 + t) * 64 + lane; if (FULL || i < avail) tup[t] = fetch(r0 + (uint64_t) i); }
#pragma unroll
        for (int t = 0; t < ROWS; t++) {
            const int i = (t0 + t) * 64 + lane;
            const bool valid = FULL || i < avail;
            const uint64_t k = valid ? put(i, tup[t]) : 0ull;
            const bool differs = ((k ^ prevLane64(k)) & hiMask) != 0ull;
            unsigned long long m = __ballot(valid && differs) & ~1ull;              // lane 0 compares with the previous row / range
            if ((FULL || (t0 + t) * 64 < avail) && ((r0 == 0 && t0 + t == 0) || ((readLane64(k, 0) ^ carry) & hiMask) != 0ull)) m |= 1ull;
            if (lane == 0) w.bits[t0 + t] = m;
            const int below = (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
            if (valid) w.ord[i] = (uint16_t) (ordBase + below + (int) ((m >> lane) & 1ull));
            ordBase += __popcll(m);
            carry = readLane64(k, 63);
        }
    };
    if (lane >= WV_FIRST / 64 && lane < WV_WORDS) w.bits[lane] = 0ull;
    if (avail >= WV_FIRST) loadRows(std::integral_constant<int, WV_FIRST / 64>(), 0, std::true_type());
    else loadRows(std::integral_constant<int, WV_FIRST / 64>(), 0, std::false_type());
    waveLdsSync();
    const int ownEnd = min(own, avail);
    const int sLast = lastSetIn(w, -1, ownEnd - 1);         // start of the last bucket this wave owns
    if (sLast < 0) return;
    const int p0 = firstSetFrom(w, 0, ownEnd);
    int loaded = min(avail, WV_FIRST);
    int eLast = firstSetFrom(w, sLast + 1, loaded);
    if (eLast < 0 && loaded < avail) {
        loadRows(std::integral_constant<int, WV_WORDS - WV_FIRST / 64>(), WV_FIRST / 64, std::false_type());
        waveLdsSync();
        loaded = avail;
        eLast = firstSetFrom(w, sLast + 1, loaded);
    }
    if (eLast < 0 && r0 + (uint64_t) avail == n) eLast = avail;         // the array ends inside the window
    int eOwn = eLast;
    if (eLast < 0) {        // the last owned bucket is longer than the window: find its end, leave it to the caller
        const uint64_t h = keyAt(r0 + (uint64_t) sLast) & hiMask;
        uint64_t lo = r0 + (uint64_t) avail, hi = n;
        while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if ((keyAt(mid) & hiMask) == h) lo = mid + 1; else hi = mid; }
        if (lane == 0) big.add(r0 + (uint64_t) sLast, lo);
        eOwn = sLast;
    }
    int g0 = p0;
    while (g0 < eOwn) {
        const int lim = min(g0 + min(BK_GROUP, (int) maxBucket), eOwn);       // (a group of several buckets never exceeds the capacity either)
        int e = (lim == eOwn) ? eOwn : lastSetIn(w, g0, lim);
        if (e < 0) { e = firstSetFrom(w, g0 + 1, eOwn); if (e < 0) e = eOwn; }     // one bucket larger than a group
        const int gm = e - g0;
        if (gm > (int) maxBucket) { if (lane == 0) big.add(r0 + (uint64_t) g0, r0 + (uint64_t) e); }
        else groupFn(g0, gm);
        g0 = e;
    }
}

constexpr int WV_IDX = 10;      // bits of a window slot / of a bucket ordinal in the sorted word

struct SortArgs {
    const uint64_t *in; uint64_t *out; uint64_t n;
    int shiftHi;                    // bucket id = key >> shiftHi (the array is sorted on it)
    int ign;                        // lowest bits that ride along uncompared
    int own; uint32_t maxBucket;    // WV_OWN / BK_MAXB; tests lower them to reach the other paths on small inputs
    BigList big;
};

// keys only, out of place: out = in with every bucket stably sorted on bits [ign, shiftHi)
__global__ __launch_bounds__(BK_NT) void k_bucket_sort(SortArgs a) {
    __shared__ uint64_t sKeyAll[BK_WAVES][WV_WIN];
    __shared__ WaveLds wAll[BK_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t r0 = ((uint64_t) blockIdx.x * BK_WAVES + wave) * (uint64_t) a.own;
    if (r0 >= a.n) return;
    uint64_t *sKey = sKeyAll[wave];
    WaveLds &w = wAll[wave];
    const uint64_t lowMask = (1ull << a.shiftHi) - 1ull;
    const int lowW = a.shiftHi - a.ign, ign = a.ign;
    waveBuckets<uint64_t>(r0, a.n, a.own, a.maxBucket, ~lowMask, a.big, w, lane,
        [&](uint64_t g) { return a.in[g]; },
        [&](int i, uint64_t k) { sKey[i] = k; return k; },
        [&](uint64_t g) { return a.in[g]; },
        [&](int g0, int gm) {
            sortGroup<uint64_t>(gm, lane,
                [&](int i) {
                    const uint64_t low = (sKey[g0 + i] & lowMask) >> ign;
                    return ((((uint64_t) w.ord[g0 + i] << lowW) | low) << WV_IDX) | (uint64_t) (g0 + i);
                },
                [&](auto &v) {
                    constexpr int R = sizeof(v) / sizeof(v[0]);
#pragma unroll
                    for (int r = 0; r < R; r++) { const int p = lane * R + r; if (p < gm) a.out[r0 + (uint64_t) (g0 + p)] = sKey[(int) (v[r] & ((1u << WV_IDX) - 1u))]; }
                });
        });
}

// ---- segments too long for one wave's registers: a block of WAVES wavefronts, 512 elements per wave
// Bitonic network over N = 512 WAVES words, element i = wave * 512 + lane * 8 + r: exchanges at distance < 8 stay inside a lane,
// < 512 go through lane permutations (as in bitonicRegs), the few at distance >= 512 through LDS (sX: N words; the image is
// laid out [r][wave][lane], so that both sides of an exchange are conflict-free).
template <int WAVES, typename W>
__device__ __forceinline__ void bitonicBlock(W (&v)[8], int lane, int wave, W *sX) {
    constexpr int R = 8, N = 64 * R * WAVES;
    const int base = wave * 64 * R + lane * R;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64 * R) {
                const int pw = wave ^ (j / (64 * R));           // the partner has the same lane and register in another wave
#pragma unroll
                for (int r = 0; r < R; r++) sX[(r * WAVES + wave) * 64 + lane] = v[r];
                __syncthreads();
                const bool lower = ((base & j) == 0);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const bool up = (((base + r) & k) == 0);
                    const W a = v[r], b = sX[(r * WAVES + pw) * 64 + lane];
                    v[r] = ((b < a) == (up == lower)) ? b : a;
                }
                __syncthreads();
            } else if (j >= R) {
                const bool lower = ((lane & (j / R)) == 0);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const bool up = (((base + r) & k) == 0);
                    const W a = v[r], b = xorLane<W>(a, j / R);
                    v[r] = ((b < a) == (up == lower)) ? b : a;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int p = r ^ j;
                    if (p > r) {
                        const bool up = (((base + r) & k) == 0);
                        const W a = v[r], b = v[p];
                        const bool sw = (b < a) == up;
                        v[r] = sw ? b : a; v[p] = sw ? a : b;
                    }
                }
            }
        }
    }
}
constexpr int BLK_IDX = 12;             // bits of an element index inside a block-sorted segment (<= 4096 elements)
struct BlockSortArgs {
    const uint64_t *in; uint64_t *out;
    const unsigned long long *list;     // (start, end) of every segment of this size class
    const unsigned int *count;          // device-side number of segments
    int shiftHi, ign;                   // as SortArgs: all elements of a segment agree above shiftHi; bits below ign ride along
};
// one block per listed segment of at most 512 WAVES elements: out = in with the segment stably sorted on bits [ign, shiftHi)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_block_sort(BlockSortArgs a) {
    constexpr int R = 8, N = 64 * R * WAVES, NT = 64 * WAVES;
    __shared__ uint64_t sKeys[N];
    __shared__ uint64_t sX[N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lowMask = (1ull << a.shiftHi) - 1ull;
    const unsigned int cnt = *a.count;
    for (unsigned int q = blockIdx.x; q < cnt; q += gridDim.x) {
        const uint64_t s = a.list[2 * (size_t) q];
        const int n = (int) min((unsigned long long) N, a.list[2 * (size_t) q + 1] - s);     // (longer ones are not listed here)
        for (int i = threadIdx.x; i < n; i += NT) sKeys[i] = a.in[s + (uint64_t) i];
        __syncthreads();
        uint64_t v[R];
        // network position (wave, lane, r) starts with input element r * NT + wave * 64 + lane: conflict-free LDS reads; the
        // element index is the low part of the word, so the order of equal keys is their input order whatever the start
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = r * NT + wave * 64 + lane;
            v[r] = (i < n) ? ((((sKeys[i] & lowMask) >> a.ign) << BLK_IDX) | (uint64_t) i) : ~0ull;
        }
        bitonicBlock<WAVES, uint64_t>(v, lane, wave, sX);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int p = wave * 64 * R + lane * R + r;
            if (p < n) sX[p] = sKeys[(int) (v[r] & ((1u << BLK_IDX) - 1u))];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += NT) a.out[s + (uint64_t) i] = sX[i];
        __syncthreads();
    }
}

// copies the listed ranges between the array and a dense staging buffer (ranges sorted by start, off = prefix sums of sizes)
template <typename T, bool GATHER>
__global__ __launch_bounds__(256) void k_big_copy(const unsigned long long *__restrict__ ranges /* start, end, off */, unsigned int cnt, T *arr, T *dense) {
    // a wave per range (the ranges are a few hundred to a few thousand elements each, there can be a million of them)
    const unsigned int lane = threadIdx.x & 63, wavesPerGrid = gridDim.x * 4;
    for (unsigned int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < cnt; r += wavesPerGrid) {
        const unsigned long long s = ranges[3 * (size_t) r], e = ranges[3 * (size_t) r + 1], o = ranges[3 * (size_t) r + 2];
        for (unsigned long long i = lane; i < e - s; i += 64) {
            if (GATHER) dense[o + i] = arr[s + i]; else arr[s + i] = dense[o + i];
        }
    }
}

inline unsigned int bigCopyGrid(unsigned int cnt) { return std::min<unsigned int>((cnt + 3) / 4, 1u << 16); }

// (start, end) list in append order -> ranges sorted by start with their offsets in a dense array: radix sort (radix.h) by start, sizes,
// exclusive scan.  Everything stays on the device except the total (and the first start, which the k-mer path wants).
__global__ void k_big_sizes(const unsigned long long *__restrict__ st, const unsigned long long *__restrict__ en, unsigned int cnt, unsigned long long *__restrict__ sz) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) sz[i] = en[i] - st[i]; else if (i == cnt) sz[i] = 0;
}
__global__ void k_big_pack(const unsigned long long *__restrict__ st, const unsigned long long *__restrict__ en, const unsigned long long *__restrict__ off, unsigned int cnt,
                           unsigned long long *__restrict__ ranges) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) { ranges[3 * (size_t) i] = st[i]; ranges[3 * (size_t) i + 1] = en[i]; ranges[3 * (size_t) i + 2] = off[i]; }
}
__global__ void k_big_split(const unsigned long long *__restrict__ list, unsigned int cnt, unsigned long long *__restrict__ st, unsigned long long *__restrict__ en) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) { st[i] = list[2 * (size_t) i]; en[i] = list[2 * (size_t) i + 1]; }
}
inline int loadBigList(hipStream_t s, const unsigned long long *bigList, unsigned int cnt, DevBuf<unsigned long long> &ranges, uint64_t &total,
                       unsigned long long *firstStart = nullptr) {
    DevBuf<unsigned long long> s0, s1, e0, e1, sz, off;
    if (!s0.alloc(cnt) || !s1.alloc(cnt) || !e0.alloc(cnt) || !e1.alloc(cnt) || !sz.alloc((size_t) cnt + 1) || !off.alloc((size_t) cnt + 1) || !ranges.alloc(3 * (size_t) cnt)) return CDM_ERR_HIP;
    const unsigned int g = (cnt + 256) / 256;
    hipLaunchKernelGGL(k_big_split, dim3(g), dim3(256), 0, s, bigList, cnt, s0.p, e0.p);
    bool inFirst = true;
    if (int rc = rx::sortPairs<unsigned long long, unsigned long long>(s, 256, s0.p, s1.p, e0.p, e1.p, (uint64_t) cnt, 0, 64, inFirst)) return rc;
    struct { unsigned long long *c; unsigned long long *current() const { return c; } } ks{inFirst ? s0.p : s1.p}, vs{inFirst ? e0.p : e1.p};
    cdmscan::ScanTemp scanTmp;                                                // alive until the synchronise below
    hipLaunchKernelGGL(k_big_sizes, dim3(g), dim3(256), 0, s, (const unsigned long long *) ks.current(), (const unsigned long long *) vs.current(), cnt, sz.p);
    if (int rc = cdmscan::exclusiveScan<unsigned long long>(s, scanTmp, sz.p, off.p, (size_t) cnt + 1)) return rc;
    hipLaunchKernelGGL(k_big_pack, dim3(g), dim3(256), 0, s, (const unsigned long long *) ks.current(), (const unsigned long long *) vs.current(), (const unsigned long long *) off.p, cnt, ranges.p);
    unsigned long long tot = 0, first = ~0ull;
    hipMemcpyAsync(&tot, off.p + cnt, 8, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(&first, ks.current(), 8, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) return CDM_ERR_HIP;
    total = tot;
    if (firstStart) *firstStart = cnt ? first : ~0ull;
    return CDM_OK;
}

// CDM_BUCKET_CAP=<largest bucket>[,<owned range>] lowers the capacities (tests: reach the big-bucket path on small inputs)
inline void capacities(int &own, uint32_t &maxBucket) {
    own = WV_OWN; maxBucket = BK_MAXB;
    if (const char *e = cdmGetenv("CDM_BUCKET_CAP")) {
        const long m = atol(e);
        if (m >= 1 && m <= BK_MAXB) maxBucket = (uint32_t) m;
        if (const char *comma = strchr(e, ',')) { const long v = atol(comma + 1); if (v >= 1 && v <= WV_OWN) own = (int) v; }
    }
}
inline size_t bigListSlots(uint64_t n, uint32_t maxBucket) { return 2 * (size_t) (n / ((uint64_t) maxBucket + 1) + 2); }

// `in` is stably sorted on key bits [shiftHi, top); afterwards `out` is stably sorted on [ign, top).  Synchronises the stream.
inline int bucketSortKeys(hipStream_t s, const uint64_t *in, uint64_t *out, uint64_t n, int shiftHi, int ign, int top) {
    if (n == 0) return CDM_OK;
    int own; uint32_t maxBucket; capacities(own, maxBucket);
    DevBuf<unsigned long long> bigList; DevBuf<unsigned int> bigCnt;
    if (!bigList.alloc(bigListSlots(n, maxBucket)) || !bigCnt.alloc(1)) return CDM_ERR_HIP;
    hipMemsetAsync(bigCnt.p, 0, 4, s);
    SortArgs a; a.in = in; a.out = out; a.n = n; a.shiftHi = shiftHi; a.ign = ign; a.own = own; a.maxBucket = maxBucket; a.big.list = bigList.p; a.big.cnt = bigCnt.p;
    const uint64_t perBlock = (uint64_t) own * BK_WAVES;
    hipLaunchKernelGGL(k_bucket_sort, dim3((unsigned) ((n + perBlock - 1) / perBlock)), dim3(BK_NT), 0, s, a);
    unsigned int cnt = 0;
    if (hipMemcpyAsync(&cnt, bigCnt.p, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return CDM_ERR_HIP;
    if (cnt == 0) return CDM_OK;
    DevBuf<unsigned long long> ranges; uint64_t total = 0;
    if (int rc = loadBigList(s, bigList.p, cnt, ranges, total)) return rc;
    if (cdmGetenv("CDM_BUCKET_STATS")) fprintf(stderr, "bucketSortKeys: n %llu shiftHi %d: %u big buckets, %llu elements\n", (unsigned long long) n, shiftHi, cnt, (unsigned long long) total);
    DevBuf<uint64_t> d0, d1;
    if (!d0.alloc(total) || !d1.alloc(total)) return CDM_ERR_HIP;
    const unsigned int grid = bigCopyGrid(cnt);
    hipLaunchKernelGGL((k_big_copy<uint64_t, true>), dim3(grid), dim3(256), 0, s, (const unsigned long long *) ranges.p, cnt, const_cast<uint64_t *>(in), d0.p);
    bool inFirst = true;
    if (int rc = rx::sortKeys<uint64_t>(s, 256, d0.p, d1.p, total, ign, top, inFirst)) return rc;
    hipLaunchKernelGGL((k_big_copy<uint64_t, false>), dim3(grid), dim3(256), 0, s, (const unsigned long long *) ranges.p, cnt, out, inFirst ? d0.p : d1.p);
    if (hipStreamSynchronize(s) != hipSuccess) return CDM_ERR_HIP;
    return CDM_OK;
}

}  // namespace bucket
