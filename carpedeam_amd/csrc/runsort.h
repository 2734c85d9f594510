// kmermatcher's second sort (kmermatcher.cpp:431: the group tuples by (representative, member id, diagonal)) without global
// radix passes over the tuples.
//
// After assignGroup the tuples sit in k-mer order, and every k-mer run is a stretch of tuples with ONE representative: the
// array is run-length compressible on the very field the sort starts with.  So the runs are sorted, not the tuples:
//   1. k_run_records               one record (rep, start, length) per maximal stretch of kept tuples with the same
//                                  representative (a stretch never crosses a 4096-tuple tile, so a length fits 13 bits);
//                                  12 bytes per ~16 tuples.  One pass (chained scan over the tiles); k_run_count + k_run_write
//                                  if the record buffer turns out too small
//   2. a stable radix sort of the records by rep (radix.h; they are written in k-mer order): 1/12 of the tuples' bytes per pass
//   3. expanding the sorted records makes the tuples of a representative contiguous, still in k-mer order (what the reference's
//      stable sort leaves for equal (rep, id, diagonal)); dropped tuples (~0) vanish on the way, so this is also the compaction.
//      k_run_gather does it for the whole array (multi-GPU hand-off); on one device the expansion happens inside step 4
//   4. segmentedSortKeys           each representative's segment is sorted on (id, diagonal): k_unit_sort (expands the unit's
//                                  records into LDS, an LDS counting split on a monotone function of (rep, id), then register
//                                  networks per group of sub-buckets) for segments up to 2048 tuples, a block-wide bitonic
//                                  network (k_block_sort) up to 4096, the global radix sort beyond that (deep pile-ups).
// The tuple array is read twice and written once here (records; local sort), against 10 reads and 9 writes of the 4 radix
// passes + bucket finish this replaces.
#pragma once
#include "bucket.h"
#include "devutil.h"

namespace runsort {

#ifndef CDM_RUN_ITEMS
#define CDM_RUN_ITEMS 16
#endif
constexpr int RUN_ITEMS = CDM_RUN_ITEMS, RUN_NT = 256, RUN_TILE = RUN_NT * RUN_ITEMS;      // 4096 tuples per tile (2048 is slower: 124 instead of 117 ms for sort 2)
static_assert(RUN_ITEMS == 8 || RUN_ITEMS == 16, "the boundary bits of a thread are one byte or one 16-bit word");
template <int N> struct RunBitsOf { typedef uint16_t T; };
template <> struct RunBitsOf<8> { typedef uint8_t T; };
typedef RunBitsOf<RUN_ITEMS>::T RunBits;
constexpr int RUN_CNT_BITS = 13;                                                 // a record's length <= RUN_TILE

// Group keys come in two forms (kmermatch.hip, packGroupKey).  Narrow: [rep | id | diagonal | strand] in one word - while that fits 63
// bits.  Wide (any DB: 2^32 sequences, contigs of millions of letters): [START | DROPPED | id | diagonal | strand] - the representative
// is not in the member's key.  It does not have to be: the first tuple of a k-mer run IS the representative's own tuple, so the first
// slot of every run with members carries GK_START and, in its id field, the representative - with GK_DROPPED if that tuple itself is
// not kept (a self tuple under --include-only-extendable).  A slot's representative is the id of the nearest START at or in front
// of it; from the run records on, the representative lives in the records (recRep / segRep), never in a key.
constexpr uint64_t GK_START = 1ull << 62, GK_DROPPED = 1ull << 63;
__host__ __device__ __forceinline__ bool gkKept(uint64_t k) { return (k >> 63) == 0ull; }      // (~0 and a dropped run start are not; narrow keys end below bit 63)
struct RunArgs {
    const uint64_t *keys;       // group keys in k-mer order, ~0 = dropped / unused slot
    uint64_t n;                 // slots
    uint64_t skipLo, skipHi;    // [skipLo, skipHi) holds only unused slots (the tail of region 1): not read
    uint64_t base = 0;          // added to a record's start: `keys` is a part of the array the records index (round 5: region 2 alone, behind the
                                // records the grouping kernel staged for region 1)
    int repShift;               // narrow: rep = key >> repShift
    int wide = 0, idShift = 0; uint64_t idMask = 0;        // wide: id = (key >> idShift) & idMask
};
// LDS image of a tile, one pad per 16 items (thread t walks items 16 t .. 16 t + 15 without bank conflicts)
__device__ __forceinline__ int runPad(int i) { return i + (i >> 4); }
constexpr int RUN_LDS = RUN_TILE + RUN_TILE / 16 + 1;

// loads the tile's representatives into LDS (coalesced; 0xFFFFFFFF = no kept tuple in the slot - 32 bits per slot instead of the key:
// half the LDS, twice the tiles in flight) and returns, for the 16 items of this thread, the bit masks "starts a record" and
// "ends the record in front of it" (= starts one, or is not kept)
constexpr uint32_t RUN_NONE = 0xFFFFFFFFu;
// wide keys: the tile's representatives from its START slots.  sRep gets the id of every START slot first (RUN_NONE elsewhere), sKept
// one bit per slot; every thread then walks its 16 consecutive slots with the id of the last START in front of them (a scan over the
// threads' last ids: wave shuffles, then the four wave tails), and what the tile's first slots need from the tiles in front - the run
// they belong to began there - is looked up in memory by wave 0: backwards from the tile, 64 slots at a time, to the nearest START
// (a run's length away; only when a kept slot stands in front of the tile's first START).
__device__ __forceinline__ void runRepsWide(const RunArgs &a, uint64_t base, bool skip, uint32_t *sRep) {
    __shared__ unsigned long long sKept[RUN_TILE / 64];
    __shared__ uint32_t sWaveLast[RUN_NT / 64], sCarry;
    __shared__ int sNeed;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { sNeed = 0; sCarry = RUN_NONE; }
#pragma unroll
    for (int j = 0; j < RUN_ITEMS; j++) {
        const int li = threadIdx.x + RUN_NT * j; const uint64_t i = base + (uint64_t) li;
        const uint64_t k = (!skip && i < a.n && !(i >= a.skipLo && i < a.skipHi)) ? a.keys[i] : ~0ull;
        const bool start = k != ~0ull && (k & GK_START) != 0ull;
        sRep[runPad(li)] = start ? (uint32_t) ((k >> a.idShift) & a.idMask) : RUN_NONE;
        const unsigned long long kb = __ballot(gkKept(k));
        if (lane == 0) sKept[(RUN_NT * j + 64 * wave) >> 6] = kb;
    }
    __syncthreads();
    // last START id among this thread's slots, then among the slots of the threads in front
    uint32_t mine = RUN_NONE;
#pragma unroll
    for (int j = 0; j < RUN_ITEMS; j++) { const uint32_t r = sRep[runPad(threadIdx.x * RUN_ITEMS + j)]; if (r != RUN_NONE) mine = r; }
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t) __shfl_up((int) incl, d, 64); if (lane >= d && incl == RUN_NONE) incl = o; }
    if (lane == 63) sWaveLast[wave] = incl;
    uint32_t carry = (uint32_t) __shfl_up((int) incl, 1, 64);
    if (lane == 0) carry = RUN_NONE;
    __syncthreads();
    for (int w = wave - 1; w >= 0 && carry == RUN_NONE; w--) carry = sWaveLast[w];
    // the walk: START slots keep their id, kept slots behind them take it, everything else is RUN_NONE
    uint32_t cur = carry; bool need = false;
    const unsigned int keptBits = (unsigned int) ((sKept[(threadIdx.x * RUN_ITEMS) >> 6] >> ((threadIdx.x * RUN_ITEMS) & 63)) & ((1u << RUN_ITEMS) - 1u));
    uint32_t out[RUN_ITEMS];
#pragma unroll
    for (int j = 0; j < RUN_ITEMS; j++) {
        const uint32_t r = sRep[runPad(threadIdx.x * RUN_ITEMS + j)];
        if (r != RUN_NONE) cur = r;
        const bool kept = (keptBits >> j) & 1u;
        out[j] = kept ? cur : RUN_NONE;
        need |= kept && cur == RUN_NONE;
    }
    if (need) sNeed = 1;
    __syncthreads();
    if (sNeed) {      // (block-uniform) a kept slot in front of the tile's first START: its run starts in a tile in front
        if (wave == 0) {
            uint64_t hi = base;                                  // slots [hi - 64, hi) are looked at next
            uint32_t found = RUN_NONE;
            while (hi > 0) {
                const uint64_t i = hi - 1 - (uint64_t) lane;
                const uint64_t k = (hi > (uint64_t) lane) ? a.keys[i] : ~0ull;
                const unsigned long long sm = __ballot(k != ~0ull && (k & GK_START) != 0ull);
                if (sm) { const int l = __ffsll(sm) - 1; found = (uint32_t) ((bucket::readLane64(k, l) >> a.idShift) & a.idMask); break; }
                hi = hi > 64 ? hi - 64 : 0;
            }
            if (lane == 0) sCarry = found;
        }
        __syncthreads();
        const uint32_t c = sCarry;
#pragma unroll
        for (int j = 0; j < RUN_ITEMS; j++) if (((keptBits >> j) & 1u) && out[j] == RUN_NONE) out[j] = c;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RUN_ITEMS; j++) sRep[runPad(threadIdx.x * RUN_ITEMS + j)] = out[j];
}
__device__ __forceinline__ void runFlags(const RunArgs &a, uint64_t base, uint32_t *sRep, unsigned int &startBits, unsigned int &boundBits) {
    const bool skip = base >= a.skipLo && base + RUN_TILE <= a.skipHi;      // block-uniform
    if (a.wide) runRepsWide(a, base, skip, sRep);
    else {
#pragma unroll
    for (int j = 0; j < RUN_ITEMS; j++) {
        const int li = threadIdx.x + RUN_NT * j; const uint64_t i = base + (uint64_t) li;
        const uint64_t k = (!skip && i < a.n && !(i >= a.skipLo && i < a.skipHi)) ? a.keys[i] : ~0ull;
        sRep[runPad(li)] = k != ~0ull ? (uint32_t) (k >> a.repShift) : RUN_NONE;
    }
    }
    __syncthreads();
    startBits = 0; boundBits = 0;
    uint32_t prev = (threadIdx.x == 0) ? RUN_NONE : sRep[runPad(threadIdx.x * RUN_ITEMS - 1)];     // a record never crosses a tile
#pragma unroll
    for (int j = 0; j < RUN_ITEMS; j++) {
        const uint32_t r = sRep[runPad(threadIdx.x * RUN_ITEMS + j)];
        const bool kept = r != RUN_NONE;
        const bool start = kept && r != prev;
        if (start) startBits |= 1u << j;
        if (start || !kept) boundBits |= 1u << j;
        prev = r;
    }
}
__global__ __launch_bounds__(RUN_NT) void k_run_count(RunArgs a, unsigned long long *__restrict__ tileCnt) {
    __shared__ uint32_t sKeys[RUN_LDS];
    unsigned int sb, bb;
    runFlags(a, (uint64_t) blockIdx.x * RUN_TILE, sKeys, sb, bb);
    const unsigned int tot = cdm_block_sum<unsigned int>((unsigned int) __popc(sb));
    if (threadIdx.x == 0) tileCnt[blockIdx.x] = tot;
}
// record j: recRep[j] = representative, recVal[j] = start << 13 | length
__global__ __launch_bounds__(RUN_NT) void k_run_write(RunArgs a, const unsigned long long *__restrict__ tileOff, uint32_t *__restrict__ recRep, uint64_t *__restrict__ recVal) {
    __shared__ uint32_t sKeys[RUN_LDS];
    __shared__ __align__(8) RunBits sBound[RUN_NT + 32 / sizeof(RunBits)];
    const uint64_t base = (uint64_t) blockIdx.x * RUN_TILE;
    unsigned int sb, bb;
    runFlags(a, base, sKeys, sb, bb);
    sBound[threadIdx.x] = (RunBits) bb;
    if (threadIdx.x < 32 / sizeof(RunBits)) sBound[RUN_NT + threadIdx.x] = 0;
    unsigned int tot;
    unsigned long long rank = tileOff[blockIdx.x] + cdm_block_excl_sum<unsigned int>((unsigned int) __popc(sb), tot);     // (its barriers publish sBound)
    const unsigned long long *words = reinterpret_cast<const unsigned long long *>(sBound);
#pragma unroll 1
    while (sb) {
        const int j = __ffs(sb) - 1;
        sb &= sb - 1;
        const int li = threadIdx.x * RUN_ITEMS + j;
        int e = RUN_TILE;                                        // end = next boundary behind li (tile end if none)
        for (int w = (li + 1) >> 6; w < RUN_TILE / 64; w++) {
            unsigned long long m = words[w];
            if (w == ((li + 1) >> 6)) m &= ~0ull << ((li + 1) & 63);
            if (m) { e = w * 64 + __ffsll(m) - 1; break; }
        }
        recRep[rank] = sKeys[runPad(li)];
        recVal[rank] = ((a.base + base + (uint64_t) li) << RUN_CNT_BITS) | (uint64_t) (e - li);
        rank++;
    }
}
// ---- count and write in ONE pass over the keys: a tile learns where its records go from the tiles in front of it through a
// chained scan with decoupled look-back (status word per tile: 2 bits of state | 62 bits of count - the word is the whole message,
// relaxed atomics do; tiles are numbered by a ticket
// taken at start, so a tile only ever waits for tiles that already run).  cap = room in recRep / recVal: a tile that would write
// beyond it raises the flag and writes nothing (the caller falls back to k_run_count / k_run_write with exact sizes).
constexpr unsigned long long RS_AGG = 1ull << 62, RS_PREFIX = 2ull << 62, RS_MASK = (1ull << 62) - 1ull;
struct RunScan { unsigned long long *status; unsigned int *ticket; unsigned long long *total; unsigned int *overflow; unsigned long long cap; unsigned long long tiles; };
__global__ __launch_bounds__(RUN_NT) void k_run_records(RunArgs a, RunScan sc, uint32_t *__restrict__ recRep, uint64_t *__restrict__ recVal) {
    __shared__ uint32_t sKeys[RUN_LDS];
    __shared__ __align__(8) RunBits sBound[RUN_NT + 32 / sizeof(RunBits)];
    __shared__ unsigned int sTile;
    __shared__ unsigned long long sPrefix;
    if (threadIdx.x == 0) sTile = atomicAdd(sc.ticket, 1u);
    __syncthreads();
    const unsigned long long tile = sTile;
    const uint64_t base = (uint64_t) tile * RUN_TILE;
    unsigned int sb, bb;
    runFlags(a, base, sKeys, sb, bb);
    sBound[threadIdx.x] = (RunBits) bb;
    if (threadIdx.x < 32 / sizeof(RunBits)) sBound[RUN_NT + threadIdx.x] = 0;
    unsigned int tot;
    const unsigned int ex = cdm_block_excl_sum<unsigned int>((unsigned int) __popc(sb), tot);        // (its barriers publish sBound)
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned long long excl = 0;
        if (tile == 0) { if (lane == 0) __hip_atomic_store(&sc.status[0], RS_PREFIX | (unsigned long long) tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else {
            if (lane == 0) __hip_atomic_store(&sc.status[tile], RS_AGG | (unsigned long long) tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long t0 = (long long) tile - 1;                    // lane l looks at tile t0 - l
            while (true) {
                const long long t = t0 - lane;
                unsigned long long v = RS_PREFIX;                   // in front of tile 0: nothing
                if (t >= 0) do { v = __hip_atomic_load(&sc.status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((v >> 62) == 0ull);
                const unsigned long long pm = __ballot((v >> 62) == 2ull);
                const int first = pm ? __ffsll(pm) - 1 : 63;       // the nearest tile with a full prefix closes the sum
                const unsigned long long part = cdm_wave_incl_sum<unsigned long long>(lane <= first ? (v & RS_MASK) : 0ull);
                excl += (unsigned long long) __shfl((long long) part, 63, 64);
                if (pm) break;
                t0 -= 64;
            }
            if (lane == 0) __hip_atomic_store(&sc.status[tile], RS_PREFIX | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            sPrefix = excl;
            if (tile + 1 == sc.tiles) *sc.total = excl + tot;
            if (excl + tot > sc.cap) *sc.overflow = 1u;
        }
    }
    __syncthreads();
    unsigned long long rank = sPrefix + ex;
    if (sPrefix + tot > sc.cap) return;
    const unsigned long long *words = reinterpret_cast<const unsigned long long *>(sBound);
#pragma unroll 1
    while (sb) {
        const int j = __ffs(sb) - 1;
        sb &= sb - 1;
        const int li = threadIdx.x * RUN_ITEMS + j;
        int e = RUN_TILE;                                        // end = next boundary behind li (tile end if none)
        for (int w = (li + 1) >> 6; w < RUN_TILE / 64; w++) {
            unsigned long long m = words[w];
            if (w == ((li + 1) >> 6)) m &= ~0ull << ((li + 1) & 63);
            if (m) { e = w * 64 + __ffsll(m) - 1; break; }
        }
        recRep[rank] = sKeys[runPad(li)];
        recVal[rank] = ((a.base + base + (uint64_t) li) << RUN_CNT_BITS) | (uint64_t) (e - li);
        rank++;
    }
}
struct RunLen { const uint64_t *recVal; __device__ __forceinline__ unsigned long long operator()(size_t j) const { return recVal[j] & ((1ull << RUN_CNT_BITS) - 1ull); } };

// The tuples of the sorted records [j0, j0 + 64) that start in front of `end` (gathered coordinates): put(offset behind dst[j0],
// tuple).  The reads are the records' 8-byte x length stretches of `keys`.  Returns the number of tuples (wave-uniform).
template <typename Put>
__device__ __forceinline__ unsigned int waveGatherRecords(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ recVal, const unsigned long long *__restrict__ dst,
                                                          uint64_t nRec, uint64_t j0, unsigned long long end, int lane, uint32_t *sOff, uint64_t *sStart, const Put &put) {
    const uint64_t j = j0 + lane;
    const bool valid = j < nRec && dst[j] < end;
    const uint64_t rv = valid ? recVal[j] : 0ull;
    const unsigned int cnt = (unsigned int) (rv & ((1ull << RUN_CNT_BITS) - 1ull));
    const unsigned int incl = cdm_wave_incl_sum<unsigned int>(cnt);
    sOff[lane] = incl - cnt; sStart[lane] = rv >> RUN_CNT_BITS;
    const unsigned int total = (unsigned int) __shfl((int) incl, 63, 64);
    bucket::waveLdsSync();
    for (unsigned int e = lane; e < total; e += 64) {
        int r = 0;                                               // last record with sOff[r] <= e (empty padding records have cnt 0)
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) if (sOff[r + s] <= e) r += s;
        put(e, keys[sStart[r] + (e - sOff[r])]);
    }
    bucket::waveLdsSync();
    return total;
}
// expands the sorted records: out[dst[j] ..] = the tuples of record j.  A wave takes 64 consecutive records; their tuples
// are consecutive in `out`, so the writes are whole lines.
__global__ __launch_bounds__(256) void k_run_gather(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ recVal, const unsigned long long *__restrict__ dst,
                                                    uint64_t nRec, uint64_t *__restrict__ out) {
    __shared__ uint32_t sOffAll[4][64];
    __shared__ uint64_t sStartAll[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t j0 = ((uint64_t) blockIdx.x * 4 + wave) * 64;
    if (j0 >= nRec) return;      // (whole wave)
    const unsigned long long d0 = dst[j0];
    waveGatherRecords(keys, recVal, dst, nRec, j0, ~0ull, lane, sOffAll[wave], sStartAll[wave], [&](unsigned int e, uint64_t t) { out[d0 + e] = t; });
}
// the same for listed ranges [s, e) of the gathered array only (both record boundaries): one block per range
__global__ __launch_bounds__(256) void k_gather_ranges(const unsigned long long *__restrict__ list, const unsigned int *__restrict__ count, const uint64_t *__restrict__ keys,
                                                       const uint64_t *__restrict__ recVal, const unsigned long long *__restrict__ dst, uint64_t nRec, uint64_t *__restrict__ out) {
    __shared__ uint32_t sOffAll[4][64];
    __shared__ uint64_t sStartAll[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int n = *count;
    for (unsigned int item = blockIdx.x; item < n; item += gridDim.x) {
        const unsigned long long s = list[2 * (size_t) item], e = list[2 * (size_t) item + 1];
        uint64_t lo = 0, hi = nRec;                              // first record with dst >= s (it starts at s)
        while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (dst[mid] < s) lo = mid + 1; else hi = mid; }
        for (uint64_t j0 = lo + 64ull * wave; j0 < nRec; j0 += 256) {
            const unsigned long long d0 = dst[j0];
            if (d0 >= e) break;
            waveGatherRecords(keys, recVal, dst, nRec, j0, e, lane, sOffAll[wave], sStartAll[wave], [&](unsigned int x, uint64_t t) { out[d0 + x] = t; });
        }
    }
}

// ---------------------------------------------------------------------------------------------- segments by size class
constexpr int U_T = 1024;             // slots per unit range of the segmented sort (k_unit_sort below)
constexpr int SEG_CLASSES = 4;          // 0..2: k_block_sort with 2, 4, 8 waves; 3: longer (the global radix sort of radix.h)
struct SegListArgs {
    const uint32_t *recRep; const unsigned long long *dst; uint64_t nRec;
    uint32_t maxWave;                   // segments up to this length are finished by k_bucket_sort
    const unsigned long long *uEnd; uint64_t units;       // the units' ends (NULL: there are no units): a segment that lies inside its unit is the unit's
    uint32_t cap[SEG_CLASSES - 1];      // capacities of the block classes
    unsigned long long *list[SEG_CLASSES]; unsigned int *cnt;      // cnt[c]
    const unsigned long long *segFirstRec = nullptr; uint64_t nSeg = 0;      // (k_seg_list_segments) first record of every segment, [nSeg] = nRec
};
// one thread per record; the first record of a representative measures its segment and lists it if a wave cannot sort it
__global__ __launch_bounds__(1024) void k_seg_list(SegListArgs a) {
    const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1; unsigned long long s = 0, e = 0;
    if (j < a.nRec) {
        const uint32_t rep = a.recRep[j];
        if (j == 0 || a.recRep[j - 1] != rep) {
            uint64_t lo = j, step = 1;                          // gallop, then bisect: recRep[lo] == rep, recRep[hi] != rep or hi == nRec
            while (lo + step < a.nRec && a.recRep[lo + step] == rep) { lo += step; step <<= 1; }
            uint64_t hi = min(lo + step, a.nRec);
            while (hi - lo > 1) { const uint64_t mid = lo + ((hi - lo) >> 1); if (a.recRep[mid] == rep) lo = mid; else hi = mid; }
            s = a.dst[j]; e = a.dst[hi];
            const unsigned long long n = e - s;
            // (with the capacities lowered below a unit's range - tests - a segment beyond maxWave can lie in the middle of a unit: it is
            // sorted / aggregated with its unit, and listing it as well would hand the fallback two overlapping ranges)
            const bool inUnit = a.uEnd && s / U_T < a.units && e <= a.uEnd[s / U_T];
            if (n > a.maxWave && !inUnit) cls = n <= a.cap[0] ? 0 : n <= a.cap[1] ? 1 : n <= a.cap[2] ? 2 : 3;
        }
    }
    // (a segment that no unit holds is rare - 49 of 3.9 M at 50 M reads: most blocks have nothing to list, and the four block-wide
    // appends below are twelve barriers)
    if (!__syncthreads_or(cls >= 0)) return;
#pragma unroll
    for (int c = 0; c < SEG_CLASSES; c++) {
        const uint32_t q = cdm_block_append(a.cnt + c, cls == c);
        if (cls == c) { a.list[c][2 * (size_t) q] = s; a.list[c][2 * (size_t) q + 1] = e; }
    }
}

// the same with the segment table at hand (aggvote.h builds one: the aggregated sort 2): a thread per SEGMENT, no search - 8.4 M threads
// with two look-ups each where k_seg_list runs 530 M threads, 8.4 M of which gallop and bisect over the records (3.6 ms at 50 M reads)
__global__ __launch_bounds__(1024) void k_seg_list_segments(SegListArgs a) {
    const uint64_t g = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1; unsigned long long s = 0, e = 0;
    if (g < a.nSeg) {
        s = a.dst[a.segFirstRec[g]]; e = a.dst[a.segFirstRec[g + 1]];
        const unsigned long long n = e - s;
        const bool inUnit = a.uEnd && s / U_T < a.units && e <= a.uEnd[s / U_T];
        if (n > a.maxWave && !inUnit) cls = n <= a.cap[0] ? 0 : n <= a.cap[1] ? 1 : n <= a.cap[2] ? 2 : 3;
    }
    if (!__syncthreads_or(cls >= 0)) return;
#pragma unroll
    for (int c = 0; c < SEG_CLASSES; c++) {
        const uint32_t q = cdm_block_append(a.cnt + c, cls == c);
        if (cls == c) { a.list[c][2 * (size_t) q] = s; a.list[c][2 * (size_t) q + 1] = e; }
    }
}

// ---------------------------------------------------------------------------------------------- k_unit_sort
// The segmented sort proper.  The array is cut into ranges of U_T slots; a block owns the segments (representatives) that START
// in its range - its "unit", at most U_CAP tuples, all staged in LDS.  Representatives are replaced by their ordinal in the unit
// (a prefix popcount over the segment-start bits), so a tuple's sort key is h = (ordinal, id), diagonal - a few dozen bits
// whatever the ids of the representatives are.  The unit is first split on the top bits of h into 256 sub-buckets with an LDS
// counting sort (a representative with 2000 tuples has members spread over the whole id range: ~8 tuples per sub-bucket), then
// every wavefront sorts its 64 sub-buckets (as one group if they hold at most 512 tuples, else in groups of consecutive
// sub-buckets) with the register network of bucket.h on words (h, diagonal, index in the unit): concatenated, the sorted groups
// are the sorted unit, ties in input order.  A segment of more than maxSeg = U_CAP - U_T tuples never fits for sure and is left
// to the caller (k_seg_list lists it); a unit with a sub-bucket beyond 512 tuples goes to the `hard` list (sorted by the global radix sort).
#ifndef CDM_U_NT
#define CDM_U_NT 512
#endif
// eight waves share one unit's LDS.  The kernel runs on the latency of its LDS / global round trips - its time goes with
// 1 / (units in flight per CU) - so the units are listed by size beforehand (k_unit_bounds, k_unit_classes) and every size class
// runs the instance whose LDS is just large enough.
#ifndef CDM_U_NT0
#define CDM_U_NT0 256
#define CDM_U_NT1 256
#endif
constexpr int U_MAXSEG = 2048, U_CAP = U_T + U_MAXSEG, U_NB = 256, U_IDXB = 12, U_ORDB = 11;
constexpr int U_CLASSES = 3;
constexpr int U_CLASS_CAP[U_CLASSES] = {1536, 2048, U_CAP};
constexpr int U_CLASS_NT[U_CLASSES] = {CDM_U_NT0, CDM_U_NT1, CDM_U_NT};
static_assert(U_CAP <= (1 << U_IDXB) && U_T <= (1 << U_ORDB), "unit sorter geometry");

// ---- the units: unit u = the segments that start in [u U_T, (u + 1) U_T), i.e. the tuples [uStart[u], uStart[u + 1]) - minus its
// last segment if that one is longer than maxSeg (then it ends at uEnd[u], the segment is listed by k_seg_list).
struct UnitBoundArgs {
    const uint32_t *recRep; const unsigned long long *dst; uint64_t nRec;
    uint32_t maxSeg; uint64_t units;
    unsigned long long *uStart, *uEnd, *uRec;       // [units + 1], [units], [units + 1]: first tuple / end / first record of a unit
    const uint32_t *segOfRec = nullptr; const unsigned long long *segFirstRec = nullptr;     // the segment table, if there is one: no search for a segment's ends
};
// one thread per record; a record with a unit boundary B in (dst[r], dst[r + 1]] finds the end (and, if need be, the start) of its
// segment: the unit that begins at B starts where the segment ends
__global__ __launch_bounds__(1024) void k_unit_bounds(UnitBoundArgs a) {
    const uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.nRec) return;
    const unsigned long long d0 = a.dst[r], d1 = a.dst[r + 1];
    const uint64_t uLo = d0 / U_T + 1, uHi = (r == a.nRec - 1) ? a.units : d1 / U_T;       // (the end of the array closes the last unit)
    if (r == 0) { a.uStart[0] = 0; a.uRec[0] = 0; }
    if (uLo > uHi) return;
    uint64_t hi, first;
    if (a.segFirstRec) { const uint32_t g = a.segOfRec[r]; first = a.segFirstRec[g]; hi = a.segFirstRec[g + 1]; }
    else {
        const uint32_t rep = a.recRep[r];
        uint64_t lo = r, step = 1;                              // last record of the segment: gallop, then bisect
        while (lo + step < a.nRec && a.recRep[lo + step] == rep) { lo += step; step <<= 1; }
        hi = min(lo + step, a.nRec);
        while (hi - lo > 1) { const uint64_t mid = lo + ((hi - lo) >> 1); if (a.recRep[mid] == rep) lo = mid; else hi = mid; }
        first = r; uint64_t out = ~0ull, st = 1;                // first record of the segment: `first` is inside, `out` outside (or -1)
        while (first >= st) { if (a.recRep[first - st] == rep) { first -= st; st <<= 1; } else { out = first - st; break; } }
        while (first - out > 1) { const uint64_t mid = out + ((first - out) >> 1); if (a.recRep[mid] == rep) first = mid; else out = mid; }
    }
    const unsigned long long e = a.dst[hi];
    const unsigned long long sSeg = a.dst[first];
    for (uint64_t u = uLo; u <= uHi; u++) {
        if (u <= a.units) { a.uStart[u] = e; a.uRec[u] = hi; }
        // the unit in front of boundary u: this segment is its last one if it starts in that unit's range
        const bool mine = sSeg >= (u - 1) * (unsigned long long) U_T;
        if (u >= 1 && u - 1 < a.units) a.uEnd[u - 1] = (mine && e - sSeg > a.maxSeg) ? sSeg : e;
    }
}
struct UnitClassArgs {
    const unsigned long long *uStart, *uEnd, *uRec; uint64_t units, n;
    uint32_t cap[U_CLASSES];
    unsigned long long *list[U_CLASSES]; unsigned int *cnt;        // (start, end, first record) per unit of the class
};
__global__ __launch_bounds__(1024) void k_unit_classes(UnitClassArgs a) {
    const uint64_t u = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1; unsigned long long s = 0, e = 0, r0 = 0;
    if (u < a.units) {
        s = a.uStart[u]; e = a.uEnd[u]; r0 = a.uRec[u];
        const unsigned long long m = e > s ? e - s : 0;
        if (m) cls = m <= a.cap[0] ? 0 : m <= a.cap[1] ? 1 : 2;
    }
#pragma unroll
    for (int c = 0; c < U_CLASSES; c++) {
        const uint32_t q = cdm_block_append(a.cnt + c, cls == c);
        if (cls == c) { a.list[c][3 * (size_t) q] = s; a.list[c][3 * (size_t) q + 1] = e; a.list[c][3 * (size_t) q + 2] = r0; }
    }
}

struct UnitArgs {
    const uint64_t *keys; const uint64_t *recVal; const unsigned long long *dst; uint64_t nRec;     // the k-mer-ordered keys and the sorted records: a unit is gathered straight from them
    uint64_t *out; uint64_t n;
    int repShift;               // segment id = key >> repShift
    int hiShift;                // (rep, id) = key >> hiShift; the bits below are sorted inside the sub-buckets, bit 0 rides along
    uint32_t maxSub;            // largest sub-bucket a wave finishes (bucket::BK_MAXB; tests lower it to reach the hard path)
    const unsigned long long *list; const unsigned int *count;     // the units of this size class
    bucket::BigList hard;
};
// start (gathered coordinates) of record j, the end of everything if there is none
__device__ __forceinline__ unsigned long long j0Next(const unsigned long long *__restrict__ dst, uint64_t nRec, uint64_t j) { return dst[min(j, nRec)]; }
#ifndef CDM_U_MINW
#define CDM_U_MINW 5
#endif
// (waves per SIMD the register allocation leaves room for: without the bound the kernel takes 118 VGPRs - 4 blocks per CU - although
// its LDS would let 7 run; measured for sort 2: 5 -> 102 ms, 6 -> 105, 7 -> 106, 8 -> 111, unbounded 113)
template <int CAP, int U_NT>
__global__ __launch_bounds__(U_NT, CDM_U_MINW) void k_unit_sort(UnitArgs a) {
    constexpr int ROUNDS = (CAP + U_NT - 1) / U_NT, U_WAVES = U_NT / 64;
    static_assert(U_NB <= U_NT && U_NB % U_WAVES == 0 && CAP % U_NT == 0, "unit sorter geometry");
    __shared__ uint64_t sKeys[CAP];             // the unit; rewritten with the ordinal in place of the representative
    // the record table of the staging phase (sGOff, sGStart) and the permutation of the sort phase share their LDS: barriers separate the two
    constexpr int SHARE_BYTES = (CAP * 2 > U_NT * 12) ? CAP * 2 : U_NT * 12;
    __shared__ __align__(8) unsigned char sShare[SHARE_BYTES];
    uint16_t *sPerm = reinterpret_cast<uint16_t *>(sShare);
    uint64_t *sGStart = reinterpret_cast<uint64_t *>(sShare);
    uint32_t *sGOff = reinterpret_cast<uint32_t *>(sShare + U_NT * 8);
    __shared__ unsigned long long sBits[ROUNDS * U_NT / 64];
    __shared__ unsigned int sPre[ROUNDS * U_NT / 64];
    __shared__ unsigned int sCnt[U_NB];
    __shared__ unsigned int sOff[U_NB + 1];
    __shared__ uint32_t sRep[U_T + 2];          // representative of the unit's segments by ordinal
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned int nUnits = *a.count;
    for (unsigned int item = blockIdx.x; item < nUnits; item += gridDim.x) {
        const uint64_t base = a.list[3 * (size_t) item];
        const int m = (int) (a.list[3 * (size_t) item + 1] - base);          // 1 .. CAP; the unit starts with a segment start
        const uint64_t rec0 = a.list[3 * (size_t) item + 2];
        // stage the unit: its records (whole ones: a unit begins and ends with a segment) expanded straight into LDS - the gathered
        // copy of the array is never written for it - then flag the segment starts (one ballot word per wave and round)
        {
            if (tid < U_NB) sCnt[tid] = 0;
            // U_NT records at a time: their offsets in the unit (dst is the prefix sum already) and source positions go to LDS, then
            // every thread fetches tuples - a binary search over the chunk's offsets finds a tuple's record
            const unsigned long long endG = base + (uint64_t) m;
            for (uint64_t c0 = rec0; c0 < a.nRec; c0 += U_NT) {
                const uint64_t j = c0 + tid;
                const unsigned long long d = j < a.nRec ? a.dst[j] : ~0ull;
                const bool valid = d < endG;
                sGOff[tid] = valid ? (uint32_t) (d - base) : (uint32_t) m;
                sGStart[tid] = valid ? (a.recVal[j] >> RUN_CNT_BITS) : 0ull;
                const int nr = __syncthreads_count(valid);              // the records of this chunk (they are the first nr of it)
                if (nr == 0) break;
                const int e0 = (int) sGOff[0], e1 = (nr == U_NT) ? (int) min((unsigned long long) m, (j0Next(a.dst, a.nRec, c0 + U_NT) - base)) : m;
                for (int e = e0 + tid; e < e1; e += U_NT) {
                    int r = 0;
#pragma unroll
                    for (int st = U_NT / 2; st > 0; st >>= 1) if (r + st < nr && (int) sGOff[r + st] <= e) r += st;
                    sKeys[e] = a.keys[sGStart[r] + (uint64_t) (e - (int) sGOff[r])];
                }
                __syncthreads();
                if (nr < U_NT) break;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROUNDS; r++) {
                const int i = r * U_NT + tid;
                const uint64_t k = i < m ? sKeys[i] : ~0ull, p = (i && i < m) ? sKeys[i - 1] : 0ull;
                const unsigned long long bm = __ballot(i < m && (i == 0 || (k >> a.repShift) != (p >> a.repShift)));
                if (lane == 0) sBits[i >> 6] = bm;
            }
            __syncthreads();
        }
        // ordinal of every segment: number of start bits in front of each word (wave 0), then a masked popcount per tuple
        if (wave == 0) {
            unsigned int run = 0;
#pragma unroll
            for (int w0 = 0; w0 < ROUNDS * U_NT / 64; w0 += 64) {
                const unsigned int c = (w0 + lane < ROUNDS * U_NT / 64) ? (unsigned int) __popcll(sBits[w0 + lane]) : 0u;
                const unsigned int incl = cdm_wave_incl_sum<unsigned int>(c);
                if (w0 + lane < ROUNDS * U_NT / 64) sPre[w0 + lane] = run + incl - c;
                run += __shfl(incl, 63, 64);
            }
        }
        __syncthreads();
        const int idBits = a.repShift - a.hiShift;
        const unsigned int nSeg = sPre[(m - 1) >> 6] + (unsigned int) __popcll(sBits[(m - 1) >> 6] & ((2ull << ((m - 1) & 63)) - 1ull));
        const uint64_t range = ((uint64_t) nSeg << idBits) - 1ull;
        const int bl = range ? 64 - __clzll((long long) range) : 0, sh = max(0, bl - 8) + a.hiShift;       // sub-bucket = t >> sh
        const uint64_t lowRep = (1ull << a.repShift) - 1ull;
        // t = the key with the representative replaced by its ordinal in the unit (bit 0, the strand, still rides along); count
        for (int i = tid; i < m; i += U_NT) {
            const unsigned long long w = sBits[i >> 6];
            const unsigned int o = sPre[i >> 6] + (unsigned int) __popcll(w & ((2ull << (i & 63)) - 1ull)) - 1u;
            const uint64_t k = sKeys[i];
            if ((w >> (i & 63)) & 1ull) sRep[o] = (uint32_t) (k >> a.repShift);
            const uint64_t t = ((uint64_t) o << a.repShift) | (k & lowRep);
            sKeys[i] = t;
            atomicAdd(&sCnt[(unsigned int) (t >> sh)], 1u);
        }
        __syncthreads();
        const unsigned int c = tid < U_NB ? sCnt[tid] : 0u;
        unsigned int tot;
        const unsigned int ex = cdm_block_excl_sum<unsigned int>(c, tot);
        if (tid < U_NB) sOff[tid] = ex;
        if (tid == 0) sOff[U_NB] = tot;
        const bool hard = __syncthreads_or(c > a.maxSub);
        if (hard) { if (tid == 0) a.hard.add(base, base + (uint64_t) m); continue; }       // (block-uniform)
        if (tid < U_NB) sCnt[tid] = ex;             // (cursor of the scatter; the order inside a sub-bucket does not matter: the
        __syncthreads();                            //  word carries the tuple's index)
        for (int i = tid; i < m; i += U_NT) sPerm[atomicAdd(&sCnt[(unsigned int) (sKeys[i] >> sh)], 1u)] = (uint16_t) i;
        __syncthreads();
        // ---- every wave sorts its share of the sub-buckets
        constexpr int PER = U_NB / U_WAVES;
        auto sortRange = [&](int g0, int gm) {
            bucket::sortGroup<uint64_t>(gm, lane,
                [&](int i) { const int e = sPerm[g0 + i]; return ((sKeys[e] >> 1) << U_IDXB) | (uint64_t) e; },
                [&](auto &v) {
                    constexpr int R = sizeof(v) / sizeof(v[0]);
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int p = lane * R + r;
                        if (p < gm) { const uint64_t t = sKeys[(int) (v[r] & ((1u << U_IDXB) - 1u))]; a.out[base + (uint64_t) (g0 + p)] = ((uint64_t) sRep[t >> a.repShift] << a.repShift) | (t & lowRep); }
                    }
                });
        };
        int b = wave * PER;
        const int bLast = b + PER;
        const int w0 = (int) sOff[b], wm = (int) sOff[bLast] - w0;
        if (wm <= bucket::BK_MAXB) { if (wm > 0) sortRange(w0, wm); }
        else while (b < bLast) {
            const int g0 = (int) sOff[b];
            int bE = b + 1;
            while (bE < bLast && (int) sOff[bE + 1] - g0 <= bucket::BK_GROUP) bE++;
            const int gm = (int) sOff[bE] - g0;
            if (gm > 0) sortRange(g0, gm);
            b = bE;
        }
        __syncthreads();        // the next unit reuses the LDS
    }
}

// `in` holds the tuples grouped by representative (bits >= shiftHi), each group in k-mer order; `out` = every group stably
// sorted on bits [1, shiftHi) (bit 0 rides along).  recRep / dst / nRec describe the groups (sorted records and their output
// offsets); hiShift = diagonal bits + 1.  CDM_UNIT_CAP / CDM_BLOCK_CAP lower the capacities (tests).
// srcKeys / recVal: the k-mer-ordered keys and the sorted records `in` is the expansion of.  `in` itself may be unwritten: the unit
// sorter gathers from the records, and the ranges the other sorters need are expanded into `in` here (k_gather_ranges).
// unitHook (aggvote.h): takes the units in k_unit_sort's place - it is handed each size class's list and the hard list it may add to
// wide (keys without the representative, RunArgs): the units go to unitHook (there is no k_unit_sort for such keys), no block sorter;
// what is left - long segments, units the hook hands back - is sorted as (key bits [1, shiftHi), range ordinal) pairs in two stable
// radix sorts, and comes out with the START / DROPPED marks cleared.
// (one block per listed range (start, end, offset in the dense arrays): the range's records expanded straight into the dense arrays -
// every tuple masked to its member bits, with the ordinal of its SEGMENT (a unit holds several representatives) next to it)
__global__ __launch_bounds__(256) void k_gather_ranges_dense(const unsigned long long *__restrict__ ranges, unsigned int cnt, const uint64_t *__restrict__ keys,
                                                             const uint64_t *__restrict__ recVal, const unsigned long long *__restrict__ dst, uint64_t nRec,
                                                             const uint32_t *__restrict__ segOfRec, uint64_t keyMask, uint64_t *__restrict__ dense, uint32_t *__restrict__ ord) {
    __shared__ uint32_t sOffAll[4][64];
    __shared__ uint64_t sStartAll[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *sOff = sOffAll[wave]; uint64_t *sStart = sStartAll[wave];
    for (unsigned int item = blockIdx.x; item < cnt; item += gridDim.x) {
        const unsigned long long s = ranges[3 * (size_t) item], e = ranges[3 * (size_t) item + 1], o = ranges[3 * (size_t) item + 2];
        uint64_t lo = 0, hi = nRec;                              // first record with dst >= s (it starts at s)
        while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (dst[mid] < s) lo = mid + 1; else hi = mid; }
        for (uint64_t j0 = lo + 64ull * wave; j0 < nRec; j0 += 256) {
            const unsigned long long d0 = dst[j0];
            if (d0 >= e) break;
            const uint64_t j = j0 + lane;
            const bool valid = j < nRec && dst[j] < e;
            const uint64_t rv = valid ? recVal[j] : 0ull;
            const unsigned int len = (unsigned int) (rv & ((1ull << RUN_CNT_BITS) - 1ull));
            const unsigned int incl = cdm_wave_incl_sum<unsigned int>(len);
            sOff[lane] = incl - len; sStart[lane] = rv >> RUN_CNT_BITS;
            const unsigned int total = (unsigned int) __shfl((int) incl, 63, 64);
            bucket::waveLdsSync();
            for (unsigned int x = lane; x < total; x += 64) {
                int r = 0;
#pragma unroll
                for (int st = 32; st > 0; st >>= 1) if (sOff[r + st] <= x) r += st;
                const unsigned long long at = o + (d0 - s) + x;
                dense[at] = keys[sStart[r] + (x - sOff[r])] & keyMask;
                ord[at] = segOfRec[j0 + (uint64_t) r];
            }
            bucket::waveLdsSync();
        }
    }
}
typedef void (*UnitHook)(hipStream_t s, unsigned int grid, const unsigned long long *list, const unsigned int *count, bucket::BigList hard, void *user);
inline int segmentedSortKeys(hipStream_t s, int cuCount, uint64_t *in, uint64_t *out, uint64_t n, int shiftHi, int hiShift, int top,
                             const uint32_t *recRep, const unsigned long long *dst, uint64_t nRec, const uint64_t *srcKeys, const uint64_t *recVal,
                             UnitHook unitHook = nullptr, void *hookUser = nullptr, bool wide = false, const uint32_t *segOfRec = nullptr, int segBits = 0,
                             const unsigned long long *segFirstRec = nullptr, uint64_t nSeg = 0) {
    using namespace bucket;
    if (n == 0) return CDM_OK;
    if (wide && (!unitHook || !segOfRec)) { cdm_set_error("segmented sort: keys without the representative need the aggregating unit kernel and its segment table"); return CDM_ERR_UNSUPPORTED; }
    uint32_t maxSeg = U_MAXSEG, blockCap = wide ? 0 : 4096;
    if (const char *e = cdmGetenv("CDM_UNIT_CAP")) { const long m = atol(e); if (m >= 1 && m <= U_MAXSEG) maxSeg = (uint32_t) m; }
    if (const char *e = cdmGetenv("CDM_BLOCK_CAP")) { const long m = atol(e); if (m >= 0 && m <= 4096) blockCap = (uint32_t) m; }
    uint32_t maxSub = BK_MAXB;
    if (const char *e = cdmGetenv("CDM_UNIT_SUB")) { const long m = atol(e); if (m >= 1 && m <= BK_MAXB) maxSub = (uint32_t) m; }
    if (shiftHi - 1 + BLK_IDX > 64) blockCap = 0;       // the block sorter's word does not hold such keys: the global radix sort takes them
    if (!unitHook && U_ORDB + shiftHi - 1 + U_IDXB > 64) maxSeg = 0; // nor does the unit sorter's (ordinal, id, diagonal, index): everything goes to the radix sort
    const uint64_t units = (n + U_T - 1) / U_T;
    const size_t listCap = (size_t) (n / ((uint64_t) maxSeg + 1) + 2);
    DevBuf<unsigned long long> lists[SEG_CLASSES], hardList, uStart, uEnd, uRec, uList[U_CLASSES]; DevBuf<unsigned int> cnt;
    constexpr int NCNT = SEG_CLASSES + 1 + U_CLASSES;
    if (!cnt.alloc(NCNT) || !hardList.alloc(2 * (size_t) (units + 1)) || !uStart.alloc(units + 2) || !uEnd.alloc(units + 1) || !uRec.alloc(units + 2)) return CDM_ERR_HIP;
    for (int c = 0; c < U_CLASSES; c++) if (!uList[c].alloc(3 * (size_t) (units + 1))) return CDM_ERR_HIP;
    for (int c = 2; c < SEG_CLASSES; c++) if (!lists[c].alloc(2 * (listCap + (c == 3 ? (size_t) units + 1 : 0)))) return CDM_ERR_HIP;
    hipMemsetAsync(cnt.p, 0, NCNT * 4, s);
    if (maxSeg) {
        // a boundary no record reaches (there is none: the records tile [0, n)) would leave its unit empty
        hipMemsetAsync(uStart.p, 0, (units + 2) * 8, s); hipMemsetAsync(uEnd.p, 0, (units + 1) * 8, s); hipMemsetAsync(uRec.p, 0, (units + 2) * 8, s);
        UnitBoundArgs ub; ub.recRep = recRep; ub.dst = dst; ub.nRec = nRec; ub.maxSeg = maxSeg; ub.units = units; ub.uStart = uStart.p; ub.uEnd = uEnd.p; ub.uRec = uRec.p;
        if (segOfRec && segFirstRec) { ub.segOfRec = segOfRec; ub.segFirstRec = segFirstRec; }
        hipLaunchKernelGGL(k_unit_bounds, CDM_GRID((nRec + 1023) / 1024, 1024), dim3(1024), 0, s, ub);
        UnitClassArgs uc; uc.uStart = uStart.p; uc.uEnd = uEnd.p; uc.uRec = uRec.p; uc.units = units; uc.n = n;
        for (int c = 0; c < U_CLASSES; c++) { uc.cap[c] = (uint32_t) U_CLASS_CAP[c]; uc.list[c] = uList[c].p; }
        uc.cnt = cnt.p + SEG_CLASSES + 1;
        hipLaunchKernelGGL(k_unit_classes, dim3((unsigned) ((units + 1023) / 1024)), dim3(1024), 0, s, uc);
        UnitArgs ua; ua.keys = srcKeys; ua.recVal = recVal; ua.dst = dst; ua.nRec = nRec; ua.out = out; ua.n = n; ua.repShift = shiftHi; ua.hiShift = hiShift; ua.maxSub = maxSub; ua.hard.list = hardList.p; ua.hard.cnt = cnt.p + SEG_CLASSES;
        const unsigned int pad = cdm_lds_pad("CDM_LDS_PAD_UNIT");
        const unsigned int grid = (unsigned int) std::min<uint64_t>(units, (uint64_t) cuCount * 64);
        if (unitHook) {
            for (int c = 0; c < U_CLASSES; c++) unitHook(s, grid, uList[c].p, cnt.p + SEG_CLASSES + 1 + c, ua.hard, hookUser);
        } else {
        ua.list = uList[0].p; ua.count = cnt.p + SEG_CLASSES + 1; hipLaunchKernelGGL((k_unit_sort<U_CLASS_CAP[0], U_CLASS_NT[0]>), dim3(grid), dim3(U_CLASS_NT[0]), pad, s, ua);
        ua.list = uList[1].p; ua.count = cnt.p + SEG_CLASSES + 2; hipLaunchKernelGGL((k_unit_sort<U_CLASS_CAP[1], U_CLASS_NT[1]>), dim3(grid), dim3(U_CLASS_NT[1]), pad, s, ua);
        ua.list = uList[2].p; ua.count = cnt.p + SEG_CLASSES + 3; hipLaunchKernelGGL((k_unit_sort<U_CLASS_CAP[2], U_CLASS_NT[2]>), dim3(grid), dim3(U_CLASS_NT[2]), pad, s, ua);
        }
    }
    // the segments no unit can hold, listed by size class: one block of 8 waves (bitonic network, bucket.h) up to 4096, the global radix sort beyond
    SegListArgs la; la.recRep = recRep; la.dst = dst; la.nRec = nRec; la.maxWave = maxSeg; la.uEnd = maxSeg ? uEnd.p : nullptr; la.units = units;
    la.cap[0] = 0; la.cap[1] = 0; la.cap[2] = blockCap;
    for (int c = 0; c < SEG_CLASSES; c++) la.list[c] = lists[c].p;      // (classes 0 and 1 stay empty: their capacities are 0)
    la.cnt = cnt.p;
    if (segFirstRec && nSeg) { la.segFirstRec = segFirstRec; la.nSeg = nSeg; hipLaunchKernelGGL(k_seg_list_segments, CDM_GRID((nSeg + 1023) / 1024, 1024), dim3(1024), 0, s, la); }
    else hipLaunchKernelGGL(k_seg_list, CDM_GRID((nRec + 1023) / 1024, 1024), dim3(1024), 0, s, la);
    const unsigned int gatherGrid = (unsigned int) std::min<uint64_t>((uint64_t) cuCount * 8, listCap + units + 1);
    BlockSortArgs ba; ba.in = in; ba.out = out; ba.shiftHi = shiftHi; ba.ign = 1;
    ba.list = lists[2].p; ba.count = cnt.p + 2;
    if (blockCap) hipLaunchKernelGGL(k_gather_ranges, dim3(gatherGrid), dim3(256), 0, s, (const unsigned long long *) lists[2].p, (const unsigned int *) (cnt.p + 2), srcKeys, recVal, dst, nRec, in);
    if (blockCap) hipLaunchKernelGGL(k_block_sort<8>, dim3((unsigned int) std::min<uint64_t>((uint64_t) cuCount * 8, listCap)), dim3(512), 0, s, ba);
    unsigned int hc[NCNT] = {0};
    if (hipMemcpyAsync(hc, cnt.p, NCNT * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return CDM_ERR_HIP;
    if (cdmGetenv("CDM_BUCKET_STATS")) fprintf(stderr, "segmentedSortKeys%s: n %llu, %llu records, %llu units (%u / %u / %u by size class): segments > %u: %u (<= %u, block sort), %u longer (global radix sort); %u hard units (global radix sort)\n",
                                            unitHook ? " (units aggregated)" : "", (unsigned long long) n, (unsigned long long) nRec, (unsigned long long) units, hc[SEG_CLASSES + 1], hc[SEG_CLASSES + 2], hc[SEG_CLASSES + 3], maxSeg, hc[2], blockCap, hc[3], hc[SEG_CLASSES]);
    const unsigned int nBig = hc[3] + hc[SEG_CLASSES];
    if (nBig == 0) return CDM_OK;
    // deep pile-ups and hard units: gather, sort on the whole key with the radix sort of radix.h (stable), scatter.  (The ranges are disjoint and
    // the array is grouped by representative, so one sort of their concatenation on the whole key sorts each of them.)
    if (hc[SEG_CLASSES]) hipMemcpyAsync(lists[3].p + 2 * (size_t) hc[3], hardList.p, 2 * (size_t) hc[SEG_CLASSES] * 8, hipMemcpyDeviceToDevice, s);
    if (!wide) {   // their tuples, expanded into `in`
        DevBuf<unsigned int> nb;
        if (!nb.alloc(1)) return CDM_ERR_HIP;
        hipMemcpyAsync(nb.p, &nBig, 4, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(k_gather_ranges, dim3(gatherGrid), dim3(256), 0, s, (const unsigned long long *) lists[3].p, (const unsigned int *) nb.p, srcKeys, recVal, dst, nRec, in);
        if (hipStreamSynchronize(s) != hipSuccess) return CDM_ERR_HIP;
    }
    DevBuf<unsigned long long> ranges; uint64_t total = 0;
    if (int rc = loadBigList(s, lists[3].p, nBig, ranges, total)) return rc;
    DevBuf<uint64_t> d0, d1;
    if (!d0.alloc(total) || !d1.alloc(total)) return CDM_ERR_HIP;
    const unsigned int g = bigCopyGrid(nBig);
    bool inFirst = true;
    if (wide) {
        // no representative in the keys: (member bits, segment ordinal) pairs, gathered straight from the records - stable on the
        // member bits, then stable on the segment
        DevBuf<uint32_t> o0, o1;
        if (!o0.alloc(total) || !o1.alloc(total)) return CDM_ERR_HIP;
        hipLaunchKernelGGL(k_gather_ranges_dense, dim3(gatherGrid), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, srcKeys, recVal, dst, nRec, segOfRec, (1ull << shiftHi) - 1ull, d0.p, o0.p);
        if (int rc = rx::sortPairs<uint64_t, uint32_t>(s, cuCount, d0.p, d1.p, o0.p, o1.p, total, 1, shiftHi, inFirst)) return rc;
        uint64_t *kA = inFirst ? d0.p : d1.p, *kB = inFirst ? d1.p : d0.p; uint32_t *oA = inFirst ? o0.p : o1.p, *oB = inFirst ? o1.p : o0.p;
        bool second = true;
        if (int rc = rx::sortPairs<uint32_t, uint64_t>(s, cuCount, oA, oB, kA, kB, total, 0, std::max(1, segBits), second)) return rc;
        hipLaunchKernelGGL((k_big_copy<uint64_t, false>), dim3(g), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, out, second ? kA : kB);
    } else {
    hipLaunchKernelGGL((k_big_copy<uint64_t, true>), dim3(g), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, const_cast<uint64_t *>(in), d0.p);
    if (int rc = rx::sortKeys<uint64_t>(s, cuCount, d0.p, d1.p, total, 1, top, inFirst)) return rc;
    hipLaunchKernelGGL((k_big_copy<uint64_t, false>), dim3(g), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, out, inFirst ? d0.p : d1.p);
    }
    if (hipStreamSynchronize(s) != hipSuccess) return CDM_ERR_HIP;
    return CDM_OK;
}

// The run records of `ra.keys` (k_run_records; k_run_count + k_run_write if there are more than one per eight slots): recRep / recVal
// get room for two buffers each (the sort's double buffers).  Synchronises the stream.
inline int makeRunRecords(hipStream_t s, const RunArgs &ra, DevBuf<uint32_t> &rr0, DevBuf<uint32_t> &rr1, DevBuf<uint64_t> &rv0, DevBuf<uint64_t> &rv1, unsigned long long &nRec) {
    const uint64_t tiles = (ra.n + RUN_TILE - 1) / RUN_TILE;
    nRec = 0;
    if (tiles == 0) return CDM_OK;
    const char *mode = cdmGetenv("CDM_RUN_RECORDS");          // "twopass": the fallback only (tests)
    unsigned long long cap = ra.n / 8 + 4096;
    if (const char *e = cdmGetenv("CDM_RUN_CAP")) cap = strtoull(e, nullptr, 10);       // tests: force the overflow path
    if (!(mode && !strcmp(mode, "twopass"))) {
        DevBuf<unsigned long long> status, total; DevBuf<unsigned int> flags;
        if (!status.alloc(tiles) || !total.alloc(1) || !flags.alloc(2) || !rr0.alloc(cap) || !rr1.alloc(cap) || !rv0.alloc(cap + 1) || !rv1.alloc(cap + 1)) { cdm_set_error("run records: out of device memory"); return CDM_ERR_HIP; }
        hipMemsetAsync(status.p, 0, tiles * 8, s); hipMemsetAsync(total.p, 0, 8, s); hipMemsetAsync(flags.p, 0, 8, s);
        RunScan sc; sc.status = status.p; sc.ticket = flags.p; sc.overflow = flags.p + 1; sc.total = total.p; sc.cap = cap; sc.tiles = tiles;
        hipLaunchKernelGGL(k_run_records, dim3((unsigned) tiles), dim3(RUN_NT), 0, s, ra, sc, rr0.p, rv0.p);
        unsigned int fl[2] = {0, 0};
        hipMemcpyAsync(&nRec, total.p, 8, hipMemcpyDeviceToHost, s); hipMemcpyAsync(fl, flags.p, 8, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("run records failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
        if (!fl[1]) return CDM_OK;
        rr0.free(); rr1.free(); rv0.free(); rv1.free();
    }
    DevBuf<unsigned long long> tileCnt, tileOff; cdmscan::ScanTemp st;
    if (!tileCnt.alloc(tiles + 1) || !tileOff.alloc(tiles + 1)) { cdm_set_error("run records: out of device memory"); return CDM_ERR_HIP; }
    hipMemsetAsync(tileCnt.p + tiles, 0, 8, s);
    hipLaunchKernelGGL(k_run_count, dim3((unsigned) tiles), dim3(RUN_NT), 0, s, ra, tileCnt.p);
    if (int rc = cdmscan::exclusiveScan<unsigned long long>(s, st, tileCnt.p, tileOff.p, (size_t) tiles + 1)) return rc;
    hipMemcpyAsync(&nRec, tileOff.p + tiles, 8, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("run records failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    if (!rr0.alloc(nRec) || !rr1.alloc(nRec) || !rv0.alloc(nRec + 1) || !rv1.alloc(nRec + 1)) { cdm_set_error("run records: out of device memory (%llu records)", nRec); return CDM_ERR_HIP; }
    if (nRec) hipLaunchKernelGGL(k_run_write, dim3((unsigned) tiles), dim3(RUN_NT), 0, s, ra, (const unsigned long long *) tileOff.p, rr0.p, rv0.p);
    return CDM_OK;
}

}  // namespace runsort
