// kmermatcher's second sort (kmermatcher.cpp:431: the group tuples by (representative, member id, diagonal)) without global
// radix passes over the tuples.
//
// After assignGroup the tuples sit in k-mer order, and every k-mer run is a stretch of tuples with ONE representative: the
// array is run-length compressible on the very field the sort starts with.  So the runs are sorted, not the tuples:
//   1. k_run_count / k_run_write   one record (rep, start, length) per maximal stretch of kept tuples with the same
//                                  representative (a stretch never crosses a 4096-tuple tile, so a length fits 13 bits);
//                                  12 bytes per ~16 tuples
//   2. a stable radix sort of the records by rep (they are written in k-mer order): 1/12 of the tuples' bytes per pass
//   3. k_run_gather                expands the sorted records: the tuples of a representative become contiguous, still in
//                                  k-mer order (what the reference's stable sort leaves for equal (rep, id, diagonal));
//                                  dropped tuples (~0) vanish on the way, so this is also the compaction
//   4. segmentedSortKeys           each representative's segment is sorted on (id, diagonal): by a wavefront in registers
//                                  (bucket.h, segments up to 512 tuples), by a block of 2/4/8 wavefronts (k_block_sort, up to
//                                  4096), by rocPRIM beyond that (deep pile-ups).
// The tuple array is read three times and written twice here (count, write, gather; local sort), against 10 reads and
// 9 writes of the 4 radix passes + bucket finish this replaces.
#pragma once
#include "bucket.h"
#include "devutil.h"

namespace runsort {

constexpr int RUN_ITEMS = 16, RUN_NT = 256, RUN_TILE = RUN_NT * RUN_ITEMS;      // 4096 tuples per tile
constexpr int RUN_CNT_BITS = 13;                                                 // a record's length <= RUN_TILE

struct RunArgs {
    const uint64_t *keys;       // group keys in k-mer order, ~0 = dropped / unused slot
    uint64_t n;                 // slots
    uint64_t skipLo, skipHi;    // [skipLo, skipHi) holds only unused slots (the tail of region 1): not read
    int repShift;               // rep = key >> repShift
};
// LDS image of a tile, one pad per 16 items (thread t walks items 16 t .. 16 t + 15 without bank conflicts)
__device__ __forceinline__ int runPad(int i) { return i + (i >> 4); }
constexpr int RUN_LDS = RUN_TILE + RUN_TILE / 16 + 1;

// loads the tile into LDS (coalesced) and returns, for the 16 items of this thread, the bit masks "starts a record" and
// "ends the record in front of it" (= starts one, or is not kept)
__device__ __forceinline__ void runFlags(const RunArgs &a, uint64_t base, uint64_t *sKeys, unsigned int &startBits, unsigned int &boundBits) {
    const bool skip = base >= a.skipLo && base + RUN_TILE <= a.skipHi;      // block-uniform
#pragma unroll
    for (int j = 0; j < RUN_ITEMS; j++) {
        const int li = threadIdx.x + RUN_NT * j; const uint64_t i = base + (uint64_t) li;
        sKeys[runPad(li)] = (!skip && i < a.n && !(i >= a.skipLo && i < a.skipHi)) ? a.keys[i] : ~0ull;
    }
    __syncthreads();
    startBits = 0; boundBits = 0;
    uint64_t prev = (threadIdx.x == 0) ? ~0ull : sKeys[runPad(threadIdx.x * RUN_ITEMS - 1)];     // a record never crosses a tile
#pragma unroll
    for (int j = 0; j < RUN_ITEMS; j++) {
        const uint64_t k = sKeys[runPad(threadIdx.x * RUN_ITEMS + j)];
        const bool kept = k != ~0ull;
        const bool start = kept && (prev == ~0ull || (prev >> a.repShift) != (k >> a.repShift));
        if (start) startBits |= 1u << j;
        if (start || !kept) boundBits |= 1u << j;
        prev = k;
    }
}
__global__ __launch_bounds__(RUN_NT) void k_run_count(RunArgs a, unsigned long long *__restrict__ tileCnt) {
    __shared__ uint64_t sKeys[RUN_LDS];
    unsigned int sb, bb;
    runFlags(a, (uint64_t) blockIdx.x * RUN_TILE, sKeys, sb, bb);
    const unsigned int tot = cdm_block_sum<unsigned int>((unsigned int) __popc(sb));
    if (threadIdx.x == 0) tileCnt[blockIdx.x] = tot;
}
// record j: recRep[j] = representative, recVal[j] = start << 13 | length
__global__ __launch_bounds__(RUN_NT) void k_run_write(RunArgs a, const unsigned long long *__restrict__ tileOff, uint32_t *__restrict__ recRep, uint64_t *__restrict__ recVal) {
    __shared__ uint64_t sKeys[RUN_LDS];
    __shared__ __align__(8) uint16_t sBound[RUN_TILE / 16 + 4];
    const uint64_t base = (uint64_t) blockIdx.x * RUN_TILE;
    unsigned int sb, bb;
    runFlags(a, base, sKeys, sb, bb);
    sBound[threadIdx.x] = (uint16_t) bb;
    if (threadIdx.x < 4) sBound[RUN_TILE / 16 + threadIdx.x] = 0;
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
        recRep[rank] = (uint32_t) (sKeys[runPad(li)] >> a.repShift);
        recVal[rank] = ((base + (uint64_t) li) << RUN_CNT_BITS) | (uint64_t) (e - li);
        rank++;
    }
}
struct RunLen { const uint64_t *recVal; __device__ __forceinline__ unsigned long long operator()(size_t j) const { return recVal[j] & ((1ull << RUN_CNT_BITS) - 1ull); } };

// expands the sorted records: out[dst[j] ..] = the tuples of record j.  A wave takes 64 consecutive records; their tuples
// are consecutive in `out`, so the writes are whole lines and the reads are the records' 8-byte x length stretches.
__global__ __launch_bounds__(256) void k_run_gather(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ recVal, const unsigned long long *__restrict__ dst,
                                                    uint64_t nRec, uint64_t *__restrict__ out) {
    __shared__ uint32_t sOffAll[4][64];
    __shared__ uint64_t sStartAll[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *sOff = sOffAll[wave]; uint64_t *sStart = sStartAll[wave];
    const uint64_t j0 = ((uint64_t) blockIdx.x * 4 + wave) * 64;
    if (j0 >= nRec) return;      // (whole wave)
    const uint64_t j = j0 + lane;
    const uint64_t rv = (j < nRec) ? recVal[j] : 0ull;
    const unsigned int cnt = (unsigned int) (rv & ((1ull << RUN_CNT_BITS) - 1ull));
    const unsigned int incl = cdm_wave_incl_sum<unsigned int>(cnt);
    sOff[lane] = incl - cnt; sStart[lane] = rv >> RUN_CNT_BITS;
    const unsigned int total = (unsigned int) __shfl((int) incl, 63, 64);
    const unsigned long long d0 = dst[j0];
    bucket::waveLdsSync();
    for (unsigned int e = lane; e < total; e += 64) {
        int r = 0;                                               // last record with sOff[r] <= e (empty padding records have cnt 0)
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) if (sOff[r + s] <= e) r += s;
        out[d0 + e] = keys[sStart[r] + (e - sOff[r])];
    }
}

// ---------------------------------------------------------------------------------------------- segments by size class
constexpr int SEG_CLASSES = 4;          // 0..2: k_block_sort with 2, 4, 8 waves; 3: longer (rocPRIM)
struct SegListArgs {
    const uint32_t *recRep; const unsigned long long *dst; uint64_t nRec;
    uint32_t maxWave;                   // segments up to this length are finished by k_bucket_sort
    uint32_t cap[SEG_CLASSES - 1];      // capacities of the block classes
    unsigned long long *list[SEG_CLASSES]; unsigned int *cnt;      // cnt[c]
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
            if (n > a.maxWave) cls = n <= a.cap[0] ? 0 : n <= a.cap[1] ? 1 : n <= a.cap[2] ? 2 : 3;
        }
    }
#pragma unroll
    for (int c = 0; c < SEG_CLASSES; c++) {
        const uint32_t q = cdm_block_append(a.cnt + c, cls == c);
        if (cls == c) { a.list[c][2 * (size_t) q] = s; a.list[c][2 * (size_t) q + 1] = e; }
    }
}

// `in` holds the tuples grouped by representative (bits >= shiftHi), each group in k-mer order; `out` = every group stably
// sorted on bits [ign, shiftHi).  recRep / dst / nRec describe the groups (sorted records and their output offsets).
inline int segmentedSortKeys(hipStream_t s, int cuCount, const uint64_t *in, uint64_t *out, uint64_t n, int shiftHi, int ign, int top,
                             const uint32_t *recRep, const unsigned long long *dst, uint64_t nRec) {
    using namespace bucket;
    if (n == 0) return CDM_OK;
    int own; uint32_t maxBucket; capacities(own, maxBucket);
    uint32_t cap[3] = {1024, 2048, 4096};
    if (const char *e = getenv("CDM_BLOCK_CAP")) { const long m = atol(e); if (m >= 1 && m <= 4096) { cap[2] = (uint32_t) m; cap[1] = std::min(cap[1], cap[2]); cap[0] = std::min(cap[0], cap[1]); } }
    if (shiftHi - ign + BLK_IDX > 64) cap[0] = cap[1] = cap[2] = 0;       // the block sorter's word does not hold such keys: everything long goes to rocPRIM
    // wave-sized segments
    {
        SortArgs a; a.in = in; a.out = out; a.n = n; a.shiftHi = shiftHi; a.ign = ign; a.own = own; a.maxBucket = maxBucket; a.big.list = nullptr; a.big.cnt = nullptr;
        const uint64_t perBlock = (uint64_t) own * BK_WAVES;
        hipLaunchKernelGGL(k_bucket_sort, dim3((unsigned) ((n + perBlock - 1) / perBlock)), dim3(BK_NT), 0, s, a);
    }
    // the longer ones, listed by size class
    const size_t listCap = (size_t) (n / ((uint64_t) maxBucket + 1) + 2);
    DevBuf<unsigned long long> lists[SEG_CLASSES]; DevBuf<unsigned int> cnt;
    if (!cnt.alloc(SEG_CLASSES)) return CDM_ERR_HIP;
    for (int c = 0; c < SEG_CLASSES; c++) if (!lists[c].alloc(2 * listCap)) return CDM_ERR_HIP;
    hipMemsetAsync(cnt.p, 0, SEG_CLASSES * 4, s);
    SegListArgs la; la.recRep = recRep; la.dst = dst; la.nRec = nRec; la.maxWave = maxBucket;
    for (int c = 0; c < 3; c++) la.cap[c] = cap[c];
    for (int c = 0; c < SEG_CLASSES; c++) la.list[c] = lists[c].p;
    la.cnt = cnt.p;
    hipLaunchKernelGGL(k_seg_list, dim3((unsigned) ((nRec + 1023) / 1024)), dim3(1024), 0, s, la);
    BlockSortArgs ba; ba.in = in; ba.out = out; ba.shiftHi = shiftHi; ba.ign = ign;
    const unsigned int grid = (unsigned int) std::min<uint64_t>((uint64_t) cuCount * 8, listCap);
    ba.list = lists[0].p; ba.count = cnt.p + 0; hipLaunchKernelGGL(k_block_sort<2>, dim3(grid), dim3(128), 0, s, ba);
    ba.list = lists[1].p; ba.count = cnt.p + 1; hipLaunchKernelGGL(k_block_sort<4>, dim3(grid), dim3(256), 0, s, ba);
    ba.list = lists[2].p; ba.count = cnt.p + 2; hipLaunchKernelGGL(k_block_sort<8>, dim3(grid), dim3(512), 0, s, ba);
    unsigned int hc[SEG_CLASSES] = {0, 0, 0, 0};
    if (hipMemcpyAsync(hc, cnt.p, SEG_CLASSES * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return CDM_ERR_HIP;
    if (getenv("CDM_BUCKET_STATS")) fprintf(stderr, "segmentedSortKeys: n %llu, %llu records: segments > %u: %u (<= %u), %u (<= %u), %u (<= %u), %u longer\n",
                                            (unsigned long long) n, (unsigned long long) nRec, maxBucket, hc[0], cap[0], hc[1], cap[1], hc[2], cap[2], hc[3]);
    if (hc[3] == 0) return CDM_OK;
    // deep pile-ups: gather, sort on the whole key with rocPRIM (stable), scatter
    DevBuf<unsigned long long> ranges; uint64_t total = 0;
    if (int rc = loadBigList(s, lists[3].p, hc[3], ranges, total)) return rc;
    DevBuf<uint64_t> d0, d1; DevBuf<char> tmp; size_t tb = 0;
    if (!d0.alloc(total) || !d1.alloc(total)) return CDM_ERR_HIP;
    const unsigned int g = bigCopyGrid(hc[3]);
    hipLaunchKernelGGL((k_big_copy<uint64_t, true>), dim3(g), dim3(256), 0, s, (const unsigned long long *) ranges.p, hc[3], const_cast<uint64_t *>(in), d0.p);
    rocprim::double_buffer<uint64_t> db(d0.p, d1.p);
    if (rocprim::radix_sort_keys(nullptr, tb, db, (size_t) total, ign, top, s) != hipSuccess || !tmp.alloc(tb + 256)) return CDM_ERR_HIP;
    if (rocprim::radix_sort_keys(tmp.p, tb, db, (size_t) total, ign, top, s) != hipSuccess) return CDM_ERR_HIP;
    hipLaunchKernelGGL((k_big_copy<uint64_t, false>), dim3(g), dim3(256), 0, s, (const unsigned long long *) ranges.p, hc[3], out, db.current());
    if (hipStreamSynchronize(s) != hipSuccess) return CDM_ERR_HIP;
    return CDM_OK;
}

}  // namespace runsort
