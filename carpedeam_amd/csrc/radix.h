// Stable LSD radix sort of (key, value) pairs (32- or 64-bit each) for kmermatcher's sort 1 (the reference: ips4o on the k-mer tuples,
// lib/mmseqs/src/linclust/kmermatcher.cpp:412; any stable order-by-key is the same array).  Hand-written "onesweep":
//
//   k_rx_hist     one read of the keys: digit counts of ALL passes (LDS histograms, one flush per block)
//   k_rx_offsets  exclusive scan of every pass's 512 counts -> where each digit's items start in the output
//   k_rx_pass     per pass ONE read and ONE write of the pairs.  A block takes a tile of 8192 consecutive pairs (ticket-numbered:
//                 a tile only waits for tiles that run already); wave w owns the 1024 consecutive pairs [1024 w, 1024 (w + 1)).
//                 Rank of a pair among the tile's pairs with its digit, in tile order (stability): per wave and round of 64 pairs
//                 the lanes with the same digit find each other with one ballot per digit bit (9), the first of them advances the
//                 wave's digit counter; counters of the waves in front are added afterwards.  Where the tile's digit-d pairs go
//                 in the output: a chained scan over the tiles, one thread per digit, with decoupled look-back (status word =
//                 2 flag bits | count, relaxed atomics: the word is the whole message).  The pairs are reordered in LDS first, so
//                 that the stores to HBM are runs of consecutive addresses per digit.
//
// 9 bits per pass: 512 digits x 8192-pair tiles = 16 pairs per run on average (128 B of keys).  HBM bound by design:
// 24.6 GB per 2^30 12-byte pairs and pass.
#pragma once
#include <cstring>
#include <algorithm>
#include "common.h"
#include "devutil.h"

namespace rx {

#ifndef CDM_RX_NT
#define CDM_RX_NT 512
#define CDM_RX_IPT 16
#endif
constexpr int NT = CDM_RX_NT, WAVES = NT / 64, IPT = CDM_RX_IPT, TILE = NT * IPT, BITS = 9, BINS = 1 << BITS, MAXPASS = 8;
static_assert(NT >= BINS && TILE <= 65536, "radix pass geometry");
#ifndef CDM_RX_LB
#define CDM_RX_LB 2
#endif
constexpr int LB = CDM_RX_LB;
constexpr unsigned long long ST_AGG = 1ull << 62, ST_PREFIX = 2ull << 62, ST_MASK = (1ull << 62) - 1ull;

struct NoValue {};      // V = NoValue: keys only
template <typename V> struct HasValue { static constexpr bool value = true; };
template <> struct HasValue<NoValue> { static constexpr bool value = false; };
template <typename K> struct HistArgs { const K *keys; uint64_t n; int beginBit, endBit, passes; unsigned long long *hist; };      // hist[pass][digit]
template <typename K> __device__ __forceinline__ uint32_t digitOf(K key, int shift, uint32_t mask) { return (uint32_t) (key >> shift) & mask; }

template <typename K>
__global__ __launch_bounds__(NT) void k_rx_hist(HistArgs<K> a) {
    __shared__ unsigned int sHist[MAXPASS][BINS];
    for (int i = threadIdx.x; i < a.passes * BINS; i += NT) (&sHist[0][0])[i] = 0u;
    __syncthreads();
    // a block's share of the array is at most 2^32 - 1 keys (the LDS counters are 32 bit): the host sizes the grid for that
    for (uint64_t t = blockIdx.x; t * TILE < a.n; t += gridDim.x) {
        const uint64_t base = t * TILE;
#pragma unroll 4
        for (int j = 0; j < IPT; j++) {
            const uint64_t i = base + (uint64_t) j * NT + threadIdx.x;
            if (i >= a.n) break;
            const K k = a.keys[i];
            for (int p = 0; p < a.passes; p++) {
                const int shift = a.beginBit + p * BITS, bits = min(BITS, a.endBit - shift);
                atomicAdd(&sHist[p][digitOf(k, shift, (1u << bits) - 1u)], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.passes * BINS; i += NT) { const unsigned int c = (&sHist[0][0])[i]; if (c) atomicAdd(&a.hist[i], (unsigned long long) c); }
}
// in place: hist[pass][digit] -> first output position of the digit
template <int UNUSED = 0>        // (a template: the header is included by several translation units)
__global__ __launch_bounds__(BINS) void k_rx_offsets(unsigned long long *hist, int passes) {
    for (int p = 0; p < passes; p++) {
        const unsigned long long c = hist[p * BINS + threadIdx.x];
        unsigned long long tot;
        const unsigned long long ex = cdm_block_excl_sum<unsigned long long>(c, tot);
        hist[p * BINS + threadIdx.x] = ex;
    }
}

// Three forms of the pass (MODE):
//   PASS_PLAIN  the n pairs, stable on the digit
//   PASS_HEAD   sort 1 of kmermatcher on SLOT KEYS (sortSlotKeys below): kin holds one 64-bit key per k-mer slot, k-mer | strand << 63, ~0 =
//               empty slot.  The pass is the most significant one: it drops the empty slots, partitions the others by the head digit (the
//               `bits` key bits from `shift` on) and writes 8-byte SLOT TUPLES [ key bits below shift : 31 | strand : 1 | slot index : 32 ] -
//               the head digit is implied by where a tuple lands (segment = digit), the slot index says which sequence and position the
//               k-mer came from (the slots are laid out by sequence: kmermatch.hip).  Keys only.
//   PASS_SEG    the array is cut into BINS segments (seg[0 .. BINS]: the head pass's digits); every segment is sorted on its own, stable:
//               tiles do not cross a segment (segTile[d] = first tile of segment d), the chained scan starts afresh at a segment's first
//               tile, digitBase is [segment][digit]
enum { PASS_PLAIN = 0, PASS_HEAD = 1, PASS_SEG = 2 };
constexpr int SLOT_REM = 31, SLOT_IDX_SHIFT = 0, SLOT_STRAND_SHIFT = 32, SLOT_KEY_SHIFT = 33;       // the slot tuple's fields
template <typename K, typename V>
struct PassArgs {
    const K *kin; K *kout; const V *vin; V *vout; uint64_t n;
    int shift, bits;
    const unsigned long long *digitBase;        // [BINS] ([BINS][BINS] in PASS_SEG)
    unsigned long long *status;                 // [tiles][BINS], zeroed
    unsigned int *ticket;                       // zeroed
    const unsigned long long *seg = nullptr; const unsigned int *segTile = nullptr;     // PASS_SEG: [BINS + 1] each
    uint32_t keepLo = 0, keepHi = 0xFFFFFFFFu;  // PASS_HEAD: keys whose digit lies outside [keepLo, keepHi) are dropped like empty slots (a rank's range of the k-mer space)
};
template <typename K, typename V, int MODE = PASS_PLAIN>
__global__ __launch_bounds__(NT) void k_rx_pass(PassArgs<K, V> a) {
    static_assert(MODE == PASS_PLAIN || (sizeof(K) == 8 && !HasValue<V>::value), "the head and segment passes sort 64-bit keys only");
    static_assert(MODE != PASS_HEAD || TILE <= 8192, "the head pass keeps a key's index in its tile in 13 bits of the exchange word");
    // one 64 KB exchange buffer, used for the keys and then for the values: two blocks per CU
    __shared__ uint64_t sBuf[TILE];
    __shared__ uint16_t sCnt[WAVES][BINS];
    __shared__ uint16_t sTileOff[BINS];
    __shared__ unsigned long long sGlobal[BINS];
    __shared__ unsigned int sTile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) sTile = atomicAdd(a.ticket, 1u);
    for (int i = tid; i < WAVES * BINS / 2; i += NT) reinterpret_cast<uint32_t *>(&sCnt[0][0])[i] = 0u;
    __syncthreads();
    const uint64_t tile = sTile;
    uint64_t base = tile * TILE, chain = tile;          // chain: tiles in front of this one in its chained scan
    uint64_t segEnd = a.n;
    const unsigned long long *digitBase = a.digitBase;
    if constexpr (MODE == PASS_SEG) {
        const int sg = __syncthreads_count(tid < BINS && (uint64_t) a.segTile[tid < BINS ? tid : 0] <= tile) - 1;      // the last segment whose first tile is not behind this one
        chain = tile - a.segTile[sg];
        base = a.seg[sg] + chain * TILE; segEnd = a.seg[sg + 1];
        digitBase += (size_t) sg * BINS;
    }
    const int items = (int) min((uint64_t) TILE, segEnd - base);
    const uint32_t mask = (1u << a.bits) - 1u;
    // ---- load: wave w owns [w * 64 IPT, (w + 1) * 64 IPT), round j its j-th 64 pairs
    K key[IPT]; V val[IPT]; uint16_t pos[IPT];
    const int w0 = wave * 64 * IPT + lane;
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const int li = w0 + 64 * j;
        if (li < items) { key[j] = a.kin[base + li]; if constexpr (HasValue<V>::value) val[j] = a.vin[base + li]; } else key[j] = (MODE == PASS_HEAD) ? (K) ~(K) 0 : (K) 0;
        if constexpr (MODE == PASS_HEAD) { const uint32_t d = digitOf(key[j], a.shift, mask); if (d < a.keepLo || d >= a.keepHi) key[j] = (K) ~(K) 0; }
    }
    // ---- rank inside the wave's stream, round by round
    uint16_t *cntW = sCnt[wave];
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const bool valid = (MODE == PASS_HEAD) ? key[j] != (K) ~(K) 0 : w0 + 64 * j < items;
        const uint32_t d = digitOf(key[j], a.shift, mask);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) peers, 0u));
        uint32_t before = 0;
        if (valid) before = cntW[d];
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) cntW[d] = (uint16_t) (before + (uint32_t) __popcll(peers));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        pos[j] = (uint16_t) (before + rank);
    }
    __syncthreads();
    // ---- per digit (thread = digit): the waves in front, the digits in front (tile); the tile's count is published at once
    const bool isDigit = tid < BINS;
    uint32_t run = 0;
    if (isDigit) {
#pragma unroll
        for (int w = 0; w < WAVES; w++) { const uint32_t c = sCnt[w][tid]; sCnt[w][tid] = (uint16_t) run; run += c; }
    }
    uint32_t tot;
    const uint32_t ex = cdm_block_excl_sum<uint32_t>(run, tot);
    unsigned long long *st = a.status + tile * BINS + tid;
    if (isDigit) {
        sTileOff[tid] = (uint16_t) ex;
        __hip_atomic_store(st, (chain == 0 ? ST_PREFIX : ST_AGG) | (unsigned long long) run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    K *sK = reinterpret_cast<K *>(sBuf);
    const int nOut = (MODE == PASS_HEAD) ? (int) tot : items;       // (the head pass drops the empty slots)
    // ---- the keys into the exchange buffer: slot = digits in front + waves in front + rank (needs nothing of the other tiles)
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const bool valid = (MODE == PASS_HEAD) ? key[j] != (K) ~(K) 0 : w0 + 64 * j < items;
        if (valid) {
            const uint32_t d = digitOf(key[j], a.shift, mask);
            pos[j] = (uint16_t) ((uint32_t) sTileOff[d] + (uint32_t) sCnt[wave][d] + (uint32_t) pos[j]);
            if constexpr (MODE == PASS_HEAD)     // exchange word: [ key bits below shift | strand | digit : 9 | index in the tile : 13 ]
                sK[pos[j]] = (K) ((((uint64_t) key[j] & ((1ull << a.shift) - 1ull)) << 23) | (((uint64_t) key[j] >> 63) << 22) | ((uint64_t) d << 13) | (uint64_t) (w0 + 64 * j));
            else sK[pos[j]] = key[j];
        }
    }
    // ---- the tiles in front: chained scan with decoupled look-back, after the exchange so that the tiles in front had that time to
    // publish their prefix; LB status words are fetched per round trip
    if (isDigit) {
        unsigned long long exclG = 0;
        if (chain != 0) {
            const unsigned long long *q = st - BINS;
            uint64_t left = chain;
            bool done = false;
            while (!done) {
                unsigned long long v[LB];
#pragma unroll
                for (int u = 0; u < LB; u++) v[u] = (uint64_t) u < left ? __hip_atomic_load(q - (size_t) u * BINS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ST_PREFIX;
#pragma unroll
                for (int u = 0; u < LB; u++) {
                    if (done) break;
                    unsigned long long x = v[u];
                    while ((x >> 62) == 0ull) x = __hip_atomic_load(q - (size_t) u * BINS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    exclG += x & ST_MASK;
                    if ((x >> 62) == 2ull) done = true;
                }
                q -= LB * (size_t) BINS; left = left > LB ? left - LB : 0;
            }
            __hip_atomic_store(st, ST_PREFIX | (exclG + run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sGlobal[tid] = digitBase[tid] + exclG - ex;         // output position of the pair at exchange slot p with this digit: sGlobal + p
    }
    __syncthreads();
    uint16_t dig[IPT];
#pragma unroll
    for (int r = 0; r < IPT; r++) {
        const int p = tid + NT * r;
        if (p < nOut) {
            const K k = sK[p];
            if constexpr (MODE == PASS_HEAD) {
                const uint64_t w = (uint64_t) k;
                dig[r] = (uint16_t) ((w >> 13) & (uint64_t) (BINS - 1));
                a.kout[sGlobal[dig[r]] + (unsigned long long) p] = (K) (((w >> 23) << SLOT_KEY_SHIFT) | (((w >> 22) & 1ull) << SLOT_STRAND_SHIFT) | (uint64_t) (uint32_t) (base + (w & 8191ull)));
            } else {
                dig[r] = (uint16_t) digitOf(k, a.shift, mask);
                a.kout[sGlobal[dig[r]] + (unsigned long long) p] = k;
            }
        }
    }
    if constexpr (HasValue<V>::value) {
        __syncthreads();
        // ---- the values the same way
        V *sV = reinterpret_cast<V *>(sBuf);
#pragma unroll
        for (int j = 0; j < IPT; j++) if (w0 + 64 * j < items) sV[pos[j]] = val[j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const int p = tid + NT * r;
            if (p < items) a.vout[sGlobal[dig[r]] + (unsigned long long) p] = sV[p];
        }
    }
}

// ---- stable compaction: the pairs whose key is not ~0 (empty slot), in order, to the front of (kout, vout); *total = how many.
// One pass: per wave and round of 64 pairs a ballot ranks the kept ones, the tiles chain their counts with the same look-back.
// (kmermatcher with a k-mer RANGE per rank leaves most slots empty: compacting first makes sort 1 as short as the range.)
constexpr int CP_NT = 256, CP_WAVES = CP_NT / 64, CP_TILE = CP_NT * IPT;
template <typename K, typename V>
struct CompactArgs { const K *kin; const V *vin; uint64_t n; K *kout; V *vout; unsigned long long *status; unsigned int *ticket; unsigned long long *total; };
template <typename K, typename V>
__global__ __launch_bounds__(CP_NT) void k_rx_compact(CompactArgs<K, V> a) {
    __shared__ unsigned int sTile, sWave[CP_WAVES];
    __shared__ unsigned long long sPrefix;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) sTile = atomicAdd(a.ticket, 1u);
    __syncthreads();
    const uint64_t tile = sTile, base = tile * CP_TILE;
    const int items = (int) min((uint64_t) CP_TILE, a.n - base);
    K key[IPT]; V val[IPT]; uint16_t pos[IPT];
    const int w0 = wave * 64 * IPT + lane;
    uint32_t kept = 0;          // bit j: pair j of this lane is kept
    uint32_t run = 0;           // wave-uniform: kept pairs of the wave so far
#pragma unroll
    for (int j = 0; j < IPT; j++) { const int li = w0 + 64 * j; key[j] = li < items ? a.kin[base + li] : (K) ~(K) 0; }
#pragma unroll
    for (int j = 0; j < IPT; j++) if (key[j] != (K) ~(K) 0) val[j] = a.vin[base + w0 + 64 * j];       // (the values of the kept pairs only)
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const bool k = key[j] != (K) ~(K) 0;
        const unsigned long long m = __ballot(k);
        pos[j] = (uint16_t) (run + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u)));
        if (k) kept |= 1u << j;
        run += (uint32_t) __popcll(m);
    }
    if (lane == 0) sWave[wave] = run;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < CP_WAVES; w++) { const uint32_t c = sWave[w]; if (w < wave) before += c; tot += c; }
    if (wave == 0) {
        unsigned long long excl = 0;
        if (tile == 0) { if (lane == 0) __hip_atomic_store(&a.status[0], ST_PREFIX | (unsigned long long) tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else {
            if (lane == 0) __hip_atomic_store(&a.status[tile], ST_AGG | (unsigned long long) tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long t0 = (long long) tile - 1;
            while (true) {
                const long long t = t0 - lane;
                unsigned long long v = ST_PREFIX;
                if (t >= 0) do { v = __hip_atomic_load(&a.status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((v >> 62) == 0ull);
                const unsigned long long pm = __ballot((v >> 62) == 2ull);
                const int first = pm ? __ffsll(pm) - 1 : 63;
                const unsigned long long part = cdm_wave_incl_sum<unsigned long long>(lane <= first ? (v & ST_MASK) : 0ull);
                excl += (unsigned long long) __shfl((long long) part, 63, 64);
                if (pm) break;
                t0 -= 64;
            }
            if (lane == 0) __hip_atomic_store(&a.status[tile], ST_PREFIX | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) { sPrefix = excl; if ((tile + 1) * CP_TILE >= a.n) *a.total = excl + tot; }
    }
    __syncthreads();
    const unsigned long long o = sPrefix + before;
#pragma unroll
    for (int j = 0; j < IPT; j++) if (kept & (1u << j)) { a.kout[o + pos[j]] = key[j]; a.vout[o + pos[j]] = val[j]; }
}
// asynchronous on s; *totalDev (device) receives the count
template <typename K, typename V>
inline int compactPairs(hipStream_t s, const K *kin, const V *vin, uint64_t n, K *kout, V *vout, unsigned long long *totalDev) {
    const uint64_t tiles = (n + CP_TILE - 1) / CP_TILE;
    hipMemsetAsync(totalDev, 0, 8, s);
    if (tiles == 0) return CDM_OK;
    unsigned long long *status = nullptr; unsigned int *ticket = nullptr;
    if (cdmMalloc(&status, tiles * 8 + 8) != hipSuccess || cdmMalloc(&ticket, 8) != hipSuccess) { if (status) cdmFree(status); cdm_set_error("compaction: out of device memory"); return CDM_ERR_HIP; }
    hipMemsetAsync(status, 0, tiles * 8, s); hipMemsetAsync(ticket, 0, 4, s);
    CompactArgs<K, V> a; a.kin = kin; a.vin = vin; a.n = n; a.kout = kout; a.vout = vout; a.status = status; a.ticket = ticket; a.total = totalDev;
    hipLaunchKernelGGL((k_rx_compact<K, V>), dim3((unsigned) tiles), dim3(CP_NT), 0, s, a);
    const hipError_t e = hipStreamSynchronize(s);       // (the scratch goes back to the pool)
    cdmFree(status); cdmFree(ticket);
    if (e != hipSuccess) { cdm_set_error("compaction failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; }
    return CDM_OK;
}

// Sorts the n pairs on key bits [beginBit, endBit), stable.  (k0, v0) hold the input; the passes alternate between the two buffer
// pairs; inFirst tells where the result is.  Asynchronous on s except for the allocations.
// passMs (may be NULL): HIP-event time of the pass launches alone, summed.
template <typename K, typename V>
inline int sortPairs(hipStream_t s, int cuCount, K *k0, K *k1, V *v0, V *v1, uint64_t n, int beginBit, int endBit, bool &inFirst, float *passMs = nullptr) {
    inFirst = true;
    if (n == 0 || endBit <= beginBit) return CDM_OK;
    const int passes = (endBit - beginBit + BITS - 1) / BITS;
    if (passes > MAXPASS) { cdm_set_error("radix sort: %d passes", passes); return CDM_ERR_INVALID; }
    const uint64_t tiles = (n + TILE - 1) / TILE;
    DevBuf<unsigned long long> hist, status; DevBuf<unsigned int> ticket;
    if (!hist.alloc((size_t) passes * BINS) || !status.alloc(tiles * BINS) || !ticket.alloc(passes)) { cdm_set_error("radix sort: out of device memory"); return CDM_ERR_HIP; }
    hipMemsetAsync(hist.p, 0, (size_t) passes * BINS * 8, s);
    hipMemsetAsync(ticket.p, 0, (size_t) passes * 4, s);
    HistArgs<K> ha; ha.keys = k0; ha.n = n; ha.beginBit = beginBit; ha.endBit = endBit; ha.passes = passes; ha.hist = hist.p;
    hipLaunchKernelGGL(k_rx_hist<K>, dim3((unsigned) std::min<uint64_t>(tiles, (uint64_t) cuCount * 8)), dim3(NT), 0, s, ha);
    hipLaunchKernelGGL(k_rx_offsets<0>, dim3(1), dim3(BINS), 0, s, hist.p, passes);
    hipEvent_t ev[2 * MAXPASS];
    if (passMs) for (int i = 0; i < 2 * passes; i++) hipEventCreate(&ev[i]);
    for (int p = 0; p < passes; p++) {
        hipMemsetAsync(status.p, 0, tiles * BINS * 8, s);
        PassArgs<K, V> pa;
        pa.kin = inFirst ? k0 : k1; pa.kout = inFirst ? k1 : k0; pa.vin = inFirst ? v0 : v1; pa.vout = inFirst ? v1 : v0; pa.n = n;
        pa.shift = beginBit + p * BITS; pa.bits = std::min(BITS, endBit - pa.shift);
        pa.digitBase = hist.p + (size_t) p * BINS; pa.status = status.p; pa.ticket = ticket.p + p;
        if (passMs) hipEventRecord(ev[2 * p], s);
        hipLaunchKernelGGL((k_rx_pass<K, V>), dim3((unsigned) tiles), dim3(NT), 0, s, pa);
        if (passMs) hipEventRecord(ev[2 * p + 1], s);
        inFirst = !inFirst;
    }
    const hipError_t e = hipStreamSynchronize(s);
    if (passMs) {
        *passMs = 0.f;
        for (int p = 0; p < passes; p++) { float ms = 0.f; if (e == hipSuccess) hipEventElapsedTime(&ms, ev[2 * p], ev[2 * p + 1]); *passMs += ms; }
        for (int i = 0; i < 2 * passes; i++) hipEventDestroy(ev[i]);
    }
    if (e != hipSuccess) { cdm_set_error("radix sort failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    return CDM_OK;
}

// keys only
template <typename K>
inline int sortKeys(hipStream_t s, int cuCount, K *k0, K *k1, uint64_t n, int beginBit, int endBit, bool &inFirst) {
    return sortPairs<K, NoValue>(s, cuCount, k0, k1, (NoValue *) nullptr, (NoValue *) nullptr, n, beginBit, endBit, inFirst);
}


// ---------------------------------------------------------------------------------------------- sort 1 on slot keys
// kmermatcher's sort 1 (kmermatcher.cpp:412) when every sequence has the same length: the extractor writes ONE 64-bit key per k-mer
// slot (k-mer | strand << 63, ~0 = empty), and which sequence and position a key belongs to is its slot index - so a tuple is 8 bytes
// instead of 12, in every pass:
//   k_rx_hist_head    digit counts of the head pass (skipped when the extractor counted while it wrote: headHist)
//   k_rx_seg_layout   the segments (= head digits) and their tiles
//   k_rx_pass<HEAD>   most significant digit first: empty slots dropped, slot tuples out (see PASS_HEAD)
//   k_rx_hist_seg     digit counts of the remaining global passes per segment (one read of the tuples)
//   k_rx_pass<SEG>    those passes, least significant digit first, inside every segment
// The result is ordered on the key bits [lowBits, topBit) - what the plain passes leave - and stable: equal keys stay in slot order.
template <int UNUSED = 0>
__global__ __launch_bounds__(NT) void k_rx_hist_head(const uint64_t *__restrict__ keys, uint64_t n, int shift, int bits, unsigned long long *__restrict__ hist) {
    __shared__ unsigned int sHist[BINS];
    for (int i = threadIdx.x; i < BINS; i += NT) sHist[i] = 0u;
    __syncthreads();
    const uint32_t mask = (1u << bits) - 1u;
    for (uint64_t t = blockIdx.x; t * TILE < n; t += gridDim.x) {
        const uint64_t base = t * TILE;
#pragma unroll 4
        for (int j = 0; j < IPT; j++) {
            const uint64_t i = base + (uint64_t) j * NT + threadIdx.x;
            if (i >= n) break;
            const uint64_t k = keys[i];
            if (k != ~0ull) atomicAdd(&sHist[digitOf(k, shift, mask)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BINS; i += NT) { const unsigned int c = sHist[i]; if (c) atomicAdd(&hist[i], (unsigned long long) c); }
}
// hist[BINS] -> seg[d] = first tuple of segment d (seg[BINS] = all of them), segTile[d] = its first tile (segTile[BINS] = all tiles)
template <int UNUSED = 0>
__global__ __launch_bounds__(BINS) void k_rx_seg_layout(const unsigned long long *__restrict__ hist, unsigned long long *__restrict__ seg, unsigned int *__restrict__ segTile) {
    const unsigned long long c = hist[threadIdx.x];
    unsigned long long tot; unsigned int tt;
    const unsigned long long ex = cdm_block_excl_sum<unsigned long long>(c, tot);
    const unsigned int tiles = (unsigned int) ((c + TILE - 1) / TILE);
    const unsigned int et = cdm_block_excl_sum<unsigned int>(tiles, tt);
    seg[threadIdx.x] = ex; segTile[threadIdx.x] = et;
    if (threadIdx.x == 0) { seg[BINS] = tot; segTile[BINS] = tt; }
}
// digit counts of `passes` passes (bits [beginBit + p BITS, ...) of the tuples, up to endBit) per segment: hist[segment][pass][digit]
struct SegHistArgs { const uint64_t *keys; const unsigned long long *seg; int beginBit, endBit, passes; unsigned long long *hist; };
template <int UNUSED = 0>
__global__ __launch_bounds__(NT) void k_rx_hist_seg(SegHistArgs a) {
    __shared__ unsigned int sHist[2][BINS];
    const uint64_t n = a.seg[BINS];
    // a block takes one contiguous stretch of the tuples and flushes its counters where the stretch crosses into the next segment
    const uint64_t per = ((n + gridDim.x - 1) / gridDim.x + NT - 1) / NT * NT;
    const uint64_t e0 = (uint64_t) blockIdx.x * per, e1 = min(n, e0 + per);
    if (e0 >= e1) return;
    const int first = __syncthreads_count(threadIdx.x < BINS && a.seg[threadIdx.x < BINS ? threadIdx.x : 0] <= e0) - 1;
    for (int i = threadIdx.x; i < 2 * BINS; i += NT) (&sHist[0][0])[i] = 0u;
    __syncthreads();
    uint64_t e = e0;
    for (int sg = first; e < e1; sg++) {
        const uint64_t stop = min(e1, (uint64_t) a.seg[sg + 1]);
        if (stop <= e) continue;
        for (uint64_t i = e + threadIdx.x; i < stop; i += NT) {
            const uint64_t k = a.keys[i];
            for (int p = 0; p < a.passes; p++) {
                const int shift = a.beginBit + p * BITS, bits = min(BITS, a.endBit - shift);
                atomicAdd(&sHist[p][digitOf(k, shift, (1u << bits) - 1u)], 1u);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < a.passes * BINS; i += NT) {
            const unsigned int c = (&sHist[0][0])[i];
            if (c) { atomicAdd(&a.hist[((size_t) sg * a.passes) * BINS + i], (unsigned long long) c); (&sHist[0][0])[i] = 0u; }
        }
        __syncthreads();
        e = stop;
    }
}
// in place: hist[segment][pass][digit] -> first output position of the digit's tuples (block = (segment, pass))
template <int UNUSED = 0>
__global__ __launch_bounds__(BINS) void k_rx_offsets_seg(unsigned long long *hist, int passes, const unsigned long long *__restrict__ seg) {
    unsigned long long *h = hist + (size_t) blockIdx.x * BINS;
    const unsigned long long c = h[threadIdx.x];
    unsigned long long tot;
    const unsigned long long ex = cdm_block_excl_sum<unsigned long long>(c, tot);
    h[threadIdx.x] = seg[blockIdx.x / passes] + ex;
}
// pass-major copy of the per-segment tables: out[pass][segment][digit] (a pass's digitBase is one contiguous [BINS][BINS] table)
template <int UNUSED = 0>
__global__ __launch_bounds__(BINS) void k_rx_seg_tables(const unsigned long long *__restrict__ hist, int passes, unsigned long long *__restrict__ out) {
    const int sg = blockIdx.x / passes, p = blockIdx.x % passes;
    out[((size_t) p * BINS + sg) * BINS + threadIdx.x] = hist[((size_t) sg * passes + p) * BINS + threadIdx.x];
}
// k0: the n slot keys; k1: a second buffer of n keys.  topBit = number of k-mer bits (2k; bit 2k and above are clear in real keys),
// lowBits = key bits left to the on-chip finish.  Needs HEAD_BITS <= topBit, lowBits <= topBit - HEAD_BITS <= SLOT_REM.
// headHist: device, [BINS] head digit counts of the real keys if the writer of k0 counted them (else NULL: counted here).
// Out: segDev (device, [BINS + 1], caller-allocated) = where each head digit's tuples start in the result; live = real tuples;
// result = the buffer (k0 or k1) that holds the sorted slot tuples.  passMs / launches: HIP-event time of the pass launches, summed.
inline int sortSlotKeys(hipStream_t s, int cuCount, uint64_t *k0, uint64_t *k1, uint64_t n, int topBit, int lowBits, const unsigned long long *headHist,
                        unsigned long long *segDev, unsigned long long &live, uint64_t *&result, float *passMs = nullptr, float *launches = nullptr,
                        uint32_t keepLo = 0, uint32_t keepHi = BINS) {
    live = 0; result = k1;
    const int headBits = std::min(BITS, topBit), shift = topBit - headBits, rem = shift - lowBits;
    if (shift > SLOT_REM || rem < 0) { cdm_set_error("slot key sort: %d key bits, %d finished on chip", topBit, lowBits); return CDM_ERR_INVALID; }
    const int segPasses = (rem + BITS - 1) / BITS;
    if (segPasses > 2) { cdm_set_error("slot key sort: %d global key bits behind the head digit (two passes take %d)", rem, 2 * BITS); return CDM_ERR_INVALID; }
    const uint64_t tilesHead = (n + TILE - 1) / TILE;
    DevBuf<unsigned long long> hist, segHist, segTables, status; DevBuf<unsigned int> ticket, segTile;
    if (!hist.alloc(BINS) || !ticket.alloc(4) || !segTile.alloc(BINS + 1)) { cdm_set_error("slot key sort: out of device memory"); return CDM_ERR_HIP; }
    hipMemsetAsync(ticket.p, 0, 16, s);
    if (headHist) {
        hipMemcpyAsync(hist.p, headHist, BINS * 8, hipMemcpyDeviceToDevice, s);
        if (const char *e = cdmGetenv("CDM_SLOT_HIST")) if (!strcmp(e, "check")) {      // tests: the writer's counts against a count of the keys
            DevBuf<unsigned long long> again; unsigned long long a[BINS], b[BINS];
            if (!again.alloc(BINS)) { cdm_set_error("slot key sort: out of device memory"); return CDM_ERR_HIP; }
            hipMemsetAsync(again.p, 0, BINS * 8, s);
            if (n) hipLaunchKernelGGL(k_rx_hist_head<0>, dim3((unsigned) std::min<uint64_t>(tilesHead, (uint64_t) cuCount * 8)), dim3(NT), 0, s, (const uint64_t *) k0, n, shift, headBits, again.p);
            hipMemcpyAsync(a, hist.p, BINS * 8, hipMemcpyDeviceToHost, s); hipMemcpyAsync(b, again.p, BINS * 8, hipMemcpyDeviceToHost, s);
            if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("slot key sort (head histogram check) failed"); return CDM_ERR_HIP; }
            for (int d = 0; d < BINS; d++) if (a[d] != b[d]) { cdm_set_error("slot key sort: head digit %d: the extraction counted %llu tuples, the slots hold %llu", d, a[d], b[d]); return CDM_ERR_HIP; }
        }
    } else {
        hipMemsetAsync(hist.p, 0, BINS * 8, s);
        if (n) hipLaunchKernelGGL(k_rx_hist_head<0>, dim3((unsigned) std::min<uint64_t>(tilesHead, (uint64_t) cuCount * 8)), dim3(NT), 0, s, (const uint64_t *) k0, n, shift, headBits, hist.p);
    }
    // (a rank's range of head digits: the others' counts are not this call's)
    if (keepLo > 0) hipMemsetAsync(hist.p, 0, (size_t) std::min<uint32_t>(keepLo, BINS) * 8, s);
    if (keepHi < (uint32_t) BINS) hipMemsetAsync(hist.p + keepHi, 0, (size_t) (BINS - keepHi) * 8, s);
    hipLaunchKernelGGL(k_rx_seg_layout<0>, dim3(1), dim3(BINS), 0, s, (const unsigned long long *) hist.p, segDev, segTile.p);
    unsigned int tilesSeg = 0;
    hipMemcpyAsync(&live, segDev + BINS, 8, hipMemcpyDeviceToHost, s); hipMemcpyAsync(&tilesSeg, segTile.p + BINS, 4, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("slot key sort (head histogram) failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    if (live > n) { cdm_set_error("slot key sort: internal error: %llu keys counted in %llu slots", live, (unsigned long long) n); return CDM_ERR_HIP; }
    if (live == 0) return CDM_OK;
    if (!status.alloc(std::max<uint64_t>(tilesHead, tilesSeg) * BINS)) { cdm_set_error("slot key sort: out of device memory"); return CDM_ERR_HIP; }
    hipEvent_t ev[6];
    if (passMs) for (int i = 0; i < 2 * (1 + segPasses); i++) hipEventCreate(&ev[i]);
    // ---- the head pass
    {
        hipMemsetAsync(status.p, 0, tilesHead * BINS * 8, s);
        PassArgs<uint64_t, NoValue> pa;
        pa.kin = k0; pa.kout = k1; pa.vin = nullptr; pa.vout = nullptr; pa.n = n; pa.shift = shift; pa.bits = headBits; pa.digitBase = segDev; pa.status = status.p; pa.ticket = ticket.p;
        pa.keepLo = keepLo; pa.keepHi = keepHi;
        if (passMs) hipEventRecord(ev[0], s);
        hipLaunchKernelGGL((k_rx_pass<uint64_t, NoValue, PASS_HEAD>), dim3((unsigned) tilesHead), dim3(NT), 0, s, pa);
        if (passMs) hipEventRecord(ev[1], s);
    }
    // ---- the passes inside the segments
    if (segPasses) {
        if (!segHist.alloc((size_t) BINS * segPasses * BINS) || !segTables.alloc((size_t) BINS * segPasses * BINS)) { cdm_set_error("slot key sort: out of device memory"); return CDM_ERR_HIP; }
        hipMemsetAsync(segHist.p, 0, (size_t) BINS * segPasses * BINS * 8, s);
        SegHistArgs ha; ha.keys = k1; ha.seg = segDev; ha.beginBit = SLOT_KEY_SHIFT + lowBits; ha.endBit = SLOT_KEY_SHIFT + shift; ha.passes = segPasses; ha.hist = segHist.p;
        hipLaunchKernelGGL(k_rx_hist_seg<0>, dim3((unsigned) std::min<uint64_t>((live + TILE - 1) / TILE, (uint64_t) cuCount * 8)), dim3(NT), 0, s, ha);
        hipLaunchKernelGGL(k_rx_offsets_seg<0>, dim3(BINS * segPasses), dim3(BINS), 0, s, segHist.p, segPasses, (const unsigned long long *) segDev);
        hipLaunchKernelGGL(k_rx_seg_tables<0>, dim3(BINS * segPasses), dim3(BINS), 0, s, (const unsigned long long *) segHist.p, segPasses, segTables.p);
        uint64_t *in = k1, *out = k0;
        for (int p = 0; p < segPasses; p++) {
            hipMemsetAsync(status.p, 0, (size_t) tilesSeg * BINS * 8, s);
            PassArgs<uint64_t, NoValue> pa;
            pa.kin = in; pa.kout = out; pa.vin = nullptr; pa.vout = nullptr; pa.n = live;
            pa.shift = SLOT_KEY_SHIFT + lowBits + p * BITS; pa.bits = std::min(BITS, SLOT_KEY_SHIFT + shift - pa.shift);
            pa.digitBase = segTables.p + (size_t) p * BINS * BINS; pa.status = status.p; pa.ticket = ticket.p + 1 + p; pa.seg = segDev; pa.segTile = segTile.p;
            if (passMs) hipEventRecord(ev[2 + 2 * p], s);
            hipLaunchKernelGGL((k_rx_pass<uint64_t, NoValue, PASS_SEG>), dim3(tilesSeg), dim3(NT), 0, s, pa);
            if (passMs) hipEventRecord(ev[3 + 2 * p], s);
            std::swap(in, out);
        }
        result = in;
    }
    const hipError_t e = hipStreamSynchronize(s);
    if (passMs) {
        *passMs = 0.f;
        for (int p = 0; p < 1 + segPasses; p++) { float ms = 0.f; if (e == hipSuccess) hipEventElapsedTime(&ms, ev[2 * p], ev[2 * p + 1]); *passMs += ms; }
        for (int i = 0; i < 2 * (1 + segPasses); i++) hipEventDestroy(ev[i]);
        if (launches) *launches = (float) (1 + segPasses);
    }
    if (e != hipSuccess) { cdm_set_error("slot key sort failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    return CDM_OK;
}

}  // namespace rx
