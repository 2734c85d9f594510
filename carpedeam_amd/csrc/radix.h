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

template <typename K, typename V>
struct PassArgs {
    const K *kin; K *kout; const V *vin; V *vout; uint64_t n;
    int shift, bits;
    const unsigned long long *digitBase;        // [BINS]
    unsigned long long *status;                 // [tiles][BINS], zeroed
    unsigned int *ticket;                       // zeroed
};
template <typename K, typename V>
__global__ __launch_bounds__(NT) void k_rx_pass(PassArgs<K, V> a) {
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
    const uint64_t tile = sTile, base = tile * TILE;
    const int items = (int) min((uint64_t) TILE, a.n - base);
    const uint32_t mask = (1u << a.bits) - 1u;
    // ---- load: wave w owns [w * 64 IPT, (w + 1) * 64 IPT), round j its j-th 64 pairs
    K key[IPT]; V val[IPT]; uint16_t pos[IPT];
    const int w0 = wave * 64 * IPT + lane;
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const int li = w0 + 64 * j;
        if (li < items) { key[j] = a.kin[base + li]; if constexpr (HasValue<V>::value) val[j] = a.vin[base + li]; } else key[j] = 0;
    }
    // ---- rank inside the wave's stream, round by round
    uint16_t *cntW = sCnt[wave];
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const bool valid = w0 + 64 * j < items;
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
        __hip_atomic_store(st, (tile == 0 ? ST_PREFIX : ST_AGG) | (unsigned long long) run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    K *sK = reinterpret_cast<K *>(sBuf);
    // ---- the keys into the exchange buffer: slot = digits in front + waves in front + rank (needs nothing of the other tiles)
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        if (w0 + 64 * j < items) {
            const uint32_t d = digitOf(key[j], a.shift, mask);
            pos[j] = (uint16_t) ((uint32_t) sTileOff[d] + (uint32_t) sCnt[wave][d] + (uint32_t) pos[j]);
            sK[pos[j]] = key[j];
        }
    }
    // ---- the tiles in front: chained scan with decoupled look-back, after the exchange so that the tiles in front had that time to
    // publish their prefix; LB status words are fetched per round trip
    if (isDigit) {
        unsigned long long exclG = 0;
        if (tile != 0) {
            const unsigned long long *q = st - BINS;
            uint64_t left = tile;
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
        sGlobal[tid] = a.digitBase[tid] + exclG - ex;       // output position of the pair at exchange slot p with this digit: sGlobal + p
    }
    __syncthreads();
    uint16_t dig[IPT];
#pragma unroll
    for (int r = 0; r < IPT; r++) {
        const int p = tid + NT * r;
        if (p < items) {
            const K k = sK[p];
            dig[r] = (uint16_t) digitOf(k, a.shift, mask);
            a.kout[sGlobal[dig[r]] + (unsigned long long) p] = k;
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

}  // namespace rx
