// kmermatcher's second sort and vote on AGGREGATED group tuples (kmermatcher.cpp:431 sort by (rep, id, diagonal), :815-930 vote).
//
// Two overlapping reads share ~20 k-mers, all on one diagonal: the 4 G group tuples of 50 M reads hold only ~0.2 G distinct
// (representative, member, diagonal) triples, and the vote needs nothing but, per triple, how many tuples carry it and the strand of
// the last of them in k-mer order.  So a representative's tuples are not sorted but COUNTED: a block expands the run records of a
// unit of representatives (runsort.h) straight into an LDS hash table keyed by (ordinal, id, diagonal) - count += 1, last =
// max(position << 1 | strand) -, sorts the few hundred distinct entries with the register network and writes them, 12 bytes each,
// with a per-representative directory.  The sorted 33 GB key array of the tuple path is never written, nor read again by the vote:
// k_vote_entries walks a representative's entries exactly as writeKmerMatcherResult walks its tuples - the running diagonal count,
// ">=" so that the later of two equally frequent diagonals wins, the walk that runs on into the NEXT representatives' entries while
// the target id stays the same (:875-887), and at the very end into the left-over tuples (k_stale_tail).
//
// What does not fit a table (a unit with more than AG_D distinct triples) or a unit (a representative with more than 2048 tuples)
// goes the tuple way - k_unit_sort's hard list, k_block_sort, the radix sort - into the sorted array, and k_rle_segment turns those
// segments into entries.  The tuple path itself (k_unit_sort for everything, k_seg_count / k_seg_place) stays: the multi-GPU split
// votes on it, CDM_KMER_VOTE=tuples selects it, and it is what a run falls back to when the entry buffer overflows.
#pragma once
#include "runsort.h"

namespace aggv {

struct Ent { uint32_t id, diag, cs; };          // cs = tuples << 1 | strand bit of the last tuple (k-mer order) that carries the triple
constexpr uint32_t PENDING = 0xFFFFFFFFu;

// ---- segments: segment g = the g-th representative that has group tuples, in ascending order (the sorted records' order)
__global__ __launch_bounds__(1024) void k_seg_flags(const uint32_t *__restrict__ recRep, uint64_t nRec, uint32_t *__restrict__ flag) {
    const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (j <= nRec) flag[j] = (j < nRec && (j == 0 || recRep[j - 1] != recRep[j])) ? 1u : 0u;
}
// segOfRec (in: exclusive scan of the flags) -> segment of every record; segRep / segFirstRec per segment
__global__ __launch_bounds__(1024) void k_seg_fill(const uint32_t *__restrict__ recRep, uint64_t nRec, uint32_t *__restrict__ segOfRec, uint32_t *__restrict__ segRep,
                                                   unsigned long long *__restrict__ segFirstRec, uint32_t *__restrict__ entCnt) {
    const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nRec) return;
    const uint32_t ex = segOfRec[j];
    if (j == nRec) { segFirstRec[ex] = nRec; return; }
    const bool start = j == 0 || recRep[j - 1] != recRep[j];
    const uint32_t g = start ? ex : ex - 1;
    segOfRec[j] = g;
    if (start) { segRep[g] = recRep[j]; segFirstRec[g] = j; entCnt[g] = PENDING; }
}

struct AggArgs {
    const uint64_t *keys; const uint64_t *recVal; const unsigned long long *dst; uint64_t nRec;     // k-mer-ordered group keys, sorted records, their offsets
    const uint32_t *segOfRec, *segRep; const unsigned long long *segFirstRec; uint64_t nSeg;
    unsigned long long *entOff; uint32_t *entCnt; unsigned long long *perRep;
    Ent *ent; unsigned long long *cursor; unsigned long long cap; unsigned int *overflow;
    int repShift, diagBits; uint32_t idBits;
    uint32_t maxD;                  // most distinct triples a unit may bring (AG_D; CDM_AGG_D lowers it: tests reach the other path)
    const uint64_t *sorted;         // the tuple path's output (k_rle_segment reads the segments the tuple sorters finished)
    const unsigned long long *list; const unsigned int *count;      // the units of this size class (start, end, first record)
    bucket::BigList hard;
    unsigned int nextClass = 0;     // (host side: which size class the next launch is for)
    int wideWord = 0;               // (host side: the entry sort needs the 128-bit word)
};
constexpr int AG_H = 1024, AG_D = 512, AG_IDX = 9, AG_ORD = 11;
// Room in the entry array is taken from the one global cursor in CHUNKS of this many entries, a block (k_unit_agg) or a wave
// (k_rle_segment) handing them out to its units from there: one atomic per unit on a single word - 1.9 M units and 0.5 M segments in the
// 50 M-read step - ran at the ~88 atomics per microsecond such a word takes, and that WAS the two kernels' time (round 5: k_rle_segment
// 6.3 ms for 0.52 M segments of 34 tuples on average, whether a block or a wave took a segment).  What a chunk's end leaves unused
// stays unused: the caller's capacity holds a chunk per block / wave on top (aggChunkSlack).
constexpr unsigned long long AG_CHUNK = 2048;
inline unsigned long long aggChunkSlack(unsigned long long units, unsigned long long segments, int cuCount) {
    const unsigned long long g = (unsigned long long) cuCount * 64;
    return (3 * std::min(units + 1, g) + std::min(segments + 1, g)) * AG_CHUNK;
}
static_assert((1 << AG_IDX) >= AG_D && (1 << AG_ORD) > runsort::U_T, "aggregation geometry");
__device__ __forceinline__ uint32_t aggHash(uint64_t k) { return (uint32_t) ((k * 0x9E3779B97F4A7C15ull) >> 40); }

// ITEMS: tuples a thread fetches before it starts inserting (all loads of a round are in flight together; CAP / NT of the size class)
#ifndef CDM_AGG_MINW
#define CDM_AGG_MINW 4
#endif
// (waves per SIMD the register allocation leaves room for: unbounded the kernel takes 112-130 VGPRs - the register network of the
// entry sort - and runs 3-4 waves per SIMD where its LDS allows 5; swept 3 / 4 / 5 / 6: 85.3 / 83.8 / 87.0 / 84.6 ms of sort 2 - within the noise; halving the
// occupancy with an LDS pad costs 30 ms: the kernel runs on the latency of its gathers and LDS round trips)
// W: word of the entry sort, (ordinal, id, diagonal, index in the compacted table) - 64 bits while that fits, 128 beyond
template <int NT, int ITEMS, typename W = uint64_t>
__global__ __launch_bounds__(NT, CDM_AGG_MINW) void k_unit_agg(AggArgs a) {
    using namespace runsort;
    // the hash table; once its entries are compacted the same memory holds the per-ordinal directory of the unit
    __shared__ __align__(8) unsigned char sTab[AG_H * 16];
    unsigned long long *tKey = reinterpret_cast<unsigned long long *>(sTab);
    unsigned int *tCnt = reinterpret_cast<unsigned int *>(sTab + AG_H * 8), *tLast = reinterpret_cast<unsigned int *>(sTab + AG_H * 12);
    unsigned int *oFirst = reinterpret_cast<unsigned int *>(sTab), *oEnt = oFirst + (U_T + 2), *oHit = oEnt + (U_T + 2);
    static_assert(3 * (U_T + 2) * 4 <= AG_H * 16, "the directory fits the table's memory");
    __shared__ unsigned long long dKey[AG_D];
    __shared__ unsigned int dCnt[AG_D], dLast[AG_D];
    __shared__ uint16_t sPerm[AG_D];
    __shared__ uint64_t sGStart[NT];
    __shared__ uint32_t sGOff[NT], sGSeg[NT];
    __shared__ unsigned int sDistinct, sOver, sMaxOrd;
    __shared__ unsigned long long sBase, sChunkAt, sChunkEnd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned int nUnits = *a.count;
    if (tid == 0) { sChunkAt = 0; sChunkEnd = 0; }             // (the loop's first barrier publishes them)
    const int lowBits = a.repShift - 1;                         // id and diagonal
    const uint64_t lowMask = (1ull << lowBits) - 1ull;
    for (unsigned int item = blockIdx.x; item < nUnits; item += gridDim.x) {
        const uint64_t base = a.list[3 * (size_t) item];
        const int m = (int) (a.list[3 * (size_t) item + 1] - base);          // the unit: whole segments, m tuples
        const uint64_t rec0 = a.list[3 * (size_t) item + 2];
        const uint32_t g0 = a.segOfRec[rec0];
        for (int i = tid; i < AG_H; i += NT) { tKey[i] = ~0ull; tCnt[i] = 0u; tLast[i] = 0u; }
        if (tid == 0) { sDistinct = 0; sOver = 0; sMaxOrd = 0; }
        __syncthreads();
        // ---- the unit's records NT at a time (as k_unit_sort stages them); every tuple goes into the table
        const unsigned long long endG = base + (uint64_t) m;
        for (uint64_t c0 = rec0; c0 < a.nRec; c0 += NT) {
            const uint64_t j = c0 + tid;
            const unsigned long long d = j < a.nRec ? a.dst[j] : ~0ull;
            const bool valid = d < endG;
            sGOff[tid] = valid ? (uint32_t) (d - base) : (uint32_t) m;
            sGStart[tid] = valid ? (a.recVal[j] >> RUN_CNT_BITS) : 0ull;
            sGSeg[tid] = valid ? a.segOfRec[j] - g0 : 0u;
            const int nr = __syncthreads_count(valid);
            if (nr == 0) break;
            const int e0 = (int) sGOff[0], e1 = (nr == NT) ? (int) min((unsigned long long) m, (j0Next(a.dst, a.nRec, c0 + NT) - base)) : m;
            if (tid == 0) sMaxOrd = max(sMaxOrd, sGSeg[nr - 1]);                    // (segments ascend with the records)
            for (int eb = e0; eb < e1; eb += NT * ITEMS) {
                uint64_t key[ITEMS]; uint32_t ordv[ITEMS];
#pragma unroll
                for (int k = 0; k < ITEMS; k++) {
                    const int e = eb + k * NT + tid;
                    if (e < e1) {
                        int r = 0;
#pragma unroll
                        for (int st = NT / 2; st > 0; st >>= 1) if (r + st < nr && (int) sGOff[r + st] <= e) r += st;
                        key[k] = a.keys[sGStart[r] + (uint64_t) (e - (int) sGOff[r])];
                        ordv[k] = sGSeg[r];
                    }
                }
#pragma unroll
                for (int k = 0; k < ITEMS; k++) {
                    const int e = eb + k * NT + tid;
                    if (e < e1) {
                        const uint64_t hk = ((uint64_t) ordv[k] << lowBits) | ((key[k] >> 1) & lowMask);
                        const unsigned int last = ((unsigned int) e << 1) | (unsigned int) (key[k] & 1ull);
                        uint32_t h = aggHash(hk) & (AG_H - 1);
                        int probe = 0;
                        for (; probe < AG_H; probe++) {
                            // (a triple is carried by ~6 tuples on average: five of six find their slot taken by their own key - a plain read
                            // tells, the compare-and-swap is for the slot that looks empty)
                            unsigned long long old = __hip_atomic_load(&tKey[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (old == ~0ull) {
                                old = atomicCAS(&tKey[h], ~0ull, (unsigned long long) hk);
                                if (old == ~0ull) { atomicAdd(&sDistinct, 1u); old = hk; }
                            }
                            if (old == hk) { atomicAdd(&tCnt[h], 1u); atomicMax(&tLast[h], last); break; }
                            h = (h + 1) & (AG_H - 1);
                        }
                        if (probe == AG_H) sOver = 1u;
                    }
                }
            }
            const bool stop = __syncthreads_or(sDistinct > a.maxD || sOver != 0u);
            if (stop || nr < NT) break;
        }
        __syncthreads();
        if (sDistinct > a.maxD || sOver) {         // too many distinct triples for the table: the tuple path sorts this unit, k_rle_segment reads it
            if (tid == 0) a.hard.add(base, base + (uint64_t) m);
            __syncthreads();
            continue;
        }
        // ---- compact the table
        unsigned int c = 0;
        for (int i = tid; i < AG_H; i += NT) c += tKey[i] != ~0ull;
        unsigned int D;
        unsigned int pos = cdm_block_excl_sum<unsigned int>(c, D);
        for (int i = tid; i < AG_H; i += NT) if (tKey[i] != ~0ull) { dKey[pos] = tKey[i]; dCnt[pos] = tCnt[i]; dLast[pos] = tLast[i]; pos++; }
        __syncthreads();
        // ---- sort the D entries by (ordinal, id, diagonal): one wave, the register network of bucket.h
        if (wave == 0) {
            bucket::sortGroup<W>((int) D, lane,
                [&](int i) { return ((W) dKey[i] << AG_IDX) | (W) i; },
                [&](auto &v) {
                    constexpr int R = sizeof(v) / sizeof(v[0]);
#pragma unroll
                    for (int r = 0; r < R; r++) { const int p = lane * R + r; if (p < (int) D) sPerm[p] = (uint16_t) ((uint32_t) v[r] & ((1u << AG_IDX) - 1u)); }
                });
        }
        const unsigned int nOrd = sMaxOrd + 1u;
        for (unsigned int o = tid; o < nOrd; o += NT) { oFirst[o] = 0xFFFFFFFFu; oEnt[o] = 0u; oHit[o] = 0u; }      // (the table's memory: its entries are in d* now)
        if (tid == 0) {
            if (sChunkAt + D > sChunkEnd) { const unsigned long long want = max(AG_CHUNK, (unsigned long long) D); sChunkAt = atomicAdd(a.cursor, want); sChunkEnd = sChunkAt + want; }
            const unsigned long long b = sChunkAt;
            sChunkAt += D;
            if (b + D > a.cap) { atomicExch(a.overflow, 1u); sBase = ~0ull; } else sBase = b;
        }
        __syncthreads();
        // ---- the directory: first entry, entries and hit-producing members of every ordinal
        const uint64_t idMask = (1ull << a.idBits) - 1ull, diagMask = (1ull << a.diagBits) - 1ull;
        for (unsigned int p = tid; p < D; p += NT) {
            const uint64_t hk = dKey[sPerm[p]], prev = p ? dKey[sPerm[p - 1]] : ~0ull;
            const uint32_t o = (uint32_t) (hk >> lowBits);
            atomicMin(&oFirst[o], p);
            atomicAdd(&oEnt[o], 1u);
            if (p == 0 || (hk >> a.diagBits) != (prev >> a.diagBits)) {                 // first entry of an (ordinal, id)
                const uint32_t id = (uint32_t) ((hk >> a.diagBits) & idMask);
                if (id != a.segRep[g0 + o]) atomicAdd(&oHit[o], 1u);                    // self tuples give no hit (:898-903)
            }
        }
        __syncthreads();
        const unsigned long long b = sBase;
        if (b != ~0ull) {
            for (unsigned int p = tid; p < D; p += NT) {
                const int i = sPerm[p]; const uint64_t hk = dKey[i];
                Ent e; e.id = (uint32_t) ((hk >> a.diagBits) & idMask); e.diag = (uint32_t) (hk & diagMask); e.cs = (dCnt[i] << 1) | (dLast[i] & 1u);
                a.ent[b + p] = e;
            }
            for (unsigned int o = tid; o < nOrd; o += NT) {
                a.entOff[g0 + o] = b + oFirst[o]; a.entCnt[g0 + o] = oEnt[o];
                a.perRep[a.segRep[g0 + o]] = oHit[o];
            }
        }
        __syncthreads();
    }
}

// the segments no k_unit_agg finished
__global__ __launch_bounds__(1024) void k_pending_list(const uint32_t *__restrict__ entCnt, uint64_t nSeg, uint32_t *__restrict__ list, unsigned int *__restrict__ cnt) {
    const uint64_t g = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const bool p = g < nSeg && entCnt[g] == PENDING;
    const uint32_t q = cdm_block_append(cnt, p);
    if (p) list[q] = (uint32_t) g;
}
// one WAVE per listed segment: its tuples stand sorted in a.sorted[dst[first record] .. dst[first record of the next segment]) -> entries.
// (A block per segment until round 5: the segments of the units that overflow the aggregation table are many and short - a few dozen
// tuples -, and 256 threads with two block-wide scans per 256 tuples spent 6.4 ms of the 50 M-read step on them.)
__global__ __launch_bounds__(256) void k_rle_segment(AggArgs a, const uint32_t *__restrict__ list, const unsigned int *__restrict__ nList) {
    const unsigned int n = *nList;
    const int lane = threadIdx.x & 63;
    const uint64_t idMask = (1ull << a.idBits) - 1ull, diagMask = (1ull << a.diagBits) - 1ull;
    const uint64_t below = (1ull << lane) - 1ull;
    unsigned long long chunkAt = 0, chunkEnd = 0;               // (lane 0's)
    for (unsigned int it = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); it < n; it += gridDim.x * (blockDim.x >> 6)) {
        const uint32_t g = list[it];
        const uint64_t s = a.dst[a.segFirstRec[g]], e = a.dst[a.segFirstRec[g + 1]];
        const uint32_t rep = a.segRep[g];
        unsigned int runs = 0, hits = 0;
        for (uint64_t i = s + lane; i < e; i += 64) {
            const uint64_t k = a.sorted[i], p = i > s ? a.sorted[i - 1] : ~0ull;
            if (i == s || (k >> 1) != (p >> 1)) runs++;
            if (i == s || (k >> (a.diagBits + 1)) != (p >> (a.diagBits + 1))) hits += ((uint32_t) ((k >> (a.diagBits + 1)) & idMask) != rep);
        }
        const unsigned int D = (unsigned int) cdm_wave_sum((int) runs), H = (unsigned int) cdm_wave_sum((int) hits);
        unsigned long long b = 0;
        if (lane == 0) {
            if (chunkAt + D > chunkEnd) { const unsigned long long want = max(AG_CHUNK, (unsigned long long) D); chunkAt = atomicAdd(a.cursor, want); chunkEnd = chunkAt + want; }
            b = chunkAt; chunkAt += D;
            if (b + D > a.cap) { atomicExch(a.overflow, 1u); b = ~0ull; }
            else { a.entOff[g] = b; a.entCnt[g] = D; a.perRep[rep] = H; }
        }
        b = (unsigned long long) __shfl((long long) b, 0, 64);
        if (b == ~0ull) continue;
        unsigned long long run = 0;                              // entries written so far
        for (uint64_t t0 = s; t0 < e; t0 += 64) {
            const uint64_t i = t0 + lane;
            bool start = false; uint64_t k = 0;
            if (i < e) { k = a.sorted[i]; start = i == s || (k >> 1) != (a.sorted[i - 1] >> 1); }
            const uint64_t m = __ballot(start);
            if (start) {
                // end of the run: the next start in this stretch of 64, or - behind it - the first index whose triple differs
                const uint64_t later = lane < 63 ? (m >> (lane + 1)) : 0ull;
                uint64_t hi;
                if (later) hi = i + (uint64_t) __ffsll((unsigned long long) later);
                else {
                    uint64_t lo = min(t0 + 63, e - 1); hi = e;          // (the stretch's last tuple belongs to this run)
                    while (hi - lo > 1) { const uint64_t mid = lo + ((hi - lo) >> 1); if ((a.sorted[mid] >> 1) == (k >> 1)) lo = mid; else hi = mid; }
                }
                Ent en; en.id = (uint32_t) ((k >> (a.diagBits + 1)) & idMask); en.diag = (uint32_t) ((k >> 1) & diagMask);
                en.cs = ((uint32_t) (hi - i) << 1) | (uint32_t) (a.sorted[hi - 1] & 1ull);
                a.ent[b + run + (unsigned long long) __popcll(m & below)] = en;
            }
            run += (unsigned long long) __popcll(m);
        }
    }
}

// ---- the vote on entries: one thread per segment, hits written behind the representative's self hit
struct VoteEntArgs {
    const Ent *ent; const unsigned long long *entOff; const uint32_t *entCnt; const uint32_t *segRep; uint64_t nSeg;
    const uint64_t *hitOff;     // per sequence: its self hit, then its hits
    const uint32_t *stale;      // k_stale_tail's list
    int diagBias;
    // Multi-GPU runs (a rank votes on the representatives it owns): what the scan of this rank's LAST target runs into is the head of the
    // next ranks' entries (k_head_entries) - cont[0] WORDS, two per entry (biased diagonal | "reverse" << 31, tuples), that apply if
    // the target is cont[1]; only if cont[2] is set does the scan go on into the left-over tuples after them.  NULL on one device.
    const uint32_t *cont = nullptr;
};
constexpr int HEAD_WORDS = 2048;        // = kmermatch.hip CONT_CAP: the words of a head the ranks exchange
// The head of a rank's entries: from its first entry on while the member id stays the same, whatever the representative (what a scan
// coming in from the rank in front runs through, kmermatcher.cpp:875-887).  out[0] = words (2 per entry; HEAD_WORDS + 1: longer than
// the list), out[1] = that id, out[2] = 1 if the head is everything this rank holds, out[3..] the entries, out[HEAD_WORDS + 3] = the
// member id of the rank's LAST entry.
__global__ void k_head_entries(const Ent *__restrict__ ent, const unsigned long long *__restrict__ entOff, const uint32_t *__restrict__ entCnt, uint64_t nSeg, uint32_t *__restrict__ out) {
    if (nSeg == 0) { out[0] = 0; out[1] = 0; out[2] = 1; out[HEAD_WORDS + 3] = 0; return; }
    const uint32_t id = ent[entOff[0]].id;
    uint32_t words = 0; bool whole = true;
    for (uint64_t g = 0; g < nSeg && whole && words <= (uint32_t) HEAD_WORDS; g++) {
        const Ent *e = ent + entOff[g]; const uint32_t c = entCnt[g];
        for (uint32_t j = 0; j < c; j++) {
            if (e[j].id != id) { whole = false; break; }
            if (words + 2 <= (uint32_t) HEAD_WORDS) { out[3 + words] = e[j].diag | ((e[j].cs & 1u) ? 0u : 1u << 31); out[4 + words] = e[j].cs >> 1; words += 2; }
            else { words = (uint32_t) HEAD_WORDS + 1; whole = false; break; }
        }
    }
    out[0] = words; out[1] = id; out[2] = whole ? 1u : 0u;
    const uint64_t gl = nSeg - 1;
    out[HEAD_WORDS + 3] = ent[entOff[gl] + entCnt[gl] - 1].id;
}
struct Walk {
    uint32_t prevDiag = 0, diagCnt = 0, maxDiag = 0, diagonal = 0, top = 0; int bestRev = 0; bool any = false;
    __device__ __forceinline__ void add(uint32_t d, uint32_t c, int rev) {        // c tuples of diagonal d, the last of them on strand `rev`
        diagCnt = (any && prevDiag == d) ? diagCnt + c : c;
        if (diagCnt >= maxDiag) { diagonal = d; maxDiag = diagCnt; bestRev = rev; }
        prevDiag = d; top += c; any = true;
    }
};
template <typename HitT>
__global__ __launch_bounds__(256) void k_vote_entries(VoteEntArgs a, HitT *__restrict__ out) {
    const uint64_t g = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.nSeg) return;
    const uint32_t rep = a.segRep[g], cnt = a.entCnt[g];
    const Ent *e = a.ent + a.entOff[g];
    uint64_t pos = a.hitOff[rep] + 1;
    uint32_t p = 0;
    while (p < cnt) {
        const uint32_t id = e[p].id;
        if (id == rep) { do p++; while (p < cnt && e[p].id == id); continue; }
        Walk w;
        do { const Ent x = e[p]; w.add(x.diag, x.cs >> 1, (x.cs & 1u) ? 0 : 1); p++; } while (p < cnt && e[p].id == id);
        if (p == cnt) {
            // the representative's last target: the reference's scan goes on while the sequence id stays the same, whatever the
            // representative (kmermatcher.cpp:875-887) - into the first entries of the next segments, and past the last of them
            // into the left-over tuples
            bool done = false;
            for (uint64_t g2 = g + 1; g2 < a.nSeg && !done; g2++) {
                const Ent *f = a.ent + a.entOff[g2]; const uint32_t c2 = a.entCnt[g2];
                uint32_t j = 0;
                for (; j < c2 && f[j].id == id; j++) w.add(f[j].diag, f[j].cs >> 1, (f[j].cs & 1u) ? 0 : 1);
                done = j < c2;
            }
            bool intoStale = !done;
            if (!done && a.cont) {                      // the end of this rank's entries: on into the next ranks' heads
                if (id == a.cont[1]) for (uint32_t j = 0; j + 1 < a.cont[0]; j += 2) w.add(a.cont[3 + j] & 0x7FFFFFFFu, a.cont[4 + j], (int) (a.cont[3 + j] >> 31));
                intoStale = a.cont[2] != 0u;
            }
            if (intoStale && id == a.stale[1]) {
                const uint32_t m = a.stale[0];
                for (uint32_t j = 0; j < m; j++) w.add(a.stale[2 + j] + (uint32_t) a.diagBias, 1u, 0);
            }
        }
        HitT h;
        h.target = id; h.score = w.bestRev ? -(int) w.top : (int) w.top; h.diagonal = (int) (short) ((int) w.diagonal - a.diagBias);
        out[pos++] = h;
    }
}

}  // namespace aggv
