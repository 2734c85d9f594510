// cyclecheck (src/assembler/cyclecheck.cpp:30-269), SURVEY.md 8(f) rank 4: which contigs close on themselves.
//
// The reference cuts a contig into thirds by k-mer position, sorts the three k-mer lists and merges them: every middle/back
// occurrence of a 22-mer against the FIRST front occurrence, every back occurrence against the FIRST middle occurrence, a hit
// counter per diagonal >= L/3, then the first diagonal whose 1 % band holds more than 0.24 hits per possible k-mer.
// Here: ONE stable radix sort (radix.h) of all contigs' (k-mer, global ordinal) pairs - the ordinals ascend with (contig, position), so a
// contig's occurrences of a k-mer end up adjacent and in position order - and a thread per occurrence that looks BACK in its run:
// the run head (gallop + bisect) is the first front occurrence, a bisect finds the first middle one.  No per-contig sort, no
// LDS limit on the contig length.  Quirks kept (the oracle lists them): position 0 belongs to no third that matters, N is
// letter 4 of a base-4 index (collisions included), the band arithmetic runs in the reference's types.
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"
#include "devutil.h"
#include "scan.h"
#include "radix.h"

namespace {
typedef unsigned long long u64;
constexpr uint32_t CYC_K = 22;

struct CycArgs {
    const uint32_t *codes, *nmask, *woff, *len; const uint8_t *hasN;
    uint32_t n, maxSeqLen;
    const u64 *koff, *hoff;      // [n+1] first k-mer ordinal / first diagonal counter of each contig
    u64 totalK, totalH;          // k-mer positions / diagonal counters of THIS BATCH of contigs [c0, c0 + nc): ordinals and counters are
    uint32_t c0, nc;             // local to it (kBase = koff[c0], hBase = hoff[c0]), so that a DB of any size goes through in batches of
    u64 kBase, hBase;            // fewer than 2^32 positions
    u64 *keys; uint32_t *vals;   // sorted: (k-mer index, ordinal)
    uint32_t *owner;             // contig of every ordinal (the k-mers of different contigs interleave in the sorted array)
    uint32_t *hits, *split;
};
__global__ void k_cyc_sizes(const uint32_t *__restrict__ len, uint32_t n, uint32_t maxSeqLen, u64 *__restrict__ nk, u64 *__restrict__ nh) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const uint32_t L = i < n ? len[i] : 0;
    const bool use = i < n && L >= CYC_K && L < maxSeqLen;          // :107-112: too long a contig is skipped
    nk[i] = use ? L - CYC_K + 1 : 0; nh[i] = use ? 2 * (L / 3) + 1 : 0;
}
__device__ __forceinline__ uint32_t ownerOf(const u64 *__restrict__ off, uint32_t n, u64 g) {     // last i with off[i] <= g
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= g) lo = mid; else hi = mid; }
    return lo;
}
// Indexer::int2index over alphabetSize - 1 = 4 (Indexer.h:76-80): sum of letter_j * 4^j, letters in MMseqs2's order A,C,T,G = 0..3, N = 4
__global__ __launch_bounds__(256) void k_cyc_kmers(CycArgs a) {
    const u64 g = (u64) blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.totalK) return;
    const uint32_t i = a.c0 + ownerOf(a.koff + a.c0, a.nc, a.kBase + g), pos = (uint32_t) (a.kBase + g - a.koff[i]);
    const uint32_t w0 = a.woff[i], last = (a.len[i] + 15) / 16 - 1;
    u64 x = (u64) cdm_window16(a.codes, w0, pos, last) | ((u64) (cdm_window16(a.codes, w0, pos + 16, last) & 0xFFFu) << 32);
    x ^= (x >> 1) & 0x5555555555555555ull;                         // A,C,G,T -> A,C,T,G
    if (a.hasN[i]) {
        const u64 bit = (u64) w0 * 16u + pos;
        const u64 m = (u64) a.nmask[bit >> 5] | ((u64) a.nmask[(bit >> 5) + 1] << 32);
        const uint32_t nw = (uint32_t) (m >> (bit & 31u)) & 0x3FFFFFu;
        const u64 sp = (u64) cdm_spread16(nw) | ((u64) cdm_spread16(nw >> 16) << 32);
        x = (x & ~(sp * 3u)) + (sp << 2);
    }
    a.keys[g] = x; a.vals[g] = (uint32_t) g; a.owner[g] = i;
}
__global__ __launch_bounds__(256) void k_cyc_hits(CycArgs a) {
    const u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 || i >= a.totalK) return;
    const u64 key = a.keys[i];
    if (a.keys[i - 1] != key) return;
    const uint32_t g = a.vals[i], c = a.owner[g];
    const uint32_t base = (uint32_t) (a.koff[c] - a.kBase), pos = g - base, L = a.len[c], third = L / 3;
    if (pos < third + 2 || a.vals[i - 1] < base) return;           // a front occurrence (or position 0) matches nothing before it; alone in its run
    // head of the run (same k-mer, same contig): gallop back, then bisect.  in(j) is monotone on [0, i]
    u64 lo = i - 1, step = 1, out = ~0ull;                         // lo: known inside; out: known outside (or -1)
    while (true) {
        if (lo < step) { out = ~0ull; break; }
        const u64 j = lo - step;
        if (a.keys[j] == key && a.vals[j] >= base) { lo = j; step <<= 1; } else { out = j; break; }
    }
    while (lo - out > 1) { const u64 mid = out + ((lo - out) >> 1); if (a.keys[mid] == key && a.vals[mid] >= base) lo = mid; else out = mid; }   // ~0 + 1 wraps to 0: fine
    u64 first = lo;
    uint32_t pf = a.vals[first] - base;
    if (pf == 0) { first++; pf = a.vals[first] - base; }            // first <= i
    uint32_t *hits = a.hits + (a.hoff[c] - a.hBase);
    if (pf >= 1 && pf <= third + 1 && pos - pf >= third) atomicAdd(&hits[pos - pf - third], 1u);
    if (pos >= 2 * third + 2) {                                      // a back occurrence: also against the first middle one
        u64 l = first, h = i;                                        // first j in [first, i) with position >= third + 2
        while (l < h) { const u64 mid = l + ((h - l) >> 1); if (a.vals[mid] - base >= third + 2) h = mid; else l = mid + 1; }
        if (l < i) {
            const uint32_t pm = a.vals[l] - base;
            if (pm <= 2 * third + 1 && pos - pm >= third) atomicAdd(&hits[pos - pm - third], 1u);
        }
    }
}
// :225-245: the first diagonal whose band of +-1 % of the diagonal's length holds more than 0.24 hits per k-mer position
__global__ __launch_bounds__(256) void k_cyc_band(CycArgs a) {
    const u64 x = (u64) blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.totalH) return;
    const uint32_t hd = a.hits[x];
    if (hd == 0) return;
    const uint32_t c = a.c0 + ownerOf(a.hoff + a.c0, a.nc, a.hBase + x), d = (uint32_t) (a.hBase + x - a.hoff[c]), L = a.len[c], third = L / 3;
    if (d >= 2 * third) return;
    const uint32_t *hits = a.hits + (a.hoff[c] - a.hBase);
    const uint32_t diag = d + third, diaglen = L - diag;
    const uint32_t gap = (uint32_t) ((double) diaglen * 0.01);
    const uint32_t lower = (uint32_t) max(0, (int) (d - gap)), upper = min(d + gap, 2 * third);
    uint32_t band = 0;
    for (uint32_t y = lower; y <= upper; y++) { const uint32_t h = hits[y]; if (h <= hd) band += h; }
    const float rate = __fdiv_rn((float) band, (float) (u64) (diaglen - CYC_K + 1));
    if ((double) rate > 0.24) atomicMin(&a.split[c], d);
}
__global__ void k_cyc_select(const uint32_t *__restrict__ len, const uint32_t *__restrict__ split, uint32_t n, int chop, uint32_t *__restrict__ selCyc, uint32_t *__restrict__ selRest,
                             uint32_t *__restrict__ splitOut) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool cyc = split[i] != 0xFFFFFFFFu;
    const uint32_t at = cyc ? split[i] + len[i] / 3 : 0;
    selCyc[i] = cyc ? (chop ? at : len[i]) : 0xFFFFFFFFu;
    selRest[i] = cyc ? 0xFFFFFFFFu : len[i];
    splitOut[i] = at;
}
}  // namespace

extern "C" int cdm_cyclecheck(cdm_ctx *ctx, const cdm_seqdb *db, uint32_t maxSeqLen, int chopCycle, cdm_seqdb **cyclic, cdm_seqdb **rest, uint32_t *splitHost) {
    if (!ctx || !db || !cyclic) { cdm_set_error("cdm_cyclecheck: NULL argument"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t n = (uint32_t) db->n;
    DevBuf<u64> nk, nh, koff, hoff;
    DevBuf<uint32_t> split, selCyc, selRest, splitOut;
    if (!nk.alloc((size_t) n + 1) || !nh.alloc((size_t) n + 1) || !koff.alloc((size_t) n + 1) || !hoff.alloc((size_t) n + 1) || !split.alloc(n) || !selCyc.alloc(n) ||
        !selRest.alloc(n) || !splitOut.alloc(n)) { cdm_set_error("cdm_cyclecheck: out of device memory"); return CDM_ERR_HIP; }
    hipLaunchKernelGGL(k_cyc_sizes, dim3((n + 256) / 256), dim3(256), 0, s, db->len, n, maxSeqLen, nk.p, nh.p);
    cdmscan::ScanTemp st1, st2;
    if (cdmscan::exclusiveScan<u64>(s, st1, nk.p, koff.p, (size_t) n + 1) != CDM_OK || cdmscan::exclusiveScan<u64>(s, st2, nh.p, hoff.p, (size_t) n + 1) != CDM_OK) return CDM_ERR_HIP;
    u64 totalK = 0, totalH = 0;
    hipMemcpyAsync(&totalK, koff.p + n, 8, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(&totalH, hoff.p + n, 8, hipMemcpyDeviceToHost, s);
    CDM_HIP(hipStreamSynchronize(s));
    CDM_HIP(hipMemsetAsync(split.p, 0xFF, (size_t) n * 4 + 4, s));
    // Batches of contigs with fewer than 2^31 k-mer positions each (CDM_CYC_BATCH=<positions>: tests): the ordinals are 32 bits wide, a
    // launch takes fewer than 2^32 threads, and the sort buffers stay at 28 bytes x 2^31 whatever the DB's size (a workflow's later
    // contig iterations hold 6 G letters and more at 25 M reads).
    u64 batch = 1ull << 31;
    if (const char *e = cdmGetenv("CDM_CYC_BATCH")) { const long long v = atoll(e); if (v > 0) batch = (u64) v; }
    std::vector<uint32_t> cuts(1, 0u);
    if (totalK > batch) {
        std::vector<u64> hk((size_t) n + 1);
        CDM_HIP(hipMemcpyAsync(hk.data(), koff.p, ((size_t) n + 1) * 8, hipMemcpyDeviceToHost, s));
        CDM_HIP(hipStreamSynchronize(s));
        uint32_t start = 0;
        for (uint32_t c = 1; c <= n; c++) if (hk[c] - hk[start] > batch && c - 1 > start) { cuts.push_back(c - 1); start = c - 1; }     // (a contig has fewer than 2^31 positions: --max-seq-len, the 2^22-letter bound of the tuple layouts)
    }
    cuts.push_back(n);
    std::vector<u64> hostK(cuts.size()), hostH(cuts.size());
    for (size_t b = 0; b < cuts.size(); b++) {
        CDM_HIP(hipMemcpyAsync(&hostK[b], koff.p + cuts[b], 8, hipMemcpyDeviceToHost, s));
        CDM_HIP(hipMemcpyAsync(&hostH[b], hoff.p + cuts[b], 8, hipMemcpyDeviceToHost, s));
    }
    CDM_HIP(hipStreamSynchronize(s));
    u64 maxK = 0, maxH = 0;
    for (size_t b = 0; b + 1 < cuts.size(); b++) { maxK = std::max(maxK, hostK[b + 1] - hostK[b]); maxH = std::max(maxH, hostH[b + 1] - hostH[b]); }
    if (maxK >= 0xFFFFFF00ull || maxH >= 0xFFFFFF00ull) { cdm_set_error("cdm_cyclecheck: a batch of contigs with 2^32 k-mer positions (%llu) or diagonals (%llu)", maxK, maxH); return CDM_ERR_UNSUPPORTED; }
    DevBuf<u64> k0, k1; DevBuf<uint32_t> v0, v1, hits, owner;
    if (maxK && (!k0.alloc(maxK) || !k1.alloc(maxK) || !v0.alloc(maxK) || !v1.alloc(maxK) || !hits.alloc(maxH) || !owner.alloc(maxK))) { cdm_set_error("cdm_cyclecheck: out of device memory"); return CDM_ERR_HIP; }
    for (size_t b = 0; b + 1 < cuts.size(); b++) {
        const u64 bK = hostK[b + 1] - hostK[b], bH = hostH[b + 1] - hostH[b];
        if (!bK) continue;
        CycArgs a = {db->codes, db->nmask, db->woff, db->len, db->hasN, n, maxSeqLen, koff.p, hoff.p, bK, bH, cuts[b], cuts[b + 1] - cuts[b], hostK[b], hostH[b], nullptr, nullptr, nullptr, nullptr, split.p};
        CDM_HIP(hipMemsetAsync(hits.p, 0, (bH + 1) * 4, s));
        a.keys = k0.p; a.vals = v0.p; a.hits = hits.p; a.owner = owner.p;
        hipLaunchKernelGGL(k_cyc_kmers, CDM_GRID((bK + 255) / 256, 256), dim3(256), 0, s, a);
        bool inFirst = true;            // stable radix sort on the 46 index bits (radix.h)
        if (int rc = rx::sortPairs<uint64_t, uint32_t>(s, ctx->cuCount, reinterpret_cast<uint64_t *>(k0.p), reinterpret_cast<uint64_t *>(k1.p), v0.p, v1.p, (uint64_t) bK, 0, 2 * CYC_K + 2, inFirst)) return rc;
        a.keys = inFirst ? k0.p : k1.p; a.vals = inFirst ? v0.p : v1.p;
        hipLaunchKernelGGL(k_cyc_hits, CDM_GRID((bK + 255) / 256, 256), dim3(256), 0, s, a);
        if (bH) hipLaunchKernelGGL(k_cyc_band, CDM_GRID((bH + 255) / 256, 256), dim3(256), 0, s, a);
    }
    if (n) hipLaunchKernelGGL(k_cyc_select, dim3((n + 255) / 256), dim3(256), 0, s, db->len, split.p, n, chopCycle, selCyc.p, selRest.p, splitOut.p);
    if (splitHost && n) CDM_HIP(hipMemcpyAsync(splitHost, splitOut.p, (size_t) n * 4, hipMemcpyDeviceToHost, s));
    CDM_HIP(hipStreamSynchronize(s));
    cdm_seqdb *c = nullptr, *r = nullptr;
    int rc = cdm_seqdb_select(ctx, db, selCyc.p, 0, &c);            // DBWriter::writeData's default: wasExtended = false (:251)
    if (rc == CDM_OK && rest) rc = cdm_seqdb_select(ctx, db, selRest.p, -1, &r);
    if (rc != CDM_OK) { if (c) cdm_seqdb_free(c); if (r) cdm_seqdb_free(r); return rc; }
    *cyclic = c; if (rest) *rest = r;
    return CDM_OK;
}
