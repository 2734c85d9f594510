// ancient_contig_merge, the queue and the extension loop on the device (src/assembler/ancientContigsResults.cpp:25-70, 187-470) - round 5.
//
// Until round 4 everything behind the per-record column counts (contig.hip k_contig_stats) ran on the host: statistics, records and
// the whole DB came down, 16 threads ran the priority queues, the grown contigs were packed and went up again - 12.5 s of the last
// iteration of the 25 M-read workflow, 47 of its 69 s in all (profiles/r05_config5_25M_laps_before.txt).  Now the host sees counters.
//
// What made the queue a host matter is its comparator: a Beta-posterior series whose terms are lgammaf / logf / log / exp of the C
// library, whose last bits order the queue, and which is no strict weak ordering - so the order is whatever libstdc++'s heap makes
// of those very values.  Both are restated here without approximating anything that decides:
//   * lgammaf and logf are read from TABLES OF THE C LIBRARY'S OWN VALUES over every float the comparator can hand them ([1, 2^19)
//     resp. [1, 2^20): 19 + 20 binades of 2^23 floats, 1.3 GB in HBM, filled once per process by host/contigmerge.cpp and copied to
//     each device), the double log of the series index from a table over the integers, deamMatches' length prior likewise; every
//     other operation of the comparator and of the gate is IEEE float / double arithmetic, written in the reference's types and
//     order (this file is compiled with -ffp-contract=off; the one fused multiply-add g++ makes on the host is spelled out on both
//     sides);
//   * exp, the one library call with an argument no table can cover, is the device's - both are within an ulp of the true value, the
//     series sums at most yMis such terms, and a comparison whose sum lands within 1e-12 of 0.45 or 0.55 (probability ~4e-12 per
//     comparison) hands its QUERY back to the host code, as does an argument beyond the tables and a sequence with letters beyond
//     ACGTN in play: cdm_contig_merge_host runs for those queries alone (contig.hip);
//   * the heap is libstdc++'s: std::push_heap (__push_heap) and std::pop_heap (__adjust_heap, then __push_heap) step for step, on the
//     query's own stretch of a record-index array.
//
// The extension loop runs in ROUNDS over all queries at once.  A round of a query is one pass of the reference's outer loop (:276-470):
// pop until the queue is empty - the first fitting candidate to the right and to the left each donate a fragment, later ones are
// parked -, then the parked hits are re-aligned on their diagonal against the grown query and pushed again.  Per round:
//   k_cq_round   a thread per active query: pushes (round 0: the gated records; later: the parked hits that still pass), the pop loop,
//                the round's (at most two) fragments, the parked list;
//   k_cq_grow    a wave per query that grew: its new letters = [left fragment] old [right fragment], 16 bases per lane and step, from
//                the input DB (targets, reverse-complemented as useReverse says) and the previous version, into this round's buffer;
//   k_cq_parked  a wave per parked hit: the two counts of updateNuclAlignment / getRYSeqId over the diagonal's overlap (XOR of 2-bit
//                windows), the new coordinates, the filter.
// Queries leave the active list as their queue runs dry; the rounds end when it is empty.  The grown contigs are gathered into a DB
// on the device and overlaid on the input (api.hip cdm_seqdb_overlay): nothing but a list of query indices and the flags comes down.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"
#include "devutil.h"
#include "scan.h"
#include "contigqueue.h"

void cdm_contig_host_tables(float *lgam, int lgTop, float *lf, int lfTop, double *logInt, size_t nInt, double *lenPrior);      // host/contigmerge.cpp

namespace {
constexpr int CQ_LG_TOP = 19, CQ_LF_TOP = 20;           // lgammaf over [1, 2^19), logf over [1, 2^20)
constexpr uint32_t CQ_INT_N = 1u << 20;                 // log of the integers below
constexpr uint32_t CQ_ONE = 0x3F800000u;
constexpr double CQ_BAND = 1e-12;                       // |sum - 0.45|, |sum - 0.55| below this: the host decides (see above)
constexpr uint32_t CQ_NONE = 0xFFFFFFFFu;
constexpr int CQ_MAX_ROUNDS = 1 << 16;

struct CqTables { const float *lgam = nullptr, *lf = nullptr; const double *logInt = nullptr, *lenPrior = nullptr; };

// ------------------------------------------------------------------------------------------------ the tables, once per process / device
struct HostTables { HostBuf<float> lgam, lf; HostBuf<double> logInt, lenPrior; bool ready = false, failed = false; };
std::mutex gTabMu;
HostTables gHost;
CqTables gDev[64];
bool gDevReady[64];

int ensureTables(int device, CqTables &out, std::string &err) {
    std::lock_guard<std::mutex> lock(gTabMu);
    if (device < 0 || device >= 64) { err = "device index beyond 63"; return CDM_ERR_INVALID; }
    if (!gDevReady[device]) {
        const size_t nLg = (size_t) CQ_LG_TOP << 23, nLf = (size_t) CQ_LF_TOP << 23;
        if (!gHost.ready) {
            if (!gHost.lgam.alloc(nLg) || !gHost.lf.alloc(nLf) || !gHost.logInt.alloc(CQ_INT_N) || !gHost.lenPrior.alloc(100001)) { err = "out of host memory (tables of the C library's lgammaf / logf)"; return CDM_ERR_INVALID; }
            cdm_contig_host_tables(gHost.lgam.data(), CQ_LG_TOP, gHost.lf.data(), CQ_LF_TOP, gHost.logInt.data(), CQ_INT_N, gHost.lenPrior.data());
            gHost.ready = true;
        }
        if (hipSetDevice(device) != hipSuccess) { err = "hipSetDevice failed"; return CDM_ERR_HIP; }
        float *a = nullptr, *b = nullptr; double *c = nullptr, *d = nullptr;
        // (plain hipMalloc: these live as long as the process, outside the arenas of the threads that come and go)
        if (hipMalloc(&a, nLg * 4) != hipSuccess || hipMalloc(&b, nLf * 4) != hipSuccess || hipMalloc(&c, (size_t) CQ_INT_N * 8) != hipSuccess || hipMalloc(&d, 100001 * 8) != hipSuccess) {
            err = "out of device memory (tables of the C library's lgammaf / logf: 1.3 GB)"; return CDM_ERR_HIP;
        }
        if (hipMemcpy(a, gHost.lgam.data(), nLg * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(b, gHost.lf.data(), nLf * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c, gHost.logInt.data(), (size_t) CQ_INT_N * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d, gHost.lenPrior.data(), 100001 * 8, hipMemcpyHostToDevice) != hipSuccess) {
            err = "copying the tables to the device failed"; return CDM_ERR_HIP;
        }
        gDev[device].lgam = a; gDev[device].lf = b; gDev[device].logInt = c; gDev[device].lenPrior = d;
        gDevReady[device] = true;
    }
    out = gDev[device];
    return CDM_OK;
}

}  // namespace
bool cdm_contig_tables_ready(int device) { std::lock_guard<std::mutex> lock(gTabMu); return device >= 0 && device < 64 && gDevReady[device]; }
namespace {
// ------------------------------------------------------------------------------------------------ per record / per query state
struct CqKey { uint32_t cons; float deam, lgBeta, lgAlphaBeta; };        // what the comparator reads of a record: alnLengthCons, deamMatch, its two cached terms
struct CqCo { int32_t qs, qe, ds, de; };                                  // the record's coordinates (oriented; rewritten when a parked hit is re-aligned)
struct CqOp { uint32_t tR, oR, lR, tL, oL, lL; uint32_t revR, revL; };    // a round's fragments: target, oriented start, letters, orientation (l = 0: none)
struct CqBuf { uint32_t *codes; uint16_t *nm; };                          // a round's buffer of grown sequences

enum : uint32_t { QF_EXTENDED = 1u, QF_FALLBACK = 2u, QF_HASN = 4u };

struct CqArgs {
    const SeqMeta *meta; const uint32_t *codes; const uint16_t *nm;       // the input DB
    const uint64_t *aoff; const AlnRec *rec; const ContigStat *st; const uint32_t *owner; uint64_t nRec; uint32_t n;
    CqTables tab;
    float mergeThr, ryThr; double likCT[2], likGA[2]; uint64_t maxSeqLen;
    CqKey *key; CqCo *co; uint8_t *gate;
    uint32_t *heap, *park;                  // [nRec]: a query's stretch is [aoff[q], aoff[q + 1])
    uint32_t *heapN, *parkN, *curLen, *verRound, *verWoff, *leftOff, *qflags;     // [n]
    const CqBuf *bufs;                      // [round]
    unsigned int *counters;                 // 0 undefined case, 1 queries handed back, 2 grown this round, 3 parked this round, 4 active next round, 5 first active
    uint32_t fallbackEvery;                 // tests: hand every k-th query back
};

__device__ __forceinline__ bool tabIndex(float x, int top, uint32_t &i) { i = __float_as_uint(x) - CQ_ONE; return i < ((uint32_t) top << 23); }

// ------------------------------------------------------------------------------------------------ the gate (:187-270), a thread per record
// deamMatches (nuclassembleUtil.cpp:1009-1044) behind its length prior: host/contigmerge.cpp deamFromPrior, operation by operation
__device__ __forceinline__ double deamFromPrior(unsigned overlap, unsigned score, double damageLik, double lengthPrior) {
    const double pMatch = 0.5f * ((((static_cast<double>(score) + 3.0f * overlap) / 5.0f) + 0.9f) / (overlap + 1)) + 0.5f * lengthPrior;
    const double pMismatch = 1 - pMatch;
    const double likelihoodRatio = pMismatch / damageLik;
    const double priorOdds = (1 - pMatch) / pMatch;
    return 1 / fma(likelihoodRatio, priorOdds, 1.0);
}
__global__ __launch_bounds__(256) void k_cq_gate(CqArgs a) {
    const uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.nRec) return;
    const AlnRec rec = a.rec[r]; const ContigStat st = a.st[r];
    const uint32_t q = a.owner[r];
    const SeqMeta qm = a.meta[q];
    const unsigned qLen = qm.len, dbLen = st.dbLen;
    const unsigned alnLength = (unsigned) max(abs(rec.qEnd - rec.qStart), abs(rec.dbEnd - rec.dbStart)) + 1u;         // Matcher::computeAlnLength
    float seqId = static_cast<float>(st.idCnt) / alnLength, rySeqId = static_cast<float>(st.idRy) / alnLength;
    bool pass = false;
    CqKey k; k.cons = 0; k.deam = 0; k.lgBeta = 0; k.lgAlphaBeta = 0;
    if (seqId >= a.mergeThr && rySeqId >= a.ryThr && qm.key != st.dbKey) {
        const bool rightStart = (unsigned) st.ds == 0 && (unsigned) st.qe == (qLen - 1);
        const bool leftStart = (unsigned) st.qs == 0 && (unsigned) st.de == (dbLen - 1);
        int tot = 0, idc = 0, idr = 0;
        if (leftStart || rightStart) {
            if (dbLen - alnLength > qLen) atomicExch(&a.counters[0], 1u);           // the reference pads with qLen - offset letters: undefined there
            else { tot = st.nnTot; idc = st.nnId; idr = st.nnRy; }
        }
        if (tot != 0) { seqId = static_cast<float>(idc) / tot; rySeqId = static_cast<float>(idr) / tot; }
        const unsigned cons = (unsigned) tot;
        unsigned minAlnLen = 500;
        minAlnLen = (alnLength < minAlnLen) ? min(minAlnLen, static_cast<unsigned>(0.2 * dbLen)) : minAlnLen;
        if (seqId >= a.mergeThr && rySeqId >= a.ryThr && alnLength >= minAlnLen) {
            float mCT = 0, mGA = 0;
            const unsigned mmCons = (1 - seqId) * cons + 0.5;
            const unsigned mCons = cons - mmCons;
            const unsigned scoreAln = mCons * 2 + mmCons * (-3);
            if (leftStart || rightStart) {
                const double likCT = a.likCT[st.rev ? 1 : 0], likGA = a.likGA[st.rev ? 1 : 0];
                const double prior = a.tab.lenPrior[min(alnLength, 100000u)];
                if (likCT > 0) { const double v = deamFromPrior(alnLength, scoreAln, likCT, prior); for (int i = 0; i < st.nCT; i++) mCT += v; }
                if (likGA > 0) { const double v = deamFromPrior(alnLength, scoreAln, likGA, prior); for (int i = 0; i < st.nGA; i++) mGA += v; }
            }
            k.cons = cons;
            k.deam = ((static_cast<float>(scoreAln) + 3.0f * cons) / 5.0f) + mCT + mGA;
            pass = true;
        }
    }
    CqCo c; c.qs = st.qs; c.qe = st.qe; c.ds = st.ds; c.de = st.de;
    a.key[r] = k; a.co[r] = c; a.gate[r] = pass ? 1 : 0;
}
// the queries with a gated record, in order
__global__ __launch_bounds__(256) void k_cq_any(CqArgs a, uint32_t *__restrict__ flag) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.n) return;
    uint32_t any = 0;
    for (uint64_t r = a.aoff[q]; r < a.aoff[q + 1] && !any; r++) any = a.gate[r];
    flag[q] = any;
    a.heapN[q] = 0; a.parkN[q] = 0; a.curLen[q] = a.meta[q].len; a.verRound[q] = CQ_NONE; a.verWoff[q] = 0; a.leftOff[q] = 0;
    a.qflags[q] = (a.meta[q].flags & 1u) ? QF_HASN : 0u;
}
__global__ __launch_bounds__(256) void k_cq_list(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos, uint32_t n, uint32_t *__restrict__ list) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n && flag[q]) list[pos[q]] = q;
}

// ------------------------------------------------------------------------------------------------ the comparator (:25-70) and libstdc++'s heap
struct Cmp {
    const CqArgs &a; const CqKey *key;          // the query's records
    bool handBack = false;
    __device__ Cmp(const CqArgs &a_, const CqKey *k) : a(a_), key(k) {}
    __device__ __forceinline__ float lgam(float x) { uint32_t i; if (!tabIndex(x, CQ_LG_TOP, i)) { handBack = true; return 0.f; } return a.tab.lgam[i]; }
    __device__ __forceinline__ float lf(float x) { uint32_t i; if (!tabIndex(x, CQ_LF_TOP, i)) { handBack = true; return 0.f; } return a.tab.lf[i]; }
    // "is record x a worse overlap than record y?"
    __device__ bool operator()(uint32_t xi, uint32_t yi) {
        const CqKey x = key[xi], y = key[yi];
        const float xMis = (x.cons - x.deam) + 1, yMis = (y.cons - y.deam) + 1;
        const float xHit = x.deam + 1, yHit = y.deam + 1;
        const double logScale = (lgam(xHit + yHit) + x.lgAlphaBeta) - (lgam(xMis + xHit + yHit) + x.lgBeta);
        double logTerm = 0.0, below = 0.0;
        for (size_t k = 0; k < yMis; k++) {
            below += exp(logTerm + logScale);
            if (k + 1 >= CQ_INT_N) { handBack = true; break; }
            logTerm = lf(xMis + k) + lf(yHit + k) - (a.tab.logInt[k + 1] + lf(k + xMis + xHit + yHit)) + logTerm;
            if (handBack) break;
        }
        if (!(fabs(below - 0.45) > CQ_BAND && fabs(below - 0.55) > CQ_BAND) || !(below < 1e300)) handBack = true;     // (a NaN or an overflow too)
        if (below < 0.45) return true;
        if (below > 0.55) return false;
        return !(x.cons > y.cons);
    }
};
// std::push_heap's __push_heap: the value at hole `hole` moves up while its parent compares less
__device__ __forceinline__ void heapSiftUp(uint32_t *h, uint32_t hole, uint32_t top, uint32_t value, Cmp &cmp) {
    uint32_t parent = (hole - 1) / 2;
    while (hole > top && cmp(h[parent], value)) { h[hole] = h[parent]; hole = parent; parent = (hole - 1) / 2; }
    h[hole] = value;
}
__device__ __forceinline__ void heapPush(uint32_t *h, uint32_t &n, uint32_t value, Cmp &cmp) { h[n] = value; n++; heapSiftUp(h, n - 1, 0, value, cmp); }
// priority_queue::pop: std::pop_heap (__pop_heap -> __adjust_heap), then pop_back
__device__ __forceinline__ void heapPop(uint32_t *h, uint32_t &n, Cmp &cmp) {
    if (n > 1) {
        const uint32_t value = h[n - 1];
        h[n - 1] = h[0];
        const uint32_t len = n - 1;
        uint32_t hole = 0, child = 0;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (cmp(h[child], h[child - 1])) child--;
            h[hole] = h[child]; hole = child;
        }
        if ((len & 1u) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); h[hole] = h[child - 1]; hole = child - 1; }
        heapSiftUp(h, hole, 0, value, cmp);
    }
    n--;
}

// ------------------------------------------------------------------------------------------------ a round of a query (:276-402)
// useReverse[target] (:136-137,198,212): the orientation of the query's LAST record with that target
__device__ __forceinline__ bool useReverse(const CqArgs &a, uint64_t r0, uint64_t r1, uint32_t target) {
    for (uint64_t r = r1; r-- > r0;) if (a.rec[r].target == target) return a.st[r].rev != 0;
    return false;
}
// -> the round's fragments (op), the number of parked hits (the query goes on if there are any), whether it grew, whether the host takes it
__device__ __forceinline__ void cqRound(const CqArgs &a, uint32_t q, uint32_t round, CqOp &op, uint32_t &parked, bool &grew, bool &back) {
    const uint64_t r0 = a.aoff[q], r1 = a.aoff[q + 1];
    uint32_t *h = a.heap + r0, *pk = a.park + r0;
    const CqKey *key = a.key + r0;
    Cmp cmp(a, key);
    uint32_t hn = a.heapN[q];
    const SeqMeta qm = a.meta[q];
    uint32_t flags = a.qflags[q];
    bool giveUp = (qm.flags & 4u) != 0 || (a.fallbackEvery && q % a.fallbackEvery == 0);
    // ---- pushes: round 0 the gated records in the order of the records, later the parked hits that still pass, in the order they were parked
    if (!giveUp) {
        if (round == 0) {
            for (uint64_t r = r0; r < r1 && !cmp.handBack; r++) if (a.gate[r]) {
                CqKey k = a.key[r];
                const float mm = k.cons - k.deam, alpha = mm + 1, beta = k.deam + 1;       // Res::cacheTerms
                k.lgBeta = cmp.lgam(beta); k.lgAlphaBeta = cmp.lgam(alpha + beta);
                a.key[r] = k;
                heapPush(h, hn, (uint32_t) (r - r0), cmp);
            }
        } else {
            const uint32_t np = a.parkN[q];
            for (uint32_t j = 0; j < np && !cmp.handBack; j++) if (a.gate[r0 + pk[j]]) heapPush(h, hn, pk[j], cmp);
        }
    }
    uint32_t np = 0, leftOff = 0, rightOff = 0;
    const unsigned qLen = a.curLen[q];
    op.lR = op.lL = 0; op.tR = op.tL = op.oR = op.oL = op.revR = op.revL = 0;
    if (!giveUp && !cmp.handBack && hn > 0) {
        while (true) {
            // selectNuclFragmentToExtendContigs (:73-91)
            bool found = false; uint32_t best = 0; CqCo c; uint32_t tLen = 0, target = 0;
            while (hn > 0 && !cmp.handBack) {
                const uint32_t ri = h[0];
                heapPop(h, hn, cmp);
                c = a.co[r0 + ri];
                const ContigStat &st = a.st[r0 + ri];
                tLen = st.dbLen;
                const bool notBoth = !(c.ds == 0 && c.qs == 0);
                const bool rightStart = c.ds == 0 && (c.de != static_cast<int>(tLen) - 1);
                const bool leftStart = c.qs == 0 && (c.qe != static_cast<int>(qLen) - 1);
                if ((rightStart || leftStart) && notBoth && st.dbKey != qm.key) { found = true; best = ri; target = a.rec[r0 + ri].target; break; }
            }
            if (!found || cmp.handBack) break;
            if (c.ds == 0) { if ((tLen - (unsigned) (c.de + 1)) <= rightOff) continue; }
            else if (c.qs == 0) { if (c.ds <= static_cast<int>(leftOff)) continue; }
            if (a.meta[target].flags & 4u) { giveUp = true; break; }                       // letters beyond ACGTN in play: the host's strings
            const unsigned ds = c.ds, de = c.de, qs = c.qs, qe = c.qe;
            const uint64_t size = (uint64_t) qLen + leftOff + rightOff;
            if (ds == 0 && qe == (qLen - 1)) {
                if (rightOff > 0) { pk[np++] = best; continue; }
                const unsigned fragLen = tLen - (de + 1);
                if (size + fragLen >= a.maxSeqLen) break;
                const bool rev = useReverse(a, r0, r1, target);
                op.tR = target; op.lR = fragLen; op.revR = rev; op.oR = de + 1;             // (oriented: the tail of the target as aligned)
                rightOff += fragLen;
            } else if (qs == 0 && de == (tLen - 1)) {
                if (leftOff > 0) { pk[np++] = best; continue; }
                const unsigned fragLen = ds;
                if (size + fragLen >= a.maxSeqLen) break;
                const bool rev = useReverse(a, r0, r1, target);
                op.tL = target; op.lL = fragLen; op.revL = rev; op.oL = 0;
                leftOff += fragLen;
            }
        }
    }
    if (giveUp || cmp.handBack) { a.qflags[q] = flags | QF_FALLBACK; a.heapN[q] = 0; a.parkN[q] = 0; back = true; return; }
    if (leftOff > 0 || rightOff > 0) { flags |= QF_EXTENDED; grew = true; a.curLen[q] = qLen + leftOff + rightOff; }
    a.leftOff[q] = leftOff;
    a.qflags[q] = flags;
    a.heapN[q] = 0;
    if (hn != 0) { a.parkN[q] = 0; return; }          // :403 the loop ends on a queue that is not empty (--max-seq-len)
    a.parkN[q] = np;
    parked = np;
}
// a wave of queries; what they append to the round's lists goes out with one atomic per wave and list (a single word takes ~88 atomics
// per microsecond: 25 M queries' worth of them was most of a round)
__global__ __launch_bounds__(64) void k_cq_round(CqArgs a, const uint32_t *__restrict__ active, uint32_t nActive, uint32_t round, uint32_t *__restrict__ next, uint32_t *__restrict__ grown,
                                                 CqOp *__restrict__ ops, uint64_t *__restrict__ parkWork) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t q = 0, parked = 0; bool grew = false, back = false;
    CqOp op; op.lR = op.lL = 0; op.tR = op.tL = op.oR = op.oL = op.revR = op.revL = 0;
    if (slot < nActive) { q = active[slot]; cqRound(a, q, round, op, parked, grew, back); }
    const uint64_t mb = __ballot(back);
    if (mb && lane == __ffsll((unsigned long long) mb) - 1) atomicAdd(&a.counters[1], (unsigned int) __popcll(mb));
    const uint32_t g = cdm_wave_append(a.counters + 2, grew);
    if (grew) { grown[g] = q; ops[g] = op; }
    uint32_t incl = parked;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t) __shfl_up((int) incl, o, 64); if (lane >= o) incl += t; }
    const uint32_t total = (uint32_t) __shfl((int) incl, 63, 64);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(&a.counters[3], total);
    base = (uint32_t) __shfl((int) base, 63, 64);
    const uint32_t w = base + incl - parked;
    for (uint32_t j = 0; j < parked; j++) parkWork[w + j] = ((uint64_t) q << 32) | j;
    const uint32_t nx = cdm_wave_append(a.counters + 4, parked > 0);
    if (parked) next[nx] = q;
}

// ------------------------------------------------------------------------------------------------ letters
// 16 N bits of the (optionally reversed) sequence starting at oriented position i
__device__ __forceinline__ uint32_t nWindow16(const uint16_t *__restrict__ nm, uint32_t w0, uint32_t lastWord, uint32_t pos) {
    const uint32_t w = pos >> 4, sh = pos & 15u;
    const uint32_t lo = nm[w0 + w], hi = (w + 1 <= lastWord) ? nm[w0 + w + 1] : 0u;
    return ((lo | (hi << 16)) >> sh) & 0xFFFFu;
}
__device__ __forceinline__ uint32_t orientedN16(const uint16_t *__restrict__ nm, uint32_t w0, uint32_t L, uint32_t lastWord, bool rc, uint32_t i) {
    if (!rc) return nWindow16(nm, w0, lastWord, i);
    const int s = (int) L - 16 - (int) i;
    const uint32_t w = (s >= 0) ? nWindow16(nm, w0, lastWord, (uint32_t) s) : ((nWindow16(nm, w0, lastWord, 0) << (-s)) & 0xFFFFu);
    return __brev(w) >> 16;
}
// a sequence as the kernels below read it: the input DB's or a round buffer's
struct Src { const uint32_t *codes; const uint16_t *nm; uint32_t w0, len, lastWord; bool hasN; };
__device__ __forceinline__ Src srcOfDb(const CqArgs &a, uint32_t t) { const SeqMeta m = a.meta[t]; Src s; s.codes = a.codes; s.nm = a.nm; s.w0 = m.woff; s.len = m.len; s.lastWord = (m.len + 15) / 16 - 1; s.hasN = (m.flags & 1u) != 0; return s; }
// 16 letters from oriented position i on: codes (an N is code 0) and N bits; positions beyond the sequence are garbage
__device__ __forceinline__ void fetch16(const Src &s, bool rc, uint32_t i, uint32_t &code, uint32_t &nb) {
    code = cdm_oriented_window16(s.codes, s.w0, s.len, s.lastWord, rc, i);
    nb = 0;
    if (s.hasN) { nb = orientedN16(s.nm, s.w0, s.len, s.lastWord, rc, i); const uint32_t m = cdm_spread16(nb); code &= ~(m | (m << 1)); }
}

// ------------------------------------------------------------------------------------------------ the grown query of a round, a wave per query
__global__ __launch_bounds__(256) void k_cq_grow(CqArgs a, const uint32_t *__restrict__ grown, const CqOp *__restrict__ ops, const uint32_t *__restrict__ newWoff, uint32_t nGrown, uint32_t first,
                                                 uint32_t round, const uint32_t *__restrict__ prevRound, const uint32_t *__restrict__ prevWoff, const uint32_t *__restrict__ prevLen) {
    const uint32_t g = first + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= nGrown) return;
    const uint32_t q = grown[g];
    const CqOp op = ops[g];
    Src old;
    if (prevRound[g] == CQ_NONE) old = srcOfDb(a, q);
    else { const CqBuf b = a.bufs[prevRound[g]]; old.codes = b.codes; old.nm = b.nm; old.w0 = prevWoff[g]; old.len = prevLen[g]; old.lastWord = (old.len + 15) / 16 - 1; old.hasN = (a.qflags[q] & QF_HASN) != 0; }
    const Src sL = op.lL ? srcOfDb(a, op.tL) : old, sR = op.lR ? srcOfDb(a, op.tR) : old;
    const uint32_t newLen = op.lL + old.len + op.lR, nw = (newLen + 15) / 16, ob = newWoff[g];
    const CqBuf out = a.bufs[round];
    uint32_t anyN = 0;
    for (uint32_t w = lane; w < nw; w += 64) {
        const uint32_t p0 = w * 16, p1 = min(newLen, p0 + 16);
        uint32_t code = 0, nb = 0;
        // the three pieces: [0, lL) the left fragment, [lL, lL + old) the query as it was, then the right fragment
        for (int piece = 0; piece < 3; piece++) {
            const uint32_t pa = piece == 0 ? 0u : piece == 1 ? op.lL : op.lL + old.len;
            const uint32_t pb = piece == 0 ? op.lL : piece == 1 ? op.lL + old.len : newLen;
            const uint32_t lo = max(p0, pa), hi = min(p1, pb);
            if (lo >= hi) continue;
            const Src &s = piece == 0 ? sL : piece == 1 ? old : sR;
            const bool rc = piece == 0 ? op.revL != 0 : piece == 1 ? false : op.revR != 0;
            const uint32_t o0 = piece == 0 ? op.oL : piece == 1 ? 0u : op.oR;
            uint32_t c, nbits;
            fetch16(s, rc, o0 + (lo - pa), c, nbits);
            const uint32_t cnt = hi - lo, sh = lo - p0;
            const uint32_t cm = cnt < 16 ? ((1u << (2 * cnt)) - 1u) : 0xFFFFFFFFu, nmk = cnt < 16 ? ((1u << cnt) - 1u) : 0xFFFFu;
            code |= (c & cm) << (2 * sh); nb |= (nbits & nmk) << sh;
        }
        out.codes[ob + w] = code; out.nm[ob + w] = (uint16_t) nb; anyN |= nb;
    }
    const uint64_t m = __ballot(anyN != 0);
    if (lane == 0) {
        uint32_t f = a.qflags[q] & ~QF_HASN;
        if (m) f |= QF_HASN;
        a.qflags[q] = f; a.verRound[q] = round; a.verWoff[q] = ob;
    }
}
__global__ __launch_bounds__(256) void k_cq_grow_sizes(CqArgs a, const uint32_t *__restrict__ grown, uint32_t nGrown, uint32_t *__restrict__ words, uint32_t *__restrict__ prevRound,
                                                       uint32_t *__restrict__ prevWoff, uint32_t *__restrict__ prevLen, const CqOp *__restrict__ ops) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > nGrown) return;
    if (g == nGrown) { words[g] = 0; return; }
    const uint32_t q = grown[g];
    const uint32_t newLen = a.curLen[q];
    words[g] = (newLen + 15) / 16;
    prevRound[g] = a.verRound[q]; prevWoff[g] = a.verWoff[q]; prevLen[g] = newLen - ops[g].lL - ops[g].lR;
}

// ------------------------------------------------------------------------------------------------ the parked hits on the grown query (:404-455), a wave each
// (nWork: the end of this launch's slice - the launch is rounded up to whole blocks, and an item done twice would be re-aligned from its own result)
__global__ __launch_bounds__(256) void k_cq_parked(CqArgs a, const uint64_t *__restrict__ work, uint32_t nWork, uint32_t first) {
    const uint32_t wi = first + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (wi >= nWork) return;
    const uint32_t q = (uint32_t) (work[wi] >> 32), j = (uint32_t) work[wi];
    const uint64_t r0 = a.aoff[q], r1 = a.aoff[q + 1];
    const uint64_t r = r0 + a.park[r0 + j];
    const uint32_t target = a.rec[r].target;
    // useReverse: the last record of the query with this target
    long long last = -1;
    for (uint64_t x = r0 + lane; x < r1; x += 64) if (a.rec[x].target == target) last = (long long) x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const long long other = __shfl_xor(last, o, 64); last = max(last, other); }
    const bool rev = a.st[last].rev != 0;
    const Src t = srcOfDb(a, target);
    Src qy;
    { const uint32_t vr = a.verRound[q]; const CqBuf b = a.bufs[vr]; qy.codes = b.codes; qy.nm = b.nm; qy.w0 = a.verWoff[q]; qy.len = a.curLen[q]; qy.lastWord = (qy.len + 15) / 16 - 1; qy.hasN = (a.qflags[q] & QF_HASN) != 0; }
    const unsigned qLen = qy.len, tLen = t.len;
    const CqCo c = a.co[r];
    const int diag = (c.qs + (int) a.leftOff[q]) - c.ds;
    const unsigned md = (unsigned) abs(diag);
    int startPos = -1, endPos = -1; unsigned diagonalLen = 0;
    bool have = false; unsigned m = 0, qFrom = 0, tFrom = 0;
    if (diag >= 0 && md < qLen) { m = min(tLen, qLen - md); qFrom = md; have = true; }
    else if (diag < 0 && md < tLen) { m = min(tLen - md, qLen); tFrom = md; have = true; }
    if (have) { diagonalLen = m; startPos = 0; endPos = (int) m - 1; }        // (no '*' without a row of original letters: those queries went back to the host)
    const int dist = (int) md;
    int qs2, qe2, ds2, de2;
    if (diag >= 0) { qs2 = startPos + dist; qe2 = endPos + dist; ds2 = startPos; de2 = endPos; }
    else { qs2 = startPos; qe2 = endPos; ds2 = startPos + dist; de2 = endPos + dist; }
    int idCnt = 0, idRy = 0;
    if (have && endPos >= startPos) {
        // identical letters over [startPos, endPos), same RY class over [startPos, endPos]
        const uint32_t colsId = (uint32_t) (endPos - startPos), colsRy = colsId + 1;
        for (uint32_t c0 = lane * 16; c0 < colsRy; c0 += 64 * 16) {
            uint32_t qc, qn, tc, tn;
            fetch16(qy, false, qFrom + c0, qc, qn);
            fetch16(t, rev, tFrom + c0, tc, tn);
            const uint32_t x = qc ^ tc, nx = cdm_spread16(qn ^ tn);
            uint32_t differ = ((x | (x >> 1)) & 0x55555555u) | nx, ryDiffer = x & 0x55555555u;         // an N is code 0: RY class 0, as ryMap has it
            const uint32_t nId = c0 < colsId ? min(16u, colsId - c0) : 0u, nRy = min(16u, colsRy - c0);
            if (nId < 16) differ &= (1u << (2 * nId)) - 1u;
            if (nRy < 16) ryDiffer &= (1u << (2 * nRy)) - 1u;
            idCnt += (int) nId - __popc(differ); idRy += (int) nRy - __popc(ryDiffer);
        }
        idCnt = cdm_wave_sum(idCnt); idRy = cdm_wave_sum(idRy);
    }
    if (lane == 0) {
        const float seqId = static_cast<float>(idCnt) / (static_cast<float>(qe2) - static_cast<float>(qs2));
        const float rySeqId = static_cast<float>(idRy) / diagonalLen;
        CqCo o; o.qs = qs2; o.qe = qe2; o.ds = ds2; o.de = de2;
        a.co[r] = o;
        a.gate[r] = (seqId >= a.mergeThr && rySeqId >= a.ryThr) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------ the result
__global__ __launch_bounds__(256) void k_cq_out_flags(CqArgs a, const uint8_t *__restrict__ ext, uint32_t *__restrict__ isGrown, uint32_t *__restrict__ words, uint8_t *__restrict__ outExt, uint8_t *__restrict__ handedBack) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > a.n) return;
    if (q == a.n) { isGrown[q] = 0; words[q] = 0; return; }
    const uint32_t f = a.qflags[q];
    const bool back = (f & QF_FALLBACK) != 0, g = !back && (f & QF_EXTENDED);
    isGrown[q] = g ? 1u : 0u;
    words[q] = g ? (a.curLen[q] + 15) / 16 : 0u;
    outExt[q] = g ? 1 : ext[q];
    handedBack[q] = back ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_cq_gather(CqArgs a, const uint32_t *__restrict__ isGrown, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ wpos, uint32_t first,
                                                   uint32_t *__restrict__ idx, uint32_t *__restrict__ len, uint32_t *__restrict__ key, uint32_t *__restrict__ codes, uint16_t *__restrict__ nm) {
    const uint32_t q = first + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= a.n || !isGrown[q]) return;
    const uint32_t L = a.curLen[q], nw = (L + 15) / 16, ob = wpos[q];
    const CqBuf b = a.bufs[a.verRound[q]];
    const uint32_t ib = a.verWoff[q];
    for (uint32_t w = lane; w < nw; w += 64) { codes[ob + w] = b.codes[ib + w]; nm[ob + w] = b.nm[ib + w]; }
    if (lane == 0) { idx[pos[q]] = q; len[pos[q]] = L; key[pos[q]] = a.meta[q].key; }
}
}  // namespace

int cdm_contig_queue_device(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, float mergeSeqIdThr, const SeqMeta *meta, const uint32_t *owner,
                            const ContigStat *dStats, CqResult *res) {
    hipStream_t s = ctx->stream;
    const uint32_t n = (uint32_t) db->n;
    const uint64_t nRec = alns->count;
    const bool timing = cdmGetenv("CDM_TIMING") != nullptr;
    auto tPrev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (timing) { hipStreamSynchronize(s); const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  contig merge: %-28s %.3f s\n", what, std::chrono::duration<double>(t - tPrev).count()); tPrev = t; } };
    CqArgs a;
    { std::string err; if (int rc = ensureTables(ctx->device, a.tab, err)) { cdm_set_error("cdm_contig_merge: %s", err.c_str()); return rc; } }
    CDM_HIP(hipSetDevice(ctx->device));
    lap("tables (first call: filled)");
    DevBuf<CqKey> key; DevBuf<CqCo> co; DevBuf<uint8_t> gate, outExt, handedBack; DevBuf<uint32_t> heap, park, heapN, parkN, curLen, verRound, verWoff, leftOff, qflags, flag, pos, listA, listB, grown;
    DevBuf<unsigned int> counters; DevBuf<CqBuf> bufs;
    if (!key.alloc(nRec) || !co.alloc(nRec) || !gate.alloc(nRec) || !heap.alloc(nRec) || !park.alloc(nRec) || !heapN.alloc(n) || !parkN.alloc(n) || !curLen.alloc(n) || !verRound.alloc(n) ||
        !verWoff.alloc(n) || !leftOff.alloc(n) || !qflags.alloc(n) || !flag.alloc((size_t) n + 1) || !pos.alloc((size_t) n + 1) || !counters.alloc(8) || !bufs.alloc(CQ_MAX_ROUNDS) || !outExt.alloc(n) || !handedBack.alloc(n)) {
        cdm_set_error("cdm_contig_merge: out of device memory"); return CDM_ERR_HIP;
    }
    a.meta = meta; a.codes = db->codes; a.nm = reinterpret_cast<const uint16_t *>(db->nmask);
    a.aoff = alns->off; a.rec = alns->rec; a.st = dStats; a.owner = owner; a.nRec = nRec; a.n = n;
    a.mergeThr = mergeSeqIdThr; a.ryThr = par->ry_seq_id_thr; a.maxSeqLen = par->max_seq_len;
    for (int rv = 0; rv < 2; rv++) { a.likCT[rv] = (double) ctx->mats[rv][5][1][3]; a.likGA[rv] = (double) ctx->mats[rv][5][2][0]; }
    a.key = key.p; a.co = co.p; a.gate = gate.p; a.heap = heap.p; a.park = park.p; a.heapN = heapN.p; a.parkN = parkN.p; a.curLen = curLen.p; a.verRound = verRound.p; a.verWoff = verWoff.p;
    a.leftOff = leftOff.p; a.qflags = qflags.p; a.bufs = bufs.p; a.counters = counters.p;
    a.fallbackEvery = cdmGetenv("CDM_CONTIG_HAND_BACK_EVERY") ? (uint32_t) atoi(cdmGetenv("CDM_CONTIG_HAND_BACK_EVERY")) : 0u;
    CDM_HIP(hipMemsetAsync(counters.p, 0, 32, s));
    if (nRec) hipLaunchKernelGGL(k_cq_gate, CDM_GRID((nRec + 255) / 256, 256), dim3(256), 0, s, a);
    if (n) hipLaunchKernelGGL(k_cq_any, dim3((n + 255) / 256), dim3(256), 0, s, a, flag.p);
    CDM_HIP(hipMemsetAsync(flag.p + n, 0, 4, s));
    cdmscan::ScanTemp st;
    if (cdmscan::exclusiveScan<uint32_t>(s, st, flag.p, pos.p, (size_t) n + 1) != CDM_OK) { cdm_set_error("cdm_contig_merge: scan failed"); return CDM_ERR_HIP; }
    uint32_t nActive = 0; unsigned int hc[8];
    CDM_HIP(hipMemcpyAsync(&nActive, pos.p + n, 4, hipMemcpyDeviceToHost, s));
    CDM_HIP(hipMemcpyAsync(hc, counters.p, 32, hipMemcpyDeviceToHost, s));
    CDM_HIP(hipStreamSynchronize(s));
    if (hc[0]) { cdm_set_error("cdm_contig_merge: a target overhangs its query by more than the query's length; the reference pads it with a negative number of letters there (undefined behaviour), not reproduced"); return CDM_ERR_UNSUPPORTED; }
    lap("gate");
    DevBuf<CqOp> ops; DevBuf<uint64_t> parkWork; DevBuf<uint32_t> gWords, gWoff, prevRound, prevWoff, prevLen;
    if (!listA.alloc(nActive) || !listB.alloc(nActive) || !grown.alloc(nActive) || !ops.alloc(nActive) || !parkWork.alloc(nRec) || !gWords.alloc((size_t) nActive + 1) || !gWoff.alloc((size_t) nActive + 1) ||
        !prevRound.alloc(nActive) || !prevWoff.alloc(nActive) || !prevLen.alloc(nActive)) { cdm_set_error("cdm_contig_merge: out of device memory"); return CDM_ERR_HIP; }
    if (n) hipLaunchKernelGGL(k_cq_list, dim3((n + 255) / 256), dim3(256), 0, s, flag.p, pos.p, n, listA.p);
    std::vector<void *> roundMem;       // the rounds' buffers, released at the end
    auto freeRounds = [&] { for (void *p : roundMem) cdmFree(p); roundMem.clear(); };
    uint32_t *act = listA.p, *nxt = listB.p;
    uint32_t round = 0; uint64_t grownTotal = 0, parkedTotal = 0;
    while (nActive > 0) {
        if (round >= (uint32_t) CQ_MAX_ROUNDS) { freeRounds(); cdm_set_error("cdm_contig_merge: more than %d rounds of the extension loop", CQ_MAX_ROUNDS); return CDM_ERR_UNSUPPORTED; }
        hipMemsetAsync(counters.p + 2, 0, 12, s);
        hipLaunchKernelGGL(k_cq_round, dim3((nActive + 63) / 64), dim3(64), 0, s, a, act, nActive, round, nxt, grown.p, ops.p, parkWork.p);
        hipMemcpyAsync(hc, counters.p, 32, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { freeRounds(); cdm_set_error("cdm_contig_merge: a round of the queue failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
        const uint32_t nGrown = hc[2], nPark = hc[3], nNext = hc[4];
        if (nGrown) {
            hipLaunchKernelGGL(k_cq_grow_sizes, dim3((nGrown + 256) / 256), dim3(256), 0, s, a, grown.p, nGrown, gWords.p, prevRound.p, prevWoff.p, prevLen.p, ops.p);
            if (cdmscan::exclusiveScan<uint32_t>(s, st, gWords.p, gWoff.p, (size_t) nGrown + 1) != CDM_OK) { freeRounds(); cdm_set_error("cdm_contig_merge: scan failed"); return CDM_ERR_HIP; }
            uint32_t words = 0;
            hipMemcpyAsync(&words, gWoff.p + nGrown, 4, hipMemcpyDeviceToHost, s);
            // (the words of a round are summed in 32 bits: beyond 68 G bases in one round's grown contigs the sum wraps - checked against the lengths' sum below)
            if (hipStreamSynchronize(s) != hipSuccess) { freeRounds(); return CDM_ERR_HIP; }
            CqBuf b; b.codes = nullptr; b.nm = nullptr;
            if (cdmMalloc(&b.codes, ((size_t) words + 2) * 4) != hipSuccess) { freeRounds(); cdm_set_error("cdm_contig_merge: out of device memory (round %u: %u code words)", round, words); return CDM_ERR_HIP; }
            roundMem.push_back(b.codes);
            if (cdmMalloc(&b.nm, ((size_t) words + 2) * 2) != hipSuccess) { freeRounds(); cdm_set_error("cdm_contig_merge: out of device memory"); return CDM_ERR_HIP; }
            roundMem.push_back(b.nm);
            hipMemcpyAsync(bufs.p + round, &b, sizeof(CqBuf), hipMemcpyHostToDevice, s);
            hipStreamSynchronize(s);        // (b is a local)
            for (uint64_t first = 0, slice = cdmSliceItems(64); first < nGrown; first += slice)
                hipLaunchKernelGGL(k_cq_grow, CDM_GRID((std::min<uint64_t>(slice, nGrown - first) * 64 + 255) / 256, 256), dim3(256), 0, s, a, grown.p, ops.p, gWoff.p, (uint32_t) std::min<uint64_t>(nGrown, first + slice), (uint32_t) first, round,
                                   prevRound.p, prevWoff.p, prevLen.p);
            grownTotal += nGrown;
        }
        if (nPark) {
            for (uint64_t first = 0, slice = cdmSliceItems(64); first < nPark; first += slice)
                hipLaunchKernelGGL(k_cq_parked, CDM_GRID((std::min<uint64_t>(slice, nPark - first) * 64 + 255) / 256, 256), dim3(256), 0, s, a, parkWork.p, (uint32_t) std::min<uint64_t>(nPark, first + slice), (uint32_t) first);
            parkedTotal += nPark;
        }
        std::swap(act, nxt);
        nActive = nNext;
        round++;
    }
    if (timing) fprintf(stderr, "  contig merge: %u rounds, %llu growths, %llu parked hits re-aligned, %u queries handed back to the host\n", round, (unsigned long long) grownTotal, (unsigned long long) parkedTotal, hc[1]);
    lap("queues + extension (device)");
    // (what the rounds worked on goes back before the results are allocated: a result placed behind these blocks would sit in the middle
    // of the arena's free space once they are released, and the next kmermatcher's 30 GB buffers would have to be mapped anew)
    key.free(); co.free(); gate.free(); heap.free(); park.free(); parkWork.free(); ops.free(); listA.free(); listB.free(); grown.free(); gWords.free(); gWoff.free();
    prevRound.free(); prevWoff.free(); prevLen.free(); heapN.free(); parkN.free(); leftOff.free();
    // ---- the grown contigs as a DB of their own, in the order of their queries
    DevBuf<uint32_t> words, wpos;
    if (!words.alloc((size_t) n + 1) || !wpos.alloc((size_t) n + 1)) { freeRounds(); cdm_set_error("cdm_contig_merge: out of device memory"); return CDM_ERR_HIP; }
    hipLaunchKernelGGL(k_cq_out_flags, dim3((n + 256) / 256), dim3(256), 0, s, a, db->ext, flag.p, words.p, outExt.p, handedBack.p);
    if (cdmscan::exclusiveScan<uint32_t>(s, st, flag.p, pos.p, (size_t) n + 1) != CDM_OK || cdmscan::exclusiveScan<uint32_t>(s, st, words.p, wpos.p, (size_t) n + 1) != CDM_OK) { freeRounds(); cdm_set_error("cdm_contig_merge: scan failed"); return CDM_ERR_HIP; }
    uint32_t m = 0, w = 0;
    hipMemcpyAsync(&m, pos.p + n, 4, hipMemcpyDeviceToHost, s); hipMemcpyAsync(&w, wpos.p + n, 4, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(hc, counters.p, 32, hipMemcpyDeviceToHost, s);
    res->outExt.resize(n); res->handedBack.clear();
    if (n) hipMemcpyAsync(res->outExt.data(), outExt.p, n, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) { freeRounds(); cdm_set_error("cdm_contig_merge: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    if (hc[1]) { res->handedBack.resize(n); CDM_HIP(hipMemcpy(res->handedBack.data(), handedBack.p, n, hipMemcpyDeviceToHost)); }
    res->nHandedBack = hc[1];
    res->grownIdx.resize(m); res->grown = nullptr;
    if (m) {
        DevBuf<uint32_t> idx, len, gkey, codes; DevBuf<uint16_t> nm;
        if (!idx.alloc(m) || !len.alloc(m) || !gkey.alloc(m) || !codes.alloc(w) || !nm.alloc(w)) { freeRounds(); cdm_set_error("cdm_contig_merge: out of device memory"); return CDM_ERR_HIP; }
        for (uint64_t first = 0, slice = cdmSliceItems(64); first < n; first += slice)
            hipLaunchKernelGGL(k_cq_gather, CDM_GRID((std::min<uint64_t>(slice, n - first) * 64 + 255) / 256, 256), dim3(256), 0, s, a, flag.p, pos.p, wpos.p, (uint32_t) first, idx.p, len.p, gkey.p, codes.p, nm.p);
        hipMemcpyAsync(res->grownIdx.data(), idx.p, (size_t) m * 4, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { freeRounds(); cdm_set_error("cdm_contig_merge: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
        freeRounds();
        if (int rc = cdm_seqdb_from_packed(ctx, codes.p, nm.p, len.p, gkey.p, m, w, 1, &res->grown)) return rc;
    }
    freeRounds();
    lap("grown contigs gathered");
    return CDM_OK;
}
