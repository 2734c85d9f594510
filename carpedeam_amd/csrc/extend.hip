// ancient_read_assemble on the device (safe mode, --unsafe 0).
//
// Replaces doNuclAssembly1 (src/assembler/ancientReadsResults.cpp:178-581).  One thread owns one query that has more than
// one alignment record and runs the reference's per-query algorithm literally:
//   A-C  end-overlap filter, identity recomputation on the corrected sequences, candidate gate          (:202-315)
//   cons safe-mode consensus = N^L query N^L (nuclassembleUtil.cpp:586-592) -> the overlap columns are the alignment itself
//   upd  updateSeqIdConsensusReads (nuclassembleUtil.cpp:377-500): ids over the non-N columns, longest overlap per side
//   D    r_s_pair / calcLikelihoodConsensus (:48-70, nuclassembleUtil.cpp:203-374): per-column log-likelihoods from the
//        host-built table, summed in software x87 extended precision in column order; excess penalty and random-alignment
//        terms in float exactly as the reference's overload resolution computes them
//   E    greedy extension with std::priority_queue order (libstdc++ binary heap: the tie order of equal scores matters),
//        parked hits re-aligned on the grown query by diagonal (DistanceCalculator.h:115-175, mode 3), updateNuclAlignment
//        (nuclassembleUtil.cpp:9-47), re-scored and re-queued                                              (:374-546)
// The grown query is never materialised while the algorithm runs: it is a list of pieces (target, start, length) around
// the original query, and a second kernel writes the packed output DB once all final lengths are known.
//
// Reverse-strand records can never pass the reference's end-overlap test, which is evaluated on the raw coordinates
// where q_start > q_end encodes the strand (:204-213: rightStart needs q_end == qLen-1 < q_start, leftStart needs
// q_start == 0 > q_end), so useReverse[] stays false and no fragment is ever reverse-complemented on this path.
#include "scan.h"

#include <cmath>
#include <cstring>

#include "common.h"
#include "devutil.h"

namespace {

struct Cand {
    int qs, qe, ds, de;
    uint32_t target, alnLen, dbLen, qLen;
    float seqId, rySeqId;
    double sLenNorm;
    uint32_t pieceStart, pieceLen;   // set when the candidate donated a fragment
    uint32_t tKey;                   // the target's DB key (the extension loop compares it with the query's at every pop)
};

struct ExtArgs {
    MetaWoff woff; MetaLen len; MetaHasN hasN; MetaExt ext; MetaKey key; MetaRaw hasRaw;      // per-sequence metadata, one record per sequence
    const uint32_t *codes, *nmask; const uint8_t *raw;
    const uint64_t *aoff;
    const AlnRec *rec;
    const uint32_t *active; const unsigned int *nActive;
    const DamageLut *lut;
    Cand *cand;            // [alignment count]
    uint32_t *lists;       // [4 * alignment count]: heap, parked, left pieces, right pieces (per-query slices)
    uint32_t *newLen;      // [n] final length (0 = not extended)
    uint32_t *nLeft, *nRight, *leftTotal;   // [n]
    double *scores;        // optional [alignment count]
    float seqIdThr, rySeqIdThr, likelihoodThr;
    float excessLog, randLog;   // logf(excessPenal), logf(randAlnPenal) from the host
    double ratioLogit;          // log(1/thr - 1): sRatio > thr  <=>  randAln - likMod < ratioLogit
    // The reference decides 1 / (1 + expl(x)) > thr (nuclassembleUtil.cpp:331-336, ancientReadsResults.cpp:62-66), the device x < ratioLogit.
    // At the default thr = 0.5 the two agree for every x (ratioLogit = 0, expl(x) < 1 <=> x < 0): ratioWindow = 0.  For another
    // --likelihood-ratio-threshold they can differ only where x lies within a few ulps of ratioLogit - there the last bit of the C library's
    // expl and of the division decide.  A candidate whose x falls inside ratioWindow (a generous bound on that) raises flags[2], and
    // the call is refused instead of answering with a possibly different decision (round 5; never seen in a test or fuzz case).
    double ratioWindow;
    float marginScale;          // 1 (CDM_EXTEND_MARGIN, tests: scales the error bounds of the plain-double likelihoods)
    uint64_t maxSeqLen;
    int unsafe; uint32_t minCov;        // --unsafe 1: consensusCaller's majority vote over the extending targets (--min-cov-safe)
    unsigned int *flags;                // [0] set when an unsafe-mode consensus would write outside its 3 qLen letters (undefined in the reference)
};

// Letter of a sequence as this module sees it.  The reference works on the original bytes: == 'N' tests, letter identity
// (nuclassembleUtil.cpp:431-433), ryMap[c] and nucleotideMap[c] (unordered_maps: 0 for everything they do not hold).  code = 0..3 for
// A,C,G,T; a letter beyond ACGTN (sequences with a row in the raw plane only) comes back as 0x100 | byte, so that == is the
// reference's letter comparison, and goes through fwdBase()/ryClass() wherever a table is indexed.
__device__ __forceinline__ void letterOf(const ExtArgs &A, uint32_t t, uint32_t w, uint32_t p, uint32_t &code, bool &isN) {
    code = cdm_base(A.codes, w, p);
    isN = false;
    if (A.hasN[t]) {
        isN = cdm_isN(A.nmask, w, p);
        if (A.raw && A.hasRaw[t]) {
            const uint8_t r = cdm_raw_at(A.raw, w, p);
            isN = r == 'N';
            if (isN) code = 0; else if (!(r == 'A' || r == 'C' || r == 'G' || r == 'T')) code = 0x100u | r;
        }
    }
}
__device__ __forceinline__ uint32_t fwdBase(uint32_t code) { return code < 4u ? code : 0u; }
__device__ __forceinline__ uint32_t ryClass(uint32_t code) { return code < 4u ? (code & 1u) : 0u; }

// ---- the query as it grows: pieces to the left (most recent first), the original, pieces to the right
struct VQuery {
    const ExtArgs *a;
    uint32_t q, qLen0, qw;
    const Cand *cand; const uint32_t *leftL, *rightL;
    uint32_t nL, nR, leftTotal, total;
    bool plain;   // no pieces yet and no N in the query: positions index the original sequence, word-wise paths apply
    __device__ void baseAt(uint32_t p, uint32_t &code, bool &isN) const {
        const ExtArgs &A = *a;
        uint32_t t, tp;
        if (p < leftTotal) {
            uint32_t acc = 0; t = 0; tp = 0;
            for (int i = (int) nL - 1; i >= 0; i--) { const Cand &c = cand[leftL[i]]; if (p < acc + c.pieceLen) { t = c.target; tp = c.pieceStart + (p - acc); break; } acc += c.pieceLen; }
        } else if (p < leftTotal + qLen0) { t = q; tp = p - leftTotal; }
        else {
            uint32_t acc = leftTotal + qLen0; t = 0; tp = 0;
            for (uint32_t i = 0; i < nR; i++) { const Cand &c = cand[rightL[i]]; if (p < acc + c.pieceLen) { t = c.target; tp = c.pieceStart + (p - acc); break; } acc += c.pieceLen; }
        }
        letterOf(A, t, A.woff[t], tp, code, isN);
    }
    // source (sequence t, position tp) of position p and the number of positions from p on that come from the same piece
    __device__ void spanAt(uint32_t p, uint32_t &t, uint32_t &tp, uint32_t &run) const {
        t = q; tp = 0; run = 1;
        if (p < leftTotal) {
            uint32_t acc = 0;
            for (int i = (int) nL - 1; i >= 0; i--) { const Cand &c = cand[leftL[i]]; if (p < acc + c.pieceLen) { t = c.target; tp = c.pieceStart + (p - acc); run = acc + c.pieceLen - p; break; } acc += c.pieceLen; }
        } else if (p < leftTotal + qLen0) { t = q; tp = p - leftTotal; run = leftTotal + qLen0 - p; }
        else {
            uint32_t acc = leftTotal + qLen0;
            for (uint32_t i = 0; i < nR; i++) { const Cand &c = cand[rightL[i]]; if (p < acc + c.pieceLen) { t = c.target; tp = c.pieceStart + (p - acc); run = acc + c.pieceLen - p; break; } acc += c.pieceLen; }
        }
    }
};

__device__ __forceinline__ void targetBaseAt(const ExtArgs &A, uint32_t t, uint32_t p, uint32_t &code, bool &isN) {
    letterOf(A, t, A.woff[t], p, code, isN);
}

// ---- consensusCaller, unsafe mode (nuclassembleUtil.cpp:570-702 + calculateConsensus :535-567).  The consensus has 3 qLen
// letters: the query in the middle third; elsewhere the majority letter of the alignments that extend the query and cover the
// position, where at least minCov of them do (uncovered, below the coverage and tied positions are 'N').  The letter is worked
// out on demand from the list consensusCaller was called with (all candidates at first, the re-aligned parked hits later):
// the mode is not the default and a pile-up is a handful of records.
struct ConsList { const Cand *cand; const uint32_t *idx; uint32_t n; uint32_t qKey; };
__device__ void consensusAt(const ExtArgs &A, const VQuery &Q, const ConsList &L, uint32_t x, uint32_t &code, bool &isN) {
    const uint32_t qLen = Q.total;
    if (x >= qLen && x < 2 * qLen) { Q.baseAt(x - qLen, code, isN); return; }
    uint32_t cnt[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < L.n; i++) {
        const Cand &c = L.cand[L.idx ? L.idx[i] : i];
        const bool rs = c.ds == 0 && (c.de != static_cast<int>(c.dbLen) - 1), ls = c.qs == 0 && (c.qe != static_cast<int>(c.qLen) - 1);
        if (!(rs || ls) || A.key[c.target] == L.qKey) continue;
        const uint32_t tLen = A.len[c.target];
        long long start;
        if (c.ds == 0 && (uint32_t) c.qe == qLen - 1) start = (long long) qLen + c.qs;                                  // right extension
        else if (c.qs == 0 && (uint32_t) c.de == tLen - 1) start = (long long) qLen - (long long) (c.dbLen - c.alnLen);   // left extension
        else continue;
        // (a target that reaches beyond the 3 qLen letters makes the reference write outside its coverage vector - undefined, but
        // never read back: only positions inside the consensus are looked at, here as there)
        if ((long long) x < start || (long long) x >= start + (long long) c.dbLen) continue;
        uint32_t tc; bool tn;
        targetBaseAt(A, c.target, (uint32_t) ((long long) x - start), tc, tn);
        cnt[tn ? 0u : fwdBase(tc)]++;               // nucleotideMap[c]: any letter outside ACGT counts as 'A'
    }
    code = 0; isN = true;
    if (cnt[0] + cnt[1] + cnt[2] + cnt[3] < A.minCov) return;
    uint32_t mx = 0; int nMax = 0;
    for (uint32_t j = 0; j < 4; j++) {
        if (cnt[j] > mx) { mx = cnt[j]; code = j; nMax = 1; } else if (cnt[j] == mx && mx > 0) nMax++;
    }
    isN = nMax != 1;
    if (isN) code = 0;
}

// identical / same-RY-class columns of q[q0..q0+n) vs t[t0..t0+n), 16 bases per XOR (neither sequence has an N)
__device__ __forceinline__ void countMatchesWords(const ExtArgs &A, uint32_t q, uint32_t q0, uint32_t t, uint32_t t0, uint32_t n, int &idCnt, int &idRy) {
    const uint32_t qw = A.woff[q], tw = A.woff[t];
    const uint32_t qLast = (A.len[q] + 15) / 16 - 1, tLast = (A.len[t] + 15) / 16 - 1;
    uint32_t mm = 0, ry = 0;
    for (uint32_t k = 0; k < n; k += 16) {
        const uint32_t x = cdm_window16(A.codes, qw, q0 + k, qLast) ^ cdm_window16(A.codes, tw, t0 + k, tLast);
        uint32_t any = (x | (x >> 1)) & 0x55555555u, lowbit = x & 0x55555555u;
        const uint32_t rem = n - k;
        if (rem < 16) { const uint32_t m = (1u << (2 * rem)) - 1u; any &= m; lowbit &= m; }
        mm += __popc(any); ry += __popc(lowbit);
    }
    idCnt = (int) (n - mm); idRy = (int) (n - ry);
}

// the same on metadata the caller holds already (no gather of the two sequences' records)
__device__ __forceinline__ void countMatchesWordsAt(const uint32_t *__restrict__ codes, uint32_t qw, uint32_t qLast, uint32_t q0, uint32_t tw, uint32_t tLast, uint32_t t0, uint32_t n, int &idCnt, int &idRy) {
    uint32_t mm = 0, ry = 0;
    for (uint32_t k = 0; k < n; k += 16) {
        const uint32_t x = cdm_window16(codes, qw, q0 + k, qLast) ^ cdm_window16(codes, tw, t0 + k, tLast);
        uint32_t any = (x | (x >> 1)) & 0x55555555u, lowbit = x & 0x55555555u;
        const uint32_t rem = n - k;
        if (rem < 16) { const uint32_t m = (1u << (2 * rem)) - 1u; any &= m; lowbit &= m; }
        mm += __popc(any); ry += __popc(lowbit);
    }
    idCnt = (int) (n - mm); idRy = (int) (n - ry);
}

// updateSeqIdConsensusReads for one candidate on the current query (nuclassembleUtil.cpp:377-500, safe-mode consensus)
__device__ void updateIds(const ExtArgs &A, const VQuery &Q, Cand &c, uint32_t &maxLeft, uint32_t &maxRight, const ConsList *cons = nullptr) {
    const uint32_t qLen = Q.total;
    const bool rightStart = (uint32_t) c.ds == 0 && (uint32_t) c.qe == (qLen - 1);
    const bool leftStart = (uint32_t) c.qs == 0 && (uint32_t) c.de == (c.dbLen - 1);
    int idCnt = 0, idRy = 0; uint32_t tot = 0;
    if (cons && (leftStart || rightStart)) {
        // unsafe mode: the padded target against the whole consensus, overhang included (:389-437); target letter j sits at
        // consensus index c0 + j
        const uint32_t offset = c.dbLen - c.alnLen;
        if (offset > qLen) A.flags[0] = 1u;         // the reference pads with qLen - offset letters: undefined
        else {
            const uint32_t c0 = leftStart ? (qLen - offset) : (2 * qLen - c.alnLen);
            for (uint32_t j = 0; j < c.dbLen && c0 + j < 3 * qLen; j++) {
                uint32_t qc, tc; bool qn, tn;
                consensusAt(A, Q, *cons, c0 + j, qc, qn); targetBaseAt(A, c.target, j, tc, tn);
                if (qn || tn) continue;
                idCnt += (qc == tc); idRy += (ryClass(qc) == ryClass(tc)); tot++;
            }
        }
    } else
    if (leftStart || rightStart) {
        // padded target against N^L query N^L: the columns where both are defined are the overlap itself
        const uint32_t offset = c.dbLen - c.alnLen;
        uint32_t q0, t0, ncol;
        if (leftStart) { t0 = offset; q0 = 0; ncol = min(c.dbLen - offset, qLen); }
        else { t0 = 0; q0 = qLen - c.alnLen; ncol = min(c.alnLen, c.dbLen); }
        if (Q.plain && !A.hasN[c.target]) { countMatchesWords(A, Q.q, q0, c.target, t0, ncol, idCnt, idRy); tot = ncol; }
        else
        for (uint32_t i = 0; i < ncol; i++) {
            uint32_t qc, tc; bool qn, tn;
            Q.baseAt(q0 + i, qc, qn); targetBaseAt(A, c.target, t0 + i, tc, tn);
            if (qn || tn) continue;
            idCnt += (qc == tc); idRy += (ryClass(qc) == ryClass(tc)); tot++;
        }
    }
    if (tot != 0) { c.seqId = static_cast<float>(idCnt) / tot; c.rySeqId = static_cast<float>(idRy) / tot; }
    if (leftStart && tot > maxLeft) maxLeft = tot; else if (rightStart && tot > maxRight) maxRight = tot;
}

// calcLikelihoodConsensus via r_s_pair; returns sRatio > threshold, sets c.sLenNorm
__device__ bool scoreCand(const ExtArgs &A, const VQuery &Q, Cand &c, uint32_t maxLeft, uint32_t maxRight, const double *logLik /* [11][4][4] fwd */,
                          const ConsList *cons = nullptr, const X87 *logLikX = nullptr /* the same table, converted once */) {
    const uint32_t qLen = Q.total;
    uint32_t maxAln = maxRight;
    if ((uint32_t) c.qs == 0 && (uint32_t) c.de == (c.dbLen - 1)) maxAln = maxLeft;
    const bool rightStart = (uint32_t) c.ds == 0 && (uint32_t) c.qe == (qLen - 1);
    const bool leftStart = (uint32_t) c.qs == 0 && (uint32_t) c.de == (c.dbLen - 1);
    X87 lik = x87_zero();
    uint32_t alnCount = 0;
    if (cons && (leftStart || rightStart)) {
        // unsafe mode: every target letter against the consensus letter above it (:225-330), flanks included
        const uint32_t offset = c.dbLen - c.alnLen;
        if (offset > qLen) A.flags[0] = 1u;
        else {
            const uint32_t c0 = leftStart ? (qLen - offset) : (2 * qLen - c.alnLen);
            uint32_t tIdx = 0;
            for (uint32_t j = 0; j < c.dbLen && c0 + j < 3 * qLen; j++) {
                uint32_t qc, tc; bool qn, tn;
                targetBaseAt(A, c.target, j, tc, tn);
                if (!tn) tIdx++;
                if (tn) continue;
                consensusAt(A, Q, *cons, c0 + j, qc, qn);
                if (qn) continue;
                alnCount++;
                const uint32_t ti = tIdx - 1;
                const uint32_t cls = ti < 5 ? ti : (ti >= c.dbLen - 5 ? 6 + (ti - (c.dbLen - 5)) : 5);
                lik = x87_add(lik, x87_from_double(logLik[(cls * 4 + fwdBase(qc)) * 4 + fwdBase(tc)]));
            }
        }
    } else
    if (leftStart || rightStart) {
        const uint32_t offset = c.dbLen - c.alnLen;
        uint32_t q0, t0, ncol;
        if (leftStart) { t0 = offset; q0 = 0; ncol = min(c.dbLen - offset, qLen); }
        else { t0 = 0; q0 = qLen - c.alnLen; ncol = min(c.alnLen, c.dbLen); }
        // tIdx counts the non-N target letters up to and including the column (pad letters are 'N'); before the overlap
        // the left-start target contributes its own prefix t[0..offset)
        uint32_t tIdx = 0;
        if (Q.plain && !A.hasN[c.target]) {
            // no N anywhere: tIdx - 1 is the target position itself; walk the two sequences a 16-base word at a time
            const uint32_t qw = A.woff[Q.q], tw = A.woff[c.target];
            const uint32_t qLast = (Q.qLen0 + 15) / 16 - 1, tLast = (c.dbLen + 15) / 16 - 1;
            const uint32_t tailFrom = c.dbLen - 5;
            if (logLikX) {
            uint32_t qNext = cdm_window16(A.codes, qw, q0, qLast), tNext = cdm_window16(A.codes, tw, t0, tLast);
            for (uint32_t k = 0; k < ncol; k += 16) {
                uint32_t qwin = qNext, twin = tNext;          // (the next 16 columns are on their way while these are summed)
                if (k + 16 < ncol) { qNext = cdm_window16(A.codes, qw, q0 + k + 16, qLast); tNext = cdm_window16(A.codes, tw, t0 + k + 16, tLast); }
                const uint32_t m = min(16u, ncol - k);
                for (uint32_t j = 0; j < m; j++) {
                    const uint32_t ti = t0 + k + j;
                    const uint32_t cls = ti < 5 ? ti : (ti >= tailFrom ? 6 + (ti - tailFrom) : 5);
                    lik = x87_acc(lik, logLikX[(cls * 4 + (qwin & 3u)) * 4 + (twin & 3u)]);
                    qwin >>= 2; twin >>= 2;
                }
            }
            } else
            for (uint32_t k = 0; k < ncol; k += 16) {
                uint32_t qwin = cdm_window16(A.codes, qw, q0 + k, qLast), twin = cdm_window16(A.codes, tw, t0 + k, tLast);
                const uint32_t m = min(16u, ncol - k);
                for (uint32_t j = 0; j < m; j++) {
                    const uint32_t ti = t0 + k + j;
                    const uint32_t cls = ti < 5 ? ti : (ti >= tailFrom ? 6 + (ti - tailFrom) : 5);
                    lik = x87_acc(lik, x87_from_double(logLik[(cls * 4 + (qwin & 3u)) * 4 + (twin & 3u)]));
                    qwin >>= 2; twin >>= 2;
                }
            }
            alnCount = ncol;
        } else {
        if (leftStart) for (uint32_t j = 0; j < t0; j++) { uint32_t tc; bool tn; targetBaseAt(A, c.target, j, tc, tn); tIdx += !tn; }
        for (uint32_t i = 0; i < ncol; i++) {
            uint32_t qc, tc; bool qn, tn;
            Q.baseAt(q0 + i, qc, qn); targetBaseAt(A, c.target, t0 + i, tc, tn);
            if (!tn) tIdx++;
            if (qn || tn) continue;
            alnCount++;
            const uint32_t ti = tIdx - 1;
            const uint32_t cls = ti < 5 ? ti : (ti >= c.dbLen - 5 ? 6 + (ti - (c.dbLen - 5)) : 5);
            lik = x87_add(lik, x87_from_double(logLik[(cls * 4 + fwdBase(qc)) * 4 + fwdBase(tc)]));
        }
        }
    }
    const uint32_t excess = maxAln - alnCount;
    const float pen = (float) excess * A.excessLog;            // unsigned * float
    lik = x87_add(lik, x87_from_double((double) pen));
    const double randAln = (double) ((float) maxAln * A.randLog);
    c.sLenNorm = x87_to_double(lik);
    X87 neg = lik; neg.s ^= 1u;
    const double x = x87_to_double(x87_add(x87_from_double(randAln), neg));   // randAln - likMod
    if (A.ratioWindow > 0.0 && fabs(x - A.ratioLogit) <= A.ratioWindow) A.flags[2] = 1u;
    return x < A.ratioLogit;
}

// The same for an end overlap of two sequences without N, in plain double - with a bound on how far that can be from what
// scoreCand() returns, so that the caller knows when the cheap sum decides.  The reference adds the alnCount column terms and the
// excess penalty one by one into a long double (64-bit significand, error <= 2^-64 per addition relative to the running sum) and
// rounds once to double; the plain sum errs by <= 2^-53 per addition.  With S = the sum of the |terms| (every running sum is
// below it), both sums lie within (alnCount + 2) * 2^-53 * S of the exact one; bound = (alnCount + 8) * 2^-52 * S is more than
// twice that.  Returns 1 / 0 when randAln - sum lies outside ratioLogit +- bound (scoreCand's answer is then the same), -1 when not;
// sLenNorm = the plain sum, bound as above.
__device__ int scoreCandApprox(const ExtArgs &A, const VQuery &Q, const Cand &c, uint32_t maxLeft, uint32_t maxRight, const double *logLik, double &sLenNorm, float &bound) {
    const uint32_t qLen = Q.total;
    const bool leftStart = (uint32_t) c.qs == 0 && (uint32_t) c.de == (c.dbLen - 1);
    const uint32_t maxAln = leftStart ? maxLeft : maxRight;
    const uint32_t offset = c.dbLen - c.alnLen;
    uint32_t q0, t0, ncol;
    if (leftStart) { t0 = offset; q0 = 0; ncol = min(c.dbLen - offset, qLen); }
    else { t0 = 0; q0 = qLen - c.alnLen; ncol = min(c.alnLen, c.dbLen); }
    const uint32_t qw = A.woff[Q.q], tw = A.woff[c.target];
    const uint32_t qLast = (Q.qLen0 + 15) / 16 - 1, tLast = (c.dbLen + 15) / 16 - 1, tailFrom = c.dbLen - 5;
    double sum = 0.0, mag = 0.0;
    uint32_t qNext = cdm_window16(A.codes, qw, q0, qLast), tNext = cdm_window16(A.codes, tw, t0, tLast);
    for (uint32_t k = 0; k < ncol; k += 16) {
        uint32_t qwin = qNext, twin = tNext;
        if (k + 16 < ncol) { qNext = cdm_window16(A.codes, qw, q0 + k + 16, qLast); tNext = cdm_window16(A.codes, tw, t0 + k + 16, tLast); }
        const uint32_t m = min(16u, ncol - k);
        for (uint32_t j = 0; j < m; j++) {
            const uint32_t ti = t0 + k + j;
            const uint32_t cls = ti < 5 ? ti : (ti >= tailFrom ? 6 + (ti - tailFrom) : 5);
            const double t = logLik[(cls * 4 + (qwin & 3u)) * 4 + (twin & 3u)];
            sum = __dadd_rn(sum, t); mag = __dadd_rn(mag, fabs(t));
            qwin >>= 2; twin >>= 2;
        }
    }
    const uint32_t excess = maxAln - ncol;
    const double pen = (double) ((float) excess * A.excessLog);
    sum = __dadd_rn(sum, pen); mag = __dadd_rn(mag, fabs(pen));
    const double randAln = (double) ((float) maxAln * A.randLog);
    const double b = fmax((double) (ncol + 8u) * 0x1p-52 * (mag + fabs(randAln)) * (double) A.marginScale, A.ratioWindow);       // (inside the window the exact sum is taken, and flags the call)
    sLenNorm = sum; bound = (float) b * 1.0000002f + 1e-37f;       // (rounded up: the float is what the queue's near-tie test adds)
    const double x = __dadd_rn(randAln, -sum);
    if (x < A.ratioLogit - b) return 1;
    if (x > A.ratioLogit + b) return 0;
    return -1;
}

// ---- std::priority_queue<scorePerRes, vector, CompareNuclResultByScoreReads> on candidate indices (libstdc++ heap order)
struct Heap {
    uint32_t *h; uint32_t n; const Cand *cand;
    __device__ bool less(uint32_t x, uint32_t y) const { return cand[x].sLenNorm < cand[y].sLenNorm; }
    __device__ void siftUp(uint32_t hole, uint32_t top, uint32_t value) {
        while (hole > top) { const uint32_t parent = (hole - 1) / 2; if (!less(h[parent], value)) break; h[hole] = h[parent]; hole = parent; }
        h[hole] = value;
    }
    __device__ void push(uint32_t v) { n++; siftUp(n - 1, 0, v); }
    __device__ uint32_t pop() {
        const uint32_t topV = h[0];
        const uint32_t len = n - 1;
        if (len > 0) {
            const uint32_t value = h[len];
            uint32_t hole = 0, child = 0;
            while (child < (len - 1) / 2) { child = 2 * (child + 1); if (less(h[child], h[child - 1])) child--; h[hole] = h[child]; hole = child; }
            if ((len & 1u) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); h[hole] = h[child - 1]; hole = child - 1; }
            siftUp(hole, 0, value);
        }
        n = len;
        return topV;
    }
};

// ---- E: the extension loop of one query (ancientReadsResults.cpp:374-546) on its heap of scored candidates; writes the query's
// result (new length, piece lists).  Shared by the two forms of the kernel below.
__device__ void extendLoop(const ExtArgs &A, VQuery &Q, Cand *cand, Heap &heap, uint32_t *parkL, uint32_t *leftL, uint32_t *rightL,
                           uint32_t maxLeft, uint32_t maxRight, const double *sLogLik, uint32_t qKey, uint32_t q) {
    bool couldExtend = false;
    while (heap.n > 0) {
        uint32_t leftOff = 0, rightOff = 0, nPark = 0;
        const uint32_t qLen = Q.total;
        while (true) {
            // selectNuclFragmentToExtendReads (:76-92)
            bool found = false; uint32_t bi = 0;
            while (heap.n > 0) {
                const uint32_t k = heap.pop();
                const Cand &c = cand[k];
                const bool notBoth = !(c.ds == 0 && c.qs == 0);
                const bool rightStart = c.ds == 0 && (c.de != static_cast<int>(c.dbLen) - 1);
                const bool leftStart = c.qs == 0 && (c.qe != static_cast<int>(c.qLen) - 1);
                if ((rightStart || leftStart) && notBoth && c.tKey != qKey) { found = true; bi = k; break; }
            }
            if (!found) break;
            Cand &b = cand[bi];
            const uint32_t tLen = b.dbLen;                   // (= the target's length, since the candidate was made)
            if (b.ds == 0) { if ((tLen - (uint32_t) (b.de + 1)) <= rightOff) continue; }
            else if (b.qs == 0) { if (b.ds <= static_cast<int>(leftOff)) continue; }
            const uint32_t ds = (uint32_t) b.ds, de = (uint32_t) b.de, qs = (uint32_t) b.qs, qe = (uint32_t) b.qe;
            if (ds == 0 && qe == (qLen - 1)) {
                if (rightOff > 0) { parkL[nPark++] = bi; continue; }
                const uint32_t fragLen = tLen - (de + 1);
                if ((uint64_t) Q.total + fragLen >= A.maxSeqLen) break;
                b.pieceStart = de + 1; b.pieceLen = fragLen;
                rightL[Q.nR++] = bi; Q.total += fragLen; rightOff += fragLen; Q.plain = false;
            } else if (qs == 0 && de == (tLen - 1)) {
                if (leftOff > 0) { parkL[nPark++] = bi; continue; }
                const uint32_t fragLen = ds;
                if ((uint64_t) Q.total + fragLen >= A.maxSeqLen) break;
                b.pieceStart = 0; b.pieceLen = fragLen;
                leftL[Q.nL++] = bi; Q.leftTotal += fragLen; Q.total += fragLen; leftOff += fragLen; Q.plain = false;
            }
        }
        if (leftOff > 0 || rightOff > 0) couldExtend = true;
        if (heap.n > 0) break;
        // ---- re-align the parked hits on the grown query (:484-509)
        const uint32_t newLen = Q.total;
        for (uint32_t i = 0; i < nPark; i++) {
            Cand &c = cand[parkL[i]];
            const uint32_t tLen = A.len[c.target];
            const int diag = (c.qs + (int) leftOff) - c.ds;
            const unsigned md = (unsigned) abs(diag);
            // ungappedAlignmentByDiagonal, mode 3 (whole overlap, score clamped at 0)
            uint32_t qOff = 0, tOff = 0, m = 0; int startPos = -1, endPos = -1; unsigned score = 0;
            bool has = false;
            if (diag >= 0 && md < newLen) { qOff = md; tOff = 0; m = min(tLen, newLen - md); has = true; }
            else if (diag < 0 && md < tLen) { qOff = 0; tOff = md; m = min(tLen - md, newLen); has = true; }
            if (has) {
                long long sc = 0;
                for (uint32_t j = 0; j < m; j++) {
                    uint32_t qc, tc; bool qn, tn;
                    Q.baseAt(qOff + j, qc, qn); targetBaseAt(A, c.target, tOff + j, tc, tn);
                    sc += (!qn && !tn && qc == tc) ? 2 : -3;
                }
                score = sc > 0 ? (unsigned) sc : 0u; startPos = 0; endPos = (int) m - 1;
                if (A.raw) {        // a '*' at either end of the overlap is left out (DistanceCalculator.h:204-220)
                    constexpr uint32_t STAR = 0x100u | '*';
                    uint32_t x0, y0, x1, y1; bool n0;
                    Q.baseAt(qOff, x0, n0); targetBaseAt(A, c.target, tOff, y0, n0); Q.baseAt(qOff + m - 1, x1, n0); targetBaseAt(A, c.target, tOff + m - 1, y1, n0);
                    if (x0 == STAR || y0 == STAR) startPos = 1;
                    if (m > 1 && (x1 == STAR || y1 == STAR)) endPos--;
                }
            }
            // updateNuclAlignment (nuclassembleUtil.cpp:9-47)
            int qs2, qe2, ds2, de2; const int dist = (int) md;
            if (diag >= 0) { qs2 = startPos + dist; qe2 = endPos + dist; ds2 = startPos; de2 = endPos; }
            else { qs2 = startPos; qe2 = endPos; ds2 = startPos + dist; de2 = endPos + dist; }
            int idCnt = 0;
            for (int j = qs2; j < qe2; j++) {
                uint32_t qc, tc; bool qn, tn;
                Q.baseAt((uint32_t) j, qc, qn); targetBaseAt(A, c.target, (uint32_t) (ds2 + (j - qs2)), tc, tn);
                idCnt += ((qn ? 4u : qc) == (tn ? 4u : tc));
            }
            c.seqId = static_cast<float>(idCnt) / (static_cast<float>(qe2) - static_cast<float>(qs2));
            c.qLen = newLen; c.dbLen = tLen; c.alnLen = has ? m : 0;
            (void) score;
            c.qs = qs2; c.qe = qe2; c.ds = ds2; c.de = de2;
        }
        ConsList consPark; consPark.cand = cand; consPark.idx = parkL; consPark.n = nPark; consPark.qKey = qKey;
        const ConsList *cons1 = A.unsafe ? &consPark : nullptr;
        for (uint32_t i = 0; i < nPark; i++) updateIds(A, Q, cand[parkL[i]], maxLeft, maxRight, cons1);
        for (uint32_t i = 0; i < nPark; i++) {
            Cand &c = cand[parkL[i]];
            const bool notInside = c.dbLen != c.alnLen;
            const bool rightStart = c.ds == 0, leftStart = c.qs == 0, notId = c.tKey != qKey;
            if (c.seqId >= A.seqIdThr && (rightStart || leftStart) && notId && notInside) {
                if (scoreCand(A, Q, c, maxLeft, maxRight, sLogLik, cons1)) heap.push(parkL[i]);
            }
        }
    }
    A.newLen[q] = couldExtend ? Q.total : 0;
    A.nLeft[q] = Q.nL; A.nRight[q] = Q.nR; A.leftTotal[q] = Q.leftTotal;
}

// RAW = false: the DB has no letters beyond ACGTN - the raw-plane branches of letterOf() fold away (a null pointer the compiler knows)
template <int MINW, bool RAW = false>
__global__ __launch_bounds__(64, MINW) void k_extend(ExtArgs A) {
    if (!RAW) A.raw = nullptr;
    __shared__ double sLogLik[11 * 16];
    for (int i = threadIdx.x; i < 11 * 16; i += blockDim.x) sLogLik[i] = (&A.lut->logLik[0][0][0][0])[i];
    __syncthreads();
    const unsigned int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= *A.nActive) return;
    const uint32_t q = A.active[item];
    const uint64_t r0 = A.aoff[q], r1 = A.aoff[q + 1];
    const uint32_t nRec = (uint32_t) (r1 - r0);
    const uint32_t qLen0 = A.len[q], qKey = A.key[q];
    Cand *cand = A.cand + r0;
    uint32_t *heapL = A.lists + 4 * r0, *parkL = heapL + nRec, *leftL = parkL + nRec, *rightL = leftL + nRec;
    VQuery Q; Q.a = &A; Q.q = q; Q.qLen0 = qLen0; Q.qw = A.woff[q]; Q.cand = cand; Q.leftL = leftL; Q.rightL = rightL;
    Q.nL = 0; Q.nR = 0; Q.leftTotal = 0; Q.total = qLen0; Q.plain = A.hasN[q] == 0;

    // ---- A-C: candidates ("notContig"), in record order.  The loop is a chain of dependent gathers - record, the target's metadata,
    // its letters - per lane: the next record and its target's metadata are fetched while the current one is worked on.
    uint32_t nCand = 0;   // candidate k lives at cand[k] (compacted), keeping the record index for the score output
    const SeqMeta *meta = A.len.m;
    const uint32_t qLastW = (qLen0 + 15) / 16 - 1;
    AlnRec recN = A.rec[r0]; SeqMeta tmN = meta[recN.target];
    for (uint32_t r = 0; r < nRec; r++) {
        const AlnRec rec = recN; const SeqMeta tm = tmN;
        if (r + 1 < nRec) { recN = A.rec[r0 + r + 1]; tmN = meta[recN.target]; }
        if (A.scores) A.scores[r0 + r] = NAN;
        const uint32_t tLen = tm.len;
        const uint32_t ds = (uint32_t) rec.dbStart, de = (uint32_t) rec.dbEnd, qs = (uint32_t) rec.qStart, qe = (uint32_t) rec.qEnd;
        const bool rightStart = ds == 0 && qe == (qLen0 - 1);
        const bool leftStart = qs == 0 && de == (tLen - 1);
        if (!rightStart && !leftStart) continue;
        if (rec.qStart > rec.qEnd) continue;   // cannot happen given the test above; reverse strand never extends
        const uint32_t alnLen = (uint32_t) max(abs(rec.qEnd - rec.qStart), abs(rec.dbEnd - rec.dbStart)) + 1u;
        float seqId = rec.seqId, rySeqId = 0.f;
        if (rec.target != qKey) {   // the reference compares the target's *id* with the query's *key* (:264)
            int idCnt = 0, idRy = 0;
            if (Q.plain && !(tm.flags & 1u)) countMatchesWordsAt(A.codes, Q.qw, qLastW, (uint32_t) rec.qStart, tm.woff, (tLen + 15) / 16 - 1, (uint32_t) rec.dbStart, (uint32_t) (rec.qEnd - rec.qStart + 1), idCnt, idRy);
            else
            for (int i = rec.qStart; i <= rec.qEnd; i++) {
                uint32_t qc, tc; bool qn, tn;
                Q.baseAt((uint32_t) i, qc, qn); targetBaseAt(A, rec.target, (uint32_t) (rec.dbStart + (i - rec.qStart)), tc, tn);
                // letters are compared: N == N; N maps to purine (0) in ryMap
                const uint32_t ql = qn ? 4u : qc, tl = tn ? 4u : tc;
                idCnt += (ql == tl);
                idRy += (ryClass(qn ? 0u : qc) == ryClass(tn ? 0u : tc));
            }
            seqId = static_cast<float>(idCnt) / alnLen; rySeqId = static_cast<float>(idRy) / alnLen;
        }
        const bool noOffset = (tLen - alnLen) == 0;
        if (((tm.flags >> 1) & 1u) == 0 && alnLen >= 30 && seqId >= A.seqIdThr && !noOffset) {
            Cand c; c.qs = rec.qStart; c.qe = rec.qEnd; c.ds = rec.dbStart; c.de = rec.dbEnd; c.target = rec.target; c.alnLen = alnLen; c.dbLen = tLen;
            c.qLen = qLen0; c.seqId = seqId; c.rySeqId = rySeqId; c.sLenNorm = 0; c.pieceStart = r; c.pieceLen = 0;   // pieceStart keeps the record index until used
            c.tKey = tm.key;
            cand[nCand++] = c;
        }
    }
    if (nCand == 0) { A.newLen[q] = 0; return; }
    uint32_t maxLeft = 0, maxRight = 0;
    ConsList consAll; consAll.cand = cand; consAll.idx = nullptr; consAll.n = nCand; consAll.qKey = qKey;
    const ConsList *cons0 = A.unsafe ? &consAll : nullptr;
    for (uint32_t k = 0; k < nCand; k++) {
        Cand &c = cand[k];
        if (A.unsafe) updateIds(A, Q, c, maxLeft, maxRight, cons0);
        else if (Q.plain && !A.hasN[c.target] && c.target != qKey) {      // (the candidate pass compares the target id with the query key)
            // updateSeqIdConsensusReads would count the very columns the candidate pass above just counted (an end overlap of
            // two sequences without N): seqId / rySeqId stand, only the longest overlap per side is updated
            const bool rightStart = (uint32_t) c.ds == 0 && (uint32_t) c.qe == (qLen0 - 1);
            const bool leftStart = (uint32_t) c.qs == 0 && (uint32_t) c.de == (c.dbLen - 1);
            const uint32_t offset = c.dbLen - c.alnLen;
            const uint32_t tot = leftStart ? min(c.dbLen - offset, qLen0) : (rightStart ? min(c.alnLen, c.dbLen) : 0u);
            if (leftStart && tot > maxLeft) maxLeft = tot; else if (rightStart && tot > maxRight) maxRight = tot;
        } else updateIds(A, Q, c, maxLeft, maxRight);
    }
    // ---- D
    Heap heap; heap.h = heapL; heap.n = 0; heap.cand = cand;
    for (uint32_t k = 0; k < nCand; k++) {
        Cand &c = cand[k];
        const bool notInside = c.dbLen != c.alnLen;
        const bool rightStart = c.ds == 0, leftStart = c.qs == 0, notId = c.tKey != qKey;
        if ((rightStart || leftStart) && notInside && notId && c.rySeqId >= A.rySeqIdThr && c.seqId >= A.seqIdThr) {
            const bool pass = scoreCand(A, Q, c, maxLeft, maxRight, sLogLik, cons0);
            if (A.scores) A.scores[r0 + c.pieceStart] = c.sLenNorm;
            if (pass) heap.push(k);
        }
    }
    // ---- E
    extendLoop(A, Q, cand, heap, parkL, leftL, rightL, maxLeft, maxRight, sLogLik, qKey, q);
}

// ---------------------------------------------------------------------------------------------- A-D, one thread per RECORD
// k_extend above walks a query's records in one lane: a pile-up of 20 records is 20 dependent rounds of (record, target metadata,
// target letters) gathers per lane, each of which pulls a whole line for 16-32 useful bytes, twice (candidate pass, scoring pass),
// and a wave is as slow as its deepest query.  The candidate test, the identities and the likelihood of a record depend on nothing
// but that record and its query - except for the longest overlap per side (maxAlnLeft/Right), a maximum over the query's
// candidates.  So, for the default mode (safe consensus, no raw plane):
//   k_xr_windows   the records are cut into windows of XR_T; a block owns the queries whose FIRST record lies in its window
//   k_xr_score     per block: pass 1 - one thread per record: candidate test + identities (:202-315, updateSeqIdConsensusReads),
//                  maximum per side into LDS; pass 2 - eligibility + calcLikelihoodConsensus per record (:317-372), sLenNorm of the
//                  records that enter the queue to sLen[]; queries with at least one such record to the work list
//   k_xr_extend    one thread per query of the work list: queue in record order (the push order of the reference), extendLoop()
// Same arithmetic, same order of pushes; a candidate that is a left AND a right overlap at once (whose contribution to the two
// maxima depends on the order of the records in the reference) sets X.fallback and the call is redone by k_extend.
constexpr int XR_NT = 512, XR_T = 448, XR_QCAP = 512, XR_LCAP = 1024, XR_BINS = 32;
struct XrArgs {
    const uint32_t *winQ;      // [windows + 1] first query of every window
    float2 *lite;              // [alignment count] (seqId, rySeqId) of a candidate record; seqId = NaN: not a candidate
    double *sLen;              // [alignment count] sLenNorm of a record that enters the queue (bit set in pushBits)
    float *sBound;             // [alignment count] 0: sLen is scoreCand's value; > 0: the plain-double sum, within this of it (scoreCandApprox)
    uint32_t *qMax;            // [2 n] maxAlnLeft, maxAlnRight of the queries on the work list
    uint32_t *elig;            // [alignment count] scratch: the records of a block that are scored (pass 1 -> pass 2)
    uint32_t *pushBits;        // [alignment count / 32 + 1] bit r: record r enters the queue
    uint32_t *work; unsigned int *nWork, *fallback;
};
__global__ void k_xr_windows(const uint64_t *__restrict__ aoff, uint32_t n, uint32_t nWin, uint32_t *__restrict__ winQ) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nWin) return;
    const uint64_t x = (uint64_t) b * XR_T;
    uint32_t lo = 0, hi = n;                 // first q in [0, n) with aoff[q] >= x, n if none
    while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (aoff[mid] < x) lo = mid + 1; else hi = mid; }
    winQ[b] = lo;
}
__device__ __forceinline__ void xrFill(Cand &c, const AlnRec &rec, uint32_t alnLen, const SeqMeta &tm, uint32_t qLen0, float seqId, float rySeqId, uint32_t r) {
    const uint32_t tLen = tm.len; c.tKey = tm.key;
    c.qs = rec.qStart; c.qe = rec.qEnd; c.ds = rec.dbStart; c.de = rec.dbEnd; c.target = rec.target; c.alnLen = alnLen; c.dbLen = tLen;
    c.qLen = qLen0; c.seqId = seqId; c.rySeqId = rySeqId; c.sLenNorm = 0; c.pieceStart = r; c.pieceLen = 0;
}
__device__ __forceinline__ uint32_t xrAlnLen(const AlnRec &rec) { return (uint32_t) max(abs(rec.qEnd - rec.qStart), abs(rec.dbEnd - rec.dbStart)) + 1u; }
__device__ __forceinline__ void xrQuery(VQuery &Q, const ExtArgs &A, uint32_t q, const SeqMeta &qm) {
    Q.a = &A; Q.q = q; Q.qLen0 = qm.len; Q.qw = qm.woff; Q.cand = nullptr; Q.leftL = nullptr; Q.rightL = nullptr;
    Q.nL = 0; Q.nR = 0; Q.leftTotal = 0; Q.total = qm.len; Q.plain = (qm.flags & 1u) == 0;
}

#ifndef CDM_XRS_MINB
#define CDM_XRS_MINB 3      // blocks per CU the record kernel's registers leave room for (1 / 3 / 4: 21.1 / 20.85 / 20.95 ms)
#endif
__global__ __launch_bounds__(XR_NT, CDM_XRS_MINB) void k_xr_score(ExtArgs A, XrArgs X) {
    A.raw = nullptr;
    __shared__ double sLogLik[11 * 16];
    __shared__ X87 sLogLikX[11 * 16];
    __shared__ uint32_t sOff[XR_QCAP + 1], sMaxL[XR_QCAP], sMaxR[XR_QCAP], sCnt[XR_QCAP];
    // the scored records of the block, ordered by the number of columns (classes of 8) so that the lanes of a wave walk overlaps
    // of about the same length; beyond XR_LCAP of them: unordered, through the scratch list
    __shared__ uint32_t sListA[XR_LCAP], sListB[XR_LCAP];
    __shared__ uint8_t sBinOf[XR_LCAP];
    __shared__ unsigned int sElig, sBinCnt[XR_BINS], sBinBase[XR_BINS];
    for (int i = threadIdx.x; i < 11 * 16; i += XR_NT) { const double v = (&A.lut->logLik[0][0][0][0])[i]; sLogLik[i] = v; sLogLikX[i] = x87_from_double(v); }
    const uint32_t qFirst = X.winQ[blockIdx.x], qEnd = X.winQ[blockIdx.x + 1];
    const SeqMeta *meta = A.len.m;
    for (uint32_t qb = qFirst; qb < qEnd; qb += XR_QCAP) {      // (more than XR_QCAP queries in a window: only when some have no record)
        const uint32_t nq = min((uint32_t) XR_QCAP, qEnd - qb);
        __syncthreads();
        const uint64_t ra = A.aoff[qb];
        for (uint32_t i = threadIdx.x; i <= nq; i += XR_NT) sOff[i] = (uint32_t) (A.aoff[qb + i] - ra);
        for (uint32_t i = threadIdx.x; i < nq; i += XR_NT) { sMaxL[i] = 0; sMaxR[i] = 0; sCnt[i] = 0; }
        if (threadIdx.x == 0) sElig = 0;
        if (threadIdx.x < XR_BINS) sBinCnt[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t nr = sOff[nq];
        // ---- pass 1: candidates (A-C) and the longest overlap per side
        for (uint32_t base = 0; base < nr; base += XR_NT) {
            const uint32_t i = base + threadIdx.x;
            if (i >= nr) continue;
            uint32_t lo = 0, hi = nq;                           // owner: the last ordinal with sOff[ord] <= i
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (sOff[mid] <= i) lo = mid; else hi = mid; }
            const uint32_t ord = lo, q = qb + ord;
            float2 out = make_float2(NAN, 0.f);
            if (sOff[ord + 1] - sOff[ord] > 1) {
                const AlnRec rec = A.rec[ra + i]; const SeqMeta qm = meta[q];
                const uint32_t qLen0 = qm.len, qKey = qm.key;
                const uint32_t ds = (uint32_t) rec.dbStart, de = (uint32_t) rec.dbEnd, qs = (uint32_t) rec.qStart, qe = (uint32_t) rec.qEnd;
                const bool rightStart = ds == 0 && qe == (qLen0 - 1);
                // (the target's metadata - a random 16-byte read - only for a record that can still be an end overlap: forward strand, and
                // either a right start or a query start at 0, which a left start needs)
                SeqMeta tm = {0, 0, 0, 0};
                const bool maybe = !(rec.qStart > rec.qEnd) && (rightStart || qs == 0);
                if (maybe) tm = meta[rec.target];
                const uint32_t tLen = tm.len;
                const bool leftStart = maybe && qs == 0 && de == (tLen - 1);
                if (maybe && (rightStart || leftStart)) {
                    const uint32_t alnLen = xrAlnLen(rec);
                    const bool plain = (qm.flags & 1u) == 0 && (tm.flags & 1u) == 0;
                    float seqId = rec.seqId, rySeqId = 0.f;
                    if (rec.target != qKey) {
                        int idCnt = 0, idRy = 0;
                        if (plain) countMatchesWordsAt(A.codes, qm.woff, (qLen0 + 15) / 16 - 1, (uint32_t) rec.qStart, tm.woff, (tLen + 15) / 16 - 1, (uint32_t) rec.dbStart, (uint32_t) (rec.qEnd - rec.qStart + 1), idCnt, idRy);
                        else
                        for (int j = rec.qStart; j <= rec.qEnd; j++) {
                            uint32_t qc, tc; bool qn, tn;
                            letterOf(A, q, qm.woff, (uint32_t) j, qc, qn); letterOf(A, rec.target, tm.woff, (uint32_t) (rec.dbStart + (j - rec.qStart)), tc, tn);
                            const uint32_t ql = qn ? 4u : qc, tl = tn ? 4u : tc;
                            idCnt += (ql == tl);
                            idRy += (ryClass(qn ? 0u : qc) == ryClass(tn ? 0u : tc));
                        }
                        seqId = static_cast<float>(idCnt) / alnLen; rySeqId = static_cast<float>(idRy) / alnLen;
                    }
                    const bool noOffset = (tLen - alnLen) == 0;
                    if (((tm.flags >> 1) & 1u) == 0 && alnLen >= 30 && seqId >= A.seqIdThr && !noOffset) {
                        if (leftStart && rightStart) X.fallback[0] = 1u;
                        uint32_t mL = 0, mR = 0;
                        if (plain && rec.target != qKey) {
                            const uint32_t offset = tLen - alnLen;
                            const uint32_t tot = leftStart ? min(tLen - offset, qLen0) : min(alnLen, tLen);
                            if (leftStart) mL = tot; else mR = tot;
                        } else {
                            Cand c; xrFill(c, rec, alnLen, tm, qLen0, seqId, rySeqId, i);
                            VQuery Q; xrQuery(Q, A, q, qm);
                            updateIds(A, Q, c, mL, mR);
                            seqId = c.seqId; rySeqId = c.rySeqId;
                        }
                        if (mL) atomicMax(&sMaxL[ord], mL);
                        if (mR) atomicMax(&sMaxR[ord], mR);
                        out = make_float2(seqId, rySeqId);
                        // D's gate (:317-330) needs nothing of the other records: the scored ones go to the block's list
                        const bool notInside = tLen != alnLen, notId = tm.key != qKey;
                        if ((rec.dbStart == 0 || rec.qStart == 0) && notInside && notId && rySeqId >= A.rySeqIdThr && seqId >= A.seqIdThr) {
                            const uint32_t slot = atomicAdd(&sElig, 1u);
                            if (slot < XR_LCAP) {
                                const uint32_t bin = min((uint32_t) XR_BINS - 1u, alnLen >> 3);
                                sListA[slot] = i; sBinOf[slot] = (uint8_t) bin; atomicAdd(&sBinCnt[bin], 1u);
                            } else X.elig[ra + slot] = i;
                        }
                    }
                }
            }
            if (out.x == out.x) X.lite[ra + i] = out;              // (read back for the records on the block's list / with a push bit only)
            if (A.scores && sOff[ord + 1] - sOff[ord] > 1) A.scores[ra + i] = NAN;
        }
        __syncthreads();
        // ---- pass 2: D - the likelihood of the scored candidates, densely over the list; those above the threshold enter the queue
        const uint32_t nElig = sElig, nListed = min(nElig, (uint32_t) XR_LCAP);
        if (threadIdx.x == 0) { unsigned int acc = 0; for (int b = 0; b < XR_BINS; b++) { sBinBase[b] = acc; acc += sBinCnt[b]; } }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < nListed; k += XR_NT) sListB[atomicAdd(&sBinBase[sBinOf[k]], 1u)] = sListA[k];
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < nElig; k += XR_NT) {
            const uint32_t i = k < XR_LCAP ? sListB[k] : X.elig[ra + k];
            uint32_t lo = 0, hi = nq;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (sOff[mid] <= i) lo = mid; else hi = mid; }
            const uint32_t ord = lo, q = qb + ord;
            const float2 l = X.lite[ra + i];
            const AlnRec rec = A.rec[ra + i]; const SeqMeta tm = meta[rec.target], qm = meta[q];
            Cand c; xrFill(c, rec, xrAlnLen(rec), tm, qm.len, l.x, l.y, i);
            VQuery Q; xrQuery(Q, A, q, qm);
            // the plain-double sum decides unless it lies within its error bound of the threshold (or the scores are asked for)
            bool pass; float bnd = 0.f;
            int quick = -1;
            if (!A.scores && Q.plain && (tm.flags & 1u) == 0) { double sl; quick = scoreCandApprox(A, Q, c, sMaxL[ord], sMaxR[ord], sLogLik, sl, bnd); c.sLenNorm = sl; }
            if (quick < 0) { pass = scoreCand(A, Q, c, sMaxL[ord], sMaxR[ord], sLogLik, nullptr, sLogLikX); bnd = 0.f; }
            else pass = quick != 0;
            if (A.scores) A.scores[ra + i] = c.sLenNorm;
            if (pass) {
                X.sLen[ra + i] = c.sLenNorm; X.sBound[ra + i] = bnd;
                atomicAdd(&sCnt[ord], 1u);
                atomicOr(&X.pushBits[(ra + i) >> 5], 1u << ((ra + i) & 31u));
            }
        }
        __syncthreads();
        for (uint32_t i0 = 0; i0 < nq; i0 += XR_NT) {
            const uint32_t i = i0 + threadIdx.x;
            const bool has = i < nq && sCnt[i] != 0;
            const uint32_t slot = cdm_block_append(X.nWork, has);
            if (has) { X.work[slot] = qb + i; X.qMax[2 * (size_t) (qb + i)] = sMaxL[i]; X.qMax[2 * (size_t) (qb + i) + 1] = sMaxR[i]; }
        }
    }
}

#ifndef CDM_XRE_MINW
#define CDM_XRE_MINW 8
#endif
__global__ __launch_bounds__(64, CDM_XRE_MINW) void k_xr_extend(ExtArgs A, XrArgs X) {
    A.raw = nullptr;
    __shared__ double sLogLik[11 * 16];
    for (int i = threadIdx.x; i < 11 * 16; i += blockDim.x) sLogLik[i] = (&A.lut->logLik[0][0][0][0])[i];
    __syncthreads();
    const unsigned int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= *X.nWork) return;
    const uint32_t q = X.work[item];
    const SeqMeta *meta = A.len.m;
    const SeqMeta qm = meta[q];
    const uint64_t r0 = A.aoff[q];
    const uint32_t nRec = (uint32_t) (A.aoff[q + 1] - r0);
    Cand *cand = A.cand + r0;
    uint32_t *heapL = A.lists + 4 * r0, *parkL = heapL + nRec, *leftL = parkL + nRec, *rightL = leftL + nRec;
    VQuery Q; xrQuery(Q, A, q, qm); Q.cand = cand; Q.leftL = leftL; Q.rightL = rightL;
    Heap heap; heap.h = heapL; heap.n = 0; heap.cand = cand;
    uint32_t nCand = 0;
    const uint32_t maxL0 = X.qMax[2 * (size_t) q], maxR0 = X.qMax[2 * (size_t) q + 1];
    for (uint64_t w = r0 >> 5; w <= (r0 + nRec - 1) >> 5; w++) {          // the records that enter the queue, in record order
        uint32_t bits = X.pushBits[w];
        if (w == (r0 >> 5)) bits &= ~0u << (r0 & 31u);
        if (w == ((r0 + nRec) >> 5)) bits &= ~(~0u << ((r0 + nRec) & 31u));
        while (bits) {
            const uint32_t b = (uint32_t) __ffs((int) bits) - 1u;
            bits &= bits - 1u;
            const uint64_t ri = (w << 5) + b;
            const AlnRec rec = A.rec[ri]; const float2 l = X.lite[ri];
            Cand c; xrFill(c, rec, xrAlnLen(rec), meta[rec.target], qm.len, l.x, l.y, (uint32_t) (ri - r0));
            c.sLenNorm = X.sLen[ri];
            c.pieceLen = __float_as_uint(X.sBound[ri]);                  // (until the queue is built)
            cand[nCand++] = c;
        }
    }
    // The queue orders by sLenNorm.  Where two plain-double sums lie within their error bounds of each other, their order (or their
    // equality) under scoreCand's values is open: those candidates are scored again, exactly.  The query has no pieces yet, and the
    // longest overlaps are the ones the first scores were taken with, so scoreCand() returns what it would have then.
    if (nCand > 1) {
        for (uint32_t i = 0; i < nCand; i++) {
            const float bi = __uint_as_float(cand[i].pieceLen);
            if (!(bi > 0.f)) continue;
            bool open = nCand > 32;                                       // (a deep pile-up: all of them, instead of all pairs)
            const double si = cand[i].sLenNorm;
            for (uint32_t j = 0; j < nCand && !open; j++)
                if (j != i) open = fabs(si - cand[j].sLenNorm) <= (double) bi + (double) fabsf(__uint_as_float(cand[j].pieceLen));
            if (open) cand[i].pieceLen |= 0x80000000u;                   // (bounds are positive floats: the sign bit is free)
        }
        for (uint32_t i = 0; i < nCand; i++)
            if (cand[i].pieceLen & 0x80000000u) { Cand c = cand[i]; c.pieceLen = 0; (void) scoreCand(A, Q, c, maxL0, maxR0, sLogLik); cand[i].sLenNorm = c.sLenNorm; }
    }
    for (uint32_t k = 0; k < nCand; k++) { cand[k].pieceLen = 0; heap.push(k); }
    extendLoop(A, Q, cand, heap, parkL, leftL, rightL, maxL0, maxR0, sLogLik, qm.key, q);
}

__global__ void k_mark_active2(const uint64_t *__restrict__ aoff, uint32_t n, uint32_t *__restrict__ active, unsigned int *__restrict__ nActive,
                               uint32_t *__restrict__ newLen) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = q < n && (aoff[q + 1] - aoff[q] > 1);
    if (q < n) newLen[q] = 0;
    const uint32_t slot = cdm_block_append(nActive, act);
    if (act) active[slot] = q;
}
// output geometry: length, words, ext flag
__global__ void k_out_meta(const uint32_t *__restrict__ len, const uint8_t *__restrict__ ext, const uint32_t *__restrict__ newLen, uint32_t n,
                           uint32_t *__restrict__ oLen, uint8_t *__restrict__ oExt, uint32_t *__restrict__ oWords, unsigned long long *__restrict__ stats) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t L = 0;
    if (q < n) {
        L = newLen[q] ? newLen[q] : len[q];
        oLen[q] = L; oExt[q] = newLen[q] ? 1 : ext[q]; oWords[q] = (L + 15) / 16;
    }
    unsigned long long sum = L; uint32_t mx = L;
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); mx = max(mx, (uint32_t) __shfl_xor((int) mx, o, 64)); }
    __shared__ unsigned long long sSum[16]; __shared__ uint32_t sMax[16];
    if ((threadIdx.x & 63) == 0) { sSum[threadIdx.x >> 6] = sum; sMax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {   // one pair of atomics per 1024-thread block
        for (unsigned w = 1; w < (blockDim.x + 63) / 64; w++) { sum += sSum[w]; mx = max(mx, sMax[w]); }
        atomicAdd(&stats[0], sum); atomicMax(&stats[1], (unsigned long long) mx);
    }
}
// one thread per output word
// One thread per output sequence (a read is 7 code words; the words of 64 neighbouring sequences are contiguous, so the strided
// stores of a wave fill whole lines between them).  A thread per output WORD needed a 26-step owner search per wave, which was
// most of the kernel's time.
__global__ __launch_bounds__(256) void k_write(ExtArgs A, const uint32_t *__restrict__ oWoff, const uint32_t *__restrict__ oLen, uint32_t n,
                                               uint32_t *__restrict__ oCodes, uint32_t *__restrict__ oNmask, uint8_t *__restrict__ oHasN) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const uint32_t L = oLen[q], nw = (L + 15) / 16, ob = oWoff[q];
    uint16_t *oN16 = reinterpret_cast<uint16_t *>(oNmask);
    uint32_t anyN = 0;
    if (A.newLen[q] == 0) {   // copy through (:564-581)
        const uint32_t ib = A.woff[q];
        const uint16_t *iN16 = reinterpret_cast<const uint16_t *>(A.nmask);
        for (uint32_t w = 0; w < nw; w++) { const uint32_t nb = iN16[ib + w]; oCodes[ob + w] = A.codes[ib + w]; oN16[ob + w] = (uint16_t) nb; anyN |= nb; }
    } else {
        const uint64_t r0 = A.aoff[q]; const uint32_t nRec = (uint32_t) (A.aoff[q + 1] - r0);
        VQuery Q; Q.a = &A; Q.q = q; Q.qLen0 = A.len[q]; Q.qw = A.woff[q]; Q.cand = A.cand + r0;
        Q.leftL = A.lists + 4 * r0 + 2 * (uint64_t) nRec; Q.rightL = Q.leftL + nRec; Q.nL = A.nLeft[q]; Q.nR = A.nRight[q]; Q.leftTotal = A.leftTotal[q]; Q.total = L; Q.plain = false;
        for (uint32_t w = 0; w < nw; w++) {
            const uint32_t cnt = min(16u, L - w * 16u);
            uint32_t code = 0, nb = 0;
            for (uint32_t j = 0; j < cnt;) {       // piece by piece: a word rarely spans more than two
                uint32_t t, tp, run;
                Q.spanAt(w * 16 + j, t, tp, run);
                const uint32_t m = min(run, cnt - j), tw = A.woff[t];
                if (!A.hasN[t]) {
                    const uint32_t bits = cdm_window16(A.codes, tw, tp, (A.len[t] + 15) / 16 - 1);
                    code |= ((m < 16) ? (bits & ((1u << (2 * m)) - 1u)) : bits) << (2 * j);
                } else {
                    for (uint32_t i = 0; i < m; i++) {
                        uint32_t c = cdm_base(A.codes, tw, tp + i);
                        if (cdm_isN(A.nmask, tw, tp + i)) { nb |= 1u << (j + i); c = 0; }
                        code |= c << (2 * (j + i));
                    }
                }
                j += m;
            }
            oCodes[ob + w] = code; oN16[ob + w] = (uint16_t) nb; anyN |= nb;
        }
    }
    if (anyN) oHasN[q] = 1;
}

// DBs with letters beyond ACGTN: an output sequence any of whose sources (the query, the donating targets) has a row of original
// letters gets one too - every letter is copied from its source as it stands (:425-470 append the target's own bytes)
__global__ __launch_bounds__(256) void k_write_raw(ExtArgs A, const uint32_t *__restrict__ oWoff, const uint32_t *__restrict__ oLen, uint32_t n,
                                                   uint8_t *__restrict__ oRaw, uint8_t *__restrict__ oHasN) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const uint32_t L = oLen[q];
    uint8_t *o = oRaw + (uint64_t) oWoff[q] * 16u;
    VQuery Q; Q.a = &A; Q.q = q; Q.qLen0 = A.len[q]; Q.qw = A.woff[q]; Q.nL = 0; Q.nR = 0; Q.leftTotal = 0; Q.total = L; Q.plain = false;
    Q.cand = nullptr; Q.leftL = Q.rightL = nullptr;
    if (A.newLen[q] != 0) {
        const uint64_t r0 = A.aoff[q]; const uint32_t nRec = (uint32_t) (A.aoff[q + 1] - r0);
        Q.cand = A.cand + r0; Q.leftL = A.lists + 4 * r0 + 2 * (uint64_t) nRec; Q.rightL = Q.leftL + nRec; Q.nL = A.nLeft[q]; Q.nR = A.nRight[q]; Q.leftTotal = A.leftTotal[q];
    }
    bool any = false;
    for (uint32_t p = 0; p < L;) { uint32_t t, tp, run; Q.spanAt(p, t, tp, run); any = any || A.hasRaw[t]; p += run; }
    if (!any) return;
    for (uint32_t p = 0; p < L;) {
        uint32_t t, tp, run;
        Q.spanAt(p, t, tp, run);
        const uint32_t tw = A.woff[t], m = min(run, L - p);
        if (A.hasRaw[t]) for (uint32_t i = 0; i < m; i++) o[p + i] = cdm_raw_at(A.raw, tw, tp + i);
        else for (uint32_t i = 0; i < m; i++) o[p + i] = (A.hasN[t] && cdm_isN(A.nmask, tw, tp + i)) ? 'N' : (uint8_t) "ACGT"[cdm_base(A.codes, tw, tp + i)];
        p += m;
    }
    oHasN[q] = 3;
}

}  // namespace

int cdm_extend_impl(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, cdm_seqdb **out, double *scores) {
    hipStream_t s = ctx->stream;
    const uint32_t n = (uint32_t) db->n;
    if (alns->n != db->n) { cdm_set_error("cdm_extend: alignment CSR / DB size mismatch"); return CDM_ERR_INVALID; }
    DevBuf<uint32_t> active, lists, newLen, nLeft, nRight, leftTotal, oWords;
    DevBuf<unsigned int> nActive, flags; DevBuf<Cand> cand; DevBuf<double> dScores; DevBuf<unsigned long long> stats;
    if (!active.alloc(n) || !lists.alloc(4 * alns->count) || !newLen.alloc(n) || !nLeft.alloc(n) || !nRight.alloc(n) || !leftTotal.alloc(n) ||
        !oWords.alloc(n) || !nActive.alloc(2) || !flags.alloc(4) || !cand.alloc(alns->count) || !stats.alloc(2) || (scores && !dScores.alloc(alns->count))) {
        cdm_set_error("cdm_extend: out of device memory"); return CDM_ERR_HIP;
    }
    hipMemsetAsync(nActive.p, 0, 8, s);
    hipMemsetAsync(flags.p, 0, 16, s);
    hipMemsetAsync(stats.p, 0, 16, s);
    if (scores) hipMemsetAsync(dScores.p, 0xFF, alns->count * 8, s);   // all-ones = NaN: records of inactive queries
    hipLaunchKernelGGL(k_mark_active2, dim3((n + 1023) / 1024), dim3(1024), 0, s, alns->off, n, active.p, nActive.p, newLen.p);
    unsigned int hAct = 0;
    hipMemcpyAsync(&hAct, nActive.p, 4, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_extend: setup failed"); return CDM_ERR_HIP; }
    DevBuf<SeqMeta> meta;
    MetaUniform uni;
    if (int rc = cdm_build_meta(ctx, db, &meta.p, &uni)) return rc;
    ExtArgs A;
    cdmSetMeta(A, meta.p, uni); A.ext.m = A.key.m = meta.p; A.ext.plain = uni.words; A.codes = db->codes; A.nmask = db->nmask; A.raw = db->raw;
    A.aoff = alns->off; A.rec = alns->rec; A.active = active.p; A.nActive = nActive.p; A.lut = ctx->lutDev; A.cand = cand.p; A.lists = lists.p;
    A.newLen = newLen.p; A.nLeft = nLeft.p; A.nRight = nRight.p; A.leftTotal = leftTotal.p; A.scores = scores ? dScores.p : nullptr;
    A.seqIdThr = par->seq_id_thr; A.rySeqIdThr = par->ry_seq_id_thr; A.likelihoodThr = par->likelihood_threshold;
    A.excessLog = std::log(par->excess_penal); A.randLog = std::log(par->rand_align_penal);   // std::log(float): float, as in the reference
    A.ratioLogit = (double) logl(1.0L / (long double) par->likelihood_threshold - 1.0L);
    {
        // |d sRatio / dx| = s (1 - s): a relative error eps of sRatio (its rounding to double, expl's last bits) moves the decision over eps
        // / (1 - s) resp. eps / s in x; ratioLogit itself is rounded once.  64 ulps of all that, where 3 would do.
        const double thr = (double) par->likelihood_threshold;
        A.ratioWindow = (par->likelihood_threshold == 0.5f || !(thr > 0.0 && thr < 1.0)) ? 0.0 : 64.0 * 0x1p-53 * (1.0 + fabs(A.ratioLogit) + 1.0 / std::min(thr, 1.0 - thr));
    }
    A.marginScale = cdmGetenv("CDM_EXTEND_MARGIN") ? (float) atof(cdmGetenv("CDM_EXTEND_MARGIN")) : 1.0f;
    A.maxSeqLen = par->max_seq_len;
    A.unsafe = par->unsafe ? 1 : 0; A.minCov = (uint32_t) std::max(0, par->min_cov_safe); A.flags = flags.p;
    hipEventRecord(ctx->ev0, s);
    const char *padEnv = cdmGetenv("CDM_LDS_PAD");          // experiments: dynamic LDS that lowers the occupancy
    const char *wEnv = cdmGetenv("CDM_EXTEND_WAVES");        // experiments: waves per SIMD the register allocation leaves room for
    const char *formEnv = cdmGetenv("CDM_EXTEND");           // "queries": k_extend (one lane per query) for every call
    const int minW = wEnv ? atoi(wEnv) : 8;
    const unsigned padB = padEnv ? (unsigned) atoi(padEnv) : 0u;
    // one thread per record for A-D (k_xr_*) where that form applies: the default mode, no raw plane
    bool perRecord = hAct && !db->raw && !A.unsafe && alns->count < 0xFFFFFFFFull / 2 && !(formEnv && !strcmp(formEnv, "queries"));
    DevBuf<uint32_t> winQ, qMax, work, elig, pushBits; DevBuf<float2> lite; DevBuf<double> sLen; DevBuf<float> sBound;
    if (perRecord) {
        const uint32_t nWin = (uint32_t) ((alns->count + XR_T - 1) / XR_T);
        if (!winQ.alloc((size_t) nWin + 1) || !qMax.alloc(2 * (size_t) n) || !work.alloc(hAct) || !lite.alloc(alns->count) || !sLen.alloc(alns->count) ||
            !elig.alloc(alns->count) || !pushBits.alloc(alns->count / 32 + 2) || !sBound.alloc(alns->count)) {
            cdm_set_error("cdm_extend: out of device memory"); return CDM_ERR_HIP;
        }
        hipMemsetAsync(pushBits.p, 0, (alns->count / 32 + 2) * 4, s);
        XrArgs X; X.sBound = sBound.p; X.elig = elig.p; X.pushBits = pushBits.p; X.winQ = winQ.p; X.lite = lite.p; X.sLen = sLen.p; X.qMax = qMax.p; X.work = work.p; X.nWork = nActive.p + 1; X.fallback = flags.p + 1;
        hipLaunchKernelGGL(k_xr_windows, dim3(nWin / 256 + 1), dim3(256), 0, s, alns->off, n, nWin, winQ.p);
        hipLaunchKernelGGL(k_xr_score, CDM_GRID(nWin, XR_NT), dim3(XR_NT), 0, s, A, X);
        unsigned int fb = 0;
        hipMemcpyAsync(&fb, flags.p + 1, 4, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_extend: scoring kernel failed"); return CDM_ERR_HIP; }
        if (fb) perRecord = false;          // a candidate that overlaps on both sides at once: the maxima depend on the record order
        else hipLaunchKernelGGL(k_xr_extend, dim3((hAct + 63) / 64), dim3(64), padB, s, A, X);
    }
    if (perRecord) {}
    else if (hAct && db->raw) hipLaunchKernelGGL((k_extend<8, true>), dim3((hAct + 63) / 64), dim3(64), padB, s, A);
    else if (hAct && minW == 8) hipLaunchKernelGGL(k_extend<8>, dim3((hAct + 63) / 64), dim3(64), padB, s, A);
    else if (hAct && minW == 6) hipLaunchKernelGGL(k_extend<6>, dim3((hAct + 63) / 64), dim3(64), padB, s, A);
    else if (hAct) hipLaunchKernelGGL(k_extend<5>, dim3((hAct + 63) / 64), dim3(64), padB, s, A);
    hipEventRecord(ctx->ev1, s);
    // ---- output DB
    cdm_seqdb *o = nullptr;
    int rc = cdm_seqdb_alloc(ctx, n, &o);
    if (rc) return rc;
    hipLaunchKernelGGL(k_out_meta, dim3((n + 1023) / 1024), dim3(1024), 0, s, db->len, db->ext, newLen.p, n, o->len, o->ext, oWords.p, stats.p);
    cdmscan::ScanTemp tmp;
    hipMemsetAsync(oWords.p + n, 0, 4, s);
    if (cdmscan::exclusiveScan<uint32_t>(s, tmp, oWords.p, o->woff, (size_t) n + 1) != CDM_OK) { cdm_seqdb_free(o); return CDM_ERR_HIP; }
    uint32_t words = 0; unsigned long long hstats[2] = {0, 0}; unsigned int hflags[4] = {0, 0, 0, 0};
    hipMemcpyAsync(hflags, flags.p, 16, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(&words, o->woff + n, 4, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(hstats, stats.p, 16, hipMemcpyDeviceToHost, s);
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_seqdb_free(o); cdm_set_error("cdm_extend: extension kernel failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    hipEventElapsedTime(&ctx->lastMs[4], ctx->ev0, ctx->ev1);
    if (hflags[0]) {
        cdm_seqdb_free(o);
        cdm_set_error("cdm_extend: --unsafe 1: a target overhangs the query by more than the query's length; the reference pads it with a negative number of letters there (undefined behaviour), not reproduced");
        return CDM_ERR_UNSUPPORTED;
    }
    if (hflags[2]) {
        cdm_seqdb_free(o);
        cdm_set_error("cdm_extend: --likelihood-ratio-threshold %g: a candidate's likelihood ratio lies within %.3g of the threshold in log-odds; the reference decides there by the last bit of the C library's expl - not reproduced (the default threshold 0.5 has no such window)", (double) par->likelihood_threshold, A.ratioWindow);
        return CDM_ERR_UNSUPPORTED;
    }
    if (hstats[0] / 16 + n >= 0xFFFFFFF0ull) {     // (the word offsets are 32-bit: an extended DB beyond 2^32 code words would wrap)
        cdm_seqdb_free(o); cdm_set_error("cdm_extend: the extended sequences need more than 2^32 code words (68 G bases) in one DB"); return CDM_ERR_UNSUPPORTED;
    }
    o->words = words; o->residues = hstats[0]; o->maxLen = (uint32_t) hstats[1];
    const uint64_t maskWords = ((uint64_t) words * 16 + 31) / 32 + 1;
    if (cdmMalloc(&o->codes, ((size_t) words + 2) * 4) != hipSuccess || cdmMalloc(&o->nmask, maskWords * 4) != hipSuccess) { cdm_seqdb_free(o); cdm_set_error("cdm_extend: out of device memory"); return CDM_ERR_HIP; }
    hipMemcpyAsync(o->key, db->key, (size_t) n * 4, hipMemcpyDeviceToDevice, s);
    hipMemsetAsync(o->hasN, 0, n, s);
    if (words) hipLaunchKernelGGL(k_write, dim3((n + 255) / 256), dim3(256), 0, s, A, o->woff, o->len, n, o->codes, o->nmask, o->hasN);
    if (words && db->raw) {
        if (int rc2 = cdm_seqdb_alloc_raw(o)) { cdm_seqdb_free(o); return rc2; }
        hipLaunchKernelGGL(k_write_raw, dim3((n + 255) / 256), dim3(256), 0, s, A, o->woff, o->len, n, o->raw, o->hasN);
    }
    if (scores) hipMemcpyAsync(scores, dScores.p, alns->count * 8, hipMemcpyDeviceToHost, s);
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_seqdb_free(o); cdm_set_error("cdm_extend: output kernel failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    *out = o;
    return CDM_OK;
}
