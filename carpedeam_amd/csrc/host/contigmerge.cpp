// Host part of ancient_contig_merge (src/assembler/ancientContigsResults.cpp:94-509): per query, from the per-record counts the
// device took (contig.hip), the contig filter, the consensus-based identities (updateSeqIdConsensus, safe mode), the damage-aware
// match count (ancientMatchCount / deamMatches, nuclassembleUtil.cpp:1009-1181), the priority queue ordered by the
// Beta-posterior comparator (:25-70) and the extension loop with the re-alignment of parked hits (:276-470).
// This file is compiled with g++ at the reference recipe's flags (carpedeam_amd/build.py): the comparator's lgammaf / logf /
// exp come from the same C library, and the queue is the same libstdc++ std::priority_queue - the comparator is not a strict
// weak ordering, so the order it produces is defined by that implementation and nothing else.
#include <algorithm>
#include <array>
#include <climits>
#include <cmath>
#include <cstring>
#include <immintrin.h>
#include <omp.h>
#include <queue>
#include <memory>
#include <string>
#include <vector>

#include "../common.h"

namespace {
// lgammaf's value without its side effect: std::lgamma(float) also stores the sign in the C library's global `signgam`, one cache
// line that every comparator call of every thread then writes - the host merge spent most of its time passing that line around
// (16 threads did the work of one in a quarter of its time).  lgammaf_r is the same function with the sign in a local.
inline float lgammaQuiet(float x) { int sign; return lgammaf_r(x, &sign); }
struct Res {            // Matcher::result_t, the fields this module uses (M/alignment/Matcher.h:33-56)
    uint32_t target = 0, dbKey = 0;
    float seqId = 0, rySeqId = 0, deamMatch = 0;
    unsigned alnLength = 0, alnLengthCons = 0;
    int qStartPos = 0, qEndPos = 0, dbStartPos = 0, dbEndPos = 0;
    unsigned qLen = 0, dbLen = 0;
    bool isRev = false;
    // the comparator's two lgamma terms that depend on this record alone (same float arithmetic as there), set by cacheTerms()
    float lgBeta = 0, lgAlphaBeta = 0;
    void cacheTerms() { const float mm = alnLengthCons - deamMatch, alpha = mm + 1, beta = deamMatch + 1; lgBeta = lgammaQuiet(beta); lgAlphaBeta = lgammaQuiet(alpha + beta); }
};
// The queue's order (ancientContigsResults.cpp:25-70): "is x a worse overlap than y?".  Every record stands for a Beta posterior over
// its mismatch rate - mismatches + 1 and damage-aware matches + 1 as the two shape parameters - and x counts as worse when the
// probability that x's rate lies below y's, a finite series in the shape parameters, is under 0.45; between 0.45 and 0.55 the
// shorter consensus overlap loses and equal ones count as worse too - which is why this is no strict weak ordering and the order
// it produces belongs to libstdc++'s heap.  The arithmetic keeps the reference's types term by term, because its last bits decide:
// shapes and their sums in float (`using namespace std` makes lgamma / log of a float the float functions), the series in double,
// the summation index a size_t that is ADDED to a float before the float log is taken.
thread_local unsigned long long tlCompares = 0, tlSeriesTerms = 0;     // CDM_TIMING statistics
struct CompareByScoreContigs {
    static float mismatches(const Res &r) { return r.alnLengthCons - r.deamMatch; }
    bool operator()(const Res &x, const Res &y) const {
        tlCompares++;
        const float xMis = mismatches(x) + 1, yMis = mismatches(y) + 1;         // first shape parameter of either posterior
        const float xHit = x.deamMatch + 1, yHit = y.deamMatch + 1;             // second
        // log B(xMis + 0, xHit + yHit) - log B(xMis, xHit) written as four lgamma values; the two that depend on x alone are cached
        const double logScale = (lgammaQuiet(xHit + yHit) + x.lgAlphaBeta) - (lgammaQuiet(xMis + xHit + yHit) + x.lgBeta);
        double logTerm = 0.0, below = 0.0;
        for (size_t k = 0; k < yMis; k++) {
            tlSeriesTerms++;
            below += std::exp(logTerm + logScale);
            logTerm = std::log(xMis + k) + std::log(yHit + k) - (std::log(k + 1) + std::log(k + xMis + xHit + yHit)) + logTerm;
        }
        if (below < 0.45) return true;
        if (below > 0.55) return false;
        return !(x.alnLengthCons > y.alnLengthCons);
    }
};
typedef std::priority_queue<Res, std::vector<Res>, CompareByScoreContigs> Queue;

inline int ryClass(char c) { return (c == 'C' || c == 'T') ? 1 : 0; }              // ryMap[c]; any other letter: 0
// getNuclRevFragment (nuclassembleUtil.cpp:67-76): the complement of what NucleotideMatrix::setupLetterMapping
// (M/commons/NucleotideMatrix.cpp:17-61) maps the letter to - lower case folds, IUPAC codes stand for one base, the rest is X -> 'N'
struct RevTable {
    char t[256];
    RevTable() {
        for (int c = 0; c < 256; c++) {
            char r = 'N';
            if ((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))
                switch (c & ~0x20) {
                    case 'A': r = 'T'; break;
                    case 'C': case 'M': case 'Y': case 'H': r = 'G'; break;
                    case 'G': case 'K': case 'B': case 'D': case 'V': case 'R': case 'S': r = 'C'; break;
                    case 'T': case 'U': case 'W': r = 'A'; break;
                    default: break;
                }
            t[c] = r;
        }
    }
};
static const RevTable REV_TABLE;
inline char revLetter(char c) { return REV_TABLE.t[(unsigned char) c]; }
std::string revComp(const char *s, size_t n) {
    const RevTable &tab = REV_TABLE;
    std::string r(n, 'N');
    for (size_t i = 0; i < n; i++) r[i] = tab.t[(unsigned char) s[n - 1 - i]];
    return r;
}
// Identical letters over [from, from + cols) and letters of the same RY class over [from, from + cols] of the query stretch qa against
// the overlap's target letters (updateNuclAlignment :28-31, getRYSeqId :78-92 on a re-aligned parked hit).  The target letters are
// fwd[j] (the target as stored) or, for a reversed target, revLetter(back[-j]) - spelled out on the fly, 32 letters at a time where
// they are all upper-case ACGT (contigs are, but for the odd N), letter by letter elsewhere.  The parked hits of the long contigs of
// the late iterations walk 1e10 columns; this loop was three passes over a copy of them.
void overlapCounts(const char *qa, const char *fwd, const char *back, size_t from, size_t cols, int &idCnt, int &idRy) {
    size_t i = from;
    const size_t endId = from + cols, endRy = from + cols + 1;
    long id = 0, ry = 0;
#ifdef __AVX2__
    const __m256i cC = _mm256_set1_epi8('C'), cT = _mm256_set1_epi8('T'), cA = _mm256_set1_epi8('A'), cG = _mm256_set1_epi8('G');
    const __m256i flip = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    // complement by the low nibble of A (1), C (3), T (4), G (7)
    const __m256i comp = _mm256_setr_epi8(0, 'T', 0, 'G', 'A', 0, 0, 'C', 0, 0, 0, 0, 0, 0, 0, 0, 0, 'T', 0, 'G', 'A', 0, 0, 'C', 0, 0, 0, 0, 0, 0, 0, 0);
    for (; i + 32 <= endId; i += 32) {
        const __m256i q = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(qa + i));
        __m256i t;
        if (fwd) t = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(fwd + i));
        else {
            // letters back[-i-31] .. back[-i], to be read backwards
            __m256i raw = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(back - i - 31));
            const __m256i acgt = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(raw, cA), _mm256_cmpeq_epi8(raw, cC)), _mm256_or_si256(_mm256_cmpeq_epi8(raw, cG), _mm256_cmpeq_epi8(raw, cT)));
            if (_mm256_movemask_epi8(acgt) != -1) {     // a letter beyond ACGT in this block: the table, letter by letter
                for (size_t j = i; j < i + 32; j++) { const char tc = revLetter(*(back - j)); id += (qa[j] == tc); ry += (ryClass(qa[j]) == ryClass(tc)); }
                continue;
            }
            raw = _mm256_permute2x128_si256(raw, raw, 1);                       // swap the halves, then reverse inside each
            raw = _mm256_shuffle_epi8(raw, flip);
            t = _mm256_shuffle_epi8(comp, _mm256_and_si256(raw, _mm256_set1_epi8(0x0F)));
        }
        id += __builtin_popcount((unsigned) _mm256_movemask_epi8(_mm256_cmpeq_epi8(q, t)));
        const __m256i qy = _mm256_or_si256(_mm256_cmpeq_epi8(q, cC), _mm256_cmpeq_epi8(q, cT)), ty = _mm256_or_si256(_mm256_cmpeq_epi8(t, cC), _mm256_cmpeq_epi8(t, cT));
        ry += __builtin_popcount((unsigned) _mm256_movemask_epi8(_mm256_cmpeq_epi8(qy, ty)));
    }
#endif
    for (; i < endRy; i++) {
        const char tc = fwd ? fwd[i] : revLetter(*(back - i));
        if (i < endId) id += (qa[i] == tc);
        ry += (ryClass(qa[i]) == ryClass(tc));
    }
    idCnt = (int) id; idRy = (int) ry;
}
// What one C->T / G->A column is worth as a match (deamMatches, nuclassembleUtil.cpp:1009-1044): the posterior odds that the column
// is damage rather than a true mismatch.  The prior for "the two sequences match here" mixes the overlap's own match rate with a
// length prior that falls off as 1.4e-9 / length^3 between 10 and 1e5 letters (in logs); the likelihood of the observed letter
// under damage comes from the damage matrix.  Float literals and double variables alternate exactly as in the reference.
// In two parts since round 5: the length prior is the only place a logarithm enters - the C library's, whose last bits the device
// cannot restate, so the device queue (contigqueue.hip) reads it from a table this very function fills (cdm_contig_host_tables) - and
// the rest is IEEE arithmetic the device repeats operation by operation.  The one contraction g++ makes at this file's flags that is
// not exact either way (1 + ratio * odds, a fused multiply-add) is spelled out, so that both sides say the same thing.
inline double lengthPriorOf(unsigned overlap) {
    const double logConst = std::log(1.4e-9);
    const unsigned longest = 1e5;
    auto logPrior = [logConst](unsigned len) { return logConst - 3.0 * std::log(len); };
    const double atShortest = logPrior(10), atLongest = logPrior(longest), here = logPrior(std::min(overlap, longest));
    const double shareOfRange = (static_cast<double>(std::abs(here) - std::abs(atLongest))) / static_cast<double>((std::abs(atShortest) - std::abs(atLongest)));
    return 1 - shareOfRange;
}
inline double deamFromPrior(unsigned overlap, unsigned score, double damageLik, double lengthPrior) {
    const double pMatch = 0.5f * ((((static_cast<double>(score) + 3.0f * overlap) / 5.0f) + 0.9f) / (overlap + 1)) + 0.5f * lengthPrior;
    const double pMismatch = 1 - pMatch;
    const double likelihoodRatio = pMismatch / damageLik;
    const double priorOdds = (1 - pMatch) / pMatch;
    return 1 / std::fma(likelihoodRatio, priorOdds, 1.0);
}
double deamMatches(unsigned overlap, unsigned score, double damageLik) { return deamFromPrior(overlap, score, damageLik, lengthPriorOf(overlap)); }
// selectNuclFragmentToExtendContigs (:73-91)
bool selectFragment(Queue &q, uint32_t queryKey, Res &out) {
    while (!q.empty()) {
        Res r = q.top(); q.pop();
        const bool notBoth = !(r.dbStartPos == 0 && r.qStartPos == 0);
        const bool rightStart = r.dbStartPos == 0 && (r.dbEndPos != static_cast<int>(r.dbLen) - 1);
        const bool leftStart = r.qStartPos == 0 && (r.qEndPos != static_cast<int>(r.qLen) - 1);
        if ((rightStart || leftStart) && notBoth && r.dbKey != queryKey) { out = r; return true; }
    }
    return false;
}
inline int nucMap(char c) { return c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0; }      // nucleotideMap[c]; any other letter: 0
// --unsafe 1 (not the workflow's default; the device statistics are the safe mode's, so this mode works on the host strings):
// consensusCaller's majority vote over the candidates that extend the query (nuclassembleUtil.cpp:570-702, calculateConsensus
// :535-567) - 3 qLen letters, the query in the middle third, elsewhere the majority letter where at least minCov candidates cover
// the position ('N' below that and on ties)
std::string unsafeConsensus(const std::vector<Res> &cands, const std::vector<SeqView> &seqs, const SeqView &q, uint32_t queryKey, unsigned minCov, bool &undefinedCase) {
    const unsigned qLen = (unsigned) q.size();
    std::vector<std::array<unsigned, 4>> cov(3 * (size_t) qLen, std::array<unsigned, 4>{{0, 0, 0, 0}});
    for (const Res &c : cands) {
        const bool rightStart = c.dbStartPos == 0 && (c.dbEndPos != static_cast<int>(c.dbLen) - 1);
        const bool leftStart = c.qStartPos == 0 && (c.qEndPos != static_cast<int>(c.qLen) - 1);
        if (!(rightStart || leftStart) || c.dbKey == queryKey) continue;
        const SeqView &t0 = seqs[c.target];
        const unsigned tLen = (unsigned) t0.size();
        long start;
        if ((unsigned) c.dbStartPos == 0 && (unsigned) c.qEndPos == (qLen - 1)) start = (long) qLen + c.qStartPos;
        else if ((unsigned) c.qStartPos == 0 && (unsigned) c.dbEndPos == (tLen - 1)) start = (long) qLen - (long) (c.dbLen - c.alnLength);
        else continue;
        if (start < 0) { undefinedCase = true; continue; }          // the reference indexes its coverage vector with a negative number there
        const std::string t = c.isRev ? revComp(t0.data(), tLen) : std::string();
        const char *ts = c.isRev ? t.data() : t0.data();
        // (a target that reaches beyond the 3 qLen letters makes the reference write behind its vector; those letters are never read back)
        for (unsigned p = 0; p < c.dbLen && (size_t) start + p < cov.size(); p++) cov[(size_t) start + p][nucMap(ts[p])] += 1;
    }
    std::string cons(3 * (size_t) qLen, 'N');
    for (size_t i = 0; i < cov.size(); i++) {
        const unsigned tot = cov[i][0] + cov[i][1] + cov[i][2] + cov[i][3];
        if (tot < minCov) continue;
        unsigned mx = 0; char nuc = 'N'; int nMax = 0;
        for (int j = 0; j < 4; j++) { if (cov[i][j] > mx) { mx = cov[i][j]; nuc = "ACGT"[j]; nMax = 1; } else if (cov[i][j] == mx && mx > 0) nMax++; }
        cons[i] = nMax > 1 ? 'N' : nuc;
    }
    for (unsigned p = 0; p < qLen; p++) cons[qLen + p] = q[p];
    return cons;
}
// the columns updateSeqIdConsensus (:705-790) and ancientMatchCount (:1047-1181) walk in that mode: the padded target against the
// whole consensus, flanks included -> defined columns, identical, same RY class, consensus C over target T, consensus G over target A
void unsafeColumns(const Res &c, const std::string &cons, const SeqView &t0, unsigned qLen, bool leftStart, int &tot, int &idc, int &idr, int &nCT, int &nGA) {
    const unsigned tLen = (unsigned) t0.size();
    const std::string t = c.isRev ? revComp(t0.data(), tLen) : std::string();
    const char *ts = c.isRev ? t.data() : t0.data();
    const unsigned offset = c.dbLen - c.alnLength;
    // leftStart: N^(qLen - offset) target against cons[0..); rightStart: target N^(qLen - offset) against the end of cons
    const size_t c0 = leftStart ? (size_t) (qLen - offset) : cons.size() - ((size_t) tLen + (qLen - offset));
    tot = idc = idr = nCT = nGA = 0;
    for (unsigned j = 0; j < tLen && c0 + j < cons.size(); j++) {
        const char cq = cons[c0 + j], ct = ts[j];
        if (cq == 'N' || ct == 'N') continue;
        tot++; idc += (cq == ct); idr += (ryClass(cq) == ryClass(ct));
        const int qb = nucMap(cq), tb = nucMap(ct);
        nCT += (qb == 1 && tb == 3); nGA += (qb == 2 && tb == 0);
    }
}
}  // namespace

// Threads of the host loops of the library.  A caller that set OMP_NUM_THREADS (or the module binary's --threads, which sets it) is
// taken at its word; without it a GPU box would hand every loop all of the machine's hardware threads - hundreds, of which one GPU's
// job owns a share (the contig phase of `bench.py --config 5` took 36 s that way and 23 s on 16 threads) - so the default is capped.
static int cdm_host_threads() {
    static const int n = [] {
        if (const char *e = cdmGetenv("CDM_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return v; }
        const int m = std::max(1, omp_get_max_threads());
        return cdmGetenv("OMP_NUM_THREADS") ? m : std::min(m, 16);
    }();
    return n;
}
void cdm_host_split(const char *blob, const std::vector<uint64_t> &offs, const std::vector<uint32_t> &lens, std::vector<SeqView> &seqs) {
    const long n = (long) offs.size();
    seqs.resize(n);
#pragma omp parallel for schedule(dynamic, 512) num_threads(cdm_host_threads())
    for (long i = 0; i < n; i++) { seqs[i].p = blob + offs[i]; seqs[i].n = lens[i]; }
}
// the strings back to back in one blob (written by all threads, first touch included)
bool cdm_host_pack(const std::vector<std::string> &seqs, HostBuf<char> &data, std::vector<uint64_t> &off, std::vector<uint32_t> &len) {
    const long m = (long) seqs.size();
    off.resize(m); len.resize(m);
    uint64_t total = 0;
    for (long j = 0; j < m; j++) { off[j] = total; len[j] = (uint32_t) seqs[j].size(); total += seqs[j].size(); }
    if (!data.alloc(total + 1)) return false;
#pragma omp parallel for schedule(dynamic, 256) num_threads(cdm_host_threads())
    for (long j = 0; j < m; j++) memcpy(data.data() + off[j], seqs[j].data(), seqs[j].size());
    return true;
}
// The C library's functions the comparator's last bits hang on, as tables for the device queue (contigqueue.hip): lgammaf over every
// float in [1, 2^lgTop), logf over every float in [1, 2^lfTop) (entry i: the float whose bits are those of 1.0f plus i), the double
// log of the integers below nInt, the length prior of deamMatches for overlaps 0 .. 100 000.  A table IS the function: exact by
// construction, whatever the library's version.  About a second on 16 threads for 19 + 20 binades.
void cdm_contig_host_tables(float *lgam, int lgTop, float *lf, int lfTop, double *logInt, size_t nInt, double *lenPrior) {
    const long nLg = (long) lgTop << 23, nLf = (long) lfTop << 23;
#pragma omp parallel num_threads(cdm_host_threads())
    {
#pragma omp for schedule(static) nowait
        for (long i = 0; i < nLg; i++) { const uint32_t u = 0x3F800000u + (uint32_t) i; float x; memcpy(&x, &u, 4); lgam[i] = lgammaQuiet(x); }
#pragma omp for schedule(static) nowait
        for (long i = 0; i < nLf; i++) { const uint32_t u = 0x3F800000u + (uint32_t) i; float x; memcpy(&x, &u, 4); lf[i] = std::log(x); }
#pragma omp for schedule(static) nowait
        for (long i = 0; i < (long) nInt; i++) logInt[i] = std::log((size_t) i);
#pragma omp for schedule(static)
        for (long i = 0; i <= 100000; i++) lenPrior[i] = lengthPriorOf((unsigned) i);
    }
}
// grownIdx (ascending) / grownSeqs: the queries that were extended and what they became; outExt: the wasExtended flag of every sequence.
// only: NULL, or a byte per query - the queries with a zero are left alone (no entry in grownIdx, outExt = 0): the few queries the
// device queue hands back (contigqueue.hip)
int cdm_contig_merge_host(const std::vector<SeqView> &seqs, const std::vector<uint32_t> &keys, const std::vector<uint8_t> &ext, const std::vector<uint64_t> &aoff,
                          const cdm_aln *recs, const ContigStat *stats, const long double mats[2][11][4][4], const cdm_ancient_params *par,
                          float mergeSeqIdThr, std::vector<uint32_t> &grownIdx, std::vector<std::string> &grownSeqs, std::vector<uint8_t> &outExt, std::string *err,
                          const uint8_t *only) {
    const size_t n = seqs.size();
    grownIdx.clear(); grownSeqs.clear(); outExt.assign(n, 0);
    std::vector<std::vector<std::pair<uint32_t, std::string>>> perThread((size_t) cdm_host_threads());
    const float ryThr = par->ry_seq_id_thr;
    bool undefinedCase = false;
    const bool timing = cdmGetenv("CDM_TIMING") != nullptr;       // per-thread seconds in: candidate gate, queue pops, string growth, parked hits
    double tSum[4] = {0, 0, 0, 0}, tMax[4] = {0, 0, 0, 0}; unsigned long long nCmp = 0, nTerms = 0;
#pragma omp parallel num_threads(cdm_host_threads())
    {
        double tl[4] = {0, 0, 0, 0}; double tm = timing ? omp_get_wtime() : 0;
        auto lap = [&](int k) { if (timing) { const double n2 = omp_get_wtime(); tl[k] += n2 - tm; tm = n2; } };
        std::vector<Res> contigs, parked;
        std::string revBuf;
        std::vector<std::pair<uint32_t, std::string>> &mine = perThread[(size_t) omp_get_thread_num()];
#pragma omp for schedule(dynamic, 100)
        for (size_t id = 0; id < n; id++) {
            if (only && !only[id]) continue;
            const uint32_t queryKey = keys[id];
            // useReverse[target] of the reference (:136-137,198,212): a per-thread array every record of the query writes, read back
            // for the targets in the queue - i.e. the orientation of the query's LAST record with that target
            auto useReverse = [&](uint32_t target) -> bool {
                for (uint64_t r = aoff[id + 1]; r-- > aoff[id];) if (recs[r].target == target) return stats[r].rev != 0;
                return false;
            };
            const SeqView &q0 = seqs[id];
            unsigned qLen = (unsigned) q0.size();
            std::string query;                              // working copy, made once a candidate exists
            contigs.clear();
            Queue queue;
            // :187-235 orientation, identities (from the device), contig filter
            const bool unsafeMode = par->unsafe != 0;
            std::string cons;                               // --unsafe 1: the majority-vote consensus of this query's candidates
            if (unsafeMode) {
                contigs.clear();
                for (uint64_t r = aoff[id]; r < aoff[id + 1]; r++) {
                    const cdm_aln &a = recs[r]; const ContigStat &st = stats[r];
                    Res x; x.target = a.target; x.dbKey = st.dbKey; x.qLen = qLen; x.dbLen = st.dbLen;
                    x.alnLength = (unsigned) std::max(std::abs(a.q_end - a.q_start), std::abs(a.db_end - a.db_start)) + 1u;
                    x.qStartPos = st.qs; x.qEndPos = st.qe; x.dbStartPos = st.ds; x.dbEndPos = st.de; x.isRev = st.rev != 0;
                    x.seqId = static_cast<float>(st.idCnt) / x.alnLength; x.rySeqId = static_cast<float>(st.idRy) / x.alnLength;
                    if (x.seqId >= mergeSeqIdThr && x.rySeqId >= ryThr && queryKey != x.dbKey) contigs.push_back(x);
                }
                bool undef = false;
                if (!contigs.empty()) cons = unsafeConsensus(contigs, seqs, q0, queryKey, (unsigned) std::max(0, par->min_cov_safe), undef);
                if (undef) {
#pragma omp atomic write
                    undefinedCase = true;
                }
            }
            for (uint64_t r = aoff[id]; r < aoff[id + 1]; r++) {
                const cdm_aln &a = recs[r]; const ContigStat &st = stats[r];
                Res x; x.target = a.target; x.dbKey = st.dbKey;
                x.qLen = qLen; x.dbLen = st.dbLen;
                x.alnLength = (unsigned) std::max(std::abs(a.q_end - a.q_start), std::abs(a.db_end - a.db_start)) + 1u;     // Matcher::computeAlnLength
                x.qStartPos = st.qs; x.qEndPos = st.qe; x.dbStartPos = st.ds; x.dbEndPos = st.de; x.isRev = st.rev != 0;
                x.seqId = static_cast<float>(st.idCnt) / x.alnLength;
                x.rySeqId = static_cast<float>(st.idRy) / x.alnLength;
                if (x.seqId >= mergeSeqIdThr && x.rySeqId >= ryThr && queryKey != x.dbKey) {
                    // :243 updateSeqIdConsensus against N^L query N^L (safe mode): the columns where both letters are defined
                    const bool rightStart = (unsigned) x.dbStartPos == 0 && (unsigned) x.qEndPos == (qLen - 1);
                    const bool leftStart = (unsigned) x.qStartPos == 0 && (unsigned) x.dbEndPos == (x.dbLen - 1);
                    int tot = 0, idc = 0, idr = 0, nCT = st.nCT, nGA = st.nGA;
                    if (leftStart || rightStart) {
                        if (x.dbLen - x.alnLength > qLen) {                         // the reference pads with qLen - offset letters
#pragma omp atomic write
                            undefinedCase = true;
                        }
                        else if (unsafeMode) unsafeColumns(x, cons, seqs[x.target], qLen, leftStart, tot, idc, idr, nCT, nGA);
                        else { tot = st.nnTot; idc = st.nnId; idr = st.nnRy; }
                    }
                    if (tot != 0) { x.seqId = static_cast<float>(idc) / tot; x.rySeqId = static_cast<float>(idr) / tot; }
                    x.alnLengthCons = (unsigned) tot;
                    // :249-270
                    unsigned minAlnLen = 500;
                    minAlnLen = (x.alnLength < minAlnLen) ? std::min(minAlnLen, static_cast<unsigned>(0.2 * x.dbLen)) : minAlnLen;
                    if (x.seqId >= mergeSeqIdThr && x.rySeqId >= ryThr && x.alnLength >= minAlnLen) {
                        // ancientMatchCount (nuclassembleUtil.cpp:1047-1181): the C->T / G->A columns each add the same posterior
                        float mCT = 0, mGA = 0;
                        unsigned mmCons = (1 - x.seqId) * x.alnLengthCons + 0.5;
                        unsigned mCons = x.alnLengthCons - mmCons;
                        unsigned scoreAln = mCons * 2 + mmCons * (-3);
                        if (leftStart || rightStart) {
                            const long double (*D)[4][4] = mats[x.isRev ? 1 : 0];
                            const double likCT = D[5][1][3], likGA = D[5][2][0];
                            if (likCT > 0) { const double v = deamMatches(x.alnLength, scoreAln, likCT); for (int i = 0; i < nCT; i++) mCT += v; }
                            if (likGA > 0) { const double v = deamMatches(x.alnLength, scoreAln, likGA); for (int i = 0; i < nGA; i++) mGA += v; }
                        }
                        x.deamMatch = ((static_cast<float>(scoreAln) + 3.0f * x.alnLengthCons) / 5.0f) + mCT + mGA;
                        x.cacheTerms();
                        queue.push(x);
                    }
                }
            }
            lap(0);
            // :276-470 extension
            if (queue.empty()) { outExt[id] = ext[id]; continue; }
            query.assign(q0.data(), q0.size());
            bool couldExtend = false;
            while (!queue.empty()) {
                unsigned leftOff = 0, rightOff = 0;
                parked.clear();
                Res best;
                while (selectFragment(queue, queryKey, best)) {
                    const SeqView &t = seqs[best.target];
                    const unsigned tLen = (unsigned) t.size();
                    if (best.dbStartPos == 0) { if ((tLen - (best.dbEndPos + 1)) <= rightOff) continue; }
                    else if (best.qStartPos == 0) { if (best.dbStartPos <= static_cast<int>(leftOff)) continue; }
                    const unsigned ds = best.dbStartPos, de = best.dbEndPos, qs = best.qStartPos, qe = best.qEndPos;
                    if (ds == 0 && qe == (qLen - 1)) {
                        if (rightOff > 0) { parked.push_back(best); continue; }
                        const unsigned fragLen = tLen - (de + 1);
                        if (query.size() + fragLen >= par->max_seq_len) break;
                        query += useReverse(best.target) ? revComp(t.data(), fragLen) : t.substr(de + 1, fragLen);
                        rightOff += fragLen;
                    } else if (qs == 0 && de == (tLen - 1)) {
                        if (leftOff > 0) { parked.push_back(best); continue; }
                        const unsigned fragLen = ds;
                        if (query.size() + fragLen >= par->max_seq_len) break;
                        query = (useReverse(best.target) ? revComp(t.data() + (tLen - ds), fragLen) : t.substr(0, fragLen)) + query;
                        leftOff += fragLen;
                    }
                }
                lap(1);
                if (leftOff > 0 || rightOff > 0) couldExtend = true;
                if (!queue.empty()) break;
                qLen = (unsigned) query.size();
                // :404-455 the parked hits on the grown query: ungappedAlignmentByDiagonal (mode 3), updateNuclAlignment, getRYSeqId
                for (Res &a : parked) {
                    // the target as the reference holds it here: its own letters, or getNuclRevFragment's (letter j = complement of letter tLen-1-j)
                    const SeqView &t0 = seqs[a.target];
                    const unsigned tLen = (unsigned) t0.size();
                    const bool rev = useReverse(a.target);
                    const int diag = (a.qStartPos + (int) leftOff) - a.dbStartPos;
                    const unsigned md = (unsigned) std::abs(diag);
                    int startPos = -1, endPos = -1; unsigned diagonalLen = 0;
                    const char *qa = nullptr; unsigned m = 0;
                    if (diag >= 0 && md < qLen) { m = std::min(tLen, qLen - md); qa = query.data() + md; }
                    else if (diag < 0 && md < tLen) { m = std::min(tLen - md, qLen); qa = query.data(); }
                    // the overlap's m letters of the target start at ta; of a reversed target only they are spelled out (not the whole contig)
                    const size_t ta = diag < 0 ? md : 0;
                    // ov[j] = letter ta + j of the target as the reference holds it here: fwd[j], or - of a reversed target, which is not
                    // spelled out - revLetter(back[-j])
                    const char *fwd = rev ? nullptr : t0.data() + ta, *back = t0.data() + (tLen - 1 - ta);
                    auto ovAt = [&](size_t j) { return fwd ? fwd[j] : revLetter(*(back - j)); };
                    if (qa) {       // computeGlobalSubstitutionStartEndDistance: the whole overlap, but for a '*' at either end (DistanceCalculator.h:204-220)
                        diagonalLen = m; startPos = (qa[0] == '*' || ovAt(0) == '*') ? 1 : 0; endPos = (int) m - 1;
                        if (endPos > 0 && (qa[m - 1] == '*' || ovAt(m - 1) == '*')) endPos--;
                    }
                    // updateNuclAlignment (nuclassembleUtil.cpp:9-47)
                    const int dist = (int) md;
                    int qs2, qe2, ds2, de2;
                    if (diag >= 0) { qs2 = startPos + dist; qe2 = endPos + dist; ds2 = startPos; de2 = endPos; }
                    else { qs2 = startPos; qe2 = endPos; ds2 = startPos + dist; de2 = endPos + dist; }
                    int idCnt = 0, idRy = 0;
                    // [qs2, qe2) for the identities (:28-31), [qs2, qe2] for the RY classes (getRYSeqId :78-92): columns startPos .. of the overlap
                    if (qa && endPos >= startPos) overlapCounts(qa, fwd, back, (size_t) startPos, (size_t) (endPos - startPos), idCnt, idRy);
                    a.seqId = static_cast<float>(idCnt) / (static_cast<float>(qe2) - static_cast<float>(qs2));
                    a.qLen = qLen; a.dbLen = tLen; a.alnLength = diagonalLen;
                    a.qStartPos = qs2; a.qEndPos = qe2; a.dbStartPos = ds2; a.dbEndPos = de2;
                    a.rySeqId = static_cast<float>(idRy) / a.alnLength;     // (no overlap left on that diagonal: 0/0 above, the hit is dropped whatever this is)
                    if (a.seqId >= mergeSeqIdThr && a.rySeqId >= ryThr) queue.push(a);
                }
                lap(3);
            }
            if (couldExtend) { mine.emplace_back((uint32_t) id, std::move(query)); outExt[id] = 1; }
            else outExt[id] = ext[id];
            lap(2);
        }
        if (timing) {
#pragma omp critical
            { for (int k = 0; k < 4; k++) { tSum[k] += tl[k]; tMax[k] = std::max(tMax[k], tl[k]); } nCmp += tlCompares; nTerms += tlSeriesTerms; }
            tlCompares = tlSeriesTerms = 0;
        }
    }
    if (timing) fprintf(stderr, "  contig merge host threads (sum / max s): gate %.2f / %.2f, queue + growth %.2f / %.2f, parked hits %.2f / %.2f, rest %.2f / %.2f\n",
                        tSum[0], tMax[0], tSum[1], tMax[1], tSum[3], tMax[3], tSum[2], tMax[2]);
    if (timing) fprintf(stderr, "  contig merge comparator: %llu calls, %llu series terms\n", nCmp, nTerms);
    {   // the grown contigs in the order of their queries
        size_t m = 0; for (auto &v : perThread) m += v.size();
        std::vector<std::pair<uint32_t, std::string> *> all; all.reserve(m);
        for (auto &v : perThread) for (auto &e : v) all.push_back(&e);
        std::sort(all.begin(), all.end(), [](const std::pair<uint32_t, std::string> *a, const std::pair<uint32_t, std::string> *b) { return a->first < b->first; });
        grownIdx.resize(m); grownSeqs.resize(m);
        for (size_t j = 0; j < m; j++) { grownIdx[j] = all[j]->first; grownSeqs[j].swap(all[j]->second); }
    }
    if (undefinedCase) { *err = "cdm_contig_merge: a target overhangs its query by more than the query's length; the reference pads it with a negative number of letters there (undefined behaviour), not reproduced"; return CDM_ERR_UNSUPPORTED; }
    return CDM_OK;
}
