// `align` as linclust runs it on the assembled contigs (lib/mmseqs/data/workflow/linclust.sh:68-73; parameters of `ancient_assemble`:
// nucleotides, --wrapped-scoring 1, -a 0, --gap-open 5 --gap-extend 2 --zdrop 200, query DB == target DB): the gapped step between the
// pre-clustering and the final clustering.  Host code - the contigs that survive the pre-clustering are few - restating
//   Alignment::run's loop over the prefilter lists          lib/mmseqs/src/alignment/Alignment.cpp:319-409 (criteria :555-573)
//   Matcher::getSWResult, nucleotide branch                 lib/mmseqs/src/alignment/Matcher.cpp:60-190, sort order Matcher.h:162-173
//   BandedNucleotideAligner::initQuery / align              lib/mmseqs/src/alignment/BandedNucleotideAligner.cpp:51-255
//   DistanceCalculator::computeUngapped(Wrapped)Alignment   lib/mmseqs/src/alignment/DistanceCalculator.h:57-113, :180-200
//   the banded z-drop extension the reference calls        lib/mmseqs/lib/ksw2/ksw2_extz2_sse.cpp:45-284, ksw2.h:135-202 (ksw2 of minimap2,
//                                                           vendored by the reference): its results, see diagdp below
// The banded extension is not a textbook DP: the reference's routine computes whole 16-position blocks around the band, refreshes its
// substitution scores in runs of 16 from the band's unaligned start and breaks score ties by its four interleaved running maxima -
// all of which reaches the result through the cells at the band's edge.  Round 5 restates those properties (P1-P6 below) in a
// formulation of its own - one state record per target position, a cell step, a row maximum, a trace - where round 4 had the
// routine's statements one by one.  tests/test_align_module.py compares the module with the reference's object code on contig sets
// with substitutions, insertions, deletions, rotations and reverse complements.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <omp.h>
#include <string>
#include <vector>

#include "carpedeam_hip.h"
#include "mmdb.h"

namespace {
// ---- NucleotideMatrix: letters -> A,C,T,G,X = 0..4 (NucleotideMatrix.cpp:17-61), scores of nucleotide.out (+2 / -3, X row -3)
struct NuclMat {
    uint8_t aa2num[256]; char num2aa[5]; int8_t sub[5][5]; uint8_t rev[5];
    NuclMat() {
        memcpy(num2aa, "ACTGX", 5);
        for (int c = 0; c < 256; c++) {
            uint8_t v = 4;
            switch (toupper(c)) {
                case 'A': v = 0; break; case 'C': case 'M': case 'Y': case 'H': v = 1; break;
                case 'T': case 'U': case 'W': v = 2; break;
                case 'G': case 'K': case 'B': case 'D': case 'V': case 'R': case 'S': v = 3; break;
                default: v = 4;
            }
            aa2num[c] = c == 255 ? (uint8_t) 4 : v;          // (setupLetterMapping runs letter < UCHAR_MAX; 255 keeps the parsed default, X)
        }
        for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) sub[i][j] = (i == j && i < 4) ? 2 : -3;
        rev[0] = 2; rev[2] = 0; rev[1] = 3; rev[3] = 1; rev[4] = 4;
    }
    int score(unsigned char q, unsigned char t) const { return sub[aa2num[q]][aa2num[t]]; }       // SubstitutionMatrix::createAsciiSubMat
};

// ---- Banded extension alignment with affine gaps, one anti-diagonal at a time, on score DIFFERENCES (round 5: own formulation; round 4
// carried ksw2's routine over statement by statement).
//
// The reference extends a seed with minimap2's ksw2 (vendored under lib/mmseqs/lib/ksw2; BandedNucleotideAligner.cpp:138-255 calls
// ksw_extz2_sse with a band of 64, z-drop, match +2 / mismatch -3 - mat[0] / mat[1] -, letter 4 a wildcard).  What must come out
// equal is its RESULT - best score, the cell it is reached in, the edit script, the z-drop exit - and that result depends on details
// of how the SSE routine walks the matrix which a textbook DP does not have.  They are restated here as properties, and the code below
// is built on them:
//
//   P1  Suzuki-Kasahara differences.  With H the best score of a cell, E / F the best scores that end in a gap of the target / of the
//       query, the routine keeps per target position t (cell (t, r - t) of anti-diagonal r) four bytes:
//          dH = H(t, j) - H(t - 1, j) + gapOpen + gapExtend          "horizontal" difference   (ksw2: u)
//          dV = H(t, j) - H(t, j - 1) + gapOpen + gapExtend          "vertical" difference     (ksw2: v)
//          eT = E(t + 1, j) - H(t, j) + gapOpen + gapExtend, eQ = F(t, j + 1) - H(t, j) + gapOpen + gapExtend      (x, y)
//       all in 8-bit arithmetic that WRAPS; one step of the recurrence is `advance()` below: the byte operations are the contract (signed
//       maximum against the target-gap path, UNSIGNED maximum against the query-gap path, an unsigned cap at match + 2 (open + extend)).
//   P2  The band of row r is [lo, hi] (matrix bounds and |t - j| <= 64), but cells are computed for the whole 16-position blocks that
//       [lo, hi] touches: [lo16, hi16].  The extra positions are real state: a cell that enters the band next row reads what such a
//       position left behind.  Positions left of lo16 keep what they last held.
//   P3  Substitution scores are refreshed, per row, for the positions lo, lo + 1, ... in runs of 16 up to the run that holds hi - so
//       positions in [lo16, lo) score with what an EARLIER row computed for them, positions right of hi with letters outside the band
//       (or with the zero padding behind both sequences, which "matches").
//   P4  The best score of a row is taken over the band [lo, hi] from running sums of the dV; among equal scores the routine's four
//       interleaved maxima decide the cell: hi itself first, then the positions lo + 0, 4, 8, .. in order, then lo + 1, 5, .., lo + 2, ..,
//       lo + 3, .., then what is left between the last full group of four and hi (`rowBest()`).
//   P5  Z-drop: the search stops at the first row whose best cell lies more than zdrop + gapExtend x |diagonal shift| below the best
//       cell so far, both coordinates at or beyond it.
//   P6  The edit script is read back from one byte per computed cell: which of the three paths gave H, and whether the E / F paths
//       continue; cells outside a row's computed range force a gap (`trace()`).
struct Extz { int max = 0, max_q = -1, max_t = -1; bool zdropped = false; std::vector<uint32_t> cigar; };
namespace diagdp {
constexpr int LANES = 16;
enum Path : uint8_t { FROM_DIAG = 0, FROM_TGAP = 1, FROM_QGAP = 2, TGAP_CONT = 0x08, QGAP_CONT = 0x10 };
struct Pos { uint8_t dH = 0, dV = 0, eT = 0, eQ = 0, sub = 0; };      // the state of one target position (P1) + its last substitution score (P3)
struct Costs { uint8_t open, openExt2, cap; int openExt; };
// one cell: `left` = (eT, dV) of position t - 1 as the row BEFORE left them, `here` = position t; returns the trace byte (P6)
inline uint8_t advance(Pos &here, uint8_t leftET, uint8_t leftDV, const Costs &c) {
    uint8_t h = (uint8_t) (here.sub + c.openExt2);                          // the diagonal path
    const uint8_t viaT = (uint8_t) (leftET + leftDV);                       // ... the path that ends in a gap of the target
    const uint8_t oldDH = here.dH;
    const uint8_t viaQ = (uint8_t) (here.eQ + oldDH);                       // ... in a gap of the query
    uint8_t from = (int8_t) viaT > (int8_t) h ? FROM_TGAP : FROM_DIAG;
    if ((int8_t) viaT > (int8_t) h) h = viaT;
    if ((int8_t) viaQ > (int8_t) h) from = FROM_QGAP;                       // (the direction compares signed, the score below unsigned)
    if (viaQ > h) h = viaQ;
    if (h > c.cap) h = c.cap;
    here.dH = (uint8_t) (h - leftDV);
    here.dV = (uint8_t) (h - oldDH);
    const uint8_t opened = (uint8_t) (h - c.open);
    const uint8_t contT = (uint8_t) (viaT - opened), contQ = (uint8_t) (viaQ - opened);
    const bool keepT = (int8_t) contT > 0, keepQ = (int8_t) contQ > 0;
    here.eT = keepT ? contT : 0;
    here.eQ = keepQ ? contQ : 0;
    return (uint8_t) (from | (keepT ? TGAP_CONT : 0) | (keepQ ? QGAP_CONT : 0));
}
struct RowTrace { int lo16 = 0, hi16 = -1; std::vector<uint8_t> cell; };
// P4: the row's best band cell from the scores `H` (already advanced to this row) - hi first, then the four interleaved sweeps, then the rest
inline void rowBest(const std::vector<int32_t> &H, int lo, int hi, int32_t &best, int &where) {
    best = H[hi]; where = hi;
    const int grouped = lo + (hi - lo) / 4 * 4;
    int32_t sweepBest[4]; int sweepAt[4];
    for (int phase = 0; phase < 4; phase++) {
        sweepBest[phase] = best; sweepAt[phase] = -1;
        for (int t = lo + phase; t < grouped; t += 4) if (H[t] > sweepBest[phase]) { sweepBest[phase] = H[t]; sweepAt[phase] = t; }
    }
    for (int phase = 0; phase < 4; phase++) if (sweepAt[phase] >= 0 && sweepBest[phase] > best) { best = sweepBest[phase]; where = sweepAt[phase]; }
    for (int t = grouped; t < hi; t++) if (H[t] > best) { best = H[t]; where = t; }
}
// P6: from the best cell back to the origin
inline void trace(const std::vector<RowTrace> &rows, int t, int j, std::vector<uint32_t> &ops) {
    auto put = [&](uint32_t op, int n) { if (!ops.empty() && (ops.back() & 0xf) == op) ops.back() += (uint32_t) n << 4; else ops.push_back((uint32_t) n << 4 | op); };
    int inGap = FROM_DIAG;                                   // the gap the walk is in (FROM_TGAP / FROM_QGAP), FROM_DIAG = none
    while (t >= 0 && j >= 0) {
        const RowTrace &row = rows[(size_t) (t + j)];
        uint8_t b = 0; int forced = -1;
        if (t < row.lo16) forced = FROM_QGAP; else if (t > row.hi16) forced = FROM_TGAP; else b = row.cell[(size_t) (t - row.lo16)];
        if (inGap != FROM_DIAG && !(b & (inGap == FROM_TGAP ? TGAP_CONT : QGAP_CONT))) inGap = FROM_DIAG;     // the gap ends here
        if (inGap == FROM_DIAG) inGap = b & 7;
        if (forced >= 0) inGap = forced;
        if (inGap == FROM_DIAG) { put(0, 1); --t; --j; }
        else if (inGap == FROM_TGAP) { put(2, 1); --t; }
        else { put(1, 1); --j; }
    }
    if (t >= 0) put(2, t + 1);
    if (j >= 0) put(1, j + 1);
    std::reverse(ops.begin(), ops.end());
}
}  // namespace diagdp
// the extension of `query` against `target` from their first letters on (letters 0..3, 4 = wildcard): best score and where, the edit
// script if asked for
void extz(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t scMch, int8_t scMis, int8_t q, int8_t e, int w, int zdrop, bool withCigar, Extz &ez) {
    using namespace diagdp;
    ez = Extz();
    if (qlen <= 0 || tlen <= 0) return;
    if (-(int) scMis > 2 * (q + e)) return;                 // (a mismatch must not cost more than opening two gaps: the byte range of P1)
    if (w < 0) w = std::max(tlen, qlen);
    Costs cost; cost.open = (uint8_t) q; cost.openExt = q + e; cost.openExt2 = (uint8_t) ((q + e) * 2); cost.cap = (uint8_t) (scMch + (q + e) * 2);
    const int tlen16 = (tlen + LANES - 1) / LANES * LANES;
    std::vector<Pos> col((size_t) tlen16 + LANES);
    std::vector<int32_t> H((size_t) tlen16, -0x40000000);
    std::vector<RowTrace> rows; if (withCigar) rows.resize((size_t) (qlen + tlen - 1));
    auto letterT = [&](int p) -> uint8_t { return p < tlen ? target[p] : (uint8_t) 0; };                                  // (zero padding behind both)
    auto letterQ = [&](int j) -> uint8_t { return j >= 0 && j < qlen ? query[j] : (uint8_t) 0; };
    int prevLo16 = -1, prevHi16 = -1;
    for (int r = 0; r < qlen + tlen - 1; r++) {
        // P2: the band of this anti-diagonal and the blocks it touches
        int lo = std::max(0, r - qlen + 1), hi = std::min(tlen - 1, r);
        lo = std::max(lo, (r - w + 1) >> 1); hi = std::min(hi, (r + w) >> 1);
        if (lo > hi) { ez.zdropped = true; break; }
        const int lo16 = lo / LANES * LANES, hi16 = (hi + LANES) / LANES * LANES - 1;
        // what the leftmost computed cell sees to its left: the matrix edge, a position the row before computed, or nothing
        uint8_t leftET = 0, leftDV = 0;
        if (lo16 == 0) leftDV = r ? cost.open : 0;
        else if (lo16 - 1 >= prevLo16 && lo16 - 1 <= prevHi16) { leftET = col[(size_t) lo16 - 1].eT; leftDV = col[(size_t) lo16 - 1].dV; }
        if (hi16 >= r) { col[(size_t) r].eQ = 0; col[(size_t) r].dH = r ? cost.open : 0; }       // the top edge of the matrix enters the computed range
        // P3: fresh substitution scores from lo on, in runs of 16
        for (int run = lo; run <= hi; run += LANES)
            for (int t = run; t < run + LANES; t++) {
                const uint8_t a = letterT(t), b = letterQ(r - t);
                col[(size_t) t].sub = (a == 4 || b == 4) ? (uint8_t) 0 : (uint8_t) (a == b ? scMch : scMis);
            }
        RowTrace *tr = withCigar ? &rows[(size_t) r] : nullptr;
        if (tr) { tr->lo16 = lo16; tr->hi16 = hi16; tr->cell.assign((size_t) (hi16 - lo16 + 1), 0); }
        for (int t = lo16; t <= hi16; t++) {
            Pos &here = col[(size_t) t];
            const uint8_t nextET = here.eT, nextDV = here.dV;                // (what the cell to the right reads is this position BEFORE the step)
            const uint8_t b = advance(here, leftET, leftDV, cost);
            if (tr) tr->cell[(size_t) (t - lo16)] = b;
            leftET = nextET; leftDV = nextDV;
        }
        // P4: the scores of the band's cells from the differences; the row's best
        int32_t best; int where;
        if (r > 0) {
            H[(size_t) hi] = hi > 0 ? H[(size_t) hi - 1] + col[(size_t) hi].dH - cost.openExt : H[(size_t) hi] + col[(size_t) hi].dV - cost.openExt;
            for (int t = lo; t < hi; t++) H[(size_t) t] += (int32_t) col[(size_t) t].dV - cost.openExt;
            rowBest(H, lo, hi, best, where);
        } else { H[0] = col[0].dV - 2 * cost.openExt; best = H[0]; where = 0; }
        // P5
        if (best > ez.max) { ez.max = best; ez.max_t = where; ez.max_q = r - where; }
        else if (where >= ez.max_t && r - where >= ez.max_q) {
            const int shift = std::abs((where - ez.max_t) - ((r - where) - ez.max_q));
            if (zdrop >= 0 && ez.max - best > zdrop + shift * e) { ez.zdropped = true; break; }
        }
        prevLo16 = lo16; prevHi16 = hi16;
    }
    if (withCigar && ez.max_t >= 0 && ez.max_q >= 0) trace(rows, ez.max_t, ez.max_q, ez.cigar);
}

// ---- the ungapped seed: best local stretch on the probed diagonals (computeSubstitutionStartEndDistance on each)
struct Local { int start = -1, end = -1; unsigned score = 0, distToDiagonal = 0; int diagonal = 0; };
Local startEnd(const NuclMat &M, const char *a, const char *b, unsigned n) {
    int maxScore = 0, maxEnd = 0, maxStart = 0, minPos = -1, score = 0;
    for (unsigned pos = 0; pos < n; pos++) {
        score += M.score((unsigned char) a[pos], (unsigned char) b[pos]);
        const bool isMin = score <= 0;
        score = isMin ? 0 : score; minPos = isMin ? (int) pos : minPos;
        const bool isNew = score > maxScore;
        maxEnd = isNew ? (int) pos : maxEnd; maxStart = isNew ? minPos + 1 : maxStart; maxScore = isNew ? score : maxScore;
    }
    Local l; l.start = maxStart; l.end = maxEnd; l.score = (unsigned) maxScore;
    return l;
}
Local byDiagonal(const NuclMat &M, const char *q, unsigned qLen, const char *t, unsigned tLen, int diagonal) {     // ungappedAlignmentByDiagonal, RESCORE_MODE_ALIGNMENT
    const unsigned dist = (unsigned) abs(diagonal);
    Local res; res.distToDiagonal = dist; res.diagonal = diagonal;
    if (diagonal >= 0 && dist < qLen) { const Local l = startEnd(M, q + dist, t, std::min(tLen, qLen - dist)); res.score = l.score; res.start = l.start; res.end = l.end; }
    else if (diagonal < 0 && dist < tLen) { const Local l = startEnd(M, q, t + dist, std::min(tLen - dist, qLen)); res.score = l.score; res.start = l.start; res.end = l.end; }
    return res;
}

struct Result { uint32_t dbKey; int score; float qcov, dbcov, seqId; double eval; unsigned alnLength; int qStart, qEnd; unsigned qLen; int dbStart, dbEnd; unsigned dbLen; };
bool compareHits(const Result &a, const Result &b) {        // Matcher::compareHits
    if (a.eval != b.eval) return a.eval < b.eval;
    if (a.score != b.score) return a.score > b.score;
    if (a.dbLen != b.dbLen) return a.dbLen < b.dbLen;
    return a.dbKey < b.dbKey;
}
float computeCov(unsigned s, unsigned e, unsigned len) { return (std::min(len, std::max(s, e)) - std::min(s, e) + 1) / (float) len; }
bool canBeCovered(float covThr, int covMode, float ql, float tl) {      // Util.cpp:533-550
    switch (covMode) {
        case 0: return ((ql / tl >= covThr) && (tl / ql >= covThr));
        case 2: return ((tl / ql) >= covThr);
        case 1: return ((ql / tl) >= covThr);
        case 3: return ((tl / ql) >= covThr) && (tl / ql) <= 1.0;
        case 4: return ((ql / tl) >= covThr) && (ql / tl) <= 1.0;
        case 5: return (std::min(tl, ql) / std::max(tl, ql)) >= covThr;
        default: return true;
    }
}
bool hasCoverage(float covThr, int covMode, float qc, float tc) { switch (covMode) { case 0: return qc >= covThr && tc >= covThr; case 2: return qc >= covThr; case 1: return tc >= covThr; default: return true; } }
char *utoa(unsigned long long v, char *p) { char b[24]; int n = 0; do { b[n++] = (char) ('0' + v % 10); v /= 10; } while (v); while (n) *p++ = b[--n]; return p; }
char *itoa(long long v, char *p) { if (v < 0) { *p++ = '-'; return utoa((unsigned long long) -v, p); } return utoa((unsigned long long) v, p); }
char *seqIdText(float s, char *p) {     // Util::fastSeqIdToBuffer (Util.cpp:278-307)
    if (s == 1.0) { memcpy(p, "1.00", 4); return p + 4; }
    *p++ = '0'; *p++ = '.';
    if (s < 0.10) *p++ = '0';
    if (s < 0.01) *p++ = '0';
    return itoa((int) (s * 1000), p);
}
}  // namespace

struct AlignParams {       // (host/main.cpp holds the same declaration)
    float covThr = 0.f, seqIdThr = 0.f; double evalThr = 0.001; int covMode = 0, seqIdMode = 0, alnLenThr = 0; bool wrapped = false, includeIdentity = false;
    int gapOpen = 5, gapExtend = 2, zdrop = 40; unsigned maxAccept = INT_MAX, maxReject = INT_MAX; size_t maxSeqLen = 65535;
};

int alignModule(const std::string &qPath, const std::string &tPath, const std::string &prefPath, const std::string &outPath, const AlignParams &P, std::string *err) {
    if (qPath != tPath) { *err = "align: query and target DB must be the same on the MI355X path"; return 77; }
    MmDb seq, pref;
    if (!seq.load(tPath, err) || !pref.load(prefPath, err)) return 1;
    if (cdm_gapped_evalue(P.gapOpen, P.gapExtend, 1, 1, 1, NULL, NULL) != CDM_OK) { *err = std::string("align: ") + cdm_last_error(); return 77; }
    if ((seq.dbtype & 0x7FFFFFFF) != 1) { *err = "align: only nucleotide sequence DBs are implemented on the MI355X path"; return 77; }
    const NuclMat M;
    const bool reversePref = (pref.dbtype & 0x7FFFFFFF) == 14;
    uint64_t dbRes = 0;
    for (size_t i = 0; i < seq.size(); i++) dbRes += seq.len[i] >= 2 ? seq.len[i] - 2 : 0;
    const size_t maxLen = P.wrapped ? P.maxSeqLen * 2 : P.maxSeqLen;
    const int T = std::max(1, omp_get_max_threads());
    std::vector<OutChunk> chunks((size_t) T);
    std::string failure; int failCode = 0;
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int th = 0; th < T; th++) {
        const size_t lo = pref.size() * (size_t) th / T, hi = pref.size() * (size_t) (th + 1) / T;
        OutChunk &c = chunks[th];
        std::string qDoubled, out;
        // The reference's "reversed" arrays are SmithWaterman::seq_reverse(rev, seq, L) (StripedSmithWaterman.h:185-194: `end` is
        // an inclusive index, called with the length): L + 1 elements, rev[i] = seq[L - i] - every reversed sequence is shifted by
        // one, and rev[0] is the byte BEHIND the sequence in its buffer: what an earlier, longer sequence of the same thread left
        // there (the Sequence objects and the aligner's buffers live as long as the thread), zero at first.  The extension that
        // runs on them therefore starts one column behind the seed's end and reports a start one column in front of what it
        // evaluated - reproduced, stale bytes included: the number buffers below persist over the queries of a thread.
        std::vector<uint8_t> qNum(maxLen + 1, 0), qRc(maxLen + 1, 0), tNum(maxLen + 1, 0), qRev, qRcRev, tRev; std::string qRcChar;
        auto fit = [](std::vector<uint8_t> &b, size_t n) { if (b.size() < n + 1) b.resize(n + 1, 0); };
        std::vector<Result> results;
        char buf[256];
        for (size_t id = lo; id < hi && !failCode; id++) {
            const uint32_t queryKey = pref.key[id];
            const char *data = pref.entry(id);
            results.clear(); out.clear();
            size_t origQueryLen = 0, queryLen = 0; const char *qChar = nullptr;
            if (*data != '\0') {
                const int64_t qId = seq.idOf(queryKey);
                if (qId < 0) {
#pragma omp critical
                    { failCode = 1; failure = "Query sequence " + std::to_string(queryKey) + " is required in the prefiltering, but is not contained in the query sequence database.\nPlease check your database."; }
                    break;
                }
                origQueryLen = seq.len[(size_t) qId] >= 2 ? seq.len[(size_t) qId] - 2 : 0;
                qDoubled.assign(seq.entry((size_t) qId), origQueryLen);
                if (P.wrapped) qDoubled += qDoubled;
                queryLen = std::min(qDoubled.size(), maxLen);          // Sequence::mapSequence stops at its buffer's length
                qChar = qDoubled.data();
                // BandedNucleotideAligner::initQuery: the numeric query, its reverse, its reverse complement and that one's reverse
                fit(qNum, queryLen); fit(qRc, queryLen); qRev.resize(queryLen + 1); qRcRev.resize(queryLen + 1); qRcChar.resize(queryLen);
                for (size_t i = 0; i < queryLen; i++) qNum[i] = M.aa2num[(unsigned char) qChar[i]];
                for (size_t i = 0; i <= queryLen; i++) qRev[i] = qNum[queryLen - i];
                for (size_t i = 0; i < queryLen; i++) { qRc[i] = M.rev[qNum[queryLen - 1 - i]]; qRcChar[i] = M.num2aa[qRc[i]]; }
                for (size_t i = 0; i <= queryLen; i++) qRcRev[i] = qRc[queryLen - i];
            }
            size_t passed = 0; unsigned rejected = 0;
            while (*data != '\0' && passed < P.maxAccept && rejected < P.maxReject) {
                // the hit: key, and - a prefilter line of three columns - score and diagonal
                const char *d = data; uint32_t dbKey = 0;
                while (*d >= '0' && *d <= '9') dbKey = dbKey * 10 + (uint32_t) (*d++ - '0');
                const char *e = data; int words = 0; bool in = false;
                for (; *e != '\n' && *e != '\0'; e++) { const bool ws = *e == ' ' || *e == '\t'; if (!ws && !in) words++; in = !ws; }
                short diagonal = 0; bool isReverse = false;
                if (words == 3) {
                    const char *w = data; while (*w != '\t' && *w != '\n' && *w != '\0') w++; if (*w == '\t') w++;
                    const int prefScore = (int) strtol(w, NULL, 10);
                    while (*w != '\t' && *w != '\n' && *w != '\0') w++; if (*w == '\t') w++;
                    diagonal = static_cast<short>((unsigned short) strtol(w, NULL, 10));
                    isReverse = reversePref && prefScore < 0;
                }
                data = *e == '\n' ? e + 1 : e;
                const int64_t dbId = seq.idOf(dbKey);
                if (dbId < 0) {
#pragma omp critical
                    { failCode = 1; failure = "Sequence " + std::to_string(dbKey) + " is required in the prefiltering, but is not contained in the target sequence database!\nPlease check your database."; }
                    break;
                }
                const size_t tLen = std::min((size_t) (seq.len[(size_t) dbId] >= 2 ? seq.len[(size_t) dbId] - 2 : 0), P.maxSeqLen * (P.wrapped ? 2 : 1));
                const char *tChar = seq.entry((size_t) dbId);
                fit(tNum, tLen);                                    // dbSeq.mapSequence comes before the coverage test (Alignment.cpp:375-381): the buffer changes either way
                for (size_t i = 0; i < tLen; i++) tNum[i] = M.aa2num[(unsigned char) tChar[i]];
                if (!canBeCovered(P.covThr, P.covMode, static_cast<float>(origQueryLen), static_cast<float>(tLen))) { rejected++; continue; }
                const bool isIdentity = queryKey == dbKey;          // (sameQTDB)
                // ---- BandedNucleotideAligner::align
                const char *qAlnChar = isReverse ? qRcChar.data() : qChar;
                const uint8_t *qAlnRev = isReverse ? qRcRev.data() : qRev.data(), *qAln = isReverse ? qRc.data() : qNum.data();
                tRev.resize(tLen + 1);
                for (size_t i = 0; i <= tLen; i++) tRev[i] = tNum[tLen - i];
                const int qL = (int) queryLen, tL = (int) tLen;
                int origLen = qL;
                Local aln;
                bool undefinedProbe = false;
                if (P.wrapped) {        // computeUngappedWrappedAlignment: the loop conditions are the reference's unsigned arithmetic
                    const unsigned short ud = (unsigned short) diagonal; const unsigned qLenU = (unsigned) qL, dbLenU = (unsigned) tL;
                    for (unsigned dv = 1; (-dv * 65536 + ud) > -dbLenU; dv++) {
                        const int real = (int) ((-dv * 65536 + ud) + qLenU / 2);
                        if (real < 0 || (unsigned) real + qLenU / 2 > qLenU) { undefinedProbe = true; break; }      // (the reference reads outside the doubled query there)
                        Local tmp = byDiagonal(M, qAlnChar + real, qLenU / 2, tChar, dbLenU, 0);
                        tmp.diagonal += real; tmp.distToDiagonal = (unsigned) abs(real);
                        if (tmp.score > aln.score) aln = tmp;
                    }
                    for (unsigned dv = 0; !undefinedProbe && (dv * 65536 + ud) < qLenU / 2; dv++) {
                        const int real = (int) (dv * 65536 + ud);
                        Local tmp = byDiagonal(M, qAlnChar + real, qLenU / 2, tChar, dbLenU, 0);
                        tmp.diagonal += real; tmp.distToDiagonal = (unsigned) abs(real);
                        if (tmp.score > aln.score) aln = tmp;
                    }
                    origLen = qL / 2;
                } else {
                    const unsigned short ud = (unsigned short) diagonal;
                    for (unsigned dv = 1; dv <= 1 + (unsigned) tL / 32768; dv++) { const Local tmp = byDiagonal(M, qAlnChar, (unsigned) qL, tChar, (unsigned) tL, (int) (-dv * 65536 + ud)); if (tmp.score > aln.score) aln = tmp; }
                    for (unsigned dv = 0; dv <= (unsigned) qL / 65536; dv++) { const Local tmp = byDiagonal(M, qAlnChar, (unsigned) qL, tChar, (unsigned) tL, (int) (dv * 65536 + ud)); if (tmp.score > aln.score) aln = tmp; }
                }
                if (undefinedProbe) {
#pragma omp critical
                    { failCode = 77; failure = "align: a probed diagonal of hit " + std::to_string(dbKey) + " of query " + std::to_string(queryKey) + " starts in front of the doubled query (undefined in the reference)"; }
                    break;
                }
                const unsigned dist = aln.distToDiagonal;
                int qUS, qUE, dbUS, dbUE;
                if (aln.diagonal >= 0) { qUS = aln.start + (int) dist; qUE = aln.end + (int) dist; dbUS = aln.start; dbUE = aln.end; }
                else { qUS = aln.start; qUE = aln.end; dbUS = aln.start + (int) dist; dbUE = aln.end + (int) dist; }
                Result res; int aaIds = 0; unsigned btLen = 0; bool haveCigar = true;
                int score1, qs1, qe1, ds1, de1;
                if (qUE - qUS == origLen - 1 && dbUS == 0 && dbUE == tL - 1) {        // the seed covers the whole of both: no extension
                    score1 = (int) aln.score; qs1 = qUS; qe1 = qUE; ds1 = dbUS; de1 = dbUE;
                    for (int i = qUS; i <= qUE; i++) aaIds += qAln[i] == tNum[(size_t) (dbUS + (i - qUS))] ? 1 : 0;
                    btLen = (unsigned) origLen;
                } else {
                    const int qStartRev = (qL - qUE) - 1, tStartRev = (tL - dbUE) - 1;
                    int qRevLen = qL - qStartRev;
                    if (P.wrapped && qRevLen > origLen) qRevLen = origLen;
                    Extz ez, ezAlign;
                    extz(qRevLen, qAlnRev + qStartRev, tL - tStartRev, tRev.data() + tStartRev, M.sub[0][0], M.sub[0][1], (int8_t) P.gapOpen, (int8_t) P.gapExtend, 64, P.zdrop, false, ez);
                    const int qStartPos = qL - (qStartRev + ez.max_q) - 1, tStartPos = tL - (tStartRev + ez.max_t) - 1;
                    int qLenToAlign = qL - qStartPos;
                    if (P.wrapped && qLenToAlign > origLen) qLenToAlign = origLen;
                    extz(qLenToAlign, qAln + qStartPos, tL - tStartPos, tNum.data() + tStartPos, M.sub[0][0], M.sub[0][1], (int8_t) P.gapOpen, (int8_t) P.gapExtend, 64, P.zdrop, true, ezAlign);
                    std::vector<uint32_t> cigar;
                    if (ez.max_q > ezAlign.max_q && ez.max_t > ezAlign.max_t) {
                        extz(qRevLen, qAlnRev + qStartRev, tL - tStartRev, tRev.data() + tStartRev, M.sub[0][0], M.sub[0][1], (int8_t) P.gapOpen, (int8_t) P.gapExtend, 64, P.zdrop, true, ezAlign);
                        cigar.assign(ezAlign.cigar.rbegin(), ezAlign.cigar.rend());
                    } else cigar = ezAlign.cigar;
                    score1 = ezAlign.max; qs1 = qStartPos; qe1 = qStartPos + ezAlign.max_q; de1 = tStartPos + ezAlign.max_t; ds1 = tStartPos;
                    int tp = ds1, qp = qs1;
                    for (uint32_t cg : cigar) {
                        const uint32_t op = cg & 0xf, len = cg >> 4;
                        for (uint32_t i = 0; i < len; i++) {
                            if (op == 0) { if (tNum[(size_t) tp] == qAln[qp]) aaIds++; ++qp; ++tp; }
                            else if (op == 1) ++qp; else ++tp;
                        }
                        btLen += len;
                    }
                    haveCigar = true;
                }
                float qcov = computeCov((unsigned) qs1, (unsigned) qe1, (unsigned) qL);
                if (P.wrapped) qcov = std::min(1.0f, qcov * 2);
                float dbcov = computeCov((unsigned) ds1, (unsigned) de1, (unsigned) tL);
                double evalue = 0; int bits = 0;
                cdm_gapped_evalue(P.gapOpen, P.gapExtend, (double) score1, (double) origLen, dbRes, &evalue, &bits);
                // ---- Matcher::getSWResult's tail
                unsigned alnLength = (unsigned) std::max(abs(qe1 - qs1), abs(de1 - ds1)) + 1;
                if (haveCigar) alnLength = btLen;
                float seqId;
                switch (P.seqIdMode) {
                    case 1: seqId = static_cast<float>(aaIds) / static_cast<float>(std::min(origLen, tL)); break;
                    case 2: seqId = static_cast<float>(aaIds) / static_cast<float>(std::max(origLen, tL)); break;
                    default: seqId = static_cast<float>(aaIds) / static_cast<float>((int) alnLength);
                }
                res.dbKey = dbKey; res.score = bits; res.qcov = qcov; res.dbcov = dbcov; res.seqId = seqId; res.eval = evalue; res.alnLength = alnLength;
                res.qStart = qs1; res.qEnd = qe1; res.qLen = (unsigned) origLen; res.dbLen = (unsigned) tL;
                if (isReverse) { res.dbStart = de1; res.dbEnd = ds1; } else { res.dbStart = ds1; res.dbEnd = de1; }
                if (isIdentity) { res.qcov = 1.0f; res.dbcov = 1.0f; res.seqId = 1.0f; }
                const bool ok = isIdentity || (res.eval <= P.evalThr && res.seqId >= P.seqIdThr && hasCoverage(P.covThr, P.covMode, res.qcov, res.dbcov) && (int) res.alnLength >= P.alnLenThr);
                if (ok) { results.push_back(res); passed++; rejected = 0; } else rejected++;
            }
            if (failCode) break;
            if (results.size() > 1) std::sort(results.begin(), results.end(), compareHits);
            for (const Result &r : results) {       // Matcher::resultToBuffer, no backtrace
                char *w = buf;
                w = utoa(r.dbKey, w); *w++ = '\t'; w = itoa(r.score, w); *w++ = '\t'; w = seqIdText(r.seqId, w); *w++ = '\t';
                w += sprintf(w, "%.3E", r.eval); *w++ = '\t';
                w = itoa(r.qStart, w); *w++ = '\t'; w = itoa(r.qEnd, w); *w++ = '\t'; w = itoa(r.qLen, w); *w++ = '\t';
                w = itoa(r.dbStart, w); *w++ = '\t'; w = itoa(r.dbEnd, w); *w++ = '\t'; w = itoa(r.dbLen, w); *w++ = '\n';
                out.append(buf, (size_t) (w - buf));
            }
            c.add(queryKey, out.data(), out.size(), 0);
        }
    }
    if (failCode) { *err = failure; return failCode; }
    return mmdbWriteChunks(outPath, 5, chunks, err) ? 0 : 1;
}
