// `align` as linclust runs it on the assembled contigs (lib/mmseqs/data/workflow/linclust.sh:68-73; parameters of `ancient_assemble`:
// nucleotides, --wrapped-scoring 1, -a 0, --gap-open 5 --gap-extend 2 --zdrop 200, query DB == target DB): the gapped step between the
// pre-clustering and the final clustering.  Host code - the contigs that survive the pre-clustering are few - restating
//   Alignment::run's loop over the prefilter lists          lib/mmseqs/src/alignment/Alignment.cpp:319-409 (criteria :555-573)
//   Matcher::getSWResult, nucleotide branch                 lib/mmseqs/src/alignment/Matcher.cpp:60-190, sort order Matcher.h:162-173
//   BandedNucleotideAligner::initQuery / align              lib/mmseqs/src/alignment/BandedNucleotideAligner.cpp:51-255
//   DistanceCalculator::computeUngapped(Wrapped)Alignment   lib/mmseqs/src/alignment/DistanceCalculator.h:57-113, :180-200
//   ksw_extz2_sse, ksw_backtrack, ksw_apply_zdrop           lib/mmseqs/lib/ksw2/ksw2_extz2_sse.cpp:45-284, ksw2.h:135-202 (ksw2 of
//                                                           minimap2, vendored by the reference; MIT)
// The banded extension is restated LANE BY LANE, not as a textbook DP: ksw2 works on 16-byte blocks that reach beyond the band (rows
// are rounded to multiples of 16), its match scores are written 16 at a time from the band's unaligned start, and what those extra
// lanes leave in the difference arrays is what a cell at the band's edge later reads - so the arrays here have ksw2's layout (u, v,
// x, y, s, the target, the reversed query in ONE zeroed buffer) and every byte operation is the scalar form of the instruction the
// reference executes (8-bit wrap-around adds, signed / unsigned 8-bit maxima).  tests/test_align_module.py compares the module with
// the reference's object code on contig sets with substitutions, insertions, deletions, rotations and reverse complements.
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

// ---- ksw2's extension alignment, restated lane by lane
struct Extz { int max = 0, max_q = -1, max_t = -1; bool zdropped = false; std::vector<uint32_t> cigar; };
inline bool applyZdrop(Extz &ez, int H, int r, int t, int zdrop, int e) {                          // ksw_apply_zdrop, is_rot = 1
    if (H > ez.max) { ez.max = H; ez.max_t = t; ez.max_q = r - t; }
    else if (t >= ez.max_t && r - t >= ez.max_q) {
        const int tl = t - ez.max_t, ql = (r - t) - ez.max_q, l = tl > ql ? tl - ql : ql - tl;
        if (zdrop >= 0 && ez.max - H > zdrop + l * e) { ez.zdropped = true; return true; }
    }
    return false;
}
inline void pushCigar(std::vector<uint32_t> &c, uint32_t op, int len) { if (c.empty() || op != (c.back() & 0xf)) c.push_back((uint32_t) len << 4 | op); else c.back() += (uint32_t) len << 4; }
// ksw_extz2_sse(km, qlen, query, tlen, target, m = 5, mat, q, e, w, zdrop, flag = KSW_EZ_EXTZ_ONLY [| KSW_EZ_SCORE_ONLY], &ez):
// match / mismatch scores mat[0] / mat[1], residue 4 a wildcard (score 0), gaps left-aligned, exact maximum
void extz(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t scMch, int8_t scMis, int8_t q, int8_t e, int w, int zdrop, bool withCigar, Extz &ez) {
    ez = Extz();
    if (qlen <= 0 || tlen <= 0) return;
    const int qe = q + e;
    const uint8_t qe2 = (uint8_t) ((q + e) * 2), maxSc = (uint8_t) (scMch + (q + e) * 2);
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    const int wl = w, wr = w, tlen_ = (tlen + 15) / 16, qlen_ = (qlen + 15) / 16;
    int nCol = qlen < tlen ? qlen : tlen;
    nCol = ((nCol < w + 1 ? nCol : w + 1) + 15) / 16 + 1;
    if (-(int) scMis > 2 * (q + e)) return;
    std::vector<uint8_t> mem((size_t) (tlen_ * 6 + qlen_ + 1) * 16, 0);
    uint8_t *u = mem.data(), *v = u + tlen_ * 16, *x = v + tlen_ * 16, *y = x + tlen_ * 16, *s = y + tlen_ * 16, *sf = s + tlen_ * 16, *qr = sf + tlen_ * 16;
    std::vector<int32_t> H((size_t) tlen_ * 16, -0x40000000);
    std::vector<uint8_t> p; std::vector<int> off, offEnd;
    if (withCigar) { p.assign(((size_t) (qlen + tlen - 1) * nCol + 1) * 16, 0); off.assign(qlen + tlen - 1, 0); offEnd.assign(qlen + tlen - 1, 0); }
    for (int t = 0; t < qlen; t++) qr[t] = query[qlen - 1 - t];
    memcpy(sf, target, (size_t) tlen);
    int lastSt = -1, lastEn = -1;
    for (int r = 0; r < qlen + tlen - 1; r++) {
        int st = 0, en = tlen - 1;
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
        if (en > (r + wl) >> 1) en = (r + wl) >> 1;
        if (st > en) { ez.zdropped = true; break; }
        const int st0 = st, en0 = en;
        st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
        uint8_t x1, v1;
        if (st > 0) { if (st - 1 >= lastSt && st - 1 <= lastEn) { x1 = x[st - 1]; v1 = v[st - 1]; } else x1 = v1 = 0; }
        else { x1 = 0; v1 = r ? (uint8_t) q : 0; }
        if (en >= r) { y[r] = 0; u[r] = r ? (uint8_t) q : 0; }
        // scores: 16 lanes at a time from the band's unaligned start (the lanes beyond en0 are written too - and a store may reach
        // the first bytes behind s, i.e. positions of the target in front of st0 that no later row reads)
        const uint8_t *qrr = qr + (qlen - 1 - r);
        for (int t = st0; t <= en0; t += 16)
            for (int k = 0; k < 16; k++) {
                const uint8_t a = sf[t + k], b = qrr[t + k];
                s[t + k] = (a == 4 || b == 4) ? (uint8_t) 0 : (uint8_t) (a == b ? scMch : scMis);
            }
        // core loop over the blocks [st, en]
        uint8_t *pr = withCigar ? p.data() + (size_t) r * nCol * 16 - st : nullptr;
        if (withCigar) { off[r] = st; offEnd[r] = en; }
        uint8_t xPrev = x1, vPrev = v1;
        for (int t = st; t <= en; t++) {
            uint8_t z = (uint8_t) (s[t] + qe2);
            const uint8_t xt1 = xPrev, vt1 = vPrev;
            xPrev = x[t]; vPrev = v[t];
            uint8_t a = (uint8_t) (xt1 + vt1);
            const uint8_t ut = u[t];
            uint8_t b = (uint8_t) (y[t] + ut);
            uint8_t d = 0;
            if (withCigar) d = ((int8_t) a > (int8_t) z) ? 1 : 0;
            z = ((int8_t) z > (int8_t) a) ? z : a;
            if (withCigar && (int8_t) b > (int8_t) z) d = 2;
            z = z > b ? z : b;                                  // unsigned
            z = z < maxSc ? z : maxSc;
            u[t] = (uint8_t) (z - vt1); v[t] = (uint8_t) (z - ut);
            z = (uint8_t) (z - (uint8_t) q);
            a = (uint8_t) (a - z); b = (uint8_t) (b - z);
            const bool ap = (int8_t) a > 0, bp = (int8_t) b > 0;
            x[t] = ap ? a : 0; y[t] = bp ? b : 0;
            if (withCigar) pr[t] = (uint8_t) (d | (ap ? 0x08 : 0) | (bp ? 0x10 : 0));
        }
        // the exact maximum of the row (four interleaved running maxima, as the reference's 4-lane loop keeps them)
        int32_t maxH, maxT;
        if (r > 0) {
            const int en1 = st0 + (en0 - st0) / 4 * 4;
            maxH = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] - qe : H[en0] + v[en0] - qe;
            maxT = en0;
            int32_t laneH[4] = {maxH, maxH, maxH, maxH}, laneT[4] = {maxT, maxT, maxT, maxT};
            int t = st0;
            for (; t < en1; t += 4)
                for (int i = 0; i < 4; i++) { H[t + i] += (int32_t) v[t + i] - qe; if (H[t + i] > laneH[i]) { laneH[i] = H[t + i]; laneT[i] = t; } }
            for (int i = 0; i < 4; i++) if (maxH < laneH[i]) { maxH = laneH[i]; maxT = laneT[i] + i; }
            for (; t < en0; t++) { H[t] += (int32_t) v[t] - qe; if (H[t] > maxH) { maxH = H[t]; maxT = t; } }
        } else { H[0] = v[0] - qe - qe; maxH = H[0]; maxT = 0; }
        if (applyZdrop(ez, maxH, r, maxT, zdrop, e)) break;
        lastSt = st; lastEn = en;
    }
    if (withCigar && ez.max_t >= 0 && ez.max_q >= 0) {          // ksw_backtrack(is_rot = 1, is_rev = 0, with_N = 0) from (max_t, max_q)
        int i = ez.max_t, j = ez.max_q, state = 0;
        const int nColB = nCol * 16;
        std::vector<uint32_t> &c = ez.cigar;
        while (i >= 0 && j >= 0) {
            const int r = i + j;
            int force = -1;
            if (i < off[r]) force = 2;
            if (i > offEnd[r]) force = 1;
            const uint32_t tmp = force < 0 ? p[(size_t) r * nColB + i - off[r]] : 0;
            if (state == 0) state = tmp & 7;
            else if (!(tmp >> (state + 2) & 1)) state = 0;
            if (state == 0) state = tmp & 7;
            if (force >= 0) state = force;
            if (state == 0) { pushCigar(c, 0, 1); --i; --j; }
            else if (state == 1 || state == 3) { pushCigar(c, 2, 1); --i; }
            else { pushCigar(c, 1, 1); --j; }
        }
        if (i >= 0) pushCigar(c, 2, i + 1);
        if (j >= 0) pushCigar(c, 1, j + 1);
        std::reverse(c.begin(), c.end());
    }
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
