// =============================================================================
// TEST INFRASTRUCTURE ONLY.  CPU restatement ("oracle") of the CarpeDeam hot path
//   kmermatcher -> rescorediagonal -> ancient_correction -> ancient_read_assemble
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build,
// call or execute this file.  The product (carpedeam_amd/, include/) never does.
//
// Parity status: PINNED.  Every stage is checked by tests/test_oracle_golden.py
// against keyed DB dumps and function-level known answers produced by the
// reference's own object code (oracle/_ref, built by oracle/Makefile.ref) and
// committed under tests/golden/.
//
// Each function cites the reference file:line it follows (paths relative to
// /root/reference; M/ = lib/mmseqs/src/).  Nothing here is copied: the reference's
// algorithms are restated on plain arrays, without its DBReader/Parameters/Sequence
// framework.
// =============================================================================
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <queue>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>
#include <sys/stat.h>
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#include <parallel/algorithm>
#endif

typedef long double ld;

// ----------------------------------------------------------------------------- DB I/O
// Format: M/commons/DBReader.cpp:773-838 (index: key, offset, length, wasExtended),
// DBWriter.cpp:193-213 (dbtype), entries are payload + '\0', length includes the NUL.
struct Db {
    std::vector<uint32_t> key;
    std::vector<size_t> off, len;
    std::vector<uint8_t> ext;
    std::string data;
    int dbtype = 0;
    size_t dataSize = 0;     // sum of index lengths (DBReader::readIndex localDataSize)
    unsigned maxSeqLen = 0;  // max index length
    uint32_t lastKey = 0;

    static bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
    static void slurp(const std::string &p, std::string &out) {
        std::ifstream f(p, std::ios::binary);
        out.append(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    }
    void load(const std::string &path) {
        if (exists(path)) slurp(path, data);
        else for (int i = 0; exists(path + "." + std::to_string(i)); i++) slurp(path + "." + std::to_string(i), data);
        std::ifstream t(path + ".dbtype", std::ios::binary);
        int32_t v = 0; t.read((char *) &v, 4); dbtype = v;
        std::ifstream ix(path + ".index");
        if (!ix.good()) { fprintf(stderr, "cannot open %s.index\n", path.c_str()); exit(1); }
        struct E { uint32_t k; size_t o, l; uint8_t e; };
        std::vector<E> es;
        std::string line;
        while (std::getline(ix, line)) {
            if (line.empty()) continue;
            E e; unsigned long long k, o, l, x = 0;
            int n = sscanf(line.c_str(), "%llu\t%llu\t%llu\t%llu", &k, &o, &l, &x);
            if (n < 3) continue;
            e.k = k; e.o = o; e.l = l; e.e = (uint8_t) x; es.push_back(e);
        }
        // DBReader::sortIndex (DBReader.cpp:238-): NOSORT still orders the index by key
        std::stable_sort(es.begin(), es.end(), [](const E &a, const E &b) { return a.k < b.k; });
        for (auto &e : es) {
            key.push_back(e.k); off.push_back(e.o); len.push_back(e.l); ext.push_back(e.e);
            dataSize += e.l; maxSeqLen = std::max<unsigned>(maxSeqLen, e.l); lastKey = std::max(lastKey, e.k);
        }
    }
    size_t size() const { return key.size(); }
    size_t getId(uint32_t k) const {  // DBReader::getId: binary search on the key-sorted index
        auto it = std::lower_bound(key.begin(), key.end(), k);
        if (it == key.end() || *it != k) return (size_t) UINT_MAX;
        return it - key.begin();
    }
    const char *getData(size_t id) const { return data.data() + off[id]; }
    size_t seqLen(size_t id) const { return len[id] >= 2 ? len[id] - 2 : 0; }  // DBReader.h:193-213
    size_t aaDbSize() const { return dataSize - 2 * size(); }                   // DBReader.cpp:540-548
};

struct DbOut {
    std::vector<std::string> payload;  // per slot
    std::vector<uint32_t> key;
    std::vector<uint8_t> ext;
    std::vector<uint8_t> used;
    void init(size_t n) { payload.resize(n); key.resize(n); ext.assign(n, 0); used.assign(n, 0); }
    void set(size_t slot, uint32_t k, const std::string &p, uint8_t e) { payload[slot] = p; key[slot] = k; ext[slot] = e; used[slot] = 1; }
    void write(const std::string &path, int dbtype) {
        FILE *d = fopen(path.c_str(), "wb"), *ix = fopen((path + ".index").c_str(), "w");
        size_t off = 0;
        std::vector<size_t> order;
        for (size_t i = 0; i < payload.size(); i++) if (used[i]) order.push_back(i);
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
        for (size_t i : order) {
            fwrite(payload[i].data(), 1, payload[i].size(), d); fputc(0, d);
            fprintf(ix, "%u\t%zu\t%zu\t%u\n", key[i], off, payload[i].size() + 1, (unsigned) ext[i]);
            off += payload[i].size() + 1;
        }
        fclose(d); fclose(ix);
        FILE *t = fopen((path + ".dbtype").c_str(), "wb"); int32_t v = dbtype; fwrite(&v, 4, 1, t); fclose(t);
    }
};

// ----------------------------------------------------------------------------- alphabets
// M/commons/NucleotideMatrix.cpp:17-61 over lib/mmseqs/data/nucleotide.out (A,C,T,G,X = 0..4)
static unsigned char AA2NUM[256];
static const char NUM2AA[5] = {'A', 'C', 'T', 'G', 'X'};
static const int REVRES[5] = {2, 3, 0, 1, 4};  // NucleotideMatrix.cpp:9-13
static signed char ASCII_SCORE[128][128];       // M/commons/SubstitutionMatrix.h:56-73, scores +2/-3, X row -3
static void initAlphabet() {
    for (int l = 0; l < 256; l++) {
        int u = toupper((char) l);
        switch (u) {
            case 'A': AA2NUM[l] = 0; break;
            case 'C': AA2NUM[l] = 1; break;
            case 'T': AA2NUM[l] = 2; break;
            case 'G': AA2NUM[l] = 3; break;
            case 'U': case 'W': AA2NUM[l] = 2; break;
            case 'K': case 'B': case 'D': case 'V': case 'R': case 'S': AA2NUM[l] = 3; break;
            case 'M': case 'Y': case 'H': AA2NUM[l] = 1; break;
            default: AA2NUM[l] = 4; break;
        }
    }
    // setupLetterMapping loops letter < UCHAR_MAX, so 255 keeps BaseMatrix's default; irrelevant for ASCII input
    for (int i = 0; i < 128; i++)
        for (int j = 0; j < 128; j++) {
            int a = AA2NUM[i], b = AA2NUM[j];
            ASCII_SCORE[i][j] = (a == b && a != 4) ? 2 : -3;
        }
}
// CarpeDeam's own maps: unordered_map<char,int>::operator[] yields 0 for any other char
static inline int nucMap(char c) { return c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0; }  // correction.cpp:170-174
static inline int ryMap(char c) { return (c == 'C' || c == 'T') ? 1 : 0; }                    // correction.cpp:176-180

// src/assembler/nuclassembleUtil.cpp:67-76
static std::string nuclRevFragment(const char *frag, size_t n) {
    std::string r(n, 'N');
    for (size_t i = 0; i < n; i++) {
        char c = NUM2AA[REVRES[AA2NUM[(unsigned char) frag[n - 1 - i]]]];
        r[i] = (c == 'X') ? 'N' : c;
    }
    return r;
}

// ----------------------------------------------------------------------------- text codecs
static char *itoa_u(unsigned long long v, char *p) { char b[24]; int n = 0; do { b[n++] = '0' + v % 10; v /= 10; } while (v); while (n) *p++ = b[--n]; return p; }
static char *itoa_i(long long v, char *p) { if (v < 0) { *p++ = '-'; return itoa_u((unsigned long long) (-v), p); } return itoa_u(v, p); }
// M/commons/Util.cpp:278-307 + the '\t' that overwrites the last written char in resultToBuffer
static char *seqIdToBuf(float seqId, char *p) {
    if (seqId == 1.0) { memcpy(p, "1.00", 4); return p + 4; }
    *p++ = '0'; *p++ = '.';
    if (seqId < 0.10) *p++ = '0';
    if (seqId < 0.01) *p++ = '0';
    return itoa_i((int) (seqId * 1000), p);
}
struct Aln {  // M/alignment/Matcher.h:33-56 (fields the path uses)
    uint32_t dbKey = 0; int score = 0; float seqId = 0; double eval = 0; unsigned alnLength = 0;
    int qStartPos = 0, qEndPos = 0; unsigned qLen = 0; int dbStartPos = 0, dbEndPos = 0; unsigned dbLen = 0;
    bool isRev = false; float rySeqId = 0;
    float deamMatch = 0; unsigned alnLengthCons = 0;   // the fork's additions (Matcher.h:55-56), used by ancient_contig_merge
};
// M/alignment/Matcher.cpp:356-404
static void alnToBuf(std::string &out, const Aln &r) {
    char b[256]; char *p = b;
    p = itoa_u(r.dbKey, p); *p++ = '\t';
    p = itoa_i(r.score, p); *p++ = '\t';
    p = seqIdToBuf(r.seqId, p); *p++ = '\t';
    p += sprintf(p, "%.3E", r.eval); *p++ = '\t';
    p = itoa_i(r.qStartPos, p); *p++ = '\t';
    p = itoa_i(r.qEndPos, p); *p++ = '\t';
    p = itoa_i((int) r.qLen, p); *p++ = '\t';
    p = itoa_i(r.dbStartPos, p); *p++ = '\t';
    p = itoa_i(r.dbEndPos, p); *p++ = '\t';
    p = itoa_i((int) r.dbLen, p); *p++ = '\n';
    out.append(b, p - b);
}
// M/alignment/Matcher.cpp:274-353
static void parseAlns(const char *d, std::vector<Aln> &out) {
    while (*d) {
        Aln r; char *e;
        r.dbKey = strtoul(d, &e, 10); d = e + 1;
        r.score = strtol(d, &e, 10); d = e + 1;
        r.seqId = (float) strtod(d, &e); d = e + 1;
        r.eval = strtod(d, &e); d = e + 1;
        int q0 = strtol(d, &e, 10); d = e + 1; int q1 = strtol(d, &e, 10); d = e + 1; int ql = strtol(d, &e, 10); d = e + 1;
        int t0 = strtol(d, &e, 10); d = e + 1; int t1 = strtol(d, &e, 10); d = e + 1; int tl = strtol(d, &e, 10); d = e;
        r.qStartPos = q0; r.qEndPos = q1; r.qLen = ql; r.dbStartPos = t0; r.dbEndPos = t1; r.dbLen = tl;
        int aq = (q0 == -1) ? 0 : q0, at = (t0 == -1) ? 0 : t0;
        r.alnLength = std::max(abs(q1 - aq), abs(t1 - at)) + 1;  // Matcher.cpp:204-206
        out.push_back(r);
        while (*d && *d != '\n') d++;
        if (*d == '\n') d++;
    }
}
struct Hit { uint32_t seqId; int prefScore; unsigned short diagonal; };  // M/prefiltering/QueryMatcher.h:35-39
static void hitToBuf(std::string &out, const Hit &h) {  // QueryMatcher.h:114-126
    char b[64]; char *p = b;
    p = itoa_u(h.seqId, p); *p++ = '\t'; p = itoa_i(h.prefScore, p); *p++ = '\t'; p = itoa_i((short) h.diagonal, p); *p++ = '\n';
    out.append(b, p - b);
}
static void parseHits(const char *d, std::vector<Hit> &out) {  // QueryMatcher.h:81-102
    while (*d) {
        Hit h; char *e;
        h.seqId = strtoul(d, &e, 10); d = e + 1; h.prefScore = strtol(d, &e, 10); d = e + 1;
        h.diagonal = (unsigned short) (short) strtol(d, &e, 10); d = e;
        out.push_back(h);
        while (*d && *d != '\n') d++;
        if (*d == '\n') d++;
    }
}

// ----------------------------------------------------------------------------- damage model
struct DiNuc { ld p[4][4]; };    // src/assembler/nuclassembleUtil.h:34-36
struct SubRates { ld s[12]; };   // nuclassembleUtil.h:38-40

// nuclassembleUtil.h:53-102 (+ lib/libgab/libgab.h allTokens :516-531, destringify<long double> :611-619 = istream >>)
static void readProf(const std::string &fn, std::vector<SubRates> &v) {
    std::ifstream f(fn);
    std::string line;
    if (!f.good()) { std::cerr << "Profile not 12 fields uniq3\n"; exit(1); }
    auto tokens = [](const std::string &l) { std::vector<std::string> t; std::string cur; for (char c : l) { if (c == '\t') { t.push_back(cur); cur.clear(); } else cur += c; } t.push_back(cur); return t; };
    if (!std::getline(f, line)) { std::cerr << "Unable to open file \n"; exit(1); }
    if (tokens(line).size() != 12) { std::cerr << "Profile not 12 fields uniq1\n"; exit(1); }
    while (std::getline(f, line)) {
        auto t = tokens(line);
        if (t.size() != 12) { std::cerr << "Profile not 12 fields uniq2\n"; exit(1); }
        SubRates r;
        for (int k = 0; k < 12; k++) { std::istringstream in(t[k]); ld x; in >> x; r.s[k] = x; }
        v.push_back(r);
    }
}
// nuclassembleUtil.cpp:821-1007
static void initDeam(const std::string &f5, const std::string &f3, std::vector<DiNuc> &all, std::vector<DiNuc> &rev) {
    std::vector<SubRates> sub5, sub3;
    if (f3 == "3p.prof" && f5 == "5p.prof") {
        for (int r = 0; r < 5; r++) { SubRates z; for (int k = 0; k < 12; k++) z.s[k] = 0.0; sub5.push_back(z); sub3.push_back(z); }
    } else { readProf(f5, sub5); readProf(f3, sub3); }
    float defBoth[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    DiNuc def;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) def.p[i][j] = defBoth[4 * i + j];
    if (!sub5.empty()) { const SubRates &o = sub5.back(); def.p[1][3] = o.s[5]; def.p[1][1] = 1 - o.s[5]; }
    if (!sub3.empty()) { const SubRates &o = sub3.front(); def.p[2][0] = o.s[6]; def.p[2][2] = 1 - o.s[6]; }
    // the reference stores these four defaults in unordered_map<int,double>: they pass through double
    double dCC = def.p[1][1], dCT = def.p[1][3], dGA = def.p[2][0], dGG = def.p[2][2];
    std::vector<DiNuc> sub;
    auto build = [&](const SubRates &o, bool fiveP) {
        DiNuc m; int idx = 0;
        for (int i = 0; i < 4; i++) {
            double sum = 0.0;
            for (int j = 0; j < 4; j++) { if (i == j) continue; m.p[i][j] = o.s[idx]; sum += o.s[idx]; idx++; }
            m.p[i][i] = 1.0 - sum;
        }
        if (fiveP) { m.p[2][0] = dGA; m.p[2][2] = dGG; } else { m.p[1][1] = dCC; m.p[1][3] = dCT; }
        sub.push_back(m);
    };
    for (auto &o : sub5) build(o, true);
    for (auto &o : sub3) build(o, false);
    all.assign(11, def);
    std::copy(sub.begin(), sub.begin() + 5, all.begin());
    std::copy(sub.end() - 5, sub.end(), all.end() - 5);
    rev = all;
    for (size_t i = 0; i < 11; i++) {
        const DiNuc &e = all[10 - i];
        rev[i].p[1][3] = e.p[2][0]; rev[i].p[1][1] = e.p[2][2];
        rev[i].p[2][0] = e.p[1][3]; rev[i].p[2][2] = e.p[1][1];
    }
}
// nuclassembleUtil.cpp:49-65
static void seqErrProf(DiNuc &m, ld err) { for (int o = 0; o < 4; o++) for (int b = 0; b < 4; b++) m.p[o][b] = (o == b) ? 1 - err : err / 3; }

static const double SMOOTHING_VALUE = 0.001;
struct Cnt { int count[4][11]; };  // nuclassembleUtil.h:26-28

// src/assembler/correction.cpp:7-123
static int mostLikeliBaseRead(int baseInQuery, unsigned qIter, const Cnt &deam, const Cnt &revs, const std::vector<DiNuc> &D,
                              const std::vector<DiNuc> &Drev, const DiNuc &seqErr, bool wasCorr, unsigned qLen) {
    ld lik[4] = {0, 0, 0, 0};
    double logQ[4], logT[4];
    unsigned cov[4] = {0, 0, 0, 0};
    for (int b = 0; b < 4; b++) for (size_t l = 0; l < D.size(); l++) cov[b] += deam.count[b][l];
    double ctRatio = static_cast<double>(cov[3]) / (cov[1] + cov[3] + cov[0] + cov[2]);
    double gaRatio = static_cast<double>(cov[0]) / (cov[1] + cov[3] + cov[0] + cov[2]);
    for (unsigned q = 0; q < 4; q++) {
        logT[q] = std::log(seqErr.p[q][baseInQuery]);  // long double log, stored as double
        if (wasCorr) logQ[q] = std::log(seqErr.p[q][baseInQuery]);
        else {
            if (ctRatio >= 0.4 || gaRatio >= 0.4) return baseInQuery;
            double deamination;
            if (qIter < 5) deamination = D[qIter].p[q][baseInQuery];
            else if (qIter >= qLen - 5) deamination = D[D.size() - (qLen - qIter)].p[q][baseInQuery];
            else deamination = D[5].p[q][baseInQuery];
            logQ[q] = std::log(std::max(deamination, SMOOTHING_VALUE));
        }
    }
    for (int qb = 0; qb < 4; qb++) {
        ld s = 0;
        for (int tb = 0; tb < 4; tb++) {
            if (cov[tb] == 0) continue;
            for (unsigned l = 0; l < D.size(); l++) {
                double dp = D[l].p[qb][tb], dr = Drev[l].p[qb][tb];
                dp = std::max(dp, SMOOTHING_VALUE); dr = std::max(dr, SMOOTHING_VALUE);
                int c = deam.count[tb][l], nr = revs.count[tb][l];
                double ldp = std::log(dp), ldr = std::log(dr);
                if (c != 0) {
                    s += (c - nr) * (logT[tb] + logQ[qb] + ldp);
                    s += nr * (logT[tb] + logQ[qb] + ldr);
                }
            }
        }
        lik[qb] = s;
    }
    return (int) (std::max_element(lik, lik + 4) - lik);
}

// nuclassembleUtil.cpp:78-92
static float rySeqIdOf(const Aln &r, const char *q, const char *t) {
    unsigned dist = 0;
    for (unsigned i = 0; i < r.alnLength; i++) if (ryMap(q[r.qStartPos + i]) != ryMap(t[r.dbStartPos + i])) dist++;
    return static_cast<float>(r.alnLength - dist) / static_cast<float>(r.alnLength);
}

struct AncientPar {
    float seqIdThr = 0.9f, randAlnPenal = 0.85f, excessPenal = 0.0625f, corrReadsRySeqId = 0.99f, likelihoodThreshold = 0.5f, rySeqIdThr = 0.99f;
    bool unsafe = false; int minCovSafe = 5; size_t maxSeqLen = 200000; std::string damage; int threads = 1;
    float mergeSeqIdThr = 0.99f;   // --min-merge-seq-id (LocalParameters.h:141,302)
};

// ----------------------------------------------------------------------------- C: ancient_correction
// src/assembler/correction.cpp:128-490
static int doCorrection(const std::string &seqPath, const std::string &alnPath, const std::string &outPath, const AncientPar &par) {
    Db seq, aln; seq.load(seqPath); aln.load(alnPath);
    DbOut out; out.init(seq.size());
    std::vector<DiNuc> D, Drev; initDeam(par.damage + "5p.prof", par.damage + "3p.prof", D, Drev);
    DiNuc seqErr; seqErrProf(seqErr, 0.01L);
#pragma omp parallel num_threads(par.threads)
    {
        std::vector<Aln> alns, reads;
        std::vector<uint8_t> useReverse(seq.size(), 0);
#pragma omp for schedule(dynamic, 100)
        for (size_t id = 0; id < seq.size(); id++) {
            uint32_t qKey = seq.key[id];
            const char *q = seq.getData(id);
            unsigned qLen = seq.seqLen(id);
            alns.clear();
            size_t aid = aln.getId(qKey);
            if (aid != (size_t) UINT_MAX) parseAlns(aln.getData(aid), alns);
            bool qWasExt = seq.ext[id];
            float avCov = 0;
            for (auto &a : alns) {  // :220-244
                size_t tid = seq.getId(a.dbKey);
                if (a.qStartPos > a.qEndPos) {
                    useReverse[tid] = 1; std::swap(a.qStartPos, a.qEndPos);
                    unsigned s = a.dbStartPos; a.dbStartPos = a.dbLen - a.dbEndPos - 1; a.dbEndPos = a.dbLen - s - 1; a.isRev = true;
                } else { useReverse[tid] = 0; a.isRev = false; }
                avCov += a.alnLength;
            }
            avCov = static_cast<float>(avCov) / qLen;
            std::vector<Cnt> deam(qLen), rev(qLen);
            std::vector<unsigned> total(qLen, 0);
            for (unsigned i = 0; i < qLen; i++) { memset(&deam[i], 0, sizeof(Cnt)); memset(&rev[i], 0, sizeof(Cnt)); }
            reads.clear();
            for (auto &a0 : alns) {  // :273-323
                Aln t = a0;
                size_t tid = seq.getId(t.dbKey);
                if (seq.ext[tid]) continue;
                const char *ts = seq.getData(tid); std::string tmp;
                if (t.isRev) { tmp = nuclRevFragment(ts, t.dbLen); ts = tmp.data(); }
                t.rySeqId = rySeqIdOf(t, q, ts);
                float thr = par.corrReadsRySeqId;
                if (t.alnLength <= 100) { thr = (static_cast<float>(t.alnLength) - 1) / static_cast<float>(t.alnLength); thr = std::floor(thr * 1000) / 1000; }
                if (t.rySeqId >= thr && t.dbStartPos == 0 && static_cast<unsigned>(t.qEndPos) == (qLen - 1)) reads.push_back(a0);
                else if (t.rySeqId >= thr && t.qStartPos == 0 && static_cast<unsigned>(t.dbEndPos) == (t.dbLen - 1)) reads.push_back(a0);
                else if (t.rySeqId >= thr && avCov < 50) reads.push_back(a0);
            }
            for (auto &r0 : reads) {  // :328-392
                Aln t = r0;
                size_t tid = seq.getId(t.dbKey);
                const char *ts = seq.getData(tid); unsigned tLen = seq.seqLen(tid); std::string tmp;
                if (useReverse[tid]) { tmp = nuclRevFragment(ts, tLen); ts = tmp.data(); }
                t.rySeqId = rySeqIdOf(t, q, ts);
                float thr = par.corrReadsRySeqId;
                if (t.alnLength <= 100) { thr = (static_cast<float>(t.alnLength) - 1) / static_cast<float>(t.alnLength); thr = std::floor(thr * 1000) / 1000; }
                if (t.rySeqId >= thr && t.seqId >= par.seqIdThr && t.alnLength >= 30) {
                    std::vector<int> idx(t.dbLen);
                    for (size_t i = 0; i < 5; ++i) idx[i] = i;
                    for (size_t i = 5; i < t.dbLen - 5; ++i) idx[i] = 5;
                    for (size_t i = 0; i < 5; ++i) idx[t.dbLen - 5 + i] = 6 + i;
                    for (unsigned p = 0; p < t.alnLength; p++) {
                        int qp = t.qStartPos + p; int tb = nucMap(ts[t.dbStartPos + p]);
                        total[qp] += 1; int cls = idx[t.dbStartPos + p];
                        deam[qp].count[tb][cls] += 1; rev[qp].count[tb][cls] += t.isRev;
                    }
                }
            }
            std::string corr(qLen, 'N');
            for (unsigned p = 0; p < qLen; p++) {  // :397-463
                int qb = nucMap(q[p]);
                if (total[p] <= 1) corr[p] = q[p];
                else corr[p] = "ACGT"[mostLikeliBaseRead(qb, p, deam[p], rev[p], D, Drev, seqErr, qWasExt, qLen)];
            }
            corr.push_back('\n');
            out.set(id, qKey, corr, qWasExt);
        }
    }
    out.write(outPath, seq.dbtype);
    return 0;
}

// ----------------------------------------------------------------------------- E: ancient_read_assemble
struct Scored { Aln r; double sLenNorm = 0, sRatio = 0; };  // nuclassembleUtil.h:42-46
struct CmpScored { bool operator()(const Scored &a, const Scored &b) const { return a.sLenNorm < b.sLenNorm; } };  // :110-121
typedef std::priority_queue<Scored, std::vector<Scored>, CmpScored> Queue;

// nuclassembleUtil.cpp:203-374
static void calcLikelihoodConsensus(Scored &s, const std::string &consensus, unsigned qLen, const char *tSeq, const std::vector<DiNuc> &Dsel,
                                    unsigned maxAln, float randAlnPenal, const DiNuc &seqErr, float excessPenal) {
    std::string tov(tSeq, s.r.dbLen);
    ld likMod = 0.0;
    auto cls = [&](size_t i) -> size_t { if (i < 5) return i; if (i >= s.r.dbLen - 5) return 6 + (i - (s.r.dbLen - 5)); return 5; };
    unsigned alnCount = 0;
    unsigned dbStart = s.r.dbStartPos, dbEnd = s.r.dbEndPos, qStart = s.r.qStartPos, qEnd = s.r.qEndPos;
    const bool rightStart = dbStart == 0 && qEnd == (qLen - 1);
    const bool leftStart = qStart == 0 && dbEnd == (s.r.dbLen - 1);
    auto column = [&](char cq, char ct, unsigned tIdx) {
        alnCount++;
        double lik = 0;
        const DiNuc &tp = Dsel[cls(tIdx - 1)];
        int qb = nucMap(cq), tb = nucMap(ct);
        for (int x = 0; x < 4; x++) {
            double match = std::max(static_cast<ld>(SMOOTHING_VALUE), tp.p[qb][x]);
            ld tErr = seqErr.p[x][tb];
            lik += (tErr * match);
        }
        likMod += log(lik);
    };
    if (leftStart) {
        unsigned offset = s.r.dbLen - s.r.alnLength;
        tov = std::string(qLen - offset, 'N') + tov;
        unsigned tIdx = 0;
        for (unsigned i = 0; i < tov.size(); i++) {
            if (tov[i] != 'N') tIdx++;
            if (!(consensus[i] == 'N' || tov[i] == 'N')) column(consensus[i], tov[i], tIdx);
        }
    } else if (rightStart) {
        unsigned offset = s.r.dbLen - s.r.alnLength;
        tov = tov + std::string(qLen - offset, 'N');
        unsigned tIdx = 0;
        for (unsigned i = 0; i < tov.size(); i++) {
            unsigned ci = consensus.size() - tov.size() + i;
            if (tov[i] != 'N') tIdx++;
            if (!(consensus[ci] == 'N' || tov[i] == 'N')) column(consensus[ci], tov[i], tIdx);
        }
    }
    unsigned excess = maxAln - alnCount;
    // `log` of a float resolves to std::log(float) in the reference (libgab.h: using namespace std): float log, float product
    likMod += (excess * std::log(excessPenal));
    double randAln = maxAln * std::log(randAlnPenal);
    double ratio = 1.0 / (1.0 + std::exp(randAln - likMod));  // long double exp (libgab.h has `using namespace std`)
    s.sLenNorm = likMod; s.sRatio = ratio;
}
// src/assembler/ancientReadsResults.cpp:48-70
static Scored rsPair(const Aln &res, const std::string &consensus, const char *tSeq, unsigned qLen, const std::vector<DiNuc> &D, const std::vector<DiNuc> &Drev,
                     unsigned maxLeft, unsigned maxRight, float rnd, const DiNuc &seqErr, float exc) {
    Scored s; s.r = res;
    unsigned maxOv = maxRight;
    if (unsigned(res.qStartPos) == 0 && unsigned(res.dbEndPos) == (res.dbLen - 1)) maxOv = maxLeft;
    calcLikelihoodConsensus(s, consensus, qLen, tSeq, res.isRev ? Drev : D, maxOv, rnd, seqErr, exc);
    return s;
}
// nuclassembleUtil.cpp:535-567, 570-702
static void consensusCaller(std::string &cons, const std::vector<Aln> &alns, const Db &seq, const char *q, unsigned qLen, uint32_t qKey, const AncientPar &par) {
    if (!par.unsafe) { for (unsigned p = 0; p < qLen; p++) cons[qLen + p] = q[p]; return; }
    std::vector<std::vector<unsigned>> cov(3 * qLen, std::vector<unsigned>(4, 0));
    for (const Aln &res : alns) {
        size_t rid = seq.getId(res.dbKey);
        const bool rightStart = res.dbStartPos == 0 && (res.dbEndPos != static_cast<int>(res.dbLen) - 1);
        const bool leftStart = res.qStartPos == 0 && (res.qEndPos != static_cast<int>(res.qLen) - 1);
        const bool notId = (res.dbKey != qKey);
        unsigned tLen = seq.seqLen(rid);
        if ((rightStart || leftStart) && notId) {
            std::string s;
            if (res.isRev) { s = nuclRevFragment(seq.getData(rid), res.dbLen); s.resize(tLen, '\0'); } else s = std::string(seq.getData(rid), tLen);
            if ((unsigned) res.dbStartPos == 0 && (unsigned) res.qEndPos == (qLen - 1)) {
                for (unsigned p = 0; p < res.dbLen; p++) cov[qLen + res.qStartPos + p][nucMap(s[p])] += 1;
            } else if ((unsigned) res.qStartPos == 0 && (unsigned) res.dbEndPos == (tLen - 1)) {
                for (unsigned p = 0; p < res.dbLen; p++) cov[qLen - (res.dbLen - res.alnLength) + p][nucMap(s[p])] += 1;
            }
        }
    }
    for (size_t i = 0; i < cov.size(); ++i) {  // calculateConsensus
        unsigned tot = cov[i][0] + cov[i][1] + cov[i][2] + cov[i][3];
        if (tot >= (unsigned) par.minCovSafe) {
            unsigned mx = 0; char nuc = 'N'; int cm = 0;
            for (int j = 0; j < 4; ++j) {
                if (cov[i][j] > mx) { mx = cov[i][j]; nuc = "ACGT"[j]; cm = 1; } else if (cov[i][j] == mx && mx > 0) cm++;
            }
            if (cm > 1) nuc = 'N';
            cons[i] = nuc;
        }
    }
    for (unsigned p = 0; p < qLen; p++) cons[qLen + p] = q[p];
}
// nuclassembleUtil.cpp:377-500
static void updateSeqIdConsensusReads(std::vector<Aln> &alns, const Db &seq, const std::string &cons, unsigned qLen, unsigned &maxLeft, unsigned &maxRight) {
    for (Aln &a : alns) {
        size_t tid = seq.getId(a.dbKey); unsigned tLen = seq.seqLen(tid);
        std::string t = a.isRev ? nuclRevFragment(seq.getData(tid), tLen) : std::string(seq.getData(tid), tLen);
        unsigned dbStart = a.dbStartPos, dbEnd = a.dbEndPos, qStart = a.qStartPos, qEnd = a.qEndPos;
        const bool rightStart = dbStart == 0 && qEnd == (qLen - 1);
        const bool leftStart = qStart == 0 && dbEnd == (a.dbLen - 1);
        int idCnt = 0, idRy = 0; unsigned tot = 0;
        if (leftStart) {
            unsigned offset = a.dbLen - a.alnLength;
            t = std::string(qLen - offset, 'N') + t;
            for (unsigned i = 0; i < t.size(); i++)
                if (!(cons[i] == 'N' || t[i] == 'N')) { idCnt += (cons[i] == t[i]); idRy += (ryMap(cons[i]) == ryMap(t[i])); tot++; }
        } else if (rightStart) {
            unsigned offset = a.dbLen - a.alnLength;
            t = t + std::string(qLen - offset, 'N');
            for (unsigned i = 0; i < t.size(); i++) {
                unsigned ci = cons.size() - t.size() + i;
                if (!(cons[ci] == 'N' || t[i] == 'N')) { idCnt += (cons[ci] == t[i]); idRy += (ryMap(cons[ci]) == ryMap(t[i])); tot++; }
            }
        }
        float sid = a.seqId, rid = a.rySeqId;
        if (tot != 0) { sid = static_cast<float>(idCnt) / tot; rid = static_cast<float>(idRy) / tot; }
        if (leftStart && tot > maxLeft) maxLeft = tot; else if (rightStart && tot > maxRight) maxRight = tot;
        a.seqId = sid; a.rySeqId = rid;
    }
}
// M/alignment/DistanceCalculator.h:115-175 (mode 3) + :204-220
struct LocalAln { int startPos = -1, endPos = -1; unsigned score = 0, diagonalLen = 0, distToDiagonal = 0; int diagonal = 0; };
static void globalSub(const char *a, const char *b, unsigned n, LocalAln &r) {
    unsigned first = (a[0] == '*' || b[0] == '*') ? 1 : 0; unsigned last = n - 1;
    if (last > 0 && (a[n - 1] == '*' || b[n - 1] == '*')) last--;
    int64_t sc = 0;
    for (unsigned p = first; p <= last; p++) sc += ASCII_SCORE[(int) a[p]][(int) b[p]];
    sc = std::max(sc, (int64_t) 0);
    r.startPos = first; r.endPos = last; r.score = (unsigned) sc;
}
static LocalAln ungappedByDiagonal(const char *q, unsigned qLen, const char *t, unsigned tLen, int diagonal) {
    unsigned md = abs(diagonal); LocalAln res; res.distToDiagonal = md; res.diagonal = diagonal;
    if (diagonal >= 0 && md < qLen) { unsigned m = std::min(tLen, qLen - md); res.diagonalLen = m; globalSub(q + md, t, m, res); }
    else if (diagonal < 0 && md < tLen) { unsigned m = std::min(tLen - md, qLen); res.diagonalLen = m; globalSub(q, t + md, m, res); }
    return res;
}
// DistanceCalculator.h:93-113
static LocalAln computeUngapped(const char *q, unsigned qLen, const char *t, unsigned tLen, unsigned short diagonal) {
    LocalAln mx;
    for (unsigned d = 1; d <= 1 + tLen / 32768; d++) { int rd = (-(int) d * 65536 + diagonal); LocalAln tmp = ungappedByDiagonal(q, qLen, t, tLen, rd); if (tmp.score > mx.score) mx = tmp; }
    for (unsigned d = 0; d <= qLen / 65536; d++) { int rd = (d * 65536 + diagonal); LocalAln tmp = ungappedByDiagonal(q, qLen, t, tLen, rd); if (tmp.score > mx.score) mx = tmp; }
    return mx;
}
// nuclassembleUtil.cpp:9-47
static void updateNuclAlignment(Aln &a, const LocalAln &al, const char *q, size_t qLen, const char *t, size_t tLen) {
    int qs, qe, ds, de; int diag = al.diagonal; int dist = std::max(abs(diag), 0);
    if (diag >= 0) { qs = al.startPos + dist; qe = al.endPos + dist; ds = al.startPos; de = al.endPos; }
    else { qs = al.startPos; qe = al.endPos; ds = al.startPos + dist; de = al.endPos + dist; }
    int idCnt = 0;
    for (int i = qs; i < qe; i++) idCnt += (q[i] == t[ds + (i - qs)]) ? 1 : 0;
    a.seqId = static_cast<float>(idCnt) / (static_cast<float>(qe) - static_cast<float>(qs));
    a.qLen = qLen; a.dbLen = tLen; a.alnLength = al.diagonalLen;
    float spc = static_cast<float>(al.score) / static_cast<float>(a.alnLength + 0.5);
    a.score = static_cast<int>(spc * 100);
    a.qStartPos = qs; a.qEndPos = qe; a.dbStartPos = ds; a.dbEndPos = de;
}
// ancientReadsResults.cpp:76-92
static bool selectFragment(Queue &qu, uint32_t qKey, Aln &out) {
    while (!qu.empty()) {
        Aln res = qu.top().r; qu.pop();
        const bool notBoth = !(res.dbStartPos == 0 && res.qStartPos == 0);
        const bool rightStart = res.dbStartPos == 0 && (res.dbEndPos != static_cast<int>(res.dbLen) - 1);
        const bool leftStart = res.qStartPos == 0 && (res.qEndPos != static_cast<int>(res.qLen) - 1);
        if ((rightStart || leftStart) && notBoth && res.dbKey != qKey) { out = res; return true; }
    }
    return false;
}
// EvalueComputation.h:22-24 with the ALP gapless parameters of nucleotide.out (see evalue section below)
static double rawScoreFromBitScore(double bits);

// ancientReadsResults.cpp:96-595
static int doAssemble(const std::string &seqPath, const std::string &alnPath, const std::string &outPath, const AncientPar &par) {
    Db seq, aln; seq.load(seqPath); aln.load(alnPath);
    DbOut out; out.init(seq.size());
    std::vector<uint8_t> wasExtended(seq.size(), 0);
    std::vector<DiNuc> D, Drev; initDeam(par.damage + "5p.prof", par.damage + "3p.prof", D, Drev);
    DiNuc seqErr; seqErrProf(seqErr, 0.001L);
    FILE *scoreLog = getenv("ORACLE_SCORES") ? fopen(getenv("ORACLE_SCORES"), "w") : NULL;
#pragma omp parallel num_threads(par.threads)
    {
        std::vector<Aln> alns, notContig, tmpAl;
        std::vector<uint8_t> useReverse(seq.size(), 0);
#pragma omp for schedule(dynamic, 100)
        for (size_t id = 0; id < seq.size(); id++) {
            uint32_t qKey = seq.key[id];
            const char *q = seq.getData(id);
            unsigned qLen = seq.seqLen(id);
            std::string query(q, qLen);
            alns.clear(); notContig.clear();
            size_t aid = aln.getId(qKey);
            if (aid != (size_t) UINT_MAX) parseAlns(aln.getData(aid), alns);
            bool couldExtend = false;
            Queue queue;
            for (auto &a : alns) {  // A :202-244
                unsigned ds = a.dbStartPos, de = a.dbEndPos, qs = a.qStartPos, qe = a.qEndPos;
                const bool rightStart = ds == 0 && qe == (qLen - 1);
                const bool leftStart = qs == 0 && de == (a.dbLen - 1);
                if (!rightStart && !leftStart) continue;
                int raw = static_cast<int>(rawScoreFromBitScore(a.score) + 0.5);
                float spc = static_cast<float>(raw) / static_cast<float>(a.alnLength + 0.5);
                a.score = static_cast<int>(spc * 100);
                size_t tid = seq.getId(a.dbKey);
                if (a.qStartPos > a.qEndPos) {
                    useReverse[tid] = 1; std::swap(a.qStartPos, a.qEndPos);
                    unsigned s = a.dbStartPos; a.dbStartPos = a.dbLen - a.dbEndPos - 1; a.dbEndPos = a.dbLen - s - 1; a.isRev = true;
                } else { useReverse[tid] = 0; a.isRev = false; }
            }
            for (auto &a : alns) {  // B :247-293
                unsigned ds = a.dbStartPos, de = a.dbEndPos, qs = a.qStartPos, qe = a.qEndPos;
                const bool rightStart = ds == 0 && qe == (qLen - 1);
                const bool leftStart = qs == 0 && de == (a.dbLen - 1);
                if (!rightStart && !leftStart) continue;
                size_t tid = seq.getId(a.dbKey); unsigned tLen = seq.seqLen(tid);
                if (tid == qKey) continue;  // id compared with key, as in the reference (:264)
                std::string t = a.isRev ? nuclRevFragment(seq.getData(tid), tLen) : std::string(seq.getData(tid), tLen);
                int idCnt = 0, idRy = 0;
                for (int i = a.qStartPos; i <= a.qEndPos; i++) {
                    char tc = t[a.dbStartPos + (i - a.qStartPos)];
                    idCnt += (q[i] == tc) ? 1 : 0; idRy += (ryMap(q[i]) == ryMap(tc)) ? 1 : 0;
                }
                a.seqId = static_cast<float>(idCnt) / a.alnLength; a.rySeqId = static_cast<float>(idRy) / a.alnLength;
            }
            for (auto &a : alns) {  // C :295-315
                unsigned ds = a.dbStartPos, de = a.dbEndPos, qs = a.qStartPos, qe = a.qEndPos;
                const bool rightStart = ds == 0 && qe == (qLen - 1);
                const bool leftStart = qs == 0 && de == (a.dbLen - 1);
                if (!rightStart && !leftStart) continue;
                bool noOffset = (a.dbLen - a.alnLength) == 0;
                size_t tid = seq.getId(a.dbKey);
                if (seq.ext[tid] == 0 && a.alnLength >= 30 && a.seqId >= par.seqIdThr && !noOffset) notContig.push_back(a);
            }
            unsigned maxLeft = 0, maxRight = 0;
            std::string cons(3 * qLen, 'N');
            consensusCaller(cons, notContig, seq, q, qLen, qKey, par);
            updateSeqIdConsensusReads(notContig, seq, cons, qLen, maxLeft, maxRight);
            for (auto &a : notContig) {  // D :332-369
                size_t tid = seq.getId(a.dbKey); unsigned tLen = seq.seqLen(tid);
                const char *ts = seq.getData(tid); std::string tmp;
                if (a.isRev) { tmp = nuclRevFragment(ts, tLen); ts = tmp.data(); }
                bool notInside = a.dbLen != a.alnLength;
                const bool rightStart = a.dbStartPos == 0, leftStart = a.qStartPos == 0, notId = (a.dbKey != qKey);
                if ((rightStart || leftStart) && notInside && notId && a.rySeqId >= par.rySeqIdThr && a.seqId >= par.seqIdThr) {
                    Scored s = rsPair(a, cons, ts, qLen, D, Drev, maxLeft, maxRight, par.randAlnPenal, seqErr, par.excessPenal);
                    if (scoreLog) {  // test hook: first-round likelihood scores, which the module itself never prints
#pragma omp critical
                        fprintf(scoreLog, "%u\t%u\t%a\t%a\n", qKey, a.dbKey, s.sLenNorm, s.sRatio);
                    }
                    if (getenv("ORACLE_TRACE") && (uint32_t) atoi(getenv("ORACLE_TRACE")) == qKey) fprintf(stderr, "D cand %u q[%d,%d] t[%d,%d] sid %.4f ry %.4f sLen %.6f ratio %.6f maxL %u maxR %u\n", a.dbKey, a.qStartPos, a.qEndPos, a.dbStartPos, a.dbEndPos, a.seqId, a.rySeqId, s.sLenNorm, s.sRatio, maxLeft, maxRight);
                    if (s.sRatio > par.likelihoodThreshold) queue.push(s);
                }
            }
            const char *qp = q;
            while (!queue.empty()) {  // E :374-546
                unsigned leftOff = 0, rightOff = 0;
                tmpAl.clear();
                Aln best; bool brk = false;
                while (selectFragment(queue, qKey, best)) {
                    size_t tid = seq.getId(best.dbKey);
                    const char *ts = seq.getData(tid); unsigned tLen = seq.seqLen(tid);
                    if (best.dbStartPos == 0) { if ((tLen - (best.dbEndPos + 1)) <= rightOff) continue; }
                    else if (best.qStartPos == 0) { if (best.dbStartPos <= static_cast<int>(leftOff)) continue; }
                    __sync_or_and_fetch(&wasExtended[tid], (uint8_t) 0x10);   // (atomic, as the reference does it at :402: other threads set bits of the same byte)
                    if (getenv("ORACLE_TRACE") && (uint32_t) atoi(getenv("ORACLE_TRACE")) == qKey) fprintf(stderr, "E pop %u q[%d,%d] t[%d,%d] lo %u ro %u qLen %u\n", best.dbKey, best.qStartPos, best.qEndPos, best.dbStartPos, best.dbEndPos, leftOff, rightOff, qLen);
                    unsigned ds = best.dbStartPos, de = best.dbEndPos, qs = best.qStartPos, qe = best.qEndPos;
                    if (ds == 0 && qe == (qLen - 1)) {
                        if (rightOff > 0) { tmpAl.push_back(best); continue; }
                        unsigned fragLen = tLen - (de + 1);
                        if (query.size() + fragLen >= par.maxSeqLen) { brk = true; break; }
                        std::string frag = useReverse[tid] ? nuclRevFragment(ts, fragLen) : std::string(ts + de + 1, fragLen);
                        query += frag; rightOff += fragLen;
                    } else if (qs == 0 && de == (tLen - 1)) {
                        if (leftOff > 0) { tmpAl.push_back(best); continue; }
                        unsigned fragLen = ds;
                        if (query.size() + fragLen >= par.maxSeqLen) { brk = true; break; }
                        std::string frag = useReverse[tid] ? nuclRevFragment(ts + (tLen - ds), fragLen) : std::string(ts, fragLen);
                        query = frag + query; leftOff += fragLen;
                    }
                }
                (void) brk;
                if (leftOff > 0 || rightOff > 0) couldExtend = true;
                if (!queue.empty()) break;
                qLen = query.length(); qp = query.c_str();
                for (Aln &a : tmpAl) {  // :484-509
                    size_t tid = seq.getId(a.dbKey); unsigned tLen = seq.seqLen(tid);
                    const char *ts = seq.getData(tid); std::string tmp;
                    if (useReverse[tid]) { tmp = nuclRevFragment(ts, tLen); ts = tmp.data(); }
                    int diag = (a.qStartPos + leftOff) - a.dbStartPos;
                    LocalAln al = ungappedByDiagonal(qp, qLen, ts, tLen, diag);
                    updateNuclAlignment(a, al, qp, qLen, ts, tLen);
                }
                std::string cons2(3 * qLen, 'N');
                consensusCaller(cons2, tmpAl, seq, qp, qLen, qKey, par);
                updateSeqIdConsensusReads(tmpAl, seq, cons2, qLen, maxLeft, maxRight);
                for (Aln &a : tmpAl) {  // :515-545
                    bool notInside = a.dbLen != a.alnLength;
                    const bool rightStart = a.dbStartPos == 0, leftStart = a.qStartPos == 0, notId = (a.dbKey != qKey);
                    if (a.seqId >= par.seqIdThr && (rightStart || leftStart) && notId && notInside) {
                        size_t tid = seq.getId(a.dbKey); const char *ts = seq.getData(tid); std::string tmp;
                        if (a.isRev) { tmp = nuclRevFragment(ts, a.dbLen); ts = tmp.data(); }
                        Scored s = rsPair(a, cons2, ts, qLen, D, Drev, maxLeft, maxRight, par.randAlnPenal, seqErr, par.excessPenal);
                        if (getenv("ORACLE_TRACE") && (uint32_t) atoi(getenv("ORACLE_TRACE")) == qKey) fprintf(stderr, "R cand %u q[%d,%d] t[%d,%d] sid %.4f sLen %.6f ratio %.6f\n", a.dbKey, a.qStartPos, a.qEndPos, a.dbStartPos, a.dbEndPos, a.seqId, s.sLenNorm, s.sRatio);
                        if (s.sRatio > par.likelihoodThreshold) queue.push(s);
                    }
                }
            }
            if (couldExtend) { query.push_back('\n'); __sync_or_and_fetch(&wasExtended[id], (uint8_t) 0x20); out.set(id, qKey, query, 1); }
        }
    }
    if (scoreLog) fclose(scoreLog);
    for (size_t id = 0; id < seq.size(); id++)  // :564-581
        if (!(wasExtended[id] & 0x20)) out.set(id, seq.key[id], std::string(seq.getData(id), seq.len[id] - 1), seq.ext[id]);
    out.write(outPath, seq.dbtype);
    return 0;
}

// ----------------------------------------------------------------------------- F4: cyclecheck
// src/assembler/cyclecheck.cpp:30-269.  Kept literally: the k-mer at sequence position 0 is filed under "back" (getCurrentPosition() is -1
// before the first nextKmer() and is compared as unsigned, :117-140); only the FIRST front / middle occurrence of a k-mer is matched
// (:167-169, :190-192); the index is the arithmetic sum of letter * 4^i with N = 4 (Indexer over alphabetSize-1, :88), so k-mers with N can
// collide with others; the band test runs in the reference's types (unsigned / size_t / float / double, :216-238).
static int doCycleCheck(const std::string &seqPath, const std::string &outPath, size_t maxSeqLen, bool chopCycle) {
    Db seq; seq.load(seqPath);
    DbOut out; out.init(seq.size());
    const size_t kmerSize = 22;
    struct KP { size_t kmer; unsigned pos; };
    auto byKmer = [](const KP &a, const KP &b) { if (a.kmer < b.kmer) return true; if (b.kmer < a.kmer) return false; return a.pos < b.pos; };
    std::vector<KP> front, middle, back; std::vector<unsigned> diagHits;
    for (size_t id = 0; id < seq.size(); id++) {
        const char *nucl = seq.getData(id);
        unsigned seqLen = seq.seqLen(id);
        if (seqLen >= maxSeqLen) continue;                                         // :107-112 (a warning, the entry is skipped)
        front.clear(); middle.clear(); back.clear();
        unsigned thirdSeqLen = seqLen / 3;
        for (int cur = -1; (size_t) ((cur + 1) + (int) kmerSize) <= (size_t) seqLen; ) {   // Sequence::hasNextKmer / nextKmer
            unsigned pos = (unsigned) cur;                                          // getCurrentPosition() BEFORE nextKmer()
            cur++;
            size_t idx = 0, pw = 1;
            for (size_t i = 0; i < kmerSize; i++) { idx += (size_t) AA2NUM[(unsigned char) nucl[cur + i]] * pw; pw *= 4; }
            KP e = {idx, (unsigned) cur};
            if (pos < thirdSeqLen + 1) front.push_back(e); else if (pos < 2 * thirdSeqLen + 1) middle.push_back(e); else back.push_back(e);
        }
        std::sort(front.begin(), front.end(), byKmer); std::sort(middle.begin(), middle.end(), byKmer); std::sort(back.begin(), back.end(), byKmer);
        unsigned kmermatches = 0;
        diagHits.assign(2 * (size_t) thirdSeqLen + 1, 0);
        size_t i = 0, j = 0, k = 0;
        while (i < front.size() && (j < back.size() || k < middle.size())) {       // :152-193
            size_t km = front[i].kmer; unsigned pos = front[i].pos;
            while (j < back.size() && back[j].kmer < km) j++;
            while (k < middle.size() && middle[k].kmer < km) k++;
            while (j < back.size() && km == back[j].kmer) { int diag = back[j].pos - pos; if (diag >= static_cast<int>(seqLen / 3)) { diagHits[diag - seqLen / 3]++; kmermatches++; } j++; }
            while (k < middle.size() && km == middle[k].kmer) { int diag = middle[k].pos - pos; if (diag >= static_cast<int>(seqLen / 3)) { diagHits[diag - seqLen / 3]++; kmermatches++; } k++; }
            i++;
            while (i < front.size() && km == front[i].kmer) i++;
        }
        j = 0; k = 0;
        while (k < middle.size() && j < back.size()) {                             // :196-220
            if (middle[k].kmer < back[j].kmer) k++;
            else if (middle[k].kmer > back[j].kmer) j++;
            else {
                size_t km = middle[k].kmer; unsigned pos = middle[k].pos;
                while (j < back.size() && km == back[j].kmer) { int diag = back[j].pos - pos; if (diag >= static_cast<int>(seqLen / 3)) { diagHits[diag - seqLen / 3]++; kmermatches++; } j++; }
                while (k < middle.size() && km == middle[k].kmer) k++;
            }
        }
        unsigned splitDiagonal = 0;
        if (kmermatches > 0) {
            for (unsigned d = 0; d < 2 * thirdSeqLen; d++) {                       // :225-245
                if (diagHits[d] == 0) continue;
                unsigned diag = d + thirdSeqLen, diaglen = seqLen - diag;
                unsigned gapwindow = diaglen * 0.01;
                unsigned lower = std::max(0, static_cast<int>(d - gapwindow));
                unsigned upper = std::min(d + gapwindow, 2 * thirdSeqLen);
                unsigned band = 0;
                for (size_t x = lower; x <= upper; x++) if (diagHits[x] <= diagHits[d]) band += diagHits[x];
                float rate = static_cast<float>(band) / (diaglen - kmerSize + 1);
                if (rate > 0.24) { splitDiagonal = diag; break; }
            }
        }
        if (splitDiagonal != 0) {
            std::string payload = chopCycle ? std::string(nucl, splitDiagonal) + "\n" : std::string(nucl, seq.len[id] - 1);
            out.set(id, seq.key[id], payload, 0);
        }
    }
    out.write(outPath, 1);
    return 0;
}

// ----------------------------------------------------------------------------- F1: ancient_contig_merge
// src/assembler/ancientContigsResults.cpp:25-70.  NOT a strict weak ordering (the fall-through `return true`); the queue below is
// libstdc++'s std::priority_queue with this very comparator, as in the reference.  Overload resolution as there (libgab.h has
// `using namespace std`): lgamma and log of float arguments are the float functions.
struct CmpContigs {
    bool operator()(const Aln &r1, const Aln &r2) const {
        float mm_count1 = r1.alnLengthCons - r1.deamMatch;
        float mm_count2 = r2.alnLengthCons - r2.deamMatch;
        float alpha1 = mm_count1 + 1;
        float alpha2 = mm_count2 + 1;
        float beta1 = r1.deamMatch + 1;
        float beta2 = r2.deamMatch + 1;
        double log_c = (std::lgamma(beta1 + beta2) + std::lgamma(alpha1 + beta1)) - (std::lgamma(alpha1 + beta1 + beta2) + std::lgamma(beta1));
        double log_r = 0.0;
        double p = 0.0;
        for (size_t idx = 0; idx < alpha2; idx++) {
            p += std::exp(log_r + log_c);
            log_r = std::log(alpha1 + idx) + std::log(beta2 + idx) - (std::log(idx + 1) + std::log(idx + alpha1 + beta1 + beta2)) + log_r;
        }
        if (p < 0.45) return true;
        if (p > 0.55) return false;
        if (r1.alnLengthCons < r2.alnLengthCons) return true;
        if (r1.alnLengthCons > r2.alnLengthCons) return false;
        return true;
    }
};
typedef std::priority_queue<Aln, std::vector<Aln>, CmpContigs> QueueContigs;
// :73-91
static bool selectFragmentContigs(QueueContigs &qu, uint32_t qKey, Aln &out) {
    while (!qu.empty()) {
        Aln res = qu.top(); qu.pop();
        const bool notBoth = !(res.dbStartPos == 0 && res.qStartPos == 0);
        const bool rightStart = res.dbStartPos == 0 && (res.dbEndPos != static_cast<int>(res.dbLen) - 1);
        const bool leftStart = res.qStartPos == 0 && (res.qEndPos != static_cast<int>(res.qLen) - 1);
        if ((rightStart || leftStart) && notBoth && res.dbKey != qKey) { out = res; return true; }
    }
    return false;
}
// nuclassembleUtil.cpp:705-790 (updateSeqIdConsensus: like the reads flavour, plus alnLengthCons)
static void updateSeqIdConsensus(std::vector<Aln> &alns, const Db &seq, const std::string &cons, unsigned qLen) {
    for (Aln &a : alns) {
        size_t tid = seq.getId(a.dbKey); unsigned tLen = seq.seqLen(tid);
        std::string t = a.isRev ? nuclRevFragment(seq.getData(tid), tLen) : std::string(seq.getData(tid), tLen);
        unsigned dbStart = a.dbStartPos, dbEnd = a.dbEndPos, qStart = a.qStartPos, qEnd = a.qEndPos;
        const bool rightStart = dbStart == 0 && qEnd == (qLen - 1);
        const bool leftStart = qStart == 0 && dbEnd == (a.dbLen - 1);
        int idCnt = 0, idRy = 0, tot = 0;
        if (leftStart) {
            unsigned offset = a.dbLen - a.alnLength;
            t = std::string(qLen - offset, 'N') + t;
            for (unsigned i = 0; i < t.size(); i++)
                if (!(cons[i] == 'N' || t[i] == 'N')) { idCnt += (cons[i] == t[i]); idRy += (ryMap(cons[i]) == ryMap(t[i])); tot++; }
        } else if (rightStart) {
            unsigned offset = a.dbLen - a.alnLength;
            t = t + std::string(qLen - offset, 'N');
            for (unsigned i = 0; i < t.size(); i++) {
                unsigned ci = cons.size() - t.size() + i;
                if (!(cons[ci] == 'N' || t[i] == 'N')) { idCnt += (cons[ci] == t[i]); idRy += (ryMap(cons[ci]) == ryMap(t[i])); tot++; }
            }
        }
        float sid = a.seqId, rid = a.rySeqId;
        if (tot != 0) { sid = static_cast<float>(idCnt) / tot; rid = static_cast<float>(idRy) / tot; }
        a.seqId = sid; a.alnLengthCons = tot; a.rySeqId = rid;
    }
}
// nuclassembleUtil.cpp:1009-1044
static double deamMatches(const Aln &res, unsigned scoreAln, double matchLik) {
    const double logAdjustmentConstant = std::log(1.4e-9);
    unsigned maxLength = 1e5;
    auto logPower = [logAdjustmentConstant](unsigned length) { return logAdjustmentConstant - 3.0 * std::log(length); };
    double logMin = logPower(10);
    double logMax = logPower(maxLength);
    double logLength = logPower(std::min(res.alnLength, maxLength));
    double fractionLength = (static_cast<double>(std::abs(logLength) - std::abs(logMax))) / static_cast<double>((std::abs(logMin) - std::abs(logMax)));
    double priorAln = 1 - fractionLength;
    double pMatch = 0.5f * ((((static_cast<double>(scoreAln) + 3.0f * res.alnLength) / 5.0f) + 0.9f) / (res.alnLength + 1)) + 0.5f * priorAln;
    double LikNoMatch = 1 - pMatch;
    double oddsRatio = LikNoMatch / matchLik;
    double odds = (1 - pMatch) / pMatch;
    double posterior = 1 / (1 + oddsRatio * odds);
    return posterior;
}
// nuclassembleUtil.cpp:1047-1181
static float ancientMatchCount(const Aln &res, const std::string &cons, unsigned qLen, const std::vector<DiNuc> &Dsel, const Db &seq) {
    float mCT = 0, mGA = 0;
    unsigned mmCons = (1 - res.seqId) * res.alnLengthCons + 0.5;
    unsigned mCons = res.alnLengthCons - mmCons;
    unsigned scoreAln = mCons * 2 + mmCons * (-3);
    size_t tid = seq.getId(res.dbKey); unsigned tLen = seq.seqLen(tid);
    std::string t = res.isRev ? nuclRevFragment(seq.getData(tid), tLen) : std::string(seq.getData(tid), tLen);
    unsigned dbStart = res.dbStartPos, dbEnd = res.dbEndPos, qStart = res.qStartPos, qEnd = res.qEndPos;
    const bool rightStart = dbStart == 0 && qEnd == (qLen - 1);
    const bool leftStart = qStart == 0 && dbEnd == (res.dbLen - 1);
    auto column = [&](char cq, char ct) {
        int qBase = nucMap(cq), tBase = nucMap(ct);
        bool dimerCT = ((4 * qBase + tBase) == 7), dimerGA = ((4 * qBase + tBase) == 8);
        double matchLik = Dsel[5].p[qBase][tBase];
        if (dimerCT && matchLik > 0) mCT += deamMatches(res, scoreAln, matchLik);
        else if (dimerGA && matchLik > 0) mGA += deamMatches(res, scoreAln, matchLik);
    };
    if (leftStart) {
        unsigned offset = res.dbLen - res.alnLength;
        t = std::string(qLen - offset, 'N') + t;
        for (unsigned i = 0; i < t.size(); i++) if (!(cons[i] == 'N' || t[i] == 'N')) column(cons[i], t[i]);
    } else if (rightStart) {
        unsigned offset = res.dbLen - res.alnLength;
        t = t + std::string(qLen - offset, 'N');
        for (unsigned i = 0; i < t.size(); i++) { unsigned ci = cons.size() - t.size() + i; if (!(cons[ci] == 'N' || t[i] == 'N')) column(cons[ci], t[i]); }
    }
    float matches = ((static_cast<float>(scoreAln) + 3.0f * res.alnLengthCons) / 5.0f) + mCT + mGA;
    return matches;
}
// nuclassembleUtil.cpp:78-92
static float getRYSeqId(const Aln &res, const char *q, const char *t) {
    int idRy = 0;
    for (int i = res.qStartPos; i <= res.qEndPos; i++) idRy += (ryMap(q[i]) == ryMap(t[res.dbStartPos + (i - res.qStartPos)])) ? 1 : 0;
    return static_cast<float>(idRy) / res.alnLength;
}
// src/assembler/ancientContigsResults.cpp:94-496
static int doContigMerge(const std::string &seqPath, const std::string &alnPath, const std::string &outPath, const AncientPar &par) {
    Db seq, aln; seq.load(seqPath); aln.load(alnPath);
    DbOut out; out.init(seq.size());
    std::vector<uint8_t> wasExtended(seq.size(), 0);
    std::vector<DiNuc> D, Drev; initDeam(par.damage + "5p.prof", par.damage + "3p.prof", D, Drev);
#pragma omp parallel num_threads(par.threads)
    {
        std::vector<Aln> alns, contigs, tmpAl;
        std::vector<uint8_t> useReverse(seq.size(), 0);
#pragma omp for schedule(dynamic, 100)
        for (size_t id = 0; id < seq.size(); id++) {
            uint32_t qKey = seq.key[id];
            const char *q = seq.getData(id);
            unsigned qLen = seq.seqLen(id);
            std::string query(q, qLen);
            alns.clear(); contigs.clear();
            size_t aid = aln.getId(qKey);
            if (aid != (size_t) UINT_MAX) parseAlns(aln.getData(aid), alns);
            bool couldExtend = false;
            QueueContigs queue;
            for (Aln &a : alns) {  // :187-235
                size_t tid = seq.getId(a.dbKey); unsigned tLen = seq.seqLen(tid);
                std::string t;
                if (a.qStartPos > a.qEndPos) {
                    useReverse[tid] = 1; a.isRev = true;
                    std::swap(a.qStartPos, a.qEndPos);
                    unsigned s0 = a.dbStartPos; a.dbStartPos = a.dbLen - a.dbEndPos - 1; a.dbEndPos = a.dbLen - s0 - 1;
                    t = nuclRevFragment(seq.getData(tid), tLen);
                } else { t = std::string(seq.getData(tid), tLen); useReverse[tid] = 0; a.isRev = false; }
                int idCnt = 0, idRy = 0;
                for (int i = a.qStartPos; i <= a.qEndPos; i++) {
                    char tc = t[a.dbStartPos + (i - a.qStartPos)];
                    idCnt += (q[i] == tc) ? 1 : 0; idRy += (ryMap(q[i]) == ryMap(tc)) ? 1 : 0;
                }
                a.seqId = static_cast<float>(idCnt) / a.alnLength; a.rySeqId = static_cast<float>(idRy) / a.alnLength;
                if (a.seqId >= par.mergeSeqIdThr && a.rySeqId >= par.rySeqIdThr && qKey != a.dbKey) contigs.push_back(a);
            }
            std::string cons(3 * qLen, 'N');
            consensusCaller(cons, contigs, seq, q, qLen, qKey, par);
            updateSeqIdConsensus(contigs, seq, cons, qLen);
            for (Aln &c : contigs) {  // :249-270
                unsigned minAlnLen = 500;
                minAlnLen = (c.alnLength < minAlnLen) ? std::min(minAlnLen, static_cast<unsigned>(0.2 * c.dbLen)) : minAlnLen;
                if (c.seqId >= par.mergeSeqIdThr && c.rySeqId >= par.rySeqIdThr && c.alnLength >= minAlnLen) {
                    c.deamMatch = ancientMatchCount(c, cons, qLen, c.isRev ? Drev : D, seq);
                    queue.push(c);
                }
            }
            const char *qp = q;
            while (!queue.empty()) {  // :276-470
                unsigned leftOff = 0, rightOff = 0;
                tmpAl.clear();
                Aln best;
                while (selectFragmentContigs(queue, qKey, best)) {
                    size_t tid = seq.getId(best.dbKey);
                    const char *ts = seq.getData(tid); unsigned tLen = seq.seqLen(tid);
                    if (best.dbStartPos == 0) { if ((tLen - (best.dbEndPos + 1)) <= rightOff) continue; }
                    else if (best.qStartPos == 0) { if (best.dbStartPos <= static_cast<int>(leftOff)) continue; }
                    unsigned ds = best.dbStartPos, de = best.dbEndPos, qs = best.qStartPos, qe = best.qEndPos;
                    if (ds == 0 && qe == (qLen - 1)) {
                        if (rightOff > 0) { tmpAl.push_back(best); continue; }
                        unsigned fragLen = tLen - (de + 1);
                        if (query.size() + fragLen >= par.maxSeqLen) break;
                        std::string frag = useReverse[tid] ? nuclRevFragment(ts, fragLen) : std::string(ts + de + 1, fragLen);
                        query += frag; rightOff += fragLen;
                    } else if (qs == 0 && de == (tLen - 1)) {
                        if (leftOff > 0) { tmpAl.push_back(best); continue; }
                        unsigned fragLen = ds;
                        if (query.size() + fragLen >= par.maxSeqLen) break;
                        std::string frag = useReverse[tid] ? nuclRevFragment(ts + (tLen - ds), fragLen) : std::string(ts, fragLen);
                        query = frag + query; leftOff += fragLen;
                    }
                }
                if (leftOff > 0 || rightOff > 0) couldExtend = true;
                if (!queue.empty()) break;
                qLen = query.length(); qp = query.c_str();
                for (Aln &a : tmpAl) {  // :404-455
                    size_t tid = seq.getId(a.dbKey); unsigned tLen = seq.seqLen(tid);
                    const char *ts = seq.getData(tid); std::string tmp;
                    if (useReverse[tid]) { tmp = nuclRevFragment(ts, tLen); ts = tmp.data(); }
                    int diag = (a.qStartPos + leftOff) - a.dbStartPos;
                    LocalAln al = ungappedByDiagonal(qp, qLen, ts, tLen, diag);
                    updateNuclAlignment(a, al, qp, qLen, ts, tLen);
                    a.rySeqId = getRYSeqId(a, qp, ts);
                    if (a.seqId >= par.mergeSeqIdThr && a.rySeqId >= par.rySeqIdThr) queue.push(a);
                }
            }
            if (couldExtend) { query.push_back('\n'); __sync_or_and_fetch(&wasExtended[id], (uint8_t) 0x20); out.set(id, qKey, query, 1); }
        }
    }
    for (size_t id = 0; id < seq.size(); id++)  // :473-487
        if (!(wasExtended[id] & 0x20)) out.set(id, seq.key[id], std::string(seq.getData(id), seq.len[id] - 1), seq.ext[id]);
    out.write(outPath, seq.dbtype);
    return 0;
}

// ----------------------------------------------------------------------------- E-value (R3)
// M/alignment/EvalueComputation.h:18-24,36-40,125-128 -> lib/mmseqs/lib/alp/sls_alignment_evaluer.cpp:989-1025,
// sls_pvalues.cpp:366-541, sls_basic.hpp:195-198.  The gapless Gumbel parameters ALP derives for
// nucleotide.out (initGapless) are constants of the matrix; the values below are the reference's own
// (probe `alp` of oracle/_ref, hex-exact) and are pinned by tests/golden/evalue.tsv.
struct Gumbel { double lambda, K, aI, aJ, alphaI, alphaJ, sigma, bI, bJ, betaI, betaJ, tau; };
static Gumbel G = {0x1.4478764a1b24ap-1, 0x1.a1c1e68ea2ab1p-2, 0x1.639ba57df0ecdp-1, 0x1.639ba57df0ecdp-1,
                   0x1.aaaae7ad40e75p-1, 0x1.aaaae7ad40e75p-1, 0x1.aaaae7ad40e75p-1, 0, 0, 0, 0, 0};
static double alpArea(double y, double m, double n) {  // sls_pvalues.cpp:366-541 (blast_=false, compute_only_area)
    const double pi = 3.1415926535897932384626433832795;
    const double const_val = 1 / sqrt(2.0 * pi);
    const double viThr = std::max(2.0 * G.alphaI / G.lambda, 0.0), vjThr = std::max(2.0 * G.alphaJ / G.lambda, 0.0), cThr = std::max(2.0 * G.sigma / G.lambda, 0.0);
    auto normal = [](double x) { return 0.5 * erfc(-sqrt(0.5) * x); };
    double tmp = G.aI * y + G.bI;
    double m_li_y = m - tmp;
    double vi_y = std::max(viThr, G.alphaI * y + G.betaI);
    double sqrt_vi_y = sqrt(vi_y);
    double m_F = (sqrt_vi_y == 0.0) ? 1e100 : m_li_y / sqrt_vi_y;
    double P_m_F = normal(m_F);
    double E_m_F = -const_val * exp(-0.5 * m_F * m_F);
    double m_li_y_P_m_F = m_li_y * P_m_F;
    double sqrt_vi_y_E_m_F = sqrt_vi_y * E_m_F;
    double p1 = m_li_y_P_m_F - sqrt_vi_y_E_m_F;
    tmp = G.aJ * y + G.bJ;
    double n_lj_y = n - tmp;
    double vj_y = std::max(vjThr, G.alphaJ * y + G.betaJ);
    double sqrt_vj_y = sqrt(vj_y);
    double n_F = (sqrt_vj_y == 0.0) ? 1e100 : n_lj_y / sqrt_vj_y;
    double P_n_F = normal(n_F);
    double E_n_F = -const_val * exp(-0.5 * n_F * n_F);
    double n_lj_y_P_n_F = n_lj_y * P_n_F;
    double sqrt_vj_y_E_n_F = sqrt_vj_y * E_n_F;
    double p2 = n_lj_y_P_n_F - sqrt_vj_y_E_n_F;
    double c_y = std::max(cThr, G.sigma * y + G.tau);
    double P_m_F_P_n_F = P_m_F * P_n_F;
    double c_y_P = c_y * P_m_F_P_n_F;
    double p1_p2 = p1 * p2;
    return p1_p2 + c_y_P;
}
// AlignmentEvaluer::area(score, seqlen1, seqlen2) passes (y, m=seqlen2, n=seqlen1) (sls_alignment_evaluer.cpp:1010-1014)
static double computeEvalue(double score, double qLen, size_t dbRes) { double epa = G.K * exp(-G.lambda * score); return epa * alpArea(score, (double) dbRes, qLen); }
static const double LOGK = log(G.K);
static double computeBitScore(double score) { return std::fma(G.lambda, score, -LOGK) / log(2.0); }  // the reference build contracts this into an FMA (-O3 -mfma, GCC default -ffp-contract=fast)
static double rawScoreFromBitScore(double bits) { return (log(G.K) + bits * std::log(2.0)) / G.lambda; }

// ----------------------------------------------------------------------------- R: rescorediagonal
struct RescorePar { float seqIdThr = 0.9f, covThr = 0.0f; double evalThr = 0.001; int covMode = 1, seqIdMode = 0, alnLenThr = 0; int threads = 1; int rescoreMode = 3; bool wrapped = false; };
static bool canBeCovered(float covThr, int covMode, float ql, float tl) {  // M/commons/Util.cpp:533-550
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
static bool hasCoverage(float covThr, int covMode, float qc, float tc) {  // Util.cpp:552-567
    switch (covMode) { case 0: return qc >= covThr && tc >= covThr; case 2: return qc >= covThr; case 1: return tc >= covThr; default: return true; }
}
static float computeCov(unsigned s, unsigned e, unsigned len) { return (std::min(len, std::max(s, e)) - std::min(s, e) + 1) / (float) len; }  // M/alignment/StripedSmithWaterman.cpp:1055-1057
static float computeSeqId(int mode, int ids, int qLen, int tLen, int alnLen) {  // Util.cpp:588-598
    switch (mode) { case 1: return static_cast<float>(ids) / static_cast<float>(std::min(qLen, tLen)); case 2: return static_cast<float>(ids) / static_cast<float>(std::max(qLen, tLen)); case 0: return static_cast<float>(ids) / static_cast<float>(alnLen); }
    return 0.0;
}
// M/alignment/rescorediagonal.cpp:45-379 (rescore mode 3, same query and target DB, no wrapped scoring, no --filter-hits)
static int doRescore(const std::string &qPath, const std::string &tPath, const std::string &prefPath, const std::string &outPath, const RescorePar &par) {
    Db seq, pref; seq.load(tPath); pref.load(prefPath);
    if (qPath != tPath) { fprintf(stderr, "oracle: rescorediagonal expects queryDB == targetDB on this path\n"); return 1; }
    const bool revPref = (pref.dbtype & 0x7FFFFFFF) == 14;
    const size_t dbRes = seq.aaDbSize();
    DbOut out; out.init(pref.size());
#pragma omp parallel num_threads(par.threads)
    {
        std::vector<Hit> hits; std::string buf, qRev;
#pragma omp for schedule(dynamic, 1)
        for (size_t id = 0; id < pref.size(); id++) {
            const char *data = pref.getData(id); uint32_t qKey = pref.key[id];
            size_t qId = UINT_MAX; const char *q = NULL; int qLen = -1;
            if (*data) {
                qId = seq.getId(qKey); q = seq.getData(qId); qLen = seq.seqLen(qId);
                if (revPref) { qRev.assign(qLen, 'X'); for (int p = qLen - 1; p > -1; p--) qRev[(qLen - 1) - p] = NUM2AA[REVRES[AA2NUM[(unsigned char) q[p]]]]; }
            }
            hits.clear(); parseHits(data, hits); buf.clear();
            for (const Hit &h : hits) {
                const char *qa = q; bool isReverse = false;
                if (revPref && h.prefScore < 0) { qa = qRev.data(); isReverse = true; }
                size_t tId = seq.getId(h.seqId);
                const bool isIdentity = (qId == tId);
                const char *t = seq.getData(tId); int dbLen = seq.seqLen(tId);
                if (!canBeCovered(par.covThr, par.covMode, (float) qLen, (float) dbLen)) continue;
                LocalAln al = computeUngapped(qa, qLen, t, dbLen, h.diagonal);
                unsigned dist = al.distToDiagonal; int diagLen = al.diagonalLen; int distance = al.score; int diagonal = al.diagonal;
                double seqId = 0; int alnLen = 0;
                double evalue = computeEvalue(distance, qLen, dbRes);
                int bitScore = static_cast<int>(computeBitScore(distance) + 0.5);
                alnLen = (al.endPos - al.startPos) + 1;
                int qs, qe, ds, de;
                if (diagonal >= 0) { qs = al.startPos + dist; qe = al.endPos + dist; ds = al.startPos; de = al.endPos; }
                else { qs = al.startPos; qe = al.endPos; ds = al.startPos + dist; de = al.endPos + dist; }
                if (evalue <= par.evalThr || isIdentity) {
                    int idCnt = 0;
                    for (int i = qs; i <= qe; i++) { char ql = qa[i] & (unsigned char) (~0x20), tl = t[ds + (i - qs)] & (unsigned char) (~0x20); idCnt += (ql == tl) ? 1 : 0; }
                    seqId = computeSeqId(par.seqIdMode, idCnt, qLen, dbLen, alnLen);
                }
                float queryCov = computeCov(qs, qe, qLen), targetCov = computeCov(ds, de, dbLen);
                if (isReverse) { qs = qLen - qs - 1; qe = qLen - qe - 1; }
                Aln r; r.dbKey = h.seqId; r.score = bitScore; r.seqId = seqId; r.eval = evalue; r.alnLength = alnLen;
                r.qStartPos = qs; r.qEndPos = qe; r.qLen = qLen; r.dbStartPos = ds; r.dbEndPos = de; r.dbLen = dbLen;
                (void) diagLen;
                bool hasCov = hasCoverage(par.covThr, par.covMode, queryCov, targetCov);
                bool hasSeqId = seqId >= (par.seqIdThr - std::numeric_limits<float>::epsilon());
                bool hasEvalue = (evalue <= par.evalThr);
                bool hasAlnLen = (alnLen >= par.alnLenThr);
                if (isIdentity || (hasAlnLen && hasCov && hasSeqId && hasEvalue)) alnToBuf(buf, r);
            }
            out.set(id, qKey, buf, 0);
        }
    }
    out.write(outPath, 5);
    return 0;
}

// M/alignment/rescorediagonal.cpp:45-379 in the mode linclust's pre-clustering runs it (lib/mmseqs/data/workflow/linclust.sh:27-31,
// src/workflow/GuidedNuclassembler.cpp:176-181): --rescore-mode 0 (Hamming: the score is the number of identical LETTERS on the
// diagonal) with --wrapped-scoring 1 (the query is doubled, so a rotation of a circular contig still lines up).  Output = prefilter
// records (key, 100 * seq. id. with the hit's strand as its sign, diagonal), dbtype of the input.
static unsigned inverseHamming(const char *a, const char *b, unsigned n) { unsigned c = 0; for (unsigned i = 0; i < n; i++) c += (a[i] == b[i]); return c; }  // DistanceCalculator.h:276-296
// DistanceCalculator.h:57-91; the loop conditions are unsigned arithmetic there and are kept as written
static LocalAln wrappedHamming(const char *q2, unsigned q2Len, const char *t, unsigned tLen, unsigned short diagonal) {
    LocalAln max; max.startPos = -1; max.endPos = -1; max.score = 0; max.diagonalLen = 0; max.distToDiagonal = 0; max.diagonal = 0;
    const unsigned half = q2Len / 2;
    auto probe = [&](int realDiagonal) {       // ungappedAlignmentByDiagonal(querySeq + realDiagonal, half, dbSeq, dbSeqLen, 0, ...), :116-128
        LocalAln tmp; tmp.startPos = -1; tmp.endPos = -1; tmp.distToDiagonal = 0; tmp.diagonal = 0; tmp.score = 0; tmp.diagonalLen = 0;
        if (0 < half) { const unsigned m = std::min(tLen, half); tmp.diagonalLen = m; tmp.score = (int) inverseHamming(q2 + realDiagonal, t, m); }
        tmp.diagonal += realDiagonal; tmp.distToDiagonal = (unsigned) abs(realDiagonal);
        if ((unsigned) tmp.score > (unsigned) max.score) max = tmp;
    };
    for (unsigned devisions = 1; (-devisions * 65536 + diagonal) > -tLen; devisions++) probe((int) (-devisions * 65536 + diagonal) + (int) half);
    for (unsigned devisions = 0; (devisions * 65536 + diagonal) < half; devisions++) probe((int) (devisions * 65536 + diagonal));
    max.diagonalLen = std::min(tLen, half);
    return max;
}
static int doRescoreHamming(const std::string &qPath, const std::string &tPath, const std::string &prefPath, const std::string &outPath, const RescorePar &par) {
    Db seq, pref; seq.load(tPath); pref.load(prefPath);
    if (qPath != tPath) { fprintf(stderr, "oracle: rescorediagonal expects queryDB == targetDB on this path\n"); return 1; }
    const bool revPref = (pref.dbtype & 0x7FFFFFFF) == 14;
    DbOut out; out.init(pref.size());
#pragma omp parallel num_threads(par.threads)
    {
        std::vector<Hit> hits; std::string buf, q2, qRev;
#pragma omp for schedule(dynamic, 1)
        for (size_t id = 0; id < pref.size(); id++) {
            const char *data = pref.getData(id); uint32_t qKey = pref.key[id];
            size_t qId = UINT_MAX; int qLen2 = -1, origLen = -1;
            if (*data) {
                qId = seq.getId(qKey); origLen = seq.seqLen(qId);
                q2.assign(seq.getData(qId), origLen); q2 += q2; qLen2 = 2 * origLen;                      // :162-167
                if (revPref) { qRev.assign(qLen2, 'X'); for (int p = qLen2 - 1; p > -1; p--) qRev[(qLen2 - 1) - p] = NUM2AA[REVRES[AA2NUM[(unsigned char) q2[p]]]]; }
            }
            hits.clear(); parseHits(data, hits); buf.clear();
            for (const Hit &h : hits) {
                const char *qa = q2.data(); bool isReverse = false;
                if (revPref && h.prefScore < 0) { qa = qRev.data(); isReverse = true; }
                size_t tId = seq.getId(h.seqId);
                const bool isIdentity = (qId == tId);
                const char *t = seq.getData(tId); int dbLen = seq.seqLen(tId);
                if (!canBeCovered(par.covThr, par.covMode, (float) origLen, (float) dbLen)) continue;
                if (dbLen > origLen) continue;                                                           // :216-220 (with a warning)
                const float targetLength = static_cast<float>(dbLen);
                LocalAln al = wrappedHamming(qa, (unsigned) qLen2, t, (unsigned) targetLength, h.diagonal);
                const int diagonalLen = (int) al.diagonalLen, distance = al.score, diagonal = al.diagonal;
                const float targetCov = static_cast<float>(diagonalLen) / static_cast<float>(dbLen), queryCov = static_cast<float>(diagonalLen) / static_cast<float>(origLen);
                const int idCnt = (static_cast<float>(distance));
                const double seqId = computeSeqId(par.seqIdMode, idCnt, origLen, dbLen, diagonalLen);
                const int alnLen = diagonalLen;
                const bool hasCov = hasCoverage(par.covThr, par.covMode, queryCov, targetCov);
                const bool hasSeqId = seqId >= (par.seqIdThr - std::numeric_limits<float>::epsilon());
                const bool hasEvalue = (0.0 <= par.evalThr);
                const bool hasAlnLen = (alnLen >= par.alnLenThr);
                if (isIdentity || (hasAlnLen && hasCov && hasSeqId && hasEvalue)) {
                    Hit o; o.seqId = h.seqId; o.prefScore = 100 * seqId; o.prefScore = isReverse ? -o.prefScore : o.prefScore; o.diagonal = (unsigned short) diagonal;
                    hitToBuf(buf, o);
                }
            }
            out.set(id, qKey, buf, 0);
        }
    }
    out.write(outPath, pref.dbtype);
    return 0;
}

// ----------------------------------------------------------------------------- K: kmermatcher
// xxHash64 of one 8-byte little-endian word (lib/mmseqs/lib/xxhash/xxhash.h, XXH64 with len=8), M/linclust/kmermatcher.cpp:33-38
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t xxh64_u64(uint64_t in, uint64_t seed) {
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL, P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
    uint64_t h = seed + P5 + 8;
    uint64_t k1 = in * P2; k1 = rotl64(k1, 31); k1 *= P1;
    h ^= k1; h = rotl64(h, 27) * P1 + P4;
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}
// M/commons/Util.cpp:601-638 (A,C,T,G = 0..3: complement = xor 2, then reverse the 2-bit groups)
static uint64_t revComplement(uint64_t kmer, int k) {
    uint64_t x = kmer ^ 0xAAAAAAAAAAAAAAAAULL;
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = __builtin_bswap64(x);
    return x >> (64 - 2 * k);
}
#define BIT63 (1ULL << 63)
template <typename T> struct __attribute__((__packed__)) KmerPos { uint64_t kmer; uint32_t id; T seqLen; T pos; };  // kmermatcher.h:49-54
struct SeqPos { unsigned short score; uint64_t kmer; unsigned pos; };                                                 // kmermatcher.h:10-14
struct KmerPar { int k = 20, kmersPerSeq = 200; float scale = 0.2f; uint64_t hashShift = 67; bool ignoreMultiKmer = true, onlyExtendable = false; int covMode = 1; float covThr = 0.0f; int threads = 1; };

template <typename T>
static int kmermatcherInner(const Db &seq, const std::string &outPath, const KmerPar &par) {
    const size_t n = seq.size();
    // computeKmerCount kmermatcher.cpp:573-582
    size_t totalKmers = 0;
    for (size_t id = 0; id < n; id++) {
        int L = (int) seq.seqLen(id);
        int adj = std::max(1, L - par.k + 2);
        totalKmers += std::min(adj, static_cast<int>(par.kmersPerSeq + (par.scale * L)));
    }
    // kmermatcherInner :620-627 single split: totalKmersPerSplit = max(1025, totalSizeNeeded/sizeof + 1)
    size_t arr = std::max<size_t>(1024 + 1, totalKmers + 1);
    std::vector<KmerPos<T>> kp(arr + 1);
    memset(kp.data(), 0xFF, sizeof(KmerPos<T>) * (arr + 1));
    // fillKmerPositionArray :77-388
    std::vector<size_t> cnt(n + 1, 0);
    std::vector<std::vector<KmerPos<T>>> per(n);
#pragma omp parallel num_threads(par.threads)
    {
        std::vector<SeqPos> kmers; std::vector<unsigned char> num;
        std::vector<unsigned short> scoreDist(65536); std::vector<unsigned> hier(128);
#pragma omp for schedule(dynamic, 100)
        for (size_t id = 0; id < n; id++) {
            std::fill(scoreDist.begin(), scoreDist.end(), 0); std::fill(hier.begin(), hier.end(), 0);
            const char *s = seq.getData(id); int L = 0; num.clear();
            while (s[L] != '\0' && s[L] != '\n' && (size_t) L < seq.seqLen(id)) { num.push_back(AA2NUM[(unsigned char) s[L]]); L++; }  // Sequence.cpp:476-489
            uint64_t h = 0; for (int i = 0; i < L; i++) h = h * 31 + num[i];  // Util::hash Util.h:338-346
            uint64_t seqHash = xxh64_u64(h, par.hashShift);
            kmers.clear();
            for (int pos = 0; pos + par.k <= L; pos++) {
                bool hasX = false; uint64_t idx = 0;
                for (int j = 0; j < par.k; j++) { hasX |= (num[pos + j] == 4); idx = (idx << 2) | num[pos + j]; }  // Indexer.h:136-143
                if (hasX) continue;
                uint64_t rc = revComplement(idx, par.k);
                if (rc == idx) continue;
                bool pickRev = rc < idx; uint64_t km = pickRev ? rc : idx;
                unsigned short hash = (unsigned short) xxh64_u64(km, par.hashShift);
                SeqPos sp; sp.kmer = pickRev ? (km & ~BIT63) : (km | BIT63); sp.pos = pickRev ? (L - pos - par.k) : pos; sp.score = hash;
                scoreDist[hash]++; hier[hash >> 9]++; kmers.push_back(sp);
            }
            size_t seqKmerCount = kmers.size();
            size_t considered = std::min(static_cast<size_t>(par.kmersPerSeq - 1 + (par.scale * L)), seqKmerCount);
            unsigned threshold = 0; size_t inBins = 0;
            if (seqKmerCount > 0) {
                size_t ht = 0;
                for (ht = 0; ht < 128 && inBins < considered; ht++) inBins += hier[ht];
                ht -= (ht > 0) ? 1 : 0;
                inBins -= hier[ht];
                for (threshold = ht * 512; threshold <= USHRT_MAX && inBins < considered; threshold++) inBins += scoreDist[threshold];
            }
            int tooMuch = (inBins - considered);
            std::vector<KmerPos<T>> &o = per[id];
            uint32_t dbKey = seq.key[id];
            { KmerPos<T> e; e.kmer = seqHash; e.id = dbKey; e.pos = 0; e.seqLen = L; o.push_back(e); }  // identity k-mer :244-267 (single split: always in range)
            if (par.ignoreMultiKmer)
                std::sort(kmers.begin(), kmers.end(), [](const SeqPos &a, const SeqPos &b) {  // compareByScoreReverse kmermatcher.h:30-46
                    if (a.score != b.score) return a.score < b.score;
                    uint64_t ak = a.kmer | BIT63, bk = b.kmer | BIT63;
                    if (ak != bk) return ak < bk;
                    return a.pos < b.pos; });
            size_t selected = 0;
            for (size_t ki = 0; ki < seqKmerCount && selected < considered; ki++) {  // :277-350
                if (par.ignoreMultiKmer) {
                    uint64_t km = kmers[ki].kmer | BIT63;
                    if (ki + 1 < seqKmerCount) {
                        uint64_t nx = kmers[ki + 1].kmer | BIT63;
                        if (km == nx) {
                            while (km == nx && ki < seqKmerCount) { ki++; if (ki >= seqKmerCount) break; nx = kmers[ki].kmer | BIT63; }
                        }
                    }
                    if (ki >= seqKmerCount) break;
                }
                if (kmers[ki].score < threshold) {
                    if (kmers[ki].score == (threshold - 1) && tooMuch) { tooMuch--; threshold -= (tooMuch == 0) ? 1 : 0; }
                    selected++;
                    KmerPos<T> e; e.kmer = kmers[ki].kmer; e.id = dbKey; e.pos = kmers[ki].pos; e.seqLen = L; o.push_back(e);
                }
            }
        }
    }
    size_t elems = 0;
    for (size_t id = 0; id < n; id++) { if (elems + per[id].size() >= arr) { fprintf(stderr, "Kmer array overflow\n"); return 1; } memcpy(&kp[elems], per[id].data(), per[id].size() * sizeof(KmerPos<T>)); elems += per[id].size(); std::vector<KmerPos<T>>().swap(per[id]); }
    auto cmp1 = [](const KmerPos<T> &a, const KmerPos<T> &b) {  // compareRepSequenceAndIdAndPosReverse kmermatcher.h:76-96
        uint64_t ak = a.kmer | BIT63, bk = b.kmer | BIT63;
        if (ak != bk) return ak < bk;
        if (a.seqLen != b.seqLen) return a.seqLen > b.seqLen;
        if (a.id != b.id) return a.id < b.id;
        return a.pos < b.pos; };
#ifdef _OPENMP
    omp_set_num_threads(par.threads);
    __gnu_parallel::sort(kp.begin(), kp.begin() + elems, cmp1);
#else
    std::sort(kp.begin(), kp.begin() + elems, cmp1);
#endif
    // assignGroup :453-562 (NUCL)
    size_t writePos = 0;
    {
        KmerPos<T> *hp = kp.data();
        uint64_t prevHash = hp[0].kmer | BIT63; uint64_t repSeqId = hp[0].id;
        size_t prevStart = 0, prevSetSize = 0; T queryLen = hp[0].seqLen; bool repIsReverse = false; T repPos = hp[0].pos;
        for (size_t e = 0; e < arr + 1; e++) {
            uint64_t cur = hp[e].kmer | BIT63;
            if (prevHash != cur) {
                for (size_t i = prevStart; i < e; i++) {
                    uint64_t km = hp[i].kmer | BIT63;
                    uint64_t rId = (km != UINT64_MAX) ? ((prevSetSize == 1) ? UINT64_MAX : repSeqId) : UINT64_MAX;
                    if (rId != UINT64_MAX) {
                        int diagonal;
                        bool targetIsReverse = (hp[i].kmer & BIT63) == 0; bool qRev = false; T qPos = 0, tPos = 0;
                        if (repIsReverse == true && targetIsReverse == false) { qPos = repPos; tPos = hp[i].pos; qRev = true; }
                        else if (repIsReverse == true && targetIsReverse == true) { qPos = (queryLen - 1) - repPos; tPos = (hp[i].seqLen - 1) - hp[i].pos; qRev = false; }
                        else if (repIsReverse == false && targetIsReverse == true) { qPos = (queryLen - 1) - repPos; tPos = (hp[i].seqLen - 1) - hp[i].pos; qRev = true; }
                        else { qPos = repPos; tPos = hp[i].pos; qRev = false; }
                        diagonal = qPos - tPos;
                        rId = qRev ? (rId & ~BIT63) : (rId | BIT63);
                        bool canBeExtended = diagonal < 0 || (diagonal > (queryLen - hp[i].seqLen));
                        bool cbc = canBeCovered(par.covThr, par.covMode, static_cast<float>(queryLen), static_cast<float>(hp[i].seqLen));
                        if ((par.onlyExtendable == false && cbc) || (canBeExtended && par.onlyExtendable == true)) {
                            hp[writePos].kmer = rId; hp[writePos].pos = diagonal; hp[writePos].seqLen = hp[i].seqLen; hp[writePos].id = hp[i].id; writePos++;
                        }
                    }
                    hp[i].kmer = (i != writePos - 1) ? UINT64_MAX : hp[i].kmer;
                }
                prevSetSize = 0; prevStart = e; repSeqId = hp[e].id;
                repIsReverse = (hp[e].kmer & BIT63) == 0;
                queryLen = hp[e].seqLen; repPos = hp[e].pos;
            }
            if (hp[e].kmer == UINT64_MAX) break;
            prevSetSize++;
            prevHash = hp[e].kmer | BIT63;
        }
    }
    auto cmp2 = [](const KmerPos<T> &a, const KmerPos<T> &b) {  // compareRepSequenceAndIdAndDiagReverse kmermatcher.h:98-114
        uint64_t ak = a.kmer | BIT63, bk = b.kmer | BIT63;
        if (ak != bk) return ak < bk;
        if (a.id != b.id) return a.id < b.id;
        return a.pos < b.pos; };
#ifdef _OPENMP
    __gnu_parallel::stable_sort(kp.begin(), kp.begin() + writePos, cmp2);
#else
    std::stable_sort(kp.begin(), kp.begin() + writePos, cmp2);
#endif
    // writeKmerMatcherResult :815-930 (1 thread) + fill-in :717-729
    DbOut out; out.init(n);
    std::vector<char> repSequence(seq.lastKey + 1, 0);
    {
        KmerPos<T> *hp = kp.data(); const size_t total = arr;
        std::string cur; size_t lastTarget = SIZE_MAX; unsigned writeSets = 0; uint64_t repSeqId = UINT64_MAX;
        auto flush = [&]() {
            if (writeSets > 0) { repSequence[repSeqId] = 1; size_t slot = seq.getId((uint32_t) repSeqId); out.set(slot, (uint32_t) repSeqId, cur, 0); }
            else if (repSeqId != UINT64_MAX) repSequence[repSeqId] = 0;
        };
        for (size_t kpos = 0; kpos < total && hp[kpos].kmer != UINT64_MAX; kpos++) {
            uint64_t ck = hp[kpos].kmer; int rm = (ck & BIT63) == 0; ck &= ~BIT63;
            if (repSeqId != ck) {
                flush(); writeSets = 0;
                lastTarget = SIZE_MAX; cur.clear(); repSeqId = ck;
                Hit h; h.seqId = (uint32_t) repSeqId; h.prefScore = 0; h.diagonal = 0; hitToBuf(cur, h);
            }
            unsigned targetId = hp[kpos].id; T diagonal = hp[kpos].pos; size_t ko = 0; T prevDiag = diagonal;
            size_t maxDiag = 0, diagCnt = 0, topScore = 0; int bestRm = rm;
            while (lastTarget != targetId && kpos + ko < total && hp[kpos + ko].id == targetId) {
                if (prevDiag == hp[kpos + ko].pos) diagCnt++; else diagCnt = 1;
                if (diagCnt >= maxDiag) { diagonal = hp[kpos + ko].pos; maxDiag = diagCnt; bestRm = (hp[kpos + ko].kmer & BIT63) == 0; }
                prevDiag = hp[kpos + ko].pos; ko++; topScore++;
            }
            if (targetId != repSeqId && lastTarget != targetId) { ; } else { lastTarget = targetId; continue; }
            Hit h; h.seqId = targetId; h.prefScore = bestRm ? -(int) topScore : (int) topScore; h.diagonal = (unsigned short) diagonal;
            hitToBuf(cur, h); lastTarget = targetId; writeSets++;
        }
        flush();
    }
    for (size_t id = 0; id < n; id++) {
        uint32_t k = seq.key[id];
        if (!repSequence[k]) { std::string b; Hit h; h.seqId = k; h.prefScore = 0; h.diagonal = 0; hitToBuf(b, h); out.set(id, k, b, seq.ext[id]); }
    }
    out.write(outPath, 14);
    return 0;
}
static int doKmermatcher(const std::string &seqPath, const std::string &outPath, const KmerPar &par) {
    Db seq; seq.load(seqPath);
    if (seq.maxSeqLen < SHRT_MAX) return kmermatcherInner<short>(seq, outPath, par);  // kmermatcher.cpp:803-808
    return kmermatcherInner<int>(seq, outPath, par);
}

// ----------------------------------------------------------------------------- CLI
static std::map<std::string, std::string> parseFlags(int argc, char **argv, std::vector<std::string> &pos) {
    std::map<std::string, std::string> f;
    for (int i = 0; i < argc; i++) {
        std::string a = argv[i];
        if (a.size() > 1 && a[0] == '-' && !(isdigit(a[1]))) { if (i + 1 < argc) { f[a] = argv[i + 1]; i++; } }
        else pos.push_back(a);
    }
    return f;
}
static AncientPar ancientPar(std::map<std::string, std::string> &f) {
    AncientPar p;
    if (f.count("--min-seq-id")) p.seqIdThr = strtof(f["--min-seq-id"].c_str(), NULL);
    if (f.count("--ext-random-align")) p.randAlnPenal = strtof(f["--ext-random-align"].c_str(), NULL);
    if (f.count("--excess-penalty")) p.excessPenal = strtof(f["--excess-penalty"].c_str(), NULL);
    if (f.count("--min-ryseq-id-corr-reads")) p.corrReadsRySeqId = strtof(f["--min-ryseq-id-corr-reads"].c_str(), NULL);
    if (f.count("--min-ryseq-id")) p.rySeqIdThr = strtof(f["--min-ryseq-id"].c_str(), NULL);
    if (f.count("--likelihood-ratio-threshold")) p.likelihoodThreshold = strtof(f["--likelihood-ratio-threshold"].c_str(), NULL);
    if (f.count("--unsafe")) p.unsafe = atoi(f["--unsafe"].c_str()) != 0;
    if (f.count("--min-merge-seq-id")) p.mergeSeqIdThr = strtof(f["--min-merge-seq-id"].c_str(), NULL);
    if (f.count("--min-cov-safe")) p.minCovSafe = atoi(f["--min-cov-safe"].c_str());
    if (f.count("--max-seq-len")) p.maxSeqLen = strtoull(f["--max-seq-len"].c_str(), NULL, 10);
    if (f.count("--ancient-damage")) p.damage = f["--ancient-damage"];
    if (f.count("--threads")) p.threads = atoi(f["--threads"].c_str());
    return p;
}
static void loadDamage(const std::string &prefix, std::vector<DiNuc> &d, std::vector<DiNuc> &r) { initDeam(prefix + "5p.prof", prefix + "3p.prof", d, r); }

int main(int argc, char **argv) {
    initAlphabet();
    if (argc < 2) { fprintf(stderr, "usage: cdm_oracle <module|probe> ...\n"); return 2; }
    std::string cmd = argv[1];
    std::vector<std::string> pos;
    auto t0 = std::chrono::steady_clock::now();
    int rc = 2;
    if (cmd == "probe") {
        std::string what = argc > 2 ? argv[2] : "";
        if (what == "damage") {
            std::vector<DiNuc> d, r; loadDamage(argv[3], d, r);
            for (int rv = 0; rv < 2; rv++) for (int i = 0; i < 11; i++) {
                const DiNuc &m = rv ? r[i] : d[i]; printf("%s %d", rv ? "rev" : "fwd", i);
                for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) printf(" %La", m.p[a][b]);
                printf("\n");
            }
            return 0;
        }
        if (what == "seqerr") { DiNuc e; seqErrProf(e, strtold(argv[3], NULL)); for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) printf("%La%c", e.p[a][b], (a == 3 && b == 3) ? '\n' : ' '); return 0; }
        if (what == "mostlikeli") {
            std::vector<DiNuc> d, r; loadDamage(argv[3], d, r); DiNuc se; seqErrProf(se, 0.01L);
            std::string line;
            while (std::getline(std::cin, line)) {
                if (line.empty()) continue;
                std::istringstream in(line); int qBase, wasCorr; unsigned qIter, qLen; in >> qBase >> qIter >> qLen >> wasCorr;
                Cnt c, rv; for (int t = 0; t < 4; t++) for (int l = 0; l < 11; l++) in >> c.count[t][l];
                for (int t = 0; t < 4; t++) for (int l = 0; l < 11; l++) in >> rv.count[t][l];
                printf("%d\n", mostLikeliBaseRead(qBase, qIter, c, rv, d, r, se, wasCorr != 0, qLen));
            }
            return 0;
        }
        if (what == "overlap") {
            std::vector<DiNuc> d, r; loadDamage(argv[3], d, r); DiNuc se; seqErrProf(se, 0.001L);
            std::string line;
            while (std::getline(std::cin, line)) {
                if (line.empty()) continue;
                std::istringstream in(line); std::string cons, tgt; unsigned qLen, dbKey, dbLen, alnLen, mL, mR; int qs, qe, ds, de, isRev; float rnd, exc;
                in >> cons >> tgt >> qLen >> dbKey >> qs >> qe >> ds >> de >> dbLen >> alnLen >> isRev >> mL >> mR >> rnd >> exc;
                Aln a; a.dbKey = dbKey; a.alnLength = alnLen; a.qStartPos = qs; a.qEndPos = qe; a.qLen = qLen; a.dbStartPos = ds; a.dbEndPos = de; a.dbLen = dbLen; a.isRev = isRev != 0;
                Scored s = rsPair(a, cons, tgt.c_str(), qLen, d, r, mL, mR, rnd, se, exc);
                printf("%a %a\n", s.sLenNorm, s.sRatio);
            }
            return 0;
        }
        if (what == "evalue") {
            size_t dbRes = strtoull(argv[3], NULL, 10); std::string line;
            while (std::getline(std::cin, line)) {
                if (line.empty()) continue;
                std::istringstream in(line); int score, qLen; in >> score >> qLen;
                double ev = computeEvalue(score, qLen, dbRes), bs = computeBitScore(score);
                printf("%a %a %.3E %a\n", ev, bs, ev, rawScoreFromBitScore(static_cast<int>(bs + 0.5)));
            }
            return 0;
        }
        return 2;
    }
    auto flags = parseFlags(argc - 2, argv + 2, pos);
    if (cmd == "kmermatcher" && pos.size() >= 2) {
        KmerPar p;
        if (flags.count("-k")) p.k = atoi(flags["-k"].c_str());
        if (flags.count("--kmer-per-seq")) p.kmersPerSeq = atoi(flags["--kmer-per-seq"].c_str());
        if (flags.count("--kmer-per-seq-scale")) p.scale = strtof(flags["--kmer-per-seq-scale"].c_str(), NULL);
        if (flags.count("--hash-shift")) p.hashShift = strtoull(flags["--hash-shift"].c_str(), NULL, 10);
        if (flags.count("--ignore-multi-kmer")) p.ignoreMultiKmer = atoi(flags["--ignore-multi-kmer"].c_str()) != 0;
        if (flags.count("--include-only-extendable")) p.onlyExtendable = atoi(flags["--include-only-extendable"].c_str()) != 0;
        if (flags.count("--cov-mode")) p.covMode = atoi(flags["--cov-mode"].c_str());
        if (flags.count("-c")) p.covThr = strtof(flags["-c"].c_str(), NULL);
        if (flags.count("--threads")) p.threads = atoi(flags["--threads"].c_str());
        rc = doKmermatcher(pos[0], pos[1], p);
    } else if (cmd == "rescorediagonal" && pos.size() >= 4) {
        RescorePar p;
        if (flags.count("--min-seq-id")) p.seqIdThr = strtof(flags["--min-seq-id"].c_str(), NULL);
        if (flags.count("-e")) p.evalThr = strtod(flags["-e"].c_str(), NULL);
        if (flags.count("--cov-mode")) p.covMode = atoi(flags["--cov-mode"].c_str());
        if (flags.count("-c")) p.covThr = strtof(flags["-c"].c_str(), NULL);
        if (flags.count("--seq-id-mode")) p.seqIdMode = atoi(flags["--seq-id-mode"].c_str());
        if (flags.count("--min-aln-len")) p.alnLenThr = atoi(flags["--min-aln-len"].c_str());
        if (flags.count("--threads")) p.threads = atoi(flags["--threads"].c_str());
        if (flags.count("--rescore-mode")) p.rescoreMode = atoi(flags["--rescore-mode"].c_str());
        if (flags.count("--wrapped-scoring")) p.wrapped = atoi(flags["--wrapped-scoring"].c_str()) != 0;
        if (p.rescoreMode == 0 && p.wrapped) rc = doRescoreHamming(pos[0], pos[1], pos[2], pos[3], p);
        else if (p.rescoreMode != 3 || p.wrapped) { fprintf(stderr, "oracle: --rescore-mode 3, or --rescore-mode 0 with --wrapped-scoring 1\n"); return 1; }
        else rc = doRescore(pos[0], pos[1], pos[2], pos[3], p);
    } else if (cmd == "ancient_correction" && pos.size() >= 3) {
        AncientPar p = ancientPar(flags); rc = doCorrection(pos[0], pos[1], pos[2], p);
    } else if (cmd == "ancient_read_assemble" && pos.size() >= 3) {
        AncientPar p = ancientPar(flags); rc = doAssemble(pos[0], pos[1], pos[2], p);
    } else if (cmd == "ancient_contig_merge" && pos.size() >= 3) {
        AncientPar p = ancientPar(flags); rc = doContigMerge(pos[0], pos[1], pos[2], p);
    } else if (cmd == "cyclecheck" && pos.size() >= 2) {
        size_t maxSeqLen = flags.count("--max-seq-len") ? strtoull(flags["--max-seq-len"].c_str(), NULL, 10) : 65535;
        rc = doCycleCheck(pos[0], pos[1], maxSeqLen, flags.count("--chop-cycle") && atoi(flags["--chop-cycle"].c_str()) != 0);
    } else { fprintf(stderr, "unknown/incomplete command %s\n", cmd.c_str()); return 2; }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "Time for processing: %.3fs\n", sec);
    return rc;
}
