// The steps either side of the hot path, host side:
//   createdb       FASTA/FASTQ[.gz] -> sequence DB + header DB + .lookup + .source      lib/mmseqs/src/util/createdb.cpp:60-340
//   convert2fasta  sequence DB + header DB -> FASTA                                     lib/mmseqs/src/util/convert2fasta.cpp
//   createhdb      "<id> len:<len>[ cycle:<0|1>]" headers for assembled sequences        src/util/createhdb.cpp
// plus readFastxAsDb, which hands the parsed reads to the device path without going through DB files (ancient_reads_loop).
// The parser follows kseq.h as KSeqWrapper drives it (lib/mmseqs/lib/kseq/kseq.h): a record starts at a line whose first
// character is '>' or '@'; name = up to the first white space, comment = rest of the line; the sequence is every following line
// (without line ends) up to a line that starts with '>', '+' or '@'; after '+' as many quality characters as sequence letters
// are skipped.
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <omp.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "mmdb.h"

namespace {
struct Entries {
    HVec<char> seqBlob, hdrBlob;               // "SEQ\n\0" / "header\n\0" per entry, in input order
    HVec<uint64_t> seqOff, hdrOff; HVec<uint32_t> seqLen, hdrLen, file;
};
template <typename V> inline void appendBytes(V &v, const char *p, size_t n) {
    if (v.capacity() - v.size() < n) v.reserve(std::max(v.capacity() * 2, v.size() + n + (64u << 20)));
    const size_t at = v.size(); v.resize(at + n); memcpy(v.data() + at, p, n);
}
// kseq_read (lib/mmseqs/lib/ksw2/kseq.h:184-233) as a state machine over blocks of the (inflated) file, so that reading / inflating
// runs in a thread of its own while this one parses; sequence letters go straight into the blob.  What is kept of kseq, malformed
// input included: the first record - and the one after a quality block - starts at the next '>' or '@' wherever it stands (:190-191);
// the name ends at any white space (isspace, :199), the comment is the rest of the line (:200); a sequence / quality line loses one
// trailing '\r' when more than one character has accumulated (:145); lines are sequence until one starts with '>', '@' or '+'
// (:206); quality lines are read until they hold as many characters as the sequence, and a record whose quality is longer or cut
// short by the end of the file ENDS the reading of that file (:227-231 return -2, KSeqWrapper.cpp:20-22).
struct FastxParser {
    Entries &e; const uint32_t fileIdx;
    enum State { SEEK, HEADER, SEQ, PLUS, QUAL, STOP } st = SEEK;
    bool midLine = false;                       // SEQ / PLUS / QUAL: the current line began in an earlier block
    size_t recSeq = 0, recHdr = 0;              // where the current record's letters / header start in the blobs
    std::string header, qual;
    bool invalidEntry = false; size_t invalidAt = 0;
    // parallel parsing of a plain file: a parser owns the records that START in front of `limit` and stops - `stoppedAt` - where the
    // first one behind it starts
    const char *limit = nullptr, *stoppedAt = nullptr;
    FastxParser(Entries &en, uint32_t f) : e(en), fileIdx(f) {}
    void beginRecord(const char *h, size_t n) {         // h = the header line behind '>' / '@', without its line end
        size_t ne = 0;
        while (ne < n && !isspace((unsigned char) h[ne])) ne++;
        if (ne == 0 && !invalidEntry) { invalidEntry = true; invalidAt = e.seqOff.size(); }
        header.assign(h, ne);
        if (ne < n && h[ne] != '\n') {                  // (the delimiter is consumed; it is never '\n' here: the line end is not part of h)
            size_t cs = ne + 1, cl = n - cs;
            if (cl > 1 && h[cs + cl - 1] == '\r') cl--;
            if (cl > 0) { header.push_back(' '); header.append(h + cs, cl); }
        }
        header.push_back('\n');
        recSeq = e.seqBlob.size(); recHdr = e.hdrBlob.size();
        appendBytes(e.hdrBlob, header.data(), header.size()); e.hdrBlob.push_back('\0');
    }
    bool refused = false;                       // a record kseq returns -2 for: the reading of the file ends there
    void dropRecord() { e.seqBlob.resize(recSeq); e.hdrBlob.resize(recHdr); refused = true; }
    void endRecord() {
        const size_t L = e.seqBlob.size() - recSeq;
        e.seqOff.push_back(recSeq); e.seqLen.push_back((uint32_t) L + 2);
        e.seqBlob.push_back('\n'); e.seqBlob.push_back('\0');
        e.hdrOff.push_back(recHdr); e.hdrLen.push_back((uint32_t) header.size() + 1);
        e.file.push_back(fileIdx);
    }
    // one (piece of a) line for the sequence / the quality; done = its line end was seen
    void seqPiece(const char *p, size_t n, bool done) {
        appendBytes(e.seqBlob, p, n);
        if (done && e.seqBlob.size() - recSeq > 1 && e.seqBlob.back() == '\r') e.seqBlob.pop_back();
    }
    // consumes [p, end) and returns how much of it is done with; what is left (an incomplete header line only) has to come again in
    // front of the next block.  eof: no more data follows.
    size_t feed(const char *const begin, const char *const end, bool eof) {
        const char *p = begin;
        while (st != STOP) {
            if (st == SEEK) {
                const char *q = p;
                while (q < end && *q != '>' && *q != '@') q++;
                if (q == end) return (size_t) (end - begin);
                if (limit && q >= limit) { stoppedAt = q; st = STOP; break; }
                p = q + 1; st = HEADER;
            } else if (st == HEADER) {
                const char *nl = (const char *) memchr(p, '\n', (size_t) (end - p));
                if (!nl && !eof) return (size_t) (p - begin);
                if (!nl && p == end) { st = STOP; break; }                  // '>' was the last character: ks_getuntil returns -1 (:199)
                const char *le = nl ? nl : end;
                beginRecord(p, (size_t) (le - p));
                p = nl ? nl + 1 : end; st = SEQ; midLine = false;
            } else if (st == SEQ) {
                if (p == end) { if (eof) { endRecord(); st = STOP; } return (size_t) (end - begin); }
                if (!midLine) {
                    const char c = *p;
                    if (c == '>' || c == '@') {
                        endRecord();
                        if (limit && p >= limit) { stoppedAt = p; st = STOP; break; }
                        p++; st = HEADER; continue;
                    }
                    if (c == '+') { st = PLUS; midLine = false; p++; continue; }
                    if (c == '\n') { p++; continue; }
                }
                const char *nl = (const char *) memchr(p, '\n', (size_t) (end - p));
                if (nl) { seqPiece(p, (size_t) (nl - p), true); p = nl + 1; midLine = false; }
                else { seqPiece(p, (size_t) (end - p), eof); p = end; midLine = !eof; }
            } else if (st == PLUS) {
                const char *nl = (const char *) memchr(p, '\n', (size_t) (end - p));
                if (nl) { p = nl + 1; st = QUAL; qual.clear(); midLine = false; }
                else { if (eof) { dropRecord(); st = STOP; } return (size_t) (end - begin); }      // no quality string: -2
            } else {   // QUAL
                const size_t L = e.seqBlob.size() - recSeq;
                if (p == end) {
                    if (!eof) return (size_t) (end - begin);
                    if (qual.size() == L) endRecord(); else dropRecord();
                    st = STOP; break;
                }
                const char *nl = (const char *) memchr(p, '\n', (size_t) (end - p));
                const char *le = nl ? nl : end;
                qual.append(p, (size_t) (le - p));
                p = nl ? nl + 1 : end;
                if (!nl && !eof) { midLine = true; continue; }
                midLine = false;
                if (qual.size() > 1 && qual.back() == '\r') qual.pop_back();
                if (qual.size() < L && !(p == end && eof)) continue;
                if (qual.size() != L) { dropRecord(); st = STOP; break; }
                endRecord(); st = SEEK;
            }
        }
        return (size_t) (end - begin);
    }
};
// reads / inflates a file block by block in a thread of its own; the parser takes the blocks in order
struct BlockReader {
    static const size_t BLOCK = 32u << 20, HEAD = 1u << 20;      // HEAD: room in front of a block for the parser's unfinished header line
    struct Block { HVec<char> buf; size_t n = 0; bool last = false; };
    gzFile f = nullptr; Block blk[2]; int filled[2] = {0, 0}; bool failed = false, stop = false;
    std::mutex m; std::condition_variable cv; std::thread th;
    bool open(const std::string &p) {
        f = gzopen(p.c_str(), "rb");
        if (!f) return false;
        gzbuffer(f, 4u << 20);
        for (auto &b : blk) b.buf.resize(HEAD + BLOCK);
        th = std::thread([this] {
            for (int i = 0;; i ^= 1) {
                { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return filled[i] == 0 || stop; }); if (stop) return; }
                size_t got = 0; bool last = false;
                while (got < BLOCK) { const int r = gzread(f, blk[i].buf.data() + HEAD + got, (unsigned) std::min<size_t>(BLOCK - got, 1u << 30)); if (r < 0) failed = true; if (r <= 0) { last = true; break; } got += (size_t) r; }
                blk[i].n = got; blk[i].last = last;
                { std::lock_guard<std::mutex> l(m); filled[i] = 1; }
                cv.notify_all();
                if (last) return;
            }
        });
        return true;
    }
    Block &take(int i) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return filled[i] == 1; }); return blk[i]; }
    void release(int i) { { std::lock_guard<std::mutex> l(m); filled[i] = 0; } cv.notify_all(); }
    ~BlockReader() { if (th.joinable()) { { std::lock_guard<std::mutex> l(m); stop = true; } cv.notify_all(); th.join(); } if (f) gzclose(f); }
};
// A plain (not compressed) file is parsed by all threads at once, each on a stretch of the mapped file.  Where a stretch starts is a
// guess - a line that begins with '>', or with '@' when the line after the next begins with '+' - and the guess is CHECKED: the
// parser of the stretch in front, run exactly like the serial one, has to arrive at that very byte looking for a record start.
// Any disagreement, any record kseq would refuse, and the file is parsed serially instead: the result is the serial one either way.
bool parsePlainParallel(const std::string &path, uint32_t fileIdx, Entries &e) {
    const int T = std::max(1, omp_get_max_threads());
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    static const size_t minBytes = getenv("CDM_INGEST_PAR_MIN") ? strtoull(getenv("CDM_INGEST_PAR_MIN"), nullptr, 10) : (64u << 20);      // (tests lower it)
    if (fstat(fd, &st) != 0 || (size_t) st.st_size < std::max<size_t>(minBytes, 2) || T < 2) { close(fd); return false; }
    const size_t S = (size_t) st.st_size;
    const char *m = (const char *) mmap(nullptr, S, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return false;
    bool ok = !(S >= 2 && (unsigned char) m[0] == 0x1f && (unsigned char) m[1] == 0x8b);      // gzip: the serial, inflating path
    std::vector<size_t> start(T + 1, S);
    start[0] = 0;
    for (int t = 1; t < T && ok; t++) {
        size_t p = S * (size_t) t / T; const size_t stop = std::min(S, p + (4u << 20));
        bool found = false;
        while (p < stop && !found) {
            const char *nl = (const char *) memchr(m + p, '\n', stop - p);
            if (!nl) break;
            p = (size_t) (nl - m) + 1;
            if (p >= S) break;
            if (m[p] == '>') found = true;
            else if (m[p] == '@') {
                const char *l2 = (const char *) memchr(m + p, '\n', S - p);
                const char *l3 = l2 ? (const char *) memchr(l2 + 1, '\n', S - (size_t) (l2 + 1 - m)) : nullptr;
                if (l3 && (size_t) (l3 + 1 - m) < S && l3[1] == '+') found = true;
            }
        }
        if (!found) ok = false;
        start[t] = p;
    }
    for (int t = 1; t < T && ok; t++) if (start[t] <= start[t - 1]) ok = false;
    std::vector<Entries> part(ok ? T : 0);
    if (ok) {
        std::vector<char> good(T, 0);
#pragma omp parallel for schedule(static, 1) num_threads(T)     // a loop over the slices: a smaller team still visits them all
        for (int t = 0; t < T; t++) {
            Entries &pe = part[t];
            const size_t span = start[t + 1] - start[t];
            pe.seqBlob.reserve(span + 2); pe.hdrBlob.reserve(span / 2 + 2);
            const size_t guess = span / 128 + 16;
            pe.seqOff.reserve(guess); pe.hdrOff.reserve(guess); pe.seqLen.reserve(guess); pe.hdrLen.reserve(guess); pe.file.reserve(guess);
            FastxParser ps(pe, fileIdx);
            ps.limit = t + 1 < T ? m + start[t + 1] : nullptr;
            ps.feed(m + start[t], m + S, true);
            // arrived exactly where the next stretch was guessed to start (the last one: at the end of the file), nothing refused
            good[t] = !ps.invalidEntry && !ps.refused && (t + 1 < T ? ps.stoppedAt == m + start[t + 1] : ps.stoppedAt == nullptr);
        }
        for (int t = 0; t < T; t++) ok = ok && good[t];
    }
    if (ok) {
        std::vector<size_t> cnt(T + 1, 0), sb(T + 1, 0), hb(T + 1, 0);
        for (int t = 0; t < T; t++) { cnt[t + 1] = cnt[t] + part[t].seqOff.size(); sb[t + 1] = sb[t] + part[t].seqBlob.size(); hb[t + 1] = hb[t] + part[t].hdrBlob.size(); }
        const size_t n0 = e.seqOff.size(), s0 = e.seqBlob.size(), h0 = e.hdrBlob.size();
        if (e.seqBlob.capacity() < s0 + sb[T]) e.seqBlob.reserve(s0 + sb[T]);
        if (e.hdrBlob.capacity() < h0 + hb[T]) e.hdrBlob.reserve(h0 + hb[T]);
        e.seqBlob.resize(s0 + sb[T]); e.hdrBlob.resize(h0 + hb[T]);
        e.seqOff.resize(n0 + cnt[T]); e.hdrOff.resize(n0 + cnt[T]); e.seqLen.resize(n0 + cnt[T]); e.hdrLen.resize(n0 + cnt[T]); e.file.resize(n0 + cnt[T]);
#pragma omp parallel for schedule(static, 1) num_threads(T)     // a loop over the slices: a smaller team still visits them all
        for (int t = 0; t < T; t++) {
            const Entries &pe = part[t];
            memcpy(e.seqBlob.data() + s0 + sb[t], pe.seqBlob.data(), pe.seqBlob.size());
            memcpy(e.hdrBlob.data() + h0 + hb[t], pe.hdrBlob.data(), pe.hdrBlob.size());
            for (size_t i = 0, k = n0 + cnt[t]; i < pe.seqOff.size(); i++, k++) {
                e.seqOff[k] = pe.seqOff[i] + s0 + sb[t]; e.hdrOff[k] = pe.hdrOff[i] + h0 + hb[t];
                e.seqLen[k] = pe.seqLen[i]; e.hdrLen[k] = pe.hdrLen[i]; e.file[k] = fileIdx;
            }
        }
    }
    munmap((void *) m, S);
    if (getenv("CDM_TIMING")) fprintf(stderr, "  %s: %s\n", path.c_str(), ok ? "parsed by all threads (every stretch start confirmed)" : "parsed serially");
    return ok;
}
bool parseFile(const std::string &path, uint32_t fileIdx, Entries &e, std::string *err) {
    if (!getenv("CDM_INGEST_SERIAL") && parsePlainParallel(path, fileIdx, e)) return true;
    BlockReader r;
    if (!r.open(path)) { *err = "Cannot open " + path; return false; }
    FastxParser ps(e, fileIdx);
    std::string carry;
    for (int i = 0;; i ^= 1) {
        BlockReader::Block &b = r.take(i);
        if (carry.size() > BlockReader::HEAD) { *err = "Header line of more than 1 MB in " + path; return false; }
        char *begin = b.buf.data() + BlockReader::HEAD - carry.size();
        memcpy(begin, carry.data(), carry.size());
        const char *end = b.buf.data() + BlockReader::HEAD + b.n;
        const size_t used = ps.feed(begin, end, b.last);
        carry.assign(begin + used, (size_t) (end - (begin + used)));
        const bool last = b.last;
        if (ps.st == FastxParser::STOP) break;          // the rest of the file is not looked at (see above)
        r.release(i);
        if (last) break;
    }
    if (r.failed) { *err = "Cannot read " + path; return false; }
    if (ps.invalidEntry) { *err = "Fasta entry " + std::to_string(ps.invalidAt) + " is invalid"; return false; }
    return true;
}
// Util::parseFastaHeader (Util.cpp:173-256): the identifier part of a header
std::string fastaId(const char *h) {
    size_t len = 0; while (h[len] && h[len] != ' ' && h[len] != '\t' && h[len] != '\n' && h[len] != '\r') len++;      // Util::skipNoneWhitespace
    const std::string header(h, len);
    if (header.empty()) return "";
    size_t offset = header.compare(0, 10, "consensus_") == 0 ? 10 : 0;
    static const struct { const char *prefix; unsigned length, bar; } dbs[] = {
        {"uc", 2, 0}, {"cl|", 3, 1}, {"sp|", 3, 1}, {"tr|", 3, 1}, {"gb|", 3, 1}, {"ref|", 4, 1}, {"pdb|", 4, 1}, {"bbs|", 4, 1}, {"lcl|", 4, 1},
        {"pir||", 5, 1}, {"prf||", 5, 1}, {"gnl|", 4, 2}, {"pat|", 4, 2}, {"gi|", 3, 3}};
    for (const auto &d : dbs) {
        if (header.compare(offset, strlen(d.prefix), d.prefix) != 0) continue;
        size_t start = offset + d.length;
        for (unsigned j = 0; j + 1 < d.bar; j++) { const size_t end = header.find_first_of('|', start); if (end == std::string::npos) return ""; start = end + 1; }
        size_t end = header.find_first_of('|', start);
        if (end == std::string::npos) end = header.find_first_of(" \n", start);
        if (end == std::string::npos) end = header.size();
        return header.substr(start, end - start);
    }
    size_t end = header.find_first_of(" \n", offset);
    if (end == std::string::npos) end = header.size();
    return header.substr(offset, end - offset);
}
// createdb's order: with --shuffle 1 entry i goes to split i % 32 and the splits are concatenated and renumbered
// (createdb.cpp:60,220,277-280; DBWriter::createRenumberedDB)
std::vector<uint32_t> entryOrder(size_t n, bool shuffle) {
    std::vector<uint32_t> o; o.reserve(n);
    if (!shuffle) { for (size_t i = 0; i < n; i++) o.push_back((uint32_t) i); return o; }
    for (unsigned s = 0; s < 32; s++) for (size_t i = s; i < n; i += 32) o.push_back((uint32_t) i);
    return o;
}
bool parseAll(const std::vector<std::string> &files, Entries &e, std::string *err) {
    for (size_t f = 0; f < files.size(); f++) if (!parseFile(files[f], (uint32_t) f, e, err)) return false;
    if (e.seqOff.empty()) { *err = "The input files have no entry. Only files in fasta/fastq[.gz] are supported"; return false; }
    return true;
}
std::string baseName(const std::string &p) { const size_t s = p.find_last_of('/'); return s == std::string::npos ? p : p.substr(s + 1); }
}  // namespace

// parsed reads in createdb's order as an in-memory sequence DB: keys 0..n-1, wasExtended 0.  The blob stays in input order - the
// shuffle is a permutation of the offsets, nothing is copied a second time.
bool readFastxAsDb(const std::vector<std::string> &files, bool shuffle, FastxDb &out, std::string *err) {
    Entries e;
    if (!parseAll(files, e, err)) return false;
    const std::vector<uint32_t> order = entryOrder(e.seqOff.size(), shuffle);
    const size_t n = order.size();
    out.key.resize(n); out.off.resize(n); out.len.resize(n);
#pragma omp parallel for schedule(static)
    for (size_t j = 0; j < n; j++) { const uint32_t i = order[j]; out.key[j] = (uint32_t) j; out.off[j] = e.seqOff[i]; out.len[j] = e.seqLen[i]; }
    out.blob.swap(e.seqBlob);
    return true;
}

// dbType: 2 = nucleotides, 0 = createdb's own guess (M/util/createdb.cpp:175-199, 253-258): the first ten entries are looked at (the
// sample counter stops there: `sampleCount % 100 == 0` is never reached again), and the DB is a nucleotide DB iff more than 90 % of
// the letters of EACH of them are A, C, G, T, U or N in either case (an empty sequence gives 0/0: not a nucleotide sequence)
int createdbModule(const std::vector<std::string> &files, const std::string &outPath, bool shuffle, int dbType, std::string *err) {
    Entries e;
    if (!parseAll(files, e, err)) return 1;
    const size_t n = e.seqOff.size();
    int seqDbType = 1;      // DBTYPE_NUCLEOTIDES
    if (dbType == 0) {
        size_t isNucl = 0, sampled = 0;
        for (size_t i = 0; i < n && sampled < 10; i++, sampled++) {
            const char *q = e.seqBlob.data() + e.seqOff[i]; const size_t L = e.seqLen[i] - 2;
            size_t cnt = 0;
            for (size_t j = 0; j < L; j++) { const int c = toupper(q[j]); cnt += (c == 'T' || c == 'A' || c == 'G' || c == 'C' || c == 'U' || c == 'N'); }
            const float frac = static_cast<float>(cnt) / static_cast<float>(L);
            if (frac > 0.9) isNucl++;
        }
        if (isNucl != sampled) seqDbType = 0;       // DBTYPE_AMINO_ACIDS: written as the reference writes it; the modules of this path refuse such a DB
    }
    const std::vector<uint32_t> order = entryOrder(n, shuffle);
    // the data files hold the entries in createdb's order: gathered by all threads, every slice straight to its place in the blob
    HVec<uint32_t> key(n), sLen(n), hLen(n); HVec<uint64_t> sOff(n + 1), hOff(n + 1); HVec<uint8_t> ext(n);
    sOff[0] = hOff[0] = 0;
    for (size_t j = 0; j < n; j++) { const uint32_t i = order[j]; key[j] = (uint32_t) j; ext[j] = 0; sLen[j] = e.seqLen[i]; hLen[j] = e.hdrLen[i]; sOff[j + 1] = sOff[j] + sLen[j]; hOff[j + 1] = hOff[j] + hLen[j]; }
    HVec<char> sBlob(sOff[n]), hBlob(hOff[n]);
    const int T = std::max(1, omp_get_max_threads());
    std::vector<std::string> lookupPart(T);
#pragma omp parallel for schedule(static, 1) num_threads(T)     // a loop over the slices: a smaller team still visits them all
    for (int t = 0; t < T; t++) {
        std::string &lk = lookupPart[t];
        for (size_t j = n * (size_t) t / T, hi = n * (size_t) (t + 1) / T; j < hi; j++) {
            const uint32_t i = order[j];
            memcpy(sBlob.data() + sOff[j], e.seqBlob.data() + e.seqOff[i], sLen[j]);
            memcpy(hBlob.data() + hOff[j], e.hdrBlob.data() + e.hdrOff[i], hLen[j]);
            lk += std::to_string(j); lk.push_back('\t'); lk += fastaId(e.hdrBlob.data() + e.hdrOff[i]); lk.push_back('\t');
            lk += std::to_string(e.file[i]); lk.push_back('\n');
        }
    }
    if (!mmdbWriteBlob(outPath, seqDbType, sBlob.data(), sBlob.size(), key.data(), sOff.data(), sLen.data(), ext.data(), n, err)) return 1;
    if (!mmdbWriteBlob(outPath + "_h", 12 /* DBTYPE_GENERIC_DB */, hBlob.data(), hBlob.size(), key.data(), hOff.data(), hLen.data(), ext.data(), n, err)) return 1;
    FILE *lf = fopen((outPath + ".lookup").c_str(), "w"), *sf = fopen((outPath + ".source").c_str(), "w");
    if (!lf || !sf) { *err = "Cannot open " + outPath + ".lookup for writing"; return 1; }
    for (const std::string &lk : lookupPart) fwrite(lk.data(), 1, lk.size(), lf);
    fclose(lf);
    for (size_t f = 0; f < files.size(); f++) fprintf(sf, "%zu\t%s\n", f, baseName(files[f]).c_str());
    fclose(sf);
    return 0;
}

int convert2fastaModule(const std::string &dbPath, const std::string &outPath, std::string *err) {
    MmDb db, hdr;
    if (!db.load(dbPath, err) || !hdr.load(dbPath + "_h", err)) return 1;
    FILE *out = fopen(outPath.c_str(), "w");
    if (!out) { *err = "Cannot open " + outPath; return 1; }
    for (size_t i = 0; i < db.size(); i++) {        // convert2fasta.cpp: '>' header (without "\n\0") '\n' sequence (without "\n\0") '\n'
        const int64_t h = hdr.idOf(db.key[i]);
        if (h < 0) { *err = "Invalid database read for key " + std::to_string(db.key[i]); fclose(out); return 1; }
        fputc('>', out); fwrite(hdr.entry(h), 1, hdr.len[h] >= 2 ? hdr.len[h] - 2 : 0, out); fputc('\n', out);
        fwrite(db.entry(i), 1, db.len[i] >= 2 ? db.len[i] - 2 : 0, out); fputc('\n', out);
    }
    if (fclose(out) != 0) { *err = "Cannot close file " + outPath; return 1; }
    return 0;
}

int createhdbModule(const std::string &seqPath, const std::string &cyclePath, const std::string &outPath, std::string *err) {
    // both DBs are opened with USE_INDEX only (createhdb.cpp:21-31): the workflow hands over a cycle "DB" that is nothing but an
    // index file written by awk (data/nuclassemble.sh:222-241), and lengths come from the index anyway
    MmDb db, cyc;
    if (!db.load(seqPath, err, true)) return 1;
    const bool hasCycle = !cyclePath.empty();
    if (hasCycle && !cyc.load(cyclePath, err, true)) return 1;
    OutChunk c;
    for (size_t id = 0; id < db.size(); id++) {      // createhdb.cpp: "<id> len:<len>[ cycle:<0|1>]\n" under the sequence's key
        std::string h = std::to_string(id) + " len:" + std::to_string(db.len[id] >= 2 ? db.len[id] - 2 : 0);
        if (hasCycle) h += std::string(" cycle:") + (cyc.idOf(db.key[id]) >= 0 ? "1" : "0");
        h.push_back('\n');
        c.add(db.key[id], h.data(), h.size(), 0);
    }
    std::vector<OutChunk> chunks; chunks.push_back(std::move(c));
    return mmdbWriteChunks(outPath + "_h", 12, chunks, err) ? 0 : 1;
}
