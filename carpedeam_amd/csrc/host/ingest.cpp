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
#include <string>
#include <vector>
#include <zlib.h>

#include "mmdb.h"

namespace {
struct LineReader {
    gzFile f = nullptr; std::vector<char> buf; size_t pos = 0, end = 0; bool eof = false;
    bool open(const std::string &p) { f = gzopen(p.c_str(), "rb"); if (f) gzbuffer(f, 1 << 20); buf.resize(1 << 22); return f != nullptr; }
    ~LineReader() { if (f) gzclose(f); }
    // next line without its line end ('\n', and a '\r' in front of it); false at the end of the file
    bool line(std::string &out) {
        out.clear();
        while (true) {
            if (pos == end) {
                if (eof) return !out.empty();
                const int n = gzread(f, buf.data(), (unsigned) buf.size());
                if (n <= 0) { eof = true; return !out.empty(); }
                pos = 0; end = (size_t) n;
            }
            const char *nl = (const char *) memchr(buf.data() + pos, '\n', end - pos);
            if (nl) {
                out.append(buf.data() + pos, nl - (buf.data() + pos));
                pos = (size_t) (nl - buf.data()) + 1;
                if (!out.empty() && out.back() == '\r') out.pop_back();
                return true;
            }
            out.append(buf.data() + pos, end - pos);
            pos = end;
        }
    }
};
struct Entries {
    std::string seqBlob, hdrBlob;               // "SEQ\n\0" / "header\n\0" per entry, in input order
    std::vector<uint64_t> seqOff, hdrOff; std::vector<uint32_t> seqLen, hdrLen, file;
};
bool parseFile(const std::string &path, uint32_t fileIdx, Entries &e, std::string *err) {
    LineReader r;
    if (!r.open(path)) { *err = "Cannot open " + path; return false; }
    std::string line, seq, header;
    bool have = r.line(line);
    while (have) {
        if (line.empty() || (line[0] != '>' && line[0] != '@')) { have = r.line(line); continue; }
        const bool fastq = line[0] == '@';
        // name + comment: kseq separates them at the first white space and createdb joins them with one blank again
        size_t ne = 1;
        while (ne < line.size() && line[ne] != ' ' && line[ne] != '\t') ne++;
        if (ne == 1) { *err = "Fasta entry " + std::to_string(e.seqOff.size()) + " is invalid"; return false; }
        header.assign(line, 1, ne - 1);
        if (ne + 1 <= line.size() && ne < line.size()) { const std::string comment = line.substr(ne + 1); if (!comment.empty()) { header.push_back(' '); header += comment; } }
        header.push_back('\n');
        seq.clear();
        have = r.line(line);
        while (have && !(line.size() && (line[0] == '>' || line[0] == '+' || line[0] == '@'))) { seq += line; have = r.line(line); }
        if (fastq && have && line[0] == '+') {          // quality: as many characters as the sequence has
            size_t q = 0;
            have = r.line(line);
            while (have && q < seq.size()) { q += line.size(); have = r.line(line); if (q >= seq.size()) break; }
        }
        e.seqOff.push_back(e.seqBlob.size()); e.seqLen.push_back((uint32_t) seq.size() + 2);
        e.seqBlob += seq; e.seqBlob.push_back('\n'); e.seqBlob.push_back('\0');
        e.hdrOff.push_back(e.hdrBlob.size()); e.hdrLen.push_back((uint32_t) header.size() + 1);
        e.hdrBlob += header; e.hdrBlob.push_back('\0');
        e.file.push_back(fileIdx);
    }
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

// parsed reads in createdb's order as an in-memory sequence DB (blob in data-file layout): keys 0..n-1, wasExtended 0
bool readFastxAsDb(const std::vector<std::string> &files, bool shuffle, std::string &blob, std::vector<uint32_t> &key, std::vector<uint64_t> &off,
                   std::vector<uint32_t> &len, std::string *err) {
    Entries e;
    if (!parseAll(files, e, err)) return false;
    const std::vector<uint32_t> order = entryOrder(e.seqOff.size(), shuffle);
    blob.clear(); blob.reserve(e.seqBlob.size());
    key.resize(order.size()); off.resize(order.size()); len.resize(order.size());
    for (size_t j = 0; j < order.size(); j++) {
        const uint32_t i = order[j];
        key[j] = (uint32_t) j; off[j] = blob.size(); len[j] = e.seqLen[i];
        blob.append(e.seqBlob, e.seqOff[i], e.seqLen[i]);
    }
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
    std::string sBlob, hBlob; sBlob.reserve(e.seqBlob.size()); hBlob.reserve(e.hdrBlob.size());
    std::vector<uint32_t> key(n), sLen(n), hLen(n); std::vector<uint64_t> sOff(n), hOff(n); std::vector<uint8_t> ext(n, 0);
    std::string lookup;
    for (size_t j = 0; j < n; j++) {
        const uint32_t i = order[j];
        key[j] = (uint32_t) j;
        sOff[j] = sBlob.size(); sLen[j] = e.seqLen[i]; sBlob.append(e.seqBlob, e.seqOff[i], e.seqLen[i]);
        hOff[j] = hBlob.size(); hLen[j] = e.hdrLen[i]; hBlob.append(e.hdrBlob, e.hdrOff[i], e.hdrLen[i]);
        lookup += std::to_string(j); lookup.push_back('\t'); lookup += fastaId(e.hdrBlob.c_str() + e.hdrOff[i]); lookup.push_back('\t');
        lookup += std::to_string(e.file[i]); lookup.push_back('\n');
    }
    if (!mmdbWriteBlob(outPath, seqDbType, sBlob.data(), sBlob.size(), key.data(), sOff.data(), sLen.data(), ext.data(), n, err)) return 1;
    if (!mmdbWriteBlob(outPath + "_h", 12 /* DBTYPE_GENERIC_DB */, hBlob.data(), hBlob.size(), key.data(), hOff.data(), hLen.data(), ext.data(), n, err)) return 1;
    FILE *lf = fopen((outPath + ".lookup").c_str(), "w"), *sf = fopen((outPath + ".source").c_str(), "w");
    if (!lf || !sf) { *err = "Cannot open " + outPath + ".lookup for writing"; return 1; }
    fwrite(lookup.data(), 1, lookup.size(), lf); fclose(lf);
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
