// The host-side modules of the final redundancy reduction of `ancient_assemble` (data/guidedNuclAssemble.sh:181-195 -> linclust,
// lib/mmseqs/data/workflow/linclust.sh:24-87): clust, createsubdb, filterdb, mergeclusters, result2repseq, and the two file modules the
// workflow scripts call between stages, rmdb and mvdb.  Plain host code (they work on cluster lists and DB indexes of the assembled
// contigs, a few thousand entries); each function follows the reference module it replaces and is compared with it on DB files
// (tests/test_cluster_modules.py) and through the whole workflow (tests/test_workflow.py).
//   clustModule           lib/mmseqs/src/clustering/Clustering.cpp:33-113, ClusteringAlgorithms.cpp:38-137,279-343 (--cluster-mode 2 / 3:
//                         greedy incremental, the mode linclust picks for --cov-mode 1, workflow/Linclust.cpp:66-72)
//   createsubdbModule     lib/mmseqs/src/util/createsubdb.cpp:10-107
//   filterdbModule        lib/mmseqs/src/util/filterdb.cpp:83-523, the --filter-file mode (:119-179, :389-404)
//   mergeclustersModule   lib/mmseqs/src/util/mergeclusters.cpp:13-153
//   result2repseqModule   lib/mmseqs/src/util/result2repseq.cpp:11-57
//   rmdbModule / mvdbModule   lib/mmseqs/src/commons/DBReader.cpp:1077-1114
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <string>
#include <vector>
#include <sys/stat.h>
#include <unistd.h>

#include "mmdb.h"

namespace {
bool exists(const std::string &p) { struct stat st; return lstat(p.c_str(), &st) == 0; }
// Util::parseKey: the first white-space delimited token of a line as an unsigned (fast_atoi: digits only)
inline uint32_t lineKey(const char *d) { uint32_t v = 0; while (*d >= '0' && *d <= '9') v = v * 10 + (uint32_t) (*d++ - '0'); return v; }
inline const char *skipLine(const char *d) { while (*d != '\n' && *d != '\0') d++; return *d == '\n' ? d + 1 : d; }
char *utoa(unsigned long long v, char *p) { char b[24]; int n = 0; do { b[n++] = (char) ('0' + v % 10); v /= 10; } while (v); while (n) *p++ = b[--n]; return p; }
// FileUtil::findDatafiles: X, or X.0 .. X.n
std::vector<std::string> dataFiles(const std::string &db) {
    std::vector<std::string> f;
    if (exists(db)) { f.push_back(db); return f; }
    for (int i = 0; exists(db + "." + std::to_string(i)); i++) f.push_back(db + "." + std::to_string(i));
    return f;
}
// FileUtil::symlinkAbs: an absolute link to the target's real path, replacing what is there
bool linkAbs(const std::string &target, const std::string &link, std::string *err) {
    if (exists(link)) unlink(link.c_str());
    char *t = realpath(target.c_str(), NULL);
    if (!t) { *err = "Could not get realpath of " + target + "!"; return false; }
    const bool ok = symlink(t, link.c_str()) == 0;
    free(t);
    if (!ok) *err = "Could not create symlink of " + target + "!";
    return ok;
}
// copyLinkDb(..., SEQUENCE_ANCILLARY, link): headers, look-up, source and taxonomy files of a sequence DB, where they exist
bool linkAncillary(const std::string &db, const std::string &out, std::string *err) {
    static const char *const SUFFIX[] = {"_h", "_h.index", "_h.dbtype", ".lookup", ".source", "_mapping", "_names.dmp", "_nodes.dmp", "_merged.dmp", "_taxonomy", NULL};
    for (int i = 0; SUFFIX[i]; i++) if (exists(db + SUFFIX[i]) && !linkAbs(db + SUFFIX[i], out + SUFFIX[i], err)) return false;
    return true;
}
bool writeDbtype(const std::string &path, int dbtype) {
    FILE *t = fopen((path + ".dbtype").c_str(), "wb");
    if (!t) return false;
    const int32_t v = dbtype; fwrite(&v, 4, 1, t); fclose(t);
    return true;
}
bool writeIndexFile(const std::string &path, const std::vector<uint32_t> &key, const std::vector<uint64_t> &off, const std::vector<uint64_t> &len, const std::vector<uint8_t> &ext) {
    FILE *f = fopen(path.c_str(), "w");
    if (!f) return false;
    char b[96];
    for (size_t i = 0; i < key.size(); i++) {       // DBWriter::indexToBuffer (DBWriter.cpp:415-427)
        char *p = utoa(key[i], b); *p++ = '\t'; p = utoa(off[i], p); *p++ = '\t'; p = utoa(len[i], p); *p++ = '\t'; p = utoa(ext[i], p); *p++ = '\n';
        fwrite(b, 1, (size_t) (p - b), f);
    }
    return fclose(f) == 0;
}
}  // namespace

// ---- clust: greedy incremental clustering.  Sequences in (length descending, index ascending) order - DBReader's SORT_BY_LENGTH with
// its total order (DBReader.cpp:301-318, DBReader.h:368-380); a sequence joins the first (= longest) sequence in whose list it
// stands, itself included; then every sequence somebody was assigned to becomes the representative of itself again
// (ClusteringAlgorithms.cpp:279-343).  Output: one entry per representative, key order, "rep\nmember\n..." with the members in key
// order (Clustering.cpp:82-113 on the sorted (rep key, member key) pairs), dbtype 6.
int clustModule(const std::string &seqPath, const std::string &alnPath, const std::string &outPath, int mode, std::string *err) {
    if (mode != 2 && mode != 3) { *err = "clust: only the greedy incremental modes (--cluster-mode 2 / 3: what linclust picks for --cov-mode 1) are implemented"; return 77; }
    MmDb seq, aln;
    if (!seq.load(seqPath, err, true) || !aln.load(alnPath, err)) return 1;
    const size_t n = seq.size();
    if (n != aln.size()) { *err = "Sequence db size != result db size"; return 1; }
    std::vector<uint32_t> order(n), rankOfIndex(n);          // local id (rank in the length order) <-> index in key order
    for (size_t i = 0; i < n; i++) order[i] = (uint32_t) i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return seq.len[a] != seq.len[b] ? seq.len[a] > seq.len[b] : a < b; });
    for (size_t r = 0; r < n; r++) rankOfIndex[order[r]] = (uint32_t) r;
    std::vector<uint32_t> assigned(n, UINT_MAX);
    for (size_t r = 0; r < n; r++) {                        // (the reference does this with an atomic minimum from many threads: same result)
        const uint32_t clusterKey = seq.key[order[r]];
        assigned[r] = std::min(assigned[r], (uint32_t) r);
        const int64_t a = aln.idOf(clusterKey);
        if (a < 0) { *err = "clust: key " + std::to_string(clusterKey) + " has no entry in the result DB"; return 1; }
        for (const char *d = aln.entry((size_t) a); *d != '\0'; d = skipLine(d)) {
            const int64_t e = seq.idOf(lineKey(d));
            if (e < 0) { *err = "Element " + std::to_string(lineKey(d)) + " contained in some alignment list, but not contained in the sequence database!"; return 1; }
            uint32_t &t = assigned[rankOfIndex[(size_t) e]];
            t = std::min(t, (uint32_t) r);
        }
    }
    for (size_t id = 0; id < n; id++) { const uint32_t a = assigned[id]; if (assigned[a] != a) assigned[a] = a; }
    std::vector<std::pair<uint32_t, uint32_t>> pairs(n);
    for (size_t r = 0; r < n; r++) pairs[r] = {seq.key[order[assigned[r]]], seq.key[order[r]]};
    std::sort(pairs.begin(), pairs.end());
    std::vector<OutChunk> chunks(1);
    OutChunk &c = chunks[0];
    std::string res; char b[24];
    for (size_t i = 0; i < n;) {
        const uint32_t rep = pairs[i].first;
        res.clear();
        res.append(b, (size_t) (utoa(rep, b) - b)); res.push_back('\n');
        for (; i < n && pairs[i].first == rep; i++) if (pairs[i].second != rep) { res.append(b, (size_t) (utoa(pairs[i].second, b) - b)); res.push_back('\n'); }
        c.add(rep, res.data(), res.size(), 0);
    }
    return mmdbWriteChunks(outPath, 6, chunks, err) ? 0 : 1;
}

// ---- createsubdb <order file | DB whose index is the order> <in DB> <out DB>: the entries of the listed keys.  --subdb-mode 1: a new
// index over the linked data file(s); 0: the entries copied.  The index is written in the order of the list and sorted by key if
// the list was not in key order (DBWriter::close(merge, !isOrdered)).  Keys the DB does not hold are skipped with a warning.
int createsubdbModule(const std::string &orderArg, const std::string &inPath, const std::string &outPath, int subDbMode, std::string *err) {
    const std::string orderFile = exists(orderArg + ".index") ? orderArg + ".index" : orderArg;
    FILE *of = fopen(orderFile.c_str(), "r");
    if (!of) { *err = "File " + orderArg + " does not exist."; return 1; }
    MmDb in;
    if (!in.load(inPath, err)) { fclose(of); return 1; }
    std::vector<uint32_t> key; std::vector<uint64_t> off, len; std::vector<uint8_t> ext;
    std::vector<OutChunk> chunks(1);
    char *line = NULL; size_t cap = 0;
    uint32_t prev = 0; bool ordered = true; uint64_t at = 0;
    while (getline(&line, &cap, of) != -1) {
        const uint32_t k = lineKey(line);
        ordered = ordered && prev <= k; prev = k;
        const int64_t id = in.idOf(k);
        if (id < 0) { fprintf(stderr, "Key %u not found in database\n", k); continue; }
        key.push_back(k); len.push_back(in.len[(size_t) id]); ext.push_back(in.ext[(size_t) id]);
        if (subDbMode == 1) off.push_back(in.off[(size_t) id]);
        else { off.push_back(at); chunks[0].data.insert(chunks[0].data.end(), in.entry((size_t) id), in.entry((size_t) id) + in.len[(size_t) id]); at += in.len[(size_t) id]; }
    }
    free(line); fclose(of);
    if (!ordered) {         // DBWriter::sortIndex through DBReader's sort by key; equal keys keep their order
        std::vector<size_t> perm(key.size());
        for (size_t i = 0; i < perm.size(); i++) perm[i] = i;
        std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
        std::vector<uint32_t> k2; std::vector<uint64_t> o2, l2; std::vector<uint8_t> e2;
        for (size_t i : perm) { k2.push_back(key[i]); o2.push_back(off[i]); l2.push_back(len[i]); e2.push_back(ext[i]); }
        key.swap(k2); off.swap(o2); len.swap(l2); ext.swap(e2);
    }
    if (!writeIndexFile(outPath + ".index", key, off, len, ext)) { *err = "Could not write " + outPath + ".index"; return 1; }
    for (const std::string &f : dataFiles(outPath)) unlink(f.c_str());
    if (subDbMode == 1) {
        const std::vector<std::string> names = dataFiles(inPath);
        if (names.size() == 1) { if (!linkAbs(names[0], outPath, err)) return 1; }
        else for (const std::string &nm : names) if (!linkAbs(nm, outPath + nm.substr(nm.rfind('.')), err)) return 1;
    } else {
        FILE *d = fopen(outPath.c_str(), "wb");
        if (!d || (chunks[0].data.size() && fwrite(chunks[0].data.data(), 1, chunks[0].data.size(), d) != chunks[0].data.size()) || fclose(d) != 0) { *err = "Could not write " + outPath; return 1; }
    }
    if (!writeDbtype(outPath, in.dbtype)) { *err = "Could not write " + outPath + ".dbtype"; return 1; }
    return linkAncillary(inPath, outPath, err) ? 0 : 1;
}

// ---- filterdb --filter-file: of every entry the lines whose first column stands in the file (first column of its lines, NULs skipped:
// a DB data file may serve as the list); entries that lose all their lines stay, empty.
int filterdbModule(const std::string &inPath, const std::string &outPath, const std::string &filterFile, std::string *err) {
    std::vector<std::string> names;
    if (exists(filterFile)) names.push_back(filterFile);
    else if (exists(filterFile + ".dbtype")) names = dataFiles(filterFile);
    else { *err = "File " + filterFile + " does not exist"; return 1; }
    std::vector<std::string> filter;
    for (const std::string &nm : names) {
        FILE *f = fopen(nm.c_str(), "r");
        if (!f) { *err = "File " + nm + " does not exist"; return 1; }
        std::string k; bool inKey = true; int c;
        while ((c = fgetc(f)) != EOF) {
            if (c == '\n') { if (!k.empty()) { filter.push_back(k); k.clear(); } inKey = true; continue; }
            if (c == ' ' || c == '\t') { inKey = false; continue; }
            if (c == '\0' || !inKey) continue;
            k.push_back((char) c);
            if (k.size() == 65536) { fclose(f); *err = "Input in file " + nm + " too long"; return 1; }
        }
        if (inKey && !k.empty()) filter.push_back(k);
        fclose(f);
    }
    std::sort(filter.begin(), filter.end());
    filter.erase(std::unique(filter.begin(), filter.end()), filter.end());
    MmDb in;
    if (!in.load(inPath, err)) return 1;
    std::vector<OutChunk> chunks(1);
    std::string buf, col;
    for (size_t i = 0; i < in.size(); i++) {
        buf.clear();
        for (const char *d = in.entry(i); *d != '\0';) {
            const char *e = d; while (*e != '\n' && *e != '\0') e++;
            const char *w = d; while (w < e && (*w == ' ' || *w == '\t')) w++;           // Util::getWordsOfLine skips leading white space
            const char *we = w; while (we < e && *we != ' ' && *we != '\t') we++;
            col.assign(w, (size_t) (we - w));
            if (std::binary_search(filter.begin(), filter.end(), col)) { buf.append(d, (size_t) (e - d)); buf.push_back('\n'); }
            d = *e == '\n' ? e + 1 : e;
        }
        chunks[0].add(in.key[i], buf.data(), buf.size(), 0);
    }
    return mmdbWriteChunks(outPath, in.dbtype, chunks, err) ? 0 : 1;
}

// ---- mergeclusters <seq DB> <out> <clustering 1> <clustering 2> ...: the members of a later step's cluster bring the clusters they
// headed in the steps before (std::list::splice: theirs are emptied); entries in key order of the sequence DB, empty ones left out
int mergeclustersModule(const std::string &seqPath, const std::string &outPath, const std::vector<std::string> &steps, std::string *err) {
    MmDb seq;
    if (!seq.load(seqPath, err, true)) return 1;
    if (steps.empty()) { *err = "mergeclusters: no clustering given"; return 1; }
    std::vector<std::list<uint32_t>> merged(seq.size());
    auto idOf = [&](uint32_t k, int64_t &id) { id = seq.idOf(k); if (id < 0) *err = "mergeclusters: key " + std::to_string(k) + " is not in the sequence DB"; return id >= 0; };
    for (size_t s = 0; s < steps.size(); s++) {
        MmDb clu;
        if (!clu.load(steps[s], err)) return 1;
        for (size_t i = 0; i < clu.size(); i++) {
            int64_t cluId; if (!idOf(clu.key[i], cluId)) return 1;
            for (const char *d = clu.entry(i); *d != '\0'; d = skipLine(d)) {
                int64_t seqId; if (!idOf(lineKey(d), seqId)) return 1;
                if (s == 0) merged[(size_t) cluId].push_back((uint32_t) seqId);
                else if (seqId != cluId) merged[(size_t) cluId].splice(merged[(size_t) cluId].end(), merged[(size_t) seqId]);
            }
        }
    }
    std::vector<OutChunk> chunks(1);
    std::string res; char b[24];
    for (size_t i = 0; i < seq.size(); i++) {
        if (merged[i].empty()) continue;
        res.clear();
        for (uint32_t m : merged[i]) { res.append(b, (size_t) (utoa(seq.key[m], b) - b)); res.push_back('\n'); }
        chunks[0].add(seq.key[i], res.data(), res.size(), 0);
    }
    return mmdbWriteChunks(outPath, 6, chunks, err) ? 0 : 1;
}

// ---- result2repseq <seq DB> <cluster DB> <out>: per cluster entry the sequence its first line names, under the entry's key
int result2repseqModule(const std::string &seqPath, const std::string &cluPath, const std::string &outPath, std::string *err) {
    MmDb seq, clu;
    if (!seq.load(seqPath, err) || !clu.load(cluPath, err)) return 1;
    std::vector<OutChunk> chunks(1);
    for (size_t i = 0; i < clu.size(); i++) {
        const char *d = clu.entry(i);
        if (*d == '\0') continue;
        const int64_t id = seq.idOf(lineKey(d));
        if (id < 0) { *err = "result2repseq: key " + std::to_string(lineKey(d)) + " is not in the sequence DB"; return 1; }
        chunks[0].add(clu.key[i], seq.entry((size_t) id), seq.len[(size_t) id] ? seq.len[(size_t) id] - 1 : 0, 0);      // (DBWriter::writeData: wasExtended defaults to 0)
    }
    if (!mmdbWriteChunks(outPath, seq.dbtype, chunks, err)) return 1;
    return linkAncillary(seqPath, outPath, err) ? 0 : 1;
}

// ---- rmdb / mvdb: DBReader::removeDb / moveDb
int rmdbModule(const std::string &db) {
    for (const std::string &f : dataFiles(db)) unlink(f.c_str());
    for (const char *s : {".index", ".dbtype", ".source", ".lookup", ".cdmbin"}) if (exists(db + s)) unlink((db + s).c_str());      // (.cdmbin: the binary side-car, host/sidecar.h)
    return 0;
}
int mvdbModule(const std::string &src, const std::string &dst, std::string *err) {
    const std::vector<std::string> files = dataFiles(src);
    auto mv = [&](const std::string &a, const std::string &b) { if (rename(a.c_str(), b.c_str()) != 0) { *err = "Could not move " + a + " to " + b; return false; } return true; };
    if (files.size() == 1) { if (!mv(files[0], dst)) return 1; }
    else for (const std::string &f : files) if (!mv(f, dst + f.substr(f.rfind('.')))) return 1;
    for (const char *s : {".index", ".dbtype", ".lookup", ".cdmbin"}) if (exists(src + s) && !mv(src + s, dst + s)) return 1;      // (a rename keeps sizes and times: the side-car's stamp still fits)
    return 0;
}
