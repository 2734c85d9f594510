#include "mmdb.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <fstream>
#include <new>
#include <numeric>
#include <omp.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

static bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
static size_t fileSize(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 ? (size_t) st.st_size : 0; }

static const size_t HUGE_MIN = 4u << 20;     // below this the C library's allocator is as good
void *hugeAlloc(size_t bytes) {
    if (bytes < HUGE_MIN) { void *p = calloc(bytes ? bytes : 1, 1); if (!p) throw std::bad_alloc(); return p; }    // zeroed like fresh pages: one contract for every size
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) throw std::bad_alloc();
    static const bool plain = getenv("CDM_NO_HUGEPAGE") != nullptr;      // (A/B switch for measurements)
    if (!plain) madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}
void hugeFree(void *p, size_t bytes) { if (bytes < HUGE_MIN) free(p); else munmap(p, bytes); }

static void *mapFile(const std::string &p, size_t *bytes) {
    const int fd = open(p.c_str(), O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return nullptr; }
    *bytes = (size_t) st.st_size;
    void *m = *bytes ? mmap(nullptr, *bytes, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    close(fd);
    return (m == MAP_FAILED) ? nullptr : m;
}
// a whole file into dst, read by all threads in 8 MB pieces
static bool readInto(const std::string &p, char *dst, size_t bytes) {
    const int fd = open(p.c_str(), O_RDONLY);
    if (fd < 0) return false;
    const size_t piece = 8u << 20, pieces = (bytes + piece - 1) / piece;
    bool ok = true;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < pieces; i++) {
        size_t at = i * piece; const size_t end = std::min(bytes, at + piece);
        while (at < end) { const ssize_t r = pread(fd, dst + at, end - at, (off_t) at); if (r <= 0) { ok = false; break; } at += (size_t) r; }
    }
    close(fd);
    return ok;
}

MmDb::~MmDb() { if (mapped) munmap(mapped, mappedBytes); }

bool MmDb::load(const std::string &path, std::string *err, bool indexOnly) {
    if (indexOnly) { owned.assign(1, '\0'); base = owned.data(); bytes = 0; }
    else if (exists(path)) {
        size_t sz = 0;
        mapped = mapFile(path, &sz);
        if (!mapped && sz) { *err = "Could not open data file " + path; return false; }
        mappedBytes = sz; base = (const char *) mapped; bytes = sz;
        if (!base) { owned.assign(1, '\0'); base = owned.data(); bytes = 0; }
    } else {            // split data files X.0 .. X.n (DBWriter's per-thread files left unmerged): offsets are global over their concatenation
        std::vector<std::string> parts; std::vector<size_t> at(1, 0);
        for (int i = 0; exists(path + "." + std::to_string(i)); i++) { parts.push_back(path + "." + std::to_string(i)); at.push_back(at.back() + fileSize(parts.back())); }
        if (parts.empty()) { *err = "Could not open data file " + path; return false; }
        owned.resize(at.back() + 1);
        owned[at.back()] = '\0';      // the sentinel behind the last entry
        for (size_t i = 0; i < parts.size(); i++) if (!readInto(parts[i], owned.data() + at[i], at[i + 1] - at[i])) { *err = "Could not read data file " + parts[i]; return false; }
        base = owned.data(); bytes = at.back();
    }
    { std::ifstream t(path + ".dbtype", std::ios::binary); int32_t v = 0; if (t.good()) t.read((char *) &v, 4); dbtype = v; }
    size_t ixBytes = 0;
    if (!exists(path + ".index")) { *err = "Could not open index file " + path + ".index"; return false; }
    void *ixMap = mapFile(path + ".index", &ixBytes);
    if (!ixMap && ixBytes) { *err = "Could not open index file " + path + ".index"; return false; }
    const char *ix = (const char *) ixMap;
    // every thread takes the lines that START in its slice of the file: counted first, then parsed straight into the columns
    const int T = std::max(1, omp_get_max_threads());
    std::vector<size_t> bound(T + 1, ixBytes), first(T + 1, 0);       // slice t = the lines that start in [bound[t], bound[t + 1]); bounds sit on line starts
    for (int t = 0; t < T; t++) {
        size_t b = ixBytes * (size_t) t / T;
        while (b > 0 && b < ixBytes && ix[b - 1] != '\n') b++;
        bound[t] = b;
    }
    bool sorted = true, inside = true;
    // a line is an entry when it has its three columns (DBReader skips anything shorter: a blank or cut-off line of a hand-made index);
    // the slices are a loop, not a team: a smaller team than asked for still visits every slice
    auto isEntry = [&](size_t p, size_t *next) {
        int tabs = 0;
        while (p < ixBytes && ix[p] != '\n') { tabs += ix[p] == '\t'; p++; }
        *next = p + 1;
        return tabs >= 2;
    };
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < T; t++) {
        size_t lines = 0, nx = 0;
        for (size_t p = bound[t]; p < bound[t + 1]; p = nx) lines += isEntry(p, &nx);
        first[t + 1] = lines;
    }
    for (int i = 0; i < T; i++) first[i + 1] += first[i];
    key.resize(first[T]); off.resize(first[T]); len.resize(first[T]); ext.resize(first[T]);
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < T; t++) {
        const size_t hi = bound[t + 1];
        size_t p = bound[t], at = first[t];
        bool ok = true, in = true;
        while (p < hi) {
            unsigned long long v[4] = {0, 0, 0, 0}; int f = 0;
            while (p < ixBytes && ix[p] != '\n') {
                if (ix[p] >= '0' && ix[p] <= '9') { if (f < 4) v[f] = v[f] * 10 + (unsigned) (ix[p] - '0'); }
                else if (ix[p] == '\t') f++;
                p++;
            }
            p++;
            if (f < 2) continue;
            key[at] = (uint32_t) v[0]; off[at] = v[1]; len[at] = v[2]; ext[at] = (uint8_t) v[3];
            if (at > first[t] && key[at - 1] > key[at]) ok = false;
            if (v[1] + v[2] > bytes) in = false;
            at++;
        }
        if (!ok) {
#pragma omp atomic write
            sorted = false;
        }
        if (!in) {
#pragma omp atomic write
            inside = false;
        }
    }
    if (ixMap) munmap(ixMap, ixBytes);
    const size_t n = key.size();
    for (int t = 1; t < T && sorted; t++) if (first[t] > 0 && first[t] < n && key[first[t] - 1] > key[first[t]]) sorted = false;
    if (!indexOnly && !inside) { *err = "index entry beyond the data file in " + path; return false; }
    if (!sorted) {      // DBReader::sortIndex: by key, entries of equal keys in file order
        std::vector<uint32_t> perm(n);
        for (size_t i = 0; i < n; i++) perm[i] = (uint32_t) i;
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        HVec<uint32_t> k2(n); HVec<uint64_t> o2(n), l2(n); HVec<uint8_t> e2(n);
        for (size_t i = 0; i < n; i++) { k2[i] = key[perm[i]]; o2[i] = off[perm[i]]; l2[i] = len[perm[i]]; e2[i] = ext[perm[i]]; }
        key.swap(k2); off.swap(o2); len.swap(l2); ext.swap(e2);
    }
    noteDense();
    return true;
}
void MmDb::adoptIndex(const uint32_t *keys, const uint32_t *payLen, const uint8_t *extFlags, size_t n, int type) {
    key.resize(n); off.resize(n); len.resize(n); ext.resize(n);
    dbtype = type;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) { key[i] = keys[i]; len[i] = (uint64_t) payLen[i] + 2; ext[i] = extFlags ? extFlags[i] : 0; }
    uint64_t at = 0;
    for (size_t i = 0; i < n; i++) { off[i] = at; at += len[i]; }
    owned.assign(1, '\0'); base = owned.data(); bytes = 0;
    noteDense();
}
void MmDb::noteDense() {
    const size_t n = key.size();
    bool all = true;
#pragma omp parallel for reduction(&& : all) schedule(static)
    for (size_t i = 0; i < n; i++) all = all && key[i] == (uint32_t) i;
    dense = all && n > 0;
}
int64_t MmDb::idOfSearch(uint32_t k) const {
    if (k < key.size() && key[k] == k) return (int64_t) k;      // (mostly dense keys)
    auto it = std::lower_bound(key.begin(), key.end(), k);
    return (it == key.end() || *it != k) ? -1 : (int64_t) (it - key.begin());
}

static char *utoaFast(unsigned long long v, char *p) { char b[24]; int n = 0; do { b[n++] = (char) ('0' + v % 10); v /= 10; } while (v); while (n) *p++ = b[--n]; return p; }
static bool writeDbtype(const std::string &path, int dbtype) {
    FILE *t = fopen((path + ".dbtype").c_str(), "wb");
    if (!t) return false;
    int32_t v = dbtype; fwrite(&v, 4, 1, t); fclose(t);
    return true;
}
// index lines of entries [0, n) with offsets base + running sum of len
static void indexText(std::string &out, const uint32_t *key, const uint32_t *len, const uint8_t *ext, size_t n, uint64_t base) {
    out.reserve(n * 28);
    char b[96];
    for (size_t i = 0; i < n; i++) {
        char *p = utoaFast(key[i], b); *p++ = '\t'; p = utoaFast(base, p); *p++ = '\t'; p = utoaFast(len[i], p); *p++ = '\t'; p = utoaFast(ext[i], p); *p++ = '\n';
        out.append(b, p - b);
        base += len[i];
    }
}
// Output files are written by all threads with pwrite() in 8 MB blocks (about 2.8 GB/s on the GPU boxes' overlay file system;
// filling a shared mapping of the pre-sized file instead was measured and is slower there: 1.0-1.4 s against 0.75 s for 1.6 GB).
static bool pwriteAll(int fd, const char *p, size_t n, uint64_t at) {
    while (n) { const ssize_t w = pwrite(fd, p, n, (off_t) at); if (w <= 0) return false; p += w; n -= (size_t) w; at += (uint64_t) w; }
    return true;
}
struct Piece { const char *p; size_t n; uint64_t at; };
// A file in RAM (tmpfs: /dev/shm, where the workflow's tmp directory belongs when there is room) takes ONE writer best: every write()
// to a file holds its inode lock, and 16 threads handing that lock round in 8 MB blocks reach 3.1 GB/s where one thread writing the
// pieces through reaches 6.1 (scripts/probes/shm_write.cpp, profiles/r05_probe_shm_write.txt; a pre-allocated shared mapping filled by
// all threads: the same 6.1).  CDM_WRITE_MODE=parallel|serial pins either.
#include <sys/vfs.h>
static bool oneWriterIsFaster(int fd) {
    static const char *mode = getenv("CDM_WRITE_MODE");
    if (mode) return !strcmp(mode, "serial");
    struct statfs st;
    return fstatfs(fd, &st) == 0 && (unsigned long) st.f_type == 0x01021994ul;       // TMPFS_MAGIC
}
static bool writePieces(int fd, uint64_t total, const std::vector<Piece> &pieces) {
    if (total == 0) return true;
    if (oneWriterIsFaster(fd)) {
        for (const Piece &pc : pieces) if (!pwriteAll(fd, pc.p, pc.n, pc.at)) return false;
        return true;
    }
    const size_t BLOCK = 8u << 20;
    std::vector<Piece> blocks;
    for (const Piece &pc : pieces) for (size_t o = 0; o < pc.n; o += BLOCK) blocks.push_back({pc.p + o, std::min(BLOCK, pc.n - o), pc.at + o});
    bool ok = true;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < blocks.size(); i++) {
        if (!pwriteAll(fd, blocks[i].p, blocks[i].n, blocks[i].at)) {
#pragma omp atomic write
            ok = false;
        }
    }
    return ok;
}
bool mmdbWriteChunks(const std::string &path, int dbtype, const std::vector<OutChunk> &chunks, std::string *err, bool splitData) {
    const size_t C = chunks.size();
    std::vector<uint64_t> base(C + 1, 0), ixBase(C + 1, 0);
    for (size_t c = 0; c < C; c++) base[c + 1] = base[c] + chunks[c].data.size();
    // Result DBs (prefilter hits, alignments) of some size go out as the reference's DBWriter leaves them: one data file per writer
    // thread, X.0 .. X.T-1, the index offsets global over their concatenation (DBWriter.cpp:135-188; DBReader.cpp:108-133 reads them
    // back, as host/mmdb.cpp does).  Buffered writes to ONE file are serialised by the file system whatever the threads; to T files
    // they are not.  Sequence DBs stay one file: the workflow scripts test and link their data file by name.
    static const uint64_t splitMin = getenv("CDM_SPLIT_MIN") ? strtoull(getenv("CDM_SPLIT_MIN"), nullptr, 10) : (64u << 20);     // (tests lower it)
    const bool split = splitData && C > 1 && base[C] >= splitMin && !getenv("CDM_SINGLE_DATA_FILE");
    const int ix = open((path + ".index").c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    const int d = split ? -1 : open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (ix < 0 || (!split && d < 0)) { if (d >= 0) close(d); if (ix >= 0) close(ix); *err = "Could not open " + path + " for writing"; return false; }
    bool ok = true;
    if (split) unlink(path.c_str());                          // (a single-file DB of an earlier run must not shadow the parts)
    else for (int i = 0; i < 4096; i++) { const std::string part = path + "." + std::to_string(i); if (unlink(part.c_str()) != 0) break; }
    std::vector<std::string> ixText(C);
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t c = 0; c < C; c++) {
        indexText(ixText[c], chunks[c].key.data(), chunks[c].len.data(), chunks[c].ext.data(), chunks[c].key.size(), base[c]);
        if (split) {
            const int f = open((path + "." + std::to_string(c)).c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
            bool mine = f >= 0 && (chunks[c].data.empty() || pwriteAll(f, chunks[c].data.data(), chunks[c].data.size(), 0));
            if (f >= 0) mine = (close(f) == 0) && mine;
            if (!mine) {
#pragma omp atomic write
                ok = false;
            }
        }
    }
    if (split) for (size_t c = C; c < 4096; c++) { const std::string part = path + "." + std::to_string(c); if (unlink(part.c_str()) != 0) break; }
    std::vector<Piece> dataPieces, ixPieces;
    for (size_t c = 0; c < C; c++) {
        ixBase[c + 1] = ixBase[c] + ixText[c].size();
        if (!split && !chunks[c].data.empty()) dataPieces.push_back({chunks[c].data.data(), chunks[c].data.size(), base[c]});
        if (!ixText[c].empty()) ixPieces.push_back({ixText[c].data(), ixText[c].size(), ixBase[c]});
    }
    {
        bool okIx = true;
        std::thread ixWriter([&] { okIx = writePieces(ix, ixBase[C], ixPieces); });
        if (!split) ok = writePieces(d, base[C], dataPieces) && ok;
        ixWriter.join();
        ok = ok && okIx;
    }
    if (d >= 0) ok = (close(d) == 0) && ok;
    ok = (close(ix) == 0) && ok;
    ok = ok && writeDbtype(path, dbtype);
    if (!ok) *err = "Could not write " + path;
    return ok;
}
// The data file of a DB that the caller fills piece by piece as the pieces come off the device (cdm_seqdb_download_stream): opened and
// sized here (the tail the pieces do not cover - the last entry's NUL - is the file's zero fill), -1 where that does not pay: the
// pieces are written by ONE thread, which is the fast way on tmpfs and the slow one elsewhere (CDM_STREAM_DB=0|1 pins either).
int mmdbOpenStreamedData(const std::string &path, size_t bytes) {
    static const char *sw = getenv("CDM_STREAM_DB");
    if ((sw && !strcmp(sw, "0")) || bytes == 0) return -1;
    const int d = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (d < 0) return -1;
    if ((!oneWriterIsFaster(d) && !(sw && !strcmp(sw, "1"))) || ftruncate(d, (off_t) bytes) != 0) { close(d); return -1; }
    for (int i = 0; i < 4096; i++) { const std::string part = path + "." + std::to_string(i); if (unlink(part.c_str()) != 0) break; }
    return d;
}
bool mmdbWritePiece(int fd, const char *data, uint64_t offset, uint64_t bytes) { return pwriteAll(fd, data, bytes, offset); }
// blob == NULL: the data file is in place already (dataFd: mmdbOpenStreamedData's, closed here) - index and dbtype only
bool mmdbWriteBlob(const std::string &path, int dbtype, const char *blob, size_t blobBytes, const uint32_t *key, const uint64_t *off,
                   const uint32_t *len, const uint8_t *ext, size_t n, std::string *err, int dataFd) {
    const bool dataElsewhere = !blob && dataFd == MMDB_DATA_ELSEWHERE;      // (the caller writes and closes the data file itself, meanwhile)
    const int d = blob ? open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644) : dataFd, ix = open((path + ".index").c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    if ((d < 0 && !dataElsewhere) || ix < 0) { if (d >= 0) close(d); if (ix >= 0) close(ix); *err = "Could not open " + path + " for writing"; return false; }
    const int T = std::max(1, omp_get_max_threads());
    std::vector<std::string> ixText(T);
    std::vector<uint64_t> ixBase(T + 1, 0);
#pragma omp parallel for schedule(static, 1) num_threads(T)     // a loop over the slices: a smaller team still visits them all
    for (int t = 0; t < T; t++) {
        const size_t lo = n * (size_t) t / T, hi = n * (size_t) (t + 1) / T;
        if (hi > lo) indexText(ixText[t], key + lo, len + lo, ext + lo, hi - lo, off[lo]);
    }
    std::vector<Piece> ixPieces;
    for (int t = 0; t < T; t++) { ixBase[t + 1] = ixBase[t] + ixText[t].size(); if (!ixText[t].empty()) ixPieces.push_back({ixText[t].data(), ixText[t].size(), ixBase[t]}); }
    // (two files, two inode locks: the index goes out beside the data)
    bool okIx = true;
    std::thread ixWriter([&] { okIx = writePieces(ix, ixBase[T], ixPieces); });
    bool ok = blob ? writePieces(d, blobBytes, std::vector<Piece>(1, Piece{blob, blobBytes, 0})) : true;
    ixWriter.join();
    ok = ok && okIx;
    ok = (dataElsewhere || close(d) == 0) & (close(ix) == 0) & ok;
    ok = ok && writeDbtype(path, dbtype);
    if (!ok) *err = "Could not write " + path;
    return ok;
}
