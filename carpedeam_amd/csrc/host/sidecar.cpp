#include "sidecar.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <omp.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

bool sideEnabled() { static const bool on = !(getenv("CDM_SIDECAR") && !strcmp(getenv("CDM_SIDECAR"), "0")); return on; }

static bool statOf(const std::string &p, uint64_t *bytes, uint64_t *mtimeNs) {
    struct stat st;
    if (stat(p.c_str(), &st) != 0) return false;
    *bytes = (uint64_t) st.st_size; *mtimeNs = (uint64_t) st.st_mtim.tv_sec * 1000000000ull + (uint64_t) st.st_mtim.tv_nsec;
    return true;
}
// sizes and modification times of X (or X.0 .. X.n: sizes summed, the latest time) and X.index
bool sideStampOf(const std::string &db, SideStamp *st) {
    *st = SideStamp();
    uint64_t b = 0, t = 0;
    if (statOf(db, &b, &t)) { st->dataBytes = b; st->dataMtimeNs = t; }
    else {
        bool any = false;
        for (int i = 0; statOf(db + "." + std::to_string(i), &b, &t); i++) { st->dataBytes += b; st->dataMtimeNs = std::max(st->dataMtimeNs, t); any = true; }
        if (!any) return false;
    }
    return statOf(db + ".index", &st->indexBytes, &st->indexMtimeNs);
}
uint64_t sideKeyHash(const uint32_t *keys, size_t n) {
    uint64_t h = 0;
#pragma omp parallel for reduction(+ : h) schedule(static)
    for (size_t i = 0; i < n; i++) h += ((uint64_t) keys[i] + 0x9E3779B97F4A7C15ull) * (uint64_t) (2 * i + 1);
    return h ^ (uint64_t) n;
}
static uint64_t pad64(uint64_t b) { return (b + 63) & ~63ull; }
// In two steps, so that the sections can be written WHILE the text DB they belong to is still being formatted and written (other
// files: other inode locks): sideWriteBody puts the sections into X.cdmbin.tmp, sideCommit - once X's files are complete - stamps the
// header with their sizes and times, writes it and renames the file into place (complete or absent: a reader never sees half a side-car).
bool sideWriteBody(const std::string &db, uint32_t kind, uint32_t flags, uint64_t n, uint64_t count, uint64_t seqN, uint64_t seqKeyHash, int dbtype, const SidePiece *pieces, int nPieces, SideHeader *h) {
    if (!sideEnabled() || nPieces > SIDE_SECTIONS) return false;
    memset(h, 0, sizeof(*h));
    memcpy(h->magic, "CDMSIDE1", 8);
    h->kind = kind; h->flags = flags; h->n = n; h->count = count; h->seqN = seqN; h->seqKeyHash = seqKeyHash; h->dbtype = dbtype;
    for (int i = 0; i < nPieces; i++) h->section[i] = pieces[i].bytes;
    const std::string tmp = sidePath(db) + ".tmp";
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return false;
    bool ok = true;
    auto put = [&](const void *p, uint64_t bytes, uint64_t at) { const char *c = (const char *) p; while (bytes && ok) { const ssize_t w = pwrite(fd, c, bytes, (off_t) at); if (w <= 0) { ok = false; break; } c += w; bytes -= (uint64_t) w; at += (uint64_t) w; } };
    uint64_t at = pad64(sizeof(SideHeader));
    for (int i = 0; i < nPieces && ok; i++) { put(pieces[i].p, pieces[i].bytes, at); at += pad64(pieces[i].bytes); }
    if (ok && ftruncate(fd, (off_t) at) != 0) ok = false;
    ok = (close(fd) == 0) && ok;
    if (!ok) unlink(tmp.c_str());
    return ok;
}
bool sideCommit(const std::string &db, SideHeader *h) {
    const std::string tmp = sidePath(db) + ".tmp";
    bool ok = sideStampOf(db, &h->stamp);
    const int fd = ok ? open(tmp.c_str(), O_WRONLY) : -1;
    ok = ok && fd >= 0 && pwrite(fd, h, sizeof(*h), 0) == (ssize_t) sizeof(*h);
    if (fd >= 0) ok = (close(fd) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), sidePath(db).c_str()) == 0;
    if (!ok) unlink(tmp.c_str());
    return ok;
}
bool sideWrite(const std::string &db, uint32_t kind, uint32_t flags, uint64_t n, uint64_t count, uint64_t seqN, uint64_t seqKeyHash, int dbtype, const SidePiece *pieces, int nPieces) {
    SideHeader h;
    return sideWriteBody(db, kind, flags, n, count, seqN, seqKeyHash, dbtype, pieces, nPieces, &h) && sideCommit(db, &h);
}
SideFile::~SideFile() { if (base) munmap((void *) base, bytes); }
const void *SideFile::section(int i) const {
    uint64_t at = pad64(sizeof(SideHeader));
    for (int j = 0; j < i; j++) at += pad64(h->section[j]);
    return base + at;
}
bool sideOpen(const std::string &db, uint32_t kind, SideFile &f) {
    if (!sideEnabled()) return false;
    const int fd = open(sidePath(db).c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t) st.st_size < sizeof(SideHeader)) { close(fd); return false; }
    void *m = mmap(nullptr, (size_t) st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return false;
    f.base = (const char *) m; f.bytes = (size_t) st.st_size; f.h = (const SideHeader *) m;
    SideStamp now;
    uint64_t need = pad64(sizeof(SideHeader));
    for (int i = 0; i < SIDE_SECTIONS; i++) need += pad64(f.h->section[i]);
    const bool good = memcmp(f.h->magic, "CDMSIDE1", 8) == 0 && f.h->kind == kind && need <= f.bytes && sideStampOf(db, &now) &&
                      now.dataBytes == f.h->stamp.dataBytes && now.dataMtimeNs == f.h->stamp.dataMtimeNs && now.indexBytes == f.h->stamp.indexBytes && now.indexMtimeNs == f.h->stamp.indexMtimeNs;
    if (!good) { munmap(m, f.bytes); f.base = nullptr; f.h = nullptr; f.bytes = 0; return false; }      // (a side-car of other files: stale, or not one at all)
    return true;
}
