#include "mmdb.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <fstream>
#include <numeric>
#include <omp.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
static void slurp(const std::string &p, std::string &out) {
    std::ifstream f(p, std::ios::binary);
    out.append(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}
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

MmDb::~MmDb() { if (mapped) munmap(mapped, mappedBytes); }

bool MmDb::load(const std::string &path, std::string *err, bool indexOnly) {
    if (indexOnly) { owned.assign(1, '\0'); base = owned.data(); bytes = 0; }
    else if (exists(path)) {
        size_t sz = 0;
        mapped = mapFile(path, &sz);
        if (!mapped && sz) { *err = "Could not open data file " + path; return false; }
        mappedBytes = sz; base = (const char *) mapped; bytes = sz;
        if (!base) { owned.assign(1, '\0'); base = owned.data(); bytes = 0; }
    } else {
        int i = 0;
        for (; exists(path + "." + std::to_string(i)); i++) slurp(path + "." + std::to_string(i), owned);
        if (i == 0) { *err = "Could not open data file " + path; return false; }
        base = owned.data(); bytes = owned.size();
    }
    { std::ifstream t(path + ".dbtype", std::ios::binary); int32_t v = 0; if (t.good()) t.read((char *) &v, 4); dbtype = v; }
    size_t ixBytes = 0;
    if (!exists(path + ".index")) { *err = "Could not open index file " + path + ".index"; return false; }
    void *ixMap = mapFile(path + ".index", &ixBytes);
    if (!ixMap && ixBytes) { *err = "Could not open index file " + path + ".index"; return false; }
    const char *ix = (const char *) ixMap;
    // every thread parses the lines that START in its slice of the file
    struct E { uint32_t k; uint64_t o, l; uint8_t e; };
    const int T = std::max(1, omp_get_max_threads());
    std::vector<std::vector<E>> parts(T);
    std::vector<size_t> bound(T + 1, ixBytes);       // slice t = the lines that start in [bound[t], bound[t + 1]); bounds sit on line starts
    for (int t = 0; t < T; t++) {
        size_t b = ixBytes * (size_t) t / T;
        while (b > 0 && b < ixBytes && ix[b - 1] != '\n') b++;
        bound[t] = b;
    }
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        const size_t hi = bound[t + 1];
        std::vector<E> &es = parts[t];
        size_t p = bound[t];
        while (p < hi) {
            unsigned long long v[4] = {0, 0, 0, 0}; int f = 0;
            while (p < ixBytes && ix[p] != '\n') {
                if (ix[p] >= '0' && ix[p] <= '9') { if (f < 4) v[f] = v[f] * 10 + (unsigned) (ix[p] - '0'); }
                else if (ix[p] == '\t') f++;
                p++;
            }
            p++;
            if (f >= 2) es.push_back({(uint32_t) v[0], v[1], v[2], (uint8_t) v[3]});
        }
    }
    if (ixMap) munmap(ixMap, ixBytes);
    size_t n = 0;
    for (auto &v : parts) n += v.size();
    std::vector<E> es; es.reserve(n);
    for (auto &v : parts) es.insert(es.end(), v.begin(), v.end());
    bool sorted = true;
    for (size_t i = 1; i < n && sorted; i++) sorted = es[i - 1].k <= es[i].k;
    if (!sorted) std::stable_sort(es.begin(), es.end(), [](const E &a, const E &b) { return a.k < b.k; });
    key.resize(n); off.resize(n); len.resize(n); ext.resize(n);
    for (size_t i = 0; i < n; i++) {
        if (!indexOnly && es[i].o + es[i].l > bytes) { *err = "index entry beyond the data file in " + path; return false; }
        key[i] = es[i].k; off[i] = es[i].o; len[i] = es[i].l; ext[i] = es[i].e;
    }
    return true;
}
int64_t MmDb::idOf(uint32_t k) const {
    if (k < key.size() && key[k] == k) return (int64_t) k;      // dense keys 0..n-1 (what createdb writes): no search
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
bool mmdbWriteChunks(const std::string &path, int dbtype, const std::vector<OutChunk> &chunks, std::string *err) {
    FILE *d = fopen(path.c_str(), "wb"), *ix = fopen((path + ".index").c_str(), "w");
    if (!d || !ix) { *err = "Could not open " + path + " for writing"; return false; }
    std::vector<uint64_t> base(chunks.size() + 1, 0);
    for (size_t c = 0; c < chunks.size(); c++) base[c + 1] = base[c] + chunks[c].data.size();
    std::vector<std::string> ixText(chunks.size());
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t c = 0; c < chunks.size(); c++) indexText(ixText[c], chunks[c].key.data(), chunks[c].len.data(), chunks[c].ext.data(), chunks[c].key.size(), base[c]);
    for (size_t c = 0; c < chunks.size(); c++) {
        if (!chunks[c].data.empty()) fwrite(chunks[c].data.data(), 1, chunks[c].data.size(), d);
        if (!ixText[c].empty()) fwrite(ixText[c].data(), 1, ixText[c].size(), ix);
    }
    const bool ok = fclose(d) == 0 && fclose(ix) == 0 && writeDbtype(path, dbtype);
    if (!ok) *err = "Could not write " + path;
    return ok;
}
bool mmdbWriteBlob(const std::string &path, int dbtype, const char *blob, size_t blobBytes, const std::vector<uint32_t> &key, const std::vector<uint64_t> &off,
                   const std::vector<uint32_t> &len, const std::vector<uint8_t> &ext, std::string *err) {
    FILE *d = fopen(path.c_str(), "wb"), *ix = fopen((path + ".index").c_str(), "w");
    if (!d || !ix) { *err = "Could not open " + path + " for writing"; return false; }
    if (blobBytes) fwrite(blob, 1, blobBytes, d);
    const size_t n = key.size();
    const int T = std::max(1, omp_get_max_threads());
    std::vector<std::string> ixText(T);
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        const size_t lo = n * (size_t) t / T, hi = n * (size_t) (t + 1) / T;
        if (hi > lo) indexText(ixText[t], key.data() + lo, len.data() + lo, ext.data() + lo, hi - lo, off[lo]);
    }
    for (auto &s : ixText) if (!s.empty()) fwrite(s.data(), 1, s.size(), ix);
    const bool ok = fclose(d) == 0 && fclose(ix) == 0 && writeDbtype(path, dbtype);
    if (!ok) *err = "Could not write " + path;
    return ok;
}
