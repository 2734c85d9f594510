#include "mmdb.h"

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <numeric>
#include <sys/stat.h>

static bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
static void slurp(const std::string &p, std::string &out) {
    std::ifstream f(p, std::ios::binary);
    out.append(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

bool MmDb::load(const std::string &path, std::string *err) {
    if (exists(path)) slurp(path, data);
    else { int i = 0; for (; exists(path + "." + std::to_string(i)); i++) slurp(path + "." + std::to_string(i), data); if (i == 0) { *err = "Could not open data file " + path; return false; } }
    { std::ifstream t(path + ".dbtype", std::ios::binary); int32_t v = 0; if (t.good()) t.read((char *) &v, 4); dbtype = v; }
    std::ifstream ix(path + ".index");
    if (!ix.good()) { *err = "Could not open index file " + path + ".index"; return false; }
    struct E { uint32_t k; uint64_t o, l; uint8_t e; };
    std::vector<E> es; std::string line;
    while (std::getline(ix, line)) {
        unsigned long long k, o, l, x = 0;
        if (sscanf(line.c_str(), "%llu\t%llu\t%llu\t%llu", &k, &o, &l, &x) < 3) continue;
        es.push_back({(uint32_t) k, o, l, (uint8_t) x});
    }
    std::stable_sort(es.begin(), es.end(), [](const E &a, const E &b) { return a.k < b.k; });
    for (auto &e : es) {
        if (e.o + e.l > data.size()) { *err = "index entry beyond the data file in " + path; return false; }
        key.push_back(e.k); off.push_back(e.o); len.push_back(e.l); ext.push_back(e.e);
    }
    return true;
}
int64_t MmDb::idOf(uint32_t k) const {
    auto it = std::lower_bound(key.begin(), key.end(), k);
    return (it == key.end() || *it != k) ? -1 : (int64_t) (it - key.begin());
}
bool MmDbWriter::close(std::string *err) {
    FILE *d = fopen(path.c_str(), "wb"), *ix = fopen((path + ".index").c_str(), "w");
    if (!d || !ix) { *err = "Could not open " + path + " for writing"; return false; }
    std::vector<size_t> order(key.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
    uint64_t off = 0;
    for (size_t i : order) {
        fwrite(payload[i].data(), 1, payload[i].size(), d); fputc(0, d);
        fprintf(ix, "%u\t%llu\t%llu\t%u\n", key[i], (unsigned long long) off, (unsigned long long) payload[i].size() + 1, (unsigned) ext[i]);
        off += payload[i].size() + 1;
    }
    fclose(d); fclose(ix);
    FILE *t = fopen((path + ".dbtype").c_str(), "wb"); int32_t v = dbtype; fwrite(&v, 4, 1, t); fclose(t);
    return true;
}
