// MMseqs2 on-disk DB access for the host binary (format as modified by the CarpeDeam fork).
//   index  : "key \t offset \t length \t wasExtended \n"   lib/mmseqs/src/commons/DBReader.cpp:773-838, DBWriter.cpp:415-427
//   data   : entries "payload\0" (sequence DBs: "SEQ\n\0"), possibly split into X.0 .. X.n   DBReader.cpp:108-133
//   dbtype : little-endian int32                              DBWriter.cpp:193-213
// Reading maps the data file (DBReader does the same, DBReader.cpp:108-) and parses the index with all threads; writing takes
// chunks that the module's threads filled for consecutive key ranges - the counterpart of DBWriter's per-thread files and their
// merge at close (DBWriter.cpp:135-188,239-241), without the temporary files.
#pragma once
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

// Large host arrays (index columns, record arrays, output text) come straight from mmap with MADV_HUGEPAGE and without reserve
// accounting: a module touches gigabytes of fresh memory once, and on 4 KB pages the page faults - not the parsing or formatting -
// were its wall time (16 threads formatted alignment text no faster than one).  Capacity is reserved by upper bound and never
// grown, so nothing is copied by a reallocation either; untouched capacity costs address space only.
void *hugeAlloc(size_t bytes);
void hugeFree(void *p, size_t bytes);
template <typename T> struct HugeAlloc {
    typedef T value_type;
    HugeAlloc() = default;
    template <typename U> HugeAlloc(const HugeAlloc<U> &) {}
    T *allocate(size_t n) { return (T *) hugeAlloc(n * sizeof(T)); }
    void deallocate(T *p, size_t n) { hugeFree(p, n * sizeof(T)); }
    // resize() leaves trivially constructible elements as the kernel hands them out (zero pages): no fill pass over fresh memory
    template <typename U, typename... A> void construct(U *p, A &&...a) { if (sizeof...(A) != 0 || !std::is_trivially_default_constructible<U>::value) ::new ((void *) p) U(std::forward<A>(a)...); }
    template <typename U> bool operator==(const HugeAlloc<U> &) const { return true; }
    template <typename U> bool operator!=(const HugeAlloc<U> &) const { return false; }
};
template <typename T> using HVec = std::vector<T, HugeAlloc<T>>;

struct MmDb {
    HVec<uint32_t> key;       // ordered by key (DBReader::sortIndex, DBReader.cpp:238-)
    HVec<uint64_t> off, len;  // len includes the trailing NUL
    HVec<uint8_t> ext;
    int dbtype = 0;
    bool dense = false;       // key[i] == i for every entry (what createdb and the modules write): idOf / keyOf need no look-up
    MmDb() = default;
    MmDb(const MmDb &) = delete;
    MmDb &operator=(const MmDb &) = delete;
    ~MmDb();
    bool load(const std::string &path, std::string *err, bool indexOnly = false);   // indexOnly: DBReader's USE_INDEX - no data file is opened (createhdb.cpp:21-31)
    // the index columns from arrays (a binary side-car, host/sidecar.h): entries in key order, payLen = payload letters (the text DB's
    // length column is payLen + 2); no data is mapped - data() / entry() are not to be called
    void adoptIndex(const uint32_t *keys, const uint32_t *payLen, const uint8_t *extFlags, size_t n, int type);
    size_t size() const { return key.size(); }
    const char *data() const { return base; }
    size_t dataSize() const { return bytes; }
    const char *entry(size_t i) const { return base + off[i]; }
    int64_t idOf(uint32_t k) const {  // -1 when absent
        if (dense) return k < key.size() ? (int64_t) k : -1;
        return idOfSearch(k);
    }
    uint32_t keyOf(size_t i) const { return dense ? (uint32_t) i : key[i]; }
    void noteDense();                // sets `dense` from the keys
private:
    int64_t idOfSearch(uint32_t k) const;
    const char *base = nullptr; size_t bytes = 0;
    void *mapped = nullptr; size_t mappedBytes = 0;   // mmap of a single data file
    HVec<char> owned;                                 // concatenation of split data files
};

// entries of consecutive keys, filled by one thread: data = "payload\0" per entry
struct alignas(128) OutChunk {     // (a cache line of its own: the threads update their chunks' vector ends on every entry)
    HVec<char> data;
    HVec<uint32_t> key, len;   // len includes the NUL
    HVec<uint8_t> ext;
    // upper bounds of what the chunk will hold (entries, payload bytes incl. the NULs): reserved once, never grown by the writers below
    void reserve(size_t entries, size_t bytes) { data.reserve(bytes); key.reserve(entries); len.reserve(entries); ext.reserve(entries); }
    void add(uint32_t k, const char *p, size_t n, uint8_t e) { data.insert(data.end(), p, p + n); data.push_back('\0'); key.push_back(k); len.push_back((uint32_t) n + 1); ext.push_back(e); }
    // in-place writing of one entry: w = open(maxBytes); ... write at w ...; close(key, end, ext) - no staging copy
    char *open(size_t maxBytes) { const size_t at = data.size(); if (data.capacity() - at < maxBytes + 1) data.reserve(std::max(data.capacity() * 2, at + maxBytes + 1)); data.resize(at + maxBytes + 1); return data.data() + at; }
    void close(uint32_t k, char *start, char *end, uint8_t e) { *end++ = '\0'; data.resize((size_t) (end - data.data())); key.push_back(k); len.push_back((uint32_t) (end - start)); ext.push_back(e); }
};
// chunks in increasing key order (chunk i holds smaller keys than chunk i + 1)
// splitData: a large DB may be written as X.0 .. X.n, one data file per chunk (result DBs; never sequence DBs)
bool mmdbWriteChunks(const std::string &path, int dbtype, const std::vector<OutChunk> &chunks, std::string *err, bool splitData = false);
// one blob that already has the data file's layout: entry i at off[i], len[i] bytes incl. the NUL
bool mmdbWriteBlob(const std::string &path, int dbtype, const char *blob, size_t blobBytes, const uint32_t *key, const uint64_t *off,
                   const uint32_t *len, const uint8_t *ext, size_t n, std::string *err, int dataFd = -1);
constexpr int MMDB_DATA_ELSEWHERE = -2;      // mmdbWriteBlob(blob = NULL, dataFd = this): index and dbtype only, the caller writes and closes the data file itself
// a DB's data file filled piece by piece as the pieces come off the device (-1: write it the usual way); mmdbWriteBlob(blob = NULL, dataFd) finishes it
int mmdbOpenStreamedData(const std::string &path, size_t bytes);
bool mmdbWritePiece(int fd, const char *data, uint64_t offset, uint64_t bytes);

// host/ingest.cpp: FASTA/FASTQ[.gz] reads as an in-memory sequence DB (what createdb would write, without the files)
struct FastxDb { HVec<char> blob; HVec<uint32_t> key, len; HVec<uint64_t> off; };     // entry j: blob[off[j] .. off[j] + len[j]) = "SEQ\n\0"
bool readFastxAsDb(const std::vector<std::string> &files, bool shuffle, FastxDb &out, std::string *err);
