// MMseqs2 on-disk DB access for the host binary (format as modified by the CarpeDeam fork).
//   index  : "key \t offset \t length \t wasExtended \n"   lib/mmseqs/src/commons/DBReader.cpp:773-838, DBWriter.cpp:415-427
//   data   : entries "payload\0" (sequence DBs: "SEQ\n\0"), possibly split into X.0 .. X.n   DBReader.cpp:108-133
//   dbtype : little-endian int32                              DBWriter.cpp:193-213
// Reading maps the data file (DBReader does the same, DBReader.cpp:108-) and parses the index with all threads; writing takes
// chunks that the module's threads filled for consecutive key ranges - the counterpart of DBWriter's per-thread files and their
// merge at close (DBWriter.cpp:135-188,239-241), without the temporary files.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct MmDb {
    std::vector<uint32_t> key;       // ordered by key (DBReader::sortIndex, DBReader.cpp:238-)
    std::vector<uint64_t> off, len;  // len includes the trailing NUL
    std::vector<uint8_t> ext;
    int dbtype = 0;
    MmDb() = default;
    MmDb(const MmDb &) = delete;
    MmDb &operator=(const MmDb &) = delete;
    ~MmDb();
    bool load(const std::string &path, std::string *err, bool indexOnly = false);   // indexOnly: DBReader's USE_INDEX - no data file is opened (createhdb.cpp:21-31)
    size_t size() const { return key.size(); }
    const char *data() const { return base; }
    size_t dataSize() const { return bytes; }
    const char *entry(size_t i) const { return base + off[i]; }
    int64_t idOf(uint32_t k) const;  // -1 when absent
private:
    const char *base = nullptr; size_t bytes = 0;
    void *mapped = nullptr; size_t mappedBytes = 0;   // mmap of a single data file
    std::string owned;                                // concatenation of split data files
};

// entries of consecutive keys, filled by one thread: data = "payload\0" per entry
struct OutChunk {
    std::string data;
    std::vector<uint32_t> key, len;   // len includes the NUL
    std::vector<uint8_t> ext;
    void add(uint32_t k, const char *p, size_t n, uint8_t e) { data.append(p, n); data.push_back('\0'); key.push_back(k); len.push_back((uint32_t) n + 1); ext.push_back(e); }
};
// chunks in increasing key order (chunk i holds smaller keys than chunk i + 1)
bool mmdbWriteChunks(const std::string &path, int dbtype, const std::vector<OutChunk> &chunks, std::string *err);
// one blob that already has the data file's layout: entry i at off[i], len[i] bytes incl. the NUL
bool mmdbWriteBlob(const std::string &path, int dbtype, const char *blob, size_t blobBytes, const std::vector<uint32_t> &key, const std::vector<uint64_t> &off,
                   const std::vector<uint32_t> &len, const std::vector<uint8_t> &ext, std::string *err);
