// MMseqs2 on-disk DB access for the host binary (format as modified by the CarpeDeam fork).
//   index  : "key \t offset \t length \t wasExtended \n"   lib/mmseqs/src/commons/DBReader.cpp:773-838, DBWriter.cpp:415-427
//   data   : entries "payload\0" (sequence DBs: "SEQ\n\0"), possibly split into X.0 .. X.n   DBReader.cpp:108-133
//   dbtype : little-endian int32                              DBWriter.cpp:193-213
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct MmDb {
    std::vector<uint32_t> key;       // ordered by key (DBReader::sortIndex, DBReader.cpp:238-)
    std::vector<uint64_t> off, len;  // len includes the trailing NUL
    std::vector<uint8_t> ext;
    std::string data;
    int dbtype = 0;
    bool load(const std::string &path, std::string *err);
    size_t size() const { return key.size(); }
    const char *entry(size_t i) const { return data.data() + off[i]; }
    int64_t idOf(uint32_t k) const;  // -1 when absent
};

struct MmDbWriter {
    std::string path; int dbtype;
    std::vector<uint32_t> key; std::vector<std::string> payload; std::vector<uint8_t> ext;
    MmDbWriter(const std::string &p, int t) : path(p), dbtype(t) {}
    void add(uint32_t k, std::string &&p, uint8_t e) { key.push_back(k); payload.push_back(std::move(p)); ext.push_back(e); }
    bool close(std::string *err);
};
