// Binary side-cars: what an owned module leaves next to a DB it wrote (or first read), for the next owned module of the workflow.
//
// The reference's modules hand everything over as TEXT DBs (prefilter hits "%u\t%d\t%d\n", QueryMatcher.h:114-126; alignments,
// Matcher.cpp:356-404, parsed back by Matcher.cpp:274-353; sequences "SEQ\n\0", DBWriter.cpp:322-427), one process per stage
// (data/nuclassemble.sh:105-136).  Those files stay what they are - byte for byte, every reference module reads them.  Beside a DB
// X an owned module also writes X.cdmbin: the same content as the consumer's parser would produce it - CSR offsets + records as
// cdm_hits_upload / cdm_alns_upload take them (packed to 8 / 16 bytes where the fields fit), sequences 2 bits per base with their
// index columns - stamped with the sizes and modification times of X's files.  A consumer that finds a side-car whose stamp matches
// the files (and, for records, whose sequence DB is the one it loaded) takes it and never maps, indexes or parses the text; anything
// else - no side-car, another stamp, CDM_SIDECAR=0 - is the text path as before.  rmdb / mvdb treat it as one of the DB's files.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "carpedeam_hip.h"
#include "mmdb.h"

enum { SIDE_SEQ = 1, SIDE_HITS = 2, SIDE_ALNS = 3 };
enum { SIDE_F_COMPACT = 1, SIDE_F_HAS_NMASK = 2, SIDE_F_HAS_RAW = 4 };
struct SideStamp { uint64_t dataBytes = 0, dataMtimeNs = 0, indexBytes = 0, indexMtimeNs = 0; };
constexpr int SIDE_SECTIONS = 8;
struct SideHeader {
    char magic[8];              // "CDMSIDE1"
    uint32_t kind, flags;
    uint64_t n;                 // sequences / queries
    uint64_t count;             // records (hits, alignments) or code words (sequences)
    uint64_t seqN, seqKeyHash;  // records: the sequence DB their indices refer to
    int32_t dbtype; uint32_t pad;
    SideStamp stamp;            // of the text DB's files when the side-car was written
    uint64_t section[SIDE_SECTIONS];        // bytes of each section; they follow the header, each starting on a 64-byte boundary
};
struct SidePiece { const void *p; uint64_t bytes; };
inline std::string sidePath(const std::string &db) { return db + ".cdmbin"; }
bool sideEnabled();
bool sideStampOf(const std::string &db, SideStamp *st);
uint64_t sideKeyHash(const uint32_t *keys, size_t n);
// writes X.cdmbin for the DB X whose text files are complete (their stamp is taken here); false: could not (the text DB stands on its own)
bool sideWrite(const std::string &db, uint32_t kind, uint32_t flags, uint64_t n, uint64_t count, uint64_t seqN, uint64_t seqKeyHash, int dbtype, const SidePiece *pieces, int nPieces);
// the same in two steps (the sections while X is still being written, the stamped header and the rename once X is complete)
bool sideWriteBody(const std::string &db, uint32_t kind, uint32_t flags, uint64_t n, uint64_t count, uint64_t seqN, uint64_t seqKeyHash, int dbtype, const SidePiece *pieces, int nPieces, SideHeader *h);
bool sideCommit(const std::string &db, SideHeader *h);
// a mapped side-car whose stamp matches X's files
struct SideFile {
    const SideHeader *h = nullptr; const char *base = nullptr; size_t bytes = 0;
    const void *section(int i) const;
    ~SideFile();
    SideFile() = default;
    SideFile(const SideFile &) = delete;
    SideFile &operator=(const SideFile &) = delete;
};
bool sideOpen(const std::string &db, uint32_t kind, SideFile &f);

// compact records (SIDE_F_COMPACT): what fits them is what the text of a read DB holds
struct SideHit8 { uint32_t target; int16_t score, diagonal; };
struct SideAln16 { uint32_t target; uint16_t rawScore, seqId1000; int16_t qStart, qEnd, dbStart, dbEnd; };
inline bool fitsHit8(const cdm_hit &h) { return h.score >= -32768 && h.score <= 32767 && h.diagonal >= -32768 && h.diagonal <= 32767; }
// the float a reader of the alignment text gets for the sequence identity: the text is the identity truncated to three decimals
// (Util::fastSeqIdToBuffer, Util.cpp:278-307; "1.00" for 1), read back as decimal / 1000 (Matcher.cpp:274-353: strtod, then float)
inline float seqIdFrom1000(uint32_t v) { return (float) ((double) v / 1000.0); }
inline uint32_t seqIdTo1000(float s) { return s == 1.0f ? 1000u : (uint32_t) (int) (s * 1000); }
