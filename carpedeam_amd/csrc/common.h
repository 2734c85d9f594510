// Internal declarations shared by the C-ABI implementation files (not installed; the public header is
// include/carpedeam_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/carpedeam_hip.h"

void cdm_set_error(const char *fmt, ...);

// Caching device allocator (api.hip): hipMalloc/hipFree of multi-GB buffers cost tens of ms each, and every stage call
// allocates its working set; freed blocks are kept per device and reused for later requests of (nearly) the same size.
// All stage entry points synchronise their stream before they return, so a block is idle when it is handed back.
hipError_t cdmMallocRaw(void **p, size_t bytes);
void cdmFree(void *p);
void cdmPoolTrim();   // give every cached block back to the driver
float cdmPoolHeadroomSwap(float f);      // sets the allocator's head room factor (cdm_pool_headroom), returns the previous one
// value of a CDM_* switch (or OMP_NUM_THREADS) from the library's snapshot of the environment (pool.h: never getenv on a call path)
const char *cdmGetenv(const char *name);
template <typename T> inline hipError_t cdmMalloc(T **p, size_t bytes) { return cdmMallocRaw(reinterpret_cast<void **>(p), bytes); }

// most left-over tuples the reference's last per-target scan may run over on the device (kmermatch.hip k_stale_tail; dist.hip)
constexpr int CDM_STALE_MAX = 62;
// RAII device buffer from the caching allocator (freed on every exit path of a stage function)
// experiments: dynamic LDS (bytes, from the environment) added to a launch to lower its occupancy
inline unsigned cdm_lds_pad(const char *name) { const char *e = cdmGetenv(name); return e ? (unsigned) atoi(e) : 0u; }

template <typename T> struct DevBuf {
    T *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) cdmFree(p); }
    bool alloc(size_t n) { return cdmMalloc(&p, (n + 1) * sizeof(T)) == hipSuccess; }
    T *release() { T *r = p; p = nullptr; return r; }
    void free() { if (p) { cdmFree(p); p = nullptr; } }
};

#define CDM_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            cdm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);   \
            return CDM_ERR_HIP;                                                                         \
        }                                                                                               \
    } while (0)

// A launch takes FEWER THAN 2^32 THREADS in all.  The runtime accepts more without an error and runs (blocks x threads) mod 2^32 of
// them (scripts/probes/big_grid.hip: 18.5 M blocks of 256 ran 441 M threads) - which is how k_contig_stats, a wave per alignment
// record, left 90 % of the statistics of 74 M records unwritten at 25 M reads until round 4.  Kernels that take a wave or a block per
// item are launched in slices (cdmSliceItems: items per launch; the kernel adds its `first` item); kernels with a thread per item sit
// behind size checks that keep the item count below 2^32 - CDM_GRID is the assertion that they do.
constexpr uint64_t CDM_MAX_LAUNCH_THREADS = (1ull << 32) - 1ull;
const char *cdmGetenv(const char *name);
inline uint64_t cdmSliceItems(unsigned threadsPerItem) {        // CDM_LAUNCH_SLICE=<items> (tests): small inputs in several launches
    if (const char *e = cdmGetenv("CDM_LAUNCH_SLICE")) { const long long v = atoll(e); if (v > 0) return (uint64_t) v; }
    return (1ull << 31) / threadsPerItem;
}
inline dim3 cdmGridChecked(uint64_t blocks, unsigned threads, const char *file, int line) {
    if (blocks * (uint64_t) threads > CDM_MAX_LAUNCH_THREADS || blocks == 0) {
        if (blocks == 0) return dim3(1);
        fprintf(stderr, "carpedeam: a launch of %llu blocks x %u threads (%s:%d) is more than the runtime runs (2^32 threads); this is a bug in the size checks in front of it\n", (unsigned long long) blocks, threads, file, line);
        abort();
    }
    return dim3((unsigned) blocks);
}
#define CDM_GRID(blocks, threads) cdmGridChecked((uint64_t) (blocks), (threads), __FILE__, __LINE__)

#define CDM_LAUNCH_CHECK()                                                                              \
    do {                                                                                                \
        hipError_t _e = hipGetLastError();                                                              \
        if (_e != hipSuccess) {                                                                         \
            cdm_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return CDM_ERR_HIP;                                                                         \
        }                                                                                               \
    } while (0)

// ------------------------------------------------------------------------------------------------- damage LUTs
// Host-built tables (host/damage.cpp) in the reference's exact mixed precision; layout used by the kernels.
struct DamageLut {
    // ancient_correction (src/assembler/correction.cpp:48-77,93-107; seqErr = 0.01)
    double logT[4][4];        // [qBase][t]      = (double) logl(seqErr.p[t][qBase])
    double logQ[12][4][4];    // [qcls][qBase][q]: qcls 0..10 = log(max((double)D[qcls].p[q][qBase], 1e-3)); 11 = contig = logT
    double logD[2][11][4][4]; // [rev][l][q][t]  = log(max((double)D(rev)[l].p[q][t], 1e-3))
    // ancient_read_assemble (nuclassembleUtil.cpp:259-276; seqErr = 0.001): log of the per-column likelihood
    double logLik[2][11][4][4];  // [rev][cls][qBase][tBase]
};
int cdm_build_damage(const char *prefix, long double mats[2][11][4][4], DamageLut *lut, std::string *err);

// ------------------------------------------------------------------------------------------------- handles
struct cdm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr, evS0 = nullptr, evS1 = nullptr;
    bool haveDamage = false;
    long double mats[2][11][4][4];
    DamageLut lutHost;
    DamageLut *lutDev = nullptr;
    float lastMs[16] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
    int cuCount = 256;
};

// Sequence DB in HBM.  Base codes A,C,G,T = 0..3 (CarpeDeam's own order, src/assembler/correction.cpp:170-174),
// 16 bases per 32-bit word, little end first; every sequence starts on a word boundary.  'N' (any X-class letter)
// is stored as code 0 plus a bit in nmask (bit index = 16 * woff[i] + pos).
// Letters beyond upper-case ACGTN (lower case, IUPAC codes, anything else a DB may carry): codes/nmask hold what
// NucleotideMatrix::setupLetterMapping (M/commons/NucleotideMatrix.cpp:17-61) maps the letter to - the view of kmermatcher, of the
// diagonal score and of every reverse complement (getNuclRevFragment, nuclassembleUtil.cpp:67-76) - and the sequence gets a row of
// the `raw` plane with its original bytes (index 16 * woff[i] + pos), which is what the forward-strand consumers look at
// (nucleotideMap[c] = 0 for everything but C,G,T; == 'N' tests; letter identity; letters copied to the output).  hasN[i] bit 0 =
// "not plain upper-case ACGT throughout" (all word-wise fast paths test it), bit 1 = the raw row is valid.
struct cdm_seqdb {
    uint64_t n = 0;
    uint64_t words = 0;     // total code words
    uint64_t residues = 0;  // sum of lengths
    uint32_t maxLen = 0;
    uint64_t nCount = 0;    // number of N letters in the whole DB
    uint32_t *woff = nullptr;   // [n+1] word offset of each sequence
    uint32_t *len = nullptr;    // [n]
    uint32_t *key = nullptr;    // [n]
    uint8_t *ext = nullptr;     // [n] wasExtended flag
    uint8_t *hasN = nullptr;    // [n] bit 0: sequence contains an N or any other letter beyond ACGT; bit 1: it has a row in raw
    uint32_t *codes = nullptr;  // [words]
    uint32_t *nmask = nullptr;  // [(words*16+31)/32]
    uint8_t *raw = nullptr;     // [words*16] original bytes of the sequences with hasN bit 1, or NULL (no such sequence in the DB)
    int device = 0;
    uint64_t serial = 0;    // unique per handle (cdm_seqdb_alloc)
};

// Per-sequence metadata gathered by target id in rescore / correction / extension: one 16-byte record instead of four arrays
// (a random target then costs one cache line, not four).  Built per call from the cdm_seqdb arrays (cdm_build_meta, api.hip);
// the proxies keep the kernels' `a.len[t]` spelling.
struct SeqMeta { uint32_t woff, len, flags, key; };      // flags: 1 = has N (not plain ACGT), 2 = wasExtended, 4 = has a raw row
// A PLAIN UNIFORM DB - every sequence of the same length, stored back to back, no flag set on any (no N, no other letter, never
// extended): fresh reads of one sequencing run, the metric's 50 M x 100 bp corpus - needs no record at all: word offset and length
// follow from the index, the flags are zero.  The three stages that gather their targets at random (rescore, correction, extension)
// then fetch ONE line per target, its letters, instead of two (round 5: the metadata line was 64 of the ~130 bytes a target cost,
// profiles/r05_pmc_calibration.json).  uw / ul: words and letters per sequence, 0 = look the record up (cdm_build_meta finds out).
struct MetaUniform { uint32_t words = 0, len = 0; };
struct MetaWoff { const SeqMeta *m; uint32_t uw = 0; __host__ __device__ uint32_t operator[](uint32_t i) const { return uw ? i * uw : m[i].woff; } };
struct MetaLen { const SeqMeta *m; uint32_t ul = 0; __host__ __device__ uint32_t operator[](uint32_t i) const { return ul ? ul : m[i].len; } };
struct MetaHasN { const SeqMeta *m; uint32_t plain = 0; __host__ __device__ uint8_t operator[](uint32_t i) const { return plain ? (uint8_t) 0 : (uint8_t) (m[i].flags & 1u); } };
struct MetaKey { const SeqMeta *m; __host__ __device__ uint32_t operator[](uint32_t i) const { return m[i].key; } };
struct MetaExt { const SeqMeta *m; uint32_t plain = 0; __host__ __device__ uint8_t operator[](uint32_t i) const { return plain ? (uint8_t) 0 : (uint8_t) ((m[i].flags >> 1) & 1u); } };
struct MetaRaw { const SeqMeta *m; uint32_t plain = 0; __host__ __device__ uint8_t operator[](uint32_t i) const { return plain ? (uint8_t) 0 : (uint8_t) ((m[i].flags >> 2) & 1u); } };
// all proxies of a stage's argument block from one table
template <typename A> inline void cdmSetMeta(A &a, const SeqMeta *m, const MetaUniform &u) {
    a.woff.m = a.len.m = a.hasN.m = a.hasRaw.m = m;
    a.woff.uw = u.words; a.len.ul = u.len; a.hasN.plain = a.hasRaw.plain = u.words;
}
int cdm_kmermatch_needs_wide_key(const cdm_seqdb *db);      // kmermatch.hip
int cdm_kmermatch_part_takes_slots(const cdm_seqdb *db, const cdm_kmer_params *par);      // kmermatch.hip
// kmermatcher's first half over ranks by ranges of the k-mer space (kmermatch.hip kmermatchPassesT; csrc/dist.hip fills the hooks from its
// communicator): the whole hit set on every rank
struct KmerRanks {
    int rank, world; void *user;
    int (*gatherHost)(void *user, const void *send, void *recv, uint64_t bytes);                                            // recv[p * bytes ..] = rank p's send
    int (*gatherDev)(void *user, const void *send, uint64_t sendBytes, void *recv, const uint64_t *recvOff, void *stream);  // device buffers, byte offsets [world + 1]
};
int cdm_kmermatch_ranks_impl(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_kmer_params *par, const KmerRanks *ranks, cdm_hits **out);
int cdm_seqdb_overlay(cdm_ctx *ctx, const cdm_seqdb *base, const cdm_seqdb *grown, const uint32_t *idxHost, const uint8_t *extHost, cdm_seqdb **out);      // api.hip
int cdm_build_meta(cdm_ctx *ctx, const cdm_seqdb *db, SeqMeta **out, MetaUniform *uniform = nullptr);      // cdmFree the result; uniform: filled (and the stream synchronised) if asked for

// (the consumers of an alignment set call this first)
#define CDM_REFUSE_UNDEFINED_ALNS(alns, who) do { if ((alns)->undefinedRecords) { cdm_set_error("%s: %llu identity record(s) carry the coordinates -1 (a sequence that scores 0 against itself: more than 40 %% N); the reference indexes the sequence with them here - undefined, not reproduced", who, (unsigned long long) (alns)->undefinedRecords); return CDM_ERR_UNSUPPORTED; } } while (0)
struct HitRec { uint32_t target; int32_t score; int32_t diagonal; };  // == cdm_hit
struct cdm_hits {
    uint64_t n = 0, count = 0;
    uint64_t *off = nullptr;  // [n+1]
    HitRec *rec = nullptr;    // [count]
};

struct AlnRec { uint32_t target; int32_t rawScore; int32_t ident; int32_t qStart, qEnd, dbStart, dbEnd; float seqId; };  // == cdm_aln
struct cdm_alns {
    uint64_t n = 0, count = 0;
    // identity records of sequences that score 0 against themselves on every probed diagonal (more than 40 % N): written with the
    // coordinates -1, as the reference writes them (rescore.hip).  The reference's downstream modules index the sequence with those
    // coordinates (undefined; a segmentation fault in practice) - cdm_correct / cdm_extend / cdm_contig_merge refuse such a set.
    uint64_t undefinedRecords = 0;
    uint64_t *off = nullptr;  // [n+1]
    AlnRec *rec = nullptr;    // [count]
    // by-product of cdm_rescore, which walks the very columns ancient_correction's RY gate looks at: purine/pyrimidine
    // mismatches per record (0xFFFF = not known, e.g. a sequence with N); only valid for the sequence DB it was computed on
    uint16_t *ryMism = nullptr; uint64_t rySerial = 0;
};

int cdm_seqdb_alloc_like(cdm_ctx *ctx, const cdm_seqdb *src, cdm_seqdb **out);  // same n/lengths/layout, codes uninitialised
int cdm_seqdb_alloc(cdm_ctx *ctx, uint64_t n, cdm_seqdb **out);
int cdm_seqdb_alloc_raw(cdm_seqdb *db);      // the raw plane for db->words code words (contents undefined)
// sub-DB: sel[i] (device) = 0xFFFFFFFF drops sequence i, else keeps its first sel[i] letters; extValue < 0 keeps the wasExtended flags
int cdm_seqdb_select(cdm_ctx *ctx, const cdm_seqdb *db, const uint32_t *sel, int extValue, cdm_seqdb **out);

// Host memory for what comes down from the device in bulk: anonymous pages, not initialised (a std::vector or std::string of that
// size is written once by its constructor before the copy writes it again), huge pages where the system hands them out on advice.
#include <sys/mman.h>
template <typename T> struct HostBuf {
    T *p = nullptr; size_t n = 0, bytes = 0;
    HostBuf() = default;
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    ~HostBuf() { release(); }
    void release() { if (p) munmap(p, bytes); p = nullptr; n = bytes = 0; }
    bool alloc(size_t count) {
        release();
        bytes = ((count * sizeof(T) + 1) + (2u << 20) - 1) & ~(size_t) ((2u << 20) - 1);
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) { bytes = 0; return false; }
        madvise(m, bytes, MADV_HUGEPAGE);
        p = static_cast<T *>(m); n = count;
        return true;
    }
    T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) const { return p[i]; }
};
// a sequence of the downloaded DB blob as the host part of ancient_contig_merge sees it (no copy, no allocation per sequence)
struct SeqView {
    const char *p = nullptr; size_t n = 0;
    size_t size() const { return n; }
    const char *data() const { return p; }
    char operator[](size_t i) const { return p[i]; }
    std::string substr(size_t a, size_t l) const { return std::string(p + a, l); }
};
// ancient_contig_merge: what the device counts per alignment record (contig.hip) for the host part (host/contigmerge.cpp)
struct ContigStat {            // per alignment record, oriented as :193-214 does
    int32_t qs, qe, ds, de;
    int32_t rev;
    int32_t idCnt, idRy;       // identical / same-RY-class letters over [qs, qe] (:217-223; N == N counts)
    int32_t nnTot, nnId, nnRy; // the same over the columns where neither letter is N (what the consensus loops see, safe mode)
    int32_t nCT, nGA;          // query C over target T, query G over target A among those (ancientMatchCount's dimers)
    uint32_t dbLen, dbKey;     // the target's length and key (the host's candidate gate reads nothing else of the target: no look-up per record there)
};

// stage implementations (one .hip file each)
int cdm_correct_impl(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, cdm_seqdb *out);
int cdm_rescore_impl(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_hits *hits, const cdm_rescore_params *par, cdm_alns **out);
int cdm_rescore_hamming_impl(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_hits *hits, const cdm_hamming_params *par, cdm_hits **out);
int cdm_kmermatch_impl(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_kmer_params *par, cdm_hits **out);
int cdm_extend_impl(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, cdm_seqdb **out, double *scores);
int cdm_synth_impl(cdm_ctx *ctx, uint64_t nTotal, uint64_t first, uint64_t n, uint32_t lo, uint32_t hi, uint64_t seed, cdm_seqdb **out);

// host E-value helpers (host/evalue.cpp)
double cdm_evalue_host(double rawScore, double qLen, uint64_t dbResidues);
int cdm_bit_score_host(double rawScore);
// the same for gapped alignments of nucleotide.out with the gap costs `ancient_assemble` uses (5, 2): host/evalue.cpp
bool cdm_gapped_costs_known(int gapOpen, int gapExtend);
double cdm_evalue_gapped_host(double rawScore, double qLen, uint64_t dbResidues);
int cdm_bit_score_gapped_host(double rawScore);
// smallest raw score whose E-value is <= thr for a query of that length (monotone in the score); INT_MAX if none up to 2*maxLen
void cdm_min_score_table(double evalThr, uint64_t dbResidues, uint32_t maxLen, std::vector<int32_t> &table);
