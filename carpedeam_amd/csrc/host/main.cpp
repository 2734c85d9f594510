// carpedeam_mi355x: the module surface of the reference's hot path on the device (reached through the front end `carpedeam`,
// host/front.c, which also forwards every other command to the reference's binary when a deployment has one).
//
//   carpedeam kmermatcher           <seqDB> <prefDB> [flags]                  lib/mmseqs/src/linclust/kmermatcher.cpp:786
//   carpedeam rescorediagonal       <qDB> <tDB> <prefDB> <alnDB> [flags]      lib/mmseqs/src/alignment/rescorediagonal.cpp:381
//   carpedeam ancient_correction    <seqDB> <alnDB> <outDB> [flags]           src/assembler/correction.cpp:492
//   carpedeam ancient_read_assemble <seqDB> <alnDB> <outDB> [flags]           src/assembler/ancientReadsResults.cpp:598
//
// Same positional arguments, flag names, on-disk DB formats and exit codes as the reference modules, so the workflow script
// (data/nuclassemble.sh:105,115,126,136) can call this binary for these four stages.  Each module reads its DBs, hands the
// work to the gfx950 library through the C ABI (include/carpedeam_hip.h) and writes the result DB with the reference's text
// codecs.  There is no CPU path: without an MI355X the module exits with an error, like any other fatal error in the
// reference ("Debug(Debug::ERROR) << ...; EXIT(EXIT_FAILURE)").
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <omp.h>
#include <unistd.h>
#include <sys/stat.h>
#include <string>
#include <thread>
#include <vector>
#include <condition_variable>
#include <mutex>

#include "carpedeam_hip.h"
#include <sys/prctl.h>
#include <sys/wait.h>
#include <signal.h>
#include <cerrno>
#include "mmdb.h"
#include "sidecar.h"

// host/ingest.cpp
int createdbModule(const std::vector<std::string> &files, const std::string &outPath, bool shuffle, int dbType, std::string *err);
int convert2fastaModule(const std::string &dbPath, const std::string &outPath, std::string *err);
int createhdbModule(const std::string &seqPath, const std::string &cyclePath, const std::string &outPath, std::string *err);
// host/cluster.cpp: the host-side modules of linclust's tail and the workflow scripts' file modules (status 77: a mode that is not implemented)
int clustModule(const std::string &seqPath, const std::string &alnPath, const std::string &outPath, int mode, std::string *err);
int createsubdbModule(const std::string &orderArg, const std::string &inPath, const std::string &outPath, int subDbMode, std::string *err);
int filterdbModule(const std::string &inPath, const std::string &outPath, const std::string &filterFile, std::string *err);
int mergeclustersModule(const std::string &seqPath, const std::string &outPath, const std::vector<std::string> &steps, std::string *err);
int result2repseqModule(const std::string &seqPath, const std::string &cluPath, const std::string &outPath, std::string *err);
int rmdbModule(const std::string &db);
// host/align.cpp: linclust's gapped alignment step on the assembled contigs
struct AlignParams {
    float covThr = 0.f, seqIdThr = 0.f; double evalThr = 0.001; int covMode = 0, seqIdMode = 0, alnLenThr = 0; bool wrapped = false, includeIdentity = false;
    int gapOpen = 5, gapExtend = 2, zdrop = 40; unsigned maxAccept = INT_MAX, maxReject = INT_MAX; size_t maxSeqLen = 65535;
};
int alignModule(const std::string &qPath, const std::string &tPath, const std::string &prefPath, const std::string &outPath, const AlignParams &P, std::string *err);
int mvdbModule(const std::string &src, const std::string &dst, std::string *err);

namespace {
std::thread *g_deviceThread = NULL;          // a module's device start-up running beside the main thread (DeviceStart): joined before any exit
void quiesce() { if (g_deviceThread && g_deviceThread->joinable() && g_deviceThread->get_id() != std::this_thread::get_id()) g_deviceThread->join(); }
[[noreturn]] void die(const std::string &msg) { quiesce(); fprintf(stderr, "%s\n", msg.c_str()); exit(EXIT_FAILURE); }
// a call the device path does not implement, found before any work was done: status 77 tells the front end (host/front.c) that the
// reference binary - if the deployment has one - may take this call instead
[[noreturn]] void unsupported(const std::string &msg) { quiesce(); fprintf(stderr, "%s\n", msg.c_str()); exit(77); }
void check(int rc, const char *what) { if (rc != CDM_OK) die(std::string(what) + ": " + cdm_last_error()); }

struct Args { std::vector<std::string> pos; std::map<std::string, std::string> flag; };
Args parse(int argc, char **argv) {
    Args a;
    for (int i = 0; i < argc; i++) {
        std::string s = argv[i];
        if (s.size() > 1 && s[0] == '-' && !isdigit((unsigned char) s[1])) { if (i + 1 < argc) { a.flag[s] = argv[i + 1]; i++; } }
        else a.pos.push_back(s);
    }
    return a;
}
// Per-module flag tables (the reference's own lists: lib/mmseqs/src/commons/Parameters.cpp:871-892 kmermatcher, :422-439
// rescorediagonal, src/commons/LocalParameters.h:155-171 assembleresults = ancient_correction / ancient_read_assemble).
//   'U' used by the MI355X path;  'N' accepted, no effect on this path in the reference either (or only on resources);
//   'V' accepted only with one of the listed values (anything else would be computed differently: refused).
// A flag that is not in the module's list is an error, as in Parameters::parseParameters (:1703 "Unrecognized parameter").
struct FlagSpec { const char *name; char kind; const char *allowed; const char *why; };
const FlagSpec KMERMATCHER_FLAGS[] = {
    {"--kmer-per-seq", 'U', 0, 0}, {"--kmer-per-seq-scale", 'U', 0, 0}, {"--cov-mode", 'U', 0, 0}, {"-k", 'U', 0, 0}, {"-c", 'U', 0, 0}, {"--hash-shift", 'U', 0, 0},
    {"--include-only-extendable", 'U', 0, 0}, {"--ignore-multi-kmer", 'U', 0, 0},
    {"--alph-size", 'N', 0, "nucleotide k-mers are never reduced (kmermatcher.cpp:604)"}, {"--min-seq-id", 'N', 0, "not read by kmermatcher"},
    {"--max-seq-len", 'N', 0, "buffer sizing"}, {"--split-memory-limit", 'N', 0, "the device path is single-split"}, {"--threads", 'N', 0, 0}, {"-v", 'N', 0, 0},
    {"--sub-mat", 'V', "*nucleotide.out*", "only the nucleotide matrix"}, {"--mask", 'V', "0", "tantan masking is not implemented"},
    {"--mask-lower-case", 'V', "0", "lower-case masking is not implemented"}, {"--spaced-kmer-mode", 'V', "0", "spaced k-mers are not implemented"},
    {"--spaced-kmer-pattern", 'V', "", "spaced k-mers are not implemented"}, {"--adjust-kmer-len", 'V', "0", "k-mer length adjustment is not implemented"},
    {"--compressed", 'V', "0", "compressed DBs are not implemented"}, {0, 0, 0, 0}};
const FlagSpec RESCORE_FLAGS[] = {
    {"-e", 'U', 0, 0}, {"-c", 'U', 0, 0}, {"--cov-mode", 'U', 0, 0}, {"--min-seq-id", 'U', 0, 0}, {"--min-aln-len", 'U', 0, 0},
    {"--add-self-matches", 'N', 0, "query DB == target DB: self matches are kept anyway (rescorediagonal.cpp:205)"}, {"--db-load-mode", 'N', 0, 0}, {"--threads", 'N', 0, 0}, {"-v", 'N', 0, 0},
    {"--seq-id-mode", 'V', "0", "only alignment-length normalisation"}, {"--rescore-mode", 'V', "3", "only the end-to-end ungapped mode CarpeDeam uses"},
    {"--wrapped-scoring", 'V', "0", "not implemented"}, {"--filter-hits", 'V', "0", "not implemented"}, {"-a", 'V', "0", "no backtrace in mode 3"},
    {"--sort-results", 'V', "0", "not implemented"}, {"--sub-mat", 'V', "*nucleotide.out*", "only the nucleotide matrix"}, {"--compressed", 'V', "0", "compressed DBs are not implemented"}, {0, 0, 0, 0}};
// the module in linclust's pre-clustering mode (linclust.sh:27-31): --rescore-mode 0 --wrapped-scoring 1
const FlagSpec RESCORE_HAMMING_FLAGS[] = {
    {"-e", 'U', 0, 0}, {"-c", 'U', 0, 0}, {"--cov-mode", 'U', 0, 0}, {"--min-seq-id", 'U', 0, 0}, {"--min-aln-len", 'U', 0, 0}, {"--seq-id-mode", 'U', 0, 0},
    {"--add-self-matches", 'N', 0, "query DB == target DB: self matches are kept anyway (rescorediagonal.cpp:205)"}, {"--db-load-mode", 'N', 0, 0}, {"--threads", 'N', 0, 0}, {"-v", 'N', 0, 0},
    {"--rescore-mode", 'V', "0", "this table is the Hamming mode's"}, {"--wrapped-scoring", 'V', "1", "the Hamming mode is implemented with wrapped scoring only"},
    {"--filter-hits", 'V', "0", "not implemented"}, {"-a", 'V', "0", "no backtrace in this mode"},
    {"--sort-results", 'V', "0", "not implemented"}, {"--sub-mat", 'V', "*nucleotide.out*", "only the nucleotide matrix"}, {"--compressed", 'V', "0", "compressed DBs are not implemented"}, {0, 0, 0, 0}};
const FlagSpec ANCIENT_FLAGS[] = {
    {"--min-seq-id", 'U', 0, 0}, {"--max-seq-len", 'U', 0, 0}, {"--ext-random-align", 'U', 0, 0}, {"--excess-penalty", 'U', 0, 0}, {"--min-ryseq-id-corr-reads", 'U', 0, 0},
    {"--likelihood-ratio-threshold", 'U', 0, 0}, {"--ancient-damage", 'U', 0, 0}, {"--unsafe", 'U', 0, 0}, {"--min-cov-safe", 'U', 0, 0},
    {"--keep-target", 'N', 0, "not read by these modules"}, {"--min-seqid-corr-reads", 'N', 0, "not read by these modules"}, {"--min-merge-seq-id", 'U', 0, 0},
    {"--min-seqid-corr-contigs", 'N', 0, "a workflow flag: it becomes the contig phase's --min-seq-id (Nuclassembler.cpp:124-126); ancient_reads_loop reads it"}, {"--threads", 'N', 0, 0}, {"-v", 'N', 0, 0},
    {"--rescore-mode", 'V', "3", "re-alignment of parked candidates is end-to-end ungapped (ancientReadsResults.cpp:502)"}, {0, 0, 0, 0}};
void checkFlags(const char *module, const Args &a, const FlagSpec *spec, const char *const *extra = NULL) {
    for (const auto &kv : a.flag) {
        const FlagSpec *f = spec;
        while (f->name && kv.first != f->name) f++;
        bool known = f->name != NULL;
        for (const char *const *e = extra; !known && e && *e; e++) known = kv.first == *e;
        if (!known) die("Unrecognized parameter \"" + kv.first + "\"");
        if (f->name && f->kind == 'V') {
            const std::string al = f->allowed;
            const bool ok = (al.size() >= 2 && al[0] == '*') ? kv.second.find(al.substr(1, al.size() - 2)) != std::string::npos : kv.second == al;
            if (!ok) unsupported(std::string(module) + ": " + kv.first + " " + kv.second + " is not supported by the MI355X path (" + f->why + "; accepted: " + (al.empty() ? "\"\"" : al) + ")");
        }
    }
}
float fflag(Args &a, const char *n, float d) { return a.flag.count(n) ? strtof(a.flag[n].c_str(), NULL) : d; }
long iflag(Args &a, const char *n, long d) { return a.flag.count(n) ? strtol(a.flag[n].c_str(), NULL, 10) : d; }

// CDM_TIMING=1: where a module's wall time goes (stderr)
std::chrono::steady_clock::time_point g_t0, g_lastLap;
struct Laps {
    bool on; std::chrono::steady_clock::time_point t;
    Laps() : on(getenv("CDM_TIMING") != NULL), t(std::chrono::steady_clock::now()) {
        if (on && g_t0.time_since_epoch().count()) fprintf(stderr, "  %-32s %.3f s\n", "(arguments, flags)", std::chrono::duration<double>(t - g_t0).count());
        g_lastLap = t;
    }
    void lap(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "  %-32s %.3f s\n", what, std::chrono::duration<double>(n - t).count()); t = n; g_lastLap = n;
    }
};
// The end of a module whose outputs are written and closed: the process leaves HERE - no release of device buffers (giving tens of GB
// back to the driver costs seconds, profiles/r05_time_modules_50M_*), no unmapping of the input files, no destructors of the
// multi-GB host buffers; the operating system and the driver take everything back at once.  (Not under a profiler or sanitizer that
// writes its results from an exit handler - ROCP_TOOL_LIBRARIES / LD_PRELOAD / CDM_NORMAL_EXIT: then the caller releases and returns.)
bool leaveAtOnce() {
    if (getenv("ROCP_TOOL_LIBRARIES") || getenv("CDM_NORMAL_EXIT")) return false;
    const char *pre = getenv("LD_PRELOAD");         // (a profiler's preload counts, any other preloaded library - a deployment's exec guard, an allocator - does not)
    return !(pre && (strstr(pre, "rocprof") || strstr(pre, "roctracer") || strstr(pre, "rocprofiler")));
}
// The caller gets its answer before this process is torn down: the work runs in a CHILD of the process the caller started (forked first
// thing in main, before anything touches the device), the parent waits for one byte - the exit status, sent when every output file is
// written and closed - and leaves with it at once.  Tearing down a module's address space (tens of GB of host buffers and mapped DB
// files at 50 M reads: 0.4-1.3 s, profiles/r05_probe_module_gap.txt) then happens beside the next module of the workflow instead of
// in front of it.  A child that ends without the byte (a crash, an error exit) is waited for and its status handed on.
// Not under a profiler (leaveAtOnce()) and not with CDM_NO_FORK=1.
int g_doneFd = -1;
void reportDone(int rc) {
    if (g_doneFd < 0) return;
    prctl(PR_SET_PDEATHSIG, 0);         // (from here on the parent's leaving is the plan, not an accident)
    const unsigned char b = (unsigned char) rc;
    if (write(g_doneFd, &b, 1) != 1) {}
    close(g_doneFd); g_doneFd = -1;
    close(0); close(1); close(2);       // (a caller that reads this process's output through pipes waits for their last writer)
}
void finishModule(int rc) {
    if (!leaveAtOnce()) return;
    const auto n = std::chrono::steady_clock::now();
    if (getenv("CDM_TIMING")) fprintf(stderr, "  %-32s %.3f s\n", "(since the last lap)", std::chrono::duration<double>(n - g_lastLap).count());
    fprintf(stderr, "Time for processing: %.3fs\n", std::chrono::duration<double>(n - g_t0).count());
    fflush(stdout); fflush(stderr);
    reportDone(rc);
    _exit(rc);
}
cdm_ctx *openCtx(int offset = 0) {
    cdm_ctx *ctx = NULL;
    const char *dev = getenv("CARPEDEAM_DEVICE");
    check(cdm_ctx_create((dev ? atoi(dev) : 0) + offset, &ctx), "Can not initialise the MI355X device");
    return ctx;
}
// ---- ancient_reads_loop --gpus N: the ranks are host threads of this process, one device each, and the library splits the read
// iterations over them (cdm_reads_iteration_dist, csrc/dist.hip) - over RCCL.  CDM_LOOP_TRANSPORT=threads (tests on a one-GPU box,
// where RCCL has no second device to give a rank): all ranks share the first device and the collectives are these few lines - the
// ranks publish their buffers, meet at a barrier and copy from each other with cdm_dev_copy.
struct RankBarrier {
    std::mutex m; std::condition_variable cv; int world = 1, waiting = 0; long generation = 0;
    void wait() {
        std::unique_lock<std::mutex> l(m);
        const long g = generation;
        if (++waiting == world) { waiting = 0; generation++; cv.notify_all(); }
        else cv.wait(l, [&] { return generation != g; });
    }
};
struct ThreadTransport {
    int world = 1; RankBarrier bar;
    std::vector<const void *> ptr; std::vector<const uint64_t *> off; std::vector<uint64_t> bytes;
    explicit ThreadTransport(int w) : world(w), ptr(w), off(w), bytes(w) { bar.world = w; }
};
struct ThreadRank { ThreadTransport *t; int rank; cdm_ctx *ctx; };
int ttAllGatherHost(void *user, const void *send, void *recv, uint64_t n) {
    ThreadRank *r = (ThreadRank *) user; ThreadTransport &t = *r->t;
    t.ptr[r->rank] = send; t.bar.wait();
    for (int p = 0; p < t.world; p++) memcpy((char *) recv + (size_t) p * n, t.ptr[p], n);
    t.bar.wait();
    return 0;
}
int ttAllToAllDev(void *user, const void *send, const uint64_t *soff, void *recv, const uint64_t *roff, void *) {
    ThreadRank *r = (ThreadRank *) user; ThreadTransport &t = *r->t;
    if (cdm_ctx_sync(r->ctx)) return -1;                 // what this rank sends is complete
    t.ptr[r->rank] = send; t.off[r->rank] = soff; t.bar.wait();
    int rc = 0;
    for (int p = 0; p < t.world && !rc; p++) {
        const uint64_t n = t.off[p][r->rank + 1] - t.off[p][r->rank];
        if (n != roff[p + 1] - roff[p]) rc = -1;
        else if (n) rc = cdm_dev_copy(r->ctx, (char *) recv + roff[p], (const char *) t.ptr[p] + t.off[p][r->rank], n);
    }
    t.bar.wait();                                        // nobody frees its send buffer before everybody has copied
    return rc;
}
int ttAllGatherDev(void *user, const void *send, uint64_t n, void *recv, const uint64_t *roff, void *) {
    ThreadRank *r = (ThreadRank *) user; ThreadTransport &t = *r->t;
    if (cdm_ctx_sync(r->ctx)) return -1;
    t.ptr[r->rank] = send; t.bytes[r->rank] = n; t.bar.wait();
    int rc = 0;
    for (int p = 0; p < t.world && !rc; p++) {
        if (t.bytes[p] != roff[p + 1] - roff[p]) rc = -1;
        else if (t.bytes[p]) rc = cdm_dev_copy(r->ctx, (char *) recv + roff[p], t.ptr[p], t.bytes[p]);
    }
    t.bar.wait();
    return rc;
}
// The device side of a module's start - runtime initialisation and context (0.1-0.2 s), damage tables, sequence upload - in a thread
// of its own, while the main thread maps and parses the module's text DBs.  Errors are kept and raised by join() on the main thread.
// ---- binary side-cars (host/sidecar.h)
// sections of a sequence side-car: keys, lengths, wasExtended flags, letter flags (hasN), code words, [N masks], [raw plane]
bool importSeqSide(cdm_ctx *ctx, const SideFile &f, cdm_seqdb **out) {
    const SideHeader &h = *f.h;
    const bool hasMask = (h.flags & SIDE_F_HAS_NMASK) != 0, hasRaw = (h.flags & SIDE_F_HAS_RAW) != 0;
    return cdm_seqdb_import_packed(ctx, f.section(4), hasMask ? f.section(5) : NULL, f.section(1), f.section(0), f.section(2), hasRaw ? f.section(6) : NULL,
                                   hasRaw ? f.section(3) : NULL, h.n, h.count, out) == CDM_OK;
}
// the device DB `h` as a side-car, in two steps: take() brings the packed form to the host while the device still holds it, write() puts
// it next to the text DB at `path` once that DB's files are complete; a failure of either leaves the text DB on its own
struct SeqSideHost {
    HVec<uint32_t> keys, lens, codes; HVec<uint8_t> ext, flags, raw; HVec<uint16_t> mask;
    uint64_t n = 0, words = 0; bool have = false, hasRaw = false;
    void take(cdm_ctx *ctx, cdm_seqdb *h) {
        have = false;
        if (!sideEnabled()) return;
        n = cdm_seqdb_size(h); words = cdm_seqdb_words(h);
        if (n == 0) return;
        keys.resize(n); lens.resize(n); codes.resize(words + 1); ext.resize(n); flags.resize(n); mask.resize(words + 1);
        hasRaw = cdm_seqdb_has_raw(h) != 0;
        if (hasRaw) raw.resize(words * 16 + 1);
        have = cdm_seqdb_export_packed(ctx, h, codes.data(), mask.data(), lens.data(), keys.data(), ext.data(), hasRaw ? raw.data() : NULL, flags.data()) == CDM_OK;
    }
    // begin(): the sections go out in a thread of their own (the caller writes the text DB meanwhile); commit(): the text DB is complete
    std::thread th; SideHeader hdr; bool bodyOk = false; std::string path;
    void begin(const std::string &p, int dbtype) {
        if (!have) return;
        path = p;
        th = std::thread([this, dbtype] {
            bool anyN = false;
            for (size_t i = 0; i < n && !anyN; i++) anyN = flags[i] != 0;
            const SidePiece pc[7] = {{keys.data(), n * 4}, {lens.data(), n * 4}, {ext.data(), n}, {flags.data(), n}, {codes.data(), words * 4}, {mask.data(), anyN ? words * 2 : 0}, {raw.data(), hasRaw ? words * 16 : 0}};
            bodyOk = sideWriteBody(path, SIDE_SEQ, (anyN ? SIDE_F_HAS_NMASK : 0) | (hasRaw ? SIDE_F_HAS_RAW : 0), n, words, 0, 0, dbtype, pc, 7, &hdr);
        });
    }
    void commit() { if (th.joinable()) { th.join(); if (bodyOk) sideCommit(path, &hdr); } }
    void write(const std::string &p, int dbtype) { begin(p, dbtype); commit(); }
};
// What a module still holds on the device goes back to the driver as soon as its last result is on the host, BEFORE the text is formatted
// and written: memory a process gives back is cleared by the driver before another process gets it (~30 ms per GB, in the background:
// profiles/r05_probe_exit.txt), and the next module of the workflow starts milliseconds after this one ends - it found that clearing in
// its way (0.35 -> 2.5 s for rescorediagonal's kernels behind kmermatcher).  Giving back is quick (10 ms for 140 GB); it is this thread's
// arenas that hold the memory (the kernels ran here), so this thread does it.
struct DeviceEnd {
    void begin(cdm_ctx *ctx, cdm_hits *hits, cdm_alns *alns, cdm_seqdb *a, cdm_seqdb *b) {
        if (hits) cdm_hits_free(hits);
        if (alns) cdm_alns_free(alns);
        if (a) cdm_seqdb_free(a);
        if (b) cdm_seqdb_free(b);
        cdm_ctx_destroy(ctx);
    }
    void join() {}
};
// A sequence DB as a module takes it: from its side-car when that matches the files - index columns from the side-car's arrays, no text
// mapped - else from the text (MmDb::load).
struct SeqInput {
    MmDb db; SideFile side; bool fromSide = false;
    void load(const std::string &path) {
        if (sideOpen(path, SIDE_SEQ, side)) {
            const SideHeader &h = *side.h;
            db.adoptIndex((const uint32_t *) side.section(0), (const uint32_t *) side.section(1), (const uint8_t *) side.section(2), h.n, h.dbtype);
            fromSide = true;
            if (getenv("CDM_TIMING")) fprintf(stderr, "  sequence DB %s: from its side-car\n", path.c_str());
            return;
        }
        std::string err; if (!db.load(path, &err)) die(err);
        if (getenv("CDM_TIMING")) fprintf(stderr, "  sequence DB %s: from the text\n", path.c_str());
    }
};
struct DeviceStart {
    std::thread th; cdm_ctx *ctx = NULL; cdm_seqdb *db = NULL;
    std::string err; int code = 0;
    void fail(int c, const std::string &m) { code = c; err = m; }
    void begin(const SeqInput *in, const std::string *damagePrefix) { begin(in ? &in->db : NULL, damagePrefix, in && in->fromSide ? &in->side : NULL); }
    void begin(const MmDb *seq, const std::string *damagePrefix, const SideFile *side = NULL) {
        th = std::thread([this, seq, damagePrefix, side] {
            const char *dev = getenv("CARPEDEAM_DEVICE");
            if (cdm_ctx_create(dev ? atoi(dev) : 0, &ctx) != CDM_OK) return fail(EXIT_FAILURE, std::string("Can not initialise the MI355X device: ") + cdm_last_error());
            if (damagePrefix && cdm_damage_load(ctx, damagePrefix->c_str()) != CDM_OK) return fail(EXIT_FAILURE, std::string("Profile not 12 fields: ") + cdm_last_error());
            if (!seq) return;
            if ((seq->dbtype & 0x7FFFFFFF) != 1) return fail(77, "The MI355X path works on nucleotide sequence DBs only (dbtype " + std::to_string(seq->dbtype & 0x7FFFFFFF) + " given)");
            if (side) { if (!importSeqSide(ctx, *side, &db)) fail(EXIT_FAILURE, std::string("Can not load the sequence DB: ") + cdm_last_error()); return; }
            std::vector<uint32_t> lens(seq->size());
            for (size_t i = 0; i < seq->size(); i++) lens[i] = seq->len[i] >= 2 ? (uint32_t) (seq->len[i] - 2) : 0;   // DBReader::getSeqLen
            if (cdm_seqdb_upload(ctx, seq->data(), seq->off.data(), lens.data(), seq->key.data(), seq->ext.data(), seq->size(), &db) != CDM_OK)
                return fail(EXIT_FAILURE, std::string("Can not load the sequence DB: ") + cdm_last_error());
        });
        g_deviceThread = &th;
    }
    void join() {
        if (th.joinable()) th.join();
        g_deviceThread = NULL;
        if (code) { fprintf(stderr, "%s\n", err.c_str()); exit(code); }
    }
};
cdm_seqdb *uploadSeqDb(cdm_ctx *ctx, const MmDb &db) {
    if ((db.dbtype & 0x7FFFFFFF) != 1) unsupported("The MI355X path works on nucleotide sequence DBs only (dbtype " + std::to_string(db.dbtype & 0x7FFFFFFF) + " given)");
    std::vector<uint32_t> lens(db.size());
    for (size_t i = 0; i < db.size(); i++) lens[i] = db.len[i] >= 2 ? (uint32_t) (db.len[i] - 2) : 0;   // DBReader::getSeqLen
    cdm_seqdb *h = NULL;
    check(cdm_seqdb_upload(ctx, db.data(), db.off.data(), lens.data(), db.key.data(), db.ext.data(), db.size(), &h), "Can not load the sequence DB");
    return h;
}
// the entries of a device DB as DB payloads ("SEQ\n"), appended to c
void appendEntries(cdm_ctx *ctx, cdm_seqdb *h, OutChunk &c) {
    const uint64_t n = cdm_seqdb_size(h);
    if (n == 0) return;
    std::vector<uint32_t> lens(n), keys(n); std::vector<uint8_t> ext(n);
    check(cdm_seqdb_meta(ctx, h, lens.data(), keys.data(), ext.data()), "meta");
    std::vector<uint64_t> offs(n); uint64_t tot = 0;
    for (uint64_t i = 0; i < n; i++) { offs[i] = tot; tot += lens[i] + 2; }
    HVec<char> buf(tot);
    check(cdm_seqdb_download(ctx, h, buf.data(), offs.data()), "download");
    c.reserve(c.key.size() + n, c.data.size() + tot);
    for (uint64_t i = 0; i < n; i++) c.add(keys[i], buf.data() + offs[i], lens[i] + 1, ext[i]);
}
// a device DB as a text DB (+ its side-car): down() takes everything off the device, write() needs the device no more
struct SeqDbOut {
    uint64_t n = 0; HVec<uint32_t> keys, elen; HVec<uint8_t> ext; HVec<uint64_t> offs; HVec<char> buf; SeqSideHost side;
    void down(cdm_ctx *ctx, cdm_seqdb *h) {
        n = cdm_seqdb_size(h);
        if (n == 0) return;
        std::vector<uint32_t> lens(n);
        keys.resize(n); ext.resize(n);
        check(cdm_seqdb_meta(ctx, h, lens.data(), keys.data(), ext.data()), "meta");
        // the download buffer has the data file's layout already: "SEQ\n\0" per entry (the NULs are the buffer's zero fill)
        offs.resize(n); elen.resize(n); uint64_t tot = 0;
        for (uint64_t i = 0; i < n; i++) { offs[i] = tot; elen[i] = lens[i] + 2; tot += lens[i] + 2; }
        buf.resize(tot);
        buf[tot - 1] = '\0';           // the download writes [0, tot - 1): "SEQ\n" per entry and the NULs between them; the last entry's NUL is ours
        check(cdm_seqdb_download(ctx, h, buf.data(), offs.data()), "download");
        side.take(ctx, h);
    }
    void write(const std::string &path, int dbtype) {
        std::string err;
        if (n == 0) { if (!mmdbWriteChunks(path, dbtype, std::vector<OutChunk>(1), &err)) die(err); return; }
        side.begin(path, dbtype);       // (beside the text: other files)
        if (!mmdbWriteBlob(path, dbtype, buf.data(), buf.size(), keys.data(), offs.data(), elen.data(), ext.data(), n, &err)) die(err);
        side.commit();
    }
};
// The same in one go where the DB's data file takes its pieces straight off the device (a RAM-backed file system, where one writer per
// file is the fast way): the text streams through the library's pinned staging buffers into the file (cdm_seqdb_download_stream) while the
// index and the side-car go out in threads of their own - the text never stands in pageable memory.  false: not here (nothing was
// written; down() + write() as before).  The device DB is still needed until this returns.
static int seqDbPieceSink(void *user, const char *data, uint64_t offset, uint64_t bytes) { return mmdbWritePiece(*(int *) user, data, offset, bytes) ? 0 : 1; }
bool streamSeqDb(cdm_ctx *ctx, cdm_seqdb *h, const std::string &path, int dbtype) {
    const uint64_t n = cdm_seqdb_size(h);
    if (n == 0) return false;
    HVec<uint32_t> keys(n), elen(n); HVec<uint8_t> ext(n); HVec<uint64_t> offs(n); std::vector<uint32_t> lens(n);
    check(cdm_seqdb_meta(ctx, h, lens.data(), keys.data(), ext.data()), "meta");
    uint64_t tot = 0;
    for (uint64_t i = 0; i < n; i++) { offs[i] = tot; elen[i] = lens[i] + 2; tot += lens[i] + 2; }
    int fd = mmdbOpenStreamedData(path, tot);
    if (fd < 0) return false;
    SeqSideHost side; side.take(ctx, h); side.begin(path, dbtype);
    bool okIx = true; std::string errIx;
    std::thread ix([&] { okIx = mmdbWriteBlob(path, dbtype, nullptr, 0, keys.data(), offs.data(), elen.data(), ext.data(), n, &errIx, MMDB_DATA_ELSEWHERE); });
    const int rc = cdm_seqdb_download_stream(ctx, h, offs.data(), 64u << 20, seqDbPieceSink, &fd);
    const bool okClose = close(fd) == 0;
    ix.join();
    if (rc != CDM_OK) die(std::string("download: ") + cdm_last_error());
    if (!okClose) die("Could not write " + path);
    if (!okIx) die(errIx);
    side.commit();
    return true;
}
void writeSeqDb(cdm_ctx *ctx, cdm_seqdb *h, const std::string &path, int dbtype) { if (streamSeqDb(ctx, h, path, dbtype)) return; SeqDbOut o; o.down(ctx, h); o.write(path, dbtype); }
// ---- text codecs
// decimal text, two digits per division
char *utoa(unsigned long long v, char *p) {
    static const char D2[] = "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
                             "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
    if (v < 10) { *p++ = (char) ('0' + v); return p; }
    if (v < 100) { memcpy(p, D2 + 2 * v, 2); return p + 2; }
    char b[24]; int n = 24;
    while (v >= 100) { const unsigned r = (unsigned) (v % 100); v /= 100; n -= 2; memcpy(b + n, D2 + 2 * r, 2); }
    if (v >= 10) { n -= 2; memcpy(b + n, D2 + 2 * v, 2); } else b[--n] = (char) ('0' + v);
    memcpy(p, b + n, (size_t) (24 - n));
    return p + (24 - n);
}
char *itoa(long long v, char *p) { if (v < 0) { *p++ = '-'; return utoa((unsigned long long) -v, p); } return utoa(v, p); }
// ---- parsers for the tab-separated records (strtol / strtod walk the locale machinery: 10x the time of these)
// unsigned / signed decimal at d; d moves behind the digits (no digits: 0, as strtol gives)
inline unsigned long parseU(const char *&d) { unsigned long v = 0; while (*d >= '0' && *d <= '9') v = v * 10 + (unsigned long) (*d++ - '0'); return v; }
inline long parseI(const char *&d) { bool neg = false; if (*d == '-') { neg = true; d++; } else if (*d == '+') d++; const long v = (long) parseU(d); return neg ? -v : v; }
// a plain decimal "digits[.digits]" with at most 15 significant digits is mantissa / 10^k, both exact doubles: the division is
// correctly rounded, i.e. the very double strtod returns.  Anything else (exponents, inf, nan, long mantissas) goes to strtod.
inline double parseDecimal(const char *&d) {
    static const double P10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    const char *p = d; unsigned long m = 0; int digits = 0, frac = 0;
    while (*p >= '0' && *p <= '9') { m = m * 10 + (unsigned long) (*p++ - '0'); digits++; }
    if (*p == '.') { p++; while (*p >= '0' && *p <= '9') { m = m * 10 + (unsigned long) (*p++ - '0'); digits++; frac++; } }
    if (digits == 0 || digits > 15 || *p == 'e' || *p == 'E' || (*p != '\t' && *p != '\n' && *p != '\0')) { char *e; const double v = strtod(d, &e); d = e; return v; }
    d = p;
    return (double) m / P10[frac];
}
inline void skipField(const char *&d) { while (*d && *d != '\t' && *d != '\n') d++; }
char *seqIdText(float s, char *p) {   // Util::fastSeqIdToBuffer + the tab overwriting its last char (Util.cpp:278-307, Matcher.cpp:362-363)
    if (s == 1.0) { memcpy(p, "1.00", 4); return p + 4; }
    *p++ = '0'; *p++ = '.';
    if (s < 0.10) *p++ = '0';
    if (s < 0.01) *p++ = '0';
    return itoa((int) (s * 1000), p);
}
// contiguous slice [lo, hi) of n items for thread t of T
inline void sliceOf(size_t n, int t, int T, size_t &lo, size_t &hi) { lo = n * (size_t) t / T; hi = n * (size_t) (t + 1) / T; }
// Slices of the queries that carry about the same number of RECORDS each (off = CSR offsets of the records; a query costs one more
// unit).  The record lists are anything but even: kmermatcher's representatives - the longest, lowest-id sequences - hold the hits of
// whole k-mer groups, so at 10 M reads the first sixteenth of the queries owned most of the text and threads beyond the first
// added nothing.
inline std::vector<size_t> balancedSlices(const uint64_t *off, size_t n, int T) {
    std::vector<size_t> b(T + 1, n);
    b[0] = 0;
    const uint64_t total = off[n] + n;
    for (int t = 1; t < T; t++) {
        const uint64_t want = total / (uint64_t) T * (uint64_t) t;
        size_t lo = b[t - 1], hi = n;
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (off[mid] + mid < want) lo = mid + 1; else hi = mid; }
        b[t] = lo;
    }
    return b;
}
// number of records (= lines) of entry i of a result DB
inline uint32_t countLines(const MmDb &db, int64_t i) {
    if (i < 0) return 0;
    const char *d = db.entry((size_t) i), *e = d + (db.len[(size_t) i] ? db.len[(size_t) i] - 1 : 0);
    uint32_t n = 0;
    while (d < e) { const char *nl = (const char *) memchr(d, '\n', (size_t) (e - d)); n++; d = nl ? nl + 1 : e; }
    return n;
}
// CSR offsets of the queries' record lists: the lines are counted first (memchr speed), the records then go straight to their place
void countRecords(const MmDb &res, const MmDb &seq, HVec<uint64_t> &off) {
    const size_t n = seq.size();
    off.resize(n + 1);
    off[0] = 0;
#pragma omp parallel for schedule(dynamic, 4096)
    for (size_t i = 0; i < n; i++) off[i + 1] = countLines(res, res.idOf(seq.key[i]));
    for (size_t i = 0; i < n; i++) off[i + 1] += off[i];
}
void parseAlnDb(const MmDb &aln, const MmDb &seq, HVec<uint64_t> &off, HVec<cdm_aln> &rec) {   // Matcher.cpp:274-353
    const double lam = 0x1.4478764a1b24ap-1, logk = log(0x1.a1c1e68ea2ab1p-2), LN2 = std::log(2.0);
    const int T = std::max(1, omp_get_max_threads());
    countRecords(aln, seq, off);
    rec.resize(off[seq.size()]);
    long badKey = -1, badEntry = -1;
    const std::vector<size_t> cut = balancedSlices(off.data(), seq.size(), T);
#pragma omp parallel for schedule(static, 1) num_threads(T)     // a loop over the slices: a smaller team still visits them all
    for (int t = 0; t < T; t++) {
        const size_t lo = cut[t], hi = cut[t + 1];
        for (size_t i = lo; i < hi; i++) {
            if (off[i + 1] == off[i]) continue;
            const char *d = aln.entry((size_t) aln.idOf(seq.key[i]));
            cdm_aln *out = rec.data() + off[i], *const end = rec.data() + off[i + 1];
            while (*d && out < end) {
                cdm_aln r;
                auto tab = [&] { if (*d == '\t') d++; };
                const uint32_t tkey = (uint32_t) parseU(d); tab();
                const int bits = (int) parseI(d); tab();
                r.seq_id = (float) parseDecimal(d); tab();
                skipField(d); tab();                                   // the E-value is not read back
                r.q_start = (int) parseI(d); tab(); r.q_end = (int) parseI(d); tab(); skipField(d); tab();
                r.db_start = (int) parseI(d); tab(); r.db_end = (int) parseI(d); tab(); skipField(d);
                const int64_t tt = seq.idOf(tkey);
                if (tt < 0) {
#pragma omp critical
                    badKey = (long) tkey;
                    break;
                }
                r.target = (uint32_t) tt; r.ident = -1;
                r.raw_score = static_cast<int>((logk + bits * LN2) / lam + 0.5);   // computeRawScoreFromBitScore as the consumers do
                *out++ = r;
                while (*d && *d != '\n') d++;
                if (*d == '\n') d++;
            }
            // the index length and the entry's NUL must agree (an embedded NUL or a wrong length would leave records unparsed or unfilled)
            if (badKey < 0 && (out != end || *d)) {
#pragma omp critical
                badEntry = (long) seq.key[i];
            }
        }
    }
    if (badKey >= 0) die("Invalid database read for key " + std::to_string(badKey));
    if (badEntry >= 0) die("Invalid database read: the entry of key " + std::to_string(badEntry) + " does not end where its index length says");
}
cdm_ancient_params ancientParams(Args &a) {
    cdm_ancient_params p;
    p.seq_id_thr = fflag(a, "--min-seq-id", 0.9f); p.corr_reads_ry_seq_id = fflag(a, "--min-ryseq-id-corr-reads", 0.99f); p.ry_seq_id_thr = 0.99f;
    p.rand_align_penal = fflag(a, "--ext-random-align", 0.85f); p.excess_penal = fflag(a, "--excess-penalty", 0.0625f);
    p.likelihood_threshold = fflag(a, "--likelihood-ratio-threshold", 0.5f); p.unsafe = (int) iflag(a, "--unsafe", 0);
    p.min_cov_safe = (int) iflag(a, "--min-cov-safe", 5); p.max_seq_len = (uint64_t) iflag(a, "--max-seq-len", 65535);
    return p;
}

// QueryMatcher::prefilterHitToBuffer per hit, one DB entry per query (kmermatcher.cpp:815-930)
void formatPrefDb(const MmDb &seq, const uint64_t *off, const cdm_hit *rec, std::vector<OutChunk> &chunks) {
    const int T = std::max(1, omp_get_max_threads());
    chunks.clear(); chunks.resize(T);
    const size_t MAXREC = 10 + 1 + 11 + 1 + 6 + 1;      // "%u\t%d\t%d\n" with a short diagonal
    const std::vector<size_t> cut = balancedSlices(off, seq.size(), T);
#pragma omp parallel for schedule(static, 1) num_threads(T)     // a loop over the slices: a smaller team still visits them all
    for (int t = 0; t < T; t++) {
        const size_t lo = cut[t], hi = cut[t + 1];
        OutChunk &c = chunks[t];
        c.reserve(hi - lo, (off[hi] - off[lo]) * MAXREC + (hi - lo));
        for (size_t i = lo; i < hi; i++) {
            char *const w0 = c.open((off[i + 1] - off[i]) * MAXREC), *w = w0;
            for (uint64_t h = off[i]; h < off[i + 1]; h++) {   // QueryMatcher::prefilterHitToBuffer
                w = utoa(seq.keyOf(rec[h].target), w); *w++ = '\t'; w = itoa(rec[h].score, w); *w++ = '\t'; w = itoa((short) rec[h].diagonal, w); *w++ = '\n';
            }
            // representatives' records carry wasExtended 0, fill-in records the sequence's flag (kmermatcher.cpp:727, DBWriter default)
            c.close(seq.key[i], w0, w, (off[i + 1] - off[i] > 1) ? 0 : seq.ext[i]);
        }
    }
}
// QueryMatcher::parsePrefilterHits: the prefilter text as CSR over the query ids
void parsePrefDb(const MmDb &pref, const MmDb &seq, HVec<uint64_t> &off, HVec<cdm_hit> &rec) {
    const int T = std::max(1, omp_get_max_threads());
    countRecords(pref, seq, off);
    rec.resize(off[seq.size()]);
    long badKey = -1, badEntry = -1;
    const std::vector<size_t> cut = balancedSlices(off.data(), seq.size(), T);
#pragma omp parallel for schedule(static, 1) num_threads(T)     // a loop over the slices: a smaller team still visits them all
    for (int t = 0; t < T; t++) {
        const size_t lo = cut[t], hi = cut[t + 1];
        for (size_t i = lo; i < hi; i++) {
            if (off[i + 1] == off[i]) continue;
            const char *d = pref.entry((size_t) pref.idOf(seq.key[i]));
            cdm_hit *out = rec.data() + off[i], *const end = rec.data() + off[i + 1];
            while (*d && out < end) {
                cdm_hit h;
                const uint32_t tkey = (uint32_t) parseU(d); if (*d == '\t') d++; h.score = (int) parseI(d); if (*d == '\t') d++; h.diagonal = (short) parseI(d);
                const int64_t tt = seq.idOf(tkey);
                if (tt < 0) {
#pragma omp critical
                    badKey = (long) tkey;
                    break;
                }
                h.target = (uint32_t) tt; *out++ = h;
                while (*d && *d != '\n') d++;
                if (*d == '\n') d++;
            }
            if (badKey < 0 && (out != end || *d)) {
#pragma omp critical
                badEntry = (long) seq.key[i];
            }
        }
    }
    if (badKey >= 0) die("Invalid database read for key " + std::to_string(badKey));
    if (badEntry >= 0) die("Invalid database read: the entry of key " + std::to_string(badEntry) + " does not end where its index length says");
}
// Matcher::resultToBuffer per record, one DB entry per query that has a prefilter entry (rescorediagonal.cpp:145-356)
// (asParsed, if given: every record as parseAlnDb would read it back from the text written here - raw score recomputed from the bit
// score, the coordinates, and in `ident` the identity in thousandths as the text truncates it: what writeAlnsSide takes.
// pref == NULL: every query has a prefilter entry - the hits came from kmermatcher's own side-car.)
void formatAlnDb(const MmDb &seq, const MmDb *pref, const uint64_t *aoff, const cdm_aln *arec, uint64_t dbRes, std::vector<OutChunk> &chunks, cdm_aln *asParsed = NULL) {
    const double lam = 0x1.4478764a1b24ap-1, logk = log(0x1.a1c1e68ea2ab1p-2), LN2 = std::log(2.0);
    const int T = std::max(1, omp_get_max_threads());
    chunks.clear(); chunks.resize(T);
    const size_t MAXREC = 10 + 11 + 5 + 14 + 6 * 11 + 10;      // key, bits, seq.id., E-value, six coordinates, separators
    const std::vector<size_t> cut = balancedSlices(aoff, seq.size(), T);
#pragma omp parallel for schedule(static, 1) num_threads(T)     // a loop over the slices: a smaller team still visits them all
    for (int t = 0; t < T; t++) {
        const size_t lo = cut[t], hi = cut[t + 1];
        OutChunk &c = chunks[t];
        c.reserve(hi - lo, (aoff[hi] - aoff[lo]) * MAXREC + (hi - lo));
        // the E-value text and bit score of (raw score, query length) recur all over a read set: formatted once per thread
        struct EvalText { int qLen = -1, score = -1, bits = 0; unsigned char n = 0; char txt[15]; };
        std::vector<EvalText> cache(1u << 14);
        for (size_t i = lo; i < hi; i++) {
            if (pref && pref->idOf(seq.key[i]) < 0) continue;
            const int qLen = (int) (seq.len[i] - 2);
            char *const w0 = c.open((aoff[i + 1] - aoff[i]) * MAXREC), *w = w0;
            for (uint64_t r = aoff[i]; r < aoff[i + 1]; r++) {   // Matcher::resultToBuffer (Matcher.cpp:356-404)
                const cdm_aln &x = arec[r];
                if (r + 16 < aoff[hi]) __builtin_prefetch(&seq.len[arec[r + 16].target]);      // (a random look-up per record: on its way 16 records ahead)
                const int alnLen = std::max(abs(x.q_end - x.q_start), abs(x.db_end - x.db_start)) + 1;
                const float sid = static_cast<float>(x.ident) / static_cast<float>(alnLen);
                w = utoa(seq.keyOf(x.target), w); *w++ = '\t';
                EvalText &ev = cache[((uint32_t) x.raw_score * 2654435761u ^ (uint32_t) qLen * 40503u) >> 18];
                if (ev.qLen != qLen || ev.score != x.raw_score) {
                    ev.qLen = qLen; ev.score = x.raw_score; ev.bits = cdm_bit_score(x.raw_score);
                    ev.n = (unsigned char) snprintf(ev.txt, sizeof(ev.txt), "%.3E", cdm_evalue(x.raw_score, qLen, dbRes));
                }
                w = itoa(ev.bits, w); *w++ = '\t';
                w = seqIdText(sid, w); *w++ = '\t';
                memcpy(w, ev.txt, ev.n); w += ev.n; *w++ = '\t';
                w = itoa(x.q_start, w); *w++ = '\t'; w = itoa(x.q_end, w); *w++ = '\t'; w = itoa(qLen, w); *w++ = '\t';
                w = itoa(x.db_start, w); *w++ = '\t'; w = itoa(x.db_end, w); *w++ = '\t'; w = itoa((int) (seq.len[x.target] - 2), w); *w++ = '\n';
                if (asParsed) {
                    cdm_aln p = x;
                    p.raw_score = static_cast<int>((logk + ev.bits * LN2) / lam + 0.5); p.ident = (int) seqIdTo1000(sid); p.seq_id = seqIdFrom1000((uint32_t) p.ident);
                    asParsed[r] = p;
                }
            }
            c.close(seq.key[i], w0, w, 0);
        }
    }
}

// the same records for the result of the Hamming mode: one entry per query that has a prefilter entry (rescorediagonal.cpp:350), flag 0
void formatRescoredPrefDb(const MmDb &seq, const MmDb &pref, const uint64_t *off, const cdm_hit *rec, std::vector<OutChunk> &chunks) {
    const int T = std::max(1, omp_get_max_threads());
    chunks.clear(); chunks.resize(T);
    const size_t MAXREC = 10 + 1 + 11 + 1 + 6 + 1;
    const std::vector<size_t> cut = balancedSlices(off, seq.size(), T);
#pragma omp parallel for schedule(static, 1) num_threads(T)     // a loop over the slices: a smaller team still visits them all
    for (int t = 0; t < T; t++) {
        const size_t lo = cut[t], hi = cut[t + 1];
        OutChunk &c = chunks[t];
        c.reserve(hi - lo, (off[hi] - off[lo]) * MAXREC + (hi - lo));
        for (size_t i = lo; i < hi; i++) {
            if (pref.idOf(seq.keyOf(i)) < 0) continue;
            char *const w0 = c.open((off[i + 1] - off[i]) * MAXREC), *w = w0;
            for (uint64_t h = off[i]; h < off[i + 1]; h++) {
                w = utoa(seq.keyOf(rec[h].target), w); *w++ = '\t'; w = itoa(rec[h].score, w); *w++ = '\t'; w = itoa((short) rec[h].diagonal, w); *w++ = '\n';
            }
            c.close(seq.keyOf(i), w0, w, 0);
        }
    }
}

// ---- record side-cars: the CSR a consumer's parser would produce from the text, as cdm_hits_upload / cdm_alns_upload take it
bool writeHitsSide(const std::string &path, const MmDb &seq, const uint64_t *off, const cdm_hit *rec, int dbtype, SideHeader *hdr) {
    if (!sideEnabled()) return false;
    const uint64_t n = seq.size(), count = off[n];
    bool fits = true;
#pragma omp parallel for reduction(&& : fits) schedule(static)
    for (uint64_t i = 0; i < count; i++) { cdm_hit h = rec[i]; h.diagonal = (short) h.diagonal; fits = fits && fitsHit8(h); }
    const uint64_t hash = sideKeyHash(seq.key.data(), n);
    if (fits) {
        HVec<SideHit8> c(count);
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < count; i++) { c[i].target = rec[i].target; c[i].score = (int16_t) rec[i].score; c[i].diagonal = (int16_t) (short) rec[i].diagonal; }
        const SidePiece pc[2] = {{off, (n + 1) * 8}, {c.data(), count * sizeof(SideHit8)}};
        return sideWriteBody(path, SIDE_HITS, SIDE_F_COMPACT, n, count, n, hash, dbtype, pc, 2, hdr);
    } else {
        HVec<cdm_hit> c(count);
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < count; i++) { c[i] = rec[i]; c[i].diagonal = (short) rec[i].diagonal; }       // (the text holds the diagonal as a short: QueryMatcher.h:114-126)
        const SidePiece pc[2] = {{off, (n + 1) * 8}, {c.data(), count * sizeof(cdm_hit)}};
        return sideWriteBody(path, SIDE_HITS, 0, n, count, n, hash, dbtype, pc, 2, hdr);
    }
}
// hits / alignments of the DB at `path` from its side-car if that belongs to these files and to this sequence DB
bool readHitsSide(const std::string &path, const MmDb &seq, HVec<uint64_t> &off, HVec<cdm_hit> &rec, int *dbtype) {
    SideFile f;
    if (!sideOpen(path, SIDE_HITS, f)) return false;
    const SideHeader &h = *f.h;
    if (h.n != seq.size() || h.seqN != seq.size() || h.seqKeyHash != sideKeyHash(seq.key.data(), seq.size())) return false;
    off.resize(h.n + 1); rec.resize(h.count);
    memcpy(off.data(), f.section(0), (h.n + 1) * 8);
    if (off[h.n] != h.count) return false;
    if (h.flags & SIDE_F_COMPACT) {
        const SideHit8 *c = (const SideHit8 *) f.section(1);
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < h.count; i++) { rec[i].target = c[i].target; rec[i].score = c[i].score; rec[i].diagonal = c[i].diagonal; }
    } else memcpy(rec.data(), f.section(1), h.count * sizeof(cdm_hit));
    *dbtype = h.dbtype;
    return true;
}
bool writeAlnsSide(const std::string &path, const MmDb &seq, const uint64_t *off, const cdm_aln *asParsed, SideHeader *hdr) {
    if (!sideEnabled()) return false;
    const uint64_t n = seq.size(), count = off[n];
    bool fits = true;
#pragma omp parallel for reduction(&& : fits) schedule(static)
    for (uint64_t i = 0; i < count; i++) {
        const cdm_aln &r = asParsed[i];
        auto s16 = [](int v) { return v >= -32768 && v <= 32767; };
        fits = fits && r.raw_score >= 0 && r.raw_score <= 65535 && s16(r.q_start) && s16(r.q_end) && s16(r.db_start) && s16(r.db_end);
    }
    const uint64_t hash = sideKeyHash(seq.key.data(), n);
    if (fits) {
        HVec<SideAln16> c(count);
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < count; i++) {
            const cdm_aln &r = asParsed[i];
            c[i].target = r.target; c[i].rawScore = (uint16_t) r.raw_score; c[i].seqId1000 = (uint16_t) r.ident;       // (ident carries the identity in thousandths here, see formatAlnDb)
            c[i].qStart = (int16_t) r.q_start; c[i].qEnd = (int16_t) r.q_end; c[i].dbStart = (int16_t) r.db_start; c[i].dbEnd = (int16_t) r.db_end;
        }
        const SidePiece pc[2] = {{off, (n + 1) * 8}, {c.data(), count * sizeof(SideAln16)}};
        return sideWriteBody(path, SIDE_ALNS, SIDE_F_COMPACT, n, count, n, hash, 5, pc, 2, hdr);
    } else {
        const SidePiece pc[2] = {{off, (n + 1) * 8}, {asParsed, count * sizeof(cdm_aln)}};
        return sideWriteBody(path, SIDE_ALNS, 0, n, count, n, hash, 5, pc, 2, hdr);
    }
}
bool readAlnsSide(const std::string &path, const MmDb &seq, HVec<uint64_t> &off, HVec<cdm_aln> &rec) {
    SideFile f;
    if (!sideOpen(path, SIDE_ALNS, f)) return false;
    const SideHeader &h = *f.h;
    if (h.n != seq.size() || h.seqN != seq.size() || h.seqKeyHash != sideKeyHash(seq.key.data(), seq.size())) return false;
    off.resize(h.n + 1); rec.resize(h.count);
    memcpy(off.data(), f.section(0), (h.n + 1) * 8);
    if (off[h.n] != h.count) return false;
    if (h.flags & SIDE_F_COMPACT) {
        const SideAln16 *c = (const SideAln16 *) f.section(1);
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < h.count; i++) {
            cdm_aln r;
            r.target = c[i].target; r.raw_score = c[i].rawScore; r.ident = -1; r.q_start = c[i].qStart; r.q_end = c[i].qEnd; r.db_start = c[i].dbStart; r.db_end = c[i].dbEnd;
            r.seq_id = seqIdFrom1000(c[i].seqId1000);
            rec[i] = r;
        }
    } else {
        const cdm_aln *c = (const cdm_aln *) f.section(1);
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < h.count; i++) { cdm_aln r = c[i]; r.seq_id = seqIdFrom1000((uint32_t) r.ident); r.ident = -1; rec[i] = r; }
    }
    return true;
}

// the record side-cars beside the text: their sections are built and written in a thread while the caller formats and writes the text DB
struct SideThread {
    std::thread th; SideHeader hdr; bool ok = false; std::string path;
    template <typename F> void begin(const std::string &p, F body) { if (!sideEnabled()) return; path = p; th = std::thread([this, body] { ok = body(&hdr); }); }
    void commit() { if (th.joinable()) { th.join(); if (ok) sideCommit(path, &hdr); } }
};

int kmermatcher(Args &a) {
    if (a.pos.size() < 2) die("Usage: carpedeam kmermatcher <i:sequenceDB> <o:prefilterDB>");
    checkFlags("kmermatcher", a, KMERMATCHER_FLAGS);
    Laps laps;
    DeviceStart dev; dev.begin((const MmDb *) NULL, NULL);                 // (the context comes up while the DB is mapped and its index parsed)
    SeqInput in; in.load(a.pos[0]); MmDb &seq = in.db; std::string err;
    laps.lap(in.fromSide ? "sequence side-car mapped" : "DB files mapped");
    dev.join(); cdm_ctx *ctx = dev.ctx; laps.lap("device context");
    cdm_seqdb *db = NULL;
    if (in.fromSide) { if ((seq.dbtype & 0x7FFFFFFF) != 1) unsupported("The MI355X path works on nucleotide sequence DBs only"); if (!importSeqSide(ctx, in.side, &db)) die(std::string("Can not load the sequence DB: ") + cdm_last_error()); }
    else db = uploadSeqDb(ctx, seq);
    laps.lap("sequences up");
    cdm_kmer_params p;
    p.kmer_size = (int) iflag(a, "-k", 15); p.kmers_per_seq = (int) iflag(a, "--kmer-per-seq", 21); p.kmers_per_seq_scale = fflag(a, "--kmer-per-seq-scale", 0.2f);
    p.hash_shift = (uint64_t) iflag(a, "--hash-shift", 67); p.ignore_multi_kmer = (int) iflag(a, "--ignore-multi-kmer", 0);
    p.include_only_extendable = (int) iflag(a, "--include-only-extendable", 0); p.cov_mode = (int) iflag(a, "--cov-mode", 0); p.cov_thr = fflag(a, "-c", 0.8f);
    cdm_hits *hits = NULL;
    check(cdm_kmermatch(ctx, db, &p, &hits), "kmermatcher");
    HVec<uint64_t> off(seq.size() + 1); HVec<cdm_hit> rec(cdm_hits_count(hits));
    check(cdm_hits_download(ctx, hits, off.data(), rec.data()), "download");
    SeqSideHost inSide; if (!in.fromSide) inSide.take(ctx, db);       // (the next modules of the workflow read this DB again)
    laps.lap("kernels, hits down");
    DeviceEnd end; end.begin(ctx, hits, NULL, db, NULL); laps.lap("device memory back to the driver");
    SideThread hitsSide; hitsSide.begin(a.pos[1], [&](SideHeader *h) { return writeHitsSide(a.pos[1], seq, off.data(), rec.data(), 14, h); });
    inSide.begin(a.pos[0], seq.dbtype);
    std::vector<OutChunk> chunks;
    formatPrefDb(seq, off.data(), rec.data(), chunks);
    laps.lap("prefilter text formatted");
    if (!mmdbWriteChunks(a.pos[1], 14, chunks, &err, true)) die(err);   // DBTYPE_PREFILTER_REV_RES (kmermatcher.cpp:682)
    laps.lap("result DB written");
    hitsSide.commit(); inSide.commit();
    laps.lap("(side-cars, written beside: waited)");
    finishModule(EXIT_SUCCESS);
    return EXIT_SUCCESS;
}

int rescorediagonal(Args &a) {
    if (a.pos.size() < 4) die("Usage: carpedeam rescorediagonal <i:queryDB> <i:targetDB> <i:prefilterDB> <o:resultDB>");
    const bool hamming = a.flag.count("--rescore-mode") && a.flag["--rescore-mode"] == "0" && a.flag.count("--wrapped-scoring") && a.flag["--wrapped-scoring"] == "1";
    checkFlags("rescorediagonal", a, hamming ? RESCORE_HAMMING_FLAGS : RESCORE_FLAGS);
    if (a.pos[0] != a.pos[1]) unsupported("rescorediagonal: query and target DB must be the same on the MI355X path");
    if (hamming) {      // linclust's pre-clustering of the assembled contigs: prefilter records in, prefilter records out
        MmDb seq, pref; std::string err; if (!seq.load(a.pos[1], &err)) die(err);
        DeviceStart dev; dev.begin(&seq, NULL);
        if (!pref.load(a.pos[2], &err)) die(err);
        HVec<uint64_t> off; HVec<cdm_hit> rec;
        parsePrefDb(pref, seq, off, rec);
        dev.join(); cdm_ctx *ctx = dev.ctx; cdm_seqdb *db = dev.db;
        cdm_hits *hits = NULL, *kept = NULL;
        check(cdm_hits_upload(ctx, db, off.data(), rec.data(), &hits), "upload");
        cdm_hamming_params p;
        p.seq_id_thr = fflag(a, "--min-seq-id", 0.0f); p.eval_thr = a.flag.count("-e") ? strtod(a.flag["-e"].c_str(), NULL) : 0.001;
        p.cov_mode = (int) iflag(a, "--cov-mode", 0); p.cov_thr = fflag(a, "-c", 0.0f); p.seq_id_mode = (int) iflag(a, "--seq-id-mode", 0); p.min_aln_len = (int) iflag(a, "--min-aln-len", 0);
        p.reverse_prefilter = (pref.dbtype & 0x7FFFFFFF) == 14;
        check(cdm_rescore_hamming(ctx, db, hits, &p, &kept), "rescorediagonal");
        HVec<uint64_t> koff(seq.size() + 1); HVec<cdm_hit> krec(cdm_hits_count(kept));
        check(cdm_hits_download(ctx, kept, koff.data(), krec.data()), "download");
        std::vector<OutChunk> chunks;
        formatRescoredPrefDb(seq, pref, koff.data(), krec.data(), chunks);
        if (!mmdbWriteChunks(a.pos[3], pref.dbtype, chunks, &err, true)) die(err);
        cdm_hits_free(kept); cdm_hits_free(hits); cdm_seqdb_free(db); cdm_ctx_destroy(ctx);
        return EXIT_SUCCESS;
    }
    if (!a.flag.count("--rescore-mode")) unsupported("rescorediagonal: --rescore-mode 3 has to be given (the module's default, 0 = Hamming distance, is implemented with --wrapped-scoring 1 only on the MI355X path)");
    Laps laps;
    SeqInput in; in.load(a.pos[1]); MmDb &seq = in.db; MmDb pref; std::string err;
    DeviceStart dev; dev.begin(&in, NULL);                  // context + sequence upload while the prefilter records are read
    HVec<uint64_t> off; HVec<cdm_hit> rec;
    int prefType = 0;
    const bool hitsFromSide = readHitsSide(a.pos[2], seq, off, rec, &prefType);
    if (hitsFromSide) laps.lap("side-cars mapped, prefilter records read");
    else {
        if (!pref.load(a.pos[2], &err)) die(err);
        laps.lap("DB files mapped");
        parsePrefDb(pref, seq, off, rec);
        laps.lap("prefilter text parsed");
    }
    dev.join(); cdm_ctx *ctx = dev.ctx; cdm_seqdb *db = dev.db; laps.lap("(device context, sequences up: waited)");
    cdm_hits *hits = NULL; cdm_alns *alns = NULL;
    check(cdm_hits_upload(ctx, db, off.data(), rec.data(), &hits), "upload"); laps.lap("  (hits up)");
    cdm_rescore_params p;
    p.seq_id_thr = fflag(a, "--min-seq-id", 0.0f); p.eval_thr = a.flag.count("-e") ? strtod(a.flag["-e"].c_str(), NULL) : 0.001;
    p.cov_mode = (int) iflag(a, "--cov-mode", 0); p.cov_thr = fflag(a, "-c", 0.0f); p.seq_id_mode = (int) iflag(a, "--seq-id-mode", 0); p.min_aln_len = (int) iflag(a, "--min-aln-len", 0);
    check(cdm_rescore(ctx, db, hits, &p, &alns), "rescorediagonal"); laps.lap("  (kernels)");
    HVec<uint64_t> aoff(seq.size() + 1); HVec<cdm_aln> arec(cdm_alns_count(alns));
    check(cdm_alns_download(ctx, alns, aoff.data(), arec.data()), "download");
    const uint64_t dbResidues = cdm_seqdb_residues(db);
    SeqSideHost inSide; if (!in.fromSide) inSide.take(ctx, db);
    laps.lap("hits up, kernels, records down");
    DeviceEnd end; end.begin(ctx, hits, alns, db, NULL); laps.lap("device memory back to the driver");
    std::vector<OutChunk> chunks;
    HVec<cdm_aln> asParsed; if (sideEnabled()) asParsed.resize(arec.size());
    formatAlnDb(seq, hitsFromSide ? NULL : &pref, aoff.data(), arec.data(), dbResidues, chunks, sideEnabled() ? asParsed.data() : NULL);
    laps.lap("alignment text formatted");
    // (an alignment side-car holds every query: only when every query has a prefilter entry - kmermatcher's output - does the text too)
    bool allPresent = hitsFromSide;
    if (!allPresent) { allPresent = true; for (size_t i = 0; i < seq.size() && allPresent; i++) allPresent = pref.idOf(seq.key[i]) >= 0; }
    SideThread alnSide; if (allPresent) alnSide.begin(a.pos[3], [&](SideHeader *h) { return writeAlnsSide(a.pos[3], seq, aoff.data(), asParsed.data(), h); });
    inSide.begin(a.pos[1], seq.dbtype);
    if (!mmdbWriteChunks(a.pos[3], 5, chunks, &err, true)) die(err);
    laps.lap("result DB written");
    alnSide.commit(); inSide.commit();
    laps.lap("(side-cars, written beside: waited)");
    finishModule(EXIT_SUCCESS);
    return EXIT_SUCCESS;
}

int ancientModule(Args &a, int mode) {      // 0 ancient_correction, 1 ancient_read_assemble, 2 ancient_contig_merge
    const bool assemble = mode == 1;
    const char *name = mode == 0 ? "ancient_correction" : mode == 1 ? "ancient_read_assemble" : "ancient_contig_merge";
    if (a.pos.size() < 3) die(std::string("Usage: carpedeam ") + name + " <i:sequenceDB> <i:alnResult> <o:reprSeqDB>");
    checkFlags(name, a, ANCIENT_FLAGS);
    if (mode >= 1 && !a.flag.count("--rescore-mode")) unsupported(std::string(name) + ": --rescore-mode 3 has to be given (the module's default, 0 = Hamming distance, is not implemented on the MI355X path)");
    Laps laps;
    SeqInput in; in.load(a.pos[0]); MmDb &seq = in.db; MmDb aln; std::string err;
    const std::string damage = a.flag.count("--ancient-damage") ? a.flag["--ancient-damage"] : "";
    DeviceStart dev; dev.begin(&in, &damage);               // context, damage tables, sequence upload while the alignment records are read
    HVec<uint64_t> off; HVec<cdm_aln> rec;
    if (readAlnsSide(a.pos[1], seq, off, rec)) laps.lap("side-cars mapped, alignment records read");
    else {
        if (!aln.load(a.pos[1], &err)) die(err);
        laps.lap("DB files mapped");
        parseAlnDb(aln, seq, off, rec); laps.lap("alignment text parsed");
    }
    dev.join(); cdm_ctx *ctx = dev.ctx; cdm_seqdb *db = dev.db; laps.lap("(device context, damage tables, sequences up: waited)");
    cdm_alns *alns = NULL; cdm_seqdb *out = NULL;
    check(cdm_alns_upload(ctx, db, off.data(), rec.data(), &alns), "upload");
    cdm_ancient_params p = ancientParams(a);
    if (assemble) check(cdm_extend(ctx, db, alns, &p, &out, NULL), "ancient_read_assemble");
    else if (mode == 2) check(cdm_contig_merge(ctx, db, alns, &p, fflag(a, "--min-merge-seq-id", 0.99f), &out), "ancient_contig_merge");
    else check(cdm_correct(ctx, db, alns, &p, &out), "ancient_correction");
    laps.lap("records up, kernels");
    SeqSideHost inSide; if (!in.fromSide) inSide.take(ctx, db);
    inSide.begin(a.pos[0], seq.dbtype);
    if (streamSeqDb(ctx, out, a.pos[2], seq.dbtype)) {      // (tmpfs: the text goes from the device into the data file piece by piece)
        laps.lap("sequences down into the DB's data file, index and side-car beside it");
        DeviceEnd end; end.begin(ctx, NULL, alns, db, out); laps.lap("device memory back to the driver");
    } else {
        SeqDbOut o; o.down(ctx, out);
        laps.lap("sequences down");
        DeviceEnd end; end.begin(ctx, NULL, alns, db, out); laps.lap("device memory back to the driver");
        o.write(a.pos[2], seq.dbtype); laps.lap("DB written, its side-car beside it");
    }
    inSide.commit();
    finishModule(EXIT_SUCCESS);
    return EXIT_SUCCESS;
}
// cyclecheck <i:sequenceDB> <o:sequenceDBcycle> (src/assembler/cyclecheck.cpp:30-269; flags: LocalParameters.h:183-187)
const FlagSpec CYCLECHECK_FLAGS[] = {{"--max-seq-len", 'U', 0, 0}, {"--chop-cycle", 'U', 0, 0}, {"--threads", 'N', 0, 0}, {"-v", 'N', 0, 0}, {0, 0, 0, 0}};
int cyclecheck(Args &a) {
    if (a.pos.size() < 2) die("Usage: carpedeam cyclecheck <i:sequenceDB> <o:sequenceDBcycle>");
    checkFlags("cyclecheck", a, CYCLECHECK_FLAGS);
    MmDb seq; std::string err; if (!seq.load(a.pos[0], &err)) die(err);
    if ((seq.dbtype & 0x7FFFFFFF) != 1) die("Module cyclecheck only supports nucleotide input database");
    cdm_ctx *ctx = openCtx();
    cdm_seqdb *db = uploadSeqDb(ctx, seq), *cyc = NULL;
    const long maxLen = std::min(iflag(a, "--max-seq-len", 65535), 0xFFFFFFFFl);
    check(cdm_cyclecheck(ctx, db, (uint32_t) maxLen, iflag(a, "--chop-cycle", 0) != 0, &cyc, NULL, NULL), "cyclecheck");
    writeSeqDb(ctx, cyc, a.pos[1], 1);
    cdm_seqdb_free(cyc); cdm_seqdb_free(db); cdm_ctx_destroy(ctx);
    return EXIT_SUCCESS;
}
// Fused reads loop: the body of `while [ $STEP -lt $NUM_IT_READS ]` of data/nuclassemble.sh:100-146
// - kmermatcher, rescorediagonal, ancient_correction, ancient_read_assemble per iteration - in one process with every
// intermediate resident in HBM; no prefilter/alignment/correction text DB is written, only the final sequence DB.
// Not a module of the reference; the per-stage modules above remain the drop-in surface.
// linclust's host-side tail and the scripts' file modules (host/cluster.cpp); flag lists: Parameters.cpp clust / createsubdb / filterdb /
// threadsandcompression / onlyverbosity
int clusterModules(const std::string &cmd, Args &a) {
    static const FlagSpec CLUST_FLAGS[] = {{"--cluster-mode", 'U', 0, 0}, {"--max-iterations", 'N', 0, "connected-component mode only"}, {"--similarity-type", 'N', 0, "set-cover mode only"},
                                           {"--threads", 'N', 0, 0}, {"-v", 'N', 0, 0}, {"--compressed", 'V', "0", "compressed DBs are not implemented"}, {0, 0, 0, 0}};
    static const FlagSpec SUBDB_FLAGS[] = {{"--subdb-mode", 'U', 0, 0}, {"-v", 'N', 0, 0}, {"--id-mode", 'V', "0", "look-up mode is not implemented"}, {0, 0, 0, 0}};
    static const FlagSpec FILTER_FLAGS[] = {{"--filter-file", 'U', 0, 0}, {"--threads", 'N', 0, 0}, {"-v", 'N', 0, 0}, {"--compressed", 'V', "0", "compressed DBs are not implemented"},
                                            {"--filter-column", 'V', "1", "only the first column"}, {"--positive-filter", 'V', "1", "only positive filtering"}, {0, 0, 0, 0}};
    static const FlagSpec PLAIN_FLAGS[] = {{"--threads", 'N', 0, 0}, {"-v", 'N', 0, 0}, {"--db-load-mode", 'N', 0, 0}, {"--compressed", 'V', "0", "compressed DBs are not implemented"}, {0, 0, 0, 0}};
    // align (Parameters.cpp: par.align): what changes the result of a nucleotide, no-backtrace, no-realign run is read; modes this path
    // does not have are refused
    static const FlagSpec ALIGN_FLAGS[] = {
        {"-e", 'U', 0, 0}, {"--min-seq-id", 'U', 0, 0}, {"--min-aln-len", 'U', 0, 0}, {"--seq-id-mode", 'U', 0, 0}, {"-c", 'U', 0, 0}, {"--cov-mode", 'U', 0, 0}, {"--max-seq-len", 'U', 0, 0},
        {"--max-rejected", 'U', 0, 0}, {"--max-accept", 'U', 0, 0}, {"--wrapped-scoring", 'U', 0, 0}, {"--gap-open", 'U', 0, 0}, {"--gap-extend", 'U', 0, 0}, {"--zdrop", 'U', 0, 0},
        {"--alignment-mode", 'N', 0, "nucleotide alignments always compute score, coverage and identity (Matcher.cpp:88)"}, {"--comp-bias-corr", 'N', 0, "amino acids only"},
        {"--add-self-matches", 'N', 0, "query DB == target DB: the identity hit is kept anyway"}, {"--db-load-mode", 'N', 0, 0}, {"--pca", 'N', 0, "profiles only"}, {"--pcb", 'N', 0, "profiles only"},
        {"--realign-score-bias", 'N', 0, "no realignment"}, {"--realign-max-seqs", 'N', 0, "no realignment"}, {"--threads", 'N', 0, 0}, {"-v", 'N', 0, 0},
        {"-a", 'V', "0", "no backtrace output"}, {"--alignment-output-mode", 'V', "0", "only alignment records"}, {"--alt-ali", 'V', "0", "not supported for nucleotides in the reference either"},
        {"--realign", 'V', "0", "not implemented"}, {"--score-bias", 'V', "0", "not implemented"}, {"--sub-mat", 'V', "*nucleotide.out*", "only the nucleotide matrix"},
        {"--compressed", 'V', "0", "compressed DBs are not implemented"}, {0, 0, 0, 0}};
    std::string err; int rc = 0;
    auto need = [&](size_t n, const char *usage) { if (a.pos.size() < n) die(std::string("Usage: carpedeam ") + usage); };
    auto nuclValue = [&](const char *flag, long dflt) -> long {      // "5" or "nucl:5,aa:11"
        if (!a.flag.count(flag)) return dflt;
        const std::string &v = a.flag[flag];
        const size_t at = v.find("nucl:");
        return strtol(v.c_str() + (at == std::string::npos ? 0 : at + 5), NULL, 10);
    };
    if (cmd == "align") {
        need(4, "align <i:queryDB> <i:targetDB> <i:resultDB> <o:alignmentDB>"); checkFlags("align", a, ALIGN_FLAGS);
        AlignParams P;
        P.covThr = fflag(a, "-c", 0.0f); P.seqIdThr = fflag(a, "--min-seq-id", 0.0f); P.evalThr = a.flag.count("-e") ? strtod(a.flag["-e"].c_str(), NULL) : 0.001;
        P.covMode = (int) iflag(a, "--cov-mode", 0); P.seqIdMode = (int) iflag(a, "--seq-id-mode", 0); P.alnLenThr = (int) iflag(a, "--min-aln-len", 0);
        P.wrapped = iflag(a, "--wrapped-scoring", 0) != 0;
        P.gapOpen = (int) nuclValue("--gap-open", 5); P.gapExtend = (int) nuclValue("--gap-extend", 2); P.zdrop = (int) iflag(a, "--zdrop", 40);
        P.maxAccept = (unsigned) iflag(a, "--max-accept", INT_MAX); P.maxReject = (unsigned) iflag(a, "--max-rejected", INT_MAX); P.maxSeqLen = (size_t) iflag(a, "--max-seq-len", 65535);
        rc = alignModule(a.pos[0], a.pos[1], a.pos[2], a.pos[3], P, &err);
    } else if (cmd == "clust") {
        need(3, "clust <i:sequenceDB> <i:resultDB> <o:clusterDB>"); checkFlags("clust", a, CLUST_FLAGS);
        rc = clustModule(a.pos[0], a.pos[1], a.pos[2], (int) iflag(a, "--cluster-mode", 0), &err);
    } else if (cmd == "createsubdb") {
        need(3, "createsubdb <i:subsetFile|DB> <i:DB> <o:DB>"); checkFlags("createsubdb", a, SUBDB_FLAGS);
        rc = createsubdbModule(a.pos[0], a.pos[1], a.pos[2], (int) iflag(a, "--subdb-mode", 0), &err);
    } else if (cmd == "filterdb") {
        need(2, "filterdb <i:resultDB> <o:resultDB> --filter-file <file>"); checkFlags("filterdb", a, FILTER_FLAGS);
        if (!a.flag.count("--filter-file")) unsupported("filterdb: only the --filter-file mode is implemented on the MI355X path");
        rc = filterdbModule(a.pos[0], a.pos[1], a.flag["--filter-file"], &err);
    } else if (cmd == "mergeclusters") {
        need(3, "mergeclusters <i:sequenceDB> <o:clusterDB> <i:clusterDB1> ... <i:clusterDBn>"); checkFlags("mergeclusters", a, PLAIN_FLAGS);
        rc = mergeclustersModule(a.pos[0], a.pos[1], std::vector<std::string>(a.pos.begin() + 2, a.pos.end()), &err);
    } else if (cmd == "result2repseq") {
        need(3, "result2repseq <i:sequenceDB> <i:resultDB> <o:sequenceDB>"); checkFlags("result2repseq", a, PLAIN_FLAGS);
        rc = result2repseqModule(a.pos[0], a.pos[1], a.pos[2], &err);
    } else if (cmd == "rmdb") {
        need(1, "rmdb <i:DB>"); checkFlags("rmdb", a, PLAIN_FLAGS);
        rc = rmdbModule(a.pos[0]);
    } else {
        need(2, "mvdb <i:srcDB> <o:dstDB>"); checkFlags("mvdb", a, PLAIN_FLAGS);
        rc = mvdbModule(a.pos[0], a.pos[1], &err);
    }
    if (rc == 77) unsupported(err);
    if (rc) die(err);
    return EXIT_SUCCESS;
}

int readsLoop(Args &a) {
    if (a.pos.size() < 2) die("Usage: carpedeam ancient_reads_loop <i:sequenceDB> <o:sequenceDB> --ancient-damage <prefix> [--num-iter-reads-only N]");
    // the contig iterations' buffers grow ~1.5x per iteration: head room in the device-memory cache lets them fit the previous iteration's blocks
    cdm_pool_headroom(getenv("CDM_POOL_HEADROOM") ? (float) atof(getenv("CDM_POOL_HEADROOM")) : 1.6f);
    {   // the workflow's own flags for the reads loop (src/commons/LocalParameters.h:283-318) on top of the stage lists
        static const char *const LOOP_FLAGS[] = {"--k-ancient-reads", "--kmer-per-seq-ancient", "--kmer-per-seq-scale-ancient", "--hash-shift", "--include-only-extendable-ancient-reads",
                                                 "-e", "--num-iter-reads-only", "--shuffle", "--num-iterations", "--k-ancient-contigs", "--include-only-extendable-ancient-contigs",
                                                 "--cycle-check", "--chop-cycle", "--gpus", NULL};
        checkFlags("ancient_reads_loop", a, ANCIENT_FLAGS, LOOP_FLAGS);
    }
    // input: a sequence DB, or - when there is no <input>.index - FASTA/FASTQ[.gz] reads, parsed and laid out as createdb would
    // (createdb.cpp:150-280, --shuffle 1 by default) and uploaded without a DB on disk in between
    MmDb seq; std::string err;
    int dbtype = 1;
    cdm_ctx *ctx = NULL; cdm_seqdb *db = NULL;
    struct stat st;
    Laps laps;
    FastxDb fx;
    const bool fromDb = stat((a.pos[0] + ".index").c_str(), &st) == 0;
    auto uploadTo = [&](cdm_ctx *c) -> cdm_seqdb * {       // (the same host copy serves every rank of a --gpus N run)
        if (fromDb) return uploadSeqDb(c, seq);
        cdm_seqdb *d = NULL;
        check(cdm_seqdb_upload(c, fx.blob.data(), fx.off.data(), fx.len.data(), fx.key.data(), NULL, fx.key.size(), &d), "Can not load the reads");
        return d;
    };
    if (fromDb) {
        if (!seq.load(a.pos[0], &err)) die(err);
        dbtype = seq.dbtype;
        laps.lap("DB files mapped");
    } else {
        if (!readFastxAsDb(std::vector<std::string>(1, a.pos[0]), iflag(a, "--shuffle", 1) != 0, fx, &err)) die(err);
        for (auto &l : fx.len) l -= 2;
        laps.lap("reads file parsed");
    }
    ctx = openCtx(); laps.lap("device context");
    db = uploadTo(ctx); laps.lap("sequences up");
    check(cdm_damage_load(ctx, a.flag.count("--ancient-damage") ? a.flag["--ancient-damage"].c_str() : ""), "Profile not 12 fields");
    cdm_kmer_params kp;
    kp.kmer_size = (int) iflag(a, "--k-ancient-reads", 20); kp.kmers_per_seq = (int) iflag(a, "--kmer-per-seq-ancient", 200);
    kp.kmers_per_seq_scale = fflag(a, "--kmer-per-seq-scale-ancient", 0.2f); kp.hash_shift = (uint64_t) iflag(a, "--hash-shift", 67);
    kp.ignore_multi_kmer = 1; kp.include_only_extendable = (int) iflag(a, "--include-only-extendable-ancient-reads", 0); kp.cov_mode = 1; kp.cov_thr = 0.0f;
    cdm_rescore_params rp;
    rp.seq_id_thr = fflag(a, "--min-seq-id", 0.9f); rp.eval_thr = a.flag.count("-e") ? strtod(a.flag["-e"].c_str(), NULL) : 0.001;
    rp.cov_mode = 1; rp.cov_thr = 0.0f; rp.seq_id_mode = 0; rp.min_aln_len = 0;
    cdm_ancient_params ap = ancientParams(a);
    if (!a.flag.count("--max-seq-len")) ap.max_seq_len = 200000;   // setGuidedNuclAssemblerWorkflowDefaults (GuidedNuclassembler.cpp:29)
    const long iters = iflag(a, "--num-iter-reads-only", 5);       // LocalParameters.h:304
    // --num-iterations N > --num-iter-reads-only M: the contig phase of the workflow loop (data/nuclassemble.sh:148-196) for the
    // remaining N - M iterations - kmermatcher with the contig parameters (Nuclassembler.cpp:118-126), rescorediagonal,
    // ancient_correction, ancient_contig_merge, and the script's cyclecheck() (:19-60; --cycle-check / --chop-cycle, both on by default
    // as setNuclAssemblerWorkflowDefaults has them): circular contigs leave the loop, cut at their split diagonal, and join the output
    // at the end (concatdbs, :204-212)
    const long total = std::max(iters, iflag(a, "--num-iterations", iters));
    cdm_kmer_params kc = kp;
    kc.kmer_size = (int) iflag(a, "--k-ancient-contigs", 22); kc.include_only_extendable = (int) iflag(a, "--include-only-extendable-ancient-contigs", 1);
    const float mergeThr = fflag(a, "--min-merge-seq-id", 0.99f);
    // the workflow builds the contig phase's rescorediagonal / assembler parameter strings after `par.seqIdThr = par.corrContigSeqId`
    // (Nuclassembler.cpp:124-126): there --min-seq-id IS --min-seqid-corr-contigs (LocalParameters.h default 0.9)
    cdm_rescore_params rc = rp; cdm_ancient_params ac = ap;
    rc.seq_id_thr = ac.seq_id_thr = fflag(a, "--min-seqid-corr-contigs", 0.9f);
    const bool cycleCheck = iflag(a, "--cycle-check", 1) != 0, chopCycle = iflag(a, "--chop-cycle", 1) != 0;
    OutChunk cyclic;
    // --gpus N: the read iterations over N ranks = N host threads of this process, a device each; every rank holds the whole DB and
    // the library splits the work (cdm_reads_iteration_dist: kmermatcher by k-mer range, one all-to-all of group keys, the other stages
    // on the owned queries, the new DBs all-gathered - the single-device result), the contig iterations likewise since round 5
    // (cdm_contig_iteration_dist: ancient_contig_merge's queue runs on the device, on the owned queries), the script's cyclecheck step on
    // every rank.  This thread is rank 0.
    const int gpus = (int) std::max<long>(1, iflag(a, "--gpus", 1));
    long firstLocal = 0;
    if (gpus > 1 || getenv("CDM_LOOP_FORCE_COMM")) {
        const char *tr = getenv("CDM_LOOP_TRANSPORT");
        const bool threadsTransport = tr && !strcmp(tr, "threads");
        // CDM_LOOP_TRANSPORT=standin (tests): the library's RCCL transport over its in-process stand-in for RCCL's calls - the ranks share
        // device 0, as with "threads", but what runs is the transport a deployment runs
        void *standin = (tr && !strcmp(tr, "standin")) ? cdm_comm_standin_group(gpus) : NULL;
        const bool oneDevice = threadsTransport || standin;
        unsigned char uid[128];
        if (!oneDevice) check(cdm_comm_unique_id(uid), "RCCL");
        ThreadTransport tt(gpus);
        const std::string damage = a.flag.count("--ancient-damage") ? a.flag["--ancient-damage"] : "";
        auto rankBody = [&](int r, cdm_ctx *c, cdm_seqdb *&d) {
            ThreadRank me{&tt, r, c};
            cdm_comm *cm = NULL;
            if (threadsTransport) { cdm_comm_ops ops{&me, ttAllGatherHost, ttAllToAllDev, ttAllGatherDev}; check(cdm_comm_create_ops(c, r, gpus, &ops, &cm), "communicator"); }
            else if (standin) check(cdm_comm_create_standin(c, standin, r, &cm), "communicator");
            else check(cdm_comm_create_rccl(c, r, gpus, uid, &cm), "RCCL communicator");
            for (long it = 0; it < total && cdm_seqdb_size(d) > 0; it++) {
                cdm_alns *alns = NULL; cdm_seqdb *next = NULL;
                const bool contigs = it >= iters;
                const auto tIt = std::chrono::steady_clock::now();
                if (contigs) check(cdm_contig_iteration_dist(c, cm, d, &kc, &rc, &ac, mergeThr, &alns, NULL, &next), "contig iteration over the ranks");
                else check(cdm_reads_iteration_dist(c, cm, d, &kp, &rp, &ap, NULL, &alns, NULL, &next), "reads iteration over the ranks");
                if (r == 0) fprintf(stderr, "STEP: %ld  sequences %llu  residues %llu -> %llu  alignments of rank 0's queries %llu  (%.3f s on %d ranks%s)\n",
                                    it, (unsigned long long) cdm_seqdb_size(d), (unsigned long long) cdm_seqdb_residues(d), (unsigned long long) cdm_seqdb_residues(next),
                                    (unsigned long long) cdm_alns_count(alns), std::chrono::duration<double>(std::chrono::steady_clock::now() - tIt).count(), gpus,
                                    threadsTransport ? ", in-process transport on one device" : standin ? ", the RCCL transport over its in-process stand-in, one device" : ", RCCL");
                cdm_alns_free(alns); cdm_seqdb_free(d);
                d = next;
                if (contigs && cycleCheck) {        // every rank holds the whole merged DB: the same check, the same rest on all of them; rank 0 keeps the circular contigs
                    cdm_seqdb *cyc = NULL, *rest = NULL;
                    check(cdm_cyclecheck(c, d, (uint32_t) std::min<uint64_t>(ap.max_seq_len, 0xFFFFFFFFull), chopCycle, &cyc, &rest, NULL), "cyclecheck");
                    if (r == 0) {
                        if (cdm_seqdb_size(cyc)) fprintf(stderr, "         %llu circular contigs set aside\n", (unsigned long long) cdm_seqdb_size(cyc));
                        appendEntries(c, cyc, cyclic);
                    }
                    cdm_seqdb_free(cyc); cdm_seqdb_free(d);
                    d = rest;
                }
            }
            cdm_comm_free(cm);
        };
        std::vector<std::thread> helpers;
        for (int r = 1; r < gpus; r++) helpers.emplace_back([&, r] {
            cdm_ctx *c = openCtx(oneDevice ? 0 : r);
            check(cdm_damage_load(c, damage.c_str()), "Profile not 12 fields");
            cdm_seqdb *d = uploadTo(c);
            rankBody(r, c, d);
            cdm_seqdb_free(d); cdm_ctx_destroy(c);
        });
        rankBody(0, ctx, db);
        for (auto &h : helpers) h.join();
        firstLocal = total;
    }
    for (long it = firstLocal; it < total && cdm_seqdb_size(db) > 0; it++) {
        cdm_hits *hits = NULL; cdm_alns *alns = NULL; cdm_seqdb *corr = NULL, *next = NULL;
        const bool contigs = it >= iters;
        const auto tIt = std::chrono::steady_clock::now();
        check(cdm_kmermatch(ctx, db, contigs ? &kc : &kp, &hits), "kmermatcher");
        check(cdm_rescore(ctx, db, hits, contigs ? &rc : &rp, &alns), "rescorediagonal");
        cdm_hits_free(hits);
        check(cdm_correct(ctx, db, alns, contigs ? &ac : &ap, &corr), "ancient_correction");
        if (contigs) check(cdm_contig_merge(ctx, corr, alns, &ac, mergeThr, &next), "ancient_contig_merge");
        else check(cdm_extend(ctx, corr, alns, &ap, &next, NULL), "ancient_read_assemble");
        fprintf(stderr, "STEP: %ld  sequences %llu  residues %llu -> %llu  alignments %llu  (%.3f s; device stages: kmermatcher %.0f, rescorediagonal %.0f, ancient_correction %.0f, %s %.0f ms)\n",
                it, (unsigned long long) cdm_seqdb_size(db), (unsigned long long) cdm_seqdb_residues(db), (unsigned long long) cdm_seqdb_residues(next), (unsigned long long) cdm_alns_count(alns),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - tIt).count(), cdm_ctx_last_kernel_ms(ctx, 8), cdm_ctx_last_kernel_ms(ctx, 9), cdm_ctx_last_kernel_ms(ctx, 10),
                contigs ? "contig statistics" : "ancient_read_assemble", cdm_ctx_last_kernel_ms(ctx, contigs ? 12 : 11));
        cdm_alns_free(alns); cdm_seqdb_free(corr); cdm_seqdb_free(db);
        db = next;
        if (contigs && cycleCheck) {
            cdm_seqdb *cyc = NULL, *rest = NULL;
            check(cdm_cyclecheck(ctx, db, (uint32_t) std::min<uint64_t>(ap.max_seq_len, 0xFFFFFFFFull), chopCycle, &cyc, &rest, NULL), "cyclecheck");
            if (cdm_seqdb_size(cyc)) fprintf(stderr, "         %llu circular contigs set aside\n", (unsigned long long) cdm_seqdb_size(cyc));
            appendEntries(ctx, cyc, cyclic);
            cdm_seqdb_free(cyc); cdm_seqdb_free(db);
            db = rest;
        }
    }
    laps.lap("iterations");
    if (cyclic.key.empty()) { writeSeqDb(ctx, db, a.pos[1], dbtype); laps.lap("sequences down, DB written"); }
    else {   // concatdbs --preserve-keys of the linear and the circular contigs, written in key order
        OutChunk all; appendEntries(ctx, db, all);
        std::vector<std::pair<uint32_t, std::pair<const OutChunk *, size_t>>> order;
        std::vector<size_t> offA(all.key.size() + 1, 0), offC(cyclic.key.size() + 1, 0);
        for (size_t i = 0; i < all.key.size(); i++) { offA[i + 1] = offA[i] + all.len[i]; order.push_back({all.key[i], {&all, i}}); }
        for (size_t i = 0; i < cyclic.key.size(); i++) { offC[i + 1] = offC[i] + cyclic.len[i]; order.push_back({cyclic.key[i], {&cyclic, i}}); }
        std::stable_sort(order.begin(), order.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
        std::vector<OutChunk> merged(1);
        for (const auto &o : order) {
            const OutChunk &c = *o.second.first; const size_t i = o.second.second;
            merged[0].add(o.first, c.data.data() + (&c == &all ? offA[i] : offC[i]), c.len[i] - 1, c.ext[i]);
        }
        if (!mmdbWriteChunks(a.pos[1], dbtype, merged, &err)) die(err);
    }
    cdm_seqdb_free(db); cdm_ctx_destroy(ctx);
    return EXIT_SUCCESS;
}
}  // namespace

namespace {
const FlagSpec CREATEDB_FLAGS[] = {   // Parameters.cpp:733-737 createdb
    {"--shuffle", 'U', 0, 0}, {"--dbtype", 'U', 0, 0}, {"--createdb-mode", 'V', "0", "only the copying mode"},
    {"--write-lookup", 'V', "1", "the .lookup file is always written"}, {"--id-offset", 'V', "0", "not implemented"}, {"--compressed", 'V', "0", "compressed DBs are not implemented"},
    {"-v", 'N', 0, 0}, {"--threads", 'N', 0, 0}, {0, 0, 0, 0}};
const FlagSpec PLAIN_FLAGS[] = {{"-v", 'N', 0, 0}, {"--threads", 'N', 0, 0}, {"--compressed", 'V', "0", "compressed DBs are not implemented"}, {"--use-fasta-header", 'V', "0", "not implemented"}, {0, 0, 0, 0}};
// The side-car of a sequence DB straight from its text, on the host (createdb has no device): the letters packed as cdm_seqdb_upload
// packs them (api.hip k_pack: A, C, G, T = 0..3, 16 per word, every sequence on a word boundary; 'N' = code 0 + a bit of the N mask).
// Only for DBs of upper-case ACGTN - any other letter takes the device's mapping and its raw plane (the first module that uploads the
// DB then writes the side-car).
void seqSideFromText(const std::string &path) {
    if (!sideEnabled()) return;
    MmDb db; std::string err;
    if (!db.load(path, &err) || (db.dbtype & 0x7FFFFFFF) != 1 || db.size() == 0) return;
    const size_t n = db.size();
    HVec<uint32_t> lens(n), woff(n + 1); HVec<uint8_t> ext(n), flags(n);
    woff[0] = 0;
    for (size_t i = 0; i < n; i++) { lens[i] = db.len[i] >= 2 ? (uint32_t) (db.len[i] - 2) : 0; ext[i] = db.ext[i]; const uint64_t w = (uint64_t) woff[i] + (lens[i] + 15) / 16; if (w > 0xFFFFFFFFull) return; woff[i + 1] = (uint32_t) w; }
    const uint64_t words = woff[n];
    HVec<uint32_t> codes(words + 1); HVec<uint16_t> mask(words + 1);
    bool other = false, anyN = false;
#pragma omp parallel for reduction(|| : other, anyN) schedule(dynamic, 4096)
    for (size_t i = 0; i < n; i++) {
        const char *sq = db.entry(i);
        const uint32_t L = lens[i];
        uint8_t fl = 0;
        for (uint32_t w = 0; w * 16 < L; w++) {
            uint32_t code = 0, nb = 0;
            const uint32_t cnt = std::min(16u, L - w * 16);
            for (uint32_t j = 0; j < cnt; j++) {
                uint32_t v = 0;
                switch (sq[w * 16 + j]) { case 'A': v = 0; break; case 'C': v = 1; break; case 'G': v = 2; break; case 'T': v = 3; break; case 'N': nb |= 1u << j; break; default: other = true; }
                code |= v << (2 * j);
            }
            codes[woff[i] + w] = code; mask[woff[i] + w] = (uint16_t) nb;
            if (nb) fl = 1;
        }
        flags[i] = fl; anyN = anyN || fl;
    }
    if (other) return;
    const SidePiece pc[7] = {{db.key.data(), n * 4}, {lens.data(), n * 4}, {ext.data(), n}, {flags.data(), n}, {codes.data(), words * 4}, {mask.data(), anyN ? words * 2 : 0}, {NULL, 0}};
    sideWrite(path, SIDE_SEQ, anyN ? SIDE_F_HAS_NMASK : 0, n, words, 0, 0, db.dbtype, pc, 7);
}
int createdb(Args &a) {
    if (a.pos.size() < 2) die("Usage: carpedeam createdb <i:fastaFile1[.gz]> ... <i:fastaFileN[.gz]> <o:sequenceDB>");
    checkFlags("createdb", a, CREATEDB_FLAGS);
    std::vector<std::string> files(a.pos.begin(), a.pos.end() - 1);
    std::string err;
    const long dbType = iflag(a, "--dbtype", 0);        // 0 = guess from the first entries, as createdb does (createdb.cpp:40-45)
    if (dbType != 0 && dbType != 2) unsupported("createdb: --dbtype " + std::to_string(dbType) + " is not supported by the MI355X path (nucleotide sequences only; accepted: 0, 2)");
    if (createdbModule(files, a.pos.back(), iflag(a, "--shuffle", 1) != 0, (int) dbType, &err)) die(err);
    seqSideFromText(a.pos.back());
    return EXIT_SUCCESS;
}
int convert2fasta(Args &a) {
    if (a.pos.size() < 2) die("Usage: carpedeam convert2fasta <i:sequenceDB> <o:fastaFile>");
    checkFlags("convert2fasta", a, PLAIN_FLAGS);
    std::string err;
    if (convert2fastaModule(a.pos[0], a.pos[1], &err)) die(err);
    return EXIT_SUCCESS;
}
int createhdb(Args &a) {
    if (a.pos.size() < 2) die("Usage: carpedeam createhdb <i:sequenceDB> [<i:sequenceDBcycle>] <o:headerDB>");
    checkFlags("createhdb", a, PLAIN_FLAGS);
    std::string err;
    if (createhdbModule(a.pos[0], a.pos.size() > 2 ? a.pos[1] : "", a.pos.back(), &err)) die(err);
    return EXIT_SUCCESS;
}
}  // namespace

// see reportDone(): true in the process that goes on with the work
static bool workInChild() {
    if (!leaveAtOnce() || getenv("CDM_NO_FORK")) return true;
    int fds[2];
    if (pipe(fds) != 0) return true;
    fflush(stdout); fflush(stderr);
    const pid_t pid = fork();
    if (pid < 0) { close(fds[0]); close(fds[1]); return true; }
    if (pid == 0) { close(fds[0]); g_doneFd = fds[1]; prctl(PR_SET_PDEATHSIG, SIGKILL); if (getppid() == 1) _exit(EXIT_FAILURE); return true; }
    close(fds[1]);
    unsigned char b = 0; ssize_t r;
    while ((r = read(fds[0], &b, 1)) < 0 && errno == EINTR) {}
    if (r == 1) _exit((int) b);
    int st = 0;
    while (waitpid(pid, &st, 0) < 0) if (errno != EINTR) _exit(EXIT_FAILURE);
    if (WIFSIGNALED(st)) { signal(WTERMSIG(st), SIG_DFL); raise(WTERMSIG(st)); _exit(128 + WTERMSIG(st)); }
    _exit(WEXITSTATUS(st));
}
int main(int argc, char **argv) {
    workInChild();
    if (argc < 2) { fprintf(stderr, "usage: carpedeam <kmermatcher|rescorediagonal|ancient_correction|ancient_read_assemble|ancient_contig_merge|cyclecheck|ancient_reads_loop|createdb|convert2fasta|createhdb|clust|createsubdb|filterdb|mergeclusters|result2repseq|rmdb|mvdb> <args>\n"); return EXIT_FAILURE; }
    const std::string cmd = argv[1];
    Args a = parse(argc - 2, argv + 2);
    {   // --threads / MMSEQS_NUM_THREADS as in Parameters.cpp:2121-2132: the host side (DB parsing, text codecs) uses them
        long th = a.flag.count("--threads") ? strtol(a.flag["--threads"].c_str(), NULL, 10) : 0;
        if (th <= 0) if (const char *e = getenv("MMSEQS_NUM_THREADS")) th = strtol(e, NULL, 10);
        omp_set_dynamic(0);      // every slice of the per-thread loops below has its thread
        if (th > 0) { omp_set_num_threads((int) th); setenv("OMP_NUM_THREADS", std::to_string(th).c_str(), 1); }      // (the library's own loops read it)
    }
    auto t0 = std::chrono::steady_clock::now(); g_t0 = t0;
    int rc;
    if (cmd == "kmermatcher") rc = kmermatcher(a);
    else if (cmd == "rescorediagonal") rc = rescorediagonal(a);
    else if (cmd == "ancient_correction") rc = ancientModule(a, 0);
    else if (cmd == "ancient_read_assemble") rc = ancientModule(a, 1);
    else if (cmd == "ancient_contig_merge") rc = ancientModule(a, 2);
    else if (cmd == "ancient_reads_loop") rc = readsLoop(a);
    else if (cmd == "align" || cmd == "clust" || cmd == "createsubdb" || cmd == "filterdb" || cmd == "mergeclusters" || cmd == "result2repseq" || cmd == "rmdb" || cmd == "mvdb") rc = clusterModules(cmd, a);
    else if (cmd == "createdb") rc = createdb(a);
    else if (cmd == "convert2fasta") rc = convert2fasta(a);
    else if (cmd == "createhdb") rc = createhdb(a);
    else if (cmd == "cyclecheck") rc = cyclecheck(a);
    else { fprintf(stderr, "Invalid Command: %s\n", cmd.c_str()); return EXIT_FAILURE; }
    fprintf(stderr, "Time for processing: %.3fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    // Every output file is written and closed, every handle and the context released: leave without the static destructors (the
    // HIP runtime's own tear-down is where one module run in ~1 500 of the fuzz campaigns ended with a signal after its work was done)
    // (not under a profiler that writes its results from an exit handler: ROCP_TOOL_LIBRARIES / a profiler's LD_PRELOAD / CDM_NORMAL_EXIT)
    if (!leaveAtOnce()) return rc;
    fflush(stdout); fflush(stderr);
    reportDone(rc);
    _exit(rc);
}
