// Multi-GPU inside the library: one rank per device (a process, or a host thread of one process), RCCL over xGMI called directly.
//
// The reference shards INSIDE its modules (kmermatcher's k-mer split lib/mmseqs/src/linclust/kmermatcher.cpp:634-663 + merge :742-784,
// rescorediagonal's query split lib/mmseqs/src/alignment/rescorediagonal.cpp:399-421); so does this file, with the exact scheme of
// DESIGN.md section 6 - bit-identical to the single-device run:
//   cdm_kmermatch_dist         rank r runs kmermatcher's first half on the r-th k-mer range (cdm_kmermatch_part), ONE all-to-all moves
//                              every group key to the owner of its representative, the second half (sort 2 + vote) runs there; three
//                              small all-gathers carry what the reference's quirks need across ranks (the left-over list of the
//                              run-past-the-end scan :875-887, the heads of the sorted arrays, the tuple counts)
//   cdm_seqdb_allgather_owned  the stages behind it work on the owned queries; the owned ranges of their result DBs are all-gathered
//   cdm_reads_iteration_dist   one iteration of the reads loop (data/nuclassemble.sh:100-146) that way
// Transports: RCCL (librccl, loaded on first use - it is half a gigabyte, and a one-device module must not pay for it), or a table
// of functions the caller supplies (cdm_comm_create_ops: the tests run W ranks on ONE device through it, where RCCL wants a device per
// rank; another collective library could be bound the same way).  carpedeam_amd/shard.py holds the same calling sequence in Python;
// it stays as the reference the tests compare this file with.
#include <chrono>
#include <condition_variable>
#include <dlfcn.h>
#include <mutex>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "devutil.h"

// ------------------------------------------------------------------------------------------------ RCCL, bound at run time
namespace {
struct Rccl {
    typedef int (*GetUniqueId)(void *);
    typedef int (*CommInitRank)(void **, int, /* ncclUniqueId by value: 128 bytes */ struct Id128, int);
    void *lib = nullptr;
    int (*getUniqueId)(void *) = nullptr;
    void *commInitRank = nullptr;
    int (*commDestroy)(void *) = nullptr;
    int (*allGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*groupStart)() = nullptr;
    int (*groupEnd)() = nullptr;
    const char *(*errorString)(int) = nullptr;
};
struct Id128 { char b[128]; };      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed by value as the API does
constexpr int NCCL_CHAR = 0;        // ncclInt8 / ncclChar
constexpr uint64_t RCCL_PIECE = 256ull << 20;      // one send / recv moves at most this many bytes
Rccl *rccl(std::string *err) {
    static Rccl *r = nullptr; static std::string why; static std::once_flag once;
    std::call_once(once, [] {
        Rccl *t = new Rccl();
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { t->lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (t->lib) break; }
        if (!t->lib) { why = std::string("librccl.so not found: ") + dlerror(); return; }
        auto sym = [&](const char *n) { void *p = dlsym(t->lib, n); if (!p && why.empty()) why = std::string("librccl has no ") + n; return p; };
        t->getUniqueId = (int (*)(void *)) sym("ncclGetUniqueId"); t->commInitRank = sym("ncclCommInitRank"); t->commDestroy = (int (*)(void *)) sym("ncclCommDestroy");
        t->allGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t)) sym("ncclAllGather");
        t->send = (int (*)(const void *, size_t, int, int, void *, hipStream_t)) sym("ncclSend");
        t->recv = (int (*)(void *, size_t, int, int, void *, hipStream_t)) sym("ncclRecv");
        t->groupStart = (int (*)()) sym("ncclGroupStart"); t->groupEnd = (int (*)()) sym("ncclGroupEnd");
        t->errorString = (const char *(*)(int)) sym("ncclGetErrorString");
        if (why.empty()) r = t;
    });
    if (!r && err) *err = why;
    return r;
}
}  // namespace

struct cdm_comm {
    cdm_ctx *ctx = nullptr; int rank = 0, world = 1;
    cdm_comm_ops ops;                   // the transport (RCCL's functions below, or the caller's)
    void *nccl = nullptr;               // ncclComm_t of the RCCL transport (or a rank of the stand-in below)
    Rccl *api = nullptr;         // RCCL's entry points - or the in-process stand-in's (tests: the transport's own code with several ranks on ONE device)
    // What RCCL reads and writes are buffers of THIS transport, from hipMalloc: the library's own buffers live in arenas of mapped
    // memory (csrc/pool.h: hipMemCreate / hipMemMap, access granted to the owning device only), and a peer device or another
    // process must never be pointed at those - RCCL may hand a user buffer's address to the peer (ranks as threads of one process).
    // A copy on either side of a transfer costs 2 x bytes / 5 TB/s against bytes / 0.15 TB/s for the link.  CDM_RCCL_DIRECT=1: RCCL
    // on the caller's buffers (A/B).
    struct Stage { void *p = nullptr; size_t bytes = 0; } stage, sendStage, recvStage;
    std::vector<uint64_t> own; uint64_t ownN = 0;       // the owners' id ranges as the last cdm_kmermatch_dist cut them (for DBs of ownN sequences)
    bool knowsFailure = false;          // this call: a rank's failure has been agreed on (or announced by this rank) - no further collective is entered
    int lastPath = 0;                   // what the last cdm_kmermatch_dist did: 1 every rank ran kmermatcher whole, 2 every rank extracted all reads and kept its k-mer range, 3 the reads were split and the tuples travelled, 4 equal k-mer slices by value (cdm_kmermatch_part), 5 the first half by ranges of the k-mer space, the kept group keys all-gathered, the second half on every rank
};
namespace {
int ensureStage(cdm_comm *c, cdm_comm::Stage &st, size_t need) {
    if (need <= st.bytes) return CDM_OK;
    CDM_HIP(hipStreamSynchronize(c->ctx->stream));
    if (st.p) (void) hipFree(st.p);
    st.p = nullptr; st.bytes = 0;
    const size_t want = need + need / 4 + 256;
    if (hipMalloc(&st.p, want) != hipSuccess && hipMalloc(&st.p, need + 256) != hipSuccess && (cdmPoolTrim(), hipMalloc(&st.p, need + 256)) != hipSuccess) {     // (the arenas give back what is free)
        (void) hipGetLastError(); st.p = nullptr; cdm_set_error("RCCL transport: out of device memory for %zu bytes of exchange buffer", need); return CDM_ERR_HIP; }
    st.bytes = need;        // (at least)
    return CDM_OK;
}
bool rcclDirect() { const char *e = cdmGetenv("CDM_RCCL_DIRECT"); return e && *e == '1'; }
}  // namespace
#define CDM_NCCL(call, what) do { const int e_ = (call); if (e_ != 0) { cdm_set_error("RCCL %s failed: %s", what, r->errorString ? r->errorString(e_) : "?"); return CDM_ERR_HIP; } } while (0)

// ---- a failure on ONE rank must not leave the others inside a collective (out of memory for an exchange buffer, a refusal that depends
// on a rank's data): every collective of the calling sequences below is entered through co*(), which first all-gathers a status word;
// a rank that failed announces its code once (Sequence's end) - its peers meet that in their next co*() or at their own end - and
// every rank returns an error from the same call.
namespace {
int agree(cdm_comm *cm, int mine, const char *where) {
    if (cm->world == 1) return mine;
    std::vector<int32_t> all((size_t) cm->world, 0);
    const int32_t m = mine;
    if (int rc = cm->ops.all_gather_host(cm->ops.user, &m, all.data(), 4)) { cm->knowsFailure = true; return rc; }
    for (int p = 0; p < cm->world; p++) if (all[p] != 0) {
        cm->knowsFailure = true;
        if (mine == CDM_OK) cdm_set_error("%s: rank %d failed with error %d (its message is that rank's cdm_last_error)", where, p, (int) all[p]);
        return mine != CDM_OK ? mine : (int) all[p];
    }
    return CDM_OK;
}
// (the RCCL transport's own exchange buffers are taken BEFORE the status goes round: an allocation that fails inside the transport, behind
// the agreement, would strand the peers in ncclGroupEnd all the same)
int prepareStages(cdm_comm *cm, uint64_t sendBytes, uint64_t recvBytes) {
    if (!cm->nccl || cm->world == 1 || rcclDirect()) return CDM_OK;
    if (int rc = ensureStage(cm, cm->sendStage, sendBytes)) return rc;
    return ensureStage(cm, cm->recvStage, recvBytes);
}
int coAllGatherHost(cdm_comm *cm, const void *send, void *recv, uint64_t bytes) {
    const int mine = (cm->nccl && cm->world > 1) ? ensureStage(cm, cm->stage, (size_t) bytes * (size_t) (cm->world + 1)) : CDM_OK;
    if (int rc = agree(cm, mine, "a collective of the multi-GPU calls")) return rc;
    return cm->ops.all_gather_host(cm->ops.user, send, recv, bytes);
}
int coAllToAllDev(cdm_comm *cm, const void *send, const uint64_t *so, void *recv, const uint64_t *ro, void *stream) {
    if (int rc = agree(cm, prepareStages(cm, so[cm->world] - so[0], ro[cm->world] - ro[0]), "a collective of the multi-GPU calls")) return rc;
    return cm->ops.all_to_all_dev(cm->ops.user, send, so, recv, ro, stream);
}
int coAllGatherDev(cdm_comm *cm, const void *send, uint64_t bytes, void *recv, const uint64_t *ro, void *stream) {
    if (int rc = agree(cm, prepareStages(cm, bytes, ro[cm->world] - ro[0]), "a collective of the multi-GPU calls")) return rc;
    return cm->ops.all_gather_dev(cm->ops.user, send, bytes, recv, ro, stream);
}
// the end of a calling sequence: a rank that failed on its own tells the others; the others learn it here at the latest
int finish(cdm_comm *cm, int rc, const char *where) {
    if (cm->world == 1 || cm->knowsFailure) return rc;
    return agree(cm, rc, where);
}
}  // namespace

// ---- RCCL transport
void standinEnter(cdm_comm *c);      // (the in-process stand-in's group calls need to know the calling rank: below)
namespace {
int rcclAllGatherHost(void *user, const void *send, void *recv, uint64_t bytes) {
    cdm_comm *c = (cdm_comm *) user; Rccl *r = c->api;
    const size_t need = (size_t) bytes * (size_t) (c->world + 1);
    if (int rc = ensureStage(c, c->stage, need)) return rc;
    char *dSend = (char *) c->stage.p, *dRecv = dSend + bytes;
    hipStream_t s = c->ctx->stream;
    CDM_HIP(hipMemcpyAsync(dSend, send, bytes, hipMemcpyHostToDevice, s));
    CDM_NCCL(r->allGather(dSend, dRecv, bytes, NCCL_CHAR, c->nccl, s), "all-gather");
    CDM_HIP(hipMemcpyAsync(recv, dRecv, bytes * (size_t) c->world, hipMemcpyDeviceToHost, s));
    CDM_HIP(hipStreamSynchronize(s));
    return CDM_OK;
}
// peer p gets send[sendOff[p], sendOff[p + 1]) and fills recv[recvOff[p], recvOff[p + 1]): grouped point-to-point transfers - over xGMI
// every pair of devices has its own link, so the W - 1 transfers of a rank run side by side
int rcclAllToAllDev(void *user, const void *send, const uint64_t *sendOff, void *recv, const uint64_t *recvOff, void *stream) {
    cdm_comm *c = (cdm_comm *) user; Rccl *r = c->api;
    standinEnter(c);
    hipStream_t s = (hipStream_t) stream;
    const int W = c->world;
    const bool direct = rcclDirect();
    const char *src = (const char *) send; char *dst = (char *) recv;
    // A rank's own share is a device copy, not a transfer (RCCL's send-to-self of 1.28 GB took 1.2 s and delivered part of it:
    // scripts/probes/dist1.py at 2 M reads) - straight from the caller's send buffer to its receive buffer; what goes to a peer goes in
    // pieces of at most RCCL_PIECE bytes, a send / recv pair each, through the transport's own buffers.
    const uint64_t selfBytes = sendOff[c->rank + 1] - sendOff[c->rank];
    if (selfBytes != recvOff[c->rank + 1] - recvOff[c->rank]) { cdm_set_error("all-to-all: rank %d sends itself %llu bytes and expects %llu", c->rank, (unsigned long long) selfBytes, (unsigned long long) (recvOff[c->rank + 1] - recvOff[c->rank])); return CDM_ERR_INVALID; }
    if (selfBytes) CDM_HIP(hipMemcpyAsync((char *) recv + recvOff[c->rank], (const char *) send + sendOff[c->rank], selfBytes, hipMemcpyDeviceToDevice, s));
    if (W > 1 && !direct) {
        if (int rc = ensureStage(c, c->sendStage, sendOff[W] - sendOff[0])) return rc;
        if (int rc = ensureStage(c, c->recvStage, recvOff[W] - recvOff[0])) return rc;
        for (int p = 0; p < W; p++) if (p != c->rank && sendOff[p + 1] > sendOff[p])
            CDM_HIP(hipMemcpyAsync((char *) c->sendStage.p + (sendOff[p] - sendOff[0]), src + sendOff[p], sendOff[p + 1] - sendOff[p], hipMemcpyDeviceToDevice, s));
        src = (const char *) c->sendStage.p - sendOff[0]; dst = (char *) c->recvStage.p - recvOff[0];
    }
    if (W > 1) {
        CDM_NCCL(r->groupStart(), "group start");
        for (int p = 0; p < W; p++) {
            if (p == c->rank) continue;
            const uint64_t ns = sendOff[p + 1] - sendOff[p], nr = recvOff[p + 1] - recvOff[p];
            for (uint64_t o = 0; o < ns; o += RCCL_PIECE) CDM_NCCL(r->send(src + sendOff[p] + o, std::min(RCCL_PIECE, ns - o), NCCL_CHAR, p, c->nccl, s), "send");
            for (uint64_t o = 0; o < nr; o += RCCL_PIECE) CDM_NCCL(r->recv(dst + recvOff[p] + o, std::min(RCCL_PIECE, nr - o), NCCL_CHAR, p, c->nccl, s), "recv");
        }
        CDM_NCCL(r->groupEnd(), "group end");
    }
    if (W > 1 && !direct) for (int p = 0; p < W; p++) if (p != c->rank && recvOff[p + 1] > recvOff[p])
        CDM_HIP(hipMemcpyAsync((char *) recv + recvOff[p], (const char *) c->recvStage.p + (recvOff[p] - recvOff[0]), recvOff[p + 1] - recvOff[p], hipMemcpyDeviceToDevice, s));
    return CDM_OK;
}
// every rank contributes sendBytes (they differ): recv[recvOff[p], recvOff[p + 1]) = rank p's
int rcclAllGatherDev(void *user, const void *send, uint64_t sendBytes, void *recv, const uint64_t *recvOff, void *stream) {
    cdm_comm *c = (cdm_comm *) user; Rccl *r = c->api;
    standinEnter(c);
    hipStream_t s = (hipStream_t) stream;
    const int W = c->world;
    const bool direct = rcclDirect();
    const char *src = (const char *) send; char *dst = (char *) recv;
    {
        const uint64_t nr = recvOff[c->rank + 1] - recvOff[c->rank];
        if (nr != sendBytes) { cdm_set_error("all-gather: rank %d contributes %llu bytes and expects %llu of itself", c->rank, (unsigned long long) sendBytes, (unsigned long long) nr); return CDM_ERR_INVALID; }
        if (nr) CDM_HIP(hipMemcpyAsync((char *) recv + recvOff[c->rank], send, nr, hipMemcpyDeviceToDevice, s));
    }
    if (W > 1 && !direct) {
        if (int rc = ensureStage(c, c->sendStage, sendBytes)) return rc;
        if (int rc = ensureStage(c, c->recvStage, recvOff[W] - recvOff[0])) return rc;
        if (sendBytes) CDM_HIP(hipMemcpyAsync(c->sendStage.p, send, sendBytes, hipMemcpyDeviceToDevice, s));
        src = (const char *) c->sendStage.p; dst = (char *) c->recvStage.p - recvOff[0];
    }
    if (W > 1) {
        CDM_NCCL(r->groupStart(), "group start");
        for (int p = 0; p < W; p++) {
            if (p == c->rank) continue;
            const uint64_t nr = recvOff[p + 1] - recvOff[p];
            for (uint64_t o = 0; o < sendBytes; o += RCCL_PIECE) CDM_NCCL(r->send(src + o, std::min(RCCL_PIECE, sendBytes - o), NCCL_CHAR, p, c->nccl, s), "send");
            for (uint64_t o = 0; o < nr; o += RCCL_PIECE) CDM_NCCL(r->recv(dst + recvOff[p] + o, std::min(RCCL_PIECE, nr - o), NCCL_CHAR, p, c->nccl, s), "recv");
        }
        CDM_NCCL(r->groupEnd(), "group end");
    }
    if (W > 1 && !direct) for (int p = 0; p < W; p++) if (p != c->rank && recvOff[p + 1] > recvOff[p])
        CDM_HIP(hipMemcpyAsync((char *) recv + recvOff[p], (const char *) c->recvStage.p + (recvOff[p] - recvOff[0]), recvOff[p + 1] - recvOff[p], hipMemcpyDeviceToDevice, s));
    return CDM_OK;
}
}  // namespace

extern "C" int cdm_comm_unique_id(void *id128) {
    std::string err; Rccl *r = rccl(&err);
    if (!r) { cdm_set_error("%s", err.c_str()); return CDM_ERR_UNSUPPORTED; }
    CDM_NCCL(r->getUniqueId(id128), "unique id");
    return CDM_OK;
}
extern "C" int cdm_comm_create_rccl(cdm_ctx *ctx, int rank, int world, const void *id128, cdm_comm **out) {
    if (!ctx || !out || !id128 || world < 1 || rank < 0 || rank >= world) { cdm_set_error("cdm_comm_create_rccl: invalid argument"); return CDM_ERR_INVALID; }
    std::string err; Rccl *r = rccl(&err);
    if (!r) { cdm_set_error("%s", err.c_str()); return CDM_ERR_UNSUPPORTED; }
    CDM_HIP(hipSetDevice(ctx->device));
    cdm_comm *c = new cdm_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->api = r;
    Id128 id; memcpy(id.b, id128, sizeof(id.b));
    const int e = ((int (*)(void **, int, Id128, int)) r->commInitRank)(&c->nccl, world, id, rank);
    if (e != 0) { delete c; cdm_set_error("ncclCommInitRank(rank %d of %d on device %d) failed: %s", rank, world, ctx->device, r->errorString ? r->errorString(e) : "?"); return CDM_ERR_HIP; }
    c->ops.user = c; c->ops.all_gather_host = rcclAllGatherHost; c->ops.all_to_all_dev = rcclAllToAllDev; c->ops.all_gather_dev = rcclAllGatherDev;
    *out = c;
    return CDM_OK;
}
// ---- a stand-in for RCCL's calls inside ONE process (tests): the ranks are host threads that share a device.  send / recv are
// recorded; the group's end waits for all ranks, then every receive copies from the matching send of its peer (the i-th receive from
// p takes p's i-th send to this rank - RCCL's matching rule), device to device.  What runs on top is the RCCL transport itself - its
// buffers, its pieces, its offsets - with a peer that is not the rank itself, which one device per box cannot offer RCCL.
namespace {
struct FakeOp { const void *buf; size_t n; int peer; };
struct FakeGroup {
    int world; std::mutex m; std::condition_variable cv; int waiting = 0; unsigned long long phase = 0; bool failed = false;
    std::vector<std::vector<FakeOp>> sends, recvs; std::vector<const void *> agSend; std::vector<size_t> agBytes;
    explicit FakeGroup(int w) : world(w), sends((size_t) w), recvs((size_t) w), agSend((size_t) w), agBytes((size_t) w) {}
    void barrier() {
        std::unique_lock<std::mutex> g(m);
        const unsigned long long ph = phase;
        if (++waiting == world) { waiting = 0; phase++; cv.notify_all(); }
        else if (!cv.wait_for(g, std::chrono::seconds(120), [&] { return phase != ph; })) { failed = true; waiting = 0; phase++; cv.notify_all(); }      // (a rank that died must not leave the others waiting for ever)
    }
};
struct FakeRank { FakeGroup *g; int rank; hipStream_t stream; };
int fakeSend(const void *buf, size_t n, int, int peer, void *comm, hipStream_t) { FakeRank *f = (FakeRank *) comm; f->g->sends[f->rank].push_back(FakeOp{buf, n, peer}); return 0; }
int fakeRecv(void *buf, size_t n, int, int peer, void *comm, hipStream_t) { FakeRank *f = (FakeRank *) comm; f->g->recvs[f->rank].push_back(FakeOp{buf, n, peer}); return 0; }
thread_local FakeRank *fakeCurrent = nullptr;       // groupStart / groupEnd carry no communicator: the rank is the calling thread's
int fakeGroupStart() { return 0; }
int fakeGroupEnd() {
    FakeRank *f = fakeCurrent; if (!f) return 1;
    FakeGroup *g = f->g;
    if (hipStreamSynchronize(f->stream) != hipSuccess) g->failed = true;       // what the sends read must be there
    g->barrier();
    std::vector<size_t> next((size_t) g->world, 0);
    for (const FakeOp &rv : g->recvs[f->rank]) {
        const std::vector<FakeOp> &theirs = g->sends[rv.peer];
        size_t &k = next[rv.peer];
        while (k < theirs.size() && theirs[k].peer != f->rank) k++;
        if (k >= theirs.size() || theirs[k].n != rv.n || hipMemcpyAsync(const_cast<void *>(rv.buf), theirs[k].buf, rv.n, hipMemcpyDeviceToDevice, f->stream) != hipSuccess) { g->failed = true; break; }
        k++;
    }
    if (hipStreamSynchronize(f->stream) != hipSuccess) g->failed = true;
    g->barrier();
    const bool bad = g->failed;
    g->sends[f->rank].clear(); g->recvs[f->rank].clear();
    g->barrier();
    return bad ? 1 : 0;
}
int fakeAllGather(const void *send, void *recv, size_t bytes, int, void *comm, hipStream_t s) {
    FakeRank *f = (FakeRank *) comm; FakeGroup *g = f->g;
    if (hipStreamSynchronize(s) != hipSuccess) g->failed = true;
    g->agSend[f->rank] = send; g->agBytes[f->rank] = bytes;
    g->barrier();
    for (int p = 0; p < g->world; p++) if (g->agBytes[p] != bytes || hipMemcpyAsync((char *) recv + (size_t) p * bytes, g->agSend[p], bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) g->failed = true;
    if (hipStreamSynchronize(s) != hipSuccess) g->failed = true;
    g->barrier();
    return g->failed ? 1 : 0;
}
int fakeDestroy(void *comm) { delete (FakeRank *) comm; return 0; }
const char *fakeError(int) { return "the in-process stand-in for RCCL failed (sizes of a send and its receive differ, or a copy failed)"; }
Rccl *fakeApi() {
    static Rccl *t = [] { Rccl *a = new Rccl(); a->send = fakeSend; a->recv = fakeRecv; a->groupStart = fakeGroupStart; a->groupEnd = fakeGroupEnd; a->allGather = fakeAllGather; a->commDestroy = fakeDestroy; a->errorString = fakeError; return a; }();
    return t;
}
}  // namespace
void standinEnter(cdm_comm *c) { fakeCurrent = (c->api == fakeApi()) ? (FakeRank *) c->nccl : nullptr; }
extern "C" void *cdm_comm_standin_group(int world) { return world > 0 ? new FakeGroup(world) : nullptr; }      // (never freed: a test's handful)
extern "C" int cdm_comm_create_standin(cdm_ctx *ctx, void *group, int rank, cdm_comm **out) {
    FakeGroup *g = (FakeGroup *) group;
    if (!ctx || !g || !out || rank < 0 || rank >= g->world) { cdm_set_error("cdm_comm_create_standin: invalid argument"); return CDM_ERR_INVALID; }
    cdm_comm *c = new cdm_comm();
    c->ctx = ctx; c->rank = rank; c->world = g->world; c->api = fakeApi();
    FakeRank *f = new FakeRank{g, rank, ctx->stream};
    c->nccl = f;
    c->ops.user = c; c->ops.all_gather_host = rcclAllGatherHost; c->ops.all_to_all_dev = rcclAllToAllDev; c->ops.all_gather_dev = rcclAllGatherDev;
    *out = c;
    return CDM_OK;
}
extern "C" int cdm_comm_create_ops(cdm_ctx *ctx, int rank, int world, const cdm_comm_ops *ops, cdm_comm **out) {
    if (!ctx || !out || !ops || !ops->all_gather_host || !ops->all_to_all_dev || !ops->all_gather_dev || world < 1 || rank < 0 || rank >= world) { cdm_set_error("cdm_comm_create_ops: invalid argument"); return CDM_ERR_INVALID; }
    cdm_comm *c = new cdm_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->ops = *ops;
    *out = c;
    return CDM_OK;
}
extern "C" void cdm_comm_free(cdm_comm *c) {
    if (!c) return;
    for (cdm_comm::Stage *st : {&c->stage, &c->sendStage, &c->recvStage}) if (st->p) { (void) hipDeviceSynchronize(); (void) hipFree(st->p); st->p = nullptr; }
    if (c->nccl && c->api && c->api->commDestroy) (void) c->api->commDestroy(c->nccl);
    delete c;
}
extern "C" int cdm_comm_owned(const cdm_comm *c, uint64_t n, uint64_t *bounds) {
    if (!c || !bounds) { cdm_set_error("cdm_comm_owned: invalid argument"); return CDM_ERR_INVALID; }
    const int W = c->world;
    if (c->ownN == n && (int) c->own.size() == W + 1) { for (int p = 0; p <= W; p++) bounds[p] = c->own[p]; return CDM_OK; }
    // equal id ranges only while nothing has been cut: a DB of another size than the one the last cdm_kmermatch_dist cut the owners for
    // would be gathered by a partition its hits and alignments were not computed for
    if (!c->own.empty()) { cdm_set_error("cdm_comm_owned: the owners' ranges were cut for a DB of %llu sequences (the last cdm_kmermatch_dist), not %llu", (unsigned long long) c->ownN, (unsigned long long) n); return CDM_ERR_INVALID; }
    for (int p = 0; p <= W; p++) bounds[p] = (uint64_t) ((unsigned __int128) n * (unsigned) p / (unsigned) W);
    return CDM_OK;
}
extern "C" int cdm_comm_rank(const cdm_comm *c) { return c->rank; }
extern "C" int cdm_comm_world(const cdm_comm *c) { return c->world; }

// ------------------------------------------------------------------------------------------------ kmermatcher over W ranks
namespace {
constexpr int STALE_LEN = CDM_STALE_MAX + 5;       // [0] count, [1] sequence id, [2..63] positions, [66] = the scan reached the end of the range
struct PartGuard { cdm_kpart *p = nullptr; ~PartGuard() { if (p) cdm_kpart_free(p); } };
// kmermatcher's first half for rank R's k-mer range, the extraction split by reads: every rank extracts its block of the sequences,
// all-to-alls (keys, values, and the whole-sequence hash tuples to the last rank) carry the tuples to the ranks of their k-mer
// ranges.  CDM_DIST_EXTRACT=all: every rank extracts every sequence and keeps its range (cdm_kmermatch_part; the A/B and what
// shard.py does).
int firstHalf(cdm_ctx *ctx, cdm_comm *cm, const cdm_seqdb *db, const cdm_kmer_params *par, cdm_kpart **out) {
    const int W = cm->world, R = cm->rank;
    // Three ways to a rank's range of the k-mer space.  "split": every rank extracts its block of the reads and the tuples travel - 49 GB
    // at 50 M reads, 0.98 s / W^2 over W - 1 links of ~50 GB/s each.  "all": every rank extracts ALL reads, orders them by slice, keeps
    // its range - 52 ms that do not shrink with W, nothing travels.  The tuples' exchange pays from about 6 ranks on (W = 4: 19 + 61 ms
    // against 52; W = 8: 13 + 15), so that is the default rule; CDM_DIST_EXTRACT=split|all|part forces one ("part": round 3's
    // cdm_kmermatch_part, whose ranges are equal slices of the k-mer space by value and far from equal shares).
    const char *how = cdmGetenv("CDM_DIST_EXTRACT");
    if (W == 1 || (how && !strcmp(how, "part"))) { cm->lastPath = 4; return cdm_kmermatch_part(ctx, db, par, R, W, out); }
    const bool everything = how ? !strcmp(how, "all") : W < 6;
    cm->lastPath = everything ? 2 : 3;
    // a DB of one read length: every rank extracts all reads on the 8-byte slot layout and its head pass keeps the rank's range of head
    // digits, cut from the head histogram into equal shares (kmermatch.hip KmerJob::headRange) - same ranges on every rank, nothing travels
    if (everything && cdm_kmermatch_part_takes_slots(db, par)) return cdm_kmermatch_part(ctx, db, par, R, W, out);
    constexpr int F = CDM_KPART_SLICES;
    if (everything) {
        if (int rc = cdm_kmermatch_split_begin(ctx, db, par, 0, 1, out)) return rc;
        std::vector<uint64_t> fineOff((size_t) F + 1);
        const void *keys, *vals, *hkeys, *hvals; int vb = 0; uint64_t nHash = 0;
        if (int rc = cdm_kpart_outgoing(*out, fineOff.data(), &keys, &vals, &vb, &hkeys, &hvals, &nHash)) return rc;
        std::vector<int> cut(1, 0);          // (every rank holds the same counts and cuts the same ranges)
        const uint64_t target = (fineOff[F] + (uint64_t) W - 1) / (uint64_t) W;
        uint64_t acc = 0;
        for (int f = 0; f < F; f++) { const uint64_t c = fineOff[f + 1] - fineOff[f]; if (acc && acc + c > target && (int) cut.size() < W) { cut.push_back(f); acc = 0; } acc += c; }
        while ((int) cut.size() < W) cut.push_back(F);
        cut.push_back(F);
        const uint64_t lo = fineOff[cut[R]], m = fineOff[cut[R + 1]] - lo, h = (R == W - 1) ? nHash : 0;
        DevBuf<uint64_t> rk, rhk; DevBuf<char> rv, rhv;       // (split_finish lets go of the extraction buffers before it takes the tuples over)
        if (!rk.alloc(m) || !rv.alloc(m * vb) || !rhk.alloc(h) || !rhv.alloc(h * vb)) { cdm_set_error("cdm_kmermatch_dist: out of device memory for %llu k-mer tuples", (unsigned long long) (m + h)); return CDM_ERR_HIP; }
        CDM_HIP(hipSetDevice(ctx->device));
        if (m) { CDM_HIP(hipMemcpyAsync(rk.p, (const char *) keys + lo * 8, m * 8, hipMemcpyDeviceToDevice, ctx->stream)); CDM_HIP(hipMemcpyAsync(rv.p, (const char *) vals + lo * vb, m * vb, hipMemcpyDeviceToDevice, ctx->stream)); }
        if (h) { CDM_HIP(hipMemcpyAsync(rhk.p, hkeys, h * 8, hipMemcpyDeviceToDevice, ctx->stream)); CDM_HIP(hipMemcpyAsync(rhv.p, hvals, h * vb, hipMemcpyDeviceToDevice, ctx->stream)); }
        CDM_HIP(hipStreamSynchronize(ctx->stream));
        if (int rc = cdm_kpart_set_range(*out, R, W)) return rc;
        return cdm_kmermatch_split_finish(ctx, *out, rk.p, rv.p, m, rhk.p, rhv.p, h, lo ? 1 : 0);
    }
    if (int rc = cdm_kmermatch_split_begin(ctx, db, par, R, W, out)) return rc;
    std::vector<uint64_t> fineOff((size_t) F + 1);
    const void *keys, *vals, *hkeys, *hvals; int vb = 0; uint64_t nHash = 0;
    if (int rc = cdm_kpart_outgoing(*out, fineOff.data(), &keys, &vals, &vb, &hkeys, &hvals, &nHash)) return rc;
    const size_t row = (size_t) F + 1;                    // tuples per fine slice of the k-mer space, then the hash tuples
    std::vector<uint64_t> counts(row), matrix(row * W);
    for (int f = 0; f < F; f++) counts[f] = fineOff[f + 1] - fineOff[f];
    counts[F] = nHash;
    if (int rc = coAllGatherHost(cm, counts.data(), matrix.data(), row * 8)) return rc;
    // the ranks' k-mer ranges: runs of fine slices with about the same number of tuples over all ranks (every rank cuts the same way)
    std::vector<int> cut(1, 0);
    {
        std::vector<uint64_t> fine((size_t) F, 0); uint64_t grand = 0;
        for (int p = 0; p < W; p++) for (int f = 0; f < F; f++) { fine[f] += matrix[p * row + f]; grand += matrix[p * row + f]; }
        const uint64_t target = (grand + (uint64_t) W - 1) / (uint64_t) W;
        uint64_t acc = 0;
        for (int f = 0; f < F; f++) { if (acc && acc + fine[f] > target && (int) cut.size() < W) { cut.push_back(f); acc = 0; } acc += fine[f]; }
        while ((int) cut.size() < W) cut.push_back(F);      // (fewer occupied slices than ranks: the last ranks get nothing)
        cut.push_back(F);
    }
    std::vector<uint64_t> off((size_t) W + 1), recvCnt((size_t) W + 1, 0), hashCnt((size_t) W + 1, 0);
    for (int p = 0; p <= W; p++) off[p] = fineOff[cut[p]];
    bool below = false;
    for (int p = 0; p < W; p++) {
        uint64_t mine = 0;
        for (int f = cut[R]; f < cut[R + 1]; f++) mine += matrix[p * row + f];
        recvCnt[p + 1] = recvCnt[p] + mine;
        hashCnt[p + 1] = hashCnt[p] + (R == W - 1 ? matrix[p * row + F] : 0);
        for (int f = 0; f < cut[R]; f++) below = below || matrix[p * row + f] != 0;
    }
    const uint64_t m = recvCnt[W], h = hashCnt[W];
    DevBuf<uint64_t> rk, rhk; DevBuf<char> rv, rhv;
    if (!rk.alloc(m) || !rv.alloc(m * vb) || !rhk.alloc(h) || !rhv.alloc(h * vb)) { cdm_set_error("cdm_kmermatch_dist: out of device memory for %llu received k-mer tuples", (unsigned long long) (m + h)); return CDM_ERR_HIP; }
    CDM_HIP(hipSetDevice(ctx->device));
    std::vector<uint64_t> so((size_t) W + 1), ro((size_t) W + 1);
    auto exchange = [&](const void *send, void *recv, const std::vector<uint64_t> &sOff, const std::vector<uint64_t> &rOff, uint64_t width) -> int {
        for (int p = 0; p <= W; p++) { so[p] = sOff[p] * width; ro[p] = rOff[p] * width; }
        return coAllToAllDev(cm, send, so.data(), recv, ro.data(), ctx->stream);
    };
    if (int rc = exchange(keys, rk.p, off, recvCnt, 8)) return rc;
    if (int rc = exchange(vals, rv.p, off, recvCnt, (uint64_t) vb)) return rc;
    std::vector<uint64_t> hashOff((size_t) W + 1, 0); hashOff[W] = nHash;       // everything to the last rank
    if (int rc = exchange(hkeys, rhk.p, hashOff, hashCnt, 8)) return rc;
    if (int rc = exchange(hvals, rhv.p, hashOff, hashCnt, (uint64_t) vb)) return rc;
    CDM_HIP(hipStreamSynchronize(ctx->stream));
    return cdm_kmermatch_split_finish(ctx, *out, rk.p, rv.p, m, rhk.p, rhv.p, h, below ? 1 : 0);
}
}  // namespace
namespace {
// a rank's view of the whole prefilter result: the rows of the queries [lo, hi) as they are, of every other query its self hit only
// (the first record of a row: k_offsets / k_self)
__global__ void k_view_offsets(const uint64_t *__restrict__ off, uint64_t n, uint64_t lo, uint64_t hi, uint64_t *__restrict__ out) {
    const uint64_t q = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (q > n) return;
    const uint64_t owned = off[hi] - off[lo];
    out[q] = q < lo ? q : q < hi ? lo + (off[q] - off[lo]) : lo + owned + (q - hi);
}
__global__ void k_view_selfs(const uint64_t *__restrict__ off, const HitRec *__restrict__ rec, uint64_t n, uint64_t lo, uint64_t hi, const uint64_t *__restrict__ newOff, HitRec *__restrict__ out) {
    const uint64_t q = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n || (q >= lo && q < hi)) return;
    out[newOff[q]] = rec[off[q]];
}
__global__ void k_pick(const uint64_t *__restrict__ a, const uint64_t *__restrict__ idx, uint32_t m, uint64_t *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) out[t] = a[idx[t]];
}
// Two ranks: kmermatcher WHOLE on every rank, no exchange at all.  The exact scheme moves the group keys - 32 GB per step at 50 M reads,
// 8 GB per rank over the one link two devices share: more time than the second device saves - so with two ranks
// only the stages behind kmermatcher are split: every rank computes all hits, cuts the same owners' ranges from the rows' sizes and
// keeps its view.  CDM_DIST_KMER=replicate|exchange forces either for any world.
// ... or, `ranges`: kmermatcher's first half split over the ranks by ranges of the k-mer space (kmermatch.hip kmermatchPassesT with its
// rank hooks: each range's sort 1 and grouping on one rank, the kept group keys - the wide form included - all-gathered), sort 2 and the
// vote on every rank.  What a DB with the wide group key takes since round 5 (until then: every rank ran kmermatcher whole), and
// CDM_DIST_KMER=ranges for any DB.
int ranksGatherHost(void *user, const void *send, void *recv, uint64_t bytes) { return coAllGatherHost((cdm_comm *) user, send, recv, bytes); }
int ranksGatherDev(void *user, const void *send, uint64_t bytes, void *recv, const uint64_t *ro, void *stream) { return coAllGatherDev((cdm_comm *) user, send, bytes, recv, ro, stream); }
int replicatedKmermatch(cdm_ctx *ctx, cdm_comm *cm, const cdm_seqdb *db, const cdm_kmer_params *par, cdm_hits **out, bool ranges = false) {
    const int W = cm->world, R = cm->rank;
    cdm_hits *full = nullptr;
    if (ranges && W > 1) {
        KmerRanks kr; kr.rank = R; kr.world = W; kr.user = cm; kr.gatherHost = ranksGatherHost; kr.gatherDev = ranksGatherDev;
        if (cdmGetenv("CDM_KMER_SORT") || cdmGetenv("CDM_KMER_SORT2")) { cdm_set_error("cdm_kmermatch_dist: the A/B switches CDM_KMER_SORT / CDM_KMER_SORT2 apply to the single-device path only"); return CDM_ERR_INVALID; }
        if (int rc = cdm_kmermatch_ranks_impl(ctx, db, par, &kr, &full)) return rc;
    } else if (int rc = cdm_kmermatch(ctx, db, par, &full)) return rc;
    struct Guard { cdm_hits *h; ~Guard() { if (h) cdm_hits_free(h); } } guard{full};
    CDM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint64_t n = db->n;
    constexpr int S = 4096;
    std::vector<uint64_t> sb((size_t) S + 1), at((size_t) S + 1), own((size_t) W + 1, 0);
    for (int t = 0; t <= S; t++) sb[t] = (uint64_t) ((unsigned __int128) n * (unsigned) t / (unsigned) S);
    {
        DevBuf<uint64_t> dIdx, dVal;
        if (!dIdx.alloc(S + 1) || !dVal.alloc(S + 1)) { cdm_set_error("cdm_kmermatch_dist: out of device memory"); return CDM_ERR_HIP; }
        CDM_HIP(hipMemcpyAsync(dIdx.p, sb.data(), (S + 1) * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pick, dim3((S + 256) / 256), dim3(256), 0, s, (const uint64_t *) full->off, (const uint64_t *) dIdx.p, (uint32_t) (S + 1), dVal.p);
        CDM_HIP(hipMemcpyAsync(at.data(), dVal.p, (S + 1) * 8, hipMemcpyDeviceToHost, s));
        CDM_HIP(hipStreamSynchronize(s));
    }
    {   // equal shares of the records (a row's records = its group's keys + the self hit: the same weight cdm_kmermatch_dist cuts by)
        const uint64_t target = (at[S] + (uint64_t) W - 1) / (uint64_t) W;
        uint64_t acc = 0; int k = 1;
        for (int t = 0; t < S && k < W; t++) { const uint64_t c = at[t + 1] - at[t]; if (acc && acc + c > target) { own[k++] = sb[t]; acc = 0; } acc += c; }
        while (k < W) own[k++] = n;
        own[W] = n;
        cm->own = own; cm->ownN = n;
    }
    const uint64_t lo = own[R], hi = own[R + 1];
    uint64_t b[2] = {0, 0};
    CDM_HIP(hipMemcpyAsync(&b[0], full->off + lo, 8, hipMemcpyDeviceToHost, s)); CDM_HIP(hipMemcpyAsync(&b[1], full->off + hi, 8, hipMemcpyDeviceToHost, s));
    CDM_HIP(hipStreamSynchronize(s));
    const uint64_t owned = b[1] - b[0], count = owned + (n - (hi - lo));
    cdm_hits *v = new cdm_hits(); v->n = n; v->count = count;
    if (cdmMalloc(&v->off, (n + 1) * 8) != hipSuccess || cdmMalloc(&v->rec, (count + 1) * sizeof(HitRec)) != hipSuccess) { cdm_hits_free(v); cdm_set_error("cdm_kmermatch_dist: out of device memory"); return CDM_ERR_HIP; }
    hipLaunchKernelGGL(k_view_offsets, CDM_GRID((n + 256) / 256, 256), dim3(256), 0, s, (const uint64_t *) full->off, n, lo, hi, v->off);
    if (n) hipLaunchKernelGGL(k_view_selfs, CDM_GRID((n + 255) / 256, 256), dim3(256), 0, s, (const uint64_t *) full->off, (const HitRec *) full->rec, n, lo, hi, (const uint64_t *) v->off, v->rec);
    if (owned) CDM_HIP(hipMemcpyAsync(v->rec + lo, full->rec + b[0], owned * sizeof(HitRec), hipMemcpyDeviceToDevice, s));
    { const hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_hits_free(v); cdm_set_error("cdm_kmermatch_dist: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    *out = v;
    return CDM_OK;
}
}  // namespace
static int kmermatchDistBody(cdm_ctx *ctx, cdm_comm *cm, const cdm_seqdb *db, const cdm_kmer_params *par, cdm_hits **out) {
    const int W = cm->world, R = cm->rank;
    {
        const char *km = cdmGetenv("CDM_DIST_KMER");
        const bool replicate = km ? !strcmp(km, "replicate") : (W == 2 && !cdmGetenv("CDM_DIST_EXTRACT"));
        // a DB that takes the wide group key - 25 M sequences with contigs: the exchange of group keys by owner carries the narrow form only;
        // its first half goes over the ranks by k-mer ranges, the kept keys are all-gathered
        const bool ranges = W > 1 && ((km && !strcmp(km, "ranges")) || (cdm_kmermatch_needs_wide_key(db) && !(km && !strcmp(km, "replicate"))));
        if (ranges) { cm->lastPath = 5; return replicatedKmermatch(ctx, cm, db, par, out, true); }
        if ((W > 1 && replicate) || cdm_kmermatch_needs_wide_key(db)) { cm->lastPath = 1; return replicatedKmermatch(ctx, cm, db, par, out); }
    }
    PartGuard g;
    if (int rc = firstHalf(ctx, cm, db, par, &g.p)) return rc;
    uint64_t info[4];
    cdm_kpart_info(g.p, info);
    // ---- the left-over list of the reference's last per-target scan: it starts at k-mer-order index J = number of ALL kept group keys
    // (assignGroup's compaction, :875-887) - which range holds that index, and what the ranges behind it begin with
    std::vector<uint64_t> infos(2 * (size_t) W);
    { const uint64_t mine[2] = {info[0], info[1]}; if (int rc = coAllGatherHost(cm, mine, infos.data(), 16)) return rc; }
    uint64_t J = 0; for (int p = 0; p < W; p++) J += infos[2 * p + 1];
    int holder = -1; uint64_t jLocal = 0;
    { uint64_t base = 0; for (int p = 0; p < W; p++) { if (J < base + infos[2 * p]) { holder = p; jLocal = J - base; break; } base += infos[2 * p]; } }
    uint32_t mineStale[STALE_LEN]; memset(mineStale, 0, sizeof(mineStale));
    if (holder >= 0 && R >= holder) if (int rc = cdm_kpart_stale(ctx, g.p, R == holder ? jLocal : 0, mineStale)) return rc;
    std::vector<uint32_t> lists((size_t) W * STALE_LEN);
    if (int rc = coAllGatherHost(cm, mineStale, lists.data(), sizeof(mineStale))) return rc;
    uint32_t stale[STALE_LEN]; memset(stale, 0, sizeof(stale));
    if (holder >= 0) {      // the scan collects tuples while they belong to one sequence; it goes on into the next range when it consumed the current one
        uint32_t cnt = 0; bool have = false; uint32_t target = 0;
        for (int p = holder; p < W; p++) {
            const uint32_t *l = lists.data() + (size_t) p * STALE_LEN;
            const uint32_t c = l[0];
            if (c) {
                if (!have) { target = l[1]; have = true; } else if (l[1] != target) break;
                for (uint32_t j = 0; j < c && cnt < (uint32_t) CDM_STALE_MAX; j++) stale[2 + cnt++] = l[2 + j];
            }
            if (!l[CDM_STALE_MAX + 4]) break;
        }
        if (cnt >= (uint32_t) CDM_STALE_MAX) { cdm_set_error("cdm_kmermatch_dist: the reference's last per-target scan would run over %d or more left-over tuples; not reproduced", CDM_STALE_MAX); return CDM_ERR_UNSUPPORTED; }
        stale[0] = cnt; stale[1] = have ? target : 0;
    }
    // ---- the all-to-all of the group keys: slice p of the keys grouped by representative goes to the owner of those representatives
    // the owners' id ranges: equal shares of (group keys + sequences), cut from all ranks' histograms over 4096 equal id ranges (representatives
    // are the longest, then lowest ids of their k-mer groups - with equal id ranges the first of 8 ranks owned 88 % of the keys)
    const void *keys = nullptr;
    const uint64_t nSeq = db->n;
    std::vector<uint64_t> own((size_t) W + 1, 0);
    {
        constexpr int S = 4096;
        std::vector<uint64_t> sb((size_t) S + 1), so((size_t) S + 1), hist((size_t) S), all((size_t) S * W);
        for (int t = 0; t <= S; t++) sb[t] = (uint64_t) ((unsigned __int128) nSeq * (unsigned) t / (unsigned) S);
        if (int rc = cdm_kpart_gather_at(ctx, g.p, S + 1, sb.data(), so.data(), &keys)) return rc;
        so[S] = info[1];
        for (int t = 0; t < S; t++) hist[t] = so[t + 1] - so[t];
        if (int rc = coAllGatherHost(cm, hist.data(), all.data(), (uint64_t) S * 8)) return rc;
        uint64_t grand = 0;
        // (weight of an id range: its group keys - sort 2, vote and the stages behind them work per key - plus its sequences: the owner of
        //  a range sends those rows of the corrected and of the next DB to every rank, about as many nanoseconds per row over the links
        //  as a key costs in the kernels)
        for (int t = 0; t < S; t++) { uint64_t c = sb[t + 1] - sb[t]; for (int p = 0; p < W; p++) c += all[(size_t) p * S + t]; hist[t] = c; grand += c; }
        const uint64_t target = (grand + (uint64_t) W - 1) / (uint64_t) W;
        uint64_t acc = 0; int at = 1;
        for (int t = 0; t < S && at < W; t++) { if (acc && acc + hist[t] > target) { own[at++] = sb[t]; acc = 0; } acc += hist[t]; }
        while (at < W) own[at++] = nSeq;
        own[W] = nSeq;
        cm->own = own; cm->ownN = nSeq;
    }
    std::vector<uint64_t> off((size_t) W + 1);
    if (int rc = cdm_kpart_gather_at(ctx, g.p, W + 1, own.data(), off.data(), &keys)) return rc;
    off[W] = info[1];
    std::vector<uint64_t> counts((size_t) W), matrix((size_t) W * W);
    for (int p = 0; p < W; p++) counts[p] = off[p + 1] - off[p];
    if (int rc = coAllGatherHost(cm, counts.data(), matrix.data(), (uint64_t) W * 8)) return rc;
    std::vector<uint64_t> sendOff((size_t) W + 1), recvOff((size_t) W + 1, 0);
    for (int p = 0; p <= W; p++) sendOff[p] = off[p] * 8;
    for (int p = 0; p < W; p++) recvOff[p + 1] = recvOff[p] + matrix[(size_t) p * W + R] * 8;      // rank p's slice for me, in rank = k-mer order
    const uint64_t nRecv = recvOff[W] / 8;
    DevBuf<uint64_t> recv;
    if (!recv.alloc(nRecv)) { cdm_set_error("cdm_kmermatch_dist: out of device memory for %llu received group keys", (unsigned long long) nRecv); return CDM_ERR_HIP; }
    CDM_HIP(hipSetDevice(ctx->device));
    if (int rc = coAllToAllDev(cm, keys, sendOff.data(), recv.p, recvOff.data(), ctx->stream)) return rc;
    CDM_HIP(hipStreamSynchronize(ctx->stream));
    // ---- second half on what arrived; the heads of the sorted arrays go round once more (a rank's last scan runs into the next ranks' first tuples)
    const int capHead = cdm_kpart_cont_cap() + 3;
    std::vector<uint32_t> head((size_t) capHead + 4, 0);       // [0..1] tuples (64 bit), [2] last target, [3] unused, [4..] the head
    uint64_t info2[2] = {0, 0};
    if (int rc = cdm_kpart_sort(ctx, g.p, nRecv ? recv.p : nullptr, nRecv, head.data() + 4, info2)) return rc;
    recv.free();
    memcpy(head.data(), &info2[0], 8); head[2] = (uint32_t) info2[1];
    std::vector<uint32_t> heads((size_t) W * head.size());
    if (int rc = coAllGatherHost(cm, head.data(), heads.data(), head.size() * 4)) return rc;
    std::vector<uint32_t> cont;
    if (info2[0] != 0) {
        const uint32_t t = (uint32_t) info2[1];
        uint32_t thenStale = 1;
        cont.assign(3, 0);
        for (int p = R + 1; p < W; p++) {
            const uint32_t *h = heads.data() + (size_t) p * head.size();
            uint64_t cnt; memcpy(&cnt, h, 8);
            if (cnt == 0) continue;
            const uint32_t *hd = h + 4;                        // hd[0] entries, hd[1] their target id, hd[2] = 1 if the head is the whole array
            if (hd[1] != t) { thenStale = 0; break; }
            if (hd[0] > (uint32_t) cdm_kpart_cont_cap()) { cdm_set_error("cdm_kmermatch_dist: the scan of a rank's last target runs over more than %d tuples of the next rank; not reproduced", cdm_kpart_cont_cap()); return CDM_ERR_UNSUPPORTED; }
            cont.insert(cont.end(), hd + 3, hd + 3 + hd[0]);
            if (!hd[2]) { thenStale = 0; break; }
        }
        cont[0] = (uint32_t) (cont.size() - 3); cont[1] = t; cont[2] = thenStale;
    }
    return cdm_kpart_vote(ctx, g.p, cont.empty() ? nullptr : cont.data(), stale, out);
}

// ------------------------------------------------------------------------------------------------ DBs: the owned ranges, all-gathered
namespace {
__global__ void k_row_flags(const uint8_t *__restrict__ hasN, uint64_t lo, uint64_t m, uint8_t *__restrict__ flags) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) flags[i] = (hasN[lo + i] & 2u) ? 1 : 0;
}
}  // namespace
static int allgatherOwnedBody(cdm_ctx *ctx, cdm_comm *cm, const cdm_seqdb *local, cdm_seqdb **out) {
    const int W = cm->world, R = cm->rank;
    CDM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<uint64_t> own((size_t) W + 1);
    if (int rc = cdm_comm_owned(cm, local->n, own.data())) return rc;
    const uint64_t n = local->n, lo = own[R], hi = own[R + 1], m = hi - lo;
    uint32_t wb[2] = {0, 0};
    if (n) { CDM_HIP(hipMemcpyAsync(&wb[0], local->woff + lo, 4, hipMemcpyDeviceToHost, s)); CDM_HIP(hipMemcpyAsync(&wb[1], local->woff + hi, 4, hipMemcpyDeviceToHost, s)); CDM_HIP(hipStreamSynchronize(s)); }
    const uint64_t w0 = wb[0], w = wb[1] - wb[0];
    std::vector<uint64_t> metas(3 * (size_t) W);
    { const uint64_t mine[3] = {m, w, local->raw ? 1ull : 0ull}; if (int rc = coAllGatherHost(cm, mine, metas.data(), 24)) return rc; }
    uint64_t nAll = 0, wAll = 0; bool anyRaw = false;
    std::vector<uint64_t> seqOff((size_t) W + 1, 0), wordOff((size_t) W + 1, 0);
    for (int p = 0; p < W; p++) { seqOff[p + 1] = seqOff[p] + metas[3 * p]; wordOff[p + 1] = wordOff[p] + metas[3 * p + 1]; anyRaw |= metas[3 * p + 2] != 0; }
    nAll = seqOff[W]; wAll = wordOff[W];
    if (nAll != n) { cdm_set_error("cdm_seqdb_allgather_owned: the ranks' DBs differ in size (%llu sequences here, %llu owned in all)", (unsigned long long) n, (unsigned long long) nAll); return CDM_ERR_INVALID; }
    DevBuf<uint32_t> codes, lens, keys; DevBuf<uint16_t> mask; DevBuf<uint8_t> ext, raw, flags, myFlags, zeros;
    if (!codes.alloc(wAll) || !mask.alloc(wAll) || !lens.alloc(nAll) || !keys.alloc(nAll) || !ext.alloc(nAll) || (anyRaw && (!raw.alloc(16 * wAll) || !flags.alloc(nAll) || !myFlags.alloc(m)))) {
        cdm_set_error("cdm_seqdb_allgather_owned: out of device memory"); return CDM_ERR_HIP;
    }
    auto gather = [&](const void *send, uint64_t unit, const std::vector<uint64_t> &offs, uint64_t count, void *recvBuf) -> int {
        std::vector<uint64_t> ro((size_t) W + 1);
        for (int p = 0; p <= W; p++) ro[p] = offs[p] * unit;
        return coAllGatherDev(cm, send, count * unit, recvBuf, ro.data(), s);
    };
    if (int rc = gather(local->codes + w0, 4, wordOff, w, codes.p)) return rc;
    if (int rc = gather(reinterpret_cast<const uint16_t *>(local->nmask) + w0, 2, wordOff, w, mask.p)) return rc;
    if (int rc = gather(local->len + lo, 4, seqOff, m, lens.p)) return rc;
    if (int rc = gather(local->key + lo, 4, seqOff, m, keys.p)) return rc;
    if (int rc = gather(local->ext + lo, 1, seqOff, m, ext.p)) return rc;
    if (anyRaw) {       // letters beyond ACGTN: the original bytes (16 per code word) and which rows count; a rank without such letters sends zeros
        const uint8_t *rawSend = nullptr;
        if (local->raw) { rawSend = local->raw + 16 * w0; if (m) hipLaunchKernelGGL(k_row_flags, CDM_GRID((m + 255) / 256, 256), dim3(256), 0, s, local->hasN, lo, m, myFlags.p); }
        else {
            if (!zeros.alloc(16 * w + 16)) { cdm_set_error("cdm_seqdb_allgather_owned: out of device memory"); return CDM_ERR_HIP; }
            CDM_HIP(hipMemsetAsync(zeros.p, 0, 16 * w + 16, s)); CDM_HIP(hipMemsetAsync(myFlags.p, 0, m + 1, s));
            rawSend = zeros.p;
        }
        if (int rc = gather(rawSend, 16, wordOff, w, raw.p)) return rc;
        if (int rc = gather(myFlags.p, 1, seqOff, m, flags.p)) return rc;
    }
    CDM_HIP(hipStreamSynchronize(s));
    cdm_seqdb *o = nullptr;
    if (int rc = cdm_seqdb_from_packed_ext(ctx, codes.p, mask.p, lens.p, keys.p, ext.p, nAll, wAll, &o)) return rc;
    if (anyRaw && wAll) if (int rc = cdm_seqdb_attach_raw(ctx, o, raw.p, flags.p)) { cdm_seqdb_free(o); return rc; }
    *out = o;
    return CDM_OK;
}

extern "C" int cdm_kmermatch_dist(cdm_ctx *ctx, cdm_comm *cm, const cdm_seqdb *db, const cdm_kmer_params *par, cdm_hits **out) {
    if (!ctx || !cm || !db || !par || !out) { cdm_set_error("cdm_kmermatch_dist: invalid argument"); return CDM_ERR_INVALID; }
    cm->knowsFailure = false; *out = nullptr;
    const int rc = kmermatchDistBody(ctx, cm, db, par, out);
    const int all = finish(cm, rc, "cdm_kmermatch_dist");
    if (all != CDM_OK && *out) { cdm_hits_free(*out); *out = nullptr; }
    return all;
}
extern "C" int cdm_seqdb_allgather_owned(cdm_ctx *ctx, cdm_comm *cm, const cdm_seqdb *local, cdm_seqdb **out) {
    if (!ctx || !cm || !local || !out) { cdm_set_error("cdm_seqdb_allgather_owned: invalid argument"); return CDM_ERR_INVALID; }
    cm->knowsFailure = false; *out = nullptr;
    const int rc = allgatherOwnedBody(ctx, cm, local, out);
    const int all = finish(cm, rc, "cdm_seqdb_allgather_owned");
    if (all != CDM_OK && *out) { cdm_seqdb_free(*out); *out = nullptr; }
    return all;
}
extern "C" int cdm_comm_last_path(const cdm_comm *c) { return c ? c->lastPath : 0; }

// One iteration of the reads loop over the ranks: the hits of the owned representatives, their alignments, and the two DBs - complete
// on every rank and identical to the single-device ones.  rpar / apar may be NULL (defaults).  Any of the outputs may be NULL.
extern "C" int cdm_reads_iteration_dist(cdm_ctx *ctx, cdm_comm *cm, const cdm_seqdb *db, const cdm_kmer_params *kpar, const cdm_rescore_params *rpar,
                                        const cdm_ancient_params *apar, cdm_hits **hitsOut, cdm_alns **alnsOut, cdm_seqdb **corrOut, cdm_seqdb **nextOut) {
    cdm_hits *hits = nullptr; cdm_alns *alns = nullptr; cdm_seqdb *cLocal = nullptr, *corr = nullptr, *nLocal = nullptr, *next = nullptr;
    int rc = cdm_kmermatch_dist(ctx, cm, db, kpar, &hits);
    if (rc == CDM_OK) rc = cdm_rescore(ctx, db, hits, rpar, &alns);
    if (rc == CDM_OK) rc = cdm_correct(ctx, db, alns, apar, &cLocal);
    if (rc == CDM_OK) rc = cdm_seqdb_allgather_owned(ctx, cm, cLocal, &corr);
    if (cLocal) cdm_seqdb_free(cLocal);
    if (rc == CDM_OK) rc = cdm_extend(ctx, corr, alns, apar, &nLocal, nullptr);
    if (rc == CDM_OK) rc = cdm_seqdb_allgather_owned(ctx, cm, nLocal, &next);
    if (nLocal) cdm_seqdb_free(nLocal);
    if (rc != CDM_OK) rc = finish(cm, rc, "cdm_reads_iteration_dist");       // (a stage that failed on this rank alone: the peers wait in the next collective)
    if (rc == CDM_OK && hitsOut) { *hitsOut = hits; hits = nullptr; }
    if (rc == CDM_OK && alnsOut) { *alnsOut = alns; alns = nullptr; }
    if (rc == CDM_OK && corrOut) { *corrOut = corr; corr = nullptr; }
    if (rc == CDM_OK && nextOut) { *nextOut = next; next = nullptr; }
    if (hits) cdm_hits_free(hits);
    if (alns) cdm_alns_free(alns);
    if (corr) cdm_seqdb_free(corr);
    if (next) cdm_seqdb_free(next);
    return rc;
}

// One iteration of the CONTIG loop (data/nuclassemble.sh:148-196) over the ranks: kmermatcher with the contig parameters over the ranks,
// rescorediagonal / ancient_correction / ancient_contig_merge on the owned queries - the reference's `omp for` over independent queries
// (ancientContigsResults.cpp:94-509; shared state: the wasExtended bytes, :271-467, which travel with the owned rows) -, the corrected and
// the merged DB all-gathered.  The script's cyclecheck step is the caller's (cdm_cyclecheck on the complete DB: every rank, same result).
extern "C" int cdm_contig_iteration_dist(cdm_ctx *ctx, cdm_comm *cm, const cdm_seqdb *db, const cdm_kmer_params *kpar, const cdm_rescore_params *rpar,
                                         const cdm_ancient_params *apar, float mergeSeqIdThr, cdm_alns **alnsOut, cdm_seqdb **corrOut, cdm_seqdb **nextOut) {
    if (!ctx || !cm || !db || !kpar || !apar) { cdm_set_error("cdm_contig_iteration_dist: invalid argument"); return CDM_ERR_INVALID; }
    cdm_hits *hits = nullptr; cdm_alns *alns = nullptr; cdm_seqdb *cLocal = nullptr, *corr = nullptr, *nLocal = nullptr, *next = nullptr;
    int rc = cdm_kmermatch_dist(ctx, cm, db, kpar, &hits);
    if (rc == CDM_OK) rc = cdm_rescore(ctx, db, hits, rpar, &alns);
    if (hits) cdm_hits_free(hits);
    if (rc == CDM_OK) rc = cdm_correct(ctx, db, alns, apar, &cLocal);
    if (rc == CDM_OK) rc = cdm_seqdb_allgather_owned(ctx, cm, cLocal, &corr);
    if (cLocal) cdm_seqdb_free(cLocal);
    if (rc == CDM_OK) rc = cdm_contig_merge(ctx, corr, alns, apar, mergeSeqIdThr, &nLocal);
    if (rc == CDM_OK) rc = cdm_seqdb_allgather_owned(ctx, cm, nLocal, &next);
    if (nLocal) cdm_seqdb_free(nLocal);
    if (rc != CDM_OK) rc = finish(cm, rc, "cdm_contig_iteration_dist");
    if (rc == CDM_OK && alnsOut) { *alnsOut = alns; alns = nullptr; }
    if (rc == CDM_OK && corrOut) { *corrOut = corr; corr = nullptr; }
    if (rc == CDM_OK && nextOut) { *nextOut = next; next = nullptr; }
    if (alns) cdm_alns_free(alns);
    if (corr) cdm_seqdb_free(corr);
    if (next) cdm_seqdb_free(next);
    return rc;
}
