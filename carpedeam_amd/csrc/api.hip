// C-ABI entry points: context, sequence DB (upload / 2-bit packing / download), hit and alignment containers.
// Stage kernels live in correct.hip, rescore.hip, kmermatch.hip, extend.hip, synth.hip.
#include <cstdarg>
#include <atomic>
#include <cstring>
#include <vector>

#include <algorithm>
#include "common.h"
#include "devutil.h"

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";
void cdm_set_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char *cdm_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------ caching allocator, CDM_* switches
#include "pool.h"
const char *cdmGetenv(const char *name) { return cdmenv::get(name); }
extern "C" void cdm_env_refresh(void) { (void) cdmenv::refresh(); }
extern "C" void cdm_pool_stats(uint64_t out[8]) {
    cdmpool::Stats &st = cdmpool::stats();
    out[0] = st.requests.load(); out[1] = st.cached.load(); out[2] = st.mallocs.load(); out[3] = st.mallocBytes.load(); out[4] = st.mallocNs.load(); out[5] = st.trims.load();
    out[6] = out[7] = 0;
    cdmpool::Registry &r = cdmpool::registry();
    std::lock_guard<std::mutex> g(r.m);
    for (cdmpool::Pool *q : r.pools) {
        std::lock_guard<std::mutex> g2(q->m);
        for (const cdmpool::Arena *a : {&q->small, &q->large}) for (const auto &kv : a->blocks) { if (kv.second.state != cdmpool::B_HOLE) out[6] += kv.second.size; if (kv.second.state == cdmpool::B_USED) out[7] += kv.second.size; }
    }
}
extern "C" void cdm_pool_headroom(float factor) { cdmpool::headroom().store(factor > 1.0f ? std::min(factor, 4.0f) : 1.0f, std::memory_order_relaxed); }
hipError_t cdmMallocRaw(void **p, size_t bytes) { return cdmpool::allocate(p, bytes); }
void cdmFree(void *p) { cdmpool::release(p); }
void cdmPoolTrim() { cdmpool::trimMine(); }
float cdmPoolHeadroomSwap(float f) { return cdmpool::headroom().exchange(f, std::memory_order_relaxed); }

// ------------------------------------------------------------------------------------------------ context
// CDM_SEGV_BACKTRACE=1 (diagnosis, scripts/stress_kpart.py): a SIGSEGV / SIGBUS / SIGABRT of the process prints the faulting thread's
// native stack (backtrace_symbols_fd: async-signal-safe) before the default action takes its course.
#include <execinfo.h>
#include <signal.h>
static void cdmFaultHandler(int sig, siginfo_t *info, void *) {
    static const char head[] = "\n*** libcarpedeam_hip: fatal signal, native stack of the faulting thread:\n";
    (void) !write(2, head, sizeof(head) - 1);
    void *frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    (void) info;
    signal(sig, SIG_DFL);
    raise(sig);
}
static void cdmInstallFaultHandler() {
    static std::once_flag once;
    std::call_once(once, [] {
        if (!cdmGetenv("CDM_SEGV_BACKTRACE")) return;
        void *warm[4]; (void) backtrace(warm, 4);       // (loads libgcc now, not inside the handler)
        struct sigaction sa; memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = cdmFaultHandler; sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_RESETHAND;
        sigaction(SIGSEGV, &sa, nullptr); sigaction(SIGBUS, &sa, nullptr); sigaction(SIGABRT, &sa, nullptr);
    });
}
extern "C" int cdm_ctx_create(int device, cdm_ctx **out) {
    if (!out) { cdm_set_error("cdm_ctx_create: out is NULL"); return CDM_ERR_INVALID; }
    cdmInstallFaultHandler();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        cdm_set_error("no HIP device available: the carpedeam MI355X path has no CPU fallback");
        return CDM_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) { cdm_set_error("device ordinal %d out of range (%d devices)", device, count); return CDM_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { cdm_set_error("hipGetDeviceProperties failed"); return CDM_ERR_NO_DEVICE; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        cdm_set_error("device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
        return CDM_ERR_NO_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) { cdm_set_error("hipSetDevice(%d) failed", device); return CDM_ERR_NO_DEVICE; }
    cdm_ctx *c = new cdm_ctx();
    c->device = device;
    c->cuCount = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->ev2) != hipSuccess || hipEventCreate(&c->ev3) != hipSuccess ||
        hipEventCreate(&c->evS0) != hipSuccess || hipEventCreate(&c->evS1) != hipSuccess ||
        cdmMalloc(&c->lutDev, sizeof(DamageLut)) != hipSuccess) {
        cdm_set_error("context resource creation failed"); delete c; return CDM_ERR_HIP;
    }
    *out = c;
    return CDM_OK;
}
extern "C" void cdm_ctx_destroy(cdm_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ev2) hipEventDestroy(c->ev2);
    if (c->ev3) hipEventDestroy(c->ev3);
    if (c->evS0) hipEventDestroy(c->evS0);
    if (c->evS1) hipEventDestroy(c->evS1);
    if (c->lutDev) cdmFree(c->lutDev);
    cdmPoolTrim();
    delete c;
}
extern "C" int cdm_ctx_sync(cdm_ctx *c) { CDM_HIP(hipSetDevice(c->device)); CDM_HIP(hipStreamSynchronize(c->stream)); return CDM_OK; }
extern "C" void *cdm_ctx_stream(cdm_ctx *c) { return (void *) c->stream; }
extern "C" float cdm_ctx_last_kernel_ms(cdm_ctx *c, int which) { return (which >= 0 && which < 16) ? c->lastMs[which] : -1.f; }

extern "C" int cdm_damage_load(cdm_ctx *c, const char *prefix) {
    std::string err;
    int rc = cdm_build_damage(prefix, c->mats, &c->lutHost, &err);
    if (rc != CDM_OK) { cdm_set_error("%s", err.c_str()); return rc; }
    CDM_HIP(hipSetDevice(c->device));
    CDM_HIP(hipMemcpyAsync(c->lutDev, &c->lutHost, sizeof(DamageLut), hipMemcpyHostToDevice, c->stream));
    CDM_HIP(hipStreamSynchronize(c->stream));
    c->haveDamage = true;
    return CDM_OK;
}
extern "C" int cdm_damage_get(cdm_ctx *c, long double *out) {
    if (!c->haveDamage) { cdm_set_error("cdm_damage_get: no damage model loaded"); return CDM_ERR_INVALID; }
    memcpy(out, c->mats, sizeof(c->mats));
    return CDM_OK;
}
extern "C" double cdm_evalue(double raw, double qLen, uint64_t dbRes) { return cdm_evalue_host(raw, qLen, dbRes); }
extern "C" int cdm_bit_score(double raw) { return cdm_bit_score_host(raw); }
extern "C" int cdm_gapped_evalue(int gapOpen, int gapExtend, double raw, double qLen, uint64_t dbRes, double *evalue, int *bits) {
    if (!cdm_gapped_costs_known(gapOpen, gapExtend)) { cdm_set_error("gapped E-values: only nucleotide.out with --gap-open 5 --gap-extend 2 (the costs ancient_assemble passes)"); return CDM_ERR_UNSUPPORTED; }
    if (evalue) *evalue = cdm_evalue_gapped_host(raw, qLen, dbRes);
    if (bits) *bits = cdm_bit_score_gapped_host(raw);
    return CDM_OK;
}

// ------------------------------------------------------------------------------------------------ sequence DB
int cdm_seqdb_alloc(cdm_ctx *ctx, uint64_t n, cdm_seqdb **out) {
    static std::atomic<uint64_t> nextSerial{1};
    cdm_seqdb *db = new cdm_seqdb();
    db->serial = nextSerial++;
    db->n = n; db->device = ctx->device;
    if (cdmMalloc(&db->woff, (n + 1) * sizeof(uint32_t)) != hipSuccess || cdmMalloc(&db->len, (n + 1) * sizeof(uint32_t)) != hipSuccess ||
        cdmMalloc(&db->key, (n + 1) * sizeof(uint32_t)) != hipSuccess || cdmMalloc(&db->ext, n + 1) != hipSuccess ||
        cdmMalloc(&db->hasN, n + 8) != hipSuccess) {
        cdm_set_error("out of device memory allocating a %llu-entry sequence DB", (unsigned long long) n);
        cdm_seqdb_free(db); return CDM_ERR_HIP;
    }
    *out = db;
    return CDM_OK;
}
static int seqdb_alloc_codes(cdm_seqdb *db, uint64_t words) {
    db->words = words;
    uint64_t maskWords = (words * 16 + 31) / 32 + 1;
    if (cdmMalloc(&db->codes, (words + 2) * sizeof(uint32_t)) != hipSuccess || cdmMalloc(&db->nmask, maskWords * sizeof(uint32_t)) != hipSuccess) {
        cdm_set_error("out of device memory allocating %llu code words", (unsigned long long) words); return CDM_ERR_HIP;
    }
    return CDM_OK;
}
int cdm_seqdb_alloc_raw(cdm_seqdb *db) {
    if (db->raw) return CDM_OK;
    if (cdmMalloc(&db->raw, db->words * 16 + 16) != hipSuccess) { cdm_set_error("out of device memory allocating the original letters of %llu code words", (unsigned long long) db->words); return CDM_ERR_HIP; }
    return CDM_OK;
}
int cdm_seqdb_alloc_like(cdm_ctx *ctx, const cdm_seqdb *src, cdm_seqdb **out) {
    cdm_seqdb *db = nullptr;
    int rc = cdm_seqdb_alloc(ctx, src->n, &db);
    if (rc) return rc;
    db->residues = src->residues; db->maxLen = src->maxLen; db->nCount = src->nCount;
    rc = seqdb_alloc_codes(db, src->words);
    if (!rc && src->raw) rc = cdm_seqdb_alloc_raw(db);
    if (rc) { cdm_seqdb_free(db); return rc; }
    hipStream_t s = ctx->stream;
    CDM_HIP(hipMemcpyAsync(db->woff, src->woff, (src->n + 1) * 4, hipMemcpyDeviceToDevice, s));
    CDM_HIP(hipMemcpyAsync(db->len, src->len, src->n * 4, hipMemcpyDeviceToDevice, s));
    CDM_HIP(hipMemcpyAsync(db->key, src->key, src->n * 4, hipMemcpyDeviceToDevice, s));
    CDM_HIP(hipMemcpyAsync(db->ext, src->ext, src->n, hipMemcpyDeviceToDevice, s));
    CDM_HIP(hipMemcpyAsync(db->hasN, src->hasN, src->n, hipMemcpyDeviceToDevice, s));
    *out = db;
    return CDM_OK;
}
extern "C" void cdm_seqdb_free(cdm_seqdb *db) {
    if (!db) return;
    hipSetDevice(db->device);
    cdmFree(db->woff); cdmFree(db->len); cdmFree(db->key); cdmFree(db->ext); cdmFree(db->hasN); cdmFree(db->codes); cdmFree(db->nmask); cdmFree(db->raw);
    delete db;
}
__global__ void k_build_meta(const uint32_t *__restrict__ woff, const uint32_t *__restrict__ len, const uint8_t *__restrict__ hasN, const uint8_t *__restrict__ ext,
                             const uint32_t *__restrict__ key, uint32_t n, SeqMeta *__restrict__ out, uint32_t uniLen, uint32_t uniWords, unsigned int *__restrict__ notUniform) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SeqMeta m; m.woff = woff[i]; m.len = len[i]; m.flags = (hasN[i] ? 1u : 0u) | (ext[i] ? 2u : 0u) | ((hasN[i] & 2u) ? 4u : 0u); m.key = key[i];
    out[i] = m;
    if (notUniform && (m.len != uniLen || m.woff != i * uniWords || m.flags != 0u)) atomicOr(notUniform, 1u);
}
// CDM_META_UNIFORM=0: the records for every DB (A/B)
int cdm_build_meta(cdm_ctx *ctx, const cdm_seqdb *db, SeqMeta **out, MetaUniform *uniform) {
    SeqMeta *m = nullptr;
    if (cdmMalloc(&m, (db->n + 1) * sizeof(SeqMeta)) != hipSuccess) { cdm_set_error("out of device memory (sequence metadata)"); return CDM_ERR_HIP; }
    // a candidate for the plain uniform form: the lengths sum to n x the longest, no N counted, no row of original letters
    const char *sw = cdmGetenv("CDM_META_UNIFORM");
    const bool candidate = uniform && db->n && db->maxLen && db->residues == db->n * (uint64_t) db->maxLen && db->nCount == 0 && !db->raw && !(sw && !strcmp(sw, "0")) &&
                           db->n * (uint64_t) ((db->maxLen + 15) / 16) < (1ull << 32);
    DevBuf<unsigned int> bad;
    if (candidate) { if (!bad.alloc(1)) { cdmFree(m); cdm_set_error("out of device memory (sequence metadata)"); return CDM_ERR_HIP; } hipMemsetAsync(bad.p, 0, 4, ctx->stream); }
    if (db->n) hipLaunchKernelGGL(k_build_meta, dim3((unsigned) ((db->n + 255) / 256)), dim3(256), 0, ctx->stream, db->woff, db->len, db->hasN, db->ext, db->key, (uint32_t) db->n, m,
                                  db->maxLen, (db->maxLen + 15) / 16, candidate ? bad.p : nullptr);
    if (uniform) { uniform->words = 0; uniform->len = 0; }
    if (candidate) {
        unsigned int h = 1;
        if (hipMemcpyAsync(&h, bad.p, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { cdmFree(m); cdm_set_error("sequence metadata: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
        if (!h) { uniform->words = (db->maxLen + 15) / 16; uniform->len = db->maxLen; }
    }
    *out = m;
    return CDM_OK;
}

extern "C" uint64_t cdm_seqdb_size(const cdm_seqdb *db) { return db->n; }
extern "C" uint64_t cdm_seqdb_residues(const cdm_seqdb *db) { return db->residues; }
extern "C" uint32_t cdm_seqdb_max_len(const cdm_seqdb *db) { return db->maxLen; }

// one thread per (sequence, word): 16 ASCII letters -> one code word + 16 N bits
__global__ void k_pack(const char *__restrict__ data, const uint64_t *__restrict__ off, const uint32_t *__restrict__ len,
                       const uint32_t *__restrict__ woff, uint64_t n, uint64_t words, uint32_t *__restrict__ codes,
                       uint32_t *__restrict__ nmask, uint8_t *__restrict__ hasN, unsigned long long *__restrict__ counters) {
    uint64_t gw = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (gw >= words) return;
    // sequence that owns word gw: last i with woff[i] <= gw
    uint64_t lo = 0, hi = n;
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (woff[mid] <= gw) lo = mid; else hi = mid; }
    const uint64_t i = lo;
    const uint32_t w = (uint32_t) (gw - woff[i]);
    const uint32_t L = len[i];
    const char *s = data + off[i] + (uint64_t) w * 16;
    const uint32_t cnt = min(16u, L - min(L, w * 16u));
    uint32_t code = 0, nb = 0, other = 0;
    for (uint32_t j = 0; j < cnt; j++) {
        const char c = s[j];
        uint32_t v = 0;
        switch (c) {
            case 'A': v = 0; break; case 'C': v = 1; break; case 'G': v = 2; break; case 'T': v = 3; break;
            case 'N': nb |= 1u << j; break;
            default:            // NucleotideMatrix::setupLetterMapping (M/commons/NucleotideMatrix.cpp:17-61): toupper, IUPAC codes to one base, the rest to X
                other++;
                switch (c & ~0x20) {    // (the letters below have no non-letter twin under the case bit)
                    case 'A': v = 0; break; case 'C': case 'M': case 'Y': case 'H': v = 1; break;
                    case 'G': case 'K': case 'B': case 'D': case 'V': case 'R': case 'S': v = 2; break;
                    case 'T': case 'U': case 'W': v = 3; break;
                    default: nb |= 1u << j; break;
                }
                if (!((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))) { v = 0; nb |= 1u << j; }
                break;
        }
        code |= v << (2 * j);
    }
    codes[gw] = code;
    // two sequence words share one mask word; sequences start on code-word (16 bit) boundaries of the mask
    uint16_t *m16 = reinterpret_cast<uint16_t *>(nmask);
    m16[gw] = (uint16_t) nb;
    if (nb) atomicAdd(&counters[0], (unsigned long long) __popc(nb));
    if (nb || other) atomicOr(reinterpret_cast<unsigned int *>(hasN + (i & ~3ull)), (1u | (other ? 2u : 0u)) << (8 * (i & 3u)));
    if (other) atomicAdd(&counters[1], (unsigned long long) other);
}
// the original bytes of the sequences that carry letters beyond ACGTN, one thread per (sequence, word)
__global__ void k_pack_raw(const char *__restrict__ data, const uint64_t *__restrict__ off, const uint32_t *__restrict__ len,
                           const uint32_t *__restrict__ woff, uint64_t n, uint64_t words, const uint8_t *__restrict__ hasN, uint8_t *__restrict__ raw) {
    uint64_t gw = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (gw >= words) return;
    uint64_t lo = 0, hi = n;
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (woff[mid] <= gw) lo = mid; else hi = mid; }
    const uint64_t i = lo;
    if (!(hasN[i] & 2u)) return;
    const uint32_t w = (uint32_t) (gw - woff[i]);
    const uint32_t L = len[i];
    const char *s = data + off[i] + (uint64_t) w * 16;
    const uint32_t cnt = min(16u, L - min(L, w * 16u));
    for (uint32_t j = 0; j < cnt; j++) raw[gw * 16 + j] = (uint8_t) s[j];
}

extern "C" int cdm_seqdb_upload(cdm_ctx *ctx, const char *data, const uint64_t *offsets, const uint32_t *lengths, const uint32_t *keys,
                                const uint8_t *ext, uint64_t n, cdm_seqdb **out) {
    if (!ctx || !data || !offsets || !lengths || !keys || !out) { cdm_set_error("cdm_seqdb_upload: NULL argument"); return CDM_ERR_INVALID; }
    if (n == 0) { cdm_set_error("cdm_seqdb_upload: empty sequence DB"); return CDM_ERR_INVALID; }
    if (n >= 0xFFFFFFFFull) { cdm_set_error("cdm_seqdb_upload: more than 2^32-1 sequences"); return CDM_ERR_UNSUPPORTED; }
    CDM_HIP(hipSetDevice(ctx->device));
    std::vector<uint32_t> woff(n + 1);
    uint64_t words = 0, residues = 0, lo = UINT64_MAX, hi = 0; uint32_t maxLen = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (i && keys[i] <= keys[i - 1]) { cdm_set_error("cdm_seqdb_upload: keys must be strictly increasing (entry %llu)", (unsigned long long) i); return CDM_ERR_INVALID; }
        woff[i] = (uint32_t) words;
        words += (lengths[i] + 15) / 16;
        residues += lengths[i];
        maxLen = std::max(maxLen, lengths[i]);
        lo = std::min(lo, offsets[i]); hi = std::max(hi, offsets[i] + lengths[i]);
        if (words >= 0xFFFFFF00ull) { cdm_set_error("cdm_seqdb_upload: more than 2^32 code words (68 G bases) in one DB"); return CDM_ERR_UNSUPPORTED; }
    }
    woff[n] = (uint32_t) words;
    cdm_seqdb *db = nullptr;
    int rc = cdm_seqdb_alloc(ctx, n, &db);
    if (rc) return rc;
    db->residues = residues; db->maxLen = maxLen;
    rc = seqdb_alloc_codes(db, words);
    if (rc) { cdm_seqdb_free(db); return rc; }
    hipStream_t s = ctx->stream;
    char *dData = nullptr; uint64_t *dOff = nullptr; unsigned long long *dCnt = nullptr;
    std::vector<uint64_t> rel(n);
    for (uint64_t i = 0; i < n; i++) rel[i] = offsets[i] - lo;
    int ret = CDM_OK;
    do {
        if (cdmMalloc(&dData, hi - lo + 16) != hipSuccess || cdmMalloc(&dOff, n * 8) != hipSuccess || cdmMalloc(&dCnt, 16) != hipSuccess) {
            cdm_set_error("out of device memory staging %llu bytes of sequence text", (unsigned long long) (hi - lo)); ret = CDM_ERR_HIP; break;
        }
        hipMemsetAsync(dCnt, 0, 16, s);
        hipMemsetAsync(db->hasN, 0, n, s);
        hipMemcpyAsync(dData, data + lo, hi - lo, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(dOff, rel.data(), n * 8, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(db->woff, woff.data(), (n + 1) * 4, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(db->len, lengths, n * 4, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(db->key, keys, n * 4, hipMemcpyHostToDevice, s);
        if (ext) hipMemcpyAsync(db->ext, ext, n, hipMemcpyHostToDevice, s); else hipMemsetAsync(db->ext, 0, n, s);
        if (words) {
            hipLaunchKernelGGL(k_pack, CDM_GRID((words + 255) / 256, 256), dim3(256), 0, s, dData, dOff, db->len, db->woff, n, words, db->codes,
                               db->nmask, db->hasN, dCnt);
        }
        unsigned long long cnt[2] = {0, 0};
        hipMemcpyAsync(cnt, dCnt, 16, hipMemcpyDeviceToHost, s);
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) { cdm_set_error("sequence upload/packing failed: %s", hipGetErrorString(e)); ret = CDM_ERR_HIP; break; }
        db->nCount = cnt[0];
        if (cnt[1]) {       // lower case / IUPAC codes / other bytes: those sequences keep their original letters beside the mapped codes
            if ((ret = cdm_seqdb_alloc_raw(db)) != CDM_OK) break;
            hipLaunchKernelGGL(k_pack_raw, CDM_GRID((words + 255) / 256, 256), dim3(256), 0, s, dData, dOff, db->len, db->woff, n, words, db->hasN, db->raw);
            e = hipStreamSynchronize(s);
            if (e != hipSuccess) { cdm_set_error("sequence upload/packing failed: %s", hipGetErrorString(e)); ret = CDM_ERR_HIP; break; }
        }
    } while (0);
    cdmFree(dData); cdmFree(dOff); cdmFree(dCnt);
    if (ret != CDM_OK) { cdm_seqdb_free(db); return ret; }
    *out = db;
    return CDM_OK;
}

extern "C" int cdm_seqdb_meta(cdm_ctx *ctx, const cdm_seqdb *db, uint32_t *lengths, uint32_t *keys, uint8_t *ext) {
    CDM_HIP(hipSetDevice(ctx->device));
    if (lengths) CDM_HIP(hipMemcpyAsync(lengths, db->len, db->n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (keys) CDM_HIP(hipMemcpyAsync(keys, db->key, db->n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ext) CDM_HIP(hipMemcpyAsync(ext, db->ext, db->n, hipMemcpyDeviceToHost, ctx->stream));
    CDM_HIP(hipStreamSynchronize(ctx->stream));
    return CDM_OK;
}

// one thread per (sequence, word): 16 letters + the trailing '\n' after the last base
__global__ void k_unpack(const uint32_t *__restrict__ codes, const uint32_t *__restrict__ nmask, const uint32_t *__restrict__ woff,
                         const uint32_t *__restrict__ len, const uint64_t *__restrict__ outOff, uint64_t n, uint64_t words, char *__restrict__ out,
                         const uint8_t *__restrict__ hasN, const uint8_t *__restrict__ raw) {
    uint64_t gw = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (gw >= words) return;
    uint64_t lo = 0, hi = n;
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (woff[mid] <= gw) lo = mid; else hi = mid; }
    const uint64_t i = lo;
    const uint32_t w = (uint32_t) (gw - woff[i]);
    const uint32_t L = len[i];
    const uint32_t cnt = min(16u, L - min(L, w * 16u));
    const uint32_t code = codes[gw];
    const uint32_t nb = reinterpret_cast<const uint16_t *>(nmask)[gw];
    char *o = out + outOff[i] + (uint64_t) w * 16;
    if (raw && (hasN[i] & 2u)) for (uint32_t j = 0; j < cnt; j++) o[j] = (char) raw[gw * 16 + j];
    else for (uint32_t j = 0; j < cnt; j++) o[j] = ((nb >> j) & 1u) ? 'N' : "ACGT"[(code >> (2 * j)) & 3u];
    if (w * 16u + cnt == L) o[cnt] = '\n';
}
extern "C" int cdm_seqdb_download(cdm_ctx *ctx, const cdm_seqdb *db, char *out, const uint64_t *outOffsets) {
    CDM_HIP(hipSetDevice(ctx->device));
    std::vector<uint32_t> len(db->n);
    CDM_HIP(hipMemcpy(len.data(), db->len, db->n * 4, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (uint64_t i = 0; i < db->n; i++) total = std::max(total, outOffsets[i] + len[i] + 1);
    char *dOut = nullptr; uint64_t *dOff = nullptr;
    if (cdmMalloc(&dOut, total + 16) != hipSuccess || cdmMalloc(&dOff, db->n * 8) != hipSuccess) { cdmFree(dOut); cdm_set_error("out of device memory in cdm_seqdb_download"); return CDM_ERR_HIP; }
    hipMemsetAsync(dOut, 0, total, ctx->stream);
    hipMemcpyAsync(dOff, outOffsets, db->n * 8, hipMemcpyHostToDevice, ctx->stream);
    // zero-length sequences own no word: their '\n' is written by the host below
    if (db->words) hipLaunchKernelGGL(k_unpack, CDM_GRID((db->words + 255) / 256, 256), dim3(256), 0, ctx->stream, db->codes, db->nmask, db->woff, db->len, dOff, db->n, db->words, dOut, db->hasN, db->raw);
    hipMemcpyAsync(out, dOut, total, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    cdmFree(dOut); cdmFree(dOff);
    if (e != hipSuccess) { cdm_set_error("cdm_seqdb_download failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; }
    for (uint64_t i = 0; i < db->n; i++) if (len[i] == 0) out[outOffsets[i]] = '\n';
    return CDM_OK;
}
// The same blob in PIECES, for a caller that writes it out as it comes (round 5: a module process's sequence DB - the copy from the
// device lands in pinned staging buffers of the library at the link's speed, and the caller's sink - a pwrite into the DB's data file -
// runs on piece i while piece i + 1 is on its way; before, the whole text came down into pageable memory at 8 GB/s and was then written
// out at the file system's 4 GB/s, one after the other).  sink(user, data, offset, bytes): consecutive pieces, data valid during the
// call; a non-zero return ends the download with CDM_ERR_INVALID.
extern "C" int cdm_seqdb_download_stream(cdm_ctx *ctx, const cdm_seqdb *db, const uint64_t *outOffsets, uint64_t pieceBytes,
                                         int (*sink)(void *user, const char *data, uint64_t offset, uint64_t bytes), void *user) {
    if (!ctx || !db || !sink || (db->n && !outOffsets)) { cdm_set_error("cdm_seqdb_download_stream: invalid argument"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    if (db->n == 0) return CDM_OK;
    if (pieceBytes < (1u << 20)) pieceBytes = 64u << 20;
    std::vector<uint32_t> len(db->n);
    CDM_HIP(hipMemcpy(len.data(), db->len, db->n * 4, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (uint64_t i = 0; i < db->n; i++) total = std::max(total, outOffsets[i] + len[i] + 1);
    // (the entries must ascend for the zero-length fix-up below to find them piece by piece; every caller's do)
    for (uint64_t i = 1; i < db->n; i++) if (outOffsets[i] < outOffsets[i - 1]) { cdm_set_error("cdm_seqdb_download_stream: the offsets must ascend"); return CDM_ERR_INVALID; }
    DevBuf<char> dOut; DevBuf<uint64_t> dOff;
    if (!dOut.alloc(total + 16) || !dOff.alloc(db->n)) { cdm_set_error("out of device memory in cdm_seqdb_download_stream"); return CDM_ERR_HIP; }
    hipStream_t s = ctx->stream;
    hipMemsetAsync(dOut.p, 0, total, s);
    hipMemcpyAsync(dOff.p, outOffsets, db->n * 8, hipMemcpyHostToDevice, s);
    if (db->words) hipLaunchKernelGGL(k_unpack, CDM_GRID((db->words + 255) / 256, 256), dim3(256), 0, s, db->codes, db->nmask, db->woff, db->len, dOff.p, db->n, db->words, dOut.p, db->hasN, db->raw);
    char *pin[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr};
    auto cleanup = [&] { for (int b = 0; b < 2; b++) { if (pin[b]) (void) hipHostFree(pin[b]); if (ev[b]) (void) hipEventDestroy(ev[b]); } };
    pieceBytes = std::min(pieceBytes, total);
    for (int b = 0; b < 2; b++) if (hipHostMalloc((void **) &pin[b], pieceBytes, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) != hipSuccess) {
        (void) hipGetLastError(); cleanup(); cdm_set_error("cdm_seqdb_download_stream: no pinned staging buffer of %llu bytes", (unsigned long long) pieceBytes); return CDM_ERR_HIP;
    }
    const uint64_t pieces = (total + pieceBytes - 1) / pieceBytes;
    auto issue = [&](uint64_t i) { const uint64_t at = i * pieceBytes, nb = std::min(pieceBytes, total - at); hipMemcpyAsync(pin[i & 1], dOut.p + at, nb, hipMemcpyDeviceToHost, s); hipEventRecord(ev[i & 1], s); };
    issue(0);
    uint64_t z = 0;             // next sequence to look at for the zero-length fix-up ('\n' of an empty sequence: it owns no code word)
    int rc = CDM_OK;
    for (uint64_t i = 0; i < pieces && rc == CDM_OK; i++) {
        if (i + 1 < pieces) issue(i + 1);
        if (hipEventSynchronize(ev[i & 1]) != hipSuccess) { cdm_set_error("cdm_seqdb_download_stream failed: %s", hipGetErrorString(hipGetLastError())); rc = CDM_ERR_HIP; break; }
        const uint64_t at = i * pieceBytes, nb = std::min(pieceBytes, total - at);
        while (z < db->n && outOffsets[z] < at + nb) { if (len[z] == 0 && outOffsets[z] >= at) pin[i & 1][outOffsets[z] - at] = '\n'; z++; }
        if (sink(user, pin[i & 1], at, nb) != 0) { cdm_set_error("cdm_seqdb_download_stream: the sink refused a piece at offset %llu", (unsigned long long) at); rc = CDM_ERR_INVALID; }
    }
    (void) hipStreamSynchronize(s);
    cleanup();
    return rc;
}
extern "C" int cdm_seqdb_synth(cdm_ctx *ctx, uint64_t nTotal, uint64_t first, uint64_t n, uint32_t lo, uint32_t hi, uint64_t seed, cdm_seqdb **out) {
    return cdm_synth_impl(ctx, nTotal, first, n, lo, hi, seed, out);
}

// ------------------------------------------------------------------------------------------------ multi-GPU hand-off
#include "scan.h"
namespace {
// sum and maximum of the lengths, on the device (the host needs two numbers of a DB it composed, not its 50 M lengths)
__global__ __launch_bounds__(256) void k_len_stats(const uint32_t *__restrict__ len, uint64_t n, unsigned long long *__restrict__ out) {
    unsigned long long sum = 0; unsigned int mx = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) { const uint32_t v = len[i]; sum += v; mx = max(mx, v); }
    sum = cdm_block_sum<unsigned long long>(sum);
    __shared__ unsigned int sMax;
    if (threadIdx.x == 0) sMax = 0;
    __syncthreads();
    atomicMax(&sMax, mx);
    __syncthreads();
    if (threadIdx.x == 0) { atomicAdd(&out[0], sum); atomicMax(&out[1], (unsigned long long) sMax); }
}
// residues / maxLen of db from its device lengths (enqueued on s; the caller synchronises, then calls the returned setter... kept simple: synchronous)
int seqdbLenStats(cdm_ctx *ctx, cdm_seqdb *db) {
    hipStream_t s = ctx->stream;
    DevBuf<unsigned long long> d; unsigned long long h[2] = {0, 0};
    if (!d.alloc(2)) { cdm_set_error("out of device memory"); return CDM_ERR_HIP; }
    CDM_HIP(hipMemsetAsync(d.p, 0, 16, s));
    if (db->n) hipLaunchKernelGGL(k_len_stats, dim3((unsigned) std::min<uint64_t>((db->n + 255) / 256, 4096)), dim3(256), 0, s, (const uint32_t *) db->len, (uint64_t) db->n, d.p);
    CDM_HIP(hipMemcpyAsync(h, d.p, 16, hipMemcpyDeviceToHost, s));
    CDM_HIP(hipStreamSynchronize(s));
    db->residues = h[0]; db->maxLen = (uint32_t) h[1];
    return CDM_OK;
}
__global__ void k_sel_from_ext(const uint32_t *__restrict__ len, const uint8_t *__restrict__ ext, uint32_t n, uint32_t *__restrict__ sel) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sel[i] = ext[i] == 1 ? len[i] : 0xFFFFFFFFu;
}
__global__ void k_sel_words(const uint32_t *__restrict__ sel, uint32_t n, uint32_t *__restrict__ selWords, uint32_t *__restrict__ selOne) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const bool s = i < n && sel[i] != 0xFFFFFFFFu;
    selWords[i] = s ? (sel[i] + 15) / 16 : 0; selOne[i] = s ? 1 : 0;
}
__global__ void k_sel_meta(const cdm_seqdb src, const uint32_t *__restrict__ sel, const uint32_t *__restrict__ rank, const uint32_t *__restrict__ wordOff, uint32_t n, int extValue, cdm_seqdb dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || sel[i] == 0xFFFFFFFFu) return;
    const uint32_t r = rank[i];
    dst.len[r] = sel[i]; dst.key[r] = src.key[i]; dst.ext[r] = extValue < 0 ? src.ext[i] : (uint8_t) extValue; dst.hasN[r] = (src.hasN[i] & 2u) ? 3 : 0; dst.woff[r] = wordOff[i];
}
__global__ void k_sel_copy(const cdm_seqdb src, const uint32_t *__restrict__ sel, const uint32_t *__restrict__ wordOff, uint32_t n, uint32_t first, cdm_seqdb dst) {
    // one wave per selected sequence (of this launch's slice, from `first` on); the kept prefix ends inside its last word: the letters behind it are cleared
    const uint32_t i = first + (uint32_t) (((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (i >= n || sel[i] == 0xFFFFFFFFu) return;
    const uint32_t L = sel[i], w = (L + 15) / 16, s0 = src.woff[i], d0 = wordOff[i], tail = L & 15u;
    for (uint32_t j = lane; j < w; j += 64) {
        uint32_t c = src.codes[s0 + j], m = reinterpret_cast<const uint16_t *>(src.nmask)[s0 + j];
        if (j == w - 1 && tail) { c &= (1u << (2 * tail)) - 1u; m &= (1u << tail) - 1u; }
        dst.codes[d0 + j] = c;
        reinterpret_cast<uint16_t *>(dst.nmask)[d0 + j] = (uint16_t) m;
    }
    if (src.hasN[i] & 2u) for (uint32_t j = lane; j < L; j += 64) dst.raw[(uint64_t) d0 * 16 + j] = src.raw[(uint64_t) s0 * 16 + j];
}
__global__ void k_words_of(const uint32_t *__restrict__ len, uint32_t n, uint32_t *__restrict__ w, uint8_t *__restrict__ ext, uint8_t extValue, uint8_t *__restrict__ hasN) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    w[i] = i < n ? (len[i] + 15) / 16 : 0;
    if (i < n) { ext[i] = extValue; hasN[i] = 0; }
}
__global__ void k_mark_hasN(const uint32_t *__restrict__ woff, const uint32_t *__restrict__ nmask, uint32_t n, uint64_t words, uint8_t *__restrict__ hasN) {
    const uint64_t gw = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (gw >= words || reinterpret_cast<const uint16_t *>(nmask)[gw] == 0) return;
    uint64_t lo = 0, hi = n;
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (woff[mid] <= gw) lo = mid; else hi = mid; }
    if (!hasN[lo]) hasN[lo] = 1;      // (a sequence that brought its raw row along keeps its 3)
}
}  // namespace
extern "C" uint64_t cdm_seqdb_words(const cdm_seqdb *db) { return db->words; }
// Sub-DB: sel[i] = 0xFFFFFFFF drops sequence i, any other value keeps its first sel[i] letters (<= len[i]); extValue < 0 keeps the
// wasExtended flags.  The order of the kept sequences is kept.
int cdm_seqdb_select(cdm_ctx *ctx, const cdm_seqdb *db, const uint32_t *sel, int extValue, cdm_seqdb **out) {
    CDM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t n = (uint32_t) db->n;
    DevBuf<uint32_t> selWords, selOne, wordOff, rank;
    if (!selWords.alloc((size_t) n + 1) || !selOne.alloc((size_t) n + 1) || !wordOff.alloc((size_t) n + 1) || !rank.alloc((size_t) n + 1)) { cdm_set_error("cdm_seqdb_select: out of device memory"); return CDM_ERR_HIP; }
    hipLaunchKernelGGL(k_sel_words, dim3((n + 256) / 256), dim3(256), 0, s, sel, n, selWords.p, selOne.p);
    cdmscan::ScanTemp st1, st2;
    if (cdmscan::exclusiveScan<uint32_t>(s, st1, selWords.p, wordOff.p, (size_t) n + 1) != CDM_OK || cdmscan::exclusiveScan<uint32_t>(s, st2, selOne.p, rank.p, (size_t) n + 1) != CDM_OK) return CDM_ERR_HIP;
    uint32_t m = 0, words = 0;
    hipMemcpyAsync(&m, rank.p + n, 4, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(&words, wordOff.p + n, 4, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_seqdb_select failed"); return CDM_ERR_HIP; }
    cdm_seqdb *o = nullptr;
    int rc = cdm_seqdb_alloc(ctx, m, &o);
    if (rc == CDM_OK) rc = seqdb_alloc_codes(o, words);
    if (rc == CDM_OK && db->raw) rc = cdm_seqdb_alloc_raw(o);
    if (rc != CDM_OK) { if (o) cdm_seqdb_free(o); return rc; }
    hipMemsetAsync(o->nmask, 0, (((uint64_t) words * 16 + 31) / 32 + 1) * 4, s);
    if (n) hipLaunchKernelGGL(k_sel_meta, dim3((n + 255) / 256), dim3(256), 0, s, *db, sel, rank.p, wordOff.p, n, extValue, *o);
    for (uint64_t first = 0, slice = cdmSliceItems(64); first < n; first += slice)
        hipLaunchKernelGGL(k_sel_copy, CDM_GRID((std::min<uint64_t>(slice, n - first) * 64 + 255) / 256, 256), dim3(256), 0, s, *db, sel, wordOff.p, n, (uint32_t) first, *o);
    hipMemcpyAsync(o->woff + m, &words, 4, hipMemcpyHostToDevice, s);
    if (words && m) hipLaunchKernelGGL(k_mark_hasN, CDM_GRID(((uint64_t) words + 255) / 256, 256), dim3(256), 0, s, o->woff, o->nmask, m, (uint64_t) words, o->hasN);
    if (int rc2 = seqdbLenStats(ctx, o)) { cdm_seqdb_free(o); cdm_set_error("cdm_seqdb_select: %s", hipGetErrorString(hipGetLastError())); return rc2; }
    *out = o;
    return CDM_OK;
}
// ---- overlay: base with some of its sequences replaced (ancient_contig_merge: the grown contigs come up from the host, the others
// never leave the device).  out[i] = grown[j] where idx[j] == i, else base[i]; keys are base's, ext comes from the caller.
namespace {
__global__ void k_ov_source(const uint32_t *__restrict__ idx, uint32_t m, uint32_t *__restrict__ src) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) src[idx[j]] = j;
}
__global__ void k_ov_words(const uint32_t *__restrict__ baseLen, const uint32_t *__restrict__ grownLen, const uint32_t *__restrict__ src, uint32_t n, uint32_t *__restrict__ w) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    w[i] = i < n ? ((src[i] == 0xFFFFFFFFu ? baseLen[i] : grownLen[src[i]]) + 15) / 16 : 0;
}
__global__ void k_ov_meta(const cdm_seqdb base, const cdm_seqdb grown, const uint32_t *__restrict__ src, const uint32_t *__restrict__ wordOff, const uint8_t *__restrict__ ext, uint32_t n, cdm_seqdb dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = src[i];
    dst.len[i] = j == 0xFFFFFFFFu ? base.len[i] : grown.len[j];
    dst.hasN[i] = j == 0xFFFFFFFFu ? base.hasN[i] : grown.hasN[j];
    dst.key[i] = base.key[i]; dst.ext[i] = ext[i]; dst.woff[i] = wordOff[i];
}
__global__ void k_ov_copy(const cdm_seqdb base, const cdm_seqdb grown, const uint32_t *__restrict__ src, const uint32_t *__restrict__ wordOff, uint32_t n, uint32_t first, cdm_seqdb dst) {
    const uint32_t i = first + (uint32_t) (((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;      // one wave per sequence of this launch's slice
    if (i >= n) return;
    const uint32_t j = src[i];
    const cdm_seqdb &from = j == 0xFFFFFFFFu ? base : grown;
    const uint32_t k = j == 0xFFFFFFFFu ? i : j;
    const uint32_t L = from.len[k], w = (L + 15) / 16, s0 = from.woff[k], d0 = wordOff[i];
    for (uint32_t x = lane; x < w; x += 64) {
        dst.codes[d0 + x] = from.codes[s0 + x];
        reinterpret_cast<uint16_t *>(dst.nmask)[d0 + x] = reinterpret_cast<const uint16_t *>(from.nmask)[s0 + x];
    }
    if (from.hasN[k] & 2u) for (uint32_t x = lane; x < L; x += 64) dst.raw[(uint64_t) d0 * 16 + x] = from.raw[(uint64_t) s0 * 16 + x];
}
}  // namespace
int cdm_seqdb_overlay(cdm_ctx *ctx, const cdm_seqdb *base, const cdm_seqdb *grown, const uint32_t *idxHost, const uint8_t *extHost, cdm_seqdb **out) {
    CDM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t n = (uint32_t) base->n, m = grown ? (uint32_t) grown->n : 0u;
    DevBuf<uint32_t> src, idx, w, wordOff; DevBuf<uint8_t> ext;
    if (!src.alloc(n) || !idx.alloc(m) || !w.alloc((size_t) n + 1) || !wordOff.alloc((size_t) n + 1) || !ext.alloc(n)) { cdm_set_error("cdm_seqdb_overlay: out of device memory"); return CDM_ERR_HIP; }
    CDM_HIP(hipMemsetAsync(src.p, 0xFF, (size_t) n * 4, s));
    if (m) CDM_HIP(hipMemcpyAsync(idx.p, idxHost, (size_t) m * 4, hipMemcpyHostToDevice, s));
    CDM_HIP(hipMemcpyAsync(ext.p, extHost, n, hipMemcpyHostToDevice, s));
    if (m) hipLaunchKernelGGL(k_ov_source, dim3((m + 255) / 256), dim3(256), 0, s, idx.p, m, src.p);
    const cdm_seqdb &g = grown ? *grown : *base;
    hipLaunchKernelGGL(k_ov_words, dim3((n + 256) / 256), dim3(256), 0, s, base->len, g.len, src.p, n, w.p);
    cdmscan::ScanTemp st;
    if (cdmscan::exclusiveScan<uint32_t>(s, st, w.p, wordOff.p, (size_t) n + 1) != CDM_OK) return CDM_ERR_HIP;
    if ((uint64_t) base->words + (grown ? grown->words : 0) >= 0xFFFFFF00ull) { cdm_set_error("cdm_seqdb_overlay: more than 2^32 code words (68 G bases) in one DB"); return CDM_ERR_UNSUPPORTED; }
    uint32_t words = 0;
    CDM_HIP(hipMemcpyAsync(&words, wordOff.p + n, 4, hipMemcpyDeviceToHost, s));
    CDM_HIP(hipStreamSynchronize(s));
    cdm_seqdb *o = nullptr;
    int rc = cdm_seqdb_alloc(ctx, n, &o);
    if (rc == CDM_OK) rc = seqdb_alloc_codes(o, words);
    if (rc == CDM_OK && (base->raw || (grown && grown->raw))) rc = cdm_seqdb_alloc_raw(o);
    if (rc != CDM_OK) { if (o) cdm_seqdb_free(o); return rc; }
    hipMemsetAsync(o->nmask, 0, (((uint64_t) words * 16 + 31) / 32 + 1) * 4, s);
    if (n) hipLaunchKernelGGL(k_ov_meta, dim3((n + 255) / 256), dim3(256), 0, s, *base, g, src.p, wordOff.p, ext.p, n, *o);
    for (uint64_t first = 0, slice = cdmSliceItems(64); first < n; first += slice)
        hipLaunchKernelGGL(k_ov_copy, CDM_GRID((std::min<uint64_t>(slice, n - first) * 64 + 255) / 256, 256), dim3(256), 0, s, *base, g, src.p, wordOff.p, n, (uint32_t) first, *o);
    hipMemcpyAsync(o->woff + n, &words, 4, hipMemcpyHostToDevice, s);
    if (int rc2 = seqdbLenStats(ctx, o)) { cdm_seqdb_free(o); cdm_set_error("cdm_seqdb_overlay: %s", hipGetErrorString(hipGetLastError())); return rc2; }
    o->nCount = base->nCount + (grown ? grown->nCount : 0);
    *out = o;
    return CDM_OK;
}
extern "C" int cdm_seqdb_select_ext(cdm_ctx *ctx, const cdm_seqdb *db, cdm_seqdb **out) {
    CDM_HIP(hipSetDevice(ctx->device));
    const uint32_t n = (uint32_t) db->n;
    DevBuf<uint32_t> sel;
    if (!sel.alloc((size_t) n + 1)) { cdm_set_error("cdm_seqdb_select_ext: out of device memory"); return CDM_ERR_HIP; }
    if (n) hipLaunchKernelGGL(k_sel_from_ext, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, db->len, db->ext, n, sel.p);
    return cdm_seqdb_select(ctx, db, sel.p, 1, out);
}
namespace {
__global__ void k_raw_flags(const uint8_t *__restrict__ hasN, uint64_t n, uint8_t *__restrict__ flags) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (hasN[i] & 2u) ? 1 : 0;
}
__global__ void k_raw_attach(const uint8_t *__restrict__ flags, uint64_t n, uint8_t *__restrict__ hasN) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags[i]) hasN[i] = 3;
}
}  // namespace
extern "C" int cdm_seqdb_has_raw(const cdm_seqdb *db) { return db->raw ? 1 : 0; }
extern "C" int cdm_seqdb_copy_raw(cdm_ctx *ctx, const cdm_seqdb *db, void *raw, void *flags) {
    if (!db->raw) { cdm_set_error("cdm_seqdb_copy_raw: the DB has no letters beyond ACGTN"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    CDM_HIP(hipMemcpyAsync(raw, db->raw, db->words * 16, hipMemcpyDeviceToDevice, s));
    if (db->n) hipLaunchKernelGGL(k_raw_flags, dim3((unsigned) ((db->n + 255) / 256)), dim3(256), 0, s, db->hasN, db->n, (uint8_t *) flags);
    CDM_HIP(hipStreamSynchronize(s));
    return CDM_OK;
}
extern "C" int cdm_seqdb_attach_raw(cdm_ctx *ctx, cdm_seqdb *db, const void *raw, const void *flags) {
    CDM_HIP(hipSetDevice(ctx->device));
    if (int rc = cdm_seqdb_alloc_raw(db)) return rc;
    hipStream_t s = ctx->stream;
    CDM_HIP(hipMemcpyAsync(db->raw, raw, db->words * 16, hipMemcpyDeviceToDevice, s));
    if (db->n) hipLaunchKernelGGL(k_raw_attach, dim3((unsigned) ((db->n + 255) / 256)), dim3(256), 0, s, (const uint8_t *) flags, db->n, db->hasN);
    CDM_HIP(hipStreamSynchronize(s));
    return CDM_OK;
}
extern "C" int cdm_seqdb_copy_packed(cdm_ctx *ctx, const cdm_seqdb *db, void *codes, void *nmask16, void *lengths, void *keys) {
    CDM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if (codes) CDM_HIP(hipMemcpyAsync(codes, db->codes, db->words * 4, hipMemcpyDeviceToDevice, s));
    if (nmask16) CDM_HIP(hipMemcpyAsync(nmask16, db->nmask, db->words * 2, hipMemcpyDeviceToDevice, s));
    if (lengths) CDM_HIP(hipMemcpyAsync(lengths, db->len, db->n * 4, hipMemcpyDeviceToDevice, s));
    if (keys) CDM_HIP(hipMemcpyAsync(keys, db->key, db->n * 4, hipMemcpyDeviceToDevice, s));
    CDM_HIP(hipStreamSynchronize(s));
    return CDM_OK;
}
extern "C" int cdm_seqdb_from_packed(cdm_ctx *ctx, const void *codes, const void *nmask16, const void *lengths, const void *keys, uint64_t n, uint64_t words,
                                     uint8_t extValue, cdm_seqdb **out) {
    CDM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    cdm_seqdb *o = nullptr;
    int rc = cdm_seqdb_alloc(ctx, n, &o);
    if (rc == CDM_OK) rc = seqdb_alloc_codes(o, words);
    if (rc != CDM_OK) { if (o) cdm_seqdb_free(o); return rc; }
    uint32_t *w = nullptr; void *tmp = nullptr;
    cdmscan::ScanTemp st;
    if (cdmMalloc(&w, (n + 1) * 4) != hipSuccess) { cdm_seqdb_free(o); cdm_set_error("cdm_seqdb_from_packed: out of device memory"); return CDM_ERR_HIP; }
    hipMemcpyAsync(o->len, lengths, n * 4, hipMemcpyDeviceToDevice, s);
    hipMemcpyAsync(o->key, keys, n * 4, hipMemcpyDeviceToDevice, s);
    hipMemcpyAsync(o->codes, codes, words * 4, hipMemcpyDeviceToDevice, s);
    hipMemsetAsync(o->nmask, 0, (((uint64_t) words * 16 + 31) / 32 + 1) * 4, s);
    hipMemcpyAsync(o->nmask, nmask16, words * 2, hipMemcpyDeviceToDevice, s);
    hipLaunchKernelGGL(k_words_of, dim3((unsigned) ((n + 256) / 256)), dim3(256), 0, s, o->len, (uint32_t) n, w, o->ext, extValue, o->hasN);
    if (cdmscan::exclusiveScan<uint32_t>(s, st, w, o->woff, (size_t) n + 1) != CDM_OK) { cdmFree(w); cdm_seqdb_free(o); return CDM_ERR_HIP; }
    if (words) hipLaunchKernelGGL(k_mark_hasN, CDM_GRID(((uint64_t) words + 255) / 256, 256), dim3(256), 0, s, o->woff, o->nmask, (uint32_t) n, words, o->hasN);
    uint32_t total = 0;
    hipMemcpyAsync(&total, o->woff + n, 4, hipMemcpyDeviceToHost, s);
    hipError_t e = hipStreamSynchronize(s);
    cdmFree(w); cdmFree(tmp);
    if (e != hipSuccess) { cdm_seqdb_free(o); cdm_set_error("cdm_seqdb_from_packed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; }
    if (total != words) { cdm_seqdb_free(o); cdm_set_error("cdm_seqdb_from_packed: lengths need %u code words, %llu given", total, (unsigned long long) words); return CDM_ERR_INVALID; }
    if (int rc2 = seqdbLenStats(ctx, o)) { cdm_seqdb_free(o); return rc2; }
    *out = o;
    return CDM_OK;
}

extern "C" int cdm_seqdb_copy_ext(cdm_ctx *ctx, const cdm_seqdb *db, void *devExt) {
    CDM_HIP(hipSetDevice(ctx->device));
    if (db->n) CDM_HIP(hipMemcpyAsync(devExt, db->ext, db->n, hipMemcpyDeviceToDevice, ctx->stream));
    CDM_HIP(hipStreamSynchronize(ctx->stream));
    return CDM_OK;
}
extern "C" int cdm_seqdb_from_packed_ext(cdm_ctx *ctx, const void *codes, const void *nmask16, const void *lengths, const void *keys, const void *devExt, uint64_t n,
                                         uint64_t words, cdm_seqdb **out) {
    cdm_seqdb *o = nullptr;
    int rc = cdm_seqdb_from_packed(ctx, codes, nmask16, lengths, keys, n, words, 0, &o);
    if (rc != CDM_OK) return rc;
    if (devExt && n) {
        if (hipMemcpyAsync(o->ext, devExt, n, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            cdm_seqdb_free(o); cdm_set_error("cdm_seqdb_from_packed_ext: copying the wasExtended flags failed"); return CDM_ERR_HIP;
        }
    }
    *out = o;
    return CDM_OK;
}

// The packed form to and from HOST memory (round 5: the binary side-cars a module process leaves next to the DB it wrote, host/sidecar.cpp -
// the next module of the workflow takes the sequences from there instead of parsing and packing the text again).  nmask16 / raw /
// rawFlags may be NULL on both sides (export: not wanted; import: the DB has no letter beyond ACGT / no raw plane).
extern "C" int cdm_seqdb_export_packed(cdm_ctx *ctx, const cdm_seqdb *db, void *codes, void *nmask16, void *lengths, void *keys, void *ext, void *raw, void *rawFlags) {
    if (!ctx || !db) { cdm_set_error("cdm_seqdb_export_packed: invalid argument"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if (codes && db->words) CDM_HIP(hipMemcpyAsync(codes, db->codes, db->words * 4, hipMemcpyDeviceToHost, s));
    if (nmask16 && db->words) CDM_HIP(hipMemcpyAsync(nmask16, db->nmask, db->words * 2, hipMemcpyDeviceToHost, s));
    if (lengths && db->n) CDM_HIP(hipMemcpyAsync(lengths, db->len, db->n * 4, hipMemcpyDeviceToHost, s));
    if (keys && db->n) CDM_HIP(hipMemcpyAsync(keys, db->key, db->n * 4, hipMemcpyDeviceToHost, s));
    if (ext && db->n) CDM_HIP(hipMemcpyAsync(ext, db->ext, db->n, hipMemcpyDeviceToHost, s));
    if (raw && db->raw && db->words) CDM_HIP(hipMemcpyAsync(raw, db->raw, db->words * 16, hipMemcpyDeviceToHost, s));
    if (rawFlags && db->n) CDM_HIP(hipMemcpyAsync(rawFlags, db->hasN, db->n, hipMemcpyDeviceToHost, s));
    CDM_HIP(hipStreamSynchronize(s));
    return CDM_OK;
}
extern "C" int cdm_seqdb_import_packed(cdm_ctx *ctx, const void *codes, const void *nmask16, const void *lengths, const void *keys, const void *ext, const void *raw,
                                       const void *rawFlags, uint64_t n, uint64_t words, cdm_seqdb **out) {
    if (!ctx || !out || (n && (!lengths || !keys)) || (words && !codes)) { cdm_set_error("cdm_seqdb_import_packed: invalid argument"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DevBuf<uint32_t> dCodes, dLen, dKey; DevBuf<uint16_t> dMask; DevBuf<uint8_t> dExt, dRaw, dFlags;
    if (!dCodes.alloc(words) || !dLen.alloc(n) || !dKey.alloc(n) || !dMask.alloc(words) || !dExt.alloc(n) || (raw && (!dRaw.alloc(words * 16) || !dFlags.alloc(n)))) {
        cdm_set_error("cdm_seqdb_import_packed: out of device memory"); return CDM_ERR_HIP;
    }
    if (words) CDM_HIP(hipMemcpyAsync(dCodes.p, codes, words * 4, hipMemcpyHostToDevice, s));
    if (n) { CDM_HIP(hipMemcpyAsync(dLen.p, lengths, n * 4, hipMemcpyHostToDevice, s)); CDM_HIP(hipMemcpyAsync(dKey.p, keys, n * 4, hipMemcpyHostToDevice, s)); }
    if (words) { if (nmask16) CDM_HIP(hipMemcpyAsync(dMask.p, nmask16, words * 2, hipMemcpyHostToDevice, s)); else CDM_HIP(hipMemsetAsync(dMask.p, 0, words * 2, s)); }
    if (n) { if (ext) CDM_HIP(hipMemcpyAsync(dExt.p, ext, n, hipMemcpyHostToDevice, s)); else CDM_HIP(hipMemsetAsync(dExt.p, 0, n, s)); }
    if (raw) { if (words) CDM_HIP(hipMemcpyAsync(dRaw.p, raw, words * 16, hipMemcpyHostToDevice, s)); if (n) CDM_HIP(hipMemcpyAsync(dFlags.p, rawFlags, n, hipMemcpyHostToDevice, s)); }
    CDM_HIP(hipStreamSynchronize(s));
    cdm_seqdb *o = nullptr;
    if (int rc = cdm_seqdb_from_packed_ext(ctx, dCodes.p, dMask.p, dLen.p, dKey.p, dExt.p, n, words, &o)) return rc;
    if (raw) {      // (the export's flags are the DB's own letter flags: bit 1 = the row of the raw plane counts - what attach wants to know)
        DevBuf<uint8_t> rowCounts;
        if (!rowCounts.alloc(n)) { cdm_seqdb_free(o); cdm_set_error("cdm_seqdb_import_packed: out of device memory"); return CDM_ERR_HIP; }
        if (n) hipLaunchKernelGGL(k_raw_flags, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, (const uint8_t *) dFlags.p, n, rowCounts.p);
        if (int rc = cdm_seqdb_attach_raw(ctx, o, dRaw.p, rowCounts.p)) { cdm_seqdb_free(o); return rc; }
    }
    *out = o;
    return CDM_OK;
}

// ------------------------------------------------------------------------------------------------ hits / alignments
// fn(lo, hi) over [0, n) on a few host threads; with `weight` (n + 1 prefix sums) the ranges carry about the same weight each
#include <thread>
template <typename F>
static void cdmHostParallel(uint64_t n, F fn, const uint64_t *weight = nullptr) {
    unsigned T = std::thread::hardware_concurrency();
    if (const char *e = cdmGetenv("OMP_NUM_THREADS")) { const int v = atoi(e); if (v > 0) T = (unsigned) v; }
    T = std::max(1u, std::min(T, 16u));
    if (n < 100000 || T == 1) { fn(0, n); return; }
    std::vector<uint64_t> cut(T + 1, n);
    cut[0] = 0;
    for (unsigned t = 1; t < T; t++) {
        if (!weight) { cut[t] = n * t / T; continue; }
        const uint64_t want = (weight[n] + n) / T * t;
        uint64_t lo = cut[t - 1], hi = n;
        while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (weight[mid] + mid < want) lo = mid + 1; else hi = mid; }
        cut[t] = lo;
    }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back([&, t] { fn(cut[t], cut[t + 1]); });
    fn(cut[0], cut[1]);
    for (auto &x : th) x.join();
}
template <typename H, typename R>
static int csr_upload(cdm_ctx *ctx, uint64_t n, const uint64_t *offsets, const R *recs, H **out) {
    CDM_HIP(hipSetDevice(ctx->device));
    for (uint64_t i = 0; i < n; i++) if (offsets[i + 1] < offsets[i]) { cdm_set_error("CSR offsets not monotone at %llu", (unsigned long long) i); return CDM_ERR_INVALID; }
    H *h = new H(); h->n = n; h->count = offsets[n];
    if (cdmMalloc(&h->off, (n + 1) * 8) != hipSuccess || cdmMalloc(&h->rec, (h->count + 1) * sizeof(R)) != hipSuccess) {
        cdm_set_error("out of device memory for %llu records", (unsigned long long) h->count); cdmFree(h->off); cdmFree(h->rec); delete h; return CDM_ERR_HIP;
    }
    CDM_HIP(hipMemcpyAsync(h->off, offsets, (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (h->count) CDM_HIP(hipMemcpyAsync(h->rec, recs, h->count * sizeof(R), hipMemcpyHostToDevice, ctx->stream));
    CDM_HIP(hipStreamSynchronize(ctx->stream));
    *out = h;
    return CDM_OK;
}
// The kernels index sequences with the record fields: everything that comes from a file is range-checked on the host first
// (a bad index in a hand-written kernel is a GPU fault, not an error code).
static int fetchLens(cdm_ctx *ctx, const cdm_seqdb *db, std::vector<uint32_t> &lens) {
    lens.resize(db->n);
    CDM_HIP(hipMemcpy(lens.data(), db->len, db->n * 4, hipMemcpyDeviceToHost));
    return CDM_OK;
}
extern "C" int cdm_hits_upload(cdm_ctx *ctx, const cdm_seqdb *db, const uint64_t *offsets, const cdm_hit *hits, cdm_hits **out) {
    static_assert(sizeof(cdm_hit) == sizeof(HitRec), "layout");
    if (!ctx || !db || !offsets || !out || (offsets[db->n] && !hits)) { cdm_set_error("cdm_hits_upload: NULL argument"); return CDM_ERR_INVALID; }
    {   // checked by several threads (a file of 10 M reads brings 37 M records)
        const uint64_t total = offsets[db->n];
        std::atomic<uint64_t> bad{UINT64_MAX};
        cdmHostParallel(total, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; i++)
                if (hits[i].target >= db->n || hits[i].diagonal < -32768 || hits[i].diagonal > 32767) { uint64_t cur = bad.load(); while (i < cur && !bad.compare_exchange_weak(cur, i)) {} return; }
        });
        const uint64_t i = bad.load();
        if (i != UINT64_MAX) {
            cdm_set_error("cdm_hits_upload: record %llu is out of range (target %u of %llu sequences, diagonal %d)", (unsigned long long) i, hits[i].target,
                          (unsigned long long) db->n, hits[i].diagonal);
            return CDM_ERR_INVALID;
        }
    }
    return csr_upload<cdm_hits, HitRec>(ctx, db->n, offsets, reinterpret_cast<const HitRec *>(hits), out);
}
extern "C" uint64_t cdm_hits_count(const cdm_hits *h) { return h->count; }
extern "C" int cdm_hits_download(cdm_ctx *ctx, const cdm_hits *h, uint64_t *offsets, cdm_hit *hits) {
    CDM_HIP(hipSetDevice(ctx->device));
    if (offsets) CDM_HIP(hipMemcpyAsync(offsets, h->off, (h->n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (hits && h->count) CDM_HIP(hipMemcpyAsync(hits, h->rec, h->count * sizeof(HitRec), hipMemcpyDeviceToHost, ctx->stream));
    CDM_HIP(hipStreamSynchronize(ctx->stream));
    return CDM_OK;
}
extern "C" void cdm_hits_free(cdm_hits *h) { if (!h) return; cdmFree(h->off); cdmFree(h->rec); delete h; }

extern "C" int cdm_alns_upload(cdm_ctx *ctx, const cdm_seqdb *db, const uint64_t *offsets, const cdm_aln *alns, cdm_alns **out) {
    static_assert(sizeof(cdm_aln) == sizeof(AlnRec), "layout");
    if (!ctx || !db || !offsets || !out || (offsets[db->n] && !alns)) { cdm_set_error("cdm_alns_upload: NULL argument"); return CDM_ERR_INVALID; }
    {
        CDM_HIP(hipSetDevice(ctx->device));
        std::vector<uint32_t> lens;
        int rc = fetchLens(ctx, db, lens);
        if (rc) return rc;
        std::atomic<uint64_t> badQ{UINT64_MAX};
        auto okRec = [&](uint64_t q, const cdm_aln &r) {
            return r.target < db->n && r.q_start >= 0 && r.q_end >= 0 && (uint32_t) r.q_start < lens[q] && (uint32_t) r.q_end < lens[q] &&
                   r.db_start >= 0 && r.db_end >= r.db_start && (uint32_t) r.db_end < lens[r.target < db->n ? r.target : 0] &&
                   abs(r.q_end - r.q_start) == r.db_end - r.db_start;   // ungapped: both spans have the same length
        };
        cdmHostParallel(db->n, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t q = lo; q < hi; q++)
                for (uint64_t i = offsets[q]; i < offsets[q + 1] && offsets[q + 1] >= offsets[q]; i++)
                    if (!okRec(q, alns[i])) { uint64_t cur = badQ.load(); while (q < cur && !badQ.compare_exchange_weak(cur, q)) {} return; }
        }, offsets);
        if (badQ.load() != UINT64_MAX) {
            const uint64_t q = badQ.load();
            for (uint64_t i = offsets[q]; i < offsets[q + 1]; i++) {
                const cdm_aln &r = alns[i];
                if (okRec(q, r)) continue;
                cdm_set_error("cdm_alns_upload: alignment record %llu of query %llu does not fit the sequence DB (target %u, q %d-%d of %u, db %d-%d)",
                              (unsigned long long) i, (unsigned long long) q, r.target, r.q_start, r.q_end, lens[q], r.db_start, r.db_end);
                return CDM_ERR_INVALID;
            }
        }
    }
    return csr_upload<cdm_alns, AlnRec>(ctx, db->n, offsets, reinterpret_cast<const AlnRec *>(alns), out);
}
extern "C" uint64_t cdm_alns_count(const cdm_alns *a) { return a->count; }
extern "C" int cdm_alns_download(cdm_ctx *ctx, const cdm_alns *a, uint64_t *offsets, cdm_aln *alns) {
    CDM_HIP(hipSetDevice(ctx->device));
    if (offsets) CDM_HIP(hipMemcpyAsync(offsets, a->off, (a->n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (alns && a->count) CDM_HIP(hipMemcpyAsync(alns, a->rec, a->count * sizeof(AlnRec), hipMemcpyDeviceToHost, ctx->stream));
    CDM_HIP(hipStreamSynchronize(ctx->stream));
    return CDM_OK;
}
extern "C" void cdm_alns_free(cdm_alns *a) { if (!a) return; cdmFree(a->off); cdmFree(a->rec); if (a->ryMism) cdmFree(a->ryMism); delete a; }

// ------------------------------------------------------------------------------------------------ stage wrappers
// device time of the whole stage call (HIP events on the context stream around everything the call launched, host round
// trips between its kernels included): lastMs[8..11] = kmermatcher, rescorediagonal, ancient_correction, ancient_read_assemble
static void stageDone(cdm_ctx *ctx, int slot) {
    hipEventRecord(ctx->evS1, ctx->stream);
    if (hipEventSynchronize(ctx->evS1) == hipSuccess) hipEventElapsedTime(&ctx->lastMs[slot], ctx->evS0, ctx->evS1);
}
extern "C" int cdm_correct(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, cdm_seqdb **out) {
    if (alns) CDM_REFUSE_UNDEFINED_ALNS(alns, "cdm_correct");
    if (!ctx || !db || !alns || !par || !out) { cdm_set_error("cdm_correct: NULL argument"); return CDM_ERR_INVALID; }
    if (!ctx->haveDamage) { cdm_set_error("cdm_correct: call cdm_damage_load first"); return CDM_ERR_INVALID; }
    if (alns->n != db->n) { cdm_set_error("cdm_correct: alignment CSR has %llu queries, DB has %llu", (unsigned long long) alns->n, (unsigned long long) db->n); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    cdm_seqdb *o = nullptr;
    int rc = cdm_seqdb_alloc_like(ctx, db, &o);
    if (rc) return rc;
    hipEventRecord(ctx->evS0, ctx->stream);
    rc = cdm_correct_impl(ctx, db, alns, par, o);
    if (rc) { cdm_seqdb_free(o); return rc; }
    stageDone(ctx, 10);
    *out = o;
    return CDM_OK;
}
extern "C" int cdm_rescore(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_hits *hits, const cdm_rescore_params *par, cdm_alns **out) {
    if (!ctx || !db || !hits || !par || !out) { cdm_set_error("cdm_rescore: NULL argument"); return CDM_ERR_INVALID; }
    if (hits->n != db->n) { cdm_set_error("cdm_rescore: hit CSR / DB size mismatch"); return CDM_ERR_INVALID; }
    if (par->seq_id_mode != 0) { cdm_set_error("cdm_rescore: only --seq-id-mode 0 is implemented"); return CDM_ERR_UNSUPPORTED; }
    CDM_HIP(hipSetDevice(ctx->device));
    hipEventRecord(ctx->evS0, ctx->stream);
    const int rc = cdm_rescore_impl(ctx, db, hits, par, out);
    if (rc == CDM_OK) stageDone(ctx, 9);
    return rc;
}
extern "C" int cdm_rescore_hamming(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_hits *hits, const cdm_hamming_params *par, cdm_hits **out) {
    if (!ctx || !db || !hits || !par || !out) { cdm_set_error("cdm_rescore_hamming: NULL argument"); return CDM_ERR_INVALID; }
    if (hits->n != db->n) { cdm_set_error("cdm_rescore_hamming: hit CSR / DB size mismatch"); return CDM_ERR_INVALID; }
    if (par->seq_id_mode < 0 || par->seq_id_mode > 2) { cdm_set_error("cdm_rescore_hamming: --seq-id-mode is 0, 1 or 2"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    return cdm_rescore_hamming_impl(ctx, db, hits, par, out);
}
extern "C" int cdm_kmermatch(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_kmer_params *par, cdm_hits **out) {
    if (!ctx || !db || !par || !out) { cdm_set_error("cdm_kmermatch: NULL argument"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    hipEventRecord(ctx->evS0, ctx->stream);
    const int rc = cdm_kmermatch_impl(ctx, db, par, out);
    if (rc == CDM_OK) stageDone(ctx, 8);
    return rc;
}
extern "C" int cdm_extend(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, cdm_seqdb **out, double *scores) {
    if (alns) CDM_REFUSE_UNDEFINED_ALNS(alns, "cdm_extend");
    if (!ctx || !db || !alns || !par || !out) { cdm_set_error("cdm_extend: NULL argument"); return CDM_ERR_INVALID; }
    if (!ctx->haveDamage) { cdm_set_error("cdm_extend: call cdm_damage_load first"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    hipEventRecord(ctx->evS0, ctx->stream);
    const int rc = cdm_extend_impl(ctx, db, alns, par, out, scores);
    if (rc == CDM_OK) stageDone(ctx, 11);
    return rc;
}
