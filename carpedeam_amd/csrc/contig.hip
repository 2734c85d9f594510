// ancient_contig_merge (src/assembler/ancientContigsResults.cpp:94-509), SURVEY.md 8(f) rank 1.
//
// Split: the per-alignment work that touches every aligned column - orientation, identities against the query, the counts
// updateSeqIdConsensus and ancientMatchCount take over the consensus (nuclassembleUtil.cpp:705-790, 1047-1181) - is k_contig_stats below,
// a wavefront per alignment record on 2-bit words.  What follows - the contig filter, the priority queue with its Beta-posterior
// comparator (:25-70: lgammaf / logf / exp of the C library decide the order, and it is no strict weak ordering), the extension loop
// and the re-alignment of parked hits - runs on the device as well since round 5 (contigqueue.hip: the comparator from tables of the
// C library's own values, libstdc++'s heap step for step).  The host code of rounds 2-4 (host/contigmerge.cpp, compiled like the
// reference) stays for --unsafe 1, for small calls in a process that has not filled those tables, for the queries the device hands
// back, and as what the tests compare the device queue with (hostQueue below).
#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "devutil.h"
#include "contigqueue.h"

namespace {
struct StatArgs {
    const SeqMeta *meta; const uint32_t *codes, *nmask; const uint8_t *raw;
    const uint64_t *aoff; const AlnRec *rec; const uint32_t *owner;   // owner[r] = query of record r
    uint64_t nRec;
    uint64_t first;              // first record of this launch (launches are slices of the records: common.h cdmSliceItems)
    ContigStat *out;
};
// hasRaw: the sequence carries letters beyond ACGTN.  Forward, the reference looks at the original byte (letter identity, == 'N',
// ryMap / nucleotideMap = 0 for what they do not hold): such a letter comes back as 0x100 | byte; reversed, at the complement of what
// NucleotideMatrix maps it to (codes / nmask)
__device__ __forceinline__ void letterAt(const StatArgs &a, uint32_t w0, uint32_t len, bool hasN, bool rev, uint32_t pos, uint32_t &code, bool &isN, bool hasRaw = false) {
    const uint32_t p = rev ? (len - 1u - pos) : pos;
    const uint32_t c = cdm_base(a.codes, w0, p);
    isN = hasN && cdm_isN(a.nmask, w0, p);
    code = rev ? (3u - c) : c;      // getNuclRevFragment maps X to 'N': an N stays an N
    if (hasRaw && !rev) {
        const uint8_t r = cdm_raw_at(a.raw, w0, p);
        isN = r == 'N';
        if (isN) code = 0; else if (!(r == 'A' || r == 'C' || r == 'G' || r == 'T')) code = 0x100u | r;
    }
}
__device__ __forceinline__ uint32_t ryOf(uint32_t code) { return code < 4u ? (code & 1u) : 0u; }
__global__ __launch_bounds__(256) void k_contig_stats(StatArgs a) {
    const uint64_t r = a.first + (((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= a.nRec) return;
    const AlnRec rec = a.rec[r];
    const uint32_t q = a.owner[r];
    const SeqMeta qm = a.meta[q], tm = a.meta[rec.target];
    ContigStat s;
    s.dbLen = tm.len; s.dbKey = tm.key;
    s.rev = rec.qStart > rec.qEnd;
    if (s.rev) { s.qs = rec.qEnd; s.qe = rec.qStart; s.ds = (int) tm.len - rec.dbEnd - 1; s.de = (int) tm.len - rec.dbStart - 1; }
    else { s.qs = rec.qStart; s.qe = rec.qEnd; s.ds = rec.dbStart; s.de = rec.dbEnd; }
    int idCnt = 0, idRy = 0, nnTot = 0, nnId = 0, nnRy = 0, nCT = 0, nGA = 0;
    const bool qHasN = (qm.flags & 1u) != 0, tHasN = (tm.flags & 1u) != 0;
    const int n = s.qe - s.qs + 1;
    if (!qHasN && !tHasN && !s.rev) {
        // 16 columns per lane and step: XOR of the two windows
        const uint32_t qLast = (qm.len + 15) / 16 - 1, tLast = (tm.len + 15) / 16 - 1;
        for (int c0 = lane * 16; c0 < n; c0 += 64 * 16) {
            const uint32_t qw = cdm_window16(a.codes, qm.woff, (uint32_t) (s.qs + c0), qLast), tw = cdm_window16(a.codes, tm.woff, (uint32_t) (s.ds + c0), tLast);
            const uint32_t x = qw ^ tw;
            uint32_t any = (x | (x >> 1)) & 0x55555555u, low = x & 0x55555555u;
            // query C (01) over target T (11): x = 10, query low bit set; query G (10) over target A (00): x = 10, query low bit clear
            uint32_t hiOnly = (x >> 1) & ~x & 0x55555555u, ct = hiOnly & qw & ~(qw >> 1), ga = hiOnly & ~qw & (qw >> 1);
            const int rem = n - c0;
            if (rem < 16) { const uint32_t m = (1u << (2 * rem)) - 1u; any &= m; low &= m; ct &= m; ga &= m; }
            const int cols = min(16, rem);
            idCnt += cols - __popc(any); idRy += cols - __popc(low); nCT += __popc(ct); nGA += __popc(ga);
        }
        idCnt = cdm_wave_sum(idCnt); idRy = cdm_wave_sum(idRy); nCT = cdm_wave_sum(nCT); nGA = cdm_wave_sum(nGA);
        nnTot = n; nnId = idCnt; nnRy = idRy;
    } else {
        for (int c = lane; c < n; c += 64) {
            uint32_t qc, tc; bool qn, tn;
            letterAt(a, qm.woff, qm.len, qHasN, false, (uint32_t) (s.qs + c), qc, qn, a.raw && (qm.flags & 4u));
            letterAt(a, tm.woff, tm.len, tHasN, s.rev != 0, (uint32_t) (s.ds + c), tc, tn, a.raw && (tm.flags & 4u));
            const uint32_t ql = qn ? 4u : qc, tl = tn ? 4u : tc;
            idCnt += (ql == tl);
            idRy += (ryOf(qn ? 0u : qc) == ryOf(tn ? 0u : tc));       // ryMap of a letter outside ACGT is 0
            if (!qn && !tn) { nnTot++; nnId += (qc == tc); nnRy += (ryOf(qc) == ryOf(tc)); nCT += (qc == 1u && tc == 3u); nGA += (qc == 2u && (tc == 0u || tc >= 4u)); }      // nucleotideMap: every target letter beyond ACGT is base 0, so G over 'a' or 'R' is a G->A column too
        }
        idCnt = cdm_wave_sum(idCnt); idRy = cdm_wave_sum(idRy); nnTot = cdm_wave_sum(nnTot); nnId = cdm_wave_sum(nnId); nnRy = cdm_wave_sum(nnRy);
        nCT = cdm_wave_sum(nCT); nGA = cdm_wave_sum(nGA);
    }
    if (lane == 0) { s.idCnt = idCnt; s.idRy = idRy; s.nnTot = nnTot; s.nnId = nnId; s.nnRy = nnRy; s.nCT = nCT; s.nGA = nGA; a.out[r] = s; }
}
__global__ void k_rec_owner(const uint64_t *__restrict__ aoff, uint32_t n, uint32_t *__restrict__ owner) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    for (uint64_t r = aoff[q]; r < aoff[q + 1]; r++) owner[r] = q;
}
}  // namespace

// host/contigmerge.cpp
// (host/contigmerge.cpp, OpenMP) the DB blob as one view per sequence
bool cdm_host_pack(const std::vector<std::string> &seqs, HostBuf<char> &data, std::vector<uint64_t> &off, std::vector<uint32_t> &len);
void cdm_host_split(const char *blob, const std::vector<uint64_t> &offs, const std::vector<uint32_t> &lens, std::vector<SeqView> &seqs);
int cdm_contig_merge_host(const std::vector<SeqView> &seqs, const std::vector<uint32_t> &keys, const std::vector<uint8_t> &ext, const std::vector<uint64_t> &aoff,
                          const cdm_aln *recs, const ContigStat *stats, const long double mats[2][11][4][4], const cdm_ancient_params *par,
                          float mergeSeqIdThr, std::vector<uint32_t> &grownIdx, std::vector<std::string> &grownSeqs, std::vector<uint8_t> &outExt, std::string *err,
                          const uint8_t *only);

namespace {
// The queue on the host (all queries, or those `only` marks): statistics, records and the DB come down, host/contigmerge.cpp runs, the
// grown contigs go up as a DB of their own.  grown may come back NULL (nothing grew).
int hostQueue(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, float mergeSeqIdThr, const ContigStat *dStats, const uint8_t *only,
              std::vector<uint32_t> &grownIdx, cdm_seqdb **grown, std::vector<uint8_t> &outExt) {
    hipStream_t s = ctx->stream;
    const uint32_t n = (uint32_t) db->n;
    const uint64_t nRec = alns->count;
    const bool timing = cdmGetenv("CDM_TIMING") != nullptr;
    auto tNow = [] { return std::chrono::steady_clock::now(); };
    auto tPrev = tNow();
    auto lap = [&](const char *what) { if (timing) { const auto t = tNow(); fprintf(stderr, "  contig merge: %-28s %.3f s\n", what, std::chrono::duration<double>(t - tPrev).count()); tPrev = t; } };
    HostBuf<ContigStat> stats; HostBuf<cdm_aln> recs; std::vector<uint64_t> aoff(n + 1);
    if (!stats.alloc(nRec) || !recs.alloc(nRec)) { cdm_set_error("cdm_contig_merge: out of host memory"); return CDM_ERR_INVALID; }
    std::vector<uint32_t> lens(n), keys(n); std::vector<uint8_t> ext(n);
    CDM_HIP(hipMemcpyAsync(stats.data(), dStats, nRec * sizeof(ContigStat), hipMemcpyDeviceToHost, s));
    CDM_HIP(hipMemcpyAsync(aoff.data(), alns->off, (n + 1) * 8, hipMemcpyDeviceToHost, s));
    if (nRec) CDM_HIP(hipMemcpyAsync(recs.data(), alns->rec, nRec * sizeof(cdm_aln), hipMemcpyDeviceToHost, s));
    CDM_HIP(hipStreamSynchronize(s));
    if (int rc = cdm_seqdb_meta(ctx, db, lens.data(), keys.data(), ext.data())) return rc;
    std::vector<uint64_t> offs(n); uint64_t tot = 0;
    for (uint32_t i = 0; i < n; i++) { offs[i] = tot; tot += lens[i] + 1; }
    HostBuf<char> blob;
    if (!blob.alloc(tot)) { cdm_set_error("cdm_contig_merge: out of host memory"); return CDM_ERR_INVALID; }
    lap("statistics + records down");
    if (int rc = cdm_seqdb_download(ctx, db, blob.data(), offs.data())) return rc;
    lap("sequences down");
    std::vector<SeqView> seqs(n); std::vector<std::string> grownSeqs;
    cdm_host_split(blob.data(), offs, lens, seqs);      // views into the blob, which stays until the host part is done
    lap("split");
    std::string err;
    if (int rc = cdm_contig_merge_host(seqs, keys, ext, aoff, recs.data(), stats.data(), ctx->mats, par, mergeSeqIdThr, grownIdx, grownSeqs, outExt, &err, only)) { cdm_set_error("%s", err.c_str()); return rc; }
    lap("queues + extension (host)");
    { std::vector<SeqView>().swap(seqs); blob.release(); stats.release(); recs.release(); }
    const size_t m = grownIdx.size();
    *grown = nullptr;
    if (m) {
        std::vector<uint64_t> gOff; std::vector<uint32_t> gLen, gKey(m);
        HostBuf<char> data;
        if (!cdm_host_pack(grownSeqs, data, gOff, gLen)) { cdm_set_error("cdm_contig_merge: out of host memory"); return CDM_ERR_INVALID; }
        for (size_t j = 0; j < m; j++) gKey[j] = keys[grownIdx[j]];
        std::vector<std::string>().swap(grownSeqs);
        lap("pack");
        if (int rc = cdm_seqdb_upload(ctx, data.data(), gOff.data(), gLen.data(), gKey.data(), nullptr, m, grown)) return rc;
        lap("upload");
    }
    return CDM_OK;
}
}  // namespace

extern "C" int cdm_contig_merge(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, float mergeSeqIdThr, cdm_seqdb **out) {
    if (!ctx || !db || !alns || !par || !out) { cdm_set_error("cdm_contig_merge: NULL argument"); return CDM_ERR_INVALID; }
    CDM_REFUSE_UNDEFINED_ALNS(alns, "cdm_contig_merge");
    if (!ctx->haveDamage) { cdm_set_error("cdm_contig_merge: call cdm_damage_load first"); return CDM_ERR_INVALID; }
    if (alns->n != db->n) { cdm_set_error("cdm_contig_merge: alignment CSR / DB size mismatch"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t n = (uint32_t) db->n;
    const uint64_t nRec = alns->count;
    DevBuf<SeqMeta> meta; DevBuf<uint32_t> owner; DevBuf<ContigStat> dStats;
    if (int rc = cdm_build_meta(ctx, db, &meta.p)) return rc;
    if (!owner.alloc(nRec) || !dStats.alloc(nRec)) { cdm_set_error("cdm_contig_merge: out of device memory"); return CDM_ERR_HIP; }
    if (n) hipLaunchKernelGGL(k_rec_owner, dim3((n + 255) / 256), dim3(256), 0, s, alns->off, n, owner.p);
    StatArgs a; a.meta = meta.p; a.codes = db->codes; a.nmask = db->nmask; a.raw = db->raw; a.aoff = alns->off; a.rec = alns->rec; a.owner = owner.p; a.nRec = nRec; a.out = dStats.p;
    hipEventRecord(ctx->ev0, s);
    for (uint64_t first = 0, slice = cdmSliceItems(64); first < nRec; first += slice) {
        a.first = first;
        hipLaunchKernelGGL(k_contig_stats, CDM_GRID((std::min(slice, nRec - first) * 64 + 255) / 256, 256), dim3(256), 0, s, a);
    }
    hipEventRecord(ctx->ev1, s);
    const bool timing = cdmGetenv("CDM_TIMING") != nullptr;
    auto tPrev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (timing) { const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "  contig merge: %-28s %.3f s\n", what, std::chrono::duration<double>(t - tPrev).count()); tPrev = t; } };
    // Where the queue runs: on the device, unless --unsafe 1 (its consensus works on the host's strings) - or the call is small and this
    // process has not filled the device queue's tables yet: they take about a second once per process, which a module process started per
    // iteration (the reference's scripts) pays every time, while the host queue takes ~80 ns per record all in.  CDM_CONTIG_QUEUE=host|device
    // pins either.
    const char *where = cdmGetenv("CDM_CONTIG_QUEUE");
    const bool small = nRec < (8ull << 20) && !cdm_contig_tables_ready(ctx->device);
    const bool onHost = par->unsafe != 0 || (where ? !strcmp(where, "host") : small);
    std::vector<uint32_t> grownIdx; std::vector<uint8_t> outExt; cdm_seqdb *grown = nullptr;
    if (onHost) {
        if (int rc = hostQueue(ctx, db, alns, par, mergeSeqIdThr, dStats.p, nullptr, grownIdx, &grown, outExt)) return rc;
        hipEventElapsedTime(&ctx->lastMs[12], ctx->ev0, ctx->ev1);
        tPrev = std::chrono::steady_clock::now();
        const int rcUp = cdm_seqdb_overlay(ctx, db, grown, grownIdx.data(), outExt.data(), out);
        if (grown) cdm_seqdb_free(grown);
        lap("overlay");
        return rcUp;
    }
    CqResult res;
    if (int rc = cdm_contig_queue_device(ctx, db, alns, par, mergeSeqIdThr, meta.p, owner.p, dStats.p, &res)) { if (res.grown) cdm_seqdb_free(res.grown); return rc; }
    hipEventElapsedTime(&ctx->lastMs[12], ctx->ev0, ctx->ev1);
    tPrev = std::chrono::steady_clock::now();
    if (!res.nHandedBack) {
        const int rcUp = cdm_seqdb_overlay(ctx, db, res.grown, res.grownIdx.data(), res.outExt.data(), out);
        if (res.grown) cdm_seqdb_free(res.grown);
        lap("overlay");
        return rcUp;
    }
    // the few queries the device handed back: the host code on those alone, its contigs overlaid on the device's
    cdm_seqdb *mid = nullptr;
    int rc = cdm_seqdb_overlay(ctx, db, res.grown, res.grownIdx.data(), res.outExt.data(), &mid);
    if (res.grown) cdm_seqdb_free(res.grown);
    if (rc != CDM_OK) return rc;
    lap("overlay");
    std::vector<uint8_t> hostExt;
    rc = hostQueue(ctx, db, alns, par, mergeSeqIdThr, dStats.p, res.handedBack.data(), grownIdx, &grown, hostExt);
    if (rc != CDM_OK) { cdm_seqdb_free(mid); return rc; }
    for (uint32_t i = 0; i < n; i++) if (res.handedBack[i]) res.outExt[i] = hostExt[i];
    tPrev = std::chrono::steady_clock::now();
    rc = cdm_seqdb_overlay(ctx, mid, grown, grownIdx.data(), res.outExt.data(), out);
    if (grown) cdm_seqdb_free(grown);
    cdm_seqdb_free(mid);
    lap("overlay of the handed-back queries");
    return rc;
}
