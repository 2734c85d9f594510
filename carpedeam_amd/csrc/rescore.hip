// rescorediagonal (--rescore-mode 3, query DB == target DB) on the device.
//
// Replaces the loop at lib/mmseqs/src/alignment/rescorediagonal.cpp:145-356:
//   per prefilter hit: query strand (:194-202), canBeCovered (:211-213), ungapped end-to-end score on the voted diagonal
//   incl. the +-65536 probing of the 16-bit diagonal (DistanceCalculator.h:93-113,115-175,204-220; scores +2/-3, any X -3),
//   E-value gate (:251, decided through a host-built "minimum passing raw score per query length" table so that the
//   double-precision ALP arithmetic stays on the host), identity count (:278-282), coverage / seq.id. filters (:304-314),
//   reverse-strand coordinate encoding (:294-297).
// One thread per hit; sequences are compared 16 bases at a time on the 2-bit words (XOR + popcount).  Hits whose two
// sequences contain an N take a per-base path.  Survivors are compacted per query with a prefix sum.
#include "scan.h"

#include <cfloat>
#include <climits>

#include "common.h"
#include "devutil.h"

namespace {

struct RescoreArgs {
    MetaWoff woff; MetaLen len; MetaHasN hasN; MetaRaw hasRaw;      // per-sequence metadata, one record per sequence
    const uint32_t *codes, *nmask; const uint8_t *raw;
    const uint64_t *hoff;
    const HitRec *hit;
    const int32_t *minScore;   // [maxLen+1]
    uint64_t nHits;
    uint32_t n;
    float seqIdThr, covThr;
    int covMode, minAlnLen;
    AlnRec *tmp;               // [nHits] candidate records
    uint16_t *tmpRy;           // [nHits] purine/pyrimidine mismatches of the candidate (0xFFFF = not counted)
    uint8_t *valid;            // [nHits]
    unsigned int *undef;       // [1] number of identity records written with the coordinates -1 (see k_rescore)
};

__device__ __forceinline__ bool canBeCovered(float covThr, int covMode, float ql, float tl) {   // M/commons/Util.cpp:533-550
    switch (covMode) {
        case 0: return ((ql / tl >= covThr) && (tl / ql >= covThr));
        case 2: return ((tl / ql) >= covThr);
        case 1: return ((ql / tl) >= covThr);
        case 3: return ((tl / ql) >= covThr) && (tl / ql) <= 1.0;
        case 4: return ((ql / tl) >= covThr) && (ql / tl) <= 1.0;
        case 5: return (fminf(tl, ql) / fmaxf(tl, ql)) >= covThr;
        default: return true;
    }
}
__device__ __forceinline__ bool hasCoverage(float covThr, int covMode, float qc, float tc) {   // Util.cpp:552-567
    switch (covMode) { case 0: return qc >= covThr && tc >= covThr; case 2: return qc >= covThr; case 1: return tc >= covThr; default: return true; }
}
__device__ __forceinline__ float computeCov(unsigned s, unsigned e, unsigned len) {           // StripedSmithWaterman.cpp:1055-1057
    return (min(len, max(s, e)) - min(s, e) + 1) / (float) len;
}

// oriented base and N flag (per-base path)
__device__ __forceinline__ void orientedBase(const RescoreArgs &a, uint32_t w0, uint32_t L, bool hasN, bool rc, uint32_t i, uint32_t &code, bool &isN) {
    const uint32_t p = rc ? (L - 1 - i) : i;
    code = cdm_base(a.codes, w0, p);
    isN = hasN && cdm_isN(a.nmask, w0, p);
    if (rc) code = 3u - code;
}

struct Diag { unsigned score; unsigned diagLen; unsigned dist; int diagonal; unsigned ident; unsigned ry; bool any; unsigned first, last; };

// ungappedAlignmentByDiagonal + computeGlobalSubstitutionStartEndDistance for one real diagonal
__device__ __forceinline__ void scoreDiagonal(const RescoreArgs &a, uint32_t qw, uint32_t qLen, bool qN, bool rc, uint32_t tw, uint32_t tLen, bool tN,
                                              int diagonal, Diag &best, bool qRaw = false, bool tRaw = false) {
    const unsigned md = (unsigned) abs(diagonal);
    uint32_t qOff, tOff, m;
    if (diagonal >= 0 && md < qLen) { qOff = md; tOff = 0; m = min(tLen, qLen - md); }
    else if (diagonal < 0 && md < tLen) { qOff = 0; tOff = md; m = min(tLen - md, qLen); }
    else return;   // res.score stays 0: never beats max (strict >)
    unsigned mism = 0, identN = 0, ry = 0xFFFFu;
    uint32_t first = 0, last = m - 1;
    if (qRaw || tRaw) {
        // computeGlobalSubstitutionStartEndDistance (DistanceCalculator.h:204-220) leaves a '*' at either end of the overlap out
        const uint8_t q0 = (qRaw && !rc) ? cdm_raw_at(a.raw, qw, qOff) : 0, t0 = tRaw ? cdm_raw_at(a.raw, tw, tOff) : 0;
        const uint8_t q1 = (qRaw && !rc) ? cdm_raw_at(a.raw, qw, qOff + m - 1) : 0, t1 = tRaw ? cdm_raw_at(a.raw, tw, tOff + m - 1) : 0;
        first = (q0 == '*' || t0 == '*') ? 1u : 0u;
        if (last > 0 && (q1 == '*' || t1 == '*')) last--;
    }
    if (!qN && !tN) {
        ry = 0;
        const uint32_t qLast = (qLen + 15) / 16 - 1, tLast = (tLen + 15) / 16 - 1;
        for (uint32_t k = 0; k < m; k += 16) {
            const uint32_t x = cdm_oriented_window16(a.codes, qw, qLen, qLast, rc, qOff + k) ^ cdm_window16(a.codes, tw, tOff + k, tLast);
            uint32_t mm = (x | (x >> 1)) & 0x55555555u;
            const uint32_t rem = m - k;
            uint32_t rr = x & 0x55555555u;                      // RY class = low bit of the code; complementing both keeps it
            if (rem < 16) { mm &= (1u << (2 * rem)) - 1u; rr &= (1u << (2 * rem)) - 1u; }
            mism += __popc(mm); ry += __popc(rr);
        }
    } else {
        for (uint32_t k = first; k <= last && k < m; k++) {
            uint32_t qc, tc; bool qn, tn;
            orientedBase(a, qw, qLen, qN, rc, qOff + k, qc, qn);
            orientedBase(a, tw, tLen, tN, false, tOff + k, tc, tn);
            const bool match = !qn && !tn && qc == tc;
            mism += !match;
            // identity count compares letters: forward N == N; the reversed query spells N as 'X' (rescorediagonal.cpp:173-179)
            if (!qRaw && !tRaw) identN += (qn && tn && !rc);
            else {
                // letters beyond ACGTN: the score above is on the mapped codes (createAsciiSubMat), the identity count on the
                // case-folded original bytes (:278-282) - of the target always, of the query on the forward strand; the reversed
                // query is spelled in mapped, complemented upper-case letters (:173-179)
                uint8_t ql = qn ? (rc ? 'X' : 'N') : (uint8_t) "ACGT"[qc], tl = tn ? 'N' : (uint8_t) "ACGT"[tc];
                if (qRaw && !rc) ql = cdm_raw_at(a.raw, qw, qOff + k);
                if (tRaw) tl = cdm_raw_at(a.raw, tw, tOff + k);
                identN += ((ql & 0xDFu) == (tl & 0xDFu)) - (int) match;
            }
        }
    }
    const uint32_t cols = last + 1u >= first ? last + 1u - first : 0u;
    const long long sc = 2ll * (cols - mism) - 3ll * mism;
    const unsigned score = sc > 0 ? (unsigned) sc : 0u;
    if (score > best.score) { best.score = score; best.diagLen = m; best.dist = md; best.diagonal = diagonal; best.ident = (cols - mism) + identN; best.ry = ry; best.any = true; best.first = first; best.last = last; }
}

__global__ __launch_bounds__(256) void k_rescore(RescoreArgs a, const uint32_t *__restrict__ hitQuery) {
    const uint64_t h = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= a.nHits) return;
    const uint32_t q = hitQuery[h];
    const HitRec hit = a.hit[h];
    const uint32_t t = hit.target;
    const uint32_t qLen = a.len[q], tLen = a.len[t], qw = a.woff[q], tw = a.woff[t];
    const bool qN = a.hasN[q] != 0, tN = a.hasN[t] != 0;
    const bool qRaw = qN && a.hasRaw[q] != 0, tRaw = tN && a.hasRaw[t] != 0;
    const bool isReverse = hit.score < 0;
    const bool isIdentity = (q == t);
    a.valid[h] = 0;
    if (!canBeCovered(a.covThr, a.covMode, (float) qLen, (float) tLen)) return;
    // computeUngappedAlignment (DistanceCalculator.h:93-113) on the 16-bit diagonal
    const unsigned short u = (unsigned short) (short) hit.diagonal;
    Diag best; best.score = 0; best.diagLen = 0; best.dist = 0; best.diagonal = 0; best.ident = 0; best.ry = 0xFFFFu; best.any = false; best.first = 0; best.last = 0;
    for (unsigned d = 1; d <= 1 + tLen / 32768; d++) scoreDiagonal(a, qw, qLen, qN, isReverse, tw, tLen, tN, (int) (-(int) d * 65536 + (int) u), best, qRaw, tRaw);
    for (unsigned d = 0; d <= qLen / 65536; d++) scoreDiagonal(a, qw, qLen, qN, isReverse, tw, tLen, tN, (int) (d * 65536 + u), best, qRaw, tRaw);
    if (!best.any) {
        // Score 0 on every probe: E-value(0) never passes, but the identity record is written whatever its score (:304).  The
        // reference's alignment then still holds its constructor's values - start = end = -1, distance 0 (DistanceCalculator.h:50) - so
        // the record carries the coordinates -1, an alignment length of 1 and, the one compared "column" being the byte in FRONT of
        // the sequence in both operands (query and target are the same DB entry), identity 1/1.  rescorediagonal's own DB is that
        // record; what indexes a sequence with it downstream is undefined in the reference, and refused here (cdm_alns::undefinedRecords).
        if (!isIdentity) return;
        AlnRec r; r.target = t; r.rawScore = 0; r.ident = 1; r.qStart = r.qEnd = r.dbStart = r.dbEnd = -1; r.seqId = 1.0f;
        a.tmp[h] = r; a.tmpRy[h] = 0xFFFFu; a.valid[h] = 1;
        atomicAdd(a.undef, 1u);
        return;
    }
    const int startPos = (int) best.first, endPos = (int) best.last;
    const int alnLen = (endPos - startPos) + 1;
    int qs, qe, ds, de;
    if (best.diagonal >= 0) { qs = startPos + (int) best.dist; qe = endPos + (int) best.dist; ds = startPos; de = endPos; }
    else { qs = startPos; qe = endPos; ds = startPos + (int) best.dist; de = endPos + (int) best.dist; }
    const bool hasEvalue = (int) best.score >= a.minScore[qLen];
    float seqId = 0.f;
    if (hasEvalue || isIdentity) seqId = static_cast<float>(best.ident) / static_cast<float>(alnLen);
    const float queryCov = computeCov(qs, qe, qLen), targetCov = computeCov(ds, de, tLen);
    if (isReverse) { qs = (int) qLen - qs - 1; qe = (int) qLen - qe - 1; }
    const bool hasCov = hasCoverage(a.covThr, a.covMode, queryCov, targetCov);
    const bool hasSeqId = (double) seqId >= (double) (a.seqIdThr - FLT_EPSILON);
    const bool hasAlnLen = alnLen >= a.minAlnLen;
    if (!(isIdentity || (hasAlnLen && hasCov && hasSeqId && hasEvalue))) return;
    AlnRec r;
    r.target = t; r.rawScore = (int) best.score; r.ident = (int) best.ident; r.qStart = qs; r.qEnd = qe; r.dbStart = ds; r.dbEnd = de;
    // what a reader of the text record gets back: fastSeqIdToBuffer truncates to 3 decimals, "1.00" for 1 (Util.cpp:278-307)
    r.seqId = (seqId == 1.0f) ? 1.0f : (float) ((double) (int) (seqId * 1000) / 1000.0);
    a.tmp[h] = r;
    a.tmpRy[h] = (uint16_t) min(best.ry, 0xFFFFu);
    a.valid[h] = 1;
}

// hit -> owning query (CSR expansion): one thread per query
__global__ void k_expand(const uint64_t *__restrict__ off, uint32_t n, uint32_t *__restrict__ owner) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    for (uint64_t h = off[q]; h < off[q + 1]; h++) owner[h] = q;
}
__global__ void k_count_valid(const uint64_t *__restrict__ off, const uint8_t *__restrict__ valid, uint32_t n, uint64_t *__restrict__ cnt) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    uint64_t c = 0;
    for (uint64_t h = off[q]; h < off[q + 1]; h++) c += valid[h];
    cnt[q] = c;
}
// Valid candidate records move to their query's output slice, keeping their order.  A thread per record (its rank = valid
// records of the same query in front of it, a short look back) keeps loads and stores coalesced; queries with more than
// SCATTER_SMALL records (deep pile-ups) are left to a thread per query so that the look back stays short.
constexpr uint32_t SCATTER_SMALL = 256;
__global__ void k_scatter(const uint64_t *__restrict__ off, const uint32_t *__restrict__ owner, const uint8_t *__restrict__ valid, const AlnRec *__restrict__ tmp,
                          const uint16_t *__restrict__ tmpRy, uint64_t nHits, const uint64_t *__restrict__ outOff, AlnRec *__restrict__ out, uint16_t *__restrict__ outRy) {
    const uint64_t h = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nHits || !valid[h]) return;
    const uint32_t q = owner[h];
    const uint64_t first = off[q];
    if (off[q + 1] - first > SCATTER_SMALL) return;
    uint64_t o = outOff[q];
    for (uint64_t j = first; j < h; j++) o += valid[j];
    out[o] = tmp[h]; outRy[o] = tmpRy[h];
}
__global__ void k_scatter_big(const uint64_t *__restrict__ off, const uint8_t *__restrict__ valid, const AlnRec *__restrict__ tmp, const uint16_t *__restrict__ tmpRy, uint32_t n,
                              const uint64_t *__restrict__ outOff, AlnRec *__restrict__ out, uint16_t *__restrict__ outRy) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n || off[q + 1] - off[q] <= SCATTER_SMALL) return;
    uint64_t o = outOff[q];
    for (uint64_t h = off[q]; h < off[q + 1]; h++) if (valid[h]) { out[o] = tmp[h]; outRy[o] = tmpRy[h]; o++; }
}
__global__ void k_len_hist(const uint32_t *__restrict__ len, uint32_t n, uint32_t *__restrict__ present) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) present[len[i]] = 1;
}

}  // namespace

int cdm_rescore_impl(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_hits *hits, const cdm_rescore_params *par, cdm_alns **out) {
    hipStream_t s = ctx->stream;
    const uint32_t n = (uint32_t) db->n;
    const uint64_t nHits = hits->count;
    if (db->maxLen >= (1u << 30)) { cdm_set_error("cdm_rescore: sequence too long"); return CDM_ERR_UNSUPPORTED; }
    if (nHits >= 0xFFFFFF00ull) { cdm_set_error("cdm_rescore: %llu prefilter hits (a call takes fewer than 2^32)", (unsigned long long) nHits); return CDM_ERR_UNSUPPORTED; }
    DevBuf<unsigned int> undef;
    if (!undef.alloc(2)) { cdm_set_error("cdm_rescore: out of device memory"); return CDM_ERR_HIP; }
    CDM_HIP(hipMemsetAsync(undef.p, 0, 8, s));
    DevBuf<uint32_t> dPresent, owner; DevBuf<int32_t> dMin; DevBuf<AlnRec> tmp; DevBuf<uint16_t> tmpRy; DevBuf<uint8_t> valid; DevBuf<uint64_t> cnt;
    if (!dPresent.alloc(db->maxLen + 1) || !dMin.alloc(db->maxLen + 1) || !owner.alloc(nHits) || !tmp.alloc(nHits) || !tmpRy.alloc(nHits) || !valid.alloc(nHits) || !cnt.alloc((size_t) n + 1)) {
        cdm_set_error("cdm_rescore: out of device memory"); return CDM_ERR_HIP;
    }
    // E-value gate table for the lengths that occur (host ALP arithmetic, host/evalue.cpp)
    std::vector<uint32_t> present(db->maxLen + 1);
    CDM_HIP(hipMemsetAsync(dPresent.p, 0, (size_t) (db->maxLen + 1) * 4, s));
    hipLaunchKernelGGL(k_len_hist, dim3((n + 255) / 256), dim3(256), 0, s, db->len, n, dPresent.p);
    CDM_HIP(hipMemcpyAsync(present.data(), dPresent.p, (size_t) (db->maxLen + 1) * 4, hipMemcpyDeviceToHost, s));
    CDM_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> minScore(db->maxLen + 1, INT_MAX);
    for (uint32_t L = 1; L <= db->maxLen; L++) {
        if (!present[L]) continue;
        int lo = 0, hi = 2 * (int) L;   // E-value decreases with the score (tests/test_gpu_rescore.py checks it for the lengths used)
        if (!(cdm_evalue_host(hi, L, db->residues) <= par->eval_thr)) continue;
        while (lo < hi) { int mid = (lo + hi) / 2; if (cdm_evalue_host(mid, L, db->residues) <= par->eval_thr) hi = mid; else lo = mid + 1; }
        minScore[L] = lo;
    }
    CDM_HIP(hipMemcpyAsync(dMin.p, minScore.data(), (size_t) (db->maxLen + 1) * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_expand, dim3((n + 255) / 256), dim3(256), 0, s, hits->off, n, owner.p);
    DevBuf<SeqMeta> meta;
    MetaUniform uni;
    if (int rc = cdm_build_meta(ctx, db, &meta.p, &uni)) return rc;
    RescoreArgs a;
    cdmSetMeta(a, meta.p, uni); a.codes = db->codes; a.nmask = db->nmask; a.raw = db->raw; a.hoff = hits->off; a.hit = hits->rec;
    a.minScore = dMin.p; a.nHits = nHits; a.n = n; a.seqIdThr = par->seq_id_thr; a.covThr = par->cov_thr; a.covMode = par->cov_mode;
    a.minAlnLen = par->min_aln_len; a.tmp = tmp.p; a.tmpRy = tmpRy.p; a.valid = valid.p; a.undef = undef.p;
    hipEventRecord(ctx->ev0, s);
    if (nHits) hipLaunchKernelGGL(k_rescore, CDM_GRID((nHits + 255) / 256, 256), dim3(256), 0, s, a, owner.p);
    hipEventRecord(ctx->ev1, s);
    hipLaunchKernelGGL(k_count_valid, dim3((n + 255) / 256), dim3(256), 0, s, hits->off, valid.p, n, cnt.p);
    CDM_LAUNCH_CHECK();
    cdm_alns *res = new cdm_alns(); res->n = n;
    struct Guard { cdm_alns *&r; bool armed = true; ~Guard() { if (armed && r) { cdm_alns_free(r); r = nullptr; } } } guard{res};
    if (cdmMalloc(&res->off, ((size_t) n + 1) * 8) != hipSuccess) { cdm_set_error("cdm_rescore: out of device memory"); return CDM_ERR_HIP; }
    cdmscan::ScanTemp scanTmp;
    CDM_HIP(hipMemsetAsync(cnt.p + n, 0, 8, s));
    if (int rc = cdmscan::exclusiveScan<uint64_t>(s, scanTmp, cnt.p, res->off, (size_t) n + 1)) return rc;
    uint64_t total = 0; unsigned int nUndef = 0;
    CDM_HIP(hipMemcpyAsync(&total, res->off + n, 8, hipMemcpyDeviceToHost, s));
    CDM_HIP(hipMemcpyAsync(&nUndef, undef.p, 4, hipMemcpyDeviceToHost, s));
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_set_error("cdm_rescore: kernel failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    res->count = total; res->undefinedRecords = nUndef;
    if (cdmMalloc(&res->rec, (total + 1) * sizeof(AlnRec)) != hipSuccess || cdmMalloc(&res->ryMism, (total + 1) * sizeof(uint16_t)) != hipSuccess) { cdm_set_error("cdm_rescore: out of device memory"); return CDM_ERR_HIP; }
    res->rySerial = db->serial;
    if (nHits) hipLaunchKernelGGL(k_scatter, CDM_GRID((nHits + 255) / 256, 256), dim3(256), 0, s, hits->off, (const uint32_t *) owner.p, valid.p, tmp.p, tmpRy.p, (uint64_t) nHits, res->off, res->rec, res->ryMism);
    hipLaunchKernelGGL(k_scatter_big, dim3((n + 255) / 256), dim3(256), 0, s, hits->off, valid.p, tmp.p, tmpRy.p, n, res->off, res->rec, res->ryMism);
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_set_error("cdm_rescore: compaction failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    hipEventElapsedTime(&ctx->lastMs[1], ctx->ev0, ctx->ev1);
    guard.armed = false;
    *out = res;
    return CDM_OK;
}
