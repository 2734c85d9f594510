// rescorediagonal in the mode linclust's pre-clustering step runs it on the assembled contigs (lib/mmseqs/data/workflow/linclust.sh:27-31,
// src/workflow/GuidedNuclassembler.cpp:176-181): --rescore-mode 0 (Hamming: the score of a diagonal is the number of positions with
// the same LETTER) with --wrapped-scoring 1 (the query is doubled, so that a circular contig cut at another place still lines up with
// its copy), query DB == target DB.  Replaces the loop of lib/mmseqs/src/alignment/rescorediagonal.cpp:145-356 for that mode;
// DistanceCalculator::computeUngappedWrappedAlignment (DistanceCalculator.h:57-91), computeInverseHammingDistance (:276-296).
//
// One wave per prefilter hit: the probed diagonals are the reference's (its loop conditions are unsigned arithmetic and are kept as
// written), the lanes share the columns of a probe.  Letters are the DB's original bytes (raw plane where a sequence has one, else
// ACGT / N from the codes); the reverse query is what the module spells out for a hit of negative score - the complement of what
// NucleotideMatrix maps each letter to, 'X' for the X class (rescorediagonal.cpp:171-177).
// Output: prefilter records (target, 100 * seq. id. with the hit's strand as its sign, diagonal) for the hits that pass the
// coverage / seq. id. / length criteria, and every identity hit; same CSR over the queries as the input.
#include <algorithm>
#include "common.h"
#include "devutil.h"
#include "scan.h"

namespace {

struct HamArgs {
    const uint32_t *woff, *len; const uint8_t *hasN; const uint32_t *codes, *nmask; const uint8_t *raw;
    const uint64_t *hoff; const HitRec *hit; uint32_t n; uint64_t count, first;       // first: first hit of this launch (slices: common.h cdmSliceItems)
    float seqIdThr, covThr; int covMode, seqIdMode, minAlnLen; int evalOk; int revPref;
    HitRec *tmp; uint32_t *valid;       // [count] record as it would be written, 1 = kept
    unsigned int *flags;                // [0]: an empty target sequence (the reference's probe loop does not end there)
};

__device__ __forceinline__ bool hamCanBeCovered(float covThr, int covMode, float ql, float tl) {        // Util.cpp:533-550
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
__device__ __forceinline__ bool hamHasCoverage(float covThr, int covMode, float qc, float tc) {             // Util.cpp:552-567
    switch (covMode) { case 0: return qc >= covThr && tc >= covThr; case 2: return qc >= covThr; case 1: return tc >= covThr; default: return true; }
}
// the sequence's letter at p as the DB file holds it
__device__ __forceinline__ uint32_t hamLetter(const HamArgs &a, uint32_t w, bool rawRow, uint32_t p) {
    if (rawRow) return cdm_raw_at(a.raw, w, p);
    return cdm_isN(a.nmask, w, p) ? (uint32_t) 'N' : (uint32_t) "ACGT"[cdm_base(a.codes, w, p)];
}
// letter x of the reversed query: num2aa[reverseResidue(aa2num[letter at p])]
__device__ __forceinline__ uint32_t hamRevLetter(const HamArgs &a, uint32_t w, uint32_t p) {
    return cdm_isN(a.nmask, w, p) ? (uint32_t) 'X' : (uint32_t) "TGCA"[cdm_base(a.codes, w, p)];
}

__global__ __launch_bounds__(256) void k_hamming(HamArgs a) {
    const uint64_t h = a.first + (uint64_t) blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (h >= a.count) return;
    // query of the hit: the last q with hoff[q] <= h
    uint32_t lo = 0, hi = a.n;
    while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (a.hoff[mid] <= h) lo = mid; else hi = mid; }
    const uint32_t q = lo;
    const HitRec hr = a.hit[h];
    const uint32_t t = hr.target;
    const uint32_t L = a.len[q], dbLen = a.len[t];
    const bool isIdentity = q == t;
    bool keep = false; HitRec out; out.target = t; out.score = 0; out.diagonal = 0;
    if (dbLen == 0 || L == 0) { if (lane == 0) { a.flags[0] = 1u; a.valid[h] = 0; } return; }
    if (hamCanBeCovered(a.covThr, a.covMode, (float) L, (float) dbLen) && !(dbLen > L)) {
        const bool isReverse = a.revPref && hr.score < 0;
        const uint32_t qw = a.woff[q], tw = a.woff[t];
        const bool qRaw = a.raw && (a.hasN[q] & 2u), tRaw = a.raw && (a.hasN[t] & 2u);
        const unsigned short diagonal = (unsigned short) hr.diagonal;
        const unsigned int dbSeqLen = (unsigned int) (float) dbLen;        // (the module passes the length as a float, :222)
        const uint32_t m = min(dbSeqLen, L);
        unsigned int best = 0; int bestDiag = 0;
        auto probe = [&](int realDiagonal) {
            // the doubled query from realDiagonal on against the target from 0 on, m columns
            unsigned int c = 0;
            for (uint32_t j = lane; j < m; j += 64) {
                const uint32_t x = (uint32_t) realDiagonal + j;        // index into the doubled (reversed) query, < 2 L
                uint32_t ql;
                if (!isReverse) ql = hamLetter(a, qw, qRaw, x >= L ? x - L : x);
                else { const uint32_t y = 2 * L - 1 - x; ql = hamRevLetter(a, qw, y >= L ? y - L : y); }
                c += (ql == hamLetter(a, tw, tRaw, j)) ? 1u : 0u;
            }
            c = (unsigned int) cdm_wave_sum((int) c);
            if (c > best) { best = c; bestDiag = realDiagonal; }
        };
        for (unsigned int devisions = 1; (-devisions * 65536 + diagonal) > -dbSeqLen; devisions++) probe((int) (-devisions * 65536 + diagonal) + (int) L);
        for (unsigned int devisions = 0; (devisions * 65536 + diagonal) < L; devisions++) probe((int) (devisions * 65536 + diagonal));
        const int diagonalLen = (int) m;
        const float targetCov = static_cast<float>(diagonalLen) / static_cast<float>(dbLen), queryCov = static_cast<float>(diagonalLen) / static_cast<float>(L);
        const int idCnt = (int) (static_cast<float>(best));
        float sidF;
        switch (a.seqIdMode) {                                                                                   // Util.cpp:588-598
            case 1: sidF = static_cast<float>(idCnt) / static_cast<float>(min((int) L, (int) dbLen)); break;
            case 2: sidF = static_cast<float>(idCnt) / static_cast<float>(max((int) L, (int) dbLen)); break;
            default: sidF = static_cast<float>(idCnt) / static_cast<float>(diagonalLen); break;
        }
        const double seqId = sidF;
        const bool hasCov = hamHasCoverage(a.covThr, a.covMode, queryCov, targetCov);
        const bool hasSeqId = seqId >= (double) (a.seqIdThr - 1.1920928955078125e-07f);
        const bool hasAlnLen = diagonalLen >= a.minAlnLen;
        keep = isIdentity || (hasAlnLen && hasCov && hasSeqId && a.evalOk);
        int score = (int) (100 * seqId);
        out.score = isReverse ? -score : score;
        out.diagonal = (int) (short) (unsigned short) bestDiag;
    }
    if (lane == 0) { a.tmp[h] = out; a.valid[h] = keep ? 1u : 0u; }
}
__global__ void k_ham_compact(const uint32_t *__restrict__ valid, const uint32_t *__restrict__ pos, const HitRec *__restrict__ tmp, uint64_t count, HitRec *__restrict__ rec) {
    const uint64_t h = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (h < count && valid[h]) rec[pos[h]] = tmp[h];
}
__global__ void k_ham_offsets(const uint64_t *__restrict__ hoff, const uint32_t *__restrict__ pos, uint32_t n, uint64_t *__restrict__ off) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q <= n) off[q] = pos[hoff[q]];
}

}  // namespace

int cdm_rescore_hamming_impl(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_hits *hits, const cdm_hamming_params *par, cdm_hits **out) {
    hipStream_t s = ctx->stream;
    const uint32_t n = (uint32_t) db->n;
    const uint64_t count = hits->count;
    if (count >= 0xFFFFFFF0ull) { cdm_set_error("cdm_rescore_hamming: more than 2^32 prefilter hits"); return CDM_ERR_UNSUPPORTED; }
    cdm_hits *o = new cdm_hits();
    o->n = n;
    DevBuf<HitRec> tmp; DevBuf<uint32_t> valid, pos; DevBuf<unsigned int> flags;
    if (cdmMalloc(&o->off, ((size_t) n + 1) * 8) != hipSuccess || !tmp.alloc(count + 1) || !valid.alloc(count + 1) || !pos.alloc(count + 1) || !flags.alloc(1)) {
        cdm_hits_free(o); cdm_set_error("cdm_rescore_hamming: out of device memory"); return CDM_ERR_HIP;
    }
    hipMemsetAsync(flags.p, 0, 4, s);
    hipMemsetAsync(valid.p, 0, (count + 1) * 4, s);
    HamArgs a;
    a.woff = db->woff; a.len = db->len; a.hasN = db->hasN; a.codes = db->codes; a.nmask = db->nmask; a.raw = db->raw;
    a.hoff = hits->off; a.hit = hits->rec; a.n = n; a.count = count;
    a.seqIdThr = par->seq_id_thr; a.covThr = par->cov_thr; a.covMode = par->cov_mode; a.seqIdMode = par->seq_id_mode; a.minAlnLen = par->min_aln_len;
    a.evalOk = (0.0 <= par->eval_thr) ? 1 : 0; a.revPref = par->reverse_prefilter ? 1 : 0;
    a.tmp = tmp.p; a.valid = valid.p; a.flags = flags.p;
    for (uint64_t first = 0, slice = cdmSliceItems(64); first < count; first += slice) {
        a.first = first;
        hipLaunchKernelGGL(k_hamming, CDM_GRID((std::min(slice, count - first) + 3) / 4, 256), dim3(256), 0, s, a);
    }
    cdmscan::ScanTemp st;
    if (int rc = cdmscan::exclusiveScan<uint32_t>(s, st, valid.p, pos.p, (size_t) count + 1)) { cdm_hits_free(o); return rc; }
    uint32_t kept = 0; unsigned int hflags = 0;
    hipMemcpyAsync(&kept, pos.p + count, 4, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(&hflags, flags.p, 4, hipMemcpyDeviceToHost, s);
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_hits_free(o); cdm_set_error("cdm_rescore_hamming: kernel failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    if (hflags) { cdm_hits_free(o); cdm_set_error("cdm_rescore_hamming: an empty sequence among the hits (the reference's probe loop does not end on one)"); return CDM_ERR_UNSUPPORTED; }
    o->count = kept;
    if (cdmMalloc(&o->rec, ((size_t) kept + 1) * sizeof(HitRec)) != hipSuccess) { cdm_hits_free(o); cdm_set_error("cdm_rescore_hamming: out of device memory"); return CDM_ERR_HIP; }
    if (count) hipLaunchKernelGGL(k_ham_compact, CDM_GRID((count + 255) / 256, 256), dim3(256), 0, s, (const uint32_t *) valid.p, (const uint32_t *) pos.p, (const HitRec *) tmp.p, count, o->rec);
    hipLaunchKernelGGL(k_ham_offsets, dim3(n / 256 + 1), dim3(256), 0, s, (const uint64_t *) hits->off, (const uint32_t *) pos.p, n, o->off);
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_hits_free(o); cdm_set_error("cdm_rescore_hamming: compaction failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    *out = o;
    return CDM_OK;
}
