// ancient_correction on the device.
//
// Replaces the OpenMP loop of doCorrection (src/assembler/correction.cpp:200-476) and mostLikeliBaseRead (:7-123).
// One wavefront (64 lanes) owns one query that has more than one alignment record; queries with only their self
// alignment have coverage <= 1 everywhere and keep their bases (:418-420) -- for them the output DB is a plain copy.
//
// Per query the wave runs three passes over the query's alignment records (CSR slice):
//   pass 0  avCov = sum(alnLength) / qLen                                              (:218-245)
//   pass 1  per record, all 64 lanes over the aligned columns: RY-space identity (nuclassembleUtil.cpp:78-92), the
//           rymer threshold (:297-301), the right/left/inside class gate (:311-322) and the pile-up gate (:359).
//           The accept bit goes to a scratch byte per record.
//   pass 2  per chunk of 64 query positions (lane = position): pile up the accepted targets into 44 (tBase, damage
//           class) counters per position (:361-385), held in LDS as counter[slot][lane] = total | reverse << 16,
//           then call the base: coverage <= 1 keeps the base, otherwise arg-max over four log-likelihoods that are
//           accumulated in software x87 extended precision in the reference's term order (:80-122), so ties and
//           near-ties resolve exactly as `long double` does on the host.
// All logs come from host-built tables (DamageLut) staged in LDS.
#include "common.h"
#include "devutil.h"

namespace {

struct CorrectArgs {
    MetaWoff woff; MetaLen len; MetaHasN hasN; MetaExt ext; MetaRaw hasRaw;      // per-sequence metadata, one record per sequence
    const uint32_t *codes, *nmask; const uint8_t *raw; uint8_t *outRaw;      // raw: original letters (general kernel only), or NULL
    const uint64_t *aoff;
    const AlnRec *rec;
    const uint16_t *ry;       // purine/pyrimidine mismatches per record from cdm_rescore (0xFFFF = count here), or NULL
    const uint32_t *active;
    const unsigned int *nActive;
    uint8_t *accept;          // [alignment count] scratch
    unsigned int *errFlag;    // (unused: the general kernel's counters are 64 bit)
    uint32_t *outCodes, *outNmask;
    const DamageLut *lut;
    float seqIdThr, corrRy;
};

constexpr int WAVES_PER_BLOCK = 2;      // general kernel: 64-bit counters (any number of records piles up), 22.5 KB of LDS per wave
constexpr int SLOTS = 44;

__device__ __forceinline__ uint32_t alnLength(const AlnRec &r) {   // Matcher::computeAlnLength, M/alignment/Matcher.cpp:204-206
    int a = abs(r.qEnd - r.qStart), b = abs(r.dbEnd - r.dbStart);
    return (uint32_t) max(a, b) + 1u;
}
// oriented copy of a record (correction.cpp:229-242)
struct Oriented { int qs, qe, ds, de; bool rev; };
__device__ __forceinline__ Oriented orient(const AlnRec &r, uint32_t dbLen) {
    Oriented o;
    if (r.qStart > r.qEnd) { o.qs = r.qEnd; o.qe = r.qStart; o.ds = (int) dbLen - r.dbEnd - 1; o.de = (int) dbLen - r.dbStart - 1; o.rev = true; }
    else { o.qs = r.qStart; o.qe = r.qEnd; o.ds = r.dbStart; o.de = r.dbEnd; o.rev = false; }
    return o;
}
// base of the oriented target (reverse complement when rev), with N -> 'A' (0) as nucleotideMap/ryMap do for any
// letter outside ACGT; getNuclRevFragment maps X to 'N', which those maps also send to 0 (nuclassembleUtil.cpp:67-76)
__device__ __forceinline__ uint32_t targetBase(const CorrectArgs &a, uint32_t tw, uint32_t tLen, bool tHasN, bool rev, uint32_t tpos) {
    uint32_t p = rev ? (tLen - 1u - tpos) : tpos;
    uint32_t c = cdm_base(a.codes, tw, p);
    if (tHasN && cdm_isN(a.nmask, tw, p)) return 0u;
    return rev ? (3u - c) : c;
}
// the same for a target that may carry letters beyond ACGTN (general kernel): forward, nucleotideMap[original byte] ('c' is 0);
// reversed, the complement of what NucleotideMatrix maps the letter to ('c' -> G), as getNuclRevFragment spells it
__device__ __forceinline__ uint32_t targetBaseRaw(const CorrectArgs &a, uint32_t t, uint32_t tw, uint32_t tLen, bool tHasN, bool rev, uint32_t tpos) {
    if (tHasN && !rev && a.raw && a.hasRaw[t]) return cdm_raw_base(cdm_raw_at(a.raw, tw, tpos));
    return targetBase(a, tw, tLen, tHasN, rev, tpos);
}

// exact arg-max in software x87, one accumulator at a time (keeps the register footprint of the callers small)
template <typename F>
__device__ __noinline__ uint32_t callBaseExact(const double *lt, const double *lq, const double *sLogD, F counts, uint64_t mask) {
    X87 bestAcc = x87_zero(); int best = 0;
    for (int qq = 0; qq < 4; qq++) {
        X87 acc = x87_zero();
#pragma unroll 1
        for (uint64_t m = mask; m; m &= m - 1) {        // ascending slots, the reference's loop order (:60-110)
            const int slot = __ffsll((unsigned long long) m) - 1;
            const uint64_t v = counts(slot);
            const int c = (int) (uint32_t) v, nr = (int) (v >> 32);
            if (c == 0) continue;
            const int tb = slot / 11, l = slot - tb * 11;
            const double base2 = __dadd_rn(lt[tb], lq[qq]);
            const double f = __dadd_rn(base2, sLogD[((0 * 11 + l) * 4 + qq) * 4 + tb]);
            const double g = __dadd_rn(base2, sLogD[((1 * 11 + l) * 4 + qq) * 4 + tb]);
            acc = x87_add(acc, x87_from_double(__dmul_rn((double) (c - nr), f)));
            acc = x87_add(acc, x87_from_double(__dmul_rn((double) nr, g)));
        }
        if (qq == 0 || x87_lt(bestAcc, acc)) { bestAcc = acc; best = qq; }   // first maximum wins (:119-122)
    }
    return (uint32_t) best;
}

// mostLikeliBaseRead (src/assembler/correction.cpp:7-123) for one query position.  counts(slot) returns
// total | (uint64) reverse << 32 for slot = tBase * 11 + damage class; mask has a bit for every slot that may be non-zero (a pile-up
// touches a handful of the 44).  keep is set when coverage <= 1 (:418-420).
constexpr uint64_t ALL_SLOTS = (1ull << 44) - 1ull;
template <typename F>
__device__ __forceinline__ uint32_t callBase(const double *sLogT, const double *sLogQ, const double *sLogD, uint32_t qb, uint32_t p, uint32_t qLen,
                                             bool qWasExt, F counts, uint64_t mask, bool &keep, uint32_t cov4 = ~0u) {
    // cov4: records per target base, 8 bits each (callers with at most 64 records count them while piling up); ~0 = sum the slots here
    uint32_t cov[4];
    if (cov4 == ~0u) {
        cov[0] = cov[1] = cov[2] = cov[3] = 0;
#pragma unroll 1
        for (uint64_t m = mask; m; m &= m - 1) {
            const int slot = __ffsll((unsigned long long) m) - 1;
            const uint32_t c = (uint32_t) counts(slot);
            const int tb = slot / 11;
            cov[0] += tb == 0 ? c : 0u; cov[1] += tb == 1 ? c : 0u; cov[2] += tb == 2 ? c : 0u; cov[3] += tb == 3 ? c : 0u;
        }
    } else { cov[0] = cov4 & 0xFFu; cov[1] = (cov4 >> 8) & 0xFFu; cov[2] = (cov4 >> 16) & 0xFFu; cov[3] = cov4 >> 24; }
    const uint32_t total = cov[0] + cov[1] + cov[2] + cov[3];
    keep = total <= 1;
    if (keep) return qb;
    if (!qWasExt) {
        // :27-33 compares double(cov) / double(total) with 0.4.  In integers: a / b >= 2 / 5.  The double 0.4 lies 2.2e-17 above 2/5, less
        // than half an ulp, so the rounded quotient of a / b == 2/5 IS that double (>= holds); any other a / b with b < 2^18 is further
        // than 1 / (5 b) > 7e-7 from 2/5, far beyond the rounding - the two tests agree on every input.
        if (5u * cov[3] >= 2u * total || 5u * cov[0] >= 2u * total) return qb;
    }
    int qcls;
    if (qWasExt) qcls = 11;
    else if (p < 5) qcls = (int) p;
    else if (p >= qLen - 5) qcls = 11 - (int) (qLen - p);
    else qcls = 5;
    const double *lq = &sLogQ[(qcls * 4 + qb) * 4];
    const double *lt = &sLogT[qb * 4];
    // ---- decision in plain double first.  All addends are <= 0, so each of the four sums (at most 88 addends) differs from
    // the reference's long double sum by less than 88 * 2^-52 * |sum|; when the largest sum beats every other one by more
    // than that, the long double arg-max is the same and the extended-precision emulation is skipped.  Near ties and exact
    // ties (the first maximum wins) go through callBaseExact.
    double sd[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (uint64_t m = mask; m; m &= m - 1) {
        const int slot = __ffsll((unsigned long long) m) - 1;
        const uint64_t v = counts(slot);
        const int c = (int) (uint32_t) v, nr = (int) (v >> 32);
        if (c == 0) continue;
        const int tb = slot / 11, l = slot - tb * 11;
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            const double base2 = __dadd_rn(lt[tb], lq[qq]);
            const double f = __dadd_rn(base2, sLogD[((0 * 11 + l) * 4 + qq) * 4 + tb]);
            const double g = __dadd_rn(base2, sLogD[((1 * 11 + l) * 4 + qq) * 4 + tb]);
            sd[qq] = __dadd_rn(sd[qq], __dmul_rn((double) (c - nr), f));
            sd[qq] = __dadd_rn(sd[qq], __dmul_rn((double) nr, g));
        }
    }
    int bestD = 0;
#pragma unroll
    for (int qq = 1; qq < 4; qq++) if (sd[bestD] < sd[qq]) bestD = qq;
    bool clear = true;
#pragma unroll
    for (int qq = 0; qq < 4; qq++)
        if (qq != bestD) clear = clear && (sd[bestD] - sd[qq] > 1e-12 * (fabs(sd[bestD]) + fabs(sd[qq])) + 1e-300);
    if (clear) return (uint32_t) bestD;
    return callBaseExact(lt, lq, sLogD, counts, mask);
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_correct(CorrectArgs a) {
    __shared__ double sLogT[16], sLogQ[12 * 16], sLogD[2 * 11 * 16];
    __shared__ uint64_t sCnt[WAVES_PER_BLOCK][SLOTS][64];       // total | reverse << 32
    for (int i = threadIdx.x; i < 16; i += blockDim.x) sLogT[i] = (&a.lut->logT[0][0])[i];
    for (int i = threadIdx.x; i < 12 * 16; i += blockDim.x) sLogQ[i] = (&a.lut->logQ[0][0][0])[i];
    for (int i = threadIdx.x; i < 2 * 11 * 16; i += blockDim.x) sLogD[i] = (&a.lut->logD[0][0][0][0])[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int nAct = *a.nActive;
    uint64_t (*cnt)[64] = sCnt[wave];

    for (unsigned int item = blockIdx.x * WAVES_PER_BLOCK + wave; item < nAct; item += gridDim.x * WAVES_PER_BLOCK) {
        const uint32_t q = a.active[item];
        const uint32_t qLen = a.len[q], qw = a.woff[q];
        const bool qHasN = a.hasN[q] != 0;
        const bool qRaw = qHasN && a.raw && a.hasRaw[q] != 0;
        const bool qWasExt = a.ext[q] != 0;
        const uint64_t r0 = a.aoff[q], r1 = a.aoff[q + 1];

        // ---- pass 0: average coverage (float sum of small integers: exact, order independent)
        int sumLen = 0;
        for (uint64_t r = r0 + lane; r < r1; r += 64) sumLen += (int) alnLength(a.rec[r]);
        sumLen = cdm_wave_sum(sumLen);
        const float avCov = static_cast<float>(static_cast<float>(sumLen)) / qLen;

        // ---- pass 1: gates
        for (uint64_t r = r0; r < r1; r++) {
            const AlnRec rec = a.rec[r];
            const uint32_t t = rec.target;
            const uint32_t tLen = a.len[t], tw = a.woff[t];
            const bool tHasN = a.hasN[t] != 0;
            const uint32_t aLen = alnLength(rec);
            const Oriented o = orient(rec, tLen);
            bool ok = a.ext[t] == 0;                              // target must be a read (:280-284)
            int mism = 0;
            const uint32_t preRy = a.ry ? a.ry[r] : 0xFFFFu;
            if (ok && preRy != 0xFFFFu) mism = (int) preRy;         // counted by cdm_rescore on the same columns
            if (ok) {
                if (preRy == 0xFFFFu) {
                for (uint32_t c = lane; c < aLen; c += 64) {
                    uint32_t qb = cdm_base(a.codes, qw, o.qs + c);
                    if (qHasN && cdm_isN(a.nmask, qw, o.qs + c)) qb = 0;
                    if (qRaw) qb = cdm_raw_base(cdm_raw_at(a.raw, qw, o.qs + c));
                    uint32_t tb = targetBaseRaw(a, t, tw, tLen, tHasN, o.rev, o.ds + c);
                    mism += ((qb & 1u) != (tb & 1u));             // RY class = low bit of the A,C,G,T = 0..3 code
                }
                mism = cdm_wave_sum(mism);
                }
                const float ryId = static_cast<float>(aLen - (uint32_t) mism) / static_cast<float>(aLen);
                float thr = a.corrRy;
                if (aLen <= 100) { thr = (static_cast<float>(aLen) - 1) / static_cast<float>(aLen); thr = floorf(thr * 1000) / 1000; }
                const bool ry = ryId >= thr;
                const bool right = o.ds == 0 && (uint32_t) o.qe == (qLen - 1);
                const bool left = o.qs == 0 && (uint32_t) o.de == (tLen - 1);
                const bool cls = right || left || (avCov < 50);
                ok = ry && cls && rec.seqId >= a.seqIdThr && aLen >= 30;
            }
            if (lane == 0) __hip_atomic_store(&a.accept[r], (uint8_t) (ok ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // accept[] is read back by this same wave below: drain the stores, and read with agent-scope (L1-bypassing) loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();

        // ---- pass 2: pile-up + call, 64 positions at a time
        const uint32_t lastWord = (qLen + 15) / 16;
        for (uint32_t base = 0; base < qLen; base += 64) {
            const uint32_t p = base + lane;
#pragma unroll 4
            for (int s = 0; s < SLOTS; s++) cnt[s][lane] = 0;
            for (uint64_t r = r0; r < r1; r++) {
                if (!__hip_atomic_load(&a.accept[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
                const AlnRec rec = a.rec[r];
                const uint32_t t = rec.target, tLen = a.len[t], tw = a.woff[t];
                const Oriented o = orient(rec, tLen);
                if ((uint32_t) o.qe < base || (uint32_t) o.qs >= base + 64) continue;   // wave uniform
                if (p >= (uint32_t) o.qs && p <= (uint32_t) o.qe && p < qLen) {
                    const uint32_t tpos = (uint32_t) o.ds + (p - (uint32_t) o.qs);
                    const uint32_t tb = targetBaseRaw(a, t, tw, tLen, a.hasN[t] != 0, o.rev, tpos);
                    const uint32_t cls = tpos < 5 ? tpos : (tpos >= tLen - 5 ? 6 + (tpos - (tLen - 5)) : 5);
                    cnt[tb * 11 + cls][lane] += 1ull + (o.rev ? (1ull << 32) : 0ull);
                }
            }
            // ---- call
            uint32_t newCode = 0; bool keep = true;
            if (p < qLen) {
                uint32_t qb = cdm_base(a.codes, qw, p);
                const bool qIsN = qHasN && cdm_isN(a.nmask, qw, p);
                if (qIsN) qb = 0;
                const uint32_t mapped = qb;
                uint8_t orig = 0;
                if (qRaw) { orig = cdm_raw_at(a.raw, qw, p); qb = cdm_raw_base(orig); }
                newCode = callBase(sLogT, sLogQ, sLogD, qb, p, qLen, qWasExt, [&](int slot) { return cnt[slot][lane]; }, ALL_SLOTS, keep);
                if (qRaw) {     // coverage <= 1 keeps the original byte (:418-420) and with it what the letter maps to
                    if (keep) newCode = mapped;
                    a.outRaw[(uint64_t) qw * 16u + p] = keep ? orig : (uint8_t) "ACGT"[newCode];
                }
            }
            // ---- write 64 positions = 4 code words (+ N bits): lanes 0..3 assemble one word each from ballots
            const uint64_t b0 = cdm_ballot((newCode & 1u) != 0), b1 = cdm_ballot((newCode & 2u) != 0);
            uint64_t nb = 0;
            if (qHasN) nb = cdm_ballot(p < qLen && keep && cdm_isN(a.nmask, qw, p));
            if (lane < 4) {
                const uint32_t w = (base >> 4) + lane;
                if (w < lastWord) {
                    const uint32_t lo = (uint32_t) (b0 >> (16 * lane)), hi = (uint32_t) (b1 >> (16 * lane));
                    a.outCodes[qw + w] = cdm_spread16(lo) | (cdm_spread16(hi) << 1);
                    if (qHasN) reinterpret_cast<uint16_t *>(a.outNmask)[qw + w] = (uint16_t) (nb >> (16 * lane));
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Fast kernel for queries with at most 64 alignment records (all but the heaviest pile-ups): lane = record in the gate
// phase (RY identity on 16-base words, XOR + popcount) and keeps the record, lane = position in the pile-up.
// Same arithmetic as k_correct; 8-bit (4-bit) pile-up counters (a slot cannot exceed the 64 (15) records).
constexpr int FAST_WAVES = 4;
// Target words staged in LDS per record (lane = record loads the words its aligned span covers, all records at once): the pile-up
// then takes its letters from LDS instead of one dependent global load per record and 64 positions.  12 words hold a span of 177
// columns wherever it starts; longer spans and targets with N keep the global path.
constexpr int STAGE_WORDS = 12;

// MAXREC: most records of a query on this instance (64, or 15 with CT = uint8_t: two 4-bit fields per counter and a quarter of the
// record slots - less LDS per block, more waves per CU; the kernel is bound by the latency of its dependent gathers, so the
// occupancy is what it runs on).  MINW: waves per SIMD the register allocation has to leave room for.
template <int MAXREC, typename CT, int MINW>
__global__ __launch_bounds__(64 * FAST_WAVES, MINW) void k_correct_fast(CorrectArgs a, const uint32_t *__restrict__ list, const unsigned int *__restrict__ nList) {
    constexpr int HB = sizeof(CT) * 4;                  // bits per field: total | reverse << HB
    constexpr int PER = 4 / sizeof(CT);                 // counters per LDS dword
    __shared__ double sLogT[16], sLogQ[12 * 16], sLogD[2 * 11 * 16];
    __shared__ uint32_t sCnt[FAST_WAVES][SLOTS * (64 / PER)];      // counter [slot][lane], PER lanes to a word
    __shared__ uint32_t sStage[FAST_WAVES][(MAXREC == 64 ? 64 : 16) * STAGE_WORDS];
    for (int i = threadIdx.x; i < 16; i += blockDim.x) sLogT[i] = (&a.lut->logT[0][0])[i];
    for (int i = threadIdx.x; i < 12 * 16; i += blockDim.x) sLogQ[i] = (&a.lut->logQ[0][0][0])[i];
    for (int i = threadIdx.x; i < 2 * 11 * 16; i += blockDim.x) sLogD[i] = (&a.lut->logD[0][0][0][0])[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int nItems = *nList;
    // the pile-up adds to a counter with ONE LDS atomic on the word that holds it (the lanes that share the word are serialised by
    // the LDS): word index of this lane's counter inside a slot row, and the lane's increments shifted to its place in the word
    uint32_t *cntWords = sCnt[wave], *stage = sStage[wave];
    const uint32_t laneWord = (uint32_t) lane / PER, laneShift = ((uint32_t) lane % PER) * (8 * sizeof(CT));
    const uint32_t incFwd = 1u << laneShift, incRev = (1u | (1u << HB)) << laneShift;
    const uint32_t laneClear = ~((uint32_t) (CT) ~(CT) 0 << laneShift);
    for (int s = lane; s < SLOTS * (64 / PER); s += 64) cntWords[s] = 0;       // a lane clears the slots it touched after every call
    // The chain list -> record offsets + query metadata -> records is walked one query AHEAD: the next query's offsets and metadata are
    // requested at the top of an iteration, its records (one per lane) and their RY counts once the gate of the current query is through,
    // so that they arrive while the pile-up runs.  (Wave-uniform values are moved to scalar registers.)
    const unsigned int stride = gridDim.x * FAST_WAVES;
    unsigned int item = blockIdx.x * FAST_WAVES + wave;
    const SeqMeta *meta = a.len.m;
    auto uni = [](uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); };
    uint32_t qCur = 0, qNext = 0; SeqMeta qmCur = {0, 0, 0, 0}; uint64_t r0Cur = 0; uint32_t nRecCur = 0; AlnRec recCur = {}; uint32_t ryCur = 0xFFFFu;
    if (item < nItems) {
        qCur = uni(list[item]); qmCur = meta[qCur]; r0Cur = a.aoff[qCur]; nRecCur = (uint32_t) (a.aoff[qCur + 1] - r0Cur);
        if ((uint32_t) lane < nRecCur) { recCur = a.rec[r0Cur + lane]; if (a.ry) ryCur = a.ry[r0Cur + lane]; }
        if (item + stride < nItems) qNext = uni(list[item + stride]);
    }
    for (; item < nItems; item += stride) {
        const uint32_t q = qCur;
        const uint32_t qLen = uni(qmCur.len), qw = uni(qmCur.woff);
        const bool qHasN = (uni(qmCur.flags) & 1u) != 0, qWasExt = (uni(qmCur.flags) & 2u) != 0;
        const uint64_t r0 = ((uint64_t) uni((uint32_t) (r0Cur >> 32)) << 32) | uni((uint32_t) r0Cur);
        const uint32_t nRec = uni(nRecCur);
        const uint32_t qLast = (qLen + 15) / 16 - 1;
        const bool hasNext = item + stride < nItems;
        const uint32_t qN = qNext;
        SeqMeta qmN = {0, 0, 0, 0}; uint64_t r0N = 0, r1N = 0;
        if (hasNext) { qmN = meta[qN]; r0N = a.aoff[qN]; r1N = a.aoff[qN + 1]; }
        if (item + 2 * stride < nItems) qNext = uni(list[item + 2 * stride]);
        // ---- gate phase, lane = record
        const AlnRec rec = recCur; uint32_t aLen = 0;
        const uint32_t ryOfRec = ryCur;
        const bool have = (uint32_t) lane < nRec;
        if (have) aLen = alnLength(rec);
        const float avCov = static_cast<float>(static_cast<float>(cdm_wave_sum((int) aLen))) / qLen;
        bool ok = false;
        uint32_t iTw = 0, iLenFlags = 0, iQs = 0, iSpan = 0, iDs = 0;     // the record as the pile-up needs it, kept in this lane's registers
        uint32_t iW0 = 0;                                                 // bit 31: the span's words are staged, from this word of the target on
        if (have) {
            const uint32_t t = rec.target, tLen = a.len[t], tw = a.woff[t];
            const bool tHasN = a.hasN[t] != 0;
            const Oriented o = orient(rec, tLen);
            ok = a.ext[t] == 0;
            if (ok) {
                uint32_t mism = ryOfRec;
                if (mism != 0xFFFFu) {
                    // counted by cdm_rescore on the same columns
                } else if (!qHasN && !tHasN) {
                    mism = 0;
                    const uint32_t tLast = (tLen + 15) / 16 - 1;
                    for (uint32_t c = 0; c < aLen; c += 16) {
                        const uint32_t x = cdm_window16(a.codes, qw, (uint32_t) o.qs + c, qLast) ^ cdm_oriented_window16(a.codes, tw, tLen, tLast, o.rev, (uint32_t) o.ds + c);
                        uint32_t mm = x & 0x55555555u;                   // RY class = low bit of the code
                        const uint32_t rem = aLen - c;
                        if (rem < 16) mm &= (1u << (2 * rem)) - 1u;
                        mism += __popc(mm);
                    }
                } else {
                    mism = 0;
                    for (uint32_t c = 0; c < aLen; c++) {
                        uint32_t qb = cdm_base(a.codes, qw, o.qs + c);
                        if (qHasN && cdm_isN(a.nmask, qw, o.qs + c)) qb = 0;
                        const uint32_t tb = targetBase(a, tw, tLen, tHasN, o.rev, o.ds + c);
                        mism += ((qb & 1u) != (tb & 1u));
                    }
                }
                const float ryId = static_cast<float>(aLen - mism) / static_cast<float>(aLen);
                float thr = a.corrRy;
                if (aLen <= 100) { thr = (static_cast<float>(aLen) - 1) / static_cast<float>(aLen); thr = floorf(thr * 1000) / 1000; }
                const bool right = o.ds == 0 && (uint32_t) o.qe == (qLen - 1);
                const bool left = o.qs == 0 && (uint32_t) o.de == (tLen - 1);
                ok = (ryId >= thr) && (right || left || (avCov < 50)) && rec.seqId >= a.seqIdThr && aLen >= 30;
            }
            // sequences on this path are shorter than 2^30 letters (the DB's word offsets are 32 bit)
            iTw = tw; iLenFlags = tLen | (o.rev ? 0x80000000u : 0u) | (tHasN ? 0x40000000u : 0u); iQs = (uint32_t) o.qs; iSpan = (uint32_t) (o.qe - o.qs); iDs = (uint32_t) o.ds;
            if (ok && !tHasN) {
                const uint32_t fLo = o.rev ? tLen - 1u - (iDs + iSpan) : iDs;       // the span on the target as stored
                const uint32_t w0 = fLo >> 4, nW = ((fLo + iSpan) >> 4) - w0 + 1u;
                if (nW <= (uint32_t) STAGE_WORDS) {
                    for (uint32_t j = 0; j < nW; j++) stage[lane * STAGE_WORDS + j] = a.codes[tw + w0 + j];
                    iW0 = w0 | 0x80000000u;
                }
            }
        }
        const uint64_t okMask = __ballot(ok);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the next query's records, on their way during the pile-up
        qCur = qN; qmCur = qmN; r0Cur = r0N; nRecCur = (uint32_t) (r1N - r0N); ryCur = 0xFFFFu;
        if (hasNext && (uint32_t) lane < nRecCur) { recCur = a.rec[r0N + lane]; if (a.ry) ryCur = a.ry[r0N + lane]; }

        // ---- pile-up + call, 64 positions at a time.  The record fields are wave-uniform: they are read out of the owning lane into
        // scalar registers, the per-lane work is the target letter, its damage class and one LDS add.
        const uint32_t lastWord = (qLen + 15) / 16;
        for (uint32_t base = 0; base < qLen; base += 64) {
            const uint32_t p = base + lane;
            uint64_t touched = 0;       // slots of this lane's column that are non-zero
            uint32_t cov4 = 0;          // records per target base (8 bits each: at most 64 records), counted on the way
            for (uint64_t m = okMask; m; m &= m - 1) {
                const int r = __ffsll((unsigned long long) m) - 1;
                const uint32_t qs = (uint32_t) __builtin_amdgcn_readlane((int) iQs, r), span = (uint32_t) __builtin_amdgcn_readlane((int) iSpan, r);
                if (qs + span < base || qs >= base + 64) continue;   // wave uniform
                const uint32_t lf = (uint32_t) __builtin_amdgcn_readlane((int) iLenFlags, r), tw = (uint32_t) __builtin_amdgcn_readlane((int) iTw, r);
                const uint32_t ds = (uint32_t) __builtin_amdgcn_readlane((int) iDs, r), tLen = lf & 0x3FFFFFFFu;
                const uint32_t w0f = (uint32_t) __builtin_amdgcn_readlane((int) iW0, r);
                const uint32_t d = p - qs;
                if (d <= span) {
                    const uint32_t tpos = ds + d;                            // position on the oriented target
                    uint32_t tb;
                    if (w0f & 0x80000000u) {
                        const bool rev = (lf & 0x80000000u) != 0;
                        const uint32_t f = rev ? tLen - 1u - tpos : tpos;
                        tb = (stage[(uint32_t) r * STAGE_WORDS + (f >> 4) - (w0f & 0x7FFFFFFFu)] >> ((f & 15u) * 2u)) & 3u;
                        if (rev) tb = 3u - tb;
                    } else if (lf & 0x40000000u) tb = targetBase(a, tw, tLen, true, (lf & 0x80000000u) != 0, tpos);
                    else if (lf & 0x80000000u) tb = 3u - cdm_base(a.codes, tw, tLen - 1u - tpos);
                    else tb = cdm_base(a.codes, tw, tpos);
                    // damage class 0..4 from the 5' end, 6..10 at the 3' end, 5 inside (accepted records are at least 30 columns long)
                    const uint32_t cls = min(tpos, 5u) + (uint32_t) max((int) tpos - (int) (tLen - 6u), 0);
                    const uint32_t slot = tb * 11u + cls;
                    atomicAdd(&cntWords[slot * (64 / PER) + laneWord], (lf & 0x80000000u) ? incRev : incFwd);
                    touched |= 1ull << slot;
                    cov4 += 1u << (8u * tb);
                }
            }
            uint32_t newCode = 0; bool keep = true;
            if (p < qLen) {
                uint32_t qb = cdm_base(a.codes, qw, p);
                const bool qIsN = qHasN && cdm_isN(a.nmask, qw, p);
                if (qIsN) qb = 0;
                newCode = callBase(sLogT, sLogQ, sLogD, qb, p, qLen, qWasExt,
                                   [&](int slot) { const uint32_t v = cntWords[slot * (64 / PER) + laneWord] >> laneShift; return (uint64_t) (v & ((1u << HB) - 1u)) | ((uint64_t) ((v >> HB) & ((1u << HB) - 1u)) << 32); },
                                   touched, keep, cov4);
            }
            for (uint64_t m = touched; m; m &= m - 1) atomicAnd(&cntWords[(__ffsll((unsigned long long) m) - 1) * (64 / PER) + laneWord], laneClear);
            const uint64_t b0 = cdm_ballot((newCode & 1u) != 0), b1 = cdm_ballot((newCode & 2u) != 0);
            uint64_t nb = 0;
            if (qHasN) nb = cdm_ballot(p < qLen && keep && cdm_isN(a.nmask, qw, p));
            if (lane < 4) {
                const uint32_t w = (base >> 4) + lane;
                if (w < lastWord) {
                    const uint32_t lo = (uint32_t) (b0 >> (16 * lane)), hi = (uint32_t) (b1 >> (16 * lane));
                    a.outCodes[qw + w] = cdm_spread16(lo) | (cdm_spread16(hi) << 1);
                    if (qHasN) reinterpret_cast<uint16_t *>(a.outNmask)[qw + w] = (uint16_t) (nb >> (16 * lane));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// queries with more than one alignment record: up to smallMax records -> small list, up to 64 -> fast list, more -> general list
__global__ void k_mark_active(const uint64_t *__restrict__ aoff, uint32_t n, uint32_t smallMax, uint32_t *__restrict__ active, uint32_t *__restrict__ activeFast,
                              uint32_t *__restrict__ activeSmall, unsigned int *__restrict__ counters) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t c = (q < n) ? aoff[q + 1] - aoff[q] : 0;
    const uint32_t s0 = cdm_wave_append(&counters[0], c > 64), s2 = cdm_block_append(&counters[2], c > smallMax && c <= 64), s3 = cdm_block_append(&counters[3], c > 1 && c <= smallMax);
    if (c > 64) active[s0] = q; else if (c > smallMax) activeFast[s2] = q; else if (c > 1) activeSmall[s3] = q;
}


// the same for a DB with letters beyond ACGTN: a query that carries such letters, or one of whose targets does, goes to the general kernel
__global__ void k_mark_active_raw(const uint64_t *__restrict__ aoff, const AlnRec *__restrict__ rec, const uint8_t *__restrict__ hasN, uint32_t n, uint32_t smallMax,
                                  uint32_t *__restrict__ active, uint32_t *__restrict__ activeFast, uint32_t *__restrict__ activeSmall, unsigned int *__restrict__ counters) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t c = (q < n) ? aoff[q + 1] - aoff[q] : 0;
    if (c > 1 && c <= 64) {
        bool odd = (hasN[q] & 2u) != 0;
        for (uint64_t r = aoff[q]; r < aoff[q + 1] && !odd; r++) odd = (hasN[rec[r].target] & 2u) != 0;
        if (odd) c = 65;
    }
    const uint32_t s0 = cdm_wave_append(&counters[0], c > 64), s2 = cdm_block_append(&counters[2], c > smallMax && c <= 64), s3 = cdm_block_append(&counters[3], c > 1 && c <= smallMax);
    if (c > 64) active[s0] = q; else if (c > smallMax) activeFast[s2] = q; else if (c > 1) activeSmall[s3] = q;
}


// test hook: one thread per count vector {qBase, qIter, qLen, wasCorr, 44 x (total | reverse << 16)}
__global__ void k_debug_call(const DamageLut *lut, const uint32_t *vec, uint32_t n, uint8_t *out) {
    __shared__ double sLogT[16], sLogQ[12 * 16], sLogD[2 * 11 * 16];
    for (int i = threadIdx.x; i < 16; i += blockDim.x) sLogT[i] = (&lut->logT[0][0])[i];
    for (int i = threadIdx.x; i < 12 * 16; i += blockDim.x) sLogQ[i] = (&lut->logQ[0][0][0])[i];
    for (int i = threadIdx.x; i < 2 * 11 * 16; i += blockDim.x) sLogD[i] = (&lut->logD[0][0][0][0])[i];
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *v = vec + (size_t) i * 48;
    bool keep;
    out[i] = (uint8_t) callBase(sLogT, sLogQ, sLogD, v[0], v[1], v[2], v[3] != 0, [&](int slot) { const uint32_t x = v[4 + slot]; return (uint64_t) (x & 0xFFFFu) | ((uint64_t) (x >> 16) << 32); }, ALL_SLOTS, keep);
}

}  // namespace

// not part of the public header: used by tests/test_gpu_correct.py through ctypes to run the reference's
// mostLikeliBaseRead known answers (tests/golden/functions/mostlikeli.tsv.gz) through the device call path
extern "C" int cdm_debug_call_bases(cdm_ctx *ctx, const uint32_t *vectors, uint32_t n, uint8_t *out) {
    if (!ctx->haveDamage) { cdm_set_error("cdm_debug_call_bases: no damage model"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    uint32_t *dv = nullptr; uint8_t *dout = nullptr;
    CDM_HIP(cdmMalloc(&dv, (size_t) n * 48 * 4));
    CDM_HIP(cdmMalloc(&dout, n));
    hipMemcpyAsync(dv, vectors, (size_t) n * 48 * 4, hipMemcpyHostToDevice, ctx->stream);
    hipLaunchKernelGGL(k_debug_call, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, ctx->lutDev, dv, n, dout);
    hipMemcpyAsync(out, dout, n, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    cdmFree(dv); cdmFree(dout);
    if (e != hipSuccess) { cdm_set_error("debug kernel failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; }
    return CDM_OK;
}

int cdm_correct_impl(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_alns *alns, const cdm_ancient_params *par, cdm_seqdb *out) {
    hipStream_t s = ctx->stream;
    const uint32_t n = (uint32_t) db->n;
    DevBuf<uint32_t> active, activeFast, activeSmall; DevBuf<unsigned int> counters; DevBuf<uint8_t> accept;
    if (!active.alloc(n) || !activeFast.alloc(n) || !activeSmall.alloc(n) || !counters.alloc(4) || !accept.alloc(alns->count)) { cdm_set_error("out of device memory in cdm_correct"); return CDM_ERR_HIP; }
    // coverage <= 1 everywhere unless the kernels overwrite: start from a copy of the input bases
    CDM_HIP(hipMemcpyAsync(out->codes, db->codes, db->words * 4, hipMemcpyDeviceToDevice, s));
    CDM_HIP(hipMemcpyAsync(out->nmask, db->nmask, ((db->words * 16 + 31) / 32) * 4, hipMemcpyDeviceToDevice, s));
    CDM_HIP(hipMemsetAsync(counters.p, 0, 16, s));   // [0] queries for the general kernel, [1] error flag, [2] queries for the fast kernel
    // CDM_CORRECT_VARIANT (experiments): "0" = one fast instance for up to 64 records; "s<W>" = small instance (<= 15 records) with W waves per SIMD
    const char *varEnv = cdmGetenv("CDM_CORRECT_VARIANT");
    const int smallW = (varEnv && varEnv[0] == 's') ? atoi(varEnv + 1) : (varEnv && varEnv[0] == '0' ? 0 : 6);
    if (db->raw) {
        CDM_HIP(hipMemcpyAsync(out->raw, db->raw, db->words * 16, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_mark_active_raw, dim3((n + 1023) / 1024), dim3(1024), 0, s, alns->off, alns->rec, db->hasN, n, smallW ? 15u : 0u, active.p, activeFast.p, activeSmall.p, counters.p);
    } else
    hipLaunchKernelGGL(k_mark_active, dim3((n + 1023) / 1024), dim3(1024), 0, s, alns->off, n, smallW ? 15u : 0u, active.p, activeFast.p, activeSmall.p, counters.p);
    DevBuf<SeqMeta> meta;
    MetaUniform uni;
    if (int rc = cdm_build_meta(ctx, db, &meta.p, &uni)) return rc;
    CorrectArgs a;
    cdmSetMeta(a, meta.p, uni); a.ext.m = meta.p; a.ext.plain = uni.words; a.codes = db->codes; a.nmask = db->nmask; a.raw = db->raw; a.outRaw = out->raw;
    a.aoff = alns->off; a.rec = alns->rec; a.ry = (alns->ryMism && alns->rySerial == db->serial) ? alns->ryMism : nullptr; a.active = active.p; a.nActive = counters.p; a.accept = accept.p; a.errFlag = counters.p + 1;
    a.outCodes = out->codes; a.outNmask = out->nmask; a.lut = ctx->lutDev; a.seqIdThr = par->seq_id_thr; a.corrRy = par->corr_reads_ry_seq_id;
    const int blocks = ctx->cuCount * 8;
    hipEventRecord(ctx->ev0, s);
    const char *padEnv = cdmGetenv("CDM_LDS_PAD");          // experiments: dynamic LDS that lowers the occupancy
    const unsigned pad = padEnv ? (unsigned) atoi(padEnv) : 0u;
    if (smallW == 8) hipLaunchKernelGGL((k_correct_fast<15, uint8_t, 8>), dim3(blocks * 2), dim3(64 * FAST_WAVES), pad, s, a, activeSmall.p, counters.p + 3);
    else if (smallW == 5) hipLaunchKernelGGL((k_correct_fast<15, uint8_t, 5>), dim3(blocks * 2), dim3(64 * FAST_WAVES), pad, s, a, activeSmall.p, counters.p + 3);
    else if (smallW) hipLaunchKernelGGL((k_correct_fast<15, uint8_t, 6>), dim3(blocks * 2), dim3(64 * FAST_WAVES), pad, s, a, activeSmall.p, counters.p + 3);
    const char *bigEnv = cdmGetenv("CDM_CORRECT_BIGW");     // experiments: waves per SIMD of the 16..64-record instance
    const int bigW = bigEnv ? atoi(bigEnv) : 5;
    if (bigW == 6) hipLaunchKernelGGL((k_correct_fast<64, uint16_t, 6>), dim3(blocks * 2), dim3(64 * FAST_WAVES), pad, s, a, activeFast.p, counters.p + 2);
    else if (bigW == 5) hipLaunchKernelGGL((k_correct_fast<64, uint16_t, 5>), dim3(blocks * 2), dim3(64 * FAST_WAVES), pad, s, a, activeFast.p, counters.p + 2);
    else hipLaunchKernelGGL((k_correct_fast<64, uint16_t, 4>), dim3(blocks), dim3(64 * FAST_WAVES), pad, s, a, activeFast.p, counters.p + 2);
    hipLaunchKernelGGL(k_correct, dim3(blocks / 2), dim3(64 * WAVES_PER_BLOCK), 0, s, a);
    hipEventRecord(ctx->ev1, s);
    CDM_LAUNCH_CHECK();
    unsigned int flags[2] = {0, 0};
    CDM_HIP(hipMemcpyAsync(flags, counters.p, 8, hipMemcpyDeviceToHost, s));
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_set_error("ancient_correction kernel failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    hipEventElapsedTime(&ctx->lastMs[0], ctx->ev0, ctx->ev1);
    return CDM_OK;
}
