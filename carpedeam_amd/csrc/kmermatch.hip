// kmermatcher (linclust-style k-mer matching) on the device, single-split semantics.
//
// Replaces lib/mmseqs/src/linclust/kmermatcher.cpp doComputation (:391-451) and the result writer (:815-930, :717-729):
//   K1 k_seq_hash, k_extract_pair, k_extract_fast, k_extract   fillKmerPositionArray :77-388  per sequence: canonical k-mers, XXH64 16-bit
//                     min-hash, per-sequence ordering by (hash, k-mer, pos) for the repeated-k-mer skipping and the bottom-m
//                     selection, + the whole-sequence hash tuple
//   K2 sort 1         :412   stable sort on the k-mer: the top 27 sort bits by three onesweep radix passes (radix.h), the low bits per
//                     bucket on chip
//   K3 k_bucket_groups (k_groups)   assignGroup :453-562  first sequence of every k-mer run by (length desc, id, pos) is the
//                     representative; members become (rep, id, diagonal, strand); singletons dropped.  Fused with the
//                     on-chip part of sort 1 (bucket.h)
//   K2 sort 2         :431   stable sort on (rep, id, diagonal) packed into one 64-bit key: the k-mer RUNS are sorted by
//                     representative (run records, runsort.h), every representative's tuples then on chip (k_unit_sort)
//   K4 k_seg_count/place   writeKmerMatcherResult :815-930  per (rep, target): shared k-mer count, most frequent diagonal
//                     (last maximum wins), strand of that diagonal's last tuple; every sequence gets a record that starts
//                     with its self hit (fill-in :717-729)
// Quirks kept on purpose (they are observable in the prefilter DB): the repeated-k-mer skip that processes the element
// after a run unconditionally (:277-350); repIsReverse starting as false for the very first k-mer group (:453-467); the
// per-target scan in the writer running on into the next representative's tuples when they have the same target id
// (:875-887).
#include <cmath>
#include <memory>
#include <cstring>

#include "common.h"
#include "devutil.h"
#include "bucket.h"
#include "scan.h"
#include "runsort.h"
#include "aggvote.h"
#include "radix.h"

namespace {

constexpr uint64_t BIT63 = 1ull << 63;
// the two buffers a sort alternates between: where the data is now, and the other one
template <typename T> struct DoubleBuf {
    T *cur = nullptr, *alt = nullptr;
    DoubleBuf() = default;
    DoubleBuf(T *c, T *a) : cur(c), alt(a) {}
    T *current() const { return cur; }
    T *alternate() const { return alt; }
};

// xxHash64 of one 8-byte word (lib/mmseqs/lib/xxhash/xxhash.h XXH64, len = 8; kmermatcher.cpp:33-38)
__host__ __device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__host__ __device__ __forceinline__ uint64_t xxh64_u64(uint64_t in, uint64_t seed) {
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL, P4 = 9650029242287828579ULL,
                   P5 = 2870177450012600261ULL;
    uint64_t h = seed + P5 + 8;
    uint64_t k1 = in * P2; k1 = rotl64(k1, 31); k1 *= P1;
    h ^= k1; h = rotl64(h, 27) * P1 + P4;
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}
// Util::revComplement (M/commons/Util.cpp:601-638) in MMseqs2's A,C,T,G = 0..3 coding: complement = xor 2
__device__ __forceinline__ uint64_t revComplement(uint64_t kmer, int k) {
    uint64_t x = kmer ^ 0xAAAAAAAAAAAAAAAAULL;
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = __builtin_bswap64(x);
    return x >> (64 - 2 * k);
}

// ------------------------------------------------------------------------------------------------ tuple layouts
// Two physical layouts of the (k-mer, strand, sequence id, sequence length, position) tuple the reference keeps in
// KmerPosition<T> (kmermatcher.h:49-54).  Region 1 of the array (one slot per k-mer position + slot 0 per sequence) is sorted
// on the low 2k key bits, region 2 (one whole-sequence hash tuple per sequence, 63 random bits) on 63 bits.
struct TupleGeom {
    int kbits, lb;              // 2k, bits of a length/position field
    uint64_t kmerSlots;         // size of region 1
    const uint32_t *lenArr;     // sequence lengths (region-2 tuples of the packed layout look their length up)
    // LayoutSlot (every sequence uniL letters): uniS slots per sequence; after sort 1 the region-1 tuples are SLOT TUPLES (radix.h) in
    // seg[0 .. BINS] segments by head digit (the k-mer bits from headShift on)
    uint32_t uniS = 0, uniL = 0; int uniK = 0; uint32_t uniMul = 0; int uniSh = 0;       // (uniMul, uniSh: division by uniS, slotSplit)
    const unsigned long long *seg = nullptr; int headShift = 0;
};
// division of a 32-bit number by an invariant d >= 1 (Granlund & Montgomery): q = (t + ((n - t) >> 1)) >> sh with t = mulhi(n, mul)
inline void divMagic(uint32_t d, uint32_t &mul, int &sh) {
    int l = 0; while ((1ull << l) < d) l++;
    mul = (uint32_t) ((((1ull << l) - d) << 32) / d + 1ull); sh = l > 0 ? l - 1 : 0;
    if (d == 1) { mul = 0; sh = 0; }       // t = 0: q = n >> 1 >> 0 would be wrong - d = 1 is special-cased in slotSplit
}
// head digit of the slot tuple at k-mer-order index idx: the last segment that starts at or in front of it
__device__ __forceinline__ uint32_t headDigit(const TupleGeom &g, uint64_t idx) {
    uint32_t d = 0;
#pragma unroll
    for (uint32_t st = rx::BINS / 2; st > 0; st >>= 1) if (g.seg[d + st] <= idx) d += st;
    return d;
}
// the same for a wave-uniform index: the nine look-ups go through the scalar cache (as vector loads they are nine L2 round trips in a
// row at the start of every wave of the grouping kernel: 67 instead of 49 ms at 50 M reads)
__device__ __forceinline__ uint32_t headDigitUniform(const TupleGeom &g, uint64_t idx) {
    const uint64_t u = ((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (idx >> 32)) << 32) | (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) idx);
    uint32_t d = 0;
#pragma unroll
    for (uint32_t st = rx::BINS / 2; st > 0; st >>= 1) { const unsigned long long b = g.seg[d + st]; if (b <= u) d += st; }
    return (uint32_t) __builtin_amdgcn_readfirstlane((int) d);
}
// slot index -> (sequence, slot of the sequence): slots are laid out sequence by sequence, uniS each (0: the whole-sequence hash tuple's
// slot, 1 + p: k-mer position p).  The quotient by a double product, corrected (exact for any 32-bit slot).
__device__ __forceinline__ void slotSplit(const TupleGeom &g, uint32_t slot, uint32_t &seq, uint32_t &r) {
    const uint32_t t = __umulhi(slot, g.uniMul);
    const uint32_t q = g.uniS == 1u ? slot : (t + ((slot - t) >> 1)) >> g.uniSh;
    seq = q; r = slot - q * g.uniS;
}
// 16 bytes: u64 key = k-mer | strand << 63, u64 value = id << 2 FB | len << FB | pos.  FB = 16: any DB with sequences below
// 65 536 letters (ids up to 2^32); FB = 20: sequences up to 2^20 letters (the reference's `int` position path,
// kmermatcher.cpp:803-808: contigs), ids up to 2^24.
template <int FB>
struct LayoutWideT {
    typedef uint64_t V;
    static constexpr bool bySlot = false;
    static constexpr uint64_t FM = (1ull << FB) - 1ull;
    __device__ static void store(uint64_t *keys, V *vals, uint64_t slot, uint64_t kmer63, bool fwd, uint32_t seq, uint32_t L, uint32_t pos, const TupleGeom &) {
        keys[slot] = kmer63 | (fwd ? BIT63 : 0ull); vals[slot] = ((uint64_t) seq << (2 * FB)) | ((uint64_t) L << FB) | pos;
    }
    __device__ static void storeHash(uint64_t *keys, V *vals, uint64_t slot, uint64_t hash64, uint32_t seq, uint32_t L, const TupleGeom &) {
        keys[slot] = hash64; vals[slot] = ((uint64_t) seq << (2 * FB)) | ((uint64_t) L << FB);
    }
    __device__ static void storeEmpty(uint64_t *keys, V *vals, uint64_t slot) { keys[slot] = ~0ull; vals[slot] = 0; }
    __device__ static uint64_t kmerOf(uint64_t key, uint64_t, const TupleGeom &) { return key & ~BIT63; }
    __device__ static uint32_t seqOf(V v) { return (uint32_t) (v >> (2 * FB)); }
    __device__ static uint32_t lenOf(uint64_t, V v, uint64_t, const TupleGeom &) { return (uint32_t) ((v >> FB) & FM); }
    __device__ static uint32_t posOf(uint64_t, V v, uint64_t, const TupleGeom &) { return (uint32_t) (v & FM); }
    // length and position of a region-1 tuple
    __device__ static void unpackR1(uint64_t, V v, const TupleGeom &, uint32_t &len, uint32_t &pos) { pos = (uint32_t) (v & FM); len = (uint32_t) ((v >> FB) & FM); }
};
typedef LayoutWideT<16> LayoutWide;
typedef LayoutWideT<20> LayoutLong;
// 16 bytes for any DB: u64 key = k-mer | strand << 63, u64 value = id << 32 | pos; the sequence's length is looked up (a DB of 2^24
// sequences or more with one of them beyond 65 534 letters: the contig iterations of a 25 M-read run, BASELINE config 5)
struct LayoutHuge {
    typedef uint64_t V;
    static constexpr bool bySlot = false;
    __device__ static void store(uint64_t *keys, V *vals, uint64_t slot, uint64_t kmer63, bool fwd, uint32_t seq, uint32_t, uint32_t pos, const TupleGeom &) {
        keys[slot] = kmer63 | (fwd ? BIT63 : 0ull); vals[slot] = ((uint64_t) seq << 32) | pos;
    }
    __device__ static void storeHash(uint64_t *keys, V *vals, uint64_t slot, uint64_t hash64, uint32_t seq, uint32_t, const TupleGeom &) { keys[slot] = hash64; vals[slot] = (uint64_t) seq << 32; }
    __device__ static void storeEmpty(uint64_t *keys, V *vals, uint64_t slot) { keys[slot] = ~0ull; vals[slot] = 0; }
    __device__ static uint64_t kmerOf(uint64_t key, uint64_t, const TupleGeom &) { return key & ~BIT63; }
    __device__ static uint32_t seqOf(V v) { return (uint32_t) (v >> 32); }
    __device__ static uint32_t lenOf(uint64_t, V v, uint64_t, const TupleGeom &g) { return g.lenArr[(uint32_t) (v >> 32)]; }
    __device__ static uint32_t posOf(uint64_t, V v, uint64_t, const TupleGeom &) { return (uint32_t) v; }
    __device__ static void unpackR1(uint64_t, V v, const TupleGeom &g, uint32_t &len, uint32_t &pos) { pos = (uint32_t) v; len = g.lenArr[(uint32_t) (v >> 32)]; }
};
// 12 bytes: u64 key = k-mer | pos << (2k + 1) | len << (2k + 1 + lb) | strand << 63, u32 value = id.  Needs 2k + 1 + 2 lb <= 63
// (k = 20: sequences up to 2047 letters); a quarter less traffic in every radix pass.  Bit 2k stays clear in every real tuple
// of region 1 (in both layouts): it is set only in the unused-slot key ~0, so sorting region 1 on bits up to and including
// 2k moves the unused slots behind all real tuples.
struct LayoutPacked {
    typedef uint32_t V;
    static constexpr bool bySlot = false;
    __device__ static void store(uint64_t *keys, V *vals, uint64_t slot, uint64_t kmer63, bool fwd, uint32_t seq, uint32_t L, uint32_t pos, const TupleGeom &g) {
        keys[slot] = kmer63 | ((uint64_t) pos << (g.kbits + 1)) | ((uint64_t) L << (g.kbits + 1 + g.lb)) | (fwd ? BIT63 : 0ull); vals[slot] = seq;
    }
    __device__ static void storeHash(uint64_t *keys, V *vals, uint64_t slot, uint64_t hash64, uint32_t seq, uint32_t L, const TupleGeom &g) {
        // region 1 (a hash that fits 2k bits): position 0, length packed above it; region 2: the full hash, length looked up
        keys[slot] = (slot < g.kmerSlots) ? (hash64 | ((uint64_t) L << (g.kbits + 1 + g.lb))) : hash64; vals[slot] = seq;
    }
    __device__ static void storeEmpty(uint64_t *keys, V *vals, uint64_t slot) { keys[slot] = ~0ull; vals[slot] = 0; }
    __device__ static uint64_t kmerOf(uint64_t key, uint64_t slot, const TupleGeom &g) { return slot < g.kmerSlots ? (key & ((1ull << g.kbits) - 1ull)) : (key & ~BIT63); }
    __device__ static uint32_t seqOf(V v) { return v; }
    __device__ static uint32_t lenOf(uint64_t key, V v, uint64_t slot, const TupleGeom &g) { return slot < g.kmerSlots ? (uint32_t) ((key >> (g.kbits + 1 + g.lb)) & ((1ull << g.lb) - 1ull)) : g.lenArr[v]; }
    __device__ static uint32_t posOf(uint64_t key, V, uint64_t slot, const TupleGeom &g) { return slot < g.kmerSlots ? (uint32_t) ((key >> (g.kbits + 1)) & ((1ull << g.lb) - 1ull)) : 0u; }
    __device__ static void unpackR1(uint64_t key, V, const TupleGeom &g, uint32_t &len, uint32_t &pos) {
        const uint32_t t = (uint32_t) (key >> (g.kbits + 1)), m = (1u << g.lb) - 1u;      // 2 lb + 1 <= 63 - 2k - 1 bits are left
        pos = t & m; len = (t >> g.lb) & m;
    }
};
// 8 bytes per tuple through all of sort 1, for DBs whose sequences all have ONE length (uniL letters, uniS = uniL - k + 2 slots each,
// n x uniS < 2^32; k <= 20): the extractor writes only u64 key = k-mer | strand << 63 at the tuple's slot (~0 = empty), and WHICH
// sequence and position a tuple belongs to is the slot's index - slot = seq x uniS + 1 + position in the forward sequence (slot 0 of a
// sequence: its whole-sequence hash tuple, if that fits 2k bits).  The head pass of the sort (radix.h sortSlotKeys) drops the empty
// slots, makes the index explicit and the head digit implicit: the sorted region 1 holds SLOT TUPLES [k-mer bits below headShift | strand
// | slot index].  Wherever a tuple is looked at - the grouping kernel's window, big buckets, the left-over scan - it is first turned into
// the (key, id) pair of LayoutPacked (slotTupleToPair), so everything behind sort 1 is that layout's code.  Region 2 (whole-sequence
// hashes) keeps (key, id) pairs; its values live in an array of their own that `vals` points kmerSlots entries in front of.
struct LayoutSlot {
    typedef uint32_t V;
    static constexpr bool bySlot = true;
    __device__ static void store(uint64_t *keys, V *, uint64_t slot, uint64_t kmer63, bool fwd, uint32_t, uint32_t, uint32_t, const TupleGeom &) { keys[slot] = kmer63 | (fwd ? BIT63 : 0ull); }
    __device__ static void storeHash(uint64_t *keys, V *vals, uint64_t slot, uint64_t hash64, uint32_t seq, uint32_t, const TupleGeom &g) { keys[slot] = hash64; if (slot >= g.kmerSlots) vals[slot] = seq; }
    __device__ static void storeEmpty(uint64_t *keys, V *, uint64_t slot) { keys[slot] = ~0ull; }
    __device__ static uint64_t kmerOf(uint64_t key, uint64_t slot, const TupleGeom &g) { return LayoutPacked::kmerOf(key, slot, g); }
    __device__ static uint32_t seqOf(V v) { return v; }
    __device__ static uint32_t lenOf(uint64_t key, V v, uint64_t slot, const TupleGeom &g) { return LayoutPacked::lenOf(key, v, slot, g); }
    __device__ static uint32_t posOf(uint64_t key, V v, uint64_t slot, const TupleGeom &g) { return LayoutPacked::posOf(key, v, slot, g); }
    __device__ static void unpackR1(uint64_t key, V v, const TupleGeom &g, uint32_t &len, uint32_t &pos) { LayoutPacked::unpackR1(key, v, g, len, pos); }
};
// sequence, stored position (kmermatcher.cpp:186: counted from the other end on the reverse strand) and strand of a slot tuple
__device__ __forceinline__ void slotFields(const TupleGeom &g, uint64_t t8, uint32_t &id, uint32_t &pos, bool &fwd) {
    fwd = ((uint32_t) (t8 >> rx::SLOT_STRAND_SHIFT) & 1u) != 0u;
    uint32_t r;
    slotSplit(g, (uint32_t) t8, id, r);
    pos = r == 0u ? 0u : (fwd ? r - 1u : g.uniL - (r - 1u) - (uint32_t) g.uniK);
}
// the slot tuple at k-mer-order index idx (head digit td) as the (key, id) pair LayoutPacked holds
__device__ __forceinline__ void slotTupleToPair(const TupleGeom &g, uint64_t t8, uint32_t td, uint64_t &key, uint32_t &id) {
    static_assert(rx::SLOT_STRAND_SHIFT == 32 && rx::SLOT_KEY_SHIFT == 33, "the tuple's high word is k-mer bits << 1 | strand");
    const uint32_t hiw = (uint32_t) (t8 >> 32);
    const bool fwd = (hiw & 1u) != 0u;
    uint32_t r;
    slotSplit(g, (uint32_t) t8, id, r);
    const uint32_t pos = r == 0u ? 0u : (fwd ? r - 1u : g.uniL - (r - 1u) - (uint32_t) g.uniK);     // (the reverse strand's position, kmermatcher.cpp:186)
    if (g.kbits + 1 >= 32) {
        // the key word by word (64-bit shifts are slow vector instructions, and this runs once per tuple): k-mer = td << headShift | low bits
        const uint32_t lo = (hiw >> 1) | (td << g.headShift);
        const uint32_t hi = (g.headShift ? td >> (32 - g.headShift) : 0u) | (pos << (g.kbits + 1 - 32)) | (g.uniL << (g.kbits + 1 + g.lb - 32)) | (fwd ? 0x80000000u : 0u);
        key = ((uint64_t) hi << 32) | lo;
    } else {
        const uint64_t kmer = ((uint64_t) td << g.headShift) | (uint64_t) (hiw >> 1);
        key = kmer | ((uint64_t) pos << (g.kbits + 1)) | ((uint64_t) g.uniL << (g.kbits + 1 + g.lb)) | (fwd ? BIT63 : 0ull);
    }
}
// ... and back (a tuple that came out of slot 0 - a whole-sequence hash that fits 2k bits, position 0 - gets the slot of position 0 on its
// strand: the index is only ever read through slotTupleToPair, which gives the same pair again)
__device__ __forceinline__ uint64_t pairToSlotTuple(const TupleGeom &g, uint64_t key, uint32_t id) {
    uint32_t len, pos;
    LayoutPacked::unpackR1(key, id, g, len, pos);
    const bool fwd = (key & BIT63) != 0ull;
    const uint32_t slot = id * g.uniS + 1u + (fwd ? pos : g.uniL - pos - (uint32_t) g.uniK);
    return ((key & ((1ull << g.headShift) - 1ull)) << rx::SLOT_KEY_SHIFT) | ((fwd ? 1ull : 0ull) << rx::SLOT_STRAND_SHIFT) | (uint64_t) slot;
}
// what the kernels that look at region 1 IN MEMORY (behind sort 1) go through: the sort bits of the tuple at idx (k-mer + unused-slot bit),
// and the tuple as a (key, value) pair
template <typename LY> __device__ __forceinline__ uint64_t memSortBits(const uint64_t *keys, uint64_t idx, const TupleGeom &g) {
    if constexpr (LY::bySlot) { if (idx < g.kmerSlots) return ((uint64_t) headDigit(g, idx) << g.headShift) | (keys[idx] >> rx::SLOT_KEY_SHIFT); }
    return keys[idx] & ((2ull << g.kbits) - 1ull);
}
template <typename LY> __device__ __forceinline__ void memPair(const uint64_t *keys, const typename LY::V *vals, uint64_t idx, const TupleGeom &g, uint64_t &key, typename LY::V &v) {
    if constexpr (LY::bySlot) { if (idx < g.kmerSlots) { uint32_t id; slotTupleToPair(g, keys[idx], headDigit(g, idx), key, id); v = id; return; } }
    key = keys[idx]; v = vals[idx];
}

struct SeqPos;
template <typename LY> struct ExtractArgs {
    const uint32_t *woff, *len, *codes, *nmask;
    const uint8_t *hasN;
    const uint32_t *list;       // sequence indices this launch handles
    uint32_t nList;
    int k, kmersPerSeq; float scale; uint64_t seed; int ignoreMultiKmer;
    uint64_t *keys; typename LY::V *vals;   // tuple array
    const uint64_t *slotOff;    // [n+1] first slot of every sequence: 1 whole-sequence tuple + one slot per k-mer position;
                                // unused slots hold the key ~0 (sorts last, dropped by k_groups)
    uint32_t *slowShort, *slowLong; unsigned int *slowCnt;   // sequences the fast kernel hands to the general one
    uint32_t *single;           // sequences k_extract_pair hands to k_extract_fast (count in slowCnt[2])
    uint32_t *slowHuge;         // sequences with 4096 k-mer positions or more (count in slowCnt[3]): k_extract with global scratch
    SeqPos *hugeSp; uint8_t *hugeSel; uint32_t hugeCap;   // that scratch: hugeCap records per block
    const unsigned int *listCount;   // device-side length of `list` for k_extract_fast (NULL: all sequences)
    uint32_t n;
    // The whole-sequence hash tuple (63 random bits) lives in a second region behind the k-mer slots, [hashBase, hashBase+n),
    // at the sequence's rank in (length desc, id asc) order, unless its key happens to fit the 2k bits of a k-mer (then it
    // stays in slot 0 of the sequence).  Region 1 is sorted on 2k bits, region 2 on 63: every key of region 1 is smaller
    // than every key of region 2, so the concatenation is the array the reference sorts on the full key.
    uint64_t hashBase; const uint32_t *rankOf;
    TupleGeom geom;
    // Multi-GPU runs split the k-mer space into ranges (the reference's MPI split, kmermatcher.cpp:634-663, by value instead of
    // by hash so that the ranges are in k-mer order): only tuples with kLo <= key < kHi are stored, the others leave their slot
    // empty; the whole-sequence hash tuples of region 2 belong to the last range.  belowFlag is set when a real tuple lies
    // below the range (the array's very first run is then not in it).  Single-GPU: kLo = 0, kHi = ~0, lastPart = 1.
    uint64_t kLo, kHi; int lastPart; unsigned int *belowFlag;
    // The other split of a multi-GPU run (round 4, cdm_kmermatch_split_*): a rank extracts the k-mers of ITS sequences only - the
    // sequences with order ranks [ordLo, ordHi) in the (length desc, id asc) slot order, all k-mer values - and the tuples then travel
    // to the owner of their k-mer range.  ordHi = 0: every sequence (one device, and the k-mer-range split above).
    uint32_t ordLo = 0, ordHi = 0;
    uint32_t uniS = 0;          // LayoutSlot: every sequence has uniS slots, sequence i the slots from i x uniS on, rank i (slotOff / rankOf are not built)
    // LayoutSlot: the extraction kernels count the head digits (k-mer bits from headShift on) of the tuples they leave in the slots -
    // the histogram of sort 1's head pass, which then needs no read of the keys of its own (NULL: not counted)
    unsigned long long *headHist = nullptr; int headShift = 0;
};
constexpr int HEAD_BINS = rx::BINS;
// a block's head digit counters: cleared at the start of an extraction kernel, added to the global ones at its end
__device__ __forceinline__ void headHistClear(unsigned int *sHead) { for (int i = threadIdx.x; i < HEAD_BINS; i += blockDim.x) sHead[i] = 0u; __syncthreads(); }
__device__ __forceinline__ void headHistFlush(const unsigned int *sHead, unsigned long long *hist) {
    __syncthreads();
    for (int i = threadIdx.x; i < HEAD_BINS; i += blockDim.x) { const unsigned int c = sHead[i]; if (c) atomicAdd(&hist[i], (unsigned long long) c); }
}
template <typename LY> __device__ __forceinline__ uint64_t slotBase(const ExtractArgs<LY> &a, uint32_t seq) { if constexpr (LY::bySlot) return (uint64_t) seq * a.uniS; else return a.slotOff[seq]; }
template <typename LY> __device__ __forceinline__ uint32_t seqRank(const ExtractArgs<LY> &a, uint32_t seq) { if constexpr (LY::bySlot) return seq; else return a.rankOf[seq]; }
template <typename LY> __device__ __forceinline__ bool ownedSeq(const ExtractArgs<LY> &a, uint32_t seq) {
    if (a.ordHi == 0) return true;
    const uint32_t r = a.rankOf[seq];
    return r >= a.ordLo && r < a.ordHi;
}
template <typename LY> __device__ __forceinline__ bool inRange(const ExtractArgs<LY> &a, uint64_t km) { return km >= a.kLo && km < a.kHi; }
template <typename LY> __device__ __forceinline__ void noteBelow(const ExtractArgs<LY> &a, bool below) {       // whole wave
    if (__ballot(below) != 0ull && (threadIdx.x & 63) == 0 && a.belowFlag[0] == 0u) a.belowFlag[0] = 1u;
}
template <typename LY>
__device__ __forceinline__ void putSeqHashTuple(const ExtractArgs<LY> &a, uint32_t seq, uint32_t L, uint64_t base, uint64_t h) {
    const uint64_t key = xxh64_u64(h, a.seed);
    const uint64_t hslot = a.hashBase + (seqRank(a, seq) - a.ordLo);
    const bool small = (key & ~BIT63) < (1ull << (2 * a.k));
    if (small) {
        if (inRange(a, key & ~BIT63)) LY::storeHash(a.keys, a.vals, base, key, seq, L, a.geom); else LY::storeEmpty(a.keys, a.vals, base);
        if constexpr (LY::bySlot) { if (a.headHist && inRange(a, key & ~BIT63)) atomicAdd(&a.headHist[(key & ~BIT63) >> a.headShift], 1ull); }      // (one sequence in 2^(63 - 2k))
        if ((key & ~BIT63) < a.kLo && a.belowFlag[0] == 0u) a.belowFlag[0] = 1u;
        LY::storeEmpty(a.keys, a.vals, hslot);
    } else {
        LY::storeEmpty(a.keys, a.vals, base);
        if (a.lastPart) LY::storeHash(a.keys, a.vals, hslot, key, seq, L, a.geom); else LY::storeEmpty(a.keys, a.vals, hslot);
    }
}

// 2k-bit window of the sequence starting at base pos, MMseqs2 coding (A,C,T,G), first base in the LOW bits
__device__ __forceinline__ uint64_t kmerWindow(const uint32_t *__restrict__ codes, uint32_t w0, uint32_t pos, uint32_t lastWord, int k) {
    const uint32_t w = pos >> 4, sh = (pos & 15u) * 2u;
    const uint64_t a = codes[w0 + w];
    const uint64_t b = (w + 1 <= lastWord) ? codes[w0 + w + 1] : 0u;
    const uint64_t c = (w + 2 <= lastWord) ? codes[w0 + w + 2] : 0u;
    uint64_t x = (a | (b << 32)) >> sh;
    if (sh) x |= c << (64 - sh);
    x ^= (x >> 1) & 0x5555555555555555ull;                     // A,C,G,T -> A,C,T,G
    return x;                                                   // 32 bases from pos on; callers mask what they need
}
// Indexer::computeKmerIdx order (first base most significant) from the window: reverse the 2-bit groups
__device__ __forceinline__ uint64_t groupsReversed(uint64_t x, int k) {
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = __builtin_bswap64(x);
    return x >> (64 - 2 * k);
}
// The whole-sequence tuple of every sequence (kmermatcher.cpp:244-267): Util::hash over the numeric sequence
// (M/commons/Util.h:338-346, h = h * 31 + c) then XXH64 (:135-138).  One thread per sequence: a serial Horner walk over packed
// words costs a wave a few instructions per sequence, where the wave-per-sequence extraction kernels spent hundreds on it.
template <typename LY>
__global__ __launch_bounds__(256) void k_seq_hash(ExtractArgs<LY> a) {
    const uint32_t seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= a.n || !ownedSeq(a, seq)) return;
    const uint32_t L = a.len[seq], w0 = a.woff[seq];
    uint64_t h = 0;
    if (a.hasN[seq]) {
        for (uint32_t i = 0; i < L; i++) {
            uint32_t c = cdm_base(a.codes, w0, i); c ^= c >> 1;
            if (cdm_isN(a.nmask, w0, i)) c = 4;
            h = h * 31 + c;
        }
    } else {
        for (uint32_t i = 0; i < L; i += 16) {
            uint32_t word = a.codes[w0 + (i >> 4)];
            const uint32_t nb = min(16u, L - i);
            for (uint32_t j = 0; j < nb; j++) { uint32_t c = word & 3u; c ^= c >> 1; h = h * 31 + c; word >>= 2; }
        }
    }
    putSeqHashTuple(a, seq, L, slotBase(a, seq), h);
}

constexpr int FAST_WAVES = 4, FAST_TABLE = 1024, FAST_CAP = 448;
// Fast path of K1: one wavefront per sequence, no per-sequence sort.  Valid when every k-mer is taken
// (positions <= kmersPerSeq - 1 + scale * L) and no canonical k-mer occurs twice in the sequence (checked with an LDS
// hash set); then the selection is "all k-mers" whatever the (hash, k-mer, pos) order.  Anything else goes to k_extract.
template <typename LY>
__global__ __launch_bounds__(64 * FAST_WAVES) void k_extract_fast(ExtractArgs<LY> a) {
    __shared__ unsigned long long sTable[FAST_WAVES][FAST_TABLE];
    __shared__ unsigned int sHead[LY::bySlot ? HEAD_BINS : 1];
    const bool countHead = LY::bySlot && a.headHist != nullptr;
    if (countHead) headHistClear(sHead);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long *table = sTable[wave];
    const int k = a.k;
    const uint32_t nItems = a.listCount ? *a.listCount : a.n;
    for (uint32_t item = blockIdx.x * FAST_WAVES + wave; item < nItems; item += gridDim.x * FAST_WAVES) {
        const uint32_t seq = a.listCount ? a.list[item] : item;
        if (!a.listCount && !ownedSeq(a, seq)) continue;          // (wave-uniform)
        const uint32_t L = a.len[seq], w0 = a.woff[seq];
        const bool hasN = a.hasN[seq] != 0;
        const uint32_t nPos = (L >= (uint32_t) k) ? (L - k + 1) : 0;
        const uint64_t base = slotBase(a, seq);
        const size_t cap = (size_t) (float) ((float) (a.kmersPerSeq - 1) + (a.scale * (float) L));
        if (nPos > cap || nPos > FAST_CAP) {   // wave uniform
            if (lane == 0) {
                if (nPos < 256) a.slowShort[atomicAdd(&a.slowCnt[0], 1u)] = seq;
                else if (nPos < 4096) a.slowLong[atomicAdd(&a.slowCnt[1], 1u)] = seq;
                else a.slowHuge[atomicAdd(&a.slowCnt[3], 1u)] = seq;
            }
            continue;
        }
        // hash set sized to the sequence (load factor <= 1/2): clearing it is a large share of this kernel's LDS traffic
        uint32_t tsize = 64; while (tsize < 2 * nPos) tsize <<= 1;
        const uint32_t tmask = tsize - 1;
        for (uint32_t i = lane; i < tsize; i += 64) table[i] = ~0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
        const uint32_t lastWord = (L + 15) / 16 - 1;
        bool dup = false, below = false;
        const uint64_t kmask = (1ull << (2 * k)) - 1ull;
        // canonical k-mer of the window w (idx = the same k-mer in Indexer order) at position pos: insert, store
        auto emit = [&](uint64_t w, uint64_t idx, uint32_t pos, bool x) {
            const uint64_t rc = (w ^ 0xAAAAAAAAAAAAAAAAull) & kmask;     // Util::revComplement(idx): window order, complemented
            if (!x && rc != idx) {
                const bool pickRev = rc < idx;
                const uint64_t km = pickRev ? rc : idx;
                const uint32_t p = pickRev ? (L - pos - k) : pos;
                if (a.ignoreMultiKmer) {
                    uint32_t h = (uint32_t) ((km * 0x9E3779B97F4A7C15ull) >> 40) & tmask;
                    while (true) {
                        const unsigned long long old = atomicCAS(&table[h], ~0ull, (unsigned long long) km);
                        if (old == ~0ull) break;
                        if (old == km) { dup = true; break; }
                        h = (h + 1) & tmask;
                    }
                }
                below |= km < a.kLo;
                if (inRange(a, km)) LY::store(a.keys, a.vals, base + 1 + pos, km, !pickRev, seq, L, p, a.geom); else LY::storeEmpty(a.keys, a.vals, base + 1 + pos);
                if (countHead && !a.ignoreMultiKmer && inRange(a, km)) atomicAdd(&sHead[km >> a.headShift], 1u);      // (no repeated-k-mer rule: what is stored stays)
            } else LY::storeEmpty(a.keys, a.vals, base + 1 + pos);
        };
        if (hasN) {
            for (uint32_t pos = lane; pos < nPos; pos += 64) {
                const uint64_t w = kmerWindow(a.codes, w0, pos, lastWord, k) & kmask;
                bool x = false;
                for (int j = 0; j < k; j++) x |= cdm_isN(a.nmask, w0, pos + j) != 0;
                emit(w, groupsReversed(w, k), pos, x);
            }
        } else {
            // two consecutive positions per lane: the second window is the first one shifted by a base, and its Indexer-order
            // k-mer follows from the first one's without a second bit reversal
            for (uint32_t pos = 2 * lane; pos < nPos; pos += 128) {
                const uint64_t xr = kmerWindow(a.codes, w0, pos, lastWord, k);      // k + 1 bases
                const uint64_t wA = xr & kmask, idxA = groupsReversed(wA, k);
                emit(wA, idxA, pos, false);
                if (pos + 1 < nPos) {
                    const uint64_t wB = (xr >> 2) & kmask, idxB = ((idxA << 2) & kmask) | ((xr >> (2 * k)) & 3ull);
                    emit(wB, idxB, pos + 1, false);
                }
            }
        }
        const bool anyDup = __ballot(dup) != 0ull;
        if (anyDup && lane == 0) a.slowShort[atomicAdd(&a.slowCnt[0], 1u)] = seq;   // rewritten by k_extract
        // head digits: a sequence that stays as written counts the k-mers its hash set holds (every one it stored); one that k_extract
        // rewrites is counted there
        if (countHead && a.ignoreMultiKmer && !anyDup) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (uint32_t i = lane; i < tsize; i += 64) { const unsigned long long km = table[i]; if (km != ~0ull && inRange(a, km)) atomicAdd(&sHead[km >> a.headShift], 1u); }
        }
        noteBelow(a, below);
        __builtin_amdgcn_wave_barrier();
    }
    if (countHead) headHistFlush(sHead, a.headHist);
}


// Short reads, two per wavefront: a half-wave per sequence, three consecutive positions per lane (one 64-bit window and one
// bit reversal serve all three), each half with its own LDS hash set.  A 100 bp read has 81 positions: 27 busy lanes per half
// instead of 41 of 64 with a wave per read, and the per-sequence bookkeeping is shared by two sequences.  Sequences that do not
// fit (N letters, more than 96 positions, not every k-mer taken) go to k_extract_fast through the `single` list.
constexpr int PAIR_POS = 96, PAIR_TABLE = 256;
template <typename LY>
__global__ __launch_bounds__(64 * FAST_WAVES) void k_extract_pair(ExtractArgs<LY> a) {
    __shared__ unsigned long long sTable[FAST_WAVES][2 * PAIR_TABLE];
    __shared__ unsigned int sHead[LY::bySlot ? HEAD_BINS : 1];
    const bool countHead = LY::bySlot && a.headHist != nullptr;
    if (countHead) headHistClear(sHead);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31;
    unsigned long long *table = sTable[wave] + half * PAIR_TABLE;
    const int k = a.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;
    const uint32_t nPairs = (a.n + 1) / 2;
    for (uint32_t pr = blockIdx.x * FAST_WAVES + wave; pr < nPairs; pr += gridDim.x * FAST_WAVES) {
        const uint32_t seq = 2 * pr + (uint32_t) half;
        const bool have = seq < a.n && ownedSeq(a, seq);
        uint32_t L = 0, w0 = 0; bool hasN = false; uint64_t base = 0;
        if (have) { L = a.len[seq]; w0 = a.woff[seq]; hasN = a.hasN[seq] != 0; base = slotBase(a, seq); }
        const uint32_t nPos = (L >= (uint32_t) k) ? (L - k + 1) : 0;
        const size_t cap = (size_t) (float) ((float) (a.kmersPerSeq - 1) + (a.scale * (float) L));
        const bool elig = have && nPos <= cap && nPos <= (uint32_t) PAIR_POS && !hasN;
        if (have && !elig && hl == 0) a.single[atomicAdd(&a.slowCnt[2], 1u)] = seq;
        for (int i = hl; i < PAIR_TABLE; i += 32) table[i] = ~0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
        bool dup = false, below = false;
        constexpr uint32_t NO_DIGIT = 0xFFFFFFFFu;
        // (returns the head digit of the tuple it stored, NO_DIGIT if the slot stays empty)
        auto emit = [&](uint64_t w, uint64_t idx, uint32_t pos) -> uint32_t {
            const uint64_t rc = (w ^ 0xAAAAAAAAAAAAAAAAull) & kmask;     // Util::revComplement(idx): window order, complemented
            if (rc != idx) {
                const bool pickRev = rc < idx;
                const uint64_t km = pickRev ? rc : idx;
                const uint32_t p = pickRev ? (L - pos - k) : pos;
                if (a.ignoreMultiKmer) {
                    uint32_t h = (uint32_t) ((km * 0x9E3779B97F4A7C15ull) >> 40) & (PAIR_TABLE - 1);
                    while (true) {
                        const unsigned long long old = atomicCAS(&table[h], ~0ull, (unsigned long long) km);
                        if (old == ~0ull) break;
                        if (old == km) { dup = true; break; }
                        h = (h + 1) & (PAIR_TABLE - 1);
                    }
                }
                below |= km < a.kLo;
                if (inRange(a, km)) { LY::store(a.keys, a.vals, base + 1 + pos, km, !pickRev, seq, L, p, a.geom); return (uint32_t) (km >> a.headShift); }
                LY::storeEmpty(a.keys, a.vals, base + 1 + pos);
            } else LY::storeEmpty(a.keys, a.vals, base + 1 + pos);
            return NO_DIGIT;
        };
        const uint32_t pos = 3u * (uint32_t) hl;
        uint32_t dA = NO_DIGIT, dB = NO_DIGIT, dC = NO_DIGIT;
        if (elig && pos < nPos) {
            const uint64_t xr = kmerWindow(a.codes, w0, pos, (L + 15) / 16 - 1, k);     // k + 2 bases
            const uint64_t wA = xr & kmask, idxA = groupsReversed(wA, k);
            dA = emit(wA, idxA, pos);
            if (pos + 1 < nPos) {
                const uint64_t idxB = ((idxA << 2) & kmask) | ((xr >> (2 * k)) & 3ull);
                dB = emit((xr >> 2) & kmask, idxB, pos + 1);
                if (pos + 2 < nPos) dC = emit((xr >> 4) & kmask, ((idxB << 2) & kmask) | ((xr >> (2 * k + 2)) & 3ull), pos + 2);
            }
        }
        const unsigned long long dm = __ballot(dup);
        const bool halfDup = (half ? (dm >> 32) : (dm & 0xFFFFFFFFull)) != 0ull;
        if (halfDup && hl == 0) a.slowShort[atomicAdd(&a.slowCnt[0], 1u)] = seq;   // rewritten by k_extract
        if (countHead && !halfDup) {        // (a sequence k_extract rewrites is counted there)
            if (dA != NO_DIGIT) atomicAdd(&sHead[dA], 1u);
            if (dB != NO_DIGIT) atomicAdd(&sHead[dB], 1u);
            if (dC != NO_DIGIT) atomicAdd(&sHead[dC], 1u);
        }
        noteBelow(a, below);
        __builtin_amdgcn_wave_barrier();
    }
    if (countHead) headHistFlush(sHead, a.headHist);
}

// sort element of the per-sequence ordering compareByScoreReverse (kmermatcher.h:30-46): (score, kmer|bit63, pos)
struct SeqPos { uint64_t a, b; };   // a = score << 48 | kmer63 >> 15 ; b = (kmer63 & 0x7FFF) << 49 | pos << 1 | forward
__device__ __forceinline__ bool spLess(const SeqPos &x, const SeqPos &y) { return x.a < y.a || (x.a == y.a && x.b < y.b); }
__device__ __forceinline__ uint64_t spKmer63(const SeqPos &x) { return ((x.a & 0xFFFFFFFFFFFFull) << 15) | (x.b >> 49); }
__device__ __forceinline__ uint32_t spScore(const SeqPos &x) { return (uint32_t) (x.a >> 48); }
__device__ __forceinline__ uint32_t spPos(const SeqPos &x) { return (uint32_t) ((x.b >> 1) & 0xFFFFFFFFFFFFull); }

// the (score, k-mer, position, strand) record of the k-mer at pos, false if it has an N or is its own reverse complement
// (Sequence::nextKmer + Indexer::computeKmerIdx, canonical pick kmermatcher.cpp:155-190)
template <typename LY>
__device__ __forceinline__ bool makeSeqPos(const ExtractArgs<LY> &a, uint32_t w0, uint32_t L, uint32_t lastWord, bool hasN, int k, uint32_t pos, SeqPos &e) {
    // k bases starting at pos in MMseqs coding (gray code of ours), first base most significant
    uint64_t idx = 0; bool x = false;
    for (int j = 0; j < k; j += 16) {
        uint32_t w = cdm_window16(a.codes, w0, pos + j, lastWord);
        w ^= (w >> 1) & 0x55555555u;                       // A,C,G,T -> A,C,T,G
        const int take = min(16, k - j);
        for (int b = 0; b < take; b++) idx = (idx << 2) | ((w >> (2 * b)) & 3u);
    }
    if (hasN) for (int j = 0; j < k; j++) x |= cdm_isN(a.nmask, w0, pos + j) != 0;
    if (x) return false;
    const uint64_t rc = revComplement(idx, k);
    if (rc == idx) return false;
    const bool pickRev = rc < idx;
    const uint64_t km = pickRev ? rc : idx;
    const uint32_t score = (uint32_t) (xxh64_u64(km, a.seed) & 0xFFFFu);
    const uint32_t p = pickRev ? (L - pos - k) : pos;
    e.a = ((uint64_t) score << 48) | (km >> 15);
    e.b = ((km & 0x7FFFull) << 49) | ((uint64_t) p << 1) | (pickRev ? 0ull : 1ull);
    return true;
}

// SequencePosition::compareByScoreReverse (kmermatcher.h:29-46): score, k-mer without the strand bit, position - NOT the strand
__device__ __forceinline__ bool spCmp(const SeqPos &x, const SeqPos &y) { return x.a < y.a || (x.a == y.a && (x.b >> 1) < (y.b >> 1)); }

// libstdc++'s std::sort (bits/stl_algo.h: introsort with median-of-three, threshold 16, heap sort below the depth limit, final
// insertion sort), statement for statement.  The reference sorts a sequence's k-mers with it (SORT_SERIAL, kmermatcher.cpp:271)
// and its comparator ignores the strand: when a sequence carries the same canonical k-mer at the same stored position on both
// strands, which of the two comes first - and with it the strand of a tuple - is whatever this algorithm leaves.  Serial, one
// thread; only such sequences come here.
__device__ void stdAdjustHeap(SeqPos *first, long holeIndex, long len, SeqPos value) {
    const long topIndex = holeIndex;
    long secondChild = holeIndex;
    while (secondChild < (len - 1) / 2) {
        secondChild = 2 * (secondChild + 1);
        if (spCmp(first[secondChild], first[secondChild - 1])) secondChild--;
        first[holeIndex] = first[secondChild];
        holeIndex = secondChild;
    }
    if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
        secondChild = 2 * (secondChild + 1);
        first[holeIndex] = first[secondChild - 1];
        holeIndex = secondChild - 1;
    }
    long parent = (holeIndex - 1) / 2;                          // __push_heap
    while (holeIndex > topIndex && spCmp(first[parent], value)) { first[holeIndex] = first[parent]; holeIndex = parent; parent = (holeIndex - 1) / 2; }
    first[holeIndex] = value;
}
__device__ void stdHeapSort(SeqPos *first, long n) {           // __partial_sort(first, last, last): make_heap + sort_heap
    if (n >= 2) {
        long parent = (n - 2) / 2;
        while (true) { const SeqPos v = first[parent]; stdAdjustHeap(first, parent, n, v); if (parent == 0) break; parent--; }
    }
    for (long last = n; last > 1;) { --last; const SeqPos v = first[last]; first[last] = first[0]; stdAdjustHeap(first, 0, last, v); }
}
__device__ void stdUnguardedLinearInsert(SeqPos *base, long last) {
    const SeqPos val = base[last];
    long next = last - 1;
    while (spCmp(val, base[next])) { base[last] = base[next]; last = next; --next; }
    base[last] = val;
}
__device__ void stdInsertionSort(SeqPos *base, long first, long last) {
    if (first == last) return;
    for (long i = first + 1; i != last; ++i) {
        if (spCmp(base[i], base[first])) { const SeqPos val = base[i]; for (long j = i; j > first; j--) base[j] = base[j - 1]; base[first] = val; }
        else stdUnguardedLinearInsert(base, i);
    }
}
__device__ void stdSort(SeqPos *base, long n) {
    if (n <= 0) return;
    // __introsort_loop with an explicit stack for its one recursive call
    long stFirst[64], stLast[64]; int stDepth[64]; int top = 0;
    int lg = 0; for (long v = n; v > 1; v >>= 1) lg++;
    stFirst[0] = 0; stLast[0] = n; stDepth[0] = 2 * lg; top = 1;
    while (top > 0) {
        top--;
        long first = stFirst[top], last = stLast[top]; int depth = stDepth[top];
        while (last - first > 16) {
            if (depth == 0) { stdHeapSort(base + first, last - first); break; }
            --depth;
            // __unguarded_partition_pivot
            const long mid = first + (last - first) / 2, ia = first + 1, ib = mid, ic = last - 1;
            long m;                                             // __move_median_to_first(first, a, b, c)
            if (spCmp(base[ia], base[ib])) { if (spCmp(base[ib], base[ic])) m = ib; else if (spCmp(base[ia], base[ic])) m = ic; else m = ia; }
            else if (spCmp(base[ia], base[ic])) m = ia; else if (spCmp(base[ib], base[ic])) m = ic; else m = ib;
            { const SeqPos t = base[first]; base[first] = base[m]; base[m] = t; }
            long lo = first + 1, hi = last;                     // __unguarded_partition(first + 1, last, first)
            while (true) {
                while (spCmp(base[lo], base[first])) ++lo;
                --hi;
                while (spCmp(base[first], base[hi])) --hi;
                if (!(lo < hi)) break;
                { const SeqPos t = base[lo]; base[lo] = base[hi]; base[hi] = t; }
                ++lo;
            }
            const long cut = lo;
            stFirst[top] = cut; stLast[top] = last; stDepth[top] = depth; top++;     // __introsort_loop(cut, last, depth)
            last = cut;
        }
    }
    // __final_insertion_sort
    if (n > 16) { stdInsertionSort(base, 0, 16); for (long i = 16; i != n; ++i) stdUnguardedLinearInsert(base, i); }
    else stdInsertionSort(base, 0, n);
}

// One workgroup of NT threads per sequence; CAP = power of two >= number of k-mers of the sequence, records in LDS; CAP = 0:
// records in a per-block slice of global scratch (sequences with 4096 positions or more: rare, speed is not the point).
template <typename LY, int CAP, int NT>
__global__ __launch_bounds__(NT) void k_extract(ExtractArgs<LY> a) {
    __shared__ SeqPos sSp[CAP ? CAP : 1];
    __shared__ uint8_t sSel[CAP ? CAP : 1];
    SeqPos *sp = CAP ? sSp : a.hugeSp + (size_t) blockIdx.x * a.hugeCap;
    uint8_t *sel = CAP ? sSel : a.hugeSel + (size_t) blockIdx.x * a.hugeCap;
    __shared__ uint32_t sN, sCursor;
    // CAP = 0: the head of the sorted records is copied to LDS for the serial selection walk (one thread chasing through global
    // scratch took milliseconds per contig; the walk ends after ~0.2 n + 200 records)
    constexpr uint32_t HEADN = CAP ? 1 : 3072;
    __shared__ SeqPos sHead[HEADN];
    __shared__ unsigned int sDigits[LY::bySlot ? HEAD_BINS : 1];
    const bool countHead = LY::bySlot && a.headHist != nullptr;
    if (countHead) headHistClear(sDigits);
    const int tid = threadIdx.x;
    for (uint32_t item = blockIdx.x; item < a.nList; item += gridDim.x) {
        const uint32_t seq = a.list[item];
        const uint32_t L = a.len[seq], w0 = a.woff[seq];
        const bool hasN = a.hasN[seq] != 0;
        const int k = a.k;
        const uint32_t nPos = (L >= (uint32_t) k) ? (L - k + 1) : 0;
        if (tid == 0) sN = 0;
        __syncthreads();
        // ---- k-mers (Sequence::nextKmer + Indexer::computeKmerIdx, canonical pick kmermatcher.cpp:155-190)
        const uint32_t lastWord = (L + 15) / 16 - 1;
        for (uint32_t pos = tid; pos < nPos; pos += NT) {
            SeqPos e;
            if (!makeSeqPos(a, w0, L, lastWord, hasN, k, pos, e)) continue;
            const uint32_t slot = atomicAdd(&sN, 1u);
            sp[slot] = e;
        }
        __syncthreads();
        const uint32_t n = sN;
        // ---- sort by (score, kmer, pos): bitonic over the next power of two, padding = max
        uint32_t np2 = 1; while (np2 < n) np2 <<= 1;
        for (uint32_t i = n + tid; i < np2; i += NT) { sp[i].a = ~0ull; sp[i].b = ~0ull; }
        __syncthreads();
        if constexpr (CAP == 0) {
            // The records are in global scratch.  The same bitonic network, but every exchange over a distance below CH happens in
            // LDS: the array is taken CH records at a time (sHead's memory - it holds the head of the sorted records only later), all
            // the network's stages that stay inside such a stretch run there, and only the exchanges over CH records or more go
            // through memory - 28 passes over the array instead of 153 for a 100 k-letter contig (this kernel was half of the
            // device time of the workflow loop's contig iterations).
            constexpr uint32_t CH = 2048;
            static_assert(CH <= HEADN, "the chunk lives in sHead");
            const uint32_t cs = min(np2, CH);
            auto ldsStages = [&](uint32_t c0, uint32_t size, uint32_t strideFrom) {      // stages of `size` with stride <= strideFrom on [c0, c0 + cs)
                for (uint32_t i = tid; i < cs; i += NT) sHead[i] = sp[c0 + i];
                __syncthreads();
                for (uint32_t stride = strideFrom; stride > 0; stride >>= 1) {
                    for (uint32_t t = tid; t < cs / 2; t += NT) {
                        const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                        const bool up = ((c0 + lo) & size) == 0;
                        SeqPos x = sHead[lo], y = sHead[hi];
                        if (spLess(y, x) == up) { sHead[lo] = y; sHead[hi] = x; }
                    }
                    __syncthreads();
                }
            };
            for (uint32_t c0 = 0; c0 < np2; c0 += cs) {          // sizes up to the chunk: wholly in LDS
                for (uint32_t i = tid; i < cs; i += NT) sHead[i] = sp[c0 + i];
                __syncthreads();
                for (uint32_t size = 2; size <= cs; size <<= 1)
                    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                        for (uint32_t t = tid; t < cs / 2; t += NT) {
                            const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                            const bool up = ((c0 + lo) & size) == 0;
                            SeqPos x = sHead[lo], y = sHead[hi];
                            if (spLess(y, x) == up) { sHead[lo] = y; sHead[hi] = x; }
                        }
                        __syncthreads();
                    }
                for (uint32_t i = tid; i < cs; i += NT) sp[c0 + i] = sHead[i];
                __syncthreads();
            }
            for (uint32_t size = 2 * cs; size <= np2 && size != 0; size <<= 1) {
                for (uint32_t stride = size >> 1; stride >= cs; stride >>= 1) {
                    for (uint32_t t = tid; t < np2 / 2; t += NT) {
                        const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                        const bool up = (lo & size) == 0;
                        SeqPos x = sp[lo], y = sp[hi];
                        if (spLess(y, x) == up) { sp[lo] = y; sp[hi] = x; }
                    }
                    __syncthreads();
                }
                for (uint32_t c0 = 0; c0 < np2; c0 += cs) {
                    ldsStages(c0, size, cs >> 1);
                    for (uint32_t i = tid; i < cs; i += NT) sp[c0 + i] = sHead[i];
                    __syncthreads();
                }
            }
        } else
        for (uint32_t size = 2; size <= np2; size <<= 1)
            for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (uint32_t t = tid; t < np2 / 2; t += NT) {
                    const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    SeqPos x = sp[lo], y = sp[hi];
                    if (spLess(y, x) == up) { sp[lo] = y; sp[hi] = x; }
                }
                __syncthreads();
            }
        // The bitonic order breaks comparator ties by the strand bit.  The reference's comparator has no such rule: if two records
        // tie (same score, k-mer, stored position, opposite strands) redo the sort the way the reference does, from the order
        // in which fillKmerPositionArray generated the records (ascending position).
        {
            int tie = 0;
            for (uint32_t i = tid; i + 1 < n; i += NT) tie |= (sp[i].a == sp[i + 1].a && (sp[i].b >> 1) == (sp[i + 1].b >> 1));
            if (__syncthreads_or(tie && a.ignoreMultiKmer)) {
                if (tid == 0) {
                    uint32_t m = 0;
                    for (uint32_t pos = 0; pos < nPos; pos++) { SeqPos e; if (makeSeqPos(a, w0, L, lastWord, hasN, k, pos, e)) sp[m++] = e; }
                    stdSort(sp, (long) m);
                }
                __syncthreads();
            }
        }
        // ---- selection (kmermatcher.cpp:224-240, 277-350)
        const size_t considered = min((size_t) (float) ((float) (a.kmersPerSeq - 1) + (a.scale * (float) L)), (size_t) n);
        uint32_t headN = 0;
        if (CAP == 0) { headN = min(n, HEADN); for (uint32_t i = tid; i < headN; i += NT) sHead[i] = sp[i]; }
        // fast path test: no two equal k-mers next to each other, and every k-mer is taken
        int dup = 0;
        for (uint32_t i = tid; i + 1 < n; i += NT) dup |= (spKmer63(sp[i]) == spKmer63(sp[i + 1]));
        const int anyDup = __syncthreads_or(dup && a.ignoreMultiKmer);
        if (!anyDup && considered == n) {
            for (uint32_t i = tid; i < n; i += NT) sel[i] = 1;
        } else {
            for (uint32_t i = tid; i < n; i += NT) sel[i] = 0;
            __syncthreads();
            // The selection walk is serial (one thread), its reads are not: with the records in global scratch (CAP = 0) the block
            // stages the next HEADN records in LDS, thread 0 walks them, and so on until the walk is done (it ends after about
            // 0.2 n + 200 records; chasing them one by one through global memory took ~1 us each).
            __shared__ uint32_t wThreshold, wDone; __shared__ int wTooMuch; __shared__ unsigned long long wKi, wSelected;
            uint32_t winLo = 0, winN = headN;                 // [winLo, winLo + winN) of the sorted records is in sHead
            auto at = [&](size_t i) -> SeqPos { return (CAP == 0 && i >= winLo && i < (size_t) winLo + winN) ? sHead[i - winLo] : sp[i]; };
            if (tid == 0) {
                wDone = (n == 0) ? 1u : 0u; wKi = 0; wSelected = 0; wThreshold = 0; wTooMuch = 0;
                if (n > 0) {
                    // threshold = (score of the considered-th smallest) + 1, inBins = #(score < threshold)  [:224-240]
                    uint32_t threshold = 0; size_t inBins = 0;
                    if (considered > 0) {
                        threshold = spScore(at(considered - 1)) + 1;
                        inBins = considered;
                        while (inBins < n && spScore(at(inBins)) < threshold) inBins++;
                    } else {
                        // the reference's loops leave threshold at the start of the first non-empty 512-bin and subtract that bin
                        threshold = (spScore(at(0)) >> 9) * 512; inBins = 0;
                    }
                    wThreshold = threshold; wTooMuch = (int) (inBins - considered);
                    if (!a.ignoreMultiKmer) {
                        // without --ignore-multi-kmer the reference does not sort (:269-275): the selection walks the k-mers in the
                        // order they were generated; the threshold above only needed the score distribution
                        uint32_t m = 0;
                        for (uint32_t pos = 0; pos < nPos; pos++) { SeqPos e; if (makeSeqPos(a, w0, L, lastWord, hasN, k, pos, e)) sp[m++] = e; }
                    }
                }
            }
            __syncthreads();
            if (CAP == 0 && !a.ignoreMultiKmer) { winN = min(n, HEADN); for (uint32_t i = tid; i < winN; i += NT) sHead[i] = sp[i]; __syncthreads(); }   // (the LDS copy held the sorted order)
            while (!wDone) {
                if (tid == 0) {
                    uint32_t threshold = wThreshold; int tooMuch = wTooMuch; size_t ki = (size_t) wKi, selected = (size_t) wSelected;
                    // walk while the record and its successor are staged (CAP != 0: everything is)
                    const size_t stop = (CAP == 0) ? ((size_t) winLo + winN >= n ? n : (size_t) winLo + winN - 1) : n;
                    for (; ki < n && selected < considered; ki++) {
                        if (ki >= stop) break;
                        if (a.ignoreMultiKmer) {
                            const uint64_t km = spKmer63(at(ki));
                            if (ki + 1 < n) {
                                uint64_t nx = spKmer63(at(ki + 1));
                                if (km == nx) {
                                    while (km == nx && ki < n) { ki++; if (ki >= n) break; nx = spKmer63(at(ki)); }
                                }
                            }
                            if (ki >= n) break;
                        }
                        const uint32_t sc = spScore(at(ki));
                        if (sc < threshold) {
                            if (sc == (threshold - 1) && tooMuch) { tooMuch--; threshold -= (tooMuch == 0) ? 1 : 0; }
                            selected++;
                            sel[ki] = 1;
                        }
                    }
                    wThreshold = threshold; wTooMuch = tooMuch; wKi = ki; wSelected = selected;
                    wDone = (ki >= n || selected >= considered) ? 1u : 0u;
                }
                __syncthreads();
                if (CAP == 0 && !wDone) {      // next window starts at the record the walk stopped at
                    winLo = (uint32_t) wKi; winN = min(n - winLo, HEADN);
                    for (uint32_t i = tid; i < winN; i += NT) sHead[i] = sp[winLo + i];
                }
                __syncthreads();
            }
        }
        __syncthreads();
        // ---- emit: 1 whole-sequence tuple (:244-267) + the selected k-mers
        {
            const uint64_t base = slotBase(a, seq);
            if (tid == 0) sCursor = 0;       // (the whole-sequence tuple :244-267 is written by k_seq_hash)
            if constexpr (LY::bySlot) { for (uint32_t i = tid; i < nPos; i += NT) LY::storeEmpty(a.keys, a.vals, base + 1 + i); }      // (a slot IS a position: every one is written, the selected ones again)
            __syncthreads();
            // selected tuples first (their order within the sequence does not matter: a global sort follows), then sentinels
            for (uint32_t i = tid; i < n; i += NT) {
                if (!sel[i]) continue;
                const SeqPos e = sp[i];
                const uint64_t km = spKmer63(e);
                if (km < a.kLo && a.belowFlag[0] == 0u) a.belowFlag[0] = 1u;
                if (!inRange(a, km)) continue;              // another rank's k-mer range
                if constexpr (LY::bySlot) {
                    const bool fwd = (e.b & 1ull) != 0;
                    LY::store(a.keys, a.vals, base + 1 + (fwd ? spPos(e) : L - spPos(e) - (uint32_t) k), km, fwd, seq, L, spPos(e), a.geom);
                    if (countHead) atomicAdd(&sDigits[km >> a.headShift], 1u);
                } else {
                const uint32_t o = atomicAdd(&sCursor, 1u);
                LY::store(a.keys, a.vals, base + 1 + o, km, (e.b & 1ull) != 0, seq, L, spPos(e), a.geom);
                }
            }
            __syncthreads();
            if constexpr (!LY::bySlot) { for (uint32_t i = sCursor + tid; i < nPos; i += NT) LY::storeEmpty(a.keys, a.vals, base + 1 + i); }
        }
        __syncthreads();
    }
    if (countHead) headHistFlush(sDigits, a.headHist);
}

// ------------------------------------------------------------------------------------------------ K3: groups
struct GroupParams {
    uint64_t n;
    int onlyExtendable, covMode; float covThr;
    uint32_t idBits, diagBits; int diagBias;
    int wide;                   // group keys without the representative (runsort.h: GK_START / GK_DROPPED mark the run starts)
    uint64_t first;             // the kernel covers the tuples [first, n)
    uint64_t firstRunIdx;       // index of the array's very first tuple in this view (0; ~0 if the view does not hold it)
    unsigned long long *stat;   // STAT_STRIPES counters: members kept
};
// Count of kept members, one atomic per wave, striped over many addresses: millions of waves hitting one counter serialise
// (370 ms instead of 53 for k_bucket_groups at 50 M reads).  cnt is wave-uniform.
constexpr int STAT_STRIPES = 4096;
__device__ __forceinline__ void waveGroupStats(unsigned long long *stat, uint32_t cnt) {
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(stat + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (STAT_STRIPES - 1)), (unsigned long long) cnt);
}
template <typename LY> struct GroupArgs : GroupParams {
    const uint64_t *keys; const typename LY::V *vals;   // sorted by k-mer
    TupleGeom geom;
};
__device__ __forceinline__ bool canBeCoveredK(float covThr, int covMode, float ql, float tl) {
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
// key layout of the second sort: [ rep | id | diagonal + bias | strand ] , strand (1 = query needs no reversal) in bit 0; the wide form
// (DBs whose ids and diagonals do not leave room for the representative in 63 bits) is [ id | diagonal + bias | strand ]
__device__ __forceinline__ uint64_t packGroupKey(const GroupParams &a, uint32_t rep, uint32_t id, int diag, bool noRev) {
    const uint64_t hi = a.wide ? (uint64_t) id : (((uint64_t) rep << a.idBits) | id);
    return (hi << (a.diagBits + 1)) | ((uint64_t) (uint32_t) (diag + a.diagBias) << 1) | (noRev ? 1ull : 0ull);
}
// wide form: the first slot of a k-mer run with members names the representative (its own tuple: id == rep), kept or not
__device__ __forceinline__ uint64_t markRunStart(const GroupParams &a, uint64_t gk, uint32_t rep) {
    return (runsort::gkKept(gk) ? gk : (runsort::GK_DROPPED | ((uint64_t) rep << (a.diagBits + 1)))) | runsort::GK_START;
}
// run start index of every tuple = inclusive max-scan of (start ? i : 0); fed to the scan through this functor
template <typename LY> struct StartIndex {
    const uint64_t *keys; TupleGeom geom; unsigned long long first;
    __device__ unsigned long long operator()(unsigned long long i) const {
        if (i == first) return i;
        const uint64_t a = keys[i], b = keys[i - 1];
        const bool start = (a == ~0ull) || (b == ~0ull) || (LY::kmerOf(a, i, geom) != LY::kmerOf(b, i - 1, geom));
        return start ? i : 0ull;
    }
};
template <typename LY> struct StartFrom {      // the scan's index starts at 0, the tuples at `first`
    StartIndex<LY> f;
    __device__ __forceinline__ unsigned long long operator()(size_t i) const { return f(f.first + (unsigned long long) i); }
};

// K3, one thread per tuple.  The tuple array was filled in (sequence length descending, id ascending, position) order and
// the radix sort is stable, so the first tuple of a k-mer run is the reference's representative (sort order
// kmermatcher.h:76-96); only a k-mer that the representative's own sequence carries twice needs a look at the next tuples.
// (rep, id, diagonal, strand) key of one member of a k-mer run, ~0 if the member is dropped (assignGroup :453-562)
__device__ __forceinline__ uint64_t groupKeyCore(const GroupParams &a, uint32_t repId, int queryLen, int repPos, bool repIsReverse,
                                                 uint32_t id, int tLen, int tPos0, bool targetIsReverse) {
    int qPos, tPos; bool qRev;
    if (repIsReverse && !targetIsReverse) { qPos = repPos; tPos = tPos0; qRev = true; }
    else if (repIsReverse && targetIsReverse) { qPos = (queryLen - 1) - repPos; tPos = (tLen - 1) - tPos0; qRev = false; }
    else if (!repIsReverse && targetIsReverse) { qPos = (queryLen - 1) - repPos; tPos = (tLen - 1) - tPos0; qRev = true; }
    else { qPos = repPos; tPos = tPos0; qRev = false; }
    // (the reference holds positions and the diagonal in `short` below 32 765 letters and in `int` above, kmermatcher.cpp:803-808;
    // below that limit the casts never change a value, so one expression serves both paths)
    const int diagonal = qPos - tPos;
    const bool canBeExtended = diagonal < 0 || (diagonal > (queryLen - tLen));
    // coverage modes 0-2 with a threshold <= 0 hold for any two positive lengths: skip the divisions
    const bool cbc = (a.covThr <= 0.0f && a.covMode <= 2 && queryLen > 0 && tLen > 0) ? true : canBeCoveredK(a.covThr, a.covMode, (float) queryLen, (float) tLen);
    const bool keep = (a.onlyExtendable == 0 && cbc) || (canBeExtended && a.onlyExtendable != 0);
    return keep ? packGroupKey(a, repId, id, diagonal, !qRev) : ~0ull;
}
template <typename LY>
__device__ __forceinline__ uint64_t groupKeyOf(const GroupParams &a, const TupleGeom &geom, uint64_t repKey, typename LY::V repVal, uint64_t repSlot, uint32_t repPos0,
                                               bool firstRun, uint64_t key, typename LY::V v, uint64_t slot) {
    // the reference initialises repIsReverse = false and only updates it when a NEW run starts (:465,:535-538):
    // the very first run of the array keeps false whatever its strand
    return groupKeyCore(a, LY::seqOf(repVal), (int) LY::lenOf(repKey, repVal, repSlot, geom), (int) repPos0, firstRun ? false : ((repKey & BIT63) == 0),
                        LY::seqOf(v), (int) LY::lenOf(key, v, slot, geom), (int) LY::posOf(key, v, slot, geom), (key & BIT63) == 0);
}

template <typename LY>
__global__ __launch_bounds__(256) void k_groups(GroupArgs<LY> a, unsigned long long *__restrict__ startIo /* in: run start, out: packed key */) {
    const uint64_t i = a.first + (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long gk = ~0ull;
    if (i < a.n) {
        const uint64_t key = a.keys[i];
        if (key != ~0ull) {                                        // (an unused slot has no group key)
            const uint64_t st = startIo[i];
            const uint64_t km = LY::kmerOf(key, i, a.geom);
            const bool hasNext = (i + 1 < a.n) && a.keys[i + 1] != ~0ull && LY::kmerOf(a.keys[i + 1], i + 1, a.geom) == km;
            if (st != i || hasNext) {                              // singletons are dropped (:479)
                uint64_t bestKey = a.keys[st]; const typename LY::V best = a.vals[st];
                const uint32_t repId = LY::seqOf(best);
                uint32_t bestPos = LY::posOf(bestKey, best, st, a.geom);
                // same sequence twice in the run: the smaller position wins (rare)
                for (uint64_t e = st + 1; e < a.n && a.keys[e] != ~0ull && LY::kmerOf(a.keys[e], e, a.geom) == km && LY::seqOf(a.vals[e]) == repId; e++) {
                    const uint32_t pe = LY::posOf(a.keys[e], a.vals[e], e, a.geom);
                    if (pe < bestPos) { bestPos = pe; bestKey = a.keys[e]; }
                }
                gk = groupKeyOf<LY>(a, a.geom, bestKey, best, st, bestPos, st == a.firstRunIdx, key, a.vals[i], i);
                if (a.wide && st == i) gk = markRunStart(a, gk, repId);
            }
        }
        startIo[i] = gk;
    }
    waveGroupStats(a.stat, (uint32_t) __popcll(__ballot(runsort::gkKept(gk))));
}

// K2b + K3 fused for region 1 when only the top bits of the k-mer went through the global radix passes (bucket.h): a wave
// sorts a group of buckets on the remaining low bits in registers, finds the k-mer runs in the sorted order and writes the
// group keys of the members straight to their final slots.  W = word of the network: (bucket ordinal, low bits, position).
// LayoutSlot: where the slots of a grouping block lie in their head-digit segment - the head digit of the block's first slot and how far
// the segment reaches to either side of it, in slots relative to that first slot (clipped to 2^30; one 16-byte load per block: looking
// the digit up in the segment table is nine dependent loads, which every wave of this latency-bound kernel paid at its start)
struct BlockHead { int32_t lo, hi; uint32_t td, pad; };
#ifndef CDM_REC_CAP
#define CDM_REC_CAP 32
#endif
constexpr int REC_CAP = CDM_REC_CAP;     // staged run records per wave of the grouping kernel (its owned slots hold ~8 k-mer runs per 128 at 20x coverage)
static_assert(REC_CAP <= 64, "a wave writes its stage out with one lane per record");
__global__ void k_block_heads(TupleGeom geom, uint64_t n, uint64_t perBlock, uint64_t blocks, BlockHead *__restrict__ out) {
    const uint64_t b = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= blocks) return;
    const uint64_t base = b * perBlock;
    const uint32_t td = headDigit(geom, base);
    const uint64_t since = base - geom.seg[td], until = geom.seg[td + 1] - base;
    BlockHead h; h.td = td; h.pad = 0; h.lo = -(int32_t) min(since, (uint64_t) 1 << 30); h.hi = (int32_t) min(until, (uint64_t) 1 << 30);
    out[b] = h;
}
// The staged run records of REC_WAVES consecutive waves of the grouping kernel, packed: off = exclusive sums of the waves' counts.
// The buckets the grouping kernel left to the caller have records of their own (bigVal: sorted by start, nBig of them) - a packed record
// moves back by the number of those that start in front of it, so that the two lists interleave in k-mer order (k_rec_place_big puts
// the others in).
constexpr int REC_WAVES = 256, REC_BIG_LDS = 256;
struct RecCount { const uint8_t *c; __device__ __forceinline__ unsigned long long operator()(size_t i) const { return c[i]; } };
__device__ __forceinline__ uint64_t lowerBoundStart(const uint64_t *__restrict__ val, uint64_t n, uint64_t start) {      // first record whose start is >= start
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if ((val[mid] >> runsort::RUN_CNT_BITS) < start) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(256) void k_rec_compact(const uint32_t *__restrict__ stRep, const uint64_t *__restrict__ stVal, const unsigned long long *__restrict__ off, uint64_t waves,
                                                     uint64_t own, const uint64_t *__restrict__ bigVal, uint64_t nBig, uint32_t *__restrict__ recRep, uint64_t *__restrict__ recVal) {
    __shared__ unsigned long long sOff[REC_WAVES + 1];
    __shared__ uint64_t sBig[REC_BIG_LDS];
    __shared__ uint64_t sB[2];
    const uint64_t w0 = (uint64_t) blockIdx.x * REC_WAVES;
    const int nw = (int) min((uint64_t) REC_WAVES, waves - w0);
    for (int i = threadIdx.x; i <= nw; i += 256) sOff[i] = off[w0 + i];
    if (threadIdx.x < 2) sB[threadIdx.x] = nBig ? lowerBoundStart(bigVal, nBig, (w0 + (threadIdx.x ? (uint64_t) nw : 0ull)) * own) : 0ull;      // the big records inside this block's slots
    __syncthreads();
    const uint64_t b0 = sB[0], nb = sB[1] - sB[0];
    for (uint64_t i = threadIdx.x; i < nb && i < (uint64_t) REC_BIG_LDS; i += 256) sBig[i] = bigVal[b0 + i] >> runsort::RUN_CNT_BITS;
    __syncthreads();
    const unsigned long long base = sOff[0], total = sOff[nw] - base;
    for (unsigned long long i = threadIdx.x; i < total; i += 256) {
        int w = 0;
#pragma unroll
        for (int st = REC_WAVES / 2; st > 0; st >>= 1) if (w + st < nw && sOff[w + st] - base <= i) w += st;
        const uint64_t src = (w0 + (uint64_t) w) * REC_CAP + (i - (sOff[w] - base));
        const uint64_t v = stVal[src], start = v >> runsort::RUN_CNT_BITS;
        uint64_t before = b0;
        if (nb <= (uint64_t) REC_BIG_LDS) { for (uint64_t q = 0; q < nb; q++) before += sBig[q] < start; }
        else before = lowerBoundStart(bigVal, nBig, start);
        recRep[base + i + before] = stRep[src]; recVal[base + i + before] = v;
    }
}
// big record b goes behind the packed records that start in front of it: those of the waves in front of the wave that owns its first
// slot, and that wave's own ones with a smaller start
__global__ __launch_bounds__(256) void k_rec_place_big(const uint32_t *__restrict__ bigRep, const uint64_t *__restrict__ bigVal, uint64_t nBig, const uint64_t *__restrict__ stVal,
                                                       const uint8_t *__restrict__ stCnt, const unsigned long long *__restrict__ off, uint64_t own, uint32_t *__restrict__ recRep, uint64_t *__restrict__ recVal) {
    const uint64_t b = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBig) return;
    const uint64_t v = bigVal[b], start = v >> runsort::RUN_CNT_BITS, w = start / own;
    uint64_t before = off[w];
    const int c = stCnt[w];
    for (int j = 0; j < c; j++) before += (stVal[w * REC_CAP + j] >> runsort::RUN_CNT_BITS) < start;
    recRep[b + before] = bigRep[b]; recVal[b + before] = v;
}
// the group keys of the big buckets (dense staging array, ranges = (start, end, offset)) with one dropped key behind every range: run
// records made from that array (k_run_records) never span two buckets
__global__ __launch_bounds__(256) void k_big_gap_copy(const unsigned long long *__restrict__ ranges, unsigned int cnt, const unsigned long long *__restrict__ dense, unsigned long long *__restrict__ gapped) {
    const unsigned int lane = threadIdx.x & 63, wavesPerGrid = gridDim.x * 4;
    for (unsigned int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < cnt; r += wavesPerGrid) {
        const unsigned long long len = ranges[3 * (size_t) r + 1] - ranges[3 * (size_t) r], o = ranges[3 * (size_t) r + 2];
        for (unsigned long long i = lane; i < len; i += 64) gapped[o + r + i] = dense[o + i];
        if (lane == 0) gapped[o + r + len] = ~0ull;
    }
}
// their records' starts from the gapped array's coordinates to slots of the key array
__global__ __launch_bounds__(256) void k_big_rec_starts(const unsigned long long *__restrict__ ranges, unsigned int cnt, uint64_t *__restrict__ recVal, uint64_t nRec) {
    const uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nRec) return;
    const uint64_t v = recVal[j], x = v >> runsort::RUN_CNT_BITS;
    unsigned int lo = 0, hi = cnt;                          // last range r with offset + r <= x
    while (hi - lo > 1) { const unsigned int mid = lo + ((hi - lo) >> 1); if (ranges[3 * (size_t) mid + 2] + mid <= x) lo = mid; else hi = mid; }
    const uint64_t slot = ranges[3 * (size_t) lo] + (x - (ranges[3 * (size_t) lo + 2] + lo));
    recVal[j] = (slot << runsort::RUN_CNT_BITS) | (v & ((1ull << runsort::RUN_CNT_BITS) - 1ull));
}
template <typename LY, typename W>
struct BucketGroupArgs : GroupParams {
    const BlockHead *blockHead = nullptr;
    // Run records of sort 2 (runsort.h), emitted while the group keys are written: every wave stages the records of the k-mer runs it
    // finishes - (representative, first slot << 13 | kept keys) per stretch of kept keys - in its own REC_CAP entries (k_rec_compact packs
    // them: the waves in order are the k-mer order).  NULL: not staged.
    uint32_t *recRep = nullptr; uint64_t *recVal = nullptr; uint8_t *recCnt = nullptr;
    uint32_t recLimit = REC_CAP;        // (CDM_REC_LIMIT lowers it: tests reach the overflow list)
    // what a wave's stage does not hold goes to a global list (a cursor, ovCap entries; in arrival order - the caller sorts it by start and
    // merges it like the big buckets' records); only a list that is too small sends the caller back to k_run_records
    uint32_t *ovRep = nullptr; uint64_t *ovVal = nullptr; unsigned long long *ovCursor = nullptr; unsigned long long ovCap = 0;
    const uint64_t *keys; const typename LY::V *vals; TupleGeom geom;
    unsigned long long *out;        // group key (or ~0) per slot, in k-mer order
    int lowBits;                    // k-mer bits the global passes left unsorted
    int own; uint32_t maxBucket; bucket::BigList big;
};
// Geometry of the grouping kernel: a smaller window than bucket.h's default - the kernel runs on the latency of its loads and LDS
// round trips (its time scales with 1 / waves per CU), so the LDS a wave needs decides its speed; k-mer buckets are ~40 tuples.
#ifndef CDM_GK_OWN
#define CDM_GK_OWN 128
#define CDM_GK_WIN 384
#define CDM_GK_FIRST 256
#endif
// With slot tuples (8 bytes per window slot, no value array) twice the owned range costs the LDS the (key, value) window did: 256 owned
// slots in a window of 512 take the kernel from 59 to 53 ms at 50 M reads (384 / 640, 512 / 768 and a 768-slot window for buckets of
// up to 512 all lose: 64-66 ms; profiles/r05_probe_grouping_geometry.txt).
#ifndef CDM_GKS_OWN
#define CDM_GKS_OWN 256
#define CDM_GKS_WIN 512
#define CDM_GKS_FIRST 384
#endif
template <typename LY> struct GkGeom {
    static constexpr int OWN = LY::bySlot ? CDM_GKS_OWN : CDM_GK_OWN, WIN = LY::bySlot ? CDM_GKS_WIN : CDM_GK_WIN, FIRST = LY::bySlot ? CDM_GKS_FIRST : CDM_GK_FIRST, MAXB = WIN - OWN;
    // (the network writes all 64 R slots of ss, R = 1, 2, 4, 8: the largest bucket is one of those sizes)
    static_assert(WIN % 64 == 0 && FIRST % 64 == 0 && FIRST < WIN && OWN <= FIRST && WIN <= (1 << bucket::WV_IDX) && (MAXB == 64 || MAXB == 128 || MAXB == 256 || MAXB == 512), "grouping kernel geometry");
};
#ifndef CDM_GK_MINW
#define CDM_GK_MINW 0      // waves per SIMD the register allocation of the grouping kernel leaves room for (scripts/build_variant.py sweeps it; 0: as many as its LDS lets run - 6 blocks of 4 waves per CU with (key, value) pairs in the window, 7 with slot tuples)
#endif
template <typename LY> constexpr int gkMinWaves() { return CDM_GK_MINW ? CDM_GK_MINW : (LY::bySlot ? 7 : 1); }
template <typename LY, typename W>
__global__ __launch_bounds__(bucket::BK_NT, gkMinWaves<LY>()) void k_bucket_groups(BucketGroupArgs<LY, W> a) {
    using namespace bucket;
    typedef typename LY::V V;
    constexpr int GK_WIN = GkGeom<LY>::WIN, GK_FIRST = GkGeom<LY>::FIRST, GK_MAXB = GkGeom<LY>::MAXB;
    __shared__ uint64_t sKeyAll[BK_WAVES][GK_WIN];
    __shared__ V sValAll[BK_WAVES][GK_WIN];
    __shared__ uint32_t sSAll[BK_WAVES][GK_MAXB];
    __shared__ WaveLdsT<GK_WIN> wAll[BK_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t r0 = ((uint64_t) blockIdx.x * BK_WAVES + wave) * (uint64_t) a.own;
    if (r0 >= a.n) return;      // (whole wave)
    uint64_t *sKey = sKeyAll[wave]; V *sVal = sValAll[wave]; uint32_t *ss = sSAll[wave];
    WaveLdsT<GK_WIN> &w = wAll[wave];
    const uint64_t hmask = (2ull << a.geom.kbits) - 1ull, lowMask = (1ull << a.lowBits) - 1ull;   // k-mer bits + the unused-slot bit
    const int lowBits = a.lowBits;
    struct Tup { uint64_t k; V v; };
    constexpr uint32_t IDXM = (1u << WV_IDX) - 1u;
    uint32_t keptCnt = 0;       // wave-uniform
    uint32_t recCount = 0, prevRowKept = 0;      // (lane 0's copies are complete) run records staged so far; was the last key of the row in front kept
    // (staged in LDS, written out in one piece at the wave's end)
    // Only the slot layout's instance stages records: with (key, value) pairs in the window the stage's LDS and registers take the kernel
    // from six blocks per CU to five (63 instead of 49 ms), more than k_run_records' pass over the keys costs.
    constexpr bool REC = LY::bySlot;
    __shared__ uint32_t sRecRep[BK_WAVES][REC ? REC_CAP : 1];
    __shared__ uint64_t sRecVal[BK_WAVES][REC ? REC_CAP : 1];
    const uint32_t recLimit = a.recLimit;
    const uint64_t waveIdx = (uint64_t) blockIdx.x * BK_WAVES + wave;
    // slot tuples (LayoutSlot): a tuple becomes its (key, id) pair as it is loaded; the head digit is the segment the index lies in -
    // the wave's own one for nearly every tuple of its window (segments are millions of tuples long)
    BlockHead bh; bh.lo = 0; bh.hi = 0; bh.td = 0; bh.pad = 0;
    const uint64_t blockBase = (uint64_t) blockIdx.x * BK_WAVES * (uint64_t) a.own;
    if constexpr (LY::bySlot) bh = a.blockHead[blockIdx.x];
    auto digitAt = [&](uint64_t g) -> uint32_t { const long long off = (long long) (g - blockBase); return (off >= (long long) bh.lo && off < (long long) bh.hi) ? bh.td : headDigit(a.geom, g); };
    waveBuckets<Tup, GK_WIN, GK_FIRST>(r0, a.n, a.own, a.maxBucket, hmask & ~lowMask, a.big, w, lane,
        [&](uint64_t g) {
            Tup t;
            if constexpr (LY::bySlot) { t.k = a.keys[g]; t.v = digitAt(g); }       // (raw: the window holds far more tuples than the wave owns)
            else { t.k = a.keys[g]; t.v = a.vals[g]; }
            return t;
        },
        [&](int i, const Tup &t) {
            sKey[i] = t.k;
            if constexpr (LY::bySlot) return ((uint64_t) t.v << a.geom.headShift) | (t.k >> rx::SLOT_KEY_SHIFT);       // what buckets are told apart by: the k-mer
            else { sVal[i] = t.v; return t.k; }
        },
        [&](uint64_t g) {
            if constexpr (LY::bySlot) return ((uint64_t) digitAt(g) << a.geom.headShift) | (a.keys[g] >> rx::SLOT_KEY_SHIFT);
            else return a.keys[g];
        },
        [&](int g0, int gm) {

            // word of the network: (bucket ordinal within the group, low k-mer bits, position within the group)
            const int idxBits = gm > 256 ? 9 : 8, ord0 = w.ord[g0];
            sortGroup<W>(gm, lane,
                [&](int i) {
                    const uint64_t low = LY::bySlot ? (uint64_t) ((uint32_t) (sKey[g0 + i] >> rx::SLOT_KEY_SHIFT)) & lowMask : sKey[g0 + i] & lowMask;      // (a slot tuple's k-mer bits sit above its index)
                    return (W) ((((W) (w.ord[g0 + i] - ord0) << lowBits | (W) low) << idxBits) | (W) i);
                },
                [&](auto &v) {
                    // per sorted position: window slot of the element, start of its run (= equal bucket and low bits; from an
                    // inclusive max-scan of the start positions over the wave) and whether it starts one
                    constexpr int R = sizeof(v) / sizeof(v[0]);
                    const W prevLast = shflUpW<W>(v[R - 1], 1);
                    int st[R], last = -1;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int p = lane * R + r;
                        const W prev = r ? v[r - 1] : prevLast;
                        if (p == 0 || (v[r] >> idxBits) != (prev >> idxBits)) last = p;
                        st[r] = last;
                    }
                    int sc = last;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(sc, d, 64); if (lane >= d) sc = max(sc, o); }
                    const int carry = __shfl_up(sc, 1, 64);
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int p = lane * R + r, s0 = st[r] < 0 ? carry : st[r];
                        ss[p] = (uint32_t) (g0 + (int) ((uint32_t) v[r] & ((1u << idxBits) - 1u))) | ((uint32_t) s0 << WV_IDX) | (s0 == p ? 1u << 31 : 0u);
                    }
                });
            waveLdsSync();
            // one copy of the member code in the instruction stream, whatever the size of the network before it
#pragma unroll 1
            for (int p = lane; p < gm; p += 64) {
                const uint32_t cw = ss[p];
                const int s0 = (int) ((cw >> WV_IDX) & 1023u), e = (int) (cw & IDXM), p0row = p - lane;
                unsigned long long gk = ~0ull;
                uint32_t recRepOfLane = 0;      // the run's representative (kept members only read it)
                const bool hasNext = (p + 1 < gm) && !(ss[p + 1] >> 31);
                if constexpr (LY::bySlot) {
                if (s0 != p || hasNext) {
                    // slot tuples: sequence, position and strand come straight out of the slot index (the k-mer itself is of no interest
                    // behind the sort; every sequence has the one length)
                    const int er = (int) (ss[s0] & IDXM);
                    uint32_t repId, bestPos, id, tPos; bool repFwd, fwd;
                    slotFields(a.geom, sKey[er], repId, bestPos, repFwd);
                    for (int t = s0 + 1; t < gm && !(ss[t] >> 31); t++) {     // same sequence twice in the run (rare)
                        uint32_t ie, pe; bool fe;
                        slotFields(a.geom, sKey[(int) (ss[t] & IDXM)], ie, pe, fe);
                        if (ie != repId) break;
                        if (pe < bestPos) { bestPos = pe; repFwd = fe; }
                    }
                    slotFields(a.geom, sKey[e], id, tPos, fwd);
                    const bool firstRun = r0 + (uint64_t) (g0 + s0) == a.firstRunIdx;
                    gk = groupKeyCore(a, repId, (int) a.geom.uniL, (int) bestPos, firstRun ? false : !repFwd, id, (int) a.geom.uniL, (int) tPos, !fwd);
                    if (a.wide && s0 == p) gk = markRunStart(a, gk, repId);
                    recRepOfLane = repId;
                }
                } else
                if (s0 != p || hasNext) {       // the staged range holds real tuples only (the unused slots sorted behind it)
                    const int er = (int) (ss[s0] & IDXM);
                    uint64_t bestKey = sKey[er]; const V best = sVal[er];
                    const uint32_t repId = LY::seqOf(best);
                    uint32_t repLen, bestPos;
                    LY::unpackR1(bestKey, best, a.geom, repLen, bestPos);
                    for (int t = s0 + 1; t < gm && !(ss[t] >> 31); t++) {     // same sequence twice in the run (rare)
                        const int et = (int) (ss[t] & IDXM);
                        if (LY::seqOf(sVal[et]) != repId) break;
                        uint32_t le, pe;
                        LY::unpackR1(sKey[et], sVal[et], a.geom, le, pe);
                        if (pe < bestPos) { bestPos = pe; bestKey = sKey[et]; }
                    }
                    const uint64_t key = sKey[e]; const V val = sVal[e];
                    uint32_t tLen, tPos;
                    LY::unpackR1(key, val, a.geom, tLen, tPos);
                    const bool firstRun = r0 + (uint64_t) (g0 + s0) == a.firstRunIdx;
                    gk = groupKeyCore(a, repId, (int) repLen, (int) bestPos, firstRun ? false : ((bestKey & BIT63) == 0), LY::seqOf(val), (int) tLen, (int) tPos, (key & BIT63) == 0);
                    if (a.wide && s0 == p) gk = markRunStart(a, gk, repId);
                    recRepOfLane = repId;
                }
                a.out[r0 + (uint64_t) (g0 + p)] = gk;
                const unsigned long long keptMask = __ballot(runsort::gkKept(gk));
                keptCnt += (uint32_t) __popcll(keptMask);
                if constexpr (REC) if (a.recRep) {
                    // records of this row of 64 sorted positions: a record begins at a kept key that starts a k-mer run or follows a key that
                    // is not kept, and ends in front of the next run start / not-kept key; a stretch that runs on from the row in front
                    // (the same k-mer run: rows of one group) lengthens that row's last record instead of beginning one
                    const uint32_t recBase = (uint32_t) __builtin_amdgcn_readfirstlane((int) recCount);       // (lane 0 runs every row: its values are complete)
                    const bool lastKept = __builtin_amdgcn_readfirstlane((int) prevRowKept) != 0;
                    const unsigned long long startMask = __ballot((cw >> 31) != 0u);
                    const bool runsOn = p0row != 0 && lastKept && (keptMask & 1ull) && !(startMask & 1ull) && recBase - 1u < recLimit;      // (a record on the overflow list is not lengthened: the stretch goes on as a record of its own)
                    const unsigned long long begins = keptMask & (startMask | ~(keptMask << 1)) & ~(runsOn ? 1ull : 0ull), ends = ~keptMask | startMask;
                    if (((begins >> lane) & 1ull) || (runsOn && lane == 0)) {
                        const unsigned long long behind = lane == 63 ? 0ull : ends >> (lane + 1);
                        const uint32_t len = behind ? (uint32_t) __ffsll(behind) : (uint32_t) (64 - lane);
                        const uint32_t j = recBase + (uint32_t) __popcll(begins & ((1ull << lane) - 1ull));
                        // (a wave's LDS operations take effect in program order: the add meets the record an earlier row wrote)
                        const uint64_t rv = ((r0 + (uint64_t) (g0 + p)) << runsort::RUN_CNT_BITS) | (uint64_t) len;
                        if (runsOn && lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&sRecVal[wave][recBase - 1u]), (unsigned long long) len);
                        else if (j < recLimit) { sRecRep[wave][j] = recRepOfLane; sRecVal[wave][j] = rv; }
                        else { const unsigned long long q = atomicAdd(a.ovCursor, 1ull); if (q < a.ovCap) { a.ovRep[q] = recRepOfLane; a.ovVal[q] = rv; } }
                    }
                    recCount = recBase + (uint32_t) __popcll(begins);
                    prevRowKept = (uint32_t) (keptMask >> 63);
                }
            }
            waveLdsSync();      // ss is reused by the next group
        });
    waveGroupStats(a.stat, keptCnt);
    recCount = (uint32_t) __builtin_amdgcn_readfirstlane((int) recCount);
    if constexpr (REC) if (a.recRep && recCount) {
        // the wave's records go out in one piece
        const uint32_t m = min(recCount, recLimit);
        waveLdsSync();
        if ((uint32_t) lane < m) { const uint64_t slot = waveIdx * (uint64_t) REC_CAP + (uint32_t) lane; a.recRep[slot] = sRecRep[wave][lane]; a.recVal[slot] = sRecVal[wave][lane]; }
        if (lane == 0) a.recCnt[waveIdx] = (uint8_t) m;
    }
}

// tiles of the vote kernels: 4096 keys (256 threads x 16 consecutive items)
// (2048 keys: the place kernel runs on blocks in flight - 35 KB of LDS per block gave 4 per CU and 19 ms)
#ifndef CDM_CP_ITEMS
#define CDM_CP_ITEMS 8
#endif
constexpr int CP_ITEMS = CDM_CP_ITEMS, CP_TILE = 256 * CP_ITEMS;
static_assert(CP_ITEMS == 8 || CP_ITEMS == 16, "vote tile: the start bits of a thread are one byte or one 16-bit word");
template <int N> struct BitsOf { typedef uint16_t T; };
template <> struct BitsOf<8> { typedef uint8_t T; };
typedef BitsOf<CP_ITEMS>::T CpBits;
// LDS index with one pad slot per 16 items: thread t walks items 16t..16t+15 without bank conflicts
__device__ __forceinline__ int padIdx(int i) { return i + (i >> 4); }
constexpr int CP_LDS = CP_TILE + CP_TILE / 16 + 1;

// ------------------------------------------------------------------------------------------------ K4: vote
struct VoteArgs {
    const uint64_t *keys;   // sorted (rep, id, diag), strand in bit 0
    uint64_t n;
    uint32_t idBits, diagBits; int diagBias;
    unsigned long long *perRep;  // [nSeq] number of hits per representative
    // The reference's per-target scan does not stop at the end of the sorted group tuples: it runs on into the tuples that
    // assignGroup's in-place compaction left behind (kmermatcher.cpp:875-887 reads hashSeqPair[kmerPos + kmerOffset].id up to
    // the end of the array), i.e. the k-mer-ordered tuples from index nGroup on, while their sequence id equals the target.
    // stale[0] = number of such tuples, stale[1] = their sequence id, stale[2..] their positions (k_stale_tail); their k-mer
    // field is UINT64_MAX by then, so they count as forward.
    const uint32_t *stale;
    // Multi-GPU runs (every rank votes on the representatives it owns): what the scan of this rank's LAST segment runs into is
    // the head of the next rank's sorted array - cont[0] entries (cont[3 + j] = biased diagonal | "reverse" << 31) that apply if
    // the target is cont[1]; only if cont[2] is set does the scan go on into the left-over tuples (`stale`) after them.  NULL on
    // a single device.
    const uint32_t *cont;
};
constexpr int STALE_MAX = CDM_STALE_MAX, CONT_CAP = 2048;
// a (rep, target != rep) segment starts at i
__device__ __forceinline__ bool validStart(const VoteArgs &a, uint64_t i, uint32_t &rep, uint32_t &target) {
    const uint64_t seg = a.keys[i] >> (a.diagBits + 1);
    if (i > 0 && (a.keys[i - 1] >> (a.diagBits + 1)) == seg) return false;
    target = (uint32_t) (seg & ((1ull << a.idBits) - 1)); rep = (uint32_t) (seg >> a.idBits);
    return target != rep;   // self tuples give no hit (:898-903)
}
// tiles of 4096 tuples: number of hit-producing segment starts per tile and per representative (coalesced, order free)
__global__ __launch_bounds__(256) void k_seg_count(VoteArgs a, unsigned long long *__restrict__ tileCnt) {
    const uint64_t base = (uint64_t) blockIdx.x * CP_TILE;
    unsigned int c = 0;
#pragma unroll
    for (int j = 0; j < CP_ITEMS; j++) {
        const uint64_t i = base + threadIdx.x + 256 * j;
        uint32_t rep, target;
        if (i < a.n && validStart(a, i, rep, target)) { c++; atomicAdd(&a.perRep[rep], 1ull); }
    }
    const unsigned int tot = cdm_block_sum<unsigned int>(c);
    if (threadIdx.x == 0) tileCnt[blockIdx.x] = tot;
}
// vote of the segment starting at tile-local index li, reading the tile from LDS and whatever lies beyond it from memory
__device__ __forceinline__ HitRec voteSegmentTile(const VoteArgs &a, const uint64_t *sKeys, uint64_t base, int li, uint32_t target) {
    const uint64_t idMask = (1ull << a.idBits) - 1, diagMask = (1ull << a.diagBits) - 1;
    const uint64_t key = sKeys[padIdx(li)];
    uint32_t prevDiag = (uint32_t) ((key >> 1) & diagMask), diagonal = prevDiag;
    uint32_t maxDiag = 0, diagCnt = 0, top = 0; int bestRev = (key & 1ull) ? 0 : 1;
    // two separate loops so that the common in-tile walk issues LDS reads only
    const int tileEnd = (int) min((uint64_t) CP_TILE, a.n - base);
    int i = li; bool done = false;
    for (; i < tileEnd; i++) {
        const uint64_t k2 = sKeys[padIdx(i)];
        if ((uint32_t) ((k2 >> (a.diagBits + 1)) & idMask) != target) { done = true; break; }
        const uint32_t d = (uint32_t) ((k2 >> 1) & diagMask);
        if (prevDiag == d) diagCnt++; else diagCnt = 1;
        if (diagCnt >= maxDiag) { diagonal = d; maxDiag = diagCnt; bestRev = (k2 & 1ull) ? 0 : 1; }
        prevDiag = d; top++;
    }
    if (!done) {
        for (uint64_t kk = base + CP_TILE; kk < a.n; kk++) {
            const uint64_t k2 = a.keys[kk];
            if ((uint32_t) ((k2 >> (a.diagBits + 1)) & idMask) != target) { done = true; break; }
            const uint32_t d = (uint32_t) ((k2 >> 1) & diagMask);
            if (prevDiag == d) diagCnt++; else diagCnt = 1;
            if (diagCnt >= maxDiag) { diagonal = d; maxDiag = diagCnt; bestRev = (k2 & 1ull) ? 0 : 1; }
            prevDiag = d; top++;
        }
    }
    bool intoStale = !done;
    if (!done && a.cont) {                      // the scan reached the end of this rank's group tuples: on into the next ranks'
        if (target == a.cont[1]) {
            const uint32_t m = a.cont[0];
            for (uint32_t j = 0; j < m; j++) {
                const uint32_t e = a.cont[3 + j], d = e & 0x7FFFFFFFu;
                if (prevDiag == d) diagCnt++; else diagCnt = 1;
                if (diagCnt >= maxDiag) { diagonal = d; maxDiag = diagCnt; bestRev = (int) (e >> 31); }
                prevDiag = d; top++;
            }
        }
        intoStale = a.cont[2] != 0u;
    }
    if (intoStale && target == a.stale[1]) {    // the scan reached the end of all group tuples: on into the left-over ones
        const uint32_t m = a.stale[0];
        for (uint32_t j = 0; j < m; j++) {
            const uint32_t d = a.stale[2 + j] + (uint32_t) a.diagBias;
            if (prevDiag == d) diagCnt++; else diagCnt = 1;
            if (diagCnt >= maxDiag) { diagonal = d; maxDiag = diagCnt; bestRev = 0; }
            prevDiag = d; top++;
        }
    }
    HitRec h;
    h.target = target;
    h.score = bestRev ? -(int) top : (int) top;
    h.diagonal = (int) (short) ((int) diagonal - a.diagBias);
    return h;
}
__global__ __launch_bounds__(256) void k_seg_place(VoteArgs a, const unsigned long long *__restrict__ tileOff, const unsigned long long *__restrict__ perRepScan,
                                                   const uint64_t *__restrict__ off, HitRec *__restrict__ out) {
    __shared__ uint64_t sKeys[CP_LDS];
    __shared__ uint64_t sPrev;
    __shared__ __align__(8) CpBits sFirst[256 + 32 / sizeof(CpBits)];       // "starts a (rep, target) segment" bits, CP_ITEMS per thread = one bit array
    const uint64_t base = (uint64_t) blockIdx.x * CP_TILE;
#pragma unroll
    for (int j = 0; j < CP_ITEMS; j++) { const int li = threadIdx.x + 256 * j; const uint64_t i = base + li; sKeys[padIdx(li)] = (i < a.n) ? a.keys[i] : ~0ull; }
    if (threadIdx.x == 0) sPrev = base ? a.keys[base - 1] : ~0ull;
    if (threadIdx.x < 32 / sizeof(CpBits)) sFirst[256 + threadIdx.x] = 0;
    __syncthreads();
    const int shift = a.diagBits + 1;
    const uint64_t idMask = (1ull << a.idBits) - 1, diagMask = (1ull << a.diagBits) - 1;
    const int tileEnd = (int) min((uint64_t) CP_TILE, a.n - base);
    unsigned int c = 0, mask = 0, firstBits = 0;
#pragma unroll
    for (int j = 0; j < CP_ITEMS; j++) {
        const int li = threadIdx.x * CP_ITEMS + j;
        if (li >= tileEnd) break;
        const uint64_t seg = sKeys[padIdx(li)] >> shift;
        const uint64_t prevSeg = ((li == 0) ? sPrev : sKeys[padIdx(li - 1)]) >> shift;
        const bool first = (base + li == 0) || prevSeg != seg;
        if (first) firstBits |= 1u << j;
        if (first && (uint32_t) (seg & idMask) != (uint32_t) (seg >> a.idBits)) { c++; mask |= 1u << j; }
    }
    sFirst[threadIdx.x] = (CpBits) firstBits;
    unsigned int pre, totC;
    pre = cdm_block_excl_sum<unsigned int>(c, totC);       // (its barriers also publish sFirst)
    unsigned long long rank = tileOff[blockIdx.x] + pre;   // number of hit-producing segments before this one, whole array
    const unsigned long long *firstWords = reinterpret_cast<const unsigned long long *>(sFirst);
#pragma unroll 1
    while (mask) {   // one copy of the walk in the instruction stream (an unrolled x16 body thrashes the instruction cache)
        const int j = __ffs(mask) - 1;
        mask &= mask - 1;
        const int li = threadIdx.x * CP_ITEMS + j;
        const uint64_t k0 = sKeys[padIdx(li)], seg = k0 >> shift;
        const uint32_t target = (uint32_t) (seg & idMask), rep = (uint32_t) (seg >> a.idBits);
        // Most segments are one run of one diagonal that ends where the next segment starts: the next start is the next set bit
        // of the bit array; if the tuple there has another target id (no run-on into the next representative) and the first
        // and last tuples of the segment agree on the diagonal (they are sorted by it), the vote is known without a walk.
        HitRec h; bool quick = false;
        {
            int e = -1;
            for (int w = (li + 1) >> 6; w < CP_TILE / 64 && e < 0; w++) {
                unsigned long long m = firstWords[w];
                if (w == ((li + 1) >> 6)) m &= ~0ull << ((li + 1) & 63);
                if (m) e = w * 64 + __ffsll(m) - 1;
            }
            if (e > 0 && e < tileEnd) {
                const uint64_t kn = sKeys[padIdx(e)], kl = sKeys[padIdx(e - 1)];
                if ((uint32_t) ((kn >> shift) & idMask) != target && ((k0 >> 1) & diagMask) == ((kl >> 1) & diagMask)) {
                    h.target = target;
                    h.score = (kl & 1ull) ? (e - li) : -(e - li);
                    h.diagonal = (int) (short) ((int) ((k0 >> 1) & diagMask) - a.diagBias);
                    quick = true;
                }
            }
        }
        if (!quick) h = voteSegmentTile(a, sKeys, base, li, target);
        out[off[rep] + 1 + (rank - perRepScan[rep])] = h;
        rank++;
    }
}
// The head of a sorted group-key array: the tuples from its start on while they have the target id of the first one, whatever
// their representative (what a scan coming in from the rank in front runs through, kmermatcher.cpp:875-887).
// out[0] = count (CONT_CAP + 1: longer than the list), out[1] = that id, out[2] = 1 if the head is the whole array, out[3..] entries
__global__ void k_head_segment(const uint64_t *__restrict__ keys, uint64_t n, uint32_t idBits, uint32_t diagBits, uint32_t *__restrict__ out) {
    const uint64_t idMask = (1ull << idBits) - 1ull, diagMask = (1ull << diagBits) - 1ull;
    if (n == 0) { out[0] = 0; out[1] = 0; out[2] = 1; return; }
    const uint32_t id = (uint32_t) ((keys[0] >> (diagBits + 1)) & idMask);
    uint64_t c = 0;
    for (; c < n && c <= (uint64_t) CONT_CAP; c++) {
        const uint64_t k2 = keys[c];
        if ((uint32_t) ((k2 >> (diagBits + 1)) & idMask) != id) break;
        if (c < (uint64_t) CONT_CAP) out[3 + c] = (uint32_t) ((k2 >> 1) & diagMask) | ((k2 & 1ull) ? 0u : 1u << 31);
    }
    out[0] = (uint32_t) c; out[1] = id; out[2] = (c == n) ? 1u : 0u;
}
__global__ void k_offsets(const unsigned long long *__restrict__ perRepScan, uint32_t n, uint64_t *__restrict__ off) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q <= n) off[q] = (uint64_t) q + perRepScan[q];        // one self hit per sequence in front of its own hits
}
__global__ void k_self(const uint64_t *__restrict__ off, uint32_t n, HitRec *__restrict__ out) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    HitRec h; h.target = q; h.score = 0; h.diagonal = 0;
    out[off[q]] = h;
}
__global__ void k_len_keys(const uint32_t *__restrict__ len, uint32_t n, uint32_t maxLen, uint32_t *__restrict__ key, uint32_t *__restrict__ val) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { key[i] = maxLen - len[i]; val[i] = i; }   // ascending key = descending length; stable sort keeps ids ascending
}
// slots of the r-th sequence in (length desc, id asc) order
__global__ void k_slot_counts(const uint32_t *__restrict__ len, const uint32_t *__restrict__ order, uint32_t n, int k, unsigned long long *__restrict__ slots) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    if (r == n) { slots[r] = 0; return; }
    const uint32_t L = len[order[r]];
    slots[r] = 1ull + ((L >= (uint32_t) k) ? (L - k + 1) : 0);
}
__global__ void k_slot_scatter(const uint32_t *__restrict__ order, const unsigned long long *__restrict__ ordOff, uint32_t n, uint64_t *__restrict__ slotOff,
                               uint32_t *__restrict__ rankOf) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) { slotOff[order[r]] = ordOff[r]; rankOf[order[r]] = r; }
    if (r == n) slotOff[n] = ordOff[n];
}
// split by reads: only the sequences with order ranks [lo, hi) get slots
__global__ void k_slot_mask(unsigned long long *__restrict__ slots, uint32_t n, uint32_t lo, uint32_t hi) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && (r < lo || r >= hi)) slots[r] = 0;
}
// Split by reads: the blocks of the slot order hold about the same number of SLOTS each (not of sequences: the order starts with the
// longest ones, and in a contig phase a tenth of the sequences holds nine tenths of the letters).  prefix = exclusive sums of the slot
// counts in that order, prefix[n] = all slots; block `blk` of `of` = the order ranks [out[0], out[1]): first rank whose prefix reaches
// total * blk / of - the same arithmetic on every rank, so the blocks tile the order.
__global__ void k_block_cuts(const unsigned long long *__restrict__ prefix, uint32_t n, uint32_t blk, uint32_t of, uint32_t *__restrict__ out) {
    const unsigned long long total = prefix[n];
    for (int side = 0; side < 2; side++) {
        const uint32_t b = blk + (uint32_t) side;
        uint32_t res = n;
        if (b < of) {
            const unsigned long long want = (unsigned long long) (((unsigned __int128) total * b) / of);
            uint32_t lo = 0, hi = n;
            while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (prefix[mid] < want) lo = mid + 1; else hi = mid; }
            res = lo;
        }
        out[side] = res;
    }
}
// first index of the keys (ordered by the `slices`-valued field at bit `shift`) whose field is >= p, for p = 0 .. slices
__global__ void k_slice_bounds(const uint64_t *__restrict__ keys, uint64_t m, int shift, uint32_t slices, unsigned long long *__restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > slices) return;
    if (p == slices) { out[p] = m; return; }
    uint64_t lo = 0, hi = m;
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (((keys[mid] >> shift) & (uint64_t) (slices - 1)) < p) lo = mid + 1; else hi = mid; }
    out[p] = lo;
}
__global__ __launch_bounds__(256) void k_reduce_stats(const unsigned long long *__restrict__ stripes, unsigned long long *__restrict__ out) {
    unsigned long long c = 0;
    for (int i = threadIdx.x; i < STAT_STRIPES; i += 256) c += stripes[i];
    c = cdm_block_sum<unsigned long long>(c);
    if (threadIdx.x == 0) out[0] = c;
}
// number of keys in front of the unused / dropped ones (key ~0) once the array is sorted on bits up to `bit`, which is set
// only in them: the first key with that bit set
__global__ void k_live_count(const uint64_t *__restrict__ keys, uint64_t n, int kbits, unsigned long long *__restrict__ out) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if ((keys[mid] >> kbits) & 1ull) hi = mid; else lo = mid + 1; }
    *out = lo;
}
// number of real tuples in region 2 (sorted on the low 63 bits; the empty slots, key ~0, are last)
__global__ void k_count_hash_tuples(const uint64_t *__restrict__ keys, uint64_t n, unsigned long long *__restrict__ out) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (keys[mid] == ~0ull) hi = mid; else lo = mid + 1; }
    *out = lo;
}
// The tuples the reference's last per-target scan runs into (see VoteArgs): k-mer-ordered real tuples from index J = nGroup on
// while their sequence id is `target`.  Region 1 is only sorted on its high bits in memory (unless `sorted`): the bucket that
// holds index J is ranked here (all pairs, one block); buckets larger than STALE_BUCKET were finished through the big-bucket
// path, which leaves them sorted in memory.  Region 2 is sorted.
constexpr int STALE_BUCKET = 2048;
template <typename LY>
struct StaleArgs {
    const uint64_t *keys; const typename LY::V *vals; TupleGeom geom;
    uint64_t live, kmerSlots, nTuples, J; int lowBits; bool sorted;
    uint32_t *out;      // [0] count, [1] their sequence id, [2..] positions
};
template <typename LY>
__global__ __launch_bounds__(256) void k_stale_tail(StaleArgs<LY> a) {
    __shared__ uint64_t sC[STALE_BUCKET];
    __shared__ uint64_t sB[2];
    __shared__ int sSel;
    const uint64_t hmask = (2ull << a.geom.kbits) - 1ull, lowMask = (1ull << a.lowBits) - 1ull;
    uint32_t cnt = 0, target = ~0u;
    for (uint64_t j = a.J; cnt < (uint32_t) STALE_MAX; j++) {
        uint64_t idx;
        if (j < a.live) {
            idx = j;
            if (!a.sorted) {
                if (threadIdx.x == 0) {     // bucket of j: equal high bits
                    const uint64_t h = (memSortBits<LY>(a.keys, j, a.geom) & hmask) >> a.lowBits;
                    uint64_t lo = 0, hi = j;
                    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (((memSortBits<LY>(a.keys, mid, a.geom) & hmask) >> a.lowBits) < h) lo = mid + 1; else hi = mid; }
                    sB[0] = lo;
                    lo = j; hi = a.live;
                    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (((memSortBits<LY>(a.keys, mid, a.geom) & hmask) >> a.lowBits) == h) lo = mid + 1; else hi = mid; }
                    sB[1] = lo;
                }
                __syncthreads();
                const uint64_t b0 = sB[0], b1 = sB[1];
                const int m = (int) min((uint64_t) STALE_BUCKET + 1, b1 - b0);
                if (m <= STALE_BUCKET) {
                    for (int i = threadIdx.x; i < m; i += blockDim.x) sC[i] = ((memSortBits<LY>(a.keys, b0 + i, a.geom) & lowMask) << 12) | (uint64_t) i;
                    if (threadIdx.x == 0) sSel = 0;
                    __syncthreads();
                    const int want = (int) (j - b0);
                    for (int e = threadIdx.x; e < m; e += blockDim.x) {
                        const uint64_t mine = sC[e]; int r = 0;
                        for (int f = 0; f < m; f++) r += sC[f] < mine;
                        if (r == want) sSel = e;
                    }
                    __syncthreads();
                    idx = b0 + (uint64_t) sSel;
                }
                __syncthreads();
            }
        } else {
            idx = a.kmerSlots + (j - a.live);
            if (idx >= a.nTuples) break;
        }
        if (a.keys[idx] == ~0ull) break;                           // end of the real tuples (the empty slots of region 2; region 1 is read below `live` only)
        uint64_t key; typename LY::V v;
        memPair<LY>(a.keys, a.vals, idx, a.geom, key, v);
        if (cnt == 0) target = LY::seqOf(v);                       // the scan can only run on for this sequence id
        else if (LY::seqOf(v) != target) break;
        if (threadIdx.x == 0) a.out[2 + cnt] = LY::posOf(key, v, idx, a.geom);
        cnt++;
    }
    if (threadIdx.x == 0) { a.out[0] = cnt; a.out[1] = target; }
}
// smallest index at which the two arrays differ (atomicMin; *out starts as ~0)
__global__ void k_first_diff(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b, uint64_t n, unsigned long long *__restrict__ out) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        if (a[i] != b[i]) { atomicMin(out, (unsigned long long) i); return; }
}
// LayoutSlot, buckets the grouping kernel left alone: their slot tuples as (key, id) pairs in the dense staging arrays, and the sorted
// pairs back as slot tuples (a wave per listed range (start, end, offset), as bucket::k_big_copy)
template <bool GATHER>
__global__ __launch_bounds__(256) void k_big_slot_pairs(const unsigned long long *__restrict__ ranges, unsigned int cnt, uint64_t *arr, TupleGeom geom, uint64_t *denseK, uint32_t *denseV) {
    const unsigned int lane = threadIdx.x & 63, wavesPerGrid = gridDim.x * 4;
    for (unsigned int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < cnt; r += wavesPerGrid) {
        const unsigned long long s = ranges[3 * (size_t) r], e = ranges[3 * (size_t) r + 1], o = ranges[3 * (size_t) r + 2];
        const uint32_t td = headDigit(geom, s);         // (a bucket lies inside one segment)
        for (unsigned long long i = lane; i < e - s; i += 64) {
            if (GATHER) { uint64_t key; uint32_t id; slotTupleToPair(geom, arr[s + i], td, key, id); denseK[o + i] = key; denseV[o + i] = id; }
            else arr[s + i] = pairToSlotTuple(geom, denseK[o + i], denseV[o + i]);
        }
    }
}
inline uint32_t bitsFor(uint64_t v) { uint32_t b = 1; while ((1ull << b) < v) b++; return b; }

// One kmermatcher run, in phases so that a multi-GPU run can exchange between them (shard.py / cdm_kmermatch_part):
//   phaseA     extraction (of this rank's k-mer range), sort 1, grouping -> group keys in k-mer order (startIo), live, nKept
//   staleTail  the left-over tuples behind global k-mer-order index J (the reference's run-past-the-end scan)
//   phaseB     sort 2 + vote -> prefilter hits
// LayoutSlot serves a DB whose sequences all have one length (Σ lengths = n x longest), of at least k letters, with fewer than 2^32 k-mer
// slots, a k-mer of 14 .. 20 letters (the tuple keeps 31 k-mer bits behind the 9-bit head digit) and lengths LayoutPacked's key holds
inline uint32_t slotsPerSeq(uint32_t L, int k) { return L >= (uint32_t) k ? L - (uint32_t) k + 2u : 1u; }
inline bool slotLayoutFits(const cdm_seqdb *db, int k) {
    if (db->n == 0 || db->residues != db->n * (uint64_t) db->maxLen || db->maxLen < (uint32_t) k) return false;
    if (2 * k + 1 <= 27 || 2 * k - rx::BITS > rx::SLOT_REM) return false;       // (k of 14 .. 20 letters: low bits left to the grouping kernel, at most 31 behind the head digit)
    if (2 * k + 1 + 2 * (int) bitsFor((uint64_t) db->maxLen + 1) > 63) return false;
    return db->n * (uint64_t) slotsPerSeq(db->maxLen, k) < (1ull << 32);
}
struct KmerJobBase {
    virtual ~KmerJobBase() {}
    virtual int phaseA() = 0;
    virtual int staleTail(unsigned long long J, bool fromStart) = 0;
    virtual int phaseB(cdm_hits **out) = 0;
    virtual int gatherByRep() = 0;
    virtual int sortFrom(const uint64_t *devKeys, uint64_t nKeys, uint32_t *head, uint64_t info[2]) = 0;
    virtual int voteWith(const uint32_t *cont, const uint32_t *staleIn, cdm_hits **out) = 0;
    // the split by reads (cdm_kmermatch_split_*): splitBegin = extraction of the owned sequences + the tuples ordered by destination
    // range; splitFinish = sort 1 + grouping on what arrived (region 1: m tuples, region 2: h whole-sequence hash tuples)
    virtual int splitBegin() = 0;
    virtual int splitFinish(const void *keys, const void *vals, uint64_t m, const void *hkeys, const void *hvals, uint64_t h, bool below) = 0;
    bool split = false;
    std::vector<unsigned long long> sendOff;        // [nparts + 1] tuples per destination range, prefix sums
    const void *sendKeys = nullptr, *sendVals = nullptr, *sendHashKeys = nullptr, *sendHashVals = nullptr; unsigned long long sendHash = 0; int valBytes = 0;
    cdm_ctx *ctx = nullptr; const cdm_seqdb *db = nullptr; cdm_kmer_params parCopy; const cdm_kmer_params *par = nullptr;
    int part = 0, nparts = 1;           // this rank's k-mer range (nparts == 1: everything)
    int block = 0, nBlocks = 0;         // splitBegin: block `block` of `nBlocks` of the sequences (0: block `part` of `nparts`), the tuples ordered by nparts ranges
    unsigned long long live = 0, nKept = 0, regionTwo = 0;      // real tuples of region 1 in this range; kept group tuples; real tuples of region 2
    bool anyBelow = false;              // a real tuple with a k-mer below this range exists (then the array's very first run is not here)
    uint64_t *gathered = nullptr;       // gatherByRep: the kept group keys grouped by representative, k-mer order inside (device)
    uint32_t staleHost[64 + 3] = {0};   // staleTail's result: [0] count, [1] sequence id, [2..] positions, [66] = 1 if the scan reached the end of this range's tuples
};
template <typename LY>
struct KmerJob : KmerJobBase {
    typedef typename LY::V V;
    hipStream_t s = nullptr; uint32_t n = 0; int k = 0;
    uint32_t idBits = 0, diagBits = 0; int diagBias = 0; const char *sortEnv = nullptr; bool lsdOnly = false;
    bool wide = false;                        // group keys without the representative (runsort.h RunArgs; packGroupKey)
    bool passes = false;                      // a range of the k-mer-range passes on one device (kmermatchPassesT): no exchange of group keys follows
    uint64_t r2Slots = 0;                     // size of region 2: n (one whole-sequence hash slot per sequence), or what arrived (split by reads)
    uint32_t ordLo = 0, ordHi = 0;            // split by reads: the order ranks of the sequences this rank extracts
    DevBuf<uint64_t> splitK; DevBuf<V> splitV;           // keepOnlyOutgoing: the ordered tuples in buffers of their own size
    DevBuf<unsigned long long> counters;      // scratch counters ([2] = number of kept group tuples)
    DevBuf<unsigned int> cls;                 // slow-path list sizes
    DevBuf<uint32_t> listShort, listLong, listSingle, listHuge;
    DevBuf<unsigned long long> slots; DevBuf<uint64_t> slotOff; DevBuf<uint32_t> rankOf;
    uint64_t kmerSlots = 0; unsigned long long nTuples = 0;
    DoubleBuf<uint64_t> keys; DoubleBuf<V> vals;
    DevBuf<uint64_t> k0, k1; DevBuf<V> v0, v1;
    V *vA = nullptr, *vB = nullptr;           // v0 / v1 as the kernels index them (LayoutSlot: values exist for region 2 only, the pointers stand kmerSlots entries in front of them)
    DevBuf<unsigned long long> segBuf;        // LayoutSlot: where every head digit's slot tuples start (rx::sortSlotKeys)
    TupleGeom geom; int lowBits = 0;
    DevBuf<unsigned long long> headHist; bool headCounted = false;        // LayoutSlot: head digit counts taken by the extraction kernels
    // run records staged by the grouping kernel (BucketGroupArgs::recRep): valid for region 1 when stagedWaves != 0
    DevBuf<uint32_t> stRep, ovRep; DevBuf<uint64_t> stVal, ovVal; DevBuf<uint8_t> stCnt; DevBuf<unsigned long long> recFlag; uint64_t stagedWaves = 0, stagedOwn = 0; unsigned long long ovCap = 0, nOvRec = 0;
    DevBuf<uint32_t> bigRecRep, bigRecRep1; DevBuf<uint64_t> bigRecVal, bigRecVal1; unsigned long long nBigRec = 0;        // ... and the records of the buckets that kernel left to the caller
    bool ownPipeline = false;                 // sortAndGroup runs for the single-device call that also runs sort 2 on its own buffers (phaseA + phaseB)
    // slot layout on a rank (cdm_kmermatch_part for a DB of one read length): the rank extracts every read, the head histogram the
    // extraction counts cuts the 512 head digits into nparts ranges of equal tuple counts (the same cuts on every rank), and the head
    // pass keeps this rank's digits only - what it drops costs a read, not a write
    bool headRange = false; uint32_t headLo = 0, headHi = 0xFFFFFFFFu;
    GroupArgs<LY> ga; DevBuf<unsigned long long> statStripes; unsigned long long *startIo = nullptr; DevBuf<uint32_t> staleBuf;
    DevBuf<uint64_t> runsOut, runsTmp;       // sort 2 "check" mode: the run-based result next to the radix one
    DevBuf<uint64_t> recvA, recvB; DevBuf<uint32_t> contBuf;      // multi-GPU second half: received keys / their sorted form, the continuation list
    const uint64_t *sorted2M = nullptr; unsigned long long nGroupM = 0;
    // the aggregated form of sort 2's result (aggvote.h): entries per representative segment; set when sort2 took that way
    bool haveEntries = false; uint64_t nSegM = 0;
    DevBuf<uint32_t> agSegOfRec, agSegRep, agEntCnt, agPending; DevBuf<unsigned long long> agSegFirstRec, agEntOff, agPerRep, agCursor; DevBuf<aggv::Ent> agEnt; DevBuf<unsigned int> agFlags;
    float msSort1 = 0;
    KmerJob(cdm_ctx *c, const cdm_seqdb *d, const cdm_kmer_params *p) { ctx = c; db = d; parCopy = *p; par = &parCopy; }
// the members every phase reads: stream, sizes, the form of the group key, the scratch counters
int init() {
    s = ctx->stream;
    n = (uint32_t) db->n;
    k = par->kmer_size;
    if (k < 4 || k > 31) { cdm_set_error("cdm_kmermatch: k must be in 4..31 (got %d)", k); return CDM_ERR_INVALID; }
    idBits = bitsFor(n); diagBits = bitsFor(2ull * db->maxLen + 2);
    // (rep, id, diagonal, strand) in one word while it fits 63 bits - 2 M sequences with contigs of 500 k letters, 50 M reads of 2 k
    // letters; beyond that (25 M sequences with contigs: BASELINE config 5) the representative leaves the key (the wide form:
    // runsort.h RunArgs) and sort 2 + vote run on aggregated entries only.  CDM_FORCE_WIDE_KEY=1: the wide form for any DB (tests).
    wide = 2 * idBits + diagBits + 1 > 63 || cdmGetenv("CDM_FORCE_WIDE_KEY") != nullptr;
    if (wide && (int) (aggv::AG_ORD + idBits + diagBits) > 64) {
        cdm_set_error("cdm_kmermatch: %u sequences x max length %u: ids and diagonals beyond %d bits are not implemented", n, db->maxLen, 64 - aggv::AG_ORD); return CDM_ERR_UNSUPPORTED;
    }
    if (wide && nparts > 1 && !passes) { cdm_set_error("cdm_kmermatch_part: %u sequences x max length %u need the wide group key, which the exchange of the k-mer-range split does not carry yet", n, db->maxLen); return CDM_ERR_UNSUPPORTED; }
    diagBias = (int) db->maxLen + 1;
    sortEnv = cdmGetenv("CDM_KMER_SORT");
    lsdOnly = sortEnv && !strcmp(sortEnv, "lsd");
    if (!counters.alloc(8)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    hipMemsetAsync(counters.p, 0, 8 * 8, s);
    geom.kbits = 2 * k; geom.lb = (int) bitsFor((uint64_t) db->maxLen + 1); geom.lenArr = db->len;
    if constexpr (LY::bySlot) {
        if (!slotLayoutFits(db, k) || split || (nparts != 1 && !headRange) || passes || lsdOnly) { cdm_set_error("cdm_kmermatch: internal error: the slot layout was chosen for a run it does not serve"); return CDM_ERR_INVALID; }
        geom.uniL = db->maxLen; geom.uniK = k; geom.uniS = slotsPerSeq(db->maxLen, k); divMagic(geom.uniS, geom.uniMul, geom.uniSh); geom.headShift = 2 * k - std::min(rx::BITS, 2 * k);
    }
    return CDM_OK;
}
int phaseA() override {
    if (int rc = init()) return rc;
    constexpr uint32_t SHORT_CAP = 256, LONG_CAP = 4096;

    if (!cls.alloc(8) || !listShort.alloc(n) || !listLong.alloc(n) || !listSingle.alloc(n) || !listHuge.alloc(n)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    if (!LY::bySlot && (!slots.alloc((size_t) n + 1) || !slotOff.alloc((size_t) n + 1) || !rankOf.alloc(n))) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    hipMemsetAsync(cls.p, 0, 8 * 4, s);
    // One slot per k-mer position + one for the whole-sequence tuple, at a fixed offset per sequence (no global counter).
    // Slots are laid out in (sequence length descending, id ascending) order: after the stable k-mer sort the first tuple
    // of every run is then the representative.
    uint64_t capacity = 0;
    if constexpr (LY::bySlot) capacity = (uint64_t) n * geom.uniS;       // one length: that order is the id order, sequence i has the slots from i x uniS on
    else {
        DevBuf<uint32_t> lk0, lk1, lv0, lv1; DevBuf<unsigned long long> ordOff;
        if (!lk0.alloc(n) || !lk1.alloc(n) || !lv0.alloc(n) || !lv1.alloc(n) || !ordOff.alloc((size_t) n + 1)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
        hipLaunchKernelGGL(k_len_keys, dim3((n + 255) / 256), dim3(256), 0, s, db->len, n, db->maxLen, lk0.p, lv0.p);
        const unsigned lenBits = bitsFor((uint64_t) db->maxLen + 2);
        cdmscan::ScanTemp st;
        bool lenFirst = true;
        if (int rc = rx::sortPairs<uint32_t, uint32_t>(s, ctx->cuCount, lk0.p, lk1.p, lv0.p, lv1.p, (uint64_t) n, 0, (int) lenBits, lenFirst)) return rc;
        DoubleBuf<uint32_t> lk(lenFirst ? lk0.p : lk1.p, lenFirst ? lk1.p : lk0.p), lv(lenFirst ? lv0.p : lv1.p, lenFirst ? lv1.p : lv0.p);
        hipLaunchKernelGGL(k_slot_counts, dim3((n + 256) / 256), dim3(256), 0, s, db->len, lv.current(), n, k, slots.p);
        if (split) {
            const unsigned blk = nBlocks ? (unsigned) block : (unsigned) part, of = nBlocks ? (unsigned) nBlocks : (unsigned) nparts;      // (the passes on one device cut the sequences into more blocks than the k-mers into ranges)
            DevBuf<uint32_t> cuts; uint32_t hc[2] = {0, 0};
            if (!cuts.alloc(2)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
            cdmscan::ScanTemp stCut;
            if (int rc = cdmscan::exclusiveScan<unsigned long long>(s, stCut, slots.p, ordOff.p, (size_t) n + 1)) return rc;
            hipLaunchKernelGGL(k_block_cuts, dim3(1), dim3(1), 0, s, (const unsigned long long *) ordOff.p, n, blk, of, cuts.p);
            hipMemcpyAsync(hc, cuts.p, 8, hipMemcpyDeviceToHost, s);
            if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: slot layout failed"); return CDM_ERR_HIP; }
            ordLo = hc[0]; ordHi = hc[1];
            hipLaunchKernelGGL(k_slot_mask, dim3((n + 255) / 256), dim3(256), 0, s, slots.p, n, ordLo, ordHi);
        }
        if (int rc = cdmscan::exclusiveScan<unsigned long long>(s, st, slots.p, ordOff.p, (size_t) n + 1)) return rc;
        hipLaunchKernelGGL(k_slot_scatter, dim3((n + 256) / 256), dim3(256), 0, s, lv.current(), ordOff.p, n, slotOff.p, rankOf.p);
        hipMemcpyAsync(&capacity, ordOff.p + n, 8, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: slot layout failed"); return CDM_ERR_HIP; }
    }
    kmerSlots = capacity;                            // region 1: k-mer slots (+ slot 0 per sequence)
    r2Slots = split ? (uint64_t) (ordHi - ordLo) : (uint64_t) n;
    capacity += r2Slots;                             // region 2: whole-sequence hash tuples
    nTuples = capacity;

    const uint64_t valSlots = LY::bySlot ? r2Slots : capacity;      // (LayoutSlot: only the whole-sequence hash tuples carry a value)
    if (!k0.alloc(capacity) || !k1.alloc(capacity) || !v0.alloc(valSlots) || !v1.alloc(valSlots)) {
        cdm_set_error("cdm_kmermatch: out of device memory for %llu k-mer tuples (%.1f GB)", (unsigned long long) capacity, (capacity * 16.0 + valSlots * 2.0 * sizeof(V)) / 1e9); return CDM_ERR_HIP;
    }
    vA = LY::bySlot ? v0.p - kmerSlots : v0.p; vB = LY::bySlot ? v1.p - kmerSlots : v1.p;
    geom.kbits = 2 * k; geom.lb = (int) bitsFor((uint64_t) db->maxLen + 1); geom.kmerSlots = kmerSlots; geom.lenArr = db->len;
    ExtractArgs<LY> ea; ea.geom = geom; ea.uniS = geom.uniS;
    ea.woff = db->woff; ea.len = db->len; ea.codes = db->codes; ea.nmask = db->nmask; ea.hasN = db->hasN;
    ea.k = k; ea.kmersPerSeq = par->kmers_per_seq; ea.scale = par->kmers_per_seq_scale; ea.seed = par->hash_shift; ea.ignoreMultiKmer = par->ignore_multi_kmer;
    ea.keys = k0.p; ea.vals = vA; ea.slotOff = slotOff.p; ea.slowShort = listShort.p; ea.slowLong = listLong.p; ea.slowHuge = listHuge.p; ea.slowCnt = cls.p; ea.n = n;
    ea.hugeSp = nullptr; ea.hugeSel = nullptr; ea.hugeCap = 0;
    ea.list = nullptr; ea.nList = 0; ea.hashBase = kmerSlots; ea.rankOf = rankOf.p;
    if constexpr (LY::bySlot) {
        const char *e = cdmGetenv("CDM_SLOT_HIST");        // "kernel": sort 1 counts the head digits itself, with a read of the keys (A/B, tests)
        if (headRange || !(e && !strcmp(e, "kernel"))) {
            if (!headHist.alloc(HEAD_BINS)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
            hipMemsetAsync(headHist.p, 0, HEAD_BINS * 8, s);
            ea.headHist = headHist.p; ea.headShift = geom.headShift; headCounted = true;
        }
    }
    {   // this rank's k-mer range: equal slices of the 2k-bit k-mer space, in k-mer order
        const unsigned __int128 space = (unsigned __int128) 1 << (2 * k);
        ea.kLo = (uint64_t) (space * (unsigned) part / (unsigned) nparts);
        ea.kHi = (part == nparts - 1) ? ~0ull : (uint64_t) (space * (unsigned) (part + 1) / (unsigned) nparts);
        ea.lastPart = (part == nparts - 1) ? 1 : 0; ea.belowFlag = cls.p + 5;
        if (split) { ea.kLo = 0; ea.kHi = ~0ull; ea.lastPart = 1; ea.ordLo = ordLo; ea.ordHi = ordHi; }     // every k-mer of the owned sequences
        if (headRange) { ea.kLo = 0; ea.kHi = ~0ull; }      // (every k-mer: the head pass keeps the rank's range; the whole-sequence hash tuples stay the last rank's)
    }
    hipEventRecord(ctx->ev0, s);
    hipLaunchKernelGGL(k_seq_hash<LY>, dim3((n + 255) / 256), dim3(256), 0, s, ea);
    ea.single = listSingle.p; ea.listCount = nullptr;
    if (k <= 30) {      // two short reads per wave; what does not fit comes back through the `single` list
        hipLaunchKernelGGL(k_extract_pair<LY>, dim3(std::min<uint32_t>(((n + 1) / 2 + FAST_WAVES - 1) / FAST_WAVES, ctx->cuCount * 16)), dim3(64 * FAST_WAVES), 0, s, ea);
        ea.list = listSingle.p; ea.listCount = cls.p + 2;
    }
    hipLaunchKernelGGL(k_extract_fast<LY>, dim3(std::min<uint32_t>((n + FAST_WAVES - 1) / FAST_WAVES, ctx->cuCount * 16)), dim3(64 * FAST_WAVES), 0, s, ea);
    ea.listCount = nullptr;
    unsigned int hcls[4] = {0, 0, 0, 0};
    hipMemcpyAsync(hcls, cls.p, 16, hipMemcpyDeviceToHost, s);
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_set_error("cdm_kmermatch: extraction failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    // sequences that need the exact per-sequence ordering (repeated k-mers, more positions than the bottom-m budget)
    if (hcls[0]) {
        ea.list = listShort.p; ea.nList = hcls[0];
        hipLaunchKernelGGL((k_extract<LY, SHORT_CAP, 64>), dim3(std::min<uint32_t>(hcls[0], ctx->cuCount * 32)), dim3(64), 0, s, ea);
    }
    if (hcls[1]) {
        ea.list = listLong.p; ea.nList = hcls[1];
        hipLaunchKernelGGL((k_extract<LY, LONG_CAP, 256>), dim3(std::min<uint32_t>(hcls[1], ctx->cuCount * 2)), dim3(256), 0, s, ea);
    }
    DevBuf<SeqPos> hugeSp; DevBuf<uint8_t> hugeSel;
    if (hcls[3]) {
        uint32_t cap = LONG_CAP; while (cap < db->maxLen) cap <<= 1;
        const uint32_t blocks = std::min<uint32_t>(hcls[3], (uint32_t) ctx->cuCount * 4);       // (17 bytes of scratch per record: 2.2 MB per block for 100 k-letter contigs)
        if (!hugeSp.alloc((size_t) blocks * cap) || !hugeSel.alloc((size_t) blocks * cap)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
        ea.list = listHuge.p; ea.nList = hcls[3]; ea.hugeSp = hugeSp.p; ea.hugeSel = hugeSel.p; ea.hugeCap = cap;
        hipLaunchKernelGGL((k_extract<LY, 0, 256>), dim3(blocks), dim3(256), 0, s, ea);
    }
    hipEventRecord(ctx->ev1, s);
    unsigned int belowHost = 0;
    hipMemcpyAsync(&belowHost, cls.p + 5, 4, hipMemcpyDeviceToHost, s);
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_set_error("cdm_kmermatch: extraction (general path) failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    anyBelow = belowHost != 0;
    hipEventElapsedTime(&ctx->lastMs[3], ctx->ev0, ctx->ev1);
    if (split) return splitPartition();
    if constexpr (LY::bySlot) if (headRange) {
        unsigned long long hh[HEAD_BINS];
        if (hipMemcpy(hh, headHist.p, sizeof(hh), hipMemcpyDeviceToHost) != hipSuccess) { cdm_set_error("cdm_kmermatch: reading the head histogram failed"); return CDM_ERR_HIP; }
        unsigned long long grand = 0; for (int d = 0; d < HEAD_BINS; d++) grand += hh[d];
        std::vector<uint32_t> cut(1, 0u);
        const unsigned long long target = (grand + (unsigned) nparts - 1) / (unsigned) nparts;
        unsigned long long acc = 0;
        for (int d = 0; d < HEAD_BINS; d++) { if (acc && acc + hh[d] > target && (int) cut.size() < nparts) { cut.push_back((uint32_t) d); acc = 0; } acc += hh[d]; }
        while ((int) cut.size() < nparts) cut.push_back((uint32_t) HEAD_BINS);
        cut.push_back((uint32_t) HEAD_BINS);
        headLo = cut[part]; headHi = cut[part + 1];
        anyBelow = false; for (uint32_t d = 0; d < headLo; d++) anyBelow = anyBelow || hh[d] != 0;
    }
    return sortAndGroup();
}
// Split by reads, second step: the real tuples of the owned sequences (compacted, still in slot order) ordered by the k-mer range
// they belong to - a stable one-digit radix sort of (range, index) and a gather -, and the real whole-sequence hash tuples, which all go
// to the last range.  What a rank receives, concatenated in rank order, is then in the global slot order: ranks own consecutive blocks
// of that order.
int splitPartition() {
    // ONE radix pass on the top 8 bits of the k-mer and the bit above it (set in unused slots only, which so end up last): the real tuples
    // ordered by CDM_KPART_SLICES = 256 slices of the k-mer space, slot order inside a slice
    DevBuf<unsigned long long> cnt, bounds;
    if (!cnt.alloc(2) || !bounds.alloc((size_t) CDM_KPART_SLICES + 2)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    unsigned long long m = 0, h = 0;
    bool inFirst = true;
    const int shift = 2 * k - 8;
    if (kmerSlots) { if (int rc = rx::sortPairs<uint64_t, V>(s, ctx->cuCount, k0.p, k1.p, v0.p, v1.p, (uint64_t) kmerSlots, shift, 2 * k + 1, inFirst)) return rc; }
    uint64_t *rK = inFirst ? k0.p : k1.p, *oK = inFirst ? k1.p : k0.p; V *rV = inFirst ? v0.p : v1.p, *oV = inFirst ? v1.p : v0.p;
    if (kmerSlots) {
        hipLaunchKernelGGL(k_live_count, dim3(1), dim3(1), 0, s, (const uint64_t *) rK, (uint64_t) kmerSlots, 2 * k, cnt.p);
        hipMemcpyAsync(&m, cnt.p, 8, hipMemcpyDeviceToHost, s);
    }
    // the hash tuples of region 2, compacted (from the extraction buffers into the other pair, behind region 1)
    if (r2Slots) {
        if (int rc = rx::compactPairs<uint64_t, V>(s, k0.p + kmerSlots, v0.p + kmerSlots, (uint64_t) r2Slots, k1.p + kmerSlots, v1.p + kmerSlots, cnt.p + 1)) return rc;
        hipMemcpyAsync(&h, cnt.p + 1, 8, hipMemcpyDeviceToHost, s);
    }
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: ordering the tuples by k-mer slice failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    sendOff.assign((size_t) CDM_KPART_SLICES + 1, 0);
    if (m) {
        hipLaunchKernelGGL(k_slice_bounds, dim3(2), dim3(256), 0, s, (const uint64_t *) rK, (uint64_t) m, shift, (uint32_t) CDM_KPART_SLICES, bounds.p);
        hipMemcpyAsync(sendOff.data(), bounds.p, ((size_t) CDM_KPART_SLICES + 1) * 8, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: ordering the tuples by k-mer slice failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    }
    (void) oK; (void) oV;
    sendKeys = rK; sendVals = rV; valBytes = (int) sizeof(V);
    sendHashKeys = k1.p + kmerSlots; sendHashVals = v1.p + kmerSlots; sendHash = h;
    return CDM_OK;
}
int splitBegin() override {
    split = true;
    // (which sequences: phaseA's slot layout, blocks of about equal slot counts)
    if (nparts != CDM_KPART_SLICES) { cdm_set_error("cdm_kmermatch: internal error: the split by reads orders its tuples by %d slices", CDM_KPART_SLICES); return CDM_ERR_INVALID; }
    return phaseA();       // (a rank without sequences of its own - fewer sequences than ranks - goes through with empty buffers)
}
int splitFinish(const void *keysIn, const void *valsIn, uint64_t m, const void *hkeys, const void *hvals, uint64_t h, bool below) override {
    // the extraction's buffers go, the received tuples become the two regions of the tuple array
    slots.free(); slotOff.free(); rankOf.free();
    listShort.free(); listLong.free(); listSingle.free(); listHuge.free();
    splitK.free(); splitV.free(); k0.free(); k1.free(); v0.free(); v1.free();        // (sent: the exchange is over)
    DevBuf<uint64_t> nk0, nk1; DevBuf<V> nv0, nv1;
    const uint64_t tot = m + h;
    if (!nk0.alloc(tot) || !nk1.alloc(tot) || !nv0.alloc(tot) || !nv1.alloc(tot)) { cdm_set_error("cdm_kmermatch: out of device memory for %llu received k-mer tuples", (unsigned long long) tot); return CDM_ERR_HIP; }
    if (m) { hipMemcpyAsync(nk0.p, keysIn, m * 8, hipMemcpyDeviceToDevice, s); hipMemcpyAsync(nv0.p, valsIn, m * sizeof(V), hipMemcpyDeviceToDevice, s); }
    if (h) { hipMemcpyAsync(nk0.p + m, hkeys, h * 8, hipMemcpyDeviceToDevice, s); hipMemcpyAsync(nv0.p + m, hvals, h * sizeof(V), hipMemcpyDeviceToDevice, s); }
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: taking over the received tuples failed"); return CDM_ERR_HIP; }
    k0.p = nk0.release(); k1.p = nk1.release(); v0.p = nv0.release(); v1.p = nv1.release(); vA = v0.p; vB = v1.p;
    kmerSlots = m; r2Slots = h; nTuples = tot; geom.kmerSlots = m; anyBelow = below;
    return sortAndGroup();
}
// after splitBegin: everything goes but what would be sent - the tuples ordered by range (splitK / splitV) and the hash tuples, which
// move out of the extraction buffers into two small ones
int keepOnlyOutgoing() {
    DevBuf<uint64_t> hk; DevBuf<V> hv;
    const unsigned long long m = sendOff.empty() ? 0 : sendOff.back();
    if (!hk.alloc(sendHash) || !hv.alloc(sendHash) || !splitK.alloc(m) || !splitV.alloc(m)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    if (m) { hipMemcpyAsync(splitK.p, sendKeys, m * 8, hipMemcpyDeviceToDevice, s); hipMemcpyAsync(splitV.p, sendVals, m * sizeof(V), hipMemcpyDeviceToDevice, s); }
    if (sendHash) { hipMemcpyAsync(hk.p, sendHashKeys, sendHash * 8, hipMemcpyDeviceToDevice, s); hipMemcpyAsync(hv.p, sendHashVals, sendHash * sizeof(V), hipMemcpyDeviceToDevice, s); }
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: keeping a block's tuples failed"); return CDM_ERR_HIP; }
    k0.free(); k1.free(); v0.free(); v1.free();
    sendKeys = splitK.p; sendVals = splitV.p;
    slots.free(); slotOff.free(); rankOf.free(); listShort.free(); listLong.free(); listSingle.free(); listHuge.free(); cls.free(); counters.free();
    keptHashK.p = hk.release(); keptHashV.p = hv.release();
    sendHashKeys = keptHashK.p; sendHashVals = keptHashV.p;
    return CDM_OK;
}
DevBuf<uint64_t> keptHashK; DevBuf<V> keptHashV;
// a range of the passes on one device: its tuples, gathered by the caller (m k-mer tuples, then h whole-sequence hash tuples), are
// taken over as they are
int rangeFinishOwned(DevBuf<uint64_t> &keysBuf, DevBuf<V> &valsBuf, uint64_t m, uint64_t h, bool below) {
    passes = true; split = true;
    if (int rc = init()) return rc;
    const uint64_t tot = m + h;
    if (!k1.alloc(tot) || !v1.alloc(tot)) { cdm_set_error("cdm_kmermatch: out of device memory for a pass over %llu k-mer tuples", (unsigned long long) tot); return CDM_ERR_HIP; }
    k0.p = keysBuf.release(); v0.p = valsBuf.release(); vA = v0.p; vB = v1.p;
    kmerSlots = m; r2Slots = h; nTuples = tot; geom.kmerSlots = m; anyBelow = below;
    return sortAndGroup();
}
int sortAndGroup() {

    // ---- sort 1: stable LSD radix sort by k-mer.  Region 1 (k-mer slots) on the 2k key bits, region 2 (whole-sequence hashes)
    // on 63 bits into the same physical buffers; the strand bit 63 rides along outside the sorted bit range.
    keys = DoubleBuf<uint64_t>(k0.p, k1.p); vals = DoubleBuf<V>(vA, vB);
    // Region 1: only the top 27 sort bits go through global passes, the low bits are finished per bucket by k_bucket_groups
    // (bucket.h); CDM_KMER_SORT=lsd sorts all 2k bits globally and keeps the separate scan + k_groups kernels (A/B).
    // With low bits left over the passes cover bits [lowBits, 2k]: bit 2k is set only in unused slots, which end up last.
    // 27 high bits = 3 onesweep passes of 9 bits (library radix sorts default to 8 bits per pass; 9 still fits the LDS and three
    // 9-bit passes take 29 ms per 2^30 tuples where four 8-bit ones take 35).
    lowBits = lsdOnly ? 0 : std::max(0, 2 * k + 1 - 27);
    const int sortTop = lowBits ? 2 * k + 1 : 2 * k;
    hipEventRecord(ctx->ev0, s);
    bool slotSorted = false;
    if constexpr (LY::bySlot) {
        // one length, 8-byte tuples: the head pass drops the empty slots and writes slot tuples, the other global passes run inside the
        // head digit's segments (rx::sortSlotKeys); `live` comes out of the head histogram
        if (!segBuf.alloc(rx::BINS + 1)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
        uint64_t *res = nullptr; unsigned long long liveSlots = 0;
        if (int rc = rx::sortSlotKeys(s, ctx->cuCount, k0.p, k1.p, (uint64_t) kmerSlots, 2 * k, lowBits, headCounted ? headHist.p : nullptr, segBuf.p, liveSlots, res, &ctx->lastMs[13], &ctx->lastMs[14],
                                      headRange ? headLo : 0u, headRange ? std::min<uint32_t>(headHi, (uint32_t) rx::BINS) : (uint32_t) rx::BINS)) return rc;
        keys = DoubleBuf<uint64_t>(res, res == k0.p ? k1.p : k0.p); vals = res == k0.p ? DoubleBuf<V>(vA, vB) : DoubleBuf<V>(vB, vA);     // (region 2's values follow its keys' buffer)
        // what those launches move at the least, in GB: the head pass reads every slot's key and writes the real ones' tuples, the passes
        // inside the segments read and write every tuple (bench.py's roofline figure)
        ctx->lastMs[15] = (float) (((double) kmerSlots * 8.0 + (double) liveSlots * 8.0 + (double) (ctx->lastMs[14] - 1.f) * (double) liveSlots * 16.0) / 1e9);
        live = liveSlots; geom.seg = segBuf.p; slotSorted = true;
    } else
    if (nparts > 1 && kmerSlots && !lsdOnly && !split) {      // (split by reads: what arrived has no empty slots)
        // a k-mer RANGE: most slots are empty.  The real tuples are compacted (stable) into the other buffers first, so that the
        // passes run over this rank's share only; behind them the result holds empty slots again, as if all had been sorted.
        DevBuf<unsigned long long> cnt;
        if (!cnt.alloc(1)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
        if (int rc = rx::compactPairs<uint64_t, V>(s, k0.p, vA, (uint64_t) kmerSlots, k1.p, vB, cnt.p)) return rc;
        unsigned long long m = 0;
        hipMemcpyAsync(&m, cnt.p, 8, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: compaction failed"); return CDM_ERR_HIP; }
        bool inFirst = true;        // "first" = (k1, v1) here
        if (int rc = rx::sortPairs<uint64_t, V>(s, ctx->cuCount, k1.p, k0.p, vB, vA, (uint64_t) m, lowBits, sortTop, inFirst, &ctx->lastMs[13])) return rc;
        ctx->lastMs[14] = (float) ((sortTop - lowBits + rx::BITS - 1) / rx::BITS);
        ctx->lastMs[15] = (float) ((double) ctx->lastMs[14] * (double) m * 2.0 * (8.0 + sizeof(V)) / 1e9);
        uint64_t *kRes = inFirst ? k1.p : k0.p; V *vRes = inFirst ? vB : vA;
        hipMemsetAsync(kRes + m, 0xFF, (size_t) (kmerSlots - m) * 8, s);        // (the values of empty slots are never read)
        keys = DoubleBuf<uint64_t>(kRes, inFirst ? k0.p : k1.p); vals = DoubleBuf<V>(vRes, inFirst ? vA : vB);
    } else {
        bool inFirst = true;
        if (int rc = rx::sortPairs<uint64_t, V>(s, ctx->cuCount, k0.p, k1.p, vA, vB, (uint64_t) kmerSlots, lowBits, sortTop, inFirst, &ctx->lastMs[13])) return rc;
        ctx->lastMs[14] = (float) ((sortTop - lowBits + rx::BITS - 1) / rx::BITS);     // its launches
        ctx->lastMs[15] = (float) ((double) ctx->lastMs[14] * (double) kmerSlots * 2.0 * (8.0 + sizeof(V)) / 1e9);
        keys = DoubleBuf<uint64_t>(inFirst ? k0.p : k1.p, inFirst ? k1.p : k0.p); vals = DoubleBuf<V>(inFirst ? vA : vB, inFirst ? vB : vA);
    }
    hipEventRecord(ctx->ev1, s);
    hipEventRecord(ctx->ev2, s);
    {
        // region 2 goes to wherever region 1 ended up (the input is always the extraction buffers k0/v0)
        uint64_t *kOut = keys.current() + kmerSlots, *kIn = k0.p + kmerSlots;
        V *vOut = vals.current() + kmerSlots, *vIn = vA + kmerSlots;
        {
            bool inFirst = true;
            if (int rc = rx::sortPairs<uint64_t, V>(s, ctx->cuCount, kIn, k1.p + kmerSlots, vIn, vB + kmerSlots, (uint64_t) r2Slots, 0, 63, inFirst)) return rc;
            uint64_t *kRes = inFirst ? kIn : k1.p + kmerSlots; V *vRes = inFirst ? vIn : vB + kmerSlots;
            if (kRes != kOut && r2Slots) { hipMemcpyAsync(kOut, kRes, (size_t) r2Slots * 8, hipMemcpyDeviceToDevice, s); hipMemcpyAsync(vOut, vRes, (size_t) r2Slots * sizeof(V), hipMemcpyDeviceToDevice, s); }
        }
    }
    hipEventRecord(ctx->ev3, s);
    // ---- K3: group keys per slot (fused bucket kernel for region 1, run-start max-scan + k_groups elsewhere), then the
    // order-preserving compaction
    ga.geom = geom;
    ga.keys = keys.current(); ga.vals = vals.current(); ga.n = nTuples; ga.onlyExtendable = par->include_only_extendable; ga.covMode = par->cov_mode;
    ga.covThr = par->cov_thr; ga.idBits = idBits; ga.diagBits = diagBits; ga.diagBias = diagBias; ga.first = 0; ga.wide = wide ? 1 : 0;
    ga.firstRunIdx = (anyBelow || (split && kmerSlots == 0)) ? ~0ull : 0ull;      // (split with nothing in region 1: index 0 is a hash tuple, which is never the first run)
    // the very first run of the (global) array is in the lowest k-mer range that has tuples
    if (!statStripes.alloc(STAT_STRIPES)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    hipMemsetAsync(statStripes.p, 0, STAT_STRIPES * 8, s);
    ga.stat = statStripes.p;
    startIo = (unsigned long long *) keys.alternate();   // free after the sort
    if (!slotSorted) live = 0;
    nKept = 0;
    if (!staleBuf.alloc(STALE_MAX + 3)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    hipMemsetAsync(staleBuf.p, 0, (STALE_MAX + 3) * 4, s);
    {
        // scan + k_groups over the tuples [first, last) of (kk, vv), group keys to io[first..last)
        auto scanGroups = [&](GroupArgs<LY> g, unsigned long long *io) -> int {
            const size_t cnt = (size_t) (g.n - g.first);
            if (cnt == 0) return CDM_OK;
            cdmscan::ScanTemp t;                                                  // alive until the synchronise below
            if (int rc = cdmscan::inclusiveMaxScanFn(s, t, StartFrom<LY>{StartIndex<LY>{g.keys, g.geom, (unsigned long long) g.first}}, io + g.first, cnt)) return rc;
            if (cnt > CDM_MAX_LAUNCH_THREADS - 256) { cdm_set_error("cdm_kmermatch: %zu tuples in one grouping launch (CDM_KMER_SORT=lsd takes fewer than 2^32)", cnt); return CDM_ERR_UNSUPPORTED; }
            hipLaunchKernelGGL(k_groups<LY>, CDM_GRID((cnt + 255) / 256, 256), dim3(256), 0, s, g, io);
            return hipStreamSynchronize(s) == hipSuccess ? CDM_OK : CDM_ERR_HIP;
        };
        int rc = CDM_OK;
        if (kmerSlots && !slotSorted) {        // real tuples of region 1 (the unused slots sort behind them in both variants)
            hipLaunchKernelGGL(k_live_count, dim3(1), dim3(1), 0, s, ga.keys, (uint64_t) kmerSlots, 2 * k, counters.p + 3);
            hipMemcpyAsync(&live, counters.p + 3, 8, hipMemcpyDeviceToHost, s);
            if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: grouping failed"); return CDM_ERR_HIP; }
        }
        if (lowBits == 0) rc = scanGroups(ga, startIo);
        else {
            int own; uint32_t maxBucket; bucket::capacities(own, maxBucket);
            own = std::min(own, GkGeom<LY>::OWN); maxBucket = std::min<uint32_t>(maxBucket, (uint32_t) GkGeom<LY>::MAXB);
            DevBuf<unsigned long long> bigList; DevBuf<unsigned int> bigCnt;
            if (!bigList.alloc(bucket::bigListSlots(kmerSlots, maxBucket)) || !bigCnt.alloc(1)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
            hipMemsetAsync(bigCnt.p, 0, 4, s);
            if (kmerSlots) hipMemsetAsync(startIo + live, 0xFF, (size_t) (kmerSlots - live) * 8, s);     // unused slots: no group key
            DevBuf<BlockHead> heads; bool headsFailed = false;        // (freed behind the synchronise below)
            auto launchFused = [&](auto wordTag) {
                typedef decltype(wordTag) W;
                BucketGroupArgs<LY, W> ba;
                static_cast<GroupParams &>(ba) = ga; ba.n = live;
                ba.keys = ga.keys; ba.vals = ga.vals; ba.geom = geom; ba.out = startIo; ba.lowBits = lowBits; ba.own = own; ba.maxBucket = maxBucket;
                ba.big.list = bigList.p; ba.big.cnt = bigCnt.p;
                const uint64_t perBlock = (uint64_t) own * bucket::BK_WAVES;
                // the run records of sort 2 come out of this kernel (on the default single-device pipeline; CDM_RUN_RECORDS=kernel|twopass: from
                // the key array, as before round 5)
                stagedWaves = 0;
                if (LY::bySlot && (ownPipeline || headRange) && live && !cdmGetenv("CDM_RUN_RECORDS") && !cdmGetenv("CDM_RUN_CAP")) {
                    const uint64_t waves = (live + (uint64_t) own - 1) / (uint64_t) own;
                    ovCap = live / 256 + 4096;
                    if (stRep.alloc(waves * REC_CAP) && stVal.alloc(waves * REC_CAP) && stCnt.alloc(waves + 1) && recFlag.alloc(2) && ovRep.alloc(ovCap) && ovVal.alloc(ovCap)) {
                        hipMemsetAsync(stCnt.p, 0, waves + 1, s); hipMemsetAsync(recFlag.p, 0, 8, s);
                        ba.recRep = stRep.p; ba.recVal = stVal.p; ba.recCnt = stCnt.p; stagedWaves = waves; stagedOwn = (uint64_t) own; nBigRec = 0;
                        ba.ovRep = ovRep.p; ba.ovVal = ovVal.p; ba.ovCursor = recFlag.p; ba.ovCap = ovCap;
                        if (const char *e = cdmGetenv("CDM_REC_LIMIT")) { const long v = atol(e); if (v >= 0 && v <= REC_CAP) ba.recLimit = (uint32_t) v; }
                    } else { stRep.free(); stVal.free(); stCnt.free(); ovRep.free(); ovVal.free(); (void) hipGetLastError(); }      // (no room: the records come from the key array)
                }
                if constexpr (LY::bySlot) {
                    const uint64_t blocks = (live + perBlock - 1) / perBlock;
                    if (!heads.alloc(blocks)) { cdm_set_error("cdm_kmermatch: out of device memory"); headsFailed = true; return; }
                    if (blocks) hipLaunchKernelGGL(k_block_heads, CDM_GRID((blocks + 255) / 256, 256), dim3(256), 0, s, geom, (uint64_t) live, perBlock, blocks, heads.p);
                    ba.blockHead = heads.p;
                }
                if (live) hipLaunchKernelGGL((k_bucket_groups<LY, W>), dim3((unsigned) ((live + perBlock - 1) / perBlock)), dim3(bucket::BK_NT), cdm_lds_pad("CDM_LDS_PAD_GROUPS"), s, ba);
            };
            if (lowBits <= 15) launchFused(uint32_t()); else launchFused(uint64_t());   // 8 bits of bucket ordinal + low bits + 9 of position in one word
            if (headsFailed) return CDM_ERR_HIP;
            unsigned int nBig = 0; unsigned long long recOver = 0;
            hipMemcpyAsync(&nBig, bigCnt.p, 4, hipMemcpyDeviceToHost, s);
            if (stagedWaves) hipMemcpyAsync(&recOver, recFlag.p, 8, hipMemcpyDeviceToHost, s);
            GroupArgs<LY> g2 = ga; g2.first = kmerSlots;                      // region 2 is sorted on all its bits
            rc = scanGroups(g2, startIo);
            nOvRec = recOver;
            if (stagedWaves && recOver > ovCap) { stagedWaves = 0; stRep.free(); stVal.free(); stCnt.free(); ovRep.free(); ovVal.free(); }     // (more records beyond the waves' stages than their list holds: from the key array after all)
            if (rc == CDM_OK && nBig) {
                // buckets the kernel left alone: gather them, sort on the whole k-mer, group, scatter the group keys back
                DevBuf<unsigned long long> ranges; uint64_t total = 0; unsigned long long firstStart = ~0ull;
                rc = bucket::loadBigList(s, bigList.p, nBig, ranges, total, &firstStart);
                if (cdmGetenv("CDM_BUCKET_STATS")) fprintf(stderr, "kmermatch sort 1: %llu slots, low bits %d: %u big buckets, %llu tuples\n", (unsigned long long) kmerSlots, lowBits, nBig, (unsigned long long) total);
                DevBuf<uint64_t> dk0, dk1; DevBuf<V> dv0, dv1; DevBuf<unsigned long long> ds;
                if (rc == CDM_OK && (!dk0.alloc(total) || !dk1.alloc(total) || !dv0.alloc(total) || !dv1.alloc(total) || !ds.alloc(total))) rc = CDM_ERR_HIP;
                if (rc == CDM_OK) {
                    const unsigned int grid = bucket::bigCopyGrid(nBig);
                    if constexpr (LY::bySlot) hipLaunchKernelGGL(k_big_slot_pairs<true>, dim3(grid), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, const_cast<uint64_t *>(ga.keys), geom, dk0.p, dv0.p);
                    else {
                    hipLaunchKernelGGL((bucket::k_big_copy<uint64_t, true>), dim3(grid), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, const_cast<uint64_t *>(ga.keys), dk0.p);
                    hipLaunchKernelGGL((bucket::k_big_copy<V, true>), dim3(grid), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, const_cast<V *>(ga.vals), dv0.p);
                    }
                    bool bigFirst = true;
                    rc = rx::sortPairs<uint64_t, V>(s, ctx->cuCount, dk0.p, dk1.p, dv0.p, dv1.p, (uint64_t) total, 0, 2 * k, bigFirst);
                    DoubleBuf<uint64_t> dk(bigFirst ? dk0.p : dk1.p, bigFirst ? dk1.p : dk0.p); DoubleBuf<V> dv(bigFirst ? dv0.p : dv1.p, bigFirst ? dv1.p : dv0.p);
                    if (rc == CDM_OK) {
                        // the sorted tuples go back in place too: k_stale_tail indexes big buckets directly
                        if constexpr (LY::bySlot) hipLaunchKernelGGL(k_big_slot_pairs<false>, dim3(grid), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, const_cast<uint64_t *>(ga.keys), geom, dk.current(), dv.current());
                        else {
                        hipLaunchKernelGGL((bucket::k_big_copy<uint64_t, false>), dim3(grid), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, const_cast<uint64_t *>(ga.keys), dk.current());
                        hipLaunchKernelGGL((bucket::k_big_copy<V, false>), dim3(grid), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, const_cast<V *>(ga.vals), dv.current());
                        }
                        GroupArgs<LY> gd = ga; gd.keys = dk.current(); gd.vals = dv.current(); gd.n = total; gd.first = 0;
                        gd.geom.kmerSlots = ~0ull;                                   // every tuple of the dense view is a region-1 tuple
                        gd.firstRunIdx = (firstStart == 0 && !anyBelow) ? 0ull : ~0ull;           // dense index 0 is the array's first tuple only then
                        rc = scanGroups(gd, ds.p);
                    }
                    if (rc == CDM_OK) {
                        hipLaunchKernelGGL((bucket::k_big_copy<unsigned long long, false>), dim3(grid), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, startIo, ds.p);
                        if (hipStreamSynchronize(s) != hipSuccess) rc = CDM_ERR_HIP;
                    }
                    if (rc == CDM_OK && stagedWaves) {
                        // the run records of these buckets (the kernel staged none for them): from their group keys, a dropped key between
                        // two buckets, the starts put back into the key array's coordinates
                        DevBuf<unsigned long long> gapped;
                        if (!gapped.alloc(total + nBig)) rc = CDM_ERR_HIP;
                        else {
                            hipLaunchKernelGGL(k_big_gap_copy, dim3(grid), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, (const unsigned long long *) ds.p, gapped.p);
                            runsort::RunArgs ra; ra.keys = (const uint64_t *) gapped.p; ra.n = total + nBig; ra.skipLo = ra.skipHi = 0; ra.repShift = (int) (idBits + diagBits + 1);
                            ra.wide = wide ? 1 : 0; ra.idShift = (int) diagBits + 1; ra.idMask = (1ull << idBits) - 1ull;
                            rc = runsort::makeRunRecords(s, ra, bigRecRep, bigRecRep1, bigRecVal, bigRecVal1, nBigRec);
                            if (rc == CDM_OK && nBigRec) hipLaunchKernelGGL(k_big_rec_starts, CDM_GRID((nBigRec + 255) / 256, 256), dim3(256), 0, s, (const unsigned long long *) ranges.p, nBig, bigRecVal.p, (uint64_t) nBigRec);
                            if (rc == CDM_OK && hipStreamSynchronize(s) != hipSuccess) rc = CDM_ERR_HIP;
                            bigRecRep1.free(); bigRecVal1.free();
                        }
                    }
                }
            }
        }
        if (rc != CDM_OK) { cdm_set_error("cdm_kmermatch: grouping failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
        hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(256), 0, s, (const unsigned long long *) statStripes.p, counters.p + 4);
        hipMemcpyAsync(&nKept, counters.p + 4, 8, hipMemcpyDeviceToHost, s);
        { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_set_error("cdm_kmermatch: grouping failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    }
    hipEventElapsedTime(&msSort1, ctx->ev0, ctx->ev1);
    regionTwo = 0;
    if (nparts > 1 && part == nparts - 1 && r2Slots) {        // real whole-sequence hash tuples (they sort in front of the empty slots of region 2)
        hipLaunchKernelGGL(k_count_hash_tuples, dim3(1), dim3(1), 0, s, (const uint64_t *) ga.keys + kmerSlots, (uint64_t) r2Slots, counters.p + 6);
        hipMemcpyAsync(&regionTwo, counters.p + 6, 8, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: grouping failed"); return CDM_ERR_HIP; }
    }
    return CDM_OK;
}
// The tuples behind the kept ones that the reference's last per-target scan may run into (VoteArgs, k_stale_tail): the real
// tuples of this range from k-mer-order index J on (fromStart: from the range's first tuple, for a scan that comes in from the
// range in front), while they belong to one sequence.  Result in staleBuf (device) and staleHost.
int staleTail(unsigned long long J, bool fromStart) override {
    memset(staleHost, 0, sizeof(staleHost));
    hipMemsetAsync(staleBuf.p, 0, (STALE_MAX + 3) * 4, s);
    const unsigned long long realTuples = live + regionTwo;       // (regionTwo is only counted for multi-range runs)
    if (nparts > 1 && J >= realTuples) { staleHost[STALE_MAX + 4] = 1; return hipStreamSynchronize(s) == hipSuccess ? CDM_OK : CDM_ERR_HIP; }
    (void) fromStart;
    StaleArgs<LY> sa;
    sa.keys = ga.keys; sa.vals = ga.vals; sa.geom = geom; sa.live = live; sa.kmerSlots = kmerSlots; sa.nTuples = nTuples; sa.J = J;
    sa.lowBits = lowBits; sa.sorted = (lowBits == 0); sa.out = staleBuf.p;
    hipLaunchKernelGGL(k_stale_tail<LY>, dim3(1), dim3(256), 0, s, sa);
    hipMemcpyAsync(staleHost, staleBuf.p, (STALE_MAX + 3) * 4, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: grouping failed"); return CDM_ERR_HIP; }
    if (staleHost[0] >= (uint32_t) STALE_MAX) {
        cdm_set_error("cdm_kmermatch: the reference's last per-target scan could run over %d or more left-over tuples of one sequence; not reproduced on the device", STALE_MAX);
        return CDM_ERR_UNSUPPORTED;
    }
    staleHost[STALE_MAX + 4] = (J + staleHost[0] >= realTuples) ? 1u : 0u;     // the scan consumed every tuple up to the end of this range
    return CDM_OK;
}
int phaseB(cdm_hits **out) override {
    if (int rc = sort2((const uint64_t *) startIo, nTuples, live, kmerSlots, keys.current(), (uint64_t *) startIo, true)) return rc;
    return vote(nullptr, true, out);
}
// kept group keys of [keysIn, keysIn + nIn) (~0 = dropped; [skipLo, skipHi) holds only ~0) -> sort 2 -> vote -> hits.
// bufA / bufB: two buffers of nIn keys (bufB may be keysIn itself).
int sort2(const uint64_t *keysIn, unsigned long long nIn, unsigned long long skipLo, unsigned long long skipHi, uint64_t *bufA, uint64_t *bufB, bool ownBuffers) {

    // ---- sort 2: by (rep, id, diagonal) = key bits 1.., stable; strand bit 0 rides along.
    // Default ("runs", runsort.h): the k-mer runs are sorted by representative, not the tuples - records of (rep, start, length),
    // a stable radix sort of those, an expanding gather that also drops the ~0 keys, then a segmented sort of every
    // representative's tuples on (id, diagonal) on chip.
    // CDM_KMER_SORT2=radix (A/B; also what CDM_KMER_SORT=lsd uses): no compaction, dropped members carry the key ~0, and bit top2
    // (the first bit above the key fields) is set only there, so sorting on bits up to and including top2 moves them behind all
    // kept members; the top 32 of those bits go through global radix passes, the rest is finished bucket by bucket on chip
    // (bucket.h).  CDM_KMER_SORT2=check runs both and compares the two arrays on the device.
    if (ownBuffers) { v0.free(); v1.free(); }                                // the tuple values are dead after k_groups
    const int top2 = (int) ((wide ? idBits : 2 * idBits) + diagBits + 1);
    const char *sort2Env = cdmGetenv("CDM_KMER_SORT2");
    const bool sort2Check = sort2Env && !strcmp(sort2Env, "check");
    const bool sort2Runs = !lsdOnly && (!sort2Env || !strcmp(sort2Env, "runs") || sort2Check);
    if (wide && (!sort2Runs || sort2Check || !ownBuffers)) { cdm_set_error("cdm_kmermatch: the wide group key runs on the default pipeline only (run records + aggregated entries)"); return CDM_ERR_UNSUPPORTED; }
    if (sort2Env && strcmp(sort2Env, "runs") && strcmp(sort2Env, "radix") && !sort2Check) { cdm_set_error("cdm_kmermatch: CDM_KMER_SORT2 must be runs, radix or check"); return CDM_ERR_INVALID; }
    unsigned long long nGroup = 0;
    const uint64_t *sorted2 = nullptr;
    hipEventRecord(ctx->ev0, s);
    if (sort2Runs) {
        using namespace runsort;
        const uint64_t *gk = keysIn;
        // the sorters read the records' tuples from keysIn while they write: the sorted array goes to the OTHER buffer (bufB may be
        // keysIn); the few segments no unit holds are expanded into their final place first and sorted there
        uint64_t *sortedOut = bufA;
        if (sort2Check) {
            if (!runsOut.alloc(nIn)) { cdm_set_error("cdm_kmermatch: out of device memory (sort 2 check)"); return CDM_ERR_HIP; }
            sortedOut = runsOut.p;
        }
        RunArgs ra; ra.keys = gk; ra.n = nIn; ra.skipLo = skipLo; ra.skipHi = skipHi; ra.repShift = (int) (idBits + diagBits + 1);
        ra.wide = wide ? 1 : 0; ra.idShift = (int) diagBits + 1; ra.idMask = (1ull << idBits) - 1ull;
        cdmscan::ScanTemp stB;
        unsigned long long nRec = 0;
        DevBuf<uint32_t> rr0, rr1; DevBuf<uint64_t> rv0, rv1; DevBuf<unsigned long long> dst;
        const bool fromStage = stagedWaves && ownBuffers && keysIn == (const uint64_t *) startIo;
        if (fromStage) { if (int rc = stagedRunRecords(ra, rr0, rr1, rv0, rv1, nRec)) return rc; }
        else if (int rc = makeRunRecords(s, ra, rr0, rr1, rv0, rv1, nRec)) return rc;
        if (!dst.alloc(nRec + 1)) { cdm_set_error("cdm_kmermatch: out of device memory (%llu run records)", nRec); return CDM_ERR_HIP; }
        if (nRec) {
            bool recFirst = true;
            if (int rc = rx::sortPairs<uint32_t, uint64_t>(s, ctx->cuCount, rr0.p, rr1.p, rv0.p, rv1.p, (uint64_t) nRec, 0, (int) idBits, recFirst)) return rc;
            DoubleBuf<uint32_t> rk(recFirst ? rr0.p : rr1.p, recFirst ? rr1.p : rr0.p); DoubleBuf<uint64_t> rv(recFirst ? rv0.p : rv1.p, recFirst ? rv1.p : rv0.p);
            // (the scan reads one element past the records: the value buffers have nRec + 1 entries, the last one's length is not used)
            hipMemsetAsync(rv.current() + nRec, 0, 8, s);
            if (int rc = cdmscan::exclusiveScanFn<unsigned long long, RunLen>(s, stB, RunLen{rv.current()}, dst.p, (size_t) nRec + 1)) return rc;
            hipMemcpyAsync(&nGroup, dst.p + nRec, 8, hipMemcpyDeviceToHost, s);
            // (the expansion of the records - k_run_gather - is not run as a pass of its own: the unit sorter expands its records
            // into LDS, the few longer segments are expanded on demand)
            if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: sort 2 (records) failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
            // Default on one device: the representatives' tuples are aggregated, not sorted (aggvote.h).  The tuple path below stays for
            // the multi-GPU split (its ranks exchange heads of the sorted array), for keys too wide for the aggregation's sort word,
            // under CDM_KMER_VOTE=tuples, and as the fallback when the entry buffer overflows.
            const char *voteEnv = cdmGetenv("CDM_KMER_VOTE");
            const bool wordFits = aggv::AG_ORD + idBits + diagBits + aggv::AG_IDX <= 64 && !(wide && cdmGetenv("CDM_FORCE_WIDE_WORD"));    // (tests: the 128-bit entry sort word for any DB)
            // (a rank's second half - sortFrom, !ownBuffers - aggregates as well since round 5: the heads the ranks exchange and the continuation
            //  of a rank's last scan are expressed on entries, aggvote.h k_head_entries / VoteEntArgs::cont; CDM_DIST_VOTE=tuples keeps the tuple path)
            const char *distVote = cdmGetenv("CDM_DIST_VOTE");
            bool aggregated = wide || ((ownBuffers || !(distVote && !strcmp(distVote, "tuples"))) && !sort2Check && !(voteEnv && !strcmp(voteEnv, "tuples")) && wordFits);
            if ((wide || fromStage) && ownBuffers && nGroup != nKept) { cdm_set_error("cdm_kmermatch: internal error: %llu group tuples counted, %llu in the run records", nKept, nGroup); return CDM_ERR_HIP; }
            if (aggregated) {
                // (the wide form has no tuple path to fall back to: an entry buffer that proves too small is tried again, larger)
                const unsigned long long slack = aggv::aggChunkSlack(nGroup / runsort::U_T, nRec, ctx->cuCount);     // (the units' and segments' chunks of the entry array: aggvote.h AG_CHUNK)
                unsigned long long capEnt = nGroup / 6 + (4ull << 20) + slack;
                if (const char *e = cdmGetenv("CDM_AGG_CAP")) capEnt = strtoull(e, nullptr, 10);      // tests: force the overflow fallback
                int rc = aggregate(sortedOut, nGroup, rk.current(), (const uint64_t *) rv.current(), dst.p, nRec, gk, top2, capEnt, !wordFits);
                while (rc == CDM_ERR_UNSUPPORTED && wide && capEnt < nGroup + 1 + slack) {
                    capEnt = std::min<unsigned long long>(nGroup + 1 + slack, std::max<unsigned long long>(capEnt * 3, 1024));
                    rc = aggregate(sortedOut, nGroup, rk.current(), (const uint64_t *) rv.current(), dst.p, nRec, gk, top2, capEnt, !wordFits);
                }
                if (rc == CDM_ERR_UNSUPPORTED && !wide) aggregated = false;      // (entry buffer too small for this input: the tuple path)
                else if (rc) return rc;
            }
            haveEntries = aggregated;
            if (!aggregated && segmentedSortKeys(s, ctx->cuCount, sortedOut, sortedOut, nGroup, (int) (idBits + diagBits + 1), (int) (diagBits + 1), top2, rk.current(), dst.p, nRec, gk,
                                  (const uint64_t *) rv.current()) != CDM_OK) {
                cdm_set_error("cdm_kmermatch: segmented sort 2 failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP;
            }
        }
        sorted2 = sortedOut;
    }
    if (!sort2Runs || sort2Check) {
        // (radix variant: the keys are sorted between keysIn and bufA by the passes of radix.h; bufB == keysIn)
        const int shiftHi2 = lsdOnly ? 1 : std::max(1, top2 + 1 - 32);
        bool g2First = true;
        if (int rc = rx::sortKeys<uint64_t>(s, ctx->cuCount, const_cast<uint64_t *>(keysIn), bufA, (uint64_t) nIn, shiftHi2, top2 + 1, g2First)) return rc;
        DoubleBuf<uint64_t> g(g2First ? const_cast<uint64_t *>(keysIn) : bufA, g2First ? bufA : const_cast<uint64_t *>(keysIn));
        unsigned long long nGroupR = 0;
        if (nIn) {
            hipLaunchKernelGGL(k_live_count, dim3(1), dim3(1), 0, s, (const uint64_t *) g.current(), (uint64_t) nIn, top2, counters.p + 2);
            hipMemcpyAsync(&nGroupR, counters.p + 2, 8, hipMemcpyDeviceToHost, s);
            if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: radix sort 2 failed"); return CDM_ERR_HIP; }
        }
        const uint64_t *sortedR = g.current();
        if (shiftHi2 > 1) {
            if (bucket::bucketSortKeys(s, g.current(), g.alternate(), nGroupR, shiftHi2, 1, top2) != CDM_OK) { cdm_set_error("cdm_kmermatch: bucket sort 2 failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
            sortedR = g.alternate();
        }
        if (sort2Check) {
            if (nGroupR != nGroup) { cdm_set_error("cdm_kmermatch: sort 2 check: %llu kept tuples by runs, %llu by radix", nGroup, nGroupR); return CDM_ERR_HIP; }
            hipMemsetAsync(counters.p + 5, 0xFF, 8, s);
            if (nGroup) hipLaunchKernelGGL(k_first_diff, dim3((unsigned) std::min<uint64_t>((nGroup + 255) / 256, 65535)), dim3(256), 0, s, sorted2, sortedR, (uint64_t) nGroup, counters.p + 5);
            unsigned long long firstDiff = ~0ull, pair[2] = {0, 0};
            hipMemcpyAsync(&firstDiff, counters.p + 5, 8, hipMemcpyDeviceToHost, s);
            if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: sort 2 check failed"); return CDM_ERR_HIP; }
            if (firstDiff != ~0ull) {
                hipMemcpy(&pair[0], sorted2 + firstDiff, 8, hipMemcpyDeviceToHost); hipMemcpy(&pair[1], sortedR + firstDiff, 8, hipMemcpyDeviceToHost);
                cdm_set_error("cdm_kmermatch: sort 2 check: first difference at %llu of %llu: runs %016llx radix %016llx (idBits %u diagBits %u)", firstDiff, nGroup, pair[0], pair[1], idBits, diagBits);
                return CDM_ERR_HIP;
            }
        }
        nGroup = nGroupR; sorted2 = sortedR;
    }
    hipEventRecord(ctx->ev1, s);
    sorted2M = sorted2; nGroupM = nGroup;
    return CDM_OK;
}
// The run records of the whole key array from what the grouping kernel staged for region 1 (k_rec_compact: the waves' records packed,
// in wave order = k-mer order) + the records of region 2 (the whole-sequence hash tuples' group keys: k_run_records on that part).
int stagedRunRecords(const runsort::RunArgs &whole, DevBuf<uint32_t> &rr0, DevBuf<uint32_t> &rr1, DevBuf<uint64_t> &rv0, DevBuf<uint64_t> &rv1, unsigned long long &nRec) {
    DevBuf<unsigned long long> off; cdmscan::ScanTemp st;
    if (!off.alloc(stagedWaves + 1)) { cdm_set_error("cdm_kmermatch: out of device memory (run records)"); return CDM_ERR_HIP; }
    if (int rc = cdmscan::exclusiveScanFn<unsigned long long, RecCount>(s, st, RecCount{stCnt.p}, off.p, (size_t) stagedWaves + 1)) return rc;      // (stCnt[stagedWaves] = 0)
    unsigned long long n1 = 0, n2 = 0;
    hipMemcpyAsync(&n1, off.p + stagedWaves, 8, hipMemcpyDeviceToHost, s);
    // region 2
    DevBuf<uint32_t> q0, q1; DevBuf<uint64_t> w0, w1;
    runsort::RunArgs r2 = whole; r2.keys = whole.keys + kmerSlots; r2.n = whole.n - kmerSlots; r2.skipLo = r2.skipHi = 0; r2.base = kmerSlots;
    if (r2.n) { if (int rc = runsort::makeRunRecords(s, r2, q0, q1, w0, w1, n2)) return rc; }
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: run records failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    // the records that are not in the waves' stages - the big buckets' and the stages' overflow - as ONE list sorted by start
    unsigned long long nb = nBigRec;
    if (nOvRec) {
        DevBuf<uint64_t> x0, x1; DevBuf<uint32_t> y0, y1;
        const unsigned long long tot = nBigRec + nOvRec;
        if (!x0.alloc(tot) || !x1.alloc(tot) || !y0.alloc(tot) || !y1.alloc(tot)) { cdm_set_error("cdm_kmermatch: out of device memory (run records)"); return CDM_ERR_HIP; }
        if (nBigRec) { hipMemcpyAsync(x0.p, bigRecVal.p, nBigRec * 8, hipMemcpyDeviceToDevice, s); hipMemcpyAsync(y0.p, bigRecRep.p, nBigRec * 4, hipMemcpyDeviceToDevice, s); }
        hipMemcpyAsync(x0.p + nBigRec, ovVal.p, nOvRec * 8, hipMemcpyDeviceToDevice, s); hipMemcpyAsync(y0.p + nBigRec, ovRep.p, nOvRec * 4, hipMemcpyDeviceToDevice, s);
        bool first = true;
        if (int rc = rx::sortPairs<uint64_t, uint32_t>(s, ctx->cuCount, x0.p, x1.p, y0.p, y1.p, (uint64_t) tot, runsort::RUN_CNT_BITS, 64, first)) return rc;
        bigRecVal.free(); bigRecRep.free();
        bigRecVal.p = first ? x0.release() : x1.release(); bigRecRep.p = first ? y0.release() : y1.release();
        nb = tot;
    }
    nRec = n1 + nb + n2;
    if (!rr0.alloc(nRec) || !rr1.alloc(nRec) || !rv0.alloc(nRec + 1) || !rv1.alloc(nRec + 1)) { cdm_set_error("cdm_kmermatch: out of device memory (%llu run records)", nRec); return CDM_ERR_HIP; }
    if (n1) hipLaunchKernelGGL(k_rec_compact, CDM_GRID((stagedWaves + REC_WAVES - 1) / REC_WAVES, 256), dim3(256), 0, s, (const uint32_t *) stRep.p, (const uint64_t *) stVal.p, (const unsigned long long *) off.p, (uint64_t) stagedWaves,
                               (uint64_t) stagedOwn, (const uint64_t *) bigRecVal.p, (uint64_t) nb, rr0.p, rv0.p);
    if (nb) hipLaunchKernelGGL(k_rec_place_big, CDM_GRID((nb + 255) / 256, 256), dim3(256), 0, s, (const uint32_t *) bigRecRep.p, (const uint64_t *) bigRecVal.p, (uint64_t) nb, (const uint64_t *) stVal.p, (const uint8_t *) stCnt.p,
                               (const unsigned long long *) off.p, (uint64_t) stagedOwn, rr0.p, rv0.p);
    if (n2) { hipMemcpyAsync(rr0.p + n1 + nb, q0.p, n2 * 4, hipMemcpyDeviceToDevice, s); hipMemcpyAsync(rv0.p + n1 + nb, w0.p, n2 * 8, hipMemcpyDeviceToDevice, s); }
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: run records (packing) failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    if (cdmGetenv("CDM_BUCKET_STATS")) fprintf(stderr, "run records: %llu staged by the grouping kernel, %llu beyond its waves' stages, %llu of its big buckets, %llu of the whole-sequence hash region\n", n1, nOvRec, nBigRec, n2);
    stRep.free(); stVal.free(); stCnt.free(); bigRecRep.free(); bigRecVal.free(); ovRep.free(); ovVal.free(); stagedWaves = 0; nBigRec = 0; nOvRec = 0;
    return CDM_OK;
}
// K4 on entries: the per-representative hit counts are known already (aggregate); offsets, self hits, then one thread per segment
int voteEntries(cdm_hits **out, const uint32_t *contDev = nullptr) {
    DevBuf<unsigned long long> perRepScan;
    if (!perRepScan.alloc((size_t) n + 1)) { cdm_set_error("cdm_kmermatch: out of device memory (vote)"); return CDM_ERR_HIP; }
    cdmscan::ScanTemp st4a;
    if (int rc = cdmscan::exclusiveScan<unsigned long long>(s, st4a, agPerRep.p, perRepScan.p, (size_t) n + 1)) return rc;
    cdm_hits *res = new cdm_hits(); res->n = n;
    if (cdmMalloc(&res->off, ((size_t) n + 1) * 8) != hipSuccess) { delete res; cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    hipLaunchKernelGGL(k_offsets, dim3((n + 256) / 256), dim3(256), 0, s, perRepScan.p, n, res->off);
    uint64_t total = 0;
    hipMemcpyAsync(&total, res->off + n, 8, hipMemcpyDeviceToHost, s);
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_hits_free(res); cdm_set_error("cdm_kmermatch: vote failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    res->count = total;
    if (cdmMalloc(&res->rec, (total + 1) * sizeof(HitRec)) != hipSuccess) { cdm_hits_free(res); cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    hipLaunchKernelGGL(k_self, dim3((n + 255) / 256), dim3(256), 0, s, res->off, n, res->rec);
    aggv::VoteEntArgs va; va.ent = agEnt.p; va.entOff = agEntOff.p; va.entCnt = agEntCnt.p; va.segRep = agSegRep.p; va.nSeg = nSegM; va.hitOff = res->off; va.stale = staleBuf.p; va.diagBias = diagBias; va.cont = contDev;
    if (nSegM) hipLaunchKernelGGL(aggv::k_vote_entries<HitRec>, dim3((unsigned) ((nSegM + 255) / 256)), dim3(256), 0, s, va, res->rec);
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_hits_free(res); cdm_set_error("cdm_kmermatch: placing hits failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    float msSort2 = 0; hipEventElapsedTime(&msSort2, ctx->ev0, ctx->ev1);
    ctx->lastMs[2] = msSort1 + msSort2; ctx->lastMs[5] = msSort1; ctx->lastMs[6] = msSort2;
    hipEventElapsedTime(&ctx->lastMs[7], ctx->ev2, ctx->ev3);
    agEnt.free(); agEntOff.free(); agEntCnt.free(); agSegRep.free(); agSegOfRec.free(); agSegFirstRec.free(); agPerRep.free(); agPending.free();
    *out = res;
    return CDM_OK;
}
// sort 2 on aggregated tuples (aggvote.h): segments, k_unit_agg in k_unit_sort's place, the tuple sorters for what no table holds,
// k_rle_segment for those.  CDM_ERR_UNSUPPORTED: the entry buffer was too small (the caller takes the tuple path).
static void aggUnitHook(hipStream_t st, unsigned int grid, const unsigned long long *list, const unsigned int *count, bucket::BigList hard, void *user) {
    aggv::AggArgs *u = reinterpret_cast<aggv::AggArgs *>(user);
    aggv::AggArgs a = *u;
    a.list = list; a.count = count; a.hard = hard;
    const int cls = (int) (u->nextClass++ % runsort::U_CLASSES);       // (called once per size class, smallest first)
    const unsigned int pad = cdm_lds_pad("CDM_LDS_PAD_AGG");
    if (u->wideWord) {
        if (cls == 0) hipLaunchKernelGGL((aggv::k_unit_agg<256, runsort::U_CLASS_CAP[0] / 256, bucket::u128>), dim3(grid), dim3(256), pad, st, a);
        else if (cls == 1) hipLaunchKernelGGL((aggv::k_unit_agg<256, runsort::U_CLASS_CAP[1] / 256, bucket::u128>), dim3(grid), dim3(256), pad, st, a);
        else hipLaunchKernelGGL((aggv::k_unit_agg<256, runsort::U_CLASS_CAP[2] / 256, bucket::u128>), dim3(grid), dim3(256), pad, st, a);
        return;
    }
    if (cls == 0) hipLaunchKernelGGL((aggv::k_unit_agg<256, runsort::U_CLASS_CAP[0] / 256>), dim3(grid), dim3(256), pad, st, a);
    else if (cls == 1) hipLaunchKernelGGL((aggv::k_unit_agg<256, runsort::U_CLASS_CAP[1] / 256>), dim3(grid), dim3(256), pad, st, a);
    else hipLaunchKernelGGL((aggv::k_unit_agg<256, runsort::U_CLASS_CAP[2] / 256>), dim3(grid), dim3(256), pad, st, a);
}
int aggregate(uint64_t *sortedOut, unsigned long long nGroup, const uint32_t *recRep, const uint64_t *recVal, const unsigned long long *dst, unsigned long long nRec,
              const uint64_t *gk, int top2, unsigned long long capEnt, bool wideWord) {
    using namespace aggv;
    cdmscan::ScanTemp st;
    agEnt.free(); agSegOfRec.free(); agPerRep.free(); agCursor.free(); agFlags.free(); agSegRep.free(); agSegFirstRec.free(); agEntOff.free(); agEntCnt.free(); agPending.free();     // (a second try)
    if (!agSegOfRec.alloc(nRec + 2) || !agPerRep.alloc((size_t) n + 1) || !agCursor.alloc(2) || !agFlags.alloc(4)) { cdm_set_error("cdm_kmermatch: out of device memory (aggregation)"); return CDM_ERR_HIP; }
    hipLaunchKernelGGL(k_seg_flags, CDM_GRID((nRec + 1024) / 1024, 1024), dim3(1024), 0, s, recRep, (uint64_t) nRec, agSegOfRec.p);
    if (int rc = cdmscan::exclusiveScan<uint32_t>(s, st, agSegOfRec.p, agSegOfRec.p, (size_t) nRec + 1)) return rc;
    uint32_t nSeg = 0;
    hipMemcpyAsync(&nSeg, agSegOfRec.p + nRec, 4, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: aggregation (segments) failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    if (!agSegRep.alloc(nSeg + 1) || !agSegFirstRec.alloc((size_t) nSeg + 2) || !agEntOff.alloc((size_t) nSeg + 1) || !agEntCnt.alloc((size_t) nSeg + 1) || !agPending.alloc((size_t) nSeg + 1) ||
        !agEnt.alloc(capEnt + 1)) {
        agEnt.free(); cdm_set_error("cdm_kmermatch: out of device memory (aggregation)"); return CDM_ERR_HIP;
    }
    hipLaunchKernelGGL(k_seg_fill, CDM_GRID((nRec + 1024) / 1024, 1024), dim3(1024), 0, s, recRep, (uint64_t) nRec, agSegOfRec.p, agSegRep.p, agSegFirstRec.p, agEntCnt.p);
    hipMemsetAsync(agPerRep.p, 0, ((size_t) n + 1) * 8, s);
    hipMemsetAsync(agCursor.p, 0, 16, s); hipMemsetAsync(agFlags.p, 0, 16, s);
    AggArgs a;
    a.keys = gk; a.recVal = recVal; a.dst = dst; a.nRec = nRec; a.segOfRec = agSegOfRec.p; a.segRep = agSegRep.p; a.segFirstRec = agSegFirstRec.p; a.nSeg = nSeg;
    a.entOff = agEntOff.p; a.entCnt = agEntCnt.p; a.perRep = agPerRep.p; a.ent = agEnt.p; a.cursor = agCursor.p; a.cap = capEnt; a.overflow = agFlags.p;
    a.maxD = AG_D;
    if (const char *e = cdmGetenv("CDM_AGG_D")) { const long v = atol(e); if (v >= 1 && v <= AG_D) a.maxD = (uint32_t) v; }
    a.repShift = (int) (idBits + diagBits + 1); a.diagBits = (int) diagBits; a.idBits = idBits; a.sorted = sortedOut; a.list = nullptr; a.count = nullptr; a.hard.list = nullptr; a.hard.cnt = nullptr;
    a.wideWord = wideWord ? 1 : 0;
    if (runsort::segmentedSortKeys(s, ctx->cuCount, sortedOut, sortedOut, nGroup, (int) (idBits + diagBits + 1), (int) (diagBits + 1), top2, recRep, dst, nRec, gk, recVal, aggUnitHook, &a, wide, agSegOfRec.p, (int) bitsFor((uint64_t) nSeg + 1), agSegFirstRec.p, (uint64_t) nSeg) != CDM_OK) {
        cdm_set_error("cdm_kmermatch: segmented sort 2 failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP;
    }
    // the segments the tuple sorters finished (deep pile-ups, units with too many distinct triples)
    hipLaunchKernelGGL(k_pending_list, dim3((unsigned) (((uint64_t) nSeg + 1023) / 1024)), dim3(1024), 0, s, (const uint32_t *) agEntCnt.p, (uint64_t) nSeg, agPending.p, agFlags.p + 1);
    unsigned int fl[4] = {0, 0, 0, 0};
    hipMemcpyAsync(fl, agFlags.p, 16, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: aggregation failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    if (fl[1] && cdmGetenv("CDM_RLE_STATS")) {      // diagnosis: how many segments the tuple sorters finished, and how long they are
        std::vector<uint32_t> pl(fl[1]); std::vector<unsigned long long> fr((size_t) nSeg + 1), dd((size_t) nRec + 1);
        hipMemcpy(pl.data(), agPending.p, (size_t) fl[1] * 4, hipMemcpyDeviceToHost); hipMemcpy(fr.data(), agSegFirstRec.p, ((size_t) nSeg + 1) * 8, hipMemcpyDeviceToHost);
        hipMemcpy(dd.data(), dst, ((size_t) nRec + 1) * 8, hipMemcpyDeviceToHost);
        unsigned long long tot = 0, mx = 0, over4k = 0, over64k = 0, inBig = 0;
        for (uint32_t g : pl) { const unsigned long long m = dd[fr[g + 1]] - dd[fr[g]]; tot += m; mx = std::max(mx, m); if (m > 4096) { over4k++; inBig += m; } if (m > 65536) over64k++; }
        fprintf(stderr, "rle segments: %u of %llu, %llu tuples (longest %llu; %llu beyond 4096 tuples holding %llu, %llu beyond 65536)\n", fl[1], (unsigned long long) nSeg, tot, mx, over4k, inBig, over64k);
    }
    if (fl[1] && !fl[0]) {
        hipLaunchKernelGGL(k_rle_segment, dim3(std::min<unsigned int>((fl[1] + 3) / 4, (unsigned int) ctx->cuCount * 16)), dim3(256), 0, s, a, (const uint32_t *) agPending.p, (const unsigned int *) (agFlags.p + 1));
        hipMemcpyAsync(fl, agFlags.p, 16, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: aggregation (segments of the tuple sorters) failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    }
    unsigned long long used = 0;
    hipMemcpy(&used, agCursor.p, 8, hipMemcpyDeviceToHost);
    if (cdmGetenv("CDM_BUCKET_STATS")) fprintf(stderr, "aggregate: %llu group tuples, %u segments, %llu entries (room for %llu), %u segments through the tuple sorters%s\n", nGroup, nSeg, used, capEnt, fl[1],
                                            fl[0] ? "; entry buffer overflow" : "");
    if (fl[0]) { agEnt.free(); return CDM_ERR_UNSUPPORTED; }
    nSegM = nSeg;
    return CDM_OK;
}
// ---- K4: count hit-producing segments (per tile and per representative), scan, vote + place.  contDev: VoteArgs::cont
int vote(const uint32_t *contDev, bool ownBuffers, cdm_hits **out) {
    if (haveEntries) return voteEntries(out, contDev);
    const uint64_t *sorted2 = sorted2M; const unsigned long long nGroup = nGroupM;
    DevBuf<unsigned long long> perRep, perRepScan, vTileCnt, vTileOff;
    const uint64_t vTiles = (nGroup + CP_TILE - 1) / CP_TILE;
    if (!perRep.alloc((size_t) n + 1) || !perRepScan.alloc((size_t) n + 1) || !vTileCnt.alloc(vTiles + 1) || !vTileOff.alloc(vTiles + 1)) {
        cdm_set_error("cdm_kmermatch: out of device memory (vote)"); return CDM_ERR_HIP;
    }
    hipMemsetAsync(perRep.p, 0, ((size_t) n + 1) * 8, s);
    hipMemsetAsync(vTileCnt.p, 0, (vTiles + 1) * 8, s);
    VoteArgs va;
    if (ownBuffers && nGroup != nKept) { cdm_set_error("cdm_kmermatch: internal error: %llu group tuples counted, %llu sorted", nKept, nGroup); return CDM_ERR_HIP; }
    va.stale = staleBuf.p; va.cont = contDev;
    va.keys = sorted2; va.n = nGroup; va.idBits = idBits; va.diagBits = diagBits; va.diagBias = diagBias; va.perRep = perRep.p;
    if (nGroup) hipLaunchKernelGGL(k_seg_count, dim3((unsigned) vTiles), dim3(256), 0, s, va, vTileCnt.p);
    cdmscan::ScanTemp st4a, st4b;
    if (int rc = cdmscan::exclusiveScan<unsigned long long>(s, st4a, perRep.p, perRepScan.p, (size_t) n + 1)) return rc;
    if (int rc = cdmscan::exclusiveScan<unsigned long long>(s, st4b, vTileCnt.p, vTileOff.p, (size_t) vTiles + 1)) return rc;
    cdm_hits *res = new cdm_hits(); res->n = n;
    if (cdmMalloc(&res->off, ((size_t) n + 1) * 8) != hipSuccess) { delete res; cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    hipLaunchKernelGGL(k_offsets, dim3((n + 256) / 256), dim3(256), 0, s, perRepScan.p, n, res->off);
    uint64_t total = 0;
    hipMemcpyAsync(&total, res->off + n, 8, hipMemcpyDeviceToHost, s);
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_hits_free(res); cdm_set_error("cdm_kmermatch: vote failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    res->count = total;
    if (cdmMalloc(&res->rec, (total + 1) * sizeof(HitRec)) != hipSuccess) { cdm_hits_free(res); cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    hipLaunchKernelGGL(k_self, dim3((n + 255) / 256), dim3(256), 0, s, res->off, n, res->rec);
    if (nGroup) hipLaunchKernelGGL(k_seg_place, dim3((unsigned) vTiles), dim3(256), cdm_lds_pad("CDM_LDS_PAD_VOTE"), s, va, vTileOff.p, perRepScan.p, res->off, res->rec);
    { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) { cdm_hits_free(res); cdm_set_error("cdm_kmermatch: placing hits failed: %s", hipGetErrorString(e)); return CDM_ERR_HIP; } }
    float msSort2 = 0; hipEventElapsedTime(&msSort2, ctx->ev0, ctx->ev1);
    ctx->lastMs[2] = msSort1 + msSort2;
    ctx->lastMs[5] = msSort1;   // sort 1 call alone (1 histogram + ceil(63/8) onesweep launches)
    ctx->lastMs[6] = msSort2;
    if (ownBuffers) hipEventElapsedTime(&ctx->lastMs[7], ctx->ev2, ctx->ev3);   // sort 1, region 2 (whole-sequence hash tuples)
    *out = res;
    return CDM_OK;
}
// Multi-GPU hand-off: the kept group keys of this range grouped by representative (k-mer order inside a representative), in
// `gathered` (the first nKept entries of the buffer the sorted tuple keys were in): run records, their stable sort by rep, the
// expanding gather (runsort.h).  Slices of it by representative range are what the ranks exchange.
int gatherByRep() override {
    using namespace runsort;
    v0.free(); v1.free();
    cdmscan::ScanTemp stB;
    RunArgs ra; ra.keys = (const uint64_t *) startIo; ra.n = nTuples; ra.skipLo = live; ra.skipHi = kmerSlots; ra.repShift = (int) (idBits + diagBits + 1);
    unsigned long long nRec = 0, nOut = 0;
    DevBuf<uint32_t> rr0, rr1; DevBuf<uint64_t> rv0, rv1; DevBuf<unsigned long long> dst;
    if (stagedWaves) { if (int rc = stagedRunRecords(ra, rr0, rr1, rv0, rv1, nRec)) return rc; }       // (the slot layout's grouping kernel wrote them already)
    else if (int rc = makeRunRecords(s, ra, rr0, rr1, rv0, rv1, nRec)) return rc;
    gathered = keys.current();
    if (nRec == 0) return CDM_OK;
    if (!dst.alloc(nRec + 1)) { cdm_set_error("cdm_kmermatch: out of device memory (%llu run records)", nRec); return CDM_ERR_HIP; }
    bool recFirst = true;
    if (int rc = rx::sortPairs<uint32_t, uint64_t>(s, ctx->cuCount, rr0.p, rr1.p, rv0.p, rv1.p, (uint64_t) nRec, 0, (int) idBits, recFirst)) return rc;
    DoubleBuf<uint32_t> rk(recFirst ? rr0.p : rr1.p, recFirst ? rr1.p : rr0.p); DoubleBuf<uint64_t> rv(recFirst ? rv0.p : rv1.p, recFirst ? rv1.p : rv0.p);
    hipMemsetAsync(rv.current() + nRec, 0, 8, s);
    if (int rc = cdmscan::exclusiveScanFn<unsigned long long, RunLen>(s, stB, RunLen{rv.current()}, dst.p, (size_t) nRec + 1)) return rc;
    hipMemcpyAsync(&nOut, dst.p + nRec, 8, hipMemcpyDeviceToHost, s);
    hipLaunchKernelGGL(k_run_gather, CDM_GRID((nRec + 255) / 256, 256), dim3(256), 0, s, (const uint64_t *) startIo, (const uint64_t *) rv.current(), (const unsigned long long *) dst.p, (uint64_t) nRec, gathered);
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: gather by representative failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    if (nOut != nKept) { cdm_set_error("cdm_kmermatch: internal error: %llu group tuples counted, %llu gathered", nKept, nOut); return CDM_ERR_HIP; }
    return CDM_OK;
}
// second half on group keys received from all ranks (device buffer; concatenated in rank = k-mer order, so that the stable
// sort by representative leaves every representative's tuples in global k-mer order): sort 2, then the head of the sorted array
// for the rank in front (head: CONT_CAP + 3 values, k_head_segment; info[0] = tuples, info[1] = target id of the last one)
int sortFrom(const uint64_t *devKeys, uint64_t nKeys, uint32_t *head, uint64_t info[2]) override {
    k0.free(); k1.free(); v0.free(); v1.free();             // phase A's tuple buffers are not needed any more
    if (!recvA.alloc(nKeys) || !recvB.alloc(nKeys) || !contBuf.alloc(CONT_CAP + 4)) { cdm_set_error("cdm_kmermatch: out of device memory for %llu received group tuples", (unsigned long long) nKeys); return CDM_ERR_HIP; }
    if (nKeys) hipMemcpyAsync(recvB.p, devKeys, nKeys * 8, hipMemcpyDeviceToDevice, s);
    if (int rc = sort2(recvB.p, nKeys, 0, 0, recvA.p, recvB.p, false)) return rc;
    if (haveEntries) {      // the head and the last target from the aggregated entries (two words per entry)
        static_assert(aggv::HEAD_WORDS == CONT_CAP, "the words of a head");
        uint32_t lastId = 0;
        hipLaunchKernelGGL(aggv::k_head_entries, dim3(1), dim3(1), 0, s, (const aggv::Ent *) agEnt.p, (const unsigned long long *) agEntOff.p, (const uint32_t *) agEntCnt.p, (uint64_t) nSegM, contBuf.p);
        hipMemcpyAsync(head, contBuf.p, (CONT_CAP + 3) * 4, hipMemcpyDeviceToHost, s);
        hipMemcpyAsync(&lastId, contBuf.p + CONT_CAP + 3, 4, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: second half (aggregation) failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
        info[0] = nGroupM; info[1] = lastId;
        return CDM_OK;
    }
    hipLaunchKernelGGL(k_head_segment, dim3(1), dim3(1), 0, s, sorted2M, (uint64_t) nGroupM, idBits, diagBits, contBuf.p);
    uint64_t last = 0;
    hipMemcpyAsync(head, contBuf.p, (CONT_CAP + 3) * 4, hipMemcpyDeviceToHost, s);
    if (nGroupM) hipMemcpyAsync(&last, sorted2M + nGroupM - 1, 8, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: second half (sort) failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
    info[0] = nGroupM; info[1] = (last >> (diagBits + 1)) & ((1ull << idBits) - 1ull);
    return CDM_OK;
}
// the vote: cont = what this rank's last scan runs into (VoteArgs::cont: count, target, into-the-left-overs flag, entries; NULL:
// nothing behind this rank's tuples but the left-over list), staleIn = the combined left-over list
int voteWith(const uint32_t *cont, const uint32_t *staleIn, cdm_hits **out) override {
    if (hipMemcpyAsync(staleBuf.p, staleIn, (STALE_MAX + 3) * 4, hipMemcpyHostToDevice, s) != hipSuccess) { cdm_set_error("cdm_kmermatch: stale list upload failed"); return CDM_ERR_HIP; }
    if (cont) {
        if (cont[0] > (uint32_t) CONT_CAP) { cdm_set_error("cdm_kmermatch: the scan of this rank's last target runs over more than %d tuples of the next ranks; not reproduced", CONT_CAP); return CDM_ERR_UNSUPPORTED; }
        if (hipMemcpyAsync(contBuf.p, cont, (3 + (size_t) cont[0]) * 4, hipMemcpyHostToDevice, s) != hipSuccess) { cdm_set_error("cdm_kmermatch: continuation list upload failed"); return CDM_ERR_HIP; }
    }
    return vote(cont ? contBuf.p : nullptr, false, out);
}
};

// kmermatcher in PASSES over the k-mer space on one device, for inputs whose tuples do not fit it at once (the reference splits the
// same way when memory is short: kmermatcher.cpp:634-663, merged :742-784).  Pass r takes the tuples whose k-mer lies in range r of P:
// the sequences are extracted block by block (B blocks of the slot order; every block's tuples ordered by range, the slice of range r
// appended - the machinery of the multi-GPU split by reads), sorted and grouped as a range of a multi-GPU run is, and the group keys it
// keeps are appended to ONE array.  The ranges in order ARE the k-mer order, so that array is what a single pass leaves for sort 2,
// and sort 2 + vote run on it as they are.  What the reference's run-past-the-end scan needs (the tuples behind k-mer-order index J =
// number of kept keys, known only at the end) comes from running the range that holds J once more.  Cost: P + 2 extractions of the
// whole DB instead of one.
// OVER RANKS (ranks != NULL; csrc/dist.hip for DBs that take the wide group key): the same passes, each range run by ONE rank - every rank
// sweeps the blocks (the counts, hence the cuts, are the same everywhere), runs the ranges it owns (range r belongs to rank r W / P:
// consecutive ranges, in rank order), and the kept group keys - run starts included in the wide form, so the array describes itself -
// are all-gathered: every rank then holds the array a single device would have built and runs sort 2 + vote on it whole.  What is
// split is the first half (extraction aside), 70 % of kmermatcher; what travels is 8 bytes per kept key to every rank.
template <typename LY>
int kmermatchPassesT(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_kmer_params *par, int P, int B, cdm_hits **out, const KmerRanks *ranks = nullptr) {
    typedef typename LY::V V;
    hipStream_t s = ctx->stream;
    const int W = ranks ? ranks->world : 1, R = ranks ? ranks->rank : 0;
    if (W > 1) P = std::min(255, (std::max(P, 1) + W - 1) / W * W);
    if (P < 1 || P > 255 || B < 1) { cdm_set_error("cdm_kmermatch: %d passes over %d blocks", P, B); return CDM_ERR_INVALID; }        // (P: at most; fewer where the tuples sit in few slices of the k-mer space)
    const bool stats = cdmGetenv("CDM_BUCKET_STATS") != nullptr;
    // (this path runs because memory is short: its buffers are planned at their exact sizes, without the allocator's head room)
    struct NoHeadroom { float was; NoHeadroom() : was(cdmPoolHeadroomSwap(1.0f)) {} ~NoHeadroom() { cdmPoolHeadroomSwap(was); } } noHeadroom;
    // ---- how many tuples of every block fall into each of F = 256 fine slices of the k-mer space ([F]: the whole-sequence hash tuples,
    // which sort behind every k-mer).  The P ranges are runs of fine slices with about the same number of tuples: equal slices of the
    // k-mer space are anything but equal in tuples (1 M synthetic reads, 3 slices: 55 / 33 / 12 %).
    constexpr int F = CDM_KPART_SLICES;
    std::vector<std::vector<unsigned long long>> cnt((size_t) B, std::vector<unsigned long long>((size_t) F + 1, 0));
    auto extractBlock = [&](KmerJob<LY> &ex, int b) -> int { ex.part = 0; ex.nparts = F; ex.block = b; ex.nBlocks = B; ex.passes = true; return ex.splitBegin(); };
    std::vector<unsigned long long> fine((size_t) F, 0); unsigned long long grand = 0;
    auto poolLine = [&](const char *what, int i) { if (stats) { uint64_t st[8]; cdm_pool_stats(st); fprintf(stderr, "kmermatch passes (%d x %d): %s %d: %.1f GB mapped, %.1f GB in use\n", P, B, what, i, st[6] / 1e9, st[7] / 1e9); } };
    poolLine("start", 0);
    // The same sweep KEEPS every block's ordered tuples while they fit (the real tuples are far fewer than the slots where a per-sequence
    // budget selects the k-mers: 40 % in a late contig iteration) - then a range is gathered from the kept blocks; where they do not fit,
    // a range extracts the blocks again.  CDM_KMER_KEEP=<bytes> (tests: 0 = never keep).
    std::vector<std::unique_ptr<KmerJob<LY>>> kept((size_t) B);
    bool keepAll = true; unsigned long long keptBytes = 0, keepBudget = 0;
    { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess) keepBudget = (unsigned long long) (0.40 * (double) tot); else (void) hipGetLastError(); }
    if (const char *e = cdmGetenv("CDM_KMER_KEEP")) keepBudget = strtoull(e, nullptr, 10);
    for (int b = 0; b < B; b++) {
        std::unique_ptr<KmerJob<LY>> ex(new KmerJob<LY>(ctx, db, par));
        if (int rc = extractBlock(*ex, b)) return rc;
        for (int f = 0; f < F; f++) { cnt[b][f] = ex->sendOff[f + 1] - ex->sendOff[f]; fine[f] += cnt[b][f]; grand += cnt[b][f]; }
        cnt[b][F] = ex->sendHash;
        if (keepAll) {
            const unsigned long long bytes = (ex->sendOff[F] + ex->sendHash) * (8ull + sizeof(V));
            if (keptBytes + bytes > keepBudget) { keepAll = false; for (auto &q : kept) q.reset(); }
            else { if (int rc = ex->keepOnlyOutgoing()) return rc; keptBytes += bytes; kept[b] = std::move(ex); }
        }
        poolLine("counted block", b);
    }
    std::vector<int> cut(1, 0);          // range r = fine slices [cut[r], cut[r + 1])
    {
        const unsigned long long target = (grand + (unsigned) P - 1) / (unsigned) P;
        unsigned long long acc = 0;
        for (int f = 0; f < F; f++) { if (acc && acc + fine[f] > target && (int) cut.size() < P) { cut.push_back(f); acc = 0; } acc += fine[f]; }
        cut.push_back(F);
    }
    P = (int) cut.size() - 1;
    // a range's tuples, gathered from the blocks; then sort 1 + grouping on them
    auto runRange = [&](int r, KmerJob<LY> &job) -> int {
        const int f0 = cut[r], f1 = cut[r + 1];
        unsigned long long m = 0, h = 0; bool below = false;
        std::vector<unsigned long long> mine((size_t) B, 0);
        for (int b = 0; b < B; b++) {
            for (int f = f0; f < f1; f++) mine[b] += cnt[b][f];
            m += mine[b]; if (r == P - 1) h += cnt[b][F];
            for (int f = 0; f < f0; f++) below = below || cnt[b][f] != 0;
        }
        DevBuf<uint64_t> rk; DevBuf<V> rv;
        if (!rk.alloc(m + h) || !rv.alloc(m + h)) { cdm_set_error("cdm_kmermatch: out of device memory for a pass over %llu k-mer tuples", m + h); return CDM_ERR_HIP; }
        unsigned long long at = 0, hat = m;
        for (int b = 0; b < B; b++) {
            if (mine[b] == 0 && !(r == P - 1 && cnt[b][F])) continue;
            std::unique_ptr<KmerJob<LY>> again;
            if (!keepAll) { again.reset(new KmerJob<LY>(ctx, db, par)); if (int rc = extractBlock(*again, b)) return rc; }
            KmerJob<LY> &ex = keepAll ? *kept[b] : *again;
            if ((unsigned long long) (ex.sendOff[f1] - ex.sendOff[f0]) != mine[b] || ex.sendHash != cnt[b][F]) { cdm_set_error("cdm_kmermatch: internal error: a block's tuple counts changed between two extractions"); return CDM_ERR_HIP; }
            if (mine[b]) {
                hipMemcpyAsync(rk.p + at, (const uint64_t *) ex.sendKeys + ex.sendOff[f0], mine[b] * 8, hipMemcpyDeviceToDevice, s);
                hipMemcpyAsync(rv.p + at, (const V *) ex.sendVals + ex.sendOff[f0], mine[b] * sizeof(V), hipMemcpyDeviceToDevice, s);
                at += mine[b];
            }
            if (r == P - 1 && cnt[b][F]) {
                hipMemcpyAsync(rk.p + hat, ex.sendHashKeys, cnt[b][F] * 8, hipMemcpyDeviceToDevice, s);
                hipMemcpyAsync(rv.p + hat, ex.sendHashVals, cnt[b][F] * sizeof(V), hipMemcpyDeviceToDevice, s);
                hat += cnt[b][F];
            }
            if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: gathering a pass's tuples failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
        }
        job.part = r; job.nparts = P;
        return job.rangeFinishOwned(rk, rv, m, h, below);
    };
    // ---- the passes: kept group keys (and, in the wide form, the dropped run starts that name a representative) in k-mer order
    DevBuf<uint64_t> G; unsigned long long gCap = 0, gCount = 0, J = 0;
    std::vector<unsigned long long> realOf((size_t) P, 0);
    DevBuf<unsigned long long> cntDev;
    if (!cntDev.alloc(1)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    auto ownerOf = [&](int r) { return (int) ((long long) r * W / P); };
    std::vector<unsigned long long> keptOf((size_t) P, 0);
    for (int r = 0; r < P; r++) {
        if (ownerOf(r) != R) continue;
        KmerJob<LY> job(ctx, db, par);
        if (int rc = runRange(r, job)) return rc;
        realOf[r] = job.live + job.regionTwo; J += job.nKept; keptOf[r] = job.nKept;
        const unsigned long long nt = job.nTuples;
        if (stats) fprintf(stderr, "kmermatch pass %d of %d: %llu tuples, %llu kept\n", r + 1, P, nt, job.nKept);
        if (nt == 0) continue;
        if (gCount + nt > gCap) {       // (room for everything this range could keep; grown by doubling)
            const unsigned long long want = std::max(gCount + nt, gCap * 2);
            DevBuf<uint64_t> bigger;
            if (!bigger.alloc(want)) { cdm_set_error("cdm_kmermatch: out of device memory for %llu group keys", want); return CDM_ERR_HIP; }
            if (gCount) hipMemcpyAsync(bigger.p, G.p, gCount * 8, hipMemcpyDeviceToDevice, s);
            if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: growing the group key array failed"); return CDM_ERR_HIP; }
            G.free(); G.p = bigger.release(); gCap = want;
        }
        DevBuf<uint8_t> d0, d1;         // (the compaction moves pairs: one byte per key stands in for the value)
        if (!d0.alloc(nt) || !d1.alloc(nt)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
        if (int rc = rx::compactPairs<uint64_t, uint8_t>(s, (const uint64_t *) job.startIo, d0.p, (uint64_t) nt, G.p + gCount, d1.p, cntDev.p)) return rc;
        unsigned long long got = 0;
        hipMemcpyAsync(&got, cntDev.p, 8, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: collecting a pass's group keys failed"); return CDM_ERR_HIP; }
        gCount += got;
    }
    if (W > 1) {
        // every range's counts from its owner, then the kept keys of all ranks in rank (= range = k-mer) order
        std::vector<unsigned long long> mine(2 * (size_t) P + 1), all((2 * (size_t) P + 1) * W);
        for (int r = 0; r < P; r++) { mine[2 * r] = realOf[r]; mine[2 * r + 1] = keptOf[r]; }
        mine[2 * (size_t) P] = gCount;
        if (int rc = ranks->gatherHost(ranks->user, mine.data(), all.data(), mine.size() * 8)) return rc;
        J = 0;
        std::vector<uint64_t> recvOff((size_t) W + 1, 0);
        for (int p = 0; p < W; p++) {
            const unsigned long long *a = all.data() + (size_t) p * mine.size();
            for (int r = 0; r < P; r++) if (ownerOf(r) == p) { realOf[r] = a[2 * r]; J += a[2 * r + 1]; }
            recvOff[p + 1] = recvOff[p] + a[2 * (size_t) P] * 8;
        }
        const unsigned long long total = recvOff[W] / 8;
        DevBuf<uint64_t> whole;
        if (!whole.alloc(total)) { cdm_set_error("cdm_kmermatch: out of device memory for %llu group keys of all ranks", total); return CDM_ERR_HIP; }
        if (!G.p && !G.alloc(0)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
        if (int rc = ranks->gatherDev(ranks->user, G.p, gCount * 8, whole.p, recvOff.data(), s)) return rc;
        if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: gathering the ranks' group keys failed: %s", hipGetErrorString(hipGetLastError())); return CDM_ERR_HIP; }
        G.free(); G.p = whole.release(); gCount = total; gCap = total;
    }
    // ---- the left-over list of the reference's last per-target scan (:875-887): from k-mer-order index J on, while the tuples belong
    // to one sequence - the range that holds index J once more, and the ranges behind it while the scan runs on (csrc/dist.hip does
    // the same across ranks; over ranks here, every rank runs that range itself: no exchange, one range's work)
    uint32_t stale[CDM_STALE_MAX + 5]; memset(stale, 0, sizeof(stale));
    if (J) {
        int holder = -1; unsigned long long jLocal = 0, base = 0;
        for (int r = 0; r < P; r++) { if (J < base + realOf[r]) { holder = r; jLocal = J - base; break; } base += realOf[r]; }
        uint32_t got = 0; bool have = false; uint32_t target = 0;
        for (int r = holder; holder >= 0 && r < P; r++) {
            KmerJob<LY> job(ctx, db, par);
            if (int rc = runRange(r, job)) return rc;
            if (int rc = job.staleTail(r == holder ? jLocal : 0, false)) return rc;
            const uint32_t *l = job.staleHost;
            if (l[0]) {
                if (!have) { target = l[1]; have = true; } else if (l[1] != target) break;
                for (uint32_t j = 0; j < l[0] && got < (uint32_t) CDM_STALE_MAX; j++) stale[2 + got++] = l[2 + j];
            }
            if (!l[CDM_STALE_MAX + 4]) break;
        }
        if (got >= (uint32_t) CDM_STALE_MAX) { cdm_set_error("cdm_kmermatch: the reference's last per-target scan would run over %d or more left-over tuples; not reproduced", CDM_STALE_MAX); return CDM_ERR_UNSUPPORTED; }
        stale[0] = got; stale[1] = have ? target : 0;
    }
    // ---- sort 2 + vote on the collected keys
    for (auto &q : kept) q.reset();
    KmerJob<LY> fin(ctx, db, par);
    fin.passes = true;
    if (int rc = fin.init()) return rc;
    if (!fin.staleBuf.alloc(STALE_MAX + 3) || !fin.k0.alloc(gCount)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    if (hipMemcpyAsync(fin.staleBuf.p, stale, (STALE_MAX + 3) * 4, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_kmermatch: stale list upload failed"); return CDM_ERR_HIP; }
    if (!G.p && !G.alloc(0)) { cdm_set_error("cdm_kmermatch: out of device memory"); return CDM_ERR_HIP; }
    fin.startIo = (unsigned long long *) G.p; fin.live = gCount; fin.kmerSlots = gCount; fin.nTuples = gCount; fin.nKept = J;
    fin.keys = DoubleBuf<uint64_t>(fin.k0.p, G.p);
    if (int rc = fin.sort2(G.p, gCount, gCount, gCount, fin.k0.p, G.p, true)) return rc;
    return fin.vote(nullptr, true, out);
}
// do the tuples of one pass (bytesPerSlot for every k-mer slot, both buffers) fit 80 % of the device?
inline bool onePassFits(const cdm_seqdb *db, double bytesPerSlot) {
    size_t fr = 0, tot = 0;
    const unsigned long long slots = db->residues + 2 * db->n;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || !tot) { (void) hipGetLastError(); return true; }
    return (double) slots * bytesPerSlot * 1.1 <= 0.80 * (double) tot;
}
template <typename LY>
int kmermatchT(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_kmer_params *par, cdm_hits **out, const KmerRanks *ranks = nullptr) {
    // One pass while the tuples fit the device: 16 bytes of keys + two values per k-mer slot, two buffers of each.  CDM_KMER_PASSES=P[,B]
    // (tests, A/B): P passes over B blocks for any DB.
    int P = 1, B = 1;
    if (const char *e = cdmGetenv("CDM_KMER_PASSES")) { P = atoi(e); const char *c = strchr(e, ','); B = c ? atoi(c + 1) : P; }
    else {
        size_t fr = 0, tot = 0;
        const unsigned long long slots = db->residues + 2 * db->n;          // (an upper bound: a slot per k-mer position and two per sequence)
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) {
            const double onePass = (double) slots * (16.0 + 2.0 * sizeof(typename LY::V)) * 1.1, budget = 0.80 * (double) tot;
            if (onePass > budget) {
                P = (int) std::min(255.0, std::ceil((double) slots * (16.0 + 2.0 * sizeof(typename LY::V) + 8.0) / (0.30 * (double) tot)));
                B = (int) std::ceil((double) slots * (32.0 + 4.0 * sizeof(typename LY::V) + 16.0) / (0.30 * (double) tot));
            }
        } else (void) hipGetLastError();
    }
    if (ranks && ranks->world > 1) return kmermatchPassesT<LY>(ctx, db, par, std::max(P, ranks->world), std::max(B, 1), out, ranks);
    if (P > 1 || B > 1) return kmermatchPassesT<LY>(ctx, db, par, std::max(P, 1), std::max(B, 1), out);
    KmerJob<LY> job(ctx, db, par);
    job.ownPipeline = true;
    if (int rc = job.phaseA()) return rc;
    if (job.nKept) if (int rc = job.staleTail(job.nKept, false)) return rc;
    return job.phaseB(out);
}

}  // namespace

constexpr uint32_t MAX_SEQ_LETTERS = 1u << 22;        // (diagonals of 24 bits: a tuple position is 32 bits wide, the group key's diagonal field is what bounds it)
static bool packedLayoutFits(const cdm_seqdb *db, int k) { return 2 * k + 1 + 2 * (int) bitsFor((uint64_t) db->maxLen + 1) <= 63; }

// ---- multi-GPU: kmermatcher in two phases with an exchange in between (include/carpedeam_hip.h, carpedeam_amd/shard.py)
struct cdm_kpart { KmerJobBase *job = nullptr; uint64_t nSeq = 0; uint32_t repShift = 0; bool gatheredDone = false; };
namespace {
// first index of `keys` (sorted by representative) whose representative is >= bound[t]
__global__ void k_rep_bounds(const uint64_t *__restrict__ keys, uint64_t n, int repShift, const uint64_t *__restrict__ bound, int nb, unsigned long long *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if ((keys[mid] >> repShift) < bound[t]) lo = mid + 1; else hi = mid; }
    out[t] = lo;
}
}  // namespace
// does cdm_kmermatch_part run this DB on the slot layout with balanced head-digit ranges? (csrc/dist.hip then takes it in place of the
// extract-everything-and-order-by-slice first half)
int cdm_kmermatch_part_takes_slots(const cdm_seqdb *db, const cdm_kmer_params *par) {
    const char *lay = cdmGetenv("CDM_KMER_LAYOUT");
    return ((!lay || !strcmp(lay, "slot")) && slotLayoutFits(db, par->kmer_size) && onePassFits(db, 16.0 + 8.0) && !cdm_kmermatch_needs_wide_key(db)) ? 1 : 0;
}
extern "C" int cdm_kmermatch_part(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_kmer_params *par, int part, int nparts, cdm_kpart **out) {
    if (!ctx || !db || !par || !out || nparts < 1 || part < 0 || part >= nparts) { cdm_set_error("cdm_kmermatch_part: invalid argument"); return CDM_ERR_INVALID; }
    if (cdmGetenv("CDM_KMER_SORT") || cdmGetenv("CDM_KMER_SORT2")) { cdm_set_error("cdm_kmermatch_part: the A/B switches CDM_KMER_SORT / CDM_KMER_SORT2 apply to the single-device path only"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    cdm_kpart *h = new cdm_kpart();
    // a DB of one read length takes the 8-byte slot layout here as on one device (round 5), the rank's range a run of head digits with
    // its share of the tuples (CDM_KMER_LAYOUT=packed|wide: the 12-byte layouts and equal slices of the k-mer space by value, as before)
    if (cdm_kmermatch_part_takes_slots(db, par)) { KmerJob<LayoutSlot> *j = new KmerJob<LayoutSlot>(ctx, db, par); j->headRange = true; h->job = j; }
    else if (packedLayoutFits(db, par->kmer_size)) h->job = new KmerJob<LayoutPacked>(ctx, db, par);
    else if (db->maxLen < 65535u) h->job = new KmerJob<LayoutWide>(ctx, db, par);
    else if (db->maxLen < (1u << 20) - 1u && db->n < (1ull << 24)) h->job = new KmerJob<LayoutLong>(ctx, db, par);
    else if (db->maxLen < MAX_SEQ_LETTERS) h->job = new KmerJob<LayoutHuge>(ctx, db, par);
    else { delete h; cdm_set_error("cdm_kmermatch_part: sequences of %u letters or more are not implemented", MAX_SEQ_LETTERS); return CDM_ERR_UNSUPPORTED; }
    h->job->part = part; h->job->nparts = nparts; h->nSeq = db->n;
    h->repShift = bitsFor(db->n) + bitsFor(2ull * db->maxLen + 2) + 1;
    const int rc = h->job->phaseA();
    if (rc != CDM_OK) { cdm_kpart_free(h); return rc; }
    *out = h;
    return CDM_OK;
}
// The split by READS of the first half: every rank extracts the k-mers of its own block of sequences (blocks of the (length desc, id asc)
// slot order, so that the blocks concatenated in rank order are that order), the tuples go to the rank of their k-mer range, and sort 1 +
// grouping run there on exactly the tuples cdm_kmermatch_part would have extracted for that range - each sequence is read once per job, not
// once per rank.
extern "C" int cdm_kmermatch_split_begin(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_kmer_params *par, int rank, int nranks, cdm_kpart **out) {
    if (!ctx || !db || !par || !out || nranks < 1 || rank < 0 || rank >= nranks) { cdm_set_error("cdm_kmermatch_split_begin: invalid argument"); return CDM_ERR_INVALID; }
    if (cdmGetenv("CDM_KMER_SORT") || cdmGetenv("CDM_KMER_SORT2")) { cdm_set_error("cdm_kmermatch_split_begin: the A/B switches CDM_KMER_SORT / CDM_KMER_SORT2 apply to the single-device path only"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    cdm_kpart *h = new cdm_kpart();
    if (packedLayoutFits(db, par->kmer_size)) h->job = new KmerJob<LayoutPacked>(ctx, db, par);
    else if (db->maxLen < 65535u) h->job = new KmerJob<LayoutWide>(ctx, db, par);
    else if (db->maxLen < (1u << 20) - 1u && db->n < (1ull << 24)) h->job = new KmerJob<LayoutLong>(ctx, db, par);
    else if (db->maxLen < MAX_SEQ_LETTERS) h->job = new KmerJob<LayoutHuge>(ctx, db, par);
    else { delete h; cdm_set_error("cdm_kmermatch_split_begin: sequences of %u letters or more are not implemented", MAX_SEQ_LETTERS); return CDM_ERR_UNSUPPORTED; }
    h->job->part = 0; h->job->nparts = CDM_KPART_SLICES; h->job->block = rank; h->job->nBlocks = nranks; h->nSeq = db->n;      // the tuples ordered by FINE slices of the k-mer space: the caller cuts the ranks' ranges from all ranks' counts
    h->repShift = bitsFor(db->n) + bitsFor(2ull * db->maxLen + 2) + 1;
    const int rc = h->job->splitBegin();
    if (rc != CDM_OK) { cdm_kpart_free(h); return rc; }
    *out = h;
    return CDM_OK;
}
extern "C" int cdm_kpart_outgoing(const cdm_kpart *h, uint64_t *offsets, const void **keys, const void **vals, int *valBytes, const void **hashKeys, const void **hashVals, uint64_t *nHash) {
    if (!h || !h->job->split || !offsets || !keys || !vals || !valBytes || !hashKeys || !hashVals || !nHash) { cdm_set_error("cdm_kpart_outgoing: invalid argument"); return CDM_ERR_INVALID; }
    for (int p = 0; p <= CDM_KPART_SLICES; p++) offsets[p] = h->job->sendOff[p];
    *keys = h->job->sendKeys; *vals = h->job->sendVals; *valBytes = h->job->valBytes;
    *hashKeys = h->job->sendHashKeys; *hashVals = h->job->sendHashVals; *nHash = h->job->sendHash;
    return CDM_OK;
}
// does this DB take the wide group key (the representative not in the members' keys)?  The exchange of group keys between ranks
// carries the narrow form only; cdm_kmermatch_dist lets every rank run kmermatcher whole for such a DB.
int cdm_kmermatch_needs_wide_key(const cdm_seqdb *db) {
    return (2 * bitsFor(db->n) + bitsFor(2ull * db->maxLen + 2) + 1 > 63 || cdmGetenv("CDM_FORCE_WIDE_KEY") != nullptr) ? 1 : 0;
}
// the k-mer range the handle is to finish as (cdm_kmermatch_dist, small worlds: every rank extracts ALL sequences - split_begin as
// block 0 of 1 - and keeps range `rank` of `nranks`, cut from its own counts)
extern "C" int cdm_kpart_set_range(cdm_kpart *h, int rank, int nranks) {
    if (!h || !h->job->split || nranks < 1 || rank < 0 || rank >= nranks) { cdm_set_error("cdm_kpart_set_range: invalid argument"); return CDM_ERR_INVALID; }
    h->job->block = rank; h->job->nBlocks = nranks;
    return CDM_OK;
}
extern "C" int cdm_kmermatch_split_finish(cdm_ctx *ctx, cdm_kpart *h, const void *keys, const void *vals, uint64_t m, const void *hashKeys, const void *hashVals, uint64_t nHash, int below) {
    if (!ctx || !h || !h->job->split || (m && (!keys || !vals)) || (nHash && (!hashKeys || !hashVals))) { cdm_set_error("cdm_kmermatch_split_finish: invalid argument"); return CDM_ERR_INVALID; }
    h->job->part = h->job->block; h->job->nparts = h->job->nBlocks;          // from here on the handle is rank `part` of `nparts` k-mer ranges, as cdm_kmermatch_part leaves it
    if (nHash && h->job->part != h->job->nparts - 1) { cdm_set_error("cdm_kmermatch_split_finish: the whole-sequence hash tuples belong to the last rank"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    return h->job->splitFinish(keys, vals, m, hashKeys, hashVals, nHash, below != 0);
}
extern "C" void cdm_kpart_free(cdm_kpart *h) { if (!h) return; delete h->job; delete h; }
extern "C" int cdm_kpart_info(const cdm_kpart *h, uint64_t info[4]) {
    info[0] = h->job->live + h->job->regionTwo; info[1] = h->job->nKept; info[2] = h->job->anyBelow ? 1 : 0; info[3] = h->nSeq;
    return CDM_OK;
}
extern "C" int cdm_kpart_stale(cdm_ctx *ctx, cdm_kpart *h, uint64_t J, uint32_t out[67]) {
    CDM_HIP(hipSetDevice(ctx->device));
    const int rc = h->job->staleTail(J, false);
    if (rc != CDM_OK) return rc;
    memcpy(out, h->job->staleHost, 67 * sizeof(uint32_t));
    return CDM_OK;
}
// offsets[t] = first gathered group key whose representative is >= bounds[t] (ascending sequence ids; nb of them).  The first call
// groups the keys by representative (run records, their stable sort, the expanding gather), later calls only look bounds up.
extern "C" int cdm_kpart_gather_at(cdm_ctx *ctx, cdm_kpart *h, int nb, const uint64_t *bounds, uint64_t *offsets, const void **devKeys) {
    if (nb < 1 || !bounds || !offsets || !devKeys) { cdm_set_error("cdm_kpart_gather_at: invalid argument"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    if (!h->gatheredDone) { const int rc = h->job->gatherByRep(); if (rc != CDM_OK) return rc; h->gatheredDone = true; }
    DevBuf<uint64_t> dBound; DevBuf<unsigned long long> dOut;
    if (!dBound.alloc(nb) || !dOut.alloc(nb)) { cdm_set_error("cdm_kpart_gather_at: out of device memory"); return CDM_ERR_HIP; }
    CDM_HIP(hipMemcpyAsync(dBound.p, bounds, (size_t) nb * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_rep_bounds, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, (const uint64_t *) h->job->gathered, (uint64_t) h->job->nKept, (int) h->repShift, (const uint64_t *) dBound.p, nb, dOut.p);
    std::vector<unsigned long long> o((size_t) nb);
    CDM_HIP(hipMemcpyAsync(o.data(), dOut.p, (size_t) nb * 8, hipMemcpyDeviceToHost, ctx->stream));
    CDM_HIP(hipStreamSynchronize(ctx->stream));
    for (int t = 0; t < nb; t++) offsets[t] = o[t];
    *devKeys = h->job->gathered;
    return CDM_OK;
}
extern "C" int cdm_kpart_gather(cdm_ctx *ctx, cdm_kpart *h, int nranks, uint64_t *offsets, const void **devKeys) {
    if (nranks < 1 || !offsets || !devKeys) { cdm_set_error("cdm_kpart_gather: invalid argument"); return CDM_ERR_INVALID; }
    std::vector<uint64_t> bound((size_t) nranks + 1);
    for (int r = 0; r <= nranks; r++) bound[r] = (uint64_t) ((unsigned __int128) h->nSeq * (unsigned) r / (unsigned) nranks);
    if (int rc = cdm_kpart_gather_at(ctx, h, nranks + 1, bound.data(), offsets, devKeys)) return rc;
    offsets[nranks] = h->job->nKept;
    return CDM_OK;
}
extern "C" int cdm_kpart_sort(cdm_ctx *ctx, cdm_kpart *h, const void *devKeys, uint64_t nKeys, uint32_t *head, uint64_t info[2]) {
    if (!head || !info || (nKeys && !devKeys)) { cdm_set_error("cdm_kpart_sort: invalid argument"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    return h->job->sortFrom((const uint64_t *) devKeys, nKeys, head, info);
}
extern "C" int cdm_kpart_vote(cdm_ctx *ctx, cdm_kpart *h, const uint32_t *cont, const uint32_t *stale, cdm_hits **out) {
    if (!out || !stale) { cdm_set_error("cdm_kpart_vote: invalid argument"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    return h->job->voteWith(cont, stale, out);
}
extern "C" int cdm_kpart_cont_cap(void) { return CONT_CAP; }
extern "C" int cdm_dev_copy(cdm_ctx *ctx, void *dst, const void *src, uint64_t bytes) {
    CDM_HIP(hipSetDevice(ctx->device));
    if (bytes) CDM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    CDM_HIP(hipStreamSynchronize(ctx->stream));
    return CDM_OK;
}

int cdm_kmermatch_impl(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_kmer_params *par, cdm_hits **out) { return cdm_kmermatch_ranks_impl(ctx, db, par, nullptr, out); }
// ranks != NULL: kmermatcher's first half split over the ranks by ranges of the k-mer space, the whole hit set on every rank (the passes
// path above; csrc/dist.hip cuts the owned view out of it)
int cdm_kmermatch_ranks_impl(cdm_ctx *ctx, const cdm_seqdb *db, const cdm_kmer_params *par, const KmerRanks *ranks, cdm_hits **out) {
    // packed 12-byte tuples when k-mer, position and length share 63 key bits; CDM_KMER_LAYOUT=wide|packed pins one (tests)
    const int k = par->kmer_size;
    const bool fits = 2 * k + 1 + 2 * (int) bitsFor((uint64_t) db->maxLen + 1) <= 63;
    bool packed = fits;
    if (const char *e = cdmGetenv("CDM_KMER_LAYOUT")) {
        if (!strcmp(e, "wide")) packed = false;
        else if (!strcmp(e, "slot")) {}
        else if (!strcmp(e, "packed")) {
            if (!fits) { cdm_set_error("cdm_kmermatch: CDM_KMER_LAYOUT=packed needs 2k + 1 + 2 x length bits <= 63 (k %d, max length %u)", k, db->maxLen); return CDM_ERR_INVALID; }
        } else { cdm_set_error("cdm_kmermatch: CDM_KMER_LAYOUT must be wide, packed or slot"); return CDM_ERR_INVALID; }
    }
    // one length throughout (reads straight from a sequencer, the bench's 50 M x 100 bp): 8-byte tuples through sort 1 (LayoutSlot) - on
    // the default single-pass pipeline only (no A/B sort variant, one device, tuples that fit the device at once)
    {
        const char *e = cdmGetenv("CDM_KMER_LAYOUT");
        const bool want = !e || !strcmp(e, "slot");
        if (e && !strcmp(e, "slot") && !slotLayoutFits(db, k)) { cdm_set_error("cdm_kmermatch: CDM_KMER_LAYOUT=slot needs sequences of one length (at least k letters), fewer than 2^32 k-mer slots and 14 <= k <= 20"); return CDM_ERR_INVALID; }
        if (want && !ranks && slotLayoutFits(db, k) && !cdmGetenv("CDM_KMER_SORT") && !cdmGetenv("CDM_KMER_PASSES") && onePassFits(db, 16.0 + 8.0)) return kmermatchT<LayoutSlot>(ctx, db, par, out);
    }
    if (packed) return kmermatchT<LayoutPacked>(ctx, db, par, out, ranks);
    if (db->maxLen < 65535u && !cdmGetenv("CDM_FORCE_HUGE_LAYOUT")) return kmermatchT<LayoutWide>(ctx, db, par, out, ranks);
    if (db->maxLen < (1u << 20) - 1u && db->n < (1ull << 24) && !cdmGetenv("CDM_FORCE_HUGE_LAYOUT")) return kmermatchT<LayoutLong>(ctx, db, par, out, ranks);
    if (db->maxLen < MAX_SEQ_LETTERS) return kmermatchT<LayoutHuge>(ctx, db, par, out, ranks);       // (CDM_FORCE_HUGE_LAYOUT=1 with CDM_KMER_LAYOUT=wide: this layout for any DB, tests)
    cdm_set_error("cdm_kmermatch: sequences of %u letters or more are not implemented", MAX_SEQ_LETTERS);
    return CDM_ERR_UNSUPPORTED;
}
