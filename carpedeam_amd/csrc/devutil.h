// Device helpers shared by the stage kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#define CDM_WAVE 64

// ---------------------------------------------------------------------------------------------- packed bases
// codes: 16 bases per u32, base i of a sequence at word woff + (i >> 4), bits 2*(i&15)..+1, A,C,G,T = 0..3
__device__ __forceinline__ uint32_t cdm_base(const uint32_t *__restrict__ codes, uint32_t woff, uint32_t pos) {
    return (codes[woff + (pos >> 4)] >> ((pos & 15u) * 2u)) & 3u;
}
// N bit of base pos of the sequence starting at word woff (bit index 16*woff + pos)
__device__ __forceinline__ uint32_t cdm_isN(const uint32_t *__restrict__ nmask, uint32_t woff, uint32_t pos) {
    uint64_t bit = (uint64_t) woff * 16u + pos;
    return (nmask[bit >> 5] >> (bit & 31u)) & 1u;
}
// original byte of base pos of a sequence with a row in the raw plane (same index space as the N bits)
__device__ __forceinline__ uint8_t cdm_raw_at(const uint8_t *__restrict__ raw, uint32_t woff, uint32_t pos) { return raw[(uint64_t) woff * 16u + pos]; }
// nucleotideMap[c] of the assembler modules (an unordered_map<char,int> holding A,C,G,T: operator[] gives 0 for every other byte;
// src/assembler/correction.cpp:170-174)
__device__ __forceinline__ uint32_t cdm_raw_base(uint8_t r) { return r == 'C' ? 1u : r == 'G' ? 2u : r == 'T' ? 3u : 0u; }
// 16 bases starting at base position pos (may straddle two words); positions beyond the sequence are garbage
__device__ __forceinline__ uint32_t cdm_window16(const uint32_t *__restrict__ codes, uint32_t woff, uint32_t pos, uint32_t lastWord) {
    uint32_t w = pos >> 4, sh = (pos & 15u) * 2u;
    uint32_t lo = codes[woff + w];
    uint32_t hi = (w + 1 <= lastWord) ? codes[woff + w + 1] : 0u;
    return sh ? ((lo >> sh) | (hi << (32u - sh))) : lo;
}
// reverse the order of the sixteen 2-bit groups of a word and complement them (A<->T, C<->G: 3 - c)
__device__ __forceinline__ uint32_t cdm_revcomp16(uint32_t x) {
    x = ~x;
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}
// 16 bases of the (optionally reverse-complemented) sequence starting at oriented position i (i < L)
__device__ __forceinline__ uint32_t cdm_oriented_window16(const uint32_t *__restrict__ codes, uint32_t w0, uint32_t L, uint32_t lastWord, bool rc, uint32_t i) {
    if (!rc) return cdm_window16(codes, w0, i, lastWord);
    const int s = (int) L - 16 - (int) i;
    const uint32_t w = (s >= 0) ? cdm_window16(codes, w0, (uint32_t) s, lastWord) : (cdm_window16(codes, w0, 0, lastWord) << (2 * (-s)));
    return cdm_revcomp16(w);
}
// spread the low 16 bits of x to the even bit positions
__device__ __forceinline__ uint32_t cdm_spread16(uint32_t x) {
    x &= 0xFFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
// collapse the even bits of x into the low 16 bits
__device__ __forceinline__ uint32_t cdm_squash16(uint32_t x) {
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}

__device__ __forceinline__ int cdm_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ uint64_t cdm_ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int cdm_wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// slot for this lane in a global append list: one atomic per wave instead of one per lane (a single word saturates at
// ~88 atomics/us, MI355X_MICROARCH.md "dequeue"); every lane of the wave must call it
__device__ __forceinline__ uint32_t cdm_wave_append(unsigned int *counter, bool pred) {
    const uint64_t m = __ballot(pred);
    if (m == 0) return 0;
    const int lane = threadIdx.x & 63, leader = __ffsll((unsigned long long) m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned int) __popcll(m));
    base = __shfl(base, leader, 64);
    return base + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
}

// block-wide version (blockDim.x threads, up to 1024): one atomic per block; every thread of the block must call it
__device__ __forceinline__ uint32_t cdm_block_append(unsigned int *counter, bool pred) {
    __shared__ uint32_t sWaveCnt[16];
    __shared__ uint32_t sBase;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const uint64_t m = __ballot(pred);
    if (lane == 0) sWaveCnt[wave] = (uint32_t) __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < nw; w++) { const uint32_t c = sWaveCnt[w]; sWaveCnt[w] = tot; tot += c; }
        sBase = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    const uint32_t r = sBase + sWaveCnt[wave] + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------- block sum / scan
// Wave-level inclusive sum (DPP-free shuffles; 6 steps), then one LDS step across the waves of the block.  Every thread of
// the block must call these; blockDim.x is a multiple of 64, at most 1024.
template <typename T> __device__ __forceinline__ T cdm_shfl_up_t(T v, int d) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit integers");
    if (sizeof(T) == 4) return (T) (unsigned int) __shfl_up((int) v, d, 64);
    const unsigned long long x = (unsigned long long) v;
    return (T) (((unsigned long long) (unsigned int) __shfl_up((int) (x >> 32), d, 64) << 32) | (unsigned int) __shfl_up((int) (unsigned int) x, d, 64));
}
template <typename T> __device__ __forceinline__ T cdm_wave_incl_sum(T v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const T o = cdm_shfl_up_t<T>(v, d); if (lane >= d) v += o; }
    return v;
}
// exclusive prefix sum of v over the block; total = sum over the block (valid in every thread)
template <typename T> __device__ __forceinline__ T cdm_block_excl_sum(T v, T &total) {
    __shared__ T sWave[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const T incl = cdm_wave_incl_sum<T>(v);
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    T base = 0, tot = 0;
    for (int w = 0; w < nw; w++) { const T c = sWave[w]; if (w < wave) base += c; tot += c; }
    __syncthreads();
    total = tot;
    return base + incl - v;
}
template <typename T> __device__ __forceinline__ T cdm_block_sum(T v) { T tot; (void) cdm_block_excl_sum<T>(v, tot); return tot; }

// ---------------------------------------------------------------------------------------------- x87 extended precision
// Software model of the x87 80-bit format (64-bit significand, round to nearest even), for the `long double`
// accumulators of the reference (src/assembler/correction.cpp:82,110-111; nuclassembleUtil.cpp:212,279).  Only what
// those accumulations need: conversion from double (exact), addition, ordering, conversion to double.
// value = (-1)^s * m * 2^(e-63); m has bit 63 set unless the value is zero (m == 0).  No infinities/NaNs (the terms are
// finite logs); exponent range is an int, so no overflow handling is needed for sums of < 2^31 doubles.
struct X87 {
    uint64_t m;
    int32_t e;
    uint32_t s;
};
__device__ __forceinline__ X87 x87_zero() { X87 r; r.m = 0; r.e = 0; r.s = 0; return r; }
__device__ __forceinline__ X87 x87_from_double(double d) {
    uint64_t b = (uint64_t) __double_as_longlong(d);
    X87 r;
    r.s = (uint32_t) (b >> 63);
    int E = (int) ((b >> 52) & 0x7FF);
    uint64_t f = b & 0xFFFFFFFFFFFFFull;
    if (E == 0) {
        if (f == 0) { r.m = 0; r.e = 0; return r; }
        int lz = __clzll((long long) f);           // subnormal: normalise
        r.m = f << lz;
        r.e = -1022 - 52 + (63 - lz);
        return r;
    }
    r.m = (1ull << 63) | (f << 11);
    r.e = E - 1023;
    return r;
}
// |a| >= |b| assumed by the callers below
__device__ __forceinline__ bool x87_mag_lt(const X87 &a, const X87 &b) {
    if (a.m == 0) return b.m != 0;
    if (b.m == 0) return false;
    if (a.e != b.e) return a.e < b.e;
    return a.m < b.m;
}
__device__ __forceinline__ X87 x87_round(uint64_t hi, uint64_t lo, int e, uint32_t s) {
    // hi has bit 63 set; lo holds the bits below the significand (bit 63 = guard, rest sticky)
    uint64_t guard = lo >> 63, sticky = (lo << 1) != 0;
    if (guard && (sticky || (hi & 1ull))) {
        hi += 1;
        if (hi == 0) { hi = 1ull << 63; e += 1; }
    }
    X87 r; r.m = hi; r.e = e; r.s = s; return r;
}
__device__ __forceinline__ X87 x87_add(X87 a, X87 b) {
    if (b.m == 0) return a;           // x + (+-0) = x  (and +0 + -0 = +0: a is returned)
    if (a.m == 0) return b;
    if (x87_mag_lt(a, b)) { X87 t = a; a = b; b = t; }
    const int d = a.e - b.e;
    if (a.s == b.s) {
        if (d >= 65) return a;        // b < half an ulp of a
        uint64_t bhi, blo;
        if (d == 0) { bhi = b.m; blo = 0; }
        else if (d < 64) { bhi = b.m >> d; blo = b.m << (64 - d); }
        else { bhi = 0; blo = b.m; }  // d == 64
        uint64_t hi = a.m + bhi, lo = blo;
        int e = a.e;
        if (hi < a.m) {               // carry out of bit 63
            lo = (lo >> 1) | (hi << 63) | (lo & 1ull);
            hi = (hi >> 1) | (1ull << 63);
            e += 1;
        }
        return x87_round(hi, lo, e, a.s);
    }
    // opposite signs: |a| > |b| or equal
    if (a.e == b.e && a.m == b.m) return x87_zero();   // exact cancellation -> +0 in round-to-nearest
    if (d > 66) return a;             // |b| < a quarter ulp even when a is a power of two
    uint64_t bhi, blo;
    if (d == 0) { bhi = b.m; blo = 0; }
    else if (d < 64) { bhi = b.m >> d; blo = b.m << (64 - d); }
    else if (d == 64) { bhi = 0; blo = b.m; }
    else { bhi = 0; blo = (b.m >> (d - 64)) | ((b.m << (128 - d)) != 0 ? 1ull : 0ull); }   // jam the lost bits
    uint64_t lo = 0 - blo;
    uint64_t hi = a.m - bhi - (blo != 0 ? 1ull : 0ull);
    int e = a.e;
    if (hi == 0) { hi = lo; lo = 0; e -= 64; }
    int lz = __clzll((long long) hi);
    if (lz) { hi = (hi << lz) | (lo >> (64 - lz)); lo <<= lz; e -= lz; }
    return x87_round(hi, lo, e, a.s);
}
// a + b as x87_add gives it, for the running sums: the common case there - same sign, the sum's exponent above the term's by
// 1..63, so |a| > |b| - skips the ordering and the general alignment
__device__ __forceinline__ X87 x87_acc(const X87 &a, const X87 &b) {
    const uint32_t d = (uint32_t) (a.e - b.e);
    if (!(a.s == b.s && (d - 1u) < 63u && a.m != 0 && b.m != 0)) return x87_add(a, b);
    // (without branches: the lanes of a wave take the carry and the round-up cases at random)
    const uint64_t bhi = b.m >> d, lo = b.m << (64u - d);
    const uint64_t hi = a.m + bhi;
    const bool c = hi < a.m;                              // carry out of bit 63: the sum moves down one bit
    const uint32_t guard = c ? (uint32_t) (hi & 1ull) : (uint32_t) (lo >> 63);
    const bool sticky = c ? (lo != 0) : ((lo << 1) != 0);
    uint64_t m = c ? ((hi >> 1) | (1ull << 63)) : hi;
    m += guard & ((uint32_t) sticky | (uint32_t) (m & 1ull));   // round to nearest even
    const bool ovf = m == 0;                              // all ones + 1
    X87 r; r.m = ovf ? (1ull << 63) : m; r.e = a.e + (int) c + (int) ovf; r.s = a.s;
    return r;
}
// a < b
__device__ __forceinline__ bool x87_lt(const X87 &a, const X87 &b) {
    if (a.m == 0 && b.m == 0) return false;
    if (a.m == 0) return b.s == 0;
    if (b.m == 0) return a.s == 1;
    if (a.s != b.s) return a.s == 1;
    return a.s ? x87_mag_lt(b, a) : x87_mag_lt(a, b);
}
// conversion to double with round to nearest even (as `double x = longDoubleValue;`)
__device__ __forceinline__ double x87_to_double(const X87 &a) {
    if (a.m == 0) return a.s ? -0.0 : 0.0;
    uint64_t m = a.m;
    int e = a.e;
    uint64_t keep = m >> 11, rest = m & 0x7FFull;
    if (rest > 0x400ull || (rest == 0x400ull && (keep & 1ull))) {
        keep += 1;
        if (keep == (1ull << 53)) { keep >>= 1; e += 1; }
    }
    // normal range only (|value| in [2^-1022, 2^1024)): the accumulated logs are O(1..1e5)
    uint64_t bits = ((uint64_t) a.s << 63) | ((uint64_t) (e + 1023) << 52) | (keep & 0xFFFFFFFFFFFFFull);
    return __longlong_as_double((long long) bits);
}
