// Device-side synthetic read generator: the counter-based generator specified in carpedeam_amd/synth.py.
// The genome is never stored: base i is a hash of (seed, i).  One thread per 16-base output word.
#include "scan.h"

#include "common.h"
#include "devutil.h"

namespace {
constexpr uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;
__host__ __device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += GOLDEN;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ uint64_t streamBase(uint64_t seed, uint64_t s) { return mix(seed * GOLDEN + s); }

// floor(p * 2^64) for the C>T rows of example/dhigh5p.prof and the G>A rows of example/dhigh3p.prof
__constant__ uint64_t T5[5] = {0x5453e2d6238da3c2ull, 0x38c447c30d306a2bull, 0x300baa582dbe7f2cull, 0x2944241c3efae792ull, 0x24dde7a743a647fdull};
__constant__ uint64_t T3[5] = {0x5433721d53cddd6eull, 0x393111f0c34c1a8aull, 0x304806290eed02cdull, 0x2a175d13d74d594full, 0x2577531db445ed4aull};

struct SynthArgs {
    uint64_t b0, b1, b2, b3, b4, b5;   // stream bases
    uint64_t G, first; uint32_t n, lo, hi;
};
__global__ void k_synth_len(SynthArgs a, uint32_t *__restrict__ len, uint32_t *__restrict__ words, uint32_t *__restrict__ key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint64_t r = a.first + i;
    const uint32_t L = (a.lo == a.hi) ? a.lo : a.lo + (uint32_t) (mix(a.b5 + r) % (uint64_t) (a.hi - a.lo + 1));
    len[i] = L; words[i] = (L + 15) / 16; key[i] = i;
}
__global__ __launch_bounds__(256) void k_synth(SynthArgs a, const uint32_t *__restrict__ woff, const uint32_t *__restrict__ len, uint64_t words,
                                               uint32_t *__restrict__ codes) {
    const uint64_t gw = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (gw >= words) return;
    uint64_t lo = 0, hi = a.n;
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (woff[mid] <= gw) lo = mid; else hi = mid; }
    const uint32_t i = (uint32_t) lo, w = (uint32_t) (gw - woff[i]), L = len[i];
    const uint64_t r = a.first + i;
    const uint64_t start = mix(a.b1 + r) % (a.G - L + 1);
    const bool rev = (mix(a.b2 + r) & 1ull) != 0;
    const uint32_t cnt = min(16u, L - min(L, w * 16u));
    uint32_t code = 0;
    for (uint32_t j = 0; j < cnt; j++) {
        const uint32_t p = w * 16 + j;
        const uint64_t gi = rev ? (start + (L - 1 - p)) : (start + p);
        uint32_t c = (uint32_t) (mix(a.b0 + gi) & 3ull);
        if (rev) c = 3u - c;
        if (p < 5 && c == 1u && mix(a.b3 + r * 8 + p) < T5[p]) c = 3u;                       // 5' C>T
        const uint32_t k3 = L - 1 - p;
        if (k3 < 5 && c == 2u && mix(a.b4 + r * 8 + k3) < T3[k3]) c = 0u;                    // 3' G>A
        code |= c << (2 * j);
    }
    codes[gw] = code;
}
}  // namespace

int cdm_synth_impl(cdm_ctx *ctx, uint64_t nTotal, uint64_t first, uint64_t n, uint32_t lo, uint32_t hi, uint64_t seed, cdm_seqdb **out) {
    if (n == 0 || n >= 0xFFFFFFFFull || lo == 0 || hi < lo || first + n > nTotal) { cdm_set_error("cdm_seqdb_synth: invalid arguments"); return CDM_ERR_INVALID; }
    CDM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const double meanL = (lo == hi) ? (double) lo : (lo + hi) / 2.0;
    SynthArgs a;
    a.G = std::max<uint64_t>((uint64_t) (nTotal * meanL / 20), (uint64_t) hi + 1);
    a.first = first; a.n = (uint32_t) n; a.lo = lo; a.hi = hi;
    a.b0 = streamBase(seed, 0); a.b1 = streamBase(seed, 1); a.b2 = streamBase(seed, 2); a.b3 = streamBase(seed, 3); a.b4 = streamBase(seed, 4); a.b5 = streamBase(seed, 5);
    cdm_seqdb *db = nullptr;
    int rc = cdm_seqdb_alloc(ctx, n, &db);
    if (rc) return rc;
    uint32_t *wordsPer = nullptr;
    cdmscan::ScanTemp scanTmp;
    int ret = CDM_OK;
    do {
        if (cdmMalloc(&wordsPer, (n + 1) * 4) != hipSuccess) { cdm_set_error("cdm_seqdb_synth: out of device memory"); ret = CDM_ERR_HIP; break; }
        hipLaunchKernelGGL(k_synth_len, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, a, db->len, wordsPer, db->key);
        hipMemsetAsync(wordsPer + n, 0, 4, s);
        if ((ret = cdmscan::exclusiveScan<uint32_t>(s, scanTmp, wordsPer, db->woff, (size_t) n + 1)) != CDM_OK) break;
        uint32_t words = 0;
        hipMemcpyAsync(&words, db->woff + n, 4, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { cdm_set_error("cdm_seqdb_synth: length kernel failed"); ret = CDM_ERR_HIP; break; }
        db->words = words;
        const uint64_t maskWords = ((uint64_t) words * 16 + 31) / 32 + 1;
        if (cdmMalloc(&db->codes, ((size_t) words + 2) * 4) != hipSuccess || cdmMalloc(&db->nmask, maskWords * 4) != hipSuccess) { cdm_set_error("cdm_seqdb_synth: out of device memory"); ret = CDM_ERR_HIP; break; }
        hipMemsetAsync(db->nmask, 0, maskWords * 4, s);
        hipMemsetAsync(db->ext, 0, n, s);
        hipMemsetAsync(db->hasN, 0, n, s);
        hipLaunchKernelGGL(k_synth, CDM_GRID(((uint64_t) words + 255) / 256, 256), dim3(256), 0, s, a, db->woff, db->len, (uint64_t) words, db->codes);
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) { cdm_set_error("cdm_seqdb_synth: generator failed: %s", hipGetErrorString(e)); ret = CDM_ERR_HIP; break; }
        db->maxLen = hi; db->nCount = 0;
        db->residues = (lo == hi) ? n * (uint64_t) lo : 0;
        if (lo != hi) {   // sum of lengths
            std::vector<uint32_t> l(n);
            hipMemcpy(l.data(), db->len, n * 4, hipMemcpyDeviceToHost);
            uint64_t t = 0; uint32_t mx = 0; for (uint32_t v : l) { t += v; mx = std::max(mx, v); }
            db->residues = t; db->maxLen = mx;
        }
    } while (0);
    cdmFree(wordsPer);
    if (ret != CDM_OK) { cdm_seqdb_free(db); return ret; }
    *out = db;
    return CDM_OK;
}
