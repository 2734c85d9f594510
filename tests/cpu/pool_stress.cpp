// The library's device-memory cache (carpedeam_amd/csrc/pool.h) against stand-ins for the HIP calls it makes, so that the very code
// the GPU library runs can be driven by many threads under ThreadSanitizer / AddressSanitizer on a box without a GPU
// (tests/test_pool_stress.py builds and runs it).  The stand-in "device" is the C heap with a byte budget: exceeding it makes
// hipMalloc fail, which drives the cache's release-everything-and-retry path; it also checks that every pointer freed was
// allocated, is freed once, and that nothing is leaked behind the caches.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <random>
#include <set>
#include <thread>
#include <unordered_map>
#include <cstring>
#include <vector>

typedef int hipError_t;
static const hipError_t hipSuccess = 0, hipErrorOutOfMemory = 2;
namespace fake {
std::mutex m;
std::set<void *> live;
std::atomic<size_t> bytes{0}, fails{0}, errors{0};
std::unordered_map<void *, size_t> *sizes = nullptr;
size_t budget = (size_t) 256 << 20;
}
static hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static hipError_t hipGetLastError() { return hipSuccess; }
static const char *hipGetErrorString(hipError_t) { return "error"; }
static hipError_t hipDeviceSynchronize() { return hipSuccess; }
static hipError_t hipMalloc(void **p, size_t n) {
    std::lock_guard<std::mutex> g(fake::m);
    if (!fake::sizes) fake::sizes = new std::unordered_map<void *, size_t>();
    if (fake::bytes + n > fake::budget) { fake::fails++; return hipErrorOutOfMemory; }
    *p = malloc(n);
    fake::live.insert(*p); (*fake::sizes)[*p] = n; fake::bytes += n;
    return hipSuccess;
}
static hipError_t hipFree(void *p) {
    std::lock_guard<std::mutex> g(fake::m);
    if (!fake::live.erase(p)) { fake::errors++; fprintf(stderr, "hipFree of a pointer that is not allocated: %p\n", p); return 1; }
    fake::bytes -= (*fake::sizes)[p]; fake::sizes->erase(p);
    free(p);
    return hipSuccess;
}
static hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }     // (ASan: a stale size writes out of bounds; an unmapped arena range faults)
// ---- the virtual-memory calls of the arenas: addresses are a PROT_NONE mapping, "device memory" is the budget above, hipMemMap makes
// the range writable and hipMemUnmap takes that away again - a block handed out over a hole, or used after its chunk went back, faults
#include <sys/mman.h>
static const hipError_t hipErrorInvalidValue = 1;
struct hipMemLocation { int type, id; };
struct hipMemAllocationProp { int type; hipMemLocation location; };
struct hipMemAccessDesc { hipMemLocation location; int flags; };
typedef struct FakeHandle { size_t size; bool mapped; } *hipMemGenericAllocationHandle_t;
static const int hipMemAllocationTypePinned = 1, hipMemLocationTypeDevice = 1, hipMemAccessFlagsProtReadWrite = 3;
namespace fake { std::atomic<size_t> reservations{0}, handles{0}; }
static hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *tot = fake::budget; *fr = fake::budget - fake::bytes; return hipSuccess; }
static hipError_t hipMemAddressReserve(void **p, size_t n, size_t, void *, unsigned long long) {
    void *q = mmap(nullptr, n, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (q == MAP_FAILED) return hipErrorOutOfMemory;
    *p = q; fake::reservations++;
    return hipSuccess;
}
static hipError_t hipMemAddressFree(void *p, size_t n) { if (munmap(p, n) != 0) { fake::errors++; return hipErrorInvalidValue; } fake::reservations--; return hipSuccess; }
static hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *h, size_t n, const hipMemAllocationProp *, unsigned long long) {
    std::lock_guard<std::mutex> g(fake::m);
    if (fake::bytes + n > fake::budget) { fake::fails++; return hipErrorOutOfMemory; }
    fake::bytes += n; fake::handles++;
    *h = new FakeHandle{n, false};
    return hipSuccess;
}
static hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) {
    std::lock_guard<std::mutex> g(fake::m);
    fake::bytes -= h->size; fake::handles--;
    delete h;
    return hipSuccess;
}
static hipError_t hipMemMap(void *p, size_t n, size_t, hipMemGenericAllocationHandle_t h, unsigned long long) {
    if (n != h->size || h->mapped) { fake::errors++; return hipErrorInvalidValue; }
    h->mapped = true;
    return mprotect(p, n, PROT_READ | PROT_WRITE) == 0 ? hipSuccess : hipErrorInvalidValue;
}
static hipError_t hipMemSetAccess(void *, size_t, const hipMemAccessDesc *, size_t) { return hipSuccess; }
static hipError_t hipMemUnmap(void *p, size_t n) {
    if (mprotect(p, n, PROT_NONE) != 0) { fake::errors++; return hipErrorInvalidValue; }
    if (madvise(p, n, MADV_DONTNEED) != 0) fake::errors++;
    return hipSuccess;
}
#include <cstring>
#define CDM_POOL_LARGE_CHUNK ((size_t) 2 << 20)
#define CDM_POOL_SMALL_CHUNK ((size_t) 256 << 10)
#define CDM_POOL_ROOMY_MIN ((size_t) 4 << 20)
#include "../../carpedeam_amd/csrc/pool.h"

// every block carries its own tag at both ends while it is in use: two blocks that overlap, or a block handed out twice, lose one
static void tagBlock(void *p, size_t n) { const uint64_t t = (uint64_t) (uintptr_t) p ^ (n * 0x9E3779B97F4A7C15ull); if (n >= 16) { memcpy(p, &t, 8); memcpy((char *) p + n - 8, &t, 8); } }
static void checkBlock(void *p, size_t n) {
    const uint64_t t = (uint64_t) (uintptr_t) p ^ (n * 0x9E3779B97F4A7C15ull); uint64_t a = t, b = t;
    if (n >= 16) { memcpy(&a, p, 8); memcpy(&b, (char *) p + n - 8, 8); }
    if (a != t || b != t) { fake::errors++; fprintf(stderr, "block %p + %zu was written over while in use\n", p, n); }
}
// The pattern that lost address space until round 5: a large working set, a small block allocated behind it, the working set freed, the
// arena trimmed - the chunks in front of the pinned block became holes that nothing mapped again, and after a few rounds the arena's
// reservation (twice the device) was used up with almost nothing in use.
static int pinAndTrim(int rounds) {
    std::vector<void *> pins;
    for (int r = 0; r < rounds; r++) {
        std::vector<void *> work;
        for (int i = 0; i < 20; i++) { void *p = nullptr; if (cdmpool::allocate(&p, (size_t) 4 << 20) != hipSuccess) { fprintf(stderr, "pin and trim: round %d: the working set was refused (%zu bytes in use of %zu)\n", r, fake::bytes.load(), fake::budget); return 1; } work.push_back(p); }
        void *pin = nullptr;
        if (cdmpool::allocate(&pin, (size_t) 1 << 20) != hipSuccess) { fprintf(stderr, "pin and trim: round %d: the pinned block was refused\n", r); return 1; }
        pins.push_back(pin);
        for (void *p : work) cdmpool::release(p);
        cdmpool::trimMine();
        if (pins.size() > 24) { cdmpool::release(pins.front()); pins.erase(pins.begin()); }
    }
    for (void *p : pins) cdmpool::release(p);
    cdmpool::trimMine();
    return 0;
}
int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 8, rounds = argc > 2 ? atoi(argv[2]) : 60;
    setenv("CDM_POOL_POISON", "0xA5", 1);           // every block handed out is written over its whole recorded size
    setenv("CDM_SOMETHING", "x", 1);
    if (argc > 3) setenv("CDM_POOL", argv[3], 1);       // "blocks": the exact-size cache; default: the arenas
    if (pinAndTrim(200)) return 1;
    std::mutex qm; std::vector<std::pair<void *, size_t>> handoff;      // blocks freed by another thread than their allocator's
    std::atomic<size_t> oom{0}, allocs{0};
    for (int round = 0; round < rounds; round++) {
        // short-lived threads: every round ends with thread exits (the caches' thread_local destructors) while others still run
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; t++) ts.emplace_back([&, t, round] {
            std::mt19937_64 rng((uint64_t) round * 1000 + t);
            std::vector<std::pair<void *, size_t>> mine;
            const int steps = 200 + (int) (rng() % 200);
            for (int i = 0; i < steps; i++) {
                const unsigned op = (unsigned) (rng() % 100);
                if (op < 45 || mine.empty()) {
                    const size_t n = (rng() % 64 == 0) ? ((size_t) 4 << 20) + (rng() % (1 << 19)) : (rng() % 16 == 0) ? (size_t) (rng() % (1 << 20)) + 1 : (size_t) (rng() % (1 << 16)) + 1;      // (4 MB and more: the head-room path of this build; up to 1 MB: both arenas)
                    void *p = nullptr;
                    if (cdmpool::allocate(&p, n) == hipSuccess) { memset(p, 0x5A, n < 4096 ? n : 4096); tagBlock(p, n); mine.emplace_back(p, n); allocs++; } else oom++;
                } else if (op < 80) {
                    const size_t k = rng() % mine.size();
                    checkBlock(mine[k].first, mine[k].second); cdmpool::release(mine[k].first); mine[k] = mine.back(); mine.pop_back();
                } else if (op < 90) {
                    const size_t k = rng() % mine.size();
                    { std::lock_guard<std::mutex> g(qm); handoff.push_back(mine[k]); }
                    mine[k] = mine.back(); mine.pop_back();
                } else if (op < 97) {
                    std::pair<void *, size_t> b{nullptr, 0};
                    { std::lock_guard<std::mutex> g(qm); if (!handoff.empty()) { b = handoff.back(); handoff.pop_back(); } }
                    if (b.first) { checkBlock(b.first, b.second); cdmpool::release(b.first); }
                } else if (op == 97) cdmpool::headroom().store(1.0f + (float) (rng() % 3) * 0.3f);
                else if (op == 98) cdmpool::trimMine();
                else (void) cdmenv::get("CDM_SOMETHING"), (void) cdmenv::refresh();
            }
            // half of the threads leave blocks in use behind (handed to the queue): their owner is gone when they are freed
            for (auto &b : mine) { if (t & 1) { std::lock_guard<std::mutex> g(qm); handoff.push_back(b); } else { checkBlock(b.first, b.second); cdmpool::release(b.first); } }
        });
        for (auto &t : ts) t.join();
    }
    for (auto &b : handoff) { checkBlock(b.first, b.second); cdmpool::release(b.first); }
    cdmpool::trimAll();
    size_t registered; { std::lock_guard<std::mutex> g(cdmpool::registry().m); registered = cdmpool::registry().blocks.size(); }
    size_t pools, ranges; { std::lock_guard<std::mutex> g(cdmpool::registry().m); pools = cdmpool::registry().pools.size(); ranges = cdmpool::registry().ranges.size(); }
    printf("pool stress: %d threads x %d rounds, %zu allocations, %zu refused for lack of memory (%zu failing driver calls), %zu bytes / %zu blocks left on the device, %zu registered, %zu errors\n",
           threads, rounds, allocs.load(), oom.load(), fake::fails.load(), fake::bytes.load(), fake::live.size() + fake::handles.load(), registered, fake::errors.load());
    // (pools outlive their threads - a thread's exit makes no driver call - and are adopted by later threads: at most one per thread that
    //  ever ran at the same time, + the main thread's)
    printf("pools: %zu (at most %d), address ranges: %zu, reservations: %zu\n", pools, threads + 1, ranges, fake::reservations.load());
    return (fake::errors || fake::bytes || !fake::live.empty() || fake::handles || registered || pools > (size_t) threads + 1 || ranges > 2 * ((size_t) threads + 1) || fake::reservations != ranges) ? 1 : 0;
}
