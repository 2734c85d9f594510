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
size_t budget = (size_t) 400 << 20;
}
static hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static hipError_t hipGetLastError() { return hipSuccess; }
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
static hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }     // (ASan: a stale size writes out of bounds)
#include <cstring>
#include "../../carpedeam_amd/csrc/pool.h"

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 8, rounds = argc > 2 ? atoi(argv[2]) : 60;
    setenv("CDM_POOL_POISON", "0xA5", 1);           // every block handed out is written over its whole recorded size
    setenv("CDM_SOMETHING", "x", 1);
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
                    const size_t n = (rng() % 64 == 0) ? ((size_t) 64 << 20) + (rng() % (8 << 20)) : (size_t) (rng() % (1 << 16)) + 1;      // (64 MB and more: the head-room path)
                    void *p = nullptr;
                    if (cdmpool::allocate(&p, n) == hipSuccess) { memset(p, 0x5A, n < 4096 ? n : 4096); mine.emplace_back(p, n); allocs++; } else oom++;
                } else if (op < 80) {
                    const size_t k = rng() % mine.size();
                    cdmpool::release(mine[k].first); mine[k] = mine.back(); mine.pop_back();
                } else if (op < 90) {
                    const size_t k = rng() % mine.size();
                    { std::lock_guard<std::mutex> g(qm); handoff.push_back(mine[k]); }
                    mine[k] = mine.back(); mine.pop_back();
                } else if (op < 97) {
                    std::pair<void *, size_t> b{nullptr, 0};
                    { std::lock_guard<std::mutex> g(qm); if (!handoff.empty()) { b = handoff.back(); handoff.pop_back(); } }
                    if (b.first) cdmpool::release(b.first);
                } else if (op == 97) cdmpool::headroom().store(1.0f + (float) (rng() % 3) * 0.3f);
                else if (op == 98) cdmpool::trimMine();
                else (void) cdmenv::get("CDM_SOMETHING"), (void) cdmenv::refresh();
            }
            // half of the threads leave blocks in use behind (handed to the queue): their owner is gone when they are freed
            for (auto &b : mine) { if (t & 1) { std::lock_guard<std::mutex> g(qm); handoff.push_back(b); } else cdmpool::release(b.first); }
        });
        for (auto &t : ts) t.join();
    }
    for (auto &b : handoff) cdmpool::release(b.first);
    cdmpool::trimAll();
    size_t registered; { std::lock_guard<std::mutex> g(cdmpool::registry().m); registered = cdmpool::registry().blocks.size(); }
    printf("pool stress: %d threads x %d rounds, %zu allocations, %zu refused for lack of memory (%zu failing hipMalloc calls), %zu bytes / %zu blocks left on the device, %zu registered, %zu errors\n",
           threads, rounds, allocs.load(), oom.load(), fake::fails.load(), fake::bytes.load(), fake::live.size(), registered, fake::errors.load());
    return (fake::errors || fake::bytes || !fake::live.empty() || registered) ? 1 : 0;
}
