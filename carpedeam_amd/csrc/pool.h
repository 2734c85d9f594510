// The library's device-memory cache and its snapshot of the CDM_* switches - host code only, included by api.hip.
// (A header of its own so that tests/cpu/pool_stress.cpp can compile exactly this code against stand-ins for hipMalloc / hipFree
// and run it under ThreadSanitizer / AddressSanitizer on a box without a GPU.)
#pragma once
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include <sys/syscall.h>
#include <unistd.h>

extern char **environ;

// ------------------------------------------------------------------------------------------------ the CDM_* switches
// Read ONCE per process (and again when cdm_env_refresh() is called: tests, A/B runs), never with getenv() on a call path: glibc's
// getenv walks `environ` without a lock, so a setenv anywhere else in the process (the HSA runtime, RCCL, libc10 and Python's
// os.environ all call it) while a rank thread is inside the library reads a freed array - the likely cause of the one segmentation
// fault seen in the ranks-as-threads harness, where one rank's thread initialised torch's device state while
// another was inside cdm_kmermatch_part with its ~16 getenv calls.
namespace cdmenv {
struct Snapshot { std::vector<std::pair<std::string, std::string>> vars; };
inline std::atomic<const Snapshot *> &current() { static std::atomic<const Snapshot *> p{nullptr}; return p; }
inline std::mutex &refreshLock() { static std::mutex m; return m; }
// copies the CDM_* and OMP_NUM_THREADS entries of `environ`; the previous snapshots stay alive (a reader may hold their strings)
inline const Snapshot *refresh() {
    static std::vector<const Snapshot *> *kept = new std::vector<const Snapshot *>();
    std::lock_guard<std::mutex> g(refreshLock());
    Snapshot *s = new Snapshot();
    kept->push_back(s);
    for (char **e = environ; e && *e; e++) {
        if (strncmp(*e, "CDM_", 4) != 0 && strncmp(*e, "OMP_NUM_THREADS=", 16) != 0) continue;
        const char *eq = strchr(*e, '=');
        if (!eq) continue;
        s->vars.emplace_back(std::string(*e, (size_t) (eq - *e)), std::string(eq + 1));
    }
    current().store(s, std::memory_order_release);
    return s;
}
inline const char *get(const char *name) {
    const Snapshot *s = current().load(std::memory_order_acquire);
    if (!s) s = refresh();
    for (const auto &kv : s->vars) if (kv.first == name) return kv.second.c_str();
    return nullptr;
}
}  // namespace cdmenv

// ------------------------------------------------------------------------------------------------ caching allocator
namespace cdmpool {
// One cache of free blocks per host thread and device: a context belongs to one host thread, and a block freed by
// one thread's stream must not be handed to another thread's stream without synchronisation (ranks as threads of one process in
// the tests).  Which block has which size and whose it is lives in ONE process-wide registry: a block freed by another thread than
// its allocator's is released and forgotten there, so no cache can meet its address again with a stale size.  The registry also
// lists the live caches, so that a thread that runs out of device memory can release what the OTHER threads have parked
// (a cache's own mutex guards its free list for that one cross-thread visitor; uncontended otherwise).
struct Pool {
    std::mutex m;
    std::multimap<size_t, void *> freeBlocks;
};
struct Registry {
    std::mutex m;
    std::unordered_map<void *, std::pair<size_t, Pool *>> blocks;
    std::vector<Pool *> pools;              // the caches of the threads that are alive
};
inline Registry &registry() { static Registry *r = new Registry(); return *r; }      // (never destroyed: thread_local pools may outlive statics)
// frees every cached block of `q`.  Lock order: registry, then pool.
inline void trimLocked(Registry &r, Pool &q) {
    std::lock_guard<std::mutex> g(q.m);
    for (auto &kv : q.freeBlocks) { r.blocks.erase(kv.second); (void) hipFree(kv.second); }
    q.freeBlocks.clear();
}
inline void trim(Pool &q) { Registry &r = registry(); std::lock_guard<std::mutex> g(r.m); trimLocked(r, q); }
inline void trimAll() { Registry &r = registry(); std::lock_guard<std::mutex> g(r.m); for (Pool *q : r.pools) trimLocked(r, *q); }
struct Pools {
    Pool p[64];
    Pools() { Registry &r = registry(); std::lock_guard<std::mutex> g(r.m); for (Pool &q : p) r.pools.push_back(&q); }
    // a thread that ends gives its cached blocks back and leaves the registry: the blocks it allocated that are still in use stay
    // registered with no owner (whoever frees them releases them), so that a later thread whose caches happen to get this
    // address is not taken for their owner.  (The main thread's end is the end of the process: the HIP runtime may be half-way
    // through its own tear-down by then, so nothing is freed there.)
    ~Pools() {
        Registry &r = registry();
        std::lock_guard<std::mutex> g(r.m);
        const bool mainThread = getpid() == (pid_t) syscall(SYS_gettid);
        for (Pool &q : p) {
            if (!mainThread) trimLocked(r, q);
            for (size_t i = 0; i < r.pools.size(); i++) if (r.pools[i] == &q) { r.pools[i] = r.pools.back(); r.pools.pop_back(); break; }
        }
        for (auto &kv : r.blocks) if (kv.second.second >= p && kv.second.second < p + 64) kv.second.second = nullptr;
    }
};
inline Pool &poolOf(int dev) { static thread_local Pools pools; return pools.p[dev & 63]; }

// Head room for workloads whose buffers GROW from call to call (the contig iterations of the workflow loop: sequences, tuples and
// records get ~1.5x longer per iteration, so no cached block ever fits the next request and every iteration maps tens of GB anew -
// which costs ~46 ms per GB on some hosts, 0.1-1.1 s per iteration at 1-2 M reads).  With a factor f > 1 a large block is allocated
// f times the request - where a cached block of at least half the size shows that the buffer grows - and a cached block up to that
// much larger than a request is taken: the next iteration's buffers fit the previous iteration's blocks.  Off (1) by default;
// `ancient_reads_loop` switches it on.  Process-wide, set from any thread.
inline std::atomic<float> &headroom() { static std::atomic<float> h{1.0f}; return h; }

inline int poisonByte() {
    // CDM_POOL_POISON=<byte>: every block handed out is filled with that byte first (tests: a kernel that reads what it never wrote
    // shows itself; fresh device memory is zero, a cached block holds its last owner's data)
    static const int poison = cdmenv::get("CDM_POOL_POISON") ? (int) strtol(cdmenv::get("CDM_POOL_POISON"), NULL, 0) & 0xFF : -1;
    return poison;
}
inline hipError_t allocate(void **p, size_t bytes) {
    int dev = 0; (void) hipGetDevice(&dev);
    Pool &pool = poolOf(dev);
    bytes = (bytes + 255) & ~(size_t) 255;
    if (bytes == 0) bytes = 256;
    const float hr = headroom().load(std::memory_order_relaxed);
    bool roomy = hr > 1.0f && bytes >= ((size_t) 64 << 20);
    const size_t take = roomy ? (size_t) ((double) bytes * hr * 1.125) : bytes + bytes / 8;
    const int poison = poisonByte();
    {
        std::unique_lock<std::mutex> g(pool.m);
        auto it = pool.freeBlocks.lower_bound(bytes);
        if (it != pool.freeBlocks.end() && it->first <= take) {
            *p = it->second; const size_t have = it->first; pool.freeBlocks.erase(it);
            g.unlock();
            if (poison >= 0) { (void) hipDeviceSynchronize(); (void) hipMemset(*p, poison, have); (void) hipDeviceSynchronize(); }
            return hipSuccess;
        }
        // head room only where growth shows: a cached block that just fails to hold the request (at least half its size) is the trace
        // of the same buffer one call earlier; a first allocation of its kind (the reads, a one-shot module) gets the exact size
        if (roomy) roomy = it != pool.freeBlocks.begin() && std::prev(it)->first >= bytes / 2;
    }
    auto record = [&](size_t size) { Registry &r = registry(); std::lock_guard<std::mutex> g(r.m); r.blocks[*p] = {size, &pool}; };
    if (roomy) {
        const size_t want = ((size_t) ((double) bytes * hr) + 255) & ~(size_t) 255;
        if (hipMalloc(p, want) == hipSuccess) {
            record(want);
            if (poison >= 0) { (void) hipMemset(*p, poison, want); (void) hipDeviceSynchronize(); }
            return hipSuccess;
        }
        (void) hipGetLastError();       // (no room for the head room: the exact size below)
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {   // out of memory with blocks parked in the caches - this thread's or another's: release them all and retry once
        (void) hipGetLastError();
        (void) hipDeviceSynchronize();       // (another thread's parked block may still be read by that thread's stream)
        trimAll();
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) record(bytes);
    if (e == hipSuccess && poison >= 0) { (void) hipMemset(*p, poison, bytes); (void) hipDeviceSynchronize(); }
    return e;
}
inline void release(void *p) {
    if (!p) return;
    int dev = 0; (void) hipGetDevice(&dev);
    Pool &pool = poolOf(dev);
    size_t bytes = 0;
    {
        Registry &r = registry();
        std::lock_guard<std::mutex> g(r.m);
        auto it = r.blocks.find(p);
        if (it != r.blocks.end() && it->second.second == &pool) bytes = it->second.first;
        else if (it != r.blocks.end()) r.blocks.erase(it);
    }
    if (bytes) { std::lock_guard<std::mutex> g(pool.m); pool.freeBlocks.emplace(bytes, p); }
    else (void) hipFree(p);
}
inline void trimMine() { int dev = 0; (void) hipGetDevice(&dev); trim(poolOf(dev)); }
}  // namespace cdmpool
