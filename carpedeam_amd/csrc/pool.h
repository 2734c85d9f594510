// The library's device-memory cache and its snapshot of the CDM_* switches - host code only, included by api.hip.
// (A header of its own so that tests/cpu/pool_stress.cpp can compile exactly this code against stand-ins for hipMalloc / hipFree
// and run it under ThreadSanitizer / AddressSanitizer on a box without a GPU.)
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include <sys/syscall.h>
#include <time.h>
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

// ------------------------------------------------------------------------------------------------ device memory
namespace cdmpool {
// Device memory comes out of ARENAS: a range of virtual addresses reserved once (hipMemAddressReserve) whose mapped part grows at its
// end (hipMemCreate + hipMemMap), cut into blocks that are split on allocation (best fit) and merged with their free neighbours on
// release.  Why not hipMalloc per buffer with a cache of freed blocks (what this was until round 4; CDM_POOL=blocks still selects
// it): memory that went back to the driver once costs 33-58 ms per GB to get again on this platform (scripts/probes/vmm_probe.hip:
// fresh memory maps in 0-9 ms per GB, freed memory is cleared before it is handed out again), and a cache of exact-size blocks keeps
// every size a growing workload ever asked for - the contig iterations of the workflow loop (buffers 1.02-1.5x larger per iteration)
// had mapped 390 GB by their last iteration at 25 M reads, ran out of memory, released everything and paid 7 s to map 120 GB again.
// An arena never gives memory back while its thread lives (out-of-memory in ANOTHER thread's arena excepted: trimAll), and any free
// range serves any request.
//
// One pool (a small-request and a large-request arena) per host thread and device: a context belongs to one host thread
// (INTEGRATION.md), and a block freed on one thread's stream must not be handed to another thread's stream without synchronisation
// (ranks as threads of one process in the tests).  A block released by another thread than its owner's goes back to the OWNER's
// arena, after a device synchronise.  A thread that ends leaves its pool behind as an orphan, which the next thread that needs one for
// the device adopts (no call into the runtime from a thread's exit: see Pools).  The registry lists pools and address ranges, so that
// release() finds the owner of a pointer and a thread that runs out of memory can make the others give back their free chunks.
#ifndef CDM_POOL_LARGE_CHUNK
#define CDM_POOL_LARGE_CHUNK ((size_t) 256 << 20)   // the large arena grows in chunks of this size,
#endif
#ifndef CDM_POOL_SMALL_CHUNK
#define CDM_POOL_SMALL_CHUNK ((size_t) 32 << 20)    // the small one in chunks of this
#endif
#ifndef CDM_POOL_ROOMY_MIN
#define CDM_POOL_ROOMY_MIN ((size_t) 64 << 20)      // head room (cdm_pool_headroom) applies to requests of this size and more
#endif
constexpr size_t RUNTIME_RESERVE = (size_t) 1 << 30;      // device memory the pool leaves alone (growArena)
constexpr size_t LARGE_CHUNK = CDM_POOL_LARGE_CHUNK, SMALL_CHUNK = CDM_POOL_SMALL_CHUNK, ROOMY_MIN = CDM_POOL_ROOMY_MIN;
constexpr size_t SMALL_MAX = (size_t) 256 << 10;    // requests below this come from the small arena (long-lived odds and ends do not cut up the large one)
enum { B_USED = 0, B_FREE = 1, B_HOLE = 2 };        // (a hole: a free range whose chunk was given back - addresses without memory)
struct Blk { size_t size; int state; };
struct Chunk { size_t off, size; hipMemGenericAllocationHandle_t h; };
struct Arena {
    char *base = nullptr; size_t reserved = 0, end = 0;       // the blocks tile [0, end)
    std::map<size_t, Blk> blocks;                             // by offset
    std::multimap<size_t, size_t> freeBySize;                 // size -> offset of the free blocks
    std::vector<Chunk> chunks;
    size_t used = 0;                                          // blocks handed out
    bool small = false;
};
struct Pool {
    std::mutex m;
    int device = 0;
    Arena small, large;
    bool orphan = false;                                      // its thread ended with blocks still in use
    std::multimap<size_t, void *> freeBlocks;                 // CDM_POOL=blocks: the cache of exact-size blocks
};
struct Range { char *lo, *hi; Pool *pool; Arena *arena; };
struct Registry {
    std::mutex m;
    std::unordered_map<void *, std::pair<size_t, Pool *>> blocks;      // CDM_POOL=blocks: size and owner of every block
    std::vector<Pool *> pools;              // the pools of the threads that are alive, and the orphans
    std::vector<Range> ranges;              // the arenas' address ranges
};
inline Registry &registry() { static Registry *r = new Registry(); return *r; }      // (never destroyed: thread_local pools may outlive statics)

// What the allocator cost and saved, process-wide (cdm_pool_stats): requests, requests served without the driver, calls that asked
// the driver for memory (hipMemCreate / hipMalloc), their bytes and the seconds they took, times free memory was given back after an
// out-of-memory.
struct Stats { std::atomic<unsigned long long> requests{0}, cached{0}, mallocs{0}, mallocBytes{0}, mallocNs{0}, trims{0}; };
inline Stats &stats() { static Stats s; return s; }
struct DriverTimer {
    struct timespec t0; size_t bytes;
    explicit DriverTimer(size_t b) : bytes(b) { clock_gettime(CLOCK_MONOTONIC, &t0); }
    void done(bool ok) {
        struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        Stats &st = stats();
        st.mallocs.fetch_add(1, std::memory_order_relaxed);
        if (ok) st.mallocBytes.fetch_add(bytes, std::memory_order_relaxed);
        st.mallocNs.fetch_add((unsigned long long) ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec)), std::memory_order_relaxed);
    }
};
// The virtual-memory calls of the driver are made by one thread at a time (ranks as threads of one process each grow an arena of their
// own; the calls are rare, and nothing says the runtime's bookkeeping of reservations and mappings takes concurrent callers).
inline std::mutex &driverLockMutex() { static std::mutex *m = new std::mutex(); return *m; }
struct DriverGuard {        // CDM_POOL_DRIVER_LOCK=0 (A/B, scripts/stress_threads.py): no serialisation
    bool on;
    DriverGuard() : on(!(cdmenv::get("CDM_POOL_DRIVER_LOCK") && cdmenv::get("CDM_POOL_DRIVER_LOCK")[0] == '0')) { if (on) driverLockMutex().lock(); }
    ~DriverGuard() { if (on) driverLockMutex().unlock(); }
};
inline hipError_t timedMalloc(void **p, size_t bytes) { DriverTimer t(bytes); const hipError_t e = hipMalloc(p, bytes); t.done(e == hipSuccess); return e; }

// ---- arena bookkeeping (the pool's mutex is held)
inline void dropFree(Arena &a, size_t off, size_t size) {
    auto r = a.freeBySize.equal_range(size);
    for (auto it = r.first; it != r.second; ++it) if (it->second == off) { a.freeBySize.erase(it); return; }
}
inline void addFree(Arena &a, size_t off, size_t size) { a.blocks[off] = Blk{size, B_FREE}; a.freeBySize.emplace(size, off); }
// takes `keep` bytes (at least `bytes`, more where head room is wanted) out of the smallest free block that holds `bytes`
inline void *takeBlock(Arena &a, size_t bytes, size_t keep) {
    auto it = a.freeBySize.lower_bound(bytes);
    if (it == a.freeBySize.end()) return nullptr;
    const size_t size = it->first, off = it->second;
    a.freeBySize.erase(it);
    const size_t mine = size > keep ? keep : size;            // (a block between `bytes` and `keep` is taken whole: that is its head room)
    a.blocks[off] = Blk{mine, B_USED};
    if (size > mine) addFree(a, off + mine, size - mine);
    a.used++;
    return a.base + off;
}
inline void giveBlock(Arena &a, size_t off) {
    auto it = a.blocks.find(off);
    size_t at = off, size = it->second.size;
    auto nx = std::next(it);
    if (nx != a.blocks.end() && nx->second.state == B_FREE) { dropFree(a, nx->first, nx->second.size); size += nx->second.size; a.blocks.erase(nx); }
    if (it != a.blocks.begin()) {
        auto pv = std::prev(it);
        if (pv->second.state == B_FREE) { dropFree(a, pv->first, pv->second.size); at = pv->first; size += pv->second.size; a.blocks.erase(it); it = pv; }
    }
    (void) it;
    a.blocks[at] = Blk{size, B_FREE};
    a.freeBySize.emplace(size, at);
    a.used--;
}
// maps more memory at the arena's end until the free block there holds `need` bytes.  Chunks of ONE size per arena: that is what
// hipMemSetAccess takes on this platform - chunks of mixed sizes in one reservation fail with "invalid argument"
// (scripts/probes/vmm_sizes.hip, vmm_uniform.hip).  What was mapped before a failure stays (a free block at the end).
inline hipError_t growArena(Arena &a, int device, size_t need) {
    const size_t chunk = a.small ? SMALL_CHUNK : LARGE_CHUNK;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    for (;;) {
        if (a.freeBySize.lower_bound(need) != a.freeBySize.end()) return hipSuccess;      // (some free block holds the request - at the end, or around a hole that got its memory back)
        // Where the next chunk goes: into a HOLE if there is one (a chunk given back by trimArena: its addresses still belong to the arena -
        // round 4 only ever mapped at the end, so every trim with a live block behind the freed chunks lost that much of the
        // reservation for good, and after a few trims allocate() reported out of memory with most of the device free), the hole
        // next to the largest free neighbourhood first; at the arena's end otherwise.
        size_t at = a.end; bool hole = false;
        {
            size_t best = 0;
            for (auto it = a.blocks.begin(); it != a.blocks.end(); ++it) {
                if (it->second.state != B_HOLE) continue;
                size_t around = 1;       // (any hole beats the end)
                if (it != a.blocks.begin() && std::prev(it)->second.state == B_FREE) around += std::prev(it)->second.size;
                if (std::next(it) != a.blocks.end() && std::next(it)->second.state == B_FREE) around += std::next(it)->second.size;
                if (around > best) { best = around; at = it->first; hole = true; }
            }
        }
        if (!hole && a.end + chunk > a.reserved) return hipErrorOutOfMemory;
        // The driver hands out MORE than the device holds - hipMemCreate kept succeeding at 307 GB mapped on a 288 GB MI355X (the 25 M-read
        // workflow's ninth iteration, round 5), 3.3 s in the calls, and kmermatcher then ran three times slower on whatever backed the
        // excess.  So the pool asks first: below a chunk plus what the runtime needs for its own launches, this is out of memory - the
        // caller trims the arenas' free chunks and tries again.
        {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) (void) hipGetLastError();
            else if (fr < chunk + std::min(RUNTIME_RESERVE, tot / 64)) {
                if (cdmenv::get("CDM_POOL_DEBUG")) fprintf(stderr, "carpedeam pool: %zu bytes free on the device, a chunk of %zu is not mapped\n", fr, chunk);
                return hipErrorOutOfMemory;
            }
        }
        hipMemGenericAllocationHandle_t h;
        DriverTimer t(chunk);
        hipError_t e;
        {
        DriverGuard dl;
        e = hipMemCreate(&h, chunk, &prop, 0);
        if (e == hipSuccess) {
            e = hipMemMap(a.base + at, chunk, 0, h, 0);
            if (e == hipSuccess) {
                hipMemAccessDesc d = {}; d.location = prop.location; d.flags = hipMemAccessFlagsProtReadWrite;
                e = hipMemSetAccess(a.base + at, chunk, &d, 1);
                if (e != hipSuccess) (void) hipMemUnmap(a.base + at, chunk);
            }
            if (e != hipSuccess) (void) hipMemRelease(h);
        }
        }
        t.done(e == hipSuccess);
        if (e != hipSuccess) {
            if (cdmenv::get("CDM_POOL_DEBUG")) fprintf(stderr, "carpedeam pool: mapping %zu bytes at offset %zu of the %s arena failed: %s\n", chunk, at, a.small ? "small" : "large", hipGetErrorString(e));
            // the caller gets the code; the runtime's per-thread "last error" must not keep it - the request may yet be served (without its
            // head room, or after a trim), and the next CDM_LAUNCH_CHECK of this thread would report "kernel launch failed: out of
            // memory" for a launch that went fine (the 25 M-read workflow's ninth iteration, round 5)
            (void) hipGetLastError();
            return e;
        }
        a.chunks.push_back(Chunk{at, chunk, h});
        if (hole) { a.blocks[at].state = B_USED; a.used++; giveBlock(a, at); }       // (a hole is one chunk: it becomes a free block, merged with its free neighbours)
        else {
            const bool tail = !a.blocks.empty() && a.blocks.rbegin()->second.state == B_FREE;
            if (tail) {
                auto last = std::prev(a.blocks.end());
                dropFree(a, last->first, last->second.size);
                last->second.size += chunk;
                a.freeBySize.emplace(last->second.size, last->first);
            } else addFree(a, a.end, chunk);
            a.end += chunk;
        }
    }
}
// gives back the chunks that lie in free blocks (their addresses become holes; holes at the end are cut off)
inline void trimArena(Arena &a) {
    for (size_t c = 0; c < a.chunks.size();) {
        const Chunk ch = a.chunks[c];
        auto it = a.blocks.upper_bound(ch.off);
        if (it == a.blocks.begin()) { c++; continue; }
        --it;
        const size_t bo = it->first, bs = it->second.size;
        if (it->second.state != B_FREE || bo + bs < ch.off + ch.size) { c++; continue; }
        { DriverGuard dl; (void) hipMemUnmap(a.base + ch.off, ch.size); (void) hipMemRelease(ch.h); }
        dropFree(a, bo, bs);
        a.blocks.erase(it);
        if (ch.off > bo) addFree(a, bo, ch.off - bo);
        a.blocks[ch.off] = Blk{ch.size, B_HOLE};
        if (bo + bs > ch.off + ch.size) addFree(a, ch.off + ch.size, bo + bs - (ch.off + ch.size));
        a.chunks[c] = a.chunks.back(); a.chunks.pop_back();
    }
    while (!a.blocks.empty() && a.blocks.rbegin()->second.state == B_HOLE) { a.end -= a.blocks.rbegin()->second.size; a.blocks.erase(std::prev(a.blocks.end())); }
}
// CDM_POOL=blocks, or a platform without the virtual-memory calls (found out at the first reservation): hipMalloc per block
inline std::atomic<int> &scheme() { static std::atomic<int> s{-1}; return s; }       // 0 blocks, 1 arenas
inline bool useArenas() {
    int v = scheme().load(std::memory_order_relaxed);
    if (v < 0) { const char *e = cdmenv::get("CDM_POOL"); v = (e && !strcmp(e, "blocks")) ? 0 : 1; scheme().store(v, std::memory_order_relaxed); }
    return v == 1;
}
inline bool ensureArena(Pool &pool, Arena &a, bool small) {
    if (a.base) return true;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || tot == 0) { (void) hipGetLastError(); return false; }
    const size_t want = small ? 512 * SMALL_CHUNK : (2 * tot + LARGE_CHUNK - 1) / LARGE_CHUNK * LARGE_CHUNK;       // (addresses, not memory)
    void *base = nullptr;
    hipError_t er;
    { DriverGuard dl; er = hipMemAddressReserve(&base, want, (size_t) 2 << 20, nullptr, 0); }
    if (er != hipSuccess || !base) {
        if (cdmenv::get("CDM_POOL_DEBUG")) fprintf(stderr, "carpedeam pool: reserving %zu bytes of addresses failed: %s\n", want, hipGetErrorString(er));
        (void) hipGetLastError(); return false;
    }
    a.base = (char *) base; a.reserved = want; a.end = 0; a.small = small;
    Registry &r = registry();
    std::lock_guard<std::mutex> g(r.m);
    r.ranges.push_back(Range{a.base, a.base + want, &pool, &a});
    return true;
}

// frees every cached block of `q` (CDM_POOL=blocks) and gives back the free chunks of its arenas.  Lock order: registry, then pool.
// syncOwner: `q` may belong to ANOTHER thread (the out-of-memory path): a range that thread released is free in the bookkeeping while
// its stream may still work on it (release() is stream-ordered for the owner), so the device is synchronised with the pool's lock held
// - the owner can neither release nor allocate in between - before anything is unmapped.
inline void trimLocked(Registry &r, Pool &q, bool syncOwner) {
    std::lock_guard<std::mutex> g(q.m);
    if (syncOwner) (void) hipDeviceSynchronize();
    for (auto &kv : q.freeBlocks) { r.blocks.erase(kv.second); (void) hipFree(kv.second); }
    q.freeBlocks.clear();
    trimArena(q.small); trimArena(q.large);
}
inline void trim(Pool &q) { Registry &r = registry(); std::lock_guard<std::mutex> g(r.m); trimLocked(r, q, false); }
// out of memory on device `dev` (the calling thread's current device): the pools of THAT device give back what they do not use - another
// device's free chunks are no help, and its streams are none of this thread's business.  dev < 0: every pool (process teardown, tests).
inline void trimAll(int dev = -1) {
    Registry &r = registry(); std::lock_guard<std::mutex> g(r.m);
    for (Pool *q : r.pools) if (dev < 0 || q->device == dev) trimLocked(r, *q, dev >= 0);
}
struct Pools {
    Pool *p[64] = {};
    // A thread that ends makes NO call into the HIP runtime here: this destructor runs among the thread's other thread-local
    // destructors, the runtime's own per-thread state may be gone already, and hipMemUnmap / hipFree from here crashed inside
    // libamdhip64 (scripts/stress_threads.py: short-lived threads, a segmentation fault within seconds; the likely second cause of the
    // harness crash of round 3, whose block cache called hipFree from this very place).  The thread's pools become ORPHANS - mapped
    // memory, free lists and all - and the next thread that needs a pool for the device adopts one (poolOf): nothing is unmapped,
    // nothing is mapped again.  Blocks of the dead thread that others still hold are freed into the pool whoever owns it by then.
    ~Pools() {
        Registry &r = registry();
        std::lock_guard<std::mutex> g(r.m);
        for (Pool *q : p) if (q) { std::lock_guard<std::mutex> g2(q->m); q->orphan = true; }
    }
};
inline Pool &poolOf(int dev) {
    static thread_local Pools pools;
    Pool *&q = pools.p[dev & 63];
    if (!q) {
        bool adopted = false;
        {
            Registry &r = registry();
            std::lock_guard<std::mutex> g(r.m);
            for (Pool *o : r.pools) {
                std::lock_guard<std::mutex> g2(o->m);
                if (o->orphan && o->device == dev) { o->orphan = false; q = o; adopted = true; break; }
            }
            if (!q) { q = new Pool(); q->device = dev; r.pools.push_back(q); }
        }
        if (adopted) (void) hipDeviceSynchronize();      // (whatever the previous owner's stream still had in flight on its free ranges)
    }
    return *q;
}

// Head room for workloads whose buffers GROW from call to call (the contig iterations of the workflow loop: sequences, tuples and
// records get 1.02-1.5x longer per iteration).  With a factor f > 1 a large block is handed out f times the request - where a free
// block of at least half the size shows that the buffer grows - and a free block up to that much larger than a request is taken
// whole: the next iteration's buffer fits where this one's was, instead of moving to the arena's end every time.  Off (1) by
// default; `ancient_reads_loop` switches it on.  Process-wide, set from any thread.
inline std::atomic<float> &headroom() { static std::atomic<float> h{1.0f}; return h; }

inline int poisonByte() {
    // CDM_POOL_POISON=<byte>: every block handed out is filled with that byte first (tests: a kernel that reads what it never wrote
    // shows itself; fresh device memory is zero, a reused range holds its last owner's data)
    static const int poison = cdmenv::get("CDM_POOL_POISON") ? (int) strtol(cdmenv::get("CDM_POOL_POISON"), NULL, 0) & 0xFF : -1;
    return poison;
}
inline hipError_t allocateBlocks(Pool &pool, void **p, size_t bytes);
inline hipError_t allocate(void **p, size_t bytes) {
    int dev = 0; (void) hipGetDevice(&dev);
    Pool &pool = poolOf(dev);
    stats().requests.fetch_add(1, std::memory_order_relaxed);
    if (!useArenas()) return allocateBlocks(pool, p, bytes);
    const bool small = bytes < SMALL_MAX;
    const size_t align = small ? 256 : (size_t) 4096;
    bytes = (bytes + align - 1) / align * align;
    if (bytes == 0) bytes = align;
    const float hr = headroom().load(std::memory_order_relaxed);
    const int poison = poisonByte();
    size_t got = 0;
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < 2; attempt++) {
        {
            std::unique_lock<std::mutex> g(pool.m);
            Arena &a = small ? pool.small : pool.large;
            if (!a.base) {
                g.unlock();
                const bool ok = ensureArena(pool, a, small);
                if (!ok) {      // no virtual-memory calls here: exact-size blocks from now on (said once)
                    if (scheme().exchange(0) != 0) fprintf(stderr, "carpedeam: hipMemAddressReserve is not available on this device; device memory comes from hipMalloc per buffer\n");
                    return allocateBlocks(pool, p, bytes);
                }
                g.lock();
            }
            bool roomy = hr > 1.0f && bytes >= ROOMY_MIN;
            if (roomy) {        // head room only where growth shows: a free block that just fails to hold the request
                auto it = a.freeBySize.lower_bound(bytes);
                roomy = it != a.freeBySize.begin() && std::prev(it)->first >= bytes / 2;
                if (!roomy && it != a.freeBySize.end() && it->first <= (size_t) ((double) bytes * hr * 1.125)) roomy = true;     // (a block with head room: kept whole)
            }
            const size_t keep = roomy ? ((size_t) ((double) bytes * hr) + align - 1) / align * align : bytes;
            void *q = takeBlock(a, bytes, keep);
            if (q) stats().cached.fetch_add(1, std::memory_order_relaxed);
            else {
                e = growArena(a, pool.device, keep);
                if (e != hipSuccess && keep > bytes) e = growArena(a, pool.device, bytes);       // (no room for the head room)
                if (e == hipSuccess) q = takeBlock(a, bytes, keep);
            }
            if (q) { *p = q; got = a.blocks[(size_t) ((char *) q - a.base)].size; break; }
        }
        if (attempt == 0) {     // out of memory: every thread's free chunks go back to the driver, then once more
            (void) hipGetLastError();
            trimAll(dev);       // (synchronises the device under every pool's lock)
            stats().trims.fetch_add(1, std::memory_order_relaxed);
        }
    }
    if (!got) {
        if (cdmenv::get("CDM_POOL_DEBUG")) {
            std::lock_guard<std::mutex> g(pool.m);
            const Arena &a = small ? pool.small : pool.large;
            size_t used = 0, fr = 0, hole = 0, largest = 0, nUsed = 0, nFree = 0;
            for (const auto &kv : a.blocks) {
                if (kv.second.state == B_USED) { used += kv.second.size; nUsed++; }
                else if (kv.second.state == B_FREE) { fr += kv.second.size; nFree++; largest = std::max(largest, kv.second.size); }
                else hole += kv.second.size;
            }
            fprintf(stderr, "carpedeam pool: no room for %zu bytes: arena of %zu bytes mapped, %zu in %zu blocks in use, %zu free in %zu blocks (largest %zu), %zu given back\n", bytes, a.end, used, nUsed, fr, nFree, largest, hole);
        }
        return e == hipSuccess ? hipErrorOutOfMemory : e;
    }
    if (poison >= 0) { (void) hipDeviceSynchronize(); (void) hipMemset(*p, poison, got); (void) hipDeviceSynchronize(); }
    return hipSuccess;
}
inline void releaseBlocks(Pool &pool, void *p);
inline void release(void *p) {
    if (!p) return;
    int dev = 0; (void) hipGetDevice(&dev);
    Pool &mine = poolOf(dev);
    Registry &r = registry();
    Pool *owner = nullptr; Arena *arena = nullptr;
    {
        std::lock_guard<std::mutex> g(r.m);
        for (const Range &x : r.ranges) if ((char *) p >= x.lo && (char *) p < x.hi) { owner = x.pool; arena = x.arena; break; }
    }
    if (!owner) { releaseBlocks(mine, p); return; }
    if (owner != &mine) (void) hipDeviceSynchronize();          // the owner's stream takes the range next: this thread's work on it must be over
    std::lock_guard<std::mutex> g(r.m);
    { std::lock_guard<std::mutex> g2(owner->m); giveBlock(*arena, (size_t) ((char *) p - arena->base)); }
}
inline void trimMine() { int dev = 0; (void) hipGetDevice(&dev); trim(poolOf(dev)); }

// ---- CDM_POOL=blocks: one hipMalloc per block, freed blocks cached by exact size per thread (the scheme of rounds 1-3)
inline hipError_t allocateBlocks(Pool &pool, void **p, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t) 255;
    if (bytes == 0) bytes = 256;
    const float hr = headroom().load(std::memory_order_relaxed);
    bool roomy = hr > 1.0f && bytes >= ROOMY_MIN;
    const size_t take = roomy ? (size_t) ((double) bytes * hr * 1.125) : bytes + bytes / 8;
    const int poison = poisonByte();
    {
        std::unique_lock<std::mutex> g(pool.m);
        auto it = pool.freeBlocks.lower_bound(bytes);
        if (it != pool.freeBlocks.end() && it->first <= take) {
            *p = it->second; const size_t have = it->first; pool.freeBlocks.erase(it);
            g.unlock();
            stats().cached.fetch_add(1, std::memory_order_relaxed);
            if (poison >= 0) { (void) hipDeviceSynchronize(); (void) hipMemset(*p, poison, have); (void) hipDeviceSynchronize(); }
            return hipSuccess;
        }
        if (roomy) roomy = it != pool.freeBlocks.begin() && std::prev(it)->first >= bytes / 2;
    }
    auto record = [&](size_t size) { Registry &r = registry(); std::lock_guard<std::mutex> g(r.m); r.blocks[*p] = {size, &pool}; };
    if (roomy) {
        const size_t want = ((size_t) ((double) bytes * hr) + 255) & ~(size_t) 255;
        if (timedMalloc(p, want) == hipSuccess) {
            record(want);
            if (poison >= 0) { (void) hipMemset(*p, poison, want); (void) hipDeviceSynchronize(); }
            return hipSuccess;
        }
        (void) hipGetLastError();       // (no room for the head room: the exact size below)
    }
    hipError_t e = timedMalloc(p, bytes);
    if (e != hipSuccess) {   // out of memory with blocks parked in the caches - this thread's or another's: release them all and retry once
        (void) hipGetLastError();
        trimAll(pool.device);
        stats().trims.fetch_add(1, std::memory_order_relaxed);
        e = timedMalloc(p, bytes);
    }
    if (e == hipSuccess) record(bytes);
    if (e == hipSuccess && poison >= 0) { (void) hipMemset(*p, poison, bytes); (void) hipDeviceSynchronize(); }
    return e;
}
inline void releaseBlocks(Pool &pool, void *p) {
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
}  // namespace cdmpool
