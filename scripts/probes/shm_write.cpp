// How fast can ONE file in /dev/shm (tmpfs) be written by 16 threads?  (round 5: a sequence DB's data file must stay one file - the
// workflow scripts test and link it by name - and pwrite() to one file is serialised by the inode lock: 2-2.5 GB/s at 5 GB)
//   g++ -O2 -fopenmp scripts/probes/shm_write.cpp -o scripts/probes/shm_write.bin && scripts/probes/shm_write.bin /dev/shm/x 5
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <omp.h>
#include <string>
#include <sys/mman.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const std::string path = argc > 1 ? argv[1] : "/dev/shm/cdm_probe";
    const size_t bytes = (size_t) (argc > 2 ? atof(argv[2]) : 5.0) * (1ull << 30);
    char *src = (char *) mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(src, bytes, MADV_HUGEPAGE);
#pragma omp parallel for
    for (size_t i = 0; i < bytes; i += 4096) src[i] = (char) i;
    const size_t BLOCK = 8u << 20, blocks = (bytes + BLOCK - 1) / BLOCK;
    for (int mode = 0; mode < 5; mode++) {
        unlink(path.c_str());
        const double t0 = now();
        const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        const char *what = "";
        if (mode == 0) {
            what = "pwrite, 8 MB blocks, all threads";
#pragma omp parallel for schedule(dynamic, 1)
            for (size_t b = 0; b < blocks; b++) { size_t at = b * BLOCK, n = std::min(BLOCK, bytes - at); while (n) { ssize_t w = pwrite(fd, src + at, n, at); if (w <= 0) break; at += w; n -= w; } }
        } else if (mode == 1 || mode == 2 || mode == 3) {
            what = mode == 1 ? "ftruncate + mmap(MAP_SHARED) + memcpy, all threads" : mode == 2 ? "fallocate + mmap + memcpy, all threads" : "ftruncate + mmap + MADV_POPULATE_WRITE per block + memcpy";
            if (mode == 2) { if (posix_fallocate(fd, 0, bytes) != 0) perror("fallocate"); } else if (ftruncate(fd, bytes) != 0) perror("ftruncate");
            char *dst = (char *) mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (dst == MAP_FAILED) { perror("mmap"); return 1; }
#pragma omp parallel for schedule(dynamic, 1)
            for (size_t b = 0; b < blocks; b++) {
                const size_t at = b * BLOCK, n = std::min(BLOCK, bytes - at);
#ifdef MADV_POPULATE_WRITE
                if (mode == 3) madvise(dst + at, n, MADV_POPULATE_WRITE);
#endif
                memcpy(dst + at, src + at, n);
            }
            munmap(dst, bytes);
        } else {
            what = "one pwrite from one thread";
            size_t at = 0, n = bytes; while (n) { ssize_t w = pwrite(fd, src + at, n, at); if (w <= 0) break; at += w; n -= w; }
        }
        close(fd);
        const double dt = now() - t0;
        printf("%-64s %.3f s  %.2f GB/s\n", what, dt, bytes / dt / 1e9); fflush(stdout);
    }
    unlink(path.c_str());
    return 0;
}
