// How many cores does this job really get?  Independent busy loops on T threads: work per second for T = 1, 2, 4, ... (g++ -O2 -fopenmp)
#include <chrono>
#include <cstdio>
#include <initializer_list>
#include <omp.h>
int main() {
    for (int T : {1, 2, 4, 8, 12, 16, 24, 32, 64, 128}) {
        const auto t0 = std::chrono::steady_clock::now();
        double sink = 0;
#pragma omp parallel num_threads(T) reduction(+ : sink)
        {
            double x = 1.0 + omp_get_thread_num();
            for (long i = 0; i < 300000000L; i++) x = x * 1.0000001 + 1e-9;
            sink += x;
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %3d: %.2f s  -> %.1f thread-loops per second (x %.1f of one thread)  [%g]\n", T, s, T / s, 0.0, sink);
        fflush(stdout);
    }
}
