// sink_probe2.cpp -- the output pipeline that avoids page faults beside fallocate(): one thread instantiates the
// pages (fallocate, 256-MB steps), a second maps them (MADV_POPULATE_WRITE) one step behind, T threads then only
// memcpy.  g++ -O2 -o tools/sink_probe2 tools/sink_probe2.cpp -lpthread
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const size_t G = (size_t)(argc > 1 ? atof(argv[1]) * (1u << 30) : (8ull << 30));
    const std::string dir = argc > 2 ? argv[2] : "/dev/shm";
    const std::string fin = dir + "/sp2_in.bin", fout = dir + "/sp2_out.bin";
    {
        std::vector<char> buf(64u << 20, 'A');
        int fd = open(fin.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
        for (size_t o = 0; o < G; o += buf.size()) if (write(fd, buf.data(), buf.size()) < 0) perror("write");
        close(fd);
    }
    int fdi = open(fin.c_str(), O_RDONLY);
    char* in = (char*)mmap(nullptr, G, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fdi, 0);
    for (size_t STEP : {size_t(64u << 20), size_t(256u << 20)})
    for (int P : {1, 2})                     // populate threads
    for (int T : {8, 16}) {
        unlink(fout.c_str());
        int fd = open(fout.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644);
        const size_t nsteps = (G + STEP - 1) / STEP;
        double t0 = now();
        char* out = (char*)mmap(nullptr, G, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        std::mutex m; std::condition_variable cv;
        size_t allocated = 0;                        // steps fallocate'd
        std::vector<char> populated(nsteps, 0);
        double tf = 0, tp = 0;
        std::thread A([&] {
            for (size_t k = 0; k < nsteps; k++) {
                double a = now();
                if (fallocate(fd, 0, k * STEP, std::min(STEP, G - k * STEP))) perror("fallocate");
                tf += now() - a;
                { std::lock_guard<std::mutex> l(m); allocated = k + 1; }
                cv.notify_all();
            }
        });
        std::atomic<size_t> next_pop{0};
        std::vector<std::thread> B;
        std::atomic<long> tp_us{0};
        for (int p = 0; p < P; p++) B.emplace_back([&] {
            for (;;) {
                size_t k = next_pop++;
                if (k >= nsteps) break;
                { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return allocated > k; }); }
                double a = now();
                if (madvise(out + k * STEP, std::min(STEP, G - k * STEP), MADV_POPULATE_WRITE)) perror("populate");
                tp_us += (long)((now() - a) * 1e6);
                { std::lock_guard<std::mutex> l(m); populated[k] = 1; }
                cv.notify_all();
            }
        });
        std::atomic<size_t> next_fill{0};
        const size_t PIECE = 4u << 20;
        const size_t npieces = (G + PIECE - 1) / PIECE;
        std::vector<std::thread> F;
        for (int t = 0; t < T; t++) F.emplace_back([&] {
            for (;;) {
                size_t i = next_fill++;
                if (i >= npieces) break;
                size_t k = i * PIECE / STEP;
                { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return populated[k] != 0; }); }
                memcpy(out + i * PIECE, in + i * PIECE, std::min(PIECE, G - i * PIECE));
            }
        });
        A.join(); for (auto& x : B) x.join(); for (auto& x : F) x.join();
        double t1 = now();
        munmap(out, G); close(fd);
        tp = tp_us.load() / 1e6;
        printf("step %3zu MB, %d populate thread(s), %2d copy threads: %.2f GB/s total (fallocate ran at %.1f GB/s, populate at %.1f GB/s per thread; munmap+close %.3f s)\n",
               STEP >> 20, P, T, G / (t1 - t0) / 1e9, G / tf / 1e9, G / tp / 1e9 * 1, now() - t1);
    }
    unlink(fout.c_str()); unlink(fin.c_str());
}
