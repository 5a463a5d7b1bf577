// sink_probe.cpp -- can anything beat one write() stream into a single tmpfs file?  (threads vs processes,
// pinned to one NUMA node or not).  g++ -O2 -o tools/sink_probe tools/sink_probe.cpp -lpthread
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pin(int cpu) { cpu_set_t s; CPU_ZERO(&s); CPU_SET(cpu, &s); sched_setaffinity(0, sizeof s, &s); }
int main(int argc, char** argv)
{
    const size_t G = (size_t)(argc > 1 ? atof(argv[1]) * (1u << 30) : (4ull << 30));
    const std::string dir = argc > 2 ? argv[2] : "/dev/shm";
    const std::string fin = dir + "/sp_in.bin", fout = dir + "/sp_out.bin";
    const size_t CH = 64u << 20;
    {
        std::vector<char> buf(CH, 'A');
        pin(2);
        int fd = open(fin.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
        double t0 = now();
        for (size_t o = 0; o < G; o += CH) if (write(fd, buf.data(), CH) < 0) perror("write");
        close(fd);
        printf("write() pinned to cpu 2: %.2f GB/s\n", G / (now() - t0) / 1e9);
    }
    int fdi = open(fin.c_str(), O_RDONLY);
    char* in = (char*)mmap(nullptr, G, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fdi, 0);
    auto fresh = [&] { unlink(fout.c_str()); return open(fout.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644); };
    // processes filling a shared mapping (each its own mm), with and without fallocate first
    for (int pre = 0; pre < 2; pre++)
        for (int T : {8}) {
            int fd = fresh();
            double t0 = now();
            if (pre) { if (fallocate(fd, 0, 0, G)) perror("fallocate"); } else if (ftruncate(fd, G)) perror("ftruncate");
            double t1 = now();
            std::vector<pid_t> kids;
            for (int t = 0; t < T; t++) {
                pid_t p = fork();
                if (p == 0) {
                    pin(4 + 2 * t);
                    size_t a = (G / T * t) & ~size_t(4095), b = t == T - 1 ? G : (G / T * (t + 1)) & ~size_t(4095);
                    char* out = (char*)mmap(nullptr, b - a, PROT_READ | PROT_WRITE, MAP_SHARED, fd, a);
                    memcpy(out, in + a, b - a);
                    _exit(0);
                }
                kids.push_back(p);
            }
            for (pid_t p : kids) waitpid(p, nullptr, 0);
            close(fd);
            printf("%s + %2d PROCESSES filling a shared mapping: total %.2f GB/s (fill alone %.2f)\n", pre ? "fallocate" : "ftruncate", T,
                   G / (now() - t0) / 1e9, G / (now() - t1) / 1e9);
        }
    // threads pinned to node 0 cores
    for (int T : {4, 8, 16}) {
        int fd = fresh();
        if (fallocate(fd, 0, 0, G)) perror("fallocate");
        double t1 = now();
        char* out = (char*)mmap(nullptr, G, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t] {
            pin(4 + 2 * t);
            size_t a = (G / T * t) & ~size_t(4095), b = t == T - 1 ? G : (G / T * (t + 1)) & ~size_t(4095);
            memcpy(out + a, in + a, b - a);
        });
        for (auto& x : th) x.join();
        munmap(out, G); close(fd);
        printf("fallocate, then %2d pinned THREADS filling a shared mapping: fill %.2f GB/s\n", T, G / (now() - t1) / 1e9);
    }

    // which factor hurts: threads not pinned, or fallocate running beside the faults?
    for (int pinned = 0; pinned < 2; pinned++)
        for (int T : {8, 16}) {
            int fd = fresh();
            if (fallocate(fd, 0, 0, G)) perror("fallocate");
            double t1 = now();
            char* out = (char*)mmap(nullptr, G, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&, t] {
                if (pinned) pin(4 + 2 * t);
                size_t a = (G / T * t) & ~size_t(4095), b = t == T - 1 ? G : (G / T * (t + 1)) & ~size_t(4095);
                memcpy(out + a, in + a, b - a);
            });
            for (auto& x : th) x.join();
            munmap(out, G); close(fd);
            printf("fallocate ALL first, %2d %s threads fill: %.2f GB/s\n", T, pinned ? "pinned" : "unpinned", G / (now() - t1) / 1e9);
        }
    for (int pinned = 0; pinned < 2; pinned++)
        for (int T : {8, 16}) {
            int fd = fresh();
            const size_t STEP = 256u << 20;
            double t0 = now(), tf = 0;
            char* out = (char*)mmap(nullptr, G, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);   // one mapping, beyond EOF at first
            std::vector<std::thread> prev;
            for (size_t o = 0; o < G; o += STEP) {
                const size_t n = std::min(STEP, G - o);
                double f0 = now();
                if (fallocate(fd, 0, o, n)) perror("fallocate");
                tf += now() - f0;
                for (auto& x : prev) x.join();
                prev.clear();
                for (int t = 0; t < T; t++) prev.emplace_back([=] {
                    if (pinned) pin(4 + 2 * t);
                    size_t a = n / T * t, b = t == T - 1 ? n : n / T * (t + 1);
                    memcpy(out + o + a, in + o + a, b - a);
                });
            }
            for (auto& x : prev) x.join();
            munmap(out, G); close(fd);
            printf("fallocate AHEAD of %2d %s threads (one mapping): %.2f GB/s total, fallocate alone ran at %.2f GB/s\n", T,
                   pinned ? "pinned" : "unpinned", G / (now() - t0) / 1e9, G / tf / 1e9);
        }
    // separate files from separate threads: does tmpfs allocation scale at all?
    for (int T : {2, 4, 8}) {
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t] {
            pin(4 + 2 * t);
            std::string f = fout + std::to_string(t);
            int fd = open(f.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
            for (size_t o = G / T * t; o < G / T * (t + 1); o += CH) if (write(fd, in + o, CH) < 0) perror("write");
            close(fd);
        });
        for (auto& x : th) x.join();
        printf("%d threads writing %d separate files: %.2f GB/s\n", T, T, G / (now() - t0) / 1e9);
        for (int t = 0; t < T; t++) unlink((fout + std::to_string(t)).c_str());
    }
    unlink(fout.c_str()); unlink(fin.c_str());
}
