// hostio_probe.cpp -- what the host side of the end-to-end path can move on this box (tmpfs in, tmpfs out, PCIe):
// the numbers DESIGN.md quotes for the writer / loader design come from here.
//   hipcc -O2 -o tools/hostio_probe tools/hostio_probe.cpp -lpthread ;  tools/hostio_probe [GiB] [dir]
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par(int T, const std::function<void(int)>& f)
{
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(f, t);
    f(0);
    for (auto& x : th) x.join();
}

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

int main(int argc, char** argv)
{
    const size_t G = (size_t)(argc > 1 ? atof(argv[1]) * (1u << 30) : (4ull << 30));
    const std::string dir = argc > 2 ? argv[2] : "/dev/shm";
    const std::string fin = dir + "/probe_in.bin", fout = dir + "/probe_out.bin";
    const size_t CH = 64u << 20;
    // ---- input file: single-thread write(), 64-MB calls
    std::vector<char> buf(CH);
    for (size_t i = 0; i < CH; i++) buf[i] = (i % 45001 == 45000) ? '\n' : "ACGT"[(i * 2654435761u >> 7) & 3];
    {
        unlink(fin.c_str());
        int fd = open(fin.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
        double t0 = now();
        for (size_t o = 0; o < G; o += CH) if (write(fd, buf.data(), CH) != (ssize_t)CH) { perror("write"); return 1; }
        close(fd);
        printf("write() 64MB calls, 1 thread, fresh file: %.2f GB/s\n", G / (now() - t0) / 1e9);
    }
    int fdi = open(fin.c_str(), O_RDONLY);
    char* in = (char*)mmap(nullptr, G, PROT_READ, MAP_PRIVATE, fdi, 0);
    // ---- reading the mapped input (first touch = minor faults) with T threads
    for (int T : {1, 8, 16, 32}) {
        munmap(in, G);
        in = (char*)mmap(nullptr, G, PROT_READ, MAP_PRIVATE, fdi, 0);
        std::atomic<size_t> nl{0};
        double t0 = now();
        par(T, [&](int t) {
            size_t a = G / T * t, b = t == T - 1 ? G : G / T * (t + 1), c = 0;
            const char* p = in + a;
            while (p < in + b) { const char* q = (const char*)memchr(p, '\n', in + b - p); if (!q) break; c++; p = q + 1; }
            nl += c;
        });
        printf("memchr over cold mmap of input, %2d threads: %.2f GB/s (%zu newlines)\n", T, G / (now() - t0) / 1e9, nl.load());
    }
    {
        double t0 = now();
        std::atomic<size_t> nl{0};
        par(8, [&](int t) {
            size_t a = G / 8 * t, b = t == 7 ? G : G / 8 * (t + 1), c = 0;
            const char* p = in + a;
            while (p < in + b) { const char* q = (const char*)memchr(p, '\n', in + b - p); if (!q) break; c++; p = q + 1; }
            nl += c;
        });
        printf("memchr over warm mmap of input,  8 threads: %.2f GB/s\n", G / (now() - t0) / 1e9);
    }
    // ---- output methods
    auto fresh = [&] { unlink(fout.c_str()); return open(fout.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0644); };
    {
        int fd = fresh();
        std::vector<iovec> iov;
        double t0 = now();
        for (size_t o = 0; o < G;) {
            iov.clear();
            for (int k = 0; k < 1000 && o < G; k++) { size_t n = std::min<size_t>(45000, G - o); iov.push_back({in + o, n}); o += n; }
            if (writev(fd, iov.data(), (int)iov.size()) < 0) { perror("writev"); return 1; }
        }
        close(fd);
        printf("writev 45KB pieces from the input mapping, 1 thread: %.2f GB/s\n", G / (now() - t0) / 1e9);
    }
    for (int T : {2, 4, 8, 16}) {
        int fd = fresh();
        if (ftruncate(fd, G)) perror("ftruncate");
        double t0 = now();
        par(T, [&](int t) {
            size_t a = G / T * t, b = t == T - 1 ? G : G / T * (t + 1);
            for (size_t o = a; o < b; o += 8u << 20) if (pwrite(fd, in + o, std::min<size_t>(8u << 20, b - o), o) < 0) perror("pwrite");
        });
        close(fd);
        printf("pwrite disjoint ranges, %2d threads: %.2f GB/s\n", T, G / (now() - t0) / 1e9);
    }
    for (int mode = 0; mode < 3; mode++)
        for (int T : {1, 4, 8, 16, 32}) {
            int fd = fresh();
            if (ftruncate(fd, G)) perror("ftruncate");
            double t0 = now();
            char* out = (char*)mmap(nullptr, G, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (out == MAP_FAILED) { perror("mmap"); return 1; }
            int adv = 0;
            if (mode == 1) adv = madvise(out, G, MADV_HUGEPAGE);
            par(T, [&](int t) {
                size_t a = G / T * t, b = t == T - 1 ? G : G / T * (t + 1);
                a &= ~size_t(4095); if (t != T - 1) b &= ~size_t(4095);
                for (size_t o = a; o < b; o += 8u << 20) {
                    size_t n = std::min<size_t>(8u << 20, b - o);
                    if (mode == 2 && madvise(out + o, n, MADV_POPULATE_WRITE)) { static std::atomic<int> once{0}; if (!once++) perror("MADV_POPULATE_WRITE"); }
                    memcpy(out + o, in + o, n);
                }
            });
            munmap(out, G);
            close(fd);
            printf("mmap shared %s, %2d threads: %.2f GB/s%s\n", mode == 0 ? "plain" : mode == 1 ? "MADV_HUGEPAGE" : "POPULATE_WRITE", T,
                   G / (now() - t0) / 1e9, adv ? " (madvise failed)" : "");
        }
    unlink(fout.c_str());
    // ---- fallocate then pwrite
    {
        int fd = fresh();
        double t0 = now();
        if (posix_fallocate(fd, 0, G)) perror("fallocate");
        double t1 = now();
        par(8, [&](int t) {
            size_t a = G / 8 * t, b = t == 7 ? G : G / 8 * (t + 1);
            for (size_t o = a; o < b; o += 8u << 20) if (pwrite(fd, in + o, std::min<size_t>(8u << 20, b - o), o) < 0) perror("pwrite");
        });
        close(fd);
        printf("fallocate %.2f GB/s, then pwrite 8 threads %.2f GB/s\n", G / (t1 - t0) / 1e9, G / (now() - t1) / 1e9);
        unlink(fout.c_str());
    }

    // ---- fallocate ahead (one thread, 256-MB steps) + T threads filling the step through a shared mapping
    for (int pop = 0; pop < 2; pop++)
    for (int T : {1, 2, 4, 8, 16}) {
        int fd = fresh();
        const size_t STEP = 256u << 20;
        double t0 = now();
        std::thread prev;
        for (size_t o = 0; o < G; o += STEP) {
            const size_t n = std::min(STEP, G - o);
            if (fallocate(fd, 0, o, n)) perror("fallocate");
            if (prev.joinable()) prev.join();
            prev = std::thread([=] {
                char* out = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, o);
                if (out == MAP_FAILED) { perror("mmap"); return; }
                par(T, [&](int t) {
                    size_t a = n / T * t, b = t == T - 1 ? n : n / T * (t + 1);
                    if (pop) madvise(out + (a & ~size_t(4095)), (b - (a & ~size_t(4095)) + 4095) & ~size_t(4095), MADV_POPULATE_WRITE);
                    memcpy(out + a, in + o + a, b - a);
                });
                munmap(out, n);
            });
        }
        prev.join();
        close(fd);
        printf("fallocate ahead + shared-mapping fill%s, %2d threads: %.2f GB/s\n", pop ? " (POPULATE_WRITE)" : "", T, G / (now() - t0) / 1e9);
    }
    unlink(fout.c_str());
    // ---- populate the input mapping in parallel, then scan
    for (int T : {8, 32}) {
        munmap(in, G);
        in = (char*)mmap(nullptr, G, PROT_READ, MAP_PRIVATE, fdi, 0);
        double t0 = now();
        par(T, [&](int t) {
            size_t a = (G / T * t) & ~size_t(4095), b = t == T - 1 ? G : (G / T * (t + 1)) & ~size_t(4095);
            if (madvise(in + a, b - a, 22 /* MADV_POPULATE_READ */)) perror("MADV_POPULATE_READ");
        });
        printf("MADV_POPULATE_READ of the input mapping, %2d threads: %.2f GB/s\n", T, G / (now() - t0) / 1e9);
    }
    // ---- to /dev/null
    {
        int fd = open("/dev/null", O_WRONLY);
        double t0 = now();
        for (size_t o = 0; o < G; o += CH) if (write(fd, in + o, std::min(CH, G - o)) < 0) perror("write");
        printf("write to /dev/null: %.2f GB/s\n", G / (now() - t0) / 1e9);
        close(fd);
    }
    // ---- PCIe
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || !ndev) { printf("no HIP device\n"); return 0; }
    double ti = now();
    hipSetDevice(0);
    hipFree(0);
    printf("HIP init: %.3f s\n", now() - ti);
    const size_t DB = 1ull << 30;
    char* d[4];
    hipStream_t st[4];
    for (int i = 0; i < 4; i++) { hipMalloc(&d[i], DB); hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); }
    for (int T : {1, 2, 3, 4}) {
        double t0 = now();
        par(T, [&](int t) {
            hipSetDevice(0);
            size_t a = G / T * t, b = t == T - 1 ? G : G / T * (t + 1);
            for (size_t o = a; o < b; o += 256u << 20) {
                hipMemcpyAsync(d[t], in + o, std::min<size_t>(256u << 20, b - o), hipMemcpyHostToDevice, st[t]);
                hipStreamSynchronize(st[t]);
            }
        });
        printf("H2D from the pageable input mapping, %d threads/streams: %.2f GB/s\n", T, G / (now() - t0) / 1e9);
    }
    {
        char* pin[4];
        const size_t PB = 256u << 20;
        double t0 = now();
        for (int i = 0; i < 4; i++) hipHostMalloc(&pin[i], PB, hipHostMallocDefault);
        printf("hipHostMalloc 4 x 256 MB: %.3f s\n", now() - t0);
        memcpy(pin[0], in, PB);
        t0 = now();
        for (int r = 0; r < 8; r++) hipMemcpyAsync(d[0], pin[0], PB, hipMemcpyHostToDevice, st[0]);
        hipStreamSynchronize(st[0]);
        printf("H2D from pinned: %.2f GB/s\n", 8.0 * PB / (now() - t0) / 1e9);
        t0 = now();
        for (int r = 0; r < 8; r++) hipMemcpyAsync(pin[1], d[0], PB, hipMemcpyDeviceToHost, st[0]);
        hipStreamSynchronize(st[0]);
        printf("D2H to pinned: %.2f GB/s\n", 8.0 * PB / (now() - t0) / 1e9);
        // staged: T threads memcpy (or pread) sub-blocks into a pinned buffer, then one DMA; two buffers alternate
        for (int how = 0; how < 2; how++)
            for (int T : {2, 4, 8}) {
                t0 = now();
                int k = 0;
                for (size_t o = 0; o < G; o += PB, k ^= 1) {
                    const size_t n = std::min(PB, G - o);
                    hipStreamSynchronize(st[k]);               // the DMA that used this buffer two rounds ago
                    par(T, [&](int t) {
                        size_t a = n / T * t, b = t == T - 1 ? n : n / T * (t + 1);
                        if (how == 0) memcpy(pin[k] + a, in + o + a, b - a);
                        else if (pread(fdi, pin[k] + a, b - a, o + a) < 0) perror("pread");
                    });
                    hipMemcpyAsync(d[k], pin[k], n, hipMemcpyHostToDevice, st[k]);
                }
                hipStreamSynchronize(st[0]); hipStreamSynchronize(st[1]);
                printf("%s into pinned with %d threads + DMA (double buffered): %.2f GB/s\n", how ? "pread " : "memcpy", T, G / (now() - t0) / 1e9);
            }
        t0 = now();
        hipError_t e = hipHostRegister(in, std::min<size_t>(G, 2ull << 30), hipHostRegisterReadOnly);
        printf("hipHostRegister of 2 GB of the input mapping: %s, %.3f s\n", hipGetErrorString(e), now() - t0);
        if (e == hipSuccess) {
            t0 = now();
            for (size_t o = 0; o < (2ull << 30); o += DB) hipMemcpyAsync(d[0], in + o, DB, hipMemcpyHostToDevice, st[0]);
            hipStreamSynchronize(st[0]);
            printf("H2D from the registered mapping: %.2f GB/s\n", (2ull << 30) / (now() - t0) / 1e9);
        }
    }
    unlink(fin.c_str());
    return 0;
}
