/* fq_multiset.c -- order-independent digest of the records of a FASTQ/FASTA file (bench.py: the reference writes
 * its records in a nondeterministic order with -t > 1, so outputs are compared as multisets).  TEST INFRASTRUCTURE.
 *   gcc -O2 -pthread -o tools/fq_multiset tools/fq_multiset.c ;  tools/fq_multiset file [lines_per_record=4] [threads]
 * prints: <records> <sum of record hashes mod 2^64> <xor of record hashes> <bytes>
 * Hundreds of GB are digested by several threads: pass 1 counts the line ends of every slice of the file, so that
 * pass 2 knows which line of a record each slice starts in and every thread hashes whole records only (a record that
 * straddles slices belongs to the slice it starts in).  The result does not depend on the number of threads. */
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static uint64_t hash(const unsigned char* p, size_t n)
{
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n * 0xFF51AFD7ED558CCDull;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
        h ^= h >> 29;
    }
    uint64_t w = 0;
    memcpy(&w, p + i, n - i);
    h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
    h ^= h >> 32;
    h *= 0xFF51AFD7ED558CCDull;
    return h ^ (h >> 33);
}

typedef struct {
    const unsigned char* d;
    size_t n, lo, hi;          /* the slice [lo, hi) */
    int per;
    uint64_t lines;            /* pass 1: line ends in the slice */
    uint64_t first_line;       /* pass 2 input: index (in the file) of the line that the first byte of the slice lies in */
    uint64_t recs, sum, x;
} Slice;

static void* count_lines(void* a)
{
    Slice* s = (Slice*)a;
    uint64_t c = 0;
    size_t at = s->lo;
    while (at < s->hi) {
        const unsigned char* q = memchr(s->d + at, '\n', s->hi - at);
        if (!q) break;
        c++;
        at = (size_t)(q - s->d) + 1;
    }
    s->lines = c;
    return NULL;
}

static void* digest(void* a)
{
    Slice* s = (Slice*)a;
    const unsigned char* d = s->d;
    const size_t n = s->n;
    size_t at = s->lo;
    uint64_t line = s->first_line;
    /* the first record that STARTS in the slice: the slice's first byte starts a line only if it follows a line end */
    if (at > 0 && d[at - 1] != '\n') {
        const unsigned char* q = memchr(d + at, '\n', n - at);
        at = q ? (size_t)(q - d) + 1 : n;
        line++;
    }
    while (at < n && line % (uint64_t)s->per) {
        const unsigned char* q = memchr(d + at, '\n', n - at);
        at = q ? (size_t)(q - d) + 1 : n;
        line++;
    }
    while (at < s->hi) {
        size_t e = at;
        for (int l = 0; l < s->per && e < n; l++) {
            const unsigned char* q = memchr(d + e, '\n', n - e);
            e = q ? (size_t)(q - d) + 1 : n;
        }
        const uint64_t h = hash(d + at, e - at);
        s->sum += h; s->x ^= h; s->recs++;
        at = e;
    }
    return NULL;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const int per = argc > 2 ? atoi(argv[2]) : 4;
    int T = argc > 3 ? atoi(argv[3]) : 0;
    int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { perror(argv[1]); return 1; }
    struct stat st;
    fstat(fd, &st);
    size_t n = (size_t)st.st_size;
    if (!n) { printf("0 0 0 0\n"); return 0; }
    const unsigned char* d = mmap(NULL, n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (d == MAP_FAILED) { perror("mmap"); return 1; }
    if (T <= 0) {
        long c = sysconf(_SC_NPROCESSORS_ONLN);
        T = c > 64 ? 64 : (c < 1 ? 1 : (int)c);
    }
    if ((size_t)T > n / (1u << 20) + 1) T = (int)(n / (1u << 20) + 1);      /* a slice is worth a thread from a MB on */
    if (per < 1) return 2;
    Slice* s = calloc((size_t)T, sizeof *s);
    pthread_t* th = calloc((size_t)T, sizeof *th);
    for (int t = 0; t < T; t++) {
        s[t].d = d; s[t].n = n; s[t].per = per;
        s[t].lo = n / (size_t)T * (size_t)t;
        s[t].hi = t + 1 == T ? n : n / (size_t)T * (size_t)(t + 1);
    }
    for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, count_lines, &s[t]);
    for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
    uint64_t lines = 0;
    for (int t = 0; t < T; t++) { s[t].first_line = lines; lines += s[t].lines; }
    for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, digest, &s[t]);
    for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
    uint64_t recs = 0, sum = 0, x = 0;
    for (int t = 0; t < T; t++) { recs += s[t].recs; sum += s[t].sum; x ^= s[t].x; }
    printf("%llu %llu %llu %zu\n", (unsigned long long)recs, (unsigned long long)sum, (unsigned long long)x, n);
    return 0;
}
