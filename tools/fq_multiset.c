/* fq_multiset.c -- order-independent digest of the records of a FASTQ/FASTA file (bench.py: the reference writes
 * its records in a nondeterministic order with -t > 1, so outputs are compared as multisets).  TEST INFRASTRUCTURE.
 *   gcc -O2 -o tools/fq_multiset tools/fq_multiset.c ;  tools/fq_multiset file [lines_per_record=4]
 * prints: <records> <sum of record hashes mod 2^64> <xor of record hashes> <bytes> */
#include <fcntl.h>
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

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const int per = argc > 2 ? atoi(argv[2]) : 4;
    int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { perror(argv[1]); return 1; }
    struct stat st;
    fstat(fd, &st);
    size_t n = (size_t)st.st_size;
    if (!n) { printf("0 0 0 0\n"); return 0; }
    const unsigned char* d = mmap(NULL, n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (d == MAP_FAILED) { perror("mmap"); return 1; }
    uint64_t recs = 0, sum = 0, x = 0;
    size_t at = 0;
    while (at < n) {
        size_t e = at;
        for (int l = 0; l < per && e < n; l++) {
            const unsigned char* q = memchr(d + e, '\n', n - e);
            e = q ? (size_t)(q - d) + 1 : n;
        }
        const uint64_t h = hash(d + at, e - at);
        sum += h; x ^= h; recs++;
        at = e;
    }
    printf("%llu %llu %llu %zu\n", (unsigned long long)recs, (unsigned long long)sum, (unsigned long long)x, n);
    return 0;
}
