/* scan_test_stride_sim.c -- how often would k_mid_flat's hot loop leave for its rare path if the candidate test ran every
 * 8th column instead of every 4th?    TEST / MEASUREMENT TOOL, not part of the product.
 *
 *   gcc -O2 -o /tmp/scan_sim tools/scan_test_stride_sim.c -lm && /tmp/scan_sim [columns]
 *
 * Plain dynamic programming (the oracle's recurrence): bottom-row values D[Q][j] of the ONT rapid adapter
 * and of its reverse complement (config C2's pair, Q = 50, k = Q - M + 1 = 16) against uniform random text, 20 M columns.
 * The hot loop tests "min over the pass's adapters of the bottom-row value <= threshold" for all 64 lanes of a wave at once
 * (tgsf_kernels.h, chunk16): every 4th column against k + 3 today (a value can fall by at most 1 a column, so columns of a
 * group of 4 that reach k show as <= k + 3 at its end).  Every 8th column needs k + 7 at the group's end, or k + 4 when the
 * group's MIDDLE column is looked at (4 columns either way).  A wave leaves the hot loop when ANY of its 64 lanes passes. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv)
{
    const char* a = "GTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA";
    const int Q = (int)strlen(a), k = 16;
    char rc[64];
    for (int i = 0; i < Q; i++) { const char c = a[Q - 1 - i]; rc[i] = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'G' ? 'C' : 'G'; }
    rc[Q] = 0;
    const char* ads[2] = {a, rc};
    const long N = argc > 1 ? atol(argv[1]) : 20000000;
    srand(1);
    static int D[2][64];
    for (int j = 0; j < 2; j++) for (int i = 0; i <= Q; i++) D[j][i] = i;
    long t4[12] = {0}, t8mid[12] = {0}, t8end[12] = {0}, n4 = 0, n8 = 0;
    int s[2][8];
    for (long c = 0; c < N; c++) {
        const char t = "ACGT"[rand() & 3];
        for (int j = 0; j < 2; j++) {
            int diag = D[j][0];
            D[j][0] = 0;                                   /* free start in the text (HW) */
            for (int i = 1; i <= Q; i++) {
                const int up = D[j][i - 1], left = D[j][i];
                int v = diag + (ads[j][i - 1] != t);
                if (up + 1 < v) v = up + 1;
                if (left + 1 < v) v = left + 1;
                diag = left; D[j][i] = v;
            }
            s[j][c & 7] = D[j][Q];
        }
        if ((c & 3) == 3) { n4++; const int m = s[0][c & 7] < s[1][c & 7] ? s[0][c & 7] : s[1][c & 7]; for (int d = 0; d < 12; d++) if (m <= k + d) t4[d]++; }
        if ((c & 7) == 7) {
            n8++;
            const int m = s[0][3] < s[1][3] ? s[0][3] : s[1][3], e = s[0][7] < s[1][7] ? s[0][7] : s[1][7];
            for (int d = 0; d < 12; d++) { if (m <= k + d) t8mid[d]++; if (e <= k + d) t8end[d]++; }
        }
    }
    printf("ONT rapid adapter + reverse complement (Q = %d, k = %d), %ld random columns\n", Q, k, N);
    printf("%-52s %12s %14s\n", "test", "P(one lane)", "P(wave of 64)");
#define ROW(name, cnt, n) printf("%-52s %12.3e %14.4f\n", name, (double)(cnt) / (n), 1.0 - pow(1.0 - (double)(cnt) / (n), 64))
    ROW("every 4th column, value <= k + 3 (today)", t4[3], n4);
    ROW("every 8th column, middle column's value <= k + 4", t8mid[4], n8);
    ROW("every 8th column, last column's value <= k + 7", t8end[7], n8);
    const double p4 = 1.0 - pow(1.0 - (double)t4[3] / n4, 64), p8 = 1.0 - pow(1.0 - (double)t8mid[4] / n8, 64);
    /* instruction model, per 8 columns of a two-adapter pass (profiles/r04_scan_isa_census.txt: 38.28 VALU a column pair, of which the
     * test is 2 x 5 a group of four; the rare path scores every column of the group for both adapters: ~12 VALU a column and adapter
     * + the notes) */
    const double hot8 = 8 * 38.28, test = 10.0, rare4 = 2 * 4 * 12.0 + 20, rare8 = 2 * 8 * 12.0 + 20;
    const double now = hot8 + 2 * p4 * rare4, then = hot8 - test + p8 * rare8;
    printf("VALU per 8 columns and wave: today %.1f (2 tests, rare path %.1f %% of them) -> every 8th %.1f (1 test, rare path %.1f %%): %+.2f %%\n",
           now, 100 * p4, then, 100 * p8, 100 * (then - now) / now);
    return 0;
}
