/*
 * tgsf_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * A CPU restatement, in this repo's own words, of TGSFilter's per-read filtering
 * hot path (reference: /root/reference, v1.11).  It exists so that the parity
 * tests can check the HIP path (libtgsf.so) bit for bit.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product never links or calls anything in oracle/.
 *
 * Parity is PINNED: this restatement is checked (tests/test_oracle_pinned.py,
 * tests/golden/) against
 *   - the reference's own edlib compiled from its sources (oracle/_ref/libedlib_ref.so)
 *     on random and adversarial (adapter, window, k) triples, and
 *   - whole-program outputs of the reference binary (oracle/_ref/tgsfilter_ref -t 1)
 *     on seeded synthetic FASTQ, committed as fixtures under tests/golden/.
 *
 * Every function cites the reference lines it follows.
 *
 * Arithmetic is written as plain dynamic programming / plain loops: this file
 * favours being obviously right over being fast.  (The long infix scan uses a
 * rolling two-column DP so that a 50 kb read costs Q*L cell updates.)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/tgsf.h"
#include "tgsf_oracle.h"

/* ------------------------------------------------------------------------- */
/* edlibAlign(query, target, k, EDLIB_MODE_HW, EDLIB_TASK_PATH) as consumed    */
/* by TGSFilter.   include/edlib.cpp:141-296 (driver), :547-704 (infix scan), */
/* :223-267 (start locations), :945-1144 (traceback).  SURVEY Appendix C.     */
/* ------------------------------------------------------------------------- */

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/*
 * Bottom row of the infix DP: D[i][0]=i, D[0][j]=0 (a match may start anywhere
 * in the window, include/edlib.cpp:584 startHout=0),
 * D[i][j] = min(D[i-1][j-1]+(q!=t), D[i-1][j]+1, D[i][j-1]+1).
 * Equality is exact byte equality (transformSequences, :1420-1459: every
 * distinct byte is its own symbol; no additional equalities are passed).
 * bottom[j] = D[Q][j] for j = 0..T.
 */
static void infix_bottom_row(const uint8_t* q, int Q, const uint8_t* t, int T, int* bottom)
{
    int* col = (int*)malloc(sizeof(int) * (size_t)(Q + 1));
    for (int i = 0; i <= Q; i++) col[i] = i;
    bottom[0] = Q;
    for (int j = 1; j <= T; j++) {
        int diag = col[0];     /* D[0][j-1] = 0 */
        col[0] = 0;            /* D[0][j]   = 0 */
        uint8_t tc = t[j - 1];
        for (int i = 1; i <= Q; i++) {
            int left = col[i];                 /* D[i][j-1] */
            int v = diag + (q[i - 1] != tc);
            if (left + 1 < v) v = left + 1;
            if (col[i - 1] + 1 < v) v = col[i - 1] + 1;
            diag = left;
            col[i] = v;
        }
        bottom[j] = col[Q];
    }
    free(col);
}

/*
 * Global (NW) DP of q (rows) against t[0..T) (columns): N[i][0]=i, N[0][j]=j.
 * Returned row-major, (Q+1) x (T+1).
 */
static int* global_matrix(const uint8_t* q, int Q, const uint8_t* t, int T)
{
    int W = T + 1;
    int* N = (int*)malloc(sizeof(int) * (size_t)(Q + 1) * (size_t)W);
    for (int j = 0; j <= T; j++) N[j] = j;
    for (int i = 1; i <= Q; i++) {
        N[i * W] = i;
        for (int j = 1; j <= T; j++) {
            int v = N[(i - 1) * W + (j - 1)] + (q[i - 1] != t[j - 1]);
            v = imin(v, N[(i - 1) * W + j] + 1);
            v = imin(v, N[i * W + (j - 1)] + 1);
            N[i * W + j] = v;
        }
    }
    return N;
}

/*
 * startLocations[i] (include/edlib.cpp:246-255): edlib reverses query and the
 * window prefix ending at `end`, runs a prefix-mode (SHW) scan with
 * k = editDistance and keeps the LAST column whose score equals the optimum,
 * i.e. the longest suffix t[s..end] whose global distance to q is `best`:
 * start = min { s : NW(q, t[s..end]) == best }.
 * Done here as a DP over reversed strings: R[i][l] = distance between the last
 * i chars of q and the last l chars of t[0..end]; R[i][0]=i, R[0][l]=l.
 */
static int start_location(const uint8_t* q, int Q, const uint8_t* t, int end, int best)
{
    int maxl = imin(end + 1, Q + best);  /* a span longer than Q+best costs more than best */
    int* prev = (int*)malloc(sizeof(int) * (size_t)(Q + 1));
    int* cur = (int*)malloc(sizeof(int) * (size_t)(Q + 1));
    for (int i = 0; i <= Q; i++) prev[i] = i;
    int best_l = -1;
    for (int l = 1; l <= maxl; l++) {
        uint8_t tc = t[end - (l - 1)];
        cur[0] = l;
        for (int i = 1; i <= Q; i++) {
            int v = prev[i - 1] + (q[Q - i] != tc);
            v = imin(v, prev[i] + 1);
            v = imin(v, cur[i - 1] + 1);
            cur[i] = v;
        }
        if (cur[Q] == best) best_l = l;
        int* tmp = prev; prev = cur; cur = tmp;
    }
    free(prev); free(cur);
    /* best_l >= 1 always: the infix optimum ending at `end` is such a span */
    return end - best_l + 1;
}

/*
 * alignmentLength of the path edlib reports for (start0, end0)
 * (include/edlib.cpp:271-284 -> obtainAlignment :1164 -> traceback :945-1144).
 * Trace back from the bottom-right cell with priority
 *   up   (consume an adapter char, EDLIB_EDOP_INSERT)   :1023
 *   left (consume a window char,   EDLIB_EDOP_DELETE)   :1057
 *   diagonal (match / mismatch)                         :1088
 * and count the columns.  When a border is reached the rest is straight
 * (:1028-1032, :1062-1067, :1093-1105).
 */
static int path_length(const uint8_t* q, int Q, const uint8_t* t, int T)
{
    int* N = global_matrix(q, Q, t, T);
    int W = T + 1;
    int i = Q, j = T, len = 0;
    while (i > 0 || j > 0) {
        if (i == 0) { len += j; break; }
        if (j == 0) { len += i; break; }
        int cur = N[i * W + j];
        if (N[(i - 1) * W + j] + 1 == cur) i--;
        else if (N[i * W + (j - 1)] + 1 == cur) j--;
        else { i--; j--; }
        len++;
    }
    free(N);
    return len;
}

/*
 * Last column of the global DP, O(Q) memory: col[i] = NW(q[0..i), t[0..T)) for i = 0..Q.
 * rev != 0: of the REVERSED strings, i.e. col[i] = NW(last i chars of q, last T chars ... all of t reversed).
 */
static void global_last_column(const uint8_t* q, int Q, const uint8_t* t, int T, int rev, int* col)
{
    for (int i = 0; i <= Q; i++) col[i] = i;
    for (int j = 1; j <= T; j++) {
        const uint8_t tc = rev ? t[T - j] : t[j - 1];
        int diag = col[0];
        col[0] = j;
        for (int i = 1; i <= Q; i++) {
            const int left = col[i];
            int v = diag + ((rev ? q[Q - i] : q[i - 1]) != tc);
            v = imin(v, left + 1);
            v = imin(v, col[i - 1] + 1);
            diag = left;
            col[i] = v;
        }
    }
}

/*
 * obtainAlignment, include/edlib.cpp:1164-1216: the path of the first location is found by traceback while the
 * traceback state -- (2 words + 1 int) per 64-row block and column, + 2 ints per column -- stays below 1 MiB
 * (:1191-1193), and by Hirschberg's divide and conquer otherwise (obtainAlignmentHirschberg, :1234-1400):
 *   the target is cut in the middle (left half targetLength / 2 columns, :1250-1251); with
 *   L[h] = NW(q[0..h), left half) and R[h] = NW(q[h..Q), right half) the query is cut at the SMALLEST h in 1..Q-1
 *   with L[h] + R[h] == best (:1321-1331: queryIdx = h - 1 ascending, first hit), else at h = 0 if
 *   leftHalfWidth + R[0] == best (:1333-1340), else at h = Q if L[Q] + rightHalfWidth == best (:1341-1349); the
 *   two quadrants recurse with their own scores (:1366-1384) and the lengths add up (:1392).
 * (edlib computes L and R inside a band of width `best`; a cell on an optimal path lies inside both bands and holds its
 * exact value there, a cell that holds more than its exact value holds more than `best` -- so the first h is the same.)
 * Degenerate quadrants (:1171-1179): an empty query or target is all deletions / insertions.
 */
static int obtain_alignment_length(const uint8_t* q, int Q, const uint8_t* t, int T, int best)
{
    if (Q == 0 || T == 0) return Q + T;
    const long long blocks = (Q + 63) / 64;
    const long long data = (2ll * 8 + 4) * blocks * T + 2ll * 4 * T;
    if (data < 1024 * 1024) return path_length(q, Q, t, T);
    const int lw = T / 2, rw = T - lw;
    int* L = (int*)malloc(sizeof(int) * (size_t)(Q + 1));
    int* Rr = (int*)malloc(sizeof(int) * (size_t)(Q + 1));
    global_last_column(q, Q, t, lw, 0, L);
    global_last_column(q, Q, t + lw, rw, 1, Rr);          /* Rr[i] = NW(last i chars of q, right half) = R[Q - i] */
    int h = -1;
    for (int x = 1; x <= Q - 1 && h < 0; x++)
        if (L[x] + Rr[Q - x] == best) h = x;
    if (h < 0 && lw + Rr[Q] == best) h = 0;
    if (h < 0 && L[Q] + rw == best) h = Q;
    int len = -1;
    if (h >= 0) {
        const int ls = h == 0 ? lw : L[h], rs = h == Q ? rw : Rr[Q - h];
        const int a = obtain_alignment_length(q, h, t, lw, ls), b = obtain_alignment_length(q + h, Q - h, t + lw, rw, rs);
        len = (a < 0 || b < 0) ? -1 : a + b;               /* (a quadrant without a cut: the whole alignment has none) */
    }
    free(L); free(Rr);
    return len;           /* (-1: no cut adds up to `best` -- the reference returns EDLIB_STATUS_ERROR; never seen) */
}

int orc_align_hw(const uint8_t* q, int Q, const uint8_t* t, int T, int k,
                 orc_alignment* out)
{
    memset(out, 0, sizeof(*out));
    out->edit_distance = -1;
    if (Q <= 0 || T <= 0 || k < 0) return -1;   /* never reached from TGSFilter (:1237, :1274) */
    int kk = imin(k, Q);                          /* include/edlib.cpp:565-567 */
    int* bottom = (int*)malloc(sizeof(int) * (size_t)(T + 1));
    infix_bottom_row(q, Q, t, T, bottom);
    int best = bottom[1];
    for (int j = 2; j <= T; j++) best = imin(best, bottom[j]);
    if (best > kk) { free(bottom); return 0; }   /* editDistance=-1, numLocations=0, alignmentLength=0 */
    int n = 0;
    for (int j = 1; j <= T; j++) n += (bottom[j] == best);
    out->edit_distance = best;
    out->num_locations = n;
    out->ends = (int*)malloc(sizeof(int) * (size_t)n);
    out->starts = (int*)malloc(sizeof(int) * (size_t)n);
    n = 0;
    for (int j = 1; j <= T; j++)                 /* all global-minimum columns, ascending (:660-672) */
        if (bottom[j] == best) out->ends[n++] = j - 1;
    free(bottom);
    for (int i = 0; i < n; i++)
        out->starts[i] = start_location(q, Q, t, out->ends[i], best);
    int s0 = out->starts[0], e0 = out->ends[0];
    out->alignment_length = obtain_alignment_length(q, Q, t + s0, e0 - s0 + 1, best);
    return 0;
}

void orc_alignment_free(orc_alignment* a)
{
    free(a->starts); free(a->ends);
    a->starts = a->ends = NULL;
}

/* ------------------------------------------------------------------------- */
/* QC accumulators                                                            */
/* ------------------------------------------------------------------------- */

/* column of the 5-wide tables: A/a 0, T/t 1, G/g 2, C/c 3, anything else only "all" (4)
 * src/TGSFilter.cpp:1462-1476 */
static inline int base_column(uint8_t b)
{
    switch (b) {
    case 'A': case 'a': return 0;
    case 'T': case 't': return 1;
    case 'G': case 'g': return 2;
    case 'C': case 'c': return 3;
    default: return 4;
    }
}

/* qual[i] - qType with qual a (signed) char, added to uint64 accumulators
 * (src/TGSFilter.cpp:1457-1458): two's complement wrap-around is the semantics. */
static inline uint64_t qvalue(uint8_t qc, int qtype)
{
    return (uint64_t)(int64_t)((int)(int8_t)qc - qtype);
}

/* CalcAvgQuality, src/TGSFilter.cpp:1436-1479.  Returns sumQ (the caller divides). */
static uint64_t calc_avg_quality(const uint8_t* seq, const uint8_t* qual, uint64_t len, int qtype,
                                 uint64_t* tab_qual, uint64_t* tab_cnt, uint64_t* rows_used)
{
    uint64_t rows = len / TGSF_BIN_WIDTH + 1;     /* :1445 */
    if (rows > *rows_used) *rows_used = rows;     /* :1446-1449 resize */
    uint64_t sum = 0;
    for (uint64_t i = 0; i < len; i++) {
        uint64_t qv = qvalue(qual[i], qtype);
        sum += qv;
        uint64_t row = i / TGSF_BIN_WIDTH;
        int c = base_column(seq[i]);
        if (c < 4) { tab_cnt[row * 5 + c]++; tab_qual[row * 5 + c] += qv; }
        tab_cnt[row * 5 + 4]++; tab_qual[row * 5 + 4] += qv;
    }
    return sum;
}

/* Get_5p_base_qual / Get_3p_base_qual, src/TGSFilter.cpp:1481-1575 */
static void end_tables(const uint8_t* seq, const uint8_t* qual, uint64_t len, int qtype, int bc_len,
                       uint64_t* q5, uint64_t* c5, uint64_t* q3, uint64_t* c3, uint64_t* rows_used)
{
    uint64_t n = (uint64_t)(bc_len < 0 ? 0 : bc_len);
    if (n > len) n = len;                         /* :1490-1493 */
    if (n > *rows_used) *rows_used = n;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t qv = qvalue(qual[i], qtype);
        int c = base_column(seq[i]);
        if (c < 4) { c5[i * 5 + c]++; q5[i * 5 + c] += qv; }
        c5[i * 5 + 4]++; q5[i * 5 + 4] += qv;
        uint64_t j = len - 1 - i;                 /* :1554-1557: position 0 is the last base */
        qv = qvalue(qual[j], qtype);
        c = base_column(seq[j]);
        if (c < 4) { c3[i * 5 + c]++; q3[i * 5 + c] += qv; }
        c3[i * 5 + 4]++; q3[i * 5 + 4] += qv;
    }
}

/* Count-only variants for records without qualities (FASTA input):
 * Get_base_counts :1577-1606, Get_5p_base_counts :1608-1640, Get_3p_base_counts :1642-1678.
 * As coded, the 5' variant tests 'G' twice (a lower-case g only reaches column 4) and the 3' variant tests
 * 'T' twice (same for a lower-case t). */
static void base_counts(const uint8_t* seq, uint64_t len, uint64_t* tab_cnt, uint64_t* rows_used)
{
    uint64_t rows = len / TGSF_BIN_WIDTH + 1;
    if (rows > *rows_used) *rows_used = rows;
    for (uint64_t i = 0; i < len; i++) {
        uint64_t row = i / TGSF_BIN_WIDTH;
        int c = base_column(seq[i]);
        if (c < 4) tab_cnt[row * 5 + c]++;
        tab_cnt[row * 5 + 4]++;
    }
}
static void end_counts(const uint8_t* seq, uint64_t len, int bc_len, uint64_t* c5, uint64_t* c3, uint64_t* rows_used)
{
    uint64_t n = (uint64_t)(bc_len < 0 ? 0 : bc_len);
    if (n > len) n = len;
    if (n > *rows_used) *rows_used = n;
    for (uint64_t i = 0; i < n; i++) {
        int c = seq[i] == 'g' ? 4 : base_column(seq[i]);               /* :1629 */
        if (c < 4) c5[i * 5 + c]++;
        c5[i * 5 + 4]++;
        uint64_t j = len - 1 - i;
        c = seq[j] == 't' ? 4 : base_column(seq[j]);                   /* :1667 */
        if (c < 4) c3[i * 5 + c]++;
        c3[i * 5 + 4]++;
    }
}

/* ------------------------------------------------------------------------- */
/* adapter search + region logic                                              */
/* ------------------------------------------------------------------------- */

typedef struct { int s, e; } region;
typedef struct { region* r; int n, cap; } region_vec;

static void rv_push(region_vec* v, int s, int e)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 16;
        v->r = (region*)realloc(v->r, sizeof(region) * (size_t)v->cap);
    }
    v->r[v->n].s = s; v->r[v->n].e = e; v->n++;
}

static int region_cmp(const void* a, const void* b)
{
    const region* x = (const region*)a; const region* y = (const region*)b;
    if (x->s != y->s) return x->s < y->s ? -1 : 1;
    if (x->e != y->e) return x->e < y->e ? -1 : 1;
    return 0;
}

/* GetEditDistance, src/TGSFilter.cpp:1218-1322 (one adapter against one read) */
static void get_edit_distance(const tgsf_params* p, const uint8_t* q, int Q,
                              const uint8_t* read, int L,
                              int* num5p, int* num3p, int* num_mid, region_vec* regions)
{
    int E = p->end_len;
    /* middle: :1233-1264 */
    int max_k = Q - p->mid_match_len + 1;
    int tsm = L - E - E;
    if (tsm >= Q) {
        orc_alignment a;
        if (orc_align_hw(q, Q, read + E, tsm, max_k, &a) == 0) {
            int mlen = a.alignment_length - a.edit_distance;      /* :1245 (0-(-1)=1 when nothing found) */
            if (mlen >= p->mid_match_len) {
                for (int i = 0; i < a.num_locations; i++) {
                    float sim = (float)mlen / (float)Q;           /* :1250 */
                    if (sim >= p->mid_sim) {
                        int ts = a.starts[i] + E - p->extra_len;
                        int te = a.ends[i] + E + 1 + p->extra_len;
                        if (ts < 0) ts = 0;
                        if (te > L) te = L;
                        (*num_mid)++;
                        rv_push(regions, ts, te);
                    }
                }
            }
            orc_alignment_free(&a);
        }
    }
    /* ends: :1266-1321 */
    int check = E + (int)((float)Q / p->end_sim);                  /* :1267 int(qLen / endSim), float division */
    if (check > L) check = L;
    max_k = Q - p->end_match_len + 1;
    if (check >= 5) {
        orc_alignment a;
        if (orc_align_hw(q, Q, read, check, max_k, &a) == 0) {
            int mlen = a.alignment_length - a.edit_distance;
            if (mlen >= p->end_match_len) {
                for (int i = 0; i < a.num_locations; i++) {
                    float sim = (float)mlen / (float)Q;
                    if (sim >= p->end_sim) { (*num5p)++; rv_push(regions, 0, a.ends[i] + 1); }
                }
            }
            orc_alignment_free(&a);
        }
        if (orc_align_hw(q, Q, read + (L - check), check, max_k, &a) == 0) {
            int mlen = a.alignment_length - a.edit_distance;
            if (mlen >= p->end_match_len) {
                for (int i = 0; i < a.num_locations; i++) {
                    float sim = (float)mlen / (float)Q;
                    if (sim >= p->end_sim) { (*num3p)++; rv_push(regions, a.starts[i] + L - check, L); }
                }
            }
            orc_alignment_free(&a);
        }
    }
}

/* adapterMap, src/TGSFilter.cpp:1325-1434.  keep[] receives {start,len} pairs. */
static void adapter_map(const tgsf_params* p, const uint8_t* read, int L,
                        region_vec* keep, uint64_t* drop, uint32_t* flags, uint32_t* trimmed)
{
    int n5 = 0, n3 = 0, nm = 0;
    region_vec regs = {0, 0, 0};
    if (p->head_trim > 0) rv_push(&regs, 0, p->head_trim >= L ? L : p->head_trim);          /* :1334-1340 */
    if (p->tail_trim > 0) {                                                                  /* :1342-1348 */
        if (p->tail_trim >= L) rv_push(&regs, 0, L); else rv_push(&regs, L - p->tail_trim, L);
    }
    for (int a = 0; a < p->n_adapters; a++)                                                  /* :1350-1352 */
        get_edit_distance(p, (const uint8_t*)p->adapters[a], p->adapter_len[a], read, L, &n5, &n3, &nm, &regs);

    /* :1354-1370 -- exactly one of DropInfo[2..9] */
    if (nm > 0 && n5 > 0 && n3 > 0) drop[2]++;
    else if (nm > 0 && n5 > 0) drop[3]++;
    else if (nm > 0 && n3 > 0) drop[4]++;
    else if (n5 > 0 && n3 > 0) drop[5]++;
    else if (nm > 0) drop[6]++;
    else if (n5 > 0) drop[7]++;
    else if (n3 > 0) drop[8]++;
    else drop[9]++;
    if (n5 > 0) *flags |= TGSF_RF_AD5P;
    if (n3 > 0) *flags |= TGSF_RF_AD3P;
    if (nm > 0) *flags |= TGSF_RF_ADMID;

    if (nm > 0 && p->discard) {                                                              /* :1372-1373 */
        drop[10] += (uint64_t)L;
        *trimmed += (uint32_t)L;
        *flags |= TGSF_RF_DISCARDED;
        free(regs.r);
        return;
    }
    if (regs.n > 1) qsort(regs.r, (size_t)regs.n, sizeof(region), region_cmp);               /* :1376-1381 */
    region_vec merged = {0, 0, 0};
    for (int i = 0; i < regs.n; i++) {                                                       /* :1383-1390 */
        if (merged.n > 0 && merged.r[merged.n - 1].e >= regs.r[i].s)
            merged.r[merged.n - 1].e = imax(merged.r[merged.n - 1].e, regs.r[i].e);
        else
            rv_push(&merged, regs.r[i].s, regs.r[i].e);
    }
    int cur = 0;
    if (merged.n >= 1) {                                                                     /* :1396-1424 */
        for (int i = 0; i < merged.n; i++) {
            int dl = merged.r[i].e - merged.r[i].s;
            drop[10] += (uint64_t)(int64_t)dl;
            *trimmed += (uint32_t)dl;
            if (dl == L) drop[11]++;
            if (merged.r[i].s > cur) {
                int kl = merged.r[i].s - cur;
                if (kl >= p->min_len && kl <= p->max_len) rv_push(keep, cur, kl);
                else { drop[11]++; drop[12] += (uint64_t)kl; }
            }
            cur = merged.r[i].e;
        }
        if (cur < L) {
            int kl = L - cur;
            if (kl >= p->min_len && kl <= p->max_len) rv_push(keep, cur, kl);
            else { drop[11]++; drop[12] += (uint64_t)kl; }
        }
    } else {                                                                                 /* :1425-1432 */
        if (L >= p->min_len && L <= p->max_len) rv_push(keep, 0, L);
        else { drop[11]++; drop[12] += (uint64_t)L; }
    }
    free(regs.r); free(merged.r);
}

/* GetKmerCount, src/TGSFilter.cpp:1703-1753: (#k-mers) - (#distinct k-mers) of a fragment, k-mers as
 * 2-bit codes A=0 C=1 G=2 T=3; any other byte (lower case, N) contributes 0 bits (:1722-1723). */
static int u64_cmp(const void* a, const void* b)
{
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : x > y;
}
static int kmer_repeat(const uint8_t* seq, int len, int k)
{
    int total = len - k + 1;
    if (total <= 0) return 0;
    uint64_t* v = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)total);
    /* :1748 masks with (1ULL << (2 * k)) - 1 from the SECOND k-mer on (the first is inserted as built, :1727).  At
     * k = 32 that shift is by 64: what the reference binary does there (x86 takes the count modulo 64: 1 << 0, so
     * the mask is 0 and every later k-mer is 0) is pinned by the goldens repeat_k32 / repeat_k32b. */
    uint64_t mask = k >= 32 ? 0ull : ((1ull << (2 * k)) - 1ull), km = 0;
    for (int i = 0; i < len; i++) {
        uint64_t c = seq[i] == 'C' ? 1 : seq[i] == 'G' ? 2 : seq[i] == 'T' ? 3 : 0;
        km = (km << 2) | c;
        if (i >= k) km &= mask;
        if (i >= k - 1) v[i - k + 1] = km;
    }
    qsort(v, (size_t)total, sizeof(uint64_t), u64_cmp);
    int distinct = 1;
    for (int i = 1; i < total; i++) distinct += v[i] != v[i - 1];
    free(v);
    return total - distinct;
}

/* ------------------------------------------------------------------------- */
/* filter_sequence for a batch, src/TGSFilter.cpp:1939-2061                    */
/* ------------------------------------------------------------------------- */

int orc_filter_batch(const tgsf_params* p, const tgsf_batch_in* in, tgsf_batch_out* out,
                     uint64_t* ctr, uint32_t n_bins)
{
    int bc = p->bc_len;
    uint64_t* drop = ctr + TGSF_CTR_DROPINFO;
    uint64_t* rows = ctr + TGSF_CTR_ROWS;
    uint64_t* T[8]; uint64_t* B[4];
    for (int t = 0; t < 8; t++) T[t] = ctr + tgsf_ctr_end_table(t, bc);
    for (int b = 0; b < 4; b++) B[b] = ctr + tgsf_ctr_bin_table(b, bc, n_bins);
    uint32_t nf = 0;

    for (uint32_t r = 0; r < in->n_reads; r++) {
        uint64_t off = in->offsets[r];
        uint64_t len64 = in->lengths ? in->lengths[r] : in->offsets[r + 1] - off;
        const uint8_t* seq = in->seq + off;
        const uint8_t* qual = p->no_qual ? seq : in->qual + (in->qual_offsets ? in->qual_offsets[r] : off);
        tgsf_read_result* rr = &out->reads[r];
        memset(rr, 0, sizeof(*rr));
        rr->frag_begin = nf;
        if (len64 == 0) continue;                 /* :1939 rawSeqLen > 0 */
        if (len64 / TGSF_BIN_WIDTH + 1 > n_bins) return TGSF_E_CAPACITY;
        int L = (int)len64;

        if (p->no_qual) {                                              /* :1954-1958 rawQualLen == 0 */
            base_counts(seq, len64, B[TGSF_B_RAW_CNT], &rows[0]);
            end_counts(seq, len64, bc, T[TGSF_T_RAW5P_CNT], T[TGSF_T_RAW3P_CNT], &rows[2]);
        } else {
            /* raw stats, always: :1942-1945 */
            uint64_t sum = calc_avg_quality(seq, qual, len64, p->qtype, B[TGSF_B_RAW_QUAL], B[TGSF_B_RAW_CNT], &rows[0]);
            double mean = (double)sum / (double)len64;                     /* :1478 */
            rr->sum_q = sum;
            if (!(mean >= 0.0 && mean < 256.0)) return TGSF_E_DATA;        /* the reference indexes out of bounds here */
            ctr[TGSF_CTR_RAW_DIFFQ + (int)mean] += len64;                  /* :1943 */
            end_tables(seq, qual, len64, p->qtype, bc, T[TGSF_T_RAW5P_QUAL], T[TGSF_T_RAW5P_CNT],
                       T[TGSF_T_RAW3P_QUAL], T[TGSF_T_RAW3P_CNT], &rows[2]);
            if (p->filter) {                                               /* :1946-1953 */
                if (mean < (double)p->min_q || mean > (double)p->max_q) {
                    drop[0]++; drop[1] += len64;
                    rr->flags |= TGSF_RF_LOWQ;
                    continue;
                }
            }
        }
        region_vec keep = {0, 0, 0};
        if (p->filter) adapter_map(p, seq, L, &keep, drop, &rr->flags, &rr->trimmed);   /* :1961-1965 */
        else rv_push(&keep, 0, L);

        if (!p->only_qc) {                                             /* :1976 */
            for (int f = 0; f < keep.n; f++) {
                int s = keep.r[f].s, fl = keep.r[f].e;                 /* rv holds {start,len} here */
                if (nf >= out->frag_capacity) { free(keep.r); return TGSF_E_CAPACITY; }
                tgsf_fragment* fr = &out->frags[nf++];
                fr->read = r; fr->start = s; fr->len = fl; fr->flags = 0; fr->sum_q = 0;
                if (p->min_repeat > 0) {                                                 /* :1982-1989 */
                    if (kmer_repeat(seq + s, fl, p->kmer) < p->min_repeat) {
                        drop[15]++; drop[16] += (uint64_t)fl;
                        fr->flags |= TGSF_FF_REPEAT;
                        continue;
                    }
                }
                if (p->no_qual) {                                                        /* :2005-2009 */
                    base_counts(seq + s, (uint64_t)fl, B[TGSF_B_CLEAN_CNT], &rows[1]);
                    end_counts(seq + s, (uint64_t)fl, bc, T[TGSF_T_CLEAN5P_CNT], T[TGSF_T_CLEAN3P_CNT], &rows[3]);
                    fr->flags |= TGSF_FF_PASS;
                    continue;
                }
                /* clean bin tables accumulate BEFORE the gate: :1994 */
                uint64_t cs = calc_avg_quality(seq + s, qual + s, (uint64_t)fl, p->qtype,
                                               B[TGSF_B_CLEAN_QUAL], B[TGSF_B_CLEAN_CNT], &rows[1]);
                fr->sum_q = cs;
                double cm = (double)cs / (double)fl;
                if (p->filter && (cm < (double)p->min_q || cm > (double)p->max_q)) {    /* :1995-2001 */
                    drop[13]++; drop[14] += (uint64_t)fl;
                    continue;
                }
                if (!(cm >= 0.0 && cm < 256.0)) { free(keep.r); return TGSF_E_DATA; }
                ctr[TGSF_CTR_CLEAN_DIFFQ + (int)cm] += (uint64_t)fl;                    /* :2002 */
                end_tables(seq + s, qual + s, (uint64_t)fl, p->qtype, bc,                /* :2003-2004 */
                           T[TGSF_T_CLEAN5P_QUAL], T[TGSF_T_CLEAN5P_CNT],
                           T[TGSF_T_CLEAN3P_QUAL], T[TGSF_T_CLEAN3P_CNT], &rows[3]);
                fr->flags |= TGSF_FF_PASS;
            }
            rr->n_frags = nf - rr->frag_begin;
        }
        free(keep.r);
    }
    out->n_frags = nf;
    return TGSF_OK;
}
