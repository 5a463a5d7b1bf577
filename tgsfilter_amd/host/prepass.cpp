#include "prepass.h"
#include <chrono>
#include <cstdio>

#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>
#include <unordered_map>
#include <vector>

#include "api.h"
#include "fatal.h"

namespace host {

const char* const kAdapterLib[22] = {
    "ATCTCTCTCTTTTCCTCCTCCTCCGTTGTTGTTGTTGAGAGAGAT",   // 0  PacBio blunt adapter
    "ATCTCTCTCAACAACAACAACGGAGGAGGAGGAAAAGAGAGAGAT",
    "AAAAAAAAAAAAAAAAAATTAACGGAGGAGGAGGA",            // 2  PacBio C2 primer
    "TCCTCCTCCTCCGTTAATTTTTTTTTTTTTTTTTT",
    "AATGTACTTCGTTCAGTTACGTATTGCT",                   // 4  ONT ligation
    "AGCAATACGTAACTGAACGAAGTACATT",
    "GCAATACGTAACTGAACGAAGT",                         // 6  ONT ligation
    "ACTTCGTTCAGTTACGTATTGC",
    "GTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA",   // 8  ONT rapid
    "TGAAGCGGCGCACGAAAAACGCGAAAGCGTTTCACGATAAATGCGAAAAC",
    "GGCGTCTGCTTGGGTGTTTAACCTTTTTGTCAGAGAGGTTCCAAGTCAGAGAGGTTCCT",          // 10 ONT 1D^2
    "AGGAACCTCTCTGACTTGGAACCTCTCTGACAAAAAGGTTAAACACCCAAGCAGACGCC",
    "GGAACCTCTCTGACTTGGAACCTCTCTGACAAAAAGGTTAAACACCCAAGCAGACGCCAGCAAT",     // 12 ONT 1D^2
    "ATTGCTGGCGTCTGCTTGGGTGTTTAACCTTTTTGTCAGAGAGGTTCCAAGTCAGAGAGGTTCC",
    "TTTTTTTTCCTGTACTTCGTTCAGTTACGTATTGCT",           // 14 LA / NA / RA / RAT top strand
    "AGCAATACGTAACTGAACGAAGTACAGGAAAAAAAA",
    "GCAATACGTAACTGAACGAAGTACAGG",                    // 16 ligation adapter bottom strand
    "CCTGTACTTCGTTCAGTTACGTATTGC",
    "ACGTAACTGAACGAAGTACAGG",                         // 18 native adapter bottom strand
    "CCTGTACTTCGTTCAGTTACGT",
    "CTTGCGGGCGGCGGACTCTCCTCTGAAGATAGAGCGACAGGCAAG",  // 20 cDNA RT adapter
    "CTTGCCTGTCGCTCTATCTTCAGAGGAGAGTCCGCCGCCCGCAAG",
};

std::string rev_comp(const std::string& s)
{
    static char comp[256];
    static bool init = false;
    if (!init) {
        memset(comp, 'N', sizeof comp);
        const char* a = "AGCTagctMRWSYKmrwsyk";
        const char* b = "TCGAtcgaKYWSRMkywsrm";
        for (int i = 0; a[i]; i++) comp[(unsigned char)a[i]] = b[i];
        init = true;
    }
    std::string r;
    r.reserve(s.size());
    for (size_t i = s.size(); i-- > 0;) r += comp[(unsigned char)s[i]];
    return r;
}

// CheckBaseContent, :1079-1146
static int check_base_content(const std::vector<std::string>& ends, int check_len, int seq_num, float end_bias)
{
    // column of a base in the tallies (4 = not counted): a table instead of a switch, the bases being random
    static const struct Col { uint8_t of[256]; Col() { memset(of, 4, sizeof of); of['A'] = of['a'] = 0; of['T'] = of['t'] = 1; of['G'] = of['g'] = 2; of['C'] = of['c'] = 3; } } col;
    std::vector<int> cnt5((size_t)check_len * 5, 0);
    for (const std::string& s : ends) {
        const size_t n = std::min(s.size(), (size_t)check_len);
        for (size_t i = 0; i < n; i++) cnt5[i * 5 + col.of[(unsigned char)s[i]]]++;
    }
    std::vector<int> cnt((size_t)check_len * 4, 0);
    for (int i = 0; i < check_len; i++) for (int j = 0; j < 4; j++) cnt[(size_t)i * 4 + j] = cnt5[(size_t)i * 5 + j];
    const int max_diff = (int)((seq_num * end_bias) / 100);          // :1097 int(float)
    int trim = 0;
    for (int i = 1; i < check_len - 1; i++) {
        const int ld = std::min(i, 5), rd = std::min(check_len - i - 1, 5);
        bool left = false, right = false;
        for (int j = 0; j < 4; j++) {
            for (int x = 1; x <= ld; x++)
                if (abs(cnt[(size_t)i * 4 + j] - cnt[(size_t)(i - x) * 4 + j]) > max_diff) { left = true; break; }
            for (int x = 1; x <= rd; x++)
                if (abs(cnt[(size_t)(i + x) * 4 + j] - cnt[(size_t)i * 4 + j]) > max_diff) { right = true; break; }
        }
        if (left && right) trim = i + 1;                                // last such position wins, :1131-1133
    }
    return trim;
}

static double pp_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double g_pp_create = 0, g_pp_align = 0; static int g_pp_calls = 0;

// adapterSearch, :1148-1209, with the 22 x N alignments done by the library.
static void adapter_search(const Options& o, const std::vector<std::string>& ends, std::string& adapter, float& depth)
{
    adapter.clear(); depth = 0.f;
    float min_sim = o.mid_sim;
    if (min_sim < 0.9) min_sim = 0.9f;                                  // :1151-1154
    if (ends.empty()) return;

    tgsf_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.min_len = 100; p.max_len = 2147483647; p.min_q = 0; p.max_q = 255; p.bc_len = 0;
    p.end_len = 0; p.extra_len = 0; p.end_sim = 0.75f; p.mid_sim = 0.9f; p.filter = 1; p.qtype = 33;
    p.n_adapters = 22;
    int k[22];
    size_t max_q = 0;
    for (int a = 0; a < 22; a++) {
        p.adapters[a] = kAdapterLib[a];
        p.adapter_len[a] = (int)strlen(kAdapterLib[a]);
        k[a] = (int)((1 - min_sim) * p.adapter_len[a]) + 1;             // :1161
        max_q = std::max(max_q, (size_t)p.adapter_len[a]);
    }
    // thresholds only size the library's traceback scratch: it must cover Q + k columns
    p.end_match_len = 2; p.mid_match_len = (int)max_q;                  // k_end = Q-1 >= k[a]
    const size_t per_call = 4096;                                       // read ends per library call
    p.max_batch_reads = (uint32_t)(per_call * 22 / (22 * 2) + 64);      // n problems <= cap_reads * A * 2
    p.max_batch_bases = 1 << 20; p.max_read_len = 1 << 16;
    tgsf_ctx* ctx = nullptr;
    const double c0 = pp_now();
    if (lib().create(&p, o.devices.empty() ? o.device : o.devices[0], &ctx) != TGSF_OK) { std::cerr << "Error: " << lib().last_error(nullptr) << std::endl; quit(255); }

    g_pp_create += pp_now() - c0;
    // totals per adapter in the reference's own container (:1150, :1171): the winner among equal totals is
    // whatever its iteration order and std::sort make of it, reproduced here by using the same ones
    std::unordered_map<int, int> maps;
    std::vector<uint8_t> buf; std::vector<uint64_t> off; std::vector<uint32_t> len; std::vector<uint8_t> aid;
    std::vector<int32_t> kk, res, eds;
    for (size_t b = 0; b < ends.size(); b += per_call) {
        const size_t e = std::min(ends.size(), b + per_call);
        buf.clear(); off.clear(); len.clear(); aid.clear(); kk.clear();
        for (size_t i = b; i < e; i++) {
            const uint64_t o0 = buf.size();
            buf.insert(buf.end(), ends[i].begin(), ends[i].end());
            if (ends[i].empty()) continue;
            for (int a = 0; a < 22; a++) { off.push_back(o0); len.push_back((uint32_t)ends[i].size()); aid.push_back((uint8_t)a); kk.push_back(k[a]); }
        }
        const uint32_t n = (uint32_t)off.size();
        if (!n) continue;
        res.assign((size_t)n * 4, 0); eds.assign((size_t)n * 2, 0);
        const double a0 = pp_now();
        if (lib().align_windows(ctx, buf.data(), buf.size(), off.data(), len.data(), aid.data(), kk.data(), n, res.data(), eds.data()) != TGSF_OK) {
            std::cerr << "Error: " << lib().last_error(ctx) << std::endl; quit(255);
        }
        g_pp_align += pp_now() - a0; g_pp_calls++;
        for (uint32_t i = 0; i < n; i++)
            if (res[(size_t)i * 4 + 1] > 0)                             // numAln > 0, :1170 (problems are read-major, as :1156-1158)
                maps[(int)aid[i]] += res[(size_t)i * 4 + 2] - res[(size_t)i * 4 + 0];   // mlen = alignmentLength - editDistance
    }
    lib().destroy(ctx);
    std::vector<std::pair<int, int>> vec(maps.begin(), maps.end());     // :1179-1182
    std::sort(vec.begin(), vec.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.second > b.second; });
    if (vec.empty()) return;
    const int best = vec.front().first;
    const std::string cand = kAdapterLib[best];
    const float mean_dep = (float)maps[best] / (float)cand.size();      // :1189
    if (mean_dep >= 2 * min_sim) { adapter = cand; depth = mean_dep; }  // :1193
}

PrepassResult run_prepass(Options& o, const std::function<bool(Rec&)>& next_record)
{
    PrepassResult R;
    const double t0 = pp_now();
    int check_len = std::max(std::max(o.end_len, o.bc_len), 100);       // :897-904
    int min_len = std::max(o.min_len, 2 * check_len);                   // :906-909
    const int max_seq = std::max(o.ad_num, o.bc_num);                   // :911-914
    std::vector<std::string> ends5, ends3;
    const bool need5 = o.filter && (o.head_trim < 0 || o.adapter_file.empty());
    const bool need3 = o.filter && (o.tail_trim < 0 || o.adapter_file.empty());
    int seq_num = 0, min_qc = 255, max_qc = 0;
    {
        Rec r;                                                          // read_fastx / read_bam, :949-1040
        const bool has_qual = o.in_type != 0;
        while (next_record(r)) {
            const int L = (int)r.len;
            if (L < min_len) continue;
            if (seq_num >= max_seq) break;
            seq_num++;
            // the read ends are only looked at by the base-content check and the adapter search
            if (need5) ends5.emplace_back(r.seq, (size_t)check_len);
            if (need3) ends3.emplace_back(rev_comp(std::string(r.seq + (L - check_len), (size_t)check_len)));
            if (has_qual)
                for (int i = 0; i < check_len; i++) { const char c = r.qual[i]; if (min_qc > c) min_qc = c; if (max_qc < c) max_qc = c; }
        }
    }
    const double t_read = pp_now() - t0;
    if (o.in_type == 1 || o.in_type == 2) {                             // Get_qType, :1042-1077
        if (min_qc >= 33 && min_qc <= 78 && max_qc >= 33 && max_qc <= 127) R.qtype = 33;
        else if (min_qc >= 64 && min_qc <= 108 && max_qc >= 64 && max_qc <= 127) R.qtype = 64;
        else R.qtype = min_qc < 55 ? 33 : 64;
        std::cerr << "INFO: base quality scoring: Phred" << R.qtype << std::endl;
        const int maxq = max_qc - R.qtype;
        if (o.min_q >= 0) {
            if (o.min_q >= maxq) {
                std::cerr << "Warning: max base quality score was: " << maxq << std::endl;
                std::cerr << "INFO: Please reset -q parameter." << std::endl;
                quit(255);
            }
        } else {
            if (maxq > 10 && o.read_type == "clr") o.min_q = 10;
            else if (maxq > 20 && o.read_type == "hifi") o.min_q = 20;
            else if (maxq > 10 && o.read_type == "ont") o.min_q = 10;
            else o.min_q = 0;
        }
    }
    const double t1 = pp_now();
    double t_bc = 0;
    if (o.filter) {                                                     // :926-945
        {   // the two checks side by side, as the reference runs them (:930-936)
            std::thread t3;
            if (o.tail_trim < 0) t3 = std::thread([&] { R.trim3p = check_base_content(ends3, check_len, seq_num, o.end_bias); });
            if (o.head_trim < 0) R.trim5p = check_base_content(ends5, check_len, seq_num, o.end_bias);
            if (t3.joinable()) t3.join();
        }
        // :1135-1137 clamps trim5p to -e BEFORE the result of the running check is stored, and the 5' and the 3'
        // check run as two threads (:930-936): what the clamp sees is the 5' result when the 3' thread gets there
        // after the 5' thread has stored it -- the order observed in practice (the 5' thread starts first).  So the
        // 5' trim is clamped exactly when both checks run; the 3' trim never is.
        if (o.head_trim < 0 && o.tail_trim < 0 && R.trim5p > o.bc_len) R.trim5p = o.bc_len;
        t_bc = pp_now() - t1;
        if (o.adapter_file.empty()) {
            adapter_search(o, ends5, R.adapter5p, R.depth5p);
            adapter_search(o, ends3, R.adapter3p, R.depth3p);
        }
    }
    if (getenv("TGSF_TIMING"))
        fprintf(stderr, "PREPASS: %d reads looked at in %.3f s | base content %.3f | adapter search %.3f (contexts %.3f, %d alignment calls %.3f)\n",
                seq_num, t_read, t_bc, pp_now() - t1 - t_bc, g_pp_create, g_pp_calls, g_pp_align);
    return R;
}

}  // namespace host
