#include "report.h"

#include <unistd.h>

#include <fstream>
#include <iostream>
#include <iterator>

#include <cmath>
#include <ctime>
#include <sstream>
#include <unordered_map>

namespace host {

std::string limit_decimals(double v, int places)
{
    std::string s = std::to_string(v);
    const size_t dot = s.find('.');
    if (dot != std::string::npos && s.size() - dot > (size_t)(places + 1)) s = s.substr(0, dot + places + 1);
    return s;
}

static int n50(const std::vector<int>& lens, uint64_t bases)
{
    uint64_t acc = 0;
    for (size_t i = lens.size(); i-- > 0;) { acc += (uint64_t)lens[i]; if (acc >= bases / 2) return lens[i]; }
    return 0;
}

// variable-width length axis: step = max(100, i/20)
static std::vector<uint64_t> axis(uint64_t first, uint64_t last)
{
    std::vector<uint64_t> idx;
    uint64_t i = first;
    while (i < last) { idx.push_back(i); uint64_t step = i / 20; if (step < 100) step = 100; i += step; }
    idx.push_back(last);
    return idx;
}

static void init_plot(LinePlot& p, size_t n, int series)
{
    static const char* const names[5] = {"A", "T", "G", "C", "Mean"};
    p.x.assign(n, 0);
    p.names.assign(names, names + series);
    p.y.assign((size_t)series, std::vector<float>(n, 0.f));
}

static void end_plots(int bc, const std::vector<uint64_t>& q, const std::vector<uint64_t>& c, LinePlot& qual, LinePlot& cont)
{
    init_plot(qual, (size_t)bc, 5);
    init_plot(cont, (size_t)bc, 4);
    for (int i = 0; i < bc; i++) {
        qual.x[(size_t)i] = cont.x[(size_t)i] = i + 1;
        const uint64_t* cq = &q[(size_t)i * 5];
        const uint64_t* cc = &c[(size_t)i * 5];
        for (int j = 0; j < 4; j++) {
            qual.y[(size_t)j][(size_t)i] = cc[j] > 0 ? static_cast<float>(cq[j]) / cc[j] : 0.f;
            cont.y[(size_t)j][(size_t)i] = cc[4] > 0 ? static_cast<float>(cc[j] * 100) / cc[4] : 0.f;
        }
        qual.y[4][(size_t)i] = cc[4] > 0 ? static_cast<float>(cq[4]) / cc[4] : 0.f;
    }
}

void side_stats(int bc_len, const std::vector<int>& lens, uint64_t bases, const SideTables& t, SideStats& out)
{
    const int num = (int)lens.size();
    const int lmin = lens.front(), lmax = lens.back();
    out.tab[0] = std::to_string(num);
    out.tab[1] = std::to_string(bases);
    out.tab[3] = std::to_string(lmin);
    out.tab[4] = std::to_string(lmax);
    out.tab[5] = std::to_string((int)(bases / (uint64_t)num));
    out.tab[6] = std::to_string(lens[(size_t)(num / 2)]);
    out.tab[7] = std::to_string(n50(lens, bases));

    // ---- Get_plot_line_data ----
    const std::vector<uint64_t> idx = axis(1, (uint64_t)lmax);
    const uint64_t vmax = idx.size() - 1;
    // lenIndexs / vecIndexs are unordered_maps read with operator[]: a missing key reads as 0 (:2676-2677)
    std::unordered_map<uint64_t, uint64_t> vec_index;
    for (uint64_t i = 0; i < vmax; i++) vec_index[idx[i]] = i;
    auto len_index = [&](uint64_t x) -> uint64_t {
        if (x == (uint64_t)lmax) return idx.back();
        if (x < idx[0] || x > (uint64_t)lmax) return 0;
        size_t lo = 0, hi = idx.size() - 1;             // idx[lo] <= x < idx[hi]
        while (hi - lo > 1) { size_t mid = (lo + hi) / 2; if (idx[mid] <= x) lo = mid; else hi = mid; }
        return idx[lo];
    };
    auto vec_of = [&](uint64_t row) -> uint64_t {
        auto it = vec_index.find(len_index(row * 100 + 1));
        return it == vec_index.end() ? 0 : it->second;
    };
    std::vector<uint64_t> mq(vmax * 5, 0), mc(vmax * 5, 0);
    if (vmax > 0)
        for (uint64_t j = 0; j < t.bin_rows; j++) {
            const uint64_t v = vec_of(j);
            for (int x = 0; x < 5; x++) { mc[v * 5 + x] += t.bin_cnt[j * 5 + x]; mq[v * 5 + x] += t.bin_qual[j * 5 + x]; }
        }
    std::vector<uint64_t> q5((size_t)bc_len * 5, 0), c5((size_t)bc_len * 5, 0), q3((size_t)bc_len * 5, 0), c3((size_t)bc_len * 5, 0);
    for (uint64_t j = 0; j < t.end_rows && j < (uint64_t)bc_len; j++)
        for (int x = 0; x < 5; x++) {
            c5[j * 5 + x] += t.c5[j * 5 + x]; q5[j * 5 + x] += t.q5[j * 5 + x];
            q3[j * 5 + x] += t.q3[j * 5 + x];
            c3[j * 5 + x] += 2 * t.c3[j * 5 + x];        // the 3' counts are added twice, :2710-2725
        }

    init_plot(out.reads_qual, vmax, 5);
    init_plot(out.contents, vmax, 4);
    uint64_t gc_sum = 0, qual_sum = 0;
    for (uint64_t i = 0; i < vmax; i++) {
        out.reads_qual.x[i] = out.contents.x[i] = (int)idx[i];
        const uint64_t* cc = &mc[i * 5];
        const uint64_t* cq = &mq[i * 5];
        gc_sum += cc[2] + cc[3];
        qual_sum += cq[4];
        for (int j = 0; j < 4; j++) {
            out.reads_qual.y[(size_t)j][i] = cc[j] > 0 ? static_cast<float>(cq[j]) / cc[j] : 0.f;
            out.contents.y[(size_t)j][i] = cc[4] > 0 ? static_cast<float>(cc[j] * 100) / cc[4] : 0.f;
        }
        out.reads_qual.y[4][i] = cc[4] > 0 ? static_cast<float>(cq[4]) / cc[4] : 0.f;
    }
    out.gc = static_cast<float>(gc_sum * 100) / bases;
    out.mean_qual = static_cast<float>(qual_sum) / bases;
    end_plots(bc_len, q5, c5, out.reads_qual5, out.contents5);
    end_plots(bc_len, q3, c3, out.reads_qual3, out.contents3);
    out.tab[2] = limit_decimals(std::round(out.gc * 1000) / 1000.0, 3);
    out.tab[8] = limit_decimals(std::round(out.mean_qual * 1000) / 1000.0, 3);

    // ---- Get_length_Dis ----
    {
        const std::vector<uint64_t> li = axis((uint64_t)lmin, (uint64_t)lmax);
        const int vm = (int)li.size() - 1;
        out.len_dis.x.assign((size_t)std::max(vm, 0), 0);
        out.len_dis.y.assign((size_t)std::max(vm, 0), 0);
        if (vm > 0) {
            std::unordered_map<uint64_t, uint64_t> cnt;
            for (int len : lens) {
                uint64_t key;
                if (len == lmax) key = li[(size_t)vm - 1];
                else {
                    size_t lo = 0, hi = li.size() - 1;
                    while (hi - lo > 1) { size_t mid = (lo + hi) / 2; if (li[mid] <= (uint64_t)len) lo = mid; else hi = mid; }
                    key = li[lo];
                }
                cnt[key]++;
            }
            for (int i = 0; i < vm; i++) { out.len_dis.x[(size_t)i] = (int)li[(size_t)i]; out.len_dis.y[(size_t)i] = cnt[li[(size_t)i]]; }
        }
    }
    // ---- Get_qual_Dis ----
    {
        int maxq = 0;
        for (int j = 0; j < 256; j++) if (t.diff_qual[j] > 0) maxq = j;
        out.qual_dis.x.resize((size_t)maxq + 1);
        out.qual_dis.y.resize((size_t)maxq + 1);
        for (int i = 0; i <= maxq; i++) {
            out.qual_dis.x[(size_t)i] = i;
            out.qual_dis.y[(size_t)i] = static_cast<float>(t.diff_qual[i] * 100) / bases;
        }
    }
}

// ---------------------------------------------------------------------------
// text
// ---------------------------------------------------------------------------
template <class T>
static std::string json_array(const std::vector<T>& v)
{
    std::stringstream ss;
    ss << "[";
    for (size_t i = 0; i < v.size(); i++) { ss << v[i]; if (i + 1 < v.size()) ss << ","; }
    ss << "]";
    return ss.str();
}

template <class T>
static int max_y(const std::vector<T>& v, int m = 0)
{
    for (const T& x : v) if ((long long)x > (long long)m) m = (int)x;       // int maxY compared/assigned like getMaxY (report.cpp:508-546)
    return m;
}

static int y_title_gap(int maxy)
{
    const std::string s = std::to_string(maxy);
    int size = (int)s.size();
    for (size_t i = 3; i < s.size(); i += 3) size += 1;
    return size * 5 + 30;
}

static void js_len(std::ostream& os, const char* key, const LenDis& d)
{
    os << key << ": {x: " << json_array(d.x) << ",\ny: " << json_array(d.y) << ",\nyTitleGap: " << y_title_gap(max_y(d.y)) << ",\n},\n";
}
static void js_qual(std::ostream& os, const char* key, const QualDis& d)
{
    os << key << ": {x: " << json_array(d.x) << ",\ny: " << json_array(d.y) << ",\nyTitleGap: " << y_title_gap(max_y(d.y)) << ",\n},\n";
}
static void js_line(std::ostream& os, const char* key, const LinePlot& d)
{
    os << key << ": {x: " << json_array(d.x) << ",y: [";
    int m = 0;
    for (size_t s = 0; s < d.y.size(); s++) {
        os << "{name:\"" << d.names[s] << "\", data:" << json_array(d.y[s]) << ", }, ";
        m = max_y(d.y[s], m);
    }
    os << "], yTitleGap:" << y_title_gap(m) << ", },";
}

static void table_row(std::ostream& os, const std::string& qc, const std::string& a, const std::string& b, const std::string& c)
{
    os << "<tr>\n    <td>" << a << "</td>\n";
    if (qc[1] != '1') os << "    <td>" << b << "</td>\n";
    if (qc[1] != '0') os << "    <td>" << c << "</td>\n";
    os << "</tr>\n";
}

// The document around the numbers: pieces of reports the reference wrote, kept as data files beside the program
// (tools/make_report_assets.py; include/report.cpp:670-700 assembles the same pieces in the same order).
static bool read_asset(const std::string& name, std::string& out)
{
    std::string dir;
    if (const char* e = getenv("TGSF_ASSETS")) dir = e;
    else {
        char exe[4096];
        const ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
        if (n <= 0) return false;
        dir.assign(exe, (size_t)n);
        const size_t slash = dir.rfind('/');
        dir = dir.substr(0, slash == std::string::npos ? 0 : slash);
        // tgsfilter_amd/bin/tgsfilter and tests/emul/tgsfilter_emul both find tgsfilter_amd/host/assets
        std::ifstream probe(dir + "/../host/assets/report_head.html");
        dir += probe ? "/../host/assets" : "/../../tgsfilter_amd/host/assets";
    }
    std::ifstream f(dir + "/" + name, std::ios::binary);
    if (!f) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}

static void write_data(std::ostream& os, const std::string& qc, const SideStats& raw, const SideStats& clean)
{
    const bool fastq = qc[0] == '1', has_raw = qc[1] != '1', has_clean = qc[1] != '0';
    os << "<script>\nvar data = {\n";
    if (has_raw) {
        js_len(os, "rawLenDis", raw.len_dis);
        js_line(os, "rawBasesContents", raw.contents);
        js_line(os, "raw5pBasesContents", raw.contents5);
        js_line(os, "raw3pBasesContents", raw.contents3);
        if (fastq) {
            js_qual(os, "rawQualDis", raw.qual_dis);
            js_line(os, "rawReadsQual", raw.reads_qual);
            js_line(os, "raw5pReadsQual", raw.reads_qual5);
            js_line(os, "raw3pReadsQual", raw.reads_qual3);
        }
    }
    if (has_clean) {
        js_len(os, "cleanLenDis", clean.len_dis);
        js_line(os, "cleanBasesContents", clean.contents);
        js_line(os, "clean5pBasesContents", clean.contents5);
        js_line(os, "clean3pBasesContents", clean.contents3);
        if (fastq) {
            js_qual(os, "cleanQualDis", clean.qual_dis);
            js_line(os, "cleanReadsQual", clean.reads_qual);
            js_line(os, "clean5pReadsQual", clean.reads_qual5);
            js_line(os, "clean3pReadsQual", clean.reads_qual3);
        }
    }
    os << "}\n</script>\n";
}

void write_report(std::ostream& os, const std::string& qc, const SideStats& raw, const SideStats& clean)
{
    static const char* const labels[9] = {"Total reads", "Total bases", "GC content (%)", "Min length (bp)", "Max length (bp)",
                                          "Mean length (bp)", "Median length (bp)", "N50 length (bp)", "Mean quality"};
    const bool fastq = qc[0] == '1', has_raw = qc[1] != '1', has_clean = qc[1] != '0';
    std::time_t now = std::time(nullptr);
    char ts[64];
    std::strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S", std::localtime(&now));
    std::string head, plots, charts, tail;
    if (read_asset("report_head.html", head) && read_asset("report_plots_" + qc + ".html", plots) &&
        read_asset("report_charts.js", charts) && read_asset("report_tail.html", tail)) {
        // the reference's document (self-contained: the chart code travels inside it), include/report.cpp:670-700
        os << head << "<table>\n";
        table_row(os, qc, "", qc[1] == '0' ? "Value" : "Before filtering", "After filtering");
        for (int i = 0; i < (fastq ? 9 : 8); i++) table_row(os, qc, labels[i], has_raw ? raw.tab[i] : "0", has_clean ? clean.tab[i] : "0");
        os << "</table>\n" << plots;
        os << "<div id=\"footer\">    <p>Generated by <a href=\"https://github.com/HuiyangYu/TGSFilter\" target=\"blank\">TGSFilter (v1.11)</a> at "
           << ts << "</p></div>";                       // include/report.cpp:105-110
        os << charts;
        write_data(os, qc, raw, clean);
        os << tail;
        return;
    }
    // the asset files are not beside the program: a plain document with the same table and data, charts from the network
    std::cerr << "Warning: report assets not found (tgsfilter_amd/host/assets); writing the plain report" << std::endl;
    os << "<html lang=\"en\">\n<head>\n<meta charset=\"utf-8\">\n<title>TGSFilter report</title>\n"
          "<style>body{font-family:sans-serif;margin:0}h1,#footer{padding:20px 10px;background:skyblue;color:#fff}"
          ".container{padding:20px}.level-2-title{margin:20px 0;font-size:28px;font-weight:bold;color:rgb(46,163,209)}"
          "table{border-collapse:collapse}td{border:1px solid #ccc;padding:6px 14px}.plot{width:48%;height:420px;display:inline-block}</style>\n"
          "</head>\n<body>\n<div id=\"container\" class=\"container\">\n<h1>TGSFilter report</h1>\n"
          "<div class=\"level-2-title\">Summary</div>\n<div>\n<table>\n";
    table_row(os, qc, "", qc[1] == '0' ? "Value" : "Before filtering", "After filtering");
    for (int i = 0; i < (fastq ? 9 : 8); i++) table_row(os, qc, labels[i], has_raw ? raw.tab[i] : "0", has_clean ? clean.tab[i] : "0");
    os << "</table>\n</div>\n<div class=\"level-2-title\">Plots</div>\n<div id=\"plots\"></div>\n</div>\n";
    os << "<div id=\"footer\">Generated by tgsfilter (MI355X build) at " << ts << "</div>\n</body>\n";
    os << "<script src=\"https://cdn.jsdelivr.net/npm/echarts@5/dist/echarts.min.js\"></script>\n";
    write_data(os, qc, raw, clean);
    // chart glue (this repo's own): one chart per entry of `data`
    os << "<script>\n"
          "if (typeof echarts !== 'undefined') {\n"
          "  var host = document.getElementById('plots');\n"
          "  Object.keys(data).forEach(function (key) {\n"
          "    var d = data[key], div = document.createElement('div');\n"
          "    div.className = 'plot'; host.appendChild(div);\n"
          "    var series = Array.isArray(d.y) && typeof d.y[0] === 'object'\n"
          "      ? d.y.map(function (s) { return {name: s.name, type: 'line', showSymbol: false, data: s.data}; })\n"
          "      : [{name: key, type: key.indexOf('LenDis') >= 0 ? 'bar' : 'line', showSymbol: false, data: d.y}];\n"
          "    echarts.init(div).setOption({title: {text: key}, tooltip: {trigger: 'axis'}, legend: {top: 24},\n"
          "      grid: {left: d.yTitleGap + 20}, xAxis: {type: 'category', data: d.x}, yAxis: {type: 'value'}, series: series});\n"
          "  });\n"
          "}\n</script>\n</html>\n";
}

}  // namespace host
