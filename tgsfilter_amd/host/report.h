// report.h -- end-of-run statistics and the HTML QC report.
// The numbers follow the reference's merge code bit for bit (Get_plot_line_data src/TGSFilter.cpp:2607-2856
// incl. the doubled 3' counts :2718-2725 and the default-0 map lookups :2676-2677; Get_length_Dis :2858-2898;
// Get_qual_Dis :2584-2605; Get_N50 :2900-2909; limitDecimalPlaces :2911-2921) and are emitted in the
// reference's table / `var data = {...}` text format (include/report.cpp:552-668).  The page around them
// (styles, chart glue) is this repo's own; ECharts is referenced, not embedded.
#pragma once
#include <cstdint>
#include <ostream>
#include <string>
#include <vector>

namespace host {

struct LinePlot {
    std::vector<int> x;
    std::vector<std::string> names;
    std::vector<std::vector<float>> y;
};
struct LenDis { std::vector<int> x; std::vector<uint64_t> y; };
struct QualDis { std::vector<int> x; std::vector<float> y; };

struct SideStats {            // "raw" or "clean" side of the report
    LenDis len_dis;
    QualDis qual_dis;
    LinePlot reads_qual, reads_qual5, reads_qual3;
    LinePlot contents, contents5, contents3;
    float gc = 0.f, mean_qual = 0.f;
    std::string tab[9];       // reads, bases, GC, min, max, mean, median, N50, mean quality
};

// tables: pointers into the library's tally vector ([rows][5] uint64 each)
struct SideTables {
    const uint64_t *bin_qual, *bin_cnt; uint64_t bin_rows;
    const uint64_t *q5, *c5, *q3, *c3; uint64_t end_rows;
    const uint64_t* diff_qual;          // [256]
};

// lens must be sorted ascending (the reference sorts before use, :3151, :3182)
void side_stats(int bc_len, const std::vector<int>& lens, uint64_t bases, const SideTables& t, SideStats& out);

std::string limit_decimals(double v, int places);

void write_report(std::ostream& os, const std::string& qc_type, const SideStats& raw, const SideStats& clean);

}  // namespace host
