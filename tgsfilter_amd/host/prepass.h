// prepass.h -- the parameter pre-pass (GetFilterParameterTask, src/TGSFilter.cpp:869-1216):
// quality encoding + default -q (P1), base-content-bias trim lengths (P2), adapter identification (P3).
// P1/P2 are a few integer passes over <= 100k read ends and stay on the host; P3 is 2 x 100k x 22
// edlib alignments and goes through tgsf_align_windows (the library's edlib-compatible entry point).
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "fastx.h"
#include "options.h"

namespace host {

extern const char* const kAdapterLib[22];         // the reference's adapter table, :2970-2991

std::string rev_comp(const std::string& s);       // rev_comp_seq + complement table, :859-867, :2954-2967

struct PrepassResult {
    int qtype = 0;                 // global qType (:80)
    int trim5p = 0, trim3p = 0;    // CheckBaseContent results
    std::string adapter5p, adapter3p;
    float depth5p = 0.f, depth3p = 0.f;
};

// Runs the pre-pass; prints the reference's INFO lines; may exit(-1) like Get_qType (:1060-1065).
// Updates o.min_q when -q was not given.
// next_record: one pass over the records of the input, in order (false at the end)
PrepassResult run_prepass(Options& o, const std::function<bool(Rec&)>& next_record);

}  // namespace host
