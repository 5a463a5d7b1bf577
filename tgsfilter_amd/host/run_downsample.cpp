// run_downsample.cpp -- downsampling (DownSampleTask, src/TGSFilter.cpp:2164-2568) behind the filter pass (run.h): keep the longest
// reads until the target is met (:2297-2344), then a QC-only pass over the kept reads (CalcAvgQuality / Get_5p/3p_base_qual
// again, :2436-2447) which also writes them, in input order.
#include "run.h"

namespace host {

static std::string full_name(const CleanRec& c)
{
    std::string nm;
    append_name(nm, c.name, c.pass_num);
    return nm;
}

std::vector<char> Run::select_kept()
{
    // Selection as the reference makes it (:2297-2344), container for container, so that ties at the cut fall
    // the same way when both programs are built with the same standard library: lengths keyed by record name
    // in an unordered_map filled in write order (:2105, :2267; a repeated name keeps its last length), handed
    // over by copy (:3142, :2169), listed in the map's iteration order, std::sort by length (descending),
    // names taken from the top; the second pass keeps every record whose name was taken (:2356).
    // (one process per GPU: the selection is over the kept fragments of ALL ranks, in input order = rank order; rank 0
    // receives every rank's names and lengths and makes it, the others wait for their keep flags)
    std::vector<std::string> all_names;                            // rank 0 of a sharded job: every fragment of the job, in input order
    std::vector<int> all_lens;
    std::vector<size_t> rank_first;                                // ... and where each rank's begin
    if (sharded) {
        BlobOut mine;
        std::vector<uint32_t> lens;
        std::string names;
        for (const CleanRec& c : clean_recs) { lens.push_back(c.len); const std::string nm = full_name(c); const uint32_t n = (uint32_t)nm.size(); names.append((const char*)&n, 4); names += nm; }
        mine.vec(lens); mine.str(names);
        const std::vector<std::string> all = link.gather(mine.s);
        for (const std::string& b : all) {
            BlobIn in2(b);
            std::vector<uint32_t> l2; std::string n2;
            in2.vec(l2); in2.str(n2);
            rank_first.push_back(all_names.size());
            size_t at = 0;
            for (uint32_t L : l2) {
                uint32_t n = 0;
                if (at + 4 > n2.size()) die("a rank of the job sent a list of names shorter than its lengths");
                memcpy(&n, n2.data() + at, 4); at += 4;
                all_names.emplace_back(n2.data() + at, n); at += n;
                all_lens.push_back((int)L);
            }
        }
        rank_first.push_back(all_names.size());
    }
    const bool selects = !sharded || link.rank == 0;
    std::unordered_map<std::string, int> seq_lens;
    uint64_t total = 0;
    if (!sharded) for (const CleanRec& c : clean_recs) { seq_lens[full_name(c)] = (int)c.len; total += c.len; }
    else for (size_t i = 0; i < all_names.size(); i++) { seq_lens[all_names[i]] = all_lens[i]; total += (uint64_t)all_lens[i]; }
    const std::unordered_map<std::string, int> handed(seq_lens), task_lens(handed);
    std::vector<std::pair<std::string, int>> vec(task_lens.begin(), task_lens.end());
    std::sort(vec.begin(), vec.end(), [](const std::pair<std::string, int>& a, const std::pair<std::string, int>& b) {
        return a.second > b.second;
    });
    uint64_t desired = 0; int want_num = 0; bool by_size = true;
    if (o.genome_size > 0 && o.desired_depth > 0) desired = o.genome_size * (uint64_t)o.desired_depth;
    else if (o.desired_frac > 0) desired = (uint64_t)(o.desired_frac * total);        // float * uint64, :2322
    else { by_size = false; want_num = o.desired_num; }
    std::unordered_set<std::string> chosen;
    uint64_t added = 0; int added_num = 0;
    for (const auto& pr : vec) {
        if (!selects) break;
        chosen.insert(pr.first);
        added += (uint64_t)pr.second; added_num++;
        down_bases += (uint64_t)pr.second; down_lens.push_back(pr.second);
        if (by_size ? added >= desired : added_num >= want_num) break;
    }
    std::vector<char> keep(clean_recs.size(), 0);
    if (!sharded) for (size_t i = 0; i < clean_recs.size(); i++) keep[i] = chosen.count(full_name(clean_recs[i])) ? 1 : 0;
    else {
        std::vector<std::string> flags;
        if (link.rank == 0)
            for (int k = 0; k < link.world; k++) {
                std::string f(rank_first[(size_t)k + 1] - rank_first[(size_t)k], '\0');
                for (size_t i = 0; i < f.size(); i++) f[i] = chosen.count(all_names[rank_first[(size_t)k] + i]) ? 1 : 0;
                flags.push_back(std::move(f));
            }
        std::string mine;
        link.scatter(flags, mine);
        if (mine.size() != keep.size()) die("the selection handed to this rank does not fit its fragments");
        if (!keep.empty()) memcpy(keep.data(), mine.data(), keep.size());
        // (the job's totals, for rank 0's statistics: every fragment that entered the selection)
        if (link.rank == 0) { down_job_recs = all_names.size(); down_job_bases = total; }
        std::vector<std::string>().swap(all_names);
    }
    return keep;
}

void Run::downsample()
{
    const double t_d0 = now_s();
    if (o.downsample) {
        const std::vector<char> keep = select_kept();
        t_dsel = now_s() - t_d0;
        second_pass(keep);
    }
    { const double c0 = now_s(); if (!o.only_qc && !mapped_out) out.close(); t_dclose = now_s() - c0; }
}

}  // namespace host
