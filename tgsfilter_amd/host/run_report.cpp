// run_report.cpp -- the end of a run (run.h): the tallies of the contexts and of the ranks summed, statistics, INFO lines, the
// HTML report (src/TGSFilter.cpp:3146-3235, :3240-3279, :3285-3328), and TGSF_TIMING's lines.
#include "run.h"

#include <sys/stat.h>

namespace host {

std::vector<uint64_t> pack_rows(const std::vector<uint64_t>& v, int32_t bc2, uint32_t nb2)
{
    const size_t head = tgsf_ctr_bin_table(0, bc2, nb2);
    std::vector<uint64_t> out(v.begin(), v.begin() + (long)head);
    for (int b = 0; b < 4; b++) {
        const size_t at = tgsf_ctr_bin_table(b, bc2, nb2), n = (size_t)std::min<uint64_t>(v[TGSF_CTR_ROWS + (b >> 1)], nb2) * 5;
        out.insert(out.end(), v.begin() + (long)at, v.begin() + (long)(at + n));
    }
    return out;
}

void add_rows(std::vector<uint64_t>& v, const std::vector<uint64_t>& ru, int32_t bc2, uint32_t nb2)
{
    const size_t head = tgsf_ctr_bin_table(0, bc2, nb2);
    if (ru.size() < head) die("a rank of the job sent a tally vector of another layout");
    uint64_t rows[4];
    for (int q = 0; q < 4; q++) rows[q] = std::max(v[TGSF_CTR_ROWS + q], ru[TGSF_CTR_ROWS + q]);
    for (size_t i = 0; i < head; i++) v[i] += ru[i];
    size_t from = head;
    for (int b = 0; b < 4; b++) {
        const size_t at = tgsf_ctr_bin_table(b, bc2, nb2), n = (size_t)std::min<uint64_t>(ru[TGSF_CTR_ROWS + (b >> 1)], nb2) * 5;
        if (from + n > ru.size()) die("a rank of the job sent a tally vector of another layout");
        for (size_t i = 0; i < n; i++) v[at + i] += ru[from + i];
        from += n;
    }
    for (int q = 0; q < 4; q++) v[TGSF_CTR_ROWS + q] = rows[q];
}

// this context's tallies added into v (sums; the four "rows used" words are maxima); of the bin tables only the rows in use travel
static void add_context(const Api& L, tgsf_ctx* c, std::vector<uint64_t>& v, std::vector<uint64_t>& part, int32_t bc, uint32_t nbins)
{
    uint64_t used[2] = {0, 0};
    if (L.counters_used(c, part.data(), part.size(), used) != TGSF_OK) die(L.last_error(c));
    uint64_t rows[4];
    for (int k = 0; k < 4; k++) rows[k] = std::max(v[TGSF_CTR_ROWS + k], part[TGSF_CTR_ROWS + k]);
    const size_t head = tgsf_ctr_bin_table(0, bc, nbins);
    for (size_t i = 0; i < head; i++) v[i] += part[i];
    for (int b = 0; b < 4; b++) {
        const size_t at = tgsf_ctr_bin_table(b, bc, nbins), n = (size_t)used[b >> 1] * 5;
        for (size_t i = 0; i < n; i++) v[at + i] += part[at + i];
    }
    for (int k = 0; k < 4; k++) v[TGSF_CTR_ROWS + k] = rows[k];
}

// One process per GPU: the job's tallies = the sum over the ranks (src/TGSFilter.cpp:3208-3213 across GPUs).
//   * Every rank sums its contexts on the host and sends rank 0 the rows in use over the ranks' sockets: rank 0's sum of those is
//     what the job reports when nothing else is available (ranks sharing a GPU, no RCCL on the box, TGSF_SHARD_EXCHANGE=socket).
//   * With a GPU per rank the contexts are also folded into one vector in HBM and the job makes ONE all-reduce of it over
//     RCCL / xGMI (include/tgsf_rccl.h; every rank entered tgsf_create with the same table rows).  The collective runs on a
//     helper thread under a deadline -- a fabric that does not answer must not hold a finished job -- and rank 0 compares its
//     result, row for row, with the sum that came over the sockets (a few hundred KB: it costs nothing to have both).  They are
//     the same numbers by construction; until the collective path has run on real multi-GPU hardware often enough, a
//     difference (a layout slip, a rank that entered with other rows) is reported and the sockets' sum is what the job uses
//     (ADVICE r5).  TGSF_SHARD_EXCHANGE=rccl turns every such fall-back into an error (the tests' way of seeing that RCCL ran).
void Run::sum_tallies()
{
    const Api& L = *api;
    tgsf_ctx* ctx = ctxs[0];
    void* rccl_comm = nullptr;
    int rccl_rc = TGSF_OK;
    std::string rccl_err;
    L.counters_len(ctx, &nw, &bc, &nbins);
    t.assign(nw, 0);
    std::vector<uint64_t> part(nw);
    t_x0 = now_s();
    own_raw_bases = raw_bases;                                         // (rank 0's raw_bases becomes the job's below)
    const char* ex = getenv("TGSF_SHARD_EXCHANGE");
    const bool must_rccl = ex && !strcmp(ex, "rccl");
    for (tgsf_ctx* c : ctxs) add_context(L, c, t, part, bc, nbins);    // this rank's (or this process's) own sums, before anything is merged on the device
    if (!sharded) return;

    if (use_rccl) {
        // The communicator has had the whole run to come up.  One that is still not there some time after the filtering is
        // over (a peer that cannot be reached, a fabric that does not answer) must not hold the job for ever: this rank says
        // so below, every rank then goes on without it, and the helper is left where it waits (the process leaves with _exit).
        double rccl_patience = 120.0;
        if (const char* e = knob("TGSF_RCCL_INIT_TIMEOUT_S")) rccl_patience = atof(e);          // test knob
        while (!rccl_state->done.load(std::memory_order_acquire) && now_s() - t_x0 < rccl_patience) usleep(2000);
        if (rccl_state->done.load(std::memory_order_acquire)) {
            rccl_up.join();
            rccl_rc = rccl_state->rc; rccl_err = rccl_state->err; rccl_comm = rccl_state->comm;
        } else {
            rccl_up.detach();
            rccl_rc = TGSF_E_HIP;
            rccl_err = "the communicator was not up " + std::to_string((int)rccl_patience) + " s after the filtering ended";
        }
        t_rccl_wait = now_s() - t_x0;
        // the communicator came up on every rank, or nobody uses it
        if (link.max_u64(rccl_rc != TGSF_OK ? 1 : 0) != 0) {
            if (must_rccl) die("RCCL communicator: " + (rccl_err.empty() ? std::string("it failed on another rank") : rccl_err));
            if (rccl_rc != TGSF_OK) std::cerr << "Warning: rank " << link.rank << ": RCCL communicator: " << rccl_err << " -- the tallies are summed over the ranks' sockets" << std::endl;
            // (a communicator that did come up here is left as it is: taking it down may wait for peers that are stuck)
            rccl_comm = nullptr;
            use_rccl = false;
        }
    }
    std::vector<uint64_t> reduced;                                     // the all-reduced vector as this rank's device holds it
    if (use_rccl) {
        for (size_t k = 1; k < ctxs.size(); k++)
            if (L.counters_merge(ctxs[0], ctxs[k]) != TGSF_OK) die(L.last_error(ctxs[0]));
        // the collective on a helper thread, under a deadline of its own (a peer that died between the communicator's
        // start and here, a link that hangs): the helper owns what it touches, and is left behind if it does not come back
        struct Reduce { std::atomic<int> done{0}; int rc = TGSF_OK; std::string err; double s = 0; };
        const std::shared_ptr<Reduce> rd = std::make_shared<Reduce>();
        double patience = 60.0;
        if (const char* e = knob("TGSF_RCCL_ALLREDUCE_TIMEOUT_S")) patience = atof(e);              // test knob
        const double a0 = now_s();
        std::thread helper([R = R, rd, c0 = ctxs[0], rccl_comm, rank = link.rank, world = link.world, a0] {
            if (const char* e = knob("TGSF_RCCL_ALLREDUCE_STALL_S")) usleep((useconds_t)(atof(e) * 1e6));   // test knob: a collective that hangs
            rd->rc = R->allreduce_counters(c0, rccl_comm, rank, world, 0, nullptr);
            if (rd->rc != TGSF_OK) rd->err = R->last_error();          // (thread-local text: taken on this thread)
            rd->s = now_s() - a0;
            rd->done.store(1, std::memory_order_release);
        });
        while (!rd->done.load(std::memory_order_acquire) && now_s() - a0 < patience) usleep(500);
        int bad = 0;
        if (rd->done.load(std::memory_order_acquire)) {
            helper.join();
            t_allreduce = rd->s;
            if (rd->rc != TGSF_OK) { bad = 1; rccl_err = "tally all-reduce: " + rd->err; }
        } else {
            helper.detach();
            bad = 1;
            rccl_err = "the tally all-reduce had not returned after " + std::to_string((int)patience) + " s";
        }
        if (link.max_u64((uint64_t)bad) != 0) {
            if (must_rccl) die(rccl_err.empty() ? std::string("tally all-reduce: it failed on another rank") : rccl_err);
            if (bad) std::cerr << "Warning: rank " << link.rank << ": " << rccl_err << " -- the tallies summed over the ranks' sockets are used" << std::endl;
            use_rccl = false;                                          // (the communicator is left as it is)
        } else {
            (void)R->comm_count(rccl_comm, &rccl_ranks);
            reduced.assign(nw, 0);
            add_context(L, ctxs[0], reduced, part, bc, nbins);        // (it holds the whole job's totals now, on every rank)
        }
    }
    // Rank 0 receives every rank's read lengths (the statistics need them sorted: N50 and the like) and its tally rows in
    // use; it alone prints the run's statistics and writes the report.
    BlobOut mine;
    mine.pod(raw_bases); mine.pod(clean_bases);
    mine.vec(raw_lens); mine.vec(clean_lens);                          // (each sorted already, beside the pipeline)
    mine.vec(pack_rows(t, bc, nbins));
    const std::vector<std::string> all = link.gather(mine.s);
    for (int k = 1; k < (int)all.size(); k++) {                        // (rank 0 only)
        BlobIn in2(all[(size_t)k]);
        uint64_t rb = 0, cb = 0;
        std::vector<int> rl, cl;
        std::vector<uint64_t> ru;
        in2.pod(rb); in2.pod(cb); in2.vec(rl); in2.vec(cl); in2.vec(ru);
        raw_bases += rb; clean_bases += cb;
        const size_t r0 = raw_lens.size(), c0 = clean_lens.size();
        raw_lens.insert(raw_lens.end(), rl.begin(), rl.end());
        clean_lens.insert(clean_lens.end(), cl.begin(), cl.end());
        std::inplace_merge(raw_lens.begin(), raw_lens.begin() + (long)r0, raw_lens.end());
        if (!o.downsample) std::inplace_merge(clean_lens.begin(), clean_lens.begin() + (long)c0, clean_lens.end());   // (a downsampling run reports the selected reads' lengths instead)
        add_rows(t, ru, bc, nbins);
    }
    tally_route = "summed on rank 0 over the ranks' sockets";
    if (use_rccl) {
        // the same numbers twice: the devices' all-reduce and (rank 0) the sum that came over the sockets
        tally_route = "RCCL all-reduce on the devices";
        if (link.rank == 0) {
            if (pack_rows(reduced, bc, nbins) == pack_rows(t, bc, nbins)) { t.swap(reduced); tally_route += ", equal to the sum over the ranks' sockets"; }
            else {
                if (must_rccl) die("the all-reduced tally vector differs from the sum of the ranks' vectors over their sockets");
                std::cerr << "Warning: the all-reduced tally vector differs from the sum of the ranks' vectors over their sockets -- the sockets' sum is used" << std::endl;
                tally_route += " (DIFFERENT from the sum over the ranks' sockets, which is used)";
            }
        }
    }
    if (timing)
        fprintf(stderr, "SHARD %d/%d: bytes [%zu, %zu) of the text on device %d -> %s | tallies: %s (communicator ready after %.3f s of waiting, all-reduce %.4f s, %d ranks in it), exchange + gather %.3f s\n",
                link.rank, link.world, text_off, text_off + text_size, o.device, out_path.c_str(), tally_route.c_str(), t_rccl_wait, t_allreduce, rccl_ranks, now_s() - t_x0);
    if (use_rccl) R->comm_destroy(rccl_comm);
}

void Run::report()
{
    auto tables = [&](const std::vector<uint64_t>& v, bool clean) {
        SideTables s;
        s.bin_qual = &v[tgsf_ctr_bin_table(clean ? TGSF_B_CLEAN_QUAL : TGSF_B_RAW_QUAL, bc, nbins)];
        s.bin_cnt = &v[tgsf_ctr_bin_table(clean ? TGSF_B_CLEAN_CNT : TGSF_B_RAW_CNT, bc, nbins)];
        s.bin_rows = v[TGSF_CTR_ROWS + (clean ? 1 : 0)];
        s.q5 = &v[tgsf_ctr_end_table(clean ? TGSF_T_CLEAN5P_QUAL : TGSF_T_RAW5P_QUAL, bc)];
        s.c5 = &v[tgsf_ctr_end_table(clean ? TGSF_T_CLEAN5P_CNT : TGSF_T_RAW5P_CNT, bc)];
        s.q3 = &v[tgsf_ctr_end_table(clean ? TGSF_T_CLEAN3P_QUAL : TGSF_T_RAW3P_QUAL, bc)];
        s.c3 = &v[tgsf_ctr_end_table(clean ? TGSF_T_CLEAN3P_CNT : TGSF_T_RAW3P_CNT, bc)];
        s.end_rows = v[TGSF_CTR_ROWS + (clean ? 3 : 2)];
        s.diff_qual = &v[clean ? TGSF_CTR_CLEAN_DIFFQ : TGSF_CTR_RAW_DIFFQ];
        return s;
    };
    SideStats raw, clean;
    const int clean_num = (int)clean_lens.size();
    if (run_filter_pass && reports()) {
        if (raw_lens.empty()) die("no reads in the input");
        std::sort(raw_lens.begin(), raw_lens.end());
        side_stats(bc, raw_lens, raw_bases, tables(t, false), raw);
        if (!o.only_qc && !o.downsample) {
            if (clean_lens.empty()) die("no reads passed the filters");  // the reference dereferences an empty vector here (:3183)
            std::sort(clean_lens.begin(), clean_lens.end());
            side_stats(bc, clean_lens, clean_bases, tables(t, true), clean);
        }
        const uint64_t* d = &t[TGSF_CTR_DROPINFO];
        std::cerr << "INFO: " << raw_lens.size() << " reads with a total of " << raw_bases << " bases were input." << std::endl;
        if (!o.only_qc) {
            std::cerr << "INFO: " << d[0] << " reads were discarded with " << d[1] << " bases due to low quality." << std::endl;
            std::cerr << "INFO: " << d[2] << " reads have adapter at 5', 3' and middle." << std::endl;
            std::cerr << "INFO: " << d[3] << " reads have adapter at 5' and middle." << std::endl;
            std::cerr << "INFO: " << d[4] << " reads have adapter at 3' and middle." << std::endl;
            std::cerr << "INFO: " << d[5] << " reads have adapter at 5' and 3' end." << std::endl;
            std::cerr << "INFO: " << d[6] << " reads only have adapter at middle." << std::endl;
            std::cerr << "INFO: " << d[7] << " reads only have adapter at 5' end." << std::endl;
            std::cerr << "INFO: " << d[8] << " reads only have adapter at 3' end." << std::endl;
            std::cerr << "INFO: " << d[9] << " reads didn't have any adapter." << std::endl;
            std::cerr << "INFO: " << d[10] << " bases were trimmed due to the adapter or base content bias." << std::endl;
            std::cerr << "INFO: " << d[11] << " reads were discarded with " << d[12] << " bases due to the short length." << std::endl;
            std::cerr << "INFO: " << d[13] << " reads were discarded with " << d[14] << " bases due to low quality after split." << std::endl;
            if (o.min_repeat > 0)
                std::cerr << "INFO: " << d[15] << " reads were discarded with " << d[16] << " bases due to short repeat length." << std::endl;
            std::cerr << "INFO: " << clean_num << " reads with a total of " << clean_bases << " bases after filtering." << std::endl;
            if (!o.downsample && !o.out_file.empty()) {
                if (!sharded) std::cerr << "INFO: Filtered reads were written to: " << o.out_file << "." << std::endl;
                else std::cerr << "INFO: Filtered reads were written to: " << o.out_file << ".part0 ... " << o.out_file << ".part" << link.world - 1
                               << " (" << link.world << " parts; concatenated in this order they are the reads in input order)." << std::endl;
            }
        }
    }
    // The parts of a job are listed in <out>.parts (written by rank 0 when everything is there).  Parts of an EARLIER job with
    // more ranks would end up in a `cat <out>.part*` beside this job's: the ones that job's own list names go, with a word; a
    // file that merely has such a name and that no list of this tool names is the user's -- it stays, with a warning.
    if (reports() && sharded && !o.out_file.empty() && !o.only_qc && !o.only_adapters) {
        const std::string list = o.out_file + ".parts";
        std::vector<std::string> earlier;
        {
            std::ifstream f(list);
            std::string line;
            if (f && std::getline(f, line) && line == "# tgsfilter parts, in order")
                while (std::getline(f, line)) if (!line.empty()) earlier.push_back(line);
        }
        std::vector<std::string> mine;
        for (int k = 0; k < link.world; k++) mine.push_back(o.out_file + ".part" + std::to_string(k));
        for (const std::string& e : earlier) {
            if (std::find(mine.begin(), mine.end(), e) != mine.end()) continue;
            // (only names of this output's own numbered parts: a list is text anyone can edit)
            const std::string stem = o.out_file + ".part";
            if (e.compare(0, stem.size(), stem) != 0 || e.size() == stem.size() || e.find_first_not_of("0123456789", stem.size()) != std::string::npos) continue;
            struct stat st;
            if (lstat(e.c_str(), &st) == 0 && S_ISREG(st.st_mode) && unlink(e.c_str()) == 0)
                std::cerr << "Warning: " << e << ", a part of an earlier job with more ranks, was removed." << std::endl;
        }
        for (int k = link.world; k < link.world + 1024; k++) {
            const std::string other = o.out_file + ".part" + std::to_string(k);
            struct stat st;
            if (lstat(other.c_str(), &st) != 0) { if (k >= link.world + 64) break; else continue; }
            std::cerr << "Warning: " << other << " exists and is not a part of this job (nor of an earlier one this tool listed): `cat " << o.out_file
                      << ".part*` would take it in -- the parts of this job are listed in " << list << "." << std::endl;
        }
        std::ofstream f(list, std::ios::trunc);
        f << "# tgsfilter parts, in order\n";
        for (const std::string& m : mine) f << m << "\n";
    }
    if (o.downsample && reports()) {                                     // :3240-3279
        if (down_lens.empty()) die("no reads to downsample");
        std::sort(down_lens.begin(), down_lens.end());
        side_stats(bc, down_lens, down_bases, tables(down_t, false), clean);
        clean.tab[8] = limit_decimals(std::round(clean.mean_qual * 1000) / 1000.0, 2);      // two places here, :3268
        if (!o.filter)
            std::cerr << "INFO: " << (sharded ? down_job_recs : clean_recs.size()) << " reads with a total of " << (sharded ? down_job_bases : clean_bases) << " bases were input." << std::endl;
        std::cerr << "INFO: " << down_lens.size() << " reads with a total of " << down_bases << " bases after downsampling." << std::endl;
        if (!o.out_file.empty()) {
            if (!sharded) std::cerr << "INFO: Downsampled reads were written to: " << o.out_file << "." << std::endl;
            else std::cerr << "INFO: Downsampled reads were written to: " << o.out_file << ".part0 ... " << o.out_file << ".part" << link.world - 1
                           << " (" << link.world << " parts; concatenated in this order they are the reads in input order)." << std::endl;
        }
    }
    std::string qc = fasta_in ? "0" : "1";                             // :3286-3291
    qc += o.only_qc ? "0" : ((!o.filter && o.downsample) ? "1" : "2"); // :3293-3299
    if (reports()) {
        std::ofstream ofs(html);
        write_report(ofs, qc, raw, clean);
        ofs.close();
        std::cerr << "INFO: Quality control report was written to: " << html << "." << std::endl;
    }
}

void Run::timing_lines()
{
    const Api& L = *api;
    if (timing) {
        fprintf(stderr, "POOL: %zu jobs, busy %.3f, freeing job state %.3f, longest job %.3f, first job at %.3f, last job done at %.3f (pipeline start = 0, planner done at %.3f)\n",
                pool->jobs_, t_busy, pool->destroy_, pool->longest_, pool->first_ - t_p0, pool->last_ - t_p0, t_f0 - t_p0);
        fprintf(stderr, "TIMING: total %.3f s | index+prepass %.3f | waiting for the library %.3f (load %.3f + device %.3f, beside the pre-pass) | "
                        "pipeline %.3f (batching %.3f, tgsf_submit summed over %zu feeders %.3f, plan+write %.3f, planner waiting %.3f, "
                        "first batch filtered after %.3f, fill tail %.3f, closing the output %.3f; stages overlap) | stats+report %.3f | %s (fallocate %.3f, mapping the reserved pages %.3f, fill threads busy %.3f summed)\n",
                now_s() - t_start, t_prepass, t_libwait, t_load, t_dev, t_pipe, t_parse, ctxs.size(), t_gpu, t_write, t_widle, t_first,
                t_fill_tail, t_close, now_s() - t_p0 - t_pipe, mapped_out ? "output: fallocate + mapped fill" : "output: writev", sink.t_falloc, reserver->t_populate_wait, t_busy);
        fprintf(stderr, "RESERVE: planner waited %.3f s for pages of the output file\n", t_drain);
    }
    if (timing) {
        // kernel time of the run: the stage durations of every batch (HIP events inside the library), summed over the
        // contexts -- batches of different contexts overlap, so this is an upper bound of the time the GPU was busy
        float tot[TGSF_N_STAGES] = {0};
        uint32_t nb = 0;
        for (tgsf_ctx* c : ctxs) {
            float ms[TGSF_N_STAGES]; uint32_t n1 = 0;
            if (c && L.stage_times(c, ms, &n1) == TGSF_OK) { nb += n1; for (int i = 0; i < TGSF_N_STAGES; i++) tot[i] += ms[i]; }
        }
        double sum = 0;
        for (int i = 0; i < TGSF_N_STAGES; i++) sum += tot[i];
        // (ONE write for the line: the ranks of a sharded job share this stderr, and a line put together from several writes gets
        // another rank's lines into its middle)
        char piece[256];
        snprintf(piece, sizeof piece, "GPU: kernels %.3f s summed over %zu contexts and %u batches (upper bound of the busy time: contexts overlap) = %.4f of the run |", sum * 1e-3, ctxs.size(), nb,
                 sum * 1e-3 / std::max(1e-9, now_s() - t_start));
        std::string line = piece;
        for (int i = 0; i < TGSF_N_STAGES; i++) if (tot[i] > 0) { snprintf(piece, sizeof piece, " %s %.1f ms", L.stage_name(i), tot[i]); line += piece; }
        line += "\n";
        fputs(line.c_str(), stderr);
    }
    if (timing) {
        // per device: what its feeders moved (text in, records out: H2D + kernels + D2H inside tgsf_submit)
        for (size_t di = 0; di < o.devices.size(); di++) {
            if (std::find(o.devices.begin(), o.devices.begin() + (long)di, o.devices[di]) != o.devices.begin() + (long)di) continue;   // (listed twice)
            double sub = 0; uint64_t by = 0, nb2 = 0; int nf2 = 0, node = -1;
            for (size_t k = 0; k < ctx_dev.size(); k++)
                if (ctx_dev[k] == o.devices[di]) { sub += dev_submit_s[k]; by += dev_bytes[k]; nb2 += dev_batches[k]; nf2++; node = std::max(node, dev_node[k]); }
            fprintf(stderr, "DEVICE %d: %llu batches, %.2f GB of text through %d feeders, tgsf_submit %.3f s summed = %.1f GB/s per feeder, %.1f GB/s for the device over the pipeline's %.3f s%s\n",
                    o.devices[di], (unsigned long long)nb2, by * 1e-9, nf2, sub, sub > 0 ? by * 1e-9 / sub : 0.0, t_pipe > 0 ? by * 1e-9 / t_pipe : 0.0, t_pipe,
                    node >= 0 ? (" (feeders bound to NUMA node " + std::to_string(node) + ")").c_str() : "");
        }
    }
    if (timing) {
        // CPU seconds by stage (cputime.h): what the threads that have ended charged, the main thread's share up to here, and
        // what the process used beyond both (the runtime's own threads).  ONE write (ranks share this stderr).
        const double proc = process_cpu_s();
        double own = 0;
        std::string line;
        char piece[160];
        for (int s = 0; s < CPU_N; s++) {
            double v = (double)cpu_ns()[s].load() * 1e-9;
            if (s == CPU_MAIN) v += (double)thread_cpu_ns() * 1e-9 - (double)cpu_ns()[CPU_PREPASS].load() * 1e-9;      // (the main thread is still running)
            own += v;
            if (v >= 0.0005) { snprintf(piece, sizeof piece, "%s %s %.3f", line.empty() ? "" : ",", cpu_stage_name(s), v); line += piece; }
        }
        snprintf(piece, sizeof piece, "CPU: %.3f s of CPU time (user + system) for %.3f Gbases = %.4f CPU-s per Gbase |", proc, (double)own_raw_bases * 1e-9,
                 own_raw_bases ? proc / ((double)own_raw_bases * 1e-9) : 0.0);
        std::string head = piece;
        snprintf(piece, sizeof piece, " | threads of the runtime and others %.3f\n", proc - own);
        fputs((head + line + piece).c_str(), stderr);
    }
    if (timing && o.downsample)
        fprintf(stderr, "DOWN: selection %.3f | QC pass over the kept reads (%s, %.2f GB of them in %.2f GB of text): context %.3f, batches + submits %.3f (its own thread; %d submits %.3f, the first %.3f) | writing them %.3f (%s) | closing the output %.3f\n",
                t_dsel, d_in_place ? "read in place" : "packed", d_kept * 1e-9, d_span * 1e-9, t_dcreate, t_dqc, n_dsubmit, t_dsubmit, t_dfirst, t_dwrite, d_mapped ? "fallocate + threads into a mapping" : "writev", t_dclose);
    // everything is written and closed: skip the teardown of multi-GB mappings and of the HIP runtime
    if (timing) {
        struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
        fprintf(stderr, "CLOCK: main entered at %.6f, leaving at %.6f (epoch seconds)\n", t_epoch0, (double)ts.tv_sec + ts.tv_nsec * 1e-9);
    }
}

}  // namespace host
