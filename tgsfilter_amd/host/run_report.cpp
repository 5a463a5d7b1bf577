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
    // One process per GPU: the job's tallies = the sum over the ranks (src/TGSFilter.cpp:3208-3213 across GPUs).  With a
    // GPU per rank: this rank's contexts folded into one vector in HBM, then ONE all-reduce of it over RCCL / xGMI
    // (include/tgsf_rccl.h; every rank has entered tgsf_create with the same table rows, see max_read_len above).
    t_x0 = now_s();
    std::vector<tgsf_ctx*> sum_ctxs = ctxs;
    if (sharded && use_rccl) {
        // The communicator has had the whole run to come up.  One that is still not there some time after the filtering is
        // over (a peer that cannot be reached, a fabric that does not answer) must not hold the job for ever: this rank says
        // so below, every rank then sums over the sockets, and the helper is left where it waits (the process leaves with _exit).
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
        // the communicator came up on every rank, or nobody uses it: the sockets carry the rows in use instead (the run's
        // results do not depend on which way the tallies travel)
        if (link.max_u64(rccl_rc != TGSF_OK ? 1 : 0) != 0) {
            const char* ex = getenv("TGSF_SHARD_EXCHANGE");
            if (ex && !strcmp(ex, "rccl")) die("RCCL communicator: " + (rccl_err.empty() ? std::string("it failed on another rank") : rccl_err));
            if (rccl_rc != TGSF_OK) std::cerr << "Warning: rank " << link.rank << ": RCCL communicator: " << rccl_err << " -- the tallies are summed over the ranks' sockets" << std::endl;
            // (a communicator that did come up here is left as it is: taking it down may wait for peers that are stuck)
            rccl_comm = nullptr;
            use_rccl = false;
        }
    }
    if (sharded && use_rccl) {
        for (size_t k = 1; k < ctxs.size(); k++)
            if (L.counters_merge(ctxs[0], ctxs[k]) != TGSF_OK) die(L.last_error(ctxs[0]));
        const double a0 = now_s();
        if (R->allreduce_counters(ctxs[0], rccl_comm, link.rank, link.world, 0, nullptr) != TGSF_OK) die(std::string("tally all-reduce: ") + R->last_error());
        t_allreduce = now_s() - a0;
        (void)R->comm_count(rccl_comm, &rccl_ranks);
        sum_ctxs.assign(1, ctxs[0]);                                   // (it holds the whole job's totals now, on every rank)
    }
    for (tgsf_ctx* c : sum_ctxs) {                                     // sums; the four "rows used" words are maxima
        uint64_t used[2] = {0, 0};                                     // of the bin tables only the rows in use travel
        if (L.counters_used(c, part.data(), nw, used) != TGSF_OK) die(L.last_error(c));
        uint64_t rows[4];
        for (int k = 0; k < 4; k++) rows[k] = std::max(t[TGSF_CTR_ROWS + k], part[TGSF_CTR_ROWS + k]);
        const size_t head = tgsf_ctr_bin_table(0, bc, nbins);
        for (size_t i = 0; i < head; i++) t[i] += part[i];
        for (int b = 0; b < 4; b++) {
            const size_t at = tgsf_ctr_bin_table(b, bc, nbins), n = (size_t)used[b >> 1] * 5;
            for (size_t i = 0; i < n; i++) t[at + i] += part[at + i];
        }
        for (int k = 0; k < 4; k++) t[TGSF_CTR_ROWS + k] = rows[k];
    }
    // Rank 0 of a sharded job receives every rank's read lengths (the statistics need them sorted: N50 and the like) and
    // -- when the tallies were not summed on the devices -- its tally rows in use; it alone prints the run's statistics
    // and writes the report.
    if (sharded) {
        BlobOut mine;
        mine.pod(raw_bases); mine.pod(clean_bases);
        mine.vec(raw_lens); mine.vec(clean_lens);                      // (each sorted already, beside the pipeline)
        std::vector<uint64_t> rows_used;
        if (!use_rccl) rows_used = pack_rows(t, bc, nbins);
        mine.vec(rows_used);
        const std::vector<std::string> all = link.gather(mine.s);
        for (int k = 1; k < (int)all.size(); k++) {                    // (rank 0 only)
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
            if (!use_rccl) add_rows(t, ru, bc, nbins);
        }
        if (timing)
            fprintf(stderr, "SHARD %d/%d: bytes [%zu, %zu) of the text on device %d -> %s | tallies: %s (communicator ready after %.3f s of waiting, all-reduce %.4f s, %d ranks in it), exchange + gather %.3f s\n",
                    link.rank, link.world, text_off, text_off + text_size, o.device, out_path.c_str(),
                    use_rccl ? "RCCL all-reduce on the devices" : "summed on rank 0 over the ranks' sockets", t_rccl_wait, t_allreduce, rccl_ranks, now_s() - t_x0);
        if (use_rccl) R->comm_destroy(rccl_comm);
    }
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
    // (parts of an earlier job with MORE ranks beside this job's would end up in a `cat <out>.part*`: they go, with a word)
    if (reports() && sharded && !o.out_file.empty() && !o.only_qc && !o.only_adapters)
        for (int k = link.world;; k++) {
            const std::string stale = o.out_file + ".part" + std::to_string(k);
            struct stat st;
            if (lstat(stale.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) break;
            if (unlink(stale.c_str()) == 0) std::cerr << "Warning: " << stale << ", a part of an earlier job with more ranks, was removed." << std::endl;
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
        snprintf(piece, sizeof piece, "CPU: %.3f s of CPU time (user + system) for %.3f Gbases = %.4f CPU-s per Gbase |", proc, (double)raw_bases * 1e-9,
                 raw_bases ? proc / ((double)raw_bases * 1e-9) : 0.0);
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
