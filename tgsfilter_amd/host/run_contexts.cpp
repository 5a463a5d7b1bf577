// run_contexts.cpp -- the pre-pass and what the filter pass needs before it starts (run.h): the adapters to search for, batch sizes,
// the output sink, the library joined, the job's RCCL communicator started, tgsf_params.
#include "run.h"

namespace host {

// ---- pre-pass, :3058-3126 ----
void Run::prepass()
{
    std::unique_ptr<CpuScope> cpu_prepass(new CpuScope(CPU_PREPASS));
    if (sharded && link.rank > 0) {
        // (rank 0 looks at the first reads of the WHOLE input, as the reference does, and broadcasts what it found: below)
    } else if (streaming) {
        std::unique_ptr<ChunkReader> cr = open_stream();
        std::shared_ptr<Chunk> ch;
        size_t at = 0;
        bool over = false;
        pp = run_prepass(o, [&](Rec& r) {
            while (!over && (!ch || at >= ch->recs.size())) {
                if (ch && !ch->message.empty()) std::cerr << ch->message << std::endl;
                if (ch && ch->last) { over = true; break; }
                ch = cr->next(o.in_file);
                at = 0;
                if (!ch) over = true;
            }
            if (over) return false;
            r = ch->recs[at++];
            return true;
        });
    } else if (sharded && link.world > 1) {
        // the sample may reach beyond this rank's part: an index of its own over the whole text, kept a little ahead of
        // the pre-pass and dropped when that has seen enough
        // (every other rank waits for what this pass finds: it may use more than this rank's share of the CPUs for a moment)
        RecordIndex whole(in.data(), in.size(), !fasta_in, std::max(scan_threads, std::min(8, cpu_budget() / 2)), 1u << 14);
        RecordIndex::Cursor cur(whole);
        pp = run_prepass(o, [&](Rec& r) { return cur.next(r); });
    } else {
        RecordIndex::Cursor cur(*records_p);
        pp = run_prepass(o, [&](Rec& r) { return cur.next(r); });
    }
    if (sharded) {                                                     // the pre-pass's constants, from rank 0 to every rank
        BlobOut b;
        if (link.rank == 0) { b.pod(pp.qtype); b.pod(pp.trim5p); b.pod(pp.trim3p); b.pod(pp.depth5p); b.pod(pp.depth3p); b.pod(o.min_q); b.str(pp.adapter5p); b.str(pp.adapter3p); }
        link.bcast(b.s);
        BlobIn r(b.s);
        r.pod(pp.qtype); r.pod(pp.trim5p); r.pod(pp.trim3p); r.pod(pp.depth5p); r.pod(pp.depth3p); r.pod(o.min_q); r.str(pp.adapter5p); r.str(pp.adapter3p);
    }
    cpu_prepass.reset();
    t_prepass = now_s() - t_start;
    if (o.filter) {
        if (o.head_trim < 0) o.head_trim = pp.trim5p;
        if (o.tail_trim < 0) o.tail_trim = pp.trim3p;
        std::cerr << "INFO: trim 5' end length: " << o.head_trim << std::endl;
        std::cerr << "INFO: trim 3' end length: " << o.tail_trim << std::endl;
        std::cerr << "INFO: min output reads length: " << o.min_len << std::endl;
        if (!fasta_in) std::cerr << "INFO: min Phred average quality score: " << o.min_q << std::endl;     // :3074-3076
        auto add = [&](const std::string& a) { if (std::find(adapters.begin(), adapters.end(), a) == adapters.end()) adapters.push_back(a); };
        if (!o.adapter_file.empty()) {                                 // Get_adapters, :2923-2942
            InputBytes af;
            if (af.open(o.adapter_file)) {
                const int ft = file_type(o.adapter_file);
                FastxReader rd(af.data(), af.size(), ft == 1);
                Record r;
                while (rd.next(r)) { add(std::string(r.seq)); add(rev_comp(std::string(r.seq))); }
            }
            int num = 0;
            for (const std::string& a : adapters) std::cerr << "INFO: input adapter " << ++num << " :" << a << std::endl;
        } else {
            std::string a5 = pp.adapter5p, a3 = pp.adapter3p;
            float d5 = pp.depth5p, d3 = pp.depth3p;
            if (d5 > 5 * d3) { a3.clear(); d3 = 0; } else if (d3 > 5 * d5) { a5.clear(); d5 = 0; }   // :3086-3092
            std::cerr << "INFO: 5' adapter: " << a5 << std::endl;
            std::cerr << "INFO: 3' adapter: " << a3 << std::endl;
            std::cerr << "INFO: mean depth of 5' adapter: " << d5 << std::endl;
            std::cerr << "INFO: mean depth of 3' adapter: " << d3 << std::endl;
            if (o.only_adapters) leave(0);
            if (!a5.empty()) { add(a5); add(rev_comp(a5)); }
            if (!a3.empty()) { add(a3); add(rev_comp(a3)); }
            if (a5.empty() && a3.empty()) {                            // :3115-3125
                if (o.read_type == "hifi" || o.read_type == "clr") {
                    add(kAdapterLib[0]); add(kAdapterLib[1]);
                    std::cerr << "INFO: set PacBio blunt adapter to trim: " << kAdapterLib[0] << std::endl;
                } else if (o.read_type == "ont") {
                    add(kAdapterLib[8]); add(kAdapterLib[9]);
                    std::cerr << "INFO: set NanoPore rapid adapter to trim: " << kAdapterLib[8] << std::endl;
                }
            }
        }
    }
}

// ---- contexts ----
void Run::make_contexts()
{
    // batches are slices of the input text: sized in text bytes (about 2 bytes per base + headers)
    batch_text = streaming ? std::min<uint64_t>(256ull << 20, chunk_bytes)
                                    : std::min<uint64_t>(256ull << 20, std::max<uint64_t>(text_size / 8 + 4096, 1 << 16));
    if (const char* e = knob("TGSF_BATCH_BYTES")) { const long long v = atoll(e); if (v > 0) batch_text = (uint64_t)v; }   // tuning / test knob
    // reads per batch: the library keeps traceback scratch for every (read, adapter, end) of a batch -- columns x words of
    // the longest alignment each; with the library adapters that is ~12 KB per read, with 256-bp adapters and loose
    // match lengths ~350 KB: keep it under 4 GB per context
    batch_reads = 1u << 16;
    {
        // (the library's rule, tgsf_lib.hip: every alignment of a batch gets the columns of the longest one, and the
        // words of the widest column class any adapter needs -- 1 / 2 / 4 words up to 64 / 128 / 256 bp, 20 beyond)
        uint64_t per_read = 0, cols = 0, words = 1;
        for (const std::string& a : adapters) {
            const int Q = (int)a.size();
            const int kmax = std::max(0, std::min(Q - 1, std::max(Q - o.end_match_len + 1, Q - o.mid_match_len + 1)));
            cols = std::max<uint64_t>(cols, (uint64_t)(Q + kmax + 2));
            words = std::max<uint64_t>(words, Q > 256 ? (uint64_t)(Q + 63) / 64 : Q > 128 ? 4 : Q > 64 ? 2 : 1);
        }
        per_read = cols * 2 * words * 8;
        if (words > 4) per_read = std::min<uint64_t>(per_read, (1ull << 20) + 16 * words) + 32 * words + 512;   // (beyond 1 MiB an alignment is cut by Hirschberg's scheme, as in edlib)
        per_read *= 3 * std::max<size_t>(adapters.size(), 1);          // two end windows + one middle alignment per adapter
        if (per_read) batch_reads = (uint32_t)std::min<uint64_t>(batch_reads, std::max<uint64_t>(256, (4ull << 30) / per_read));
    }
    fastq_out = o.out_type == 1;
    run_filter_pass = o.filter || o.only_qc;                // :3061; with -F the input goes straight to downsampling
    {
        const char* w = getenv("TGSF_WRITER");                         // "writev": always the single-stream writer
        const bool may_map = !o.only_qc && !o.out_gz && !o.downsample && run_filter_pass && !o.out_file.empty() &&
                             !(w && !strcmp(w, "writev"));
        // address space for the output mapping: what the input could turn into (a streamed input's text size is unknown)
        // (address space only: pages exist where records are laid out.  A record's header is repeated in front of each of
        // its fragments, so an output can outgrow its input -- by a factor only headers of kilobytes reach.)
        if (may_map && !sink.is_open()) sink.open(out_path, streaming ? std::max<uint64_t>(64ull << 30, 64ull * in.size())
                                                                      : 4 * (uint64_t)text_size + (1ull << 30));
        Options oo = o;
        oo.out_file = out_path;
        if (!o.only_qc && !sink.is_open() && !out.open(oo)) leave(1);
    }
    api = &lib();
    const Api& L = *api;                                              // joins the loader thread
    lib_times(t_load, t_dev);
    t_libwait = now_s() - t_start - t_prepass;
    // How the tallies of a sharded job will be summed at the end: on the devices, one RCCL all-reduce, when every rank has
    // a GPU of its own -- the communicator is set up NOW, on a helper thread beside the filtering (RCCL's first
    // initialisation takes longer than a small run) -- or over the ranks' sockets when ranks share a GPU (RCCL refuses two
    // ranks on one device), where the collective library is missing, or with TGSF_SHARD_EXCHANGE=socket.
    if (sharded) {
        const char* ex = getenv("TGSF_SHARD_EXCHANGE");
        if (shard_may_use_rccl) R = rccl_lib();
        char bus[64] = {0};
        int node = -1;
        if (L.device_location(o.device, bus, (int)sizeof bus, &node) != TGSF_OK) snprintf(bus, sizeof bus, "device%d", o.device);
        BlobOut mine;
        mine.str(bus);
        mine.pod<int>(R ? 1 : 0);
        const std::vector<std::string> all = link.gather(mine.s);
        // verdict: [0] a GPU per rank (what the NUMA binding of the feeders goes by), [1] the RCCL exchange is on (+ the communicator's id)
        std::string verdict(2, '0');
        if (link.rank == 0) {
            std::vector<std::string> seen;
            bool ok = R != nullptr, own = true;
            for (const std::string& a : all) {
                BlobIn in2(a);
                std::string b2; int have = 0;
                in2.str(b2); in2.pod(have);
                own = own && std::find(seen.begin(), seen.end(), b2) == seen.end();
                ok = ok && have;
                seen.push_back(b2);
            }
            ok = ok && own;
            verdict[0] = own ? '1' : '0';
            if (ok) {
                char id[TGSF_RCCL_ID_BYTES];
                if (R->unique_id(id) == TGSF_OK) { verdict[1] = '1'; verdict.append(id, sizeof id); }
                else if (ex && !strcmp(ex, "rccl")) die(std::string("TGSF_SHARD_EXCHANGE=rccl: ") + R->last_error());
            } else if (ex && !strcmp(ex, "rccl")) die("TGSF_SHARD_EXCHANGE=rccl: ranks share a GPU, or libtgsf_rccl.so does not load on every rank");
        }
        link.bcast(verdict);
        ranks_own_gpus = verdict.size() >= 2 && verdict[0] == '1';
        use_rccl = verdict.size() == 2 + TGSF_RCCL_ID_BYTES && verdict[1] == '1';
        if (use_rccl)
            // (the helper owns its state: should the communicator never come up, the run goes on without it -- below -- and the
            // helper is left behind where it waits)
            rccl_up = std::thread([R = R, up = rccl_state, verdict, device = o.device, rank = link.rank, world = link.world] {
                if (const char* e = knob("TGSF_RCCL_STALL_S")) usleep((useconds_t)(atof(e) * 1e6));       // test knob: a communicator that is late
                up->rc = R->comm_init(device, verdict.data() + 2, rank, world, &up->comm);
                if (up->rc != TGSF_OK) up->err = R->last_error();       // (thread-local text: taken on this thread)
                up->done.store(1, std::memory_order_release);
            });
    }
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.min_len = o.min_len; p.max_len = o.max_len; p.min_q = o.min_q < 0 ? 0.f : o.min_q; p.max_q = o.max_q;
    p.bc_len = o.bc_len; p.head_trim = o.head_trim < 0 ? 0 : o.head_trim; p.tail_trim = o.tail_trim < 0 ? 0 : o.tail_trim;
    p.end_len = o.end_len; p.end_match_len = o.end_match_len; p.mid_match_len = o.mid_match_len; p.extra_len = o.extra_len;
    p.end_sim = o.end_sim; p.mid_sim = o.mid_sim; p.discard = o.discard; p.filter = o.filter; p.only_qc = o.only_qc;
    p.min_repeat = o.min_repeat; p.kmer = o.kmer; p.qtype = pp.qtype ? pp.qtype : 33;
    p.no_qual = fasta_in ? 1 : 0;
    if (adapters.size() > TGSF_MAX_ADAPTERS) die("more than " + std::to_string(TGSF_MAX_ADAPTERS) + " adapter sequences");
    p.n_adapters = (int)adapters.size();
    for (size_t a = 0; a < adapters.size(); a++) { p.adapters[a] = adapters[a].data(); p.adapter_len[a] = (int)adapters[a].size(); }
    p.max_batch_reads = batch_reads;
    // rows of the per-100-bp tables: from the longest read when the index is complete by now (it usually is: it runs
    // at tens of GB/s beside the device bring-up), else from what the file could hold
    if (streaming) p.max_read_len = 1u << 26;
    else if (sharded) {
        // every rank needs the same table rows (the all-reduce sums vectors of one layout): the longest read of the whole
        // job when the parts are indexed in a moment anyway, what the file could hold otherwise
        if (in.size() <= (1ull << 30)) { records_p->wait_complete(); p.max_read_len = (uint32_t)std::max<uint64_t>(link.max_u64(records_p->longest()), 1024); }
        else p.max_read_len = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(in.size() / 2, 1024), 1u << 26);
    }
    else if (records_p->complete()) p.max_read_len = std::max<uint32_t>(records_p->longest(), 1024);
    else p.max_read_len = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(in.size() / 2, 1024), 1u << 26);
    // capacity is in buffer bytes: a text slice must fit, and so must one record of the longest read on its own
    // (header + two lines of max_read_len)
    p.max_batch_bases = std::max<uint64_t>(batch_text, 2ull * p.max_read_len + (1u << 16)) + (1 << 20);
    // Per device of --devices: a few contexts, each with its own feeder thread.  A feeder's tgsf_submit is
    // synchronous (H2D from the pageable input mapping, kernels, D2H): one of them moves ~26 GB/s over the link, two
    // or three together saturate it (~55 GB/s) and keep the kernels of one batch under the copy of another.  Batches
    // are dealt to whichever feeder is free, the planner re-sequences them, the tallies are merged at the end
    // (host side).
    int per_dev = (streaming || text_size > (64u << 20)) ? 3 : 1;
    if (const char* e = getenv("TGSF_CTX_PER_DEVICE")) { const int v = atoi(e); if (v >= 1 && v <= 8) per_dev = v; }
    for (int d : o.devices) for (int k = 0; k < per_dev; k++) ctx_dev.push_back(d);
    ctxs.assign(ctx_dev.size(), nullptr);
    t_p0 = now_s();
}

}  // namespace host
