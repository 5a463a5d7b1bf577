// run_pipeline.cpp -- the filter pass of a run (run.h):
//   indexer -> batcher (reader_body) -> [queue] -> GPU feeders (feed: tgsf_submit) -> [queue] -> ordered planner (writer_body) -> fill threads.
// The reference moves one read at a time as three std::string copies through lock-free queues to N worker threads
// (TGSFilterTask, src/TGSFilter.cpp:1755-2162); here reads are only INDEXED on the host (the mmap'ed FASTQ text itself is the
// batch buffer: sequence and quality lines are read in place by the kernels), filtered on the GPU through the C ABI, and the
// kept fragments are formatted straight from the input text in input order (= the reference's -t 1 order).
#include "run.h"

#include <sched.h>

namespace host {

// read_fastx, :1845-1870 (batches of indexed records)
void Run::reader_body()
{
    CpuScope cpu(CPU_BATCHER);
    if (!run_filter_pass) { for (size_t d = 0; d < ctxs.size(); d++) to_gpu->put(nullptr); return; }
    std::unique_ptr<RecordIndex::Cursor> rd;
    if (!streaming) rd.reset(new RecordIndex::Cursor(*records_p));
    std::unique_ptr<ChunkReader> cr;
    if (streaming) cr = open_stream();
    std::shared_ptr<Chunk> ch;                                     // streamed input: the chunk being dealt into batches
    size_t ch_at = 0;
    bool over = false;
    Rec r;
    auto fresh = [&] { return store.get(); };
    std::shared_ptr<Batch> b = fresh();
    const double t0 = now_s();
    double waited = 0;
    uint64_t next_id = 0;
    auto flush = [&] {
        if (b->recs.empty()) return;
        const double w0 = now_s();
        b->id = next_id++;
        to_gpu->put(std::move(b));
        waited += now_s() - w0;
        b = fresh();
    };
    auto next_record = [&]() {
        if (!streaming) return rd->next(r);
        while (!over && (!ch || ch_at >= ch->recs.size())) {
            flush();                                               // a batch never spans two chunks
            if (ch && !ch->message.empty()) std::cerr << ch->message << std::endl;
            if (ch && ch->last) { over = true; break; }
            ch = cr->next(o.in_file);
            ch_at = 0;
            if (!ch) { over = true; break; }
            stream_text.store(cr->text_bytes());
            stream_share.store(cr->consumed());
        }
        if (over) return false;
        r = ch->recs[ch_at++];
        return true;
    };
    // (a sharded job: the cuts are checked against what the reader sees, see text_off above)
    auto bad_cut = [&](size_t at) {
        die("--ranks / --shard: the input cannot be cut near byte " + std::to_string(at) + " the way one sequential reader would read it "
            "(lines the reader skips, or a malformed record, near there): run it without --ranks / --shard");
    };
    const char* last_end = nullptr;
    while (next_record()) {
        const size_t L = r.len;
        if (L > p.max_read_len) die("read longer than the supported maximum");
        const char* rec_end = (fasta_in ? r.seq : r.qual) + L;
        if (sharded && !last_end && link.rank > 0 && r.name != text + 1) bad_cut(text_off);
        last_end = rec_end;
        if (!b->recs.empty() && ((uint64_t)(rec_end - b->base) > batch_text || b->recs.size() >= batch_reads)) flush();
        if (b->recs.empty()) { b->base = r.name; b->hold = ch; }
        b->off.push_back((uint64_t)(r.seq - b->base));
        b->qoff.push_back(fasta_in ? b->off.back() : (uint64_t)(r.qual - b->base));
        b->len.push_back((uint32_t)L); b->recs.push_back(r);
        b->span = (uint64_t)(rec_end - b->base);
        if (b->span > p.max_batch_bases) die("record larger than a batch");
        b->bases += L;
        raw_bases += L; raw_lens.push_back((int)L);
    }
    if (sharded && link.rank + 1 < link.world) {                   // the last record ends exactly where the next rank begins
        const char* e = last_end ? last_end : text;
        const char* const end = text + text_size;
        if (e < end && *e == '\r') e++;
        if (e < end && *e == '\n') e++;
        if (e != end || !records_p->end_message().empty()) bad_cut(text_off + text_size);
    }
    flush();
    for (size_t d = 0; d < ctxs.size(); d++) to_gpu->put(nullptr);       // one end marker per feeder
    t_parse = now_s() - t0 - waited;
    std::sort(raw_lens.begin(), raw_lens.end());                   // for the statistics (:3151), beside the rest of the pipeline
}

// Several GPUs: a device's feeders -- and the pinned staging buffers tgsf_create allocates from them -- stay on the
// CPUs of the GPU's own NUMA node, so that no feeder pushes its copies across the socket link.
void Run::bind_to_node_of(size_t k)
{
    const Api& L = *api;
    int node = -1;
    char bus[64];
    if (L.device_location(ctx_dev[k], bus, (int)sizeof bus, &node) == TGSF_OK && node >= 0) {
        const char* sysfs = knob("TGSF_SYSFS_NODES");                  // test knob: another directory in place of /sys/devices/system/node
        std::ifstream f(std::string(sysfs ? sysfs : "/sys/devices/system/node") + "/node" + std::to_string(node) + "/cpulist");
        std::string list;
        cpu_set_t set;
        CPU_ZERO(&set);
        int n_set = 0;
        if (f && std::getline(f, list)) {                      // "0-63,128-191"
            size_t i = 0;
            while (i < list.size()) {
                const int a = atoi(list.c_str() + i);
                int b = a;
                size_t j = list.find_first_of(",-", i);
                if (j != std::string::npos && list[j] == '-') { b = atoi(list.c_str() + j + 1); j = list.find(',', j); }
                for (int c2 = a; c2 <= b && c2 < CPU_SETSIZE; c2++) { CPU_SET(c2, &set); n_set++; }
                i = j == std::string::npos ? list.size() : j + 1;
            }
        }
        // within what the caller allows (taskset, numactl, a container's cpuset): never a wider mask than it came with
        cpu_set_t allowed;
        if (n_set > 0 && sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
            n_set = 0;
            for (int c2 = 0; c2 < CPU_SETSIZE; c2++) {
                if (CPU_ISSET(c2, &set) && !CPU_ISSET(c2, &allowed)) CPU_CLR(c2, &set);
                if (CPU_ISSET(c2, &set)) n_set++;
            }
        }
        if (n_set > 0 && sched_setaffinity(0, sizeof set, &set) == 0) dev_node[k] = node;
    }
}

// filter_sequence, :1919-2064, one batch per call
void Run::feed(size_t k)
{
    CpuScope cpu(CPU_FEEDER);
    const Api& L = *api;
    if (numa_bind) bind_to_node_of(k);
    if (L.create(&p, ctx_dev[k], &ctxs[k]) != TGSF_OK) die(L.last_error(nullptr));
    tgsf_ctx* fctx = ctxs[k];
    if (timing) (void)L.profile(fctx, 1);                          // HIP events around the stages of every batch (GPU: line)
    for (;;) {
        std::shared_ptr<Batch> b = to_gpu->get();
        if (!b) break;
        const double g0 = now_s();
        if (b->res.size() < b->recs.size()) b->res.resize(b->recs.size());
        const size_t fneed = (size_t)(b->bases / (uint64_t)std::max(p.min_len, 1)) + b->recs.size() + 16;
        if (b->frags.size() < fneed) b->frags.resize(fneed);
        const uint8_t* btext = reinterpret_cast<const uint8_t*>(b->base);
        tgsf_batch_in bi;
        memset(&bi, 0, sizeof bi);
        bi.seq = btext; bi.qual = btext;                           // one buffer: the FASTQ text itself
        bi.offsets = b->off.data(); bi.qual_offsets = b->qoff.data(); bi.lengths = b->len.data();
        bi.n_reads = (uint32_t)b->recs.size(); bi.n_bytes = b->span;
        tgsf_batch_out bo{b->res.data(), b->frags.data(), (uint32_t)b->frags.size(), 0};
        if (L.submit(fctx, &bi, &bo) != TGSF_OK) die(L.last_error(fctx));
        { std::lock_guard<std::mutex> l(gpu_time_m); t_gpu += now_s() - g0; if (t_first == 0) t_first = now_s() - t_p0; }
        dev_submit_s[k] += now_s() - g0; dev_bytes[k] += b->span; dev_batches[k]++;
        b->n_frags = bo.n_frags;
        to_writer->put(std::move(b));
    }
    to_writer->put(nullptr);
}

// a written batch: its mappings go to the releaser thread where they are dropped at all (from the 16 fill threads at once
// that cost 7-15 thread-seconds of a run and slowed everything beside them), the batch itself back to the store
void Run::batch_done(std::shared_ptr<Batch> b)
{
    if (b->out_bytes && release_output) to_release->put({b->dst, b->out_bytes});
    if (release_input) to_release->put({b->base, b->span});
    store.put(std::move(b));
}

void Run::filter_pass()
{
    // ---- pipeline ----
    to_gpu.reset(new Channel<std::shared_ptr<Batch>>(2 + ctxs.size()));
    to_writer.reset(new Channel<std::shared_ptr<Batch>>(256));
    if (!run_filter_pass) {                                            // get_fastx_SeqLen, :2256-2269
        RecordIndex::Cursor rd(*records_p);
        Rec r;
        while (rd.next(r)) {
            clean_recs.push_back({std::string_view(r.name, r.name_len), 1, r.seq, r.qual, r.len});
            clean_bases += r.len;
        }
    }

    std::thread reader([this] { reader_body(); });

    // Several GPUs: a device's feeders -- and the pinned staging buffers tgsf_create allocates from them -- stay
    // on the CPUs of the GPU's own NUMA node, so that no feeder pushes its copies across the socket link.  With one GPU
    // binding was measured within noise and is left off; TGSF_NUMA=1 / 0 forces it on / off.
    // (a job of rank processes on GPUs of their own is the same case, one device per process)
    numa_bind = o.devices.size() > 1 || (link.world > 1 && ranks_own_gpus);     // (decided from the ranks' bus ids, make_contexts: ranks that share a GPU are not bound)
    if (const char* e = getenv("TGSF_NUMA")) numa_bind = atoi(e) > 0;
    dev_node.assign(ctx_dev.size(), -1);
    dev_submit_s.assign(ctx_dev.size(), 0.0);
    dev_bytes.assign(ctx_dev.size(), 0);
    dev_batches.assign(ctx_dev.size(), 0);
    std::vector<std::thread> feeders;
    for (size_t k = 0; k < ctxs.size(); k++) feeders.emplace_back([this, k] { feed(k); });

    fill_threads = std::max(1, std::min(o.n_thread, 16));
    fill_min = 1u << 20;                                      // bytes worth a job of their own
    if (const char* e = knob("TGSF_FILL_MIN_BYTES")) { const long long v = atoll(e); if (v > 0) fill_min = (uint64_t)v; }   // test knob
    pool.reset(new Pool(sink.is_open() ? fill_threads : 1));
    populate_threads = std::max(1, std::min(o.n_thread, 32));      // short bursts between two fallocates: the more the shorter
    if (const char* e = knob("TGSF_POPULATE_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) populate_threads = v; }   // tuning knob (tests/manual/e2e_cpu.py)
    // (the pages of a reserved stride are mapped by many threads BETWEEN two fallocates: beside one, page faults on the file
    // take its inode's lock and both crawl -- measured, DESIGN appendix)
    populate.reset(new Pool(sink.is_open() ? populate_threads : 0, CPU_POPULATE));
    // (a streamed input is decoder-bound: small strides keep the mapped part of the output -- it counts as resident -- small)
    stride_bytes = streaming ? (128ull << 20) : (2ull << 30);
    if (const char* e = knob("TGSF_STRIDE_BYTES")) { const long long v = atoll(e); if (v > 0) stride_bytes = (uint64_t)v; }   // tuning / test knob
    // While the library loads and the device comes up pages of the output file are instantiated already, up to a quarter
    // of the input's size (what a run keeps is not known yet; a surplus is cut off at the end).
    end_early();                                                       // (what it reserved is mapped by the reserver's first round)
    reserver.reset(new Reserver(sink, *populate, stride_bytes, false));
    if (sink.is_open())
        reserver->start((!streaming && text_size > (256u << 20)) ? (uint64_t)text_size / 4 : 0);
    {
        uint64_t early_min = 1ull << 30;
        if (const char* e = knob("TGSF_DOWN_EARLY_MIN")) early_min = strtoull(e, nullptr, 10);          // tests: small inputs too
        if (o.downsample && !streaming && in.mapped() && (uint64_t)text_size >= early_min) {
            uint64_t spec = (uint64_t)text_size / 4;
            if (o.genome_size > 0 && o.desired_depth > 0) spec = std::min<uint64_t>(spec, (2 * o.genome_size * (uint64_t)o.desired_depth) / (uint64_t)link.world + (uint64_t)text_size / 64);
            open_dsink(4 * (uint64_t)text_size + (1ull << 30), spec);
        }
    }
    // Mappings of written batches (input text, output file).  One process (the default): the teardown is on the caller's
    // clock -- 90 ns per page of the input, ~200 ns per dirty page of the output if it all waited for the exit -- so the
    // mappings are dropped piece by piece during the run, by ONE background thread (several only get in each other's way).
    // Both mappings carry MADV_SEQUENTIAL: unmapping a page of a mapping without it marks the page accessed, and 20 M pages
    // moving between the LRU lists slow the fallocate beside them by a quarter.  TGSF_DETACH=1: nothing is
    // dropped during the run (the child's address space goes in the background), except where the resident size matters
    // (a streamed input: the mapped part of the output counts as resident).
    const bool sync_exit = !detached();                              // one process (the default): the teardown is on the clock
    release_input = sync_exit && !streaming && in.mapped() && !o.downsample;
    release_output = sync_exit || streaming;
    const uint64_t release_piece = 16u << 20;
    to_release.reset(new Channel<std::pair<const char*, uint64_t>>(1 << 16));
    std::thread releaser([this, release_piece] {
        CpuScope cpu(CPU_RELEASER);
        for (;;) {
            const std::pair<const char*, uint64_t> r = to_release->get();
            if (!r.first) break;
            for (uint64_t o2 = 0; o2 < r.second; o2 += release_piece) MappedSink::release(r.first + o2, std::min<uint64_t>(release_piece, r.second - o2));
        }
    });
    std::thread writer([this] { writer_body(); });
    reader.join();
    for (std::thread& f : feeders) f.join();
    writer.join();
    reserver->finish();
    mapped_out = sink.is_open();
    t_f0 = now_s();
    pool->finish();
    populate->finish();
    to_release->put({nullptr, 0});
    releaser.join();
    t_busy = pool->busy_s();
    t_fill_tail = now_s() - t_f0;
    sink.close();
    t_close = now_s() - t_f0 - t_fill_tail;
    t_pipe = now_s() - t_p0;
}

}  // namespace host
