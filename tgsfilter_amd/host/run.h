// run.h -- one run of the command line, stage by stage.  main() (main.cpp) parses the options and calls the stages in this
// order; what one stage leaves for the next lives in the Run object:
//   set_up_ranks        one process per GPU: --ranks forks, --shard meets the other ranks (shard.h)              run_setup.cpp
//   check_file_types    suffixes -> formats, report name (:2993-3033)
//   open_input          library load started, input mapped / opened for streaming, this rank's byte range, record index
//   open_output_early   the output file's pages start to exist beside the pre-pass
//   prepass             P1-P3 (:3058-3126), broadcast in a sharded job; the adapters to search for                run_contexts.cpp
//   make_contexts       batch sizes, output sink, library joined, RCCL communicator started, tgsf_params
//   filter_pass         indexer -> batcher -> feeders (tgsf_submit) -> ordered planner -> fill threads             run_pipeline.cpp, run_writer.cpp
//   downsample          DownSampleTask (:2164-2568): selection over all ranks' fragments, QC pass, writing        run_downsample.cpp, run_second_pass.cpp
//   sum_tallies         contexts -> ranks (RCCL all-reduce, checked against the sum over the sockets) -> rank 0   run_report.cpp
//   report              statistics, INFO lines, HTML report (:3146-3328); stale part files of earlier jobs
//   timing_lines        TGSF_TIMING's lines
// Replaces main (src/TGSFilter.cpp:2945-3332) and TGSFilterTask (:1755-2162).
#pragma once
#include <memory>

#include "api.h"
#include "pipeline.h"
#include "prepass.h"
#include "report.h"
#include "shard.h"

namespace host {

// CPUs' worth of time this process may use: the hardware's threads, or less where a control group caps it (main.cpp)
int cpu_budget();
// ends the process: flushes, lets the output sink take back what it reserved on a failure, tells a waiting parent (TGSF_DETACH)
[[noreturn]] void leave(int code);
// true when the work runs in a child whose teardown goes on after the caller has its status (TGSF_DETACH=1)
bool detached();

struct Run {
    explicit Run(Options& opts) : o(opts) {}

    // ---- the stages, in order ----
    int set_up_ranks();                 // non-zero: stop with that exit status
    int check_file_types();
    void open_input();
    void open_output_early();
    void prepass();
    void make_contexts();
    void filter_pass();
    void downsample();
    void sum_tallies();
    void report();
    void timing_lines();

    // ---- options, ranks ----
    Options& o;
    RankLink link;
    bool shard_may_use_rccl = false;
    bool sharded = false;               // --ranks / --shard (also with one rank: the same program path, one part file)
    bool timing = false;                // TGSF_TIMING: stage wall times on stderr (not part of the surface)
    double t_epoch0 = 0, t_start = 0;
    double t_drain = 0, t_prepass = 0, t_pipe = 0, t_parse = 0, t_gpu = 0, t_write = 0, t_widle = 0, t_first = 0;

    // ---- input ----
    std::string html;
    bool fasta_in = false;              // records without qualities: count-only tallies, no Q gate
    InputBytes in;
    bool streaming = false;
    size_t chunk_bytes = 64u << 20;
    int scan_threads = 1;
    size_t text_off = 0, text_size = 0; // this rank's part of the text
    const char* text = nullptr;
    std::unique_ptr<RecordIndex> records_p;
    std::unique_ptr<ChunkReader> open_stream();

    // ---- output ----
    std::string out_path;
    MappedSink sink;
    std::atomic<bool> early_stop{false};
    std::thread early;
    void end_early() { if (early.joinable()) { early_stop = true; early.join(); } }
    Output out;
    bool fastq_out = false, run_filter_pass = true;

    // ---- pre-pass ----
    PrepassResult pp;
    std::vector<std::string> adapters;

    // ---- contexts ----
    uint64_t batch_text = 0;
    uint32_t batch_reads = 1u << 16;
    const Api* api = nullptr;
    double t_load = 0, t_dev = 0, t_libwait = 0;
    const RcclApi* R = nullptr;
    struct RcclUp { std::atomic<int> done{0}; int rc = TGSF_OK; std::string err; void* comm = nullptr; };
    std::shared_ptr<RcclUp> rccl_state = std::make_shared<RcclUp>();
    std::thread rccl_up;
    bool use_rccl = false;
    bool ranks_own_gpus = false;        // a sharded job whose ranks sit on distinct devices (from the gather of their bus ids)
    tgsf_params p;
    std::vector<int> ctx_dev;
    std::vector<tgsf_ctx*> ctxs;
    double t_p0 = 0;

    // ---- filter pass ----
    std::unique_ptr<Channel<std::shared_ptr<Batch>>> to_gpu, to_writer;
    std::vector<int> raw_lens, clean_lens;
    uint64_t raw_bases = 0, clean_bases = 0;
    uint64_t own_raw_bases = 0;         // what THIS process read (a sharded job's rank 0 ends with the job's total in raw_bases)
    std::vector<CleanRec> clean_recs;   // only filled when downsampling follows
    BatchStore store;
    std::atomic<uint64_t> stream_text{0};   // streamed input: text handed out so far ...
    std::atomic<double> stream_share{0.0};  // ... out of this share of the file's bytes
    std::mutex gpu_time_m;
    bool numa_bind = false;
    std::vector<int> dev_node;
    std::vector<double> dev_submit_s;
    std::vector<uint64_t> dev_bytes, dev_batches;
    int fill_threads = 1, populate_threads = 1;
    uint64_t fill_min = 1u << 20, stride_bytes = 2ull << 30;
    std::unique_ptr<Pool> pool, populate;
    std::unique_ptr<Reserver> reserver;
    bool release_input = false, release_output = false;
    std::unique_ptr<Channel<std::pair<const char*, uint64_t>>> to_release;
    bool mapped_out = false;
    double t_f0 = 0, t_busy = 0, t_fill_tail = 0, t_close = 0;
    void reader_body();
    void feed(size_t k);
    void bind_to_node_of(size_t k);
    void writer_body();
    void fill_job(const std::shared_ptr<Batch>& b, size_t lo, size_t hi);
    void batch_done(std::shared_ptr<Batch> b);

    // ---- downsampling ----
    std::unique_ptr<MappedSink> dsink;
    std::unique_ptr<Pool> dpop;
    std::unique_ptr<Reserver> dres;
    bool open_dsink(uint64_t capacity, uint64_t speculative);
    uint64_t down_bases = 0, down_job_recs = 0, down_job_bases = 0;
    std::vector<int> down_lens;
    std::vector<uint64_t> down_t;
    double t_dsel = 0, t_dcreate = 0, t_dqc = 0, t_dwrite = 0, t_dclose = 0, t_dsubmit = 0, t_dfirst = 0, d_kept = 0, d_span = 0;
    int n_dsubmit = 0;
    bool d_in_place = false, d_mapped = false;
    std::vector<char> select_kept();    // the reference's selection (:2297-2344), over all ranks' fragments: keep flag per clean_recs entry
    void second_pass(const std::vector<char>& keep);

    // ---- tallies, report ----
    uint64_t nw = 0;
    int32_t bc = 0;
    uint32_t nbins = 0;
    std::vector<uint64_t> t;            // the job's tally vector (include/tgsf.h layout)
    double t_x0 = 0, t_rccl_wait = 0, t_allreduce = 0;
    int rccl_ranks = 0;
    std::string tally_route;            // how the tallies of a sharded job travelled (SHARD line)
    bool reports() const { return !sharded || link.rank == 0; }
};

// A rank's tally vector as it travels to rank 0 over the sockets: everything in front of the four per-100-bp tables, then of
// each of those only the rows in use; and its sum into rank 0's vector (the four "rows used" words are maxima).
std::vector<uint64_t> pack_rows(const std::vector<uint64_t>& v, int32_t bc, uint32_t nbins);
void add_rows(std::vector<uint64_t>& v, const std::vector<uint64_t>& ru, int32_t bc, uint32_t nbins);

}  // namespace host
