// run_setup.cpp -- the stages of a run before the filter pass (run.h): ranks, file types, input, early output, pre-pass, contexts.
#include "run.h"

#include <sched.h>
#include <sys/stat.h>

namespace host {

// ---- one process per GPU (shard.h) ----
int Run::set_up_ranks()
{
    if (o.ranks >= 1 && o.shard_world > 0) { std::cerr << "Error: --ranks starts the ranks itself; --shard is for a rank started by another launcher" << std::endl; return 1; }
    if (o.ranks >= 1 || o.shard_world >= 1) {
        const int world = o.ranks >= 1 ? o.ranks : o.shard_world;
        const char* why = nullptr;
        if (o.out_file.empty() && !o.only_qc && !o.only_adapters) why = "every rank writes a part file of its own: -o is needed";
        else if (file_type(o.in_file) == 2 || (o.in_file.size() > 3 && o.in_file.compare(o.in_file.size() - 3, 3, ".gz") == 0))
            why = "the ranks take byte ranges of a plain FASTQ / FASTA text";
        if (why) { std::cerr << "Error: --ranks / --shard: " << why << std::endl; return 1; }
        if (world > 1024) { std::cerr << "Error: --ranks " << world << std::endl; return 1; }
        // a GPU per rank? (--ranks knows before it forks; ranks started by another launcher find out when they meet)
        {
            std::vector<int> dv;
            for (int r = 0; r < world && o.ranks >= 1; r++) dv.push_back(o.devices.empty() ? r : o.devices[(size_t)r % o.devices.size()]);
            std::sort(dv.begin(), dv.end());
            shard_may_use_rccl = o.ranks < 1 || std::adjacent_find(dv.begin(), dv.end()) == dv.end();
            const char* ex = getenv("TGSF_SHARD_EXCHANGE");
            if (ex && !strcmp(ex, "socket")) shard_may_use_rccl = false;
        }
        if (o.ranks >= 1) {
            fork_ranks(world, link);                                       // returns in the N children only
            // rank r on device r, or on the r-th entry of --devices (cyclically: several ranks may share a GPU)
            o.device = o.devices.empty() ? link.rank : o.devices[(size_t)link.rank % o.devices.size()];
        } else {
            if (o.rendezvous.empty()) { std::cerr << "Error: --shard needs --rendezvous <path>: where the ranks of the job meet (a unix socket)" << std::endl; return 1; }
            setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);                    // (as fork_ranks does for its children: shard.h)
            link.rendezvous(o.shard_rank, world, o.rendezvous);
        }
        o.devices.assign(1, o.device);
        // -t is the job's: every rank takes its share -- or, on a node whose CPUs outnumber it (the reference clamps -t to 32:
        // 4 threads a rank with 8 GPUs), its share of the CPUs this job may use, up to -t
        o.n_thread = std::max(2, std::min(o.n_thread, std::max(o.n_thread / world, cpu_budget() / world)));
    }
    sharded = o.ranks >= 1 || o.shard_world >= 1;
    return 0;
}

// file types and report name, :2993-3033
int Run::check_file_types()
{
    o.in_type = file_type(o.in_file);
    const std::string prefix = file_prefix(o.in_file);
    html = prefix + ".html";
    if (!o.out_file.empty() && !o.only_qc) {
        html = file_prefix(o.out_file) + ".html";
        o.out_type = file_type(o.out_file);
        if (file_extension(o.out_file) == "gz") o.out_gz = true;
    } else {
        o.out_type = o.fasta_out ? 0 : (o.in_type == 2 ? 1 : o.in_type);
    }
    if (o.in_type == 3 || o.out_type == 3) {
        std::cerr << "Error: The file name suffix should be '.[fastq|fq|fasta|fa][.gz] or .[sam|bam]'" << std::endl;
        if (o.in_type == 3) std::cerr << "Error: Please check your input file name: " << o.in_file << std::endl;
        else std::cerr << "Error: Please check your output file name: " << o.out_file << std::endl;
        return 1;
    }
    if (o.in_type == 0 && o.out_type == 1) { std::cerr << "Error: Fasta format input file can't output fastq format file" << std::endl; return 1; }
    // rows of SURVEY 8(f) that are not built yet fail loudly instead of silently doing something else
    fasta_in = o.in_type == 0;                              // records without qualities: count-only tallies, no Q gate
    return 0;
}

std::unique_ptr<ChunkReader> Run::open_stream()
{
    std::string err;
    std::unique_ptr<TextSource> src = open_text(in.data(), in.size(), o.in_type == 2, err);
    if (!src) { std::cerr << "Error: " << err << " (" << o.in_file << ")" << std::endl; fflush(nullptr); _exit(255); }
    return std::unique_ptr<ChunkReader>(new ChunkReader(std::move(src), !fasta_in, chunk_bytes, 20));   // <= 20 x 64 MB of text alive
}

void Run::open_input()
{
    // loading the HIP library, device bring-up and kernel loading run beside the input open, the indexing and
    // the pre-pass (api.h)
    if (o.devices.empty()) o.devices.push_back(o.device);
    lib_start(o.devices);
    if (sharded && shard_may_use_rccl) rccl_start();                   // (librccl is large: loaded beside the pre-pass, only where it can be used)

    // Compressed / BAM / SAM input beyond a size is STREAMED: decoded piece by piece in bounded memory, once for the
    // pre-pass (which stops after its sample of reads) and once for the filter pass, as the reference reads it twice
    // (:949-1040, :1845-1917).  Smaller ones are decoded whole (below), plain files are mapped.  A downsampling run keeps
    // its kept fragments addressed in the input text, so it takes the whole-file way.
    const bool coded = o.in_type == 2 || (o.in_file.size() > 3 && o.in_file.compare(o.in_file.size() - 3, 3, ".gz") == 0);
    uint64_t stream_min = 256ull << 20;
    if (const char* e = knob("TGSF_STREAM_MIN_BYTES")) stream_min = strtoull(e, nullptr, 10);      // test knob
    if (coded && !o.downsample) {
        if (!in.open_raw(o.in_file)) leave(1);
        streaming = in.size() >= stream_min;
    }
    if (!streaming && !in.open(o.in_file, o.in_type == 2)) leave(1);   // SAM/BAM: decoded to FASTQ text (read_bam, :1872-1917)
    chunk_bytes = [] { const char* e = knob("TGSF_CHUNK_BYTES"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)(64u << 20); }();
    // mapped / decoded input: the records are indexed once, in the background, for the pre-pass and for the filter pass
    const int budget = std::max(1, cpu_budget() / link.world);         // (a sharded job: every rank takes its share)
    // indexing runs ahead of everything else and is memory-bound from a few threads on: half the CPU budget at most
    scan_threads = std::max(1, std::min({o.n_thread, 32, std::max(2, budget / 2)}));
    // This rank's part of the text: all of it, or -- one process per GPU, shard.h -- the rank-th of `world` byte ranges,
    // cut where find_record_start proposes.  The proposal is checked against the record reader's own view from both
    // sides (the reader thread below): this rank's first record must begin exactly at the cut, and its last record must
    // end exactly at the next rank's cut.  By induction from rank 0, which starts at byte 0, every rank then reads its
    // records exactly as the reference's one sequential reader does (FastxReader, src/TGSFilter.cpp:521-782: its only
    // state between two records is the position in the text); a text that cannot be cut that way -- lines that make the
    // reader skip, a malformed record -- ends the run with a message instead of being read differently.
    text_off = 0; text_size = in.size();
    if (sharded) {
        if (streaming || (in.size() > 0 && !in.mapped())) die("--ranks / --shard: the input is not a plain text file");
        const size_t lo = find_record_start(in.data(), in.size(), (size_t)((unsigned __int128)in.size() * (unsigned)link.rank / (unsigned)link.world), !fasta_in);
        const size_t hi = link.rank + 1 == link.world ? in.size()
                        : find_record_start(in.data(), in.size(), (size_t)((unsigned __int128)in.size() * (unsigned)(link.rank + 1) / (unsigned)link.world), !fasta_in);
        text_off = lo; text_size = hi > lo ? hi - lo : 0;
    }
    text = in.data() + text_off;
    if (!streaming) records_p.reset(new RecordIndex(text, text_size, !fasta_in, scan_threads));
}

void Run::open_output_early()
{
    // The output file's pages are the critical path of a run that writes a tmpfs file: their instantiation
    // starts NOW, beside the pre-pass and the device bring-up -- if the file does not exist yet (an existing one is not
    // touched before the run is certain to write it: a run that ends in its pre-pass leaves it as it was, as the reference
    // does; a file created here is removed again on such a path).
    // (a rank of a sharded job writes its own part: the parts, concatenated in rank order, are the single process's file)
    out_path = sharded && !o.out_file.empty() ? o.out_file + ".part" + std::to_string(link.rank) : o.out_file;
    {
        const char* w = getenv("TGSF_WRITER");                         // "writev": always the single-stream writer
        uint64_t early_min = 256ull << 20;
        if (const char* e = knob("TGSF_EARLY_OPEN_MIN")) early_min = strtoull(e, nullptr, 10);           // tests: small inputs too
        const bool may_map_early = !o.only_qc && !o.out_gz && !o.downsample && (o.filter || o.only_qc) && !o.out_file.empty() &&
                                   !(w && !strcmp(w, "writev")) && !o.only_adapters && !streaming && in.mapped() &&
                                   (uint64_t)text_size >= early_min;
        if (may_map_early && sink.open(out_path, 4 * (uint64_t)text_size + (1ull << 30), true))
            early = std::thread([this] {
                CpuScope cpu(CPU_FALLOCATE);
                const uint64_t limit = (uint64_t)text_size / 4;        // what a run keeps is not known yet; a surplus is cut off at the end
                while (!early_stop.load() && sink.reserved() < limit)
                    if (!sink.reserve_to(std::min<uint64_t>(limit, sink.reserved() + (256u << 20)), false)) break;   // (a nearly full file system: not this thread's call)
            });
    }
}

}  // namespace host
