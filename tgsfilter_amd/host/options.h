// options.h -- the tgsfilter command-line surface (kept as the reference codes it, not as it documents it).
// Reference: Para_A24 src/TGSFilter.cpp:82-172, TGSFilter_usage :33-77, TGSFilter_cmd :198-503.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace host {

struct Options {
    std::string in_file, out_file;
    int min_len = 1000;               // -l (clamped to >= 100, :232-234)
    int max_len = 2147483647;         // -L
    float min_q = -1.f;               // -q (resolved by the pre-pass when < 0)
    float max_q = 255.f;              // -Q
    int bc_num = 100000;              // -n
    int bc_len = 150;                 // -e
    float end_bias = 1.f;             // -b
    int head_trim = -1, tail_trim = -1;   // -5 / -3 (auto when < 0)
    std::string adapter_file;         // -a
    bool only_adapters = false;       // -A
    int ad_num = 100000;              // -N
    int end_len = 150;                // -E
    int end_match_len = 4;            // -m (the usage text says 15; the code default is 4, :148)
    int mid_match_len = 35;           // -M
    int extra_len = 50;               // -T
    float end_sim = 0.f, mid_sim = 0.f;   // -s / -S (0: by read type)
    bool discard = false;             // -D
    uint64_t genome_size = 0;         // -g
    int desired_depth = 0;            // -d
    int desired_num = 0;              // -r
    float desired_frac = 0.f;         // -R
    bool downsample = false;
    bool filter = true;               // cleared by -F / --qc
    int kmer = 11;                    // -k
    int min_repeat = 0;               // -p
    bool only_qc = false;             // --qc
    bool fasta_out = false;           // -f
    std::string read_type;            // -x
    int n_thread = 16;                // -t
    int comp_level = 6;               // -c
    int in_type = 3, out_type = 3;    // 0 fasta, 1 fastq, 2 bam, 3 unknown (:839-857)
    bool out_gz = false;
    // this build only
    int device = 0;                   // --device <n>
    std::vector<int> devices;         // --devices a,b,...: one context + feeder thread per entry (default: {device})
    // one process per GPU over one input (shard.h): --ranks N forks N rank processes (rank r on device r, or on the r-th
    // entry of --devices, taken cyclically); --shard r/N is one rank of a job started by another launcher
    // ("env": RANK / WORLD_SIZE / LOCAL_RANK as torchrun sets them), which finds the others at --rendezvous <path>
    int ranks = 0;
    int shard_rank = -1, shard_world = 0;
    std::string rendezvous;
};

int print_usage();
// Returns 0 to continue, 1 to stop with exit status 1 (same contract as TGSFilter_cmd);
// calls exit(-1) where the reference does.
int parse_args(int argc, char** argv, Options& o);
// The rank `--shard r/N` / `--shard env` names on this command line, or -1 (no such option, or one parse_args will refuse): main
// needs it BEFORE parse_args prints its own INFO lines, which a rank above 0 leaves to rank 0.  Walks the tokens as parse_args
// does, so that a VALUE that happens to read "shard" (`-o shard`) is not taken for the option.
int shard_rank_on_command_line(int argc, char** argv);

std::string file_extension(const std::string& path);      // :814-822
std::string file_prefix(const std::string& path);         // :824-837
int file_type(const std::string& path);                   // :839-857

}  // namespace host
