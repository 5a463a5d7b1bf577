// options.cpp -- argv -> Options.  Behaviour follows TGSFilter_cmd (src/TGSFilter.cpp:198-503):
// every token in flag position must start with '-', ALL '-' are stripped from it (so --qc == -qc),
// values are taken with atoi/atof, messages and exit codes are the reference's.
#include "options.h"

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <map>
#include <thread>

namespace host {

int print_usage()
{
    // the usage text is part of the CLI surface (src/TGSFilter.cpp:33-77)
    static const char* const lines[] = {
        "Usage: tgsfilter -i TGS.raw.fq.gz -x ont -o TGS.clean.fq.gz",
        " Input/Output options:",
        "   -i   <str>   input of bam/fasta/fastq file",
        "   -x   <str>   read type (ont|clr|hifi)",
        "   -o   <str>   output of fasta/fastq file instead of stdout",
        " Basic filter options:",
        "   -l   <int>   min length of read to out [1000]",
        "   -L   <int>   max length of read to out",
        "   -q  <float>  min Phred average quality score",
        "   -Q  <float>  max Phred average quality score",
        "   -n   <int>   read number for base content check [100000]",
        "   -e   <int>   read end length for base content check [150]",
        "   -b  <float>  bias (%) of adjacent base content at read end [1]",
        "   -5   <int>   trim bases from the 5' end of the read",
        "   -3   <int>   trim bases from the 3' end of the read",
        " Adapter filter options:",
        "   -a   <str>   adapter sequence file ",
        "   -A           disable reads filter, only for adapter identify",
        "   -N   <int>   read number for adapter identify [100000]",
        "   -E   <int>   read end length for adapter trim [150]",
        "   -m   <int>   min match length for end adapter [15]",
        "   -M   <int>   min match length for middle adapter [35]",
        "   -T   <int>   extra trim length for middle adpter on both side [50]",
        "   -s  <float>  min similarity for end adapter",
        "   -S  <float>  min similarity for middle adapter",
        "   -D           discard reads with middle adapter instead of split",
        " Downsampling options:",
        "   -g   <str>   genome size (k/m/g)",
        "   -d   <int>   downsample to the desired coverage (requires -g) ",
        "   -r   <int>   downsample to the desired number of reads ",
        "   -R  <float>  downsample to the desired fraction of reads ",
        "   -k   <int>   kmer size for repeat evaluations [11] ",
        "   -p   <int>   min repeat length of reads [0] ",
        "   -F           disable reads filter, only for downsampling",
        " Other options:",
        "   --qc         disable all filter, only for quality control ",
        "   -f           force FASTA output (discard quality) ",
        "   -c   <int>   compression level (0-9) for compressed output [6]",
        "   -t   <int>   number of threads [16]",
        "   -h           show help [v1.11]",
        "",
    };
    for (const char* l : lines) std::cout << l << "\n";
    return 1;
}

static uint64_t genome_size(const std::string& g)          // GetGenomeSize, :178-196
{
    const double number = std::stod(g.substr(0, g.size() - 1));
    switch (g.back()) {
    case 'k': case 'K': return static_cast<uint64_t>(number * 1000.0);
    case 'm': case 'M': return static_cast<uint64_t>(number * 1000000.0);
    case 'g': case 'G': return static_cast<uint64_t>(number * 1000000000.0);
    default:
        std::cerr << "Error: Genome size should end with k/m/g or K/M/G" << std::endl;
        return 0;
    }
}

std::string file_extension(const std::string& path)
{
    const size_t dot = path.rfind('.');
    return dot == std::string::npos ? std::string() : path.substr(dot + 1);
}

std::string file_prefix(const std::string& path)
{
    std::string ext = file_extension(path), prefix = path;
    if (ext == "gz") { prefix = path.substr(0, path.rfind('.')); ext = file_extension(prefix); }
    if (ext == "fq" || ext == "fastq" || ext == "fa" || ext == "fasta" || ext == "bam" || ext == "sam" ||
        ext == "BAM" || ext == "SAM")
        prefix = prefix.substr(0, prefix.rfind('.'));
    return prefix;
}

int file_type(const std::string& path)
{
    std::string ext = file_extension(path);
    if (ext == "gz") ext = file_extension(path.substr(0, path.rfind('.')));
    if (ext == "fa" || ext == "fasta") return 0;
    if (ext == "fq" || ext == "fastq") return 1;
    if (ext == "sam" || ext == "SAM" || ext == "bam" || ext == "BAM") return 2;
    return 3;
}

// where `--shard env` puts the job's rendezvous socket when --rendezvous does not say: never a predictable name in a directory
// everybody can write to
static std::string private_socket_dir()
{
    const char* x = getenv("XDG_RUNTIME_DIR");
    struct stat st;
    if (x && *x && stat(x, &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & 077) == 0) return x;
    const std::string d = "/tmp/tgsfilter-" + std::to_string((unsigned)geteuid());
    if (mkdir(d.c_str(), 0700) != 0 && errno != EEXIST) { std::cerr << "Error: --shard env: cannot create " << d << ": " << strerror(errno) << " (give --rendezvous)" << std::endl; exit(-1); }
    if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & 077) != 0) {
        std::cerr << "Error: --shard env: " << d << " is not a directory of this user's alone (give --rendezvous)" << std::endl;
        exit(-1);
    }
    return d;
}

// the flags that take a value (the keys of parse_args's table: checked there)
static const char* const kWithValue[] = {"i", "o", "x", "l", "L", "q", "Q", "n", "e", "b", "5", "3", "a", "N", "E", "m", "M", "T", "s", "S", "g", "d",
                                         "r", "R", "k", "p", "c", "t", "device", "ranks", "rendezvous", "shard", "devices"};

int shard_rank_on_command_line(int argc, char** argv)
{
    // the walk of parse_args below: a token in flag position names its flag with every '-' taken out; a flag with a value
    // takes the next token whatever that looks like (`-o shard` is an output file)
    for (int i = 1; i < argc; i++) {
        if (argv[i][0] != '-') return -1;                    // (parse_args stops with an error there)
        std::string flag = argv[i];
        flag.erase(std::remove(flag.begin(), flag.end(), '-'), flag.end());
        const bool takes = std::find(std::begin(kWithValue), std::end(kWithValue), flag) != std::end(kWithValue);
        if (!takes) continue;
        if (i + 1 == argc) return -1;
        const char* v = argv[++i];
        if (flag != "shard") continue;
        if (!strcmp(v, "env")) { const char* r = getenv("RANK"); return r ? atoi(r) : -1; }
        return strchr(v, '/') ? atoi(v) : -1;                // "r/N", as parse_args reads it
    }
    return -1;
}

int parse_args(int argc, char** argv, Options& o)
{
    if (argc <= 2) { print_usage(); return 1; }
    using Setter = std::function<int(const char*)>;          // returns 1 to stop
    std::map<std::string, Setter> with_value = {
        {"i", [&](const char* v) { o.in_file = v; return 0; }},
        {"o", [&](const char* v) { o.out_file = v; return 0; }},
        {"x", [&](const char* v) { o.read_type = v; return 0; }},
        {"l", [&](const char* v) { o.min_len = std::max(atoi(v), 100); return 0; }},
        {"L", [&](const char* v) { o.max_len = atoi(v); return 0; }},
        {"q", [&](const char* v) { o.min_q = (float)atof(v); return 0; }},
        {"Q", [&](const char* v) { o.max_q = (float)atof(v); return 0; }},
        {"n", [&](const char* v) { o.bc_num = atoi(v); return 0; }},
        {"e", [&](const char* v) { o.bc_len = atoi(v); return 0; }},
        {"b", [&](const char* v) { o.end_bias = (float)atof(v); return 0; }},
        {"5", [&](const char* v) { o.head_trim = atoi(v); return 0; }},
        {"3", [&](const char* v) { o.tail_trim = atoi(v); return 0; }},
        {"a", [&](const char* v) { o.adapter_file = v; return 0; }},
        {"N", [&](const char* v) { o.ad_num = atoi(v); return 0; }},
        {"E", [&](const char* v) { o.end_len = atoi(v); return 0; }},
        {"m", [&](const char* v) { o.end_match_len = atoi(v); return 0; }},
        {"M", [&](const char* v) { o.mid_match_len = atoi(v); return 0; }},
        {"T", [&](const char* v) { o.extra_len = atoi(v); return 0; }},
        {"s", [&](const char* v) {
             o.end_sim = (float)atof(v);
             if (o.end_sim < 0.7) { o.end_sim = 0.7f; std::cerr << "Warning: re set -s to : " << o.end_sim << std::endl; }
             return 0; }},
        {"S", [&](const char* v) {
             o.mid_sim = (float)atof(v);
             if (o.mid_sim < 0.8) { o.mid_sim = 0.8f; std::cerr << "Warning: reset -S to : " << o.mid_sim << std::endl; }
             return 0; }},
        {"g", [&](const char* v) { o.genome_size = genome_size(v); return o.genome_size == 0 ? 1 : 0; }},
        {"d", [&](const char* v) { o.desired_depth = atoi(v); return 0; }},
        {"r", [&](const char* v) { o.desired_num = atoi(v); return 0; }},
        {"R", [&](const char* v) { o.desired_frac = (float)atof(v); return 0; }},
        {"k", [&](const char* v) { o.kmer = atoi(v); return 0; }},
        {"p", [&](const char* v) { o.min_repeat = atoi(v); return 0; }},
        {"c", [&](const char* v) { o.comp_level = atoi(v); return 0; }},
        {"t", [&](const char* v) { o.n_thread = atoi(v); return 0; }},
        {"device", [&](const char* v) { o.device = atoi(v); return 0; }},
        {"ranks", [&](const char* v) { o.ranks = atoi(v); return 0; }},
        {"rendezvous", [&](const char* v) { o.rendezvous = v; return 0; }},
        {"shard", [&](const char* v) {                        // "r/N", or "env"
             if (!strcmp(v, "env")) {
                 const char* r = getenv("RANK"); const char* w = getenv("WORLD_SIZE"); const char* l = getenv("LOCAL_RANK");
                 if (!r || !w) { std::cerr << "Error: --shard env needs RANK and WORLD_SIZE" << std::endl; return 1; }
                 o.shard_rank = atoi(r); o.shard_world = atoi(w);
                 if (l) o.device = atoi(l);
                 // (a launcher that hands out RANK / WORLD_SIZE also names the job by its rendezvous port: the ranks of one node
                 // meet at a socket named after it unless --rendezvous says where)
                 // -- in the user's own runtime directory ($XDG_RUNTIME_DIR), else in a directory only this user can enter
                 if (o.rendezvous.empty()) { const char* port = getenv("MASTER_PORT"); if (port && *port) o.rendezvous = private_socket_dir() + "/tgsfilter." + port + ".sock"; }
             } else {
                 const char* slash = strchr(v, '/');
                 if (!slash) { std::cerr << "Error: --shard takes <rank>/<ranks>" << std::endl; return 1; }
                 o.shard_rank = atoi(v); o.shard_world = atoi(slash + 1);
             }
             if (o.shard_world < 1 || o.shard_rank < 0 || o.shard_rank >= o.shard_world) { std::cerr << "Error: --shard " << v << ": no such rank" << std::endl; return 1; }
             return 0; }},
        {"devices", [&](const char* v) {                      // not in the reference: batches are dealt to several GPUs
             o.devices.clear();
             for (const char* c = v; *c;) {
                 o.devices.push_back(atoi(c));
                 while (*c && *c != ',') c++;
                 if (*c == ',') c++;
             }
             return 0; }},
    };
    std::map<std::string, std::function<void()>> switches = {
        {"A", [&] { o.only_adapters = true; }},
        {"D", [&] { o.discard = true; }},
        {"F", [&] { o.filter = false; }},
        {"qc", [&] { o.only_qc = true; }},
        {"f", [&] { o.fasta_out = true; }},
    };
    for (const char* k : kWithValue) if (!with_value.count(k)) { std::cerr << "Error: option table out of step: " << k << std::endl; return 1; }
    if (with_value.size() != sizeof kWithValue / sizeof kWithValue[0]) { std::cerr << "Error: option table out of step" << std::endl; return 1; }
    for (int i = 1; i < argc; i++) {
        if (argv[i][0] != '-') {
            std::cerr << "Error: command option error! please check." << std::endl;
            return 1;
        }
        std::string flag = argv[i];
        flag.erase(std::remove(flag.begin(), flag.end(), '-'), flag.end());
        auto wv = with_value.find(flag);
        if (wv != with_value.end()) {
            if (i + 1 == argc) { std::cerr << "Error: Lack Argument for [ -" << flag << " ]" << std::endl; return 1; }
            if (wv->second(argv[++i])) return 1;
            continue;
        }
        auto sw = switches.find(flag);
        if (sw != switches.end()) { sw->second(); continue; }
        if (flag == "help" || flag == "h") { print_usage(); return 1; }
        std::cerr << "Error: UnKnow argument -" << flag << std::endl;
        return 1;
    }

    if (o.in_file.empty()) { std::cerr << "Error: lack argument for the must: -i " << std::endl; exit(-1); }
    if (access(o.in_file.c_str(), 0) != 0) { std::cerr << "Error: Can't find this file for -i " << o.in_file << std::endl; exit(-1); }
    if (o.only_qc) o.filter = false;

    if (o.filter) {
        if (o.read_type.empty()) { std::cerr << "Error: lack argument for the must: -x " << std::endl; exit(-1); }
        const std::string rt = o.read_type;
        if (rt == "CLR" || rt == "clr") { std::cerr << "INFO: read type: PacBio continuous long read (clr)." << std::endl; o.read_type = "clr"; }
        else if (rt == "HIFI" || rt == "hifi" || rt == "CCS" || rt == "ccs") {
            std::cerr << "INFO: read type: PacBio highly accurate long reads (hifi)." << std::endl; o.read_type = "hifi";
        } else if (rt == "ONT" || rt == "ont") { std::cerr << "INFO: read type: NanoPore reads (ont)." << std::endl; o.read_type = "ont"; }
        else { std::cerr << "Error: read type should be : clr/hifi/ccs/ont or CLR/HIFI/CCS/ONT." << std::endl; exit(-1); }
        // read-type defaults, :439-457
        if (o.mid_sim == 0) o.mid_sim = o.read_type == "hifi" ? 0.95f : 0.9f;
        if (o.end_sim == 0) o.end_sim = o.read_type == "hifi" ? 0.9f : (o.read_type == "clr" ? 0.8f : 0.75f);
        std::cerr << "INFO: min similarity for middle adapter: " << o.mid_sim << std::endl;
        std::cerr << "INFO: min similarity for end adapter: " << o.end_sim << std::endl;
    }

    if (o.desired_num > 0 || o.desired_frac > 0) o.downsample = true;
    else if (o.genome_size > 0 || o.desired_depth > 0) {
        if (o.genome_size > 0 && o.desired_depth > 0) o.downsample = true;
        else if (o.genome_size > 0) { std::cerr << "Error: The desired depth was required, along with the genome size!" << std::endl; exit(-1); }
        else { std::cerr << "Error: The genome size was required, along with the desired depth!" << std::endl; exit(-1); }
    }
    if (!o.filter && !o.downsample && !o.only_qc) {
        std::cerr << "Error: Please set functional parameters for filter, downsampling or quality control." << std::endl;
        exit(-1);
    }
    // -t clamp, :488-499 (the value only sizes host helper threads here; the filtering runs on the GPU)
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw > 0 && o.n_thread > (int)hw - 1) {
        o.n_thread = (int)hw - 1;
        if (o.n_thread <= 32) std::cerr << "Warning: reset -t to: " << o.n_thread << std::endl;
    }
    if (o.n_thread > 32) { o.n_thread = 32; std::cerr << "Warning: reset -t to: " << o.n_thread << std::endl; }
    return 0;
}

}  // namespace host
