// main.cpp -- `tgsfilter` for MI355X: the reference's command line, stderr lines and report around
// the batch pipeline  reader -> [batch queue] -> GPU feeder (tgsf_submit) -> [batch queue] -> writer.
//
// Replaces main (src/TGSFilter.cpp:2945-3332) and TGSFilterTask (:1755-2162): the reference moves one
// read at a time as three std::string copies through lock-free queues to N worker threads; here reads
// are only INDEXED on the host (the mmap'ed FASTQ text itself is the batch buffer: sequence and quality
// lines are read in place by the kernels), filtered on the GPU through the C ABI, and the kept
// fragments are formatted straight from the input text in input order (= the reference's -t 1 order).
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>

#include "fastx.h"
#include "options.h"
#include "prepass.h"
#include "report.h"
#include "tgsf.h"

using namespace host;

namespace {

// A batch is a slice [base, base+span) of the input text plus an index of its records: nothing is
// copied on the host; the library reads sequence and quality lines in place (tgsf_batch_in.qual_offsets).
struct Batch {
    const char* base = nullptr;              // first byte of the slice (inside the mmap'ed input)
    uint64_t span = 0;                       // bytes of the slice
    std::vector<uint64_t> off, qoff;         // sequence / quality line of each read, relative to base
    std::vector<uint32_t> len;
    std::vector<std::string_view> names;     // views into the input
    std::vector<tgsf_read_result> res;
    std::vector<tgsf_fragment> frags;
    uint32_t n_frags = 0;
    uint64_t bases = 0;
    uint64_t id = 0;                         // position in the input: the writer puts batches back in order
};

// a kept fragment, addressed in the input text (used when a downsampling pass follows the filter pass)
struct CleanRec {
    std::string_view name;
    int pass_num;
    const char* seq;
    const char* qual;
    uint32_t len;
};

template <class T>
class Channel {                               // small bounded queue; nullptr closes it
public:
    explicit Channel(size_t cap) : cap_(cap) {}
    void put(T v) {
        std::unique_lock<std::mutex> l(m_);
        not_full_.wait(l, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(v));
        not_empty_.notify_one();
    }
    T get() {
        std::unique_lock<std::mutex> l(m_);
        not_empty_.wait(l, [&] { return !q_.empty(); });
        T v = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return v;
    }
private:
    std::mutex m_;
    std::condition_variable not_full_, not_empty_;
    std::deque<T> q_;
    size_t cap_;
};

// newSeqName, src/TGSFilter.cpp:1680-1701: ":<n>" goes before the first whitespace of the header
void append_name(std::string& out, std::string_view raw, int number)
{
    if (number < 2) { out.append(raw); return; }
    const std::string add = ":" + std::to_string(number);
    size_t i = 0;
    while (i < raw.size() && !std::isspace((unsigned char)raw[i])) i++;
    out.append(raw.substr(0, i));
    out += add;
    out.append(raw.substr(i));
}

class Output {                                // plain or per-record gzip members (:2020-2053, :786-812)
public:
    bool open(const Options& o) {
        gz_ = o.out_gz;
        if (o.out_file.empty()) f_ = stdout;
        else f_ = fopen(o.out_file.c_str(), "wb");
        if (!f_) { std::cerr << "Error: Failed to open file: " << o.out_file << std::endl; return false; }
        if (gz_) {
            setvbuf(f_, nullptr, _IOFBF, 8 << 20);
            level_ = o.comp_level;
            gz_threads_ = std::max(1, std::min(o.n_thread, 32));
        } else {
            fflush(f_);
            fd_ = fileno(f_);
        }
        return true;
    }
    // Plain output: the pieces of a record (header, sequence, separator, qualities) are gathered with
    // writev straight from the input text -- no intermediate record string.
    void piece(const char* p, size_t n) {
        if (!n) return;
        if (gz_) { gzbuf_.append(p, n); return; }
        iov_.push_back({const_cast<char*>(p), n});
        if (iov_.size() >= 1000) flush_iov();
    }
    // small generated text (":<n>" suffixes, separators that are not in the input)
    void text(const std::string& t) {
        if (gz_) { gzbuf_ += t; return; }
        if (pool_.size() + t.size() > pool_.capacity()) flush_iov();
        const size_t o0 = pool_.size();
        pool_ += t;
        iov_.push_back({&pool_[o0], t.size()});
    }
    void end_record() {
        if (!gz_) return;
        ends_.push_back(gzbuf_.size());
        if (gzbuf_.size() >= (256u << 20)) flush_gz();
    }
    // One gzip member per record, as the reference writes them (:786-812, compressed by its worker threads):
    // the records gathered since the last flush are split into byte-balanced runs, each run is compressed
    // member by member on its own thread, and the runs are written in order.
    void flush_gz() {
        if (ends_.empty()) return;
        const size_t nrec = ends_.size();
        const int T = (int)std::min<size_t>((size_t)gz_threads_, nrec);
        std::vector<std::vector<char>> outv((size_t)T);
        std::vector<size_t> cut((size_t)T + 1, nrec);
        cut[0] = 0;
        for (int t = 1; t < T; t++) {
            const size_t target = gzbuf_.size() / (size_t)T * (size_t)t;
            cut[(size_t)t] = (size_t)(std::lower_bound(ends_.begin(), ends_.end(), target) - ends_.begin());
        }
        std::atomic<bool> bad{false};
        auto work = [&](int t) {
            z_stream z;
            memset(&z, 0, sizeof z);
            if (deflateInit2(&z, level_, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = true; return; }
            std::vector<char>& o = outv[(size_t)t];
            for (size_t r = cut[(size_t)t]; r < cut[(size_t)t + 1]; r++) {
                const size_t b = r ? ends_[r - 1] : 0, n = ends_[r] - b;
                deflateReset(&z);
                const size_t at = o.size(), room = deflateBound(&z, (uLong)n) + 64;
                o.resize(at + room);
                z.next_in = (Bytef*)(gzbuf_.data() + b); z.avail_in = (uInt)n;
                z.next_out = (Bytef*)(o.data() + at); z.avail_out = (uInt)room;
                if (deflate(&z, Z_FINISH) != Z_STREAM_END) bad = true;
                o.resize(at + room - z.avail_out);
            }
            deflateEnd(&z);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (std::thread& x : th) x.join();
        if (bad) { std::cerr << "Error: compression failed" << std::endl; exit(-1); }
        for (const auto& o : outv) if (!o.empty() && fwrite(o.data(), 1, o.size(), f_) != o.size()) { std::cerr << "Error: write failed" << std::endl; exit(-1); }
        gzbuf_.clear();
        ends_.clear();
    }
    void flush_iov() {
        if (gz_) { flush_gz(); return; }
        size_t i = 0;
        while (i < iov_.size()) {
            ssize_t w = writev(fd_, &iov_[i], (int)std::min<size_t>(iov_.size() - i, 1000));
            if (w < 0) { std::cerr << "Error: write failed" << std::endl; exit(-1); }
            size_t left = (size_t)w;
            while (i < iov_.size() && left >= iov_[i].iov_len) { left -= iov_[i].iov_len; i++; }
            if (left) { iov_[i].iov_base = (char*)iov_[i].iov_base + left; iov_[i].iov_len -= left; }
        }
        iov_.clear();
        pool_.clear();
    }
    void close() {
        flush_iov();
        if (f_ && f_ != stdout) fclose(f_); else if (f_) fflush(f_);
        f_ = nullptr;
    }
    Output() { pool_.reserve(1 << 16); }
private:
    FILE* f_ = nullptr;
    int fd_ = -1;
    bool gz_ = false;
    int level_ = 6, gz_threads_ = 1;
    std::string gzbuf_, pool_;              // gz: the records since the last flush, back to back
    std::vector<size_t> ends_;              // ... and where each of them ends
    std::vector<iovec> iov_;
};

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

[[noreturn]] void die(const std::string& msg)
{
    std::cerr << "Error: " << msg << std::endl;
    exit(-1);
}

}  // namespace

int main(int argc, char** argv)
{
    Options o;
    if (parse_args(argc, argv, o)) return 1;
    const bool timing = getenv("TGSF_TIMING") != nullptr;      // stage wall times on stderr (not part of the surface)
    const double t_start = now_s();
    double t_prepass = 0, t_create = 0, t_pipe = 0, t_parse = 0, t_gpu = 0, t_write = 0, t_widle = 0, t_first = 0;

    // file types and report name, :2993-3033
    o.in_type = file_type(o.in_file);
    const std::string prefix = file_prefix(o.in_file);
    std::string html = prefix + ".html";
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
    const bool fasta_in = o.in_type == 0;                              // records without qualities: count-only tallies, no Q gate

    // HIP start-up and kernel loading run beside the input open and the pre-pass
    if (o.devices.empty()) o.devices.push_back(o.device);
    std::thread warm([&] { for (int d : o.devices) (void)tgsf_prepare_device(d); });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } warm_join{warm};

    InputBytes in;
    if (!in.open(o.in_file, o.in_type == 2)) return 1;            // SAM/BAM: decoded to FASTQ text (read_bam, :1872-1917)

    // ---- pre-pass, :3058-3126 ----
    PrepassResult pp = run_prepass(o, in);
    t_prepass = now_s() - t_start;
    std::vector<std::string> adapters;
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
                const int t = file_type(o.adapter_file);
                FastxReader rd(af.data(), af.size(), t == 1);
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
            if (o.only_adapters) return 0;
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

    // ---- context ----
    // batches are slices of the input text: sized in text bytes (about 2 bytes per base + headers)
    const uint64_t batch_text = std::min<uint64_t>(512ull << 20, std::max<uint64_t>(in.size() / 4 + 4096, 1 << 16));
    const uint32_t batch_reads = 1u << 16;
    tgsf_params p;
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
    p.max_read_len = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(in.size() / 2, 1024), 1u << 26);
    // capacity is in buffer bytes: a text slice must fit, and so must one record of the longest read on its own
    // (header + two lines of max_read_len)
    p.max_batch_bases = std::max<uint64_t>(batch_text, 2ull * p.max_read_len + (1u << 16)) + (1 << 20);
    // one context (and one feeder thread) per device of --devices; batches are dealt to whichever feeder is
    // free, the writer re-sequences them, the tallies are merged at the end (host side)
    if (warm.joinable()) warm.join();
    std::vector<tgsf_ctx*> ctxs(o.devices.size(), nullptr);
    const double t_c0 = now_s();
    for (size_t d = 0; d < ctxs.size(); d++)
        if (tgsf_create(&p, o.devices[d], &ctxs[d]) != TGSF_OK) die(tgsf_last_error(nullptr));
    tgsf_ctx* ctx = ctxs[0];
    t_create = now_s() - t_c0;
    const double t_p0 = now_s();

    // ---- pipeline ----
    Channel<std::unique_ptr<Batch>> to_gpu(2 + o.devices.size()), to_writer(2 + o.devices.size());
    std::vector<int> raw_lens, clean_lens;
    uint64_t raw_bases = 0, clean_bases = 0;
    const bool fastq_out = o.out_type == 1;
    Output out;
    if (!o.only_qc && !out.open(o)) return 1;
    std::vector<CleanRec> clean_recs;                                  // only filled when downsampling follows
    const bool run_filter_pass = o.filter || o.only_qc;                // :3061; with -F the input goes straight to downsampling
    if (!run_filter_pass) {                                            // get_fastx_SeqLen, :2256-2269
        FastxReader rd(in.data(), in.size(), !fasta_in);
        Record r;
        while (rd.next(r)) {
            clean_recs.push_back({r.name, 1, r.seq.data(), r.qual.data(), (uint32_t)r.seq.size()});
            clean_bases += r.seq.size();
        }
    }

    const int scan_threads = std::max(1, std::min(o.n_thread, 8));
    std::thread reader([&] {                                           // read_fastx, :1845-1870 (index only)
        if (!run_filter_pass) { for (size_t d = 0; d < ctxs.size(); d++) to_gpu.put(nullptr); return; }
        FastxReader rd(in.data(), in.size(), !fasta_in, scan_threads);
        Record r;
        auto fresh = [&] {
            std::unique_ptr<Batch> nb(new Batch);
            nb->off.reserve(batch_reads); nb->qoff.reserve(batch_reads); nb->len.reserve(batch_reads); nb->names.reserve(batch_reads);
            return nb;
        };
        std::unique_ptr<Batch> b = fresh();
        const double t0 = now_s();
        double waited = 0;
        uint64_t next_id = 0;
        auto flush = [&] {
            if (b->names.empty()) return;
            const double w0 = now_s();
            b->id = next_id++;
            to_gpu.put(std::move(b));
            waited += now_s() - w0;
            b = fresh();
        };
        while (rd.next(r)) {
            const size_t L = r.seq.size();
            if (L > p.max_read_len) die("read longer than the supported maximum");
            const char* rec_end = (fasta_in ? r.seq.data() : r.qual.data()) + L;
            if (!b->names.empty() && ((uint64_t)(rec_end - b->base) > batch_text || b->names.size() >= batch_reads)) flush();
            if (b->names.empty()) b->base = r.name.data();
            b->off.push_back((uint64_t)(r.seq.data() - b->base));
            b->qoff.push_back(fasta_in ? b->off.back() : (uint64_t)(r.qual.data() - b->base));
            b->len.push_back((uint32_t)L); b->names.push_back(r.name);
            b->span = (uint64_t)(rec_end - b->base);
            if (b->span > p.max_batch_bases) die("record larger than a batch");
            b->bases += L;
            raw_bases += L; raw_lens.push_back((int)L);
        }
        flush();
        for (size_t d = 0; d < ctxs.size(); d++) to_gpu.put(nullptr);       // one end marker per feeder
        t_parse = now_s() - t0 - waited;
    });

    std::mutex gpu_time_m;
    auto feed = [&](tgsf_ctx* fctx) {                                  // filter_sequence, :1919-2064, one batch per call
        for (;;) {
            std::unique_ptr<Batch> b = to_gpu.get();
            if (!b) break;
            const double g0 = now_s();
            b->res.resize(b->names.size());
            b->frags.resize((size_t)(b->bases / (uint64_t)std::max(p.min_len, 1)) + b->names.size() + 16);
            const uint8_t* text = reinterpret_cast<const uint8_t*>(b->base);
            tgsf_batch_in bi;
            memset(&bi, 0, sizeof bi);
            bi.seq = text; bi.qual = text;                             // one buffer: the FASTQ text itself
            bi.offsets = b->off.data(); bi.qual_offsets = b->qoff.data(); bi.lengths = b->len.data();
            bi.n_reads = (uint32_t)b->names.size(); bi.n_bytes = b->span;
            tgsf_batch_out bo{b->res.data(), b->frags.data(), (uint32_t)b->frags.size(), 0};
            if (tgsf_submit(fctx, &bi, &bo) != TGSF_OK) die(tgsf_last_error(fctx));
            { std::lock_guard<std::mutex> l(gpu_time_m); t_gpu += now_s() - g0; if (t_first == 0) t_first = now_s() - t_p0; }
            b->n_frags = bo.n_frags;
            to_writer.put(std::move(b));
        }
        to_writer.put(nullptr);
    };
    std::vector<std::thread> feeders;
    for (tgsf_ctx* c : ctxs) feeders.emplace_back(feed, c);

    std::thread writer([&] {                                           // record formatting :2011-2053 + write_output :2095-2145
        const std::string lead(1, fastq_out ? '@' : '>'), nl("\n"), sep("\n+\n");
        std::string name;
        std::map<uint64_t, std::unique_ptr<Batch>> held;               // batches that arrived ahead of their turn
        uint64_t want = 0;
        size_t open_feeders = ctxs.size();
        for (;;) {
            std::unique_ptr<Batch> b;
            auto it = held.find(want);
            if (it != held.end()) { b = std::move(it->second); held.erase(it); }
            else {
                if (open_feeders == 0) break;
                const double i0 = now_s();
                b = to_writer.get();
                t_widle += now_s() - i0;
                if (!b) { open_feeders--; continue; }
                if (b->id != want) { const uint64_t id = b->id; held[id] = std::move(b); continue; }
            }
            want++;
            const double w0 = now_s();
            for (size_t r = 0; r < b->names.size(); r++) {
                int pass_num = 1;
                const tgsf_read_result& rr = b->res[r];
                for (uint32_t f = rr.frag_begin; f < rr.frag_begin + rr.n_frags; f++) {
                    const tgsf_fragment& fr = b->frags[f];
                    if (!(fr.flags & TGSF_FF_PASS)) continue;
                    if (o.downsample) {                                // kept in memory instead of a tmp file (:3129-3137)
                        clean_recs.push_back({b->names[r], pass_num++, b->base + b->off[r] + fr.start,
                                              b->base + b->qoff[r] + fr.start, (uint32_t)fr.len});
                        clean_bases += (uint64_t)fr.len;
                        clean_lens.push_back(fr.len);
                        continue;
                    }
                    out.text(lead);
                    if (pass_num < 2) out.piece(b->names[r].data(), b->names[r].size());
                    else { name.clear(); append_name(name, b->names[r], pass_num); out.text(name); }
                    pass_num++;
                    out.text(nl);
                    out.piece(b->base + b->off[r] + fr.start, (size_t)fr.len);
                    if (fastq_out) {
                        out.text(sep);
                        out.piece(b->base + b->qoff[r] + fr.start, (size_t)fr.len);
                    }
                    out.text(nl);
                    out.end_record();
                    clean_bases += (uint64_t)fr.len;
                    clean_lens.push_back(fr.len);
                }
            }
            if (!o.only_qc) out.flush_iov();                             // the batch (and its views) goes away
            t_write += now_s() - w0;
        }
    });
    reader.join();
    for (std::thread& f : feeders) f.join();
    writer.join();
    t_pipe = now_s() - t_p0;

    // ---- downsampling: DownSampleTask, :2164-2568 ----
    // keep the longest reads until the target is met (:2297-2344), then a QC-only pass over the kept reads
    // (CalcAvgQuality / Get_5p/3p_base_qual again, :2436-2447) which also writes them, in input order.
    uint64_t down_bases = 0;
    std::vector<int> down_lens;
    std::vector<uint64_t> down_t;
    if (o.downsample) {
        // Selection as the reference makes it (:2297-2344), container for container, so that ties at the cut fall
        // the same way when both programs are built with the same standard library: lengths keyed by record name
        // in an unordered_map filled in write order (:2105, :2267; a repeated name keeps its last length), handed
        // over by copy (:3142, :2169), listed in the map's iteration order, std::sort by length (descending),
        // names taken from the top; the second pass keeps every record whose name was taken (:2356).
        auto full_name = [&](const CleanRec& c) {
            std::string nm;
            append_name(nm, c.name, c.pass_num);
            return nm;
        };
        std::unordered_map<std::string, int> seq_lens;
        uint64_t total = 0;
        for (const CleanRec& c : clean_recs) { seq_lens[full_name(c)] = (int)c.len; total += c.len; }
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
            chosen.insert(pr.first);
            added += (uint64_t)pr.second; added_num++;
            down_bases += (uint64_t)pr.second; down_lens.push_back(pr.second);
            if (by_size ? added >= desired : added_num >= want_num) break;
        }
        std::vector<char> keep(clean_recs.size(), 0);
        for (size_t i = 0; i < clean_recs.size(); i++) keep[i] = chosen.count(full_name(clean_recs[i])) ? 1 : 0;
        tgsf_params qp = p;
        qp.filter = 0; qp.only_qc = 1; qp.n_adapters = 0; qp.min_repeat = 0;
        // The reference's second pass re-reads what the filter pass wrote (:3129-3137): after a FASTA output
        // (-f, or FASTA input) the records carry no qualities, so this pass takes the count-only tallies.
        const bool down_no_qual = fasta_in || (run_filter_pass && !fastq_out);
        qp.no_qual = down_no_qual ? 1 : 0;
        qp.max_batch_bases = (1ull << 30); qp.max_batch_reads = 1u << 16;
        tgsf_ctx* qctx = nullptr;
        if (tgsf_create(&qp, o.devices[0], &qctx) != TGSF_OK) die(tgsf_last_error(nullptr));
        std::vector<uint8_t> bs, bq; std::vector<uint64_t> boff; std::vector<uint32_t> blen;
        std::vector<tgsf_read_result> bres; std::vector<tgsf_fragment> bfr(16);
        auto run = [&] {
            if (blen.empty()) return;
            bres.resize(blen.size());
            bs.resize(bs.size() + 64); bq.resize(bq.size() + 64);
            tgsf_batch_in bi; memset(&bi, 0, sizeof bi);
            bi.seq = bs.data(); bi.qual = bq.data(); bi.offsets = boff.data(); bi.lengths = blen.data();
            bi.n_reads = (uint32_t)blen.size(); bi.n_bytes = bs.size() - 64;
            tgsf_batch_out bo{bres.data(), bfr.data(), (uint32_t)bfr.size(), 0};
            if (tgsf_submit(qctx, &bi, &bo) != TGSF_OK) die(tgsf_last_error(qctx));
            bs.clear(); bq.clear(); boff.clear(); blen.clear();
        };
        const std::string lead(1, fastq_out ? '@' : '>'), nl("\n"), sep("\n+\n");
        std::string name;
        for (size_t i = 0; i < clean_recs.size(); i++) {
            if (!keep[i]) continue;
            const CleanRec& c = clean_recs[i];
            if (bs.size() + c.len > qp.max_batch_bases - (1u << 20) || blen.size() >= qp.max_batch_reads) run();
            const size_t o0 = (bs.size() + 15) & ~size_t(15);
            bs.resize(o0); bq.resize(o0);
            bs.insert(bs.end(), c.seq, c.seq + c.len);
            if (!down_no_qual) bq.insert(bq.end(), c.qual, c.qual + c.len); else bq.resize(bs.size());
            boff.push_back(o0); blen.push_back(c.len);
            out.text(lead);
            if (c.pass_num < 2) out.piece(c.name.data(), c.name.size());
            else { name.clear(); append_name(name, c.name, c.pass_num); out.text(name); }
            out.text(nl);
            out.piece(c.seq, c.len);
            if (fastq_out) { out.text(sep); out.piece(c.qual, c.len); }
            out.text(nl);
            out.end_record();
        }
        run();
        uint64_t qnw = 0; int32_t qbc = 0; uint32_t qnb = 0;
        tgsf_counters_len(qctx, &qnw, &qbc, &qnb);
        down_t.resize(qnw);
        if (tgsf_counters(qctx, down_t.data(), qnw) != TGSF_OK) die(tgsf_last_error(qctx));
        tgsf_destroy(qctx);
    }
    if (!o.only_qc) out.close();

    // ---- statistics, stderr, report: :3146-3235, :3240-3279, :3285-3328 ----
    uint64_t nw = 0; int32_t bc = 0; uint32_t nbins = 0;
    tgsf_counters_len(ctx, &nw, &bc, &nbins);
    std::vector<uint64_t> t(nw, 0), part(nw);
    for (tgsf_ctx* c : ctxs) {                                         // sums; the four "rows used" words are maxima
        if (tgsf_counters(c, part.data(), nw) != TGSF_OK) die(tgsf_last_error(c));
        uint64_t rows[4];
        for (int k = 0; k < 4; k++) rows[k] = std::max(t[TGSF_CTR_ROWS + k], part[TGSF_CTR_ROWS + k]);
        for (uint64_t i = 0; i < nw; i++) t[i] += part[i];
        for (int k = 0; k < 4; k++) t[TGSF_CTR_ROWS + k] = rows[k];
        tgsf_destroy(c);
    }
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
    if (run_filter_pass) {
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
            if (!o.downsample && !o.out_file.empty()) std::cerr << "INFO: Filtered reads were written to: " << o.out_file << "." << std::endl;
        }
    }
    if (o.downsample) {                                                // :3240-3279
        if (down_lens.empty()) die("no reads to downsample");
        std::sort(down_lens.begin(), down_lens.end());
        side_stats(bc, down_lens, down_bases, tables(down_t, false), clean);
        clean.tab[8] = limit_decimals(std::round(clean.mean_qual * 1000) / 1000.0, 2);      // two places here, :3268
        if (!o.filter)
            std::cerr << "INFO: " << clean_recs.size() << " reads with a total of " << clean_bases << " bases were input." << std::endl;
        std::cerr << "INFO: " << down_lens.size() << " reads with a total of " << down_bases << " bases after downsampling." << std::endl;
        if (!o.out_file.empty()) std::cerr << "INFO: Downsampled reads were written to: " << o.out_file << "." << std::endl;
    }
    std::string qc = fasta_in ? "0" : "1";                             // :3286-3291
    qc += o.only_qc ? "0" : ((!o.filter && o.downsample) ? "1" : "2"); // :3293-3299
    std::ofstream ofs(html);
    write_report(ofs, qc, raw, clean);
    ofs.close();
    std::cerr << "INFO: Quality control report was written to: " << html << "." << std::endl;
    if (timing)
        fprintf(stderr, "TIMING: total %.3f s | prepass %.3f | tgsf_create %.3f | pipeline %.3f (parse+pack %.3f, tgsf_submit %.3f, "
                        "format+write %.3f, writer waiting %.3f, first batch filtered after %.3f; stages overlap) | stats+report %.3f\n",
                now_s() - t_start, t_prepass, t_create, t_pipe, t_parse, t_gpu, t_write, t_widle, t_first, now_s() - t_p0 - t_pipe);
    // everything is written and closed: skip the teardown of multi-GB mappings and of the HIP runtime
    fflush(nullptr);
    _exit(0);
}
