// run_writer.cpp -- the output side of the filter pass (run.h): the ordered planner and the fill jobs; the early sink of a
// downsampling run.
#include "run.h"

namespace host {

// records [lo, hi) of the batch's layout copied into the mapped output file
void Run::fill_job(const std::shared_ptr<Batch>& b, size_t lo, size_t hi)
{
    using Emit = Batch::Emit;
    Batch& bb = *b;
    std::string nm;
    for (size_t i = lo; i < hi; i++) {
        const Emit& e = bb.em[i];
        const Rec& rec = bb.recs[e.read];
        const tgsf_fragment& fr = bb.frags[e.frag];
        char* d = bb.dst + e.at;
        *d++ = fastq_out ? '@' : '>';
        if (e.pass_num < 2) { memcpy(d, rec.name, rec.name_len); d += rec.name_len; }
        else { nm.clear(); append_name(nm, std::string_view(rec.name, rec.name_len), e.pass_num); memcpy(d, nm.data(), nm.size()); d += nm.size(); }
        *d++ = '\n';
        stream_copy(d, rec.seq + fr.start, (size_t)fr.len); d += fr.len;
        if (fastq_out) {
            memcpy(d, "\n+\n", 3); d += 3;
            stream_copy(d, rec.qual + fr.start, (size_t)fr.len); d += fr.len;
        }
        *d++ = '\n';
    }
    stream_fence();
    if (--bb.left == 0) batch_done(b);
}

// record formatting :2011-2053 + write_output :2095-2145.  The planner takes the batches in input order (= the reference's
// -t 1 order), lays the records of a batch out in the output file and hands runs of them to the fill threads (MappedSink);
// or, for the other kinds of output, gathers the pieces and writes them itself (Output).
void Run::writer_body()
{
    using Emit = Batch::Emit;
    CpuScope cpu(CPU_PLANNER);
    const std::string lead(1, fastq_out ? '@' : '>'), nl("\n"), sep("\n+\n");
    std::string name;
    std::map<uint64_t, std::shared_ptr<Batch>> held;               // batches that arrived ahead of their turn
    uint64_t want = 0, in_seen = 0;
    size_t open_feeders = ctxs.size();
    for (;;) {
        std::shared_ptr<Batch> b;
        auto it = held.find(want);
        if (it != held.end()) { b = std::move(it->second); held.erase(it); }
        else {
            if (open_feeders == 0) break;
            const double i0 = now_s();
            b = to_writer->get();
            t_widle += now_s() - i0;
            if (!b) { open_feeders--; continue; }
            if (b->id != want) { const uint64_t id = b->id; held[id] = std::move(b); continue; }
        }
        want++;
        const double w0 = now_s();
        const bool fill = sink.is_open();
        uint64_t at = 0;
        for (size_t r = 0; r < b->recs.size(); r++) {
            int pass_num = 1;
            const tgsf_read_result& rr = b->res[r];
            const Rec& rec = b->recs[r];
            const std::string_view rname(rec.name, rec.name_len);
            for (uint32_t f = rr.frag_begin; f < rr.frag_begin + rr.n_frags; f++) {
                const tgsf_fragment& fr = b->frags[f];
                if (!(fr.flags & TGSF_FF_PASS)) continue;
                if (o.downsample) {                                // kept in memory instead of a tmp file (:3129-3137)
                    clean_recs.push_back({rname, pass_num++, rec.seq + fr.start, rec.qual + fr.start, (uint32_t)fr.len});
                    clean_bases += (uint64_t)fr.len;
                    clean_lens.push_back(fr.len);
                    continue;
                }
                clean_bases += (uint64_t)fr.len;
                clean_lens.push_back(fr.len);
                if (o.only_qc) { pass_num++; continue; }
                if (fill) {
                    b->em.push_back({(uint32_t)r, f, pass_num, at});
                    size_t nlen = rname.size();
                    if (pass_num >= 2) { int v = pass_num; nlen += 1; while (v) { nlen++; v /= 10; } }
                    at += 1 + nlen + 1 + (uint64_t)fr.len + (fastq_out ? 3 + (uint64_t)fr.len : 0) + 1;
                    pass_num++;
                    continue;
                }
                out.text(lead);
                if (pass_num < 2) out.piece(rname.data(), rname.size());
                else { name.clear(); append_name(name, rname, pass_num); out.text(name); }
                pass_num++;
                out.text(nl);
                out.piece(rec.seq + fr.start, (size_t)fr.len);
                if (fastq_out) {
                    out.text(sep);
                    out.piece(rec.qual + fr.start, (size_t)fr.len);
                }
                out.text(nl);
                out.end_record();
            }
        }
        in_seen += b->span;
        if (fill && at) {
            if (sink.planned() + at > sink.capacity()) die("output more than four times the size of the input: larger than the space mapped for it (TGSF_WRITER=writev writes such a file)");
            {
                // How far the file will go: what is left of the input times the share of it that was written so far
                // (plus a little).  (A streamed input's text size is estimated from the share of the file decoded so far.)
                const double share = in_seen ? (double)(sink.planned() + at) / (double)in_seen : 1.0;
                const double sh = stream_share.load();
                const uint64_t in_total = !streaming ? (uint64_t)text_size
                                        : (uint64_t)((double)stream_text.load() / (sh > 1e-6 ? sh : 1e-6));
                uint64_t goal = sink.planned() + at + (uint64_t)(share * 1.02 * (double)(in_total - std::min<uint64_t>(in_seen, in_total)));
                goal = std::min<uint64_t>(std::max<uint64_t>(goal, sink.planned() + at), sink.capacity());
                reserver->want(goal, sink.planned() + at);
                const double d0 = now_s();
                reserver->wait_ready(sink.planned() + at);            // instantiated AND mapped: the fill jobs take no fault
                t_drain += now_s() - d0;
            }
            b->dst = sink.place(at);
            b->out_bytes = at;
            const size_t n = b->em.size();
            const int parts = (int)std::min<size_t>((size_t)fill_threads, std::max<size_t>(1, at / fill_min));
            b->left = parts;
            size_t lo = 0;
            for (int k = 0; k < parts; k++) {                      // byte-balanced runs of records
                size_t hi = n;
                if (k + 1 < parts) {
                    const uint64_t target = at / (uint64_t)parts * (uint64_t)(k + 1);
                    hi = (size_t)(std::lower_bound(b->em.begin() + (long)lo, b->em.end(), target,
                                                   [](const Emit& e, uint64_t tgt) { return e.at < tgt; }) - b->em.begin());
                }
                pool->add([this, b, lo, hi] { fill_job(b, lo, hi); });
                lo = hi;
            }
        }
        if (!o.only_qc && !fill) out.flush_iov();                   // the batch (and its views) goes away
        if (!(fill && at)) batch_done(b);                           // written (or nothing to write)
        t_write += now_s() - w0;
    }
    if (!o.downsample) std::sort(clean_lens.begin(), clean_lens.end());   // for the statistics (:3182), beside the last fill jobs
}

// A downsampling run writes its output only after the whole filter pass (the selection needs every fragment's length,
// :2297-2344) -- but the file can be instantiated meanwhile: while the filter pass is busy with the link to the device,
// pages for what the selection may keep are reserved and mapped (a quarter of the input at most, and no more than twice
// the bases asked for with -g/-d; a surplus is cut off at the end).  Large plain outputs only.
bool Run::open_dsink(uint64_t capacity, uint64_t speculative)
{
    const char* w = getenv("TGSF_WRITER");
    if (o.out_gz || o.out_file.empty() || (w && !strcmp(w, "writev"))) return false;
    std::unique_ptr<MappedSink> d(new MappedSink);
    if (!d->open(out_path, capacity)) return false;
    dsink = std::move(d);
    dpop.reset(new Pool(populate_threads, CPU_POPULATE));
    dres.reset(new Reserver(*dsink, *dpop, stride_bytes, false));
    dres->start(speculative);
    return true;
}

}  // namespace host
