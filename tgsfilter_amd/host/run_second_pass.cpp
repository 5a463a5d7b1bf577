// run_second_pass.cpp -- the second pass of a downsampling run (run.h; DownSampleTask, src/TGSFilter.cpp:2346-2568): the kept
// records go through the QC tallies again (on a thread of their own) while this one writes them, in input order.
#include "run.h"

namespace host {

void Run::second_pass(const std::vector<char>& keep)
{
    const Api& L = *api;
    tgsf_params qp = p;
    qp.filter = 0; qp.only_qc = 1; qp.n_adapters = 0; qp.min_repeat = 0;
    // The reference's second pass re-reads what the filter pass wrote (:3129-3137): after a FASTA output
    // (-f, or FASTA input) the records carry no qualities, so this pass takes the count-only tallies.
    const bool down_no_qual = fasta_in || (run_filter_pass && !fastq_out);
    qp.no_qual = down_no_qual ? 1 : 0;
    qp.max_batch_bases = (1ull << 30); qp.max_batch_reads = 1u << 16;
    if (const char* e = knob("TGSF_DOWN_BATCH_BYTES")) { const long long v = atoll(e); if (v > (1 << 20) + 65536) qp.max_batch_bases = (uint64_t)v; }   // test knob: several slices of a small input
    tgsf_ctx *qctx = nullptr, *qctx2 = nullptr;
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
        const double s0 = now_s();
        if (L.submit(qctx, &bi, &bo) != TGSF_OK) die(L.last_error(qctx));
        t_dsubmit += now_s() - s0; if (!n_dsubmit++) t_dfirst = now_s() - s0;
        bs.clear(); bq.clear(); boff.clear(); blen.clear();
    };
    // The kept records go to the device on a thread of their own while this one writes them.  Where they make up a fair
    // part of the text between them they are read in place (the text itself is the batch, as in the filter pass: one
    // copy to the device, none on the host); a very thin selection is packed first.
    // (the writer took the batches as they came back from the feeders: the kept records are sorted by address first)
    std::vector<uint32_t> by_addr;
    uint64_t kept_bytes = 0, kept_span = 0;
    for (size_t i = 0; i < clean_recs.size(); i++) {
        if (!keep[i]) continue;
        by_addr.push_back((uint32_t)i);
        kept_bytes += (down_no_qual ? 1u : 2u) * (uint64_t)clean_recs[i].len;
    }
    std::sort(by_addr.begin(), by_addr.end(), [&](uint32_t x, uint32_t y) { return clean_recs[x].seq < clean_recs[y].seq; });
    if (!by_addr.empty()) {
        const CleanRec& a = clean_recs[by_addr.front()];
        const CleanRec& z = clean_recs[by_addr.back()];
        kept_span = (uint64_t)((down_no_qual ? z.seq : std::max(z.seq, z.qual)) + z.len - a.seq);
        for (uint32_t i : by_addr) if (!down_no_qual && clean_recs[i].qual < clean_recs[i].seq) kept_span = 0;   // (never: FASTQ text)
    }
    const char* force = knob("TGSF_DOWN_QC");                     // "text" / "packed": tests run both ways
    const bool in_place = kept_span > 0 && (force ? !strcmp(force, "text") : kept_bytes * 12 >= kept_span);   // (packing runs at a tenth of the copy to the device)
    d_in_place = in_place; d_kept = (double)kept_bytes; d_span = (double)kept_span;
    std::thread qc_pass([&] {
        CpuScope cpu(CPU_DOWNSAMPLE);
        const double q0 = now_s();
        if (L.create(&qp, o.devices[0], &qctx) != TGSF_OK) die(L.last_error(nullptr));
        t_dcreate = now_s() - q0;
        if (in_place) {
            // Slices of the text (up to 1 GB each, the kept records indexed in place) go to the device from two feeders
            // with a context each when there is much of it: one tgsf_submit stream moves 22-29 GB/s over the link, two
            // together about what it carries (as in the filter pass).
            struct TextBatch { const char* base = nullptr; uint64_t span = 0; std::vector<uint64_t> off, qoff; std::vector<uint32_t> len; };
            int workers = kept_span >= (2ull << 30) ? 2 : 1;
            if (const char* e = knob("TGSF_DOWN_FEEDERS")) workers = atoi(e) >= 2 ? 2 : 1;      // tests run both on small inputs
            Channel<std::shared_ptr<TextBatch>> todo(2);
            std::mutex tm;
            auto work = [&](tgsf_ctx* c) {
                std::vector<tgsf_read_result> res;
                std::vector<tgsf_fragment> fr(16);
                for (;;) {
                    std::shared_ptr<TextBatch> tb = todo.get();
                    if (!tb) break;
                    res.resize(tb->len.size());
                    tgsf_batch_in bi; memset(&bi, 0, sizeof bi);
                    bi.seq = bi.qual = reinterpret_cast<const uint8_t*>(tb->base);
                    bi.offsets = tb->off.data(); bi.qual_offsets = tb->qoff.data(); bi.lengths = tb->len.data();
                    bi.n_reads = (uint32_t)tb->len.size(); bi.n_bytes = tb->span;
                    tgsf_batch_out bo{res.data(), fr.data(), (uint32_t)fr.size(), 0};
                    const double s0 = now_s();
                    if (L.submit(c, &bi, &bo) != TGSF_OK) die(L.last_error(c));
                    std::lock_guard<std::mutex> l(tm);
                    t_dsubmit += now_s() - s0; if (!n_dsubmit++) t_dfirst = now_s() - s0;
                }
            };
            std::thread second;
            if (workers == 2) second = std::thread([&] {
                CpuScope cpu2(CPU_DOWNSAMPLE);
                if (L.create(&qp, o.devices[0], &qctx2) != TGSF_OK) die(L.last_error(nullptr));
                work(qctx2);
            });
            std::thread first([&] { CpuScope cpu2(CPU_DOWNSAMPLE); work(qctx); });
            std::shared_ptr<TextBatch> tb(new TextBatch);
            auto flush_text = [&] {
                if (tb->len.empty()) return;
                todo.put(std::move(tb));
                tb.reset(new TextBatch);
            };
            for (uint32_t i : by_addr) {
                const CleanRec& c = clean_recs[i];
                const char* e = down_no_qual ? c.seq + c.len : c.qual + c.len;
                if (tb->base && ((uint64_t)(e - tb->base) > qp.max_batch_bases - (1u << 20) || tb->len.size() >= qp.max_batch_reads)) flush_text();
                if (!tb->base) tb->base = c.seq;
                tb->off.push_back((uint64_t)(c.seq - tb->base));
                tb->qoff.push_back((uint64_t)((down_no_qual ? c.seq : c.qual) - tb->base));
                tb->len.push_back(c.len);
                tb->span = std::max(tb->span, (uint64_t)(e - tb->base));
                if (tb->span > qp.max_batch_bases) die("record larger than a batch");
            }
            flush_text();
            for (int k = 0; k < workers; k++) todo.put(nullptr);
            first.join();
            if (second.joinable()) second.join();
        } else {
            const size_t room = (size_t)std::min<uint64_t>(qp.max_batch_bases, kept_bytes / (down_no_qual ? 1 : 2) + 16 * by_addr.size() + 128);
            bs.reserve(room); bq.reserve(room);
            for (size_t i = 0; i < clean_recs.size(); i++) {
                if (!keep[i]) continue;
                const CleanRec& c = clean_recs[i];
                if (bs.size() + c.len > qp.max_batch_bases - (1u << 20) || blen.size() >= qp.max_batch_reads) run();
                const size_t o0 = (bs.size() + 15) & ~size_t(15);
                bs.resize(o0); bq.resize(o0);
                bs.insert(bs.end(), c.seq, c.seq + c.len);
                if (!down_no_qual) bq.insert(bq.end(), c.qual, c.qual + c.len); else bq.resize(bs.size());
                boff.push_back(o0); blen.push_back(c.len);
            }
            run();
        }
        t_dqc = now_s() - q0 - t_dcreate;
    });
    const double w0 = now_s();
    const std::string lead(1, fastq_out ? '@' : '>'), nl("\n"), sep("\n+\n");
    std::string name;
    // A regular file of some size is written as the filter pass writes its own (MappedSink): every record's place is
    // known, so the pages are instantiated in one go and threads copy the records in -- a single writev stream is a
    // 6-GB/s copy under the inode lock.
    bool down_mapped = false;
    {
        const char* mn = knob("TGSF_DOWN_MAP_MIN");              // tests force the mapped way on small outputs
        const uint64_t map_min = mn ? strtoull(mn, nullptr, 10) : (256ull << 20);
        std::vector<uint32_t> kept;
        std::vector<uint64_t> at;
        uint64_t total_out = 0;
        const char* w2 = getenv("TGSF_WRITER");
        if (!o.out_gz && !o.out_file.empty() && !(w2 && !strcmp(w2, "writev"))) {
            for (size_t i = 0; i < clean_recs.size(); i++) {
                if (!keep[i]) continue;
                const CleanRec& c = clean_recs[i];
                size_t nm = c.name.size();
                if (c.pass_num >= 2) { name.clear(); append_name(name, c.name, c.pass_num); nm = name.size(); }
                kept.push_back((uint32_t)i); at.push_back(total_out);
                total_out += 1 + nm + 1 + c.len + (fastq_out ? 3 + (uint64_t)c.len : 0) + 1;
            }
        }
        if (!dsink && total_out >= std::max<uint64_t>(map_min, 1)) open_dsink(total_out, 0);
        if (dsink && !kept.empty()) {
            // stride by stride, as the filter pass writes its own output: the reserver instantiates and maps the file
            // ahead, the records of every piece that is ready are copied in by the pool's threads
            dres->want(total_out, total_out);
            const int T = std::max(1, std::min(o.n_thread, 16));
            Pool dfill(T);
            char* const base = dsink->place(0);
            // (one process: the mappings of written pieces are dropped behind the fill jobs by one thread, as in the filter pass)
            Channel<std::pair<const char*, uint64_t>> dropped(1 << 12);
            std::thread dropper([&] {
                CpuScope cpu2(CPU_RELEASER);
                for (;;) {
                    const std::pair<const char*, uint64_t> r = dropped.get();
                    if (!r.first) break;
                    MappedSink::release(r.first, r.second);
                }
            });
            const uint64_t piece = std::max<uint64_t>(1, std::min<uint64_t>(64ull << 20, stride_bytes / 4 + 1));
            size_t r0 = 0;
            while (r0 < kept.size()) {
                size_t r1 = (size_t)(std::lower_bound(at.begin() + (long)r0, at.end(), at[r0] + piece) - at.begin());
                if (r1 <= r0) r1 = r0 + 1;
                const uint64_t end = r1 < at.size() ? at[r1] : total_out;
                dres->wait_ready(end);
                dfill.add([&, r0, r1] {
                    std::string nm;
                    for (size_t r = r0; r < r1; r++) {
                        const CleanRec& c = clean_recs[kept[r]];
                        char* w = base + at[r];
                        *w++ = fastq_out ? '@' : '>';
                        if (c.pass_num < 2) { memcpy(w, c.name.data(), c.name.size()); w += c.name.size(); }
                        else { nm.clear(); append_name(nm, c.name, c.pass_num); memcpy(w, nm.data(), nm.size()); w += nm.size(); }
                        *w++ = '\n';
                        stream_copy(w, c.seq, c.len); w += c.len;
                        if (fastq_out) { memcpy(w, "\n+\n", 3); w += 3; stream_copy(w, c.qual, c.len); w += c.len; }
                        *w++ = '\n';
                    }
                    stream_fence();
                    if (release_output) dropped.put({base + at[r0], (r1 < at.size() ? at[r1] : total_out) - at[r0]});
                });
                r0 = r1;
            }
            dfill.finish();
            dropped.put({nullptr, 0});
            dropper.join();
            dres->finish();
            dpop->finish();
            dsink->place(total_out);
            dsink->close();
            down_mapped = true;
        } else if (dsink) {                                        // nothing kept: an empty file
            dres->finish(); dpop->finish(); dsink->place(0); dsink->close(); down_mapped = true;
        }
    }
    for (size_t i = 0; i < clean_recs.size() && !down_mapped; i++) {
        if (!keep[i]) continue;
        const CleanRec& c = clean_recs[i];
        out.text(lead);
        if (c.pass_num < 2) out.piece(c.name.data(), c.name.size());
        else { name.clear(); append_name(name, c.name, c.pass_num); out.text(name); }
        out.text(nl);
        out.piece(c.seq, c.len);
        if (fastq_out) { out.text(sep); out.piece(c.qual, c.len); }
        out.text(nl);
        out.end_record();
    }
    t_dwrite = now_s() - w0; d_mapped = down_mapped;
    qc_pass.join();
    uint64_t qnw = 0; int32_t qbc = 0; uint32_t qnb = 0;
    L.counters_len(qctx, &qnw, &qbc, &qnb);
    down_t.resize(qnw);
    if (L.counters(qctx, down_t.data(), qnw) != TGSF_OK) die(L.last_error(qctx));
    L.destroy(qctx);
    if (qctx2) {                                                   // the second feeder's tallies: sums, maxima for the "rows used" words
        std::vector<uint64_t> t2(qnw);
        if (L.counters(qctx2, t2.data(), qnw) != TGSF_OK) die(L.last_error(qctx2));
        L.destroy(qctx2);
        for (uint64_t i = 0; i < qnw; i++)
            down_t[i] = (i >= TGSF_CTR_ROWS && i < TGSF_CTR_ROWS + 4) ? std::max(down_t[i], t2[i]) : down_t[i] + t2[i];
    }
    if (sharded) {                                                 // the second pass's tallies of the whole job, on rank 0
        BlobOut mine;
        mine.vec(pack_rows(down_t, qbc, qnb));
        const std::vector<std::string> all = link.gather(mine.s);
        for (int k = 1; k < (int)all.size(); k++) {
            BlobIn in2(all[(size_t)k]);
            std::vector<uint64_t> ru;
            in2.vec(ru);
            add_rows(down_t, ru, qbc, qnb);
        }
    }
}

}  // namespace host
