// shard.h -- one process per GPU over ONE input file: `tgsfilter --ranks N` (this program forks its N ranks) or
// `tgsfilter --shard r/N --rendezvous <path>` (a launcher started them: torchrun, mpirun, a batch system).
//
// Replaces, across GPUs, the reference's fan-out of reads over worker threads (src/TGSFilter.cpp:1808-1842) and the merge
// of their tallies at the end (:3208-3213, :2673-2725, :2586-2597).  Reads are independent, so the DATA path has no
// exchange at all:
//   * rank r filters the r-th of N byte ranges of the input text, cut at record boundaries (find_record_start; the
//     cut is verified from both sides, see main.cpp: a rank's last record must end exactly where the next rank's
//     first record begins, or the run stops -- nothing is ever cut differently from the reference's sequential reader);
//   * rank r writes <out>.part<r>: the parts, concatenated in rank order, are byte for byte the file the single
//     process writes (= the reference's -t 1 order) -- N files from N processes, no shared inode (ONE
//     tmpfs file is bound by the kernel's page instantiation under one inode lock whatever the number of GPUs);
//     parts numbered from N upward that an earlier job with more ranks left beside them are removed, with a warning;
//   * the pre-pass (quality encoding, trims, adapter identification) runs on rank 0 only and its constants are
//     broadcast;
//   * at the end the tally vectors are summed -- on the devices, one RCCL all-reduce over xGMI (libtgsf_rccl), when
//     every rank has a GPU of its own; over the ranks' sockets when ranks share a GPU (the 1-GPU test set-up: RCCL
//     refuses two ranks on one device) -- and rank 0, which also receives the read-length vectors, prints the
//     statistics and writes the one report;
//   * a downsampling run (-g/-d, -r, -R) selects among ALL reads: rank 0 receives every rank's kept fragments (name, length,
//     in input order), makes the reference's selection over them, container for container, and hands every rank the keep
//     flags of its own; the second (QC) pass and the writing are per rank again, its tallies summed like the first's.
// What travels between the ranks on the host is small (a few hundred bytes of constants, 4 bytes per read of
// lengths, the tally rows in use): a star of stream sockets around rank 0 (RankLink) carries it.
#pragma once
#include <poll.h>
#include <signal.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fatal.h"

namespace host {

class RankLink {
public:
    int rank = 0, world = 1;
    bool active() const { return world > 1; }
    // launcher mode: the sockets were made before the fork (rank 0: one per other rank; the others: one, to rank 0)
    void adopt(int r, int w, std::vector<int> fds) { rank = r; world = w; fds_ = std::move(fds); }
    // external launcher: rank 0 listens on a unix socket at `path`, the others connect (and wait for it to appear)
    void rendezvous(int r, int w, const std::string& path, double timeout_s = 120.0) {
        rank = r; world = w;
        if (w <= 1) return;
        sockaddr_un sa;
        memset(&sa, 0, sizeof sa);
        sa.sun_family = AF_UNIX;
        if (path.size() >= sizeof sa.sun_path) die("--rendezvous: path too long for a unix socket");
        memcpy(sa.sun_path, path.c_str(), path.size());
        if (r == 0) {
            struct stat st;
            if (lstat(path.c_str(), &st) == 0) {        // a socket a job that was killed left behind goes; anything else is the user's
                if (!S_ISSOCK(st.st_mode)) die("--rendezvous: " + path + " exists and is not a socket");
                if (st.st_uid != geteuid()) die("--rendezvous: the socket at " + path + " belongs to another user");
                // ... and one that somebody still listens on is a LIVE job's: it stays
                const int probe = socket(AF_UNIX, SOCK_STREAM, 0);
                const bool live = probe >= 0 && connect(probe, (sockaddr*)&sa, sizeof sa) == 0;
                if (probe >= 0) ::close(probe);
                if (live) die("--rendezvous: another job is listening at " + path + " (give this one its own --rendezvous path)");
                if (unlink(path.c_str()) != 0) die("--rendezvous: cannot remove the stale socket at " + path + ": " + strerror(errno));
            }
            const int ls = socket(AF_UNIX, SOCK_STREAM, 0);
            if (ls < 0 || bind(ls, (sockaddr*)&sa, sizeof sa) != 0 || listen(ls, w) != 0)
                die("--rendezvous: cannot listen on " + path + ": " + strerror(errno));
            fds_.assign((size_t)w, -1);
            for (int k = 1; k < w; k++) {
                pollfd pf{ls, POLLIN, 0};
                if (poll(&pf, 1, (int)(timeout_s * 1000)) <= 0) { unlink(path.c_str()); die("--rendezvous: " + std::to_string(w - k) + " of " + std::to_string(w) + " ranks did not show up at " + path); }
                const int c = accept(ls, nullptr, nullptr);
                int32_t who = -1;
                pollfd pc{c, POLLIN, 0};                // (a peer that connects and says nothing does not hold the job for ever)
                if (c < 0 || !same_user(c) || poll(&pc, 1, (int)(timeout_s * 1000)) <= 0 || !io(c, &who, sizeof who, false) || who < 1 || who >= w || fds_[(size_t)who] >= 0) { unlink(path.c_str()); die("--rendezvous: bad greeting at " + path + " (a peer of another user, or not a rank of this job)"); }
                fds_[(size_t)who] = c;
            }
            ::close(ls);
            unlink(path.c_str());                       // everybody is connected: the name is not needed any more
        } else {
            int c = -1;
            for (double waited = 0;; waited += 0.05) {
                c = socket(AF_UNIX, SOCK_STREAM, 0);
                if (c >= 0 && connect(c, (sockaddr*)&sa, sizeof sa) == 0) break;
                if (c >= 0) ::close(c);
                if (waited > timeout_s) die("--rendezvous: rank 0 is not listening at " + path);
                usleep(50000);
            }
            const int32_t who = r;
            // (the constants of the run -- trims, adapters, the quality threshold -- come from whoever listens there: the same user only)
            if (!same_user(c)) die("--rendezvous: the process listening at " + path + " belongs to another user");
            if (!io(c, &who, sizeof who, true)) die("--rendezvous: cannot greet rank 0");
            fds_.assign(1, c);
        }
    }
    // rank 0's bytes to everybody
    void bcast(std::string& blob) {
        if (!active()) return;
        if (rank == 0) { for (int k = 1; k < world; k++) send_blob(fds_[(size_t)k], blob, k); }
        else recv_blob(fds_[0], blob, 0);
    }
    // everybody's bytes to rank 0 (there: one entry per rank, its own included; elsewhere: empty)
    std::vector<std::string> gather(const std::string& mine) {
        std::vector<std::string> all;
        if (rank == 0) {
            all.resize((size_t)world);
            all[0] = mine;
            for (int k = 1; k < world; k++) recv_blob(fds_[(size_t)k], all[(size_t)k], k);
        } else send_blob(fds_[0], mine, 0);
        return all;
    }
    // rank 0's k-th blob to rank k (its own stays with it)
    void scatter(std::vector<std::string>& blobs, std::string& mine) {
        if (!active()) { if (!blobs.empty()) mine = blobs[0]; return; }
        if (rank == 0) { mine = blobs[0]; for (int k = 1; k < world; k++) send_blob(fds_[(size_t)k], blobs[(size_t)k], k); }
        else recv_blob(fds_[0], mine, 0);
    }
    uint64_t max_u64(uint64_t v) {
        if (!active()) return v;
        std::string b((const char*)&v, sizeof v);
        const std::vector<std::string> all = gather(b);
        if (rank == 0) { for (const std::string& s : all) { uint64_t x = 0; memcpy(&x, s.data(), sizeof x); v = x > v ? x : v; } b.assign((const char*)&v, sizeof v); }
        bcast(b);
        memcpy(&v, b.data(), sizeof v);
        return v;
    }
    void barrier() { (void)max_u64(0); }
    void close_all() { for (int& f : fds_) if (f >= 0) { ::close(f); f = -1; } }
private:
    // the peer of a connected unix socket runs as this process's user (SO_PEERCRED)
    static bool same_user(int fd) {
        struct ucred cr;
        socklen_t len = sizeof cr;
        return getsockopt(fd, SOL_SOCKET, SO_PEERCRED, &cr, &len) == 0 && len == sizeof cr && cr.uid == geteuid();
    }
    static bool io(int fd, const void* p, size_t n, bool wr) {
        char* c = (char*)const_cast<void*>(p);
        while (n) {
            const ssize_t k = wr ? ::send(fd, c, n, MSG_NOSIGNAL) : ::recv(fd, c, n, 0);
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) return false;
            c += k; n -= (size_t)k;
        }
        return true;
    }
    void send_blob(int fd, const std::string& b, int peer) {
        const uint64_t n = b.size();
        if (!io(fd, &n, sizeof n, true) || !io(fd, b.data(), b.size(), true)) lost(peer);
    }
    void recv_blob(int fd, std::string& b, int peer) {
        uint64_t n = 0;
        if (!io(fd, &n, sizeof n, false)) lost(peer);
        // (what the ranks exchange is small: constants, 4 bytes per read of lengths, names of kept fragments, tally rows)
        if (n > kMaxBlob) die("rank " + std::to_string(rank) + ": a message of " + std::to_string(n) + " bytes from rank " + std::to_string(peer) + ": not one of this job's");
        b.resize((size_t)n);
        if (n && !io(fd, &b[0], (size_t)n, false)) lost(peer);
    }
    static constexpr uint64_t kMaxBlob = 16ull << 30;
    [[noreturn]] void lost(int peer) { die("rank " + std::to_string(rank) + ": rank " + std::to_string(peer) + " of the job is gone (it ended with an error, or was killed)"); }
    std::vector<int> fds_;
};

// little helpers to put values into / take them out of a blob
struct BlobOut {
    std::string s;
    template <class T> void pod(const T& v) { s.append((const char*)&v, sizeof v); }
    void str(const std::string& v) { const uint64_t n = v.size(); pod(n); s.append(v); }
    template <class T> void vec(const std::vector<T>& v) { const uint64_t n = v.size(); pod(n); if (n) s.append((const char*)v.data(), n * sizeof(T)); }
};
struct BlobIn {
    const std::string& s;
    size_t at = 0;
    explicit BlobIn(const std::string& b) : s(b) {}
    void need(size_t n) const { if (at + n > s.size()) die("a message between the ranks of the job is shorter than it says"); }
    template <class T> void pod(T& v) { need(sizeof v); memcpy(&v, s.data() + at, sizeof v); at += sizeof v; }
    void str(std::string& v) { uint64_t n = 0; pod(n); need((size_t)n); v.assign(s.data() + at, (size_t)n); at += (size_t)n; }
    template <class T> void vec(std::vector<T>& v) { uint64_t n = 0; pod(n); need((size_t)n * sizeof(T)); v.resize((size_t)n); if (n) memcpy(v.data(), s.data() + at, (size_t)n * sizeof(T)); at += (size_t)n * sizeof(T); }
};

// First byte of the first record that starts at or after `from`: the start of a line that begins with '@' whose
// third line begins with '+' and whose second line is not empty and as long as the fourth (FASTQ; a quality line may
// begin with '@' too, but then the line after the next is a sequence line) -- or with '>' and a non-empty next line
// (FASTA).  size when there is none.  This only PROPOSES a cut; main.cpp verifies it against the record reader's
// own view from both sides before anything is filtered.
inline size_t find_record_start(const char* data, size_t size, size_t from, bool fastq)
{
    if (from == 0) return 0;
    if (from >= size) return size;
    auto next_line = [&](size_t p) {                   // start of the line after the one p is in (size: none)
        const void* nl = p < size ? memchr(data + p, '\n', size - p) : nullptr;
        return nl ? (size_t)((const char*)nl - data) + 1 : size;
    };
    auto line_len = [&](size_t p) {                    // without "\n" / "\r\n"
        const size_t e = next_line(p);
        size_t n = e - p;
        if (n && data[p + n - 1] == '\n') n--;
        if (n && data[p + n - 1] == '\r') n--;
        return n;
    };
    size_t p = data[from - 1] == '\n' ? from : next_line(from);
    while (p < size) {
        if (fastq) {
            if (data[p] == '@') {
                const size_t l2 = next_line(p), l3 = next_line(l2), l4 = next_line(l3);
                if (l3 < size && data[l3] == '+' && l4 < size && line_len(l2) > 0 && line_len(l2) == line_len(l4)) return p;
            }
        } else if (data[p] == '>') {
            const size_t l2 = next_line(p);
            if (l2 < size && line_len(l2) > 0) return p;
        }
        p = next_line(p);
    }
    return size;
}

// `--ranks N`: fork the N ranks (before any thread or GPU state exists in this process).  Returns in every CHILD with
// `link` set up; the parent waits for the children and leaves with the first non-zero exit status (ending the others).
inline void fork_ranks(int world, RankLink& link)
{
    std::vector<int> hub((size_t)world, -1), spoke((size_t)world, -1);
    for (int k = 1; k < world; k++) {
        int sv[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) die(std::string("socketpair: ") + strerror(errno));
        hub[(size_t)k] = sv[0]; spoke[(size_t)k] = sv[1];
    }
    fflush(nullptr);
    std::vector<pid_t> kids((size_t)world, -1);
    for (int r = 0; r < world; r++) {
        const pid_t pid = fork();
        if (pid < 0) { for (int j = 0; j < r; j++) kill(kids[(size_t)j], SIGTERM); die(std::string("fork: ") + strerror(errno)); }
        if (pid == 0) {
            std::vector<int> fds;
            if (r == 0) { fds = hub; for (int k = 1; k < world; k++) ::close(spoke[(size_t)k]); }
            else {
                fds.assign(1, spoke[(size_t)r]);
                for (int k = 1; k < world; k++) { ::close(hub[(size_t)k]); if (k != r) ::close(spoke[(size_t)k]); }
            }
            link.adopt(r, world, std::move(fds));
            // (several processes on the GPUs of one node: the host driver of this pool only supports dmabuf IPC -- without this
            // RCCL's and HIP's cross-process sharing fails with "hipIpcGetMemHandle: invalid argument"; a value already set stays)
            setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
            return;
        }
        kids[(size_t)r] = pid;
    }
    for (int k = 1; k < world; k++) { ::close(hub[(size_t)k]); ::close(spoke[(size_t)k]); }
    // SIGINT / SIGTERM reach the whole process group when they come from a terminal; one sent to this process alone is passed on
    static std::vector<pid_t>* g_kids = nullptr;
    g_kids = &kids;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = [](int sig) { if (g_kids) for (pid_t p : *g_kids) if (p > 0) kill(p, sig); };
    sigaction(SIGINT, &sa, nullptr);
    sigaction(SIGTERM, &sa, nullptr);
    int code = 0, left = world;
    while (left > 0) {
        int st = 0;
        const pid_t p = wait(&st);
        if (p < 0) { if (errno == EINTR) continue; break; }
        left--;
        const int c = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
        for (pid_t& k : kids) if (k == p) k = -1;
        if (c != 0 && code == 0) {
            code = c;
            for (pid_t k : kids) if (k > 0) kill(k, SIGTERM);           // the others would wait for it for ever
        }
    }
    _exit(code);
}

}  // namespace host
