// host_common.h - what the three hosts of dump1090_amd share (round 6: main.cpp was 1,185 lines holding all of them):
//
//   main.cpp               the option parser (the reference's spellings and defaults for this path, dump1090.c:2869-2897,2922; :299-319)
//   host_single.cpp        ONE process, N devices: reader -> lanes -> resolver thread                       (run_single)
//   host_ranks.cpp         one PROCESS per GPU: the fork, rank 0's reader for a pipe / --loop, the lanes and the three-stage round loop
//   host_ranks_rccl.cpp      ... the record lists gathered to rank 0 over RCCL / xGMI (libmodes_gather.so)   (north_star's shape)
//   host_ranks_shared.cpp    ... or every rank resolving its own batches, confirmed through a shared mapping (--resolve-on-ranks)
//
// Header-only helpers (inline): options, the sink, the readers (a file in parallel slices, a pipe at the pace it delivers), a lane.
#ifndef MODES_HOST_COMMON_H
#define MODES_HOST_COMMON_H

#include <atomic>
#include <cerrno>
#include <csignal>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <algorithm>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <fcntl.h>
#include <poll.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/modes_gather.h"
#include "../../include/modes_gfx950.h"
#include "../../include/modes_host.h"

namespace modes_cli {

struct Options {
    std::string filename;
    bool loop = false, raw = false, onlyaddr = false, stats = false, sbs = false, raw_net = false, timing = false;
    int fix_errors = 1, check_crc = 1, aggressive = 0;
    std::vector<int> devices;              // HIP ordinals, one per "GPU" of the split (the same ordinal may repeat)
    uint64_t batch_blocks = 512;           // 128 MiB of samples per GPU call
    int read_threads = 16;                 // parallel pread() slices for regular files (8 GiB file, 256-core host: 42 GB/s at 8, 45 at 32, 30 at 64)
    int depth = 2;                         // batches in flight per device (lanes = depth x devices).  Two: one is read / copied to the GPU while
                                           // the other one's kernels run and its records are resolved.  A third lane only helps when the resolve of
                                           // a batch takes longer than reading the next one (message-dense input), and costs 128 MiB more pinned
                                           // memory whose set-up competes with the first reads: 8 GiB file, whole process 0.57 s against 0.71
    bool use_mmap = true;                  // regular files: copy out of a mapping of the file instead of pread()
    int resolve_threads = 8;               // --raw only: pieces of a batch resolved in parallel (modes_host_resolve_raw_mt)
    bool clean_exit = false;               // --clean-exit: free everything before returning (default: the process just ends)
    int ranks = 0;                         // --ranks N: one process per GPU, record lists gathered to rank 0 over RCCL
    uint32_t gather_cap = 1u << 18;        // --gather-records: records per rank and round the gather buffers hold
    char **argv = nullptr;                 // for the one restart --ranks may need (the other IPC mode)
    uint32_t gather_cands = 0;             // --gather-candidates: preamble positions per rank and round (--stats); 0 = positions of a batch / 64
    bool resolve_on_ranks = false;         // --resolve-on-ranks: with --ranks and --raw, every rank resolves its own batch; only text reaches rank 0
    bool read_threads_given = false, resolve_threads_given = false;   // (else: clamped to the process's CPU budget, modes_host_cpu_budget)
    int flush_ms = 66;                     // --flush-ms: a pipe's batch is submitted when it is full OR this long after it began, whole buffers
                                           // only (one 256 KiB buffer is 65.5 ms of air time at 2 Msps: the reference's own cadence)
};

struct Sink {
    const Options *opt;
    modes_host *host;
    std::string out;
    modes_tracker *tracker;               // --sbs: aircraft table behind the BaseStation lines
};

inline void show_help() {
    printf(
        "--ifile <filename>       Read data from file (use '-' for stdin).\n"
        "--loop                   With --ifile, read the same file in a loop.\n"
        "--raw                    Show only messages hex values.\n"
        "--no-fix                 Disable single-bits error correction using CRC.\n"
        "--no-crc-check           Disable messages with broken CRC (discouraged).\n"
        "--aggressive             More CPU for more messages (two bits fixes, ...).\n"
        "--stats                  With --ifile print stats at exit. No other output.\n"
        "--onlyaddr               Show only ICAO addresses (testing purposes).\n"
        "--sbs                    Print the BaseStation lines the reference serves on port 30003.\n"
        "--raw-net                Print the raw lines the reference serves on port 30002.\n"
        "--gpu <ordinal>          HIP device to run on (default: 0).\n"
        "--gpus <n>               Split the stream over HIP devices 0..n-1 (batch b runs on device b mod n).\n"
        "--gpu-list <a,b,...>     The same with explicit ordinals; an ordinal may repeat (several contexts on one device).\n"
        "--ranks <n>              One PROCESS per GPU (this one forks n-1 more): rank r takes batches r, r+n, ... of a regular\n"
        "                         file on device r (or --gpu-list), the record lists are gathered to rank 0 over RCCL.\n"
        "                         A pipe (--ifile -) and --loop have ONE reader: rank 0 reads and hands every rank its batches\n"
        "                         through shared memory.  --gpus <n> (one process, the same devices) is the faster of the two for\n"
        "                         any input below ~96 GB: a communicator takes 1.6 s to start (a minute on a fresh box).\n"
        "--resolve-on-ranks       With --ranks and --raw / --onlyaddr / --raw-net / --stats: every rank resolves its own batches from\n"
        "                         a guessed whitelist, the ranks confirm each other in stream order through shared memory and rank 0\n"
        "                         prints their texts (--stats: adds their counters up) - no record leaves its rank, no communicator\n"
        "                         is made; the output is the same.  On a pipe the whitelist's 60 s run on rank 0's clock, read once\n"
        "                         per round of n batches.\n"
        "--gather-records <n>     With --ranks: records per rank and round the gather buffers hold (default: 262144).\n"
        "--gather-candidates <n>  With --ranks --stats: preamble positions per rank and round (default: a batch's positions / 64).\n"
        "--batch-blocks <n>       256 KiB buffers per GPU call (default: 512).  A file always fills its batches; a pipe\n"
        "                         (--ifile -, a FIFO) is served at the pace it delivers: see --flush-ms.\n"
        "--flush-ms <n>           Input that cannot seek: submit what has arrived - whole 256 KiB buffers - when the batch is\n"
        "                         full or <n> ms after it began (default: 66 = one buffer at 2 Msps).  A fast pipe still gets\n"
        "                         full batches; a live one is printed within two buffers, like the reference's own loop.\n"
        "--depth <n>              Batches in flight per device (default: 2; --ranks: at least 3).\n"
        "--read-threads <n>       Threads reading a regular file (default: 16, or what the CPU budget - affinity, cgroup quota - leaves).\n"
        "--no-mmap                Read a regular file with pread() instead of copying out of a mapping of it.\n"
        "--resolve-threads <n>    With --raw: threads that resolve one batch (default: 8, or what the CPU budget leaves; the listing\n"
        "                         does not depend on it).\n"
        "--timing                 Print a JSON line with the phase times to stderr.\n"
        "--clean-exit             Release every buffer, context and mapping before exiting (default: leave it to the process\n"
        "                         exit - unmapping 8 GiB and unpinning the buffers is a quarter of a short run's wall clock).\n"
        "--help                   Show this help.\n");
}

// useModesMessage (dump1090.c:1802-1820) for the non-interactive, non-network case.
inline void on_message(const struct modesMessage *mm, uint32_t, uint32_t, void *user) {
    Sink *s = static_cast<Sink *>(user);
    if (s->opt->stats || !modes_host_wants(s->host, mm)) return;
    char line[1024];
    int n;
    if (s->opt->sbs) {
        // dump1090.c:1806-1808 with an SBS client connected; the wall clock stamps the CPR frames like mstime()
        struct timeval tv;
        gettimeofday(&tv, nullptr);
        const modes_aircraft *a = modes_tracker_receive(s->tracker, mm, s->opt->check_crc,
                                                        (int64_t)tv.tv_sec * 1000 + tv.tv_usec / 1000);
        n = a ? modes_format_sbs(mm, a, line, sizeof line) : 0;
    }
    else if (s->opt->raw_net) n = modes_format_raw_net(mm, line);
    else if (s->opt->onlyaddr) n = modes_format_onlyaddr(mm, line);
    else if (s->opt->raw) n = modes_format_raw(mm, line);
    else                  n = modes_format_verbose(mm, s->opt->check_crc, line, sizeof line);   // dump1090.c:1333-1450
    s->out.append(line, (size_t)n);
}

// A fixed set of worker threads that run fn(0..n-1) and wait: the slices of one parallel file read.
class Pool {
public:
    explicit Pool(int n) {
        for (int t = 0; t < n; t++) workers_.emplace_back([this] { work(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    void run(int n, const std::function<void(int)> &fn) {
        std::unique_lock<std::mutex> g(m_);
        fn_ = &fn; next_ = 0; total_ = n; left_ = n;
        cv_.notify_all();
        done_.wait(g, [this] { return left_ == 0; });
        fn_ = nullptr;
    }
    int size() const { return (int)workers_.size(); }
private:
    void work() {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [this] { return stop_ || (fn_ && next_ < total_); });
            if (stop_) return;
            const int i = next_++;
            const std::function<void(int)> *fn = fn_;
            g.unlock();
            (*fn)(i);
            g.lock();
            if (--left_ == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *fn_ = nullptr;
    int next_ = 0, total_ = 0, left_ = 0;
    bool stop_ = false;
};

// A regular file is read by several threads at once (pread on disjoint slices): one thread copying
// out of the page cache is ~10x slower than the PCIe link that follows.  *got < want only at end of file.
inline bool read_parallel(Pool &pool, int fd, const uint8_t *map, size_t map_len, off_t *pos, uint8_t *dst, size_t want, size_t *got) {
    *got = 0;
    if (want == 0) return true;
    if (map) {                                                               // the file is mapped: plain copies, no system call per slice
        const size_t have = (size_t)*pos < map_len ? map_len - (size_t)*pos : 0, n = std::min(want, have);
        const int ns = pool.size();
        const size_t sl = (n / (size_t)ns + 4095) & ~(size_t)4095;
        const uint8_t *src = map + *pos;
        pool.run(ns, [&](int t) {
            const size_t lo = (size_t)t * sl;
            if (lo < n) memcpy(dst + lo, src + lo, std::min(sl, n - lo));
        });
        *got = n;
        *pos += (off_t)n;
        return true;
    }
    const int nslices = pool.size();
    const size_t slice = (want / (size_t)nslices + 4095) & ~(size_t)4095;
    std::vector<ssize_t> done((size_t)nslices, 0);
    const off_t pos0 = *pos;
    pool.run(nslices, [&](int t) {
        const size_t lo = (size_t)t * slice;
        if (lo >= want) return;
        const size_t n = std::min(slice, want - lo);
        size_t have = 0;
        while (have < n) {
            ssize_t r = pread(fd, dst + lo + have, n - have, pos0 + (off_t)(lo + have));
            if (r < 0) { if (errno == EINTR) continue; done[(size_t)t] = -1; return; }
            if (r == 0) break;
            have += (size_t)r;
        }
        done[(size_t)t] = (ssize_t)have;
    });
    for (int t = 0; t < nslices; t++) {
        const size_t lo = (size_t)t * slice;
        if (lo >= want) break;
        if (done[(size_t)t] < 0) return false;
        *got += (size_t)done[(size_t)t];
        if ((size_t)done[(size_t)t] < std::min(slice, want - lo)) break;      // end of file inside this slice
    }
    *pos += (off_t)*got;
    return true;
}

inline bool read_full(int fd, uint8_t *dst, size_t want, size_t *got) {
    *got = 0;
    while (*got < want) {
        ssize_t n = read(fd, dst + *got, want - *got);
        if (n < 0) { if (errno == EINTR) continue; return false; }
        if (n == 0) break;
        *got += (size_t)n;
    }
    return true;
}

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Input that cannot seek (a pipe, a FIFO, a socket): the reference prints a buffer's messages 65 ms after its samples arrived
// (dump1090.c:460-512 hands over ONE buffer; :2969-2990 decodes it at once), and a host that waits for 128 MiB before its first
// GPU call would sit on a 2 Msps stream for 33 s.  So a batch read from such an input ends when it is full, when the stream
// ends, or - at least one whole buffer being there - flush_s after the read began; until a whole buffer is there it waits
// without a deadline.  dst[0 .. have) already holds bytes (what the previous batch read beyond its last whole buffer).
// *n = bytes at dst afterwards, *eof = the stream has ended.
inline bool read_paced(int fd, uint8_t *dst, size_t have, size_t want, double flush_s, size_t *n, bool *eof) {
    const double t0 = now_s();
    *n = have;
    *eof = false;
    while (*n < want) {
        int timeout = -1;                                                        // no whole buffer yet: wait for as long as it takes
        if (*n >= MODES_DATA_LEN) {
            const double left = t0 + flush_s - now_s();
            if (left <= 0) break;
            timeout = (int)(left * 1e3) + 1;
        }
        struct pollfd pf{fd, POLLIN, 0};
        const int pr = poll(&pf, 1, timeout);
        if (pr < 0) { if (errno == EINTR) continue; return false; }
        if (pr == 0) break;                                                      // the deadline, with whole buffers in hand
        const ssize_t r = read(fd, dst + *n, want - *n);
        if (r < 0) { if (errno == EINTR || errno == EAGAIN) continue; return false; }
        if (r == 0) { *eof = true; break; }
        *n += (size_t)r;
    }
    return true;
}

struct Lane {
    modes_gpu *gpu = nullptr;
    uint8_t *buf = nullptr;
    int device = 0;
    std::atomic<int> ready{0};             // 0: being set up (another thread), 1: usable, -1: set-up failed (`error` says why)
    std::string error;
};

inline bool write_all(int fd, const void *p, size_t n) {
    const char *c = static_cast<const char *>(p);
    while (n) { ssize_t w = write(fd, c, n); if (w < 0) { if (errno == EINTR) continue; return false; } c += w; n -= (size_t)w; }
    return true;
}
inline bool read_all(int fd, void *p, size_t n) {
    char *c = static_cast<char *>(p);
    while (n) { ssize_t r = read(fd, c, n); if (r < 0) { if (errno == EINTR) continue; return false; } if (r == 0) return false; c += r; n -= (size_t)r; }
    return true;
}

// the hosts (host_single.cpp, host_ranks.cpp)
int run_single(Options &opt, double t_start);
int run_ranks(const Options &opt, double t_start);

}  // namespace modes_cli
#endif
