// dump1090_amd - the C/C++ host of the reference's --ifile path on top of the two libraries:
//
//   read (--ifile <file>|-)  ->  libmodes_gfx950.so (scan + demod + order on the GPU(s), many buffers per call)
//                            ->  libmodes_host.so   (in-order resolve, decodeModesMessage, sink)
//                            ->  stdout (--raw / --onlyaddr / --stats; --sbs / --raw-net: the lines of the reference's TCP sinks)
//
// It keeps the reference's spellings and defaults for the flags of this path
// (dump1090.c:2869-2897,2922; defaults :299-319) and processes EVERY buffer the reference's
// reader publishes (the reference itself drops the last one most of the time: SURVEY.md 3.4).
// Live radio, networking, interactive mode and the debug dumps are out of scope.
//
// Structure (the reference's reader thread / main thread pair, dump1090.c:460-527 and 2965-2990, widened):
//
//   reader (main thread + a pool of pread workers)          resolver thread
//   batch b -> lane b mod L: pinned buffer, GPU context  -> fetch(lane) in batch order, modes_host_resolve,
//   on device b mod N; submit = async H2D + kernels         print; the lane is free again
//
// A batch is a contiguous range of 256 KiB buffers plus the 476-byte carry in front (dump1090.c:481), so
// batches - and therefore GPUs (--gpus N: batch b runs on device b mod N, one context per lane) - share nothing
// but those 476 input bytes; the record lists come back per batch, already in stream order, and are resolved
// strictly in batch order by the one thread that owns the ICAO whitelist.
//
// --ranks N is the other way to use N GPUs (north_star's): ONE PROCESS PER GPU - this program forks N - 1 copies of
// itself - and the record lists travel to rank 0 from device memory over RCCL / xGMI (libmodes_gather.so, loaded on demand):
// rank r demodulates batches r, r + N, r + 2N, ... of the stream, round g of the gather carries batches
// gN .. gN + N - 1, so rank order is stream order and rank 0 resolves and prints every round as it arrives (run_ranks).  A regular
// file is mapped by every rank; a pipe or --loop is read by rank 0 alone, which deals the batches out through shared memory.

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

namespace {

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

void show_help() {
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
void on_message(const struct modesMessage *mm, uint32_t, uint32_t, void *user) {
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
bool read_parallel(Pool &pool, int fd, const uint8_t *map, size_t map_len, off_t *pos, uint8_t *dst, size_t want, size_t *got) {
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

bool read_full(int fd, uint8_t *dst, size_t want, size_t *got) {
    *got = 0;
    while (*got < want) {
        ssize_t n = read(fd, dst + *got, want - *got);
        if (n < 0) { if (errno == EINTR) continue; return false; }
        if (n == 0) break;
        *got += (size_t)n;
    }
    return true;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Input that cannot seek (a pipe, a FIFO, a socket): the reference prints a buffer's messages 65 ms after its samples arrived
// (dump1090.c:460-512 hands over ONE buffer; :2969-2990 decodes it at once), and a host that waits for 128 MiB before its first
// GPU call would sit on a 2 Msps stream for 33 s.  So a batch read from such an input ends when it is full, when the stream
// ends, or - at least one whole buffer being there - flush_s after the read began; until a whole buffer is there it waits
// without a deadline.  dst[0 .. have) already holds bytes (what the previous batch read beyond its last whole buffer).
// *n = bytes at dst afterwards, *eof = the stream has ended.
bool read_paced(int fd, uint8_t *dst, size_t have, size_t want, double flush_s, size_t *n, bool *eof) {
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


// ---------------------------------------------------------------------------------------------------------------------
// --ranks N: one process per GPU, the record lists gathered to rank 0 over RCCL (include/modes_gather.h).
// ---------------------------------------------------------------------------------------------------------------------
struct GatherApi {
    void *dl = nullptr;
    decltype(&modes_gather_unique_id) unique_id = nullptr;
    decltype(&modes_gather_create) create = nullptr;
    decltype(&modes_gather_destroy) destroy = nullptr;
    decltype(&modes_gather_last_error) last_error = nullptr;
    decltype(&modes_gather_output) output = nullptr;
    decltype(&modes_gather_set_empty) set_empty = nullptr;
    decltype(&modes_gather_counts) counts = nullptr;
    decltype(&modes_gather_records) records = nullptr;
    decltype(&modes_gather_wait) wait = nullptr;
    decltype(&modes_gather_get_stats) get_stats = nullptr;
    decltype(&modes_gather_set_candidates) set_candidates = nullptr;
    decltype(&modes_gather_candidates) candidates = nullptr;
    // libmodes_gather.so sits next to libmodes_gfx950.so; it is loaded only here because it pulls in librccl (0.5 GB)
    bool load() {
        Dl_info info;
        std::string dir = ".";
        if (dladdr(reinterpret_cast<void *>(&modes_gpu_create), &info) && info.dli_fname) {
            dir = info.dli_fname;
            const size_t slash = dir.rfind('/');
            dir = slash == std::string::npos ? "." : dir.substr(0, slash);
        }
        dl = dlopen((dir + "/libmodes_gather.so").c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!dl) { fprintf(stderr, "--ranks: %s\n", dlerror()); return false; }
#define SYM(name) if (!(name = reinterpret_cast<decltype(name)>(dlsym(dl, "modes_gather_" #name)))) { fprintf(stderr, "--ranks: modes_gather_" #name " missing\n"); return false; }
        SYM(unique_id) SYM(create) SYM(destroy) SYM(last_error) SYM(output) SYM(set_empty) SYM(counts) SYM(records) SYM(wait) SYM(get_stats) SYM(set_candidates) SYM(candidates)
#undef SYM
        return true;
    }
};

bool write_all(int fd, const void *p, size_t n) {
    const char *c = static_cast<const char *>(p);
    while (n) { ssize_t w = write(fd, c, n); if (w < 0) { if (errno == EINTR) continue; return false; } c += w; n -= (size_t)w; }
    return true;
}
bool read_all(int fd, void *p, size_t n) {
    char *c = static_cast<char *>(p);
    while (n) { ssize_t r = read(fd, c, n); if (r < 0) { if (errno == EINTR) continue; return false; } if (r == 0) return false; c += r; n -= (size_t)r; }
    return true;
}

int run_ranks(const Options &opt, double t_start) {
    const int N = opt.ranks;
    // A regular file is mapped by every rank, which takes its own batches.  A pipe (--ifile -) or an endless replay (--loop) has ONE reader:
    // rank 0 reads it on a thread of its own into slots of a shared mapping made before the fork (one slot per rank and batch in flight:
    // batch b belongs to rank b mod N), and rank r copies its batch from its slot to its pinned buffer exactly as it would copy it out of a
    // file mapping - one read() more per byte than the file path, in front of N PCIe links (round 5; before: refused, --gpus N named).
    const bool feed = opt.loop || opt.filename == "-";
    const int depth = std::max(3, opt.depth);        // three stages are in flight per rank (round q submits, q - 1 exchanges, q - 2 is resolved)
    const size_t batch_bytes = (size_t)opt.batch_blocks * MODES_DATA_LEN;
    // seq: 0 = free, b + 1 = holds batch b (carry + nbytes new bytes, buffers first_block ..; eof: the stream ends here and the batch carries
    // the EOF buffer).  A file's batches all have batch_blocks buffers; a pipe's have what had arrived when they were cut (read_paced).
    struct FeedSlot { std::atomic<uint64_t> seq; uint64_t nbytes, first_block; int eof; };
    struct FeedHead { std::atomic<uint64_t> total; std::atomic<int> failed; };  // total: batches of the stream, ~0 until the reader has seen the end
    const size_t slot_bytes = (MODES_CARRY_BYTES + batch_bytes + 4095) & ~(size_t)4095;
    const size_t nslots = (size_t)N * (size_t)depth;
    uint8_t *feed_mem = nullptr;
    FeedHead *feed_head = nullptr;
    FeedSlot *feed_slots = nullptr;
    if (feed) {
        const size_t ctl = (sizeof(FeedHead) + nslots * sizeof(FeedSlot) + 4095) & ~(size_t)4095;
        void *m = mmap(nullptr, ctl + nslots * slot_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) { perror("--ranks: shared buffers"); return 1; }
        feed_head = new (m) FeedHead;
        feed_head->total.store(~0ull);
        feed_head->failed.store(0);
        feed_slots = reinterpret_cast<FeedSlot *>(static_cast<uint8_t *>(m) + sizeof(FeedHead));
        for (size_t i = 0; i < nslots; i++) { new (&feed_slots[i]) FeedSlot; feed_slots[i].seq.store(0); feed_slots[i].nbytes = 0; feed_slots[i].first_block = 0; feed_slots[i].eof = 0; }
        feed_mem = static_cast<uint8_t *>(m) + ctl;
    }
    // --resolve-on-ranks (include/modes_host.h "resolve on the ranks that demodulated"; dump1090_amd/distributed.py has the same protocol over
    // torch.distributed): every rank resolves its own batch of a round from a guessed whitelist; what the ranks tell each other - guesses,
    // what they wrote, their texts - lies in a mapping made before the fork, a sequence number per (round slot, rank) and table says when
    // it is there.  The ranks of a round confirm each other IN ORDER: rank r waits for rank r - 1 to be final, so the state it rebuilds from
    // the tables of the ranks before it is the true one; it checks its logged answers against it (and resolves again if one is wrong) and
    // is final itself.  Rank 0 prints the texts of a round in rank order.  No record leaves its rank, no communicator exists.
    const bool rr = opt.resolve_on_ranks;
    struct RrRank {                                                           // one per (round slot, rank)
        std::atomic<uint64_t> guess_seq, final_seq;                           // round + 1 once `guess` / everything else is published
        uint64_t lines, nbytes;
        uint32_t guess[MODES_ICAO_SLOTS];
        uint32_t w_addr[MODES_ICAO_SLOTS];
        int64_t w_seen[MODES_ICAO_SLOTS];
        uint8_t written[MODES_ICAO_SLOTS];
    };
    struct RrHead { std::atomic<uint64_t> printed; std::atomic<int> failed; std::atomic<uint64_t> reruns; int64_t now[16]; };   // printed: rounds rank 0 has written out
    struct RrTotals { std::atomic<uint64_t> ready; modes_host_stats st; };    // --stats: a rank's nine counters when its last round is final
    RrHead *rr_head = nullptr;
    RrRank *rr_ranks = nullptr;
    RrTotals *rr_totals = nullptr;
    char *rr_text = nullptr;
    const size_t rr_text_cap = ((size_t)opt.gather_cap * 62 + 64 + 4095) & ~(size_t)4095;     // two 31-byte lines per record at most
    if (rr) {
        if (depth > 16) { fprintf(stderr, "--resolve-on-ranks: --depth %d (at most 16)\n", depth); return 1; }
        const size_t ctl = (sizeof(RrHead) + (size_t)depth * (size_t)N * sizeof(RrRank) + (size_t)N * sizeof(RrTotals) + 4095) & ~(size_t)4095;
        void *m = mmap(nullptr, ctl + (size_t)depth * (size_t)N * rr_text_cap, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) { perror("--resolve-on-ranks: shared buffers"); return 1; }
        rr_head = new (m) RrHead;
        rr_head->printed.store(0); rr_head->failed.store(0); rr_head->reruns.store(0);
        rr_ranks = reinterpret_cast<RrRank *>(static_cast<uint8_t *>(m) + sizeof(RrHead));
        for (size_t i = 0; i < (size_t)depth * (size_t)N; i++) { new (&rr_ranks[i]) RrRank; rr_ranks[i].guess_seq.store(0); rr_ranks[i].final_seq.store(0); }
        rr_totals = reinterpret_cast<RrTotals *>(rr_ranks + (size_t)depth * (size_t)N);
        for (int r = 0; r < N; r++) { new (&rr_totals[r]) RrTotals; rr_totals[r].ready.store(0); }
        rr_text = static_cast<char *>(m) + ctl;
    }
    if (!opt.devices.empty() && (int)opt.devices.size() != N) { fprintf(stderr, "--ranks %d with a --gpu-list of %zu devices\n", N, opt.devices.size()); return 1; }
    // this pool's host driver only supports dmabuf IPC: without this RCCL's cross-process buffers fail (hipIpcGetMemHandle:
    // invalid argument).  Kept if the caller has set it.
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    // the unique id travels from rank 0 to rank r through a pipe made before the fork; no HIP / RCCL call precedes the fork
    std::vector<int> rd((size_t)N, -1), wr((size_t)N, -1);
    for (int r = 1; r < N && !rr; r++) {
        int fds[2];
        if (pipe(fds) != 0) { perror("pipe"); return 1; }
        rd[(size_t)r] = fds[0];
        wr[(size_t)r] = fds[1];
    }
    int rank = 0;
    std::vector<pid_t> kids;
    const pid_t parent = getpid();
    for (int r = 1; r < N; r++) {
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); for (pid_t k : kids) kill(k, SIGKILL); return 1; }
        if (pid == 0) {
            rank = r; kids.clear();
            // a rank never outlives rank 0 (which may die inside a collective the others would wait in for ever)
            prctl(PR_SET_PDEATHSIG, SIGKILL);
            if (getppid() != parent) _exit(1);
            break;
        }
        kids.push_back(pid);
    }
    // Rank 0 watches the others: a rank that exits with an error (or is killed) leaves its peers inside an RCCL call that
    // never returns, so the job ends there and then - the other ranks are killed, the status is 1.  Clean exits are recorded
    // for finish().
    std::vector<int> kid_status(kids.size(), -1);                           // -1: running; else the wait status
    std::mutex kid_mu;
    std::atomic<bool> watch_stop{false};
    std::thread watchdog;
    // The communicator's first transfer between two processes' devices is where a wrong IPC mode shows (this pool's hosts only do
    // dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0 - a guess made on one-GPU boxes).  HSA reads the variable when the runtime starts,
    // so the other value needs new processes: rank 0 ends its peers and runs the same command line once more with it - a wrong
    // guess then costs seconds, not the job.  Nothing has been printed by then.  A peer whose probe fails exits with kProbeStatus.
    constexpr int kProbeStatus = 75;
    int saved_stdout = -1;                                                   // the real stdout once fd 1 has been given to the libraries
    auto restart_with_other_ipc_mode = [&]() {                               // rank 0 only; returns only when there is no second try
        if (N < 2 || getenv("MODES_IPC_RETRIED") || !opt.argv) return;
        const char *cur = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
        const char *other = (cur && !strcmp(cur, "0")) ? "1" : "0";
        fprintf(stderr, "--ranks: the first transfer over the new communicator failed with HSA_ENABLE_IPC_MODE_LEGACY=%s; starting over with %s\n",
                cur ? cur : "unset", other);
        fflush(stderr);
        for (size_t i = 0; i < kids.size(); i++) if (kid_status[i] == -1) kill(kids[i], SIGKILL);
        for (size_t i = 0; i < kids.size(); i++) if (kid_status[i] == -1) { int st; waitpid(kids[i], &st, 0); }
        setenv("HSA_ENABLE_IPC_MODE_LEGACY", other, 1);
        setenv("MODES_IPC_RETRIED", "1", 1);
        if (saved_stdout >= 0) dup2(saved_stdout, 1);
        execv("/proc/self/exe", opt.argv);
        perror("--ranks: execv");
    };
    if (rank == 0 && !kids.empty())
        watchdog = std::thread([&] {
            while (!watch_stop.load()) {
                {
                    std::lock_guard<std::mutex> lk(kid_mu);
                    for (size_t i = 0; i < kids.size(); i++) {
                        int st = 0;
                        if (kid_status[i] != -1 || waitpid(kids[i], &st, WNOHANG) != kids[i]) continue;
                        kid_status[i] = st;
                        if (WIFEXITED(st) && WEXITSTATUS(st) == 0) continue;
                        if (WIFEXITED(st) && WEXITSTATUS(st) == kProbeStatus) restart_with_other_ipc_mode();
                        fprintf(stderr, "--ranks: rank %zu ended with status %d%s; stopping the other ranks\n", i + 1,
                                WIFEXITED(st) ? WEXITSTATUS(st) : WTERMSIG(st), WIFEXITED(st) ? "" : " (signal)");
                        for (size_t j = 0; j < kids.size(); j++) if (kid_status[j] == -1) kill(kids[j], SIGKILL);
                        fflush(stderr);
                        _exit(1);
                    }
                }
                usleep(50 * 1000);
            }
        });
    for (int r = 1; r < N && !rr; r++) {                                     // keep only this rank's end(s)
        if (rank == 0) close(rd[(size_t)r]);
        else { close(wr[(size_t)r]); if (r != rank) close(rd[(size_t)r]); }
    }
    auto finish = [&](int rc) {                                              // rank 0: the job's status is the worst rank's
        if (rank != 0) { fflush(stdout); fflush(stderr); _exit(rc); }
        watch_stop.store(true);
        if (watchdog.joinable()) watchdog.join();
        for (size_t i = 0; i < kids.size(); i++) {
            int st = kid_status[i];
            if (st == -1) {
                if (rc) kill(kids[i], SIGKILL);                              // rank 0 failed: its peers may be waiting for it
                if (waitpid(kids[i], &st, 0) < 0) st = 1;
            }
            if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 1;
        }
        return rc;
    };
    // RCCL prints a version banner on stdout when a communicator is made: stdout is the message sink of this program, so
    // the library side of the process gets stderr as its stdout and the sink keeps the real one
    fflush(stdout);
    saved_stdout = dup(1);
    FILE *out = saved_stdout >= 0 ? fdopen(saved_stdout, "w") : nullptr;
    if (!out || dup2(2, 1) < 0) { perror("--ranks: stdout"); return finish(1); }
    GatherApi G;
    if (!rr && !G.load()) return finish(1);
    const double t_loaded = now_s();
    unsigned char id[MODES_GATHER_ID_BYTES];
    if (rr) {
    } else if (rank == 0) {
        if (G.unique_id(id) != MODES_OK) { fprintf(stderr, "--ranks: %s\n", G.last_error(nullptr)); return finish(1); }
        for (int r = 1; r < N; r++) { if (!write_all(wr[(size_t)r], id, sizeof id)) { perror("--ranks: id pipe"); return finish(1); } close(wr[(size_t)r]); }
    } else {
        if (!read_all(rd[(size_t)rank], id, sizeof id)) { fprintf(stderr, "--ranks: rank %d got no id from rank 0\n", rank); return finish(1); }
        close(rd[(size_t)rank]);
    }
    const int device = opt.devices.empty() ? rank : opt.devices[(size_t)rank];
    int fd = -1;
    size_t size = 0;
    const uint8_t *map = nullptr;
    if (!feed) {
        fd = open(opt.filename.c_str(), O_RDONLY);
        struct stat sb;
        if (fd == -1 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { perror("Opening data file"); return finish(1); }
        size = (size_t)sb.st_size;
        // One process per GPU pays RCCL's start-up - 1.6-1.8 s warm, a minute or more on a fresh box (profiles/r06/rccl_init_time.txt) -
        // before the first byte; one process driving all the devices (--gpus N) does not, and reads a file at ~50 GB/s
        // (profiles/r06/e2e_cli.json).  N readers at ~40 GB/s each win that time back only beyond kRanksPaysFromBytes (INTEGRATION.md 2b has
        // the arithmetic): say so once.
        constexpr double kRanksPaysFromBytes = 96e9;
        if (rank == 0 && !rr && (double)size < kRanksPaysFromBytes && !getenv("MODES_RANKS_QUIET"))          // (--resolve-on-ranks makes no communicator)
            fprintf(stderr, "--ranks %d: %.1f GiB is below the ~%.0f GB from which one process per GPU is faster than --gpus %d (one process, the "
                            "same devices, no communicator to start)\n", N, size / 1073741824.0, kRanksPaysFromBytes / 1e9, N);
        map = size ? static_cast<const uint8_t *>(mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0)) : nullptr;
        if (size && map == MAP_FAILED) { perror("mmap"); return finish(1); }
    } else if (rank == 0) {
        fd = opt.filename == "-" ? 0 : open(opt.filename.c_str(), O_RDONLY);
        if (fd == -1) { perror("Opening data file"); return finish(1); }
    }

    // --stats: the preamble positions of every batch travel to rank 0 with its records (the second list of the gather)
    const uint64_t batch_positions = opt.batch_blocks * (uint64_t)MODES_BLOCK_STRIDE;
    const uint32_t cap_cands = !opt.stats ? 0u : opt.gather_cands ? opt.gather_cands : (uint32_t)std::max<uint64_t>(4096, batch_positions / 64);
    modes_gather_config gc{device, rank, N, opt.gather_cap, (uint32_t)depth, cap_cands};
    modes_gather *g = nullptr;
    const double t_id = now_s();
    if (rr) {
    } else if (const int crc = G.create(&gc, id, &g); crc != MODES_OK) {
        fprintf(stderr, "--ranks: rank %d: %s\n", rank, G.last_error(nullptr));
        if (crc == MODES_GATHER_ERR_PROBE) {
            if (rank != 0) { fflush(stderr); _exit(kProbeStatus); }              // rank 0's watchdog takes it from here
            {
                std::lock_guard<std::mutex> lk(kid_mu);                          // (not while the watchdog is reaping)
                restart_with_other_ipc_mode();
            }
        }
        return finish(1);
    }
    const double t_comm = now_s();
    std::vector<Lane> lanes((size_t)depth);
    for (int l = 0; l < depth; l++) {
        modes_gpu_config cfg{};
        cfg.device = device;
        cfg.fix_errors = opt.fix_errors;
        cfg.aggressive = opt.aggressive ? 1 : 0;
        cfg.keep_candidates = opt.stats ? 1 : 0;
        void *d_rec = nullptr, *d_cnt = nullptr, *p = nullptr;
        uint64_t cap = 0;
        if (modes_gpu_create(&cfg, &lanes[(size_t)l].gpu) != MODES_OK) { fprintf(stderr, "rank %d: GPU init failed: %s\n", rank, modes_gpu_last_error(nullptr)); return finish(1); }
        modes_gpu_set_timing(lanes[(size_t)l].gpu, 0);
        if (rr) {                                                            // the list stays on this rank: the context's own pinned list (modes_gpu_fetch)
            if (modes_gpu_host_alloc(lanes[(size_t)l].gpu, MODES_CARRY_BYTES + batch_bytes, &p) != MODES_OK) {
                fprintf(stderr, "rank %d: %s\n", rank, modes_gpu_last_error(lanes[(size_t)l].gpu));
                return finish(1);
            }
        } else if (G.output(g, (uint32_t)l, &d_rec, &cap, &d_cnt) != MODES_OK || modes_gpu_set_output(lanes[(size_t)l].gpu, d_rec, cap, d_cnt) != MODES_OK ||
            modes_gpu_host_alloc(lanes[(size_t)l].gpu, MODES_CARRY_BYTES + batch_bytes, &p) != MODES_OK) {
            fprintf(stderr, "rank %d: %s / %s\n", rank, G.last_error(g), modes_gpu_last_error(lanes[(size_t)l].gpu));
            return finish(1);
        }
        lanes[(size_t)l].buf = static_cast<uint8_t *>(p);
    }
    modes_host_config hcfg{opt.fix_errors, opt.aggressive ? 1 : 0, opt.check_crc, 0};
    modes_host *host = (rank == 0 || rr) ? modes_host_create(&hcfg) : nullptr;
    modes_host *probe = rr ? modes_host_create(&hcfg) : nullptr;              // (--resolve-on-ranks: checks logged answers against a rebuilt state)
    Sink sink{&opt, host, {}, (rank == 0 && opt.sbs) ? modes_tracker_create() : nullptr};
    const bool raw_fast = opt.raw && !opt.stats && !opt.sbs && !opt.raw_net && !opt.onlyaddr;
    uint64_t n_messages_out = 0;
    const double t_ready = now_s();

    // Batch b of the stream (the single-process host's geometry: a short - possibly empty - batch ends the stream and
    // carries the EOF buffer, dump1090.c:484-510); round q of the gather = batches qN .. qN + N - 1, rank r takes batch qN + r.
    // A file's batches are known from its size; a fed stream's (pipe, --loop) when the reader sees the end (never with --loop).
    uint64_t nbatches = feed ? ~0ull : size / batch_bytes + 1;
    auto rounds_of = [&](uint64_t nb) { return nb == ~0ull ? ~0ull : (nb + (uint64_t)N - 1) / (uint64_t)N; };
    uint64_t nrounds = rounds_of(nbatches);
    uint64_t fed_bytes = 0;                                                  // rank 0's reader: bytes of the stream so far (--timing)
    std::thread reader;
    if (feed && rank == 0)
        reader = std::thread([&] {
            // dump1090.c:460-512 for N consumers: batch b = the previous batch's last 476 bytes + the next batch_bytes of the stream;
            // --loop seeks back and keeps filling the same batch (:488-494); a short batch ends the stream
            std::vector<uint8_t> tail(MODES_CARRY_BYTES, 127), pend;
            // input that cannot seek is served at the pace it delivers (read_paced): a batch is what had arrived - whole buffers - when it
            // was full or --flush-ms after it began; the bytes read beyond the last whole buffer open the next batch
            const bool paced = lseek(fd, 0, SEEK_CUR) == (off_t)-1;
#ifdef F_SETPIPE_SZ
            if (paced) (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);
#endif
            uint64_t first_block = 0;
            for (uint64_t b = 0;; b++) {
                FeedSlot &sl = feed_slots[b % nslots];
                for (int spin = 0; sl.seq.load(std::memory_order_acquire) != 0; spin++) {
                    if (feed_head->failed.load()) return;
                    usleep(spin < 100 ? 50 : 1000);
                }
                uint8_t *dst = feed_mem + (b % nslots) * slot_bytes;
                if (b) memcpy(dst, tail.data(), MODES_CARRY_BYTES);
                uint8_t *data = dst + (b ? MODES_CARRY_BYTES : 0);
                size_t got = 0;
                bool ended = false;
                if (paced) {
                    if (!pend.empty()) memcpy(data, pend.data(), pend.size());
                    size_t n = 0;
                    if (!read_paced(fd, data, pend.size(), batch_bytes, opt.flush_ms * 1e-3, &n, &ended)) { perror("read"); feed_head->failed.store(1); return; }
                    pend.clear();
                    got = n;
                    if (!ended && n < batch_bytes) { got = n - n % MODES_DATA_LEN; pend.assign(data + got, data + n); }
                } else {
                    if (!read_full(fd, data, batch_bytes, &got)) { perror("read"); feed_head->failed.store(1); return; }
                    while (got < batch_bytes && opt.loop && fd != 0) {
                        if (lseek(fd, 0, SEEK_SET) == -1) break;
                        size_t more = 0;
                        if (!read_full(fd, data + got, batch_bytes - got, &more)) { perror("read"); feed_head->failed.store(1); return; }
                        if (more == 0) break;                                    // empty file
                        got += more;
                    }
                    ended = got < batch_bytes;
                }
                if (got >= MODES_CARRY_BYTES) memcpy(tail.data(), data + got - MODES_CARRY_BYTES, MODES_CARRY_BYTES);
                fed_bytes += got;
                sl.nbytes = got;
                sl.first_block = first_block;
                sl.eof = ended ? 1 : 0;
                first_block += got / MODES_DATA_LEN;
                if (ended) feed_head->total.store(b + 1, std::memory_order_release);                  // this batch carries the EOF buffer
                sl.seq.store(b + 1, std::memory_order_release);
                if (ended) return;
            }
        });
    Pool pool(std::max(1, opt.read_threads / N));
    std::vector<char> has((size_t)depth, 0);
    int rc = 0;
    auto fail_rank = [&](const char *what, const char *text) { fprintf(stderr, "rank %d: %s: %s\n", rank, what, text); rc = 1; };
    // ---- --resolve-on-ranks: round qq of this rank (its batch's kernels are queued; the next batch's already run) ----
    std::vector<uint32_t> truth_addr(MODES_ICAO_SLOTS, 0), st_addr(MODES_ICAO_SLOTS);       // the whitelist every round < qq left (the same on every rank)
    std::vector<int64_t> truth_seen(MODES_ICAO_SLOTS, 0), st_seen(MODES_ICAO_SLOTS);
    std::vector<modes_icao_lookup> lookups;
    uint64_t rr_applied = 0;                                                 // rounds whose writes are in `truth`
    auto rr_at = [&](uint64_t round, int r) -> RrRank & { return rr_ranks[(size_t)(round % (uint64_t)depth) * (size_t)N + (size_t)r]; };
    auto rr_wait = [&](const std::atomic<uint64_t> &a, uint64_t v) {         // false: another rank failed (or this one's peers are gone)
        for (int spin = 0; a.load(std::memory_order_acquire) < v;) {
            if (rr_head->failed.load()) return false;
            if (spin < 100000) spin++;                                           // (saturates: a live pipe can keep a rank here for minutes)
            if (spin > 200) usleep(spin < 2000 ? 20 : 1000);                     // short waits spin, long ones - a batch of a live stream - sleep
        }
        return true;
    };
    auto rr_apply = [&](std::vector<uint32_t> &addr, std::vector<int64_t> &seen, const RrRank &w) {
        for (uint32_t sidx = 0; sidx < MODES_ICAO_SLOTS; sidx++)
            if (w.written[sidx]) { addr[sidx] = w.w_addr[sidx]; seen[sidx] = w.w_seen[sidx]; }
    };
    auto rr_round = [&](uint64_t qq, bool have) -> bool {
        const int l = (int)(qq % (uint64_t)depth);
        RrRank &me = rr_at(qq, rank);
        // the slot's previous tenant: round qq - depth, printed?
        if (qq >= (uint64_t)depth && !rr_wait(rr_head->printed, qq - (uint64_t)depth + 1)) return false;
        const modes_record *recs = nullptr;
        uint64_t nrec = 0;
        const uint64_t *cands = nullptr;                                     // --stats: every preamble position of the batch (dump1090.c:1651)
        uint64_t ncand = 0;
        if (have) {
            modes_gpu_result res{};
            if (modes_gpu_fetch(lanes[(size_t)l].gpu, &res) != MODES_OK) { fail_rank("GPU demodulation failed", modes_gpu_last_error(lanes[(size_t)l].gpu)); return false; }
            recs = res.records;
            nrec = res.n_records;
            cands = res.candidates;
            ncand = res.n_candidates;
            if (nrec > opt.gather_cap) { fail_rank("resolve", "a batch's records exceed --gather-records (the text buffers are sized by it)"); return false; }
        }
        modes_host_whitelist_guess(host, &recs, &nrec, 1, me.guess, opt.resolve_threads);
        if (rank == 0) rr_head->now[l] = feed ? (int64_t)time(nullptr) : 0;   // one clock per round: rank 0's (a live stream: dump1090.c:913,924)
        me.guess_seq.store(qq + 1, std::memory_order_release);
#ifdef MODES_TEST_HOOKS                                                          // (stub builds: a rank that dies between its guess and its final tables)
        if (const char *die = getenv("MODES_RR_DIE"); die && atoi(die) == rank && strchr(die, ':') && (uint64_t)atoll(strchr(die, ':') + 1) == qq) raise(SIGKILL);
#endif
        // the state the round starts from: every earlier round, final on every rank
        for (; rr_applied < qq; rr_applied++)
            for (int r = 0; r < N; r++) {
                if (!rr_wait(rr_at(rr_applied, r).final_seq, rr_applied + 1)) return false;
                rr_apply(truth_addr, truth_seen, rr_at(rr_applied, r));
            }
        st_addr = truth_addr;
        st_seen = truth_seen;
        if (!rr_wait(rr_at(qq, 0).guess_seq, qq + 1)) return false;
        const int64_t now = rr_head->now[l];
        for (int r = 0; r < rank; r++) {                                     // ... overlaid with what the ranks before this one expect to write
            const RrRank &o = rr_at(qq, r);
            if (!rr_wait(o.guess_seq, qq + 1)) return false;
            for (uint32_t sidx = 0; sidx < MODES_ICAO_SLOTS; sidx++)
                if (o.guess[sidx] != MODES_ICAO_NONE) { st_addr[sidx] = o.guess[sidx]; st_seen[sidx] = now; }
        }
#ifdef MODES_TEST_HOOKS                                                          // (the stub builds of tools/sanitize_host.sh: a wrong start on purpose)
        if (const char *sp = getenv("MODES_RR_SPOIL"); sp && *sp && rank > 0) { std::fill(st_addr.begin(), st_addr.end(), 0u); std::fill(st_seen.begin(), st_seen.end(), (int64_t)0); }
#endif
        char *text = rr_text + ((size_t)l * (size_t)N + (size_t)rank) * rr_text_cap;
        lookups.resize((size_t)nrec * 2 + 16);
        uint64_t nb = 0, nl = 0, lines = 0;
        // --raw: the lean resolve on several threads.  The other sinks this mode serves go through the general resolve and the host's
        // own sink: --onlyaddr / --raw-net format their line there, --stats prints nothing and counts (with the batch's preamble
        // positions: the counters of dump1090.c:2993-3006 are sums of per-batch counts, rank 0 adds the ranks' up at the end).
        // A resolve that is repeated starts from the counters the first one found.
        modes_host_stats st_before;
        modes_host_get_stats(host, &st_before);
        auto resolve_from = [&](const std::vector<uint32_t> &addr, const std::vector<int64_t> &seen) {
            modes_host_set_time(host, now);
            modes_host_set_whitelist(host, addr.data(), seen.data());
            if (raw_fast) {
                lines = modes_host_resolve_raw_spec(host, &recs, &nrec, 1, text, rr_text_cap, &nb, opt.resolve_threads, me.written, lookups.data(), lookups.size(), &nl);
                return;
            }
            modes_host_set_stats(host, &st_before);
            sink.out.clear();
            lines = modes_host_resolve_spec(host, recs, nrec, cands, ncand, on_message, &sink, me.written, lookups.data(), lookups.size(), &nl);
            nb = sink.out.size();
            if (nb < rr_text_cap) memcpy(text, sink.out.data(), (size_t)nb);
            sink.out.clear();
        };
        resolve_from(st_addr, st_seen);
        // confirmation, in rank order: the ranks before this one are final -> their tables give the true start
        if (rank > 0) {
            if (!rr_wait(rr_at(qq, rank - 1).final_seq, qq + 1)) return false;
            st_addr = truth_addr;
            st_seen = truth_seen;
            for (int r = 0; r < rank; r++) rr_apply(st_addr, st_seen, rr_at(qq, r));
            modes_host_set_time(probe, now);
            modes_host_set_whitelist(probe, st_addr.data(), st_seen.data());
            if (nl > lookups.size() || !modes_host_whitelist_check(probe, lookups.data(), nl)) {
                resolve_from(st_addr, st_seen);                              // rare: an answer taken from the guess was wrong
                rr_head->reruns.fetch_add(1);
            }
        }
        if (nb >= rr_text_cap) { fail_rank("resolve", "the text of a batch outgrew its buffer"); return false; }
        modes_host_get_whitelist(host, me.w_addr, me.w_seen);
        me.lines = lines;
        me.nbytes = nb;
        me.final_seq.store(qq + 1, std::memory_order_release);
        if (rank == 0) {                                                     // the round's listing, rank after rank
            for (int r = 0; r < N; r++) {
                const RrRank &o = rr_at(qq, r);
                if (!rr_wait(o.final_seq, qq + 1)) return false;
                if (o.nbytes) fwrite(rr_text + ((size_t)l * (size_t)N + (size_t)r) * rr_text_cap, 1, (size_t)o.nbytes, out);
                n_messages_out += o.lines;
            }
            fflush(out);
            // (every rank reads a round's tables when it starts the NEXT round; the slot is written again depth rounds later, by ranks that
            //  have been through the round after this one - which needs every rank final there, i.e. past its reading of these)
            rr_head->printed.store(qq + 1, std::memory_order_release);
        }
        return true;
    };
    // ---- the three stages of a round, and the order they run in ----
    // submit(q): this rank's batch of round q -> pinned buffer -> H2D + kernels.  exchange(q): lengths and lists to rank 0 (or, resolving on
    // the ranks, the whole of rr_round).  resolve(q): rank 0 resolves and prints what arrived.  With input at hand the stages run one round
    // apart - submit(q), exchange(q - 1), resolve(q - 2): the kernels of a round run under the exchange of the round before, the transfers
    // under the resolve of the round before that.  When the NEXT batch is not there yet (a pipe at the radio's pace) nothing is held back for
    // it: the rounds in flight are exchanged, resolved and printed while the rank waits (output lagged two batch times).  The order
    // of the collective calls is the same on every rank either way - round after round -, only when a rank issues them differs.
    enum class Input { Ready, NotYet, Ended, Failed };
    auto input_state = [&](uint64_t q) -> Input {                            // of round q, without waiting
        if (nrounds != ~0ull && q >= nrounds) return Input::Ended;
        if (!feed) return Input::Ready;                                      // (a rank without a batch in the last round still takes part in it)
        const uint64_t b = q * (uint64_t)N + (uint64_t)rank;
        if (feed_slots[b % nslots].seq.load(std::memory_order_acquire) == b + 1) return Input::Ready;
        const uint64_t total = feed_head->total.load(std::memory_order_acquire);
        if (total != ~0ull) {
            if (nbatches == ~0ull) { nbatches = total; nrounds = rounds_of(nbatches); }
            if (q >= nrounds) return Input::Ended;
            if (b >= total) return Input::Ready;                             // the round exists, this rank has no batch in it
        }
        return feed_head->failed.load() ? Input::Failed : Input::NotYet;
    };
    auto submit = [&](uint64_t q) {
        const int l = (int)(q % (uint64_t)depth);
        const uint64_t b = q * (uint64_t)N + (uint64_t)rank;
        const uint8_t *src = nullptr;
        size_t got = 0;
        uint64_t first_block = b * opt.batch_blocks;
        bool last = false;
        FeedSlot *slot = nullptr;
        if (feed) {
            slot = &feed_slots[b % nslots];
            if (slot->seq.load(std::memory_order_acquire) != b + 1) slot = nullptr;          // (input_state said Ready: no batch of this rank in the round)
            has[(size_t)l] = slot != nullptr;
            if (slot) { src = feed_mem + (b % nslots) * slot_bytes; got = (size_t)slot->nbytes; first_block = slot->first_block; last = slot->eof != 0; }
        } else {
            has[(size_t)l] = b < nbatches;
            if (has[(size_t)l]) {
                const size_t lo = (size_t)b * batch_bytes;
                got = std::min(batch_bytes, size - std::min(size, lo));
                src = map + lo - (b ? MODES_CARRY_BYTES : 0);
                last = got < batch_bytes;
            }
        }
        if (!has[(size_t)l]) return;
        const size_t carry = b ? MODES_CARRY_BYTES : 0;
        const size_t n = carry + got, sl = (n / (size_t)pool.size() + 4095) & ~(size_t)4095;
        uint8_t *dst = lanes[(size_t)l].buf;
        pool.run(pool.size(), [&](int t) { const size_t o = (size_t)t * sl; if (o < n) memcpy(dst + o, src + o, std::min(sl, n - o)); });
        if (slot) slot->seq.store(0, std::memory_order_release);             // the reader may fill it again
        const uint64_t nblocks = got / MODES_DATA_LEN + (last ? 1 : 0);      // (+ the EOF buffer)
        if (modes_gpu_submit_host(lanes[(size_t)l].gpu, dst, n, first_block * (uint64_t)MODES_DATA_LEN - carry, first_block, nblocks) != MODES_OK)
            fail_rank("GPU demodulation failed", modes_gpu_last_error(lanes[(size_t)l].gpu));
    };
    auto exchange = [&](uint64_t q) {                                        // kernels done -> lengths -> transfers
        const int l = (int)(q % (uint64_t)depth);
        if (rr) {
            if (!rr_round(q, has[(size_t)l] != 0) && !rc) fail_rank("resolve", "another rank failed");
            return;
        }
        if (has[(size_t)l]) {
            modes_gpu_result res{};
            // (a list that outgrew the buffers still goes through the length exchange: every rank then fails together)
            const int frc = modes_gpu_fetch_device(lanes[(size_t)l].gpu, &res);
            if (frc != MODES_OK && frc != MODES_ERR_OVERFLOW) fail_rank("GPU demodulation failed", modes_gpu_last_error(lanes[(size_t)l].gpu));
            else if (opt.stats && G.set_candidates(g, (uint32_t)l, res.candidates, res.n_candidates) != MODES_OK) fail_rank("gather", G.last_error(g));
        } else if (G.set_empty(g, (uint32_t)l) != MODES_OK) fail_rank("gather", G.last_error(g));
        if (!rc && (G.counts(g, (uint32_t)l) != MODES_OK || G.records(g, (uint32_t)l) != MODES_OK)) fail_rank("gather", G.last_error(g));
    };
    auto resolve = [&](uint64_t q) {                                         // rank 0 resolves what arrived
        if (rr) return;                                                      // (rr_round has printed the round)
        const int l = (int)(q % (uint64_t)depth);
        const modes_record *recs = nullptr;
        uint64_t nrec = 0;
        if (G.wait(g, (uint32_t)l, &recs, &nrec, nullptr) != MODES_OK) { fail_rank("gather", G.last_error(g)); return; }
        if (rank != 0) return;
        if (feed) modes_host_set_time(host, (int64_t)time(nullptr));       // a live stream: the whitelist's 60 s run on the wall clock (dump1090.c:913,924)
        const uint64_t *cands = nullptr;
        uint64_t ncand = 0;
        if (opt.stats && G.candidates(g, (uint32_t)l, &cands, &ncand) != MODES_OK) { fail_rank("gather", G.last_error(g)); return; }
        if (raw_fast) {                                                      // the listing goes out from where the resolve's threads wrote it
            modes_text_piece pieces[80];
            uint32_t np = 0;
            n_messages_out += modes_host_resolve_raw_pieces(host, &recs, &nrec, 1, pieces, 80, &np, nullptr, opt.resolve_threads);
            for (uint32_t i = 0; i < np; i++) fwrite(pieces[i].base, 1, (size_t)pieces[i].len, out);
            if (np) fflush(out);
        } else
            n_messages_out += modes_host_resolve(host, recs, nrec, cands, ncand, on_message, &sink);
        if (!sink.out.empty()) {
            fwrite(sink.out.data(), 1, sink.out.size(), out);
            fflush(out);
            sink.out.clear();
        }
    };
    uint64_t ns = 0, nx = 0, nr = 0;                                         // next round to submit / exchange / resolve
    for (int idle_spins = 0; !rc;) {
        bool did = false;
        Input in = input_state(ns);
        if (in == Input::Failed) { fail_rank("input", "the reader failed"); break; }
        if (in == Input::Ready && ns - nr < (uint64_t)depth) {               // (the lane of round ns is free once round ns - depth is resolved)
            submit(ns++);
            did = true;
            if (rc) break;
            in = input_state(ns);
        }
        // the next batch is at hand: stay one round behind it (its kernels cover this exchange); it is not: nothing waits for it
        const bool at_hand = in == Input::Ready && ns - nr < (uint64_t)depth;
        if (nx < (at_hand && ns ? ns - 1 : ns)) { exchange(nx++); did = true; if (rc) break; }
        if (nr < (at_hand && nx ? nx - 1 : nx)) { resolve(nr++); did = true; if (rc) break; }
        if (in == Input::Ended && nr == ns) break;
        if (did) { idle_spins = 0; continue; }
        if (idle_spins < 100000) idle_spins++;
        usleep(idle_spins < 200 ? 50 : 1000);                                // a live pipe: a batch interval is tens of milliseconds
    }
    const double t_end = now_s();
    if (feed && rc) feed_head->failed.store(1);                              // (the reader and the other ranks stop waiting for slots)
    if (rr && rc) rr_head->failed.store(1);
    if (reader.joinable()) { if (rc) reader.detach(); else reader.join(); }
    if (feed) size = (size_t)fed_bytes;                                      // what --timing reports (rank 0 knows it)
    if (rc) {
        // A rank that leaves the round loop with an error has peers that wait inside a collective it will never issue; they never
        // reach their own teardown, and RCCL's communicator destroy may wait for them (it synchronises the ranks of a node).  So
        // nothing is torn down on this path: a peer reports and exits at once (rank 0's watchdog then ends the job), rank 0 ends
        // the other ranks first and leaves the rest to the process exit.
        fflush(out);
        fflush(stderr);
        if (rank != 0) _exit(rc);
        _exit(finish(rc));
    }
    if (rr && opt.stats) {                                                   // every rank's counters -> rank 0, which adds them up
        modes_host_get_stats(host, &rr_totals[rank].st);
        rr_totals[rank].ready.store(1, std::memory_order_release);
        if (rank == 0) {
            modes_host_stats sum{};
            for (int r = 0; r < N; r++) {
                if (!rr_wait(rr_totals[r].ready, 1)) { fail_rank("resolve", "another rank failed"); break; }
                const modes_host_stats &o = rr_totals[r].st;
                sum.valid_preamble += o.valid_preamble > 0 ? o.valid_preamble : 0;   // (a rank that never had a batch saw no positions)
                sum.out_of_phase += o.out_of_phase; sum.demodulated += o.demodulated; sum.goodcrc += o.goodcrc; sum.badcrc += o.badcrc;
                sum.fixed += o.fixed; sum.single_bit_fix += o.single_bit_fix; sum.two_bits_fix += o.two_bits_fix;
            }
            if (rc) { fflush(out); fflush(stderr); rr_head->failed.store(1); _exit(finish(rc)); }
            char text[512];
            modes_format_stats(&sum, text);
            fputs(text, out);
            fflush(out);
        }
    } else if (rank == 0 && opt.stats) {                                     // dump1090.c:2993-3006
        modes_host_stats hs;
        modes_host_get_stats(host, &hs);
        char text[512];
        modes_format_stats(&hs, text);
        fputs(text, out);
        fflush(out);
    }
    if (rank == 0 && opt.timing && rr) {
        const double stream_s = t_end - t_ready;
        fprintf(stderr, "{\"bytes\": %zu, \"ranks\": %d, \"rounds\": %llu, \"init_s\": %.4f, \"stream_s\": %.4f, \"total_s\": %.4f, \"stream_GBps\": %.2f, "
                        "\"sink_calls\": %llu, \"resolve_on\": \"ranks\", \"reruns\": %llu}\n",
                size, N, (unsigned long long)nrounds, t_ready - t_start, stream_s, t_end - t_start, stream_s > 0 ? size / stream_s / 1e9 : 0.0,
                (unsigned long long)n_messages_out, (unsigned long long)rr_head->reruns.load());
    } else if (rank == 0 && opt.timing) {
        modes_gather_stats st{};
        G.get_stats(g, &st);
        const double stream_s = t_end - t_ready;
        fprintf(stderr,
                "{\"bytes\": %zu, \"ranks\": %d, \"rounds\": %llu, \"init_s\": %.4f, \"stream_s\": %.4f, \"total_s\": %.4f, \"stream_GBps\": %.2f, "
                "\"init\": {\"load_gather_library_s\": %.4f, \"unique_id_s\": %.4f, \"communicator_s\": %.4f, \"lanes_s\": %.4f}, "
                "\"sink_calls\": %llu, \"rccl\": {\"version\": %d, \"nranks\": %d, \"p2p_ops\": %llu, \"bytes_received\": %llu, \"gather_ms\": %.3f}}\n",
                size, N, (unsigned long long)nrounds, t_ready - t_start, stream_s, t_end - t_start, stream_s > 0 ? size / stream_s / 1e9 : 0.0,
                t_loaded - t_start, t_id - t_loaded, t_comm - t_id, t_ready - t_comm,
                (unsigned long long)n_messages_out, st.rccl_version, st.nranks, (unsigned long long)st.p2p_ops,
                (unsigned long long)st.bytes_received, st.gather_ms);
    }
    if (rr && !opt.clean_exit) {
        // Everything is printed and no communicator exists whose teardown the ranks would have to do together: like the one-process
        // host, leave the unpinning, the unmapping and the runtime's exit handlers to the kernel (a third of a short run's wall clock).
        fflush(out);
        fflush(stderr);
        if (rank != 0) _exit(0);
        _exit(finish(0));
    }
    if (host) modes_host_destroy(host);
    if (probe) modes_host_destroy(probe);
    modes_tracker_destroy(sink.tracker);
    for (auto &ln : lanes) { modes_gpu_host_free(ln.gpu, ln.buf); modes_gpu_destroy(ln.gpu); }
    if (!rr) G.destroy(g);
    if (map) munmap(const_cast<uint8_t *>(map), size);
    if (fd > 0) close(fd);
    return finish(rc);
}

}  // namespace

int main(int argc, char **argv) {
    const double t_start = now_s();
    Options opt;
    opt.argv = argv;
    int single_device = 0, ngpus = 0;
    for (int j = 1; j < argc; j++) {
        const bool more = j + 1 < argc;
        const char *a = argv[j];
        if (!strcmp(a, "--ifile") && more) opt.filename = argv[++j];
        else if (!strcmp(a, "--loop")) opt.loop = true;
        else if (!strcmp(a, "--no-fix")) opt.fix_errors = 0;
        else if (!strcmp(a, "--no-crc-check")) opt.check_crc = 0;
        else if (!strcmp(a, "--raw")) opt.raw = true;
        else if (!strcmp(a, "--onlyaddr")) opt.onlyaddr = true;
        else if (!strcmp(a, "--sbs")) opt.sbs = true;
        else if (!strcmp(a, "--raw-net")) opt.raw_net = true;
        else if (!strcmp(a, "--aggressive")) opt.aggressive++;
        else if (!strcmp(a, "--stats")) opt.stats = true;
        else if (!strcmp(a, "--timing")) opt.timing = true;
        else if (!strcmp(a, "--clean-exit")) opt.clean_exit = true;
        else if (!strcmp(a, "--gpu") && more) single_device = atoi(argv[++j]);
        else if (!strcmp(a, "--gpus") && more) ngpus = atoi(argv[++j]);
        else if (!strcmp(a, "--gpu-list") && more) {
            opt.devices.clear();
            for (const char *p = argv[++j]; *p;) {
                char *end;
                opt.devices.push_back((int)strtol(p, &end, 10));
                if (end == p) { fprintf(stderr, "--gpu-list: bad ordinal in '%s'\n", argv[j]); return 1; }
                p = *end == ',' ? end + 1 : end;
            }
        }
        else if (!strcmp(a, "--ranks") && more) opt.ranks = atoi(argv[++j]);
        else if (!strcmp(a, "--resolve-on-ranks")) opt.resolve_on_ranks = true;
        else if (!strcmp(a, "--gather-records") && more) opt.gather_cap = (uint32_t)strtoul(argv[++j], nullptr, 10);
        else if (!strcmp(a, "--gather-candidates") && more) opt.gather_cands = (uint32_t)strtoul(argv[++j], nullptr, 10);
        else if (!strcmp(a, "--batch-blocks") && more) opt.batch_blocks = strtoull(argv[++j], nullptr, 10);
        else if (!strcmp(a, "--flush-ms") && more) opt.flush_ms = std::max(0, atoi(argv[++j]));
        else if (!strcmp(a, "--depth") && more) opt.depth = std::max(1, atoi(argv[++j]));
        else if (!strcmp(a, "--read-threads") && more) { opt.read_threads = std::max(1, atoi(argv[++j])); opt.read_threads_given = true; }
        else if (!strcmp(a, "--no-mmap")) opt.use_mmap = false;
        else if (!strcmp(a, "--resolve-threads") && more) { opt.resolve_threads = std::max(1, atoi(argv[++j])); opt.resolve_threads_given = true; }
        else if (!strcmp(a, "--help")) { show_help(); return 0; }
        else {
            fprintf(stderr, "Unknown or not enough arguments for option '%s'.\n\n", a);
            show_help();
            return 1;
        }
    }
    if (opt.filename.empty()) {
        fprintf(stderr, "dump1090_amd demodulates files only: give --ifile <file> (or '-').\n");
        return 1;
    }
    if (opt.batch_blocks == 0) opt.batch_blocks = 1;
    {   // The pools' defaults are for a whole machine; a container may be held to a fraction of it (a 256-thread host behind a cgroup
        // quota of 16: modes_host_cpu_budget).  Threads beyond the budget only take CPU time from each other - and from the threads
        // that launch and poll: one per lane, plus the resolver.  Explicit --read-threads / --resolve-threads are honoured.
        const int budget = modes_host_cpu_budget(), procs = opt.ranks > 0 ? opt.ranks : 1;
        const int spare = std::max(1, budget / procs - 2);
        if (!opt.read_threads_given) opt.read_threads = std::min(opt.read_threads, std::max(1, spare * (opt.ranks > 0 ? procs : 1)));   // (--ranks divides it by N again)
        if (!opt.resolve_threads_given) opt.resolve_threads = std::min(opt.resolve_threads, spare);
    }
    if (opt.ranks > 0) {
        // a regular file shorter than a batch: buffers of its size (see below) - every rank pins depth x batch, rank 0 also
        // depth x ranks x the gather capacity
        struct stat sb;
        if (opt.filename != "-" && stat(opt.filename.c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) {
            const uint64_t in_file = (uint64_t)sb.st_size / MODES_DATA_LEN + 1;
            if (in_file < opt.batch_blocks) opt.batch_blocks = in_file;
        }
        if (ngpus > 0) { fprintf(stderr, "--ranks and --gpus are two ways to use N GPUs: give one of them\n"); return 1; }
        if (opt.resolve_on_ranks && (opt.sbs || !(opt.raw || opt.stats || opt.onlyaddr || opt.raw_net))) {
            // --raw, --onlyaddr, --raw-net: a line is a function of its message; --stats: the nine counters are sums of per-batch counts.
            // --sbs is not: a BaseStation line reads the AIRCRAFT TABLE (positions from CPR pairs, speed and track of earlier messages,
            // dump1090.c:2069-2167, :2397-2448) - state that crosses every batch in stream order, like the whitelist but 200 bytes per
            // aircraft and written by nearly every message: there is nothing to guess.  The verbose dump is 11 lines a message: rank 0's.
            fprintf(stderr, "--resolve-on-ranks serves --raw, --onlyaddr, --raw-net and --stats; --sbs needs the aircraft table in stream order and the "
                            "verbose dump is too much text to ship: leave it out for those\n");
            return 1;
        }
        return run_ranks(opt, t_start);
    }
    if (opt.resolve_on_ranks) { fprintf(stderr, "--resolve-on-ranks goes with --ranks <n>\n"); return 1; }
    if (opt.devices.empty()) {
        if (ngpus > 0) for (int d = 0; d < ngpus; d++) opt.devices.push_back(d);
        else opt.devices.push_back(single_device);
    }

    int fd = 0;
    if (opt.filename != "-" && (fd = open(opt.filename.c_str(), O_RDONLY)) == -1) {
        perror("Opening data file");
        return 1;
    }

    {   // a regular file shorter than a batch: pinned buffers of its size, not of the default 128 MiB each (they are most of
        // the start-up time of a run on a small file); the whole file is then one batch with its EOF buffer
        struct stat sb;
        if (fd != 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) {
            const uint64_t in_file = (uint64_t)sb.st_size / MODES_DATA_LEN + 1;
            if (in_file < opt.batch_blocks) opt.batch_blocks = in_file;
        }
    }
    // One lane = one GPU context + one pinned buffer; lane l lives on device l mod N, so that batch b (lane
    // b mod L, L a multiple of N) runs on device b mod N.  The contexts of different devices are created
    // concurrently (HIP initialises each device on first use).
    const int ndev = (int)opt.devices.size();
    const int nlanes = ndev * opt.depth;
    const size_t batch_bytes = (size_t)opt.batch_blocks * MODES_DATA_LEN;
    // The lanes are set up by one thread per device WHILE the stream already runs on the lanes that exist: the first context
    // pays for the start of the HIP runtime (0.15-0.25 s, nothing to overlap it with), every further lane - a context, its
    // tables, 128 MiB of pinned memory - would add ~20 ms each in front of the first read if the reader waited for all of them.
    std::vector<Lane> lanes((size_t)nlanes);
    std::vector<double> t_created((size_t)nlanes, 0.0), t_pinned((size_t)nlanes, 0.0);      // --timing: when each lane had its context / buffer
    auto make = [&](int l) {
        Lane &ln = lanes[(size_t)l];
        modes_gpu_config gcfg{};
        gcfg.device = opt.devices[(size_t)(l % ndev)];
        gcfg.fix_errors = opt.fix_errors;
        gcfg.aggressive = opt.aggressive ? 1 : 0;
        gcfg.keep_candidates = opt.stats ? 1 : 0;
        ln.device = gcfg.device;
        if (modes_gpu_create(&gcfg, &ln.gpu) != MODES_OK) {
            ln.error = std::string("GPU init failed: ") + modes_gpu_last_error(nullptr);
            ln.ready.store(-1);
            return false;
        }
        modes_gpu_set_timing(ln.gpu, 0);                              // no timing events between the kernels: batches run back to back
        t_created[(size_t)l] = now_s();
        void *p = nullptr;
        if (modes_gpu_host_alloc(ln.gpu, MODES_CARRY_BYTES + batch_bytes, &p) != MODES_OK) {
            ln.error = std::string("pinned buffer: ") + modes_gpu_last_error(ln.gpu);
            ln.ready.store(-1);
            return false;
        }
        ln.buf = static_cast<uint8_t *>(p);
        t_pinned[(size_t)l] = now_s();
        ln.ready.store(1);
        return true;
    };
    std::vector<std::thread> lane_makers;
    for (int d = 0; d < ndev; d++)
        lane_makers.emplace_back([&, d] { for (int l = d; l < nlanes; l += ndev) if (!make(l)) break; });
    auto lane_ready = [&](int l) -> bool {                                   // blocks until lane l is set up; false: it failed
        Lane &ln = lanes[(size_t)l];
        while (ln.ready.load() == 0) usleep(200);
        if (ln.ready.load() < 0) { fprintf(stderr, "%s\n", ln.error.c_str()); return false; }
        return true;
    };
    if (!lane_ready(0)) { for (auto &t : lane_makers) t.join(); return 1; }
    modes_host_config hcfg{opt.fix_errors, opt.aggressive ? 1 : 0, opt.check_crc, 0};
    modes_host *host = modes_host_create(&hcfg);
    if (!host) { fprintf(stderr, "modes_host_create failed\n"); for (auto &t : lane_makers) t.join(); return 1; }
    Sink sink{&opt, host, {}, opt.sbs ? modes_tracker_create() : nullptr};
    const bool live = opt.loop || fd == 0;               // a pipe or an endless replay: the whitelist TTL follows the wall clock
    const double t_ready = now_s();

    // ---- hand-off between the reader (this thread) and the resolver ----
    std::mutex m;
    std::condition_variable cv;
    uint64_t submitted = 0, resolved = 0;                // batches
    bool reader_done = false, failed = false;
    uint64_t n_messages_out = 0;

    const bool raw_fast = opt.raw && !opt.stats && !opt.sbs && !opt.raw_net && !opt.onlyaddr;
    std::thread resolver([&] {
        for (uint64_t b = 0;; b++) {
            {
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [&] { return submitted > b || reader_done || failed; });
                if (failed || (submitted <= b && reader_done)) return;
            }
            Lane &ln = lanes[(size_t)(b % (uint64_t)nlanes)];
            modes_gpu_result res{};
            if (modes_gpu_fetch(ln.gpu, &res) != MODES_OK) {
                fprintf(stderr, "GPU demodulation failed: %s\n", modes_gpu_last_error(ln.gpu));
                std::lock_guard<std::mutex> g(m);
                failed = true;
                cv.notify_all();
                return;
            }
            if (live) modes_host_set_time(host, (int64_t)time(nullptr));          // dump1090.c:913,924
            if (raw_fast) {                                                       // the --raw listing of a long batch, several threads;
                modes_text_piece pieces[80];                                      // it goes out from where they wrote it
                uint32_t np = 0;
                const modes_record *recs = res.records;
                const uint64_t nrec = res.n_records;
                n_messages_out += modes_host_resolve_raw_pieces(host, &recs, &nrec, 1, pieces, 80, &np, nullptr, opt.resolve_threads);
                for (uint32_t i = 0; i < np; i++) fwrite(pieces[i].base, 1, (size_t)pieces[i].len, stdout);
                if (np) fflush(stdout);
            } else
            n_messages_out += modes_host_resolve(host, res.records, res.n_records, res.candidates, res.n_candidates, on_message, &sink);
            if (!sink.out.empty()) {
                fwrite(sink.out.data(), 1, sink.out.size(), stdout);
                fflush(stdout);
                sink.out.clear();
            }
            {
                std::lock_guard<std::mutex> g(m);
                resolved = b + 1;
            }
            cv.notify_all();
        }
    });

    // Batch b covers buffers [first, first+n): host bytes = 476-byte carry + n*262144 new bytes.
    // --loop replays a file forever through the sequential path; a plain regular file is read in parallel
    const bool seekable = !opt.loop && fd != 0 && lseek(fd, 0, SEEK_CUR) != (off_t)-1;
    Pool pool(seekable ? opt.read_threads : 1);
    const uint8_t *map = nullptr;
    size_t map_len = 0;
    if (seekable && opt.use_mmap) {
        struct stat sb;
        if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            void *m2 = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_SHARED, fd, 0);
            if (m2 != MAP_FAILED) {
                map = static_cast<const uint8_t *>(m2);
                map_len = (size_t)sb.st_size;
                (void)madvise(m2, map_len, MADV_SEQUENTIAL);
            }
        }
    }
    // ranges of the mapping whose bytes have been copied: unmapped by a helper thread while the stream runs
    std::mutex unmap_m;
    std::condition_variable unmap_cv;
    std::vector<std::pair<uint8_t *, size_t>> unmap_q;
    bool unmap_stop = false;
    size_t unmapped_to = 0;                 // [0, unmapped_to) of the mapping has been handed to the unmapper (batches are consecutive)
    std::thread unmapper([&] {
        for (;;) {
            std::vector<std::pair<uint8_t *, size_t>> work;
            {
                std::unique_lock<std::mutex> g(unmap_m);
                unmap_cv.wait(g, [&] { return unmap_stop || !unmap_q.empty(); });
                if (unmap_q.empty()) return;
                work.swap(unmap_q);
            }
            for (auto &r : work) if (r.second) munmap(r.first, r.second);
        }
    });
    auto stop_unmapper = [&] {
        { std::lock_guard<std::mutex> g(unmap_m); unmap_stop = true; }
        unmap_cv.notify_all();
        if (unmapper.joinable()) unmapper.join();
    };
    off_t file_pos = 0;
    uint64_t first_block = 0, total_bytes = 0;
    size_t carry = 0;                       // valid carry bytes at the front of the current buffer (0 for the first batch)
    uint8_t carry_bytes[MODES_CARRY_BYTES];
    bool eof = false;
    int rc = 0;
    // a pipe is served at the pace it delivers (read_paced): batches of whole buffers, the bytes read beyond the last whole one wait here
    const bool paced = !seekable && lseek(fd, 0, SEEK_CUR) == (off_t)-1;
    std::vector<uint8_t> pend;
    if (paced) {
        pend.reserve(MODES_DATA_LEN);
#ifdef F_SETPIPE_SZ
        (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);                                  // (a FIFO's default 64 KiB is 16 wake-ups per buffer)
#endif
    }
    for (uint64_t b = 0; !eof; b++) {
        {   // the lane of batch b is free once batch b - nlanes has been resolved
            std::unique_lock<std::mutex> g(m);
            cv.wait(g, [&] { return failed || b < resolved + (uint64_t)nlanes; });
            if (failed) { rc = 1; break; }
        }
        if (!lane_ready((int)(b % (uint64_t)nlanes))) { rc = 1; break; }
        Lane &ln = lanes[(size_t)(b % (uint64_t)nlanes)];
        if (carry) memcpy(ln.buf, carry_bytes, MODES_CARRY_BYTES);               // dump1090.c:481
        size_t got = 0;
        uint8_t *dst = ln.buf + carry;
        const off_t pos_before = file_pos;
        bool ok, ended = false;
        if (paced) {
            if (!pend.empty()) memcpy(dst, pend.data(), pend.size());
            size_t n = 0;
            ok = read_paced(fd, dst, pend.size(), batch_bytes, opt.flush_ms * 1e-3, &n, &ended);
            pend.clear();
            got = n;
            if (ok && !ended && n < batch_bytes) {                           // the deadline: whole buffers go, the rest waits for the next batch
                got = n - n % MODES_DATA_LEN;
                pend.assign(dst + got, dst + n);
            }
        } else
            ok = seekable ? read_parallel(pool, fd, map, map_len, &file_pos, dst, batch_bytes, &got)
                          : read_full(fd, dst, batch_bytes, &got);
        if (!ok) { perror("read"); rc = 1; break; }
        // The batch has been copied out of the mapping and is never looked at again: hand its pages back now, on a thread of its
        // own, instead of leaving 2 M page-table entries of an 8 GiB file to the exit of the process (~0.1 s there, and nothing
        // overlaps it).  (Batches start at multiples of 256 KiB: page aligned.)
        if (map && got && getenv("MODES_HOST_KEEP_MAPPING") == nullptr) {
            {
                std::lock_guard<std::mutex> g(unmap_m);
                unmap_q.emplace_back(const_cast<uint8_t *>(map) + pos_before, (size_t)got & ~(size_t)4095);
                unmapped_to = (size_t)pos_before + ((size_t)got & ~(size_t)4095);
            }
            unmap_cv.notify_one();
        }
        while (got < batch_bytes && opt.loop && fd != 0 && !seekable) { // dump1090.c:488-494
            if (lseek(fd, 0, SEEK_SET) == -1) break;
            size_t more = 0;
            if (!read_full(fd, dst + got, batch_bytes - got, &more)) { perror("read"); rc = 1; break; }
            if (more == 0) break;                                      // empty file
            got += more;
        }
        if (rc) break;
        total_bytes += got;
        // The reader publishes one buffer per full 262144 bytes and one more at EOF
        // (dump1090.c:484-510): a short batch ends the stream with floor(got/262144)+1 buffers.
        uint64_t nblocks = got / MODES_DATA_LEN;
        if (paced ? ended : got < batch_bytes) { eof = true; nblocks += 1; }
        const uint64_t byte0 = first_block * (uint64_t)MODES_DATA_LEN - carry;
        if (modes_gpu_submit_host(ln.gpu, ln.buf, carry + got, byte0, first_block, nblocks) != MODES_OK) {
            fprintf(stderr, "GPU demodulation failed: %s\n", modes_gpu_last_error(ln.gpu));
            rc = 1;
            break;
        }
        if (!eof) {                                                     // the last 476 bytes travel to the next batch
            memcpy(carry_bytes, ln.buf + carry + got - MODES_CARRY_BYTES, MODES_CARRY_BYTES);
            carry = MODES_CARRY_BYTES;
            first_block += nblocks;
        }
        {
            std::lock_guard<std::mutex> g(m);
            submitted = b + 1;
        }
        cv.notify_all();
    }
    const double t_read_done = now_s();                                     // the last batch is submitted
    {
        std::lock_guard<std::mutex> g(m);
        reader_done = true;
        if (rc) failed = true;
    }
    cv.notify_all();
    resolver.join();
    if (failed) rc = 1;
    const double t_end = now_s();
    stop_unmapper();

    if (rc == 0 && opt.stats) {                                        // dump1090.c:2993-3006
        modes_host_stats st;
        modes_host_get_stats(host, &st);
        char text[512];
        modes_format_stats(&st, text);
        fputs(text, stdout);
    }
    fflush(stdout);
    const double t_flushed = now_s();
    for (auto &t : lane_makers) t.join();                                   // (a stream shorter than the lanes' set-up)
    if (!opt.clean_exit) {
        // Everything is printed.  What is left - unpinning and freeing the lanes' buffers (~0.06 s), unmapping the file
        // (~0.1 s for 8 GiB), the HIP runtime's own exit handlers (~0.1 s) - the kernel does for a dead process anyway.
        if (opt.timing) {
            const double stream_s = t_end - t_ready;
            fprintf(stderr,
                    "{\"bytes\": %llu, \"devices\": %d, \"lanes\": %d, \"init_s\": %.4f, \"stream_s\": %.4f, \"total_s\": %.4f, "
                    "\"stream_GBps\": %.2f, \"stream_Msamples_per_s\": %.1f, \"sink_calls\": %llu, "
                    "\"init\": {\"first_context_s\": %.4f, \"first_buffer_s\": %.4f, \"all_lanes_s\": %.4f}, \"drain_s\": %.4f, \"teardown\": null}\n",
                    (unsigned long long)total_bytes, ndev, nlanes, t_ready - t_start, stream_s, now_s() - t_start,
                    stream_s > 0 ? total_bytes / stream_s / 1e9 : 0.0, stream_s > 0 ? total_bytes / 2 / stream_s / 1e6 : 0.0,
                    (unsigned long long)n_messages_out, t_created[0] - t_start, t_pinned[0] - t_start,
                    *std::max_element(t_pinned.begin(), t_pinned.end()) - t_start, t_end - t_read_done);
            fflush(stderr);
        }
        _exit(rc);
    }
    modes_host_destroy(host);
    modes_tracker_destroy(sink.tracker);
    const double t_host = now_s();
    for (auto &ln : lanes) {
        modes_gpu_host_free(ln.gpu, ln.buf);
        modes_gpu_destroy(ln.gpu);
    }
    const double t_lanes = now_s();
    // Only what the unmapper has not released: the pages it gave back during the run are free address space, and whatever was
    // allocated since (growing output buffers, late lanes' pinned memory, HIP's pools, thread stacks) may live there by now -
    // unmapping the whole original range would take that memory away from under its owners.
    if (map && unmapped_to < map_len) munmap(const_cast<uint8_t *>(map) + unmapped_to, map_len - unmapped_to);
    if (fd > 0) close(fd);
    const double t_unmapped = now_s();
    if (opt.timing) {
        const double stream_s = t_end - t_ready;
        fprintf(stderr,
                "{\"bytes\": %llu, \"devices\": %d, \"lanes\": %d, \"init_s\": %.4f, \"stream_s\": %.4f, \"total_s\": %.4f, "
                "\"stream_GBps\": %.2f, \"stream_Msamples_per_s\": %.1f, \"sink_calls\": %llu, "
                "\"init\": {\"first_context_s\": %.4f, \"first_buffer_s\": %.4f, \"all_lanes_s\": %.4f}, \"drain_s\": %.4f, "
                "\"teardown\": {\"flush_s\": %.4f, \"host_s\": %.4f, \"lanes_s\": %.4f, \"unmap_s\": %.4f}}\n",
                (unsigned long long)total_bytes, ndev, nlanes, t_ready - t_start, stream_s, t_unmapped - t_start,
                stream_s > 0 ? total_bytes / stream_s / 1e9 : 0.0, stream_s > 0 ? total_bytes / 2 / stream_s / 1e6 : 0.0,
                (unsigned long long)n_messages_out, t_created[0] - t_start, t_pinned[0] - t_start,
                *std::max_element(t_pinned.begin(), t_pinned.end()) - t_start, t_end - t_read_done,
                t_flushed - t_end, t_host - t_flushed, t_lanes - t_host, t_unmapped - t_lanes);
    }
    return rc;
}
