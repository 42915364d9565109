// host_ranks.cpp - dump1090_amd --ranks N, the part both ways of resolving share: ONE PROCESS PER GPU - this program forks N - 1 copies of
// itself before any HIP call -, rank r demodulates batches r, r + N, r + 2N, ... of the stream on its own device, round q = batches
// qN .. qN + N - 1, so rank order is stream order and rank 0 prints every round as it is complete.  A regular file is mapped by every
// rank; a pipe or --loop is read by rank 0 alone, which deals the batches out through shared memory (at the pace the pipe delivers:
// read_paced).  What happens to a round's records is the stages' business (host_ranks.h): host_ranks_rccl.cpp gathers the lists to rank 0
// over RCCL / xGMI (north_star's shape), host_ranks_shared.cpp leaves them where they are (--resolve-on-ranks).
#include "host_ranks.h"

namespace modes_cli {

int run_ranks(const Options &opt, double t_start) {
    const int N = opt.ranks;
    // A regular file is mapped by every rank, which takes its own batches.  A pipe (--ifile -) or an endless replay (--loop) has ONE reader:
    // rank 0 reads it on a thread of its own into slots of a shared mapping made before the fork (one slot per rank and batch in flight:
    // batch b belongs to rank b mod N), and rank r copies its batch from its slot to its pinned buffer exactly as it would copy it out of a
    // file mapping - one read() more per byte than the file path, in front of N PCIe links (round 5; before: refused, --gpus N named).
    RanksRun run(opt, N, std::max(3, opt.depth), opt.loop || opt.filename == "-", (size_t)opt.batch_blocks * MODES_DATA_LEN);
    run.t_start = t_start;
    const bool feed = run.feed;
    const int depth = run.depth;
    const size_t batch_bytes = run.batch_bytes;
    int &rank = run.rank;
    int &rc = run.rc;
    uint64_t &nrounds = run.nrounds;
    std::vector<Lane> &lanes = run.lanes;
    std::vector<char> &has = run.has;
    std::unique_ptr<RoundStages> stages = opt.resolve_on_ranks ? make_shared_stages() : make_rccl_stages();
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
    if (!opt.devices.empty() && (int)opt.devices.size() != N) { fprintf(stderr, "--ranks %d with a --gpu-list of %zu devices\n", N, opt.devices.size()); return 1; }
    // this pool's host driver only supports dmabuf IPC: without this RCCL's cross-process buffers fail (hipIpcGetMemHandle:
    // invalid argument).  Kept if the caller has set it.
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    if (!stages->before_fork(run)) return 1;                                 // no HIP / RCCL call precedes the fork
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
    // never returns - or waiting for tables it will never publish -, so the job ends there and then: the other ranks are killed, the
    // status is 1.  Clean exits are recorded for finish().
    std::vector<int> kid_status(kids.size(), -1);                           // -1: running; else the wait status
    std::atomic<bool> watch_stop{false};
    std::thread watchdog;
    // The communicator's first transfer between two processes' devices is where a wrong IPC mode shows (this pool's hosts only do
    // dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0 - a guess made on one-GPU boxes).  HSA reads the variable when the runtime starts,
    // so the other value needs new processes: rank 0 ends its peers and runs the same command line once more with it - a wrong
    // guess then costs seconds, not the job.  Nothing has been printed by then.  A peer whose probe fails exits with kProbeStatus.
    int saved_stdout = -1;                                                   // the real stdout once fd 1 has been given to the libraries
    run.restart_with_other_ipc_mode = [&]() {                                // rank 0 only; returns only when there is no second try
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
                    std::lock_guard<std::mutex> lk(run.kid_mu);
                    for (size_t i = 0; i < kids.size(); i++) {
                        int st = 0;
                        if (kid_status[i] != -1 || waitpid(kids[i], &st, WNOHANG) != kids[i]) continue;
                        kid_status[i] = st;
                        if (WIFEXITED(st) && WEXITSTATUS(st) == 0) continue;
                        if (WIFEXITED(st) && WEXITSTATUS(st) == kProbeStatus) run.restart_with_other_ipc_mode();
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
    stages->after_fork(run);
    run.finish = [&](int status) {                                           // rank 0: the job's status is the worst rank's
        if (rank != 0) { fflush(stdout); fflush(stderr); _exit(status); }
        watch_stop.store(true);
        if (watchdog.joinable()) watchdog.join();
        for (size_t i = 0; i < kids.size(); i++) {
            int st = kid_status[i];
            if (st == -1) {
                if (status) kill(kids[i], SIGKILL);                          // rank 0 failed: its peers may be waiting for it
                if (waitpid(kids[i], &st, 0) < 0) st = 1;
            }
            if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) status = status ? status : 1;
        }
        return status;
    };
    auto &finish = run.finish;
    // RCCL prints a version banner on stdout when a communicator is made: stdout is the message sink of this program, so
    // the library side of the process gets stderr as its stdout and the sink keeps the real one
    fflush(stdout);
    saved_stdout = dup(1);
    FILE *out = run.out = saved_stdout >= 0 ? fdopen(saved_stdout, "w") : nullptr;
    if (!out || dup2(2, 1) < 0) { perror("--ranks: stdout"); return finish(1); }
    const int device = run.device = opt.devices.empty() ? rank : opt.devices[(size_t)rank];
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
        if (rank == 0 && !stages->every_rank_resolves() && (double)size < kRanksPaysFromBytes && !getenv("MODES_RANKS_QUIET"))   // (--resolve-on-ranks makes no communicator)
            fprintf(stderr, "--ranks %d: %.1f GiB is below the ~%.0f GB from which one process per GPU is faster than --gpus %d (one process, the "
                            "same devices, no communicator to start)\n", N, size / 1073741824.0, kRanksPaysFromBytes / 1e9, N);
        map = size ? static_cast<const uint8_t *>(mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0)) : nullptr;
        if (size && map == MAP_FAILED) { perror("mmap"); return finish(1); }
    } else if (rank == 0) {
        fd = opt.filename == "-" ? 0 : open(opt.filename.c_str(), O_RDONLY);
        if (fd == -1) { perror("Opening data file"); return finish(1); }
    }
    if (!stages->connect(run)) return finish(1);                             // the gather library and its communicator - or nothing
    lanes = std::vector<Lane>((size_t)depth);
    has.assign((size_t)depth, 0);
    for (int l = 0; l < depth; l++) {
        modes_gpu_config cfg{};
        cfg.device = device;
        cfg.fix_errors = opt.fix_errors;
        cfg.aggressive = opt.aggressive ? 1 : 0;
        cfg.keep_candidates = opt.stats ? 1 : 0;
        void *p = nullptr;
        if (modes_gpu_create(&cfg, &lanes[(size_t)l].gpu) != MODES_OK) { fprintf(stderr, "rank %d: GPU init failed: %s\n", rank, modes_gpu_last_error(nullptr)); return finish(1); }
        modes_gpu_set_timing(lanes[(size_t)l].gpu, 0);
        if (!stages->lane_output(run, l)) return finish(1);                  // the gather's device buffers, or the context's own pinned list
        if (modes_gpu_host_alloc(lanes[(size_t)l].gpu, MODES_CARRY_BYTES + batch_bytes, &p) != MODES_OK) {
            fprintf(stderr, "rank %d: %s\n", rank, modes_gpu_last_error(lanes[(size_t)l].gpu));
            return finish(1);
        }
        lanes[(size_t)l].buf = static_cast<uint8_t *>(p);
    }
    modes_host_config hcfg{opt.fix_errors, opt.aggressive ? 1 : 0, opt.check_crc, 0};
    modes_host *host = run.host = (rank == 0 || stages->every_rank_resolves()) ? modes_host_create(&hcfg) : nullptr;
    run.sink.host = host;
    run.sink.tracker = (rank == 0 && opt.sbs) ? modes_tracker_create() : nullptr;
    run.raw_fast = opt.raw && !opt.stats && !opt.sbs && !opt.raw_net && !opt.onlyaddr;
    const double t_ready = run.t_ready = now_s();
    (void)t_ready;

    // Batch b of the stream (the single-process host's geometry: a short - possibly empty - batch ends the stream and
    // carries the EOF buffer, dump1090.c:484-510); round q of the gather = batches qN .. qN + N - 1, rank r takes batch qN + r.
    // A file's batches are known from its size; a fed stream's (pipe, --loop) when the reader sees the end (never with --loop).
    uint64_t nbatches = feed ? ~0ull : size / batch_bytes + 1;
    auto rounds_of = [&](uint64_t nb) { return nb == ~0ull ? ~0ull : (nb + (uint64_t)N - 1) / (uint64_t)N; };
    nrounds = rounds_of(nbatches);
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
    auto fail_rank = [&](const char *what, const char *text) { run.fail(what, text); };
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
    auto exchange = [&](uint64_t q) { stages->exchange(run, q); };
    auto resolve = [&](uint64_t q) { stages->resolve(run, q); };
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
    if (rc) stages->failed(run);
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
    if (opt.stats) stages->print_stats(run);                                 // dump1090.c:2993-3006
    if (rank == 0 && opt.timing) stages->print_timing(run, size, t_end);
    if (stages->nothing_to_tear_down_together() && !opt.clean_exit) {
        // Everything is printed and no communicator exists whose teardown the ranks would have to do together: like the one-process
        // host, leave the unpinning, the unmapping and the runtime's exit handlers to the kernel (a third of a short run's wall clock).
        fflush(out);
        fflush(stderr);
        if (rank != 0) _exit(0);
        _exit(finish(0));
    }
    if (host) modes_host_destroy(host);
    modes_tracker_destroy(run.sink.tracker);
    for (auto &ln : lanes) { modes_gpu_host_free(ln.gpu, ln.buf); modes_gpu_destroy(ln.gpu); }
    stages->teardown(run);
    if (map) munmap(const_cast<uint8_t *>(map), size);
    if (fd > 0) close(fd);
    return finish(rc);
}

}  // namespace modes_cli
