// host_ranks_shared.cpp - dump1090_amd --ranks N --resolve-on-ranks (include/modes_host.h "resolve on the ranks that demodulated";
// dump1090_amd/distributed.py has the same protocol over torch.distributed): every rank resolves its own batch of a round from a guessed
// whitelist; what the ranks tell each other - guesses, what they wrote, their texts - lies in a mapping made before the fork, a sequence
// number per (round slot, rank) and table says when it is there.  The ranks of a round confirm each other IN ORDER: rank r waits for
// rank r - 1 to be final, so the state it rebuilds from the tables of the ranks before it is the true one; it checks its logged answers
// against it (and resolves again if one is wrong) and is final itself.  Rank 0 prints the texts of a round in rank order.  No record
// leaves its rank and NO COMMUNICATOR EXISTS: no RCCL start-up, no IPC mode to guess, and the ranks may share a device.
//
// Serves --raw (the lean multi-threaded resolve), --onlyaddr / --raw-net (the general resolve, the host's own sink) and --stats (every
// rank counts its own batches, preamble positions included; rank 0 adds the nine counters up).  --sbs stays with the gather: its lines
// read the aircraft table, which every message writes.  On a live stream the whitelist's 60 s run on rank 0's clock, read once per round.
#include "host_ranks.h"

namespace modes_cli {
namespace {

struct Rank {                                                                 // one per (round slot, rank)
    std::atomic<uint64_t> guess_seq, final_seq;                               // round + 1 once `guess` / everything else is published
    uint64_t lines, nbytes;
    uint32_t guess[MODES_ICAO_SLOTS];
    uint32_t w_addr[MODES_ICAO_SLOTS];
    int64_t w_seen[MODES_ICAO_SLOTS];
    uint8_t written[MODES_ICAO_SLOTS];
};
struct Head { std::atomic<uint64_t> printed; std::atomic<int> failed; std::atomic<uint64_t> reruns; int64_t now[16]; };   // printed: rounds rank 0 has written out
struct Totals { std::atomic<uint64_t> ready; modes_host_stats st; };          // --stats: a rank's nine counters when its last round is final

struct SharedStages : RoundStages {
    Head *head = nullptr;
    Rank *ranks = nullptr;
    Totals *totals = nullptr;
    char *text_mem = nullptr;
    size_t text_cap = 0;
    int N = 0, depth = 0;
    modes_host *probe = nullptr;                                              // checks logged answers against a rebuilt state
    std::vector<uint32_t> truth_addr, st_addr;                                // the whitelist every round < qq left (the same on every rank)
    std::vector<int64_t> truth_seen, st_seen;
    std::vector<modes_icao_lookup> lookups;
    uint64_t applied = 0;                                                     // rounds whose writes are in `truth`

    const char *name() const override { return "ranks"; }
    bool every_rank_resolves() const override { return true; }
    bool nothing_to_tear_down_together() const override { return true; }

    Rank &at(uint64_t round, int r) { return ranks[(size_t)(round % (uint64_t)depth) * (size_t)N + (size_t)r]; }
    bool wait(const std::atomic<uint64_t> &a, uint64_t v) {                   // false: another rank failed (or this one's peers are gone)
        for (int spin = 0; a.load(std::memory_order_acquire) < v;) {
            if (head->failed.load()) return false;
            if (spin < 100000) spin++;                                        // (saturates: a live pipe can keep a rank here for minutes)
            if (spin > 200) usleep(spin < 2000 ? 20 : 1000);                  // short waits spin, long ones - a batch of a live stream - sleep
        }
        return true;
    }
    static void apply(std::vector<uint32_t> &addr, std::vector<int64_t> &seen, const Rank &w) {
        for (uint32_t sidx = 0; sidx < MODES_ICAO_SLOTS; sidx++)
            if (w.written[sidx]) { addr[sidx] = w.w_addr[sidx]; seen[sidx] = w.w_seen[sidx]; }
    }

    bool before_fork(RanksRun &run) override {
        N = run.N;
        depth = run.depth;
        text_cap = ((size_t)run.opt.gather_cap * 62 + 64 + 4095) & ~(size_t)4095;         // two 31-byte lines per record at most
        if (depth > 16) { fprintf(stderr, "--resolve-on-ranks: --depth %d (at most 16)\n", depth); return false; }
        const size_t ctl = (sizeof(Head) + (size_t)depth * (size_t)N * sizeof(Rank) + (size_t)N * sizeof(Totals) + 4095) & ~(size_t)4095;
        void *m = mmap(nullptr, ctl + (size_t)depth * (size_t)N * text_cap, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) { perror("--resolve-on-ranks: shared buffers"); return false; }
        head = new (m) Head;
        head->printed.store(0); head->failed.store(0); head->reruns.store(0);
        ranks = reinterpret_cast<Rank *>(static_cast<uint8_t *>(m) + sizeof(Head));
        for (size_t i = 0; i < (size_t)depth * (size_t)N; i++) { new (&ranks[i]) Rank; ranks[i].guess_seq.store(0); ranks[i].final_seq.store(0); }
        totals = reinterpret_cast<Totals *>(ranks + (size_t)depth * (size_t)N);
        for (int r = 0; r < N; r++) { new (&totals[r]) Totals; totals[r].ready.store(0); }
        text_mem = static_cast<char *>(m) + ctl;
        truth_addr.assign(MODES_ICAO_SLOTS, 0); st_addr.assign(MODES_ICAO_SLOTS, 0);
        truth_seen.assign(MODES_ICAO_SLOTS, 0); st_seen.assign(MODES_ICAO_SLOTS, 0);
        return true;
    }
    bool connect(RanksRun &run) override {
        modes_host_config hcfg{run.opt.fix_errors, run.opt.aggressive ? 1 : 0, run.opt.check_crc, 0};
        probe = modes_host_create(&hcfg);
        return probe != nullptr;
    }
    bool lane_output(RanksRun &, int) override { return true; }               // the list stays on this rank: the context's own pinned list (modes_gpu_fetch)

    // round qq of this rank (its batch's kernels are queued; the next batch's already run)
    bool round(RanksRun &run, uint64_t qq, bool have) {
        const int l = (int)(qq % (uint64_t)run.depth);
        Rank &me = at(qq, run.rank);
        // the slot's previous tenant: round qq - run.depth, printed?
        if (qq >= (uint64_t)run.depth && !wait(head->printed, qq - (uint64_t)run.depth + 1)) return false;
        const modes_record *recs = nullptr;
        uint64_t nrec = 0;
        const uint64_t *cands = nullptr;                                     // --stats: every preamble position of the batch (dump1090.c:1651)
        uint64_t ncand = 0;
        if (have) {
            modes_gpu_result res{};
            if (modes_gpu_fetch(run.lanes[(size_t)l].gpu, &res) != MODES_OK) { run.fail("GPU demodulation failed", modes_gpu_last_error(run.lanes[(size_t)l].gpu)); return false; }
            recs = res.records;
            nrec = res.n_records;
            cands = res.candidates;
            ncand = res.n_candidates;
            if (nrec > run.opt.gather_cap) { run.fail("resolve", "a batch's records exceed --gather-records (the text buffers are sized by it)"); return false; }
        }
        modes_host_whitelist_guess(run.host, &recs, &nrec, 1, me.guess, run.opt.resolve_threads);
        if (run.rank == 0) head->now[l] = run.feed ? (int64_t)time(nullptr) : 0;   // one clock per round: rank 0's (a live stream: dump1090.c:913,924)
        me.guess_seq.store(qq + 1, std::memory_order_release);
#ifdef MODES_TEST_HOOKS                                                          // (stub builds: a rank that dies between its guess and its final tables)
        if (const char *die = getenv("MODES_RR_DIE"); die && atoi(die) == run.rank && strchr(die, ':') && (uint64_t)atoll(strchr(die, ':') + 1) == qq) raise(SIGKILL);
#endif
        // the state the round starts from: every earlier round, final on every rank
        for (; applied < qq; applied++)
            for (int r = 0; r < run.N; r++) {
                if (!wait(at(applied, r).final_seq, applied + 1)) return false;
                apply(truth_addr, truth_seen, at(applied, r));
            }
        st_addr = truth_addr;
        st_seen = truth_seen;
        if (!wait(at(qq, 0).guess_seq, qq + 1)) return false;
        const int64_t now = head->now[l];
        for (int r = 0; r < run.rank; r++) {                                     // ... overlaid with what the ranks before this one expect to write
            const Rank &o = at(qq, r);
            if (!wait(o.guess_seq, qq + 1)) return false;
            for (uint32_t sidx = 0; sidx < MODES_ICAO_SLOTS; sidx++)
                if (o.guess[sidx] != MODES_ICAO_NONE) { st_addr[sidx] = o.guess[sidx]; st_seen[sidx] = now; }
        }
#ifdef MODES_TEST_HOOKS                                                          // (the stub builds of tools/sanitize_host.sh: a wrong start on purpose)
        if (const char *sp = getenv("MODES_RR_SPOIL"); sp && *sp && run.rank > 0) { std::fill(st_addr.begin(), st_addr.end(), 0u); std::fill(st_seen.begin(), st_seen.end(), (int64_t)0); }
#endif
        char *text = text_mem + ((size_t)l * (size_t)run.N + (size_t)run.rank) * text_cap;
        lookups.resize((size_t)nrec * 2 + 16);
        uint64_t nb = 0, nl = 0, lines = 0;
        // --raw: the lean resolve on several threads.  The other sinks this mode serves go through the general resolve and the run.host's
        // own sink: --onlyaddr / --raw-net format their line there, --stats prints nothing and counts (with the batch's preamble
        // positions: the counters of dump1090.c:2993-3006 are sums of per-batch counts, rank 0 adds the ranks' up at the end).
        // A resolve that is repeated starts from the counters the first one found.
        modes_host_stats st_before;
        modes_host_get_stats(run.host, &st_before);
        auto resolve_from = [&](const std::vector<uint32_t> &addr, const std::vector<int64_t> &seen) {
            modes_host_set_time(run.host, now);
            modes_host_set_whitelist(run.host, addr.data(), seen.data());
            if (run.raw_fast) {
                lines = modes_host_resolve_raw_spec(run.host, &recs, &nrec, 1, text, text_cap, &nb, run.opt.resolve_threads, me.written, lookups.data(), lookups.size(), &nl);
                return;
            }
            modes_host_set_stats(run.host, &st_before);
            run.sink.out.clear();
            lines = modes_host_resolve_spec(run.host, recs, nrec, cands, ncand, on_message, &run.sink, me.written, lookups.data(), lookups.size(), &nl);
            nb = run.sink.out.size();
            if (nb < text_cap) memcpy(text, run.sink.out.data(), (size_t)nb);
            run.sink.out.clear();
        };
        resolve_from(st_addr, st_seen);
        // confirmation, in rank order: the ranks before this one are final -> their tables give the true start
        if (run.rank > 0) {
            if (!wait(at(qq, run.rank - 1).final_seq, qq + 1)) return false;
            st_addr = truth_addr;
            st_seen = truth_seen;
            for (int r = 0; r < run.rank; r++) apply(st_addr, st_seen, at(qq, r));
            modes_host_set_time(probe, now);
            modes_host_set_whitelist(probe, st_addr.data(), st_seen.data());
            if (nl > lookups.size() || !modes_host_whitelist_check(probe, lookups.data(), nl)) {
                resolve_from(st_addr, st_seen);                              // rare: an answer taken from the guess was wrong
                head->reruns.fetch_add(1);
            }
        }
        if (nb >= text_cap) { run.fail("resolve", "the text of a batch outgrew its buffer"); return false; }
        modes_host_get_whitelist(run.host, me.w_addr, me.w_seen);
        me.lines = lines;
        me.nbytes = nb;
        me.final_seq.store(qq + 1, std::memory_order_release);
        if (run.rank == 0) {                                                     // the round's listing, run.rank after rank
            for (int r = 0; r < run.N; r++) {
                const Rank &o = at(qq, r);
                if (!wait(o.final_seq, qq + 1)) return false;
                if (o.nbytes) fwrite(text_mem + ((size_t)l * (size_t)run.N + (size_t)r) * text_cap, 1, (size_t)o.nbytes, run.out);
                run.n_messages_out += o.lines;
            }
            fflush(run.out);
            // (every rank reads a round's tables when it starts the NEXT round; the slot is written again run.depth rounds later, by ranks that
            //  have been through the round after this one - which needs every rank final there, i.e. past its reading of these)
            head->printed.store(qq + 1, std::memory_order_release);
        }
        return true;
    }
    void exchange(RanksRun &run, uint64_t q) override {
        if (!round(run, q, run.has[(size_t)(q % (uint64_t)depth)] != 0) && !run.rc) run.fail("resolve", "another rank failed");
    }
    void resolve(RanksRun &, uint64_t) override {}                            // (round() has printed the round)
    void failed(RanksRun &) override { head->failed.store(1); }

    void print_stats(RanksRun &run) override {                               // every rank's counters -> rank 0, which adds them up
        modes_host_get_stats(run.host, &totals[run.rank].st);
        totals[run.rank].ready.store(1, std::memory_order_release);
        if (run.rank != 0) return;
        modes_host_stats sum{};
        for (int r = 0; r < N; r++) {
            if (!wait(totals[r].ready, 1)) { run.fail("resolve", "another rank failed"); break; }
            const modes_host_stats &o = totals[r].st;
            sum.valid_preamble += o.valid_preamble > 0 ? o.valid_preamble : 0;   // (a rank that never had a batch saw no positions)
            sum.out_of_phase += o.out_of_phase; sum.demodulated += o.demodulated; sum.goodcrc += o.goodcrc; sum.badcrc += o.badcrc;
            sum.fixed += o.fixed; sum.single_bit_fix += o.single_bit_fix; sum.two_bits_fix += o.two_bits_fix;
        }
        if (run.rc) { fflush(run.out); fflush(stderr); head->failed.store(1); _exit(run.finish(run.rc)); }
        char text[512];
        modes_format_stats(&sum, text);
        fputs(text, run.out);
        fflush(run.out);
    }
    void print_timing(RanksRun &run, size_t size, double t_end) override {
        const double stream_s = t_end - run.t_ready;
        fprintf(stderr, "{\"bytes\": %zu, \"ranks\": %d, \"rounds\": %llu, \"init_s\": %.4f, \"stream_s\": %.4f, \"total_s\": %.4f, \"stream_GBps\": %.2f, "
                        "\"sink_calls\": %llu, \"resolve_on\": \"ranks\", \"reruns\": %llu}\n",
                size, N, (unsigned long long)run.nrounds, run.t_ready - run.t_start, stream_s, t_end - run.t_start, stream_s > 0 ? size / stream_s / 1e9 : 0.0,
                (unsigned long long)run.n_messages_out, (unsigned long long)head->reruns.load());
    }
    void teardown(RanksRun &) override { if (probe) modes_host_destroy(probe); }
};

}  // namespace

std::unique_ptr<RoundStages> make_shared_stages() { return std::unique_ptr<RoundStages>(new SharedStages); }

}  // namespace modes_cli
