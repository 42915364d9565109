// host_ranks.h - dump1090_amd --ranks N: what host_ranks.cpp (the fork, the reader, the lanes, the round loop) shares with the two ways
// of getting N ranks' records resolved: host_ranks_rccl.cpp (the lists travel to rank 0 over RCCL, which resolves them all) and
// host_ranks_shared.cpp (--resolve-on-ranks: every rank resolves its own, the ranks confirm each other through a shared mapping).
#ifndef MODES_HOST_RANKS_H
#define MODES_HOST_RANKS_H

#include "host_common.h"

namespace modes_cli {

constexpr int kProbeStatus = 75;          // exit status of a peer whose first transfer over the new communicator did not complete

// One --ranks process: owned by run_ranks, handed to the stages.
struct RanksRun {
    const Options &opt;
    const int N;                          // processes = GPUs of the split
    int rank = 0;
    const int depth;                      // lanes per rank: three stages are in flight (round q submits, q - 1 exchanges, q - 2 is resolved)
    const bool feed;                      // a pipe or --loop: rank 0 reads, the batches are dealt out through shared memory
    const size_t batch_bytes;
    int device = 0;
    std::vector<Lane> lanes;
    std::vector<char> has;                // per lane: this rank has a batch in the round the lane carries
    modes_host *host = nullptr;           // rank 0 (RCCL) / every rank (resolving on the ranks)
    Sink sink;
    FILE *out = nullptr;                  // the real stdout (fd 1 belongs to the libraries: RCCL prints a banner there)
    bool raw_fast = false;                // --raw and nothing else: the lean multi-threaded resolve
    uint64_t n_messages_out = 0;
    uint64_t nrounds = ~0ull;             // of the stream; ~0 until known (a fed stream: when the reader has seen the end)
    int rc = 0;
    double t_start = 0, t_ready = 0;
    std::mutex kid_mu;                    // rank 0: not while the watchdog is reaping
    std::function<int(int)> finish;       // rank 0: reap the other ranks -> the job's status; a peer: _exit(rc)
    std::function<void()> restart_with_other_ipc_mode;      // rank 0 only; returns only when there is no second try

    RanksRun(const Options &o, int n, int d, bool f, size_t bb) : opt(o), N(n), depth(d), feed(f), batch_bytes(bb), sink{&o, nullptr, {}, nullptr} {}
    void fail(const char *what, const char *text) { fprintf(stderr, "rank %d: %s: %s\n", rank, what, text); rc = 1; }
};

// The half of a round that differs between the two hosts, and what each of them needs around the loop.
struct RoundStages {
    virtual ~RoundStages() {}
    virtual const char *name() const = 0;
    virtual bool before_fork(RanksRun &) = 0;                 // mappings and pipes the ranks share: made while there is one process
    virtual void after_fork(RanksRun &) {}                    // (close the other ranks' ends)
    virtual bool connect(RanksRun &) = 0;                     // the gather library and its communicator / nothing at all
    virtual bool lane_output(RanksRun &, int l) = 0;          // where lane l's kernels write their list
    virtual bool every_rank_resolves() const = 0;
    virtual void exchange(RanksRun &, uint64_t q) = 0;        // round q: kernels done -> what travels, travels
    virtual void resolve(RanksRun &, uint64_t q) = 0;         // round q: rank 0 prints
    virtual void failed(RanksRun &) {}                        // this rank leaves the loop with an error: tell who can be told
    virtual void print_stats(RanksRun &) = 0;                 // --stats, dump1090.c:2993-3006
    virtual void print_timing(RanksRun &, size_t bytes, double t_end) = 0;
    virtual bool nothing_to_tear_down_together() const = 0;   // the default exit may leave everything to the kernel
    virtual void teardown(RanksRun &) = 0;
};
std::unique_ptr<RoundStages> make_rccl_stages();              // host_ranks_rccl.cpp
std::unique_ptr<RoundStages> make_shared_stages();            // host_ranks_shared.cpp

}  // namespace modes_cli
#endif
