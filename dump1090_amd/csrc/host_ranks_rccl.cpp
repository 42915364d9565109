// host_ranks_rccl.cpp - dump1090_amd --ranks N, the record lists gathered to rank 0 over RCCL / xGMI (include/modes_gather.h):
// the kernels of every rank write list and length straight into the gather's device buffers, round q's lengths travel in one all-gather, the
// lists as grouped send / recv to their final offsets on rank 0 (for --stats: the preamble positions as a second list), one device-to-host
// copy, and rank 0 - which owns the one piece of cross-buffer state, the ICAO whitelist - resolves the round and prints it.
#include "host_ranks.h"

namespace modes_cli {
namespace {

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

struct RcclStages : RoundStages {
    GatherApi G;
    modes_gather *g = nullptr;
    std::vector<int> rd, wr;                                                  // the unique id's way from rank 0 to rank r: pipes made before the fork
    unsigned char id[MODES_GATHER_ID_BYTES];
    double t_loaded = 0, t_id = 0, t_comm = 0;

    const char *name() const override { return "root"; }
    bool every_rank_resolves() const override { return false; }
    bool nothing_to_tear_down_together() const override { return false; }

    bool before_fork(RanksRun &run) override {
        rd.assign((size_t)run.N, -1);
        wr.assign((size_t)run.N, -1);
        for (int r = 1; r < run.N; r++) {
            int fds[2];
            if (pipe(fds) != 0) { perror("pipe"); return false; }
            rd[(size_t)r] = fds[0];
            wr[(size_t)r] = fds[1];
        }
        return true;
    }
    void after_fork(RanksRun &run) override {                                // keep only this rank's end(s)
        for (int r = 1; r < run.N; r++) {
            if (run.rank == 0) close(rd[(size_t)r]);
            else { close(wr[(size_t)r]); if (r != run.rank) close(rd[(size_t)r]); }
        }
    }
    bool connect(RanksRun &run) override {
        const Options &opt = run.opt;
        const int rank = run.rank, N = run.N;
        if (!G.load()) return false;
        t_loaded = now_s();
        if (rank == 0) {
            if (G.unique_id(id) != MODES_OK) { fprintf(stderr, "--ranks: %s\n", G.last_error(nullptr)); return false; }
            for (int r = 1; r < N; r++) { if (!write_all(wr[(size_t)r], id, sizeof id)) { perror("--ranks: id pipe"); return false; } close(wr[(size_t)r]); }
        } else {
            if (!read_all(rd[(size_t)rank], id, sizeof id)) { fprintf(stderr, "--ranks: rank %d got no id from rank 0\n", rank); return false; }
            close(rd[(size_t)rank]);
        }
        // --stats: the preamble positions of every batch travel to rank 0 with its records (the second list of the gather)
        const uint64_t batch_positions = opt.batch_blocks * (uint64_t)MODES_BLOCK_STRIDE;
        const uint32_t cap_cands = !opt.stats ? 0u : opt.gather_cands ? opt.gather_cands : (uint32_t)std::max<uint64_t>(4096, batch_positions / 64);
        modes_gather_config gc{run.device, rank, N, opt.gather_cap, (uint32_t)run.depth, cap_cands};
        t_id = now_s();
        if (const int crc = G.create(&gc, id, &g); crc != MODES_OK) {
            fprintf(stderr, "--ranks: rank %d: %s\n", rank, G.last_error(nullptr));
            if (crc == MODES_GATHER_ERR_PROBE) {
                if (rank != 0) { fflush(stderr); _exit(kProbeStatus); }          // rank 0's watchdog takes it from here
                {
                    std::lock_guard<std::mutex> lk(run.kid_mu);                  // (not while the watchdog is reaping)
                    run.restart_with_other_ipc_mode();
                }
            }
            return false;
        }
        t_comm = now_s();
        return true;
    }
    bool lane_output(RanksRun &run, int l) override {                        // the kernels write list and length into the gather's device buffers
        void *d_rec = nullptr, *d_cnt = nullptr;
        uint64_t cap = 0;
        if (G.output(g, (uint32_t)l, &d_rec, &cap, &d_cnt) != MODES_OK || modes_gpu_set_output(run.lanes[(size_t)l].gpu, d_rec, cap, d_cnt) != MODES_OK) {
            fprintf(stderr, "rank %d: %s / %s\n", run.rank, G.last_error(g), modes_gpu_last_error(run.lanes[(size_t)l].gpu));
            return false;
        }
        return true;
    }
    void exchange(RanksRun &run, uint64_t q) override {                      // kernels done -> lengths -> transfers
        const int l = (int)(q % (uint64_t)run.depth);
        if (run.has[(size_t)l]) {
            modes_gpu_result res{};
            // (a list that outgrew the buffers still goes through the length exchange: every rank then fails together)
            const int frc = modes_gpu_fetch_device(run.lanes[(size_t)l].gpu, &res);
            if (frc != MODES_OK && frc != MODES_ERR_OVERFLOW) run.fail("GPU demodulation failed", modes_gpu_last_error(run.lanes[(size_t)l].gpu));
            else if (run.opt.stats && G.set_candidates(g, (uint32_t)l, res.candidates, res.n_candidates) != MODES_OK) run.fail("gather", G.last_error(g));
        } else if (G.set_empty(g, (uint32_t)l) != MODES_OK) run.fail("gather", G.last_error(g));
        if (!run.rc && (G.counts(g, (uint32_t)l) != MODES_OK || G.records(g, (uint32_t)l) != MODES_OK)) run.fail("gather", G.last_error(g));
    }
    void resolve(RanksRun &run, uint64_t q) override {                       // rank 0 resolves what arrived
        const Options &opt = run.opt;
        const int l = (int)(q % (uint64_t)run.depth);
        const modes_record *recs = nullptr;
        uint64_t nrec = 0;
        if (G.wait(g, (uint32_t)l, &recs, &nrec, nullptr) != MODES_OK) { run.fail("gather", G.last_error(g)); return; }
        if (run.rank != 0) return;
        // a live stream: the whitelist's 60 s run on the wall clock (dump1090.c:913,924) - read once per ROUND of N batches here (the
        // one-process host reads it per batch: next to the TTL a listing can differ by that much)
        if (run.feed) modes_host_set_time(run.host, (int64_t)time(nullptr));
        const uint64_t *cands = nullptr;
        uint64_t ncand = 0;
        if (opt.stats && G.candidates(g, (uint32_t)l, &cands, &ncand) != MODES_OK) { run.fail("gather", G.last_error(g)); return; }
        if (run.raw_fast) {                                                  // the listing goes out from where the resolve's threads wrote it
            modes_text_piece pieces[80];
            uint32_t np = 0;
            run.n_messages_out += modes_host_resolve_raw_pieces(run.host, &recs, &nrec, 1, pieces, 80, &np, nullptr, opt.resolve_threads);
            for (uint32_t i = 0; i < np; i++) fwrite(pieces[i].base, 1, (size_t)pieces[i].len, run.out);
            if (np) fflush(run.out);
        } else
            run.n_messages_out += modes_host_resolve(run.host, recs, nrec, cands, ncand, on_message, &run.sink);
        if (!run.sink.out.empty()) {
            fwrite(run.sink.out.data(), 1, run.sink.out.size(), run.out);
            fflush(run.out);
            run.sink.out.clear();
        }
    }
    void print_stats(RanksRun &run) override {
        if (run.rank != 0) return;
        modes_host_stats hs;
        modes_host_get_stats(run.host, &hs);
        char text[512];
        modes_format_stats(&hs, text);
        fputs(text, run.out);
        fflush(run.out);
    }
    void print_timing(RanksRun &run, size_t size, double t_end) override {
        modes_gather_stats st{};
        G.get_stats(g, &st);
        const double stream_s = t_end - run.t_ready;
        fprintf(stderr,
                "{\"bytes\": %zu, \"ranks\": %d, \"rounds\": %llu, \"init_s\": %.4f, \"stream_s\": %.4f, \"total_s\": %.4f, \"stream_GBps\": %.2f, "
                "\"init\": {\"load_gather_library_s\": %.4f, \"unique_id_s\": %.4f, \"communicator_s\": %.4f, \"lanes_s\": %.4f}, "
                "\"sink_calls\": %llu, \"rccl\": {\"version\": %d, \"nranks\": %d, \"p2p_ops\": %llu, \"bytes_received\": %llu, \"gather_ms\": %.3f}}\n",
                size, run.N, (unsigned long long)run.nrounds, run.t_ready - run.t_start, stream_s, t_end - run.t_start, stream_s > 0 ? size / stream_s / 1e9 : 0.0,
                t_loaded - run.t_start, t_id - t_loaded, t_comm - t_id, run.t_ready - t_comm,
                (unsigned long long)run.n_messages_out, st.rccl_version, st.nranks, (unsigned long long)st.p2p_ops,
                (unsigned long long)st.bytes_received, st.gather_ms);
    }
    void teardown(RanksRun &) override { G.destroy(g); }
};

}  // namespace

std::unique_ptr<RoundStages> make_rccl_stages() { return std::unique_ptr<RoundStages>(new RcclStages); }

}  // namespace modes_cli
