// host_single.cpp - dump1090_amd as ONE process driving N devices (--gpu / --gpus / --gpu-list): the reference's reader thread / main
// thread pair (dump1090.c:460-527, :2965-2990) widened to batches of buffers and lanes of contexts.
//
//   reader (main thread + a pool of pread workers)          resolver thread
//   batch b -> lane b mod L: pinned buffer, GPU context  -> fetch(lane) in batch order, modes_host_resolve,
//   on device b mod N; submit = async H2D + kernels         print; the lane is free again
//
// A regular file fills its batches; input that cannot seek is served at the pace it delivers (read_paced, host_common.h).
#include "host_common.h"

namespace modes_cli {

int run_single(Options &opt, double t_start) {
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

}  // namespace modes_cli
