// dump1090_amd - the C/C++ host of the reference's --ifile path on top of the two libraries:
//
//   read (--ifile <file>|-)  ->  libmodes_gfx950.so (scan + demod on the GPU, many buffers per call)
//                            ->  libmodes_host.so   (in-order resolve, decodeModesMessage, sink)
//                            ->  stdout (--raw / --onlyaddr / --stats; --sbs / --raw-net: the lines of the reference's TCP sinks)
//
// It keeps the reference's spellings and defaults for the flags of this path
// (dump1090.c:2869-2897,2922; defaults :299-319) and processes EVERY buffer the reference's
// reader publishes (the reference itself drops the last one most of the time: SURVEY.md 3.4).
// Live radio, networking, interactive mode and the debug dumps are out of scope.

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/time.h>
#include <unistd.h>

#include "../../include/modes_gfx950.h"
#include "../../include/modes_host.h"

namespace {

struct Options {
    std::string filename;
    bool loop = false, raw = false, onlyaddr = false, stats = false, sbs = false, raw_net = false;
    int fix_errors = 1, check_crc = 1, aggressive = 0;
    int device = 0;
    uint64_t batch_blocks = 512;           // 128 MiB of samples per GPU call (end to end, 8 GiB file: 0.58 s; 1024: 0.67 s)
    int read_threads = 8;                  // parallel pread() slices for regular files
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
        "--batch-blocks <n>       256 KiB buffers per GPU call (default: 512).\n"
        "--read-threads <n>       Threads reading a regular file (default: 8).\n"
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

// A regular file is read by several threads at once (pread on disjoint slices): one thread copying
// out of the page cache is ~10x slower than the PCIe link that follows.  Falls back to read() for
// pipes / stdin.  *got < want only at end of file.
bool read_parallel(int fd, off_t *pos, uint8_t *dst, size_t want, size_t *got, int nthreads) {
    *got = 0;
    if (want == 0) return true;
    const size_t slice = (want / (size_t)nthreads + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    std::vector<ssize_t> done((size_t)nthreads, 0);
    for (int t = 0; t < nthreads; t++) {
        const size_t lo = (size_t)t * slice;
        if (lo >= want) break;
        const size_t n = std::min(slice, want - lo);
        th.emplace_back([=, &done] {
            size_t have = 0;
            while (have < n) {
                ssize_t r = pread(fd, dst + lo + have, n - have, *pos + (off_t)(lo + have));
                if (r < 0) { if (errno == EINTR) continue; done[(size_t)t] = -1; return; }
                if (r == 0) break;
                have += (size_t)r;
            }
            done[(size_t)t] = (ssize_t)have;
        });
    }
    for (auto &x : th) x.join();
    for (size_t t = 0; t < th.size(); t++) {
        if (done[t] < 0) return false;
        *got += (size_t)done[t];
        if ((size_t)done[t] < std::min(slice, want - t * slice)) break;       // end of file inside this slice
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

}  // namespace

int main(int argc, char **argv) {
    Options opt;
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
        else if (!strcmp(a, "--gpu") && more) opt.device = atoi(argv[++j]);
        else if (!strcmp(a, "--batch-blocks") && more) opt.batch_blocks = strtoull(argv[++j], nullptr, 10);
        else if (!strcmp(a, "--read-threads") && more) opt.read_threads = std::max(1, atoi(argv[++j]));
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

    int fd = 0;
    if (opt.filename != "-" && (fd = open(opt.filename.c_str(), O_RDONLY)) == -1) {
        perror("Opening data file");
        return 1;
    }

    modes_gpu_config gcfg{};
    gcfg.device = opt.device;
    gcfg.fix_errors = opt.fix_errors;
    gcfg.aggressive = opt.aggressive ? 1 : 0;
    gcfg.keep_candidates = opt.stats ? 1 : 0;
    // Two GPU contexts and two pinned host buffers, used alternately: while batch b is copied to HBM
    // and demodulated, batch b+1 is read from the file - the job of the reference's reader thread
    // (dump1090.c:460-527, 2965-2990).  Results are resolved strictly in batch order.
    modes_gpu *gpu[2] = {nullptr, nullptr};
    uint8_t *buf[2] = {nullptr, nullptr};
    const size_t batch_bytes = (size_t)opt.batch_blocks * MODES_DATA_LEN;
    for (int k = 0; k < 2; k++) {
        if (modes_gpu_create(&gcfg, &gpu[k]) != MODES_OK) {
            fprintf(stderr, "GPU init failed: %s\n", modes_gpu_last_error(nullptr));
            return 1;
        }
        void *p = nullptr;
        if (modes_gpu_host_alloc(gpu[k], MODES_CARRY_BYTES + batch_bytes, &p) != MODES_OK) {
            fprintf(stderr, "pinned buffer: %s\n", modes_gpu_last_error(gpu[k]));
            return 1;
        }
        buf[k] = static_cast<uint8_t *>(p);
    }
    modes_host_config hcfg{opt.fix_errors, opt.aggressive ? 1 : 0, opt.check_crc, 0};
    modes_host *host = modes_host_create(&hcfg);
    Sink sink{&opt, host, {}, opt.sbs ? modes_tracker_create() : nullptr};

    // fetch + resolve + print the batch in flight on context k
    auto finish = [&](int k) -> bool {
        modes_gpu_result res{};
        if (modes_gpu_fetch(gpu[k], &res) != MODES_OK) {
            fprintf(stderr, "GPU demodulation failed: %s\n", modes_gpu_last_error(gpu[k]));
            return false;
        }
        modes_host_resolve(host, res.records, res.n_records, res.candidates, res.n_candidates, on_message, &sink);
        if (!sink.out.empty()) {
            fwrite(sink.out.data(), 1, sink.out.size(), stdout);
            fflush(stdout);
            sink.out.clear();
        }
        return true;
    };

    // Batch b covers buffers [first, first+n): host bytes = 476-byte carry + n*262144 new bytes.
    // --loop replays a file forever through the sequential path; a plain regular file is read in parallel
    const bool seekable = !opt.loop && fd != 0 && lseek(fd, 0, SEEK_CUR) != (off_t)-1;
    off_t file_pos = 0;
    uint64_t first_block = 0;
    size_t carry = 0;                       // valid carry bytes at the front of the current buffer (0 for the first batch)
    bool eof = false;
    int rc = 0, cur = 0;
    bool pending = false;                   // a batch is in flight on context 1-cur
    while (!eof) {
        size_t got = 0;
        uint8_t *dst = buf[cur] + carry;
        const bool ok = seekable ? read_parallel(fd, &file_pos, dst, batch_bytes, &got, opt.read_threads)
                                 : read_full(fd, dst, batch_bytes, &got);
        if (!ok) { perror("read"); rc = 1; break; }
        while (got < batch_bytes && opt.loop && fd != 0 && !seekable) { // dump1090.c:488-494
            if (lseek(fd, 0, SEEK_SET) == -1) break;
            size_t more = 0;
            if (!read_full(fd, dst + got, batch_bytes - got, &more)) { perror("read"); rc = 1; break; }
            if (more == 0) break;                                      // empty file
            got += more;
        }
        // The reader publishes one buffer per full 262144 bytes and one more at EOF
        // (dump1090.c:484-510): a short batch ends the stream with floor(got/262144)+1 buffers.
        uint64_t nblocks = got / MODES_DATA_LEN;
        if (got < batch_bytes) { eof = true; nblocks += 1; }
        const uint64_t byte0 = first_block * (uint64_t)MODES_DATA_LEN - carry;
        if (modes_gpu_submit_host(gpu[cur], buf[cur], carry + got, byte0, first_block, nblocks) != MODES_OK) {
            fprintf(stderr, "GPU demodulation failed: %s\n", modes_gpu_last_error(gpu[cur]));
            rc = 1;
            break;
        }
        if (pending && !finish(1 - cur)) { rc = 1; pending = false; break; }
        pending = true;
        // carry the last 476 bytes into the next batch's buffer (dump1090.c:481)
        if (!eof) {
            memcpy(buf[1 - cur], buf[cur] + carry + got - MODES_CARRY_BYTES, MODES_CARRY_BYTES);
            carry = MODES_CARRY_BYTES;
            first_block += nblocks;
            cur = 1 - cur;
        }
    }
    if (pending && rc == 0 && !finish(cur)) rc = 1;

    if (rc == 0 && opt.stats) {                                        // dump1090.c:2993-3006
        modes_host_stats st;
        modes_host_get_stats(host, &st);
        char text[512];
        modes_format_stats(&st, text);
        fputs(text, stdout);
    }
    modes_host_destroy(host);
    modes_tracker_destroy(sink.tracker);
    for (int k = 0; k < 2; k++) {
        modes_gpu_host_free(gpu[k], buf[k]);
        modes_gpu_destroy(gpu[k]);
    }
    if (fd > 0) close(fd);
    return rc;
}
