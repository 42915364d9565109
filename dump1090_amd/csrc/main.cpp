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
// This file: the option parser.  The hosts: host_single.cpp (one process, N devices), host_ranks.cpp + host_ranks_rccl.cpp /
// host_ranks_shared.cpp (one process per GPU); host_common.h has what they share.
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

#include "host_common.h"

using namespace modes_cli;

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

    return run_single(opt, t_start);
}
