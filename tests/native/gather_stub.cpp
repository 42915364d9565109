// gather_stub.cpp - TEST SCAFFOLDING: include/modes_gather.h implemented over POSIX shared memory instead of RCCL, so that
// the one-process-per-GPU mode of the C host (dump1090_amd --ranks N: the fork, the id pipes, the round-robin batches, the
// rounds of the gather, the EOF batch, ranks that run out of batches) can run with N = 2, 3 on a machine without GPUs -
// together with tests/native/gpu_stub.cpp, which writes its "device" lists straight into this stub's buffers.  Built as
// libmodes_gather.so next to the test binary (tools/sanitize_host.sh ranks-host); never part of the product.
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/modes_gather.h"

namespace {
struct Shared {                      // at the front of the segment
    pthread_barrier_t barrier;
    int ready;
};
}  // namespace

struct modes_gather {
    modes_gather_config cfg{};
    std::string name, err;
    size_t bytes = 0;
    uint8_t *base = nullptr;
    bool set_flag = false;                        // set_candidates was called since the last counts
    std::vector<std::vector<modes_record>> out;   // root, per slot: the concatenation of the round being waited for
    std::vector<std::vector<uint64_t>> counts;   // per slot
    std::vector<std::vector<uint64_t>> cands;    // root, per slot: the second list (preamble positions) of the round being waited for
    modes_gather_stats st{};
    Shared *sh() const { return reinterpret_cast<Shared *>(base); }
    // per (slot, rank): an 8-byte length (+ the second list's at byte 8), cap_records records, cap_candidates positions
    size_t stride() const { return 64 + (size_t)cfg.cap_records * sizeof(modes_record) + (size_t)cfg.cap_candidates * 8; }
    uint8_t *cand_cell(uint32_t slot, int rank) const { return cell(slot, rank) + 64 + (size_t)cfg.cap_records * sizeof(modes_record); }
    uint8_t *cell(uint32_t slot, int rank) const { return base + 4096 + ((size_t)slot * (size_t)cfg.nranks + (size_t)rank) * stride(); }
};

extern "C" {

int modes_gather_abi_version(void) { return MODES_GATHER_ABI; }
const char *modes_gather_last_error(const modes_gather *g) { return g ? g->err.c_str() : "gather stub"; }

int modes_gather_unique_id(void *id) {
    memset(id, 0, MODES_GATHER_ID_BYTES);
    snprintf(static_cast<char *>(id), MODES_GATHER_ID_BYTES, "/modes_gather_stub_%d_%ld", (int)getpid(), random());
    return MODES_OK;
}

int modes_gather_create(const modes_gather_config *cfg, const void *id, modes_gather **out) {
    // MODES_STUB_PROBE_FAIL=<ranks, comma separated or "all">: the first transfer over the "communicator" never completes on these
    // ranks - unless this is the host's second try (MODES_IPC_RETRIED), which succeeds: the restart path of dump1090_amd --ranks
    if (const char *f = getenv("MODES_STUB_PROBE_FAIL"))
        if (!getenv("MODES_IPC_RETRIED")) {
            char want[16];
            snprintf(want, sizeof want, "%d", cfg->rank);
            if (!strcmp(f, "all") || strstr(f, want)) { usleep(100 * 1000); return MODES_GATHER_ERR_PROBE; }
            pause();                                                         // the others would wait in the rendezvous for ever
        }
    modes_gather *g = new modes_gather;
    g->cfg = *cfg;
    if (g->cfg.nslots == 0) g->cfg.nslots = 3;
    g->name = static_cast<const char *>(id);
    g->bytes = 4096 + (size_t)g->cfg.nslots * (size_t)cfg->nranks * g->stride();
    int fd = -1;
    if (cfg->rank == 0) {
        fd = shm_open(g->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)g->bytes) != 0) { perror("gather stub: shm"); return MODES_ERR_HIP; }
    } else {
        for (int tries = 0; tries < 20000 && fd < 0; tries++) { fd = shm_open(g->name.c_str(), O_RDWR, 0600); if (fd < 0) usleep(500); }
        if (fd < 0) { perror("gather stub: shm_open"); return MODES_ERR_HIP; }
        for (int tries = 0; tries < 20000; tries++) {                         // the root sizes the segment before anyone maps it
            off_t len = lseek(fd, 0, SEEK_END);
            if (len >= (off_t)g->bytes) break;
            usleep(500);
        }
    }
    g->base = static_cast<uint8_t *>(mmap(nullptr, g->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    close(fd);
    if (g->base == MAP_FAILED) { perror("gather stub: mmap"); return MODES_ERR_HIP; }
    if (cfg->rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&g->sh()->barrier, &a, (unsigned)cfg->nranks);
        __atomic_store_n(&g->sh()->ready, 1, __ATOMIC_RELEASE);
    } else {
        while (!__atomic_load_n(&g->sh()->ready, __ATOMIC_ACQUIRE)) usleep(200);
    }
    pthread_barrier_wait(&g->sh()->barrier);                                 // collective, like ncclCommInitRank
    if (cfg->rank == 0) shm_unlink(g->name.c_str());
    g->counts.assign(g->cfg.nslots, std::vector<uint64_t>((size_t)cfg->nranks, 0));
    g->out.resize(g->cfg.nslots);
    g->cands.resize(g->cfg.nslots);
    g->st.nranks = cfg->nranks;
    g->st.rank = cfg->rank;
    g->st.rccl_version = 1;
    *out = g;
    return MODES_OK;
}

void modes_gather_destroy(modes_gather *g) {
    if (!g) return;
    if (g->base && g->base != MAP_FAILED) munmap(g->base, g->bytes);
    delete g;
}

int modes_gather_output(modes_gather *g, uint32_t slot, void **d_records, uint64_t *capacity, void **d_count) {
    if (slot >= g->cfg.nslots) return MODES_ERR_ARG;
    uint8_t *c = g->cell(slot, g->cfg.rank);
    *d_count = c;                                                            // the "device" buffers are this rank's cell of the segment
    *d_records = c + 64;
    *capacity = g->cfg.cap_records;
    return MODES_OK;
}

int modes_gather_set_empty(modes_gather *g, uint32_t slot) {
    memset(g->cell(slot, g->cfg.rank), 0, 8);
    return MODES_OK;
}

int modes_gather_set_candidates(modes_gather *g, uint32_t slot, const uint64_t *candidates, uint64_t n) {
    if (slot >= g->cfg.nslots || !g->cfg.cap_candidates) { g->err = "set_candidates: no second list"; return MODES_ERR_ARG; }
    uint8_t *c = g->cell(slot, g->cfg.rank);
    memcpy(c + 8, &n, 8);                                                    // the true length, whatever fits
    const uint64_t fit = n < g->cfg.cap_candidates ? n : g->cfg.cap_candidates;
    if (fit) memcpy(g->cand_cell(slot, g->cfg.rank), candidates, fit * 8);
    g->set_flag = true;
    return MODES_OK;
}

int modes_gather_counts(modes_gather *g, uint32_t slot) {
    if (!g->set_flag) memset(g->cell(slot, g->cfg.rank) + 8, 0, 8);         // no second list in this round
    g->set_flag = false;
    pthread_barrier_wait(&g->sh()->barrier);                                 // every rank's list and length are in the segment
    return MODES_OK;
}

int modes_gather_records(modes_gather *g, uint32_t slot) {
    uint64_t total = 0;
    std::vector<uint64_t> &cnt = g->counts[slot];
    std::vector<uint64_t> ccnt((size_t)g->cfg.nranks, 0);
    for (int r = 0; r < g->cfg.nranks; r++) {
        memcpy(&cnt[(size_t)r], g->cell(slot, r), 8);
        memcpy(&ccnt[(size_t)r], g->cell(slot, r) + 8, 8);
        if (cnt[(size_t)r] > g->cfg.cap_records || ccnt[(size_t)r] > g->cfg.cap_candidates) { g->err = "a list exceeds the gather buffers"; return MODES_ERR_OVERFLOW; }
        total += cnt[(size_t)r];
    }
    if (g->cfg.rank == 0) {
        std::vector<modes_record> &out = g->out[slot];
        out.resize(total);
        size_t at = 0;
        for (int r = 0; r < g->cfg.nranks; r++) {
            memcpy(out.data() + at, g->cell(slot, r) + 64, cnt[(size_t)r] * sizeof(modes_record));
            at += cnt[(size_t)r];
            if (r) { g->st.p2p_ops += cnt[(size_t)r] ? 1 : 0; g->st.bytes_received += cnt[(size_t)r] * sizeof(modes_record); }
        }
        std::vector<uint64_t> &cs = g->cands[slot];
        cs.clear();
        for (int r = 0; r < g->cfg.nranks; r++) {
            const uint64_t *p = reinterpret_cast<const uint64_t *>(g->cand_cell(slot, r));
            cs.insert(cs.end(), p, p + ccnt[(size_t)r]);
        }
    }
    g->st.calls++;
    pthread_barrier_wait(&g->sh()->barrier);                                 // the root has taken the lists: the cells may be reused
    return MODES_OK;
}

int modes_gather_wait(modes_gather *g, uint32_t slot, const modes_record **records, uint64_t *n_records, const uint64_t **counts) {
    const bool root = g->cfg.rank == 0;
    if (records) *records = root && !g->out[slot].empty() ? g->out[slot].data() : nullptr;
    if (n_records) *n_records = root ? g->out[slot].size() : 0;
    if (counts) *counts = g->counts[slot].data();
    return MODES_OK;
}

int modes_gather_candidates(modes_gather *g, uint32_t slot, const uint64_t **candidates, uint64_t *n) {
    const bool root = g->cfg.rank == 0;
    *candidates = root && !g->cands[slot].empty() ? g->cands[slot].data() : nullptr;
    *n = root ? g->cands[slot].size() : 0;
    return MODES_OK;
}

int modes_gather_get_stats(const modes_gather *g, modes_gather_stats *out) { *out = g->st; return MODES_OK; }

}  // extern "C"
