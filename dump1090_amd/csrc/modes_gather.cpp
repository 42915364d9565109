// modes_gather.cpp - libmodes_gather.so (include/modes_gather.h): the gather of the per-GPU record lists to rank 0 over
// RCCL, behind a C ABI.  One process per GPU; the only collective of the N-GPU path.
//
// Per call in flight (slot): the rank's list and its 8-byte length live in device buffers the demodulation kernels write
// (modes_gpu_set_output); the exchange is
//     counts   ncclAllGather of the lengths (8 bytes per rank) + copy to pinned host memory
//     records  when the lengths are on the host: every rank with records sends exactly n * 64 bytes, rank 0 receives each
//              list at its final offset of one contiguous device buffer (its own list is already at the front) - grouped
//              ncclSend / ncclRecv, 7 peers -> root over 7 distinct xGMI links, no padding, no staging copy - then ONE
//              device-to-host copy of the concatenation on rank 0.
// Everything is queued on the gather's own stream; the host blocks only in modes_gather_records (for the lengths) and in
// modes_gather_wait.  A group of one rank sends its list to itself into a second half of its buffer and compares.
//
// Line numbers cite /root/reference/dump1090.c where the reference has a counterpart; the exchange itself has none.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/modes_gather.h"

static_assert(sizeof(ncclUniqueId) == MODES_GATHER_ID_BYTES, "ncclUniqueId size");

namespace {

struct Slot {
    uint8_t *d_records = nullptr;              // root: room for every rank's list (two lists in a group of one); others: their own
    unsigned long long *d_count = nullptr;     // [2]: this rank's lengths - records (written by finalize_kernel), preamble positions
    unsigned long long *d_all = nullptr;       // [2 * nranks]
    unsigned long long *h_all = nullptr;       // pinned
    unsigned long long *d_cands = nullptr;     // the second list (cap_candidates > 0): root: every rank's; others: their own
    unsigned long long *h_cands = nullptr;     // pinned, root only: the concatenation
    unsigned long long *h_stage = nullptr;     // pinned: this rank's list on its way to d_cands ([cap_candidates]) and, behind it, its length
    std::vector<uint64_t> ccounts;             // candidate lengths of the call whose transfers were queued
    uint64_t ctotal = 0;
    bool cands_set = false;
    uint8_t *h_records = nullptr;              // pinned, root only
    hipEvent_t ev_c0 = nullptr, ev_counts = nullptr, ev_r0 = nullptr, ev_records = nullptr;
    std::vector<uint64_t> counts;              // lengths of the call whose records were queued
    uint64_t total = 0;
    bool counts_queued = false, records_queued = false, loopback = false;
};

thread_local char g_error[512] = "";

}  // namespace

struct modes_gather {
    modes_gather_config cfg{};
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    std::vector<Slot> slots;
    modes_gather_stats st{};
    std::string err;
    uint8_t *d_probe = nullptr;        // the first-contact probe's 128 bytes (modes_gather_create); released with the object
};

static int fail(modes_gather *g, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->err = buf;
    else { strncpy(g_error, buf, sizeof g_error - 1); g_error[sizeof g_error - 1] = 0; }
    return code;
}

#define HIP_TRY(g, call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return fail(g, MODES_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define NCCL_TRY(g, call)                                                                                \
    do {                                                                                                 \
        ncclResult_t r_ = (call);                                                                        \
        if (r_ != ncclSuccess) return fail(g, MODES_ERR_HIP, "%s: %s", #call, ncclGetErrorString(r_));  \
    } while (0)

extern "C" {

int modes_gather_abi_version(void) { return MODES_GATHER_ABI; }

const char *modes_gather_last_error(const modes_gather *g) { return g ? g->err.c_str() : g_error; }

int modes_gather_unique_id(void *id) {
    if (!id) return fail(nullptr, MODES_ERR_ARG, "unique_id: null");
    ncclUniqueId u;
    NCCL_TRY(nullptr, ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return MODES_OK;
}

int modes_gather_create(const modes_gather_config *cfg, const void *id, modes_gather **out) {
    if (!cfg || !id || !out) return fail(nullptr, MODES_ERR_ARG, "modes_gather_create: null argument");
    *out = nullptr;
    if (cfg->nranks < 1 || cfg->rank < 0 || cfg->rank >= cfg->nranks || cfg->cap_records == 0)
        return fail(nullptr, MODES_ERR_ARG, "modes_gather_create: rank %d of %d, %u records", cfg->rank, cfg->nranks, cfg->cap_records);
    modes_gather *g = new (std::nothrow) modes_gather;
    if (!g) return fail(nullptr, MODES_ERR_NOMEM, "out of memory");
    g->cfg = *cfg;
    if (g->cfg.nslots == 0) g->cfg.nslots = 3;
    auto bail = [&](int rc) { fail(nullptr, rc, "%s", g->err.c_str()); modes_gather_destroy(g); return rc; };
#define CREATE_HIP(call)                                                                                                  \
    do {                                                                                                                  \
        hipError_t e_ = (call);                                                                                           \
        if (e_ != hipSuccess) { fail(g, MODES_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); return bail(MODES_ERR_HIP); } \
    } while (0)
    CREATE_HIP(hipSetDevice(cfg->device));
    CREATE_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    {
        ncclResult_t r = ncclCommInitRank(&g->comm, cfg->nranks, u, cfg->rank);
        if (r != ncclSuccess) { fail(g, MODES_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", cfg->rank, cfg->nranks, ncclGetErrorString(r)); return bail(MODES_ERR_HIP); }
    }
    {   // the probe: a 64-byte ring over the new communicator, watched - a transfer that never completes is the failure mode of
        // a wrong IPC mode / a missing P2P path, and nothing else would ever report it
        double limit = 300.0;                                               // generous: cold boxes have taken 60 - 435 s for ncclCommInitRank alone (profiles/r08/e2e_ranks.txt);
                                                                            // a slow start must not be taken for a hang - the other IPC mode does not work on this pool
        if (const char *v = getenv("MODES_GATHER_PROBE_SECONDS")) limit = atof(v);
        if (limit > 0.0) {
            CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&g->d_probe), 128));     // (owned by g: every way out through bail() frees it)
            uint8_t *const d_probe = g->d_probe;
            CREATE_HIP(hipMemsetAsync(d_probe, 0x5a, 64, g->stream));
            const int next = (cfg->rank + 1) % cfg->nranks, prev = (cfg->rank + cfg->nranks - 1) % cfg->nranks;
            ncclResult_t r = ncclGroupStart();
            if (r == ncclSuccess) r = ncclRecv(d_probe + 64, 64, ncclUint8, prev, g->comm, g->stream);
            if (r == ncclSuccess) r = ncclSend(d_probe, 64, ncclUint8, next, g->comm, g->stream);
            const ncclResult_t e = ncclGroupEnd();
            if (r == ncclSuccess) r = e;
            hipError_t q = hipErrorNotReady;
            const auto t0 = std::chrono::steady_clock::now();
            while (r == ncclSuccess) {
                q = hipStreamQuery(g->stream);
                if (q != hipErrorNotReady) break;
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) break;
                usleep(500);
            }
            const char *ipc = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
            if (r != ncclSuccess || q != hipSuccess) {
                fail(g, MODES_GATHER_ERR_PROBE, "probe: the first 64-byte ncclSend/ncclRecv ring of rank %d of %d %s (%s; HSA_ENABLE_IPC_MODE_LEGACY=%s)",
                     cfg->rank, cfg->nranks, r != ncclSuccess ? "was refused" : q == hipErrorNotReady ? "did not complete in time" : "failed",
                     r != ncclSuccess ? ncclGetErrorString(r) : hipGetErrorString(q), ipc ? ipc : "unset");
                // (no teardown - g, its stream, the communicator and the probe's 128 bytes stay: a communicator with a transfer stuck
                //  in it cannot be destroyed.  The caller must not create another communicator and has to end the process:
                //  modes_gather.h says so.)
                fail(nullptr, MODES_GATHER_ERR_PROBE, "%s", g->err.c_str());
                return MODES_GATHER_ERR_PROBE;
            }
            uint8_t back[64];
            CREATE_HIP(hipMemcpy(back, d_probe + 64, 64, hipMemcpyDeviceToHost));
            (void)hipFree(d_probe);
            g->d_probe = nullptr;
            for (uint8_t b : back)
                if (b != 0x5a) { fail(g, MODES_GATHER_ERR_PROBE, "probe: the ring delivered other bytes than were sent"); return bail(MODES_GATHER_ERR_PROBE); }
        }
    }
    const bool root = cfg->rank == 0;
    const size_t own = (size_t)cfg->cap_records * sizeof(modes_record);
    const size_t lists = root ? (size_t)(cfg->nranks > 1 ? cfg->nranks : 2) : 1;
    g->slots.resize(g->cfg.nslots);
    for (Slot &s : g->slots) {
        CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&s.d_records), own * lists));
        CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&s.d_count), 2 * sizeof(unsigned long long)));
        CREATE_HIP(hipMemset(s.d_count, 0, 2 * sizeof(unsigned long long)));
        CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&s.d_all), 2 * sizeof(unsigned long long) * (size_t)cfg->nranks));
        CREATE_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_all), 2 * sizeof(unsigned long long) * (size_t)cfg->nranks, hipHostMallocDefault));
        if (cfg->cap_candidates) {
            const size_t cown = (size_t)cfg->cap_candidates * sizeof(unsigned long long);
            CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&s.d_cands), cown * (root ? (size_t)cfg->nranks : 1)));
            if (root) CREATE_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_cands), cown * (size_t)cfg->nranks, hipHostMallocDefault));
            CREATE_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_stage), cown + sizeof(unsigned long long), hipHostMallocDefault));
        }
        s.ccounts.assign((size_t)cfg->nranks, 0);
        if (root) CREATE_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_records), own * lists, hipHostMallocDefault));
        CREATE_HIP(hipEventCreate(&s.ev_c0));
        CREATE_HIP(hipEventCreate(&s.ev_counts));
        CREATE_HIP(hipEventCreate(&s.ev_r0));
        CREATE_HIP(hipEventCreate(&s.ev_records));
        s.counts.assign((size_t)cfg->nranks, 0);
    }
#undef CREATE_HIP
    g->st.nranks = cfg->nranks;
    g->st.rank = cfg->rank;
    (void)ncclGetVersion(&g->st.rccl_version);
    *out = g;
    return MODES_OK;
}

void modes_gather_destroy(modes_gather *g) {
    if (!g) return;
    (void)hipSetDevice(g->cfg.device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    for (Slot &s : g->slots) {
        if (s.d_records) (void)hipFree(s.d_records);
        if (s.d_count) (void)hipFree(s.d_count);
        if (s.d_all) (void)hipFree(s.d_all);
        if (s.h_all) (void)hipHostFree(s.h_all);
        if (s.h_records) (void)hipHostFree(s.h_records);
        if (s.d_cands) (void)hipFree(s.d_cands);
        if (s.h_cands) (void)hipHostFree(s.h_cands);
        if (s.h_stage) (void)hipHostFree(s.h_stage);
        for (hipEvent_t e : {s.ev_c0, s.ev_counts, s.ev_r0, s.ev_records})
            if (e) (void)hipEventDestroy(e);
    }
    if (g->d_probe) (void)hipFree(g->d_probe);
    if (g->comm) (void)ncclCommDestroy(g->comm);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

static Slot *slot_of(modes_gather *g, uint32_t slot) { return g && slot < g->slots.size() ? &g->slots[slot] : nullptr; }

int modes_gather_output(modes_gather *g, uint32_t slot, void **d_records, uint64_t *capacity, void **d_count) {
    Slot *s = slot_of(g, slot);
    if (!s || !d_records || !capacity || !d_count) return g ? fail(g, MODES_ERR_ARG, "output: bad slot or null argument") : MODES_ERR_ARG;
    *d_records = s->d_records;
    *capacity = g->cfg.cap_records;
    *d_count = s->d_count;
    return MODES_OK;
}

int modes_gather_set_empty(modes_gather *g, uint32_t slot) {
    Slot *s = slot_of(g, slot);
    if (!s) return g ? fail(g, MODES_ERR_ARG, "set_empty: slot %u of %zu", slot, g->slots.size()) : MODES_ERR_ARG;
    HIP_TRY(g, hipSetDevice(g->cfg.device));
    HIP_TRY(g, hipMemsetAsync(s->d_count, 0, sizeof(unsigned long long), g->stream));      // (the second length is reset by every counts)
    return MODES_OK;
}

int modes_gather_set_candidates(modes_gather *g, uint32_t slot, const uint64_t *candidates, uint64_t n) {
    Slot *s = slot_of(g, slot);
    if (!s) return g ? fail(g, MODES_ERR_ARG, "set_candidates: slot %u of %zu", slot, g->slots.size()) : MODES_ERR_ARG;
    if (!g->cfg.cap_candidates) return fail(g, MODES_ERR_ARG, "set_candidates: the communicator was made without a second list (cap_candidates = 0)");
    if (n && !candidates) return fail(g, MODES_ERR_ARG, "set_candidates: null list");
    if (s->counts_queued) return fail(g, MODES_ERR_STATE, "set_candidates: slot %u has an exchange in flight", slot);
    HIP_TRY(g, hipSetDevice(g->cfg.device));
    // (the staging buffer is free: the slot's previous exchange - which its H2D copy preceded on the gather's stream - has been
    // waited for, or this call would have returned MODES_ERR_STATE)
    unsigned long long *h_n = s->h_stage + g->cfg.cap_candidates;
    *h_n = n;                                                                // the TRUE length travels, whatever fits (every rank then fails together)
    const uint64_t fit = n < g->cfg.cap_candidates ? n : g->cfg.cap_candidates;
    if (fit) {
        memcpy(s->h_stage, candidates, (size_t)fit * sizeof(uint64_t));
        HIP_TRY(g, hipMemcpyAsync(s->d_cands, s->h_stage, (size_t)fit * sizeof(uint64_t), hipMemcpyHostToDevice, g->stream));
    }
    HIP_TRY(g, hipMemcpyAsync(s->d_count + 1, h_n, sizeof(unsigned long long), hipMemcpyHostToDevice, g->stream));
    s->cands_set = true;
    return MODES_OK;
}

int modes_gather_counts(modes_gather *g, uint32_t slot) {
    Slot *s = slot_of(g, slot);
    if (!s) return g ? fail(g, MODES_ERR_ARG, "counts: slot %u of %zu", slot, g->slots.size()) : MODES_ERR_ARG;
    if (s->counts_queued) return fail(g, MODES_ERR_STATE, "counts: slot %u already has an exchange in flight (wait for it first)", slot);
    HIP_TRY(g, hipSetDevice(g->cfg.device));
    HIP_TRY(g, hipEventRecord(s->ev_c0, g->stream));
    if (!s->cands_set) HIP_TRY(g, hipMemsetAsync(s->d_count + 1, 0, sizeof(unsigned long long), g->stream));   // no second list in this round
    s->cands_set = false;
    NCCL_TRY(g, ncclAllGather(s->d_count, s->d_all, 2, ncclUint64, g->comm, g->stream));                       // (records, preamble positions) per rank
    HIP_TRY(g, hipMemcpyAsync(s->h_all, s->d_all, 2 * sizeof(unsigned long long) * (size_t)g->cfg.nranks, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(g, hipEventRecord(s->ev_counts, g->stream));
    s->counts_queued = true;
    s->records_queued = false;
    return MODES_OK;
}

int modes_gather_records(modes_gather *g, uint32_t slot) {
    Slot *s = slot_of(g, slot);
    if (!s) return g ? fail(g, MODES_ERR_ARG, "records: slot %u of %zu", slot, g->slots.size()) : MODES_ERR_ARG;
    if (!s->counts_queued || s->records_queued) return fail(g, MODES_ERR_STATE, "records: call modes_gather_counts for slot %u first", slot);
    HIP_TRY(g, hipSetDevice(g->cfg.device));
    HIP_TRY(g, hipEventSynchronize(s->ev_counts));                           // the step's one synchronisation
    const int n = g->cfg.nranks, me = g->cfg.rank;
    s->total = s->ctotal = 0;
    for (int r = 0; r < n; r++) {
        const unsigned long long nrec = s->h_all[2 * r], ncand = s->h_all[2 * r + 1];
        s->counts[(size_t)r] = nrec;
        s->ccounts[(size_t)r] = ncand;
        s->total += nrec;
        s->ctotal += ncand;
        if (nrec > g->cfg.cap_records) {                                     // every rank sees the same lengths: all fail together
            s->counts_queued = false;
            return fail(g, MODES_ERR_OVERFLOW, "rank %d produced %llu records, the gather buffers hold %u per rank", r, nrec, g->cfg.cap_records);
        }
        if (ncand > g->cfg.cap_candidates) {
            s->counts_queued = false;
            return fail(g, MODES_ERR_OVERFLOW, "rank %d found %llu preamble positions, the gather buffers hold %u per rank", r, ncand, g->cfg.cap_candidates);
        }
    }
    const size_t rec = sizeof(modes_record);
    HIP_TRY(g, hipEventRecord(s->ev_r0, g->stream));
    s->loopback = false;
    uint64_t ops = 0, rx = 0, tx = 0;
    NCCL_TRY(g, ncclGroupStart());
    struct GroupGuard {                                                      // an error return between start and end must not leave the group open
        bool open = true;
        ~GroupGuard() { if (open) (void)ncclGroupEnd(); }
    } group;
    if (me == 0) {
        // rank order = stream order: the root's own list already sits at the front, every other list lands behind its predecessor's
        size_t off = (size_t)s->counts[0] * rec;
        for (int r = 1; r < n; r++) {
            const size_t nb = (size_t)s->counts[(size_t)r] * rec;
            if (nb == 0) continue;
            NCCL_TRY(g, ncclRecv(s->d_records + off, nb, ncclUint8, r, g->comm, g->stream));
            off += nb;
            rx += nb;
            ops++;
        }
        size_t coff = (size_t)s->ccounts[0];                                 // the second list: the same pattern, 8-byte elements
        for (int r = 1; r < n; r++) {
            const size_t ne = (size_t)s->ccounts[(size_t)r];
            if (ne == 0) continue;
            NCCL_TRY(g, ncclRecv(s->d_cands + coff, ne, ncclUint64, r, g->comm, g->stream));
            coff += ne;
            rx += ne * sizeof(uint64_t);
            ops++;
        }
        if (n == 1 && s->counts[0]) {                                        // loopback: the same calls on the one GPU there is
            const size_t nb = (size_t)s->counts[0] * rec, half = (size_t)g->cfg.cap_records * rec;
            NCCL_TRY(g, ncclRecv(s->d_records + half, nb, ncclUint8, 0, g->comm, g->stream));
            NCCL_TRY(g, ncclSend(s->d_records, nb, ncclUint8, 0, g->comm, g->stream));
            rx += nb;
            tx += nb;
            ops += 2;
            s->loopback = true;
        }
    } else {
        if (s->counts[(size_t)me]) {
            const size_t nb = (size_t)s->counts[(size_t)me] * rec;
            NCCL_TRY(g, ncclSend(s->d_records, nb, ncclUint8, 0, g->comm, g->stream));
            tx += nb;
            ops++;
        }
        if (s->ccounts[(size_t)me]) {
            NCCL_TRY(g, ncclSend(s->d_cands, (size_t)s->ccounts[(size_t)me], ncclUint64, 0, g->comm, g->stream));
            tx += (size_t)s->ccounts[(size_t)me] * sizeof(uint64_t);
            ops++;
        }
    }
    group.open = false;
    NCCL_TRY(g, ncclGroupEnd());
    if (me == 0 && s->total) {
        HIP_TRY(g, hipMemcpyAsync(s->h_records, s->d_records, (size_t)s->total * rec, hipMemcpyDeviceToHost, g->stream));
        if (s->loopback) {
            const size_t half = (size_t)g->cfg.cap_records * rec;
            HIP_TRY(g, hipMemcpyAsync(s->h_records + half, s->d_records + half, (size_t)s->counts[0] * rec, hipMemcpyDeviceToHost, g->stream));
        }
    }
    if (me == 0 && s->ctotal)
        HIP_TRY(g, hipMemcpyAsync(s->h_cands, s->d_cands, (size_t)s->ctotal * sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(g, hipEventRecord(s->ev_records, g->stream));
    s->records_queued = true;
    g->st.calls++;
    g->st.p2p_ops += ops;
    g->st.bytes_received += rx;
    g->st.bytes_sent += tx;
    return MODES_OK;
}

int modes_gather_wait(modes_gather *g, uint32_t slot, const modes_record **records, uint64_t *n_records, const uint64_t **counts) {
    Slot *s = slot_of(g, slot);
    if (!s) return g ? fail(g, MODES_ERR_ARG, "wait: slot %u of %zu", slot, g->slots.size()) : MODES_ERR_ARG;
    if (!s->records_queued) return fail(g, MODES_ERR_STATE, "wait: no exchange queued for slot %u", slot);
    HIP_TRY(g, hipSetDevice(g->cfg.device));
    HIP_TRY(g, hipEventSynchronize(s->ev_records));
    float a = 0.f, b = 0.f;
    (void)hipEventElapsedTime(&a, s->ev_c0, s->ev_counts);
    (void)hipEventElapsedTime(&b, s->ev_r0, s->ev_records);
    g->st.gather_ms += a + b;
    s->counts_queued = s->records_queued = false;
    if (s->loopback) {
        const size_t half = (size_t)g->cfg.cap_records * sizeof(modes_record);
        if (memcmp(s->h_records, s->h_records + half, (size_t)s->counts[0] * sizeof(modes_record)) != 0)
            return fail(g, MODES_ERR_HIP, "loopback: the record list that went through ncclSend / ncclRecv differs from the one sent");
    }
    const bool root = g->cfg.rank == 0;
    if (records) *records = root && s->total ? reinterpret_cast<const modes_record *>(s->h_records) : nullptr;
    if (n_records) *n_records = root ? s->total : 0;
    if (counts) *counts = s->counts.data();
    return MODES_OK;
}

int modes_gather_candidates(modes_gather *g, uint32_t slot, const uint64_t **candidates, uint64_t *n) {
    Slot *s = slot_of(g, slot);
    if (!s || !candidates || !n) return g ? fail(g, MODES_ERR_ARG, "candidates: bad slot or null argument") : MODES_ERR_ARG;
    if (s->counts_queued || s->records_queued) return fail(g, MODES_ERR_STATE, "candidates: wait for slot %u first", slot);
    const bool root = g->cfg.rank == 0;
    *candidates = root && s->ctotal ? reinterpret_cast<const uint64_t *>(s->h_cands) : nullptr;
    *n = root ? s->ctotal : 0;
    return MODES_OK;
}

int modes_gather_get_stats(const modes_gather *g, modes_gather_stats *out) {
    if (!g || !out) return MODES_ERR_ARG;
    *out = g->st;
    return MODES_OK;
}

}  // extern "C"
