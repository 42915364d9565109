// gpu_stub.cpp - TEST SCAFFOLDING for tools/sanitize_host.sh tsan-host: the entry points of include/modes_gfx950.h that
// dump1090_amd/csrc/main.cpp calls, implemented on the CPU with the ORACLE's stateless functions (oracle/modes_oracle.c),
// so that the C host - reader thread, resolver thread, lanes handed from one to the other - can run under ThreadSanitizer
// on a machine without a GPU.  Linked into a test binary only; the product library has no such path (no CPU fallback).
//
// "Asynchronous" like the real thing: modes_gpu_submit_host only remembers the host buffer; the records are computed in
// modes_gpu_fetch - on the resolver thread, while the reader thread is already filling and submitting the next lanes.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/modes_gfx950.h"
#include "../../oracle/modes_oracle.h"

struct modes_gpu {
    modes_gpu_config cfg{};
    const uint8_t *iq = nullptr;
    uint64_t nbytes = 0, byte0 = 0, first_block = 0, nblocks = 0;
    bool in_flight = false;
    std::vector<modes_record> recs;
    std::vector<uint64_t> cands;
    std::string err;
    modes_record *out_records = nullptr;         // modes_gpu_set_output: the caller's "device" list and its length
    uint64_t out_cap = 0;
    unsigned long long *out_count = nullptr;
};

extern "C" {

int modes_gpu_abi_version(void) { return MODES_GFX950_ABI; }
const char *modes_gpu_last_error(const modes_gpu *g) { return g ? g->err.c_str() : "stub"; }
int modes_gpu_create(const modes_gpu_config *cfg, modes_gpu **out) {
    // MODES_STUB_FAIL_DEVICE=<d>: the "GPU" d does not come up (how the host treats a rank that fails while its peers wait)
    if (const char *f = getenv("MODES_STUB_FAIL_DEVICE"))
        if (atoi(f) == cfg->device) return MODES_ERR_HIP;
    *out = new modes_gpu;
    (*out)->cfg = *cfg;
    return MODES_OK;
}
void modes_gpu_destroy(modes_gpu *g) { delete g; }
int modes_gpu_set_timing(modes_gpu *, int) { return MODES_OK; }
int modes_gpu_host_alloc(modes_gpu *, size_t n, void **out) { *out = new uint8_t[n ? n : 1]; return MODES_OK; }
void modes_gpu_host_free(modes_gpu *, void *p) { delete[] static_cast<uint8_t *>(p); }

int modes_gpu_submit_host(modes_gpu *g, const uint8_t *iq, uint64_t nbytes, uint64_t byte0, uint64_t first_block, uint64_t nblocks) {
    if (g->in_flight) { g->err = "submit_host: a detect is already in flight"; return MODES_ERR_STATE; }
    // MODES_STUB_FAIL_SUBMIT=<d>:<k>: the k-th submit (from 0) of this process on "GPU" d fails - a rank that fails MID-STREAM,
    // with its peers already inside the next round's collective
    if (const char *f = getenv("MODES_STUB_FAIL_SUBMIT")) {
        static int submits = 0;
        int d = -1, k = -1;
        if (sscanf(f, "%d:%d", &d, &k) == 2 && d == g->cfg.device && submits++ == k) { g->err = "stub: submit fails on request"; return MODES_ERR_HIP; }
    }
    g->iq = iq; g->nbytes = nbytes; g->byte0 = byte0; g->first_block = first_block; g->nblocks = nblocks;
    g->in_flight = true;
    return MODES_OK;
}

// buffer k of the stream from the bytes this call holds: stream bytes [262144 k - 476, 262144 (k + 1)), 127 outside
static void frame(const modes_gpu *g, uint64_t k, uint8_t *out) {
    const int64_t lo = (int64_t)(k * ORC_DATA_LEN) - (int64_t)ORC_CARRY_BYTES;
    for (uint32_t i = 0; i < ORC_BLOCK_BYTES; i++) {
        const int64_t s = lo + i - (int64_t)g->byte0;
        out[i] = (s >= 0 && (uint64_t)s < g->nbytes) ? g->iq[s] : 127;
    }
}

int modes_gpu_fetch(modes_gpu *g, modes_gpu_result *res) {
    if (!g->in_flight) { g->err = "fetch: no detect in flight"; return MODES_ERR_STATE; }
    g->in_flight = false;
    g->recs.clear();
    g->cands.clear();
    const int maxfix = g->cfg.fix_errors ? (g->cfg.aggressive ? 2 : 1) : 0;
    std::vector<uint8_t> bytes(ORC_BLOCK_BYTES);
    std::vector<uint16_t> mag(ORC_BLOCK_SAMPLES);
    std::vector<uint32_t> js(ORC_SCAN_POSITIONS);
    for (uint64_t k = g->first_block; k < g->first_block + g->nblocks; k++) {
        frame(g, k, bytes.data());
        orc_magnitude(bytes.data(), ORC_BLOCK_SAMPLES, mag.data());
        const size_t n = orc_block_candidates(mag.data(), ORC_BLOCK_SAMPLES, js.data(), js.size());
        for (size_t c = 0; c < n; c++) {
            if (g->cfg.keep_candidates) g->cands.push_back(k * MODES_BLOCK_STRIDE + js[c]);
            orc_record o;
            orc_record_at(mag.data(), js[c], maxfix, &o);
            if (!o.att[0].gate_ok) continue;
            modes_record r;
            memset(&r, 0, sizeof r);
            r.block = (uint32_t)k;
            r.j = o.j;
            static_assert(sizeof(orc_attempt) == sizeof(modes_attempt), "attempt layout");
            memcpy(&r.att[0], &o.att[0], sizeof(modes_attempt));
            memcpy(&r.att[1], &o.att[1], sizeof(modes_attempt));
            if (!r.att[1].gate_ok) { r.att[1].syndrome = 0; r.att[1].nfix = 0; r.att[1].fixpos[0] = r.att[1].fixpos[1] = 0xff; }
            g->recs.push_back(r);
        }
    }
    memset(res, 0, sizeof *res);
    res->records = g->recs.data();
    res->n_records = g->recs.size();
    res->candidates = g->cands.empty() ? nullptr : g->cands.data();
    res->n_candidates = g->cands.size();
    res->n_preambles = g->cands.size();
    return MODES_OK;
}

int modes_gpu_demod_host(modes_gpu *g, const uint8_t *iq, uint64_t nbytes, uint64_t byte0, uint64_t first_block, uint64_t nblocks,
                         modes_gpu_result *res) {
    const int rc = modes_gpu_submit_host(g, iq, nbytes, byte0, first_block, nblocks);
    return rc != MODES_OK ? rc : modes_gpu_fetch(g, res);
}

// --ranks: the caller's list and length (with tests/native/gather_stub.cpp: cells of a shared-memory segment)
int modes_gpu_set_output(modes_gpu *g, void *d_records, uint64_t capacity, void *d_count) {
    g->out_records = static_cast<modes_record *>(d_records);
    g->out_cap = capacity;
    g->out_count = static_cast<unsigned long long *>(d_count);
    return MODES_OK;
}
int modes_gpu_fetch_device(modes_gpu *g, modes_gpu_result *res) {
    const int rc = modes_gpu_fetch(g, res);
    if (rc != MODES_OK) return rc;
    if (g->out_count) *g->out_count = g->recs.size();                      // the true length, whatever fits
    if (g->recs.size() > g->out_cap) { g->err = "records exceed max_records"; return MODES_ERR_OVERFLOW; }
    if (g->out_records) memcpy(g->out_records, g->recs.data(), g->recs.size() * sizeof(modes_record));
    res->records = g->out_records;
    return MODES_OK;
}

}  // extern "C"
