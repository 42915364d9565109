/* modes_gather.h - C ABI of libmodes_gather.so: the ONE exchange of the N-GPU path, the gather of the per-GPU record
 * lists to rank 0 over RCCL (xGMI), for hosts that run one process per GPU.
 *
 * The reference is one process on one CPU core; its main loop (dump1090.c:2965-2990) hands one 256 KiB buffer at a time
 * to detectModeS().  Widened to N GPUs the stream shards by whole buffers - the only overlap is the 476-byte carry each
 * rank reads from the input itself (dump1090.c:481), so there is no data-path collective - and the one piece of
 * cross-buffer state, the ICAO whitelist (dump1090.c:896-925), lives on rank 0, which therefore needs every rank's
 * records, in stream order.  This library is that step behind a C boundary:
 *
 *     rank r, per GPU call                                     rank 0 in addition
 *     -------------------------------------------------------  ---------------------------------------------
 *     modes_gather_output(g, slot, &d_rec, &cap, &d_cnt)
 *     modes_gpu_set_output(gpu, d_rec, cap, d_cnt)              (the kernels write list + length there)
 *     modes_gpu_detect / modes_gpu_submit_host ... modes_gpu_fetch_device(gpu, &res)
 *     modes_gather_counts(g, slot)      all-gather of the 8-byte lengths, asynchronous
 *     modes_gather_records(g, slot)     exact-size send (rank 0: receives, each list at its final offset)
 *     modes_gather_wait(g, slot, ...)                           the concatenation in RANK ORDER, in pinned host memory
 *
 * Rank order is stream order when rank r demodulates batch (round * N + r) - what dump1090_amd --ranks N does - or when
 * the ranks own contiguous buffer ranges (bench.py).  Several slots let a host keep several calls in flight.  Every rank
 * must issue its modes_gather_counts / modes_gather_records calls in the same order (RCCL executes a communicator's
 * operations in issue order).  A group of ONE rank sends its list to itself (loopback): the same RCCL calls on one GPU.
 *
 * Plain C, like modes_gfx950.h: every function returns 0 or a negative MODES_ERR_*; modes_gather_last_error() has the text.
 * dump1090_amd/distributed.py is the same exchange for Python hosts (torch.distributed); both speak to RCCL directly,
 * neither goes through the other.
 */
#ifndef MODES_GATHER_H
#define MODES_GATHER_H

#include <stddef.h>
#include <stdint.h>

#include "modes_gfx950.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MODES_GATHER_ID_BYTES 128          /* an ncclUniqueId: made by rank 0, handed to every rank by the host (pipe, file, ...) */

typedef struct modes_gather modes_gather;

typedef struct {
    int32_t  device;        /* HIP device of this rank (one rank per device: RCCL refuses two ranks on one GPU)        */
    int32_t  rank;          /* 0 .. nranks-1; rank 0 is the root: it owns the first buffers and the whitelist          */
    int32_t  nranks;
    uint32_t cap_records;   /* records per rank and call the buffers hold (more: MODES_ERR_OVERFLOW on EVERY rank)     */
    uint32_t nslots;        /* calls in flight (sets of buffers); 0 = 3                                                */
    uint32_t cap_candidates;/* --stats hosts: preamble positions per rank and call the buffers of the SECOND list hold
                               (modes_gather_set_candidates); 0 = no second list (ABI 1's `reserved`)                   */
} modes_gather_config;

typedef struct {
    int32_t  nranks, rank;
    int32_t  rccl_version;          /* ncclGetVersion()                                                                */
    uint32_t reserved;
    uint64_t calls;                 /* modes_gather_records() calls so far                                             */
    uint64_t p2p_ops;               /* ncclSend + ncclRecv this rank issued                                            */
    uint64_t bytes_received;        /* root: record bytes that arrived from other ranks (or through the loopback)      */
    uint64_t bytes_sent;
    double   gather_ms;             /* GPU time of the exchanges on the gather's stream (counts + records), summed      */
} modes_gather_stats;

/* modes_gather_create only: the communicator came up but its first transfers did not complete - see below */
#define MODES_GATHER_ERR_PROBE (-6)

/* Rank 0: a fresh id (MODES_GATHER_ID_BYTES bytes).  Every rank passes the same id to modes_gather_create. */
int  modes_gather_unique_id(void *id);
/* Collective: returns when all nranks ranks have joined AND a 64-byte ncclSend / ncclRecv ring (rank r -> r + 1) has completed on
 * the new communicator: the first transfer between two devices of two processes is where a wrong IPC mode or a missing
 * peer-to-peer path shows, and it shows as a hang.  The probe has 300 s ($MODES_GATHER_PROBE_SECONDS; 0 = no probe; generous
 * because on a fresh box RCCL's own start has taken 60-435 s while the image pages in); when it
 * runs out the call returns MODES_GATHER_ERR_PROBE and the text names the value of HSA_ENABLE_IPC_MODE_LEGACY the process
 * ran with - the host's cue to start the job once more with the other one (dump1090_amd --ranks does).
 * AFTER MODES_GATHER_ERR_PROBE THE PROCESS MUST EXIT: a communicator with a transfer stuck in it cannot be torn down, so the
 * object, its stream and the communicator are deliberately left behind; do not create another communicator in this process
 * (a restart means a new process - HSA reads the IPC mode at process start anyway).  Every other failure frees everything. */
int  modes_gather_create(const modes_gather_config *cfg, const void *id, modes_gather **out);
void modes_gather_destroy(modes_gather *g);
/* Text of the last error on g (or of the last failed create / unique_id of this thread when g == NULL). */
const char *modes_gather_last_error(const modes_gather *g);

/* The device buffers of `slot` for modes_gpu_set_output: the rank's ordered list (capacity = cap_records) and its
 * 8-byte length. */
int  modes_gather_output(modes_gather *g, uint32_t slot, void **d_records, uint64_t *capacity, void **d_count);
/* This rank has no GPU call in this round (the stream ran out): its length for `slot` is 0.  Asynchronous, ordered before
 * the slot's next modes_gather_counts. */
int  modes_gather_set_empty(modes_gather *g, uint32_t slot);
/* Queue the all-gather of the lengths.  Call when the kernels that write d_count are complete (modes_gpu_fetch_device
 * has returned) - nothing is queued into the detect's stream.  Asynchronous. */
int  modes_gather_counts(modes_gather *g, uint32_t slot);
/* Waits for the lengths, then queues the transfers (and, on rank 0, the copy of the whole list to pinned host memory).
 * MODES_ERR_OVERFLOW - on every rank alike - when some rank's list exceeds cap_records.  Asynchronous otherwise. */
int  modes_gather_records(modes_gather *g, uint32_t slot);
/* Waits for the transfers of `slot`.  Rank 0: *records = every rank's records in rank order (host memory, valid until the
 * slot's next modes_gather_counts), *n_records their number, *counts the nranks lengths.  Other ranks: NULL / 0 / their view
 * of the lengths. */
int  modes_gather_wait(modes_gather *g, uint32_t slot, const modes_record **records, uint64_t *n_records,
                       const uint64_t **counts);
int  modes_gather_get_stats(const modes_gather *g, modes_gather_stats *out);

/* The second list, for hosts that print the reference's --stats (dump1090.c:2993-3006): the counters need every preamble
 * position of the stream on the rank that resolves (dump1090.c:1651 counts preambles whose first noise gate fails, too), so
 * each rank's positions travel to rank 0 with its records - same length exchange (the all-gather carries both lengths), same
 * group of exact-size transfers, one more device-to-host copy on rank 0.
 * modes_gather_set_candidates: this rank's positions of the call in `slot` - framed coordinates, ascending, HOST memory
 *   (modes_gpu_result.candidates after modes_gpu_fetch_device with keep_candidates; copied before the function returns) - or
 *   n = 0.  Call it before the slot's modes_gather_counts; without it the slot's second list is empty in that round.
 *   MODES_ERR_ARG when the communicator was made with cap_candidates == 0, MODES_ERR_OVERFLOW - on every rank, from
 *   modes_gather_records - when some rank's list exceeds cap_candidates.
 * modes_gather_candidates: after modes_gather_wait, rank 0: every rank's positions in rank order (host memory, valid until the
 *   slot's next modes_gather_counts); other ranks: NULL / 0. */
int  modes_gather_set_candidates(modes_gather *g, uint32_t slot, const uint64_t *candidates, uint64_t n);
int  modes_gather_candidates(modes_gather *g, uint32_t slot, const uint64_t **candidates, uint64_t *n);

#define MODES_GATHER_ABI 2
int  modes_gather_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
