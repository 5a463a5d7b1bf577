/*
 * tgsf_rccl.h -- the one collective of a multi-GPU job (optional library libtgsf_rccl.so, links RCCL).
 *
 * Reads shard across GPUs with no data-path exchange: every rank filters its own batches through
 * include/tgsf.h.  What the ranks share are the additive tallies -- the reference merges them over its worker
 * threads at the end of a run, src/TGSFilter.cpp:3208-3213 (DropInfo), :2673-2725 and :2586-2597 (the QC tables);
 * across GPUs that merge is ONE sum all-reduce of the flat tally vector, on the device, over RCCL / xGMI
 * (with check_layout != 0 a 2-word all-reduce of the vector lengths goes first: a set-up check, see below).
 *
 * One process per GPU (the usual RCCL set-up: ncclGetUniqueId on rank 0, handed to the others by the launcher,
 * ncclCommInitRank on every rank).  A single process driving several GPUs does not need this library: it merges
 * the vectors of its contexts on the host (tgsf_counters_used), as tgsfilter --devices does.
 */
#ifndef TGSF_RCCL_H
#define TGSF_RCCL_H

#include "tgsf.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Sum the tally vector of `ctx` over all ranks of `nccl_comm` (an ncclComm_t), in place, in HBM: afterwards
 * tgsf_counters(ctx) / tgsf_counters_used(ctx) return the totals of the whole job on every rank.  Call once, after
 * the last batch (it waits for the context first), on every rank, with contexts created with the same bc_len and
 * max_read_len (the vector layout depends on them; a rank holding a different layout fails with TGSF_E_INVALID
 * before any data moves -- lengths are compared with a first, 2-word all-reduce at `check_layout` != 0; pass 0 when
 * the caller has established it at set-up).
 *
 * The four "rows used" words of the vector are maxima, not sums: every rank carries them in a slot of its own
 * behind the vector (zeros elsewhere), so that the same sum delivers all of them and each rank takes the maximum.
 * hip_stream: the stream to run on (NULL: the context's).  Returns TGSF_OK or a negative tgsf_status; text through
 * tgsf_rccl_last_error().
 */
int tgsf_rccl_allreduce_counters(tgsf_ctx* ctx, void* nccl_comm, int rank, int world, int check_layout, void* hip_stream);

/*
 * The communicator of the job, for a caller that does not bind RCCL itself (the command line's rank processes,
 * tgsfilter --ranks / --shard): rank 0 draws the job's id (128 bytes, ncclGetUniqueId) and hands it to the other ranks
 * by whatever connects them; every rank then joins with tgsf_rccl_comm_init -- collective, and slow the first time
 * (RCCL's topology discovery: call it on a helper thread beside the filtering, well before the all-reduce) -- on the
 * device its contexts live on.  `device`: HIP device index, made current for the calling thread.
 * tgsf_rccl_comm_count: ranks RCCL itself sees in the communicator (ncclCommCount).
 */
#define TGSF_RCCL_ID_BYTES 128
int tgsf_rccl_unique_id(void* id128);
int tgsf_rccl_comm_init(int device, const void* id128, int rank, int world, void** nccl_comm);
int tgsf_rccl_comm_count(void* nccl_comm, int* n_ranks);
void tgsf_rccl_comm_destroy(void* nccl_comm);

const char* tgsf_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* TGSF_RCCL_H */
