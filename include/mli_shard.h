/*
 * mli_shard.h -- C ABI for the row-sharded engine group (BASELINE config 5): the continuous batch split by
 * row over the GPUs of one node, below Python.  The reference has no multi-GPU code (its README lists it as a plan); what
 * this binds is N of its engines (include/inferencer.h:18-32: start_paged_attention_*_inference_engine), one per GPU, each
 * with its own page pool, page table, lengths and scheduler, in ONE host process with one host thread per GPU.  Rows are
 * independent in every kernel of the path, so the only exchange is the all-gather of the generated token ids
 * (decoder_result, int32 [n_batch, n_forward_rounds] per rank) after every iteration: one ncclAllGather per rank and step
 * through RCCL (/opt/rocm/include/rccl/rccl.h; communicators from ncclCommInitAll, i.e. xGMI peer-to-peer inside the
 * process), enqueued on the engine's own stream behind the forward that produced the tokens.
 *
 * librccl.so is opened on the first mli_shard_group_create (dlopen), so single-GPU users of libmli_hip.so never load it.
 * A group of one rank runs the same code path (a 1-rank communicator); that is what the GPU tests of this repository can
 * reach -- no N > 1 run has been possible on the 1-GPU test boxes: N > 1 is UNMEASURED.
 */
#ifndef MLI_SHARD_H
#define MLI_SHARD_H

#include "mli_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mli_shard_group mli_shard_group;

typedef struct {
    long long total_tokens;   /* over all ranks */
    double seconds;           /* wall time of mli_shard_group_run */
    long long iterations;     /* lock-step iterations (= all-gathers per rank) */
    int finished;             /* items finished, over all ranks */
    int ranks_seen;           /* result of an ncclAllReduce(sum) of ones at creation: must equal n_ranks */
    double gather_us;         /* mean host-visible cost of enqueueing one all-gather (the collective itself runs on the
                                 engine's stream beside the next iteration's host work) */
} mli_shard_stats;

/* config: what EVERY rank gets (n_batch = rows per GPU, n_blocks = pages per GPU); config->device is ignored, rank r runs
 * on devices[r] (distinct ordinals).  Weights as in mli_engine_create (host pointers, replicated to every GPU).  Returns 0
 * or a negative error (mli_shard_last_error()). */
int mli_shard_group_create(const mli_engine_config* config, int n_ranks, const int* devices, const float* emb_table,
                           const float* pos_table, const float* wk, const float* wq, const float* wv,
                           mli_shard_group** out);
/* The same group over a LOOPBACK exchange: every rank's engine lives on ONE device (`device`), and the all-gather is n_ranks
 * device-to-device copies per rank on the engine's stream instead of a collective (RCCL refuses a communicator with one GPU
 * twice).  For hosts that run several engine replicas on one GPU (each replica's host bookkeeping overlaps the others' kernels)
 * -- and it is how the lock-step logic (ranks that finish early, empty ranks, failures) is tested on a one-GPU box.  Everything
 * else (item dealing, stepping, stats, mli_shard_group_gathered) is identical; stats.ranks_seen = n_ranks by construction. */
int mli_shard_group_create_loopback(const mli_engine_config* config, int n_ranks, int device, const float* emb_table,
                                    const float* pos_table, const float* wk, const float* wq, const float* wv,
                                    mli_shard_group** out);
void mli_shard_group_destroy(mli_shard_group* group);

int mli_shard_group_size(const mli_shard_group* group);

/* Queue one item on rank (id mod n_ranks) -- the global queue dealt round-robin, so every rank's scheduler sees the
 * same kind of load. */
int mli_shard_group_add_item(mli_shard_group* group, int id, const int* tokens, int n_tokens);

/* Run every rank to completion in lock step: iteration = every rank's mli_engine_step, then its all-gather; the group
 * stops when every rank's queue is empty (ranks that finish early keep joining the collective). */
int mli_shard_group_run(mli_shard_group* group, mli_shard_stats* stats);

/* Rank `rank`'s copy of the gathered token ids of the last iteration: device pointer on devices[rank],
 * int32 [n_ranks][n_batch * n_forward_rounds]. */
int mli_shard_group_gathered(mli_shard_group* group, int rank, void** device_ptr, int* count);

/* The engine of one rank (finished items, stats); owned by the group. */
mli_engine* mli_shard_group_engine(mli_shard_group* group, int rank);

const char* mli_shard_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MLI_SHARD_H */
