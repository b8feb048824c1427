/* The multi-GPU step through the C ABI: one process per GPU, every rank decodes its
 * contiguous shard and the two exchange steps run through the library's own RCCL entry points.  The caller
 * supplies the broadcast of the 128-byte communicator id (MPI_Bcast, a socket, a shared file ...) and the device
 * buffers.  The Makefile compiles this file against include/lsr_hip.h (object only; it has no main). */
#include <stddef.h>

#include "lsr_hip.h"

typedef void (*broadcast_fn)(void* buffer, size_t bytes, int root);

struct shard_buffers {                 /* device pointers of this rank's shard */
    const float* audio_d;              /* concatenated clips */
    const int64_t* sample_off_d;       /* [n_local + 1] */
    const int64_t* frame_off_d;        /* [n_local + 1] */
    int64_t frames;                    /* total frames of the shard */
    int max_frames;                    /* longest utterance of the shard */
    float* scores_d;                   /* [n_local] */
    uint8_t* labels_d;                 /* [n_local][16] */
    int32_t* n_labels_d;               /* [n_local] */
    uint8_t* all_labels_d;             /* [n_total][16]: every rank's records in global order */
    int32_t* all_n_labels_d;           /* [n_total] */
    double* stats_d;                   /* packed training statistics (count, sum_x, sum_xx, transition counts) */
    int64_t stats_len;
};

int lsr_example_open(int local_rank, int rank, int world, broadcast_fn broadcast, lsr_ctx** out) {
    unsigned char id[LSR_COMM_ID_BYTES];
    int rc = lsr_ctx_create(local_rank, out);
    if (rc) return rc;
    if (rank == 0 && (rc = lsr_comm_get_unique_id(*out, id))) return rc;
    broadcast(id, sizeof id, 0);
    return lsr_comm_init(*out, id, rank, world);
}

/* decode: replaces the pool.map of scripts/project5_test_ndigits_with_sil.py:29-42 */
int lsr_example_decode_step(lsr_ctx* ctx, int rank, int world, int64_t n_total, const struct shard_buffers* b) {
    int64_t lo, hi;
    int rc;
    lsr_shard_bounds(n_total, rank, world, &lo, &hi);
    rc = lsr_recognize_batch(ctx, b->audio_d, b->sample_off_d, b->frame_off_d, (int)(hi - lo), b->frames, b->max_frames,
                             b->scores_d, NULL, b->labels_d, b->n_labels_d, 16);
    if (rc) return rc;
    rc = lsr_allgather_labels(ctx, b->labels_d, b->n_labels_d, hi - lo, 16, n_total, b->all_labels_d, b->all_n_labels_d);
    if (rc) return rc;
    return lsr_synchronize(ctx);
}

/* training: replaces the pool.map of hidden_markov_model.py:746-750 (the statistics of this rank's alignment are in
 * stats_d; afterwards every rank holds the sums over all ranks and runs the same M-step) */
int lsr_example_reduce_statistics(lsr_ctx* ctx, const struct shard_buffers* b) {
    int rc = lsr_allreduce_stats(ctx, b->stats_d, b->stats_len);
    return rc ? rc : lsr_synchronize(ctx);
}

int lsr_example_close(lsr_ctx* ctx) {
    int rc = lsr_comm_destroy(ctx);
    return rc ? rc : lsr_ctx_destroy(ctx);
}
