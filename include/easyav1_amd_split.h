/*
 * easyav1_amd_split.h -- the tile-column split of ONE stream over several GPUs (BASELINE.json configs[3]) as a
 * NATIVE driver with its collective: what easyav1_amd/parallel.py's TileSplit did from Python over torch.distributed up to round 4 is
 * available to a C application that links libeasyav1_amd.so (round-4 verdict, missing #2).
 *
 * The reference has no counterpart: it is one process with dav1d's tile threads inside it (/root/reference/src/easyav1.c:1401-1405,
 * Dav1dSettings.n_threads).  Here every rank -- one process per GPU -- owns a contiguous, superblock-aligned range of tile columns of
 * every frame (the algorithm, and why 16 columns of halo suffice: easyav1_amd_video.h, "Tile-column split"); per frame and rank
 *
 *     reconstruction of the strip -> ALL-GATHER of the 2 x 16 edge columns -> in-loop filters -> ALL-GATHER of the finished strips
 *
 * The collective is a function the split calls twice per frame.  Two are built in:
 *   RCCL   ncclAllGather on the split's stream (xGMI between the GPUs of a node).  librccl.so is loaded when the first such split is made
 *          (dlopen: the library itself does not depend on it); the communicator is made from a 128-byte unique id that rank 0 obtains
 *          (ea1_rccl_unique_id) and hands to the other ranks by whatever means the application has (a file, MPI, a socket, torch.distributed)
 *   any    ea1_all_gather_fn: the caller's own transport (the tests run N ranks as N threads of one process on one GPU with a gather
 *          through a shared table; a gloo rehearsal stages through host memory)
 *
 * All buffers are device memory.  Calls of one split come from one thread at a time; different ranks are different splits.
 */
#ifndef EASYAV1_AMD_SPLIT_H
#define EASYAV1_AMD_SPLIT_H

#include <stddef.h>
#include <stdint.h>

#include "easyav1_amd_ir_packet.h"
#include "easyav1_amd_video.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ea1_split ea1_split;

/* All-gather of `bytes` bytes per rank: d_send (this rank's contribution) -> d_recv (world x bytes, rank-major; the own contribution
 * appears at its place too), ordered behind the work already enqueued on `stream` and complete -- or enqueued on `stream` -- when the
 * function returns.  0 = success, anything else fails the frame with EA1_EHIP. */
typedef int (*ea1_all_gather_fn)(void *userdata, const void *d_send, void *d_recv, size_t bytes, void *stream);

/* x_bounds: world + 1 ascending luma columns, x_bounds[0] = 0, x_bounds[world] = the context's width, multiples of 64 in between
 * (superblock aligned; with 128-sample superblocks multiples of 128): rank q owns [x_bounds[q], x_bounds[q + 1]).  The context is the
 * rank's own (ea1_recon_create_batch on its GPU); it must outlive the split. */
int ea1_split_create(ea1_split **out, ea1_recon *ctx, const uint32_t *x_bounds, int world, int rank, ea1_all_gather_fn gather, void *userdata);

/* The same over RCCL.  unique_id: the 128 bytes of ea1_rccl_unique_id() as rank 0 made them (every rank passes the same bytes; the call
 * is collective: it returns when all `world` ranks have made theirs).  EA1_ENODEV when librccl.so cannot be loaded. */
#define EA1_RCCL_UNIQUE_ID_BYTES 128
int ea1_rccl_unique_id(void *id128);
int ea1_split_create_rccl(ea1_split **out, ea1_recon *ctx, const uint32_t *x_bounds, int world, int rank, const void *unique_id);

/* this rank's job for a parsed frame: the work lists hold the strip's own blocks (ea1_job_create_strip) */
int ea1_split_job_create(ea1_split *s, const ea1_ir_frame *frame, ea1_job **out);

/* One frame of every chain of the context (jobs: ea1_recon_num_chains() entries of ea1_split_job_create, NULL = chain idle; all ranks pass
 * the same frame positions): when the work enqueued here has run, reference slot dst_slot of every chain holds the WHOLE finished picture
 * on every rank.  stream: the HIP stream everything is enqueued on (NULL = the default stream). */
int ea1_split_decode(ea1_split *s, ea1_job *const *jobs, int dst_slot, void *stream);

/* bytes this rank contributes to / receives from the two collectives of one frame (all chains together): for reports */
void ea1_split_traffic(const ea1_split *s, size_t *halo_sent, size_t *halo_received, size_t *strip_sent, size_t *strip_received);

void ea1_split_destroy(ea1_split *s);

#ifdef __cplusplus
}
#endif
#endif
