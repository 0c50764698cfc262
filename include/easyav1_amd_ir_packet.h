/*
 * easyav1_amd_ir_packet.h -- a parsed frame as bytes inside an AV1 temporal unit.
 *
 * BASELINE.json's north_star keeps the OBU + entropy / symbol parse on the host and hands the GPU
 * "one frame in parsed form" (easyav1_amd_video.h).  This repository has the headers side of that
 * host parse (easyav1_amd_obu.h) but not the block-level symbol decoder, so nothing here can turn
 * OBU_TILE_GROUP bytes into ea1_block[] / coefficients.  To run the WHOLE public interface
 * (include/easyav1.h: easyav1_init_* -> easyav1_decode_next -> easyav1_get_video_frame, the
 * reference's src/easyav1.c:2360-2428 decode_video path and its tools/easyav1_benchmark.c) on real
 * container files anyway, a frame may travel through the container ALREADY PARSED: the temporal
 * unit of a V_AV1 track carries, next to its ordinary OBUs (temporal delimiter, sequence header),
 * one OBU_METADATA whose metadata_type lies in the range the AV1 specification leaves to private
 * use (6.7.1: values of 32 and above are "unregistered user private") and whose payload is the
 * frame's hand-over structures back to back.  A conforming AV1 decoder skips such an OBU; the
 * shim of this repository (easyav1_amd/csrc/easyav1_api.cpp) reconstructs it on the GPU.  It is a
 * test and benchmark vehicle for the drop-in boundary, not a bitstream format.
 *
 * Payload of the metadata OBU (all little-endian):
 *     leb128   metadata_type = EA1_IR_METADATA_TYPE                     (2 bytes: 0xA1 0x1D)
 *     u8[4]    "EA1F"
 *     u32      version = 1
 *     u32      sizeof(ea1_frame_desc), sizeof(ea1_block), sizeof(ea1_txblock), sizeof(ea1_lr_unit)
 *     u32      num_sb (entries of cdef_idx), lr_count[3]
 *     ea1_frame_desc
 *     ea1_block[num_blocks], ea1_txblock[num_tx], int16 coefs[num_coefs], u8 aux[num_aux],
 *     u8 cdef_idx[num_sb], ea1_lr_unit lr[0][..], lr[1][..], lr[2][..]
 *     u8       0x80   (trailing bits of the metadata OBU)
 * Arrays follow each other without padding; the parser copies them into aligned storage.
 */
#ifndef EASYAV1_AMD_IR_PACKET_H
#define EASYAV1_AMD_IR_PACKET_H

#include <stddef.h>
#include <stdint.h>

#include "easyav1_amd_video.h"

#ifdef __cplusplus
extern "C" {
#endif

#define EA1_IR_METADATA_TYPE 0xEA1u
#define EA1_IR_VERSION 1u

/* one parsed frame, host memory: exactly the arguments of ea1_job_create */
typedef struct ea1_ir_frame {
    ea1_frame_desc desc;
    const ea1_block *blocks;
    const ea1_txblock *txs;
    const int16_t *coefs;
    const uint8_t *aux;
    const uint8_t *cdef_idx;
    uint32_t num_sb;
    const ea1_lr_unit *lr[3];
    uint32_t lr_count[3];
} ea1_ir_frame;

/* Hand-over of n parsed frames at once on up to `threads` host threads of the library (a player that keeps many
 * GOP chains in flight hands over one frame of every chain per step, easyav1_amd_video.h): out[k] = the job of
 * frames[k].  All or nothing: on an error no job is left behind. */
int ea1_job_create_batch(ea1_recon *r, const ea1_ir_frame *const *frames, int n, int threads, ea1_job **out);

/* The same hand-over as two calls, for a caller that pipelines frames itself (libeasyav1.so's video sink):
 *   ea1_job_prepare  the host half of ea1_job_create -- validation of every record, counting, packing into pinned
 *                    staging; no HIP call, any thread, any number of them at once
 *   ea1_job_commit   the device half for n prepared jobs -- their copies, then ONE launch of each derivation kernel
 *                    and ONE event for all of them; one thread at a time per context
 * A prepared job that is never committed is released with ea1_job_destroy.  On an error of ea1_job_commit the jobs stay
 * the caller's to destroy. */
int ea1_job_prepare(ea1_recon *r, const ea1_ir_frame *frame, ea1_job **out);
int ea1_job_commit(ea1_recon *r, ea1_job *const *jobs, int n);

/* bytes of the metadata OBU payload for `f` (0 on a NULL argument) */
size_t ea1_ir_packet_bytes(const ea1_ir_frame *f);
/* writes the payload (ea1_ir_packet_bytes(f) bytes) */
int ea1_ir_packet_write(const ea1_ir_frame *f, uint8_t *out, size_t cap);
/* 1 when `payload` (the payload of an OBU_METADATA) starts like a parsed frame, else 0 */
int ea1_ir_packet_probe(const uint8_t *payload, size_t size);

/* Parses a payload into aligned storage owned by the returned object; `out` points into it.
 * Every count and size is checked against the payload size (the bytes come from a file). */
typedef struct ea1_ir_storage ea1_ir_storage;
int ea1_ir_packet_parse(const uint8_t *payload, size_t size, ea1_ir_storage **storage, ea1_ir_frame *out);
void ea1_ir_storage_destroy(ea1_ir_storage *s);
/* The same into memory of the caller: `mem` 8-byte aligned, `cap` bytes; size + 64 bytes always suffice.  With memory from
 * ea1_host_alloc the coefficients then travel to the device by DMA straight from where this call put them (no staging copy). */
int ea1_ir_packet_parse_into(const uint8_t *payload, size_t size, void *mem, size_t cap, ea1_ir_frame *out);
/* only the frame's description (its size above all), after the same checks of the counts against the payload size */
int ea1_ir_packet_peek(const uint8_t *payload, size_t size, ea1_frame_desc *desc);

#ifdef __cplusplus
}
#endif
#endif
