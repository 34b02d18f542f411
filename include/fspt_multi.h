/*
 * fspt_multi.h - libfspt on several GPUs of one node.
 *
 * Pixels are independent (tracer.fs:516-517) and the RNG depends on pixel coordinates and randBase only (camera.fs:38,
 * tracer.fs:458), so a frame cut into 32x32 tiles dealt round-robin to the devices (fspt_target_set_shard, fspt.h) and
 * assembled afterwards is bit-identical to a single-GPU render: no collective on the data path, ONE exchange at read-out.
 * Two kinds of host: one process driving all devices (fspt_multi_*), or one process per device that moves the bytes
 * itself (fspt_target_pack_tiles / fspt_target_unpack_tiles around its own collective).  Included by fspt.h.
 */
#ifndef FSPT_MULTI_H
#define FSPT_MULTI_H

#include "fspt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 * One frame over several GPUs of a node, driven by ONE host thread (README.md:28: "tiled rendering" is a TODO in the
 * reference).  One target per listed device owns every n_devices-th 32x32 tile (the RNG depends on pixel coordinates
 * and randBase only - camera.fs:38, tracer.fs:458 - so the assembled frame is bit-identical to a single-GPU render).
 * The draw calls enqueue on every device and return; NO data moves while rendering.  fspt_multi_read_radiance /
 * fspt_multi_draw do the one exchange: packed tiles travel to devices[0] with peer copies (xGMI) on the devices' own
 * streams and are scattered into its accumulator.  A device may be listed more than once (a 1-GPU box exercises the
 * path that way).
 * ---------------------------------------------------------------------- */
typedef struct fspt_multi fspt_multi;
int fspt_multi_create(const fspt_scene_desc *desc, const int *devices, uint32_t n_devices,
                      uint32_t width, uint32_t height, fspt_multi **out);
int fspt_multi_destroy(fspt_multi *m);
/* the per-device target i (for fspt_target_set_pipeline / _set_tail / _prepare / counters); owned by m */
int fspt_multi_target(fspt_multi *m, uint32_t i, fspt_target **out);
int fspt_multi_camera(fspt_multi *m, const float P[3], const float I[3], float fov_scale, const float lens[2],
                      float rand_base);                                     /* drawCamera on every device   */
int fspt_multi_trace(fspt_multi *m, uint32_t tick, float rand_base, float env_theta, uint32_t num_bounces);
int fspt_multi_render(fspt_multi *m, const fspt_camera_params *cam, uint32_t first_tick, uint32_t n_ticks,
                      uint64_t seed);                                       /* fspt_render on every device  */
int fspt_multi_clear(fspt_multi *m);
int fspt_multi_sync(fspt_multi *m);
/* Gather (see above), then what fspt_read_radiance / fspt_draw do on the assembled frame.  Blocking. */
int fspt_multi_read_radiance(fspt_multi *m, float *out);
int fspt_multi_draw(fspt_multi *m, float exposure, float saturation, int denoise, float max_sigma, uint8_t *out_rgba8);
/* Bytes that crossed between devices in the most recent gather (the exchange's payload: 16 bytes per foreign pixel). */
int fspt_multi_last_gather_bytes(fspt_multi *m, uint64_t *bytes);
/* Where the time of the most recent render + read-out went, per device: ms[i][0] = GPU time of device i's last render call
 * (fspt_last_kernel_ms of its target), [1] = its pack kernel(s), [2] = its copy / send / reduce to devices[0], [3] = the
 * scatter of its tiles on devices[0]; -1 where a stage did not run (devices[0] packs nothing in the copy / gather modes; in
 * the reduce mode only entry 0 has a [3]: the copy of the reduced frame into the accumulator).  n_devices must be the
 * count the handle was made with.  Waits for the devices.  (Round 6: so that the first run on more than one physical GPU
 * shows imbalance and exchange cost in one pass - none has happened yet) */
int fspt_multi_last_stage_ms(fspt_multi *m, float *ms /* [n_devices][4] */, uint32_t n_devices);
int fspt_multi_size(fspt_multi *m, uint32_t *width, uint32_t *height);  /* the frame fspt_multi_create was given */
/* How target i's tiles reach devices[0]: bit 0 = its device can write devices[0]'s memory (the direction the gather
 * copy runs), bit 1 = the reverse mapping; 0 = staged through the host; a target on devices[0] itself reports 3. */
int fspt_multi_peer_access(fspt_multi *m, uint32_t i, int *mask);

/* The read-out exchange of fspt_multi_read_radiance / fspt_multi_draw:
 *   FSPT_EXCHANGE_PEER_COPY (default)  every device's packed tiles -> devices[0] with hipMemcpyPeerAsync (above);
 *   FSPT_EXCHANGE_RCCL_GATHER          the same packed tiles as ncclSend / ncclRecv pairs in one group;
 *   FSPT_EXCHANGE_RCCL_REDUCE          ncclReduce(SUM) of full frames that are zero outside each device's own tiles
 *                                      (SURVEY 8e / BASELINE north_star: "an RCCL reduce of the radiance buffer over xGMI";
 *                                      every pixel has one owner, so the sum equals the gather bit for bit).
 * The RCCL modes create one communicator rank per device (ncclCommInitAll over the device list, which must not name a
 * device twice) the first time one is selected; librccl is loaded then (dlopen), never before - FSPT_E_STATE when it
 * cannot be.  All three give the same frame. */
enum { FSPT_EXCHANGE_PEER_COPY = 0, FSPT_EXCHANGE_RCCL_GATHER = 1, FSPT_EXCHANGE_RCCL_REDUCE = 2 };
int fspt_multi_set_exchange(fspt_multi *m, int mode);
int fspt_multi_get_exchange(fspt_multi *m, int *mode, int *rccl_version);  /* either pointer may be NULL; version 0 = RCCL not loaded */

/* ---- one process per GPU (the reference has no counterpart: README.md:28 "Tiled rendering" is a TODO) ------------------
 * A host that runs one process per device (bench.py --gpus N, fspt_amd/distributed.py: torch.distributed over RCCL) moves
 * the read-out's bytes itself; these are the two ends of that exchange.  A shard's pixels in work-index order (the order
 * fspt_target_set_shard deals tiles out: tile after tile, 8x8 patches inside a tile) are its `slots`:
 * owned tiles x tile^2, pixels of a ragged last tile row / column that lie outside the frame are holes.
 *   fspt_target_pack_tiles    the target's OWN pixels (its shard) -> packed[slots][channels], device memory;
 *   fspt_target_unpack_tiles  packed pixels of shard `shard` of `n_shards` (same tile size) -> the accumulator;
 * channels = 4: RGBA as stored; 3: RGB only - the alpha of a traced pixel is the constant 1 (tracer.fs:517), unpack
 * writes it.  Both execute recorded ticks first and return when the kernel has finished (the caller's collective runs
 * on a stream of its own). */
int fspt_target_shard_slots(fspt_target *target, uint32_t shard, uint32_t n_shards, uint64_t *slots);
int fspt_target_pack_tiles(fspt_target *target, void *packed_device, uint32_t channels);
int fspt_target_unpack_tiles(fspt_target *target, const void *packed_device, uint32_t shard, uint32_t n_shards, uint32_t channels);

#ifdef __cplusplus
}
#endif
#endif /* FSPT_MULTI_H */
