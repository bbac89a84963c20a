// SPDX-License-Identifier: BSD-3-Clause
/*
 * nnn_node.h -- one object for all the GPUs of a node.
 *
 * The reference's hosts keep a vector of states and walk it (`states` in src/nnnoiseless.rs:305-320, the per-channel loop of
 * src/signal.rs:102-104).  Streams are independent (src/denoise.rs:125 `Send + Sync`, no shared mutable state), so a node of
 * several MI355X shards them: device i of n owns the contiguous block [lo_i, hi_i) of the streams (balanced: the first
 * n_streams mod n devices take one more), its own state slab and model replica, and NOTHING crosses between devices -- no
 * data-path collective.  A nnn_node is that split as a library object: one nnn_batch per device, one host thread per device that
 * enqueues (and, for host buffers, transfers) that device's share, fan-out and join inside every call.  The threads are pinned to
 * the CPUs local to their device (nnn_node_shard_cpus) and make their batches at the same time.  A host that would
 * otherwise hand-write the loop over devices, the stream split and the threads calls this instead of nnn_batch_*.
 *
 * Plain C ABI, same conventions as nnn_batch.h: 0 on success, nnn_last_error() has the text, nothing falls back to a CPU path.
 */
#ifndef NNN_NODE_H
#define NNN_NODE_H

#include "nnn_batch.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nnn_node nnn_node;

/* n_streams x DenoiseState::with_model(model) (model NULL = the built-in weights) over the listed HIP devices (ordinals may
 * repeat: two shards on one device).  opts as nnn_batch_create_opts (NULL = defaults); every shard gets the same options.
 * n_streams >= n_devices.  NULL on failure. */
nnn_node *nnn_node_create(const RNNModel *model, int n_streams, const int *devices, int n_devices, const nnn_batch_opts *opts);
void nnn_node_destroy(nnn_node *n);
int nnn_node_num_streams(const nnn_node *n);
int nnn_node_num_shards(const nnn_node *n);
/* shard i: its device ordinal and its block [*lo, *hi) of the node's streams */
int nnn_node_shard(const nnn_node *n, int i, int *device, int *lo, int *hi);
/* the CPUs shard i's host thread is pinned to -- the local_cpulist of its device's PCI function, e.g. "0-63,128-191"; "" when the
 * platform does not say or the node runs its shards on the caller's thread.  Owned by the node. */
const char *nnn_node_shard_cpus(const nnn_node *n, int i);
/* the shard's own batch, for everything nnn_batch.h offers per device (taps, schedule, snapshots); owned by the node */
nnn_batch *nnn_node_batch(nnn_node *n, int i);
int nnn_node_reset(nnn_node *n);
/* A processing call that fails on any shard (the first failing shard's text in nnn_last_error) still runs and joins the other
 * shards, which have then advanced while the failing one has not: the node is FAILED from there on -- every later processing call
 * is refused with the original text -- until nnn_node_reset, exactly as a single batch stays faulted until nnn_batch_reset.
 * nnn_node_reset clears the condition only if every shard resets.  While the node is failed, nnn_node_synchronize, nnn_node_fault
 * and the per-shard handles of nnn_node_batch stay usable (for inspection and draining: the shards sit at different frame counts).
 * A nnn_node is NOT thread-safe: one host thread drives it (its state, the failed flag included, carries no lock); the batches
 * inside it keep their own locking. */

/* n_frames x process_frame for every stream of the node, host buffers laid out as for nnn_batch_process_host over ALL streams:
 *   sample i of frame t of stream s: in[s * stream_stride + t * frame_stride + i], out likewise (may alias in)
 *   VAD: vad[t * n_streams + s] (NULL to skip)
 * Every device's share goes up, through its kernels and back on that device's own thread and copy streams, all devices at once;
 * the call returns when every shard is done.  Page-locked buffers (nnn_host_alloc) are transferred by DMA. */
int nnn_node_process_host(nnn_node *n, const float *in, float *out, float *vad, int n_frames, size_t stream_stride, size_t frame_stride);
/* The same on the packed PCM formats of nnn_batch_process_pcm_host; layout->channels must divide every shard's stream count
 * (the split keeps channel groups together when n_streams / channels divides evenly over the shards: shards are cut in whole
 * groups). */
int nnn_node_process_pcm_host(nnn_node *n, const void *in, void *out, float *vad, int n_frames, const nnn_pcm_layout *layout);
/* Buffers resident on the devices: shard i's share in device memory of ITS device, d_in[i] / d_out[i] / d_vad[i] laid out as for
 * nnn_batch_process_device over that shard's streams (d_vad may be NULL, or hold NULLs).  Asynchronous: enqueued on every shard's
 * own stream; nnn_node_synchronize waits for all of them. */
int nnn_node_process_device(nnn_node *n, const float *const *d_in, float *const *d_out, float *const *d_vad, int n_frames,
                            size_t stream_stride, size_t frame_stride);
/* The same with a table of HIP streams, hip_streams[i] a hipStream_t of shard i's device to enqueue that shard's share on (the table
 * or an entry may be NULL: the shard's batch's own non-blocking stream, which is NOT ordered with the caller's null stream -- see
 * nnn_batch_process_device), and with the tables' length stated: n_tables must equal nnn_node_num_shards (the plain form has no
 * count and trusts the caller's tables to be long enough). */
int nnn_node_process_device_streams(nnn_node *n, const float *const *d_in, float *const *d_out, float *const *d_vad,
                                    void *const *hip_streams, int n_tables, int n_frames, size_t stream_stride, size_t frame_stride);
int nnn_node_synchronize(nnn_node *n);
/* 1 if any shard reports nnn_batch_fault */
int nnn_node_fault(const nnn_node *n);

#ifdef __cplusplus
}
#endif
#endif /* NNN_NODE_H */
