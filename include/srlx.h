/*
 * srlx.h -- C ABI of libsrlx.so, the MI355X (gfx950) native data path behind the
 * SRL (pocokhc/simple_distributed_rl) Memory / Worker / Trainer plugin surface.
 *
 * Conventions
 *   - every entry point returns an int status (SRLX_OK == 0, negative = error); no
 *     exceptions cross the ABI; srlx_last_error() returns a thread-local message.
 *   - handles are opaque, owned by the caller, created on ONE HIP device.
 *   - `on_device` = 0: array arguments are HOST pointers; the call stages them through
 *     pinned memory, runs on `stream` and returns after the results are in the host arrays.
 *     `on_device` = 1: array arguments are DEVICE pointers on the handle's device; the call
 *     only enqueues work on `stream` (no host synchronisation, HIP-graph capturable).
 *   - `stream` is a hipStream_t passed as void* (NULL = the handle's own stream).
 *   - calls on one handle must be serialised by the caller (the Python shim holds a lock),
 *     matching the reference, whose memory is only ever touched under the GIL
 *     (srl/base/run/play_mp.py:248-286).
 *
 * Each group cites the reference interface it replaces (paths relative to the
 * reference repository root).  The reference-side binding a maintainer would add is
 * shown in INTEGRATION.md.
 */
#ifndef SRLX_H
#define SRLX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRLX_VERSION 1

#define SRLX_OK 0
#define SRLX_ERR_INVALID (-1)            /* bad argument */
#define SRLX_ERR_HIP (-2)                /* a HIP runtime call failed (see srlx_last_error) */
#define SRLX_ERR_NOMEM (-3)
#define SRLX_ERR_UNIFORMS_EXHAUSTED (-4) /* sample(): retries consumed the whole uniform stream */
#define SRLX_ERR_UNSUPPORTED (-5)

const char *srlx_last_error(void);
int srlx_version(void);
int srlx_device_count(int *out_count);
/* name (e.g. "gfx950"), CU count and total HBM bytes of a device */
int srlx_device_info(int device, char *arch_name, int arch_name_len, int *cu_count, int64_t *hbm_bytes);

/* ------------------------------------------------------------------------------------
 * Proportional prioritized replay (GPU-resident sum-tree)
 *
 * Replaces: srl/rl/memories/priority_memories/proportional_memory.py:13-205
 *           (SumTree + ProportionalMemory) and its pybind11 twin
 *           srl/rl/memories/priority_memories/cpp_module/src/proportional_memory.cpp:14-248,
 *           behind IPriorityMemory (srl/rl/memories/priority_memories/imemory.py:7-34).
 *
 * The tree is the reference's implicit heap: 2*capacity-1 float64 nodes, leaf slot j is
 * node j+capacity-1, parent(i) = (i-1)/2.  `sample` returns TREE indices (what the
 * reference hands back as update_args).  Tree contents are bit-identical to the reference
 * after the same call sequence: updates propagate fp64 deltas to ancestors in call order.
 * ------------------------------------------------------------------------------------ */
typedef struct srlx_per srlx_per_t;

/* how the `prio` array of add/update is to be interpreted */
#define SRLX_PRIO_NONE 0 /* add only: priority=None -> current max_priority (proportional_memory.py:121-122) */
#define SRLX_PRIO_F64 1  /* float64 values, p = (|x|+eps)^alpha in fp64 (:124; :172 with float64/int/list input) */
#define SRLX_PRIO_F32 2  /* float32 values, p = (|x|+eps)^alpha evaluated in float32 like numpy (:172), widened */
#define SRLX_PRIO_RAW 3  /* float64 values already transformed by the caller (_restore_skip, :123) */

/* ProportionalMemory.__init__ / clear  (proportional_memory.py:96-115; cpp :100-121) */
int srlx_per_create(srlx_per_t **out, int64_t capacity, double alpha, double beta_initial, double beta_steps,
                    int has_duplicate, double epsilon, int device);
int srlx_per_destroy(srlx_per_t *h);
int srlx_per_clear(srlx_per_t *h, void *stream);
/* length() (:117-118).  Host mirror; exact as long as adds go through srlx_per_add. */
int64_t srlx_per_length(const srlx_per_t *h);
int64_t srlx_per_capacity(const srlx_per_t *h);

/* add() x n  (:120-129, SumTree.add :71-79).  Equivalent to n sequential reference add()
 * calls writing ring slots write, write+1, ... (mod capacity).  prio may be NULL for
 * SRLX_PRIO_NONE.  n <= capacity. */
int srlx_per_add(srlx_per_t *h, int64_t n, const void *prio, int prio_kind, int on_device, void *stream);

/* sample()  (:131-169, SumTree._retrieve :56-66).
 *   uniforms[0..n_uniforms): the values the reference's random.random() calls (:147) would
 *     return, in order; one per descent attempt, so n_uniforms >= batch_size and more if
 *     zero-priority or duplicate draws are rejected (:150-157).
 *   step       : beta schedule input (:138-140); if d_step != NULL (device int64) it is read
 *                on the device instead (graph-capturable).
 *   out_idx    : int64[batch_size] tree indices
 *   out_w      : float64[batch_size] IS weights / max (may be NULL)
 *   out_w32    : float32[batch_size] same, cast as PriorityReplayBuffer.sample does
 *                (srl/rl/memories/priority_replay_buffer.py:235) (may be NULL)
 *   out_used   : int64[1] number of uniforms consumed, or -1 if the stream was exhausted.
 * on_device=0 additionally returns SRLX_ERR_UNIFORMS_EXHAUSTED in that case. */
int srlx_per_sample(srlx_per_t *h, int64_t batch_size, int64_t step, const int64_t *d_step, const double *uniforms,
                    int64_t n_uniforms, int64_t *out_idx, double *out_w, float *out_w32, int64_t *out_used,
                    int on_device, void *stream);

/* update()  (:171-177, SumTree.update :81-86).  indices are tree indices from sample();
 * duplicates inside one call see each other's writes in list order, as in the reference. */
int srlx_per_update(srlx_per_t *h, int64_t n, const int64_t *indices, const void *prio, int prio_kind, int on_device,
                    void *stream);

/* backup()/restore()  (:179-205).  tree_host: 2*capacity-1 float64 (HOST pointers always). */
int srlx_per_backup(srlx_per_t *h, double *max_priority, int64_t *size, int64_t *write, double *tree_host);
int srlx_per_restore(srlx_per_t *h, double max_priority, int64_t size, int64_t write, const double *tree_host);
/* restore() from a backup of a different capacity (:195-205): clear, then re-add the
 * first old_size leaves of the old tree with _restore_skip. */
int srlx_per_restore_resized(srlx_per_t *h, int64_t old_capacity, int64_t old_size, const double *old_tree_host);

/* raw device views (zero-copy wrapping by the host runtime, tests) */
int srlx_per_tree_ptr(srlx_per_t *h, void **d_tree, int64_t *tree_len);
/* device struct { double max_priority; int64 size; int64 write; int64 pad; } */
int srlx_per_state_ptr(srlx_per_t *h, void **d_state);
/* re-read size/write from the device after HIP-graph replays that contained adds */
int srlx_per_refresh(srlx_per_t *h, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SRLX_H */
