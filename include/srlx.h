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
 *     `on_device` = 2 (srlx_per_add / _sample / _update; round 6): HOST pointers, asynchronous -- the arguments are copied
 *     into a device-visible pinned slot that the kernel reads over the link: add / update return without synchronising (later
 *     calls on the stream are ordered behind them), sample synchronises once and has no copy commands.  Arguments that do not
 *     fit a 16 KB slot take the synchronous path.
 *   - `stream` is a hipStream_t passed as void* (NULL = HIP's default stream, which is also
 *     PyTorch's default stream).
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
/* A HIP stream of priority level -1 (high) / 0 (normal) / 1 (low), non-blocking.  The device engines run the ACTORS' side on a low-priority stream: HIP keeps
 * one pool of hardware queues per level, and a HIP graph replays its branches on normal-priority internal streams -- on a pool of its own the actors' stream
 * never queues behind an update's branch (what the reference's actor / trainer PROCESSES get for free: play_mp.py:471-642). */
int srlx_stream_create(int priority_level, void **out_stream);
int srlx_stream_destroy(void *stream);
/* Measurement aid: one-thread launch writing the device's constant-rate wall clock (wall_clock64: 100 MHz) into d_buf[index]; capturable into HIP graphs. */
int srlx_debug_stamp(uint64_t *d_buf, int index, void *stream);
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
 * node j+capacity-1, parent(i) = (i-1)/2 -- LOGICALLY; in HBM three levels share a 128-byte
 * line.  `sample` returns logical TREE indices (what the reference hands back as update_args).  Tree contents are bit-identical to the reference
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
/* switch the duplicate rule of later srlx_per_sample calls (the shim completes a batch WITH duplicates when a has_duplicate=False draw
 * cannot be satisfied: the reference does the same after 9999 tries per draw, proportional_memory.py:146-158) */
int srlx_per_set_has_duplicate(srlx_per_t *h, int has_duplicate);
/* d_counter (device int64, caller-owned, NULL to switch off): every device-side srlx_per_update call adds 1 to it -- the learner's
 * train_count (trainer.py: `self.train_count += 1` after the priority write-back, model_torch.py:113-122) kept on the device without
 * a launch of its own.  Launches queued after the update see the new value. */
int srlx_per_set_update_counter(srlx_per_t *h, int64_t *d_counter);
/* every later APPENDING srlx_per_add also adds 1 to these int64 device counters (either may be NULL), inside its own launch: an engine's ring position and the
 * counter of its policy generator advance with the add that closes a lock-step instead of in launches of their own (readers ran in earlier launches). */
int srlx_per_set_add_counters(srlx_per_t *h, int64_t *d_counter0, int64_t *d_counter1);
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
/* The host loop of the reference's memory -- add one item, sample, update (srl/rl/memories/priority_replay_buffer.py:205-258; tests/quick/rl/memories/speedtest.py:15-58)
 * -- as ONE launch per sample: `n_add` <= 16 adds queued since the tree was last observed (host values: final leaf priorities with SRLX_PRIO_RAW, or add_values = NULL
 * with SRLX_PRIO_NONE) are applied inside the sampling launch, in order, exactly like srlx_per_add would; uniforms in and results out travel through a device-visible
 * pinned slot and the host spins on a completion flag the kernel stores last (no stream synchronisation).  Host pointers; results as srlx_per_sample(on_device = 0). */
int srlx_per_sample_after_adds(srlx_per_t *h, int64_t n_add, const double *add_values, int add_kind, int64_t batch_size, int64_t step, const double *uniforms,
                               int64_t n_uniforms, int64_t *out_idx, double *out_w, float *out_w32, int64_t *out_used, void *stream);
/* The same with the uniforms given as consecutive MT19937 outputs, two 32-bit words per uniform -- what `random.getrandbits(64 * n).to_bytes(8 * n, "little")` yields
 * and `random.random()` would have consumed (CPython: (a >> 5) * 2^26 + (b >> 6)) / 2^53 of consecutive outputs a, b): the host shim hands the generator's raw
 * words over instead of building n Python floats.  out_slots (or NULL): the data slot of every index (tree index - (capacity - 1)). */
int srlx_per_sample_after_adds_mt(srlx_per_t *h, int64_t n_add, const double *add_values, int add_kind, int64_t batch_size, int64_t step,
                                  const uint32_t *mt_words, int64_t n_uniforms, int64_t *out_idx, double *out_w, float *out_w32,
                                  int64_t *out_used, int64_t *out_slots, void *stream);
/* The learner's call without the launch that draws its uniforms: uniform j is what srlx_rng_uniform(seed, d_counter, n_uniforms, u)
 * would have put into u[j], and *d_counter advances the same way -- results identical to that call followed by
 * srlx_per_sample(..., u, n_uniforms, ..., on_device = 1).  Device pointers only; n_uniforms within the single-workgroup sampler's range. */
int srlx_per_sample_keyed(srlx_per_t *h, int64_t batch_size, const int64_t *d_step, uint64_t seed, int64_t *d_counter, int64_t n_uniforms,
                          int64_t *d_out_idx, double *d_out_w, float *d_out_w32, int64_t *d_out_used, void *stream);

/* update()  (:171-177, SumTree.update :81-86).  indices are tree indices from sample();
 * duplicates inside one call see each other's writes in list order, as in the reference. */
int srlx_per_update(srlx_per_t *h, int64_t n, const int64_t *indices, const void *prio, int prio_kind, int on_device,
                    void *stream);

/* update() of the n CONSECUTIVE ring slots first_slot, first_slot+1, ... (mod capacity) in slot order:
 * identical in effect to srlx_per_update with indices first_slot+capacity-1, ... but processed by the
 * bulk contiguous-run kernels (O(n log n) work instead of the general kernel's O(n^2 log n)). */
int srlx_per_set_range(srlx_per_t *h, int64_t first_slot, int64_t n, const void *prio, int prio_kind, int on_device,
                       void *stream);

/* backup()/restore()  (:179-205).  tree_host: 2*capacity-1 float64 (HOST pointers always). */
int srlx_per_backup(srlx_per_t *h, double *max_priority, int64_t *size, int64_t *write, double *tree_host);
int srlx_per_restore(srlx_per_t *h, double max_priority, int64_t size, int64_t write, const double *tree_host);
/* restore() from a backup of a different capacity (:195-205): clear, then re-add the
 * first old_size leaves of the old tree with _restore_skip. */
int srlx_per_restore_resized(srlx_per_t *h, int64_t old_capacity, int64_t old_size, const double *old_tree_host);

/* raw device view of the tree in its BLOCKED physical layout (16 doubles per 128-byte block, see
 * csrc/srlx_per.hip struct Tree); n_doubles = allocated doubles.  Use backup() for heap order. */
int srlx_per_tree_ptr(srlx_per_t *h, void **d_tree, int64_t *n_doubles);
/* device struct { double max_priority; int64 size; int64 write; int64 pad; } */
/* *d_out (device double) = max_priority as of this point of the stream: what `add(batch, None)` would use (proportional_memory.py:121-122) -- for callers that
 * assemble final priorities themselves (the engines' actor-side initial priorities, rainbow.py:389-398) and add them as SRLX_PRIO_RAW. */
int srlx_per_max_priority(srlx_per_t *h, double *d_out, void *stream);
int srlx_per_state_ptr(srlx_per_t *h, void **d_state);
/* re-read size/write from the device after HIP-graph replays that contained adds */
int srlx_per_refresh(srlx_per_t *h, void *stream);

/* SRLX_PRIO_NONE with a validity mask (uint8[n]): p = mask ? max_priority : 0.  Used by the
 * vectorised actor, where the ring position holding an episode's terminal frame has no transition. */
#define SRLX_PRIO_NONE_MASKED 4
/* add only, float32[n] ESTIMATES of |td| made on the actor side (rainbow.py:389-398; srlx_store_actor_td): x >= 0 -> p = (|x| + eps)^alpha on the widened value
 * (proportional_memory.py:124), x = -1 -> the current max_priority (priority = None), x = -2 -> 0 (the lock-step completed no item for the lane). */
#define SRLX_PRIO_EST_F32 5

/* ------------------------------------------------------------------------------------
 * Counter-based device RNG (the vectorised path has no reference stream to match; the
 * single-env plugin path keeps drawing from Python's `random`).  out[i] = u53(mix(seed, *d_counter, i));
 * afterwards *d_counter += 1.  Restated for tests in oracle/hot_path_oracle.py:rng_uniform.
 * ------------------------------------------------------------------------------------ */
int srlx_rng_uniform(uint64_t seed, int64_t *d_counter, int64_t n, double *d_out, void *stream);
/* rows frames of frame_bytes each, gathered through a frame-offset table (byte offsets from d_frame_base, -1 = none) into d_out[rows][frame_bytes], and the
 * table re-based onto d_out (d_rel_off[row] = row * frame_bytes, or -1): a sampled batch as ONE message from a replay rank to a learner rank -- the
 * device-path counterpart of the batches the reference's memory process puts on its queue (srl/base/run/play_mp_memory.py:253-351). */
int srlx_pack_frames(const uint8_t *d_frame_base, const int64_t *d_frame_off, int64_t rows, int64_t frame_bytes, uint8_t *d_out, int64_t *d_rel_off, void *stream);
/* A keyed pseudo-random permutation of 0..n-1 (int64), key = (seed, *d_counter); advances *d_counter by one.  Device state only: replayable inside a
 * HIP graph (the PPO engine's minibatch shuffles -- the role of the reference's per-epoch shuffle of the collected batch, srl/algorithms/ppo/ppo.py). */
int srlx_rng_permutation(uint64_t seed, int64_t *d_counter, int64_t n, int64_t *d_out, void *stream);
/* `count` permutations [count][n] in one launch: what `count` successive calls would have written (*d_counter + 0 .. count - 1); *d_counter += count. */
int srlx_rng_permutations(uint64_t seed, int64_t *d_counter, int64_t n, int count, int64_t *d_out, void *stream);

/* ------------------------------------------------------------------------------------
 * Device-resident transition store for E lock-stepped environments (uint8 or float32 frames).
 *
 * Replaces, for the vectorised engine: WorkerRun frame stacking + tracking ring
 * (srl/base/rl/worker_run.py:310-358,548-610, srl/base/spaces/box.py:303-312), the Rainbow
 * worker's n-step item assembly incl. terminal padding (srl/algorithms/rainbow/rainbow.py:331-400),
 * the zlib/pickle item storage of PriorityReplayBuffer (srl/rl/memories/priority_replay_buffer.py:205-217,242-243)
 * and the list->ndarray batch assembly of calc_target_q (rainbow.py:190-194).
 *
 * Layout: frames [E][ring_len][obs_elems] (one n-step window is contiguous), scalars [E][ring_len].
 * All envs share one ring position p that advances once per srlx_store_commit_step.  PER slot
 * j <-> (env j % E, item time j / E); PER capacity = E * item_len, item_len = ring_len - (n_step + window).
 * ------------------------------------------------------------------------------------ */
typedef struct srlx_store srlx_store_t;
#define SRLX_OBS_U8 0  /* uint8 frames, presented to the network as float32 u8/255 (image_processor.py:140-142) */
#define SRLX_OBS_F32 1 /* float32 observations, copied */

int srlx_store_create(srlx_store_t **out, int64_t n_envs, int64_t ring_len, int64_t obs_elems, int obs_dtype, int window,
                      int n_step, int n_actions, int reward_clip, uint64_t seed, int device);
int srlx_store_destroy(srlx_store_t *h);
int64_t srlx_store_item_len(const srlx_store_t *h);   /* ring_len - (n_step + window) */
int64_t srlx_store_per_capacity(const srlx_store_t *h); /* n_envs * item_len */
/* start: write every env's first observation (device [E][obs_elems]) at position 0 */
int srlx_store_reset_all(srlx_store_t *h, const void *d_first_obs, void *stream);
/* policy input at the current position: out float32 [E][window][obs_elems] (oldest frame first,
 * zeros before the episode start: worker_run.py:277 default states + box.py:303-312) */
int srlx_store_stack_current(srlx_store_t *h, float *d_out, void *stream);
/* one lock-step commit (rainbow.py:331-352 add_tracking): action/reward/terminated/done of the
 * step taken at p, next observation -> p+1.  Envs whose previous step was `done` are in their reset
 * step: position p (the terminal frame) is marked as holding no transition and d_next_obs is their new
 * episode's first frame.  Also emits the validity mask uint8[E] of the items that became complete
 * (position p-(n_step-1)) for srlx_per_add(..., SRLX_PRIO_NONE_MASKED). */
int srlx_store_commit_step(srlx_store_t *h, const int32_t *d_actions, const float *d_rewards, const uint8_t *d_terminated,
                           const uint8_t *d_done, const void *d_next_obs, uint8_t *d_item_mask, void *stream);
/* The same commit as ONE launch with two options (the round-4 lock-step):
 *   d_next_frame_table  int64 [E][window] or NULL: the frame-offset table of the NEXT policy pass (what srlx_store_frame_table_current would write after
 *                       the position has advanced), so the next pass needs no launch of its own for it (uint8 stores)
 *   advance             1: p advances inside the launch (srlx_store_commit_step = this with NULL, 1); 0: p stays and the caller advances it later --
 *                       srlx_store_advance, or srlx_per_set_add_counters on the position view (srlx_store_views).  Ring slot p + 1 and the scalars of p are
 *                       referenced by no stored item, so with advance = 0 the commit may run while a learner still reads the ring; p itself feeds the
 *                       learner's item lookup and must not move under it.
 *   d_bump              int64 device counter or NULL: advanced by one when the launch's last block is done (the counter of the policy generator that
 *                       srlx_qnet_forward_u8_policy only reads) */
int srlx_store_commit_step_ex(srlx_store_t *h, const int32_t *d_actions, const float *d_rewards, const uint8_t *d_terminated, const uint8_t *d_done,
                              const void *d_next_obs, uint8_t *d_item_mask, int64_t *d_next_frame_table, int advance, int64_t *d_bump, void *stream);
/* The commit of a lock-step that arrived as PACKED RECORDS (what actor ranks ship to a learner rank; replaces the unpickling of queued items in the reference's
 * trainer-side drain thread, srl/base/run/play_mp.py:248-286): environment e is lane e % envs_per_record of record e / envs_per_record; a record =
 * [action int32 x per | reward float32 x per | terminated uint8 x per | done uint8 x per | extra float32 x per x extra_floats], records record_stride bytes apart;
 * d_next_obs is the frames of all E environments in environment order.  d_est_records (or NULL): a packed buffer of the same shape whose FIRST extra field holds
 * actor-side initial-priority estimates (srl/algorithms/rainbow/rainbow.py:389-398) for the items THIS commit completes; d_est_out float32 [E] receives them, -2
 * where the commit completed no item, -1 (= "use max_priority") without d_est_records -- the array srlx_per_add(SRLX_PRIO_EST_F32) takes. */
int srlx_store_commit_step_packed(srlx_store_t *h, const uint8_t *d_records, int64_t record_stride, int64_t envs_per_record, int extra_floats, const void *d_next_obs,
                                  uint8_t *d_item_mask, const uint8_t *d_est_records, float *d_est_out, int advance, void *stream);
/* The commit at an EXPLICIT ring position (the host's count of commits) that leaves the device-resident position alone: a store whose tree add runs one lock-step
 * behind its ring commit (device/rainbow.py: the add of lock-step t rides on a side branch of update t + 1, off the lock-step's serial tail) keeps the device
 * position as the LEARNER's view -- it advances with the tree add (srlx_per_set_add_counters), so sampled leaves always resolve against the ring the tree
 * describes -- while the actors' commits run ahead of it by one.  Such a store needs one spare ring slot: srlx_store_set_item_slack(h, 1) right after
 * srlx_store_create (item_len = ring_len - (n_step + window) - slack; the slot a commit overwrites then belongs to no item even of the older view). */
int srlx_store_commit_step_at(srlx_store_t *h, int64_t position, const int32_t *d_actions, const float *d_rewards, const uint8_t *d_terminated, const uint8_t *d_done,
                              const void *d_next_obs, uint8_t *d_item_mask, int64_t *d_next_frame_table, int64_t *d_bump, void *stream);
int srlx_store_set_item_slack(srlx_store_t *h, int slack);
int srlx_store_advance(srlx_store_t *h, void *stream);
/* device views: int64 position p; uint8 needs_reset[E]; int32 step_in_episode[E] (of position p) */
int srlx_store_views(srlx_store_t *h, void **d_pos, void **d_needs_reset, void **d_step_in_ep);
/* sampled PER tree indices -> training batch (rainbow.py:190-194 + terminal padding :354-372):
 *   obs      float32 [B][n_step+1][window][obs_elems]
 *   actions  int32   [B][n_step]      rewards float32 [B][n_step]    terminated float32 [B][n_step] */
int srlx_store_gather_nstep(srlx_store_t *h, int64_t batch, const int64_t *d_tree_idx, float *d_obs, int32_t *d_actions,
                            float *d_rewards, float *d_terminated, void *stream);

/* epsilon-greedy over a batch of Q rows (rainbow.py:301-329, dqn.py:192-211):
 *   u[e][0] < eps[e] -> the floor(u[e][1]*n_valid)-th valid action, else first argmax of q with
 *   invalid actions at -inf.  invalid: uint8 [E][A] or NULL. */
int srlx_policy_epsilon_greedy(int64_t n_envs, int n_actions, const float *d_q, const float *d_eps, const double *d_u,
                               const uint8_t *d_invalid, int32_t *d_actions, void *stream);

/* synthetic 84x84-style environment batch: i.i.d. uint8 frames, reward in
 * {-1,0,1}, episodes of episode_len steps ending `terminated`. Reads the store's reset/step state. */
int srlx_synth_env_step(srlx_store_t *h, int64_t episode_len, void *d_next_obs, float *d_rewards, uint8_t *d_terminated,
                        uint8_t *d_done, void *stream);
/* The same at an EXPLICIT ring position (the host's count of commits; < 0: the device-resident position): for a store whose device position is the learner's view
 * and trails the actors' commits by one lock-step (srlx_store_commit_step_at) -- the environments belong to the actors' side and must see the actors' position. */
int srlx_synth_env_step_at(srlx_store_t *h, int64_t position, int64_t episode_len, void *d_next_obs, float *d_rewards, uint8_t *d_terminated, uint8_t *d_done,
                           void *stream);

/* Episode ledger of E device-resident environments, one call per lock-step BEFORE the commit.
 * Replaces the per-step host bookkeeping of srl/base/env/env_run.py:334-352 (step counter, episode reward sums) and
 * srl/base/run/core_play.py:200-214 (episode_rewards_list / last_episode_* at episode end).
 *   d_skip      uint8 [E] or NULL: lanes that did not take an environment step in this lock-step (the store's
 *               needs_reset view: they only received the first frame of their next episode)
 *   d_ep_return float32 [E], d_ep_len int32 [E]: running return / length of every environment (caller zeroes them once)
 *   d_ring      float32 [ring_cap][2]: (return, length) of finished episodes, appended in environment order
 *   d_totals    32 bytes: int64 episodes finished, int64 environment steps, float64 sum of returns, int64 sum of lengths */
int srlx_episode_account(int64_t n_envs, const float *d_rewards, const uint8_t *d_done, const uint8_t *d_skip, float *d_ep_return,
                         int32_t *d_ep_len, float *d_ring, int64_t ring_cap, void *d_totals, void *stream);

/* Frame preprocessing on the device (replaces srl/rl/processors/image_processor.py:104-151: cv2.cvtColor RGB2GRAY, the trimming slice,
 * cv2.resize INTER_LINEAR, the 0to1 / -1to1 normalisation) for a batch of uint8 images [n][src_h][src_w][src_channels]:
 *   to_gray      3 channels -> 1 with OpenCV's 14-bit coefficients
 *   trim_*       the window rows [top, bottom) x columns [left, right) (0, 0, src_h, src_w = no trimming)
 *   out_h/out_w  bilinear resize with OpenCV's 8-bit fixed-point arithmetic (== the window size: no resize)
 *   d_out_u8     uint8 [n][out_h][out_w][c] (what the frame ring stores) and / or
 *   d_out_f32    float32 of the same shape, normalize 0: as is, 1: / max_val ("0to1"), 2: * 2 / max_val - 1 ("-1to1") */
int srlx_image_preprocess(int64_t n_images, int src_h, int src_w, int src_channels, const uint8_t *d_src, int to_gray, int trim_top, int trim_left,
                          int trim_bottom, int trim_right, int out_h, int out_w, uint8_t *d_out_u8, float *d_out_f32, int normalize, double max_val, void *stream);

/* ------------------------------------------------------------------------------------
 * Fused learner arithmetic
 *
 * srlx_nstep_td_huber_priority replaces, in one launch: the numpy n-step/retrace target of
 * CommonInterfaceParameter.calc_target_q (srl/algorithms/rainbow/rainbow.py:226-287), the
 * selected-Q + HuberLoss(target*w, q*w) of Trainer.train (srl/algorithms/rainbow/model_torch.py:103-105),
 * its gradient w.r.t. the online Q rows (seed for backward) and the new priorities |target - q| (:113).
 *   q_on_next, q_tg_next : float32 [B][n][A]  online / target net on states s_1..s_n
 *   q_on_0               : float32 [B][A]     online net on s_0 (requires grad on the torch side)
 *   actions int32 [B][n], rewards/terminated float32 [B][n], invalid_next uint8 [B][n][A] or NULL
 *   weights float32 [B] (IS weights)
 * outputs: target f32 [B], loss f32 [1], grad_q0 f32 [B][A], priorities f32 [B]
 * ------------------------------------------------------------------------------------ */
int srlx_nstep_td_huber_priority(int64_t batch, int n_step, int n_actions, const float *d_q_on_next,
                                 const float *d_q_tg_next, const float *d_q_on_0, const int32_t *d_actions,
                                 const float *d_rewards, const float *d_terminated, const uint8_t *d_invalid_next,
                                 const float *d_weights, double discount, double retrace_h, int enable_double_dqn,
                                 int enable_rescale, float *d_target, float *d_loss, float *d_grad_q0,
                                 float *d_priorities, void *stream);
/* Same, with the online network's Q rows of s_0..s_n in ONE buffer float32 [B][n+1][A] -- what a single forward over
 * all states of the sampled items produces (row 0 = s_0, rows 1..n = s_1..s_n): no strided copies in front of it. */
int srlx_nstep_td_huber_priority_packed(int64_t batch, int n_step, int n_actions, const float *d_q_on_all,
                                        const float *d_q_tg_next, const int32_t *d_actions, const float *d_rewards,
                                        const float *d_terminated, const uint8_t *d_invalid_next, const float *d_weights,
                                        double discount, double retrace_h, int enable_double_dqn, int enable_rescale,
                                        float *d_target, float *d_loss, float *d_grad_q0, float *d_priorities, void *stream);

/* 1-step (double-)DQN target (srl/algorithms/dqn/dqn.py:144-176, rainbow_nomultisteps.py:10-43):
 * invalid next actions are masked with min(q) of the WHOLE batch, not -inf (dqn.py:160,164).
 *   q_on_next/q_tg_next f32 [B][A]; rewards f32 [B]; undone f32 [B]; out target f32 [B]
 *   f64_accum=1: dqn.py:171 (int `undone` array promotes the expression to float64, cast at :176);
 *   f64_accum=0: rainbow_nomultisteps.py:38 (all float32).
 *   d_discount_per_sample (f32 [B], NULL = use `discount`): Agent57_light's per-actor gamma,
 *   srl/algorithms/agent57_light/agent57_light.py:218-268 (`undone` is its `dones` = int(not terminated));
 *   float32 arithmetic, requires f64_accum=0. */
int srlx_dqn_target(int64_t batch, int n_actions, const float *d_q_on_next, const float *d_q_tg_next,
                    const float *d_rewards, const float *d_undone, const uint8_t *d_invalid_next, double discount,
                    const float *d_discount_per_sample, int enable_double_dqn, int enable_rescale, int f64_accum,
                    float *d_target, void *stream);

/* torch.optim.Adam(params, lr) .step() (srl/algorithms/rainbow/model_torch.py:71,109; dqn/model_torch.py:75,119) for up
 * to 16 float32 parameter tensors in one launch.  The four pointer arrays and numels are HOST arrays of n_tensors
 * entries (device pointers, 16-byte aligned; exp_avg / exp_avg_sq are the optimizer state, zero before the first step);
 * *d_step = optimizer steps already taken (device scalar, so the call is HIP-graph replayable; the caller increments it). */
int srlx_adam_step(int n_tensors, float *const *d_params, const float *const *d_grads, float *const *d_exp_avg,
                   float *const *d_exp_avg_sq, const int64_t *numels, double lr, double beta1, double beta2, double eps,
                   const int64_t *d_step, void *stream);

/* GAE reverse scan per environment (srl/algorithms/ppo/ppo.py:389-404): for each env, episodes are
 * delimited by done[t]; the last step of an episode uses delta = r - V (no bootstrap, :396-397).
 *   rewards, values, done(uint8) laid out [T][E]; out advantages f32 [T][E].
 *   last_values f32 [E] bootstraps a horizon cut that is not an episode end (NULL = no bootstrap,
 *   which is what the reference does for every episode end, truncation included). */
int srlx_gae_scan(int64_t n_envs, int64_t horizon, const float *d_rewards, const float *d_values, const uint8_t *d_done,
                  const float *d_last_values, double discount, double gae_lambda, float *d_adv, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Rank-based prioritised replay: srl/rl/memories/priority_memories/rankbased_memory.py:13-77.
 * Which ranks are drawn depends only on N, alpha and numpy's generator and stays on the host (bit-identical
 * np.random.choice); the data-dependent half -- np.argsort(-priorities[:N]) on EVERY sample (:47) -- is a descending
 * radix sort of (priority, index) pairs in HBM + a gather of the drawn ranks.  Ties: the device sort is stable
 * (lower index first), numpy's introsort is not: results agree whenever the priorities are distinct.
 *   srlx_rank_set        : add (:33-40, d_idx NULL: slots (start + i) % capacity) / update (:60-62) of float32 priorities
 *   srlx_rank_select     : out[i] = index of the ranks[i]-th largest of priorities[0..n_live)
 *   srlx_rank_priorities : device pointer of the float32 [capacity] priority array (backup / restore, :64-77)
 * ------------------------------------------------------------------------------------------------ */
typedef struct srlx_rank srlx_rank_t;
int srlx_rank_create(srlx_rank_t **out, int64_t capacity, int device);
int srlx_rank_destroy(srlx_rank_t *h);
int srlx_rank_set(srlx_rank_t *h, int64_t n, const int64_t *d_idx, const float *d_val, int64_t start, void *stream);
int srlx_rank_select(srlx_rank_t *h, int64_t n_live, int64_t n, const int64_t *d_ranks, int64_t *d_out, void *stream);
int srlx_rank_priorities(srlx_rank_t *h, float **d_prio);

/* ------------------------------------------------------------------------------------------------
 * PPO on the vectorised path (BASELINE config 5).  The reference module needs TensorFlow
 * (srl/algorithms/ppo/ppo.py:6-7) and cannot be imported in the build container: parity UNPINNED, restated
 * from the cited lines and checked against oracle/hot_path_oracle.py + torch autograd.
 *   srlx_ppo_normal_act  : ppo.py:316-339 + srl/rl/tf/distributions/normal_dist_block.py:13-20,64-74,144-149:
 *       action = loc + exp(clip(log_scale)) * N(0,1)  (keyed counter RNG, *d_counter += 1), per-dimension
 *       log-probability floored at log(1e-6) (ppo.py:322); deterministic != 0: action = loc (evaluation, :318-319).
 *       All arrays f32 [n] (n = envs x action_dim).
 *   srlx_ppo_loss_normal : compute_train_loss (ppo.py:102-169) for a Normal policy, forward + gradient seeds.
 *       loc/log_scale/action/old_logpi f32 [B][action_dim]; advantage/v/v_target/old_v f32 [B];
 *       baseline_advantage: advantage -= stop_gradient(v) (:121-122); surrogate_clip 1 = "clip" (:127-137),
 *       0 = "" (:148-149) ("kl" needs tensorflow_probability in the reference and is not offered);
 *       losses f32 [3] = policy, value, entropy as the reference reports them;
 *       grad_loc/grad_log_scale f32 [B][action_dim], grad_v f32 [B] = d(policy+value+entropy)/d(.)
 *   srlx_ppo_loss_logpi  : the same given the policy head's log-probabilities f32 [B][n_logpi]
 *       (Categorical: n_logpi = 1, the taken action); returns d loss / d new_logpi and d loss / d v.
 *   srlx_pendulum_step   : Pendulum-shaped synthetic environments (obs (cos th, sin th, thdot), one torque in
 *       [-2, 2], reward -(th^2 + .1 thdot^2 + .001 u^2), time limit `episode_len` -> done + auto-reset).
 *       state f32 [E][2], step_in_episode i32 [E], action f32 [E], obs f32 [E][3], reward f32 [E], done u8 [E].
 * ------------------------------------------------------------------------------------------------ */
int srlx_ppo_normal_act(int64_t n, const float *d_loc, const float *d_log_scale, double log_scale_min, double log_scale_max,
                        uint64_t seed, int64_t *d_counter, int deterministic, float *d_action, float *d_logprob, void *stream);
int srlx_ppo_loss_normal(int64_t batch, int action_dim, const float *d_loc, const float *d_log_scale, double log_scale_min,
                         double log_scale_max, const float *d_action, const float *d_old_logpi, const float *d_advantage,
                         const float *d_v, const float *d_v_target, const float *d_old_v, int baseline_advantage,
                         int surrogate_clip, double policy_clip_range, int enable_value_clip, double value_clip_range,
                         double value_loss_weight, double entropy_weight, float *d_losses, float *d_grad_loc,
                         float *d_grad_log_scale, float *d_grad_v, void *stream);
int srlx_ppo_loss_logpi(int64_t batch, int n_logpi, const float *d_new_logpi, const float *d_old_logpi, const float *d_advantage,
                        const float *d_v, const float *d_v_target, const float *d_old_v, int baseline_advantage,
                        int surrogate_clip, double policy_clip_range, int enable_value_clip, double value_clip_range,
                        double value_loss_weight, double entropy_weight, float *d_losses, float *d_grad_logpi, float *d_grad_v,
                        void *stream);
int srlx_pendulum_step(int64_t n_envs, float *d_state, int32_t *d_step_in_episode, const float *d_action, int64_t episode_len,
                       uint64_t seed, int64_t *d_counter, float *d_obs, float *d_reward, uint8_t *d_done, void *stream);

/* ------------------------------------------------------------------------------------------------
 * PPO's actor-critic in libsrlx (round 6): the network of srl/algorithms/ppo/ppo.py:55-99 with the reference's default blocks
 * (config.py:47: hidden (64, 64), value (64,), policy (64,); Normal head: loc + log_scale layers) -- forward, loss, backward, clip and
 * Adam -- so that an iteration is ~50 launches instead of ~1700 framework kernels.  Parameters: ONE float32 vector in the order of
 * `ActorCritic.parameters()` (weights [out][in]): w1 [64][obs], b1, w2 [64][64], b2, wv [64][64], bv, wvo [1][64], bvo, wp [64][64], bp,
 * wloc [A][64], bloc, wls [A][64], bls; obs_dim <= 8, action_dim <= 4.  float32, fmaf accumulation in ascending input order.
 *   srlx_ppo_net_param_count    : length of that vector (-1: geometry not covered).
 *   srlx_ppo_net_forward        : v f32 [n], loc / log_scale f32 [n][A] of obs f32 [n][obs_dim] (evaluation, tests, environments
 *       stepped by the host side).
 *   srlx_ppo_net_rollout        : ppo.py:316-339 (policy) + the environment + :389-404 (GAE) for `horizon` steps of `n_envs`
 *       (a multiple of 16) Pendulum-shaped environments in ONE launch: the arithmetic of srlx_ppo_normal_act (key: act_seed,
 *       *d_act_counter + t, 2 (env x A + dim)), srlx_pendulum_step (env_seed, *d_env_counter + t) and srlx_gae_scan; both counters
 *       advance by `horizon`.  env_obs f32 [E][3] = the observation it starts from and (afterwards) ends at; b_obs f32 [T+1][E][3],
 *       b_act / b_logp f32 [T][E][A], b_val / b_rew / b_adv f32 [T][E], b_done u8 [T][E], last_v f32 [E] = V(s_T),
 *       episode_return f32 [E] (running), finished f32 [2] += (sum of finished episodes' returns, their count).
 *   srlx_ppo_net_minibatch      : one minibatch of compute_train_loss (:102-169) + backward: rows i64 [minibatch] index the flattened
 *       [T x E] buffers (b_val = the rollout's values = old_v); partials f32 [srlx_ppo_net_partials_floats]: scratch; grad f32
 *       [param_count] = d loss / d parameters (sums in a fixed order: deterministic); losses f32 [3] or NULL.
 *   srlx_ppo_net_adam           : the gradient (read only) scaled by grad_scale (1 / world size behind a data-parallel all-reduce); global-norm clip
 *       (max_grad_norm, 0 = off: torch.nn.utils.clip_grad_norm_, ppo.py:240-241); torch.optim.Adam step; d_step int64 [2]: [0] += 1 (steps
 *       taken), [1] = the launch's arrival counter (zero it once; it is zero again behind every call).
 * ------------------------------------------------------------------------------------------------ */
int srlx_ppo_net_param_count(int obs_dim, int action_dim);
int srlx_ppo_net_partials_floats(int obs_dim, int action_dim);
int srlx_ppo_net_rollout_max_horizon(int action_dim); /* longest horizon srlx_ppo_net_rollout takes (its per-step records live in the workgroup's LDS) */
int srlx_ppo_net_forward(int64_t n, int obs_dim, int action_dim, const float *d_params, const float *d_obs, float *d_v, float *d_loc,
                         float *d_log_scale, void *stream);
int srlx_ppo_net_rollout(int64_t n_envs, int64_t horizon, int action_dim, const float *d_params, float *d_env_state,
                         int32_t *d_step_in_episode, float *d_env_obs, int64_t episode_len, uint64_t env_seed, int64_t *d_env_counter,
                         uint64_t act_seed, int64_t *d_act_counter, double log_scale_min, double log_scale_max, double discount,
                         double gae_lambda, float *d_b_obs, float *d_b_act, float *d_b_logp, float *d_b_val, float *d_b_rew,
                         uint8_t *d_b_done, float *d_b_adv, float *d_last_v, float *d_episode_return, float *d_finished, void *stream);
int srlx_ppo_net_minibatch(int64_t minibatch, const int64_t *d_rows, int obs_dim, int action_dim, const float *d_params,
                           const float *d_b_obs, const float *d_b_act, const float *d_b_logp, const float *d_b_adv,
                           const float *d_b_v_target, const float *d_b_val, double log_scale_min, double log_scale_max,
                           int baseline_advantage, int surrogate_clip, double policy_clip_range, int enable_value_clip,
                           double value_clip_range, double value_loss_weight, double entropy_weight, float *d_partials, float *d_grad,
                           float *d_losses, void *stream);
int srlx_ppo_net_adam(int obs_dim, int action_dim, float *d_params, float *d_grad, float *d_exp_avg, float *d_exp_avg_sq,
                      int64_t *d_step, double lr, double beta1, double beta2, double eps, double max_grad_norm, double grad_scale,
                      void *stream);

/* ------------------------------------------------------------------------------------------------
 * Never-Give-Up intrinsic reward + Agent57_light priorities
 *
 * srlx_ngu_t: one bounded episodic memory per environment, [E][emb_dim][capacity] float32 in HBM
 * (replaces `collections.deque(maxlen=episodic_memory_capacity)` of numpy vectors,
 * srl/algorithms/agent57_light/agent57_light.py:310-311).
 *   srlx_ngu_episodic_reward : agent57_light.py:473-513 for every environment at once: distances of the
 *       new embedding to every stored one, the k nearest, pseudo-count reward, then append (the oldest
 *       entry is dropped when full).  d_emb f32 [E][emb_dim]; d_reset u8 [E] (NULL = none) empties an
 *       environment's memory BEFORE the query (`on_reset`, :310-311); d_active u8 [E] (NULL = all) skips
 *       environments entirely; d_reward f32 [E].  1 <= k <= 16.
 *   srlx_ngu_reset           : empty every memory.
 *   srlx_ngu_counts          : device pointer of the int64 [E] append counters (live = min(count, capacity)).
 *   srlx_ngu_lifelong_reward : agent57_light.py:515-529, min(max(1 + mean((t-p)^2), 1), L) per row of
 *       d_target/d_train f32 [n][dim].
 *   srlx_agent57_priority    : srl/algorithms/agent57_light/model_torch.py:442,367-373:
 *       td = target - q[action]; priorities = |td_ext + beta[actor] * td_int| (d_target_int NULL: |td_ext|).
 *       d_td_ext / d_td_int (nullable) receive the signed TD errors.  d_q_ext == d_q_int == NULL: d_target_* already
 *       hold TD errors (Agent57's per-sequence means, srl/algorithms/agent57/model_torch.py:388-391).
 * ------------------------------------------------------------------------------------------------ */
typedef struct srlx_ngu srlx_ngu_t;
int srlx_ngu_create(srlx_ngu_t **out, int64_t n_envs, int emb_dim, int64_t capacity, int k, double epsilon,
                    double cluster_distance, double pseudo_counts, int device);
int srlx_ngu_destroy(srlx_ngu_t *h);
int srlx_ngu_reset(srlx_ngu_t *h, void *stream);
int srlx_ngu_counts(srlx_ngu_t *h, int64_t **d_counts);
int srlx_ngu_episodic_reward(srlx_ngu_t *h, const float *d_emb, const uint8_t *d_reset, const uint8_t *d_active,
                             float *d_reward, void *stream);
int srlx_ngu_lifelong_reward(int64_t n, int dim, const float *d_target, const float *d_train, double lifelong_max,
                             float *d_reward, void *stream);
/* Agent57 (LSTM, sequence replay) learner arithmetic after the forwards, srl/algorithms/agent57/agent57.py:301-379
 * (calc_target_q: double-DQN gains with the actor's gamma, greedy-policy retrace coefficients, backward target
 * recursion) + model_torch.py:469-492 (selected Q, HuberLoss(target*w, q*w) over [seq][batch], mean TD error):
 *   q / q_target f32 [B][S+1][A] (online / target network over the S+1 in-sequence states), actions i32 [B][S],
 *   rewards / dones f32 [B][S] (dones = 0 after a terminal step), invalid_next u8 [B][S][A] or NULL,
 *   discounts / weights f32 [B];  out: target f32 [S][B], loss f32 [1], grad_q f32 [B][S+1][A] = d loss / d q,
 *   td_mean f32 [B] = mean_t(q[a_t] - target_t);  scratch >= 2*B*S floats. */
int srlx_agent57_seq_td(int64_t batch, int seq_len, int n_actions, const float *d_q, const float *d_q_target,
                        const int32_t *d_actions, const float *d_rewards, const float *d_dones, const uint8_t *d_invalid_next,
                        const float *d_discounts, const float *d_weights, double retrace_h, int enable_double_dqn,
                        int enable_rescale, float *d_target, float *d_loss, float *d_grad_q, float *d_td_mean, float *d_scratch,
                        void *stream);
/* Sliding-window UCB meta-controller of Agent57(_light), one per device-resident environment (replaces the per-actor-process
 * controller of srl/algorithms/agent57_light/agent57_light.py:317-353).  State (caller-allocated device arrays): ring_arm int32
 * [E][window], ring_reward f32 [E][window], head int32 [E] (0), n_recent int32 [E] (0), count int32 [E][N] (ALL ONES), sum f64 [E][N]
 * (0), arm int32 [E] (-1).  Environments with done[e] != 0 (NULL = all) book `episode_reward[e]` for their current arm and choose the
 * next: arms 0..N-1 in order first, then with u[e][0] < epsilon the arm floor(u[e][1] * N), else the UCB maximiser with ties broken by
 * u[e][2].  u: f64 [E][3] uniforms. */
int srlx_agent57_ucb_step(int64_t n_envs, int n_arms, int window, int32_t *d_ring_arm, float *d_ring_reward, int32_t *d_head, int32_t *d_n_recent,
                          int32_t *d_count, double *d_sum, int32_t *d_arm, const uint8_t *d_done, const float *d_episode_reward, const double *d_u,
                          double epsilon, double beta, void *stream);
int srlx_agent57_priority(int64_t batch, int n_actions, const float *d_target_ext, const float *d_q_ext,
                          const float *d_target_int, const float *d_q_int, const int32_t *d_actions,
                          const int32_t *d_actor_idx, const float *d_beta_list, float *d_td_ext, float *d_td_int,
                          float *d_priorities, void *stream);

/* Frame-offset tables: byte offsets of uint8 frames inside the ring (-1 = all-zero history), consumed by
 * srlx_qnet_forward_u8 so that the first convolution reads the ring directly.
 *   srlx_store_obs_base            : device pointer of the frame ring (+ bytes per frame)
 *   srlx_store_frame_table_current : int64 [E][window] for the policy step at the current position
 *   srlx_store_gather_items        : like srlx_store_gather_nstep, but instead of float32 pixels it emits
 *                                    int64 [B][k_count][window] offsets of states k_begin..k_begin+k_count-1
 *   srlx_store_gather_obs          : float32 [B][k_count][window][obs_elems] pixels of a state range of the
 *                                    items located by the preceding gather_items call (same stream)
 *   srlx_store_gather_train        : the hand-written training pass's gather in ONE launch: n-step scalars as in
 *                                    gather_nstep, int64 [B][n+1][window] offsets of s_0..s_n (online network) and,
 *                                    if not NULL, int64 [B][n][window] offsets of s_1..s_n (target network) */
int srlx_store_obs_base(srlx_store_t *h, void **d_base, int64_t *frame_bytes);
/* sampled tree indices -> (environment, ring slot of the item's first transition, ring slot of the transition before it or -1 when the
 * item starts an episode): lets an engine gather per-step fields it keeps in its own [ring slot][env] arrays (Agent57's UVFA inputs) */
int srlx_store_locate(srlx_store_t *h, int64_t batch, const int64_t *d_tree_idx, int64_t *d_env, int64_t *d_slot, int64_t *d_prev_slot, void *stream);
int srlx_store_frame_table_current(srlx_store_t *h, int64_t *d_out, void *stream);
int srlx_store_gather_items(srlx_store_t *h, int64_t batch, const int64_t *d_tree_idx, int k_begin, int k_count, int64_t *d_frame_off,
                            int32_t *d_actions, float *d_rewards, float *d_terminated, void *stream);
int srlx_store_gather_train(srlx_store_t *h, int64_t batch, const int64_t *d_tree_idx, int64_t *d_frame_off_all, int64_t *d_frame_off_next,
                            int32_t *d_actions, float *d_rewards, float *d_terminated, void *stream);
/* srlx_per_sample_keyed + srlx_store_gather_train as ONE launch: the learner's draw (proportional_memory.py:131-169), then the location, n-step scalars
 * (rainbow.py:190-194 with the terminal padding of :354-372) and frame-offset tables of the drawn items of `store`, whose leaf j is this tree's slot j.  Up to 64
 * items; device pointers only; outputs as the two calls'. */
int srlx_per_sample_gather_train(srlx_per_t *h, srlx_store_t *store, int64_t batch_size, const int64_t *d_step, uint64_t seed, int64_t *d_counter, int64_t n_uniforms,
                                 int64_t *d_out_idx, float *d_out_w32, int64_t *d_out_used, int64_t *d_frame_off_all, int64_t *d_frame_off_next, int32_t *d_actions,
                                 float *d_rewards, float *d_terminated, void *stream);
/* Actor-side initial priorities (rainbow.py:389-398: a distributed worker hands `abs(calc_target_q([batch]) - q[action])` to memory.add): for the E items the LAST
 * commit completed -- PER slots first_slot .. first_slot + E - 1 (mod per_capacity), lane e's item in slot first_slot + e -- the n-step / retrace TD of
 * rainbow.py:226-287 on CACHED Q rows: d_q_hist f32 [n_step + 1][E][n_actions] is a ring over the acting passes, row (base_slot + k) mod (n_step + 1) holds
 * Q(s_k) of the item's k-th state (the online rows stand in for the target network's).  d_est[e] = |target - Q(s_0, a_0)|, or -1 when an episode ends inside
 * the item's window (its last states were never evaluated by an actor: the caller uses max_priority), or -2 when d_item_mask[e] == 0 (no item).  Feeds
 * srlx_per_add(..., SRLX_PRIO_EST_F32). */
int srlx_store_actor_td(srlx_store_t *h, int64_t first_slot, int64_t per_capacity, const float *d_q_hist, int base_slot, const uint8_t *d_item_mask, double discount,
                        double retrace_h, int enable_double_dqn, int enable_rescale, float *d_est, void *stream);
int srlx_store_gather_obs(srlx_store_t *h, int64_t batch, int k_begin, int k_count, float *d_obs, void *stream);

/* ------------------------------------------------------------------------------------
 * Q-network inference on the matrix cores: float32 in, float32 out, float32 accumulation everywhere.
 * Arithmetic: v_mfma_f32_32x32x2_f32 (products of two float32 operands), except where a float32 product is evaluated
 * on the 16x faster 16-bit pipe as a sum of EXACT partial products, each exact in the float32 accumulator.  FORWARD passes of the 84x84x4 geometry
 * (round 6): two float16 parts per operand, x = hi + lo / 2048 with hi = f16(x), lo = f16((x - hi) * 2048) -- 22 significand bits, lo a normal float16
 * wherever hi is -- on v_mfma_f32_32x32x16_f16: conv1 (the uint8 pixel is one float16, the filter two parts: two products), conv2 / conv3 and the first
 * dense layer (three of the four products, the cross terms in a 2^11-scaled accumulator; the dropped one is below 2^-24 |a b|).  BACKWARD GEMMs: three bf16
 * parts per operand on v_mfma_f32_32x32x16_bf16, six of the nine products (gradients live below float16's normal range).  Q-values agree with the
 * all-float32 pipe to float32 round-off (3e-7 of max |Q| against float64 for either; the tolerance promised against the reference is 1e-5).  The float16
 * split holds activations up to 65 504: srlx_qnet_range_flags reports a pass that met a larger one.  Environment switches (read once per process):
 * SRLX_CONV_BF16X3=1 (convolutions on three bf16 parts, float32's range), SRLX_CONV1_F32=1 (all convolutions) / SRLX_CONV23_F32=1 / SRLX_FC1_F32=1
 * keep them on the float32 pipe.
 *
 * Replaces the no-grad forwards of the reference's torch modules -- DQNImageBlock
 * (srl/rl/torch_/blocks/dqn_image_block.py:10-67) + DuelingNetworkBlock
 * (srl/rl/torch_/blocks/dueling_network.py:8-59) behind pred_q / pred_target_q
 * (srl/algorithms/rainbow/model_torch.py:55-67) -- for the actor's policy step and the
 * learner's online/target evaluation of s_1..s_n.
 *   dueling_type: 0 "average", 1 "max", 2 "" (naive)
 *   srlx_qnet_bind: 12 device pointers to float32 parameters that the kernels read IN PLACE (no copy;
 *   they must stay valid and may be updated between calls):
 *     conv1.weight [F][window][8][8]      conv1.bias
 *     conv2.weight [2F][4][4][F]  (torch channels_last memory of a [2F][F][4][4] weight)   conv2.bias
 *     conv3.weight [2F][3][3][2F] (channels_last)                                         conv3.bias
 *     fc1.weight [2*hidden][flat]: rows 0..hidden-1 = V head, the rest = A head; columns in NHWC
 *                flatten order (pixel-major, channel-minor)                                fc1.bias
 *     v2.weight [1][hidden]  v2.bias   a2.weight [A][hidden]  a2.bias
 *   srlx_qnet_forward_u8 : input = uint8 frames + offset table [batch][window]
 *   srlx_qnet_forward_f32: input = float32 [batch][window][H][W] (channels first)
 *   output float32 [batch][n_actions]
 * ------------------------------------------------------------------------------------ */
typedef struct srlx_qnet srlx_qnet_t;
int srlx_qnet_create(srlx_qnet_t **out, int in_h, int in_w, int window, int filters, int hidden, int n_actions, int dueling_type, int64_t max_batch,
                     int device);
int srlx_qnet_destroy(srlx_qnet_t *h);
int srlx_qnet_bind(srlx_qnet_t *h, const float *const *d_params);
int srlx_qnet_forward_u8(srlx_qnet_t *h, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, float *d_q, void *stream);
/* The batched Worker.policy step as one call (srl/algorithms/rainbow/rainbow.py:301-329 behind srl/base/rl/worker_run.py:316-322): srlx_qnet_forward_u8 whose head
 * kernel also selects the actions -- epsilon-greedy exactly as srlx_policy_epsilon_greedy would on the Q rows with the uniforms srlx_rng_uniform(seed, d_counter,
 * 2 * batch, u) would write (row e uses u[2 e], u[2 e + 1]).  *d_counter is READ only: advance it once per pass yourself (srlx_store_commit_step_ex's d_bump does).
 *   d_eps float32 [batch], d_invalid uint8 [batch][n_actions] or NULL, d_actions int32 [batch], d_q_copy float32 [batch][n_actions] or NULL (a second copy of Q) */
int srlx_qnet_forward_u8_policy(srlx_qnet_t *h, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, float *d_q, const float *d_eps, uint64_t seed,
                                const int64_t *d_counter, const uint8_t *d_invalid, int32_t *d_actions, float *d_q_copy, void *stream);
/* Weights handed from a learner to the actors that share its GPU without a copy on the lock-step's serial tail (the role of the parameter board the reference's
 * trainer publishes and its actors poll, srl/base/run/play_mp.py:289-303,151-165; there a pickled state_dict, here two device-resident parameter sets):
 *   srlx_qnet_actor_sets_enable   an actor handle gets two sets of everything a policy pass reads (packed convolution filters, the first dense layer as bf16
 *                                 operand planes, biases and head vectors)
 *   srlx_qnet_publish             (d_bump: an int64 device counter the launch advances by one, or NULL -- the update's step count, once every reader is done)
 *                                 packs h_src's convolution filters for its own next forwards (they then skip the packing launch until
 *                                 srlx_qnet_weights_changed) and, with h_actor, writes set `set` in the same launch; with_fc1 != 0 also splits the first dense
 *                                 layer's weight into the set's planes (the out-of-band publish: start-up, restore, a target sync)
 *   srlx_qnet_fuse_adam_fc1_planes  the fused Adam epilogue of srlx_qnet_fuse_adam_fc1 ALSO writes the updated weight as planes to d_planes_out (an actor set's,
 *                                 srlx_qnet_actor_set_planes); NULL switches it off
 *   srlx_qnet_actor_set_select    the next forwards of the actor handle read set 0 / 1 (-1: its bound parameters again)
 *   srlx_qnet_set_pack_sticky     a forward's packed filters stay valid until srlx_qnet_weights_changed (a target network: weights change at a sync only)
 *   srlx_qnet_weights_changed     the bound tensors were written by somebody else: packed filters and operand planes derived from them are stale */
int srlx_qnet_actor_sets_enable(srlx_qnet_t *h);
int srlx_qnet_actor_set_planes(srlx_qnet_t *h, int set, void **d_planes);
int srlx_qnet_actor_set_select(srlx_qnet_t *h, int set);
int srlx_qnet_publish(srlx_qnet_t *h_src, srlx_qnet_t *h_actor, int set, int with_fc1, int64_t *d_bump, void *stream);
/* The replay's priority write-back (rainbow/model_torch.py:113-114: memory.update(batch, priorities)) as part of the backward pass: with a sink set, every
 * srlx_qnet_backward_td_u8 / _backward_u8 launches srlx_per_update(per, n, d_indices, d_priorities, prio_kind, on_device = 1) first thing on its weight-gradient
 * branch -- right behind the kernel that produced the priorities, beside the gradient kernels -- instead of the caller doing it after the optimiser step.
 * The branch joins the caller's stream before the call's last launch, so the tree is updated when the backward returns to the caller's stream order.
 * per = NULL removes the sink.  Pointers are device pointers that must stay valid (captured into HIP graphs with the pass). */
int srlx_qnet_set_priority_sink(srlx_qnet_t *h, srlx_per_t *per, int64_t n, const int64_t *d_indices, const void *d_priorities, int prio_kind);
/* a caller-owned HIP event (hipEvent_t, NULL: none) the priority sink's branch waits for before its write-back: a learner rank that commits incoming
 * transitions BESIDE its update (device/dist.py: tree add on a side stream between the update's draw and its write-back) records it behind the add, so the
 * tree sees draw -> add -> write-back in that order whatever the streams do */
int srlx_qnet_set_sink_wait(srlx_qnet_t *h, void *event);
/* ... and one recorded on that branch right BEHIND the write-back: whatever must see the written-back priorities -- the NEXT update's draw, which an engine runs
 * here, beside the rest of the backward pass, instead of at the head of the next update (device/rainbow.py) -- waits for it */
int srlx_qnet_set_sink_done(srlx_qnet_t *h, void *event);
/* ... and a caller-owned stream (hipStream_t, NULL: none) the write-back is launched on instead of the weight-gradient branch, behind the kernel that produced the
 * priorities (and the wait event, if any).  For a learner rank whose write-back waits for a long ingest: the weight gradients then do not queue behind that wait.
 * The CALLER joins that stream (the event of srlx_qnet_set_sink_done is recorded behind the write-back). */
int srlx_qnet_set_sink_stream(srlx_qnet_t *h, void *stream);
/* on != 0: a backward pass enqueues the first kernel of its data-gradient chain BEFORE the launches of its weight-gradient branch (default: behind them).  No
 * effect on results or on eager execution order constraints; under stream capture it decides which of the graph's internal streams the critical chain stays on
 * (finding 14). */
int srlx_qnet_set_main_first(srlx_qnet_t *h, int on);
/* splits = 2: conv3's data-gradient GEMM of the backward pass is split over K in two (twice the workgroups, each half as long; the pad fold adds the two partial
 * slabs) -- for a handle that has the GPU to itself (a learner-only rank), where one workgroup per CU is a serial chain; 1: one pass over K (default).  The sum over K
 * associates differently: results agree to float32 rounding, not bit for bit, with the unsplit pass. */
int srlx_qnet_set_dgrad_split(srlx_qnet_t *h, int splits);

/* a caller-owned HIP event (hipEvent_t, NULL: none) recorded on the backward pass's stream right behind its head kernel: with srlx_qnet_backward_td_u8 the TD
 * targets, loss and new priorities exist from there on, so the priority write-back (srlx_per_update) can run beside the gradient kernels on another stream */
int srlx_qnet_set_td_event(srlx_qnet_t *h, void *event);
/* Measurement aid (tools/lockstep_phases.py): with a buffer set, every backward pass launches srlx_debug_stamp at fixed points -- d_buf[16] head kernel done,
 * [17] first dense layer's data gradient, [18] conv3's data gradient + fold, [19] conv2's data gradient + fold, [20] conv1's weight gradient (caller's stream);
 * [21] priority sink, [22] conv3's weight gradient + reduction, [23] conv2's, [24] the first dense layer's (weight-gradient branch); every forward pass of the
 * handle: [10] convolutions, [11] first dense layer, [12] head.  NULL: none (production). */
int srlx_qnet_set_stamp_buffer(srlx_qnet_t *h, uint64_t *d_buf);
/* Where a backward pass launches the first dense layer's Adam-fused weight gradient (the update's largest kernel: 240 MB): 0 = last on the weight-gradient
 * branch (default), 1 = first on it, 2 = on a branch of its own as soon as the data gradient has read the weights.  2 makes a captured update three
 * branches wide: use it only with the actors on a stream of another priority level (srlx_stream_create). */
int srlx_qnet_set_fc1_branch(srlx_qnet_t *h, int order);
int srlx_qnet_fuse_adam_fc1_planes(srlx_qnet_t *h, void *d_planes_out);
int srlx_qnet_set_pack_sticky(srlx_qnet_t *h, int on);
/* splits > 0: the chip-filling first-dense-layer launches of a handle with operand planes use half-CU workgroups (256 threads, 72 KB of LDS, `splits` K splits:
 * a steady stream of short workgroups that leaves room for a learner's kernels on every compute unit); 0 (default): CU-filling workgroups, the fastest form alone. */
int srlx_qnet_set_fc1_neighbour(srlx_qnet_t *h, int splits);
/* a learner's handle: operand planes also for its 96 / 128-row passes (convolution kernel writes float32 act3 AND planes; first dense layer on the half-CU kernel
 * with the staging-split GEMM's split-K shape: bit-identical results); d_weight_planes = BORROWED planes of the bound weight (an actor set's, srlx_qnet_actor_set_planes:
 * the set the last update published holds the online network's current weight) or NULL for the handle's own (srlx_qnet_refresh_fc1_planes) */
int srlx_qnet_set_planes_small(srlx_qnet_t *h, int on, const void *d_weight_planes);
int srlx_qnet_weights_changed(srlx_qnet_t *h);
/* The image block alone (DQNImageBlock, srl/rl/torch_/blocks/dqn_image_block.py:29-54: three convolutions with replicate padding + ReLU) for
 * networks whose dense part is not the dueling head of this handle (Agent57_light's UVFA Q-networks, its embedding and RND networks,
 * agent57_light/model_torch.py:35-64,98-131,160-193): d_features = float32 [batch][OH3*OW3][2*filters] -- the post-ReLU conv3 output in
 * PIXEL-major order (torch's `flatten` of the NCHW tensor is channel-major: transpose the last two axes).  Only entries 0..5 of
 * srlx_qnet_bind are read. */
int srlx_qnet_forward_convs_u8(srlx_qnet_t *h, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, float *d_features, void *stream);
/* Measurement hook: the NEXT forward on this handle records the two caller-owned HIP events (hipEvent_t) on its stream right
 * around the dominant kernel -- the one k_convnet_fused launch of the Atari geometry (the filter-packing launch before it excluded), or the
 * conv2 + conv3 launches of k_gemm<AConv> elsewhere -- so that bench.py can time that kernel live, on the stream it runs on.  NULL events
 * switch it off. */
int srlx_qnet_set_probe(srlx_qnet_t *h, void *ev_start, void *ev_end);
/* the same for the first dense layer's GEMM launch (FC1: the second-largest kernel of a lock-step) */
int srlx_qnet_set_probe_fc1(srlx_qnet_t *h, void *ev_start, void *ev_end);
/* ... and the KERNEL's own span (round 5): d_span[0] / d_span[1] (pre-set by the caller to UINT64_MAX / 0) receive min(first workgroup in) / max(last workgroup out)
 * of the device's 100 MHz wall clock for the NEXT operand-planes first-dense-layer launch -- what rocprofv3's kernel trace reports as that kernel's duration, without
 * the queue wait an event bracket on the launch stream includes when other streams hold the compute units. */
int srlx_qnet_set_fc1_span(srlx_qnet_t *h, uint64_t *d_span);
/* The first dense layer of chip-filling launches (>= 512 rows, a multiple of 128: the actors' policy pass) on PRE-SPLIT operands.  The layer evaluates
 * float32 x float32 as three exact products of two float16 parts per operand; by default both operands are split while staging, in every workgroup, every
 * K-slab.  A handle with planes enabled keeps its weight as the two parts ([K/32][unit][2][4][8 f16]: the float32's own 4 bytes per value), the convolution kernel writes its output
 * in the same form, and the GEMM runs without conversions (bit-identical results).  The planes are a CACHE of the float32 weight: after every change of
 * the weight call srlx_qnet_refresh_fc1_planes (d_src_wf = NULL: split the bound weight; otherwise split `d_src_wf`, and when d_copy_dst != NULL also
 * write the float32 values there -- the per-lock-step refresh of an actor's private copy of the online network, play_mp.py:121-165, and its planes in
 * one pass), or srlx_qnet_invalidate_fc1_planes to fall back to the staging split until the next refresh. */
int srlx_qnet_enable_fc1_planes(srlx_qnet_t *h);
int srlx_qnet_refresh_fc1_planes(srlx_qnet_t *h, const float *d_src_wf, float *d_copy_dst, void *stream);
int srlx_qnet_invalidate_fc1_planes(srlx_qnet_t *h);
/* the weight-gradient branch of srlx_qnet_backward_u8 runs on a stream of the handle's own; a caller that confines the learner to a set
 * of CUs (hipExtStreamCreateWithCUMask) hands in a stream carrying that mask instead (caller-owned, must outlive the handle's use) */
int srlx_qnet_set_side_stream(srlx_qnet_t *h, void *stream);
/* measurement aid: d_phase_stamps = device uint64 [8 waves][8] (or NULL to switch off): the fused convolution kernel's workgroup 0
 * records its shader clock at the phase boundaries (start, frames issued, staged, conv1 done, barrier, conv2 done, barrier, end) */
int srlx_qnet_set_debug(srlx_qnet_t *h, void *d_phase_stamps);
/* Round 6: the fused convolution kernel evaluates its float32 products as three exact products of two float16 parts per operand (x = hi + lo / 2048).  An
 * activation above 65 504 does not fit; the kernel then sets bit (layer - 1) of a device word of the handle instead of failing silently.  *out_bits = that word
 * (blocking device-to-host copy: call where the host has synchronised; the host side raises, device/qnet.py:check_ranges).  SRLX_CONV_BF16X3=1 selects the
 * three-part bf16 split of rounds 3-5, whose range is float32's. */
int srlx_qnet_range_flags(srlx_qnet_t *h, int *out_bits);
/* Training on the vectorised path (replaces `loss.backward()` + the framework forward it needs,
 * srl/algorithms/rainbow/model_torch.py:103-109):
 *   srlx_qnet_enable_training : from now on every forward keeps its post-ReLU hidden layer, and gradient scratch for up
 *       to max_train_batch (<= 64) samples is allocated.  Covers the DQN image block with 32 filters, dueling "average"/"".
 *   srlx_qnet_backward_u8     : given d loss / d Q  f32 [batch][n_actions] for the samples at rows 0, stride, 2*stride, ...
 *       of the LAST srlx_qnet_forward_u8 on this handle (same d_frame_base / d_frame_off), writes every parameter
 *       gradient into d_grads[12] (same order and memory layouts as srlx_qnet_bind: the torch parameters' own layouts).
 *   srlx_qnet_fuse_adam_fc1   : the first dense layer's weight [2*hidden][flat] holds 97 % of the network's parameters.  With its
 *       Adam state bound here (d_exp_avg / d_exp_avg_sq in the weight's layout, d_steps_taken = device scalar of optimiser steps
 *       already applied, the same one srlx_adam_step reads), srlx_qnet_backward_u8 applies `optimizer.step()` (model_torch.py:109)
 *       to that tensor inside its weight-gradient kernel: d_grads[6] is NOT written and the caller's srlx_adam_step must leave the
 *       tensor out.  d_exp_avg = NULL unbinds.  Refused for NoisyLinear handles (the sigma gradient needs the weight gradient). */
int srlx_qnet_enable_training(srlx_qnet_t *h, int64_t max_train_batch);
int srlx_qnet_fuse_adam_fc1(srlx_qnet_t *h, float *d_exp_avg, float *d_exp_avg_sq, double lr, double beta1, double beta2, double eps,
                            const int64_t *d_steps_taken);
/* srlx_qnet_fuse_adam_rest (round 5; after srlx_qnet_fuse_adam_fc1, whose hyper-parameters and step count it uses): `optimizer.step()` (model_torch.py:109) for
 * EVERY OTHER tensor inside the launch that finishes its gradient -- the convolution weights and biases in the epilogue of their gradient reductions, the first
 * dense layer's bias and the head's second layers in the small-vector range of the packing launch of srlx_qnet_publish -- so that no optimiser launch is left on
 * the update's tail (srlx_adam_step is then not called at all).  The three arrays are indexed like the gradient list of srlx_qnet_backward_u8 (entry 6 ignored);
 * every later backward pass must be handed the same gradient buffers, and EVERY backward pass must be followed by srlx_qnet_publish(h, ...) (which completes the
 * step; the next backward pass fails loudly otherwise).  The arithmetic is srlx_adam_step's, element by element: bit-equal results.  d_grads == NULL: off. */
int srlx_qnet_fuse_adam_rest(srlx_qnet_t *h, const float *const *d_grads, float *const *d_exp_avg, float *const *d_exp_avg_sq);
int srlx_qnet_backward_u8(srlx_qnet_t *h, int64_t batch, int64_t sample_stride, const uint8_t *d_frame_base, const int64_t *d_frame_off,
                          const float *d_grad_q, float *const *d_grads, void *stream);
/* The image block alone (networks whose dense part is not this handle's dueling head: Agent57_light's UVFA Q-networks, embedding and RND networks --
 * srl/algorithms/agent57_light/model_torch.py:18-117): given d loss / d features f32 [batch][pixels * channels] (the layout srlx_qnet_forward_convs_u8
 * returns; rows = the samples at rows 0, stride, 2*stride, ... of the LAST forward on this training-enabled handle), the ReLU mask of the kept
 * activations is applied and the convolution part of loss.backward() (model_torch.py:437-439) runs: d_grads[0..5] = conv1 w, b, conv2 w, b, conv3 w, b in
 * each parameter's own memory layout.  Same kernels, streams and fixed summation orders as the convolution half of srlx_qnet_backward_u8: run to run
 * the gradients are bit-identical. */
int srlx_qnet_backward_convs_u8(srlx_qnet_t *h, int64_t batch, int64_t sample_stride, const uint8_t *d_frame_base, const int64_t *d_frame_off,
                                const float *d_grad_features, float *const *d_grads, void *stream);
/* srlx_nstep_td_huber_priority_packed + srlx_qnet_backward_u8 in one call (one launch less on the learner's chain): the head kernel of
 * the backward pass evaluates the TD target / Huber loss / gradient seed / priorities itself (arguments and results as in
 * srlx_nstep_td_huber_priority_packed, bit-equal; d_q_on_all = the [batch][n_step+1][n_actions] output of the LAST forward on this handle,
 * i.e. sample_stride = n_step + 1) and back-propagates that seed. */
int srlx_qnet_backward_td_u8(srlx_qnet_t *h, int64_t batch, int n_step, const uint8_t *d_frame_base, const int64_t *d_frame_off, const float *d_q_on_all,
                             const float *d_q_tg_next, const int32_t *d_actions, const float *d_rewards, const float *d_terminated,
                             const uint8_t *d_invalid_next, const float *d_weights, double discount, double retrace_h, int enable_double_dqn,
                             int enable_rescale, float *d_target, float *d_loss, float *d_grad_q0, float *d_priorities, float *const *d_grads, void *stream);
int srlx_qnet_forward_f32(srlx_qnet_t *h, int64_t batch, const float *d_obs_nchw, float *d_q, void *stream);
/* Several networks over the SAME frames (round 6; Agent57_light evaluates five networks on the state a lock-step has just produced: the two UVFA Q-networks of the next
 * Worker.policy, agent57_light.py:355-363, and the embedding / RND networks of the intrinsic reward, :383-391):
 *   srlx_qnet_forward_convs_multi_u8 : the image blocks of `n` <= 8 inference handles (84 x 84 x 4, operand planes valid, batch >= 512 in multiples of 128) as ONE
 *       launch of n x batch workgroups -- one ramp and one tail instead of n; each handle is left with fresh operand planes;
 *   srlx_qnet_forward_dense_planes   : a handle's dense layers (first dense layer on the planes + its head: dueling / hidden-layer mode, UVFA terms) on those planes --
 *       together the two calls give what srlx_qnet_forward_u8 gives, bit for bit. */
int srlx_qnet_forward_convs_multi_u8(srlx_qnet_t *const *hs, int n, int64_t batch, const uint8_t *d_frame_base, const int64_t *d_frame_off, void *stream);
int srlx_qnet_forward_dense_planes(srlx_qnet_t *h, int64_t batch, float *d_q, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Agent57_light's five networks on srlx_qnet handles (round 6; SURVEY 8 a18, BASELINE configs[3]).
 * Replaces srl/algorithms/agent57_light/model_torch.py:18-117 (QNetwork with UVFA inputs, _EmbeddingNetwork, _LifelongNetwork), :244-255 (four Adams) and
 * :384-443 (_update_q) -- through round 5 the dense parts ran in torch (hipBLASLt + ATen elementwise).
 *   srlx_qnet_bind_uvfa       : a Q-network's UVFA inputs (:35-64: previous extrinsic reward, previous intrinsic reward, one-hot previous action, one-hot actor,
 *       concatenated BEHIND the image features).  Their columns of the dueling block's two first layers are kept apart from the image columns as
 *       d_wx float32 [n_cols][2 * hidden] (COLUMN-major: unit u of column c at c * 2 * hidden + u; units 0..hidden-1 = value stream, the rest = advantage stream) and
 *       enter the layer as rank-1 terms in the head kernel: reward * column, one column per one-hot.  col_* = first column of an input or -1 (input absent).
 *       Before srlx_qnet_actor_sets_enable: a published set carries a copy of the columns.
 *   srlx_qnet_set_uvfa_inputs : the per-row inputs of the next forward / backward passes, device arrays indexed by the forward's row (float32 rewards, int32 indices;
 *       NULL where the input is absent).  BORROWED until replaced (captured into HIP graphs with the passes).
 *   srlx_qnet_fuse_adam_uvfa  : gradient buffer of the columns (same layout; every backward pass writes it, beside the data-gradient chain) and, with Adam state
 *       (after srlx_qnet_fuse_adam_rest), their optimiser step in the packing launch of srlx_qnet_publish.
 *   srlx_qnet_set_td_extras   : the TD prologue of srlx_qnet_backward_td_u8 with a per-sample discount float32 [batch] (the sampled actor's gamma,
 *       agent57_light.py:263; n_step = 1 makes the prologue exactly calc_target_q :218-268) and the SIGNED TD error target - q float32 [batch] (:442).
 *   srlx_qnet_set_head_mode   : mode 1 = the handle ends behind the first dense layer (the embedding block :78 and the lifelong networks' hidden block :111): `d_q` of
 *       the forward entry points is float32 [batch][out_cols], the first out_cols post-ReLU units (a layer narrower than the GEMM's tile is padded with zero rows);
 *       d_ln_w / d_ln_b != NULL: nn.LayerNorm over all 2 * hidden units first (:112, inference only).  srlx_qnet_backward_u8 then takes d loss / d that output
 *       [batch][out_cols]; d_grads[8..11] are ignored (bind small dummies).  mode 0 = the dueling head.
 * ------------------------------------------------------------------------------------------------ */
int srlx_qnet_bind_uvfa(srlx_qnet_t *h, const float *d_wx, int n_cols, int col_ext, int col_int, int col_action, int n_action_in, int col_actor, int n_actor);
int srlx_qnet_set_uvfa_inputs(srlx_qnet_t *h, const float *d_r_ext, const float *d_r_int, const int32_t *d_action, const int32_t *d_actor);
int srlx_qnet_fuse_adam_uvfa(srlx_qnet_t *h, float *d_grad_wx, float *d_exp_avg, float *d_exp_avg_sq);
int srlx_qnet_set_td_extras(srlx_qnet_t *h, const float *d_discount_per_sample, float *d_td_signed);
int srlx_qnet_set_head_mode(srlx_qnet_t *h, int mode, int out_cols, const float *d_ln_w, const float *d_ln_b, double ln_eps);
/* Agent57_light around its networks, one launch each (csrc/srlx_agent57.hip):
 *   srlx_agent57_policy         : Worker.policy (agent57_light.py:355-375) for E environments: q = q_ext + beta[arm] * q_int, epsilon[arm]-greedy with the keyed uniforms
 *       (seed, *d_counter, 2 e / 2 e + 1) like srlx_qnet_forward_u8_policy (the counter is only read).  d_arm = NULL: evaluation (test_beta / test_epsilon, :294-297).
 *   srlx_agent57_post_step      : Worker.on_step's bookkeeping (:377-432): the item fields the frame store does not keep (rows of the caller's [ring slot][env] arrays:
 *       actor, previous action, previous rewards, intrinsic reward = episodic * lifelong :383-391), then previous action / rewards and the episode reward of the lanes
 *       that took a step (d_reset_lane == 0).
 *   srlx_agent57_begin_episodes : Worker.on_reset (:288-311) for the lanes in d_done (NULL: all): random previous action (keyed: seed, *d_counter, lane), zero previous
 *       rewards and episode reward; d_reset_lane (or NULL) := d_done, d_live_lane (or NULL) := !d_done.
 *   srlx_agent57_gather_inputs  : change_batches_format (:165-216) for a drawn batch located by srlx_store_locate: the online network's rows interleaved (2 b = s_0 with the
 *       inputs of model_torch.py:427-433, 2 b + 1 = s_1 with :294-299), the target network's rows (s_1), discount[b] = discount_list[actor] (:287), r_int[b].
 *   srlx_agent57_emb_tail       : the embedding network behind its two embeddings (model_torch.py:87-95) + MSE against the one-hot action (:343) + backward + Adam (:345-347)
 *       in ONE single-workgroup launch.  d_emb float32 [2 batch][emb_dim] (rows 2 b = f(s), 2 b + 1 = f(s')); the four pointer tables are HOST arrays of 6 device
 *       pointers: out_block weight [hidden][2 emb_dim], its bias, LayerNorm weight, bias, out_block_out1 weight [n_actions][hidden], its bias; d_exp_avg = NULL:
 *       gradients only.  d_steps_taken = optimiser steps already applied (device scalar).  Out: d_loss [1], d_grad_emb [2 batch][emb_dim].
 *   srlx_agent57_rnd_tail       : the predictor's LayerNorm (:112,116) + MSE against the target network's output (:357) + backward + Adam for the LayerNorm parameters;
 *       d_hidden / d_target = the predictor's post-ReLU hidden layer / the target network's output, `batch` rows of `dim` floats row_stride floats apart; d_mirror_* (or NULL) receive the updated LayerNorm parameters once more (the copy the actors
 *       read next). */
int srlx_agent57_policy(int64_t n_envs, int n_actions, const float *d_q_ext, const float *d_q_int, const int32_t *d_arm, const float *d_beta_list, const float *d_eps_list,
                        double test_beta, double test_epsilon, uint64_t seed, const int64_t *d_counter, int32_t *d_actions, float *d_q_out, void *stream);
int srlx_agent57_post_step(int64_t n_envs, const int32_t *d_actions, const int32_t *d_arm, const float *d_rewards, const uint8_t *d_reset_lane, const float *d_episodic,
                           const float *d_lifelong, int32_t *d_prev_action, float *d_prev_r_ext, float *d_prev_r_int, float *d_episode_reward, float *d_x_r_int,
                           float *d_x_prev_r_ext, float *d_x_prev_r_int, int32_t *d_x_actor, int32_t *d_x_prev_action, void *stream);
int srlx_agent57_begin_episodes(int64_t n_envs, int n_actions, const uint8_t *d_done, uint64_t seed, const int64_t *d_counter, int32_t *d_prev_action, float *d_prev_r_ext,
                                float *d_prev_r_int, float *d_episode_reward, uint8_t *d_reset_lane, uint8_t *d_live_lane, void *stream);
int srlx_agent57_gather_inputs(int64_t batch, int64_t n_envs, const int64_t *d_loc_env, const int64_t *d_loc_slot, const int32_t *d_actions, const float *d_rewards,
                               const float *d_x_r_int, const float *d_x_prev_r_ext, const float *d_x_prev_r_int, const int32_t *d_x_actor, const int32_t *d_x_prev_action,
                               const float *d_discount_list, float *d_on_r_ext, float *d_on_r_int, int32_t *d_on_action, int32_t *d_on_actor, float *d_tg_r_ext,
                               float *d_tg_r_int, int32_t *d_tg_action, int32_t *d_tg_actor, float *d_discount, float *d_r_int, void *stream);
/* Multi-GPU (BASELINE configs[3]: 7 actor GPUs + 1 learner GPU; replaces the pickled 11-field items of srl/base/run/play_mp.py:76-118 and their unpickling in the
 * trainer's drain thread, :248-286):
 *   srlx_agent57_pack_record   : an actor rank's lock-step as ONE packed record uint8 [(10 + 4 * 5) * n_envs] in the layout srlx_store_commit_step_packed takes with
 *       extra_floats = 5: [action | reward | terminated | done | (intrinsic reward, arm, previous action, previous extrinsic reward, previous intrinsic reward) per lane].
 *   srlx_agent57_unpack_fields : the learner rank's inverse for the five fields: environment g = lane g % envs_per_record of record g / envs_per_record; they land
 *       in row (*d_position mod ring_len) of the learner's [ring slot][environment] arrays -- the slot the ring commit of the same slab writes; the position is read
 *       on the device (srlx_store_views), so the launch replays inside a captured update. */
int srlx_agent57_pack_record(int64_t n_envs, const int32_t *d_actions, const float *d_rewards, const uint8_t *d_terminated, const uint8_t *d_done, const float *d_x_r_int,
                             const int32_t *d_x_actor, const int32_t *d_x_prev_action, const float *d_x_prev_r_ext, const float *d_x_prev_r_int, uint8_t *d_record, void *stream);
int srlx_agent57_unpack_fields(int64_t n_records, int64_t envs_per_record, const uint8_t *d_records, int64_t record_stride, const int64_t *d_position, int64_t ring_len,
                               float *d_x_r_int, int32_t *d_x_actor, int32_t *d_x_prev_action, float *d_x_prev_r_ext, float *d_x_prev_r_int, void *stream);
int srlx_agent57_emb_tail(int64_t batch, int emb_dim, int hidden, int n_actions, const float *d_emb, const int32_t *d_actions, float *const *d_params, float *const *d_grads,
                          float *const *d_exp_avg, float *const *d_exp_avg_sq, double ln_eps, double lr, double beta1, double beta2, double eps, const int64_t *d_steps_taken,
                          float *d_loss, float *d_grad_emb, void *stream);
int srlx_agent57_rnd_tail(int64_t batch, int dim, int64_t row_stride, const float *d_hidden, const float *d_target, float *d_ln_w, float *d_ln_b, float *d_grad_ln_w, float *d_grad_ln_b,
                          float *d_exp_avg_w, float *d_exp_avg_sq_w, float *d_exp_avg_b, float *d_exp_avg_sq_b, float *d_mirror_w, float *d_mirror_b, double ln_eps, double lr,
                          double beta1, double beta2, double eps, const int64_t *d_steps_taken, float *d_loss, float *d_grad_hidden, void *stream);

/* NoisyLinear dense layers (replaces srl/rl/torch_/modules/noisy_linear.py:26-52, the dense layers of the reference's
 * rainbow.Config.set_atari_config(), rainbow.py:116-148): W = w_mu + w_sigma * eps, b = b_mu + b_sigma * eps, eps ~ N(0,1)
 * independent per element, ONE draw per forward call shared by all its rows.
 *   srlx_qnet_bind_noisy      : d_sigma[6] = sigma tensors of {fc1 weight [2*hidden][flat], fc1 bias, v2 weight, v2 bias, a2 weight,
 *       a2 bias}, same layouts as the mu tensors given to srlx_qnet_bind (entries 6..11).  From then on every forward first
 *       materialises the effective tensors of a fresh draw (keyed counter generator: eps is a function of (seed, draw id, tensor,
 *       element), so the backward pass regenerates it).
 *   srlx_qnet_redraw_rows     : re-evaluates the dense layers with ANOTHER fresh draw for rows 0, stride, 2*stride, ... of the last
 *       forward and overwrites their q rows -- the reference's train step evaluates q_online(s_1..s_n) and q_online(s_0) in two
 *       forward calls, i.e. under two draws (rainbow.py:220, model_torch.py:103); the backward pass then belongs to this draw.
 *   srlx_qnet_bind_noisy_grads: gradient tensors of the six sigmas; srlx_qnet_backward_u8 fills them (d_grads[6..11] are then the
 *       gradients of the mu tensors).
 *   srlx_qnet_noisy_effective : copies an effective tensor (0 fc1 weight .. 5 a2 bias) into d_out (device, may be NULL) and reports
 *       its element count and the id of the draw it holds (tests, diagnostics). */
int srlx_qnet_bind_noisy(srlx_qnet_t *h, const float *const *d_sigma, uint64_t seed);
int srlx_qnet_bind_noisy_grads(srlx_qnet_t *h, float *const *d_grad_sigma);
int srlx_qnet_redraw_rows(srlx_qnet_t *h, int64_t rows, int64_t row_stride, float *d_q, void *stream);
int srlx_qnet_noisy_effective(srlx_qnet_t *h, int which, float *d_out, int64_t *n_elems, int64_t *draw_id, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SRLX_H */
