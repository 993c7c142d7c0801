"""oracle/hot_path_oracle.py -- TEST INFRASTRUCTURE ONLY (not product code).

numpy restatements of the reference's host-side arithmetic on the rollout -> replay -> train path,
each following the cited reference lines (paths relative to the reference repository root).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Parity status: PINNED.  tests/test_hot_path_oracle_golden.py checks every function below against vectors
recorded from the imported reference by oracle/gen_golden_algo.py (tests/golden/target_q_*.npz,
train_step_*.npz, dqn_target_*.npz, functions.npz, rollout_items_*.npz).  `gae` follows
srl/algorithms/ppo/ppo.py:389-404, whose module needs TensorFlow and cannot be imported here:
parity UNPINNED for that one function (restated from source only).
The NGU / Agent57_light functions at the end (episodic and lifelong novelty, per-actor-discount target,
mixed priority) are PINNED by tests/golden/ngu_*.npz, agent57_light_target_*.npz and
train_step_agent57_light.npz (oracle/gen_golden_agent57.py).
"""
import numpy as np

# ------------------------------------------------------------------------------------------
# counter RNG of the vectorised path (definition; mirrored by csrc/srlx_rollout.hip)
# ------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    z = np.asarray(z, np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def rng_u64(seed, a, b):
    with np.errstate(over="ignore"):
        s = np.uint64(seed) + np.asarray(a, np.uint64) * np.uint64(0xD1342543DE82EF95)
        return _mix64(_mix64(s) + np.asarray(b, np.uint64) * np.uint64(0xAEF17502108EF2D9))


def u53(x):
    return (np.asarray(x, np.uint64) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def rng_uniform(seed, counter, n):
    return u53(rng_u64(seed, np.uint64(counter), np.arange(n, dtype=np.uint64)))


# ------------------------------------------------------------------------------------------
# srl/rl/functions.py:10-17
# ------------------------------------------------------------------------------------------
def rescaling(x, eps=0.001):
    return np.sign(x) * (np.sqrt(np.abs(x) + 1.0) - 1.0) + eps * x


def inverse_rescaling(x, eps=0.001):
    n = np.sqrt(1.0 + 4.0 * eps * (np.abs(x) + 1.0 + eps)) - 1.0
    n = n / (2.0 * eps)
    return np.sign(x) * ((n**2) - 1.0)


# ------------------------------------------------------------------------------------------
# srl/algorithms/rainbow/rainbow.py:185-287  CommonInterfaceParameter.calc_target_q, the part after
# the two network forwards.  Array form of the nested-list inputs:
#   q_online : (B, n, A) online net on s_1..s_n   (double DQN; (B, n-1, A) otherwise, :208-213)
#   q_target : (B, n, A) target net on s_1..s_n
#   actions  : (B, n) int   a_0..a_{n-1}          (one-hot in the reference)
#   reward, done : (B, n) float32                 invalid : (B, n, A) bool (next_invalid_actions)
# ------------------------------------------------------------------------------------------
def nstep_target(q_online, q_target, actions, reward, done, invalid, discount, retrace_h, double_dqn, rescale, np_dtype=np.float32):
    q_online = np.array(q_online, np_dtype, copy=True)
    q_target = np.array(q_target, np_dtype, copy=True)
    B, n, A = q_target.shape
    reward = np.asarray(reward, np_dtype)
    done = np.asarray(done, np_dtype)
    multi_discounts = np.tile(np.array([discount**k for k in range(n)], dtype=np_dtype), (B, 1))  # :182,187
    onehot = np.eye(A, dtype=np_dtype)[np.asarray(actions)]  # (B, n, A)
    n_action = onehot[:, 1:, :]  # :200

    q = np.sum(q_online[:, : n - 1, :] * n_action, axis=2)  # :231
    q = np.insert(q, 0, 0, axis=1)  # :233

    if invalid is None:
        invalid = np.zeros((B, n, A), bool)
    idx1, idx2, idx3 = np.nonzero(invalid)  # :241-243
    if double_dqn:
        q_online[idx1, idx2, idx3] = -np.inf  # :245-247
        n_act_idx = np.argmax(q_online, axis=2)
    else:
        q_target[idx1, idx2, idx3] = -np.inf  # :248-250
        n_act_idx = np.argmax(q_target, axis=2)
    maxq = np.take_along_axis(q_target, np.expand_dims(n_act_idx, axis=2), axis=2)  # :251
    maxq = np.squeeze(maxq, axis=2)
    if rescale:
        maxq = inverse_rescaling(maxq)  # :255
    gains = reward + (1 - done) * discount * maxq  # :258
    if rescale:
        gains = rescaling(gains)  # :261
    td_errors = gains - q  # :263

    pi_probs = np.argmax(n_action, axis=2) == n_act_idx[:, 1:]  # :268
    pi_probs = np.transpose(pi_probs, (1, 0))  # :272
    retrace_list = [np.ones((B,))]
    retrace = np.ones((B,))
    for k in range(n - 1):
        retrace *= retrace_h * pi_probs[k]  # :280
        retrace_list.append(retrace.copy())
    retrace_list = np.asarray(retrace_list).transpose((1, 0))  # :284
    target_q = np.sum(td_errors * multi_discounts * retrace_list, axis=1, dtype=np_dtype)  # :285
    return target_q


# ------------------------------------------------------------------------------------------
# srl/algorithms/rainbow/model_torch.py:103-105,113 (and dqn/model_torch.py:112-114,122):
#   q = sum(Q(s) * onehot); loss = HuberLoss()(target*w, q*w); priorities = |target - q|
# plus d loss / d Q(s) (what autograd hands to the network's backward).
# ------------------------------------------------------------------------------------------
def huber_loss_grad_priority(q_all, a0, target, weights):
    q_all = np.asarray(q_all, np.float32)
    target = np.asarray(target, np.float32)
    w = np.asarray(weights, np.float32)
    B, A = q_all.shape
    q = q_all[np.arange(B), a0]
    tw, qw = target * w, q * w
    diff = (tw - qw).astype(np.float64)
    z = np.abs(diff)
    loss = np.where(z < 1.0, 0.5 * z * z, z - 0.5).mean()
    grad = np.zeros((B, A), np.float32)
    grad[np.arange(B), a0] = (-(w.astype(np.float64) * np.clip(diff, -1, 1)) / B).astype(np.float32)
    return np.float32(loss), grad, np.abs(target - q)


# ------------------------------------------------------------------------------------------
# srl/algorithms/dqn/dqn.py:144-176 (f64_accum=True: `undone` is an int array there, so numpy
# promotes the target expression to float64) and
# srl/algorithms/rainbow/rainbow_nomultisteps.py:10-43 (f64_accum=False: all float32).
# ------------------------------------------------------------------------------------------
def dqn_target(q_online_next, q_target_next, reward, undone, invalid, discount, double_dqn, rescale, f64_accum):
    n_q_target = np.array(q_target_next, np.float32, copy=True)
    B, A = n_q_target.shape
    reward = np.asarray(reward, np.float32)
    undone = np.asarray(undone, np.int64 if f64_accum else np.float32)
    i1, i2 = np.nonzero(invalid) if invalid is not None else (np.array([], int), np.array([], int))
    if double_dqn:
        n_q = np.array(q_online_next, np.float32, copy=True)
        n_q[i1, i2] = np.min(n_q)  # dqn.py:160
        n_act_idx = np.argmax(n_q, axis=1)
        maxq = n_q_target[np.arange(B), n_act_idx]
    else:
        n_q_target[i1, i2] = np.min(n_q_target)  # dqn.py:164
        maxq = np.max(n_q_target, axis=1)
    if rescale:
        maxq = inverse_rescaling(maxq)
    target_q = reward + undone * discount * maxq  # dqn.py:171
    if rescale:
        target_q = rescaling(target_q)
    return target_q.astype(np.float32)


# ------------------------------------------------------------------------------------------
# srl/algorithms/ppo/ppo.py:389-404 generalised to a [T][E] lock-step rollout: every `done` step is
# the last step of an episode (delta = r - V, no bootstrap); a horizon cut may bootstrap from
# last_values (no reference equivalent; None = treat like an episode end).  Parity UNPINNED.
# ------------------------------------------------------------------------------------------
def gae(rewards, values, done, last_values, discount, lam):
    rewards = np.asarray(rewards, np.float32)
    values = np.asarray(values, np.float32)
    T, E = rewards.shape
    adv = np.zeros((T, E), np.float32)
    g = np.float32(discount)
    gl = np.float32(discount * lam)
    for e in range(E):
        gae_v = np.float32(0)
        for i in reversed(range(T)):
            if done[i, e]:
                delta = rewards[i, e] - values[i, e]
                gae_v = np.float32(0)
            elif i == T - 1:
                if last_values is None:
                    delta = rewards[i, e] - values[i, e]
                else:
                    delta = (rewards[i, e] + g * np.float32(last_values[e])) - values[i, e]
            else:
                delta = (rewards[i, e] + g * values[i + 1, e]) - values[i, e]
            gae_v = np.float32(delta + gl * gae_v)
            adv[i, e] = gae_v
    return adv


# ------------------------------------------------------------------------------------------
# epsilon-greedy (srl/algorithms/rainbow/rainbow.py:301-329) with explicit uniforms; the random
# branch picks the floor(u2 * n_valid)-th valid action (the reference uses random.choice on the
# same list; a device stream cannot reproduce Python's _randbelow bit stream)
# ------------------------------------------------------------------------------------------
def epsilon_greedy(q, eps, u, invalid=None):
    q = np.asarray(q, np.float32)
    E, A = q.shape
    out = np.zeros(E, np.int32)
    for e in range(E):
        inv = np.zeros(A, bool) if invalid is None else np.asarray(invalid[e], bool)
        if u[e, 0] < float(eps[e]):
            valid = [a for a in range(A) if not inv[a]]
            k = min(int(u[e, 1] * len(valid)), len(valid) - 1)
            out[e] = valid[k]
        else:
            qq = q[e].copy()
            qq[inv] = -np.inf
            out[e] = int(np.argmax(qq))
    return out


# ------------------------------------------------------------------------------------------
# Lock-step transition store: a plain-Python model of E environments following the reference's
# per-environment semantics --
#   frame stacking with zero history at episode start  (srl/base/rl/worker_run.py:277,316-322;
#       srl/base/spaces/box.py:303-312; oldest frame first)
#   n-step item = n+1 stacked states, n (action, reward, terminated) rows, and after an episode end
#       rows repeat the terminal state with reward 0 / terminated 1 / a random action
#       (srl/algorithms/rainbow/rainbow.py:331-372)
#   reward clipping to {-1,0,1} (rainbow.py:337-343)
# ------------------------------------------------------------------------------------------
class StoreOracle:
    TERM, DONE, INVALID = 1, 2, 4

    def __init__(self, n_envs, ring_len, obs_elems, window, n_step, n_actions, reward_clip, seed, u8=True):
        self.E, self.L, self.F, self.W, self.n, self.A = n_envs, ring_len, obs_elems, window, n_step, n_actions
        self.reward_clip, self.seed, self.u8 = reward_clip, np.uint64(seed), u8
        self.item_len = ring_len - (n_step + window)
        self.obs = np.zeros((n_envs, ring_len, obs_elems), np.uint8 if u8 else np.float32)
        self.action = np.zeros((n_envs, ring_len), np.int32)
        self.reward = np.zeros((n_envs, ring_len), np.float32)
        self.flags = np.zeros((n_envs, ring_len), np.uint8)
        self.step_in_ep = np.zeros((n_envs, ring_len), np.int32)
        self.needs_reset = np.zeros(n_envs, bool)
        self.pos = 0

    def reset_all(self, first_obs):
        self.pos = 0
        self.obs[:, 0] = first_obs
        self.step_in_ep[:, 0] = 0
        self.flags[:, 0] = 0
        self.needs_reset[:] = False

    def _norm(self, frame):
        return frame.astype(np.float32) / np.float32(255.0) if self.u8 else frame.astype(np.float32)

    def stack_at(self, e, x):
        out = np.zeros((self.W, self.F), np.float32)
        sie = self.step_in_ep[e, x % self.L]
        for c in range(self.W):
            back = self.W - 1 - c
            if back <= sie:
                out[c] = self._norm(self.obs[e, (x - back) % self.L])
        return out

    def stack_current(self):
        return np.stack([self.stack_at(e, self.pos) for e in range(self.E)])

    def commit_step(self, actions, rewards, terminated, done, next_obs):
        p, L = self.pos, self.L
        mask = np.zeros(self.E, np.uint8)
        for e in range(self.E):
            r, r1 = p % L, (p + 1) % L
            self.obs[e, r1] = next_obs[e]
            if self.needs_reset[e]:
                self.flags[e, r] = self.INVALID
                self.action[e, r] = 0
                self.reward[e, r] = 0
                self.step_in_ep[e, r1] = 0
                self.needs_reset[e] = False
            else:
                rew = np.float32(rewards[e])
                if self.reward_clip:
                    rew = np.float32(-1 if rew < 0 else (1 if rew > 0 else 0))
                self.flags[e, r] = (self.TERM if terminated[e] else 0) | (self.DONE if done[e] else 0)
                self.action[e, r] = actions[e]
                self.reward[e, r] = rew
                self.step_in_ep[e, r1] = self.step_in_ep[e, r] + 1
                self.needs_reset[e] = bool(done[e])
            q = p - (self.n - 1)
            mask[e] = 1 if (q >= 0 and not (self.flags[e, q % L] & self.INVALID)) else 0
        self.pos += 1
        return mask

    def locate(self, tree_idx):
        N = self.E * self.item_len
        j = min(max(int(tree_idx) - (N - 1), 0), N - 1)
        e, tau = j % self.E, j // self.E
        p_last = self.pos - 1
        p_add = p_last - ((p_last - tau) % self.item_len)
        return e, max(p_add - (self.n - 1), 0)

    def gather_item(self, e, q):
        n, L = self.n, self.L
        obs = np.zeros((n + 1, self.W, self.F), np.float32)
        actions = np.zeros(n, np.int32)
        rewards = np.zeros(n, np.float32)
        terminated = np.zeros(n, np.float32)
        jd = n
        for k in range(n):
            if self.flags[e, (q + k) % L] & self.DONE:
                jd = k
                break
        for k in range(n + 1):
            obs[k] = self.stack_at(e, q + min(k, jd + 1))
        for k in range(n):
            r = (q + k) % L
            if k <= jd:
                actions[k] = self.action[e, r]
                rewards[k] = self.reward[e, r]
                terminated[k] = 1.0 if self.flags[e, r] & self.TERM else 0.0
            else:
                key = np.uint64(e * 0x100000000 + (q & 0xFFFFFFFF))
                actions[k] = int(rng_u64(self.seed ^ np.uint64(0x70616464), key, np.uint64(k)) % np.uint64(self.A))
                rewards[k] = 0.0
                terminated[k] = 1.0
        return obs, actions, rewards, terminated, jd

    def gather_nstep(self, tree_indices):
        outs = [self.gather_item(*self.locate(ti))[:4] for ti in tree_indices]
        return tuple(np.stack([o[i] for o in outs]) for i in range(4))


def synth_env_step(store: StoreOracle, episode_len):
    """Mirror of srlx_synth_env_step (synthetic workload definition)."""
    E, F, p = store.E, store.F, store.pos
    next_obs = np.zeros((E, F), np.uint8 if store.u8 else np.float32)
    rewards = np.zeros(E, np.float32)
    term = np.zeros(E, np.uint8)
    done = np.zeros(E, np.uint8)
    for e in range(E):
        key = np.uint64(e * 0x100000000 + ((p + 1) & 0xFFFFFFFF))
        if store.u8:
            words = rng_u64(store.seed, key, np.arange((F + 7) // 8, dtype=np.uint64))
            next_obs[e] = words.view(np.uint8)[:F]
        else:
            next_obs[e] = (2.0 * u53(rng_u64(store.seed, key, np.arange(F, dtype=np.uint64))) - 1.0).astype(np.float32)
        if store.needs_reset[e]:
            continue
        key0 = np.uint64(e * 0x100000000 + (p & 0xFFFFFFFF))
        rewards[e] = float(int(rng_u64(store.seed ^ np.uint64(0x726577), key0, np.uint64(0)) % np.uint64(3)) - 1)
        d = 1 if store.step_in_ep[e, p % store.L] + 1 >= episode_len else 0
        term[e] = d
        done[e] = d
    return next_obs, rewards, term, done


# ------------------------------------------------------------------------------------------
# NGU episodic novelty, srl/algorithms/agent57_light/agent57_light.py:473-513.  One memory per
# environment; a bounded deque (oldest entry dropped).  All arithmetic is numpy float32 like the
# reference (np.linalg.norm of a float32 vector = sqrt(x.dot(x)); np.sort/np.mean of float32).
# ------------------------------------------------------------------------------------------
def ngu_episodic_from_distances(dist, k, epsilon, cluster_distance, pseudo_counts):
    """agent57_light.py:493-513 given the float32 distances to every stored embedding.  np.mean / np.sum of
    a short float32 vector add in numpy's pairwise order (8 partial sums for n >= 8, then the tail; a plain
    loop for n < 8) -- the device kernel follows the same order."""
    near = np.sort(np.asarray(dist, np.float32))[:k]
    ave = np.mean(near)  # :497
    dn = near if ave == 0.0 else near / ave  # :498-502
    dn = np.maximum(dn - cluster_distance, 0)  # :505 (python floats are weak under NEP 50: stays float32)
    dn = epsilon / (dn + epsilon)  # :508
    n_visits = np.sum(dn)
    return np.float32(1 / (np.sqrt(n_visits) + pseudo_counts))  # :512


class EpisodicMemoryOracle:
    def __init__(self, capacity, k=10, epsilon=0.001, cluster_distance=0.008, pseudo_counts=0.1):
        self.capacity, self.k, self.epsilon, self.cluster_distance, self.c = capacity, k, epsilon, cluster_distance, pseudo_counts
        self.entries = []

    def reset(self):
        self.entries = []

    def step(self, emb, exact_dot=True):
        emb = np.asarray(emb, np.float32)
        if not self.entries:  # :483-485
            self.entries.append(emb.copy())
            return np.float32(1 / self.c)
        if exact_dot:  # per entry x.dot(x) like np.linalg.norm(m - cont_state, ord=2), :488
            dist = np.array([np.sqrt((m - emb).dot(m - emb)) for m in self.entries], np.float32)
        else:
            diff = np.stack(self.entries) - emb
            dist = np.sqrt(np.einsum("ij,ij->i", diff, diff)).astype(np.float32)
        self.entries.append(emb.copy())  # :491
        if len(self.entries) > self.capacity:
            self.entries.pop(0)
        return ngu_episodic_from_distances(dist, self.k, self.epsilon, self.cluster_distance, self.c)


def ngu_lifelong_reward(target, train, lifelong_max):
    """agent57_light.py:515-529 for a batch: 1 + mean squared RND error, clipped to [1, L]."""
    target, train = np.asarray(target, np.float32), np.asarray(train, np.float32)
    err = np.array([np.square(t - p).mean() for t, p in zip(target, train)], np.float32)
    return np.minimum(np.maximum(np.float32(1) + err, np.float32(1)), np.float32(lifelong_max)).astype(np.float32)


def agent57_target(q_online_next, q_target_next, rewards, dones, batch_discount, invalid, double_dqn, rescale):
    """agent57_light.py:218-268: 1-step (double-)DQN target with a per-sample discount (the actor's gamma);
    `dones` is the reference's continue flag int(not terminated) as float32; everything stays float32."""
    return dqn_target(q_online_next, q_target_next, rewards, dones, invalid, np.asarray(batch_discount, np.float32), double_dqn, rescale, False)


def agent57_priority(td_ext, td_int, batch_beta):
    """agent57_light/model_torch.py:367-373: priorities = |td_ext + beta_actor * td_int| (float32)."""
    return np.abs(np.asarray(td_ext, np.float32) + np.asarray(batch_beta, np.float32) * np.asarray(td_int, np.float32)).astype(np.float32)


# ------------------------------------------------------------------------------------------
# PPO (srl/algorithms/ppo/ppo.py; the module imports TensorFlow and cannot be imported here: parity UNPINNED,
# restated from the cited lines in float32 numpy)
# ------------------------------------------------------------------------------------------
def normal_logprob(x, loc, log_scale):
    """srl/rl/tf/distributions/normal_dist_block.py:13-20"""
    x, loc, log_scale = (np.asarray(t, np.float32) for t in (x, loc, log_scale))
    return (np.float32(-0.5 * np.log(2 * np.pi)) - log_scale - np.float32(0.5) * ((x - loc) / np.exp(log_scale)) ** 2).astype(np.float32)


def ppo_loss(new_logpi, old_logpi, advantage, v, v_target, old_v, baseline_advantage, surrogate_clip, policy_clip_range, enable_value_clip,
             value_clip_range, value_loss_weight, entropy_weight):
    """compute_train_loss, ppo.py:102-169 ("clip" and "" surrogates): returns (policy_loss, value_loss, entropy_loss).
    new_logpi/old_logpi [B][K]; advantage/v/v_target/old_v [B] (the reference carries them as [B][1])."""
    f = np.float32
    lp, olp = np.asarray(new_logpi, f), np.asarray(old_logpi, f)
    v, vt = np.asarray(v, f)[:, None], np.asarray(v_target, f)[:, None]
    adv = np.asarray(advantage, f)[:, None]
    if baseline_advantage:
        adv = adv - v  # :121-122
    ratio = np.exp(lp - olp)  # :126
    if surrogate_clip:
        rc = np.clip(ratio, f(1 - policy_clip_range), f(1 + policy_clip_range))
        policy = np.minimum(ratio * adv, rc * adv)  # :128-137
    else:
        policy = ratio * adv
    policy_loss = -np.mean(policy, dtype=np.float64)  # :152
    if enable_value_clip:
        ov = np.asarray(old_v, f)[:, None]
        vc = np.clip(v, ov - f(value_clip_range), ov + f(value_clip_range))
        value = np.maximum((v - vt) ** 2, (vc - vt) ** 2)  # :155-157
    else:
        value = (v - vt) ** 2
    value_loss = value_loss_weight * np.mean(value, dtype=np.float64)  # :161
    entropy = np.sum(-np.exp(lp) * lp, axis=-1)  # :166
    entropy_loss = entropy_weight * -np.mean(entropy, dtype=np.float64)  # :167
    return f(policy_loss), f(value_loss), f(entropy_loss)


def pendulum_step(state, t_in_ep, action, episode_len):
    """The Pendulum-shaped synthetic workload of BASELINE config 5 (definition; mirrored by csrc/srlx_ppo.hip),
    without the reset draw: returns (new_state, new_t, obs, reward, done) where done envs keep their stepped state."""
    f = np.float32
    th, thd = np.asarray(state, f)[:, 0].copy(), np.asarray(state, f)[:, 1].copy()
    u = np.clip(np.asarray(action, f), f(-2), f(2))
    an = np.mod(th + f(np.pi), f(2 * np.pi)) - f(np.pi)
    reward = -(an * an + f(0.1) * thd * thd + f(0.001) * u * u)
    thd = np.clip(thd + (f(15.0) * np.sin(th) + f(3.0) * u) * f(0.05), f(-8), f(8)).astype(f)
    th = (th + thd * f(0.05)).astype(f)
    t = np.asarray(t_in_ep) + 1
    done = t >= episode_len
    obs = np.stack([np.cos(th), np.sin(th), thd], axis=1).astype(f)
    return np.stack([th, thd], axis=1), t, obs, reward.astype(f), done


# ------------------------------------------------------------------------------------------
# Agent57 (LSTM, sequence replay): srl/algorithms/agent57/agent57.py:301-379 (calc_target_q) and
# srl/algorithms/agent57/model_torch.py:469-492.  PINNED by tests/golden/agent57_target_*.npz, train_step_agent57.npz.
# ------------------------------------------------------------------------------------------
def agent57_seq_target(q, q_target, actions, rewards, dones, invalid, discounts, retrace_h, double_dqn, rescale):
    """q / q_target [B][S+1][A]; actions int [B][S]; rewards / dones [B][S]; invalid bool [B][S][A] or None;
    discounts [B].  Returns target float32 [S][B]."""
    f = np.float32
    q, q_target = np.array(q, f, copy=True), np.array(q_target, f, copy=True)
    B, S = np.asarray(actions).shape
    action_q = np.take_along_axis(q[:, :-1, :], np.asarray(actions)[..., None].astype(np.int64), axis=2)[..., 0].T  # before the in-place mask below
    n_q, n_qt = q[:, 1:, :], q_target[:, 1:, :]
    mask = np.zeros(n_q.shape, bool) if invalid is None else np.asarray(invalid, bool)
    if double_dqn:  # :318-320
        n_q[mask] = -np.inf
        greedy = np.argmax(n_q, axis=2)
    else:
        n_qt[mask] = -np.inf
        greedy = np.argmax(n_qt, axis=2)
    maxq = np.take_along_axis(n_qt, greedy[..., None], axis=2)[..., 0]
    if rescale:
        maxq = inverse_rescaling(maxq)
    disc = np.asarray(discounts, f)
    gains = np.asarray(rewards, f) + np.asarray(dones, f) * disc[:, None] * maxq  # :334
    if rescale:
        gains = rescaling(gains)
    gains = gains.T  # [S][B]
    on_policy = (np.asarray(actions) == greedy).T  # :351 the taken action equals the greedy one
    coef = np.ones((S, B), f)  # retrace_seq[t] * discounts_seq[t]
    retrace = np.ones(B, f)
    dseq = disc.copy()
    for t in range(S):
        coef[t] = retrace * dseq
        if t + 1 < S:
            retrace = (retrace.astype(np.float64) * (retrace_h * on_policy[t])).astype(f)  # float32 array *= float64 array, :361
            dseq = dseq * disc
    target = np.zeros((S, B), f)
    next_td = np.zeros(B, f)
    for t in reversed(range(S)):  # :369-376
        target[t] = gains[t] + coef[t] * next_td
        next_td = target[t] - action_q[t]
    return target


def agent57_seq_loss(q, target, actions, weights):
    """model_torch.py:471-492: Huber(target*w, q[a]*w) averaged over [S][B]; d loss / d q; mean TD error per row."""
    f = np.float32
    q = np.asarray(q, f)
    B, S = np.asarray(actions).shape
    aq = np.take_along_axis(q[:, :-1, :], np.asarray(actions)[..., None].astype(np.int64), axis=2)[..., 0].T  # [S][B]
    w = np.asarray(weights, f)[None, :]
    d = aq * w - target * w
    ad = np.abs(d)
    loss = f(np.mean(np.where(ad < 1, 0.5 * d * d, ad - 0.5), dtype=np.float64))
    g = np.where(ad < 1, d, np.sign(d)) * w / f(S * B)
    grad = np.zeros_like(q)
    np.put_along_axis(grad[:, :-1, :], np.asarray(actions)[..., None].astype(np.int64), g.T[..., None], axis=2)
    return loss, grad, np.mean(aq - target, axis=0)


# ------------------------------------------------------------------------------------------
# rank-based memory, srl/rl/memories/priority_memories/rankbased_memory.py:42-58 (PINNED by rankbased_trace.npz)
# ------------------------------------------------------------------------------------------
def rankbased_sample(priorities_live, batch_size, alpha, beta):
    """Draws with numpy's GLOBAL generator exactly like the reference: returns (buffer indices, weights)."""
    n = len(priorities_live)
    order = np.argsort(-np.asarray(priorities_live, np.float32))  # :47
    probs = (1 / np.arange(1, n + 1)) ** alpha
    probs /= probs.sum()
    ranks = np.random.choice(n, size=batch_size, p=probs, replace=False)  # == np.random.choice(order, ..., p=probs): same stream
    w = (n * probs[ranks]) ** (-beta)
    return order[ranks], w / w.max()
