/*
 * oracle/per_oracle.c -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * CPU restatement, in plain C, of the reference's proportional prioritized
 * replay memory (sum-tree).  It follows the *Python* class operation by
 * operation (the Python class is the parity oracle; its pybind11 C++ twin is
 * not seedable and uses different float widths):
 *
 *   reference file: srl/rl/memories/priority_memories/proportional_memory.py
 *     SumTree.__init__        :43-47   -> per_oracle_create / per_oracle_clear
 *     SumTree._propagate      :49-54   -> propagate()
 *     SumTree._retrieve       :56-66   -> retrieve()
 *     SumTree.add             :71-79   -> tree_add()
 *     SumTree.update          :81-86   -> tree_update()
 *     SumTree.get             :88-92   -> (inlined in per_oracle_sample)
 *     ProportionalMemory.add  :120-129 -> per_oracle_add*
 *     ProportionalMemory.sample :131-169 -> per_oracle_sample
 *     ProportionalMemory.update :171-177 -> per_oracle_update_f32 / _f64 / _raw
 *     backup / restore        :179-205 -> per_oracle_get_state / _set_state /
 *                                          per_oracle_restore_resized
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py replays the call traces
 * recorded from the imported reference (oracle/gen_golden.py ->
 * tests/golden/per_trace_*.npz) through this file and requires bit-equal
 * indices, bit-equal tree contents and <=1e-15 relative weights.
 *
 * The only things that may import/link this file are tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Random numbers: the reference draws `random.random()` (MT19937, 53 bit) once
 * per descent attempt (:147).  Here the caller supplies that stream as an
 * array of doubles in [0,1); per_oracle_sample reports how many it consumed.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int64_t capacity;
    double alpha;
    double beta_initial;
    double beta_steps; /* kept as the number the caller passed (int in python) */
    double epsilon;
    int has_duplicate;
    /* state */
    double max_priority;
    int64_t size;
    int64_t write;
    int64_t tree_len; /* 2*capacity-1 */
    double *tree;
} per_oracle_t;

/* proportional_memory.py:49-54 */
static void propagate(double *tree, int64_t idx, double change) {
    while (idx != 0) {
        int64_t parent = (idx - 1) / 2;
        tree[parent] += change;
        idx = parent;
    }
}

/* proportional_memory.py:56-66 */
static int64_t retrieve(const double *tree, int64_t tree_len, double val) {
    int64_t idx = 0;
    for (;;) {
        int64_t left = 2 * idx + 1;
        if (left >= tree_len) return idx;
        if (val <= tree[left]) {
            idx = left;
        } else {
            idx = left + 1;
            val -= tree[left];
        }
    }
}

/* proportional_memory.py:81-86 */
static void tree_update(per_oracle_t *m, int64_t tree_idx, double priority) {
    double change = priority - m->tree[tree_idx];
    m->tree[tree_idx] = priority;
    propagate(m->tree, tree_idx, change);
}

/* proportional_memory.py:71-79 */
static void tree_add(per_oracle_t *m, double priority) {
    int64_t tree_idx = m->write + m->capacity - 1;
    tree_update(m, tree_idx, priority);
    m->write += 1;
    if (m->write >= m->capacity) m->write = 0;
}

/* proportional_memory.py:112-115 */
void per_oracle_clear(per_oracle_t *m) {
    memset(m->tree, 0, sizeof(double) * (size_t)m->tree_len);
    m->write = 0;
    m->max_priority = 1.0;
    m->size = 0;
}

per_oracle_t *per_oracle_create(int64_t capacity, double alpha, double beta_initial, double beta_steps,
                                int has_duplicate, double epsilon) {
    if (capacity <= 0) return NULL;
    per_oracle_t *m = (per_oracle_t *)calloc(1, sizeof(per_oracle_t));
    if (!m) return NULL;
    m->capacity = capacity;
    m->alpha = alpha;
    m->beta_initial = beta_initial;
    m->beta_steps = beta_steps;
    m->epsilon = epsilon;
    m->has_duplicate = has_duplicate;
    m->tree_len = 2 * capacity - 1;
    m->tree = (double *)malloc(sizeof(double) * (size_t)m->tree_len);
    if (!m->tree) {
        free(m);
        return NULL;
    }
    per_oracle_clear(m);
    return m;
}

void per_oracle_destroy(per_oracle_t *m) {
    if (!m) return;
    free(m->tree);
    free(m);
}

int64_t per_oracle_length(const per_oracle_t *m) { return m->size; }
double per_oracle_total(const per_oracle_t *m) { return m->tree[0]; }
double per_oracle_max_priority(const per_oracle_t *m) { return m->max_priority; }
int64_t per_oracle_write(const per_oracle_t *m) { return m->write; }
const double *per_oracle_tree(const per_oracle_t *m) { return m->tree; }
int64_t per_oracle_tree_len(const per_oracle_t *m) { return m->tree_len; }

static void bump_size(per_oracle_t *m) {
    m->size += 1;
    if (m->size > m->capacity) m->size = m->capacity;
}

/* proportional_memory.py:120-129.
 * mode 0: priority is None      -> max_priority              (:121-122)
 * mode 1: python float priority -> (abs(p)+eps)**alpha       (:124), fp64 libm pow
 * mode 2: _restore_skip=True    -> value used as is          (:123)
 * A numpy float32 *scalar* priority (what distributed Rainbow passes,
 * rainbow.py:398) is widened with float() by the caller first: under NEP-50
 * numpy the reference would otherwise leak float32 into the tree's ancestor
 * sums, under numpy 1.x it would not; the C++ twin takes a double
 * (proportional_memory.cpp:124).  See oracle/gen_golden.py trace (4).
 */
void per_oracle_add(per_oracle_t *m, double priority, int mode) {
    double p;
    if (mode == 0) {
        p = m->max_priority;
    } else if (mode == 1) {
        p = pow(fabs(priority) + m->epsilon, m->alpha);
    } else {
        p = priority;
    }
    tree_add(m, p);
    bump_size(m);
}

/* proportional_memory.py:131-169.
 *
 * uniforms[0..n_uniforms): the values random.random() would return, in order.
 * Returns the number of uniforms consumed (>= batch_size), or -1 if the
 * stream ran out before batch_size draws were accepted.
 * out_idx    : tree indices (what the reference returns as update_args)
 * out_weight : importance weights after max normalisation, fp64 like the
 *              reference (`np.empty(batch_size)` is float64, :134)
 * out_prio   : the leaf priority that was read for each accepted draw
 *              (not returned by the reference; exported for tests)
 */
int64_t per_oracle_sample(const per_oracle_t *m, int64_t batch_size, int64_t step, const double *uniforms,
                          int64_t n_uniforms, int64_t *out_idx, double *out_weight, double *out_prio) {
    const double total = m->tree[0]; /* :135 */
    /* :138-140  beta = beta_initial + (1 - beta_initial) * step / beta_steps */
    double beta = m->beta_initial + ((1.0 - m->beta_initial) * (double)step) / m->beta_steps;
    if (beta > 1.0) beta = 1.0;

    int64_t used = 0;
    int64_t idx = 0;
    double priority = 0.0;
    for (int64_t i = 0; i < batch_size; i++) {
        for (int tries = 0; tries < 9999; tries++) { /* :146 */
            if (used >= n_uniforms) return -1;
            double r = uniforms[used++] * total; /* :147 */
            idx = retrieve(m->tree, m->tree_len, r); /* :148 */
            priority = m->tree[idx];
            if (priority == 0.0) continue; /* :150-152 */
            if (!m->has_duplicate) {       /* :155-156 */
                int dup = 0;
                for (int64_t k = 0; k < i; k++)
                    if (out_idx[k] == idx) {
                        dup = 1;
                        break;
                    }
                if (dup) continue;
            }
            break;
        }
        out_idx[i] = idx;
        if (out_prio) out_prio[i] = priority;
        double prob = priority / total;                        /* :163 */
        out_weight[i] = pow((double)m->size * prob, -beta);    /* :164 */
    }
    /* :167 weights = weights / weights.max() */
    double wmax = out_weight[0];
    for (int64_t i = 1; i < batch_size; i++)
        if (out_weight[i] > wmax) wmax = out_weight[i];
    for (int64_t i = 0; i < batch_size; i++) out_weight[i] = out_weight[i] / wmax;
    return used;
}

static void apply_update(per_oracle_t *m, int64_t tree_idx, double priority) {
    tree_update(m, tree_idx, priority); /* :175 */
    if (m->max_priority < priority) m->max_priority = priority; /* :176-177 */
}

/* proportional_memory.py:171-177 with a float32 `priorities` array: numpy
 * evaluates (np.abs(p) + eps) ** alpha in float32 (:172) and the loop widens
 * each element with float() (:174).  float32 `power` is the correctly rounded
 * value here; numpy's AVX512-SVML loop may differ from it by one float32 ulp
 * for alpha not in {0, 0.5, 1} (measured in this container) - alpha = 0.5, the
 * BASELINE configuration, takes numpy's sqrt fast path and is exact. */
void per_oracle_update_f32(per_oracle_t *m, int64_t n, const int64_t *indices, const float *priorities) {
    float a = (float)m->alpha;
    float e = (float)m->epsilon;
    for (int64_t i = 0; i < n; i++) {
        float x = fabsf(priorities[i]) + e;
        float p = (a == 0.5f) ? sqrtf(x) : (float)pow((double)x, (double)a);
        apply_update(m, indices[i], (double)p);
    }
}

/* same with a float64 / int / python-list `priorities` (numpy promotes to
 * float64; test_priority_memories.py:71, speedtest.py:54) */
void per_oracle_update_f64(per_oracle_t *m, int64_t n, const int64_t *indices, const double *priorities) {
    for (int64_t i = 0; i < n; i++) {
        double x = fabs(priorities[i]) + m->epsilon;
        double p = (m->alpha == 0.5) ? sqrt(x) : pow(x, m->alpha);
        apply_update(m, indices[i], p);
    }
}

/* priorities already transformed by the caller (used to replay the reference's
 * own numpy transform bit for bit) */
void per_oracle_update_raw(per_oracle_t *m, int64_t n, const int64_t *indices, const double *priorities) {
    for (int64_t i = 0; i < n; i++) apply_update(m, indices[i], priorities[i]);
}

/* backup()/restore() same-capacity path, proportional_memory.py:179-194 */
void per_oracle_get_state(const per_oracle_t *m, double *max_priority, int64_t *size, int64_t *write,
                          double *tree_out) {
    *max_priority = m->max_priority;
    *size = m->size;
    *write = m->write;
    if (tree_out) memcpy(tree_out, m->tree, sizeof(double) * (size_t)m->tree_len);
}

void per_oracle_set_state(per_oracle_t *m, double max_priority, int64_t size, int64_t write, const double *tree_in) {
    m->max_priority = max_priority;
    m->size = size;
    m->write = write;
    memcpy(m->tree, tree_in, sizeof(double) * (size_t)m->tree_len);
}

/* restore() different-capacity path, proportional_memory.py:195-205: clear,
 * then re-add the first `size` leaves of the old tree with _restore_skip. */
void per_oracle_restore_resized(per_oracle_t *m, int64_t old_capacity, int64_t old_size, const double *old_tree) {
    per_oracle_clear(m);
    for (int64_t i = 0; i < old_size; i++) per_oracle_add(m, old_tree[i + old_capacity - 1], 2);
}
