"""ctypes bindings to oracle/_build/liboracle.so (the CPU restatement; test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
_LIB = None

c_i64 = ctypes.c_int64
c_f64 = ctypes.c_double
c_p = ctypes.c_void_p


def _ptr(a):
    return a.ctypes.data_as(c_p) if a is not None else None


def load_oracle():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "per_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/liboracle.so"])
    lib = ctypes.CDLL(path)
    lib.per_oracle_create.restype = c_p
    lib.per_oracle_create.argtypes = [c_i64, c_f64, c_f64, c_f64, ctypes.c_int, c_f64]
    lib.per_oracle_destroy.argtypes = [c_p]
    lib.per_oracle_clear.argtypes = [c_p]
    lib.per_oracle_length.restype = c_i64
    lib.per_oracle_length.argtypes = [c_p]
    lib.per_oracle_total.restype = c_f64
    lib.per_oracle_total.argtypes = [c_p]
    lib.per_oracle_max_priority.restype = c_f64
    lib.per_oracle_max_priority.argtypes = [c_p]
    lib.per_oracle_write.restype = c_i64
    lib.per_oracle_write.argtypes = [c_p]
    lib.per_oracle_tree.restype = ctypes.POINTER(c_f64)
    lib.per_oracle_tree.argtypes = [c_p]
    lib.per_oracle_tree_len.restype = c_i64
    lib.per_oracle_tree_len.argtypes = [c_p]
    lib.per_oracle_add.argtypes = [c_p, c_f64, ctypes.c_int]
    lib.per_oracle_sample.restype = c_i64
    lib.per_oracle_sample.argtypes = [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_p, c_p]
    lib.per_oracle_update_f32.argtypes = [c_p, c_i64, c_p, c_p]
    lib.per_oracle_update_f64.argtypes = [c_p, c_i64, c_p, c_p]
    lib.per_oracle_update_raw.argtypes = [c_p, c_i64, c_p, c_p]
    lib.per_oracle_get_state.argtypes = [c_p, c_p, c_p, c_p, c_p]
    lib.per_oracle_set_state.argtypes = [c_p, c_f64, c_i64, c_i64, c_p]
    lib.per_oracle_restore_resized.argtypes = [c_p, c_i64, c_i64, c_p]
    _LIB = lib
    return lib


ADD_NONE, ADD_PYFLOAT, ADD_RAW = 0, 1, 2


class OraclePER:
    """Thin OO wrapper over the C restatement of ProportionalMemory."""

    def __init__(self, capacity, alpha=0.6, beta_initial=0.4, beta_steps=1_000_000, has_duplicate=True, epsilon=1e-4):
        self.lib = load_oracle()
        self.capacity = int(capacity)
        self.h = self.lib.per_oracle_create(
            self.capacity, float(alpha), float(beta_initial), float(beta_steps), int(bool(has_duplicate)), float(epsilon)
        )
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.per_oracle_destroy(self.h)
            self.h = None

    def clear(self):
        self.lib.per_oracle_clear(self.h)

    def length(self):
        return int(self.lib.per_oracle_length(self.h))

    def total(self):
        return float(self.lib.per_oracle_total(self.h))

    @property
    def max_priority(self):
        return float(self.lib.per_oracle_max_priority(self.h))

    @property
    def write(self):
        return int(self.lib.per_oracle_write(self.h))

    def tree(self):
        n = int(self.lib.per_oracle_tree_len(self.h))
        return np.ctypeslib.as_array(self.lib.per_oracle_tree(self.h), shape=(n,)).copy()

    def add(self, priority=None, mode=None):
        if mode is None:
            if priority is None:
                mode = ADD_NONE
            else:
                mode = ADD_PYFLOAT
        self.lib.per_oracle_add(self.h, 0.0 if priority is None else float(priority), int(mode))

    def sample(self, batch_size, step, uniforms):
        u = np.ascontiguousarray(uniforms, dtype=np.float64)
        idx = np.empty(batch_size, np.int64)
        w = np.empty(batch_size, np.float64)
        p = np.empty(batch_size, np.float64)
        used = self.lib.per_oracle_sample(self.h, int(batch_size), int(step), _ptr(u), int(u.size), _ptr(idx), _ptr(w), _ptr(p))
        return int(used), idx, w, p

    def update(self, indices, priorities, raw=False):
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        pr = np.asarray(priorities)
        if raw:
            pr = np.ascontiguousarray(pr, np.float64)
            self.lib.per_oracle_update_raw(self.h, idx.size, _ptr(idx), _ptr(pr))
        elif pr.dtype == np.float32:
            pr = np.ascontiguousarray(pr)
            self.lib.per_oracle_update_f32(self.h, idx.size, _ptr(idx), _ptr(pr))
        else:
            pr = np.ascontiguousarray(pr, np.float64)
            self.lib.per_oracle_update_f64(self.h, idx.size, _ptr(idx), _ptr(pr))

    def get_state(self):
        mp = c_f64()
        size = c_i64()
        write = c_i64()
        tree = np.empty(2 * self.capacity - 1, np.float64)
        self.lib.per_oracle_get_state(self.h, ctypes.byref(mp), ctypes.byref(size), ctypes.byref(write), _ptr(tree))
        return mp.value, size.value, write.value, tree

    def set_state(self, max_priority, size, write, tree):
        tree = np.ascontiguousarray(tree, np.float64)
        assert tree.size == 2 * self.capacity - 1
        self.lib.per_oracle_set_state(self.h, float(max_priority), int(size), int(write), _ptr(tree))

    def restore_resized(self, old_capacity, old_size, old_tree):
        old_tree = np.ascontiguousarray(old_tree, np.float64)
        self.lib.per_oracle_restore_resized(self.h, int(old_capacity), int(old_size), _ptr(old_tree))


# -- golden trace replay -------------------------------------------------------------
OP_ADD_NONE, OP_ADD_PY, OP_SAMPLE, OP_UPDATE_F32, OP_UPDATE_F64 = 0, 1, 2, 3, 4


def iter_trace(z):
    """Yields (kind, payload) from a tests/golden/per_trace_*.npz file."""
    code, a, b = z["op_code"], z["op_a"], z["op_b"]
    nu, ou, oi, op, ow = z["op_n_uniforms"], z["op_off_u"], z["op_off_i"], z["op_off_p"], z["op_off_w"]
    for k in range(code.size):
        c = int(code[k])
        if c == OP_ADD_NONE:
            yield "add", dict(priority=None)
        elif c == OP_ADD_PY:
            yield "add", dict(priority=float(a[k]))
        elif c == OP_SAMPLE:
            B = int(b[k])
            yield "sample", dict(
                batch_size=B,
                step=int(a[k]),
                uniforms=z["pool_u"][ou[k] : ou[k] + nu[k]],
                indices=z["pool_idx"][oi[k] : oi[k] + B],
                weights=z["pool_w"][ow[k] : ow[k] + B],
            )
        else:
            n = int(b[k])
            pri = z["pool_pri"][op[k] : op[k] + n]
            yield "update", dict(
                indices=z["pool_idx"][oi[k] : oi[k] + n],
                priorities=pri.astype(np.float32) if c == OP_UPDATE_F32 else pri,
                transformed=z["pool_tx"][op[k] : op[k] + n],
            )
