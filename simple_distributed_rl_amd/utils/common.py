"""srl/utils/common.py subset: set_seed (:13-44), save_file/load_file (:117-152), load_module (:155-163)."""
import importlib
import lzma
import pickle
import random

import numpy as np


def set_seed(seed, enable_gpu: bool = False):
    if seed is None:
        return
    random.seed(seed)
    np.random.seed(seed)
    try:
        import torch

        torch.manual_seed(seed)
        if enable_gpu and torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
    except ImportError:
        pass


def load_module(entry_point: str):
    if "<locals>" in entry_point:
        raise ValueError(f"entry_point of a local class cannot be imported: {entry_point}")
    if ":" not in entry_point:
        raise ValueError(f"entry_point must be 'module.path:ClassName', got '{entry_point}'")
    mod_name, cls_name = entry_point.split(":")
    return getattr(importlib.import_module(mod_name), cls_name)


def save_file(path: str, dat, compress: bool = True):
    """Same on-disk format as the reference (srl/utils/common.py:117-134): an lzma container of a pickle
    when `compress`, a plain pickle otherwise -- files are interchangeable both ways."""
    import os

    try:
        if compress:
            with lzma.open(path, "w") as f:
                f.write(pickle.dumps(dat))
        else:
            with open(path, "wb") as f:
                pickle.dump(dat, f)
    except Exception:
        if os.path.isfile(path):
            os.remove(path)
        raise


def load_file(path: str):
    """common.py:137-152: sniff the xz magic, else plain pickle."""
    with open(path, "rb") as f:
        is_xz = f.read(6) == bytes.fromhex("fd377a585a00")
    if is_xz:
        with lzma.open(path) as f:
            return pickle.loads(f.read())
    with open(path, "rb") as f:
        return pickle.load(f)
