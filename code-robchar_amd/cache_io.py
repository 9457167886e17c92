"""Writers / readers of the cached-results layout (`.mc` fidelity caches, `.mcm` metric caches).

Reference layout (mcsim.py:457-459, :501): ``json.dump({algo: [L][C][K]})`` and ``json.dump({algo: {metric: [L][C]}})``.
Two things differ here, both on the WRITING side only - the files stay loadable by the reference's own `json.load`
cache-hit branches (mcsim.py:396-397, :504-506):

* tensors are formatted by the native encoder `rc_json_encode_f64` (include/robchar_hip.h) straight from NumPy
  memory - shortest round-trip digits, all host threads - instead of `tolist()` + the Python encoder (the reference
  spends ~0.5 s per 1e5 values there, which would dominate the whole path once the arithmetic is on the GPU);
* the whole-dict re-dump after every algorithm (mcsim.py:457-459) is an APPEND: the file is a valid JSON object after
  every algorithm, but earlier algorithms' text is not re-encoded or re-written.

Large tensors (more values than `json_max_values`) go to a `.npy` sidecar per algorithm, and the `.mc` file becomes a
small JSON index ``{"__robchar_npy__": 1, algo: {"npy": <file name>, "shape": [L, C, K]}}`` - NOT readable by the
reference (a 1000 x 1e5 level is 2 GB of JSON there); `load_mc` maps the sidecars back (memory-mapped).
"""
from __future__ import annotations

import ctypes
import json
import os
from typing import Dict

import numpy as np

from . import _lib

NPY_MARKER = "__robchar_npy__"


def encode_array(arr: np.ndarray, nthreads: int = 0) -> memoryview:
    """JSON text (bytes) of a float64 array as nested lists, via the native encoder."""
    a = np.ascontiguousarray(arr, dtype=np.float64)
    if a.ndim == 0:
        a = a.reshape(1)
    lib = _lib.load()
    shape = (ctypes.c_longlong * a.ndim)(*a.shape)
    cap = lib.rc_json_bound_f64(a.ndim, shape)
    if cap < 0:
        raise ValueError("rc_json_bound_f64: bad shape")
    buf = np.empty(int(cap), dtype=np.uint8)
    n = lib.rc_json_encode_f64(ctypes.c_void_p(a.ctypes.data), a.ndim, shape, ctypes.c_void_p(buf.ctypes.data),
                               int(cap), int(nthreads))
    if n < 0:
        raise ValueError("rc_json_encode_f64 failed")
    return memoryview(buf)[: int(n)]


def _write_array(fh, arr: np.ndarray, nthreads: int = 0) -> None:
    """Stream the JSON text of `arr` to the binary file object `fh` (native encoder writing to its descriptor)."""
    a = np.ascontiguousarray(arr, dtype=np.float64)
    if a.ndim == 0:
        a = a.reshape(1)
    lib = _lib.load()
    fh.flush()
    shape = (ctypes.c_longlong * a.ndim)(*a.shape)
    n = lib.rc_json_write_f64(fh.fileno(), ctypes.c_void_p(a.ctypes.data), a.ndim, shape, int(nthreads))
    if n < 0:
        raise OSError("rc_json_write_f64 failed")
    fh.seek(0, os.SEEK_END)              # the descriptor moved underneath the buffered object


_MID_MIN, _MID_MAX = 4096, 1 << 20     # leaves of this size are encoded side by side, one thread each (see write_json)
_pool = None


def _encode_pool():
    global _pool
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1), thread_name_prefix="rc-json")
    return _pool


def warm_up() -> None:
    """Start the encoder threads now (a process's first `.mcm` write otherwise pays ~10 ms for creating them)."""
    import time
    pool = _encode_pool()
    list(pool.map(time.sleep, [0.002] * pool._max_workers))


def _mid_leaves(value, acc):
    if isinstance(value, np.ndarray):
        if _MID_MIN <= value.size < _MID_MAX:
            acc.append(value)
    elif isinstance(value, dict):
        for v in value.values():
            _mid_leaves(v, acc)
    return acc


def _write_value(fh, value, pre=None) -> None:
    """One JSON value: arrays through the native encoder, dicts recursively, everything else via json.dumps.
    `pre`: {id(array): encoded text} of leaves that were encoded ahead (write_json)."""
    if isinstance(value, np.ndarray):
        if pre is not None and id(value) in pre:
            fh.write(pre[id(value)])
        elif value.size < _MID_MIN:
            fh.write(encode_array(value))
        else:
            _write_array(fh, value)
    elif isinstance(value, dict):
        fh.write(b"{")
        for i, (k, v) in enumerate(value.items()):
            if i:
                fh.write(b", ")
            fh.write(json.dumps(str(k)).encode() + b": ")
            _write_value(fh, v, pre)
        fh.write(b"}")
    else:
        fh.write(json.dumps(value).encode())


def write_json(obj: dict, path: str) -> None:
    """`json.dump(obj, open(path, "w"))` for a (nested) dict whose leaves may be NumPy arrays.
    A `.mcm` file is many mid-size leaves ({algo: {metric: [L][C]}}: 60 arrays of 11 000 values at the paper's scale),
    each too small for the encoder's own threads to pay: those are encoded side by side, one thread per leaf (ctypes
    releases the GIL), and written in order - 11 ms -> 2 ms of a 24 ms `get_metrics_dict()` call."""
    leaves = _mid_leaves(obj, [])
    pre = None
    if len(leaves) > 1:
        texts = list(_encode_pool().map(lambda a: encode_array(a, 1), leaves))
        pre = {id(a): t for a, t in zip(leaves, texts)}
    with open(path, "wb") as fh:
        _write_value(fh, obj, pre)


class McWriter:
    """Incremental writer of one `.mc` file: `dump(simdict)` after every algorithm, like mcsim.py:457-459, appends the
    algorithms that are not in the file yet (the file is a complete JSON object after every call).

    Resuming: a writer opened on a file that `load_mc` has just read (`McWriter.resume`) adopts the loaded algorithms as
    already written, so an interrupted run's cache only GROWS - nothing that is on disk is re-encoded, and an `.npy`
    sidecar is never rewritten from the memory map that `load_mc` returned for it.  Sidecars and fresh index files are
    written to a temporary name and moved into place (`os.replace`)."""

    def __init__(self, path: str, json_max_values: int, cache_format: str = "auto"):
        self.path = path
        self.json_max_values = int(json_max_values)
        self.cache_format = cache_format
        self.written = {}            # algo -> the value object that was written (held, so identity is meaningful)
        self.mode = None             # "json" | "npy" once decided
        self._keep_mode = False      # resumed file: stay in the format the file has

    @classmethod
    def resume(cls, path: str, loaded: Dict[str, object], json_max_values: int, cache_format: str = "auto") -> "McWriter":
        """Writer over an EXISTING `.mc` file whose content is `loaded` (what `load_mc(path)` returned)."""
        w = cls(path, json_max_values, cache_format)
        with open(path, "rb") as fh:
            head = fh.read(64)
            fh.seek(0, os.SEEK_END)
            size = fh.tell()
            fh.seek(max(0, size - 64))
            tail = fh.read()
        is_npy = json.dumps(NPY_MARKER).encode() in head
        mode = "npy" if is_npy else "json"
        if cache_format in ("json", "npy") and cache_format != mode:
            return w                                     # a different format was asked for: first dump rewrites the file
        end = tail.rstrip()
        if not end.endswith(b"}"):
            return w                                     # not a complete JSON object (interrupted append): rewrite
        if len(tail) != len(end):                        # trailing whitespace / newline: the append seeks over "}"
            with open(path, "r+b") as fh:
                fh.truncate(size - (len(tail) - len(end)))
        w.mode, w._keep_mode = mode, True
        w.written = dict(loaded)
        return w

    def _choose(self, simdict) -> str:
        if self._keep_mode and self.mode is not None:
            return self.mode
        if self.cache_format in ("json", "npy"):
            return self.cache_format
        total = sum(_count(v) for v in simdict.values())
        return "json" if total <= self.json_max_values else "npy"

    def _save_sidecar(self, side: str, val) -> np.ndarray:
        """`np.save(side, val)` - except when `val` IS the memory map of `side` (then the data is already there; writing
        would truncate the file under the map), and never in place: temporary file + `os.replace`."""
        if isinstance(val, np.memmap) and getattr(val, "filename", None) and os.path.exists(side) \
                and os.path.samefile(val.filename, side):
            return val
        arr = np.asarray(val, dtype=np.float64)
        tmp = side + ".tmp%d" % os.getpid()
        try:
            with open(tmp, "wb") as fh:
                np.save(fh, arr)
            os.replace(tmp, side)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        return arr

    def dump(self, simdict: Dict[str, object]) -> None:
        mode = self._choose(simdict)
        fresh = (self.mode != mode or not os.path.exists(self.path)
                 or any(a not in simdict or self.written[a] is not simdict[a] for a in self.written))
        if fresh:
            self.written = {}
            self.mode = mode
            self._keep_mode = False
            tmp = self.path + ".tmp%d" % os.getpid()
            with open(tmp, "wb") as fh:
                fh.write(b"{" + (json.dumps(NPY_MARKER).encode() + b": 1" if mode == "npy" else b"") + b"}")
            os.replace(tmp, self.path)
        todo = [a for a in simdict if a not in self.written]
        if not todo:
            return
        with open(self.path, "r+b") as fh:
            fh.seek(-1, os.SEEK_END)                     # over the closing brace
            empty = fh.tell() == 1                       # the file is "{}": no separator before the first entry
            first = empty
            for algo in todo:
                if not first:
                    fh.write(b", ")
                first = False
                fh.write(json.dumps(algo).encode() + b": ")
                val = simdict[algo]
                if mode == "npy":
                    side = self.path + "." + algo + ".npy"
                    arr = self._save_sidecar(side, val)
                    fh.write(json.dumps({"npy": os.path.basename(side), "shape": list(arr.shape)}).encode())
                else:
                    _write_value(fh, val if isinstance(val, (np.ndarray, dict)) else np.asarray(val, dtype=np.float64))
                self.written[algo] = val
            fh.write(b"}")


def _count(v) -> int:
    """Number of values of a tensor-like WITHOUT materialising it (a device-resident handle has a `shape`)."""
    if hasattr(v, "shape"):
        return int(np.prod(v.shape))
    return int(np.asarray(v).size)


def load_mc(path: str):
    """`json.load(open(path, "rb"))` (mcsim.py:366-367); an npy-index file is resolved to memory-mapped arrays."""
    with open(path, "rb") as fh:
        data = json.load(fh)
    if isinstance(data, dict) and data.get(NPY_MARKER):
        out = {}
        for algo, ref in data.items():
            if algo == NPY_MARKER:
                continue
            out[algo] = np.load(os.path.join(os.path.dirname(path), ref["npy"]), mmap_mode="r")
        return out
    return data
