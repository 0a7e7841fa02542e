"""ORACLE — test infrastructure only (never imported by the product package).

Python loaders for the two CPU checkers of the alignment DP:

* ``mas_port``      — ``oracle/mas_ref.c`` (plain-C restatement of
  reference monotonic_align/core.pyx:5-42) through ctypes.
* ``mas_reference`` — ``oracle/_ref/core*.so``: the reference's own Cython routine,
  compiled by ``make -C oracle ref`` from /root/reference where it lies.  Present only
  if that build ran in the build container; ``have_reference()`` says so.

Both take ``neg_cent[b,t_t,t_s]`` float32 and int32 lengths and return the int32 0/1
path, mirroring reference monotonic_align/__init__.py:6-19 (copy in, zeroed path out).
"""
import ctypes
import glob
import importlib.util
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_ref = None


def build():
    """Compile the C restatement (and the reference module when /root/reference exists)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)
    if os.path.isdir("/root/reference/monotonic_align") and not glob.glob(os.path.join(_HERE, "_ref", "core*.so")):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)


def _load_port():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libmas_oracle.so")
        if not os.path.exists(path):
            build()
        _lib = ctypes.CDLL(path)
        _lib.oracle_mas_f32.restype = ctypes.c_int
        _lib.oracle_mas_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3
    return _lib


def have_reference():
    return bool(glob.glob(os.path.join(_HERE, "_ref", "core*.so")))


def _load_reference():
    global _ref
    if _ref is None:
        so = glob.glob(os.path.join(_HERE, "_ref", "core*.so"))
        if not so:
            raise FileNotFoundError("oracle/_ref/core*.so missing: run `make -C oracle ref` in the build container")
        spec = importlib.util.spec_from_file_location("core", so[0])
        _ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_ref)
    return _ref


def _prep(neg_cent, t_ys, t_xs):
    values = np.ascontiguousarray(np.array(neg_cent, dtype=np.float32, copy=True))
    assert values.ndim == 3
    t_ys = np.ascontiguousarray(np.asarray(t_ys, dtype=np.int32))
    t_xs = np.ascontiguousarray(np.asarray(t_xs, dtype=np.int32))
    path = np.zeros(values.shape, dtype=np.int32)
    return path, values, t_ys, t_xs


def mas_port(neg_cent, t_ys, t_xs):
    lib = _load_port()
    path, values, t_ys, t_xs = _prep(neg_cent, t_ys, t_xs)
    b, t_t, t_s = values.shape
    rc = lib.oracle_mas_f32(path.ctypes.data, values.ctypes.data, t_ys.ctypes.data, t_xs.ctypes.data, b, t_t, t_s)
    if rc != 0:
        raise ValueError(f"{-rc} item(s) outside the domain 1 <= t_x <= t_y")
    return path


def mas_reference(neg_cent, t_ys, t_xs):
    mod = _load_reference()
    path, values, t_ys, t_xs = _prep(neg_cent, t_ys, t_xs)
    mod.maximum_path_c(path, values, t_ys, t_xs)
    return path


def lengths_from_mask(mask):
    """reference monotonic_align/__init__.py:16-17: t_t = mask.sum(1)[:,0], t_s = mask.sum(2)[:,0]."""
    mask = np.asarray(mask)
    return mask.sum(1)[:, 0].astype(np.int32), mask.sum(2)[:, 0].astype(np.int32)
