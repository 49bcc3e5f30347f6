"""Drop-in for StatisticalModel/util.py.

gaussian_function (A1, util.py:20-36, log branch) runs on the GPU through the scoring kernel (a
1-mixture GMM with weight 1).  log_sum_exp / matrix_log_sum_exp (A2/A3, util.py:54-92) are kept as
host helpers with the reference's semantics (quirk Q4) because the drop-in classes merge the tiny
per-unit accumulators ((S-2,S) matrices) with them; the hot loops that call them in the reference
(forward, backward, xi) are fused into the HIP kernels and never come through here.
"""
import numpy as np


def gaussian_function(y, mean, cov, dimension, log=False, standard=False):
    if not log or standard:
        raise NotImplementedError('only log=True, standard=False is on the hot path (SURVEY quirk Q3)')
    from ..runtime import scratch_engine
    from .._lib import PCL_F64
    eng = scratch_engine()
    y = np.asarray(y, dtype=np.float64).reshape(1, -1)
    diag = np.asarray(cov, dtype=np.float64)
    diag = diag.diagonal() if diag.ndim == 2 else diag            # util.py:23
    eng.load_model(np.asarray(mean, np.float64).reshape(1, 1, -1), diag.reshape(1, 1, -1), np.ones((1, 1)))
    eng.load_frames(y)
    b = eng.batch([3], [1], [0])
    b.set_states([np.array([-1, 0, -2], dtype=np.int32)])
    b.score(PCL_F64)
    out = float(b.get('B')[0][1, 0])
    b.close()
    return out


def log_sum_exp(p_list, vector=False):
    """util.py:54-77: max-shifted; returns the max itself when |max| is inf; vector=True reduces
    each first-axis slice entirely."""
    def one(v):
        v = np.asarray(v, dtype=np.float64)
        top = np.max(v)
        if np.isinf(top):
            return top
        with np.errstate(all='ignore'):
            return top + np.log(np.sum(np.exp(v - top)))
    if vector:
        out = [one(row) for row in p_list]
        return np.array(out) if isinstance(p_list, np.ndarray) else out
    return one(p_list)


def matrix_log_sum_exp(array_list, axis_x):
    """util.py:80-92: elementwise LSE over a list of equal-shape matrices, first axis_x rows."""
    stack = np.stack([np.asarray(a, dtype=np.float64)[:axis_x] for a in array_list], axis=0)
    with np.errstate(all='ignore'):
        top = stack.max(axis=0)
        safe = np.where(np.isinf(top), 0.0, top)
        out = safe + np.log(np.exp(stack - safe).sum(axis=0))
    return np.where(np.isinf(top), top, out)


_NPY_HEADERS = {}
_NEXT_NAME = {}            # names taken within the current second by this process: {stamp: {prefix: next k}}


def save_acc_file(directory, name, stamp, val):
    """Write one accumulator as <directory>/<name>_<stamp>[kkk].npy without overwriting a file of the same second (the
    reference does overwrite).  Same bytes as np.save; the header is cached per (shape, dtype) and the
    name is claimed with an exclusive create, because a worker flush writes several hundred of these small files and
    np.save's per-file stat/format work was most of the flush."""
    import io
    import os
    val = np.asarray(val, order='C')           # (np.ascontiguousarray would turn the 0-d alpha accumulator into shape (1,))
    if val.dtype.hasobject:
        raise TypeError('accumulators are plain numeric arrays')
    key = (val.shape, val.dtype.str)
    head = _NPY_HEADERS.get(key)
    if head is None:
        buf = io.BytesIO()
        np.lib.format.write_array_header_1_0(buf, {'descr': val.dtype.str, 'fortran_order': False, 'shape': val.shape})
        head = _NPY_HEADERS[key] = buf.getvalue()
    if stamp not in _NEXT_NAME:
        _NEXT_NAME.clear()
        _NEXT_NAME[stamp] = {}
    taken = _NEXT_NAME[stamp]
    prefix = '%s/%s_%d' % (directory, name, stamp)
    k = taken.get(prefix, 0)
    while True:
        f = prefix + ('%03d.npy' % k if k else '.npy')
        try:
            fd = open(f, 'xb')
            break
        except FileExistsError:                 # (another process, or an earlier run, within the same second)
            k += 1
        except FileNotFoundError:
            os.makedirs(directory, exist_ok=True)
    taken[prefix] = k + 1
    with fd:
        fd.write(head)
        fd.write(val.data if val.ndim else val.tobytes())
    return f
