"""ctypes binding of libpoccala_hip.so (C-ABI: include/poccala_hip.h).

The HIP library is the product; there is NO CPU fallback.  Importing this module
without the built library, or calling into it without a GPU, raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('POCCALA_HIP_LIB', os.path.join(_HERE, 'libpoccala_hip.so'))   # override: kernel A/B builds

PCL_F32, PCL_F64 = 0, 1
PCL_MODEL_Q1_SUMVAR, PCL_MODEL_LOGDET = 0, 1
PCL_ROW_ENTRY, PCL_ROW_EXIT = -1, -2
PCL_MAX_PASS = 16
GET = dict(B=0, alpha=1, beta=2, lgamma=3, ksai=4, gamma=5, pi=6, logp=7, npass=8, qtrace=9, path=10, point=11, ksai_nz=12)

# every symbol include/poccala_hip.h declares: (restype, argtypes)
_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
PROTOTYPES = {
    'pcl_init': (_i, [_i, C.POINTER(_vp)]),
    'pcl_destroy': (_i, [_vp]),
    'pcl_last_error': (C.c_char_p, [_vp]),
    'pcl_sync': (_i, [_vp]),
    'pcl_device_info': (_i, [_vp, C.c_char_p, _i, C.POINTER(_i), C.POINTER(C.c_size_t)]),
    'pcl_kernel_time': (_i, [_vp, C.c_char_p, C.POINTER(C.c_float), C.POINTER(_i)]),
    'pcl_model_upload': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i]),
    'pcl_frames_upload': (_i, [_vp, C.c_int64, _i, _vp, _i]),
    'pcl_frames_stage': (_i, [_vp, C.c_int64, _i, _vp]),
    'pcl_frames_swap': (_i, [_vp]),
    'pcl_host_alloc': (_i, [_vp, C.c_size_t, C.POINTER(C.c_void_p)]),
    'pcl_host_free': (_i, [_vp, _vp]),
    'pcl_batch_create': (_i, [_vp, _i, _vp, _vp, _vp, C.POINTER(_vp)]),
    'pcl_batch_destroy': (_i, [_vp]),
    'pcl_batch_set_transitions': (_i, [_vp, _vp, _vp]),
    'pcl_batch_set_states': (_i, [_vp, _vp]),
    'pcl_batch_set_emissions': (_i, [_vp, _vp]),
    'pcl_batch_set_posteriors': (_i, [_vp, _vp]),
    'pcl_batch_score': (_i, [_vp, _i]),
    'pcl_batch_forward_backward': (_i, [_vp, _i, _d]),
    'pcl_batch_viterbi': (_i, [_vp, _i]),
    'pcl_batch_get': (_i, [_vp, _i, _vp]),
    'pcl_batch_sizes': (_i, [_vp, _vp, _vp, _vp, _vp]),
    'pcl_batch_fetch_async': (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    'pcl_batch_fetch_wait': (_i, [_vp]),
    'pcl_clock_probe': (_i, [_vp, _i, C.POINTER(C.c_double)]),
    'pcl_stats_zero': (_i, [_vp]),
    'pcl_batch_accumulate': (_i, [_vp, _i]),
    'pcl_accumulate_prune': (_i, [_vp, C.c_double]),
    'pcl_stats_download': (_i, [_vp, _vp, _vp, _vp, _vp]),
    'pcl_mstep': (_i, [_vp, _d]),
    'pcl_model_download': (_i, [_vp, _vp, _vp, _vp]),
    'pcl_model_conditioning': (_i, [_vp, _vp, _vp]),
    'pcl_model_split_info': (_i, [_vp, _vp, _vp]),
    'pcl_coarse_counter': (_i, [_vp, _vp, _i]),
    'pcl_coarse_counters': (_i, [_vp, _vp, _vp, _i]),
    'pcl_score_occupancy': (_i, [_vp, _i]),
    'pcl_batch_regroup': (_i, [_vp, _vp, _i, _vp, _vp]),
    'pcl_mfcc': (_i, [_vp, _i, _vp, _vp, _i, _d, _d, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, C.c_int64]),
    'pcl_timing_enable': (_i, [_vp, _i]),
    'pcl_device_count': (_i, [C.POINTER(_i)]),
    'pcl_units_upload': (_i, [_vp, _i, _i, _vp, _vp]),
    'pcl_units_download': (_i, [_vp, _vp]),
    'pcl_batch_create_labels': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    'pcl_batch_refresh_transitions': (_i, [_vp]),
    'pcl_batch_accumulate_hmm': (_i, [_vp]),
    'pcl_hmm_acc_zero': (_i, [_vp]),
    'pcl_hmm_acc_download': (_i, [_vp, _vp, _vp]),
    'pcl_mstep_transitions': (_i, [_vp]),
    'pcl_lexicon_upload': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'pcl_batch_decode': (_i, [_vp, _d, _i, _i, _i, _d, _d]),
    'pcl_batch_decode_get': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pcl_comm_init_host': (_i, [_vp, _i, _i, _vp, _vp]),
    'pcl_comm_info': (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    'pcl_em_exchange': (_i, [_vp, _d, _i, _i]),
    'pcl_pipe_info': (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    'pcl_batch_accumulate_exchange': (_i, [_vp, _i, _d, _i, _i, _i]),
    'pcl_accumulate_exchange_idle': (_i, [_vp, _d, _i, _i, _i]),
    'pcl_comm_unique_id': (_i, [_vp]),
    'pcl_comm_init': (_i, [_vp, _i, _i, _vp]),
    'pcl_stats_allreduce': (_i, [_vp]),
    'pcl_comm_destroy': (_i, [_vp]),
}


class PoccalaHipError(RuntimeError):
    """A C-ABI call returned a negative pcl_status."""

    def __init__(self, code, msg):
        super().__init__('libpoccala_hip: %s (status %d)' % (msg, code))
        self.code = code


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)   # pcl_allgather_fn

_lib = None


HW_QUEUES = None      # set by load(): {'value': GPU_MAX_HW_QUEUES as this process has it, 'effective': False if HIP was already loaded}


def _hip_runtime_loaded():
    """True when libamdhip64 is already mapped into this process (its environment knobs were read then)."""
    try:
        with open('/proc/self/maps') as f:
            return any('libamdhip64' in line for line in f)
    except OSError:
        return False


def load():
    """dlopen the library and bind every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                          '(or make -C poccala_amd/csrc).  There is no CPU fallback.' % LIB_PATH)
    # A context owns five HIP streams (main, dynamic programming, producer / frame staging, descriptors, downloads).  The runtime
    # multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default: two of the five then share a queue, and a command
    # that waits for an event (the download of step k waits for its forward-backward) holds up whatever sits behind it in that
    # queue (the scoring kernel of step k + 1).  Read when the HIP runtime starts: set before the library (and with it
    # libamdhip64) is loaded, unless the caller has chosen a value.
    global HW_QUEUES
    hip_was_loaded = _hip_runtime_loaded()
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    HW_QUEUES = dict(value=os.environ['GPU_MAX_HW_QUEUES'], effective=not hip_was_loaded)
    if hip_was_loaded and int(HW_QUEUES['value'] or 0) >= 8:
        # (the variable is read when the HIP runtime starts: a process that imported another HIP user first -- torch -- may run the
        #  five streams of a context on four hardware queues; same results, the step-to-step overlap of bench.py suffers)
        import warnings
        warnings.warn('poccala_amd: libamdhip64 was loaded before this library; GPU_MAX_HW_QUEUES=%s may not be in effect '
                      '(export it before the process starts)' % HW_QUEUES['value'], RuntimeWarning, stacklevel=2)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)        # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(a):
    """Pointer to a C-contiguous NumPy array (or None)."""
    if a is None:
        return None
    assert a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(C.c_void_p)


def as_c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)
