"""Batched host API over the C-ABI (include/poccala_hip.h).

`Engine` owns one GPU context (model + frame matrix + E-step statistics); `Batch` is a set of
sentence-level HMMs processed together.  The drop-in classes in StatisticalModel/ and
AcousticModel/ call these with batch size 1; bench.py and the batched entry points
(`score_labels`, `estep_labels`, `align_labels`) call them with whole corpus shards.

Reference call stacks replaced:
  (B) AcousticModel.multi_embedded_training_1  AcousticModel/AcousticModel.py:884-916
  (C) AcousticModel.multi_process_data         AcousticModel/AcousticModel.py:723-768
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import GET, PCL_F32, PCL_F64, PCL_MAX_PASS, PCL_ROW_ENTRY, PCL_ROW_EXIT, PoccalaHipError, as_c, ptr


def device_count():
    """HIP devices visible to this process (0 without a GPU)."""
    n = C.c_int(0)
    _lib.load().pcl_device_count(C.byref(n))
    return n.value


class Engine(object):
    """One GPU: `Engine(device)`.  Raises if libpoccala_hip.so is missing or no GPU is present."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        rc = self._lib.pcl_init(int(device), C.byref(self._ctx))
        if rc != 0:
            msg = self._lib.pcl_last_error(None)
            self._ctx = None
            raise PoccalaHipError(rc, (msg or b'pcl_init failed').decode())
        self.device = int(device)
        self.J = self.M = self.D = 0
        self.n_units = self.S = 0
        self.F = 0
        self._model_key = self._frames_key = None
        self._stream_pool, self._stream_pinned = {}, [None]       # Decoder.decode_stream: batches per chunk shape, staging buffer
        self._batches = weakref.WeakSet()   # live batches: destroyed before the context (a sweep makes millions: dead ones leave by themselves)
        self._pinned = []    # page-locked host allocations (pinned_empty)
        self._pinned_sizes, self._pinned_named = {}, {}
        self._slot_busy = {}   # result slot -> weakref of the batch with un-waited copies into it
        self._staged = None

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            raise PoccalaHipError(rc, self._lib.pcl_last_error(self._ctx).decode())

    def close(self):
        if getattr(self, '_ctx', None):
            for b in list(self._batches):
                b.close()
            self._batches = weakref.WeakSet()
            self._stream_pool, self._stream_pinned = {}, [None]
            self._lib.pcl_sync(self._ctx)             # result copies into the page-locked buffers freed below may still be in flight
            for p in self._pinned:
                self._lib.pcl_host_free(self._ctx, p)
            self._pinned, self._pinned_sizes, self._pinned_named = [], {}, {}
            self._lib.pcl_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self._lib.pcl_sync(self._ctx))

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int()
        mem = C.c_size_t()
        self._check(self._lib.pcl_device_info(self._ctx, name, 256, C.byref(cus), C.byref(mem)))
        return dict(name=name.value.decode(), cus=cus.value, hbm_bytes=mem.value)

    def enable_timing(self, on=True):
        """Record HIP events around every kernel launch for `kernel_time` (off by default: a training run pays nothing)."""
        self._check(self._lib.pcl_timing_enable(self._ctx, 1 if on else 0))

    def kernel_time(self, which):
        """(total_ms, launches) of a kernel group since the last query (HIP events on the ctx stream; `enable_timing` first)."""
        ms = C.c_float()
        n = C.c_int()
        self._check(self._lib.pcl_kernel_time(self._ctx, which.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def clock_probe(self, spin_us=1000):
        """The shader clock (MHz) the chip holds right now, measured on the device beside whatever is running (s_memtime against
        the constant 100 MHz s_memrealtime over `spin_us` microseconds); rocm-smi's sclk is the requested level."""
        mhz = C.c_double()
        self._check(self._lib.pcl_clock_probe(self._ctx, int(spin_us), C.byref(mhz)))
        return mhz.value

    # ------------------------------------------------------------------ model / frames
    def load_model(self, mean, var, weight, logdet=False):
        """mean, var (J,M,D) -- var is the DIAGONAL of the reference's (M,D,D) covariance (util.py:23);
        weight (J,M).  `logdet=True` swaps quirk Q1's constant for the textbook one (not parity)."""
        mean = as_c(mean, np.float64)
        var = as_c(var, np.float64)
        weight = as_c(weight, np.float64)
        if mean.ndim != 3 or var.shape != mean.shape or weight.shape != mean.shape[:2]:
            raise ValueError('model shapes: mean %s var %s weight %s' % (mean.shape, var.shape, weight.shape))
        J, M, D = mean.shape
        key = self._digest((mean, var, weight), ('model', bool(logdet)))
        if key is not None and key == self._model_key:
            return                                   # the same content is resident (and no M-step has touched it)
        self._check(self._lib.pcl_model_upload(self._ctx, J, M, D, ptr(mean), ptr(var), ptr(weight),
                                               1 if logdet else 0))
        self.J, self.M, self.D = J, M, D
        self._model_key = key

    _DIGEST_MAX = 64 << 20

    @staticmethod
    def _digest(arrays, tag):
        """Content key of small uploads (the drop-in classes re-send the same unit model / utterance for every call);
        None above 64 MB, where hashing would cost more than it saves."""
        if sum(a.nbytes for a in arrays) > Engine._DIGEST_MAX:
            return None
        head = repr((tag, [(a.shape, str(a.dtype)) for a in arrays])).encode()
        try:                                   # a 128-bit non-cryptographic hash at memory speed where it is installed
            import xxhash
            h = xxhash.xxh3_128(head)
        except ImportError:
            import hashlib
            h = hashlib.blake2b(head, digest_size=16)
        for a in arrays:
            h.update(memoryview(np.ascontiguousarray(a)).cast('B'))
        return h.digest()

    def load_frames(self, frames):
        """(F,D) float32 or float64 MFCC rows of every utterance of the shard, concatenated."""
        frames = np.asarray(frames)
        if frames.ndim != 2:
            raise ValueError('frames must be (F,D)')
        if frames.dtype == np.float32:
            f, dt = as_c(frames, np.float32), PCL_F32
        else:
            f, dt = as_c(frames, np.float64), PCL_F64
        key = self._digest((f,), 'frames')
        if key is not None and key == self._frames_key:
            return
        self._check(self._lib.pcl_frames_upload(self._ctx, f.shape[0], f.shape[1], ptr(f), dt))
        self.F = f.shape[0]
        self._frames_key = key

    # ------------------------------------------------------------------ streaming (BASELINE config 5: a corpus fed chunk by chunk)
    def stage_frames(self, frames):
        """Queue the H2D copy of the NEXT chunk's (F,D) float32 rows on the copy stream, into the frame slot that is not
        current; returns at once.  The array must stay untouched until swap_frames() returns (a pinned_empty() array makes
        the copy truly asynchronous)."""
        f = np.asarray(frames)
        if f.ndim != 2 or f.dtype != np.float32 or not f.flags['C_CONTIGUOUS']:
            raise ValueError('stage_frames: a C-contiguous (F,D) float32 array')
        self._check(self._lib.pcl_frames_stage(self._ctx, f.shape[0], f.shape[1], ptr(f)))
        self._staged = f

    def swap_frames(self):
        """Wait for the staged copy and make that chunk the current frame matrix (batches created from now on index it)."""
        self._check(self._lib.pcl_frames_swap(self._ctx))
        self.F = self._staged.shape[0]
        self._staged = None
        self._frames_key = None

    def pinned_empty(self, shape, dtype=np.float32):
        """A NumPy array over page-locked host memory (freed with the engine)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._check(self._lib.pcl_host_alloc(self._ctx, n, C.byref(p)))
        self._pinned.append(p)
        self._pinned_sizes[p.value] = n
        buf = (C.c_char * n).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def pinned_free(self, arr_or_ptr):
        """Release one pinned_empty() allocation before the engine closes."""
        addr = arr_or_ptr if isinstance(arr_or_ptr, int) else arr_or_ptr.__array_interface__['data'][0]
        for k, p in enumerate(self._pinned):
            if p.value == addr:
                self._lib.pcl_host_free(self._ctx, p)
                del self._pinned[k]
                self._pinned_sizes.pop(addr, None)
                for name in [n for n, a in self._pinned_named.items() if a.__array_interface__['data'][0] == addr]:
                    del self._pinned_named[name]
                return
        raise ValueError('not a pinned_empty() allocation of this engine')

    def pinned_bytes(self):
        """Page-locked host bytes this engine holds (soak tests assert that a ragged stream keeps it bounded)."""
        return int(sum(self._pinned_sizes.get(p.value, 0) for p in self._pinned))

    def _pinned_views(self, name, shapes):
        """Arrays of the given (shape, dtype) list over ONE grow-only page-locked buffer per name (kept with the engine, reused by
        every caller of that name; grown by doubling).  A block that was outgrown is RETIRED, not freed: views handed out earlier
        may still be alive (a dict from an earlier result_buffers() call passes fetch_async's size checks), and an asynchronous
        DMA into freed page-locked memory corrupts the host.  Retired blocks go with the engine; doubling bounds
        them by the size of the live block."""
        need = [int(np.prod(sh)) * np.dtype(dt).itemsize for sh, dt in shapes]
        off = np.concatenate([[0], np.cumsum([(n + 63) // 64 * 64 for n in need])])
        total = int(off[-1]) or 64
        cur = self._pinned_named.get(name)
        if cur is None or cur.nbytes < total:
            cur = self.pinned_empty((max(total, 2 * (cur.nbytes if cur is not None else 0)),), np.uint8)
            self._pinned_named[name] = cur               # (the outgrown block stays in self._pinned until close())
        return [cur[int(off[k]):int(off[k]) + need[k]].view(dt).reshape(sh) for k, (sh, dt) in enumerate(shapes)]

    def _slots_of(self, arrays):
        """result slots whose LIVE block holds any of the arrays"""
        out = set()
        for a in arrays:
            addr = a.__array_interface__['data'][0]
            for name, blk in self._pinned_named.items():
                if name.startswith('fetch_results_'):
                    lo = blk.__array_interface__['data'][0]
                    if lo <= addr < lo + blk.nbytes:
                        out.add(int(name[len('fetch_results_'):]))
        return out

    def batch(self, N, T, frame_begin=None):
        return Batch(self, N, T, frame_begin)

    # ------------------------------------------------------------------ unit inventory, label-built batches
    def load_units(self, unit_trans):
        """unit_trans: (n_units, S, S) transition matrices (LHMM.transmat of every unit HMM, AcousticModel.py:174-181).
        Unit i owns GMM states i*(S-2) .. of the loaded model.  ln A is taken here with np.log, the values the
        reference's forward / Viterbi use (LHMM.py:340,571)."""
        t = as_c(unit_trans, np.float64)
        if t.ndim != 3 or t.shape[1] != t.shape[2]:
            raise ValueError('unit_trans must be (n_units, S, S)')
        with np.errstate(divide='ignore'):
            lt = np.ascontiguousarray(np.log(t))
        self._check(self._lib.pcl_units_upload(self._ctx, t.shape[0], t.shape[1], ptr(t), ptr(lt)))
        self.n_units, self.S = int(t.shape[0]), int(t.shape[1])

    def units_download(self):
        t = np.empty((self.n_units, self.S, self.S))
        self._check(self._lib.pcl_units_download(self._ctx, ptr(t)))
        return t

    def label_batch(self, unit_ids, T, frame_begin):
        """Sentence HMMs of a list of labels (AcousticModel.embedded for every utterance at once, in the library)."""
        return Batch(self, None, T, frame_begin, unit_ids=unit_ids)

    def load_lexicon(self, tree):
        """The flat pronunciation tree of Lexicon.PronunciationLexicon.compile (units = ids of the loaded inventory)."""
        nu = as_c(tree['node_units'], np.int32)
        nn = as_c(tree['node_nunits'], np.int32)
        if nu.shape != (len(nn), 2):                  # the device layout is [n_nodes][2] (PronunciationLexicon.compile(max_units=2))
            raise ValueError('tree node_units must be (n_nodes, 2), got %r' % (nu.shape,))
        cp = as_c(tree['child_ptr'], np.int32)
        ci = as_c(tree['child_idx'], np.int32) if len(tree['child_idx']) else np.zeros(1, dtype=np.int32)
        nw = as_c(tree['node_word'], np.int32)
        ro = as_c(tree['roots'], np.int32)
        self._check(self._lib.pcl_lexicon_upload(self._ctx, len(nn), ptr(nu), ptr(nn), ptr(cp), ptr(ci), ptr(nw), len(ro), ptr(ro)))
        self.n_nodes = len(nn)

    def all_state_batch(self, T, frame_begin):
        """Batch whose emission rows are [entry, every GMM state 0..J-1, exit]: what the decoder reads (BASELINE config 5:
        no label, every state of the inventory is scored for every frame)."""
        T = as_c(T, np.int32).reshape(-1)
        b = Batch(self, np.full(len(T), self.J + 2, dtype=np.int32), T, frame_begin)
        rows = np.concatenate([[PCL_ROW_ENTRY], np.arange(self.J), [PCL_ROW_EXIT]]).astype(np.int32)
        b.set_states([rows] * len(T))
        return b

    def hmm_acc_zero(self):
        self._check(self._lib.pcl_hmm_acc_zero(self._ctx))

    def hmm_acc_download(self):
        """(ksai_acc (n_units, S-2, S), gamma_acc (n_units, S-2)): log domain, -inf where nothing was added
        (LHMM.ksai_acc / LHMM.gamma_acc of every unit, merged over all label positions of all batches)."""
        ks = np.empty((self.n_units, self.S - 2, self.S))
        ga = np.empty((self.n_units, self.S - 2))
        self._check(self._lib.pcl_hmm_acc_download(self._ctx, ptr(ks), ptr(ga)))
        return ks, ga

    def mstep_transitions(self):
        """LHMM.update_param's transition update for every unit that occurred (LHMM.py:519-520)."""
        self._check(self._lib.pcl_mstep_transitions(self._ctx))

    # ------------------------------------------------------------------ E-step statistics
    def stats_zero(self):
        self._check(self._lib.pcl_stats_zero(self._ctx))

    def stats_download(self, moments=True):
        """Linear-domain sums: acc (J,M), alpha_acc (J,), and unless moments=False mean_acc (J,M,D), cov_acc (J,M,D)."""
        acc = np.empty((self.J, self.M))
        al = np.empty((self.J,))
        if not moments:
            self._check(self._lib.pcl_stats_download(self._ctx, ptr(acc), ptr(al), None, None))
            return dict(acc=acc, alpha_acc=al)
        me = np.empty((self.J, self.M, self.D))
        co = np.empty((self.J, self.M, self.D))
        self._check(self._lib.pcl_stats_download(self._ctx, ptr(acc), ptr(al), ptr(me), ptr(co)))
        return dict(acc=acc, alpha_acc=al, mean_acc=me, cov_acc=co)

    def accumulate_prune(self, log2_threshold):
        """Approximate mode of Batch.accumulate: leave out (frame, state) pairs with gamma_t(j) < 2^log2_threshold
        (default: only pairs that are exactly zero in the kernel's arithmetic)."""
        self._check(self._lib.pcl_accumulate_prune(self._ctx, float(log2_threshold)))

    def mstep(self, c_covariance=1e-3):
        """GMM.update_param for every state on the device (Clustering.py:682-693); rebuilds the scoring layouts."""
        self._check(self._lib.pcl_mstep(self._ctx, float(c_covariance)))
        self._model_key = None                       # the resident model is no longer what was uploaded

    def model_download(self):
        """(mean (J,M,D), var (J,M,D), weight (J,M)) float64 master copy."""
        mean = np.empty((self.J, self.M, self.D))
        var = np.empty((self.J, self.M, self.D))
        w = np.empty((self.J, self.M))
        self._check(self._lib.pcl_model_download(self._ctx, ptr(mean), ptr(var), ptr(w)))
        return mean, var, w

    def model_conditioning(self):
        """(cond (J,) float32, cond_max): conditioning of the centred expansion per state; states above cond_max
        are scored and accumulated by the direct-form kernels instead of the matrix-core ones."""
        cond = np.empty(self.J, dtype=np.float32)
        cmax = np.empty(1, dtype=np.float32)
        self._check(self._lib.pcl_model_conditioning(self._ctx, ptr(cond), ptr(cmax)))
        return cond, float(cmax[0])

    def score_occupancy(self, workgroups_per_cu):
        """0: the scoring kernel fills the CUs (default); 2: it leaves a third of each CU's registers to kernels of other streams
        (the streamed decoder's token passing).  Same results either way."""
        self._check(self._lib.pcl_score_occupancy(self._ctx, int(workgroups_per_cu)))

    def model_split_info(self):
        """(n_off (J,) int32, limit): mixtures per state that are off the matrix-core path (their own conditioning is beyond
        cond_max; the direct-form kernels evaluate them and the parts are merged); a state with more than `limit` of them
        leaves the matrix cores as a whole."""
        n_off = np.empty(self.J, dtype=np.int32)
        lim = np.empty(1, dtype=np.int32)
        self._check(self._lib.pcl_model_split_info(self._ctx, ptr(n_off), ptr(lim)))
        return n_off, int(lim[0])

    def coarse_pairs(self, reset=True):
        """(frame, mixture) pairs of off-pipe mixtures evaluated exactly since the last reset (counted under PCL_COARSE_STATS=1)."""
        n = np.zeros(1, dtype=np.uint64)
        self._check(self._lib.pcl_coarse_counter(self._ctx, ptr(n), 1 if reset else 0))
        return int(n[0])

    def coarse_counters(self, reset=True):
        """(pairs evaluated exactly, tiles the coarse pass gave up and the direct-form subset kernel rescored) since the last reset
        (counted under PCL_COARSE_STATS=1)."""
        n, g = np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.uint64)
        self._check(self._lib.pcl_coarse_counters(self._ctx, ptr(n), ptr(g), 1 if reset else 0))
        return int(n[0]), int(g[0])

    # ------------------------------------------------------------------ RCCL
    def comm_unique_id(self):
        buf = np.zeros(128, dtype=np.uint8)
        rc = self._lib.pcl_comm_unique_id(ptr(buf))
        if rc != 0:
            raise PoccalaHipError(rc, self._lib.pcl_last_error(None).decode())
        return buf.tobytes()

    def comm_init(self, rank, nranks, unique_id):
        """ncclCommInitRank.  RCCL prints a version banner with C stdio on stdout; a caller that owes its parent
        exactly one line of stdout (bench.py) would see it arrive AFTER its own output when libc flushes at exit, so
        file descriptor 1 points at stderr while the communicator is created."""
        import ctypes, os, sys
        buf = np.frombuffer(bytes(unique_id), dtype=np.uint8).copy()
        libc = ctypes.CDLL(None)
        sys.stdout.flush()
        libc.fflush(None)
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            rc = self._lib.pcl_comm_init(self._ctx, int(rank), int(nranks), ptr(buf))
            libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        self._check(rc)

    def comm_init_host(self, rank, nranks, allgather_bytes):
        """Rehearsal transport for several ranks on ONE device (RCCL refuses duplicate devices): `allgather_bytes(b)`
        returns every rank's bytes in rank order (poccala_amd.distributed.Control.allgather_bytes)."""
        def cb(user, send, nbytes, recv_all):
            try:
                parts = allgather_bytes(C.string_at(send, nbytes))
                if len(parts) != nranks or any(len(x) != nbytes for x in parts):
                    return 1
                C.memmove(recv_all, b''.join(parts), nbytes * nranks)
                return 0
            except Exception:          # never let an exception cross the C boundary
                return 2
        self._host_cb = _lib.ALLGATHER_FN(cb)       # keep the thunk alive as long as the context
        self._check(self._lib.pcl_comm_init_host(self._ctx, int(rank), int(nranks), self._host_cb, None))

    def comm_info(self):
        r, n, t, c = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.pcl_comm_info(self._ctx, C.byref(r), C.byref(n), C.byref(t), C.byref(c)))
        return dict(rank=r.value, nranks=n.value, transport={0: 'none', 1: 'rccl', 2: 'host-rehearsal'}[t.value], rccl_nranks=c.value)

    def pipe_info(self):
        """(chunks, released_early) of the last accumulate_exchange: how many state chunks left for the exchange while the pass still ran."""
        k, e = C.c_int(), C.c_int()
        self._check(self._lib.pcl_pipe_info(self._ctx, C.byref(k), C.byref(e)))
        return k.value, e.value

    def stats_allreduce(self):
        self._check(self._lib.pcl_stats_allreduce(self._ctx))

    def wire_payload(self):
        """The cheaper wire format of the E-step exchange for callers that opt in (bench.py --payload auto): float32 as soon as there is
        a wire (the 1.94 GB of statistics and the model cross xGMI as f32, half the bytes; every rank continues
        from the same rounded model; mean_acc travels as mean_acc - (bias + c_j) acc so that the rounding is not amplified), float64
        on one rank.  NOT the default: with it the re-estimated model depends on the world size."""
        return PCL_F32 if self.comm_info()['nranks'] > 1 else PCL_F64

    def accumulate_exchange_idle(self, c_covariance=1e-3, payload=PCL_F64, update_transitions=False, n_chunks=8):
        """Batch.accumulate_exchange for a rank that has no batch for this last pass (the others do): the same chunk exchanges, in the
        same order, on whatever this rank's statistics hold."""
        self._check(self._lib.pcl_accumulate_exchange_idle(self._ctx, float(c_covariance), int(payload), 1 if update_transitions else 0, int(n_chunks)))
        self._model_key = None

    def em_exchange(self, c_covariance=1e-3, payload=PCL_F64, update_transitions=False):
        """reduce-scatter of the statistics by state range -> M-step on the owned states -> all-gather of the model
        (+ merge of the per-unit HMM accumulators, + the transition update on request).  One rank: the M-step.
        payload: PCL_F64 (default: the reference's reducer is float64, LHMM.py:256-290 / Clustering.py:314-367, and the model then
        does not depend on the world size) or PCL_F32 (half the bytes on the wire; `wire_payload()` picks it when there is a wire)."""
        self._check(self._lib.pcl_em_exchange(self._ctx, float(c_covariance), int(payload), 1 if update_transitions else 0))
        self._model_key = None


class Batch(object):
    """U sentence HMMs.  N[u] states, T[u] frames; matrices cross the boundary in the reference's
    (N,T) float64 layout, one array per utterance."""

    def __init__(self, engine, N, T, frame_begin=None, unit_ids=None):
        self.eng = engine
        self._lib = engine._lib
        self.T = as_c(T, np.int32).reshape(-1)
        self._b = C.c_void_p()
        if unit_ids is not None:
            # label-built: N_u = (S-2) L_u + 2 (AcousticModel.py:966); structure built by pcl_batch_create_labels
            if isinstance(unit_ids, np.ndarray) and unit_ids.ndim == 2:      # equal-length labels as one (U, L) array: no per-utterance Python
                lens = np.full(unit_ids.shape[0], unit_ids.shape[1], dtype=np.int32)
                flat = np.ascontiguousarray(unit_ids, dtype=np.int32).reshape(-1)
            else:
                lens = np.array([len(l) for l in unit_ids], dtype=np.int32)
                flat = np.ascontiguousarray(np.concatenate([np.asarray(l, dtype=np.int32).reshape(-1) for l in unit_ids]), dtype=np.int32)
            self.N = ((engine.S - 2) * lens + 2).astype(np.int32)
            if self.N.shape != self.T.shape or self.N.size == 0 or frame_begin is None:
                raise ValueError('labels, T and frame_begin must be equal-length, non-empty')
            self.U = int(self.N.size)
            fb = as_c(frame_begin, np.int64).reshape(-1)
            logpi = np.ascontiguousarray(np.log(1.0 / self.N.astype(np.float64)))     # np.log(np.ones(N) / N), AcousticModel.py:1005
            self.label_len, self.labels = lens, flat
            engine._check(self._lib.pcl_batch_create_labels(engine._ctx, self.U, ptr(lens), ptr(flat), ptr(self.T), ptr(fb),
                                                            ptr(logpi), C.byref(self._b)))
        else:
            self.N = as_c(N, np.int32).reshape(-1)
            if self.N.shape != self.T.shape or self.N.size == 0:
                raise ValueError('N and T must be equal-length, non-empty')
            self.U = int(self.N.size)
            fb = None if frame_begin is None else as_c(frame_begin, np.int64).reshape(-1)
            engine._check(self._lib.pcl_batch_create(engine._ctx, self.U, ptr(self.N), ptr(self.T), ptr(fb),
                                                     C.byref(self._b)))
        engine._batches.add(self)
        self._fetch_slots = set()                     # engine result slots with un-waited copies of this batch (fetch_async)
        n64, t64 = self.N.astype(np.int64), self.T.astype(np.int64)
        self._nt_off = np.concatenate([[0], np.cumsum(n64 * t64)])
        self._nn_off = np.concatenate([[0], np.cumsum(n64 * n64)])
        self._n_off = np.concatenate([[0], np.cumsum(n64)])
        self._t_off = np.concatenate([[0], np.cumsum(t64)])
        self._nnz_off = None
        self.nz_index = None

    def close(self):
        if getattr(self, '_b', None):
            if getattr(self.eng, '_ctx', None):       # the context frees its batches when it closes
                if getattr(self, '_fetch_slots', None):       # copies into an engine slot nobody waited for: the slot's next user must not race them
                    self._lib.pcl_batch_fetch_wait(self._b)
                    self._fetch_slots.clear()
                self._lib.pcl_batch_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        self.eng._check(rc)

    def _ragged(self, arrays, shape_of, dtype=np.float64):
        out = []
        for u, a in enumerate(arrays):
            a = np.asarray(a, dtype=dtype)
            if a.shape != shape_of(u):
                raise ValueError('utterance %d: expected shape %s, got %s' % (u, shape_of(u), a.shape))
            out.append(a.reshape(-1))
        if len(out) != self.U:
            raise ValueError('expected %d arrays, got %d' % (self.U, len(out)))
        return np.ascontiguousarray(np.concatenate(out), dtype=dtype)

    # ------------------------------------------------------------------ inputs
    def set_transitions(self, logA, logpi):
        """logA[u] (N,N) = np.log(transmat), logpi[u] (N,) = np.log(pi) -- logged by the caller so that
        Viterbi sees bit-identical operands (LHMM.py:571,577)."""
        a = self._ragged(logA, lambda u: (self.N[u], self.N[u]))
        p = self._ragged(logpi, lambda u: (self.N[u],))
        # stored transitions (ln A > -inf) per utterance, row-major: the order of get('ksai_nz')
        self.nz_index = [np.nonzero(~np.isneginf(np.asarray(m, dtype=np.float64))) for m in logA]
        self._nnz_off = np.concatenate([[0], np.cumsum([len(ix[0]) for ix in self.nz_index])])
        self._check(self._lib.pcl_batch_set_transitions(self._b, ptr(a), ptr(p)))

    def set_states(self, row_state):
        s = self._ragged(row_state, lambda u: (self.N[u],), dtype=np.int32)
        self._check(self._lib.pcl_batch_set_states(self._b, ptr(s)))

    def set_emissions(self, B):
        b = self._ragged(B, lambda u: (self.N[u], self.T[u]))
        self._check(self._lib.pcl_batch_set_emissions(self._b, ptr(b)))

    def set_posteriors(self, lgamma):
        """ln gamma_t(i) per utterance (N,T), for accumulate() without a forward-backward on this batch."""
        g = self._ragged(lgamma, lambda u: (self.N[u], self.T[u]))
        self._check(self._lib.pcl_batch_set_posteriors(self._b, ptr(g)))

    # ------------------------------------------------------------------ kernels
    def score(self, precision=PCL_F32):
        self._check(self._lib.pcl_batch_score(self._b, int(precision)))

    def forward_backward(self, fix_pi=False, threshold=0.64):
        self._check(self._lib.pcl_batch_forward_backward(self._b, 1 if fix_pi else 0, float(threshold)))

    def viterbi(self, end_state_back=False):
        self._check(self._lib.pcl_batch_viterbi(self._b, 1 if end_state_back else 0))

    def accumulate(self, precision=PCL_F32):
        self._check(self._lib.pcl_batch_accumulate(self._b, int(precision)))

    def accumulate_exchange(self, precision=PCL_F32, c_covariance=1e-3, payload=PCL_F64, update_transitions=False, n_chunks=8):
        """The last accumulate pass of an E-step and the exchange in one pipelined call (pcl_batch_accumulate_exchange): state
        chunks leave for reduce-scatter -> M-step -> all-gather -> derive as soon as the pass is done with them.
        payload as in Engine.em_exchange (float64 unless the caller opts in to the f32 wire)."""
        self._check(self._lib.pcl_batch_accumulate_exchange(self._b, int(precision), float(c_covariance), int(payload),
                                                             1 if update_transitions else 0, int(n_chunks)))
        self.eng._model_key = None

    def accumulate_hmm(self):
        """Per-unit ksai_acc / gamma_acc of every label position (LHMM.update_acc + add_acc); label-built batches only."""
        self._check(self._lib.pcl_batch_accumulate_hmm(self._b))

    def decode(self, beam=0.85, min_distinct=8, candidate=5, max_tokens=4096):
        """Token passing over the loaded pronunciation tree (Decoder.py:91-167, 250-288) for every utterance of an all-state
        batch.  Returns a list (one dict per utterance): final = [(node, score, hist)] best first, history = [(prev, node)],
        n_tokens (T,) live tokens after every frame, overflow."""
        self.decode_launch(beam, min_distinct, candidate, max_tokens)
        return self.decode_results()

    def decode_launch(self, beam=0.85, min_distinct=8, candidate=5, max_tokens=4096):
        """Queue the token passing (on the library's second stream, behind this batch's scoring) and return at once."""
        e = self.eng.S - 2
        lp = np.log(np.array([1.0 / (e + 2), 1.0 / (2 * e + 2)]))               # np.log(np.ones(N) / N), AcousticModel.py:1005
        self._check(self._lib.pcl_batch_decode(self._b, float(beam), int(min_distinct), int(candidate), int(max_tokens), float(lp[0]), float(lp[1])))
        self._dec_candidate = int(candidate)

    def decode_fetch(self):
        """Wait for the queued token passing of THIS batch (nothing else) and bring its result arrays to the host."""
        U, c, tm = self.U, self._dec_candidate, int(self.T.max())
        # page-locked result buffers: a device-to-host copy into pageable memory is staged by the runtime and was seen to return only
        # when the scoring kernel of the NEXT chunk had finished (59 ms per fetch instead of 37).  They belong to the ENGINE, one
        # grow-only set: the call below is synchronous and the arrays are copied out before it returns, so every batch can share them
        # (a ragged stream makes a new batch per chunk: per-batch buffers grew without bound and put 9 hipHostMalloc on every chunk)
        shapes = [((U,), np.int32), ((U, c), np.int32), ((U, c), np.float64), ((U, c), np.int32), ((U,), np.int32),
                  ((U, tm), np.int32), ((U, tm), np.int32), ((U, tm), np.int32), ((U,), np.int32)]
        nf, node, score, hist, hn, hp, hnode, nt, ov = self.eng._pinned_views('decode_results', shapes)
        self._check(self._lib.pcl_batch_decode_get(self._b, ptr(nf), ptr(node), ptr(score), ptr(hist), ptr(hn), ptr(hp), ptr(hnode), ptr(nt), ptr(ov)))
        return tuple(x.copy() for x in (nf, node, score, hist, hn, hp, hnode, nt, ov)) + (self.T.copy(),)

    @staticmethod
    def decode_unpack(raw):
        """decode_fetch's arrays as the per-utterance dicts decode() returns (host work only)."""
        nf, node, score, hist, hn, hp, hnode, nt, ov, T = raw
        out = []
        for u in range(len(nf)):
            out.append(dict(final=list(zip(node[u, :nf[u]].tolist(), score[u, :nf[u]].tolist(), hist[u, :nf[u]].tolist())),
                            history=list(zip(hp[u, :hn[u]].tolist(), hnode[u, :hn[u]].tolist())),
                            n_tokens=nt[u, :T[u]].copy(), overflow=bool(ov[u])))
        return out

    def decode_results(self):
        return self.decode_unpack(self.decode_fetch())

    def refresh_transitions(self):
        """Take the engine's CURRENT unit transitions (after mstep_transitions / em_exchange); label-built batches only."""
        self._check(self._lib.pcl_batch_refresh_transitions(self._b))

    # ------------------------------------------------------------------ results on their way to the host while the GPU goes on
    def _result_shapes(self):
        """name -> (elements, dtype) of fetch_async's destinations for THIS batch (pcl_batch_sizes)."""
        nnz = np.zeros(1, dtype=np.int64)            # (label-built batches: the library built the transition lists)
        self._check(self._lib.pcl_batch_sizes(self._b, None, None, None, ptr(nnz)))
        return dict(logp=(self.U, np.float64), lgamma=(int(self._nt_off[-1]), np.float64), ksai_nz=(int(nnz[0]), np.float64),
                    path=(int(self._t_off[-1]), np.int32), point=(self.U, np.float64))

    def result_buffers(self, want=('logp', 'lgamma', 'ksai_nz', 'path', 'point'), slot=0):
        """Page-locked destination arrays for fetch_async.  They belong to the ENGINE, one grow-only set per `slot` (a pipeline that
        keeps k result sets in flight uses slots 0 .. k-1): calling this once per batch of a stream re-uses the slot's memory
        instead of page-locking ~150 MB per call until the engine closes.  ALIASING: every batch that asks for the same slot gets
        views of the SAME memory -- two batches with results in flight at once need two slots.  Asking for a slot while another
        live batch still has un-waited copies into it raises instead of handing out memory a DMA is writing."""
        busy = self.eng._slot_busy.get(slot)
        other = busy() if busy is not None else None
        if other is not None and other is not self and getattr(other, '_b', None) and other._fetch_slots:
            raise RuntimeError('result_buffers: slot %d still receives the results of another batch (fetch_wait() it first, or use another slot)' % slot)
        shapes = self._result_shapes()
        names = [k for k in ('logp', 'lgamma', 'ksai_nz', 'path', 'point') if k in want]
        views = self.eng._pinned_views('fetch_results_%d' % slot, [((shapes[k][0],), shapes[k][1]) for k in names])
        return dict(zip(names, views))

    def fetch_async(self, bufs):
        """Queue the device-to-host copies of the results named in `bufs` (result_buffers()) behind everything this batch has
        queued; returns at once.  fetch_wait() blocks until they have landed; the next compute call on this batch waits for them
        on the device.  Every buffer is checked against this batch's sizes first: a stale or short buffer would otherwise be a
        host-memory overrun written by an asynchronous DMA."""
        shapes = self._result_shapes()
        for k, a in bufs.items():
            if k not in shapes:
                raise KeyError('fetch_async: unknown result %r' % (k,))
            n, dt = shapes[k]
            if not isinstance(a, np.ndarray) or a.dtype != dt or a.size != n or not a.flags['C_CONTIGUOUS']:
                raise ValueError('fetch_async: %s must be a C-contiguous %s array of %d elements (got %s %s)'
                                 % (k, np.dtype(dt).name, n, getattr(a, 'dtype', type(a)), getattr(a, 'shape', '')))
        g = lambda k: ptr(bufs[k]) if k in bufs else None
        self._check(self._lib.pcl_batch_fetch_async(self._b, g('logp'), g('lgamma'), g('ksai_nz'), g('path'), g('point')))
        for slot in self.eng._slots_of(bufs.values()):          # (engine-owned destinations: the slot is busy until fetch_wait)
            self.eng._slot_busy[slot] = weakref.ref(self)
            self._fetch_slots.add(slot)

    def fetch_wait(self):
        self._check(self._lib.pcl_batch_fetch_wait(self._b))
        self._fetch_slots.clear()

    def lgamma_views(self, flat):
        """fetch_async's time-major ln gamma_t(j) as the reference's (N, T) matrices (transposed views, no copy)."""
        return [flat[self._nt_off[u]:self._nt_off[u + 1]].reshape(self.T[u], self.N[u]).T for u in range(self.U)]

    # ------------------------------------------------------------------ outputs
    def regroup(self, row_unit, gmm_num):
        """Per-frame regrouping after viterbi() (AcousticModel.py:758-764 + __get_gmmdata :629-644): row_unit[u] = (N_u,)
        unit id of every HMM row.  Returns (frame_unit, frame_k): per utterance (T_u,) int32 arrays, the unit the path
        is in and the GMM state (slice of the unit's run) each frame is given to."""
        ru = self._ragged(row_unit, lambda u: (self.N[u],), dtype=np.int32)
        fu = np.empty(int(self._t_off[-1]), dtype=np.int32)
        fk = np.empty(int(self._t_off[-1]), dtype=np.int32)
        self._check(self._lib.pcl_batch_regroup(self._b, ptr(ru), int(gmm_num), ptr(fu), ptr(fk)))
        cut = lambda flat: [flat[self._t_off[u]:self._t_off[u + 1]] for u in range(self.U)]
        return cut(fu), cut(fk)

    def get(self, what):
        """List of per-utterance arrays (or a (U,...) array for per-utterance scalars)."""
        code = GET[what]
        if what in ('B', 'alpha', 'beta', 'lgamma'):
            flat = np.empty(int(self._nt_off[-1]))
            self._check(self._lib.pcl_batch_get(self._b, code, ptr(flat)))
            return [flat[self._nt_off[u]:self._nt_off[u + 1]].reshape(self.N[u], self.T[u]) for u in range(self.U)]
        if what == 'ksai':
            flat = np.empty(int(self._nn_off[-1]))
            self._check(self._lib.pcl_batch_get(self._b, code, ptr(flat)))
            return [flat[self._nn_off[u]:self._nn_off[u + 1]].reshape(self.N[u], self.N[u]) for u in range(self.U)]
        if what == 'ksai_nz':
            if self._nnz_off is None:
                raise RuntimeError('set_transitions first')
            flat = np.empty(int(self._nnz_off[-1]))
            self._check(self._lib.pcl_batch_get(self._b, code, ptr(flat)))
            return [flat[self._nnz_off[u]:self._nnz_off[u + 1]] for u in range(self.U)]
        if what in ('gamma', 'pi'):
            flat = np.empty(int(self._n_off[-1]))
            self._check(self._lib.pcl_batch_get(self._b, code, ptr(flat)))
            return [flat[self._n_off[u]:self._n_off[u + 1]] for u in range(self.U)]
        if what == 'path':
            flat = np.empty(int(self._t_off[-1]), dtype=np.int32)
            self._check(self._lib.pcl_batch_get(self._b, code, ptr(flat)))
            return [flat[self._t_off[u]:self._t_off[u + 1]] for u in range(self.U)]
        if what in ('logp', 'point'):
            out = np.empty(self.U)
        elif what == 'npass':
            out = np.empty(self.U, dtype=np.int32)
        elif what == 'qtrace':
            out = np.empty((self.U, PCL_MAX_PASS))
        else:
            raise KeyError(what)
        self._check(self._lib.pcl_batch_get(self._b, code, ptr(out)))
        return out


# ---------------------------------------------------------------------- sentence HMM construction
def embedded_structure(n_units, unit_trans, s=5):
    """Host part of AcousticModel.embedded (AcousticModel/AcousticModel.py:957-1014) for one label:
    the (N,N) transition matrix and the uniform 1/N pi.  unit_trans: list of (S,S) matrices, one
    per label position.  The emission rows are laid out by `embedded_row_states`."""
    e = s - 2
    n = e * n_units + 2
    a = np.zeros((n, n))
    a[:s - 1, :s] = unit_trans[0][:-1]
    for i in range(n_units):
        lo = i * e + 1
        a[lo:lo + e, lo - 1:lo - 1 + s] = unit_trans[i][1:-1]
    pi = np.ones(n) / n
    return a, pi


def embedded_row_states(unit_state_ids, s=5):
    """Row -> GMM state id map of the sentence HMM: entry row, each unit's S-2 emitting states,
    exit row (AcousticModel.py:990-1001).  unit_state_ids: (L, S-2) global GMM state ids."""
    ids = np.asarray(unit_state_ids, dtype=np.int32).reshape(-1)
    return np.concatenate([[PCL_ROW_ENTRY], ids, [PCL_ROW_EXIT]]).astype(np.int32)


def make_sentence_batch(engine, unit_ids, T, frame_begin, unit_trans, s=5):
    """Build the batch for a list of labels.  unit_ids[u]: sequence of unit indices (label of
    utterance u); unit i owns GMM states i*(S-2) .. i*(S-2)+S-3 of the uploaded model and the
    transition matrix unit_trans[i] (S,S).  Returns (batch, N)."""
    e = s - 2
    n = np.array([e * len(l) + 2 for l in unit_ids], dtype=np.int32)
    b = engine.batch(n, T, frame_begin)
    log_a, log_pi, rows = [], [], []
    with np.errstate(divide='ignore'):
        for lab in unit_ids:
            lab = np.asarray(lab, dtype=np.int64)
            a, pi = embedded_structure(len(lab), [unit_trans[i] for i in lab], s)
            log_a.append(np.log(a))
            log_pi.append(np.log(pi))
            rows.append(embedded_row_states(lab[:, None] * e + np.arange(e)[None, :], s))
    b.set_transitions(log_a, log_pi)
    b.set_states(rows)
    return b, n
