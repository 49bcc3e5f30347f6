"""Drop-in for the hot-path part of AcousticModel/AcousticModel.py.

Mirrored (starred): `embedded` :957-1014, `viterbi` :1016-1027, `discriminate`
:937-955, `VirtualState` :1029-1043, plus `init_unit` :164-226 / `init_parameter` :228-240 so that a
per-utterance E-step or alignment can be written exactly as the reference's workers write it
(`multi_embedded_training_1` :884-916, `multi_process_data` :723-768).  Batched equivalents that keep
everything on the GPU are `estep_batch` / `align_batch`.  Orchestration (Pool fan-out, file walking,
flat start, audio) is out of scope.
"""
import os

import numpy as np

from ..Exceptions import NullLog, UnitFileExistsError
from .._lib import PCL_F32, PCL_F64
from ..runtime import default_engine
from ..StatisticalModel.Clustering import Clustering
from ..StatisticalModel.DataInitialization import DataInitialization
from ..StatisticalModel.LHMM import LHMM


class AcousticModel(DataInitialization):
    def __init__(self, log=None, unit_type='XIF_tone', mode=0, processes=None, job_id=0, console=True, state_num=5,
                 mix_level=1, dct_num=13, delta_1=True, delta_2=True, parameters_path=None):
        super().__init__()
        self.log = log if log is not None else NullLog()
        self.__unit_type = unit_type
        self.__state_num = state_num
        self.__mix_level = mix_level
        self.__vector_size = dct_num * (3 if delta_2 else 2 if delta_1 else 1)     # AcousticModel.py:84-88
        self.__address = parameters_path or os.environ.get('parameters_file_path', '.')
        self.__loaded_units = []
        self.processes = processes or 1
        # the two workers at batched speed (multi_embedded_training_1 / multi_process_data below): what the calls queued
        self.__queued = {'train': [], 'align': []}
        self.__queued_frames = 0
        self.__flush_registered = self.__warned_worker = False
        self.__data_files = 0
        self.flush_frames = 1 << 19            # flush by itself once this many frames are queued (bounds the host copy)
        self.worker_precision = None           # None: f32-class E-step, float64 alignment (bit-exact paths)
        self.worker_units = None               # {unit: LHMM}: unit models held in memory instead of re-read from the parameter tree
        self.worker_engine = None

    loaded_units = property(lambda self: self.__loaded_units)
    statenum = property(lambda self: self.__state_num)

    def load_unit(self, unit_type=None, unit_file=None):
        """AcousticModel.load_unit (AcousticModel.py:134-161): the inventory is the file `$unit_file_path/<unit_type>` (unit_type given:
        it becomes the model's unit type); a missing file raises UnitFileExistsError; the parameter tree `<parameters path>/<unit_type>`
        is created.  Line 1 of the file is a description, the rest comma-separated units.  `unit_file=` (keyword, not in the
        reference) reads a file by its path instead and creates nothing."""
        if unit_file is None:
            if unit_type:
                self.__unit_type = unit_type
            base = os.environ.get('unit_file_path')
            if base is None:
                raise KeyError('unit_file_path')             # (the reference reads os.environ['unit_file_path'] at import, AcousticModel.py:25)
            unit_file = os.path.abspath(base) + '/' + self.__unit_type
            if not os.path.exists(unit_file):
                raise UnitFileExistsError(self.__unit_type, self.log)
            os.makedirs(os.path.join(self.__address, self.__unit_type), exist_ok=True)
        with open(unit_file) as f:
            f.readline()
            for line in f:
                self.__loaded_units.extend(u for u in line.strip('\n').split(',') if u)

    # ------------------------------------------------------------------ unit HMMs
    def init_unit(self, unit, new_log=True, fix_code=0):
        """5-state left-right unit HMM: entry / S-2 GMM states / exit, flat-start transitions
        (AcousticModel.py:164-226)."""
        s = self.__state_num
        states = {i: unit for i in range(s)}
        transmat = np.zeros((s, s))
        transmat[0][1] = 1.
        for j in range(1, s - 1):
            transmat[j][j] = 0.5
            transmat[j][j + 1] = 0.5
        gmm = [Clustering.GMM(self.log, dimension=self.__vector_size, mix_level=self.__mix_level, gmm_id=k)
               for k in range(s - 2)]
        prof = [AcousticModel.VirtualState(1.)] + gmm + [AcousticModel.VirtualState(0.)]
        return LHMM(states, s, self.log, transmat=transmat, profunc=prof, fix_code=fix_code)

    def unit_path(self, unit):
        return '%s/%s/%s' % (self.__address, self.__unit_type, unit)

    def init_parameter(self, unit, hmm):
        hmm.init_parameter(self.unit_path(unit))
        for i in range(1, self.__state_num - 1):
            hmm.profunction[i].init_parameter(self.unit_path(unit))

    def save_parameter(self, unit, hmm):
        os.makedirs(self.unit_path(unit), exist_ok=True)
        hmm.save_parameter(self.unit_path(unit))
        for i in range(1, self.__state_num - 1):
            hmm.profunction[i].save_parameter(self.unit_path(unit))

    def save_acc(self, unit, hmm):
        os.makedirs(self.unit_path(unit), exist_ok=True)
        hmm.save_acc(self.unit_path(unit))
        for i in range(1, self.__state_num - 1):
            hmm.profunction[i].save_acc(self.unit_path(unit))

    # ------------------------------------------------------------------ A7 embedded (AcousticModel.py:957-1014)
    def embedded(self, label, hmm_list, data_index, alter=15):
        s = self.__state_num
        e = s - 2
        n = e * len(hmm_list) + 2

        def embedded_states():
            names = [label[0]]
            for u in label:
                names.extend([u] * e)
            names.append(label[len(label) - 1])
            return dict(enumerate(names))

        def embedded_transmat():
            a = np.zeros((n, n))
            a[:s - 1, :s] = hmm_list[0].transmat[:-1]
            for i in range(len(label)):
                lo = i * e + 1
                a[lo:lo + e, lo - 1:lo - 1 + s] = hmm_list[i].transmat[1:-1]
            return a

        def embedded_prob():
            rows = [hmm_list[0].B_p[data_index][0:-1]]
            for i in range(1, len(label)):
                rows.append(hmm_list[i].B_p[data_index][1:-1, :])
            rows.append(hmm_list[len(label) - 1].B_p[data_index][-1:, :])
            return np.concatenate(rows, axis=0)

        def embedded_pi():
            return np.ones((n,)) / n

        funcs = [embedded_states, embedded_transmat, embedded_prob, embedded_pi]
        return [funcs[k]() for k in range(4) if 2 ** (3 - k) & alter]

    # ------------------------------------------------------------------ A14 wrappers
    def viterbi(self, complex_states, complex_transmat, complex_prob, complex_pi):
        return LHMM.viterbi(self.log, complex_states, complex_transmat, complex_prob, complex_pi, convert=True,
                            show_mark_state=True)

    @staticmethod
    def discriminate(unit, sequence):
        """Frame indices labelled `unit`, split into contiguous runs (AcousticModel.py:937-955)."""
        loc = np.where(np.asarray(sequence) == unit)[0]
        if len(loc) == 0:
            return []
        return np.split(loc, np.where(np.diff(loc) != 1)[0] + 1)

    class VirtualState(object):
        """Constant-score pdf of the non-emitting entry/exit states (AcousticModel.py:1029-1043)."""

        constant_score = True                  # point() does not look at the frame: callers may evaluate it once per utterance

        def __init__(self, p=0.):
            self.__p = p

        def point(self, x, log=False, standard=False, record=False):
            if log:
                with np.errstate(divide='ignore'):
                    return np.log(self.__p)
            return self.__p

    # ------------------------------------------------------------------ batched, GPU-resident equivalents
    def _model_arrays(self, unit_hmms):
        units = sorted(unit_hmms)
        idx = {u: i for i, u in enumerate(units)}
        arrs = [unit_hmms[u].profunction[1 + k].model_arrays() for u in units for k in range(self.__state_num - 2)]
        trans = [np.asarray(unit_hmms[u].transmat, dtype=np.float64) for u in units]
        return units, idx, (np.stack([a[0] for a in arrs]), np.stack([a[1] for a in arrs]), np.stack([a[2] for a in arrs])), trans

    def _sentence_batch(self, labels, data_list, unit_hmms, engine):
        units, idx, (mean, var, w), trans = self._model_arrays(unit_hmms)
        engine.load_model(mean, var, w)
        engine.load_units(np.stack(trans))
        lens = np.array([len(d) for d in data_list], dtype=np.int32)
        begin = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
        engine.load_frames(np.concatenate([np.asarray(d) for d in data_list], axis=0))
        unit_ids = [np.array([idx[u] for u in lab], dtype=np.int32) for lab in labels]
        b = engine.label_batch(unit_ids, lens, begin)        # AcousticModel.embedded for every utterance, in the library
        return b, b.N, units, idx

    def align_batch(self, labels, data_list, unit_hmms, precision=PCL_F64, engine=None):
        """Forced alignment of many utterances at once (call stack C, AcousticModel.py:723-768):
        returns [(point, unit-name sequence)] like AcousticModel.viterbi does per utterance."""
        engine = engine or default_engine()
        b, n, units, idx = self._sentence_batch(labels, data_list, unit_hmms, engine)
        b.score(precision)
        b.viterbi()
        pts, paths = b.get('point'), b.get('path')
        b.close()
        e = self.__state_num - 2
        out = []
        for u, lab in enumerate(labels):
            names = np.array([lab[0]] + [x for x in lab for _ in range(e)] + [lab[-1]])
            out.append((float(pts[u]), names[paths[u]]))
        return out

    def estep_batch(self, labels, data_list, unit_hmms, fix_code=0, precision=PCL_F32, engine=None):
        """E-step of many utterances at once (call stack B, AcousticModel.py:884-916).  Returns
        (stats, hmm_acc, logp): `stats` = linear-domain GMM statistics per state (engine.stats_download,
        states ordered unit-major over sorted(unit_hmms)); `hmm_acc[unit]` = (ksai_acc (S-2,S),
        gamma_acc (S-2,)) log-domain, merged over every occurrence as LHMM.add_acc does (LHMM.py:149-161,
        473-500).  Everything per utterance x label position runs in the library: the sentence HMMs are built
        from the labels there (pcl_batch_create_labels) and the per-unit merge is a kernel
        (pcl_batch_accumulate_hmm); nothing here loops over utterances."""
        engine = engine or default_engine()
        b, n, units, idx = self._sentence_batch(labels, data_list, unit_hmms, engine)
        b.score(precision)
        b.forward_backward(fix_pi=bool(fix_code & 1))
        engine.stats_zero()                       # GMM statistics and the per-unit HMM accumulators
        stats = None
        if not fix_code & 2:
            b.accumulate(precision)
            stats = engine.stats_download()
        hmm_acc = {}
        if not fix_code & 4:
            b.accumulate_hmm()
            ks, ga = engine.hmm_acc_download()
            seen = np.bincount(b.labels, minlength=len(units)) > 0
            hmm_acc = {unit: (ks[i], ga[i]) for i, unit in enumerate(units) if seen[i]}
        logp = b.get('logp')
        b.close()
        return stats, hmm_acc, logp

    # ------------------------------------------------------------------ what follows the two workers in the reference
    def segment_batch(self, labels, data_list, unit_hmms, precision=PCL_F64, engine=None):
        """Alignment -> per-unit data segments for many utterances (the loop at AcousticModel.py:758-764:
        discriminate + __save_data): returns {unit: [frame blocks]} where each block is the (n,D) slice of one
        contiguous run of that unit, in utterance order; an utterance whose path misses a label unit is
        dropped, as the reference does (:754-757)."""
        out = {}
        dropped = []
        for u, (point, names) in enumerate(self.align_batch(labels, data_list, unit_hmms, precision, engine)):
            if len(set(names)) < len(set(labels[u])):
                dropped.append(u)
                continue
            data = np.asarray(data_list[u])
            for unit in set(labels[u]):
                for loc in AcousticModel.discriminate(unit, names):
                    out.setdefault(unit, []).append(data[loc])
        return out, dropped

    def eq_segment(self, data, arg=None, mode='e', save=None):
        """AcousticModel.__eq_segment (AcousticModel.py:587-627).  mode 'e': the utterance is cut into len(arg) equal
        chunks, one per label unit, handed to `save(unit, block)` (the reference's __save_data pickles them); the
        remainder frames are dropped.  mode 'g': one block cut into `arg` slices, the last takes the remainder."""
        if mode == 'e':
            chunk = len(data) // len(arg)
            for i, u in enumerate(arg):
                if save is not None:
                    save(u, data[i * chunk:(i + 1) * chunk])
            return None
        if mode == 'g':
            chunk = len(data) // arg
            out = [data[k * chunk:(k + 1) * chunk] for k in range(arg - 1)]
            out.append(data[(arg - 1) * chunk:])
            return out
        raise ValueError('eq_segment: mode must be \'e\' or \'g\'')   # the reference raises ClassError here

    def get_gmmdata(self, data):
        """AcousticModel.__get_gmmdata (AcousticModel.py:629-644): the data of each of the unit's S-2 GMM states."""
        gmm_num = self.__state_num - 2
        parts = [self.eq_segment(block, gmm_num, mode='g') for block in data]
        return [np.concatenate([p_[k] for p_ in parts], axis=0) for k in range(gmm_num)]

    def regroup_batch(self, labels, data_list, unit_hmms, precision=PCL_F64, engine=None):
        """Forced alignment -> the re-estimation data of every GMM state, for many utterances, on the device: Viterbi
        (`align_batch`'s kernels), then `pcl_batch_regroup` instead of discriminate + __save_data + __get_gmmdata
        (AcousticModel.py:758-764, 629-644).  Returns ({unit: [S-2 arrays (n_k, D)]}, dropped): for each unit the frames
        of its GMM states, in utterance then time order; an utterance whose path misses a label unit is dropped
        (:754-757)."""
        engine = engine or default_engine()
        b, n, units, idx = self._sentence_batch(labels, data_list, unit_hmms, engine)
        b.score(precision)
        b.viterbi()
        s = self.__state_num
        row_unit = []
        for lab in labels:
            ids = np.repeat([idx[u] for u in lab], s - 2)
            row_unit.append(np.concatenate([[ids[0]], ids, [ids[-1]]]).astype(np.int32))     # entry / exit rows: first / last unit
        fu, fk = b.regroup(row_unit, s - 2)
        b.close()
        names = {i: u for u, i in idx.items()}
        out, dropped = {}, []
        for u, lab in enumerate(labels):
            if len(np.unique(fu[u])) < len(set(lab)):
                dropped.append(u)
                continue
            data = np.asarray(data_list[u])
            for i in np.unique(fu[u]):
                slot = out.setdefault(names[i], [[] for _ in range(s - 2)])
                for k in range(s - 2):
                    slot[k].append(data[(fu[u] == i) & (fk[u] == k)])
        d = np.asarray(data_list[0]).shape[1]
        return ({unit: [np.concatenate(parts, axis=0) if parts else np.zeros((0, d)) for parts in slots]
                 for unit, slots in out.items()}, dropped)

    # ------------------------------------------------------------------ the reference's two workers, at batched speed
    # The reference fans its corpus out over a Pool, one call per utterance (AcousticModel.py:861-870, :709-712): every call
    # builds one LHMM per label position, scores, embeds, runs Baum-Welch / Viterbi and writes files.  Called like that, one
    # utterance at a time, a GPU sees 300 frames per launch (extra.zero_change_route: ~5 k frames/s).  The two methods below keep
    # the reference's signatures and file outputs but DEFER the work: a call queues (label, data); flush_workers() -- called by
    # the caller after its loop, and by itself every `flush_frames` frames -- runs everything queued as one batch through
    # estep_batch / segment_batch and writes what the per-utterance workers would have written, merged per unit:
    #   multi_embedded_training_1 -> <unit>/HMM/{ksai-acc,gamma-acc}/*.npy, <unit>/GMM_k/{acc,alpha-acc,mean-acc,covariance-acc}/*.npy
    #                                (one file set per unit and flush instead of one per utterance x label position; the merge of
    #                                 multi_embedded_training_2 -- LHMM.init_acc / GMM.init_acc, a log-sum-exp over the files --
    #                                 gives the same sums)
    #   multi_process_data        -> <unit>/data/*.pkl, one pickle per contiguous run of the unit (AcousticModel.__save_data)
    # No Pool, no Controller: only these two entry points.
    def _worker_units(self, units, init):
        if self.worker_units is not None:
            return {u: self.worker_units[u] for u in units}
        out = {}
        for u in units:                          # init_unit + init_parameter ONCE per unit and flush (the reference: per label position)
            hmm = self.init_unit(unit=u, new_log=init)
            self.init_parameter(unit=u, hmm=hmm)
            out[u] = hmm
        return out

    def _enqueue(self, kind, item):
        """Queue one worker call.  The data are COPIED (the reference's Pool pickles them at call time, AcousticModel.py:865-870: a caller
        may refill its feature buffer right after the call).  Inside a multiprocessing worker (the reference's own pattern:
        pool.apply_async(self.multi_embedded_training_1, ...)) the process may end with the task, and multiprocessing children leave
        through os._exit without running atexit handlers, so nothing is deferred there: the call is flushed at once (the
        per-utterance rate; the batched rate needs the calls made from one process), with one warning."""
        import multiprocessing as mp
        self.__queued[kind].append(item)
        self.__queued_frames += len(item[1])
        if not self.__flush_registered:
            import atexit
            import weakref
            ref = weakref.ref(self)
            atexit.register(lambda: ref() is not None and ref()._flush_at_exit())
            self.__flush_registered = True
        if mp.parent_process() is not None:
            if not self.__warned_worker:
                import warnings
                warnings.warn('poccala_amd.AcousticModel: multi_embedded_training_1 / multi_process_data called inside a multiprocessing worker; '
                              'the deferred batch cannot outlive the task, so every call is flushed at once (one utterance per launch chain). '
                              'Call the workers from one process and flush_workers() after the loop for the batched rate.', RuntimeWarning)
                self.__warned_worker = True
            self.flush_workers()
        elif self.__queued_frames >= self.flush_frames:
            self.flush_workers()

    def _flush_at_exit(self):
        """atexit: whatever is still queued is run and written (with a warning: the caller forgot flush_workers())."""
        if self.__queued['train'] or self.__queued['align']:
            import warnings
            warnings.warn('poccala_amd.AcousticModel: %d queued worker calls were still waiting at interpreter exit; flushing them now '
                          '(call flush_workers() after the loop)' % (len(self.__queued['train']) + len(self.__queued['align'])), RuntimeWarning)
            try:
                self.flush_workers()
            except Exception as e:             # noqa: the interpreter is going down; say what is lost
                warnings.warn('poccala_amd.AcousticModel: the exit flush failed, the queued accumulators are LOST: %r' % (e,), RuntimeWarning)

    def __del__(self):
        try:
            q = self.__queued
        except AttributeError:
            return
        if q['train'] or q['align']:
            import warnings
            warnings.warn('poccala_amd.AcousticModel: dropped with %d queued worker calls that were never flushed (flush_workers()): their '
                          'accumulator / data files were NOT written' % (len(q['train']) + len(q['align'])), RuntimeWarning)

    def multi_embedded_training_1(self, label, data, init, *args):
        """AcousticModel.multi_embedded_training_1(label, data, init, show_q, load_num, file_count, fix_code)
        (AcousticModel.py:884-916), deferred: see flush_workers."""
        fix_code = int(args[3]) if len(args) > 3 else 0
        self._enqueue('train', (list(label), np.array(data, copy=True), bool(init), fix_code))

    def multi_process_data(self, label, data, init, *args):
        """AcousticModel.multi_process_data(label, data, init, load_num, file_count, fix_code) (AcousticModel.py:723-768):
        init -> the utterance is cut into equal chunks per label unit at once (__eq_segment mode 'e', host work); otherwise the
        forced alignment is deferred: see flush_workers."""
        if init:
            self.eq_segment(np.asarray(data), list(label), mode='e', save=self.save_data)
            return
        self._enqueue('align', (list(label), np.array(data, copy=True), False, 0))

    def save_data(self, unit, unit_data):
        """AcousticModel.__save_data (AcousticModel.py:331-351): one pickle per block under <unit>/data/ (the reference names the
        file by pid and second, so two blocks of one second overwrite each other; a running number is appended here)."""
        import pickle
        import time
        path = self.unit_path(unit) + '/data'
        os.makedirs(path, exist_ok=True)
        self.__data_files += 1
        with open('%s/%s_data_%s_%s_%06d.pkl' % (path, unit, os.getpid(), int(time.time()), self.__data_files), 'wb') as f:
            pickle.dump(unit_data, f, protocol=pickle.HIGHEST_PROTOCOL)

    def flush_workers(self):
        """Run everything multi_embedded_training_1 / multi_process_data queued, as batches.  Returns
        {'train': (utterances, frames, logp array), 'align': (utterances, frames, dropped utterance indices)}."""
        out = {}
        queued, self.__queued = self.__queued, {'train': [], 'align': []}
        self.__queued_frames = 0
        eng = self.worker_engine
        if queued['train']:
            lp_all, n_utt, n_fr = [], 0, 0
            # one E-step per (fix_code, init): fix_code decides what is accumulated, init whether the unit logs start afresh
            # (init_unit(new_log=init), AcousticModel.py:897) -- every queued call keeps its own flag
            # groups in the order their first call was queued (the reference applies `init` per call, in call order: a first call with
            # init=True must restart the unit logs BEFORE the later init=False calls write to them -- ADVICE r5: sorted() ran False first)
            for fix_code, init in dict.fromkeys((q[3], q[2]) for q in queued['train']):
                part = [q for q in queued['train'] if q[3] == fix_code and q[2] == init]
                labels, datas = [q[0] for q in part], [q[1] for q in part]
                units = self._worker_units(sorted({u for lab in labels for u in lab}), init)
                stats, hmm_acc, logp = self.estep_batch(labels, datas, units, fix_code=fix_code,
                                                        precision=PCL_F32 if self.worker_precision is None else self.worker_precision, engine=eng)
                self.save_batch_acc(stats, hmm_acc, units)
                lp_all.append(logp)
                n_utt += len(part)
                n_fr += int(sum(len(d) for d in datas))
            out['train'] = (n_utt, n_fr, np.concatenate(lp_all))
        if queued['align']:
            labels, datas = [q[0] for q in queued['align']], [q[1] for q in queued['align']]
            units = self._worker_units(sorted({u for lab in labels for u in lab}), False)
            blocks, dropped = self.segment_batch(labels, datas, units, precision=PCL_F64 if self.worker_precision is None else self.worker_precision,
                                                 engine=eng)
            for unit, runs in blocks.items():                                   # AcousticModel.py:758-764
                for run in runs:
                    self.save_data(unit, run)
            for u in dropped:
                self.log.note('viterbi切分失败', cls='w')                         # AcousticModel.py:755 (the utterance is discarded)
            out['align'] = (len(labels), int(sum(len(d) for d in datas)), dropped)
        return out

    def save_batch_acc(self, stats, hmm_acc, unit_hmms):
        """Write the result of `estep_batch` as reference-format accumulator files (log domain, float64,
        SURVEY T3): <unit>/HMM/{ksai-acc,gamma-acc}/..., <unit>/GMM_k/{acc,alpha-acc,mean-acc,covariance-acc}/...
        so that the reference's multi_embedded_training_2 (AcousticModel.py:918-935) can merge them."""
        units = sorted(unit_hmms)
        e = self.__state_num - 2
        if stats is not None:
            with np.errstate(divide='ignore'):                          # ln 0 = -inf for a state no frame reached
                ln = {k: np.log(stats[k]) for k in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc')}
        for ui, unit in enumerate(units):
            hmm = unit_hmms[unit]
            touched = False
            # the files hold THIS batch's accumulators only (the reference writes one set per utterance from fresh unit
            # objects, AcousticModel.py:897-913): start from ln 0, or a second call would save batch 1 + batch 2 and the
            # merge (init_acc) would count batch 1 twice
            hmm.reset_acc()
            if unit in hmm_acc:
                hmm.add_acc(hmm_acc[unit][0], hmm_acc[unit][1])
                touched = True
            if stats is not None:
                for k in range(e):
                    j = ui * e + k
                    g = hmm.profunction[1 + k]
                    g.acc = ln['acc'][j]
                    g.alpha_acc = float(ln['alpha_acc'][j])
                    g.mean_acc = ln['mean_acc'][j]
                    g.covariance_acc = ln['cov_acc'][j]
                    touched = touched or stats['alpha_acc'][j] > 0
            if touched:
                self.save_acc(unit, hmm)
                hmm.reset_acc()
