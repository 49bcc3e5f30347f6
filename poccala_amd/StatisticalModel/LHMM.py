"""Drop-in for StatisticalModel/LHMM.py: the log-domain HMM class of the reference, same constructor,
properties and methods, with forward/backward/xi/gamma/pi and Viterbi running on the GPU through
libpoccala_hip.so (hmm_dp.hip) and emission scoring through gmm_score.hip.

Class surface mirrored: LHMM(states, statesnum, log, t, transmat, profunc, probmat, pi,
hmm_list, fix_code) LHMM.py:19-20; cal_observation_pro :163; baulm_welch :526; update_acc :473;
add_acc :149; update_param :509; viterbi (static) :546; properties :100-145; save/init :192-290.
"""
import configparser
import os
import time

import numpy as np

from ..runtime import scratch_engine
from .._lib import PCL_F64
from .DataInitialization import DataInitialization
from .util import log_sum_exp, matrix_log_sum_exp, save_acc_file


def _np_log(a):
    with np.errstate(divide='ignore'):
        return np.log(np.asarray(a, dtype=np.float64))


class LHMM(DataInitialization):
    def __init__(self, states, statesnum, log, t=None, transmat=None, profunc=None, probmat=None, pi=None,
                 hmm_list=None, fix_code=0):
        super().__init__()
        self.__states = states
        self.__statesnum = statesnum
        self.__hmm_size = len(states)
        self.log = log
        self.__t = [] if t is None else t
        n = len(states)
        self.__profunction = profunc
        self.__transmat = transmat if transmat is not None else np.ones((n, n)) / n
        self.__pi = pi if pi is not None else np.ones((n,)) / n
        assert profunc is not None or probmat is not None, 'one of profunc / probmat must be given'   # LHMM.py:69
        self.__result_f = None
        self.__result_b = None
        self.__result_p = probmat
        self.__ksai = np.zeros((n, n))
        self.__gamma = np.zeros((n,))
        self.__lgamma = None
        self.__ksai_acc = np.full((statesnum - 2, statesnum), -np.inf)     # LHMM.py:84-85
        self.__gamma_acc = np.full((statesnum - 2,), -np.inf)
        self.__acc_file = True
        self.__hmm_list = [self] if hmm_list is None else hmm_list
        self.__fix_code = None
        self.__fix_list = None
        self.fix_code = fix_code
        self.q_trace = []
        self.n_pass = 0

    # ------------------------------------------------------------------ properties (LHMM.py:100-145)
    states = property(lambda self: self.__states)
    t = property(lambda self: self.__t)
    transmat = property(lambda self: self.__transmat)
    B_p = property(lambda self: self.__result_p)
    pi = property(lambda self: self.__pi)
    profunction = property(lambda self: self.__profunction)
    ksai_acc = property(lambda self: self.__ksai_acc)
    gamma_acc = property(lambda self: self.__gamma_acc)

    @property
    def fix_code(self):
        return self.__fix_code

    @fix_code.setter
    def fix_code(self, fix_code):
        """3-bit mask [transmat, pdf, pi] (LHMM.py:35-36,140-145); the pdf is locked automatically for
        an HMM that was given probmat and is its own hmm_list."""
        self.__fix_code = fix_code
        self.__fix_list = [bool(fix_code & 2 ** e) for e in range(2, -1, -1)]
        if self.__profunction is None and self in self.__hmm_list:
            self.__fix_list[1] = True

    # ------------------------------------------------------------------ small mutators (LHMM.py:295-331)
    def clear_result_buffer(self):
        self.__result_f = None
        self.__result_b = None

    def clear_data(self):
        super().clear_data()
        self.__t = []

    def change_t(self, t):
        self.__t = t

    def add_T(self, t):
        self.__t.extend(t)

    def change_pi(self, pi):
        self.__pi = pi

    def change_A(self, transmat):
        self.__transmat = transmat

    # ------------------------------------------------------------------ A12 add_acc (LHMM.py:149-161)
    def add_acc(self, ksai_value, gamma_value):
        self.__ksai_acc = matrix_log_sum_exp([self.__ksai_acc, ksai_value], axis_x=self.__statesnum - 2)
        self.__gamma_acc = matrix_log_sum_exp([self.__gamma_acc.reshape(1, -1), gamma_value.reshape(1, -1)],
                                              axis_x=1).reshape(-1)

    def reset_acc(self):
        """Accumulators back to ln 0 (their state after __init__, LHMM.py:84-85)."""
        self.__ksai_acc = np.full((self.__statesnum - 2, self.__statesnum), -np.inf)
        self.__gamma_acc = np.full((self.__statesnum - 2,), -np.inf)

    # ------------------------------------------------------------------ A6 cal_observation_pro (LHMM.py:163-187)
    def cal_observation_pro(self, data, data_t, normalize=False, standard=False, precision=PCL_F64):
        if standard:
            raise NotImplementedError('standard=True is never used on the hot path (SURVEY quirk Q3)')
        n = len(self.__states)
        lens = [int(data_t[d]) for d in range(len(data))]
        gmm_rows = [i for i in range(n) if hasattr(self.__profunction[i], 'model_arrays')]
        out = [np.empty((n, lens[d])) for d in range(len(data))]
        if gmm_rows:
            eng = scratch_engine()
            arrs = [self.__profunction[i].model_arrays() for i in gmm_rows]
            frames = np.concatenate([np.asarray(data[d], dtype=np.float64)[:lens[d]] for d in range(len(data))], axis=0)
            if frames.shape[1] != arrs[0][0].shape[1]:
                from ..Exceptions import DataDimensionError
                raise DataDimensionError(arrs[0][0].shape[1], frames.shape[1], self.log)
            eng.load_model(np.stack([a[0] for a in arrs]), np.stack([a[1] for a in arrs]), np.stack([a[2] for a in arrs]))
            eng.load_frames(frames)
            begin = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int64)
            b = eng.batch([len(gmm_rows) + 2] * len(data), lens, begin)
            rows = np.concatenate([[-1], np.arange(len(gmm_rows)), [-2]]).astype(np.int32)
            b.set_states([rows] * len(data))
            b.score(precision)
            scored = b.get('B')
            b.close()
            for d in range(len(data)):
                for k, i in enumerate(gmm_rows):
                    out[d][i] = scored[d][1 + k]
        for i in range(n):
            if i in gmm_rows:
                continue
            prof = self.__profunction[i]           # any object with .point (VirtualState: constant ln p)
            for d in range(len(data)):
                if getattr(prof, 'constant_score', False) and lens[d]:      # the reference calls it once per frame (LHMM.py:176-180)
                    out[d][i] = prof.point(data[d][0], log=True, standard=standard, record=True)
                else:
                    out[d][i] = [prof.point(data[d][f], log=True, standard=standard, record=True) for f in range(lens[d])]
        if normalize:
            for d in range(len(data)):
                out[d] = out[d] - np.array([[log_sum_exp(out[d][j])] for j in range(n)])
        self.__result_p = out

    # ------------------------------------------------------------------ A8..A11 + A10 baulm_welch (LHMM.py:526-544)
    def _device_pass(self, log_pi, fix_pi, threshold):
        eng = scratch_engine()
        n_utt = self.datasize if self.datasize else len(self.__result_p)
        n = self.__hmm_size
        ts = [self.__result_p[d].shape[1] for d in range(n_utt)]
        b = eng.batch([n] * n_utt, ts)
        la = _np_log(self.__transmat)
        b.set_transitions([la] * n_utt, [log_pi] * n_utt)
        b.set_emissions([np.asarray(self.__result_p[d], dtype=np.float64) for d in range(n_utt)])
        b.forward_backward(fix_pi=fix_pi, threshold=threshold)
        res = {k: b.get(k) for k in ('alpha', 'beta', 'ksai', 'gamma', 'pi', 'logp', 'npass', 'qtrace', 'lgamma')}
        b.close()
        return res

    def baulm_welch(self, show_q=False):
        if len(self.__t) == 0:
            self.__t = [len(self.data[i]) for i in range(self.datasize)]
        if self.__profunction is not None and (self.__result_p is None or self.__result_f is None):
            self.cal_observation_pro(self.data, self.__t, normalize=False)            # LHMM.py:384-386
        fix_pi = self.__fix_list[2]
        n_utt = self.datasize if self.datasize else len(self.__result_p)
        if n_utt == 1:
            # the hot path: one utterance per embedded HMM (AcousticModel.py:906-910); pass loop on the device
            res = self._device_pass(_np_log(self.__pi), fix_pi, 0.64)
            npass = int(res['npass'][0])
            qs = [-np.inf] + [float(q) for q in res['qtrace'][0][:npass - 1]]
            self.__ksai, self.__gamma = res['ksai'][0], res['gamma'][0]
            if not fix_pi:
                self.__pi = res['pi'][0]
        else:
            # several utterances in one LHMM couple through the merged pi (LHMM.py:454-466): one device
            # pass per iteration (threshold = inf stops after a single pass), merge on the host
            q, qs = -np.inf, []
            while True:
                qs.append(q)
                res = self._device_pass(_np_log(self.__pi).reshape(-1), True, np.inf)
                self.__ksai = matrix_log_sum_exp(res['ksai'], axis_x=self.__hmm_size)
                self.__gamma = matrix_log_sum_exp([g.reshape(1, -1) for g in res['gamma']], axis_x=1).reshape(-1)
                if not fix_pi:
                    # (1,N) and un-normalised (sums to the number of utterances), exactly as LHMM.py:465-466 leaves it
                    self.__pi = np.exp(matrix_log_sum_exp([lg[:, 0].reshape(1, -1) for lg in res['lgamma']], axis_x=1))
                q_new = log_sum_exp(np.concatenate([a[:, -1] for a in res['alpha']]))       # LHMM.py:417-422
                if q_new - q > 0.64:                                                        # LHMM.py:539
                    q = q_new
                    continue
                break
        for q in qs:
            self.log.note('HMM current likelihood:%f' % q, cls='i', show_console=show_q)   # LHMM.py:535
        self.q_trace, self.n_pass = qs, len(qs)
        self.__result_f, self.__result_b, self.__lgamma = res['alpha'], res['beta'], res['lgamma']
        self.update_acc()

    # ------------------------------------------------------------------ A12 update_acc (LHMM.py:473-507)
    def update_acc(self):
        ksai_view = self.__ksai[1:-1, :]
        gamma_view = self.__gamma[1:-1]
        e = self.__statesnum - 2
        n_utt = self.datasize if self.datasize else len(self.__result_p)
        for idx in range(n_utt):
            if not self.__fix_list[1]:
                l_in = self.__lgamma[idx][1:-1]               # (alpha+beta)[1:-1] - sum_value, computed on the device
                b_in = self.__result_p[idx][1:-1, :]
            x0 = y0 = 0
            for hmm in self.__hmm_list:
                if not self.__fix_list[0]:
                    hmm.add_acc(ksai_view[y0:y0 + e, x0:x0 + self.__statesnum], gamma_view[y0:y0 + e])
                if not self.__fix_list[1]:
                    gmms = hmm.profunction[1:-1]
                    for i in range(e):
                        gmms[i].update_acc(l_in[y0 + i], b_in[y0 + i], self.data[idx])
                y0 += e
                x0 += e

    # ------------------------------------------------------------------ A15 update_param (LHMM.py:509-524)
    def update_param(self, show_q=False, show_a=False, c_covariance=1e-3):
        if not self.__acc_file:
            return
        if not self.__fix_list[0]:
            with np.errstate(all='ignore'):
                self.__transmat[1:-1, :] = np.exp(self.__ksai_acc - self.__gamma_acc.reshape((self.__statesnum - 2, 1)))
        if not self.__fix_list[1]:
            for i in range(1, len(self.__profunction) - 1):
                self.__profunction[i].update_param(show_q=show_q, c_covariance=c_covariance)
        self.log.note('HMM transition matrix:\n' + str(self.__transmat), cls='i', show_console=show_a)

    # ------------------------------------------------------------------ A14 viterbi (LHMM.py:546-609)
    @staticmethod
    def viterbi(log, states, transmat, prob, pi, convert=False, end_state_back=False, show_mark_state=False):
        prob = np.asarray(prob, dtype=np.float64)
        s_len, t = prob.shape
        assert s_len == len(states), 'number of states does not match the score matrix'     # LHMM.py:563
        eng = scratch_engine()
        b = eng.batch([s_len], [t])
        b.set_transitions([_np_log(transmat)], [_np_log(pi)])     # np.log on the host: bit-identical operands
        b.set_emissions([prob])
        b.viterbi(end_state_back=end_state_back)
        point = float(b.get('point')[0])
        mark_state = b.get('path')[0].astype(np.float64)          # the reference returns float64 indices
        b.close()
        if convert:
            c_mark_state = np.array([states[k] for k in mark_state])
            log.note('Viterbi Sequence:\n' + str(c_mark_state), cls='i', show_console=show_mark_state)
            return point, c_mark_state
        log.note('Viterbi Sequence:\n' + str(mark_state), cls='i', show_console=show_mark_state)
        return point, mark_state

    # ------------------------------------------------------------------ T3 files (LHMM.py:192-290)
    def save_parameter(self, path):
        p = path + '/HMM'
        os.makedirs(p, exist_ok=True)
        np.save(p + '/transmat.npy', self.__transmat)
        np.save(p + '/pi.npy', self.__pi)
        cfg = configparser.ConfigParser()
        cfg.add_section('Configuration')
        cfg.set('Configuration', 'FIX_CODE', str(self.__fix_code))
        with open(p + '/HMM_config.ini', 'w+') as f:
            cfg.write(f)

    def init_parameter(self, path):
        p = path + '/HMM'
        self.__transmat = np.load(p + '/transmat.npy')
        self.__pi = np.load(p + '/pi.npy')
        # HMM_config.ini is write-only in the reference (LHMM.py:248-254 passes a file object to
        # ConfigParser.read, so fix_code is never read back); kept

    def save_acc(self, path):
        p = path + '/HMM'
        stamp = int(time.time())
        for sub, name, val in (('ksai-acc', 'ksai_acc', self.__ksai_acc), ('gamma-acc', 'gamma_acc', self.__gamma_acc)):
            save_acc_file(p + '/' + sub, name, stamp, val)

    def init_acc(self, path):
        p = path + '/HMM'
        dk, dg = p + '/ksai-acc', p + '/gamma-acc'
        self.__acc_file = os.path.isdir(dk) and os.path.isdir(dg)
        if not self.__acc_file:
            return
        ks = [np.load(os.path.join(dk, f)) for f in sorted(os.listdir(dk))]
        gs = [np.load(os.path.join(dg, f)) for f in sorted(os.listdir(dg))]
        if ks:
            self.__ksai_acc = matrix_log_sum_exp(ks, axis_x=self.__statesnum - 2)
            self.__gamma_acc = log_sum_exp(np.array(gs).T, vector=True)
