"""Drop-in for the hot-path part of StatisticalModel/Clustering.py: `Clustering.GMM`.

Same constructor, properties, `point`, `update_acc`, `update_param`, save/init of parameters and
accumulators (the reference's .npy directory layout).  Scoring and the E-step statistics
run on the GPU through libpoccala_hip.so; accumulators are kept in the reference's LOG domain on the
host so that a reference `multi_embedded_training_2` can consume the files this class writes.
Stand-alone EM / SMEM (Clustering.py:373-719 except update_acc/update_param) and the clustering
initialisers are out of scope.
"""
import configparser
import os
import time

import numpy as np

from ..Exceptions import DataDimensionError, NullLog
from .._lib import PCL_F32, PCL_F64
from ..runtime import scratch_engine
from .DataInitialization import DataInitialization
from .util import log_sum_exp, save_acc_file


def _lse2(a, b):
    """Elementwise util.log_sum_exp of two arrays (util.py:54-77): max + ln(sum exp(. - max)), the max itself where it is
    infinite (quirk Q4).  Same arithmetic as the reference's per-row calls, without the Python loop over mixtures."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    top = np.maximum(a, b)
    safe = np.where(np.isinf(top), 0.0, top)
    with np.errstate(all='ignore'):
        out = safe + np.log(np.exp(a - safe) + np.exp(b - safe))
    return np.where(np.isinf(top), top, out)


def _diag_of(covariance, m, d):
    """(M,D,D) full matrices of which only the diagonal is used (util.py:23), or (M,D) diagonals."""
    c = np.asarray(covariance, dtype=np.float64)
    if c.ndim == 3:
        return np.ascontiguousarray(np.diagonal(c, axis1=1, axis2=2))
    if c.shape == (m, d):
        return c
    raise ValueError('covariance must be (M,D,D) or (M,D), got %s' % (c.shape,))


class Clustering(DataInitialization):
    def __init__(self):
        super().__init__()

    class GMM(object):
        def __init__(self, log=None, dimension=1, mix_level=1, data=None, alpha=None, mean=None, variance=None,
                     covariance=None, differentiation=True, gmm_id=0, precision='f64'):
            self.log = log if log is not None else NullLog()
            self.__dimension = dimension
            self.__mix_level = mix_level
            self.__data = np.array(data) if data else None
            if mean is None:
                mean = np.random.random((mix_level, dimension)) if differentiation else np.zeros((mix_level, dimension))
            self.__mean = np.asarray(mean, dtype=np.float64)
            if covariance is not None:
                self.__covariance = covariance
            elif variance is not None:
                self.__covariance = np.array([np.diag(np.asarray(variance)[i]) for i in range(mix_level)])
            elif differentiation:
                self.__covariance = np.diag(np.random.random((dimension,))).reshape((1, dimension, dimension)).repeat(mix_level, axis=0)
            else:
                self.__covariance = np.eye(dimension).reshape((1, dimension, dimension)).repeat(mix_level, axis=0)
            self.__alpha = np.ones((mix_level,)) / mix_level if alpha is None else alpha
            self.__bias = 100.                                              # Clustering.py:103
            self.__gmm_id = gmm_id
            self.__precision = PCL_F64 if precision == 'f64' else PCL_F32
            self.__record_frames = 0
            # accumulators, log domain, initial -inf (Clustering.py:96-101)
            self.__alpha_acc = -np.inf
            self.__mean_acc = np.full((mix_level, dimension), -np.inf)
            self.__covariance_acc = list(np.full((mix_level, dimension), -np.inf))     # a list of per-mixture rows, as in the reference
            self.__acc = np.full((mix_level,), -np.inf)

        # ---------------------------------------------------------------- properties (Clustering.py:122-229)
        mean = property(lambda self: self.__mean, lambda self, v: setattr(self, '_GMM__mean', v))
        covariance = property(lambda self: self.__covariance, lambda self, v: setattr(self, '_GMM__covariance', v))
        alpha = property(lambda self: self.__alpha, lambda self, v: setattr(self, '_GMM__alpha', v))
        acc = property(lambda self: self.__acc, lambda self, v: setattr(self, '_GMM__acc', v))
        alpha_acc = property(lambda self: self.__alpha_acc, lambda self, v: setattr(self, '_GMM__alpha_acc', v))
        mean_acc = property(lambda self: self.__mean_acc, lambda self, v: setattr(self, '_GMM__mean_acc', v))
        bias = property(lambda self: self.__bias, lambda self, v: setattr(self, '_GMM__bias', v))

        @property
        def covariance_acc(self):
            """The reference's getter returns the bias (Clustering.py:206-209); kept."""
            return self.__bias

        @covariance_acc.setter
        def covariance_acc(self, v):
            self.__covariance_acc = v

        @property
        def dimension(self):
            return self.__dimension

        @property
        def mixture(self):
            return self.__mix_level

        @property
        def gmm_id(self):
            return self.__gmm_id

        @property
        def data(self):
            return self.__data

        @data.setter
        def data(self, data):
            """Clustering.py:166-174 (the per-sample gamma it also resets belongs to the stand-alone EM, out of scope)."""
            self.__data = np.array(data)

        def add_data(self, data):
            """Clustering.py:106-116.  The reference tests `if self.__data:`, which raises for an array of more than one row, so only
            the first call on an empty object works there; here further calls append, which is what the line below it was written to do."""
            if self.__data is not None and len(self.__data):
                self.__data = np.append(self.__data, np.array(data), axis=0)
            else:
                self.__data = np.array(data)

        def clear_data(self):
            """Clustering.py:118-120."""
            self.__data = None

        def diag_variance(self):
            return _diag_of(self.__covariance, self.__mix_level, self.__dimension)

        def model_arrays(self):
            """(mean (M,D), var (M,D), weight (M,)) as the engine takes them."""
            return (np.asarray(self.__mean, np.float64), self.diag_variance(), np.asarray(self.__alpha, np.float64))

        # ---------------------------------------------------------------- A4  point (Clustering.py:740-767)
        def point(self, x, log=False, standard=False, record=False):
            x = np.asarray(x, dtype=np.float64).reshape(-1)
            if len(x) != self.dimension:
                raise DataDimensionError(self.dimension, len(x), self.log)
            if not log or standard:
                raise NotImplementedError('only log=True, standard=False is on the hot path (SURVEY quirk Q3)')
            out = self.point_frames(x[None, :])[0]
            if record:
                self.__record_frames += 1      # the per-mixture record is recomputed on the device in update_acc
            return out

        def point_frames(self, frames):
            """ln b(o_t) for a (T,D) block of frames: one scoring launch."""
            frames = np.asarray(frames, dtype=np.float64)
            if frames.ndim != 2 or frames.shape[1] != self.dimension:
                raise DataDimensionError(self.dimension, frames.shape[-1], self.log)
            eng = scratch_engine()
            mean, var, w = self.model_arrays()
            eng.load_model(mean[None], var[None], w[None])
            eng.load_frames(frames)
            t = frames.shape[0]
            b = eng.batch([3], [t], [0])
            b.set_states([np.array([-1, 0, -2], dtype=np.int32)])
            b.score(self.__precision)
            out = b.get('B')[0][1].copy()
            b.close()
            return out

        # ---------------------------------------------------------------- A13  update_acc (Clustering.py:653-680)
        def update_acc(self, l_value, b_value, o_value):
            l_value = np.asarray(l_value, dtype=np.float64)
            b_value = np.asarray(b_value, dtype=np.float64)
            o_value = np.asarray(o_value, dtype=np.float64)
            t = o_value.shape[0]
            eng = scratch_engine()
            mean, var, w = self.model_arrays()
            eng.load_model(mean[None], var[None], w[None])
            eng.load_frames(o_value)
            b = eng.batch([3], [t], [0])
            b.set_states([np.array([-1, 0, -2], dtype=np.int32)])
            ninf = np.full(t, -np.inf)
            b.set_emissions([np.stack([np.zeros(t), b_value, ninf])])
            b.set_posteriors([np.stack([ninf, l_value, ninf])])
            eng.stats_zero()
            b.accumulate(self.__precision)
            st = eng.stats_download()
            b.close()
            with np.errstate(divide='ignore'):
                self.__acc = _lse2(np.log(st['acc'][0]), self.__acc)
                # alpha_acc: the reference adds ln gamma_t(j) of EVERY frame (Clustering.py:667); the device
                # sums only frames that do not underflow, identical to the last bit of the float64 sum
                self.__alpha_acc = log_sum_exp(np.append(l_value, self.__alpha_acc))
                self.__mean_acc = _lse2(np.log(st['mean_acc'][0]), self.__mean_acc)
                self.__covariance_acc = list(_lse2(np.log(st['cov_acc'][0]), np.asarray(self.__covariance_acc)))
            self.__record_frames = 0

        # ---------------------------------------------------------------- A15  update_param (Clustering.py:682-693)
        def update_param(self, show_q=False, c_covariance=1e-3):
            self.log.note('training GMM_%d by Baum-Welch, %d mixtures' % (self.__gmm_id, self.__mix_level), cls='i',
                          show_console=show_q)
            with np.errstate(all='ignore'):
                self.__alpha = np.exp(self.__acc - self.__alpha_acc)
                self.__mean = np.exp(self.__mean_acc - self.__acc.reshape(-1, 1)) - self.__bias
                cov = np.array(self.__covariance, dtype=np.float64, copy=True)
                if cov.ndim != 3:
                    cov = np.array([np.diag(r) for r in cov])
                for i in range(self.__mix_level):
                    c = np.exp(self.__covariance_acc[i] - self.__acc[i])
                    if (c < c_covariance).any():
                        self.log.note('variance below the floor, corrected to %s' % c_covariance, cls='w')
                        c[c < c_covariance] = c_covariance
                    cov[i] = np.diag(c)
            self.__covariance = cov

        # ---------------------------------------------------------------- T3  files (Clustering.py:234-367)
        def _dir(self, path):
            return path + '/GMM_%d' % self.__gmm_id

        def save_parameter(self, path):
            p = self._dir(path)
            os.makedirs(p, exist_ok=True)
            np.save(p + '/GMM_means.npy', self.__mean)
            cov = np.asarray(self.__covariance)
            if cov.ndim == 2:                       # always write (M,D,D) for reference compatibility
                cov = np.array([np.diag(r) for r in cov])
            np.save(p + '/GMM_covariance.npy', cov)
            np.save(p + '/GMM_weight.npy', self.__alpha)
            cfg = configparser.ConfigParser()
            cfg.add_section('Configuration')
            cfg.set('Configuration', 'MIXTURE', str(self.__mix_level))
            cfg.set('Configuration', 'DIMENSION', str(self.__dimension))
            cfg.set('Configuration', 'BIAS', str(self.__bias))
            with open(p + '/GMM_config.ini', 'w+') as f:
                cfg.write(f)

        def init_parameter(self, path):
            p = self._dir(path)
            if not os.path.exists(p):
                raise FileNotFoundError('model parameter directory %s does not exist' % p)
            self.__mean = np.load(p + '/GMM_means.npy')
            self.__covariance = np.load(p + '/GMM_covariance.npy')      # (M,D,D) or (M,D) accepted
            self.__alpha = np.load(p + '/GMM_weight.npy')
            self.__mix_level, self.__dimension = self.__mean.shape
            # the reference never reads the .ini back (it passes a file object to ConfigParser.read,
            # Clustering.py:304-312); the arrays define mixture and dimension

        def save_acc(self, path):
            p = self._dir(path)
            stamp = int(time.time())
            for sub, name, val in (('acc', 'GMM_acc', self.__acc), ('alpha-acc', 'GMM_alpha_acc', self.__alpha_acc),
                                   ('mean-acc', 'GMM_mean_acc', self.__mean_acc),
                                   ('covariance-acc', 'GMM_covariance_acc', np.asarray(self.__covariance_acc))):
                save_acc_file(p + '/' + sub, name, stamp, val)

        def init_acc(self, path):
            """Merge every accumulator file under the unit directory (Clustering.py:314-367)."""
            p = self._dir(path)

            def files(sub):
                d = p + '/' + sub
                return [np.load(os.path.join(d, f)) for f in sorted(os.listdir(d))] if os.path.isdir(d) else []
            with np.errstate(all='ignore'):
                for a in files('acc'):
                    self.__acc = log_sum_exp(np.stack([self.__acc, a], axis=1), vector=True)
                for a in files('alpha-acc'):
                    self.__alpha_acc = log_sum_exp(np.array([self.__alpha_acc, float(a)]))
                for a in files('mean-acc'):
                    for i in range(self.__mix_level):
                        self.__mean_acc[i] = log_sum_exp(np.stack([self.__mean_acc[i], a[i]], axis=1), vector=True)
                for a in files('covariance-acc'):
                    for i in range(self.__mix_level):
                        self.__covariance_acc[i] = log_sum_exp(np.stack([self.__covariance_acc[i], a[i]], axis=1), vector=True)

        # ---------------------------------------------------------------- out of scope
        def em(self, *a, **k):
            raise NotImplementedError('stand-alone GMM EM / SMEM is outside the hot path (SURVEY section 2 row 3)')
