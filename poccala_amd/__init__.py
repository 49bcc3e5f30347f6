"""poccala_amd -- MI355X (gfx950) engine for the GMM-HMM hot path of Byshx/Poccala.

Layout
  csrc/                 hand-written HIP kernels + the C-ABI (include/poccala_hip.h)
  _lib.py               ctypes binding (no CPU fallback: fails loudly without the library / a GPU)
  engine.py             batched host API: Engine, Batch, SentenceBatch
  StatisticalModel/     drop-in classes mirroring the reference's LHMM / Clustering.GMM / util
  AcousticModel/        drop-in AcousticModel helpers (embedded, viterbi, discriminate, VirtualState)
"""
from ._lib import PCL_F32, PCL_F64, PoccalaHipError  # noqa: F401
from .engine import Engine, Batch  # noqa: F401

__all__ = ['Engine', 'Batch', 'PCL_F32', 'PCL_F64', 'PoccalaHipError']
