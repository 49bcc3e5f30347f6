"""Process-wide Engines used by the drop-in classes (one GPU per process, as the reference runs one worker process per
utterance -- AcousticModel/AcousticModel.py:708-714,861-870).

Two contexts on the one device:
  default_engine()   the batched entry points (AcousticModel.estep_batch / align_batch / ..., bench.py): holds the corpus shard,
                     the whole model and the E-step statistics of the batch in flight;
  scratch_engine()   the per-object calls the reference's own call stacks make (Clustering.GMM.point / update_acc per state,
                     LHMM.cal_observation_pro per unit, LHMM.baulm_welch / viterbi per utterance): they upload a handful of
                     states and frames each time and zero / download their OWN statistics, so they can be interleaved with
                     a batched E-step without touching its model, frames or statistics.
Uploads are skipped when the same content is already resident (Engine.load_model / load_frames compare a digest), which
is what the zero-change route does all the time: `point` is called once per frame with the same GMM.
"""
import os

_engine = None
_scratch = None


def _device():
    return int(os.environ.get('POCCALA_DEVICE', os.environ.get('LOCAL_RANK', '0')))


def default_engine():
    """Engine on device $POCCALA_DEVICE (default: $LOCAL_RANK, else 0).  Raises without a GPU."""
    global _engine
    if _engine is None:
        from .engine import Engine
        _engine = Engine(_device())
    return _engine


def scratch_engine():
    """The private context of the per-object drop-in calls (same device)."""
    global _scratch
    if _scratch is None:
        from .engine import Engine
        _scratch = Engine(_device())
    return _scratch


def set_default_engine(engine):
    global _engine
    _engine = engine
