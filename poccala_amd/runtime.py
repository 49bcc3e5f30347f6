"""Process-wide default Engine used by the drop-in classes (one GPU per process, as the reference
runs one worker process per utterance -- AcousticModel/AcousticModel.py:708-714,861-870)."""
import os

_engine = None


def default_engine():
    """Engine on device $POCCALA_DEVICE (default: $LOCAL_RANK, else 0).  Raises without a GPU."""
    global _engine
    if _engine is None:
        from .engine import Engine
        dev = int(os.environ.get('POCCALA_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        _engine = Engine(dev)
    return _engine


def set_default_engine(engine):
    global _engine
    _engine = engine
