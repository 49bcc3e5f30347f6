"""The full-size synthetic models of the GPU suite, made once per session.

synth.make_model for the C4 shard draws 2 x 240 M numbers: 5-10 s on a GPU box's host, and six full-size tests asked for the same model
(three more for the C5 shard's) -- a sixth of the suite's time.  The arrays are handed out read-only: a test that wants to edit a model
copies it."""
import functools


@functools.lru_cache(maxsize=2)
def _made(units, M, D, seed):
    from poccala_amd import synth
    mean, var, w, trans = synth.make_model(units, M, D, seed=seed)
    for a in (mean, var, w):
        a.setflags(write=False)
    return mean, var, w, tuple(trans)


def full_size_model(cfg, seed):
    """(mean, var, w, trans) of synth.make_model(cfg['units'], cfg['M'], cfg['D'], seed=seed), cached"""
    mean, var, w, trans = _made(cfg['units'], cfg['M'], cfg['D'], seed)
    return mean, var, w, list(trans)
