"""torch.distributed (gloo) helpers for the CPU world-2 tests of the N > 1 host logic (tests/test_multi_gpu_host.py).
Test infrastructure: the product package imports no PyTorch (north star); GPU ranks use poccala_amd.distributed.Control and RCCL
inside libpoccala_hip.so."""
import numpy as np


def broadcast_unique_id(engine, dist, rank):
    """rank 0 creates the 128-byte ncclUniqueId, everyone receives it."""
    box = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def allreduce_logsumexp(arr, dist):
    """Elementwise log-sum-exp of a float64 array over all ranks (util.log_sum_exp semantics: an
    element that is -inf everywhere stays -inf)."""
    import torch
    a = np.ascontiguousarray(arr, dtype=np.float64)
    top = torch.from_numpy(a.copy())
    dist.all_reduce(top, op=dist.ReduceOp.MAX)
    top = top.numpy()
    safe = np.where(np.isinf(top), 0.0, top)
    with np.errstate(all='ignore'):
        s = torch.from_numpy(np.exp(a - safe))
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    with np.errstate(all='ignore'):
        out = safe + np.log(s.numpy())
    return np.where(np.isinf(top), top, out)


def allreduce_sum_host(arr, dist):
    """Sum of a host array over ranks (control-plane sized data only; GPU statistics use RCCL)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()
