"""Host side of the multi-GPU path: one process per GPU, utterances sharded across ranks.

The data path needs no collective for scoring / forward-backward / Viterbi (utterances are
independent, AcousticModel/AcousticModel.py:865-870).  The E-step has ONE exchange: the GMM
statistics are summed by RCCL inside libpoccala_hip.so (pcl_stats_allreduce); the tiny per-unit HMM
accumulators are un-normalised LOG values and are merged here with a
max-then-sum all-reduce, which is the log-sum-exp the reference's file reducer computes
(StatisticalModel/LHMM.py:272-290).  `dist` is torch.distributed (gloo for control traffic).
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous, balanced split of utterance indices: rank r owns [lo, hi)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


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


# ---------------------------------------------------------------------------------------------------
# Control plane without torch: a star over TCP (rank 0 = hub) for the few tiny host-side exchanges a
# GPU rank needs (barrier, max of a float, broadcast of the 128-byte RCCL id, log-domain merge of the
# per-unit HMM accumulators).  GPU processes therefore never import torch: torch's wheel bundles its
# own libamdhip64 / librccl (same sonames as /opt/rocm's) and two HIP runtimes in one process crash
# at exit.  Rendezvous: MASTER_ADDR and MASTER_PORT + 1 + k from the torch.distributed.run env
# (MASTER_PORT itself belongs to the launcher's store).
# ---------------------------------------------------------------------------------------------------
import os
import pickle
import socket
import struct
import time

_MAGIC = b'PCLCTRL1'


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack('!Q', len(data)) + data)


def _recv(sock):
    def exact(n):
        buf = b''
        while len(buf) < n:
            chunk = sock.recv(n - len(buf))
            if not chunk:
                raise ConnectionError('control plane: peer closed')
            buf += chunk
        return buf
    (n,) = struct.unpack('!Q', exact(8))
    return pickle.loads(exact(n))


class Control(object):
    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=300.0):
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else rank
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else world
        self.peers = []
        self.hub = None
        if self.world == 1:
            return
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        base = int(port if port is not None else int(os.environ.get('MASTER_PORT', '29500')) + 1)
        if self.rank == 0:
            srv = None
            for k in range(32):
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, base + k))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError('control plane: no free port in [%d,%d)' % (base, base + 32))
            srv.listen(self.world)
            srv.settimeout(timeout)
            slots = [None] * self.world
            while sum(s is not None for s in slots[1:]) < self.world - 1:
                c, _ = srv.accept()
                c.settimeout(timeout)
                hello = _recv(c)
                if not (isinstance(hello, tuple) and hello[0] == _MAGIC):
                    c.close()
                    continue
                slots[hello[1]] = c
                _send(c, _MAGIC)
            srv.close()
            self.peers = slots
        else:
            deadline = time.time() + timeout
            while self.hub is None:
                for k in range(32):
                    try:
                        s = socket.create_connection((addr, base + k), timeout=2.0)
                        s.settimeout(5.0)                   # a foreign service on this port must not stall us
                        _send(s, (_MAGIC, self.rank))
                        if _recv(s) == _MAGIC:
                            s.settimeout(timeout)
                            self.hub = s
                            break
                        s.close()
                    except (OSError, ConnectionError, pickle.UnpicklingError, struct.error, EOFError, ValueError):
                        continue
                if self.hub is None:
                    if time.time() > deadline:
                        raise RuntimeError('control plane: cannot reach rank 0 at %s:%d+' % (addr, base))
                    time.sleep(0.2)

    def allgather(self, obj):
        """List of every rank's object, in rank order, on every rank."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            items = [obj] + [_recv(self.peers[r]) for r in range(1, self.world)]
            for r in range(1, self.world):
                _send(self.peers[r], items)
            return items
        _send(self.hub, obj)
        return _recv(self.hub)

    def barrier(self):
        self.allgather(None)

    def broadcast(self, obj, src=0):
        return self.allgather(obj if self.rank == src else None)[src]

    def allreduce_max(self, x):
        return max(self.allgather(float(x)))

    def allreduce_sum(self, arr):
        return np.sum(np.stack(self.allgather(np.asarray(arr, dtype=np.float64))), axis=0)

    def allreduce_logsumexp(self, arr):
        stack = np.stack(self.allgather(np.asarray(arr, dtype=np.float64)))
        with np.errstate(all='ignore'):
            top = stack.max(axis=0)
            safe = np.where(np.isinf(top), 0.0, top)
            out = safe + np.log(np.exp(stack - safe).sum(axis=0))
        return np.where(np.isinf(top), top, out)

    def close(self):
        for s in [self.hub] + list(self.peers):
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self.hub, self.peers = None, []
