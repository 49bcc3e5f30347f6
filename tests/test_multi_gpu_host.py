"""world_size-2 gloo tests (CPU) of the host side of the multi-GPU path: sharding, the log-domain
all-reduce of the HMM accumulators, and that sharded E-step statistics sum to the unsharded ones.
The per-rank compute here is the oracle (test infrastructure); on the GPU box the same host code
drives libpoccala_hip.so and RCCL."""
import os
import socket

import numpy as np
import pytest
import multiprocessing as mp     # torch is imported only inside the spawned gloo workers, never in the pytest process

from poccala_amd.distributed import shard_range


def test_shard_range_partitions_everything():
    for n in (1, 7, 8, 1024, 8191):
        for world in (1, 2, 3, 8):
            got = [shard_range(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(got[r][1] == got[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in got]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import poccala_oracle as po
    from poccala_amd import synth
    from _dist_torch import allreduce_logsumexp, allreduce_sum_host
    from poccala_amd.distributed import shard_range
    units, M, D, U, T, L = 3, 4, 5, 6, 30, 2
    mean, var, w, trans = synth.make_model(units, M, D, seed=5)
    frames, lens, begin = synth.make_frames(U, T, D, seed=6)
    labels = synth.make_labels(U, L, units, seed=7)
    model = {u: dict(trans=trans[u], gmms=[(mean[u * 3 + k], var[u * 3 + k], w[u * 3 + k]) for k in range(3)]) for u in range(units)}

    def stats_for(utts):
        acc = np.zeros((units * 3, M))
        ks = np.full((units, 3, 5), -np.inf)
        for u in utts:
            x = frames[begin[u]:begin[u] + lens[u]].astype(np.float64)
            bw, accs, _ = po.estep_utterance(x, list(labels[u]), model)
            for pos, unit in enumerate(labels[u]):
                ks[unit] = np.logaddexp(ks[unit], accs[pos].ksai_acc)
                for k in range(3):
                    acc[unit * 3 + k] += np.exp(accs[pos].gmm[k]['acc'])
        return acc, ks
    lo, hi = shard_range(U, rank, world)
    acc, ks = stats_for(range(lo, hi))
    acc_all = allreduce_sum_host(acc, dist)          # stands in for pcl_stats_allreduce (RCCL) on the GPU box
    ks_all = allreduce_logsumexp(ks, dist)
    if rank == 0:
        ref_acc, ref_ks = stats_for(range(U))
        q.put((np.allclose(acc_all, ref_acc, rtol=1e-12), np.array_equal(np.isneginf(ks_all), np.isneginf(ref_ks)),
               np.allclose(ks_all[np.isfinite(ref_ks)], ref_ks[np.isfinite(ref_ks)], rtol=1e-12)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_estep_statistics_sum_to_unsharded_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) == (True, True, True)


def _ctl_worker(rank, world, port, q):
    from poccala_amd.distributed import Control
    ctl = Control(rank, world, addr='127.0.0.1', port=port)
    ctl.barrier()
    got = ctl.allgather({'r': rank})
    uid = ctl.broadcast(b'x' * 128 if rank == 0 else None, src=0)
    mx = ctl.allreduce_max(1.5 + rank)
    a = np.array([[-np.inf, -3.0 - rank], [2.0 * rank, -np.inf if rank else 0.5]])
    lse = ctl.allreduce_logsumexp(a)
    sm = ctl.allreduce_sum(np.ones(3) * (rank + 1))
    ctl.barrier()
    ctl.close()
    if rank == 0:
        q.put((got, uid, mx, lse, sm))


def test_tcp_control_plane_world2():
    """The torch-free control plane bench.py uses between GPU ranks."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ctl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, uid, mx, lse, sm = q.get(timeout=60)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [{'r': 0}, {'r': 1}] and uid == b'x' * 128 and mx == 2.5
    ref = np.array([[-np.inf, np.logaddexp(-3.0, -4.0)], [np.logaddexp(0.0, 2.0), 0.5]])
    assert np.isneginf(lse[0, 0])
    np.testing.assert_allclose(lse[np.isfinite(ref)], ref[np.isfinite(ref)], rtol=1e-14)
    np.testing.assert_allclose(sm, 3.0)


def _auth_worker(kind, port, q):
    from poccala_amd.distributed import Control
    if kind == 'hub':
        ctl = Control(0, 2, addr='127.0.0.1', port=port, token=b'job-secret', timeout=60)
        got = ctl.allgather(np.arange(3, dtype=np.float64))
        ctl.close()
        q.put(('hub', [g.tolist() for g in got]))
    elif kind == 'good':
        import time
        time.sleep(1.5)                       # let the impostors try first
        ctl = Control(1, 2, addr='127.0.0.1', port=port, token=b'job-secret', timeout=60)
        ctl.allgather(np.arange(3, dtype=np.float64) + 10)
        ctl.close()
    else:                                     # wrong token, or a rank id outside the job
        try:
            Control(1 if kind == 'bad-token' else 7, 2 if kind == 'bad-token' else 8, addr='127.0.0.1', port=port,
                    token=b'wrong' if kind == 'bad-token' else b'job-secret', timeout=3)
            q.put((kind, 'connected'))
        except (RuntimeError, ValueError):
            q.put((kind, 'refused'))


def test_control_plane_refuses_wrong_token_and_bad_rank():
    """the hub must authenticate its peers, range-check rank ids and never unpickle what it receives."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    kinds = ['hub', 'bad-token', 'bad-rank', 'good']
    procs = [ctx.Process(target=_auth_worker, args=(k, port, q)) for k in kinds]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=60) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res['bad-token'] == 'refused' and res['bad-rank'] == 'refused'
    assert res['hub'] == [[0.0, 1.0, 2.0], [10.0, 11.0, 12.0]]


def test_control_plane_frames_are_not_pickle():
    import poccala_amd.distributed as d
    src = open(d.__file__).read()
    assert 'import pickle' not in src and 'pickle.loads' not in src
    for obj in (None, b'\x00' * 128, 2.5, np.array([[1.0, -np.inf]]), np.arange(4, dtype=np.int32), {'a': [1, 2]}):
        kind, data = d._encode(obj)
        back = d._decode(kind, data)
        if isinstance(obj, np.ndarray):
            assert back.dtype == obj.dtype and np.array_equal(back, obj)
        else:
            assert back == obj


def test_control_plane_refuses_a_guessable_key_off_loopback(monkeypatch):
    """without POCCALA_CTRL_TOKEN the HMAC key falls back to the run id and port, which anyone can guess; that is
    accepted on a loopback address only."""
    from poccala_amd import distributed as dist
    monkeypatch.delenv('POCCALA_CTRL_TOKEN', raising=False)
    assert dist._token('127.0.0.1') and dist._token('localhost')
    with pytest.raises(RuntimeError):
        dist._token('10.1.2.3')
    with pytest.raises(RuntimeError):
        dist.Control(rank=1, world=2, addr='10.1.2.3', port=1, timeout=0.1)
    monkeypatch.setenv('POCCALA_CTRL_TOKEN', 'secret')
    assert dist._token('10.1.2.3') == b'secret'
