"""Host side of the multi-GPU path: one process per GPU, utterances sharded across ranks.

The data path needs no collective for scoring / forward-backward / Viterbi (utterances are
independent, AcousticModel/AcousticModel.py:865-870).  The E-step has ONE exchange, inside
libpoccala_hip.so over RCCL (pcl_em_exchange: the GMM statistics by state range, the per-unit HMM
accumulators -- un-normalised LOG values, SURVEY quirk Q5 -- by a max-then-sum all-reduce, the
log-sum-exp the reference's file reducer computes, StatisticalModel/LHMM.py:272-290).  This module
is the host side: the utterance split and a small TCP control plane.  No PyTorch here (north star);
the torch.distributed (gloo) helpers the CPU world-2 tests use live in tests/_dist_torch.py.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous, balanced split of utterance indices: rank r owns [lo, hi)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------------
# Control plane without torch: a star over TCP (rank 0 = hub) for the few tiny host-side exchanges a
# GPU rank needs (barrier, max of a float, broadcast of the 128-byte RCCL id, per-rank timings).  GPU
# processes therefore never import torch: torch's wheel bundles its own libamdhip64 / librccl (same
# sonames as /opt/rocm's) and two HIP runtimes in one process crash at exit.  Rendezvous: MASTER_ADDR
# and MASTER_PORT + 1 + k from the launcher's env (MASTER_PORT itself belongs to the launcher's store).
#
# Wire format: fixed binary frames, never pickle -- a frame is  magic(4) kind(1) length(8)  + payload,
# kind in {none, bytes, float64/int64/int32 ndarray (dtype code, ndim, shape, raw data), float, JSON
# text}; nothing received is ever executed or unpickled.  Every connection is authenticated both ways
# with HMAC-SHA256 over a fresh challenge, keyed by a token the launcher hands to all ranks
# (POCCALA_CTRL_TOKEN; bench.py's self-spawn makes a random one per job); rank ids are range-checked
# and duplicates refused.  The hub binds MASTER_ADDR only (127.0.0.1 on one node).
# ---------------------------------------------------------------------------------------------------
import hashlib
import hmac
import json
import os
import socket
import struct
import time

_MAGIC = b'PCL2'
_K_NONE, _K_BYTES, _K_ARRAY, _K_FLOAT, _K_JSON = 0, 1, 2, 3, 4
_DTYPES = {0: np.dtype('<f8'), 1: np.dtype('<i8'), 2: np.dtype('<i4')}
_DCODE = {v: k for k, v in _DTYPES.items()}
_MAX_FRAME = 1 << 30


def _loopback(addr):
    """Does MASTER_ADDR name this machine's loopback interface?  IPv4 literals are parsed (a hostname such as 127.example.com is
    not one); a hostname counts when every address it resolves to is a loopback address -- which covers 'localhost' and, on
    the usual single-node images, the node's own hostname (/etc/hosts maps it to 127.0.1.1).  The hub socket is AF_INET, so an
    IPv6 literal is never accepted."""
    import ipaddress
    try:
        ip = ipaddress.ip_address(addr)
        return ip.version == 4 and ip.is_loopback
    except ValueError:
        pass
    try:
        infos = socket.getaddrinfo(addr, None, socket.AF_INET)
    except OSError:
        return False
    return bool(infos) and all(ipaddress.ip_address(i[4][0]).is_loopback for i in infos)


def _token(addr='127.0.0.1'):
    """The HMAC key of the control plane.  POCCALA_CTRL_TOKEN when the launcher set one (bench.py's self-spawn makes a random
    one per job).  Without it -- a foreign launcher such as torch.distributed.run -- the only thing the ranks share is the run
    id and port, which anyone can guess: that fallback is accepted on a loopback address only (one node, the hub is not
    reachable from outside); a hub on a routable address refuses to start without an explicit token."""
    t = os.environ.get('POCCALA_CTRL_TOKEN')
    if t is None:
        if not _loopback(addr):
            raise RuntimeError('control plane: MASTER_ADDR=%s does not resolve to a loopback address and POCCALA_CTRL_TOKEN is not set; '
                               'export the same random POCCALA_CTRL_TOKEN on every rank of a multi-node job (a single-node job whose '
                               'MASTER_ADDR is the node\'s routable hostname needs it too, or MASTER_ADDR=127.0.0.1)' % addr)
        t = 'run:%s:%s' % (os.environ.get('TORCHELASTIC_RUN_ID', ''), os.environ.get('MASTER_PORT', ''))
    return t.encode()


def _exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError('control plane: peer closed')
        buf += chunk
    return bytes(buf)


def _encode(obj):
    if obj is None:
        return _K_NONE, b''
    if isinstance(obj, (bytes, bytearray, memoryview)):
        return _K_BYTES, bytes(obj)
    if isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        if a.dtype not in _DCODE:
            a = a.astype(np.float64)
        return _K_ARRAY, struct.pack('<BB', _DCODE[a.dtype], a.ndim) + struct.pack('<%dq' % a.ndim, *a.shape) + a.tobytes()
    if isinstance(obj, (float, np.floating)):
        return _K_FLOAT, struct.pack('<d', float(obj))
    return _K_JSON, json.dumps(obj).encode()            # small dict / list / int / str control data


def _decode(kind, data):
    if kind == _K_NONE:
        return None
    if kind == _K_BYTES:
        return data
    if kind == _K_ARRAY:
        code, ndim = struct.unpack_from('<BB', data, 0)
        if code not in _DTYPES or ndim > 8:
            raise ValueError('control plane: bad array header')
        shape = struct.unpack_from('<%dq' % ndim, data, 2)
        dt = _DTYPES[code]
        off = 2 + 8 * ndim
        if any(d < 0 for d in shape) or int(np.prod(shape, dtype=np.int64)) * dt.itemsize != len(data) - off:
            raise ValueError('control plane: array size mismatch')
        return np.frombuffer(data, dtype=dt, offset=off).reshape(shape).copy()
    if kind == _K_FLOAT:
        return struct.unpack('<d', data)[0]
    if kind == _K_JSON:
        return json.loads(data.decode())
    raise ValueError('control plane: unknown frame kind %d' % kind)


def _send(sock, obj):
    kind, data = _encode(obj)
    sock.sendall(_MAGIC + struct.pack('<BQ', kind, len(data)) + data)


def _recv(sock):
    hdr = _exact(sock, 13)
    if hdr[:4] != _MAGIC:
        raise ValueError('control plane: bad magic')
    kind, n = struct.unpack('<BQ', hdr[4:])
    if n > _MAX_FRAME:
        raise ValueError('control plane: frame too large')
    return _decode(kind, _exact(sock, n))


def _mac(token, *parts):
    return hmac.new(token, b'|'.join(parts), hashlib.sha256).digest()


class Control(object):
    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=300.0, token=None):
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else rank
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else world
        self.peers = []
        self.hub = None
        if self.world == 1:
            return
        if not 0 <= self.rank < self.world:
            raise ValueError('control plane: rank %d outside [0,%d)' % (self.rank, self.world))
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        token = _token(addr) if token is None else token
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
                try:
                    c.settimeout(3.0)            # (the accept loop is serial: a silent stranger may hold it this long, no longer)
                    # challenge -> (rank, HMAC(challenge | rank | client nonce)) -> HMAC(client nonce | 'hub')
                    challenge = os.urandom(16)
                    c.sendall(_MAGIC + challenge)
                    reply = _exact(c, 4 + 4 + 16 + 32)
                    if reply[:4] != _MAGIC:
                        raise ValueError('bad magic')
                    (r,) = struct.unpack('<I', reply[4:8])
                    nonce, mac = reply[8:24], reply[24:56]
                    if not hmac.compare_digest(mac, _mac(token, challenge, struct.pack('<I', r), nonce)):
                        raise ValueError('authentication failed')
                    if not 0 < r < self.world or slots[r] is not None:
                        raise ValueError('rank %d out of range or already connected' % r)
                    c.sendall(_mac(token, nonce, b'hub'))
                    c.settimeout(timeout)
                    slots[r] = c
                except (OSError, ValueError, ConnectionError, struct.error):
                    c.close()                    # a stranger, a wrong token or a duplicate: drop it, keep listening
            srv.close()
            self.peers = slots
        else:
            deadline = time.time() + timeout
            while self.hub is None:
                for k in range(32):
                    s = None
                    try:
                        s = socket.create_connection((addr, base + k), timeout=2.0)
                        s.settimeout(5.0)                   # a foreign service on this port must not stall us
                        hello = _exact(s, 20)
                        if hello[:4] != _MAGIC:
                            raise ValueError('not the hub')
                        nonce = os.urandom(16)
                        rb = struct.pack('<I', self.rank)
                        s.sendall(_MAGIC + rb + nonce + _mac(token, hello[4:], rb, nonce))
                        if not hmac.compare_digest(_exact(s, 32), _mac(token, nonce, b'hub')):
                            raise ValueError('the hub failed to authenticate')
                        s.settimeout(timeout)
                        self.hub = s
                        break
                    except (OSError, ConnectionError, struct.error, ValueError):
                        if s is not None:
                            s.close()
                        continue
                if self.hub is None:
                    if time.time() > deadline:
                        raise RuntimeError('control plane: cannot reach rank 0 at %s:%d+' % (addr, base))
                    time.sleep(0.2)

    def allgather(self, obj):
        """List of every rank's object, in rank order, on every rank.  Objects: None, bytes, float, float64 / int
        ndarrays, or JSON-serialisable control data."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            items = [obj] + [_recv(self.peers[r]) for r in range(1, self.world)]
            for r in range(1, self.world):
                for it in items:
                    _send(self.peers[r], it)
            return items
        _send(self.hub, obj)
        return [_recv(self.hub) for _ in range(self.world)]

    def allgather_bytes(self, data):
        """Every rank's byte string in rank order (the transport of Engine.comm_init_host)."""
        return [bytes(x) for x in self.allgather(bytes(data))]

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
