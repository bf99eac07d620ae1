"""
Multi-GPU candidate sweep: the candidate array is cut into contiguous row blocks,
one per rank (one process per GPU); every rank holds the whole fitted GP
(refitting redundantly is cheaper than broadcasting L^-1 at these N) and sweeps
its own block with no data-path communication.  One exchange per sweep then
agrees on the arg-max: each rank writes (value, global index, valid, payload)
into its own slot of a zero buffer, a single all-reduce(sum) gathers the slots
(RCCL has no MAXLOC), and every rank applies the same deterministic reduction:
largest value, ties to the lowest global index (numpy.argmax order).

The transport is any object with .world_size, .rank, .argmax(val, idx, payload) and
.allreduce_sum(buf): `RcclArgmax` is libibo_hip's RCCL path over xGMI (csrc/comm.hip);
`SocketComm` carries the same slot protocol over a Unix-domain socket on one host -- for
worlds RCCL cannot form (several ranks on ONE device: RCCL refuses two ranks per GPU) and
for debugging, never the fast path; the CPU tests drive the identical protocol
(fill_slot / reduce_slots) over gloo with a transport of their own (tests/gloo_transport.py).
"""
import ctypes

import numpy as np

from . import _lib


def shard_bounds(M, world_size, rank):
    """contiguous block [start, stop) of rank `rank` among `world_size`;
    the first M % world_size ranks take one extra row"""
    base, rem = divmod(int(M), int(world_size))
    start = rank * base + min(rank, rem)
    stop = start + base + (1 if rank < rem else 0)
    return start, stop


def reduce_slots(buf, world_size, npayload):
    """final reduction over the gathered slot buffer -> (val, idx, payload, rank);
    idx = -1 when no rank had an admissible candidate.  Mirrors slot_argmax in csrc/comm.hip."""
    slot = 3 + npayload
    buf = np.asarray(buf, dtype=float).reshape(world_size, slot)
    best = None
    for r in range(world_size):
        v, i, ok = buf[r, 0], int(buf[r, 1]), buf[r, 2]
        if ok == 0.0:
            continue
        if best is None or v > best[0] or (v == best[0] and i < best[1]):
            best = (v, i, buf[r, 3:].copy(), r)
    if best is None:
        return 0.0, -1, np.zeros(npayload), -1
    return best


def fill_slot(world_size, rank, val, idx, payload):
    npayload = len(payload)
    slot = 3 + npayload
    buf = np.zeros(world_size * slot)
    if idx is not None and idx >= 0 and val == val:
        buf[rank * slot:rank * slot + 3] = (val, float(idx), 1.0)
        buf[rank * slot + 3:(rank + 1) * slot] = payload
    return buf


class RcclArgmax(object):
    """slot protocol over RCCL/xGMI through libibo_hip (one communicator per process)"""
    device_exchange = True        # sharded_sweep may hand the whole step to ibo_acq_sweep_exchange

    def __init__(self, world_size, rank, unique_id, device=None):
        self.world_size, self.rank = world_size, rank
        self.device = _lib.default_device() if device is None else device
        h = ctypes.c_void_p()
        _lib.check(_lib.lib.ibo_comm_init(self.device, world_size, rank, unique_id, ctypes.byref(h)))
        self.h = h

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(_lib.lib.ibo_comm_get_unique_id(buf))
        return buf.raw

    def argmax(self, val, idx, payload=()):
        payload = _lib.f64(np.asarray(payload, dtype=float).reshape(-1))
        n = len(payload)
        bv = ctypes.c_double(); bi = ctypes.c_int64(); br = ctypes.c_int()
        bp = np.zeros(max(n, 1))
        _lib.check(_lib.lib.ibo_comm_argmax(self.h, float(val), int(idx), _lib.dp(payload) if n else None, n,
                                            ctypes.byref(bv), ctypes.byref(bi), _lib.dp(bp), ctypes.byref(br)))
        return bv.value, bi.value, bp[:n], br.value

    def allreduce_sum(self, buf):
        out = _lib.f64(np.array(buf, dtype=np.float64))
        _lib.check(_lib.lib.ibo_comm_allreduce_sum(self.h, _lib.dp(out), out.size))
        return out

    def barrier(self):
        _lib.check(_lib.lib.ibo_comm_barrier(self.h))

    def nranks(self):
        """ranks in the communicator as RCCL itself counts them (ncclCommCount)"""
        n = ctypes.c_int(0)
        _lib.check(_lib.lib.ibo_comm_count(self.h, ctypes.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "h", None):
            _lib.lib.ibo_comm_destroy(self.h)
            self.h = None


class SocketComm(object):
    """The slot protocol and the sum over a Unix-domain socket, host memory only: rank 0 listens on `address`
    (a filesystem path), gathers every rank's fp64 buffer, adds them IN RANK ORDER and sends the sum back to all.
    Every slot of the arg-max exchange is non-zero on exactly one rank, so the sum there is exact and every rank
    applies the same final reduction as after ncclAllReduce (csrc/comm.hip: slot_argmax).  What it is for: running
    the sharded paths with more than one rank where RCCL cannot -- two rank processes on the ONE GPU of a build box
    (tests/test_gpu_two_ranks.py) -- with no second GPU runtime and no torch in the process."""

    def __init__(self, world_size, rank, address, timeout_s=300.0):
        import socket
        import struct
        import time
        self.world_size, self.rank, self.address = int(world_size), int(rank), address
        self._peers, self._listener = {}, None
        if self.world_size == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(address)
            srv.listen(self.world_size)
            srv.settimeout(timeout_s)
            self._listener = srv
            while len(self._peers) < self.world_size - 1:
                c, _ = srv.accept()
                c.settimeout(timeout_s)
                r = struct.unpack("<q", self._recv_exact(c, 8))[0]
                if not (0 < r < self.world_size) or r in self._peers:
                    raise RuntimeError("SocketComm: unexpected rank %d on %s" % (r, address))
                self._peers[r] = c
        else:
            t0 = time.time()
            while True:
                c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    c.connect(address)
                    break
                except OSError:
                    c.close()
                    if time.time() - t0 > timeout_s:
                        raise RuntimeError("SocketComm: rank 0 never listened on %s" % address)
                    time.sleep(0.01)
            c.settimeout(timeout_s)
            c.sendall(struct.pack("<q", self.rank))
            self._peers[0] = c

    @staticmethod
    def _recv_exact(sock, n):
        chunks, got = [], 0
        while got < n:
            b = sock.recv(min(n - got, 1 << 20))
            if not b:
                raise RuntimeError("SocketComm: peer closed the connection")
            chunks.append(b)
            got += len(b)
        return b"".join(chunks)

    def allreduce_sum(self, buf):
        a = np.ascontiguousarray(buf, dtype=np.float64).reshape(-1).copy()
        if self.world_size == 1:
            return a
        if self.rank == 0:
            for r in range(1, self.world_size):
                a += np.frombuffer(self._recv_exact(self._peers[r], a.nbytes), dtype=np.float64)
            raw = a.tobytes()
            for r in range(1, self.world_size):
                self._peers[r].sendall(raw)
            return a
        self._peers[0].sendall(a.tobytes())
        return np.frombuffer(self._recv_exact(self._peers[0], a.nbytes), dtype=np.float64).copy()

    def argmax(self, val, idx, payload=()):
        payload = np.asarray(payload, dtype=float).reshape(-1)
        out = self.allreduce_sum(fill_slot(self.world_size, self.rank, val, idx, payload))
        return reduce_slots(out, self.world_size, len(payload))

    def barrier(self):
        self.allreduce_sum(np.zeros(1))

    def nranks(self):
        return self.world_size

    def close(self):
        import os
        for c in self._peers.values():
            try:
                c.close()
            except OSError:
                pass
        self._peers = {}
        if self._listener is not None:
            self._listener.close()
            self._listener = None
            try:
                os.unlink(self.address)
            except OSError:
                pass


def _rendezvous_path():
    """where rank 0 leaves the RCCL unique id for the other ranks of a one-node job.
    IBO_COMM_ID_FILE names the file outright (bench.py's self-launcher creates a fresh private
    directory per run and passes it down).  Otherwise the file lives in a per-user 0700 directory
    and its name carries the launcher's pid, MASTER_PORT and torch-elastic's run id + restart
    count, so a restarted attempt never reads the id of the attempt before it."""
    import os
    explicit = os.environ.get("IBO_COMM_ID_FILE")
    if explicit:
        return explicit
    base = os.environ.get("IBO_COMM_DIR") or os.path.join("/tmp", "ibo_rccl_%d" % os.getuid())
    try:
        os.mkdir(base, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(base)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("rendezvous directory %s is not a private directory of this user" % base)
    nonce = "%s_%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    nonce = "".join(ch if ch.isalnum() or ch in "-_" else "_" for ch in nonce)
    return os.path.join(base, "id_%d_%s_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"), nonce))


def exchange_unique_id(world_size, rank, timeout_s=600.0):
    """Hand rank 0's RCCL unique id to every rank of a ONE-NODE job without pulling a second
    GPU runtime into the process (torch bundles its own HIP/HSA; two runtimes in one process
    fail with "no ROCm-capable device" or corrupt the heap at exit).  Rank 0 creates the file
    exclusively (O_EXCL | O_NOFOLLOW, mode 0600) under a temporary name and renames it into
    place; the others poll, and accept only a regular file of the right size owned by this user
    and not older than the launcher.  Multi-node launches pass the id by their own means."""
    import os
    import time
    path = _rendezvous_path()
    if rank == 0:
        uid = RcclArgmax.unique_id()
        try:
            os.unlink(path)
        except OSError:
            pass
        tmp = "%s.%d.tmp" % (path, os.getpid())
        try:
            os.unlink(tmp)
        except OSError:
            pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(uid)
        os.rename(tmp, path)
        return uid, path
    try:
        born = os.stat("/proc/%d" % os.getppid()).st_ctime - 5.0     # ignore files older than the launcher
    except OSError:
        born = 0.0
    import stat
    t0 = time.time()
    while True:
        try:
            fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
            try:
                st = os.fstat(fd)
                if (stat.S_ISREG(st.st_mode) and st.st_uid == os.getuid() and st.st_size == _lib.COMM_ID_BYTES
                        and st.st_mtime >= born):
                    return os.read(fd, _lib.COMM_ID_BYTES), path
            finally:
                os.close(fd)
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise RuntimeError("timed out waiting for the RCCL unique id at %s" % path)
        time.sleep(0.01)


def sharded_sweep(model, local_candidates, start, comm, **sweep_kw):
    """sweep this rank's block (rows [start, start+len)) and agree on the global arg-max.
    Returns dict(best_val, best_idx (global), best_x, best_rank, kernel_ms)."""
    from .acquisition import sweep
    sweep_kw.pop('index_base', None)
    if getattr(comm, "device_exchange", False) and isinstance(local_candidates, _lib.DeviceArray) and not sweep_kw.get("outputs"):
        # RCCL: sweep and exchange in one device-side call -- the local arg-max never visits the host on its way into the all-reduce
        r = sweep(model, local_candidates, index_base=start, exchange=comm, **sweep_kw)
        return dict(best_val=r["global_val"], best_idx=r["global_idx"], best_x=r["global_x"], best_rank=r["global_rank"],
                    kernel_ms=r["kernel_ms"], local=r)
    r = sweep(model, local_candidates, index_base=start, **sweep_kw)
    D = local_candidates.shape[1]
    if r["best_idx"] >= 0:
        li = r["best_idx"] - start
        if isinstance(local_candidates, _lib.DeviceArray):
            x = local_candidates.view_rows(li, li + 1).to_host()[0]
        else:
            x = np.asarray(local_candidates[li], dtype=float)
    else:
        x = np.zeros(D)
    v, i, p, rk = comm.argmax(r["best_val"], r["best_idx"], x)
    return dict(best_val=v, best_idx=i, best_x=p, best_rank=rk, kernel_ms=r["kernel_ms"], local=r)


def sharded_nlml_grid(Kernel, thetas, X, Y, comm, noise=1e-3, device=None, local_eval=None):
    """marginal-likelihood grid with the theta-points cut into contiguous blocks, one per rank
    (SURVEY 8e): each rank evaluates its block on its GPU, one all-reduce(sum) over a zero-padded
    buffer gathers all values, every rank returns (values, argmin).  local_eval(thetas_block) ->
    values replaces the GPU evaluation in the CPU protocol test."""
    thetas = np.asarray(thetas, dtype=float)
    T = len(thetas)
    a, b = shard_bounds(T, comm.world_size, comm.rank)
    buf = np.zeros(2 * T)
    if b > a:
        if local_eval is None:
            from .gaussianprocess.trainhyper import nlml_values
            vals = nlml_values(None, X, Y, noise, device, spec=Kernel._ibo_spec_rows(thetas[a:b]))
        else:
            vals = np.asarray(local_eval(thetas[a:b]), dtype=float)
        ok = np.isfinite(vals)
        buf[a:b] = np.where(ok, vals, 0.0)
        buf[T + a:T + b] = ok.astype(float)           # NaN (not positive definite) cannot ride a sum
    out = comm.allreduce_sum(buf)
    vals = np.where(out[T:] > 0.5, out[:T], np.nan)
    return vals, int(np.nanargmin(vals))


def sharded_gallery(GP, bounds, N, local_candidates, start, comm, **kw):
    """fastUCBGallery with the candidate step sharded: every rank holds the model and runs the
    (sequential, single-GPU) DIRECT step redundantly, sweeps its own candidate block, and the
    per-round arg-max travels in the one RCCL exchange together with the winner's coordinates,
    so all ranks hallucinate the same observation and stay in lock-step."""
    from .acquisition.gallery import fastUCBGallery
    return fastUCBGallery(GP, bounds, N, candidates=local_candidates, comm=comm, index_base=start, **kw)
