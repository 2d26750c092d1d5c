"""Data-parallel plumbing: rank discovery, shard ranges, RCCL bootstrap.

The reference is single-process; this is new.  One process per
GPU, launched by ``python -m torch.distributed.run`` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment).  The data path has
exactly one exchange per step: a flat fp32 sum-all-reduce of the gradient buffer
(RCCL over xGMI, issued by libtheanet_hip.so on the compute stream).

The 128-byte RCCL unique id is exchanged with a tiny TCP rendezvous written on
plain sockets (rank 0 listens on MASTER_PORT + 101) so the hot path never
imports torch (torch ships its own HIP runtime).  CPU tests exercise the shard
math and the rendezvous with world_size 2, and the host-buffer reduction through
``torch.distributed``'s gloo backend (``HostGroup``).
"""
import os
import sys
import socket
import struct
import time

import numpy as np

RDZV_PORT_OFFSET = 101


class World:
    def __init__(self, rank=0, size=1, local_rank=0, master_addr="127.0.0.1", master_port=29500, dry=False):
        self.rank, self.size, self.local_rank = rank, size, local_rank
        self.master_addr, self.master_port = master_addr, master_port
        self.dry = dry          # plan only (bench.py --dry-multi): shards and buffers of this rank, no communicator

    @classmethod
    def from_env(cls, env=None):
        env = os.environ if env is None else env
        return cls(int(env.get("RANK", 0)), int(env.get("WORLD_SIZE", 1)),
                   int(env.get("LOCAL_RANK", 0)), env.get("MASTER_ADDR", "127.0.0.1"),
                   int(env.get("MASTER_PORT", 29500)))


def shard_rows(global_batch, world_size, rank):
    """Rows [lo, hi) of a minibatch owned by ``rank`` (equal shards; the loss is a
    batch MEAN -- outlayers.py:50-51 -- so equal shards make g = mean_r g_r)."""
    if global_batch % world_size:
        raise ValueError("BATCH_SZ %d is not divisible by the number of GPUs %d"
                         % (global_batch, world_size))
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


def minibatch_row0(index, global_batch, world_size, rank):
    """First dataset row of minibatch ``index`` for ``rank`` (neuralnet.py:222-224)."""
    lo, _ = shard_rows(global_batch, world_size, rank)
    return index * global_batch + lo


def flat_layout(sizes, align=64):
    """Offsets of tensors of ``sizes`` elements inside ONE flat fp32 buffer (each start
    aligned to ``align`` elements = 256 B) followed by one extra scalar slot (the cost), so a
    single all-reduce moves every gradient and the loss.  Returns (offsets, cost_offset,
    n_reduce) where n_reduce = number of elements the all-reduce must cover."""
    offsets, total = [], 0
    for n in sizes:
        offsets.append(total)
        total += -(-int(n) // align) * align
    return offsets, total, total + 1


# --------------------------------------------------------------------------- #
# socket rendezvous: broadcast of a small blob from rank 0, and a barrier
# --------------------------------------------------------------------------- #

def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return buf


class Rendezvous:
    """Star topology over TCP: rank 0 accepts world-1 connections and keeps them."""

    def __init__(self, world, timeout=300.0):
        self.world = world
        self.conns = []          # rank 0: sockets to every other rank
        self.sock = None         # other ranks: socket to rank 0
        if world.size == 1:
            return
        port = world.master_port + RDZV_PORT_OFFSET
        if world.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("", port))
            srv.listen(world.size)
            srv.settimeout(timeout)
            peers = {}
            while len(peers) < world.size - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                r = struct.unpack("<i", _recv_exact(c, 4))[0]
                peers[r] = c
            srv.close()
            self.conns = [peers[r] for r in sorted(peers)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((world.master_addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.settimeout(timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.sendall(struct.pack("<i", world.rank))
            self.sock = s

    def broadcast(self, blob, nbytes):
        """Rank 0's ``blob`` (bytes) is returned on every rank."""
        if self.world.size == 1:
            return blob
        if self.world.rank == 0:
            assert len(blob) == nbytes
            for c in self.conns:
                c.sendall(blob)
            return blob
        return _recv_exact(self.sock, nbytes)

    def gather_max(self, value):
        """max over ranks of a python float, returned on every rank (timing)."""
        if self.world.size == 1:
            return value
        if self.world.rank == 0:
            vals = [value] + [struct.unpack("<d", _recv_exact(c, 8))[0] for c in self.conns]
            m = max(vals)
            for c in self.conns:
                c.sendall(struct.pack("<d", m))
            return m
        self.sock.sendall(struct.pack("<d", value))
        return struct.unpack("<d", _recv_exact(self.sock, 8))[0]

    def barrier(self):
        self.gather_max(0.0)

    def allreduce_host(self, arr, op="sum"):
        """In-place all-reduce of a HOST float32 array through rank 0 (summed in rank order, so every
        rank gets the same bits): the exchange of the CPU backend's data-parallel ranks."""
        if self.world.size == 1:
            return arr
        nbytes = arr.nbytes
        if self.world.rank == 0:
            acc = arr.copy()
            for c in self.conns:
                other = np.frombuffer(_recv_exact(c, nbytes), dtype=arr.dtype).reshape(arr.shape)
                acc = acc + other if op == "sum" else np.maximum(acc, other)
            arr[...] = acc
            blob = arr.tobytes()
            for c in self.conns:
                c.sendall(blob)
        else:
            self.sock.sendall(arr.tobytes())
            arr[...] = np.frombuffer(_recv_exact(self.sock, nbytes), dtype=arr.dtype).reshape(arr.shape)
        return arr

    def rsag_host(self, arr):
        """The CPU backend's restatement of tn_allreduce_sum_rsag on a HOST float32 vector, in place: rank r owns
        elements [r q, (r + 1) q), q = n // world; every slice is summed in rank order at its owner (rank 0 stands in
        for the owners here: the star of sockets has no rank-to-rank links) and gathered by everybody; the n % world
        elements behind the slices go through the all-reduce.  Rank order at the owner = rank order of
        ``allreduce_host``: the same bits."""
        W = self.world.size
        if W == 1:
            return arr
        flat = arr.reshape(-1)
        n = flat.size
        q = n // W
        body = q * W
        if q:
            if self.world.rank == 0:
                parts = [flat[:body].copy()] + [np.frombuffer(_recv_exact(c, 4 * body), dtype=np.float32) for c in self.conns]
                out = np.empty(body, np.float32)
                for owner in range(W):                      # reduce-scatter: the owner's slice, contributions in rank order
                    sl = slice(owner * q, (owner + 1) * q)
                    acc = parts[0][sl].copy()
                    for p_ in parts[1:]:
                        acc = acc + p_[sl]
                    out[sl] = acc
                flat[:body] = out                           # all-gather
                blob = out.tobytes()
                for c in self.conns:
                    c.sendall(blob)
            else:
                self.sock.sendall(flat[:body].tobytes())
                flat[:body] = np.frombuffer(_recv_exact(self.sock, 4 * body), dtype=np.float32)
        if n > body:
            self.allreduce_host(flat[body:], "sum")
        return arr

    def close(self):
        for c in self.conns:
            c.close()
        if self.sock:
            self.sock.close()
        self.conns, self.sock = [], None


_rdzv = None


def get_rendezvous():
    """The process-wide socket rendezvous of the launch (created on first use; every rank must reach
    its first use -- train.py's seed broadcast or the RCCL bootstrap -- at the same point)."""
    global _rdzv
    if _rdzv is None:
        _rdzv = Rendezvous(get_world())
    return _rdzv


def broadcast_int(value):
    """Rank 0's integer on every rank (e.g. the SEED every replica must build its net from)."""
    r = get_rendezvous()
    return struct.unpack("<q", r.broadcast(struct.pack("<q", int(value)), 8))[0]


def agree(value, what="value", rdzv=None):
    """Raise unless ``value`` (a float) is identical on every rank -- replicas that disagree on their
    weights would silently all-reduce gradients of different nets."""
    r = rdzv if rdzv is not None else get_rendezvous()
    hi, lo = r.gather_max(float(value)), -r.gather_max(-float(value))
    if hi != lo:
        raise RuntimeError("data-parallel ranks disagree on %s (min %r, max %r): every rank must build "
                           "the net from the same SEED / weights" % (what, lo, hi))


# set (on every rank alike) when a communicator's start-up self-test found the reduce-scatter + all-gather form
# unequal to one all-reduce: every bucket then keeps the plain all-reduce (DeviceGroup.self_test_rsag)
_rsag_disabled = False


def collective_algo(nfloats, world_size, env=None):
    """Which form a bucket's sum takes: 'rsag' = direct reduce-scatter + all-gather
    (tn_allreduce_sum_rsag) for buckets of TN_DP_RSAG_MIN_BYTES (default 8 MB) or more -- on the fully connected xGMI a
    ring moves 2 (W-1)/W S through every link in turn, the two half-collectives 2 S/W per link pair: wide6's 67 MB dense
    bucket 820 us against 117 us of wire at 8 GPUs --, 'allreduce' (one RCCL call, the library's own choice of
    algorithm) for the small, latency-bound ones.  TN_DP_ALGO=allreduce / rsag forces one form for every bucket
    (rsag also with one rank: the GPU tests exercise the entry point that way).  The choice depends on the bucket
    size and the environment only: the same on every rank."""
    env = os.environ if env is None else env
    force = env.get("TN_DP_ALGO", "auto")
    if force in ("allreduce", "rsag"):
        return force
    if world_size < 2 or _rsag_disabled:
        return "allreduce"
    return "rsag" if 4 * int(nfloats) >= int(env.get("TN_DP_RSAG_MIN_BYTES", 8 << 20)) else "allreduce"


class DeviceGroup:
    """RCCL communicator bound to a theanet_amd Context (GPU path)."""

    def __init__(self, ctx, world, rdzv=None):
        import ctypes
        self.ctx, self.world = ctx, world
        self.rdzv = rdzv if rdzv is not None else (get_rendezvous() if world is get_world() else Rendezvous(world))
        idbuf = ctypes.create_string_buffer(128)
        if world.rank == 0:
            ctx.call("tn_comm_unique_id", idbuf)
        blob = self.rdzv.broadcast(idbuf.raw, 128)
        idbuf = ctypes.create_string_buffer(blob, 128)
        ctx.call("tn_comm_init", idbuf, world.rank, world.size)
        # every collective issued on this communicator, as a running hash of (sequence number, kind,
        # element count): all ranks must issue the same sequence (the two streams of the pipelined
        # schedule alternate on ONE communicator) -- verify_order() compares it across ranks
        self.n_issued, self.order_hash = 0, 0
        self.check_every_call = os.environ.get("TN_DP_CHECK_ORDER") == "1"
        if world.size > 1 and os.environ.get("TN_COMM_SELFTEST", "1") != "0":
            self.self_test()

    def self_test(self, timeout=None):
        """One-shot check of a fresh communicator, run at creation when there is more than one rank: three
        all-reduces of rank-stamped vectors in the patterns the schedules use (stream 0, stream 1, then one issued
        from stream 0 onto the communication stream, all on ONE communicator, each ordered behind the previous by an
        event), then the
        sums are verified on every rank.  A watchdog polls the last event (tn_event_query) and RAISES after
        ``timeout`` seconds (TN_COMM_SELFTEST_TIMEOUT, default 60) instead of hanging in a sync: a device-side
        ordering problem of the alternation shows up here, by name, and not as a silent stall of the first
        training step."""
        import ctypes
        ctx, R, r = self.ctx, self.world.size, self.world.rank
        timeout = float(os.environ.get("TN_COMM_SELFTEST_TIMEOUT", 60)) if timeout is None else timeout
        n = 1024
        pat = (np.arange(n) % 7 + 1).astype(np.float32)
        bufs = [ctx.array((r + 1) * pat), ctx.array((r + 1) * 2 * pat)]
        evs = []
        for _ in range(3):
            e = ctypes.c_void_p()
            ctx.call("tn_event_create", ctypes.byref(e))
            evs.append(e)
        try:
            for k, (stream, b) in enumerate(((0, 0), (1, 1), (0, 0))):
                ctx.call("tn_stream_select", stream)
                if k:
                    ctx.call("tn_event_wait", evs[k - 1])
                if k == 2:       # the pipelined schedule's form: on the communication stream, event behind it
                    self.allreduce_sum_async(bufs[b], n, evs[k])
                else:
                    self.allreduce_sum(bufs[b], n)
                    ctx.call("tn_event_record", evs[k])
            ctx.call("tn_stream_select", 0)
            done, t0 = ctypes.c_int(0), time.time()
            while True:
                ctx.call("tn_event_query", evs[2], ctypes.byref(done))
                if done.value:
                    break
                if time.time() - t0 > timeout:
                    raise RuntimeError(
                        "theanet_amd: communicator self-test timed out after %.0f s on rank %d of %d: three all-reduces "
                        "alternating between the context's two streams did not finish (RCCL / xGMI bring-up, or a rank "
                        "that never joined)" % (timeout, r, R))
                time.sleep(0.002)
            tri = R * (R + 1) / 2.0
            want = [tri * R * pat, tri * 2 * pat]        # buffer 0 went through two all-reduces
            for b, w in zip(bufs, want):
                got = b.get_value()
                if not np.array_equal(got, w.astype(np.float32)):
                    bad = int(np.argmax(got != w))
                    raise RuntimeError("theanet_amd: communicator self-test failed on rank %d of %d: element %d is %r, "
                                       "expected %r" % (r, R, bad, float(got[bad]), float(w[bad])))
        finally:
            ctx.call("tn_stream_select", 0)
            for e in evs:
                ctx.lib.tn_event_destroy(ctx.h, e)
        self.verify_order()
        self.self_test_rsag(timeout)

    def self_test_rsag(self, timeout=60.0):
        """The reduce-scatter + all-gather form (tn_allreduce_sum_rsag: in-place ncclReduceScatter into the rank's own
        slice, ncclAllGather, a small all-reduce for the n % world floats behind the slices) against ONE all-reduce of
        the same vectors, on every rank, before any gradient bucket may take it: integer-valued rank-stamped data (sums
        exact in any order, so a differing BIT is a wrong offset or count, not a rounding), lengths that the world size
        does and does not divide, on the compute stream and on the communication stream behind an event.  A mismatch
        on ANY rank switches the form off on EVERY rank (comm.collective_algo then answers 'allreduce'
        for every bucket) with a loud message -- it does not raise: the plain all-reduce just passed its own test.
        TN_DP_ALGO=rsag still forces the form."""
        import ctypes
        global _rsag_disabled
        ctx, R, r = self.ctx, self.world.size, self.world.rank
        if R < 2 or os.environ.get("TN_DP_ALGO", "auto") == "allreduce":
            return True
        ok, why = 1.0, ""
        cpu = ctx.backend == "cpu"
        ev = ctypes.c_void_p()
        if not cpu:
            ctx.call("tn_event_create", ctypes.byref(ev))
        try:
            for n, on_comm in ((R * 1024, False), (R * 1024 + R - 1, False), (5 * R + 3, True), (3, False),
                               (R * 4096 + 1, True)):
                pat = ((np.arange(n) * 7 + 3) % 251).astype(np.float32)
                src = (r + 1) * pat + r
                a, b = ctx.array(src), ctx.array(src)
                if cpu:
                    self.rdzv.allreduce_host(self._host_view(a, n), "sum")
                    self.rdzv.rsag_host(self._host_view(b, n))
                else:
                    ctx.call("tn_allreduce_sum", a.ptr, n)
                    if on_comm:
                        ctx.call("tn_allreduce_sum_rsag", b.ptr, n, 1, ev)
                        done, t0 = ctypes.c_int(0), time.time()
                        while not done.value:
                            ctx.call("tn_event_query", ev, ctypes.byref(done))
                            if time.time() - t0 > timeout:
                                raise RuntimeError("theanet_amd: communicator self-test timed out after %.0f s on rank %d of "
                                                   "%d: reduce-scatter + all-gather of %d floats on the communication "
                                                   "stream did not finish (set TN_DP_ALGO=allreduce to skip the form)"
                                                   % (timeout, r, R, n))
                            time.sleep(0.002)
                    else:
                        ctx.call("tn_allreduce_sum_rsag", b.ptr, n, 0, None)
                    ctx.sync()
                want = (R * (R + 1) / 2.0) * pat + R * (R - 1) / 2.0
                ga, gb = a.get_value(), b.get_value()
                if os.environ.get("TN_TEST_BREAK_RSAG") == str(r) and n > 8:     # (tests: a wrong offset on one rank)
                    gb = np.roll(gb, 1)
                if not np.array_equal(ga, want.astype(np.float32)):
                    raise RuntimeError("theanet_amd: all-reduce of %d floats is wrong on rank %d of %d" % (n, r, R))
                if ok and not np.array_equal(gb, ga):       # (no early exit: the ranks stay in step through every case)
                    bad = int(np.argmax(gb != ga))
                    ok, why = 0.0, "%d floats: element %d is %r, one all-reduce gives %r" % (n, bad, float(gb[bad]), float(ga[bad]))
        finally:
            if not cpu:
                ctx.lib.tn_event_destroy(ctx.h, ev)
        all_ok = -self.rdzv.gather_max(-ok)          # the minimum over ranks
        if all_ok < 1.0:
            _rsag_disabled = True
            print("theanet_amd: WARNING: the reduce-scatter + all-gather all-reduce failed its start-up check%s; every "
                  "gradient bucket keeps one ncclAllReduce (TN_DP_ALGO=allreduce)" %
                  ((" on this rank (rank %d: %s)" % (r, why)) if not ok else " on another rank"), file=sys.stderr, flush=True)
        self.rsag_checked = all_ok >= 1.0
        return self.rsag_checked

    def _note(self, kind, count):
        self.n_issued += 1
        import zlib      # (python's hash() of a str is salted per process: not comparable across ranks)
        self.order_hash = (self.order_hash * 1000003 +
                           zlib.crc32(("%d %s %d" % (self.n_issued, kind, int(count))).encode())) % (1 << 52)
        if self.check_every_call:
            self.verify_order()

    def verify_order(self):
        """Raise unless every rank has issued the same sequence of collectives so far (host-side
        bookkeeping compared over the socket rendezvous; call at points where the host syncs anyway)."""
        agree(float(self.order_hash), "the order of collectives issued so far (%d on this rank)" % self.n_issued,
              self.rdzv)

    def _host_view(self, darr, n):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_float * n).from_address(darr.ptr))

    def allreduce_sum(self, darr, count=None):
        n = darr.size if count is None else count
        algo = collective_algo(n, self.world.size)
        self._note("sum:" + algo, n)
        if self.ctx.backend == "cpu" and self.world.size > 1:      # "device" memory is host memory there
            self.ctx.rec_tainted = True                            # (host-side work: not a step tn_net_step can replay)
            if algo == "rsag":
                self.rdzv.rsag_host(self._host_view(darr, n))
            else:
                self.rdzv.allreduce_host(self._host_view(darr, n), "sum")
        elif algo == "rsag":
            self.ctx.call("tn_allreduce_sum_rsag", darr.ptr, n, 0, None)
        else:
            self.ctx.call("tn_allreduce_sum", darr.ptr, n)

    def allreduce_sum_async(self, darr, count=None, done_ev=None):
        """The sum on the context's communication stream (tn_allreduce_sum_async / tn_allreduce_sum_rsag): behind what
        the compute stream holds so far, beside what it does next; ``done_ev`` is recorded behind it for the consumer
        to wait on."""
        n = darr.size if count is None else count
        algo = collective_algo(n, self.world.size)
        self._note("sum:" + algo, n)
        if self.ctx.backend == "cpu" and self.world.size > 1:      # host code is synchronous: nothing to order
            self.ctx.rec_tainted = True
            if algo == "rsag":
                self.rdzv.rsag_host(self._host_view(darr, n))
            else:
                self.rdzv.allreduce_host(self._host_view(darr, n), "sum")
        elif algo == "rsag":
            self.ctx.call("tn_allreduce_sum_rsag", darr.ptr, n, 1, done_ev)
        else:
            self.ctx.call("tn_allreduce_sum_async", darr.ptr, n, done_ev)

    def allreduce_max(self, darr, count=None):
        n = darr.size if count is None else count
        self._note("max", n)
        if self.ctx.backend == "cpu" and self.world.size > 1:
            self.rdzv.allreduce_host(self._host_view(darr, n), "max")
        else:
            self.ctx.call("tn_allreduce_max", darr.ptr, n)

    def barrier(self):
        self.ctx.sync()
        self.rdzv.barrier()
        if self.world.size > 1:
            self.verify_order()


class HostGroup:
    """Same reduction contract on HOST numpy buffers through torch.distributed
    (gloo).  Used by the CPU test-suite to cover the N>1 logic without GPUs."""

    def __init__(self, world):
        import torch.distributed as dist
        self.world = world
        self.dist = dist
        if not dist.is_initialized():
            dist.init_process_group(
                "gloo", init_method="tcp://%s:%d" % (world.master_addr, world.master_port),
                rank=world.rank, world_size=world.size)

    def allreduce_sum(self, arr):
        import torch
        t = torch.from_numpy(arr)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return arr

    def barrier(self):
        self.dist.barrier()


_world = None


def get_world():
    global _world
    if _world is None:
        _world = World.from_env()
    return _world
