"""Data-parallel gradient exchange for the flat gradient arena: one process per GPU, RCCL over xGMI.

The reference has no distributed code at all; data parallelism is the one exchange step the
hot path needs: per-sample work is independent (GroupNorm normalises within a sample), so ranks
take equal shards of the global batch and sum-all-reduce the flat fp32 gradient once per step.  The arena is cut
into contiguous buckets in backward-readiness order (`SimpleUnet.grad_buckets`: 4 natural ones; `GMK_GRAD_BUCKETS` = 1, 2 or 4
merges neighbours); each bucket's all-reduce is issued the moment its last gradient kernel has been enqueued, from a third
("exchange") stream that waits for the data-gradient stream AND the weight-gradient side stream - so RCCL sees the bucket final,
while the data-gradient chain itself never waits for the weight gradients (round 2 joined the two streams on the main stream
in front of every bucket: four stalls of the dgrad chain per backward pass) - and runs underneath the rest of the backward pass.  xGMI is point-to-point and a 24 MB all-reduce is latency-bound, hence
few, large buckets.  The 1/world scaling is folded into the fused Adam kernel (`grad_scale`).

CU carve-out: the convolution kernels are persistent grids of one 160 KiB-LDS workgroup per CU — they leave RCCL's reduction
kernels nowhere to run until a whole kernel drains.  With world > 1 the persistent grids are therefore limited to
256 - `GMK_RCCL_CUS` CUs (default 8: RCCL's ring kernels use a handful of workgroups per channel) WHILE BUCKETS ARE IN FLIGHT - from the
first all-reduce of a backward pass to `finish()`; the forward pass, the backward in front of the first bucket and the samplers keep all
256 - costing the convolutions 3 % of the chip for about half of a step and buying overlap of the exchange with the backward pass.  `configure_rccl_env()` - called by the drivers BEFORE the
process group exists - can cap RCCL at that many channels (`NCCL_MAX_NCHANNELS`, one workgroup per channel): a reduction kernel wider
than the carve-out would take its extra CUs at the next kernel boundary, and the following persistent grid (sized for 248 CUs) would then
run a straggler round on whatever is left.  The cap is OPT-IN (`GMK_RCCL_CAP=1`) since round 4: no multi-GPU node was available to measure
it, RCCL may clamp or re-plan its rings under it, and a default nobody has measured should not shape the first scaling curve.  The bench's
N > 1 line carries an A/B of the carve-out itself (`exchange.ab`).
"""
import os

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def exchanging():
    """Does a train step run the gradient exchange?  With more than one rank, always.  `GMK_FORCE_EXCHANGE=1` (round 6) also runs it in a
    ONE-rank process group: a one-rank sum is the identity, so the step's results do not change, but the communicator is created, the four bucket
    all-reduces are issued from the exchange stream behind the two gradient streams, and the persistent kernels run under the carved CU limit
    while they fly - the only execution of RCCL beside the backward pass a one-GPU box can give (tests/test_gpu_ddp.py, bench.py --gpus 1)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("GMK_FORCE_EXCHANGE", "0") == "1"


def shard_batch(x, r=None, w=None):
    """Rank r's contiguous, equal shard of a global batch (dim 0 must divide evenly: equal shards keep the
    mean-of-means equal to the global mean)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    n = x.shape[0]
    if n % w:
        raise ValueError(f"global batch {n} does not split evenly over {w} ranks")
    per = n // w
    return x[r * per:(r + 1) * per]


def merge_buckets(natural, count):
    """Merge the natural buckets (in readiness order) into `count` groups of neighbours.
    -> [(start, end, last natural index of the group)]; every group must stay one contiguous arena range."""
    if count not in (1, 2, 4) or len(natural) != 4:
        raise ValueError(f"GMK_GRAD_BUCKETS must be 1, 2 or 4 (got {count})")
    per = len(natural) // count
    groups = []
    for g in range(count):
        members = natural[g * per:(g + 1) * per]
        start, end = min(s for s, _ in members), max(e for _, e in members)
        if sum(e - s for s, e in members) != end - start:
            raise ValueError("merged gradient bucket is not contiguous")
        groups.append((start, end, g * per + per - 1))
    return groups


def configure_rccl_env():
    """Keep RCCL's reduction kernels inside the CUs `reserve_cus_for_rccl` leaves free.  Environment only: call before
    `init_process_group` (RCCL reads it when the communicator is created); a value the user exported wins."""
    keep = int(os.environ.get("GMK_RCCL_CUS", "8"))
    if os.environ.get("GMK_RCCL_CAP", "0") == "1" and keep > 0 and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(keep))
    return os.environ.get("NCCL_MAX_NCHANNELS")


def carved_cu_limit():
    """(full, carved): the CU count the persistent kernels normally use and the one that leaves GMK_RCCL_CUS CUs to RCCL while gradient
    buckets are in flight (see the module docstring); carved is None when nothing is to be carved (no exchange (see `exchanging`), GMK_RCCL_CUS=0, or a user-fixed
    GMK_CU_LIMIT)."""
    from ._lib import lib
    full = lib.gmk_get_cu_limit()
    keep = int(os.environ.get("GMK_RCCL_CUS", "8"))
    if exchanging() and keep > 0 and "GMK_CU_LIMIT" not in os.environ:
        return full, full - keep
    return full, None


class GradSync:
    """Bucketed, overlapped sum-all-reduce of `net.flat_grads`.  `hook(k)` is SimpleUnet.backward_hip's
    `on_grads_ready` callback (k = natural bucket index, called in order 0..3); `finish()` makes the current stream wait for
    every outstanding bucket."""

    def __init__(self, net, group=None, buckets=None):
        self.net = net
        self.group = group
        count = int(buckets if buckets is not None else os.environ.get("GMK_GRAD_BUCKETS", "4"))
        self.buckets = merge_buckets(net.grad_buckets(), count)
        self._fire = {last: (s, e) for s, e, last in self.buckets}
        self.works = []
        self.issued = []                      # (natural index, start, end) of every all-reduce issued this step (tests, bench)
        self.last_issued = []                 # ... of the last finished step, with the CU limit the persistent kernels ran under behind each
        # the carve-out is applied only while buckets are in flight: from the first all-reduce of a backward pass to finish().  The forward
        # pass, the part of the backward in front of the first bucket and the samplers (no collectives) keep the whole chip
        self._cu_full, self.cu_limit = carved_cu_limit() if net.flat_params.is_cuda else (None, None)
        self._carved = False
        self._limits_seen = []                # gmk_get_cu_limit() behind each all-reduce of this step
        self._comm = None                     # the exchange stream (GPU only)
        self._exposed = []                    # (event before, event after) around finish()'s waits: what the step still waits for

    def hook(self, k):
        if not exchanging() or k not in self._fire:
            return
        s, e = self._fire[k]
        self.issued.append((k, s, e))
        grads = self.net.flat_grads[s:e]
        if not grads.is_cuda:
            self.works.append(dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=grads.device)
        comm = self._comm
        if self.cu_limit is not None and not self._carved:    # every persistent kernel launched from here on leaves CUs to RCCL
            from ._lib import check, lib
            check(lib.gmk_set_cu_limit(self.cu_limit), "set_cu_limit")
            self._carved = True
        comm.wait_stream(torch.cuda.current_stream())         # colsums, GroupNorm parameter gradients, embedding GEMMs
        if getattr(self.net, "_side", None) is not None:
            comm.wait_stream(self.net._side)                  # the bucket's weight gradients
        with torch.cuda.stream(comm):                         # the backend orders its own stream behind the stream current HERE
            self.works.append(dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        from ._lib import lib as _l
        self._limits_seen.append(_l.gmk_get_cu_limit())

    def finish(self):
        """The current stream waits for every outstanding bucket; the wait is bracketed by events (`exposed_ms` in describe())."""
        timed = bool(self.works) and self.net.flat_grads.is_cuda
        if timed:
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
        for w in self.works:
            w.wait()
        if self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)
        if self._carved:                                      # what is launched from here on runs behind the exchange: the whole chip again
            from ._lib import check, lib
            check(lib.gmk_set_cu_limit(self._cu_full), "set_cu_limit")
            self._carved = False
        if timed:
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            self._exposed.append((e0, e1))
            del self._exposed[:-64]
        self.works.clear()
        self.last_issued = [(k, s, e, lim) for (k, s, e), lim in zip(self.issued, self._limits_seen + [None] * len(self.issued))]
        self.issued.clear()
        self._limits_seen.clear()

    def abort(self):
        """A step that raised between hook() and finish(): drain what was issued and give the persistent kernels the whole chip back."""
        try:
            for w in self.works:
                w.wait()
        finally:
            self.works.clear()
            self.issued.clear()
            self._limits_seen.clear()
            if self._carved:
                from ._lib import lib
                lib.gmk_set_cu_limit(self._cu_full)
                self._carved = False

    def set_carve(self, keep):
        """Change the number of CUs left to RCCL while buckets are in flight (0: none); bench.py's `exchange.ab` measures both.  Only between steps."""
        assert not self._carved and not self.works
        if self._cu_full is not None and exchanging() and "GMK_CU_LIMIT" not in os.environ:
            self.cu_limit = self._cu_full - keep if keep > 0 else None
        self._exposed.clear()

    def exposed_ms(self):
        """Mean time the main stream spent waiting in finish() over the last (up to 64) steps; call after a device synchronize."""
        done = [(a, b) for a, b in self._exposed if b.query()]
        return round(sum(a.elapsed_time(b) for a, b in done) / len(done), 4) if done else None

    def broadcast_params(self, src=0):
        if world() > 1:
            dist.broadcast(self.net.flat_params, src=src, group=self.group)
            self.net.mark_params_changed()

    def describe(self):
        """What the exchange looks like from this rank (bench.py puts it beside the scaling numbers)."""
        info = {"world": world(), "backend": dist.get_backend() if exchanging() else None, "forced": exchanging() and world() == 1,
                "bucket_bytes": [4 * (e - s) for s, e, _ in self.buckets], "persistent_kernel_cus": self.cu_limit,
                "rccl_max_channels": os.environ.get("NCCL_MAX_NCHANNELS"),
                "exposed_ms": self.exposed_ms(),
                "issue": "each bucket from a third stream behind the data-gradient and weight-gradient streams (no join on the main stream)"}
        if torch.cuda.is_available():
            try:
                info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as exc:                      # the query needs the RCCL library; report, do not fail a bench over it
                info["rccl_version"] = f"unavailable ({type(exc).__name__})"
        return info
