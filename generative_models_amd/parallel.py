"""Data-parallel gradient exchange for the flat gradient arena: one process per GPU, RCCL over xGMI.

The reference has no distributed code at all; data parallelism is the one exchange step the
hot path needs: per-sample work is independent (GroupNorm normalises within a sample), so ranks
take equal shards of the global batch and sum-all-reduce the flat fp32 gradient once per step.  The arena is cut
into 4 contiguous buckets in backward-readiness order (`SimpleUnet.grad_buckets`); each bucket's all-reduce is
issued the moment its last gradient kernel has been enqueued, so it runs on RCCL's stream underneath the rest of
the backward pass.  xGMI is point-to-point and a 24 MB all-reduce is latency-bound, hence few, large buckets.
The 1/world scaling is folded into the fused Adam kernel (`grad_scale`).
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


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


class GradSync:
    """Bucketed, overlapped sum-all-reduce of `net.flat_grads`.  `hook(k)` is SimpleUnet.backward_hip's
    `on_grads_ready` callback; `finish()` makes the current stream wait for every outstanding bucket."""

    def __init__(self, net, group=None):
        self.net = net
        self.group = group
        self.buckets = net.grad_buckets()
        self.works = []

    def hook(self, k):
        if world() == 1:
            return
        s, e = self.buckets[k]
        self.works.append(dist.all_reduce(self.net.flat_grads[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for w in self.works:
            w.wait()
        self.works.clear()

    def broadcast_params(self, src=0):
        if world() > 1:
            dist.broadcast(self.net.flat_params, src=src, group=self.group)
            self.net.mark_params_changed()
