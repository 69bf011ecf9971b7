"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed
backend "nccl" on ROCm) all-reduce of contiguous buckets of the flat gradient
buffer, issued as soon as the backward pass has finished a bucket so the
transfer over xGMI overlaps the remaining backward kernels.

The reference has no distributed code at all; this is the one
exchange step data-parallel training needs.  Buckets = model.bucket_ranges():
[embedding(+condition)], [layer 0], ..., [layer N-1], [head]; backward completes
them in reverse order.  The sum is turned into a mean by folding 1/world_size
into the optimiser's grad_scale (FusedAdamW.step(grad_scale=...)), so no extra
pass over the gradients is needed and the global-norm clip sees averaged grads.
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, flat_grads_fn, bucket_ranges, group=None):
        self._flat = flat_grads_fn
        self.ranges = list(bucket_ranges)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._works = []
        self._done = set()

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def hook(self, bucket_index):
        """Called by the engine's backward when bucket `bucket_index` is final on the compute stream."""
        if self.world == 1:
            return
        if bucket_index in self._done:
            raise RuntimeError("bucket %d reduced twice in one step" % bucket_index)
        self._done.add(bucket_index)
        lo, hi = self.ranges[bucket_index]
        if hi > lo:
            self._works.append(dist.all_reduce(self._flat()[lo:hi], op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))

    def finish(self):
        """Make the compute stream wait for every outstanding bucket (no host block on GPU backends)."""
        if self.world == 1:
            return
        if len(self._done) != len(self.ranges):
            missing = sorted(set(range(len(self.ranges))) - self._done)
            raise RuntimeError("backward did not report buckets %s" % missing)
        for w in self._works:
            w.wait()
        self._works.clear()
        self._done.clear()


def broadcast_params(flat_params, src=0, group=None):
    """Identical initial weights on every rank."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
