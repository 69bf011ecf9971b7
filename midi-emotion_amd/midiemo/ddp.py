"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed
backend "nccl" on ROCm) all-reduce of contiguous buckets of the flat gradient
buffer, issued while the backward pass is still running so that the transfer over
xGMI overlaps it (by default inside the attention-backward windows, see GradAllReducer).

The reference has no distributed code at all; this is the one
exchange step data-parallel training needs.  Buckets = model.bucket_ranges():
[embedding(+condition)], [layer 0], ..., [layer N-1], [head]; backward completes
them in reverse order.  The sum is turned into a mean by folding 1/world_size
into the optimiser's grad_scale (FusedAdamW.step(grad_scale=...)), so no extra
pass over the gradients is needed and the global-norm clip sees averaged grads.
"""
import collections

import torch
import torch.distributed as dist


class GradAllReducer:
    """policy (env MIDIEMO_DDP_POLICY):
      "window" (default) -- a finished bucket is parked and launched when the engine announces a comm window
                 (hook(-1): the attention backward of the next layer, ~0.5 ms of kernels with thousands of small blocks,
                 is about to be enqueued).  Why: an RCCL channel is a 256-VGPR, 37 KB-LDS workgroup that owns its CU for
                 the whole collective, and the GEMMs are persistent kernels of exactly one block per CU -- every CU taken
                 away makes one of their blocks wait for a whole tile round (up to 2x for that launch), whereas the
                 attention kernels just lose k/256 of their throughput.
      "eager"  -- launch in hook(i) as soon as the bucket is final (maximal overlap, GEMMs included).
      "end"    -- one all-reduce over the whole flat buffer in finish() (no overlap, no contention).
      "auto"   -- data-driven choice between "end" and "window" (round 5): the first AUTO_PROBE steps run under "end", the
                 next AUTO_PROBE under "window"; each step's span from its first bucket hook (start of the backward) to the
                 end of the waits in finish() is timed with device events, the first step of each phase is dropped, the
                 means are MAX-reduced over the ranks (every rank decides from the same two numbers) and the cheaper policy
                 runs from then on (MEDIAN of the kept spans since round 6, AUTO_PROBE = 5: four samples per policy).  Measured on one MI355X: "window" costs 0.2 ms per step even with nothing to exchange
                 (RCCL's channels take CUs from the attention backward, profiles/r04_measurements.txt), so it only pays
                 when the exposed all-reduce of "end" is longer than that.  `decision` holds the numbers (bench.py prints
                 them as ddp.policy_decision).  One host synchronisation, once, at the decision.
    compress (env MIDIEMO_DDP_COMPRESS=bf16, opt-in; SURVEY 8e "41.2 MB bf16-compressed"): a bucket travels as bf16 -- cast
                 into a staging buffer, summed by the collective in bf16, cast back into the f32 flat gradient after the wait.
                 Halves the payload (82.4 -> 41.2 MB per step at the headline model); costs two cast passes and bf16
                 rounding of every reduced gradient element (the sum of `world` bf16 values rounded to bf16: ~2^-9 relative
                 per element -- the global-norm clip and Adam then see those values on EVERY rank alike: a ring / tree
                 all-reduce hands every rank the same bits, so the ranks stay bit-identical, asserted in the tests)."""
    AUTO_PROBE = 5

    def __init__(self, flat_grads_fn, bucket_ranges, group=None, policy=None, compress=None):
        import os
        compress = compress if compress is not None else os.environ.get("MIDIEMO_DDP_COMPRESS", "")
        if compress not in ("", "0", "none", "bf16"):
            raise ValueError("MIDIEMO_DDP_COMPRESS must be bf16 (or unset)")
        self.compress = compress == "bf16"
        self._cbuf = None
        self._copy_back = []
        self._flat = flat_grads_fn
        self.ranges = list(bucket_ranges)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # one rank has nothing to exchange; MIDIEMO_DDP_FORCE=1 (test hook) issues the collectives anyway, so the RCCL
        # path (asynchronous all-reduce on RCCL's stream, work handles, comm windows) can be exercised on a 1-GPU box
        self._active = self.world > 1 or (dist.is_initialized() and bool(os.environ.get("MIDIEMO_DDP_FORCE")))
        self.policy = policy or os.environ.get("MIDIEMO_DDP_POLICY", "window")
        if self.policy not in ("window", "eager", "end", "auto"):
            raise ValueError("MIDIEMO_DDP_POLICY must be window, eager, end or auto")
        self.decision = None
        self._auto = None
        if self.policy == "auto":
            self._auto = {"step": 0, "spans": {"end": [], "window": []}, "t0": None}
            self.policy = "end"
        self._works = []
        self._pending = []
        self._done = set()
        # exposed communication: device time the compute stream spends in finish() waiting for bucket work handles
        # (one event pair per step while `timing` is on; read with exposed_ms() after a synchronize)
        self.timing = False
        self._wait_events = collections.deque(maxlen=4096)     # bounded: timing may stay on without exposed_ms() calls

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def _launch(self, lo, hi):
        if hi <= lo:
            return
        flat = self._flat()
        if self.compress:
            if self._cbuf is None or self._cbuf.numel() != flat.numel() or self._cbuf.device != flat.device:
                self._cbuf = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
            buf = self._cbuf[lo:hi]
            buf.copy_(flat[lo:hi])                              # on the compute stream; the collective waits for it
            self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._copy_back.append((lo, hi))
            return
        self._works.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _flush(self):
        # adjacent parked buckets travel as one collective
        runs = []
        for lo, hi in sorted(self._pending):
            if runs and runs[-1][1] == lo:
                runs[-1][1] = hi
            else:
                runs.append([lo, hi])
        self._pending.clear()
        for lo, hi in runs:
            self._launch(lo, hi)

    def hook(self, bucket_index):
        """Called by the engine's backward: bucket_index >= 0 -- that bucket is final on the compute stream;
        -1 -- a comm window opens (see the class docstring)."""
        if not self._active:
            return
        if self._auto is not None and self._auto["t0"] is None:
            self._auto["t0"] = self._stamp()                   # first hook of the step: the backward has just started
        if bucket_index < 0:
            if self.policy == "window":
                self._flush()
            return
        if bucket_index in self._done:
            raise RuntimeError("bucket %d reduced twice in one step" % bucket_index)
        self._done.add(bucket_index)
        if self.policy == "eager":
            self._launch(*self.ranges[bucket_index])
        else:
            self._pending.append(tuple(self.ranges[bucket_index]))

    def finish(self):
        """Launch what is still parked, then make the compute stream wait for every outstanding bucket (no host
        block on GPU backends)."""
        if not self._active:
            return
        if len(self._done) != len(self.ranges):
            missing = sorted(set(range(len(self.ranges))) - self._done)
            raise RuntimeError("backward did not report buckets %s" % missing)
        self._flush()
        if self.timing:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        for w in self._works:
            w.wait()
        for lo, hi in self._copy_back:                          # compressed buckets: bf16 sums back into the f32 gradients
            self._flat()[lo:hi].copy_(self._cbuf[lo:hi])
        self._copy_back.clear()
        if self.timing:
            b.record()
            self._wait_events.append((a, b))
        self._works.clear()
        self._done.clear()
        if self._auto is not None:
            self._auto_step()

    # ---- "auto" policy ------------------------------------------------------------------------------------------------
    def _stamp(self):
        """A point in time on the compute stream (device event) or, for host tensors (gloo tests), on the host clock."""
        if self._flat().is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    @staticmethod
    def _span_ms(a, b):
        return a.elapsed_time(b) if isinstance(a, torch.cuda.Event) else (b - a) * 1e3

    def _auto_step(self):
        st = self._auto
        if st["t0"] is not None:
            st["spans"][self.policy].append((st["t0"], self._stamp()))
        st["t0"] = None
        st["step"] += 1
        n = self.AUTO_PROBE
        if st["step"] == n:
            self.policy = "window"
        elif st["step"] == 2 * n:
            if self._flat().is_cuda:
                torch.cuda.synchronize()
            mean = {}
            for pol, sp in st["spans"].items():
                ms = sorted([self._span_ms(a, b) for a, b in sp[1:]] or [self._span_ms(a, b) for a, b in sp])
                mean[pol] = ms[len(ms) // 2]
            t = torch.tensor([mean["end"], mean["window"]], dtype=torch.float64, device=self._flat().device)
            if dist.is_initialized() and self.world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            end_ms, window_ms = float(t[0]), float(t[1])
            self.policy = "window" if window_ms < end_ms else "end"
            self.decision = {"chosen": self.policy, "backward_plus_exchange_ms": {"end": round(end_ms, 4), "window": round(window_ms, 4)},
                             "probe_steps_per_policy": n, "basis": "first bucket hook -> end of finish() waits, max over ranks"}
            self._auto = None

    def exposed_ms(self, clear=True):
        """Per finish() call since timing was switched on: milliseconds between the compute stream reaching the waits and
        getting past them = communication that was NOT hidden behind the backward pass (what is left parked for finish()
        is launched right before the first event, so its whole duration counts).  Call after torch.cuda.synchronize()."""
        out = [a.elapsed_time(b) for a, b in self._wait_events]
        if clear:
            self._wait_events.clear()
        return out


def broadcast_params(flat_params, src=0, group=None):
    """Identical initial weights on every rank."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
