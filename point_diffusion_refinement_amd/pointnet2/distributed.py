"""Data-parallel gradient averaging for training -- the role of reference pointnet2/distributed.py:94-146
(`apply_gradient_allreduce`): every rank holds a replica, gradients are averaged over ranks after backward.

The reference flattens ALL gradients into one tensor and issues a single blocking all-reduce once backward has
finished.  Here (one process per GPU, RCCL over xGMI through `torch.distributed`, gloo on CPU):

  * parameters are broadcast from rank 0 once (as the reference does);
  * gradients are packed into fixed flat BUCKETS in reverse registration order (the order backward produces
    them); a bucket's all-reduce is launched asynchronously the moment its last gradient has been accumulated, so
    the ring transfers of the deep layers' gradients run while backward is still working on the shallow ones;
  * `synchronize()` (queued automatically at the end of backward) waits for the outstanding handles, scales by
    1 / world_size and copies the averaged values back into `param.grad`.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of the whole 39 MB model moves
2 (W-1)/W * 39 MB over ONE link per direction (~0.5 ms at 8 GPUs), so a few buckets of 4-16 MB keep every transfer
bandwidth-bound rather than latency-bound while still overlapping with backward.
"""
import torch
import torch.distributed as dist
from torch.autograd import Variable


class GradientAllReduce:
    def __init__(self, module, bucket_bytes=8 << 20, group=None):
        assert dist.is_available() and dist.is_initialized(), "init_process_group first"
        self.module, self.group = module, group
        self.world = dist.get_world_size(group)
        for t in module.state_dict().values():                 # replicas start identical (distributed.py:104-107)
            if torch.is_tensor(t):
                dist.broadcast(t, 0, group=group)
        params = [p for p in module.parameters() if p.requires_grad]
        self.buckets = []                                       # [(flat buffer, [(param, offset, numel)])]
        cur, cur_bytes = [], 0
        for p in reversed(params):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        self._where = {}
        for bi, (_, items) in enumerate(self.buckets):
            for i, (p, off, n) in enumerate(items):
                self._where[p] = (bi, off, n, i)
        self._pending = [len(items) for _, items in self.buckets]
        self._handles = []
        self._queued = False
        self._next = 0
        self._fired = set()
        for p in params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def _close(self, params):
        assert all(p.dtype == params[0].dtype and p.device == params[0].device for p in params), \
            "a gradient bucket holds parameters of one dtype on one device"
        total = sum(p.numel() for p in params)
        # one extra element per parameter behind the gradients: 1 when THIS rank produced a gradient for it.  Summed by
        # the same all-reduce it tells every rank whether ANY rank did -- a parameter no rank used keeps .grad = None,
        # exactly like the reference's `param.grad is not None` filter (distributed.py:112) and a single-GPU run
        flat = torch.zeros(total + len(params), dtype=params[0].dtype, device=params[0].device)
        items, off = [], 0
        for p in params:
            items.append((p, off, p.numel()))
            off += p.numel()
        self.buckets.append((flat, items))

    def _on_grad(self, p):
        bi, off, n, i = self._where[p]
        flat, items = self.buckets[bi]
        flat.narrow(0, off, n).copy_(p.grad.reshape(-1))
        # "this rank produced a gradient": the flag slot is set ON THE DEVICE (a fill launch, no host transfer inside
        # the autograd hook: round 4 built the flags on the host and copied them in _launch -- a blocking copy from
        # pageable memory per bucket that stalled backward exactly where the buckets exist to overlap it)
        flat.narrow(0, flat.numel() - len(items) + i, 1).fill_(1)
        self._fired.add(p)
        self._pending[bi] -= 1
        # a complete bucket starts its ring transfer now -- in BUCKET ORDER (collectives are matched across
        # ranks by issue order, so every rank must launch 0, 1, 2, ... whatever order its hooks fire in)
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
        if not self._queued:                                    # once per backward: finish after the engine is done
            self._queued = True
            Variable._execution_engine.queue_callback(self.synchronize)

    def _launch(self, bi):
        flat, items = self.buckets[bi]
        total = flat.numel() - len(items)
        fired = [p in self._fired for p, _, _ in items]
        if self._pending[bi] > 0:                               # parameters without a gradient contribute zeros
            for (p, off, n), f in zip(items, fired):
                if not f:                                      # (also covers a stale .grad of an earlier step)
                    flat.narrow(0, off, n).zero_()
        # (the flag slots were zeroed by the previous synchronize / at construction and set by _on_grad)
        self._handles.append((bi, dist.all_reduce(flat, group=self.group, async_op=True)))
        self._next = bi + 1

    def synchronize(self):
        """Wait for the outstanding all-reduces; write the averaged gradients back."""
        # EVERY bucket is reduced on EVERY rank once a backward has run (participation must not depend on which
        # parameters happened to receive a gradient on this rank: another rank may have produced one, and ranks
        # issuing different numbers of collectives hang).  Parameters without a gradient contribute zeros.
        while self._next < len(self.buckets):
            self._launch(self._next)
        for _, h in self._handles:
            h.wait()
        # per parameter: how many ranks produced a gradient -- the flag slots of ALL buckets in ONE device-to-host
        # transfer behind the waits (one .tolist() per bucket was one host synchronisation per bucket)
        flags = torch.cat([self.buckets[bi][0].narrow(0, self.buckets[bi][0].numel() - len(self.buckets[bi][1]),
                                                       len(self.buckets[bi][1])) for bi, _ in self._handles]).tolist() \
            if self._handles else []
        pos = 0
        for bi, _ in self._handles:
            flat, items = self.buckets[bi]
            total = flat.numel() - len(items)
            used = flags[pos:pos + len(items)]
            pos += len(items)
            flat.narrow(0, 0, total).div_(self.world)
            flat.narrow(0, total, len(items)).zero_()          # flag slots ready for the next backward
            for (p, off, n), cnt in zip(items, used):
                avg = flat.narrow(0, off, n).view_as(p)
                if p.grad is not None and p in self._fired:
                    p.grad.copy_(avg)
                elif cnt > 0:
                    # no local gradient, but another rank produced one: every rank must apply the SAME averaged
                    # gradient or the replicas diverge at the next optimizer step
                    if p.grad is not None:
                        p.grad.copy_(avg)
                    else:
                        p.grad = avg.clone()
                # else: no rank used this parameter in this backward -- .grad stays as it is (None after
                # zero_grad(set_to_none=True)), as in the reference and in a single-process run
        self._handles = []
        self._pending = [len(items) for _, items in self.buckets]
        self._queued = False
        self._next = 0
        self._fired = set()


def apply_gradient_allreduce(module, bucket_bytes=8 << 20, group=None):
    """Same entry point as the reference (distributed.py:94): returns `module`, whose backward now leaves
    rank-averaged gradients in `param.grad`.  The reducer is kept on the module as `_pdr_grad_allreduce`."""
    module._pdr_grad_allreduce = GradientAllReduce(module, bucket_bytes=bucket_bytes, group=group)
    return module
