"""Data-parallel gradient exchange for the fine-tune step (replaces the two
DistributedDataParallel wrappers of reference finetune_speaker_v2.py:144-145).

One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).
Design for the MI355X node: xGMI is point-to-point, so few LARGE collectives
beat DDP's 25 MiB default — gradients live in a handful of flat, contiguous buckets (param.grad
tensors are views into them, no copy-in/copy-out), and each bucket's all-reduce is launched from
a post-accumulate-grad hook as soon as its last gradient has been produced, i.e. overlapped with
the rest of the backward.  Averages like DDP (sum / world).
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_parameters(module, src=0):
    """DDP's constructor-time broadcast (rank 0 -> all) of parameters and buffers."""
    if world() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


class GradBuckets:
    """Flat gradient storage + overlapped all-reduce for one network."""

    def __init__(self, params, bucket_bytes=64 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = world()
        # buckets are filled in reverse registration order ~ the order backward produces gradients
        self.buckets = []          # (flat tensor, [params])
        self._bucket_of = {}
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._pending = [0] * len(self.buckets)
        self._work = []
        self._launched = [False] * len(self.buckets)
        self._enabled = True
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _close(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
        off = 0
        for p in plist:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self._bucket_of[p] = len(self.buckets)
        self.buckets.append((flat, plist))

    def zero_grad(self):
        """One memset per bucket (replaces optimizer.zero_grad(); grads stay views of the buckets)."""
        for i, (flat, plist) in enumerate(self.buckets):
            flat.zero_()
            self._pending[i] = len(plist)
            self._launched[i] = False
            for p in plist:                      # autograd may have replaced .grad (e.g. set_to_none)
                if p.grad is None or p.grad.data_ptr() < flat.data_ptr() or p.grad.data_ptr() >= flat.data_ptr() + flat.numel() * flat.element_size():
                    self._rebind(i)
                    break
        self._work = []

    def _rebind(self, i):
        flat, plist = self.buckets[i]
        off = 0
        for p in plist:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def _launch(self, i):
        if self._launched[i]:
            return
        self._launched[i] = True
        if self.world == 1:
            return
        flat = self.buckets[i][0]
        if dist.get_backend(self.group) == "nccl":
            self._work.append(dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        else:
            self._work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat))

    def _on_grad(self, p):
        if not self._enabled:
            return
        i = self._bucket_of[p]
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def finish(self):
        """Launch whatever did not complete (parameters without a gradient this step) and wait."""
        for i in range(len(self.buckets)):
            self._launch(i)
        for w in self._work:
            if isinstance(w, tuple):
                w[0].wait()
                w[1].div_(self.world)
            else:
                w.wait()
        self._work = []

    def enabled(self, flag):
        self._enabled = flag
