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
    staged = dist.get_backend() != "nccl"            # gloo rehearsal with device tensors: through a host copy
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if staged and t.is_cuda:
                h = t.data.cpu()
                dist.broadcast(h, src)
                t.data.copy_(h)
            else:
                dist.broadcast(t.data, src)


class GradBuckets:
    """Flat gradient buckets + overlapped all-reduce for one network.

    Gradients are produced by autograd as it likes (param.grad starts each step as None, so the
    engine installs the incoming tensor without an extra add kernel per parameter — ~1000 launches
    per step for this model).  A post-accumulate-grad hook counts a bucket's parameters; when the
    last one has arrived the bucket's gradients are packed into its flat buffer by ONE multi-tensor
    copy, param.grad is re-pointed at the flat views, and the bucket's all-reduce starts on the
    communication stream while backward continues.  With a single process there is nothing to
    exchange and the buckets are never packed."""

    def __init__(self, params, bucket_bytes=64 << 20, group=None, force=False):
        """force: build the buckets and run the pack / all-reduce path even with ONE rank (an initialised process group is
        still required): the averaging is then the identity, but every RCCL call of the N > 1 path executes — how a one-GPU
        box exercises and times it (tests/test_rccl_single_gpu.py, bench.py --exchange)."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = world()
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())
        self.buckets = []          # (flat tensor, [params], [views])
        self._bucket_of = {}
        if self.active:
            cur, cur_bytes = [], 0
            for p in reversed(self.params):      # reverse registration order ~ order backward produces gradients
                nbytes = p.numel() * p.element_size()
                if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype):
                    self._close(cur)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nbytes
            if cur:
                self._close(cur)
            self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._pending = [0] * len(self.buckets)
        self._work = []
        self._launched = [False] * len(self.buckets)
        self._enabled = True
        self._manual = False

    def _close(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
        views, off = [], 0
        for p in plist:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
            self._bucket_of[p] = len(self.buckets)
        self.buckets.append((flat, plist, views))

    def zero_grad(self):
        """Replaces optimizer.zero_grad(): drop the gradients (set_to_none) and re-arm the buckets."""
        for p in self.params:
            p.grad = None
        for i, (_, plist, _) in enumerate(self.buckets):
            self._pending[i] = len(plist)
            self._launched[i] = False
        self._work = []

    def _launch(self, i):
        if self._launched[i]:
            return
        self._launched[i] = True
        flat, plist, views = self.buckets[i]
        have = [(v, p.grad) for v, p in zip(views, plist) if p.grad is not None]
        missing = [v for v, p in zip(views, plist) if p.grad is None]
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(views, plist):
            p.grad = v
        if dist.get_backend(self.group) == "nccl":
            self._work.append(dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        elif flat.is_cuda:
            self._host_all_reduce(flat)
        else:
            self._work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat))

    # ---- manual mode (captured step): no hooks; pack() inside the graph, all_reduce() eagerly between graph replays
    def manual(self, flag):
        self._manual = flag

    INPLACE_MIN = 1 << 20        # bytes: a run of gradients that already lie back to back is exchanged where it is from this size on

    def pack(self):
        """Manual mode's gather step, graph-capturable (no collective).  Gradients that already lie back to back in one
        allocation — the weight arenas write all of a network's convolution gradients into one flat buffer — are exchanged in
        place: each such run of at least INPLACE_MIN bytes becomes a flat alias.  The rest (biases, norms, embeddings) is copied
        into one flat leftover buffer by ONE multi-tensor copy and param.grad is pointed at its views.  Leaves
        self._exchange = the flat tensors all_reduce() averages."""
        self._exchange = []
        if not self.active:
            return
        items, missing = [], []
        for p in self.params:
            g = p.grad
            if g is None:
                missing.append(p)
            elif g.dtype == p.dtype and g.is_contiguous():
                items.append((g.untyped_storage().data_ptr(), g.data_ptr(), g.numel() * g.element_size(), p))
            else:
                missing.append(p)                       # (copied below: not a layout we can alias)
        items.sort(key=lambda it: (it[0], it[1]))
        runs, cur = [], []
        for it in items:
            if cur and (cur[-1][0] != it[0] or cur[-1][1] + cur[-1][2] != it[1] or cur[-1][3].dtype != it[3].dtype):
                runs.append(cur); cur = []
            cur.append(it)
        if cur:
            runs.append(cur)
        # (addresses differ from rank to rank: the exchange order must not — regions go in the order of their first parameter)
        order = {id(p): i for i, p in enumerate(self.params)}
        runs.sort(key=lambda run: min(order[id(it[3])] for it in run))
        exchange, rest = [], list(missing)
        for run in runs:
            nbytes = sum(it[2] for it in run)
            if nbytes >= self.INPLACE_MIN:
                g0 = run[0][3].grad
                n = nbytes // g0.element_size()
                exchange.append(torch.empty(0, dtype=g0.dtype, device=g0.device).set_(g0.untyped_storage(), g0.storage_offset(), (n,)))
            else:
                rest += [it[3] for it in run]
        if rest:
            rest.sort(key=lambda p: order[id(p)])
            layout = tuple((id(p), p.numel()) for p in rest)
            if getattr(self, "_rest_layout", None) != layout:
                p0 = rest[0]
                if any(p.dtype != p0.dtype for p in rest):
                    raise RuntimeError("GradBuckets.pack: parameters of one network must share a dtype")
                self._rest_flat = torch.zeros(sum(p.numel() for p in rest), dtype=p0.dtype, device=p0.device)
                self._rest_views, off = [], 0
                for p in rest:
                    self._rest_views.append(self._rest_flat[off:off + p.numel()].view_as(p)); off += p.numel()
                self._rest_layout = layout
            have = [(v, p.grad) for v, p in zip(self._rest_views, rest) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
            none = [v for v, p in zip(self._rest_views, rest) if p.grad is None]
            if none:
                torch._foreach_zero_(none)
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            for v, p in zip(self._rest_views, rest):
                p.grad = v
            exchange.append(self._rest_flat)
        self._exchange = exchange

    def all_reduce(self):
        """Blocking (stream-ordered) average over the ranks of what pack() gathered."""
        for flat in self._exchange:
            if dist.get_backend(self.group) == "nccl":
                dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
            elif flat.is_cuda:
                self._host_all_reduce(flat)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                flat.div_(self.world)

    def _host_all_reduce(self, flat):
        """Rehearsal path (gloo with device tensors, e.g. several ranks sharing one GPU in a test): average through a host
        copy, synchronously.  The production backend is "nccl" (RCCL over xGMI)."""
        host = flat.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
        flat.copy_(host.div_(self.world))

    def _on_grad(self, p):
        if not self._enabled or self._manual:
            return
        i = self._bucket_of[p]
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def finish(self):
        """Launch whatever did not complete (parameters without a gradient this step) and wait."""
        if self._manual:
            return
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
