"""Weight arena: the fp32 master parameters of a whole group of convolutions are turned into the HIP
kernels' operands by ONE launch per forward (csrc/weight_prep.hip: weight-norm, tap-major layout,
padding, compute dtype, plus the tap-reversed transposed copy the data-gradient calls need), and
their weight gradients are mapped back to parameter gradients by ONE launch per backward.

Autograd sees one node (`PrepFn`) whose outputs are fp32 *handles*, one per convolution (slice); the
fused layer nodes (decoder_cl.DecoderFn, wn_cl.WNFn, wn_cl.ConvCLFn) take a handle as their weight
input, resolve it here to the low-precision operands, write their fp32 weight gradient straight
into the arena's `dw` region and return that view as the handle's gradient — so nothing is copied
and the parameter gradients (weight_v, weight_g, weight) come out of a single kernel.

Scope: `with weight_arena.scope(synthesizer): ...` prepares every registered convolution of the
generator's posterior encoder, flow and decoder; outside a scope the layer nodes fall back to
per-layer torch preparation (same numerics, many more launches).
"""
import ctypes

import torch

from . import _lib

_DT = {torch.float32: 0, torch.bfloat16: 2}
_current = None            # dict (id(module), part) -> handle tensor, valid inside a scope
_registry = {}             # handle.data_ptr() -> (arena, index)
_bias_registry = {}        # bias handle.data_ptr() -> (arena, bias index)


class Spec:
    def __init__(self, module, part=None, row_lo=0, n_rows=None, c_out_p=None, c_in_p=None, transpose=False, torch_layout=False, groups=1,
                 bias=False):
        """bias=True: the module's bias is managed by the arena too (padded to c_out_p, fp32): wn_cl.bias_of() then hands the layer
        node an arena bias handle, and the bias gradient reaches the parameter through PrepFn.backward — i.e. AFTER the deferred
        second stages of the weight-gradient launches (which also produce the bias gradients) have run."""
        self.module, self.part, self.transpose, self.torch_layout, self.groups = module, part, transpose, torch_layout, groups
        self.bias = module.bias if (bias and getattr(module, "bias", None) is not None) else None
        assert self.bias is None or (row_lo == 0 and n_rows is None and not transpose)
        has_g = hasattr(module, "weight_g")
        self.v = module.weight_v if has_g else module.weight
        self.g = module.weight_g if has_g else None
        d0, d1, k = self.v.shape[:3]              # Conv2d (k, 1) weights [c_out, c_in, k, 1] are read as [c_out, c_in, k]
        self.k = k
        if transpose:                         # ConvTranspose1d [c_in][c_out][k], rows = input channels
            self.c_in, self.c_out = d0, d1
            self.row_lo, self.n_rows = 0, d0
            self.c_in_p, self.c_out_p = c_in_p or d0, d1
            self.numel = k * d1 * self.c_in_p
            self.fwd_shape, self.bwd_shape = (1, k * d1, self.c_in_p), (1, self.c_in_p, k * d1)
        else:                                 # Conv1d [c_out][c_in][k], rows = output channels
            self.c_in = d1
            self.row_lo = row_lo
            self.n_rows = self.c_out = (d0 - row_lo) if n_rows is None else n_rows
            self.c_out_p, self.c_in_p = c_out_p or self.c_out, c_in_p or d1
            self.numel = k * self.c_out_p * self.c_in_p
            self.fwd_shape, self.bwd_shape = (k, self.c_out_p, self.c_in_p), (k, self.c_in_p, self.c_out_p)
            if torch_layout:                  # operand = weight-normed parameter in its own layout (library convolutions)
                assert row_lo == 0 and n_rows is None and not c_out_p and not c_in_p
                self.fwd_shape = self.bwd_shape = tuple(self.v.shape)
            if groups > 1:                    # grouped Conv1d [c_out][c_in/groups][k] -> dense block-diagonal operands, compact dw
                assert row_lo == 0 and n_rows is None and not c_out_p and not c_in_p and not torch_layout
                self.c_in = self.c_in_p = d1 * groups
                self.numel = k * d0 * self.c_in
                self.fwd_shape, self.bwd_shape = (k, d0, self.c_in), (k, self.c_in, d0)
        self.dw_shape = (self.k, self.c_out, self.c_in // groups) if groups > 1 else self.fwd_shape


def param_order(specs):
    """The parameters behind `specs` in the order their gradients lie in an arena's flat gradient buffer (optim.FlatAdamW lays
    its flat parameter buffer out the same way, so the whole buffer is ONE run of its update kernel)."""
    params, seen = [], set()
    for s in specs:
        for p in (s.v, s.g):
            if p is not None and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
    return params


class WeightArena:
    def __init__(self, specs, dtype):
        self.specs, self.dtype = specs, dtype
        dev = specs[0].v.device
        self.params = param_order(specs)
        pidx = {id(p): i for i, p in enumerate(self.params)}
        # parameter-gradient arena: one region per parameter, torch layout
        self.p_off, off = [], 0
        for p in self.params:
            self.p_off.append(off)
            off += p.numel()
        self.dparam = torch.zeros(off, dtype=torch.float32, device=dev)
        self.dparam_views = [self.dparam[o:o + p.numel()].view_as(p) for o, p in zip(self.p_off, self.params)]
        # operand arenas
        offs, off, row0 = [], 0, 0
        tiles = []
        ents = (_lib.PrepEntry * len(specs))()
        for i, s in enumerate(specs):
            offs.append(off)
            e = ents[i]
            e.v, e.g = s.v.data_ptr(), (s.g.data_ptr() if s.g is not None else None)
            e.off, e.off_dv = off, self.p_off[pidx[id(s.v)]]
            e.off_dg = self.p_off[pidx[id(s.g)]] if s.g is not None else 0
            e.layout, e.c_out, e.c_in, e.k = (3 if s.groups > 1 else (2 if s.torch_layout else (1 if s.transpose else 4))), s.c_out, s.c_in, s.k
            if e.layout == 4:                                   # transposed operand by the tiled LDS transpose (coalesced)
                tiles += [(i, tap, co0, ci0) for tap in range(s.k) for co0 in range(0, s.c_out_p, 64) for ci0 in range(0, s.c_in_p, 64)]
            e.groups = s.groups
            e.c_out_p, e.c_in_p, e.row_lo, e.n_rows, e.row0 = s.c_out_p, s.c_in_p, s.row_lo, s.n_rows, row0
            row0 += s.n_rows
            off += (s.numel + 63) & ~63                         # keep every operand 128-byte aligned
        self.total_rows, self.n = row0, len(specs)
        self.table = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(dev)
        self.n_tiles = len(tiles)
        self.tiles = torch.tensor(tiles, dtype=torch.int32, device=dev) if tiles else None
        self.w_fwd = torch.zeros(off, dtype=dtype, device=dev)
        self.w_bwd = torch.zeros(off, dtype=dtype, device=dev)
        self.handle = self.w_fwd if dtype == torch.float32 else torch.empty(off, dtype=torch.float32, device=dev)
        self.dw = torch.zeros(off, dtype=torch.float32, device=dev)
        view = lambda buf, o, s, shape: buf[o:o + s.numel].view(shape)
        self.fwd = [view(self.w_fwd, o, s, s.fwd_shape) for o, s in zip(offs, specs)]
        self.bwd = [view(self.w_bwd, o, s, s.bwd_shape) for o, s in zip(offs, specs)]
        # torch-layout operands are consumed by autograd-aware library ops: their handle is the operand itself
        # (grouped specs: the handle has the shape of the compact weight gradient the layer node returns for it)
        cview = lambda buf, o, s: buf[o:o + s.dw_shape[0] * s.dw_shape[1] * s.dw_shape[2]].view(s.dw_shape)
        self.handles = [cview(self.handle, o, s) if s.groups > 1 else view(self.w_fwd if s.torch_layout else self.handle, o, s, s.fwd_shape)
                        for o, s in zip(offs, specs)]
        self.dws = [cview(self.dw, o, s) if s.groups > 1 else view(self.dw, o, s, s.fwd_shape) for o, s in zip(offs, specs)]
        # arena-managed biases: fp32, padded to c_out_p, one flat buffer + one flat gradient buffer
        self.bias_specs = [i for i, s in enumerate(specs) if s.bias is not None]
        self.bias_params = [specs[i].bias for i in self.bias_specs]
        self.b_off, boff = [], 0
        for i in self.bias_specs:
            self.b_off.append(boff)
            boff += (specs[i].c_out_p + 7) & ~7
        self.b_flat = torch.zeros(max(boff, 8), dtype=torch.float32, device=dev)
        self.db_flat = torch.zeros(max(boff, 8), dtype=torch.float32, device=dev)
        self.bias_handles = [self.b_flat[o:o + specs[i].c_out_p] for o, i in zip(self.b_off, self.bias_specs)]
        self.bias_live = [self.b_flat[o:o + specs[i].c_out] for o, i in zip(self.b_off, self.bias_specs)]
        self.db_views = [self.db_flat[o:o + specs[i].c_out_p] for o, i in zip(self.b_off, self.bias_specs)]
        for j, h in enumerate(self.bias_handles):
            _bias_registry[h.data_ptr()] = (self, j)
        self.ptrs = [p.data_ptr() for p in self.params + self.bias_params]
        # Every forward overwrites the shared operand / gradient buffers.  `gen` counts forwards, `claimed` the weight-gradient
        # slots written since the last forward: PrepFn.backward refuses to hand out gradients computed from operands a later
        # forward has overwritten, and a slot written twice in one backward (a module used twice in one graph) is refused too.
        self.gen, self.claimed = 0, {}
        for i, h in enumerate(self.handles):
            _registry[h.data_ptr()] = (self, i)

    @property
    def defer(self):
        """The collector of deferred weight-gradient second stages for the CURRENT stream (a layer node resolves its weights —
        and with them this collector — in its forward, and its backward runs on the same stream): a sub-graph that runs on a
        side stream (kernels.SideBranch) gets its own collector, flushed on that stream by flush_deferred()."""
        if not self.w_fwd.is_cuda:
            return None
        from . import kernels
        cur = torch.cuda.current_stream(self.w_fwd.device)
        defers = self.__dict__.setdefault("_defers", {})
        hit = defers.get(cur.cuda_stream)
        if hit is None:
            hit = defers[cur.cuda_stream] = (kernels.DeferredReductions(self.w_fwd.device), cur)
        return hit[0]

    def flush_deferred(self):
        """Run every collector's pending second stages, each on the stream its launches ran on; the current stream then waits
        for the others (inside a capture: one join edge)."""
        if not self.w_fwd.is_cuda:
            return
        cur = torch.cuda.current_stream(self.w_fwd.device)
        for key, (d, stream) in list(self.__dict__.get("_defers", {}).items()):
            if not d.pending and not d.batch:
                continue
            if key == cur.cuda_stream:
                d.flush()
            else:
                with torch.cuda.stream(stream):
                    d.flush()
                cur.wait_stream(stream)

    def stale(self):
        """Parameters re-allocated (e.g. .to(device)) or replaced (remove_weight_norm folds weight_g / weight_v into `weight`)."""
        if any(p.data_ptr() != q for p, q in zip(self.params + self.bias_params, self.ptrs)):
            return True
        return any((s.g is not None) != ("weight_g" in s.module._parameters) for s in self.specs)

    def prepare(self):
        """-> (weight handles, bias handles)"""
        out = PrepFn.apply(self, *self.params, *self.bias_params)
        return out[:self.n], out[self.n:]


class PrepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, arena, *params):
        arena.flush_deferred()                       # leftovers of a backward that never reached this node
        arena.gen += 1
        arena.claimed.clear()
        ctx.gen = arena.gen
        rc = _lib.lib().vits_weight_prep(arena.table.data_ptr(), arena.n, arena.total_rows, _DT[arena.dtype],
                                         arena.w_fwd.data_ptr(), arena.w_bwd.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "vits_weight_prep")
        if arena.n_tiles:
            rc = _lib.lib().vits_weight_prep_transpose(arena.tiles.data_ptr(), arena.n_tiles, arena.table.data_ptr(), _DT[arena.dtype],
                                                       arena.w_fwd.data_ptr(), arena.w_bwd.data_ptr(), _lib.stream_ptr())
            _lib.check(rc, "vits_weight_prep_transpose")
        if arena.bias_params:
            torch._foreach_copy_(arena.bias_live, [b.detach() for b in arena.bias_params])
        ctx.arena = arena
        return tuple(h.detach() for h in arena.handles) + tuple(h.detach() for h in arena.bias_handles)

    @staticmethod
    def backward(ctx, *grads):
        arena = ctx.arena
        dws, dbs = grads[:arena.n], grads[arena.n:]
        if ctx.gen != arena.gen:
            raise RuntimeError(
                "weight_arena: this backward belongs to forward #%d, but forward #%d has since overwritten the arena's shared "
                "operands and gradient buffers (two forwards of one network before a backward, e.g. loss(net(a)) + loss(net(b))): "
                "run each forward's backward before the next forward, or batch the inputs" % (ctx.gen, arena.gen))
        arena.flush_deferred()                       # the deferred second stages of this network's weight-gradient launches
        arena.claimed.clear()
        # gradient accumulation (no zero_grad between two backwards): a param.grad that still aliases `dparam` would be
        # overwritten below and then added to itself by autograd — detach it into its own storage first
        for params, buf in ((arena.params, arena.dparam), (arena.bias_params, arena.db_flat)):
            base = buf.untyped_storage().data_ptr()
            for p in params:
                if p.grad is not None and p.grad.untyped_storage().data_ptr() == base:
                    p.grad = p.grad.clone()
        for j, d in enumerate(dbs):
            if d is None:
                arena.db_views[j].zero_()
            elif d.data_ptr() != arena.db_views[j].data_ptr():
                arena.db_views[j].copy_(d.reshape(arena.db_views[j].shape))
        for i, d in enumerate(dws):
            if d is None:
                arena.dws[i].zero_()
            elif d.data_ptr() != arena.dws[i].data_ptr():
                arena.dws[i].copy_(d.reshape(arena.dws[i].shape))
        rc = _lib.lib().vits_weight_prep_bwd(arena.table.data_ptr(), arena.n, arena.total_rows, arena.dw.data_ptr(),
                                             arena.dparam.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "vits_weight_prep_bwd")
        # FRESH views: autograd's AccumulateGrad installs an incoming gradient as param.grad without a copy only when
        # nobody else references the tensor object; handing out the cached views costs one copy kernel per parameter
        return (None, *[arena.dparam[o:o + p.numel()].view_as(p) for o, p in zip(arena.p_off, arena.params)],
                *[arena.db_flat[o:o + p.numel()].view_as(p) for o, p in zip(arena.b_off, arena.bias_params)])


class Resolved:
    """What a layer node needs for one convolution weight (`defer`: the arena's collector of weight-gradient second stages,
    run in one launch right before the arena maps the weight gradients back to parameter gradients)."""
    __slots__ = ("fwd", "bwd", "dw", "defer", "arena", "index")

    def __init__(self, fwd, bwd, dw, defer=None, arena=None, index=-1):
        self.fwd, self.bwd, self.dw, self.defer, self.arena, self.index = fwd, bwd, dw, defer, arena, index

    def claim_dw(self, owner=None):
        """The arena's weight-gradient view for this convolution (None outside an arena), claimed by the autograd node `owner`
        for the current backward: a claim by a DIFFERENT node before the arena's own backward has run means the same weight
        is used twice in one graph, and autograd would sum the one shared view with itself.  (The same node claiming again is
        a repeated backward over a retained graph: fine, it rewrites the same values.)"""
        if self.arena is not None:
            prev = self.arena.claimed.get(self.index)
            if prev is not None and prev != id(owner):
                raise RuntimeError("weight_arena: a convolution weight is used twice in one autograd graph; the arena keeps one "
                                   "gradient buffer per weight (call the module once per forward, or outside weight_arena.scope)")
            self.arena.claimed[self.index] = id(owner)
        return self.dw


_constants = {}            # data_ptr -> Resolved, for constant operands registered with register_constant()


def register_constant(w_fwd):
    """A constant (non-trainable) kernel-layout operand, e.g. a DFT basis: its data-gradient copy is made once."""
    r = Resolved(w_fwd, w_fwd.flip(0).transpose(1, 2).contiguous(), None)
    _constants[w_fwd.data_ptr()] = r
    return w_fwd


def resolve(w, dtype):
    """w: a handle from an arena, a registered constant, or any fp32 kernel-layout weight [k][c_out][c_in] (fallback)."""
    c = _constants.get(w.data_ptr())
    if c is not None and c.fwd.dtype == dtype and c.fwd.shape == w.shape:
        return c
    hit = _registry.get(w.data_ptr())
    if hit is not None and hit[0].dtype == dtype and tuple(w.shape) == tuple(hit[0].handles[hit[1]].shape):
        a, i = hit
        return Resolved(a.fwd[i], a.bwd[i], a.dws[i], a.defer, a, i)
    wd = w.detach().to(dtype)
    return Resolved(wd, None, None)


def bwd_operand(res):
    """Tap-reversed transposed operand for the data-gradient call."""
    if res.bwd is not None:
        return res.bwd
    return res.fwd.flip(0).transpose(1, 2).contiguous()


def handle_for(module, part=None):
    """Inside a scope: the prepared handle of `module` (or of one of its row slices); else None."""
    if _current is None:
        return None
    return _current.get((id(module), part))


def bias_handle_for(module):
    """Inside a scope: the arena-managed (padded, fp32) bias of `module`, if its Spec asked for one; else None."""
    if _current is None:
        return None
    return _current.get((id(module), "__bias__"))


def bias_slot(bias):
    """(arena, index) if `bias` is an arena bias handle, else None."""
    return _bias_registry.get(bias.data_ptr()) if bias is not None else None


class scope:
    """Prepare all registered convolutions of `root` for the duration of a forward pass."""

    def __init__(self, root, collect):
        self.root, self.collect = root, collect

    def __enter__(self):
        global _current
        dtype = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
        cache = self.root.__dict__.setdefault("_weight_arenas", {})
        arena = cache.get(dtype)
        if arena is None or arena.stale():
            arena = WeightArena(self.collect(self.root), dtype)
            cache[dtype] = arena
        handles, bias_handles = arena.prepare()
        self.prev = _current
        _current = {(id(s.module), s.part): h for s, h in zip(arena.specs, handles)}
        for i, h in zip(arena.bias_specs, bias_handles):
            _current[(id(arena.specs[i].module), "__bias__")] = h
        return arena

    def __exit__(self, *exc):
        global _current
        _current = self.prev
        return False
