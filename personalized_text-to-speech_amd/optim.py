"""AdamW on flat buffers: the two optimizer updates of the fine-tune step (reference finetune_speaker_v2.py:113-120, 213-214,
230-231) and the gradient norm it logs (commons.clip_grad_value_(params, None), commons.py:149-164) as a handful of launches
of csrc/adamw.hip instead of torch's per-tensor-list kernels.

`FlatAdamW` IS a torch.optim.AdamW as far as its surface goes (param_groups, state_dict() / load_state_dict() in torch's
format with `step` / `exp_avg` / `exp_avg_sq` per parameter, LR schedulers), so checkpoints written by the reference's
utils.save_checkpoint load into it and vice versa.  Underneath, every parameter is re-pointed at a view of ONE flat fp32
buffer, the moments are views of two more, and step() hands the kernel a by-value table of (gradient run, flat offset, n):
gradients that lie contiguously in memory in flat order — the weight arena's whole gradient buffer — are one table entry.
There is no fallback: the device must be a GPU and libvitsmi.so must load."""
import ctypes

import torch

from . import _lib


class FlatAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, runs=()):
        """runs: lists of parameters whose GRADIENTS are known to lie back to back in memory in that order (the weight arena's
        parameter order): they are laid out first and packed without padding, so that each run is one table entry."""
        params = list(params)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdamW: one parameter group")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW needs GPU parameters (there is no CPU path)")
        if any(p.dtype != torch.float32 or p.device != dev for p in params):
            raise ValueError("FlatAdamW: fp32 parameters on one device")
        known = {id(p) for p in params}
        order, seen, starts = [], set(), set()
        for run in runs:
            run = [p for p in run if id(p) in known and id(p) not in seen]
            if run:
                starts.add(id(run[0]))
            for p in run:
                seen.add(id(p)); order.append((p, True))
        order += [(p, False) for p in params if id(p) not in seen]
        self._order, self._off, off = [], {}, 0
        for p, packed in order:
            if not packed or id(p) in starts:
                off = (off + 3) & ~3                               # 16-byte aligned starts (vector accesses)
            self._order.append(p); self._off[id(p)] = off
            off += p.numel()
        total = (off + 3) & ~3
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.dev_state = torch.tensor([float(lr), 0.0], dtype=torch.float32, device=dev)      # [lr, completed steps]
        self._lr_host = float(lr)
        with torch.no_grad():
            for p in self._order:
                view = self._view(self.flat_p, p)
                view.copy_(p.data)
                p.data = view
        self._install_state()
        self._partials = torch.empty(sum((p.numel() + 4095) // 4096 for p in params), dtype=torch.float32, device=dev)
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.grad_norm = self._norm[0]           # L2 norm of the gradients the last step() saw (0-d view, overwritten in place)

    def _view(self, flat, p):
        o = self._off[id(p)]
        return flat[o:o + p.numel()].view(p.shape)

    def _install_state(self):
        for p in self._order:
            self.state[p] = {"step": self.dev_state[1], "exp_avg": self._view(self.flat_m, p), "exp_avg_sq": self._view(self.flat_v, p)}

    def state_dict(self):
        """torch's format.  `step` is written as one host tensor PER parameter (torch.optim.AdamW keeps it on the host unless
        capturable): handing out the shared device counter would make a plain torch AdamW that loads this dict increment one
        scalar once per parameter."""
        sd = super().state_dict()
        step = float(self.dev_state[1])
        sd["state"] = {k: dict(st, step=torch.tensor(step, dtype=torch.float32)) for k, st in sd["state"].items()}   # (copies of the live dicts)
        return sd

    def load_state_dict(self, state_dict):
        """torch's format (what the reference's checkpoints hold): moments are copied into the flat buffers."""
        super().load_state_dict(state_dict)
        loaded = {p: dict(st) for p, st in self.state.items()}
        self._install_state()
        steps = set()
        with torch.no_grad():
            for p, st in loaded.items():
                if "exp_avg" in st:
                    self.state[p]["exp_avg"].copy_(st["exp_avg"]); self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
                    steps.add(float(st["step"]))
            # one shared step counter: torch keeps one per parameter, and they only differ when some parameters were skipped
            # in some steps (no gradient); the bias corrections then follow the most-stepped parameter
            self.dev_state[1] = max(steps) if steps else 0.0
        self._sync_lr(force=True)

    def _sync_lr(self, force=False):
        lr = float(self.param_groups[0]["lr"])
        if force or lr != self._lr_host:
            self.dev_state[0:1].fill_(lr)
            self._lr_host = lr

    def _entries(self):
        """-> (ctypes array of vits_adamw_entry, count, tensors to keep alive until the launches are enqueued)"""
        raw, keep = [], []
        for p in self._order:
            g = p.grad
            if g is None:
                continue
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device:
                g = g.to(device=p.device, dtype=torch.float32).contiguous()
                keep.append(g)
            ptr, off, n = g.data_ptr(), self._off[id(p)], g.numel()
            if raw and raw[-1][0] + 4 * raw[-1][2] == ptr and raw[-1][1] + raw[-1][2] == off and raw[-1][2] + n < (1 << 31):
                raw[-1][2] += n
            else:
                raw.append([ptr, off, n])
        arr = (_lib.AdamwEntry * max(len(raw), 1))()
        for e, (ptr, off, n) in zip(arr, raw):
            e.g, e.offset, e.n = ptr, off, n
        return arr, len(raw), keep

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FlatAdamW.step takes no closure")
        self._sync_lr()
        arr, n, keep = self._entries()
        if n == 0:
            return None
        grp = self.param_groups[0]
        L, s = _lib.lib(), _lib.stream_ptr()
        blocks = L.vits_adamw_blocks(ctypes.addressof(arr), n)
        e0 = _lib.timer.start("vits_adamw")
        _lib.check(L.vits_adamw(self.flat_p.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(), ctypes.addressof(arr), n,
                                self.dev_state.data_ptr(), grp["betas"][0], grp["betas"][1], grp["eps"], grp["weight_decay"],
                                self._partials.data_ptr(), self._partials.numel(), s), "vits_adamw")
        _lib.check(L.vits_gradnorm_final(self._partials.data_ptr(), blocks, self._norm.data_ptr(), self.dev_state.data_ptr(), 1, s),
                   "vits_gradnorm_final")
        elems = sum(e.n for e in arr[:n])
        _lib.timer.stop("vits_adamw", e0, (0.0, 28.0 * elems))          # (flops, algorithmic bytes: g, p, m, v read; p, m, v written)
        self.last_entries = n
        del keep
        return None


def grad_norm_l2(grads):
    """Global L2 norm of a list of fp32 GPU gradient tensors as a 0-d tensor: one pass of csrc/adamw.hip in norm-only mode
    (commons.clip_grad_value_(..., None), reference commons.py:149-164)."""
    keep, raw = [], []
    for g in grads:
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.float().contiguous()
        keep.append(g)
        if raw and raw[-1][0] + 4 * raw[-1][1] == g.data_ptr() and raw[-1][1] + g.numel() < (1 << 31):
            raw[-1][1] += g.numel()
        elif g.numel():
            raw.append([g.data_ptr(), g.numel()])
    dev = keep[0].device
    arr = (_lib.AdamwEntry * max(len(raw), 1))()
    for e, (ptr, n) in zip(arr, raw):
        e.g, e.offset, e.n = ptr, 0, n
    L, s = _lib.lib(), _lib.stream_ptr()
    blocks = L.vits_adamw_blocks(ctypes.addressof(arr), len(raw))
    partials = torch.empty(max(blocks, 1), dtype=torch.float32, device=dev)
    out = torch.empty(1, dtype=torch.float32, device=dev)
    _lib.check(L.vits_adamw(None, None, None, ctypes.addressof(arr), len(raw), None, 0.0, 0.0, 0.0, 0.0, partials.data_ptr(), blocks, s), "vits_adamw")
    _lib.check(L.vits_gradnorm_final(partials.data_ptr(), blocks, out.data_ptr(), None, 0, s), "vits_gradnorm_final")
    return out[0]
