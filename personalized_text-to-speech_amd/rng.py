"""Noise source of the model.  The reference draws noise inline with torch.randn_like /
torch.randn / torch.rand (models.py:67,90,240,520; commons.py:65).  Routing those draws through
this object keeps the default behaviour (torch's generator on the tensor's device) while letting
parity tests replay the exact tensors the reference drew (tests/golden/*.npz `noise*`)."""
import torch


class _Noise:
    def __init__(self):
        self._replay = None

    def replay(self, tensors):
        """Context manager: serve `tensors` (in order) instead of drawing fresh noise."""
        src = self

        class _Ctx:
            def __enter__(self_inner):
                src._replay = list(tensors)
                return src

            def __exit__(self_inner, *exc):
                left = len(src._replay)
                src._replay = None
                if exc[0] is None and left:
                    raise RuntimeError(f"{left} replayed noise tensor(s) were never consumed")

        return _Ctx()

    def _next(self, shape, device, dtype):
        t = self._replay.pop(0)
        if tuple(t.shape) != tuple(shape):
            raise RuntimeError(f"replayed noise has shape {tuple(t.shape)}, the model asked for {tuple(shape)}")
        return t.to(device=device, dtype=dtype)

    def randn_like(self, x):
        if self._replay is not None:
            return self._next(x.shape, x.device, x.dtype)
        return torch.randn_like(x)

    def randn(self, *shape, device=None, dtype=None):
        if self._replay is not None:
            return self._next(shape, device, dtype)
        return torch.randn(*shape, device=device, dtype=dtype)

    def rand(self, *shape, device=None, dtype=None):
        if self._replay is not None:
            return self._next(shape, device, dtype)
        return torch.rand(*shape, device=device, dtype=dtype)


noise = _Noise()
