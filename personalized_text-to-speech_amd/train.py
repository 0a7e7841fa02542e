"""The fine-tune step — mirror of reference finetune_speaker_v2.py:174-232 (one iteration of
train_and_evaluate's hot loop): G forward, mel targets, D step, G step, two AdamW updates.

Differences from the reference, none of which changes a result:
  * bf16 autocast instead of fp16 autocast + GradScaler (no loss scaling needed); `amp=False`
    runs everything in fp32 (the parity mode);
  * gradients are reduced through distributed.GradBuckets (overlapped flat all-reduce) instead
    of two DDP wrappers, and D's parameters do not take gradients during the G step (the
    reference computes and all-reduces them there only to zero them at the next iteration,
    finetune_speaker_v2.py:210,218,228);
  * AdamW and the logged gradient norm are one multi-tensor HIP pass over flat buffers (optim.FlatAdamW); grad norms are
    0-d tensors, discriminator_loss keeps tensors: no host synchronisation inside the step.
"""
import contextlib

import torch
from torch.nn import functional as F

from . import commons, weight_arena
from . import kernels as K
from .optim import FlatAdamW
from .distributed import GradBuckets, broadcast_parameters
from .losses import discriminator_loss, feature_loss, generator_loss, kl_loss
from .mel_processing import mel_spectrogram_torch, spec_to_mel_torch
from .models import MultiPeriodDiscriminator, SynthesizerTrn


class FineTuner:
    def __init__(self, hps, device, amp=True, bucket_bytes=64 << 20, discriminator_seed=None, force_exchange=False):
        self.hps, self.device, self.amp = hps, torch.device(device), amp
        torch.manual_seed(hps.train.seed)                       # finetune_speaker_v2.py:70
        m = {k: v for k, v in hps.model.items()}
        self.net_g = SynthesizerTrn(hps.n_symbols, hps.data.filter_length // 2 + 1,
                                    hps.train.segment_size // hps.data.hop_length,
                                    n_speakers=hps.data.n_speakers, **m).to(self.device)
        if discriminator_seed is not None:                      # parity tests: a discriminator reproducible on its own
            torch.manual_seed(discriminator_seed)
        self.net_d = MultiPeriodDiscriminator(hps.model.use_spectral_norm).to(self.device)
        broadcast_parameters(self.net_g)
        broadcast_parameters(self.net_d)
        # AdamW + gradient norm as one multi-tensor HIP pass over flat buffers (optim.py, csrc/adamw.hip); the parameters the
        # weight arenas manage are laid out in the arenas' gradient order, so each arena's gradient buffer is one run
        kw = dict(betas=tuple(hps.train.betas), eps=hps.train.eps)
        self.optim_g = FlatAdamW(self.net_g.parameters(), hps.train.learning_rate,
                                 runs=[weight_arena.param_order(SynthesizerTrn._arena_specs(self.net_g))], **kw)
        self.optim_d = FlatAdamW(self.net_d.parameters(), hps.train.learning_rate,
                                 runs=[weight_arena.param_order(MultiPeriodDiscriminator._arena_specs(self.net_d))], **kw)
        self._graph = None
        # Sub-graphs run as side-stream branches (kernels.SideBranch): "enc_p" = the text encoder next to the posterior encoder +
        # flow, "dp" = the stochastic duration predictor next to the decoder / discriminators, forward and backward
        # (34.5 -> 27.3 ms/step together; replays bitwise reproducible — DESIGN.md §6b tells how the "dp" branch exposed the
        # spline kernel's irreproducibility under concurrency and what cured it).  Also available, off: "mel" (slower).
        # "prior" (with both of the above): alignment scores, alignment search and the expansion of the prior join the duration
        # predictor on the side stream, next to the decoder (19.85 -> 19.67 ms/step, A/B on one box).
        self.side_branches = frozenset(("enc_p", "dp", "prior"))
        # One stream for everything that is ever captured (warm-up steps, captures, the bucket hooks' registration): autograd
        # runs a parameter's AccumulateGrad on the stream that was current when that node was CREATED and a post-accumulate
        # hook pins the node, and a capture must not have to synchronise with the legacy default stream (step_padded).
        self._capture_stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self._graph_pool = None
        with self.on_capture_stream():
            self.buckets_g = GradBuckets(self.net_g.parameters(), bucket_bytes, force=force_exchange)
            self.buckets_d = GradBuckets(self.net_d.parameters(), bucket_bytes, force=force_exchange)
        self.sched_g = torch.optim.lr_scheduler.ExponentialLR(self.optim_g, gamma=hps.train.lr_decay)
        self.sched_d = torch.optim.lr_scheduler.ExponentialLR(self.optim_d, gamma=hps.train.lr_decay)
        self.net_g.train()
        self.net_d.train()

    @contextlib.contextmanager
    def on_capture_stream(self):
        """Run the body on the tuner's capture stream, ordered after what the current stream holds and before what it does next
        (eager warm-up steps of a run that will be captured belong here; a no-op on the CPU)."""
        cs = self._capture_stream
        if cs is None:
            yield
            return
        cur = torch.cuda.current_stream(self.device)
        cs.wait_stream(cur)
        with torch.cuda.stream(cs):
            yield
        cur.wait_stream(cs)

    @staticmethod
    def _capture_mode():
        """Error mode of a graph capture.  With an initialised process group RCCL's watchdog thread polls events all the time;
        under the default "global" mode its hipEventQuery during our capture is an illegal call that takes the process down
        ("operation not permitted when stream is capturing" — found by tests/test_rccl_single_gpu.py, the first time the nccl
        branches ran).  "thread_local" confines the check to the capturing thread."""
        import torch.distributed as dist
        return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"

    def _autocast(self):
        if self.amp and self.device.type == "cuda":
            return torch.autocast("cuda", dtype=torch.bfloat16)
        return contextlib.nullcontext()

    def step(self, batch):
        """batch = (x, x_lengths, spec, spec_lengths, y, y_lengths, speakers), already on the device.
        Three phases separated by the two points where data-parallel ranks must have exchanged gradients."""
        manual = self.buckets_d._manual                      # an eager step next to a captured one exchanges gradients the eager way
        self.buckets_d.manual(False); self.buckets_g.manual(False)
        try:
            self._phase_a(batch)
            self.buckets_d.finish()
            self._phase_b()
            self.buckets_g.finish()
            return self._phase_c()
        finally:
            self.buckets_d.manual(manual); self.buckets_g.manual(manual)

    def _phase_a(self, batch):
        """Generator forward, mel targets, discriminator forward on (y, y_hat.detach()) and the backward of its loss."""
        hps = self.hps
        x, x_lengths, spec, spec_lengths, y, y_lengths, speakers = batch
        seg_frames = hps.train.segment_size // hps.data.hop_length
        # side-stream branch of the duration predictor: not together with hook-mode bucket all-reduces (their packing copies
        # would run on whichever stream a gradient arrives on)
        branches = self.side_branches if (not self.buckets_g.active or self.buckets_g._manual) else frozenset()
        self.net_g.side_branches = branches

        with self._autocast():
            y_hat, l_length, attn, ids_slice, x_mask, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q) = \
                self.net_g(x, x_lengths, spec, spec_lengths, speakers)
            mel = spec_to_mel_torch(spec.float(), hps.data.filter_length, hps.data.n_mel_channels, hps.data.sampling_rate,
                                    hps.data.mel_fmin, hps.data.mel_fmax)
            y_mel = commons.slice_segments(mel, ids_slice, seg_frames)
            # mel of the generated waveform as a side branch: its backward (generator step) then overlaps the discriminators'
            mel_branch = K.SideBranch(y_hat.device, y_hat, lane=2) if ("mel" in branches and y_hat.is_cuda) else None
            if mel_branch is not None:
                mel_branch.__enter__()
            try:
                y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1), hps.data.filter_length, hps.data.n_mel_channels,
                                                  hps.data.sampling_rate, hps.data.hop_length, hps.data.win_length,
                                                  hps.data.mel_fmin, hps.data.mel_fmax)
            finally:
                if mel_branch is not None:
                    mel_branch.__exit__(None, None, None)
            y = commons.slice_segments(y, ids_slice, hps.train.segment_size, ids_scale=hps.data.hop_length)

            # ---- discriminator step (finetune_speaker_v2.py:205-214)
            y_d_hat_r, y_d_hat_g, _, _ = self.net_d(y, y_hat.detach())
            if mel_branch is not None:
                y_hat_mel = mel_branch.join(y_hat_mel)
        loss_disc, losses_disc_r, losses_disc_g = discriminator_loss(y_d_hat_r, y_d_hat_g)
        self.buckets_d.zero_grad()
        loss_disc.backward()
        self._st = dict(y=y, y_hat=y_hat, l_length=l_length, y_mel=y_mel, y_hat_mel=y_hat_mel, z_p=z_p, logs_q=logs_q, m_p=m_p,
                        logs_p=logs_p, z_mask=z_mask, loss_disc=loss_disc.detach())

    def _phase_b(self):
        """Discriminator update, then the generator losses against the UPDATED discriminator and their backward
        (finetune_speaker_v2.py:216-232); D is frozen for this backward."""
        hps, st = self.hps, self._st
        self.optim_d.step()                                     # (also leaves the L2 norm of the gradients it consumed)
        st["grad_norm_d"] = self.optim_d.grad_norm
        for p in self.net_d.parameters():
            p.requires_grad_(False)
        self.buckets_d.enabled(False)
        try:
            with self._autocast():
                y_d_hat_r, y_d_hat_g, fmap_r, fmap_g = self.net_d(st["y"], st["y_hat"])
            loss_dur = torch.sum(st["l_length"].float())
            loss_mel = F.l1_loss(st["y_mel"].float(), st["y_hat_mel"].float()) * hps.train.c_mel
            loss_kl = kl_loss(st["z_p"], st["logs_q"], st["m_p"], st["logs_p"], st["z_mask"]) * hps.train.c_kl
            loss_fm = feature_loss(fmap_r, fmap_g)
            loss_gen, losses_gen = generator_loss(y_d_hat_g)
            loss_gen_all = loss_gen + loss_fm + loss_mel + loss_dur + loss_kl
            self.buckets_g.zero_grad()
            loss_gen_all.backward()
        finally:
            for p in self.net_d.parameters():
                p.requires_grad_(True)
            self.buckets_d.enabled(True)
        self._out = dict(loss_disc=st["loss_disc"], loss_gen=loss_gen.detach(), loss_fm=loss_fm.detach(), loss_mel=loss_mel.detach(),
                         loss_dur=loss_dur.detach(), loss_kl=loss_kl.detach(), grad_norm_d=st["grad_norm_d"])
        self._st = None

    def _phase_c(self):
        out = self._out
        self.optim_g.step()
        out["grad_norm_g"] = self.optim_g.grad_norm
        return out

    # ---------------------------------------------------------------------------------------------
    # hipGraph execution of the step (single GPU): the step is thousands of small kernels and is
    # CPU-launch-bound in eager mode; once shapes are fixed (one length bucket) the whole iteration —
    # both forwards, both backwards, both AdamW updates — is captured once and replayed.
    # ---------------------------------------------------------------------------------------------
    def capture(self, batch, warmup=3, verify=True):
        """Warm up (workspace growth, optimizer state) on a side stream, then capture one step on `batch`'s storage.  Afterwards
        `replay()` runs one iteration on whatever has been copied into those tensors.  verify: run verify_replay() once before
        returning (two replays and one eager step from the same state must agree; state is restored) — a captured graph whose
        side-stream branches race, or that depends on memory a replay does not re-initialise, never reaches training."""
        from . import _lib
        assert self.device.type == "cuda" and self._graph is None
        timer_was, _lib.timer.enabled = _lib.timer.enabled, False
        self._static_batch = batch
        with self.on_capture_stream():
            for _ in range(warmup):
                self.step(batch)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, pool=self._graph_pool, stream=self._capture_stream, capture_error_mode=self._capture_mode()):
            self._static_out = self.step(batch)
        self._graph_pool = self._graph.pool()
        _lib.timer.enabled = timer_was
        if verify:
            self.verify_replay()
        return self._static_out

    def capture_segments(self, batch, warmup=3, verify=True):
        """Data-parallel form of capture(): the step as THREE graphs sharing one memory pool, cut at the two points where
        ranks exchange gradients.  Each graph ends by packing its network's gradients into the flat buckets; the bucket
        all-reduces run eagerly between the replays (RCCL is not captured), then the next graph reads the reduced
        gradients from the same flat views."""
        from . import _lib
        assert self.device.type == "cuda" and self._graph is None
        timer_was, _lib.timer.enabled = _lib.timer.enabled, False
        self._static_batch = batch
        with self.on_capture_stream():               # (bench.py at N > 1 warms up itself, the same way: these steps issue collectives)
            for _ in range(warmup):
                self.step(batch)
        torch.cuda.synchronize()
        self.buckets_d.manual(True)
        self.buckets_g.manual(True)
        ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        mode, cs = self._capture_mode(), self._capture_stream
        with torch.cuda.graph(ga, pool=self._graph_pool, stream=cs, capture_error_mode=mode):
            self._phase_a(batch)
            self.buckets_d.pack()
        self._graph_pool = ga.pool()
        with torch.cuda.graph(gb, pool=self._graph_pool, stream=cs, capture_error_mode=mode):
            self._phase_b()
            self.buckets_g.pack()
        with torch.cuda.graph(gc, pool=self._graph_pool, stream=cs, capture_error_mode=mode):
            self._static_out = self._phase_c()
        self._graph = (ga, gb, gc)
        _lib.timer.enabled = timer_was
        if verify:                                   # (collectives inside: every rank captures, so every rank verifies)
            self.verify_replay()
        return self._static_out

    def _state_tensors(self):
        # every parameter and both moments are views of the optimizers' flat buffers
        ts = []
        for opt in (self.optim_g, self.optim_d):
            ts += [opt.flat_p, opt.flat_m, opt.flat_v, opt.dev_state]
        return ts

    def verify_replay(self, rtol=None):
        """Replays the captured step twice and runs it once eagerly, each time from the SAME parameters, optimizer state and
        generator state (so all three draw identical noise and dropout masks: torch's graph-safe Philox offsets follow the
        eager sequence), and requires
          * the two replays to be identical: all eight reported scalars and a checksum over every parameter of both networks;
          * replay and eager to agree within `rtol` on the same quantities (default 1e-5: the replayed graph launches the same
            deterministic kernels in the same order on the same noise, so in practice they agree bit for bit).
        State is restored afterwards.  Guards the measurement against graph-replay hazards (stale memset nodes, buffers a
        replay depends on from the previous one).  Returns the replayed scalars."""
        assert self._graph is not None
        if rtol is None:
            # Three or more ranks: the captured step all-reduces the arenas' gradient buffers in place while the eager step
            # reduces 64 MiB buckets — other offsets, other chunking, so RCCL sums a given element over the ranks in another
            # order and the updates differ in the last bits (two ranks: a + b either way).  The check then still catches what it
            # is for (a race or a stale buffer changes a loss in its first digits), not bit equality.
            rtol = 1e-5 if (not self.buckets_g.active or self.buckets_g.world <= 2) else 2e-3
        ts = self._state_tensors()
        snap = [t.detach().clone() for t in ts]
        rng = torch.cuda.get_rng_state(self.device)
        flats = [(self.optim_g.flat_p, snap[0]), (self.optim_d.flat_p, snap[4])]      # every parameter of G / of D

        def restore():
            with torch.no_grad():
                for t, s in zip(ts, snap):
                    t.copy_(s)
            torch.cuda.set_rng_state(rng, self.device)

        def run(fn):
            restore()
            out = fn()
            torch.cuda.synchronize()
            vals = {k: float(v) for k, v in out.items()}
            with torch.no_grad():
                vals["param_checksum"] = float(sum(p.double().abs().sum() for p, _ in flats))
                vals["update_checksum"] = float(sum((p.double() - s.double()).abs().sum() for p, s in flats))
            return vals

        a = run(self.replay)
        b = run(self.replay)
        e = run(lambda: self.step(self._static_batch))
        restore()
        close = lambda u, v, tol: u == v or abs(u - v) <= tol * max(abs(u), abs(v), 1e-6)
        bad = [(k, a[k], b[k]) for k in a if not close(a[k], b[k], 1e-6)]
        if bad:
            raise RuntimeError(f"graph replay is not reproducible: {bad}")
        off = [(k, a[k], e[k]) for k in a if not close(a[k], e[k], rtol)]
        if off:
            raise RuntimeError(f"graph replay disagrees with the eager step (rtol {rtol}): {off}")
        self.replay_vs_eager = {k: abs(a[k] - e[k]) / max(abs(e[k]), 1e-12) for k in a}
        return a

    def load_batch(self, batch):
        for dst, src in zip(self._static_batch, batch):
            dst.copy_(src, non_blocking=True)

    def replay(self):
        if isinstance(self._graph, tuple):
            ga, gb, gc = self._graph
            ga.replay()
            self.buckets_d.all_reduce()
            gb.replay()
            self.buckets_g.all_reduce()
            gc.replay()
        else:
            self._graph.replay()
        return self._static_out

    # ---------------------------------------------------------------------------------------------
    # Real batches change shape every step (bucketed by spectrogram length, finetune_speaker_v2.py:77 /
    # data_utils.DistributedBucketSampler); a hipGraph has fixed shapes.  step_padded() pads a batch up to its
    # bucket's shape — padding is inert: every layer masks by the lengths, the losses see slices and
    # masked sums (tests/test_shape_policy_gpu.py: padded == unpadded, fp32) — and keeps ONE captured
    # graph per padded shape: the first batch of a shape runs eagerly (and sizes the workspaces), the second
    # captures, later ones only copy their data in and replay.  All graphs share one memory pool (they never
    # run concurrently) and one capture stream (per-stream scratch is then shared too).
    # ---------------------------------------------------------------------------------------------
    def step_padded(self, batch, policy):
        """One training iteration on `batch` padded to `policy`'s bucket shape; captured per shape from its second occurrence on.
        Single-process form (data-parallel runs use capture_segments on one fixed shape)."""
        padded = policy.pad(batch)
        if self.device.type != "cuda" or self.buckets_g.active:
            return self.step(padded)
        # Everything — the eager first step of a shape too — runs on ONE non-default stream: autograd keeps the stream of the
        # step that created a parameter's AccumulateGrad node, and a capture that has to synchronise with the legacy default
        # stream is illegal (hipStreamEndCapture crashes on it; FineTuner.capture warms up on a side stream for the same reason).
        cs, cur = self._capture_stream, torch.cuda.current_stream(self.device)
        key = tuple(tuple(t.shape) for t in padded)
        cache = self.__dict__.setdefault("_shape_graphs", {})
        ent = cache.get(key)
        if ent is None:
            cache[key] = {}
            cs.wait_stream(cur)
            with torch.cuda.stream(cs):
                out = self.step(padded)                           # eager: a real step, and the warm-up of this shape
            for t in padded:
                t.record_stream(cs)
            cur.wait_stream(cs)
            return out
        if "graph" not in ent:
            from . import _lib
            timer_was, _lib.timer.enabled = _lib.timer.enabled, False
            ent["batch"] = tuple(t.clone() for t in padded)
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, pool=self._graph_pool, stream=cs, capture_error_mode=self._capture_mode()):
                ent["out"] = self.step(ent["batch"])
            if self._graph_pool is None:
                self._graph_pool = g.pool()
            ent["graph"] = g
            _lib.timer.enabled = timer_was
        for dst, src in zip(ent["batch"], padded):
            dst.copy_(src, non_blocking=True)
        ent["graph"].replay()
        return ent["out"]

    def epoch_end(self):
        self.sched_g.step()            # ExponentialLR per epoch, finetune_speaker_v2.py:157-158
        self.sched_d.step()
        # the AdamW kernel reads lr from device memory; a replayed graph never passes FlatAdamW.step(), so refresh it here
        # (a device fill outside the graph, stream-ordered before the next replay)
        self.optim_g._sync_lr()
        self.optim_d._sync_lr()


class ShapePolicy:
    """Pads a collated batch (x, x_lengths, spec, spec_lengths, y, y_lengths, speakers) to one of a few fixed shapes: the
    spectrogram length up to the upper boundary of its sampler bucket (finetune_speaker_v2.py:77: [32, 300, 400, ..., 1000]),
    the token length up to the next of `t_x_steps`.  Lengths are untouched, so the padding is masked everywhere."""

    def __init__(self, boundaries, hop_length, t_x_steps=(128, 256, 384)):
        self.boundaries, self.hop, self.t_x_steps = sorted(boundaries), hop_length, sorted(t_x_steps)

    def padded_shape(self, t_x, t_y):
        ty = next((b for b in self.boundaries if b >= t_y), t_y)
        tx = next((s for s in self.t_x_steps if s >= t_x), t_x)
        return tx, ty

    def pad(self, batch):
        x, x_lengths, spec, spec_lengths, y, y_lengths, speakers = batch
        tx, ty = self.padded_shape(x.size(1), spec.size(2))
        if tx != x.size(1):
            x = F.pad(x, (0, tx - x.size(1)))
        if ty != spec.size(2):
            spec = F.pad(spec, (0, ty - spec.size(2)))
        if y.size(2) != ty * self.hop:
            y = F.pad(y, (0, ty * self.hop - y.size(2))) if y.size(2) < ty * self.hop else y[:, :, :ty * self.hop]
        return (x, x_lengths, spec, spec_lengths, y, y_lengths, speakers)


def evaluate(hps, generator, batch, max_len=1000):
    """One evaluation pass of the reference (finetune_speaker_v2.py:313-368) without its TensorBoard plumbing: the first item
    of `batch` goes through infer(); returns what the reference hands to its writer — the generated waveform cut to its length,
    its mel, and the ground-truth mel / waveform — and leaves the generator in train mode."""
    x, x_lengths, spec, spec_lengths, y, y_lengths, speakers = [t[:1] for t in batch]
    generator.eval()
    try:
        with torch.no_grad():
            y_hat, attn, mask, *_ = generator.infer(x, x_lengths, speakers, max_len=max_len)
            y_hat_lengths = mask.sum([1, 2]).long() * hps.data.hop_length
            mel = spec_to_mel_torch(spec.float(), hps.data.filter_length, hps.data.n_mel_channels, hps.data.sampling_rate,
                                    hps.data.mel_fmin, hps.data.mel_fmax)
            y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1).float(), hps.data.filter_length, hps.data.n_mel_channels,
                                              hps.data.sampling_rate, hps.data.hop_length, hps.data.win_length,
                                              hps.data.mel_fmin, hps.data.mel_fmax)
    finally:
        generator.train()
    return {"gen/mel": y_hat_mel[0], "gen/audio": y_hat[0, :, :int(y_hat_lengths[0])], "gt/mel": mel[0],
            "gt/audio": y[0, :, :int(y_lengths[0])], "attn": attn[0, 0]}


def synthetic_batch(hps, batch_size, t_y_range, device, seed=1234, rank=0, spec_fn=None, frames_per_token=5):
    """Deterministic synthetic minibatch of SURVEY.md §8(d): lengths linspace(lo, hi) sorted
    descending (TextAudioSpeakerCollate, data_utils.py:129-131), text ids with interspersed blanks
    (commons.py:24-27, T_x = 2n+1, n = T_y / frames_per_token: 5 gives C2's T_x <= 201 at 500 frames, 8 gives C1's
    T_x = (101, 81) at (400, 320) frames), waveforms = 3 sinusoids + noise peak-normalised to 0.5,
    spec = spectrogram_torch(wav), speaker ids round-robin."""
    if spec_fn is None:
        from .mel_processing import spectrogram_torch
        spec_fn = lambda w: spectrogram_torch(w, hps.data.filter_length, hps.data.sampling_rate, hps.data.hop_length, hps.data.win_length)
    gen = torch.Generator().manual_seed(seed + 1000 * rank)
    lo, hi = t_y_range
    hop = hps.data.hop_length
    t_y = torch.linspace(lo, hi, batch_size).round().long().flip(0)
    n_tok = torch.round(t_y / frames_per_token).long()
    t_x = 2 * n_tok + 1
    B, T_x, T_y = batch_size, int(t_x.max()), int(t_y.max())
    x = torch.zeros(B, T_x, dtype=torch.long)
    for i in range(B):
        ids = torch.randint(1, hps.n_symbols, (int(n_tok[i]),), generator=gen)
        x[i, 1:2 * int(n_tok[i]):2] = ids
    wav = torch.zeros(B, T_y * hop)
    tt = torch.arange(T_y * hop) / hps.data.sampling_rate
    for i in range(B):
        n = int(t_y[i]) * hop
        f = torch.rand(3, generator=gen) * 3920 + 80
        w = sum(torch.sin(2 * torch.pi * f[k] * tt[:n] + k) for k in range(3)) + torch.randn(n, generator=gen) * 0.01
        wav[i, :n] = 0.5 * w / w.abs().max()
    sid = torch.arange(B) % min(hps.data.n_speakers, 10)
    wav = wav.to(device)
    spec = spec_fn(wav)
    frame = torch.arange(T_y, device=device)[None, :] < t_y.to(device)[:, None]
    spec = spec * frame[:, None, :]
    return (x.to(device), t_x.to(device), spec, t_y.to(device), wav.unsqueeze(1), (t_y * hop).to(device), sid.to(device))
