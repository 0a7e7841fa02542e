#!/usr/bin/env python3
"""Headline benchmark: 22.05 kHz waveform samples / second of the VITS fine-tune step
(G forward, D step, G step, two AdamW updates — reference finetune_speaker_v2.py:174-232) on
N MI355X GPUs of one node, one process per GPU over RCCL.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): configs/modified_finetune_speaker.json
(n_speakers=13), per-rank batch 16, T_y = linspace(200, 500) frames, T_x = 2*round(T_y/5)+1,
bf16 autocast, synthetic seeded data, random-init weights.  Weak scaling: every rank runs the
same per-rank batch on its own data.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime initialises; see the package __init__

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(cfgs, seconds_budget=40.0, threads=16):
    """The CPU path timed on this box's host cores (rank 0, N = 1 only), on BASELINE config C1's shapes
    (configs/finetune_speaker.json: n_speakers = 999, batch 2, T_y = (400, 320), T_x = (161, 129)): the oracle's torch-cpu fp32
    restatement of the same step (kind "port": the reference script asserts CUDA and cannot run on a CPU, and its sources do
    not travel to the GPU box) — one warm-up step, then the median of the steps that fit the budget (up to 6).  The alignment DP
    is timed separately on ONE core (the reference build has no OpenMP): the reference's own Cython routine compiled into
    oracle/_ref when that is present, else the C restatement."""
    import statistics
    from importlib import import_module
    import numpy as np
    from oracle import vits_torch as O
    from oracle import mas as omas
    tr = import_module("personalized_text-to-speech_amd.train")
    P = import_module("personalized_text-to-speech_amd")
    name, B, t_y = cfgs.WORKLOADS["C1"]
    hps = cfgs.get(name)
    torch.manual_seed(0)
    torch.set_num_threads(threads)           # the GPU box's CPU share for one GPU (torch-cpu oversubscribes badly on all 128 hardware threads)
    m = {k: v for k, v in hps.model.items()}
    g = P.SynthesizerTrn(hps.n_symbols, hps.data.filter_length // 2 + 1, hps.train.segment_size // hps.data.hop_length,
                         n_speakers=hps.data.n_speakers, **m)
    d = P.MultiPeriodDiscriminator(False)
    sd_g = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in g.state_dict().items()}
    sd_d = {k: v.detach().clone().requires_grad_(True) for k, v in d.state_dict().items()}
    del g, d
    opt_g = torch.optim.AdamW([v for v in sd_g.values() if v.requires_grad], hps.train.learning_rate, betas=hps.train.betas, eps=hps.train.eps)
    opt_d = torch.optim.AdamW(list(sd_d.values()), hps.train.learning_rate, betas=hps.train.betas, eps=hps.train.eps)
    batch = tr.synthetic_batch(hps, B, t_y, "cpu", frames_per_token=8,          # T_y = (400, 320), T_x = (101, 81)
                               spec_fn=lambda w: O.spectrogram(w, hps.data.filter_length, hps.data.hop_length, hps.data.win_length))
    x, spec = batch[0], batch[2]
    hp = dict(hps.data); hp.update(hps.train)
    T_x = x.size(1)
    times, t_total = [], 0.0
    while True:
        noise = [torch.randn(B, hps.model.inter_channels, spec.size(2)), torch.randn(B, 2, T_x), torch.rand(B)]
        t0 = time.perf_counter()
        loss_disc, rest = O.train_losses(sd_g, sd_d, hps.model, hp, batch, noise)
        opt_d.zero_grad(); loss_disc.backward(); opt_d.step()
        loss_gen_all, _ = O.generator_losses(sd_d, hp, *rest)
        opt_g.zero_grad(); loss_gen_all.backward(); opt_g.step()
        dt = time.perf_counter() - t0
        times.append(dt)
        t_total += dt
        if len(times) >= 6 or (len(times) >= 2 and t_total > seconds_budget):
            break
    timed = times[1:]                                    # the first step is the warm-up
    step_s = statistics.median(timed)
    # alignment DP on one core: C1's shape and C3's (b = 64, 800 x 321)
    dp = {}
    fn, kind = omas.mas_port, "port (oracle/mas_ref.c)"
    try:
        if omas.have_reference():
            fn, kind = omas.mas_reference, "reference (core.pyx compiled into oracle/_ref)"
    except Exception:
        pass
    rng = np.random.default_rng(0)
    for tag, (bb, ty, tx) in dict(c1=(2, 400, 101), c3=(64, 800, 321)).items():
        nc = rng.standard_normal((bb, ty, tx)).astype(np.float32)
        t_ys, t_xs = np.full(bb, ty, np.int32), np.full(bb, tx, np.int32)
        fn(nc, t_ys, t_xs)
        reps = []
        for _ in range(5):
            t0 = time.perf_counter(); fn(nc, t_ys, t_xs); reps.append(time.perf_counter() - t0)
        dp[tag] = dict(shape=[bb, ty, tx], ms=statistics.median(reps) * 1e3, Mcell_per_s=bb * ty * tx / statistics.median(reps) / 1e6)
    return dict(value=B * hps.train.segment_size / step_s, unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample=f"C1 shapes (finetune_speaker.json, batch {B}, T_y=(400,320), T_x=({T_x},{int(batch[1][1])})): oracle torch-cpu fp32 step + alignment DP, "
                       f"median of {len(timed)} step(s) after 1 warm-up, {step_s:.2f} s/step, {t_total:.1f} s in all",
                alignment_dp=dict(cores=1, kind=kind, **dp))


def alignment_gpu(P, device):
    """vits_mas_f32 (zero fill + DP/backtrack launches) on the shapes cpu_baseline.alignment_dp times on one host core, plus
    C2's: HIP events on the launch stream, median of 10 after 2 warm-up calls."""
    import statistics
    import numpy as np
    out, rng = {}, np.random.default_rng(0)
    for tag, (bb, ty, tx) in dict(c1=(2, 400, 101), c2=(16, 500, 201), c3=(64, 800, 321)).items():
        nc = torch.from_numpy(rng.standard_normal((bb, ty, tx)).astype(np.float32)).to(device)
        t_ys = torch.full((bb,), ty, dtype=torch.int32, device=device); t_xs = torch.full((bb,), tx, dtype=torch.int32, device=device)
        reps = []
        for i in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); P.monotonic_align.maximum_path_lengths(nc, t_ys, t_xs); e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                reps.append(e0.elapsed_time(e1))
        ms = statistics.median(reps)
        out[tag] = dict(shape=[bb, ty, tx], ms=ms, Mcell_per_s=bb * ty * tx / ms / 1e3)
    return out


def secondary_fp32(tr, hps, device, batch, batch_size):
    """The same workload in the fp32 parity mode (the mode that meets BASELINE.json's 1e-3 tolerance): eager launches,
    1 warm-up + 3 timed steps.  Reported beside the bf16 headline, never as `value`."""
    ft = tr.FineTuner(hps, device, amp=False)
    ft.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ft.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    del ft
    torch.cuda.empty_cache()
    return dict(fp32_ms_per_step=dt * 1e3, fp32_samples_per_s=batch_size * hps.train.segment_size / dt, execution="eager launches, 3 steps")


def bench_infer(args, P, cfgs, device):
    """--workload C4 (BASELINE.json configs[3]): SynthesizerTrn.infer() on 32 prompts of 513 tokens, durations forced to 861
    frames (10.0 s at 22.05 kHz) per item; value = generated waveform samples per second (eager launches, one GPU)."""
    hps = cfgs.get(cfgs.C4["config"])
    torch.manual_seed(hps.train.seed)
    g = P.SynthesizerTrn(hps.n_symbols, hps.data.filter_length // 2 + 1, hps.train.segment_size // hps.data.hop_length,
                         n_speakers=hps.data.n_speakers, **hps.model).to(device).eval()
    x, xl, sid, dur = cfgs.c4_inputs(hps, device)
    ac = torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.fp32)

    def step():
        with torch.no_grad(), ac:
            return g.infer(x, xl, sid=sid, noise_scale=cfgs.C4["noise_scale"], noise_scale_w=cfgs.C4["noise_scale_w"], durations=dur)[0]

    for _ in range(args.warmup):
        o = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        o = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(o).all())
    P._lib.timer.enabled = True; P._lib.timer.reset()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    P._lib.timer.enabled = False
    summ = P._lib.timer.summary()
    per_kernel = {n: dict(launches_per_step=sm["calls"] / 2, ms_per_step=sm["total_ms"] / 2, avg_launch_us=sm["avg_ms"] * 1e3,
                          algorithmic_GBps=sm["units_total"][1] / (sm["total_ms"] * 1e-3) / 1e9) for n, sm in summ.items()}
    dom = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_step"])
    roof = dict(kernel=dom, bound="hbm", achieved=per_kernel[dom]["algorithmic_GBps"], peak=HBM_PEAK_GBS, unit="GB/s",
                frac=per_kernel[dom]["algorithmic_GBps"] / HBM_PEAK_GBS, traffic=None, avg_launch_us=per_kernel[dom]["avg_launch_us"],
                all_kernels=per_kernel)
    samples = o.size(0) * o.size(-1) * args.steps
    return dict(metric="22.05 kHz waveform samples/sec, VITS infer()", value=samples / elapsed, unit="samples/s", n_gpus=1,
                steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True, scaling="weak",
                vs_baseline=None, dtype="fp32" if args.fp32 else "bf16", data="synthetic",
                config=dict(workload=f"C4: infer(), batch {o.size(0)}, T_x = {x.size(1)} tokens, {o.size(-1)} samples per item (durations forced)",
                            parallelism="dp1", execution="eager launches"), roofline=roof)


def emit(line):
    """The ONE line of the contract, on the process's real stdout."""
    os.write(_REAL_STDOUT, (json.dumps(line) + "\n").encode())


_REAL_STDOUT = 1


def main():
    # Native libraries print to stdout too (RCCL's version banner at the first collective): everything except the JSON line
    # goes to stderr, the line itself to the descriptor stdout had when the process started.
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--branches", default=None, help="comma list of side-stream branches (enc_p,dp,prior,mel); default: enc_p,dp,prior")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--fp32", action="store_true", help="parity mode: no bf16 autocast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the fp32 (parity mode) timing reported next to the bf16 headline")
    ap.add_argument("--segmented", action="store_true", help="force the three-graph (data-parallel) form of the captured step at N = 1")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying the captured hipGraph")
    ap.add_argument("--exchange", action="store_true", help="N = 1 only: run the data-parallel gradient exchange anyway (one-rank RCCL group, "
                    "three graphs with bucket all-reduces between them) to time what that form costs on one GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("VITS_BENCH_ONE_DEVICE") == "1":     # rehearsal of the N > 1 code path on a one-GPU box (with the gloo backend)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("VITS_DIST_BACKEND", "nccl") == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(os.environ["VITS_DIST_BACKEND"], rank=rank, world_size=world)

    if world == 1 and args.exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    coll_device = device if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")   # small control collectives

    from importlib import import_module
    P = import_module("personalized_text-to-speech_amd")
    cfgs = import_module("personalized_text-to-speech_amd.configs")
    tr = import_module("personalized_text-to-speech_amd.train")
    P._lib.lib()                                          # fail loudly if the HIP library is missing

    if args.workload == "C4":
        if world != 1:
            raise SystemExit("workload C4 (inference) runs as independent replicas: launch it with --gpus 1 per GPU")
        emit(bench_infer(args, P, cfgs, device))
        return
    cfg_name, batch_size, t_y_range = cfgs.WORKLOADS[args.workload]
    hps = cfgs.get(cfg_name)
    tuner = tr.FineTuner(hps, device, amp=not args.fp32, force_exchange=world == 1 and args.exchange)
    if args.branches is not None:
        tuner.side_branches = frozenset(b for b in args.branches.split(",") if b)
    batch = tr.synthetic_batch(hps, batch_size, t_y_range, device, rank=rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = not args.eager
    branch_note = None
    if use_graph and world == 1:
        cap = tuner.capture_segments if (args.segmented or args.exchange) else tuner.capture
        cap(batch, warmup=3, verify=False)           # (verified four times right below)
        try:
            for _ in range(4):                           # same state -> same result, and agrees with the eager step: four rounds,
                tuner.verify_replay()                    # because a race between side-stream branches would be intermittent
        except RuntimeError as e:
            if not tuner.side_branches:
                raise
            # never time a graph that failed verification: recapture the step without side-stream branches (serial) and verify that
            branch_note = f"side-stream branches {sorted(tuner.side_branches)} switched off after: {str(e)[:160]}"
            print("bench: " + branch_note, file=sys.stderr)
            tuner._graph = None
            tuner.side_branches = frozenset()
            cap(batch, warmup=1, verify=False)
            for _ in range(2):
                tuner.verify_replay()
    elif use_graph:
        # N > 1: three graphs cut where the ranks exchange gradients, RCCL all-reduce between the replays.  Every stage that
        # issues collectives (warm-up steps, verify_replay's replays and eager step) runs on ALL ranks or on none, and the
        # per-rank try/except blocks contain no collective before their outcome has been agreed on (MIN all-reduce of a flag):
        #   1. warm-up: eager steps with bucket all-reduces — symmetric; an exception here is fatal for the job (raised)
        #   2. capture: no collective inside (pack() only)                      -> flag
        #   3. verify_replay on every rank (same number of collectives each; it raises only after its last one) -> flag
        def agreed(ok):
            flag = torch.tensor([1 if ok else 0], device=coll_device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag) == 1

        with tuner.on_capture_stream():                  # (the stream the captures use: FineTuner.on_capture_stream)
            for _ in range(3):
                tuner.step(batch)
        torch.cuda.synchronize()
        why = None
        try:
            tuner.capture_segments(batch, warmup=0, verify=False)     # (verify_replay follows, after the ranks agreed that capture worked)
        except Exception as e:                            # noqa: BLE001 - reported, then a collective decision
            why = f"capture: {type(e).__name__}: {e}"
        use_graph = agreed(why is None)
        if use_graph:
            try:
                tuner.verify_replay()
            except RuntimeError as e:
                why = f"verify_replay: {e}"
            use_graph = agreed(why is None)
        if not use_graph:
            print(f"[bench rank {rank}] captured step unavailable ({why or 'another rank failed'}); every rank falls back to eager launches",
                  file=sys.stderr, flush=True)
            tuner._graph = None
            tuner.buckets_d.manual(False); tuner.buckets_g.manual(False)
    step = tuner.replay if use_graph else (lambda: tuner.step(batch))
    trace = os.environ.get("VITS_BENCH_TRACE") == "1"      # debugging aid: sync + log every step
    for i in range(args.warmup):
        out = step()
        if trace:
            torch.cuda.synchronize()
            bad = [n for n, q in list(tuner.net_g.named_parameters()) + list(tuner.net_d.named_parameters()) if not torch.isfinite(q).all()]
            print(f"warmup {i} ok", {k: round(float(v), 4) for k, v in out.items()}, "non-finite params:", bad[:4], file=sys.stderr, flush=True)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step()
        if trace:
            torch.cuda.synchronize(); print(f"step {i} ok", file=sys.stderr, flush=True)
    sync()
    elapsed = time.perf_counter() - t0
    # per-kernel roofline leg: the same step, launched eagerly with a HIP event pair around every C-ABI
    # launch of this repo's kernels (events cannot be recorded inside a replayed graph); recorded on the
    # stream the kernels are launched on.  Untimed for the headline, same shapes and data.
    P._lib.timer.enabled = True
    P._lib.timer.reset()
    for _ in range(3):
        tuner.step(batch)
    torch.cuda.synchronize()
    summ_branches = P._lib.timer.summary()
    # ... and once more with the side-stream branches off: every launch then has the GPU to itself, which is what a kernel's
    # duration in the rocprofv3 trace under profiles/ measures — `roofline.frac` is quoted on this clock, the one above
    # (launches sharing the GPU with the branches, as in the real step) is `frac_with_branches`
    branches_were, tuner.side_branches = tuner.side_branches, frozenset()
    tuner.step(batch)
    P._lib.timer.reset()
    for _ in range(3):
        tuner.step(batch)
    torch.cuda.synchronize()
    tuner.side_branches = branches_were
    P._lib.timer.enabled = False
    pair_ms = P._lib.KernelTimer.pair_overhead_ms()
    # what a caller without a stable shape gets: the same step launched from Python, no graph (3 steps, untimed for the headline)
    t1 = time.perf_counter()
    for _ in range(3):
        tuner.step(batch)
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t1) / 3 * 1e3
    if world > 1:
        t = torch.tensor([elapsed], device=coll_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    losses = {k: float(v) for k, v in out.items()}
    if not all(v == v and abs(v) != float("inf") for v in losses.values()):
        raise SystemExit(f"non-finite losses: {losses}")

    if rank == 0:
        samples = world * batch_size * hps.train.segment_size * args.steps
        T_x, T_y = batch[0].size(1), batch[2].size(2)
        # roofline of the dominant hand-written kernel: algorithmic bytes
        # per launch / average launch time from HIP events recorded inside the timed region
        summ = P._lib.timer.summary()
        roof, per_kernel = None, {}
        for name, sm in summ.items():
            flop_or_units, byts = sm["units_total"]
            per_kernel[name] = dict(launches_per_step=sm["calls"] / 3, ms_per_step=sm["total_ms"] / 3,
                                    avg_launch_us=sm["avg_ms"] * 1e3, algorithmic_GBps=byts / (sm["total_ms"] * 1e-3) / 1e9,
                                    TFLOPs=(flop_or_units / (sm["total_ms"] * 1e-3) / 1e12) if "conv" in name else None)
        if per_kernel:
            dom = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_step"])        # dominant hand-written kernel
            d, sm = per_kernel[dom], summ[dom]
            traffic = None                       # HBM bytes per launch from the committed PMC passes (not collectable in-process)
            try:
                traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[dom]["bytes_per_launch"]
            except Exception:
                pass
            sb = summ_branches.get(dom)
            with_branches = (sb["units_total"][1] / (sb["total_ms"] * 1e-3) / 1e9) if sb else None
            # an event pair around nothing measures pair_ms: per launch that much of a sample is not the kernel
            net_ms = max(sm["total_ms"] - pair_ms * sm["calls"], 0.5 * sm["total_ms"])
            net_gbps = sm["units_total"][1] / (net_ms * 1e-3) / 1e9
            roof = dict(kernel=dom, bound="hbm", achieved=net_gbps, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=net_gbps / HBM_PEAK_GBS, traffic=traffic, avg_launch_us=net_ms / sm["calls"] * 1e3,
                        frac_raw_events=d["algorithmic_GBps"] / HBM_PEAK_GBS, avg_launch_us_raw_events=d["avg_launch_us"],
                        event_pair_overhead_us=pair_ms * 1e3,
                        frac_with_branches=(with_branches / HBM_PEAK_GBS) if with_branches else None,
                        avg_launch_us_with_branches=(sb["avg_ms"] * 1e3) if sb else None,
                        clock="HIP events around every launch of 3 eager steps with the side-stream branches off (a launch alone on the GPU, "
                              "as rocprofv3 times it), minus what an event pair around nothing measures (event_pair_overhead_us); "
                              "*_raw_events: without that subtraction; *_with_branches: raw events with the branches on",
                        bytes_per_launch=sm["units_per_call"][1], launches_per_step=d["launches_per_step"],
                        mfma_TFLOPs=(d["TFLOPs"] * sm["total_ms"] / net_ms) if d["TFLOPs"] else None, mfma_peak_TFLOPs=(2500.0 if not args.fp32 else 157.3),
                        mfma_frac=(d["TFLOPs"] * sm["total_ms"] / net_ms / (2500.0 if not args.fp32 else 157.3)) if d["TFLOPs"] else None,
                        note="algorithmic bytes = x + y (+res) + w once per launch, summed over the launches of one step, / their summed HIP-event time; "
                             "traffic (PMC FETCH_SIZE/WRITE_SIZE) is collected in separate rocprofv3 --pmc passes, see profiles/",
                        all_kernels=per_kernel)
        line = dict(metric="22.05 kHz waveform samples/sec, VITS fine-tune fwd+bwd", value=samples / elapsed, unit="samples/s",
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3,
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype="fp32" if args.fp32 else "bf16",
                    data="synthetic", eager_ms_per_step=eager_ms,
                    config=dict(workload=f"{args.workload}: {cfg_name}.json, per-rank batch {batch_size}, T_y<= {T_y} frames, "
                                         f"T_x<= {T_x} tokens, segment {hps.train.segment_size} samples, fwd+bwd+AdamW (G and D)",
                                parallelism=f"dp{world}", execution=("hipGraph replay" if world == 1 and not (args.segmented or args.exchange) else "three hipGraphs, gradient all-reduce between them") if use_graph else "eager launches",
                                side_stream_branches=sorted(tuner.side_branches), kernels=P.kernels.BACKENDS, losses=losses),
                    roofline=roof)
        if branch_note:
            line["config"]["note"] = branch_note
        if world == 1 and not args.fp32 and not args.no_secondary:
            line["secondary"] = secondary_fp32(tr, hps, device, batch, batch_size)
            # BASELINE.json configs[3] (C4, inference) beside the headline: the same numbers `--workload C4` prints on its own
            c4 = argparse.Namespace(**{**vars(args), "steps": 5, "warmup": 2})
            r = bench_infer(c4, P, cfgs, device)
            line["secondary"]["c4_infer"] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config")}
            line["secondary"]["c4_infer"]["roofline_frac"] = r["roofline"]["frac"]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfgs)
            line["cpu_baseline"]["alignment_dp"]["gpu"] = alignment_gpu(P, device)
        emit(line)
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
