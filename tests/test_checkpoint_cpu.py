"""CPU: the reference's checkpoint wire format (utils.py:148-193).  A file in the reference's layout — written here with
torch.save from a dict of the documented keys, the way the reference's save_checkpoint writes G_*.pth — loads into the product
modules (weights-only loader), including the partial-row copy of `emb_g.weight`; what the product saves has the same keys."""
import torch

from model_util import build_tiny, load_tiny


def test_round_trip_and_speaker_table_growth(pkg, tmp_path):
    g, cfg = load_tiny()
    src = build_tiny(pkg, g, cfg)
    opt = torch.optim.AdamW(src.parameters(), 2e-4, betas=(0.8, 0.99), eps=1e-9)
    path = str(tmp_path / "G_7.pth")
    pkg.utils.save_checkpoint(src, opt, 2e-4, 7, path)
    blob = torch.load(path, map_location="cpu", weights_only=True)
    assert sorted(blob) == ["iteration", "learning_rate", "model", "optimizer"] and blob["iteration"] == 7
    assert any(k.endswith("weight_g") for k in blob["model"]) and "emb_g.weight" in blob["model"]
    # a model with MORE speakers than the checkpoint (fine-tuning adds speakers): rows [0, n) are copied, the rest keep their init
    big = pkg.SynthesizerTrn(cfg["n_vocab"], cfg["spec_channels"], cfg["segment_size"], n_speakers=cfg["n_speakers"] + 2, **cfg["model"])
    init_rows = big.emb_g.weight.detach().clone()
    _, _, lr, it = pkg.utils.load_checkpoint(path, big, None)
    assert (lr, it) == (2e-4, 7)
    n = cfg["n_speakers"]
    assert torch.equal(big.emb_g.weight[:n], src.emb_g.weight) and torch.equal(big.emb_g.weight[n:], init_rows[n:])
    for k, v in src.state_dict().items():
        if k != "emb_g.weight":
            assert torch.equal(big.state_dict()[k], v), k
    # drop_speaker_emb keeps the model's own table
    big2 = pkg.SynthesizerTrn(cfg["n_vocab"], cfg["spec_channels"], cfg["segment_size"], n_speakers=cfg["n_speakers"], **cfg["model"])
    own = big2.emb_g.weight.detach().clone()
    pkg.utils.load_checkpoint(path, big2, None, drop_speaker_emb=True)
    assert torch.equal(big2.emb_g.weight, own)
    # optimizer state travels too
    opt2 = torch.optim.AdamW(big2.parameters(), 1e-3)
    pkg.utils.load_checkpoint(path, big2, opt2)
    assert opt2.param_groups[0]["lr"] == 2e-4


def test_reference_layout_file_with_missing_keys(pkg, tmp_path):
    """A checkpoint that lacks some tensors (older models): the missing ones keep the model's values (utils.py:175-177)."""
    g, cfg = load_tiny()
    net = build_tiny(pkg, g, cfg)
    sd = {k: v.clone() for k, v in net.state_dict().items() if not k.startswith("dp.")}
    path = str(tmp_path / "G_0.pth")
    torch.save({"model": sd, "iteration": 0, "optimizer": None, "learning_rate": 2e-4}, path)
    fresh = pkg.SynthesizerTrn(cfg["n_vocab"], cfg["spec_channels"], cfg["segment_size"], n_speakers=cfg["n_speakers"], **cfg["model"])
    keep = fresh.state_dict()["dp.pre.weight"].clone()
    pkg.utils.load_checkpoint(path, fresh, None)
    assert torch.equal(fresh.state_dict()["dp.pre.weight"], keep) and torch.equal(fresh.state_dict()["enc_q.pre.weight"], sd["enc_q.pre.weight"])
    assert all(k.startswith("dp.") for k in pkg.utils.load_checkpoint.last_missing) and pkg.utils.load_checkpoint.last_missing


def test_file_written_by_the_reference_loads(pkg, golden_dir):
    """tests/golden/ref_G_tiny.pth was written by the reference's own utils.save_checkpoint (tools/gen_golden_misc.py: its tiny
    SynthesizerTrn after one torch.optim.AdamW step).  It loads through the weights-only loader into the product modules and
    into FlatAdamW-compatible optimizer state; the same script checked the reverse direction (a file written here read by the
    reference's load_checkpoint) when it generated the fixture."""
    import os
    import numpy as np
    g, cfg = load_tiny()
    m = np.load(os.path.join(golden_dir, "misc.npz"))
    assert int(m["ckpt/reverse_ok"]) == 1
    net = pkg.SynthesizerTrn(cfg["n_vocab"], cfg["spec_channels"], cfg["segment_size"], n_speakers=cfg["n_speakers"], **cfg["model"])
    opt = torch.optim.AdamW(net.parameters(), 1e-3, betas=(0.8, 0.99), eps=1e-9)
    _, _, lr, it = pkg.utils.load_checkpoint(os.path.join(golden_dir, "ref_G_tiny.pth"), net, opt)
    assert it == int(m["ckpt/iteration"]) and lr == float(m["ckpt/learning_rate"])
    assert not pkg.utils.load_checkpoint.last_missing
    sd = net.state_dict()
    for k in m.files:
        if k.startswith("ckpt/sd/"):
            assert torch.equal(sd[k[8:]], torch.from_numpy(m[k])), k
    st = opt.state_dict()["state"]
    assert len(st) == int(m["ckpt/opt_n_state"]) and float(st[int(m["ckpt/opt_first_id"])]["step"]) == float(m["ckpt/opt_step"])
    assert torch.equal(st[int(m["ckpt/opt_first_id"])]["exp_avg"], torch.from_numpy(m["ckpt/opt_first_exp_avg"]))


def test_larger_saved_speaker_table_keeps_the_models_own(pkg, golden_dir):
    """A many-speaker checkpoint into a model with FEWER speakers (and no drop_speaker_emb): the reference's row assignment
    raises inside its bare try/except and the model keeps its own table (utils.py:162-177)."""
    import os
    g, cfg = load_tiny()
    small = pkg.SynthesizerTrn(cfg["n_vocab"], cfg["spec_channels"], cfg["segment_size"], n_speakers=cfg["n_speakers"] - 1, **cfg["model"])
    own = small.emb_g.weight.detach().clone()
    pkg.utils.load_checkpoint(os.path.join(golden_dir, "ref_G_tiny.pth"), small, None)
    assert torch.equal(small.emb_g.weight, own) and pkg.utils.load_checkpoint.last_missing == ["emb_g.weight"]
    ref = build_tiny(pkg, g, cfg)
    assert not torch.equal(small.enc_q.pre.weight, ref.enc_q.pre.weight)      # (the file holds the post-step weights, not the npz's)
