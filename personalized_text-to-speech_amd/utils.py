"""Checkpoint wire format of the reference (utils.py:148-193): `load_checkpoint` / `save_checkpoint` with the same
signatures, the same dict layout {'model', 'iteration', 'optimizer', 'learning_rate'} and the same state_dict keys
(`weight_g` / `weight_v` of the weight-normed convolutions included — tests/test_product_host.py), so files written by
either side load on the other: the reference's public G_0.pth / D_0.pth fine-tuning starts from load here, and what
`save_checkpoint` writes is what the reference's VC_inference.py reads.

Differences, none visible in the files:
  * `torch.load(..., weights_only=True)`: nothing from a checkpoint file is executed (the reference unpickles freely);
  * a key missing from the file keeps the model's own tensor, as in the reference (utils.py:175-177), but the names are
    returned instead of being logged one by one; so does an `emb_g.weight` that does not fit into the model's table (more
    rows, other width): the reference's bare `except` keeps the model's table there too.
"""
import json
import logging
import os

import torch

from .configs import HParams

logger = logging.getLogger(__name__)


def load_checkpoint(checkpoint_path, model, optimizer=None, drop_speaker_emb=False):
    """-> (model, optimizer, learning_rate, iteration), reference utils.py:148-180.  `emb_g.weight` is copied row-wise into
    the (possibly larger) speaker table of `model` unless drop_speaker_emb (fine-tuning adds speakers: utils.py:166-171)."""
    assert os.path.isfile(checkpoint_path), checkpoint_path
    checkpoint_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    iteration = checkpoint_dict["iteration"]
    learning_rate = checkpoint_dict["learning_rate"]
    if optimizer is not None and checkpoint_dict.get("optimizer") is not None:
        optimizer.load_state_dict(checkpoint_dict["optimizer"])
    saved_state_dict = checkpoint_dict["model"]
    target = model.module if hasattr(model, "module") else model
    state_dict = target.state_dict()
    new_state_dict, missing = {}, []
    for k, v in state_dict.items():
        if k not in saved_state_dict:                  # (a tensor of another shape is handed on: load_state_dict raises, as there)
            missing.append(k)
            new_state_dict[k] = v
        elif k == "emb_g.weight":
            saved = saved_state_dict[k]
            if not drop_speaker_emb:
                if saved.dim() == 2 and saved.shape[0] <= v.shape[0] and saved.shape[1] == v.shape[1]:
                    v = v.clone()
                    v[:saved.shape[0], :] = saved.to(v)
                else:
                    # a table with MORE rows than the model's (a many-speaker G_0.pth into a few-speaker fine-tune config) or
                    # another gin_channels: the reference's row assignment raises inside its bare try/except and the model
                    # keeps its own table (utils.py:162-177) — same here, reported with the missing keys
                    missing.append(k)
            new_state_dict[k] = v
        else:
            new_state_dict[k] = saved_state_dict[k]
    target.load_state_dict(new_state_dict)
    if missing:
        logger.info("%d tensors are not in the checkpoint (kept as initialised): %s ...", len(missing), missing[:3])
    logger.info("Loaded checkpoint '%s' (iteration %s)", checkpoint_path, iteration)
    load_checkpoint.last_missing = missing
    return model, optimizer, learning_rate, iteration


def save_checkpoint(model, optimizer, learning_rate, iteration, checkpoint_path):
    """reference utils.py:183-193"""
    logger.info("Saving model and optimizer state at iteration %s to %s", iteration, checkpoint_path)
    target = model.module if hasattr(model, "module") else model

    def own_storage(obj):
        # parameters and optimizer moments may be views of flat buffers (optim.FlatAdamW): written as they are, every tensor
        # would drag the whole flat storage into the file; the reference's files hold one storage per tensor
        if torch.is_tensor(obj):
            return obj.detach().clone()
        if isinstance(obj, dict):
            out = type(obj)((k, own_storage(v)) for k, v in obj.items())
            if hasattr(obj, "_metadata"):
                out._metadata = obj._metadata                  # (module versions torch's load_state_dict looks at)
            return out
        if isinstance(obj, (list, tuple)):
            return type(obj)(own_storage(v) for v in obj)
        return obj

    torch.save({"model": own_storage(target.state_dict()), "iteration": iteration,
                "optimizer": own_storage(optimizer.state_dict()) if optimizer is not None else None,
                "learning_rate": learning_rate}, checkpoint_path)


def get_hparams_from_file(config_path):
    """reference utils.py:359-365: the JSON config as an attribute-style nested dict (+ n_symbols from its `symbols` list)."""
    with open(config_path, "r", encoding="utf-8") as f:
        config = json.load(f)
    if "symbols" in config:
        config["n_symbols"] = len(config["symbols"])
    return HParams(config)
