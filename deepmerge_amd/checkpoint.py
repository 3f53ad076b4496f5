"""Checkpoint interop with the reference trainer (Train_SMT.py:325-331 writes, :206-216 reads back):

    {"net": net.state_dict(), "optimizer": optimizer.state_dict(), "epoch", "time", "scales", "depth", "name"}

`net` keys / shapes / dtypes are those of the reference modules (parity-tested manifests), so a reference `.pth`
loads into the drop-in modules unchanged.  The optimizer entry is `torch.optim.Adam.state_dict()` layout over
`filter(requires_grad, net.parameters())` (Train_SMT.py:192-193); PairTrainer keeps Adam's moments in flat buffers,
and this module converts both ways.  Entries exist only for parameters that have received a gradient, like upstream.
Known difference: the fused Adam keeps ONE step count, so a parameter whose first gradient arrives late (after un-freezing,
say) is bias-corrected with the global step rather than its own; the reference's train() never does that.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

_GROUP_DEFAULTS = {"weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                   "differentiable": False, "fused": None}


def _ordered_params(net: torch.nn.Module):
    return [p for p in net.parameters() if p.requires_grad]


def optimizer_state_dict(trainer) -> Dict[str, Any]:
    """PairTrainer's Adam state as `torch.optim.Adam(...).state_dict()` would hold it."""
    fp = trainer.fp
    where = {id(p): o for p, o in zip(fp.params, fp.offsets)}
    params = _ordered_params(trainer.net)
    state = {}
    if trainer.step_count > 0:
        # torch.optim.Adam creates a parameter's state lazily, at its first step WITH a gradient: parameters that never receive one
        # (final_features.*, head.* on the designed-feature path) have no entry upstream.  Here their moments are exactly zero forever
        # (zero gradient in, zero out), which is how they are recognised -- a parameter whose gradients were all EXACTLY zero so far (a
        # dead unit) is indistinguishable and gets no entry either; reloading it is harmless (zero moments are what it has).  One pass over
        # the flat buffers and one host copy decide all parameters (it used to be two host syncs per parameter).
        nz = ((trainer.m != 0) | (trainer.v != 0)).to(torch.int32).cumsum(0, dtype=torch.int64)
        lo = torch.tensor([where[id(p)] for p in params], dtype=torch.int64, device=nz.device)
        hi = lo + torch.tensor([p.numel() for p in params], dtype=torch.int64, device=nz.device)
        before = torch.where(lo > 0, nz[(lo - 1).clamp(min=0)], torch.zeros_like(lo))
        touched = (nz[hi - 1] - before).cpu().tolist()
        for i, p in enumerate(params):
            o, n = where[id(p)], p.numel()
            if touched[i] == 0:
                continue
            state[i] = {"step": torch.tensor(float(trainer.step_count)),
                        "exp_avg": trainer.m[o:o + n].view_as(p).detach().clone().cpu(),
                        "exp_avg_sq": trainer.v[o:o + n].view_as(p).detach().clone().cpu()}
    # `dm_step_count`: the fused Adam's one step count, kept even when no parameter has an entry (torch ignores unknown group keys)
    group = dict(_GROUP_DEFAULTS, lr=trainer.lr, betas=tuple(trainer.betas), eps=trainer.eps, params=list(range(len(params))),
                 dm_step_count=int(trainer.step_count))
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(trainer, sd: Dict[str, Any]) -> None:
    """Inverse of optimizer_state_dict.  Parameters without an entry (never received a gradient upstream, e.g. `head`)
    get zero moments, which is what torch's lazy state initialisation amounts to."""
    fp = trainer.fp
    where = {id(p): o for p, o in zip(fp.params, fp.offsets)}
    params = _ordered_params(trainer.net)
    groups = sd.get("param_groups", [])
    if len(groups) != 1 or len(groups[0]["params"]) != len(params):
        raise ValueError(f"optimizer state has {sum(len(g['params']) for g in groups)} parameters in {len(groups)} group(s); "
                         f"the model has {len(params)} in one")
    g = groups[0]
    trainer.lr, trainer.betas, trainer.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
    trainer.m.zero_()
    trainer.v.zero_()
    steps = set()
    for slot, i in enumerate(g["params"]):
        st = sd["state"].get(i)
        if st is None:
            continue
        p = params[slot]
        o, n = where[id(p)], p.numel()
        if tuple(st["exp_avg"].shape) != tuple(p.shape):
            raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} != parameter shape {tuple(p.shape)}")
        trainer.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
        trainer.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
        steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError(f"per-parameter step counts differ ({sorted(steps)}); the fused Adam keeps one step count")
    trainer.step_count = steps.pop() if steps else int(g.get("dm_step_count", 0))


def save_checkpoint(path: str, trainer, epoch: int, elapsed: float) -> Dict[str, Any]:
    """Write the reference's checkpoint dict (Train_SMT.py:325-331)."""
    net = trainer.net
    state = {"net": {k: v.detach().cpu().clone() for k, v in net.state_dict().items()},
             "optimizer": optimizer_state_dict(trainer),
             "epoch": epoch,
             "time": round(float(elapsed), 2),
             "scales": getattr(net, "input_image_scales", None),
             "depth": getattr(net, "depth", None),
             "name": getattr(net, "name", type(net).__name__)}
    torch.save(state, path)
    return state


def load_checkpoint(path: str, net: torch.nn.Module, trainer=None, map_location: Optional[str] = "cpu") -> Dict[str, Any]:
    """Load a checkpoint written by save_checkpoint or by the reference trainer.  Returns the dict (epoch, time, ...).
    With a PairTrainer, its flat buffers, bf16 weight mirror and Adam moments are refreshed too."""
    state = torch.load(path, map_location=map_location, weights_only=False)
    net.load_state_dict(state["net"], strict=True)        # in place: flat-buffer views stay attached
    if trainer is not None:
        if trainer.net is not net:
            raise ValueError("trainer.net is not the module being loaded")
        trainer.fp.refresh_lp()
        if "optimizer" in state:
            load_optimizer_state_dict(trainer, state["optimizer"])
    return state
