"""Checkpoint wire format: Lightning `.ckpt` envelopes with the reference's layout.

The reference trains a LightningModule `DefaultYolov5Experiment` whose network is the attribute `net`
(kod/lightning/experiments/yv5_baseline/exp.py:36-58) and checkpoints it with lightning's ModelCheckpoint
(kod/configs/callbacks/model_checkpoint.yaml, kod/lightning/tasks/trainer.py:122-137).  A `.ckpt` is a
`torch.save`d dict; the entries a resume / README eval command needs are

    state_dict         {"net." + key: tensor}  - the network's 360 keys (yv5s), fp32
    optimizer_states   [torch.optim.SGD.state_dict()] for the optimizer SmartOptimizer builds
                       (kod/nn/optim/smart.py:20-60): groups bias_params | decay_params | norm_params, parameters
                       numbered consecutively group by group in module-walk order, state[i]["momentum_buffer"]
    lr_schedulers      [LambdaLR.state_dict()]
    epoch, global_step, pytorch-lightning_version

Here the parameters / momentum live in the engine's flat arenas, so this module maps between the two.
Host-only Python (checkpoint I/O stays Python per the north star).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

GROUP_NAMES = ("bias_params", "decay_params", "norm_params")
NET_PREFIX = "net."
LIGHTNING_VERSION = "2.0.9"


def optimizer_param_order(net: nn.Module) -> Tuple[List[str], List[str], List[str]]:
    """Parameter names per optimizer group in the order SmartOptimizer.__call__ collects them (smart.py:20-35)."""
    norm_types = tuple(v for k, v in nn.__dict__.items() if "Norm" in k and isinstance(v, type))
    bias, decay, norm = [], [], []
    for mod_name, m in net.named_modules():
        for p_name, _ in m.named_parameters(recurse=False):
            full = f"{mod_name}.{p_name}" if mod_name else p_name
            if p_name == "bias":
                bias.append(full)
            elif p_name == "weight" and isinstance(m, norm_types):
                norm.append(full)
            else:
                decay.append(full)
    return bias, decay, norm


def optimizer_state_dict(net, optimizer) -> Dict:
    """torch.optim.SGD.state_dict() layout for a SmartSGD over `net` (momentum read from the engine's arena)."""
    groups = optimizer_param_order(net)
    params = dict(net.named_parameters())
    eng = net.engine() if getattr(optimizer, "steps_taken", 0) > 0 else None
    state, pgs, idx = {}, [], 0
    for names, g in zip(groups, optimizer.param_groups):
        ids = []
        for n in names:
            if eng is not None:
                off, numel = eng.layout[n]
                state[idx] = {"momentum_buffer": eng.m_arena[off:off + numel].view(params[n].shape).detach().clone().cpu()}
            ids.append(idx)
            idx += 1
        pgs.append(dict(lr=float(g["lr"]), momentum=float(g["momentum"]), dampening=0, weight_decay=float(g["weight_decay"]),
                        nesterov=True, maximize=False, foreach=None, differentiable=False, name=g["name"],
                        initial_lr=float(g.get("initial_lr", g["lr"])), params=ids))
    return {"state": state, "param_groups": pgs}


def load_optimizer_state_dict(net, optimizer, sd: Dict):
    groups = optimizer_param_order(net)
    params = dict(net.named_parameters())
    assert len(sd["param_groups"]) == 3, "expected SmartOptimizer's three parameter groups"
    eng = net.engine()
    eng.m_arena.zero_()
    loaded = 0
    for names, g_saved, g in zip(groups, sd["param_groups"], optimizer.param_groups):
        assert len(names) == len(g_saved["params"]), (g_saved.get("name"), len(names), len(g_saved["params"]))
        for key in ("lr", "momentum", "weight_decay", "initial_lr"):
            if key in g_saved:
                g[key] = g_saved[key]
        for n, i in zip(names, g_saved["params"]):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is None or st.get("momentum_buffer") is None:
                continue
            buf = st["momentum_buffer"]
            assert tuple(buf.shape) == tuple(params[n].shape), (n, buf.shape, params[n].shape)
            off, numel = eng.layout[n]
            eng.m_arena[off:off + numel].copy_(buf.reshape(-1).to(eng.m_arena.device, torch.float32))
            loaded += 1
    optimizer.steps_taken = 1 if loaded else 0
    return loaded


def save_checkpoint(path: str, net, optimizer=None, epoch: int = 0, global_step: int = 0,
                    lr_scheduler_state: Optional[Dict] = None, extra: Optional[Dict] = None) -> Dict:
    ckpt = {
        "epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": LIGHTNING_VERSION,
        "state_dict": {NET_PREFIX + k: v.detach().clone().cpu() for k, v in net.state_dict().items()},
        "loops": {}, "callbacks": {},
        "optimizer_states": [optimizer_state_dict(net, optimizer)] if optimizer is not None else [],
        "lr_schedulers": [lr_scheduler_state] if lr_scheduler_state is not None else [],
    }
    if extra:
        ckpt.update(extra)
    torch.save(ckpt, path)
    return ckpt


def load_checkpoint(path: str, net, optimizer=None, strict: bool = True) -> Dict:
    """Loads the network (and optimizer) from a reference-layout `.ckpt`; returns the remaining entries."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"]
    net_sd = {k[len(NET_PREFIX):]: v for k, v in sd.items() if k.startswith(NET_PREFIX)}
    if not net_sd:                      # a bare network state_dict (README-style weight files)
        net_sd = sd
    net.load_state_dict(net_sd, strict=strict)
    if optimizer is not None and ckpt.get("optimizer_states"):
        load_optimizer_state_dict(net, optimizer, ckpt["optimizer_states"][0])
    return {k: v for k, v in ckpt.items() if k not in ("state_dict", "optimizer_states")}
