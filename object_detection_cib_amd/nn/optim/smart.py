"""SmartSGD - the optimizer the reference assembles from SmartOptimizer + torch.optim.SGD
(kod/nn/optim/smart.py:11-60, kod/configs/nn/optimizers/smart_sgd.yaml) as ONE fused HIP launch over the
engine's flat arenas.  `param_groups` keeps the reference's three named groups (bias_params, decay_params,
norm_params) with mutable lr / momentum so OptimizerWarmupUpdater (warmup.py:39-58) works unchanged."""
from __future__ import annotations


class SmartSGD:
    def __init__(self, net, lr: float = 0.01, momentum: float = 0.937, weight_decay: float = 5e-4,
                 nesterov: bool = True, world_size: int = 1):
        if not nesterov:
            raise NotImplementedError("the fused kernel implements nesterov=True (reference config)")
        self.net = net
        self.world_size = world_size
        self.steps_taken = 0          # torch keeps no momentum_buffer before the first step (checkpoint layout)
        self.param_groups = [
            dict(name="bias_params", lr=lr, initial_lr=lr, momentum=momentum, weight_decay=0.0, nesterov=True),
            dict(name="decay_params", lr=lr, initial_lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=True),
            dict(name="norm_params", lr=lr, initial_lr=lr, momentum=momentum, weight_decay=0.0, nesterov=True),
        ]

    def zero_grad(self, set_to_none: bool = True):
        for p in self.net.parameters():
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self):
        g = self.param_groups
        self.net.engine().sgd_step([float(x["lr"]) for x in g], [float(x["momentum"]) for x in g],
                                   [float(x["weight_decay"]) for x in g], 1.0 / self.world_size)
        self.steps_taken += 1

    def state_dict(self):
        """torch.optim.SGD.state_dict() layout (lightning/checkpoint.py)."""
        from ...lightning.checkpoint import optimizer_state_dict
        return optimizer_state_dict(self.net, self)

    def load_state_dict(self, sd):
        from ...lightning.checkpoint import load_optimizer_state_dict
        load_optimizer_state_dict(self.net, self, sd)
