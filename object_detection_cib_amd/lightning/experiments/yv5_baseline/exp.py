"""DefaultYolov5Experiment without Lightning: the arithmetic and call order of
kod/lightning/experiments/yv5_baseline/exp.py:36-185 (training_step, validation_step, configure_optimizers,
on_before_optimizer_step) plus the epoch loop Lightning's Trainer.fit provides (automatic optimisation:
zero_grad -> backward -> warm-up hook -> optimizer.step; LambdaLR.step per epoch).  `lightning` is not
installed in the target image; a LightningModule shell can wrap these methods one to one.
"""
from __future__ import annotations

import contextlib
import gc
from functools import partial
from typing import Callable, Optional, Sequence

import torch

from .... import _lib
from ....core.types import FeatureShape
from ....core.nms import non_max_suppression
from ....nn.optim.smart import SmartOptimizer, FusedSGD
from ....nn.optim.schedulers import LinearScheduler
from ...callbacks.map_eval import DeviceMAPEvaluator
from .layers import get_detections
from .type_defs import LayerwiseAnchorInfo
from .warmup import OptimizerWarmupUpdater


@contextlib.contextmanager
def _gc_paused():
    """The epoch loops run at ~10 ms per batch; a generation-2 collection of the interpreter (triggered every few
    batches by the many small host objects a batch creates) walks the whole heap and stalls the loop for 60-80 ms -
    measured: validation 32 ms/batch with the collector on, 10 ms with it paused.  Collect once, before and after."""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()
        gc.collect()            # reference cycles built while paused (they may hold device tensors) go now


class DefaultYolov5Experiment:
    """Constructor = the reference's (exp.py:37-58: net, loss, anchor_info, smart_optimizer, lr_scheduler,
    optimizer_warmup_updater, val_nms_conf_threshold, val_nms_iou_threshold); the keyword-only extras stand in for
    what Lightning's Trainer supplies (`trainer.max_epochs`) or are build-side switches.  smart_optimizer /
    lr_scheduler default to the reference's configs (configs/nn/optimizers/smart_sgd.yaml, schedulers/linear.yaml)."""

    def __init__(self, net, loss, anchor_info: LayerwiseAnchorInfo, smart_optimizer: Optional[SmartOptimizer] = None,
                 lr_scheduler: Optional[Callable] = None,
                 optimizer_warmup_updater: Optional[OptimizerWarmupUpdater] = None,
                 val_nms_conf_threshold: float = 0.001, val_nms_iou_threshold: float = 0.6, *, max_epochs: int = 300,
                 graphed: bool = False, max_targets: int = 4096, world_size: int = 1):
        self.net, self.loss, self.anchor_info = net, loss, anchor_info
        self.smart_optimizer = smart_optimizer or SmartOptimizer(
            partial(torch.optim.SGD, lr=0.01, momentum=0.937, nesterov=True), weight_decay=0.0005)
        self.lr_scheduler = lr_scheduler or partial(LinearScheduler, lrf=0.01)
        self.max_epochs = max_epochs
        self.optimizer_warmup_updater = optimizer_warmup_updater
        self.val_nms_conf_threshold = val_nms_conf_threshold
        self.val_nms_iou_threshold = val_nms_iou_threshold
        self.global_step = 0
        self.current_epoch = 0
        self.logged = {}
        self.world_size = world_size
        self.optimizer = self.scheduler = None
        # graphed=True: the optimisation step is captured once as a hipGraph (engine/graphed.py) and replayed - same
        # arithmetic, no per-launch Python; batches must keep one shape and at most max_targets boxes
        self.graphed, self.max_targets, self._gstep = graphed, max_targets, None
        self._geval = {}              # input shape -> GraphedEvalForward
        # cross-rank validation: "auto" (default) = the group the network was made data-parallel over
        # (Yolov5Network.configure_distributed) with sync "mean", i.e. what the reference logs under DDP
        # (`log_dict(results, sync_dist=True)`, pycoco_map_eval.py:139-142); without such a group every rank reports its
        # own shard.  None = never a collective (rank-0-only validation cannot hang on its peers); an explicit process
        # group + "mean" / "global" = DeviceMAPEvaluator.get_report's two DDP modes
        self.val_process_group, self.val_sync = "auto", "mean"

    # exp.py:156-162
    def configure_optimizers(self):
        if self.optimizer is None:
            self.optimizer = self.smart_optimizer(self.net)
            if isinstance(self.optimizer, FusedSGD):
                self.optimizer.world_size = self.world_size
            self.scheduler = self.lr_scheduler(optimizer=self.optimizer, max_epochs=self.max_epochs)
        return [self.optimizer], [self.scheduler]

    @property
    def sch_fn(self):
        self.configure_optimizers()
        return self.scheduler.sch_fn

    def get_metrics_to_display(self):
        return ["box", "cls", "obj"]

    # exp.py:104-138
    def training_step(self, batch, batch_idx: int = 0):
        images, targets, _ = batch
        net_result = self.net(images)
        shape = FeatureShape(width=images.shape[3], height=images.shape[2])
        lr = self.loss(shape, net_result, targets)
        B = images.shape[0]
        total = B * (lr.localization + lr.classification + lr.objectness)
        self.logged = {"obj": lr.objectness.detach(), "cls": lr.classification.detach(), "box": lr.localization.detach()}
        return total

    # exp.py:140-154
    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int = 0):
        images, targets, _ = batch
        if self.graphed:          # eval forward + decode replayed as one hipGraph per input shape (engine/graphed.py)
            from ....engine.graphed import GraphedEvalForward
            key = tuple(images.shape)
            if key not in self._geval:
                self._geval[key] = GraphedEvalForward(self.net, self.anchor_info, key[0], key[2], key[3])
            det = self._geval[key](images if images.dtype == torch.float32 else images.float())
            return targets, non_max_suppression(det, self.val_nms_conf_threshold, self.val_nms_iou_threshold)
        was = self.net.training
        self.net.eval()
        res = self.net(images)
        det = get_detections(FeatureShape(width=images.shape[3], height=images.shape[2]), res, self.anchor_info)
        out = non_max_suppression(det, self.val_nms_conf_threshold, self.val_nms_iou_threshold)
        self.net.train(was)
        return targets, out

    # exp.py:164-185 + Lightning automatic optimisation
    def optimize(self, batch, num_training_batches: int):
        self.configure_optimizers()
        if self.graphed:
            return self._optimize_graphed(batch, num_training_batches)
        self.optimizer.zero_grad(set_to_none=True)
        total = self.training_step(batch)
        total.backward()
        self._warmup(num_training_batches)
        self.optimizer.step()
        self.global_step += 1
        return total

    def _warmup(self, num_training_batches: int):
        if self.optimizer_warmup_updater is not None:
            nw = max(round(num_training_batches * self.optimizer_warmup_updater.warmup_epochs), 100)
            if self.global_step <= nw:
                self.optimizer_warmup_updater(current_step=self.global_step, current_epoch=self.current_epoch,
                                              max_warmup_steps=nw, sch_fn=self.sch_fn, optimizer=self.optimizer)

    def _optimize_graphed(self, batch, num_training_batches: int):
        from ....engine.graphed import GraphedTrainStep
        images, targets, _ = batch
        if self._gstep is None:
            B, _, H, W = images.shape
            self._gstep = GraphedTrainStep(self.net, self.loss, B, H, W, self.max_targets).capture(images, targets)
        self._warmup(num_training_batches)
        lr, mom, wd = self.optimizer.hyper()
        total, (box, obj, cls) = self._gstep(images, targets, lr, mom, wd, 1.0 / self.optimizer.world_size)
        self.optimizer.steps_taken += 1
        # the replayed graph contains the SGD update: tell torch's bookkeeping that a step happened (LambdaLR.step() checks
        # optimizer._step_count to warn about a scheduler stepped before the optimizer)
        self.optimizer._step_count = getattr(self.optimizer, "_step_count", 0) + 1
        self.logged = {"obj": obj, "cls": cls, "box": box}
        self.global_step += 1
        return total.clone()

    def end_epoch(self):
        """Lightning steps the epoch-interval scheduler: LambdaLR sets lr = initial_lr * lr_lambda(epoch)."""
        self.configure_optimizers()
        self.current_epoch += 1
        self.scheduler.step()

    def fit_epoch(self, batches: Sequence, num_training_batches: Optional[int] = None):
        n = num_training_batches or len(batches)
        _lib.limit_host_threads()      # the epoch loops are the library's own: size torch's host pool to the cgroup share
        with _gc_paused():
            losses = [self.optimize(b, n).detach() for b in batches]
        self.end_epoch()
        return torch.stack(losses)

    def validate(self, batches: Sequence, num_classes: int, class_names=None) -> dict:
        ev = DeviceMAPEvaluator(num_classes, class_names)
        _lib.limit_host_threads()
        with _gc_paused():
            for b in batches:
                targets, dets = self.validation_step(b)
                ev.add_batch(targets, dets)
        pg = self._val_group()
        if pg is not None and self.val_process_group == "auto":
            self._all_ranks_validate(pg)
        return ev.get_report(pg, self.val_sync)

    def _all_ranks_validate(self, pg, timeout_s: float = 300.0):
        """The default cross-rank report ("auto") is a collective: every rank of the data-parallel group must call
        validate().  A caller that validates on a subset of ranks (set val_process_group = None for that) would otherwise
        hang in the report's all-reduce with no message; where the group's backend can time a barrier (gloo) it is
        checked here first and the error says what to change."""
        import datetime
        import torch.distributed as dist
        if dist.get_backend(pg) != "gloo":
            return
        try:
            dist.monitored_barrier(pg, timeout=datetime.timedelta(seconds=timeout_s))
        except RuntimeError as e:
            raise RuntimeError("validate(): not every rank of the data-parallel group entered validation within "
                               f"{timeout_s:.0f} s. The default val_process_group='auto' averages the report over that group "
                               "(the reference's log_dict(sync_dist=True)); to validate on a subset of ranks set "
                               "experiment.val_process_group = None.") from e

    def _val_group(self):
        pg = self.val_process_group
        if isinstance(pg, str) and pg == "auto":
            eng = getattr(self.net, "_engine", None)
            if eng is None or getattr(eng, "world_size", 1) <= 1:
                return None
            import torch.distributed as dist
            return eng.process_group if eng.process_group is not None else dist.group.WORLD
        return pg
