"""Oracle: target assignment, CIoU, YOLOv5 loss, decode, NMS (plain torch, CPU).

TEST INFRASTRUCTURE - see oracle/__init__.py.  Restates:

* assigner            kod/core/label_assignment/yv5.py:45-319
* IoU family          kod/core/bbox/iou.py:40-246
* loss                kod/lightning/experiments/yv5_baseline/loss.py:25-248
* train-step scalar   kod/lightning/experiments/yv5_baseline/exp.py:104-138
* decode              kod/lightning/experiments/yv5_baseline/layers.py:15-155
* NMS                 kod/core/nms.py:9-75 (+ torchvision 0.15.2 ops.nms: greedy,
                      score-descending, suppress IoU > thr, no +1 in areas)
"""
from __future__ import annotations

import math
from typing import NamedTuple, Sequence

import torch
import torch.nn.functional as F

# anchors: kod/configs/anchor_boxes/voc_s{8,16,32}.yaml == kod/test_utils/anchor_boxes.py:6-31
ANCHORS = {8: ((10, 13), (16, 30), (33, 23)),
           16: ((30, 61), (62, 45), (59, 119)),
           32: ((116, 90), (156, 198), (373, 326))}
STRIDES = (8, 16, 32)
OBJ_BALANCE = (4.0, 1.0, 0.4)             # kod/configs/nn/losses/yv5.yaml:11-16
LAMBDA_CLS, LAMBDA_BOX, LAMBDA_OBJ = 0.5, 0.05, 1.0


class Target(NamedTuple):                 # kod/data/detection.py:24-26
    boxes: torch.Tensor                   # [n,4] xyxy pixels (f64 in the reference)
    labels: torch.Tensor                  # [n] int64


class Assigned(NamedTuple):               # yv5.py:25-37 flattened
    samples: torch.Tensor
    anchors_idx: torch.Tensor
    grid_y: torch.Tensor
    grid_x: torch.Tensor
    labels: torch.Tensor
    gt_boxes: torch.Tensor                # [m,4] (cx - cell, cy - cell, w, h) grid units
    anchors: torch.Tensor                 # [m,2] anchor wh in grid units
    fw: int
    fh: int


def assign_level(img_w: int, img_h: int, targets: Sequence[Target], stride: int,
                 anchors_px=None, threshold: float = 4.0) -> Assigned:
    """One pyramid level of Yolov5LabelAssigner (yv5.py:207-296)."""
    anchors_px = ANCHORS[stride] if anchors_px is None else anchors_px
    rows = []
    for i, t in enumerate(targets):                                  # yv5.py:85-121
        n = t.boxes.shape[0]
        r = torch.zeros((n, 6), dtype=torch.float32)
        if n:
            b = t.boxes
            cxcywh = torch.stack(((b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2,
                                  b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]), -1) * (1 / stride)
            r[:, 2:6] = cxcywh
            r[:, 1] = t.labels
            r[:, 0] = i
        rows.append(r)
    T = torch.cat(rows, 0) if rows else torch.zeros((0, 6))
    A = torch.tensor([(w * 1 / stride, h * 1 / stride) for w, h in anchors_px],
                     dtype=torch.float32)                            # yv5.py:226-238
    na = A.shape[0]
    # anchor-major replication + ratio filter (yv5.py:123-176)
    rep = torch.cat([T[None].expand(na, -1, -1),
                     torch.arange(na, dtype=torch.float32)[:, None, None].expand(-1, T.shape[0], 1)], -1)
    ratio = rep[..., 4:6] / A[:, None, :]
    keep = torch.max(ratio, 1.0 / ratio).max(2).values < threshold
    Fm = rep[keep]                                                   # [n,7], anchor-major then target order
    # neighbour cells (yv5.py:178-205)
    fmap = torch.tensor([img_w / stride, img_h / stride], dtype=torch.float32)
    g = Fm[:, 2:4]
    gi = fmap - g
    jk = (g % 1 < 0.5) & (g > 1)
    lm = (gi % 1 < 0.5) & (gi > 1)
    masks = [torch.ones(Fm.shape[0], dtype=torch.bool), jk[:, 0], jk[:, 1], lm[:, 0], lm[:, 1]]
    offs = [(0.0, 0.0), (0.5, 0.0), (0.0, 0.5), (-0.5, 0.0), (0.0, -0.5)]
    sel = torch.cat([Fm[m] for m in masks], 0)
    off = torch.cat([torch.tensor(o, dtype=torch.float32).expand(int(m.sum()), 2)
                     for m, o in zip(masks, offs)], 0)
    cxcy, wh = sel[:, 2:4], sel[:, 4:6]
    gij = (cxcy - off).long()                                        # yv5.py:259 (truncation)
    fw, fh = img_w // stride, img_h // stride
    aidx = sel[:, 6].long()
    return Assigned(samples=sel[:, 0].long(), anchors_idx=aidx,
                    grid_y=gij[:, 1].clamp(0, fh - 1), grid_x=gij[:, 0].clamp(0, fw - 1),
                    labels=sel[:, 1].long(), gt_boxes=torch.cat((cxcy - gij, wh), 1),
                    anchors=A[aidx], fw=fw, fh=fh)


def assign(img_w: int, img_h: int, targets: Sequence[Target], threshold: float = 4.0):
    return tuple(assign_level(img_w, img_h, targets, s, threshold=threshold) for s in STRIDES)


# ----------------------------------------------------------------------------- IoU
def _xyxy(b):
    return b.unbind(-1)


def iou_family(b1: torch.Tensor, b2: torch.Tensor, kind: str = "ciou", eps: float = 1e-7):
    """Aligned IoU / GIoU / DIoU / CIoU (iou.py:77-95,136-246)."""
    x1, y1, x2, y2 = _xyxy(b1)
    x1g, y1g, x2g, y2g = _xyxy(b2)
    inter = (torch.min(x2, x2g) - torch.max(x1, x1g)).clamp(0) * \
            (torch.min(y2, y2g) - torch.max(y1, y1g)).clamp(0)
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / (union + eps)
    if kind == "iou":
        return iou
    cw = torch.max(x2, x2g) - torch.min(x1, x1g)
    chh = torch.max(y2, y2g) - torch.min(y1, y1g)
    if kind == "giou":
        area = cw * chh
        return iou - torch.abs(area - union) / torch.abs(area + eps)
    diag = cw ** 2 + chh ** 2
    dist = ((x1 + x2) / 2 - (x1g + x2g) / 2) ** 2 + ((y1 + y2) / 2 - (y1g + y2g) / 2) ** 2
    D = dist / (diag + eps)
    if kind == "diou":
        return iou - D
    w1, h1, w2, h2 = x2 - x1, y2 - y1, x2g - x1g, y2g - y1g
    v = (4 / math.pi ** 2) * (torch.atan(w2 / (h2 + eps)) - torch.atan(w1 / (h1 + eps))) ** 2
    with torch.no_grad():
        alpha = v / ((1 - iou) + v + eps)
    return iou - D - alpha * v


def _to_xyxy(c):
    cx, cy, w, h = c.unbind(-1)
    return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), -1)


# ----------------------------------------------------------------------------- loss
class LossOut(NamedTuple):                # type_defs.py:28-31
    localization: torch.Tensor
    objectness: torch.Tensor
    classification: torch.Tensor


def level_losses(box, obj, cls, a: Assigned, balance: float, pos_weight=None, iou_kind="ciou", iou_eps: float = 1e-7):
    """loss.py:65-164 for one level; returns (box_mean, obj_scaled, cls_mean, iou)."""
    idx = (a.samples, a.anchors_idx, a.grid_y, a.grid_x)
    p = box[idx]
    pxy = p[:, :2].sigmoid() * 2 - 0.5
    pwh = (p[:, 2:4].sigmoid() * 2) ** 2 * a.anchors
    iou = iou_family(_to_xyxy(torch.cat((pxy, pwh), 1)), _to_xyxy(a.gt_boxes), iou_kind, iou_eps).squeeze()
    l_box = (1 - iou).mean()
    tobj = torch.zeros_like(obj).squeeze(-1)
    tobj[idx] = iou.clamp(0).type(tobj.dtype)            # NOT detached (loss.py:113-118)
    l_obj = balance * F.binary_cross_entropy_with_logits(obj, tobj.unsqueeze(-1), reduction="mean")
    pc = cls[idx]
    onehot = torch.zeros_like(pc)
    onehot[range(onehot.shape[0]), a.labels] = 1
    l_cls = F.binary_cross_entropy_with_logits(pc, onehot, reduction="mean", pos_weight=pos_weight)
    return l_box, l_obj, l_cls, iou


def yolo_loss(img_w: int, img_h: int, net_out, targets: Sequence[Target],
              pos_weight=None, iou_kind="ciou", iou_eps: float = 1e-7) -> LossOut:
    """Yolov5Loss.forward (loss.py:166-248)."""
    asg = assign(img_w, img_h, targets)
    lb = lo = lc = 0.0
    for head, a, bal in zip(net_out, asg, OBJ_BALANCE):
        b, o, c, _ = level_losses(head.box, head.obj, head.cls, a, bal, pos_weight, iou_kind, iou_eps)
        lb, lo, lc = lb + b, lo + o, lc + c
    nc = net_out[0].cls.shape[-1]
    return LossOut(LAMBDA_BOX * lb, LAMBDA_OBJ * (img_w / 640) ** 2 * lo, LAMBDA_CLS * (nc / 80) * lc)


def train_step_total(loss: LossOut, batch_size: int) -> torch.Tensor:
    """exp.py:124-130."""
    return batch_size * (loss.localization + loss.classification + loss.objectness)


# ----------------------------------------------------------------------------- decode + NMS
def decode(net_out, img_w: int, img_h: int) -> torch.Tensor:
    """get_detections (exp.py:70-102, layers.py:55-63,143-153) -> [B, sum(A*h*w), 5+nc]."""
    boxes, objs, clss = [], [], []
    for head, s in zip(net_out, STRIDES):
        fh, fw = img_h // s, img_w // s
        yv, xv = torch.meshgrid(torch.arange(fh).float(), torch.arange(fw).float(), indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, 1, fh, fw, 2)
        anc = torch.tensor(ANCHORS[s], dtype=torch.float32).view(1, -1, 1, 1, 2)
        b = head.box
        xy = (b[..., 0:2].sigmoid() * 2 + grid - 0.5) * s
        wh = (b[..., 2:4].sigmoid() * 2) ** 2 * anc
        B = b.shape[0]
        boxes.append(_to_xyxy(torch.cat((xy, wh), -1).reshape(B, -1, 4)))
        objs.append(head.obj.sigmoid().reshape(B, -1, 1))
        clss.append(head.cls.sigmoid().reshape(B, -1, head.cls.shape[-1]))
    return torch.cat((torch.cat(boxes, 1), torch.cat(objs, 1), torch.cat(clss, 1)), -1)


def greedy_nms(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    """torchvision.ops.nms semantics (stable score-descending order)."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    n = b.shape[0]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            r = b[i + 1:]
            w = (torch.min(b[i, 2], r[:, 2]) - torch.max(b[i, 0], r[:, 0])).clamp(0)
            h = (torch.min(b[i, 3], r[:, 3]) - torch.max(b[i, 1], r[:, 1])).clamp(0)
            inter = w * h
            iou = inter / (area[i] + area[i + 1:] - inter)
            dead[i + 1:] |= iou > thr
    return order[torch.tensor(keep, dtype=torch.long)]


def nms(det: torch.Tensor, conf_thres: float = 0.25, nms_thres: float = 0.45):
    """non_max_suppression (nms.py:9-75), multi-label path (nc > 1) and best-class path."""
    nc = det.shape[2] - 5
    out = []
    for x in det:
        x = x[x[:, 4] > conf_thres]
        if not x.shape[0]:
            out.append(torch.zeros((0, 6)))
            continue
        x = x.clone()
        x[:, 5:] *= x[:, 4:5]
        box = x[:, :4]
        if nc > 1:
            i, j = (x[:, 5:] > conf_thres).nonzero(as_tuple=False).T
            x = torch.cat((box[i], x[i, j + 5, None], j[:, None].float()), 1)
        else:
            conf, j = x[:, 5:].max(1, keepdim=True)
            x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
        if not x.shape[0]:
            out.append(torch.zeros((0, 6)))
            continue
        if x.shape[0] > 30000:
            x = x[x[:, 4].argsort(descending=True)[:30000]]
        k = greedy_nms(x[:, :4] + x[:, 5:6] * 4096, x[:, 4], nms_thres)[:300]
        out.append(x[k])
    return out
