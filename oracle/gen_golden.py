"""Generate tests/golden/*.npz by running the REAL reference (``kod``) here.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

Each fixture is data only: seeded inputs and the reference's outputs.  The
synthetic inputs are produced by ``oracle.synth`` so tests can regenerate the
exact same inputs without the reference.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch

from oracle import ref_import as R
from oracle import synth

# `python -m oracle.gen_golden` rewrites tests/golden/ in place; KOD_GOLDEN_OUT=<dir> writes elsewhere (the CPU test
# tests/test_oracle_golden.py::test_fixtures_regenerate_from_the_reference regenerates into a temp dir and diffs)
OUT = os.environ.get("KOD_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def _ref_anchor_info():
    FS = R.ref("kod.core.types").FeatureShape
    AI = R.ref("kod.core.anchors.info").AnchorBoxInfo
    from oracle.detection import ANCHORS
    return {s: AI(stride=s, boxes_wh=[FS(width=w, height=h) for w, h in ANCHORS[s]]) for s in (8, 16, 32)}


def _ref_assigner():
    ya = R.ref("kod.core.label_assignment.yv5")
    ai = _ref_anchor_info()
    return ya.Yolov5LabelAssigner(ya.AssignmentAnchorInfo(ll=ai[8], ml=ai[16], hl=ai[32]), threshold=4.0)


def _ref_targets(tg):
    DT = R.ref("kod.data.detection").DetectionTarget
    return tuple(DT(boxes=b, labels=l) for b, l in tg)


def gen_iou():
    iou = R.ref("kod.core.bbox.iou")
    g = torch.Generator().manual_seed(7)
    xy = torch.rand(64, 2, generator=g) * 10
    b1 = torch.cat((xy, xy + torch.rand(64, 2, generator=g) * 8 + 0.05), 1)
    xy2 = xy + torch.randn(64, 2, generator=g) * 2
    b2 = torch.cat((xy2, xy2 + torch.rand(64, 2, generator=g) * 8 + 0.05), 1)
    b1[:3] = torch.tensor([(0, 0, 2, 2), (1, 1, 4, 3), (.5, .5, 1.5, 2.5)])
    b2[:3] = torch.tensor([(1, 1, 3, 3), (0, 0, 2, 4), (.5, .5, 1.5, 2.5)])
    out = {"b1": _np(b1), "b2": _np(b2)}
    for kind in ("iou", "giou", "diou", "ciou"):
        out[kind] = _np(iou.IoUCalculator(iou.IoUType(kind), 1e-7)(b1, b2))
    b1g = b1.clone().requires_grad_(True)
    iou.compute_ciou(b1g, b2).sum().backward()
    out["ciou_grad_b1"] = _np(b1g.grad)
    # backward of every kind w.r.t. both box tensors under a seeded (non-uniform) upstream gradient
    gout = torch.randn(64, generator=torch.Generator().manual_seed(8))
    out["gout"] = _np(gout)
    for kind in ("iou", "giou", "diou", "ciou"):
        a, b = b1.clone().requires_grad_(True), b2.clone().requires_grad_(True)
        (iou.IoUCalculator(iou.IoUType(kind), 1e-7)(a, b) * gout).sum().backward()
        out[f"{kind}_gw_b1"], out[f"{kind}_gw_b2"] = _np(a.grad), _np(b.grad)      # d sum(out * gout) / d boxes
    np.savez(os.path.join(OUT, "iou.npz"), **out)


def gen_samplers():
    """kod/data/samplers.py:41-138 on a synthetic DatasetInfo (oracle.synth.dataset_info): the ClassAware index
    stream under torch.manual_seed(2023), repeat factors (mean / max / no sqrt / threshold) and RepeatFactor draws."""
    import datetime
    S = R.ref("kod.data.samplers")
    C = R.ref("kod.data.cache")
    B = R.ref("kod.core.bbox.boxes")
    spec = synth.dataset_info_spec(48, 6, seed=5)
    meta = C.ImageMetadata(width=64, height=48, num_channels=3, mime_type="image/jpeg", size_bytes=1)
    samples = [C.SampleInfo(id=sid, image_path=f"/nowhere/{sid}.jpg", image_metadata=meta,
                            targets=[C.TargetInfo(bounding_box=B.XYXYBoundingBox(*bb), class_name=cn) for bb, cn in tg])
               for sid, tg in spec["samples"]]
    ds = C.DatasetInfo(name="synthetic", date=datetime.datetime(2023, 1, 1), classes=list(spec["classes"]), samples=samples)
    out = {}
    torch.manual_seed(2023)
    cas = S.ClassAwareSampler(ds)
    out["class_aware_epoch0"] = np.array(list(iter(cas)), dtype=np.int64)
    out["class_aware_epoch1"] = np.array(list(iter(cas)), dtype=np.int64)
    import contextlib, io
    for tag, kw in (("mean", dict()), ("max", dict(reduction="max")), ("nosqrt", dict(use_sqrt=False)),
                    ("thr05", dict(threshold=0.5))):
        with contextlib.redirect_stdout(io.StringIO()):
            rfs = S.RepeatFactorSampler(ds, **kw)
        out[f"repeat_factors_{tag}"] = np.array(rfs.image_repeat_factors, dtype=np.float64)
        out[f"repeat_draws_{tag}"] = np.array(list(iter(rfs)), dtype=np.int64)
    torch.manual_seed(7)
    rc = S.RandomCycleSampler([10, 11, 12, 13, 14])
    out["random_cycle"] = np.array([next(rc) for _ in range(13)], dtype=np.int64)
    out["instance_count"] = np.array(list(ds.get_instance_count().values()), dtype=np.int64)
    flt = R.ref("kod.data.filter").filter_dataset(ds, "sub", [spec["classes"][1], spec["classes"][3]])
    out["filter_ids"] = np.array([int(s.id) for s in flt.samples], dtype=np.int64)
    out["filter_ntargets"] = np.array([len(s.targets) for s in flt.samples], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "samplers.npz"), **out)


def gen_assigner():
    FS = R.ref("kod.core.types").FeatureShape
    asg = _ref_assigner()
    out = {}
    for name, (size, tg) in synth.assigner_cases().items():
        res = asg(FS(width=size, height=size), _ref_targets(tg))
        for lvl, r in zip(("ll", "ml", "hl"), res):
            p = f"{name}.{lvl}."
            out[p + "samples"] = _np(r.indices.samples)
            out[p + "anchors_idx"] = _np(r.indices.anchors)
            out[p + "grid_y"] = _np(r.indices.grid_y)
            out[p + "grid_x"] = _np(r.indices.grid_x)
            out[p + "labels"] = _np(r.labels)
            out[p + "gt_boxes"] = _np(r.gt_boxes)
            out[p + "anchors"] = _np(r.anchors)
    np.savez_compressed(os.path.join(OUT, "assigner.npz"), **out)


def _ref_loss(weights=None):
    L = R.ref("kod.lightning.experiments.yv5_baseline.loss")
    iou = R.ref("kod.core.bbox.iou")
    return L.Yolov5Loss(_ref_assigner(), L.Yolov5LossParams.get_default(),
                        iou.IoUCalculator(iou.IoUType.ciou, 1e-7), weights)


def _ref_netresult(heads):
    N = R.ref("kod.nn.networks.yolov5")
    H = R.ref("kod.nn.heads.types")
    return N.Yolov5NetworkResult(*[H.DetectionHeadResult(*h) for h in heads])


def gen_loss():
    FS = R.ref("kod.core.types").FeatureShape
    out = {}
    for name, (size, nc, B, tg, weights) in synth.loss_cases().items():
        heads = synth.head_logits(B, size, nc, seed=11)
        leaves = [[t.clone().requires_grad_(True) for t in h] for h in heads]
        res = _ref_loss(weights)(FS(width=size, height=size), _ref_netresult(leaves), _ref_targets(tg))
        total = B * (res.localization + res.classification + res.objectness)
        p = name + "."
        out[p + "loss"] = np.array([res.localization.item(), res.objectness.item(),
                                    res.classification.item(), total.item()], dtype=np.float64)
        if torch.isfinite(total):
            total.backward()
            for lvl, h in zip(("ll", "ml", "hl"), leaves):
                for nm, t in zip(("box", "obj", "cls"), h):
                    out[p + f"{lvl}.{nm}.grad"] = _np(t.grad).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **out)


def gen_network():
    """Seeded random-init nets: outputs, loss, per-parameter gradient norms, BN buffers."""
    N = R.ref("kod.nn.networks.yolov5")
    FS = R.ref("kod.core.types").FeatureShape
    out = {}
    for name, (widen, deepen, nc, B, size, seed) in synth.network_cases().items():
        torch.manual_seed(seed)
        net = N.Yolov5Network(3, nc, widen_factor=widen, deepen_factor=deepen).train()
        x, tg = synth.batch(B, size, nc, seed)
        res = net(x)
        lr = _ref_loss()(FS(width=size, height=size), res, _ref_targets(tg))
        total = B * (lr.localization + lr.classification + lr.objectness)
        total.backward()
        p = name + "."
        out[p + "loss"] = np.array([lr.localization.item(), lr.objectness.item(),
                                    lr.classification.item(), total.item()], dtype=np.float64)
        names = [k for k, _ in net.named_parameters()]
        out[p + "param_names"] = np.array(names)
        out[p + "grad_norms"] = np.array([v.grad.double().norm().item() for _, v in net.named_parameters()])
        out[p + "grad_sums"] = np.array([v.grad.double().sum().item() for _, v in net.named_parameters()])
        out[p + "param_norms"] = np.array([v.double().norm().item() for _, v in net.named_parameters()])
        sd = net.state_dict()
        rm = [k for k in sd if k.endswith("running_mean")]
        out[p + "running_mean_norms"] = np.array([sd[k].double().norm().item() for k in rm])
        out[p + "running_var_norms"] = np.array([sd[k.replace("_mean", "_var")].double().norm().item() for k in rm])
        for lvl, h in zip(("ll", "ml", "hl"), res):
            for nm, t in zip(("box", "obj", "cls"), h):
                t = t.detach()
                out[p + f"{lvl}.{nm}.stats"] = np.array([t.double().sum().item(), t.double().abs().sum().item(),
                                                         t.double().pow(2).sum().item()])
                if size <= 64:
                    out[p + f"{lvl}.{nm}"] = _np(t.contiguous()).astype(np.float32)
        if size <= 64:
            g0 = dict(net.named_parameters())["backbone.stem.0.weight"].grad
            out[p + "stem_weight_grad"] = _np(g0).astype(np.float32)
            out[p + "hl_cls_bias_grad"] = _np(dict(net.named_parameters())["hl_head.cls_head.conv.bias"].grad)
    np.savez_compressed(os.path.join(OUT, "network.npz"), **out)


def gen_decode_nms():
    E = R.ref("kod.lightning.experiments.yv5_baseline.layers")
    nms = R.ref("kod.core.nms")
    ai = _ref_anchor_info()
    FS = R.ref("kod.core.types").FeatureShape
    out = {}
    for name, (size, nc, B, seed, scale) in synth.decode_cases().items():
        heads = synth.head_logits(B, size, nc, seed=seed, scale=scale)
        preds = [E.Yolov5Prediction(stride=s, image_feature_shape=FS(width=size, height=size),
                                    anchor_box_shapes=ai[s].boxes_wh)(*[t.clone() for t in h])
                 for s, h in zip((8, 16, 32), heads)]
        det = E.Yolov5PredictionAssembler()([p.box for p in preds], [p.obj for p in preds],
                                            [p.cls for p in preds])
        p = name + "."
        out[p + "det"] = _np(det).astype(np.float32)
        for conf, thr in ((0.001, 0.6), (0.25, 0.45)):
            res = nms.non_max_suppression(det.clone(), conf_thres=conf, nms_thres=thr)
            out[p + f"nms_{conf}_{thr}.counts"] = np.array([r.shape[0] for r in res])
            out[p + f"nms_{conf}_{thr}.rows"] = _np(torch.cat(list(res), 0)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "decode_nms.npz"), **out)


def gen_optim():
    W = R.ref("kod.lightning.experiments.yv5_baseline.warmup")
    S = R.ref("kod.nn.optim.smart")
    sch = R.ref("kod.nn.optim.schedulers")
    from functools import partial
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, bias=True), torch.nn.BatchNorm2d(4),
                              torch.nn.Conv2d(4, 2, 1, bias=False))
    opt = S.SmartOptimizer(partial(torch.optim.SGD, lr=0.01, momentum=0.937, nesterov=True), 0.0005)(net)
    for pg in opt.param_groups:
        pg["initial_lr"] = pg["lr"]
    fn = partial(sch.sch_linear, max_epochs=300, lrf=0.01)
    upd = W.OptimizerWarmupUpdater(3, 0.1, 0.8, 0.937)
    out = {"group_names": np.array([pg["name"] for pg in opt.param_groups]),
           "group_sizes": np.array([len(pg["params"]) for pg in opt.param_groups]),
           "group_wd": np.array([pg["weight_decay"] for pg in opt.param_groups]),
           "sch_linear": np.array([fn(e) for e in (0, 1, 150, 299)])}
    steps = [0, 1, 10, 330, 660]
    lrs, moms = [], []
    for st in steps:
        upd(current_step=st, current_epoch=st // 220, max_warmup_steps=660, sch_fn=fn, optimizer=opt)
        lrs.append([pg["lr"] for pg in opt.param_groups])
        moms.append([pg["momentum"] for pg in opt.param_groups])
    out["warmup_steps"], out["warmup_lr"], out["warmup_momentum"] = np.array(steps), np.array(lrs), np.array(moms)
    # 5-step trajectory with warm-up (nw = 100), fixed synthetic gradients
    params = [p for pg in opt.param_groups for p in pg["params"]]
    out["traj_p0"] = np.concatenate([_np(p).ravel() for p in params])
    g = torch.Generator().manual_seed(5)
    grads = []
    for st in range(5):
        upd(current_step=st, current_epoch=0, max_warmup_steps=100, sch_fn=fn, optimizer=opt)
        gs = [torch.randn(p.shape, generator=g) for p in params]
        for p, gg in zip(params, gs):
            p.grad = gg.clone()
        grads.append(np.concatenate([_np(x).ravel() for x in gs]))
        opt.step()
    out["traj_grads"] = np.stack(grads)
    out["traj_p5"] = np.concatenate([_np(p).ravel() for p in params])
    out["traj_sizes"] = np.array([p.numel() for p in params])
    out["traj_group_of_param"] = np.array([gi for gi, pg in enumerate(opt.param_groups) for _ in pg["params"]])
    np.savez_compressed(os.path.join(OUT, "optim.npz"), **out)


def gen_mosaic():
    M = R.ref("kod.data.mosaic")
    AS = R.ref("kod.data.types").AugmentedSample
    out = {}
    for name, (S, seed) in synth.mosaic_cases().items():
        samples = synth.source_samples(4, S, seed)
        random.seed(seed)
        res, border = M.MosaicAugmentor(S)([AS(image=i, bboxes=b, labels=l) for i, b, l in samples])
        p = name + "."
        out[p + "bboxes"], out[p + "labels"] = res.bboxes, res.labels
        out[p + "border"] = np.array(border)
        img = res.image
        out[p + "image_sum"] = np.array([int(img.astype(np.int64).sum())])
        out[p + "image_rowsum"] = img.astype(np.int64).sum(axis=(1, 2))
        out[p + "image_colsum"] = img.astype(np.int64).sum(axis=(0, 2))
        if S <= 64:
            out[p + "image"] = img
    np.savez_compressed(os.path.join(OUT, "mosaic.npz"), **out)


def gen_affine_boxes():
    """Matrix builders + box transform of random_perspective (no cv2 needed, degrees=0)."""
    D = R.ref("kod.data.augmentations.default")
    FS = R.ref("kod.core.types").FeatureShape
    out = {}
    rng = np.random.default_rng(51)
    ap = D.AffineParams()
    vals = [D.get_affine_random_values(ap, rng) for _ in range(6)]
    out["rand_values"] = np.array([list(v) for v in vals])
    S = 64
    fs_in = FS(width=2 * S, height=2 * S)
    fs_out = D._get_feat_shape(width=2 * S, height=2 * S, border=(-S // 2, -S // 2))
    out["feat_shape_out"] = np.array([fs_out.width, fs_out.height])
    mats, boxes_out, keep_out = [], [], []
    boxes = synth.source_samples(1, 2 * S, seed=77)[0][1]
    boxes = np.concatenate([boxes, np.array([[1.0, 2.0, 120.0, 100.0], [60.0, 60.0, 63.5, 90.0]])])
    out["boxes_in"] = boxes
    for v in vals:
        Rm = np.eye(3)
        Rm[0, 0] = Rm[1, 1] = v.scale          # cv2.getRotationMatrix2D(angle=0, center=(0,0), scale)
        M = (D._get_T(v.translate_x, v.translate_y, fs_out) @ D._get_S(v.shear_x, v.shear_y) @ Rm
             @ D._get_P(v.perspective_x, v.perspective_y) @ D._get_C(fs_in))
        mats.append(M)
        pb = D._process_affine_bboxes(bboxes=boxes, M=M, feat_shape=fs_out, perspective=False)
        boxes_out.append(pb)
        keep_out.append(D._box_candidates(orig_bboxes=boxes.T * v.scale, proc_bboxes=pb.T))
    out["matrices"], out["boxes_out"], out["keep"] = np.stack(mats), np.stack(boxes_out), np.stack(keep_out)
    # horizontal flip of boxes + mixup (pure numpy / torch in the reference)
    AS = R.ref("kod.data.types").AugmentedSample
    img = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    fl = D.horizontal_flip(AS(image=img, bboxes=boxes[:3].copy(), labels=np.arange(3)))
    out["flip_boxes"], out["flip_image"] = fl.bboxes, np.ascontiguousarray(fl.image)
    np.random.seed(2023)
    a = AS(torch.arange(24, dtype=torch.float32).reshape(3, 2, 4) / 24, boxes[:2], np.array([1, 2]))
    b = AS(torch.arange(24, dtype=torch.float32).flip(0).reshape(3, 2, 4) / 24, boxes[2:3], np.array([3]))
    mx = D.mixup(a, b)
    out["mixup_image"], out["mixup_boxes"], out["mixup_labels"] = _np(mx.image), mx.bboxes, mx.labels
    np.savez_compressed(os.path.join(OUT, "affine.npz"), **out)


def gen_protocol():
    """The reference's REAL per-sample data path - DetectionDataset.__getitem__ (kod/data/detection.py:102-156) with the
    real MosaicAugmentor and TrainSampleAugmentor(rng_seed=51) (kod/data/augmentations/default.py:411-488) - run for 64
    consecutive samples per configuration over a synthetic pool, with recording stand-ins for cv2 / albumentations
    (oracle/ref_import.py install_recording).  Stored per sample: the indices read (shuffled mosaic partners, mixup
    partners), the mosaic's boxes, every affine matrix and output size handed to cv2.warpAffine (+ the CRC of the canvas it
    was handed), the three HSV look-up tables, the flip outcome, the mixup ratio, final boxes / labels and the CRC of the
    final image (pixels through the oracle's OpenCV restatement: composition pinned, OpenCV's arithmetic not)."""
    import datetime
    import zlib
    rec = R.Recorder()
    R.install_recording(rec)
    D = R.ref("kod.data.augmentations.default")
    Det = R.ref("kod.data.detection")
    Mo = R.ref("kod.data.mosaic")
    C = R.ref("kod.data.cache")
    AS = R.ref("kod.data.types").AugmentedSample
    S, n, N = synth.PROTOCOL_S, synth.PROTOCOL_POOL, synth.PROTOCOL_N
    pool = synth.protocol_pool()
    samples = [C.SampleInfo(id=str(i), image_path=f"/nowhere/{i}.jpg",
                            image_metadata=C.ImageMetadata(width=im.shape[1], height=im.shape[0], num_channels=3,
                                                           mime_type="image/jpeg", size_bytes=1), targets=[])
               for i, (im, _, _) in enumerate(pool)]
    ds_info = C.DatasetInfo(name="synthetic", date=datetime.datetime(2023, 1, 1), classes=[str(c) for c in range(10)], samples=samples)
    flip_fn, beta_fn = D.horizontal_flip, np.random.beta

    def rec_flip(data):
        rec("flip")
        return flip_fn(data)

    def rec_beta(a, b):
        r = beta_fn(a, b)
        rec("beta", r=float(r))
        return r

    def reader(sample, letter_box):
        assert letter_box is False                    # mosaic on => the reader is asked for the un-letter-boxed image
        i = int(sample.id)
        rec("read", index=i)
        im, bb, lb = pool[i]
        return AS(image=im, bboxes=bb.copy(), labels=lb.copy())

    out = {}
    D.horizontal_flip, np.random.beta = rec_flip, rec_beta
    try:
        for name, (mixup_prob, side, over) in synth.PROTOCOL_CASES.items():
            hsv = over.get("hsv", (0.015, 0.7, 0.4))
            params = D.AugParams(affine_params=D.AffineParams(degrees=over.get("degrees", 0.0), shear=over.get("shear", 0.0),
                                                              perspective=over.get("perspective", 0.0)),
                                 hsv_params=D.HSVParams(*hsv), flip_lr_prob=over.get("flip", 0.5),
                                 image_color_transforms=bool(over.get("color", False)))
            rec.color_rng = random.Random(synth.PROTOCOL_COLOR_SEED) if over.get("color") else None
            rec.albu13 = bool(over.get("albu13", False))
            aug = D.TrainSampleAugmentor(params, rng_seed=51)

            def augmentor(sample, border=(0, 0), _aug=aug):
                rec("augment", boxes=np.array(sample.bboxes), labels=np.array(sample.labels), border=tuple(border),
                    canvas_crc=zlib.crc32(np.ascontiguousarray(sample.image).tobytes()))
                return _aug(sample, border)

            sampler = None
            if side:
                w, si = synth.protocol_side_channel()
                sampler = type("SideChannel", (), dict(image_repeat_factors=w, sampler_indices=si))()
            ds = Det.DetectionDataset(ds_info, reader, augmentor, enable_ram_cache=False,
                                      mosaic_augmentor=Mo.MosaicAugmentor(S), mixup_prob=mixup_prob, sampler=sampler)
            random.seed(2023)
            np.random.seed(2023)
            idx = np.full((N, 8), -1, np.int64)
            Ms = np.full((N, 2, 3, 3), np.nan)
            dsize = np.zeros((N, 2, 2), np.int64)
            luts = np.zeros((N, 2, 3, 256), np.uint8)
            n_lut = np.zeros((N, 2), np.int64)
            flips = np.full((N, 2), -1, np.int64)
            rr = np.full(N, np.nan)
            ccrc = np.zeros((N, 2), np.int64)
            icrc = np.zeros(N, np.int64)
            mb, mbn, fb, fl, fn = [], np.zeros((N, 2), np.int64), [], [], np.zeros(N, np.int64)
            col = np.zeros((N, 2, 3), np.int64)                       # colour stage per augmentor call: ops bit mask, blur ksize, median ksize
            col_clip = np.zeros((N, 2))                               # ... CLAHE clip limit
            col_pos = np.full((N, 2), -1, np.int64)                   # ... 1 = the stage ran AFTER the warp and BEFORE the first LUT
            n13 = np.zeros(N, np.int64)                                # albumentations-1.3 mode: gate draws on the global generator per sample
            for k in range(N):
                rec.take()
                smp = ds[k % n]
                ev = rec.take()
                n13[k] = sum(1 for e, _ in ev if e == "albu13_draw")
                assert smp.image_info is None
                reads = [p["index"] for e, p in ev if e == "read"]
                idx[k, :len(reads)] = reads
                stage = -1
                for e, p in ev:
                    if e == "augment":
                        stage += 1
                        assert p["border"] == (-S // 2, -S // 2)
                        mb.append(p["boxes"]); mbn[k, stage] = len(p["boxes"]); ccrc[k, stage] = p["canvas_crc"]
                        flips[k, stage] = 0
                    elif e == "warpAffine":
                        assert p["src_crc"] == ccrc[k, stage] and p["src_shape"] == (2 * S, 2 * S, 3)
                        M = np.eye(3); M[:2] = p["M"]
                        Ms[k, stage] = M; dsize[k, stage] = p["dsize"]
                    elif e == "warpPerspective":
                        assert p["src_crc"] == ccrc[k, stage] and p["src_shape"] == (2 * S, 2 * S, 3) and p["M"].shape == (3, 3)
                        Ms[k, stage] = p["M"]; dsize[k, stage] = p["dsize"]
                    elif e == "LUT":
                        luts[k, stage, n_lut[k, stage]] = p["lut"]; n_lut[k, stage] += 1
                    elif e == "color_stage":
                        assert p["n"] == 4
                        col_pos[k, stage] = int(not np.isnan(Ms[k, stage]).all() and n_lut[k, stage] == 0 and flips[k, stage] == 0)
                    elif e == "color":
                        col[k, stage, 0] |= p["bit"]
                        if p["op"] == "Blur":
                            col[k, stage, 1] = p["ksize"]
                        elif p["op"] == "MedianBlur":
                            col[k, stage, 2] = p["ksize"]
                        elif p["op"] == "CLAHE":
                            col_clip[k, stage] = p["clip_limit"]
                    elif e == "flip":
                        flips[k, stage] = 1
                    elif e == "beta":
                        rr[k] = p["r"]
                img = smp.img.numpy() if hasattr(smp.img, "numpy") else np.asarray(smp.img)
                assert img.dtype == np.float32 and img.shape == (3, S, S)
                icrc[k] = zlib.crc32(np.ascontiguousarray(img).tobytes())
                fb.append(_np(smp.target.boxes)); fl.append(_np(smp.target.labels)); fn[k] = len(fl[-1])
            p = name + "."
            out.update({p + "indices": idx, p + "M": Ms, p + "dsize": dsize, p + "luts": luts, p + "n_lut": n_lut, p + "flip": flips,
                        p + "mixup_r": rr, p + "canvas_crc": ccrc, p + "image_crc": icrc,
                        p + "mosaic_boxes": np.concatenate(mb, 0), p + "mosaic_counts": mbn,
                        p + "boxes": np.concatenate(fb, 0).astype(np.float64), p + "labels": np.concatenate(fl, 0).astype(np.int64),
                        p + "counts": fn})
            if over.get("color"):
                out.update({p + "color": col, p + "color_clip": col_clip, p + "color_pos": col_pos})
            if over.get("albu13"):
                out[p + "legacy_draws"] = n13
    finally:
        D.horizontal_flip, np.random.beta = flip_fn, beta_fn
        rec.albu13 = False
    np.savez_compressed(os.path.join(OUT, "protocol.npz"), **out)


def gen_sppf():
    """kod.nn.layers.sppf.SPPFBottleneck in its three forms (cascade of one kernel size; parallel pools of a kernel-size
    sequence; without the leading conv): seeded weights, output, input gradient and parameter gradients."""
    L = R.ref("kod.nn.layers.sppf")
    N = R.ref("kod.nn.networks.yolov5")
    A = R.ref("kod.nn.layers.activations")
    out = {}
    for name, (cin, cout, ks, first, B, H, W, seed) in synth.sppf_cases().items():
        torch.manual_seed(seed)
        m = L.SPPFBottleneck(cin, cout, kernel_sizes=ks, use_conv_first=first, norm_layer=N.Yolov5BatchNorm2d,
                             activation_layer=A.SiLUInplace).train()
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, cin, H, W, generator=g).requires_grad_(True)
        y = m(x)
        w = torch.randn(y.shape, generator=g)
        (y * w).sum().backward()
        p = name + "."
        out[p + "keys"] = np.array(list(m.state_dict().keys()))
        out[p + "y"], out[p + "dx"] = _np(y), _np(x.grad)
        for k, v in m.named_parameters():
            out[p + "param." + k], out[p + "grad." + k] = _np(v), _np(v.grad)
    np.savez_compressed(os.path.join(OUT, "sppf.npz"), **out)


def main():
    import sys
    assert R.available(), "reference checkout not found"
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])                      # e.g. `python -m oracle.gen_golden gen_iou gen_samplers`
    for fn in (gen_iou, gen_assigner, gen_loss, gen_network, gen_decode_nms, gen_optim, gen_mosaic,
               gen_affine_boxes, gen_samplers, gen_protocol, gen_sppf):
        if only and fn.__name__ not in only:
            continue
        try:
            fn()
            print("ok  ", fn.__name__)
        except Exception as e:                                   # noqa: BLE001
            print("FAIL", fn.__name__, type(e).__name__, e)
            raise


if __name__ == "__main__":
    main()
