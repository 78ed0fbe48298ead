"""The CPU oracle against golden vectors produced by the real reference
(oracle/gen_golden.py).  This is what pins the oracle; everything here runs on CPU."""
import random

import numpy as np
import pytest
import torch

from oracle import datapath, detection as D, optim as O, synth
from oracle.network import OracleYolov5, HeadOut, NetOut


def _targets(tg):
    return [D.Target(b, l) for b, l in tg]


def test_iou_family(golden):
    g = golden("iou")
    b1, b2 = torch.from_numpy(g["b1"]), torch.from_numpy(g["b2"])
    for kind in ("iou", "giou", "diou", "ciou"):
        np.testing.assert_array_equal(D.iou_family(b1, b2, kind).numpy(), g[kind])
    # SURVEY Appendix B.2 known answers
    np.testing.assert_allclose(g["ciou"][:3], [0.0317460, 0.0836961, 1.0], atol=1e-6)
    b1g = b1.clone().requires_grad_(True)
    D.iou_family(b1g, b2, "ciou").sum().backward()
    np.testing.assert_array_equal(b1g.grad.numpy(), g["ciou_grad_b1"])


@pytest.mark.parametrize("case", list(synth.assigner_cases()))
def test_assigner(golden, case):
    g = golden("assigner")
    size, tg = synth.assigner_cases()[case]
    res = D.assign(size, size, _targets(tg))
    for lvl, a in zip(("ll", "ml", "hl"), res):
        p = f"{case}.{lvl}."
        for k in ("samples", "anchors_idx", "grid_y", "grid_x", "labels", "gt_boxes", "anchors"):
            np.testing.assert_array_equal(getattr(a, k).numpy(), g[p + k], err_msg=p + k)


def test_assigner_kat_values(golden):
    """SURVEY Appendix B.1 hand-checked values."""
    g = golden("assigner")
    assert g["kat.ll.anchors_idx"].tolist() == [0, 1, 1, 2, 2, 0, 1, 1, 2, 2, 1, 2]
    assert g["kat.hl.grid_x"].tolist() == [6, 13, 6, 13, 6, 5, 12, 5, 12, 5, 6, 13, 6, 13, 6]
    np.testing.assert_allclose(g["kat.ll.gt_boxes"][0], (0.5, 0.125, 2.5, 3.75))


@pytest.mark.parametrize("case", list(synth.loss_cases()))
def test_loss(golden, case):
    g = golden("loss")
    size, nc, B, tg, w = synth.loss_cases()[case]
    heads = [[t.clone().requires_grad_(True) for t in h] for h in synth.head_logits(B, size, nc, seed=11)]
    out = NetOut(*[HeadOut(*h) for h in heads])
    pw = torch.tensor(w) if w is not None else None
    res = D.yolo_loss(size, size, out, _targets(tg), pos_weight=pw)
    total = D.train_step_total(res, B)
    got = np.array([res.localization.item(), res.objectness.item(), res.classification.item(), total.item()])
    np.testing.assert_array_equal(got, g[case + ".loss"])
    if not np.isfinite(got[3]):           # a level without matches -> NaN (reference quirk, loss.py:96)
        assert case in ("nan_level128", "rand64")
        assert np.isnan(got[0]) and np.isnan(got[2]) and np.isfinite(got[1])
        return
    total.backward()
    for lvl, h in zip(("ll", "ml", "hl"), heads):
        for nm, t in zip(("box", "obj", "cls"), h):
            np.testing.assert_array_equal(t.grad.numpy(), g[f"{case}.{lvl}.{nm}.grad"])


@pytest.mark.parametrize("case", list(synth.network_cases()))
def test_network(golden, case):
    g = golden("network")
    widen, deepen, nc, B, size, seed = synth.network_cases()[case]
    torch.manual_seed(seed)
    net = OracleYolov5(3, nc, widen, deepen).train()
    names = [k for k, _ in net.named_parameters()]
    assert names == g[case + ".param_names"].tolist()
    np.testing.assert_array_equal(
        np.array([v.double().norm().item() for v in net.parameters()]), g[case + ".param_norms"])
    x, tg = synth.batch(B, size, nc, seed)
    res = net(x)
    lr = D.yolo_loss(size, size, res, _targets(tg))
    total = D.train_step_total(lr, B)
    total.backward()
    got = np.array([lr.localization.item(), lr.objectness.item(), lr.classification.item(), total.item()])
    np.testing.assert_allclose(got, g[case + ".loss"], rtol=1e-6)
    gn = np.array([v.grad.double().norm().item() for v in net.parameters()])
    np.testing.assert_allclose(gn, g[case + ".grad_norms"], rtol=2e-4, atol=1e-7)
    sd = net.state_dict()
    rm = [k for k in sd if k.endswith("running_mean")]
    np.testing.assert_allclose([sd[k].double().norm().item() for k in rm], g[case + ".running_mean_norms"], rtol=1e-5)
    np.testing.assert_allclose([sd[k.replace("_mean", "_var")].double().norm().item() for k in rm],
                               g[case + ".running_var_norms"], rtol=1e-5)
    if size <= 64:
        for lvl, h in zip(("ll", "ml", "hl"), res):
            for nm, t in zip(("box", "obj", "cls"), h):
                np.testing.assert_allclose(t.detach().numpy(), g[f"{case}.{lvl}.{nm}"], rtol=1e-5, atol=1e-6)


def test_network_census():
    """yv5s nc=10 has 7,046,599 params, 360 state_dict keys, 66 convs."""
    net = OracleYolov5(3, 10, 0.5, 0.33)
    assert sum(p.numel() for p in net.parameters()) == 7046599
    assert len(net.state_dict()) == 360
    assert sum(isinstance(m, torch.nn.Conv2d) for m in net.modules()) == 66
    b, d, n = O.param_groups(net)
    assert (len(b), len(d), len(n)) == (66, 66, 57)


@pytest.mark.parametrize("case", list(synth.decode_cases()))
def test_decode_nms(golden, case):
    g = golden("decode_nms")
    size, nc, B, seed, scale = synth.decode_cases()[case]
    heads = synth.head_logits(B, size, nc, seed=seed, scale=scale)
    det = D.decode(NetOut(*[HeadOut(*h) for h in heads]), size, size)
    np.testing.assert_array_equal(det.numpy(), g[case + ".det"])
    for conf, thr in ((0.001, 0.6), (0.25, 0.45)):
        res = D.nms(det.clone(), conf, thr)
        assert [r.shape[0] for r in res] == g[f"{case}.nms_{conf}_{thr}.counts"].tolist()
        np.testing.assert_array_equal(torch.cat(res, 0).numpy(), g[f"{case}.nms_{conf}_{thr}.rows"])


def test_optim(golden):
    g = golden("optim")
    assert g["group_names"].tolist() == list(O.GROUP_NAMES)
    np.testing.assert_allclose(g["sch_linear"], [O.sch_linear(e) for e in (0, 1, 150, 299)], rtol=0, atol=0)
    for st, lr, mom in zip(g["warmup_steps"], g["warmup_lr"], g["warmup_momentum"]):
        w = O.warmup_values(int(st), int(st) // 220, 660)
        np.testing.assert_array_equal([w[n][0] for n in O.GROUP_NAMES], lr)
        np.testing.assert_array_equal([w[n][1] for n in O.GROUP_NAMES], mom)
    # Appendix B.6 spot values
    np.testing.assert_allclose(g["warmup_lr"][1], [0.0998636, 1.515e-05, 1.515e-05], rtol=1e-3)
    # 5-step trajectory
    sizes, grp = g["traj_sizes"], g["traj_group_of_param"]
    ps = [torch.from_numpy(c.copy()) for c in np.split(g["traj_p0"], np.cumsum(sizes)[:-1])]
    bufs = [None] * len(ps)
    wd = g["group_wd"]
    for st in range(5):
        w = O.warmup_values(st, 0, 100)
        gs = np.split(g["traj_grads"][st], np.cumsum(sizes)[:-1])
        for i, p in enumerate(ps):
            lr, mom = w[O.GROUP_NAMES[grp[i]]]
            bufs[i] = O.sgd_nesterov_step(p, torch.from_numpy(gs[i].copy()), bufs[i], lr, mom, float(wd[grp[i]]))
    np.testing.assert_allclose(np.concatenate([p.numpy() for p in ps]), g["traj_p5"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("case", list(synth.mosaic_cases()))
def test_mosaic(golden, case):
    g = golden("mosaic")
    S, seed = synth.mosaic_cases()[case]
    samples = synth.source_samples(4, S, seed)
    random.seed(seed)
    img, bb, lb, border, _ = datapath.mosaic(samples, S)
    np.testing.assert_array_equal(bb, g[case + ".bboxes"])
    np.testing.assert_array_equal(lb, g[case + ".labels"])
    assert list(border) == g[case + ".border"].tolist()
    np.testing.assert_array_equal(img.astype(np.int64).sum(axis=(1, 2)), g[case + ".image_rowsum"])
    np.testing.assert_array_equal(img.astype(np.int64).sum(axis=(0, 2)), g[case + ".image_colsum"])
    if S <= 64:
        np.testing.assert_array_equal(img, g[case + ".image"])


def test_affine_boxes_flip_mixup(golden):
    g = golden("affine")
    rng = np.random.default_rng(51)
    S = 64
    for i in range(6):
        draws = datapath.affine_draws(rng)
        np.testing.assert_array_equal(np.array(draws), g["rand_values"][i])
        M, (wo, ho) = datapath.affine_matrix(draws, 2 * S, 2 * S, border=(-S // 2, -S // 2))
        assert [wo, ho] == g["feat_shape_out"].tolist()
        np.testing.assert_allclose(M, g["matrices"][i], rtol=0, atol=1e-12)
        nb, keep = datapath.affine_boxes(g["boxes_in"], M, wo, ho, draws[3])
        np.testing.assert_allclose(nb, g["boxes_out"][i], rtol=0, atol=1e-9)
        np.testing.assert_array_equal(keep, g["keep"][i])
    np.testing.assert_array_equal(datapath.flip_boxes(g["boxes_in"][:3], 6), g["flip_boxes"])
    np.random.seed(2023)
    r = np.random.beta(32.0, 32.0)
    a = torch.arange(24, dtype=torch.float32).reshape(3, 2, 4) / 24
    b = torch.arange(24, dtype=torch.float32).flip(0).reshape(3, 2, 4) / 24
    np.testing.assert_array_equal(datapath.mixup_blend(a, b, r).numpy(), g["mixup_image"])


def test_val_preprocessing_restatement_properties():
    """The OpenCV-resize / albumentations letter-box restatement (parity unpinned: neither library is in the image):
    identity at equal size, constants stay constant, agreement with float bilinear (half-pixel centres) to < 1 LSB,
    geometry of LongestMaxSize + centred PadIfNeeded."""
    import torch.nn.functional as F
    from oracle import datapath as D
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(D.resize_linear_u8(img, 53, 37), img)
    assert (D.resize_linear_u8(np.full((10, 14, 3), 77, np.uint8), 28, 20) == 77).all()
    t = torch.from_numpy(img).permute(2, 0, 1)[None].float()
    for hw in ((74, 106), (28, 40), (37, 200), (111, 53)):
        ref = F.interpolate(t, size=hw, mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        got = D.resize_linear_u8(img, hw[1], hw[0]).astype(np.float64)
        assert np.abs(got - ref).max() < 1.0, hw
    assert D.val_geometry(480, 640, 640) == (480, 640, 80, 0)
    assert D.val_geometry(333, 500, 640) == (426, 640, 107, 0)
    assert D.val_geometry(1000, 750, 640) == (640, 480, 0, 80)
    out, b = D.val_sample(img, np.array([[1.0, 2.0, 30.0, 20.0]]), 64)
    assert out.shape == (3, 64, 64) and out.dtype == np.float32
    assert out[0, 0, 0] == np.float32(114) / np.float32(255)           # padded border row
    nh, nw, top, left = D.val_geometry(37, 53, 64)
    np.testing.assert_allclose(b, [[1.0 / 53 * nw + left, 2.0 / 37 * nh + top, 30.0 / 53 * nw + left, 20.0 / 37 * nh + top]])


def _protocol_case(golden, case):
    """(mixup_prob, weights, sampler_indices, aug overrides, {field: array}) of one recorded configuration"""
    g = golden("protocol")
    mixup_prob, side, over = synth.PROTOCOL_CASES[case]
    w, si = synth.protocol_side_channel() if side else (None, None)
    return mixup_prob, w, si, over, {k[len(case) + 1:]: g[k] for k in g.files if k.startswith(case + ".")}


@pytest.mark.parametrize("case", list(synth.PROTOCOL_CASES))
def test_per_sample_protocol_vs_reference(golden, case):
    """oracle.datapath.train_sample against what the reference's REAL DetectionDataset.__getitem__ +
    TrainSampleAugmentor(rng_seed=51) did on the same pool and seeds (tests/golden/protocol.npz, recorded through the cv2
    / albumentations stand-ins of oracle/ref_import.py): per sample the indices read, the mosaic's boxes, every affine
    matrix, the HSV look-up tables, the flip outcome, the mixup ratio, the final boxes / labels - bit for bit - and the CRC
    of the final image (the composition of the stages; OpenCV's own pixel arithmetic is the oracle's restatement on
    both sides)."""
    import zlib
    mixup_prob, w, si, over, g = _protocol_case(golden, case)
    S, n, N = synth.PROTOCOL_S, synth.PROTOCOL_POOL, synth.PROTOCOL_N
    pool = synth.protocol_pool()
    aug = {}
    if "degrees" in over:
        aug.update(degrees=over["degrees"], shear=over["shear"], perspective=over.get("perspective", 0.0))
    if "flip" in over:
        aug["flip_prob"] = over["flip"]
    if "hsv" in over:
        aug["hsv"] = over["hsv"]
    if over.get("color"):                 # image_color_transforms=True: the colour stage's own generator (see datapath.color_gate)
        aug["color"] = random.Random(synth.PROTOCOL_COLOR_SEED)
    if over.get("albu13"):                # albumentations 1.3.x: every gate draws on python's global generator
        aug["color"] = aug["albu13"] = random
    random.seed(2023)
    np.random.seed(2023)
    rng = np.random.default_rng(51)
    ob = om = fired = 0
    for k in range(N):
        log = {}
        img, bb, lb = datapath.train_sample(pool, k % n, S, rng, mixup_prob=mixup_prob, weights=w, sampler_indices=si,
                                            aug=aug, log=log)
        want_idx = [int(i) for i in g["indices"][k] if i >= 0]
        assert log["indices"] == want_idx, (case, k)
        for st, stage in enumerate(log["stages"]):
            cnt = int(g["mosaic_counts"][k, st])
            assert np.array_equal(log["mosaic_boxes"][st], g["mosaic_boxes"][om:om + cnt]), (case, k, st)
            om += cnt
            assert np.array_equal(stage["M"], g["M"][k, st]), (case, k, st)
            assert tuple(stage["dsize"]) == tuple(g["dsize"][k, st])
            if stage["luts"] is None:
                assert g["n_lut"][k, st] == 0
            else:
                assert g["n_lut"][k, st] == 3 and all(np.array_equal(stage["luts"][c], g["luts"][k, st, c]) for c in range(3))
            assert int(stage["flip"]) == g["flip"][k, st]
            if over.get("color"):
                # the reference ran its albumentations stage AFTER the warp and BEFORE the first HSV look-up (color_pos = 1), and
                # the recording's draws (which transform fired, with what parameter) are the oracle's
                assert g["color_pos"][k, st] == 1
                ops, kb, km, clip = stage["color"]
                assert (ops, kb, km) == tuple(int(v) for v in g["color"][k, st]) and clip == g["color_clip"][k, st], (case, k, st)
                fired += bin(ops).count("1")
            else:
                assert stage["color"] is None
        assert (len(log["stages"]) == 2) == bool(g["flip"][k, 1] >= 0)
        if log["mixup_r"] is None:
            assert np.isnan(g["mixup_r"][k])
        else:
            assert log["mixup_r"] == g["mixup_r"][k]
        cnt = int(g["counts"][k])
        assert np.array_equal(bb, g["boxes"][ob:ob + cnt]) and np.array_equal(lb, g["labels"][ob:ob + cnt]), (case, k)
        ob += cnt
        assert img.dtype == np.float32 and zlib.crc32(np.ascontiguousarray(img).tobytes()) == g["image_crc"][k], (case, k)
    assert ob == len(g["boxes"]) and om == len(g["mosaic_boxes"])
    if over.get("albu13"):
        assert int(g["legacy_draws"].sum()) == 3 * int((g["color_pos"] == 1).sum())        # Compose + ToFloat + ToTensorV2 per call
    elif over.get("color"):
        assert fired >= 4 and set(np.unique(g["color"][..., 0])) >= {0, 1, 2, 8, 12}     # every transform fired in the recording


@pytest.mark.parametrize("case", list(synth.sppf_cases()))
def test_sppf_forms_vs_reference(golden, case):
    """oracle.network.SPPF in the reference's three forms (kod/nn/layers/sppf.py:27-83: one kernel size = the cascade,
    a kernel-size sequence = parallel pools, use_conv_first=False) against the reference module's own output and
    gradients under the same seed (tests/golden/sppf.npz)."""
    from oracle.network import SPPF
    g = golden("sppf")
    cin, cout, ks, first, B, H, W, seed = synth.sppf_cases()[case]
    torch.manual_seed(seed)
    m = SPPF(cin, cout, ks, first).train()
    p = case + "."
    assert list(m.state_dict().keys()) == [str(k) for k in g[p + "keys"]]
    for k, v in m.named_parameters():
        assert np.array_equal(v.detach().numpy(), g[p + "param." + k]), k
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=gen).requires_grad_(True)
    y = m(x)
    w = torch.randn(y.shape, generator=gen)
    (y * w).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), g[p + "y"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), g[p + "dx"], rtol=1e-5, atol=1e-6)
    for k, v in m.named_parameters():
        np.testing.assert_allclose(v.grad.numpy(), g[p + "grad." + k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_fixtures_regenerate_from_the_reference(tmp_path):
    """The pinning recipe is ONE command: `python -m oracle.gen_golden` from a clean process with an empty output
    directory exits 0 and reproduces every committed fixture array for array (the generators used to
    depend on their order - the reference binds albumentations' ToTensorV2 at import).  Needs the reference checkout
    (build container only; skipped on the GPU box)."""
    import os
    import subprocess
    import sys
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("reference checkout not present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {**os.environ, "KOD_GOLDEN_OUT": str(tmp_path), "PYTHONDONTWRITEBYTECODE": "1"}
    r = subprocess.run([sys.executable, "-m", "oracle.gen_golden"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    assert made == ["affine.npz", "assigner.npz", "decode_nms.npz", "iou.npz", "loss.npz", "mosaic.npz", "network.npz",
                    "optim.npz", "protocol.npz", "samplers.npz", "sppf.npz"], made
    n = 0
    for f in made:
        new, old = np.load(tmp_path / f, allow_pickle=False), np.load(os.path.join(root, "tests", "golden", f), allow_pickle=False)
        assert sorted(new.files) == sorted(old.files), f
        for k in new.files:
            a, b = new[k], old[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (f, k)
            if a.dtype.kind in "fc":
                assert np.array_equal(a, b, equal_nan=True), (f, k)
            else:
                assert np.array_equal(a, b), (f, k)
            n += 1
    assert n > 500


def test_color_transform_restatements_against_independent_implementations():
    """The colour stage's pixel restatements (oracle/datapath.py; albumentations / OpenCV are absent) against what IS in this
    image: the median filter against Pillow's rank filter (a median is a median - the same replicate border, every kernel
    size, bit for bit), ToGray within one level of Pillow's ITU-R 601 luma (different rounding), the box blur against its
    definition in float64, CLAHE's properties (identity on flat images up to the Lab round trip, monotone look-up tables,
    reflect-padded tiles for sizes that are no multiple of 8) and the 8-bit Lab round trip (grey ramp within one level,
    OpenCV's documented values for the primaries)."""
    pytest.importorskip("PIL")
    from PIL import Image, ImageFilter
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for k in (3, 5, 7):
        pil = np.asarray(Image.fromarray(img).filter(ImageFilter.MedianFilter(k)))
        np.testing.assert_array_equal(datapath.median_blur_u8(img, k), pil)
        pad = np.pad(img.astype(np.float64), ((k // 2, k // 2), (k // 2, k // 2), (0, 0)), mode="reflect")
        box = sum(pad[dy:dy + 37, dx:dx + 53] for dy in range(k) for dx in range(k)) / (k * k)
        assert np.abs(datapath.blur_u8(img, k).astype(np.float64) - box).max() <= 0.5 + 1e-9
    luma = np.asarray(Image.fromarray(img).convert("L")).astype(int)
    assert np.abs(datapath.to_gray_u8(img)[..., 0].astype(int) - luma).max() <= 1
    T = datapath.lab_tables()
    prim = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 0, 0]]], np.uint8)
    np.testing.assert_array_equal(datapath.rgb2lab_u8(prim, T)[0], [[255, 128, 128], [136, 208, 195], [224, 42, 211], [82, 207, 20], [0, 128, 128]])
    ramp = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, -1)
    assert np.abs(datapath.lab2rgb_u8(datapath.rgb2lab_u8(ramp, T), T).astype(int) - ramp.astype(int)).max() <= 1
    flat = np.full((64, 64, 3), 97, np.uint8)
    assert np.abs(datapath.clahe_u8(flat, 3.0).astype(int) - 97).max() <= 160      # a flat tile's whole histogram is one bin: its LUT jumps
    smooth = np.clip(np.add.outer(np.arange(60), np.arange(52))[..., None] * 2 + np.array([0, 10, 20]), 0, 255).astype(np.uint8)
    out = datapath.clahe_u8(smooth, 2.0)
    assert out.shape == smooth.shape and out.dtype == np.uint8
    L = datapath.rgb2lab_u8(smooth, T)[..., 0]
    eq = datapath.clahe_plane_u8(L, 2.0)
    assert int(eq.max()) - int(eq.min()) >= int(L.max()) - int(L.min())            # equalisation stretches, never collapses, the range
    # the gate: one Compose draw + four transform draws per call, parameters only for what fired
    import random as _r

    class Count(_r.Random):
        n = 0

        def random(self):
            Count.n += 1
            return super().random()
    g = Count(5)
    fired = [datapath.color_gate(g) for _ in range(400)]
    assert Count.n >= 5 * 400 and sum(1 for f in fired if f[0]) in range(4, 40)
    assert all((f[1] in (3, 5, 7)) == bool(f[0] & 1) and (f[2] in (3, 5, 7)) == bool(f[0] & 2) and ((1.0 <= f[3] <= 4.0) == bool(f[0] & 8)) for f in fired)
