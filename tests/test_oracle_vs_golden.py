"""CPU: the oracle (oracle/cpu_ref.py) against fixtures produced by the reference itself
(tests/golden/make_fixtures.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref


def _check_summary(t, fx, key, rtol=2e-5, atol=1e-6):
    flat = t.detach().reshape(-1)
    assert list(t.shape) == list(fx[f"{key}.shape"])
    idx = torch.from_numpy(fx[f"{key}.idx"])
    np.testing.assert_allclose(flat[idx].numpy(), fx[f"{key}.val"], rtol=rtol, atol=atol)
    d = flat.double()
    assert float(d.sum()) == pytest.approx(float(fx[f"{key}.sum"]), rel=1e-4, abs=1e-3)
    assert float(d.abs().sum()) == pytest.approx(float(fx[f"{key}.abs_sum"]), rel=1e-5)
    assert float((d * d).sum()) == pytest.approx(float(fx[f"{key}.sq_sum"]), rel=1e-5)


def test_known_answers(golden):
    fx = golden("kat")
    g = cpu_ref.gram_matrix(torch.from_numpy(fx["gram_in"]))
    np.testing.assert_array_equal(g.numpy(), fx["gram"])
    np.testing.assert_allclose(g.numpy()[0], [[21.083334, 54.083332], [54.083332, 159.08333]], rtol=1e-7)
    gu = cpu_ref.gram_matrix(torch.from_numpy(fx["gram_in"]), should_normalize=False)
    np.testing.assert_array_equal(gu.numpy()[0], [[506, 1298], [1298, 3818]])
    tv = cpu_ref.total_variation(torch.from_numpy(fx["tv_in"]))
    assert float(tv) == float(fx["tv"])
    assert float(tv) == pytest.approx(7.0077858, rel=1e-7)
    p = cpu_ref.prepare_img(fx["prep_in"])
    np.testing.assert_array_equal(p.numpy(), fx["prep"])
    np.testing.assert_allclose(p.numpy()[0, :, 0, 0], [-123.675, -112.688446, -96.3469], rtol=1e-6)
    np.testing.assert_array_equal(cpu_ref.unprepare_img(p), fx["unprep"])
    np.testing.assert_array_equal(cpu_ref.gram_matrix(torch.from_numpy(fx["gram_rand_in"])).numpy(), fx["gram_rand"])


@pytest.mark.parametrize("tag", ["even", "odd", "tiny"])
def test_bicubic_half(golden, tag):
    fx = golden("bicubic")
    x = torch.from_numpy(fx[f"{tag}_x"]).requires_grad_(True)
    y = cpu_ref.bicubic_half(x)
    np.testing.assert_array_equal(y.detach().numpy(), fx[f"{tag}_y"])
    (y * torch.from_numpy(fx[f"{tag}_gy"])).sum().backward()
    np.testing.assert_array_equal(x.grad.numpy(), fx[f"{tag}_gx"])
    if tag != "odd":   # exact 1/2: the fixed 4-tap filter
        np.testing.assert_allclose(cpu_ref.bicubic_half_fixed(x.detach()).numpy(), fx[f"{tag}_y"], rtol=1e-5, atol=1e-6)


def test_vgg_features(golden, vgg_weights):
    fx = golden("vgg_48x80")
    assert int(fx["content_index"]) == cpu_ref.CONTENT_INDEX
    assert tuple(fx["style_indices"]) == cpu_ref.STYLE_INDICES
    assert list(fx["layer_names"]) == ["relu1_1", "relu2_1", "relu3_1", "relu4_1", "conv4_2", "relu5_1"]
    np.testing.assert_array_equal(cpu_ref.synthetic_image(48, 80, seed=3), fx["img"])
    x = cpu_ref.prepare_img(fx["img"]).requires_grad_(True)
    outs = cpu_ref.vgg19_features(x, vgg_weights)
    g = torch.Generator().manual_seed(int(fx["grad_seed"]))
    loss = 0
    for i, o in enumerate(outs):
        _check_summary(o, fx, f"out{i}")
        _check_summary(cpu_ref.gram_matrix(o), fx, f"gram{i}")
        loss = loss + (o * (torch.randn(o.shape, generator=g) / o.numel())).sum()
    assert float(outs[4].min()) == 0.0          # "conv4_2" is ReLU(conv4_2)
    np.testing.assert_allclose(outs[5].detach().numpy(), fx["out5_full"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(outs[4].detach().numpy(), fx["out4_full"], rtol=1e-5, atol=1e-5)
    loss.backward()
    np.testing.assert_allclose(x.grad.numpy(), fx["grad"], rtol=1e-4, atol=1e-7)


def _closure_from_fixture(fx, vgg_weights, nlev):
    tg = [cpu_ref.LevelTargets(cpu_ref.prepare_img(fx[f"content{i}"]), cpu_ref.prepare_img(fx[f"style{i}"]), vgg_weights)
          for i in range(nlev)]
    return cpu_ref.closure_eval(cpu_ref.prepare_img(fx["x_img"]), tg, vgg_weights, 1e3, 4e5, 1e2)


@pytest.mark.parametrize("name,nlev", [("closure_64x96_L1", 2), ("closure_50x76_L0", 1)])
def test_closure_small(golden, vgg_weights, name, nlev):
    fx = golden(name)
    loss, grad, rows = _closure_from_fixture(fx, vgg_weights, nlev)
    assert float(loss) == pytest.approx(float(fx["total"]), rel=1e-6)
    np.testing.assert_allclose(np.array(rows), fx["rows"], rtol=1e-6)
    ref = fx["grad"]
    rel_l2 = np.linalg.norm(grad.numpy() - ref) / np.linalg.norm(ref)
    assert rel_l2 < 1e-6
    # every loss term alone: (cw,0,0), (0,sw,0), (0,0,tvw) through the reference's LossBuilder
    tg = [cpu_ref.LevelTargets(cpu_ref.prepare_img(fx[f"content{i}"]), cpu_ref.prepare_img(fx[f"style{i}"]), vgg_weights)
          for i in range(nlev)]
    for tag, wts in (("c", (1e3, 0.0, 0.0)), ("s", (0.0, 4e5, 0.0)), ("tv", (0.0, 0.0, 1e2))):
        loss, grad, _ = cpu_ref.closure_eval(cpu_ref.prepare_img(fx["x_img"]), tg, vgg_weights, *wts)
        assert float(loss) == pytest.approx(float(fx[f"total_{tag}"]), rel=1e-6)
        ref = fx[f"grad_{tag}"]
        assert np.linalg.norm(grad.numpy() - ref) / np.linalg.norm(ref) < 1e-6, tag


def test_forced_decisions_are_a_no_op_on_own_decisions(vgg_weights):
    """cpu_ref.Decisions taken from the evaluation's own activations must reproduce that evaluation bit for bit (the
    GPU tests hand the DEVICE pass's decisions to the oracle; this pins the mechanism itself)."""
    c, s = _levels(48, 80, 2, 1), _levels(40, 56, 2, 2)
    tg = [cpu_ref.LevelTargets(cpu_ref.prepare_img(a), cpu_ref.prepare_img(b), vgg_weights) for a, b in zip(c, s)]
    xt = cpu_ref.prepare_img((0.5 * c[0] + 0.5 * cpu_ref.synthetic_image(48, 80, 9)).astype(np.float32))
    rec = []
    loss, grad, rows = cpu_ref.closure_eval(xt, tg, vgg_weights, 1e3, 4e5, 1e2, record=rec)
    assert len(rec) == 2 and len(rec[0]) == 13
    dec = [cpu_ref.Decisions([torch.relu(p) for p in r]) for r in rec]
    loss2, grad2, rows2 = cpu_ref.closure_eval(xt, tg, vgg_weights, 1e3, 4e5, 1e2, decisions=dec)
    assert float(loss2) == float(loss) and rows2 == rows
    assert torch.equal(grad2, grad)
    # flipping ONE deep unit changes the gradient over its receptive field - the effect the forced mode removes
    dec[0].relu[9][0, 5, 2, 3] = ~dec[0].relu[9][0, 5, 2, 3]
    _, grad3, _ = cpu_ref.closure_eval(xt, tg, vgg_weights, 1e3, 4e5, 1e2, decisions=dec)
    assert not torch.equal(grad3, grad)


def _levels(h, w, nlev, seed):
    import torch.nn.functional as F
    top = cpu_ref.synthetic_image(h, w, seed)
    out = [top]
    t = torch.from_numpy(top).permute(2, 0, 1).unsqueeze(0)
    for l in range(1, nlev):
        d = F.interpolate(t, size=(h >> l, w >> l), mode="bicubic", align_corners=False)
        out.append(d.squeeze(0).permute(1, 2, 0).contiguous().numpy())
    return out


def test_closure_L0_256x384(golden, vgg_weights):
    fx = golden("closure_256x384_L0")
    c, s = _levels(256, 384, 1, 1), _levels(256, 384, 1, 2)
    assert float(c[0].astype(np.float64).sum()) == float(fx["content_sum"])
    assert float(s[0].astype(np.float64).sum()) == float(fx["style_sum"])
    tg = [cpu_ref.LevelTargets(cpu_ref.prepare_img(c[0]), cpu_ref.prepare_img(s[0]), vgg_weights)]
    loss, grad, rows = cpu_ref.closure_eval(cpu_ref.prepare_img(c[0]), tg, vgg_weights, 1e3, 4e5, 1e2)
    assert float(loss) == pytest.approx(float(fx["total"]), rel=1e-6)
    np.testing.assert_allclose(np.array(rows), fx["rows"], rtol=1e-6)
    _check_summary(grad, fx, "grad", rtol=1e-4, atol=1e-7)


def test_adam_trajectory_small(golden, vgg_weights):
    fx = golden("traj_adam_64x96_L1_12")
    c, s = _levels(64, 96, 2, 1), _levels(64, 96, 2, 2)
    rec, imgs = [], []
    for img, step in cpu_ref.run_process(c, s, c[0], vgg_weights, "adam", 12, record=rec):
        imgs.append((img, step))
    assert [st for _, st in imgs] == list(fx["steps"])
    rows = np.array([r["rows"] for r in rec])
    np.testing.assert_allclose(rows, fx["rows"], rtol=2e-5)
    np.testing.assert_allclose(imgs[0][0], fx["after_1"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(imgs[-1][0], fx["final"], rtol=0, atol=2e-4)


@pytest.mark.parametrize("tag,max_eval", [("shipped", 1), ("legacy", 26)])
def test_lbfgs_trajectory_small(golden, vgg_weights, tag, max_eval):
    fx = golden(f"traj_lbfgs_128x192_L1_{tag}")
    c, s = _levels(128, 192, 2, 1), _levels(128, 192, 2, 2)
    rec, imgs = [], []
    prev = c[0]
    moved = []
    for img, step in cpu_ref.run_process(c, s, c[0], vgg_weights, "lbfgs", 40, lbfgs_max_eval=max_eval, record=rec):
        imgs.append((img, step))
        moved.append(bool(np.any(img != prev)))
        prev = img
    assert [st for _, st in imgs] == list(fx["steps"])
    assert moved == list(fx["moved"])
    rows = np.array([r["rows"] for r in rec])
    assert rows.shape == fx["rows"].shape
    np.testing.assert_allclose(rows, fx["rows"], rtol=1e-4)
    _check_summary(torch.from_numpy(imgs[-1][0]), fx, "final", rtol=0, atol=1e-4)


@pytest.mark.slow
def test_adam_trajectory_L0_config1(golden, vgg_weights):
    """BASELINE config 1 on the CPU path: 384x256, 50 Adam iterations."""
    fx = golden("traj_adam_256x384_50")
    c, s = _levels(256, 384, 1, 1), _levels(256, 384, 1, 2)
    rec, last = [], None
    for img, step in cpu_ref.run_process(c, s, c[0], vgg_weights, "adam", 50, record=rec):
        if step == 1:
            _check_summary(torch.from_numpy(img), fx, "img_after_1", rtol=0, atol=1e-5)
        last = img
    rows = np.array([r["rows"] for r in rec])
    np.testing.assert_allclose(rows, fx["rows"], rtol=1e-4)
    _check_summary(torch.from_numpy(last), fx, "final", rtol=0, atol=5e-4)


def test_config2_geometry_prefixes(golden, vgg_weights):
    """The fixtures of BASELINE config 2's pyramid (768x512 + 384x256) hold hundreds of closures of the reference (25 min
    of CPU to make); the oracle is held to their first closures here (the whole runs are the GPU tests' reference):
    Adam 3 iterations, L-BFGS as shipped 4 closures."""
    c, s = _levels(512, 768, 2, 1), _levels(512, 768, 2, 2)
    fx = golden("traj_adam_512x768_L1_100")
    rec = []
    for img, step in cpu_ref.run_process(c, s, c[0], vgg_weights, "adam", 3, record=rec):
        if step == 1:
            _check_summary(torch.from_numpy(img), fx, "after_1", rtol=0, atol=1e-5)
    np.testing.assert_allclose(np.array([r["rows"] for r in rec]), fx["rows"][:3], rtol=2e-5)
    fx = golden("traj_lbfgs_512x768_L1_500")
    rec, moved, prev = [], [], c[0]
    for img, step in cpu_ref.run_process(c, s, c[0], vgg_weights, "lbfgs", 4, record=rec):
        moved.append(bool(np.any(img != prev)))
        prev = img
    assert moved == list(fx["moved"][:2]) and list(fx["steps"][:2]) == [2, 4]
    np.testing.assert_allclose(np.array([r["rows"] for r in rec]), fx["rows"][:4], rtol=1e-4)


def test_config3_workload_first_closure(golden, vgg_weights):
    """The oracle at the headline workload's size (L=2: 1536x1024 + 768x512 + 384x256): the loss rows of the first closure
    of the reference's Adam run (tests/golden/traj_adam_1024x1536_L2_16.npz; the 16-iteration runs are the GPU tests')."""
    c, s = _levels(1024, 1536, 3, 1), _levels(1024, 1536, 3, 2)
    init = (0.7 * c[0] + 0.3 * cpu_ref.synthetic_image(1024, 1536, seed=3)).astype(np.float32)
    fx = golden("traj_adam_1024x1536_L2_16")
    rec = []
    for img, step in cpu_ref.run_process(c, s, init, vgg_weights, "adam", 1, record=rec):
        _check_summary(torch.from_numpy(img), fx, "after_1", rtol=0, atol=1e-5)
    np.testing.assert_allclose(np.array([r["rows"] for r in rec]), fx["rows"][:1], rtol=2e-5)
