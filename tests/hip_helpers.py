"""Shared helpers of the GPU parity tests (tests/test_hip_*.py)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import cpu_ref

# Whole-gradient cap of a teacher-forced closure.  Arithmetic agrees to ~1e-6; what sets this bound are ReLU /
# max-pool DECISIONS: a pre-activation within one ulp of 0 lands on different sides in two fp32 evaluations that sum in
# different orders, and one flipped unit at conv4/conv5 depth moves the pixel gradient over its whole receptive field
# (measured: one flipped unit of 49152 at ReLU(conv4_2) = 2e-3 of the content gradient at 64x96; the torch-fp32 oracle
# itself sits 2e-4 ... 2.4e-3 from an fp64 evaluation of the same closure).  The cap alone would also pass a wrong or
# missing small loss term, so every closure test goes through assert_grad_close below, per loss term.
GRAD_RTOL = 3e-3
# outside the receptive fields of flipped units the gradient must agree like any other fp32 quantity
BULK_RTOL = 2e-5

CW, SW, TVW = 1e3, 4e5, 1e2
TERMS = (("all", (CW, SW, TVW)), ("content", (CW, 0.0, 0.0)), ("style", (0.0, SW, 0.0)), ("tv", (0.0, 0.0, TVW)))

_REPORT = os.environ.get("NST_TEST_REPORT")


def report(line: str) -> None:
    """Measurements the tolerances were chosen from: appended to $NST_TEST_REPORT when set (gpurun_out/...)."""
    print(line)
    if _REPORT:
        with open(_REPORT, "a") as f:
            f.write(line + "\n")


def dev(t):
    return t.contiguous().to("cuda:0")


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def check_rows(rows, ref_rows, rtol, cw=CW, sw=SW, tvw=TVW):
    """Loss rows (total, content, style, tv): totals relatively; each component by its weighted
    contribution to the level total (a content loss of 1e-10 is rounding noise, not a quantity)."""
    rows = np.asarray(rows, dtype=np.float64)
    ref = np.asarray(ref_rows, dtype=np.float64)
    assert rows.shape == ref.shape
    np.testing.assert_allclose(rows[..., 0], ref[..., 0], rtol=rtol)
    for j, wgt in ((1, cw), (2, sw), (3, tvw)):
        err = np.abs(rows[..., j] - ref[..., j]) * wgt
        assert np.all(err <= rtol * np.abs(ref[..., 0])), (j, float(err.max()))


def rows_rel_err(rows, ref_rows):
    """Largest relative error of the level totals (what check_rows' rtol is compared with)."""
    rows = np.asarray(rows, dtype=np.float64)
    ref = np.asarray(ref_rows, dtype=np.float64)
    return float(np.max(np.abs(rows[..., 0] - ref[..., 0]) / np.abs(ref[..., 0])))


def levels(h, w, nlev, seed):
    """Pyramid of synthetic images, highest-res first (same construction as tests/golden/make_fixtures.py)."""
    top = cpu_ref.synthetic_image(h, w, seed)
    out = [top]
    t = torch.from_numpy(top).permute(2, 0, 1).unsqueeze(0)
    for l in range(1, nlev):
        d = F.interpolate(t, size=(h >> l, w >> l), mode="bicubic", align_corners=False)
        out.append(d.squeeze(0).permute(1, 2, 0).contiguous().numpy())
    return out


def setup(eng, contents, styles):
    nlev = len(contents)
    h, w = contents[0].shape[:2]
    eng.configure(nlev, h, w)
    for i in range(nlev):
        eng.set_targets(i, dev(cpu_ref.prepare_img(contents[i])), dev(cpu_ref.prepare_img(styles[i])))


def oracle_targets(contents, styles, weights):
    return [cpu_ref.LevelTargets(cpu_ref.prepare_img(c), cpu_ref.prepare_img(s), weights) for c, s in zip(contents, styles)]


def grad_stats(got, ref):
    """(rel-L2 of the whole, share of entries inside flipped receptive fields, rel-L2 of the rest)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    full = rel_l2(got, ref)
    err = np.abs(got - ref)
    bad = err > 1e-4 * np.abs(ref).max()
    bulk = float(np.linalg.norm((got - ref)[~bad]) / max(np.linalg.norm(ref[~bad]), 1e-30))
    return full, float(bad.mean()), bulk


def assert_grad_close(got, ref, what="", cap=GRAD_RTOL, max_flipped=0.02, bulk=BULK_RTOL):
    """Flip-aware gradient comparison.  Either the whole gradient agrees to `bulk` (2e-5 rel-L2), or: the entries
    that differ by more than 1e-4 of the largest gradient - the receptive fields of ReLU / pooling decisions that fell
    on the other side - are few (`max_flipped` of the pixels), everything outside them agrees to `bulk`, and the whole
    stays under `cap`."""
    full, flipped, rest = grad_stats(got, ref)
    report(f"grad {what}: rel-L2 {full:.2e}, flipped-field entries {flipped:.3%}, rest rel-L2 {rest:.2e}")
    if full < bulk:
        return
    assert flipped < max_flipped and rest < bulk and full < cap, (what, full, flipped, rest)


def check_summary(t, fx, key, atol, rtol=0.0):
    """A tensor against a fixture summary written by make_fixtures.summarize (sampled values + moments)."""
    flat = torch.as_tensor(t).detach().reshape(-1).cpu()
    assert list(torch.as_tensor(t).shape) == list(fx[f"{key}.shape"])
    idx = torch.from_numpy(fx[f"{key}.idx"])
    np.testing.assert_allclose(flat[idx].numpy(), fx[f"{key}.val"], rtol=rtol, atol=atol)
    return float(np.max(np.abs(flat[idx].numpy() - fx[f"{key}.val"])))


# ---- comparison under equal ReLU / pooling decisions -------------------------------------------------------------
def device_decisions(eng, x=None):
    """cpu_ref.Decisions of the closure the engine evaluated last, one per pyramid level: ReLU / pooling decisions from
    the level's activations and, when the level-0 image x is given, the signs its total-variation term took (levels >= 1:
    from the device's own down-sampled image, nst_level_image)."""
    out = []
    for l in range(eng.levels):
        img = None
        if x is not None:
            img = (x if l == 0 else eng.level_image(l)).cpu().reshape(1, 3, *eng.level_shape(l))
        out.append(cpu_ref.Decisions([a.cpu() for a in eng.level_activations(l)], img))
    return out


def tv_sign_disagreements(level_imgs, dec):
    """(share of neighbour pairs whose difference has another sign in this (oracle) evaluation than in `dec`, largest
    |difference| at such a pair - in the prepared image's units, where one ulp of a value near 128 is 1.5e-5)."""
    pairs = flips = 0
    worst = 0.0
    for y, d in zip(level_imgs, dec):
        for diff, s in ((y[:, :, :, :-1] - y[:, :, :, 1:], d.tv[0]), (y[:, :, :-1, :] - y[:, :, 1:, :], d.tv[1])):
            bad = torch.sign(diff) != s
            pairs += bad.numel()
            n = int(bad.sum())
            flips += n
            if n:
                worst = max(worst, float(diff[bad].abs().max()))
    return flips / max(pairs, 1), worst


def decision_disagreements(pre, dec):
    """Where this (oracle) evaluation's own decisions differ from `dec` (the device pass's): (share of ReLU units,
    largest |pre-activation| at such a unit relative to the layer's rms, share of pooling windows with a positive
    maximum that chose another position, largest gap between the two candidates relative to the layer's rms)."""
    units = flips = 0
    worst = 0.0
    windows = moved = 0
    worst_gap = 0.0
    for li, ((name, _, _), p) in enumerate(zip(cpu_ref.VGG19_CONVS, pre)):
        rms = float(p.double().pow(2).mean().sqrt())
        diff = (p > 0) != dec.relu[li]
        units += diff.numel()
        n = int(diff.sum())
        flips += n
        if n:
            worst = max(worst, float(p[diff].abs().max()) / rms)
        if name in cpu_ref.POOL_AFTER:
            a = torch.relu(p)
            own_val, own_idx = torch.nn.functional.max_pool2d(a, 2, 2, return_indices=True)
            idx = dec.pool[name]
            other_val = a.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
            d = (own_idx != idx) & (own_val > 0)
            windows += d.numel()
            m = int(d.sum())
            moved += m
            if m:
                worst_gap = max(worst_gap, float((own_val - other_val)[d].max()) / rms)
    return flips / max(units, 1), worst, moved / max(windows, 1), worst_gap


# a decision may differ between two fp32 evaluations only where the quantity it tests is within accumulated rounding of
# the decision point: |pre-activation| (or the gap between two pooling candidates) below this share of the layer's rms
NEAR_TIE = 2e-5          # measured: <= 4e-6 (bf16x3), <= 1.1e-6 otherwise
# neighbour differences whose sign may differ: a few ulps of a prepared pixel value (|v| <= 152: one ulp = 1.5e-5)
TV_NEAR_TIE = 2e-4


def closure_vs_oracle_under_equal_decisions(eng, xt, tg, weights, what, terms=TERMS, grad_tol=BULK_RTOL, loss_tol=1e-5,
                                            cap=GRAD_RTOL, own_cache=None):
    """The strict form of the closure parity test.  (`own_cache`: see below.)  For the weighted sum and for every loss term alone:
    (1) losses against the oracle's own evaluation (rel <= 1e-5);
    (2) the device pass's ReLU / pooling decisions differ from the oracle's own only at near-ties (NEAR_TIE), in a
        small share of the units; likewise the signs its total-variation term takes (differences of neighbouring pixels
        that are rounding noise of the down-sampling on flat image regions: |difference| <= TV_NEAR_TIE);
    (3) under the DEVICE's decisions (cpu_ref.Decisions) the oracle's gradient must be the device's: rel-L2 <=
        2e-5 over the WHOLE gradient, no entry excluded (measured 3e-7 ... 3e-6);
    (4) against the oracle's own decisions the whole stays under `cap` = GRAD_RTOL (what the flipped near-ties cost);
    (5) on the device the terms add up: g(all) = g(content) + g(style) + g(tv) to fp32 summation error - every term
        is in the sum with its weight (the decisions of the four device passes are the same: one forward)."""
    nlev = eng.levels
    xd = dev(xt)
    parts = {}
    for name, (cw, sw, tvw) in terms:
        grad, losses = eng.closure(xd, cw, sw, tvw)
        dec = device_decisions(eng, xd)
        losses = losses.cpu().numpy()
        g = grad.cpu().numpy()
        parts[name] = g.astype(np.float64)
        # the oracle's evaluation under its OWN decisions does not depend on the engine under test: tests that hold several
        # engines / schedules to the same job pass a dict and it is computed once per loss term
        if own_cache is not None and name in own_cache:
            loss, grad_own, rows, rec = own_cache[name]
        else:
            rec = []
            loss, grad_own, rows = cpu_ref.closure_eval(xt, tg, weights, cw, sw, tvw, record=rec)
            if own_cache is not None:
                own_cache[name] = (loss, grad_own, rows, rec)
        assert float(losses[-1]) == pytest_approx(float(loss), loss_tol), (what, name, float(losses[-1]), float(loss))
        check_rows(losses[:-1].reshape(nlev, 4), np.array(rows), 2 * loss_tol, cw, sw, tvw)
        if name == "tv":                                                    # no network: only the sign decisions
            lv = [xt]
            for l in range(1, nlev):
                lv.append(cpu_ref.bicubic_half(lv[-1]))
            tshare, ttie = tv_sign_disagreements(lv, dec)
            _, grad_forced, _ = cpu_ref.closure_eval(xt, tg, weights, cw, sw, tvw, decisions=dec)
            e_forced, e_own = rel_l2(g, grad_forced.numpy()), rel_l2(g, grad_own.numpy())
            report(f"closure {what} [tv]: gradient rel-L2 under equal signs {e_forced:.2e}, under the oracle's own {e_own:.2e}; "
                   f"signs differ at {tshare:.2e} of the neighbour pairs (largest |difference| there {ttie:.1e})")
            assert ttie < TV_NEAR_TIE and tshare < 0.3, (what, tshare, ttie)      # flat (clipped) regions can be a large share
            assert e_forced < 5e-6, (what, e_forced)
            continue
        stats = [decision_disagreements(rec[l], dec[l]) for l in range(nlev)]
        share = max(s[0] for s in stats); tie = max(s[1] for s in stats)
        pshare = max(s[2] for s in stats); ptie = max(s[3] for s in stats)
        loss_f, grad_forced, _ = cpu_ref.closure_eval(xt, tg, weights, cw, sw, tvw, decisions=dec)
        e_forced, e_own = rel_l2(g, grad_forced.numpy()), rel_l2(g, grad_own.numpy())
        report(f"closure {what} [{name}]: gradient rel-L2 under equal decisions {e_forced:.2e}, under the oracle's own {e_own:.2e}; "
               f"decisions differ at {share:.2e} of the ReLU units (largest |pre|/rms there {tie:.1e}) and {pshare:.2e} of the "
               f"pooling windows (gap/rms {ptie:.1e})")
        assert share < 1e-3 and tie < NEAR_TIE and pshare < 1e-3 and ptie < NEAR_TIE, (what, name, stats)
        assert abs(float(loss_f) - float(loss)) <= 1e-5 * abs(float(loss))   # forcing near-ties does not move the loss
        assert e_forced < grad_tol, (what, name, e_forced)
        assert e_own < cap, (what, name, e_own)
    if all(k in parts for k in ("all", "content", "style", "tv")):
        s = parts["content"] + parts["style"] + parts["tv"]
        add = float(np.linalg.norm(parts["all"] - s) / np.linalg.norm(s))
        report(f"closure {what}: |g(all) - sum of the terms| / |.| = {add:.1e}")
        assert add < 2e-6, (what, add)


def pytest_approx(value, rel):
    import pytest
    return pytest.approx(value, rel=rel)


# ---- the shipped L-BFGS against a run of the reference, its first two steps taken apart ---------------------------------
def totals(rows):
    return np.asarray(rows)[:, :, 0].sum(axis=1)


def lbfgs_run(e, x, closures, max_eval=1, nlev=3):
    from artstyletransfer_amd.engine import PixelOptimizer
    opt = PixelOptimizer(e, "lbfgs", 10.0, max_eval)
    rows, steps, moved, xs = [], [], [], [x.clone()]
    total = 0
    while total < closures:
        info, r = opt.step(x, CW, SW, TVW)
        total = info.total_closures
        steps.append(total)
        moved.append(bool(info.accepted))
        rows.extend(list(r[:, :-1].reshape(-1, nlev, 4)))
        xs.append(x.clone())
    opt.close()
    return np.array(rows), steps, moved, xs


def first_step(g0, lr=10.0):
    """s of L-BFGS' first step: t d with d = -g0 and t = min(1, 1/|g0|_1) lr (torch:optim/lbfgs.py:414, :454-457).  NOT
    x1 - x0: the step is a few ulps of the pixel values, so the difference of the two images is quantisation noise."""
    t = min(1.0, 1.0 / float(g0.double().abs().sum())) * lr
    return g0 * (-t)


def _fp64_direction(g1, y, s):
    """torch:optim/lbfgs.py:396-442 with ONE curvature pair, in double."""
    g1, y, s = g1.double(), y.double(), s.double()
    ys = float(y.dot(s))
    q = -g1
    al = float(s.dot(q)) / ys
    q = q - al * y
    r = q * (ys / float(y.dot(y)))
    be = float(y.dot(r)) / ys
    return r + (al - be) * s, ys


def lbfgs_vs_reference_taken_apart(eng, vgg_weights, c_lv, s_lv, x_init, fx, closures, what, oracle_optimiser=True):
    """The shipped L-BFGS (max_eval 1) on a three-level job against a run of the reference (fixture `fx`: rows, steps, moved,
    after_1): identical closure counts and accept / reject sequence, and the first two steps taken apart - see the numbered
    blocks.  x_init: the prepared (1,3,H,W) start image (CPU tensor).  oracle_optimiser=False leaves block (3) out (four
    oracle closures at full size: a minute of host time) - for a job whose second step every run rejects, where (3) has only
    a decision to compare."""
    from artstyletransfer_amd.engine import StyleEngine
    NLEV = len(c_lv)
    H0, W0 = x_init.shape[2], x_init.shape[3]
    setup(eng, c_lv, s_lv)
    rows, steps, moved, xs = lbfgs_run(eng, dev(x_init), closures, nlev=NLEV)
    assert steps == list(fx["steps"])                      # closures per optimizer.step
    assert moved == list(fx["moved"])                      # accept / reject sequence
    tot, ref = totals(rows), totals(fx["rows"])
    err = np.abs(tot - ref) / ref
    acc = [i for i, m in enumerate(moved) if m]
    report(f"{what}: accepted steps {acc} of {len(moved)} "
           f"(reference {[i for i, m in enumerate(fx['moved']) if m]}); total-loss rel err per closure " + np.array2string(err, precision=1))
    # closures 0-2: x0, the first trial point (t = min(1, 1/|g|_1) lr: a tiny step) and its re-evaluation
    check_rows(rows[:3], fx["rows"][:3], 2e-5)
    final_err = abs(tot[-2] - ref[-2]) / ref[-2] if not moved[-1] else err[-1]
    assert moved[0], "the first step (a 1/|g|_1-long step down the gradient) is accepted in the reference's run"

    # ---- (1) teacher-forced: the oracle AT the device's own iterates (after step 1, after step 2), under the device's
    # decisions: losses 1e-5, the whole gradient 2e-5 - the closure is right at the points the landing depends on
    tg = [cpu_ref.LevelTargets(cpu_ref.prepare_img(c), cpu_ref.prepare_img(s), vgg_weights) for c, s in zip(c_lv, s_lv)]
    grads, decs = [], []
    lr2 = 10.0 * 0.999 ** 2                                # the group's lr read before step 2's first closure (two decays so far)
    for k in ((0, 1, 2) if (oracle_optimiser or moved[1]) else (0, 1)):
        if k == 2 and not moved[1]:
            # step 2 was rejected: the trial point it evaluated is rebuilt from the device's own direction
            s0, y0 = first_step(grads[0]), grads[1] - grads[0]
            ys0 = float(y0.double().dot(s0.double()))
            xs[2] = xs[1] + lr2 * eng.lbfgs_direction(grads[1], [y0], [s0], [1.0 / ys0], ys0 / float(y0.double().dot(y0.double())), 0).view_as(xs[1])
        g, l = eng.closure(xs[k], CW, SW, TVW)
        dec = device_decisions(eng, xs[k])
        grads.append(g.reshape(-1).clone())
        decs.append(dec)
        if k == 0:
            continue
        lo, go, ro = cpu_ref.closure_eval(xs[k].cpu(), tg, vgg_weights, CW, SW, TVW, decisions=dec)
        e_l = abs(float(l[-1].cpu()) - float(lo)) / float(lo)
        e_g = rel_l2(g.cpu().numpy(), go.numpy())
        report(f"{what}, teacher-forced at the device's iterate after step {k}: loss rel {e_l:.1e}, gradient rel-L2 under equal decisions {e_g:.1e}")
        assert e_l < 1e-5 and e_g < 2e-5
        check_rows(l[:-1].cpu().numpy().reshape(NLEV, 4), np.array(ro), 2e-5)

    # ---- (2) the step-2 direction from the DEVICE's own (g0, g1, s) against an fp64 recursion, and the iterate it gives
    s1 = first_step(grads[0])
    y1 = grads[1] - grads[0]
    d64, ys = _fp64_direction(grads[1].cpu(), y1.cpu(), s1.cpu())
    hd = ys / float(y1.double().dot(y1.double()))
    cond = float(grads[1].double().norm() / y1.double().norm())
    d_dev = eng.lbfgs_direction(grads[1], [y1], [s1], [1.0 / ys], hd, 0)
    e_d = rel_l2(d_dev.cpu().numpy(), d64.numpy())
    report(f"{what}, step-2 direction (nst_lbfgs_direction on the device's g0, g1, s) vs fp64: rel-L2 {e_d:.1e}; |g1| / |g1 - g0| = {cond:.1e} "
           f"(what a relative error of the gradient is multiplied by in y = g1 - g0), y.s = {ys:.3e}, H_diag = {hd:.3e}")
    assert e_d < 2e-5
    if len(moved) > 1 and moved[1]:
        e_x = rel_l2((xs[2].reshape(-1).double().cpu() - xs[1].reshape(-1).double().cpu()).numpy(), (lr2 * d64).numpy())
        report(f"{what}, the device's step 2 (x2 - x1) vs lr * the fp64 direction from its own gradients: rel-L2 {e_x:.1e}")
        assert e_x < 1e-4

    # ---- (3) the ORACLE's optimiser (torch's L-BFGS restated) on the oracle's closure evaluated under the device's
    # decisions at the corresponding point (closure 0: x0; closures 1, 2: the iterate after step 1; closure 3: after step 2)
    if not oracle_optimiser:
        assert not moved[1], "a job whose second step is accepted needs block (3)"
    st = cpu_ref.LbfgsState(max_eval=1)
    xo = x_init.reshape(-1).clone()
    calls, orc = [], []
    point = {0: 0, 1: 1, 2: 1, 3: 2}

    def closure(xf):
        k = len(calls)
        loss, g, r = cpu_ref.closure_eval(xf.view(1, 3, H0, W0), tg, vgg_weights, CW, SW, TVW, decisions=decs[point[k]])
        calls.append(float(loss))
        orc.append(np.array(r))
        return float(loss), g.reshape(-1)

    lr = 10.0
    o_tot = e_land = None
    if oracle_optimiser:
        for _ in range(2):
            cpu_ref.lbfgs_step(st, xo, lr, closure)
            lr *= 0.999 ** 2
        o_tot = np.array(calls)
        e_land = np.abs(o_tot - tot[:4]) / tot[:4]
        e_ref = np.abs(o_tot - ref[:4]) / ref[:4]
        e_x2 = rel_l2(xo.numpy(), xs[2 if moved[1] else 1].reshape(-1).cpu().numpy())       # (a rejected step 2 leaves x1)
        report(f"{what}, the oracle's optimiser under the DEVICE's decisions, closures 0-3: rel diff to the device " + np.array2string(e_land, precision=1)
               + ", to the reference's run " + np.array2string(e_ref, precision=1) + f"; the device's own closures vs the reference " + np.array2string(err[:4], precision=1)
               + f"; iterate after step 2 oracle vs device rel-L2 {e_x2:.1e}")
        assert e_land[:3].max() < 2e-5
    # ---- (4) every convolution direct (nst_options.h2_winograd = 0): the Winograd default does not move the landing
    other = StyleEngine(vgg_weights, 0, h2_winograd=False)
    try:
        setup(other, c_lv, s_lv)
        r2, st2, mv2, _ = lbfgs_run(other, dev(x_init), 4, nlev=NLEV)
    finally:
        other.close()
    t2 = totals(r2)
    e_dir = np.abs(t2[:4] - tot[:4]) / tot[:4]
    report(f"{what}, all-direct convolutions vs the default (Winograd launches), closures 0-3: " + np.array2string(e_dir, precision=1)
           + f"; vs the reference " + np.array2string(np.abs(t2[:4] - ref[:4]) / ref[:4], precision=1))
    assert mv2 == moved[:2] and e_dir[:3].max() < 2e-5
    # The landing of step 2: with the closure right at every visited point (1), the direction right given the gradients (2),
    # what remains is the map (g0, g1) -> x2 itself: y = g1 - g0 is |g1| / |y| times smaller than the gradients, so a
    # relative gradient difference e becomes ~e |g1| / |y| in H_diag = y.s / y.y and in the step.  Bounds: the oracle under
    # the device's decisions must land where the device lands far closer than the device is to the reference's run (the
    # decisions, not the products, carry the difference) ...
    LAND = 3e-2
    if moved[1]:
        # measured on the round-2 start image: device vs the reference's run 2.4e-2, all-direct build vs the device 9.7e-3 -
        # and the ORACLE under the device's decisions vs the device 8.5e-5: given the same near-tie ReLU / pooling decisions
        # the reference's arithmetic lands where the device lands
        assert err[3] < LAND and e_dir[3] < LAND
        assert e_land[3] < 2e-4, e_land
    else:
        # Rejected in the reference's run and here.  Where the first step is as short as it is from the reference's own
        # start image (y.s within two decades of torch's 1e-10 guard, |g1| / |y| ~ 3e3), the curvature pair is rounding
        # noise of two gradient evaluations and the two-loop recursion divides by it: the trial point's loss is a number of
        # one implementation's arithmetic, not of the algorithm (measured: reference 1.96e6, the oracle under the device's
        # decisions 78x that, the device 110x, the all-direct build 310x - every one of them far above f, so every run
        # rejects the step and keeps the same image; on the CPU alone the reference's fp32 y is 117 % away from the fp64 y
        # of the same two images: tools/diag_first_pair_noise.py, profiles/r03_lbfgs_first_pair_noise.txt).  What is
        # compared is that decision.
        runs = [(tot[3], tot[2]), (ref[3], ref[2]), (t2[3], t2[2])] + ([(o_tot[3], o_tot[2])] if o_tot is not None else [])
        for trial, kept in runs:
            assert trial > 2.0 * kept
    # ... and the run ends at the reference's loss level
    report(f"{what}: loss at the last accepted point, device {tot[-2] if not moved[-1] else tot[-1]:.6e} vs reference "
           f"{ref[-2] if not moved[-1] else ref[-1]:.6e} (rel {final_err:.2e})")
    assert final_err < (LAND if any(moved[1:]) else 1e-3)      # no amplified step in the run: SURVEY 8(c)'s 1e-3
    # rejected steps re-evaluate the kept image: bitwise the same row; their trial points are far up the loss surface
    for i in range(1, len(steps)):
        if not moved[i - 1]:
            first = steps[i - 1]                            # index of step i's first closure
            prev_first = steps[i - 2] if i >= 2 else 0
            assert np.array_equal(rows[first], rows[prev_first]), i
    check_summary(eng.unprepare_img(xs[1]).cpu(), fx, "after_1", atol=1e-4)


