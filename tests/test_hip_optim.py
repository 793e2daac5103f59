"""GPU: the optimisers (nst_opt_step, nst_adam_step, nst_lbfgs_direction) and whole trajectories against the oracle and
against fixtures produced by running the reference's NeuralStyleTransfer.process - including BASELINE configs 1 and 2 at
their full iteration counts (50 Adam iterations at 384x256; 500 L-BFGS closures at 768x512 + 384x256).

Two kinds of test:
* teacher-forced - the state comes from the oracle / the fixture, ONE update is made on the device: tight tolerances,
  stated where they are applied;
* free-running - the device walks the whole trajectory alone.  Adam's early steps are +-lr * sign(g), so a gradient
  entry whose sign differs (a ReLU decision that flipped, see hip_helpers.GRAD_RTOL) moves that pixel by +-lr and the
  pixel trajectories separate; what must agree is the loss level and, for L-BFGS, the accept / reject sequence.  The
  tolerances are the measured ones with headroom (measurements: $NST_TEST_REPORT, committed as
  profiles/r02_parity_measurements.txt)."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref
from hip_helpers import (CW, SW, TERMS, TVW, check_rows, check_summary, closure_vs_oracle_under_equal_decisions, dev, levels,
                         oracle_targets, rel_l2, report, rows_rel_err, setup)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(vgg_weights):
    from artstyletransfer_amd.engine import StyleEngine
    e = StyleEngine(vgg_weights, 0)
    yield e
    e.close()


# ---------------------------------------------------------------- Adam
def test_adam_update_kernel_vs_torch(eng):
    """nst_adam_step on random state against the oracle's AdamState.update (= torch's own _single_tensor_adam ops on the
    CPU): m and v bit for bit (the kernel rounds as torch's kernels do: fma forms, 1-beta from the double difference),
    x to 1 ulp-ish (torch's vectorised sqrt is not correctly rounded in ~0.6 % of the lanes)."""
    n = 3 * 67 * 45 + 3                      # n mod 4 = 0 would hide tail handling
    g = torch.Generator().manual_seed(0)
    grad = torch.randn(n, generator=g) * 3
    st = cpu_ref.AdamState(n)
    st.m = torch.randn(n, generator=g) * 0.5
    st.v = torch.rand(n, generator=g) * 4
    x = torch.randn(n, generator=g) * 50
    for k, lr in ((1, 9.99), (7, 9.93), (400, 6.7)):
        st.k = k - 1
        xd, md, vd = dev(x.clone()), dev(st.m.clone()), dev(st.v.clone())
        eng.adam_step(xd, dev(grad), md, vd, k, lr)
        xr = x.clone()
        st.update(xr, grad, lr)
        assert torch.equal(md.cpu(), st.m), k
        assert torch.equal(vd.cpu(), st.v), k
        np.testing.assert_allclose(xd.cpu().numpy(), xr.numpy(), rtol=3e-7, atol=1e-6)
        same = float((xd.cpu() == xr).float().mean())
        report(f"adam kernel k={k}: x bit-identical in {same:.4%} of the entries")
        assert same > 0.95


def test_adam_teacher_forced(eng, vgg_weights):
    """The oracle drives the trajectory.  At every x_k: the HIP closure reproduces loss and gradient, and ONE HIP Adam
    update (nst_adam_step) from the oracle's (x, m, v) state with the ORACLE's gradient lands on the oracle's next x,
    m, v: relative 1e-6 (bias corrections, beta2 and 1-beta2 all matter from k = 2 on)."""
    c, s = levels(64, 96, 2, 1), levels(64, 96, 2, 2)
    setup(eng, c, s)
    tg = oracle_targets(c, s, vgg_weights)
    x = cpu_ref.prepare_img(c[0]).contiguous()
    adam = cpu_ref.AdamState(x.numel())
    lr = 10.0
    for k in range(1, 7):
        lr *= 0.999
        loss, grad, rows = cpu_ref.closure_eval(x, tg, vgg_weights, CW, SW, TVW)
        closure_vs_oracle_under_equal_decisions(eng, x, tg, vgg_weights, f"adam teacher-forced k={k}", terms=TERMS[:1])
        xd, md, vd = dev(x.reshape(-1).clone()), dev(adam.m.clone()), dev(adam.v.clone())
        eng.adam_step(xd, dev(grad.reshape(-1)), md, vd, k, lr)
        xf = x.reshape(-1).clone()
        adam.update(xf, grad.reshape(-1), lr)                     # the oracle's step k
        assert adam.k == k
        np.testing.assert_allclose(md.cpu().numpy(), adam.m.numpy(), rtol=1e-6, atol=1e-30)
        np.testing.assert_allclose(vd.cpu().numpy(), adam.v.numpy(), rtol=1e-6, atol=1e-30)
        np.testing.assert_allclose(xd.cpu().numpy(), xf.numpy(), rtol=1e-6, atol=2e-6)
        x = xf.view(x.shape)


def test_adam_trajectory_vs_reference(eng, vgg_weights, golden):
    from artstyletransfer_amd.engine import PixelOptimizer
    fx = golden("traj_adam_64x96_L1_12")
    c, s = levels(64, 96, 2, 1), levels(64, 96, 2, 2)
    setup(eng, c, s)
    x = dev(cpu_ref.prepare_img(c[0]))
    opt = PixelOptimizer(eng, "adam")
    rows = []
    for k in range(12):
        info, r = opt.step(x, CW, SW, TVW)
        assert info.closures == 1 and info.total_closures == k + 1
        rows.append(r[0, :-1].reshape(2, 4))
        if k == 0:
            img = eng.unprepare_img(x).cpu().numpy()
            np.testing.assert_allclose(img, fx["after_1"], rtol=0, atol=2e-5)
    rows = np.array(rows)
    report(f"adam 64x96 12 steps free-running: worst level-total rel err {rows_rel_err(rows, fx['rows']):.2e}, "
           f"mean |img diff| {np.mean(np.abs(eng.unprepare_img(x).cpu().numpy() - fx['final'])):.2e}")
    check_rows(rows[:1], fx["rows"][:1], 2e-5)
    # free-running on a tiny image: see the module docstring (measured 3.7e-3 at worst in round 1)
    check_rows(rows, fx["rows"], 1e-2)
    assert np.mean(np.abs(eng.unprepare_img(x).cpu().numpy() - fx["final"])) < 2e-2
    assert info.lr == pytest.approx(10.0 * 0.999 ** 12, rel=1e-6)
    opt.close()


def _run_adam(eng, x, iters):
    from artstyletransfer_amd.engine import PixelOptimizer
    opt = PixelOptimizer(eng, "adam")
    rows, imgs = [], {}
    nlev = eng.levels
    for k in range(iters):
        info, r = opt.step(x, CW, SW, TVW)
        rows.append(r[0, :-1].reshape(nlev, 4))
        if k + 1 in (1, 10):
            imgs[k + 1] = eng.unprepare_img(x).cpu()
    opt.close()
    return np.array(rows), imgs, info


def _adam_final_with_direct_convolutions(vgg_weights, c, s, start_img, iters, default_last_rows, ref_last_rows):
    """The same Adam run on a second engine whose every convolution is direct (nst_options.h2_winograd = 0): (relative
    distance of its final total loss from the reference's, from the default build's)."""
    from artstyletransfer_amd.engine import StyleEngine
    other = StyleEngine(vgg_weights, 0, h2_winograd=False)
    try:
        setup(other, c, s)
        r2, _, _ = _run_adam(other, dev(cpu_ref.prepare_img(start_img)), iters)
    finally:
        other.close()
    mine, ref, dflt = r2[-1][:, 0].sum(), np.asarray(ref_last_rows)[:, 0].sum(), np.asarray(default_last_rows)[:, 0].sum()
    return abs(mine - ref) / ref, abs(mine - dflt) / dflt


def test_config1_adam_50_iterations_vs_reference(eng, vgg_weights, golden):
    """BASELINE config 1 (single 384x256 level, 50 Adam iterations, content image as the start) free-running on the
    device against the reference's own run (tests/golden/traj_adam_256x384_50.npz): the image after the first step,
    every closure's loss row, the final loss and the final image."""
    fx = golden("traj_adam_256x384_50")
    c, s = levels(256, 384, 1, 1), levels(256, 384, 1, 2)
    setup(eng, c, s)
    x = dev(cpu_ref.prepare_img(c[0]))
    rows, imgs, info = _run_adam(eng, x, 50)
    assert info.total_closures == 50 == int(fx["steps"][-1])
    worst1 = check_summary(imgs[1], fx, "img_after_1", atol=2e-5)
    final = eng.unprepare_img(x).cpu()
    d_final = float(np.max(np.abs(final.reshape(-1)[torch.from_numpy(fx["final.idx"])].numpy() - fx["final.val"])))
    err = np.abs(rows[:, 0, 0] - fx["rows"][:, 0, 0]) / fx["rows"][:, 0, 0]
    report(f"config 1 (adam 50 @256x384): first image max diff {worst1:.1e}; level-total rel err: first {err[0]:.1e}, "
           f"worst {err.max():.2e} at it {int(err.argmax())}, final {err[-1]:.2e}; final image max sampled diff {d_final:.2e}")
    check_rows(rows[:1], fx["rows"][:1], 2e-5)
    check_rows(rows, fx["rows"], 5e-3)                                   # every closure of the run
    # Final-loss parity.  SURVEY 8(c) started from 1e-3 "to be set by measurement".  Adam's early steps are +-lr sign(g), so
    # the run is chaotic in the last bits of the gradient: the SAME device with another equally valid fp32 arithmetic (every
    # convolution direct instead of Winograd where it applies) ends as far from the default as the default ends from the
    # reference - measured over the builds of rounds 2-3 on this 98 k-pixel job: 4.0e-4 ... 1.2e-3 against the reference.
    # The bound is 2.5e-3 here and stays 1e-3 on the 2 M-pixel headline job (tests/test_hip_config3.py: 4.6e-5 ... 5.2e-4),
    # where the loss averages over twenty times as many pixels.
    e_direct, d_direct = _adam_final_with_direct_convolutions(vgg_weights, c, s, c[0], 50, rows[-1], fx["rows"][-1])
    report(f"config 1: the all-direct arithmetic ends {e_direct:.2e} from the reference and {d_direct:.2e} from the default build")
    assert err[-1] < 2.5e-3 and e_direct < 2.5e-3
    assert float((final.double() ** 2).sum()) == pytest.approx(float(fx["final.sq_sum"]), rel=1e-3)
    assert d_final < 0.1                                                 # same picture ([0,1] units; Adam moves +-lr/255 per step)


def test_config2_geometry_adam_100_iterations_vs_reference(eng, vgg_weights, golden):
    """The config-2 pyramid (768x512 + 384x256) under Adam, 100 iterations - a job whose image moves at every step -
    against the reference's run (tests/golden/traj_adam_512x768_L1_100.npz)."""
    fx = golden("traj_adam_512x768_L1_100")
    c, s = levels(512, 768, 2, 1), levels(512, 768, 2, 2)
    setup(eng, c, s)
    x = dev(cpu_ref.prepare_img(c[0]))
    rows, imgs, info = _run_adam(eng, x, 100)
    assert info.total_closures == 100 == int(fx["steps"][-1])
    check_summary(imgs[1], fx, "after_1", atol=2e-5)
    d10 = float(np.mean(np.abs(imgs[10].reshape(-1)[torch.from_numpy(fx["after_10.idx"])].numpy() - fx["after_10.val"])))
    tot = rows[:, :, 0].sum(axis=1)
    ref = fx["rows"][:, :, 0].sum(axis=1)
    err = np.abs(tot - ref) / ref
    report(f"config-2 geometry (adam 100 @512x768+256x384): total-loss rel err first {err[0]:.1e}, worst {err.max():.2e} at it "
           f"{int(err.argmax())}, final {err[-1]:.2e}; mean sampled |img diff| after 10 its {d10:.2e}")
    check_rows(rows[:1], fx["rows"][:1], 2e-5)
    check_rows(rows, fx["rows"], 5e-3)
    # (final-loss bound: see test_config1_adam_50_iterations_vs_reference; measured 8.1e-4 ... 1.1e-3 over the builds)
    e_direct, d_direct = _adam_final_with_direct_convolutions(vgg_weights, c, s, c[0], 100, rows[-1], fx["rows"][-1])
    report(f"config-2 geometry: the all-direct arithmetic ends {e_direct:.2e} from the reference and {d_direct:.2e} from the default build")
    assert err[-1] < 2.5e-3 and e_direct < 2.5e-3
    final = eng.unprepare_img(x).cpu()
    assert float((final.double() ** 2).sum()) == pytest.approx(float(fx["final.sq_sum"]), rel=1e-3)


# ---------------------------------------------------------------- L-BFGS
def _run_lbfgs(eng, x, closures, max_eval, nlev):
    from artstyletransfer_amd.engine import PixelOptimizer
    opt = PixelOptimizer(eng, "lbfgs", lbfgs_max_eval=max_eval)
    rows, steps, moved, first = [], [], [], None
    total = 0
    while total < closures:
        info, r = opt.step(x, CW, SW, TVW)
        total = info.total_closures
        steps.append(total)
        moved.append(bool(info.accepted))
        rows.extend(list(r[:, :-1].reshape(-1, nlev, 4)))
        if first is None:
            first = eng.unprepare_img(x).cpu()
    hist = opt.history()
    opt.close()
    return np.array(rows), steps, moved, first, hist


@pytest.mark.parametrize("tag,max_eval", [("shipped", 1), ("legacy", 26)])
def test_lbfgs_trajectory_vs_reference(eng, vgg_weights, golden, tag, max_eval):
    fx = golden(f"traj_lbfgs_128x192_L1_{tag}")
    c, s = levels(128, 192, 2, 1), levels(128, 192, 2, 2)
    setup(eng, c, s)
    x = dev(cpu_ref.prepare_img(c[0]))
    rows, steps, moved, _, _ = _run_lbfgs(eng, x, 40, max_eval, 2)
    if tag == "shipped":
        # identical closure count per step and identical accept/reject sequence
        assert steps == list(fx["steps"])
        assert moved == list(fx["moved"])
        check_rows(rows, fx["rows"], 1e-3)
    else:
        # a real line search amplifies rounding differences (the CPU oracle itself drifts ~0.5% from the reference
        # after 40 closures when one gradient ulp differs): the first step (its interpolated trial points already
        # differ by 2e-3) and the final loss level
        n0 = int(fx["steps"][0])
        assert steps[0] == n0
        check_rows(rows[:2], fx["rows"][:2], 2e-5)
        check_rows(rows[:n0], fx["rows"][:n0], 1e-2)
        ref_last = fx["rows"][int(fx["steps"][-2])][:, 0].sum()
        mine_last = rows[steps[-2]][:, 0].sum() if steps[-2] < len(rows) else rows[-1][:, 0].sum()
        report(f"lbfgs legacy 128x192 40 closures: loss at the last step start {mine_last:.5e} vs reference {ref_last:.5e}")
        assert mine_last == pytest.approx(ref_last, rel=0.05)


def test_config2_lbfgs_500_closures_vs_reference(eng, vgg_weights, golden):
    """BASELINE config 2 at its full length: L=1 pyramid (768x512 + 384x256), L-BFGS exactly as the reference constructs
    it, 500 closure evaluations, free-running on the device, against the reference's own run
    (tests/golden/traj_lbfgs_512x768_L1_500.npz, 25 min of CPU): the closure count of every optimizer.step, the
    accept / reject sequence, the loss rows of all 500 closures, the final loss (<= 1e-3) and the image."""
    fx = golden("traj_lbfgs_512x768_L1_500")
    c, s = levels(512, 768, 2, 1), levels(512, 768, 2, 2)
    setup(eng, c, s)
    x = dev(cpu_ref.prepare_img(c[0]))
    rows, steps, moved, first, hist = _run_lbfgs(eng, x, 500, 1, 2)
    assert steps == list(fx["steps"]) and steps[-1] == 500
    assert moved == list(fx["moved"])
    tot, ref = rows[:, :, 0].sum(axis=1), fx["rows"][:, :, 0].sum(axis=1)
    err = np.abs(tot - ref) / ref
    report(f"config 2 (lbfgs 500 @512x768+256x384): {sum(moved)} of {len(moved)} steps accepted (reference {int(fx['moved'].sum())}), "
           f"history pairs {hist[0]}; total-loss rel err worst {err.max():.2e}, final {err[-1]:.2e}")
    check_rows(rows, fx["rows"], 1e-3)
    assert err[-1] < 1e-3
    check_summary(first, fx, "after_1", atol=1e-4)
    final = eng.unprepare_img(x).cpu()
    worst = check_summary(final, fx, "final", atol=2e-3)
    report(f"config 2: final image max sampled diff {worst:.2e}")


def test_config2_geometry_lbfgs_line_search_vs_reference(eng, vgg_weights, golden):
    """The same pyramid with the 25-evaluation line search (max_eval = 26: what the reference's constructor arguments meant
    before torch 2.10): steps are accepted, the curvature history fills - 100 closures against the reference's run.  A
    line search turns one-ulp differences of f and g.d into different trial points, so what is compared is the first
    step exactly and the loss level along the run."""
    fx = golden("traj_lbfgs_512x768_L1_legacy_100")
    c, s = levels(512, 768, 2, 1), levels(512, 768, 2, 2)
    setup(eng, c, s)
    x = dev(cpu_ref.prepare_img(c[0]))
    rows, steps, moved, _, hist = _run_lbfgs(eng, x, 100, 26, 2)
    # the first closure (x0) and the first trial point (t = min(1, 1/|g|_1) lr) are the reference's; from the first
    # interpolated trial point on the runs part (measured: the reference spends 8 closures in its first step, the device 7)
    check_rows(rows[:2], fx["rows"][:2], 2e-5)
    # loss at the start of every optimizer.step (the accepted points) against the reference's, step by step while both
    # runs made the same number of closures, and the level reached at the end
    ref_starts = [0] + [int(v) for v in fx["steps"][:-1]]
    my_starts = [0] + steps[:-1]
    ref_f = np.array([fx["rows"][i][:, 0].sum() for i in ref_starts if i < len(fx["rows"])])
    my_f = np.array([rows[i][:, 0].sum() for i in my_starts if i < len(rows)])
    k = min(len(ref_f), len(my_f))
    rel = np.abs(my_f[:k] - ref_f[:k]) / ref_f[:k]
    report(f"config-2 geometry (lbfgs max_eval 26, 100 closures): {len(steps)} steps (reference {len(fx['steps'])}), history pairs "
           f"{hist[0]}; accepted-point loss rel err per step {np.array2string(rel, precision=1)}; last {my_f[-1]:.5e} vs {ref_f[-1]:.5e}")
    assert all(moved) and hist[0] >= len(steps) - 2
    # measured: 32 steps in both runs, accepted-point losses within 5.3e-3 of the reference's at every step
    assert rel[0] < 1e-5 and rel.max() < 3e-2
    assert my_f[-1] < 0.8 * my_f[0]                                  # the job makes progress ...
    assert my_f[-1] == pytest.approx(ref_f[-1], rel=3e-2)            # ... to the reference's loss level


def _config3_job(eng):
    c, s = levels(1024, 1536, 3, 1), levels(1024, 1536, 3, 2)
    setup(eng, c, s)
    init = (0.7 * c[0] + 0.3 * cpu_ref.synthetic_image(1024, 1536, seed=3)).astype(np.float32)
    return dev(cpu_ref.prepare_img(init))


def test_config3_workload_adam_prefix_vs_reference(eng, vgg_weights, golden):
    """BASELINE config 3's workload - the one bench.py times: L=2, 1536x1024 + 768x512 + 384x256 - under Adam for the 16
    iterations the CPU reference affords in the container (tests/golden/traj_adam_1024x1536_L2_16.npz, make_fixtures.py
    fx_config3_prefix): every closure's loss rows on all three levels, the image after the first step and after the last."""
    fx = golden("traj_adam_1024x1536_L2_16")
    x = _config3_job(eng)
    rows, imgs, info = _run_adam(eng, x, 16)
    assert info.total_closures == 16 == int(fx["steps"][-1])
    worst1 = check_summary(imgs[1], fx, "after_1", atol=2e-5)
    tot, ref = rows[:, :, 0].sum(axis=1), fx["rows"][:, :, 0].sum(axis=1)
    err = np.abs(tot - ref) / ref
    final = eng.unprepare_img(x).cpu()
    d_final = float(np.max(np.abs(final.reshape(-1)[torch.from_numpy(fx["final.idx"])].numpy() - fx["final.val"])))
    report(f"config-3 workload (adam 16 @L=2): first image max diff {worst1:.1e}; total-loss rel err first {err[0]:.1e}, worst "
           f"{err.max():.2e} at it {int(err.argmax())}, final {err[-1]:.2e}; final image max sampled diff {d_final:.2e}")
    check_rows(rows[:1], fx["rows"][:1], 2e-5)
    check_rows(rows, fx["rows"], 2e-3)
    assert err[-1] < 1e-3
    assert float((final.double() ** 2).sum()) == pytest.approx(float(fx["final.sq_sum"]), rel=1e-4)


def test_config3_workload_lbfgs_prefix_vs_reference(eng, vgg_weights, golden):
    """The round-2 fixture of the config-3 workload (tests/golden/traj_lbfgs_1024x1536_L2_16.npz: 16 closures of L-BFGS as
    the reference constructs it from a content image blended with synthetic noise - a start from which the reference's run
    ACCEPTS its second step, one lr = 10 step along a direction scaled by (y.s)/(y.y) with y the difference of two
    gradients a 1/|g|_1-long step apart).  Closure counts, accept / reject sequence, loss rows; and that landing point taken
    apart (hip_helpers.lbfgs_vs_reference_taken_apart): the closure teacher-forced at the device's own iterates under equal
    decisions (losses 1e-5, whole gradient 2e-5), the direction from the device's own gradients against fp64, the oracle's
    optimiser under the device's decisions, the all-direct build.  (tests/test_hip_config3.py runs the same on the job the
    reference's own driver builds.)"""
    from hip_helpers import lbfgs_vs_reference_taken_apart
    c, s = levels(1024, 1536, 3, 1), levels(1024, 1536, 3, 2)
    init = (0.7 * c[0] + 0.3 * cpu_ref.synthetic_image(1024, 1536, seed=3)).astype(np.float32)
    lbfgs_vs_reference_taken_apart(eng, vgg_weights, c, s, cpu_ref.prepare_img(init), golden("traj_lbfgs_1024x1536_L2_16"), 16,
                                   "config-3 workload (lbfgs 16 closures @L=2, round-2 start image)")


@pytest.mark.parametrize("gram", [True, False])
@pytest.mark.parametrize("h,w", [(35, 51), (64, 96)])
def test_lbfgs_update_arithmetic_teacher_forced(vgg_weights, h, w, gram):
    """The optimiser arithmetic alone - two-loop recursion on the device, line search, curvature pairs - against the
    oracle's L-BFGS driven by the SAME closure (the HIP one), so that only the update arithmetic can differ:
    25-evaluation line search, lr 1, so that steps are accepted and the history grows.  35 x 51 gives n = 5355
    (n mod 4 = 3: the scalar tails of the vector kernels).  gram (nst_options.lbfgs_gram) True: the default direction
    from inner products (one multi-dot and one multi-axpy pass over the history); False: the sequential recursion, one
    fused launch per pair."""
    from artstyletransfer_amd.engine import PixelOptimizer, StyleEngine
    eng = StyleEngine(vgg_weights, 0, lbfgs_gram=gram)
    try:
        c, s = levels(h, w, 1, 1), levels(h, w, 1, 2)
        setup(eng, c, s)
        x0 = cpu_ref.prepare_img((0.6 * c[0] + 0.4 * s[0]).astype(np.float32)).contiguous()

        def hip_closure(flat):
            g, l = eng.closure(dev(flat.view_as(x0)), CW, SW, TVW)
            return float(l[-1].cpu()), g.cpu().reshape(-1).clone()

        steps = 10
        st = cpu_ref.LbfgsState(max_eval=26)                     # oracle optimiser on the HIP closure
        xo = x0.clone().reshape(-1)
        lr, ref_losses, ref_evals = 1.0, [], []
        for _ in range(steps):
            before = st.func_evals
            ref_losses.append(cpu_ref.lbfgs_step(st, xo, lr, hip_closure))
            ref_evals.append(st.func_evals - before)
            lr *= 0.999 ** ref_evals[-1]
        xh = dev(x0.clone())
        opt = PixelOptimizer(eng, "lbfgs", 1.0, 26)
        losses, evals = [], []
        for _ in range(steps):
            info, _rows = opt.step(xh, CW, SW, TVW)
            losses.append(float(info.loss)); evals.append(int(info.closures))
        assert opt.history()[0] == len(st.old_dirs) == info.history
        opt.close()
        assert evals == ref_evals                                # the same line-search decisions in every step
        assert len(st.old_dirs) >= steps - 2                     # the history did grow
        # The dot products differ in rounding (torch: fp32 pairwise; here: fp32 per lane, double across lanes) and the
        # line search amplifies that from step to step: measured 5e-6 / 6e-7 / 1.2e-5 in steps 2-4, 3e-3 by step 10
        # (35 x 51); below 2e-4 throughout at 64 x 96.
        np.testing.assert_allclose(losses[:4], ref_losses[:4], rtol=5e-5)
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-2)
        assert rel_l2(xh.cpu().numpy().reshape(-1), xo.numpy()) < 5e-2
    finally:
        eng.close()


@pytest.mark.parametrize("form", [0, 1])
@pytest.mark.parametrize("m", [0, 1, 7, 100])
def test_lbfgs_direction_from_an_oracle_state(eng, m, form):
    """nst_lbfgs_direction: d = -H g from a GIVEN curvature history (random, positive y.s) against the oracle's two-loop
    recursion (torch:optim/lbfgs.py:396-442 restated in cpu_ref.lbfgs_step) evaluated in fp64 on the same fp32 vectors.
    Both forms; m = 100 is a full history.  rel-L2 <= 2e-5 (measured ~1e-6 for the sequential form; the inner-product
    form subtracts in coefficient space and loses a digit more on near-collinear histories)."""
    n = 3 * 53 * 37 + 2
    g = torch.Generator().manual_seed(100 + m)
    grad = torch.randn(n, generator=g)
    ys, ss, ro = [], [], []
    for i in range(m):
        s_i = torch.randn(n, generator=g) * 0.1
        y_i = 2.0 * s_i + 0.3 * torch.randn(n, generator=g) * 0.1        # positive curvature: y.s > 0
        ys.append(y_i); ss.append(s_i)
        ro.append(float(1.0 / y_i.dot(s_i)))
    h_diag = float(ys[-1].dot(ss[-1]) / ys[-1].dot(ys[-1])) if m else 1.0
    # fp64 two-loop recursion
    q = -grad.double()
    al = [0.0] * m
    for i in range(m - 1, -1, -1):
        al[i] = float(ss[i].double().dot(q)) * ro[i]
        q = q - al[i] * ys[i].double()
    r = q * h_diag
    for i in range(m):
        be = float(ys[i].double().dot(r)) * ro[i]
        r = r + (al[i] - be) * ss[i].double()
    d = eng.lbfgs_direction(dev(grad), [dev(t) for t in ys], [dev(t) for t in ss], ro, h_diag, form)
    err = rel_l2(d.cpu().numpy(), r.numpy())
    report(f"lbfgs direction m={m} form={form}: rel-L2 vs fp64 recursion {err:.2e}")
    assert err < 2e-5
