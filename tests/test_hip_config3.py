"""GPU: BASELINE config 3 - the workload bench.py times (L=2: 1536x1024 + 768x512 + 384x256) - against runs of the
REFERENCE'S OWN job driver: tests/golden/job_*_1024x1536_L2.npz are written by make_fixtures.py fx_config3_* from the
reference's `neural_style_transfer()` with Config() defaults except levels_num = 3, `content+noise` init under
np.random.seed(0), synthetic 3:2 originals.  The start image is therefore the reference's own
structured-noise image (its cv2 operator calls served by oracle/cv2_ref.py); here it is rebuilt by the oracle's restatement
of that job driver, which tests/test_oracle_jobsetup.py holds bit-exact to the reference, and checked against the fixture.

* Adam, 100 iterations (the image moves at every step): every closure's loss rows, the final loss <= 1e-3;
* L-BFGS as the reference constructs it: closure count per optimizer.step and accept / reject sequence identical, and the
  landing point of the accepted steps taken apart - teacher-forced closure at the device's own iterates (losses 1e-5,
  whole gradient 2e-5 under equal decisions), the step-2 direction from the device's own (g0, g1, s) against an fp64
  recursion, the oracle's optimiser driven under the device's decisions, the same run with every convolution direct;
* L-BFGS with the 25-evaluation line search (steps accepted, history filling), 40 closures.
* The product's `neural_style_transfer()` generator end to end on the same job."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref, cv2_ref
from hip_helpers import (CW, SW, TVW, check_rows, dev, lbfgs_run, lbfgs_vs_reference_taken_apart, report, setup, totals)

pytestmark = pytest.mark.gpu

H0, W0, NLEV = 1024, 1536, 3
CFG = dict(noise_factor=0.95, noise_levels=(9, 18, 36, -1, 0), central=(0.30, 0.20, 0.10, 0.20, 0.20),
           peripheral=(0.20, 0.30, 0.40, 0.10, 0.00), dispersion=(0.20, 0.30, 0.40, 0.60, 0.30))       # config.py:14-18


@pytest.fixture(scope="module")
def job(golden):
    """The config-3 job as the reference's driver builds it: (content levels, style levels, start image), numpy HWC."""
    content = cpu_ref.synthetic_image(H0, W0, seed=1)
    style = cpu_ref.synthetic_image(H0, W0, seed=2)
    c_lv = [cv2_ref.resize_cubic(content, *cv2_ref.level_size(H0, W0, l)) for l in range(NLEV - 1, -1, -1)]
    s_lv = [cv2_ref.resize_cubic(style, *cv2_ref.level_size(H0, W0, l)) for l in range(NLEV - 1, -1, -1)]
    np.random.seed(0)
    init, tag = cv2_ref.initial_image("content+noise", content, style, c_lv[0], s_lv[0], NLEV - 1, CFG["noise_factor"],
                                      CFG["noise_levels"], CFG["central"], CFG["peripheral"], CFG["dispersion"])
    assert tag == "content" and init.dtype == np.float32 and init.shape == (H0, W0, 3)
    fx = golden("job_adam100_1024x1536_L2")
    # the start image of the reference's run (numpy's exp / the summation order of the kernel sums may differ in the
    # last bit between hosts: 1e-6 in [0, 1] units, not bit-exact as in the container)
    flat = init.reshape(-1)
    assert np.max(np.abs(flat[fx["init.idx"]] - fx["init.val"])) < 1e-6
    assert float((flat.astype(np.float64) ** 2).sum()) == pytest.approx(float(fx["init.sq_sum"]), rel=1e-9)
    return c_lv, s_lv, init


@pytest.fixture(scope="module")
def eng(vgg_weights):
    from artstyletransfer_amd.engine import StyleEngine
    e = StyleEngine(vgg_weights, 0)
    yield e
    e.close()


def _sampled(img, fx, key):
    flat = torch.as_tensor(img).reshape(-1)
    return np.abs(flat[torch.from_numpy(fx[f"{key}.idx"])].numpy() - fx[f"{key}.val"])


# ---------------------------------------------------------------- Adam, 100 iterations
def test_config3_adam_100_iterations_vs_reference(eng, job, golden):
    from artstyletransfer_amd.engine import PixelOptimizer
    fx = golden("job_adam100_1024x1536_L2")
    c_lv, s_lv, init = job
    setup(eng, c_lv, s_lv)
    x = dev(cpu_ref.prepare_img(init))
    opt = PixelOptimizer(eng, "adam")
    rows, imgs = [], {}
    for k in range(100):
        info, r = opt.step(x, CW, SW, TVW)
        rows.append(r[0, :-1].reshape(NLEV, 4))
        if k + 1 in (1, 2, 4, 10, 50, 100):
            imgs[k + 1] = eng.unprepare_img(x).cpu()
    opt.close()
    rows = np.array(rows)
    assert info.total_closures == 100 == int(fx["steps"][-1]) and list(fx["steps"]) == list(range(1, 101))
    assert np.allclose(fx["percent"], np.arange(1, 101), atol=1e-9)              # percent = step / iters_num * 100 (:370)
    tot, ref = totals(rows), totals(fx["rows"])
    err = np.abs(tot - ref) / ref
    d1 = float(_sampled(imgs[1], fx, "after_1").max())
    dn = {k: float(_sampled(imgs[k], fx, f"after_{k}" if k < 100 else "final").mean()) for k in (2, 4, 10, 50, 100)}
    report(f"config 3 (adam 100 @L=2, the reference's own start image): total-loss rel err first {err[0]:.1e}, worst {err.max():.2e} at it "
           f"{int(err.argmax())}, final {err[-1]:.2e}; loss {ref[0]:.4e} -> {ref[-1]:.4e}; first image max diff {d1:.1e}; mean sampled "
           f"|img diff| after 2/4/10/50/100 its " + "/".join(f"{dn[k]:.1e}" for k in (2, 4, 10, 50, 100)))
    check_rows(rows[:1], fx["rows"][:1], 2e-5)
    check_rows(rows, fx["rows"], 5e-3)                                   # every closure, every level, every term
    assert err[-1] < 1e-3                                                # final-loss parity
    assert d1 < 2e-5
    final = imgs[100]
    assert float((final.double() ** 2).sum()) == pytest.approx(float(fx["final.sq_sum"]), rel=1e-3)
    assert dn[100] < 2e-2                                                # the same picture, [0, 1] units (Adam moves +-lr/255 a step)


def test_config3_product_generator_end_to_end(vgg_weights, job, golden):
    """The drop-in entry point itself on config 3: `neural_style_transfer()` (device pyramid + device structured-noise
    image + nst_opt_step loop + per-step yield), Adam, 100 iterations, against the reference's run of the same call."""
    import asyncio
    from artstyletransfer_amd import config, neural_nets
    import artstyletransfer_amd.neural_style_transfer as nst
    neural_nets.set_weights(vgg_weights)
    fx = golden("job_adam100_1024x1536_L2")
    content = cpu_ref.synthetic_image(H0, W0, seed=1)
    style = cpu_ref.synthetic_image(H0, W0, seed=2)
    cfg = config.Config(levels_num=3, optimizer="adam", iters_num=100)

    async def run():
        out = []
        np.random.seed(0)
        async for percent, img in nst.neural_style_transfer(
                nst.ContentStylePair(("c", content), ("s", style)), cfg.content_weight, cfg.style_weight, cfg.tv_weight,
                cfg.optimizer, cfg.model, cfg.init_method, cfg.iters_num, cfg.levels_num, cfg.noise_factor, cfg.noise_levels,
                cfg.noise_levels_central_amplitude, cfg.noise_levels_peripheral_amplitude, cfg.noise_levels_dispersion):
            out.append((percent, img if len(out) in (0, 9, 49, 99) else None))
        return out

    out = asyncio.run(run())
    assert np.allclose([p for p, _ in out], fx["percent"], atol=1e-9) and len(out) == 100
    s1 = _sampled(out[0][1], fx, "after_1")
    d1, far1 = float(s1.mean()), float((s1 > 1e-3).mean())
    d10, d50, d100 = (float(_sampled(out[k][1], fx, key).mean()) for k, key in ((9, "after_10"), (49, "after_50"), (99, "final")))
    sq = float((out[99][1].astype(np.float64) ** 2).sum())
    report(f"config 3 through the product's neural_style_transfer() (adam 100): first yielded image mean sampled |diff| vs the reference's {d1:.1e} ({far1:.2%} of the samples beyond 1e-3: "
           f"Adam's first step is lr * sign(g) = 0.039, so a gradient entry whose sign differs moves that pixel by 0.078); "
           f"mean sampled |diff| after 10/50/100 its {d10:.1e}/{d50:.1e}/{d100:.1e}; final sum of squares rel {abs(sq / float(fx['final.sq_sum']) - 1):.1e}")
    assert out[0][1].shape == (H0, W0, 3) and out[0][1].dtype == np.float32
    assert d1 < 1e-4 and far1 < 0.01       # device-built start image (<= 2e-5 from the reference's) + one Adam step
    assert d100 < 2e-2 and sq == pytest.approx(float(fx["final.sq_sum"]), rel=1e-3)


# ---------------------------------------------------------------- L-BFGS as the reference constructs it
def test_config3_lbfgs_vs_reference_and_its_landing_points_taken_apart(eng, vgg_weights, job, golden):
    """The headline job under L-BFGS as the reference constructs it, from the reference's own start image, 24 closures."""
    c_lv, s_lv, init = job
    # (from this start image every run rejects the second step - its curvature pair is rounding noise, measured in round 3:
    # profiles/r03_parity_measurements.txt - so the oracle-optimiser block, which has only that decision to compare here, is
    # left to the round-2 start image of tests/test_hip_optim.py, where the second step is accepted)
    lbfgs_vs_reference_taken_apart(eng, vgg_weights, c_lv, s_lv, cpu_ref.prepare_img(init), golden("job_lbfgs24_1024x1536_L2"), 24,
                                   "config 3 (lbfgs as shipped, 24 closures @L=2, the reference's own start image)",
                                   oracle_optimiser=False)


def test_config3_lbfgs_line_search_vs_reference(eng, job, golden):
    """The same job with the 25-evaluation line search (max_eval = 26: what the reference's constructor arguments meant
    before torch 2.10): steps are accepted, the curvature history fills - 40 closures.  A line search turns one-ulp
    differences of f and g.d into other trial points: compared are the first closures exactly and the loss at the start
    of every optimizer.step."""
    fx = golden("job_lbfgs_legacy40_1024x1536_L2")
    c_lv, s_lv, init = job
    setup(eng, c_lv, s_lv)
    rows, steps, moved, _ = lbfgs_run(eng, dev(cpu_ref.prepare_img(init)), 40, max_eval=26)
    check_rows(rows[:2], fx["rows"][:2], 2e-5)
    ref_steps = [int(v) for v in fx["steps"]]
    # accepted-point loss after n closures: the first closure of step k+1 evaluates the image step k kept
    ref_n = [0] + [n for n in ref_steps[:-1] if n < len(fx["rows"])]
    my_n = [0] + [n for n in steps[:-1] if n < len(rows)]
    ref_f = np.array([totals(fx["rows"][i:i + 1])[0] for i in ref_n])
    my_f = np.array([totals(rows[i:i + 1])[0] for i in my_n])
    ref_at = np.exp(np.interp(my_n, ref_n, np.log(ref_f)))           # the reference's curve at the device's closure counts
    ratio = my_f / ref_at
    report(f"config 3 (lbfgs max_eval 26, 40 closures @L=2): {len(steps)} steps (reference {len(ref_steps)}), closures after each step {steps} "
           f"(reference {ref_steps}); accepted-point loss over the reference's at equal closure counts {np.array2string(ratio, precision=3)}; "
           f"last {my_f[-1]:.5e} after {my_n[-1]} closures vs {ref_f[-1]:.5e} after {ref_n[-1]}")
    assert all(moved)
    # The first step starts from a 1/|g|_1-long trial point whose curvature information is rounding noise (see the shipped
    # run above): its interpolated trial points differ from the first one on (measured: the reference spends 7 closures in
    # its first step, the device 10), and the runs are two different descents of the same surface afterwards.  Held: the
    # loss reached per closure spent stays within a factor 1.5 of the reference's curve (measured <= 1.25) and ends within 5 % of it, at a fifth of
    # the start value.
    assert abs(ratio[0] - 1) < 1e-5 and np.all(np.abs(np.log(ratio)) < np.log(1.5))
    assert my_f[-1] < 0.25 * my_f[0] and abs(np.log(ratio[-1])) < np.log(1.05)
