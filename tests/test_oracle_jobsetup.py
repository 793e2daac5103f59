"""CPU: the job-driver logic of SURVEY rows f-1 / f-2 against what the REFERENCE'S OWN `neural_style_transfer()` produced
(tests/golden/jobsetup.npz, written by make_fixtures.py fx_jobsetup: the reference's job driver executed unmodified, its
four cv2 operator calls served by oracle/cv2_ref.py).  Pinned to the reference by this file: the pyramid size rule and
level order (neural_style_transfer.py:211-226, :249-263), the granularity -> spot-grid rule and the envelope accumulation
(:265-313), the draw order of np.random.permutation (:422-432), gaussian_mask (:396-418), the Sobel / clip / blur / a = 5
blend weight (:331-343) and the init-method branch with the `level` it resizes to (:350-362).  Only the four cv2 operators
themselves stay "pinned by independent implementation" (tests/test_oracle_cv2.py).

* oracle/cv2_ref.py restates that logic: held BIT-EXACT (same operator arithmetic, so any difference is logic);
* artstyletransfer_amd/host_image.py (the product's `resize`, `gaussian_mask`, `make_style_noise` helpers and the host
  restatement of device_image.py): held to the fp32-vs-double tolerance of its bicubic kernel.
The device path (device_image.py + image_ops.hip) is held to the same fixture in tests/test_hip_parity.py."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref, cv2_ref

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


from jobsetup_cases import JOBSETUP_CASES as CASES  # noqa: E402  (numbers only: sizes, seeds, Config keyword arguments)

DEFAULTS = dict(init_method="content+noise", noise_factor=0.95, noise_levels=(9, 18, 36, -1, 0),
                noise_levels_central_amplitude=(0.30, 0.20, 0.10, 0.20, 0.20),
                noise_levels_peripheral_amplitude=(0.20, 0.30, 0.40, 0.10, 0.00),
                noise_levels_dispersion=(0.20, 0.30, 0.40, 0.60, 0.30))            # config.py:6-18


# sampled values are compared bit for bit; the fixture's moments were summed by torch, here by numpy (another order of the
# same double additions)
EXACT_SQ = 1e-13


def job_inputs(case):
    tag, (ch, cw, cs), (sh, sw, ss), seed, kw = case
    cfg = dict(DEFAULTS)
    cfg.update(kw)
    return tag, cpu_ref.synthetic_image(ch, cw, cs), cpu_ref.synthetic_image(sh, sw, ss), seed, cfg


def summary_diff(arr, fx, key):
    """(max |sampled value - fixture|, relative error of the sum of squares) of an array against a make_fixtures.summarize
    record; shapes must be equal."""
    a = np.ascontiguousarray(arr)
    assert list(a.shape) == list(fx[f"{key}.shape"]), (key, a.shape, fx[f"{key}.shape"])
    flat = a.reshape(-1)
    d = float(np.max(np.abs(flat[fx[f"{key}.idx"]].astype(np.float64) - fx[f"{key}.val"].astype(np.float64))))
    sq = float((flat.astype(np.float64) ** 2).sum())
    return d, abs(sq - float(fx[f"{key}.sq_sum"])) / max(float(fx[f"{key}.sq_sum"]), 1e-30)


def oracle_job(content, style, seed, cfg):
    """The oracle's restatement of the job driver: pyramids highest resolution first + initial image."""
    n = cfg["levels_num"]
    c_lv = [cv2_ref.resize_cubic(content, *cv2_ref.level_size(*content.shape[:2], l)) for l in range(n - 1, -1, -1)]
    s_lv = [cv2_ref.resize_cubic(style, *cv2_ref.level_size(*style.shape[:2], l)) for l in range(n - 1, -1, -1)]
    np.random.seed(seed)
    init, tag = cv2_ref.initial_image(cfg["init_method"], content, style, c_lv[0], s_lv[0], max(n - 1, 0), cfg["noise_factor"],
                                      cfg["noise_levels"], cfg["noise_levels_central_amplitude"],
                                      cfg["noise_levels_peripheral_amplitude"], cfg["noise_levels_dispersion"])
    return c_lv, s_lv, init, tag


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_job_driver_is_the_references(golden, case):
    fx = golden("jobsetup")
    tag, content, style, seed, cfg = job_inputs(case)
    c_lv, s_lv, init, name = oracle_job(content, style, seed, cfg)
    assert [list(a.shape) for a in c_lv] == fx[f"{tag}.content_shapes"].tolist()       # order: highest resolution first
    assert [list(a.shape) for a in s_lv] == fx[f"{tag}.style_shapes"].tolist()
    for l, (c, s) in enumerate(zip(c_lv, s_lv)):
        assert c.dtype == np.float32 and s.dtype == np.float32
        for arr, key in ((c, f"{tag}.content{l}"), (s, f"{tag}.style{l}")):
            d, sq = summary_diff(arr, fx, key)
            assert d == 0.0 and sq < EXACT_SQ, (key, d, sq)
    assert str(init.dtype) == str(fx[f"{tag}.init_dtype"])
    d, sq = summary_diff(init, fx, f"{tag}.init")
    assert d == 0.0 and sq < EXACT_SQ, (tag, d, sq)          # bit-exact: the same RNG draws in the same order
    expect = {"random": "random", "content": "content-name", "style": "style-name"}[name]
    assert expect == str(fx[f"{tag}.init_name"])
    assert float(fx[f"{tag}.lr_start"]) == 10.0                                          # :367


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_host_mirror_job_driver_vs_reference(golden, case):
    """host_image.py computes its bicubic in fp32 (torch's kernel); the reference-made fixture (oracle operators) in
    double.  Bound as in tests/test_oracle_cv2.py: 2 * (2e-6 + 1.5e-7 * max(h, w)) per resize, twice where an up-sampled
    noise grid (white-noise contrast) is blended."""
    from artstyletransfer_amd import host_image as hi
    fx = golden("jobsetup")
    tag, content, style, seed, cfg = job_inputs(case)
    n = cfg["levels_num"]
    c_lv = [hi.resize_to_level(content, l) for l in range(n - 1, -1, -1)]
    s_lv = [hi.resize_to_level(style, l) for l in range(n - 1, -1, -1)]
    tol = lambda a: 2 * (2e-6 + 1.5e-7 * max(a.shape[:2]))
    for l, (c, s) in enumerate(zip(c_lv, s_lv)):
        assert summary_diff(c, fx, f"{tag}.content{l}")[0] <= tol(c)
        assert summary_diff(s, fx, f"{tag}.style{l}")[0] <= tol(s)
    np.random.seed(seed)
    init, name = hi.initial_image(cfg["init_method"], content, style, c_lv[0], s_lv[0], max(n - 1, 0), cfg["noise_factor"],
                                  cfg["noise_levels"], cfg["noise_levels_central_amplitude"],
                                  cfg["noise_levels_peripheral_amplitude"], cfg["noise_levels_dispersion"])
    d, sq = summary_diff(init, fx, f"{tag}.init")
    assert d <= 4 * tol(init) and sq < 1e-5, (tag, d, sq)
    assert {"random": "random", "content": "content-name", "style": "style-name"}[name] == str(fx[f"{tag}.init_name"])


def test_public_helpers_follow_the_reference(golden):
    """The helpers the reference exposes by name - resize(img, level) (async), gaussian_mask, make_style_noise - on the
    product's module against the fixture's first case (level shapes and the level-0 content image)."""
    import asyncio
    import artstyletransfer_amd.neural_style_transfer as nst
    fx = golden("jobsetup")
    tag, content, style, seed, cfg = job_inputs(CASES[0])
    top = asyncio.run(nst.resize(content, 1))
    low = asyncio.run(nst.resize(content, 0))
    assert list(top.shape) == fx[f"{tag}.content_shapes"][0].tolist() and list(low.shape) == fx[f"{tag}.content_shapes"][1].tolist()
    assert summary_diff(low, fx, f"{tag}.content1")[0] < 1e-4
    m = nst.gaussian_mask((40, 60, 3), 0.3, 0.2, 0.25)
    np.testing.assert_allclose(m, cv2_ref.gaussian_mask((40, 60, 3), 0.3, 0.2, 0.25), atol=1e-15)
    np.random.seed(4)
    a = nst.make_style_noise(style, (7, 9, 3))
    np.random.seed(4)
    b = cv2_ref.make_style_noise(style, (7, 9, 3))
    np.testing.assert_allclose(a, b, atol=1e-5)
