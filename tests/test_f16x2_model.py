"""CPU model of the arithmetic of conv_h2.hip / gram_h2 (two scaled fp16 pieces per fp32 operand, main and
2^-11-weighted cross products in separate fp32 accumulators) against fp64, next to the alternatives it replaced.
This pins the *claim* the GPU path rests on - that the scheme is as accurate as an fp32 matrix instruction - with
plain numpy, where no GPU is needed; the GPU tests then check the kernels against the oracle and against fp64."""
import numpy as np
import pytest

K, M = 4608, 1500          # K of the widest VGG19 layer (512 channels x 9 taps)


def _operands(seed):
    rng = np.random.default_rng(seed)
    a = np.maximum(rng.normal(0, 100, (M, K)), 0).astype(np.float32)       # post-ReLU activations, half of them 0
    a *= rng.random((M, K)) < 0.5
    w = rng.normal(0, 0.02, K).astype(np.float32)
    return a, w


def _acc_f32(chains):
    """MFMA-like accumulation: per 16-wide K block every chain adds its exact block sum into the fp32 accumulator."""
    acc = np.zeros(M, np.float32)
    for k0 in range(0, K, 16):
        for t in chains:
            acc = (acc.astype(np.float64) + t[:, k0:k0 + 16].sum(1)).astype(np.float32)
    return acc


def _scale(x):      # power of two bringing max|x| into [2^14, 2^15): conv_h2.hip::tensor_scale / make_h2
    return 2.0 ** (14 - np.floor(np.log2(np.abs(x).max())))


def _cut2(x, s):    # conv_h2.hip::cut2x4
    xs = (x * np.float32(s)).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = ((xs - hi.astype(np.float32)).astype(np.float32) * np.float32(2048)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def _cut3_bf16(x):  # conv_bf3.hip::cut3
    x = x.astype(np.float32)
    h = (x.view(np.uint32) & 0xFFFF0000).view(np.float32)
    r1 = (x - h).astype(np.float32)
    m = (r1.view(np.uint32) & 0xFFFF0000).view(np.float32)
    l = (r1 - m).astype(np.float32)
    return h.astype(np.float64), m.astype(np.float64), l.astype(np.float64)


@pytest.mark.parametrize("seed", [0, 1])
def test_f16x2_is_as_accurate_as_an_fp32_matrix_instruction(seed):
    a, w = _operands(seed)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    err = lambda x: float(np.linalg.norm(x - ref) / np.linalg.norm(ref))
    exact_products = a.astype(np.float64) * w.astype(np.float64)
    e_f32 = err(_acc_f32([exact_products]))                                   # fp32 MFMA: exact products, fp32 accumulate
    ah, am, al = _cut3_bf16(a)
    wh, wm, wl = _cut3_bf16(w)
    e_bf3 = err(_acc_f32([al * wh, ah * wl, am * wm, am * wh, ah * wm, ah * wh]))   # six chains, ONE accumulator
    sa, sw = _scale(a), _scale(w)
    ah, al = _cut2(a, sa)
    wh, wl = _cut2(w, sw)
    main = _acc_f32([ah * wh]).astype(np.float64)
    cross = _acc_f32([al * wh, ah * wl]).astype(np.float64)                    # its rounding enters scaled by 2^-11
    e_h2 = err((main + cross / 2048.0) / (sa * sw))
    assert e_h2 < 1.25 * e_f32, (e_h2, e_f32)       # measured 3.2e-7 vs 3.0e-7
    assert e_h2 < e_bf3, (e_h2, e_bf3)              # measured 3.2e-7 vs 7.0e-7
    assert e_h2 < 5e-7


def test_f16x2_pieces_reconstruct_to_23_bits():
    rng = np.random.default_rng(3)
    x = (rng.normal(0, 1, 200000) * 10.0 ** rng.uniform(-6, 0, 200000)).astype(np.float32)   # 6 decades below the max
    s = _scale(x)
    hi, lo = _cut2(x, s)
    back = (hi + lo / 2048.0) / s
    rel = np.abs(back - x.astype(np.float64)) / np.abs(x.astype(np.float64))
    big = np.abs(x) > np.abs(x).max() * 2.0 ** -29          # elements whose lo piece is a normal fp16
    assert rel[big].max() <= 2.0 ** -22                     # (2^-23 but for the ties of the fp16 rounding)
    assert np.abs(back - x)[~big].max() <= np.abs(x).max() * 2.0 ** -39 if (~big).any() else True
