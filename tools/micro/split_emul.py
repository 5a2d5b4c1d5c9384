"""CPU emulation of the two-piece fp16 product of ndcn_amd/csrc/split16.h: what the choice of the power-of-two scales costs or
buys when the operands have outliers (round-4 review, "What's weak" 1).  numpy only; prints max |err| / sum |s w| per case.

  python tools/micro/split_emul.py            # table: scale target 2^0 (rounds 3-4) vs 2^15, global vs per-output-row W scale
"""
import numpy as np


def f16_trunc(x, nearest=False):
    """x (float64) -> the fp16 value (as float64) by round-toward-zero (v_cvt_pkrtz) or to nearest; subnormals honoured"""
    ax = np.abs(x)
    e = np.floor(np.log2(np.where(ax > 0, ax, 1.0)))
    e = np.maximum(e, -14.0)
    q = np.exp2(e - 10.0)
    r = x / q
    r = np.rint(r) if nearest else np.trunc(r)
    out = r * q
    return np.where(ax > 0, np.clip(out, -65504.0, 65504.0), 0.0)


def pow2_scale(maxabs, top):
    """power of two that brings maxabs into [2^(top-1), 2^top)"""
    e = np.floor(np.log2(np.where(maxabs > 0, maxabs, 1.0)))
    return np.exp2(top - 1 - e)


def split_product(S, W, top, w_per_row, s_nearest=False):
    """S [n, k] fp32, W [o, k] fp32 -> S W^T by the three products of split16.h (exact accumulation: isolates the operand error)"""
    S = S.astype(np.float64)
    W = W.astype(np.float64)
    ss = pow2_scale(np.abs(S).max(axis=1, keepdims=True), top)
    ws = pow2_scale(np.abs(W).max(axis=1, keepdims=True), top) if w_per_row else pow2_scale(np.abs(W).max(), top) * np.ones((W.shape[0], 1))
    Sx, Wx = S * ss, W * ws
    s0 = f16_trunc(Sx, s_nearest)
    s1 = f16_trunc(Sx - s0, s_nearest)
    w0 = f16_trunc(Wx, True)
    w1 = f16_trunc(Wx - w0, True)
    acc = s1 @ w0.T + s0 @ w1.T + s0 @ w0.T
    return acc / ss / ws.T


def case(name, S, W):
    S = S.astype(np.float32)
    W = W.astype(np.float32)
    ref = S.astype(np.float64) @ W.astype(np.float64).T
    mag = np.abs(S).astype(np.float64) @ np.abs(W).astype(np.float64).T
    f32 = np.zeros_like(ref, dtype=np.float32)
    for k in range(S.shape[1]):                                     # an fp32 fma chain (one rounding per step, emulated as round(acc + p))
        f32 = (f32.astype(np.float64) + S[:, k:k + 1].astype(np.float64) * W[:, k].astype(np.float64)[None, :]).astype(np.float32)
    row = [np.max(np.abs(f32 - ref) / mag)]
    for top, per_row in ((0, False), (0, True), (15, False), (15, True)):
        got = split_product(S, W, top, per_row)
        row.append(np.max(np.abs(got - ref) / mag))
    print('%-44s fp32 %.1e | top 2^0 global %.1e  per-row %.1e | top 2^15 global %.1e  per-row %.1e' % ((name,) + tuple(row)))


def main():
    rng = np.random.default_rng(0)
    n, H = 512, 256
    S = rng.random((n, H))
    W = (rng.random((H, H)) - 0.5) / 8
    case('U(0,1) x U(+-1/16)', S, W)
    for p in (4, 8, 12, 16):
        W2 = W.copy(); W2[17, 33] *= 2.0 ** p
        case('one weight x 2^%d' % p, S, W2)
    for p in (8, 12):
        W2 = W.copy(); W2[17, :] *= 2.0 ** p
        case('one output row x 2^%d' % p, S, W2)
    W3 = np.exp(rng.normal(size=(H, H)) * np.log(10.0) * 1.0) * np.sign(rng.random((H, H)) - 0.5) * 1e-3
    case('log-normal weights, sigma = 1 decade', W=W3, S=S)
    W4 = 10.0 ** rng.uniform(-6, 0, size=(H, H)) * np.sign(rng.random((H, H)) - 0.5)
    case('log-uniform weights over 6 decades', W=W4, S=S)
    S2 = S.copy(); S2[:, 5] *= 2.0 ** 12
    W5 = W.copy(); W5[:, 5] *= 2.0 ** -12
    case('S channel x 2^12 against a weight column x 2^-12', S2, W5)
    S3 = S.copy(); S3[:, 5] *= 2.0 ** 12
    case('S channel x 2^12, default weights', S3, W)
    S4 = 10.0 ** rng.uniform(-6, 0, size=(n, H))
    case('log-uniform S over 6 decades x log-uniform W', S4, W4)


if __name__ == '__main__':
    main()
