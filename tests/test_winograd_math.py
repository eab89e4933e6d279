"""CPU checks of the Winograd transforms the HIP kernels hard-code (csrc/misc_kernels.hip): the F(4x4,3x3) matrices are
parsed from the source and, with the F(2x2,3x3) ones, must reproduce a direct 3x3 correlation and its adjoints."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, 'e-osvos_amd', 'csrc', 'misc_kernels.hip')).read()


def _parse(name, rows, cols):
    m = re.search(r'__device__ constexpr float %s\[%d\]\[%d\] = (\{.*?\});' % (name, rows, cols), SRC, re.S)
    assert m, name
    txt = m.group(1).replace('f', '').replace('{', '[').replace('}', ']')
    return np.array(eval(txt), dtype=np.float64).reshape(rows, cols)     # literals like 1. / 6 evaluate as written


F4 = dict(BT=_parse('BT', 6, 6), G=_parse('G', 6, 3), AT=_parse('AT', 4, 6))
F2 = dict(BT=np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float),
          G=np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float),
          AT=np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float))


def _corr(d, g, m):
    return np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(m)] for i in range(m)])


def test_winograd_forward_identity():
    rng = np.random.default_rng(0)
    for F, m in ((F2, 2), (F4, 4)):
        for _ in range(5):
            d, g = rng.standard_normal((m + 2, m + 2)), rng.standard_normal((3, 3))
            U, V = F['G'] @ g @ F['G'].T, F['BT'] @ d @ F['BT'].T
            Y = F['AT'] @ (U * V) @ F['AT'].T
            np.testing.assert_allclose(Y, _corr(d, g, m), rtol=1e-10, atol=1e-10)


def test_winograd_adjoints():
    """dM = A dY A^T, dd = B dV B^T, dg = G^T dU G are the gradients of Y w.r.t. M, d and g (what wino*_grad,
    wino*_dgrad_output and wino*_wgrad_finish compute)."""
    rng = np.random.default_rng(1)
    for F, m in ((F2, 2), (F4, 4)):
        d, g, dY = rng.standard_normal((m + 2, m + 2)), rng.standard_normal((3, 3)), rng.standard_normal((m, m))
        A, B = F['AT'].T, F['BT'].T
        U, V = F['G'] @ g @ F['G'].T, F['BT'] @ d @ B
        dM = A @ dY @ A.T
        dd = B @ (dM * U) @ B.T              # d/d(input patch) of <dY, Y>
        dg = F['G'].T @ (dM * V) @ F['G']    # d/d(filter)
        eps = 1e-6
        for (arr, grad) in ((d, dd), (g, dg)):
            num = np.zeros_like(arr)
            for idx in np.ndindex(arr.shape):
                a2 = arr.copy(); a2[idx] += eps
                y2 = _corr(a2, g, m) if arr is d else _corr(d, a2, m)
                num[idx] = ((y2 - _corr(d, g, m)) * dY).sum() / eps
            np.testing.assert_allclose(grad, num, rtol=1e-4, atol=1e-5)
