"""INTEGRATION.md section 2 is executable documentation: the fenced Python block a maintainer would paste to bind
``libapgp.so`` with nothing but ctypes is extracted and RUN here (the block once declared
``inv_metric`` with 16 entries while ``apgp_kernel_t`` had 32, and nothing noticed).

CPU: the block's ``Kernel`` structure has the size and field offsets of the package's own binding and of
``include/apgp.h``.  MI355X: its ``compute`` + ``log_likelihood`` reproduce the oracle's log-likelihood.
Reference call sites the block stands for: /root/reference/approxposterior/gpUtils.py:74-78,178."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _section2_block():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. C-ABI level"):text.index("## 3. Build / run")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 1, "section 2 holds ONE python block: the binding"
    return blocks[0]


def _run_block():
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)                      # the block uses repository-relative paths, as its text says
    try:
        exec(compile(_section2_block(), "INTEGRATION.md#2", "exec"), ns)
    finally:
        os.chdir(cwd)
    return ns


def test_documented_struct_matches_the_header_and_the_package_binding():
    from approxposterior_amd import _lib
    ns = _run_block()
    K, mine = ns["Kernel"], _lib.KernelStruct
    assert ctypes.sizeof(K) == ctypes.sizeof(mine)
    for name, _ in mine._fields_:
        assert getattr(K, name).offset == getattr(mine, name).offset, name
        assert getattr(K, name).size == getattr(mine, name).size, name
    header = open(os.path.join(ROOT, "include", "apgp.h")).read()
    max_dim = int(re.search(r"#define\s+APGP_MAX_DIM\s+(\d+)", header).group(1))
    assert ns["MAX_DIM"] == max_dim == _lib.MAX_DIM
    assert K.lin_coef.offset == 8 + 8 + 8 + 8 * max_dim       # two int32, amp, diag_add, inv_metric[MAX_DIM]


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,amp", [(300, 2, 1.0), (1000, 8, 1.0), (257, 5, 3.7)])
def test_documented_binding_reproduces_the_oracle_log_likelihood(n, d, amp):
    import george_oracle as go
    ns = _run_block()
    rs = np.random.RandomState(n)
    X = rs.uniform(-5, 5, size=(n, d))
    y = -np.sum(100.0 * (X[:, 1:] - X[:, :-1] ** 2) ** 2 + (1 - X[:, :-1]) ** 2, axis=1) / 100.0
    log_M = np.log(rs.uniform(4.0, 12.0, size=d))
    mean = float(np.median(y))
    k, Xd, L, out5 = ns["compute"](X, log_M, white_noise=-12.0, amp=amp)
    rec = out5.cpu().numpy()
    assert rec[4] == 0                                       # LAPACK info: positive definite
    ll = ns["log_likelihood"](L, y, mean, float(rec[0]))
    ko = go.ExpSquaredKernel(np.exp(log_M), ndim=d)
    if amp != 1.0:
        ko = amp * ko
    o = go.GP(kernel=ko, fit_mean=True, mean=mean, white_noise=-12.0, fit_white_noise=False)
    o.compute(X)
    want = o.log_likelihood(y)
    assert abs(ll - want) <= 1e-10 * abs(want), (ll, want)
    assert abs(rec[0] - o.log_determinant) <= 1e-11 * abs(o.log_determinant)
