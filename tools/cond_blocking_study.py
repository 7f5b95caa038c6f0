#!/usr/bin/env python
"""How large may the explicitly inverted diagonal blocks of a blocked forward substitution be?
(developer study behind csrc/sweep.hip's substitution form; CPU, NumPy, fixtures of tests/golden/.)

sigma^2 = amp - |L^-1 k*|^2 on the conditioning-ladder fixtures (true cond(K) 1e8 .. 8.5e15, each
with a 60-digit mpmath truth), computed in fp64 by
  cho   the oracle (george's cho_solve form)            trsm  one triangular solve (substitution)
  inv   explicit L^-1 product (the inverse-form sweep)
  sub<nb>   blocked substitution, nb x nb diagonal blocks inverted, V_i = D_i^-1 (k*_i - sum_j L_ij V_j)
  fold<nb>  the same with D_i^-1 folded into the rows (V_i = D_i^-1 k*_i - sum_j (D_i^-1 L_ij) V_j)
  hier4     16-row steps whose 16 x 16 diagonal solve is itself `sub4`  (what the kernel does)
Result (median / max relative error of sigma^2 against truth, printed below): every form is in
cho_solve's class up to cond 1e13; at 8.5e15 the explicit inverse is ~200x off, `fold` degrades from
nb = 8, `sub` from nb = 16, and sub4 / hier4 stay within 2-3x of cho_solve.  Hence: 4 x 4 inverses,
applied AFTER the subtraction."""
import os, sys
import numpy as np
from scipy.linalg import solve_triangular
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import george_oracle as go


def build(g):
    D = g["theta"].shape[1]; p = g["p"]
    k = go.Product(go.ConstantKernel(p[1], ndim=D), go.ExpSquaredKernel(np.exp(p[2:]), ndim=D))
    gp = go.GP(kernel=k, fit_mean=True, mean=float(p[0]), white_noise=float(g["white_noise"]), fit_white_noise=False)
    gp.compute(g["theta"])
    return gp


def sub(L, Ks, nb):
    n = L.shape[0]; V = np.zeros_like(Ks)
    for i0 in range(0, n, nb):
        i1 = min(n, i0 + nb)
        Dinv = solve_triangular(L[i0:i1, i0:i1], np.eye(i1 - i0), lower=True)
        V[i0:i1] = Dinv @ (Ks[i0:i1] - L[i0:i1, :i0] @ V[:i0])
    return V


def fold(L, Ks, nb):
    n = L.shape[0]; V = np.zeros_like(Ks)
    for i0 in range(0, n, nb):
        i1 = min(n, i0 + nb)
        Dinv = solve_triangular(L[i0:i1, i0:i1], np.eye(i1 - i0), lower=True)
        V[i0:i1] = Dinv @ Ks[i0:i1] + (-Dinv @ L[i0:i1, :i0]) @ V[:i0]
    return V


def hier4(L, Ks):
    n = L.shape[0]; V = np.zeros_like(Ks)
    for i0 in range(0, n, 16):
        i1 = min(n, i0 + 16)
        V[i0:i1] = sub(L[i0:i1, i0:i1], Ks[i0:i1] - L[i0:i1, :i0] @ V[:i0], 4)
    return V


for name in ["rosen2d_n50_amp_cond1e8", "rosen2d_n50_amp_cond1e11", "rosen2d_n50_amp_cond1e13", "rosen2d_n50_amp_opt_illcond"]:
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    gp = build(g); amp = 2 * np.exp(g["p"][1])
    K = gp.kernel.get_value(g["theta"]); K[np.diag_indices_from(K)] += np.exp(float(g["white_noise"]))
    L = np.linalg.cholesky(K)
    Ks = gp.kernel.get_value(g["theta"], g["cands"])
    vt = g["var_truth"]
    W = solve_triangular(L, np.eye(len(L)), lower=True)
    forms = [("cho", g["var"]), ("trsm", amp - (solve_triangular(L, Ks, lower=True) ** 2).sum(0)),
             ("inv", amp - ((W @ Ks) ** 2).sum(0)), ("hier4", amp - (hier4(L, Ks) ** 2).sum(0))]
    for nb in (2, 4, 8, 16):
        forms.append(("sub%d" % nb, amp - (sub(L, Ks, nb) ** 2).sum(0)))
        forms.append(("fold%d" % nb, amp - (fold(L, Ks, nb) ** 2).sum(0)))
    print("%s: true cond %.2e, estimate (max L_ii / min L_ii)^2 %.2e" % (name, g["cond"], (L.diagonal().max() / L.diagonal().min()) ** 2))
    for k, v in forms:
        e = np.abs(v - vt)
        print("   %-7s rel median %.2e max %.2e | abs/amp max %.2e" % (k, np.median(e / np.abs(vt)), (e / np.abs(vt)).max(), e.max() / amp))
