"""Developer report (run on the GPU box): HIP path vs oracle / golden fixtures,
printing the actual error levels.  Test infrastructure -- imports the oracle."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import george_oracle as go  # noqa: E402
from approxposterior_amd import gp as agp  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def build(mod, g):
    D = g["theta"].shape[1]
    p = g["p"]
    if int(g["fit_amp"]):
        k = mod.Product(mod.ConstantKernel(p[1], ndim=D), mod.ExpSquaredKernel(np.exp(p[2:]), ndim=D))
    else:
        k = mod.ExpSquaredKernel(np.exp(p[1:]), ndim=D)
    gp = mod.GP(kernel=k, fit_mean=True, mean=float(p[0]), white_noise=float(g["white_noise"]),
                fit_white_noise=False)
    gp.compute(g["theta"])
    return gp


def relerr(a, b):
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    fin = np.isfinite(a) & np.isfinite(b)
    same_nonfinite = np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~fin & ~np.isnan(a)], b[~fin & ~np.isnan(b)])
    d = np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), 1e-300)
    return (float(d.max()) if d.size else 0.0), same_nonfinite


def main():
    names = [f[:-4] for f in sorted(os.listdir(GOLDEN)) if f.endswith(".npz")]
    for name in names:
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        gp = build(agp, g)
        y = g["y"]
        ll = gp.log_likelihood(y)
        mu, var = gp.predict(y, g["cands"], return_var=True)
        mu1 = gp.predict(y, g["cands"], return_cov=False, return_var=False)
        bounds = list(zip(g["lo"], g["hi"]))
        print("== %s N=%d D=%d cond=%.3g cond_est=%.3g" % (name, len(y), g["theta"].shape[1], g["cond"], gp.cond_estimate))
        print("   ll rel %.3e  logdet rel %.3e" % (abs(ll - g["ll"]) / abs(g["ll"]), abs(gp.log_determinant - g["logdet"]) / abs(g["logdet"])))
        al = gp._alpha.cpu().numpy()
        print("   alpha rel(max-norm) %.3e" % (np.abs(al - g["alpha"]).max() / np.abs(g["alpha"]).max()))
        print("   mu  max rel %.3e ; mean-only vs sweep %.3e" % (relerr(mu, g["mu"])[0], np.abs(mu - mu1).max()))
        amp = gp._kernel_struct().amp
        print("   var max abs/amp %.3e  max rel %.3e  (min var/amp %.3e)" % (np.abs(var - g["var"]).max() / amp, relerr(var, g["var"])[0], g["var"].min() / amp))
        for kind, key in (("agp", "u_agp"), ("bape", "u_bape"), ("jones", "u_jones")):
            bi, bu, u, m_, v_ = gp.acquire(y, g["cands"], kind, bounds=bounds, return_all=True)
            ref = g[key]
            e, same = relerr(u, ref)
            refm = np.where(np.isnan(ref), np.inf, ref)
            ri = int(np.argmin(refm)) if np.isfinite(refm).any() else -1
            print("   %-5s max rel %.3e nonfinite-match %s argmin %d (ref %d) u %.10g (ref %.10g)" % (kind, e, same, bi, ri, bu, refm[ri] if ri >= 0 else np.nan))
        grad = gp.grad_log_likelihood(y)
        print("   grad rel %s" % np.array2string(np.abs(grad - g["grad"]) / np.maximum(np.abs(g["grad"]), 1e-300), precision=2))

    # larger: C2 / C3 shapes vs oracle on a candidate subsample
    from scipy.optimize import rosen
    for (N, D, M, metric) in ((1024, 2, 4096, 2.0), (4096, 8, 8192, 8.0)):
        rs = np.random.RandomState(0)
        X = rs.uniform(-5, 5, size=(N, D))
        y = np.array([-rosen(x) / 100.0 for x in X])
        cands = np.random.RandomState(1).uniform(-5, 5, size=(M, D))
        t0 = time.time()
        ko = go.ExpSquaredKernel(np.full(D, metric), ndim=D)
        gpo = go.GP(kernel=ko, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
        gpo.compute(X)
        mo, vo = gpo.predict(y, cands[:512], return_var=True)
        llo = gpo.log_likelihood(y)
        t1 = time.time()
        k = agp.ExpSquaredKernel(np.full(D, metric), ndim=D)
        gp = agp.GP(kernel=k, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
        gp.compute(X)
        ll = gp.log_likelihood(y)
        import torch
        torch.cuda.synchronize(); t2 = time.time()
        bi, bu, u, mu, var = gp.acquire(y, cands, "agp", bounds=[(-5, 5)] * D, return_all=True)
        torch.cuda.synchronize(); t3 = time.time()
        bi, bu, u, mu, var = gp.acquire(y, cands, "agp", bounds=[(-5, 5)] * D, return_all=True)
        torch.cuda.synchronize(); t4 = time.time()
        print("== big N=%d D=%d M=%d cond_est %.3g : ll rel %.3e mu rel %.3e var rel %.3e  (oracle %.2fs, fit %.3fs, first sweep %.3fs, second %.3fs)" % (
            N, D, M, gp.cond_estimate, abs(ll - llo) / abs(llo), relerr(mu[:512], mo)[0], relerr(var[:512], vo)[0], t1 - t0, t2 - t1, t3 - t2, t4 - t3))
        uo = -(mo + 0.5 * np.log(2 * np.pi * np.e * vo))
        print("   argmin(first 512) gpu %d oracle %d" % (int(np.argmin(u[:512])), int(np.argmin(uo))))


if __name__ == "__main__":
    main()
