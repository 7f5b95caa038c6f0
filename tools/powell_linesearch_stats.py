"""Where SciPy's Powell spends its evaluations inside a line search, and how often the abscissae that gpUtils._powellAhead
guesses are the ones asked for next (the odds quoted in DESIGN.md section 5d / docs/experiments.md "Round 6").  CPU only: the
oracle GP (oracle/george_oracle.py -- a developer script, like the tests it may import the oracle) at N x D with random start
points; the call sites are read from SciPy's frames exactly as _powellAhead reads them.
Usage: python tools/powell_linesearch_stats.py [N] [D] [seeds]"""
import collections, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import george_oracle as go
from scipy.optimize import minimize, rosen

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
tot = collections.Counter()
for seed in range(seeds):
    rs = np.random.RandomState(seed)
    X = rs.uniform(-5, 5, size=(n, d))
    y = np.array([-rosen(x) / 100.0 for x in X])
    gp = go.GP(kernel=go.ExpSquaredKernel(np.fabs(rs.randn(d)), ndim=d), fit_mean=True, mean=np.median(y), white_noise=-12,
               fit_white_noise=False)
    gp.compute(X)
    seq = []

    def nll(p):
        fr = sys._getframe(2)
        rec = None
        if fr.f_code.co_name == "myfunc":
            site = fr.f_back
            if site.f_code.co_name == "optimize":
                L = site.f_locals
                rec = ("opt", L["iter"], float(L["x"]), float(L["u"]), float(L["tol1"]))
            else:
                rec = (site.f_code.co_name, float(fr.f_locals["alpha"]))
        if (np.fabs(p)[1:] > 20).any():
            v = np.inf
        else:
            gp.set_parameter_vector(p)
            ll = gp.log_likelihood(y, quiet=True)
            v = -ll if np.isfinite(ll) else np.inf
        seq.append((rec, v))
        return v
    with np.errstate(all="ignore"):
        minimize(nll, [np.median(y)] + [rs.randn() for _ in range(d)], method="powell")
    searches, cur = [], None
    for rec, v in seq:
        if rec is not None and rec[0] == "bracket" and rec[1] == 0.0:
            cur = []
            searches.append(cur)
        if cur is not None and rec is not None:
            cur.append((rec, v))
    for ls in searches:
        tot["line searches"] += 1
        tot["evaluations (without f(0))"] += len(ls) - 1
        nbr = sum(1 for rec, _ in ls if rec[0] == "bracket")
        brent = [(rec, v) for rec, v in ls if rec[0] == "opt"]
        tot["Brent evaluations"] += len(brent)
        if len(ls) < 3:
            continue
        swapped = ls[2][0][1] < 0            # third bracket point -1.618034: f(0) < f(1)
        tot["f(0) < f(1) (third point -1.618)" if swapped else "f(0) > f(1) (third point 2.618)"] += 1
        if nbr == 3:
            tot["bracket closed by the third value | %s" % ("-1.618" if swapped else "2.618")] += 1
            if brent:
                fx = ls[0][1] if swapped else ls[1][1]
                tot["Brent's first step: no improvement | %s" % ("-1.618" if swapped else "2.618")] += int(brent[0][1] > fx)
        for k, (rec, v) in enumerate(brent):
            _, it, x, u, tol1 = rec
            if u == x + tol1 or u == x - tol1:
                tot["tolerance steps"] += 1
                if k + 1 < len(brent):
                    nx = brent[k + 1][0]
                    tot["tolerance step followed by its mirror image"] += int(nx[3] == (x - tol1 if u == x + tol1 else x + tol1))
                else:
                    tot["tolerance step = last evaluation of the search"] += 1
for k in sorted(tot):
    print("%-60s %6d" % (k, tot[k]))
