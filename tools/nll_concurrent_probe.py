"""How much do several _nll evaluations cost when they run SIDE BY SIDE on one MI355X?  (VERDICT round 5, weak 3:
an evaluation occupies 34-67 CUs and is latency-bound.)  T host threads, one torch stream and one GP object each,
every thread runs R evaluations (set_parameter_vector + log_likelihood) back to back; ctypes releases the GIL inside
apgp_nll_eval.  Prints ms per evaluation seen by one thread and the aggregate evaluations per ms."""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from approxposterior_amd import gp as agp
from approxposterior_amd import _lib
from bench import synthetic_c3


def main():
    lib = _lib.load()
    for n in (512, 832, 1152, 2048):
        X, y = synthetic_c3(n, 8)
        for T in (1, 2, 3, 4, 5, 6, 8):
            streams = [torch.cuda.Stream() for _ in range(T)]
            gps = []
            for s in streams:
                with torch.cuda.stream(s):
                    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(8, 8.0), ndim=8), fit_mean=True, mean=np.median(y),
                               white_noise=-12, fit_white_noise=False)
                    g.compute(X)
                    gps.append(g)
            R = 60
            fb0 = lib.apgp_potrf_fallbacks()
            bar = threading.Barrier(T + 1)
            out = [None] * T

            def work(k):
                g, s = gps[k], streams[k]
                p = g.get_parameter_vector()
                with torch.cuda.stream(s):
                    for i in range(5):
                        g.set_parameter_vector(p + 1e-3 * i); g.log_likelihood(y, quiet=True)
                    bar.wait()
                    t0 = time.time()
                    for i in range(R):
                        g.set_parameter_vector(p + 1e-3 * (i % 3)); g.log_likelihood(y, quiet=True)
                    out[k] = (time.time() - t0) / R * 1e3
            th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
            for t in th:
                t.start()
            bar.wait()
            t0 = time.time()
            for t in th:
                t.join()
            wall = time.time() - t0
            print("n=%5d threads=%d  ms/eval per thread %.3f (max %.3f)  aggregate %.2f evals/ms  fallbacks %d"
                  % (n, T, float(np.mean(out)), max(out), T * R / (wall * 1e3), lib.apgp_potrf_fallbacks() - fb0), flush=True)
            lib.apgp_potrf_mode(0)      # ends any back-off between the configurations


if __name__ == "__main__":
    main()
