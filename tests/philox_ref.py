"""NumPy restatement of the device candidate generator (csrc/ensemble.hip, box_candidates_kernel) -- test infrastructure:
the GPU test pins the kernel to it, the gloo test's stub GP draws its candidates with it."""
import numpy as np


def philox_box_numpy(m, D, lo, hi, seed, offset):
    """NumPy restatement of csrc/ensemble.hip box_candidates_kernel: Philox4x32-10, counter = (row low, row high, d / 2,
    0x43414e44), key = seed; u = 53-bit uniform in (0, 1); value = fma(span, u, lo) (the product span * u is exact enough to
    compare to 1 ulp: the kernel fuses it)."""
    rows = (np.arange(m, dtype=np.uint64) + np.uint64(offset))
    out = np.empty((m, D))
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    for d in range(0, D, 2):
        c = [rows & mask, rows >> np.uint64(32), np.full(m, d >> 1, dtype=np.uint64), np.full(m, 0x43414E44, dtype=np.uint64)]
        k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
        for _ in range(10):
            p0, p1 = M0 * c[0], M1 * c[2]
            c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask, p1 & mask, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask, p0 & mask]
            k0 = (k0 + np.uint64(0x9E3779B9)) & mask
            k1 = (k1 + np.uint64(0xBB67AE85)) & mask
        u0 = ((c[0] >> np.uint64(5)).astype(np.float64) * 67108864.0 + (c[1] >> np.uint64(6)).astype(np.float64) + 0.5) / 9007199254740992.0
        u1 = ((c[2] >> np.uint64(5)).astype(np.float64) * 67108864.0 + (c[3] >> np.uint64(6)).astype(np.float64) + 0.5) / 9007199254740992.0
        out[:, d] = lo[d] + (hi[d] - lo[d]) * u0
        if d + 1 < D:
            out[:, d + 1] = lo[d + 1] + (hi[d + 1] - lo[d + 1]) * u1
    return out
