// Shared helpers for the libapgp.so HIP sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/apgp.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

void apgp_set_error(const char* fmt, ...);

#define APGP_CHECK_ARG(cond, msg)                                   \
    do {                                                            \
        if (!(cond)) {                                              \
            apgp_set_error("%s: bad argument: %s", __func__, msg);  \
            return -1;                                              \
        }                                                           \
    } while (0)

#define APGP_CHECK_LAUNCH()                                                       \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            apgp_set_error("%s: HIP error: %s", __func__, hipGetErrorString(e__)); \
            return -2;                                                            \
        }                                                                         \
    } while (0)

// Feature dimension padded to a template-friendly width (zero scale => zero
// contribution of the padded coordinates).
static inline int apgp_dpad(int d) { return d <= 2 ? 2 : d <= 4 ? 4 : d <= 8 ? 8 : 16; }
// doubles per row of the packed training stream: scaled x (Dpad) | alpha | 0
static inline int apgp_xs_stride(int d) { return apgp_dpad(d) + 2; }

static inline int64_t apgp_round_up(int64_t n, int64_t b) { return (n + b - 1) / b * b; }

// Scaled-coordinate / amplitude constants shared by every kernel that
// evaluates the squared-exponential kernel, so that K (Gram), K* (sweep) and
// dK (gradient) use bit-identical expressions:
//   k(x,x') = exp(-(sum_d (xs_d - xs'_d)^2 - log_amp)),  xs = x * sqrt(inv_metric/2)
struct KernConst {
    double sc[APGP_MAX_DIM];
    double log_amp;
    double amp;
    double diag_add;
    int ndim;
    int dpad;
};

static inline int apgp_make_kernconst(const apgp_kernel_t* k, KernConst* c) {
    if (!k || k->ndim < 1 || k->ndim > APGP_MAX_DIM) return -1;
    if (!(k->amp > 0.0)) return -1;
    c->ndim = k->ndim;
    c->dpad = apgp_dpad(k->ndim);
    c->amp = k->amp;
    c->log_amp = log(k->amp);
    c->diag_add = k->diag_add;
    for (int d = 0; d < APGP_MAX_DIM; ++d)
        c->sc[d] = d < k->ndim ? sqrt(0.5 * k->inv_metric[d]) : 0.0;
    return 0;
}
