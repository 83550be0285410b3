/*
 * ed_oracle.c -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product (elasticdeform_amd/) may import, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Parity status: PINNED.  tests/test_oracle.py compares this restatement with the
 * real reference (compiled from /root/reference into oracle/_ref by oracle/Makefile) and with the
 * golden vectors under tests/golden/ that tests/golden/gen_golden.py produced by importing the
 * reference; float64 / float32 / integer results are bit-identical (same IEEE double operations
 * in the same order; both are built without FMA contraction).
 *
 * Plain C99, no Python, no NumPy.  Arrays come in as `edhip_array` descriptors (include/edhip.h)
 * whose `data` are HOST pointers here.  The code is written from the behavioural spec in
 * SURVEY.md Appendix A; each function cites the reference lines whose arithmetic it restates.
 * The loop structure is our own: a flat odometer over the deformed output axes instead of
 * NI_Iterator, explicit tap-index lists instead of byte-offset tables.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "edhip.h"

#define MAXAX 16 /* the oracle accepts more deformed axes than the GPU build does */

static void set_err(char* err, size_t n, const char* msg)
{
    if (err && n) {
        strncpy(err, msg, n - 1);
        err[n - 1] = 0;
    }
}

/* ---- element access: (double)*(T*)p, deform.c:282-285 -------------------------------------- */
static int load_elem(const char* p, int dt, double* v)
{
    switch (dt) {
    case EDHIP_BOOL: *v = (double)*(const unsigned char*)p; return 1;
    case EDHIP_U8:   *v = (double)*(const uint8_t*)p; return 1;
    case EDHIP_I8:   *v = (double)*(const int8_t*)p; return 1;
    case EDHIP_U16:  *v = (double)*(const uint16_t*)p; return 1;
    case EDHIP_I16:  *v = (double)*(const int16_t*)p; return 1;
    case EDHIP_U32:  *v = (double)*(const uint32_t*)p; return 1;
    case EDHIP_I32:  *v = (double)*(const int32_t*)p; return 1;
    case EDHIP_U64:  *v = (double)*(const uint64_t*)p; return 1;
    case EDHIP_I64:  *v = (double)*(const int64_t*)p; return 1;
    case EDHIP_F32:  *v = (double)*(const float*)p; return 1;
    case EDHIP_F64:  *v = *(const double*)p; return 1;
    default: return 0;
    }
}

/* forward store: float -> C cast; signed -> round half away from zero, clamp; unsigned -> +0.5
 * for positives else 0, clamp; bool -> C cast.  deform.c:287-306,906-919 */
#define ST_UINT(T, MAXV)                                   \
    do {                                                   \
        t = t > 0 ? t + 0.5 : 0;                           \
        t = t > (double)(MAXV) ? (double)(MAXV) : t;       \
        t = t < 0 ? 0 : t;                                 \
        *(T*)p = (T)t;                                     \
    } while (0)
#define ST_INT(T, MINV, MAXV)                              \
    do {                                                   \
        t = t > 0 ? t + 0.5 : t - 0.5;                     \
        t = t > (double)(MAXV) ? (double)(MAXV) : t;       \
        t = t < (double)(MINV) ? (double)(MINV) : t;       \
        *(T*)p = (T)t;                                     \
    } while (0)

static int store_forward(char* p, int dt, double t)
{
    switch (dt) {
    case EDHIP_BOOL: *(unsigned char*)p = (unsigned char)t; return 1;
    case EDHIP_U8:   ST_UINT(uint8_t, UINT8_MAX); return 1;
    case EDHIP_U16:  ST_UINT(uint16_t, UINT16_MAX); return 1;
    case EDHIP_U32:  ST_UINT(uint32_t, UINT32_MAX); return 1;
    case EDHIP_U64:  ST_UINT(uint64_t, UINT64_MAX); return 1;
    case EDHIP_I8:   ST_INT(int8_t, INT8_MIN, INT8_MAX); return 1;
    case EDHIP_I16:  ST_INT(int16_t, INT16_MIN, INT16_MAX); return 1;
    case EDHIP_I32:  ST_INT(int32_t, INT32_MIN, INT32_MAX); return 1;
    case EDHIP_I64:  ST_INT(int64_t, INT64_MIN, INT64_MAX); return 1;
    case EDHIP_F32:  *(float*)p = (float)t; return 1;
    case EDHIP_F64:  *(double*)p = t; return 1;
    default: return 0;
    }
}

/* gradient accumulate: *(T*)p += (T)t for every dtype, deform.c:309-312,974-987 */
static int accumulate(char* p, int dt, double t)
{
    switch (dt) {
    case EDHIP_BOOL: *(unsigned char*)p += (unsigned char)t; return 1;
    case EDHIP_U8:   *(uint8_t*)p += (uint8_t)t; return 1;
    case EDHIP_I8:   *(int8_t*)p += (int8_t)t; return 1;
    case EDHIP_U16:  *(uint16_t*)p += (uint16_t)t; return 1;
    case EDHIP_I16:  *(int16_t*)p += (int16_t)t; return 1;
    case EDHIP_U32:  *(uint32_t*)p += (uint32_t)t; return 1;
    case EDHIP_I32:  *(int32_t*)p += (int32_t)t; return 1;
    case EDHIP_U64:  *(uint64_t*)p += (uint64_t)t; return 1;
    case EDHIP_I64:  *(int64_t*)p += (int64_t)t; return 1;
    case EDHIP_F32:  *(float*)p += (float)t; return 1;
    case EDHIP_F64:  *(double*)p += t; return 1;
    default: return 0;
    }
}

/* line buffer -> array store of the spline filters: plain C cast for every dtype
 * (from_nd_image.c:422-431; SciPy's ni_support.c does the same) */
static int store_cast(char* p, int dt, double t)
{
    switch (dt) {
    case EDHIP_BOOL: *(unsigned char*)p = (unsigned char)t; return 1;
    case EDHIP_U8:   *(uint8_t*)p = (uint8_t)t; return 1;
    case EDHIP_I8:   *(int8_t*)p = (int8_t)t; return 1;
    case EDHIP_U16:  *(uint16_t*)p = (uint16_t)t; return 1;
    case EDHIP_I16:  *(int16_t*)p = (int16_t)t; return 1;
    case EDHIP_U32:  *(uint32_t*)p = (uint32_t)t; return 1;
    case EDHIP_I32:  *(int32_t*)p = (int32_t)t; return 1;
    case EDHIP_U64:  *(uint64_t*)p = (uint64_t)t; return 1;
    case EDHIP_I64:  *(int64_t*)p = (int64_t)t; return 1;
    case EDHIP_F32:  *(float*)p = (float)t; return 1;
    case EDHIP_F64:  *(double*)p = t; return 1;
    default: return 0;
    }
}

/* ---- boundary map of a real coordinate (legacy SciPy <= 1.5 semantics), deform.c:47-128 ---- */
static double map_coordinate(double c, int64_t len, int mode)
{
    if (c < 0) {
        switch (mode) {
        case EDHIP_MODE_MIRROR:
            if (len <= 1) {
                c = 0;
            } else {
                int64_t period = 2 * len - 2;
                c = period * (int64_t)(-c / period) + c;
                c = c <= 1 - len ? c + period : -c;
            }
            break;
        case EDHIP_MODE_REFLECT:
            if (len <= 1) {
                c = 0;
            } else {
                int64_t period = 2 * len;
                if (c < -period)
                    c = period * (int64_t)(-c / period) + c;
                c = c < -len ? c + period : -c - 1;
            }
            break;
        case EDHIP_MODE_WRAP:
            if (len <= 1) {
                c = 0;
            } else {
                int64_t period = len - 1;
                c += period * ((int64_t)(-c / period) + 1);
            }
            break;
        case EDHIP_MODE_NEAREST:
            c = 0;
            break;
        case EDHIP_MODE_CONSTANT:
            c = -1;
            break;
        }
    } else if (c > len - 1) {
        switch (mode) {
        case EDHIP_MODE_MIRROR:
            if (len <= 1) {
                c = 0;
            } else {
                int64_t period = 2 * len - 2;
                c -= period * (int64_t)(c / period);
                if (c >= len)
                    c = period - c;
            }
            break;
        case EDHIP_MODE_REFLECT:
            if (len <= 1) {
                c = 0;
            } else {
                int64_t period = 2 * len;
                c -= period * (int64_t)(c / period);
                if (c >= len)
                    c = period - c - 1;
            }
            break;
        case EDHIP_MODE_WRAP:
            if (len <= 1) {
                c = 0;
            } else {
                int64_t period = len - 1;
                c -= period * (int64_t)(c / period);
            }
            break;
        case EDHIP_MODE_NEAREST:
            c = len - 1;
            break;
        case EDHIP_MODE_CONSTANT:
            c = -1;
            break;
        }
    }
    return c;
}

/* mirror map of an integer tap index onto [0, len), used for every mode once a filter window
 * sticks out (deform.c:668-683 for the displacement grid, :795-810 for the inputs).  The
 * reference divides in npy_intp and narrows the quotient to int; identical for any real size. */
static int64_t mirror_index(int64_t idx, int64_t len)
{
    if (len <= 1)
        return 0;
    int64_t period = 2 * len - 2;
    if (idx < 0) {
        idx = period * (int)(-idx / period) + idx;
        idx = idx <= 1 - len ? idx + period : -idx;
    } else if (idx >= len) {
        idx -= period * (int)(idx / period);
        if (idx >= len)
            idx = period - idx;
    }
    return idx;
}

/* B-spline basis weights at real position x, deform.c:160-268.  Order 0: nothing (the
 * reference returns "unsupported" and never reads the weights, deform.c:897,959). */
static void spline_weights(double x, int order, double* w)
{
    double y, z, t;
    int i;
    x -= floor(order & 1 ? x : x + 0.5);
    y = x;
    z = 1.0 - x;
    switch (order) {
    case 1:
        w[0] = 1.0 - x;
        break;
    case 2:
        w[1] = 0.75 - x * x;
        y = 0.5 - x;
        w[0] = 0.5 * y * y;
        break;
    case 3:
        w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
        w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
        w[0] = z * z * z / 6.0;
        break;
    case 4:
        t = x * x;
        w[2] = t * (t * 0.25 - 0.625) + 115.0 / 192.0;
        y = 1.0 + x;
        w[1] = y * (y * (y * (5.0 - y) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
        w[3] = z * (z * (z * (5.0 - z) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
        y = 0.5 - x;
        t = y * y;
        w[0] = t * t / 24.0;
        break;
    case 5:
        t = y * y;
        w[2] = t * (t * (0.25 - y / 12.0) - 0.5) + 0.55;
        t = z * z;
        w[3] = t * (t * (0.25 - z / 12.0) - 0.5) + 0.55;
        y += 1.0;
        w[1] = y * (y * (y * (y * (y / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
        z += 1.0;
        w[4] = z * (z * (z * (z * (z / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
        y = 1.0 - x;
        t = y * y;
        w[0] = y * t * t / 120.0;
        break;
    default:
        return;
    }
    w[order] = 1.0;
    for (i = 0; i < order; ++i)
        w[order] -= w[i];
}

/* start of the (order+1)-tap window for coordinate c, deform.c:657-661,784-788 */
static int64_t window_start(double c, int order)
{
    if (order & 1)
        return (int64_t)floor(c) - order / 2;
    return (int64_t)floor(c + 0.5) - order / 2;
}

/* fill idx[0..order] with the tap indices of the window on an axis of length len: consecutive
 * if it fits, mirror-mapped otherwise (deform.c:664-690,791-817) */
static void window_taps(int64_t start, int order, int64_t len, int64_t* idx)
{
    int l;
    if (start < 0 || start + order >= len) {
        for (l = 0; l <= order; ++l)
            idx[l] = mirror_index(start + l, len);
    } else {
        for (l = 0; l <= order; ++l)
            idx[l] = start + l;
    }
}

typedef struct {
    int nstep;                        /* number of non-deformed axes of this input */
    int64_t total;                    /* product of their extents (taken from the input) */
    int64_t isz[EDHIP_MAX_DIMS], osz[EDHIP_MAX_DIMS];
    int64_t istr[EDHIP_MAX_DIMS], ostr[EDHIP_MAX_DIMS];
} step_axes;

/*
 * DeformGrid, deform.c:340-1043: forward gather or gradient scatter-add.
 * Returns EDHIP_OK or an error code with a message.
 */
int edo_deform(int gradient, int ninputs, const edhip_array* inputs,
               const edhip_array* displacement, const int64_t* output_offset,
               const edhip_array* outputs, int naxis, const int32_t* axis,
               const int32_t* orders, const int32_t* modes, const double* cvals,
               const double* affine, char* err, size_t errlen)
{
    int64_t I[MAXAX], O[MAXAX], off[MAXAX], ncp[MAXAX], o[MAXAX];
    int64_t dtap[MAXAX][4];
    int64_t (*itap)[6] = NULL;        /* [naxis][order+1] for the current input */
    double (*iw)[6] = NULL;
    double** dw = NULL;               /* dw[k] -> O[k]*4 displacement weights, deform.c:639-647 */
    step_axes* steps = NULL;
    double displ[MAXAX];
    int64_t size = 1, kk;
    int k, h, l, ii, status = EDHIP_OK;
    const int dorder = 3;             /* deform.c:375 */

    if (naxis < 1 || naxis > MAXAX || ninputs < 1) {
        set_err(err, errlen, "invalid number of inputs/axes");
        return EDHIP_ERR_INVALID;
    }
    for (k = 0; k < naxis; ++k) {
        I[k] = inputs[0].shape[axis[k]];          /* deform.c:383 */
        O[k] = outputs[0].shape[axis[k]];         /* deform.c:384 */
        size *= O[k];
        off[k] = output_offset ? output_offset[k] : 0;  /* deform.c:439-446 */
        ncp[k] = displacement->shape[k + 1];      /* deform.c:449-451 */
    }

    itap = malloc(sizeof(*itap) * naxis);
    iw = malloc(sizeof(*iw) * naxis);
    dw = calloc(naxis, sizeof(*dw));
    steps = calloc(ninputs, sizeof(*steps));
    if (!itap || !iw || !dw || !steps) {
        status = EDHIP_ERR_MEMORY;
        goto done;
    }

    /* displacement weights per axis per output index, deform.c:639-647 */
    for (k = 0; k < naxis; ++k) {
        int64_t j;
        dw[k] = malloc(sizeof(double) * 4 * (O[k] > 0 ? O[k] : 1));
        if (!dw[k]) {
            status = EDHIP_ERR_MEMORY;
            goto done;
        }
        for (j = 0; j < O[k]; ++j) {
            double cp = (double)(ncp[k] - 1) * (double)(j + off[k]) / (double)(I[k] - 1);
            spline_weights(cp, dorder, dw[k] + 4 * j);
        }
    }

    /* non-deformed ("step") axes of every input, deform.c:405-436 */
    for (ii = 0; ii < ninputs; ++ii) {
        step_axes* s = &steps[ii];
        int d;
        s->total = 1;
        for (d = 0; d < inputs[ii].ndim; ++d) {
            int deformed = 0;
            for (k = 0; k < naxis; ++k)
                if (axis[ii * naxis + k] == d)
                    deformed = 1;
            if (!deformed) {
                s->isz[s->nstep] = inputs[ii].shape[d];
                s->osz[s->nstep] = outputs[ii].shape[d];
                s->istr[s->nstep] = inputs[ii].stride_bytes[d];
                s->ostr[s->nstep] = outputs[ii].stride_bytes[d];
                s->total *= inputs[ii].shape[d];
                s->nstep++;
            }
        }
    }

    for (k = 0; k < naxis; ++k)
        o[k] = 0;

    for (kk = 0; kk < size; ++kk) {
        /* ---- displacement at this output voxel: cubic B-spline of the grid, deform.c:650-758 */
        for (k = 0; k < naxis; ++k) {
            double cp = (double)(ncp[k] - 1) * (double)(o[k] + off[k]) / (double)(I[k] - 1);
            window_taps(window_start(cp, dorder), dorder, ncp[k], dtap[k]);
        }
        for (h = 0; h < naxis; ++h) {
            int t4[MAXAX];
            double acc = 0.0;
            for (k = 0; k < naxis; ++k)
                t4[k] = 0;
            for (;;) {
                const char* p = (const char*)displacement->data + displacement->stride_bytes[0] * h;
                double coeff;
                for (k = 0; k < naxis; ++k)
                    p += displacement->stride_bytes[k + 1] * dtap[k][t4[k]];
                if (!load_elem(p, displacement->dtype, &coeff)) {
                    set_err(err, errlen, "data type not supported");
                    status = EDHIP_ERR_DTYPE;
                    goto done;
                }
                for (k = 0; k < naxis; ++k)
                    coeff *= dw[k][4 * o[k] + t4[k]];
                acc += coeff;
                /* next tap, last axis fastest (deform.c:623-636) */
                for (k = naxis - 1; k >= 0; --k) {
                    if (t4[k] < dorder) {
                        t4[k]++;
                        break;
                    }
                    t4[k] = 0;
                }
                if (k < 0)
                    break;
            }
            displ[h] = acc;
        }

        /* ---- every input shares o, off, displ, affine; own order/mode/cval/axes, deform.c:762 */
        for (ii = 0; ii < ninputs; ++ii) {
            const edhip_array* in = &inputs[ii];
            const edhip_array* out = &outputs[ii];
            const int32_t* ax = axis + ii * naxis;
            const int order = orders[ii];
            const step_axes* s = &steps[ii];
            int constant = 0;
            int64_t obase = 0, ss;

            for (h = 0; h < naxis; ++h) {
                double cc;
                if (affine) {                                   /* deform.c:771-776 */
                    cc = 0.0;
                    for (l = 0; l < naxis; ++l)
                        cc += affine[h * (naxis + 1) + l] * (double)o[l];
                    cc += affine[h * (naxis + 1) + naxis];
                } else {
                    cc = (double)o[h];
                }
                cc = map_coordinate(cc + off[h] + displ[h], I[h], modes[ii]);  /* deform.c:781 */
                if (cc > -1.0) {
                    window_taps(window_start(cc, order), order, I[h], itap[h]);
                    spline_weights(cc, order, iw[h]);
                } else {
                    constant = 1;                               /* deform.c:819-822 */
                    break;
                }
            }
            for (k = 0; k < naxis; ++k)
                obase += out->stride_bytes[ax[k]] * o[k];

            for (ss = 0; ss < s->total; ++ss) {                 /* deform.c:828-838 */
                int64_t ioff = 0, ooff = 0, si = ss, so = ss;
                int tt[MAXAX];
                for (l = 0; l < s->nstep; ++l) {
                    ioff += s->istr[l] * (si % s->isz[l]);
                    si /= s->isz[l];
                    ooff += s->ostr[l] * (so % s->osz[l]);
                    so /= s->osz[l];
                }
                char* po = (char*)out->data + obase + ooff;

                if (!gradient) {
                    double t = 0.0;
                    if (!constant) {                            /* deform.c:843-901 */
                        for (k = 0; k < naxis; ++k)
                            tt[k] = 0;
                        for (;;) {
                            const char* p = (const char*)in->data + ioff;
                            double coeff;
                            for (k = 0; k < naxis; ++k)
                                p += in->stride_bytes[ax[k]] * itap[k][tt[k]];
                            if (!load_elem(p, in->dtype, &coeff)) {
                                set_err(err, errlen, "data type not supported");
                                status = EDHIP_ERR_DTYPE;
                                goto done;
                            }
                            if (order > 0)
                                for (k = 0; k < naxis; ++k)
                                    coeff *= iw[k][tt[k]];
                            t += coeff;
                            for (k = naxis - 1; k >= 0; --k) {
                                if (tt[k] < order) {
                                    tt[k]++;
                                    break;
                                }
                                tt[k] = 0;
                            }
                            if (k < 0)
                                break;
                        }
                    } else {
                        t = cvals[ii];                          /* deform.c:903 */
                    }
                    if (!store_forward(po, out->dtype, t)) {
                        set_err(err, errlen, "data type not supported");
                        status = EDHIP_ERR_DTYPE;
                        goto done;
                    }
                } else if (!constant) {                         /* deform.c:926-996 */
                    double grad;
                    if (!load_elem(po, out->dtype, &grad)) {
                        set_err(err, errlen, "data type not supported");
                        status = EDHIP_ERR_DTYPE;
                        goto done;
                    }
                    for (k = 0; k < naxis; ++k)
                        tt[k] = 0;
                    for (;;) {
                        char* p = (char*)in->data + ioff;
                        double coeff = grad;
                        if (order > 0)
                            for (k = 0; k < naxis; ++k)
                                coeff *= iw[k][tt[k]];
                        for (k = 0; k < naxis; ++k)
                            p += in->stride_bytes[ax[k]] * itap[k][tt[k]];
                        if (!accumulate(p, in->dtype, coeff)) {
                            set_err(err, errlen, "data type not supported");
                            status = EDHIP_ERR_DTYPE;
                            goto done;
                        }
                        for (k = naxis - 1; k >= 0; --k) {
                            if (tt[k] < order) {
                                tt[k]++;
                                break;
                            }
                            tt[k] = 0;
                        }
                        if (k < 0)
                            break;
                    }
                }
            }
        }

        /* next output voxel, last deformed axis fastest (from_scipy.h:67-79) */
        for (k = naxis - 1; k >= 0; --k) {
            if (++o[k] < O[k])
                break;
            o[k] = 0;
        }
    }

done:
    if (dw)
        for (k = 0; k < naxis; ++k)
            free(dw[k]);
    free(dw);
    free(itap);
    free(iw);
    free(steps);
    if (status == EDHIP_ERR_MEMORY)
        set_err(err, errlen, "out of memory");
    return status;
}

/* ---- spline prefilter ----------------------------------------------------------------------- */

/* poles and overall gain.
 * transpose: computed with sqrt() exactly as deform.c:1063-1088 does.
 * forward: SciPy (>= 1.6; pinned here against 1.15.3, the version on the image) stores the same
 * poles as correctly rounded decimal literals, which differ from the sqrt() expressions by an
 * ulp or so (up to 3e-13 relative for the second pole of orders 4/5, where the expression
 * cancels).  With the literals below the restatement is bit-identical to
 * scipy.ndimage.spline_filter1d (tests/test_oracle.py). */
static int spline_poles(int order, int transpose, double* pole, double* gain)
{
    int n = 0, h;
    if (transpose) {
        switch (order) {
        case 2: n = 1; pole[0] = sqrt(8.0) - 3.0; break;
        case 3: n = 1; pole[0] = sqrt(3.0) - 2.0; break;
        case 4:
            n = 2;
            pole[0] = sqrt(664.0 - sqrt(438976.0)) + sqrt(304.0) - 19.0;
            pole[1] = sqrt(664.0 + sqrt(438976.0)) - sqrt(304.0) - 19.0;
            break;
        case 5:
            n = 2;
            pole[0] = sqrt(67.5 - sqrt(4436.25)) + sqrt(26.25) - 6.5;
            pole[1] = sqrt(67.5 + sqrt(4436.25)) - sqrt(26.25) - 6.5;
            break;
        default: break;
        }
    } else {
        switch (order) {
        case 2: n = 1; pole[0] = -0.171572875253809902396622551580603843; break;
        case 3: n = 1; pole[0] = -0.267949192431122706472553658494127633; break;
        case 4:
            n = 2;
            pole[0] = -0.361341225900220177092212841325675255;
            pole[1] = -0.013725429297339121360331226939128204;
            break;
        case 5:
            n = 2;
            pole[0] = -0.430575347099973791851434783493520110;
            pole[1] = -0.043096288203264653822712376822550182;
            break;
        default: break;
        }
    }
    *gain = 1.0;
    for (h = 0; h < n; ++h)
        *gain *= (1.0 - pole[h]) * (1.0 - 1.0 / pole[h]);
    return n;
}

/* forward prefilter of one line, mirror boundary: SciPy ndimage.spline_filter1d as restated in
 * SURVEY.md Appendix A step 12 (third-party arithmetic; call sites deform_grid.py:160,168,271).
 * gain first, then per pole: exact mirror initialisation, causal recursion, anti-causal
 * initialisation, anti-causal recursion. */
static void prefilter_line(double* c, int64_t n, int npoles, const double* pole, double gain)
{
    int64_t i;
    int h;
    if (n < 2)
        return;
    for (i = 0; i < n; ++i)
        c[i] *= gain;
    for (h = 0; h < npoles; ++h) {
        const double z = pole[h];
        const double zn1 = pow(z, (double)(n - 1));
        double zi = z;
        c[0] = c[0] + zn1 * c[n - 1];
        for (i = 1; i < n - 1; ++i) {
            c[0] += zi * (c[i] + zn1 * c[n - 1 - i]);
            zi *= z;
        }
        c[0] /= 1 - zn1 * zn1;
        for (i = 1; i < n; ++i)
            c[i] += z * c[i - 1];
        c[n - 1] = (z * c[n - 2] + c[n - 1]) * z / (z * z - 1);
        for (i = n - 2; i >= 0; --i)
            c[i] = z * (c[i + 1] - c[i]);
    }
}

/* transpose of the prefilter on one line: NI_SplineFilter1DGrad's inner loop, deform.c:1116-1156 */
static void prefilter_transpose_line(double* ln, int64_t len, int npoles, const double* pole,
                                     double gain)
{
    int64_t ll;
    int h;
    if (len <= 1)
        return;
    for (h = 0; h < npoles; ++h) {
        const double p = pole[h];
        const int max = (int)ceil(log(1e-15) / log(fabs(p)));   /* deform.c:1046,1119 */
        double sum = p * ln[0];
        ln[0] = -p * ln[0];
        for (ll = 1; ll < len - 1; ++ll) {
            sum = p * (sum + ln[ll]);
            ln[ll] = p * (ln[ll - 1] - ln[ll]);
        }
        sum = (p / (p * p - 1.0)) * (sum + ln[len - 1]);
        ln[len - 2] += p * sum;
        ln[len - 1] = sum;
        for (ll = len - 2; ll >= 0; --ll)
            ln[ll] += p * ln[ll + 1];
        if (max < len) {
            double zn = p;
            for (ll = 1; ll < len; ++ll) {
                ln[ll] += zn * ln[0];
                zn *= p;
            }
        } else {
            double zn = p;
            const double iz = 1.0 / p;
            double z2n = pow(p, (double)(len - 1));
            ln[0] = ln[0] / (1.0 - z2n * z2n);
            ln[len - 1] += z2n * ln[0];
            z2n *= z2n * iz;
            for (ll = 1; ll <= len - 2; ++ll) {
                ln[ll] += (zn + z2n) * ln[0];
                zn *= p;
                z2n *= iz;
            }
        }
    }
    for (ll = 0; ll < len; ++ll)
        ln[ll] *= gain;
}

/*
 * spline_filter1d (transpose == 0) or NI_SplineFilter1DGrad (transpose != 0, deform.c:1049-1168)
 * along `axis`; every line goes through a double buffer and is cast to the output dtype once.
 * input and output may alias.
 */
int edo_spline_filter1d(const edhip_array* input, const edhip_array* output, int axis, int order,
                        int transpose, char* err, size_t errlen)
{
    double pole[2], gain, *buf;
    int npoles, d, nd = input->ndim;
    int64_t len, nlines = 1, line, i;
    int64_t idx[EDHIP_MAX_DIMS];

    if (order < 0 || order > 5) {                              /* _deform_grid.c:71-74 */
        set_err(err, errlen, "spline order not supported");
        return EDHIP_ERR_INVALID;
    }
    if (axis < 0)
        axis += nd;                                            /* _deform_grid.c:75-77 */
    if (axis < 0 || axis >= nd) {
        set_err(err, errlen, "invalid axis");
        return EDHIP_ERR_INVALID;
    }
    npoles = spline_poles(order, transpose != 0, pole, &gain);
    len = input->shape[axis];
    for (d = 0; d < nd; ++d)
        if (d != axis)
            nlines *= input->shape[d];
    if (len < 1 || nlines < 1)
        return EDHIP_OK;
    buf = malloc(sizeof(double) * len);
    if (!buf) {
        set_err(err, errlen, "out of memory");
        return EDHIP_ERR_MEMORY;
    }
    for (d = 0; d < nd; ++d)
        idx[d] = 0;
    for (line = 0; line < nlines; ++line) {
        const char* pi = (const char*)input->data;
        char* po = (char*)output->data;
        for (d = 0; d < nd; ++d) {
            if (d == axis)
                continue;
            pi += idx[d] * input->stride_bytes[d];
            po += idx[d] * output->stride_bytes[d];
        }
        for (i = 0; i < len; ++i) {
            if (!load_elem(pi + i * input->stride_bytes[axis], input->dtype, &buf[i])) {
                free(buf);
                set_err(err, errlen, "data type not supported");
                return EDHIP_ERR_DTYPE;
            }
        }
        if (npoles > 0) {
            if (transpose)
                prefilter_transpose_line(buf, len, npoles, pole, gain);
            else
                prefilter_line(buf, len, npoles, pole, gain);
        }
        for (i = 0; i < len; ++i) {
            if (!store_cast(po + i * output->stride_bytes[axis], output->dtype, buf[i])) {
                free(buf);
                set_err(err, errlen, "data type not supported");
                return EDHIP_ERR_DTYPE;
            }
        }
        for (d = nd - 1; d >= 0; --d) {
            if (d == axis)
                continue;
            if (++idx[d] < input->shape[d])
                break;
            idx[d] = 0;
        }
    }
    free(buf);
    return EDHIP_OK;
}

int edo_version(void) { return EDHIP_VERSION; }
