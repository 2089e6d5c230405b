/* ---------------------------------------------------------------------------
 * TEST INFRASTRUCTURE ONLY -- NOT PART OF THE PRODUCT PATH.
 *
 * CPU oracle for the MI355X-native torch-interpol hot path: a plain-C
 * restatement of the reference algorithm (balbasty/torch-interpol @2024_10_08)
 *   interpol/pushpull.py:35-233  (dispatch: iso0 / iso1 / nd)
 *   interpol/nd.py:10-464        (generic B-spline pull/push/grad/pushgrad/hess)
 *   interpol/iso0.py, iso1.py    (nearest / linear special cases)
 *   interpol/bounds.py:30-89     (index wrap + sign)
 *   interpol/splines.py:30-195   (weights and derivatives, orders 0-7)
 *   interpol/coeff.py:35-347     (interpolating prefilter)
 *
 * Parity pinning: this oracle is checked (tests/test_oracle_vs_reference.py,
 * run in the build container where /root/reference is importable) against the
 * reference itself, and everywhere against the golden vectors committed under
 * tests/golden/ that were generated from the reference by
 * tests/golden/make_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (torch-interpol_amd/interpol) never does.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp -shared).
 * ------------------------------------------------------------------------- */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define SIGN_NONE 2   /* Bound.transform returned None (bounds.py:88-89) */

typedef struct {
    int  dim;            /* D in {1,2,3}                                        */
    int  bound[3];       /* bounds.py:8-15 codes 0..6                           */
    int  order[3];       /* splines.py:7-15 codes 0..7                          */
    int  extrapolate;    /* bounds.py:18-21 : 0 no, 1 yes, 2 hist               */
    int  force_nd;       /* skip the iso0/iso1 dispatch (call nd.* directly)    */
    int  threads;        /* OpenMP threads                                      */
    int  vol_bcast;      /* indexed volume has batch 1, broadcast over B        */
    int  grid_bcast;     /* grid has batch 1                                    */
    int  val_bcast;      /* push/pushgrad source values have batch 1            */
    int  _pad;
    long B, C;
    long vol_shape[3];   /* lattice that is indexed: input (pull/grad/hess) or
                            target (push/count/pushgrad) spatial shape          */
    long vol_numel;      /* prod(vol_shape[:dim])                               */
    long n_samples;      /* sample points per batch item                        */
} oracle_cfg;

/* Python-style remainder: result in [0, m) for m > 0 (torch.remainder) */
static inline long pymod(long a, long m) { long r = a % m; return r < 0 ? r + m : r; }

/* ---- bounds.py:30-60 Bound.index ---------------------------------------- */
long oracle_bound_index(int type, long i, long n)
{
    switch (type) {
    case 0: case 1:                                   /* zero / replicate */
        return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    case 3: case 5: {                                 /* dct2 / dst2 */
        long n2 = n * 2;
        i = (i < 0) ? (n2 - 1) - pymod(-i - 1, n2) : pymod(i, n2);
        if (i >= n) i = -i + (n2 - 1);
        return i;
    }
    case 2: {                                         /* dct1 */
        if (n == 1) return 0;
        long n2 = (n - 1) * 2;
        i = pymod(i < 0 ? -i : i, n2);
        if (i >= n) i = -i + n2;
        return i;
    }
    case 4: {                                         /* dst1 */
        long n2 = 2 * (n + 1);
        if (i < 0) i = -i - 2;
        i = pymod(i, n2);
        if (i > n) i = -i + (n2 - 2);
        if (i == -1) i = 0;
        if (i == n) i = n - 1;
        return i;
    }
    case 6:                                           /* dft */
        return pymod(i, n);
    default:
        return i;
    }
}

/* ---- bounds.py:62-89 Bound.transform (SIGN_NONE when it returns None) ---- */
int oracle_bound_sign(int type, long i, long n)
{
    switch (type) {
    case 4: {                                         /* dst1 (quirk B-3 kept) */
        if (n == 1) return SIGN_NONE;
        long n2 = 2 * (n + 1);
        if (i < 0) i = -i + (n - 1);
        i = pymod(i, n2);
        int x = (i == 0) ? 0 : 1;
        if (pymod(i, n + 1) == n) x = 0;
        long q = i / (n + 1);                         /* floor_div on a non-negative */
        if (pymod(q, 2) > 0) x = -x;
        return x;
    }
    case 5: {                                         /* dst2 */
        if (i < 0) i = n - 1 - i;
        long q = i / n;
        return (pymod(q, 2) > 0) ? -1 : 1;
    }
    case 0:                                           /* zero */
        return (i < 0 || i >= n) ? 0 : 1;
    default:
        return SIGN_NONE;
    }
}

/* ---- coeff.py:35-65 get_poles -------------------------------------------- */
int oracle_get_poles(int order, double *poles)
{
    switch (order) {
    case 0: case 1: return 0;
    case 2: poles[0] = sqrt(8.) - 3.; return 1;
    case 3: poles[0] = sqrt(3.) - 2.; return 1;
    case 4:
        poles[0] = sqrt(664. - sqrt(438976.)) + sqrt(304.) - 19.;
        poles[1] = sqrt(664. + sqrt(438976.)) - sqrt(304.) - 19.;
        return 2;
    case 5:
        poles[0] = sqrt(67.5 - sqrt(4436.25)) + sqrt(26.25) - 6.5;
        poles[1] = sqrt(67.5 + sqrt(4436.25)) - sqrt(26.25) - 6.5;
        return 2;
    case 6:
        poles[0] = -0.488294589303044755130118038883789062112279161239377608394;
        poles[1] = -0.081679271076237512597937765737059080653379610398148178525368;
        poles[2] = -0.00141415180832581775108724397655859252786416905534669851652709;
        return 3;
    case 7:
        poles[0] = -0.5352804307964381655424037816816460718339231523426924148812;
        poles[1] = -0.122554615192326690515272264359357343605486549427295558490763;
        poles[2] = -0.0091486948096082769285930216516478534156925639545994482648003;
        return 3;
    default: return -1;
    }
}

#define REAL float
#define SUFFIX _f32
#define REAL_IS_DOUBLE 0
#define FABS fabsf
#define FLOOR floorf
#define NEARBYINT nearbyintf
#include "interpol_oracle_body.inc"
#undef REAL
#undef SUFFIX
#undef REAL_IS_DOUBLE
#undef FABS
#undef FLOOR
#undef NEARBYINT

#define REAL double
#define SUFFIX _f64
#define REAL_IS_DOUBLE 1
#define FABS fabs
#define FLOOR floor
#define NEARBYINT nearbyint
#include "interpol_oracle_body.inc"
#undef REAL
#undef SUFFIX
#undef REAL_IS_DOUBLE
#undef FABS
#undef FLOOR
#undef NEARBYINT

int oracle_abi_version(void) { return 1; }
