/*
 * hk_detmath.h — the numeric contract for transcendental functions on the kart hot path.
 *
 * Why this exists.  The reference evaluates Mathf.Sin/Cos/Atan2/Exp/Pow as the platform's double
 * libm result rounded to float (e.g. HierarchicalKartAgent.cs:734,821-831,1341-1344,
 * ArcadeKart.cs:300) and Math.Sin/Cos/Atan2 in double (KartLQRDynamics.cs:45-48, HKA:1343).  No two libm's
 * (Mono/.NET CRT, glibc, ROCm ocml) agree in the last bit, so "bit-reproducible episode outcomes" is only
 * definable against ONE pinned implementation.  This header is that implementation: pure IEEE-754
 * double + - * / (no FMA: build with -ffp-contract=off), identical on the host (gcc) and on gfx950 (hipcc),
 * so the HIP kernels and the CPU oracle produce the same bits.  tests/test_detmath.py bounds every function
 * against an 80-digit mpmath reference (<= 2 ulp), i.e. ~1e-16 relative from whatever libm the reference ran
 * on — nine orders of magnitude inside the 1e-4 parity tolerance of BASELINE.json.
 *
 * Domain: arguments that occur on the path (|x| < 1e5 for sin/cos, finite for atan2, x in [-700, 700] for exp).
 * Constants: tools/gen_detmath.py (mpmath, 80 digits).
 */
#ifndef HK_DETMATH_H
#define HK_DETMATH_H

#if defined(__HIPCC__)
#define HK_HD __host__ __device__ static inline
#else
#define HK_HD static inline
#endif

#define HK_PI_F 3.14159274f /* UnityEngine.Mathf.PI as float */

/* Horner step of the polynomial kernels.  The contract is mul then add (two roundings); -DHK_DETMATH_FMA=1 is a TIMING experiment
 * (tools/build_variant.py): one fused step, other bits, never shipped. */
#ifndef HK_DETMATH_FMA
#define HK_DETMATH_FMA 0
#endif
#if HK_DETMATH_FMA
#define HK_HORNER(p, z, c) __builtin_fma((p), (z), (c))
#else
#define HK_HORNER(p, z, c) ((p) * (z) + (c))
#endif

HK_HD double hk_fabs(double x) { return x < 0.0 ? -x : x; }

/* r = x - k*pi/2 (3-part Cody-Waite), returns k mod 4 */
HK_HD int hk__rem_pio2(double x, double* r)
{
    const double INV_PIO2 = 0.63661977236758138;
    const double P1 = 1.5707963267341256, P2 = 6.077100506303966e-11, P3 = 2.0222662487959506e-21;
    double t = x * INV_PIO2;
    int k = (int)(t < 0.0 ? t - 0.5 : t + 0.5);      /* |x| < 1e9 on this path (32-bit convert: native on gfx950) */
    double fk = (double)k;
    double y = x - fk * P1;
    y = y - fk * P2;
    y = y - fk * P3;
    *r = y;
    return (int)(k & 3);
}

HK_HD double hk__ksin(double r)
{
    double z = r * r;
    double p = 1.9572941063391263e-20;
    p = HK_HORNER(p, z, -8.2206352466243295e-18);
    p = HK_HORNER(p, z, 2.8114572543455206e-15);
    p = HK_HORNER(p, z, -7.6471637318198164e-13);
    p = HK_HORNER(p, z, 1.6059043836821613e-10);
    p = HK_HORNER(p, z, -2.505210838544172e-08);
    p = HK_HORNER(p, z, 2.7557319223985893e-06);
    p = HK_HORNER(p, z, -0.00019841269841269841);
    p = HK_HORNER(p, z, 0.0083333333333333332);
    p = HK_HORNER(p, z, -0.16666666666666666);
    return r + (r * z) * p;
}

HK_HD double hk__kcos(double r)
{
    double z = r * r;
    double p = -8.8967913924505741e-22;
    p = HK_HORNER(p, z, 4.1103176233121648e-19);
    p = HK_HORNER(p, z, -1.5619206968586225e-16);
    p = HK_HORNER(p, z, 4.7794773323873853e-14);
    p = HK_HORNER(p, z, -1.1470745597729725e-11);
    p = HK_HORNER(p, z, 2.08767569878681e-09);
    p = HK_HORNER(p, z, -2.7557319223985888e-07);
    p = HK_HORNER(p, z, 2.4801587301587302e-05);
    p = HK_HORNER(p, z, -0.0013888888888888889);
    p = HK_HORNER(p, z, 0.041666666666666664);
    return (1.0 - 0.5 * z) + (z * z) * p;
}

HK_HD double hk_sin(double x)
{
    double r;
    int q = hk__rem_pio2(x, &r);
    switch (q) {
    case 0: return hk__ksin(r);
    case 1: return hk__kcos(r);
    case 2: return -hk__ksin(r);
    default: return -hk__kcos(r);
    }
}

HK_HD double hk_cos(double x)
{
    double r;
    int q = hk__rem_pio2(x, &r);
    switch (q) {
    case 0: return hk__kcos(r);
    case 1: return -hk__ksin(r);
    case 2: return -hk__kcos(r);
    default: return hk__ksin(r);
    }
}

/* sin and cos of the same argument: one range reduction, both kernels evaluated once (branch-free), results selected by
 * quadrant.  Bit-identical to (hk_sin(x), hk_cos(x)) by construction — same reduction, same kernels (tests/test_detmath.py).
 * The HIP kernels call this where they need the pair (a wave's lanes sit in different quadrants, so the switch of hk_sin /
 * hk_cos made every lane evaluate both kernels anyway — twice when the pair was needed). */
HK_HD void hk_sincos(double x, double* s, double* c)
{
    double r;
    const int q = hk__rem_pio2(x, &r);
    const double ks = hk__ksin(r), kc = hk__kcos(r);
    const double a = (q & 1) ? kc : ks;          /* |sin| source */
    const double b = (q & 1) ? ks : kc;          /* |cos| source */
    *s = (q & 2) ? -a : a;                       /* q: 0 ks, 1 kc, 2 -ks, 3 -kc */
    *c = ((q + 1) & 2) ? -b : b;                 /* q: 0 kc, 1 -ks, 2 -kc, 3 ks */
}

/* The same pair for an argument that is almost always near zero (the per-tick AngleAxis rotations: a few degrees): for
 * |x| < 0.78 < pi/4 the reduction picks k = 0 and leaves r = x exactly, so the two kernels can be evaluated on x directly —
 * bit-identical to hk_sincos, without the reduction and the quadrant selects; any other argument takes the general path. */
HK_HD void hk_sincos_near0(double x, double* s, double* c)
{
    if (hk_fabs(x) < 0.78) { *s = hk__ksin(x); *c = hk__kcos(x); return; }
    hk_sincos(x, s, c);
}

/* atan(j / 8), j = 0 .. 8, split hi + lo — as a chain of selects on literals, not a table: a table indexed by a computed j sits in constant
 * memory, and on the GPU every lookup is a per-lane global load in the middle of a dependent chain (seven of them per assembled player) */
HK_HD double hk__atan_hi(int j)
{
    double r = 0.0;
    r = j == 1 ? 0.12435499454676144 : r; r = j == 2 ? 0.24497866312686414 : r; r = j == 3 ? 0.35877067027057225 : r;
    r = j == 4 ? 0.46364760900080609 : r; r = j == 5 ? 0.55859931534356244 : r; r = j == 6 ? 0.64350110879328437 : r;
    r = j == 7 ? 0.71882999962162453 : r; r = j == 8 ? 0.78539816339744828 : r;
    return r;
}
HK_HD double hk__atan_lo(int j)
{
    double r = 0.0;
    r = j == 1 ? -3.1253241424539383e-18 : r; r = j == 2 ? 1.0698755618734451e-17 : r; r = j == 3 ? -2.4623815582638635e-17 : r;
    r = j == 4 ? 2.2698777452961687e-17 : r; r = j == 5 ? -5.4556305485916264e-18 : r; r = j == 6 ? 1.5834785051444286e-17 : r;
    r = j == 7 ? -2.1478388444456983e-17 : r; r = j == 8 ? 3.061616997868383e-17 : r;
    return r;
}

/* atan(t) for t in [0,1] */
HK_HD double hk__atan01(double t)
{
    int j = (int)(t * 8.0 + 0.5);
    double c = (double)j * 0.125;
    double u = (t - c) / (1.0 + t * c);
    double z = u * u;
    double p = 0.058823529411764705;
    p = HK_HORNER(p, z, -0.066666666666666666);
    p = HK_HORNER(p, z, 0.076923076923076927);
    p = HK_HORNER(p, z, -0.090909090909090912);
    p = HK_HORNER(p, z, 0.1111111111111111);
    p = HK_HORNER(p, z, -0.14285714285714285);
    p = HK_HORNER(p, z, 0.20000000000000001);
    p = HK_HORNER(p, z, -0.33333333333333331);
    double a = u + (u * z) * p;
    return hk__atan_hi(j) + (hk__atan_lo(j) + a);
}

/* atan2 with the usual quadrant conventions; atan2(0,0) = 0, atan2(+-0, x<0) = +-pi */
HK_HD double hk_atan2(double y, double x)
{
    const double PI_HI = 3.1415926535897931, PI_LO = 1.2246467991473532e-16;
    const double PIO2_HI = 1.5707963267948966, PIO2_LO = 6.123233995736766e-17;
    double ax = hk_fabs(x), ay = hk_fabs(y);
    /* one evaluation of atan01 on min / max (on a GPU the lanes of a wave sit on both sides of the diagonal: written as two
       branches, both were executed, divisions included); atan2(0, 0): 0 / 1 -> atan01(0) = 0 */
    const int lower = ay <= ax;                       /* below the diagonal: atan(ay / ax), else pi/2 - atan(ax / ay) */
    const double num = lower ? ay : ax;
    double dnm = lower ? ax : ay;
    if (dnm == 0.0) dnm = 1.0;
    const double a = hk__atan01(num / dnm);
    double r = lower ? a : (PIO2_HI - a) + PIO2_LO;
    if (x < 0.0) r = (PI_HI - r) + PI_LO;
    return y < 0.0 ? -r : r;
}

HK_HD double hk_exp(double x)
{
    const double INV_LN2 = 1.4426950408889634, LN2_HI = 0.69314718036912382, LN2_LO = 1.9082149292705877e-10;
    double t = x * INV_LN2;
    if (t > 2000.0) t = 2000.0;
    if (t < -2000.0) t = -2000.0;
    int k = (int)(t < 0.0 ? t - 0.5 : t + 0.5);
    double fk = (double)k;
    double r = (x - fk * LN2_HI) - fk * LN2_LO;
    double p = 1.1470745597729725e-11;
    p = HK_HORNER(p, r, 1.6059043836821613e-10);
    p = HK_HORNER(p, r, 2.08767569878681e-09);
    p = HK_HORNER(p, r, 2.505210838544172e-08);
    p = HK_HORNER(p, r, 2.7557319223985888e-07);
    p = HK_HORNER(p, r, 2.7557319223985893e-06);
    p = HK_HORNER(p, r, 2.4801587301587302e-05);
    p = HK_HORNER(p, r, 0.00019841269841269841);
    p = HK_HORNER(p, r, 0.0013888888888888889);
    p = HK_HORNER(p, r, 0.0083333333333333332);
    p = HK_HORNER(p, r, 0.041666666666666664);
    p = HK_HORNER(p, r, 0.16666666666666666);
    p = HK_HORNER(p, r, 0.5);
    double e = 1.0 + (r + (r * r) * p);
    /* scale by 2^k, k in [-1022, 1023] on this path */
    union { unsigned long long u; double d; } s;
    if (k < -1022) return 0.0;
    if (k > 1023) k = 1023;
    s.u = (unsigned long long)(k + 1023) << 52;
    return e * s.d;
}

/* natural log, x > 0 normal:  x = m * 2^e, m in [sqrt(1/2), sqrt(2));  log x = e ln2 + 2 atanh((m-1)/(m+1)) */
HK_HD double hk_log(double x)
{
    const double LN2_HI = 0.69314718036912382, LN2_LO = 1.9082149292705877e-10;
    union { unsigned long long u; double d; } b;
    b.d = x;
    int e = (int)((b.u >> 52) & 0x7ff) - 1023;
    b.u = (b.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL; /* m in [1, 2) */
    double m = b.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 0.037037037037037035;          /* 1/27 */
    p = HK_HORNER(p, z, 0.040000000000000001);         /* 1/25 */
    p = HK_HORNER(p, z, 0.043478260869565216);         /* 1/23 */
    p = HK_HORNER(p, z, 0.047619047619047616);         /* 1/21 */
    p = HK_HORNER(p, z, 0.052631578947368418);         /* 1/19 */
    p = HK_HORNER(p, z, 0.058823529411764705);         /* 1/17 */
    p = HK_HORNER(p, z, 0.066666666666666666);         /* 1/15 */
    p = HK_HORNER(p, z, 0.076923076923076927);         /* 1/13 */
    p = HK_HORNER(p, z, 0.090909090909090912);         /* 1/11 */
    p = HK_HORNER(p, z, 0.1111111111111111);           /* 1/9 */
    p = HK_HORNER(p, z, 0.14285714285714285);          /* 1/7 */
    p = HK_HORNER(p, z, 0.20000000000000001);          /* 1/5 */
    p = HK_HORNER(p, z, 0.33333333333333331);          /* 1/3 */
    double l = 2.0 * (s + (s * z) * p);
    double fe = (double)e;
    return fe * LN2_HI + (l + fe * LN2_LO);
}

/* ---- Mathf.* = libm in double, result ROUNDED TO FLOAT (Q8).
 * The float callers (the per-tick physics: four sine / cosine pairs and an exponential per kart and tick) do not need the last 29 bits of
 * the double: these kernels evaluate in double to <= 2^-44 relative (truncated series on the reduced argument, every step ONE explicit
 * fused multiply-add: exact IEEE, the same bits from gcc and hipcc) and round once.  The result is the correctly rounded float — what
 * (float)Math.Sin((double)x) gives for any libm — except where the true value lies within 2^-44 of a rounding boundary (a fraction of
 * ~1e-6 of the arguments: there it is the neighbouring float).  tests/test_detmath.py bounds both against mpmath.  The double entry points
 * above (Math.Sin / Math.Cos / Math.Atan2 of KartLQRDynamics.cs and HKA.AngleDifference) keep their full <= 2 ulp kernels. */
#define HK_FMA(a, b, c) __builtin_fma((a), (b), (c))
HK_HD double hk__ksin_f(double r)       /* |r| <= pi/4 (+ rounding): r - r^3/3! + ... - r^15/15!, remainder < 2^-52 r */
{
    const double z = r * r;
    double p = -7.6471637318198164e-13;                 /* -1/15! */
    p = HK_FMA(p, z, 1.6059043836821613e-10);           /*  1/13! */
    p = HK_FMA(p, z, -2.505210838544172e-08);           /* -1/11! */
    p = HK_FMA(p, z, 2.7557319223985893e-06);           /*  1/9!  */
    p = HK_FMA(p, z, -0.00019841269841269841);          /* -1/7!  */
    p = HK_FMA(p, z, 0.0083333333333333332);            /*  1/5!  */
    p = HK_FMA(p, z, -0.16666666666666666);             /* -1/3!  */
    return HK_FMA(r * z, p, r);
}
HK_HD double hk__kcos_f(double r)       /* 1 - r^2/2! + ... + r^16/16! */
{
    const double z = r * r;
    double p = 4.7794773323873853e-14;                  /*  1/16! */
    p = HK_FMA(p, z, -1.1470745597729725e-11);          /* -1/14! */
    p = HK_FMA(p, z, 2.08767569878681e-09);             /*  1/12! */
    p = HK_FMA(p, z, -2.7557319223985888e-07);          /* -1/10! */
    p = HK_FMA(p, z, 2.4801587301587302e-05);           /*  1/8!  */
    p = HK_FMA(p, z, -0.0013888888888888889);           /* -1/6!  */
    p = HK_FMA(p, z, 0.041666666666666664);             /*  1/4!  */
    p = HK_FMA(p, z, -0.5);
    return HK_FMA(p, z, 1.0);
}
HK_HD int hk__rem_pio2_f(double x, double* r)     /* the reduction of hk__rem_pio2 with fused steps */
{
    const double INV_PIO2 = 0.63661977236758138;
    const double P1 = 1.5707963267341256, P2 = 6.077100506303966e-11, P3 = 2.0222662487959506e-21;
    const double t = x * INV_PIO2;
    const int k = (int)(t < 0.0 ? t - 0.5 : t + 0.5);
    const double fk = (double)k;
    double y = HK_FMA(-fk, P1, x);
    y = HK_FMA(-fk, P2, y);
    y = HK_FMA(-fk, P3, y);
    *r = y;
    return (int)(k & 3);
}
HK_HD void hk__sincos_f(double x, double* s, double* c)
{
    /* the three-term reduction below is exact for |k| < ~2^20 and its int cast needs a finite argument: beyond 1e5 (and for inf / NaN) the
     * double kernels with their full reduction take over (never on the hot path: headings are kept in [0, 2 pi)) */
    if (!(hk_fabs(x) < 1.0e5)) { hk_sincos(x, s, c); return; }
    double r;
    const int q = hk__rem_pio2_f(x, &r);
    const double ks = hk__ksin_f(r), kc = hk__kcos_f(r);
    const double a = (q & 1) ? kc : ks;
    const double b = (q & 1) ? ks : kc;
    *s = (q & 2) ? -a : a;
    *c = ((q + 1) & 2) ? -b : b;
}
HK_HD void hk_sincosf(float x, float* s, float* c) { double ds, dc; hk__sincos_f((double)x, &ds, &dc); *s = (float)ds; *c = (float)dc; }
HK_HD float hk_sinf(float x) { float s, c; hk_sincosf(x, &s, &c); (void)c; return s; }
HK_HD float hk_cosf(float x) { float s, c; hk_sincosf(x, &s, &c); (void)s; return c; }
/* an argument that is almost always near zero (the per-tick AngleAxis rotations, the wheels' steer angle): for |x| < 0.78 < pi/4 the
 * reduction picks k = 0 and leaves r = x exactly, so the kernels run on x directly — bit-identical to hk_sincosf */
HK_HD void hk_sincosf_near0(float x, float* s, float* c)
{
    const double dx = (double)x;
    if (hk_fabs(dx) < 0.78) { *s = (float)hk__ksin_f(dx); *c = (float)hk__kcos_f(dx); return; }
    hk_sincosf(x, s, c);
}
HK_HD double hk__atan01_f(double t)     /* atan(t), t in [0, 1]: the table of hk__atan01, the series of the remainder to u^9 (|u| <= 1/16: next term < 2^-47) */
{
    const int j = (int)((t == t ? t : 0.0) * 8.0 + 0.5);          /* (a NaN quotient — inf / inf, a NaN input — must not reach the int cast; it still propagates through u) */
    const double c = (double)j * 0.125;
    const double u = (t - c) / HK_FMA(t, c, 1.0);
    const double z = u * u;
    double p = 0.1111111111111111;
    p = HK_FMA(p, z, -0.14285714285714285);
    p = HK_FMA(p, z, 0.20000000000000001);
    p = HK_FMA(p, z, -0.33333333333333331);
    return hk__atan_hi(j) + HK_FMA(u * z, p, u);
}
HK_HD float hk_atan2f(float fy, float fx)
{
    const double PI = 3.1415926535897931, PIO2 = 1.5707963267948966;
    const double x = (double)fx, y = (double)fy;
    const double ax = hk_fabs(x), ay = hk_fabs(y);
    const int lower = ay <= ax;
    const double num = lower ? ay : ax;
    double dnm = lower ? ax : ay;
    if (dnm == 0.0) dnm = 1.0;
    const double a = hk__atan01_f(num / dnm);
    double r = lower ? a : PIO2 - a;
    if (x < 0.0) r = PI - r;
    return (float)(__builtin_signbit(fy) ? -r : r);      /* the SIGN BIT of y, as Math.Atan2: atan2(-0, x < 0) = -pi, atan2(-0, x > 0) = -0 */
}
HK_HD float hk_expf(float fx)           /* x = k ln2 + r, |r| <= ln2 / 2: e^r to r^11 / 11! (next term < 2^-47), scaled by 2^k */
{
    const double INV_LN2 = 1.4426950408889634, LN2_HI = 0.69314718036912382, LN2_LO = 1.9082149292705877e-10;
    const double x = (double)fx;
    double t = x * INV_LN2;
    if (t > 2000.0) t = 2000.0;
    if (t < -2000.0) t = -2000.0;
    int k = (int)(t < 0.0 ? t - 0.5 : t + 0.5);
    const double fk = (double)k;
    const double r = HK_FMA(-fk, LN2_LO, HK_FMA(-fk, LN2_HI, x));
    double p = 2.505210838544172e-08;                   /* 1/11! */
    p = HK_FMA(p, r, 2.7557319223985888e-07);           /* 1/10! */
    p = HK_FMA(p, r, 2.7557319223985893e-06);           /* 1/9!  */
    p = HK_FMA(p, r, 2.4801587301587302e-05);           /* 1/8!  */
    p = HK_FMA(p, r, 0.00019841269841269841);           /* 1/7!  */
    p = HK_FMA(p, r, 0.0013888888888888889);            /* 1/6!  */
    p = HK_FMA(p, r, 0.0083333333333333332);            /* 1/5!  */
    p = HK_FMA(p, r, 0.041666666666666664);             /* 1/4!  */
    p = HK_FMA(p, r, 0.16666666666666666);              /* 1/3!  */
    p = HK_FMA(p, r, 0.5);
    p = HK_FMA(p, r, 1.0);
    const double e = HK_FMA(p, r, 1.0);
    union { unsigned long long u; double d; } sc;
    if (k < -1022) return 0.0f;
    if (k > 1023) k = 1023;
    sc.u = (unsigned long long)(k + 1023) << 52;
    return (float)(e * sc.d);
}
HK_HD float hk_logf(float x) { return (float)hk_log((double)x); }

/* Pure-fp32 exp for the RL actor's Swish (x * 1 / (1 + exp(-x))): Cody-Waite reduction + degree-5 polynomial (the classic
 * cephes expf scheme), every multiply-add an explicit fmaf, so the oracle (gcc) and the kernels (hipcc) agree bit for bit.
 * <= 2 ulp of fp32 on [-87, 88] (tests/test_detmath.py); inputs are clamped to that range.  NOT used where the reference
 * calls Math.Exp / Mathf.Exp (those go through hk_exp): the exp inside Barracuda's Sigmoid kernel is not bit-defined, so
 * here only oracle == GPU matters. */
HK_HD float hk_expf_fast_core(float x)     /* x already inside [-87, 88] */
{
    const float fk = __builtin_rintf(x * 1.44269504f);
    float r = __builtin_fmaf(fk, -0.693359375f, x);
    r = __builtin_fmaf(fk, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float e = __builtin_fmaf(p, r * r, r);
    e = e + 1.0f;
    union { unsigned int u; float f; } sc;
    sc.u = (unsigned int)((int)fk + 127) << 23;
    return e * sc.f;
}
HK_HD float hk_expf_fast(float x)
{
    x = x > 88.0f ? 88.0f : (x < -87.0f ? -87.0f : x);
    return hk_expf_fast_core(x);
}

/* Swish of the RL actor (Sigmoid then Mul in the exported graph): s * (1 / (1 + exp(-s))) with the exp above.  The device
 * evaluates the same expression with its clamp as one v_med3_f32 (csrc/hk_swish.h; tests/test_swish_device.py compares the two
 * over every fp32 bit pattern on the GPU). */
HK_HD float hk_swishf(float s)
{
    return s * (1.0f / (1.0f + hk_expf_fast(-s)));
}

#endif /* HK_DETMATH_H */
