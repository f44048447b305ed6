/* detmath.h — bit-reproducible scalar math shared by the HIP kernels, the host
 * library and (by inclusion) the CPU oracle and the reference-kernel shim.
 *
 * Why this exists: the reference kernel calls the OpenCL built-ins cos, log, pow,
 * pown, atan2pi, sqrt, normalize, mix, sign, min, max, clamp, fabs, dot, cross
 * (/root/reference/src/render.cl:151-153,157,162,177,383,390,427-462,480). OpenCL
 * leaves their results implementation-defined within ULP bounds (cos <=4, log <=3,
 * pow <=16, atan2pi <=6 ulp; + - * exact IEEE). A path tracer is chaotic: a 1-ulp
 * difference flips a hit/miss or a `material->x > random_float()` branch, so parity
 * "within 1e-4 per pixel" is only reachable when every realisation of the kernel
 * uses the SAME rounding everywhere. detmath pins one realisation built from nothing
 * but IEEE-754 correctly rounded + - * / sqrt, fused multiply-add (fmaf: one rounding,
 * fully specified by IEEE-754, a single instruction on gfx950 and on every x86-64 since
 * Haswell / Zen), conversions and integer ops, so the x86-64 build (gcc/clang) and the
 * gfx950 build (hipcc, default correctly-rounded f32 divide/sqrt, f32 denormals on)
 * produce identical bits. tests/test_detmath.py checks the ULP bounds against libm.
 *
 * Revision 2 (round 2): the built-ins are written the way a GPU's OpenCL library
 * writes them -- dot / cross / mix and the polynomial kernels of cos / log / atan2pi
 * as FMA chains, normalize as v * rsqrt(dot(v, v)) with a division-free Newton rsqrt --
 * because on gfx950 they are most of the trace kernel's instructions (DESIGN.md
 * "Numerics"). The OpenCL bounds they have to meet: dot / cross / mix: at least as
 * accurate as the unfused forms; normalize <= 2 + n ulp per component (n = 3);
 * cos <= 4, log <= 3, atan2pi <= 6, pow <= 16 ulp. Fused operations appear ONLY
 * where written as dm_fmaf: everything else, including every expression of the
 * reference kernel itself, stays unfused (-ffp-contract=off).
 *
 * Every file that includes this header MUST be compiled with -ffp-contract=off.
 * Polynomial coefficients: simple-raytracer_amd/tools/gen_detmath_coeffs.py.
 */
#ifndef SRT_DETMATH_H
#define SRT_DETMATH_H

/* The numerics are FROZEN at this revision: tests/golden/*.npz carry it (make_golden.py stamps it), tests/golden/MANIFEST.json
 * records it next to the files' hashes and this header's, and tests/test_golden_manifest.py fails when any of them moves.
 * Changing a built-in below means: bump the number, regenerate the goldens, re-run tests/test_oracle_statistics.py with
 * /root/reference present and commit its report under profiles/ (DESIGN.md "Numerics"). */
#define DM_REVISION 2

#include <stdint.h>

#if defined(__HIPCC__)
#define DM_FN __host__ __device__ static __forceinline__
#else
#define DM_FN static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define DM_SQRTF(x) __builtin_sqrtf(x)
#define DM_SQRT(x) __builtin_sqrt(x)
#else
#define DM_SQRTF(x) __builtin_sqrtf(x)
#define DM_SQRT(x) __builtin_sqrt(x)
#endif

/* ---- bit casts ------------------------------------------------------------ */
DM_FN uint32_t dm_f2u(float f) {
	uint32_t u;
	__builtin_memcpy(&u, &f, 4);
	return u;
}
DM_FN float dm_u2f(uint32_t u) {
	float f;
	__builtin_memcpy(&f, &u, 4);
	return f;
}
DM_FN uint64_t dm_d2u(double d) {
	uint64_t u;
	__builtin_memcpy(&u, &d, 8);
	return u;
}
DM_FN double dm_u2d(uint64_t u) {
	double d;
	__builtin_memcpy(&d, &u, 8);
	return d;
}

#define DM_INF_F dm_u2f(0x7f800000u)
#define DM_NAN_F dm_u2f(0x7fc00000u)

/* ---- the "loose" OpenCL built-ins, pinned --- */
/* OpenCL C 6.15.4: fmin-less integer-style definitions for min/max. */
DM_FN float dm_min(float x, float y) { return y < x ? y : x; }
DM_FN float dm_max(float x, float y) { return x < y ? y : x; }
DM_FN float dm_clamp(float x, float lo, float hi) { return dm_min(dm_max(x, lo), hi); }
DM_FN float dm_fabs(float x) { return dm_u2f(dm_f2u(x) & 0x7fffffffu); }
/* sign(): 1 for x>0, -1 for x<0, the (signed) zero itself for +-0, 0 for NaN. */
DM_FN float dm_sign(float x) {
	if (x > 0.0f) return 1.0f;
	if (x < 0.0f) return -1.0f;
	if (x == 0.0f) return x;
	return 0.0f;
}
/* The one fused operation: a * b + c with a single rounding (IEEE-754 fusedMultiplyAdd). */
DM_FN float dm_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
/* mix(x, y, a) = x + (y - x) * a, the product and sum fused */
DM_FN float dm_mix(float x, float y, float a) { return dm_fmaf(y - x, a, x); }
/* dot(a, b) = fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)) */
DM_FN float dm_dot3(float ax, float ay, float az, float bx, float by, float bz) { return dm_fmaf(az, bz, dm_fmaf(ay, by, ax * bx)); }
/* one component of cross(a, b): p * q - r * t as fma(p, q, -(r * t)) */
DM_FN float dm_cross1(float p, float q, float r, float t) { return dm_fmaf(p, q, -(r * t)); }
/* 1 / sqrt(d) without a division or a square root: the classic exponent-halving first guess (3.4 % off) and three
 * Newton steps y <- y + y (0.5 - (0.5 d y) y), each two FMAs / products and one FMA; the last one is written in the
 * correction form, which leaves ~1 ulp for normal d > 0. The products are ordered so that d = 0 stays finite (a large
 * value: normalize(0) = 0, as OpenCL specifies); d = inf or NaN does not give a number. */
DM_FN float dm_rsqrtf(float d) {
	float y = dm_u2f(0x5f375a86u - (dm_f2u(d) >> 1));
	const float h = 0.5f * d;
	y = y * dm_fmaf(-(h * y), y, 1.5f);
	y = y * dm_fmaf(-(h * y), y, 1.5f);
	y = dm_fmaf(y, dm_fmaf(-(h * y), y, 0.5f), y);
	return y;
}
/* normalize(v) = v * rsqrt(dot(v, v)); out may alias v */
DM_FN void dm_normalize3(const float v[3], float out[3]) {
	const float r = dm_rsqrtf(dm_dot3(v[0], v[1], v[2], v[0], v[1], v[2]));
	out[0] = v[0] * r, out[1] = v[1] * r, out[2] = v[2] * r;
}
/* sqrt is IEEE correctly rounded on both targets. */
DM_FN float dm_sqrtf(float x) { return DM_SQRTF(x); }

/* ---- cos(x): accurate for |x| <= 16384 (the kernel feeds [0, 2*pi]) --------- */
/* Cody-Waite reduction by pi/2 = C1 + C2 + C3 (three floats, 72 bits) with one FMA per
 * part, then degree-2-in-z tails for sin and cos evaluated as FMA chains. */
DM_FN float dm_cosf(float x) {
	const float TWO_OVER_PI = 6.36619747e-01f;
	const float C1 = 1.570796371e+00f;          /* 0x3fc90fdb = float(pi/2) */
	const float C2 = -4.371138829e-08f;         /* 0xb33bbd2e = float(pi/2 - C1) */
	const float C3 = -1.715124510e-15f;         /* 0xa6f72ced */
	float ax = dm_fabs(x);
	if (!(ax < 16384.0f)) {
		/* out of the supported range (or NaN/inf): NaN for non-finite, else best effort */
		if (!(ax < DM_INF_F)) return DM_NAN_F;
	}
	int k = (int)dm_fmaf(ax, TWO_OVER_PI, 0.5f);
	float fk = (float)k;
	float r = dm_fmaf(-fk, C1, ax);
	r = dm_fmaf(-fk, C2, r);
	r = dm_fmaf(-fk, C3, r);
	float z = r * r;
	int odd = k & 1;
	/* coefficients: sin tail (odd quadrants) or cos tail (even quadrants) */
	float c0 = odd ? -1.66666642e-01f : 4.16666642e-02f;
	float c1 = odd ? 8.33272468e-03f : -1.38882792e-03f;
	float c2 = odd ? -1.95828557e-04f : 2.45428964e-05f;
	float p = dm_fmaf(z, dm_fmaf(z, c2, c1), c0);
	float s_res = dm_fmaf(r * z, p, r);                              /* sin(r) */
	float c_res = dm_fmaf(z * z, p, dm_fmaf(-0.5f, z, 1.0f));        /* cos(r) */
	float res = odd ? s_res : c_res;
	/* cos(x): k&3 = 0: cos r, 1: -sin r, 2: -cos r, 3: sin r */
	int neg = ((k + 1) >> 1) & 1;
	return neg ? -res : res;
}

/* ---- log(x), natural, full float domain ----------------------------------- */
DM_FN float dm_logf(float x) {
	const float LN2_HI = 6.93138123e-01f;   /* 0x3f317180 */
	const float LN2_LO = 9.05800061e-06f;   /* 0x3717f7d1 */
	const float L0 = 6.66666687e-01f;
	const float L1 = 4.00001287e-01f;
	const float L2 = 2.85499692e-01f;
	const float L3 = 2.33534276e-01f;
	uint32_t ix = dm_f2u(x);
	int k = 0;
	if (ix < 0x00800000u || (ix >> 31)) {
		if ((ix << 1) == 0) return -DM_INF_F;          /* log(+-0) = -inf */
		if (ix >> 31) return DM_NAN_F;                 /* log(<0) = NaN  */
		k -= 25;                                       /* subnormal: scale up */
		x = x * 33554432.0f;
		ix = dm_f2u(x);
	}
	if (ix >= 0x7f800000u) return x;                   /* inf or NaN */
	k += (int)(ix >> 23) - 127;
	ix &= 0x007fffffu;
	/* normalise the mantissa to [sqrt(1/2), sqrt(2)) */
	uint32_t i = (ix + 0x4afb20u) & 0x00800000u;
	x = dm_u2f(ix | (i ^ 0x3f800000u));
	k += (int)(i >> 23);
	float f = x - 1.0f;
	float s = f / (2.0f + f);
	float z = s * s;
	float R = z * dm_fmaf(z, dm_fmaf(z, dm_fmaf(z, L3, L2), L1), L0);
	float hfsq = (0.5f * f) * f;
	float dk = (float)k;
	return dm_fmaf(dk, LN2_HI, f - (hfsq - dm_fmaf(s, hfsq + R, dk * LN2_LO)));
}

/* ---- double-precision helpers for pow / atan2pi (once per escaping path) --- */
/* log(x) for finite normal positive double x, rel. error ~1e-16. */
DM_FN double dm_log_pos_d(double x) {
	uint64_t ux = dm_d2u(x);
	int k = (int)(ux >> 52) - 1023;
	ux = (ux & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
	double m = dm_u2d(ux);                              /* [1, 2) */
	if (m > 1.4142135623730951) {
		m = m * 0.5;
		k += 1;
	}
	double s = (m - 1.0) / (m + 1.0);
	double z = s * s;
	/* atanh series: log m = 2 s (1 + z/3 + z^2/5 + ... + z^11/23) */
	double p = 1.0 / 23.0;
	p = p * z + 1.0 / 21.0;
	p = p * z + 1.0 / 19.0;
	p = p * z + 1.0 / 17.0;
	p = p * z + 1.0 / 15.0;
	p = p * z + 1.0 / 13.0;
	p = p * z + 1.0 / 11.0;
	p = p * z + 1.0 / 9.0;
	p = p * z + 1.0 / 7.0;
	p = p * z + 1.0 / 5.0;
	p = p * z + 1.0 / 3.0;
	p = p * z + 1.0;
	return (double)k * 0.6931471805599453 + (2.0 * s) * p;
}

/* exp(t) for |t| < 700, rel. error ~1e-16. */
DM_FN double dm_exp_d(double t) {
	const double INV_LN2 = 1.4426950408889634;
	const double LN2_HI = 0x1.62e42fee00000p-1;
	const double LN2_LO = 0x1.a39ef35793c76p-33;
	double fn = t * INV_LN2;
	int n = (int)(fn < 0.0 ? fn - 0.5 : fn + 0.5);
	double dn = (double)n;
	double r = (t - dn * LN2_HI) - dn * LN2_LO;
	/* Taylor to r^13/13!, |r| <= 0.35 */
	double p = 1.0 / 6227020800.0;
	p = p * r + 1.0 / 479001600.0;
	p = p * r + 1.0 / 39916800.0;
	p = p * r + 1.0 / 3628800.0;
	p = p * r + 1.0 / 362880.0;
	p = p * r + 1.0 / 40320.0;
	p = p * r + 1.0 / 5040.0;
	p = p * r + 1.0 / 720.0;
	p = p * r + 1.0 / 120.0;
	p = p * r + 1.0 / 24.0;
	p = p * r + 1.0 / 6.0;
	p = p * r + 0.5;
	p = p * r + 1.0;
	p = p * r + 1.0;
	/* scale by 2^n, n in [-1020, 1020] here */
	double sc = dm_u2d((uint64_t)(n + 1023) << 52);
	return p * sc;
}

/* Exponent of the integer fast path of dm_powf: n in 1..32 when y is exactly that
 * integer, else 0. (The kernel's exponent is the wave-uniform SceneData.sun_focus,
 * 25.0 by default, src/main.cpp:123, so the choice of path never diverges.) */
DM_FN int dm_pow_small_int(float y) {
	if (!(y >= 1.0f && y <= 32.0f)) return 0;
	int n = (int)y;
	return ((float)n == y) ? n : 0;
}

/* x^n, 1 <= n <= 32, by binary exponentiation in DOUBLE (at most 5 squarings + 4
 * products, relative error < 1e-15), rounded once to float: < 0.5000001 ulp. Signs,
 * zeros, inf and NaN come out right by plain IEEE multiplication; overflow/underflow
 * happen in the final conversion (|x|^32 stays far inside the double range). */
DM_FN float dm_powi(float x, int n) {
	double b = (double)x, r = b;
	int first = 1;
	unsigned un = (unsigned)n;
	while (un) {
		if (un & 1u) {
			r = first ? b : r * b;
			first = 0;
		}
		un >>= 1;
		if (un) b = b * b;
	}
	return (float)r;
}

/* pow(x, y), float in / float out. Small positive integer y: dm_powi. Otherwise
 * computed in double: <1 ulp. Follows the OpenCL/C99 special-case table for the cases
 * a renderer can reach. */
DM_FN float dm_powf(float x, float y) {
	if (y == 0.0f) return 1.0f;
	if (x == 1.0f) return 1.0f;
	if (x != x || y != y) return DM_NAN_F;
	{
		int n = dm_pow_small_int(y);
		if (n) return dm_powi(x, n);
	}
	uint32_t uy = dm_f2u(y);
	float ay = dm_fabs(y);
	/* integer-ness / parity of y */
	int y_is_int = 0, y_is_odd = 0;
	if (ay >= 16777216.0f) {
		y_is_int = 1;
	} else {
		int iy = (int)ay;
		if ((float)iy == ay) {
			y_is_int = 1;
			y_is_odd = iy & 1;
		}
	}
	float ax = dm_fabs(x);
	int neg_result = 0;
	if (dm_f2u(x) >> 31) {
		if (ax == 0.0f || ax == DM_INF_F) {
			neg_result = y_is_odd;
		} else {
			if (!y_is_int) return DM_NAN_F;
			neg_result = y_is_odd;
		}
	}
	float res;
	if (ax == 0.0f) {
		res = (uy >> 31) ? DM_INF_F : 0.0f;
	} else if (ax == DM_INF_F) {
		res = (uy >> 31) ? 0.0f : DM_INF_F;
	} else if (ay == DM_INF_F) {
		if (ax == 1.0f) res = 1.0f;
		else res = ((ax > 1.0f) != (int)(uy >> 31)) ? DM_INF_F : 0.0f;
	} else {
		double t = (double)y * dm_log_pos_d((double)ax);
		if (t > 90.0) res = DM_INF_F;
		else if (t < -105.0) res = 0.0f;
		else res = (float)dm_exp_d(t);
	}
	return neg_result ? -res : res;
}

/* pown(double, int) by binary exponentiation; pown(x,5) = x * ((x*x)*(x*x)).
 * (reference: shlick_reflectance, src/render.cl:177, evaluated in fp64) */
DM_FN double dm_pown_d(double x, int n) {
	unsigned un = n < 0 ? 0u - (unsigned)n : (unsigned)n;
	double result = 1.0, base = x;
	while (un) {
		if (un & 1u) result = result * base;
		un >>= 1;
		if (un) base = base * base;
	}
	return n < 0 ? 1.0 / result : result;
}

/* atan2pi(y, x) = atan2(y, x) / pi, OpenCL C 7.5.1 edge cases. Float only:
 * a = min/max in [0,1] (one IEEE division), atan(a)/pi = a * P(a^2) with a degree-9
 * near-minimax P (tools/gen_detmath_coeffs.py; FMA Horner), then exact-constant quadrant folds
 * 0.5 - t and 1 - t. Measured max error < 3 ulp against libm (bound: 6). */
DM_FN float dm_atan2pif(float y, float x) {
	if (x != x || y != y) return DM_NAN_F;
	uint32_t sy = dm_f2u(y) >> 31, sx = dm_f2u(x) >> 31;
	float ax = dm_fabs(x), ay = dm_fabs(y);
	float r;
	if (ay == 0.0f) {
		r = sx ? 1.0f : 0.0f;
	} else if (ax == 0.0f) {
		r = 0.5f;
	} else if (ay == DM_INF_F) {
		r = (ax == DM_INF_F) ? (sx ? 0.75f : 0.25f) : 0.5f;
	} else if (ax == DM_INF_F) {
		r = sx ? 1.0f : 0.0f;
	} else {
		int swap = ay > ax;
		float mn = swap ? ax : ay, mx = swap ? ay : ax;
		float a = mn / mx;
		float z = a * a;
		float p = -5.414992338e-04f;
		p = dm_fmaf(p, z, 3.338322509e-03f);
		p = dm_fmaf(p, z, -9.661298245e-03f);
		p = dm_fmaf(p, z, 1.817217097e-02f);
		p = dm_fmaf(p, z, -2.657799982e-02f);
		p = dm_fmaf(p, z, 3.479872271e-02f);
		p = dm_fmaf(p, z, -4.539104179e-02f);
		p = dm_fmaf(p, z, 6.365585327e-02f);
		p = dm_fmaf(p, z, -1.061031148e-01f);
		p = dm_fmaf(p, z, 3.183098733e-01f);
		float t = a * p;
		if (swap) t = 0.5f - t;
		if (sx) t = 1.0f - t;
		r = t;
	}
	return sy ? -r : r;
}

/* One channel of the LINEAR image filter (OpenCL 3.0 8.2): w00 T00 + w10 T10 + w01 T01 + w11 T11 as an FMA chain. */
DM_FN float dm_bilinear(float w00, float t00, float w10, float t10, float w01, float t01, float w11, float t11) {
	return dm_fmaf(w11, t11, dm_fmaf(w01, t01, dm_fmaf(w10, t10, w00 * t00)));
}

#endif /* SRT_DETMATH_H */
