/* ulp_check.c — measures detmath against glibc's double-precision libm.
 * usage: ulp_check <stride>   (stride 1 = every float in the domain)
 * prints one line per function: name max_ulp_error n_tested
 * OpenCL bounds to satisfy: cos<=4, log<=3, pow<=16, atan2pi<=6 ulp.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "detmath.h"

static double ulp_err(float got, double want) {
	if (isnan(want)) return isnan(got) ? 0.0 : 1e30;
	if (isinf(want)) return (isinf(got) && (got > 0) == (want > 0)) ? 0.0 : 1e30;
	float wf = (float)want;
	if (isinf(wf)) return (isinf(got)) ? 0.0 : 1e30;
	/* ulp of the correctly rounded result */
	int e;
	double aw = fabs(want);
	double u;
	if (aw < 1.17549435e-38) u = 1.401298464324817e-45;
	else {
		frexp(aw, &e);
		u = ldexp(1.0, e - 24);
	}
	return fabs((double)got - want) / u;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd(void) {
	rng_state = rng_state * 747796405u + 2891336453u;
	uint32_t r = ((rng_state >> ((rng_state >> 28) + 4)) ^ rng_state) * 277803737u;
	return (r >> 22) ^ r;
}

int main(int argc, char **argv) {
	unsigned stride = argc > 1 ? (unsigned)atoi(argv[1]) : 997;
	if (stride == 0) stride = 1;
	/* cos on [0, 2*pi] (every float the kernel can produce lives here) */
	{
		double worst = 0;
		unsigned long n = 0;
		uint32_t hi = dm_f2u(6.2831855f);
#pragma omp parallel for reduction(max : worst) reduction(+ : n) schedule(static)
		for (long b = 0; b <= (long)hi; b += stride) {
			float x = dm_u2f((uint32_t)b);
			double e = ulp_err(dm_cosf(x), cos((double)x));
			if (e > worst) worst = e;
			n++;
		}
		printf("cos %.4f %lu\n", worst, n);
	}
	/* cos on a wider band, sampled */
	{
		double worst = 0;
		for (int i = 0; i < 2000000; i++) {
			float x = ((float)rnd() / 4294967296.0f) * 200.0f - 100.0f;
			double e = ulp_err(dm_cosf(x), cos((double)x));
			if (e > worst) worst = e;
		}
		printf("cos_wide %.4f 2000000\n", worst);
	}
	/* log on (0, 1] and sampled over all positive floats */
	{
		double worst = 0;
		unsigned long n = 0;
#pragma omp parallel for reduction(max : worst) reduction(+ : n) schedule(static)
		for (long b = 1; b <= 0x3f800000L; b += stride) {
			float x = dm_u2f((uint32_t)b);
			double e = ulp_err(dm_logf(x), log((double)x));
			if (e > worst) worst = e;
			n++;
		}
		printf("log01 %.4f %lu\n", worst, n);
		worst = 0;
		n = 0;
#pragma omp parallel for reduction(max : worst) reduction(+ : n) schedule(static)
		for (long b = 0x3f800000L; b < 0x7f800000L; b += (long)stride * 4) {
			float x = dm_u2f((uint32_t)b);
			double e = ulp_err(dm_logf(x), log((double)x));
			if (e > worst) worst = e;
			n++;
		}
		printf("log_big %.4f %lu\n", worst, n);
	}
	/* pow: x in [0,1.5], y in (0, 200]; plus texel gamma pow(b/255, 2.2) */
	{
		double worst = 0;
		for (int i = 0; i < 3000000; i++) {
			float x = ((float)rnd() / 4294967296.0f) * 1.5f;
			float y = ((float)rnd() / 4294967296.0f) * ((i & 1) ? 200.0f : 30.0f);
			double e = ulp_err(dm_powf(x, y), pow((double)x, (double)y));
			if (e > worst) worst = e;
		}
		for (int i = 0; i < 1000000; i++) {
			float x = ((float)rnd() / 4294967296.0f) * 100.0f;
			float y = ((float)rnd() / 4294967296.0f) * 40.0f - 20.0f;
			double e = ulp_err(dm_powf(x, y), pow((double)x, (double)y));
			if (e > worst) worst = e;
		}
		for (int b = 0; b < 256; b++) {
			float x = (float)b / 255.0f;
			double e = ulp_err(dm_powf(x, 2.2f), pow((double)x, (double)2.2f));
			if (e > worst) worst = e;
		}
		printf("pow %.4f 4000256\n", worst);
		/* integer-exponent fast path: every n in 1..32, x over [0, 1.2] and a few wider */
		worst = 0;
		unsigned long n = 0;
		for (int e = 1; e <= 32; e++) {
			for (int i = 0; i < 400000; i++) {
				float x = ((float)rnd() / 4294967296.0f) * ((i & 7) ? 1.2f : 8.0f);
				if (i & 64) x = -x;
				double err = ulp_err(dm_powf(x, (float)e), pow((double)x, (double)e));
				if (err > worst) worst = err;
				n++;
			}
		}
		printf("powi %.4f %lu\n", worst, n);
	}
	/* atan2pi */
	{
		double worst = 0;
		for (int i = 0; i < 40000000; i++) {
			float y = ((float)rnd() / 4294967296.0f) * 2.0f - 1.0f;
			float x = ((float)rnd() / 4294967296.0f) * 2.0f - 1.0f;
			if (i & 1) y *= 1e-4f;
			if (i & 2) x *= 1e-3f;
			double want = atan2((double)y, (double)x) / 3.14159265358979323846;
			double e = ulp_err(dm_atan2pif(y, x), want);
			if (e > worst) worst = e;
		}
		printf("atan2pi %.4f 40000000\n", worst);
	}
	/* normalize: every component against v / |v| in double; bound 2 + n = 5 ulp. Random directions with component
	 * magnitudes from 1e-6 to 1e3 (the kernel normalises Gaussian triples, camera rays and mixes of unit vectors),
	 * plus rsqrt alone on every stride-th float in [2^-100, 2^100] */
	{
		double worst = 0;
		for (int i = 0; i < 20000000; i++) {
			float v[3], o[3];
			for (int c = 0; c < 3; c++) {
				float m = ((float)rnd() / 4294967296.0f) * 2.0f - 1.0f;
				int e = (int)(rnd() % 31u) - 20;
				v[c] = ldexpf(m, (i & 3) ? 0 : e);
			}
			dm_normalize3(v, o);
			double len = sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]);
			if (len == 0.0) continue;
			for (int c = 0; c < 3; c++) {
				double e = ulp_err(o[c], (double)v[c] / len);
				if (e > worst) worst = e;
			}
		}
		printf("normalize %.4f 20000000\n", worst);
		worst = 0;
		unsigned long n = 0;
		uint32_t lo = dm_f2u(0x1p-100f), hi = dm_f2u(0x1p100f);
#pragma omp parallel for reduction(max : worst) reduction(+ : n) schedule(static)
		for (long b = lo; b <= (long)hi; b += stride) {
			float x = dm_u2f((uint32_t)b);
			double e = ulp_err(dm_rsqrtf(x), 1.0 / sqrt((double)x));
			if (e > worst) worst = e;
			n++;
		}
		printf("rsqrt %.4f %lu\n", worst, n);
	}
	/* special values */
	{
		int bad = 0;
		float inf = INFINITY;
		bad += !(dm_logf(0.0f) == -inf);
		bad += !(dm_logf(1.0f) == 0.0f);
		bad += !(isnan(dm_logf(-1.0f)));
		bad += !(dm_cosf(0.0f) == 1.0f);
		bad += !(dm_powf(0.0f, 25.0f) == 0.0f);
		bad += !(dm_powf(1.0f, 25.0f) == 1.0f);
		bad += !(dm_powf(0.5f, 0.0f) == 1.0f);
		bad += !(dm_powf(2.0f, 3.0f) == 8.0f);
		bad += !(dm_powf(-2.0f, 3.0f) == -8.0f);
		bad += !(isnan(dm_powf(-2.0f, 0.5f)));
		{
			float zero[3] = {0.0f, 0.0f, 0.0f}, o[3];
			dm_normalize3(zero, o); /* OpenCL: normalize of the zero vector is the zero vector */
			bad += !(o[0] == 0.0f && o[1] == 0.0f && o[2] == 0.0f);
			float unit[3] = {0.0f, -3.0f, 0.0f};
			dm_normalize3(unit, o);
			bad += !(o[0] == 0.0f && fabsf(o[1] + 1.0f) < 3e-7f && o[2] == 0.0f);
		}
		bad += !(dm_atan2pif(0.0f, 1.0f) == 0.0f);
		bad += !(dm_atan2pif(0.0f, -1.0f) == 1.0f);
		bad += !(dm_atan2pif(-0.0f, -1.0f) == -1.0f);
		bad += !(dm_atan2pif(1.0f, 0.0f) == 0.5f);
		bad += !(dm_atan2pif(-1.0f, 0.0f) == -0.5f);
		bad += !(fabsf(dm_atan2pif(1.0f, 1.0f) - 0.25f) <= 3e-8f);
		bad += !(fabsf(dm_atan2pif(1.0f, -1.0f) - 0.75f) <= 6e-8f);
		bad += !(isnan(dm_atan2pif(NAN, 1.0f)));
		bad += !(dm_pown_d(0.5, 5) == 0.03125);
		bad += !(dm_sign(-0.0f) == 0.0f && dm_sign(3.0f) == 1.0f && dm_sign(-3.0f) == -1.0f && dm_sign(NAN) == 0.0f);
		bad += !(dm_min(1.0f, 2.0f) == 1.0f && dm_max(1.0f, 2.0f) == 2.0f);
		printf("specials %d 21\n", bad);
	}
	return 0;
}
