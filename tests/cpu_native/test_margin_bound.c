/* CPU check of the error budget behind the guarded walk's margins (docs/LOG.md §3b item 3, rt_accel.h kGuardGammaBound):
 * every hit the reference's hit_sphere COMPUTES — a true hit with a rounded root, or a phantom hit of a ray that
 * misses — lies within  gamma |o - c|^2 / (2 r)  of the sphere's surface, gamma = 24 * 2^-24.
 * Adversarial rays: far origins (up to 2000 units), tiny radii (down to 0.01), directions aimed AT the silhouette and a
 * few ulps inside / outside it (where hb^2 - a c cancels), unnormalised.  Geometry is evaluated in long double on
 * the float inputs.  Prints the largest budget actually used, in units of 2^-24.
 * Uses the oracle's hit_sphere (oracle/rt_oracle.c: include/sphere.h:24-53). */
#include <stdio.h>
#include <stdlib.h>
#include "../../oracle/rt_oracle.c"

static uint32_t rs = 0x9e3779b9u;
static double urand(void) { rs = rs * 1664525u + 1013904223u; return (rs >> 8) * (1.0 / 16777216.0); }

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 400000;
    const double u = 5.9604644775390625e-8;
    double worst = 0.0; long hits = 0, phantom = 0;
    for (long it = 0; it < n; ++it) {
        rt_sphere s;
        memset(&s, 0, sizeof s);
        for (int a = 0; a < 3; a++) s.center.e[a] = (float)((urand() - 0.5) * 200.0);
        s.radius = (float)(0.01 * pow(100.0, urand()));                 /* 0.01 … 1 */
        const double D = 2.0 * pow(1000.0, urand());                    /* 2 … 2000 from the centre */
        double dir[3], w[3];
        do { for (int a = 0; a < 3; a++) dir[a] = urand() - 0.5; } while (dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2] < 1e-3);
        double nl = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        for (int a = 0; a < 3; a++) dir[a] /= nl;
        ray r;
        for (int a = 0; a < 3; a++) r.o.e[a] = (float)(s.center.e[a] - D * dir[a]);
        /* w: unit vector perpendicular to dir */
        do {
            for (int a = 0; a < 3; a++) w[a] = urand() - 0.5;
            const double dp = w[0] * dir[0] + w[1] * dir[1] + w[2] * dir[2];
            for (int a = 0; a < 3; a++) w[a] -= dp * dir[a];
            nl = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        } while (nl < 1e-3);
        for (int a = 0; a < 3; a++) w[a] /= nl;
        static const double offs[] = {0.0, 1e-7, -1e-7, 3e-7, -3e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-3, -0.5};
        const double off = offs[it % 12];
        const double scale = 0.05 * pow(400.0, urand());                /* |d| from 0.05 D to 20 D */
        for (int a = 0; a < 3; a++) r.d.e[a] = (float)(scale * ((s.center.e[a] + w[a] * s.radius * (1.0 + off)) - r.o.e[a]));
        hitrec rec;
        if (!hit_sphere(&r, 0.001f, 1e30f, &rec, &s)) continue;
        hits++;
        long double oc2 = 0, p2 = 0, od = 0, dd = 0;
        for (int a = 0; a < 3; a++) {
            const long double oc = (long double)r.o.e[a] - s.center.e[a];
            const long double p = oc + (long double)rec.t * r.d.e[a];
            oc2 += oc * oc; p2 += p * p; od += oc * r.d.e[a]; dd += (long double)r.d.e[a] * r.d.e[a];
        }
        const long double miss2 = oc2 - od * od / dd;                  /* squared distance of the line from the centre */
        if (miss2 > (long double)s.radius * s.radius) phantom++;
        const long double departure = fabsl(sqrtl(p2) - s.radius);
        const long double used = departure * 2.0L * s.radius / oc2 / u;   /* the gamma this hit needed, in ulps */
        if ((double)used > worst) worst = (double)used;
    }
    printf("rays %ld  computed hits %ld  phantom hits %ld  largest budget used %.3f ulp (bound 24)\n", n, hits, phantom, worst);
    return worst < 24.0 ? 0 : 1;
}
