/* examples/c_driver.c -- the C ABI of include/tnqs.h driven from plain C: the same call sequence the Julia shim of
 * INTEGRATION.md (and the Python host) issues.  One TFIM Trotter layer (Rx on every site, Rzz on every edge, gates listed
 * colour by colour) on an L x L open square lattice from the all-up product state, then <Z_v> on every vertex.
 *
 *   gcc -O2 -Iinclude examples/c_driver.c -o c_driver -Ltensornetworkquantumsimulator.jl_amd -ltnqs_hip -lm \
 *       -Wl,-rpath,$PWD/tensornetworkquantumsimulator.jl_amd
 *   ./c_driver 3 2            (L = 3, maxdim = 2)   ->  one line per vertex: "v  Re<Z>  Im<Z>", then "truncerr_sum ..."
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "tnqs.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, tnqs_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 3, maxdim = argc > 2 ? atoi(argv[2]) : 2;
    const double hx = 1.0, J = 0.5, dt = 0.25;
    const int nv = L * L, ne = 2 * L * (L - 1);
    int32_t* es = malloc(sizeof(int32_t) * ne); int32_t* ed = malloc(sizeof(int32_t) * ne); int32_t* sd = malloc(sizeof(int32_t) * nv);
    int e = 0;
    for (int r = 0; r < L; ++r) for (int c = 0; c < L; ++c) {          /* vertex id = r * L + c */
        sd[r * L + c] = 2;
        if (c + 1 < L) { es[e] = r * L + c; ed[e] = r * L + c + 1; ++e; }
        if (r + 1 < L) { es[e] = r * L + c; ed[e] = (r + 1) * L + c; ++e; }
    }
    tnqs_handle h;
    CHECK(tnqs_create(nv, ne, es, ed, sd, TNQS_C128, 0, &h));
    /* circuit: Rx(2 hx dt) everywhere, then Rzz(2 J dt) per edge, horizontal-even / horizontal-odd / vertical-even / vertical-odd */
    const int ng = nv + ne;
    int32_t* nverts = malloc(sizeof(int32_t) * ng); int32_t* verts = malloc(sizeof(int32_t) * (nv + 2 * ne));
    double* mats = calloc((size_t)nv * 8 + (size_t)ne * 32, sizeof(double));
    size_t mo = 0, vo = 0; int g = 0;
    const double th = 2 * hx * dt, c1 = cos(th / 2), s1 = sin(th / 2);
    for (int v = 0; v < nv; ++v) {                                       /* Rx = [[c, -i s], [-i s, c]], column-major complex128 */
        nverts[g++] = 1; verts[vo++] = v;
        double* m = mats + mo; m[0] = c1; m[3] = -s1; m[5] = -s1; m[6] = c1; mo += 8;
    }
    const double ph = 2 * J * dt / 2;                                    /* Rzz(theta) = diag(e^{-i theta/2}, e^{+}, e^{+}, e^{-}) */
    for (int colour = 0; colour < 4; ++colour)
        for (int k = 0; k < ne; ++k) {
            int a = es[k], b = ed[k], horiz = (b == a + 1);
            int par = horiz ? (a % L) % 2 : (a / L) % 2;
            if ((horiz ? 0 : 2) + par != colour) continue;
            nverts[g++] = 2; verts[vo++] = a; verts[vo++] = b;
            double* m = mats + mo;
            for (int d = 0; d < 4; ++d) { double sgn = (d == 0 || d == 3) ? -1.0 : 1.0; m[2 * (d + 4 * d)] = cos(ph); m[2 * (d + 4 * d) + 1] = sgn * sin(ph); }
            mo += 32;
        }
    tnqs_apply_opts ao = { maxdim, 1e-12, 1, -1.0, 1 };
    tnqs_bp_opts bo = { 200, 1e-13, 1, 0, NULL, NULL };
    double* terr = calloc(ng, sizeof(double));
    tnqs_apply_stats st;
    CHECK(tnqs_apply_gates(h, ng, nverts, verts, mats, &ao, &bo, terr, &st));
    double* ops = calloc((size_t)nv * 8, sizeof(double)); double* out = calloc((size_t)nv * 2, sizeof(double));
    for (int v = 0; v < nv; ++v) { ops[8 * v + 0] = 1.0; ops[8 * v + 6] = -1.0; }       /* Z */
    CHECK(tnqs_expect_all(h, ops, out));
    for (int v = 0; v < nv; ++v) printf("%d %.12f %.3e\n", v, out[2 * v], out[2 * v + 1]);
    double tsum = 0; for (int k = 0; k < ng; ++k) tsum += terr[k];
    int chi = 0; CHECK(tnqs_maxvirtualdim(h, &chi));
    printf("truncerr_sum %.12e maxvirtualdim %d bp_updates %d\n", tsum, chi, st.n_bp_updates);
    CHECK(tnqs_destroy(h));
    return 0;
}
