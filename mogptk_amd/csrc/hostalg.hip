// hostalg.hip -- host-side pair algebra of the MOSM kernel in native code (no device work).
// Reference: gpr/multioutput.py:178-204 (the cross-spectral parameters of every channel pair) and the reverse-mode gradient autograd
// takes through them.  The Python mirror (mogptk_amd/gpr/multioutput.py) does the same with ~60 small numpy calls = 0.25 ms per
// evaluation, next to 13 ms of device work at configs[1]; these two loops take microseconds.  Used for the plain MOSM kernel only -- its
// siblings (MOSK, uMOSM, MOHSM) override pieces of the algebra and stay on the numpy path.
#include "../../include/mogp_hip.h"

#include <cmath>
#include <vector>

namespace {
inline int64_t i3(int a, int q, int d, int Q, int D) { return ((int64_t)a * Q + q) * D + d; }
}

extern "C" {

// table [C][C][Q][2+3D] = [A, Psi, V_d, M_d, Delta_d] from the constrained weight (C,Q), mean / variance / delay (C,Q,D), phase (C,Q)
int mogp_mosm_terms(int C, int Q, int D, const double* w, const double* mu, const double* v, const double* th, const double* ph,
                    double twopi, double phase_scale, double* table) {
    if (C <= 0 || Q <= 0 || D <= 0 || !w || !mu || !v || !th || !ph || !table) return MOGP_EINVAL;
    const int W = 2 + 3 * D;
    const double pi2 = M_PI * M_PI;
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j)
            for (int q = 0; q < Q; ++q) {
                double* row = table + (((int64_t)i * C + j) * Q + q) * W;
                if (i == j) {                                              // reference :183-187
                    double prod = 1.0;
                    for (int d = 0; d < D; ++d) {
                        const double vi = v[i3(i, q, d, Q, D)];
                        prod *= vi;
                        row[2 + d] = vi; row[2 + D + d] = mu[i3(i, q, d, Q, D)]; row[2 + 2 * D + d] = 0.0;
                    }
                    row[0] = w[i * Q + q] * w[i * Q + q] * twopi * std::sqrt(prod);
                    row[1] = 0.0;
                    continue;
                }
                double e = 0.0, prod = 1.0;
                for (int d = 0; d < D; ++d) {
                    const double vi = v[i3(i, q, d, Q, D)], vj = v[i3(j, q, d, Q, D)], mi = mu[i3(i, q, d, Q, D)], mj = mu[i3(j, q, d, Q, D)];
                    const double inv = 1.0 / (vi + vj), dm = mi - mj;
                    e += dm * inv * dm;
                    const double V = 2.0 * vi * inv * vj;
                    prod *= V;
                    row[2 + d] = V;
                    row[2 + D + d] = inv * (vi * mj + vj * mi);
                    row[2 + 2 * D + d] = th[i3(i, q, d, Q, D)] - th[i3(j, q, d, Q, D)];
                }
                row[0] = w[i * Q + q] * w[j * Q + q] * std::exp(-pi2 * e) * (twopi * std::sqrt(prod));
                row[1] = phase_scale * (ph[i * Q + q] - ph[j * Q + q]);
            }
    return MOGP_OK;
}

// chain rule table -> constrained parameters.  gtable is zero for i < j and carries the double count of the off-diagonal pairs already.
int mogp_mosm_terms_backward(int C, int Q, int D, const double* w, const double* mu, const double* v, const double* th, const double* ph,
                             double twopi, double phase_scale, const double* gtable, double* gw, double* gmu, double* gv, double* gth,
                             double* gph) {
    (void)th; (void)ph;
    if (C <= 0 || Q <= 0 || D <= 0 || !w || !mu || !v || !gtable || !gw || !gmu || !gv || !gth || !gph) return MOGP_EINVAL;
    const int W = 2 + 3 * D;
    const double pi2 = M_PI * M_PI;
    for (int k = 0; k < C * Q; ++k) gw[k] = gph[k] = 0.0;
    for (int k = 0; k < C * Q * D; ++k) gmu[k] = gv[k] = gth[k] = 0.0;
    std::vector<double> inv(D), dm(D), V(D), M(D);
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j)
            for (int q = 0; q < Q; ++q) {
                const double* g = gtable + (((int64_t)i * C + j) * Q + q) * W;
                const double gA = g[0];
                if (i == j) {
                    double prod = 1.0;
                    for (int d = 0; d < D; ++d) prod *= v[i3(i, q, d, Q, D)];
                    const double root = twopi * std::sqrt(prod);
                    const double wi = w[i * Q + q];
                    gw[i * Q + q] += 2.0 * gA * root * wi;                                  // d (w^2 root) / d w
                    const double A = wi * wi * root;
                    for (int d = 0; d < D; ++d) {
                        gv[i3(i, q, d, Q, D)] += gA * A / (2.0 * v[i3(i, q, d, Q, D)]) + g[2 + d];
                        gmu[i3(i, q, d, Q, D)] += g[2 + D + d];
                    }
                    continue;
                }
                double e = 0.0, prod = 1.0;
                for (int d = 0; d < D; ++d) {
                    const double vi = v[i3(i, q, d, Q, D)], vj = v[i3(j, q, d, Q, D)], mi = mu[i3(i, q, d, Q, D)], mj = mu[i3(j, q, d, Q, D)];
                    inv[d] = 1.0 / (vi + vj); dm[d] = mi - mj;
                    e += dm[d] * inv[d] * dm[d];
                    V[d] = 2.0 * vi * inv[d] * vj;
                    M[d] = inv[d] * (vi * mj + vj * mi);
                    prod *= V[d];
                }
                const double F = std::exp(-pi2 * e) * (twopi * std::sqrt(prod));            // A without the magnitude
                const double wi = w[i * Q + q], wj = w[j * Q + q];
                const double gmag = gA * F, gAA = gA * F * wi * wj;
                gw[i * Q + q] += gmag * wj;
                gw[j * Q + q] += gmag * wi;
                gph[i * Q + q] += phase_scale * g[1];
                gph[j * Q + q] -= phase_scale * g[1];
                for (int d = 0; d < D; ++d) {
                    const int64_t a = i3(i, q, d, Q, D), b = i3(j, q, d, Q, D);
                    const double vi = v[a], vj = v[b], mi = mu[a], mj = mu[b];
                    const double gV = g[2 + d], gMi = g[2 + D + d] * inv[d], gD = g[2 + 2 * D + d];
                    gth[a] += gD; gth[b] -= gD;
                    const double dA = gAA * (-2.0 * pi2) * (dm[d] * inv[d]);
                    gmu[a] += dA + gMi * vj;
                    gmu[b] += gMi * vi - dA;
                    const double inv2 = inv[d] * inv[d];
                    const double common = gAA * (pi2 * dm[d] * dm[d] * inv2), half = 0.5 * gAA / V[d];
                    gv[a] += common + (half + gV) * (2.0 * vj * vj * inv2) + gMi * (mj - M[d]);
                    gv[b] += common + (half + gV) * (2.0 * vi * vi * inv2) + gMi * (mi - M[d]);
                }
            }
    return MOGP_OK;
}

}  // extern "C"
