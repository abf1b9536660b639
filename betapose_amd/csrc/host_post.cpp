// Host-side (CPU, f64) tail of the hot path: the 2D-3D pose solve.
//
// The reference calls cv2.solvePnP(..., flags=SOLVEPNP_ITERATIVE) + cv2.Rodrigues
// (utils/utils.py:17-41).  OpenCV is a third-party dependency that is not in the
// reference tree and not installable here, so this is an own implementation of the
// published algorithm for N >= 6 non-planar points: DLT initialisation (12x12 normal
// matrix, smallest eigenvector, nearest rotation by SVD) followed by Levenberg-Marquardt
// on the pixel reprojection error over (rotation, translation).  The refinement runs to
// convergence of the same least-squares objective OpenCV minimises, so results agree
// to solver tolerance, not bit-for-bit ("parity unpinned" vs OpenCV).
#include <cmath>
#include <cstring>
#include <vector>

namespace bp {

// cyclic Jacobi for a symmetric n x n matrix (row-major, destroyed); V columns = eigenvectors
static void jacobi_eig(double* A, int n, double* V, double* w) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}

static double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
static void mul33(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    std::memcpy(C, T, sizeof(T));
}

// nearest rotation to M (polar decomposition through the eigen-decomposition of M^T M)
static void nearest_rotation(const double* M, double* R) {
    double MtM[9], V[9], w[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) MtM[i * 3 + j] = M[i] * M[j] + M[3 + i] * M[3 + j] + M[6 + i] * M[6 + j];
    jacobi_eig(MtM, 3, V, w);
    // U = M V S^-1 ; R = U V^T = M V S^-1 V^T
    double S[9] = {0};
    for (int i = 0; i < 3; ++i) {
        const double s = std::sqrt(std::max(w[i], 1e-300));
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) S[r * 3 + c] += V[r * 3 + i] * V[c * 3 + i] / s;
    }
    mul33(M, S, R);
    if (det3(R) < 0) {
        // flip the direction of least stretch
        int m = 0;
        for (int i = 1; i < 3; ++i)
            if (w[i] < w[m]) m = i;
        double S2[9] = {0};
        for (int i = 0; i < 3; ++i) {
            const double s = std::sqrt(std::max(w[i], 1e-300)) * (i == m ? -1.0 : 1.0);
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) S2[r * 3 + c] += V[r * 3 + i] * V[c * 3 + i] / s;
        }
        mul33(M, S2, R);
    }
}

static void rodrigues_exp(const double* w, double* R) {
    const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double a, b;
    if (th < 1e-8) { a = 1.0 - th * th / 6.0; b = 0.5 - th * th / 24.0; }
    else { a = std::sin(th) / th; b = (1.0 - std::cos(th)) / (th * th); }
    const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double K2[9];
    mul33(K, K, K2);
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * K[i] + b * K2[i];
}

static bool solve6(double* A, double* b) {   // Gaussian elimination with partial pivoting, in place
    const int n = 6;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(A[r * n + c]) > std::fabs(A[piv * n + c])) piv = r;
        if (std::fabs(A[piv * n + c]) < 1e-300) return false;
        if (piv != c) {
            for (int k = 0; k < n; ++k) std::swap(A[c * n + k], A[piv * n + k]);
            std::swap(b[c], b[piv]);
        }
        for (int r = c + 1; r < n; ++r) {
            const double f = A[r * n + c] / A[c * n + c];
            for (int k = c; k < n; ++k) A[r * n + k] -= f * A[c * n + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; --r) {
        double s = b[r];
        for (int k = r + 1; k < n; ++k) s -= A[r * n + k] * b[k];
        b[r] = s / A[r * n + r];
    }
    return true;
}

static double reproj_cost(const double* P, const double* U, int n, const double* K, const double* R, const double* t,
                          double* res) {
    double c = 0;
    for (int i = 0; i < n; ++i) {
        const double* X = P + 3 * i;
        const double Y0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
        const double Y1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
        const double Y2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
        if (!(Y2 > 1e-9)) return HUGE_VAL;   // behind the camera: reject the step
        const double u = K[0] * Y0 / Y2 + K[2], v = K[4] * Y1 / Y2 + K[5];
        const double ru = u - U[2 * i], rv = v - U[2 * i + 1];
        if (res) { res[2 * i] = ru; res[2 * i + 1] = rv; }
        c += ru * ru + rv * rv;
    }
    return c;
}

int solve_pnp(const double* P, const double* U, int n, const double* K, double* Rout, double* tout) {
    if (n < 6) return -1;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    // ---- DLT on normalised image coordinates; object points are centred and scaled first
    // (Hartley conditioning: X_n = s (X - c)), which keeps the 12x12 system well conditioned for
    // objects a few centimetres across; the similarity is undone analytically below.
    double cen[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) cen[k] += P[3 * i + k] / n;
    double md = 0;
    for (int i = 0; i < n; ++i) {
        const double dx = P[3 * i] - cen[0], dy = P[3 * i + 1] - cen[1], dz = P[3 * i + 2] - cen[2];
        md += std::sqrt(dx * dx + dy * dy + dz * dz) / n;
    }
    if (!(md > 0)) return -2;
    const double sN = std::sqrt(3.0) / md;
    double LL[144] = {0};
    for (int i = 0; i < n; ++i) {
        const double x = (U[2 * i] - cx) / fx, y = (U[2 * i + 1] - cy) / fy;
        const double X = sN * (P[3 * i] - cen[0]), Y = sN * (P[3 * i + 1] - cen[1]), Z = sN * (P[3 * i + 2] - cen[2]);
        const double r1[12] = {X, Y, Z, 1, 0, 0, 0, 0, -x * X, -x * Y, -x * Z, -x};
        const double r2[12] = {0, 0, 0, 0, X, Y, Z, 1, -y * X, -y * Y, -y * Z, -y};
        for (int a = 0; a < 12; ++a)
            for (int b = 0; b < 12; ++b) LL[a * 12 + b] += r1[a] * r1[b] + r2[a] * r2[b];
    }
    double V[144], w[12];
    jacobi_eig(LL, 12, V, w);
    int m = 0;
    for (int i = 1; i < 12; ++i)
        if (w[i] < w[m]) m = i;
    double RR[9], tt[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) RR[r * 3 + c] = V[(r * 4 + c) * 12 + m];
        tt[r] = V[(r * 4 + 3) * 12 + m];
    }
    // P' = mu [R/s | R c + t]: fix the sign so the object centre lies in front of the camera
    if (tt[2] < 0) {
        for (int i = 0; i < 9; ++i) RR[i] = -RR[i];
        for (int i = 0; i < 3; ++i) tt[i] = -tt[i];
    }
    double nrm = 0;
    for (int i = 0; i < 9; ++i) nrm += RR[i] * RR[i];
    nrm = std::sqrt(nrm);
    if (!(nrm > 0)) return -2;
    double R[9], t[3];
    nearest_rotation(RR, R);
    const double mu = nrm / std::sqrt(3.0);   // = mu_true / s
    for (int i = 0; i < 3; ++i) {
        const double tc = tt[i] / (mu * sN);   // R c + t
        t[i] = tc - (R[i * 3] * cen[0] + R[i * 3 + 1] * cen[1] + R[i * 3 + 2] * cen[2]);
    }

    // ---- Levenberg-Marquardt, left-multiplicative rotation update
    std::vector<double> res(2 * n), res2(2 * n);
    double cost = reproj_cost(P, U, n, K, R, t, res.data());
    double lambda = 1e-3;
    for (int it = 0; it < 100; ++it) {
        double JtJ[36] = {0}, Jtr[6] = {0};
        for (int i = 0; i < n; ++i) {
            const double* X = P + 3 * i;
            const double a0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
            const double a1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
            const double a2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
            const double Y0 = a0 + t[0], Y1 = a1 + t[1], Y2 = a2 + t[2];
            const double iz = 1.0 / Y2;
            const double du[3] = {fx * iz, 0, -fx * Y0 * iz * iz};
            const double dv[3] = {0, fy * iz, -fy * Y1 * iz * iz};
            // dY/dw = -[a]x  (columns), dY/dt = I
            const double dYdw[9] = {0, a2, -a1, -a2, 0, a0, a1, -a0, 0};
            double Ju[6], Jv[6];
            for (int c = 0; c < 3; ++c) {
                Ju[c] = du[0] * dYdw[c] + du[1] * dYdw[3 + c] + du[2] * dYdw[6 + c];
                Jv[c] = dv[0] * dYdw[c] + dv[1] * dYdw[3 + c] + dv[2] * dYdw[6 + c];
                Ju[3 + c] = du[c];
                Jv[3 + c] = dv[c];
            }
            for (int a = 0; a < 6; ++a) {
                Jtr[a] += Ju[a] * res[2 * i] + Jv[a] * res[2 * i + 1];
                for (int b = 0; b < 6; ++b) JtJ[a * 6 + b] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
            }
        }
        bool improved = false;
        for (int tries = 0; tries < 30 && !improved; ++tries) {
            double A[36], d[6];
            for (int i = 0; i < 36; ++i) A[i] = JtJ[i];
            for (int i = 0; i < 6; ++i) { A[i * 6 + i] *= (1.0 + lambda); d[i] = -Jtr[i]; }
            if (!solve6(A, d)) { lambda *= 10; continue; }
            double dR[9], Rn[9], tn[3];
            rodrigues_exp(d, dR);
            mul33(dR, R, Rn);
            for (int i = 0; i < 3; ++i) tn[i] = t[i] + d[3 + i];
            const double c2 = reproj_cost(P, U, n, K, Rn, tn, res2.data());
            if (c2 < cost) {
                const double step = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
                const double rel = (cost - c2) / std::max(cost, 1e-300);
                std::memcpy(R, Rn, sizeof(Rn));
                std::memcpy(t, tn, sizeof(tn));
                res.swap(res2);
                cost = c2;
                lambda = std::max(lambda * 0.1, 1e-16);
                improved = true;
                if (step < 1e-14 || rel < 1e-16) it = 1000;
            } else {
                lambda *= 10;
            }
        }
        if (!improved) break;
    }
    // re-orthonormalise against drift
    double Rn[9];
    nearest_rotation(R, Rn);
    std::memcpy(Rout, Rn, sizeof(Rn));
    std::memcpy(tout, t, sizeof(t));
    return 0;
}

}  // namespace bp
