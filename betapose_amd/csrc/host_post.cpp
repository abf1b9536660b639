// Host-side (CPU, f64 / f32) tail of the hot path: the 2D-3D pose solve and the parametric pose NMS.
//
// ---- pnp (utils/utils.py:17-41): cv2.solvePnP(points_3D, points_2D, cameraMatrix, zeros(8)) + cv2.Rodrigues.
// OpenCV is a third-party dependency that is neither in the reference tree nor installable here (the reference pins no
// version), so solve_pnp_iterative() RESTATES the published algorithm behind the default flag SOLVEPNP_ITERATIVE --
// cvFindExtrinsicCameraParams2 of OpenCV's calib3d (calibration.cpp, 2.4 ... 4.6), step for step, in f64 as there:
//   1. image points -> normalised coordinates (x, y) = ((u - cx) / fx, (v - cy) / fy)         [zero distortion]
//   2. Mc = mean of the object points, MM = sum (M - Mc)(M - Mc)^T, singular values W, right vectors V
//   3. planar model (W[2] / W[1] < 1e-3): rotate the plane to z = 0, homography to the normalised points, columns
//      h1, h2 normalised, t = h3 * 2 / (|h1| + |h2|), third column h1 x h2, Rodrigues round trip to orthonormalise
//      otherwise DLT: 2N x 12 system, singular vector of the smallest singular value of L^T L -> [RR | tt]; negated if
//      det(RR) < 0; R = U V^T of RR's SVD; t = tt * |R|_F / |RR|_F          (object points are NOT pre-conditioned)
//   4. r = Rodrigues(R); CvLevMarq(6 parameters, 2N errors, criteria = 20 iterations | FLT_EPSILON) on (r, t):
//      err = project(r, t) - observed [pixels], J = [dp/dr | dp/dt]; lambda = 10^lambdaLg10 from -3; step solves
//      (J^T J with its diagonal scaled by 1 + lambda) d = J^T err, param = previous - d; a step that raises |err|
//      is retried with lambdaLg10 + 1 (up to 16); after an accepted step lambdaLg10 - 1 (down to -16); stop after 20
//      accepted steps or when |param - previous| / |previous| < FLT_EPSILON.
// The result is therefore what OpenCV returns (a 20-step, float-epsilon-terminated minimiser), not the exact optimum.
// Not restated: findHomography's own LM polish in the planar branch (the pose LM that follows refines the same
// reprojection error) and SVBkSb's singular-value thresholding (a pivoted elimination solves the damped 6x6 system).
// Parity against a real cv2 build is NOT pinned (no OpenCV here): the oracle holds an independent numpy restatement
// of the same steps (oracle/post_ref.py) and tests pin the two to each other and to known answers.
//
// ---- solve_pnp_ransac: the variant the reference keeps commented out (utils/utils.py:32-36, reprojectionError 12):
// hypotheses from 6-point samples through the solver above, inliers by pixel distance, adaptive trial count at
// confidence 0.99 (at most 100), final solve on the inliers.  OpenCV's own RNG stream and EPnP kernel are not
// restated; the sampler is a fixed-seed LCG so results are reproducible.
//
// ---- pose_nms (pPose_nms.py:24-122): the greedy cluster / merge over n candidate poses, f32 like the reference.
#include <algorithm>
#include <cfloat>
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

// U V^T of M's SVD = the orthogonal polar factor M (M^T M)^(-1/2), through the eigen-decomposition of M^T M
// (det(M) > 0 on every path that calls this, so the factor is a rotation)
static void polar_rotation(const double* M, double* R) {
    double MtM[9], V[9], w[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) MtM[i * 3 + j] = M[i] * M[j] + M[3 + i] * M[3 + j] + M[6 + i] * M[6 + j];
    jacobi_eig(MtM, 3, V, w);
    double S[9] = {0};
    for (int i = 0; i < 3; ++i) {
        const double s = std::sqrt(std::max(w[i], 1e-300));
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) S[r * 3 + c] += V[r * 3 + i] * V[c * 3 + i] / s;
    }
    mul33(M, S, R);
}

// Rodrigues vector -> matrix (cvRodrigues2, vector input)
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

// rotation matrix -> Rodrigues vector (cvRodrigues2, matrix input; angle in [0, pi])
static void rodrigues_log(const double* R, double* r) {
    const double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = 0.5 * std::sqrt(rx * rx + ry * ry + rz * rz);
    double c = 0.5 * (R[0] + R[4] + R[8] - 1.0);
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double th = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0.0; return; }
        // angle pi: the axis from the diagonal, signs from the off-diagonal sums
        double t = (R[0] + 1) * 0.5;
        r[0] = std::sqrt(std::max(t, 0.0));
        t = (R[4] + 1) * 0.5;
        r[1] = std::sqrt(std::max(t, 0.0)) * (R[1] < 0 ? -1.0 : 1.0);
        t = (R[8] + 1) * 0.5;
        r[2] = std::sqrt(std::max(t, 0.0)) * (R[2] < 0 ? -1.0 : 1.0);
        if (std::fabs(r[0]) < std::fabs(r[1]) && std::fabs(r[0]) < std::fabs(r[2]) && (R[5] > 0) != (r[1] * r[2] > 0)) r[2] = -r[2];
        const double nr = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        for (int i = 0; i < 3; ++i) r[i] *= th / std::max(nr, 1e-300);
        return;
    }
    const double k = th / (2.0 * s);
    r[0] = rx * k; r[1] = ry * k; r[2] = rz * k;
}

static bool solve_n(double* A, double* b, int n) {   // Gaussian elimination with partial pivoting, in place
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

// cvProjectPoints2 with zero distortion: pixel residuals err = proj - observed and, when J != null, the 2N x 6
// Jacobian [dp/dr | dp/dt] (r = Rodrigues vector; d(R X)/dr = -R [X]x Jr(r), Jr = right Jacobian of SO(3))
static void project_residuals(const double* P, const double* U, int n, const double* K, const double* prm, double* err,
                              double* J) {
    double R[9];
    rodrigues_exp(prm, R);
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double Jr[9];
    if (J) {
        const double* w = prm;
        const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        double a, b;
        if (th < 1e-6) { a = 0.5 - th * th / 24.0; b = 1.0 / 6.0 - th * th / 120.0; }
        else { a = (1.0 - std::cos(th)) / (th * th); b = (th - std::sin(th)) / (th * th * th); }
        const double Kx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        double K2[9];
        mul33(Kx, Kx, K2);
        for (int i = 0; i < 9; ++i) Jr[i] = (i % 4 == 0 ? 1.0 : 0.0) - a * Kx[i] + b * K2[i];
    }
    for (int i = 0; i < n; ++i) {
        const double* X = P + 3 * i;
        const double Y0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + prm[3];
        const double Y1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + prm[4];
        const double Y2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + prm[5];
        const double iz = Y2 != 0.0 ? 1.0 / Y2 : 1.0;                      // as OpenCV: z = z ? 1./z : 1
        err[2 * i] = fx * Y0 * iz + cx - U[2 * i];
        err[2 * i + 1] = fy * Y1 * iz + cy - U[2 * i + 1];
        if (!J) continue;
        // dY/dr = -R [X]x Jr
        const double Xx[9] = {0, -X[2], X[1], X[2], 0, -X[0], -X[1], X[0], 0};
        double T[9], D[9];
        mul33(R, Xx, T);
        mul33(T, Jr, D);
        const double du[3] = {fx * iz, 0, -fx * Y0 * iz * iz};
        const double dv[3] = {0, fy * iz, -fy * Y1 * iz * iz};
        double* Ju = J + (2 * i) * 6;
        double* Jv = J + (2 * i + 1) * 6;
        for (int c = 0; c < 3; ++c) {
            Ju[c] = -(du[0] * D[c] + du[1] * D[3 + c] + du[2] * D[6 + c]);
            Jv[c] = -(dv[0] * D[c] + dv[1] * D[3 + c] + dv[2] * D[6 + c]);
            Ju[3 + c] = du[c];
            Jv[3 + c] = dv[c];
        }
    }
}

static double norm_l2(const double* v, int n) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += v[i] * v[i];
    return std::sqrt(s);
}

// homography  m ~ H (x, y, 1)  by the normalised DLT (9x9 normal matrix, smallest eigenvector)
static bool homography_dlt(const double* xy, const double* m, int n, double* H) {
    double c0[2] = {0, 0}, c1[2] = {0, 0};
    for (int i = 0; i < n; ++i) { c0[0] += xy[2 * i] / n; c0[1] += xy[2 * i + 1] / n; c1[0] += m[2 * i] / n; c1[1] += m[2 * i + 1] / n; }
    double d0 = 0, d1 = 0;
    for (int i = 0; i < n; ++i) {
        d0 += std::hypot(xy[2 * i] - c0[0], xy[2 * i + 1] - c0[1]) / n;
        d1 += std::hypot(m[2 * i] - c1[0], m[2 * i + 1] - c1[1]) / n;
    }
    if (!(d0 > 0) || !(d1 > 0)) return false;
    const double s0 = std::sqrt(2.0) / d0, s1 = std::sqrt(2.0) / d1;
    double A[81] = {0};
    for (int i = 0; i < n; ++i) {
        const double x = s0 * (xy[2 * i] - c0[0]), y = s0 * (xy[2 * i + 1] - c0[1]);
        const double u = s1 * (m[2 * i] - c1[0]), v = s1 * (m[2 * i + 1] - c1[1]);
        const double r1[9] = {x, y, 1, 0, 0, 0, -u * x, -u * y, -u};
        const double r2[9] = {0, 0, 0, x, y, 1, -v * x, -v * y, -v};
        for (int a = 0; a < 9; ++a)
            for (int b = 0; b < 9; ++b) A[a * 9 + b] += r1[a] * r1[b] + r2[a] * r2[b];
    }
    double V[81], w[9];
    jacobi_eig(A, 9, V, w);
    int k = 0;
    for (int i = 1; i < 9; ++i)
        if (w[i] < w[k]) k = i;
    double Hn[9];
    for (int i = 0; i < 9; ++i) Hn[i] = V[i * 9 + k];
    // H = T1^-1 Hn T0,  T = [s 0 -s c; 0 s -s c; 0 0 1]
    const double T0[9] = {s0, 0, -s0 * c0[0], 0, s0, -s0 * c0[1], 0, 0, 1};
    const double T1i[9] = {1 / s1, 0, c1[0], 0, 1 / s1, c1[1], 0, 0, 1};
    double T[9];
    mul33(Hn, T0, T);
    mul33(T1i, T, H);
    return true;
}

int solve_pnp_iterative(const double* P, const double* U, int n, const double* K, double* Rout, double* tout) {
    if (n < 4) return -1;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    std::vector<double> mn(2 * n);
    for (int i = 0; i < n; ++i) { mn[2 * i] = (U[2 * i] - cx) / fx; mn[2 * i + 1] = (U[2 * i + 1] - cy) / fy; }
    // ---- step 2: spread of the model
    double Mc[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) Mc[k] += P[3 * i + k] / n;
    double MM[9] = {0};
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) MM[a * 3 + b] += (P[3 * i + a] - Mc[a]) * (P[3 * i + b] - Mc[b]);
    double Vm[9], Wm[3];
    jacobi_eig(MM, 3, Vm, Wm);
    int ord[3] = {0, 1, 2};
    std::sort(ord, ord + 3, [&](int a, int b) { return Wm[a] > Wm[b]; });
    double prm[6];
    double R[9], t[3];
    if (!(Wm[ord[0]] > 0)) return -2;
    if (Wm[ord[2]] / std::max(Wm[ord[1]], 1e-300) < 1e-3) {
        // ---- planar model: R_transform = V^T (rows = principal directions, the plane normal last)
        if (n < 4) return -1;
        double Rt[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Rt[r * 3 + c] = Vm[c * 3 + ord[r]];
        if (Rt[6] * Rt[6] + Rt[7] * Rt[7] < 1e-10) {
            const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            std::memcpy(Rt, I, sizeof(I));
        }
        if (det3(Rt) < 0)
            for (int i = 0; i < 9; ++i) Rt[i] = -Rt[i];
        double Tt[3];
        for (int r = 0; r < 3; ++r) Tt[r] = -(Rt[r * 3] * Mc[0] + Rt[r * 3 + 1] * Mc[1] + Rt[r * 3 + 2] * Mc[2]);
        std::vector<double> xy(2 * n);
        for (int i = 0; i < n; ++i)
            for (int r = 0; r < 2; ++r)
                xy[2 * i + r] = Rt[r * 3] * P[3 * i] + Rt[r * 3 + 1] * P[3 * i + 1] + Rt[r * 3 + 2] * P[3 * i + 2] + Tt[r];
        double H[9];
        if (homography_dlt(xy.data(), mn.data(), n, H)) {
            double h1[3] = {H[0], H[3], H[6]}, h2[3] = {H[1], H[4], H[7]}, h3[3] = {H[2], H[5], H[8]};
            // a homography is defined up to sign: keep the plane in front of the camera
            const double zc = h3[2];
            if (zc < 0)
                for (int k = 0; k < 3; ++k) { h1[k] = -h1[k]; h2[k] = -h2[k]; h3[k] = -h3[k]; }
            const double n1 = norm_l2(h1, 3), n2 = norm_l2(h2, 3);
            if (!(n1 > 0) || !(n2 > 0)) return -2;
            for (int k = 0; k < 3; ++k) { h1[k] /= n1; h2[k] /= n2; t[k] = h3[k] * 2.0 / (n1 + n2); }
            const double hx[3] = {h1[1] * h2[2] - h1[2] * h2[1], h1[2] * h2[0] - h1[0] * h2[2], h1[0] * h2[1] - h1[1] * h2[0]};
            double Hm[9] = {h1[0], h2[0], hx[0], h1[1], h2[1], hx[1], h1[2], h2[2], hx[2]};
            double rv[3], Hp[9];
            polar_rotation(Hm, Hp);          // cvRodrigues2 orthonormalises a matrix input through its SVD
            rodrigues_log(Hp, rv);
            rodrigues_exp(rv, Hm);
            for (int r = 0; r < 3; ++r) t[r] += Hm[r * 3] * Tt[0] + Hm[r * 3 + 1] * Tt[1] + Hm[r * 3 + 2] * Tt[2];
            mul33(Hm, Rt, R);
        } else {
            const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            std::memcpy(R, I, sizeof(I));
            t[0] = t[1] = t[2] = 0;
        }
    } else {
        // ---- DLT on the raw object coordinates
        if (n < 6) return -1;
        double LL[144] = {0};
        for (int i = 0; i < n; ++i) {
            const double x = mn[2 * i], y = mn[2 * i + 1];
            const double X = P[3 * i], Y = P[3 * i + 1], Z = P[3 * i + 2];
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
        if (det3(RR) < 0) {
            for (int i = 0; i < 9; ++i) RR[i] = -RR[i];
            for (int i = 0; i < 3; ++i) tt[i] = -tt[i];
        }
        const double sc = norm_l2(RR, 9);
        if (!(sc > 0)) return -2;
        polar_rotation(RR, R);
        for (int i = 0; i < 3; ++i) t[i] = tt[i] * (std::sqrt(3.0) / sc);
    }
    rodrigues_log(R, prm);
    prm[3] = t[0]; prm[4] = t[1]; prm[5] = t[2];

    // ---- CvLevMarq
    const int ne = 2 * n;
    std::vector<double> err(ne), J((size_t)ne * 6);
    double prev[6], JtJ[36], JtErr[6];
    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = 0;
    auto step = [&]() {
        const double lambda = std::exp(lambdaLg10 * std::log(10.0));
        double A[36], d[6];
        std::memcpy(A, JtJ, sizeof(A));
        std::memcpy(d, JtErr, sizeof(d));
        for (int i = 0; i < 6; ++i) A[i * 6 + i] *= 1.0 + lambda;
        if (!solve_n(A, d, 6)) std::memset(d, 0, sizeof(d));
        for (int i = 0; i < 6; ++i) prm[i] = prev[i] - d[i];
    };
    for (;;) {
        // CALC_J at the current parameters
        project_residuals(P, U, n, K, prm, err.data(), J.data());
        std::memset(JtJ, 0, sizeof(JtJ));
        std::memset(JtErr, 0, sizeof(JtErr));
        for (int e = 0; e < ne; ++e)
            for (int a = 0; a < 6; ++a) {
                JtErr[a] += J[(size_t)e * 6 + a] * err[e];
                for (int b = 0; b < 6; ++b) JtJ[a * 6 + b] += J[(size_t)e * 6 + a] * J[(size_t)e * 6 + b];
            }
        std::memcpy(prev, prm, sizeof(prev));
        if (iters == 0) prevErrNorm = norm_l2(err.data(), ne);
        step();
        // CHECK_ERR
        double errNorm;
        for (;;) {
            project_residuals(P, U, n, K, prm, err.data(), nullptr);
            errNorm = norm_l2(err.data(), ne);
            if (errNorm > prevErrNorm && ++lambdaLg10 <= 16) { step(); continue; }
            break;
        }
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        double dn = 0;
        for (int i = 0; i < 6; ++i) dn += (prm[i] - prev[i]) * (prm[i] - prev[i]);
        const double rel = std::sqrt(dn) / std::max(norm_l2(prev, 6), DBL_MIN);
        if (++iters >= 20 || rel < (double)FLT_EPSILON) break;
        prevErrNorm = errNorm;
    }
    rodrigues_exp(prm, Rout);          // cv2.Rodrigues(R_exp)
    tout[0] = prm[3]; tout[1] = prm[4]; tout[2] = prm[5];
    for (int i = 0; i < 9; ++i)
        if (!std::isfinite(Rout[i])) return -2;
    for (int i = 0; i < 3; ++i)
        if (!std::isfinite(tout[i])) return -2;
    return 0;
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

// ---- the opt-in "refined" solver: Hartley-conditioned DLT (object points centred and scaled, which removes the
// near-null pure-translation vector that makes the raw DLT above flip on small, distant objects) and the same
// reprojection objective minimised to convergence (left-multiplicative LM).  Not what the reference calls; offered
// because the raw-DLT initialisation of SOLVEPNP_ITERATIVE lands in a wrong basin on a measurable share of noisy
// inputs.
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

int solve_pnp_refined(const double* P, const double* U, int n, const double* K, double* Rout, double* tout) {
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
            if (!solve_n(A, d, 6)) { lambda *= 10; continue; }
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


// kept name of the C-ABI's worker
int solve_pnp(const double* P, const double* U, int n, const double* K, double* Rout, double* tout) {
    return solve_pnp_iterative(P, U, n, K, Rout, tout);
}

int solve_pnp_ransac(const double* P, const double* U, int n, const double* K, double reproj_err, int max_trials,
                     double confidence, double* Rout, double* tout, unsigned char* inlier_mask) {
    const int MS = 6;                                   // sample size: the DLT initialiser needs 6 points
    if (n < MS) return -1;
    if (n == MS) {
        if (inlier_mask) std::memset(inlier_mask, 1, n);
        return solve_pnp_iterative(P, U, n, K, Rout, tout);
    }
    unsigned long long state = 0x9E3779B97F4A7C15ull;  // fixed seed: reproducible
    auto rnd = [&](int m) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        return (int)((state >> 33) % (unsigned long long)m);
    };
    std::vector<unsigned char> best(n, 0), cur(n);
    int best_cnt = 0, trials = std::max(1, max_trials);
    std::vector<double> sp(3 * MS), su(2 * MS), err(2 * n);
    double R[9], t[3], prm[6];
    for (int it = 0; it < trials; ++it) {
        int idx[MS];
        for (int k = 0; k < MS;) {
            const int c = rnd(n);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup |= idx[j] == c;
            if (!dup) idx[k++] = c;
        }
        for (int k = 0; k < MS; ++k) {
            std::memcpy(&sp[3 * k], P + 3 * idx[k], 3 * sizeof(double));
            std::memcpy(&su[2 * k], U + 2 * idx[k], 2 * sizeof(double));
        }
        if (solve_pnp_iterative(sp.data(), su.data(), MS, K, R, t) != 0) continue;
        rodrigues_log(R, prm);
        prm[3] = t[0]; prm[4] = t[1]; prm[5] = t[2];
        project_residuals(P, U, n, K, prm, err.data(), nullptr);
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            cur[i] = err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1] <= reproj_err * reproj_err;
            cnt += cur[i];
        }
        if (cnt > best_cnt) {
            best_cnt = cnt;
            best = cur;
            // RANSACUpdateNumIters: trials needed to draw one all-inlier sample at the requested confidence
            const double ep = 1.0 - (double)cnt / n;
            const double num = std::log(std::max(1.0 - confidence, DBL_MIN));
            const double den = std::log(std::max(1.0 - std::pow(1.0 - ep, MS), DBL_MIN));
            if (den < 0 && num / den < trials) trials = std::max(it + 1, (int)std::ceil(num / den));
        }
    }
    if (best_cnt < MS) return -2;
    std::vector<double> ip, iu;
    for (int i = 0; i < n; ++i)
        if (best[i]) {
            ip.insert(ip.end(), P + 3 * i, P + 3 * i + 3);
            iu.insert(iu.end(), U + 2 * i, U + 2 * i + 2);
        }
    if (inlier_mask) std::memcpy(inlier_mask, best.data(), n);
    return solve_pnp_iterative(ip.data(), iu.data(), best_cnt, K, Rout, tout);
}

// ------------------------------------------------------------------------------------------------ pose NMS (f32)
// pPose_nms.py:13-20
static const float kDelta1 = 1.f, kMu = 1.7f, kDelta2 = 2.65f, kGamma = 22.48f;
static const float kScoreThreds = 0.3f, kAreaThres = 0.f, kAlpha = 0.1f;
static const int kMatchThreds = 5;

// bboxes [n][4], bbox_scores [n], preds [n][K][2], scores [n][K]  ->  up to n merged poses:
// out_pick [m] (index of the kept candidate), out_pose [m][K][2] (already - 0.3), out_score [m][K], out_prop [m];
// returns m.  Greedy cluster (pPose_nms.py:44-70), weighted merge (:204-237), filters (:85-110).
int pose_nms(const float* bboxes, const float* bbox_scores, const float* preds, const float* scores_in, int n, int K,
             int* out_pick, float* out_pose, float* out_score, float* out_prop) {
    if (n <= 0 || K <= 0) return 0;
    std::vector<float> scores(scores_in, scores_in + (size_t)n * K);
    for (float& s : scores)
        if (s == 0.f) s = 1e-5f;                                            // pPose_nms.py:33
    std::vector<float> ref_dists(n), human(n);
    for (int i = 0; i < n; ++i) {
        const float w = bboxes[4 * i + 2] - bboxes[4 * i], h = bboxes[4 * i + 3] - bboxes[4 * i + 1];
        ref_dists[i] = kAlpha * std::max(w, h);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += scores[(size_t)i * K + k];
        human[i] = s / (float)K;
    }
    auto dist = [&](int a, int b, int k) {
        const float dx = preds[((size_t)a * K + k) * 2] - preds[((size_t)b * K + k) * 2];
        const float dy = preds[((size_t)a * K + k) * 2 + 1] - preds[((size_t)b * K + k) * 2 + 1];
        return std::sqrt(dx * dx + dy * dy);
    };
    std::vector<int> ids(n);
    for (int i = 0; i < n; ++i) ids[i] = i;
    std::vector<int> pick;
    std::vector<std::vector<int>> merge_ids;
    if (n == 1) {
        pick.push_back(0);
        merge_ids.push_back({0});
    } else {
        while (!ids.empty()) {
            int pid = 0;                                                    // first maximum, as torch.argmax on ties
            for (int j = 1; j < (int)ids.size(); ++j)
                if (human[ids[j]] > human[ids[pid]]) pid = j;
            const int ref = ids[pid];
            pick.push_back(ref);
            const float ref_dist = ref_dists[ref];
            std::vector<int> dele, keep;
            for (int j = 0; j < (int)ids.size(); ++j) {
                const int c = ids[j];
                float sd = 0.f, ex = 0.f;
                int nmatch = 0;
                for (int k = 0; k < K; ++k) {
                    const float d = dist(ref, c, k);
                    if (d <= 1.f) sd += std::tanh(scores[(size_t)ref * K + k] / kDelta1) * std::tanh(scores[(size_t)c * K + k] / kDelta1);
                    ex += std::exp(-d / kDelta2);
                    nmatch += (d / std::min(ref_dist, 7.f)) <= 1.f;
                }
                const float simi = sd + kMu * ex;
                if (simi > kGamma || nmatch >= kMatchThreds) dele.push_back(j);
                else keep.push_back(j);
            }
            if (dele.empty()) {                                             // pPose_nms.py:63-64
                dele.push_back(pid);
                keep.clear();
                for (int j = 0; j < (int)ids.size(); ++j)
                    if (j != pid) keep.push_back(j);
            }
            std::vector<int> m;
            for (int j : dele) m.push_back(ids[j]);
            merge_ids.push_back(m);
            std::vector<int> next;
            for (int j : keep) next.push_back(ids[j]);
            ids.swap(next);
        }
    }
    int m = 0;
    std::vector<float> mp((size_t)K * 2), ms(K);
    for (size_t j = 0; j < pick.size(); ++j) {
        const int pk = pick[j];
        float mx = -HUGE_VALF;
        for (int k = 0; k < K; ++k) mx = std::max(mx, scores[(size_t)pk * K + k]);
        if (mx < kScoreThreds) continue;
        if (n == 1) {
            for (int k = 0; k < K; ++k) {
                mp[2 * k] = preds[2 * k]; mp[2 * k + 1] = preds[2 * k + 1];
                ms[k] = scores[k];
            }
        } else {
            const std::vector<int>& mid = merge_ids[j];
            const float lim = std::min(ref_dists[pk], 15.f);
            for (int k = 0; k < K; ++k) {
                float wsum = 0.f;
                for (int c : mid)
                    if (dist(pk, c, k) <= lim) wsum += scores[(size_t)c * K + k];
                float px = 0.f, py = 0.f, sc = 0.f;
                for (int c : mid) {
                    const float msk = dist(pk, c, k) <= lim ? scores[(size_t)c * K + k] : 0.f;
                    const float nw = msk / wsum;
                    px += preds[((size_t)c * K + k) * 2] * nw;
                    py += preds[((size_t)c * K + k) * 2 + 1] * nw;
                    sc += msk * nw;
                }
                mp[2 * k] = px; mp[2 * k + 1] = py; ms[k] = sc;
            }
        }
        float smax = -HUGE_VALF, ssum = 0.f, xmin = HUGE_VALF, xmax = -HUGE_VALF, ymin = HUGE_VALF, ymax = -HUGE_VALF;
        for (int k = 0; k < K; ++k) {
            smax = std::max(smax, ms[k]); ssum += ms[k];
            xmin = std::min(xmin, mp[2 * k]); xmax = std::max(xmax, mp[2 * k]);
            ymin = std::min(ymin, mp[2 * k + 1]); ymax = std::max(ymax, mp[2 * k + 1]);
        }
        if (smax < kScoreThreds) continue;
        if (1.5f * 1.5f * (xmax - xmin) * (ymax - ymin) < kAreaThres) continue;
        out_pick[m] = pk;
        for (int k = 0; k < K; ++k) {
            out_pose[((size_t)m * K + k) * 2] = mp[2 * k] - 0.3f;
            out_pose[((size_t)m * K + k) * 2 + 1] = mp[2 * k + 1] - 0.3f;
            out_score[(size_t)m * K + k] = ms[k];
        }
        out_prop[m] = ssum / (float)K + bbox_scores[pk] + 1.25f * smax;
        ++m;
    }
    return m;
}

}  // namespace bp
