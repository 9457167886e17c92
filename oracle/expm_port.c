/* expm_port.c - plain-C restatement of the reference's per-sample path  --  TEST / BASELINE INFRASTRUCTURE ONLY.
 *
 * Follows the reference's ALGORITHM SHAPE sample by sample (noise_model.py:98-109): assemble the dense complex
 * N x N Hamiltonian  H = HH + Z + diag(x[:N])  (HH: noise_model.py:79-85, Z: noise_model.py:122-147), form
 * U = expm(-i T H), return |U[out,in]|^2.  The matrix exponential lives in a third-party dependency of the
 * reference (scipy.linalg.expm, pinned scipy==1.7.1, call site noise_model.py:105) that is not under
 * /root/reference; what is restated here is its PUBLISHED algorithm: Pade approximants of order 3/5/7/9/13 with
 * scaling and squaring, thresholds theta_m and coefficients from N. J. Higham, "The scaling and squaring method
 * for the matrix exponential revisited", SIAM J. Matrix Anal. Appl. 26(4), 2005 (1-norm based order selection).
 *
 * Used by: tests/test_oracle_c_port.py (cross-check against oracle/robchar_oracle.py and the golden vectors) and
 * bench.py's cpu_baseline leg.  The product (code-robchar_amd/) never loads it.
 * Build: make -C oracle   ->  oracle/librc_oracle_port.so
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NMAX 16
typedef double complex cplx;

static void matmul(int n, const cplx* A, const cplx* B, cplx* C) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            cplx s = 0;
            for (int k = 0; k < n; ++k) s += A[i * n + k] * B[k * n + j];
            C[i * n + j] = s;
        }
}

static double norm1(int n, const cplx* A) {
    double best = 0;
    for (int j = 0; j < n; ++j) {
        double s = 0;
        for (int i = 0; i < n; ++i) s += cabs(A[i * n + j]);
        if (s > best) best = s;
    }
    return best;
}

/* solve P X = Q in place of Q (Gaussian elimination, partial pivoting) */
static void solve(int n, cplx* P, cplx* Q) {
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (cabs(P[r * n + c]) > cabs(P[piv * n + c])) piv = r;
        if (piv != c)
            for (int j = 0; j < n; ++j) {
                cplx t = P[c * n + j]; P[c * n + j] = P[piv * n + j]; P[piv * n + j] = t;
                t = Q[c * n + j]; Q[c * n + j] = Q[piv * n + j]; Q[piv * n + j] = t;
            }
        const cplx inv = 1.0 / P[c * n + c];
        for (int r = c + 1; r < n; ++r) {
            const cplx f = P[r * n + c] * inv;
            if (f == 0) continue;
            for (int j = c; j < n; ++j) P[r * n + j] -= f * P[c * n + j];
            for (int j = 0; j < n; ++j) Q[r * n + j] -= f * Q[c * n + j];
        }
    }
    for (int c = n - 1; c >= 0; --c) {
        const cplx inv = 1.0 / P[c * n + c];
        for (int j = 0; j < n; ++j) {
            cplx s = Q[c * n + j];
            for (int k = c + 1; k < n; ++k) s -= P[c * n + k] * Q[k * n + j];
            Q[c * n + j] = s * inv;
        }
    }
}

static const double B3[] = {120, 60, 12, 1};
static const double B5[] = {30240, 15120, 3360, 420, 30, 1};
static const double B7[] = {17297280, 8648640, 1995840, 277200, 25200, 1512, 56, 1};
static const double B9[] = {17643225600., 8821612800., 2075673600., 302702400., 30270240., 2162160., 110880., 3960., 90., 1.};
static const double B13[] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                             129060195264000., 10559470521600., 670442572800., 33522128640., 1323241920.,
                             40840800., 960960., 16380., 182., 1.};

/* E = expm(A), A is n x n row-major */
static void expm(int n, const cplx* Ain, cplx* E) {
    cplx A[NMAX * NMAX], A2[NMAX * NMAX], A4[NMAX * NMAX], A6[NMAX * NMAX], A8[NMAX * NMAX];
    cplx U[NMAX * NMAX], V[NMAX * NMAX], W[NMAX * NMAX], W2[NMAX * NMAX];
    const int nn = n * n;
    memcpy(A, Ain, sizeof(cplx) * nn);
    const double nrm = norm1(n, A);
    if (!(nrm <= 1.0e300)) {                    /* an infinite / NaN entry: no exponential (and no (int)ceil(log2(inf)) squarings) */
        for (int i = 0; i < nn; ++i) E[i] = NAN;
        return;
    }
    int s = 0, m;
    if (nrm <= 1.495585217958292e-2) m = 3;
    else if (nrm <= 2.539398330063230e-1) m = 5;
    else if (nrm <= 9.504178996162932e-1) m = 7;
    else if (nrm <= 2.097847961257068e0) m = 9;
    else {
        m = 13;
        const double theta13 = 5.371920351148152e0;
        if (nrm > theta13) {
            s = (int)ceil(log2(nrm / theta13));
            const double sc = ldexp(1.0, -s);
            for (int i = 0; i < nn; ++i) A[i] *= sc;
        }
    }
    matmul(n, A, A, A2);
    if (m == 13) {
        matmul(n, A2, A2, A4);
        matmul(n, A4, A2, A6);
        for (int i = 0; i < nn; ++i) W[i] = B13[13] * A6[i] + B13[11] * A4[i] + B13[9] * A2[i];
        matmul(n, A6, W, W2);
        for (int i = 0; i < nn; ++i) W2[i] += B13[7] * A6[i] + B13[5] * A4[i] + B13[3] * A2[i];
        for (int i = 0; i < n; ++i) W2[i * n + i] += B13[1];
        matmul(n, A, W2, U);
        for (int i = 0; i < nn; ++i) W[i] = B13[12] * A6[i] + B13[10] * A4[i] + B13[8] * A2[i];
        matmul(n, A6, W, V);
        for (int i = 0; i < nn; ++i) V[i] += B13[6] * A6[i] + B13[4] * A4[i] + B13[2] * A2[i];
        for (int i = 0; i < n; ++i) V[i * n + i] += B13[0];
    } else {
        const double* b = (m == 3) ? B3 : (m == 5) ? B5 : (m == 7) ? B7 : B9;
        if (m >= 5) matmul(n, A2, A2, A4);
        if (m >= 7) matmul(n, A4, A2, A6);
        if (m >= 9) matmul(n, A6, A2, A8);
        for (int i = 0; i < nn; ++i) {
            cplx u = b[3] * A2[i], v = b[2] * A2[i];
            if (m >= 5) { u += b[5] * A4[i]; v += b[4] * A4[i]; }
            if (m >= 7) { u += b[7] * A6[i]; v += b[6] * A6[i]; }
            if (m >= 9) { u += b[9] * A8[i]; v += b[8] * A8[i]; }
            W[i] = u; V[i] = v;
        }
        for (int i = 0; i < n; ++i) { W[i * n + i] += b[1]; V[i * n + i] += b[0]; }
        matmul(n, A, W, U);
    }
    /* (V - U) X = (V + U) */
    for (int i = 0; i < nn; ++i) { W[i] = V[i] - U[i]; E[i] = V[i] + U[i]; }
    solve(n, W, E);
    for (int k = 0; k < s; ++k) {
        matmul(n, E, E, W);
        memcpy(E, W, sizeof(cplx) * nn);
    }
}

/* fid[c][k] = | expm(-i |x_N| (HH + Z(draws[c][k]) + diag(x)))[out,in] |^2 ; NaN controller rows give NaN */
int rc_oracle_expm_fidelity(int N, int in, int out, const double* h0_diag, const double* h0_offdiag, int ring,
                            const double* ctrl, const double* draws, long long C, long long K, double* fid,
                            int nthreads) {
    if (N < 2 || N > NMAX || in < 0 || in >= N || out < 0 || out >= N) return -1;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (long long c = 0; c < C; ++c)
        for (long long k = 0; k < K; ++k) {
            const double* x = ctrl + c * (N + 1);
            int bad = 0;
            for (int i = 0; i <= N; ++i) bad |= (x[i] != x[i]);
            if (bad) { fid[c * K + k] = NAN; continue; }
            const double* g = draws ? draws + (c * K + k) * 3 * N : 0;
            cplx H[NMAX * NMAX], E[NMAX * NMAX];
            memset(H, 0, sizeof(cplx) * N * N);
            for (int i = 0; i < N; ++i)
                H[i * N + i] = x[i] + (h0_diag ? h0_diag[i] : 0.0) + (g ? g[3 * i] : 0.0);
            for (int i = 1; i < N; ++i) {
                const double J = h0_offdiag ? h0_offdiag[i - 1] : 1.0;
                const double re = J + (g ? g[3 * i + 1] : 0.0), im = g ? g[3 * i + 2] : 0.0;
                H[i * N + i - 1] = re + I * im;        /* Z[i][i-1] = nn + 1j*nn2, noise_model.py:141 */
                H[(i - 1) * N + i] = re - I * im;
            }
            if (ring) { H[(N - 1) * N] += 1.0; H[N - 1] += 1.0; }
            const double T = fabs(x[N]);
            for (int i = 0; i < N * N; ++i) H[i] *= -I * T;
            expm(N, H, E);
            const cplx phi = E[out * N + in];
            fid[c * K + k] = creal(phi) * creal(phi) + cimag(phi) * cimag(phi);
        }
    return 0;
}

int rc_oracle_max_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
