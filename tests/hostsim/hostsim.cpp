// TEST INFRASTRUCTURE ONLY (CPU).  Compiles sympa_amd/csrc/siegel_math.hpp with g++ so the exact
// per-pair arithmetic the gfx950 kernels run can be checked against the oracle and the golden
// vectors in a container without a GPU.  Nothing in sympa_amd/ loads this library; the product
// path fails loudly when the HIP library is missing.
#include <cmath>
#include <cstdint>
#include "../../sympa_amd/csrc/siegel_math.hpp"
#include "../../sympa_amd/csrc/siegel_math_bwd.hpp"
#include "../../sympa_amd/csrc/siegel_math_bwd_split.hpp"
#include "../../sympa_amd/csrc/siegel_table_math.hpp"
#include "../../sympa_amd/csrc/siegel_math_generic.hpp"
#include "../../sympa_amd/csrc/spd_math.hpp"
#include "../../sympa_amd/csrc/spd_math_bwd.hpp"
#include "../../sympa_amd/csrc/tridiag_invit.hpp"

namespace {
template <int N>
void run(const double* z1, const double* z2, int64_t b, int model, int metric, const double* w, double eps,
         double* out, double* vvd, int32_t* status) {
    int st = 0;
    for (int64_t i = 0; i < b; ++i) {
        const double* p1 = z1 + i * 2 * N * N;
        const double* p2 = z2 + i * 2 * N * N;
        double* vv = vvd ? vvd + i * N : nullptr;
        out[i] = (model == sympa::MODEL_UPPER)
                     ? sympa::pair_distance<N, sympa::MODEL_UPPER>(p1, p2, metric, w, 1.0 / eps, vv, st)
                     : sympa::pair_distance<N, sympa::MODEL_BOUNDED>(p1, p2, metric, w, 1.0 / eps, vv, st);
    }
    if (status) *status = st;
}
}  // namespace

extern "C" int sympa_hostsim_dist(const double* z1, const double* z2, int64_t b, int n, int model, int metric,
                                  const double* w, double eps, double* out, double* vvd, int32_t* status) {
    switch (n) {
        case 1: run<1>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 2: run<2>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 3: run<3>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 4: run<4>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 5: run<5>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 6: run<6>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 7: run<7>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        case 8: run<8>(z1, z2, b, model, metric, w, eps, out, vvd, status); return 0;
        default: return -2;
    }
}

namespace {
template <int N, int MODEL>
void run_packed(const double* z1, const double* z2, int64_t b, int metric, const double* w, double eps, double* out,
                int32_t* status) {
    int st = 0;
    for (int64_t i = 0; i < b; ++i) {
        sympa::CMat<N> a, c, e;
        sympa::load_point<N>(z1 + i * 2 * N * N, a);
        sympa::load_point<N>(z2 + i * 2 * N * N, c);
        double p1[sympa::PointPack<N, MODEL>::LEN], p2[sympa::PointPack<N, MODEL>::LEN];
        const bool ok1 = sympa::pack_point<N, MODEL>(a, p1);
        const bool ok2 = sympa::pack_point<N, MODEL>(c, p2);
        sympa::e_from_packed<N, MODEL>(p1, p2, e);
        out[i] = sympa::distance_from_e<N, MODEL>(e, ok1 && ok2, metric, w, 1.0 / eps, nullptr, st);
    }
    if (status) *status = st;
}
// the INDEXED packed forward's per-pair path (csrc/siegel_packed_kernel.hpp): the second point's packed row whole, the first
// point's triangles subtracted from it in place, only the first point's factor kept; e_from_packed<DIFF> + distance_from_h
template <int N, int MODEL>
void run_packed_diff(const double* z1, const double* z2, int64_t b, int metric, const double* w, double eps, double* out,
                     int32_t* status) {
    using P = sympa::PointPack<N, MODEL>;
    struct FactorOnly {
        const double* a;
        double operator[](int k) const { return a[k - 2 * P::TRI]; }
    };
    for (int64_t i = 0; i < b; ++i) {
        sympa::CMat<N> a, c, e;
        sympa::load_point<N>(z1 + i * 2 * N * N, a);
        sympa::load_point<N>(z2 + i * 2 * N * N, c);
        double p1[P::LEN], p2[P::LEN];
        const bool ok1 = sympa::pack_point<N, MODEL>(a, p1);
        const bool ok2 = sympa::pack_point<N, MODEL>(c, p2);
        for (int k = 0; k < 2 * P::TRI; ++k) p2[k] -= p1[k];
        const FactorOnly f{p1 + 2 * P::TRI};
        sympa::e_from_packed<N, MODEL, true>(f, p2, e);
        sympa::Herm<N> h;
        sympa::gram<N>(e, h);
        int st = 0;
        out[i] = sympa::distance_from_h<N, MODEL>(h, ok1 && ok2, metric, w, 1.0 / eps, nullptr, st);
        if (status) status[0] |= st;
    }
}

template <int N>
void run_packed_n(const double* z1, const double* z2, int64_t b, int model, int metric, const double* w, double eps,
                  double* out, int32_t* status) {
    // metric ids >= 16: the same through the in-place difference form of the indexed packed forward
    if (metric >= 16) {
        if (model == sympa::MODEL_UPPER) run_packed_diff<N, sympa::MODEL_UPPER>(z1, z2, b, metric - 16, w, eps, out, status);
        else run_packed_diff<N, sympa::MODEL_BOUNDED>(z1, z2, b, metric - 16, w, eps, out, status);
        return;
    }
    if (model == sympa::MODEL_UPPER) run_packed<N, sympa::MODEL_UPPER>(z1, z2, b, metric, w, eps, out, status);
    else run_packed<N, sympa::MODEL_BOUNDED>(z1, z2, b, metric, w, eps, out, status);
}
}  // namespace

// the all-pairs kernel's per-pair path: both points packed (inverted factor), E = A1 (Z2 - Z1) A2^T
extern "C" int sympa_hostsim_dist_packed(const double* z1, const double* z2, int64_t b, int n, int model, int metric,
                                         const double* w, double eps, double* out, int32_t* status) {
    switch (n) {
        case 1: run_packed_n<1>(z1, z2, b, model, metric, w, eps, out, status); return 0;
        case 2: run_packed_n<2>(z1, z2, b, model, metric, w, eps, out, status); return 0;
        case 3: run_packed_n<3>(z1, z2, b, model, metric, w, eps, out, status); return 0;
        case 4: run_packed_n<4>(z1, z2, b, model, metric, w, eps, out, status); return 0;
        case 5: run_packed_n<5>(z1, z2, b, model, metric, w, eps, out, status); return 0;
        case 6: run_packed_n<6>(z1, z2, b, model, metric, w, eps, out, status); return 0;
        case 7: run_packed_n<7>(z1, z2, b, model, metric, w, eps, out, status); return 0;
        case 8: run_packed_n<8>(z1, z2, b, model, metric, w, eps, out, status); return 0;
        default: return -2;
    }
}

extern "C" int sympa_hostsim_dist_generic(const double* z1, const double* z2, int64_t b, int n, int model, int metric,
                                          const double* w, double eps, double* out, double* vvd, int32_t* status) {
    if (n < 1 || n > sympa::GENERIC_MAX_N) return -2;
    int st = 0;
    sympa::GenericWork work;
    for (int64_t i = 0; i < b; ++i)
        out[i] = sympa::pair_distance_generic(work, z1 + i * 2 * n * n, z2 + i * 2 * n * n, n, model, metric, w, 1.0 / eps,
                                              vvd ? vvd + i * n : nullptr, st);
    if (status) *status = st;
    return 0;
}

namespace {
template <int N>
void run_bwd(const double* z1, const double* z2, const double* go, int64_t b, int model, int metric, const double* w,
             double eps, double* out, double* g1, double* g2, double* gw, int32_t* status) {
    int st = 0;
    double gwacc[N];
    for (int k = 0; k < N; ++k) gwacc[k] = 0.0;
    for (int64_t i = 0; i < b; ++i) {
        sympa::CMat<N> a, c, ga, gc;
        sympa::load_point<N>(z1 + i * 2 * N * N, a);
        sympa::load_point<N>(z2 + i * 2 * N * N, c);
        out[i] = (model == sympa::MODEL_UPPER)
                     ? sympa::pair_backward<N, sympa::MODEL_UPPER>(a, c, metric, w, 1.0 / eps, go[i], ga, gc, gwacc, st)
                     : sympa::pair_backward<N, sympa::MODEL_BOUNDED>(a, c, metric, w, 1.0 / eps, go[i], ga, gc, gwacc, st);
        for (int r = 0; r < N; ++r)
            for (int s = 0; s < N; ++s) {
                g1[i * 2 * N * N + r * N + s] = ga.re[r][s];
                g1[i * 2 * N * N + N * N + r * N + s] = ga.im[r][s];
                g2[i * 2 * N * N + r * N + s] = gc.re[r][s];
                g2[i * 2 * N * N + N * N + r * N + s] = gc.im[r][s];
            }
    }
    for (int k = 0; k < N; ++k) gw[k] = gwacc[k];
    if (status) *status = st;
}
}  // namespace

extern "C" int sympa_hostsim_dist_bwd(const double* z1, const double* z2, const double* go, int64_t b, int n, int model,
                                      int metric, const double* w, double eps, double* out, double* g1, double* g2,
                                      double* gw, int32_t* status) {
    switch (n) {
        case 1: run_bwd<1>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 2: run_bwd<2>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 3: run_bwd<3>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 4: run_bwd<4>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 5: run_bwd<5>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 6: run_bwd<6>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 7: run_bwd<7>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 8: run_bwd<8>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        default: return -2;
    }
}

namespace {
// the two-stage adjoint of dims 5..8 (siegel_math_bwd_split.hpp): stage 1 -> pack (scaled by go) -> stage 2
template <int N, int MODEL>
void run_bwd_split_m(const double* z1, const double* z2, const double* go, int64_t b, int metric, const double* w, double eps,
                     double* out, double* g1, double* g2, double* gw, int32_t* status) {
    int st = 0;
    double gwacc[N];
    for (int k = 0; k < N; ++k) gwacc[k] = 0.0;
    for (int64_t i = 0; i < b; ++i) {
        sympa::CMat<N> a, c, ga, gc;
        sympa::load_point<N>(z1 + i * 2 * N * N, a);
        sympa::load_point<N>(z2 + i * 2 * N * N, c);
        double pack[sympa::AdjPack<N, MODEL>::LEN], gwl[N];
        for (int k = 0; k < N; ++k) gwl[k] = 0.0;
        out[i] = sympa::pair_adjoint_spectral<N, MODEL>(a, c, metric, w, 1.0 / eps, pack, gwl, st);
        for (int k = 0; k < sympa::AdjPack<N, MODEL>::LEN; ++k) pack[k] *= go[i];
        for (int k = 0; k < N; ++k) gwacc[k] += gwl[k] * go[i];
        if constexpr (MODEL == sympa::MODEL_UPPER) {
            // the register-ordered form the upper-model kernel runs (planes emitted one at a time, upper triangles)
            double gstash[N * (N + 1) / 2], staged[N][N];
            sympa::Tri<N, false> parked[2];
            sympa::pair_adjoint_gradient_upper<N>(
                a, c, [&](int k) { return pack[k]; },
                [&](int which, const sympa::Tri<N, false>& l) { parked[which] = l; },
                [&](int which, sympa::Tri<N, false>& l) { l = parked[which]; },
                [&](int k, double g) { gstash[k] = g; }, [&](int k) { return gstash[k]; },
                [&](const double (&m)[N][N]) {
                    for (int r = 0; r < N; ++r)
                        for (int s = r; s < N; ++s) staged[r][s] = m[r][s];
                },
                [&](int point, int plane, double sign) {
                    sympa::CMat<N>& g = point == 0 ? ga : gc;
                    for (int r = 0; r < N; ++r)
                        for (int s = r; s < N; ++s) {
                            if (plane == 0) { g.re[r][s] = sign * staged[r][s]; g.re[s][r] = sign * staged[r][s]; }
                            else { g.im[r][s] = sign * staged[r][s]; g.im[s][r] = sign * staged[r][s]; }
                        }
                });
        } else {
            sympa::pair_adjoint_gradient<N, MODEL>(a, c, pack, ga, gc);
        }
        for (int r = 0; r < N; ++r)
            for (int s = 0; s < N; ++s) {
                g1[i * 2 * N * N + r * N + s] = ga.re[r][s];
                g1[i * 2 * N * N + N * N + r * N + s] = ga.im[r][s];
                g2[i * 2 * N * N + r * N + s] = gc.re[r][s];
                g2[i * 2 * N * N + N * N + r * N + s] = gc.im[r][s];
            }
    }
    for (int k = 0; k < N; ++k) gw[k] = gwacc[k];
    if (status) *status = st;
}
template <int N>
void run_bwd_split(const double* z1, const double* z2, const double* go, int64_t b, int model, int metric, const double* w,
                   double eps, double* out, double* g1, double* g2, double* gw, int32_t* status) {
    if (model == sympa::MODEL_UPPER) run_bwd_split_m<N, sympa::MODEL_UPPER>(z1, z2, go, b, metric, w, eps, out, g1, g2, gw, status);
    else run_bwd_split_m<N, sympa::MODEL_BOUNDED>(z1, z2, go, b, metric, w, eps, out, g1, g2, gw, status);
}
}  // namespace

extern "C" int sympa_hostsim_dist_bwd_split(const double* z1, const double* z2, const double* go, int64_t b, int n, int model,
                                            int metric, const double* w, double eps, double* out, double* g1, double* g2,
                                            double* gw, int32_t* status) {
    switch (n) {
        case 2: run_bwd_split<2>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 3: run_bwd_split<3>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 4: run_bwd_split<4>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 5: run_bwd_split<5>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 6: run_bwd_split<6>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 7: run_bwd_split<7>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        case 8: run_bwd_split<8>(z1, z2, go, b, model, metric, w, eps, out, g1, g2, gw, status); return 0;
        default: return -2;
    }
}

namespace {
// both eigenvector routes of stage 1 on H = E^H E of upper-model pairs: err[0] = max |Hbar_ql - Hbar_invit| / max |Hbar_ql| for the
// spectral function phi = 1 / (1 + lambda), err[1] = max |V^H V - I| of the inverse-iteration route, err[2] = max |lambda_ql - lambda_invit| (sorted) / lambda_max
template <int N>
void run_eig_routes(const double* z1, const double* z2, int64_t b, double* err) {
    err[0] = err[1] = err[2] = 0.0;
    for (int64_t p = 0; p < b; ++p) {
        sympa::CMat<N> a, c, e;
        sympa::load_point<N>(z1 + p * 2 * N * N, a);
        sympa::load_point<N>(z2 + p * 2 * N * N, c);
        sympa::Tri<N, false> l1, l2;
        sympa::chol_real<N>(a.im, l1);
        sympa::chol_real<N>(c.im, l2);
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) { e.re[i][j] = c.re[i][j] - a.re[i][j]; e.im[i][j] = c.im[i][j] - a.im[i][j]; }
        sympa::solve_left<N, false>(l1, e);
        sympa::solve_right_t<N, false>(l2, e);
        sympa::Herm<N> h1, h2;
        sympa::gram<N>(e, h1);
        h2 = h1;
        sympa::CMat<N> v1, v2, hb1, hb2;
        sympa::herm_eigen_vectors_ql<N>(h1, v1);
        sympa::herm_eigen_vectors_invit<N>(h2, v2);
        double p1[N], p2[N], l1s[N], l2s[N], lmax = 1e-300;
        for (int i = 0; i < N; ++i) { p1[i] = 1.0 / (1.0 + h1.d[i]); p2[i] = 1.0 / (1.0 + h2.d[i]); l1s[i] = h1.d[i]; l2s[i] = h2.d[i]; }
        sympa::sort_ascending<N>(l1s);
        sympa::sort_ascending<N>(l2s);
        for (int i = 0; i < N; ++i) lmax = std::fmax(lmax, std::fabs(l1s[i]));
        for (int i = 0; i < N; ++i) err[2] = std::fmax(err[2], std::fabs(l1s[i] - l2s[i]) / lmax);
        sympa::herm_from_eig<N>(v1, p1, hb1);
        sympa::herm_from_eig<N>(v2, p2, hb2);
        double mx = 1e-300, df = 0.0;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                mx = std::fmax(mx, std::fmax(std::fabs(hb1.re[i][j]), std::fabs(hb1.im[i][j])));
                df = std::fmax(df, std::fmax(std::fabs(hb1.re[i][j] - hb2.re[i][j]), std::fabs(hb1.im[i][j] - hb2.im[i][j])));
            }
        err[0] = std::fmax(err[0], df / mx);
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                double tr = 0.0, ti = 0.0;
                for (int k = 0; k < N; ++k) {
                    tr += v2.re[k][i] * v2.re[k][j] + v2.im[k][i] * v2.im[k][j];
                    ti += v2.re[k][i] * v2.im[k][j] - v2.im[k][i] * v2.re[k][j];
                }
                err[1] = std::fmax(err[1], std::fmax(std::fabs(tr - (i == j ? 1.0 : 0.0)), std::fabs(ti)));
            }
    }
}
}  // namespace

extern "C" int sympa_hostsim_eig_routes(const double* z1, const double* z2, int64_t b, int n, double* err) {
    switch (n) {
        case 5: run_eig_routes<5>(z1, z2, b, err); return 0;
        case 6: run_eig_routes<6>(z1, z2, b, err); return 0;
        case 7: run_eig_routes<7>(z1, z2, b, err); return 0;
        case 8: run_eig_routes<8>(z1, z2, b, err); return 0;
        default: return -2;
    }
}

namespace {
template <int N>
int run_table(int op, int model, const double* z, const double* g, double* out, int64_t b, double lr, double wd,
              double eps, int32_t* projected) {
    int st = 0, moved = 0;
    for (int64_t i = 0; i < b; ++i) {
        sympa::CMat<N> a, gg, r;
        sympa::load_full<N>(z + i * 2 * N * N, a);
        if (g) sympa::load_full<N>(g + i * 2 * N * N, gg);
        if (op == 2) {
            if (model == 0) sympa::egrad2rgrad<N, sympa::MODEL_UPPER>(a, gg, r); else sympa::egrad2rgrad<N, sympa::MODEL_BOUNDED>(a, gg, r);
            sympa::store_full<N>(out + i * 2 * N * N, r);
        } else {
            bool m;
            if (op == 0) m = (model == 0) ? sympa::projx<N, sympa::MODEL_UPPER>(a, eps, st) : sympa::projx<N, sympa::MODEL_BOUNDED>(a, eps, st);
            else m = (model == 0) ? sympa::rsgd_row<N, sympa::MODEL_UPPER>(a, gg, lr, wd, eps, st) : sympa::rsgd_row<N, sympa::MODEL_BOUNDED>(a, gg, lr, wd, eps, st);
            moved += m ? 1 : 0;
            sympa::store_full<N>(out + i * 2 * N * N, a);
        }
    }
    if (projected) *projected = moved;
    return st;
}
}  // namespace

// op: 0 projx, 1 rsgd step (out = new rows), 2 egrad2rgrad
extern "C" int sympa_hostsim_table(int op, int model, int n, const double* z, const double* g, double* out, int64_t b,
                                   double lr, double wd, double eps, int32_t* projected) {
    switch (n) {
        case 1: return run_table<1>(op, model, z, g, out, b, lr, wd, eps, projected);
        case 2: return run_table<2>(op, model, z, g, out, b, lr, wd, eps, projected);
        case 3: return run_table<3>(op, model, z, g, out, b, lr, wd, eps, projected);
        case 4: return run_table<4>(op, model, z, g, out, b, lr, wd, eps, projected);
        case 5: return run_table<5>(op, model, z, g, out, b, lr, wd, eps, projected);
        case 6: return run_table<6>(op, model, z, g, out, b, lr, wd, eps, projected);
        case 7: return run_table<7>(op, model, z, g, out, b, lr, wd, eps, projected);
        case 8: return run_table<8>(op, model, z, g, out, b, lr, wd, eps, projected);
        default: return -2;
    }
}

extern "C" int sympa_hostsim_spd_dist(const double* x, const double* y, int64_t b, int n, double* out, int32_t* status) {
    if (n < 1 || n > sympa::SPD_MAX_N) return -2;
    int st = 0;
    sympa::SpdWork w;
    for (int64_t i = 0; i < b; ++i) out[i] = sympa::spd_pair_distance(w, x + i * n * n, y + i * n * n, n, st);
    if (status) *status = st;
    return 0;
}

extern "C" int sympa_hostsim_spd_bwd(const double* x, const double* y, int64_t b, int n, double* out, double* gx, double* gy,
                                     int32_t* status) {
    if (n < 1 || n > sympa::SPD_MAX_N) return -2;
    int st = 0;
    sympa::SpdBwdWork w;
    for (int64_t i = 0; i < b; ++i)
        out[i] = sympa::spd_pair_backward(w, x + i * n * n, y + i * n * n, n, gx + i * n * n, gy + i * n * n, st);
    if (status) *status = st;
    return 0;
}

// op 0: projx, 1: rsgd step (in place on a copy written to out), 2: egrad2rgrad
extern "C" int sympa_hostsim_spd_table(int op, int n, const double* x, const double* g, double* out, int64_t b, double lr,
                                       double wd, int32_t* moved) {
    if (n < 1 || n > sympa::SPD_MAX_N) return -2;
    int st = 0, mv = 0;
    sympa::SpdRowWork w;
    for (int64_t i = 0; i < b; ++i) {
        const double* px = x + i * n * n;
        double* po = out + i * n * n;
        if (op == 0) mv += sympa::spd_row_projx(w, px, n, po, st) ? 1 : 0;
        else if (op == 2) sympa::spd_row_egrad2rgrad(w, px, g + i * n * n, n, po);
        else {
            for (int k = 0; k < n * n; ++k) po[k] = px[k];
            sympa::spd_row_rsgd(w, po, g + i * n * n, n, lr, wd, 1.0, st);
        }
    }
    if (moved) *moved = mv;
    return st;
}

// Eigenvalues of b symmetric s x s matrices (full row-major in): packed one-lane Householder (spd_math.hpp
// tridiag_packed, the trailing-block routine of the lanes-per-pair kernels) + the runtime QL.
template <int S>
static void run_tridiag_packed(const double* a, int64_t b, double* eig) {
    for (int64_t q = 0; q < b; ++q) {
        double pk[S * (S + 1) / 2], d[S], e2[S];
        for (int i = 0; i < S; ++i)
            for (int j = 0; j <= i; ++j) pk[i * (i + 1) / 2 + j] = a[q * S * S + i * S + j];
        for (int i = 0; i < S; ++i) e2[i] = 0.0;
        sympa::tridiag_packed<S>(pk, d, e2);
        sympa::tridiag_ql_runtime(d, e2, S);
        for (int i = 0; i < S; ++i) eig[q * S + i] = d[i];
    }
}
extern "C" int sympa_hostsim_tridiag_packed(const double* a, int64_t b, int s, double* eig) {
    switch (s) {
        case 2: run_tridiag_packed<2>(a, b, eig); return 0;
        case 3: run_tridiag_packed<3>(a, b, eig); return 0;
        case 4: run_tridiag_packed<4>(a, b, eig); return 0;
        case 5: run_tridiag_packed<5>(a, b, eig); return 0;
        case 6: run_tridiag_packed<6>(a, b, eig); return 0;
        case 7: run_tridiag_packed<7>(a, b, eig); return 0;
        case 8: run_tridiag_packed<8>(a, b, eig); return 0;
        case 9: run_tridiag_packed<9>(a, b, eig); return 0;
        case 10: run_tridiag_packed<10>(a, b, eig); return 0;
        case 12: run_tridiag_packed<12>(a, b, eig); return 0;
        case 16: run_tridiag_packed<16>(a, b, eig); return 0;
        default: return -2;
    }
}


// Eigen-decomposition of b symmetric tridiagonal S x S matrices (d [b, S], e [b, S]: e[i] = T[i+1][i], e[S-1] ignored) the way
// the two-phase SPD backward does it: eigenvalues by the lockstep PWK QL (siegel_math.hpp), sorted, eigenvectors by inverse
// iteration (tridiag_invit.hpp).  lam [b, S] ascending, z [b, S, S]: row i = eigenvector i; flag [b] = 1 where a block of
// more than INVIT_KEEP + 1 close eigenvalues was met (the kernel routes such pairs to the QL-with-vectors kernel).
template <int S>
static void run_tridiag_invit(const double* dd, const double* ee, int64_t b, double* lam_out, double* z_out, int32_t* flag) {
    for (int64_t q = 0; q < b; ++q) {
        double d[S], e[S], w[S], e2[S];
        for (int i = 0; i < S; ++i) { d[i] = dd[q * S + i]; e[i] = (i < S - 1) ? ee[q * S + i] : 0.0; w[i] = d[i]; e2[i] = e[i] * e[i]; }
        const bool conv = sympa::tridiag_ql_lockstep<S>(w, e2);
        sympa::sort_ascending<S>(w);
        double* zq = z_out + q * S * S;
        const bool ok = sympa::tridiag_eigvecs_invit<S>(d, e, w, [&](auto IC, const double (&x)[S]) {
            constexpr int i = decltype(IC)::value;
            for (int j = 0; j < S; ++j) zq[i * S + j] = x[j];
        });
        for (int i = 0; i < S; ++i) lam_out[q * S + i] = w[i];
        flag[q] = (ok ? 0 : 1) | (conv ? 0 : 2);
    }
}
extern "C" int sympa_hostsim_tridiag_invit(const double* d, const double* e, int64_t b, int s, double* lam, double* z, int32_t* flag) {
    switch (s) {
        case 3: run_tridiag_invit<3>(d, e, b, lam, z, flag); return 0;
        case 8: run_tridiag_invit<8>(d, e, b, lam, z, flag); return 0;
        case 12: run_tridiag_invit<12>(d, e, b, lam, z, flag); return 0;
        case 16: run_tridiag_invit<16>(d, e, b, lam, z, flag); return 0;
        default: return -2;
    }
}
