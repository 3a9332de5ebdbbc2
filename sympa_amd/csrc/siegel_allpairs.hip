// All-pairs distance matrix with per-point factor reuse (Runner.build_distance_matrix,
// sympa/runner.py:142-154, feeds the mAP metric with N forward calls of N pairs each).
//
// Every point enters N pairs, so (1) a pack kernel factors each point ONCE (Y = L L^T / I - W W^H = C C^H), inverts the
// triangular factor and stores point + inverse factor in a tile-transposed image  pack[tile][k][lane]  (tile = 64
// points): reading "entry k of the 64 points of a tile" is one contiguous 512-byte access, no gather and no LDS.
// (2) The pair kernel gives every wave 64 COLUMN points (one per lane, loaded once) and a strip of 16 ROW points that
// are wave-uniform: their packed entries are fetched with scalar loads and enter the FMAs as scalar operands.  A pair
// then costs  E = A_i (Z_j - Z_i) A_j^T  (two triangular products: no Cholesky, no division, no square root, no index
// arithmetic) + the eigenvalue part -- ~10 % fewer instructions than the pairwise kernel and no gather at all.
// (3) d(i, j) = d(j, i): for the full matrix only the tile pairs with column tile >= row tile are computed and every
// value is stored twice (the mirrored store is a scattered 8-byte store per lane, ~1 instruction in 2000).
#include "siegel_common.hpp"

namespace {
using namespace sympa_hip;

template <int N, int MODEL>
__global__ __launch_bounds__(BLOCK) void allpairs_pack_kernel(const double* __restrict__ table, const int64_t num_rows,
                                                              double* __restrict__ pack, int32_t* status) {
    using P = sympa::PointPack<N, MODEL>;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;       // grid covers whole tiles of 64
    const int64_t ii = i < num_rows ? i : num_rows - 1;                // tail lanes of the last tile repeat the last point
    sympa::CMat<N> z;
    sympa::load_point<N>(table + ii * (2 * N * N), z);
    double p[P::LEN];
    const bool ok = sympa::pack_point<N, MODEL>(z, p);
    const int64_t tile = i >> 6;
    const int lane = (int)(i & 63);
    if (tile < (num_rows + 63) / 64) {                                 // the grid is rounded up to whole blocks
        double* dst = pack + (tile * P::LEN) * 64 + lane;
#pragma unroll
        for (int k = 0; k < P::LEN; ++k) dst[(int64_t)k * 64] = p[k];
    }
    if (status != nullptr) {
        const int bad = (i < num_rows && !ok) ? 1 : 0;
        const unsigned long long m = __ballot(bad);
        if (m != 0ull && (threadIdx.x & 63) == 0) {
            atomicOr(&status[0], sympa::ST_NOT_PD);
            atomicAdd(&status[1], (int)__popcll(m));
        }
    }
}

struct AllPairsArgs {
    const double* pack;
    int64_t num_rows;
    int64_t row_begin, row_end;     // rows of the matrix to produce
    const double* metric_w;
    const double* scale;
    double inv_scale_coef, inv_eps;
    double* out;                    // [row_end - row_begin, num_rows]
    int32_t* status;
    int metric;
    int symmetric;                  // full matrix: compute column tile >= row tile only, store both (i, j) and (j, i)
    int row0;                       // first row of the grid: row_begin rounded down to a multiple of 64
    int rows_per_wave;              // 1, 2, 4, 8 or 16: a block of 4 waves covers 4 * rows_per_wave rows of one row tile
};

// wave-uniform packed row point: entry k through a scalar (uniform-address) load
template <int LEN>
struct UniformPack {
    double v[LEN];
    __device__ __forceinline__ double operator[](int k) const { return v[k]; }
};

// column point of a lane kept in LDS: entry k of the 64 points of the tile is one conflict-free 512-byte row
struct LdsPack {
    const double* base;       // tile + lane
    __device__ __forceinline__ double operator[](int k) const { return base[k * 64]; }
};

// dims 8: the packed column point (108 doubles upper, 136 bounded) does not fit the register file beside E and H
// (round 2: 9.1 ms against 6.2 ms for the pairwise kernel at N = 5 041).  The four waves of a block work on the SAME column
// tile (they differ in their row strips), so the tile -- 55 / 70 KB, already laid out [entry][lane] in the workspace -- is
// copied into LDS once per block and every lane reads its column's entries from there, each once per row point.
template <int N>
constexpr bool allpairs_lds_columns() { return N >= 8; }

template <int N, int MODEL>
__global__ __launch_bounds__(BLOCK) void allpairs_kernel(const AllPairsArgs a) {
    using P = sympa::PointPack<N, MODEL>;
    constexpr bool LDS_COLS = allpairs_lds_columns<N>();
    __shared__ double col_tile[LDS_COLS ? P::LEN * 64 : 1];
    const int jt = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t i0 = (int64_t)a.row0 + ((int64_t)blockIdx.y * 4 + wave) * a.rows_per_wave;
    const int it = (int)(i0 >> 6);
    if constexpr (LDS_COLS) {
        // block-uniform exit only (a barrier follows): the first wave of the block has the smallest row tile
        const int it0 = (int)(((int64_t)a.row0 + (int64_t)blockIdx.y * 4 * a.rows_per_wave) >> 6);
        if (a.symmetric && jt < it0) return;
        const double* src = a.pack + ((int64_t)jt * P::LEN) * 64;
        for (int t = threadIdx.x; t < P::LEN * 64 / 2; t += BLOCK)
            reinterpret_cast<v2d*>(col_tile)[t] = reinterpret_cast<const v2d*>(src)[t];
        __syncthreads();
    }
    if (a.symmetric && jt < it) return;
    const int64_t j = (int64_t)jt * 64 + lane;
    const bool jlive = j < a.num_rows;
    // my column point
    double pj_regs[LDS_COLS ? 1 : P::LEN];
    if constexpr (!LDS_COLS) {
        const double* src = a.pack + ((int64_t)jt * P::LEN) * 64 + lane;
#pragma unroll
        for (int k = 0; k < P::LEN; ++k) pj_regs[k] = src[(int64_t)k * 64];
    }
    const LdsPack pj_lds{col_tile + lane};
    double sc = 1.0;
    if (a.scale != nullptr) sc = fmax(a.scale[0] * a.inv_scale_coef, 0.1);     // model.py:40-41
    int st = 0;
    for (int rr = 0; rr < a.rows_per_wave; ++rr) {
        const int64_t i = i0 + rr;                       // wave-uniform
        if (i >= a.row_end) break;
        if (i < a.row_begin) continue;
        UniformPack<P::LEN> pi;
        {
            const double* src = a.pack + ((int64_t)it * P::LEN) * 64 + (int)(i & 63);
#pragma unroll
            for (int k = 0; k < P::LEN; ++k) pi.v[k] = src[(int64_t)k * 64];
        }
        sympa::CMat<N> e;
        if constexpr (LDS_COLS) sympa::e_from_packed<N, MODEL>(pi, pj_lds, e);
        else sympa::e_from_packed<N, MODEL>(pi, pj_regs, e);
        double d = sympa::distance_from_e<N, MODEL>(e, true, a.metric, a.metric_w, a.inv_eps, nullptr, st) * sc;
        if (jlive) {
            if (!a.symmetric) {
                __builtin_nontemporal_store(d, a.out + (i - a.row_begin) * a.num_rows + j);
            } else if (j >= i) {           // (diagonal tiles evaluate both orders of a pair: only i <= j is kept)
                __builtin_nontemporal_store(d, a.out + i * a.num_rows + j);
                if (j > i) __builtin_nontemporal_store(d, a.out + j * a.num_rows + i);
            }
        }
    }
    if (a.status != nullptr) {
        const int flagged = (jlive && st != 0) ? 1 : 0;
        const unsigned long long m = __ballot(flagged);
        if (m != 0ull) {
            if (flagged) atomicOr(&a.status[0], st);
            if (lane == 0) atomicAdd(&a.status[1], (int)__popcll(m));
        }
    }
}

template <int N, int MODEL>
int launch_allpairs(const double* table, int64_t num_rows, const AllPairsArgs& a0, double* pack, hipStream_t s) {
    const int64_t tiles = (num_rows + 63) / 64;
    AllPairsArgs a = a0;
    a.pack = pack;
    hipLaunchKernelGGL((allpairs_pack_kernel<N, MODEL>), dim3((unsigned)((tiles * 64 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s,
                       table, num_rows, pack, a.status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    a.row0 = (int)(a.row_begin / 64 * 64);
    const int64_t rows = a.row_end - a.row0;
    // rows per wave: as many as keep >= ~8 waves per SIMD in the grid (a wave reloads nothing between its rows)
    int rpw = 16;
    while (rpw > 1 && tiles * ((rows + 4 * rpw - 1) / (4 * rpw)) * 4 / (a.symmetric ? 2 : 1) < 8192) rpw >>= 1;
    a.rows_per_wave = rpw;
    const int64_t row_blocks = (rows + 4 * rpw - 1) / (4 * rpw);
    hipLaunchKernelGGL((allpairs_kernel<N, MODEL>), dim3((unsigned)tiles, (unsigned)row_blocks), dim3(BLOCK), 0, s, a);
    e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

template <int N>
int launch_allpairs_n(const double* table, int64_t num_rows, const AllPairsArgs& a, double* pack, int model, hipStream_t s) {
    return model == SYMPA_MODEL_UPPER ? launch_allpairs<N, sympa::MODEL_UPPER>(table, num_rows, a, pack, s)
                                      : launch_allpairs<N, sympa::MODEL_BOUNDED>(table, num_rows, a, pack, s);
}

int pack_len(int n, int model) {
    const int tri = n * (n + 1) / 2, low = n * (n - 1) / 2;
    return 2 * tri + n + (model == SYMPA_MODEL_UPPER ? low : 2 * low);
}

}  // namespace

extern "C" {

int64_t sympa_all_pairs_workspace_bytes(int64_t num_rows, int n, int model) {
    if (num_rows <= 0 || n < 1 || n > SYMPA_MAX_DIMS_ALL_PAIRS_PACKED) return 0;
    if (model != SYMPA_MODEL_UPPER && model != SYMPA_MODEL_BOUNDED) return 0;
    return ((num_rows + 63) / 64) * 64 * (int64_t)pack_len(n, model) * 8;
}

int sympa_all_pairs_dist_packed(const double* table, int64_t num_rows, int n, int64_t row_begin, int64_t row_count,
                                int model, int metric, const double* metric_w, double eps, const double* scale,
                                double scale_coef, double* out, void* workspace, int64_t workspace_bytes,
                                int32_t* status, int flags, void* stream) {
    if (num_rows <= 0 || row_begin < 0 || row_count < 0 || row_begin + row_count > num_rows)
        return fail(SYMPA_ERR_BAD_ARG, "row block outside the table");
    if (row_count == 0) return 0;
    if (table == nullptr || out == nullptr) return fail(SYMPA_ERR_BAD_ARG, "null buffer");
    if (model != SYMPA_MODEL_UPPER && model != SYMPA_MODEL_BOUNDED) return fail(SYMPA_ERR_BAD_ARG, "unknown model");
    if (metric < SYMPA_METRIC_RIEM || metric > SYMPA_METRIC_WSUM) return fail(SYMPA_ERR_BAD_ARG, "unknown metric");
    if (metric == SYMPA_METRIC_WSUM && metric_w == nullptr) return fail(SYMPA_ERR_BAD_ARG, "metric wsum needs metric_w");
    if (!(eps > 0.0) || !(1.0 / eps < 1e300)) return fail(SYMPA_ERR_BAD_ARG, "eps must be > 0");
    if (scale != nullptr && !(scale_coef != 0.0)) return fail(SYMPA_ERR_BAD_ARG, "scale_coef must be non-zero");
    if (n < 1 || n > SYMPA_MAX_DIMS_ALL_PAIRS_PACKED)
        return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "packed all-pairs kernel: dims outside [1, SYMPA_MAX_DIMS_ALL_PAIRS_PACKED]");
    const int64_t need = sympa_all_pairs_workspace_bytes(num_rows, n, model);
    if (workspace == nullptr || workspace_bytes < need) return fail(SYMPA_ERR_BAD_ARG, "workspace too small");
    if (num_rows > (int64_t)0x7fffffff) return fail(SYMPA_ERR_BAD_ARG, "more than 2^31-1 table rows");
    AllPairsArgs a;
    std::memset(&a, 0, sizeof(a));
    a.num_rows = num_rows;
    a.row_begin = row_begin;
    a.row_end = row_begin + row_count;
    a.metric_w = metric_w;
    a.scale = scale;
    a.inv_scale_coef = 1.0 / scale_coef;
    a.inv_eps = 1.0 / eps;
    a.out = out;
    a.status = status;
    a.metric = metric;
    // the mirrored stores are scattered 8-byte stores: worth it where a pair costs >= ~1000 instructions (dims >= 3)
    a.symmetric = (row_begin == 0 && row_count == num_rows && n >= 3 && !(flags & SYMPA_FLAG_NO_SYMMETRY)) ? 1 : 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    double* pack = reinterpret_cast<double*>(workspace);
    switch (n) {
        case 1: return launch_allpairs_n<1>(table, num_rows, a, pack, model, s);
        case 2: return launch_allpairs_n<2>(table, num_rows, a, pack, model, s);
        case 3: return launch_allpairs_n<3>(table, num_rows, a, pack, model, s);
        case 4: return launch_allpairs_n<4>(table, num_rows, a, pack, model, s);
        case 5: return launch_allpairs_n<5>(table, num_rows, a, pack, model, s);
        case 6: return launch_allpairs_n<6>(table, num_rows, a, pack, model, s);
        case 7: return launch_allpairs_n<7>(table, num_rows, a, pack, model, s);
        case 8: return launch_allpairs_n<8>(table, num_rows, a, pack, model, s);
        default: break;
    }
    return fail(SYMPA_ERR_UNSUPPORTED_DIMS, "packed all-pairs kernel: unsupported dims");
}

}  // extern "C"
