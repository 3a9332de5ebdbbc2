// The packed indexed forward of the upper model at dims 7, 8 with TWO waves per SIMD (round 5; what sympa_model_forward_packed /
// sympa_model_forward_batches_packed launch there).
//
// Why.  One pair per lane, the n = 8 forward holds E (128 doubles) next to a factor and the accumulators: 412 registers, ONE wave per
// SIMD, and a lone wave issues an fp64 instruction every ~8.5 cycles; a second resident wave is worth 1.33x to the eigenvalue stage
// (tools/microbench/eigen_occupancy.hip).  A two-kernel form that gave the eigenvalue stage its second wave lost the gain to a front
// kernel that had nothing to hide its gathers behind (profiles/r05_packed_forward.txt, block 7).  Here both halves live in ONE
// persistent kernel of at most 256 registers:
//   * FRONT, two LANES per pair, 32 pairs per pass: E = A1 (Z2 - Z1) A2^T with the REAL factors of the upper model never mixes the
//     planes, so lane 2p works on Re E of pair p and lane 2p + 1 on Im E (64 doubles of E per lane); the inverted factors are read from
//     the LDS row by row as the products need them, the triangles arrive through a ring of eight-row passes and are subtracted in place.
//   * H = E^H E column by column: Re H[c][k] = own Gram sum + the partner's (one exchange), Im H[c][k] = sum_r (Im E[r][k] Re E[r][c] -
//     Re E[r][k] Im E[r][c]) with the partner's column c arriving through 32-bit DPP moves (quad_perm [1,0,3,2]).  Column c of E is dead
//     when row c of H is complete, and each lane KEEPS only half of H (the even lane Re H[c][k], the odd lane Im H[c][k], the diagonal
//     split): E shrinks as fast as the kept half of H grows, the pass never holds more than ~130 registers of the two.
//   * a tile of 64 pairs is two passes; the first pass's half of H waits in 64 registers while the second runs (166 + 64 registers);
//     then 2 x 32 ds_bpermute pairs turn [pass][pair][plane] into one pair per lane (lane L < 32: pair L of the first pass, else pair
//     L - 32 of the second) and the EIGENVALUE stage runs as in the one-pair-per-lane kernels: 170 registers.
//   * persistent waves, software-pipelined by HALF tiles: the next pass's factor tile and first two triangle passes are issued before
//     the current pass's Gram sums (and the eigenvalue stage, every other pass), the ids one pass earlier still.
#pragma once
#include "siegel_packed_kernel.hpp"

namespace sympa_hip {

template <int N>
struct Pair2 {
    using P = sympa::PointPack<N, sympa::MODEL_UPPER>;
    static constexpr int TRI = P::TRI;                       // doubles of one plane's triangle
    static constexpr int TC = TRI / 2;                       // ... in 16-byte chunks
    static constexpr int ZC = 2 * TC;                        // chunks of both triangles: what a Z pass fetches of a row
    static constexpr int ALEN = N + P::LOW;                  // doubles of the inverted factor (diagonal, strict lower part)
    static constexpr int AC = ALEN / 2;
    static constexpr int ROW_DOUBLES = PackRow<N, sympa::MODEL_UPPER>::ROW_DOUBLES;
    static constexpr int ZPITCH = ZC | 1;                    // LDS slots per row, odd
    static constexpr int ZROWS = 16;                         // pairs per Z step: two steps per point
    static constexpr int ZBUF = ZROWS * ZPITCH;
    static constexpr int APITCH = AC | 1;
    static constexpr int A_PER_INSTR = 64 / APITCH;          // factor rows per DMA instruction (3 at n = 8)
    static constexpr int A_INSTR = (32 + A_PER_INSTR - 1) / A_PER_INSTR;
    static constexpr int A_INSTR_SLOTS = A_PER_INSTR * APITCH;
    static constexpr int ATILE = (32 / A_PER_INSTR) * A_INSTR_SLOTS + (32 % A_PER_INSTR) * APITCH;   // (the last instruction's idle lanes write nothing)
    static constexpr int BUF = (ZBUF > ATILE) ? ZBUF : ATILE;   // each of the two LDS regions: a Z step, later a factor tile
    static constexpr int NP = N * (N - 1) / 2;               // pairs (j < k): slot p holds Re H (even lane) / Im H (odd lane)
    static constexpr int ND = (N + 1) / 2;                   // slot NP + q holds d[q] (even lane) / d[ND + q] (odd lane)
    static constexpr int SL = NP + ND;                       // kept doubles per lane and pass
    static_assert(TRI % 2 == 0 && ALEN % 2 == 0, "chunk-aligned planes: dims 7, 8");
    static_assert(A_PER_INSTR >= 2 && A_PER_INSTR <= 4, "factor rows per DMA instruction");
};

constexpr bool packed2_dims_ok(int n) { return n == 7 || n == 8; }
// slot i of the strict upper triangle, row-major: (j, k)
constexpr int pair2_row(int n, int i) { int j = 0; while (i >= n - 1 - j) { i -= n - 1 - j; ++j; } return j; }
constexpr int pair2_col(int n, int i) { int j = 0; while (i >= n - 1 - j) { i -= n - 1 - j; ++j; } return j + 1 + i; }
// compile-time loop (the indices above have to be constant expressions: a register array indexed by a value the optimiser folds
// too late lives in scratch memory)
template <int I, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// swap a value with the other lane of my pair (lanes 2p, 2p + 1): 32-bit DPP moves
__device__ __forceinline__ int pair_swap_i(const int x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ double pair_swap(const double x) {
    const long long b = __double_as_longlong(x);
    const int lo = pair_swap_i((int)(b & 0xffffffffll));
    const int hi = pair_swap_i((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double lane_gather(const int byte_addr, const double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// the factor rows of the 32 pairs of a pass: A_PER_INSTR rows per DMA instruction, row of pair q at
// tile + (q / A_PER_INSTR) * A_INSTR_SLOTS + (q % A_PER_INSTR) * APITCH
template <int N>
__device__ __forceinline__ void pair2_issue_factors(const double* __restrict__ pack, const int row, v2d* __restrict__ tile) {
    using S = Pair2<N>;
    const int lane = threadIdx.x & 63;
    const int sub = lane / S::APITCH;                       // which of the rows of an instruction I fetch for
    const int c = lane - sub * S::APITCH;                   // chunk of that row (c == AC: the padding slot, nothing fetched)
    const bool on = sub < S::A_PER_INSTR && c < S::AC;
#pragma unroll
    for (int j = 0; j < S::A_INSTR; ++j) {
        // pair q = A_PER_INSTR j + sub; its row index sits in lanes 2q, 2q + 1
        int rr = __builtin_amdgcn_readlane(row, 2 * ((S::A_PER_INSTR * j + 0) & 31));
        if (S::A_PER_INSTR > 1) { const int r1 = __builtin_amdgcn_readlane(row, 2 * ((S::A_PER_INSTR * j + 1) & 31)); rr = sub == 1 ? r1 : rr; }
        if (S::A_PER_INSTR > 2) { const int r2 = __builtin_amdgcn_readlane(row, 2 * ((S::A_PER_INSTR * j + 2) & 31)); rr = sub == 2 ? r2 : rr; }
        if (S::A_PER_INSTR > 3) { const int r3 = __builtin_amdgcn_readlane(row, 2 * ((S::A_PER_INSTR * j + 3) & 31)); rr = sub == 3 ? r3 : rr; }
        const double* src = pack + (int64_t)rr * S::ROW_DOUBLES + 2 * (S::ZC + c);
        if (on && S::A_PER_INSTR * j + sub < 32)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(tile + j * S::A_INSTR_SLOTS), 16, 0, 0);
    }
}

// one Z pass: the two triangles (chunks [0, ZC)) of the rows of pairs 8 pass .. 8 pass + 7, one DMA instruction per row
template <int N>
__device__ __forceinline__ void pair2_issue_zpass(const double* __restrict__ pack, const int row, const int pass, v2d* __restrict__ buf) {
    using S = Pair2<N>;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < S::ZROWS; ++j) {
        const int rr = __builtin_amdgcn_readlane(row, 2 * (S::ZROWS * pass + j));
        const double* src = pack + (int64_t)rr * S::ROW_DOUBLES + 2 * lane;
        if (lane < S::ZC) __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(buf + j * S::ZPITCH), 16, 0, 0);
    }
}

// my plane's triangle of the rows of a pass: SECOND = false: dh = it (the pair's second point); true: dh -= it (the first)
template <int N, bool SECOND>
__device__ __forceinline__ void pair2_read_zpass(const v2d* __restrict__ buf, const int pass, double (&dh)[Pair2<N>::TRI]) {
    using S = Pair2<N>;
    const int lane = threadIdx.x & 63;
    const int pr = lane >> 1, part = lane & 1;
    if ((pr / S::ZROWS) == pass) {
        const v2d* mine = buf + (pr % S::ZROWS) * S::ZPITCH + part * S::TC;
        constexpr int G = 9;
#pragma unroll
        for (int c0 = 0; c0 < S::TC; c0 += G) {
            v2d q[G];
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (c0 + g < S::TC) q[g] = mine[c0 + g];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int c = c0 + g;
                if (c < S::TC) {
                    if (SECOND) { dh[2 * c] -= q[g].x; dh[2 * c + 1] -= q[g].y; }
                    else { dh[2 * c] = q[g].x; dh[2 * c + 1] = q[g].y; }
                }
            }
        }
    }
}

// One pass: 32 pairs, two lanes each.  In: the pass's rows (row1 / row2 per lane = per pair), its factor tile and first two triangle
// passes already in flight (issued by the pass before: `mid` of that pass), everything else waited for by the caller.  `mid` runs
// when both LDS regions are free again (after E): the caller issues the next pass's head there.  Out: what this lane keeps of H.
template <int N, class Mid>
__device__ __forceinline__ bool pair2_pass(const double* __restrict__ pack, const volatile int* __restrict__ rows, v2d* __restrict__ r0,
                                           v2d* __restrict__ r1, Mid&& mid, double (&out)[Pair2<N>::SL]) {
    using S = Pair2<N>;
    const int lane = threadIdx.x & 63;
    const int pr = lane >> 1, part = lane & 1;
    // the pass's table rows wait in the LDS (rows[pr], rows[32 + pr]): a register that lives from the id check to the last DMA
    // instruction of the pass is the first thing the allocator spills, and every reload of it sits between two DMA instructions
    // behind an s_waitcnt vmcnt(0) -- each gather instruction then waits for the one before it
    const int row1 = rows[pr], row2 = rows[32 + pr];
    // ---- D = Z2 - Z1: four steps of sixteen rows through the two regions (steps 0, 1 = the second point's rows, in flight since the
    // middle of the pass before; 2, 3 = the first point's), then the two factor tiles into the same regions
    double dh[S::TRI];
#pragma unroll
    for (int k = 0; k < S::TRI; ++k) dh[k] = 0.0;
    wave_lds_fence();
    pair2_read_zpass<N, false>(r0, 0, dh);                 // step 0 (landed: the caller's vmcnt(0))
    __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0): the reads of the region being refilled are complete
    wave_lds_fence();
    pair2_issue_zpass<N>(pack, row1, 0, r0);               // step 2
    pair2_read_zpass<N, false>(r1, 1, dh);                 // step 1
    __builtin_amdgcn_s_waitcnt(0xC07F);
    wave_lds_fence();
    pair2_issue_zpass<N>(pack, row1, 1, r1);               // step 3
    wait_vmcnt<S::ZROWS>();                                // step 2 has landed
    wave_lds_fence();
    pair2_read_zpass<N, true>(r0, 0, dh);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    wave_lds_fence();
    pair2_issue_factors<N>(pack, row1, r0);                // A1
    wait_vmcnt<S::A_INSTR>();                              // step 3 has landed
    wave_lds_fence();
    pair2_read_zpass<N, true>(r1, 1, dh);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    wave_lds_fence();
    pair2_issue_factors<N>(pack, row2, r1);                // A2
    // (derived from the lane id HERE, every pass, behind an empty asm: as a loop invariant the 72 addresses of the factor entries
    // are hoisted out of the tile loop -- 72 registers, all of them spilled -- instead of being one base register + immediates)
    int pr_ = pr;
    asm volatile("" : "+v"(pr_));
    const int aoff = (pr_ / S::A_PER_INSTR) * S::A_INSTR_SLOTS + (pr_ % S::A_PER_INSTR) * S::APITCH;
    // volatile: every factor entry is read where its product needs it (hoisted to the top, the 36 doubles of a factor are 72
    // registers of a kernel that has to stay at 256)
    const volatile double* a1 = reinterpret_cast<const volatile double*>(r0 + aoff);     // [diag N | strict lower, row-major]
    const volatile double* a2 = reinterpret_cast<const volatile double*>(r1 + aoff);
    double e[N][N];                                                     // my plane of D, then of T, then of E
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = r; c < N; ++c) { e[r][c] = dh[sympa::tri_index(N, r, c)]; e[c][r] = e[r][c]; }
    wait_vmcnt<S::A_INSTR>();                              // A1 has landed (A2 may still be in flight)
    wave_lds_fence();
    const bool ok1 = sympa::d_finite(a1[0]);
    // T = A1 D, rows N-1 .. 0 (row r needs rows k <= r of D only)
#pragma unroll
    for (int rr = 0; rr < N; ++rr) {
        const int r = N - 1 - rr;
        const double dg = a1[r];
        double lr[N];
#pragma unroll
        for (int k = 0; k < r; ++k) lr[k] = a1[N + sympa::low_index(r, k)];
#pragma unroll
        for (int c = 0; c < N; ++c) {
            double x = dg * e[r][c];
#pragma unroll
            for (int k = 0; k < r; ++k) x = sympa::d_fma(lr[k], e[k][c], x);
            e[r][c] = x;
        }
    }
    // pin T, then E: the pass is ONE basic block of ~3 000 instructions, and left to itself the instruction selection interleaves
    // the products with each other and with the Gram sums (every value of T, E and the partial sums alive at once: 100+
    // registers in scratch); an empty asm that reads and writes a value in place costs nothing and fixes the order
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) asm volatile("" : "+v"(e[r][c]));
    __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0): A2 has landed
    wave_lds_fence();
    const bool ok = ok1 && sympa::d_finite(a2[0]);
    // E = T A2^T, columns N-1 .. 0
#pragma unroll
    for (int cc = 0; cc < N; ++cc) {
        const int c = N - 1 - cc;
        const double dg = a2[c];
        double lr[N];
#pragma unroll
        for (int k = 0; k < c; ++k) lr[k] = a2[N + sympa::low_index(c, k)];
#pragma unroll
        for (int r = 0; r < N; ++r) {
            double x = e[r][c] * dg;
#pragma unroll
            for (int k = 0; k < c; ++k) x = sympa::d_fma(e[r][k], lr[k], x);
            e[r][c] = x;
        }
    }
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) asm volatile("" : "+v"(e[r][c]));
    // ---- both LDS regions are free
    __builtin_amdgcn_s_waitcnt(0xC07F);
    wave_lds_fence();
    mid();
    // ---- H = E^H E, row c of H at a time; what this lane keeps of it goes to out[]
#pragma unroll
    for (int c = 0; c < N; ++c) {
        __builtin_amdgcn_sched_barrier(0);             // one column at a time
        {
            double g = 0.0;
#pragma unroll
            for (int r = 0; r < N; ++r) g = sympa::d_fma(e[r][c], e[r][c], g);
            g += pair_swap(g);
            if (c < S::ND) out[S::NP + c] = g;
            else out[S::NP + c - S::ND] = part ? g : out[S::NP + c - S::ND];
        }
        if (c + 1 < N) {
            double oc[N];                              // the partner's column c
#pragma unroll
            for (int r = 0; r < N; ++r) oc[r] = pair_swap(e[r][c]);
#pragma unroll
            for (int k = c + 1; k < N; ++k) {
                double g = 0.0, x = 0.0;
#pragma unroll
                for (int r = 0; r < N; ++r) {
                    g = sympa::d_fma(e[r][c], e[r][k], g);
                    x = sympa::d_fma(e[r][k], oc[r], x);
                }
                g += pair_swap(g);                     // Re H[c][k]
                const double him = x - pair_swap(x);   // in the odd lane: Im H[c][k] = sum_r (Im E[r][k] Re E[r][c] - Re E[r][k] Im E[r][c])
                out[c * N - c * (c + 1) / 2 + (k - c - 1)] = part ? him : g;
            }
        }
    }
    return ok;
}

template <int N>
__global__ __launch_bounds__(64, 2) void packed_forward2_kernel(const PackedArgs a) {
    using S = Pair2<N>;
    __shared__ v2d r0[S::BUF];
    __shared__ v2d r1[S::BUF];
    const int lane = threadIdx.x & 63;
    const int pr = lane >> 1, part = lane & 1;
    unsigned t = blockIdx.x;
    if (t >= a.tiles) return;
    // the second resident wave of every SIMD starts late: two waves in step wait for their rows together and then contend for the
    // issue slots together; out of step one's arithmetic runs behind the other's gathers
    if (a.stagger > 0 && blockIdx.x >= (gridDim.x >> 1))
        for (int j = 0; j < a.stagger; ++j) __builtin_amdgcn_s_sleep(127);
    // ids of pass `half` of tile t: lane l serves pair 32 half + (l >> 1) of the tile
    auto ids_of = [&](const unsigned tile, const unsigned half, int64_t& x1, int64_t& x2) {
        const int kb = packed_batch_of(a, tile);
        const unsigned t0 = (kb == 0) ? 0u : a.tile_end[kb - 1];
        const int64_t i = (int64_t)(tile - t0) * 64 + half * 32 + pr;
        const int64_t ii = i < a.b[kb] ? i : a.b[kb] - 1;
        x1 = __builtin_nontemporal_load(a.idx1[kb] + ii * a.stride1);
        x2 = __builtin_nontemporal_load(a.idx2[kb] + ii * a.stride2);
    };
    __shared__ int rws[2][2][32];                                       // [slot][row1 | row2][pair of the pass]
    auto put_rows = [&](const int slot, const int row1, const int row2) {
        wave_lds_fence();
        rws[slot][0][pr] = row1;                                        // (both lanes of a pair write the same values)
        rws[slot][1][pr] = row2;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        wave_lds_fence();
    };
    auto issue_head = [&](const int slot) {                             // steps 0, 1: the second point's rows of the pass in `slot`
        const int row2 = reinterpret_cast<const volatile int*>(&rws[slot][1][0])[pr];
        pair2_issue_zpass<N>(a.pack, row2, 0, r0);
        pair2_issue_zpass<N>(a.pack, row2, 1, r1);
    };
    // software pipeline by passes: the head (the second point's triangles) of a pass is issued in the middle of the pass before it,
    // its ids one pass earlier still
    int64_t x1, x2;
    int c1, c2, sta;                                                    // (sta: the first pass of the current tile)
    ids_of(t, 0, x1, x2);
    packed_ids_check(a, x1, x2, c1, c2, sta);
    put_rows(0, c1, c2);
    issue_head(0);
    ids_of(t, 1, x1, x2);
    // lane L of the eigenvalue stage takes what lanes (2L, 2L + 1) of the first pass (L < 32) or (2(L - 32), 2(L - 32) + 1) of the
    // second kept: one gather from lane 2L / 2(L - 32) + 1 of a register that holds the first pass's value in the even lanes and
    // the second pass's in the odd lanes
    const int gather_addr = 4 * ((lane < 32) ? 2 * lane : 2 * (lane - 32) + 1);
    for (;;) {
        const unsigned tn = t + gridDim.x;
        const bool more = tn < a.tiles;                                 // wave-uniform
        __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0): the first pass's head, the second pass's ids, earlier stores
        int stb;
        packed_ids_check(a, x1, x2, c1, c2, stb);
        put_rows(1, c1, c2);
        double first[S::SL];                                            // the first pass's half of H while the second runs
        const bool ok_a = pair2_pass<N>(a.pack, reinterpret_cast<const volatile int*>(&rws[0][0][0]), r0, r1, [&]() {
            issue_head(1);
            x1 = 0;
            x2 = 0;
            if (more) ids_of(tn, 0, x1, x2);
        }, first);
        const int st_first = sta | (ok_a ? 0 : sympa::ST_NOT_PD);
        __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0): the second pass's head, the next tile's first ids
        packed_ids_check(a, x1, x2, c1, c2, sta);      // (the next tile's first pass; zeros when there is none)
        put_rows(0, c1, c2);
        double second[S::SL];
        const bool ok_b = pair2_pass<N>(a.pack, reinterpret_cast<const volatile int*>(&rws[1][0][0]), r0, r1, [&]() {
            x1 = 0;
            x2 = 0;
            if (more) {
                issue_head(0);
                ids_of(tn, 1, x1, x2);
            }
        }, second);
        const int st_second = stb | (ok_b ? 0 : sympa::ST_NOT_PD);
        // ---- one pair per lane: lane L < 32 <- pair L of the first pass, lane L >= 32 <- pair L - 32 of the second
        // (first[i] and second[i] pass through ONE empty asm: nothing below may start before the second pass's Gram sums are
        // complete -- the exchanges of first[] hoisted above them are 64 more live registers)
#pragma unroll
        for (int i = 0; i < S::SL; ++i) asm volatile("" : "+v"(first[i]), "+v"(second[i]));
        sympa::Herm<N> h;
        static_for<0, S::SL>([&](auto I) {
            constexpr int i = decltype(I)::value;
            // (the exchanges run in ALL lanes, outside the selects: a DPP move under a divergent branch reads inactive lanes)
            const double second_x = pair_swap(second[i]), first_x = pair_swap(first[i]);
            const double from_even = part ? second_x : first[i];                  // even lanes: first pass; odd lanes: second pass
            const double from_odd = part ? second[i] : first_x;
            const double xe = lane_gather(gather_addr, from_even);                // what the even lane of my pair kept
            const double xo = lane_gather(gather_addr, from_odd);                 // what the odd lane kept
            if constexpr (i < S::NP) {
                constexpr int j = pair2_row(N, i), k = pair2_col(N, i);            // slot i = pair (j, k), j < k, row-major
                h.re[j][k] = xe;
                h.im[j][k] = xo;
            } else {
                h.d[i - S::NP] = xe;
                if constexpr (S::ND + i - S::NP < N) h.d[S::ND + i - S::NP] = xo;
            }
        });
        int flags = __builtin_amdgcn_ds_bpermute(gather_addr, part ? st_second : st_first);
        const int kb = packed_batch_of(a, t);
        const unsigned t0 = (kb == 0) ? 0u : a.tile_end[kb - 1];
        const int64_t i = (int64_t)(t - t0) * 64 + lane;
        const bool live = i < a.b[kb];
        double d = sympa::distance_from_h<N, sympa::MODEL_UPPER>(h, true, a.metric, a.metric_w, a.inv_eps, nullptr, flags);
        if (flags & sympa::ST_BAD_INDEX) d = __builtin_nan("");
        if (a.scale != nullptr) d *= fmax(a.scale[0] * a.inv_scale_coef, 0.1);   // model.py:40-41
        if (live) __builtin_nontemporal_store(d, a.out[kb] + i);
        if (a.status != nullptr) {
            const int flagged = (live && flags != 0) ? 1 : 0;
            const unsigned long long m = __ballot(flagged);
            if (m != 0ull) {
                if (flagged) atomicOr(&a.status[0], flags);
                if (lane == 0) atomicAdd(&a.status[1], (int)__popcll(m));
            }
        }
        if (!more) break;
        t = tn;
    }
}

template <int N>
int launch_packed_forward2(const PackedArgs& a, unsigned cus, hipStream_t s) {
    const unsigned res = resident_blocks(packed_forward2_kernel<N>, cus);
    hipLaunchKernelGGL((packed_forward2_kernel<N>), dim3(a.tiles < res ? a.tiles : res), dim3(64), 0, s, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace sympa_hip
