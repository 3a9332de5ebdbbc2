// SPD model, n = 16 (BASELINE.json configs[4]): sixteen lanes per pair, four pairs per wavefront.
//
// One lane per pair (spd_math.hpp) needs two 16 x 16 working matrices per lane: 4 KB of scratch per lane, and the
// kernel runs at the speed of scratch memory.  Here lane r of a group of 16 lanes owns ROW r of every matrix of its
// pair (16 doubles = 32 VGPRs per matrix), so the O(n^3) part with many operands stays in registers:
//
//   - elements of another row are read with the DPP modifier row_newbcast:j -- the only DPP mode the double-precision
//     ALU has, and exactly "lane j of my group of 16" -- fused into the FMA:  v_fmac_f64_dpp acc, x(lane j), y
//     is one instruction for  acc += X[j][.] * y;
//   - the symmetric matrices are staged through the LDS by whole-row DMA (global_load_lds_dwordx4: 1 KB per
//     instruction, 16 instructions per round of 4 pairs), one round ahead of the arithmetic;
//   - X = Lh D Lh^T (unit lower factor, right-looking), B = (Y - X) Lh^-T, a transpose through the LDS,
//     M = D^-1/2 (B^T Lh^-T) D^-1/2 (= L^-1 (Y - X) L^-T), then the FIRST M - TB Householder steps with the reflector of
//     step k taken from column k (one element per lane) and broadcast from there;
//   - d_k and e_k^2 of those steps come out as group-uniform values (every lane of the group has them): in round t lane
//     t of the group keeps them, and the TB x TB block that is left travels through the LDS into that lane's registers
//     (round 3: a step in this layout costs ~75 instructions of group-uniform work next to its useful FMAs and serves
//     four pairs, the same step one pair per lane serves 64).  After 16 rounds EVERY lane holds one pair's partial
//     tridiagonal form and trailing block, and the wave runs the rest -- the block's tridiagonalisation, PWK QL
//     (dsterf), log1p, norm -- ONE PAIR PER LANE, all 64 lanes busy.  Lane 16 g + t owns pair 4 t + g of the wave's 64.
//
// Same arithmetic as spd_math.hpp (same formula, same QL, same log1p), different order of summation: the two kernels
// agree to ~1e-14 and the tests check the one against the other and both against the oracle.
#pragma once

#include <type_traits>

#include "siegel_gather.hpp"
#include "spd_math.hpp"

namespace spd_coop {

constexpr int N = 16;                            // lanes per pair (a DPP row); matrices are M x M, M <= N
constexpr int ROUNDS = 16;                       // 4 pairs per round, 64 pairs per wave

// Lanes per pair.  Default: a whole DPP row of sixteen.  A translation unit that defines SYMPA_COOP_HALF before including
// this header gets GROUPS OF EIGHT -- two pairs per DPP row, eight per wave and step, for matrices with M <= 8: every
// per-lane instruction (the scalar recurrences, the square roots, the reductions' arithmetic) then serves eight pairs
// instead of four with four of them phantoms, and only the DPP instructions are issued twice, once per half of the row
// with the other half masked off (bank_mask) and the broadcast lane offset by eight.
#ifdef SYMPA_COOP_HALF
constexpr int GROUP = 8;
#else
constexpr int GROUP = 16;
#endif
constexpr int GROUPS_PER_WAVE = 64 / GROUP;

// Rounds per wave for the kernels whose waves are independent of each other's pairs (backward, table rows): one wave
// per SIMD is resident (512 registers), so up to 1024 x 4 pairs run at once -- spread a small batch over all SIMDs.
inline int coop_rounds(const long long b, const int waves_per_simd = 1, const int pairs_per_round = 4) {
    const long long slots = 1024ll * pairs_per_round * waves_per_simd;      // pairs in flight: 1024 SIMDs x waves x pairs
    const long long r = (b + slots - 1) / slots;
    return (int)(r < 1 ? 1 : (r > ROUNDS ? ROUNDS : r));
}
// The row-per-lane routines below are templates over the matrix size M <= 16: lane r < M of a group of sixteen owns
// row r, lanes r >= M are phantoms -- they execute the same instructions on values nobody reads (every DPP broadcast
// takes its source from a lane j < M, the group sums mask them out, the LDS transposes give them columns nobody
// wrote), and still take their turn as the lane that keeps a pair's tridiagonal form for the one-pair-per-lane
// QL phase.  A wave instruction costs the same with 10 or 16 active rows, so the time goes with M^2, not M^3:
// M = n exactly instead of padding every n to 16 is worth (16 / n)^2.
constexpr int TILE_BYTES = 4 * 2 * N * N * 8;    // one round: 4 pairs x {X, Y} x 2 KB
constexpr int LDS_BYTES = TILE_BYTES;            // the transpose of a round reuses the (consumed) tile

template <int I, int E, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, E>(f);
    }
}

// The DPP instructions are `asm volatile`: to the compiler a non-volatile asm is a pure per-thread function of its
// operands, which a cross-lane read is not, and with plain `asm` one build of the backward kernel (siegel_coop_bwd_kernel
// <16>, loop bound a kernel argument) came out with every pair wrong while the same source with a constant loop bound was
// right -- some legal-for-pure-functions motion of the fmac.  Volatile pins every DPP instruction to its place in the
// control flow and to the order of the other DPP instructions (measured cost in the forward kernels: 1-10 %).
// settle() stays plain asm: it is a per-lane identity and may move with its value.
#ifndef SYMPA_COOP_ASM
#define SYMPA_COOP_ASM asm volatile
#endif
#ifndef SYMPA_COOP_ASM_SETTLE
#define SYMPA_COOP_ASM_SETTLE asm
#endif

// value of lane J of my group of 16 lanes
template <int J>
__device__ __forceinline__ double bcast(const double v) {
#ifdef SYMPA_COOP_HALF
    static_assert(J < 8, "half groups: lanes 0..7");
    // Two bank-masked movs, written as asm: the pair of 64-bit __builtin_amdgcn_update_dpp calls with bank masks compiled to
    // movs that read the WRONG source register after an inline-asm producer (tools/microbench/half_group_check.hip: the
    // Cholesky pivots came from a stale register).  The wait states for the source travel inside the statement.
    double res;
    SYMPA_COOP_ASM("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\ts_nop 1\n\t"
                   "v_mov_b64_dpp %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xc bound_ctrl:1" : "=&v"(res) : "v"(v), "n"(J), "n"(J + 8));
    return res;
#else
    return __longlong_as_double(__builtin_amdgcn_update_dpp(0ll, __double_as_longlong(v), 0x150 + J, 0xf, 0xf, true));
#endif
}

// acc += x(lane J of my group) * y  /  acc -= ...   in one DP-ALU DPP instruction.
// Hazard: a DPP source VGPR must not have been written by the two preceding VALU instructions.  The compiler's hazard
// recogniser inserts the wait states for the DPP instructions it generates itself but sees neither the DPP read nor the
// VGPR write inside inline asm.  Two layers:
//  1. every value that is used as a DPP source is passed through settle() after its last write: an s_nop that the value
//     "depends on", which therefore sits between the program's own write and every DPP read;
//  2. the register allocator may still put a COPY of the source (v_mov_b64, v_accvgpr_read) right in front of the asm
//     statement -- found once: `v_mov_b64 v[30:31], v[26:27]; v_fmac_f64_dpp .., v[30:31], ..` in siegel_coop_bwd_kernel
//     <14>, most pairs wrong.  tools/check_dpp_hazards.py scans the generated ISA of every kernel for a VALU write of a
//     DPP source within two wait states (over all predecessors of a block); __graft_entry__.build() runs it on every
//     translation unit that includes this header and recompiles a unit that fails with SYMPA_COOP_NOP_IN_ASM, which
//     carries the two wait states inside every asm statement (always safe; measured 15-60 % slower, so not the default).
#if defined(SYMPA_COOP_NOP_IN_ASM) || defined(SYMPA_COOP_HALF)      // half groups: accumulators are read as "old" values too
#define SYMPA_COOP_NOP "s_nop 1\n\t"
#else
#define SYMPA_COOP_NOP ""
#endif
__device__ __forceinline__ double settle(double v) {
    SYMPA_COOP_ASM_SETTLE("s_nop 1" : "+v"(v));
    return v;
}
#ifdef SYMPA_COOP_HALF
template <int J>
__device__ __forceinline__ void fmac_bc(double& acc, const double x, const double y) {
    // The two instructions write the same register under complementary bank masks.  They must NOT be back to back: the
    // second one re-writes the lanes it has masked off with the value the register had BEFORE the first one (measured,
    // tools/microbench/dpp_bank_mask.hip: lanes 0-7 lose their update) -- the "old" value of a masked lane is read like a
    // DPP source, two wait states after a VALU write.
    SYMPA_COOP_ASM(SYMPA_COOP_NOP "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0x3\n\ts_nop 1\n\t"
                   "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xc" : "+v"(acc) : "v"(x), "v"(y), "n"(J), "n"(J + 8));
}
template <int J>
__device__ __forceinline__ void fnmac_bc(double& acc, const double x, const double y) {
    SYMPA_COOP_ASM(SYMPA_COOP_NOP "v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0x3\n\ts_nop 1\n\t"
                   "v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xc" : "+v"(acc) : "v"(x), "v"(y), "n"(J), "n"(J + 8));
}
#else
template <int J>
__device__ __forceinline__ void fmac_bc(double& acc, const double x, const double y) {
    SYMPA_COOP_ASM(SYMPA_COOP_NOP "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(J));
}
template <int J>
__device__ __forceinline__ void fnmac_bc(double& acc, const double x, const double y) {
    SYMPA_COOP_ASM(SYMPA_COOP_NOP "v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(J));
}
#endif

// sum over the 16 lanes of my group, result in every lane (32-bit DPP rotations: the DP ALU has no row_ror).
// The result is BITWISE the same in the sixteen lanes: rotations by 8, 4, 2, 1 add the same two numbers in both lanes of
// every exchange.  That only holds if the argument is a rounded value -- with -ffp-contract=fast the compiler would
// otherwise fuse a product in the caller's argument into the first addition (fma(t, t, other lane's ROUNDED t^2)), and
// the lanes of a pair would disagree in the last bit, which the redundant per-lane QL of spd_coop_bwd.hpp cannot
// tolerate (its predicates must agree).  The empty asm makes the argument opaque.
__device__ __forceinline__ double group_sum(double v) {
    asm("" : "+v"(v));
#define SPD_COOP_ROR(CTRL)                                                                          \
    {                                                                                               \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);    \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);    \
        v += __hiloint2double(hi, lo);                                                              \
    }
#ifdef SYMPA_COOP_HALF
    // eight lanes: exchange inside the quads (quad_perm [1,0,3,2], then [2,3,0,1]), then with the other quad of my half
    // (row_half_mirror: lane i <-> 7 - i; every lane of a quad holds the quad's sum by then) -- bitwise uniform as above
    SPD_COOP_ROR(0xB1) SPD_COOP_ROR(0x4E) SPD_COOP_ROR(0x141)
#else
    SPD_COOP_ROR(0x128) SPD_COOP_ROR(0x124) SPD_COOP_ROR(0x122) SPD_COOP_ROR(0x121)
#endif
#undef SPD_COOP_ROR
    return v;
}

// LDS slot (16 bytes = 2 doubles) of the pair of columns c = col / 2 of row r inside a 2 KB matrix image: the XOR makes
// the sixteen lanes that read one column hit sixteen different slots
__device__ __forceinline__ int tile_slot(const int r, const int c) { return r * 8 + (c ^ ((r >> 1) & 7)); }

// a <- a L^-T for the rows held one per lane:  a[j] = (a[j] - sum_{k<j} a[k] L[j][k]) / L[j][j];
// L[j][k] is register k of lane j.
template <int M>
__device__ __forceinline__ void solve_right_lt(double (&a)[M], const double (&l)[M], const double (&rd)[M]) {
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        sfor<0, j>([&](auto K) {
            constexpr int k = K;
            fnmac_bc<j>(a[j], l[k], a[k]);
        });
        a[j] *= rd[j];
    });
}

// The same for two right-hand sides at once (the Re and Im planes of a complex matrix against a real factor).  The DPP
// statements are `asm volatile` and keep their program order, so independent dependency chains have to be interleaved
// in the source: a[j] and b[j] are two accumulators instead of one.
template <int M>
__device__ __forceinline__ void solve_right_lt2(double (&a)[M], double (&b)[M], const double (&l)[M], const double (&rd)[M]) {
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        sfor<0, j>([&](auto K) {
            constexpr int k = K;
            fnmac_bc<j>(a[j], l[k], a[k]);
            fnmac_bc<j>(b[j], l[k], b[k]);
        });
        a[j] *= rd[j];
        b[j] *= rd[j];
    });
}

// Cholesky X = L L^T of the matrix held one row per lane, right-looking, in place: after step j, register j of
// lane i >= j holds L[i][j]; rd[j] = 1 / L[j][j] (group-uniform).  Returns "all pivots positive".
template <int M>
__device__ __forceinline__ bool cholesky_rows(double (&x)[M], double (&rd)[M]) {
    bool pd = true;
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        const double piv = bcast<j>(settle(x[j]));
        pd = pd && (piv > 0.0);
        const double rr = sympa::d_rsqrt(piv);
        rd[j] = rr;
        x[j] = settle(x[j] * rr);
        sfor<j + 1, M>([&](auto K) {
            constexpr int k = K;
            fnmac_bc<k>(x[k], x[j], x[j]);      // X[i][k] -= L[k][j] L[i][j]
        });
    });
    return pd;
}

// Two independent factorisations interleaved (Y1 and Y2 of a pair): the pivot's rsqrt chain of one hides behind the
// trailing update of the other.
template <int M>
__device__ __forceinline__ void cholesky_rows2(double (&x)[M], double (&rdx)[M], double (&y)[M], double (&rdy)[M], bool& pdx,
                                               bool& pdy) {
    pdx = true; pdy = true;
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        const double px = bcast<j>(settle(x[j]));
        const double py = bcast<j>(settle(y[j]));
        pdx = pdx && (px > 0.0);
        pdy = pdy && (py > 0.0);
        const double rx = sympa::d_rsqrt(px), ry = sympa::d_rsqrt(py);
        rdx[j] = rx; rdy[j] = ry;
        x[j] = settle(x[j] * rx);
        y[j] = settle(y[j] * ry);
        sfor<j + 1, M>([&](auto K) {
            constexpr int k = K;
            fnmac_bc<k>(x[k], x[j], x[j]);
            fnmac_bc<k>(y[k], y[j], y[j]);
        });
    });
}

// dst <- src in the lanes of `mask` only (a wave-uniform 64-bit lane mask), one VALU instruction: the select
// `dst = cond ? src : dst` costs two v_cndmask_b32 for a double.  EXEC is narrowed to (EXEC & mask) for the one move and
// restored from the saved copy, so the statement is correct under any EXEC it finds.
__device__ __forceinline__ void mov_in_lanes(double& dst, const double src, const unsigned long long mask) {
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %1, %3\n\tv_mov_b64 %0, %2\n\ts_mov_b64 exec, %1"
                 : "+v"(dst), "=&s"(saved) : "v"(src), "s"(mask) : "scc");
}

// X = Lh D Lh^T with Lh UNIT lower triangular, right-looking, in place: after step j, register j of lane i > j holds
// Lh[i][j] = L[i][j] / L[j][j].  No array of reciprocal diagonals (32 registers): lane j captures its own pivot D_j and
// `rdl` = 1 / sqrt(D_j) = 1 / L[j][j] is ONE value per lane, computed for all sixteen pivots at once after the loop; the
// triangular solves against a unit factor need no scaling, and  L^-1 A L^-T = D^-1/2 (Lh^-1 A Lh^-T) D^-1/2  puts the
// scaling into one per-lane multiply of the transposed intermediate and one broadcast multiply at the end.
template <int M>
__device__ __forceinline__ bool ldl_rows(double (&x)[M], double& rdl) {
    double mine = 1.0;
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        const double t = settle(x[j]);
        const double piv = bcast<j>(t);
        mov_in_lanes(mine, t, 0x0001000100010001ull << j);
        x[j] = settle(t * sympa::d_rcp(piv));
        sfor<j + 1, M>([&](auto K) {
            constexpr int k = K;
            fnmac_bc<k>(x[k], x[j], t);         // X[i][k] -= Lh[k][j] X[i][j]   (X[i][j] = Lh[i][j] D_j: the unscaled column)
        });
    });
    rdl = sympa::d_rsqrt(mine);
    // all pivots of my group positive (a NaN pivot compares false; the phantoms hold 1)
    const unsigned long long pos = __ballot(mine > 0.0);
    return ((pos >> (threadIdx.x & 48)) & 0xffffull) == 0xffffull;      // my group's sixteen bits
}

// a <- a Lh^-T for a unit lower factor:  a[j] -= sum_{k<j} a[k] Lh[j][k]
template <int M>
__device__ __forceinline__ void solve_right_unit(double (&a)[M], const double (&l)[M]) {
    sfor<0, M>([&](auto J) {
        constexpr int j = J;
        sfor<0, j>([&](auto K) {
            constexpr int k = K;
            fnmac_bc<j>(a[j], l[k], a[k]);
        });
    });
}

// m <- y^T for the rows held one per lane, through 2 KB of LDS private to my group
// STRIDE = doubles per row of the buffer.  With N = 16 (128 bytes) the sixteen lanes of a group write their 16-byte chunk c
// to addresses 128 bytes apart: two sets of banks for the whole wave, a 32-way conflict on every ds_write_b128
// (SQ_LDS_BANK_CONFLICT: 58 % of the LDS-busy cycles of the spd forward kernel).  STRIDE = TPAD = 18 (144 bytes, still
// 16-byte aligned) puts the sixteen rows of a group on sixteen different sets of four banks; the transposed read
// (sixteen consecutive doubles of one row) is conflict-free either way.  A group's buffer is TBUF doubles.
constexpr int TPAD = 18;
// doubles of LDS per group for transpose_rows.  The extra 128 (sixteen lanes per pair) / 16 (eight) bytes shift neighbouring
// groups onto complementary banks for the transposed read / the chunk write.
constexpr int TBUF = N * TPAD + (GROUP == 16 ? 16 : 2);
template <int M, int STRIDE = TPAD>
__device__ __forceinline__ void transpose_rows(const double (&y)[M], double (&m)[M], double* __restrict__ tbuf, const int r) {
    wave_lds_fence();
    sfor<0, M>([&](auto J) { tbuf[r * STRIDE + J] = y[J]; });
    wave_lds_fence();
    sfor<0, M>([&](auto J) { m[J] = tbuf[J * STRIDE + r]; });
}

// dst[0 .. M*M) += the M x M block whose row r is `v` in lane r (fp64 atomics), through the group's LDS tile so that
// consecutive lanes add to consecutive doubles (one 128-byte line per group and instruction instead of one per lane)
template <int M>
__device__ __forceinline__ void scatter_plane(const double (&v)[M], double* __restrict__ tile, double* __restrict__ dst,
                                              const int r, const bool on) {
    wave_lds_fence();
    if (r < M) sfor<0, M>([&](auto J) { tile[r * M + J] = v[J]; });
    wave_lds_fence();
    constexpr int CHUNKS = (M * M + GROUP - 1) / GROUP;
    sfor<0, CHUNKS>([&](auto K) {
        constexpr int k = K;
        const int idx = k * GROUP + r;
        const double val = (idx < M * M) ? tile[idx] : 0.0;
        if (on && val != 0.0) atomicAdd(dst + idx, val);
    });
}

// First half of a round: rows x (of X) and y (of Y) of my pair in; Cholesky factor (x, rd) and the rows m of
// W^T = L^-1 (Y - X) out.  `tbuf` = TBUF doubles of LDS private to my group for the transpose.  Returns "X is PD".
template <int M>
__device__ __forceinline__ bool reduce_pair_front(double (&x)[M], double (&y)[M], double& rdl, double (&m)[M],
                                                  double* __restrict__ tbuf, const int r) {
    sfor<0, M>([&](auto J) { y[J] -= x[J]; });      // A = Y - X
    const bool pd = ldl_rows(x, rdl);
    solve_right_unit(y, x);                          // B = A Lh^-T     (W = B D^-1/2)
    transpose_rows(y, m, tbuf, r);
    sfor<0, M>([&](auto J) { m[J] *= rdl; });        // row j of W^T = column j of B / L[j][j]
    return pd;
}

// The same with the factorisation taken from the packed table (spd.hip, PACKED): x = the factor rows Lh, rdl = D^-1/2 of my row (NaN
// when the point was not positive definite at pack time), y = A = Y - X already.
template <int M>
__device__ __forceinline__ bool reduce_pair_front_factored(const double (&x)[M], double (&y)[M], const double rdl, double (&m)[M],
                                                           double* __restrict__ tbuf, const int r) {
    solve_right_unit(y, x);
    transpose_rows(y, m, tbuf, r);
    sfor<0, M>([&](auto J) { m[J] *= rdl; });
    const unsigned long long fin = __ballot(rdl == rdl);
    return ((fin >> (threadIdx.x & 48)) & 0xffffull) == 0xffffull;
}

// Trailing block handed to the one-pair-per-lane phase (spd_math.hpp tridiag_packed).  A Householder step in the
// row-per-lane layout costs ~75 wave instructions of group-uniform scalar work and reductions next to its 3 (M - k - 1)
// useful DPP FMAs, and serves FOUR pairs; the same step one pair per lane serves 64.  So only the first M - TB steps
// run here; the TB x TB block that is left goes through the LDS to the lane that keeps the pair (36 doubles for
// TB = 8 -- what the register budget of two waves per SIMD allows), which finishes the tridiagonalisation after the
// last round together with the 63 other pairs of the wave.
template <int M>
constexpr int trailing_block() {
    return M < 10 ? M : 10;
}
template <int TB>
constexpr int packed_len() { return TB >= 3 ? TB * (TB + 1) / 2 : 1; }

// The packed lower triangle of the block my group left in `hand`.  Call it under `if (lane keeps that pair)`: a real
// branch (55 loads under the EXEC mask for TB = 10), the other lanes keep the blocks of their own pairs.
template <int TB>
__device__ __forceinline__ void take_block(double (&blk)[packed_len<TB>()], const double* __restrict__ hand) {
    sfor<0, TB>([&](auto I) {
        constexpr int i = I;
        sfor<0, i + 1>([&](auto J) { blk[i * (i + 1) / 2 + J] = hand[i * TB + J]; });
    });
}

// Second half: M = W^T L^-T = L^-1 (Y - X) L^-T, then the first M - TB steps of its tridiagonalisation: (d, e2)[0 .. M - TB)
// kept by the lane with keep = true (`keepmask`: those lanes as a wave mask), and the trailing block into `hand`
// (TB * TB doubles of LDS private to my group).  TB < 3: all of it here.
template <int M, int TB>
__device__ __forceinline__ void reduce_pair_back(double (&m)[M], const double (&x)[M], const double rdl,
                                                 const int r, const bool keep, const unsigned long long keepmask,
                                                 double (&d)[M], double (&e2)[M], double* __restrict__ hand,
                                                 double* __restrict__ hand_dummy) {
    solve_right_unit(m, x);
    {
        const double rs = settle(rdl);
        sfor<0, M>([&](auto J) { m[J] *= bcast<J>(rs); });      // column j / L[j][j]
    }
    constexpr int KS = (TB >= 3) ? M - TB : M - 2;      // steps taken here

    // Householder tridiagonalisation.  The reflector of step k is taken from COLUMN k, one element per lane (my
    // own register k), and broadcast from there for every use: with a single source for v the update is an exact
    // similarity whatever rounding-level asymmetry M carries.  (Mixing row k of lane k with my own column element
    // is inconsistent by that asymmetry RELATIVE TO |v|, which is large when the eliminated column is small.)
    sfor<0, KS>([&](auto K) {
        constexpr int k = K;
        const double col = settle(m[k]);
        const double x0 = bcast<k + 1>(col);
        const double dk = bcast<k>(col);
        const double tail = (r > k + 1 && r < M) ? col : 0.0;
        const double s2 = group_sum(tail * tail);
        const double n2 = sympa::d_fma(x0, x0, s2);
        mov_in_lanes(d[k], dk, keepmask);
        mov_in_lanes(e2[k], n2, keepmask);
        const double nx = sympa::d_sqrt(n2);
        const double v0 = x0 + copysign(nx, x0);
        const double den = sympa::d_fma(v0, v0, s2);
        // a zero column (den = 0): v = 0, so p = q = 0 whatever the (finite) beta
        const double hb = sympa::d_rcp(fmax(den, 1e-300));         // beta / 2
        const double beta = hb + hb;
        const double vi = settle((r == k + 1) ? v0 : tail);        // 0 in the finished rows (r <= k) and the phantoms
        // p_i = beta sum_j M[i][j] v_j
        double ps[2] = {0.0, 0.0};
        sfor<k + 1, M>([&](auto J) { fmac_bc<J>(ps[J % 2], vi, m[J]); });
        double p = beta * (ps[0] + ps[1]);
        // a phantom row (r >= M) holds whatever the LDS held where nobody wrote (its transposed read): possibly NaN, and
        // 0 * NaN would poison the sum below
        if constexpr (M < N) p = (r < M) ? p : 0.0;
        const double kk = hb * group_sum(vi * p);
        const double q = settle(sympa::d_fma(-kk, vi, p));
        // M <- M - q v^T - v q^T.  Not masked to the trailing block: a finished row r <= k (v_r = 0) receives -v_j q_r,
        // which is what the two-sided reflection does to it (its eliminated entries); a phantom row (p = q = 0) keeps
        // whatever it holds.  Neither is read again.
        sfor<k + 1, M>([&](auto J) {
            constexpr int j = J;
            fnmac_bc<j>(m[j], vi, q);       // - v_j q_i
            fnmac_bc<j>(m[j], q, vi);       // - q_j v_i
        });
    });
    if constexpr (TB >= 3) {
        // rows KS .. M-1 of the trailing block, one per lane -> the LDS -> packed lower triangle in the keeping lane.
        // The keeping lane collects it (take_block) at the start of the NEXT round, behind that round's loads, so that
        // nobody waits for these 55 loads here.
        wave_lds_fence();       // the previous round's block has been collected
        // No branch around the stores (a divergent region here costs the register allocator 20-40 registers, measured on the
        // Siegel kernels): the lanes outside the block store their junk into the dummy row `hand_dummy` (one for the wave).
        {
            double* const row = (r >= KS && r < M) ? hand + (r - KS) * TB : hand_dummy;
            sfor<KS, M>([&](auto J) { row[J - KS] = m[J]; });
        }
    } else {
        const double last = settle(m[M - 1]);
        const double dm = bcast<M - 2>(settle(m[M - 2]));
        const double dn = bcast<M - 1>(last);
        const double en = bcast<M - 2>(last);
        mov_in_lanes(d[M - 2], dm, keepmask);
        mov_in_lanes(d[M - 1], dn, keepmask);
        mov_in_lanes(e2[M - 2], en * en, keepmask);
    }
}

}  // namespace spd_coop
