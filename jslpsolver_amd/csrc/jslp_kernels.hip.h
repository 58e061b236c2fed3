// jslp_kernels.hip.h -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the dense-tableau simplex.
//
// Data layout in HBM: the tableau is row-major fp64 with row stride `ld` = width rounded up to
// 16 doubles (128 B), row 0 = reduced costs, column 0 = RHS -- the reference layout (tableau.ts:49-54) with
// cache-line aligned rows so that every wave issues 16-byte coalesced loads along a row.
//
// One pivot of the reference (simplex.ts:25-413) is split into two device steps:
//   select_step  (one workgroup): leaving-row / entering-column selection with wave-shuffle + LDS
//                reductions carrying (value, index) so the reference's strict-compare FIRST-INDEX
//                tie-breaks are reproduced, cycle check, then "prepare": gather the pivot column into
//                pcol[], normalise the pivot row into prow[] (and in place), swap the index maps.
//   update step  (whole chip): A[r,c] = A[r,c] - pcol[r]*prow[c] streamed with double2 loads, both
//                roundings kept (no FMA contraction: __dmul_rn / __dsub_rn), IEEE division for column c*.
// The same device functions serve two launch shapes:
//   * k_select + k_update: one tableau spread over all CUs (large LPs, HBM-bandwidth bound)
//   * k_simplex_wg: one WORKGROUP runs a whole simplex() for one tableau (small tableaus and batches of
//     independent branch-and-bound nodes: grid = #nodes, no host round trip per pivot).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define JSLP_WG_THREADS 1024      // workgroup size of the select step / the per-node simplex kernel
#define JSLP_UPD_THREADS 256      // workgroup size of the streaming row update
#define JSLP_UPD_ROWS 8           // rows per workgroup in the streaming row update
#define JSLP_UPD_COLS (JSLP_UPD_THREADS * 2)

enum { ST_RUNNING = 0, ST_DONE = 1 };
enum { ERR_NONE = 0, ERR_HIST_FULL = 1, ERR_ITER_LIMIT = 2, ERR_CUT_ARG = 3, ERR_CAPACITY = 4 };

// Per-tableau device state (one per slot).  Plain ints so the host can read it back with one copy.
struct DevState {
    int32_t H;                  // current height (grows with cuts)
    int32_t last_element_index; // getNewElementIndex counter (tableau.ts:393-401)
    int32_t status;             // ST_*
    int32_t phase;              // 1 or 2
    int32_t feasible, bounded, optimal, unbounded_var;
    int32_t it1, it2;           // pivots in phase 1 / phase 2
    int32_t entered_phase2;
    int32_t cycle_phase;        // 0, 1, 2
    int32_t hist_n;             // entries of the current phase's (leaving, entering) history
    int32_t do_pivot;           // select_step chose a pivot for the update step
    int32_t pr, pc;             // pivot row / column
    int32_t err;                // ERR_*
    int32_t iters_left;         // safety cap against endless cycling when the cycle check is off
    double quot;                // A[pr,pc] before normalisation
    double obj_cell;            // A[0,0] when the call ended
    long long trace_n;          // pivots recorded since upload
    // snapshot scalars (savedState)
    int32_t s_H, s_last_element_index;
};

// Everything a step needs for ONE tableau.
struct Ctx {
    double* A;
    int32_t* vibr;
    int32_t* vibc;
    int32_t* rbv;
    int32_t* cbv;
    const uint8_t* unr;
    double* prow;
    double* pcol;
    DevState* st;
    int2* hist;
    int2* trace;
    long long trace_cap;
    int32_t hist_cap;
    int32_t ld, W;
    int32_t check_cycles;
    int32_t batch;        // partial-pricing batch size (simplex.ts:118-124)
    int32_t use_partial;  // simplex.ts:127
    double precision;
};

// Base pointers + per-slot strides: slot s of a batch owns the s-th tableau copy.
struct Slots {
    double* A;       long long A_stride;
    int32_t* vibr;   int32_t vibr_stride;
    int32_t* vibc;   int32_t vibc_stride;
    int32_t* rbv;    int32_t idx_stride;
    int32_t* cbv;
    const uint8_t* unr;
    double* prow;    int32_t prow_stride;
    double* pcol;    int32_t pcol_stride;
    DevState* st;
    int2* hist;      int32_t hist_cap;
    int2* trace;     long long trace_cap;   // only slot 0 traces
    int32_t ld, W;
    int32_t batch, use_partial;
    double precision;
};

__device__ __forceinline__ Ctx slot_ctx(const Slots& s, int slot, int check_cycles) {
    Ctx c;
    c.A = s.A + (long long)slot * s.A_stride;
    c.vibr = s.vibr + (long long)slot * s.vibr_stride;
    c.vibc = s.vibc + (long long)slot * s.vibc_stride;
    c.rbv = s.rbv + (long long)slot * s.idx_stride;
    c.cbv = s.cbv + (long long)slot * s.idx_stride;
    c.unr = s.unr;
    c.prow = s.prow + (long long)slot * s.prow_stride;
    c.pcol = s.pcol + (long long)slot * s.pcol_stride;
    c.st = s.st + slot;
    c.hist = s.hist + (long long)slot * s.hist_cap;
    c.hist_cap = s.hist_cap;
    c.trace = s.trace;
    c.trace_cap = slot == 0 ? s.trace_cap : 0;
    c.ld = s.ld;
    c.W = s.W;
    c.check_cycles = check_cycles;
    c.batch = s.batch;
    c.use_partial = s.use_partial;
    c.precision = s.precision;
    return c;
}

// the reference's zero test `!(v >= -1e-16 && v <= 1e-16)` (simplex.ts:356,372,379): NaN counts as non-zero
__device__ __forceinline__ bool nonzero16(double v) { return !(v >= -1e-16 && v <= 1e-16); }

// ---------------------------------------------------------------------------------------------------
// (value, index) candidates and their reductions.  `i == 0` means "no candidate" (row/column 0 is never
// selectable).  All orders are total on (key..., index) so the result does not depend on thread mapping.
// ---------------------------------------------------------------------------------------------------
struct Cand {
    double v;
    int32_t i;
    int32_t b;  // pricing batch id (phase 2), otherwise 0
};

// strict "a is better than b" orders ------------------------------------------------------------------
struct MinFirst {  // smallest value, first index on ties (phase-1 row, ratio test)
    __device__ __forceinline__ bool operator()(const Cand& a, const Cand& b) const {
        if (a.i == 0) return false;
        if (b.i == 0) return true;
        return a.v < b.v || (a.v == b.v && a.i < b.i);
    }
};
struct MaxFirst {  // largest value, first index on ties (phase-1 column)
    __device__ __forceinline__ bool operator()(const Cand& a, const Cand& b) const {
        if (a.i == 0) return false;
        if (b.i == 0) return true;
        return a.v > b.v || (a.v == b.v && a.i < b.i);
    }
};
struct PriceFirst {  // first batch holding a candidate, best value inside it, first index on ties
    __device__ __forceinline__ bool operator()(const Cand& a, const Cand& b) const {
        if (a.i == 0) return false;
        if (b.i == 0) return true;
        if (a.b != b.b) return a.b < b.b;
        return a.v > b.v || (a.v == b.v && a.i < b.i);
    }
};

template <class Better>
__device__ __forceinline__ Cand wave_reduce(Cand x, Better better) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Cand y;
        y.v = __shfl_down(x.v, off, 64);
        y.i = __shfl_down(x.i, off, 64);
        y.b = __shfl_down(x.b, off, 64);
        if (better(y, x)) x = y;
    }
    return x;
}

struct Smem {
    Cand wave[JSLP_WG_THREADS / 64];
    Cand result;
    int32_t flag;
    int32_t flag2;
};

// Block-wide reduction: wave shuffles, then the 16 wave leaders through LDS.  Returns the winner to all threads.
template <class Better>
__device__ __forceinline__ Cand block_reduce(Cand x, Better better, Smem& sm) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    x = wave_reduce(x, better);
    __syncthreads();  // protects sm.wave / sm.result from the previous use
    if (lane == 0) sm.wave[w] = x;
    __syncthreads();
    if (w == 0) {
        Cand y;
        if (lane < nw) y = sm.wave[lane];
        else { y.v = 0; y.i = 0; y.b = 0; }
        y = wave_reduce(y, better);
        if (lane == 0) sm.result = y;
    }
    __syncthreads();
    return sm.result;
}

// ---------------------------------------------------------------------------------------------------
// prepare_pivot: steps 1-2 of pivot() (simplex.ts:330-364) + the bookkeeping, for the pivot (pr, pc).
// pcol[] must already hold column pc when `pcol_ready`.  Ends with st->do_pivot = 1.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void prepare_pivot(const Ctx& c, int pr, int pc, bool pcol_ready, Smem& sm) {
    DevState* st = c.st;
    const int H = st->H, W = c.W, ld = c.ld;
    const int tid = threadIdx.x, nt = blockDim.x;
    double* A = c.A;
    const double quot = A[(long long)pr * ld + pc];  // simplex.ts:335
    int any = 0;
    for (int r = tid; r < H; r += nt) {
        double k;
        if (pcol_ready && r > 0) {
            k = c.pcol[r];
        } else {
            k = A[(long long)r * ld + pc];
            c.pcol[r] = k;
        }
        any |= (r != pr && nonzero16(k));
    }
    // any row that will execute the inner loop of simplex.ts:367-391 lazily zeroes the tiny pivot-row entries
    const int anyrow = __syncthreads_or(any);  // also orders the quot read above against the row write below
    double* prow_A = A + (long long)pr * ld;
    for (int col = tid; col < ld; col += nt) {
        double v = 0.0;
        if (col < W) {
            const double val = prow_A[col];
            const bool innz = nonzero16(val);       // :356
            v = innz ? val / quot : 0.0;            // :357 / :361  (IEEE division)
            bool in_list = innz;
            if (col == pc) v = 1.0 / quot;          // :364 (membership of pc in nonZeroColumns is decided by `val`)
            if (in_list && anyrow && !nonzero16(v) && v != 0.0) v = 0.0;  // :381-383
            prow_A[col] = v;
        }
        c.prow[col] = v;
    }
    if (tid == 0) {
        const int leaving = c.vibr[pr], entering = c.vibc[pc];  // :339-349
        c.vibr[pr] = entering;
        c.vibc[pc] = leaving;
        c.rbv[entering] = pr;
        c.rbv[leaving] = -1;
        c.cbv[entering] = -1;
        c.cbv[leaving] = pc;
        if (st->trace_n < c.trace_cap) c.trace[st->trace_n] = make_int2(pr, pc);
        st->trace_n += 1;
        st->pr = pr;
        st->pc = pc;
        st->quot = quot;
        st->do_pivot = 1;
    }
    __syncthreads();
}

// checkForCycles (simplex.ts:415-440).  The check runs after every append and the phase stops at the first
// hit, so a NEW repeated block always ends at the newest entry: "the history's suffix is a square XX".
// (hit / no-hit is what drives the solver; the exact [start, length] message is rebuilt on the host.)
__device__ __forceinline__ bool suffix_is_square(const int2* h, int n, Smem& sm) {
    int found = 0;
    const int2 last = h[n - 1];
    for (int L = 1 + threadIdx.x; 2 * L <= n; L += blockDim.x) {
        const int2 a = h[n - 1 - L];
        if (a.x != last.x || a.y != last.y) continue;
        bool eq = true;
        for (int i = 0; i < L - 1; i++) {
            const int2 x = h[n - 2 * L + i], y = h[n - L + i];
            if (x.x != y.x || x.y != y.y) { eq = false; break; }
        }
        if (eq) found = 1;
    }
    return __syncthreads_or(found) != 0;
}

__device__ __forceinline__ void finish(const Ctx& c) {  // thread 0 only
    c.st->status = ST_DONE;
    c.st->do_pivot = 0;
    c.st->obj_cell = c.A[0];
}

// ---------------------------------------------------------------------------------------------------
// select_step: one iteration of the phase-1 / phase-2 `while (true)` loops up to (not including) the row
// elimination.  Executed by ONE workgroup; control flow is uniform (decisions come from block reductions).
// ---------------------------------------------------------------------------------------------------
__device__ void select_step(const Ctx& c, Smem& sm) {
    DevState* st = c.st;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int H = st->H, W = c.W, ld = c.ld;
    const double precision = c.precision;
    const double* A = c.A;
    int phase = st->phase;
    if (tid == 0) st->do_pivot = 0;
    if (st->iters_left <= 0) {
        if (tid == 0) { st->err = ERR_ITER_LIMIT; finish(c); }
        __syncthreads();
        return;
    }
    int pr = 0, pc = 0;
    bool pcol_ready = false;

    if (phase == 1) {
        // leaving row: most negative RHS below -precision, first index on ties (simplex.ts:39-49)
        Cand best; best.v = -precision; best.i = 0; best.b = 0;
        for (int r = 1 + tid; r < H; r += nt) {
            const double v = A[(long long)r * ld];
            if (v < best.v) { best.v = v; best.i = r; }
        }
        best = block_reduce(best, MinFirst(), sm);
        if (best.i == 0) {  // :51-54 -> feasible; phase 2 starts in this same step
            phase = 2;
            if (tid == 0) { st->feasible = 1; st->phase = 2; st->entered_phase2 = 1; st->hist_n = 0; }
            __syncthreads();
        } else {
            pr = best.i;
            // entering column: max -cost/coef over unrestricted or coef < -precision (simplex.ts:56-71)
            const double* row = A + (long long)pr * ld;
            Cand q; q.v = -INFINITY; q.i = 0; q.b = 0;
            for (int col = 1 + tid; col < W; col += nt) {
                const double coef = row[col];
                const bool un = c.unr[c.vibc[col]] != 0;
                if (un || coef < -precision) {
                    const double quo = -A[col] / coef;
                    if (q.v < quo) { q.v = quo; q.i = col; }
                }
            }
            q = block_reduce(q, MaxFirst(), sm);
            if (q.i == 0) {  // :73-76 infeasible
                if (tid == 0) { st->feasible = 0; finish(c); }
                __syncthreads();
                return;
            }
            pc = q.i;
        }
    }

    if (phase == 2) {
        // Dantzig pricing with the reference's batch rule (simplex.ts:118-219): the first batch
        // [1..B], [B+1..2B], ... that holds a candidate wins; inside it the largest value, first index.
        Cand e; e.v = precision; e.i = 0; e.b = 0;
        int neg_flag = 0;
        for (int col = 1 + tid; col < W; col += nt) {
            const double rc = A[col];
            const bool un = c.unr[c.vibc[col]] != 0;
            const int b = c.use_partial ? (col - 1) / c.batch : 0;
            double val; int ng;
            if (un && rc < 0) { val = -rc; ng = 1; } else { val = rc; ng = 0; }
            // per-thread running best in the same total order as the reduction; a candidate must beat
            // `precision` (strict >), which every thread applies itself
            if (val > precision) {
                Cand cand; cand.v = val; cand.i = col; cand.b = b;
                if (PriceFirst()(cand, e)) { e = cand; neg_flag = ng; }
            }
        }
        e = block_reduce(e, PriceFirst(), sm);
        if (e.i == 0) {  // optimal (simplex.ts:265-269); setEvaluation happens on the host from obj_cell
            if (tid == 0) { st->optimal = 1; finish(c); }
            __syncthreads();
            return;
        }
        pc = e.i;
        // isReducedCostNegative of the winner: recompute (cheap, uniform)
        {
            const double rc = A[pc];
            const bool un = c.unr[c.vibc[pc]] != 0;
            neg_flag = (un && rc < 0) ? 1 : 0;
        }
        // ratio test (simplex.ts:271-296) in its order-free form: r_deg = first row passing the
        // degenerate test wins outright, otherwise first-index argmin of the accepted quotients.
        Cand m; m.v = INFINITY; m.i = 0; m.b = 0;
        int rdeg = 0x7fffffff;
        for (int r = tid; r < H; r += nt) {
            const double colv = A[(long long)r * ld + pc];
            c.pcol[r] = colv;  // the update step needs the whole column anyway (row 0 included)
            if (r == 0) continue;
            const double rhs = A[(long long)r * ld];
            if (-precision < colv && colv < precision) continue;
            if (colv > 0 && precision > rhs && rhs > -precision) {
                if (r < rdeg) rdeg = r;
                continue;
            }
            const double quo = neg_flag ? -rhs / colv : rhs / colv;
            if (quo > precision && m.v > quo) { m.v = quo; m.i = r; }
        }
        // min over rdeg
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_down(rdeg, off, 64);
            rdeg = o < rdeg ? o : rdeg;
        }
        __syncthreads();
        if (tid == 0) sm.flag = 0x7fffffff;
        __syncthreads();
        if ((tid & 63) == 0 && rdeg != 0x7fffffff) atomicMin(&sm.flag, rdeg);
        m = block_reduce(m, MinFirst(), sm);  // contains the barriers that publish sm.flag
        rdeg = sm.flag;
        if (rdeg != 0x7fffffff) {
            pr = rdeg;
        } else if (m.i != 0) {
            pr = m.i;
        } else {  // unbounded (simplex.ts:298-303)
            if (tid == 0) { st->bounded = 0; st->unbounded_var = c.vibc[pc]; finish(c); }
            __syncthreads();
            return;
        }
        pcol_ready = true;
    }

    // cycle check (simplex.ts:78-93 / 305-320): append first, test, stop WITHOUT pivoting on a hit
    if (c.check_cycles) {
        const int n = st->hist_n;
        if (n >= c.hist_cap) {
            if (tid == 0) { st->err = ERR_HIST_FULL; finish(c); }
            __syncthreads();
            return;
        }
        if (tid == 0) {
            c.hist[n] = make_int2(c.vibr[pr], c.vibc[pc]);
            st->hist_n = n + 1;
        }
        __syncthreads();
        if (suffix_is_square(c.hist, n + 1, sm)) {
            if (tid == 0) { st->cycle_phase = phase; st->feasible = 0; finish(c); }
            __syncthreads();
            return;
        }
    }

    prepare_pivot(c, pr, pc, pcol_ready, sm);
    if (tid == 0) {
        if (phase == 1) st->it1 += 1; else st->it2 += 1;
        st->iters_left -= 1;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------
// Row elimination (simplex.ts:367-391) for a set of rows.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double eliminate(double a, double k, double p) {
    // matrix[r,c] - coefficient * v0 with BOTH roundings (JavaScript never fuses)
    return __dsub_rn(a, __dmul_rn(k, p));
}

// Whole chip, one launch per pivot: workgroup (bx, by) owns rows [by*8, by*8+8) x columns [bx*512, +512).
// Every lane keeps its two pivot-row values in registers, the 8 loads of a lane are issued back to back
// (16 B each, 1 KiB per wave-instruction, fully coalesced) before the first dependent use.
__global__ void __launch_bounds__(JSLP_UPD_THREADS) k_update(Ctx c) {
    const DevState* st = c.st;
    if (st->status != ST_RUNNING || !st->do_pivot) return;
    const int H = st->H, ld = c.ld;
    const int pr = st->pr, pc = st->pc;
    const double quot = st->quot;
    const int c0 = blockIdx.x * JSLP_UPD_COLS + threadIdx.x * 2;
    if (c0 >= ld) return;
    const double2 p = *reinterpret_cast<const double2*>(c.prow + c0);
    const bool v0 = nonzero16(p.x), v1 = nonzero16(p.y);
    const bool has_pc = (pc == c0) || (pc == c0 + 1);
    if (!v0 && !v1 && !has_pc) return;  // nothing in these two columns changes (sparse pivot rows)
    const int r0 = blockIdx.y * JSLP_UPD_ROWS;
    double k[JSLP_UPD_ROWS];
    bool act[JSLP_UPD_ROWS];
    double2 a[JSLP_UPD_ROWS];
#pragma unroll
    for (int i = 0; i < JSLP_UPD_ROWS; i++) {
        const int r = r0 + i;
        k[i] = (r < H) ? c.pcol[r] : 0.0;
        act[i] = (r < H) && (r != pr) && nonzero16(k[i]);  // :370-375 row gate
    }
#pragma unroll
    for (int i = 0; i < JSLP_UPD_ROWS; i++)
        if (act[i]) a[i] = *reinterpret_cast<const double2*>(c.A + (long long)(r0 + i) * ld + c0);
#pragma unroll
    for (int i = 0; i < JSLP_UPD_ROWS; i++) {
        if (!act[i]) continue;
        double2 x = a[i];
        if (v0) x.x = eliminate(x.x, k[i], p.x);
        if (v1) x.y = eliminate(x.y, k[i], p.y);
        if (has_pc) {  // :387 overwrites whatever the loop did to column pc
            const double nv = -k[i] / quot;
            if (pc == c0) x.x = nv; else x.y = nv;
        }
        *reinterpret_cast<double2*>(c.A + (long long)(r0 + i) * ld + c0) = x;
    }
}

// The same elimination done by ONE workgroup (per-node kernel): waves take rows, lanes take column pairs.
__device__ __forceinline__ void update_rows_wg(const Ctx& c) {
    const DevState* st = c.st;
    const int H = st->H, ld = c.ld, pr = st->pr, pc = st->pc;
    const double quot = st->quot;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int r = w; r < H; r += nw) {
        const double k = c.pcol[r];
        if (r == pr || !nonzero16(k)) continue;  // wave-uniform gate
        double* row = c.A + (long long)r * ld;
        for (int c0 = lane * 2; c0 < ld; c0 += 128) {
            const double2 p = *reinterpret_cast<const double2*>(c.prow + c0);
            const bool v0 = nonzero16(p.x), v1 = nonzero16(p.y);
            const bool has_pc = (pc == c0) || (pc == c0 + 1);
            if (!v0 && !v1 && !has_pc) continue;
            double2 x = *reinterpret_cast<const double2*>(row + c0);
            if (v0) x.x = eliminate(x.x, k, p.x);
            if (v1) x.y = eliminate(x.y, k, p.y);
            if (has_pc) {
                const double nv = -k / quot;
                if (pc == c0) x.x = nv; else x.y = nv;
            }
            *reinterpret_cast<double2*>(row + c0) = x;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Kernels
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(JSLP_WG_THREADS) k_select(Ctx c) {
    __shared__ Smem sm;
    if (c.st->status != ST_RUNNING) return;
    select_step(c, sm);
}

// Tableau.pivot(r, c) on its own
__global__ void __launch_bounds__(JSLP_WG_THREADS) k_prepare(Ctx c, int pr, int pc) {
    __shared__ Smem sm;
    if (threadIdx.x == 0) c.st->status = ST_RUNNING;
    __syncthreads();
    prepare_pivot(c, pr, pc, false, sm);
}
__global__ void k_end_pivot(Ctx c) {
    c.st->status = ST_DONE;
    c.st->do_pivot = 0;
}

// simplex() entry: `this.bounded = true; phase1(); if (feasible) phase2()` (simplex.ts:14-23)
__device__ __forceinline__ void begin_simplex(DevState* st, int iters_cap) {
    st->status = ST_RUNNING;
    st->phase = 1;
    st->bounded = 1;
    st->optimal = 0;
    st->unbounded_var = -1;
    st->it1 = 0;
    st->it2 = 0;
    st->entered_phase2 = 0;
    st->cycle_phase = 0;
    st->hist_n = 0;
    st->do_pivot = 0;
    st->err = st->err == ERR_CUT_ARG || st->err == ERR_CAPACITY ? st->err : ERR_NONE;
    st->iters_left = iters_cap;
}
__global__ void k_begin(Slots s, int first_slot, int iters_cap) {
    begin_simplex(s.st + first_slot + blockIdx.x, iters_cap);
}

// One workgroup = one whole simplex() on one tableau (slot first_slot + blockIdx.x).
__global__ void __launch_bounds__(JSLP_WG_THREADS) k_simplex_wg(Slots s, int first_slot, int check_cycles, int iters_cap) {
    __shared__ Smem sm;
    const Ctx c = slot_ctx(s, first_slot + blockIdx.x, check_cycles);
    if (threadIdx.x == 0) begin_simplex(c.st, iters_cap);
    __syncthreads();
    if (c.st->err != ERR_NONE) {  // a bad cut list: report, do not solve
        if (threadIdx.x == 0) finish(c);
        return;
    }
    for (;;) {
        select_step(c, sm);
        if (!c.st->do_pivot) break;
        update_rows_wg(c);
        __syncthreads();
    }
}

// restore (backup.ts:53-105): snapshot -> slots [first_slot, first_slot+gridDim.y).  Grid-stride copy.
struct Snapshot {
    const double* A;
    const int32_t *vibr, *vibc, *rbv, *cbv;
    int32_t n_idx;
};
__global__ void __launch_bounds__(256) k_restore(Slots s, Snapshot snap, int first_slot) {
    const int slot = first_slot + blockIdx.y;
    DevState* st = s.st + slot;
    const int H = s.st[0].s_H;  // every slot shares slot 0's snapshot scalars
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long long)gridDim.x * blockDim.x;
    const long long n2 = (long long)H * s.ld / 2;
    const double2* src = reinterpret_cast<const double2*>(snap.A);
    double2* dst = reinterpret_cast<double2*>(s.A + (long long)slot * s.A_stride);
    for (long long i = tid; i < n2; i += nt) dst[i] = src[i];
    int32_t* vibr = s.vibr + (long long)slot * s.vibr_stride;
    for (long long i = tid; i < H; i += nt) vibr[i] = snap.vibr[i];
    int32_t* vibc = s.vibc + (long long)slot * s.vibc_stride;
    for (long long i = tid; i < s.W; i += nt) vibc[i] = snap.vibc[i];
    int32_t* rbv = s.rbv + (long long)slot * s.idx_stride;
    int32_t* cbv = s.cbv + (long long)slot * s.idx_stride;
    for (long long i = tid; i < snap.n_idx; i += nt) { rbv[i] = snap.rbv[i]; cbv[i] = snap.cbv[i]; }
    if (tid == 0) {
        st->H = H;
        st->last_element_index = s.st[0].s_last_element_index;
        st->err = ERR_NONE;
    }
}

// save (backup.ts:13-51): slot 0 -> snapshot
struct SnapshotW {
    double* A;
    int32_t *vibr, *vibc, *rbv, *cbv;
    int32_t n_idx;
};
__global__ void __launch_bounds__(256) k_save(Slots s, SnapshotW snap) {
    DevState* st = s.st;
    const int H = st->H;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long long)gridDim.x * blockDim.x;
    const long long n2 = (long long)H * s.ld / 2;
    const double2* src = reinterpret_cast<const double2*>(s.A);
    double2* dst = reinterpret_cast<double2*>(snap.A);
    for (long long i = tid; i < n2; i += nt) dst[i] = src[i];
    for (long long i = tid; i < H; i += nt) snap.vibr[i] = s.vibr[i];
    for (long long i = tid; i < s.W; i += nt) snap.vibc[i] = s.vibc[i];
    for (long long i = tid; i < snap.n_idx; i += nt) { snap.rbv[i] = s.rbv[i]; snap.cbv[i] = s.cbv[i]; }
    if (tid == 0) {
        st->s_H = H;
        st->s_last_element_index = st->last_element_index;
    }
}

// addCutConstraints (cutting-strategies.ts:16-72): node (first_node + blockIdx.x) appends its cuts
// [offs[n], offs[n+1]) to slot (first_slot + blockIdx.x).
struct Cuts {
    const int32_t* offs;
    const int8_t* type;
    const int32_t* var;
    const double* value;
};
__global__ void __launch_bounds__(256) k_add_cuts(Slots s, Cuts cuts, int first_slot, int first_node, int cap_rows) {
    const int slot = first_slot + blockIdx.x, node = first_node + blockIdx.x;
    DevState* st = s.st + slot;
    double* A = s.A + (long long)slot * s.A_stride;
    int32_t* vibr = s.vibr + (long long)slot * s.vibr_stride;
    int32_t* rbv = s.rbv + (long long)slot * s.idx_stride;
    int32_t* cbv = s.cbv + (long long)slot * s.idx_stride;
    const int a = cuts.offs[node], n = cuts.offs[node + 1] - a;
    const int H = st->H, W = s.W, ld = s.ld;
    if (H + n > cap_rows) {
        if (threadIdx.x == 0) st->err = ERR_CAPACITY;
        return;
    }
    for (int h = 0; h < n; h++) {
        const int vi = cuts.var[a + h];
        const double sign = cuts.type[a + h] == 0 ? -1.0 : 1.0;  // "min" -> -1 (:41)
        const double value = cuts.value[a + h];
        double* cut = A + (long long)(H + h) * ld;
        const int var_row = (vi >= 0 && vi < s.idx_stride) ? rbv[vi] : -2;
        const int var_col = (vi >= 0 && vi < s.idx_stride) ? cbv[vi] : -1;
        if (var_row == -2 || (var_row == -1 && var_col < 0)) {
            if (threadIdx.x == 0) st->err = ERR_CUT_ARG;
            return;
        }
        if (var_row == -1) {  // non-basic variable: unit row (:46-53)
            for (int col = threadIdx.x; col < ld; col += blockDim.x) {
                double v = 0.0;
                if (col == 0) v = sign * value;
                else if (col == var_col) v = sign;
                cut[col] = v;
            }
        } else {  // basic variable: negated copy of its row (:54-62)
            const double* src = A + (long long)var_row * ld;
            for (int col = threadIdx.x; col < ld; col += blockDim.x) {
                double v = 0.0;
                if (col == 0) v = sign * (value - src[0]);
                else if (col < W) v = -sign * src[col];
                cut[col] = v;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int h = 0; h < n; h++) {  // getNewElementIndex + map updates (:64-69)
            const int slack = st->last_element_index++;
            if (slack >= s.idx_stride) { st->err = ERR_CAPACITY; break; }
            vibr[H + h] = slack;
            rbv[slack] = H + h;
            cbv[slack] = -1;
        }
        st->H = H + n;
    }
}

// Read-back for the host tree: RHS column + varIndexByRow of each slot, and the slot's state.
__global__ void __launch_bounds__(256) k_gather(Slots s, int first_slot, double* rhs, int32_t* rows, DevState* states,
                                                int out_stride, int first_out) {
    const int slot = first_slot + blockIdx.x, o = first_out + blockIdx.x;
    const DevState* st = s.st + slot;
    const double* A = s.A + (long long)slot * s.A_stride;
    const int32_t* vibr = s.vibr + (long long)slot * s.vibr_stride;
    const int H = st->H;
    for (int r = threadIdx.x; r < H; r += blockDim.x) {
        if (rhs) rhs[(long long)o * out_stride + r] = A[(long long)r * s.ld];
        if (rows) rows[(long long)o * out_stride + r] = vibr[r];
    }
    if (threadIdx.x == 0) states[o] = *st;
}
