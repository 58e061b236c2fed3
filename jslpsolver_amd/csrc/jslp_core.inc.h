// jslp_core.inc.h -- the per-tableau simplex core (selection, pivot preparation, row update, the one-workgroup
// solver), written once over `real_t` and compiled twice by jslp_kernels.hip.h:
//   * real_t = double, at global scope: THE engine (bit-exact with the reference, every parity test runs on it)
//   * real_t = float, in namespace f32: the fp32 twin behind jslp_engine_simplex_f32 -- the same pivoting rules on a
//     half-as-wide tableau, used only by the fp32-vs-fp64 tolerance sweep; it has no reference
//     to be exact against and never touches the fp64 state.
// Needs, at the point of inclusion: real_t, real2_t, rmul_rn / rsub_rn for real_t, DevState, the ST_* / ERR_* enums and
// the JSLP_* launch constants.  No include guard on purpose.

// Everything a step needs for ONE tableau.
struct Ctx {
    real_t* A;
    int32_t* vibr;
    int32_t* vibc;
    int32_t* rbv;
    int32_t* cbv;
    const uint8_t* unr;
    real_t* prow;
    real_t* pcol;
    uint8_t* dirty;       // [cap_rows] row differs from the snapshot (maintained by the per-node kernel only)
    real_t* rhs;          // [cap_rows] contiguous mirror of column 0 (valid iff st->rhs_valid; nullptr = read A with a stride)
    real_t* oo;           // optional objectives: n_opt rows of ld doubles (optionalObjectives[o].reducedCosts)
    int32_t n_opt;
    DevState* st;
    int2* hist;
    int2* trace;
    long long trace_cap;
    int32_t hist_cap;
    int32_t ld, W;
    int32_t check_cycles;
    int32_t batch;        // partial-pricing batch size (simplex.ts:118-124)
    int32_t use_partial;  // simplex.ts:127
    int32_t stop_at_phase2;  // hand phase 2 to the fused pipeline instead of continuing here
    int32_t has_unr;         // any unrestricted variable at all (else the per-column map lookups are skipped)
    real_t precision;
    cnt_t* cnt;              // work counters (nullptr = counting off)
};

// Base pointers + per-slot strides: slot s of a batch owns the s-th tableau copy.
struct Slots {
    real_t* A;       long long A_stride;
    int32_t* vibr;   int32_t vibr_stride;
    int32_t* vibc;   int32_t vibc_stride;
    int32_t* rbv;    int32_t idx_stride;
    int32_t* cbv;
    const uint8_t* unr;
    real_t* prow;    int32_t prow_stride;
    real_t* pcol;    int32_t pcol_stride;
    uint8_t* dirty;  // stride = pcol_stride
    real_t* rhs;     // stride = pcol_stride; see Ctx::rhs
    real_t* oo;      long long oo_stride;  // n_opt * ld per slot
    int32_t n_opt;
    DevState* st;
    int2* hist;      int32_t hist_cap;
    int2* trace;     long long trace_cap;   // only slot 0 traces
    int32_t ld, W;
    int32_t batch, use_partial;
    int32_t has_unr;
    real_t precision;
    cnt_t* cnt;              // see Ctx::cnt
    const int32_t* watch;    // watched variable indexes (compact read-back), n_watch entries
    int32_t n_watch;
    const int32_t* watch_pos;  // variable index -> its position in `watch` (-1 elsewhere); nullptr when a variable is listed twice
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
    c.dirty = s.dirty + (long long)slot * s.pcol_stride;
    c.rhs = s.rhs ? s.rhs + (long long)slot * s.pcol_stride : nullptr;
    c.oo = s.oo ? s.oo + (long long)slot * s.oo_stride : nullptr;
    c.n_opt = s.n_opt;
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
    c.stop_at_phase2 = 0;
    c.has_unr = s.has_unr;
    c.precision = s.precision;
    c.cnt = s.cnt;
    return c;
}

// the reference's zero test `!(v >= -1e-16 && v <= 1e-16)` (simplex.ts:356,372,379): NaN counts as non-zero
__device__ __forceinline__ bool nonzero16(real_t v) { return !(v >= -1e-16 && v <= 1e-16); }

__device__ __forceinline__ real_t eliminate(real_t a, real_t k, real_t p) {
    // matrix[r,c] - coefficient * v0 with BOTH roundings (JavaScript never fuses)
    return rsub_rn(a, rmul_rn(k, p));
}

// ---------------------------------------------------------------------------------------------------
// (value, index) candidates and their reductions.  `i == 0` means "no candidate" (row/column 0 is never
// selectable).  All orders are total on (key..., index) so the result does not depend on thread mapping.
// ---------------------------------------------------------------------------------------------------
struct Cand {
    real_t v;
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
        const bool take = better(y, x);  // field-wise selects on one predicate (see price_row)
        x.v = take ? y.v : x.v;
        x.i = take ? y.i : x.i;
        x.b = take ? y.b : x.b;
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
    real_t* A = c.A;
    const real_t quot = A[(long long)pr * ld + pc];  // simplex.ts:335
    int any = 0, n_gated = 0;
    for (int r = tid; r < H; r += nt) {
        real_t k;
        if (pcol_ready && r > 0) {
            k = c.pcol[r];
        } else {
            k = A[(long long)r * ld + pc];
            c.pcol[r] = k;
        }
        const int gated = (r != pr && nonzero16(k));
        any |= gated;
        n_gated += gated;
    }
    // any row that will execute the inner loop of simplex.ts:367-391 lazily zeroes the tiny pivot-row entries
    const int anyrow = __syncthreads_or(any);  // also orders the quot read above against the row write below
    real_t* prow_A = A + (long long)pr * ld;
    int n_cols = 0;
    for (int col = tid; col < ld; col += nt) {
        real_t v = 0.0;
        if (col < W) {
            const real_t val = prow_A[col];
            const bool innz = nonzero16(val);       // :356
            v = innz ? val / quot : 0.0;            // :357 / :361  (IEEE division)
            bool in_list = innz;
            if (col == pc) v = 1.0 / quot;          // :364 (membership of pc in nonZeroColumns is decided by `val`)
            if (in_list && anyrow && !nonzero16(v) && v != 0.0) v = 0.0;  // :381-383
            prow_A[col] = v;
            if (col == 0 && c.rhs) c.rhs[pr] = v;
            n_cols += (nonzero16(v) || col == pc) ? 1 : 0;
        }
        c.prow[col] = v;
    }
    if (c.cnt) {  // work counters (uniform branch): cells simplex.ts:376-387 touches = gated rows x live pivot-row columns
        __syncthreads();  // sm.flag / sm.flag2 are free here
        if (tid == 0) { sm.flag = 0; sm.flag2 = 0; }
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) { n_gated += __shfl_down(n_gated, off, 64); n_cols += __shfl_down(n_cols, off, 64); }
        if ((tid & 63) == 0) { atomicAdd(&sm.flag, n_gated); atomicAdd(&sm.flag2, n_cols); }
        __syncthreads();
        if (tid == 0) {
            atomicAdd(c.cnt + CNT_CELLS, (cnt_t)sm.flag * (cnt_t)sm.flag2);
            atomicAdd(c.cnt + CNT_ROWS, (cnt_t)sm.flag);
        }
    }
    // optional objectives (simplex.ts:394-412): same elimination with exact `!== 0` tests, on the final pivot row
    for (int o = 0; o < c.n_opt; o++) {
        real_t* rc = c.oo + (long long)o * ld;
        __syncthreads();                       // prow[] complete / previous objective done
        const real_t coefficient = rc[pc];     // every thread reads it before anyone overwrites rc[pc]
        __syncthreads();
        if (coefficient != 0.0) {
            for (int col = tid; col < W; col += nt) {
                if (col == pc) { rc[col] = -coefficient / quot; continue; }
                const real_t v0 = c.prow[col];
                if (v0 != 0.0) rc[col] = eliminate(rc[col], coefficient, v0);
            }
        }
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
    const real_t precision = c.precision;
    const real_t* A = c.A;
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
            const real_t v = c.rhs ? c.rhs[r] : A[(long long)r * ld];
            if (v < best.v) { best.v = v; best.i = r; }
        }
        best = block_reduce(best, MinFirst(), sm);
        if (best.i == 0) {  // :51-54 -> feasible; phase 2 starts in this same step
            phase = 2;
            if (tid == 0) { st->feasible = 1; st->phase = 2; st->entered_phase2 = 1; st->hist_n = 0; }
            __syncthreads();
            if (c.stop_at_phase2) {
                if (tid == 0) { st->status = ST_PHASE1_DONE; st->do_pivot = 0; }
                __syncthreads();
                return;
            }
        } else {
            pr = best.i;
            // entering column: max -cost/coef over unrestricted or coef < -precision (simplex.ts:56-71)
            const real_t* row = A + (long long)pr * ld;
            Cand q; q.v = -INFINITY; q.i = 0; q.b = 0;
            for (int col = 1 + tid; col < W; col += nt) {
                const real_t coef = row[col];
                const bool un = c.has_unr && c.unr[c.vibc[col]] != 0;
                if (un || coef < -precision) {
                    const real_t quo = -A[col] / coef;
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
        int st_opt_row = -1;  // which optional objective supplied the entering column (-1: the main cost row)
        for (int col = 1 + tid; col < W; col += nt) {
            const real_t rc = A[col];
            const bool un = c.has_unr && c.unr[c.vibc[col]] != 0;
            const int b = c.use_partial ? (col - 1) / c.batch : 0;
            real_t val; int ng;
            if (un && rc < 0) { val = -rc; ng = 1; } else { val = rc; ng = 0; }
            // per-thread running best in the same total order as the reduction; a candidate must beat
            // `precision` (strict >), which every thread applies itself
            if (val > precision) {
                Cand cand; cand.v = val; cand.i = col; cand.b = b;
                const bool take = PriceFirst()(cand, e);
                e.v = take ? cand.v : e.v;
                e.i = take ? cand.i : e.i;
                e.b = take ? cand.b : e.b;
                neg_flag = take ? ng : neg_flag;
            }
        }
        e = block_reduce(e, PriceFirst(), sm);
        // simplex.ts:221-263: no column prices out on the main row -> break the tie on the optional objectives, in
        // priority order, among the columns whose reduced cost is within +-precision on every earlier row
        for (int o = 0; e.i == 0 && o < c.n_opt; o++) {
            Cand x; x.v = precision; x.i = 0; x.b = 0;
            for (int col = 1 + tid; col < W; col += nt) {
                const real_t rc0 = A[col];
                bool deferred = -precision < rc0 && rc0 < precision;
                for (int q = 0; deferred && q < o; q++) {
                    const real_t rq = c.oo[(long long)q * ld + col];
                    deferred = -precision < rq && rq < precision;
                }
                if (!deferred) continue;
                const real_t rc = c.oo[(long long)o * ld + col];
                if (-precision < rc && rc < precision) continue;
                const bool un = c.has_unr && c.unr[c.vibc[col]] != 0;
                const real_t val = (un && rc < 0) ? -rc : rc;
                const bool take = val > x.v;  // strict: first index wins ties inside a thread (ascending columns)
                x.v = take ? val : x.v;
                x.i = take ? col : x.i;
            }
            e = block_reduce(x, PriceFirst(), sm);
            if (e.i != 0) st_opt_row = o;
        }
        if (e.i == 0) {  // optimal (simplex.ts:265-269); setEvaluation happens on the host from obj_cell
            if (tid == 0) { st->optimal = 1; finish(c); }
            __syncthreads();
            return;
        }
        pc = e.i;
        // isReducedCostNegative of the winner: recompute (cheap, uniform)
        {
            const real_t rc = st_opt_row < 0 ? A[pc] : c.oo[(long long)st_opt_row * ld + pc];
            const bool un = c.has_unr && c.unr[c.vibc[pc]] != 0;
            neg_flag = (un && rc < 0) ? 1 : 0;
        }
        // ratio test (simplex.ts:271-296) in its order-free form: r_deg = first row passing the
        // degenerate test wins outright, otherwise first-index argmin of the accepted quotients.
        Cand m; m.v = INFINITY; m.i = 0; m.b = 0;
        int rdeg = 0x7fffffff;
        for (int r = tid; r < H; r += nt) {
            const real_t colv = A[(long long)r * ld + pc];
            c.pcol[r] = colv;  // the update step needs the whole column anyway (row 0 included)
            if (r == 0) continue;
            const real_t rhs = c.rhs ? c.rhs[r] : A[(long long)r * ld];
            if (-precision < colv && colv < precision) continue;
            if (colv > 0 && precision > rhs && rhs > -precision) {
                if (r < rdeg) rdeg = r;
                continue;
            }
            const real_t quo = neg_flag ? -rhs / colv : rhs / colv;
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

// Whole chip, one launch per pivot: workgroup (bx, by) owns rows [by*8, by*8+8) x columns [bx*512, +512).
// Every lane keeps its two pivot-row values in registers, the 8 loads of a lane are issued back to back
// (16 B each, 1 KiB per wave-instruction, fully coalesced) before the first dependent use.
__global__ void __launch_bounds__(JSLP_UPD_THREADS) k_update(Ctx c) {
    const DevState* st = c.st;
    if (st->status != ST_RUNNING || !st->do_pivot) return;
    const int H = st->H, ld = c.ld;
    const int pr = st->pr, pc = st->pc;
    const real_t quot = st->quot;
    const int c0 = blockIdx.x * JSLP_UPD_COLS + threadIdx.x * 2;
    if (c0 >= ld) return;
    const real2_t p = *reinterpret_cast<const real2_t*>(c.prow + c0);
    const bool v0 = nonzero16(p.x), v1 = nonzero16(p.y);
    const bool has_pc = (pc == c0) || (pc == c0 + 1);
    if (!v0 && !v1 && !has_pc) return;  // nothing in these two columns changes (sparse pivot rows)
    const int r0 = blockIdx.y * JSLP_UPD_ROWS;
    real_t k[JSLP_UPD_ROWS];
    bool act[JSLP_UPD_ROWS];
    real2_t a[JSLP_UPD_ROWS];
#pragma unroll
    for (int i = 0; i < JSLP_UPD_ROWS; i++) {
        const int r = r0 + i;
        k[i] = (r < H) ? c.pcol[r] : 0.0;
        act[i] = (r < H) && (r != pr) && nonzero16(k[i]);  // :370-375 row gate
    }
#pragma unroll
    for (int i = 0; i < JSLP_UPD_ROWS; i++)
        if (act[i]) a[i] = *reinterpret_cast<const real2_t*>(c.A + (long long)(r0 + i) * ld + c0);
#pragma unroll
    for (int i = 0; i < JSLP_UPD_ROWS; i++) {
        if (!act[i]) continue;
        real2_t x = a[i];
        if (v0) x.x = eliminate(x.x, k[i], p.x);
        if (v1) x.y = eliminate(x.y, k[i], p.y);
        if (has_pc) {  // :387 overwrites whatever the loop did to column pc
            const real_t nv = -k[i] / quot;
            if (pc == c0) x.x = nv; else x.y = nv;
        }
        *reinterpret_cast<real2_t*>(c.A + (long long)(r0 + i) * ld + c0) = x;
    }
}

// The same elimination done by ONE workgroup (per-node kernel).  The pivot column is read once by all threads in
// parallel and the rows that pass the reference's gate (simplex.ts:370-375) are compacted into an LDS list, so
// the waves only ever touch rows the reference touches (a Monster_II pivot: ~10 of 945) and never chain dependent
// global loads; lanes take column pairs.  Touched rows are flagged dirty for the next restore().
template <int CAP>
struct ActSmem {
    int32_t n;
    int32_t row[CAP];
    real_t k[CAP];
};

__device__ __forceinline__ void update_row_wave(const Ctx& c, int r, real_t k, int pc, real_t quot, int lane) {
    const int ld = c.ld;
    real_t* row = c.A + (long long)r * ld;
    for (int c0 = lane * 2; c0 < ld; c0 += 128) {
        const real2_t p = *reinterpret_cast<const real2_t*>(c.prow + c0);
        const bool v0 = nonzero16(p.x), v1 = nonzero16(p.y);
        const bool has_pc = (pc == c0) || (pc == c0 + 1);
        if (!v0 && !v1 && !has_pc) continue;
        real2_t x = *reinterpret_cast<const real2_t*>(row + c0);
        if (v0) x.x = eliminate(x.x, k, p.x);
        if (v1) x.y = eliminate(x.y, k, p.y);
        if (has_pc) {
            const real_t nv = -k / quot;
            if (pc == c0) x.x = nv; else x.y = nv;
        }
        *reinterpret_cast<real2_t*>(row + c0) = x;
        if (c0 == 0 && c.rhs) c.rhs[r] = x.x;
    }
}

template <int CAP>
__device__ __forceinline__ void update_rows_wg(const Ctx& c, ActSmem<CAP>& act) {
    const DevState* st = c.st;
    const int H = st->H, pr = st->pr, pc = st->pc;
    const real_t quot = st->quot;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    if (tid == 0) { act.n = 0; c.dirty[pr] = 1; }
    __syncthreads();
    for (int r = tid; r < H; r += nt) {
        const real_t k = c.pcol[r];
        if (r != pr && nonzero16(k)) {
            const int idx = atomicAdd(&act.n, 1);
            if (idx < CAP) { act.row[idx] = r; act.k[idx] = k; }
            c.dirty[r] = 1;
        }
    }
    __syncthreads();
    const int n = act.n;
    if (n <= CAP) {
        for (int i = w; i < n; i += nw) update_row_wave(c, act.row[i], act.k[i], pc, quot, lane);
    } else {  // more gated-in rows than the list holds: walk all rows (wave-uniform gate)
        for (int r = w; r < H; r += nw) {
            const real_t k = c.pcol[r];
            if (r == pr || !nonzero16(k)) continue;
            update_row_wave(c, r, k, pc, quot, lane);
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
    if (threadIdx.x == 0) { c.st->status = ST_RUNNING; c.st->gen = 0; c.st->rhs_valid = 0; }
    __syncthreads();
    prepare_pivot(c, pr, pc, false, sm);
}
__global__ void k_end_pivot(Ctx c) {
    c.st->status = ST_DONE;
    c.st->do_pivot = 0;
}

// simplex() entry: `this.bounded = true; phase1(); if (feasible) phase2()` (simplex.ts:14-23)
// (zero: the batch kernels of jslp_wglds.hip.h pass an opaque 0 -- the compiler otherwise builds ONE 16-byte vector of zeros for these
//  stores in the kernel's prologue and keeps it in scratch for the whole batch)
__device__ __forceinline__ void begin_simplex(DevState* st, int iters_cap, int zero = 0) {
    st->status = ST_RUNNING;
    st->phase = 1;
    st->bounded = 1;
    st->optimal = zero;
    st->unbounded_var = -1;
    st->it1 = zero;
    st->it2 = zero;
    st->entered_phase2 = zero;
    st->cycle_phase = zero;
    st->hist_n = zero;
    st->do_pivot = zero;
    st->err = st->err == ERR_CUT_ARG || st->err == ERR_CAPACITY ? st->err : ERR_NONE;
    st->iters_left = iters_cap;
}
__global__ void k_begin(Slots s, int first_slot, int iters_cap) {
    begin_simplex(s.st + first_slot + blockIdx.x, iters_cap);
    s.st[first_slot + blockIdx.x].gen = 0;  // the chip-wide kernels do not maintain dirty-row flags ...
    s.st[first_slot + blockIdx.x].rhs_valid = 0;  // ... nor the RHS mirror
}

// One workgroup = one whole simplex() on one tableau (slot first_slot + blockIdx.x).
// THREADS = workgroup size the launch uses, CAP = capacity of the LDS list of gated-in rows (more rows than that: every
// row is walked).  <1024, 4096> is the latency shape for ONE tableau; batches of nodes trade per-node latency for nodes
// in flight per CU with smaller workgroups (chosen by the host, jslp_hip.hip).
template <int CAP>
__device__ __forceinline__ void simplex_wg(const Ctx& c, Smem& sm, ActSmem<CAP>& act, int iters_cap) {
    if (threadIdx.x == 0) begin_simplex(c.st, iters_cap);
    __syncthreads();
    if (c.rhs && !c.st->rhs_valid) {  // first use after a chip-wide solve / an upload: one strided gather, then contiguous
        const int H = c.st->H;
        for (int r = threadIdx.x; r < H; r += blockDim.x) c.rhs[r] = c.A[(long long)r * c.ld];
        __syncthreads();
        if (threadIdx.x == 0) c.st->rhs_valid = 1;
        __syncthreads();
    }
    if (c.st->err != ERR_NONE) {  // a bad cut list: report, do not solve
        if (threadIdx.x == 0) finish(c);
        return;
    }
    for (;;) {
        select_step(c, sm);
        if (!c.st->do_pivot) break;
        update_rows_wg(c, act);
        __syncthreads();
    }
}
template <int THREADS, int CAP>
__global__ void __launch_bounds__(THREADS) k_simplex_wg(Slots s, int first_slot, int check_cycles, int iters_cap) {
    __shared__ Smem sm;
    __shared__ ActSmem<CAP> act;
    const Ctx c = slot_ctx(s, first_slot + blockIdx.x, check_cycles);
    simplex_wg(c, sm, act, iters_cap);
}
