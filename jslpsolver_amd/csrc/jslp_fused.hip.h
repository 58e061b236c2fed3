// jslp_fused.hip.h -- the fused phase-2 pipeline (one launch per pivot) for tableaus that do not fit the registers.
// Included by jslp_kernels.hip.h (needs the simplex core, DevState and the launch constants defined there).
#pragma once

// ===================================================================================================
// Fused phase-2 pipeline for one LARGE tableau: ONE launch per pivot.
//
// Launch t (a) finishes the selection of pivot t from what launch t-1 left behind -- the entering column
// f_pc, the per-workgroup ratio-test candidates cands[] and the pivot column pcol[] -- (b) streams the whole
// tableau once, out of place (read buf[in], write buf[in^1]: no workgroup ever reads a cell another one is
// writing, so any workgroup may read the pivot row and the cost row straight from the input), and (c) while
// the updated rows are still in registers prepares pivot t+1: every workgroup re-derives the updated cost row
// and prices it (same result everywhere), then its rows' ratio-test candidates and pivot-column entries.
// Workgroup w owns `rpb` consecutive rows; lane pairs own two adjacent columns (16-byte accesses).
// Preconditions checked by the host: phase 2, ld <= 4096 (two column tiles per lane), H <= 64 * 256,
// precision >= 1e-15 (then the entering cost is never "tiny" and simplex.ts:381-383 always fires).
// The cycle check is done by workgroup 0 alone; the other workgroups pivot speculatively into the OTHER
// buffer, which is simply not adopted when the check (or unboundedness) stops the solve.
// ===================================================================================================
__device__ __forceinline__ void copy_state(DevState* dst, const DevState* src) {
    static_assert(sizeof(DevState) % 8 == 0, "DevState is copied as 8-byte words");
    const unsigned long long* a = reinterpret_cast<const unsigned long long*>(src);
    unsigned long long* b = reinterpret_cast<unsigned long long*>(dst);
#pragma unroll
    for (unsigned i = 0; i < sizeof(DevState) / 8; i++) b[i] = a[i];
}

struct FCand {       // per-workgroup ratio-test summary
    double q;        // smallest accepted quotient among its rows (first index on ties)
    double kq;       // pivot-column entry of that row (becomes `quot` if it wins)
    double kdeg;     // pivot-column entry of row rdeg
    int32_t r;       // row of q, 0 = none
    int32_t rdeg;    // first row passing the degenerate test (simplex.ts:285-289), 0x7fffffff = none
};
#define JSLP_F_THREADS 1024
#define JSLP_F_RG 8          // rows processed per group (kept in registers)
#define JSLP_F_COLLECT 64    // rows whose candidate-column entries are collected in LDS between two barriers (a power of two, <= 64: one lane of wave 0 per row)
#define JSLP_F_MAXG 256      // workgroups (= CUs)

struct FusedCtx {
    Ctx c;               // slot 0 (maps, history, trace, canonical state)
    double* buf[2];      // buf[0] = c.A
    FCand* cands[2];
    double* pcol[2];
    DevState* fst[2];
    int32_t G, rpb;
    int32_t H;   // height is fixed during a simplex() call: known to the host, so no load gates the prefetch
    int32_t nt;  // non-temporal hints on the streamed tableau cells
    // UNR builds: "the variable of column c / of row r is unrestricted", double-buffered like the candidates (every workgroup
    // reads [launch & 1], workgroup 0 writes [(launch + 1) & 1]): the maps themselves are swapped by workgroup 0 at the END of a
    // launch, so another workgroup reading them mid-launch could see either side of the swap
    uint8_t* ucol[2];
    uint8_t* urow[2];
    // optional objectives (c.n_opt rows of ld doubles): ping-pong with the tableau buffers (oo[i] belongs to buf[i]); workgroup 0
    // writes the updated rows, every workgroup reads the input side
    double* oo[2];
};

__device__ __forceinline__ void fcand_consider(FCand& best, int r, double colv, double rhs, double precision, int neg = 0) {
    // one row of the ratio test (simplex.ts:276-296), rows visited in ascending order
    if (-precision < colv && colv < precision) return;
    if (colv > 0 && precision > rhs && rhs > -precision) {
        if (r < best.rdeg) { best.rdeg = r; best.kdeg = colv; }
        return;
    }
    const double quo = neg ? -rhs / colv : rhs / colv;  // neg = isReducedCostNegative (unrestricted entering variable, simplex.ts:282)
    if (quo > precision && best.q > quo) { best.q = quo; best.r = r; best.kq = colv; }
}

__device__ __forceinline__ bool fcand_better(const FCand& a, const FCand& b) {  // MinFirst on (q, r)
    if (a.r == 0) return false;
    if (b.r == 0) return true;
    return a.q < b.q || (a.q == b.q && a.r < b.r);
}

// reduce FCands held by the lanes of ONE wave (lanes >= n hold "none")
__device__ __forceinline__ FCand fcand_wave_reduce(FCand x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        FCand y;
        y.q = __shfl_down(x.q, off, 64);
        y.kq = __shfl_down(x.kq, off, 64);
        y.kdeg = __shfl_down(x.kdeg, off, 64);
        y.r = __shfl_down(x.r, off, 64);
        y.rdeg = __shfl_down(x.rdeg, off, 64);
        if (fcand_better(y, x)) { x.q = y.q; x.r = y.r; x.kq = y.kq; }
        if (y.rdeg < x.rdeg) { x.rdeg = y.rdeg; x.kdeg = y.kdeg; }
    }
    return x;
}

struct FSmem {
    Smem red;
    FCand wave[JSLP_F_THREADS / 64];
    FCand win;
    double col[JSLP_F_COLLECT];  // the entering column's new entries / the new RHS of up to JSLP_F_COLLECT rows between two barriers (k_pivot_fused)
    double rhs[JSLP_F_COLLECT];
    int neg;  // pricing with unrestricted variables: isReducedCostNegative of the winner
};

__device__ __forceinline__ FCand fcand_none() {
    FCand x; x.q = INFINITY; x.kq = 0; x.kdeg = 0; x.r = 0; x.rdeg = 0x7fffffff;
    return x;
}

__device__ __forceinline__ FCand fcand_block_reduce(FCand x, FSmem& sm) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    x = fcand_wave_reduce(x);
    __syncthreads();
    if (lane == 0) sm.wave[w] = x;
    __syncthreads();
    if (w == 0) {
        FCand y = lane < nw ? sm.wave[lane] : fcand_none();
        y = fcand_wave_reduce(y);
        if (lane == 0) sm.win = y;
    }
    __syncthreads();
    return sm.win;
}

// Pricing (simplex.ts:118-219) of the cost-row pairs a lane holds -- NT tiles of 2 x JSLP_F_THREADS columns, the lane's pair of
// tile t at columns c0 + t * 2 * JSLP_F_THREADS -- reduced over the workgroup.  UNR: bit (2 t + j) of `um` says the variable of
// that column is unrestricted: it prices with |reduced cost| and hands isReducedCostNegative (*neg) to the ratio test
// (simplex.ts:164-177, 282).  Returns the entering column (0 = optimal).
#define JSLP_F_TW (2 * JSLP_F_THREADS)  // columns per tile

// One cell of an optional objective row after the pivot (pr, pc) (simplex.ts:394-412: the same elimination as for the cost row,
// but with exact `!== 0` tests, on the FINAL normalised pivot row entry pval)
__device__ __forceinline__ double oo_cell_after(double rc, double coefficient, int col, int pc, double quot, double pval) {
    if (coefficient != 0.0) {
        if (col == pc) return -coefficient / quot;
        if (pval != 0.0) return eliminate(rc, coefficient, pval);
    }
    return rc;
}
// workgroup 0: every optional objective row, input side -> output side
template <int NT>
__device__ __forceinline__ void fused_update_oo(const Ctx& c, const double* oo_in, double* oo_out, int c0, int pc, double quot,
                                                const double2 (&p)[NT]) {
    const int ld = c.ld, W = c.W;
    for (int o = 0; o < c.n_opt; o++) {
        const double* rin = oo_in + (long long)o * ld;
        double* rout = oo_out + (long long)o * ld;
        const double coefficient = rin[pc];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int ct = c0 + t * JSLP_F_TW;
            if (ct >= ld) continue;
            double2 rc = *reinterpret_cast<const double2*>(rin + ct);
            if (ct < W) rc.x = oo_cell_after(rc.x, coefficient, ct, pc, quot, p[t].x);
            if (ct + 1 < W) rc.y = oo_cell_after(rc.y, coefficient, ct + 1, pc, quot, p[t].y);
            *reinterpret_cast<double2*>(rout + ct) = rc;
        }
    }
}
// simplex.ts:221-263: no column prices out on the main row -> the optional objectives break the tie, in priority order, among
// the columns whose reduced cost is within +-precision on the main row and on every earlier objective.  x[] = the main cost row
// (already updated); the objective rows are read from oo_in and, when `after` (a pivot (pr, pc) is being applied in this very
// launch), brought up to date on the fly.  Returns the entering column (0 = none) and which row named it / its sign.
template <int NT, bool UNR>
__device__ __forceinline__ int price_optional(const double2 (&x)[NT], int c0, unsigned um, const Ctx& c, FSmem& sm, const double* oo_in,
                                              bool after, int pc, double quot, const double2 (&p)[NT], int* neg) {
    const double precision = c.precision;
    const int ld = c.ld;
    for (int o = 0; o < c.n_opt; o++) {
        Cand best; best.v = precision; best.i = 0; best.b = 0;
        int bneg = 0;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int col = c0 + t * JSLP_F_TW + j;
                if (col < 1 || col >= c.W) continue;
                const double rc0 = j ? x[t].y : x[t].x;
                const double pval = j ? p[t].y : p[t].x;
                bool deferred = -precision < rc0 && rc0 < precision;
                for (int q = 0; deferred && q <= o; q++) {
                    const double* rq = oo_in + (long long)q * ld;
                    double v = rq[col];
                    if (after) v = oo_cell_after(v, rq[pc], col, pc, quot, pval);
                    if (q < o) { deferred = -precision < v && v < precision; continue; }
                    if (-precision < v && v < precision) break;  // (q == o) this objective does not price the column out either
                    const bool un = UNR && ((um >> (2 * t + j)) & 1u);
                    const double val = (un && v < 0) ? -v : v;
                    const bool take = val > best.v;  // strict: my columns ascend, ties keep the earlier one
                    best.v = take ? val : best.v;
                    best.i = take ? col : best.i;
                    bneg = take ? ((un && v < 0) ? 1 : 0) : bneg;
                }
            }
        const Cand e = block_reduce(best, PriceFirst(), sm.red);
        if (e.i != 0) {
            if (UNR) {
                if (threadIdx.x == 0) sm.neg = 0;
                __syncthreads();
                if (best.i == e.i) sm.neg = bneg;  // (one lane holds that column)
                __syncthreads();
                *neg = sm.neg;
            }
            return e.i;
        }
    }
    return 0;
}
template <int NT, bool UNR>
__device__ __forceinline__ int price_row(const double2 (&x)[NT], int c0, unsigned um, const Ctx& c, FSmem& sm, int* neg) {
    // (candidates written out field by field: a struct select inside an unrolled loop was observed to keep the FIRST
    //  column's index with the SECOND column's value on gfx950 / ROCm 7.2)
    double bv = c.precision;
    int bi = 0, bb = 0;
#pragma unroll
    for (int t = 0; t < NT; t++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {  // my columns ascend: earlier batch first, bigger value inside a batch, first index on ties
            const int col = c0 + t * JSLP_F_TW + j;
            const double rc = j ? x[t].y : x[t].x;
            const double val = (UNR && ((um >> (2 * t + j)) & 1u) && rc < 0) ? -rc : rc;
            const bool ok = col >= 1 && col < c.W && val > c.precision;
            const int b = c.use_partial ? (col - 1) / c.batch : 0;
            const bool take = ok && (bi == 0 || b < bb || (b == bb && val > bv));
            bv = take ? val : bv;
            bi = take ? col : bi;
            bb = take ? b : bb;
        }
    }
    Cand e;
    e.v = bv; e.i = bi; e.b = bb;
    e = block_reduce(e, PriceFirst(), sm.red);
    const int cn = e.i;
    if (UNR) {  // the lane holding the winner knows the sign of its reduced cost
        if (threadIdx.x == 0) sm.neg = 0;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (cn != 0 && cn == c0 + t * JSLP_F_TW + j) sm.neg = (((um >> (2 * t + j)) & 1u) && (j ? x[t].y : x[t].x) < 0) ? 1 : 0;
        __syncthreads();
        *neg = sm.neg;
    }
    return cn;
}

// streamed tableau cells: every cell is read once and written once per launch, so optionally bypass the
// cache retention policy (non-temporal) to keep L2 for the small shared vectors
__device__ __forceinline__ double2 ld_stream(const double* p, int nt) {
    const double2* q = reinterpret_cast<const double2*>(p);
    if (nt) {
        double2 v;
        v.x = __builtin_nontemporal_load(&q->x);
        v.y = __builtin_nontemporal_load(&q->y);
        return v;
    }
    return *q;
}
__device__ __forceinline__ void st_stream(double* p, double2 v, int nt) {
    double2* q = reinterpret_cast<double2*>(p);
    if (nt) {
        __builtin_nontemporal_store(v.x, &q->x);
        __builtin_nontemporal_store(v.y, &q->y);
    } else {
        *q = v;
    }
}

// NT: column tiles per lane (ld <= NT * 2048); UNR: unrestricted variables present (per-column flags read from the maps at the
// top of every launch; the entering column's isReducedCostNegative travels with f_pc, bit 30)
// OPT: optional objectives present (workgroup 0 keeps their rows up to date; they break pricing ties; an entering column they
// name may have a ~0 main cost, and then the tiny-entry rule of simplex.ts:381-383 is not decidable here: ST_P1_SLOW, see k_fused_p1)
template <int NT, bool UNR, bool OPT>
__global__ void __launch_bounds__(JSLP_F_THREADS) k_pivot_fused(FusedCtx f, int launch) {
    __shared__ FSmem sm;
    const Ctx& c = f.c;
    const int tid = threadIdx.x, b = blockIdx.x;
    const bool init = launch == 0;
    const DevState* sin = init ? c.st : f.fst[launch & 1];
    DevState* sout = f.fst[(launch + 1) & 1];
    const int in_buf = init ? 0 : ((launch - 1) & 1);
    const double* Min = f.buf[in_buf];
    double* Mout = f.buf[in_buf ^ 1];
    const FCand* cin = f.cands[launch & 1];
    FCand* cout = f.cands[(launch + 1) & 1];
    const double* pin = f.pcol[launch & 1];
    double* pout = f.pcol[(launch + 1) & 1];
    const int ld = c.ld, W = c.W;
    const double precision = c.precision;

    const int H = f.H;
    const int c0 = tid * 2;  // my pair of tile t: columns c0 + t * JSLP_F_TW, +1
    const int r_begin = b * f.rpb, r_end = min(H, r_begin + f.rpb);
    // Everything whose address does not depend on the selection is requested FIRST, before any barrier: the
    // workgroup's first row group (8 rows x 16 B per lane), its pivot-column entries, the cost row, the
    // candidates and the state.  The selection below then runs while these loads are in flight.
    double2 a[JSLP_F_RG];
    double k[JSLP_F_RG];
    double2 row0[NT];
    double k0 = 0.0;
    FCand mine = fcand_none();
    // (vmcnt retires in order: the selection's own inputs go first so that waiting for them leaves the bulk
    // row loads in flight)
    const int status = sin->status;
    const int pc_raw = sin->f_pc;
    const int pc_in = pc_raw & 0x3fffffff;
    const int neg_in = UNR ? ((pc_raw >> 30) & 1) : 0;
    const int iters_left_in = sin->iters_left;
    unsigned um = 0;  // UNR: which of my columns carry an unrestricted variable
    if (UNR) {
        const uint8_t* ucin = f.ucol[launch & 1];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int col = c0 + t * JSLP_F_TW + j;
                // (launch 0 changes no map: it reads them directly and workgroup 0 builds the first flag arrays from them)
                if (col >= 1 && col < W && (init ? c.unr[c.vibc[col]] : ucin[col]) != 0) um |= 1u << (2 * t + j);
            }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) row0[t] = make_double2(0, 0);
    if (!init) {
        if (tid < f.G) mine = cin[tid];
        k0 = pin[0];
#pragma unroll
        for (int t = 0; t < NT; t++)
            if (c0 + t * JSLP_F_TW < ld) row0[t] = *reinterpret_cast<const double2*>(Min + c0 + t * JSLP_F_TW);
#pragma unroll
        for (int i = 0; i < JSLP_F_RG; i++) {
            const int r = r_begin + i;
            k[i] = r < r_end ? pin[r] : 0.0;
            a[i] = make_double2(0, 0);
            if (r < r_end && c0 < ld) a[i] = ld_stream(Min + (long long)r * ld + c0, f.nt);
        }
    }
    const bool live = init ? (status == ST_PHASE1_DONE) : (status == ST_RUNNING);
    if (!live) {  // the solve already ended (or never reached phase 2): carry the state forward
        if (b == 0 && tid == 0) { copy_state(sout, sin); if (init) sout->f_final_buf = 0; }
        return;
    }
    if (init) {
        // first phase-2 step: price the current cost row, then scan the entering column for candidates
        double2 r0[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            r0[t] = make_double2(0, 0);
            if (c0 + t * JSLP_F_TW < ld) r0[t] = *reinterpret_cast<const double2*>(Min + c0 + t * JSLP_F_TW);
        }
        int neg = 0;
        int cn = price_row<NT, UNR>(r0, c0, um, c, sm, &neg);
        if (OPT && cn == 0) cn = price_optional<NT, UNR>(r0, c0, um, c, sm, f.oo[0], false, 0, 1.0, r0, &neg);
        if (cn == 0) {  // already optimal (simplex.ts:265-269)
            if (b == 0 && tid == 0) {
                copy_state(sout, sin); DevState& s = *sout;
                s.status = ST_DONE; s.optimal = 1; s.do_pivot = 0; s.obj_cell = Min[0]; s.f_final_buf = 0;
            }
            return;
        }
        FCand best = fcand_none();
        if (tid < 64) {
            for (int r = r_begin + tid; r < r_end; r += 64) {
                const double colv = Min[(long long)r * ld + cn];
                pout[r] = colv;
                if (r >= 1) fcand_consider(best, r, colv, Min[(long long)r * ld], precision, neg);
            }
            // lanes hold rows in interleaved order: the reduction's (q, r) / min-rdeg orders are total
            best = fcand_wave_reduce(best);
            if (tid == 0) cout[b] = best;
        }
        if (UNR && b == 0) {
            uint8_t* uco = f.ucol[(launch + 1) & 1];
            uint8_t* uro = f.urow[(launch + 1) & 1];
            for (int col = tid; col < W; col += JSLP_F_THREADS) uco[col] = col >= 1 ? c.unr[c.vibc[col]] : 0;
            for (int r = tid; r < H; r += JSLP_F_THREADS) uro[r] = r >= 1 ? c.unr[c.vibr[r]] : 0;
        }
        if (b == 0 && tid == 0) {
            copy_state(sout, sin); DevState& s = *sout;
            s.status = ST_RUNNING; s.f_pc = cn | (neg << 30); s.f_final_buf = 0; s.do_pivot = 0;
        }
        return;
    }

    // ---- STEP: pivot t ---------------------------------------------------------------------------
    const int pc = pc_in;
    (void)neg_in;  // (the candidates of this pivot were built with it by the previous launch)
    if (iters_left_in <= 0) {
        if (b == 0 && tid == 0) {
            copy_state(sout, sin); DevState& s = *sout;
            s.err = ERR_ITER_LIMIT; s.status = ST_DONE; s.do_pivot = 0; s.obj_cell = Min[0]; s.f_final_buf = in_buf;
        }
        return;
    }
    const FCand win = fcand_block_reduce(mine, sm);
    int pr; double quot;
    if (win.rdeg != 0x7fffffff) { pr = win.rdeg; quot = win.kdeg; }
    else if (win.r != 0) { pr = win.r; quot = win.kq; }
    else {  // unbounded (simplex.ts:298-303): nothing is pivoted, the input buffer is final
        if (b == 0 && tid == 0) {
            copy_state(sout, sin); DevState& s = *sout;
            s.bounded = 0; s.unbounded_var = c.vibc[pc]; s.status = ST_DONE; s.do_pivot = 0;
            s.obj_cell = Min[0]; s.f_final_buf = in_buf;
        }
        return;
    }
    // workgroup 0: cycle check before anything is committed (simplex.ts:305-320)
    if (b == 0 && c.check_cycles) {
        const int n = sin->hist_n;
        if (n >= c.hist_cap) {
            if (tid == 0) {
                copy_state(sout, sin); DevState& s = *sout;
                s.err = ERR_HIST_FULL; s.status = ST_DONE; s.do_pivot = 0; s.obj_cell = Min[0]; s.f_final_buf = in_buf;
            }
            return;
        }
        if (tid == 0) c.hist[n] = make_int2(c.vibr[pr], c.vibc[pc]);
        __syncthreads();
        if (suffix_is_square(c.hist, n + 1, sm.red)) {
            if (tid == 0) {
                copy_state(sout, sin); DevState& s = *sout;
                s.hist_n = n + 1; s.cycle_phase = 2; s.feasible = 0; s.status = ST_DONE; s.do_pivot = 0;
                s.obj_cell = Min[0]; s.f_final_buf = in_buf;
            }
            return;  // the other workgroups write the other buffer, which nobody adopts
        }
    }
    // UNR: the entering column inherits the LEAVING variable (simplex.ts:339-349): its flag for the pricing below; workgroup 0
    // hands the flag arrays of the new basis to the next launch
    if (UNR) {
        const bool leaving_unr = f.urow[launch & 1][pr] != 0;
        bool entering_unr = false;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (pc == c0 + t * JSLP_F_TW + j) {
                    entering_unr = ((um >> (2 * t + j)) & 1u) != 0;
                    um = (um & ~(1u << (2 * t + j))) | ((leaving_unr ? 1u : 0u) << (2 * t + j));
                    if (b == 0) f.urow[(launch + 1) & 1][pr] = entering_unr ? 1 : 0;
                }
        if (b == 0) {
            uint8_t* uco = f.ucol[(launch + 1) & 1];
            const uint8_t* uri = f.urow[launch & 1];
            uint8_t* uro = f.urow[(launch + 1) & 1];
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int col = c0 + t * JSLP_F_TW + j;
                    if (col < W) uco[col] = (um >> (2 * t + j)) & 1u;
                }
            for (int r = tid; r < H; r += JSLP_F_THREADS)
                if (r != pr) uro[r] = uri[r];
        }
    }
    // normalised pivot row in registers (simplex.ts:352-364; anyrow is true in phase 2, see header)
    double2 p[NT];
    bool v0[NT], v1[NT], has_pc[NT];
    int tiny = 0;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int ct = c0 + t * JSLP_F_TW;
        p[t] = make_double2(0, 0);
        if (ct < ld) {
            const double2 pv = *reinterpret_cast<const double2*>(Min + (long long)pr * ld + ct);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int col = ct + j;
                const double val = j ? pv.y : pv.x;
                double v = 0.0;
                if (col < W) {
                    const bool innz = nonzero16(val);
                    v = innz ? val / quot : 0.0;
                    if (col == pc) v = 1.0 / quot;
                    if (innz && !nonzero16(v) && v != 0.0) { tiny = 1; v = 0.0; }
                }
                if (j) p[t].y = v; else p[t].x = v;
            }
        }
        v0[t] = nonzero16(p[t].x); v1[t] = nonzero16(p[t].y);
        has_pc[t] = ct < ld && ((pc == ct) || (pc == ct + 1));
    }
    if (OPT) {  // (without optional objectives the entering cost is > precision: some other row always has an entry)
        if (__syncthreads_or(tiny) && !nonzero16(k0)) {
            if (b == 0 && tid == 0) {
                copy_state(sout, sin); DevState& s = *sout;
                s.status = ST_P1_SLOW; s.do_pivot = 0; s.f_final_buf = in_buf;
            }
            return;
        }
        if (b == 0) fused_update_oo<NT>(c, f.oo[in_buf], f.oo[in_buf ^ 1], c0, pc, quot, p);
    }
    (void)tiny;
    // updated cost row (every workgroup derives the same values), priced for pivot t+1
    double2 n0[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        n0[t] = row0[t];
        if (nonzero16(k0)) {
            if (v0[t]) n0[t].x = eliminate(n0[t].x, k0, p[t].x);
            if (v1[t]) n0[t].y = eliminate(n0[t].y, k0, p[t].y);
            if (has_pc[t]) { const double nv = -k0 / quot; if (pc == c0 + t * JSLP_F_TW) n0[t].x = nv; else n0[t].y = nv; }
        }
    }
    int neg = 0;
    int cn = price_row<NT, UNR>(n0, c0, um, c, sm, &neg);  // 0 => optimal after this pivot
    if (OPT && cn == 0) cn = price_optional<NT, UNR>(n0, c0, um, c, sm, f.oo[in_buf], true, pc, quot, p, &neg);

    // ---- stream my rows ---------------------------------------------------------------------------
    // Round 6: (a) ROLLING prefetch -- as soon as row i of a step (a group of JSLP_F_RG rows x one tile) has been stored, the load of
    // row i of the NEXT step goes out into the same registers: the loads of a step used to leave together only after the previous step's last store (and,
    // behind the two barriers of the candidate collection, after those too), so every step began with one exposed memory latency; (b) the entering
    // column's new entries are collected for up to JSLP_F_COLLECT rows between barriers (they were 8: two barriers per step) and lane j of wave 0 then
    // looks at row j of that block -- the (q, r) / min-rdeg orders of the reduction are total, which lane sees which row does not matter.
    FCand best = fcand_none();  // kept by wave 0: lane j sees rows r_begin + j, + JSLP_F_COLLECT, ...
#ifndef JSLP_F_ROLL_MAXNT
#define JSLP_F_ROLL_MAXNT 1
#endif
#ifndef JSLP_F_COLLECT_ROLL
#define JSLP_F_COLLECT_ROLL JSLP_F_COLLECT
#endif
#ifndef JSLP_F_COLLECT_STEP
#define JSLP_F_COLLECT_STEP JSLP_F_RG
#endif
    constexpr int CR = JSLP_F_COLLECT_ROLL, CS = JSLP_F_COLLECT_STEP;  // rows between two collection barriers, rolling / step-at-a-time loop
    static_assert(CR <= JSLP_F_COLLECT && CS <= JSLP_F_COLLECT && (CR & (CR - 1)) == 0 && (CS & (CS - 1)) == 0 && CR >= JSLP_F_RG && CS >= JSLP_F_RG, "collection block");
    if (NT <= JSLP_F_ROLL_MAXNT) {
        const int n_steps = ((r_end - r_begin + JSLP_F_RG - 1) / JSLP_F_RG) * NT;
        int g0 = r_begin, t_cur = 0;
        for (int step = 0; step < n_steps; step++) {
            const int tn = (t_cur + 1 == NT) ? 0 : t_cur + 1, gn = (t_cur + 1 == NT) ? g0 + JSLP_F_RG : g0;  // the next step
            const bool more = step + 1 < n_steps;
            const int ct = c0 + t_cur * JSLP_F_TW, ctn = c0 + tn * JSLP_F_TW;
            const bool colok = ct < ld, colokn = ctn < ld;
            // (NT is a template parameter but the tile of a step is not a compile-time constant any more: p / v0 / v1 / has_pc are picked by selects)
            double2 pt = p[0]; bool v0t = v0[0], v1t = v1[0], hpt = has_pc[0];
    #pragma unroll
            for (int t = 1; t < NT; t++)
                if (t_cur == t) { pt = p[t]; v0t = v0[t]; v1t = v1[t]; hpt = has_pc[t]; }
    #pragma unroll
            for (int i = 0; i < JSLP_F_RG; i++) {
                const int r = g0 + i;
                if (r < r_end) {
                    double2 x = a[i];
                    if (r == pr) {
                        x = pt;
                    } else if (nonzero16(k[i])) {
                        if (v0t) x.x = eliminate(x.x, k[i], pt.x);
                        if (v1t) x.y = eliminate(x.y, k[i], pt.y);
                        if (hpt) { const double nv = -k[i] / quot; if (pc == ct) x.x = nv; else x.y = nv; }
                    }
                    if (colok) st_stream(Mout + (long long)r * ld + ct, x, f.nt);
                    if (cn != 0) {
                        const int slot = (r - r_begin) & (CR - 1);
                        if (cn == ct) sm.col[slot] = x.x; else if (cn == ct + 1) sm.col[slot] = x.y;
                        if (t_cur == 0 && tid == 0) sm.rhs[slot] = x.x;
                    }
                }
                if (more) {  // row i of the next step: its registers are free now
                    const int rn = gn + i;
                    if (tn == 0) k[i] = rn < r_end ? pin[rn] : 0.0;
                    a[i] = make_double2(0, 0);
                    if (rn < r_end && colokn) a[i] = ld_stream(Min + (long long)rn * ld + ctn, f.nt);
                }
            }
            // a block of CR rows is complete (or the last row is done): wave 0 looks at it
            const int done_rows = min(r_end, g0 + JSLP_F_RG) - r_begin;
            if (cn != 0 && t_cur == NT - 1 && ((done_rows & (CR - 1)) == 0 || !more)) {
                __syncthreads();
                const int blk0 = r_begin + ((done_rows - 1) & ~(CR - 1));
                if (tid < CR && blk0 + tid < r_begin + done_rows) {
                    const int r = blk0 + tid;
                    const double colv = sm.col[tid];
                    pout[r] = colv;
                    if (r >= 1) fcand_consider(best, r, colv, sm.rhs[tid], precision, neg);
                }
                if (more) __syncthreads();
            }
            g0 = gn; t_cur = tn;
        }
    } else {
        // two or more tiles per lane: a step's loads leave together, as before (rolling measured slower there: profiles/r06_streaming_rolling_prefetch.md)
        for (int g0 = r_begin; g0 < r_end; g0 += JSLP_F_RG) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int ct = c0 + t * JSLP_F_TW;
                const bool colok = ct < ld;
                if (g0 != r_begin || t != 0) {  // the first group's first tile was prefetched at the top
#pragma unroll
                    for (int i = 0; i < JSLP_F_RG; i++) {
                        const int r = g0 + i;
                        if (t == 0) k[i] = r < r_end ? pin[r] : 0.0;
                        a[i] = make_double2(0, 0);
                        if (r < r_end && colok) a[i] = ld_stream(Min + (long long)r * ld + ct, f.nt);
                    }
                }
#pragma unroll
                for (int i = 0; i < JSLP_F_RG; i++) {
                    const int r = g0 + i;
                    if (r >= r_end) break;
                    double2 x = a[i];
                    if (r == pr) {
                        x = p[t];
                    } else if (nonzero16(k[i])) {
                        if (v0[t]) x.x = eliminate(x.x, k[i], p[t].x);
                        if (v1[t]) x.y = eliminate(x.y, k[i], p[t].y);
                        if (has_pc[t]) { const double nv = -k[i] / quot; if (pc == ct) x.x = nv; else x.y = nv; }
                    }
                    if (colok) st_stream(Mout + (long long)r * ld + ct, x, f.nt);
                    if (cn != 0) {
                        const int slot = (r - r_begin) & (CS - 1);
                        if (cn == ct) sm.col[slot] = x.x; else if (cn == ct + 1) sm.col[slot] = x.y;
                        if (t == 0 && tid == 0) sm.rhs[slot] = x.x;
                    }
                }
            }
            const int done_rows = min(r_end, g0 + JSLP_F_RG) - r_begin;
            const bool more = g0 + JSLP_F_RG < r_end;
            if (cn != 0 && ((done_rows & (CS - 1)) == 0 || !more)) {
                __syncthreads();
                const int blk0 = r_begin + ((done_rows - 1) & ~(CS - 1));
                if (tid < CS && blk0 + tid < r_begin + done_rows) {
                    const int r = blk0 + tid;
                    const double colv = sm.col[tid];
                    pout[r] = colv;
                    if (r >= 1) fcand_consider(best, r, colv, sm.rhs[tid], precision, neg);
                }
                if (more) __syncthreads();
            }
        }
    }
    if (cn != 0 && tid < 64) {
        best = fcand_wave_reduce(best);
        if (tid == 0) cout[b] = best;
    }
    // ---- workgroup 0 commits the pivot (simplex.ts:339-349) and publishes the next state -----------------
    if (b == 0 && tid == 0) {
        copy_state(sout, sin); DevState& s = *sout;
        const int leaving = c.vibr[pr], entering = c.vibc[pc];
        c.vibr[pr] = entering;
        c.vibc[pc] = leaving;
        c.rbv[entering] = pr;
        c.rbv[leaving] = -1;
        c.cbv[entering] = -1;
        c.cbv[leaving] = pc;
        if (s.trace_n < c.trace_cap) c.trace[s.trace_n] = make_int2(pr, pc);
        s.trace_n += 1;
        if (c.check_cycles) s.hist_n += 1;
        s.it2 += 1;
        s.iters_left -= 1;
        s.pr = pr; s.pc = pc; s.quot = quot;
        s.f_pc = cn | (neg << 30);
        s.f_final_buf = in_buf ^ 1;
        if (cn == 0) {  // optimal after this pivot
            s.status = ST_DONE; s.optimal = 1; s.do_pivot = 0;
            s.obj_cell = n0[0].x;  // thread 0 owns column 0 of the updated cost row
        }
    }
}

// ===================================================================================================
// Fused PHASE 1 (simplex.ts:25-98), one launch per pivot, same out-of-place streaming as k_pivot_fused.
// What launch t-1 leaves behind is only the per-workgroup "most negative RHS" candidate of its rows (FCand: q = the RHS value,
// r = the row).  Launch t reduces them to the leaving row pr (none: phase 1 is over -> ST_PHASE1_DONE, the phase-2 pipeline
// takes over), EVERY workgroup then derives the entering column from row pr and the cost row (both read from the input buffer:
// 2 x 16-32 KB of L2 hits; max -cost/coef over unrestricted or coef < -precision, first index on ties), loads its rows'
// entries of that column, and streams its rows; while the updated rows are in registers it collects the next candidate.
// Workgroup 0 runs the cycle check and commits the maps.
// The one case a workgroup cannot decide alone: simplex.ts:381-383 zeroes the tiny (< 1e-16, non-zero) entries of the stored
// pivot row iff ANY other row has an entry in the pivot column.  The cost row's entry (every workgroup has it) settles that
// almost always; when it does not and a tiny entry exists, nothing is pivoted and the state says ST_P1_SLOW: the host runs
// that ONE pivot through k_select + k_update and restarts this pipeline.
// ===================================================================================================
#define JSLP_F_RPB_MAX 64   // rows per workgroup (cap_rows <= 64 * 256)
template <int NT, bool UNR>
__global__ void __launch_bounds__(JSLP_F_THREADS) k_fused_p1(FusedCtx f, int launch) {
    __shared__ FSmem sm;
    __shared__ double s_kcol[JSLP_F_RPB_MAX];
    __shared__ double s_xq[2];
    const Ctx& c = f.c;
    const int tid = threadIdx.x, b = blockIdx.x;
    const bool init = launch == 0;
    const DevState* sin = init ? c.st : f.fst[launch & 1];
    DevState* sout = f.fst[(launch + 1) & 1];
    // (launch 0 does not pivot: it reads buf[0] and leaves the tableau there; launch t >= 1 reads what launch t-1 wrote)
    const int in_buf = init ? 0 : ((launch - 1) & 1);
    const double* Min = f.buf[in_buf];
    double* Mout = f.buf[in_buf ^ 1];
    const FCand* cin = f.cands[launch & 1];
    FCand* cout = f.cands[(launch + 1) & 1];
    const int ld = c.ld, W = c.W;
    const double precision = c.precision;
    const int H = f.H;
    const int c0 = tid * 2;
    const int r_begin = b * f.rpb, r_end = min(H, r_begin + f.rpb);
    const int status = sin->status;
    const bool live = status == ST_RUNNING && sin->phase == 1;
    if (!live) {  // phase 1 already ended (or never ran): carry the state forward
        if (b == 0 && tid == 0) { copy_state(sout, sin); if (init) sout->f_final_buf = 0; }
        return;
    }
    if (init) {  // candidates of the first pivot; the flag arrays of the unrestricted variables
        FCand best = fcand_none();
        if (tid < 64) {
            for (int r = r_begin + tid; r < r_end; r += 64) {
                const double v = Min[(long long)r * ld];
                if (r >= 1 && v < -precision && (best.r == 0 || v < best.q)) { best.q = v; best.r = r; }  // (rows ascend per lane)
            }
            best = fcand_wave_reduce(best);
            if (tid == 0) cout[b] = best;
        }
        if (UNR && b == 0) {
            uint8_t* uco = f.ucol[(launch + 1) & 1];
            uint8_t* uro = f.urow[(launch + 1) & 1];
            for (int col = tid; col < W; col += JSLP_F_THREADS) uco[col] = col >= 1 ? c.unr[c.vibc[col]] : 0;
            for (int r = tid; r < H; r += JSLP_F_THREADS) uro[r] = r >= 1 ? c.unr[c.vibr[r]] : 0;
        }
        if (b == 0 && tid == 0) { copy_state(sout, sin); sout->f_final_buf = 0; sout->do_pivot = 0; }
        return;
    }
    // ---- STEP ---------------------------------------------------------------------------------------------------------
    if (sin->iters_left <= 0) {
        if (b == 0 && tid == 0) {
            copy_state(sout, sin); DevState& s = *sout;
            s.err = ERR_ITER_LIMIT; s.status = ST_DONE; s.do_pivot = 0; s.obj_cell = Min[0]; s.f_final_buf = in_buf;
        }
        return;
    }
    FCand mine = fcand_none();
    if (tid < f.G) mine = cin[tid];
    const FCand win = fcand_block_reduce(mine, sm);
    if (win.r == 0) {  // simplex.ts:51-54: feasible, phase 2 starts with a fresh history
        if (b == 0 && tid == 0) {
            copy_state(sout, sin); DevState& s = *sout;
            s.feasible = 1; s.phase = 2; s.entered_phase2 = 1; s.hist_n = 0; s.status = ST_PHASE1_DONE; s.do_pivot = 0;
            s.f_final_buf = in_buf;
        }
        return;
    }
    const int pr = win.r;
    // entering column (simplex.ts:56-71) from row pr and the cost row, both as the previous launch left them
    unsigned um = 0;
    double2 rowp[NT], row0[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int ct = c0 + t * JSLP_F_TW;
        rowp[t] = make_double2(0, 0); row0[t] = make_double2(0, 0);
        if (ct < ld) {
            rowp[t] = *reinterpret_cast<const double2*>(Min + (long long)pr * ld + ct);
            row0[t] = *reinterpret_cast<const double2*>(Min + ct);
        }
        if (UNR) {
            const uint8_t* ucin = f.ucol[launch & 1];
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (ct + j >= 1 && ct + j < W && ucin[ct + j] != 0) um |= 1u << (2 * t + j);
        }
    }
    Cand q; q.v = -INFINITY; q.i = 0; q.b = 0;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = c0 + t * JSLP_F_TW + j;
            const double coef = j ? rowp[t].y : rowp[t].x;
            const bool un = UNR && ((um >> (2 * t + j)) & 1u);
            if (col >= 1 && col < W && (un || coef < -precision)) {
                const double quo = -(j ? row0[t].y : row0[t].x) / coef;
                const bool take = q.v < quo;  // (my columns ascend: ties keep the earlier one)
                q.v = take ? quo : q.v;
                q.i = take ? col : q.i;
            }
        }
    q = block_reduce(q, MaxFirst(), sm.red);
    if (q.i == 0) {  // simplex.ts:73-76: infeasible
        if (b == 0 && tid == 0) {
            copy_state(sout, sin); DevState& s = *sout;
            s.feasible = 0; s.status = ST_DONE; s.do_pivot = 0; s.obj_cell = Min[0]; s.f_final_buf = in_buf;
        }
        return;
    }
    const int pc = q.i;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (pc == c0 + t * JSLP_F_TW + j) { s_xq[0] = j ? rowp[t].y : rowp[t].x; s_xq[1] = j ? row0[t].y : row0[t].x; }
    // my rows' entries of the pivot column (one strided trip), while workgroup 0 runs the cycle check
    if (tid < JSLP_F_RPB_MAX && r_begin + tid < r_end) s_kcol[tid] = Min[(long long)(r_begin + tid) * ld + pc];
    if (b == 0 && c.check_cycles) {  // simplex.ts:78-93
        const int n = sin->hist_n;
        if (n >= c.hist_cap) {
            if (tid == 0) {
                copy_state(sout, sin); DevState& s = *sout;
                s.err = ERR_HIST_FULL; s.status = ST_DONE; s.do_pivot = 0; s.obj_cell = Min[0]; s.f_final_buf = in_buf;
            }
            return;
        }
        if (tid == 0) c.hist[n] = make_int2(c.vibr[pr], c.vibc[pc]);
        __syncthreads();
        if (suffix_is_square(c.hist, n + 1, sm.red)) {
            if (tid == 0) {
                copy_state(sout, sin); DevState& s = *sout;
                s.hist_n = n + 1; s.cycle_phase = 1; s.feasible = 0; s.status = ST_DONE; s.do_pivot = 0;
                s.obj_cell = Min[0]; s.f_final_buf = in_buf;
            }
            return;  // the other workgroups write the other buffer, which nobody adopts
        }
    }
    __syncthreads();
    const double quot = s_xq[0], k0 = s_xq[1];
    // normalised pivot row (simplex.ts:352-364) under "some other row has an entry in column pc"
    double2 p[NT];
    bool v0[NT], v1[NT], has_pc[NT];
    int tiny = 0;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int ct = c0 + t * JSLP_F_TW;
        p[t] = make_double2(0, 0);
        if (ct < ld) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int col = ct + j;
                const double val = j ? rowp[t].y : rowp[t].x;
                double v = 0.0;
                if (col < W) {
                    const bool innz = nonzero16(val);
                    v = innz ? val / quot : 0.0;
                    if (col == pc) v = 1.0 / quot;
                    if (innz && !nonzero16(v) && v != 0.0) { tiny = 1; v = 0.0; }
                }
                if (j) p[t].y = v; else p[t].x = v;
            }
        }
        v0[t] = nonzero16(p[t].x); v1[t] = nonzero16(p[t].y);
        has_pc[t] = ct < ld && ((pc == ct) || (pc == ct + 1));
    }
    if (__syncthreads_or(tiny) && !nonzero16(k0)) {
        // whether the tiny entries are zeroed depends on the other workgroups' rows: not decidable here
        if (b == 0 && tid == 0) {
            copy_state(sout, sin); DevState& s = *sout;
            s.status = ST_P1_SLOW; s.do_pivot = 0; s.f_final_buf = in_buf;
        }
        return;
    }
    if (c.n_opt > 0 && b == 0) fused_update_oo<NT>(c, f.oo[in_buf], f.oo[in_buf ^ 1], c0, pc, quot, p);  // simplex.ts:394-412
    if (UNR) {  // the entering column inherits the leaving variable's flag; workgroup 0 hands the flag arrays on
        const bool leaving_unr = f.urow[launch & 1][pr] != 0;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (pc == c0 + t * JSLP_F_TW + j) {
                    const bool entering_unr = ((um >> (2 * t + j)) & 1u) != 0;
                    um = (um & ~(1u << (2 * t + j))) | ((leaving_unr ? 1u : 0u) << (2 * t + j));
                    if (b == 0) f.urow[(launch + 1) & 1][pr] = entering_unr ? 1 : 0;
                }
        if (b == 0) {
            uint8_t* uco = f.ucol[(launch + 1) & 1];
            const uint8_t* uri = f.urow[launch & 1];
            uint8_t* uro = f.urow[(launch + 1) & 1];
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int col = c0 + t * JSLP_F_TW + j;
                    if (col < W) uco[col] = (um >> (2 * t + j)) & 1u;
                }
            for (int r = tid; r < H; r += JSLP_F_THREADS)
                if (r != pr) uro[r] = uri[r];
        }
    }
    // ---- stream my rows; the next pivot's candidate (most negative RHS below -precision, first row on ties) ---------------
    FCand best = fcand_none();  // kept by lanes 0..7 of wave 0: lane i sees rows r_begin + i, + 8, ... in order
    double2 a[JSLP_F_RG];
    for (int g0 = r_begin; g0 < r_end; g0 += JSLP_F_RG) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int ct = c0 + t * JSLP_F_TW;
            const bool colok = ct < ld;
#pragma unroll
            for (int i = 0; i < JSLP_F_RG; i++) {
                const int r = g0 + i;
                a[i] = make_double2(0, 0);
                if (r < r_end && colok) a[i] = ld_stream(Min + (long long)r * ld + ct, f.nt);
            }
#pragma unroll
            for (int i = 0; i < JSLP_F_RG; i++) {
                const int r = g0 + i;
                if (r >= r_end) break;
                const double ki = s_kcol[r - r_begin];
                double2 x = a[i];
                if (r == pr) {
                    x = p[t];
                } else if (nonzero16(ki)) {
                    if (v0[t]) x.x = eliminate(x.x, ki, p[t].x);
                    if (v1[t]) x.y = eliminate(x.y, ki, p[t].y);
                    if (has_pc[t]) { const double nv = -ki / quot; if (pc == ct) x.x = nv; else x.y = nv; }
                }
                if (colok) st_stream(Mout + (long long)r * ld + ct, x, f.nt);
                if (t == 0 && tid == 0) sm.rhs[i] = x.x;
            }
        }
        __syncthreads();
        if (tid < JSLP_F_RG && g0 + tid < r_end) {
            const int r = g0 + tid;
            const double v = sm.rhs[tid];
            if (r >= 1 && v < -precision && (best.r == 0 || v < best.q)) { best.q = v; best.r = r; }
        }
        __syncthreads();
    }
    if (tid < 64) {
        best = fcand_wave_reduce(best);
        if (tid == 0) cout[b] = best;
    }
    if (b == 0 && tid == 0) {  // simplex.ts:339-349
        copy_state(sout, sin); DevState& s = *sout;
        const int leaving = c.vibr[pr], entering = c.vibc[pc];
        c.vibr[pr] = entering;
        c.vibc[pc] = leaving;
        c.rbv[entering] = pr;
        c.rbv[leaving] = -1;
        c.cbv[entering] = -1;
        c.cbv[leaving] = pc;
        if (s.trace_n < c.trace_cap) c.trace[s.trace_n] = make_int2(pr, pc);
        s.trace_n += 1;
        if (c.check_cycles) s.hist_n += 1;
        s.it1 += 1;
        s.iters_left -= 1;
        s.pr = pr; s.pc = pc; s.quot = quot;
        s.f_final_buf = in_buf ^ 1;
    }
}

// after ST_P1_SLOW was adopted into the canonical state: k_select takes the next pivot (status = ST_RUNNING); then, in phase 2,
// the pipeline starts over from its first launch (status = ST_PHASE1_DONE)
__global__ void k_p1_resume(DevState* st, int status) { st->status = status; }

// End of the fused pipeline: adopt the final state and make buf[0] hold the final tableau.
__global__ void __launch_bounds__(256) k_fused_finish(FusedCtx f, int last_launch) {
    const DevState* fin = f.fst[(last_launch + 1) & 1];
    const int H = f.H;
    if (fin->f_final_buf == 1) {
        const long long n2 = (long long)H * f.c.ld / 2;
        const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long long)gridDim.x * blockDim.x;
        const double2* src = reinterpret_cast<const double2*>(f.buf[1]);
        double2* dst = reinterpret_cast<double2*>(f.buf[0]);
        for (long long i = tid; i < n2; i += nt) dst[i] = src[i];
        const long long no = (long long)f.c.n_opt * f.c.ld;  // the optional objectives travel with their buffer
        for (long long i = tid; i < no; i += nt) f.oo[0][i] = f.oo[1][i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        copy_state(f.c.st, fin);
    }
}

