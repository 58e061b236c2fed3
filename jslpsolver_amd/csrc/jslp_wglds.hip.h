// jslp_wglds.hip.h -- the one-workgroup simplex() with its selection state in LDS.
// Included by jslp_kernels.hip.h after the core (needs Ctx / Slots / Smem / Cand / block_reduce / the ST_* enums).
#pragma once

// ===================================================================================================
// One workgroup = one whole simplex() of one tableau copy, like simplex_wg (jslp_core.inc.h), for tableaus whose
// selection state fits in LDS.  What a pivot of the generic kernel costs is not bandwidth but a chain of ~10 dependent
// global-memory round trips (state words, cost row, pivot column written to and re-read from HBM, the RHS mirror, the
// normalised pivot row) separated by workgroup barriers.  Here the cost row, the RHS column, the pivot column and the
// normalised pivot row live in LDS for the whole solve and the loop-carried scalars in registers, so a pivot touches
// global memory three times: the strided pivot-column gather, the pivot row (+ the two map entries it swaps), and the
// rows the reference's gate lets through (simplex.ts:370-375).  Everything else -- pricing (simplex.ts:118-219), the
// ratio test (:271-296), the row-gate compaction -- runs out of LDS.
// Same selection rules, same arithmetic, same order-free reductions as select_step / prepare_pivot / update_rows_wg:
// the parity tests run every fixture through both.
// Not handled here (the host launches the generic kernel instead): optional objectives, tableaus whose four vectors
// exceed the dynamic LDS budget.
// ===================================================================================================
struct WgLds {
    double* r0;     // [ld]   cost row (row 0), kept in step with A by the row update
    double* prow;   // [ld]   normalised pivot row of the current pivot
    double* pcol;   // [Hc]   pivot column of the current pivot
    double* rhs;    // [Hc]   column 0 (the authoritative copy during the solve; written back to the slot's mirror at the end)
    int32_t* list;  // [Hc]   rows passing the gate / rows to restore
};
__host__ __device__ __forceinline__ size_t wglds_bytes(int ld, int cap_rows) {
    const size_t hc = ((size_t)cap_rows + 1) & ~(size_t)1;
    return 8 * (2 * (size_t)ld + 2 * hc) + 4 * hc;
}
__device__ __forceinline__ WgLds wglds_carve(double* base, int ld, int cap_rows) {
    const int hc = (cap_rows + 1) & ~1;
    WgLds L;
    L.r0 = base;
    L.prow = base + ld;
    L.pcol = base + 2 * ld;
    L.rhs = base + 2 * ld + hc;
    L.list = reinterpret_cast<int32_t*>(base + 2 * ld + 2 * hc);
    return L;
}

// one gated row, one wave: row <- row - k * prow on the live columns (simplex.ts:376-387); column 0 and row 0 are mirrored in LDS
__device__ __forceinline__ void wglds_update_row(const Ctx& c, const WgLds& L, int r, double k, int pc, double quot, int lane) {
    const int ld = c.ld;
    double* row = c.A + (long long)r * ld;
    for (int c0 = lane * 2; c0 < ld; c0 += 128) {
        const double2 p = *reinterpret_cast<const double2*>(L.prow + c0);
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
        if (c0 == 0) L.rhs[r] = x.x;
        if (r == 0) *reinterpret_cast<double2*>(L.r0 + c0) = x;
    }
}

__device__ void simplex_wg_lds(const Ctx& c, Smem& sm, const WgLds& L, int iters_cap) {
    DevState* st = c.st;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, w = tid >> 6, nw = nt >> 6;
    if (tid == 0) begin_simplex(st, iters_cap);
    __syncthreads();
    const int H = st->H, W = c.W, ld = c.ld;  // the height is fixed during a simplex() call
    const double precision = c.precision;
    double* A = c.A;
    {   // column 0 and row 0 into LDS (the slot's contiguous RHS mirror when it is valid, else one strided gather)
        const bool mirrored = c.rhs && st->rhs_valid;
        for (int r = tid; r < H; r += nt) L.rhs[r] = mirrored ? c.rhs[r] : A[(long long)r * ld];
        for (int col = tid; col < ld; col += nt) L.r0[col] = A[col];
    }
    const int err0 = st->err;
    long long trace_n = st->trace_n;
    __syncthreads();
    if (err0 != ERR_NONE) {  // a bad cut list: report, do not solve
        if (tid == 0) finish(c);
        return;
    }
    int phase = 1, it1 = 0, it2 = 0, hist_n = 0, iters_left = iters_cap, entered2 = 0;
    // outcome: 0 running, 1 optimal, 2 unbounded, 3 cycle, 4 infeasible, 5 iteration cap, 6 history full
    int outcome = 0, unbounded_col = 0;
    while (outcome == 0) {
        if (iters_left <= 0) { outcome = 5; break; }
        int pr = 0, pc = 0, neg_flag = 0;
        if (phase == 1) {
            // leaving row: most negative RHS below -precision, first index on ties (simplex.ts:39-49)
            Cand best; best.v = -precision; best.i = 0; best.b = 0;
            for (int r = 1 + tid; r < H; r += nt) {
                const double v = L.rhs[r];
                if (v < best.v) { best.v = v; best.i = r; }
            }
            best = block_reduce(best, MinFirst(), sm);
            if (best.i == 0) {  // :51-54 feasible: phase 2 starts in this same iteration with a fresh history (:102)
                phase = 2; entered2 = 1; hist_n = 0;
            } else {
                pr = best.i;
                // entering column: max -cost/coef over unrestricted or coef < -precision (simplex.ts:56-71)
                const double* row = A + (long long)pr * ld;
                Cand q; q.v = -INFINITY; q.i = 0; q.b = 0;
                for (int col = 1 + tid; col < W; col += nt) {
                    const double coef = row[col];
                    const bool un = c.has_unr && c.unr[c.vibc[col]] != 0;
                    if (un || coef < -precision) {
                        const double quo = -L.r0[col] / coef;
                        if (q.v < quo) { q.v = quo; q.i = col; }
                    }
                }
                q = block_reduce(q, MaxFirst(), sm);
                if (q.i == 0) { outcome = 4; break; }  // :73-76 infeasible
                pc = q.i;
                for (int r = tid; r < H; r += nt) L.pcol[r] = A[(long long)r * ld + pc];
                __syncthreads();
            }
        }
        if (phase == 2) {
            // Dantzig pricing with the reference's batch rule (simplex.ts:118-219) on the LDS cost row
            Cand e; e.v = precision; e.i = 0; e.b = 0;
            for (int col = 1 + tid; col < W; col += nt) {
                const double rc = L.r0[col];
                const bool un = c.has_unr && c.unr[c.vibc[col]] != 0;
                const int b = c.use_partial ? (col - 1) / c.batch : 0;
                const double val = (un && rc < 0) ? -rc : rc;
                if (val > precision) {
                    Cand cand; cand.v = val; cand.i = col; cand.b = b;
                    const bool take = PriceFirst()(cand, e);
                    e.v = take ? cand.v : e.v;
                    e.i = take ? cand.i : e.i;
                    e.b = take ? cand.b : e.b;
                }
            }
            e = block_reduce(e, PriceFirst(), sm);
            if (e.i == 0) { outcome = 1; break; }  // optimal (simplex.ts:265-269)
            pc = e.i;
            {
                const double rc = L.r0[pc];
                const bool un = c.has_unr && c.unr[c.vibc[pc]] != 0;
                neg_flag = (un && rc < 0) ? 1 : 0;
            }
            // ratio test (simplex.ts:271-296) in its order-free form; the strided column gather is the pivot's first global trip
            Cand m; m.v = INFINITY; m.i = 0; m.b = 0;
            int rdeg = 0x7fffffff;
            for (int r = tid; r < H; r += nt) {
                const double colv = A[(long long)r * ld + pc];
                L.pcol[r] = colv;
                if (r == 0) continue;
                const double rhs = L.rhs[r];
                if (-precision < colv && colv < precision) continue;
                if (colv > 0 && precision > rhs && rhs > -precision) {
                    if (r < rdeg) rdeg = r;
                    continue;
                }
                const double quo = neg_flag ? -rhs / colv : rhs / colv;
                if (quo > precision && m.v > quo) { m.v = quo; m.i = r; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const int o = __shfl_down(rdeg, off, 64);
                rdeg = o < rdeg ? o : rdeg;
            }
            __syncthreads();
            if (tid == 0) sm.flag = 0x7fffffff;
            __syncthreads();
            if (lane == 0 && rdeg != 0x7fffffff) atomicMin(&sm.flag, rdeg);
            m = block_reduce(m, MinFirst(), sm);  // contains the barriers that publish sm.flag and L.pcol
            rdeg = sm.flag;
            if (rdeg != 0x7fffffff) pr = rdeg;
            else if (m.i != 0) pr = m.i;
            else { outcome = 2; unbounded_col = pc; break; }  // unbounded (simplex.ts:298-303)
        }
        // cycle check (simplex.ts:78-93 / 305-320): append first, test, stop WITHOUT pivoting on a hit
        if (c.check_cycles) {
            if (hist_n >= c.hist_cap) { outcome = 6; break; }
            if (tid == 0) c.hist[hist_n] = make_int2(c.vibr[pr], c.vibc[pc]);
            __syncthreads();
            hist_n += 1;
            if (suffix_is_square(c.hist, hist_n, sm)) { outcome = 3; break; }
        }
        // ---- pivot (simplex.ts:330-413): maps, normalised pivot row (second global trip), then the gated rows (third) ----
        const double quot = L.pcol[pr];  // = A[pr, pc] (:335)
        int any = 0, n_gated = 0;
        for (int r = tid; r < H; r += nt) {
            const int gated = (r != pr && nonzero16(L.pcol[r]));
            any |= gated;
            n_gated += gated;
        }
        const int anyrow = __syncthreads_or(any);
        double* prow_A = A + (long long)pr * ld;
        int n_cols = 0;
        for (int col = tid; col < ld; col += nt) {
            double v = 0.0;
            if (col < W) {
                const double val = prow_A[col];
                const bool innz = nonzero16(val);       // :356
                v = innz ? val / quot : 0.0;            // :357 / :361
                if (col == pc) v = 1.0 / quot;          // :364
                if (innz && anyrow && !nonzero16(v) && v != 0.0) v = 0.0;  // :381-383
                prow_A[col] = v;
                if (col == 0) L.rhs[pr] = v;
                n_cols += (nonzero16(v) || col == pc) ? 1 : 0;
            }
            L.prow[col] = v;
        }
        if (tid == 0) {
            const int leaving = c.vibr[pr], entering = c.vibc[pc];  // :339-349
            c.vibr[pr] = entering;
            c.vibc[pc] = leaving;
            c.rbv[entering] = pr;
            c.rbv[leaving] = -1;
            c.cbv[entering] = -1;
            c.cbv[leaving] = pc;
            if (trace_n < c.trace_cap) c.trace[trace_n] = make_int2(pr, pc);
            sm.flag2 = 0;  // length of the gated-row list
            c.dirty[pr] = 1;
        }
        trace_n += 1;
        if (phase == 1) it1 += 1; else it2 += 1;
        iters_left -= 1;
        if (c.cnt) {  // work counters (uniform branch)
            for (int off = 32; off > 0; off >>= 1) { n_gated += __shfl_down(n_gated, off, 64); n_cols += __shfl_down(n_cols, off, 64); }
            __syncthreads();
            if (tid == 0) { sm.flag = 0; sm.wave[0].i = 0; }
            __syncthreads();
            if (lane == 0) { atomicAdd(&sm.flag, n_gated); atomicAdd(&sm.wave[0].i, n_cols); }
            __syncthreads();
            if (tid == 0) {
                atomicAdd(c.cnt + CNT_CELLS, (cnt_t)sm.flag * (cnt_t)sm.wave[0].i);
                atomicAdd(c.cnt + CNT_ROWS, (cnt_t)sm.flag);
            }
        }
        __syncthreads();  // L.prow complete, sm.flag2 reset
        // the rows that pass the reference's gate, compacted into the LDS list (a Monster_II pivot: ~10 of 945)
        for (int r = tid; r < H; r += nt) {
            if (r != pr && nonzero16(L.pcol[r])) {
                const int idx = atomicAdd(&sm.flag2, 1);
                L.list[idx] = r;  // the list holds every row: it cannot overflow
                c.dirty[r] = 1;
            }
        }
        __syncthreads();
        const int n = sm.flag2;
        for (int i = w; i < n; i += nw) {
            const int r = L.list[i];
            wglds_update_row(c, L, r, L.pcol[r], pc, quot, lane);
        }
        __syncthreads();
    }
    // ---- epilogue: scalars back into the state, column 0 back into the slot's mirror ---------------------------------------
    if (c.rhs)
        for (int r = tid; r < H; r += nt) c.rhs[r] = L.rhs[r];
    if (tid == 0) {
        st->phase = phase;
        st->it1 = it1;
        st->it2 = it2;
        st->hist_n = hist_n;
        st->iters_left = iters_left;
        st->trace_n = trace_n;
        st->entered_phase2 = entered2;
        if (entered2) st->feasible = 1;
        if (c.rhs) st->rhs_valid = 1;
        if (outcome == 1) st->optimal = 1;
        if (outcome == 2) { st->bounded = 0; st->unbounded_var = c.vibc[unbounded_col]; }
        if (outcome == 3) { st->cycle_phase = phase; st->feasible = 0; }
        if (outcome == 4) st->feasible = 0;
        if (outcome == 5) st->err = ERR_ITER_LIMIT;
        if (outcome == 6) st->err = ERR_HIST_FULL;
        st->status = ST_DONE;
        st->do_pivot = 0;
        st->obj_cell = L.r0[0];
    }
    __syncthreads();
}

// simplex() of slots [first_slot, first_slot + gridDim.x): the LDS twin of k_simplex_wg
template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_simplex_lds(Slots s, int first_slot, int check_cycles, int iters_cap, int cap_rows) {
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    __shared__ Smem sm;
    const Ctx c = slot_ctx(s, first_slot + blockIdx.x, check_cycles);
    const WgLds L = wglds_carve(lds_dyn, s.ld, cap_rows);
    simplex_wg_lds(c, sm, L, iters_cap);
}

// The LDS twin of k_node_wg: ONE branch-and-bound child per workgroup in ONE launch -- restore of the rows the previous node
// dirtied, the index maps, addCutConstraints, simplex() and the read-back (see k_node_wg for the contract).
template <int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS == 512 ? 8 : 4) k_node_lds(Slots s, Snapshot snap, Cuts cuts, int first_node, int check_cycles,
                                                      int iters_cap, int cap_rows, double* rhs_out, int32_t* rows_out,
                                                      DevState* state_out, int out_stride, int first_out,
                                                      unsigned* done_flag, unsigned done_seq) {
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    __shared__ Smem sm;
    const WgLds L = wglds_carve(lds_dyn, s.ld, cap_rows);
    const int slot = blockIdx.x, node = first_node + blockIdx.x, o = first_out + blockIdx.x;
    DevState* st = s.st + slot;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int gen = s.st[0].s_gen, H = s.st[0].s_H, ld2 = s.ld / 2;  // every slot shares slot 0's snapshot scalars
    if (gen == 0 || st->gen != gen) {  // must not happen (host bookkeeping): refuse rather than restore wrongly
        if (tid == 0) {
            st->err = ERR_NOT_SYNCED; st->status = ST_DONE; state_out[o] = *st;
            if (done_flag) { __threadfence_system(); __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
        return;
    }
    double* A = s.A + (long long)slot * s.A_stride;
    uint8_t* dirty = s.dirty + (long long)slot * s.pcol_stride;
    double* rhs = s.rhs + (long long)slot * s.pcol_stride;
    // restore(): the dirty rows, found by all threads at once and compacted into the LDS list
    if (tid == 0) sm.flag2 = 0;
    __syncthreads();
    for (int r = tid; r < H; r += blockDim.x)
        if (dirty[r]) L.list[atomicAdd(&sm.flag2, 1)] = r;
    __syncthreads();
    const int n = sm.flag2;
    if (s.cnt && tid == 0) atomicAdd(s.cnt + CNT_RESTORED, (cnt_t)n);
    const double2* src = reinterpret_cast<const double2*>(snap.A);
    double2* dst = reinterpret_cast<double2*>(A);
    for (int i = w; i < n; i += nw) {
        const int r = L.list[i];
        for (int k = lane; k < ld2; k += 64) dst[(long long)r * ld2 + k] = src[(long long)r * ld2 + k];
        if (lane == 0) { dirty[r] = 0; rhs[r] = snap.rhs[r]; }
    }
    int32_t* vibr = s.vibr + (long long)slot * s.vibr_stride;
    int32_t* vibc = s.vibc + (long long)slot * s.vibc_stride;
    int32_t* rbv = s.rbv + (long long)slot * s.idx_stride;
    int32_t* cbv = s.cbv + (long long)slot * s.idx_stride;
    for (int i = tid; i < H; i += blockDim.x) vibr[i] = snap.vibr[i];
    for (int i = tid; i < s.W; i += blockDim.x) vibc[i] = snap.vibc[i];
    for (int i = tid; i < snap.n_idx; i += blockDim.x) { rbv[i] = snap.rbv[i]; cbv[i] = snap.cbv[i]; }
    if (tid == 0) {
        st->H = H;
        st->last_element_index = s.st[0].s_last_element_index;
        st->err = ERR_NONE;
    }
    __syncthreads();
    add_cuts_slot(s, cuts, slot, node, cap_rows);
    __syncthreads();
    const Ctx c = slot_ctx(s, slot, check_cycles);
    simplex_wg_lds(c, sm, L, iters_cap);
    gather_slot(s, slot, rhs_out, rows_out, state_out, out_stride, o);
    if (done_flag) {
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
