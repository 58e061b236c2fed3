// jslp_resident_pipe.hip.h -- the LEAN register-resident kernel's phase 2: a software-pipelined pivot loop.
// Included by jslp_resident.hip.h between its building blocks (ResCtx / RSmem / ResRegs / price_row_lds) and its kernel.
#pragma once

// ===================================================================================================================
// Why a second loop.  k_simplex_resident's general pivot loop (jslp_resident.hip.h) runs its steps strictly one after the
// other: summary -> publish -> gather -> decide -> fetch the winning row -> update my rows -> price -> summary ...; the row
// update (16 waves x 8 rows x 2 columns of two-rounding eliminations, ~3 k cycles) and the fabric hop of the NEXT pivot's
// summary (~1 us) both sit on the critical path although neither needs the other.  Here the loop is rotated:
//
//   winner row of pivot t lands
//     -> normalise it, update the COST row only, price it                  (entering column of pivot t+1: simplex.ts:118-219)
//     -> the ONE lane that holds column pc(t+1) evaluates what pivot t makes of that column in my rows, lane 0 of the
//        workgroup does the same for column 0, the wave that owns pc(t+1) runs the ratio test on the two       (simplex.ts:271-296)
//     -> the summary of pivot t+1 is published, the row that can win is brought up to date and published
//     -> ONLY NOW the bulk of pivot t's row update (simplex.ts:367-391), while the summaries cross the fabric
//     -> gather, decide, fetch the winning row of pivot t+1 ...
//
// Same arithmetic on the same operands in the same order for every cell (each cell still receives exactly one
// `a - k * p` with both roundings per pivot; "early" values are computed from the same inputs as the bulk update computes
// them later), so the pivot sequence and every bit of the final tableau are unchanged -- the parity tests do not
// distinguish the two loops.
//
// Protocol differences from the general loop (all-gather form):
//   * the ratio-test summary is ONE 16-byte granule per workgroup: {tag32 | q_lo32}{tag16 | kind | row15 | q_hi32} -- the winner's
//     pivot-column entry (`quot`) is no longer published: it is column pc of the winning row, which every workgroup fetches
//     anyway (the lane that holds the column broadcasts it through LDS under the barrier the row flag needs);
//     a workgroup with a degenerate row publishes only that (its quotient candidates cannot win: simplex.ts:285-289);
//   * four waves poll (lane w = workgroup w, one 16-byte sc1 load each) and reduce their 64 summaries in registers (DPP);
//     the four partial results meet in LDS under the barrier that also drains the row stores: no separate decision stage;
//   * 6 workgroup barriers per pivot instead of 8 (cycle check off).
// ===================================================================================================================
#ifndef JSLP_G16_STRIDE
// bytes between two workgroups' 16-byte summary granules: one granule per 64-byte line.  Granules that share a line cost the
// chip-wide all-gather 7.7 k cycles per round against 4.8 k with a line each (tools/micro/xcd_handoff_bench.hip, r03_d): the
// write-through stores of different CUs to one line serialise at the memory side
#define JSLP_G16_STRIDE 64
#endif
#ifndef JSLP_G16_REPL
// copies of every summary granule.  A chip-wide all-gather is bound by its READERS: 256 CUs polling one line are served one
// after the other by that line's memory channel (micro-benchmark, r03_d: 8 publishers -> 256 pollers 4.5 k cycles per round,
// 32 -> 32 pollers 2.2 k).  The publishing wave's lanes 0..REPL-1 store one copy each (one instruction), workgroup b polls copy
// b % REPL: 256 / REPL readers per line.
#define JSLP_G16_REPL 8
#endif
#ifndef JSLP_PIPE_PRUNE
// Speculative row publication pruned inside the XCD.  Every workgroup used to publish the one row of its own that can win
// (16 KB write-through each: 4 MB per pivot chip-wide, one row in 256 consumed) and then had to wait for that burst to drain
// before it could raise its row flag.  Now the workgroups that share an XCD -- b % 8 as observed on gfx950; any grouping is
// CORRECT, see below -- also exchange their summaries through the XCD's own L2 (plain 16-byte store, sc1 load: 2 k cycles per
// exchange against 4.8-7.7 k chip-wide, tools/micro/xcd_handoff_bench.hip) while the row update runs, and a workgroup that
// sees a better candidate than its own among them does not publish.  The chip-wide winner never sees a better candidate, so
// it always publishes: a summary that is late, stale or never visible (wrong guess about the placement) only means one more
// row published, never a missing one.  ~8 rows per pivot instead of 256.
#define JSLP_PIPE_PRUNE 1
#endif
#ifndef JSLP_PIPE_EARLYPOLL
#define JSLP_PIPE_EARLYPOLL 0  // 1: the first poll of the gather is issued before the bulk update and examined after it
#endif

__device__ __forceinline__ double readlane_f64(double x, int src_lane) {  // src_lane must be wave-uniform
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src_lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), src_lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int THREADS, int CPT, int ROWS>
__device__ __forceinline__ void resident_phase2_pipe(const ResCtx& f, RSmem& sm, ResRegs<CPT, ROWS>& R, int it1_start, int it2_start,
                                                     const int (&pb)[CPT]) {
    const Ctx& c = f.c;
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ld = c.ld, W = c.W;
    const double precision = c.precision;
    const int c0 = tid * CPT;
    const bool colok = c0 < ld;
    const int r_begin = b * f.rpb, r_end = min(f.H, r_begin + f.rpb);
    double (&a)[ROWS][CPT] = R.a;
    double (&r0)[CPT] = R.r0;
    typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
    const int pub_stride = (int)((const char*)f.rows_pub[1] - (const char*)f.rows_pub[0]);  // (both carved from one arena, [0] first)
    const auto rsrc_rows = __builtin_amdgcn_make_buffer_rsrc(f.rows_pub[0], 0, pub_stride + f.G * ld * 8, 0x00020000);
    const auto rsrc_g16 = __builtin_amdgcn_make_buffer_rsrc(f.gran16, 0, 2 * JSLP_G16_REPL * JSLP_F_MAXG * JSLP_G16_STRIDE, 0x00020000);
    const auto rsrc_g1 = __builtin_amdgcn_make_buffer_rsrc(f.g1, 0, 2 * JSLP_F_MAXG * 16, 0x00020000);  // [2][8 groups][32 members] x 16 bytes
    const int grp = b & 7, mem = b >> 3;  // my group (the XCD block b is observed to run on) and my place in it
#ifdef JSLP_DEBUG_RESIDENT
    u64_t (&rt_acc)[8] = R.rt_acc;
    u64_t& rt_prev = R.rt_prev;
#endif

    // the pending pivot (its row update has not reached my registers yet)
    double p[CPT];      // its normalised pivot row, my columns
    unsigned nzm = 0;   // bit j: p[j] is non-zero by the reference's test (simplex.ts:379)
    int pr_p = 0, pc_p = 0, par_p = 0;
    bool pend = false;
#pragma unroll
    for (int j = 0; j < CPT; j++) p[j] = 0.0;
    int okslot = 0;

    // row i of mine <- pivot (pr_p, pc_p): exactly the general loop's step F for one row
#define JSLP_PIPE_UPDATE_ROW(i)                                                                             \
    do {                                                                                                    \
        const int r_ = r_begin + (i);                                                                       \
        if (r_ >= r_end) break;                                                                             \
        if (r_ == 0) { /* workgroup 0 mirrors the cost row */                                               \
            _Pragma("unroll") for (int j = 0; j < CPT; j++) a[i][j] = r0[j];                                \
            break;                                                                                          \
        }                                                                                                   \
        if (r_ == pr_p) {                                                                                   \
            _Pragma("unroll") for (int j = 0; j < CPT; j++) a[i][j] = p[j];                                 \
            break;                                                                                          \
        }                                                                                                   \
        const double ki_ = kis[i];                                                                          \
        if (nonzero16(ki_)) {                                                                               \
            _Pragma("unroll") for (int j = 0; j < CPT; j++)                                                 \
                if ((nzm >> j) & 1u) a[i][j] = eliminate(a[i][j], ki_, p[j]);                               \
            if (has_pc_p) {                                                                                 \
                const double nv_ = sm.nv[i];                                                                \
                _Pragma("unroll") for (int j = 0; j < CPT; j++)                                             \
                    if (pc_p == c0 + j) a[i][j] = nv_;                                                      \
            }                                                                                               \
        }                                                                                                   \
    } while (0)

    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < ROWS; i++) sm.rhsb[i] = a[i][0];
        reset_reductions(sm);
    }
    __syncthreads();

    while (R.end_code == 0) {
        const bool has_pc_p = pend && colok && pc_p >= c0 && pc_p < c0 + CPT;
        // ---- exits that hand the tableau on: bring it up to date first --------------------------------------------------
        {
            int leave = 0;
            if ((R.it1 - it1_start) + (R.it2 - it2_start) >= f.iters_cap) leave = 4;
            else if (c.check_cycles && !(R.hist_n < JSLP_R_LHIST && R.hist_n < c.hist_cap)) leave = 8;  // history outgrows LDS: the general kernel continues
            if (leave) { R.end_code = leave; break; }  // (the pending update is applied behind the loop)
        }
        const unsigned epoch = R.epoch;
        const int par = epoch & 1;
        const unsigned tag = epoch + 1;
        const int pc = R.pc;
        if (f.test_abort_epoch >= 0 && (int)epoch == f.test_abort_epoch && b == f.G - 1) {  // tests: a workgroup gives up
            if (tid == 0) AG_STORE(f.abort_flag, 1u);
            R.end_code = 5;
            break;
        }
        RT_MARK(7);
        // ---- S: ratio test for column pc (simplex.ts:271-296) by the wave that holds the column: the ONE lane that holds it
        //         evaluates my rows' entries (what the pending pivot makes of them), readlanes hand entry i to lane i, lanes 0..ROWS-1
        //         classify their row in parallel (one division each), DPP reductions fold the verdicts -- no LDS round trip, no
        //         workgroup barrier inside -----------------------------------------------------------------------------------------
        if (wv == ((pc / CPT) >> 6)) {
            const int ol = __builtin_amdgcn_readfirstlane((pc / CPT) & 63);  // the lane that holds column pc
            const int jsel = __builtin_amdgcn_readfirstlane(pc % CPT);        // ... as its column jsel
            // lane i < ROWS receives row i's current entry, the pending pivot's row entry of that column and its non-zero flag,
            // and applies the pending pivot to it (what the bulk update will compute for that cell)
            double x = 0.0, pj = 0.0;
            unsigned nzj = 0;
#pragma unroll
            for (int j = 0; j < CPT; j++)
                if (jsel == j) {  // uniform
#pragma unroll
                    for (int i = 0; i < ROWS; i++) {
                        const double xi = readlane_f64(a[i][j], ol);
                        x = lane == i ? xi : x;
                    }
                    pj = readlane_f64(p[j], ol);
                    nzj = ((unsigned)__builtin_amdgcn_readlane((int)nzm, ol) >> j) & 1u;
                }
            const int r = r_begin + lane;
            double colv = 0.0;
            if (lane < ROWS) {
                colv = x;
                if (pend) {
                    const double ki = sm.colb[par_p][lane];
                    const double nvv = sm.nv[lane];
                    if (r == pr_p) colv = pj;
                    else if (r != 0 && r < r_end && nonzero16(ki)) colv = pc_p == pc ? nvv : (nzj ? eliminate(x, ki, pj) : x);
                }
                sm.colb[par][lane] = colv;  // every thread's bulk update of THIS pivot reads it (after the barrier below)
            }
            int kind = 0;  // 0 skip, 1 degenerate winner, 2 quotient candidate
            double quo = 0.0;
            if (lane < ROWS) {
                const double rhs = sm.rhsb[lane];
                if (r >= 1 && r < r_end && !(-precision < colv && colv < precision)) {
                    if (colv > 0 && precision > rhs && rhs > -precision) kind = 1;
                    else { quo = rhs / colv; kind = quo > precision ? 2 : 0; }
                }
            }
            int brdeg = kind == 1 ? r : 0x7fffffff;  // rows ascend with the lane: the smallest row is the first one
            brdeg = min(brdeg, __builtin_amdgcn_update_dpp(brdeg, brdeg, 0xB1, 0xf, 0xf, false));
            brdeg = min(brdeg, __builtin_amdgcn_update_dpp(brdeg, brdeg, 0x4E, 0xf, 0xf, false));
            brdeg = min(brdeg, __builtin_amdgcn_update_dpp(brdeg, brdeg, 0x141, 0xf, 0xf, false));
            brdeg = min(brdeg, __builtin_amdgcn_update_dpp(brdeg, brdeg, 0x140, 0xf, 0xf, false));
            brdeg = __builtin_amdgcn_readlane(brdeg, 0);  // (ROWS <= 16: the candidates sit in the first 16-lane row)
            KI bk;  // quotients are > precision > 0: positive doubles order like their bit patterns; ties -> first row
            bk.k = kind == 2 ? (u64_t)__double_as_longlong(quo) : KI_NONE_KEY;
            bk.i = kind == 2 ? r : 0x7fffffff;
            bk.pad = 0;
            bk = ki_min(bk, ki_dpp<0xB1>(bk));
            bk = ki_min(bk, ki_dpp<0x4E>(bk));
            bk = ki_min(bk, ki_dpp<0x141>(bk));
            bk = ki_min(bk, ki_dpp<0x140>(bk));
            bk = ki_readlane(bk, 0);
            if (lane < JSLP_G16_REPL) {
                const bool deg = brdeg != 0x7fffffff;
                const bool have = bk.k != KI_NONE_KEY;
                const int row = deg ? brdeg : (have ? bk.i : 0);  // the only row of mine that can win (0: none)
                const u64_t qb = (deg || !have) ? 0ull : bk.k;
                v4u_t g;
                g.x = (unsigned)qb;
                g.y = tag;
                g.z = (unsigned)(qb >> 32);
                g.w = ((tag & 0xffffu) << 16) | (deg ? 0x8000u : 0u) | (unsigned)row;
                __builtin_amdgcn_raw_buffer_store_b128(g, rsrc_g16, ((par * JSLP_G16_REPL + lane) * JSLP_F_MAXG + b) * JSLP_G16_STRIDE, 0, 16);  // aux 16 = sc1
                if (lane == 0) {
                    sm.pubrow = row;
                    if (JSLP_PIPE_PRUNE) {  // the XCD-local copy: a PLAIN store (stays in this XCD's L2), and my own summary for the pruning wave
                        __builtin_amdgcn_raw_buffer_store_b128(g, rsrc_g1, ((par * 8 + grp) * 32 + mem) * 16, 0, 0);
                        sm.myg[0] = g.x; sm.myg[1] = g.y; sm.myg[2] = g.z; sm.myg[3] = g.w;
                    }
                }
            }
        }
        __syncthreads();
        RT_MARK(0);
        // ---- P + U: the pending pivot's row update (simplex.ts:367-391), ONE pass over my rows; the row that can win is
        //         published (16-byte write-through stores) as soon as it is up to date.  The summaries are crossing the fabric
        //         meanwhile ----------------------------------------------------------------------------------------------------------
        const int pubrow = sm.pubrow;
        bool swept = true;
        const bool poller = tid < JSLP_F_MAXG;
        const bool used = tid < f.G;
        v4u_t g;
        g.x = 0; g.y = tag; g.z = 0; g.w = (tag & 0xffffu) << 16;  // lanes beyond the grid: "no candidate"
        const int goff = ((par * JSLP_G16_REPL + (b % JSLP_G16_REPL)) * JSLP_F_MAXG + tid) * JSLP_G16_STRIDE;
        if (JSLP_PIPE_EARLYPOLL && poller && used) g = __builtin_amdgcn_raw_buffer_load_b128(rsrc_g16, goff, 0, 16);
        if (JSLP_PIPE_PRUNE && wv == THREADS / 64 - 1) {
            // the last wave: is my candidate the best one among the workgroups of my XCD?  Lane l looks at member l of my group
            // (a few bounded polls of the XCD-local copies; what is not visible by then counts as "no candidate")
            int pubme = 0;
            if (pubrow != 0) {
                const int w = grp + 8 * lane;
                const bool other = lane < 32 && w < f.G && lane != mem;
                v4u_t h;
                h.x = sm.myg[0]; h.y = sm.myg[1]; h.z = sm.myg[2]; h.w = sm.myg[3];  // (my own lane and the unused ones: my summary)
                const int hoff = ((par * 8 + grp) * 32 + lane) * 16;
                bool ok = !other;
#pragma unroll 1
                for (int spin = 0; spin < 4; spin++) {
                    if (other && !ok) {
                        const v4u_t t = __builtin_amdgcn_raw_buffer_load_b128(rsrc_g1, hoff, 0, 16);
                        if (t.y == tag && (t.w >> 16) == (tag & 0xffffu)) { h = t; ok = true; }
                    }
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                const int row = (int)(h.w & 0x7fffu);
                const bool deg = (h.w & 0x8000u) != 0u;
                int rdeg = (deg && row != 0) ? row : 0x7fffffff;
                rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0xB1, 0xf, 0xf, false));
                rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0x4E, 0xf, 0xf, false));
                rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0x141, 0xf, 0xf, false));
                rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0x140, 0xf, 0xf, false));
                rdeg = min(min(__builtin_amdgcn_readlane(rdeg, 0), __builtin_amdgcn_readlane(rdeg, 16)),
                           min(__builtin_amdgcn_readlane(rdeg, 32), __builtin_amdgcn_readlane(rdeg, 48)));
                KI y;
                const bool cand = !deg && row != 0;
                y.k = cand ? ((u64_t)h.x | ((u64_t)h.z << 32)) : KI_NONE_KEY;
                y.i = cand ? row : 0x7fffffff;
                y.pad = 0;
                y = ki_wave_min(y);
                const int best = rdeg != 0x7fffffff ? rdeg : (y.k == KI_NONE_KEY ? 0 : y.i);
                pubme = best == pubrow ? 1 : 0;
            }
            if (lane == 0) sm.pubme = pubme;
        }
        double kis[ROWS];  // the pending pivot's column entries of my rows: ROWS broadcast reads in flight together, one wait
#pragma unroll
        for (int i = 0; i < ROWS; i++) kis[i] = sm.colb[par_p][i];
        if (JSLP_PIPE_PRUNE) {
            if (pend) {
#pragma unroll
                for (int i = 0; i < ROWS; i++) JSLP_PIPE_UPDATE_ROW(i);
            }
            __syncthreads();  // the pruning wave's verdict (the summaries are still crossing the fabric: this barrier is not on the critical path)
        }
        const bool publish = pubrow != 0 && (!JSLP_PIPE_PRUNE || sm.pubme != 0);
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            if (!JSLP_PIPE_PRUNE && pend) JSLP_PIPE_UPDATE_ROW(i);
            if (publish && r_begin + i == pubrow && colok) {  // (uniform but for colok)
                const int off = par * pub_stride + (b * ld + c0) * 8;
#pragma unroll
                for (int j = 0; j < CPT; j += 2) {
                    if (c0 + j >= ld) continue;
                    const u64_t lo = (u64_t)__double_as_longlong(a[i][j]), hi = (u64_t)__double_as_longlong(a[i][j + 1]);
                    v4u_t v;
                    v.x = (unsigned)lo; v.y = (unsigned)(lo >> 32); v.z = (unsigned)hi; v.w = (unsigned)(hi >> 32);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_rows, off + j * 8, 0, 16);
                }
            }
        }
        pend = false;
        RT_MARK(2);
        // ---- C: gather: lane w of the first four waves polls workgroup w's granule ------------------------------------------------
        if (poller) {
            unsigned spins = 0;
            bool first = JSLP_PIPE_EARLYPOLL != 0;
            for (;;) {
                if (!first && used) g = __builtin_amdgcn_raw_buffer_load_b128(rsrc_g16, goff, 0, 16);
                first = false;
                const bool ok = g.y == tag && (g.w >> 16) == (tag & 0xffffu);
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) { swept = false; break; }
                if (spins > f.spin_limit) { if (lane == 0) AG_STORE(f.abort_flag, 1u); swept = false; break; }
            }
            // my workgroup's summary -> the wave's: first degenerate row, else smallest quotient (first row on ties)
            const int row = (int)(g.w & 0x7fffu);
            const bool deg = (g.w & 0x8000u) != 0u;
            int rdeg = (deg && row != 0) ? row : 0x7fffffff;
            rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0xB1, 0xf, 0xf, false));
            rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0x4E, 0xf, 0xf, false));
            rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0x141, 0xf, 0xf, false));
            rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0x140, 0xf, 0xf, false));
            rdeg = min(min(__builtin_amdgcn_readlane(rdeg, 0), __builtin_amdgcn_readlane(rdeg, 16)),
                       min(__builtin_amdgcn_readlane(rdeg, 32), __builtin_amdgcn_readlane(rdeg, 48)));
            KI x;  // quotients are > precision > 0: positive doubles order like their bit patterns
            const bool cand = !deg && row != 0;
            x.k = cand ? ((u64_t)g.x | ((u64_t)g.z << 32)) : KI_NONE_KEY;
            x.i = cand ? row : 0x7fffffff;
            x.pad = 0;
            x = ki_wave_min(x);
            if (lane == 0) { sm.part_k[wv] = x.k; sm.part_r[wv] = x.k == KI_NONE_KEY ? 0 : x.i; sm.part_rdeg[wv] = rdeg; }
        }
        RT_MARK(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: my row stores are written through
        const int all_swept = __syncthreads_and(swept ? 1 : 0);
        if (!all_swept) { R.end_code = 5; break; }
        if (tid == 0 && publish) AG_STORE(f.rowflag[par] + b, (u64_t)tag);  // every wave has drained: the row is visible
        RT_MARK(3);
        // ---- D: every thread folds the four partial results ------------------------------------------------------------------
        int pr = 0, stop = 0;
        {
            u64_t wk = sm.part_k[0];
            int wr = sm.part_r[0], wrdeg = sm.part_rdeg[0];
#pragma unroll
            for (int i = 1; i < JSLP_F_MAXG / 64; i++) {
                const u64_t k2 = sm.part_k[i];
                const int r2 = sm.part_r[i], rd2 = sm.part_rdeg[i];
                const bool take = r2 != 0 && (wr == 0 || k2 < wk || (k2 == wk && r2 < wr));
                wk = take ? k2 : wk;
                wr = take ? r2 : wr;
                wrdeg = rd2 < wrdeg ? rd2 : wrdeg;
            }
            if (wrdeg != 0x7fffffff) pr = wrdeg;
            else if (wr != 0) pr = wr;
            else stop = 3;  // unbounded (simplex.ts:298-303)
        }
        if (!stop && c.check_cycles) {  // simplex.ts:305-320 by every workgroup, on its own LDS history
            if (tid == 0) {
                const int2 pair = make_int2(sm.lvibr[pr], sm.lvibc[pc]);
                sm.lhist[R.hist_n] = pair;
                if (b == 0) c.hist[R.hist_n] = pair;  // the host's cycle message; the general kernel's history should this one outgrow LDS
            }
            __syncthreads();
            R.hist_n += 1;
            if (suffix_is_square(sm.lhist, R.hist_n, sm.f.red)) stop = 1;
        }
        if (stop == 3) { R.end_code = 2; R.unbounded_col = pc; break; }
        if (stop == 1) { R.end_code = 3; break; }
        // ---- E: the winning row, loaded speculatively together with its flag (re-loaded in the rare case the flag was not up
        //         yet); the lane that holds column pc broadcasts quot = A[pr, pc] --------------------------------------------------
        const int bw = pr / f.rpb;
        const int off_in = par * pub_stride + (bw * ld + c0) * 8;
        const bool has_pc = colok && pc >= c0 && pc < c0 + CPT;
        double pv[CPT];
#pragma unroll
        for (int j = 0; j < CPT; j++) pv[j] = 0.0;
        double quot = 0.0;
        for (;;) {
            u64_t flag = 0;
            if (tid == 0) flag = AG_LOAD(f.rowflag[par] + bw);
            if (colok) {
#pragma unroll
                for (int j = 0; j < CPT; j += 2) {
                    if (c0 + j >= ld) continue;
                    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_rows, off_in + j * 8, 0, 16);
                    pv[j] = __longlong_as_double((long long)((u64_t)v.x | ((u64_t)v.y << 32)));
                    pv[j + 1] = __longlong_as_double((long long)((u64_t)v.z | ((u64_t)v.w << 32)));
                }
            }
            if (has_pc) {
#pragma unroll
                for (int j = 0; j < CPT; j++)
                    if (pc == c0 + j) sm.xq2[okslot] = pv[j];
            }
            if (tid == 0) {
                int ok = 1;
                if ((unsigned)flag != tag) {
                    unsigned spins = 0;
                    ok = 2;  // the row must be re-read once the flag is up
                    while ((unsigned)AG_LOAD(f.rowflag[par] + bw) != tag) {
                        __builtin_amdgcn_s_sleep(1);
                        ++spins;
                        if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) { ok = 0; break; }
                        if (spins > f.spin_limit) { AG_STORE(f.abort_flag, 1u); ok = 0; break; }
                    }
                }
                sm.okx[okslot] = ok;
                sm.p_batch = 0x7fffffff; sm.p_val = 0; sm.p_col = 0x7fffffff;  // pricing's reductions (reset before a barrier)
            }
            __syncthreads();
            const int okv = sm.okx[okslot];
            quot = sm.xq2[okslot];
            okslot ^= 1;  // the next use writes the other words: one barrier per use
            if (okv == 2) continue;
            if (okv == 0) R.end_code = 5;
            break;
        }
        if (R.end_code == 5) break;
        RT_MARK(4);
        // ---- N: normalised pivot row (simplex.ts:352-364; phase 2: some other row is always eliminated, so the tiny entries
        //         simplex.ts:381-383 zeroes are zero) -----------------------------------------------------------------------------
        nzm = 0;
        if (colok) {
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const int col = c0 + j;
                const double val = pv[j];
                double v = 0.0;
                if (col < W) {
                    const bool innz = nonzero16(val);
                    v = innz ? val / quot : 0.0;
                    if (col == pc) v = 1.0 / quot;
                    if (innz && !nonzero16(v) && v != 0.0) v = 0.0;
                }
                p[j] = v;
                nzm |= nonzero16(v) ? (1u << j) : 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < CPT; j++) p[j] = 0.0;
        }
        // the pivot column's own new entries (-k / quot, simplex.ts:386) and column 0 of my rows after this pivot (sm.rhsb is
        // the ratio test's copy of that column: it receives what the bulk update will give a[i][0]): lanes 0..ROWS-1 of wave 0
        if (wv == 0) {
            const double p0 = readlane_f64(p[0], 0);
            const unsigned nz0 = (unsigned)__builtin_amdgcn_readfirstlane((int)nzm) & 1u;
            if (lane < ROWS) {
                const int r = r_begin + lane;
                const double ki = sm.colb[par][lane];
                double v = sm.rhsb[lane];
                if (r == pr) v = p0;
                else if (r != 0 && r < r_end && nonzero16(ki) && nz0) v = eliminate(v, ki, p0);
                sm.rhsb[lane] = v;
                sm.nv[lane] = -ki / quot;
            }
        }
        // ---- R0: the cost row (every workgroup its own copy) -----------------------------------------------------------------------
        {
            const double k0 = R.k0;
            if (nonzero16(k0)) {
#pragma unroll
                for (int j = 0; j < CPT; j++)
                    if ((nzm >> j) & 1u) r0[j] = eliminate(r0[j], k0, p[j]);
                if (has_pc) {
                    const double nv0 = -k0 / quot;
#pragma unroll
                    for (int j = 0; j < CPT; j++)
                        if (pc == c0 + j) r0[j] = nv0;
                }
            }
        }
        // ---- commit the basis change (simplex.ts:339-349): every workgroup's LDS maps, workgroup 0 the global ones ---------------------
        if (tid == THREADS - 64) {  // (not thread 0: its wave carries the column-0 work above)
            const int leaving = sm.lvibr[pr], entering = sm.lvibc[pc];
            sm.lvibr[pr] = entering;
            sm.lvibc[pc] = leaving;
            if (b == 0) {
                c.vibr[pr] = entering;
                c.vibc[pc] = leaving;
                c.rbv[entering] = pr;
                c.rbv[leaving] = -1;
                c.cbv[entering] = -1;
                c.cbv[leaving] = pc;
                if (R.trace_n < c.trace_cap) c.trace[R.trace_n] = make_int2(pr, pc);
            }
        }
        R.trace_n += 1;
        R.it2 += 1;
        R.epoch = epoch + 1;
        pend = true; pr_p = pr; pc_p = pc; par_p = par;
        RT_MARK(5);
        // ---- G: price the new cost row -> entering column of the next pivot ----------------------------------------------------------
        {
            int neg_unused = 0;
            R.pc = price_row_lds<CPT, false>(r0, c0, pb, c, sm, &R.k0, 0u, &neg_unused);
        }
        RT_MARK(6);
        if (R.pc == 0) R.end_code = 1;  // optimal (simplex.ts:265-269)
    }
    if (pend && R.end_code != 5) {  // whoever leaves with a pivot pending (optimal, iteration cap, hand-over) brings the rows up to date
        const bool has_pc_p = colok && pc_p >= c0 && pc_p < c0 + CPT;
        double kis[ROWS];
#pragma unroll
        for (int i = 0; i < ROWS; i++) kis[i] = sm.colb[par_p][i];
#pragma unroll
        for (int i = 0; i < ROWS; i++) JSLP_PIPE_UPDATE_ROW(i);
    }
#undef JSLP_PIPE_UPDATE_ROW
}
