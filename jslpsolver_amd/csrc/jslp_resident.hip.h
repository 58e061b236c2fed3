// jslp_resident.hip.h -- the register-resident whole-solve kernel (one cooperative launch per simplex()).
// Included by jslp_kernels.hip.h after jslp_fused.hip.h (shares FCand / FSmem and the launch constants with it).
#pragma once

// ===================================================================================================
// Register-resident phase 2: the whole tableau lives in the VGPRs of the chip for the whole solve.
//
// 256 CUs x 512 KB of vector registers = 128 MB; a 2001 x 2016 fp64 tableau is 32 MB.  Workgroup w keeps its
// (<= 8) rows in registers -- lane pair (c0, c0+1) of each row -- together with a private copy of the cost row,
// so a pivot moves NO tableau bytes through HBM: per pivot each workgroup publishes 32 bytes of ratio-test
// summary plus the one row that would become the pivot row if it wins (16 KB), all workgroups meet at ONE
// grid barrier, read the <= 256 summaries and the winning row back, and update their registers.
// One cooperative launch runs the entire phase 2 (no host round trip, no kernel boundary per pivot).
//
// Inter-workgroup hand-off follows cdna_hip_programming.md Guideline 16: every shared word is written and read
// with 8-byte agent-scope relaxed atomics (sc1, write-through / L1-bypassing), every storing wave drains its
// stores (s_waitcnt vmcnt(0)) before the workgroup's leader arrives at the barrier counter, one lane polls with
// s_sleep, buffers alternate by pivot parity, every spin is bounded and raises a device-wide abort flag.
// Preconditions (host): those of the fused pipeline, plus H <= 8 * G (register residency) and a successful
// hipLaunchCooperativeKernel (all workgroups co-resident).
// ===================================================================================================
#define JSLP_R_ROWS 8       // rows per workgroup of the default geometry (H <= 8 * 256)
#define JSLP_R_MAXROWS 32   // ... of the XCD-local geometry (512 lanes x 2 columns x 32 rows on <= 32 workgroups of ONE XCD: H <= 1024); the tall chip-wide geometry keeps 16 (512 lanes x 4 columns x 16 rows: H <= 16 * 256)
#define JSLP_R_MAXH (16 * JSLP_F_MAXG)  // tallest tableau any geometry takes (the LDS copy of the row map)
#define JSLP_XL_MAXG 32      // workgroups of the XCD-local geometry: the CUs of one XCD
#define JSLP_XL_SPREAD 8     // ... launched as every 8th block of the grid (observed dispatch: block b -> XCD b % 8; the kernel checks it)
#define JSLP_R_GRAN 8        // 8-byte granules per workgroup summary (7 used)
#define JSLP_R_LDMAX 4096    // widest padded row any geometry takes (512 lanes x 8 columns)
#define JSLP_R_LUNR 8192     // variable indexes whose unrestricted flag fits the LDS copy (more: gather-by-leader protocol)
#define JSLP_R_LHIST 10240   // cycle-check history entries kept in LDS (80 KB)
// The lean kernel's cycle check (jslp_resident_pipe.hip.h): its LDS holds the first JSLP_PIPE_LHIST pairs of the history and, in the
// rest of the general build's 80 KB history plus 16 KB more, a 2^19-bit filter of the pairs seen (two hashes per pair); the whole
// history of every workgroup also goes to its own slice of a global buffer (JSLP_PIPE_GHIST pairs each), which the rare suffix
// test reads beyond the LDS part -- so a solve longer than the LDS history no longer hands over to the general build
#define JSLP_PIPE_LHIST 4096
#define JSLP_PIPE_CYCBITS (1 << 19)
#define JSLP_PIPE_GHIST (1 << 17)
#define JSLP_R_CYCEXTRA ((JSLP_PIPE_CYCBITS / 8 - (JSLP_R_LHIST - JSLP_PIPE_LHIST) * 8) / 4)  // words of the filter beyond the general history
// hand-off words of one engine (one allocation, zeroed per launch): [2][G][8] granules, [2][G] row flags, [2][G] chip-wide OR flags,
// 32 words of decision / verdict, then the lean kernel's [2][G] 16-byte summary granules (JSLP_G16_STRIDE bytes apart)
#define JSLP_R_SYNC_WORDS_GENERAL (2 * JSLP_F_MAXG * (JSLP_R_GRAN + 2) + 32)
#define JSLP_R_FLAGCOPIES 16  // copies of every row flag, one per fetching wave (2 KB apart: 4096 waves reading ONE word per pivot is a hot spot)
// round 6: the lean phase 2 of the 2- / 4-column geometries publishes its candidate row NORMALISED (jslp_resident_pipe.hip.h, NPUB); what used to be
// the 8-byte flag word per (parity, fetching wave, workgroup) is a 32-byte RECORD there: {checksum word, quot, -k0 / quot, 0}, behind the flag copies
#define JSLP_R_REC_BYTES 32
#define JSLP_R_REC_WORDS (2 * JSLP_R_FLAGCOPIES * JSLP_F_MAXG * (JSLP_R_REC_BYTES / 8))
#define JSLP_R_REC_OFF ((2 * JSLP_F_MAXG * 8 + 2 * JSLP_R_FLAGCOPIES * JSLP_F_MAXG) * 8)  // bytes from the first summary granule (ResCtx::gran16) to the first record
#define JSLP_R_SYNC_WORDS (JSLP_R_SYNC_WORDS_GENERAL + 2 * JSLP_F_MAXG * 8 + 2 * JSLP_R_FLAGCOPIES * JSLP_F_MAXG + JSLP_R_REC_WORDS + JSLP_F_MAXG)  // (+ the lean kernel's granules, 64 bytes apart; + the row-flag copies; + the row records; + the XCD-local build's placement census)
#ifndef JSLP_RES_FAST
#define JSLP_RES_FAST 1      // one barrier around the row flag (step E); the -k/quot entries of the pivot column computed by eight lanes in parallel while the winning row is in flight instead of one after the other by the lane that owns the column (step F)
#endif
#ifndef JSLP_RES_ALLGATHER
#define JSLP_RES_ALLGATHER 1  // no unrestricted variables, cycle check off (both phases): EVERY workgroup gathers the <= 256 summaries itself (coalesced: thread t polls granules t, t + blockDim, ...; the payloads meet in LDS) and takes the leader's decision redundantly -- the decision broadcast and its poll (one fabric hop) disappear
#endif
// Measured in round 2 and dropped (the code is in the history: commits 46ab798 ... 31f93eb), config 3a, 2001 x 2001:
//   * the row update of a pivot kept pending until the next summary is out (two placements): 96.0 k and 89.9 k against 102.5 k
//     pivots/s (the wait it was meant to fill is not idle enough);
//   * pricing by ONE wave over an LDS copy of the cost row: 93.2 k against 103.2 k pivots/s (the wave walks up to 40 batches
//     before it meets a candidate);
//   * pricing with one DPP (batch, key, index) reduction per wave and one barrier: 7.5 k against 3.8 k cycles per pivot (all 16
//     waves pay the wave stage and the 16-entry scan; the three LDS-atomic rounds keep 15 of them parked);
//   * pricing in two barriers (batch by ballot + LDS atomic, then a DPP (key, column) reduction in the one or two waves holding
//     the winning batch): 3.9-4.2 k against 3.2-3.6 k cycles;
//   * the whole ratio-test summary by the ONE lane that holds the pivot column (one barrier instead of two): no faster, its
//     eight divisions run one after the other;
//   * the pricing rounds of the next pivot interleaved with thirds of the row update: 120.1 k against 123.4 k pivots/s;
//   * the row flag raised before the gather instead of behind it: no faster (step E0 is one fabric round trip either way);
//   * 256-lane geometries (one wave per SIMD, 512 registers per lane): 72.5 k against 105.7 k pivots/s;
//   * no speculative candidate rows (leaderless protocol): the owner of the winning row publishes it after the decision as tagged
//     8-byte granules the others poll -- 0.77 MB instead of 4.19 MB of HBM traffic per pivot (PMC), but 50.4 k against 123 k
//     pivots/s: the narrow sc1 stores and 256 workgroups polling 2 x 2001 granules cost more than the hop they replace (and the
//     mere presence of the branch cost the default path 9 %: 112.8 k);
//   * DPP reductions in the LEADER's decision: wrong for phase 1, whose keys are negative (bits order positive doubles only).
#ifndef JSLP_RES_DPP_DECIDE
#define JSLP_RES_DPP_DECIDE 3  // all-gather protocol (phase 2: quotients > 0): bit 0 = (quotient, row) minimum, bit 1 = first degenerate row, by DPP exchanges + readlanes instead of ds_bpermute shuffles (120.9 k vs 119.2 k pivots/s, r02_x)
#endif
typedef unsigned long long u64_t;

struct ResCtx {
    Ctx c;
    u64_t* gran[2];       // [G][8] data-tagged granules {tag = epoch + 1 : 32 | payload : 32}: q, kq, kdeg halves, rows
    u64_t* rows_pub[2];   // [G][ld] candidate rows (doubles as 8-byte words)
    u64_t* rowflag[2];    // [G] epoch tag: the workgroup's candidate row of that epoch is fully written through
    u64_t* rowflagc[2];   // [copies][MAXG] the same flag, one copy per fetching wave (every wave looks at the flag itself: step E)
    unsigned* abort_flag; // set when any spin gives up
    u64_t* decision[2];   // leader's per-pivot decision: 3 tagged granules {pr | stop << 16}, {quot lo}, {quot hi}
    u64_t* verdict[2];    // phase 1 only: leader's cycle-check verdict {tag | stop} (the entering column is known late there)
    u64_t* gor[2];        // [G] rare slow path: tagged per-workgroup flags for a chip-wide OR
    int2* hist_all;       // [G][JSLP_PIPE_GHIST] every workgroup's own copy of the cycle-check history (lean kernel; nullptr: LDS part only)
    u64_t* gran16;        // [2][MAXG] 16-byte summary granules of the lean kernel's pipelined phase 2, 64 bytes apart (jslp_resident_pipe.hip.h)
    const Ctx* cdev;      // the same Ctx in device memory: what ONE thread needs once per pivot (global maps, trace, history) is read from there
    u64_t* census;        // [MAXG] XCD-local build: {0xA5A5A5A5 | XCC id} of every participating workgroup, written once per launch
    int32_t G, rpb, H;
    int32_t n_idx;             // variable indexes in use (the LDS copy of the unrestricted flags covers JSLP_R_LUNR of them)
    int32_t iters_cap;
    uint32_t spin_limit;       // bound of every poll loop (polls, not cycles); reaching it raises the device-wide abort flag
    int32_t test_abort_epoch;  // tests only (JSLP_TEST_RESIDENT_ABORT): the last workgroup aborts the hand-off at this pivot; -1 = never
    int32_t test_late_wave0;   // tests only (JSLP_TEST_RESIDENT_LATE_WAVE0): the wave holding thread 0 reaches every row fetch ~8 k cycles late
    u64_t* dbg;  // JSLP_DEBUG_RESIDENT builds only
};

// Test hooks of the register-resident kernels -- JSLP_TEST_RESIDENT_ABORT (a workgroup gives up at pivot k), JSLP_TEST_RESIDENT_LATE_WAVE0
// (a late wave at every row fetch; the chaos sleeps of -DJSLP_CHAOS_BUILD) and JSLP_SPIN_LIMIT -- are compiled into the TEST library only
// (libjslp_hip_chaos.so: -DJSLP_CHAOS_BUILD, which implies -DJSLP_TEST_HOOKS; tests/conftest.py `hip_hooks_lib`).  In the shipped
// library they are constants: three kernel arguments fewer to keep alive through the pivot loop (164 -> 151 SGPR spills in the
// headline instance) and no hook code in it: config 3a 147.8 k -> 154.3 k pivots/s (r04_t).
#if defined(JSLP_CHAOS_BUILD) && !defined(JSLP_TEST_HOOKS)
#define JSLP_TEST_HOOKS 1
#endif
#ifdef JSLP_TEST_HOOKS
#define F_TEST_ABORT f.test_abort_epoch
#define F_TEST_LATE f.test_late_wave0
#define F_SPIN f.spin_limit
#else
#define F_TEST_ABORT (-1)
#define F_TEST_LATE 0
#define F_SPIN JSLP_SPIN_LIMIT_DEFAULT
#endif
#define F_ITERS f.iters_cap  // (the iteration cap stays a run-time value: it is what ends a solve that cycles with the cycle check off)
#define AG_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define AG_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define JSLP_SPIN_LIMIT_DEFAULT (1u << 22)
#ifndef JSLP_POLL_SLEEP
#define JSLP_POLL_SLEEP 6
#endif

struct RSmem {
    FSmem f;
    u64_t w_q[4];        // leader: per sweep wave, bits of its smallest quotient / its row / its first degenerate row
    int32_t w_r[4], w_rdeg[4];
    // LDS-atomic reductions (a handful of participants each; far cheaper than 12 ds_bpermute stages)
    u64_t p_val;      // pricing: bits of the best value in the winning batch
    int32_t p_batch;  // pricing: first batch holding a candidate
    int32_t p_col;    // pricing: first column with that value
    u64_t l_q;        // leader: bits of the smallest accepted quotient
    double l_k;       // leader: pivot-column entry of the winning row
    int32_t l_rdeg, l_r;
    int32_t ok;
    unsigned okbad;      // step E: number (per workgroup, ever increasing) of the last row fetch in which some wave gave up waiting for the row flag
    double nv[JSLP_R_MAXROWS + 1];  // -k / quot of my rows' pivot-column entries (and of the cost row's), one lane each
    unsigned gsum[JSLP_F_MAXG * JSLP_R_GRAN];  // all-gather by every workgroup: the payloads of everybody's summary granules
    int32_t p_neg;    // pricing: isReducedCostNegative of the winning column (unrestricted variables only, simplex.ts:164-177)
    int32_t pubrow;
    unsigned dec[4];
    double xq[2];   // phase 1: quot and k0 broadcast by the lane pair owning column pc
    double quo[JSLP_R_MAXROWS];
    int32_t kind[JSLP_R_MAXROWS];
    double col[JSLP_R_MAXROWS];  // my rows' entries in the pivot column / in column 0
    double rhs[JSLP_R_MAXROWS];
    // pipelined phase 2 (jslp_resident_pipe.hip.h): pivot-column entries by pivot parity (the pending update still needs the
    // previous pivot's), column 0 of my rows, the four polling waves' partial decisions, quot broadcast with the row flag
    double colb[2][JSLP_R_MAXROWS];
    double rhsb[JSLP_R_MAXROWS];
    u64_t part_k[JSLP_F_MAXG / 64];
    int32_t part_r[JSLP_F_MAXG / 64], part_rdeg[JSLP_F_MAXG / 64];
    double xq2[2];
    double ook[2][4];  // the optional objectives' entries of the pivot column, broadcast with quot (OPT builds of the lean kernel)
    // All-gather protocol: what only workgroup 0 knows in the gather-by-leader protocol lives in EVERY workgroup's LDS -- the
    // row / column maps (swapped at every pivot like the global ones), the unrestricted flag of every variable index, and the
    // cycle-check history (the first JSLP_R_LHIST entries; a longer solve continues with the leader protocol, whose check
    // reads the global history workgroup 0 has been mirroring all along)
    int32_t lvibr[JSLP_R_MAXH];
    int32_t lvibc[JSLP_R_LDMAX];
    uint8_t lunr[JSLP_R_LUNR];
    int2 lhist[JSLP_R_LHIST];
    // lean kernel: bits of the (leaving, entering) pairs seen in this phase, by two hashes -- the filter starts inside lhist (behind
    // the lean kernel's JSLP_PIPE_LHIST pairs) and ends here.  A repeated block can only end at a pair that occurred before, so a
    // pair with a clear bit needs no suffix test at all (exact: bits are only ever set; a collision merely runs the test)
    unsigned cycbits_tail[JSLP_R_CYCEXTRA];
    int32_t cyc_need, cyc_filter_on;
    // lean pipelined phase 2 (price_row_pipe): the pricing's reduction words, double-buffered by pivot parity (reset one pivot ahead, a barrier away
    // from every use) -- first batch holding a candidate, best value in it, first column with that value, isReducedCostNegative of the winner
    int32_t pw_batch[2], pw_col[2], pw_neg[2];
    u64_t pw_val[2];
    int32_t cm_ent, cm_leav;  // lean pipelined loops, workgroup 0: the basis change whose GLOBAL commit is still pending (written and read by the one committing thread)
    // ... and what that thread needs to issue the commit as seven fire-and-forget stores (no trip to the device copy of the context in front of them)
    int32_t* gp_vibr; int32_t* gp_vibc; int32_t* gp_rbv; int32_t* gp_cbv; int2* gp_trace; long long gp_trace_cap;
    int2* gp_hist;  // ... and workgroup 0's global copy of the cycle-check history (thread 0: one store per pivot, no trip through the device copy of the context)
};

// The leader's last FOUR waves gather every workgroup's summary of this epoch (all-gather with the data as the flag,
// Guideline 16 R2): lane w of the 256 owns workgroup w and re-reads its 7 granules -- all in flight per pass -- until
// every tag matches, so the complete summary ends up in that lane's registers.  Returns false on abort (per wave).
#define JSLP_SWEEP_LANES 256
struct SweptCand {
    u64_t qbits;   // bits of the smallest accepted quotient (positive doubles order like their bits); ~0 when none
    double kq, kdeg;
    int32_t r, rdeg;
};
__device__ __forceinline__ bool sweep_summary(const ResCtx& f, int par, unsigned tag, int w, SweptCand& out) {
    const bool used = w < f.G;
    const u64_t* g = f.gran[par] + (long long)(used ? w : 0) * JSLP_R_GRAN;
    unsigned spins = 0;
    u64_t x[JSLP_R_GRAN - 1];
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < JSLP_R_GRAN - 1; j++) x[j] = used ? AG_LOAD(g + j) : ((u64_t)tag << 32);
#pragma unroll
        for (int j = 0; j < JSLP_R_GRAN - 1; j++) ok = ok && (unsigned)(x[j] >> 32) == tag;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) return false;
        if (spins > F_SPIN) { if ((w & 63) == 0) AG_STORE(f.abort_flag, 1u); return false; }
    }
    const u64_t qb = (x[0] & 0xffffffffull) | (x[1] << 32);
    const u64_t kqb = (x[2] & 0xffffffffull) | (x[3] << 32);
    const u64_t kdb = (x[4] & 0xffffffffull) | (x[5] << 32);
    const unsigned rr = (unsigned)x[6];
    out.r = used ? (int32_t)(rr & 0xffffu) : 0;
    const unsigned rd = rr >> 16;
    out.rdeg = (!used || rd == 0xffffu) ? 0x7fffffff : (int32_t)rd;
    out.qbits = out.r != 0 ? qb : ~0ull;
    out.kq = __longlong_as_double((long long)kqb);
    out.kdeg = __longlong_as_double((long long)kdb);
    return true;
}

// maximum of a 64-bit key over the wave, result in every lane (four DPP exchanges inside the 16-lane rows + readlanes across them)
template <int CTRL>
__device__ __forceinline__ u64_t u64_dpp(u64_t x) {
    const int lo = (int)(unsigned)x, hi = (int)(unsigned)(x >> 32);
    return ((u64_t)(unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false) << 32) |
           (u64_t)(unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ u64_t u64_readlane(u64_t x, int l) {
    return ((u64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), l) << 32) |
           (u64_t)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, l);
}
__device__ __forceinline__ u64_t u64_max(u64_t a, u64_t b) { return a > b ? a : b; }
__device__ __forceinline__ u64_t u64_wave_max(u64_t x) {
    x = u64_max(x, u64_dpp<0xB1>(x));
    x = u64_max(x, u64_dpp<0x4E>(x));
    x = u64_max(x, u64_dpp<0x141>(x));
    x = u64_max(x, u64_dpp<0x140>(x));
    return u64_max(u64_max(u64_readlane(x, 0), u64_readlane(x, 16)), u64_max(u64_readlane(x, 32), u64_readlane(x, 48)));
}
// (Round 4 tried two register-level forms of this reduction on config 3a, both bit-identical, both slower than the three rounds of LDS
//  atomics below -- 168.7 k pivots/s: (a) every wave folds batch, value and column with a DPP maximum and ballots, 16 bytes per wave, ONE
//  barrier, every thread scans the 16 results: 139.7 k -- with 16 waves on 4 SIMDs, work replicated in every wave costs four times its
//  instruction count; (b) round 1 as below, then only the one to five waves holding lanes of the winning batch fold value and column in
//  registers, second barrier, every thread reads those results: 158.8 k, phase 1 -- which does not price -- slower with it (14 more
//  spilled SGPRs in the loop).  The atomics' participants are few: a batch is 50-500 columns.)
// Pricing (simplex.ts:118-219, no unrestricted variables) of the cost-row pair (columns c0, c0+1) each lane
// holds, reduced with three LDS atomics: first batch holding a candidate, best value in it, first column with
// that value.  Positive doubles order like their bit patterns.  Returns the column (0 = none) and its value.
// `sm.p_*` must have been reset (p_batch = INT_MAX, p_val = 0, p_col = INT_MAX) before a preceding barrier.
// UNR: columns whose variable is unrestricted (bit j of `unr`) price with |rc| and hand isReducedCostNegative to the ratio
// test (simplex.ts:164-177, 282); *value receives the signed reduced cost of the winner, *neg the flag.
template <int CPT, bool UNR>
__device__ __forceinline__ int price_row_lds(const double (&x)[CPT], int c0, const int (&pb)[CPT], const Ctx& c, RSmem& sm,
                                             double* value, unsigned unr, int* neg) {
    double bv = c.precision;
    int bi = 0, bb = 0;
#pragma unroll
    for (int j = 0; j < CPT; j++) {  // my columns in order: earlier batch first, bigger value inside a batch, first index on ties
        const int col = c0 + j;
        const double val = (UNR && ((unr >> j) & 1u) && x[j] < 0) ? -x[j] : x[j];
        const bool ok = col >= 1 && col < c.W && val > c.precision;
        const bool take = ok && (bi == 0 || pb[j] < bb || (pb[j] == bb && val > bv));
        bv = take ? val : bv;
        bi = take ? col : bi;
        bb = take ? pb[j] : bb;
    }
    {   // batch ids grow with the lane index: the wave's earliest batch is that of its first candidate lane
        const unsigned long long m = __ballot(bi != 0);
        if (m != 0ull) {
            const int first = __ffsll((long long)m) - 1;
            const int wave_b = __builtin_amdgcn_readlane(bb, first);
            if ((threadIdx.x & 63) == 0) atomicMin(&sm.p_batch, wave_b);
        }
    }
    __syncthreads();
    const int wb = sm.p_batch;
    if (wb == 0x7fffffff) return 0;  // uniform: no candidate anywhere -> optimal
    const u64_t bits = (u64_t)__double_as_longlong(bv);
    if (bi != 0 && bb == wb) atomicMax(&sm.p_val, bits);
    __syncthreads();
    const u64_t wv = sm.p_val;
    if (bi != 0 && bb == wb && bits == wv) atomicMin(&sm.p_col, bi);
    __syncthreads();
    const int pcol = sm.p_col;
    double v = __longlong_as_double((long long)wv);
    if (UNR) {  // the lane holding the winner knows the sign of its reduced cost
#pragma unroll
        for (int j = 0; j < CPT; j++)
            if (pcol == c0 + j) sm.p_neg = (((unr >> j) & 1u) && x[j] < 0) ? 1 : 0;
        __syncthreads();
        *neg = sm.p_neg;
        if (*neg) v = -v;
    }
    *value = v;
    return pcol;
}

#ifdef JSLP_DEBUG_RESIDENT
#define RT_MARK(i) do { const u64_t _now = __builtin_amdgcn_s_memtime(); rt_acc[i] += _now - rt_prev; rt_prev = _now; } while (0)
// round 6: event stamps of EIGHT consecutive pivots (epochs JSLP_STAMP_E0 ...) from thread 0 of EVERY workgroup, on the constant-rate clock all
// CUs share (s_memrealtime, 100 MHz): who publishes its summary last, when the gather closes, when the row is in -- tools/resident_stamps.py
#ifndef JSLP_STAMP_E0
#define JSLP_STAMP_E0 4096u
#endif
#define RT_STAMP(k) do { if (tid == 0 && f.dbg && R.epoch - JSLP_STAMP_E0 < 8u) f.dbg[(((R.epoch - JSLP_STAMP_E0) * JSLP_F_MAXG) + b) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RT_MARK(i) do { } while (0)
#define RT_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ void reset_reductions(RSmem& sm) {  // one thread, before a barrier
    sm.p_batch = 0x7fffffff; sm.p_val = 0; sm.p_col = 0x7fffffff;
    sm.l_rdeg = 0x7fffffff; sm.l_q = ~0ull; sm.l_r = 0x7fffffff;
}

// Chip-wide OR of one flag per workgroup (rare slow path of phase 1, see the lazily-zeroed pivot-row entries):
// every workgroup publishes a tagged granule and polls everybody else's.  Returns -1 on abort.
__device__ __forceinline__ int global_or(const ResCtx& f, int par, unsigned tag, int flag, RSmem& sm, int b) {
    const int tid = threadIdx.x;
    if (tid == 0) AG_STORE(f.gor[par] + b, ((u64_t)tag << 32) | (unsigned)(flag ? 1 : 0));
    int mine = 0, ok = 1;
    if (tid < f.G) {
        unsigned spins = 0;
        for (;;) {
            const u64_t x = AG_LOAD(f.gor[par] + tid);
            if ((unsigned)(x >> 32) == tag) { mine = (int)(x & 1u); break; }
            __builtin_amdgcn_s_sleep(JSLP_POLL_SLEEP);
            ++spins;
            if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) { ok = 0; break; }
            if (spins > F_SPIN) { AG_STORE(f.abort_flag, 1u); ok = 0; break; }
        }
    }
    const int bad = __syncthreads_or(ok ? 0 : 1);
    const int any = __syncthreads_or(mine);
    return bad ? -1 : (any ? 1 : 0);
}

// Loop-carried state of the resident kernel (kept in registers: every member is a scalar or a fully unrolled array)
template <int CPT, int ROWS>
struct ResRegs {
    double a[ROWS][CPT];  // my rows: CPT adjacent columns per lane
    double r0[CPT];              // my copy of the cost row
    double oo[3][CPT];           // my copies of the optional objective rows (lean OPT builds; JSLP_R_MAXOPT)
    double k0;
    unsigned unr;  // bit j: the variable of my column j is unrestricted (UNR builds)
    int neg;       // isReducedCostNegative of the entering column (phase 2, UNR builds)
    int pc, end_code, unbounded_col, hist_n, it1, it2;
    unsigned epoch;
    unsigned efetch;  // row fetches so far (step E: see sm.okbad)
    long long trace_n;
#ifdef JSLP_DEBUG_RESIDENT
    u64_t rt_retries;  // row fetches that found the winner's flag not up yet (lean kernel, step E)
    u64_t rt_acc[8];
    u64_t rt_prev;
#endif
};

// One phase of the solve.  PHASE is a compile-time constant so that the phase-2 loop -- the hot one -- carries none of
// the phase-1 branches; returns when the solve ends (R.end_code != 0) or, for PHASE == 1, when phase 1 is over
// (end_code stays 0 and the caller starts phase 2).
// LEAN (k_simplex_resident<.., LEAN = true>): the all-gather protocol only -- the gather-by-leader branches, which the general
// kernel needs for unrestricted variables beyond the LDS copy and for histories that outgrow LDS, are compiled out; a solve
// that comes to need them leaves with end_code 8 and the host continues it with the general kernel.
template <int PHASE, int THREADS, int CPT, int ROWS, bool UNR, bool LEAN = false>
__device__ __forceinline__ void resident_phase(const ResCtx& f, RSmem& sm, ResRegs<CPT, ROWS>& R, int it1_start, int it2_start,
                                               const int (&pb)[CPT]) {
    const Ctx& c = f.c;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int ld = c.ld, W = c.W, H = f.H;
    const double precision = c.precision;
    const int c0 = tid * CPT;
    const bool colok = c0 < ld;
    const int r_begin = b * f.rpb, r_end = min(H, r_begin + f.rpb);
    const int sweep0 = (int)blockDim.x - JSLP_SWEEP_LANES;  // the leader's last four waves sweep
    constexpr int phase = PHASE;
    double (&a)[ROWS][CPT] = R.a;
    double (&r0)[CPT] = R.r0;
    double& k0 = R.k0;
    int& pc = R.pc;
    int& end_code = R.end_code;
    int& unbounded_col = R.unbounded_col;
    int& hist_n = R.hist_n;
    int& it1 = R.it1;
    int& it2 = R.it2;
    unsigned& epoch = R.epoch;
    long long& trace_n = R.trace_n;
#ifdef JSLP_DEBUG_RESIDENT
    u64_t (&rt_acc)[8] = R.rt_acc;
    u64_t& rt_prev = R.rt_prev;
#endif
    (void)H;
    // candidate rows travel as 16-byte write-through (sc1) buffer stores / loads: a lane holds its columns as adjacent pairs,
    // and the same bytes issued as 8-byte agent atomics cost one fabric write each (cdna_hip_programming.md, Guideline 16 R1
    // and pitfall 7: narrow sc1 stores are 2.7x the time per byte) -- half the stores, half the counted write traffic
    typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
    const int pub_bytes = f.G * ld * 8;
    const auto rsrc0 = __builtin_amdgcn_make_buffer_rsrc(f.rows_pub[0], 0, pub_bytes, 0x00020000);
    const auto rsrc1 = __builtin_amdgcn_make_buffer_rsrc(f.rows_pub[1], 0, pub_bytes, 0x00020000);
    unsigned& efetch = R.efetch;  // row fetches of this workgroup so far (uniform; one count for both phases: sm.okbad only grows)
    while (end_code == 0) {
        if ((it1 - it1_start) + (it2 - it2_start) >= F_ITERS) { end_code = 4; break; }
        if (LEAN && c.check_cycles && !(hist_n < JSLP_R_LHIST && hist_n < c.hist_cap)) { end_code = 8; break; }  // history outgrows LDS
        const int par = epoch & 1;
        const unsigned tag = epoch + 1;
        if (F_TEST_ABORT >= 0 && (int)epoch == F_TEST_ABORT && b == f.G - 1) {  // tests: a workgroup gives up
            if (tid == 0) AG_STORE(f.abort_flag, 1u);
            end_code = 5;
            break;
        }
        RT_MARK(7);
        // ---- A: my rows' summary: phase 2 = ratio test for column pc (simplex.ts:276-296); phase 1 = most negative RHS
        //         below -precision (simplex.ts:39-49) -------------------------------------------------------------------
        bool has_pc = phase == 2 && colok && pc >= c0 && pc < c0 + CPT;
        double* colnow = sm.col;
        if (has_pc) {  // (conditional stores, not selects among register-array elements: those end up in scratch)
#pragma unroll
            for (int j = 0; j < CPT; j++)
                if (pc == c0 + j) {
#pragma unroll
                    for (int i = 0; i < ROWS; i++) colnow[i] = a[i][j];
                }
        }
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < ROWS; i++) sm.rhs[i] = a[i][0];
            reset_reductions(sm);
        }
        __syncthreads();
        if (tid < 64) {
            // lanes 0..7 classify one row each (the division runs in parallel), then every lane of wave 0 merges the
            // eight verdicts in row order: same result in all lanes, no shuffles
            if (tid < ROWS) {
                const int r = r_begin + tid;
                const double colv = colnow[tid], rhs = sm.rhs[tid];
                int kind = 0;  // 0 skip, 1 degenerate winner, 2 quotient candidate (phase 1: RHS candidate)
                double quo = 0.0;
                if (phase == 1) {
                    if (r >= 1 && r < r_end && rhs < -precision) { quo = rhs; kind = 2; }
                } else if (r >= 1 && r < r_end && !(-precision < colv && colv < precision)) {
                    if (colv > 0 && precision > rhs && rhs > -precision) kind = 1;
                    else { quo = (UNR && R.neg) ? -rhs / colv : rhs / colv; kind = quo > precision ? 2 : 0; }
                }
                sm.quo[tid] = quo;
                sm.kind[tid] = kind;
            }
            // (wave 0 only: LDS writes above are visible to the same wave after the wave-level sync below)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            FCand mine = fcand_none();
#pragma unroll
            for (int i = 0; i < ROWS; i++) {
                const int kind = sm.kind[i];
                const double quo = sm.quo[i], colv = colnow[i];
                const int r = r_begin + i;
                if (kind == 1) { if (r < mine.rdeg) { mine.rdeg = r; mine.kdeg = colv; } }
                else if (kind == 2 && mine.q > quo) { mine.q = quo; mine.r = r; mine.kq = colv; }
            }
            if (tid < JSLP_R_GRAN - 1) {  // lanes 0..6 publish one tagged granule each
                const u64_t qb = (u64_t)__double_as_longlong(mine.q), kqb = (u64_t)__double_as_longlong(mine.kq),
                            kdb = (u64_t)__double_as_longlong(mine.kdeg);
                unsigned payload;
                switch (tid) {
                    case 0: payload = (unsigned)qb; break;
                    case 1: payload = (unsigned)(qb >> 32); break;
                    case 2: payload = (unsigned)kqb; break;
                    case 3: payload = (unsigned)(kqb >> 32); break;
                    case 4: payload = (unsigned)kdb; break;
                    case 5: payload = (unsigned)(kdb >> 32); break;
                    default: payload = (unsigned)mine.r | ((mine.rdeg == 0x7fffffff ? 0xffffu : (unsigned)mine.rdeg) << 16); break;
                }
                AG_STORE(f.gran[par] + (long long)b * JSLP_R_GRAN + tid, ((u64_t)tag << 32) | payload);
            }
            if (tid == 0) sm.pubrow = mine.rdeg != 0x7fffffff ? mine.rdeg : mine.r;  // the only row of mine that can win
        }
        __syncthreads();
        RT_MARK(0);
        // ---- B: publish that row (write-through 8-byte agent stores); its own flag follows the drain ---------------
        const int pubrow = sm.pubrow;
        if (pubrow != 0 && colok) {
            const int off = (b * ld + c0) * 8;
#pragma unroll
            for (int i = 0; i < ROWS; i++)
                if (r_begin + i == pubrow) {  // uniform
#pragma unroll
                    for (int j = 0; j < CPT; j += 2) {
                        if (c0 + j >= ld) continue;  // (a lane's last pairs may lie beyond the row when CPT does not divide 16)
                        const u64_t lo = (u64_t)__double_as_longlong(a[i][j]), hi = (u64_t)__double_as_longlong(a[i][j + 1]);
                        v4u_t v;
                        v.x = (unsigned)lo; v.y = (unsigned)(lo >> 32); v.z = (unsigned)hi; v.w = (unsigned)(hi >> 32);
                        if (par) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc1, off + j * 8, 0, 16);  // aux 16 = sc1
                        else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc0, off + j * 8, 0, 16);
                    }
                }
        }
        RT_MARK(1);
        // ---- C: workgroup 0 is the LEADER: its last four waves gather everybody's tagged summaries (data = flag)
        //         while all other waves, everywhere, drain their row stores -----------------------------------------------
        bool swept = true;
        SweptCand sc;
        sc.qbits = ~0ull; sc.kq = 0; sc.kdeg = 0; sc.r = 0; sc.rdeg = 0x7fffffff;
        // all-gather by every workgroup (JSLP_RES_ALLGATHER): no leader, no decision broadcast
        const bool allg = LEAN || (JSLP_RES_ALLGATHER != 0 && (!UNR || f.n_idx <= JSLP_R_LUNR) && (!c.check_cycles || (hist_n < JSLP_R_LHIST && hist_n < c.hist_cap)));
        const bool sweeper = b == 0 && !allg && tid >= sweep0;
        if (allg) {
            // thread t polls granules t, t + blockDim, ... of the [G][8] array (adjacent lanes, adjacent granules: 64-byte requests)
            // until every tag matches; the payloads go to LDS, where lane w of the last four waves picks workgroup w's seven up
            const int NG = f.G * JSLP_R_GRAN;
            constexpr int NQ = (JSLP_F_MAXG * JSLP_R_GRAN + THREADS - 1) / THREADS;  // granules per thread: 2 at 1024 lanes, 4 at 512
            u64_t x[NQ];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const int g = tid + q * (int)blockDim.x;
                    const bool used = g < NG && (g & (JSLP_R_GRAN - 1)) != JSLP_R_GRAN - 1;
                    x[q] = used ? AG_LOAD(f.gran[par] + g) : ((u64_t)tag << 32);
                    ok = ok && (unsigned)(x[q] >> 32) == tag;
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) { swept = false; break; }
                if (spins > F_SPIN) { if ((tid & 63) == 0) AG_STORE(f.abort_flag, 1u); swept = false; break; }
            }
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const int g = tid + q * (int)blockDim.x;
                if (g < JSLP_F_MAXG * JSLP_R_GRAN) sm.gsum[g] = (unsigned)x[q];
            }
        } else if (sweeper) swept = sweep_summary(f, par, tag, tid - sweep0, sc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: my stores of the row have reached the L2
        const int all_swept = __syncthreads_and(swept ? 1 : 0);
        if (!all_swept) { end_code = 5; break; }
        if (tid < THREADS / 64 && pubrow != 0) {
            // ... and memory: the acknowledgement of a write-through store only means the XCD's L2 has it (jslp_resident_pipe.hip.h,
            // "the winner releases its row"); this build decides later, so every publishing workgroup pays one buffer_wbl2 here
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            AG_STORE(f.rowflagc[par] + tid * JSLP_F_MAXG + b, (u64_t)tag);  // one copy of the flag per fetching wave
        }
        RT_MARK(2);
        // ---- D: the leader decides (winner, unboundedness, cycle check) and broadcasts three tagged granules -----------
        int pr = 0, stop = 0;
        double quot = 0.0;
        if (allg) {
            // every workgroup decides for itself, out of the LDS copy of everybody's summary: ONE wave (lane l folds workgroups
            // l, l + 64, l + 128, l + 192, then six shuffle stages on the keys), one barrier, and the winner's pivot-column
            // entry is read back from the same LDS copy by everybody
            if (tid >= (int)blockDim.x - 64) {
                const int l = tid & 63;
                u64_t q = ~0ull;
                int r = 0, rdeg = 0x7fffffff;
#pragma unroll
                for (int k4 = 0; k4 < JSLP_F_MAXG / 64; k4++) {
                    const int wg = l + 64 * k4;
                    if (wg < f.G) {
                        const unsigned* gq = sm.gsum + wg * JSLP_R_GRAN;
                        const unsigned rr = gq[6];
                        const int r2 = (int)(rr & 0xffffu);
                        const unsigned rd = rr >> 16;
                        const int rd2 = rd == 0xffffu ? 0x7fffffff : (int)rd;
                        const u64_t q2 = r2 != 0 ? ((u64_t)gq[0] | ((u64_t)gq[1] << 32)) : ~0ull;
                        const double d2 = __longlong_as_double((long long)q2), d1 = __longlong_as_double((long long)q);
                        const bool take = r2 != 0 && (r == 0 || d2 < d1 || (d2 == d1 && r2 < r));
                        q = take ? q2 : q;
                        r = take ? r2 : r;
                        rdeg = rd2 < rdeg ? rd2 : rdeg;
                    }
                }
                if (JSLP_RES_DPP_DECIDE) {
                    // DPP exchanges inside the 16-lane rows + readlanes across them instead of 24 ds_bpermute round trips
                    // (positive doubles order like their bits; ~0 = no candidate)
                    if (JSLP_RES_DPP_DECIDE & 1) {
                        KI x;  // (key_asc: the unsigned order of the keys is the numeric order -- phase 1's candidates are negative)
                        x.k = r != 0 ? key_asc(__longlong_as_double((long long)q)) : KI_NONE_KEY; x.i = r != 0 ? r : 0x7fffffff; x.pad = 0;
                        x = ki_wave_min(x);
                        r = x.k == KI_NONE_KEY ? 0 : x.i;
                    } else {
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) {
                            const u64_t q2 = __shfl_xor(q, off, 64);
                            const int r2 = __shfl_xor(r, off, 64);
                            const double d2 = __longlong_as_double((long long)q2), d1 = __longlong_as_double((long long)q);
                            const bool take = r2 != 0 && (r == 0 || d2 < d1 || (d2 == d1 && r2 < r));
                            q = take ? q2 : q;
                            r = take ? r2 : r;
                        }
                    }
                    if (JSLP_RES_DPP_DECIDE & 2) {
                        rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0xB1, 0xf, 0xf, false));
                        rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0x4E, 0xf, 0xf, false));
                        rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0x141, 0xf, 0xf, false));
                        rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0x140, 0xf, 0xf, false));
                        rdeg = min(min(__builtin_amdgcn_readlane(rdeg, 0), __builtin_amdgcn_readlane(rdeg, 16)),
                                   min(__builtin_amdgcn_readlane(rdeg, 32), __builtin_amdgcn_readlane(rdeg, 48)));
                    } else {
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) { const int rd2 = __shfl_xor(rdeg, off, 64); rdeg = rd2 < rdeg ? rd2 : rdeg; }
                    }
                } else {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const u64_t q2 = __shfl_xor(q, off, 64);
                    const int r2 = __shfl_xor(r, off, 64), rd2 = __shfl_xor(rdeg, off, 64);
                    const double d2 = __longlong_as_double((long long)q2), d1 = __longlong_as_double((long long)q);
                    const bool take = r2 != 0 && (r == 0 || d2 < d1 || (d2 == d1 && r2 < r));
                    q = take ? q2 : q;
                    r = take ? r2 : r;
                    rdeg = rd2 < rdeg ? rd2 : rdeg;
                }
                }
                if (l == 0) { sm.w_r[0] = r; sm.w_rdeg[0] = rdeg; }
            }
            __syncthreads();
            const int wr = sm.w_r[0], wrdeg = sm.w_rdeg[0];
            if (wrdeg != 0x7fffffff) pr = wrdeg;
            else if (wr != 0) pr = wr;
            else stop = phase == 1 ? 4 : 3;  // phase 1: no violated row -> feasible (:51-54); phase 2: unbounded (:298-303)
            if (!stop) {  // the owner of row pr published both of its entries (phase 1 finds quot later, with the entering column)
                const unsigned* gq = sm.gsum + (pr / f.rpb) * JSLP_R_GRAN;
                const u64_t kb = wrdeg != 0x7fffffff ? ((u64_t)gq[4] | ((u64_t)gq[5] << 32)) : ((u64_t)gq[2] | ((u64_t)gq[3] << 32));
                quot = __longlong_as_double((long long)kb);
            }
            if (!stop && phase == 2 && c.check_cycles) {  // simplex.ts:305-320 by every workgroup, on its own LDS history
                if (tid == 0) {
                    const int2 pair = make_int2(sm.lvibr[pr], sm.lvibc[pc]);
                    sm.lhist[hist_n] = pair;
                    if (b == 0) c.hist[hist_n] = pair;  // the host's cycle message, and the leader protocol's history should this one outgrow LDS
                }
                __syncthreads();
                hist_n += 1;
                if (suffix_is_square(sm.lhist, hist_n, sm.f.red)) stop = 1;
            }
            if (UNR && !stop && tid == 0) sm.dec[0] = sm.lunr[sm.lvibr[pr]] ? (1u << 24) : 0u;  // "the leaving variable is unrestricted"
            if (UNR) __syncthreads();
        } else if (b == 0) {
            // (min rdeg) else (min q, then min r): each sweep wave reduces its 64 summaries with shuffles on the keys
            // only, the four wave results meet in LDS, the lane that holds the winner supplies its pivot-column entry
            if (sweeper) {
                u64_t q = sc.qbits;
                int r = sc.r, rdeg = sc.rdeg;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const u64_t q2 = __shfl_xor(q, off, 64);
                    const int r2 = __shfl_xor(r, off, 64), rd2 = __shfl_xor(rdeg, off, 64);
                    const double d2 = __longlong_as_double((long long)q2), d1 = __longlong_as_double((long long)q);
                    const bool take = r2 != 0 && (r == 0 || d2 < d1 || (d2 == d1 && r2 < r));  // smallest value, first row
                    q = take ? q2 : q;
                    r = take ? r2 : r;
                    rdeg = rd2 < rdeg ? rd2 : rdeg;
                }
                if ((tid & 63) == 0) {
                    const int wv = (tid - sweep0) >> 6;
                    sm.w_q[wv] = q; sm.w_r[wv] = r; sm.w_rdeg[wv] = rdeg;
                }
            }
            __syncthreads();
            u64_t wq = ~0ull;
            int wr = 0, wrdeg = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const u64_t q2 = sm.w_q[i];
                const int r2 = sm.w_r[i], rd2 = sm.w_rdeg[i];
                const double d2 = __longlong_as_double((long long)q2), d1 = __longlong_as_double((long long)wq);
                const bool take = r2 != 0 && (wr == 0 || d2 < d1 || (d2 == d1 && r2 < wr));
                wq = take ? q2 : wq;
                wr = take ? r2 : wr;
                wrdeg = rd2 < wrdeg ? rd2 : wrdeg;
            }
            if (wrdeg != 0x7fffffff) pr = wrdeg;
            else if (wr != 0) pr = wr;
            else stop = phase == 1 ? 4 : 3;  // phase 1: no violated row -> feasible (:51-54); phase 2: unbounded (:298-303)
            if (!stop && sweeper && tid - sweep0 == pr / f.rpb)
                sm.l_k = wrdeg != 0x7fffffff ? sc.kdeg : sc.kq;  // the owner of row pr published both entries
            __syncthreads();
            quot = stop ? 0.0 : sm.l_k;
            if (!stop && phase == 2 && c.check_cycles) {  // simplex.ts:305-320, before anything is committed
                if (hist_n >= c.hist_cap) {
                    stop = 2;
                } else {
                    if (tid == 0) c.hist[hist_n] = make_int2(c.vibr[pr], c.vibc[pc]);
                    __syncthreads();
                    hist_n += 1;
                    if (suffix_is_square(c.hist, hist_n, sm.f.red)) stop = 1;
                }
            }
            if (tid < 3) {
                const u64_t qb = (u64_t)__double_as_longlong(quot);
                unsigned lead = (unsigned)pr | ((unsigned)stop << 16);
                // the entering column inherits the LEAVING variable (simplex.ts:339-349): every workgroup needs to know whether
                // that variable is unrestricted; only this workgroup sees the row map
                if (UNR && tid == 0 && !stop && c.unr[c.vibr[pr]] != 0) lead |= 1u << 24;
                const unsigned payload = tid == 0 ? lead : (tid == 1 ? (unsigned)qb : (unsigned)(qb >> 32));
                AG_STORE(f.decision[par] + tid, ((u64_t)tag << 32) | payload);
                if (UNR && tid == 0) sm.dec[0] = lead;
            }
            if (UNR) __syncthreads();
        } else {
            if (tid < 64) {
                unsigned spins = 0;
                int ok = 1;
                u64_t x = 0;
                for (;;) {
                    bool have = true;
                    if (tid < 3) { x = AG_LOAD(f.decision[par] + tid); have = (unsigned)(x >> 32) == tag; }
                    if (__all(have)) break;
                    __builtin_amdgcn_s_sleep(JSLP_POLL_SLEEP);  // 250 workgroups poll this one line: keep the load on it light
                    ++spins;
                    if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) { ok = 0; break; }
                    if (spins > F_SPIN) { if (tid == 0) AG_STORE(f.abort_flag, 1u); ok = 0; break; }
                }
                if (tid < 3) sm.dec[tid] = (unsigned)x;
                if (tid == 0) sm.ok = ok;
            }
            __syncthreads();
            if (!sm.ok) { end_code = 5; break; }
            pr = (int)(sm.dec[0] & 0xffffu);
            stop = (int)((sm.dec[0] >> 16) & 0xffu);
            quot = __longlong_as_double((long long)((u64_t)sm.dec[1] | ((u64_t)sm.dec[2] << 32)));
        }
        const bool leaving_unr = UNR && ((sm.dec[0] >> 24) & 1u);
        RT_MARK(3);
        if (stop == 3) { end_code = 2; unbounded_col = pc; break; }
        if (stop == 1) { end_code = 3; break; }
        if (stop == 2) { end_code = 6; break; }
        if (stop == 4) {  // phase 1 is over: the caller starts phase 2 with a fresh history (simplex.ts:14-23, 102)
            hist_n = 0;
            epoch += 1;
            return;
        }
        // the entries the pivot column itself receives (-k / quot, simplex.ts:386; row 0: -k0 / quot): one lane each, now, while
        // the winning row is in flight -- the lane that owns the column would run these divisions one after the other in step F
        constexpr bool PARDIV = PHASE == 2 && JSLP_RES_FAST != 0;
        if (PARDIV && tid <= ROWS) sm.nv[tid] = -(tid < ROWS ? sm.col[tid] : k0) / quot;
        // ---- E: the winning row: loaded speculatively together with its flag; re-loaded in the rare case the flag
        //         (which follows the winner's drain) was not up yet ------------------------------------------------------
        const int bw = pr / f.rpb;
        const int off_in = (bw * ld + c0) * 8;
        double pv[CPT];
#pragma unroll
        for (int j = 0; j < CPT; j++) pv[j] = 0.0;
        {
            // EVERY wave waits for its own copy of the flag, then loads its columns of the row: a
            // row loaded by a wave that ran ahead of the wave holding thread 0 (a cold instruction cache is enough) used to be accepted
            // on thread 0's later look at the flag.  A flag seen up means the row's stores were acknowledged before the flag store was
            // issued; my loads of the row are issued after my load of the flag.  A wave that gives up raises sm.okbad to this fetch's
            // number (the count only grows: no reset, no second word).
            efetch += 1;
            unsigned spins = 0;
            for (;;) {
                if (__builtin_amdgcn_readfirstlane((int)(F_TEST_LATE != 0 && (tid >> 6) == 0))) __builtin_amdgcn_s_sleep(127);  // (tests: the skew that used to break the fetch; a SCALAR branch -- s_sleep ignores EXEC, and predicated by EXEC alone it ran in every wave at every fetch: 157 k -> 102 k pivots/s)
                const u64_t flag = AG_LOAD(f.rowflagc[par] + (tid >> 6) * JSLP_F_MAXG + bw);
                if ((unsigned)flag == tag) break;  // (wave-uniform: one word, one request)
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                bool dead = false;
                if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) dead = true;
                if (spins > F_SPIN) { if ((tid & 63) == 0) AG_STORE(f.abort_flag, 1u); dead = true; }
                if (dead) { if ((tid & 63) == 0) atomicMax(&sm.okbad, efetch); break; }
            }
            asm volatile("" ::: "memory");
            if (colok) {  // the row, behind MY look at the flag (not next to it: the loads must not be served before the flag is)
#pragma unroll
                for (int j = 0; j < CPT; j += 2) {
                    if (c0 + j >= ld) continue;
                    const v4u_t v = par ? __builtin_amdgcn_raw_buffer_load_b128(rsrc1, off_in + j * 8, 0, 16)
                                        : __builtin_amdgcn_raw_buffer_load_b128(rsrc0, off_in + j * 8, 0, 16);
                    pv[j] = __longlong_as_double((long long)((u64_t)v.x | ((u64_t)v.y << 32)));
                    pv[j + 1] = __longlong_as_double((long long)((u64_t)v.z | ((u64_t)v.w << 32)));
                }
            }
            __syncthreads();
            if (sm.okbad == efetch) end_code = 5;
            if (!JSLP_RES_FAST) __syncthreads();
        }
        if (end_code == 5) break;
        RT_MARK(4);
        bool anyrow = true;  // phase 2: the entering cost is > precision, so some row always runs simplex.ts:381-383
        if (phase == 1) {
            // entering column: max -cost/coef over coef < -precision (simplex.ts:56-71; no unrestricted variables here)
            Cand best; best.v = -INFINITY; best.i = 0; best.b = 0;
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const int col = c0 + j;
                const double coef = pv[j];
                if (col >= 1 && col < W && ((UNR && ((R.unr >> j) & 1u)) || coef < -precision)) {
                    const double quo = -r0[j] / coef;
                    const bool take = best.v < quo;
                    best.v = take ? quo : best.v;
                    best.i = take ? col : best.i;
                }
            }
            best = block_reduce(best, MaxFirst(), sm.f.red);
            if (best.i == 0) { end_code = 7; break; }  // infeasible (simplex.ts:73-76), uniform
            pc = best.i;
            has_pc = colok && pc >= c0 && pc < c0 + CPT;
            if (has_pc) {
#pragma unroll
                for (int j = 0; j < CPT; j++)
                    if (pc == c0 + j) {
#pragma unroll
                        for (int i = 0; i < ROWS; i++) sm.col[i] = a[i][j];
                        sm.xq[0] = pv[j];  // quot = A[pr, pc]
                        sm.xq[1] = r0[j];  // k0 = A[0, pc]
                    }
            }
            __syncthreads();
            quot = sm.xq[0];
            k0 = sm.xq[1];
            if (c.check_cycles && allg) {  // simplex.ts:78-93 by every workgroup, on its own LDS history
                if (tid == 0) {
                    const int2 pair = make_int2(sm.lvibr[pr], sm.lvibc[pc]);
                    sm.lhist[hist_n] = pair;
                    if (b == 0) c.hist[hist_n] = pair;
                }
                __syncthreads();
                hist_n += 1;
                if (suffix_is_square(sm.lhist, hist_n, sm.f.red)) { end_code = 3; break; }
            } else if (c.check_cycles) {  // simplex.ts:78-93: only now is the (leaving, entering) pair known
                if (b == 0) {
                    int cstop = 0;
                    if (hist_n >= c.hist_cap) {
                        cstop = 2;
                    } else {
                        if (tid == 0) c.hist[hist_n] = make_int2(c.vibr[pr], c.vibc[pc]);
                        __syncthreads();
                        hist_n += 1;
                        if (suffix_is_square(c.hist, hist_n, sm.f.red)) cstop = 1;
                    }
                    if (tid == 0) AG_STORE(f.verdict[par], ((u64_t)tag << 32) | (unsigned)cstop);
                    stop = cstop;
                } else {
                    if (tid == 0) {
                        unsigned spins = 0;
                        int v = -1;
                        for (;;) {
                            const u64_t x = AG_LOAD(f.verdict[par]);
                            if ((unsigned)(x >> 32) == tag) { v = (int)(x & 3u); break; }
                            __builtin_amdgcn_s_sleep(JSLP_POLL_SLEEP);
                            ++spins;
                            if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) break;
                            if (spins > F_SPIN) { AG_STORE(f.abort_flag, 1u); break; }
                        }
                        sm.ok = v;
                    }
                    __syncthreads();
                    stop = sm.ok;
                    __syncthreads();
                }
                if (stop < 0) { end_code = 5; break; }
                if (stop == 1) { end_code = 3; break; }
                if (stop == 2) { end_code = 6; break; }
            }
        }
        double p[CPT];  // normalised pivot row (simplex.ts:352-364)
#pragma unroll
        for (int j = 0; j < CPT; j++) p[j] = 0.0;
        int tiny = 0;   // entries simplex.ts:381-383 zeroes as soon as ANY other row is eliminated
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
                    if (innz && !nonzero16(v) && v != 0.0) tiny |= 1 << j;
                }
                p[j] = v;
            }
        }
        if (phase == 1 && __syncthreads_or(tiny)) {
            // rare: the stored pivot row depends on whether any OTHER row has a non-zero entry in column pc
            int local_any = 0;
#pragma unroll
            for (int i = 0; i < ROWS; i++) {
                const int r = r_begin + i;
                if (r < r_end && r != pr && nonzero16(sm.col[i])) local_any = 1;
            }
            const int g = global_or(f, par, tag, local_any, sm, b);
            if (g < 0) { end_code = 5; break; }
            anyrow = g != 0;
        }
        bool nz[CPT];
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            if (anyrow && (tiny & (1 << j))) p[j] = 0.0;
            nz[j] = nonzero16(p[j]);
        }
        RT_MARK(5);
        // ---- F: update registers: cost row (every workgroup the same), then my rows ------------------------------------
        if (nonzero16(k0)) {
#pragma unroll
            for (int j = 0; j < CPT; j++)
                if (nz[j]) r0[j] = eliminate(r0[j], k0, p[j]);
            if (has_pc) {
                const double nv = PARDIV ? sm.nv[ROWS] : -k0 / quot;
#pragma unroll
                for (int j = 0; j < CPT; j++)
                    if (pc == c0 + j) r0[j] = nv;
            }
        }
        // (a macro, not a lambda: through a closure the register arrays a[][] / p[] / nz[] end up in scratch)
#define JSLP_RES_UPDATE_ROWS(LO, HI)                                                                        \
        _Pragma("unroll") for (int i = (LO); i < (HI); i++) {                                               \
            const int r = r_begin + i;                                                                      \
            if (r >= r_end) continue;                                                                       \
            if (r == 0) { /* workgroup 0 owns the cost row */                                               \
                _Pragma("unroll") for (int j = 0; j < CPT; j++) a[i][j] = r0[j];                            \
                continue;                                                                                   \
            }                                                                                               \
            if (r == pr) {                                                                                  \
                _Pragma("unroll") for (int j = 0; j < CPT; j++) a[i][j] = p[j];                             \
                continue;                                                                                   \
            }                                                                                               \
            const double ki = sm.col[i]; /* pivot-column entry of row i (still in LDS from step A) */       \
            if (nonzero16(ki)) {                                                                            \
                _Pragma("unroll") for (int j = 0; j < CPT; j++)                                             \
                    if (nz[j]) a[i][j] = eliminate(a[i][j], ki, p[j]);                                      \
                if (has_pc) {                                                                           \
                    const double nv = PARDIV ? sm.nv[i] : -ki / quot;                                       \
                    _Pragma("unroll") for (int j = 0; j < CPT; j++)                                         \
                        if (pc == c0 + j) a[i][j] = nv;                                                 \
                }                                                                                           \
            }                                                                                               \
        }
        JSLP_RES_UPDATE_ROWS(0, ROWS)
#undef JSLP_RES_UPDATE_ROWS
        if (tid == 0) {  // every workgroup's LDS maps (simplex.ts:339-349)
            const int leaving = sm.lvibr[pr], entering = sm.lvibc[pc];
            sm.lvibr[pr] = entering;
            sm.lvibc[pc] = leaving;
        }
        // workgroup 0 commits the basis change (simplex.ts:339-349)
        if (b == 0 && tid == 0) {
            const int leaving = c.vibr[pr], entering = c.vibc[pc];
            c.vibr[pr] = entering;
            c.vibc[pc] = leaving;
            c.rbv[entering] = pr;
            c.rbv[leaving] = -1;
            c.cbv[entering] = -1;
            c.cbv[leaving] = pc;
            if (trace_n < c.trace_cap) c.trace[trace_n] = make_int2(pr, pc);
        }
        if (UNR && has_pc) {
#pragma unroll
            for (int j = 0; j < CPT; j++)
                if (pc == c0 + j) {
                    R.unr = (R.unr & ~(1u << j)) | ((leaving_unr ? 1u : 0u) << j);
                }
        }
        trace_n += 1;
        if (phase == 1) it1 += 1; else it2 += 1;
        epoch += 1;
        RT_MARK(6);
        // ---- G: phase 2: price the new cost row -> entering column of the next pivot -----------------------------------
        if (phase == 2) {
            pc = price_row_lds<CPT, UNR>(r0, c0, pb, c, sm, &k0, R.unr, &R.neg);
            if (pc == 0) end_code = 1;
        }
    }
}

#include "jslp_resident_pipe.hip.h"  // the lean kernel's phase 2 (needs everything above, is needed by the kernel below)

// THREADS x CPT >= ld: <1024, 2> = lane pairs of columns, 4 waves per SIMD; <512, 4> = half the waves to synchronise,
// twice the independent work per lane (and 256 VGPRs per lane).
// XL = the XCD-LOCAL build (round 4): tableaus of up to 1024 x 1024 (8 MB: the 16 MB of vector registers of ONE XCD take them) on
// <= 32 workgroups that all sit on the same XCD -- the grid is JSLP_XL_SPREAD x G blocks of which every 8th works (observed
// dispatch: block b runs on XCD b % 8; HIP promises nothing, so the participants compare their XCC ids once per launch and give
// up -- ERR_BARRIER: the host rolls back and takes another path -- when they differ).  Same loops, same arithmetic; what changes
// is the transport (jslp_resident_pipe.hip.h, `XL`): the XCD's own L2 instead of memory, no write-back fence.
template <int THREADS, int CPT, int ROWS, bool UNR, bool LEAN = false, bool OPT = false, bool CHK = true, bool XL = false>
__global__ void __launch_bounds__(THREADS) k_simplex_resident(ResCtx f) {
    static_assert(!(OPT && UNR), "optional objectives with unrestricted variables: the fused pipeline");
    static_assert(!XL || LEAN, "XCD-local transport: lean builds only");
    if (XL && (blockIdx.x % JSLP_XL_SPREAD) != 0) return;  // (the seven blocks in between only steer the dispatcher)
    static_assert(!OPT || LEAN, "optional objectives: lean builds only");
    static_assert(ROWS <= JSLP_R_MAXROWS, "RSmem holds one entry per row of the workgroup");
    static_assert(sizeof(RSmem) <= 160 * 1024, "one workgroup per CU: all of the CU's LDS, no more");
    __shared__ RSmem sm;
    ResRegs<CPT, ROWS> R;
#ifdef JSLP_DEBUG_RESIDENT
    for (int i = 0; i < 8; i++) R.rt_acc[i] = 0;
    R.rt_retries = 0;
    R.rt_prev = __builtin_amdgcn_s_memtime();
#endif
    const Ctx& c = f.c;
    const int tid = threadIdx.x, b = XL ? (int)(blockIdx.x / JSLP_XL_SPREAD) : (int)blockIdx.x;
    const int ld = c.ld, H = f.H;
    const int c0 = tid * CPT;
    const bool colok = c0 < ld;
    const int r_begin = b * f.rpb, r_end = min(H, r_begin + f.rpb);
    DevState* st = c.st;
    static_assert(CPT % 2 == 0, "lanes load and store their columns as 16-byte pairs");
    if (XL && tid == 0) {  // placement census, first half: my XCC id leaves now (write-through: correct wherever the others sit), read below
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu;  // HW_REG_XCC_ID[3:0]
        AG_STORE(f.census + b, (0xA5A5A5A5ull << 32) | xcc);
    }

    // ---- load my rows and the cost row into registers ---------------------------------------------------
    double (&a)[ROWS][CPT] = R.a;
    double (&r0)[CPT] = R.r0;
#pragma unroll
    for (int j = 0; j < CPT; j += 2) {
        double2 t = make_double2(0, 0);
        if (c0 + j < ld) t = *reinterpret_cast<const double2*>(c.A + c0 + j);
        r0[j] = t.x; r0[j + 1] = t.y;
    }
#pragma unroll
    for (int o = 0; o < 3; o++)
#pragma unroll
        for (int j = 0; j < CPT; j++) R.oo[o][j] = (OPT && o < c.n_opt && c0 + j < ld) ? c.oo[(long long)o * ld + c0 + j] : 0.0;
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
        const int r = r_begin + i;
        const bool mine = i < f.rpb && r < r_end && colok;
#pragma unroll
        for (int j = 0; j < CPT; j += 2) {
            double2 t = make_double2(0, 0);
            if (mine && c0 + j < ld) t = *reinterpret_cast<const double2*>(c.A + (long long)r * ld + c0 + j);
            a[i][j] = t.x; a[i][j + 1] = t.y;
        }
    }
    const int status0 = st->status;
    R.hist_n = st->hist_n;
    const int it1_start = st->it1, it2_start = st->it2;
    R.it1 = it1_start; R.it2 = it2_start;
    R.trace_n = st->trace_n;
    // fresh simplex() (k_begin ran: phase 1 first) or a hand-over after a phase 1 done elsewhere
    if (status0 != ST_RUNNING && status0 != ST_PHASE1_DONE) return;  // uniform
    int phase = status0 == ST_PHASE1_DONE ? 2 : 1;

#ifdef JSLP_DEBUG_RESIDENT
    if (f.dbg) {  // micro-costs in this kernel's own geometry
        u64_t t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 64; i++) __syncthreads();
        u64_t t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 64; i++) { atomicMin(&sm.p_batch, tid + i); }
        __syncthreads();
        u64_t t2 = __builtin_amdgcn_s_memtime();
        u64_t acc = 0;
        for (int i = 0; i < 16; i++) { acc += AG_LOAD(f.rowflag[0] + ((acc + i) & 63)); }
        u64_t t3 = __builtin_amdgcn_s_memtime();
        double dv = 1.0 + (double)tid;
        for (int i = 0; i < 16; i++) dv = 3.0 / dv + 1.0;
        u64_t t4 = __builtin_amdgcn_s_memtime();
        if (tid == 0 && b == 1) {
            u64_t* o = f.dbg + (long long)512 * f.G * 2 + 12288 + 64;
            o[0] = (t1 - t0) / 64; o[1] = (t2 - t1) / 64; o[2] = (t3 - t2) / 16; o[3] = (t4 - t3) / 16; o[4] = acc + (u64_t)dv;
        }
        __syncthreads();
    }
#endif
    // every workgroup's own copy of the maps (and of the unrestricted flags): see RSmem
    for (int r = tid; r < f.H; r += (int)blockDim.x) sm.lvibr[r] = c.vibr[r];
    for (int col = tid; col < c.W; col += (int)blockDim.x) sm.lvibc[col] = c.vibc[col];
    if (UNR)
        for (int v = tid; v < f.n_idx && v < JSLP_R_LUNR; v += (int)blockDim.x) sm.lunr[v] = c.unr[v];
    if (tid == 0) { reset_reductions(sm); sm.okbad = 0u; }
    if (LEAN && tid == THREADS - 64) { sm.gp_vibr = c.vibr; sm.gp_vibc = c.vibc; sm.gp_rbv = c.rbv; sm.gp_cbv = c.cbv; sm.gp_trace = c.trace; sm.gp_trace_cap = c.trace_cap; sm.gp_hist = c.hist; }
    __syncthreads();
    // pricing batch of my columns (simplex.ts:118-127): fixed for the whole solve
    int pb[CPT];
#pragma unroll
    for (int j = 0; j < CPT; j++) pb[j] = c.use_partial && c0 + j >= 1 ? (c0 + j - 1) / c.batch : 0;
    R.k0 = 0.0;  // reduced cost of the entering column = cost-row entry of column pc
    R.unr = 0;
    R.neg = 0;
    if (UNR) {  // which of my columns carry an unrestricted variable (model.unrestrictedVariables, tableau.ts:57)
#pragma unroll
        for (int j = 0; j < CPT; j++)
            if (c0 + j >= 1 && c0 + j < c.W && c.unr[c.vibc[c0 + j]] != 0) R.unr |= 1u << j;
    }
    R.pc = 0;
    R.end_code = 0;  // 1 optimal, 2 unbounded, 3 cycle, 4 iteration cap, 5 aborted hand-off, 6 history full, 7 infeasible
    R.unbounded_col = 0;
    R.epoch = 0;
    R.efetch = 0;
    if (XL) {  // placement census, second half: every participant sits on MY XCD, or nobody starts (plain stores into one XCD's L2 are
               // invisible from another: the loops below would read stale summaries)
        int same = 1;
        if (tid < f.G) {
            const unsigned mine = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu;
            unsigned spins = 0;
            for (;;) {
                const u64_t x = AG_LOAD(f.census + tid);
                if ((unsigned)(x >> 32) == 0xA5A5A5A5u) { same = ((unsigned)x & 0xfu) == mine ? 1 : 0; break; }
                __builtin_amdgcn_s_sleep(JSLP_POLL_SLEEP);
                ++spins;
                if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) { same = 0; break; }
                if (spins > F_SPIN) { AG_STORE(f.abort_flag, 1u); same = 0; break; }
            }
        }
        if (!__syncthreads_and(same)) {
            if (tid == 0) AG_STORE(f.abort_flag, 1u);
            R.end_code = 5;
        }
    }
    // The lean build of the tall / wide geometries is a phase-2 kernel: the host runs their phase 1 through the fused pipeline
    // (run_simplex), the phase-1 loop is not even compiled for them (next to 64-72 MB of tableau it does not fit the registers: it
    // spilled ~0.5 KB per lane, and r03_j saw a 3001 x 3001 solve go wrong after ONE pass through that spilling code).
    constexpr bool P2ONLY = LEAN && !XL && (ROWS > JSLP_R_ROWS || CPT > 4);
    if (phase == 1 && R.end_code == 0) {
        if (P2ONLY) {
            R.end_code = 5;  // never launched like this; if it were, the host rolls back and streams (like an aborted hand-off)
        } else {
            if (LEAN) resident_phase1_pipe<THREADS, CPT, ROWS, OPT, CHK, XL, UNR>(f, sm, R, it1_start, it2_start);
            else resident_phase<1, THREADS, CPT, ROWS, UNR, LEAN>(f, sm, R, it1_start, it2_start, pb);
            if (R.end_code == 0) phase = 2;
        }
    }
    if (R.end_code == 0) {  // phase 2 (simplex.ts:100-325): first entering column, then the hot loop
        if (LEAN) {
            resident_phase2_pipe<THREADS, CPT, ROWS, OPT, CHK, XL, UNR>(f, sm, R, it1_start, it2_start, pb);  // (prices at the top of its loop)
        } else {
            R.pc = price_row_lds<CPT, UNR>(r0, c0, pb, c, sm, &R.k0, R.unr, &R.neg);
            if (R.pc == 0) R.end_code = 1;
            else resident_phase<2, THREADS, CPT, ROWS, UNR>(f, sm, R, it1_start, it2_start, pb);
        }
    }
    const int end_code = R.end_code, unbounded_col = R.unbounded_col, hist_n = R.hist_n, it1 = R.it1, it2 = R.it2;
    const unsigned epoch = R.epoch;
    const long long trace_n = R.trace_n;
    (void)epoch;


#ifdef JSLP_DEBUG_RESIDENT
    if (f.dbg && tid == 0 && (b == 0 || b == 100 || b == f.G - 1)) {
        u64_t* o = f.dbg + (long long)512 * f.G * 2 + 12288 + (b == 0 ? 0 : (b == 100 ? 16 : 32));
        for (int i = 0; i < 8; i++) o[i] = R.rt_acc[i];
        o[8] = epoch;
        o[9] = R.rt_retries;
    }
#endif
    // ---- epilogue: registers -> tableau, workgroup 0 -> state -----------------------------------------------------------
    if (end_code != 5) {
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            const int r = r_begin + i;
            if (i < f.rpb && r < r_end && colok) {
#pragma unroll
                for (int j = 0; j < CPT; j += 2)
                    if (c0 + j < ld) *reinterpret_cast<double2*>(c.A + (long long)r * ld + c0 + j) = make_double2(a[i][j], a[i][j + 1]);
            }
        }
    }
    if (OPT && b == 0 && end_code != 5 && colok) {  // the optional objective rows go back with the tableau (workgroup 0's copies: all are equal)
#pragma unroll
        for (int o = 0; o < 3; o++)
            if (o < c.n_opt) {
#pragma unroll
                for (int j = 0; j < CPT; j++)
                    if (c0 + j < ld) c.oo[(long long)o * ld + c0 + j] = R.oo[o][j];
            }
    }
    if (b == 0 && tid == 0) {
        st->it1 = it1;
        st->it2 = it2;
        st->trace_n = trace_n;
        st->hist_n = hist_n;
        st->iters_left -= (it1 - it1_start) + (it2 - it2_start);
        st->do_pivot = 0;
        // end_code 8 (lean kernel only): the solve goes on in the general kernel, which picks the phase up from the status
        st->status = end_code == 8 ? (phase == 2 ? ST_PHASE1_DONE : ST_RUNNING) : ST_DONE;
        st->phase = phase;
        if (phase == 2) { st->entered_phase2 = 1; st->feasible = 1; }  // phase 1 found no violated row (simplex.ts:51-54)
        st->obj_cell = r0[0];  // column 0 of the cost row
        if (end_code == 1) st->optimal = 1;
        if (end_code == 2) { st->bounded = 0; st->unbounded_var = c.vibc[unbounded_col]; }
        if (end_code == 3) { st->cycle_phase = phase; st->feasible = 0; }
        if (end_code == 7) st->feasible = 0;
        if (end_code == 4) st->err = ERR_ITER_LIMIT;
        if (end_code == 5) st->err = ERR_BARRIER;
        if (end_code == 6) st->err = ERR_HIST_FULL;
    }
}
