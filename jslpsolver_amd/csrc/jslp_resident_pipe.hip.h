// jslp_resident_pipe.hip.h -- the LEAN register-resident kernel's phase 2: a software-pipelined pivot loop.
// Included by jslp_resident.hip.h between its building blocks (ResCtx / RSmem / ResRegs) and its kernel.
#pragma once

// ===================================================================================================================
// Why a second loop.  k_simplex_resident's general pivot loop (jslp_resident.hip.h) runs its steps strictly one after the
// other: summary -> publish -> gather -> decide -> fetch the winning row -> update my rows -> price -> summary ...; the row
// update (16 waves x 8 rows x 2 columns of two-rounding eliminations, ~3 k cycles) and the fabric hop of the NEXT pivot's
// summary (~2 us) both sit on the critical path although neither needs the other.  Here the loop is rotated:
//
//   winner row of pivot t lands
//     -> normalise it, update the COST row only                                                    (simplex.ts:352-364, 394-412)
//     -> price it: entering column of pivot t+1                                                    (simplex.ts:118-219)
//     -> the wave that holds that column evaluates what pivot t makes of the column in my rows (eight lanes, one row each)
//        and runs the ratio test on it                                                              (simplex.ts:271-296)
//     -> the summary of pivot t+1 is published
//     -> ONLY NOW the row update of pivot t (simplex.ts:367-391), while the summaries cross the fabric; the row that can win
//        is published from inside that pass
//     -> gather, decide, fetch the winning row of pivot t+1 ...
//
// Same arithmetic on the same operands in the same order for every cell (each cell still receives exactly one
// `a - k * p` with both roundings per pivot; "early" values are computed from the same inputs as the update pass computes
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
//   * pricing moved to the top of the loop (the loop is entered with the tableau whole and leaves a pending update to its epilogue);
//   * 6 workgroup barriers per pivot instead of 8 (cycle check off).
//
// Round 6 (profiles/r06_headline_sections.md): the 2- / 4-column geometries with <= 8 rows per workgroup publish the candidate row already NORMALISED, with
// quot and the cost row's new entry in a checksummed 32-byte record (NPUB: no division and no barrier between the row fetch and the next pricing); ONE wave polls
// the summaries, four looks per lane; the pricing's reduction words are double-buffered by pivot parity; workgroup 0's global commit follows one iteration
// later as fire-and-forget stores; 5 workgroup barriers per pivot.
// (The lab notebook of these loops -- every variant measured and left out, with its numbers -- lives in profiles/r02_r04_measurement_log.md (rounds 3 and 4)
//  and profiles/r06_headline_sections.md (round 6); the rejected variants' code is profiles/r06_rejected_switches.patch.)
// ===================================================================================================================
// -DJSLP_CHAOS_BUILD (tests / diagnosis only, never the shipped library): at every phase boundary of the pipelined loops one wave of
// the workgroup -- a different one per pivot and boundary -- sleeps ~6 k cycles (JSLP_TEST_RESIDENT_LATE_WAVE0=2), and with =3 every
// fifth workgroup sleeps ~4 k more: whatever in these loops relies on waves or workgroups arriving together shows up as lost pivots.
// (=2 exactly also makes wave 0 of every publishing workgroup raise its flag word ~8 k cycles BEFORE it stores its columns of the
//  candidate row -- the litmus of the checksummed hand-over, JSLP_PUBLISH_ROW_PLAIN; odd values make wave 0 late at every row fetch;
//  =4 / =6: the TORN row -- wave 0's even / odd lanes store their pairs, the flag word goes up, the other lanes store ~8 k cycles later:
//  a reader's first looks find a row that is half this epoch's and half the one two epochs back, which only the checksum can tell)
#ifdef JSLP_CHAOS_BUILD
#undef RT_MARK
#define RT_MARK(p)                                                                                                        \
    do {                                                                                                                  \
        if (F_TEST_LATE >= 2) {                                                                                     \
            const unsigned who_ = (R.epoch * 7u + (unsigned)(p) * 3u) % (unsigned)(THREADS / 64);                         \
            if (__builtin_amdgcn_readfirstlane((int)((unsigned)(threadIdx.x >> 6) == who_))) __builtin_amdgcn_s_sleep(100); \
            if (F_TEST_LATE >= 3 && (R.epoch + (unsigned)(p)) % 5u == (unsigned)blockIdx.x % 5u) __builtin_amdgcn_s_sleep(60); \
        }                                                                                                                 \
    } while (0)
#endif
// (Round 6 pruned the switches of measured-and-rejected variants out of this header -- winner-only publication with and without tags
//  inside the data, quot read next to the row, early / mid / ahead looks at the summaries, the row update split around the gather, the
//  in-wave pricing, the per-row update form of the 2- / 4-column geometries: profiles/r06_rejected_switches.patch re-applies them, DESIGN.md
//  section 5 has their numbers.)
#ifndef JSLP_PIPE_ROW_CHECKSUM
// 1: the candidate row is handed over END TO END: each wave of the publishing workgroup stores, right behind its 16-byte stores of the
// row, ONE 8-byte word = (64-bit checksum of what it stored) ^ (epoch tag x odd constant) into its copy of the row flag; the wave of a
// reader that fetches those columns loads flag and row TOGETHER, recomputes the checksum and repeats the look until the two agree.
// Nothing is assumed about the order in which the fabric delivers write-through stores -- the reader verifies what it received -- so
// the winner's release (buffer_wbl2 + flag store AFTER the decision), the readers' wait for that flag and the drain in front of the
// gather's barrier all leave the critical path.  0: round 3's hand-over (winner drains, fences, raises the flag; readers wait, load).
#define JSLP_PIPE_ROW_CHECKSUM 1
#endif
// round 6: ONE wave gathers the <= 256 summaries (lane l looks at workgroups l, l + 64, l + 128, l + 192: four 16-byte loads in flight per lane) instead of
// four waves of one load per lane each.  The four waves' looks returned at four different times and the barrier that closes the gather waited
// for the slowest; every thread then folded four partial results.  One wave: one round trip, one partial result (config 3a 172.6 k -> 179 k
// pivots/s together with the parity-buffered pricing words, 3b with the cycle check 145 k -> 155 k).
#ifndef JSLP_PIPE_K_BROADCAST
#define JSLP_PIPE_K_BROADCAST 1  // round 6: see JSLP_XL_UPDATE_PASS_M
#endif
#ifndef JSLP_PIPE_NORM_PUB
#define JSLP_PIPE_NORM_PUB 1  // round 6: the candidate row leaves NORMALISED (see NPUB in resident_phase2_pipe)
#endif
#ifndef JSLP_PIPE_ROW_CHECKSUM_MAXCPT
#define JSLP_PIPE_ROW_CHECKSUM_MAXCPT 4  // the 6- / 8-column geometries keep round 3's hand-over: checksummed they ran 3001 x 3001 113.7 k -> 97.1 k and 2001 x 4001 116.0 k -> 100.8 k pivots/s (r04_y: three / four pairs to hash per lane in front of the pivot row's normalisation, in kernels at their register limit)
#endif
// the checksum: every 32-bit word a lane stores, times its own odd 32-bit constant, summed in 64 bits (v_mad_u64_u32: one instruction
// per word); the lane's sum times (2 x thread + 1) -- lanes with the SAME old and the SAME new content (a stale line covers four lanes'
// pairs) must not cancel; the two 32-bit halves of that, each summed over the wave (DPP adds, no carries between the halves).  A stale
// or torn row goes unnoticed only if both half-sums survive: ~2^-64 per such event, themselves rare (tools/resident_stress.py counts them)
#define JSLP_CK_K32(k) ((0x9E3779B1u * (unsigned)(2 * (k) + 1)) | 0x80000001u)
#define JSLP_CK_PAIR(CK, LO, HI, j)                                                                                               \
    do {                                                                                                                          \
        (CK) += (u64_t)(unsigned)(LO) * JSLP_CK_K32(2 * (j)) + (u64_t)(unsigned)((LO) >> 32) * JSLP_CK_K32(2 * (j) + 1);           \
        (CK) += (u64_t)(unsigned)(HI) * JSLP_CK_K32(2 * (j) + 2) + (u64_t)(unsigned)((HI) >> 32) * JSLP_CK_K32(2 * (j) + 3);       \
    } while (0)
#define JSLP_CK_TAGMIX(tag) ((u64_t)(tag) * 0xD6E8FEB86659FD93ull)    // what separates this epoch's flag word from the one two epochs back in the same slot
#ifndef JSLP_PIPE_POLL_MISSING_ONLY
#define JSLP_PIPE_POLL_MISSING_ONLY 1  // the gather re-issues only the looks whose summary is still missing (0: all four looks of every lane, every time)
#endif
#ifndef JSLP_PIPE_S_VIA_LDS
#define JSLP_PIPE_S_VIA_LDS 1
#endif
#define JSLP_PUB_SKEW 256     // bytes added to a workgroup's slot of the candidate-row buffer (see SLOT)
#ifndef JSLP_G16_STRIDE
#define JSLP_G16_STRIDE 64   // bytes between two workgroups' summary granules (a 64-byte line each)
#endif
#define JSLP_PIPE_KCHUNK 8    // pivot-column entries the update pass keeps in flight (registers: the tall / wide geometries have few to spare)

// A HOST-requested abort, in the SHIPPED build (the test hooks above exist in the test library only).  Every JSLP_HOST_ABORT_PERIOD-th pivot,
// the first polling wave of the LAST workgroup -- in the RETRY path of its look at the summaries, which a healthy pivot passes through
// anyway (the summaries take 2-4 k cycles to arrive, the first look leaves right behind the row update) -- looks at one word of pinned host
// memory (its address sits behind the device copy of the context: no kernel argument).  When the host has raised it, the wave raises the
// device-wide abort flag and reports the gather as failed: its workgroup leaves through the exit every timed-out gather takes, everybody
// else meets its silence in the next gather, sees the flag after 64 polls and leaves too -- the epilogue is skipped, the host rolls slot 0
// back and solves through the streaming kernels (run_simplex).  Raised by the engine itself JSLP_INJECT_RESIDENT_ABORT_US microseconds
// after the launch (tests/test_pool_and_extras.py: the rollback path of the library users load).
// (First form, measured r05_a: a check at the top of the pivot loop with a workgroup barrier around the word -- one scalar branch per pivot,
//  taken once in 1024 -- cost config 3a 169.5 k -> 153.1 k pivots/s and phase 1 157 k -> 134 k: the extra exit changed the loop's register
//  allocation, 90 -> 107 VGPRs.  In the retry path the fast path's code is the round-4 code.)
#define JSLP_HOST_ABORT_PERIOD 1024u
#ifndef JSLP_HOST_ABORT
#define JSLP_HOST_ABORT 1
#endif
// (round 6: at those pivots that wave treats its FIRST look as failed whatever it found -- JSLP_HOST_ABORT_FORCE_RETRY -- so that the word is
//  looked at every 1024 pivots for certain: with one polling wave whose first look leaves ~1 k cycles later than round 5's the last
//  workgroup's look began to succeed at once and a 2833-pivot solve missed its abort -- tests/test_pool_and_extras.py, n = 1000)
#define JSLP_HOST_ABORT_FORCE_RETRY(SPINS) (JSLP_HOST_ABORT && (SPINS) == 0u && (epoch & (JSLP_HOST_ABORT_PERIOD - 1u)) == JSLP_HOST_ABORT_PERIOD - 1u && b == f.G - 1 && wv == HAWV)
#define JSLP_HOST_ABORT_IN_SPIN(SPINS, SWEPT)                                                                                     \
    if (JSLP_HOST_ABORT && (SPINS) == 1u && (epoch & (JSLP_HOST_ABORT_PERIOD - 1u)) == JSLP_HOST_ABORT_PERIOD - 1u && b == f.G - 1 && wv == HAWV) { \
        const unsigned* ha_ = *reinterpret_cast<const unsigned* const*>(reinterpret_cast<const char*>(f.cdev) + sizeof(Ctx));     \
        if (ha_ != nullptr && __hip_atomic_load(ha_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { /* (one address: uniform) */ \
            if (lane == 0) AG_STORE(f.abort_flag, 1u);                                                                            \
            (SWEPT) = false;                                                                                                      \
            break;                                                                                                                \
        }                                                                                                                         \
    }

__device__ __forceinline__ double readlane_f64(double x, int src_lane) {  // src_lane must be wave-uniform
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src_lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), src_lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}


// ---- the lean kernel's cycle check (simplex.ts:78-93 / 305-320) ------------------------------------------------------------------
static_assert(offsetof(RSmem, cycbits_tail) == offsetof(RSmem, lhist) + sizeof(int2) * JSLP_R_LHIST, "the filter runs from inside lhist into cycbits_tail");
static_assert((JSLP_R_LHIST - JSLP_PIPE_LHIST) * 8 + JSLP_R_CYCEXTRA * 4 == JSLP_PIPE_CYCBITS / 8, "filter size");
__device__ __forceinline__ unsigned* cyc_bits(RSmem& sm) { return reinterpret_cast<unsigned*>(&sm.lhist[JSLP_PIPE_LHIST]); }
// first step (thread 0): has this (leaving, entering) pair been seen in this phase?  Sets its two bits.
__device__ __forceinline__ int cyc_pair_seen(RSmem& sm, int2 pair) {
    static_assert(JSLP_PIPE_CYCBITS == (1 << 19), "the hashes below yield 19 bits");
    unsigned* bits = cyc_bits(sm);
    const unsigned k = (unsigned)pair.x * 0x9E3779B1u ^ (unsigned)pair.y * 0x85EBCA6Bu;
    const unsigned h1 = k >> (32 - 19), h2 = (k * 0xC2B2AE35u + 0x27D4EB2Fu) >> (32 - 19);
    const unsigned o1 = bits[h1 >> 5], m1 = 1u << (h1 & 31u);
    bits[h1 >> 5] = o1 | m1;
    const unsigned o2 = bits[h2 >> 5], m2 = 1u << (h2 & 31u);  // (read after the first write: h1 and h2 may share a word)
    bits[h2 >> 5] = o2 | m2;
    return ((o1 & m1) != 0u && (o2 & m2) != 0u) ? 1 : 0;
}
// the suffix test (suffix_is_square of jslp_core.inc.h) over a history whose first JSLP_PIPE_LHIST pairs sit in LDS and the rest
// in this workgroup's global slice; rare (only behind a filter hit), so the global reads do not matter
// (`last` = the newest pair, handed over in a register: thread 0 stored it a moment ago)
__device__ __forceinline__ bool cyc_suffix_is_square(const int2* lds, const int2* glob, int n, int2 last, Smem& red) {
    auto at = [&](int i) -> int2 { return i < JSLP_PIPE_LHIST ? lds[i] : glob[i]; };
    int found = 0;
    for (int L = 1 + threadIdx.x; 2 * L <= n; L += blockDim.x) {
        const int2 a = at(n - 1 - L);
        if (a.x != last.x || a.y != last.y) continue;
        bool eq = true;
        for (int i = 0; i < L - 1; i++) {
            const int2 x = at(n - 2 * L + i), y = at(n - L + i);
            if (x.x != y.x || x.y != y.y) { eq = false; break; }
        }
        if (eq) found = 1;
    }
    (void)red;
    return __syncthreads_or(found) != 0;
}

// the two 32-bit halves of a 64-bit word, each summed over the wave modulo 2^32 (every exchange pairs disjoint groups: lanes, pairs,
// quads, the halves of a 16-lane row; then the four rows) -- the checksummed row hand-over's reduction
__device__ __forceinline__ u64_t u64_wave_add_halves(u64_t x) {
    unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
#define JSLP_ADD_DPP(CTRL)                                                                          \
    lo += (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);           \
    hi += (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
    JSLP_ADD_DPP(0xB1) JSLP_ADD_DPP(0x4E) JSLP_ADD_DPP(0x141) JSLP_ADD_DPP(0x140)
#undef JSLP_ADD_DPP
    const unsigned slo = ((unsigned)__builtin_amdgcn_readlane((int)lo, 0) + (unsigned)__builtin_amdgcn_readlane((int)lo, 16)) +
                         ((unsigned)__builtin_amdgcn_readlane((int)lo, 32) + (unsigned)__builtin_amdgcn_readlane((int)lo, 48));
    const unsigned shi = ((unsigned)__builtin_amdgcn_readlane((int)hi, 0) + (unsigned)__builtin_amdgcn_readlane((int)hi, 16)) +
                         ((unsigned)__builtin_amdgcn_readlane((int)hi, 32) + (unsigned)__builtin_amdgcn_readlane((int)hi, 48));
    return (u64_t)slo | ((u64_t)shi << 32);
}
// a lane's checksum -> the wave's word
#define JSLP_CK_WAVE(CK) u64_wave_add_halves((CK) * (u64_t)(unsigned)(2 * tid + 1))

// ---- pricing of the lean pipelined phase 2 (simplex.ts:118-219), round 6 -------------------------------------------------------------------
// price_row_lds's three LDS-atomic rounds (first batch holding a candidate, best value in it, first column with that value) on reduction words
// that are DOUBLE-BUFFERED by pivot parity: the words of the next pivot are reset right behind this pivot's first barrier -- two barriers behind
// their last read, two in front of their next use -- so no reset sits in front of the gather's barrier or behind the row fetch any more (the
// QDIRECT flows have no barrier there).  `*claim` = this wave holds the entering column and runs the ratio test.
// (Measured and rejected twice -- round 4, and round 6 with the registers to spare: when the winning batch lies inside ONE wave, ~60 % of the
//  pivots, that wave folds value and column in registers and goes straight on to the ratio test, two barriers fewer: 163.1 k against 179.1 k
//  pivots/s on config 3a, profiles/r06_rejected_switches.patch.)
// Returns the entering column (0: none -> optimal; uniform).
template <int THREADS, int CPT, bool UNR>
__device__ __forceinline__ int price_row_pipe(const double (&x)[CPT], int c0, const int (&pb)[CPT], const Ctx& c, RSmem& sm, int par,
                                              double* value, unsigned unr, int* neg, bool* claim) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double bv = c.precision;
    int bi = 0, bb = 0, bneg = 0;
#pragma unroll
    for (int j = 0; j < CPT; j++) {  // my columns in order: earlier batch first, bigger value inside a batch, first index on ties
        const int col = c0 + j;
        const bool isneg = UNR && ((unr >> j) & 1u) && x[j] < 0;
        const double val = isneg ? -x[j] : x[j];
        const bool ok = col >= 1 && col < c.W && val > c.precision;
        const bool take = ok && (bi == 0 || pb[j] < bb || (pb[j] == bb && val > bv));
        bv = take ? val : bv;
        bi = take ? col : bi;
        bb = take ? pb[j] : bb;
        bneg = take ? (isneg ? 1 : 0) : bneg;
    }
    {   // batch ids grow with the lane index: the wave's earliest batch is that of its first candidate lane
        const unsigned long long m = __ballot(bi != 0);
        if (m != 0ull) {
            const int first = __ffsll((long long)m) - 1;
            const int wave_b = __builtin_amdgcn_readlane(bb, first);
            if (lane == 0) atomicMin(&sm.pw_batch[par], wave_b);
        }
    }
    __syncthreads();
    const int wb = sm.pw_batch[par];
    if (tid == THREADS - 1) { sm.pw_batch[par ^ 1] = 0x7fffffff; sm.pw_val[par ^ 1] = 0; sm.pw_col[par ^ 1] = 0x7fffffff; }  // the NEXT pivot's words: last read two barriers ago, first used behind two more
    *claim = false;
    if (wb == 0x7fffffff) return 0;  // uniform: no candidate anywhere -> optimal
    const u64_t bits = (u64_t)__double_as_longlong(bv);
    const bool cand = bi != 0 && bb == wb;
    if (cand) atomicMax(&sm.pw_val[par], bits);
    __syncthreads();
    const u64_t wvv = sm.pw_val[par];
    if (cand && bits == wvv) atomicMin(&sm.pw_col[par], bi);
    __syncthreads();
    const int pcol = sm.pw_col[par];
    double v = __longlong_as_double((long long)wvv);
    if (UNR) {  // the lane holding the winner knows the sign of its reduced cost
        if (cand && bi == pcol) sm.pw_neg[par] = bneg;
        __syncthreads();
        *neg = sm.pw_neg[par];
        if (*neg) v = -v;
    }
    *value = v;
    *claim = wv == ((pcol / CPT) >> 6);
    return pcol;
}

#define JSLP_R_MAXOPT 3  // optional objective rows the lean kernel keeps in registers (priorities "strong" / "medium" / "weak": model.ts:141-160)

// simplex.ts:221-263 on the lanes' own copies: no column prices out on the main row -> the optional objectives break the tie, in
// priority order, among the columns whose reduced cost is within +-precision on the main row and on every earlier objective.
// Returns the entering column (0: none -> optimal).  A rare path (a generic three-barrier block reduction per objective).
template <int CPT>
__device__ __forceinline__ int price_optional_regs(const double (&x)[CPT], const double (&oo)[JSLP_R_MAXOPT][CPT], int c0, const Ctx& c, RSmem& sm) {
    const double precision = c.precision;
    for (int o = 0; o < c.n_opt && o < JSLP_R_MAXOPT; o++) {
        Cand best; best.v = precision; best.i = 0; best.b = 0;
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const int col = c0 + j;
            if (col < 1 || col >= c.W) continue;
            bool deferred = -precision < x[j] && x[j] < precision;
#pragma unroll
            for (int q = 0; q < JSLP_R_MAXOPT; q++) {
                if (q > o || !deferred) break;
                const double v = oo[q][j];
                if (q < o) { deferred = -precision < v && v < precision; continue; }
                if (-precision < v && v < precision) break;  // this objective does not price the column out either
                const bool take = v > best.v;  // strict: my columns ascend, ties keep the earlier one (no unrestricted variables here)
                best.v = take ? v : best.v;
                best.i = take ? col : best.i;
            }
        }
        const Cand e = block_reduce(best, PriceFirst(), sm.f.red);
        if (e.i != 0) return e.i;
    }
    return 0;
}

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

// ---- XCD-local build (XL): the pending pivot's row update, restructured for 32 rows x 2 columns per lane.  JSLP_PIPE_UPDATE_ROW
// spends ~55 instructions per row on special cases (is this the pivot row / the cost row / does my lane hold the pivot column / is this
// the row to publish) next to the 4 that do the work: 230 cycles per row, 10-13 k cycles per pivot at 32 rows (r04_c).  Here the row
// loop is the gate `|k| > 1e-16` (simplex.ts:370-375; uniform: k is a broadcast LDS read) and the eliminations -- without the
// per-column gate when every column of the wave is live (dense pivot row; padding columns hold +0 in every row and in p, and
// (+0) - k * (+0) = +0) -- and the special cases follow ONCE per pivot: the pivot column's own entries by the one wave that holds
// it, the pivot row by its owner, the cost-row mirror by workgroup 0.  Same operands, same two roundings per cell.
// (a chain `if (i == row) a[i][j] = ...` over the unrolled rows is what the optimizer likes to fold into ONE store at a[row][j] -- a
//  dynamic index, which sends the whole register array to scratch: 456 bytes per lane, 1648 scratch instructions, 28-47 k cycles per
//  update pass (r04_d).  An empty asm per iteration makes every comparison's operand its own opaque value.)
#define JSLP_OPAQUE_SGPR(x) ({ int o_ = (x); asm volatile("" : "+s"(o_)); o_; })
#define JSLP_XL_UPDATE_PASS()                                                                                                     \
    do {                                                                                                                          \
        /* lane i < ROWS holds the pivot-column entry of my row i; the gate |k| > 1e-16 (simplex.ts:370-375) of all rows is ONE     \
           ballot, a row's k reaches the multiplier by two readlanes (scalar operands): per (row, column) a scalar bit test, a     \
           scalar branch and the two roundings -- no per-row LDS read, no per-cell select; the per-column gate is the EXEC mask of \
           the column's pass over the rows */                                                                                      \
        const double kl_ = lane < ROWS ? sm.colb[par_p][lane] : 0.0;                                                              \
        const unsigned km_ = (unsigned)__ballot(nonzero16(kl_)); /* (rows beyond the workgroup's share hold zeros: bit clear) */   \
        /* (measured, r04_e ... r04_g, XCD-local build: per-row LDS reads + per-cell SELECTS 5.4-6.9 k cycles per pass; readlane multipliers  \
            3.9-5.9 k; one pass over the rows with the column gate as an EXEC-masked branch per cell 5.1-8.5 k; a second, gate-free copy of \
            the loop for dense pivot rows costs the register allocator ~400 spills.  Round 6, JSLP_PIPE_K_BROADCAST: the multipliers of eight \
            rows come as BROADCAST LDS reads (one address for the whole wave) into vector registers -- a cell is then scalar bit test,    \
            branch, v_mul_f64, v_add_f64; the two v_readlane + s_nop per cell, which sixteen lock-step waves paid on the VALU port, are gone) */ \
        if (JSLP_PIPE_K_BROADCAST) {                                                                                              \
            _Pragma("unroll") for (int i0 = 0; i0 < ROWS; i0 += 8) {                                                              \
                double kb_[8];                                                                                                    \
                _Pragma("unroll") for (int i = 0; i < 8; i++) kb_[i] = (i0 + i < ROWS) ? sm.colb[par_p][i0 + i] : 0.0;            \
                _Pragma("unroll") for (int j = 0; j < CPT; j++) {                                                                 \
                    if ((nzm >> j) & 1u) {                                                                                        \
                        _Pragma("unroll") for (int i = 0; i < 8; i++)                                                             \
                            if (i0 + i < ROWS && (km_ & (1u << (i0 + i)))) a[i0 + i][j] = eliminate(a[i0 + i][j], kb_[i], p[j]);  \
                    }                                                                                                             \
                }                                                                                                                 \
            }                                                                                                                     \
        } else {                                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < CPT; j++) {                                                                         \
            if ((nzm >> j) & 1u) {                                                                                                \
                _Pragma("unroll") for (int i = 0; i < ROWS; i++)                                                                  \
                    if (km_ & (1u << i)) a[i][j] = eliminate(a[i][j], readlane_f64(kl_, i), p[j]);                                \
            }                                                                                                                     \
        }                                                                                                                         \
        }                                                                                                                         \
        if (wv == ((pc_p / CPT) >> 6)) { /* the pivot column itself: -k / quot (simplex.ts:386), one lane of this wave */         \
            const int ol_ = __builtin_amdgcn_readfirstlane((pc_p / CPT) & 63), js_ = __builtin_amdgcn_readfirstlane(pc_p % CPT);  \
            /* (NPUB builds: -k / quot is computed HERE, by the one wave that needs it, behind the summary -- round 6; the others: by wave 0 in step N) */ \
            const double nvl_ = lane < ROWS ? (NPUB ? -kl_ / quot_p : sm.nv[lane]) : 0.0;                                         \
            _Pragma("unroll") for (int j = 0; j < CPT; j++)                                                                       \
                if (js_ == j) {                                                                                                   \
                    _Pragma("unroll") for (int i = 0; i < ROWS; i++)                                                              \
                        if (km_ & (1u << i)) { const double nv_ = readlane_f64(nvl_, i); if (lane == ol_) a[i][j] = nv_; }        \
                }                                                                                                                 \
        }                                                                                                                         \
        {   /* the pivot row (simplex.ts:352-364): its owner only */                                                              \
            const int ip_ = __builtin_amdgcn_readfirstlane(pr_p - r_begin);                                                       \
            if (ip_ >= 0 && ip_ < ROWS) {                                                                                         \
                _Pragma("unroll") for (int i = 0; i < ROWS; i++)                                                                  \
                    if (i == JSLP_OPAQUE_SGPR(ip_)) { _Pragma("unroll") for (int j = 0; j < CPT; j++) a[i][j] = p[j]; }           \
            }                                                                                                                     \
        }                                                                                                                         \
        if (b == 0) { _Pragma("unroll") for (int j = 0; j < CPT; j++) a[0][j] = r0[j]; } /* workgroup 0 mirrors the cost row */   \
    } while (0)
// the row that can win, with the epoch tag INSIDE the data: {lo32 | tag}{hi32 | tag} per double (RCCL's LL scheme) -- the readers
// poll the row itself, no flag, no drain, no ordering between stores to rely on; twice the bytes, which one XCD's L2 does not notice
#define JSLP_XL_PUBLISH_ROW(ROW)                                                                                                  \
    do {                                                                                                                          \
        const int ipub_ = __builtin_amdgcn_readfirstlane((ROW) - r_begin);                                                        \
        if (colok) {                                                                                                              \
            _Pragma("unroll") for (int i = 0; i < ROWS; i++)                                                                      \
                if (i == JSLP_OPAQUE_SGPR(ipub_)) {                                                                                                 \
                    const int off_ = par * pub_stride + (b * ld + c0) * 16;                                                       \
                    _Pragma("unroll") for (int j = 0; j < CPT; j++) {                                                             \
                        if (c0 + j >= ld) continue;                                                                               \
                        const u64_t d_ = (u64_t)__double_as_longlong(a[i][j]);                                                    \
                        v4u_t v_;                                                                                                 \
                        v_.x = (unsigned)d_; v_.y = tag; v_.z = (unsigned)(d_ >> 32); v_.w = tag;                                 \
                        __builtin_amdgcn_raw_buffer_store_b128(v_, rsrc_rows, off_ + j * 16, 0, ST_AUX);                          \
                    }                                                                                                             \
                }                                                                                                                 \
        }                                                                                                                         \
    } while (0)
// the chip-wide builds' candidate row (untagged: the winner's flag follows its release): the same compare chain, the slot layout of SLOT / PERM
#define JSLP_PUBLISH_ROW_PLAIN(ROW)                                                                                               \
    do {                                                                                                                          \
        const int ipub_ = __builtin_amdgcn_readfirstlane((ROW) - r_begin);                                                        \
        /* (tests, JSLP_TEST_RESIDENT_LATE_WAVE0=2: wave 0 raises its flag word ~8 k cycles BEFORE it stores its part of the row -- \
           the order the fabric is allowed to produce; the readers' checksum must send them round again.  =4 / =6: a TORN row --    \
           the even (=4) / odd (=6) lanes of wave 0 store their pairs, the word goes up, and only ~8 k cycles later the other      \
           lanes store theirs: every reader's first looks find half of the wave's 1 KB new and half two epochs old) */             \
        const bool flag_first_ = CKS && __builtin_amdgcn_readfirstlane((int)(F_TEST_LATE == 2 && wv == 0)) != 0;                  \
        const bool torn_ = CKS && __builtin_amdgcn_readfirstlane((int)((F_TEST_LATE == 4 || F_TEST_LATE == 6) && wv == 0)) != 0;  \
        const bool early_lane_ = !torn_ || ((lane & 1) == (F_TEST_LATE == 6 ? 1 : 0));                                            \
        u64_t ck_ = 0;                                                                                                            \
        double cand_[CPT]; /* my columns of the row that can win (the compare chain: see JSLP_OPAQUE_SGPR) */                     \
        _Pragma("unroll") for (int j = 0; j < CPT; j++) cand_[j] = 0.0;                                                           \
        _Pragma("unroll") for (int i = 0; i < ROWS; i++)                                                                          \
            if (i == JSLP_OPAQUE_SGPR(ipub_)) { /* (each copy behind an opaque barrier: as plain selects the optimizer folds the chain into ONE   \
                dynamically indexed read of a[][] -- the whole register array went to scratch in the tall geometry, 600 B per lane, 23 k pivots/s) */ \
                _Pragma("unroll") for (int j = 0; j < CPT; j++) { double t_ = a[i][j]; asm volatile("" : "+v"(t_)); cand_[j] = t_; }               \
            }                                                                                                                     \
        double quotc_ = 0.0, nv0c_ = 0.0;                                                                                         \
        if (NPUB) {                                                                                                               \
            /* round 6: the row leaves NORMALISED (simplex.ts:352-364): should it win, quot is its entry of the entering column --  \
               which the ratio-test wave left in LDS for every row of mine (sm.colb: the value the update pass has just given the   \
               cell) -- so the two divisions per lane every reader used to run between the row fetch and the next pricing run       \
               HERE, once, while the summaries cross the fabric; quot itself and the cost row's new entry of the column            \
               (-k0 / quot, simplex.ts:386: every workgroup prices with the same cost row) travel in the row record */             \
            quotc_ = sm.colb[par][ipub_];                                                                                         \
            /* (the two divisions only ONE lane of the workgroup needs -- 1 / quot for the entering column's own cell, -k0 / quot for the    \
               reader lane that holds that column, which reads THIS wave's record -- run in the one wave that holds the column, behind a    \
               uniform branch: as per-lane selects all sixteen lock-step waves paid for them) */                                   \
            const bool holds_pc_ = __builtin_amdgcn_readfirstlane((int)(wv == ((pub_pc / CPT) >> 6))) != 0;                       \
            double rq_ = 0.0;                                                                                                     \
            if (holds_pc_) { rq_ = 1.0 / quotc_; nv0c_ = -pub_k0 / quotc_; }                                                      \
            _Pragma("unroll") for (int j = 0; j < CPT; j++) {                                                                     \
                const int col_ = c0 + j;                                                                                          \
                const double val_ = cand_[j];                                                                                     \
                double v_ = 0.0;                                                                                                  \
                if (col_ < W) {                                                                                                   \
                    const bool innz_ = nonzero16(val_);                                                                           \
                    v_ = innz_ ? val_ / quotc_ : 0.0;                                                                             \
                    if (col_ == pub_pc) v_ = rq_;                                                                                 \
                    if (innz_ && !nonzero16(v_) && v_ != 0.0) v_ = 0.0; /* (phase 2: some other row is always eliminated -- simplex.ts:381-383) */ \
                }                                                                                                                 \
                cand_[j] = v_;                                                                                                    \
            }                                                                                                                     \
        }                                                                                                                         \
        const int off_ = PERM ? par * pub_stride + b * SLOT + lane_off : par * pub_stride + (b * ld + c0) * 8;                    \
        if (colok) {                                                                                                              \
            _Pragma("unroll") for (int j = 0; j < CPT; j += 2) {                                                                  \
                if (c0 + j >= ld) continue;                                                                                       \
                const u64_t lo_ = (u64_t)__double_as_longlong(cand_[j]), hi_ = (u64_t)__double_as_longlong(cand_[j + 1]);         \
                v4u_t v_;                                                                                                         \
                v_.x = (unsigned)lo_; v_.y = (unsigned)(lo_ >> 32); v_.z = (unsigned)hi_; v_.w = (unsigned)(hi_ >> 32);           \
                if (!flag_first_ && early_lane_) __builtin_amdgcn_raw_buffer_store_b128(v_, rsrc_rows, off_ + (j >> 1) * PAIR_STEP, 0, ST_AUX); \
                if (CKS) JSLP_CK_PAIR(ck_, lo_, hi_, j);                                                                          \
            }                                                                                                                     \
        }                                                                                                                         \
        if (CKS) { if (NPUB) JSLP_CKS_RAISE_REC(ck_, quotc_, nv0c_); else JSLP_CKS_RAISE_FLAG(ck_); }                              \
        if (flag_first_ || torn_) {                                                                                               \
            __builtin_amdgcn_s_sleep(127);                                                                                        \
            if (colok && (flag_first_ || !early_lane_)) {                                                                         \
                _Pragma("unroll") for (int j = 0; j < CPT; j += 2) {                                                              \
                    if (c0 + j >= ld) continue;                                                                                   \
                    const u64_t lo_ = (u64_t)__double_as_longlong(cand_[j]), hi_ = (u64_t)__double_as_longlong(cand_[j + 1]);     \
                    v4u_t v_;                                                                                                     \
                    v_.x = (unsigned)lo_; v_.y = (unsigned)(lo_ >> 32); v_.z = (unsigned)hi_; v_.w = (unsigned)(hi_ >> 32);       \
                    __builtin_amdgcn_raw_buffer_store_b128(v_, rsrc_rows, off_ + (j >> 1) * PAIR_STEP, 0, ST_AUX);                \
                }                                                                                                                 \
            }                                                                                                                     \
        }                                                                                                                         \
    } while (0)
// (checksummed hand-over) this wave's word behind its stores of the row: no wait for their acknowledgement, no fence -- the reader checks
#define JSLP_CKS_RAISE_FLAG(CK)                                                                                                   \
    do {                                                                                                                          \
        const u64_t x_ = JSLP_CK_WAVE(CK) ^ JSLP_CK_TAGMIX(tag);                                                                  \
        if (lane == 0) AG_STORE(f.rowflagc[par] + wv * JSLP_F_MAXG + b, x_);                                                      \
    } while (0)
// round 6, NPUB builds: the 32-byte RECORD of (parity, wave, workgroup) instead of the 8-byte word: {checksum word, quot, -k0 / quot, 0}.  quot and
// the cost row's entry are folded into the checksum word (odd multipliers), so a record whose halves belong to different epochs fails like a torn row
#define JSLP_CK_QMIX(Q, NV) (((u64_t)__double_as_longlong(Q) * 0x9E3779B97F4A7C15ull) ^ ((u64_t)__double_as_longlong(NV) * 0xC2B2AE3D27D4EB4Full))
#define JSLP_REC_OFF(PAR, WV, B) (JSLP_R_REC_OFF + (((PAR) * JSLP_R_FLAGCOPIES + (WV)) * JSLP_F_MAXG + (B)) * JSLP_R_REC_BYTES)
#define JSLP_CKS_RAISE_REC(CK, Q, NV)                                                                                             \
    do {                                                                                                                          \
        const u64_t x_ = JSLP_CK_WAVE(CK) ^ JSLP_CK_TAGMIX(tag) ^ JSLP_CK_QMIX(Q, NV);                                            \
        if (lane == 0) {                                                                                                          \
            const u64_t qb_ = (u64_t)__double_as_longlong(Q), nb_ = (u64_t)__double_as_longlong(NV);                              \
            v4u_t r0_, r1_;                                                                                                       \
            r0_.x = (unsigned)x_; r0_.y = (unsigned)(x_ >> 32); r0_.z = (unsigned)qb_; r0_.w = (unsigned)(qb_ >> 32);             \
            r1_.x = (unsigned)nb_; r1_.y = (unsigned)(nb_ >> 32); r1_.z = tag; r1_.w = 0u;                                        \
            __builtin_amdgcn_raw_buffer_store_b128(r0_, rsrc_g16, JSLP_REC_OFF(par, wv, b), 0, ST_AUX);                           \
            __builtin_amdgcn_raw_buffer_store_b128(r1_, rsrc_g16, JSLP_REC_OFF(par, wv, b) + 16, 0, ST_AUX);                      \
        }                                                                                                                         \
    } while (0)
// ... and the fetch: flag word (NPUB: record) and row in ONE look, repeated until the checksum of what arrived matches the word
#define JSLP_CKS_ISSUE_LOOK()                                                                                                     \
    do {                                                                                                                          \
        if (__builtin_amdgcn_readfirstlane((int)((F_TEST_LATE & 1) != 0 && wv == 0))) __builtin_amdgcn_s_sleep(127);              \
        if (NPUB) { /* (one address per wave: the record, 32 bytes, in flight with the row) */                                    \
            rec0_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc_g16, JSLP_REC_OFF(par, wv, bw), 0, 16);                            \
            rec1_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc_g16, JSLP_REC_OFF(par, wv, bw) + 16, 0, 16);                       \
        } else {                                                                                                                  \
            flag_ = AG_LOAD(f.rowflagc[par] + wv * JSLP_F_MAXG + bw);                                                             \
        }                                                                                                                         \
    } while (0)
#define JSLP_CKS_FETCH_ROW()                                                                                                      \
    do {                                                                                                                          \
        efetch += 1;                                                                                                              \
        unsigned spins_ = 0;                                                                                                      \
        const u64_t tmix_ = JSLP_CK_TAGMIX(tag);                                                                                  \
        u64_t flag_ = 0;                                                                                                          \
        v4u_t rec0_, rec1_;                                                                                                       \
        rec0_.x = rec0_.y = rec0_.z = rec0_.w = 0u; rec1_ = rec0_;                                                                \
        for (;;) {                                                                                                                \
            JSLP_CKS_ISSUE_LOOK(); /* (ONE site: with a second copy of these loads at the loop's bottom the register allocator split the   \
                row array's live ranges around the loop and the tall geometry spilled 600 B per lane -- 125 k -> 23 k pivots/s, r06) */    \
            u64_t qm_ = 0;                                                                                                        \
            u64_t ck_ = 0;                                                                                                        \
            if (colok) {                                                                                                          \
                _Pragma("unroll") for (int j = 0; j < CPT; j += 2) {                                                              \
                    if (c0 + j >= ld) continue;                                                                                   \
                    const v4u_t v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc_rows, off_in + (j >> 1) * PAIR_STEP, 0, 16); /* (straight into  \
                        pv: staged through an array of their own these loads cost the tall geometry, at its register limit, 600 B of scratch per lane) */ \
                    const u64_t lo_ = (u64_t)v_.x | ((u64_t)v_.y << 32), hi_ = (u64_t)v_.z | ((u64_t)v_.w << 32);                 \
                    pv[j] = __longlong_as_double((long long)lo_);                                                                 \
                    pv[j + 1] = __longlong_as_double((long long)hi_);                                                             \
                    JSLP_CK_PAIR(ck_, lo_, hi_, j);                                                           \
                }                                                                                                                 \
            }                                                                                                                     \
            if (NPUB) {                                                                                                           \
                flag_ = (u64_t)rec0_.x | ((u64_t)rec0_.y << 32);                                                                  \
                fq = __longlong_as_double((long long)((u64_t)rec0_.z | ((u64_t)rec0_.w << 32)));                                  \
                fnv = __longlong_as_double((long long)((u64_t)rec1_.x | ((u64_t)rec1_.y << 32)));                                 \
                qm_ = JSLP_CK_QMIX(fq, fnv);                                                                                      \
            }                                                                                                                     \
            const u64_t x_ = JSLP_CK_WAVE(ck_) ^ tmix_ ^ qm_;                                                                     \
            if (__builtin_amdgcn_readfirstlane((int)(x_ == flag_))) break;                                                        \
            JSLP_RT_RETRY();                                                                                                      \
            if (lane == 0) atomicAdd(f.abort_flag + 1, 1u); /* (health counter: looks that found flag word and row disagreeing) */ \
            __builtin_amdgcn_s_sleep(1);                                                                                          \
            ++spins_;                                                                                                             \
            bool dead_ = false;                                                                                                   \
            if ((spins_ & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) dead_ = true;                                                 \
            if (spins_ > F_SPIN) { if (lane == 0) AG_STORE(f.abort_flag, 1u); dead_ = true; }                                     \
            if (dead_) { if (lane == 0) atomicMax(&sm.okbad, efetch); break; }                                                    \
        }                                                                                                                         \
    } while (0)
#ifdef JSLP_DEBUG_RESIDENT
#define JSLP_RT_RETRY() do { if (tid == 0) R.rt_retries += 1; } while (0)
#else
#define JSLP_RT_RETRY() do { } while (0)
#endif
// ... and its fetch: every lane polls ITS 16-byte words of the winner's row until both tags of each are this epoch's
#define JSLP_XL_FETCH_ROW(PCCOL, QUOT)                                                                                            \
    do {                                                                                                                          \
        efetch += 1;                                                                                                              \
        unsigned spins_ = 0;                                                                                                      \
        const int off_ = par * pub_stride + (bw * ld + c0) * 16;                                                                  \
        const int offq_ = par * pub_stride + (bw * ld + (PCCOL)) * 16; /* (PCCOL > 0: every lane also reads the row's entry of the \
            entering column -- quot, simplex.ts:333 -- one address per wave: no LDS broadcast, no barrier behind the fetch) */      \
        for (;;) {                                                                                                                \
            if (__builtin_amdgcn_readfirstlane((int)(F_TEST_LATE != 0 && wv == 0))) __builtin_amdgcn_s_sleep(127);          \
            bool ok_ = true;                                                                                                      \
            if (colok) {                                                                                                          \
                _Pragma("unroll") for (int j = 0; j < CPT; j++) {                                                                 \
                    if (c0 + j >= ld) continue;                                                                                   \
                    const v4u_t v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc_rows, off_ + j * 16, 0, 16);                      \
                    ok_ = ok_ && v_.y == tag && v_.w == tag;                                                                      \
                    pv[j] = __longlong_as_double((long long)((u64_t)v_.x | ((u64_t)v_.z << 32)));                                 \
                }                                                                                                                 \
            }                                                                                                                     \
            if ((PCCOL) > 0) {                                                                                                    \
                const v4u_t q_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc_rows, offq_, 0, 16);                                  \
                ok_ = ok_ && q_.y == tag && q_.w == tag;                                                                          \
                (QUOT) = __longlong_as_double((long long)((u64_t)q_.x | ((u64_t)q_.z << 32)));                                    \
            }                                                                                                                     \
            if (__all(ok_)) break;                                                                                                \
            __builtin_amdgcn_s_sleep(1);                                                                                          \
            ++spins_;                                                                                                             \
            bool dead_ = false;                                                                                                   \
            if ((spins_ & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) dead_ = true;                                                 \
            if (spins_ > F_SPIN) { if (lane == 0) AG_STORE(f.abort_flag, 1u); dead_ = true; }                               \
            if (dead_) { if (lane == 0) atomicMax(&sm.okbad, efetch); break; }                                                    \
        }                                                                                                                         \
    } while (0)

// round 6: the GLOBAL half of a pivot's basis change (simplex.ts:339-349: the four index maps in HBM, the pivot trace) is committed by ONE thread
// of workgroup 0 -- a dependent trip to memory (the eight pointers come from the device copy of the context) that used to sit between the row
// fetch and the next pricing: workgroup 0 reached every pricing barrier ~0.9 k cycles behind the other 250, and the whole chip waited for its
// summary in the gather (section table, r06_a: pricing 3.2 k cycles in workgroup 0 against 2.36 k everywhere else).  The LDS maps are still
// swapped where they were; the global stores follow in the NEXT iteration's update section (while the summaries cross the fabric) or behind
// the loop, as fire-and-forget stores through pointers kept in LDS.  Same stores, same order, one pivot later.
// (the tall OPT build commits in place, as round 5 did -- through the device copy of the context, in step N: see CM_AT_N)
#define JSLP_PIPE_COMMIT_GLOBAL_NOW()                                                                                             \
    do {                                                                                                                          \
        const Ctx& g_ = *f.cdev;                                                                                                  \
        g_.vibr[pr] = entering;                                                                                                   \
        g_.vibc[pc] = leaving;                                                                                                    \
        g_.rbv[entering] = pr;                                                                                                    \
        g_.rbv[leaving] = -1;                                                                                                     \
        g_.cbv[entering] = -1;                                                                                                    \
        g_.cbv[leaving] = pc;                                                                                                     \
        if (R.trace_n < g_.trace_cap) g_.trace[R.trace_n] = make_int2(pr, pc);                                                    \
    } while (0)
// (workgroup 0's global copy of the cycle-check history: thread 0's store through a pointer kept in LDS -- `f.cdev->hist[...]` was a dependent trip to memory in
//  front of the store, in the one workgroup everybody waits for: the cycle check cost 0.53 us per pivot of which this was the larger part, round 6)
#define JSLP_PIPE_HIST_GLOBAL(PAIR)                                                                                               \
    do {                                                                                                                          \
        jslp_gi32_t* const h_ = (jslp_gi32_t*)jslp_uniform_ptr(sm.gp_hist) + 2 * R.hist_n;                                        \
        h_[0] = (PAIR).x; h_[1] = (PAIR).y;                                                                                       \
    } while (0)
#define JSLP_PIPE_SWAP_LDS_MAPS()                                                                                                 \
    do {                                                                                                                          \
        if (lpend && tid == THREADS - 64) {                                                                                       \
            const int leaving_ = sm.lvibr[pr_p], entering_ = sm.lvibc[pc_p];                                                      \
            sm.lvibr[pr_p] = entering_;                                                                                           \
            sm.lvibc[pc_p] = leaving_;                                                                                            \
            if (b == 0) { sm.cm_ent = entering_; sm.cm_leav = leaving_; }                                                         \
        }                                                                                                                         \
        lpend = false;                                                                                                            \
    } while (0)
__device__ __forceinline__ void* jslp_uniform_ptr(const void* p) {  // the same pointer, as a wave-uniform (scalar) value
    const unsigned long long b = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
}
typedef __attribute__((address_space(1))) int32_t jslp_gi32_t;  // (a pointer into GLOBAL memory, as opposed to a generic one)
#define JSLP_PIPE_COMMIT_GLOBAL()                                                                                                 \
    do {                                                                                                                          \
        if (cpend && tid == THREADS - 64 && b == 0) {                                                                             \
            const int ent_ = sm.cm_ent, leav_ = sm.cm_leav; /* (pointers from LDS: nothing here waits for memory; GLOBAL stores, not flat \
                ones -- behind a flat access the compiler waits vmcnt(0) wherever it waits at all, which would undo the look-ahead poll) */ \
            /* (the base pointers as SCALARS -- readfirstlane of what LDS returned -- so that a store is `global_store v_offset, v_data, s[base]`:   \
                as seven 64-bit vector addresses the commit cost the tall OPT build, at its register limit, 550 B of scratch per lane) */ \
            jslp_gi32_t* const vibr_ = (jslp_gi32_t*)jslp_uniform_ptr(sm.gp_vibr);                                               \
            jslp_gi32_t* const vibc_ = (jslp_gi32_t*)jslp_uniform_ptr(sm.gp_vibc);                                               \
            jslp_gi32_t* const rbv_ = (jslp_gi32_t*)jslp_uniform_ptr(sm.gp_rbv);                                                 \
            jslp_gi32_t* const cbv_ = (jslp_gi32_t*)jslp_uniform_ptr(sm.gp_cbv);                                                 \
            vibr_[pr_p] = ent_;                                                                                                   \
            vibc_[pc_p] = leav_;                                                                                                  \
            rbv_[ent_] = pr_p;                                                                                                    \
            rbv_[leav_] = -1;                                                                                                     \
            cbv_[ent_] = -1;                                                                                                      \
            cbv_[leav_] = pc_p;                                                                                                   \
            if (R.trace_n - 1 < sm.gp_trace_cap) {                                                                                \
                jslp_gi32_t* const tr_ = (jslp_gi32_t*)jslp_uniform_ptr(sm.gp_trace) + 2 * (R.trace_n - 1);                       \
                tr_[0] = pr_p; tr_[1] = pc_p;                                                                                     \
            }                                                                                                                     \
        }                                                                                                                         \
        cpend = false;                                                                                                            \
    } while (0)

// CHK = false: a build without any of the cycle check's code (the host launches it when the check is off: with the test's loops
// inlined in the middle of the pivot loop the check-OFF solve ran 2 % slower)
// UNR (round 4): unrestricted variables in the lean loops -- a per-lane bit mask of the columns whose variable is unrestricted (R.unr,
// like the general build's); pricing takes |reduced cost| on them and hands isReducedCostNegative to the ratio test (simplex.ts:164-177,
// 282); phase 1 admits them whatever the sign of their coefficient (:56-71); the entering column inherits the LEAVING variable and
// with it that variable's flag (:339-349), read from the workgroup's LDS copy of the flags (sm.lunr: n_idx <= JSLP_R_LUNR -- the host
// sends larger models to the general build)
template <int THREADS, int CPT, int ROWS, bool OPT, bool CHK, bool XL = false, bool UNR = false>
__device__ __forceinline__ void resident_phase2_pipe(const ResCtx& f, RSmem& sm, ResRegs<CPT, ROWS>& R, int it1_start, int it2_start,
                                                     const int (&pb)[CPT]) {
    const Ctx& c = f.c;
    const int tid = threadIdx.x, b = XL ? (int)(blockIdx.x / JSLP_XL_SPREAD) : (int)blockIdx.x, lane = tid & 63, wv = tid >> 6;
    // XL (XCD-local build): all <= 32 workgroups sit on ONE XCD (checked at launch: k_simplex_resident's census), whose L2 is their
    // common point of coherence -- summaries, candidate rows and row flags leave as PLAIN stores (they stay in that L2; `sc1` stores
    // would drop the line and send the readers to memory) and are read with `sc1` loads (L1-bypassing, L2-served); no write-back
    // fence anywhere (tools/micro/xcd_handoff_bench.hip flavour 1: 1.9 k cycles per 32 -> 32 exchange against 8.3 k chip-wide)
    constexpr int ST_AUX = XL ? 0 : 16;           // aux of the hand-off stores: 16 = sc1 (write-through to memory)
    constexpr bool TAGGED = XL;                   // rows travel with their tags: 16 bytes per double
    constexpr bool CKS = !TAGGED && JSLP_PIPE_ROW_CHECKSUM != 0 && CPT <= JSLP_PIPE_ROW_CHECKSUM_MAXCPT;  // checksummed hand-over of the candidate rows (see JSLP_PIPE_ROW_CHECKSUM)
    // round 6 -- NPUB: the candidate row is published NORMALISED (JSLP_PUBLISH_ROW_PLAIN) and quot / the cost row's new entry of the entering column come
    // with the row's record: between the row fetch and the next pricing no division and no barrier are left (the update pass that publishes
    // is JSLP_XL_UPDATE_PASS + JSLP_PUBLISH_ROW_PLAIN: the 2- / 4-column geometries; optional objectives need the raw row for their tiny-entry rule)
    // (ROWS <= 8: the tall geometry <512, 4, 16> sits at its register limit -- with the record's loads and the normalisation's temporaries it spilled
    //  600 B per lane and ran 4001 x 2001 at 23 k pivots/s instead of 125 k; it keeps the raw row and round 4's 8-byte flag word)
    constexpr bool NPUB = CKS && !OPT && !XL && CPT <= 4 && ROWS <= 8 && JSLP_PIPE_NORM_PUB != 0;
    constexpr bool QDIRECT = TAGGED || NPUB;  // quot comes with the fetch: no barrier behind it
    // the ratio test's transposition (entry i of the entering column from the ONE lane that holds it to lane i) through LDS: as `x = lane
    // == i ? readlane(a[i][j]) : x` the compiler precomputes the 64-bit lane masks, spills them and pays two reloads, two moves and two
    // selects per row on top of the readlanes -- in the one wave the summary waits for
    constexpr bool S_LDS = XL || JSLP_PIPE_S_VIA_LDS != 0;
    // the pending pivot's row update in the XCD-local build's form (JSLP_XL_UPDATE_PASS: one ballot for the row gate, readlane multipliers,
    // the special rows fixed up once per pivot) instead of JSLP_PIPE_UPDATE_ROW's ~55 instructions per row
    constexpr bool UPD_NEW = XL || (CPT <= 4 && !(OPT && ROWS > 8));  // (the 6- / 8-column geometries and the tall OPT build would spill 14-94 VGPRs with it)
    // candidate rows in the publication buffer (chip-wide builds): pair j of lane t at ((j / 2) * THREADS + t) * 16 inside the
    // workgroup's slot -- a wave's store of one pair is 1 KB of whole lines (with lane t's CPT columns adjacent, as they sit in the
    // tableau, the 512-thread geometries wrote 16 bytes into each of 64 lines per instruction: the partial-line writers of round 3);
    // only lane t of the other workgroups ever reads what lane t wrote, so the permutation is invisible outside these two loops.
    // Measured (r04_q, pivots/s, adjacent -> permuted): 2001 x 4001 `<512,8,8>` 88.5 k -> 113.2 k, 3001 x 3001 `<512,6,12>` 102.0 k ->
    // 106.4 k, but 4001 x 2001 `<512,4,16>` 114.4 k -> 111.6 k (two adjacent pairs per lane are half a line already; apart they are
    // two requests): permuted from 6 columns per lane up
    constexpr bool PERM = CPT >= 6;
    // (the 2- and 4-column geometries keep round 3's addressing to the letter -- slots ld doubles apart: with the constant stride the
    //  tall `<512,4,16>` instance, at its register limit, came out 2 % slower, skewed or not)
    const int SLOT = PERM ? THREADS * CPT * 8 + JSLP_PUB_SKEW : f.c.ld * 8;
    constexpr int PAIR_STEP = PERM ? THREADS * 16 : 16;           // bytes from a lane's pair j to its pair j + 2
    const int lane_off = PERM ? tid * 16 : tid * CPT * 8;          // ... and where its first pair sits in the slot
    // the tall OPT build, at its register limit, issues the pending pivot's global commit where round 5 committed (step N: next to the row update's
    // temporaries the block cost it 550 B of scratch per lane); everybody else in the update section, off the critical path
    constexpr bool CM_AT_N = OPT && ROWS > 8;
    constexpr int POLLWV = 1;  // the ONE polling wave (not wave 0: it carries the column-0 work; not the last: the commit): lane l looks at workgroups l, l + 64, ...
    constexpr int HAWV = POLLWV;  // the wave whose retry path looks at the host's abort word (JSLP_HOST_ABORT_IN_SPIN)
    static_assert(THREADS / 64 > POLLWV, "the polling wave exists");
    const int ld = c.ld, W = c.W;
    const double precision = c.precision;
    const int c0 = tid * CPT;
    const bool colok = c0 < ld;
    const int r_begin = b * f.rpb, r_end = min(f.H, r_begin + f.rpb);
    double (&a)[ROWS][CPT] = R.a;
    double (&r0)[CPT] = R.r0;
    typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
    const int pub_stride = (int)((const char*)f.rows_pub[1] - (const char*)f.rows_pub[0]);  // (both carved from one arena, [0] first)
    const auto rsrc_rows = __builtin_amdgcn_make_buffer_rsrc(f.rows_pub[0], 0, pub_stride + (TAGGED ? f.G * ld * 16 : f.G * (CPT >= 6 ? THREADS * CPT * 8 + JSLP_PUB_SKEW : ld * 8)), 0x00020000);
    const auto rsrc_g16 = __builtin_amdgcn_make_buffer_rsrc(f.gran16, 0, JSLP_R_REC_OFF + JSLP_R_REC_WORDS * 8, 0x00020000);  // (the summary granules, the row-flag copies and the row records: one descriptor)
#ifdef JSLP_DEBUG_RESIDENT
    u64_t (&rt_acc)[8] = R.rt_acc;
    u64_t& rt_prev = R.rt_prev;
#endif

    // the pending pivot (its row update has not reached my registers yet)
    double p[CPT];      // its normalised pivot row, my columns
    unsigned nzm = 0;   // bit j: p[j] is non-zero by the reference's test (simplex.ts:379)
    int pr_p = 0, pc_p = 0, par_p = 0;  // its row, column and parity
    double quot_p = 1.0;                // its pivot element (NPUB builds: -k / quot of the pivot column's own entries is computed when needed)
    bool pend = false;
    bool cpend = false;  // the pending pivot's global commit has not been issued yet (JSLP_PIPE_COMMIT_GLOBAL)
    bool lpend = false;   // QDIRECT flows: the pending pivot's swap of my LDS maps has not happened yet (JSLP_PIPE_SWAP_LDS_MAPS)
#pragma unroll
    for (int j = 0; j < CPT; j++) p[j] = 0.0;
    int okslot = 0;
    unsigned& efetch = R.efetch;  // row fetches of this workgroup so far (uniform; one count for both phases: sm.okbad only grows)


    // the seen-pair filter of the cycle check covers a history that starts here (a phase entered with pairs already in it -- not
    // something the hosts do -- keeps testing every pivot)
    if (tid == 0) sm.cyc_filter_on = R.hist_n == 0 ? 1 : 0;  // (nothing of the cycle check stays in registers across the loop)
    for (int i = tid; i < JSLP_PIPE_CYCBITS / 32; i += THREADS) cyc_bits(sm)[i] = 0u;
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < ROWS; i++) sm.rhsb[i] = a[i][0];
        for (int q = 0; q < 2; q++) { sm.pw_batch[q] = 0x7fffffff; sm.pw_val[q] = 0; sm.pw_col[q] = 0x7fffffff; sm.pw_neg[q] = 0; }
    }
    __syncthreads();

    while (R.end_code == 0) {
        const bool has_pc_p = pend && colok && pc_p >= c0 && pc_p < c0 + CPT;
        // ---- exits that hand the tableau on (the pending update is applied behind the loop) ---------------------------------
        if ((R.it1 - it1_start) + (R.it2 - it2_start) >= F_ITERS) { R.end_code = 4; break; }
        if ((CHK && c.check_cycles) && !(R.hist_n < (f.hist_all ? JSLP_PIPE_GHIST : JSLP_PIPE_LHIST) && R.hist_n < c.hist_cap)) { R.end_code = 8; break; }  // history outgrows its room: the general kernel continues
        const unsigned epoch = R.epoch;
        const int par = epoch & 1;
        const unsigned tag = epoch + 1;
        if (F_TEST_ABORT >= 0 && (int)epoch == F_TEST_ABORT && b == f.G - 1) {  // tests: a workgroup gives up
            if (tid == 0) AG_STORE(f.abort_flag, 1u);
            R.end_code = 5;
            break;
        }
        RT_MARK(7);
        // ---- G: price the cost row -> entering column (simplex.ts:118-219; three LDS-atomic rounds) -------------------------------
        double k0 = 0.0;  // reduced cost of the entering column
        int pc;
        bool claim = false;  // (price_row_pipe: this wave holds the entering column and runs the ratio test)
        {
            int neg_now = 0;
            RT_STAMP(5);  // at the pricing
            pc = price_row_pipe<THREADS, CPT, UNR>(r0, c0, pb, c, sm, par, &k0, R.unr, &neg_now, &claim);
            RT_STAMP(6);  // priced
            JSLP_PIPE_SWAP_LDS_MAPS();  // (the pending pivot's basis change in my LDS maps: every wave is past its reads of them -- the pricing's barriers)
            if (UNR) R.neg = neg_now;  // isReducedCostNegative of the entering column (simplex.ts:164-177): the ratio test's sign
        }
        bool opt_enter = false;  // the entering column is named by an optional objective: its main cost is within +-precision
        if (OPT && pc == 0 && c.n_opt > 0) {
            pc = price_optional_regs<CPT>(r0, R.oo, c0, c, sm);
            if (pc != 0) {
                opt_enter = true;
                if (colok && pc >= c0 && pc < c0 + CPT) {
#pragma unroll
                    for (int j = 0; j < CPT; j++)
                        if (pc == c0 + j) sm.xq[0] = r0[j];
                }
                __syncthreads();
                k0 = sm.xq[0];
            }
        }
        if (QDIRECT && !OPT && efetch != 0u && sm.okbad == efetch) { R.end_code = 5; pend = false; break; }  // (a wave gave up in the previous pivot's row fetch: its registers hold a stale row)
        if (pc == 0) { R.end_code = 1; break; }  // uniform: optimal (simplex.ts:265-269)
        RT_MARK(6);
        // ---- S: ratio test for column pc (simplex.ts:271-296) by the wave that holds the column: the ONE lane that holds it hands
        //      entry i to lane i (readlanes), lanes 0..ROWS-1 apply the pending pivot to their entry and classify their row in
        //      parallel (one division each), DPP reductions fold the verdicts -- no LDS round trip, no workgroup barrier inside ------
        if (OPT && opt_enter) claim = wv == ((pc / CPT) >> 6);  // (an entering column named by an optional objective: every thread knows it)
        if (claim) {
            const int ol = __builtin_amdgcn_readfirstlane((pc / CPT) & 63);  // the lane that holds column pc
            const int jsel = __builtin_amdgcn_readfirstlane(pc % CPT);        // ... as its column jsel
            double x = 0.0, pj = 0.0;
            unsigned nzj = 0;
#pragma unroll
            for (int j = 0; j < CPT; j++)
                if (jsel == j) {  // uniform
                    if (S_LDS) {  // the holding lane lays its entries out in LDS (one 8-byte store per row, one lane) instead of readlanes
                        if (lane == ol) {
#pragma unroll
                            for (int i = 0; i < ROWS; i++) sm.quo[i] = a[i][j];
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < ROWS; i++) {
                            const double xi = readlane_f64(a[i][j], ol);
                            x = lane == i ? xi : x;
                        }
                    }
                    pj = readlane_f64(p[j], ol);
                    nzj = ((unsigned)__builtin_amdgcn_readlane((int)nzm, ol) >> j) & 1u;
                }
            if (S_LDS) {
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS stores
                x = lane < ROWS ? sm.quo[lane] : 0.0;
            }
            const int r = r_begin + lane;
            double colv = 0.0;
            int kind = 0;  // 0 skip, 1 degenerate winner, 2 quotient candidate
            double quo = 0.0;
            if (lane < ROWS) {
                colv = x;
                if (pend) {  // what the pending pivot makes of this cell (the update pass will compute the same)
                    const double ki = sm.colb[par_p][lane];
                    if (r == pr_p) colv = pj;
                    else if (r != 0 && r < r_end && nonzero16(ki)) {
                        if (__builtin_amdgcn_readfirstlane(pc_p) == __builtin_amdgcn_readfirstlane(pc)) colv = NPUB ? -ki / quot_p : sm.nv[lane];  // (uniform, rare: the column that has just left enters again -- simplex.ts:386)
                        else colv = nzj ? eliminate(x, ki, pj) : x;
                    }
                }
                sm.colb[par][lane] = colv;  // every thread's update pass of THIS pivot reads it (after the barrier below)
                const double rhs = sm.rhsb[lane];
                if (r >= 1 && r < r_end && !(-precision < colv && colv < precision)) {
                    if (colv > 0 && precision > rhs && rhs > -precision) kind = 1;
                    else { quo = (UNR && R.neg) ? -rhs / colv : rhs / colv; kind = quo > precision ? 2 : 0; }  // simplex.ts:282
                }
            }
            int brdeg = kind == 1 ? r : 0x7fffffff;  // rows ascend with the lane: the smallest row is the first one
            brdeg = min(brdeg, __builtin_amdgcn_update_dpp(brdeg, brdeg, 0xB1, 0xf, 0xf, false));
            brdeg = min(brdeg, __builtin_amdgcn_update_dpp(brdeg, brdeg, 0x4E, 0xf, 0xf, false));
            brdeg = min(brdeg, __builtin_amdgcn_update_dpp(brdeg, brdeg, 0x141, 0xf, 0xf, false));
            if (ROWS > 8) brdeg = min(brdeg, __builtin_amdgcn_update_dpp(brdeg, brdeg, 0x140, 0xf, 0xf, false));  // (<= 8 rows: lanes 0..7 hold them, three exchanges fold them into lane 0)
            brdeg = ROWS > 16 ? min(__builtin_amdgcn_readlane(brdeg, 0), __builtin_amdgcn_readlane(brdeg, 16))
                              : __builtin_amdgcn_readlane(brdeg, 0);  // (ROWS <= 16: the candidates sit in the first 16-lane row; <= 32: in the first two)
            static_assert(ROWS <= 32, "the summary's DPP folds cover two 16-lane rows");
            KI bk;  // quotients are > precision > 0: positive doubles order like their bit patterns; ties -> first row
            bk.k = kind == 2 ? (u64_t)__double_as_longlong(quo) : KI_NONE_KEY;
            bk.i = kind == 2 ? r : 0x7fffffff;
            bk.pad = 0;
            bk = ki_min(bk, ki_dpp<0xB1>(bk));
            bk = ki_min(bk, ki_dpp<0x4E>(bk));
            bk = ki_min(bk, ki_dpp<0x141>(bk));
            if (ROWS > 8) bk = ki_min(bk, ki_dpp<0x140>(bk));
            bk = ROWS > 16 ? ki_min(ki_readlane(bk, 0), ki_readlane(bk, 16)) : ki_readlane(bk, 0);
            if (lane == 0) {
                const bool deg = brdeg != 0x7fffffff;
                const bool have = bk.k != KI_NONE_KEY;
                const int row = deg ? brdeg : (have ? bk.i : 0);  // the only row of mine that can win (0: none)
                const u64_t qb = (deg || !have) ? 0ull : bk.k;
                v4u_t g;
                g.x = (unsigned)qb;
                g.y = tag;
                g.z = (unsigned)(qb >> 32);
                g.w = ((tag & 0xffffu) << 16) | (deg ? 0x8000u : 0u) | (unsigned)row;
                __builtin_amdgcn_raw_buffer_store_b128(g, rsrc_g16, (par * JSLP_F_MAXG + b) * JSLP_G16_STRIDE, 0, ST_AUX);
                sm.pubrow = row;
            }
        }
        __syncthreads();
        const int pubrow = sm.pubrow;
        if (!CM_AT_N) JSLP_PIPE_COMMIT_GLOBAL();  // (the pending pivot's global maps + trace, while the summaries cross the fabric)
        RT_STAMP(0);  // summary stored (the claiming wave stored it in front of the barrier)
        const double pub_k0 = k0;
        const int pub_pc = pc;
        (void)pub_k0; (void)pub_pc;
        RT_MARK(0);
        // ---- U + P: the pending pivot's row update (simplex.ts:367-391), ONE pass over my rows; the row that can win is
        //         published (16-byte write-through stores) as soon as it is up to date.  The summaries are crossing the fabric
        //         meanwhile ----------------------------------------------------------------------------------------------------------
        bool swept = true;
        const bool poller = wv == POLLWV;
        v4u_t g;
        g.x = 0; g.y = tag; g.z = 0; g.w = (tag & 0xffffu) << 16;  // workgroups beyond the grid: "no candidate"
        // (the pending pivot's column entries of my rows: broadcast LDS reads, JSLP_PIPE_KCHUNK of them in flight together)
        if (UPD_NEW) {
            if (pend) JSLP_XL_UPDATE_PASS();
            if (pubrow != 0) {
                if (TAGGED) JSLP_XL_PUBLISH_ROW(pubrow);
                else JSLP_PUBLISH_ROW_PLAIN(pubrow);
            }
        } else {
        u64_t ck = 0;
        // (tests: the same hooks as JSLP_PUBLISH_ROW_PLAIN -- flag word first (=2), torn row (=4 / =6) -- for the builds that publish from inside this pass)
        const bool flag_first_o = CKS && __builtin_amdgcn_readfirstlane((int)(F_TEST_LATE == 2 && wv == 0)) != 0;
        const bool torn_o = CKS && __builtin_amdgcn_readfirstlane((int)((F_TEST_LATE == 4 || F_TEST_LATE == 6) && wv == 0)) != 0;
        const bool early_lane_o = !torn_o || ((lane & 1) == (F_TEST_LATE == 6 ? 1 : 0));
#pragma unroll
        for (int i0 = 0; i0 < ROWS; i0 += JSLP_PIPE_KCHUNK) {
            double kis[ROWS];
#pragma unroll
            for (int i = i0; i < i0 + JSLP_PIPE_KCHUNK && i < ROWS; i++) kis[i] = sm.colb[par_p][i];
#pragma unroll
            for (int i = i0; i < i0 + JSLP_PIPE_KCHUNK && i < ROWS; i++) {
                if (pend) JSLP_PIPE_UPDATE_ROW(i);
                if (pubrow != 0 && r_begin + i == pubrow && colok) {  // (uniform but for colok)
                    const int off = PERM ? par * pub_stride + b * SLOT + lane_off : par * pub_stride + (b * ld + c0) * 8;
#pragma unroll
                    for (int j = 0; j < CPT; j += 2) {
                        if (c0 + j >= ld) continue;
                        const u64_t lo = (u64_t)__double_as_longlong(a[i][j]), hi = (u64_t)__double_as_longlong(a[i][j + 1]);
                        v4u_t v;
                        v.x = (unsigned)lo; v.y = (unsigned)(lo >> 32); v.z = (unsigned)hi; v.w = (unsigned)(hi >> 32);
                        if (!flag_first_o && early_lane_o) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_rows, off + (j >> 1) * PAIR_STEP, 0, ST_AUX);
                        if (CKS) JSLP_CK_PAIR(ck, lo, hi, j);
                    }
                }
            }
        }
        if (CKS && pubrow != 0) JSLP_CKS_RAISE_FLAG(ck);
        if (CKS && pubrow != 0 && (flag_first_o || torn_o)) {  // (tests only) what the hook held back leaves now, behind the flag word
            __builtin_amdgcn_s_sleep(127);
            const int ipub_o = __builtin_amdgcn_readfirstlane(pubrow - r_begin);
            if (colok && (flag_first_o || !early_lane_o)) {
#pragma unroll
                for (int i = 0; i < ROWS; i++)
                    if (i == JSLP_OPAQUE_SGPR(ipub_o)) {
                        const int off = PERM ? par * pub_stride + b * SLOT + lane_off : par * pub_stride + (b * ld + c0) * 8;
#pragma unroll
                        for (int j = 0; j < CPT; j += 2) {
                            if (c0 + j >= ld) continue;
                            const u64_t lo = (u64_t)__double_as_longlong(a[i][j]), hi = (u64_t)__double_as_longlong(a[i][j + 1]);
                            v4u_t v;
                            v.x = (unsigned)lo; v.y = (unsigned)(lo >> 32); v.z = (unsigned)hi; v.w = (unsigned)(hi >> 32);
                            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_rows, off + (j >> 1) * PAIR_STEP, 0, ST_AUX);
                        }
                    }
            }
        }
        }
        pend = false;
        RT_STAMP(1);  // update + publication issued
        RT_MARK(2);
        // ---- C: gather: ONE wave, lane l looks at the granules of workgroups l, l + 64, l + 128, l + 192 (four 16-byte loads in flight) --------
        if (poller) {
            unsigned spins = 0;
            v4u_t gq[JSLP_F_MAXG / 64];
#if JSLP_PIPE_POLL_MISSING_ONLY
            unsigned have = 0;  // bit q: my look q has its summary (a look beyond the grid holds a copy of my own workgroup's: it is there)
#pragma unroll
            for (int q = 0; q < JSLP_F_MAXG / 64; q++) {
                gq[q] = g;
                if (lane + 64 * q >= f.G) have |= 1u << q;
            }
#endif
            for (;;) {
                bool ok = true;
#if JSLP_PIPE_POLL_MISSING_ONLY
                // Round 6: only the looks whose summary has NOT arrived go out again.  Most summaries are there at the first look, and a look is an sc1 load
                // that memory serves: 251 workgroups x 251 granules per round trip otherwise -- traffic that delays the very stores it waits for
                // (184.7 k -> 193.3 k pivots/s on config 3a; two staggered sets of looks, i.e. MORE traffic, measured 158 k: profiles/r06_poll_traffic.md)
#pragma unroll
                for (int q = 0; q < JSLP_F_MAXG / 64; q++)
                    if (!((have >> q) & 1u)) gq[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_g16, (par * JSLP_F_MAXG + lane + 64 * q) * JSLP_G16_STRIDE, 0, 16);
#pragma unroll
                for (int q = 0; q < JSLP_F_MAXG / 64; q++)
                    if (gq[q].y == tag && (gq[q].w >> 16) == (tag & 0xffffu)) have |= 1u << q;
                ok = have == (1u << (JSLP_F_MAXG / 64)) - 1u;
#else
#pragma unroll
                for (int q = 0; q < JSLP_F_MAXG / 64; q++) {
                    gq[q] = g;
                    if (lane + 64 * q < f.G) gq[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_g16, (par * JSLP_F_MAXG + lane + 64 * q) * JSLP_G16_STRIDE, 0, 16);
                }
#pragma unroll
                for (int q = 0; q < JSLP_F_MAXG / 64; q++) ok = ok && gq[q].y == tag && (gq[q].w >> 16) == (tag & 0xffffu);
#endif
                if (__all(ok) && !JSLP_HOST_ABORT_FORCE_RETRY(spins)) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                JSLP_HOST_ABORT_IN_SPIN(spins, swept);
                if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) { swept = false; break; }
                if (spins > F_SPIN) { if (lane == 0) AG_STORE(f.abort_flag, 1u); swept = false; break; }
            }
            // my four workgroups' summaries -> mine -> the wave's: first degenerate row, else smallest quotient (first row on ties)
            int rdeg = 0x7fffffff;
            KI x = ki_none();
#pragma unroll
            for (int q = 0; q < JSLP_F_MAXG / 64; q++) {
                const int row = (int)(gq[q].w & 0x7fffu);
                const bool deg = (gq[q].w & 0x8000u) != 0u;
                if (deg && row != 0) rdeg = min(rdeg, row);
                KI y;
                const bool cand = !deg && row != 0;
                y.k = cand ? ((u64_t)gq[q].x | ((u64_t)gq[q].z << 32)) : KI_NONE_KEY;
                y.i = cand ? row : 0x7fffffff;
                y.pad = 0;
                x = ki_min(x, y);
            }
            rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0xB1, 0xf, 0xf, false));
            rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0x4E, 0xf, 0xf, false));
            rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0x141, 0xf, 0xf, false));
            rdeg = min(rdeg, __builtin_amdgcn_update_dpp(rdeg, rdeg, 0x140, 0xf, 0xf, false));
            rdeg = min(min(__builtin_amdgcn_readlane(rdeg, 0), __builtin_amdgcn_readlane(rdeg, 16)),
                       min(__builtin_amdgcn_readlane(rdeg, 32), __builtin_amdgcn_readlane(rdeg, 48)));
            x = ki_wave_min(x);
            if (lane == 0) { sm.part_k[0] = x.k; sm.part_r[0] = x.k == KI_NONE_KEY ? 0 : x.i; sm.part_rdeg[0] = rdeg; }
        }
        RT_STAMP(2);  // my wave's 64 summaries are in
        RT_MARK(1);
        if (!TAGGED && !CKS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: my stores of the candidate row have reached the L2 (the winner's release below builds on it; XL / winner-only: the row carries its tags)
        const int all_swept = __syncthreads_and(swept ? 1 : 0);
        if (!all_swept) { R.end_code = 5; break; }
        RT_STAMP(3);  // gather closed
        RT_MARK(3);
        // ---- D: every thread reads the polling wave's verdict ------------------------------------------------------------------
        int pr = 0, stop = 0;
        {
            const int wr = sm.part_r[0], wrdeg = sm.part_rdeg[0];
            if (wrdeg != 0x7fffffff) pr = wrdeg;
            else if (wr != 0) pr = wr;
            else stop = 3;  // unbounded (simplex.ts:298-303)
        }
        if (!stop && (CHK && c.check_cycles)) {  // simplex.ts:305-320 by every workgroup, on its own LDS history
            if (tid == 0) {
                const int2 pair = make_int2(sm.lvibr[pr], sm.lvibc[pc]);
                if (R.hist_n < JSLP_PIPE_LHIST) sm.lhist[R.hist_n] = pair;
                if (f.hist_all) f.hist_all[(size_t)b * JSLP_PIPE_GHIST + R.hist_n] = pair;  // my own copy of the whole history
                if (b == 0) JSLP_PIPE_HIST_GLOBAL(pair);  // the host's cycle message; the general kernel's history should this one outgrow its room
                sm.cyc_need = sm.cyc_filter_on ? cyc_pair_seen(sm, pair) : 1;
            }
            __syncthreads();
            R.hist_n += 1;
            if (sm.cyc_need != 0) {  // (uniform) only a pair that occurred before can end a repeated block
                if (cyc_suffix_is_square(sm.lhist, f.hist_all ? f.hist_all + (size_t)b * JSLP_PIPE_GHIST : nullptr, R.hist_n, make_int2(sm.lvibr[pr], sm.lvibc[pc]), sm.f.red)) stop = 1;
            }
        }
        if (stop == 3) { R.end_code = 2; R.unbounded_col = pc; break; }
        if (stop == 1) { R.end_code = 3; break; }
        // the entering column inherits the LEAVING variable (simplex.ts:339-349) and with it its "unrestricted" flag; the maps still
        // hold this pivot's leaving variable (the commit at the end of the iteration is a barrier away)
        const bool leaving_unr = UNR && sm.lunr[sm.lvibr[pr]] != 0;
        // ---- the winner releases its row: ONLY the workgroup that holds row pr raises row flags, behind a real agent-scope release.
        //      The acknowledgement of a write-through (sc1) store does NOT mean the data has reached memory -- only the XCD's L2:
        //      with `s_waitcnt vmcnt(0)` alone in front of the flag stores, other XCDs saw the flag before the row about once in 10^5
        //      pivots on the tall / wide shapes, whose lanes write partial lines (tools/resident_stress.py: 601 x 3001 wrong in 1 of
        //      40 solves, 4001 x 2001 in 1 of 4, grid time-outs when the replicated cost rows diverged).  `buffer_wbl2 sc1` (what the
        //      fence emits) closes that.  Issued by every wave of every workgroup it cost 30 us per pivot (the L2 serialises them);
        //      by one wave of every publishing workgroup at the end of the gather 109 k pivots/s; by one wave of the ONE workgroup
        //      whose row is going to be read, here, 139 k -- 149 k with the drain in front of the barrier that closes the gather instead of
        //      a barrier of its own here (155 k with the unsound release) -------------------------------------------------------------
        const int bw = pr / f.rpb;
        if (!TAGGED && !CKS && bw == b) {  // (uniform: pr is the row I published -- my candidate was the chip's best; XL: tags inside the row, no flag)
            if (tid < THREADS / 64) {  // (every wave's stores reached the L2 before the barrier that closed the gather)
                if (XL) {  // ... which is where every reader looks: a plain flag store behind the drained row stores is the whole release
                    *reinterpret_cast<volatile u64_t*>(f.rowflagc[par] + tid * JSLP_F_MAXG + b) = (u64_t)tag;
                } else {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // buffer_wbl2 sc1 + s_waitcnt vmcnt(0): ... and memory
                    AG_STORE(f.rowflagc[par] + tid * JSLP_F_MAXG + b, (u64_t)tag);  // one copy of the flag per fetching wave
                }
            }
        }
        // ---- E: the winning row: every wave waits for its copy of the flag and loads its columns; the lane that holds column pc
        //         broadcasts quot = A[pr, pc] --------------------------------------------------------------------------------------
        const int off_in = PERM ? par * pub_stride + bw * SLOT + lane_off : par * pub_stride + (bw * ld + c0) * 8;
        const bool has_pc = colok && pc >= c0 && pc < c0 + CPT;
        double pv[CPT];
#pragma unroll
        for (int j = 0; j < CPT; j++) pv[j] = 0.0;
        double quot = 0.0;
        double fq = 0.0, fnv = 0.0;  // NPUB: quot and -k0 / quot as they came with the row's record
        double ook[JSLP_R_MAXOPT];  // the optional objectives' entries of column pc (OPT builds)
#pragma unroll
        for (int o = 0; o < JSLP_R_MAXOPT; o++) ook[o] = 0.0;
        {
            // EVERY wave waits for its own copy of the flag, then loads its columns of the row
            // (see resident_phase's step E); a wave that gives up raises sm.okbad to this fetch's number
            if (TAGGED) {
                JSLP_XL_FETCH_ROW(pc, quot);
            } else if (CKS) {
                JSLP_CKS_FETCH_ROW();
            } else {
            efetch += 1;
            unsigned spins = 0;
            for (;;) {
                if (__builtin_amdgcn_readfirstlane((int)(F_TEST_LATE != 0 && wv == 0))) __builtin_amdgcn_s_sleep(127);  // (tests: the skew that used to break the fetch; a scalar branch: s_sleep ignores EXEC)
                const u64_t flag = AG_LOAD(f.rowflagc[par] + wv * JSLP_F_MAXG + bw);
                if ((unsigned)flag == tag) break;  // (wave-uniform: one word, one request)
#ifdef JSLP_DEBUG_RESIDENT
                if (tid == 0) R.rt_retries += 1;
#endif
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                bool dead = false;
                if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) dead = true;
                if (spins > F_SPIN) { if (lane == 0) AG_STORE(f.abort_flag, 1u); dead = true; }
                if (dead) { if (lane == 0) atomicMax(&sm.okbad, efetch); break; }
            }
            asm volatile("" ::: "memory");
            if (colok) {  // the row, now that MY look at the flag found it up (the flag comes late by construction -- only the winner raises
                          // it, after the decision --, so loading the row next to it would only add 4 MB of dead traffic per look)
#pragma unroll
                for (int j = 0; j < CPT; j += 2) {
                    if (c0 + j >= ld) continue;
                    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_rows, off_in + (j >> 1) * PAIR_STEP, 0, 16);
                    pv[j] = __longlong_as_double((long long)((u64_t)v.x | ((u64_t)v.y << 32)));
                    pv[j + 1] = __longlong_as_double((long long)((u64_t)v.z | ((u64_t)v.w << 32)));
                }
            }
            }  // !XL
            if (NPUB) quot = fq;
            if (!NPUB && has_pc) {
#pragma unroll
                for (int j = 0; j < CPT; j++)
                    if (pc == c0 + j) {
                        sm.xq2[okslot] = pv[j];
                        if (OPT) {
#pragma unroll
                            for (int o = 0; o < JSLP_R_MAXOPT; o++) sm.ook[okslot][o] = R.oo[o][j];
                        }
                    }
            }
            if (!(QDIRECT && !OPT)) {  // (XL / NPUB: quot came with the fetch, and a wave that gave up is noticed behind the next pricing's first barrier)
            __syncthreads();
            if (sm.okbad == efetch) R.end_code = 5;
            quot = sm.xq2[okslot];
            }
            if (OPT) {
#pragma unroll
                for (int o = 0; o < JSLP_R_MAXOPT; o++) ook[o] = sm.ook[okslot][o];
            }
            okslot ^= 1;  // the next use writes the other words: one barrier per use
        }
        if (R.end_code == 5) break;
        RT_STAMP(4);  // row fetched
        RT_MARK(4);
        // ---- N: normalised pivot row (simplex.ts:352-364; phase 2: some other row is always eliminated, so the tiny entries
        //         simplex.ts:381-383 zeroes are zero) -----------------------------------------------------------------------------
        nzm = 0;
        int tiny = 0;
        if (NPUB) {  // the row arrived normalised (lanes beyond ld fetched nothing: zeros)
#pragma unroll
            for (int j = 0; j < CPT; j++) p[j] = pv[j];
        } else if (colok) {
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const int col = c0 + j;
                const double val = pv[j];
                double v = 0.0;
                if (col < W) {
                    const bool innz = nonzero16(val);
                    v = innz ? val / quot : 0.0;
                    if (col == pc) v = 1.0 / quot;  // (as a real branch -- one lane of the workgroup needs this division -- measured no faster: r04_u)
                    if (innz && !nonzero16(v) && v != 0.0) {
                        if (OPT && opt_enter) tiny |= 1 << j;  // (decided below)
                        else v = 0.0;
                    }
                }
                p[j] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < CPT; j++) p[j] = 0.0;
        }
        if (OPT && opt_enter) {
            // an entering column named by an optional objective may have NO other row to eliminate (its main cost is ~0 too): the
            // tiny entries are zeroed only if some other row has an entry in the column (simplex.ts:381-383) -- a chip-wide answer
            if (__syncthreads_or(tiny)) {
                int local_any = 0;
#pragma unroll
                for (int i = 0; i < ROWS; i++) {
                    const int r = r_begin + i;
                    if (r >= 1 && r < r_end && r != pr && nonzero16(sm.colb[par][i])) local_any = 1;
                }
                if (b == 0 && nonzero16(k0)) local_any = 1;  // (row 0, the cost row: simplex.ts:367 runs r from 0)
                const int gany = global_or(f, par, tag, local_any, sm, b);
                if (gany < 0) { R.end_code = 5; break; }
                if (gany != 0) {
#pragma unroll
                    for (int j = 0; j < CPT; j++)
                        if (tiny & (1 << j)) p[j] = 0.0;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CPT; j++) nzm |= nonzero16(p[j]) ? (1u << j) : 0u;
        if (OPT) {  // the optional objective rows (simplex.ts:394-412: exact `!== 0` tests, on the FINAL pivot row), every lane its columns
#pragma unroll
            for (int o = 0; o < JSLP_R_MAXOPT; o++)
                if (o < c.n_opt) {
#pragma unroll
                    for (int j = 0; j < CPT; j++)
                        if (c0 + j < W) R.oo[o][j] = oo_cell_after(R.oo[o][j], ook[o], c0 + j, pc, quot, p[j]);
                }
        }
        // the pivot column's own new entries (-k / quot, simplex.ts:386) and column 0 of my rows after this pivot (sm.rhsb is
        // the ratio test's copy of that column: it receives what the update pass will give a[i][0]): lanes 0..ROWS-1 of wave 0
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
                if (!NPUB) sm.nv[lane] = -ki / quot;  // (NPUB: computed where it is needed -- the update pass, the rare re-entering column)
            }
        }
        // ---- R0: the cost row (every workgroup its own copy) -----------------------------------------------------------------------
        if (nonzero16(k0)) {
#pragma unroll
            for (int j = 0; j < CPT; j++)
                if ((nzm >> j) & 1u) r0[j] = eliminate(r0[j], k0, p[j]);
            if (has_pc) {
                const double nv0 = NPUB ? fnv : -k0 / quot;  // (NPUB: the publisher's division, same operands -- every workgroup holds the same cost row)
#pragma unroll
                for (int j = 0; j < CPT; j++)
                    if (pc == c0 + j) r0[j] = nv0;
            }
        }
        // ---- commit the basis change (simplex.ts:339-349): every workgroup's LDS maps, workgroup 0 the global ones ---------------------
        // (QDIRECT flows have no barrier between the row fetch and this point: a wave that is through with its fetch must not swap the maps
        //  while a slower one is still reading them behind the gather -- the leaving variable's "unrestricted" flag, the cycle check's pair.
        //  There the swap waits for the next pricing's barriers: JSLP_PIPE_SWAP_LDS_MAPS.  Found by the chaos build, round 6.)
        if (!QDIRECT && tid == THREADS - 64) {  // (not thread 0: its wave carries the column-0 work above)
            const int leaving = sm.lvibr[pr], entering = sm.lvibc[pc];
            sm.lvibr[pr] = entering;
            sm.lvibc[pc] = leaving;
            if (b == 0) {
                if (CM_AT_N) JSLP_PIPE_COMMIT_GLOBAL_NOW();
                else { sm.cm_ent = entering; sm.cm_leav = leaving; }  // (the global maps and the trace follow one iteration later: JSLP_PIPE_COMMIT_GLOBAL)
            }
        }
        lpend = QDIRECT;
        if (UNR && has_pc) {
#pragma unroll
            for (int j = 0; j < CPT; j++)
                if (pc == c0 + j) R.unr = (R.unr & ~(1u << j)) | ((leaving_unr ? 1u : 0u) << j);
        }
        R.trace_n += 1;
        R.it2 += 1;
        R.epoch = epoch + 1;
        pend = true; pr_p = pr; pc_p = pc; par_p = par; quot_p = quot;
        cpend = !CM_AT_N;
        RT_MARK(5);
    }
    JSLP_PIPE_SWAP_LDS_MAPS();  // (an exit in front of the pricing)
    JSLP_PIPE_COMMIT_GLOBAL();  // (a basis change whose global half is still pending)
    if (pend && R.end_code != 5) {  // whoever leaves with a pivot pending (optimal, iteration cap, hand-over) brings the rows up to date
        const bool has_pc_p = colok && pc_p >= c0 && pc_p < c0 + CPT;
        if (UPD_NEW) {
            (void)has_pc_p;
            JSLP_XL_UPDATE_PASS();
        } else
#pragma unroll
        for (int i0 = 0; i0 < ROWS; i0 += JSLP_PIPE_KCHUNK) {
            double kis[ROWS];
#pragma unroll
            for (int i = i0; i < i0 + JSLP_PIPE_KCHUNK && i < ROWS; i++) kis[i] = sm.colb[par_p][i];
#pragma unroll
            for (int i = i0; i < i0 + JSLP_PIPE_KCHUNK && i < ROWS; i++) JSLP_PIPE_UPDATE_ROW(i);
        }
    }
}


// ===================================================================================================================
// Phase 1 (simplex.ts:25-98) in the same pipelined form.  Per pivot: leaving row = most negative RHS below -precision (first
// index on ties), entering column = max -cost / coefficient over the pivot row's entries below -precision (first index on
// ties), cycle check, pivot.  The summary a workgroup publishes needs only column 0 of its rows, whose master copy lives in
// LDS: the eight lanes that bring it up to date when a pivot is decided (`rhs - k * p0`) fold the next summary in the same
// breath, so pivot t+1's all-gather starts before pivot t's row update, which then overlaps it as in phase 2.  The entering
// column is a workgroup-wide maximum: one DPP maximum per wave + one LDS atomic per wave on an order-preserving key, then an
// atomicMin on the column among the lanes that hold that value.  Returns with R.end_code == 0 when phase 1 is over (feasible);
// the tableau is whole again then.
// ===================================================================================================================
template <int THREADS, int CPT, int ROWS, bool OPT, bool CHK, bool XL = false, bool UNR = false>
__device__ __forceinline__ void resident_phase1_pipe(const ResCtx& f, RSmem& sm, ResRegs<CPT, ROWS>& R, int it1_start, int it2_start) {
    const Ctx& c = f.c;
    const int tid = threadIdx.x, b = XL ? (int)(blockIdx.x / JSLP_XL_SPREAD) : (int)blockIdx.x, lane = tid & 63, wv = tid >> 6;
    // XL (XCD-local build): all <= 32 workgroups sit on ONE XCD (checked at launch: k_simplex_resident's census), whose L2 is their
    // common point of coherence -- summaries, candidate rows and row flags leave as PLAIN stores (they stay in that L2; `sc1` stores
    // would drop the line and send the readers to memory) and are read with `sc1` loads (L1-bypassing, L2-served); no write-back
    // fence anywhere (tools/micro/xcd_handoff_bench.hip flavour 1: 1.9 k cycles per 32 -> 32 exchange against 8.3 k chip-wide)
    constexpr int ST_AUX = XL ? 0 : 16;           // aux of the hand-off stores: 16 = sc1 (write-through to memory)
    constexpr bool TAGGED = XL;                   // rows travel with their tags: 16 bytes per double
    constexpr bool CKS = !TAGGED && JSLP_PIPE_ROW_CHECKSUM != 0 && CPT <= JSLP_PIPE_ROW_CHECKSUM_MAXCPT;  // checksummed hand-over of the candidate rows (see JSLP_PIPE_ROW_CHECKSUM)
    constexpr int HAWV = 1;
    constexpr bool NPUB = false;  // (phase 1 knows its entering column only once the pivot row has arrived: the row travels as it is)
    const double quot_p = 1.0, pub_k0 = 0.0;
    const int pub_pc = 0;
    double fq = 0.0, fnv = 0.0;
    (void)quot_p; (void)pub_k0; (void)pub_pc; (void)fq; (void)fnv;
    // the ratio test's transposition (entry i of the entering column from the ONE lane that holds it to lane i) through LDS: as `x = lane
    // == i ? readlane(a[i][j]) : x` the compiler precomputes the 64-bit lane masks, spills them and pays two reloads, two moves and two
    // selects per row on top of the readlanes -- in the one wave the summary waits for
    constexpr bool S_LDS = XL || JSLP_PIPE_S_VIA_LDS != 0;
    // the pending pivot's row update in the XCD-local build's form (JSLP_XL_UPDATE_PASS: one ballot for the row gate, readlane multipliers,
    // the special rows fixed up once per pivot) instead of JSLP_PIPE_UPDATE_ROW's ~55 instructions per row
    constexpr bool UPD_NEW = XL || (CPT <= 4 && !(OPT && ROWS > 8));  // (the 6- / 8-column geometries and the tall OPT build would spill 14-94 VGPRs with it)
    // candidate rows in the publication buffer (chip-wide builds): pair j of lane t at ((j / 2) * THREADS + t) * 16 inside the
    // workgroup's slot -- a wave's store of one pair is 1 KB of whole lines (with lane t's CPT columns adjacent, as they sit in the
    // tableau, the 512-thread geometries wrote 16 bytes into each of 64 lines per instruction: the partial-line writers of round 3);
    // only lane t of the other workgroups ever reads what lane t wrote, so the permutation is invisible outside these two loops.
    // Measured (r04_q, pivots/s, adjacent -> permuted): 2001 x 4001 `<512,8,8>` 88.5 k -> 113.2 k, 3001 x 3001 `<512,6,12>` 102.0 k ->
    // 106.4 k, but 4001 x 2001 `<512,4,16>` 114.4 k -> 111.6 k (two adjacent pairs per lane are half a line already; apart they are
    // two requests): permuted from 6 columns per lane up
    constexpr bool PERM = CPT >= 6;
    // (the 2- and 4-column geometries keep round 3's addressing to the letter -- slots ld doubles apart: with the constant stride the
    //  tall `<512,4,16>` instance, at its register limit, came out 2 % slower, skewed or not)
    const int SLOT = PERM ? THREADS * CPT * 8 + JSLP_PUB_SKEW : f.c.ld * 8;
    constexpr int PAIR_STEP = PERM ? THREADS * 16 : 16;           // bytes from a lane's pair j to its pair j + 2
    const int lane_off = PERM ? tid * 16 : tid * CPT * 8;          // ... and where its first pair sits in the slot
    constexpr bool CM_AT_N = OPT && ROWS > 8;  // (see phase 2)
    constexpr int POLLWV = 1;  // the one polling wave (wave 0 folds the next summary out of column 0; the last wave commits)
    const int ld = c.ld, W = c.W;
    const double precision = c.precision;
    const int c0 = tid * CPT;
    const bool colok = c0 < ld;
    const int r_begin = b * f.rpb, r_end = min(f.H, r_begin + f.rpb);
    double (&a)[ROWS][CPT] = R.a;
    double (&r0)[CPT] = R.r0;
    typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
    const int pub_stride = (int)((const char*)f.rows_pub[1] - (const char*)f.rows_pub[0]);
    const auto rsrc_rows = __builtin_amdgcn_make_buffer_rsrc(f.rows_pub[0], 0, pub_stride + (TAGGED ? f.G * ld * 16 : f.G * (CPT >= 6 ? THREADS * CPT * 8 + JSLP_PUB_SKEW : ld * 8)), 0x00020000);
    const auto rsrc_g16 = __builtin_amdgcn_make_buffer_rsrc(f.gran16, 0, JSLP_R_REC_OFF + JSLP_R_REC_WORDS * 8, 0x00020000);  // (the summary granules, the row-flag copies and the row records: one descriptor)
#ifdef JSLP_DEBUG_RESIDENT
    u64_t (&rt_acc)[8] = R.rt_acc;
    u64_t& rt_prev = R.rt_prev;
#endif
    double p[CPT];
    unsigned nzm = 0;
    int pr_p = 0, pc_p = 0, par_p = 0;
    bool pend = false;
    bool cpend = false;  // the pending pivot's global commit has not been issued yet (JSLP_PIPE_COMMIT_GLOBAL)
#pragma unroll
    for (int j = 0; j < CPT; j++) p[j] = 0.0;
    unsigned& efetch = R.efetch;  // row fetches of this workgroup so far (uniform; one count for both phases: sm.okbad only grows)
    bool done = false;  // phase 1 is over: no row below -precision

    if (tid == 0) sm.cyc_filter_on = R.hist_n == 0 ? 1 : 0;  // (see phase 2)
    for (int i = tid; i < JSLP_PIPE_CYCBITS / 32; i += THREADS) cyc_bits(sm)[i] = 0u;
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < ROWS; i++) sm.rhsb[i] = a[i][0];
        sm.p_val = 0; sm.p_col = 0x7fffffff;
    }
    __syncthreads();

    while (R.end_code == 0) {
        const bool has_pc_p = pend && colok && pc_p >= c0 && pc_p < c0 + CPT;
        if ((R.it1 - it1_start) + (R.it2 - it2_start) >= F_ITERS) { R.end_code = 4; break; }
        if ((CHK && c.check_cycles) && !(R.hist_n < (f.hist_all ? JSLP_PIPE_GHIST : JSLP_PIPE_LHIST) && R.hist_n < c.hist_cap)) { R.end_code = 8; break; }
        const unsigned epoch = R.epoch;
        const int par = epoch & 1;
        const unsigned tag = epoch + 1;
        if (F_TEST_ABORT >= 0 && (int)epoch == F_TEST_ABORT && b == f.G - 1) {
            if (tid == 0) AG_STORE(f.abort_flag, 1u);
            R.end_code = 5;
            break;
        }
        RT_MARK(7);
        // ---- S: my most negative RHS below -precision (simplex.ts:39-49): lanes 0..ROWS-1 of wave 0 on the LDS copy of column 0 ----
        if (wv == 0) {
            const int r = r_begin + lane;
            u64_t key = KI_NONE_KEY;
            if (lane < ROWS) {
                const double v = sm.rhsb[lane];
                if (r >= 1 && r < r_end && v < -precision) key = key_asc(v);  // unsigned order of the keys = numeric order: min = most negative
            }
            KI bk;
            bk.k = key; bk.i = key != KI_NONE_KEY ? r : 0x7fffffff; bk.pad = 0;
            bk = ki_min(bk, ki_dpp<0xB1>(bk));
            bk = ki_min(bk, ki_dpp<0x4E>(bk));
            bk = ki_min(bk, ki_dpp<0x141>(bk));
            if (ROWS > 8) bk = ki_min(bk, ki_dpp<0x140>(bk));
            bk = ROWS > 16 ? ki_min(ki_readlane(bk, 0), ki_readlane(bk, 16)) : ki_readlane(bk, 0);
            if (lane == 0) {
                const bool have = bk.k != KI_NONE_KEY;
                const int row = have ? bk.i : 0;
                const u64_t qb = have ? bk.k : 0ull;
                v4u_t g;
                g.x = (unsigned)qb;
                g.y = tag;
                g.z = (unsigned)(qb >> 32);
                g.w = ((tag & 0xffffu) << 16) | (unsigned)row;
                __builtin_amdgcn_raw_buffer_store_b128(g, rsrc_g16, (par * JSLP_F_MAXG + b) * JSLP_G16_STRIDE, 0, ST_AUX);
                sm.pubrow = row;
            }
        }
        __syncthreads();
        const int pubrow = sm.pubrow;
        if (!CM_AT_N) JSLP_PIPE_COMMIT_GLOBAL();  // (the pending pivot's global maps + trace, while the summaries cross the fabric)
        RT_MARK(0);
        // ---- U + P: the pending pivot's row update, the candidate row published from inside the pass ----------------------------
        bool swept = true;
        const bool poller = wv == POLLWV;
        v4u_t g;
        g.x = 0; g.y = tag; g.z = 0; g.w = (tag & 0xffffu) << 16;
        if (UPD_NEW) {
            if (pend) JSLP_XL_UPDATE_PASS();
            if (pubrow != 0) {
                if (TAGGED) JSLP_XL_PUBLISH_ROW(pubrow);
                else JSLP_PUBLISH_ROW_PLAIN(pubrow);
            }
        } else {
        u64_t ck = 0;
        // (tests: the same hooks as JSLP_PUBLISH_ROW_PLAIN -- flag word first (=2), torn row (=4 / =6) -- for the builds that publish from inside this pass)
        const bool flag_first_o = CKS && __builtin_amdgcn_readfirstlane((int)(F_TEST_LATE == 2 && wv == 0)) != 0;
        const bool torn_o = CKS && __builtin_amdgcn_readfirstlane((int)((F_TEST_LATE == 4 || F_TEST_LATE == 6) && wv == 0)) != 0;
        const bool early_lane_o = !torn_o || ((lane & 1) == (F_TEST_LATE == 6 ? 1 : 0));
#pragma unroll
        for (int i0 = 0; i0 < ROWS; i0 += JSLP_PIPE_KCHUNK) {
            double kis[ROWS];
#pragma unroll
            for (int i = i0; i < i0 + JSLP_PIPE_KCHUNK && i < ROWS; i++) kis[i] = sm.colb[par_p][i];
#pragma unroll
            for (int i = i0; i < i0 + JSLP_PIPE_KCHUNK && i < ROWS; i++) {
                if (pend) JSLP_PIPE_UPDATE_ROW(i);
                if (pubrow != 0 && r_begin + i == pubrow && colok) {
                    const int off = PERM ? par * pub_stride + b * SLOT + lane_off : par * pub_stride + (b * ld + c0) * 8;
#pragma unroll
                    for (int j = 0; j < CPT; j += 2) {
                        if (c0 + j >= ld) continue;
                        const u64_t lo = (u64_t)__double_as_longlong(a[i][j]), hi = (u64_t)__double_as_longlong(a[i][j + 1]);
                        v4u_t v;
                        v.x = (unsigned)lo; v.y = (unsigned)(lo >> 32); v.z = (unsigned)hi; v.w = (unsigned)(hi >> 32);
                        if (!flag_first_o && early_lane_o) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_rows, off + (j >> 1) * PAIR_STEP, 0, ST_AUX);
                        if (CKS) JSLP_CK_PAIR(ck, lo, hi, j);
                    }
                }
            }
        }
        if (CKS && pubrow != 0) JSLP_CKS_RAISE_FLAG(ck);
        if (CKS && pubrow != 0 && (flag_first_o || torn_o)) {  // (tests only) what the hook held back leaves now, behind the flag word
            __builtin_amdgcn_s_sleep(127);
            const int ipub_o = __builtin_amdgcn_readfirstlane(pubrow - r_begin);
            if (colok && (flag_first_o || !early_lane_o)) {
#pragma unroll
                for (int i = 0; i < ROWS; i++)
                    if (i == JSLP_OPAQUE_SGPR(ipub_o)) {
                        const int off = PERM ? par * pub_stride + b * SLOT + lane_off : par * pub_stride + (b * ld + c0) * 8;
#pragma unroll
                        for (int j = 0; j < CPT; j += 2) {
                            if (c0 + j >= ld) continue;
                            const u64_t lo = (u64_t)__double_as_longlong(a[i][j]), hi = (u64_t)__double_as_longlong(a[i][j + 1]);
                            v4u_t v;
                            v.x = (unsigned)lo; v.y = (unsigned)(lo >> 32); v.z = (unsigned)hi; v.w = (unsigned)(hi >> 32);
                            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_rows, off + (j >> 1) * PAIR_STEP, 0, ST_AUX);
                        }
                    }
            }
        }
        }
        pend = false;
        RT_MARK(2);
        // ---- C: gather ----------------------------------------------------------------------------------------------------
        if (poller) {  // one wave, four summaries per lane (see phase 2)
            unsigned spins = 0;
            v4u_t gq[JSLP_F_MAXG / 64];
#if JSLP_PIPE_POLL_MISSING_ONLY
            unsigned have = 0;  // bit q: my look q has its summary (a look beyond the grid holds a copy of my own workgroup's: it is there)
#pragma unroll
            for (int q = 0; q < JSLP_F_MAXG / 64; q++) {
                gq[q] = g;
                if (lane + 64 * q >= f.G) have |= 1u << q;
            }
#endif
            for (;;) {
                bool ok = true;
#if JSLP_PIPE_POLL_MISSING_ONLY
                // Round 6: only the looks whose summary has NOT arrived go out again.  Most summaries are there at the first look, and a look is an sc1 load
                // that memory serves: 251 workgroups x 251 granules per round trip otherwise -- traffic that delays the very stores it waits for
                // (184.7 k -> 193.3 k pivots/s on config 3a; two staggered sets of looks, i.e. MORE traffic, measured 158 k: profiles/r06_poll_traffic.md)
#pragma unroll
                for (int q = 0; q < JSLP_F_MAXG / 64; q++)
                    if (!((have >> q) & 1u)) gq[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_g16, (par * JSLP_F_MAXG + lane + 64 * q) * JSLP_G16_STRIDE, 0, 16);
#pragma unroll
                for (int q = 0; q < JSLP_F_MAXG / 64; q++)
                    if (gq[q].y == tag && (gq[q].w >> 16) == (tag & 0xffffu)) have |= 1u << q;
                ok = have == (1u << (JSLP_F_MAXG / 64)) - 1u;
#else
#pragma unroll
                for (int q = 0; q < JSLP_F_MAXG / 64; q++) {
                    gq[q] = g;
                    if (lane + 64 * q < f.G) gq[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_g16, (par * JSLP_F_MAXG + lane + 64 * q) * JSLP_G16_STRIDE, 0, 16);
                }
#pragma unroll
                for (int q = 0; q < JSLP_F_MAXG / 64; q++) ok = ok && gq[q].y == tag && (gq[q].w >> 16) == (tag & 0xffffu);
#endif
                if (__all(ok) && !JSLP_HOST_ABORT_FORCE_RETRY(spins)) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                JSLP_HOST_ABORT_IN_SPIN(spins, swept);
                if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) { swept = false; break; }
                if (spins > F_SPIN) { if (lane == 0) AG_STORE(f.abort_flag, 1u); swept = false; break; }
            }
            KI x = ki_none();
#pragma unroll
            for (int q = 0; q < JSLP_F_MAXG / 64; q++) {
                const int row = (int)(gq[q].w & 0x7fffu);
                KI y;
                y.k = row != 0 ? ((u64_t)gq[q].x | ((u64_t)gq[q].z << 32)) : KI_NONE_KEY;
                y.i = row != 0 ? row : 0x7fffffff;
                y.pad = 0;
                x = ki_min(x, y);
            }
            x = ki_wave_min(x);
            if (lane == 0) { sm.part_k[0] = x.k; sm.part_r[0] = x.k == KI_NONE_KEY ? 0 : x.i; }
        }
        RT_MARK(1);
        if (!TAGGED && !CKS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: my stores of the candidate row have reached the L2
        const int all_swept = __syncthreads_and(swept ? 1 : 0);
        if (!all_swept) { R.end_code = 5; break; }
        RT_MARK(3);
        // ---- D ---------------------------------------------------------------------------------------------------------------
        const int pr = sm.part_r[0];  // (the polling wave's verdict)
        if (pr == 0) { done = true; break; }  // no violated row: feasible (simplex.ts:51-54); uniform
        const bool leaving_unr = UNR && sm.lunr[sm.lvibr[pr]] != 0;  // (see phase 2)
        // ---- the winner releases its row (see phase 2) ---------------------------------------------------------------------------
        const int bw = pr / f.rpb;
        if (!TAGGED && !CKS && bw == b && tid < THREADS / 64) {  // (every wave's stores reached the L2 before the barrier that closed the gather)
            if (XL) {
                *reinterpret_cast<volatile u64_t*>(f.rowflagc[par] + tid * JSLP_F_MAXG + b) = (u64_t)tag;
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                AG_STORE(f.rowflagc[par] + tid * JSLP_F_MAXG + b, (u64_t)tag);
            }
        }
        // ---- E: the pivot row (= the winner's candidate row) -------------------------------------------------------------------
        const int off_in = PERM ? par * pub_stride + bw * SLOT + lane_off : par * pub_stride + (bw * ld + c0) * 8;
        double pv[CPT];
#pragma unroll
        for (int j = 0; j < CPT; j++) pv[j] = 0.0;
        {
            // (every wave its own look at the flag and its own repeats: see phase 2)
            if (TAGGED) {
                double q_unused = 0.0;
                JSLP_XL_FETCH_ROW(0, q_unused);
                (void)q_unused;
            } else if (CKS) {
                JSLP_CKS_FETCH_ROW();
            } else {
            efetch += 1;
            unsigned spins = 0;
            for (;;) {
                if (__builtin_amdgcn_readfirstlane((int)(F_TEST_LATE != 0 && wv == 0))) __builtin_amdgcn_s_sleep(127);  // (tests: the skew that used to break the fetch; a scalar branch: s_sleep ignores EXEC)
                const u64_t flag = AG_LOAD(f.rowflagc[par] + wv * JSLP_F_MAXG + bw);
                if ((unsigned)flag == tag) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                bool dead = false;
                if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) dead = true;
                if (spins > F_SPIN) { if (lane == 0) AG_STORE(f.abort_flag, 1u); dead = true; }
                if (dead) { if (lane == 0) atomicMax(&sm.okbad, efetch); break; }
            }
            asm volatile("" ::: "memory");
            if (colok) {  // the row, now that MY look at the flag found it up (the flag comes late by construction -- only the winner raises
                          // it, after the decision --, so loading the row next to it would only add 4 MB of dead traffic per look)
#pragma unroll
                for (int j = 0; j < CPT; j += 2) {
                    if (c0 + j >= ld) continue;
                    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_rows, off_in + (j >> 1) * PAIR_STEP, 0, 16);
                    pv[j] = __longlong_as_double((long long)((u64_t)v.x | ((u64_t)v.y << 32)));
                    pv[j + 1] = __longlong_as_double((long long)((u64_t)v.z | ((u64_t)v.w << 32)));
                }
            }
            }  // !XL
            __syncthreads();
            if (sm.okbad == efetch) R.end_code = 5;
        }
        if (R.end_code == 5) break;
        RT_MARK(4);
        // ---- E1: entering column = max -cost / coefficient over coefficient < -precision, first index on ties (simplex.ts:56-71):
        //      my best, the wave's maximum (DPP), one LDS atomic per wave; then the first column among the lanes holding that value -----
        double bq = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const int col = c0 + j;
            const double coef = pv[j];
            if (col >= 1 && col < W && ((UNR && ((R.unr >> j) & 1u)) || coef < -precision)) {
                const double quo = -r0[j] / coef;
                // (my columns ascend: ties keep the earlier one.  UNR: an unrestricted column may have a zero coefficient -- the
                //  reference's `maxQuotient < quotient` from -Infinity lets +Infinity win, never NaN or -Infinity)
                const bool take = UNR ? (bq < quo) : (bi == 0 || bq < quo);
                bq = take ? quo : bq;
                bi = take ? col : bi;
            }
        }
        const u64_t qkey = bi != 0 ? key_asc(bq) : 0ull;  // (key_asc > 0 for every double; -0 folded into +0: simplex.ts compares them equal)
        {
            const u64_t wmax = u64_wave_max(qkey);
            if (lane == 0 && wmax != 0ull) atomicMax(&sm.p_val, wmax);
        }
        __syncthreads();
        const u64_t wkey = sm.p_val;
        if (wkey == 0ull) { R.end_code = 7; break; }  // infeasible (simplex.ts:73-76); uniform
        if (bi != 0 && qkey == wkey) atomicMin(&sm.p_col, bi);
        __syncthreads();
        const int pc = sm.p_col;
        const bool has_pc = colok && pc >= c0 && pc < c0 + CPT;
        if (has_pc) {  // the lane that holds the column: quot = A[pr, pc], k0 = cost[pc], and my rows' entries of the column
#pragma unroll
            for (int j = 0; j < CPT; j++)
                if (pc == c0 + j) {
                    sm.xq[0] = pv[j];
                    sm.xq[1] = r0[j];
#pragma unroll
                    for (int i = 0; i < ROWS; i++) sm.colb[par][i] = a[i][j];
                    if (OPT) {
#pragma unroll
                        for (int o = 0; o < JSLP_R_MAXOPT; o++) sm.ook[0][o] = R.oo[o][j];
                    }
                }
        }
        __syncthreads();
        const double quot = sm.xq[0], k0 = sm.xq[1];
        double ook[JSLP_R_MAXOPT];
#pragma unroll
        for (int o = 0; o < JSLP_R_MAXOPT; o++) ook[o] = OPT ? sm.ook[0][o] : 0.0;
        if (tid == 0) { sm.p_val = 0; sm.p_col = 0x7fffffff; }  // (everybody has read them; the next round is barriers away)
        if ((CHK && c.check_cycles)) {  // simplex.ts:78-93 by every workgroup, on its own LDS history
            if (tid == 0) {
                const int2 pair = make_int2(sm.lvibr[pr], sm.lvibc[pc]);
                if (R.hist_n < JSLP_PIPE_LHIST) sm.lhist[R.hist_n] = pair;
                if (f.hist_all) f.hist_all[(size_t)b * JSLP_PIPE_GHIST + R.hist_n] = pair;  // my own copy of the whole history
                if (b == 0) JSLP_PIPE_HIST_GLOBAL(pair);
                sm.cyc_need = sm.cyc_filter_on ? cyc_pair_seen(sm, pair) : 1;
            }
            __syncthreads();
            R.hist_n += 1;
            if (sm.cyc_need != 0) {  // (uniform)
                if (cyc_suffix_is_square(sm.lhist, f.hist_all ? f.hist_all + (size_t)b * JSLP_PIPE_GHIST : nullptr, R.hist_n, make_int2(sm.lvibr[pr], sm.lvibc[pc]), sm.f.red)) { R.end_code = 3; break; }
            }
        }
        // ---- N: normalised pivot row (simplex.ts:352-364); the tiny entries simplex.ts:381-383 zeroes as soon as ANY other row
        //      is eliminated need a chip-wide answer in the rare pivot that has them -----------------------------------------------
        nzm = 0;
        int tiny = 0;
        if (colok) {
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const int col = c0 + j;
                const double val = pv[j];
                double v = 0.0;
                if (col < W) {
                    const bool innz = nonzero16(val);
                    v = innz ? val / quot : 0.0;
                    if (col == pc) v = 1.0 / quot;  // (as a real branch -- one lane of the workgroup needs this division -- measured no faster: r04_u)
                    if (innz && !nonzero16(v) && v != 0.0) tiny |= 1 << j;
                }
                p[j] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < CPT; j++) p[j] = 0.0;
        }
        if (__syncthreads_or(tiny)) {
            int local_any = 0;
#pragma unroll
            for (int i = 0; i < ROWS; i++) {
                const int r = r_begin + i;
                if (r < r_end && r != pr && nonzero16(sm.colb[par][i])) local_any = 1;
            }
            if (b == 0 && nonzero16(k0)) local_any = 1;  // (row 0, the cost row, counts: simplex.ts:367 runs r from 0)
            const int gany = global_or(f, par, tag, local_any, sm, b);
            if (gany < 0) { R.end_code = 5; break; }
            if (gany != 0) {
#pragma unroll
                for (int j = 0; j < CPT; j++)
                    if (tiny & (1 << j)) p[j] = 0.0;
            }
        }
#pragma unroll
        for (int j = 0; j < CPT; j++) nzm |= nonzero16(p[j]) ? (1u << j) : 0u;
        if (OPT) {  // the optional objective rows follow every pivot (simplex.ts:394-412)
#pragma unroll
            for (int o = 0; o < JSLP_R_MAXOPT; o++)
                if (o < c.n_opt) {
#pragma unroll
                    for (int j = 0; j < CPT; j++)
                        if (c0 + j < W) R.oo[o][j] = oo_cell_after(R.oo[o][j], ook[o], c0 + j, pc, quot, p[j]);
                }
        }
        // column 0 of my rows after this pivot, and the pivot column's own new entries: lanes 0..ROWS-1 of wave 0 (the same
        // lanes fold the next summary out of it at the top of the loop: no barrier in between)
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
        if (tid == THREADS - 64) {
            const int leaving = sm.lvibr[pr], entering = sm.lvibc[pc];
            sm.lvibr[pr] = entering;
            sm.lvibc[pc] = leaving;
            if (b == 0) {
                if (CM_AT_N) JSLP_PIPE_COMMIT_GLOBAL_NOW();
                else { sm.cm_ent = entering; sm.cm_leav = leaving; }  // (the global maps and the trace follow one iteration later: JSLP_PIPE_COMMIT_GLOBAL)
            }
        }
        if (UNR && has_pc) {
#pragma unroll
            for (int j = 0; j < CPT; j++)
                if (pc == c0 + j) R.unr = (R.unr & ~(1u << j)) | ((leaving_unr ? 1u : 0u) << j);
        }
        R.trace_n += 1;
        R.it1 += 1;
        R.epoch = epoch + 1;
        pend = true; pr_p = pr; pc_p = pc; par_p = par;
        cpend = !CM_AT_N;
        RT_MARK(5);
    }
    JSLP_PIPE_COMMIT_GLOBAL();  // (a basis change whose global half is still pending)
    if (pend && R.end_code != 5) {
        const bool has_pc_p = colok && pc_p >= c0 && pc_p < c0 + CPT;
        if (UPD_NEW) {
            (void)has_pc_p;
            JSLP_XL_UPDATE_PASS();
        } else
#pragma unroll
        for (int i0 = 0; i0 < ROWS; i0 += JSLP_PIPE_KCHUNK) {
            double kis[ROWS];
#pragma unroll
            for (int i = i0; i < i0 + JSLP_PIPE_KCHUNK && i < ROWS; i++) kis[i] = sm.colb[par_p][i];
#pragma unroll
            for (int i = i0; i < i0 + JSLP_PIPE_KCHUNK && i < ROWS; i++) JSLP_PIPE_UPDATE_ROW(i);
        }
    }
    if (done) {  // simplex.ts:14-23, 102: phase 2 starts with a fresh history
        R.hist_n = 0;
        R.epoch += 1;
    }
}
#undef JSLP_PIPE_UPDATE_ROW
#undef JSLP_OPAQUE_SGPR
#undef JSLP_XL_UPDATE_PASS
#undef JSLP_XL_PUBLISH_ROW
#undef JSLP_PUBLISH_ROW_PLAIN
#undef JSLP_XL_FETCH_ROW
#undef JSLP_CKS_FETCH_ROW
#undef JSLP_CKS_ISSUE_LOOK
#undef JSLP_CKS_RAISE_FLAG
#undef JSLP_CKS_RAISE_REC
#undef JSLP_REC_OFF
#undef JSLP_CK_QMIX
#undef JSLP_RT_RETRY
#undef JSLP_PIPE_COMMIT_GLOBAL
#undef JSLP_PIPE_SWAP_LDS_MAPS
#undef JSLP_PIPE_HIST_GLOBAL
#undef JSLP_PIPE_COMMIT_GLOBAL_NOW
